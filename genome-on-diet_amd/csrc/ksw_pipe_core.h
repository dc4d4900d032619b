// K1, short alignments as a SKEWED PIPELINE: the geometry and the per-lane pieces of ksw_extd2_pipe_kernel (ksw_pipe.hip.h), shared
// with the host lock-step emulator (tests/emul/pipe_emul.cpp).
//
// A 150 x 150 short-read alignment is a FULL matrix: its band (w >= max(qlen, tlen) - 1) never cuts anything, anti-diagonal r holds the
// cells t in [max(0, r - qlen + 1), min(r, tlen - 1)] -- one cell on the first row, 150 in the middle, one on the last.  The grouped
// kernels (ksw_extd2_wave_kernel<10>) give every 16-cell block of the target a lane for all qlen + tlen - 1 rows, so on average half
// of the lanes compute cells outside the matrix.  But block j (target positions 16 j .. 16 j + 15) only holds cells of the matrix on
// the rows r in [16 j, 16 j + 15 + qlen - 1]: qlen + 15 rows of the qlen + tlen - 1.  Here a lane therefore leaves an alignment as
// soon as its block has run out of the matrix and starts the same block of the group's NEXT alignment, P = qlen + 15 rows after it
// started the current one: lane j works on alignment n during the steps [n P + 16 j, (n + 1) P + 16 j).  A group of G = ceil(tlen / 16)
// lanes then finishes an alignment every P steps instead of every qlen + tlen - 1 (165 instead of 299 for 150 x 150), with at most two
// alignments in flight per group (16 (G - 1) < P): the low lanes on alignment n at row rA, the high lanes on n - 1 at row rA + P.
// What a lane receives from the lane below (DPP) belongs to its own alignment whenever the cell it feeds is inside the matrix:
//     cell (r, 16 j) reads (r - 1, 16 j - 1), a cell of the matrix iff r - 1 <= 16 j - 2 + qlen, and lane j - 1 leaves the alignment
//     after row 16 (j - 1) + P - 1 = 16 j - 2 + qlen: exactly then.
// Everything outside the matrix may hold anything: a cell outside never feeds a cell inside (its successors (r + 1, t) and
// (r + 1, t + 1) have the same or a larger query index when it lies below the matrix, and above it -- t > r -- the reference resets the
// cell t == r + 1 on the next row: SR/ksw2_extd2_sse.c:160-163), and the walk never leaves the matrix.  So this form needs no band
// arithmetic, no score selectors and no "first computed block" logic: lane 0 of a group always takes the boundary scalars, every other
// lane what the lane below hands over.
#pragma once
#include "ksw_wave_core.h"

struct PipeGeo {
	int qlen, tlen;
	int G, NG;       // lanes per alignment (one per 16-cell block of the target), alignments side by side in a wavefront
	int P;           // steps between two alignments of a group
	int mlast, sl;   // block and cell of the last target column
	int rend;        // last anti-diagonal
	int QS, TS, BS;  // LDS bytes per group for the query (zero-padded to P bytes and more) and the target; bytes of one buffer
	int TOFF, DOFF;  // offsets of the target blocks / the per-group descriptors inside a buffer (queries at 0)
};
#define GDP_BUF_BYTES 3584 // the largest BS (G == 2: 32 groups of 64 + 32 + 16 bytes)

// Does the pipeline take this geometry?  Full matrix (the band terms of SR/ksw2_extd2_sse.c:138-141 never bind), a target of 17..256
// bases (2..16 blocks), lengths close enough for "at most two alignments in flight per group" and for the padded query buffer.
static inline __host__ __device__ bool gd_pipe_geometry_ok(int qlen, int tlen, int w)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	if (tlen < 17 || tlen > 256 || qlen < 17) return false;
	if (w < qlen - 1 || w < tlen - 1) return false; // (exactly the bands that never bind: tests/emul/pipe_emul.cpp compares every row; a 151-base read at bw = 150 is one)
	return qlen - tlen <= 15 && tlen - qlen <= 15;
}

static inline __host__ __device__ PipeGeo gd_pipe_geo(int qlen, int tlen)
{
	PipeGeo g;
	g.qlen = qlen, g.tlen = tlen;
	g.G = (tlen + 15) >> 4, g.NG = 64 / g.G;
	g.P = qlen + 15;
	g.mlast = (tlen - 1) >> 4, g.sl = (tlen - 1) & 15;
	g.rend = qlen + tlen - 2;
	g.QS = 16 * (g.G + 2), g.TS = 16 * g.G;
	g.TOFF = g.NG * g.QS, g.DOFF = g.TOFF + g.NG * g.TS, g.BS = g.DOFF + g.NG * 16;
	return g;
}

// rows of block `sub` that hold cells of the matrix: r in [16 sub, min(16 sub + 15, tlen - 1) + qlen - 1]
GDW_HD int gdp_valid_rows(const PipeGeo &g, int sub)
{
	const int last_t = 16 * sub + 15 < g.tlen - 1 ? 16 * sub + 15 : g.tlen - 1;
	return last_t + g.qlen - 16 * sub;
}

// a lane starts its block of a new alignment: the reference's initial fill (SR/ksw2_extd2_sse.c:107,111-116); no query byte faces the
// block yet (its first cell meets query[0] on the row that follows)
GDW_HD void gdp_start(WaveLane &L, const WaveK &K, const u32 tb[4], int sub)
{
	L.blk = sub;
#pragma unroll
	for (int k = 0; k < 8; ++k) L.U[k] = K.uv0, L.V[k] = K.uv0, L.X[k] = K.cx, L.Y[k] = K.cy, L.X2[k] = K.cx2, L.Y2[k] = K.cy2;
#pragma unroll
	for (int g = 0; g < 4; ++g) L.Tb[g] = tb[g], L.Qc[g] = 0, L.Sb[g] = K.s0;
	L.tn = (tb[0] | tb[1] | tb[2] | tb[3]) & 0x04040404u;
	L.R = sub == 0 ? -K.qe8 : 0; // (block 0: the tracker of row 0 is v - (q + e); the other blocks take theirs from the block below)
}

// query window advance with the lane's own fresh byte (cell 0 of block `sub` faces query[r - 16 sub]), and all 16 scores anew
GDW_HD void gdp_query_scores(WaveLane &L, const WaveK &K, u32 qbyte, bool any_tn)
{
	L.Qc[3] = gdw_alignbyte(L.Qc[3], L.Qc[2], 3);
	L.Qc[2] = gdw_alignbyte(L.Qc[2], L.Qc[1], 3);
	L.Qc[1] = gdw_alignbyte(L.Qc[1], L.Qc[0], 3);
	L.Qc[0] = gdw_alignbyte(L.Qc[0], qbyte << 24, 3);
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		u32 x = L.Tb[g] ^ L.Qc[g];
		if (any_tn) x |= L.Tb[g] & ~(L.Qc[g] << 1) & 0x04040404u; // (see gdw_update_scores)
		L.Sb[g] = gdw_perm(K.lut_hi, K.lut_lo, x);
	}
}
