// K1, short alignments (full matrices: 150 x 150 short reads) as a skewed pipeline: see ksw_pipe_core.h for the scheme and why it is
// exact.  One wavefront = NG = 64 / G groups of G lanes; every group works through the `np` alignments of its column of the pipe's
// id table, all of ONE geometry (qlen, tlen), a lane moving on to the next alignment as soon as its block has left the matrix.
// Results (backtrace layout, score, status) are those of ksw_extd2_wave_kernel<10 | 8 | 16>, which it replaces for such batches;
// the walk stays with ksw_backtrack_kernel.
//
// Per step a lane issues the DP of its 16 cells (gdw_compute, as every other form), one ds_read_u8 for the query byte its block meets
// next (the group's query sits in LDS: no lane depends on the lane below for it, and the row loop issues no vector-memory load),
// and one 16-byte backtrace store while its block is inside the matrix.  Target blocks, queries and the per-alignment descriptor are
// staged through LDS one alignment ahead, in two buffers: alignment n + 1 is fetched when the last lane of the group has moved on to
// alignment n (nobody reads the buffer of n - 1 any more), 16 (G - 1) steps into the period -- long before lane 0 needs it.
#pragma once
#include "ksw_wave.hip.h"
#include "ksw_pipe_core.h"

// one wavefront's work: np alignments per group, ids at task_ids[id_off + n * NG + g] (-1: none), all of geometry qlen x tlen
struct PipeWave { int32_t id_off, qlen, tlen, np, row_bytes, pad[3]; };

template <bool DUAL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void ksw_extd2_pipe_kernel(const KswTask *__restrict__ tasks, const int32_t *__restrict__ task_ids,
                                                           const PipeWave *__restrict__ pipes, int n_pipes,
                                                           const uint8_t *__restrict__ qseq, const uint8_t *__restrict__ tseq,
                                                           uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                           int32_t *__restrict__ score_out, WaveK K)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[2 * GDP_BUF_BYTES];
	const int lane = threadIdx.x & 63;
	const int pw = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
	if (pw >= n_pipes) return;
	const int id_off = __builtin_amdgcn_readfirstlane(pipes[pw].id_off), np = __builtin_amdgcn_readfirstlane(pipes[pw].np);
	const int row_bytes = __builtin_amdgcn_readfirstlane(pipes[pw].row_bytes);
	const PipeGeo Gm = gd_pipe_geo(__builtin_amdgcn_readfirstlane(pipes[pw].qlen), __builtin_amdgcn_readfirstlane(pipes[pw].tlen));
	const int G = Gm.G, NG = Gm.NG, P = Gm.P, qlen = Gm.qlen, tlen = Gm.tlen;
	// lane -> (group, block); the 64 - NG * G (< G) spare lanes shadow the first blocks of group 0 without ever being live
	int grp = 0, sub = lane;
	while (sub >= G) sub -= G, ++grp; // (once per wavefront)
	const bool spare = grp >= NG;
	if (spare) grp = 0;
	const u32 m_first = sub == 0 ? ~0u : 0u;
	const int nvalid = spare ? 0 : gdp_valid_rows(Gm, sub);

	// alignment nn of every group into buffer nn & 1: each lane its 16 target and 16 (+ 16 for two lanes: the zero padding up to
	// P + 1 bytes and beyond) query bytes, lane 0 of the group the descriptor {backtrace offset, task id, live}
	auto fetch = [&](const int nn) __attribute__((always_inline)) {
		int tid_ = -1, live_ = 0;
		int64_t qo = 0, to = 0, bo = 0;
		if (!spare && nn < np) tid_ = task_ids[id_off + nn * NG + grp];
		if (tid_ >= 0) {
			live_ = status[tid_] == GD_ST_PENDING; // (else the exact-match pre-filter answered this one)
			qo = tasks[tid_].qoff, to = tasks[tid_].toff, bo = tasks[tid_].bt_off;
		}
		u32 tw[4] = {0, 0, 0, 0}, qw[4] = {0, 0, 0, 0}, qx[4] = {0, 0, 0, 0};
		if (live_) {
#pragma unroll
			for (int b = 0; b < 16; ++b) {
				const int t = 16 * sub + b, t2 = 16 * (sub + G) + b;
				if (t < tlen) tw[b >> 2] |= (u32)tseq[to + t] << (8 * (b & 3));
				if (t < qlen) qw[b >> 2] |= (u32)qseq[qo + t] << (8 * (b & 3));
				if (sub < 2 && t2 < qlen) qx[b >> 2] |= (u32)qseq[qo + t2] << (8 * (b & 3));
			}
		}
		if (!spare) {
			uint8_t *B = lds + (nn & 1) * Gm.BS;
			*reinterpret_cast<uint4 *>(B + Gm.TOFF + grp * Gm.TS + 16 * sub) = make_uint4(tw[0], tw[1], tw[2], tw[3]);
			*reinterpret_cast<uint4 *>(B + grp * Gm.QS + 16 * sub) = make_uint4(qw[0], qw[1], qw[2], qw[3]);
			if (sub < 2) *reinterpret_cast<uint4 *>(B + grp * Gm.QS + 16 * (sub + G)) = make_uint4(qx[0], qx[1], qx[2], qx[3]);
			if (sub == 0) *reinterpret_cast<uint4 *>(B + Gm.DOFF + grp * 16) = make_uint4((u32)(uint64_t)bo, (u32)((uint64_t)bo >> 32), (u32)tid_, (u32)live_);
		}
		__syncthreads(); // (one wavefront per workgroup: orders the LDS writes before the reads of the other lanes)
	};

	WaveLane L;
	{
		const u32 z[4] = {0, 0, 0, 0};
		gdp_start(L, K, z, sub);
	}
	int left = 0, Rf = 0, tid = -1;
	u32 qaddr = 0;
	uint8_t *p_row = bt;
	bool any_tn = false;
	// row r - 1 values of the lane below, taken at the END of a step: the lane below may start its next alignment before the step
	// that consumes them (its last row on this alignment is the very row the cell above it still needs: ksw_pipe_core.h)
	u32 pX = 0, pV = 0, pX2 = 0;
	fetch(0);
	for (int n = 0; n <= np; ++n) { // period n: lane 0 of every group starts alignment n (n == np: the last alignments drain)
		for (int rA = 0; rA < P;) {
			const int jsw = rA >> 4; // rows [16 jsw, 16 jsw + 15] of the period: block jsw changes alignment on the first of them
			if (jsw < G) {
				if (sub == jsw) {
					if (left <= 0 && tid >= 0 && sub == Gm.mlast) score_out[tid] = Rf >> 3, status[tid] = GD_ST_DONE; // (tid >= 0: a live alignment, all rows done)
					const uint8_t *B = lds + (n & 1) * Gm.BS;
					const uint4 d = *reinterpret_cast<const uint4 *>(B + Gm.DOFF + grp * 16);
					const uint4 t4 = *reinterpret_cast<const uint4 *>(B + Gm.TOFF + grp * Gm.TS + 16 * sub);
					const u32 tb[4] = {t4.x, t4.y, t4.z, t4.w};
					gdp_start(L, K, tb, sub);
					const bool live = n < np && d.w != 0 && !spare;
					tid = live ? (int)d.z : -1;
					left = live ? nvalid : 0;
					Rf = 0;
					qaddr = (u32)((n & 1) * Gm.BS + grp * Gm.QS);
					p_row = bt + (((uint64_t)d.y << 32) | d.x) + (size_t)sub * 16 + (size_t)(16 * sub) * (size_t)row_bytes;
				}
				any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0;
				if (jsw == G - 1) {
					if (n == np) return; // every lane has handed in its last alignment
					fetch(n + 1);
				}
			}
			const int rows_here = P - rA < 16 ? P - rA : 16;
			const int r_stop = rA + rows_here;
#pragma unroll 1
			for (bool first = true; rA < r_stop; ++rA, first = false) {
				// (1) the query byte this lane's first cell meets
				const u32 qb = lds[qaddr];
				++qaddr;
				// (2) the cell t == r of the lanes in their first 16 rows (all of them on lane 0's alignment, at row rA), query, scores
				WaveRow W;
				W.r = rA, W.st0 = 0, W.en0 = 0, W.st_ = 0, W.en_ = 0, W.up = 0;
				W.use_array = 0, W.set_tr = 1, W.ukey = gdw_edge_key(K, rA), W.v1key = W.ukey;
				W.m_first_valid = 1, W.m_first = m_first;
				if (jsw < G) gdw_reset_tr(L, K, W);
				gdp_query_scores(L, K, qb, any_tn);
				// (3) the 16 cells
				u32 out[4];
				gdw_compute<DUAL>(L, K, W, pX, pV, pX2, out);
				pX = gdw_ror1<64>(L.X[7]), pV = gdw_ror1<64>(L.V[7]), pX2 = DUAL ? gdw_ror1<64>(L.X2[7]) : 0u; // for the next step
				if (left > 0) *reinterpret_cast<uint4 *>(p_row) = make_uint4(out[0], out[1], out[2], out[3]);
				p_row += row_bytes;
				// (4) score trackers
				L.R += gdw_lo(L.V[0]) - K.B1;
				if (first && jsw >= 1 && jsw < G) {
					const int h = (int)gdw_ror1<64>((u32)gdw_track_handoff(L));
					if (sub == jsw) L.R = h + gdw_lo(L.U[0]);
				}
				if (sub == Gm.mlast) {
					if (left == qlen) Rf = gdw_track_to_slot(L, Gm.sl);
					else if (left > 0 && left < qlen) Rf += gdw_cell(L.V, Gm.sl) - K.B1;
				}
				--left;
			}
		}
	}
}

static inline void gd_launch_pipe(const KswTask *tasks, const int32_t *ids, const PipeWave *pipes, int n_pipes, const uint8_t *q, const uint8_t *t,
                                  uint8_t *bt, int32_t *status, int32_t *score, KswConst C, hipStream_t s, bool single)
{
	if (n_pipes <= 0) return;
	WaveK K;
	gdw_make_consts(C, K);
	if (single) hipLaunchKernelGGL((ksw_extd2_pipe_kernel<false>), dim3(n_pipes), dim3(64), 0, s, tasks, ids, pipes, n_pipes, q, t, bt, status, score, K);
	else hipLaunchKernelGGL((ksw_extd2_pipe_kernel<true>), dim3(n_pipes), dim3(64), 0, s, tasks, ids, pipes, n_pipes, q, t, bt, status, score, K);
}
