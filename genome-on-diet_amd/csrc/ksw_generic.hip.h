// K1 (generic form): banded dual-affine global alignment, one 64-lane wavefront per alignment, DP state in an
// LDS sliding window.  Takes ANY geometry whose anti-diagonal window fits the LDS ring (cap cells); it is the
// catch-all behind the register-resident kernels of ksw_wave.hip.h (wide ONT bands, tiny or degenerate inputs).
//
// Semantics follow SR/ksw2_extd2_sse.c:34-401 literally, including everything SURVEY.md "Notes K1" lists:
//   - the computed window of anti-diagonal r is the 16-aligned [st0/16*16, en0|15]; padding cells are computed
//     and stored and feed in-band cells of later anti-diagonals (:138-147)
//   - s[] is rewritten only for [st0, st0+16*ceil((en0-st0+1)/16)); other computed cells see stale s (:166-180)
//   - boundary scalars x1/x21/v1 (:149-159) and the u/y/y2 reset of cell t == r (:160-163)
//   - 8-bit wrapping arithmetic, signed compares, direction byte with strict '>' priority a,b,a2,b2 (:235-273)
//   - approximate-max walk H0 (:367-383); score only if the last anti-diagonal reaches t = tlen-1 (:381)
//
// State arrays are rings of `cap` bytes indexed by t & (cap-1).  A ring slot is (re)initialised to the
// reference's initial fill (:111-116, s = 0 from kcalloc :107) when its 16-cell block is first touched, which is
// equivalent to the reference's flat arrays because the upper edge of the touched region never moves down.
#pragma once
#include <hip/hip_runtime.h>
#include "ksw_common.h"

__device__ __forceinline__ int8_t gd_w8(int v) { return (int8_t)(uint8_t)(v & 0xff); }

__global__ __launch_bounds__(64) void ksw_extd2_generic_kernel(const KswTask *__restrict__ tasks,
                                                               const int32_t *__restrict__ task_ids,
                                                               const uint8_t *__restrict__ qseq,
                                                               const uint8_t *__restrict__ tseq,
                                                               uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                               int32_t *__restrict__ score_out, KswConst K, int cap)
{
	extern __shared__ int8_t gd_lds[];
	const int lane = threadIdx.x;
	const int tid = task_ids[blockIdx.x];
	if (status[tid] != GD_ST_PENDING) return; // exact-match pre-filter already answered
	const KswTask T = tasks[tid];
	const int mask = cap - 1;
	int8_t *u = gd_lds, *v = u + cap, *x = v + cap, *y = x + cap, *x2 = y + cap, *y2 = x2 + cap, *s = y2 + cap;
	const uint8_t *query = qseq + T.qoff, *target = tseq + T.toff;
	const int qlen = T.qlen, tlen = T.tlen;
	const int w = T.w < 0 ? (tlen > qlen ? tlen : qlen) : T.w;
	const int TL16 = (tlen + 15) / 16 * 16;
	uint8_t *p = bt + T.bt_off;
	const size_t row_bytes = (size_t)T.row_bytes;
	const int8_t nqe = gd_w8(-K.q - K.e), nqe2 = gd_w8(-K.q2 - K.e2);
	const int8_t qe = gd_w8(K.q + K.e), qe2 = gd_w8(K.q2 + K.e2), q8 = gd_w8(K.q), q28 = gd_w8(K.q2);
	const int8_t sc_mch = gd_w8(K.sc_mch), sc_mis = gd_w8(K.sc_mis), sc_N = gd_w8(K.sc_N);

	int hi_init = -1;           // highest cell index whose ring slot has been initialised
	int last_st = -1, last_en = -1;
	int H0 = 0, last_H0_t = 0, score = GD_NEG_INF, zdropped = 0;

	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st0, en0;
		gd_band(r, qlen, tlen, w, st0, en0);
		if (st0 > en0) { zdropped = 1; break; }
		const int st = st0 & ~15, en = en0 | 15;
		const int up = st0 + (((en0 - st0 + 16) >> 4) << 4); // one past the last rewritten s cell
		int hi = en;
		{
			int h2 = (up - 1) | 15;
			if (h2 > TL16 - 1) h2 = TL16 - 1;
			if (h2 > hi) hi = h2;
		}
		for (int t = hi_init + 1 + lane; t <= hi; t += 64) {
			const int c = t & mask;
			u[c] = v[c] = x[c] = y[c] = nqe;
			x2[c] = y2[c] = nqe2;
			s[c] = 0;
		}
		if (hi > hi_init) hi_init = hi;
		__syncthreads();
		// boundary conditions (:149-163); every lane evaluates the same scalars
		int8_t x1, x21, v1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) {
				const int c = (st - 1) & mask;
				x1 = x[c], x21 = x2[c], v1 = v[c];
			} else x1 = nqe, x21 = nqe2, v1 = nqe;
		} else {
			x1 = nqe, x21 = nqe2;
			v1 = r == 0 ? nqe : r < K.long_thres ? gd_w8(-K.e) : r == K.long_thres ? gd_w8(K.long_diff) : gd_w8(-K.e2);
		}
		if (en >= r && lane == 0) {
			const int c = r & mask;
			y[c] = nqe, y2[c] = nqe2;
			u[c] = r == 0 ? nqe : r < K.long_thres ? gd_w8(-K.e) : r == K.long_thres ? gd_w8(K.long_diff) : gd_w8(-K.e2);
		}
		// score row (:165-180). cells >= TL16 would land in sf[] of the reference and are never read back.
		for (int t = st0 + lane; t < up; t += 64) {
			if (t < TL16) {
				const int j = r - t; // query index of cell (r,t); outside [0,qlen) the reference reads zero padding
				const uint8_t tb = t < tlen ? target[t] : 0;
				const uint8_t qb = (j >= 0 && j < qlen) ? query[j] : 0;
				// score table of the parity target's kernel (ksw_extd2_avx512, SR/ksw2_extd2_avx.c:183-209,310-313):
				// query N is 8, index = low nibble of target ^ query: 0 match, 1-3 mismatch, 4-12 sc_N, 13-15 zero.
				// Identical to the SSE rule (:165-180) for bytes 0..4; differs for the byte 7 that LR/map.c:1634 produces
				// for an N of a reverse-complemented read.
				const int x = (tb ^ (qb == 4 ? 8 : qb)) & 15;
				s[t & mask] = x == 0 ? sc_mch : x <= 3 ? sc_mis : x <= 12 ? sc_N : (int8_t)0;
			}
		}
		__syncthreads();
		// core loop: 64-cell chunks from the top of the window down, so that cell t reads row r-1 of t-1 before any
		// lane of a later chunk overwrites it (within a chunk all LDS reads are issued before the writes).
		uint8_t *pr = p + (size_t)r * row_bytes;
		for (int base = en - 63; base + 63 >= st; base -= 64) {
			const int t = base + lane;
			const bool on = t >= st;
			int8_t z = 0, xt1 = 0, vt1 = 0, x2t1 = 0, ut = 0, yt = 0, y2t = 0;
			if (on) {
				const int c = t & mask, c1 = (t - 1) & mask;
				z = s[c];
				if (t == st) xt1 = x1, vt1 = v1, x2t1 = x21;
				else xt1 = x[c1], vt1 = v[c1], x2t1 = x2[c1];
				ut = u[c], yt = y[c], y2t = y2[c];
			}
			__syncthreads();
			if (on) {
				const int c = t & mask;
				int8_t a = gd_w8(xt1 + vt1), b = gd_w8(yt + ut), a2 = gd_w8(x2t1 + vt1), b2 = gd_w8(y2t + ut);
				int d = a > z ? 1 : 0;  z = a > z ? a : z;
				d = b > z ? 2 : d;      z = b > z ? b : z;
				d = a2 > z ? 3 : d;     z = a2 > z ? a2 : z;
				d = b2 > z ? 4 : d;     z = b2 > z ? b2 : z;
				z = z < sc_mch ? z : sc_mch;
				u[c] = gd_w8(z - vt1), v[c] = gd_w8(z - ut);
				int8_t tmp = gd_w8(z - q8);
				a = gd_w8(a - tmp), b = gd_w8(b - tmp);
				tmp = gd_w8(z - q28);
				a2 = gd_w8(a2 - tmp), b2 = gd_w8(b2 - tmp);
				x[c]  = gd_w8((a  > 0 ? a  : 0) - qe),  d |= a  > 0 ? 0x08 : 0;
				y[c]  = gd_w8((b  > 0 ? b  : 0) - qe),  d |= b  > 0 ? 0x10 : 0;
				x2[c] = gd_w8((a2 > 0 ? a2 : 0) - qe2), d |= a2 > 0 ? 0x20 : 0;
				y2[c] = gd_w8((b2 > 0 ? b2 : 0) - qe2), d |= b2 > 0 ? 0x40 : 0;
				pr[t - st] = (uint8_t)d;
			}
			__syncthreads();
		}
		// approximate max (:367-383); uniform, every lane keeps its own copy
		if (r > 0) {
			if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
				const int d0 = v[last_H0_t & mask], d1 = u[(last_H0_t + 1) & mask];
				if (d0 > d1) H0 += d0;
				else H0 += d1, ++last_H0_t;
			} else if (last_H0_t >= st0 && last_H0_t <= en0) {
				H0 += v[last_H0_t & mask];
			} else {
				++last_H0_t, H0 += u[last_H0_t & mask];
			}
		} else H0 = v[0] - (K.q + K.e), last_H0_t = 0;
		if (r == qlen + tlen - 2 && en0 == tlen - 1) score = H0;
		last_st = st, last_en = en;
		__syncthreads();
	}
	if (lane == 0) {
		score_out[tid] = score;
		status[tid] = zdropped ? GD_ST_ZDROPPED : GD_ST_DONE;
	}
}
