// K3, exact-max mode (flag without KSW_EZ_APPROX_MAX): ksw_extz2_sse with the per-cell H array (SR/ksw2_extz2_sse.c:31-312,
// exact branch :226-268), one 64-lane wavefront per alignment, DP state and H in an LDS sliding window.  The APPROX_MAX mode --
// the only one GDiet's live path passes -- runs on the register-resident kernels (ksw_wave.hip.h); this kernel exists for
// BASELINE config 2 ("flags KSW_EZ_APPROX_MAX and 0") and returns everything ksw_extz_t carries: max / max_q / max_t (z-drop
// bookkeeping, SR/ksw2.h:172-188), mqe / mqe_t, mte / mte_q, score, zdropped, reach_end.
//
// Literal semantics: the unsigned-biased recurrence (z = s + 2(q+e), signed compares for the direction, unsigned max / min for
// the value, :38-56,185-203), 16-aligned computed window with stale s[], boundary x1 = v1 = 0 / q rule (:126-131), H[en0] from
// H[en0-1] + u, the other cells += v (:229-243), the maximum of the row with the reference's tie order (H[en0] first, then the
// four interleaved lanes of its 4-way unrolled scan in lane order, each keeping its FIRST maximum, then the scalar tail), mte_q
// computed from the 16-aligned `en` as the reference does (:259), z-drop on the row maximum, score = H[tlen-1] on the last row.
#pragma once
#include <hip/hip_runtime.h>
#include "ksw_common.h"

struct GdExtzOut { // ksw_extz_t without the CIGAR (SR/ksw2.h:31-40)
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
};

struct KswzConst {
	int32_t q, e, sc_mch, sc_mis, sc_N, zdrop, end_bonus, flag;
};
#define GD_EZ_EXTZ_ONLY 0x40

// start[2*tid], start[2*tid+1]: the cell (i0, j0) the backtrack starts from (SR/ksw2_extz2_sse.c:296-305); status DONE = walk,
// ZDROPPED = no CIGAR
__global__ __launch_bounds__(64) void ksw_extz2_exact_kernel(const KswTask *__restrict__ tasks, int n, const uint8_t *__restrict__ qseq,
                                                             const uint8_t *__restrict__ tseq, uint8_t *__restrict__ bt,
                                                             int32_t *__restrict__ status, int32_t *__restrict__ score_out,
                                                             GdExtzOut *__restrict__ ez_out, int32_t *__restrict__ start, KswzConst K, int cap)
{
	extern __shared__ uint8_t gdz_lds[];
	const int lane = threadIdx.x;
	const int tid = blockIdx.x;
	if (tid >= n) return;
	const KswTask T = tasks[tid];
	const int mask = cap - 1;
	uint8_t *u = gdz_lds, *v = u + cap, *x = v + cap, *y = x + cap, *s = y + cap;
	int32_t *H = (int32_t *)(s + cap);
	const uint8_t *query = qseq + T.qoff, *target = tseq + T.toff;
	const int qlen = T.qlen, tlen = T.tlen;
	const int w = T.w < 0 ? (tlen > qlen ? tlen : qlen) : T.w;
	const int TL16 = (tlen + 15) / 16 * 16;
	uint8_t *p = bt + T.bt_off;
	const size_t row_bytes = (size_t)T.row_bytes;
	const int qe = K.q + K.e;
	const uint8_t qe2 = (uint8_t)(qe * 2), max_sc_ = (uint8_t)(K.sc_mch + qe * 2), q8 = (uint8_t)K.q;
	const int8_t sc_mch = (int8_t)K.sc_mch, sc_mis = (int8_t)K.sc_mis, sc_N = (int8_t)K.sc_N;

	int hi_init = -1, last_st = -1, last_en = -1;
	// ez (every lane keeps the same copy)
	int32_t ez_max = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = GD_NEG_INF, ez_mqe_t = -1, ez_mte = GD_NEG_INF, ez_mte_q = -1, ez_score = GD_NEG_INF;
	int zdropped = 0;

	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st0, en0;
		gd_band(r, qlen, tlen, w, st0, en0);
		if (st0 > en0) { zdropped = 1; break; }
		const int st = st0 & ~15, en = en0 | 15;
		const int up = st0 + (((en0 - st0 + 16) >> 4) << 4);
		int hi = en;
		{
			int h2 = (up - 1) | 15;
			if (h2 > TL16 - 1) h2 = TL16 - 1;
			if (h2 > hi) hi = h2;
		}
		for (int t = hi_init + 1 + lane; t <= hi; t += 64) { // the reference's kcalloc'd state (:96-103) and H = KSW_NEG_INF (:106)
			const int c = t & mask;
			u[c] = v[c] = x[c] = y[c] = s[c] = 0;
			H[c] = GD_NEG_INF;
		}
		if (hi > hi_init) hi_init = hi;
		__syncthreads();
		uint8_t x1, v1; // :126-131
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[(st - 1) & mask], v1 = v[(st - 1) & mask];
			else x1 = v1 = 0;
		} else x1 = 0, v1 = r ? q8 : 0;
		__syncthreads();
		if (en >= r && lane == 0) y[r & mask] = 0, u[r & mask] = r ? q8 : 0;
		for (int t = st0 + lane; t < up; t += 64) // score row (:133-148); cells >= TL16 land in sf[] there and are never read back
			if (t < TL16) {
				const int j = r - t;
				const uint8_t tb = t < tlen ? target[t] : 0, qb = (j >= 0 && j < qlen) ? query[j] : 0;
				s[t & mask] = (uint8_t)((tb == 4 || qb == 4) ? sc_N : tb == qb ? sc_mch : sc_mis);
			}
		__syncthreads();
		uint8_t *pr = p + (size_t)r * row_bytes;
		for (int base = en - 63; base + 63 >= st; base -= 64) { // top-down: cell t reads row r-1 of t-1 before a later chunk overwrites it
			const int t = base + lane;
			const bool on = t >= st;
			uint8_t z = 0, xt1 = 0, vt1 = 0, ut = 0, yt = 0;
			if (on) {
				const int c = t & mask, c1 = (t - 1) & mask;
				z = (uint8_t)(s[c] + qe2);
				if (t == st) xt1 = x1, vt1 = v1;
				else xt1 = x[c1], vt1 = v[c1];
				ut = u[c], yt = y[c];
			}
			__syncthreads();
			if (on) {
				const int c = t & mask;
				uint8_t a = (uint8_t)(xt1 + vt1), b = (uint8_t)(yt + ut);
				int d = (int8_t)a > (int8_t)z ? 1 : 0;
				z = (uint8_t)((int8_t)z > (int8_t)a ? z : a);
				d = (int8_t)b > (int8_t)z ? 2 : d;
				z = z > b ? z : b;
				z = z < max_sc_ ? z : max_sc_;
				u[c] = (uint8_t)(z - vt1), v[c] = (uint8_t)(z - ut);
				const uint8_t zz = (uint8_t)(z - q8);
				a = (uint8_t)(a - zz), b = (uint8_t)(b - zz);
				x[c] = (int8_t)a > 0 ? a : 0, d |= (int8_t)a > 0 ? 0x08 : 0;
				y[c] = (int8_t)b > 0 ? b : 0, d |= (int8_t)b > 0 ? 0x10 : 0;
				pr[t - st] = (uint8_t)d;
			}
			__syncthreads();
		}
		// ---- exact maximum of the row (:226-268) ----
		int32_t max_H, max_t;
		if (r > 0) {
			const int32_t h_en0 = en0 > 0 ? H[(en0 - 1) & mask] + (int32_t)u[en0 & mask] - qe : H[en0 & mask] + (int32_t)v[en0 & mask] - qe;
			__syncthreads(); // every lane has read the old H[en0 - 1]
			const int en1 = st0 + (en0 - st0) / 4 * 4;
			// candidate order of the reference: H[en0]; lanes 0..3 of the unrolled scan, each with its first maximum; the tail
			int32_t bh = h_en0, bt_ = en0;
			uint32_t brank = 0;
			if (lane != 0) bh = INT32_MIN, brank = 0xffffffffu;
			for (int t = st0 + lane; t < en0; t += 64) {
				const int32_t h = H[t & mask] + (int32_t)v[t & mask] - qe;
				H[t & mask] = h;
				const uint32_t rank = t < en1 ? ((uint32_t)(1 + ((t - st0) & 3)) << 24) + (uint32_t)((t - st0) >> 2) : (5u << 24) + (uint32_t)(t - en1);
				if (h > bh || (h == bh && rank < brank)) bh = h, brank = rank, bt_ = t;
			}
			if (lane == 0) H[en0 & mask] = h_en0;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const int32_t oh = __shfl_xor(bh, o, 64), ot = __shfl_xor(bt_, o, 64);
				const uint32_t orank = (uint32_t)__shfl_xor((int)brank, o, 64);
				if (oh > bh || (oh == bh && orank < brank)) bh = oh, brank = orank, bt_ = ot;
			}
			max_H = bh, max_t = bt_;
			__syncthreads();
		} else {
			const int32_t h0 = (int32_t)v[0] - qe - qe;
			__syncthreads();
			if (lane == 0) H[0] = h0;
			max_H = h0, max_t = 0;
			__syncthreads();
		}
		const int32_t H_en0 = H[en0 & mask], H_st0 = H[st0 & mask];
		if (en0 == tlen - 1 && H_en0 > ez_mte) ez_mte = H_en0, ez_mte_q = r - en; // (`en`, not en0: as the reference)
		if (r - st0 == qlen - 1 && H_st0 > ez_mqe) ez_mqe = H_st0, ez_mqe_t = st0;
		{ // ksw_apply_zdrop, rotated form (SR/ksw2.h:172-188)
			const int t = max_t;
			if (max_H > ez_max) ez_max = max_H, ez_max_t = t, ez_max_q = r - t;
			else if (t >= ez_max_t && r - t >= ez_max_q) {
				const int tl = t - ez_max_t, ql = (r - t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
				if (K.zdrop >= 0 && ez_max - max_H > K.zdrop + l * K.e) { zdropped = 1; break; }
			}
		}
		if (r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H[(tlen - 1) & mask];
		last_st = st, last_en = en;
		__syncthreads();
	}
	if (lane == 0) {
		// where the walk starts (:296-305)
		int i0 = -1, j0 = -1, reach_end = 0, st_code = GD_ST_ZDROPPED;
		if (!zdropped && !(K.flag & GD_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1, st_code = GD_ST_DONE;
		else if (!zdropped && (K.flag & GD_EZ_EXTZ_ONLY) && ez_mqe + K.end_bonus > ez_max) reach_end = 1, i0 = ez_mqe_t, j0 = qlen - 1, st_code = GD_ST_DONE;
		else if (ez_max_t >= 0 && ez_max_q >= 0) i0 = ez_max_t, j0 = ez_max_q, st_code = GD_ST_DONE;
		GdExtzOut o;
		o.max = ez_max, o.zdropped = zdropped, o.max_q = ez_max_q, o.max_t = ez_max_t, o.mqe = ez_mqe, o.mqe_t = ez_mqe_t, o.mte = ez_mte, o.mte_q = ez_mte_q;
		o.score = ez_score, o.reach_end = reach_end;
		ez_out[tid] = o;
		score_out[tid] = ez_score;
		start[2 * tid] = i0, start[2 * tid + 1] = j0;
		status[tid] = st_code;
	}
}
