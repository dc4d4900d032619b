// SURVEY 8f rank 4 (kernel half): ksw_exts2_sse (SR/ksw2_exts2_sse.c:34-416, SR/ksw2.h:71) -- the splice-aware extension alignment
// of minimap2, which GDiet keeps in its tree but never calls -- one 64-lane wavefront per alignment, DP state (and H, in the
// exact-maximum mode) in an LDS sliding window, the same literal style as ksw_extz2_exact_kernel.
//
// Literal semantics (SSE4.1 form, the one ksw2_dispatch.c:94-106 selects): signed 8-bit wrapping recurrence with the three
// candidates a (E), b (F), a2a (the long-gap state entered through an acceptor signal) in strict-'>' priority order (:258-264;
// KSW_EZ_RIGHT: '>=' order :303-309); x2 = max(a2, donor) - q2 with continuation bit 0x20 (:279-288); no band (st = max(0, r-qlen+1),
// en = min(tlen-1, r), :176-181), 16-aligned computed window with the stale s[] of the rows before; boundary scalars x1 / x21 / v1
// and the u[r] / y[r] reset with the long_thres ladder (:183-196); donor / acceptor penalties from the target's GT..AG / CT..AC
// signals, their GTr / yAG flanks and the caller's junction annotation (:119-171), evaluated per cell here instead of being stored;
// exact maximum (:339-383) or approximate maximum (:384-402) with z-drop; backtrack from the last cell, or from (max_t, max_q)
// after a z-drop / with KSW_EZ_EXTZ_ONLY (:407-413), state 3 printed as N_SKIP because min_intron_len = long_thres > 0 (SR/ksw2.h:153).
#pragma once
#include <hip/hip_runtime.h>
#include "ksw_common.h"
#include "ksw_extz2_exact.hip.h" // GdExtzOut

#define GD_EZ_SCORE_ONLY 0x01
#define GD_EZ_RIGHT 0x02
#define GD_EZ_GENERIC_SC 0x04
#define GD_EZ_APPROX_MAX 0x08
#define GD_EZ_APPROX_DROP 0x10
#define GD_EZ_REV_CIGAR 0x80
#define GD_EZ_SPLICE_FOR 0x100
#define GD_EZ_SPLICE_REV 0x200
#define GD_EZ_SPLICE_FLANK 0x400

struct KswsConst {
	int32_t q, e, q2, noncan, zdrop, junc_bonus, flag;
	int32_t sc_mch, sc_mis, sc_N, long_thres, long_diff;
	int8_t mat[25];
};

static __device__ __forceinline__ int8_t gdx_add8(int a, int b) { return (int8_t)(uint8_t)((uint8_t)a + (uint8_t)b); }
static __device__ __forceinline__ int8_t gdx_sub8(int a, int b) { return (int8_t)(uint8_t)((uint8_t)a - (uint8_t)b); }

// donor[t] / acceptor[t] of the reference (:119-171); t >= tlen: the memset value
static __device__ __forceinline__ int8_t gdx_donor(int t, const uint8_t *target, int tlen, const uint8_t *junc, const KswsConst &K)
{
	const int flag = K.flag;
	if (!(flag & (GD_EZ_SPLICE_FOR | GD_EZ_SPLICE_REV))) return 0;
	int8_t d = (int8_t)-K.noncan;
	const int semi = flag & GD_EZ_SPLICE_FLANK ? -K.noncan / 2 : 0;
	const bool fw = !(flag & GD_EZ_REV_CIGAR);
	if (t < tlen - 4) {
		int can = 0;
		const int t1 = target[t + 1], t2 = target[t + 2], t3 = target[t + 3];
		if (fw) {
			if ((flag & GD_EZ_SPLICE_FOR) && t1 == 2 && t2 == 3) can = 1;
			if ((flag & GD_EZ_SPLICE_REV) && t1 == 1 && t2 == 3) can = 1;
			if (can && (t3 == 0 || t3 == 2)) can = 2;
		} else {
			if ((flag & GD_EZ_SPLICE_FOR) && t1 == 2 && t2 == 0) can = 1;
			if ((flag & GD_EZ_SPLICE_REV) && t1 == 1 && t2 == 0) can = 1;
			if (can && (t3 == 1 || t3 == 3)) can = 2;
		}
		if (can) d = can == 2 ? 0 : (int8_t)semi;
	}
	if (junc && t < tlen - 1) {
		const int jb = junc[t + 1];
		if (fw ? (((flag & GD_EZ_SPLICE_FOR) && (jb & 1)) || ((flag & GD_EZ_SPLICE_REV) && (jb & 8)))
		       : (((flag & GD_EZ_SPLICE_FOR) && (jb & 2)) || ((flag & GD_EZ_SPLICE_REV) && (jb & 4))))
			d = gdx_add8(d, K.junc_bonus);
	}
	return d;
}
static __device__ __forceinline__ int8_t gdx_acceptor(int t, const uint8_t *target, int tlen, const uint8_t *junc, const KswsConst &K)
{
	const int flag = K.flag;
	if (!(flag & (GD_EZ_SPLICE_FOR | GD_EZ_SPLICE_REV))) return 0;
	int8_t d = (int8_t)-K.noncan;
	const int semi = flag & GD_EZ_SPLICE_FLANK ? -K.noncan / 2 : 0;
	const bool fw = !(flag & GD_EZ_REV_CIGAR);
	if (t >= 2 && t < tlen) {
		int can = 0;
		const int t0 = target[t], tm1 = target[t - 1], tm2 = target[t - 2];
		if (fw) {
			if ((flag & GD_EZ_SPLICE_FOR) && tm1 == 0 && t0 == 2) can = 1;
			if ((flag & GD_EZ_SPLICE_REV) && tm1 == 0 && t0 == 1) can = 1;
			if (can && (tm2 == 1 || tm2 == 3)) can = 2;
		} else {
			if ((flag & GD_EZ_SPLICE_FOR) && tm1 == 3 && t0 == 2) can = 1;
			if ((flag & GD_EZ_SPLICE_REV) && tm1 == 3 && t0 == 1) can = 1;
			if (can && (tm2 == 0 || tm2 == 2)) can = 2;
		}
		if (can) d = can == 2 ? 0 : (int8_t)semi;
	}
	if (junc && t < tlen) {
		const int jb = junc[t];
		if (fw ? (((flag & GD_EZ_SPLICE_FOR) && (jb & 2)) || ((flag & GD_EZ_SPLICE_REV) && (jb & 4)))
		       : (((flag & GD_EZ_SPLICE_FOR) && (jb & 1)) || ((flag & GD_EZ_SPLICE_REV) && (jb & 8))))
			d = gdx_add8(d, K.junc_bonus);
	}
	return d;
}

// junc: one annotation byte per target base (concatenated like tseq) or nullptr.  The CIGAR is written by lane 0 at the end (BAM
// ops: 0 M, 1 I, 2 D, 3 N); n_cigar = 0 with KSW_EZ_SCORE_ONLY or when nothing is to be walked.
__global__ __launch_bounds__(64) void ksw_exts2_kernel(const KswTask *__restrict__ tasks, int n, const uint8_t *__restrict__ qseq,
                                                       const uint8_t *__restrict__ tseq, const uint8_t *__restrict__ juncs, uint8_t *__restrict__ bt,
                                                       GdExtzOut *__restrict__ ez_out, int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar,
                                                       KswsConst K, int cap)
{
	extern __shared__ uint8_t gdx_lds[];
	const int lane = threadIdx.x;
	const int tid = blockIdx.x;
	if (tid >= n) return;
	const KswTask T = tasks[tid];
	const int mask = cap - 1;
	int8_t *u = (int8_t *)gdx_lds, *v = u + cap, *x = v + cap, *y = x + cap, *x2 = y + cap, *s = x2 + cap;
	int32_t *H = (int32_t *)(s + cap + ((8 - (6 * cap) % 8) % 8)); // (cap is a power of two >= 256: already aligned)
	const uint8_t *query = qseq + T.qoff, *target = tseq + T.toff, *junc = juncs ? juncs + T.toff : nullptr;
	const int qlen = T.qlen, tlen = T.tlen;
	const int TL16 = (tlen + 15) / 16 * 16;
	uint8_t *p = bt + T.bt_off;
	const size_t row_bytes = (size_t)T.row_bytes;
	const int flag = K.flag, qe = K.q + K.e;
	const bool with_cigar = !(flag & GD_EZ_SCORE_ONLY), approx_max = (flag & GD_EZ_APPROX_MAX) != 0, right = with_cigar && (flag & GD_EZ_RIGHT);
	const int8_t nqe = (int8_t)(-K.q - K.e), nq2 = (int8_t)-K.q2;

	int hi_init = -1, last_st = -1, last_en = -1;
	int32_t ez_max = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = GD_NEG_INF, ez_mqe_t = -1, ez_mte = GD_NEG_INF, ez_mte_q = -1, ez_score = GD_NEG_INF;
	int32_t H0 = 0, last_H0_t = 0;
	int zdropped = 0;

	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st0 = 0, en0 = tlen - 1;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		const int st = st0 & ~15, en = en0 | 15;
		const int up = st0 + (((en0 - st0 + 16) >> 4) << 4);
		int hi = en;
		{
			int h2 = (up - 1) | 15;
			if (h2 > TL16 - 1) h2 = TL16 - 1;
			if (h2 > hi) hi = h2;
		}
		for (int t = hi_init + 1 + lane; t <= hi; t += 64) { // the reference's initial fill (:102-107)
			const int c = t & mask;
			u[c] = v[c] = x[c] = y[c] = nqe, x2[c] = nq2, s[c] = 0;
			if (!approx_max) H[c] = GD_NEG_INF;
		}
		if (hi > hi_init) hi_init = hi;
		__syncthreads();
		const int8_t ladder = r == 0 ? nqe : r < K.long_thres ? (int8_t)-K.e : r == K.long_thres ? (int8_t)K.long_diff : (int8_t)0;
		int8_t x1, x21, v1; // :183-191
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[(st - 1) & mask], x21 = x2[(st - 1) & mask], v1 = v[(st - 1) & mask];
			else x1 = nqe, x21 = nq2, v1 = nqe;
		} else x1 = nqe, x21 = nq2, v1 = ladder;
		__syncthreads();
		if (en >= r && lane == 0) y[r & mask] = nqe, u[r & mask] = ladder;
		for (int t = st0 + lane; t < up; t += 64) // score row (:197-217); cells >= TL16 land in sf[] there and are never read back
			if (t < TL16) {
				const int j = r - t;
				const uint8_t tb = t < tlen ? target[t] : 0, qb = (j >= 0 && j < qlen) ? query[j] : 0;
				int8_t sc;
				if (flag & GD_EZ_GENERIC_SC) sc = K.mat[(tb < 5 ? tb : 4) * 5 + (qb < 5 ? qb : 4)];
				else sc = (int8_t)((tb == 4 || qb == 4) ? K.sc_N : tb == qb ? K.sc_mch : K.sc_mis);
				s[t & mask] = sc;
			}
		__syncthreads();
		uint8_t *pr = p + (size_t)r * row_bytes;
		for (int base = en - 63; base + 63 >= st; base -= 64) { // top-down: cell t reads row r-1 of t-1 before a later chunk overwrites it
			const int t = base + lane;
			const bool on = t >= st;
			int8_t z = 0, xt1 = 0, vt1 = 0, x2t1 = 0, ut = 0, yt = 0;
			if (on) {
				const int c = t & mask, c1 = (t - 1) & mask;
				z = s[c];
				if (t == st) xt1 = x1, vt1 = v1, x2t1 = x21;
				else xt1 = x[c1], vt1 = v[c1], x2t1 = x2[c1];
				ut = u[c], yt = y[c];
			}
			__syncthreads();
			if (on) {
				const int c = t & mask;
				const int8_t dn = gdx_donor(t, target, tlen, junc, K), ac = gdx_acceptor(t, target, tlen, junc, K);
				int8_t a = gdx_add8(xt1, vt1), b = gdx_add8(yt, ut), a2 = gdx_add8(x2t1, vt1);
				const int8_t a2a = gdx_add8(a2, ac);
				int d;
				if (!right) {
					d = a > z ? 1 : 0;
					z = z > a ? z : a;
					d = b > z ? 2 : d;
					z = z > b ? z : b;
					d = a2a > z ? 3 : d;
					z = z > a2a ? z : a2a;
				} else {
					d = z > a ? 0 : 1;
					z = z > a ? z : a;
					d = z > b ? d : 2;
					z = z > b ? z : b;
					d = z > a2a ? d : 3;
					z = z > a2a ? z : a2a;
				}
				u[c] = gdx_sub8(z, vt1), v[c] = gdx_sub8(z, ut);
				const int8_t tmp = gdx_sub8(z, K.q);
				a = gdx_sub8(a, tmp), b = gdx_sub8(b, tmp), a2 = gdx_sub8(a2, gdx_sub8(z, K.q2));
				if (!right) {
					x[c] = gdx_sub8(a > 0 ? a : 0, qe), d |= a > 0 ? 0x08 : 0;
					y[c] = gdx_sub8(b > 0 ? b : 0, qe), d |= b > 0 ? 0x10 : 0;
					x2[c] = gdx_sub8(a2 > dn ? a2 : dn, K.q2), d |= a2 > dn ? 0x20 : 0;
				} else {
					x[c] = gdx_sub8(0 > a ? 0 : a, qe), d |= 0 > a ? 0 : 0x08;
					y[c] = gdx_sub8(0 > b ? 0 : b, qe), d |= 0 > b ? 0 : 0x10;
					x2[c] = gdx_sub8(dn > a2 ? dn : a2, K.q2), d |= dn > a2 ? 0 : 0x20;
				}
				if (with_cigar) pr[t - st] = (uint8_t)d;
			}
			__syncthreads();
		}
		if (!approx_max) { // ---- exact maximum of the row (:339-383) ----
			int32_t max_H, max_t;
			if (r > 0) {
				const int32_t h_en0 = en0 > 0 ? H[(en0 - 1) & mask] + (int32_t)u[en0 & mask] : H[en0 & mask] + (int32_t)v[en0 & mask];
				__syncthreads(); // every lane has read the old H[en0 - 1]
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				// candidate order of the reference: H[en0]; lanes 0..3 of the unrolled scan, each with its first maximum; the tail
				int32_t bh = h_en0, bt_ = en0;
				uint32_t brank = 0;
				if (lane != 0) bh = INT32_MIN, brank = 0xffffffffu;
				for (int t = st0 + lane; t < en0; t += 64) {
					const int32_t h = H[t & mask] + (int32_t)v[t & mask];
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
				const int32_t h0 = (int32_t)v[0] - qe;
				__syncthreads();
				if (lane == 0) H[0] = h0;
				max_H = h0, max_t = 0;
				__syncthreads();
			}
			const int32_t H_en0 = H[en0 & mask], H_st0 = H[st0 & mask];
			if (en0 == tlen - 1 && H_en0 > ez_mte) ez_mte = H_en0, ez_mte_q = r - en; // (`en`, not en0: as the reference)
			if (r - st0 == qlen - 1 && H_st0 > ez_mqe) ez_mqe = H_st0, ez_mqe_t = st0;
			{ // ksw_apply_zdrop, rotated form with e = 0 (SR/ksw2.h:172-188; :380)
				const int t = max_t;
				if (max_H > ez_max) ez_max = max_H, ez_max_t = t, ez_max_q = r - t;
				else if (t >= ez_max_t && r - t >= ez_max_q) {
					if (K.zdrop >= 0 && ez_max - max_H > K.zdrop) { zdropped = 1; break; }
				}
			}
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H[(tlen - 1) & mask];
		} else { // ---- approximate maximum (:384-402) ----
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					const int32_t d0 = v[last_H0_t & mask], d1 = u[(last_H0_t + 1) & mask];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t & mask];
				} else {
					++last_H0_t, H0 += u[last_H0_t & mask];
				}
			} else H0 = (int32_t)v[0] - qe, last_H0_t = 0;
			if (flag & GD_EZ_APPROX_DROP) {
				const int t = last_H0_t;
				if (H0 > ez_max) ez_max = H0, ez_max_t = t, ez_max_q = r - t;
				else if (t >= ez_max_t && r - t >= ez_max_q) {
					if (K.zdrop >= 0 && ez_max - H0 > K.zdrop) { zdropped = 1; break; }
				}
			}
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H0;
		}
		last_st = st, last_en = en;
		__syncthreads();
	}
	if (lane != 0) return;
	GdExtzOut o;
	o.max = ez_max, o.zdropped = zdropped, o.max_q = ez_max_q, o.max_t = ez_max_t, o.mqe = ez_mqe, o.mqe_t = ez_mqe_t, o.mte = ez_mte, o.mte_q = ez_mte_q;
	o.score = ez_score, o.reach_end = 0;
	ez_out[tid] = o;
	// ---- backtrack (:407-413; SR/ksw2.h:131-163 with is_rot = 1, min_intron_len = long_thres) ----
	int nc = 0;
	if (with_cigar) {
		int i = -1, j = -1;
		if (!zdropped && !(flag & GD_EZ_EXTZ_ONLY)) i = tlen - 1, j = qlen - 1;
		else if (ez_max_t >= 0 && ez_max_q >= 0) i = ez_max_t, j = ez_max_q;
		const bool walk = i >= 0 || j >= 0 || (!zdropped && !(flag & GD_EZ_EXTZ_ONLY));
		if (walk) {
			__threadfence(); // the backtrace rows were written by the other lanes of this wavefront
			uint32_t *cg = cigar + T.cig_off;
			const int ccap = T.cig_cap, mil = K.long_thres;
			uint32_t last = 0;
			int have = 0, state = 0;
			auto push = [&](uint32_t op, uint32_t len) {
				if (have && (last & 0xf) == op) last += len << 4;
				else {
					if (have) { if (nc < ccap) cg[nc] = last; ++nc; }
					last = len << 4 | op, have = 1;
				}
			};
			while (i >= 0 && j >= 0) {
				const int r = i + j;
				int st0 = 0, en0 = tlen - 1, force_state = -1;
				if (st0 < r - qlen + 1) st0 = r - qlen + 1;
				if (en0 > r) en0 = r;
				const int off = st0 & ~15, off_end = en0 | 15;
				if (i < off) force_state = 2;
				if (i > off_end) force_state = 1;
				const uint32_t tmp = force_state < 0 ? p[(size_t)r * row_bytes + (size_t)(i - off)] : 0;
				if (state == 0) state = tmp & 7;
				else if (!(tmp >> (state + 2) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force_state >= 0) state = force_state;
				if (state == 0) push(0, 1), --i, --j;
				else if (state == 1 || (state == 3 && mil <= 0)) push(2, 1), --i;
				else if (state == 3 && mil > 0) push(3, 1), --i;
				else push(1, 1), --j;
			}
			if (i >= 0) push(mil > 0 && i >= mil ? 3 : 2, (uint32_t)(i + 1));
			if (j >= 0) push(1, (uint32_t)(j + 1));
			if (have) { if (nc < ccap) cg[nc] = last; ++nc; }
			if (!(flag & GD_EZ_REV_CIGAR) && nc <= ccap)
				for (int k = 0; k < nc >> 1; ++k) {
					const uint32_t t0 = cg[k];
					cg[k] = cg[nc - 1 - k], cg[nc - 1 - k] = t0;
				}
		}
	}
	n_cigar[tid] = nc;
}
