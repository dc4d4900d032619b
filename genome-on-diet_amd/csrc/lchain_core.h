// SURVEY 8f rank 4 (chaining half): the pair score of mg_lchain_dp, written once for the device kernel (lchain.hip.h) and for the
// host test driver (tests/emul/lchain_emul.cpp).  comput_sc (SR/lchain.c:91-122) with mg_log2 (SR/mmpriv.h:146-157); the float
// terms are evaluated one operation at a time (no contraction into fused multiply-adds: the reference's build has none).
#pragma once
#include <stdint.h>
#if !defined(__HIPCC__)
#define __host__
#define __device__
#endif

struct GdChainOpt { // the scalar arguments of mg_lchain_dp
	int32_t max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc;
	float chn_pen_gap, chn_pen_skip;
	int32_t is_cdna, n_seg;
};

#define GDL_SEG_SHIFT 48
#define GDL_SEG_MASK (0xffULL << GDL_SEG_SHIFT)

static __host__ __device__ inline float gdl_mg_log2(float x) // SR/mmpriv.h:146-157
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	union { float f; uint32_t i; } z = {x};
	float log_2 = (float)(int)(((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	const float t1 = -0.34484843f * z.f;
	const float t2 = t1 + 2.02466578f;
	const float t3 = t2 * z.f;
	const float t4 = t3 - 0.67487759f;
	log_2 += t4;
	return log_2;
}

// SR/lchain.c:91-122
static __host__ __device__ inline int32_t gdl_comput_sc(uint64_t aix, uint64_t aiy, uint64_t ajx, uint64_t ajy, const GdChainOpt &O)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const int32_t dq = (int32_t)aiy - (int32_t)ajy;
	const int32_t sidi = (int32_t)((aiy & GDL_SEG_MASK) >> GDL_SEG_SHIFT), sidj = (int32_t)((ajy & GDL_SEG_MASK) >> GDL_SEG_SHIFT);
	if (dq <= 0 || dq > O.max_dist_x) return INT32_MIN;
	const int32_t dr = (int32_t)(aix - ajx);
	if (sidi == sidj && (dr == 0 || dq > O.max_dist_y)) return INT32_MIN;
	const int32_t dd = dr > dq ? dr - dq : dq - dr;
	if (sidi == sidj && dd > O.bw) return INT32_MIN;
	if (O.n_seg > 1 && !O.is_cdna && sidi == sidj && dr > O.max_dist_y) return INT32_MIN;
	const int32_t dg = dr < dq ? dr : dq;
	const int32_t q_span = (int32_t)(ajy >> 32 & 0xff);
	int32_t sc = q_span < dg ? q_span : dg;
	if (dd || dg > q_span) {
		const float l1 = O.chn_pen_gap * (float)dd, l2 = O.chn_pen_skip * (float)dg;
		const float lin_pen = l1 + l2;
		const float log_pen = dd >= 1 ? gdl_mg_log2((float)(dd + 1)) : 0.0f;
		if (O.is_cdna || sidi != sidj) {
			if (sidi != sidj && dr == 0) ++sc;
			else if (dr > dq || sidi != sidj) sc -= (int)(lin_pen < log_pen ? lin_pen : log_pen);
			else {
				const float h = .5f * log_pen;
				sc -= (int)(lin_pen + h);
			}
		} else {
			const float h = .5f * log_pen;
			sc -= (int)(lin_pen + h);
		}
	}
	return sc;
}

