/*
 * oracle/gdo_lchain.c -- CPU ORACLE (test infrastructure, NOT product code) for SURVEY 8f rank 4 (chaining half).
 *
 * Plain-C restatement of mg_lchain_dp (SR/lchain.c:124-190) with its helpers comput_sc (:91-122), mg_chain_backtrack (:9-53),
 * compact_a (:55-89), the in-place radix sort the last two call (radix_sort_128x: SR/ksort.h:101-151 instantiated at
 * SR/misc.c:155-156 -- NOT stable, so its exact permutation is restated) and mg_log2 (SR/mmpriv.h:146-157).  GDiet keeps this
 * code of minimap2 in its tree and never calls it.  Pinned by oracle/pin_rank4.py against the reference's own
 * mg_lchain_dp (oracle/_ref/libgdiet_sr_avx.so) and by the golden vectors tests/golden/lchain_dp.npz written from it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "gdo_lchain.h"

typedef struct { uint64_t x, y; } a128_t;

/* ---- SR/ksort.h:98-151 with rskey(a) = a.x, sizeof_key = 8 ---- */
#define RS_MIN_SIZE 64
#define RS_MAX_BITS 8
typedef struct { a128_t *b, *e; } rsbucket_t;

static void rs_insertsort(a128_t *beg, a128_t *end)
{
	a128_t *i;
	for (i = beg + 1; i < end; ++i)
		if (i->x < (i - 1)->x) {
			a128_t *j, tmp = *i;
			for (j = i; j > beg && tmp.x < (j - 1)->x; --j) *j = *(j - 1);
			*j = tmp;
		}
}

static void rs_sort(a128_t *beg, a128_t *end, int n_bits, int s)
{
	a128_t *i;
	int size = 1 << n_bits, m = size - 1;
	rsbucket_t *k, b[1 << RS_MAX_BITS], *be = b + size;
	for (k = b; k != be; ++k) k->b = k->e = beg;
	for (i = beg; i != end; ++i) ++b[i->x >> s & m].e;
	for (k = b + 1; k != be; ++k) k->e += (k - 1)->e - beg, k->b = (k - 1)->e;
	for (k = b; k != be;) {
		if (k->b != k->e) {
			rsbucket_t *l;
			if ((l = b + (k->b->x >> s & m)) != k) {
				a128_t tmp = *k->b, swap;
				do {
					swap = tmp, tmp = *l->b, *l->b++ = swap;
					l = b + (tmp.x >> s & m);
				} while (l != k);
				*k->b++ = tmp;
			} else ++k->b;
		} else ++k;
	}
	for (b->b = beg, k = b + 1; k != be; ++k) k->b = (k - 1)->e;
	if (s) {
		s = s > n_bits ? s - n_bits : 0;
		for (k = b; k != be; ++k)
			if (k->e - k->b > RS_MIN_SIZE) rs_sort(k->b, k->e, n_bits, s);
			else if (k->e - k->b > 1) rs_insertsort(k->b, k->e);
	}
}

void gdo_radix_sort_128x(uint64_t *a /* n pairs (x, y) */, int64_t n)
{
	a128_t *beg = (a128_t *)a, *end = beg + n;
	if (end - beg <= RS_MIN_SIZE) rs_insertsort(beg, end);
	else rs_sort(beg, end, RS_MAX_BITS, (8 - 1) * RS_MAX_BITS);
}

/* SR/mmpriv.h:146-157 */
static float mg_log2(float x)
{
	union { float f; uint32_t i; } z = {x};
	float log_2 = ((z.i >> 23) & 255) - 128;
	z.i &= ~(255 << 23);
	z.i += 127 << 23;
	log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return log_2;
}

#define SEG_SHIFT 48
#define SEG_MASK (0xffULL << SEG_SHIFT)

/* SR/lchain.c:91-122 */
static int32_t comput_sc(const a128_t *ai, const a128_t *aj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, float chn_pen_gap,
                         float chn_pen_skip, int is_cdna, int n_seg)
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	int32_t sidi = (ai->y & SEG_MASK) >> SEG_SHIFT;
	int32_t sidj = (aj->y & SEG_MASK) >> SEG_SHIFT;
	if (dq <= 0 || dq > max_dist_x) return INT32_MIN;
	dr = (int32_t)(ai->x - aj->x);
	if (sidi == sidj && (dr == 0 || dq > max_dist_y)) return INT32_MIN;
	dd = dr > dq ? dr - dq : dq - dr;
	if (sidi == sidj && dd > bw) return INT32_MIN;
	if (n_seg > 1 && !is_cdna && sidi == sidj && dr > max_dist_y) return INT32_MIN;
	dg = dr < dq ? dr : dq;
	q_span = aj->y >> 32 & 0xff;
	sc = q_span < dg ? q_span : dg;
	if (dd || dg > q_span) {
		float lin_pen, log_pen;
		lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
		log_pen = dd >= 1 ? mg_log2(dd + 1) : 0.0f;
		if (is_cdna || sidi != sidj) {
			if (sidi != sidj && dr == 0) ++sc;
			else if (dr > dq || sidi != sidj) sc -= (int)(lin_pen < log_pen ? lin_pen : log_pen);
			else sc -= (int)(lin_pen + .5f * log_pen);
		} else sc -= (int)(lin_pen + .5f * log_pen);
	}
	return sc;
}

/* SR/lchain.c:9-53 */
static uint64_t *chain_backtrack(int64_t n, const int32_t *f, const int64_t *p, int32_t *v, int32_t *t, int32_t min_cnt, int32_t min_sc,
                                 int32_t *n_u_, int32_t *n_v_)
{
	a128_t *z;
	uint64_t *u;
	int64_t i, k, n_z, n_v;
	int32_t n_u;
	*n_u_ = *n_v_ = 0;
	for (i = 0, n_z = 0; i < n; ++i)
		if (f[i] >= min_sc) ++n_z;
	if (n_z == 0) return 0;
	z = (a128_t *)malloc(sizeof(a128_t) * n_z);
	for (i = 0, k = 0; i < n; ++i)
		if (f[i] >= min_sc) z[k].x = f[i], z[k++].y = i;
	gdo_radix_sort_128x((uint64_t *)z, n_z);
	memset(t, 0, n * 4);
	for (k = n_z - 1, n_v = n_u = 0; k >= 0; --k) {
		int64_t n_v0 = n_v;
		int32_t sc;
		for (i = z[k].y; i >= 0 && t[i] == 0; i = p[i]) ++n_v, t[i] = 1;
		sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) ++n_u;
		else n_v = n_v0;
	}
	u = (uint64_t *)malloc(8 * (n_u > 0 ? n_u : 1));
	memset(t, 0, n * 4);
	for (k = n_z - 1, n_v = n_u = 0; k >= 0; --k) {
		int64_t n_v0 = n_v;
		int32_t sc;
		for (i = z[k].y; i >= 0 && t[i] == 0; i = p[i]) v[n_v++] = i, t[i] = 1;
		sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u[n_u++] = (uint64_t)sc << 32 | (n_v - n_v0);
		else n_v = n_v0;
	}
	free(z);
	*n_u_ = n_u, *n_v_ = n_v;
	return u;
}

/* SR/lchain.c:55-89; a is read only here (the reference reuses and frees it) */
static a128_t *compact_a(int32_t n_u, uint64_t *u, int32_t n_v, int32_t *v, const a128_t *a_in, int64_t n)
{
	a128_t *a, *b, *w;
	uint64_t *u2;
	int64_t i, j, k;
	a = (a128_t *)malloc(sizeof(a128_t) * (n > 0 ? n : 1));
	memcpy(a, a_in, sizeof(a128_t) * n);
	b = (a128_t *)malloc(sizeof(a128_t) * (n_v > 0 ? n_v : 1));
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	w = (a128_t *)malloc(sizeof(a128_t) * n_u);
	for (i = k = 0; i < n_u; ++i) {
		w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | i;
		k += (int32_t)u[i];
	}
	gdo_radix_sort_128x((uint64_t *)w, n_u);
	u2 = (uint64_t *)malloc(8 * n_u);
	for (i = k = 0; i < n_u; ++i) {
		int32_t j2 = (int32_t)w[i].y, nn = (int32_t)u[j2];
		u2[i] = u[j2];
		memcpy(&a[k], &b[w[i].y >> 32], nn * sizeof(a128_t));
		k += nn;
	}
	memcpy(u, u2, n_u * 8);
	memcpy(b, a, k * sizeof(a128_t));
	free(a), free(w), free(u2);
	return b;
}

/* SR/lchain.c:124-190.  a: n pairs (x, y), not modified.  Returns the rearranged anchors (malloc'd, sum of the low words of u[]
 * pairs) or NULL; *u malloc'd (n_u entries: score << 32 | n_anchors).  With f_out / p_out (n entries each, may be NULL) the DP
 * arrays of the fill stage are returned as well. */
uint64_t *gdo_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc, float chn_pen_gap,
                        float chn_pen_skip, int is_cdna, int n_seg, int64_t n, const uint64_t *a_, int *n_u_, uint64_t **_u, int32_t *f_out,
                        int64_t *p_out)
{
	const a128_t *a = (const a128_t *)a_;
	int32_t *f, *t, *v, n_u, n_v, mmax_f = 0;
	int64_t *p, i, j, max_ii, st = 0;
	uint64_t *u;
	a128_t *b;
	*_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) return 0;
	if (max_dist_x < bw) max_dist_x = bw;
	if (max_dist_y < bw && !is_cdna) max_dist_y = bw;
	p = (int64_t *)malloc(8 * n), f = (int32_t *)malloc(4 * n), v = (int32_t *)malloc(4 * n), t = (int32_t *)calloc(n, 4);
	for (i = 0, max_ii = -1; i < n; ++i) {
		int64_t max_j = -1, end_j;
		int32_t max_f = a[i].y >> 32 & 0xff, n_skip = 0;
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + max_dist_x)) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc = comput_sc(&a[i], &a[j], max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, is_cdna, n_seg);
			if (sc == INT32_MIN) continue;
			sc += f[j];
			if (sc > max_f) {
				max_f = sc, max_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = i;
		}
		end_j = j;
		if (max_ii < 0 || a[i].x - a[max_ii].x > (int64_t)max_dist_x) {
			int32_t max = INT32_MIN;
			max_ii = -1;
			for (j = i - 1; j >= st; --j)
				if (max < f[j]) max = f[j], max_ii = j;
		}
		if (max_ii >= 0 && max_ii < end_j) {
			int32_t tmp = comput_sc(&a[i], &a[max_ii], max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, is_cdna, n_seg);
			if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) max_f = tmp + f[max_ii], max_j = max_ii;
		}
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
		if (max_ii < 0 || (a[i].x - a[max_ii].x <= (int64_t)max_dist_x && f[max_ii] < f[i])) max_ii = i;
		if (mmax_f < max_f) mmax_f = max_f;
	}
	if (f_out) memcpy(f_out, f, 4 * n);
	if (p_out) memcpy(p_out, p, 8 * n);
	u = chain_backtrack(n, f, p, v, t, min_cnt, min_sc, &n_u, &n_v);
	*n_u_ = n_u, *_u = u;
	free(p), free(f), free(t);
	if (n_u == 0) {
		free(v);
		if (u) free(u), *_u = 0;
		return 0;
	}
	b = compact_a(n_u, u, n_v, v, a, n);
	free(v);
	return (uint64_t *)b;
}
