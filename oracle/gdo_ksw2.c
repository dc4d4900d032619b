/*
 * oracle/gdo_ksw2.c -- CPU ORACLE (test infrastructure, NOT product code; see gdo_ksw2.h).
 *
 * Scalar, cell-at-a-time restatement of the reference's SSE kernels.  The SSE code computes whole 16-lane
 * blocks, so cells outside the band [st0,en0] but inside the 16-aligned window [st,en] ("padding lanes") are
 * computed from whatever the state arrays hold and DO feed in-band cells on later anti-diagonals.  To reproduce
 * that literally we keep the same flat byte layout as the reference (u|v|x|y|x2|y2|s|sf|qr|slack, each tlen_*16
 * bytes; SR/ksw2_extd2_sse.c:107-110) including its out-of-array spills (the score row s[] may spill into sf[],
 * sf[] reads may run into qr[], qr[] reads may run into the zeroed slack), and we walk blocks of 16 with a
 * one-byte carry exactly as the shifted loads do (:38-56).
 *
 * All int8 arithmetic wraps (SSE paddb/psubb), signed compares (pcmpgtb), SSE4.1 semantics for max/min/blend
 * (that is the variant the dispatcher picks on any AVX-512 host; SR/ksw2_dispatch.c:55-107).
 */
#include <stdlib.h>
#include <string.h>
#include "gdo_ksw2.h"

typedef int8_t i8;
typedef uint8_t u8t;

static inline i8 add8(i8 a, i8 b) { return (i8)(u8t)((u8t)a + (u8t)b); }
static inline i8 sub8(i8 a, i8 b) { return (i8)(u8t)((u8t)a - (u8t)b); }
static inline i8 max8(i8 a, i8 b) { return a > b ? a : b; }
static inline i8 min8(i8 a, i8 b) { return a < b ? a : b; }

/* SR/ksw2.h:165-170 */
static void reset_extz(gdo_extz_t *ez)
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = GDO_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

/* SR/ksw2.h:172-188 (is_rot == 1 form) */
static int apply_zdrop(gdo_extz_t *ez, int32_t H, int r, int t, int zdrop, i8 e)
{
	if (H > (int32_t)ez->max) {
		ez->max = H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
		l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) {
			ez->zdropped = 1;
			return 1;
		}
	}
	return 0;
}

/* SR/ksw2.h:115-125 */
static uint32_t *push_cigar(int *n_cigar, int *m_cigar, uint32_t *cigar, uint32_t op, int len)
{
	if (*n_cigar == 0 || op != (cigar[(*n_cigar) - 1] & 0xf)) {
		if (*n_cigar == *m_cigar) {
			*m_cigar = *m_cigar ? (*m_cigar) << 1 : 4;
			cigar = (uint32_t *)realloc(cigar, (size_t)(*m_cigar) << 2);
		}
		cigar[(*n_cigar)++] = (uint32_t)len << 4 | op;
	} else cigar[(*n_cigar) - 1] += (uint32_t)len << 4;
	return cigar;
}

/* SR/ksw2.h:131-163, rotated (anti-diagonal) layout only */
void gdo_backtrack(int is_rev, int min_intron_len, const uint8_t *p, const int *off, const int *off_end, int n_col,
                   int i0, int j0, int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{
	int n_cigar = 0, m_cigar = *m_cigar_, i = i0, j = j0, r, state = 0;
	uint32_t *cigar = *cigar_, tmp;
	while (i >= 0 && j >= 0) {
		int force_state = -1;
		r = i + j;
		if (i < off[r]) force_state = 2;
		if (off_end && i > off_end[r]) force_state = 1;
		tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 0, 1), --i, --j;
		else if (state == 1 || (state == 3 && min_intron_len <= 0)) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, 1), --i;
		else if (state == 3 && min_intron_len > 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 3, 1), --i;
		else cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, 1), --j;
	}
	if (i >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, min_intron_len > 0 && i >= min_intron_len ? 3 : 2, i + 1);
	if (j >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, j + 1);
	if (!is_rev)
		for (i = 0; i < n_cigar >> 1; ++i)
			tmp = cigar[i], cigar[i] = cigar[n_cigar - 1 - i], cigar[n_cigar - 1 - i] = tmp;
	*m_cigar_ = m_cigar, *n_cigar_ = n_cigar, *cigar_ = cigar;
}

/* SR/exact_match_sse.c:23-91: whole 16-byte chunks first, then a zero-padded tail chunk */
int gdo_exact_match(int qlen, const uint8_t *query, int tlen, const uint8_t *target)
{
	int t, i;
	(void)tlen;
	for (t = 0; t + 16 <= qlen; t += 16)
		for (i = 0; i < 16; ++i)
			if (target[t + i] != query[t + i]) return 0;
	for (i = 0; i < qlen % 16; ++i)
		if (target[qlen / 16 * 16 + i] != query[qlen / 16 * 16 + i]) return 0;
	return 1;
}

/* band of anti-diagonal r; SR/ksw2_extd2_sse.c:132-147. returns 0 if empty */
static inline int band(int r, int qlen, int tlen, int w, int *st0, int *en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	*st0 = st, *en0 = en;
	return st <= en;
}

/* score row; SR/ksw2_extd2_sse.c:165-184.  s/sf/qrr are pointers into the shared flat buffer. */
static void fill_scores(int flag, int st0, int en0, u8t *s, const u8t *sf, const u8t *qrr, int m, const i8 *mat,
                        i8 sc_mch, i8 sc_mis, i8 sc_N)
{
	int t, i;
	if (flag & GDO_EZ_AVX512_SC) {
		/* the AVX-512 port (SR/ksw2_extd2_avx.c:183-209,310-313): query N is stored as 8, the score is a 16-entry
		 * table indexed by the low nibble of target ^ query: 0 match, 1-3 mismatch, 4-12 sc_N, 13-15 zero.  Same as the
		 * SSE rule for bytes 0..4, different for any other byte (the reverse-complemented N = 7 of LR/map.c:1634). */
		for (t = st0; t <= en0; t += 16) {
			u8t sq[16], st[16];
			memcpy(sq, sf + t, 16), memcpy(st, qrr + t, 16);
			for (i = 0; i < 16; ++i) {
				const int x = (sq[i] ^ (st[i] == 4 ? 8 : st[i])) & 15;
				s[t + i] = (u8t)(x == 0 ? sc_mch : x <= 3 ? sc_mis : x <= 12 ? sc_N : 0);
			}
		}
	} else if (!(flag & GDO_EZ_GENERIC_SC)) {
		for (t = st0; t <= en0; t += 16) {
			u8t sq[16], st[16];
			memcpy(sq, sf + t, 16), memcpy(st, qrr + t, 16); /* loads complete before the store */
			for (i = 0; i < 16; ++i) {
				i8 sc = sq[i] == st[i] ? sc_mch : sc_mis;
				if (sq[i] == (u8t)(m - 1) || st[i] == (u8t)(m - 1)) sc = sc_N;
				s[t + i] = (u8t)sc;
			}
		}
	} else {
		for (t = st0; t <= en0; ++t) s[t] = (u8t)mat[sf[t] * m + qrr[t]];
	}
}

void gdo_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                   gdo_extz_t *ez)
{
	int r, t, i, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, max_sc, min_sc;
	int long_thres, long_diff;
	int with_cigar = !(flag & GDO_EZ_SCORE_ONLY), approx_max = !!(flag & GDO_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	u8t *mem, *qr, *sf, *p = 0;
	i8 *u, *v, *x, *y, *x2, *y2, *s;
	i8 sc_mch, sc_mis, sc_N, nqe, nqe2;

	reset_extz(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;
	if (q2 + e2 < q + e) t = q, q = q2, q2 = t, t = e, e = e2, e2 = t; /* :78 */
	qe = q + e;
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? (i8)-e2 : mat[m * m - 1]; /* :87 */
	nqe = (i8)(-q - e), nqe2 = (i8)(-q2 - e2);

	if (w < 0) w = tlen > qlen ? tlen : qlen;
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return; /* :100 */

	long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0; /* :102-105 */
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	/* flat layout of :107-110 with a 16-byte aligned base => exactly 16 zero bytes of slack after qr */
	mem = (u8t *)calloc((size_t)tlen_ * 8 + qlen_ + 1, 16);
	u = (i8 *)mem, v = u + tlen_ * 16, x = v + tlen_ * 16, y = x + tlen_ * 16, x2 = y + tlen_ * 16, y2 = x2 + tlen_ * 16;
	s = y2 + tlen_ * 16, sf = (u8t *)(s + tlen_ * 16), qr = sf + tlen_ * 16;
	memset(u, nqe, tlen_ * 16), memset(v, nqe, tlen_ * 16), memset(x, nqe, tlen_ * 16), memset(y, nqe, tlen_ * 16);
	memset(x2, nqe2, tlen_ * 16), memset(y2, nqe2, tlen_ * 16);
	if (!approx_max) {
		H = (int32_t *)malloc((size_t)tlen_ * 16 * 4);
		for (t = 0; t < tlen_ * 16; ++t) H[t] = GDO_NEG_INF;
	}
	if (with_cigar) {
		p = (u8t *)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int *)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st, en, st0, en0, st_, en_;
		i8 x1, x21, v1;
		u8t *qrr = qr + (qlen - 1 - r);
		if (!band(r, qlen, tlen, w, &st0, &en0)) { ez->zdropped = 1; break; } /* :142-145 */
		st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
		/* boundary conditions, :149-163 */
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = nqe, x21 = nqe2, v1 = nqe;
		} else {
			x1 = nqe, x21 = nqe2;
			v1 = r == 0 ? nqe : r < long_thres ? (i8)-e : r == long_thres ? (i8)long_diff : (i8)-e2;
		}
		if (en >= r) {
			y[r] = nqe, y2[r] = nqe2;
			u[r] = r == 0 ? nqe : r < long_thres ? (i8)-e : r == long_thres ? (i8)long_diff : (i8)-e2;
		}
		fill_scores(flag, st0, en0, (u8t *)s, sf, qrr, m, mat, sc_mch, sc_mis, sc_N);
		/* core loop, :186-322: 16-lane blocks, ascending, one-byte carries between blocks */
		st_ = st / 16, en_ = en / 16;
		if (with_cigar) off[r] = st, off_end[r] = en;
		for (t = st_; t <= en_; ++t) {
			i8 ox[16], ov[16], ox2[16]; /* the block's row r-1 values (loaded before any store) */
			u8t *pr = with_cigar ? p + ((size_t)r * n_col_ - st_ + t) * 16 : 0;
			memcpy(ox, x + t * 16, 16), memcpy(ov, v + t * 16, 16), memcpy(ox2, x2 + t * 16, 16);
			for (i = 0; i < 16; ++i) {
				int c = t * 16 + i;
				i8 z = s[c], xt1 = i ? ox[i - 1] : x1, vt1 = i ? ov[i - 1] : v1, x2t1 = i ? ox2[i - 1] : x21;
				i8 ut = u[c], a, b, a2, b2, tmp, d = 0;
				a = add8(xt1, vt1), b = add8(y[c], ut), a2 = add8(x2t1, vt1), b2 = add8(y2[c], ut);
				if (!(flag & GDO_EZ_RIGHT) || !with_cigar) { /* left-aligned gaps, :235-243 (score-only: :196-200) */
					d = a > z ? 1 : 0;  z = max8(z, a);
					d = b > z ? 2 : d;  z = max8(z, b);
					d = a2 > z ? 3 : d; z = max8(z, a2);
					d = b2 > z ? 4 : d; z = max8(z, b2);
				} else { /* right-aligned, :282-290 */
					d = z > a ? 0 : 1;  z = max8(z, a);
					d = z > b ? d : 2;  z = max8(z, b);
					d = z > a2 ? d : 3; z = max8(z, a2);
					d = z > b2 ? d : 4; z = max8(z, b2);
				}
				z = min8(z, sc_mch);
				u[c] = sub8(z, vt1), v[c] = sub8(z, ut); /* block2, :58-66 */
				tmp = sub8(z, q), a = sub8(a, tmp), b = sub8(b, tmp);
				tmp = sub8(z, q2), a2 = sub8(a2, tmp), b2 = sub8(b2, tmp);
				if (!(flag & GDO_EZ_RIGHT) || !with_cigar) { /* :261-272 */
					x[c]  = sub8(a  > 0 ? a  : 0, (i8)qe),        d |= a  > 0 ? 0x08 : 0;
					y[c]  = sub8(b  > 0 ? b  : 0, (i8)qe),        d |= b  > 0 ? 0x10 : 0;
					x2[c] = sub8(a2 > 0 ? a2 : 0, (i8)(q2 + e2)), d |= a2 > 0 ? 0x20 : 0;
					y2[c] = sub8(b2 > 0 ? b2 : 0, (i8)(q2 + e2)), d |= b2 > 0 ? 0x40 : 0;
				} else { /* :308-319 */
					x[c]  = sub8(0 > a  ? 0 : a,  (i8)qe),        d |= 0 > a  ? 0 : 0x08;
					y[c]  = sub8(0 > b  ? 0 : b,  (i8)qe),        d |= 0 > b  ? 0 : 0x10;
					x2[c] = sub8(0 > a2 ? 0 : a2, (i8)(q2 + e2)), d |= 0 > a2 ? 0 : 0x20;
					y2[c] = sub8(0 > b2 ? 0 : b2, (i8)(q2 + e2)), d |= 0 > b2 ? 0 : 0x40;
				}
				if (pr) pr[i] = (u8t)d;
			}
			x1 = ox[15], v1 = ov[15], x21 = ox2[15];
		}
		if (!approx_max) { /* exact max with a 32-bit score array, :323-366 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else { /* approximate max, :367-383 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = v[0] - qe, last_H0_t = 0;
			if ((flag & GDO_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem);
	if (!approx_max) free(H);
	if (with_cigar) { /* :389-400 */
		int rev_cigar = !!(flag & GDO_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & GDO_EZ_EXTZ_ONLY))
			gdo_backtrack(rev_cigar, 0, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		else if (!ez->zdropped && (flag & GDO_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1;
			gdo_backtrack(rev_cigar, 0, p, off, off_end, n_col_ * 16, ez->mqe_t, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		} else if (ez->max_t >= 0 && ez->max_q >= 0)
			gdo_backtrack(rev_cigar, 0, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		free(p), free(off);
	}
}

/*
 * Single-affine variant, SR/ksw2_extz2_sse.c:31-312.  Differences from extd2: five state arrays, state kept
 * non-negative (x,y >= 0; u,v biased by q+e), the candidate z = s + 2(q+e) is compared signed against a but
 * unsigned against b (:34,:46-47,:185-203, SSE4.1 branch), boundary defaults x1 = v1 = 0 (:126-131).
 */
void gdo_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, gdo_extz_t *ez)
{
	int r, t, i, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, max_sc, min_sc;
	int with_cigar = !(flag & GDO_EZ_SCORE_ONLY), approx_max = !!(flag & GDO_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	u8t *mem, *qr, *sf, *p = 0;
	u8t *u, *v, *x, *y, *s;
	i8 sc_mch, sc_mis, sc_N;
	u8t qe2, max_sc_;

	reset_extz(ez);
	if (m <= 0 || qlen <= 0 || tlen <= 0) return;
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? (i8)-e : mat[m * m - 1];
	qe2 = (u8t)((q + e) * 2), max_sc_ = (u8t)(mat[0] + (q + e) * 2);

	if (w < 0) w = tlen > qlen ? tlen : qlen;
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;

	mem = (u8t *)calloc((size_t)tlen_ * 6 + qlen_ + 1, 16); /* u|v|x|y|s|sf|qr|slack, :98-100 */
	u = mem, v = u + tlen_ * 16, x = v + tlen_ * 16, y = x + tlen_ * 16, s = y + tlen_ * 16, sf = s + tlen_ * 16, qr = sf + tlen_ * 16;
	if (!approx_max) {
		H = (int32_t *)malloc((size_t)tlen_ * 16 * 4);
		for (t = 0; t < tlen_ * 16; ++t) H[t] = GDO_NEG_INF;
	}
	if (with_cigar) {
		p = (u8t *)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int *)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st, en, st0, en0, st_, en_;
		u8t x1, v1;
		u8t *qrr = qr + (qlen - 1 - r);
		if (!band(r, qlen, tlen, w, &st0, &en0)) { ez->zdropped = 1; break; }
		st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
		if (st > 0) { /* :126-131 */
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], v1 = v[st - 1];
			else x1 = v1 = 0;
		} else x1 = 0, v1 = r ? (u8t)q : 0;
		if (en >= r) y[r] = 0, u[r] = r ? (u8t)q : 0;
		fill_scores(flag, st0, en0, s, sf, qrr, m, mat, sc_mch, sc_mis, sc_N);
		st_ = st / 16, en_ = en / 16;
		if (with_cigar) off[r] = st, off_end[r] = en;
		for (t = st_; t <= en_; ++t) {
			u8t ox[16], ov[16];
			u8t *pr = with_cigar ? p + ((size_t)r * n_col_ - st_ + t) * 16 : 0;
			memcpy(ox, x + t * 16, 16), memcpy(ov, v + t * 16, 16);
			for (i = 0; i < 16; ++i) {
				int c = t * 16 + i;
				u8t z = (u8t)(s[c] + qe2), xt1 = i ? ox[i - 1] : x1, vt1 = i ? ov[i - 1] : v1, ut = u[c];
				u8t a = (u8t)(xt1 + vt1), b = (u8t)(y[c] + ut), d = 0, zz;
				if (!(flag & GDO_EZ_RIGHT) || !with_cigar) { /* :185-190 */
					d = (i8)a > (i8)z ? 1 : 0;
					z = (u8t)max8((i8)z, (i8)a);
					d = (i8)b > (i8)z ? 2 : d;
				} else { /* :206-211 */
					d = (i8)z > (i8)a ? 0 : 1;
					z = (u8t)max8((i8)z, (i8)a);
					d = (i8)z > (i8)b ? d : 2;
				}
				z = z > b ? z : b; /* block2 :46-52: unsigned max, then unsigned min */
				z = z < max_sc_ ? z : max_sc_;
				u[c] = (u8t)(z - vt1), v[c] = (u8t)(z - ut);
				zz = (u8t)(z - (u8t)q), a = (u8t)(a - zz), b = (u8t)(b - zz);
				if (!(flag & GDO_EZ_RIGHT) || !with_cigar) {
					x[c] = (i8)a > 0 ? a : 0, d |= (i8)a > 0 ? 0x08 : 0;
					y[c] = (i8)b > 0 ? b : 0, d |= (i8)b > 0 ? 0x10 : 0;
				} else {
					x[c] = 0 > (i8)a ? 0 : a, d |= 0 > (i8)a ? 0 : 0x08;
					y[c] = 0 > (i8)b ? 0 : b, d |= 0 > (i8)b ? 0 : 0x10;
				}
				if (pr) pr[i] = d;
			}
			x1 = ox[15], v1 = ov[15];
		}
		if (!approx_max) { /* :226-268 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] - qe : H[en0] + v[en0] - qe;
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v[t + i] - qe;
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v[t] - qe;
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else { /* :269-285 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t] - qe, d1 = u[last_H0_t + 1] - qe;
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t] - qe;
				} else {
					++last_H0_t, H0 += u[last_H0_t] - qe;
				}
				if ((flag & GDO_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e)) break;
			} else H0 = v[0] - qe - qe, last_H0_t = 0;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem);
	if (!approx_max) free(H);
	if (with_cigar) {
		int rev_cigar = !!(flag & GDO_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & GDO_EZ_EXTZ_ONLY))
			gdo_backtrack(rev_cigar, 0, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		else if (!ez->zdropped && (flag & GDO_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1;
			gdo_backtrack(rev_cigar, 0, p, off, off_end, n_col_ * 16, ez->mqe_t, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		} else if (ez->max_t >= 0 && ez->max_q >= 0)
			gdo_backtrack(rev_cigar, 0, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		free(p), free(off);
	}
}

/* ---- ksw_exts2 (splice-aware extension; SURVEY 8f rank 4: not called by GDiet) ---------------------------------------------
 * restates SR/ksw2_exts2_sse.c:34-416 in its SSE4.1 form (the form ksw2_dispatch.c:94-106 picks on any CPU with SSE4.1; the SSE2
 * emulation differs only in its score-only x2 update, :236-239, marked "TODO: check if this is correct" there).  Flat memory layout
 * as the reference's (:91-95), so that 16-byte stores past the end of s[] land where they land there. */
void gdo_ksw_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc,
                   gdo_extz_t *ez)
{
	int r, t, i, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, max_sc, min_sc, long_thres, long_diff;
	int with_cigar = !(flag & GDO_EZ_SCORE_ONLY), approx_max = !!(flag & GDO_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	u8t *mem, *qr, *sf, *p = 0;
	i8 *u, *v, *x, *y, *x2, *donor, *acceptor;
	u8t *s;
	i8 sc_mch, sc_mis, sc_N;

	reset_extz(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0 || q2 <= q + e) return; /* :72 */
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? (i8)-e : mat[m * m - 1];
	tlen_ = (tlen + 15) / 16;
	n_col_ = ((qlen < tlen ? qlen : tlen) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return; /* :90 */
	long_thres = (q2 - q) / e - 1;
	if (q2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * e - (q2 - q);

	mem = (u8t *)calloc((size_t)tlen_ * 9 + qlen_ + 1, 16); /* u|v|x|y|x2|donor|acceptor|s|sf|qr|slack, :97-101 */
	u = (i8 *)mem, v = u + tlen_ * 16, x = v + tlen_ * 16, y = x + tlen_ * 16, x2 = y + tlen_ * 16;
	donor = x2 + tlen_ * 16, acceptor = donor + tlen_ * 16;
	s = (u8t *)(acceptor + tlen_ * 16), sf = s + tlen_ * 16, qr = sf + tlen_ * 16;
	memset(u, -q - e, (size_t)tlen_ * 16 * 4);
	memset(x2, -q2, (size_t)tlen_ * 16);
	if (!approx_max) {
		H = (int32_t *)malloc((size_t)tlen_ * 16 * 4);
		for (t = 0; t < tlen_ * 16; ++t) H[t] = GDO_NEG_INF;
	}
	if (with_cigar) {
		p = (u8t *)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int *)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	if (flag & (GDO_EZ_SPLICE_FOR | GDO_EZ_SPLICE_REV)) { /* donor / acceptor signals, :119-171 (0/1/2/3 encoding assumed there too) */
		int semi_cost = flag & GDO_EZ_SPLICE_FLANK ? -noncan / 2 : 0;
		const int fw = !(flag & GDO_EZ_REV_CIGAR);
		memset(donor, -noncan, (size_t)tlen_ * 16);
		memset(acceptor, -noncan, (size_t)tlen_ * 16);
		for (t = 0; t < tlen - 4; ++t) {
			int can_type = 0;
			if (fw) {
				if ((flag & GDO_EZ_SPLICE_FOR) && target[t + 1] == 2 && target[t + 2] == 3) can_type = 1; /* GTr... */
				if ((flag & GDO_EZ_SPLICE_REV) && target[t + 1] == 1 && target[t + 2] == 3) can_type = 1; /* CTr... */
				if (can_type && (target[t + 3] == 0 || target[t + 3] == 2)) can_type = 2;
			} else {
				if ((flag & GDO_EZ_SPLICE_FOR) && target[t + 1] == 2 && target[t + 2] == 0) can_type = 1; /* GAy... */
				if ((flag & GDO_EZ_SPLICE_REV) && target[t + 1] == 1 && target[t + 2] == 0) can_type = 1; /* CAy... */
				if (can_type && (target[t + 3] == 1 || target[t + 3] == 3)) can_type = 2;
			}
			if (can_type) donor[t] = can_type == 2 ? 0 : (i8)semi_cost;
		}
		if (junc)
			for (t = 0; t < tlen - 1; ++t)
				if (fw ? (((flag & GDO_EZ_SPLICE_FOR) && (junc[t + 1] & 1)) || ((flag & GDO_EZ_SPLICE_REV) && (junc[t + 1] & 8)))
				       : (((flag & GDO_EZ_SPLICE_FOR) && (junc[t + 1] & 2)) || ((flag & GDO_EZ_SPLICE_REV) && (junc[t + 1] & 4))))
					donor[t] = add8(donor[t], junc_bonus);
		for (t = 2; t < tlen; ++t) {
			int can_type = 0;
			if (fw) {
				if ((flag & GDO_EZ_SPLICE_FOR) && target[t - 1] == 0 && target[t] == 2) can_type = 1; /* ...yAG */
				if ((flag & GDO_EZ_SPLICE_REV) && target[t - 1] == 0 && target[t] == 1) can_type = 1; /* ...yAC */
				if (can_type && (target[t - 2] == 1 || target[t - 2] == 3)) can_type = 2;
			} else {
				if ((flag & GDO_EZ_SPLICE_FOR) && target[t - 1] == 3 && target[t] == 2) can_type = 1; /* ...rTG */
				if ((flag & GDO_EZ_SPLICE_REV) && target[t - 1] == 3 && target[t] == 1) can_type = 1; /* ...rTC */
				if (can_type && (target[t - 2] == 0 || target[t - 2] == 2)) can_type = 2;
			}
			if (can_type) acceptor[t] = can_type == 2 ? 0 : (i8)semi_cost;
		}
		if (junc)
			for (t = 0; t < tlen; ++t)
				if (fw ? (((flag & GDO_EZ_SPLICE_FOR) && (junc[t] & 2)) || ((flag & GDO_EZ_SPLICE_REV) && (junc[t] & 4)))
				       : (((flag & GDO_EZ_SPLICE_FOR) && (junc[t] & 1)) || ((flag & GDO_EZ_SPLICE_REV) && (junc[t] & 8))))
					acceptor[t] = add8(acceptor[t], junc_bonus);
	}

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0, st_, en_;
		i8 x1, x21, v1;
		u8t *qrr = qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) { /* :183-191 */
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = (i8)(-q - e), x21 = (i8)-q2, v1 = (i8)(-q - e);
		} else {
			x1 = (i8)(-q - e), x21 = (i8)-q2;
			v1 = r == 0 ? (i8)(-q - e) : r < long_thres ? (i8)-e : r == long_thres ? (i8)long_diff : 0;
		}
		if (en >= r) {
			y[r] = (i8)(-q - e);
			u[r] = r == 0 ? (i8)(-q - e) : r < long_thres ? (i8)-e : r == long_thres ? (i8)long_diff : 0;
		}
		fill_scores(flag & GDO_EZ_GENERIC_SC, st0, en0, s, sf, qrr, m, mat, sc_mch, sc_mis, sc_N); /* :197-217 */
		st_ = st / 16, en_ = en / 16;
		if (with_cigar) off[r] = st, off_end[r] = en;
		for (t = st_; t <= en_; ++t) {
			i8 ox[16], ov[16], ox2[16];
			u8t *pr = with_cigar ? p + ((size_t)r * n_col_ - st_ + t) * 16 : 0;
			memcpy(ox, x + t * 16, 16), memcpy(ov, v + t * 16, 16), memcpy(ox2, x2 + t * 16, 16);
			for (i = 0; i < 16; ++i) {
				const int c = t * 16 + i;
				i8 z = (i8)s[c], xt1 = i ? ox[i - 1] : x1, vt1 = i ? ov[i - 1] : v1, x2t1 = i ? ox2[i - 1] : x21, ut = u[c];
				i8 a = add8(xt1, vt1), b = add8(y[c], ut), a2 = add8(x2t1, vt1), a2a = add8(a2, acceptor[c]), tmp, dn = donor[c];
				u8t d;
				if (!(flag & GDO_EZ_RIGHT) || !with_cigar) { /* :258-264 (and the score-only form :228-230: same z) */
					d = a > z ? 1 : 0;
					z = max8(z, a);
					d = b > z ? 2 : d;
					z = max8(z, b);
					d = a2a > z ? 3 : d;
					z = max8(z, a2a);
				} else { /* :303-309 */
					d = z > a ? 0 : 1;
					z = max8(z, a);
					d = z > b ? d : 2;
					z = max8(z, b);
					d = z > a2a ? d : 3;
					z = max8(z, a2a);
				}
				u[c] = sub8(z, vt1), v[c] = sub8(z, ut); /* block2 :61-66 */
				tmp = sub8(z, q);
				a = sub8(a, tmp), b = sub8(b, tmp), a2 = sub8(a2, sub8(z, q2));
				if (!(flag & GDO_EZ_RIGHT) || !with_cigar) {
					x[c] = sub8(a > 0 ? a : 0, (i8)qe), d |= a > 0 ? 0x08 : 0;
					y[c] = sub8(b > 0 ? b : 0, (i8)qe), d |= b > 0 ? 0x10 : 0;
					x2[c] = sub8(max8(a2, dn), q2), d |= a2 > dn ? 0x20 : 0;
				} else {
					x[c] = sub8(0 > a ? 0 : a, (i8)qe), d |= 0 > a ? 0 : 0x08;
					y[c] = sub8(0 > b ? 0 : b, (i8)qe), d |= 0 > b ? 0 : 0x10;
					x2[c] = sub8(max8(dn, a2), q2), d |= dn > a2 ? 0 : 0x20;
				}
				if (pr) pr[i] = d;
			}
			x1 = ox[15], v1 = ov[15], x21 = ox2[15];
		}
		if (!approx_max) { /* :339-383 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, 0)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else { /* :384-402 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = v[0] - qe, last_H0_t = 0;
			if ((flag & GDO_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, 0)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem);
	if (!approx_max) free(H);
	if (with_cigar) { /* :407-413 */
		int rev_cigar = !!(flag & GDO_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & GDO_EZ_EXTZ_ONLY))
			gdo_backtrack(rev_cigar, long_thres, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		else if (ez->max_t >= 0 && ez->max_q >= 0)
			gdo_backtrack(rev_cigar, long_thres, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		free(p), free(off);
	}
}
