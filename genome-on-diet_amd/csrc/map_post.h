// P1 on raw arrays: mm_fix_cigar + mm_update_extra (LR/align.c:93-172,259-318), written once for the host stages (map_host.h) and
// for the device kernel map_post_kernel (map_kernels.hip.h: one thread per alignment right behind the backtrack, so that the
// host never has to fetch 15 kbp of reference and walk 15 kbp of CIGAR per candidate).
// Floating point follows the reference operation by operation -- a double accumulator, float mg_log2 -- with contraction into
// fused multiply-adds switched off (the reference's x86-64 build has none; the GPU would fuse by default and round differently).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GDP_HD __host__ __device__ __forceinline__
#else
#define GDP_HD static inline
#endif

struct GdPostOut { // what P1 leaves in mm_reg1_t / mm_extra_t besides the CIGAR
	int32_t qshift, tshift; // leading I / D removed from the CIGAR: the caller moves qs (or qe on the reverse strand) / rs by these
	int32_t mlen, blen, dp_max;
	uint32_t n_ambi;
};

GDP_HD float gdp_mg_log2(float x) // LR/mmpriv.h:146-157
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	union { float f; uint32_t i; } z = {x};
	float log_2 = (float)(((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	const float t1 = -0.34484843f * z.f;
	const float t2 = t1 + 2.02466578f;
	const float t3 = t2 * z.f;
	const float t4 = t3 - 0.67487759f;
	log_2 += t4;
	return log_2;
}

// mm_fix_cigar on cg[0..*n): left-aligns indels, merges I/D runs, drops zero-length ops and a leading I / D
GDP_HD void gdp_fix_cigar(uint32_t *cg, uint32_t *n_io, const uint8_t *qseq, const uint8_t *tseq, int32_t *qshift, int32_t *tshift)
{
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	uint32_t k, n = *n_io;
	*qshift = *tshift = 0;
	if (n <= 1) return;
	for (k = 0; k < n; ++k) { // indel left alignment
		const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == 0) toff += len, qoff += len;
		else if (op == 1 || op == 2) {
			if (k > 0 && k < n - 1 && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
				int l, prev_len = (int)(cg[k - 1] >> 4);
				if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break; }
				else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break; }
				if (l > 0) cg[k - 1] -= (uint32_t)l << 4, cg[k + 1] += (uint32_t)l << 4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	for (k = 0; k + 2 < n; ++k) { // fix CIGAR like 5I6D7I   (k < n_cigar - 2 with unsigned n_cigar >= 2)
		if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k + 1] & 0xf) == 3) {
			uint32_t l, s[3] = {0, 0, 0};
			for (l = k; l < n; ++l) {
				const uint32_t op = cg[l] & 0xf;
				if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				cg[k] = s[1] << 4 | 1, cg[k + 1] = s[2] << 4 | 2;
				for (k += 2; k < l; ++k) cg[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		uint32_t l = 0;
		for (k = 0; k < n; ++k) if (cg[k] >> 4 != 0) cg[l++] = cg[k];
		n = l;
		for (k = l = 0; k < n; ++k)
			if (k == n - 1 || (cg[k] & 0xf) != (cg[k + 1] & 0xf)) cg[l++] = cg[k];
			else cg[k + 1] += cg[k] >> 4 << 4;
		n = l;
	}
	if ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2) { // get rid of leading I or D
		const int32_t l = (int32_t)(cg[0] >> 4);
		if ((cg[0] & 0xf) == 1) *qshift = l;
		else *tshift = l;
		--n;
		for (k = 0; k < n; ++k) cg[k] = cg[k + 1];
	}
	*n_io = n;
}

// mm_update_extra: fix the CIGAR, then mlen / blen / n_ambi / dp_max.  mat: the 5 x 5 matrix of the caller (match a, mismatch -b,
// 0 against N); an index past it (query byte 7 = N of a reverse-complemented read against a target N) is undefined in the reference
// and taken as 0.  n_ambi_in: the record's count before the call (the reference adds to it).
GDP_HD void gdp_update_extra(uint32_t *cg, uint32_t *n_io, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, int log_gap,
                             GdPostOut *out)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	int32_t qshift, tshift, toff = 0, qoff = 0;
	double s = 0.0, mx = 0.0;
	gdp_fix_cigar(cg, n_io, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift;
	int32_t blen = 0, mlen = 0;
	uint32_t n_ambi_tot = 0;
	const uint32_t n = *n_io;
	if (!log_gap) {
		// The short-read form (no logarithmic gap cost): every term of the running score is an integer, so the reference's double accumulator
		// holds integers and int32 arithmetic gives the same values.  Eight bases per load; a group without an ambiguous base (codes 0..3
		// on both sides: no bit 2 anywhere) scores mat[0] / mat[1] by equality, any other group goes through the table base by base.
		int32_t si = 0, mxi = 0;
		const int32_t mch = mat[0], mis = mat[1];
		bool uniform = true; // (one match and one mismatch score among the four bases: what mm_set_opt builds; else everything through the table)
		for (int i = 0; i < 4; ++i)
			for (int j = 0; j < 4; ++j) uniform &= mat[i * 5 + j] == (i == j ? mch : mis);
		for (uint32_t k = 0; k < n; ++k) {
			const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
			if (op == 0) {
				int n_ambi = 0, n_diff = 0;
				uint32_t l = 0;
				for (; l + 8 <= len; l += 8) {
					uint64_t a, b;
					__builtin_memcpy(&a, qseq + qoff + l, 8), __builtin_memcpy(&b, tseq + toff + l, 8);
					if (uniform && !((a | b) & 0x0404040404040404ull)) {
						const uint64_t x = a ^ b;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
						for (int j = 0; j < 8; ++j) {
							const int d = ((x >> (8 * j)) & 0xff) != 0;
							n_diff += d;
							si += d ? mis : mch;
							if (si < 0) si = 0;
							else mxi = mxi > si ? mxi : si;
						}
					} else
						for (int j = 0; j < 8; ++j) {
							const int cq = (int)((a >> (8 * j)) & 0xff), ct = (int)((b >> (8 * j)) & 0xff);
							if (ct > 3 || cq > 3) ++n_ambi;
							else if (ct != cq) ++n_diff;
							si += (ct * 5 + cq < 25) ? (int32_t)mat[ct * 5 + cq] : 0;
							if (si < 0) si = 0;
							else mxi = mxi > si ? mxi : si;
						}
				}
				for (; l < len; ++l) {
					const int cq = qseq[qoff + l], ct = tseq[toff + l];
					if (ct > 3 || cq > 3) ++n_ambi;
					else if (ct != cq) ++n_diff;
					si += (ct * 5 + cq < 25) ? (int32_t)mat[ct * 5 + cq] : 0;
					if (si < 0) si = 0;
					else mxi = mxi > si ? mxi : si;
				}
				blen += (int32_t)len - n_ambi, mlen += (int32_t)len - (n_ambi + n_diff), n_ambi_tot += (uint32_t)n_ambi;
				toff += len, qoff += len;
			} else if (op == 1 || op == 2) {
				int n_ambi = 0;
				for (uint32_t l = 0; l < len; ++l)
					if ((op == 1 ? qseq[qoff + l] : tseq[toff + l]) > 3) ++n_ambi;
				blen += (int32_t)len - n_ambi, n_ambi_tot += (uint32_t)n_ambi;
				si -= (int32_t)(q + e);
				if (si < 0) si = 0;
				if (op == 1) qoff += len; else toff += len;
			} else if (op == 3) toff += len;
		}
		out->qshift = qshift, out->tshift = tshift, out->mlen = mlen, out->blen = blen, out->n_ambi = n_ambi_tot;
		out->dp_max = mxi; // ((int32_t)(mx + .499) of an integer-valued mx)
		return;
	}
	for (uint32_t k = 0; k < n; ++k) {
		const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			for (uint32_t l = 0; l < len; ++l) {
				const int cq = qseq[qoff + l], ct = tseq[toff + l];
				if (ct > 3 || cq > 3) ++n_ambi;
				else if (ct != cq) ++n_diff;
				s += (ct * 5 + cq < 25) ? (double)mat[ct * 5 + cq] : 0.0;
				if (s < 0) s = 0;
				else mx = mx > s ? mx : s;
			}
			blen += (int32_t)len - n_ambi, mlen += (int32_t)len - (n_ambi + n_diff), n_ambi_tot += (uint32_t)n_ambi;
			toff += len, qoff += len;
		} else if (op == 1 || op == 2) {
			int n_ambi = 0;
			for (uint32_t l = 0; l < len; ++l)
				if ((op == 1 ? qseq[qoff + l] : tseq[toff + l]) > 3) ++n_ambi;
			blen += (int32_t)len - n_ambi, n_ambi_tot += (uint32_t)n_ambi;
			if (log_gap) {
				const double pen = (double)e * (double)gdp_mg_log2(1.0f + (float)len);
				const double tot = (double)q + pen;
				s -= tot;
			} else s -= (double)(q + e);
			if (s < 0) s = 0;
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	out->qshift = qshift, out->tshift = tshift, out->mlen = mlen, out->blen = blen, out->n_ambi = n_ambi_tot;
	out->dp_max = (int32_t)(mx + .499);
}
