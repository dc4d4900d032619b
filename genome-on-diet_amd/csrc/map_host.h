// Host side of the per-read path (LongReads variant): G1 candidate geometry, P1 mm_update_extra/mm_fix_cigar,
// P2 concatenate_cigars, P3 mm_set_sam_params, and the SAM record writer.  These stay on the host by design
// (float/double threshold arithmetic and ~10 bug-compatibility items, <1 % of the time); the device
// stages (map_stages.h, ksw_*.h) sit between gd_lr_link_and_boxes() and gd_lr_finish().
// Everything here is a from-scratch restatement; each function names the reference lines it follows.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "map_stages.h"
#include "map_post.h"

#define GD_F_NO_PRINT_2ND 0x4000
#define GD_F_SR 0x1000
#define GD_F_FRAG_MODE 0x2000
#define GD_NEG_INF_SCORE (-0x40000000)

// the fields of mm_mapopt_t / mm_idxopt_t the path reads (LR/minimap.h:137-214)
struct GdMapOpt {
	int64_t flag = 0;
	int k = 19, w = 19;
	int a = 1, b = 4, q = 6, e = 2, q2 = 26, e2 = 1;
	uint32_t bw = 1000;
	int min_dp_max = 40, best_n = 5;
	float q_occ_frac = 0.01f;
	int32_t mid_occ = 0, max_max_occ = 4095, occ_dist = 500;
	int max_frag_len = 0;
	uint32_t vt_dis = 100, vt_nb_loc = 3;
	float vt_cov = 0.03f, vt_f = 0.05f, vt_df1 = 0.01f, vt_df2 = 0.01f;
	uint32_t max_max_gap = 50000, max_min_gap = 4000;
	float max_seeds = 0.1f;
	// ShortReads variant only (SR/minimap.h:149-150,196-197, SR/main.c:163-172)
	float min_cnt = 1.0f, rec_threshold_frac = 0.0f, bw_frac = 0.05f;
	int32_t bw_min = 500, bw_max = 1500, af_max_loc = 20;
	GdPattern pat;
};

struct GdSeqInfo { std::string name; uint64_t offset; uint32_t len; };

// read-only view of the reference sequences (mm_idx_t::S, 4-bit packed; LR/mmpriv.h:31-32)
struct GdRefView {
	const uint32_t *S;
	const GdSeqInfo *seq;
	uint32_t n_seq;
};

// mm_idx_getseq (LR/index.c:157-166); returns the number of bases written or -1
static inline int gd_getseq(const GdRefView &R, uint32_t rid, uint32_t st, uint32_t en, uint8_t *out)
{
	if (rid >= R.n_seq || st >= R.seq[rid].len) return -1;
	if (en > R.seq[rid].len) en = R.seq[rid].len;
	const uint64_t st1 = R.seq[rid].offset + st, en1 = R.seq[rid].offset + en;
	for (uint64_t i = st1; i < en1; ++i) out[i - st1] = (uint8_t)(R.S[i >> 3] >> ((i & 7) << 2) & 0xf);
	return (int)(en - st);
}

// mm_reg1_t + mm_extra_t (LR/minimap.h:105-131) as one plain record
struct GdReg {
	int32_t id = 0, cnt = 0, rid = 0, score = 0, qs = 0, qe = 0, rs = 0, re = 0, parent = 0, subsc = 0, mlen = 0, blen = 0;
	uint32_t mapq = 0, rev = 0, sam_pri = 0;
	int32_t dp_score = 0, dp_max = 0;
	uint32_t n_ambi = 0;
	bool has_p = false;
	std::vector<uint32_t> cigar;
};

struct GdCand { // vt_t of LR/map.c:1033-1045 plus the DP box
	GdVt v;
	int next = -1; // index of the linked candidate (s1->next)
	int concat = 0, valid = 0;
	GdReg r;
	// DP box (LR/map.c:1659-1713)
	uint32_t target_id = 0, target_start = 0, target_end = 0, query_start = 0, query_end = 0, qlen = 0, tlen = 0;
	uint32_t qseq_off = 0; // offset into the strand-specific encoded read the DP query starts at
	int32_t exact_score = GD_NEG_INF_SCORE; // != NEG_INF: try the exact-match pre-filter (LR/map.c:1748)
};

// (GdCandBox -- a candidate between the box stage and the post-processing, trivially copyable -- lives in map_stages.h: the ShortReads
// box stage also runs on the device)
static inline GdCandBox gd_cand_box(const GdCand &c)
{
	GdCandBox b;
	b.v = c.v, b.next = c.next, b.concat = c.concat, b.valid = c.valid, b.target_id = c.target_id, b.target_start = c.target_start, b.target_end = c.target_end;
	b.query_start = c.query_start, b.query_end = c.query_end, b.qlen = c.qlen, b.tlen = c.tlen, b.qseq_off = c.qseq_off, b.exact_score = c.exact_score;
	return b;
}
static inline void gd_cand_unbox(const GdCandBox &b, GdCand &c)
{
	c.v = b.v, c.next = b.next, c.concat = b.concat, c.valid = b.valid, c.target_id = b.target_id, c.target_start = b.target_start, c.target_end = b.target_end;
	c.query_start = b.query_start, c.query_end = b.query_end, c.qlen = b.qlen, c.tlen = b.tlen, c.qseq_off = b.qseq_off, c.exact_score = b.exact_score;
}

// ---- G1 (second half): link candidates for CIGAR concatenation and derive the DP boxes, LR/map.c:1467-1590,1654-1713
static inline void gd_lr_link_and_boxes(std::vector<GdCand> &C, const GdMapOpt &O, const GdRefView &R, uint32_t qlen_sum)
{
	const unsigned n = (unsigned)C.size();
	const uint32_t max_max_gap = O.max_max_gap, max_min_gap = O.max_min_gap;
	for (unsigned i = 0; i < n; ++i) C[i].next = -1, C[i].concat = 0;
	// the reference clears next/concat only for the entries that passed the vt_f cut (:1386-1387); entries added by the
	// second voting round keep the zero-initialised values of their stack copy -- same thing.
	for (unsigned i = 0; i < n; i++) {
		GdVt &s1 = C[i].v;
		for (unsigned j = 0; j < n; j++) {
			if (j == i) continue;
			GdVt &s2 = C[j].v;
			if (!(C[j].concat == 0 && s1.str == s2.str && s1.chrom_id == s2.chrom_id)) continue;
			auto consider = [&](bool better) {
				if (C[i].next < 0) C[i].next = (int)j;
				else if (better) C[i].next = (int)j;
			};
			if (s1.str) {
				const bool better = C[i].next >= 0 && s2.last_query_loc > C[C[i].next].v.last_query_loc;
				const bool worse3 = C[i].next >= 0 && s2.last_query_loc < C[C[i].next].v.last_query_loc;
				if (s2.last_query_loc < s1.first_query_loc && s1.last_target_loc > s2.first_target_loc && s1.first_target_loc < s2.first_target_loc) {
					if (s2.last_query_loc + max_max_gap > s1.first_query_loc) consider(better);
				} else if (s2.last_query_loc < s1.first_query_loc && s1.last_target_loc < s2.first_target_loc) {
					if ((s2.last_query_loc + max_min_gap > s1.first_query_loc || (uint32_t)s1.last_target_loc + max_min_gap > (uint32_t)s2.first_target_loc) &&
					    s2.last_query_loc + max_max_gap > s1.first_query_loc && (uint32_t)s1.last_target_loc + max_max_gap > (uint32_t)s2.first_target_loc)
						consider(better);
				} else if (s2.last_query_loc > s1.first_query_loc && s1.last_target_loc < s2.first_target_loc &&
				           s2.last_query_loc < s1.last_query_loc && s2.first_query_loc < s1.first_query_loc) {
					if ((uint32_t)s1.last_target_loc + max_max_gap > (uint32_t)s2.first_target_loc) consider(worse3);
				}
			} else {
				const bool better = C[i].next >= 0 && s2.first_query_loc < C[C[i].next].v.first_query_loc;
				if (s1.last_query_loc < s2.first_query_loc && s1.last_target_loc > s2.first_target_loc && s1.first_target_loc < s2.first_target_loc) {
					if (s1.last_query_loc + max_max_gap > s2.first_query_loc) consider(better);
				} else if (s1.last_query_loc < s2.first_query_loc && s1.last_target_loc < s2.first_target_loc) {
					if ((s1.last_query_loc + max_min_gap > s2.first_query_loc || (uint32_t)s1.last_target_loc + max_min_gap > (uint32_t)s2.first_target_loc) &&
					    (uint32_t)s1.last_target_loc + max_max_gap > (uint32_t)s2.first_target_loc && s1.last_query_loc + max_max_gap > s2.first_query_loc)
						consider(better);
				} else if (s1.last_query_loc > s2.first_query_loc && s1.last_target_loc < s2.first_target_loc &&
				           s1.first_query_loc < s2.first_query_loc && s1.last_query_loc < s2.last_query_loc) {
					if ((uint32_t)s1.last_target_loc + max_max_gap > (uint32_t)s2.first_target_loc) consider(better);
				}
			}
		}
		if (C[i].next >= 0) { // adjust the boundaries, :1560-1589
			GdVt &s2 = C[C[i].next].v;
			C[C[i].next].concat = 1;
			if (s1.str) {
				if (s2.last_query_loc < s1.first_query_loc && s1.last_target_loc < s2.first_target_loc) {
					const uint32_t diffq = s1.first_query_loc - s2.last_query_loc, difft = (uint32_t)(s2.first_target_loc - s1.last_target_loc);
					const uint32_t mn = difft > diffq ? diffq : difft;
					s2.last_query_loc += mn, s1.last_target_loc += (int32_t)mn, s1.first_query_loc -= mn, s2.first_target_loc -= (int32_t)mn;
				}
			} else {
				if (s1.last_query_loc < s2.first_query_loc && s1.last_target_loc < s2.first_target_loc) {
					const uint32_t diffq = s2.first_query_loc - s1.last_query_loc, difft = (uint32_t)(s2.first_target_loc - s1.last_target_loc);
					const uint32_t mn = difft > diffq ? diffq : difft;
					s1.last_query_loc += mn, s1.last_target_loc += (int32_t)mn, s2.first_query_loc -= mn, s2.first_target_loc -= (int32_t)mn;
				}
			}
			if (s2.last_target_loc < s1.last_target_loc) s1.last_target_loc = s2.last_target_loc - 1;
		}
	}
	// DP boxes, :1654-1713
	for (unsigned i = 0; i < n; i++) {
		GdCand &c = C[i];
		c.valid = 1;
		const int str = (int)c.v.str;
		c.target_id = c.v.chrom_id;
		uint32_t target_start = (uint32_t)c.v.first_target_loc, target_end = (uint32_t)c.v.last_target_loc, query_start, query_end;
		if (str) query_end = qlen_sum - 1 - c.v.first_query_loc, query_start = qlen_sum - 1 - c.v.last_query_loc;
		else query_start = c.v.first_query_loc, query_end = c.v.last_query_loc;
		if (!(qlen_sum > 300)) {
			const int32_t chrom_len = c.target_id < R.n_seq ? (int32_t)R.seq[c.target_id].len : 0;
			if (target_start < query_start) query_start -= target_start, target_start = 0;
			else target_start -= query_start, query_start = 0;
			if ((uint32_t)chrom_len + query_end < qlen_sum + target_end) query_end += (uint32_t)chrom_len - target_end - 1, target_end = (uint32_t)chrom_len - 1;
			else target_end += qlen_sum - query_end - 1, query_end = qlen_sum - 1;
		}
		c.qseq_off = query_start;
		c.qlen = query_end - query_start + 1;
		c.tlen = target_end - target_start + 1;
		if (str) {
			const uint32_t tmp = qlen_sum - 1 - query_start;
			query_start = qlen_sum - 1 - query_end, query_end = tmp;
		}
		c.target_start = target_start, c.target_end = target_end, c.query_start = query_start, c.query_end = query_end;
		c.exact_score = (qlen_sum < 300 && c.qlen == c.tlen) ? (int32_t)(qlen_sum * (uint32_t)O.a) : GD_NEG_INF_SCORE;
	}
}

// ---- P1: mm_fix_cigar + mm_update_extra, LR/align.c:93-172,259-318: the arithmetic lives in map_post.h (shared with the device
// kernel); here only the record bookkeeping
static inline float gd_mg_log2(float x) { return gdp_mg_log2(x); }

// a record takes over what P1 computed for its alignment (on the host just now, or on the device right behind the backtrack)
static inline void gd_apply_post(GdReg &r, const GdPostOut &P)
{
	if (P.qshift) { if (r.rev) r.qe -= P.qshift; else r.qs += P.qshift; }
	r.rs += P.tshift;
	r.mlen = P.mlen, r.blen = P.blen, r.n_ambi += P.n_ambi, r.dp_max = P.dp_max;
}

static inline void gd_update_extra(GdReg &r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, int log_gap)
{
	if (!r.has_p) return;
	uint32_t n = (uint32_t)r.cigar.size();
	GdPostOut P;
	gdp_update_extra(r.cigar.data(), &n, qseq, tseq, mat, q, e, log_gap, &P);
	r.cigar.resize(n);
	gd_apply_post(r, P);
}

// ---- P2: concatenate_cigars, LR/map.c:41-640 (bug-compatible: junction search adds al_start_a twice, :267,:493) ---------
static inline int gd_concatenate_cigars(GdReg &rs_, const GdReg &re_, const uint8_t *qseq, uint8_t str, uint32_t read_len,
                                        const GdRefView &R, uint32_t sc_mch, uint32_t sc_mis, uint32_t gapo1, uint32_t gape1,
                                        uint32_t gapo2, uint32_t gape2)
{
	const uint32_t tstart = (uint32_t)rs_.rs, tend = (uint32_t)re_.re, tstart_junc = (uint32_t)re_.rs, tend_junc = (uint32_t)rs_.re;
	const uint32_t qstart = str ? read_len - rs_.qe : (uint32_t)rs_.qs, qend = str ? read_len - re_.qs : (uint32_t)re_.qe;
	const uint32_t qstart_junc = str ? read_len - re_.qe : (uint32_t)re_.qs, qend_junc = str ? read_len - rs_.qs : (uint32_t)rs_.qe;
	if (tend_junc <= tstart_junc && qend_junc <= qstart_junc) return 1;
	if (tend_junc >= tend || tstart >= tstart_junc) return 1;
	if (qend_junc >= qend || qstart >= qstart_junc) return 1;
	const uint32_t size_start = (uint32_t)(rs_.re - rs_.rs), size_end = (uint32_t)(re_.re - re_.rs);
	std::vector<uint8_t> tseq((size_start > size_end ? size_start : size_end) + 16, 0);
	auto gap = [&](uint32_t len) { const uint32_t p1 = gapo1 + len * gape1, p2 = gapo2 + len * gape2; return p1 < p2 ? p1 : p2; };
	auto oe = [&](uint32_t len, unsigned &o, unsigned &e) { const uint32_t p1 = gapo1 + len * gape1, p2 = gapo2 + len * gape2; if (p1 < p2) o = gapo1, e = gape1; else o = gapo2, e = gape2; };
	unsigned juncq, junct, cigar_pos;
	int score;
	const std::vector<uint32_t> &cs = rs_.cigar, &ce = re_.cigar;
	if (qend_junc > qstart_junc) {
		gd_getseq(R, (uint32_t)rs_.rid, tstart, tend_junc, tseq.data());
		const uint32_t jl = qend_junc - qstart_junc;
		std::vector<int> A(jl + 1, 0), B(jl + 1, 0);
		int al = 0;
		uint32_t toff = 0, qo = qstart;
		for (uint32_t i = 0; i < cs.size(); i++) {
			const uint32_t op = cs[i] & 0xf, len = cs[i] >> 4;
			if (op == 0) {
				for (unsigned j = 0; j < len; j++) {
					if (qo + j >= qstart_junc && qo + j - qstart_junc < jl) A[qo + j - qstart_junc] = al;
					if (qseq[qo + j] == tseq[toff + j]) al += (int)sc_mch; else al -= (int)sc_mis;
				}
				qo += len, toff += len;
			} else if (op == 1) {
				unsigned o, e;
				oe(len, o, e);
				if (qo + len <= qstart_junc) al -= (int)gap(len);
				else if (qo < qstart_junc) {
					al -= (int)(o + e * (qstart_junc - qo));
					for (unsigned j = 0; j < qo + len - qstart_junc; j++) { if (j < jl) A[j] = al; al -= (int)e; }
				} else {
					if (qo - qstart_junc < jl) A[qo - qstart_junc] = al;
					al -= (int)(o + e);
					for (unsigned j = 1; j < len; j++) { if (qo + j - qstart_junc < jl) A[qo + j - qstart_junc] = al; al -= (int)e; }
				}
				qo += len;
			} else if (op == 2) al -= (int)gap(len), toff += len;
			else if (op == 3) toff += len;
		}
		gd_getseq(R, (uint32_t)re_.rid, tstart_junc, tend, tseq.data());
		toff = 0;
		uint32_t qe = qstart_junc;
		al = re_.score;
		for (uint32_t i = 0; i < ce.size() && qe <= qend_junc; i++) {
			const uint32_t op = ce[i] & 0xf, len = ce[i] >> 4;
			if (op == 0) {
				for (unsigned j = 0; j < len; j++) {
					if (qe + j < qend_junc) {
						if (qseq[qe + j] == tseq[toff + j]) al -= (int)sc_mch; else al += (int)sc_mis;
						B[qe + j - qstart_junc] = al;
					} else break;
				}
				qe += len, toff += len;
			} else if (op == 1) {
				unsigned o, e;
				oe(len, o, e);
				al += (int)o;
				for (unsigned j = 0; j < len; j++) { if (qe + j < qend_junc) { al += (int)e; B[qe + j - qstart_junc] = al; } else break; }
				qe += len;
			} else if (op == 2) al += (int)gap(len), toff += len;
			else if (op == 3) toff += len;
		}
		int max_score = A[0] + B[0];
		juncq = 0;
		for (unsigned st = 1; st < jl; st++) {
			const int tot = A[st] + A[st]; // sic (:267)
			if (tot > max_score) max_score = tot, juncq = st;
		}
		score = max_score;
		juncq += qstart_junc;
		rs_.cigar.resize(cs.size() + ce.size() + 2);
		qo = qstart;
		uint32_t i, toffs = (uint32_t)rs_.rs;
		const uint32_t ncs = (uint32_t)(rs_.cigar.size() - ce.size() - 2);
		for (i = 0; i < ncs; i++) {
			const uint32_t op = rs_.cigar[i] & 0xf, len = rs_.cigar[i] >> 4;
			if (op == 0) {
				if (qo + len >= juncq) {
					const uint32_t nl = juncq - qo;
					rs_.cigar[i] = 0 | (nl << 4);
					qo += nl, toffs += nl;
					i++;
					break;
				}
				qo += len, toffs += len;
			} else if (op == 1) {
				if (qo + len >= juncq) { juncq = qo; break; }
				qo += len;
			} else if (op == 2 || op == 3) toffs += len;
		}
		junct = toffs, cigar_pos = i;
	} else {
		const uint32_t jl = tend_junc - tstart_junc;
		std::vector<int> A(jl + 1, 0), B(jl + 1, 0);
		gd_getseq(R, (uint32_t)rs_.rid, tstart, tend_junc, tseq.data());
		uint32_t toff = 0, qo = qstart;
		int al = 0;
		const uint32_t so = tstart_junc - tstart;
		for (uint32_t i = 0; i < cs.size(); i++) {
			const uint32_t op = cs[i] & 0xf, len = cs[i] >> 4;
			if (op == 0) {
				for (unsigned j = 0; j < len; j++) {
					if (toff + j >= so && toff + j - so < jl) A[toff + j - so] = al;
					if (qseq[qo + j] == tseq[toff + j]) al += (int)sc_mch; else al -= (int)sc_mis;
				}
				qo += len, toff += len;
			} else if (op == 2) {
				unsigned o, e;
				oe(len, o, e);
				if (toff + len <= so) al -= (int)gap(len);
				else if (toff < so) {
					al -= (int)(o + e * (so - toff));
					for (unsigned j = 0; j < toff + len - so; j++) { if (j < jl) A[j] = al; al -= (int)e; }
				} else {
					if (toff - so < jl) A[toff - so] = al;
					al -= (int)(o + e);
					for (unsigned j = 1; j < len; j++) { if (toff + j - so < jl) A[toff + j - so] = al; al -= (int)e; }
				}
				toff += len;
			} else if (op == 1) al -= (int)gap(len), qo += len;
			else if (op == 3) toff += len;
		}
		gd_getseq(R, (uint32_t)re_.rid, (uint32_t)re_.rs, (uint32_t)re_.re, tseq.data());
		toff = 0;
		uint32_t qe = qstart_junc;
		al = 0;
		const uint32_t eo = tend_junc - tstart_junc;
		for (uint32_t i = 0; i < ce.size() && toff <= eo; i++) {
			const uint32_t op = ce[i] & 0xf, len = ce[i] >> 4;
			if (op == 0) {
				for (unsigned j = 0; j < len; j++) {
					if (toff + j < eo) {
						if (qseq[qe + j] == tseq[toff + j]) al -= (int)sc_mch; else al += (int)sc_mis;
						B[toff + j] = al;
					} else break;
				}
				qe += len, toff += len;
			} else if (op == 2) {
				unsigned o, e;
				oe(len, o, e);
				al += (int)o;
				for (unsigned j = 0; j < len; j++) { if (toff + j < eo) { al += (int)e; B[toff + j] = al; } else break; }
				toff += len;
			} else if (op == 1) al += (int)gap(len), qe += len;
			else if (op == 3) toff += len;
		}
		int max_score = A[0] + B[0];
		junct = 0;
		for (unsigned st = 1; st < jl; st++) {
			const int tot = A[st] + A[st]; // sic (:493)
			if (tot > max_score) max_score = tot, junct = st;
		}
		score = max_score;
		junct += tstart_junc;
		rs_.cigar.resize(cs.size() + ce.size() + 2);
		qo = qstart;
		uint32_t i, toffs = (uint32_t)rs_.rs;
		const uint32_t ncs = (uint32_t)(rs_.cigar.size() - ce.size() - 2);
		for (i = 0; i < ncs; i++) {
			const uint32_t op = rs_.cigar[i] & 0xf, len = rs_.cigar[i] >> 4;
			if (op == 0) {
				if (toffs + len >= junct) {
					const uint32_t nl = junct - toffs;
					rs_.cigar[i] = 0 | (nl << 4);
					qo += nl, toffs += nl;
					i++;
					break;
				}
				qo += len, toffs += len;
			} else if (op == 2) {
				if (toffs + len >= junct) { junct = toffs; break; }
				toffs += len;
			} else if (op == 1) qo += len;
			else if (op == 3) toffs += len;
		}
		juncq = qo, cigar_pos = i;
	}
	uint32_t toffe = (uint32_t)re_.rs, qoffend = qstart_junc;
	unsigned i = cigar_pos;
	int crossed = 0;
	for (uint32_t j = 0; j < ce.size(); j++) {
		const uint32_t op = ce[j] & 0xf, len = ce[j] >> 4;
		if (crossed) rs_.cigar[i++] = ce[j];
		if (op == 0) qoffend += len, toffe += len;
		else if (op == 1) qoffend += len;
		else if (op == 2 || op == 3) toffe += len;
		if (crossed == 0 && qoffend >= juncq && toffe >= junct) {
			const uint32_t tar_len = toffe - junct, que_len = qoffend - juncq;
			if (que_len > tar_len) {
				const uint32_t len2 = que_len - tar_len;
				score -= (int)gap(len2);
				rs_.cigar[i++] = 1 | (len2 << 4);
				if (tar_len != 0) rs_.cigar[i++] = 0 | (tar_len << 4);
			} else if (que_len < tar_len) {
				const uint32_t len2 = tar_len - que_len;
				score -= (int)gap(len2);
				rs_.cigar[i++] = 2 | (len2 << 4);
				if (que_len != 0) rs_.cigar[i++] = 0 | (que_len << 4);
			} else rs_.cigar[i++] = 0 | (tar_len << 4);
			crossed = 1;
		}
	}
	rs_.cigar.resize(i);
	rs_.dp_score = score, rs_.score = score;
	if (str) rs_.qs = re_.qs; else rs_.qe = re_.qe;
	rs_.re = re_.re;
	return 0;
}

// ---- P3: mm_set_sam_params, LR/hit.c:494-557 (bug-compatible mapq ladder) ----------------------------------------------
static inline void gd_set_sam_params(std::vector<GdReg> &regs, unsigned qlen, unsigned match_score, unsigned max_nb_sec)
{
	const int n_regs = (int)regs.size();
	const int supp_threshold = (int)(0.8 * (float)(regs[0].qe - regs[0].qs));
	unsigned nb_sec = 0;
	int dp_max2 = 0;
	regs[0].sam_pri = 1, regs[0].parent = regs[0].id;
	for (int i = 1; i < n_regs; i++) {
		regs[i].sam_pri = 0;
		if (regs[i].qe - regs[i].qs > supp_threshold) nb_sec++, regs[i].mapq = 0, regs[i].parent = regs[i].id + 1, dp_max2 = regs[i].score;
		else regs[i].mapq = 60, regs[i].parent = regs[i].id;
	}
	for (int i = 1; i < n_regs - 1; i++) {
		if (regs[i].parent != regs[i].id) {
			for (int j = i + 1; j < n_regs; j++) {
				if (regs[j].parent == regs[j].id) { std::swap(regs[i], regs[j]); break; }
				else if (regs[i].score < regs[j].score) std::swap(regs[i], regs[j]);
			}
		}
	}
	if (max_nb_sec < nb_sec) nb_sec = max_nb_sec;
	if (nb_sec > 9) regs[0].mapq = 0;
	else if (nb_sec > 6) regs[0].mapq = 1;
	else if (nb_sec > 4) regs[0].mapq = 2;
	else if (nb_sec == 3) regs[0].mapq = 3;
	else if (nb_sec == 2) regs[0].mapq = 5;
	else if (nb_sec == 1) {
		const int dp_max = regs[0].score;
		const float identity = (float)regs[0].mlen / regs[0].blen;
		// 54 * identity * (dp_max - dp_max2) / (qlen * match_score - dp_max2) + 5 : float, then int, then unsigned int arithmetic
		regs[0].mapq = (uint32_t)(54 * identity * (dp_max - dp_max2) / (qlen * match_score - dp_max2) + 5) & 0xff;
	} else regs[0].mapq = 60;
}

// ---- LR/map.c:1734-1913: per-candidate record construction after the DP, concatenation, filtering, ordering ----------------
struct GdDpResult { int32_t score; const uint32_t *cigar; int32_t n_cigar; };

// enc_for / enc_rev: the read encoded 0-4 on the forward strand / reverse-complemented (LR/map.c:1622-1643)
static inline void gd_lr_finish(std::vector<GdCand> &C, const std::vector<GdDpResult> &dp, const GdMapOpt &O, const GdRefView &R,
                                uint32_t qlen_sum, const uint8_t *enc_for, const uint8_t *enc_rev, std::vector<GdReg> &out, FILE *trace = nullptr,
                                const GdPostOut *post = nullptr /* P1 already done on the device: one entry per candidate, CIGARs already fixed */)
{
	out.clear();
	const unsigned n = (unsigned)C.size();
	const int g = O.a, bb = O.b < 0 ? O.b : -O.b;
	int8_t mat[25];
	for (int i = 0; i < 25; ++i) mat[i] = (i / 5 == 4 || i % 5 == 4) ? 0 : (i / 5 == i % 5 ? (int8_t)g : (int8_t)bb);
	std::vector<uint8_t> tseq;
	for (unsigned i = 0; i < n; i++) {
		GdCand &c = C[i];
		if (dp[i].score == GD_NEG_INF_SCORE) { c.valid = 0; continue; }
		GdReg r;
		r.rid = (int32_t)c.target_id, r.score = dp[i].score, r.qs = (int32_t)c.query_start, r.qe = (int32_t)c.query_end + 1;
		r.rs = (int32_t)c.target_start, r.re = (int32_t)c.target_end + 1, r.rev = c.v.str;
		r.has_p = true;
		r.cigar.assign(dp[i].cigar, dp[i].cigar + dp[i].n_cigar);
		r.dp_score = dp[i].score;
		if (post) gd_apply_post(r, post[i]);
		else {
			tseq.assign((size_t)c.tlen + 16, 0);
			gd_getseq(R, c.target_id, c.target_start, c.target_end + 1, tseq.data());
			const uint8_t *qseq = (c.v.str ? enc_rev : enc_for) + c.qseq_off;
			gd_update_extra(r, qseq, tseq.data(), mat, (int8_t)O.q, (int8_t)O.e, !(O.flag & GD_F_SR));
		}
		const uint32_t clip0 = r.rev ? qlen_sum - r.qe : (uint32_t)r.qs, clip1 = r.rev ? (uint32_t)r.qs : qlen_sum - r.qe;
		if (!(clip0 < qlen_sum && clip1 < qlen_sum)) { c.valid = 0; continue; }
		c.r = r;
	}
	for (unsigned i = 0; i < n; i++) { // :1856-1874
		while (C[i].valid && C[i].next >= 0 && C[C[i].next].valid) {
			GdCand &nx = C[C[i].next];
			if (trace) // the reference's --print-seeds lines, :1858-1863
				fprintf(trace, "CONQ[%u, %u] || [%u, %u]\nCONT [%u, %u] || [%u, %u]\n", (unsigned)C[i].r.qs, (unsigned)C[i].r.qe, (unsigned)nx.r.qs, (unsigned)nx.r.qe,
				        (unsigned)C[i].r.rs, (unsigned)C[i].r.re, (unsigned)nx.r.rs, (unsigned)nx.r.re);
			if (gd_concatenate_cigars(C[i].r, nx.r, C[i].v.str ? enc_rev : enc_for, (uint8_t)C[i].v.str, qlen_sum, R, (uint32_t)O.a, (uint32_t)O.b,
			                          (uint32_t)O.q, (uint32_t)O.e, (uint32_t)O.q2, (uint32_t)O.e2) == 0) {
				nx.valid = 0;
				C[i].next = nx.next;
			} else C[i].next = -1;
		}
	}
	for (unsigned i = 0; i < n; i++)
		if (C[i].valid) {
			if (C[i].r.dp_score < O.min_dp_max) C[i].valid = 0;
		}
	for (unsigned i = 0; i < n; i++) // :1893-1908 -- bug-compatible: at most ONE swap per inserted record
		if (C[i].valid) {
			out.push_back(C[i].r);
			const size_t j = out.size() - 1;
			if (j > 0 && out[j].score > out[j - 1].score) std::swap(out[j], out[j - 1]);
		}
	if (!out.empty()) gd_set_sam_params(out, qlen_sum, (unsigned)O.a, (O.flag & GD_F_NO_PRINT_2ND) ? 0u : (unsigned)O.best_n);
}

// ---- G2: candidate geometry of the ShortReads variant, SR/map.c:779-839 -------------------------------------------------------
// Turns the voted diagonals into len x len DP boxes; candidates the reference skips (:796-801) are removed, order is kept.
// DP box of one short-read candidate (SR/map.c:780-930); false: the candidate is dropped
static inline bool gd_sr_box_one(const GdVt &v, const GdMapOpt &O, const GdRefView &R, uint32_t qlen_sum, GdCandBox &b)
{
	return gd_sr_box_core(v, O.k, O.a, qlen_sum, v.chrom_id < R.n_seq ? (int32_t)R.seq[v.chrom_id].len : 0, b);
}

static inline void gd_sr_boxes(std::vector<GdCand> &C, const GdMapOpt &O, const GdRefView &R, uint32_t qlen_sum)
{
	std::vector<GdCand> out;
	out.reserve(C.size());
	for (const GdCand &c0 : C) {
		GdCandBox b;
		if (!gd_sr_box_one(c0.v, O, R, qlen_sum, b)) continue;
		GdCand c = c0;
		const int next = c.next, concat = c.concat; // (untouched by the box stage)
		gd_cand_unbox(b, c);
		c.next = next, c.concat = concat;
		out.push_back(c);
	}
	C.swap(out);
}

// ---- SR/map.c:931-984: records after the DP -- mm_update_extra, clip / min_dp_max filter, insertion by score, mapq ------------
static inline void gd_sr_finish(std::vector<GdCand> &C, const std::vector<GdDpResult> &dp, const GdMapOpt &O, const GdRefView &R,
                                uint32_t qlen_sum, const uint8_t *enc_for, const uint8_t *enc_rev, std::vector<GdReg> &out, const GdPostOut *post = nullptr)
{
	out.clear();
	const int g = O.a, bb = O.b < 0 ? O.b : -O.b;
	int8_t mat[25];
	for (int i = 0; i < 25; ++i) mat[i] = (i / 5 == 4 || i % 5 == 4) ? 0 : (i / 5 == i % 5 ? (int8_t)g : (int8_t)bb);
	out.reserve(C.size());
	for (size_t i = 0; i < C.size(); ++i) {
		const GdCand &c = C[i];
		// a band that emptied leaves score = KSW_NEG_INF and no CIGAR; dp_score < min_dp_max then drops the record (:955-958)
		if (dp[i].score == GD_NEG_INF_SCORE && O.min_dp_max > GD_NEG_INF_SCORE) continue;
		GdReg r;
		r.rid = (int32_t)c.target_id, r.score = dp[i].score, r.qs = (int32_t)c.query_start, r.qe = (int32_t)c.query_end + 1;
		r.rs = (int32_t)c.target_start, r.re = (int32_t)c.target_end + 1, r.rev = c.v.str;
		r.has_p = true;
		r.cigar.assign(dp[i].cigar, dp[i].cigar + (dp[i].n_cigar > 0 ? dp[i].n_cigar : 0));
		r.dp_score = dp[i].score;
		if (post) gd_apply_post(r, post[i]);
		else {
			static thread_local std::vector<uint8_t> tseq; // (scratch kept per thread; only P1 on the host comes here)
			tseq.assign((size_t)c.tlen + 16, 0);
			gd_getseq(R, c.target_id, c.target_start, c.target_end + 1, tseq.data());
			const uint8_t *qseq = (c.v.str ? enc_rev : enc_for) + c.qseq_off;
			gd_update_extra(r, qseq, tseq.data(), mat, (int8_t)O.q, (int8_t)O.e, !(O.flag & GD_F_SR));
		}
		const uint32_t clip0 = r.rev ? qlen_sum - r.qe : (uint32_t)r.qs, clip1 = r.rev ? (uint32_t)r.qs : qlen_sum - r.qe;
		if (!(clip0 < qlen_sum && clip1 < qlen_sum) || r.dp_score < O.min_dp_max) continue;
		out.push_back(std::move(r));
		for (size_t k = out.size() - 1; k > 0; k--) { // full insertion by descending score (:965-973)
			if (out[k].score > out[k - 1].score) std::swap(out[k], out[k - 1]);
			else break;
		}
	}
	if (!out.empty()) gd_set_sam_params(out, qlen_sum, (unsigned)O.a, (O.flag & GD_F_NO_PRINT_2ND) ? 0u : (unsigned)O.best_n);
}

// ---- SAM record, LR/format.c:412-599 for the single-segment case the path produces ------------------------------------------
static inline void gd_fmt_int(std::string &s, long long v) // decimal, as "%lld" (hand-rolled: a SAM record holds a dozen of them)
{
	char b[24];
	int n = 24;
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do b[--n] = (char)('0' + u % 10), u /= 10; while (u);
	if (v < 0) b[--n] = '-';
	s.append(b + n, (size_t)(24 - n));
}

// write_tags, LR/format.c:292-324 (the records of this path always carry a CIGAR; inv / split / trans_strand are never set)
static inline void gd_write_tags(std::string &s, const GdReg &r)
{
	const char type = r.id == r.parent ? 'P' : 'S';
	if (r.has_p) {
		s += "\tNM:i:"; gd_fmt_int(s, r.blen - r.mlen + (int)r.n_ambi);
		s += "\tms:i:"; gd_fmt_int(s, r.dp_max);
		s += "\tAS:i:"; gd_fmt_int(s, r.dp_score);
		s += "\tnn:i:"; gd_fmt_int(s, r.n_ambi);
	}
	s += "\ttp:A:"; s += type; s += "\tcm:i:"; gd_fmt_int(s, r.cnt); s += "\ts1:i:"; gd_fmt_int(s, r.score);
	if (r.parent == r.id) { s += "\ts2:i:"; gd_fmt_int(s, r.subsc); }
	if (r.has_p) { // mm_event_identity, LR/align.c:961-982
		int32_t n_gap = 0, n_gapo = 0;
		for (uint32_t cg : r.cigar) if ((cg & 0xf) == 1 || (cg & 0xf) == 2) ++n_gapo, n_gap += (int32_t)(cg >> 4);
		const double ident = (double)r.mlen / (r.blen + (int)r.n_ambi - n_gap + n_gapo);
		const double div = 1.0 - ident;
		char buf[16];
		if (div == 0.0) buf[0] = '0', buf[1] = 0;
		else snprintf(buf, 16, "%.4f", 1.0 - ident);
		s += "\tde:f:"; s += buf;
	}
}

// ---- PAF record, mm_write_paf3 (LR/format.c:326-367) with rep_len = 0; reg_idx < 0: the unmapped line of --paf-no-hit ----------
#define GD_F_OUT_CG 0x020
#define GD_F_PAF_NO_HIT 0x8000000
#define GD_F_QSTRAND (0x100000000LL)
static inline void gd_write_paf(std::string &s, const GdRefView &R, const char *qname, int l_seq, const std::vector<GdReg> &regs, int reg_idx, int64_t opt_flag)
{
	s += qname; s += '\t'; gd_fmt_int(s, l_seq);
	if (reg_idx < 0 || reg_idx >= (int)regs.size()) { s += "\t0\t0\t*\t*\t0\t0\t0\t0\t0\t0\trl:i:0"; return; }
	const GdReg &r = regs[reg_idx];
	s += '\t'; gd_fmt_int(s, r.qs); s += '\t'; gd_fmt_int(s, r.qe); s += '\t'; s += "+-"[r.rev]; s += '\t';
	s += R.seq[r.rid].name; s += '\t'; gd_fmt_int(s, R.seq[r.rid].len);
	if ((opt_flag & GD_F_QSTRAND) && r.rev) { s += '\t'; gd_fmt_int(s, (int64_t)R.seq[r.rid].len - r.re); s += '\t'; gd_fmt_int(s, (int64_t)R.seq[r.rid].len - r.rs); }
	else { s += '\t'; gd_fmt_int(s, r.rs); s += '\t'; gd_fmt_int(s, r.re); }
	s += '\t'; gd_fmt_int(s, r.mlen); s += '\t'; gd_fmt_int(s, r.blen); s += '\t'; gd_fmt_int(s, r.mapq);
	gd_write_tags(s, r);
	s += "\trl:i:0";
	if (r.has_p && (opt_flag & GD_F_OUT_CG)) {
		s += "\tcg:Z:";
		for (uint32_t cg : r.cigar) { gd_fmt_int(s, cg >> 4); s += "MIDNSHP=XB"[cg & 0xf]; }
	}
}

static inline void gd_write_sam(std::string &s, const GdRefView &R, const char *qname, const char *seq, const char *qual, int l_seq,
                                const std::vector<GdReg> &regs, int reg_idx, int64_t opt_flag)
{
	static const char comp_tab[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ"; // seq_comp_table (LR/bseq.c) restricted to letters
	const GdReg *r = reg_idx >= 0 && reg_idx < (int)regs.size() ? &regs[reg_idx] : nullptr;
	s += qname;
	int flag = 0;
	if (!r) flag |= 0x4;
	else {
		if (r->rev) flag |= 0x10;
		if (r->parent != r->id) flag |= 0x100;
		else if (!r->sam_pri) flag |= 0x800;
	}
	s += '\t'; gd_fmt_int(s, flag);
	auto put_seq = [&](const char *p, int l, int rev, int comp) {
		if (!rev) { s.append(p, (size_t)l); return; }
		const size_t at = s.size();
		s.resize(at + (size_t)l);
		char *d = &s[at];
		for (int i = 0; i < l; ++i) {
			int c = (unsigned char)p[l - 1 - i];
			if (comp && c < 128) {
				if (c >= 'A' && c <= 'Z') c = comp_tab[c - 'A'];
				else if (c >= 'a' && c <= 'z') c = comp_tab[c - 'a'] + 32;
			}
			d[i] = (char)c;
		}
	};
	if (!r) {
		s += "\t*\t0\t0\t*\t*\t0\t0\t";
		put_seq(seq, l_seq, 0, 0);
		s += '\t';
		if (qual) put_seq(qual, l_seq, 0, 0); else s += '*';
	} else {
		s += '\t'; s += R.seq[r->rid].name; s += '\t'; gd_fmt_int(s, r->rs + 1); s += '\t'; gd_fmt_int(s, r->mapq); s += '\t';
		if (!r->has_p) s += '*';
		else {
			const uint32_t clip0 = r->rev ? (uint32_t)(l_seq - r->qe) : (uint32_t)r->qs, clip1 = r->rev ? (uint32_t)r->qs : (uint32_t)(l_seq - r->qe);
			const char clip_char = (flag & 0x800) ? 'H' : 'S';
			if (clip0) { gd_fmt_int(s, clip0); s += clip_char; }
			for (uint32_t cg : r->cigar) { gd_fmt_int(s, cg >> 4); s += "MIDNSHP=XB"[cg & 0xf]; }
			if (clip1) { gd_fmt_int(s, clip1); s += clip_char; }
		}
		s += "\t*\t0\t0\t";
		if ((flag & 0x900) == 0) {
			put_seq(seq, l_seq, r->rev, r->rev);
			s += '\t';
			if (qual) put_seq(qual, l_seq, r->rev, 0); else s += '*';
		} else if (flag & 0x100) s += "*\t*";
		else {
			put_seq(seq + r->qs, r->qe - r->qs, r->rev, r->rev);
			s += '\t';
			if (qual) put_seq(qual + r->qs, r->qe - r->qs, r->rev, 0); else s += '*';
		}
		gd_write_tags(s, *r);
		// SA tag for the primary line when supplementary alignments exist, :565-590
		if (r->parent == r->id && r->has_p && regs.size() > 1) {
			int n_sa = 0;
			for (size_t i = 0; i < regs.size(); ++i) if ((int)i != reg_idx && regs[i].parent == regs[i].id && regs[i].has_p) ++n_sa;
			if (n_sa > 0) {
				s += "\tSA:Z:";
				for (size_t i = 0; i < regs.size(); ++i) {
					const GdReg *q = &regs[i];
					if ((int)i == reg_idx || q->parent != q->id || !q->has_p) continue;
					int l_M, l_I = 0, l_D = 0;
					if (q->qe - q->qs < q->re - q->rs) l_M = q->qe - q->qs, l_D = (q->re - q->rs) - l_M;
					else l_M = q->re - q->rs, l_I = (q->qe - q->qs) - l_M;
					const int clip5 = q->rev ? l_seq - q->qe : q->qs, clip3 = q->rev ? q->qs : l_seq - q->qe;
					s += R.seq[q->rid].name; s += ','; gd_fmt_int(s, q->rs + 1); s += ','; s += "+-"[q->rev]; s += ',';
					if (clip5) { gd_fmt_int(s, clip5); s += 'S'; }
					if (l_M) { gd_fmt_int(s, l_M); s += 'M'; }
					if (l_I) { gd_fmt_int(s, l_I); s += 'I'; }
					if (l_D) { gd_fmt_int(s, l_D); s += 'D'; }
					if (clip3) { gd_fmt_int(s, clip3); s += 'S'; }
					s += ','; gd_fmt_int(s, q->mapq); s += ','; gd_fmt_int(s, q->blen - q->mlen + (int)q->n_ambi); s += ';';
				}
			}
		}
	}
	s += "\trl:i:0"; // rep_len is never set on this path (bug-compat item 8): mm_write_sam3 prints rl:i:0
	(void)opt_flag;
}
