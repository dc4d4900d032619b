// S1-S7 / V1,V3 of SURVEY.md 8a: the per-read seeding and voting stages of mm_map_frag (LongReads variant),
// written once as host/device functions.  On the GPU they run one read per thread (map_kernels.hip.h); the same
// code is compiled for the host by the CPU tests (tests/emul/stages_host.cpp) to be checked against the reference
// without a GPU.  Input sequences are nt4 bytes (0-3 = ACGT, 4 = N): seq_nt4_table maps those bytes to themselves
// (LR/sketch.c:11-18, first five entries), so sketching the encoded read equals sketching the ASCII read.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GDM_HD __host__ __device__ inline
#define GDM_HDM __host__ __device__ inline // member functions
#define GDM_HDI __host__ __device__ __forceinline__ // must be inlined for the address space of its pointer arguments to be seen
#else
#define GDM_HD static inline
#define GDM_HDM inline
#define GDM_HDI static inline
#endif

#define GDM_MAX_W 64      // window sizes above this are rejected by the planner (the reference allows < 256)
#define GDM_MAX_ONES 40   // LR/sketch.c:1927

struct GdMini { uint64_t x, y; }; // mm128_t: x = hash<<8 | span, y = rid<<32 | lastPos<<1 | strand (LR/minimap.h:69-72)

struct GdPattern { // -Z / -W
	int W, ones;
	int ones_loc[GDM_MAX_ONES];
	char Z[64];
};

static inline bool gd_pattern_init(GdPattern &P, const char *Z, int W)
{
	if (W <= 0 || W > 63) return false;
	P.W = W, P.ones = 0;
	for (int g = 0; g < W; ++g) {
		P.Z[g] = Z[g];
		if (Z[g] == '1') { if (P.ones >= GDM_MAX_ONES) return false; P.ones_loc[P.ones++] = g; }
	}
	P.Z[W] = 0;
	return P.ones > 0;
}

// Thomas Wang's invertible integer hash, LR/sketch.c:25-34
GDM_HD uint64_t gd_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

// number of pattern-1 positions in [shift, len)  (LR/sketch.c:1942-1948, :2187-2193)
GDM_HD unsigned gd_diet_len(const GdPattern &P, unsigned len, unsigned shift)
{
	if (shift > len) return 0; // the reference underflows here (reads shorter than the pattern phase); never reached by real reads
	unsigned d = ((len - shift) / P.W) * P.ones, off = (len - shift) % P.W;
	for (unsigned i = 0; i < off; ++i)
		if (P.Z[i] == '1') ++d;
	return d;
}

// The winnowing automaton shared by mm_sketch (:1640-1765), mm_sketch2_sub (:1780-1905) and mm_sketch3 (:1968-2138).
// `emit(m)` is called for every minimizer the reference pushes, in the same order; it returns true to stop
// (the two capped variants return from inside the loop).  final_ge: the final flush uses l >= w+k-1 (sketch2/3 and the
// AVX-512 mm_sketch, :607) instead of l > w+k-1 (scalar mm_sketch, :1760).
// Chunked execution (used by the wave-parallel kernels, where each lane sketches one slice of the read): the automaton's
// state after >= k-1+w steps is a pure function of the last w k-mers and of l (the run length of non-N bases, which only
// matters below w+k), so a slice [i_emit, i_end) can be produced exactly by starting w+k steps earlier with the true l
// and suppressing every emission before i_emit.  i_begin/l_init describe that warm-up start; do_final = the slice is the
// last one and performs the end-of-sequence flush.
// win / wstride: storage of the w-entry window (entry j at win[j * wstride]); nullptr = a local array.  The wave-parallel device
// kernels pass lane-interleaved LDS (a dynamically indexed local array lives in scratch memory there, an order of magnitude slower).
// EXT_WIN (a compile-time choice, and the whole chain force-inlined on the device): with a run-time choice between the caller's
// LDS window and a local array the window pointer is a generic one -- every access a flat_load / flat_store through the vector
// memory path (~500 cycles) instead of a ds_read / ds_write (~64), and the local array costs 1 KB of scratch per lane even if unused.
template <class Emit, bool EXT_WIN = false>
GDM_HDI void gd_sketch_range(const uint8_t *str, unsigned i_begin, unsigned i_emit, unsigned i_end, int l_init, bool do_final, int w, int k,
                             uint32_t rid, unsigned shift, const GdPattern &P, bool final_ge, Emit &emit, GdMini *win = nullptr, int wstride = 1)
{
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
	uint64_t kmer[2] = {0, 0};
	GdMini own[EXT_WIN ? 1 : GDM_MAX_W], mn = {UINT64_MAX, UINT64_MAX};
	GdMini *const wb = EXT_WIN ? win : own;
	const int ws = EXT_WIN ? wstride : 1;
#define buf(j_) wb[(j_) * ws]
	int l = l_init, buf_pos = 0, min_pos = 0;
	for (int j = 0; j < w; ++j) buf(j).x = buf(j).y = UINT64_MAX;
#define GDM_EMIT(m_) do { if (i >= i_emit) { if (emit(m_)) return; } } while (0)
	// get_real_location (:20-23) without a division per base: quotient and remainder of i by P.ones carried along; a pattern with
	// a single 1 (every preset of the reference: "10") needs no table look-up per base
	unsigned iq = i_begin / P.ones, ir = i_begin % P.ones;
	const unsigned loc0 = P.ones_loc[0];
	const bool one_1 = P.ones == 1;
	// EXT_WIN (the wave-parallel device kernels): the read is fetched eight aligned bytes at a time -- one global load per four
	// bases of a "10" pattern instead of one per base (the caller's buffer has 8 bytes of slack at its end)
	uint64_t cache8 = 0;
	uintptr_t cache_at = ~(uintptr_t)0;
	for (unsigned i = i_begin; i < i_end; ++i) {
		const unsigned real = iq * P.W + (one_1 ? loc0 : P.ones_loc[ir]) + shift;
		if (++ir == P.ones) ir = 0, ++iq;
		uint8_t sb;
		if (EXT_WIN) {
			const uintptr_t a = (uintptr_t)(str + real), al = a & ~(uintptr_t)7;
			if (al != cache_at) cache8 = *reinterpret_cast<const uint64_t *>(al), cache_at = al;
			sb = (uint8_t)(cache8 >> (8 * (a & 7)));
		} else sb = str[real];
		const int c = sb < 4 ? sb : 4;
		GdMini info = {UINT64_MAX, UINT64_MAX};
		if (c < 4) {
			const int span = l + 1 < k ? l + 1 : k;
			kmer[0] = (kmer[0] << 2 | (uint64_t)c) & mask;
			kmer[1] = (kmer[1] >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			++l;
			if (kmer[0] != kmer[1]) { // palindromic k-mers are skipped (strand unknown)
				const int z = kmer[0] < kmer[1] ? 0 : 1;
				if (l >= k) {
					info.x = gd_hash64(kmer[z], mask) << 8 | (uint64_t)span;
					info.y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)real << 1 | (uint64_t)z;
				}
			}
		} else {
			if (l >= w + k - 1 && mn.x != UINT64_MAX) GDM_EMIT(mn);
			l = 0;
		}
		buf(buf_pos) = info;
		if (info.x <= mn.x) { // a new minimum (ties: the rightmost wins); write the old one
			if (l >= w + k && mn.x != UINT64_MAX) GDM_EMIT(mn);
			mn = info, min_pos = buf_pos;
		} else if (buf_pos == min_pos) { // the old minimum left the window
			if (l >= w + k - 1 && mn.x != UINT64_MAX) GDM_EMIT(mn);
			// rescan in window order (buf_pos+1 .. w-1, 0 .. buf_pos), the LAST of equal minima wins (:2045-2052).  Only the hashes are
			// read, four independent loads at a time -- in the wave-parallel kernels some lane is here on almost every step, so the
			// whole wavefront pays this loop's latency per base -- and y once, for the winner.
			uint64_t bx = UINT64_MAX;
			int bp = buf_pos + 1 < w ? buf_pos + 1 : 0;
			for (int t = 0; t < w; t += 4) {
				uint64_t xs[4];
				int js[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					int j = buf_pos + 1 + t + q;
					if (j >= w) j -= w;
					js[q] = j;
					xs[q] = t + q < w ? buf(j).x : UINT64_MAX;
				}
#pragma unroll
				for (int q = 0; q < 4; ++q)
					if (t + q < w && bx >= xs[q]) bx = xs[q], bp = js[q];
			}
			mn.x = bx, mn.y = buf(bp).y, min_pos = bp;
			if (l >= w + k - 1 && mn.x != UINT64_MAX) { // identical k-mers in the window (:2053-2089), same order
				for (int t = 0; t < w; t += 4) {
					uint64_t xs[4];
					int js[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						int j = buf_pos + 1 + t + q;
						if (j >= w) j -= w;
						js[q] = j;
						xs[q] = t + q < w ? buf(j).x : UINT64_MAX;
					}
#pragma unroll
					for (int q = 0; q < 4; ++q)
						if (t + q < w && mn.x == xs[q]) {
							const GdMini d = buf(js[q]);
							if (mn.y != d.y) GDM_EMIT(d);
						}
				}
			}
		}
		if (l == w + k - 1 && mn.x != UINT64_MAX) { // first full window: identical k-mers were not written yet
			for (int j = buf_pos + 1; j < w; ++j)
				if (mn.x == buf(j).x && buf(j).y != mn.y) GDM_EMIT(buf(j));
			for (int j = 0; j < buf_pos; ++j)
				if (mn.x == buf(j).x && buf(j).y != mn.y) GDM_EMIT(buf(j));
		}
		if (++buf_pos == w) buf_pos = 0;
	}
#undef GDM_EMIT
#undef buf
	if (do_final && (final_ge ? l >= w + k - 1 : l > w + k - 1) && mn.x != UINT64_MAX) emit(mn);
}

template <class Emit>
GDM_HD void gd_sketch_core(const uint8_t *str, unsigned diet_len, int w, int k, uint32_t rid, unsigned shift,
                           const GdPattern &P, bool final_ge, Emit &emit)
{
	gd_sketch_range(str, 0, 0, diet_len, 0, true, w, k, rid, shift, P, final_ge, emit);
}

// slice [i_emit, i_end) of a sketch over diet_len sparsified bases: finds the warm-up start and the run length of non-N
// bases in front of it, then runs the automaton (see gd_sketch_range)
template <class Emit, bool EXT_WIN = false>
GDM_HDI void gd_sketch_slice(const uint8_t *str, unsigned diet_len, unsigned i_emit, unsigned i_end, int w, int k, uint32_t rid,
                             unsigned shift, const GdPattern &P, bool final_ge, Emit &emit, GdMini *win = nullptr, int wstride = 1)
{
	const unsigned wu = (unsigned)(w + k);
	const unsigned i_begin = i_emit > wu ? i_emit - wu : 0;
	int l0 = 0;
	if (i_begin > 0) { // run of non-N bases ending right before i_begin, saturated at w+k+1 (only l < w+k is ever compared)
		while (l0 < w + k + 1 && (unsigned)l0 < i_begin) {
			const unsigned i = i_begin - 1 - (unsigned)l0;
			const unsigned real = (i / P.ones) * P.W + P.ones_loc[i % P.ones] + shift;
			if (str[real] >= 4) break;
			++l0;
		}
	}
	gd_sketch_range<Emit, EXT_WIN>(str, i_begin, i_emit, i_end, l0, i_end >= diet_len, w, k, rid, shift, P, final_ge, emit, win, wstride);
}

// ---- flat index view (device mirror of mm_idx_t's buckets; built by map_index.h) -------------------------------
// open-addressing table keyed by the minimizer hash value (x >> 8); every key owns a run of the position array
// sorted by y, exactly the order mm_idx_get returns (LR/index.c:84-100,241-264; singletons included).
struct GdIdxView {
	int32_t k, w;
	uint32_t tbits;        // table has 1 << tbits slots
	const uint64_t *tkey;  // UINT64_MAX = empty
	const uint64_t *tval;  // start << 32 | n   (start < 2^32: at most 4G positions)
	const uint64_t *pos;   // y values
};

GDM_HD uint32_t gd_idx_slot(uint64_t minier, uint32_t tbits) { return (uint32_t)((minier * 0x9E3779B97F4A7C15ULL) >> (64 - tbits)); }

// mm_idx_get: number of occurrences (0 if absent) and the start of the run in pos[]
GDM_HD uint32_t gd_idx_get(const GdIdxView &I, uint64_t minier, uint64_t *start)
{
	const uint32_t m = (1u << I.tbits) - 1;
	uint32_t s = gd_idx_slot(minier, I.tbits);
	for (;;) {
		const uint64_t kk = I.tkey[s];
		if (kk == minier) {
			const uint64_t v = I.tval[s];
			*start = v >> 32;
			return (uint32_t)v;
		}
		if (kk == UINT64_MAX) return 0;
		s = (s + 1) & m;
	}
}

// ---- S1: mm_sketch2 (LR/sketch.c:2143-2225) -------------------------------------------------------------------
struct GdEmitCount { // mm_sketch2_sub: append, count, stop at the cap (:1817-1820 ...)
	GdMini *out;
	unsigned n, cap_n, max_out;
	bool overflow;
	GDM_HDM bool operator()(const GdMini &m)
	{
		if (n < max_out) out[n] = m; else overflow = true;
		++n;
		return n == cap_n;
	}
};

// writes all phases' seeds to out[] back to back, counts per phase to shift_n[W]; returns total (or ~0u on overflow)
GDM_HD unsigned gd_sketch2(const uint8_t *str, int len, int w, int k, const GdPattern &P, float max_seeds, GdMini *out,
                           unsigned max_out, uint32_t *shift_n)
{
	unsigned len_crop, total = 0;
	uint32_t cap;
	if (max_seeds < 1) len_crop = (unsigned)((float)max_seeds * len), cap = UINT32_MAX;
	else len_crop = len, cap = (uint32_t)max_seeds;
	for (int shift = 0; shift < P.W; ++shift) {
		const unsigned dl = gd_diet_len(P, len_crop, shift);
		GdEmitCount e = {out + total, 0, cap, max_out - total, false};
		gd_sketch_core(str, dl, w, k, 0, shift, P, true, e);
		if (e.overflow) return ~0u;
		shift_n[shift] = e.n;
		total += e.n;
		if (cap == UINT32_MAX) len_crop = len, cap = e.n;
	}
	return total;
}

// ---- S3: mm_get_shift (LR/seed.c:166-194) -----------------------------------------------------------------------
GDM_HD unsigned gd_get_shift(const GdIdxView &I, const GdMini *mv, const uint32_t *shift_n, int W)
{
	unsigned shift = 0, best = 0;
	const GdMini *p = mv;
	for (int i = 0; i < W; ++i) {
		unsigned cur = 0;
		for (uint32_t k = 0; k < shift_n[i]; ++k) {
			uint64_t st;
			cur += gd_idx_get(I, p[k].x >> 8, &st);
		}
		if (cur > best) shift = i, best = cur;
		p += shift_n[i];
	}
	return shift;
}

// ---- S2: mm_sketch3 (LR/sketch.c:1908-2139) ---------------------------------------------------------------------
struct GdEmitCap { // push; when p->n == MAX_NB_SEEDS return that seed's real location (:2010-2012 ...)
	GdMini *out;
	unsigned n, cap_n, max_out;
	uint32_t ret;
	bool overflow;
	GDM_HDM bool operator()(const GdMini &m)
	{
		if (n < max_out) out[n] = m; else overflow = true;
		++n;
		if (n == cap_n) { ret = (uint32_t)(m.y >> 1); return true; }
		return false;
	}
};

// returns tmp_extracted_len; *n_out = number of minimizers (or ~0u on overflow of out[])
GDM_HD unsigned gd_sketch3(const uint8_t *str, unsigned len, int w, int k, const GdPattern &P, int shift,
                           uint32_t max_nb_seeds, GdMini *out, unsigned max_out, unsigned *n_out)
{
	if (shift < 0) shift = 0;
	const unsigned dl = gd_diet_len(P, len, (unsigned)shift);
	GdEmitCap e = {out, 0, max_nb_seeds, max_out, len, false};
	gd_sketch_core(str, dl, w, k, 0, (unsigned)shift, P, true, e);
	*n_out = e.overflow ? ~0u : e.n;
	return e.ret;
}

// ---- small in-place sorts used per read (one thread per read; n is a few hundred to a few thousand) ---------------
// bottom-up merge sort of 64-bit keys with a scratch buffer; returns the buffer that holds the result
GDM_HD uint64_t *gd_msort_u64(uint64_t *a, uint64_t *tmp, unsigned n)
{
	for (unsigned wd = 1; wd < n; wd <<= 1) {
		for (unsigned lo = 0; lo < n; lo += 2 * wd) {
			unsigned mid = lo + wd < n ? lo + wd : n, hi = lo + 2 * wd < n ? lo + 2 * wd : n;
			unsigned i = lo, j = mid, o = lo;
			while (i < mid && j < hi) tmp[o++] = a[j] < a[i] ? a[j++] : a[i++];
			while (i < mid) tmp[o++] = a[i++];
			while (j < hi) tmp[o++] = a[j++];
		}
		uint64_t *t = a; a = tmp; tmp = t;
	}
	return a;
}

// ---- S4: mm_seed_mz_flt (LR/seed.c:5-29) ---------------------------------------------------------------------------
// drop query minimizers whose hash occurs more than q_occ_max times in the read and more than n*q_occ_frac.
// scratch: 2*n uint64.  returns the new count.
GDM_HD unsigned gd_mz_flt(GdMini *mv, unsigned n, int32_t q_occ_max, float q_occ_frac, uint64_t *scratch)
{
	if ((int64_t)n <= (int64_t)q_occ_max || q_occ_frac <= 0.0f || q_occ_max <= 0) return n;
	// all spans are equal (k), so sorting x sorts the hash; find over-frequent x values
	uint64_t *a = scratch, *t = scratch + n;
	for (unsigned i = 0; i < n; ++i) a[i] = mv[i].x;
	uint64_t *s = gd_msort_u64(a, t, n);
	uint64_t *bad = (s == a) ? t : a; // list of over-frequent x values (sorted)
	unsigned nbad = 0;
	for (unsigned st = 0, i = 1; i <= n; ++i)
		if (i == n || s[i] != s[st]) {
			const int32_t cnt = (int32_t)(i - st);
			if (cnt > q_occ_max && (float)cnt > (float)n * q_occ_frac) bad[nbad++] = s[st]; // nbad <= st: no overlap with unread s[]
			st = i;
		}
	if (nbad == 0) return n;
	unsigned j = 0;
	for (unsigned i = 0; i < n; ++i) {
		unsigned lo = 0, hi = nbad; // binary search
		const uint64_t x = mv[i].x;
		while (lo < hi) { unsigned md = (lo + hi) >> 1; if (bad[md] < x) lo = md + 1; else hi = md; }
		if (!(lo < nbad && bad[lo] == x) && mv[i].x != 0) mv[j++] = mv[i];
	}
	return j;
}

// ---- S5: mm_collect_matches2 = mm_seed_collect_all + mm_seed_select + filter (LR/seed.c:36-164) ---------------------
struct GdSeed { // mm_seed_t (LR/mmpriv.h:43-49) reduced to what the live path reads
	uint32_t n;      // occurrences in the index
	uint32_t q_pos;  // lastPos<<1 | strand of the query minimizer
	uint32_t start;  // start of the run in the index position array
	uint32_t flt;
};

#define GDM_MAX_MAX_HIGH_OCC 128

GDM_HD void gd_seed_select(int32_t n, GdSeed *a, int len, int max_occ, int max_max_occ, int dist)
{
	uint64_t b[GDM_MAX_MAX_HIGH_OCC];
	int32_t i, last0, m;
	if (n == 0 || n == 1) return;
	for (i = m = 0; i < n; ++i)
		if ((int32_t)a[i].n > max_occ) ++m;
	if (m == 0) return;
	for (i = 0, last0 = -1; i <= n; ++i) {
		if (i == n || (int32_t)a[i].n <= max_occ) {
			if (i - last0 > 1) {
				const int32_t ps = last0 < 0 ? 0 : (int32_t)(a[last0].q_pos >> 1);
				const int32_t pe = i == n ? len : (int32_t)(a[i].q_pos >> 1);
				int32_t j, k, st = last0 + 1, en = i;
				int32_t max_high_occ = (int32_t)((double)(pe - ps) / dist + .499);
				if (max_high_occ > 0) {
					if (max_high_occ > GDM_MAX_MAX_HIGH_OCC) max_high_occ = GDM_MAX_MAX_HIGH_OCC;
					for (j = st, k = 0; j < en && k < max_high_occ; ++j, ++k) b[k] = (uint64_t)a[j].n << 32 | (uint32_t)j;
					// the reference keeps the k smallest in a max-heap; the kept SET only depends on "replace the current
					// maximum when a strictly smaller n arrives", so a linear scan for the maximum is equivalent
					for (; j < en; ++j) {
						int32_t mx = 0;
						for (int32_t q = 1; q < k; ++q)
							if (b[q] > b[mx]) mx = q;
						if ((int32_t)a[j].n < (int32_t)(b[mx] >> 32)) b[mx] = (uint64_t)a[j].n << 32 | (uint32_t)j;
					}
					for (j = 0; j < k; ++j) a[(uint32_t)b[j]].flt = 1;
				}
				for (j = st; j < en; ++j) a[j].flt ^= 1;
				for (j = st; j < en; ++j)
					if ((int32_t)a[j].n > max_max_occ) a[j].flt = 1;
			}
			last0 = i;
		}
	}
}

// second half of mm_collect_matches2: m[0..n_all) holds one entry per query minimizer in sketch order (n == 0: absent from
// the index); drop the absent ones, apply mm_seed_select / the max_occ filter, keep the unfiltered.  Returns the number
// of kept seeds; *n_a = total occurrences of the kept seeds.
GDM_HD int gd_collect_finish(GdSeed *m, int n_all, int qlen, int max_occ, int max_max_occ, int dist, int64_t *n_a)
{
	int n_m0 = 0, n_m = 0;
	for (int i = 0; i < n_all; ++i)
		if (m[i].n) m[n_m0++] = m[i];
	if (dist > 0 && max_max_occ > max_occ) gd_seed_select(n_m0, m, qlen, max_occ, max_max_occ, dist);
	else
		for (int i = 0; i < n_m0; ++i)
			if ((int32_t)m[i].n > max_occ) m[i].flt = 1;
	*n_a = 0;
	for (int i = 0; i < n_m0; ++i)
		if (!m[i].flt) *n_a += m[i].n, m[n_m++] = m[i];
	return n_m;
}

// first half: one index probe per query minimizer (mm_seed_collect_all, LR/seed.c:36-62)
GDM_HD void gd_collect_probe(const GdIdxView &I, const GdMini &mz, GdSeed &q)
{
	uint64_t st = 0;
	q.n = gd_idx_get(I, mz.x >> 8, &st), q.q_pos = (uint32_t)mz.y, q.start = (uint32_t)st, q.flt = 0;
}

GDM_HD int gd_collect_matches2(const GdIdxView &I, const GdMini *mv, unsigned n_mv, int qlen, int max_occ, int max_max_occ,
                               int dist, GdSeed *m, int64_t *n_a)
{
	for (unsigned i = 0; i < n_mv; ++i) gd_collect_probe(I, mv[i], m[i]);
	return gd_collect_finish(m, (int)n_mv, qlen, max_occ, max_max_occ, dist, n_a);
}

// ---- S6: collect_seed_hits (LR/map.c:861-955): occurrences -> loc_t on the forward / reverse strand -------------------
struct GdLoc { uint64_t target; uint32_t query, pad; }; // loc_t (LR/map.c:738-741), 16-byte stride

#define GDM_F_FOR_ONLY 0x100000
#define GDM_F_REV_ONLY 0x200000

GDM_HD void gd_seed_hits(const GdIdxView &I, const GdSeed *m, int n_m, int64_t flag, uint32_t tmp_extracted_len,
                         GdLoc *a_for, GdLoc *a_rev, unsigned *n_for, unsigned *n_rev)
{
	unsigned nf = 0, nr = 0;
	for (int i = 0; i < n_m; ++i) {
		const GdSeed &q = m[i];
		for (uint32_t k = 0; k < q.n; ++k) {
			const uint64_t r = I.pos[(uint64_t)q.start + k];
			if (flag & (GDM_F_FOR_ONLY | GDM_F_REV_ONLY)) { // skip_seed, LR/map.c:724-730
				if ((r & 1) == (q.q_pos & 1)) { if (flag & GDM_F_REV_ONLY) continue; }
				else if (flag & GDM_F_FOR_ONLY) continue;
			}
			const uint32_t qpos = q.q_pos >> 1;
			const unsigned str = (unsigned)((r & 1) ^ (q.q_pos & 1));
			uint32_t loc = (uint32_t)r >> 1;
			const uint64_t chrom = r >> 32;
			if (str) {
				loc = loc + qpos;
				a_rev[nr].target = chrom << 32 | loc, a_rev[nr].query = qpos, a_rev[nr].pad = 0, ++nr;
			} else {
				loc = loc + tmp_extracted_len - qpos;
				a_for[nf].target = chrom << 32 | loc, a_for[nf].query = qpos, a_for[nf].pad = 0, ++nf;
			}
		}
	}
	*n_for = nf, *n_rev = nr;
}

// ---- S7: sort by target.  Any order among equal targets gives the same vote result (only min/max/count over a run
// are used and ref_loc takes the -- equal -- target), so a plain key sort replaces the reference's run merge ---------
GDM_HD GdLoc *gd_sort_locs(GdLoc *a, GdLoc *tmp, unsigned n)
{
	for (unsigned wd = 1; wd < n; wd <<= 1) {
		for (unsigned lo = 0; lo < n; lo += 2 * wd) {
			unsigned mid = lo + wd < n ? lo + wd : n, hi = lo + 2 * wd < n ? lo + 2 * wd : n;
			unsigned i = lo, j = mid, o = lo;
			while (i < mid && j < hi) tmp[o++] = a[j].target < a[i].target ? a[j++] : a[i++];
			while (i < mid) tmp[o++] = a[i++];
			while (j < hi) tmp[o++] = a[j++];
		}
		GdLoc *t = a; a = tmp; tmp = t;
	}
	return a;
}

// ---- V1: vote, LongReads (LR/map.c:1052-1180) -----------------------------------------------------------------------------
struct GdVt { // vt_t (LR/map.c:1033-1045) without the alignment record
	uint32_t chrom_id;
	int32_t first_target_loc, last_target_loc;
	uint32_t first_query_loc, last_query_loc;
	uint32_t score;
	uint32_t str;
};

GDM_HD uint64_t gd_vt_loc(const GdLoc &c, int str, int32_t tel)
{
	// (tmp_extracted_len - query) is evaluated in 32-bit unsigned arithmetic in the reference (int32 - uint32)
	return str ? (c.target - c.query) : c.target - (uint64_t)(uint32_t)((uint32_t)tel - c.query);
}

GDM_HD void gd_vt_insert(GdVt *seqs, unsigned &out_len, unsigned max_n, unsigned counter, uint64_t ft, uint64_t lt, uint32_t fq,
                         uint32_t lq, int str, bool &skip)
{
	skip = false;
	if (out_len == max_n) {
		if (seqs[out_len - 1].score >= counter) { skip = true; return; }
	} else out_len++;
	GdVt v;
	v.chrom_id = (uint32_t)(ft >> 32), v.first_target_loc = (int32_t)(uint32_t)ft, v.last_target_loc = (int32_t)(uint32_t)lt;
	v.first_query_loc = fq, v.last_query_loc = lq, v.str = (uint32_t)str, v.score = counter;
	seqs[out_len - 1] = v;
	for (unsigned k = out_len - 1; k > 0; k--) {
		if (seqs[k].score > seqs[k - 1].score) { GdVt t = seqs[k]; seqs[k] = seqs[k - 1]; seqs[k - 1] = t; }
		else break;
	}
}

// LocSrc: anything with operator[](unsigned) -> GdLoc, read front to back (a pointer on the host; on the device a wavefront-wide
// prefetching view, see map_kernels.hip.h)
template <class LocSrc>
GDM_HD void gd_vote(LocSrc loc, unsigned len, int str, GdVt *seqs, unsigned *nb_seqs, uint32_t vt_distance, int32_t tel,
                    unsigned max_n, uint32_t cov_thr)
{
	if (len == 0) return;
	unsigned out_len = *nb_seqs, counter = 1;
	uint64_t ft = gd_vt_loc(loc[0], str, tel), lt = ft, ref_loc = loc[0].target;
	uint32_t fq = loc[0].query, lq = loc[0].query;
	for (unsigned i = 1; i < len; i++) {
		const GdLoc cur = loc[i];
		if (cur.target - ref_loc <= vt_distance) {
			counter++;
			if (cur.query < fq) fq = cur.query, ref_loc = cur.target;
			if (cur.query > lq) lq = cur.query;
			const uint64_t l = gd_vt_loc(cur, str, tel);
			if (l > lt) lt = l;
			if (l < ft) ft = l;
		} else {
			if (lq - fq > cov_thr) {
				bool skip;
				gd_vt_insert(seqs, out_len, max_n, counter, ft, lt, fq, lq, str, skip);
			}
			ft = lt = gd_vt_loc(cur, str, tel);
			fq = lq = cur.query, ref_loc = cur.target, counter = 1;
		}
	}
	if (lq - fq > cov_thr) {
		bool skip;
		gd_vt_insert(seqs, out_len, max_n, counter, ft, lt, fq, lq, str, skip);
	}
	*nb_seqs = out_len;
}

// ---- V3: vote_2 (LR/map.c:1182-1271): best single run restricted to query interval (min,max) ------------------------------
template <class LocSrc>
GDM_HD void gd_vote2(LocSrc loc, unsigned len, int str, GdVt *vt, uint32_t vt_distance, int32_t tel, uint32_t qmin, uint32_t qmax)
{
	if (len == 0) return;
	GdVt best = *vt;
	unsigned counter = 1;
	uint64_t ft = gd_vt_loc(loc[0], str, tel), lt = ft, ref_loc = loc[0].target;
	uint32_t fq = loc[0].query, lq = loc[0].query;
	for (unsigned i = 1; i <= len; i++) {
		if (i < len && loc[i].target - ref_loc <= vt_distance) {
			const GdLoc cur = loc[i];
			if (cur.query < qmax && cur.query > qmin) {
				counter++;
				if (cur.query < fq) fq = cur.query, ref_loc = cur.target;
				if (cur.query > lq) lq = cur.query;
				const uint64_t l = gd_vt_loc(cur, str, tel);
				if (l > lt) lt = l;
				if (l < ft) ft = l;
			}
		} else {
			if (counter > best.score && lq < qmax && fq > qmin) {
				best.chrom_id = (uint32_t)(ft >> 32), best.first_target_loc = (int32_t)(uint32_t)ft, best.last_target_loc = (int32_t)(uint32_t)lt;
				best.first_query_loc = fq, best.last_query_loc = lq, best.str = (uint32_t)str, best.score = counter;
			}
			if (i < len) {
				const GdLoc cur = loc[i];
				ft = lt = gd_vt_loc(cur, str, tel);
				fq = lq = cur.query, ref_loc = cur.target, counter = 1;
			}
		}
	}
	*vt = best;
}

// ---- V1+G1 (first half): the candidate list of one long read, LR/map.c:1342-1445 ---------------------------------------------
struct GdLrVoteOpt {
	uint32_t vt_dis, vt_nb_loc, bw;
	float vt_cov, vt_f, vt_df1, vt_df2;
	int32_t k;
};

#define GDM_MAX_VT 24 // vt_nb_loc + 2 (LongReads) / AF_max_loc (ShortReads, default 20) must fit

// seqs[] must hold vt_nb_loc+2 entries.  Returns the number of candidates (0: unmapped).
template <class LocSrc>
GDM_HD unsigned gd_lr_candidates(LocSrc a_for, unsigned n_for, LocSrc a_rev, unsigned n_rev, uint32_t qlen_sum,
                                 int32_t tel, const GdLrVoteOpt &O, GdVt *seqs)
{
	const uint32_t cov_thr = (uint32_t)((float)qlen_sum * O.vt_cov); // :1342
	unsigned nb = 0;
	gd_vote(a_for, n_for, 0, seqs, &nb, O.vt_dis, tel, O.vt_nb_loc, cov_thr);
	gd_vote(a_rev, n_rev, 1, seqs, &nb, O.vt_dis, tel, O.vt_nb_loc, cov_thr);
	if (nb == 0) return 0;
	// density filter, :1355-1363 -- bug-compatible: the copy goes the wrong way (seqs[i] = seqs[nb_df])
	unsigned nb_df = 0;
	for (unsigned i = 0; i < nb; i++)
		if ((float)seqs[i].score > O.vt_df1 * (float)(seqs[i].last_target_loc - seqs[i].first_target_loc)) {
			seqs[i] = seqs[nb_df];
			nb_df++;
		}
	nb = nb_df;
	if (nb == 0) return 0;
	uint32_t qrstart = qlen_sum, qrend = 0;
	const unsigned filt = (unsigned)((float)seqs[0].score * O.vt_f); // :1377
	for (unsigned i = 0; i < nb; i++) {
		if (seqs[i].score < filt) { nb = i; break; }
		seqs[i].first_query_loc -= (uint32_t)(O.k - 1);
		seqs[i].first_target_loc -= (O.k - 1);
		if ((double)(uint32_t)(seqs[i].last_query_loc - seqs[i].first_query_loc) + 0.5 * (double)O.bw <
		    (double)(int32_t)(seqs[i].last_target_loc - seqs[i].first_target_loc))
			seqs[i].last_target_loc = (int32_t)((double)(uint32_t)(seqs[i].first_target_loc + seqs[i].last_query_loc - seqs[i].first_query_loc) + 0.5 * (double)O.bw);
		if (seqs[i].first_query_loc < qrstart) qrstart = seqs[i].first_query_loc;
		if (seqs[i].last_query_loc > qrend) qrend = seqs[i].last_query_loc;
	}
	// second round on an uncovered read head / tail, :1403-1445
	for (int side = 0; side < 2; ++side) {
		const bool go = side == 0 ? qrstart > cov_thr : qlen_sum - qrend > cov_thr;
		if (!go) continue;
		GdVt v2;
		v2.chrom_id = 0, v2.first_target_loc = v2.last_target_loc = 0, v2.first_query_loc = v2.last_query_loc = 0, v2.score = 0, v2.str = 0;
		const uint32_t mn = side == 0 ? 0 : qrend, mx = side == 0 ? qrstart : qlen_sum;
		gd_vote2(a_for, n_for, 0, &v2, O.vt_dis, tel, mn, mx);
		gd_vote2(a_rev, n_rev, 1, &v2, O.vt_dis, tel, mn, mx);
		v2.first_query_loc -= (uint32_t)(O.k - 1);
		v2.first_target_loc -= (O.k - 1);
		if ((float)v2.score > O.vt_df2 * (float)(v2.last_target_loc - v2.first_target_loc)) {
			if ((double)(uint32_t)(v2.last_query_loc - v2.first_query_loc) + 0.5 * (double)O.bw <
			    (double)(int32_t)(v2.last_target_loc - v2.first_target_loc))
				v2.last_target_loc = (int32_t)((double)(uint32_t)(v2.first_target_loc + v2.last_query_loc - v2.first_query_loc) + 0.5 * (double)O.bw);
			seqs[nb++] = v2;
		}
	}
	return nb;
}

// ---- V2: vote, ShortReads (SR/map.c:447-584): threshold on the hit COUNT of a run, one recovery candidate ----------------------
struct GdSrVoteOpt {
	float min_cnt, rec_threshold_frac, bw_frac; // -n FLOAT1,FLOAT2 ; -r FLOAT,..
	int32_t bw_min, bw_max, af_max_loc;         // -r ..,INT,INT ; --AF_max_loc
	uint32_t max_nb_seeds;                      // frag mode cap of mm_sketch3 (SR/map.c:621-622), UINT32_MAX otherwise
	int32_t frag_mode;
};

// band width = vote distance of one read, SR/map.c:624-631 (float product truncated to unsigned, then clamped with the
// int bounds converted to unsigned)
GDM_HD uint32_t gd_sr_bw(int qlen, const GdSrVoteOpt &O)
{
	uint32_t bw = (uint32_t)(float)((float)qlen * O.bw_frac);
	if ((uint32_t)O.bw_min > bw) bw = (uint32_t)O.bw_min;
	else if ((uint32_t)O.bw_max < bw) bw = (uint32_t)O.bw_max;
	return bw;
}

// In GdVt the ShortReads vt_t (SR/map.c:431-440) keeps target_loc in first_target_loc; last_target_loc is unused (0).
GDM_HD void gd_sr_vt_set(GdVt &v, uint64_t target_loc, uint32_t fq, uint32_t lq, int str, unsigned counter, uint32_t tel)
{
	// (int32)(target_loc & UINT32_MAX) + (str ? extracted_len : -(extracted_len + tmp_extracted_len)) with extracted_len = 0
	// (:502-504; the unsigned negation wraps, the sum is then converted back to int32)
	v.chrom_id = (uint32_t)(target_loc >> 32);
	v.first_target_loc = str ? (int32_t)(uint32_t)target_loc : (int32_t)((uint32_t)target_loc - tel);
	v.last_target_loc = 0, v.first_query_loc = fq, v.last_query_loc = lq, v.str = (uint32_t)str, v.score = counter;
}

GDM_HD void gd_sr_vt_close(GdVt *pot, unsigned &out_len, GdVt &recovery, unsigned counter, uint64_t target_loc, uint32_t fq, uint32_t lq,
                           int str, uint32_t tel, unsigned vt_threshold, unsigned max_n, unsigned vt_rec_threshold)
{
	if (counter > vt_threshold) {
		if (out_len == max_n) {
			if (pot[out_len - 1].score >= counter) return;
		} else out_len++;
		gd_sr_vt_set(pot[out_len - 1], target_loc, fq, lq, str, counter, tel);
		for (unsigned k = out_len - 1; k > 0; k--) {
			if (pot[k].score > pot[k - 1].score) { GdVt t = pot[k]; pot[k] = pot[k - 1]; pot[k - 1] = t; }
			else break;
		}
	} else if (out_len == 0 && counter > vt_rec_threshold && counter > recovery.score)
		gd_sr_vt_set(recovery, target_loc, fq, lq, str, counter, tel);
}

template <class LocSrc>
GDM_HD void gd_vote_sr(LocSrc loc, unsigned len, int str, GdVt *pot, unsigned *nb, uint32_t vt_distance, uint32_t tel, GdVt &recovery,
                       unsigned vt_threshold, unsigned max_n, unsigned vt_rec_threshold)
{
	if (len == 0) return;
	unsigned out_len = *nb, counter = 1;
	uint64_t target_loc = loc[0].target;
	uint32_t fq = loc[0].query, lq = loc[0].query;
	for (unsigned i = 1; i < len; i++) {
		const GdLoc cur = loc[i];
		if (cur.target - target_loc <= vt_distance) {
			counter++;
			if (cur.query < fq) target_loc = cur.target, fq = cur.query;
			if (cur.query > lq) lq = cur.query;
		} else {
			gd_sr_vt_close(pot, out_len, recovery, counter, target_loc, fq, lq, str, tel, vt_threshold, max_n, vt_rec_threshold);
			target_loc = cur.target, fq = lq = cur.query, counter = 1;
		}
	}
	gd_sr_vt_close(pot, out_len, recovery, counter, target_loc, fq, lq, str, tel, vt_threshold, max_n, vt_rec_threshold);
	*nb = out_len;
}

// the candidate list of one short read, SR/map.c:664-699; n_mv = mv.n after mm_seed_mz_flt.  pot[] holds af_max_loc entries.
template <class LocSrc>
GDM_HD unsigned gd_sr_candidates(LocSrc a_for, unsigned n_for, LocSrc a_rev, unsigned n_rev, uint32_t qlen_sum, uint32_t tel,
                                 uint32_t n_mv, const GdSrVoteOpt &O, GdVt *pot)
{
	const bool frag = O.frag_mode && tel < qlen_sum;
	unsigned vt_threshold = frag ? (unsigned)((float)O.max_nb_seeds * O.min_cnt) : (unsigned)((float)n_mv * O.min_cnt);
	const unsigned vt_rec_threshold = frag ? (unsigned)((float)O.max_nb_seeds * O.rec_threshold_frac) : (unsigned)((float)n_mv * O.rec_threshold_frac);
	if (vt_threshold == 0) vt_threshold = 1;
	const uint32_t bw = gd_sr_bw((int)qlen_sum, O);
	GdVt recovery;
	recovery.chrom_id = 0, recovery.first_target_loc = recovery.last_target_loc = 0, recovery.first_query_loc = recovery.last_query_loc = 0;
	recovery.score = 0, recovery.str = 0;
	unsigned nb = 0;
	gd_vote_sr(a_for, n_for, 0, pot, &nb, bw, tel, recovery, vt_threshold, (unsigned)O.af_max_loc, vt_rec_threshold);
	gd_vote_sr(a_rev, n_rev, 1, pot, &nb, bw, tel, recovery, vt_threshold, (unsigned)O.af_max_loc, vt_rec_threshold);
	if (nb == 0) {
		if (recovery.score == 0) return 0;
		pot[0] = recovery;
		nb = 1;
	}
	return nb;
}

// ---- G2: candidate geometry of the ShortReads variant, SR/map.c:779-839, as plain arithmetic for both sides (the host stages call
// it through gd_sr_box_one, map_host.h; on the device map_sr_box_kernel does) --------------------------------------------------------
#ifndef GD_NEG_INF_SCORE
#define GD_NEG_INF_SCORE (-0x40000000)
#endif
// a candidate between the box stage and the post-processing, without its (still empty) record: trivially copyable, so that the
// candidates of a whole batch can live in one reused flat buffer
struct GdCandBox {
	GdVt v;
	int next, concat, valid;
	uint32_t target_id, target_start, target_end, query_start, query_end, qlen, tlen, qseq_off;
	int32_t exact_score;
};
// tlen: length of the contig v.chrom_id (0 when the id is out of range); false: the reference skips the candidate (:796-801)
GDM_HD bool gd_sr_box_core(const GdVt &v, int k_, int a_, uint32_t qlen_sum, int32_t tlen, GdCandBox &b)
{
	const int str = (int)v.str;
	const uint32_t target_id = v.chrom_id;
	uint32_t start_offset, end_offset;
	int32_t target_loc = v.first_target_loc;
	if (str) target_loc -= (k_ - 1);
	int32_t target_start = target_loc, target_end = target_loc;
	if (qlen_sum > 300) {
		if (v.first_query_loc == v.last_query_loc) return false;
		start_offset = v.first_query_loc - (uint32_t)(k_ - 1);
		end_offset = v.last_query_loc;
		if (str) {
			target_end = (int32_t)((uint32_t)target_end - start_offset);
			target_start = (int32_t)((uint32_t)target_start - end_offset);
			if (target_start < 0) {
				end_offset += (uint32_t)target_start;
				target_start = 0;
			}
		} else {
			target_start = (int32_t)((uint32_t)target_start + start_offset);
			target_end = (int32_t)((uint32_t)target_end + end_offset);
			if (target_end + 1 > tlen) {
				end_offset = (uint32_t)(tlen - 1 - target_start) + start_offset;
				target_end = tlen - 1;
			}
		}
	} else {
		if (str) {
			if (target_end > tlen - 1) {
				start_offset = (uint32_t)(target_end - (tlen - 1));
				target_end = tlen - 1;
			} else start_offset = 0;
			if ((uint32_t)target_end < qlen_sum - start_offset - 1) { // int32 against unsigned: compared as unsigned (:816)
				end_offset = start_offset + (uint32_t)target_end;
				target_start = 0;
			} else {
				end_offset = qlen_sum - 1;
				target_start = (int32_t)((uint32_t)target_end - (end_offset - start_offset));
			}
		} else {
			if (target_start < 0) {
				start_offset = (uint32_t)(-target_start);
				target_start = 0;
			} else start_offset = 0;
			if ((uint32_t)(tlen - target_start) < qlen_sum - start_offset) { // (:831) unsigned compare as well
				end_offset = (uint32_t)(tlen - 1 - target_start) + start_offset;
				target_end = tlen - 1;
			} else {
				end_offset = qlen_sum - 1;
				target_end = (int32_t)((uint32_t)target_start + (end_offset - start_offset));
			}
		}
	}
	const uint32_t len = end_offset - start_offset + 1;
	b.v = v, b.next = -1, b.concat = 0, b.valid = 1;
	b.target_id = target_id, b.target_start = (uint32_t)target_start, b.target_end = (uint32_t)target_end;
	b.query_start = start_offset, b.query_end = end_offset, b.qlen = len, b.tlen = len;
	b.qseq_off = str ? qlen_sum - 1 - end_offset : start_offset; // qs = &qs_rev[qlen_sum-1-end_offset] / &qs_for[start_offset]
	b.exact_score = qlen_sum < 300 ? (int32_t)(qlen_sum * (uint32_t)a_) : GD_NEG_INF_SCORE; // :873-908
	return true;
}
