// Host-side index of the sparsified reference: the flat, GPU-friendly mirror of mm_idx_t (T2 of SURVEY.md 8a).
//   build:  mm_sketch over every contig (LR/sketch.c:1577, AVX-512 flush rule :607 = the parity target GDiet_avx)
//           -> sort by (hash, y) -> one open-addressing table hash -> run of positions   (LR/index.c:216-270)
//   import: the same flat arrays can be filled from a reference-built mm_idx_t by walking its buckets
//           (INTEGRATION.md shows that stub); the GPU side only ever sees the flat form.
// The 4-bit packed sequence array S and the (name,len,offset) table are kept exactly as mm_idx_t has them.
#pragma once
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "map_host.h"
#include "map_stages.h"

struct GdIndex {
	int k = 0, w = 0;
	GdPattern pat;
	std::vector<GdSeqInfo> seq;
	std::vector<uint32_t> S;       // 4-bit packed bases, 8 per word (LR/mmpriv.h:31)
	uint32_t tbits = 0;
	std::vector<uint64_t> tkey, tval, pos;
	uint64_t n_keys = 0;
	std::vector<uint32_t> key_counts; // occurrences per distinct key (for mm_idx_cal_max_occ)

	GdIdxView view() const { GdIdxView v; v.k = k, v.w = w, v.tbits = tbits, v.tkey = tkey.data(), v.tval = tval.data(), v.pos = pos.data(); return v; }
	GdRefView ref() const { GdRefView r; r.S = S.data(), r.seq = seq.data(), r.n_seq = (uint32_t)seq.size(); return r; }
};

struct GdEmitVec {
	std::vector<GdMini> *v;
	bool operator()(const GdMini &m) { v->push_back(m); return false; }
};

static inline uint8_t gd_nt4(unsigned char c) // seq_nt4_table, LR/sketch.c:11-18
{
	switch (c) {
	case 0: case 'A': case 'a': return 0;
	case 1: case 'C': case 'c': return 1;
	case 2: case 'G': case 'g': return 2;
	case 3: case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

// mm_idx_cal_max_occ (LR/index.c:190-210): the (1-f) quantile of the per-key occurrence counts, + 1
static inline int32_t gd_index_cal_max_occ(const GdIndex &I, float f)
{
	if (f <= 0.) return INT32_MAX;
	std::vector<uint32_t> a(I.key_counts);
	const size_t n = a.size();
	if (n == 0) return 1;
	size_t kk = (size_t)(uint32_t)((1. - f) * n);
	if (kk >= n) kk = n - 1;
	std::nth_element(a.begin(), a.begin() + kk, a.end());
	return (int32_t)(a[kk] + 1);
}

// names[i], seqs[i] (ASCII).  final_ge = true reproduces GDiet_avx (the parity target); false the scalar GDiet build.
struct GdSeqSpan { const char *p; size_t n; size_t size() const { return n; } bool empty() const { return n == 0; } char operator[](size_t i) const { return p[i]; } };

static inline void gd_index_build(GdIndex &I, const std::vector<std::string> &names, const std::vector<GdSeqSpan> &seqs, int k, int w,
                                  const GdPattern &pat, int n_threads, bool final_ge = true)
{
	I.k = k, I.w = w, I.pat = pat;
	const size_t n = seqs.size();
	uint64_t sum = 0;
	I.seq.resize(n);
	for (size_t i = 0; i < n; ++i) I.seq[i].name = names[i], I.seq[i].offset = sum, I.seq[i].len = (uint32_t)seqs[i].size(), sum += seqs[i].size();
	I.S.assign((sum + 7) / 8, 0);
	std::vector<std::vector<GdMini>> mins(n);
	std::atomic<size_t> next(0);
	auto work = [&]() {
		for (;;) {
			const size_t i = next.fetch_add(1);
			if (i >= n) break;
			const GdSeqSpan &s = seqs[i];
			std::vector<uint8_t> enc(s.size());
			for (size_t j = 0; j < s.size(); ++j) enc[j] = gd_nt4((unsigned char)s[j]);
			// pack into S: whole words are plain stores; the first/last word of a contig may be shared with its neighbours
			const uint64_t off = I.seq[i].offset, end = off + s.size();
			for (uint64_t wd = off >> 3; wd <= (end ? (end - 1) >> 3 : 0) && !s.empty(); ++wd) {
				uint32_t val = 0;
				const uint64_t lo = std::max<uint64_t>(wd << 3, off), hi = std::min<uint64_t>((wd << 3) + 8, end);
				for (uint64_t o = lo; o < hi; ++o) val |= (uint32_t)enc[o - off] << ((o & 7) << 2);
				if (lo == (wd << 3) && hi == (wd << 3) + 8) I.S[wd] = val;
				else __atomic_fetch_or(&I.S[wd], val, __ATOMIC_RELAXED);
			}
			if (!s.empty()) {
				GdEmitVec e = {&mins[i]};
				const unsigned dl = gd_diet_len(pat, (unsigned)s.size(), 0);
				gd_sketch_core(enc.data(), dl, w, k, (uint32_t)i, 0, pat, final_ge, e);
			}
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < std::max(1, n_threads); ++t) th.emplace_back(work);
	for (auto &t : th) t.join();
	// all minimizers, sorted by (hash, y): bucket by the low bits first so that the sort parallelises
	size_t total = 0;
	for (auto &v : mins) total += v.size();
	const int B = 10;
	std::vector<size_t> cnt((1u << B) + 1, 0);
	for (auto &v : mins) for (auto &m : v) ++cnt[((m.x >> 8) & ((1u << B) - 1)) + 1];
	for (size_t b = 0; b < (1u << B); ++b) cnt[b + 1] += cnt[b];
	std::vector<GdMini> all(total);
	{
		std::vector<size_t> fill(cnt.begin(), cnt.end() - 1);
		for (auto &v : mins) { for (auto &m : v) all[fill[(m.x >> 8) & ((1u << B) - 1)]++] = m; std::vector<GdMini>().swap(v); }
	}
	next = 0;
	auto sortw = [&]() {
		for (;;) {
			const size_t b = next.fetch_add(1);
			if (b >= (1u << B)) break;
			std::sort(all.begin() + cnt[b], all.begin() + cnt[b + 1], [](const GdMini &a, const GdMini &c) { return a.x != c.x ? a.x < c.x : a.y < c.y; });
		}
	};
	th.clear();
	for (int t = 0; t < std::max(1, n_threads); ++t) th.emplace_back(sortw);
	for (auto &t : th) t.join();
	// distinct keys
	I.pos.resize(total);
	I.key_counts.clear();
	std::vector<std::pair<uint64_t, uint64_t>> keys; // (hash, start<<32|n)
	for (size_t b = 0; b < (1u << B); ++b) {
		size_t st = cnt[b];
		for (size_t j = cnt[b]; j <= cnt[b + 1]; ++j) {
			if (j == cnt[b + 1] || (all[j].x >> 8) != (all[st].x >> 8)) {
				if (j > st) {
					keys.emplace_back(all[st].x >> 8, (uint64_t)st << 32 | (uint64_t)(j - st));
					I.key_counts.push_back((uint32_t)(j - st));
				}
				st = j;
			}
		}
	}
	for (size_t j = 0; j < total; ++j) I.pos[j] = all[j].y;
	std::vector<GdMini>().swap(all);
	I.n_keys = keys.size();
	I.tbits = 4;
	while ((1ull << I.tbits) < 2 * keys.size() + 16) ++I.tbits;
	I.tkey.assign(1ull << I.tbits, UINT64_MAX);
	I.tval.assign(1ull << I.tbits, 0);
	const uint32_t mask = (uint32_t)((1ull << I.tbits) - 1);
	// parallel insertion: a slot is claimed with a compare-and-swap on its key, then its value is written by the owner
	next = 0;
	const size_t nk = keys.size();
	auto insw = [&]() {
		for (;;) {
			const size_t b0 = next.fetch_add(65536);
			if (b0 >= nk) break;
			for (size_t j = b0; j < std::min(nk, b0 + 65536); ++j) {
				uint32_t s = gd_idx_slot(keys[j].first, I.tbits);
				for (;;) {
					uint64_t expect = UINT64_MAX;
					if (__atomic_compare_exchange_n(&I.tkey[s], &expect, keys[j].first, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
					s = (s + 1) & mask;
				}
				I.tval[s] = keys[j].second;
			}
		}
	};
	th.clear();
	for (int t = 0; t < std::max(1, n_threads); ++t) th.emplace_back(insw);
	for (auto &t : th) t.join();
}
