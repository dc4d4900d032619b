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

// ---- flat form -> GdIndex (what gdiet_hip_index_import and the .mmi reader feed) -------------------------------------------------
// keys[i] = minimizer hash (x >> 8), cnt[i] its number of occurrences, pos = the occurrence lists (y values, each ascending)
// concatenated in key order.
static inline void gd_index_from_flat(GdIndex &h, int k, int w, const GdPattern &P, int n_seq, const char *const *names, const uint32_t *lens,
                                      const uint64_t *offsets, const uint32_t *S, uint64_t n_keys, const uint64_t *keys, const uint32_t *cnt,
                                      const uint64_t *pos)
{
	h.k = k, h.w = w, h.pat = P;
	h.seq.resize(n_seq);
	uint64_t sum = 0;
	for (int i = 0; i < n_seq; ++i)
		h.seq[i].name = names && names[i] ? names[i] : "", h.seq[i].len = lens[i], h.seq[i].offset = offsets[i], sum = std::max<uint64_t>(sum, offsets[i] + lens[i]);
	h.S.assign(S, S + (sum + 7) / 8);
	h.n_keys = n_keys;
	h.key_counts.assign(cnt, cnt + n_keys);
	uint64_t tot = 0;
	for (uint64_t i = 0; i < n_keys; ++i) tot += cnt[i];
	h.pos.assign(pos, pos + tot);
	h.tbits = 4;
	while ((1ull << h.tbits) < 2 * n_keys + 16) ++h.tbits;
	h.tkey.assign(1ull << h.tbits, UINT64_MAX), h.tval.assign(1ull << h.tbits, 0);
	const uint32_t mask = (uint32_t)((1ull << h.tbits) - 1);
	uint64_t st = 0;
	for (uint64_t i = 0; i < n_keys; ++i) {
		uint32_t s = gd_idx_slot(keys[i], h.tbits);
		while (h.tkey[s] != UINT64_MAX) s = (s + 1) & mask;
		h.tkey[s] = keys[i], h.tval[s] = st << 32 | cnt[i];
		st += cnt[i];
	}
}

// ---- .mmi files (LR/index.c:480-571: mm_idx_dump / mm_idx_load) -----------------------------------------------------------------
// Layout: "MMI\2", u32 w,k,b,n_seq,flag; per sequence u8 l, name[l], u32 len; per bucket (1<<b of them) i32 n, u64 p[n],
// u32 size, size x {u64 key, u64 val}; then u32 S[(sum_len+7)/8].  key = (minimizer >> b) << 1 | singleton, the bucket number is
// the low b bits of the minimizer; val = the position itself (singleton) or start << 32 | count into p[] (LR/index.c:241-264).
// The file does not store the pattern (LR/index.c:484-486): -Z / -W must be given again, as with the reference.
static inline bool gd_index_read_mmi(GdIndex &h, const char *path, const GdPattern &P, std::string &err)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { err = std::string("cannot open ") + path; return false; }
	auto fail = [&](const char *why) { fclose(fp); err = std::string(path) + ": " + why; return false; };
	char magic[4];
	uint32_t x[5];
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "MMI\2", 4) != 0) return fail("not an MMI\\2 index");
	if (fread(x, 4, 5, fp) != 5) return fail("truncated header");
	const uint32_t w = x[0], k = x[1], b = x[2], n_seq = x[3];
	if (b > 28 || n_seq == 0) return fail("implausible header");
	std::vector<std::string> names(n_seq);
	std::vector<uint32_t> lens(n_seq);
	std::vector<uint64_t> offs(n_seq);
	uint64_t sum_len = 0;
	for (uint32_t i = 0; i < n_seq; ++i) {
		uint8_t l;
		if (fread(&l, 1, 1, fp) != 1) return fail("truncated sequence table");
		names[i].resize(l);
		if (l && fread(&names[i][0], 1, l, fp) != l) return fail("truncated sequence table");
		if (fread(&lens[i], 4, 1, fp) != 1) return fail("truncated sequence table");
		offs[i] = sum_len, sum_len += lens[i];
	}
	std::vector<uint64_t> keys, pos, p, kv;
	std::vector<uint32_t> cnt;
	for (uint64_t bi = 0; bi < (1ull << b); ++bi) {
		int32_t n;
		uint32_t size;
		if (fread(&n, 4, 1, fp) != 1 || n < 0) return fail("truncated bucket");
		p.resize((size_t)n);
		if (n && fread(p.data(), 8, (size_t)n, fp) != (size_t)n) return fail("truncated bucket");
		if (fread(&size, 4, 1, fp) != 1) return fail("truncated bucket");
		kv.resize(2 * (size_t)size);
		if (size && fread(kv.data(), 8, 2 * (size_t)size, fp) != 2 * (size_t)size) return fail("truncated bucket");
		for (uint32_t j = 0; j < size; ++j) {
			const uint64_t key = kv[2 * j], val = kv[2 * j + 1];
			keys.push_back((key >> 1) << b | bi);
			if (key & 1) cnt.push_back(1), pos.push_back(val);
			else {
				const uint64_t st = val >> 32, c = (uint32_t)val;
				if (st + c > (uint64_t)n) return fail("bucket entry out of range");
				cnt.push_back((uint32_t)c);
				pos.insert(pos.end(), p.begin() + st, p.begin() + st + c);
			}
		}
	}
	std::vector<uint32_t> S((sum_len + 7) / 8);
	if (!S.empty() && fread(S.data(), 4, S.size(), fp) != S.size()) return fail("truncated sequence data (an index written with MM_I_NO_SEQ cannot be used)");
	fclose(fp);
	std::vector<const char *> np(n_seq);
	for (uint32_t i = 0; i < n_seq; ++i) np[i] = names[i].c_str();
	gd_index_from_flat(h, (int)k, (int)w, P, (int)n_seq, np.data(), lens.data(), offs.data(), S.data(), keys.size(), keys.data(), cnt.data(), pos.data());
	return true;
}

// khash's slot placement (LR/khash.h:199-330), emulated for the one call sequence the index builder makes: kh_init, kh_resize(n_keys),
// then kh_put of every key of the bucket in ascending order (LR/index.c:216-264, worker_post).  mm_idx_dump writes a bucket's entries
// in SLOT order (LR/index.c:497-516), so a byte-identical .mmi needs the slots, not just the set of keys.  hash = key >> 1 (idx_hash,
// LR/index.c:20), truncated to 32 bits; linear-quadratic probing i += ++step; rehash with the kick-out process of kh_resize.
struct GdKhEmu {
	uint32_t n_buckets = 0, size = 0, n_occupied = 0, upper_bound = 0;
	std::vector<uint8_t> flag; // 2 = empty, 1 = deleted, 0 = live (khash packs these two bits sixteen to a word)
	std::vector<uint64_t> keys;
	std::vector<uint32_t> vals; // index of the entry
	static uint32_t roundup32(uint32_t x) { --x, x |= x >> 1, x |= x >> 2, x |= x >> 4, x |= x >> 8, x |= x >> 16; return ++x; }
	void resize(uint32_t new_n)
	{
		new_n = roundup32(new_n);
		if (new_n < 4) new_n = 4;
		if (size >= (uint32_t)(new_n * 0.77 + 0.5)) return; // requested size is too small
		std::vector<uint8_t> nf(new_n, 2);
		if (n_buckets < new_n) keys.resize(new_n), vals.resize(new_n);
		for (uint32_t j = 0; j != n_buckets; ++j) {
			if (flag[j] != 0) continue;
			uint64_t key = keys[j];
			uint32_t val = vals[j];
			const uint32_t new_mask = new_n - 1;
			flag[j] = 1;
			for (;;) { // kick-out process
				uint32_t step = 0, i = (uint32_t)(key >> 1) & new_mask;
				while (nf[i] != 2) i = (i + (++step)) & new_mask;
				nf[i] = 0;
				if (i < n_buckets && flag[i] == 0) { // kick out the existing element
					std::swap(keys[i], key), std::swap(vals[i], val);
					flag[i] = 1;
				} else {
					keys[i] = key, vals[i] = val;
					break;
				}
			}
		}
		if (n_buckets > new_n) keys.resize(new_n), vals.resize(new_n);
		flag.swap(nf);
		n_buckets = new_n, n_occupied = size, upper_bound = (uint32_t)(n_buckets * 0.77 + 0.5);
	}
	void put(uint64_t key, uint32_t val) // keys are distinct (asserted by the reference: `absent`)
	{
		if (n_occupied >= upper_bound) {
			if (n_buckets > (size << 1)) resize(n_buckets - 1);
			else resize(n_buckets + 1);
		}
		const uint32_t mask = n_buckets - 1;
		uint32_t step = 0, i = (uint32_t)(key >> 1) & mask, x = n_buckets, site = n_buckets;
		if (flag[i] == 2) x = i;
		else {
			const uint32_t last = i;
			while (flag[i] != 2 && (flag[i] == 1 || keys[i] >> 1 != key >> 1)) {
				if (flag[i] == 1) site = i;
				i = (i + (++step)) & mask;
				if (i == last) { x = site; break; }
			}
			if (x == n_buckets) x = (flag[i] == 2 && site != n_buckets) ? site : i;
		}
		if (flag[x] == 2) keys[x] = key, vals[x] = val, flag[x] = 0, ++size, ++n_occupied;
		else if (flag[x] == 1) keys[x] = key, vals[x] = val, flag[x] = 0, ++size;
	}
};

// mm_idx_dump (LR/index.c:480-517), byte for byte: per bucket the position array p[] in the order worker_post fills it (keys
// ascending, each list ascending), then the hash table's live slots in slot order
static inline bool gd_index_write_mmi(const GdIndex &h, const char *path, int bucket_bits, std::string &err)
{
	FILE *fp = fopen(path, "wb");
	if (!fp) { err = std::string("cannot create ") + path; return false; }
	const uint32_t b = (uint32_t)bucket_bits;
	uint32_t x[5] = {(uint32_t)h.w, (uint32_t)h.k, b, (uint32_t)h.seq.size(), 0};
	fwrite("MMI\2", 1, 4, fp), fwrite(x, 4, 5, fp);
	uint64_t sum_len = 0;
	for (const GdSeqInfo &s : h.seq) {
		const uint8_t l = (uint8_t)std::min<size_t>(s.name.size(), 255);
		fwrite(&l, 1, 1, fp), fwrite(s.name.data(), 1, l, fp), fwrite(&s.len, 4, 1, fp);
		sum_len += s.len;
	}
	std::vector<std::vector<uint32_t>> slots(1u << b); // table slots by bucket (low b bits of the minimizer)
	for (size_t sl = 0; sl < h.tkey.size(); ++sl)
		if (h.tkey[sl] != UINT64_MAX) slots[h.tkey[sl] & ((1u << b) - 1)].push_back((uint32_t)sl);
	std::vector<uint64_t> p, kv, ekey, eval;
	for (uint32_t bi = 0; bi < (1u << b); ++bi) {
		std::vector<uint32_t> &sl_of = slots[bi];
		std::sort(sl_of.begin(), sl_of.end(), [&](uint32_t a, uint32_t c) { return h.tkey[a] < h.tkey[c]; }); // radix_sort_128x: ascending minimizer
		p.clear(), kv.clear(), ekey.clear(), eval.clear();
		GdKhEmu kh;
		if (!sl_of.empty()) kh.resize((uint32_t)sl_of.size());
		for (uint32_t sl : sl_of) {
			const uint64_t st = h.tval[sl] >> 32, c = (uint32_t)h.tval[sl], key = h.tkey[sl] >> b << 1;
			kh.put(key, (uint32_t)ekey.size());
			if (c == 1) ekey.push_back(key | 1), eval.push_back(h.pos[st]);
			else {
				ekey.push_back(key), eval.push_back((uint64_t)p.size() << 32 | c);
				p.insert(p.end(), h.pos.begin() + st, h.pos.begin() + st + c);
			}
		}
		for (uint32_t k = 0; k < kh.n_buckets; ++k)
			if (kh.flag[k] == 0) kv.push_back(ekey[kh.vals[k]]), kv.push_back(eval[kh.vals[k]]);
		const int32_t n = (int32_t)p.size();
		const uint32_t size = (uint32_t)(kv.size() / 2);
		fwrite(&n, 4, 1, fp), fwrite(p.data(), 8, p.size(), fp), fwrite(&size, 4, 1, fp), fwrite(kv.data(), 8, kv.size(), fp);
	}
	fwrite(h.S.data(), 4, (sum_len + 7) / 8, fp);
	const bool ok = fflush(fp) == 0 && !ferror(fp);
	fclose(fp);
	if (!ok) err = std::string("write error on ") + path;
	return ok;
}
