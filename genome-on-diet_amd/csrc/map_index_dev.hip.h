// Index build on the device: mm_sketch over the reference (LR/sketch.c:156/1577 via the shared automaton
// of map_stages.h, AVX-512 flush rule = GDiet_avx), sort of the (minimizer, position) pairs, run-length encoding into distinct keys
// and the open-addressing table -- the same flat index gd_index_build (map_index.h) makes on the host, built where it will live.
//   host    : contig table, 4-bit packing of S (the host post-processing needs S anyway)
//   device  : unpack S -> nt4 bytes; one thread per slice of 4096 sparsified bases runs the winnowing automaton twice (count,
//             then emit: exact slices, see gd_sketch_slice); two stable radix sorts (by y, then by hash: hipCUB); run-length
//             encode; one thread per distinct key claims its table slot with a 64-bit compare-and-swap.
// The host copies of the table (for export / .mmi dump) are fetched lazily; the occurrence counts stay on the device, sorted, for
// mm_idx_cal_max_occ.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "map_index.h"

#define GD_IDX_SLICE 4096u

struct GdIdxSlice { uint64_t seq_off; uint32_t rid, diet_len, i_emit, i_end; };

__global__ __launch_bounds__(256) void idx_unpack_kernel(const uint32_t *__restrict__ S, uint64_t *__restrict__ nt4x8, uint64_t n_words)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_words) return;
	const uint32_t wd = S[i];
	uint64_t out = 0; // 8 nibbles -> 8 bytes, base o of the word in byte o
	for (int b = 0; b < 8; ++b) out |= (uint64_t)((wd >> (4 * b)) & 0xf) << (8 * b);
	nt4x8[i] = out;
}

struct GdEmitCountOnly {
	unsigned n;
	__device__ bool operator()(const GdMini &) { ++n; return false; }
};
struct GdEmitStore {
	uint64_t *hash, *y;
	unsigned n;
	__device__ bool operator()(const GdMini &m) { hash[n] = m.x >> 8, y[n] = m.y, ++n; return false; }
};

// pass 0: count the minimizers of every slice; pass 1: write them at the slice's offset
__global__ __launch_bounds__(64) void idx_sketch_kernel(int pass, uint32_t n_slices, const GdIdxSlice *__restrict__ sl, const uint8_t *__restrict__ nt4,
                                                        int w, int k, GdPattern P, uint32_t *__restrict__ counts, const uint64_t *__restrict__ offs,
                                                        uint64_t *__restrict__ hash, uint64_t *__restrict__ y)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_slices) return;
	const GdIdxSlice s = sl[i];
	const uint8_t *str = nt4 + s.seq_off;
	if (pass == 0) {
		GdEmitCountOnly e = {0};
		gd_sketch_slice(str, s.diet_len, s.i_emit, s.i_end, w, k, s.rid, 0, P, true, e);
		counts[i] = e.n;
	} else {
		GdEmitStore e = {hash + offs[i], y + offs[i], 0};
		gd_sketch_slice(str, s.diet_len, s.i_emit, s.i_end, w, k, s.rid, 0, P, true, e);
	}
}

__global__ __launch_bounds__(256) void idx_insert_kernel(uint64_t n_keys, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ cnt,
                                                         const uint64_t *__restrict__ start, uint64_t *__restrict__ tkey, uint64_t *__restrict__ tval,
                                                         uint32_t tbits)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_keys) return;
	const uint32_t mask = (uint32_t)((1ull << tbits) - 1);
	uint32_t s = gd_idx_slot(keys[i], tbits);
	for (;;) {
		const unsigned long long old = atomicCAS((unsigned long long *)&tkey[s], (unsigned long long)UINT64_MAX, (unsigned long long)keys[i]);
		if (old == (unsigned long long)UINT64_MAX) break;
		s = (s + 1) & mask;
	}
	tval[s] = start[i] << 32 | cnt[i];
}

__global__ __launch_bounds__(256) void idx_widen_kernel(uint64_t n, const uint32_t *__restrict__ in, uint64_t *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = in[i];
}

struct GdDevTmp { // frees everything it allocated when it goes out of scope
	std::vector<void *> v;
	template <class T> hipError_t alloc(T **p, size_t n) { hipError_t e = hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)); if (e == hipSuccess) v.push_back(*p); return e; }
	void keep(void *p) { v.erase(std::remove(v.begin(), v.end(), p), v.end()); }
	~GdDevTmp() { for (void *p : v) (void)hipFree(p); }
};

#define GD_IDX_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e__); return false; } } while (0)

// Sequence table + 4-bit packing of S on host threads (the part of gd_index_build that is not sketching)
static inline void gd_index_pack_S(GdIndex &I, const std::vector<std::string> &names, const std::vector<GdSeqSpan> &seqs, int n_threads)
{
	const size_t n = seqs.size();
	uint64_t sum = 0;
	I.seq.resize(n);
	for (size_t i = 0; i < n; ++i) I.seq[i].name = names[i], I.seq[i].offset = sum, I.seq[i].len = (uint32_t)seqs[i].size(), sum += seqs[i].size();
	I.S.assign((sum + 7) / 8, 0);
	// word-aligned chunks of 1 Mbases: a chunk never shares a word with another (contig boundaries inside a word are handled
	// by walking the contigs that overlap the chunk)
	const uint64_t CH = 1u << 20;
	const uint64_t n_chunks = (sum + CH - 1) / CH;
	std::atomic<uint64_t> next(0);
	auto work = [&]() {
		for (;;) {
			const uint64_t c = next.fetch_add(1);
			if (c >= n_chunks) break;
			const uint64_t lo = c * CH, hi = std::min(sum, lo + CH);
			// first contig overlapping lo
			size_t ci = (size_t)(std::upper_bound(I.seq.begin(), I.seq.end(), lo, [](uint64_t v, const GdSeqInfo &s) { return v < s.offset; }) - I.seq.begin());
			ci = ci ? ci - 1 : 0;
			for (; ci < n && I.seq[ci].offset < hi; ++ci) {
				const uint64_t off = I.seq[ci].offset, a = std::max(lo, off), b = std::min<uint64_t>(hi, off + I.seq[ci].len);
				for (uint64_t o = a; o < b; ++o) I.S[o >> 3] |= (uint32_t)gd_nt4((unsigned char)seqs[ci][o - off]) << ((o & 7) << 2);
			}
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < std::max(1, n_threads); ++t) th.emplace_back(work);
	for (auto &t : th) t.join();
}

// Builds ix->h's sequence part on the host and the table part on the device (d_tkey, d_tval, d_pos, d_S, d_cnt_sorted).
static bool gd_index_build_device(gdiet_index *ix, const std::vector<std::string> &names, const std::vector<GdSeqSpan> &seqs, int k, int w,
                                  const GdPattern &pat, int n_threads, hipStream_t st, std::string &err)
{
	GdIndex &I = ix->h;
	I.k = k, I.w = w, I.pat = pat;
	gd_index_pack_S(I, names, seqs, n_threads);
	const uint64_t n_words = I.S.size();
	GdDevTmp T;
	uint32_t *d_S = nullptr;
	GD_IDX_HIP(T.alloc(&d_S, n_words + 2));
	GD_IDX_HIP(hipMemcpyAsync(d_S, I.S.data(), n_words * 4, hipMemcpyHostToDevice, st));
	uint64_t *d_nt4 = nullptr;
	GD_IDX_HIP(T.alloc(&d_nt4, n_words + 8));
	if (n_words) hipLaunchKernelGGL(idx_unpack_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, d_S, d_nt4, n_words);
	// slices
	std::vector<GdIdxSlice> sl;
	for (size_t c = 0; c < seqs.size(); ++c) {
		if (seqs[c].empty()) continue;
		const unsigned dl = gd_diet_len(pat, (unsigned)seqs[c].size(), 0);
		for (unsigned a = 0; a < dl || a == 0; a += GD_IDX_SLICE) { // a contig with dl == 0 still gets its (empty) final flush
			GdIdxSlice s = {I.seq[c].offset, (uint32_t)c, dl, a, std::min(dl, a + GD_IDX_SLICE)};
			sl.push_back(s);
			if (dl == 0) break;
		}
	}
	const uint32_t ns = (uint32_t)sl.size();
	GdIdxSlice *d_sl = nullptr;
	uint32_t *d_cnt = nullptr;
	uint64_t *d_cnt64 = nullptr, *d_off = nullptr;
	GD_IDX_HIP(T.alloc(&d_sl, ns));
	GD_IDX_HIP(T.alloc(&d_cnt, ns));
	GD_IDX_HIP(T.alloc(&d_cnt64, ns));
	GD_IDX_HIP(T.alloc(&d_off, (size_t)ns + 1));
	GD_IDX_HIP(hipMemcpyAsync(d_sl, sl.data(), sizeof(GdIdxSlice) * ns, hipMemcpyHostToDevice, st));
	uint64_t total = 0;
	void *d_tmp = nullptr;
	size_t tmp_bytes = 0;
	if (ns) {
		hipLaunchKernelGGL(idx_sketch_kernel, dim3((ns + 63) / 64), dim3(64), 0, st, 0, ns, d_sl, (const uint8_t *)d_nt4, w, k, pat, d_cnt, (const uint64_t *)nullptr,
		                   (uint64_t *)nullptr, (uint64_t *)nullptr);
		hipLaunchKernelGGL(idx_widen_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, (uint64_t)ns, d_cnt, d_cnt64);
		GD_IDX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt64, d_off, (int)ns, st));
		GD_IDX_HIP(T.alloc((uint8_t **)&d_tmp, tmp_bytes));
		GD_IDX_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cnt64, d_off, (int)ns, st));
		uint64_t last_off = 0;
		uint32_t last_cnt = 0;
		GD_IDX_HIP(hipMemcpyAsync(&last_off, d_off + (ns - 1), 8, hipMemcpyDeviceToHost, st));
		GD_IDX_HIP(hipMemcpyAsync(&last_cnt, d_cnt + (ns - 1), 4, hipMemcpyDeviceToHost, st));
		GD_IDX_HIP(hipStreamSynchronize(st));
		total = last_off + last_cnt;
	}
	if (total >= (1ull << 31)) { err = "more than 2^31 minimizers: split the reference (the device sort works on 32-bit counts)"; return false; }
	// minimizers: (hash, y) in slice order, then sorted by (hash, y)
	uint64_t *d_h0 = nullptr, *d_y0 = nullptr, *d_h1 = nullptr, *d_y1 = nullptr;
	GD_IDX_HIP(T.alloc(&d_h0, total));
	GD_IDX_HIP(T.alloc(&d_y0, total));
	GD_IDX_HIP(T.alloc(&d_h1, total));
	GD_IDX_HIP(T.alloc(&d_y1, total));
	uint64_t n_keys = 0;
	uint64_t *d_keys = nullptr, *d_start = nullptr;
	uint32_t *d_kcnt = nullptr;
	if (total) {
		hipLaunchKernelGGL(idx_sketch_kernel, dim3((ns + 63) / 64), dim3(64), 0, st, 1, ns, d_sl, (const uint8_t *)d_nt4, w, k, pat, d_cnt, (const uint64_t *)d_off, d_h0, d_y0);
		int rid_bits = 1;
		while ((1ull << rid_bits) < seqs.size()) ++rid_bits;
		size_t b1 = 0, b2 = 0;
		GD_IDX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, d_y0, d_y1, d_h0, d_h1, (int)total, 0, 32 + rid_bits, st));
		GD_IDX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b2, d_h1, d_h0, d_y1, d_y0, (int)total, 0, 2 * k, st));
		void *d_t2 = nullptr;
		GD_IDX_HIP(T.alloc((uint8_t **)&d_t2, std::max(b1, b2)));
		GD_IDX_HIP(hipcub::DeviceRadixSort::SortPairs(d_t2, b1, d_y0, d_y1, d_h0, d_h1, (int)total, 0, 32 + rid_bits, st)); // by y (stable)
		GD_IDX_HIP(hipcub::DeviceRadixSort::SortPairs(d_t2, b2, d_h1, d_h0, d_y1, d_y0, (int)total, 0, 2 * k, st));        // then by hash (stable)
		// distinct keys, their counts and starts
		GD_IDX_HIP(T.alloc(&d_keys, total));
		GD_IDX_HIP(T.alloc(&d_kcnt, total));
		uint64_t *d_nruns = nullptr;
		GD_IDX_HIP(T.alloc(&d_nruns, 1));
		size_t b3 = 0;
		GD_IDX_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, b3, d_h0, d_keys, d_kcnt, d_nruns, (int)total, st));
		void *d_t3 = nullptr;
		GD_IDX_HIP(T.alloc((uint8_t **)&d_t3, b3));
		GD_IDX_HIP(hipcub::DeviceRunLengthEncode::Encode(d_t3, b3, d_h0, d_keys, d_kcnt, d_nruns, (int)total, st));
		GD_IDX_HIP(hipMemcpyAsync(&n_keys, d_nruns, 8, hipMemcpyDeviceToHost, st));
		GD_IDX_HIP(hipStreamSynchronize(st));
		GD_IDX_HIP(T.alloc(&d_start, n_keys));
		uint64_t *d_kcnt64 = d_h1; // reuse: the first sort's output is dead
		hipLaunchKernelGGL(idx_widen_kernel, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, st, n_keys, d_kcnt, d_kcnt64);
		size_t b4 = 0;
		GD_IDX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b4, d_kcnt64, d_start, (int)n_keys, st));
		void *d_t4 = nullptr;
		GD_IDX_HIP(T.alloc((uint8_t **)&d_t4, b4));
		GD_IDX_HIP(hipcub::DeviceScan::ExclusiveSum(d_t4, b4, d_kcnt64, d_start, (int)n_keys, st));
	}
	// table
	I.n_keys = n_keys;
	I.tbits = 4;
	while ((1ull << I.tbits) < 2 * n_keys + 16) ++I.tbits;
	const uint64_t tsz = 1ull << I.tbits;
	uint64_t *d_tkey = nullptr, *d_tval = nullptr;
	uint32_t *d_csort = nullptr;
	GD_IDX_HIP(T.alloc(&d_tkey, tsz));
	GD_IDX_HIP(T.alloc(&d_tval, tsz));
	GD_IDX_HIP(hipMemsetAsync(d_tkey, 0xff, tsz * 8, st));
	GD_IDX_HIP(hipMemsetAsync(d_tval, 0, tsz * 8, st));
	if (n_keys) {
		hipLaunchKernelGGL(idx_insert_kernel, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, st, n_keys, d_keys, d_kcnt, d_start, d_tkey, d_tval, (uint32_t)I.tbits);
		// occurrence counts, ascending, for mm_idx_cal_max_occ
		GD_IDX_HIP(T.alloc(&d_csort, n_keys));
		size_t b5 = 0;
		GD_IDX_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, b5, d_kcnt, d_csort, (int)n_keys, 0, 32, st));
		void *d_t5 = nullptr;
		GD_IDX_HIP(T.alloc((uint8_t **)&d_t5, b5));
		GD_IDX_HIP(hipcub::DeviceRadixSort::SortKeys(d_t5, b5, d_kcnt, d_csort, (int)n_keys, 0, 32, st));
	}
	GD_IDX_HIP(hipStreamSynchronize(st));
	GD_IDX_HIP(hipGetLastError());
	// hand the results over
	ix->d_S = d_S, T.keep(d_S);
	ix->d_tkey = d_tkey, T.keep(d_tkey);
	ix->d_tval = d_tval, T.keep(d_tval);
	if (total) ix->d_pos = d_y0, T.keep(d_y0);
	else { uint64_t *d = nullptr; GD_IDX_HIP(hipMalloc((void **)&d, 8)); ix->d_pos = d; }
	ix->d_cnt_sorted = d_csort, ix->n_pos = total;
	if (d_csort) T.keep(d_csort);
	ix->dview.k = k, ix->dview.w = w, ix->dview.tbits = I.tbits;
	ix->dview.tkey = (const uint64_t *)ix->d_tkey, ix->dview.tval = (const uint64_t *)ix->d_tval, ix->dview.pos = (const uint64_t *)ix->d_pos;
	ix->host_tables = false;
	return true;
}

// the host copies of a device-built table, fetched when someone needs them (export, .mmi dump)
static bool gd_index_fetch_host(gdiet_index *ix, std::string &err)
{
	if (ix->host_tables) return true;
	GdIndex &I = ix->h;
	const uint64_t tsz = 1ull << I.tbits;
	I.tkey.resize(tsz), I.tval.resize(tsz), I.pos.resize(ix->n_pos);
	GD_IDX_HIP(hipMemcpy(I.tkey.data(), ix->d_tkey, tsz * 8, hipMemcpyDeviceToHost));
	GD_IDX_HIP(hipMemcpy(I.tval.data(), ix->d_tval, tsz * 8, hipMemcpyDeviceToHost));
	if (ix->n_pos) GD_IDX_HIP(hipMemcpy(I.pos.data(), ix->d_pos, ix->n_pos * 8, hipMemcpyDeviceToHost));
	I.key_counts.clear();
	for (uint64_t s = 0; s < tsz; ++s)
		if (I.tkey[s] != UINT64_MAX) I.key_counts.push_back((uint32_t)I.tval[s]);
	ix->host_tables = true;
	return true;
}
