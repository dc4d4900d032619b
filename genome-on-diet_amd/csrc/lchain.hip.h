// SURVEY 8f rank 4 (chaining half): mg_lchain_dp (SR/lchain.c:124-190, SR/mmpriv.h:102) -- minimap2's anchor chaining, which
// GDiet keeps in its tree but never calls (its voting replaces it).  The O(n x max_iter) fill of f[] / p[] / v[] runs on the
// device, one wavefront per read: an anchor's predecessors are scored 64 at a time (comput_sc :91-122 is a pure function of the
// two anchors; its float penalty terms are evaluated without contraction, as the reference's build has no FMA), and the
// sequential part of the scan -- the running maximum, the max_skip counter and its early exit (:150-163) -- is replayed in
// order on the scalar unit over only those lanes that can change it.  Why that is exact:
//   * a predecessor j with sc + f[j] <= the running maximum at the start of the chunk and t[j] != i changes nothing;
//   * the marks t[p[j]] = i (:162) of a chunk can all be written before its lanes are tested: a mark only ever targets an index
//     below its own j (p[j] < j), i.e. a lane tested later, and a mark made by a lane the sequential loop would not have reached
//     (after the break) targets only indices the loop does not reach either; marks of earlier anchors never equal the current i.
// The chains themselves (mg_chain_backtrack :9-53, compact_a :55-89: pointer chasing and two unstable in-place radix sorts whose
// exact permutation shows in the output) are built on host threads from f / p / v: lchain_host.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lchain_core.h" // GdChainOpt, gdl_comput_sc (shared with the host-side test driver)

// a: anchors (x, y) of all reads, read i at aoff[i] .. aoff[i+1]; f / p / v / t: one int32 per anchor (p: index inside the read or -1).
// max_dist_x / max_dist_y in O are already raised to bw where the reference does so (:136-137).
__global__ __launch_bounds__(64) void lchain_fill_kernel(int n_reads, const uint64_t *__restrict__ a, const int64_t *__restrict__ aoff, GdChainOpt O,
                                                         int32_t *__restrict__ f_, int32_t *__restrict__ p_, int32_t *__restrict__ v_, int32_t *__restrict__ t_)
{
	const int rd = blockIdx.x, lane = threadIdx.x;
	if (rd >= n_reads) return;
	const int64_t base = aoff[rd];
	const int n = (int)(aoff[rd + 1] - base);
	const uint64_t *A = a + 2 * base;
	int32_t *f = f_ + base, *p = p_ + base, *v = v_ + base, *t = t_ + base;
	for (int i = lane; i < n; i += 64) t[i] = 0;
	__threadfence_block();
	__syncthreads();
	int st = 0, max_ii = -1;
	for (int i = 0; i < n; ++i) {
		const uint64_t aix = A[2 * i], aiy = A[2 * i + 1];
		int32_t max_f = (int32_t)(aiy >> 32 & 0xff), n_skip = 0;
		int max_j = -1;
		while (st < i && (aix >> 32 != A[2 * st] >> 32 || aix > A[2 * st] + (uint64_t)(int64_t)O.max_dist_x)) ++st; // :146
		if (i - st > O.max_iter) st = i - O.max_iter;
		int end_j = st - 1;
		bool broke = false;
		for (int jtop = i - 1; jtop >= st && !broke; jtop -= 64) {
			const int j = jtop - lane;
			bool valid = false, tj = false;
			int32_t cand = INT32_MIN;
			if (j >= st) {
				const int32_t sc = gdl_comput_sc(aix, aiy, A[2 * j], A[2 * j + 1], O);
				if (sc != INT32_MIN) {
					valid = true, cand = sc + f[j];
					const int32_t pj = p[j];
					if (pj >= 0) t[pj] = i; // :162 (see the header comment for why it may come first)
				}
			}
			__threadfence_block();
			__syncthreads();
			if (valid) tj = t[j] == i;
			const int32_t max_f0 = max_f;
			uint64_t m = __ballot(valid && (cand > max_f0 || tj));
			while (m) {
				const int l = __builtin_ctzll(m);
				m &= m - 1;
				const int32_t c = __shfl(cand, l, 64);
				if (c > max_f) {
					max_f = c, max_j = jtop - l;
					if (n_skip > 0) --n_skip;
				} else if (__shfl((int)tj, l, 64)) {
					if (++n_skip > O.max_skip) { broke = true, end_j = jtop - l; break; }
				}
			}
			__syncthreads();
		}
		// (the distance test is UNSIGNED, as in the reference, whose uint64 operand wins the conversion: where the anchors pass from one
		// target or strand to the next -- minimap2 keeps the strand in bit 63 -- the difference wraps to >= 2^63 and forces the rescan)
		if (max_ii < 0 || (uint64_t)(aix - A[2 * max_ii]) > (uint64_t)(int64_t)O.max_dist_x) { // :165-170: the best f in the window, the largest j among equals
			int32_t bf = INT32_MIN;
			int bj = -1;
			for (int j = i - 1 - lane; j >= st; j -= 64) {
				const int32_t fj = f[j];
				if (bf < fj) bf = fj, bj = j; // (descending j per lane: the first maximum is the largest j)
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const int32_t of = __shfl_xor(bf, o, 64);
				const int oj = __shfl_xor(bj, o, 64);
				if (of > bf || (of == bf && oj > bj)) bf = of, bj = oj;
			}
			max_ii = bj;
		}
		if (max_ii >= 0 && max_ii < end_j) { // :171-176
			const int32_t tmp = gdl_comput_sc(aix, aiy, A[2 * max_ii], A[2 * max_ii + 1], O);
			if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) max_f = tmp + f[max_ii], max_j = max_ii;
		}
		const int32_t vi = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
		if (lane == 0) f[i] = max_f, p[i] = max_j, v[i] = vi;
		if (max_ii < 0 || ((uint64_t)(aix - A[2 * max_ii]) <= (uint64_t)(int64_t)O.max_dist_x && f[max_ii] < max_f)) max_ii = i;
		__threadfence_block();
		__syncthreads();
	}
}
