// Host driver of the product's chaining code (test infrastructure): the fill of mg_lchain_dp (SR/lchain.c:139-181) done
// sequentially on the CPU with the product's own pair score (genome-on-diet_amd/csrc/lchain_core.h: the float arithmetic the device
// kernel compiles) followed by the product's host stage (lchain_host.h: backtrack, compaction, the restated radix sort).  Built as
// a shared library by tests/test_rank4_oracle.py and compared with the reference's golden outputs -- the part of
// gdiet_hip_lchain_dp_batch that can be checked without a GPU.
#include <stdint.h>
#include <vector>
#include "lchain_core.h"
#include "lchain_host.h"

extern "C" int lchain_emul(int64_t n, const uint64_t *a_, int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                           float chn_pen_gap, float chn_pen_skip, int is_cdna, int n_seg, uint64_t *u, uint64_t *a_out, int64_t *n_v)
{
	const GdlPair *a = (const GdlPair *)a_;
	GdChainOpt O;
	O.max_dist_x = max_dist_x < bw ? bw : max_dist_x;
	O.max_dist_y = (max_dist_y < bw && !is_cdna) ? bw : max_dist_y;
	O.bw = bw, O.max_skip = max_skip, O.max_iter = max_iter, O.min_cnt = min_cnt, O.min_sc = min_sc;
	O.chn_pen_gap = chn_pen_gap, O.chn_pen_skip = chn_pen_skip, O.is_cdna = is_cdna, O.n_seg = n_seg;
	std::vector<int32_t> f(n), p(n), v(n), t(n, 0);
	int64_t st = 0, max_ii = -1;
	for (int64_t i = 0; i < n; ++i) {
		int64_t max_j = -1, end_j, j;
		int32_t max_f = (int32_t)(a[i].y >> 32 & 0xff), n_skip = 0;
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)(int64_t)O.max_dist_x)) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc = gdl_comput_sc(a[i].x, a[i].y, a[j].x, a[j].y, O);
			if (sc == INT32_MIN) continue;
			sc += f[j];
			if (sc > max_f) {
				max_f = sc, max_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		end_j = j;
		if (max_ii < 0 || (uint64_t)(a[i].x - a[max_ii].x) > (uint64_t)(int64_t)O.max_dist_x) { // unsigned, as SR/lchain.c:165
			int32_t mx = INT32_MIN;
			max_ii = -1;
			for (j = i - 1; j >= st; --j)
				if (mx < f[j]) mx = f[j], max_ii = j;
		}
		if (max_ii >= 0 && max_ii < end_j) {
			const int32_t tmp = gdl_comput_sc(a[i].x, a[i].y, a[max_ii].x, a[max_ii].y, O);
			if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) max_f = tmp + f[max_ii], max_j = max_ii;
		}
		f[i] = max_f, p[i] = (int32_t)max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
		if (max_ii < 0 || ((uint64_t)(a[i].x - a[max_ii].x) <= (uint64_t)(int64_t)O.max_dist_x && f[max_ii] < f[i])) max_ii = i;
	}
	std::vector<GdlPair> z, b;
	std::vector<int32_t> tt;
	std::vector<uint64_t> u2;
	return gdl_chains_of_read(n, a, f.data(), p.data(), v.data(), min_cnt, min_sc, u, (GdlPair *)a_out, n_v, z, tt, b, u2);
}
