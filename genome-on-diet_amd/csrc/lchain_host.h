// SURVEY 8f rank 4 (chaining half), host stage: the chains of one read from the f / p / v arrays the device filled --
// mg_chain_backtrack (SR/lchain.c:9-53) and compact_a (:55-89).  Both sort with radix_sort_128x (SR/ksort.h:101-151 at
// SR/misc.c:155-156), an in-place most-significant-digit radix sort that is NOT stable: the order of chains with equal scores, and
// through it which chain claims a shared anchor first, is the order that very algorithm leaves, so it is restated here
// (gdl_radix_sort) instead of replaced.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

struct GdlPair { uint64_t x, y; };

static inline void gdl_rs_insertsort(GdlPair *beg, GdlPair *end)
{
	for (GdlPair *i = beg + 1; i < end; ++i)
		if (i->x < (i - 1)->x) {
			GdlPair *j, tmp = *i;
			for (j = i; j > beg && tmp.x < (j - 1)->x; --j) *j = *(j - 1);
			*j = tmp;
		}
}
static inline void gdl_rs_sort(GdlPair *beg, GdlPair *end, int n_bits, int s)
{
	struct Bucket { GdlPair *b, *e; };
	const int size = 1 << n_bits, m = size - 1;
	Bucket b[256], *be = b + size, *k;
	for (k = b; k != be; ++k) k->b = k->e = beg;
	for (GdlPair *i = beg; i != end; ++i) ++b[i->x >> s & m].e;
	for (k = b + 1; k != be; ++k) k->e += (k - 1)->e - beg, k->b = (k - 1)->e;
	for (k = b; k != be;) {
		if (k->b != k->e) {
			Bucket *l;
			if ((l = b + (k->b->x >> s & m)) != k) { // cycle leader permutation: every displaced element goes to the head of its bucket
				GdlPair tmp = *k->b, swap;
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
			if (k->e - k->b > 64) gdl_rs_sort(k->b, k->e, n_bits, s);
			else if (k->e - k->b > 1) gdl_rs_insertsort(k->b, k->e);
	}
}
static inline void gdl_radix_sort(GdlPair *beg, GdlPair *end)
{
	if (end - beg <= 64) gdl_rs_insertsort(beg, end);
	else gdl_rs_sort(beg, end, 8, 56);
}

// a: the read's n anchors; f / p / v from lchain_fill_kernel (v is overwritten, as in the reference).  u_out (room for n entries)
// receives score << 32 | n_anchors per chain, a_out (room for n anchors) the chains' anchors; returns n_u, *n_v_ = anchors written.
static inline int gdl_chains_of_read(int64_t n, const GdlPair *a, const int32_t *f, const int32_t *p, int32_t *v, int32_t min_cnt, int32_t min_sc,
                                     uint64_t *u_out, GdlPair *a_out, int64_t *n_v_, std::vector<GdlPair> &z, std::vector<int32_t> &t, std::vector<GdlPair> &b,
                                     std::vector<uint64_t> &u2)
{
	*n_v_ = 0;
	int64_t n_z = 0, i, k, n_v;
	int32_t n_u;
	for (i = 0; i < n; ++i) n_z += f[i] >= min_sc;
	if (n_z == 0) return 0;
	z.resize((size_t)n_z);
	for (i = 0, k = 0; i < n; ++i)
		if (f[i] >= min_sc) z[k].x = (uint64_t)(int64_t)f[i], z[k++].y = (uint64_t)i; // (z[k].x = f[i]: int32 -> uint64, sign-extended as in C)
	gdl_radix_sort(z.data(), z.data() + n_z);
	t.assign((size_t)n, 0);
	// (the reference counts the chains in a first pass only to size u[]; one pass suffices here)
	for (k = n_z - 1, n_v = 0, n_u = 0; k >= 0; --k) {
		const int64_t n_v0 = n_v;
		for (i = (int64_t)z[k].y; i >= 0 && t[i] == 0; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
		const int32_t sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u_out[n_u++] = (uint64_t)(uint32_t)sc << 32 | (uint64_t)(n_v - n_v0);
		else n_v = n_v0;
	}
	if (n_u == 0) return 0;
	// compact_a: anchors of every chain in ascending order, chains sorted by the position of their first anchor
	b.resize((size_t)n_v);
	for (i = 0, k = 0; i < n_u; ++i) {
		const int32_t k0 = (int32_t)k, ni = (int32_t)u_out[i];
		for (int32_t j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	z.resize((size_t)n_u);
	for (i = k = 0; i < n_u; ++i) {
		z[i].x = b[k].x, z[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u_out[i];
	}
	gdl_radix_sort(z.data(), z.data() + n_u);
	u2.resize((size_t)n_u);
	for (i = k = 0; i < n_u; ++i) {
		const int32_t j = (int32_t)z[i].y, nn = (int32_t)u_out[j];
		u2[i] = u_out[j];
		memcpy(&a_out[k], &b[z[i].y >> 32], (size_t)nn * sizeof(GdlPair));
		k += nn;
	}
	memcpy(u_out, u2.data(), (size_t)n_u * 8);
	*n_v_ = k;
	return n_u;
}
