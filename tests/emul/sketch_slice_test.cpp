// CPU check of the exact-slicing property the wave-parallel seed kernel relies on: the concatenation of 64 (or any number
// of) gd_sketch_slice() outputs equals the sequential gd_sketch_core() output, including reads with Ns, tandem repeats and
// low-complexity runs (where the "identical k-mer" emission rules fire).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <vector>
#define __host__
#define __device__
#include "map_stages.h"

struct EmitV { std::vector<GdMini> *v; bool operator()(const GdMini &m) { v->push_back(m); return false; } };

int main(int argc, char **argv)
{
	std::mt19937 g(argc > 1 ? atoi(argv[1]) : 1);
	const int iters = argc > 2 ? atoi(argv[2]) : 300;
	int bad = 0;
	for (int it = 0; it < iters; ++it) {
		const int k = it % 3 == 0 ? 15 : it % 3 == 1 ? 19 : 21, w = it % 3 == 0 ? 10 : it % 3 == 1 ? 19 : 11;
		GdPattern P;
		const char *pats[] = {"10", "110", "1", "1110", "100"};
		const char *pz = pats[it % 5];
		gd_pattern_init(P, pz, (int)strlen(pz));
		const unsigned len = 100 + g() % 20000;
		std::vector<uint8_t> s(len + 8);
		for (auto &c : s) c = g() & 3;
		if (it % 4 == 1) for (unsigned i = 0; i < len; ++i) if (g() % 300 == 0) s[i] = 4;               // scattered Ns
		if (it % 4 == 2) { unsigned p = g() % len, n = 50 + g() % 400; for (unsigned i = p; i < p + n && i < len; ++i) s[i] = 4; } // N run
		if (it % 5 == 3) { unsigned p = g() % len, n = 200 + g() % 2000, u = 1 + g() % 7; for (unsigned i = p + u; i < p + n && i < len; ++i) s[i] = s[i - u]; } // tandem repeat
		if (it % 7 == 5) { unsigned p = g() % len, n = 100 + g() % 800; for (unsigned i = p; i < p + n && i < len; ++i) s[i] = 0; } // homopolymer
		const unsigned shift = g() % P.W;
		const unsigned dl = gd_diet_len(P, len, shift);
		std::vector<GdMini> seq, par;
		EmitV e1 = {&seq};
		gd_sketch_core(s.data(), dl, w, k, 0, shift, P, true, e1);
		const unsigned nchunk = 1 + g() % 64, chunk = (dl + nchunk - 1) / nchunk;
		for (unsigned c = 0; c < nchunk && chunk; ++c) {
			const unsigned i0 = c * chunk, i1 = i0 + chunk < dl ? i0 + chunk : dl;
			if (i0 >= dl) break;
			EmitV e2 = {&par};
			gd_sketch_slice(s.data(), dl, i0, i1, w, k, 0, shift, P, true, e2);
		}
		bool ok = seq.size() == par.size();
		for (size_t i = 0; ok && i < seq.size(); ++i) ok = seq[i].x == par[i].x && seq[i].y == par[i].y;
		if (!ok) { ++bad; if (bad < 5) fprintf(stderr, "MISMATCH it=%d len=%u k=%d w=%d pat=%s shift=%u chunks=%u seq=%zu par=%zu\n", it, len, k, w, pz, shift, nchunk, seq.size(), par.size()); }
	}
	printf("sketch_slice_test iters=%d mismatches=%d\n", iters, bad);
	return bad != 0;
}
