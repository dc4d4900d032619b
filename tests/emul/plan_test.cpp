// CPU test: the O(1) admission test of the register-resident DP kernels (gd_wave_geometry_ok, ksw_wave_core.h) against its
// block-by-block definition (gd_wave_geometry_ok_loop) on random geometries -- short and long, narrow and wide bands, every lane count
// the kernels are built for.  usage: plan_test [n_cases]   (prints "<cases> <admitted> <differ>")
#define __host__
#define __device__
#include <initializer_list>
#include <stdio.h>
#include <stdlib.h>
#include "ksw_wave_core.h"

int main(int argc, char **argv)
{
	const long n_it = argc > 1 ? atol(argv[1]) : 2000000;
	srand(7);
	long n = 0, diff = 0, ok = 0;
	for (long it = 0; it < n_it; ++it) {
		const int wmax = it % 4 == 0 ? 40 : it % 4 == 1 ? 400 : it % 4 == 2 ? 2000 : 4000;
		const int w = 1 + rand() % wmax;
		const int lmax = it % 5 == 0 ? 300 : it % 5 == 1 ? 6000 : it % 5 == 2 ? 40000 : it % 5 == 3 ? 200000 : 2000;
		int qlen = 1 + rand() % lmax, tlen = qlen + (rand() % (2 * w + 40)) - w - 20;
		if (rand() % 5 == 0) tlen = 1 + rand() % lmax;
		if (rand() % 9 == 0) tlen = qlen + (rand() % 5) - 2, qlen += (rand() % 3) - 1; // near-square, as mapped reads are
		if (tlen < 1) tlen = 1;
		if (qlen < 1) qlen = 1;
		for (int lanes : {8, 10, 16, 64, 96, 128}) {
			const bool a = gd_wave_geometry_ok_loop(qlen, tlen, w, lanes), b = gd_wave_geometry_ok(qlen, tlen, w, lanes);
			++n, ok += a;
			if (a != b) {
				if (diff < 5) fprintf(stderr, "DIFF qlen %d tlen %d w %d lanes %d: loop %d fast %d\n", qlen, tlen, w, lanes, (int)a, (int)b);
				++diff;
			}
		}
	}
	printf("%ld %ld %ld\n", n, ok, diff);
	return diff != 0;
}
