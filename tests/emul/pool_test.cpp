// CPU test of csrc/host_pool.h: several caller threads (the batches in flight) run parallel loops through ONE pool at the same
// time; every item of every loop must be executed exactly once, and the loops must return only when all their items are done.
// Built with -fsanitize=thread by tests/test_host_pool.py.
#include "host_pool.h"
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
	const int callers = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 200, width = argc > 3 ? atoi(argv[3]) : 6;
	GdPool pool;
	std::atomic<long> bad{0};
	std::vector<std::thread> th;
	for (int c = 0; c < callers; ++c)
		th.emplace_back([&, c] {
			for (int r = 0; r < rounds; ++r) {
				const int n = 1 + ((c * 7919 + r * 104729) % 5000); // from below the serial threshold to a few hundred chunks
				std::vector<int> hits(n, 0);
				std::atomic<long> sum{0};
				pool.run(width, n, [&](int i) { hits[i] += 1; sum.fetch_add(i); });
				long want = (long)n * (n - 1) / 2;
				if (sum.load() != want) bad.fetch_add(1);
				for (int i = 0; i < n; ++i) if (hits[i] != 1) bad.fetch_add(1);
			}
		});
	for (auto &t : th) t.join();
	if ((int)pool.th.size() > width - 1) { printf("pool grew to %zu threads for width %d\n", pool.th.size(), width); return 1; }
	printf("callers %d rounds %d width %d threads %zu bad %ld\n", callers, rounds, width, pool.th.size(), bad.load());
	return bad.load() ? 1 : 0;
}
