// CPU test harness of csrc/fastx_reader.h (built with -fsanitize=thread by tests/test_fastx_reader.py): reads a file with 1 and
// with N parser threads in small blocks and checks that both give the same records and batch boundaries.
#include "fastx_reader.h"
#include <stdio.h>

static std::string digest(const char *path, int threads, size_t block, int64_t chunk, long *n_out)
{
	GdFastx *fx = gd_fastx_open(path);
	if (!fx) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
	fx->n_threads = threads, fx->block_size = block;
	std::string d;
	long n = 0;
	for (;;) {
		bool bad = false;
		const int k = fx->read_batch(chunk, true, true, false, &bad);
		if (k <= 0 && !bad) break;
		for (int i = 0; i < k; ++i) {
			d += fx->v_name[i], d += '\t', d += fx->v_comment[i] ? fx->v_comment[i] : "-", d += '\t', d += fx->v_seq[i], d += '\t';
			d += fx->v_qual[i] ? fx->v_qual[i] : "-", d += '\n';
		}
		d += bad ? "==bad==\n" : "==\n";
		n += k;
	}
	gd_fastx_close(fx);
	*n_out = n;
	return d;
}

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	long n1 = 0, n2 = 0;
	const std::string a = digest(argv[1], 1, 1 << 16, 20000, &n1);
	for (int threads : {2, 4, 7})
		for (size_t block : {(size_t)2048, (size_t)1 << 16}) {
			const std::string b = digest(argv[1], threads, block, 20000, &n2);
			if (a != b) { printf("MISMATCH threads %d block %zu (%ld vs %ld records)\n", threads, block, n1, n2); return 1; }
		}
	printf("ok %ld records\n", n1);
	return 0;
}
