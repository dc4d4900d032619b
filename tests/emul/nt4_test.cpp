// CPU check of csrc/nt4_encode.h: the vectorised ASCII -> nt4 encoder against the byte-wise table on every byte value at every
// alignment and on random buffers of awkward lengths.  Prints "bad 0" when all agree.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nt4_encode.h"

int main()
{
	long bad = 0, checked = 0;
	std::vector<char> src(4096 + 64);
	std::vector<uint8_t> dst(src.size() + 64, 0xAA);
	for (int off = 0; off < 33; ++off) { // every byte value, at every offset inside a 32-byte step
		for (int v = 0; v < 256; ++v) src[off + v] = (char)v;
		gd_nt4_encode(src.data() + off, dst.data() + off, 256);
		for (int v = 0; v < 256; ++v, ++checked) bad += dst[off + v] != gd_nt4_byte((unsigned char)v);
	}
	unsigned s = 12345;
	const char alpha[] = "ACGTacgtNnUuRY-*\0\1\2\3\4\x80\xff";
	for (int rep = 0; rep < 2000; ++rep) {
		s = s * 1103515245u + 12345u;
		const size_t n = (s >> 8) % 4000, off = (s >> 3) % 31;
		for (size_t i = 0; i < n; ++i) { s = s * 1103515245u + 12345u; src[off + i] = (s >> 28) < 13 ? alpha[(s >> 16) % (sizeof alpha - 1)] : (char)(s >> 9); }
		for (size_t i = 0; i < n + 8; ++i) dst[off + i] = 0xAA;
		gd_nt4_encode(src.data() + off, dst.data() + off, n);
		for (size_t i = 0; i < n; ++i, ++checked) bad += dst[off + i] != gd_nt4_byte((unsigned char)src[off + i]);
		for (size_t i = n; i < n + 8; ++i) bad += dst[off + i] != 0xAA; // nothing written past the end
	}
	printf("checked %ld bad %ld\n", checked, bad);
	return bad != 0;
}
