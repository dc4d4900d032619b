// ASCII -> nt4 of a whole read (seq_nt4_table, LR/sketch.c:11-18: A/a 0, C/c 1, G/g 2, T/t/U/u 3, the raw bytes 0-3 themselves, anything
// else 4), 32 bytes per step where the CPU has AVX2.  A mini-batch of HiFi reads is 77 MB; byte by byte the encoding kept the host
// threads busy for longer than the copy to the device takes, beside the host stages of the batches in flight.
#pragma once
#include <stddef.h>
#include <stdint.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

static inline uint8_t gd_nt4_byte(unsigned char c)
{
	switch (c) {
	case 0: case 'A': case 'a': return 0;
	case 1: case 'C': case 'c': return 1;
	case 2: case 'G': case 'g': return 2;
	case 3: case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

#if defined(__x86_64__)
// A byte is a letter of the alphabet iff its low nibble is one the letters use AND its high nibble is of the matching kind:
// bit 0 = high nibble 4 / 6 (A C G, a c g), bit 1 = high nibble 5 / 7 (T U, t u), bit 2 = high nibble 0 (the raw codes 0-3).
__attribute__((target("avx2"))) static inline void gd_nt4_encode_avx2(const char *src, uint8_t *dst, size_t n)
{
	const __m256i lo_kind = _mm256_setr_epi8(4, 5, 4, 5, 2, 2, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 4, 5, 4, 5, 2, 2, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0);
	const __m256i hi_kind = _mm256_setr_epi8(4, 0, 0, 0, 1, 2, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 4, 0, 0, 0, 1, 2, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0);
	const __m256i lo_code = _mm256_setr_epi8(0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0); // code of the LETTER with that low nibble
	const __m256i nib = _mm256_set1_epi8(0x0f), four = _mm256_set1_epi8(4), zero = _mm256_setzero_si256();
	size_t i = 0;
	for (; i + 32 <= n; i += 32) {
		const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i));
		const __m256i lo = _mm256_and_si256(c, nib), hi = _mm256_and_si256(_mm256_srli_epi16(c, 4), nib);
		const __m256i ok = _mm256_and_si256(_mm256_shuffle_epi8(lo_kind, lo), _mm256_shuffle_epi8(hi_kind, hi)); // != 0: a letter or a raw code
		const __m256i raw = _mm256_cmpeq_epi8(hi, zero);                                                          // high nibble 0: the code is the byte itself
		const __m256i code = _mm256_blendv_epi8(_mm256_shuffle_epi8(lo_code, lo), lo, raw);
		_mm256_storeu_si256((__m256i *)(dst + i), _mm256_blendv_epi8(code, four, _mm256_cmpeq_epi8(ok, zero)));
	}
	for (; i < n; ++i) dst[i] = gd_nt4_byte((unsigned char)src[i]);
}
#endif

static inline void gd_nt4_encode(const char *src, uint8_t *dst, size_t n)
{
#if defined(__x86_64__)
	static const bool avx2 = __builtin_cpu_supports("avx2");
	if (avx2) { gd_nt4_encode_avx2(src, dst, n); return; }
#endif
	for (size_t i = 0; i < n; ++i) dst[i] = gd_nt4_byte((unsigned char)src[i]);
}
