/*
 * oracle/gdo_ksw2.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C scalar restatement of the reference's banded affine-gap DP kernels and backtrack.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * What is restated (reference = /root/reference, SR/ = GDiet-ShortReads/, identical in GDiet-LongReads/):
 *   gdo_ksw_extd2   <- SR/ksw2_extd2_sse.c:34-401   (dual affine gap; the only DP kernel on the live path)
 *   gdo_ksw_extz2   <- SR/ksw2_extz2_sse.c:31-312   (single affine gap; BASELINE config 2)
 *   gdo_backtrack   <- SR/ksw2.h:131-163 (+ ksw_push_cigar :115-125)
 *   gdo_exact_match <- SR/exact_match_sse.c:23-91
 *   gdo_ksw_exts2   <- SR/ksw2_exts2_sse.c:34-416   (splice-aware extension; SURVEY 8f rank 4, dead code in GDiet)
 *   gdo_lchain_dp   <- SR/lchain.c:9-190            (mg_lchain_dp + mg_chain_backtrack + compact_a; oracle/gdo_lchain.c; same rank)
 *
 * Pinning: oracle/pin_ksw2.py checks these against the reference itself (oracle/_ref/libgdiet_*.so built by
 * oracle/Makefile.ref: ksw_extd2_sse, ksw_extd2_avx512, ksw_extz2_sse, exact_match_sse) on seeded fuzz, and
 * tests/test_oracle_ksw2.py checks them against the committed golden vectors in tests/golden/ksw2_*.bin that
 * were produced by the reference.
 */
#ifndef GDO_KSW2_H
#define GDO_KSW2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDO_NEG_INF (-0x40000000)

/* flag bits: same numeric values as SR/ksw2.h:9-18 */
#define GDO_EZ_SCORE_ONLY  0x01
#define GDO_EZ_RIGHT       0x02
#define GDO_EZ_GENERIC_SC  0x04
#define GDO_EZ_APPROX_MAX  0x08
#define GDO_EZ_APPROX_DROP 0x10
#define GDO_EZ_EXTZ_ONLY   0x40
#define GDO_EZ_REV_CIGAR   0x80
#define GDO_EZ_SPLICE_FOR  0x100
#define GDO_EZ_SPLICE_REV  0x200
#define GDO_EZ_SPLICE_FLANK 0x400
/* ours: score cells the way ksw_extd2_avx512 does (see fill_scores); only differs from the SSE rule for bytes outside 0..4 */
#define GDO_EZ_AVX512_SC   0x10000

/* mirrors ksw_extz_t, SR/ksw2.h:31-40 (plain ints instead of bit-fields) */
typedef struct {
	uint32_t max;
	int zdropped;
	int max_q, max_t;
	int mqe, mqe_t;
	int mte, mte_q;
	int score;
	int m_cigar, n_cigar;
	int reach_end;
	uint32_t *cigar; /* malloc'd; caller frees */
} gdo_extz_t;

void gdo_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                   gdo_extz_t *ez);

void gdo_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, gdo_extz_t *ez);

/* SURVEY 8f rank 4 (not called by GDiet): the splice-aware extension kernel, SR/ksw2_exts2_sse.c:34-416; junc may be NULL */
void gdo_ksw_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc,
                   gdo_extz_t *ez);

/* returns 1 if query[0..qlen) == target[0..qlen) under the reference's 16-byte-chunk rule, else 0 */
int gdo_exact_match(int qlen, const uint8_t *query, int tlen, const uint8_t *target);

/* p: backtrace matrix, rotated layout; see SR/ksw2.h:127-163 */
void gdo_backtrack(int is_rev, int min_intron_len, const uint8_t *p, const int *off, const int *off_end, int n_col,
                   int i0, int j0, int *m_cigar_, int *n_cigar_, uint32_t **cigar_);

#ifdef __cplusplus
}
#endif
#endif
