/*
 * gdiet_hip.h -- C ABI of the MI355X (gfx950) implementation of Genome-on-Diet's per-read mapping hot path.
 *
 * Plain C, plain pointers and sizes: this is what the reference's C host (map.c's worker loop) binds.
 * Every entry point names the reference interface it replaces (paths relative to the reference tree;
 * SR/ = GDiet-ShortReads/, LR/ = GDiet-LongReads/; the two trees share every file cited here unless noted).
 *
 * Conventions
 *   - all functions return 0 on success, a negative GDIET_E_* code on failure; gdiet_hip_strerror() explains it.
 *     There is NO CPU fallback inside the library: if no gfx950 device / code object is usable the call fails.
 *   - "host" entry points take host pointers and do H2D/D2H themselves; "_dev" entry points take device
 *     pointers (e.g. torch tensors' data_ptr()) plus a hipStream_t passed as void*, and never synchronise.
 *   - sequences are nt4-encoded bytes 0..4 (A,C,G,T,N) exactly as the reference feeds ksw2 (LR/map.c:1622-1643,
 *     SR/index.c:183-196 mm_idx_getseq2).
 */
#ifndef GDIET_HIP_H
#define GDIET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDIET_OK            0
#define GDIET_E_NODEVICE   (-1) /* no HIP device / not gfx950 */
#define GDIET_E_HIP        (-2) /* a HIP runtime call failed */
#define GDIET_E_PARAM      (-3) /* invalid or unsupported argument (see gdiet_hip_strerror) */
#define GDIET_E_NOMEM      (-4) /* device workspace too small for the batch */
#define GDIET_E_CIGAR_CAP  (-5) /* a CIGAR did not fit the caller's capacity (n_cigar[i] holds the needed size) */

#define GDIET_KSW_NEG_INF (-0x40000000) /* KSW_NEG_INF, SR/ksw2.h:7 */

/* flag bits understood by the DP entry points; numeric values of SR/ksw2.h:9-18 */
#define GDIET_EZ_APPROX_MAX 0x08 /* the only mode the live path uses: LR/map.c:1742, SR/map.c:867 */

typedef struct gdiet_ctx gdiet_ctx; /* one per (process, GPU); owns the stream(s) and the device workspace */

/* scoring of one DP batch: the scalar arguments of ksw_extd2_sse (SR/ksw2.h:68-69) with the 5x5 matrix reduced
 * to what the non-GENERIC_SC path reads (mat[0], mat[1], mat[24]; SR/ksw2_extd2_sse.c:85-87). */
typedef struct {
	int8_t match;      /* mat[0]   (> 0) */
	int8_t mismatch;   /* mat[1]   (< 0) */
	int8_t sc_ambi;    /* mat[24]  (0 on the live path => ambiguous bases score -e2) */
	int8_t q, e, q2, e2; /* gap open / extend of the two affine models, as passed by the caller (un-swapped) */
	int8_t reserved;
	int32_t flag;      /* GDIET_EZ_APPROX_MAX (global alignment with CIGAR, left-aligned gaps, no z-drop) */
} gdiet_ksw_score_t;

/* ---- life cycle --------------------------------------------------------------------------------------------- */
int gdiet_hip_init(gdiet_ctx **ctx, int device_ordinal);   /* replaces nothing: GPU context for one device */
void gdiet_hip_destroy(gdiet_ctx *ctx);
const char *gdiet_hip_strerror(const gdiet_ctx *ctx);       /* text of the last failure on this context */
int gdiet_hip_device_name(const gdiet_ctx *ctx, char *buf, size_t len);
/* which kernel variants handled the last batch: bit0 = register-resident 64-lane wave kernel, bit1 = generic LDS kernel,
 * bit2 = register-resident short-alignment kernels (several alignments per wavefront), bit3 = two-blocks-per-lane kernel (wide bands),
 * bit4 = some of bit2's alignments (full matrices of one geometry: a short-read batch) ran as skewed pipelines (ksw_pipe.hip.h) */
int gdiet_hip_last_kernel_mask(const gdiet_ctx *ctx);

/* ---- B3: batched banded dual-affine global alignment -------------------------------------------------------
 * Replaces, for a whole batch, the per-candidate sequence of calls in mm_map_frag:
 *     [exact_match_sse]  ->  ksw_extd2_avx512 | ksw_extd2_sse  ->  ksw_backtrack        (LR/map.c:1748-1806,
 *     SR/map.c:873-929; kernels SR/ksw2_extd2_sse.c:34-401, SR/ksw2_extd2_avx.c:72-913, SR/ksw2.h:131-163,
 *     SR/exact_match_sse.c:23-91).
 *
 * For alignment i (0 <= i < n):
 *   query  = qseq[qoff[i] .. qoff[i+1])       target = tseq[toff[i] .. toff[i+1])
 *   w[i]   = band width argument of ksw_extd2 (w < 0: max(qlen,tlen) as in the reference)
 *   exact_score[i] (may be NULL): if non-NULL and exact_score[i] != GDIET_KSW_NEG_INF and qlen == tlen, the
 *            exact-match pre-filter runs first; on a match score[i] = exact_score[i] and the CIGAR is "<qlen>M"
 *            (the caller passes qlen_sum * a -- bug-compatibility item 5 of SURVEY 8; LR/map.c:1782).
 * Outputs
 *   score[i]   = ez.score (GDIET_KSW_NEG_INF if the band emptied or the corner was not reached)
 *   n_cigar[i] = ez.n_cigar; cigar ops (BAM encoding len<<4|op, op 0=M 1=I 2=D) at cigar[cigar_off[i] ..)
 *   the caller provides cigar_off[n+1]; capacity of alignment i is cigar_off[i+1]-cigar_off[i]
 *   (qlen+tlen always suffices).
 * How a result is obtained is the library's business, what it is is the reference's: short alignments (a target of at most 256 bases) whose
 * result can be proven without the DP -- an N-free pair of equal length with so few mismatches that no gapped path can reach the main
 * diagonal's score: m (a + b) <= a + 2 (q + e) -- are answered by the pre-filter kernel with exactly what ksw_extd2 + ksw_backtrack return
 * for them (tests/test_oracle_ksw2.py pins the property on the reference's own ksw_extd2_sse).  GDIET_DIAG_SHORTCUT=0 in
 * the environment sends every alignment through the DP kernels and the walk.
 */
int gdiet_hip_ksw_extd2_batch(gdiet_ctx *ctx, int n,
                              const uint8_t *qseq, const int64_t *qoff,
                              const uint8_t *tseq, const int64_t *toff,
                              const int32_t *w, const int32_t *exact_score,
                              const gdiet_ksw_score_t *sc,
                              int32_t *score, int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off);

/* same, everything already resident in HBM (device pointers); asynchronous on `stream` (hipStream_t).
 * Workspace for the backtrace matrices comes from the context (gdiet_hip_reserve). */
int gdiet_hip_ksw_extd2_batch_dev(gdiet_ctx *ctx, int n,
                                  const uint8_t *d_qseq, const int64_t *d_qoff,
                                  const uint8_t *d_tseq, const int64_t *d_toff,
                                  const int32_t *d_w, const int32_t *d_exact_score,
                                  const gdiet_ksw_score_t *sc, /* host */
                                  int32_t *d_score, int32_t *d_n_cigar, uint32_t *d_cigar, const int64_t *d_cigar_off,
                                  const int64_t *h_qoff, const int64_t *h_toff, const int32_t *h_w, /* host copies for planning */
                                  void *stream);

/* ---- K3: batched banded SINGLE-affine global alignment -------------------------------------------------------------
 * Replaces ksw_extz2_sse + ksw_backtrack (SR/ksw2_extz2_sse.c:31-312, SR/ksw2.h:62; BASELINE config 2; not called by the live
 * mapping path).  Same batch layout and outputs as gdiet_hip_ksw_extd2_batch; sc->q / sc->e are the gap costs, sc->q2 / sc->e2
 * are ignored, sc->flag must be GDIET_EZ_APPROX_MAX (the only mode GDiet ever passes; the exact-maximum mode is
 * gdiet_hip_ksw_extz2_batch_ex below); sequence bytes must be 0..4 (the one
 * out-of-alphabet byte GDiet produces, 7 for a reverse-complemented N, only ever reaches ksw_extd2).  Implementation note: in
 * that mode ksw_extz2(q,e) and ksw_extd2(q,e,q,e) visit identical cells with identical values (the unsigned bias of extz2 is a
 * re-labelling, and a second gap model equal to the first can never win the priority chain), so the register-resident kernels
 * run their single-affine form -- the dual-affine recurrence with the X2 / Y2 half dropped -- and the generic LDS kernel runs
 * with both gap models equal; tests pin this against ksw_extz2_sse's own outputs (tests/golden/ksw2_extz2.npz). */
int gdiet_hip_ksw_extz2_batch(gdiet_ctx *ctx, int n,
                              const uint8_t *qseq, const int64_t *qoff,
                              const uint8_t *tseq, const int64_t *toff,
                              const int32_t *w, const gdiet_ksw_score_t *sc,
                              int32_t *score, int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off);

/* K3, exact-maximum mode: ksw_extz2_sse called WITHOUT KSW_EZ_APPROX_MAX (SR/ksw2_extz2_sse.c:226-268: the per-cell H array, the
 * row maximum with z-drop, SR/ksw2.h:172-188), the second mode SURVEY 8d names for BASELINE config 2.  sc->flag = 0 or
 * GDIET_EZ_EXTZ_ONLY (SR/ksw2.h:15); zdrop / end_bonus as the reference's arguments (zdrop < 0: no z-drop).  ez[i] receives every
 * scalar of ksw_extz_t (SR/ksw2.h:31-40); the CIGAR is the walk from the last cell, from (mqe_t, qlen-1) when EXTZ_ONLY reaches
 * the query end, or from (max_t, max_q) after a z-drop, exactly as :296-305.  One wavefront per alignment, state in LDS. */
#define GDIET_EZ_EXTZ_ONLY 0x40
typedef struct {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
} gdiet_ksw_extz_t;
int gdiet_hip_ksw_extz2_batch_ex(gdiet_ctx *ctx, int n,
                                 const uint8_t *qseq, const int64_t *qoff,
                                 const uint8_t *tseq, const int64_t *toff,
                                 const int32_t *w, const gdiet_ksw_score_t *sc, int32_t zdrop, int32_t end_bonus,
                                 gdiet_ksw_extz_t *ez, int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off);

/* ---- SURVEY 8f rank 4 (kernel half): batched splice-aware extension alignment -------------------------------------------
 * Replaces ksw_exts2_sse + ksw_backtrack (SR/ksw2_exts2_sse.c:34-416, SR/ksw2.h:71-72,131-163).  GDiet keeps this kernel of
 * minimap2 in its tree (SR/align.c:363 is its only call site, under MM_F_SPLICE) but no GDiet preset reaches it; it is provided
 * for a minimap2-compatible splice mode.  Arguments as the reference's: mat = the 5 x 5 matrix (m = 5), q / e the short gap,
 * q2 the long-gap (intron) open cost (must exceed q + e), noncan the penalty of a non-canonical splice site, junc = one
 * annotation byte per target base laid out like tseq (NULL: none) with junc_bonus, flag = any of KSW_EZ_SCORE_ONLY 0x01,
 * RIGHT 0x02, GENERIC_SC 0x04, APPROX_MAX 0x08, APPROX_DROP 0x10, EXTZ_ONLY 0x40, REV_CIGAR 0x80, SPLICE_FOR 0x100,
 * SPLICE_REV 0x200, SPLICE_FLANK 0x400 (SR/ksw2.h:9-18).  No band (the reference has none).  ez[i]: every scalar of ksw_extz_t;
 * CIGAR in BAM ops incl. N (3) for long gaps of at least long_thres bases.  One wavefront per alignment, state in LDS:
 * min(qlen, tlen) is limited to ~8000. */
int gdiet_hip_ksw_exts2_batch(gdiet_ctx *ctx, int n,
                              const uint8_t *qseq, const int64_t *qoff,
                              const uint8_t *tseq, const int64_t *toff, const uint8_t *junc,
                              const int8_t *mat, int8_t q, int8_t e, int8_t q2, int8_t noncan, int32_t zdrop, int8_t junc_bonus, int32_t flag,
                              gdiet_ksw_extz_t *ez, int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off);

/* ---- SURVEY 8f rank 4 (chaining half): batched anchor chaining ------------------------------------------------------------
 * Replaces mg_lchain_dp (SR/lchain.c:124-190, SR/mmpriv.h:102-104) for a batch of reads; like ksw_exts2 it is code of minimap2
 * that GDiet keeps and never calls (GDiet votes instead of chaining), provided for a minimap2-compatible mode.  a = the
 * anchors of all reads as (x, y) pairs of uint64 (x: rev<<63 | tid<<32 | tpos -- minimap2's layout, SR/hit.c:26; any layout whose
 * upper 32 bits name the target + strand and whose lower 32 bits are the position chains alike --, y: flags<<40 | q_span<<32 | q_pos; per read
 * sorted by x as minimap2 sorts them), read i = pairs aoff[i] .. aoff[i+1]; the scalar parameters are the reference's.
 * Outputs, per read at the read's own offset (a chain set never has more chains or anchors than the read has anchors):
 * n_u[i] chains, u[aoff[i] + k] = score<<32 | n_anchors of chain k, a_out (pairs, at 2 * aoff[i]) = the chains' anchors, n_v[i]
 * of them -- the very arrays mg_lchain_dp returns (same chain order, same anchor order).  The reference frees its input; here
 * `a` stays untouched.  The O(n * max_iter) fill runs on the GPU, one wavefront per read; the chains are cut on host threads. */
int gdiet_hip_lchain_dp_batch(gdiet_ctx *ctx, int n_reads, const uint64_t *a, const int64_t *aoff,
                              int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t max_skip, int32_t max_iter, int32_t min_cnt, int32_t min_sc,
                              float chn_pen_gap, float chn_pen_skip, int32_t is_cdna, int32_t n_seg,
                              int32_t *n_u, int64_t *n_v, uint64_t *u, uint64_t *a_out);

/* make sure the context owns at least `bytes` of device workspace (backtrace arena); returns GDIET_E_NOMEM
 * if the device cannot provide it.  gdiet_hip_ksw_extd2_batch() grows the arena by itself. */
int gdiet_hip_reserve(gdiet_ctx *ctx, size_t bytes);
/* bytes of backtrace arena the batch (host-side lengths) needs */
size_t gdiet_hip_ksw_workspace_bytes(int n, const int64_t *qoff, const int64_t *toff, const int32_t *w);
/* restrict dispatch: 0 = automatic (default), 1 = force the generic LDS kernel, 2 = wave kernel only (fails
 * with GDIET_E_PARAM for alignments it cannot take).  For tests and A/B measurements. */
int gdiet_hip_set_kernel_mode(gdiet_ctx *ctx, int mode);
/* 1 (default): a 64-lane DP launch with more alignments than the GPU holds at once is issued as a head launch (the longest
 * alignments, one per resident wavefront slot) and a tail launch on a second stream, so that the head's backtrack overlaps the
 * tail's DP.  0: one launch, one backtrack (what bench.py uses for its roofline passes: one kernel, one duration). */
int gdiet_hip_set_dp_split(gdiet_ctx *ctx, int on);
/* Wavefronts per SIMD the 64-lane DP kernel (long reads) is launched for: 5 (default: 96 VGPRs, the best throughput of a full pipeline)
 * or 4 (a fifth of every SIMD's registers stays free, so the seeding / voting kernels of the NEXT batch run beside the
 * DP kernel instead of trickling through as its wavefronts retire -- two batches in flight then keep the DP kernels back to back, at
 * two instead of three step times of latency; bench.py latency_mode).  Same arithmetic, same results. */
int gdiet_hip_set_dp_waves(gdiet_ctx *ctx, int waves_per_simd);
/* average device time (ms, HIP events on the launch stream) of the DP kernel(s) and of the backtrack kernel of the
 * most recent *_dev / host batch; only valid after the stream has been synchronised. */
int gdiet_hip_last_kernel_ms(gdiet_ctx *ctx, float *dp_ms, float *backtrack_ms);
/* The shader clock the 64-lane DP kernel sustained, from its own wavefronts: each stamps s_memtime (shader clock) and s_memrealtime
 * (constant 100 MHz) around its DP rows.  Median and minimum over the wavefronts of the most recent launches of the process, and the
 * median duration of one wavefront's rows.  (Says whether the kernel ran throttled: profiles/r03_clock.md.) */
int gdiet_hip_last_dp_clock(gdiet_ctx *ctx, double *sclk_mhz_median, double *sclk_mhz_min, double *wavefront_ms_median);
/* work of the most recent DP launch: DP cells = sum (qlen+tlen-1)*min(w+1,qlen,tlen), and the algorithmic bytes of
 * SURVEY.md 8d = cells + (qlen+tlen) + qlen + ceil(tlen/2) per alignment (what bench.py's roofline divides by the kernel time) */
int gdiet_hip_last_dp_work(const gdiet_ctx *ctx, uint64_t *cells, uint64_t *alg_bytes);

/* ---- B1: the per-read mapping path for a whole batch of reads (LongReads variant; ShortReads variant with MM_F_SR) ----
 * Replaces step 1 of worker_pipeline -- kt_for(n_threads, worker_for, ...) -> mm_map_frag() per read
 * (LR/map.c:2132-2137 -> :1975-2022 -> :1273-1940): sketch2/get_shift/sketch3, seed filter + collect, hit sort,
 * vote/vote_2, candidate geometry, exact-match / ksw_extd2 / backtrack, mm_update_extra, concatenate_cigars,
 * mm_set_sam_params.  The seeding/voting stages and the DP run on the GPU; geometry and CIGAR post-processing
 * run on host threads inside this library (they need the float/double comparisons and the bug-compatible
 * control flow of the reference).
 */

/* device-resident index: the flat mirror of mm_idx_t (LR/minimap.h:79-96, LR/index.c:29-34,84-100) */
typedef struct gdiet_index gdiet_index;

/* Build the index from nt ASCII sequences with this library's own builder (same minimizers, same position order
 * as mm_idx_gen of GDiet_avx: LR/index.c:306-412, LR/sketch.c:156/1577) and upload it.  pattern/pattern_len = -Z/-W.
 * Limits, refused with GDIET_E_PARAM: 1 <= k <= 28 as in the reference (LR/main.c), and 1 <= w <= 64 -- the reference accepts
 * w < 256; the winnowing windows of the sketch kernels live in registers / LDS sized for 64 (every preset uses w <= 19).  For
 * w in {2, 3, 5, 6} the reference's own scalar and AVX-512 sketches disagree; this library follows the scalar one. */
int gdiet_hip_index_build(gdiet_ctx *ctx, gdiet_index **idx, int n_seq, const char *const *names,
                          const char *const *seqs, const uint32_t *lens, int k, int w, const char *pattern,
                          int pattern_len, int n_threads);
/* Upload an index that the caller already holds in flat form -- what a reference-side stub produces by walking
 * mm_idx_t::B[]: keys[i] = minimizer hash (x>>8), cnt[i] its number of occurrences, pos = the
 * occurrence lists (y values, each list sorted ascending) concatenated in key order; S = mm_idx_t::S. */
int gdiet_hip_index_import(gdiet_ctx *ctx, gdiet_index **idx, int k, int w, const char *pattern, int pattern_len,
                           int n_seq, const char *const *names, const uint32_t *lens, const uint64_t *offsets,
                           const uint32_t *S, uint64_t n_keys, const uint64_t *keys, const uint32_t *cnt,
                           const uint64_t *pos);
/* the inverse of gdiet_hip_index_import: sizes first (array arguments NULL), then the arrays (caller-allocated: n_keys keys and
 * counts, n_pos positions, n_S_words words of S, n_seq offsets).  What a maintainer needs to write the index back into an
 * mm_idx_t / .mmi (LR/index.c:428-470 mm_idx_dump) or to check it against mm_idx_get. */
int gdiet_hip_index_export(const gdiet_index *idx, uint64_t *n_keys, uint64_t *n_pos, uint64_t *n_S_words, uint64_t *keys,
                           uint32_t *cnt, uint64_t *pos, uint32_t *S, uint64_t *offsets);
/* Read an .mmi file written by the reference (mm_idx_dump, LR/index.c:480-517; `GDiet -d`) and upload it.  The file does not store
 * the pattern, so -Z / -W are given again, exactly as on the reference's command line. */
int gdiet_hip_index_load_mmi(gdiet_ctx *ctx, gdiet_index **idx, const char *path, const char *pattern, int pattern_len);
/* Write the .mmi file mm_idx_dump (LR/index.c:480-517) writes for this index, byte for byte: the position arrays in the order
 * worker_post fills them and every bucket's entries in the slot order of the reference's khash table (its resize / put sequence is
 * emulated: LR/index.c:216-264, LR/khash.h:199-330).  bucket_bits = mm_idxopt_t::bucket_bits (14 by default). */
int gdiet_hip_index_dump_mmi(gdiet_ctx *ctx, const gdiet_index *idx, const char *path, int bucket_bits);
void gdiet_hip_index_destroy(gdiet_ctx *ctx, gdiet_index *idx);
/* mm_idx_cal_max_occ (LR/index.c:190-210) */
int32_t gdiet_hip_index_cal_max_occ(const gdiet_index *idx, float frac);
uint64_t gdiet_hip_index_n_keys(const gdiet_index *idx);

/* the fields of mm_mapopt_t (LR/minimap.h:145-214) this path reads; fill them from the reference's struct */
typedef struct {
	int64_t flag;                 /* MM_F_* bits; only NO_PRINT_2ND, SR (selects the variant), FRAG_MODE, FOR_ONLY, REV_ONLY are interpreted */
	int32_t a, b, q, e, q2, e2;
	uint32_t bw;
	int32_t min_dp_max, best_n;
	float q_occ_frac;
	int32_t mid_occ, max_max_occ, occ_dist, max_frag_len;
	uint32_t vt_dis, vt_nb_loc;
	float vt_cov, vt_f, vt_df1, vt_df2;
	uint32_t max_max_gap, max_min_gap;
	float max_seeds;
	/* read only when flag has MM_F_SR: the ShortReads variant of mm_map_frag (SR/map.c:586-984; SR/minimap.h:149-150,196-197) */
	float min_cnt, rec_threshold_frac;  /* -n FLOAT1,FLOAT2 */
	float bw_frac;                      /* -r FLOAT,INT,INT: band = vote distance = clamp(qlen*bw_frac, bw_min, bw_max) */
	int32_t bw_min, bw_max;
	int32_t AF_max_loc;                 /* --AF_max_loc */
} gdiet_mapopt_t;

/* one alignment record: mm_reg1_t + mm_extra_t (LR/minimap.h:105-131) flattened */
typedef struct {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, mlen, blen;
	uint32_t mapq, rev, sam_pri;
	int32_t dp_score, dp_max;
	uint32_t n_ambi, n_cigar;
	uint32_t *cigar;              /* points into the allocation the record belongs to */
} gdiet_reg_t;

/* Map n_reads single-segment reads (ASCII sequences).  On return regs[i] is a malloc'd array of n_regs[i] records
 * (NULL when n_regs[i] == 0, as LR/map.c:1915), in the order mm_map_frag leaves them.  The records and CIGARs of a batch share
 * a few large allocations: release them with ONE gdiet_hip_free_regs call on the same n_reads / n_regs / regs arrays, never with
 * free() on a single regs[i]. */
int gdiet_hip_map_batch(gdiet_ctx *ctx, const gdiet_index *idx, const gdiet_mapopt_t *opt, int n_reads,
                        const char *const *seqs, const int32_t *lens, int32_t *n_regs, gdiet_reg_t **regs);
void gdiet_hip_free_regs(int n_reads, int32_t *n_regs, gdiet_reg_t **regs);

/* B2, the per-read level of the boundary: mm_map_frag's call shape (LR/minimap.h:390, LR/map.c:1273-1940) for ONE fragment of n_segs
 * segments.  As in the reference only segment 0 is sketched and aligned (qlen_sum aside: paired-end input is effectively unsupported
 * there, SURVEY bug-compatibility item 7): n_regs[0] / regs[0] receive its records, the other segments none.  Release with
 * gdiet_hip_free_regs(n_segs, n_regs, regs).  A wavefront-per-alignment device path gains nothing from one read at a time: this entry
 * exists for callers that are written against mm_map_frag; throughput lives in gdiet_hip_map_batch / _submit. */
int gdiet_hip_map_frag(gdiet_ctx *ctx, const gdiet_index *idx, const gdiet_mapopt_t *opt, int n_segs, const int32_t *qlens,
                       const char *const *seqs, int32_t *n_regs, gdiet_reg_t **regs);

/* B4, the sketch / seed level of the boundary (LR/mmpriv.h:65-76), for a batch of reads: what mm_sketch2 + mm_get_shift (the pattern phase
 * shift[i]), mm_sketch3 (tmp_extracted_len[i]), mm_seed_mz_flt (n_mv[i]: minimizers left after the query-side filter) and
 * mm_collect_matches2 (the kept seeds: mm_seed_t reduced to n = occurrences in the index and q_pos = lastPos<<1 | strand, in sketch
 * order; their occurrence lists = the y values mm_idx_get returns, rid<<32 | lastPos<<1 | strand) produce for every read -- the output of
 * the seeding kernel (LR/map.c:1296-1325).  Read i owns seeds[seed_off[i] .. seed_off[i+1]) and occ[occ_off[i] .. occ_off[i+1]) (the
 * lists of its seeds back to back).  *seeds and *occ are malloc'd: free() them. */
typedef struct { uint32_t n, q_pos; } gdiet_seed_t;
int gdiet_hip_seed_batch(gdiet_ctx *ctx, const gdiet_index *idx, const gdiet_mapopt_t *opt, int n_reads, const char *const *seqs,
                         const int32_t *lens, int32_t *shift, uint32_t *tmp_extracted_len, uint32_t *n_mv, int64_t *seed_off,
                         int64_t *occ_off, gdiet_seed_t **seeds, uint64_t **occ);

/* Reads of the most recent map call on this context that were given up on and came back unmapped (n_regs = 0) while the rest of their
 * batch was mapped: a candidate's DP box lay outside its read / contig or had wrapped to an absurd size -- input on which the
 * reference reads stale heap memory (LR/map.c:1654-1806 with mm_idx_getseq2 returning short / -1), i.e. has no defined result.
 * last_call / total (since gdiet_hip_init) / what (a one-line description of the last one; valid until the next map call) may be NULL. */
int gdiet_hip_map_failed_reads(const gdiet_ctx *ctx, int64_t *last_call, int64_t *total, const char **what);

/* ---- single-process multi-GPU fan-out of one mini-batch -----------------------------------------------------------
 * What step 1 of worker_pipeline calls when the process owns several GPUs (LR/map.c:2132-2137: kt_for over the fragments of the
 * mini-batch; step 2 prints in input order, LR/kthread.c:97-121): the mini-batch is cut into n_ctx CONTIGUOUS read ranges of equal DP
 * cost (gdiet_hip_read_ranges_by_cost), range k is mapped by ctxs[k] against idxs[k] (the index replicated per device: build / import
 * / load it once per context) on a caller thread of its own, and every range writes its records straight into n_regs / regs at its
 * offset -- the arrays come back exactly as gdiet_hip_map_batch on one context would fill them (same records, same order; release
 * them with ONE gdiet_hip_free_regs call).  No collective: reads are independent.  Give every context its share of the host's CPUs
 * first (gdiet_hip_set_host_threads(ctx, gdiet_hip_effective_cpus() / n_ctx)).  Errors are joined: the code of the first failing
 * range is returned, its text (with device and read range) is gdiet_hip_strerror(ctxs[0]), and no records are handed back. */
int gdiet_hip_map_batch_multi(int n_ctx, gdiet_ctx *const *ctxs, const gdiet_index *const *idxs, const gdiet_mapopt_t *opt,
                              int n_reads, const char *const *seqs, const int32_t *lens, int32_t *n_regs, gdiet_reg_t **regs);
/* bounds[0..n_parts]: contiguous ranges of equal DP cost, a read costing (2 len - 1) * min(band + 1, len) cells (balance by
 * "sum l_seq^2-ish DP cost, not read count").  Host arithmetic only. */
int gdiet_hip_read_ranges_by_cost(int n_reads, const int32_t *lens, int n_parts, int32_t band, int32_t *bounds);

/* Two-step form for measurements: stage a batch in HBM once, then map it (repeatedly).  Only the second call belongs
 * to a timed region whose inputs are "already resident in HBM". */
typedef struct gdiet_read_batch gdiet_read_batch;
int gdiet_hip_batch_upload(gdiet_ctx *ctx, gdiet_read_batch **batch, int n_reads, const char *const *seqs, const int32_t *lens);
int gdiet_hip_map_uploaded(gdiet_ctx *ctx, const gdiet_index *idx, const gdiet_mapopt_t *opt, gdiet_read_batch *batch,
                           int32_t *n_regs, gdiet_reg_t **regs);
void gdiet_hip_batch_destroy(gdiet_ctx *ctx, gdiet_read_batch *batch);
/* Several batches in flight: _submit starts the whole path for a resident batch on a lane of its own (stream, scratch) and
 * returns a ticket, _wait joins it (results in the n_regs / regs arrays given to _submit).  Up to
 * gdiet_hip_set_inflight() tickets (default 2, at most 8) may be open; wait for them in submission order.  The latency-bound
 * stages of one batch overlap the DP kernel of another, exactly as the reference's kt_pipeline overlaps the steps of consecutive
 * mini-batches (LR/map.c:2094-2170); results are those of gdiet_hip_map_uploaded.  The lanes share one backtrace arena (~34 MB
 * per 15 kbp alignment: two whole-batch arenas would not fit in HBM), so their DP stages take turns -- unless a batch's backtrace
 * is small enough for every lane to hold one of its own (gdiet_hip_set_inflight sets the limit: 70 % of HBM / n, or
 * GDIET_LANE_ARENA_GB), in which case the DP kernels of the batches in flight overlap as well. */
typedef struct gdiet_map_ticket gdiet_map_ticket;
int gdiet_hip_set_inflight(gdiet_ctx *ctx, int n);
int gdiet_hip_map_submit(gdiet_ctx *ctx, const gdiet_index *idx, const gdiet_mapopt_t *opt, gdiet_read_batch *batch,
                         int32_t *n_regs, gdiet_reg_t **regs, gdiet_map_ticket **ticket);
int gdiet_hip_map_wait(gdiet_ctx *ctx, gdiet_map_ticket *ticket);
/* seconds spent in the stages of the most recent map call: [0] seed kernel, [1] vote kernel, [2] host geometry,
 * [3] gather + DP + backtrack kernels, [4] host post-processing, [5] transfers/other */
int gdiet_hip_map_stage_seconds(const gdiet_ctx *ctx, double out[6]);
/* software-pipeline depth of gdiet_hip_map_uploaded / gdiet_hip_map_batch: the batch is cut into slices that run the whole
 * chain on `n` independent lanes (stream + workspace + host threads each), overlapping the latency-bound stages of one slice
 * with the DP kernel of the others.  1 (default) = no pipelining.  Results do not depend on it. */
int gdiet_hip_set_map_lanes(gdiet_ctx *ctx, int n);
/* number of host threads used for geometry / CIGAR post-processing (default: gdiet_hip_effective_cpus(), at most 64) */
int gdiet_hip_set_host_threads(gdiet_ctx *ctx, int n);
/* CPUs the process may really use: min(hardware threads, affinity mask, cgroup CPU quota).  The reference sizes its worker pool
 * with -t (LR/main.c:85 n_threads); a caller that derives -t from the machine should use this figure. */
int gdiet_hip_effective_cpus(void);

/* One SAM record exactly as mm_write_sam3 prints it for a single-segment read (LR/format.c:412-599); reg_idx < 0 writes
 * the unmapped record.  Returns the number of bytes needed (excluding the terminating NUL); writes at most cap bytes. */
size_t gdiet_hip_sam_record(const gdiet_index *idx, const char *qname, const char *seq, const char *qual, int32_t l_seq,
                            const gdiet_reg_t *regs, int32_t n_regs, int32_t reg_idx, int64_t opt_flag, char *buf, size_t cap);

/* All records of a batch, in input order, one per line (the body of a SAM file without the header), formatted on the context's
 * host threads -- step 2 of worker_pipeline (LR/map.c:2139-2170) for a whole mini-batch.  *out is malloc'd (free() it); returns
 * its length.  quals may be NULL, and so may its entries. */
size_t gdiet_hip_sam_batch(gdiet_ctx *ctx, const gdiet_index *idx, int n_reads, const char *const *qnames, const char *const *seqs,
                           const char *const *quals, const int32_t *lens, const int32_t *n_regs, gdiet_reg_t *const *regs,
                           int64_t opt_flag, char **out);

/* The same into a buffer the caller keeps from mini-batch to mini-batch: *buf / *cap start as NULL / 0, are realloc'd when too small, and
 * are free()d by the caller at the end.  (The SAM text of a long-read mini-batch is ~150 MB; a fresh allocation of that size costs its
 * page faults every time.)  Returns the length of the text. */
size_t gdiet_hip_sam_batch_into(gdiet_ctx *ctx, const gdiet_index *idx, int n_reads, const char *const *qnames, const char *const *seqs,
                                const char *const *quals, const int32_t *lens, const int32_t *n_regs, gdiet_reg_t *const *regs,
                                int64_t opt_flag, char **buf, size_t *cap);

/* The same for PAF output: mm_write_paf3 (LR/format.c:326-367) as step 2 prints it when MM_F_OUT_SAM is off (LR/map.c:2163-2185).
 * opt_flag: MM_F_OUT_CG (0x20, `-c`) adds the cg:Z: tag, MM_F_PAF_NO_HIT (0x8000000, --paf-no-hit) the lines of unmapped reads,
 * MM_F_NO_PRINT_2ND drops secondary records, MM_F_QSTRAND is honoured.  *out is malloc'd; returns its length. */
size_t gdiet_hip_paf_batch(gdiet_ctx *ctx, const gdiet_index *idx, int n_reads, const char *const *qnames, const int32_t *lens,
                           const int32_t *n_regs, gdiet_reg_t *const *regs, int64_t opt_flag, char **out);

/* Read input, step 0 of worker_pipeline (LR/map.c:2095-2131): FASTA / FASTQ, plain or gzip ("-" = stdin), one mini-batch per
 * call.  Replaces mm_bseq_open / mm_bseq_read3 / mm_bseq_close (LR/bseq.c:38-58, 80-121) with the same record grammar
 * (kseq_read, LR/kseq.h:191-232: multi-line records, names up to the first white space, the rest of the header as comment,
 * "\r\n" line ends, U -> T) and the same batching rule (records until their lengths sum to chunk_size; in fragment mode the mates
 * of the last read as well).  The arrays and strings belong to the reader and stay valid until the next call on it; comments[i]
 * and quals[i] are NULL where the reference would store none; with_qual / with_comment of the FIRST call hold for the whole file
 * (the reader parses ahead).  *n_reads == 0: end of input.  Returns GDIET_OK, or
 * GDIET_W_TRUNCATED when the input ended in a malformed record (the batch holds the records before it; the reference prints a
 * warning and goes on likewise), or GDIET_E_PARAM (cannot open / read error). */
#define GDIET_W_TRUNCATED 1
typedef struct gdiet_fastx gdiet_fastx;
int gdiet_hip_fastx_open(gdiet_fastx **fx, const char *path);
int gdiet_hip_fastx_read(gdiet_fastx *fx, int64_t chunk_size, int with_qual, int with_comment, int frag_mode, int32_t *n_reads,
                         const char *const **names, const char *const **comments, const char *const **seqs,
                         const char *const **quals, const int32_t **lens);
/* Several mini-batches in flight: take the batch the last gdiet_hip_fastx_read returned out of the reader.  Its arrays and strings
 * (the pointers that call handed out) then stay valid until gdiet_hip_fastx_batch_free, whatever is read in the meantime. */
typedef struct gdiet_fastx_batch gdiet_fastx_batch;
gdiet_fastx_batch *gdiet_hip_fastx_detach(gdiet_fastx *fx);
void gdiet_hip_fastx_batch_free(gdiet_fastx_batch *b);
/* parser threads (default 1).  Blocks of the file are cut at places that look like the start of a four-line FASTQ record and the
 * stretches parsed side by side; a stretch that does not end exactly where the next one began proves the cut wrong, and the rest of
 * the block is parsed again in sequence -- so the records are those of the sequential grammar whatever the input looks like. */
int gdiet_hip_fastx_set_threads(gdiet_fastx *fx, int n);
void gdiet_hip_fastx_close(gdiet_fastx *fx);

#ifdef __cplusplus
}
#endif
#endif
