/* Reference-side binding of libgdiet_hip.so (level B1): what a maintainer of Genome-on-Diet adds to the tree.
 * Compiled TOGETHER WITH THE REFERENCE'S OWN SOURCES (its minimap.h / mmpriv.h / khash.h / bseq.h) by oracle/build_ref_hip.py,
 * which copies the reference to a temporary directory, inserts the two call sites below into map.c and main.c and links the
 * result against the library with -Wl,--no-undefined -> oracle/_ref/gdiet_lr_hip.  With GDIET_HIP=1 in the environment that
 * binary runs the reference's own main(), reader and mm_write_sam3 around this library's per-read path; without it, it is GDiet_avx.
 *
 * GDIET_HIP_DEVICES=0,1,2,3 (default "0"): the HIP devices the process maps on -- one context and one copy of the index per entry, every
 * mini-batch cut into contiguous read ranges of equal DP cost (gdiet_hip_map_batch_multi).  A device may be listed
 * twice (two contexts on one GPU: how the one-GPU test box exercises the fan-out).
 *
 *   main.c, after mm_mapopt_update():   if (gdiet_glue_enabled()) gdiet_glue_index(mi, &opt);
 *   map.c, step 1 of worker_pipeline:   if (gdiet_glue_enabled()) gdiet_glue_map_step(s->n_frag, s->seg_off, s->n_seg, s->seq, s->n_reg, s->reg, p->opt);
 *                                       else kt_for(p->n_threads, worker_for, in, s->n_frag);          (LR/map.c:2132-2137)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#include "bseq.h"
#include "khash.h"
#include "gdiet_hip.h"

/* the index's hash table type, as LR/index.c:20-34 declares it (private to index.c there) */
#define idx_hash(a) ((a) >> 1)
#define idx_eq(a, b) ((a) >> 1 == (b) >> 1)
KHASH_INIT(idx, uint64_t, uint64_t, 1, idx_hash, idx_eq)
typedef khash_t(idx) idxhash_t;
typedef struct mm_idx_bucket_s {
	mm128_v a;
	int32_t n;
	uint64_t *p;
	void *h;
} mm_idx_bucket_t;

#define GLUE_MAX_DEV 16
static gdiet_ctx *g_ctx[GLUE_MAX_DEV];
static gdiet_index *g_idx[GLUE_MAX_DEV];
static int g_n;
static long long g_failed_reads;

#ifdef PROFILE
#include "profile.h" /* the reference's [PROFILING] counters (LR/profile.h:10-21), fed from the library's stage clocks below */
#endif

int gdiet_glue_enabled(void)
{
	static int on = -1;
	if (on < 0) {
		const char *e = getenv("GDIET_HIP");
		on = e && atoi(e) != 0;
	}
	return on;
}

static void die(const char *what)
{
	fprintf(stderr, "[gdiet_hip] %s: %s\n", what, g_ctx[0] ? gdiet_hip_strerror(g_ctx[0]) : "no context");
	exit(1); /* set-up failures only (no device, no memory for the index): no CPU fallback once the GPU path was asked for */
}

static void glue_open(void)
{
	const char *e = getenv("GDIET_HIP_DEVICES");
	int dev[GLUE_MAX_DEV], n = 0, k;
	if (g_n) return;
	if (e && *e) {
		char *end;
		while (*e && n < GLUE_MAX_DEV) {
			dev[n++] = (int)strtol(e, &end, 10);
			if (end == e) { fprintf(stderr, "[gdiet_hip] GDIET_HIP_DEVICES: cannot parse '%s'\n", e); exit(1); }
			e = *end == ',' ? end + 1 : end;
		}
	}
	if (n == 0) dev[n++] = 0;
	for (k = 0; k < n; ++k)
		if (gdiet_hip_init(&g_ctx[k], dev[k])) {
			fprintf(stderr, "[gdiet_hip] gdiet_hip_init(device %d): is there a gfx950 device?\n", dev[k]);
			exit(1);
		}
	g_n = n;
	if (n > 1) { /* the contexts share the host's CPUs */
		int t = gdiet_hip_effective_cpus() / n;
		for (k = 0; k < n; ++k) gdiet_hip_set_host_threads(g_ctx[k], t > 1 ? t : 1);
	}
}

/* mm_idx_t -> gdiet_hip_index_import: walk the buckets' hash tables (LR/index.c:84-100 is the lookup this inverts) */
void gdiet_glue_index(const mm_idx_t *mi, const mm_mapopt_t *opt)
{
	uint64_t n_keys = 0, n_pos = 0, j = 0, o = 0;
	int b, d;
	uint32_t i;
	glue_open();
	for (d = 0; d < g_n; ++d)
		if (g_idx[d]) gdiet_hip_index_destroy(g_ctx[d], g_idx[d]), g_idx[d] = 0;
	for (b = 0; b < 1 << mi->b; ++b) {
		idxhash_t *h = (idxhash_t *)mi->B[b].h;
		khint_t k;
		if (!h) continue;
		for (k = 0; k < kh_end(h); ++k)
			if (kh_exist(h, k)) ++n_keys, n_pos += (kh_key(h, k) & 1) ? 1 : (uint32_t)kh_val(h, k);
	}
	{
		uint64_t *keys = (uint64_t *)malloc(8 * (n_keys + 1)), *pos = (uint64_t *)malloc(8 * (n_pos + 1));
		uint32_t *cnt = (uint32_t *)malloc(4 * (n_keys + 1));
		const char **names = (const char **)malloc(sizeof(char *) * mi->n_seq);
		uint32_t *lens = (uint32_t *)malloc(4 * mi->n_seq);
		uint64_t *offs = (uint64_t *)malloc(8 * mi->n_seq);
		for (b = 0; b < 1 << mi->b; ++b) {
			idxhash_t *h = (idxhash_t *)mi->B[b].h;
			khint_t k;
			if (!h) continue;
			for (k = 0; k < kh_end(h); ++k)
				if (kh_exist(h, k)) {
					const uint64_t key = kh_key(h, k), val = kh_val(h, k);
					keys[j] = (key >> 1) << mi->b | (uint64_t)b; /* undo ">> mi->b << 1" of LR/index.c:248; the bucket is the low b bits */
					if (key & 1) cnt[j] = 1, pos[o++] = val;     /* singleton: the value is y itself (LR/index.c:251-253) */
					else {
						cnt[j] = (uint32_t)val;
						memcpy(pos + o, mi->B[b].p + (val >> 32), 8 * (size_t)(uint32_t)val);
						o += (uint32_t)val;
					}
					++j;
				}
		}
		for (i = 0; i < mi->n_seq; ++i) names[i] = mi->seq[i].name, lens[i] = mi->seq[i].len, offs[i] = mi->seq[i].offset;
		for (d = 0; d < g_n; ++d) /* the index is replicated: one device copy per context */
			if (gdiet_hip_index_import(g_ctx[d], &g_idx[d], mi->k, mi->w, opt->pattern, opt->pattern_len, (int)mi->n_seq, names, lens, offs, mi->S, n_keys,
			                           keys, cnt, pos)) {
				fprintf(stderr, "[gdiet_hip] gdiet_hip_index_import (context %d): %s\n", d, gdiet_hip_strerror(g_ctx[d]));
				exit(1);
			}
		free(keys), free(pos), free(cnt), free((void *)names), free(lens), free(offs);
	}
}

static gdiet_mapopt_t glue_opt(const mm_mapopt_t *o)
{
	gdiet_mapopt_t g;
	memset(&g, 0, sizeof g);
	g.flag = o->flag;
	g.a = o->a, g.b = o->b, g.q = o->q, g.e = o->e, g.q2 = o->q2, g.e2 = o->e2;
	g.min_dp_max = o->min_dp_max, g.best_n = o->best_n, g.q_occ_frac = o->q_occ_frac;
	g.mid_occ = o->mid_occ, g.max_max_occ = o->max_max_occ, g.occ_dist = o->occ_dist, g.max_frag_len = o->max_frag_len;
	g.max_seeds = o->max_seeds;
#ifdef GDIET_SHORTREADS /* GDiet-ShortReads/minimap.h */
	g.min_cnt = o->min_cnt, g.rec_threshold_frac = o->rec_threshold_frac;
	g.bw_frac = o->bw_frac, g.bw_min = o->bw_min, g.bw_max = o->bw_max, g.AF_max_loc = o->AF_max_loc;
#else /* GDiet-LongReads/minimap.h */
	g.bw = o->bw, g.vt_dis = o->vt_dis, g.vt_nb_loc = o->vt_nb_loc, g.vt_cov = o->vt_cov, g.vt_f = o->vt_f;
	g.vt_df1 = o->vt_df1, g.vt_df2 = o->vt_df2, g.max_max_gap = o->max_max_gap, g.max_min_gap = o->max_min_gap;
#endif
	return g;
}

/* step 1 of worker_pipeline for one mini-batch: fills n_reg[] / reg[] exactly as kt_for(worker_for) does (libc allocations that
 * step 2 frees: LR/map.c:2163-2167).  Returns 0, or non-zero when the library could not map the batch: the caller then runs
 * kt_for(worker_for) on it -- the reference's own path -- and the run goes on ("shim returns non-zero => caller falls back
 * to CPU path"); a read the library gave up on (degenerate DP box) comes back unmapped and is counted. */
int gdiet_glue_map_step(int n_frag, const int *seg_off, const int *n_seg, const mm_bseq1_t *seq, int *n_reg, mm_reg1_t **reg, const mm_mapopt_t *opt)
{
	int i, j;
	const char **seqs = (const char **)malloc(sizeof(char *) * (n_frag + 1));
	int32_t *lens = (int32_t *)malloc(4 * (n_frag + 1)), *n_regs = (int32_t *)calloc(n_frag + 1, 4);
	gdiet_reg_t **regs = (gdiet_reg_t **)calloc(n_frag + 1, sizeof(*regs));
	gdiet_mapopt_t go = glue_opt(opt);
	(void)n_seg; /* single-segment reads: the path maps segment 0 of every fragment, as mm_map_frag does (SURVEY bug-compat item 7) */
	for (i = 0; i < n_frag; ++i) seqs[i] = seq[seg_off[i]].seq, lens[i] = seq[seg_off[i]].l_seq;
	if (gdiet_hip_map_batch_multi(g_n, g_ctx, (const gdiet_index *const *)g_idx, &go, n_frag, seqs, lens, n_regs, regs)) {
		fprintf(stderr, "[gdiet_hip] gdiet_hip_map_batch_multi failed for a mini-batch of %d reads (%s): mapping it with the reference's worker_for\n", n_frag,
		        gdiet_hip_strerror(g_ctx[0]));
		free(regs), free(n_regs), free(lens), free((void *)seqs);
		return 1;
	}
	{
		long long last = 0;
		const char *what = 0;
		gdiet_hip_map_failed_reads(g_ctx[0], (int64_t *)&last, 0, &what);
		if (last) g_failed_reads += last, fprintf(stderr, "[gdiet_hip] %s\n", what);
	}
#ifdef PROFILE
	{ /* [PROFILING] lines (LR/profile.h:10-21, LR/main.c:685): the counters are nanoseconds summed over worker threads in the reference; the
	   * GPU path has one clock per stage and mini-batch: seed kernel (sketch2 + get_shift + sketch3 + filter + lookup; the pattern
	   * alignment is part of that kernel and is not timed apart), vote kernel, gather + DP + backtrack (+ the host stages around them) */
		int d;
		for (d = 0; d < g_n; ++d) {
			double st[6];
			if (gdiet_hip_map_stage_seconds(g_ctx[d], st)) continue;
			atomic_fetch_add(&pf_seeding, (uint_least64_t)(st[0] * 1e9));
			atomic_fetch_add(&pf_voting, (uint_least64_t)(st[1] * 1e9));
			atomic_fetch_add(&pf_sequence_alignment, (uint_least64_t)((st[2] + st[3] + st[4]) * 1e9));
		}
	}
#endif
	for (i = 0; i < n_frag; ++i) {
		const int off = seg_off[i];
		n_reg[off] = n_regs[i];
		reg[off] = n_regs[i] ? (mm_reg1_t *)calloc(n_regs[i], sizeof(mm_reg1_t)) : 0; /* LR/map.c:1915: NULL when unmapped */
		for (j = 0; j < n_regs[i]; ++j) {
			const gdiet_reg_t *g = &regs[i][j];
			mm_reg1_t *r = &reg[off][j];
			uint32_t capacity = g->n_cigar + sizeof(mm_extra_t) / 4;
			kroundup32(capacity);
			r->id = g->id, r->cnt = g->cnt, r->rid = g->rid, r->score = g->score, r->qs = g->qs, r->qe = g->qe, r->rs = g->rs, r->re = g->re;
			r->parent = g->parent, r->subsc = g->subsc, r->mlen = g->mlen, r->blen = g->blen;
			r->mapq = g->mapq, r->rev = g->rev, r->sam_pri = g->sam_pri;
			r->p = (mm_extra_t *)calloc(capacity, 4); /* LR/map.c:1830-1836 */
			r->p->capacity = capacity, r->p->dp_score = g->dp_score, r->p->dp_max = g->dp_max;
			r->p->n_ambi = g->n_ambi, r->p->n_cigar = g->n_cigar;
			memcpy(r->p->cigar, g->cigar, 4 * (size_t)g->n_cigar);
		}
	}
	gdiet_hip_free_regs(n_frag, n_regs, regs);
	free(regs), free(n_regs), free(lens), free((void *)seqs);
	return 0;
}

void gdiet_glue_close(void)
{
	int d;
	if (g_failed_reads) fprintf(stderr, "[gdiet_hip] %lld read(s) left unmapped by the GPU path (degenerate DP boxes)\n", g_failed_reads);
	for (d = 0; d < g_n; ++d) {
		if (g_idx[d]) gdiet_hip_index_destroy(g_ctx[d], g_idx[d]), g_idx[d] = 0;
		if (g_ctx[d]) gdiet_hip_destroy(g_ctx[d]), g_ctx[d] = 0;
	}
	g_n = 0;
}
