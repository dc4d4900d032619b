// B1: batch driver of the per-read mapping path.  Included at the end of gdiet_hip.hip (needs gdiet_ctx, GD_HIP, gd_grow).
//   host threads : encode reads, candidate linking + DP boxes (gd_lr_link_and_boxes), post-processing (gd_lr_finish)
//   device       : map_seed_kernel -> map_vote_kernel -> map_gather_kernel -> ksw batch (exact-match, DP, backtrack)
#pragma once
#include <atomic>
#include <memory>
#include <cstddef>
#include <chrono>
#include <thread>
#include <mutex>
#include "map_host.h"
#include "map_index.h"
#include "map_kernels.hip.h"

struct gdiet_index {
	GdIndex h;
	void *d_tkey = nullptr, *d_tval = nullptr, *d_pos = nullptr, *d_S = nullptr;
	GdIdxView dview;
	// device-built index (map_index_dev.hip.h): the table exists on the device only until someone asks for the host copies
	bool host_tables = true;
	uint32_t *d_cnt_sorted = nullptr; // occurrence counts of the keys, ascending (mm_idx_cal_max_occ)
	uint64_t n_pos = 0;
	// contig table on the device (lengths, offsets into S), made on first use by the device-side box stage of the ShortReads variant
	uint32_t *d_seq_len = nullptr;
	uint64_t *d_seq_off = nullptr;
	std::mutex seq_mu;
};

struct gdiet_read_batch {
	int n = 0;
	std::vector<int64_t> roff;          // n+1
	std::vector<uint8_t> enc;           // nt4, forward strand, all reads packed (host copy for post-processing)
	void *d_reads = nullptr, *d_roff = nullptr;
};

#include "map_index_dev.hip.h"

#include "host_pool.h"
#include "nt4_encode.h"

static void gd_pool_free(void *pool) { delete (GdPool *)pool; }

static GdPool *gd_pool(gdiet_ctx *ctx)
{
	gdiet_ctx *owner = ctx->parent ? ctx->parent : ctx; // an async lane works in its parent's pool
	static std::mutex create_mu; // (taken on every call: a handful per batch; a plain pointer must not be double-checked without it)
	std::lock_guard<std::mutex> lk(create_mu);
	if (!owner->pool) owner->pool = new GdPool();
	return (GdPool *)owner->pool;
}

template <class F> static void gd_parallel_for(gdiet_ctx *ctx, int n_threads, int n, F f) { gd_pool(ctx)->run(n_threads, n, f); }

static double gd_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int gd_index_upload(gdiet_ctx *ctx, gdiet_index *ix)
{
	GdIndex &h = ix->h;
	auto up = [&](void **d, const void *src, size_t bytes) -> int {
		if (bytes == 0) bytes = 8;
		GD_HIP(hipMalloc(d, bytes));
		if (src) GD_HIP(hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice));
		return GDIET_OK;
	};
	int rc;
	if ((rc = up(&ix->d_tkey, h.tkey.data(), h.tkey.size() * 8))) return rc;
	if ((rc = up(&ix->d_tval, h.tval.data(), h.tval.size() * 8))) return rc;
	if ((rc = up(&ix->d_pos, h.pos.empty() ? nullptr : h.pos.data(), h.pos.size() * 8))) return rc;
	if ((rc = up(&ix->d_S, h.S.empty() ? nullptr : h.S.data(), h.S.size() * 4 + 8))) return rc;
	ix->dview.k = h.k, ix->dview.w = h.w, ix->dview.tbits = h.tbits;
	ix->dview.tkey = (const uint64_t *)ix->d_tkey, ix->dview.tval = (const uint64_t *)ix->d_tval, ix->dview.pos = (const uint64_t *)ix->d_pos;
	return GDIET_OK;
}

extern "C" int gdiet_hip_index_build(gdiet_ctx *ctx, gdiet_index **out, int n_seq, const char *const *names, const char *const *seqs,
                                     const uint32_t *lens, int k, int w, const char *pattern, int pattern_len, int n_threads)
{
	if (!ctx || !out || n_seq <= 0 || !seqs || !lens) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	GdPattern P;
	if (!gd_pattern_init(P, pattern, pattern_len)) { ctx->err = "bad pattern"; return GDIET_E_PARAM; }
	if (w <= 0 || w > GDM_MAX_W || k <= 0 || k > 28) { ctx->err = "k must be in [1,28] and w in [1,64]"; return GDIET_E_PARAM; }
	gdiet_index *ix = new gdiet_index();
	std::vector<std::string> nm(n_seq);
	std::vector<GdSeqSpan> sq(n_seq);
	for (int i = 0; i < n_seq; ++i) nm[i] = names && names[i] ? names[i] : "", sq[i].p = seqs[i], sq[i].n = lens[i];
	const int nt = n_threads > 0 ? n_threads : gd_effective_cpus();
	if (ctx->index_on_device) { // sketch, sort and table build on the GPU (map_index_dev.hip.h); GDIET_INDEX_BUILD=host selects the host builder
		if (!gd_index_build_device(ix, nm, sq, k, w, P, nt, ctx->stream, ctx->err)) { gdiet_hip_index_destroy(ctx, ix); return GDIET_E_HIP; }
		*out = ix;
		return GDIET_OK;
	}
	gd_index_build(ix->h, nm, sq, k, w, P, nt, true);
	int rc = gd_index_upload(ctx, ix);
	if (rc) { delete ix; return rc; }
	*out = ix;
	return GDIET_OK;
}

extern "C" int gdiet_hip_index_import(gdiet_ctx *ctx, gdiet_index **out, int k, int w, const char *pattern, int pattern_len, int n_seq,
                                      const char *const *names, const uint32_t *lens, const uint64_t *offsets, const uint32_t *S,
                                      uint64_t n_keys, const uint64_t *keys, const uint32_t *cnt, const uint64_t *pos)
{
	if (!ctx || !out || n_seq <= 0 || !lens || !offsets || !S || !keys || !cnt || !pos) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	GdPattern P;
	if (!gd_pattern_init(P, pattern, pattern_len)) { ctx->err = "bad pattern"; return GDIET_E_PARAM; }
	if (w <= 0 || w > GDM_MAX_W || k <= 0 || k > 28) { ctx->err = "k must be in [1,28] and w in [1,64]"; return GDIET_E_PARAM; }
	gdiet_index *ix = new gdiet_index();
	gd_index_from_flat(ix->h, k, w, P, n_seq, names, lens, offsets, S, n_keys, keys, cnt, pos);
	int rc = gd_index_upload(ctx, ix);
	if (rc) { delete ix; return rc; }
	*out = ix;
	return GDIET_OK;
}

// flat view of the index, the inverse of gdiet_hip_index_import (keys in table order; each position list ascending)
extern "C" int gdiet_hip_index_export(const gdiet_index *ix, uint64_t *n_keys, uint64_t *n_pos, uint64_t *n_S_words, uint64_t *keys,
                                      uint32_t *cnt, uint64_t *pos, uint32_t *S, uint64_t *offsets)
{
	if (!ix) return GDIET_E_PARAM;
	{ std::string e; if (!gd_index_fetch_host(const_cast<gdiet_index *>(ix), e)) return GDIET_E_HIP; }
	const GdIndex &h = ix->h;
	if (n_keys) *n_keys = h.n_keys;
	if (n_pos) *n_pos = h.pos.size();
	if (n_S_words) *n_S_words = h.S.size();
	if (keys && cnt) {
		uint64_t j = 0;
		for (size_t s = 0; s < h.tkey.size(); ++s)
			if (h.tkey[s] != UINT64_MAX) keys[j] = h.tkey[s], cnt[j] = (uint32_t)h.tval[s], ++j;
		if (j != h.n_keys) return GDIET_E_PARAM;
	}
	if (pos && keys && cnt) { // lists concatenated in the order of keys[]
		uint64_t o = 0;
		for (size_t s = 0; s < h.tkey.size(); ++s)
			if (h.tkey[s] != UINT64_MAX) {
				const uint64_t st = h.tval[s] >> 32, n = (uint32_t)h.tval[s];
				for (uint64_t t = 0; t < n; ++t) pos[o++] = h.pos[st + t];
			}
	}
	if (S) memcpy(S, h.S.data(), 4 * h.S.size());
	if (offsets) for (size_t i = 0; i < h.seq.size(); ++i) offsets[i] = h.seq[i].offset;
	return GDIET_OK;
}

// ---- .mmi files: parser / writer in map_index.h (gd_index_read_mmi / gd_index_write_mmi) ---------------------------------------
extern "C" int gdiet_hip_index_load_mmi(gdiet_ctx *ctx, gdiet_index **out, const char *path, const char *pattern, int pattern_len)
{
	if (!ctx || !out || !path) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	GdPattern P;
	if (!gd_pattern_init(P, pattern, pattern_len)) { ctx->err = "bad pattern"; return GDIET_E_PARAM; }
	gdiet_index *ix = new gdiet_index();
	if (!gd_index_read_mmi(ix->h, path, P, ctx->err)) { delete ix; return GDIET_E_PARAM; }
	if (ix->h.w <= 0 || ix->h.w > GDM_MAX_W || ix->h.k <= 0 || ix->h.k > 28) { // the sketch kernels' windows are sized for these (gdiet_hip_index_build)
		ctx->err = std::string(path) + ": k must be in [1,28] and w in [1,64]";
		delete ix;
		return GDIET_E_PARAM;
	}
	int rc = gd_index_upload(ctx, ix);
	if (rc) { delete ix; return rc; }
	*out = ix;
	return GDIET_OK;
}

extern "C" int gdiet_hip_index_dump_mmi(gdiet_ctx *ctx, const gdiet_index *ix, const char *path, int bucket_bits)
{
	if (!ctx || !ix || !path || bucket_bits < 1 || bucket_bits > 28) return GDIET_E_PARAM;
	if (!gd_index_fetch_host(const_cast<gdiet_index *>(ix), ctx->err)) return GDIET_E_HIP;
	return gd_index_write_mmi(ix->h, path, bucket_bits, ctx->err) ? GDIET_OK : GDIET_E_PARAM;
}

extern "C" void gdiet_hip_index_destroy(gdiet_ctx *ctx, gdiet_index *ix)
{
	if (!ix) return;
	if (ctx) (void)hipSetDevice(ctx->device);
	void *p[] = {ix->d_tkey, ix->d_tval, ix->d_pos, ix->d_S, ix->d_cnt_sorted, ix->d_seq_len, ix->d_seq_off};
	for (void *q : p) if (q) (void)hipFree(q);
	delete ix;
}

extern "C" int32_t gdiet_hip_index_cal_max_occ(const gdiet_index *ix, float frac)
{
	if (!ix) return 0;
	if (ix->host_tables) return gd_index_cal_max_occ(ix->h, frac);
	// device-built: the counts are on the device, sorted; mm_idx_cal_max_occ (LR/index.c:190-210) is their (1-f) quantile + 1
	if (frac <= 0.) return INT32_MAX;
	const size_t n = ix->h.n_keys;
	if (n == 0) return 1;
	size_t kk = (size_t)(uint32_t)((1. - frac) * n);
	if (kk >= n) kk = n - 1;
	uint32_t v = 0;
	if (hipMemcpy(&v, ix->d_cnt_sorted + kk, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
	return (int32_t)(v + 1);
}
extern "C" uint64_t gdiet_hip_index_n_keys(const gdiet_index *ix) { return ix ? ix->h.n_keys : 0; }

extern "C" int gdiet_hip_set_host_threads(gdiet_ctx *ctx, int n)
{
	if (!ctx || n < 1) return GDIET_E_PARAM;
	ctx->host_threads = n;
	return GDIET_OK;
}

extern "C" int gdiet_hip_map_stage_seconds(const gdiet_ctx *ctx, double out[6])
{
	if (!ctx || !out) return GDIET_E_PARAM;
	for (int i = 0; i < 6; ++i) out[i] = ctx->stage_s[i];
	return GDIET_OK;
}

extern "C" int gdiet_hip_batch_upload(gdiet_ctx *ctx, gdiet_read_batch **out, int n, const char *const *seqs, const int32_t *lens)
{
	if (!ctx || !out || n < 0 || (n && (!seqs || !lens))) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	static const bool trace = getenv("GDIET_TRACE_STAGES") != nullptr;
	double t_[5] = {gd_now(), 0, 0, 0, 0};
	gdiet_read_batch *b = new gdiet_read_batch();
	b->n = n;
	b->roff.assign(n + 1, 0);
	for (int i = 0; i < n; ++i) b->roff[i + 1] = b->roff[i] + (lens[i] > 0 ? lens[i] : 0);
	// the host copy of the encoded reads lives in a buffer taken from the context's pool (returned by gdiet_hip_batch_destroy): first
	// touch of fresh memory -- 77 MB per HiFi mini-batch -- cost ten times the encoding itself
	const size_t enc_len = (size_t)b->roff[n] + 8;
	{
		std::lock_guard<std::mutex> lk(ctx->enc_mu);
		if (!ctx->enc_pool.empty()) b->enc.swap(ctx->enc_pool.back()), ctx->enc_pool.pop_back();
	}
	if (b->enc.size() < enc_len) b->enc.resize(enc_len + (enc_len >> 3));
	t_[1] = gd_now();
	gd_parallel_for(ctx, ctx->host_threads, n, [&](int i) {
		if (lens[i] > 0) gd_nt4_encode(seqs[i], b->enc.data() + b->roff[i], (size_t)lens[i]);
	});
	t_[2] = gd_now();
	// stream-ordered allocation: a plain hipMalloc / hipFree per mini-batch synchronises the whole device, i.e. every batch in flight
	hipError_t e = hipMallocAsync(&b->d_reads, enc_len + 64, ctx->stream); // (slack: the seed kernel reads aligned 8-byte words)
	if (e == hipSuccess) e = hipMallocAsync(&b->d_roff, sizeof(int64_t) * (n + 1), ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(b->d_reads, b->enc.data(), enc_len, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(b->d_roff, b->roff.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream);
	t_[3] = gd_now();
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	t_[4] = gd_now();
	if (trace) fprintf(stderr, "[gdiet upload, ms] n=%d alloc %.2f encode %.2f enqueue %.2f sync %.2f\n", n, 1e3 * (t_[1] - t_[0]), 1e3 * (t_[2] - t_[1]), 1e3 * (t_[3] - t_[2]), 1e3 * (t_[4] - t_[3]));
	if (e != hipSuccess) { ctx->err = std::string("batch upload: ") + hipGetErrorString(e); delete b; return GDIET_E_HIP; }
	*out = b;
	return GDIET_OK;
}

extern "C" void gdiet_hip_batch_destroy(gdiet_ctx *ctx, gdiet_read_batch *b)
{
	if (!b) return;
	if (ctx) (void)hipSetDevice(ctx->device);
	if (ctx) {
		{
			std::lock_guard<std::mutex> lk(ctx->enc_mu);
			if (ctx->enc_pool.size() < 6) ctx->enc_pool.push_back(std::move(b->enc));
		}
		if (b->d_reads) (void)hipFreeAsync(b->d_reads, ctx->stream);
		if (b->d_roff) (void)hipFreeAsync(b->d_roff, ctx->stream);
	} else {
		if (b->d_reads) (void)hipFree(b->d_reads);
		if (b->d_roff) (void)hipFree(b->d_roff);
	}
	delete b;
}

// Records leave the library in slabs of many reads each: a short-read batch has a quarter of a million reads, and an allocation
// per read made the caller's gdiet_hip_free_regs the slowest stage of the pipeline.  Every host worker fills slabs of its own (a
// read's records are written, and its temporaries released, by the thread that computed them), so no lock is involved.  Every
// read's record array is preceded by a 16-byte GdRegHead; `head` marks the ones that start an allocation.
struct GdRegHead { uint32_t magic, head; uint64_t pad; };
static const uint32_t GD_REG_MAGIC = 0x67645247u;
static const size_t GD_REG_SLAB = 1 << 20;

struct GdRegCursor { uint64_t call = 0; char *p = nullptr; size_t left = 0; };
static GdRegCursor &gd_reg_cursor() // (this thread's: fetched once per chunk of reads -- a thread_local of a shared library costs a call per access)
{
	static thread_local GdRegCursor C;
	return C;
}
static void *gd_reg_slab_take(GdRegCursor &C, uint64_t call_id, size_t bytes, bool &head)
{
	head = false;
	if (C.call != call_id || C.left < bytes) { // what is left of the previous slab stays with its records
		const size_t sz = std::max(bytes, GD_REG_SLAB);
		char *m = (char *)malloc(sz);
		if (!m) return nullptr;
		C.call = call_id, C.p = m, C.left = sz, head = true;
	}
	void *at = C.p;
	C.p += bytes, C.left -= bytes;
	return at;
}

static size_t gd_regs_bytes(const std::vector<GdReg> &v)
{
	if (v.empty()) return 0;
	size_t words = 0;
	for (const GdReg &g : v) words += g.cigar.size() + 1;
	return sizeof(GdRegHead) + v.size() * sizeof(gdiet_reg_t) + ((words * 4 + 7) & ~(size_t)7);
}

static void gd_regs_fill(const std::vector<GdReg> &v, void *at, bool head, int32_t *n_regs, gdiet_reg_t **regs)
{
	*n_regs = (int32_t)v.size();
	*regs = nullptr;
	if (v.empty()) return;
	GdRegHead *h = (GdRegHead *)at;
	h->magic = GD_REG_MAGIC, h->head = head ? 1 : 0, h->pad = 0;
	gdiet_reg_t *r = (gdiet_reg_t *)(h + 1);
	uint32_t *cg = (uint32_t *)(r + v.size());
	for (size_t i = 0; i < v.size(); ++i) {
		const GdReg &g = v[i];
		memset(&r[i], 0, sizeof(r[i]));
		r[i].id = g.id, r[i].cnt = g.cnt, r[i].rid = g.rid, r[i].score = g.score, r[i].qs = g.qs, r[i].qe = g.qe, r[i].rs = g.rs, r[i].re = g.re;
		r[i].parent = g.parent, r[i].subsc = g.subsc, r[i].mlen = g.mlen, r[i].blen = g.blen, r[i].mapq = g.mapq, r[i].rev = g.rev, r[i].sam_pri = g.sam_pri;
		r[i].dp_score = g.dp_score, r[i].dp_max = g.dp_max, r[i].n_ambi = g.n_ambi, r[i].n_cigar = (uint32_t)g.cigar.size();
		r[i].cigar = cg;
		if (!g.cigar.empty()) memcpy(cg, g.cigar.data(), g.cigar.size() * 4);
		cg[g.cigar.size()] = 0;
		cg += g.cigar.size() + 1;
	}
	*regs = r;
}

extern "C" void gdiet_hip_free_regs(int n, int32_t *n_regs, gdiet_reg_t **regs)
{
	if (!n_regs || !regs) return;
	std::vector<void *> blocks; // freed after the scan: the other reads' heads live inside them
	for (int i = 0; i < n; ++i) {
		if (regs[i]) {
			GdRegHead *h = (GdRegHead *)regs[i] - 1;
			if (h->magic == GD_REG_MAGIC && h->head) blocks.push_back(h);
		}
		regs[i] = nullptr, n_regs[i] = 0;
	}
	for (void *b : blocks) free(b);
}

static void gd_opt_from_c(const gdiet_mapopt_t *o, const gdiet_index *ix, GdMapOpt &O)
{
	O.flag = o->flag, O.k = ix->h.k, O.w = ix->h.w, O.a = o->a, O.b = o->b, O.q = o->q, O.e = o->e, O.q2 = o->q2, O.e2 = o->e2;
	O.bw = o->bw, O.min_dp_max = o->min_dp_max, O.best_n = o->best_n, O.q_occ_frac = o->q_occ_frac, O.mid_occ = o->mid_occ;
	O.max_max_occ = o->max_max_occ, O.occ_dist = o->occ_dist, O.max_frag_len = o->max_frag_len, O.vt_dis = o->vt_dis, O.vt_nb_loc = o->vt_nb_loc;
	O.vt_cov = o->vt_cov, O.vt_f = o->vt_f, O.vt_df1 = o->vt_df1, O.vt_df2 = o->vt_df2, O.max_max_gap = o->max_max_gap, O.max_min_gap = o->max_min_gap;
	O.max_seeds = o->max_seeds, O.pat = ix->h.pat;
	O.min_cnt = o->min_cnt, O.rec_threshold_frac = o->rec_threshold_frac, O.bw_frac = o->bw_frac, O.bw_min = o->bw_min, O.bw_max = o->bw_max;
	O.af_max_loc = o->AF_max_loc;
}

static void gd_drop_async_lanes(gdiet_ctx *ctx)
{
	for (int i = 0; i < GD_MAX_INFLIGHT; ++i)
		if (ctx->async_lane[i] && !ctx->async_busy[i]) gdiet_hip_destroy(ctx->async_lane[i]), ctx->async_lane[i] = nullptr;
}

static int gd_check_opt(gdiet_ctx *ctx, const GdMapOpt &O)
{
	if (O.flag & GD_F_SR) {
		if (O.af_max_loc < 1 || O.af_max_loc > GDM_MAX_VT) { ctx->err = "AF_max_loc must be in [1," + std::to_string(GDM_MAX_VT) + "]"; return GDIET_E_PARAM; }
	} else if (O.vt_nb_loc + 2 > GDM_MAX_VT) { ctx->err = "vt_nb_loc too large"; return GDIET_E_PARAM; }
	if (O.mid_occ <= 0) { ctx->err = "mid_occ must be set (mm_mapopt_update)"; return GDIET_E_PARAM; }
	return GDIET_OK;
}

// a contiguous slice of a resident read batch
struct GdBatchView {
	int n;
	const int64_t *roff;   // n+1 absolute offsets into enc / d_reads
	const uint8_t *enc;    // host copy (whole batch)
	const uint8_t *d_reads;
	const int64_t *d_roff; // device copy of roff (already offset to the slice)
};

// LDS capacities of the wave seed / vote kernels (entries; the launch sizes its dynamic LDS from them): upper bounds, lowered for A/B
// measurements with GDIET_SEED_SORT_CAP / GDIET_VOTE_CAP -- less LDS per wavefront = more wavefronts per CU, longer lists go through the
// kernels' global-memory paths (every path is exact)
static int gd_sort_cap_max()
{
	static const int v = [] { const char *e = getenv("GDIET_SEED_SORT_CAP"); int c = e ? atoi(e) : MAP_SORT_CAP_MAX; int p = MAP_SORT_CAP; while (p < c && p < MAP_SORT_CAP_MAX) p <<= 1; return p; }();
	return v;
}
static unsigned gd_vote_cap_max()
{
	static const unsigned v = [] { const char *e = getenv("GDIET_VOTE_CAP"); unsigned c = e ? (unsigned)atoi(e) : MAP_VOTE_CAP, p = 256; while (p < c && p < MAP_VOTE_CAP) p <<= 1; return p; }();
	return v;
}

// B4 (gdiet_hip_seed_batch): what the seeding stage leaves, copied out right behind the seed kernel instead of going on
struct GdSeedExport {
	std::vector<MapSeedOut> out;   // per read: shift, tmp_extracted_len, n_mv, n_seeds, n_a
	std::vector<GdSeed> seeds;     // the kept seeds of all reads, back to back
};

// the whole per-read path for one slice, on ctx's own stream and buffers (ctx is a lane: the parent context or one of its children)
static int gd_map_range(gdiet_ctx *ctx, const gdiet_index *ix, const GdMapOpt &O, const GdBatchView &B, int32_t *n_regs, gdiet_reg_t **regs, GdSeedExport *sx = nullptr)
{
	// GDIET_TRACE_STAGES=1: wall time of the host-side sub-steps of this call to stderr (development aid)
	static const bool trace = getenv("GDIET_TRACE_STAGES") != nullptr;
	double tr_t = trace ? gd_now() : 0;
	const double tr_t0 = tr_t;
	std::string tr_s;
	auto mark = [&](const char *what) {
		if (!trace) return;
		const double now = gd_now();
		char b[64];
		snprintf(b, sizeof b, " %s %.2f", what, 1e3 * (now - tr_t));
		tr_s += b, tr_t = now;
	};

	(void)hipSetDevice(ctx->device);
	const int n = B.n;
	for (int i = 0; i < 6; ++i) ctx->stage_s[i] = 0;
	if (n == 0) return GDIET_OK;
	if (!sx) for (int i = 0; i < n; ++i) n_regs[i] = 0, regs[i] = nullptr; // whatever happens below, gdiet_hip_free_regs on these arrays is safe
	hipStream_t s = ctx->stream;
	int rc;
	double t0 = gd_now();
	// ---- scratch layout -------------------------------------------------------------------------------------------
	// minimizer lists: len/3 + 512 entries per read cover every density the presets produce; a read that overflows its list
	// (tiny windows, homopolymer reads) makes the whole batch retry once with the hard bound
	std::vector<MapReadScratch> sc(n);
	uint64_t tot = 0;
	auto layout = [&](bool full) -> int {
		tot = 0;
		for (int i = 0; i < n; ++i) {
			const uint32_t len = (uint32_t)(B.roff[i + 1] - B.roff[i]);
			// hard bound: one minimizer per base, plus the per-lane staging lists of the wavefront sketch (64 lists of ceil(len/64) + w + 2)
			sc[i].mv_cap = full ? len + 64 * (uint32_t)(O.w + 4) : len / 3 + 512, sc[i].mv_off = tot, sc[i].u64_off = 2 * tot, sc[i].seed_off = tot, sc[i].pad = 0;
			tot += sc[i].mv_cap;
		}
		int rc2;
		if ((rc2 = gd_grow(ctx, ctx->m_sc, sizeof(MapReadScratch) * n))) return rc2;
		if ((rc2 = gd_grow(ctx, ctx->m_mv, sizeof(GdMini) * tot))) return rc2;
		if ((rc2 = gd_grow(ctx, ctx->m_u64, sizeof(uint64_t) * 2 * tot))) return rc2;
		if ((rc2 = gd_grow(ctx, ctx->m_seed, sizeof(GdSeed) * tot))) return rc2;
		GD_HIP(hipMemcpyAsync(ctx->m_sc.p, sc.data(), sizeof(MapReadScratch) * n, hipMemcpyHostToDevice, s));
		return GDIET_OK;
	};
	if ((rc = layout(false))) return rc;
	if ((rc = gd_grow(ctx, ctx->m_seedout, sizeof(MapSeedOut) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->m_voteout, sizeof(MapVoteOut) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->m_hitoff, sizeof(int64_t) * (n + 1)))) return rc;
	MapDevOpt D;
	D.k = O.k, D.w = O.w, D.max_seeds = O.max_seeds, D.q_occ_frac = O.q_occ_frac, D.mid_occ = O.mid_occ, D.max_max_occ = O.max_max_occ, D.occ_dist = O.occ_dist;
	D.max_nb_seeds = (O.flag & GD_F_FRAG_MODE) ? (O.max_frag_len == 0 ? 800u : (uint32_t)O.max_frag_len) : UINT32_MAX;
	D.flag = O.flag, D.pat = O.pat;
	D.vote.vt_dis = O.vt_dis, D.vote.vt_nb_loc = O.vt_nb_loc, D.vote.bw = O.bw, D.vote.vt_cov = O.vt_cov, D.vote.vt_f = O.vt_f;
	D.vote.vt_df1 = O.vt_df1, D.vote.vt_df2 = O.vt_df2, D.vote.k = O.k;
	const bool is_sr = (O.flag & GD_F_SR) != 0;
	D.is_sr = is_sr;
	// The seeding / voting wavefronts of a batch run beside the DP wavefronts of the batch before it (five per SIMD, VALU-bound, older): at
	// the default priority they get the issue slots those leave, hold their scarce slots for a long time, and the batch is ready only when
	// that DP kernel ends -- with two batches in flight the next DP kernel then starts ~5 ms late (834 Mbases/s).  At s_setprio(2) they are
	// through in ~60 ms, the DP kernels follow each other back to back with TWO batches in flight (875-881 Mbases/s) and a batch spends two
	// step times in the pipeline instead of three (p50 174 instead of 261 ms).  GDIET_SIDE_PRIO=0..3 for A/B runs.
	{ static const int side_prio = getenv("GDIET_SIDE_PRIO") ? atoi(getenv("GDIET_SIDE_PRIO")) : 2; D.prio = side_prio; }
	{ // LDS sort capacity of the wave seed kernel: ~1.25 x the minimizers expected of the longest read (2 / (w + 1) of its sparsified bases)
		int64_t max_len = 0;
		for (int i = 0; i < n; ++i) max_len = std::max<int64_t>(max_len, B.roff[i + 1] - B.roff[i]);
		const double est = 1.25 * 2.0 / (O.w + 1) * gd_diet_len(O.pat, (unsigned)max_len, 0);
		int cap = MAP_SORT_CAP;
		while (cap < gd_sort_cap_max() && cap < est) cap <<= 1;
		D.sort_cap = cap;
	}
	// Reads of very different lengths (ONT: log-normal up to 150 kbp): one launch per capacity class, each read in the class its own
	// length asks for, so that a 30 kbp read does not hold the 128 KB of LDS the longest read of the batch needs -- 128 KB is one
	// wavefront per CU, and never beside a DP kernel that keeps 48 KB of it.  (A read is treated exactly as if it were the longest read
	// of its batch: every path of the kernel is exact, the capacity only selects between them.)
	std::vector<int32_t> seed_ids;
	std::vector<std::pair<int, int>> seed_classes; // (capacity, reads), longest class first; empty: one launch over all reads
	if (D.sort_cap > MAP_SORT_CAP && n > 1) {
		std::vector<int> cap_of(n);
		int n_cls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (int i = 0; i < n; ++i) {
			const double est = 1.25 * 2.0 / (O.w + 1) * gd_diet_len(O.pat, (unsigned)(B.roff[i + 1] - B.roff[i]), 0);
			int cap = MAP_SORT_CAP, c = 0;
			while (cap < gd_sort_cap_max() && cap < est) cap <<= 1, ++c;
			cap_of[i] = c, ++n_cls[c];
		}
		int used = 0;
		for (int c = 0; c < 8; ++c) used += n_cls[c] > 0;
		if (used > 1) {
			seed_ids.reserve(n);
			for (int c = 7; c >= 0; --c) {
				if (!n_cls[c]) continue;
				seed_classes.push_back({MAP_SORT_CAP << c, n_cls[c]});
				for (int i = 0; i < n; ++i) if (cap_of[i] == c) seed_ids.push_back(i);
			}
		}
	}
	D.sr.min_cnt = O.min_cnt, D.sr.rec_threshold_frac = O.rec_threshold_frac, D.sr.bw_frac = O.bw_frac, D.sr.bw_min = O.bw_min, D.sr.bw_max = O.bw_max;
	D.sr.af_max_loc = O.af_max_loc, D.sr.max_nb_seeds = D.max_nb_seeds, D.sr.frag_mode = (O.flag & GD_F_FRAG_MODE) != 0;
	const uint8_t *d_reads = (const uint8_t *)B.d_reads;
	const int64_t *d_roff = (const int64_t *)B.d_roff;
	ctx->stage_s[5] += gd_now() - t0, t0 = gd_now();
	// ---- S1-S5 ----------------------------------------------------------------------------------------------------
	if ((rc = gd_host_grow(ctx, ctx->h_seedout, sizeof(MapSeedOut) * (size_t)n))) return rc;
	MapSeedOut *so = (MapSeedOut *)ctx->h_seedout.p;
	const size_t seed_lds = std::max<size_t>((size_t)O.w * 64 * sizeof(GdMini), (size_t)D.sort_cap * sizeof(uint64_t));
	D.seed_lds = (uint32_t)seed_lds;
	if (seed_lds > 64 * 1024) GD_HIP(hipFuncSetAttribute((const void *)map_seed_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)seed_lds));
	for (int attempt = 0; attempt < 2; ++attempt) {
		// one read per thread (the plain sequential form) for short reads -- a 150 bp read has ~75 sparsified bases, far too few to
		// split over 64 lanes (measured 18x faster at 150 bp) -- and on request (GDIET_SEED_KERNEL=thread) for A/B checks
		if (ctx->seed_thread_kernel == 1 || (ctx->seed_thread_kernel == 0 && (B.roff[n] - B.roff[0]) / n < 1024))
			hipLaunchKernelGGL(map_seed_kernel, dim3((n + 63) / 64), dim3(64), (size_t)O.w * 64 * sizeof(GdMini), s, n, d_reads, d_roff, ix->dview, D, (const MapReadScratch *)ctx->m_sc.p,
			                   (GdMini *)ctx->m_mv.p, (uint64_t *)ctx->m_u64.p, (GdSeed *)ctx->m_seed.p, (MapSeedOut *)ctx->m_seedout.p);
		else if (!seed_classes.empty()) { // one read per wavefront, one launch per LDS capacity class
			if ((rc = gd_grow(ctx, ctx->m_seedids, sizeof(int32_t) * (size_t)n))) return rc;
			GD_HIP(hipMemcpyAsync(ctx->m_seedids.p, seed_ids.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
			int at = 0;
			for (const auto &cls : seed_classes) {
				MapDevOpt Dc = D;
				Dc.sort_cap = cls.first;
				const size_t lds_c = std::max<size_t>((size_t)O.w * 64 * sizeof(GdMini), (size_t)cls.first * sizeof(uint64_t));
				Dc.seed_lds = (uint32_t)lds_c;
				hipLaunchKernelGGL(map_seed_wave_kernel, dim3(cls.second), dim3(64), lds_c, s, cls.second, d_reads, d_roff, ix->dview, Dc, (const MapReadScratch *)ctx->m_sc.p,
				                   (GdMini *)ctx->m_mv.p, (uint64_t *)ctx->m_u64.p, (GdSeed *)ctx->m_seed.p, (MapSeedOut *)ctx->m_seedout.p, (const int32_t *)ctx->m_seedids.p + at);
				at += cls.second;
			}
		} else // one read per wavefront: 64 exact slices of the winnowing automaton + parallel index probes
			hipLaunchKernelGGL(map_seed_wave_kernel, dim3(n), dim3(64), seed_lds, s, n, d_reads, d_roff, ix->dview, D, (const MapReadScratch *)ctx->m_sc.p,
			                   (GdMini *)ctx->m_mv.p, (uint64_t *)ctx->m_u64.p, (GdSeed *)ctx->m_seed.p, (MapSeedOut *)ctx->m_seedout.p, (const int32_t *)nullptr);
		GD_HIP(hipGetLastError()); // a refused launch (LDS size, grid) must not pass for stale seed records
		GD_HIP(hipMemcpyAsync(so, ctx->m_seedout.p, sizeof(MapSeedOut) * n, hipMemcpyDeviceToHost, s));
		GD_HIP(gd_stream_wait(ctx, s));
		bool overflow = false;
		for (int i = 0; i < n; ++i) overflow |= so[i].n_seeds < 0;
		if (!overflow) break;
		if (attempt == 1) { ctx->err = "minimizer scratch overflow even with one entry per base"; return GDIET_E_NOMEM; }
		if ((rc = layout(true))) return rc;
	}
	ctx->stage_s[0] += gd_now() - t0, t0 = gd_now();
	mark("seed");
	if (sx) { // B4: the seeds themselves are the result
		sx->out.assign(so, so + n);
		int64_t tot_seeds = 0;
		for (int i = 0; i < n; ++i) tot_seeds += so[i].n_seeds > 0 ? so[i].n_seeds : 0;
		sx->seeds.resize((size_t)tot_seeds);
		int64_t at = 0;
		for (int i = 0; i < n; ++i)
			if (so[i].n_seeds > 0) {
				GD_HIP(hipMemcpyAsync(sx->seeds.data() + at, (const GdSeed *)ctx->m_seed.p + sc[i].seed_off, sizeof(GdSeed) * (size_t)so[i].n_seeds, hipMemcpyDeviceToHost, s));
				at += so[i].n_seeds;
			}
		GD_HIP(gd_stream_wait(ctx, s));
		return GDIET_OK;
	}
	std::vector<int64_t> hoff(n + 1, 0);
	for (int i = 0; i < n; ++i) {
		hoff[i + 1] = hoff[i] + (so[i].n_seeds > 0 ? so[i].n_a : 0);
	}
	if ((rc = gd_grow(ctx, ctx->m_hits, sizeof(GdLoc) * 3 * (size_t)(hoff[n] + 1)))) return rc;
	GD_HIP(hipMemcpyAsync(ctx->m_hitoff.p, hoff.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s));
	// ---- S6, S7, V1, V3, G1a -----------------------------------------------------------------------------------------
	// long reads: one read per wavefront (parallel expansion + LDS sort); short reads (a handful of hits each): one read per thread
	const int spread = ctx->spread && (B.roff[n] - B.roff[0]) / n >= 1024;
	if (spread == 1 && ctx->vote_wave) {
		// LDS sort buffer of the batch: the largest hit count of a read bounds either strand (see the kernel)
		int64_t max_hits = 0;
		for (int i = 0; i < n; ++i) max_hits = std::max(max_hits, hoff[i + 1] - hoff[i]);
		unsigned vote_cap = 256;
		while (vote_cap < gd_vote_cap_max() && (int64_t)vote_cap < max_hits) vote_cap <<= 1;
		// the kernel also holds static LDS (its candidate list): with the full sort buffer the total passes 64 KB, which a launch is
		// only granted after the attribute has been raised
		if (sizeof(GdLoc) * (size_t)vote_cap + sizeof(GdVt) * GDM_MAX_VT + 64 > 64 * 1024)
			GD_HIP(hipFuncSetAttribute((const void *)map_vote_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(GdLoc) * (size_t)vote_cap)));
		hipLaunchKernelGGL(map_vote_wave_kernel, dim3(n), dim3(64), sizeof(GdLoc) * (size_t)vote_cap, s, n, d_roff, ix->dview, D, (const MapReadScratch *)ctx->m_sc.p,
		                   (const GdSeed *)ctx->m_seed.p, (const MapSeedOut *)ctx->m_seedout.p, (const int64_t *)ctx->m_hitoff.p, (GdLoc *)ctx->m_hits.p,
		                   (MapVoteOut *)ctx->m_voteout.p, vote_cap);
	} else
		hipLaunchKernelGGL(map_vote_kernel, dim3(spread ? n : (n + 63) / 64), dim3(64), 0, s, n, d_roff, ix->dview, D, (const MapReadScratch *)ctx->m_sc.p,
		                   (const GdSeed *)ctx->m_seed.p, (const MapSeedOut *)ctx->m_seedout.p, (const int64_t *)ctx->m_hitoff.p, (GdLoc *)ctx->m_hits.p,
		                   (MapVoteOut *)ctx->m_voteout.p, spread);
	GD_HIP(hipGetLastError());
	// ---- G1b / G2 ---------------------------------------------------------------------------------------------------------------
	// ShortReads: the box stage on the device (map_sr_box_kernel / map_sr_fill_kernel): the candidates never visit the host; what comes
	// back are the dense per-box tables the DP planner and the record stage read.  GDIET_SR_BOXES=host keeps the host stages.
	const bool sr_dev = is_sr && ctx->sr_boxes_on_device && (int64_t)n * std::min<int64_t>(O.af_max_loc, GDM_MAX_VT) < ((int64_t)1 << 28);
	const GdRefView R = ix->h.ref();
	std::vector<int> cfirst(n + 1, 0), ccount(n, 0), box_first(n + 1, 0);
	GdCandBox *cflat = nullptr;
	int nb = 0;
	MapBox *boxes = nullptr;
	std::vector<int64_t> qoff, toff, coff;
	std::vector<int32_t> bw, ex;
	size_t nbp = 64;
	int64_t *d_coff = nullptr;
	int32_t *d_ex = nullptr, *d_score = nullptr, *d_ncig = nullptr;
	auto aux_layout = [&]() -> int { // d_coff | d_ex | d_score | d_ncig in one buffer, the last three 256-byte aligned as their host copies
		nbp = ((size_t)std::max(nb, 1) + 63) & ~(size_t)63;
		int rc2;
		if ((rc2 = gd_grow(ctx, ctx->m_aux, sizeof(int64_t) * (nb + 1) + sizeof(int32_t) * 3 * nbp + 1024))) return rc2;
		d_coff = (int64_t *)ctx->m_aux.p;
		d_ex = (int32_t *)(((uintptr_t)(d_coff + nb + 1) + 255) & ~(uintptr_t)255), d_score = d_ex + nbp, d_ncig = d_score + nbp;
		return GDIET_OK;
	};
	if (sr_dev) {
		const int slots = (int)std::min<int64_t>(O.af_max_loc, GDM_MAX_VT);
		{ // the contig table of the index on the device (once)
			gdiet_index *ixm = const_cast<gdiet_index *>(ix);
			std::lock_guard<std::mutex> lk(ixm->seq_mu);
			if (!ixm->d_seq_len) {
				std::vector<uint32_t> sl(R.n_seq);
				std::vector<uint64_t> so(R.n_seq);
				for (uint32_t q = 0; q < R.n_seq; ++q) sl[q] = R.seq[q].len, so[q] = R.seq[q].offset;
				uint32_t *dl = nullptr;
				uint64_t *d_o = nullptr;
				GD_HIP(hipMalloc(&dl, sizeof(uint32_t) * std::max<size_t>(R.n_seq, 1)));
				GD_HIP(hipMalloc(&d_o, sizeof(uint64_t) * std::max<size_t>(R.n_seq, 1)));
				GD_HIP(hipMemcpy(dl, sl.data(), sizeof(uint32_t) * R.n_seq, hipMemcpyHostToDevice));
				GD_HIP(hipMemcpy(d_o, so.data(), sizeof(uint64_t) * R.n_seq, hipMemcpyHostToDevice));
				ixm->d_seq_off = d_o, ixm->d_seq_len = dl;
			}
		}
		// per-read tables: cnt[n] | box_first[n + 1] (int32), sumlen[n] | len_first[n + 1] (int64), totals
		const size_t tab_bytes = sizeof(int64_t) * (2 * (size_t)n + 2) + sizeof(int32_t) * (2 * (size_t)n + 2) + sizeof(MapSrTotals) + 64;
		if ((rc = gd_grow(ctx, ctx->m_srtab, tab_bytes))) return rc;
		if ((rc = gd_grow(ctx, ctx->m_srbox, sizeof(GdCandBox) * (size_t)n * slots))) return rc;
		int64_t *d_sumlen = (int64_t *)ctx->m_srtab.p, *d_lenfirst = d_sumlen + n;
		int32_t *d_cnt = (int32_t *)(d_lenfirst + n + 1), *d_boxfirst = d_cnt + n;
		MapSrTotals *d_tot = (MapSrTotals *)(d_boxfirst + n + 1);
		static const char *fault_env = getenv("GDIET_FAULT_BOX");
		const MapSrTotals zero = {0, -1};
		GD_HIP(hipMemcpyAsync(d_tot, &zero, sizeof zero, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(map_sr_box_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, d_roff, (const MapVoteOut *)ctx->m_voteout.p, (const uint32_t *)ix->d_seq_len, R.n_seq,
		                   O.k, O.a, slots, fault_env ? atoi(fault_env) : -1, (GdCandBox *)ctx->m_srbox.p, d_cnt, d_sumlen, d_tot);
		// exclusive scans over n + 1 entries (the last input is never read as a value: the scans' last outputs are the totals)
		size_t b1 = 0, b2 = 0;
		GD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b1, d_cnt, d_boxfirst, n + 1, s));
		GD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, d_sumlen, d_lenfirst, n + 1, s));
		if ((rc = gd_grow(ctx, ctx->m_srscan, std::max(b1, b2) + 64))) return rc;
		size_t bb = ctx->m_srscan.cap;
		GD_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->m_srscan.p, bb, d_cnt, d_boxfirst, n + 1, s));
		bb = ctx->m_srscan.cap;
		GD_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->m_srscan.p, bb, d_sumlen, d_lenfirst, n + 1, s));
		// the two totals and the failure count: 24 bytes, the only thing the host waits for before it can size the batch's buffers
		struct { int32_t nb; int32_t pad; int64_t bases; MapSrTotals t; } h_tot;
		GD_HIP(hipMemcpyAsync(&h_tot.nb, d_boxfirst + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
		GD_HIP(hipMemcpyAsync(&h_tot.bases, d_lenfirst + n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
		GD_HIP(hipMemcpyAsync(&h_tot.t, d_tot, sizeof(MapSrTotals), hipMemcpyDeviceToHost, s));
		GD_HIP(gd_stream_wait(ctx, s));
		ctx->stage_s[1] += gd_now() - t0, t0 = gd_now();
		mark("vote");
		nb = h_tot.nb;
		ctx->failed_last = h_tot.t.n_failed, ctx->failed_total += h_tot.t.n_failed;
		if (h_tot.t.n_failed) {
			char msg[160];
			snprintf(msg, sizeof msg, "%lld read(s) of the batch left unmapped: degenerate DP box (candidate window outside the read / contig), last: read %d of the call",
			         (long long)h_tot.t.n_failed, h_tot.t.last_failed);
			ctx->warn = msg;
		}
		if ((rc = aux_layout())) return rc;
		if (nb > 0) {
			if ((rc = gd_grow(ctx, ctx->m_boxes, sizeof(MapBox) * (size_t)nb))) return rc;
			if ((rc = gd_grow(ctx, ctx->m_srcand, sizeof(GdCandBox) * (size_t)nb + sizeof(int64_t) * ((size_t)nb + 1) + sizeof(int32_t) * (size_t)nb + 256))) return rc;
			GdCandBox *d_cand = (GdCandBox *)ctx->m_srcand.p;
			int64_t *d_qoff = (int64_t *)(((uintptr_t)(d_cand + nb) + 63) & ~(uintptr_t)63);
			int32_t *d_bw = (int32_t *)(d_qoff + nb + 1);
			MapSrFillOut FO = {(MapBox *)ctx->m_boxes.p, d_cand, d_qoff, d_coff, d_bw, d_ex};
			hipLaunchKernelGGL(map_sr_fill_kernel, dim3((n + 1 + 63) / 64), dim3(64), 0, s, n, d_roff, (const int32_t *)d_cnt, (const int32_t *)d_boxfirst, (const int64_t *)d_lenfirst,
			                   (const GdCandBox *)ctx->m_srbox.p, slots, (const uint32_t *)ix->d_seq_len, (const uint64_t *)ix->d_seq_off, R.n_seq, D.sr, FO);
			// host copies of what the planner and the record stage read
			if ((rc = gd_host_grow(ctx, ctx->h_cand, sizeof(GdCandBox) * (size_t)nb))) return rc;
			cflat = (GdCandBox *)ctx->h_cand.p;
			qoff.resize(nb + 1), coff.resize(nb + 1), bw.resize(nb), ex.resize(nb);
			GD_HIP(hipMemcpyAsync(cflat, d_cand, sizeof(GdCandBox) * (size_t)nb, hipMemcpyDeviceToHost, s));
			GD_HIP(hipMemcpyAsync(qoff.data(), d_qoff, sizeof(int64_t) * ((size_t)nb + 1), hipMemcpyDeviceToHost, s));
			GD_HIP(hipMemcpyAsync(coff.data(), d_coff, sizeof(int64_t) * ((size_t)nb + 1), hipMemcpyDeviceToHost, s));
			GD_HIP(hipMemcpyAsync(bw.data(), d_bw, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, s));
			GD_HIP(hipMemcpyAsync(ex.data(), d_ex, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, s));
		} else qoff.assign(1, 0), coff.assign(1, 0);
		GD_HIP(hipMemcpyAsync(ccount.data(), d_cnt, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
		// the windows can be gathered while the tables travel (same stream: the gather kernel is queued behind the fill kernel)
		if (nb > 0) {
			if ((rc = gd_grow(ctx, ctx->m_q, (size_t)h_tot.bases + 64))) return rc;
			if ((rc = gd_grow(ctx, ctx->m_t, (size_t)h_tot.bases + 64))) return rc;
			if ((rc = gd_grow(ctx, ctx->m_cig, sizeof(uint32_t) * (2 * (size_t)h_tot.bases + 1)))) return rc;
			hipLaunchKernelGGL(map_gather_kernel, dim3(nb), dim3(64), 0, s, nb, (const MapBox *)ctx->m_boxes.p, d_reads, (const uint32_t *)ix->d_S,
			                   (uint8_t *)ctx->m_q.p, (uint8_t *)ctx->m_t.p);
		}
		GD_HIP(gd_stream_wait(ctx, s));
		for (int i = 0; i < n; ++i) box_first[i + 1] = box_first[i] + ccount[i];
		cfirst = box_first;
		if (box_first[n] != nb) { ctx->err = "box tables of the device and the host disagree"; return GDIET_E_HIP; }
		toff = qoff;
	} else {
	// only the head of every record can be in use: n_cand + at most AF_max_loc (ShortReads) / vt_nb_loc + 2 (LongReads) candidates;
	// the host copy is packed to that size (a full-size array would be 680 B per read: 178 MB to allocate and clear per 262 k short reads)
	const size_t vo_head = offsetof(MapVoteOut, cand) + sizeof(GdVt) * std::min<size_t>(is_sr ? (size_t)O.af_max_loc : (size_t)O.vt_nb_loc + 2, GDM_MAX_VT);
	if (ctx->h_vo.size() < vo_head * (size_t)n) ctx->h_vo.resize(vo_head * (size_t)n + (vo_head * (size_t)n >> 2));
	const uint8_t *vo_raw = ctx->h_vo.data();
	GD_HIP(hipMemcpy2DAsync(ctx->h_vo.data(), vo_head, ctx->m_voteout.p, sizeof(MapVoteOut), vo_head, (size_t)n, hipMemcpyDeviceToHost, s));
	GD_HIP(gd_stream_wait(ctx, s));
	ctx->stage_s[1] += gd_now() - t0, t0 = gd_now();
	mark("vote");
	// ---- G1b: linking + DP boxes (host threads) ------------------------------------------------------------------------
	// candidates of all reads in one flat array (capacity: what the vote kernel reported; the box stage may drop some).  Per-read
	// vectors would be allocated by the workers and released by this thread -- a quarter of a million cross-thread frees per
	// short-read batch, each contending for another thread's malloc arena.
	for (int i = 0; i < n; ++i) cfirst[i + 1] = cfirst[i] + (int)reinterpret_cast<const MapVoteOut *>(vo_raw + vo_head * (size_t)i)->n_cand;
	if ((rc = gd_host_grow(ctx, ctx->h_cand, sizeof(GdCandBox) * (size_t)cfirst[n]))) return rc;
	cflat = (GdCandBox *)ctx->h_cand.p; // entries [cfirst[i], cfirst[i] + ccount[i]) are written below, nothing else is read
	mark("g:prefix");
	gd_parallel_for(ctx, ctx->lane_threads, n, [&](int i) {
		const MapVoteOut &vo_i = *reinterpret_cast<const MapVoteOut *>(vo_raw + vo_head * (size_t)i); // head of the record only
		const unsigned nc = vo_i.n_cand;
		if (!nc) return;
		if (is_sr) { // straight into the flat array: a quarter of a million reads per batch, nothing allocated per read
			int k = 0;
			for (unsigned j = 0; j < nc; ++j)
				if (gd_sr_box_one(vo_i.cand[j], O, R, (uint32_t)(B.roff[i + 1] - B.roff[i]), cflat[(size_t)cfirst[i] + k])) ++k;
			ccount[i] = k;
			return;
		}
		std::vector<GdCand> C(nc);
		for (unsigned j = 0; j < nc; ++j) C[j].v = vo_i.cand[j];
		gd_lr_link_and_boxes(C, O, R, (uint32_t)(B.roff[i + 1] - B.roff[i]));
		ccount[i] = (int)std::min<size_t>(C.size(), nc);
		for (int j = 0; j < ccount[i]; ++j) cflat[(size_t)cfirst[i] + j] = gd_cand_box(C[j]);
	});
	// A DP box that lies outside its read or is absurdly large (a wrapped coordinate: the reference reads stale heap memory there, its
	// result is undefined) fails ITS READ -- it comes back unmapped (n_regs = 0) and is counted -- not the batch: a production run
	// must not be lost to one pathological read.  GDIET_FAULT_BOX=<read index> (fault injection for the tests: no read built so far
	// produces such a box) marks the boxes of that read of every batch as degenerate.
	{
		static const char *fault_env = getenv("GDIET_FAULT_BOX");
		const int fault = fault_env ? atoi(fault_env) : -1;
		int64_t n_failed = 0;
		int last_bad = -1;
		for (int i = 0; i < n; ++i) {
			const uint32_t rl = (uint32_t)(B.roff[i + 1] - B.roff[i]);
			bool bad = i == fault && ccount[i] > 0;
			for (int j = 0; j < ccount[i] && !bad; ++j) {
				const GdCandBox &c = cflat[(size_t)cfirst[i] + j];
				bad = c.qlen == 0 || c.tlen == 0 || c.qlen > rl || c.qseq_off + c.qlen > rl || c.tlen > 8u * rl + 100000u;
			}
			if (bad) ccount[i] = 0, ++n_failed, last_bad = i;
		}
		ctx->failed_last = n_failed, ctx->failed_total += n_failed;
		if (n_failed) {
			char msg[160];
			snprintf(msg, sizeof msg, "%lld read(s) of the batch left unmapped: degenerate DP box (candidate window outside the read / contig), last: read %d of the call",
			         (long long)n_failed, last_bad);
			ctx->warn = msg;
		}
	}
	for (int i = 0; i < n; ++i) box_first[i + 1] = box_first[i] + ccount[i];
	mark("g:boxes");
	nb = box_first[n];
	if ((rc = gd_host_grow(ctx, ctx->h_boxes, sizeof(MapBox) * (size_t)std::max(nb, 1)))) return rc;
	boxes = (MapBox *)ctx->h_boxes.p;
	qoff.assign(nb + 1, 0), toff.assign(nb + 1, 0), coff.assign(nb + 1, 0);
	bw.resize(nb), ex.resize(nb);
	// window offsets: a running sum over the boxes in batch order; the boxes themselves are filled by the host threads
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < ccount[i]; ++j) {
			const GdCandBox &c = cflat[(size_t)cfirst[i] + j];
			const int b = box_first[i] + j;
			qoff[b + 1] = qoff[b] + c.qlen, toff[b + 1] = toff[b] + c.tlen;
			coff[b + 1] = coff[b] + c.qlen + c.tlen;
		}
	mark("g:offsets");
	gd_parallel_for(ctx, ctx->lane_threads, n, [&](int i) {
		const uint32_t rl = (uint32_t)(B.roff[i + 1] - B.roff[i]);
		for (int j = 0; j < ccount[i]; ++j) {
			const GdCandBox &c = cflat[(size_t)cfirst[i] + j];
			const int b = box_first[i] + j;
			MapBox &M = boxes[b];
			M.read_off = B.roff[i], M.read_len = rl, M.qseq_off = c.qseq_off, M.qlen = c.qlen, M.tlen = c.tlen, M.rev = c.v.str;
			// a window hanging off a contig (or a wrapped coordinate) reads stale memory in the reference; here the part that
			// does not exist is zero-filled and absurd sizes are refused
			uint32_t avail = 0;
			uint64_t src = 0;
			if (c.target_id < R.n_seq && c.target_start < R.seq[c.target_id].len) {
				avail = std::min<uint32_t>(c.tlen, R.seq[c.target_id].len - c.target_start);
				src = R.seq[c.target_id].offset + c.target_start;
			}
			M.t_avail = avail, M.t_src = src;
			M.q_dst = qoff[b], M.t_dst = toff[b];
			bw[b] = is_sr ? (int32_t)gd_sr_bw((int)rl, D.sr) : (int32_t)O.bw, ex[b] = c.exact_score; // SR/map.c:624-631,925 ; LR/map.c:1800
		}
	});
	}
	ctx->stage_s[2] += gd_now() - t0, t0 = gd_now();
	mark("g:fill");
	// scores | CIGAR lengths come back in ONE copy with both ends 256-byte aligned.  (As two copies the second, starting at an address
	// that is not a multiple of 16, took the runtime's dword copy kernel with 1024-thread workgroups: 16 wavefronts that need four
	// free slots on every SIMD of one CU, which the DP kernel of the next batch -- five wavefronts per SIMD -- never leaves.  That
	// 37 KB copy then finished only in the tail of the other batch's DP kernel, 45-60 ms later (kernel trace, round 2): invisible in
	// the throughput with three batches in flight, but it is the batch's latency.  Pinned host buffers were tried as well and cost 8 %
	// of the step: the host stages write these tables.)
	nbp = ((size_t)std::max(nb, 1) + 63) & ~(size_t)63;
	if ((rc = gd_host_grow(ctx, ctx->h_res, sizeof(int32_t) * 2 * nbp))) return rc;
	int32_t *h_score = (int32_t *)ctx->h_res.p, *h_ncig = h_score + nbp;
	const bool post_dev = ctx->post_on_device != 0;
	if (post_dev && (rc = gd_host_grow(ctx, ctx->h_post, sizeof(GdPostOut) * (size_t)std::max(nb, 1)))) return rc;
	const GdPostOut *h_post = (const GdPostOut *)ctx->h_post.p;
	static const bool no_export = getenv("GDIET_POST_EXPORT") && atoi(getenv("GDIET_POST_EXPORT")) == 0;
	uint32_t *h_cig = nullptr;
	bool exported = false; // the DP results came to the host with map_post_kernel's own stores
	std::vector<int64_t> poff(1, 0);
	// an async lane shares its parent's backtrace arena (two whole-batch arenas do not fit in HBM, and concurrent DP kernels of
	// smaller batches measured slower): the DP stages take turns
	std::unique_lock<std::mutex> dp_lock;
	if (nb > 0) {
		if (!sr_dev) { // (device-side box stage: the tables are on the device already and the windows gathered)
			if ((rc = gd_grow(ctx, ctx->m_boxes, sizeof(MapBox) * nb))) return rc;
			if ((rc = gd_grow(ctx, ctx->m_q, (size_t)qoff[nb] + 64))) return rc;
			if ((rc = gd_grow(ctx, ctx->m_t, (size_t)toff[nb] + 64))) return rc;
			if ((rc = aux_layout())) return rc;
			if ((rc = gd_grow(ctx, ctx->m_cig, sizeof(uint32_t) * ((size_t)coff[nb] + 1)))) return rc;
			GD_HIP(hipMemcpyAsync(ctx->m_boxes.p, boxes, sizeof(MapBox) * nb, hipMemcpyHostToDevice, s));
			GD_HIP(hipMemcpyAsync(d_coff, coff.data(), sizeof(int64_t) * (nb + 1), hipMemcpyHostToDevice, s));
			GD_HIP(hipMemcpyAsync(d_ex, ex.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, s));
			hipLaunchKernelGGL(map_gather_kernel, dim3(nb), dim3(64), 0, s, nb, (const MapBox *)ctx->m_boxes.p, d_reads, (const uint32_t *)ix->d_S,
			                   (uint8_t *)ctx->m_q.p, (uint8_t *)ctx->m_t.p);
		}
		gdiet_ksw_score_t ks;
		ks.match = (int8_t)O.a, ks.mismatch = (int8_t)(O.b < 0 ? O.b : -O.b), ks.sc_ambi = 0, ks.q = (int8_t)O.q, ks.e = (int8_t)O.e, ks.q2 = (int8_t)O.q2, ks.e2 = (int8_t)O.e2;
		ks.reserved = 0, ks.flag = GDIET_EZ_APPROX_MAX;
		// In an async lane the DP stage (pre-filter, DP + backtrack kernels) runs on a stream of its own and is ENQUEUED behind the
		// DP stage of whichever lane used the shared arena last (arena_ev), so that consecutive DP kernels follow each other on the
		// GPU without a host round trip in between; dp_mu only orders the enqueueing.
		hipStream_t sd = s;
		if (ctx->parent) {
			sd = ctx->stream_dp;
			GD_HIP(hipEventRecord(ctx->gather_ev, s)); // windows gathered (and everything else queued on s)
			GD_HIP(hipStreamWaitEvent(sd, ctx->gather_ev, 0));
			dp_lock = std::unique_lock<std::mutex>(ctx->parent->dp_mu, std::defer_lock); // taken inside, after the planning
		}
		rc = gd_ksw_batch_dev(ctx, nb, (const uint8_t *)ctx->m_q.p, (const uint8_t *)ctx->m_t.p, d_ex, &ks, d_score, d_ncig, (uint32_t *)ctx->m_cig.p, d_coff,
		                      qoff.data(), toff.data(), bw.data(), sd, coff.data(), ex.data(), ctx->parent ? ctx->parent->arena_ev : nullptr,
		                      ctx->parent ? &dp_lock : nullptr);
		if (rc) { // kernels of this stage may already be queued in the shared arena: let them finish before the next lane takes its turn
			(void)hipStreamSynchronize(sd);
			return rc; // (the lock, if taken, is released by dp_lock's destructor)
		}
	mark("d:plan+enqueue");
		if (ctx->parent) {
			if (!ctx->own_arena) GD_HIP(hipEventRecord(ctx->parent->arena_ev, sd)); // the backtrack is done by then: the CIGARs sit in this lane's own buffer
			if (dp_lock.owns_lock()) dp_lock.unlock();
		}
		if (post_dev) { // P1 on the device, behind the backtrack of this stage (reads this lane's own windows and CIGAR slots, not the arena)
			if ((rc = gd_grow(ctx, ctx->m_post, sizeof(GdPostOut) * (size_t)nb))) { (void)hipStreamSynchronize(sd); return rc; }
			MapPostOpt PO;
			const int g_ = O.a, bb_ = O.b < 0 ? O.b : -O.b;
			for (int i = 0; i < 25; ++i) PO.mat[i] = (i / 5 == 4 || i % 5 == 4) ? 0 : (i / 5 == i % 5 ? (int8_t)g_ : (int8_t)bb_);
			PO.q = (int8_t)O.q, PO.e = (int8_t)O.e, PO.log_gap = !(O.flag & GD_F_SR);
			// A batch of wide-band alignments (the checkpointed kernels: four wavefronts of 128 registers per SIMD, i.e. no room for anything
			// else while they run): the device-to-host copies below are kernels of the runtime and would start only when DP wavefronts of
			// the NEXT batch retire -- ~1 s for 400 KB, and that second is part of the lane's cycle (kernel trace of the ONT path, round 2).
			// map_post_kernel starts in the gap between the two DP kernels, so it writes the results to page-locked host memory itself.
			// long alignments: one wavefront per alignment (mm_fix_cigar on lane 0, the walk over the bases by all lanes: 22 -> ~2 ms per HiFi
			// batch); short reads keep one alignment per thread (hundreds of thousands of 150-base walks).  GDIET_POST_WAVE=0 / 1 forces one.
			static const char *pw_env = getenv("GDIET_POST_WAVE");
			const bool post_wave = pw_env ? atoi(pw_env) != 0 : coff[nb] / std::max(nb, 1) >= 2000;
			const bool xport = (ctx->last_mask & 8) != 0 && !no_export;
			if (xport) {
				const size_t need = sizeof(GdPostOut) * (size_t)nb + sizeof(int32_t) * 2 * nbp + 256;
				if (need > ctx->h_pin.cap) {
					if (ctx->h_pin.p) (void)hipHostFree(ctx->h_pin.p);
					ctx->h_pin.p = nullptr, ctx->h_pin.cap = 0;
					if (hipHostMalloc(&ctx->h_pin.p, need + need / 2, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->h_pin.p = nullptr; }
					else ctx->h_pin.cap = need + need / 2;
				}
			}
			if (xport && ctx->h_pin.p) {
				int32_t *x_score = (int32_t *)ctx->h_pin.p, *x_ncig = x_score + nbp;
				GdPostOut *x_post = (GdPostOut *)(x_ncig + nbp);
				if (post_wave) {
					hipLaunchKernelGGL(map_fix_cigar_kernel, dim3((nb + 63) / 64), dim3(64), 0, sd, nb, (const MapBox *)ctx->m_boxes.p, (const uint8_t *)ctx->m_q.p,
					                   (const uint8_t *)ctx->m_t.p, (const int64_t *)d_coff, (uint32_t *)ctx->m_cig.p, d_ncig, (const int32_t *)d_score, x_post);
					hipLaunchKernelGGL(map_post_wave_kernel, dim3(nb), dim3(64), 0, sd, nb, (const MapBox *)ctx->m_boxes.p, (const uint8_t *)ctx->m_q.p,
					                   (const uint8_t *)ctx->m_t.p, (const int64_t *)d_coff, (uint32_t *)ctx->m_cig.p, d_ncig, (const int32_t *)d_score, PO, x_post, x_score, x_ncig);
				}
				else hipLaunchKernelGGL(map_post_kernel, dim3((nb + 63) / 64), dim3(64), 0, sd, nb, (const MapBox *)ctx->m_boxes.p, (const uint8_t *)ctx->m_q.p,
				                   (const uint8_t *)ctx->m_t.p, (const int64_t *)d_coff, (uint32_t *)ctx->m_cig.p, d_ncig, (const int32_t *)d_score, PO, x_post, x_score, x_ncig);
				h_score = x_score, h_ncig = x_ncig, h_post = x_post;
				exported = true;
			} else {
				if (post_wave) {
					hipLaunchKernelGGL(map_fix_cigar_kernel, dim3((nb + 63) / 64), dim3(64), 0, sd, nb, (const MapBox *)ctx->m_boxes.p, (const uint8_t *)ctx->m_q.p,
					                   (const uint8_t *)ctx->m_t.p, (const int64_t *)d_coff, (uint32_t *)ctx->m_cig.p, d_ncig, (const int32_t *)d_score, (GdPostOut *)ctx->m_post.p);
					hipLaunchKernelGGL(map_post_wave_kernel, dim3(nb), dim3(64), 0, sd, nb, (const MapBox *)ctx->m_boxes.p, (const uint8_t *)ctx->m_q.p,
					                   (const uint8_t *)ctx->m_t.p, (const int64_t *)d_coff, (uint32_t *)ctx->m_cig.p, d_ncig, (const int32_t *)d_score, PO, (GdPostOut *)ctx->m_post.p,
					                   (int32_t *)nullptr, (int32_t *)nullptr);
				}
				else hipLaunchKernelGGL(map_post_kernel, dim3((nb + 63) / 64), dim3(64), 0, sd, nb, (const MapBox *)ctx->m_boxes.p, (const uint8_t *)ctx->m_q.p,
				                   (const uint8_t *)ctx->m_t.p, (const int64_t *)d_coff, (uint32_t *)ctx->m_cig.p, d_ncig, (const int32_t *)d_score, PO, (GdPostOut *)ctx->m_post.p,
				                   (int32_t *)nullptr, (int32_t *)nullptr);
				GD_HIP(hipMemcpyAsync(ctx->h_post.p, ctx->m_post.p, sizeof(GdPostOut) * (size_t)nb, hipMemcpyDeviceToHost, sd));
			}
		}
		if (!exported) GD_HIP(hipMemcpyAsync(h_score, d_score, sizeof(int32_t) * 2 * nbp, hipMemcpyDeviceToHost, sd));
		GD_HIP(gd_stream_wait(ctx, sd));
	mark("d:wait");
		// CIGARs are short compared with their capacity (qlen+tlen): pack them on the device, then one copy
		for (int b = 0; b < nb; ++b) if (h_ncig[b] > coff[b + 1] - coff[b]) { ctx->err = "CIGAR capacity exceeded"; return GDIET_E_CIGAR_CAP; }
		poff.assign(nb + 1, 0);
		for (int b = 0; b < nb; ++b) poff[b + 1] = poff[b] + std::max(h_ncig[b], 0);
		if ((rc = gd_host_grow(ctx, ctx->h_cig, sizeof(uint32_t) * ((size_t)poff[nb] + 1)))) return rc;
		h_cig = (uint32_t *)ctx->h_cig.p;
		if (poff[nb] > 0) {
			if ((rc = gd_grow(ctx, ctx->m_pack, sizeof(uint32_t) * (size_t)poff[nb] + sizeof(int64_t) * (nb + 1) + 64))) return rc;
			int64_t *d_poff = (int64_t *)ctx->m_pack.p;
			uint32_t *d_packed = (uint32_t *)(d_poff + nb + 1);
			GD_HIP(hipMemcpyAsync(d_poff, poff.data(), sizeof(int64_t) * (nb + 1), hipMemcpyHostToDevice, s));
			hipLaunchKernelGGL(map_pack_cigar_kernel, dim3(nb), dim3(64), 0, s, nb, (const uint32_t *)ctx->m_cig.p, (const int64_t *)d_coff, (const int64_t *)d_poff, d_packed);
			GD_HIP(hipMemcpyAsync(h_cig, d_packed, sizeof(uint32_t) * (size_t)poff[nb], hipMemcpyDeviceToHost, s));
			GD_HIP(gd_stream_wait(ctx, s));
		}
	}
	if (dp_lock.owns_lock()) dp_lock.unlock();
	ctx->stage_s[3] += gd_now() - t0, t0 = gd_now();
	mark("d:pack");
	// ---- P1-P3 (host threads) ---------------------------------------------------------------------------------------------
	static std::atomic<uint64_t> call_counter{0};
	const uint64_t call_id = ++call_counter; // names the slabs of this call (GdRegSlab)
	std::atomic<int> no_mem{0};
	// (chunks of reads: the worker's scratch -- thread_local vectors, each access a call in a shared library -- is looked up once per chunk,
	// not six times per read; a short-read batch runs this body a quarter of a million times: chunks of 256 there)
	const int post_chunk = std::max(16, std::min(256, n / std::max(1, 8 * ctx->lane_threads))), n_post_chunks = (n + post_chunk - 1) / post_chunk; // (a long-read batch of 5 120 reads still spreads over every thread)
	gd_parallel_for(ctx, ctx->lane_threads, n_post_chunks, [&](int ch) {
		// scratch of the worker thread, reused from read to read (the workers are persistent)
		static thread_local std::vector<GdCand> C_tl;
		static thread_local std::vector<uint8_t> rev_tl;
		static thread_local std::vector<GdDpResult> dp_tl;
		static thread_local std::vector<GdReg> out_tl;
		std::vector<GdCand> &C = C_tl;
		std::vector<uint8_t> &rev = rev_tl;
		std::vector<GdDpResult> &dp = dp_tl;
		std::vector<GdReg> &out = out_tl;
		GdRegCursor &cursor = gd_reg_cursor();
		const int i_end = std::min(n, (ch + 1) * post_chunk);
		for (int i = ch * post_chunk; i < i_end; ++i) {
			n_regs[i] = 0, regs[i] = nullptr;
			const size_t nc = (size_t)ccount[i];
			if (!nc) continue;
			// the records grow CIGARs: this thread's copy.  (ShortReads: gd_sr_finish reads the box fields only, which gd_cand_unbox sets all of --
			// the elements are reused as they are; LongReads: fresh records, concatenate_cigars works inside them)
			if (is_sr) C.resize(nc);
			else C.assign(nc, GdCand());
			for (size_t j = 0; j < nc; ++j) gd_cand_unbox(cflat[(size_t)cfirst[i] + j], C[j]);
			const uint32_t rl = (uint32_t)(B.roff[i + 1] - B.roff[i]);
			const uint8_t *enc = B.enc + B.roff[i];
			// the reverse-complemented read: for P1 on the host, else only where a reverse-strand candidate may be concatenated (P2)
			bool need_rev = false;
			for (auto &c : C) need_rev |= c.v.str != 0 && (!post_dev || c.next >= 0);
			if (need_rev) { rev.resize(rl); for (uint32_t j = 0; j < rl; ++j) rev[rl - 1 - j] = enc[j] ^ 3; }
			dp.resize(nc);
			for (size_t j = 0; j < nc; ++j) {
				const int b = box_first[i] + (int)j;
				dp[j].score = h_score[b], dp[j].n_cigar = h_ncig[b], dp[j].cigar = h_cig + poff[b];
			}
			out.clear();
			const GdPostOut *pre = post_dev ? h_post + box_first[i] : nullptr;
			if (is_sr) gd_sr_finish(C, dp, O, R, rl, enc, need_rev ? rev.data() : enc, out, pre);
			else gd_lr_finish(C, dp, O, R, rl, enc, need_rev ? rev.data() : enc, out, nullptr, pre);
			if (out.empty()) continue;
			bool head = false;
			void *at = gd_reg_slab_take(cursor, call_id, gd_regs_bytes(out), head);
			if (!at) { no_mem.store(1); continue; }
			gd_regs_fill(out, at, head, &n_regs[i], &regs[i]);
		}
	});
	if (no_mem.load()) { gdiet_hip_free_regs(n, n_regs, regs); ctx->err = "out of host memory for the records"; return GDIET_E_NOMEM; }
	ctx->stage_s[4] += gd_now() - t0;
	mark("post");
	if (trace) fprintf(stderr, "[gdiet stages, ms] lane=%p start=%.2f end=%.2f n=%d%s\n", (void *)ctx, 1e3 * fmod(tr_t0, 1000.0), 1e3 * fmod(gd_now(), 1000.0), n, tr_s.c_str());
	return GDIET_OK;
}


extern "C" int gdiet_hip_set_map_lanes(gdiet_ctx *ctx, int n)
{
	if (!ctx || n < 1 || n > 16) return GDIET_E_PARAM;
	ctx->map_lanes = n;
	return GDIET_OK;
}

extern "C" int gdiet_hip_map_uploaded(gdiet_ctx *ctx, const gdiet_index *ix, const gdiet_mapopt_t *copt, gdiet_read_batch *B,
                                      int32_t *n_regs, gdiet_reg_t **regs)
{
	if (!ctx || !ix || !copt || !B || !n_regs || !regs) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	const int n = B->n;
	for (int i = 0; i < 6; ++i) ctx->stage_s[i] = 0;
	if (n == 0) return GDIET_OK;
	GdMapOpt O;
	gd_opt_from_c(copt, ix, O);
	{ const int rc = gd_check_opt(ctx, O); if (rc) return rc; }
	ctx->last_was_async = false;
	for (int i = 0; i < GD_MAX_INFLIGHT; ++i)
		if (ctx->async_busy[i]) { ctx->err = "batches submitted with gdiet_hip_map_submit are still in flight"; return GDIET_E_PARAM; }

	const int lanes = std::max(1, std::min(ctx->map_lanes, (n + 255) / 256));
	if (lanes == 1) {
		// the backtrace arena of a whole batch is ~40 % of HBM: lanes left over from a pipelined call must give theirs back first
		for (gdiet_ctx *c : ctx->children) gdiet_hip_destroy(c);
		ctx->children.clear();
		ctx->lane_threads = ctx->host_threads;
		GdBatchView V = {n, B->roff.data(), B->enc.data(), (const uint8_t *)B->d_reads, (const int64_t *)B->d_roff};
		return gd_map_range(ctx, ix, O, V, n_regs, regs);
	}
	// Software pipeline over slices of the batch: every lane (a child context with its own stream, workspace and host threads)
	// runs the whole chain for one slice at a time, so the latency-bound kernels (seed, vote, backtrack), the transfers and the
	// host stages of one slice overlap with the DP kernel of the others.  No result depends on the slicing.
	if (ctx->arena.p) { // ... and the parent's whole-batch arena is not needed while the lanes hold their own
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipFree(ctx->arena.p);
		ctx->arena.p = nullptr, ctx->arena.cap = 0;
	}
	while ((int)ctx->children.size() < lanes) {
		gdiet_ctx *c = nullptr;
		int rc = gdiet_hip_init(&c, ctx->device);
		if (rc) { ctx->err = "cannot create a pipeline lane"; return rc; }
		c->kernel_mode = ctx->kernel_mode, c->seed_thread_kernel = ctx->seed_thread_kernel, c->spread = ctx->spread, c->bt_wave = ctx->bt_wave, c->dp_split = ctx->dp_split, c->fuse_bt = ctx->fuse_bt, c->vote_wave = ctx->vote_wave, c->wide_two_waves = ctx->wide_two_waves, c->wide_ckpt = ctx->wide_ckpt, c->post_on_device = ctx->post_on_device;
		ctx->children.push_back(c);
	}
	const int n_slices = std::min(n, lanes * ctx->slices_per_lane);
	std::atomic<int> next(0);
	std::vector<int> rcs(lanes, 0);
	std::vector<std::thread> th;
	for (int l = 0; l < lanes; ++l)
		th.emplace_back([&, l]() {
			gdiet_ctx *c = ctx->children[l];
			c->kernel_mode = ctx->kernel_mode;
			c->lane_threads = std::max(1, ctx->host_threads / lanes);
			double acc[6] = {0, 0, 0, 0, 0, 0};
			for (;;) {
				const int sl = next.fetch_add(1);
				if (sl >= n_slices || rcs[l]) break;
				const int lo = (int)((int64_t)n * sl / n_slices), hi = (int)((int64_t)n * (sl + 1) / n_slices);
				GdBatchView V = {hi - lo, B->roff.data() + lo, B->enc.data(), (const uint8_t *)B->d_reads, (const int64_t *)B->d_roff + lo};
				rcs[l] = gd_map_range(c, ix, O, V, n_regs + lo, regs + lo);
				for (int i = 0; i < 6; ++i) acc[i] += c->stage_s[i];
			}
			for (int i = 0; i < 6; ++i) c->stage_s[i] = acc[i];
		});
	for (auto &t : th) t.join();
	ctx->last_mask = 0, ctx->last_cells = 0, ctx->last_alg_bytes = 0, ctx->failed_last = 0;
	for (int l = 0; l < lanes; ++l) {
		gdiet_ctx *c = ctx->children[l];
		if (rcs[l]) { ctx->err = c->err; return rcs[l]; }
		if (c->failed_total) ctx->failed_last += c->failed_total, ctx->failed_total += c->failed_total, ctx->warn = c->warn, c->failed_total = 0; // (a lane maps several slices per call)
		for (int i = 0; i < 6; ++i) ctx->stage_s[i] += c->stage_s[i];
		ctx->last_mask |= c->last_mask;
	}
	return GDIET_OK;
}

// ---- several batches in flight ---------------------------------------------------------------------------------------------------
// gdiet_hip_map_submit starts the whole path for one resident batch on a lane of its own (stream, scratch) and returns;
// gdiet_hip_map_wait joins it.  The seeding / voting kernels and the host stages of one batch then run while the DP kernel of
// another owns the GPU; the DP stages themselves take turns in the shared backtrace arena.  The reference overlaps its mini-batches the same way (kt_pipeline, LR/map.c:2094-2170).  Results do not depend on it.
struct gdiet_map_ticket {
	std::thread th;
	int rc = 0, lane = 0;
	GdMapOpt O;
};

// gdiet_hip_destroy with tickets still open: their lane threads are joined (results dropped) before anything is released
static void gd_join_open_tickets(gdiet_ctx *ctx)
{
	std::vector<void *> open;
	{
		std::lock_guard<std::mutex> guard(ctx->async_mu);
		open.swap(ctx->open_tickets);
	}
	for (void *p : open) {
		gdiet_map_ticket *t = (gdiet_map_ticket *)p;
		if (t->th.joinable()) t->th.join();
		ctx->async_busy[t->lane] = false;
		delete t;
	}
}

extern "C" int gdiet_hip_set_inflight(gdiet_ctx *ctx, int n)
{
	if (!ctx || n < 1 || n > GD_MAX_INFLIGHT) return GDIET_E_PARAM;
	for (int i = 0; i < GD_MAX_INFLIGHT; ++i) if (ctx->async_busy[i]) { ctx->err = "batches are in flight"; return GDIET_E_PARAM; }
	gd_drop_async_lanes(ctx);
	ctx->async_depth = n, ctx->async_next = 0;
	// private backtrace arenas for the lanes when all of them fit beside the index and the batches' own buffers
	size_t free_b = 0, total_b = 0;
	(void)hipSetDevice(ctx->device);
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError(), total_b = 0;
	ctx->lane_arena_cap = n > 1 ? (size_t)((double)total_b * 0.70 / n) : 0;
	if (const char *e = getenv("GDIET_LANE_ARENA_GB")) ctx->lane_arena_cap = (size_t)(atof(e) * 1e9);
	return GDIET_OK;
}

extern "C" int gdiet_hip_map_submit(gdiet_ctx *ctx, const gdiet_index *ix, const gdiet_mapopt_t *copt, gdiet_read_batch *B, int32_t *n_regs,
                                    gdiet_reg_t **regs, gdiet_map_ticket **out)
{
	if (!ctx || !ix || !copt || !B || !n_regs || !regs || !out) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	std::lock_guard<std::mutex> guard(ctx->async_mu); // (submit and wait may come from different threads of the caller's pipeline)
	gdiet_map_ticket *t = new gdiet_map_ticket();
	gd_opt_from_c(copt, ix, t->O);
	int rc = gd_check_opt(ctx, t->O);
	if (rc) { delete t; return rc; }
	const int l = ctx->async_next % ctx->async_depth;
	if (ctx->async_busy[l]) { delete t; ctx->err = "too many batches in flight: wait for the oldest ticket first"; return GDIET_E_PARAM; }
	// the backtrace arenas are the big consumers of HBM: the intra-batch lanes give theirs back
	for (gdiet_ctx *c : ctx->children) gdiet_hip_destroy(c);
	ctx->children.clear();
	if (!ctx->async_lane[l]) {
		gdiet_ctx *c = nullptr;
		rc = gdiet_hip_init(&c, ctx->device);
		if (rc) { delete t; ctx->err = "cannot create an async lane"; return rc; }
		c->parent = ctx;
		ctx->async_lane[l] = c;
	}
	gdiet_ctx *c = ctx->async_lane[l];
	c->kernel_mode = ctx->kernel_mode, c->seed_thread_kernel = ctx->seed_thread_kernel, c->spread = ctx->spread, c->bt_wave = ctx->bt_wave, c->dp_split = ctx->dp_split, c->fuse_bt = ctx->fuse_bt, c->vote_wave = ctx->vote_wave, c->wide_two_waves = ctx->wide_two_waves, c->wide_ckpt = ctx->wide_ckpt, c->post_on_device = ctx->post_on_device;
	c->lane_threads = c->host_threads = ctx->host_threads; // all lanes draw from the parent's pool
	ctx->async_busy[l] = true, ctx->async_next++;
	t->lane = l;
	ctx->open_tickets.push_back(t);
	t->th = std::thread([=]() {
		GdBatchView V = {B->n, B->roff.data(), B->enc.data(), (const uint8_t *)B->d_reads, (const int64_t *)B->d_roff};
		t->rc = B->n ? gd_map_range(c, ix, t->O, V, n_regs, regs) : GDIET_OK;
	});
	*out = t;
	return GDIET_OK;
}

extern "C" int gdiet_hip_map_wait(gdiet_ctx *ctx, gdiet_map_ticket *t)
{
	if (!ctx || !t) return GDIET_E_PARAM;
	if (t->th.joinable()) t->th.join();
	std::lock_guard<std::mutex> guard(ctx->async_mu);
	for (size_t i = 0; i < ctx->open_tickets.size(); ++i)
		if (ctx->open_tickets[i] == t) { ctx->open_tickets.erase(ctx->open_tickets.begin() + i); break; }
	gdiet_ctx *c = ctx->async_lane[t->lane];
	const int rc = t->rc;
	if (rc) ctx->err = c->err;
	ctx->failed_last = c->failed_last, ctx->failed_total += c->failed_last;
	if (c->failed_last) ctx->warn = c->warn;
	for (int i = 0; i < 6; ++i) ctx->stage_s[i] = c->stage_s[i];
	ctx->last_mask = c->last_mask, ctx->last_cells = c->last_cells, ctx->last_alg_bytes = c->last_alg_bytes;
	if (!rc && c->last_cells) { // the lane's DP-stage events of this batch (its streams are idle: the thread has joined)
		ctx->last_was_async = false;
		if (gdiet_hip_last_kernel_ms(c, &ctx->async_dp_ms, &ctx->async_bt_ms) == GDIET_OK) ctx->last_was_async = true;
	}
	ctx->async_busy[t->lane] = false;
	delete t;
	return rc;
}

extern "C" int gdiet_hip_map_batch(gdiet_ctx *ctx, const gdiet_index *ix, const gdiet_mapopt_t *opt, int n, const char *const *seqs,
                                   const int32_t *lens, int32_t *n_regs, gdiet_reg_t **regs)
{
	gdiet_read_batch *b = nullptr;
	int rc = gdiet_hip_batch_upload(ctx, &b, n, seqs, lens);
	if (rc) return rc;
	rc = gdiet_hip_map_uploaded(ctx, ix, opt, b, n_regs, regs);
	gdiet_hip_batch_destroy(ctx, b);
	return rc;
}

// B2: the per-read level of the boundary, mm_map_frag's call shape (LR/minimap.h:390; LR/map.c:1273): one fragment of n_segs segments.
// As in the reference only segment 0 is sketched and aligned (SURVEY bug-compatibility item 7); the other segments get no records.
extern "C" int gdiet_hip_map_frag(gdiet_ctx *ctx, const gdiet_index *ix, const gdiet_mapopt_t *opt, int n_segs, const int32_t *qlens, const char *const *seqs,
                                  int32_t *n_regs, gdiet_reg_t **regs)
{
	if (!ctx || !ix || !opt || n_segs < 1 || !qlens || !seqs || !n_regs || !regs) return GDIET_E_PARAM;
	for (int i = 0; i < n_segs; ++i) n_regs[i] = 0, regs[i] = nullptr;
	if (qlens[0] <= 0) return GDIET_OK; // (LR/map.c:1288: qlen_sum == 0 returns without records)
	return gdiet_hip_map_batch(ctx, ix, opt, 1, seqs, qlens, n_regs, regs);
}

// B4: the sketch / seed level of the boundary for a batch of reads -- mm_sketch2 + mm_get_shift, mm_sketch3, mm_seed_mz_flt,
// mm_collect_matches2 (LR/mmpriv.h:65-76; LR/map.c:1296-1325) -- i.e. the output of the seeding kernel, copied out.
extern "C" int gdiet_hip_seed_batch(gdiet_ctx *ctx, const gdiet_index *ix, const gdiet_mapopt_t *copt, int n, const char *const *seqs, const int32_t *lens,
                                    int32_t *shift, uint32_t *tmp_extracted_len, uint32_t *n_mv, int64_t *seed_off, int64_t *occ_off, gdiet_seed_t **seeds,
                                    uint64_t **occ)
{
	if (!ctx || !ix || !copt || n < 0 || (n && (!seqs || !lens)) || !shift || !tmp_extracted_len || !n_mv || !seed_off || !occ_off || !seeds || !occ) return GDIET_E_PARAM;
	*seeds = nullptr, *occ = nullptr;
	seed_off[0] = occ_off[0] = 0;
	if (n == 0) return GDIET_OK;
	(void)hipSetDevice(ctx->device);
	for (int i = 0; i < GD_MAX_INFLIGHT; ++i)
		if (ctx->async_busy[i]) { ctx->err = "batches submitted with gdiet_hip_map_submit are still in flight"; return GDIET_E_PARAM; }
	GdMapOpt O;
	gd_opt_from_c(copt, ix, O);
	int rc = gd_check_opt(ctx, O);
	if (rc) return rc;
	{ std::string e; if (!gd_index_fetch_host(const_cast<gdiet_index *>(ix), e)) { ctx->err = e; return GDIET_E_HIP; } } // the occurrence lists are copied from the host tables
	gdiet_read_batch *b = nullptr;
	if ((rc = gdiet_hip_batch_upload(ctx, &b, n, seqs, lens))) return rc;
	for (gdiet_ctx *c : ctx->children) gdiet_hip_destroy(c);
	ctx->children.clear();
	ctx->lane_threads = ctx->host_threads;
	GdSeedExport sx;
	GdBatchView V = {n, b->roff.data(), b->enc.data(), (const uint8_t *)b->d_reads, (const int64_t *)b->d_roff};
	rc = gd_map_range(ctx, ix, O, V, nullptr, nullptr, &sx);
	gdiet_hip_batch_destroy(ctx, b);
	if (rc) return rc;
	for (int i = 0; i < n; ++i) {
		const MapSeedOut &o = sx.out[i];
		shift[i] = o.shift, tmp_extracted_len[i] = o.tel, n_mv[i] = o.n_mv;
		seed_off[i + 1] = seed_off[i] + (o.n_seeds > 0 ? o.n_seeds : 0), occ_off[i + 1] = occ_off[i] + (o.n_seeds > 0 ? o.n_a : 0);
	}
	gdiet_seed_t *sd = (gdiet_seed_t *)malloc(sizeof(gdiet_seed_t) * (size_t)std::max<int64_t>(seed_off[n], 1));
	uint64_t *oc = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(occ_off[n], 1));
	if (!sd || !oc) { free(sd), free(oc); ctx->err = "out of host memory"; return GDIET_E_NOMEM; }
	int64_t at = 0;
	for (int64_t j = 0; j < seed_off[n]; ++j) {
		const GdSeed &g = sx.seeds[(size_t)j];
		sd[j].n = g.n, sd[j].q_pos = g.q_pos;
		for (uint32_t t = 0; t < g.n; ++t) oc[at++] = ix->h.pos[(size_t)g.start + t];
	}
	if (at != occ_off[n]) { free(sd), free(oc); ctx->err = "seed tables disagree"; return GDIET_E_HIP; }
	*seeds = sd, *occ = oc;
	return GDIET_OK;
}

extern "C" size_t gdiet_hip_sam_record(const gdiet_index *ix, const char *qname, const char *seq, const char *qual, int32_t l_seq,
                                       const gdiet_reg_t *regs, int32_t n_regs, int32_t reg_idx, int64_t opt_flag, char *buf, size_t cap)
{
	if (!ix) return 0;
	std::vector<GdReg> v(n_regs > 0 ? n_regs : 0);
	for (int i = 0; i < n_regs; ++i) {
		const gdiet_reg_t &r = regs[i];
		GdReg &g = v[i];
		g.id = r.id, g.cnt = r.cnt, g.rid = r.rid, g.score = r.score, g.qs = r.qs, g.qe = r.qe, g.rs = r.rs, g.re = r.re, g.parent = r.parent, g.subsc = r.subsc;
		g.mlen = r.mlen, g.blen = r.blen, g.mapq = r.mapq, g.rev = r.rev, g.sam_pri = r.sam_pri, g.dp_score = r.dp_score, g.dp_max = r.dp_max, g.n_ambi = r.n_ambi;
		g.has_p = true, g.cigar.assign(r.cigar, r.cigar + r.n_cigar);
	}
	std::string s;
	gd_write_sam(s, ix->h.ref(), qname, seq, qual, l_seq, v, reg_idx, opt_flag);
	if (buf && cap) { const size_t m = std::min(cap - 1, s.size()); memcpy(buf, s.data(), m); buf[m] = 0; }
	return s.size();
}

// All SAM records of a batch, in input order, formatted on the context's host threads (step 2 of the reference's pipeline prints
// them one read at a time on one thread: LR/map.c:2139-2170 -> mm_write_sam3).  A read without alignments gives its unmapped
// record; with MM_F_NO_PRINT_2ND secondary records are skipped, exactly as the reference's output loop does.
// chunk strings of the batch formatters, kept between calls (a chunk of HiFi records is megabytes: allocated fresh, every call pays
// for its pages again -- 300 MB of first touches per 5 120-read mini-batch)
static void gd_fmt_take(gdiet_ctx *ctx, std::vector<std::string> &rec)
{
	std::lock_guard<std::mutex> lk(ctx->enc_mu);
	for (std::string &r : rec) {
		if (ctx->fmt_pool.empty()) break;
		r.swap(ctx->fmt_pool.back()), ctx->fmt_pool.pop_back();
		r.clear();
	}
}
static void gd_fmt_give(gdiet_ctx *ctx, std::vector<std::string> &rec)
{
	std::lock_guard<std::mutex> lk(ctx->enc_mu);
	for (std::string &r : rec)
		if (ctx->fmt_pool.size() < 256 && r.capacity() > 4096) ctx->fmt_pool.emplace_back(), ctx->fmt_pool.back().swap(r);
}

// reads per chunk of the batch formatters: a few chunks per host thread, so that a mini-batch of 5 120 long reads keeps every thread busy
// (chunks of 512 reads gave ten), but never so small that a quarter of a million short reads become tens of thousands of strings
static int gd_fmt_chunk(const gdiet_ctx *ctx, int n_reads)
{
	const int want = n_reads / std::max(1, 6 * ctx->host_threads);
	return std::max(16, std::min(512, want));
}

static size_t gd_sam_batch_impl(gdiet_ctx *ctx, const gdiet_index *ix, int n_reads, const char *const *qnames, const char *const *seqs,
                                const char *const *quals, const int32_t *lens, const int32_t *n_regs, gdiet_reg_t *const *regs,
                                int64_t opt_flag, char **buf_io, size_t *cap_io /* null: *buf_io is malloc'd to size */)
{
	const int CH = gd_fmt_chunk(ctx, n_reads), n_ch = (n_reads + CH - 1) / CH;
	std::vector<std::string> rec((size_t)n_ch);
	gd_fmt_take(ctx, rec);
	gd_parallel_for(ctx, ctx->host_threads, n_ch, [&](int c) {
		static thread_local std::vector<GdReg> v; // per-thread scratch, reused from read to read
		std::string &s = rec[c];
		const int i1 = std::min(n_reads, (c + 1) * CH);
		size_t guess = 0;
		for (int i = c * CH; i < i1; ++i) guess += (2 * (size_t)lens[i] + 200) * (size_t)std::max(1, n_regs[i]);
		s.reserve(guess);
		for (int i = c * CH; i < i1; ++i) {
			const int nr = n_regs[i];
			v.resize(nr > 0 ? nr : 0);
			for (int j = 0; j < nr; ++j) {
				const gdiet_reg_t &r = regs[i][j];
				GdReg &g = v[j];
				g.id = r.id, g.cnt = r.cnt, g.rid = r.rid, g.score = r.score, g.qs = r.qs, g.qe = r.qe, g.rs = r.rs, g.re = r.re, g.parent = r.parent, g.subsc = r.subsc;
				g.mlen = r.mlen, g.blen = r.blen, g.mapq = r.mapq, g.rev = r.rev, g.sam_pri = r.sam_pri, g.dp_score = r.dp_score, g.dp_max = r.dp_max, g.n_ambi = r.n_ambi;
				g.has_p = true, g.cigar.assign(r.cigar, r.cigar + r.n_cigar);
			}
			const char *q = quals ? quals[i] : nullptr;
			if (nr <= 0) gd_write_sam(s, ix->h.ref(), qnames[i], seqs[i], q, lens[i], v, -1, opt_flag), s += '\n'; // (the writer appends)
			else
				for (int j = 0; j < nr; ++j) {
					if ((opt_flag & GD_F_NO_PRINT_2ND) && v[j].id != v[j].parent) continue;
					gd_write_sam(s, ix->h.ref(), qnames[i], seqs[i], q, lens[i], v, j, opt_flag), s += '\n';
				}
		}
	});
	std::vector<size_t> at((size_t)n_ch + 1, 0);
	for (int c = 0; c < n_ch; ++c) at[c + 1] = at[c] + rec[c].size();
	const size_t tot = at[n_ch];
	char *buf;
	if (cap_io) { // the caller's buffer, grown when it is too small
		buf = *buf_io;
		if (!buf || *cap_io < tot + 1) {
			const size_t cap = tot + 1 + (tot >> 3);
			char *nb = (char *)realloc(buf, cap);
			if (!nb) { gd_fmt_give(ctx, rec); return 0; }
			buf = nb, *buf_io = nb, *cap_io = cap;
		}
	} else {
		buf = (char *)malloc(tot + 1);
		if (!buf) { gd_fmt_give(ctx, rec); return 0; }
		*buf_io = buf;
	}
	gd_parallel_for(ctx, ctx->host_threads, n_ch, [&](int c) { memcpy(buf + at[c], rec[c].data(), rec[c].size()); });
	buf[tot] = 0;
	gd_fmt_give(ctx, rec);
	return tot;
}

extern "C" size_t gdiet_hip_sam_batch(gdiet_ctx *ctx, const gdiet_index *ix, int n_reads, const char *const *qnames, const char *const *seqs,
                                      const char *const *quals, const int32_t *lens, const int32_t *n_regs, gdiet_reg_t *const *regs,
                                      int64_t opt_flag, char **out)
{
	if (!ctx || !ix || n_reads < 0 || !qnames || !seqs || !lens || !n_regs || !regs || !out) return 0;
	*out = nullptr;
	return gd_sam_batch_impl(ctx, ix, n_reads, qnames, seqs, quals, lens, n_regs, regs, opt_flag, out, nullptr);
}

// the same into a buffer the caller keeps from mini-batch to mini-batch (*buf / *cap: start with NULL / 0; realloc'd when too small; free()
// it at the end): the text of a long-read mini-batch is ~150 MB, and a fresh allocation of that size is paid for page by page every time
extern "C" size_t gdiet_hip_sam_batch_into(gdiet_ctx *ctx, const gdiet_index *ix, int n_reads, const char *const *qnames, const char *const *seqs,
                                           const char *const *quals, const int32_t *lens, const int32_t *n_regs, gdiet_reg_t *const *regs,
                                           int64_t opt_flag, char **buf, size_t *cap)
{
	if (!ctx || !ix || n_reads < 0 || !qnames || !seqs || !lens || !n_regs || !regs || !buf || !cap) return 0;
	return gd_sam_batch_impl(ctx, ix, n_reads, qnames, seqs, quals, lens, n_regs, regs, opt_flag, buf, cap);
}

// All PAF lines of a batch (mm_write_paf3, LR/format.c:326-367, as step 2 prints them when MM_F_OUT_SAM is off: LR/map.c:2163-2185).
// opt_flag: MM_F_OUT_CG adds the cg:Z: tag, MM_F_PAF_NO_HIT the lines of unmapped reads, MM_F_NO_PRINT_2ND drops secondary records.
extern "C" size_t gdiet_hip_paf_batch(gdiet_ctx *ctx, const gdiet_index *ix, int n_reads, const char *const *qnames, const int32_t *lens,
                                      const int32_t *n_regs, gdiet_reg_t *const *regs, int64_t opt_flag, char **out)
{
	if (!ctx || !ix || n_reads < 0 || !qnames || !lens || !n_regs || !regs || !out) return 0;
	*out = nullptr;
	const int CH = 512, n_ch = (n_reads + CH - 1) / CH;
	std::vector<std::string> rec((size_t)n_ch);
	gd_parallel_for(ctx, ctx->host_threads, n_ch, [&](int c) {
		static thread_local std::vector<GdReg> v;
		std::string &s = rec[c];
		const int i1 = std::min(n_reads, (c + 1) * CH);
		for (int i = c * CH; i < i1; ++i) {
			const int nr = n_regs[i];
			v.resize(nr > 0 ? nr : 0);
			for (int j = 0; j < nr; ++j) {
				const gdiet_reg_t &r = regs[i][j];
				GdReg &g = v[j];
				g.id = r.id, g.cnt = r.cnt, g.rid = r.rid, g.score = r.score, g.qs = r.qs, g.qe = r.qe, g.rs = r.rs, g.re = r.re, g.parent = r.parent, g.subsc = r.subsc;
				g.mlen = r.mlen, g.blen = r.blen, g.mapq = r.mapq, g.rev = r.rev, g.sam_pri = r.sam_pri, g.dp_score = r.dp_score, g.dp_max = r.dp_max, g.n_ambi = r.n_ambi;
				g.has_p = true, g.cigar.assign(r.cigar, r.cigar + r.n_cigar);
			}
			if (nr <= 0) {
				if (opt_flag & GD_F_PAF_NO_HIT) gd_write_paf(s, ix->h.ref(), qnames[i], lens[i], v, -1, opt_flag), s += '\n';
			} else
				for (int j = 0; j < nr; ++j) {
					if ((opt_flag & GD_F_NO_PRINT_2ND) && v[j].id != v[j].parent) continue;
					gd_write_paf(s, ix->h.ref(), qnames[i], lens[i], v, j, opt_flag), s += '\n';
				}
		}
	});
	size_t tot = 0;
	for (const std::string &r : rec) tot += r.size();
	char *buf = (char *)malloc(tot + 1);
	if (!buf) return 0;
	size_t at = 0;
	for (const std::string &r : rec) memcpy(buf + at, r.data(), r.size()), at += r.size();
	buf[tot] = 0;
	*out = buf;
	return tot;
}
