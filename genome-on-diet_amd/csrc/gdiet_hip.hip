// C-ABI shim (include/gdiet_hip.h) over the HIP kernels.  Host side is plain C++; no torch types anywhere.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <chrono>
#include <deque>
#include <unordered_map>
#include <memory>
#include <sched.h>

#include "../../include/gdiet_hip.h"
#include "ksw_common.h"
#include "ksw_generic.hip.h"
#include "ksw_backtrack.hip.h"
#include "ksw_wave.hip.h"
#include "ksw_pipe.hip.h"
#include "ksw_extz2_exact.hip.h"
#include "ksw_exts2.hip.h"

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
};

#define GD_MAX_INFLIGHT 8 // batches in flight per context (gdiet_hip_set_inflight)
struct gdiet_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	std::string err;
	char name[256] = {0};
	int kernel_mode = 0;
	int last_mask = 0;
	DevBuf arena;               // backtrace matrices
	DevBuf tasks, ids, status;  // per-batch descriptors
	DevBuf diag;                // per-alignment score of the main diagonal (ksw_exact_match_kernel -> ksw_backtrack_kernel)
	DevBuf pipes, pipe_runs, pipe_dst; // PipeWave / PipeRun records of the batch and the compacted id lists of its runs (ksw_pipe.hip.h)
	DevBuf qseq, tseq, score, ncig, cigar; // host-API staging
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	// head / tail split of a big DP launch (see gdiet_hip_ksw_extd2_batch_dev)
	hipStream_t stream2 = nullptr;
	hipStream_t stream_dp = nullptr;   // stream of the DP stage in an async lane (GDIET_DP_PRIORITY=1 raises its priority)
	hipEvent_t ev2[3] = {nullptr, nullptr, nullptr}; // go, DP of the tail done, tail done
	int dp_split = 1, wave_slots = 5120, last_split = 0;
	int dp_waves = 5;                  // wavefronts per SIMD the 64-lane DP kernel is launched for (gdiet_hip_set_dp_waves / GDIET_DP_WAVES: 5 or 4)
	int wide_ckpt = -1;                // GDIET_WIDE_CKPT: 1 / 0 force / forbid the checkpointed wide-band kernel, default by batch size
	int wide_two_waves = -1;           // GDIET_WIDE_TWO_WAVES: 1 / 0 force the two-wavefront / two-blocks-per-lane kernel for wide bands, default by count
	int vote_wave = 1;                 // GDIET_VOTE_WAVE=0: the sequential vote kernel for long reads too
	int index_on_device = 1;           // GDIET_INDEX_BUILD=host: gdiet_hip_index_build sketches and sorts on host threads instead
	int post_on_device = 1;            // GDIET_POST=host: mm_fix_cigar / mm_update_extra on host threads instead of map_post_kernel
	int fuse_bt = 1;                   // GDIET_FUSE_BT=0: the 64-lane kernel leaves the backtrack to the separate kernel
	bool single_affine = false;        // set for the duration of a gdiet_hip_ksw_extz2_batch call: single-affine kernel variants
	std::vector<int32_t> h_ids;
	std::vector<PipeWave> h_pipes;
	std::vector<PipeRun> h_pipe_runs;
	// per-read mapping path (map_pipeline.hip.h)
	DevBuf m_sc, m_mv, m_u64, m_seed, m_seedout, m_voteout, m_hitoff, m_hits, m_boxes, m_q, m_t, m_aux, m_cig, m_pack, m_post, m_seedids;
	DevBuf m_srbox, m_srtab, m_srscan, m_srcand; // device-side box stage of the ShortReads variant (map_pipeline.hip.h)
	int sr_boxes_on_device = 1;        // GDIET_SR_BOXES=host: candidate geometry of the ShortReads variant on host threads instead
	std::vector<uint8_t> h_vo;  // host copy of the vote records' heads, kept between batches
	DevBuf h_pin; // page-locked: scores, CIGAR lengths and P1 results of a wide-band batch, written by map_post_kernel itself
	DevBuf h_boxes, h_cand, h_tasks, h_seedout, h_res, h_cig, h_post; // HOST buffers kept between batches (gd_host_grow): the per-batch tables of a
	                                                           // short-read batch are tens of MB each, and allocated fresh they cost page faults
	int host_threads = 8;
	int lane_threads = 8;              // host threads this lane may use inside gd_map_range
	void *pool = nullptr;              // GdPool (map_pipeline.hip.h), created on first use
	// batches in flight (gdiet_hip_map_submit / _wait): up to GD_MAX_INFLIGHT lane contexts (own stream and scratch) that share THIS
	// context's backtrace arena, one DP stage at a time
	gdiet_ctx *parent = nullptr;       // set in an async lane
	std::mutex dp_mu;                  // orders the lanes' DP stages: held while one ENQUEUES its stage behind arena_ev
	hipEvent_t gather_ev = nullptr;    // this lane's windows are gathered (its DP stream waits for it)
	hipEvent_t wait_ev = nullptr;      // hipEventBlockingSync: the long waits of the mapping path sleep instead of spinning (gd_stream_wait)
	int blocking_wait = 0;             // GDIET_SYNC=block: wait on a hipEventBlockingSync event instead of hipStreamSynchronize
	hipEvent_t arena_ev = nullptr;     // recorded after the last DP stage that was enqueued: the arena is free once it has completed
	size_t lane_arena_cap = 0;         // a lane whose batch needs no more backtrace than this works in an arena of its own (set with the depth)
	bool own_arena = false;            // (lane) the last DP stage did
	bool shared_sticky = false;        // (lane) a recent batch did not fit a private arena
	gdiet_ctx *async_lane[GD_MAX_INFLIGHT] = {};
	bool async_busy[GD_MAX_INFLIGHT] = {};
	std::mutex async_mu;               // guards the ticket bookkeeping of submit / wait
	std::vector<void *> open_tickets;  // gdiet_map_ticket* submitted and not yet waited for (joined by gdiet_hip_destroy)
	std::vector<std::vector<uint8_t>> enc_pool; // host buffers of destroyed read batches, reused by the next uploads
	std::vector<std::string> fmt_pool;           // chunk strings of gdiet_hip_sam_batch / _paf_batch, reused (guarded by enc_mu)
	std::mutex enc_mu;
	int async_next = 0, async_depth = 2;
	bool last_was_async = false;       // gdiet_hip_last_kernel_ms then reports the lane's events, copied at gdiet_hip_map_wait
	float async_dp_ms = 0, async_bt_ms = 0;
	int map_lanes = 1;                 // software-pipeline depth of gdiet_hip_map_uploaded
	int slices_per_lane = 1;           // GDIET_SLICES_PER_LANE
	std::vector<gdiet_ctx *> children; // the lanes (child contexts on the same device)
	int seed_thread_kernel = 0;
	int bt_wave = 1;                   // GDIET_BT_WAVE=0: always the one-walk-per-thread backtrack kernel
	int spread = 1;                    // the serial vote kernel runs one read per wavefront (GDIET_SPREAD=0: one per thread)
	double stage_s[6] = {0, 0, 0, 0, 0, 0};
	uint64_t last_cells = 0, last_alg_bytes = 0; // of the most recent DP launch
	// reads the most recent map call gave up on (a DP box outside its read / contig: undefined behaviour in the reference); they come back
	// with n_regs = 0 while the rest of the batch is mapped.  failed_total: since the context was created.
	int64_t failed_last = 0, failed_total = 0;
	std::string warn;                  // what the last such read was (gdiet_hip_map_failed_reads)
};

#define GD_HIP(call)                                                                              \
	do {                                                                                          \
		hipError_t e__ = (call);                                                                  \
		if (e__ != hipSuccess) {                                                                  \
			ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);                        \
			return GDIET_E_HIP;                                                                   \
		}                                                                                         \
	} while (0)

// The long waits of the mapping path (a lane waits ~100 ms for a HiFi DP stage).  Default: hipStreamSynchronize -- measured on the
// GPU box it costs no CPU time worth mentioning (56.2 s of process CPU either way over a 20 s bench run: the runtime sleeps on the
// completion signal after a short spin) and wakes up faster.  GDIET_SYNC=block waits on a hipEventBlockingSync event instead
// (-2 % HiFi, -11 % ShortReads throughput): for hosts where the runtime's wait does spin.
static hipError_t gd_stream_wait(gdiet_ctx *ctx, hipStream_t s)
{
	if (!ctx->blocking_wait || !ctx->wait_ev) return hipStreamSynchronize(s);
	hipError_t e = hipEventRecord(ctx->wait_ev, s);
	if (e != hipSuccess) return e;
	return hipEventSynchronize(ctx->wait_ev);
}

static int gd_grow(gdiet_ctx *ctx, DevBuf &b, size_t bytes)
{
	if (bytes <= b.cap) return GDIET_OK;
	// growth is rare (first batches); it synchronises the stream because older work may still read the buffer
	GD_HIP(hipStreamSynchronize(ctx->stream));
	if (b.p) GD_HIP(hipFree(b.p));
	b.p = nullptr, b.cap = 0;
	size_t want = bytes + (bytes >> 3) + 4096;
	hipError_t e = hipMalloc(&b.p, want);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		want = bytes;
		e = hipMalloc(&b.p, want);
	}
	if (e != hipSuccess) {
		(void)hipGetLastError();
		ctx->err = "hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e);
		b.p = nullptr;
		return GDIET_E_NOMEM;
	}
	b.cap = want;
	if (getenv("GDIET_TRACE_ALLOC")) fprintf(stderr, "[gdiet] device buffer grown to %zu bytes\n", want);
	return GDIET_OK;
}

// CPUs this process may really use: the smallest of the hardware count, the affinity mask and the cgroup CPU quota (a container
// with a 16-CPU quota on a 256-thread host must not run 256 workers: they are throttled together every scheduling period)
static int gd_effective_cpus()
{
	int n = (int)std::max(1u, std::thread::hardware_concurrency());
	cpu_set_t set;
	CPU_ZERO(&set);
	if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = std::min(n, CPU_COUNT(&set));
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
		char q[64];
		long long period = 0;
		if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") && period > 0) n = std::min<long long>(n, std::max<long long>(1, (atoll(q) + period - 1) / period));
		fclose(f);
	} else {
		long long quota = -1, period = 0;
		if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
		if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
		if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
	}
	return n;
}

extern "C" int gdiet_hip_effective_cpus(void) { return gd_effective_cpus(); }

extern "C" int gdiet_hip_init(gdiet_ctx **out, int device)
{
	if (!out) return GDIET_E_PARAM;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return GDIET_E_NODEVICE;
	gdiet_ctx *ctx = new gdiet_ctx();
	ctx->device = device;
	hipDeviceProp_t prop;
	if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
		delete ctx;
		return GDIET_E_NODEVICE;
	}
	snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { // the code object is gfx950-only; fail loudly, no fallback
		delete ctx;
		return GDIET_E_NODEVICE;
	}
	if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
		delete ctx;
		return GDIET_E_HIP;
	}
	for (int i = 0; i < 4; ++i)
		if (hipEventCreate(&ctx->ev[i]) != hipSuccess) {
			delete ctx;
			return GDIET_E_HIP;
		}
	{
		int lo = 0, hi = 0; // numerically lower = higher priority
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
		const char *pe = getenv("GDIET_DP_PRIORITY");
		const bool prio = pe && atoi(pe) != 0; // measured: no gain from a raised priority, the separate stream is what matters
		if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio ? hi : lo) != hipSuccess) { delete ctx; return GDIET_E_HIP; }
		if (hipStreamCreateWithPriority(&ctx->stream_dp, hipStreamNonBlocking, prio ? hi : lo) != hipSuccess) { delete ctx; return GDIET_E_HIP; }
	}
	for (int i = 0; i < 3; ++i)
		if (hipEventCreate(&ctx->ev2[i]) != hipSuccess) { delete ctx; return GDIET_E_HIP; }
	if (hipEventCreateWithFlags(&ctx->wait_ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { delete ctx; return GDIET_E_HIP; }
	{ const char *sy = getenv("GDIET_SYNC"); if (sy) ctx->blocking_wait = strcmp(sy, "block") == 0; }
	if (hipEventCreateWithFlags(&ctx->arena_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->gather_ev, hipEventDisableTiming) != hipSuccess) { delete ctx; return GDIET_E_HIP; }
	ctx->wave_slots = prop.multiProcessorCount * 4 * 5; // CUs x SIMDs x resident wavefronts of the 64-lane DP kernel (94 VGPRs)
	ctx->host_threads = std::min(64, gd_effective_cpus());
	{
		const char *e = getenv("GDIET_SEED_KERNEL");
		ctx->seed_thread_kernel = e && !strcmp(e, "thread") ? 1 : e && !strcmp(e, "wave") ? 2 : 0; // 0: by read length
		const char *sl = getenv("GDIET_SLICES_PER_LANE");
		if (sl && atoi(sl) > 0) ctx->slices_per_lane = atoi(sl);
		const char *bw = getenv("GDIET_BT_WAVE");
		if (bw) ctx->bt_wave = atoi(bw) != 0;
		const char *tw = getenv("GDIET_WIDE_TWO_WAVES");
		if (tw) ctx->wide_two_waves = atoi(tw) != 0;
		const char *wc = getenv("GDIET_WIDE_CKPT");
		if (wc) ctx->wide_ckpt = atoi(wc) != 0;
		const char *vw = getenv("GDIET_VOTE_WAVE");
		if (vw) ctx->vote_wave = atoi(vw) != 0;
		const char *ib = getenv("GDIET_INDEX_BUILD");
		if (ib) ctx->index_on_device = strcmp(ib, "host") != 0;
		const char *po = getenv("GDIET_POST");
		if (po) ctx->post_on_device = strcmp(po, "host") != 0;
		const char *dw = getenv("GDIET_DP_WAVES");
		if (dw && atoi(dw) == 4) ctx->dp_waves = 4;
		const char *sb = getenv("GDIET_SR_BOXES");
		if (sb) ctx->sr_boxes_on_device = strcmp(sb, "host") != 0;
		const char *fb = getenv("GDIET_FUSE_BT");
		if (fb) ctx->fuse_bt = atoi(fb) != 0;
		const char *ds = getenv("GDIET_DP_SPLIT");
		if (ds) ctx->dp_split = atoi(ds) != 0;
		const char *sp = getenv("GDIET_SPREAD");
		if (sp) ctx->spread = atoi(sp) != 0;
	}
	*out = ctx;
	return GDIET_OK;
}

static void gd_pool_free(void *pool); // map_pipeline.hip.h
template <class F> static void gd_parallel_for(gdiet_ctx *ctx, int n_threads, int n, F f); // map_pipeline.hip.h
static void gd_join_open_tickets(gdiet_ctx *ctx);

extern "C" void gdiet_hip_destroy(gdiet_ctx *ctx)
{
	if (!ctx) return;
	gd_join_open_tickets(ctx); // batches still in flight run to their end first: their lanes use this context's pool, arena and streams
	if (ctx->pool) gd_pool_free(ctx->pool), ctx->pool = nullptr;
	for (gdiet_ctx *c : ctx->children) gdiet_hip_destroy(c);
	ctx->children.clear();
	for (int i = 0; i < GD_MAX_INFLIGHT; ++i)
		if (ctx->async_lane[i]) gdiet_hip_destroy(ctx->async_lane[i]), ctx->async_lane[i] = nullptr;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	DevBuf *bufs[] = {&ctx->arena, &ctx->tasks, &ctx->ids, &ctx->status, &ctx->qseq, &ctx->tseq, &ctx->score, &ctx->ncig, &ctx->cigar,
	                  &ctx->m_sc, &ctx->m_mv, &ctx->m_u64, &ctx->m_seed, &ctx->m_seedout, &ctx->m_voteout, &ctx->m_hitoff, &ctx->m_hits,
	                  &ctx->m_boxes, &ctx->m_q, &ctx->m_t, &ctx->m_aux, &ctx->m_cig, &ctx->m_pack, &ctx->m_post, &ctx->m_seedids, &ctx->pipes, &ctx->pipe_runs, &ctx->pipe_dst, &ctx->diag};
	for (DevBuf *b : bufs)
		if (b->p) (void)hipFree(b->p);
	if (ctx->h_pin.p) (void)hipHostFree(ctx->h_pin.p);
	DevBuf *hosts[] = {&ctx->h_boxes, &ctx->h_cand, &ctx->h_tasks, &ctx->h_seedout, &ctx->h_res, &ctx->h_cig, &ctx->h_post};
	for (DevBuf *b : hosts) free(b->p);
	for (int i = 0; i < 4; ++i)
		if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
	for (int i = 0; i < 3; ++i)
		if (ctx->ev2[i]) (void)hipEventDestroy(ctx->ev2[i]);
	if (ctx->arena_ev) (void)hipEventDestroy(ctx->arena_ev);
	if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
	if (ctx->gather_ev) (void)hipEventDestroy(ctx->gather_ev);
	if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
	if (ctx->stream_dp) (void)hipStreamDestroy(ctx->stream_dp);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

#ifdef GD_CLOCK_STAMP
// measurement build only (tools/clock_probe.py; not declared in include/gdiet_hip.h): the raw per-wavefront stamps
extern "C" int gdiet_hip_debug_clock_stamps(unsigned long long *out, int n_slots)
{
	if (n_slots > GD_CLOCK_SLOTS) n_slots = GD_CLOCK_SLOTS;
	return hipMemcpyFromSymbol(out, HIP_SYMBOL(gd_clock_stamps), sizeof(unsigned long long) * 4 * (size_t)n_slots) == hipSuccess ? n_slots : -1;
}
#endif
// the clock stamps of the 64-lane DP kernel's wavefronts (ksw_wave.hip.h), reduced: per wavefront sclk = ticks(s_memtime) /
// ticks(s_memrealtime) x 100 MHz.  Call after the batch has completed.
extern "C" int gdiet_hip_last_dp_clock(gdiet_ctx *ctx, double *sclk_mhz_median, double *sclk_mhz_min, double *wavefront_ms_median)
{
	if (!ctx) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	std::vector<unsigned long long> st((size_t)GD_CLOCK_SLOTS * 4);
	GD_HIP(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(gd_clock_stamps), sizeof(unsigned long long) * st.size()));
	std::vector<double> f, ms;
	for (int i = 0; i < GD_CLOCK_SLOTS; ++i) {
		const unsigned long long mt0 = st[4 * i], rt0 = st[4 * i + 1], mt1 = st[4 * i + 2], rt1 = st[4 * i + 3];
		if (rt1 > rt0 && mt1 > mt0 && rt1 - rt0 > 1000) f.push_back((double)(mt1 - mt0) / (double)(rt1 - rt0) * 100.0), ms.push_back((double)(rt1 - rt0) * 1e-5); // (>= 10 us)
	}
	if (f.empty()) { ctx->err = "no 64-lane DP kernel has run yet"; return GDIET_E_PARAM; }
	std::sort(f.begin(), f.end()), std::sort(ms.begin(), ms.end());
	if (sclk_mhz_median) *sclk_mhz_median = f[f.size() / 2];
	if (sclk_mhz_min) *sclk_mhz_min = f.front();
	if (wavefront_ms_median) *wavefront_ms_median = ms[ms.size() / 2];
	return GDIET_OK;
}


extern "C" const char *gdiet_hip_strerror(const gdiet_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

extern "C" int gdiet_hip_device_name(const gdiet_ctx *ctx, char *buf, size_t len)
{
	if (!ctx || !buf || !len) return GDIET_E_PARAM;
	snprintf(buf, len, "%s", ctx->name);
	return GDIET_OK;
}

extern "C" int gdiet_hip_last_kernel_mask(const gdiet_ctx *ctx) { return ctx ? ctx->last_mask : 0; }

extern "C" int gdiet_hip_set_kernel_mode(gdiet_ctx *ctx, int mode)
{
	if (!ctx || mode < 0 || mode > 2) return GDIET_E_PARAM;
	ctx->kernel_mode = mode;
	return GDIET_OK;
}

extern "C" int gdiet_hip_set_dp_waves(gdiet_ctx *ctx, int waves_per_simd)
{
	if (!ctx || (waves_per_simd != 4 && waves_per_simd != 5)) return GDIET_E_PARAM;
	ctx->dp_waves = waves_per_simd;
	return GDIET_OK;
}

extern "C" int gdiet_hip_set_dp_split(gdiet_ctx *ctx, int on)
{
	if (!ctx) return GDIET_E_PARAM;
	ctx->dp_split = on != 0;
	for (int i = 0; i < GD_MAX_INFLIGHT; ++i) if (ctx->async_lane[i]) ctx->async_lane[i]->dp_split = ctx->dp_split;
	return GDIET_OK;
}

extern "C" int gdiet_hip_reserve(gdiet_ctx *ctx, size_t bytes)
{
	if (!ctx) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	return gd_grow(ctx, ctx->arena, bytes);
}

// ---- planning ----------------------------------------------------------------------------------------------

// number of cells the generic kernel's LDS ring must hold for this geometry (see ksw_generic.hip.h)
static int gd_generic_cap(int qlen, int tlen, int w)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int n = std::min(std::min(qlen, tlen), w + 1);
	int need = n + 64, cap = 256;
	while (cap < need) cap <<= 1;
	return cap;
}

static inline size_t gd_align256(size_t x) { return (x + 255) & ~(size_t)255; }

static const int gd_group_lanes = getenv("GDIET_GROUP_LANES") ? atoi(getenv("GDIET_GROUP_LANES")) : 0; // 16: always four alignments per wavefront
// GDIET_SR_PIPE=0: no skewed pipelines (ksw_pipe.hip.h), short alignments on the grouped kernels only
static const bool gd_use_pipe = !(getenv("GDIET_SR_PIPE") && atoi(getenv("GDIET_SR_PIPE")) == 0);

// decide kernel + backtrace geometry of one alignment
static void gd_plan_one(int mode, bool wave_scoring_ok, int qlen, int tlen, int w, int32_t &kind, int32_t &row_bytes)
{
	const int ncol = gd_ncol16(qlen, tlen, w);
	kind = GD_KIND_GENERIC, row_bytes = ncol * 16;
	if (mode == 1 || !wave_scoring_ok) return;
	if (gd_wave_supported(qlen, tlen, w, 64)) {
		if (gd_wave_supported(qlen, tlen, w, 16)) {
			// short alignments: several per wavefront.  Targets of <= 128 / 160 bases keep every block in a lane of its own: groups of
			// 8 / 10 lanes (8 / 6 alignments per wavefront) instead of one DPP row of 16 each
			// (the reference's n_col_ counts one block more -- the spill of the score row above the window -- but beyond the target's
			// last block that spill is never read)
			const int g = gd_group_lanes == 16 ? 16 : tlen <= 128 ? 8 : tlen <= 160 ? 10 : 16;
			kind = GD_KIND_WAVE16, row_bytes = g * 16;
		}
		else if (gd_use_pipe && gd_pipe_geometry_ok(qlen, tlen, w)) kind = GD_KIND_WAVE16, row_bytes = 16 * 16; // 241..256 bases, full matrix: one block more than the 16-lane groups hold -- the pipelines take it (every run, however short)
		else kind = GD_KIND_WAVE64, row_bytes = 64 * 16;
	} else if (gd_wave_supported(qlen, tlen, w, 128)) kind = GD_KIND_WAVE128; // row_bytes stays n_col_*16
}

extern "C" size_t gdiet_hip_ksw_workspace_bytes(int n, const int64_t *qoff, const int64_t *toff, const int32_t *w)
{
	size_t tot = 0;
	for (int i = 0; i < n; ++i) {
		const int qlen = (int)(qoff[i + 1] - qoff[i]), tlen = (int)(toff[i + 1] - toff[i]);
		if (qlen <= 0 || tlen <= 0) continue;
		int32_t kind, rb;
		gd_plan_one(0, true, qlen, tlen, w[i], kind, rb);
		const size_t a = (size_t)(qlen + tlen - 1) * (size_t)rb;
		const size_t b = (size_t)(qlen + tlen - 1) * (size_t)gd_ncol16(qlen, tlen, w[i]) * 16; // forced-generic worst case
		tot += gd_align256(std::max(a, b) + 64);
	}
	return tot;
}

static int gd_consts(gdiet_ctx *ctx, const gdiet_ksw_score_t *sc, KswConst &K)
{
	if (!sc) { ctx->err = "scoring is NULL"; return GDIET_E_PARAM; }
	if (sc->flag != GDIET_EZ_APPROX_MAX) {
		ctx->err = "only flag == GDIET_EZ_APPROX_MAX (the live path's mode) is implemented";
		return GDIET_E_PARAM;
	}
	int q = sc->q, e = sc->e, q2 = sc->q2, e2 = sc->e2;
	if (q2 + e2 < q + e) std::swap(q, q2), std::swap(e, e2); // SR/ksw2_extd2_sse.c:78
	K.q = q, K.e = e, K.q2 = q2, K.e2 = e2;
	K.sc_mch = sc->match, K.sc_mis = sc->mismatch;
	K.sc_N = sc->sc_ambi == 0 ? -e2 : sc->sc_ambi;
	// :96-100 early return "if (-min_sc > 2 * (q + e)) return" leaves score = NEG_INF for every pair; the live
	// path can never get there (mm_check_opt), so it is reported as a parameter error instead.
	int min_sc = std::min<int>(std::min<int>(sc->mismatch, sc->match), std::min<int>(sc->sc_ambi, 0));
	if (-min_sc > 2 * (q + e)) { ctx->err = "-min_sc > 2*(q+e): the reference returns without aligning"; return GDIET_E_PARAM; }
	if ((q + e) + (q2 + e2) > 127) { ctx->err = "(q+e)+(q2+e2) > 127 (mm_check_opt, options.c:218)"; return GDIET_E_PARAM; }
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	K.long_thres = long_thres;
	K.long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	return GDIET_OK;
}

// host buffer that only grows and is kept between batches (contents are not preserved).  Plain pageable memory: pinning these
// was measured -- neutral for short reads, and 2-3 % slower for HiFi batches (the asynchronous device-to-host copy of the DP
// results then waits behind the other batches' kernels instead of being staged at once).
static int gd_host_grow(gdiet_ctx *ctx, DevBuf &b, size_t bytes)
{
	if (bytes <= b.cap) return GDIET_OK;
	free(b.p);
	b.p = nullptr, b.cap = 0;
	const size_t want = bytes + (bytes >> 2) + 4096;
	if (posix_memalign(&b.p, 256, want)) b.p = nullptr; // (256-byte aligned: the runtime's copy kernels pick their form by the alignment of both ends)
	if (!b.p) { ctx->err = "out of host memory (" + std::to_string(want) + " bytes)"; return GDIET_E_NOMEM; }
	b.cap = want;
	return GDIET_OK;
}

// ---- device-pointer entry point ----------------------------------------------------------------------------

// The work of gdiet_hip_ksw_extd2_batch_dev.  h_cigar_off / h_exact_score: host copies of the two small device arrays the
// planner needs; a caller that has them (the mapping pipeline) passes them and the call then never waits for the stream --
// which is what lets it be enqueued behind another batch's DP stage.
static int gd_ksw_batch_dev(gdiet_ctx *ctx, int n, const uint8_t *d_qseq, const uint8_t *d_tseq, const int32_t *d_exact_score,
                            const gdiet_ksw_score_t *sc, int32_t *d_score, int32_t *d_n_cigar, uint32_t *d_cigar, const int64_t *d_cigar_off,
                            const int64_t *h_qoff, const int64_t *h_toff, const int32_t *h_w, void *stream_, const int64_t *h_cigar_off,
                            const int32_t *h_exact_score, hipEvent_t arena_free = nullptr /* the kernels (not the descriptor copies) wait for it */,
                            std::unique_lock<std::mutex> *arena_turn = nullptr /* locked here once the planning is done, left locked */)
{
	if (!ctx) return GDIET_E_PARAM;
	if (n <= 0) return GDIET_OK;
	if (!d_qseq || !d_tseq || !d_score || !d_n_cigar || !d_cigar || !h_qoff || !h_toff || !h_w) {
		ctx->err = "NULL argument";
		return GDIET_E_PARAM;
	}
	(void)hipSetDevice(ctx->device);
	hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
	KswConst K;
	int rc = gd_consts(ctx, sc, K);
	if (rc) return rc;
	// d_cigar_off / d_exact_score are small: planning needs them on the host
	std::vector<int64_t> h_cig_own;
	std::vector<int32_t> h_ex_own;
	if (!h_cigar_off || (d_exact_score && !h_exact_score)) {
		h_cig_own.resize(n + 1);
		GD_HIP(hipMemcpyAsync(h_cig_own.data(), d_cigar_off, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, stream));
		if (d_exact_score) {
			h_ex_own.resize(n);
			GD_HIP(hipMemcpyAsync(h_ex_own.data(), d_exact_score, sizeof(int32_t) * n, hipMemcpyDeviceToHost, stream));
		}
		GD_HIP(hipStreamSynchronize(stream));
		h_cigar_off = h_cig_own.data(), h_exact_score = d_exact_score ? h_ex_own.data() : nullptr;
	}
	const int64_t *h_cig = h_cigar_off;
	const int32_t *h_ex = h_exact_score;

	// GDIET_TRACE_STAGES: where the planner's time goes (one line per call on stderr)
	static const bool plan_trace = getenv("GDIET_TRACE_STAGES") != nullptr;
	std::string plan_s;
	double plan_t = plan_trace ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
	auto plan_mark = [&](const char *what) {
		if (!plan_trace) return;
		const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
		char b[64];
		snprintf(b, sizeof b, " %s %.2f", what, 1e3 * (t - plan_t));
		plan_s += b, plan_t = t;
	};
	if ((rc = gd_host_grow(ctx, ctx->h_tasks, sizeof(KswTask) * (size_t)n))) return rc;
	KswTask *h_tasks = (KswTask *)ctx->h_tasks.p;
	const bool wave_scoring_ok = gd_wave_scoring_ok(K);
	size_t bt = 0;
	uint64_t cells_sum = 0, alg_sum = 0;
	std::vector<int32_t> ids[4];
	int max_cap = 0;
	ctx->last_mask = 0;
	// kernel + backtrace geometry of every alignment first, on the host threads: the admission test of the wave kernels walks the blocks
	// of the band (~1 000 steps for a 15 kbp alignment: 4-5 ms for the 9 400 alignments of a HiFi batch on one thread -- time that sat
	// between the gather kernel and the DP kernel whenever a batch was not ready early).  Slices with a memo each: a short-read
	// batch repeats a few geometries.
	// (the same pass fills every other field of the descriptor, checks it, adds up the roofline accounting and lists the alignments by
	// kind -- per slice, joined in slice order afterwards: done by one thread this was 3-5 ms per 262 144 short alignments, most of a
	// short-read batch's planning)
	struct PlanSlice { uint64_t cells = 0, alg = 0; int max_cap = 0, err = 0; uint32_t mask = 0; std::vector<int32_t> ids[4]; };
	const int n_sl = std::max(1, std::min(64, n / 256));
	std::vector<PlanSlice> slices((size_t)n_sl);
	{
		gd_parallel_for(ctx, ctx->lane_threads, n_sl, [&](int sl) {
			struct { int qlen = -1, tlen = -1, w = 0; int32_t kind = 0, row_bytes = 0; } memo;
			PlanSlice &S = slices[sl];
			const int i0 = (int)((int64_t)n * sl / n_sl), i1 = (int)((int64_t)n * (sl + 1) / n_sl);
			for (int k = 0; k < 4; ++k) S.ids[k].reserve((size_t)(i1 - i0));
			for (int i = i0; i < i1; ++i) {
				KswTask &T = h_tasks[i];
				T.qlen = (int)(h_qoff[i + 1] - h_qoff[i]), T.tlen = (int)(h_toff[i + 1] - h_toff[i]), T.w = h_w[i];
				T.kind = GD_KIND_GENERIC, T.row_bytes = 0;
				T.qoff = h_qoff[i], T.toff = h_toff[i];
				T.cig_off = h_cig[i], T.cig_cap = (int32_t)std::min<int64_t>(h_cig[i + 1] - h_cig[i], 0x7fffffff);
				T.exact_score = d_exact_score ? h_ex[i] : GD_NEG_INF;
				T.pad = 0, T.bt_off = 0;
				if (T.qlen <= 0 || T.tlen <= 0) { S.err |= 1; continue; } // (refused below)
				if (T.qlen == memo.qlen && T.tlen == memo.tlen && T.w == memo.w) T.kind = memo.kind, T.row_bytes = memo.row_bytes;
				else {
					gd_plan_one(ctx->kernel_mode, wave_scoring_ok, T.qlen, T.tlen, T.w, T.kind, T.row_bytes);
					memo.qlen = T.qlen, memo.tlen = T.tlen, memo.w = T.w, memo.kind = T.kind, memo.row_bytes = T.row_bytes;
				}
				if (ctx->kernel_mode == 2 && T.kind == GD_KIND_GENERIC) S.err |= 2;
				if (T.kind == GD_KIND_GENERIC) {
					const int cap = gd_generic_cap(T.qlen, T.tlen, T.w);
					if (cap * 7 > 160 * 1024 - 1024) S.err |= 4;
					S.max_cap = std::max(S.max_cap, cap);
				}
				{ // accounting for the roofline: SURVEY.md 8d's per-alignment figure
					const uint64_t wb = (uint64_t)(T.w < 0 ? std::max(T.qlen, T.tlen) : T.w) + 1;
					const uint64_t band = std::min<uint64_t>(wb, (uint64_t)std::min(T.qlen, T.tlen));
					const uint64_t cells = (uint64_t)(T.qlen + T.tlen - 1) * band;
					S.cells += cells;
					S.alg += cells + (uint64_t)(T.qlen + T.tlen) + (uint64_t)T.qlen + (uint64_t)(T.tlen + 1) / 2;
				}
				S.ids[T.kind].push_back(i);
				S.mask |= T.kind == GD_KIND_GENERIC ? 2 : T.kind == GD_KIND_WAVE16 ? 4 : T.kind == GD_KIND_WAVE128 ? 8 : 1;
			}
		});
	}
	plan_mark("kinds");
	{
		int err = 0;
		size_t cnt[4] = {0, 0, 0, 0};
		for (const PlanSlice &S : slices) {
			err |= S.err, cells_sum += S.cells, alg_sum += S.alg, max_cap = std::max(max_cap, S.max_cap), ctx->last_mask |= (int)S.mask;
			for (int k = 0; k < 4; ++k) cnt[k] += S.ids[k].size();
		}
		// (the first failure in the order the sequential form reported them)
		if (err & 1) { ctx->err = "empty sequence in batch (the reference returns without aligning)"; return GDIET_E_PARAM; }
		if (err & 2) { ctx->err = "alignment does not fit the wave kernel"; return GDIET_E_PARAM; }
		if (err & 4) { ctx->err = "band wider than the LDS window of the generic kernel"; return GDIET_E_PARAM; }
		for (int k = 0; k < 4; ++k) {
			ids[k].resize(cnt[k]);
			size_t at = 0;
			for (const PlanSlice &S : slices) {
				if (!S.ids[k].empty()) memcpy(ids[k].data() + at, S.ids[k].data(), S.ids[k].size() * sizeof(int32_t));
				at += S.ids[k].size();
			}
		}
	}
	plan_mark("fields");
	bool wide_ck = false;
	if (!ids[GD_KIND_WAVE128].empty() && ctx->fuse_bt && !(ctx->single_affine && K.q == K.q2 && K.e == K.e2)) {
		size_t full = 0;
		for (int32_t id : ids[GD_KIND_WAVE128]) full += (size_t)(h_tasks[id].qlen + h_tasks[id].tlen - 1) * (size_t)h_tasks[id].row_bytes;
		wide_ck = ctx->wide_ckpt == 1 || (ctx->wide_ckpt < 0 && ((int)ids[GD_KIND_WAVE128].size() >= ctx->wave_slots / 5 || full > ((size_t)100 << 30)));
	}
	for (int i = 0; i < n; ++i) {
		KswTask &T = h_tasks[i];
		T.bt_off = (int64_t)bt;
		if (wide_ck && T.kind == GD_KIND_WAVE128) bt += gd_align256(gd_ck_bytes(T.qlen, T.tlen, T.row_bytes) + 64);
		else bt += gd_align256((size_t)(T.qlen + T.tlen - 1) * (size_t)T.row_bytes + 64);
	}
	plan_mark("bt_off");
	// longest alignments first inside each class: the tail of the grid is then made of short jobs (a class whose members all have
	// one geometry -- a short-read batch -- is in order already)
	for (int k = 0; k < 4; ++k) {
		bool uniform = true;
		for (size_t j = 1; j < ids[k].size() && uniform; ++j) {
			const KswTask &A = h_tasks[ids[k][0]], &B = h_tasks[ids[k][j]];
			uniform = A.qlen == B.qlen && A.tlen == B.tlen && A.w == B.w;
		}
		if (uniform) continue;
		// (a stable sort by geometry, done by grouping: a short-read batch has 400 k alignments but a few dozen geometries)
		struct Geo { int qlen, tlen, w; std::vector<int32_t> members; };
		std::vector<Geo> geos;
		std::unordered_map<uint64_t, std::vector<int>> slot_of; // hash of the geometry -> geos[] entries with that hash
		int g_prev = -1;
		for (int32_t id : ids[k]) {
			const KswTask &A = h_tasks[id];
			if (g_prev >= 0 && geos[g_prev].qlen == A.qlen && geos[g_prev].tlen == A.tlen && geos[g_prev].w == A.w) { geos[g_prev].members.push_back(id); continue; }
			const uint64_t h = ((uint64_t)(uint32_t)A.qlen * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)A.tlen * 0xC2B2AE3D27D4EB4Full) ^ (uint64_t)(uint32_t)A.w;
			std::vector<int> &cand = slot_of[h];
			int g = -1;
			for (int c : cand) if (geos[c].qlen == A.qlen && geos[c].tlen == A.tlen && geos[c].w == A.w) { g = c; break; }
			if (g < 0) { g = (int)geos.size(); geos.push_back(Geo{A.qlen, A.tlen, A.w, {}}); cand.push_back(g); }
			geos[g].members.push_back(id), g_prev = g;
		}
		std::vector<int> order(geos.size());
		for (size_t g = 0; g < geos.size(); ++g) order[g] = (int)g;
		std::sort(order.begin(), order.end(), [&](int a, int b) {
			const Geo &A = geos[a], &B = geos[b];
			if ((int64_t)A.qlen + A.tlen != (int64_t)B.qlen + B.tlen) return (int64_t)A.qlen + A.tlen > (int64_t)B.qlen + B.tlen;
			if (A.qlen != B.qlen) return A.qlen > B.qlen; // equal geometries are neighbours (16-lane quartets below)
			return A.w > B.w;
		});
		size_t at = 0;
		for (int g : order) for (int32_t id : geos[g].members) ids[k][at++] = id;
	}
	plan_mark("order");
	// checkpointed wide-band alignments whose band fits a 96-block ring (w = 1300: 83 blocks) go to the form with one block + one half
	// block per lane; the rest (wider bands) keep two blocks per lane.  Both lists stay longest-first.  GDIET_WIDE_RING=128 forces the latter.
	size_t n_ring96 = 0;
	if (wide_ck) {
		static const bool ring128_only = getenv("GDIET_WIDE_RING") && atoi(getenv("GDIET_WIDE_RING")) == 128;
		std::vector<int32_t> &v = ids[GD_KIND_WAVE128];
		if (!ring128_only)
			n_ring96 = (size_t)(std::stable_partition(v.begin(), v.end(), [&](int32_t id) {
				const KswTask &A = h_tasks[id];
				return gd_wave_supported(A.qlen, A.tlen, A.w, 96);
			}) - v.begin());
	}
	// the short-alignment kernels run 4 / 6 / 8 alignments of identical (qlen, tlen, w) per wavefront (groups of 16 / 10 / 8 lanes):
	// cut the sorted list into such groups, one list per group width (-1 pads an incomplete group)
	std::vector<int32_t> groups[3]; // [0]: 16 lanes, [1]: 10, [2]: 8
	// Full matrices (a short-read batch: w >= both lengths) of one geometry, enough of them to keep every group of a wavefront busy for a
	// few alignments, run as skewed pipelines instead (ksw_pipe.hip.h): a wavefront takes np alignments per group, sized so that the
	// run fills the GPU's wavefront slots once.  GDIET_SR_PIPE=0 keeps the grouped kernels; GDIET_PIPE_NP forces np.
	const bool use_pipe = gd_use_pipe;
	static const int pipe_np_forced = getenv("GDIET_PIPE_NP") ? atoi(getenv("GDIET_PIPE_NP")) : 0;
	static const int pipe_np_min = getenv("GDIET_PIPE_NP_MIN") ? std::max(1, atoi(getenv("GDIET_PIPE_NP_MIN"))) : 8; // see pipe_compact_kernel
	std::vector<int32_t> pipe_ids;
	ctx->h_pipes.clear(), ctx->h_pipe_runs.clear();
	{
		const std::vector<int32_t> &v = ids[GD_KIND_WAVE16];
		size_t i = 0;
		while (i < v.size()) {
			const KswTask &A = h_tasks[v[i]];
			if (use_pipe && gd_pipe_geometry_ok(A.qlen, A.tlen, A.w)) {
				size_t j = i + 1;
				while (j < v.size() && h_tasks[v[j]].qlen == A.qlen && h_tasks[v[j]].tlen == A.tlen && h_tasks[v[j]].row_bytes == A.row_bytes &&
				       gd_pipe_geometry_ok(A.qlen, A.tlen, h_tasks[v[j]].w)) ++j;
				const PipeGeo geo = gd_pipe_geo(A.qlen, A.tlen);
				const size_t m = j - i;
				if (m >= (size_t)(2 * geo.NG) || !gd_wave_supported(A.qlen, A.tlen, A.w, 16)) { // (the second: nothing else takes it, see gd_plan_one)
					// Alignments per group of a wavefront.  The kernel has 4 wavefront slots per SIMD.  A batch on its own (synchronous call): one
					// round of wavefronts over 70 % of the slots (100 000 pairs: np 6 -> 2.18 ms, 8 -> 2.48; 12 000 pairs: np 1 -> 0.30 ms,
					// 8 -> 0.84 -- a short run wants many short pipes, filling and draining is cheaper than an empty GPU).  A lane of a context
					// with batches in flight: half of the slots and at least 8 per group -- two batches' kernels share the GPU, and the longer
					// pipes lose less to filling and draining (26.4 -> 28.7 M reads/s with eight batches in flight) -- but never fewer than 256 wavefronts.
					const size_t all_slots = (size_t)(ctx->wave_slots / 5 * 4), groups = (m + geo.NG - 1) / geo.NG;
					size_t np;
					if (pipe_np_forced > 0) np = (size_t)pipe_np_forced;
					else if (!ctx->parent) np = std::max<size_t>(1, (groups + all_slots * 7 / 10 - 1) / (all_slots * 7 / 10));
					else np = std::min(std::max<size_t>(8, (groups + all_slots / 2 - 1) / (all_slots / 2)), std::max<size_t>(1, groups / 256));
					np = std::min(np, groups);
					const size_t n_waves = (m + geo.NG * np - 1) / (geo.NG * np);
					PipeRun R;
					memset(&R, 0, sizeof(R));
					R.src_off = (int32_t)pipe_ids.size(), R.dst_off = R.src_off, R.m = (int32_t)m, R.wave_off = (int32_t)ctx->h_pipes.size(), R.n_waves = (int32_t)n_waves, R.ng = geo.NG, R.np_min = (int32_t)std::max<size_t>(1, std::min<size_t>(np, (size_t)pipe_np_min));
					for (size_t k = i; k < j; ++k) pipe_ids.push_back(v[k]);
					PipeWave W; // (id_off, cnt, np: pipe_compact_kernel, once the pre-filter has answered)
					memset(&W, 0, sizeof(W));
					W.qlen = A.qlen, W.tlen = A.tlen, W.row_bytes = A.row_bytes;
					ctx->h_pipes.insert(ctx->h_pipes.end(), n_waves, W);
					ctx->h_pipe_runs.push_back(R);
					i = j;
					continue;
				}
			}
			const int gl = A.row_bytes >> 4, per = 64 / gl, which = gl == 16 ? 0 : gl == 10 ? 1 : 2;
			size_t j = i + 1;
			while (j < v.size() && j < i + per) {
				const KswTask &B = h_tasks[v[j]];
				if (B.qlen != A.qlen || B.tlen != A.tlen || B.w != A.w) break;
				++j;
			}
			for (size_t k = i; k < i + per; ++k) groups[which].push_back(k < j ? v[k] : -1);
			i = j;
		}
	}
	plan_mark("groups");
	ctx->h_ids.clear();
	size_t id_off[4], group_off[3] = {0, 0, 0};
	for (int k = 0; k < 4; ++k) {
		id_off[k] = ctx->h_ids.size();
		if (k == GD_KIND_WAVE16) {
			for (int g = 0; g < 3; ++g) group_off[g] = ctx->h_ids.size(), ctx->h_ids.insert(ctx->h_ids.end(), groups[g].begin(), groups[g].end());
			for (PipeRun &R : ctx->h_pipe_runs) R.src_off += (int32_t)ctx->h_ids.size(); // (relative to the batch's whole id list from here on)
			ctx->h_ids.insert(ctx->h_ids.end(), pipe_ids.begin(), pipe_ids.end());
			if (!ctx->h_pipes.empty()) ctx->last_mask |= 16;
		} else ctx->h_ids.insert(ctx->h_ids.end(), ids[k].begin(), ids[k].end());
	}
	ctx->last_cells = cells_sum, ctx->last_alg_bytes = alg_sum;
	// An async lane works in its parent's arena, taking turns behind parent->arena_ev -- unless the batch's backtrace is small
	// enough for every lane in flight to hold one of its own (long-read batches of few reads: their DP kernels then overlap, which
	// fills the GPU while one batch's longest alignments are still running).
	// (sticky: a lane that had to fall back to the shared arena stays there until its batches are clearly below the cap again --
	// giving a 50 GB arena back and allocating it anew every other batch costs more than taking turns)
	if (ctx->parent && bt > ctx->parent->lane_arena_cap) ctx->shared_sticky = true;
	else if (ctx->parent && bt <= ctx->parent->lane_arena_cap / 4 * 3) ctx->shared_sticky = false;
	bool own = ctx->parent && bt <= ctx->parent->lane_arena_cap && !ctx->shared_sticky;
	if (own && bt > ctx->arena.cap) { // grown in big steps: freeing device memory stalls every lane
		const size_t want = std::min(ctx->parent->lane_arena_cap, std::max<size_t>(bt + (bt >> 2), (size_t)1 << 30));
		if (gd_grow(ctx, ctx->arena, std::max(bt, want - (want >> 3) - 4096)) && gd_grow(ctx, ctx->arena, bt)) own = false, ctx->err.clear(); // no room: take turns in the shared one
	}
	if (ctx->parent && !own && ctx->arena.p) { (void)hipFree(ctx->arena.p); ctx->arena.p = nullptr, ctx->arena.cap = 0; }
	ctx->own_arena = own;
	if (own) arena_free = nullptr, arena_turn = nullptr;
	DevBuf &arena = ctx->parent && !own ? ctx->parent->arena : ctx->arena;
	if (arena_turn) arena_turn->lock(); // everything above was this batch's own planning: only the use of the arena is ordered
	if (ctx->parent && !own && bt > arena.cap) GD_HIP(hipEventSynchronize(ctx->parent->arena_ev)); // growing it: the previous user must be done
	if ((rc = gd_grow(ctx, arena, bt))) {
		// a context working synchronously after batches were in flight: its idle lanes may still hold private arenas
		// (a lane that takes its turn in the shared arena reclaims the private arenas of its idle siblings the same way)
		bool freed = false;
		if (rc == GDIET_E_NOMEM) {
			gdiet_ctx *owner = ctx->parent ? ctx->parent : ctx;
			std::lock_guard<std::mutex> guard(owner->async_mu); // async_busy[] belongs to submit / wait
			for (int i = 0; i < GD_MAX_INFLIGHT; ++i) {
				gdiet_ctx *c = owner->async_lane[i];
				if (c && c != ctx && !owner->async_busy[i] && c->arena.p) { (void)hipFree(c->arena.p); c->arena.p = nullptr, c->arena.cap = 0, freed = true; }
			}
		}
		if (!freed || (rc = gd_grow(ctx, arena, bt))) return rc;
		ctx->err.clear();
	}
	plan_mark("arena");
	if ((rc = gd_grow(ctx, ctx->tasks, sizeof(KswTask) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->ids, sizeof(int32_t) * ctx->h_ids.size()))) return rc;
	if ((rc = gd_grow(ctx, ctx->status, sizeof(int32_t) * n))) return rc;
	GD_HIP(hipMemcpyAsync(ctx->tasks.p, h_tasks, sizeof(KswTask) * n, hipMemcpyHostToDevice, stream));
	GD_HIP(hipMemcpyAsync(ctx->ids.p, ctx->h_ids.data(), sizeof(int32_t) * ctx->h_ids.size(), hipMemcpyHostToDevice, stream));
	if (!ctx->h_pipes.empty()) {
		if ((rc = gd_grow(ctx, ctx->pipes, sizeof(PipeWave) * ctx->h_pipes.size()))) return rc;
		if ((rc = gd_grow(ctx, ctx->pipe_runs, sizeof(PipeRun) * ctx->h_pipe_runs.size()))) return rc;
		if ((rc = gd_grow(ctx, ctx->pipe_dst, sizeof(int32_t) * pipe_ids.size()))) return rc;
		GD_HIP(hipMemcpyAsync(ctx->pipes.p, ctx->h_pipes.data(), sizeof(PipeWave) * ctx->h_pipes.size(), hipMemcpyHostToDevice, stream));
		GD_HIP(hipMemcpyAsync(ctx->pipe_runs.p, ctx->h_pipe_runs.data(), sizeof(PipeRun) * ctx->h_pipe_runs.size(), hipMemcpyHostToDevice, stream));
	}

	plan_mark("upload");
	const KswTask *d_tasks = (const KswTask *)ctx->tasks.p;
	const int32_t *d_ids = (const int32_t *)ctx->ids.p;
	int32_t *d_status = (int32_t *)ctx->status.p;
	uint8_t *d_bt = (uint8_t *)arena.p;

	if (arena_free) GD_HIP(hipStreamWaitEvent(stream, arena_free, 0)); // descriptors are across; only the kernels queue behind the arena's last user
	GD_HIP(hipEventRecord(ctx->ev[0], stream));
	// (GDIET_DIAG_SHORTCUT=0: every short alignment goes through the DP and is walked back, also those the pre-filter could answer from the
	// main diagonal's score -- see ksw_exact_match_kernel)
	static const bool diag_shortcut = !(getenv("GDIET_DIAG_SHORTCUT") && atoi(getenv("GDIET_DIAG_SHORTCUT")) == 0);
	int32_t *d_diag = nullptr;
	if (diag_shortcut && !ids[GD_KIND_WAVE16].empty()) {
		if ((rc = gd_grow(ctx, ctx->diag, sizeof(int32_t) * (size_t)n))) return rc;
		d_diag = (int32_t *)ctx->diag.p;
	}
	hipLaunchKernelGGL(ksw_exact_match_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_tasks, n, d_qseq, d_tseq,
	                   d_status, d_score, d_n_cigar, d_cigar, d_diag, (int)K.sc_mch, (int)K.sc_mis, d_diag && K.sc_mis <= K.sc_mch && K.q + K.e > 0 ? (int)(K.sc_mch + 2 * (K.q + K.e)) + 1 : 0);
	// Head / tail split of a big 64-lane launch.  The grid is sorted longest-first, so the first `wave_slots` alignments start at
	// once and the rest fill in as slots free up -- it is the latter that finish last.  Launched as two kernels (head on the
	// caller's stream, tail on a second one), the head's backtrack runs while the tail is still in the DP, and only the tail's
	// (shorter, fewer) walks remain after the last DP wavefront.  Same kernels, same work, same results.
	const int n64 = (int)ids[GD_KIND_WAVE64].size();
	const bool single = ctx->single_affine && K.q == K.q2 && K.e == K.e2;
	const bool fuse = ctx->fuse_bt != 0; // the 64-lane kernel walks its own alignments back (status TRACED: the kernels below skip them)
	const bool split = !single && !fuse && ctx->dp_split && n64 > ctx->wave_slots + ctx->wave_slots / 8;
	const int n_head = split ? ctx->wave_slots : n64;
	const bool bt_wave = ctx->bt_wave && cells_sum / (uint64_t)n > 200000; // long walks: one wavefront each; short reads: one walk per thread
	auto backtrack = [&](const int32_t *list, int cnt, hipStream_t st) {
		if (cnt <= 0) return;
		if (bt_wave) hipLaunchKernelGGL(ksw_backtrack_wave_kernel, dim3((cnt + 3) / 4), dim3(256), 0, st, d_tasks, cnt, d_bt, d_status, d_score, d_n_cigar, d_cigar, list);
		else hipLaunchKernelGGL(ksw_backtrack_kernel, dim3((cnt + 63) / 64), dim3(64), 0, st, d_tasks, cnt, d_bt, d_status, d_score, d_n_cigar, d_cigar, 0, list,
		                        (const int32_t *)nullptr, (const int32_t *)d_diag);
	};
	ctx->last_split = split;
	if (split) {
		GD_HIP(hipEventRecord(ctx->ev2[0], stream)); // the tail may start once the pre-filter has answered
		GD_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev2[0], 0));
	}
	if (n64 > 0)
		gd_launch_wave64(d_tasks, d_ids + id_off[GD_KIND_WAVE64], n_head, d_qseq, d_tseq, d_bt, d_status, d_score, K, stream, split ? 1 : 0, single,
		                 fuse ? d_n_cigar : nullptr, fuse ? d_cigar : nullptr, ctx->parent ? ctx->parent->dp_waves : ctx->dp_waves);
	if (split) {
		gd_launch_wave64(d_tasks, d_ids + id_off[GD_KIND_WAVE64] + n_head, n64 - n_head, d_qseq, d_tseq, d_bt, d_status, d_score, K, ctx->stream2, 2, false,
		                 fuse ? d_n_cigar : nullptr, fuse ? d_cigar : nullptr);
		GD_HIP(hipEventRecord(ctx->ev2[1], ctx->stream2));
		backtrack(d_ids + id_off[GD_KIND_WAVE64] + n_head, n64 - n_head, ctx->stream2);
		GD_HIP(hipEventRecord(ctx->ev2[2], ctx->stream2));
	}
	if (!ids[GD_KIND_WAVE16].empty()) {
		// The short-alignment kernels CAN walk their own alignments back (every group's first lane, gd_bt_thread_walk), but it does not pay:
		// a wavefront then holds its slot for a few hundred dependent steps of six lanes -- DP kernel 7.6 -> 10.8 ms per 262 144 short reads
		// against 1.65 ms of the separate backtrack kernel it saves (17.9 -> 15.1 M reads/s whole path; K3 alone 35.5 -> 26.6 M pairs/s).
		// GDIET_FUSE_BT_GROUPS=1 switches it on (same results: the GPU suite passes either way).
		static const bool fuse_groups = getenv("GDIET_FUSE_BT_GROUPS") && atoi(getenv("GDIET_FUSE_BT_GROUPS")) != 0;
		int32_t *g_nc = fuse && fuse_groups ? d_n_cigar : nullptr;
		uint32_t *g_cg = fuse && fuse_groups ? d_cigar : nullptr;
		gd_launch_wave_groups<16>(d_tasks, d_ids + group_off[0], (int)(groups[0].size() / 4), d_qseq, d_tseq, d_bt, d_status, d_score, K, stream, single, g_nc, g_cg);
		gd_launch_wave_groups<10>(d_tasks, d_ids + group_off[1], (int)(groups[1].size() / 6), d_qseq, d_tseq, d_bt, d_status, d_score, K, stream, single, g_nc, g_cg);
		gd_launch_wave_groups<8>(d_tasks, d_ids + group_off[2], (int)(groups[2].size() / 8), d_qseq, d_tseq, d_bt, d_status, d_score, K, stream, single, g_nc, g_cg);
		int max_run = 0;
		for (const PipeRun &R : ctx->h_pipe_runs) max_run = std::max(max_run, (int)R.m);
		gd_launch_pipe(d_tasks, d_ids, (PipeRun *)ctx->pipe_runs.p, (int)ctx->h_pipe_runs.size(), max_run, (int32_t *)ctx->pipe_dst.p, (PipeWave *)ctx->pipes.p,
		               (int)ctx->h_pipes.size(), d_qseq, d_tseq, d_bt, d_status, d_score, K, stream, single);
	}
	if (!ids[GD_KIND_WAVE128].empty()) {
		// few wide-band alignments (the arena bounds how many 50 kbp ONT alignments fit): two wavefronts share one, halving the
		// serial chain; plenty of them: one wavefront each, two blocks per lane, no barrier
		const int n128 = (int)ids[GD_KIND_WAVE128].size();
		const bool two = !wide_ck && (ctx->wide_two_waves == 1 || (ctx->wide_two_waves < 0 && n128 < ctx->wave_slots / 2));
		if (wide_ck) {
			if (n_ring96) gd_launch_wave96c(d_tasks, d_ids + id_off[GD_KIND_WAVE128], (int)n_ring96, d_qseq, d_tseq, d_bt, d_status, d_score, K, stream, d_n_cigar, d_cigar);
			if ((size_t)n128 > n_ring96)
				gd_launch_wave128(d_tasks, d_ids + id_off[GD_KIND_WAVE128] + n_ring96, n128 - (int)n_ring96, d_qseq, d_tseq, d_bt, d_status, d_score, K, stream, d_n_cigar, d_cigar, true);
		} else if (two)
			gd_launch_wave2x64(d_tasks, d_ids + id_off[GD_KIND_WAVE128], n128, d_qseq, d_tseq, d_bt, d_status, d_score, K, stream,
			                   fuse ? d_n_cigar : nullptr, fuse ? d_cigar : nullptr);
		else
			gd_launch_wave128(d_tasks, d_ids + id_off[GD_KIND_WAVE128], n128, d_qseq, d_tseq, d_bt, d_status, d_score, K, stream,
			                  fuse ? d_n_cigar : nullptr, fuse ? d_cigar : nullptr);
	}
	if (!ids[GD_KIND_GENERIC].empty()) {
		const size_t lds = (size_t)max_cap * 7;
		if (lds > 64 * 1024)
			GD_HIP(hipFuncSetAttribute((const void *)ksw_extd2_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipLaunchKernelGGL(ksw_extd2_generic_kernel, dim3((unsigned)ids[GD_KIND_GENERIC].size()), dim3(64), lds, stream,
		                   d_tasks, d_ids + id_off[GD_KIND_GENERIC], d_qseq, d_tseq, d_bt, d_status, d_score, K, max_cap);
	}
	GD_HIP(hipEventRecord(ctx->ev[1], stream));
	if (split) { // everything but the tail: the lists of the four kinds sit back to back in d_ids, the tail is the end of the 64-lane list
		backtrack(d_ids, (int)id_off[GD_KIND_WAVE64] + n_head, stream);
		backtrack(d_ids + id_off[GD_KIND_WAVE64] + n64, (int)(ctx->h_ids.size() - id_off[GD_KIND_WAVE64] - n64), stream);
		GD_HIP(hipStreamWaitEvent(stream, ctx->ev2[2], 0)); // join: later work on the caller's stream sees the tail's results too
	} else backtrack(d_ids, (int)ctx->h_ids.size(), stream);
	GD_HIP(hipEventRecord(ctx->ev[2], stream));
	GD_HIP(hipGetLastError());
	plan_mark("launch");
	if (plan_trace) fprintf(stderr, "[gdiet dp planner, ms] n=%d%s\n", n, plan_s.c_str());
	return GDIET_OK;
}

extern "C" int gdiet_hip_ksw_extd2_batch_dev(gdiet_ctx *ctx, int n, const uint8_t *d_qseq, const int64_t *d_qoff,
                                             const uint8_t *d_tseq, const int64_t *d_toff, const int32_t *d_w,
                                             const int32_t *d_exact_score, const gdiet_ksw_score_t *sc,
                                             int32_t *d_score, int32_t *d_n_cigar, uint32_t *d_cigar,
                                             const int64_t *d_cigar_off, const int64_t *h_qoff, const int64_t *h_toff,
                                             const int32_t *h_w, void *stream_)
{
	(void)d_qoff, (void)d_toff, (void)d_w;
	return gd_ksw_batch_dev(ctx, n, d_qseq, d_tseq, d_exact_score, sc, d_score, d_n_cigar, d_cigar, d_cigar_off, h_qoff, h_toff, h_w, stream_, nullptr, nullptr);
}

extern "C" int gdiet_hip_last_dp_work(const gdiet_ctx *ctx, uint64_t *cells, uint64_t *alg_bytes)
{
	if (!ctx) return GDIET_E_PARAM;
	if (cells) *cells = ctx->last_cells;
	if (alg_bytes) *alg_bytes = ctx->last_alg_bytes;
	return GDIET_OK;
}

extern "C" int gdiet_hip_last_kernel_ms(gdiet_ctx *ctx, float *dp_ms, float *bt_ms)
{
	if (!ctx) return GDIET_E_PARAM;
	if (ctx->last_was_async) {
		if (dp_ms) *dp_ms = ctx->async_dp_ms;
		if (bt_ms) *bt_ms = ctx->async_bt_ms;
		return GDIET_OK;
	}
	float a = 0, b = 0;
	GD_HIP(hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]));
	if (ctx->last_split) { // the DP phase ends with the later of the two launches
		float a2 = 0;
		GD_HIP(hipEventElapsedTime(&a2, ctx->ev[0], ctx->ev2[1]));
		a = std::max(a, a2);
	}
	GD_HIP(hipEventElapsedTime(&b, ctx->ev[0], ctx->ev[2]));
	b -= a; // what remains after the last DP wavefront
	if (dp_ms) *dp_ms = a;
	if (bt_ms) *bt_ms = b;
	return GDIET_OK;
}

// ---- host-pointer entry point ------------------------------------------------------------------------------

// The synchronous kernel-level entry points grow and write the context's arena / sequence buffers on ctx->stream; the batches in flight
// of gdiet_hip_map_submit share that arena (dp_mu / arena_ev), so such a call while tickets are open could free the arena under a
// lane's running DP kernel.  Refused, as gdiet_hip_map_uploaded refuses it.
static bool gd_tickets_open(gdiet_ctx *ctx)
{
	std::lock_guard<std::mutex> guard(ctx->async_mu);
	for (int i = 0; i < GD_MAX_INFLIGHT; ++i)
		if (ctx->async_busy[i]) { ctx->err = "batches submitted with gdiet_hip_map_submit are still in flight: wait for their tickets first"; return true; }
	return false;
}

extern "C" int gdiet_hip_ksw_extd2_batch(gdiet_ctx *ctx, int n, const uint8_t *qseq, const int64_t *qoff,
                                         const uint8_t *tseq, const int64_t *toff, const int32_t *w,
                                         const int32_t *exact_score, const gdiet_ksw_score_t *sc, int32_t *score,
                                         int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off)
{
	if (!ctx) return GDIET_E_PARAM;
	if (n <= 0) return GDIET_OK;
	if (!qseq || !qoff || !tseq || !toff || !w || !score || !n_cigar || !cigar || !cigar_off) {
		ctx->err = "NULL argument";
		return GDIET_E_PARAM;
	}
	if (gd_tickets_open(ctx)) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	int rc;
	hipStream_t s = ctx->stream;
	const size_t qb = (size_t)qoff[n], tb = (size_t)toff[n], cb = (size_t)cigar_off[n];
	// layout of the small per-batch arrays behind the sequences: [cigar_off (n+1) i64][exact (n) i32]
	const size_t aux = sizeof(int64_t) * (n + 1) + sizeof(int32_t) * n;
	if ((rc = gd_grow(ctx, ctx->qseq, qb + 64))) return rc;
	if ((rc = gd_grow(ctx, ctx->tseq, tb + 64 + aux + 16))) return rc;
	if ((rc = gd_grow(ctx, ctx->score, sizeof(int32_t) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->ncig, sizeof(int32_t) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->cigar, sizeof(uint32_t) * (cb + 1)))) return rc;
	uint8_t *d_q = (uint8_t *)ctx->qseq.p, *d_t = (uint8_t *)ctx->tseq.p;
	int64_t *d_cigoff = (int64_t *)(d_t + ((tb + 64 + 15) & ~(size_t)15));
	int32_t *d_ex = (int32_t *)(d_cigoff + n + 1);
	GD_HIP(hipMemcpyAsync(d_q, qseq, qb, hipMemcpyHostToDevice, s));
	GD_HIP(hipMemcpyAsync(d_t, tseq, tb, hipMemcpyHostToDevice, s));
	GD_HIP(hipMemcpyAsync(d_cigoff, cigar_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s));
	if (exact_score) GD_HIP(hipMemcpyAsync(d_ex, exact_score, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
	rc = gdiet_hip_ksw_extd2_batch_dev(ctx, n, d_q, nullptr, d_t, nullptr, nullptr, exact_score ? d_ex : nullptr, sc,
	                                   (int32_t *)ctx->score.p, (int32_t *)ctx->ncig.p, (uint32_t *)ctx->cigar.p, d_cigoff,
	                                   qoff, toff, w, s);
	if (rc) return rc;
	GD_HIP(hipMemcpyAsync(score, ctx->score.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
	GD_HIP(hipMemcpyAsync(n_cigar, ctx->ncig.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
	GD_HIP(hipMemcpyAsync(cigar, ctx->cigar.p, sizeof(uint32_t) * cb, hipMemcpyDeviceToHost, s));
	GD_HIP(hipStreamSynchronize(s));
	for (int i = 0; i < n; ++i)
		if (n_cigar[i] > cigar_off[i + 1] - cigar_off[i]) {
			ctx->err = "CIGAR of alignment " + std::to_string(i) + " needs " + std::to_string(n_cigar[i]) + " ops";
			return GDIET_E_CIGAR_CAP;
		}
	return GDIET_OK;
}

// ---- K3: single-affine form ----------------------------------------------------------------------------------
extern "C" int gdiet_hip_ksw_extz2_batch(gdiet_ctx *ctx, int n, const uint8_t *qseq, const int64_t *qoff, const uint8_t *tseq,
                                         const int64_t *toff, const int32_t *w, const gdiet_ksw_score_t *sc, int32_t *score,
                                         int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off)
{
	if (!ctx) return GDIET_E_PARAM;
	if (!sc) { ctx->err = "scoring is NULL"; return GDIET_E_PARAM; }
	gdiet_ksw_score_t s2 = *sc;
	s2.q2 = sc->q, s2.e2 = sc->e; // ksw_extz2(q,e) == ksw_extd2(q,e,q,e) cell for cell in APPROX_MAX mode (see include/gdiet_hip.h)
	ctx->single_affine = true;     // the 16- and 64-lane kernels then run their single-affine form (no X2 / Y2 half)
	const int rc = gdiet_hip_ksw_extd2_batch(ctx, n, qseq, qoff, tseq, toff, w, nullptr, &s2, score, n_cigar, cigar, cigar_off);
	ctx->single_affine = false;
	return rc;
}

// exact-max mode of ksw_extz2 (flag without APPROX_MAX): ksw_extz2_exact.hip.h
extern "C" int gdiet_hip_ksw_extz2_batch_ex(gdiet_ctx *ctx, int n, const uint8_t *qseq, const int64_t *qoff, const uint8_t *tseq,
                                            const int64_t *toff, const int32_t *w, const gdiet_ksw_score_t *sc, int32_t zdrop, int32_t end_bonus,
                                            gdiet_ksw_extz_t *ez, int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off)
{
	if (!ctx) return GDIET_E_PARAM;
	if (n <= 0) return GDIET_OK;
	if (!qseq || !qoff || !tseq || !toff || !w || !sc || !ez || !n_cigar || !cigar || !cigar_off) { ctx->err = "NULL argument"; return GDIET_E_PARAM; }
	if (sc->flag & ~GD_EZ_EXTZ_ONLY) {
		ctx->err = "gdiet_hip_ksw_extz2_batch_ex takes flag 0 or KSW_EZ_EXTZ_ONLY (exact maximum); APPROX_MAX: gdiet_hip_ksw_extz2_batch";
		return GDIET_E_PARAM;
	}
	static_assert(sizeof(gdiet_ksw_extz_t) == sizeof(GdExtzOut), "public and kernel record differ");
	if (gd_tickets_open(ctx)) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	hipStream_t s = ctx->stream;
	KswzConst K;
	K.q = sc->q, K.e = sc->e, K.sc_mch = sc->match, K.sc_mis = sc->mismatch, K.sc_N = sc->sc_ambi == 0 ? -sc->e : sc->sc_ambi;
	K.zdrop = zdrop, K.end_bonus = end_bonus, K.flag = sc->flag;
	{ // :88-90: the reference returns without aligning
		const int min_sc = std::min<int>(std::min<int>(sc->mismatch, sc->match), std::min<int>(sc->sc_ambi, 0));
		if (-min_sc > 2 * (K.q + K.e)) { ctx->err = "-min_sc > 2*(q+e): the reference returns without aligning"; return GDIET_E_PARAM; }
	}
	int rc;
	if ((rc = gd_host_grow(ctx, ctx->h_tasks, sizeof(KswTask) * (size_t)n))) return rc;
	KswTask *h_tasks = (KswTask *)ctx->h_tasks.p;
	size_t bt = 0;
	int max_cap = 0;
	for (int i = 0; i < n; ++i) {
		KswTask &T = h_tasks[i];
		T.qoff = qoff[i], T.toff = toff[i], T.qlen = (int)(qoff[i + 1] - qoff[i]), T.tlen = (int)(toff[i + 1] - toff[i]), T.w = w[i];
		if (T.qlen <= 0 || T.tlen <= 0) { ctx->err = "empty sequence in batch (the reference returns without aligning)"; return GDIET_E_PARAM; }
		T.cig_off = cigar_off[i], T.cig_cap = (int32_t)std::min<int64_t>(cigar_off[i + 1] - cigar_off[i], 0x7fffffff);
		T.exact_score = GD_NEG_INF, T.kind = GD_KIND_GENERIC, T.pad = 0;
		T.row_bytes = gd_ncol16(T.qlen, T.tlen, T.w) * 16;
		T.bt_off = (int64_t)bt;
		bt += gd_align256((size_t)(T.qlen + T.tlen - 1) * (size_t)T.row_bytes + 64);
		max_cap = std::max(max_cap, gd_generic_cap(T.qlen, T.tlen, T.w));
	}
	const size_t lds = (size_t)max_cap * 9;
	if (lds > 160 * 1024 - 1024) { ctx->err = "band wider than the LDS window of the exact-maximum kernel"; return GDIET_E_PARAM; }
	const size_t qb = (size_t)qoff[n], tb = (size_t)toff[n], cb = (size_t)cigar_off[n];
	if ((rc = gd_grow(ctx, ctx->arena, bt))) return rc;
	if ((rc = gd_grow(ctx, ctx->tasks, sizeof(KswTask) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->status, sizeof(int32_t) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->qseq, qb + 64))) return rc;
	if ((rc = gd_grow(ctx, ctx->tseq, tb + 64))) return rc;
	if ((rc = gd_grow(ctx, ctx->score, sizeof(int32_t) * 3 * (size_t)n + sizeof(GdExtzOut) * (size_t)n))) return rc; // score | start (2n) | ez
	if ((rc = gd_grow(ctx, ctx->ncig, sizeof(int32_t) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->cigar, sizeof(uint32_t) * (cb + 1)))) return rc;
	int32_t *d_score = (int32_t *)ctx->score.p, *d_start = d_score + n;
	GdExtzOut *d_ez = (GdExtzOut *)(d_start + 2 * (size_t)n);
	GD_HIP(hipMemcpyAsync(ctx->qseq.p, qseq, qb, hipMemcpyHostToDevice, s));
	GD_HIP(hipMemcpyAsync(ctx->tseq.p, tseq, tb, hipMemcpyHostToDevice, s));
	GD_HIP(hipMemcpyAsync(ctx->tasks.p, h_tasks, sizeof(KswTask) * n, hipMemcpyHostToDevice, s));
	if (lds > 64 * 1024) GD_HIP(hipFuncSetAttribute((const void *)ksw_extz2_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	ctx->last_mask = 2, ctx->last_was_async = false, ctx->last_split = 0;
	GD_HIP(hipEventRecord(ctx->ev[0], s));
	hipLaunchKernelGGL(ksw_extz2_exact_kernel, dim3(n), dim3(64), lds, s, (const KswTask *)ctx->tasks.p, n, (const uint8_t *)ctx->qseq.p, (const uint8_t *)ctx->tseq.p,
	                   (uint8_t *)ctx->arena.p, (int32_t *)ctx->status.p, d_score, d_ez, d_start, K, max_cap);
	GD_HIP(hipEventRecord(ctx->ev[1], s));
	hipLaunchKernelGGL(ksw_backtrack_kernel, dim3((n + 63) / 64), dim3(64), 0, s, (const KswTask *)ctx->tasks.p, n, (const uint8_t *)ctx->arena.p,
	                   (const int32_t *)ctx->status.p, d_score, (int32_t *)ctx->ncig.p, (uint32_t *)ctx->cigar.p, 0, (const int32_t *)nullptr, (const int32_t *)d_start);
	GD_HIP(hipEventRecord(ctx->ev[2], s));
	GD_HIP(hipMemcpyAsync(ez, d_ez, sizeof(GdExtzOut) * n, hipMemcpyDeviceToHost, s));
	GD_HIP(hipMemcpyAsync(n_cigar, ctx->ncig.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
	GD_HIP(hipMemcpyAsync(cigar, ctx->cigar.p, sizeof(uint32_t) * cb, hipMemcpyDeviceToHost, s));
	GD_HIP(hipStreamSynchronize(s));
	GD_HIP(hipGetLastError());
	for (int i = 0; i < n; ++i)
		if (n_cigar[i] > cigar_off[i + 1] - cigar_off[i]) { ctx->err = "CIGAR of alignment " + std::to_string(i) + " needs " + std::to_string(n_cigar[i]) + " ops"; return GDIET_E_CIGAR_CAP; }
	return GDIET_OK;
}

// ksw_exts2 (splice-aware extension; not called by GDiet): ksw_exts2.hip.h
extern "C" int gdiet_hip_ksw_exts2_batch(gdiet_ctx *ctx, int n, const uint8_t *qseq, const int64_t *qoff, const uint8_t *tseq, const int64_t *toff,
                                         const uint8_t *junc, const int8_t *mat, int8_t q, int8_t e, int8_t q2, int8_t noncan, int32_t zdrop,
                                         int8_t junc_bonus, int32_t flag, gdiet_ksw_extz_t *ez, int32_t *n_cigar, uint32_t *cigar, const int64_t *cigar_off)
{
	if (!ctx) return GDIET_E_PARAM;
	if (n <= 0) return GDIET_OK;
	if (!qseq || !qoff || !tseq || !toff || !mat || !ez || !n_cigar || !cigar || !cigar_off) { ctx->err = "NULL argument"; return GDIET_E_PARAM; }
	const int32_t known = GD_EZ_SCORE_ONLY | GD_EZ_RIGHT | GD_EZ_GENERIC_SC | GD_EZ_APPROX_MAX | GD_EZ_APPROX_DROP | GD_EZ_EXTZ_ONLY | GD_EZ_REV_CIGAR |
	                      GD_EZ_SPLICE_FOR | GD_EZ_SPLICE_REV | GD_EZ_SPLICE_FLANK;
	if (flag & ~known) { ctx->err = "gdiet_hip_ksw_exts2_batch: unknown flag bit"; return GDIET_E_PARAM; }
	if (q2 <= q + e || e <= 0) { ctx->err = "ksw_exts2 needs q2 > q + e (and e > 0): the reference returns without aligning"; return GDIET_E_PARAM; } // :72
	if (gd_tickets_open(ctx)) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	hipStream_t s = ctx->stream;
	KswsConst K;
	K.q = q, K.e = e, K.q2 = q2, K.noncan = noncan, K.zdrop = zdrop, K.junc_bonus = junc_bonus, K.flag = flag;
	K.sc_mch = mat[0], K.sc_mis = mat[1], K.sc_N = mat[24] == 0 ? -e : mat[24];
	memcpy(K.mat, mat, 25);
	{ // :86-90
		int min_sc = mat[1];
		for (int t = 1; t < 25; ++t) min_sc = std::min<int>(min_sc, mat[t]);
		if (-min_sc > 2 * (q + e)) { ctx->err = "-min_sc > 2*(q+e): the reference returns without aligning"; return GDIET_E_PARAM; }
	}
	K.long_thres = (q2 - q) / e - 1; // :92-95
	if (q2 > q + e + K.long_thres * e) ++K.long_thres;
	K.long_diff = K.long_thres * e - (q2 - q);
	int rc;
	if ((rc = gd_host_grow(ctx, ctx->h_tasks, sizeof(KswTask) * (size_t)n))) return rc;
	KswTask *h_tasks = (KswTask *)ctx->h_tasks.p;
	size_t bt = 0;
	int max_cap = 0;
	for (int i = 0; i < n; ++i) {
		KswTask &T = h_tasks[i];
		T.qoff = qoff[i], T.toff = toff[i], T.qlen = (int)(qoff[i + 1] - qoff[i]), T.tlen = (int)(toff[i + 1] - toff[i]), T.w = -1;
		if (T.qlen <= 0 || T.tlen <= 0) { ctx->err = "empty sequence in batch (the reference returns without aligning)"; return GDIET_E_PARAM; }
		T.cig_off = cigar_off[i], T.cig_cap = (int32_t)std::min<int64_t>(cigar_off[i + 1] - cigar_off[i], 0x7fffffff);
		T.exact_score = GD_NEG_INF, T.kind = GD_KIND_GENERIC, T.pad = 0;
		T.row_bytes = (((T.qlen < T.tlen ? T.qlen : T.tlen) + 15) / 16 + 1) * 16; // n_col_ * 16, :80
		T.bt_off = (int64_t)bt;
		if (!(flag & GD_EZ_SCORE_ONLY)) bt += gd_align256((size_t)(T.qlen + T.tlen - 1) * (size_t)T.row_bytes + 64);
		max_cap = std::max(max_cap, gd_generic_cap(T.qlen, T.tlen, -1));
	}
	const size_t lds = (size_t)max_cap * 10 + 16;
	if (lds > 160 * 1024 - 1024) { ctx->err = "alignment longer than the LDS window of the ksw_exts2 kernel (min(qlen, tlen) <= ~8000)"; return GDIET_E_PARAM; }
	const size_t qb = (size_t)qoff[n], tb = (size_t)toff[n], cb = (size_t)cigar_off[n];
	if ((rc = gd_grow(ctx, ctx->arena, bt + 256))) return rc;
	if ((rc = gd_grow(ctx, ctx->tasks, sizeof(KswTask) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->qseq, qb + 64))) return rc;
	if ((rc = gd_grow(ctx, ctx->tseq, 2 * tb + 128))) return rc; // target | junction annotation
	if ((rc = gd_grow(ctx, ctx->score, sizeof(GdExtzOut) * (size_t)n))) return rc;
	if ((rc = gd_grow(ctx, ctx->ncig, sizeof(int32_t) * n))) return rc;
	if ((rc = gd_grow(ctx, ctx->cigar, sizeof(uint32_t) * (cb + 1)))) return rc;
	uint8_t *d_junc = junc ? (uint8_t *)ctx->tseq.p + tb + 64 : nullptr;
	GD_HIP(hipMemcpyAsync(ctx->qseq.p, qseq, qb, hipMemcpyHostToDevice, s));
	GD_HIP(hipMemcpyAsync(ctx->tseq.p, tseq, tb, hipMemcpyHostToDevice, s));
	if (junc) GD_HIP(hipMemcpyAsync(d_junc, junc, tb, hipMemcpyHostToDevice, s));
	GD_HIP(hipMemcpyAsync(ctx->tasks.p, h_tasks, sizeof(KswTask) * n, hipMemcpyHostToDevice, s));
	if (lds > 64 * 1024) GD_HIP(hipFuncSetAttribute((const void *)ksw_exts2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	ctx->last_mask = 2, ctx->last_was_async = false, ctx->last_split = 0;
	GD_HIP(hipEventRecord(ctx->ev[0], s));
	hipLaunchKernelGGL(ksw_exts2_kernel, dim3(n), dim3(64), lds, s, (const KswTask *)ctx->tasks.p, n, (const uint8_t *)ctx->qseq.p, (const uint8_t *)ctx->tseq.p,
	                   (const uint8_t *)d_junc, (uint8_t *)ctx->arena.p, (GdExtzOut *)ctx->score.p, (int32_t *)ctx->ncig.p, (uint32_t *)ctx->cigar.p, K, max_cap);
	GD_HIP(hipEventRecord(ctx->ev[1], s));
	GD_HIP(hipEventRecord(ctx->ev[2], s));
	GD_HIP(hipMemcpyAsync(ez, ctx->score.p, sizeof(GdExtzOut) * n, hipMemcpyDeviceToHost, s));
	GD_HIP(hipMemcpyAsync(n_cigar, ctx->ncig.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
	GD_HIP(hipMemcpyAsync(cigar, ctx->cigar.p, sizeof(uint32_t) * cb, hipMemcpyDeviceToHost, s));
	GD_HIP(hipStreamSynchronize(s));
	GD_HIP(hipGetLastError());
	for (int i = 0; i < n; ++i)
		if (n_cigar[i] > cigar_off[i + 1] - cigar_off[i]) { ctx->err = "CIGAR of alignment " + std::to_string(i) + " needs " + std::to_string(n_cigar[i]) + " ops"; return GDIET_E_CIGAR_CAP; }
	return GDIET_OK;
}

#include "map_pipeline.hip.h"
#include "map_multi.h"

// ---- SURVEY 8f rank 4, chaining half: mg_lchain_dp for a batch of reads (lchain.hip.h on the device, lchain_host.h on host threads) ----
#include "lchain.hip.h"
#include "lchain_host.h"
extern "C" int gdiet_hip_lchain_dp_batch(gdiet_ctx *ctx, int n_reads, const uint64_t *a, const int64_t *aoff, int32_t max_dist_x, int32_t max_dist_y,
                                         int32_t bw, int32_t max_skip, int32_t max_iter, int32_t min_cnt, int32_t min_sc, float chn_pen_gap,
                                         float chn_pen_skip, int32_t is_cdna, int32_t n_seg, int32_t *n_u, int64_t *n_v, uint64_t *u, uint64_t *a_out)
{
	if (!ctx) return GDIET_E_PARAM;
	if (n_reads <= 0) return GDIET_OK;
	if (!a || !aoff || !n_u || !n_v || !u || !a_out) { ctx->err = "NULL argument"; return GDIET_E_PARAM; }
	const int64_t tot = aoff[n_reads];
	for (int i = 0; i < n_reads; ++i) {
		if (aoff[i + 1] < aoff[i] || aoff[i + 1] - aoff[i] > 0x7fffffff) { ctx->err = "anchor offsets must ascend (at most 2^31 - 1 anchors per read)"; return GDIET_E_PARAM; }
		n_u[i] = 0, n_v[i] = 0;
	}
	if (tot == 0) return GDIET_OK;
	if (gd_tickets_open(ctx)) return GDIET_E_PARAM;
	(void)hipSetDevice(ctx->device);
	hipStream_t s = ctx->stream;
	GdChainOpt O;
	O.max_dist_x = max_dist_x < bw ? bw : max_dist_x; // SR/lchain.c:136-137
	O.max_dist_y = (max_dist_y < bw && !is_cdna) ? bw : max_dist_y;
	O.bw = bw, O.max_skip = max_skip, O.max_iter = max_iter, O.min_cnt = min_cnt, O.min_sc = min_sc;
	O.chn_pen_gap = chn_pen_gap, O.chn_pen_skip = chn_pen_skip, O.is_cdna = is_cdna, O.n_seg = n_seg;
	int rc;
	// device: anchors | offsets | f | p | v | t   (qseq / tseq / score / status buffers of the context reused as plain workspace)
	if ((rc = gd_grow(ctx, ctx->qseq, sizeof(uint64_t) * 2 * (size_t)tot + 64))) return rc;
	if ((rc = gd_grow(ctx, ctx->tasks, sizeof(int64_t) * ((size_t)n_reads + 1) + 64))) return rc;
	if ((rc = gd_grow(ctx, ctx->score, sizeof(int32_t) * 4 * (size_t)tot + 64))) return rc;
	if ((rc = gd_host_grow(ctx, ctx->h_res, sizeof(int32_t) * 3 * (size_t)tot + 64))) return rc;
	int32_t *d_f = (int32_t *)ctx->score.p, *d_p = d_f + tot, *d_v = d_p + tot, *d_t = d_v + tot;
	int32_t *h_f = (int32_t *)ctx->h_res.p, *h_p = h_f + tot, *h_v = h_p + tot;
	GD_HIP(hipMemcpyAsync(ctx->qseq.p, a, sizeof(uint64_t) * 2 * (size_t)tot, hipMemcpyHostToDevice, s));
	GD_HIP(hipMemcpyAsync(ctx->tasks.p, aoff, sizeof(int64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice, s));
	ctx->last_mask = 0, ctx->last_was_async = false, ctx->last_split = 0;
	GD_HIP(hipEventRecord(ctx->ev[0], s));
	hipLaunchKernelGGL(lchain_fill_kernel, dim3(n_reads), dim3(64), 0, s, n_reads, (const uint64_t *)ctx->qseq.p, (const int64_t *)ctx->tasks.p, O, d_f, d_p, d_v, d_t);
	GD_HIP(hipEventRecord(ctx->ev[1], s));
	GD_HIP(hipEventRecord(ctx->ev[2], s));
	GD_HIP(hipMemcpyAsync(h_f, d_f, sizeof(int32_t) * 3 * (size_t)tot, hipMemcpyDeviceToHost, s));
	GD_HIP(hipStreamSynchronize(s));
	GD_HIP(hipGetLastError());
	gd_parallel_for(ctx, ctx->lane_threads, n_reads, [&](int i) {
		static thread_local std::vector<GdlPair> z, b;
		static thread_local std::vector<int32_t> t;
		static thread_local std::vector<uint64_t> u2;
		const int64_t o = aoff[i], n = aoff[i + 1] - o;
		if (n <= 0) return;
		n_u[i] = gdl_chains_of_read(n, (const GdlPair *)a + o, h_f + o, h_p + o, h_v + o, min_cnt, min_sc, u + o, (GdlPair *)a_out + o, &n_v[i], z, t, b, u2);
	});
	return GDIET_OK;
}

#ifdef GD_SEED_PROF
extern "C" int gdiet_hip_debug_seed_prof(unsigned long long *out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(gd_seed_prof), 64); }
#endif
// ---- read input (input half) ---------------------------------------------------------------------------
#include "fastx_reader.h"
struct gdiet_fastx { GdFastx *r; };

extern "C" int gdiet_hip_fastx_open(gdiet_fastx **fx, const char *path)
{
	if (!fx) return GDIET_E_PARAM;
	*fx = nullptr;
	GdFastx *r = gd_fastx_open(path);
	if (!r) return GDIET_E_PARAM;
	*fx = new gdiet_fastx{r};
	return GDIET_OK;
}

extern "C" int gdiet_hip_fastx_read(gdiet_fastx *fx, int64_t chunk_size, int with_qual, int with_comment, int frag_mode, int32_t *n_reads,
                                    const char *const **names, const char *const **comments, const char *const **seqs,
                                    const char *const **quals, const int32_t **lens)
{
	if (!fx || !fx->r || !n_reads || !names || !seqs || !lens) return GDIET_E_PARAM;
	bool bad = false;
	const int n = fx->r->read_batch(chunk_size, with_qual != 0, with_comment != 0, frag_mode != 0, &bad);
	if (n < 0) return GDIET_E_PARAM;
	*n_reads = n;
	*names = fx->r->v_name.data(), *seqs = fx->r->v_seq.data(), *lens = fx->r->v_len.data();
	if (comments) *comments = fx->r->v_comment.data();
	if (quals) *quals = fx->r->v_qual.data();
	return bad ? GDIET_W_TRUNCATED : GDIET_OK;
}

struct gdiet_fastx_batch { GdFastxBatch *b; };
extern "C" gdiet_fastx_batch *gdiet_hip_fastx_detach(gdiet_fastx *fx)
{
	if (!fx || !fx->r) return nullptr;
	return new gdiet_fastx_batch{gd_fastx_detach(fx->r)};
}
extern "C" void gdiet_hip_fastx_batch_free(gdiet_fastx_batch *b)
{
	if (!b) return;
	delete b->b;
	delete b;
}

extern "C" int gdiet_hip_fastx_set_threads(gdiet_fastx *fx, int n)
{
	if (!fx || !fx->r || n < 1 || n > 64) return GDIET_E_PARAM;
	fx->r->n_threads = n;
	return GDIET_OK;
}

extern "C" void gdiet_hip_fastx_close(gdiet_fastx *fx)
{
	if (!fx) return;
	gd_fastx_close(fx->r);
	delete fx;
}
