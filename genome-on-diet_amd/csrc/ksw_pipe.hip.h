// K1, short alignments (full matrices: 150 x 150 short reads) as a skewed pipeline: see ksw_pipe_core.h for the scheme and why it is
// exact.  One wavefront = NG = 64 / G groups of G lanes; every group works through the `np` alignments of its column of the pipe's
// id table, all of ONE geometry (qlen, tlen), a lane moving on to the next alignment as soon as its block has left the matrix.
// Results (backtrace layout, score, status) are those of ksw_extd2_wave_kernel<10 | 8 | 16>, which it replaces for such batches;
// the walk stays with ksw_backtrack_kernel.
//
// Per step a lane issues the DP of its 16 cells (gdw_compute, as every other form), one ds_read_u8 for the query byte its block meets
// next (the group's query sits in LDS: no lane depends on the lane below for it, and the row loop issues no vector-memory load),
// and one 16-byte backtrace store while its block is inside the matrix.  Target blocks, queries and the per-alignment descriptor are
// staged through LDS one alignment ahead, in two buffers: alignment n + 1 is fetched when the last lane of the group has moved on to
// alignment n (nobody reads the buffer of n - 1 any more), 16 (G - 1) steps into the period -- long before lane 0 needs it.
#pragma once
#include "ksw_wave.hip.h"
#include "ksw_pipe_core.h"

// one wavefront's work: cnt alignments of geometry qlen x tlen, np = ceil(cnt / NG) per group, alignment n of group g = ids[id_off + n * NG + g]
struct PipeWave { int32_t id_off, qlen, tlen, np, row_bytes, cnt, pad[2]; };
// A run of alignments of one geometry = n_waves consecutive PipeWave records.  The planner (host) knows which alignments the run has, not
// which of them the exact-match pre-filter will answer (22 % of a 1 %-error short-read batch): pipe_compact_kernel, between the
// pre-filter and the DP, keeps the ids that are still pending and deals them out to the run's wavefronts in equal shares -- a slot of a
// pipe that holds an answered alignment would cost its qlen + 15 steps all the same.
struct PipeRun { int32_t src_off, m, dst_off, wave_off, n_waves, ng, count, done, np_min, pad[3]; }; // count / done: zero when uploaded, the kernel's counters

// Blocks of GDP_COMPACT_THREADS x GDP_COMPACT_ITEMS ids each, grid (chunks of the longest run, runs).  The order of a run's alignments
// does not matter (every alignment is computed on its own), so a wavefront reserves room for its pending ids with one atomic add on the
// run's counter; the block that finishes last deals the run out to its wavefronts.
#define GDP_COMPACT_THREADS 256
#define GDP_COMPACT_ITEMS 8
__global__ __launch_bounds__(GDP_COMPACT_THREADS) void pipe_compact_kernel(PipeRun *__restrict__ runs, const int32_t *__restrict__ ids,
                                                                         const int32_t *__restrict__ status, int32_t *__restrict__ dst, PipeWave *__restrict__ pipes)
{
	__shared__ int s_last;
	PipeRun &R = runs[blockIdx.y];
	const int m = R.m, per_block = GDP_COMPACT_THREADS * GDP_COMPACT_ITEMS, n_chunks = (m + per_block - 1) / per_block;
	if ((int)blockIdx.x >= n_chunks) return;
	const int lane = threadIdx.x & 63;
	const int32_t *src = ids + R.src_off;
	int32_t *out = dst + R.dst_off;
	int id[GDP_COMPACT_ITEMS], live[GDP_COMPACT_ITEMS], mine = 0;
	const int first = (int)blockIdx.x * per_block + (int)threadIdx.x * GDP_COMPACT_ITEMS;
#pragma unroll
	for (int k = 0; k < GDP_COMPACT_ITEMS; ++k) id[k] = first + k < m ? src[first + k] : -1;
#pragma unroll
	for (int k = 0; k < GDP_COMPACT_ITEMS; ++k) live[k] = id[k] >= 0 && status[id[k]] == GD_ST_PENDING, mine += live[k];
	int incl = mine; // inclusive prefix over the wavefront
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const int v = __shfl_up(incl, d, 64);
		if (lane >= d) incl += v;
	}
	int base = 0;
	if (lane == 63 && incl) base = atomicAdd(&R.count, incl);
	base = __shfl(base, 63, 64);
	int at = base + incl - mine;
#pragma unroll
	for (int k = 0; k < GDP_COMPACT_ITEMS; ++k)
		if (live[k]) out[at++] = id[k];
	// the last block of the run: equal shares, the first (total mod n_waves) wavefronts one alignment more
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(&R.done, 1) == n_chunks - 1;
	__syncthreads();
	if (!s_last) return;
	__threadfence();
	// (what the pre-filter left may be a fraction of the run -- a fifth of a 1 %-error short-read batch: pipes of np_min alignments per group
	// at least, on as many of the run's wavefronts as that takes; the others find cnt == 0 and leave at once)
	const int total = atomicAdd(&R.count, 0);
	int n_act = (total + R.ng * R.np_min - 1) / (R.ng * R.np_min);
	n_act = n_act < 1 ? 1 : n_act > R.n_waves ? R.n_waves : n_act;
	const int q = total / n_act, rem = total % n_act;
	for (int w = threadIdx.x; w < R.n_waves; w += GDP_COMPACT_THREADS) {
		const int cnt = w < n_act ? q + (w < rem) : 0, a0 = w < n_act ? w * q + (w < rem ? w : rem) : 0;
		PipeWave &W = pipes[R.wave_off + w];
		W.id_off = R.dst_off + a0, W.cnt = cnt, W.np = (cnt + R.ng - 1) / R.ng;
	}
}

// Four wavefronts per workgroup, each with a pipe and an LDS region of its own and no barrier between them: the dispatcher then puts one
// on each SIMD of a CU (single-wavefront workgroups were spread unevenly: 2 743 of 3 922 wavefronts resident on average, the kernel as
// long as the fullest SIMD needs).
#define GDP_BLOCK_WAVES 4
template <bool DUAL>
__global__ __launch_bounds__(64 * GDP_BLOCK_WAVES) __attribute__((amdgpu_waves_per_eu(4))) void ksw_extd2_pipe_kernel(const KswTask *__restrict__ tasks, const int32_t *__restrict__ task_ids,
                                                           const PipeWave *__restrict__ pipes, int n_pipes,
                                                           const uint8_t *__restrict__ qseq, const uint8_t *__restrict__ tseq,
                                                           uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                           int32_t *__restrict__ score_out, WaveK K)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds_all[GDP_BLOCK_WAVES][2 * GDP_BUF_BYTES];
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int pw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * GDP_BLOCK_WAVES + wv);
	if (pw >= n_pipes) return;
	uint8_t *const lds = lds_all[wv];
	const int id_off = __builtin_amdgcn_readfirstlane(pipes[pw].id_off), np = __builtin_amdgcn_readfirstlane(pipes[pw].np), cnt = __builtin_amdgcn_readfirstlane(pipes[pw].cnt);
	if (np <= 0) return; // (a wavefront the compaction left without work; no barrier binds the wavefronts of a workgroup)
	const int row_bytes = __builtin_amdgcn_readfirstlane(pipes[pw].row_bytes);
	const PipeGeo Gm = gd_pipe_geo(__builtin_amdgcn_readfirstlane(pipes[pw].qlen), __builtin_amdgcn_readfirstlane(pipes[pw].tlen));
	const int G = Gm.G, NG = Gm.NG, P = Gm.P, qlen = Gm.qlen, tlen = Gm.tlen;
	// lane -> (group, block); the 64 - NG * G (< G) spare lanes shadow the first blocks of group 0 without ever being live
	int grp = 0, sub = lane;
	while (sub >= G) sub -= G, ++grp; // (once per wavefront)
	const bool spare = grp >= NG;
	if (spare) grp = 0;
	const u32 m_first = sub == 0 ? ~0u : 0u;
	const int nvalid = spare ? 0 : gdp_valid_rows(Gm, sub);

	// alignment nn of every group into buffer nn & 1: each lane its 16 target and 16 (+ 16 for two lanes: the zero padding up to
	// P + 1 bytes and beyond) query bytes, lane 0 of the group the descriptor {backtrace offset, task id, live}
	auto fetch = [&](const int nn) __attribute__((always_inline)) {
		int tid_ = -1, live_ = 0;
		int64_t qo = 0, to = 0, bo = 0;
		if (!spare && nn * NG + grp < cnt) tid_ = task_ids[id_off + nn * NG + grp];
		if (tid_ >= 0) {
			live_ = status[tid_] == GD_ST_PENDING; // (else the exact-match pre-filter answered this one)
			qo = tasks[tid_].qoff, to = tasks[tid_].toff, bo = tasks[tid_].bt_off;
		}
		// (one 16-byte piece at a time: gathered, stored, forgotten -- the DP state of the lane stays in its registers meanwhile)
		auto piece = [&](const uint8_t *__restrict__ src, const int first, const int len, uint8_t *dst) __attribute__((always_inline)) {
			u32 w4[4] = {0, 0, 0, 0};
			if (live_) {
#pragma unroll
				for (int b = 0; b < 16; ++b)
					if (first + b < len) w4[b >> 2] |= (u32)src[first + b] << (8 * (b & 3));
			}
			*reinterpret_cast<uint4 *>(dst) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
		};
		if (!spare) {
			uint8_t *B = lds + (nn & 1) * Gm.BS;
			piece(tseq + to, 16 * sub, tlen, B + Gm.TOFF + grp * Gm.TS + 16 * sub);
			piece(qseq + qo, 16 * sub, qlen, B + grp * Gm.QS + 16 * sub);
			if (sub < 2) piece(qseq + qo, 16 * (sub + G), qlen, B + grp * Gm.QS + 16 * (sub + G));
			if (sub == 0) *reinterpret_cast<uint4 *>(B + Gm.DOFF + grp * 16) = make_uint4((u32)(uint64_t)bo, (u32)((uint64_t)bo >> 32), (u32)tid_, (u32)live_);
		}
		// (the region belongs to this wavefront alone and a wavefront's LDS operations complete in order: only the compiler must not move
		// the other lanes' reads above these writes)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};

	WaveLane L;
	{
		const u32 z[4] = {0, 0, 0, 0};
		gdp_start(L, K, z, sub);
	}
	int left = 0, Rf = 0, tid = -1;
	u32 qaddr = 0;
	uint8_t *p_row = bt;
	bool any_tn = false;
	// row r - 1 values of the lane below, taken at the END of a step: the lane below may start its next alignment before the step
	// that consumes them (its last row on this alignment is the very row the cell above it still needs: ksw_pipe_core.h)
	u32 pX = 0, pV = 0, pX2 = 0;
	u32 qb = 0; // the query byte of the next step, read a step ahead (the LDS latency then lies behind a whole row of arithmetic)
	// the boundary key of row rA (v1 of block 0, u of the cell t == r): it changes on four rows of a period only
	int ekey = K.key_open;
	const int lt = K.long_thres;
	fetch(0);
	// One step.  KK = 0..7: the lanes of block `jsw` (mask m_reset: their half of register KK) are in their first 16 rows and reset the
	// cell t == r in register KK; KK = -1: the generic form of the reset (a chunk cut short by the end of the period); KK = -2: none.
	// TRACK: the chunk in which block mlast reaches the last query row (below).
	auto step = [&](const int rA, const int jsw, const bool first, const u32 m_reset, const bool track, auto ktag) __attribute__((always_inline)) {
		constexpr int KK = decltype(ktag)::value;
		if (rA == 0 || rA == 1 || rA == lt || rA == lt + 1) ekey = gdw_edge_key(K, rA);
		WaveRow W;
		W.r = rA, W.st0 = 0, W.en0 = 0, W.st_ = 0, W.en_ = 0, W.up = 0;
		W.use_array = 0, W.set_tr = 1, W.ukey = ekey, W.v1key = ekey;
		W.m_first_valid = 1, W.m_first = m_first;
		if (KK >= 0) {
			L.U[KK] = gdw_bfi(m_reset, gdw_pack2(ekey), L.U[KK]);
			L.Y[KK] = gdw_bfi(m_reset, K.cy, L.Y[KK]);
			if (DUAL) L.Y2[KK] = gdw_bfi(m_reset, K.cy2, L.Y2[KK]);
		} else if (KK == -1) gdw_reset_tr(L, K, W);
		gdp_query_scores(L, K, qb, any_tn);
		qb = lds[qaddr]; // (for the next step; a lane that changes alignment first re-reads it)
		++qaddr;
		u32 out[4];
		gdw_compute<DUAL>(L, K, W, pX, pV, pX2, out);
		pX = gdw_ror1<64>(L.X[7]), pV = gdw_ror1<64>(L.V[7]), pX2 = DUAL ? gdw_ror1<64>(L.X2[7]) : 0u; // for the next step
		if (left > 0) *reinterpret_cast<uint4 *>(p_row) = make_uint4(out[0], out[1], out[2], out[3]);
		p_row += row_bytes;
		L.R += gdw_lo(L.V[0]) - K.B1;
		if (first && jsw >= 1 && jsw < G) { // block jsw enters the matrix: its tracker from the block below
			const int h = (int)gdw_ror1<64>((u32)gdw_track_handoff(L));
			if (sub == jsw) L.R = h + gdw_lo(L.U[0]);
		}
		// The score = H of the last cell.  Block mlast tracks H at its cell 0 down to the last query row (r* = 16 mlast + qlen - 1, the
		// first row of chunk mlast - 1 of the NEXT period: r* - P = 16 (mlast - 1)); there the tracker walks along the anti-diagonal to the
		// last target column (sl cells, all on rows of the query) and follows that column down for the sl rows that remain.
		if (track && sub == Gm.mlast && left > 0) {
			if (first) Rf = gdw_track_to_slot(L, Gm.sl);
			else Rf += gdw_cell(L.V, Gm.sl) - K.B1;
		}
		--left;
		__builtin_amdgcn_sched_barrier(0); // (the steps of an unrolled chunk one after the other: interleaved they do not fit the registers)
	};
	for (int n = 0; n <= np; ++n) { // period n: lane 0 of every group starts alignment n (n == np: the last alignments drain)
		for (int rA = 0; rA < P;) {
			const int jsw = rA >> 4; // rows [16 jsw, 16 jsw + 15] of the period: block jsw changes alignment on the first of them
			if (jsw < G) {
				if (sub == jsw) {
					if (tid >= 0 && sub == Gm.mlast) score_out[tid] = Rf >> 3, status[tid] = GD_ST_DONE; // (a live alignment, all rows done)
					const uint8_t *B = lds + (n & 1) * Gm.BS;
					const uint4 d = *reinterpret_cast<const uint4 *>(B + Gm.DOFF + grp * 16);
					const uint4 t4 = *reinterpret_cast<const uint4 *>(B + Gm.TOFF + grp * Gm.TS + 16 * sub);
					const u32 tb[4] = {t4.x, t4.y, t4.z, t4.w};
					gdp_start(L, K, tb, sub);
					const bool live = n < np && d.w != 0 && !spare;
					tid = live ? (int)d.z : -1;
					left = live ? nvalid : 0;
					Rf = 0;
					qaddr = (u32)((n & 1) * Gm.BS + grp * Gm.QS);
					qb = lds[qaddr];
					++qaddr;
					p_row = bt + (((uint64_t)d.y << 32) | d.x) + (size_t)sub * 16 + (size_t)(16 * sub) * (size_t)row_bytes;
				}
				any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0;
				if (jsw == G - 1) {
					if (n == np) return; // every lane has handed in its last alignment
					fetch(n + 1);
				}
			}
			const bool track = jsw == Gm.mlast - 1;
			if (jsw < G && P - rA >= 16) {
				// sixteen rows with the cell t == r of block jsw in register 0..7, low then high half
#pragma unroll 1
				for (int half = 0; half < 2; ++half) {
					const u32 m_reset = sub == jsw ? (half ? 0xffff0000u : 0x0000ffffu) : 0u;
					step(rA, jsw, half == 0, m_reset, track, std::integral_constant<int, 0>());
					step(rA + 1, jsw, false, m_reset, track, std::integral_constant<int, 1>());
					step(rA + 2, jsw, false, m_reset, track, std::integral_constant<int, 2>());
					step(rA + 3, jsw, false, m_reset, track, std::integral_constant<int, 3>());
					step(rA + 4, jsw, false, m_reset, track, std::integral_constant<int, 4>());
					step(rA + 5, jsw, false, m_reset, track, std::integral_constant<int, 5>());
					step(rA + 6, jsw, false, m_reset, track, std::integral_constant<int, 6>());
					step(rA + 7, jsw, false, m_reset, track, std::integral_constant<int, 7>());
					rA += 8;
				}
			} else {
				const int r_stop = P - rA < 16 ? P : rA + 16;
				if (jsw < G) {
#pragma unroll 1
					for (bool first = true; rA < r_stop; ++rA, first = false) step(rA, jsw, first, 0u, track, std::integral_constant<int, -1>());
				} else {
#pragma unroll 1
					for (bool first = true; rA < r_stop; ++rA, first = false) step(rA, jsw, first, 0u, false, std::integral_constant<int, -2>());
				}
			}
		}
	}
}

// ids: the batch's id list (the runs' alignments as the planner listed them); dst: room for as many ids as the runs hold
static inline void gd_launch_pipe(const KswTask *tasks, const int32_t *all_ids, PipeRun *runs, int n_runs, int max_run, int32_t *dst, PipeWave *pipes, int n_pipes,
                                  const uint8_t *q, const uint8_t *t, uint8_t *bt, int32_t *status, int32_t *score, KswConst C, hipStream_t s, bool single)
{
	if (n_pipes <= 0 || n_runs <= 0) return;
	const int per_block = GDP_COMPACT_THREADS * GDP_COMPACT_ITEMS;
	hipLaunchKernelGGL(pipe_compact_kernel, dim3((max_run + per_block - 1) / per_block, n_runs), dim3(GDP_COMPACT_THREADS), 0, s, runs, all_ids, status, dst, pipes);
	const int32_t *ids = dst;
	WaveK K;
	gdw_make_consts(C, K);
	const dim3 grid((n_pipes + GDP_BLOCK_WAVES - 1) / GDP_BLOCK_WAVES), block(64 * GDP_BLOCK_WAVES);
	// GDIET_PIPE_WAVES = 3 / 2: at most that many wavefronts of this kernel per SIMD (a workgroup puts one on each SIMD of its CU; an unused
	// dynamic LDS allocation caps the workgroups per CU), the registers of the fourth stay free for the other kernels of the batches in flight
	static const int cap_waves = getenv("GDIET_PIPE_WAVES") ? atoi(getenv("GDIET_PIPE_WAVES")) : 4;
	size_t lds = 0;
	if (cap_waves == 2 || cap_waves == 3) lds = (size_t)(160 * 1024) / (cap_waves + 1) + 1024 - (size_t)GDP_BLOCK_WAVES * 2 * GDP_BUF_BYTES;
	if (single) hipLaunchKernelGGL((ksw_extd2_pipe_kernel<false>), grid, block, lds, s, tasks, ids, pipes, n_pipes, q, t, bt, status, score, K);
	else hipLaunchKernelGGL((ksw_extd2_pipe_kernel<true>), grid, block, lds, s, tasks, ids, pipes, n_pipes, q, t, bt, status, score, K);
}
