// K1 (register-resident form): one 64-lane wavefront per alignment, DP state in VGPRs, lane-to-lane dependency
// through DPP wave_ror:1, two cells per packed 16-bit VALU instruction, 16 B of backtrace per lane per
// anti-diagonal written with one coalesced global_store_dwordx4 (1 KiB per wavefront-row).
// All arithmetic lives in ksw_wave_core.h (shared with the host lock-step emulator); this file is the row loop.
//
// Roofline note: per anti-diagonal a wavefront issues ~310 VALU + ~120 SALU instructions for 1024 cells and
// stores 1024 B: the kernel is bound by VALU issue (one wave64 instruction per ~4.2 cycles per SIMD, whatever its encoding:
// profiles/r02_valu_issue.md), at 0.22-0.23 of what the 8 TB/s HBM roofline would allow for the 1 B/cell backtrace.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ksw_common.h"
#include "ksw_wave_core.h"

// lane l receives the value of lane l-1 (mod LANES): DPP wave_ror:1 (GFX9 encoding 0x13C) for the 64-lane form,
// row_ror:1 (0x121; a DPP "row" is 16 lanes) for the 16-lane form
template <int LANES> __device__ __forceinline__ u32 gdw_ror1(u32 v)
{
	// every lane has a source lane under a rotate, so no "old" value is needed: bound_ctrl form, no register to pre-clear
	// (groups of 10 or 8 lanes hold all blocks of their alignment, block b in lane b of the group: lane b-1 is simply the previous
	// lane of the wavefront, and block 0 never looks at what it receives)
	if (LANES != 16) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x13C, 0xF, 0xF, true);
	return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x121, 0xF, 0xF, true);
}

// the one fresh query byte per anti-diagonal, read through the scalar cache: an aligned dword at a wave-uniform address
// (s_load_dword), so that the row loop issues no vector-memory load at all and never has to wait on its own backtrace stores
__device__ __forceinline__ u32 gdw_seam_byte(const uint8_t *query, int qlen, int j)
{
	if (j < 0 || j >= qlen) return 0u;
	const uintptr_t a = (uintptr_t)(query + j);
	// constant address space + wave-uniform address => s_load_dword; the aligned dword holding a valid byte stays inside its allocation
	typedef const __attribute__((address_space(4))) u32 *gdw_const_u32p;
	const u32 wd = *(gdw_const_u32p)(a & ~(uintptr_t)3);
	return (wd >> (8 * (a & 3))) & 0xffu;
}

// gd_band for wave-uniform arguments, kept on the scalar unit (left to itself the compiler evaluates en0 with v_min3_i32 and two
// v_readfirstlane_b32: five vector instructions per anti-diagonal)
__device__ __forceinline__ void gdw_band_uniform(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	r = __builtin_amdgcn_readfirstlane(r); // (wave-uniform by contract; the compiler cannot always prove it)
	int a = r - qlen + 1, b = (r - w + 1) >> 1, c = (r + w) >> 1, t1 = tlen - 1, st, en;
	asm("s_max_i32 %0, %1, %2\n\ts_max_i32 %0, %0, 0" : "=&s"(st) : "s"(a), "s"(b) : "scc");
	asm("s_min_i32 %0, %1, %2\n\ts_min_i32 %0, %0, %3" : "=&s"(en) : "s"(t1), "s"(r), "s"(c) : "scc");
	st0 = st, en0 = en;
}

__device__ __forceinline__ const uint8_t *gdw_uniform_ptr(const uint8_t *p, int src_lane)
{
	const uintptr_t a = (uintptr_t)p;
	const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)a, src_lane), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(a >> 32), src_lane);
	return (const uint8_t *)((uintptr_t)hi << 32 | lo);
}

static inline bool gd_wave_scoring_ok(const KswConst &C)
{
	WaveK K;
	if (!gdw_make_consts(C, K)) return false;
	// no 8-bit wrap-around anywhere in the reference recurrence for these magnitudes (see ksw_wave_core.h)
	const int mis = C.sc_mis < 0 ? -C.sc_mis : C.sc_mis, n = C.sc_N < 0 ? -C.sc_N : C.sc_N;
	return C.sc_mch > 0 && C.sc_mch + 2 * (C.q2 + C.e2) + (mis > n ? mis : n) + (C.q + C.e) <= 120;
}

static inline bool gd_wave_supported(int qlen, int tlen, int w, int lanes)
{
	return qlen + tlen < (1 << 21) && gd_wave_geometry_ok(qlen, tlen, w, lanes); // (a row's offset into the backtrace is a 32-bit quantity)
}

// LANES == 64: one alignment per wavefront (task_ids[slot]).
// LANES == 16: four alignments OF IDENTICAL GEOMETRY (qlen, tlen, w) per wavefront, one per DPP row of 16 lanes
//              (task_ids[4*slot + row]; -1 = empty row, which shadows row 0 without storing).  Identical geometry keeps every
//              band / boundary quantity of the row loop wave-uniform (SGPRs), exactly as in the 64-lane form; only the
//              sequence and backtrace pointers differ between the rows.  Short reads (150 x 150) all share one geometry.
// LANES == 10 / 8: six / eight alignments of identical geometry per wavefront, for targets of at most 160 / 128 bases: every
//              16-cell block of such an alignment has a lane of its own for the whole run, so a group needs no ring and can be any
//              run of consecutive lanes.  A 150 x 150 alignment uses 10 of the 16 lanes of a DPP row; six groups of ten fill 60 of
//              the 64 lanes (lanes 60-63 shadow group 0 without storing).
// TAG only names the launch (0: a whole batch; 1 / 2: the head / tail launch of a split batch, see gdiet_hip.hip) so that a
// profile lists them apart.
// DUAL = false is the single-affine (ksw_extz2) form of the same kernel: see gdw_compute.
// Every wavefront of the 64-lane kernel stamps s_memtime (the shader clock) and s_memrealtime (the constant 100 MHz reference) around its
// DP rows -- four scalar stores per wavefront, nothing kept in registers in between -- so that the clock the kernel really sustained can be
// reported beside its duration (gdiet_hip_last_dp_clock; bench.py roofline.sclk_mhz; profiles/r03_clock.md).  The table is shared by
// the contexts of a process: with several DP kernels in flight the stamps of a slot are those of whichever wrote last; any of them do.
#define GD_CLOCK_SLOTS 16384
__device__ unsigned long long gd_clock_stamps[GD_CLOCK_SLOTS * 4];
template <int LANES, int TAG = 0, bool DUAL = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LANES == 64 ? 5 : 4))) // 64-lane form: 5 wavefronts per SIMD = at most 96 VGPRs
void ksw_extd2_wave_kernel(const KswTask *__restrict__ tasks,
                                                             const int32_t *__restrict__ task_ids, int n_slots,
                                                             const uint8_t *__restrict__ qseq,
                                                             const uint8_t *__restrict__ tseq,
                                                             uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                             int32_t *__restrict__ score_out, WaveK K,
                                                             int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar)
{
	constexpr int NG = 64 / LANES; // alignments per wavefront
	const int lane = threadIdx.x & 63;
	const bool spare = lane >= NG * LANES; // (10-lane groups only: lanes 60-63)
	const int sub = spare ? lane - NG * LANES : lane % LANES, row = spare ? 0 : lane / LANES;
	const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
	if (slot >= n_slots) return;
	int tid, live = 1;
	if (LANES == 64) {
		tid = __builtin_amdgcn_readfirstlane(task_ids[slot]);
		if (__builtin_amdgcn_readfirstlane(status[tid]) != GD_ST_PENDING) return;
	} else {
		tid = task_ids[NG * slot + row];
		const int tid0 = __builtin_amdgcn_readfirstlane(tid);
		if (tid < 0 || spare) tid = tid0, live = 0;
		else if (status[tid] != GD_ST_PENDING) live = 0; // the exact-match pre-filter answered this one
		if (!__builtin_amdgcn_ballot_w64(live)) return;
	}
	const KswTask *Tp = tasks + tid;
	// (clock stamps, indexed by the task id -- live to the end of the kernel anyway -- so that the row loop carries nothing for them)
	if (LANES == 64 && lane == 0) gd_clock_stamps[(tid & (GD_CLOCK_SLOTS - 1)) * 4] = __builtin_amdgcn_s_memtime(), gd_clock_stamps[(tid & (GD_CLOCK_SLOTS - 1)) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
	const int qlen = __builtin_amdgcn_readfirstlane(Tp->qlen), tlen = __builtin_amdgcn_readfirstlane(Tp->tlen);
	int w = __builtin_amdgcn_readfirstlane(Tp->w);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const uint8_t *query = qseq + Tp->qoff, *target = tseq + Tp->toff;
	uint8_t *p = bt + Tp->bt_off + (size_t)sub * 16;
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;
	// wave-uniform copies of the query pointers for the scalar seam load
	const uint8_t *qg[NG];
#pragma unroll
	for (int g = 0; g < NG; ++g) qg[g] = gdw_uniform_ptr(query, g * LANES);

	// The 16 backtrace bytes of a lane.  64-lane form: the alignment's backtrace is a buffer resource (base in SGPRs), the row a scalar
	// offset and the lane a constant 32-bit vector offset, so the row loop carries no 64-bit vector address (2-3 VGPRs of the 96).
	typedef u32 gdw_u32x4 __attribute__((ext_vector_type(4)));
	__amdgpu_buffer_rsrc_t bt_rsrc = __builtin_amdgcn_make_buffer_rsrc(bt + (LANES == 64 ? Tp->bt_off : 0), 0, (int)0xffffffffu, 0x00020000);
	auto store_row = [&](const int r, const u32 out[4]) __attribute__((always_inline)) {
		if (LANES == 64) {
			const gdw_u32x4 d = {out[0], out[1], out[2], out[3]};
			__builtin_amdgcn_raw_buffer_store_b128(d, bt_rsrc, lane * 16, r * (LANES * 16), 0); // (qlen + tlen < 2^21: gd_wave_supported)
		} else if (live) *reinterpret_cast<uint4 *>(p + (size_t)r * (LANES * 16)) = make_uint4(out[0], out[1], out[2], out[3]);
	};
	WaveLane L;
	gdw_load_block(L, K, sub, 0, query, qlen, target, tlen);
	bool any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0; // wave-uniform: does any lane hold a target N?
	int prev_st_ = 0, prev_st0 = -1, prev_up = -1, prev_en0 = -1, have_f = 0, Rf = 0;
	// One anti-diagonal.  STEADY rows are those in the middle of the matrix -- no cell t == r left to reset, the window no longer
	// starts at block 0, the last target column not yet reached, r > 0 -- for which the boundary-key ladder, gdw_reset_tr and the
	// final tracker are compiled out: 28 000 of the 30 000 anti-diagonals of a HiFi alignment, -3 % kernel time.
	auto dp_row = [&](const int r, auto steady_tag) __attribute__((always_inline)) {
		constexpr bool STEADY = decltype(steady_tag)::value;
		WaveRow W;
		W.r = r;
		gdw_band_uniform(r, qlen, tlen, w, W.st0, W.en0);
		W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
		W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
		const int advanced = W.st_ > prev_st_;
		W.use_array = advanced;
		W.v1key = STEADY ? K.key_open : (W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open);
		W.set_tr = STEADY ? 0 : (W.en0 | 15) >= r;
		W.ukey = STEADY ? 0 : gdw_edge_key(K, r);
		// (1) row r-1 values of the previous lane, fetched before any lane is touched
		const u32 pX = gdw_ror1<LANES>(L.X[7]), pV = gdw_ror1<LANES>(L.V[7]), pX2 = gdw_ror1<LANES>(L.X2[7]), pQ = gdw_ror1<LANES>(L.Qc[3]);
		// (2) query window advance; the lane whose block fell below the window takes over block +LANES
		if (STEADY || r > 0) {
			const int j = r - (prev_st_ << 4);
			u32 seam = gdw_seam_byte(qg[0], qlen, j);
#pragma unroll
			for (int g = 1; g < NG; ++g) {
				const u32 sg = gdw_seam_byte(qg[g], qlen, j);
				seam = row == g ? sg : seam;
			}
			gdw_shift_query(L, pQ, L.blk == prev_st_, seam);
		}
		if (advanced) {
			if (L.blk < W.st_) gdw_load_block(L, K, L.blk + LANES, r, query, qlen, target, tlen);
			any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0;
		}
		// (3) scalar fix-ups and the score row
		if (!STEADY && W.set_tr) gdw_reset_tr(L, K, W);
		if (W.st0 != prev_st0 || W.up != prev_up || advanced) gdw_make_sel(L, W.st0, W.up);
		gdw_update_scores(L, K, any_tn);
		// (4) DP cells of the lanes inside the reference's 16-aligned window
		if (L.blk <= W.en_) {
			u32 out[4];
			gdw_compute<DUAL>(L, K, W, pX, pV, pX2, out);
			store_row(r, out);
		}
		// (5) score trackers
		if (!STEADY && r == 0) L.R = gdw_lo(L.V[0]) - K.B1 - K.qe8;
		else L.R += gdw_lo(L.V[0]) - K.B1;
		if ((STEADY || r > 0) && W.en0 != prev_en0 && (W.en0 & 15) == 0) {
			const int h = (int)gdw_ror1<LANES>((u32)gdw_track_handoff(L));
			if (L.blk == W.en_) L.R = h + gdw_lo(L.U[0]);
		}
		if (!STEADY && W.en0 == tlen - 1) {
			if (L.blk == mlast) {
				if (!have_f) Rf = gdw_track_to_slot(L, sl);
				else Rf += gdw_cell(L.V, sl) - K.B1;
			}
			have_f = 1;
		}
		prev_st_ = W.st_, prev_st0 = W.st0, prev_up = W.up, prev_en0 = W.en0;
	};
	// The rows [rA, rS) in the middle of a long alignment come in pairs: on the first (r - w + 1 even) st0 has just moved up by one
	// cell, on the second en0 has; both are pure functions of m = (r - w + 1) >> 1 there (gdw_steady_rows).  Written out as a pair,
	// the band limits need no min / max, the selectors are rebuilt only when they change, and what can happen only on one of the two
	// rows (a block retiring; a block entering the band) is tested only there: -40 scalar and -10 vector instructions per row.
	const int nblkA = (w - 1 + 16) >> 4, nblkB = (w + 16) >> 4; // blocks whose scores are rewritten on the first / second row of a pair
	u32 m_lowest = 0; // 0 / ~0: this lane holds the lowest block of the window (of the row just done)
	auto pair_row = [&](const int r, const int m, auto a_tag) __attribute__((always_inline)) {
		constexpr bool ROW_A = decltype(a_tag)::value;
		WaveRow W;
		W.r = r, W.st0 = m, W.en0 = ROW_A ? m + w - 1 : m + w;
		W.st_ = m >> 4, W.en_ = W.en0 >> 4;
		W.up = m + ((ROW_A ? nblkA : nblkB) << 4);
		const int advanced = ROW_A && (m & 15) == 0;
		const int pst_ = W.st_ - advanced; // st_ of the row before
		W.use_array = advanced, W.v1key = K.key_open, W.set_tr = 0, W.ukey = 0;
		const u32 pX = gdw_ror1<LANES>(L.X[7]), pV = gdw_ror1<LANES>(L.V[7]), pX2 = gdw_ror1<LANES>(L.X2[7]), pQ = gdw_ror1<LANES>(L.Qc[3]);
		{
			const int j = r - (pst_ << 4);
			u32 seam = gdw_seam_byte(qg[0], qlen, j);
#pragma unroll
			for (int g = 1; g < NG; ++g) {
				const u32 sg = gdw_seam_byte(qg[g], qlen, j);
				seam = row == g ? sg : seam;
			}
			gdw_shift_query_m(L, pQ, m_lowest, seam); // (the lane mask of the row before: its st_ is this row's pst_)
		}
		if (advanced) {
			if (L.blk < W.st_) gdw_load_block(L, K, L.blk + LANES, r, query, qlen, target, tlen);
			any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0;
		}
		if (ROW_A || nblkA != nblkB) {
			u32 lo[4], hi[4];
			gdw_sel_uniform(m & 15, lo, hi);
			gdw_pick_sel(L, W.st_, W.up >> 4, (W.up >> 4) - W.st_ >= LANES - 1, lo, hi, m_lowest);
		}
		W.m_first_valid = 1, W.m_first = m_lowest; // the lane mask "holds the lowest block", for the boundary scalars in gdw_compute
		gdw_update_scores(L, K, any_tn);
		if (L.blk <= W.en_) {
			u32 out[4];
			gdw_compute<DUAL>(L, K, W, pX, pV, pX2, out);
			store_row(r, out);
		}
		L.R += gdw_lo(L.V[0]); // (the bias B1 of every V key is taken off once, after the loop: one sign-extending add per row)
		if (!ROW_A && (W.en0 & 15) == 0) {
			const int h = (int)gdw_ror1<LANES>((u32)gdw_track_handoff(L));
			if (L.blk == W.en_) L.R = h + gdw_lo(L.U[0]);
		}
	};
	{
		// steady rows: (en0 | 15) < r and st0 >= 16 hold from r = w + 48 on (en0 <= (r + w) / 2, st0 >= (r - w + 1) / 2); the last target
		// column is reached at r = max(tlen - 1, 2 (tlen - 1) - w)
		int rA, rS;
		gdw_steady_rows(qlen, tlen, w, rA, rS);
		const int t1_ = tlen - 1, rB0 = 2 * t1_ - w, rB = rB0 > t1_ ? rB0 : t1_;
		int r = 0;
		for (; r <= rend && r < rA; ++r) dp_row(r, std::false_type());
		if (r == rA && rS > rA) {
			int m = (rA - w + 1) >> 1;
			m_lowest = L.blk == prev_st_ ? ~0u : 0u;
			for (; r < rS; r += 2, ++m) {
				pair_row(r, m, std::true_type());
				pair_row(r + 1, m, std::false_type());
			}
			L.R -= (rS - rA) * K.B1; // the trackers of the paired rows accumulated the V keys with their bias (every lane alike, also across the hand-overs)
			--m; // the band of the last row, for the rows that follow
			prev_st_ = m >> 4, prev_st0 = m, prev_up = m + (nblkB << 4), prev_en0 = m + w;
		}
		for (; r <= rend && r < rB; ++r) dp_row(r, std::true_type());
		for (; r <= rend; ++r) dp_row(r, std::false_type());
	}
	// The 64-lane form (one alignment per wavefront) walks its own alignment back right away when given the CIGAR buffers: the
	// walk is latency-bound and overlaps the DP of the other resident wavefronts, instead of a separate pass after the last one.
	const bool fuse = cigar != nullptr; // (the grouped forms: given the CIGAR buffers, every group's first lane walks its alignment back, below)
	if (LANES == 64 && lane == 0) // (the DP rows only: the walk below is latency-bound)
		gd_clock_stamps[(tid & (GD_CLOCK_SLOTS - 1)) * 4 + 2] = __builtin_amdgcn_s_memtime(), gd_clock_stamps[(tid & (GD_CLOCK_SLOTS - 1)) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
	if (L.blk == mlast && live) {
		score_out[tid] = Rf >> 3;
		status[tid] = fuse ? GD_ST_TRACED : GD_ST_DONE;
	}
	if (fuse) {
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); // this wavefront's own backtrace stores: complete, and not served from a stale L1 line
		if (LANES == 64) gd_bt_wave_walk(*Tp, tid, bt, n_cigar, cigar, lane);
		else if (live && sub == 0) {
			// short alignments, several per wavefront: one walk per group on its first lane -- a few hundred dependent steps through bytes
			// the wavefront has just written, while the SIMD's other wavefronts are in their DP rows; no separate backtrack pass after
			// the last DP wavefront (ksw_backtrack_kernel: 1.6-2.1 ms per 262 144 short reads)
			const KswTask T = *Tp;
			gd_bt_thread_walk(T, tid, bt, n_cigar, cigar, T.tlen - 1, T.qlen - 1);
		}
	}
}

static inline int gd_wave64_block()
{
	static int bs = 0;
	if (!bs) { const char *e = getenv("GDIET_WAVE64_BLOCK"); bs = e ? atoi(e) : 64; if (bs != 64 && bs != 128 && bs != 256) bs = 64; }
	return bs;
}

static inline void gd_launch_wave64(const KswTask *tasks, const int32_t *ids, int n, const uint8_t *q, const uint8_t *t,
                                    uint8_t *bt, int32_t *status, int32_t *score, KswConst C, hipStream_t s, int tag = 0, bool single = false,
                                    int32_t *n_cigar = nullptr, uint32_t *cigar = nullptr /* both given: fused backtrack */,
                                    int waves_per_simd = 5 /* 4: see gdiet_hip_set_dp_waves */)
{
	WaveK K;
	gdw_make_consts(C, K);
	// one wavefront per workgroup: a finished wavefront frees its slot at once instead of waiting for its three block mates
	const int bs = gd_wave64_block();
	const dim3 grid((n + bs / 64 - 1) / (bs / 64)), block(bs);
	// Four wavefronts per SIMD instead of five: the kernel uses no LDS, so an (unused) dynamic allocation of a sixteenth of the CU's 160 KB
	// per wavefront caps the CU at 16 of them; the fifth wavefront's 96 registers per SIMD then stay free for other kernels.
	const size_t lds = waves_per_simd == 4 ? (size_t)(160 * 1024 / 16) * (bs / 64) : 0;
	if (single) hipLaunchKernelGGL((ksw_extd2_wave_kernel<64, 0, false>), grid, block, lds, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
	else if (tag == 1) hipLaunchKernelGGL((ksw_extd2_wave_kernel<64, 1>), grid, block, lds, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
	else if (tag == 2) hipLaunchKernelGGL((ksw_extd2_wave_kernel<64, 2>), grid, block, lds, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
	else hipLaunchKernelGGL((ksw_extd2_wave_kernel<64, 0>), grid, block, lds, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
}
// ids: 64 / G task ids per wavefront (identical geometry; -1 pads an incomplete group), n_groups wavefronts; G = 16, 10 or 8
template <int G> static inline void gd_launch_wave_groups(const KswTask *tasks, const int32_t *ids, int n_groups, const uint8_t *q, const uint8_t *t,
                                                          uint8_t *bt, int32_t *status, int32_t *score, KswConst C, hipStream_t s, bool single = false,
                                                          int32_t *n_cigar = nullptr, uint32_t *cigar = nullptr /* both given: fused backtrack */)
{
	WaveK K;
	gdw_make_consts(C, K);
	if (n_groups <= 0) return;
	if (single) hipLaunchKernelGGL((ksw_extd2_wave_kernel<G, 0, false>), dim3((n_groups + 3) / 4), dim3(256), 0, s, tasks, ids, n_groups, q, t, bt, status, score, K, n_cigar, cigar);
	else hipLaunchKernelGGL((ksw_extd2_wave_kernel<G, 0>), dim3((n_groups + 3) / 4), dim3(256), 0, s, tasks, ids, n_groups, q, t, bt, status, score, K, n_cigar, cigar);
}

// ---- wide bands (ONT, w = 1300): 128 blocks in flight, TWO per lane ------------------------------------------------------
// Block position p = blk mod 128 lives in lane p >> 1, sub-block p & 1, so a lane owns 32 consecutive cells.  The t-1
// neighbour of sub-block 1 is sub-block 0 of the same lane (plain register copies taken before the row is updated), the
// neighbour of sub-block 0 is sub-block 1 of the previous lane (DPP wave_ror:1).  Everything else is the 64-lane kernel
// applied to each sub-block.
struct Wave128State {
	WaveLane L0, L1;
	bool any_tn;
	int prev_st_, prev_st0, prev_up, prev_en0, have_f, Rf;
};

__device__ __forceinline__ void gdw128_init(Wave128State &S, const WaveK &K, int lane, const uint8_t *query, int qlen, const uint8_t *target, int tlen)
{
	gdw_load_block(S.L0, K, 2 * lane, 0, query, qlen, target, tlen);
	gdw_load_block(S.L1, K, 2 * lane + 1, 0, query, qlen, target, tlen);
	S.any_tn = __builtin_amdgcn_ballot_w64((S.L0.tn | S.L1.tn) != 0) != 0;
	S.prev_st_ = 0, S.prev_st0 = -1, S.prev_up = -1, S.prev_en0 = -1, S.have_f = 0, S.Rf = 0;
}

// one anti-diagonal.  pr: where this row's backtrace goes (the reference's own n_col_ blocks, block b at (b - st_) * 16,
// SR/ksw2.h:142: 1.5x less than a 128-block ring); STORE = false: the row is computed for its state and score only (first pass of
// the checkpointed form: the flag / direction bytes are dead code then)
template <bool STORE>
__device__ __forceinline__ void gdw128_row(Wave128State &S, const WaveK &K, int r, int qlen, int tlen, int w, const uint8_t *query, const uint8_t *target,
                                           uint8_t *pr, int mlast, int sl)
{
	constexpr int NBLK = 128;
	WaveLane &L0 = S.L0, &L1 = S.L1;
	WaveRow W;
	W.r = r;
	gdw_band_uniform(r, qlen, tlen, w, W.st0, W.en0);
	W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
	W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
	const int advanced = W.st_ > S.prev_st_;
	W.use_array = advanced;
	W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open;
	W.set_tr = (W.en0 | 15) >= r;
	W.ukey = gdw_edge_key(K, r);
	// (1) row r-1 values of the block below each sub-block, taken before anything is touched
	const u32 pX0 = gdw_ror1<64>(L1.X[7]), pV0 = gdw_ror1<64>(L1.V[7]), pX20 = gdw_ror1<64>(L1.X2[7]), pQ0 = gdw_ror1<64>(L1.Qc[3]);
	const u32 pX1 = L0.X[7], pV1 = L0.V[7], pX21 = L0.X2[7], pQ1 = L0.Qc[3];
	// (2) query window advance; a sub-block that fell below the window takes over block +128
	if (r > 0) {
		const u32 seam = gdw_seam_byte(query, qlen, r - (S.prev_st_ << 4));
		gdw_shift_query(L0, pQ0, L0.blk == S.prev_st_, seam);
		gdw_shift_query(L1, pQ1, L1.blk == S.prev_st_, seam);
	}
	if (advanced) {
		if (L0.blk < W.st_) gdw_load_block(L0, K, L0.blk + NBLK, r, query, qlen, target, tlen);
		if (L1.blk < W.st_) gdw_load_block(L1, K, L1.blk + NBLK, r, query, qlen, target, tlen);
		S.any_tn = __builtin_amdgcn_ballot_w64((L0.tn | L1.tn) != 0) != 0;
	}
	// (3) scalar fix-ups and the score row
	if (W.set_tr) gdw_reset_tr(L0, K, W), gdw_reset_tr(L1, K, W);
	if (W.st0 != S.prev_st0 || W.up != S.prev_up || advanced) gdw_make_sel(L0, W.st0, W.up), gdw_make_sel(L1, W.st0, W.up);
	gdw_update_scores(L0, K, S.any_tn);
	gdw_update_scores(L1, K, S.any_tn);
	// (4) DP cells of the sub-blocks inside the reference's 16-aligned window
	if (L0.blk <= W.en_) {
		u32 out[4];
		gdw_compute(L0, K, W, pX0, pV0, pX20, out);
		if (STORE) *reinterpret_cast<uint4 *>(pr + ((L0.blk - W.st_) << 4)) = make_uint4(out[0], out[1], out[2], out[3]);
	}
	if (L1.blk <= W.en_) {
		u32 out[4];
		gdw_compute(L1, K, W, pX1, pV1, pX21, out);
		if (STORE) *reinterpret_cast<uint4 *>(pr + ((L1.blk - W.st_) << 4)) = make_uint4(out[0], out[1], out[2], out[3]);
	}
	// (5) score trackers
	if (r == 0) L0.R = gdw_lo(L0.V[0]) - K.B1 - K.qe8, L1.R = gdw_lo(L1.V[0]) - K.B1 - K.qe8;
	else L0.R += gdw_lo(L0.V[0]) - K.B1, L1.R += gdw_lo(L1.V[0]) - K.B1;
	if (r > 0 && W.en0 != S.prev_en0 && (W.en0 & 15) == 0) {
		const int h0 = (int)gdw_ror1<64>((u32)gdw_track_handoff(L1)), h1 = gdw_track_handoff(L0);
		if (L0.blk == W.en_) L0.R = h0 + gdw_lo(L0.U[0]);
		if (L1.blk == W.en_) L1.R = h1 + gdw_lo(L1.U[0]);
	}
	if (W.en0 == tlen - 1) {
		if (L0.blk == mlast) {
			if (!S.have_f) S.Rf = gdw_track_to_slot(L0, sl);
			else S.Rf += gdw_cell(L0.V, sl) - K.B1;
		}
		if (L1.blk == mlast) {
			if (!S.have_f) S.Rf = gdw_track_to_slot(L1, sl);
			else S.Rf += gdw_cell(L1.V, sl) - K.B1;
		}
		S.have_f = 1;
	}
	S.prev_st_ = W.st_, S.prev_st0 = W.st0, S.prev_up = W.up, S.prev_en0 = W.en0;
}

__global__ __launch_bounds__(128) void ksw_extd2_wave128_kernel(const KswTask *__restrict__ tasks,
                                                                const int32_t *__restrict__ task_ids, int n_tasks,
                                                                const uint8_t *__restrict__ qseq,
                                                                const uint8_t *__restrict__ tseq,
                                                                uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                                int32_t *__restrict__ score_out, WaveK K,
                                                                int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar)
{
	const int lane = threadIdx.x & 63;
	const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
	if (slot >= n_tasks) return;
	const int tid = __builtin_amdgcn_readfirstlane(task_ids[slot]);
	if (__builtin_amdgcn_readfirstlane(status[tid]) != GD_ST_PENDING) return;
	const KswTask *Tp = tasks + tid;
	const int qlen = __builtin_amdgcn_readfirstlane(Tp->qlen), tlen = __builtin_amdgcn_readfirstlane(Tp->tlen);
	int w = __builtin_amdgcn_readfirstlane(Tp->w);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const uint8_t *query = qseq + Tp->qoff, *target = tseq + Tp->toff;
	uint8_t *p = bt + Tp->bt_off;
	const size_t row_bytes = (size_t)__builtin_amdgcn_readfirstlane(Tp->row_bytes);
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;

	Wave128State S;
	gdw128_init(S, K, lane, query, qlen, target, tlen);
	for (int r = 0; r <= rend; ++r) gdw128_row<true>(S, K, r, qlen, tlen, w, query, target, p + (size_t)r * row_bytes, mlast, sl);
	const bool fuse = cigar != nullptr; // walk the alignment back right away (see ksw_extd2_wave_kernel)
	if (S.L0.blk == mlast || S.L1.blk == mlast) {
		score_out[tid] = S.Rf >> 3;
		status[tid] = fuse ? GD_ST_TRACED : GD_ST_DONE;
	}
	if (fuse) {
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
		gd_bt_wave_walk(*Tp, tid, bt, n_cigar, cigar, lane);
	}
}

// ---- the same with a CHECKPOINTED backtrace ------------------------------------------------------------------------------------
// A 50 kbp ONT alignment writes 134 MB of backtrace, so only ~1 500 fit the HBM at once -- fewer than the GPU has wavefront
// slots, and the rate is bound by how many are in flight.  Here the first pass stores no backtrace at all, only a snapshot of the
// lanes' registers every GD_CK_ROWS anti-diagonals (35 KB each); the second pass walks the alignment back chunk by chunk from the
// end: restore the snapshot of the chunk, recompute its rows (the cone of the walk only, below) into a GD_CK_ROWS-row buffer (1 MB), let the walk consume them, go
// on with the chunk below.  ~1.85x the arithmetic (the first pass drops the flag / direction bytes), 4.9 MB instead of 134 MB per
// alignment: thousands in flight.  Rows, scores and CIGARs are those of the kernel above (same row function, same walk).
#define GD_CK_ROWS 480  // the cells a walk can visit inside one chunk, and all they depend on, span at most 62 HALF blocks (gdw_cone_row_half; <= 31 blocks for gdw_cone_row)
#define GD_CK_REGS 136 // dwords per lane per snapshot: 2 x (48 state + 16 Sb/Tb/Qc/SEL + tn, blk, R) + Rf + pad
static inline __host__ __device__ size_t gd_ck_bytes(int qlen, int tlen, int /*row_bytes*/)
{
	const size_t rows = (size_t)qlen + tlen - 1, n_ck = (rows + GD_CK_ROWS - 1) / GD_CK_ROWS;
	return n_ck * (size_t)GD_CK_REGS * 64 * 4 + (size_t)GD_CK_ROWS * 1024; // snapshots + one chunk of 64-block rows
}

__device__ __forceinline__ void gdw_lane_save(const WaveLane &L, u32 *d) // d: this lane's column of the snapshot, stride 64 dwords
{
#pragma unroll
	for (int k = 0; k < 8; ++k) d[(k) * 64] = L.U[k], d[(8 + k) * 64] = L.V[k], d[(16 + k) * 64] = L.X[k], d[(24 + k) * 64] = L.Y[k], d[(32 + k) * 64] = L.X2[k], d[(40 + k) * 64] = L.Y2[k];
#pragma unroll
	for (int g = 0; g < 4; ++g) d[(48 + g) * 64] = L.Sb[g], d[(52 + g) * 64] = L.Tb[g], d[(56 + g) * 64] = L.Qc[g], d[(60 + g) * 64] = L.SEL[g];
	d[64 * 64] = (L.Tb[0] | L.Tb[1] | L.Tb[2] | L.Tb[3]) & 0x04040404u /* = L.tn */, d[65 * 64] = (u32)L.blk, d[66 * 64] = (u32)L.R;
}
__device__ __forceinline__ void gdw_lane_load(WaveLane &L, const u32 *d)
{
#pragma unroll
	for (int k = 0; k < 8; ++k) L.U[k] = d[(k) * 64], L.V[k] = d[(8 + k) * 64], L.X[k] = d[(16 + k) * 64], L.Y[k] = d[(24 + k) * 64], L.X2[k] = d[(32 + k) * 64], L.Y2[k] = d[(40 + k) * 64];
#pragma unroll
	for (int g = 0; g < 4; ++g) L.Sb[g] = d[(48 + g) * 64], L.Tb[g] = d[(52 + g) * 64], L.Qc[g] = d[(56 + g) * 64], L.SEL[g] = d[(60 + g) * 64];
	L.tn = d[64 * 64], L.blk = (int32_t)d[65 * 64], L.R = (int32_t)d[66 * 64];
}

// ---- second pass of the checkpointed form: only the CONE of the walk is recomputed ---------------------------------------------
// A walk that enters a chunk at cell (i1, r1) moves down by at most one target position per anti-diagonal, so inside the chunk it
// stays within t in [i1 - (r1 - r), i1] on row r; and a cell (r, t) depends on (r-1, t-1) and (r-1, t) only, so those cells depend
// on nothing outside that same cone.  With at most GD_CK_ROWS = 480 rows the cone spans <= 31 blocks (62 half blocks): the second pass recomputes
// the 64 blocks [b0, b0 + 63], b0 = (i1 >> 4) - 63, ONE per lane and without a ring (half the instructions of the two-blocks-per-
// lane row), restored from the snapshot of the chunk.  Cells of those blocks below the cone come out wrong (their t-1 inputs are
// not computed) and are never read: wrongness spreads upwards by one cell per row, exactly as fast as the cone's lower edge.
__device__ __forceinline__ void gdw_cone_restore(WaveLane &L, const WaveK &K, const u32 *ck_chunk /* snapshot of the chunk, lane 0 */, int blk, int r0,
                                                 const uint8_t *query, int qlen, const uint8_t *target, int tlen)
{
	bool have = false;
	if (blk >= 0) { // ring position blk mod 128 of the first pass: lane (p >> 1), sub-block (p & 1)
		const int pos = blk & 127;
		gdw_lane_load(L, ck_chunk + (pos >> 1) + (pos & 1) * 67 * 64);
		L.tn = (L.Tb[0] | L.Tb[1] | L.Tb[2] | L.Tb[3]) & 0x04040404u; // (recomputed: the two halves of a block of the 96-block ring do not know each other's)
		have = L.blk == blk; // (else that position held another block at the time: this one had retired, or had not entered the ring)
	}
	if (!have) gdw_fresh_block(L, K, blk, r0 > 0 ? r0 - 1 : 0, query, qlen, target, tlen); // (query bytes of row r0 - 1: the row loop shifts first)
}

// one anti-diagonal of the cone: lane l holds block b0 + l throughout; pr: this row of the chunk buffer (64 blocks of 16 bytes)
__device__ __forceinline__ void gdw_cone_row(WaveLane &L, const WaveK &K, bool any_tn, int r, int qlen, int tlen, int w, const uint8_t *query, int b0, int lane,
                                             int &prev_st_, int &prev_st0, int &prev_up, uint8_t *pr)
{
	WaveRow W;
	W.r = r;
	gdw_band_uniform(r, qlen, tlen, w, W.st0, W.en0);
	W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
	W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
	const int advanced = W.st_ > prev_st_;
	W.use_array = advanced;
	W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open;
	W.set_tr = (W.en0 | 15) >= r;
	W.ukey = gdw_edge_key(K, r);
	const u32 pX = gdw_ror1<64>(L.X[7]), pV = gdw_ror1<64>(L.V[7]), pX2 = gdw_ror1<64>(L.X2[7]), pQ = gdw_ror1<64>(L.Qc[3]);
	// the query streams through the lanes as in the ring; lane 0 takes the byte its cell 0 faces from memory (what block b0 - 1 would hand up)
	if (r > 0) gdw_shift_query(L, pQ, lane == 0, gdw_seam_byte(query, qlen, r - (b0 << 4)));
	if (W.set_tr) gdw_reset_tr(L, K, W);
	if (W.st0 != prev_st0 || W.up != prev_up || advanced) gdw_make_sel(L, W.st0, W.up);
	gdw_update_scores(L, K, any_tn);
	if (L.blk >= W.st_ && L.blk <= W.en_) {
		u32 out[4];
		gdw_compute<true>(L, K, W, pX, pV, pX2, out);
		*reinterpret_cast<uint4 *>(pr + (lane << 4)) = make_uint4(out[0], out[1], out[2], out[3]);
	}
	prev_st_ = W.st_, prev_st0 = W.st0, prev_up = W.up;
}

// the same with one HALF block per lane: a chunk of 480 anti-diagonals has a cone of at most 481 cells = 62 half blocks, so the 64
// lanes hold the half blocks [hb0, hb0 + 63], hb0 = (i1 >> 3) - 63 -- four packed registers per state array instead of eight, half
// the instructions of gdw_cone_row again.  Rows of the chunk buffer: 64 x 8 bytes.
__device__ __forceinline__ void gdw_cone_restore_half(WaveHalf &H, const WaveK &K, const u32 *ck_chunk, int hidx, int r0,
                                                      const uint8_t *query, int qlen, const uint8_t *target, int tlen)
{
	const int blk = hidx >> 1, half = hidx & 1;
	bool have = false;
	if (blk >= 0) {
		const int pos = blk & 127;
		const u32 *d = ck_chunk + (pos >> 1) + (pos & 1) * 67 * 64;
		if ((int32_t)d[65 * 64] == blk) {
			WaveLane L;
			gdw_lane_load(L, d);
			gdw_half_from_lane(L, half, H);
			have = true;
		}
	}
	if (!have) gdw_fresh_half(H, K, blk, half, r0 > 0 ? r0 - 1 : 0, query, qlen, target, tlen);
}

__device__ __forceinline__ void gdw_cone_row_half(WaveHalf &H, const WaveK &K, bool any_tn, int r, int qlen, int tlen, int w, const uint8_t *query, int hb0, int lane,
                                                  int &prev_st_, int &prev_st0, int &prev_up, uint8_t *pr)
{
	WaveRow W;
	W.r = r;
	gdw_band_uniform(r, qlen, tlen, w, W.st0, W.en0);
	W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
	W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
	const int advanced = W.st_ > prev_st_;
	W.use_array = advanced;
	W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open;
	W.set_tr = (W.en0 | 15) >= r;
	W.ukey = gdw_edge_key(K, r);
	const u32 pX = gdw_ror1<64>(H.X[3]), pV = gdw_ror1<64>(H.V[3]), pX2 = gdw_ror1<64>(H.X2[3]), pQ = gdw_ror1<64>(H.Qc[1]);
	if (r > 0) gdw_shift_query_half(H, pQ, lane == 0, gdw_seam_byte(query, qlen, r - (hb0 << 3)));
	if (W.set_tr) gdw_reset_tr_half(H, K, W);
	if (W.st0 != prev_st0 || W.up != prev_up || advanced) gdw_make_sel_half(H, W.st0, W.up);
	gdw_update_scores_half(H, K, any_tn);
	if (H.blk >= W.st_ && H.blk <= W.en_) {
		u32 out[2];
		gdw_compute_half<true>(H, K, W, pX, pV, pX2, out);
		*reinterpret_cast<uint2 *>(pr + (lane << 3)) = make_uint2(out[0], out[1]);
	}
	prev_st_ = W.st_, prev_st0 = W.st0, prev_up = W.up;
}

// one anti-diagonal of an interior chunk (H.SEL = "all eight scores are fresh", set by the caller)
__device__ __forceinline__ void gdw_cone_row_half_fast(WaveHalf &H, const WaveK &K, bool any_tn, int r, int qlen, const uint8_t *query, int hb0, int lane, uint8_t *pr)
{
	WaveRow W;
	W.r = r, W.use_array = 1; // (no lane holds the first block of the window: nothing else of W is read)
	const u32 pX = gdw_ror1<64>(H.X[3]), pV = gdw_ror1<64>(H.V[3]), pX2 = gdw_ror1<64>(H.X2[3]), pQ = gdw_ror1<64>(H.Qc[1]);
	gdw_shift_query_half(H, pQ, lane == 0, gdw_seam_byte(query, qlen, r - (hb0 << 3)));
	gdw_update_scores_half(H, K, any_tn);
	u32 out[2];
	gdw_compute_half<true>(H, K, W, pX, pV, pX2, out);
	*reinterpret_cast<uint2 *>(pr + (lane << 3)) = make_uint2(out[0], out[1]);
}

__global__ __launch_bounds__(128) void ksw_extd2_wave128c_kernel(const KswTask *__restrict__ tasks,
                                                                 const int32_t *__restrict__ task_ids, int n_tasks,
                                                                 const uint8_t *__restrict__ qseq,
                                                                 const uint8_t *__restrict__ tseq,
                                                                 uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                                 int32_t *__restrict__ score_out, WaveK K,
                                                                 int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar)
{
	const int lane = threadIdx.x & 63;
	const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
	if (slot >= n_tasks) return;
	const int tid = __builtin_amdgcn_readfirstlane(task_ids[slot]);
	if (__builtin_amdgcn_readfirstlane(status[tid]) != GD_ST_PENDING) return;
	const KswTask *Tp = tasks + tid;
	const int qlen = __builtin_amdgcn_readfirstlane(Tp->qlen), tlen = __builtin_amdgcn_readfirstlane(Tp->tlen);
	int w = __builtin_amdgcn_readfirstlane(Tp->w);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const uint8_t *query = qseq + Tp->qoff, *target = tseq + Tp->toff;
	const size_t row_bytes = (size_t)__builtin_amdgcn_readfirstlane(Tp->row_bytes);
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;
	const int n_ck = (rend + GD_CK_ROWS) / GD_CK_ROWS; // chunks of GD_CK_ROWS anti-diagonals
	u32 *ck = reinterpret_cast<u32 *>(bt + Tp->bt_off);                       // [n_ck][GD_CK_REGS][64]
	uint8_t *chunk = bt + Tp->bt_off + (size_t)n_ck * GD_CK_REGS * 64 * 4;    // [GD_CK_ROWS][row_bytes]

	// ---- pass 1: state and score, a snapshot at the start of every chunk ----
	Wave128State S;
	gdw128_init(S, K, lane, query, qlen, target, tlen);
	int ck_row = 0, ck_idx = 0;
	for (int r = 0; r <= rend; ++r) {
		if (r == ck_row) { // (GD_CK_ROWS is not a power of two: a counter instead of a division per row)
			u32 *d = ck + (size_t)ck_idx * GD_CK_REGS * 64 + lane;
			ck_row += GD_CK_ROWS, ++ck_idx;
			gdw_lane_save(S.L0, d), gdw_lane_save(S.L1, d + 67 * 64);
			d[134 * 64] = (u32)S.Rf;
		}
		gdw128_row<false>(S, K, r, qlen, tlen, w, query, target, nullptr, mlast, sl);
	}
	if (S.L0.blk == mlast || S.L1.blk == mlast) {
		score_out[tid] = S.Rf >> 3;
		status[tid] = GD_ST_TRACED;
	}
	// ---- pass 2: from the last chunk down, recompute the cone of the walk inside the chunk and let the walk consume it ----
	GdWalk Wk;
	gd_walk_init(Wk, qlen, tlen);
	WaveLane L;
	for (int kk = n_ck - 1; kk >= 0 && Wk.i >= 0 && Wk.j >= 0; --kk) {
		const int k = __builtin_amdgcn_readfirstlane(kk);
		const int i1 = __builtin_amdgcn_readfirstlane(Wk.i), rtop = i1 + __builtin_amdgcn_readfirstlane(Wk.j); // where the walk stands
		const int r0 = k * GD_CK_ROWS;
		if (rtop < r0) continue; // (cannot happen: a step moves at most two anti-diagonals down; kept for safety)
		const int r1 = rtop < r0 + GD_CK_ROWS - 1 ? rtop : r0 + GD_CK_ROWS - 1; // rows above the walk are not needed
		const int b0 = (i1 >> 4) - 63;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); // the snapshot stores of pass 1 / the previous walk's reads of `chunk`
		gdw_cone_restore(L, K, ck + (size_t)k * GD_CK_REGS * 64, b0 + lane, r0, query, qlen, target, tlen);
		const bool any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0;
		int prev_st_ = 0, prev_st0 = -1, prev_up = -1;
		if (r0 > 0) { // the loop-carried band quantities of anti-diagonal r0 - 1
			int st0, en0;
			gdw_band_uniform(r0 - 1, qlen, tlen, w, st0, en0);
			prev_st_ = st0 >> 4, prev_st0 = st0, prev_up = st0 + (((en0 - st0 + 16) >> 4) << 4);
		}
		for (int r = r0; r <= r1; ++r) gdw_cone_row(L, K, any_tn, r, qlen, tlen, w, query, b0, lane, prev_st_, prev_st0, prev_up, chunk + (size_t)(r - r0) * 1024);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); // this wavefront's own stores: complete, and not served from a stale L1 line
		gd_walk_rows(Wk, *Tp, chunk, r0, qlen, tlen, w, cigar, lane, 1024, b0);
	}
	gd_walk_finish(Wk, *Tp, tid, n_cigar, cigar, lane);
}

// ---- the checkpointed form on a 96-block ring: one block + one half block per lane (ksw_wave_core.h, "half blocks") ------------
// First pass only differs: lane l holds block position l (F) and half (l & 1) of block position 64 + (l >> 1) (H); the ring order is
// F(0) .. F(63), H(0) .. H(63), back to F(0), so the row r-1 values of "the (half) block below" are one wave_ror:1 of F's and one of
// H's registers, swapped in lane 0.  Snapshots are written in the 128-position record format of the kernel above (a half block
// stores its 16-bit halves of the eight registers of each array), so the second pass -- gdw_cone_restore / gdw_cone_row / the walk
// -- is the very same code.  12 packed registers per state array instead of 16: -25 % instructions in the first pass.
struct Wave96State {
	WaveLane F;
	WaveHalf H;
	bool any_tn;
	int prev_st_, prev_st0, prev_up, prev_en0, have_f, Rf;
};

__device__ __forceinline__ void gdw96_row(Wave96State &S, const WaveK &K, int r, int qlen, int tlen, int w, const uint8_t *query, const uint8_t *target, int lane,
                                          int mlast, int sl)
{
	constexpr int NBLK = 96;
	WaveLane &F = S.F;
	WaveHalf &H = S.H;
	WaveRow W;
	W.r = r;
	gdw_band_uniform(r, qlen, tlen, w, W.st0, W.en0);
	W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
	W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
	const int advanced = W.st_ > S.prev_st_;
	W.use_array = advanced;
	W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open;
	W.set_tr = (W.en0 | 15) >= r;
	W.ukey = gdw_edge_key(K, r);
	// (1) row r-1 values of the (half) block below, taken before anything is touched
	const bool l0 = lane == 0;
	const u32 aX = gdw_ror1<64>(F.X[7]), aV = gdw_ror1<64>(F.V[7]), aX2 = gdw_ror1<64>(F.X2[7]), aQ = gdw_ror1<64>(F.Qc[3]);
	const u32 bX = gdw_ror1<64>(H.X[3]), bV = gdw_ror1<64>(H.V[3]), bX2 = gdw_ror1<64>(H.X2[3]), bQ = gdw_ror1<64>(H.Qc[1]);
	const u32 fX = l0 ? bX : aX, fV = l0 ? bV : aV, fX2 = l0 ? bX2 : aX2, fQ = l0 ? bQ : aQ;
	const u32 hX = l0 ? aX : bX, hV = l0 ? aV : bV, hX2 = l0 ? aX2 : bX2, hQ = l0 ? aQ : bQ;
	// (2) query window advance; what fell below the window takes over block +96
	if (r > 0) {
		const u32 seam = gdw_seam_byte(query, qlen, r - (S.prev_st_ << 4));
		gdw_shift_query(F, fQ, F.blk == S.prev_st_, seam);
		gdw_shift_query_half(H, hQ, H.blk == S.prev_st_ && H.half == 0, seam);
	}
	if (advanced) {
		if (F.blk < W.st_) gdw_load_block(F, K, F.blk + NBLK, r, query, qlen, target, tlen);
		if (H.blk < W.st_) gdw_load_half(H, K, H.blk + NBLK, H.half, r, query, qlen, target, tlen);
		S.any_tn = __builtin_amdgcn_ballot_w64(((F.Tb[0] | F.Tb[1] | F.Tb[2] | F.Tb[3] | H.Tb[0] | H.Tb[1]) & 0x04040404u) != 0) != 0; // (from the bytes: no tn registers in this pass)
	}
	// (3) scalar fix-ups and the score row
	if (W.set_tr) gdw_reset_tr(F, K, W), gdw_reset_tr_half(H, K, W);
	if (W.st0 != S.prev_st0 || W.up != S.prev_up || advanced) gdw_make_sel(F, W.st0, W.up), gdw_make_sel_half(H, W.st0, W.up);
	gdw_update_scores(F, K, S.any_tn);
	gdw_update_scores_half(H, K, S.any_tn);
	// (4) DP cells of what lies inside the reference's 16-aligned window
	if (F.blk <= W.en_) {
		u32 out[4];
		gdw_compute(F, K, W, fX, fV, fX2, out); // (the flag / direction bytes are dead code here)
	}
	if (H.blk <= W.en_) gdw_compute_half(H, K, W, hX, hV, hX2);
	// (5) score trackers
	if (r == 0) F.R = gdw_lo(F.V[0]) - K.B1 - K.qe8, H.R = gdw_lo(H.V[0]) - K.B1 - K.qe8;
	else F.R += gdw_lo(F.V[0]) - K.B1, H.R += gdw_lo(H.V[0]) - K.B1;
	if (r > 0 && W.en0 != S.prev_en0 && (W.en0 & 7) == 0) { // a block, or the second half of one, enters the band
		const int ha = (int)gdw_ror1<64>((u32)gdw_track_handoff(F)), hb = (int)gdw_ror1<64>((u32)gdw_track_handoff_half(H));
		const int hf = l0 ? hb : ha, hh = l0 ? ha : hb;
		if ((W.en0 & 15) == 0) {
			if (F.blk == W.en_) F.R = hf + gdw_lo(F.U[0]);
			if (H.blk == W.en_ && H.half == 0) H.R = hh + gdw_lo(H.U[0]);
		} else if (H.blk == W.en_ && H.half == 1) H.R = hh + gdw_lo(H.U[0]);
	}
	if (W.en0 == tlen - 1) {
		if (F.blk == mlast) {
			if (!S.have_f) S.Rf = gdw_track_to_slot(F, sl);
			else S.Rf += gdw_cell(F.V, sl) - K.B1;
		}
		if (H.blk == mlast && H.half == (sl >> 3)) {
			if (!S.have_f) S.Rf = gdw_track_to_slot_half(H, sl & 7);
			else S.Rf += gdw_cell_half(H.V, sl & 7) - K.B1;
		}
		S.have_f = 1;
	}
	S.prev_st_ = W.st_, S.prev_st0 = W.st0, S.prev_up = W.up, S.prev_en0 = W.en0;
}

// Rows [rA, rS) of a long alignment (gdw_steady_rows) as pairs, as in ksw_extd2_wave_kernel: on the first row of a pair st0 has just
// moved up by one cell, on the second en0 has; both follow from m = (r - w + 1) >> 1 without min / max, neither the first block, the
// cell t == r nor the last target column is involved, a block can retire only on the first row and enter the band only on the
// second.  m_lowF / m_lowH: 0 / ~0 lane masks "holds the lowest block (its first half)" of the row just done.
template <bool ROW_A>
__device__ __forceinline__ void gdw96_pair_row(Wave96State &S, const WaveK &K, int r, int m, int w, int nblkA, int nblkB, const uint8_t *query, int qlen,
                                               const uint8_t *target, int tlen, int lane, u32 &m_lowF, u32 &m_lowH, u32 *lds /* this lane's 12 dwords: Tb, SEL of F and H */)
{
	constexpr int NBLK = 96;
	WaveLane &F = S.F;
	WaveHalf &H = S.H;
	WaveRow W;
	W.r = r, W.st0 = m, W.en0 = ROW_A ? m + w - 1 : m + w;
	W.st_ = m >> 4, W.en_ = W.en0 >> 4;
	W.up = m + ((ROW_A ? nblkA : nblkB) << 4);
	const int advanced = ROW_A && (m & 15) == 0;
	const int pst_ = W.st_ - advanced; // st_ of the row before
	W.use_array = advanced, W.v1key = K.key_open, W.set_tr = 0, W.ukey = 0;
	const bool l0 = lane == 0;
	const u32 aX = gdw_ror1<64>(F.X[7]), aV = gdw_ror1<64>(F.V[7]), aX2 = gdw_ror1<64>(F.X2[7]), aQ = gdw_ror1<64>(F.Qc[3]);
	const u32 bX = gdw_ror1<64>(H.X[3]), bV = gdw_ror1<64>(H.V[3]), bX2 = gdw_ror1<64>(H.X2[3]), bQ = gdw_ror1<64>(H.Qc[1]);
	const u32 fX = l0 ? bX : aX, fV = l0 ? bV : aV, fX2 = l0 ? bX2 : aX2, fQ = l0 ? bQ : aQ;
	const u32 hX = l0 ? aX : bX, hV = l0 ? aV : bV, hX2 = l0 ? aX2 : bX2, hQ = l0 ? aQ : bQ;
	// the target bytes and the score selectors of the lane live in LDS between the rows (read once per row, rewritten every 16 / 2
	// rows): twelve registers the row loop needs for the pair form's lane masks and its temporaries (the kernel sits at the 128-register limit)
	{
		const uint4 t4 = *reinterpret_cast<const uint4 *>(lds), s4 = *reinterpret_cast<const uint4 *>(lds + 4), h4 = *reinterpret_cast<const uint4 *>(lds + 8);
		F.Tb[0] = t4.x, F.Tb[1] = t4.y, F.Tb[2] = t4.z, F.Tb[3] = t4.w;
		F.SEL[0] = s4.x, F.SEL[1] = s4.y, F.SEL[2] = s4.z, F.SEL[3] = s4.w;
		H.Tb[0] = h4.x, H.Tb[1] = h4.y, H.SEL[0] = h4.z, H.SEL[1] = h4.w;
	}
	{
		const u32 seam = gdw_seam_byte(query, qlen, r - (pst_ << 4));
		gdw_shift_query_m(F, fQ, m_lowF, seam); // (the lane masks of the row before: its st_ is this row's pst_)
		gdw_shift_query_half_m(H, hQ, m_lowH, seam);
	}
	bool dirty = false;
	if (advanced) {
		if (F.blk < W.st_) gdw_load_block(F, K, F.blk + NBLK, r, query, qlen, target, tlen);
		if (H.blk < W.st_) gdw_load_half(H, K, H.blk + NBLK, H.half, r, query, qlen, target, tlen);
		S.any_tn = __builtin_amdgcn_ballot_w64(((F.Tb[0] | F.Tb[1] | F.Tb[2] | F.Tb[3] | H.Tb[0] | H.Tb[1]) & 0x04040404u) != 0) != 0;
		dirty = true;
	}
	if (ROW_A || nblkA != nblkB) {
		u32 lo[4], hi[4];
		gdw_sel_uniform(m & 15, lo, hi);
		gdw_pick_sel(F, W.st_, W.up >> 4, (W.up >> 4) - W.st_ >= NBLK - 1, lo, hi, m_lowF);
		gdw_make_sel_half(H, W.st0, W.up);
		m_lowH = (H.blk == W.st_ && H.half == 0) ? ~0u : 0u;
		dirty = true;
	}
	if (dirty) {
		*reinterpret_cast<uint4 *>(lds) = make_uint4(F.Tb[0], F.Tb[1], F.Tb[2], F.Tb[3]);
		*reinterpret_cast<uint4 *>(lds + 4) = make_uint4(F.SEL[0], F.SEL[1], F.SEL[2], F.SEL[3]);
		*reinterpret_cast<uint4 *>(lds + 8) = make_uint4(H.Tb[0], H.Tb[1], H.SEL[0], H.SEL[1]);
	}
	W.m_first_valid = 1, W.m_first = m_lowF, W.m_first_h = m_lowH;
	gdw_update_scores(F, K, S.any_tn);
	gdw_update_scores_half(H, K, S.any_tn);
	if (F.blk <= W.en_) {
		u32 out[4];
		gdw_compute(F, K, W, fX, fV, fX2, out);
	}
	if (H.blk <= W.en_) gdw_compute_half(H, K, W, hX, hV, hX2);
	F.R += gdw_lo(F.V[0]), H.R += gdw_lo(H.V[0]); // (the bias B1 of every V key is taken off once, after the loop)
	if (!ROW_A && (W.en0 & 7) == 0) { // a block, or the second half of one, enters the band
		const int ha = (int)gdw_ror1<64>((u32)gdw_track_handoff(F)), hb = (int)gdw_ror1<64>((u32)gdw_track_handoff_half(H));
		const int hf = l0 ? hb : ha, hh = l0 ? ha : hb;
		if ((W.en0 & 15) == 0) {
			if (F.blk == W.en_) F.R = hf + gdw_lo(F.U[0]);
			if (H.blk == W.en_ && H.half == 0) H.R = hh + gdw_lo(H.U[0]);
		} else if (H.blk == W.en_ && H.half == 1) H.R = hh + gdw_lo(H.U[0]);
	}
	S.prev_st_ = W.st_; // (a snapshot may follow)
}

// a snapshot in the record format gdw_cone_restore reads: the record of ring position p = blk mod 128 sits at column p >> 1, field
// offset (p & 1) * 67; a half block writes its 16-bit halves of the packed registers and its two dwords of the byte arrays
__device__ __forceinline__ void gdw96_save(const Wave96State &S, u32 *ck_chunk /* lane 0 */, int lane)
{
	{
		const int pos = S.F.blk & 127;
		gdw_lane_save(S.F, ck_chunk + (pos >> 1) + (pos & 1) * 67 * 64);
	}
	const WaveHalf &H = S.H;
	const int pos = H.blk & 127;
	u32 *d = ck_chunk + (pos >> 1) + (pos & 1) * 67 * 64;
	uint16_t *d16 = reinterpret_cast<uint16_t *>(d) + H.half; // low / high 16 bits of every dword
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		d16[(size_t)(k) * 128] = (uint16_t)gdw_half_cell16(H.U, k), d16[(size_t)(8 + k) * 128] = (uint16_t)gdw_half_cell16(H.V, k);
		d16[(size_t)(16 + k) * 128] = (uint16_t)gdw_half_cell16(H.X, k), d16[(size_t)(24 + k) * 128] = (uint16_t)gdw_half_cell16(H.Y, k);
		d16[(size_t)(32 + k) * 128] = (uint16_t)gdw_half_cell16(H.X2, k), d16[(size_t)(40 + k) * 128] = (uint16_t)gdw_half_cell16(H.Y2, k);
	}
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		const int gg = 2 * H.half + g;
		d[(48 + gg) * 64] = H.Sb[g], d[(52 + gg) * 64] = H.Tb[g], d[(56 + gg) * 64] = H.Qc[g], d[(60 + gg) * 64] = H.SEL[g];
	}
	d[65 * 64] = (u32)H.blk;
	if (H.half == 0) d[66 * 64] = (u32)H.R;
	// the 32 ring positions nobody holds: their records must not look like a block (gdw_cone_restore compares the block number)
	if (lane < 32) {
		const int top = __builtin_amdgcn_readfirstlane(S.prev_st_) + 96 + lane; // (S.prev_st_ <= every held block < S.prev_st_ + 96 at a snapshot)
		const int p2 = top & 127;
		ck_chunk[(p2 >> 1) + (p2 & 1) * 67 * 64 + 65 * 64] = 0xffffffffu;
	}
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4))) void ksw_extd2_wave96c_kernel(const KswTask *__restrict__ tasks,
                                                                const int32_t *__restrict__ task_ids, int n_tasks,
                                                                const uint8_t *__restrict__ qseq,
                                                                const uint8_t *__restrict__ tseq,
                                                                uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                                int32_t *__restrict__ score_out, WaveK K,
                                                                int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar)
{
	const int lane = threadIdx.x & 63;
	const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
	if (slot >= n_tasks) return;
	const int tid = __builtin_amdgcn_readfirstlane(task_ids[slot]);
	if (__builtin_amdgcn_readfirstlane(status[tid]) != GD_ST_PENDING) return;
	const KswTask *Tp = tasks + tid;
	const int qlen = __builtin_amdgcn_readfirstlane(Tp->qlen), tlen = __builtin_amdgcn_readfirstlane(Tp->tlen);
	int w = __builtin_amdgcn_readfirstlane(Tp->w);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const uint8_t *query = qseq + Tp->qoff, *target = tseq + Tp->toff;
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;
	const int n_ck = (rend + GD_CK_ROWS) / GD_CK_ROWS;
	u32 *ck = reinterpret_cast<u32 *>(bt + Tp->bt_off);
	uint8_t *chunk = bt + Tp->bt_off + (size_t)n_ck * GD_CK_REGS * 64 * 4;

	// ---- pass 1: state and score on the 96-block ring, a snapshot at the start of every chunk ----
	{
		Wave96State S;
		gdw_load_block(S.F, K, lane, 0, query, qlen, target, tlen);
		gdw_load_half(S.H, K, 64 + (lane >> 1), lane & 1, 0, query, qlen, target, tlen);
		S.any_tn = __builtin_amdgcn_ballot_w64(((S.F.Tb[0] | S.F.Tb[1] | S.F.Tb[2] | S.F.Tb[3] | S.H.Tb[0] | S.H.Tb[1]) & 0x04040404u) != 0) != 0;
		S.prev_st_ = 0, S.prev_st0 = -1, S.prev_up = -1, S.prev_en0 = -1, S.have_f = 0, S.Rf = 0;
		int ck_row = 0, ck_idx = 0;
		__shared__ u32 lds_all[2 * 64 * 12]; // (two wavefronts per workgroup)
		u32 *lds = lds_all + ((threadIdx.x >> 6) * 64 + lane) * 12;
		auto to_lds = [&]() __attribute__((always_inline)) {
			*reinterpret_cast<uint4 *>(lds) = make_uint4(S.F.Tb[0], S.F.Tb[1], S.F.Tb[2], S.F.Tb[3]);
			*reinterpret_cast<uint4 *>(lds + 4) = make_uint4(S.F.SEL[0], S.F.SEL[1], S.F.SEL[2], S.F.SEL[3]);
			*reinterpret_cast<uint4 *>(lds + 8) = make_uint4(S.H.Tb[0], S.H.Tb[1], S.H.SEL[0], S.H.SEL[1]);
		};
		auto from_lds = [&]() __attribute__((always_inline)) {
			const uint4 t4 = *reinterpret_cast<const uint4 *>(lds), s4 = *reinterpret_cast<const uint4 *>(lds + 4), h4 = *reinterpret_cast<const uint4 *>(lds + 8);
			S.F.Tb[0] = t4.x, S.F.Tb[1] = t4.y, S.F.Tb[2] = t4.z, S.F.Tb[3] = t4.w;
			S.F.SEL[0] = s4.x, S.F.SEL[1] = s4.y, S.F.SEL[2] = s4.z, S.F.SEL[3] = s4.w;
			S.H.Tb[0] = h4.x, S.H.Tb[1] = h4.y, S.H.SEL[0] = h4.z, S.H.SEL[1] = h4.w;
		};
		auto snapshot = [&](const int r) __attribute__((always_inline)) {
			if (r == ck_row) {
				gdw96_save(S, ck + (size_t)ck_idx * GD_CK_REGS * 64, lane);
				ck_row += GD_CK_ROWS, ++ck_idx;
			}
		};
		int rA, rS, r = 0;
		gdw_steady_rows(qlen, tlen, w, rA, rS);
		for (; r <= rend && r < rA; ++r) {
			snapshot(r);
			gdw96_row(S, K, r, qlen, tlen, w, query, target, lane, mlast, sl);
		}
		if (r == rA && rS > rA) { // the rows in the middle, as pairs
			const int nblkA = (w - 1 + 16) >> 4, nblkB = (w + 16) >> 4;
			int m = (rA - w + 1) >> 1;
			u32 m_lowF = S.F.blk == S.prev_st_ ? ~0u : 0u, m_lowH = (S.H.blk == S.prev_st_ && S.H.half == 0) ? ~0u : 0u;
			to_lds();
			for (; r < rS; r += 2, ++m) {
				if (r == ck_row || r + 1 == ck_row) from_lds(); // (a snapshot stores the whole lane)
				snapshot(r);
				gdw96_pair_row<true>(S, K, r, m, w, nblkA, nblkB, query, qlen, target, tlen, lane, m_lowF, m_lowH, lds);
				snapshot(r + 1);
				gdw96_pair_row<false>(S, K, r + 1, m, w, nblkA, nblkB, query, qlen, target, tlen, lane, m_lowF, m_lowH, lds);
			}
			from_lds();
			S.F.R -= (rS - rA) * K.B1, S.H.R -= (rS - rA) * K.B1; // the trackers of the paired rows accumulated the V keys with their bias
			--m; // the band of the last row, for the rows that follow
			S.prev_st_ = m >> 4, S.prev_st0 = m, S.prev_up = m + (nblkB << 4), S.prev_en0 = m + w;
		}
		for (; r <= rend; ++r) {
			snapshot(r);
			gdw96_row(S, K, r, qlen, tlen, w, query, target, lane, mlast, sl);
		}
		if (S.F.blk == mlast || (S.H.blk == mlast && S.H.half == (sl >> 3))) {
			score_out[tid] = S.Rf >> 3;
			status[tid] = GD_ST_TRACED;
		}
	}
	// ---- pass 2: the cone of the walk with one half block per lane ----
	GdWalk Wk;
	gd_walk_init(Wk, qlen, tlen);
	WaveHalf Hc;
	for (int kk = n_ck - 1; kk >= 0 && Wk.i >= 0 && Wk.j >= 0; --kk) {
		const int k = __builtin_amdgcn_readfirstlane(kk);
		const int i1 = __builtin_amdgcn_readfirstlane(Wk.i), rtop = i1 + __builtin_amdgcn_readfirstlane(Wk.j);
		const int r0 = k * GD_CK_ROWS;
		if (rtop < r0) continue;
		const int r1 = rtop < r0 + GD_CK_ROWS - 1 ? rtop : r0 + GD_CK_ROWS - 1;
		const int hb0 = (i1 >> 3) - 63;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
		gdw_cone_restore_half(Hc, K, ck + (size_t)k * GD_CK_REGS * 64, hb0 + lane, r0, query, qlen, target, tlen);
		const bool any_tn = __builtin_amdgcn_ballot_w64(Hc.tn != 0) != 0;
		int prev_st_ = 0, prev_st0 = -1, prev_up = -1;
		if (r0 > 0) {
			int st0, en0;
			gdw_band_uniform(r0 - 1, qlen, tlen, w, st0, en0);
			prev_st_ = st0 >> 4, prev_st0 = st0, prev_up = st0 + (((en0 - st0 + 16) >> 4) << 4);
		}
		if (gdw_cone_interior_half(r0, r1, hb0, qlen, tlen, w)) { // (wave-uniform; almost every chunk of a read whose walk stays away from the band's edges)
			Hc.SEL[0] = Hc.SEL[1] = 0x07060504u;
			for (int r = r0; r <= r1; ++r) gdw_cone_row_half_fast(Hc, K, any_tn, r, qlen, query, hb0, lane, chunk + (size_t)(r - r0) * 512);
		} else
			for (int r = r0; r <= r1; ++r) gdw_cone_row_half(Hc, K, any_tn, r, qlen, tlen, w, query, hb0, lane, prev_st_, prev_st0, prev_up, chunk + (size_t)(r - r0) * 512);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
		gd_walk_rows(Wk, *Tp, chunk, r0, qlen, tlen, w, cigar, lane, 512, hb0, true);
	}
	gd_walk_finish(Wk, *Tp, tid, n_cigar, cigar, lane);
}

static inline void gd_launch_wave128(const KswTask *tasks, const int32_t *ids, int n, const uint8_t *q, const uint8_t *t,
                                     uint8_t *bt, int32_t *status, int32_t *score, KswConst C, hipStream_t s, int32_t *n_cigar = nullptr,
                                     uint32_t *cigar = nullptr, bool checkpointed = false /* needs n_cigar / cigar: it walks its own alignments back */)
{
	WaveK K;
	gdw_make_consts(C, K);
	if (checkpointed) hipLaunchKernelGGL(ksw_extd2_wave128c_kernel, dim3((n + 1) / 2), dim3(128), 0, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
	else hipLaunchKernelGGL(ksw_extd2_wave128_kernel, dim3((n + 1) / 2), dim3(128), 0, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
}

static inline void gd_launch_wave96c(const KswTask *tasks, const int32_t *ids, int n, const uint8_t *q, const uint8_t *t, uint8_t *bt, int32_t *status,
                                     int32_t *score, KswConst C, hipStream_t s, int32_t *n_cigar, uint32_t *cigar)
{
	WaveK K;
	gdw_make_consts(C, K);
	hipLaunchKernelGGL(ksw_extd2_wave96c_kernel, dim3((n + 1) / 2), dim3(128), 0, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
}

// ---- wide bands, few alignments: TWO wavefronts per alignment ----------------------------------------------------------------
// 50 kbp ONT alignments hold 134 MB of backtrace each, so only ~1 500 fit the arena at once: fewer than the GPU has wavefront
// slots, and the kernel time is the serial chain of the longest one.  Here a workgroup of two wavefronts shares one alignment:
// block position p = blk mod 128 lives in wavefront p >> 6, lane p & 63 (one block per lane, as in the 64-lane kernel), so the
// chain per anti-diagonal is half as long as in the two-blocks-per-lane form.  Inside a wavefront the t-1 neighbour comes through
// DPP as before; lane 0 takes it from lane 63 of the other wavefront through LDS (double-buffered by the parity of the row: one
// workgroup barrier per anti-diagonal).  Backtrace layout and results are those of ksw_extd2_wave128_kernel.
__global__ __launch_bounds__(128) void ksw_extd2_wave2x64_kernel(const KswTask *__restrict__ tasks,
                                                                 const int32_t *__restrict__ task_ids, int n_tasks,
                                                                 const uint8_t *__restrict__ qseq,
                                                                 const uint8_t *__restrict__ tseq,
                                                                 uint8_t *__restrict__ bt, int32_t *__restrict__ status,
                                                                 int32_t *__restrict__ score_out, WaveK K,
                                                                 int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar)
{
	constexpr int NBLK = 128;
	__shared__ u32 xch[2][2][8]; // [row parity][wavefront]: X[7], V[7], X2[7], Qc[3] of lane 63, and the tracker hand-off
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int slot = blockIdx.x;
	if (slot >= n_tasks) return;
	const int tid = __builtin_amdgcn_readfirstlane(task_ids[slot]);
	if (__builtin_amdgcn_readfirstlane(status[tid]) != GD_ST_PENDING) return;
	const KswTask *Tp = tasks + tid;
	const int qlen = __builtin_amdgcn_readfirstlane(Tp->qlen), tlen = __builtin_amdgcn_readfirstlane(Tp->tlen);
	int w = __builtin_amdgcn_readfirstlane(Tp->w);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const uint8_t *query = qseq + Tp->qoff, *target = tseq + Tp->toff;
	uint8_t *p = bt + Tp->bt_off;
	const size_t row_bytes = (size_t)__builtin_amdgcn_readfirstlane(Tp->row_bytes);
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;

	WaveLane L;
	gdw_load_block(L, K, wv * 64 + lane, 0, query, qlen, target, tlen);
	bool any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0;
	int prev_st_ = 0, prev_st0 = -1, prev_up = -1, prev_en0 = -1, have_f = 0, Rf = 0;
	for (int r = 0; r <= rend; ++r) {
		WaveRow W;
		W.r = r;
		gdw_band_uniform(r, qlen, tlen, w, W.st0, W.en0);
		W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
		W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
		const int advanced = W.st_ > prev_st_;
		W.use_array = advanced;
		W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open;
		W.set_tr = (W.en0 | 15) >= r;
		W.ukey = gdw_edge_key(K, r);
		// (1) row r-1 values of the block below: DPP inside the wavefront, LDS across the two
		u32 pX = gdw_ror1<64>(L.X[7]), pV = gdw_ror1<64>(L.V[7]), pX2 = gdw_ror1<64>(L.X2[7]), pQ = gdw_ror1<64>(L.Qc[3]);
		u32 *mine = xch[r & 1][wv];
		const u32 *other = xch[r & 1][wv ^ 1];
		if (lane == 63) mine[0] = L.X[7], mine[1] = L.V[7], mine[2] = L.X2[7], mine[3] = L.Qc[3];
		__syncthreads();
		if (lane == 0) pX = other[0], pV = other[1], pX2 = other[2], pQ = other[3];
		// (2) query window advance; the lane whose block fell below the window takes over block +128
		if (r > 0) gdw_shift_query(L, pQ, L.blk == prev_st_, gdw_seam_byte(query, qlen, r - (prev_st_ << 4)));
		if (advanced) {
			if (L.blk < W.st_) gdw_load_block(L, K, L.blk + NBLK, r, query, qlen, target, tlen);
			any_tn = __builtin_amdgcn_ballot_w64(L.tn != 0) != 0;
		}
		// (3) scalar fix-ups and the score row
		if (W.set_tr) gdw_reset_tr(L, K, W);
		if (W.st0 != prev_st0 || W.up != prev_up || advanced) gdw_make_sel(L, W.st0, W.up);
		gdw_update_scores(L, K, any_tn);
		// (4) DP cells of the lanes inside the reference's 16-aligned window
		if (L.blk <= W.en_) {
			u32 out[4];
			gdw_compute<true>(L, K, W, pX, pV, pX2, out);
			*reinterpret_cast<uint4 *>(p + (size_t)r * row_bytes + ((L.blk - W.st_) << 4)) = make_uint4(out[0], out[1], out[2], out[3]);
		}
		// (5) score trackers
		if (r == 0) L.R = gdw_lo(L.V[0]) - K.B1 - K.qe8;
		else L.R += gdw_lo(L.V[0]) - K.B1;
		if (r > 0 && W.en0 != prev_en0 && (W.en0 & 15) == 0) { // uniform over the workgroup: a block enters the band
			const int ho = gdw_track_handoff(L);
			int h = (int)gdw_ror1<64>((u32)ho);
			if (lane == 63) mine[4] = (u32)ho;
			__syncthreads();
			if (lane == 0) h = (int)other[4];
			if (L.blk == W.en_) L.R = h + gdw_lo(L.U[0]);
		}
		if (W.en0 == tlen - 1) {
			if (L.blk == mlast) {
				if (!have_f) Rf = gdw_track_to_slot(L, sl);
				else Rf += gdw_cell(L.V, sl) - K.B1;
			}
			have_f = 1;
		}
		prev_st_ = W.st_, prev_st0 = W.st0, prev_up = W.up, prev_en0 = W.en0;
	}
	const bool fuse = cigar != nullptr;
	if (L.blk == mlast) {
		score_out[tid] = Rf >> 3;
		status[tid] = fuse ? GD_ST_TRACED : GD_ST_DONE;
	}
	if (fuse) { // one of the two wavefronts walks the alignment back, once both have stored their last rows
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
		__syncthreads();
		if (wv == 0) gd_bt_wave_walk(*Tp, tid, bt, n_cigar, cigar, lane);
	}
}

static inline void gd_launch_wave2x64(const KswTask *tasks, const int32_t *ids, int n, const uint8_t *q, const uint8_t *t,
                                      uint8_t *bt, int32_t *status, int32_t *score, KswConst C, hipStream_t s, int32_t *n_cigar = nullptr,
                                      uint32_t *cigar = nullptr)
{
	WaveK K;
	gdw_make_consts(C, K);
	hipLaunchKernelGGL(ksw_extd2_wave2x64_kernel, dim3(n), dim3(128), 0, s, tasks, ids, n, q, t, bt, status, score, K, n_cigar, cigar);
}
