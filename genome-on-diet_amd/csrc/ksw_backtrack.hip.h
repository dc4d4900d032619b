// K0 (exact-match pre-filter, SR/exact_match_sse.c:23-91) and K2 (ksw_backtrack, SR/ksw2.h:131-163).
// Both are one-thread-per-alignment kernels: K0 is a byte compare of <300 bp, K2 a dependent pointer chase of
// qlen+tlen one-byte loads whose latency is hidden by running thousands of alignments side by side.
#pragma once
#include <hip/hip_runtime.h>
#include "ksw_common.h"

// position of cell (r, i) inside a backtrace row, for the two layouts the DP kernels write
//   generic: the reference's own layout, column i - off[r] with off[r] = st0(r)/16*16 (SR/ksw2.h:142)
//   wave:    lane-major ring: 16-cell block i>>4 lives in lane (i>>4) mod L (L = row_bytes/16; the two-blocks-per-lane
//            kernel instead keeps the generic block position (i>>4) - off/16); inside the
//            16 bytes of a lane cells are stored in the order the packed registers hold them
//            (byte 4g+h holds cell 2g+(h&1)+8*(h>>1): cells k, k+1, k+8, k+9 of register pair g; see ksw_wave.hip.h)
__device__ __forceinline__ size_t gd_bt_index(const KswTask &T, int r, int i, int off)
{
	if (T.kind == GD_KIND_GENERIC) return (size_t)r * T.row_bytes + (size_t)(i - off);
	const int lanes = T.row_bytes >> 4;
	const int c = i & 15;
	const int g = (c & 7) >> 1, h = (c & 1) | ((c >> 3) << 1);
	// 16/64-lane kernels: ring of `lanes` blocks (the 10- and 8-lane groups of the short-read kernel hold every block of their
	// alignment at once: no ring); two-blocks-per-lane kernel: the reference's window-relative block position
	const int bpos = T.kind == GD_KIND_WAVE128 ? (i >> 4) - (off >> 4) : (lanes & (lanes - 1)) ? (i >> 4) : ((i >> 4) & (lanes - 1));
	return (size_t)r * T.row_bytes + (size_t)(bpos << 4) + (g << 2) + h;
}

// byte of the register-resident kernels -> the reference's backtrace byte
static inline __host__ __device__ uint32_t gd_bt_decode(uint32_t b)
{
	const uint32_t nb = ~b;
	return (4u - (b & 7u)) | ((nb >> 4) & 0x08u) | ((nb >> 2) & 0x10u) | (nb & 0x20u) | ((nb << 2) & 0x40u);
}

// Pre-filter (exact_match_sse, LR/map.c:1748-1806), widened by what can be PROVEN about the DP without running it.  For an N-free pair
// with qlen == tlen = n and m mismatches the main diagonal scores D = (n - m) a - m b (a = sc_mch, b = -sc_mis).
//   (1) No DP.  A path from corner to corner that is not the diagonal holds at least one insertion run and one deletion run of equal total
//       length G >= 1; every run costs at least min(q + e, q2 + e2) = q + e (normalised so), and only n - G pairs are left to score at most a
//       each: such a path scores at most (n - 1) a - 2 (q + e).  If m (a + b) <= a + 2 (q + e) no other path beats the diagonal, so
//       ksw_extd2 / ksw_extz2 return score D and, by (2) -- which only needs the score to EQUAL D, so a tie is fine --, the CIGAR "<n>M":
//       the alignment is answered here (status EXACT) exactly as the reference's own pre-filter answers m == 0.  With the short-read
//       scoring (a 2, b 8, q 12, e 2): m <= 3 -- 93 % of 150-base reads at 1 % substitutions.  `gap_thr` = a + 2 (q + e) + 1 (the
//       comparison below is strict); 0 switches (1) off.
//   (2) No walk.  diag[tid] = D for the alignments that do go through the DP (GD_NEG_INF where there is an N or qlen != tlen): if the DP
//       score equals it, H(n, n) = sum over the diagonal of [H(k, k) - H(k - 1, k - 1)] with every term >= s(k, k), so equality makes the
//       diagonal move a maximum in every cell of the diagonal, and the reference's priority order takes the diagonal move first
//       (SR/ksw2_extd2_sse.c:235-242): the walk from the last cell stays on the diagonal (ksw_backtrack_kernel writes "<n>M" unwalked).
__global__ __launch_bounds__(64) void ksw_exact_match_kernel(const KswTask *__restrict__ tasks, int n,
                                                             const uint8_t *__restrict__ qseq,
                                                             const uint8_t *__restrict__ tseq,
                                                             int32_t *__restrict__ status, int32_t *__restrict__ score,
                                                             int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar,
                                                             int32_t *__restrict__ diag, int sc_mch, int sc_mis, int gap_thr)
{
	const int tid = blockIdx.x * blockDim.x + threadIdx.x;
	if (tid >= n) return;
	const KswTask T = tasks[tid];
	int st = GD_ST_PENDING, dg = GD_NEG_INF;
	const bool want_exact = T.exact_score != GD_NEG_INF, want_diag = diag && T.kind == GD_KIND_WAVE16;
	if ((want_exact || want_diag) && T.qlen == T.tlen && T.qlen > 0) {
		const uint8_t *q = qseq + T.qoff, *t = tseq + T.toff;
		// eight bases per load and compare (the windows sit at arbitrary byte offsets of the packed buffers: loads through memcpy, which the
		// compiler may turn into unaligned 8-byte accesses where the target allows them), the tail byte by byte.  Without the diagonal's
		// score to compute a thread stops at its first difference.
		int n_mis = 0, k = 0;
		uint64_t any = 0;
		for (; (want_diag || !n_mis) && k + 8 <= T.qlen; k += 8) {
			uint64_t a, b;
			__builtin_memcpy(&a, q + k, 8), __builtin_memcpy(&b, t + k, 8);
			const uint64_t x = a ^ b; // (nt4 codes: three bits per byte)
			n_mis += __builtin_popcountll((x | x >> 1 | x >> 2) & 0x0101010101010101ull);
			any |= a | b;
		}
		for (; (want_diag || !n_mis) && k < T.qlen; ++k) n_mis += q[k] != t[k], any |= (uint64_t)(q[k] | t[k]);
		if (want_exact && !n_mis) {
			st = GD_ST_EXACT;
			score[tid] = T.exact_score;
			n_cigar[tid] = 1;
			if (T.cig_cap >= 1) cigar[T.cig_off] = (uint32_t)T.qlen << 4; // "<qlen>M", LR/map.c:1783-1784
		}
		if (want_diag && !(any & 0x0404040404040404ull)) { // (codes >= 4: N and its complements)
			dg = (T.qlen - n_mis) * sc_mch + n_mis * sc_mis;
			if (st == GD_ST_PENDING && n_mis * (sc_mch - sc_mis) < gap_thr) { // (1): no other path can reach the diagonal's score
				st = GD_ST_EXACT;
				score[tid] = dg;
				n_cigar[tid] = 1;
				if (T.cig_cap >= 1) cigar[T.cig_off] = (uint32_t)T.qlen << 4;
			}
		}
	}
	status[tid] = st;
	if (diag) diag[tid] = dg;
}

// ksw_backtrack (SR/ksw2.h:131-163) of one alignment by ONE THREAD, from cell (i, j): the body of ksw_backtrack_kernel, and the tail of
// the short-alignment DP kernels, whose group leaders walk their own alignments back (ksw_wave.hip.h)
__device__ __forceinline__ void gd_bt_thread_walk(const KswTask &T, int tid, const uint8_t *__restrict__ bt, int32_t *__restrict__ n_cigar,
                                                  uint32_t *__restrict__ cigar, int i, int j)
{
	const int qlen = T.qlen, tlen = T.tlen;
	const int w = T.w < 0 ? (tlen > qlen ? tlen : qlen) : T.w;
	const uint8_t *p = bt + T.bt_off;
	uint32_t *cg = cigar + T.cig_off;
	const int cap = T.cig_cap;
	int nc = 0, state = 0;
	uint32_t last = 0; // the op run being extended (kept in a register, flushed on change)
	int have = 0;
#define GD_PUSH(op_, len_)                                                       \
	do {                                                                         \
		if (have && (last & 0xf) == (uint32_t)(op_)) last += (uint32_t)(len_) << 4; \
		else {                                                                   \
			if (have) { if (nc < cap) cg[nc] = last; ++nc; }                     \
			last = (uint32_t)(len_) << 4 | (uint32_t)(op_), have = 1;            \
		}                                                                        \
	} while (0)
	// The walk is a chain of dependent one-byte loads (one HBM round trip per step if done naively).  Alignments stay on a
	// diagonal most of the time, so the bytes of the next GD_BT_PF cells along the current diagonal are fetched together
	// (independent loads, one round trip) and consumed for as long as the path really moves diagonally; any indel step
	// leaves the predicted diagonal and triggers a new fetch from the cell it arrives at.  Pure latency hiding: the visited
	// cells and the state machine are exactly those of ksw_backtrack.
#define GD_BT_PF 16
	while (i >= 0 && j >= 0) {
		uint8_t pf[GD_BT_PF];
		int fs[GD_BT_PF];
		const int i0 = i, j0 = j;
#pragma unroll
		for (int k = 0; k < GD_BT_PF; ++k) {
			const int ik = i0 - k, jk = j0 - k;
			fs[k] = -1, pf[k] = 0;
			if (ik >= 0 && jk >= 0) {
				const int r = ik + jk;
				int st0, en0;
				gd_band(r, qlen, tlen, w, st0, en0);
				const int off = st0 & ~15, off_end = en0 | 15;
				if (ik < off) fs[k] = 2;
				if (ik > off_end) fs[k] = 1;
				if (fs[k] < 0) pf[k] = p[gd_bt_index(T, r, ik, off)];
			}
		}
#pragma unroll
		for (int k = 0; k < GD_BT_PF; ++k) {
			if (i != i0 - k || j != j0 - k || i < 0 || j < 0) break; // left the predicted diagonal (or finished)
			const int force_state = fs[k];
			uint32_t tmp = pf[k];
			if (force_state < 0 && T.kind != GD_KIND_GENERIC) {
				// wave kernels store (4-d) | nY2<<4 | nX2<<5 | nY<<6 | nX<<7 with n* = "no continuation" (bit 3 undefined); rebuild
				// the reference's byte d | cX<<3 | cY<<4 | cX2<<5 | cY2<<6 (SR/ksw2.h:127-130)
				tmp = gd_bt_decode(tmp);
			}
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) { GD_PUSH(0, 1); --i, --j; }
			else if (state == 1 || state == 3) { GD_PUSH(2, 1); --i; }
			else { GD_PUSH(1, 1); --j; }
		}
	}
#undef GD_BT_PF
	if (i >= 0) GD_PUSH(2, i + 1);
	if (j >= 0) GD_PUSH(1, j + 1);
	if (have) { if (nc < cap) cg[nc] = last; ++nc; }
#undef GD_PUSH
	n_cigar[tid] = nc;
	if (nc <= cap)
		for (int k = 0; k < nc >> 1; ++k) {
			const uint32_t t0 = cg[k];
			cg[k] = cg[nc - 1 - k], cg[nc - 1 - k] = t0;
		}
}

__global__ __launch_bounds__(64) void ksw_backtrack_kernel(const KswTask *__restrict__ tasks, int n,
                                                           const uint8_t *__restrict__ bt,
                                                           const int32_t *__restrict__ status,
                                                           int32_t *__restrict__ score, int32_t *__restrict__ n_cigar,
                                                           uint32_t *__restrict__ cigar, int spread, const int32_t *__restrict__ task_ids,
                                                           const int32_t *__restrict__ start = nullptr /* (i0, j0) per task; default: the last cell */,
                                                           const int32_t *__restrict__ diag = nullptr /* ksw_exact_match_kernel's diagonal scores */)
{
	// spread = 1: one alignment per WAVEFRONT (lane 0 walks, the other lanes idle).  Kept for experiments only: it measured 2x
	// SLOWER than one walk per thread, whose 64 x 16 prefetched loads per wavefront hide the latency better.
	int tid = blockIdx.x * blockDim.x + threadIdx.x;
	if (spread) {
		if (threadIdx.x & 63) return;
		tid >>= 6;
	}
	if (tid >= n) return;
	if (task_ids) { tid = task_ids[tid]; if (tid < 0) return; } // a sub-list of the batch (-1: padding of a 16-lane quartet)
	const int st = status[tid];
	if (st == GD_ST_EXACT || st == GD_ST_TRACED) return;
	if (st != GD_ST_DONE) { // band emptied (zdropped): no CIGAR, score stays KSW_NEG_INF (SR/ksw2_extd2_sse.c:142-145,391)
		n_cigar[tid] = 0;
		if (st == GD_ST_ZDROPPED) score[tid] = GD_NEG_INF;
		return;
	}
	const KswTask T = tasks[tid];
	if (diag && !start && T.qlen == T.tlen && diag[tid] != GD_NEG_INF && score[tid] == diag[tid]) { // the walk would stay on the main diagonal (see ksw_exact_match_kernel)
		n_cigar[tid] = 1;
		if (T.cig_cap >= 1) cigar[T.cig_off] = (uint32_t)T.qlen << 4;
		return;
	}
	gd_bt_thread_walk(T, tid, bt, n_cigar, cigar, start ? start[2 * tid] : T.tlen - 1, start ? start[2 * tid + 1] : T.qlen - 1);
}

// K2 for long alignments: one WAVEFRONT per walk.  All 64 lanes fetch the next 64 cells of the current diagonal at once (one
// round trip instead of one per cell), then the walk itself -- wave-uniform by construction -- runs as scalar code that pulls the
// prefetched bytes out of the vector registers with v_readlane.  Same visited cells, same state machine as ksw_backtrack
// (SR/ksw2.h:131-163); only the latency is hidden differently than in the one-walk-per-thread kernel above, which stays the
// better choice for short reads (hundreds of thousands of 300-step walks).
// the walk of one alignment by one whole wavefront (all 64 lanes must call it together); writes the CIGAR and n_cigar[tid]
__device__ __forceinline__ void gd_bt_wave_walk(const KswTask &T, int tid, const uint8_t *__restrict__ bt, int32_t *__restrict__ n_cigar,
                                                uint32_t *__restrict__ cigar, int lane)
{
	const int qlen = __builtin_amdgcn_readfirstlane(T.qlen), tlen = __builtin_amdgcn_readfirstlane(T.tlen);
	const int w0 = __builtin_amdgcn_readfirstlane(T.w);
	const int w = w0 < 0 ? (tlen > qlen ? tlen : qlen) : w0;
	const int kind = __builtin_amdgcn_readfirstlane(T.kind);
	const uint8_t *p = bt + T.bt_off;
	uint32_t *cg = cigar + T.cig_off;
	const int cap = __builtin_amdgcn_readfirstlane(T.cig_cap);
	int nc = 0, i = tlen - 1, j = qlen - 1, state = 0, have = 0;
	uint32_t last = 0;
#define GD_PUSHW(op_, len_)                                                      \
	do {                                                                         \
		if (have && (last & 0xf) == (uint32_t)(op_)) last += (uint32_t)(len_) << 4; \
		else {                                                                   \
			if (have) { if (nc < cap && lane == 0) cg[nc] = last; ++nc; }        \
			last = (uint32_t)(len_) << 4 | (uint32_t)(op_), have = 1;            \
		}                                                                        \
	} while (0)
	while (i >= 0 && j >= 0) {
		const int i0 = i, j0 = j;
		const int ik = i0 - lane, jk = j0 - lane;
		int fs = -1;
		uint32_t pf = 0;
		if (ik >= 0 && jk >= 0) {
			const int r = ik + jk;
			int st0, en0;
			gd_band(r, qlen, tlen, w, st0, en0);
			const int off = st0 & ~15, off_end = en0 | 15;
			if (ik < off) fs = 2;
			if (ik > off_end) fs = 1;
			if (fs < 0) pf = p[gd_bt_index(T, r, ik, off)];
		}
		for (int k = 0; k < 64; ++k) {
			if (i != i0 - k || j != j0 - k || i < 0 || j < 0) break; // left the fetched diagonal (or finished)
			const int force_state = __builtin_amdgcn_readlane(fs, k);
			uint32_t tmp = (uint32_t)__builtin_amdgcn_readlane((int)pf, k);
			if (force_state < 0 && kind != GD_KIND_GENERIC) { // wave-kernel byte -> the reference's byte (see ksw_backtrack_kernel)
				tmp = gd_bt_decode(tmp);
			}
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) { GD_PUSHW(0, 1); --i, --j; }
			else if (state == 1 || state == 3) { GD_PUSHW(2, 1); --i; }
			else { GD_PUSHW(1, 1); --j; }
		}
	}
	if (i >= 0) GD_PUSHW(2, i + 1);
	if (j >= 0) GD_PUSHW(1, j + 1);
	if (have) { if (nc < cap && lane == 0) cg[nc] = last; ++nc; }
#undef GD_PUSHW
	if (lane == 0) n_cigar[tid] = nc;
	if (nc <= cap) { // reverse in place, 64 swaps at a time (the stores above are this wavefront's own: program order suffices)
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // lane 0 wrote the ops, all lanes read them
		for (int k = lane; k < nc >> 1; k += 64) {
			const uint32_t t0 = cg[k];
			cg[k] = cg[nc - 1 - k], cg[nc - 1 - k] = t0;
		}
	}
}

// The same walk in resumable form, for a backtrace that exists one chunk of anti-diagonals at a time (ksw_extd2_wave128c_kernel):
// gd_walk_rows consumes cells while their anti-diagonal is >= r0 (rows of `chunk`, row r at (r - r0) * row_bytes, the layout of
// the two-blocks-per-lane kernel or of the cone) and returns with its state, to be called again on the chunk below.
struct GdWalk { int i, j, state, have, nc; uint32_t last; };
__device__ __forceinline__ void gd_walk_init(GdWalk &W, int qlen, int tlen) { W.i = tlen - 1, W.j = qlen - 1, W.state = 0, W.have = 0, W.nc = 0, W.last = 0; }
__device__ __forceinline__ void gd_walk_push(GdWalk &W, uint32_t *cg, int cap, int lane, uint32_t op, uint32_t len)
{
	if (W.have && (W.last & 0xf) == op) W.last += len << 4;
	else {
		if (W.have) { if (W.nc < cap && lane == 0) cg[W.nc] = W.last; ++W.nc; }
		W.last = len << 4 | op, W.have = 1;
	}
}
// cone_stride > 0: rows of `chunk` hold the 64 blocks [cone_b0, cone_b0 + 63] (gdw_cone_row) instead of the blocks of the band
__device__ __forceinline__ void gd_walk_rows(GdWalk &W, const KswTask &T, const uint8_t *__restrict__ chunk, int r0, int qlen, int tlen, int w,
                                             uint32_t *__restrict__ cigar, int lane, int cone_stride = 0, int cone_b0 = 0, bool cone_half = false)
{
	uint32_t *cg = cigar + T.cig_off;
	const int cap = __builtin_amdgcn_readfirstlane(T.cig_cap);
	const size_t row_bytes = cone_stride > 0 ? (size_t)cone_stride : (size_t)__builtin_amdgcn_readfirstlane(T.row_bytes);
	while (W.i >= 0 && W.j >= 0 && W.i + W.j >= r0) {
		const int i0 = W.i, j0 = W.j;
		const int ik = i0 - lane, jk = j0 - lane;
		int fs = -1;
		uint32_t pf = 0;
		if (ik >= 0 && jk >= 0 && ik + jk >= r0) {
			const int r = ik + jk;
			int st0, en0;
			gd_band(r, qlen, tlen, w, st0, en0);
			const int off = st0 & ~15, off_end = en0 | 15;
			if (ik < off) fs = 2;
			if (ik > off_end) fs = 1;
			if (fs < 0) {
				if (cone_half) { // rows of 64 half blocks (gdw_cone_row_half): cone_b0 is a half-block index, byte 4g + h = cell 2g + (h & 1) + 4 (h >> 1)
					const int c = ik & 7, g = (c & 3) >> 1, h = (c & 1) | ((c >> 2) << 1);
					pf = chunk[(size_t)(r - r0) * row_bytes + (size_t)(((ik >> 3) - cone_b0) << 3) + (g << 2) + h];
				} else {
					const int c = ik & 15, g = (c & 7) >> 1, h = (c & 1) | ((c >> 3) << 1);
					const int b = (ik >> 4) - (cone_stride > 0 ? cone_b0 : off >> 4);
					pf = chunk[(size_t)(r - r0) * row_bytes + (size_t)(b << 4) + (g << 2) + h];
				}
			}
		}
		for (int k = 0; k < 64; ++k) {
			if (W.i != i0 - k || W.j != j0 - k || W.i < 0 || W.j < 0 || W.i + W.j < r0) break; // left the fetched diagonal, the chunk, or finished
			const int force_state = __builtin_amdgcn_readlane(fs, k);
			uint32_t tmp = (uint32_t)__builtin_amdgcn_readlane((int)pf, k);
			if (force_state < 0) tmp = gd_bt_decode(tmp);
			if (W.state == 0) W.state = tmp & 7;
			else if (!(tmp >> (W.state + 2) & 1)) W.state = 0;
			if (W.state == 0) W.state = tmp & 7;
			if (force_state >= 0) W.state = force_state;
			if (W.state == 0) { gd_walk_push(W, cg, cap, lane, 0, 1); --W.i, --W.j; }
			else if (W.state == 1 || W.state == 3) { gd_walk_push(W, cg, cap, lane, 2, 1); --W.i; }
			else { gd_walk_push(W, cg, cap, lane, 1, 1); --W.j; }
		}
	}
}
__device__ __forceinline__ void gd_walk_finish(GdWalk &W, const KswTask &T, int tid, int32_t *__restrict__ n_cigar, uint32_t *__restrict__ cigar, int lane)
{
	uint32_t *cg = cigar + T.cig_off;
	const int cap = __builtin_amdgcn_readfirstlane(T.cig_cap);
	if (W.i >= 0) gd_walk_push(W, cg, cap, lane, 2, (uint32_t)(W.i + 1));
	if (W.j >= 0) gd_walk_push(W, cg, cap, lane, 1, (uint32_t)(W.j + 1));
	if (W.have) { if (W.nc < cap && lane == 0) cg[W.nc] = W.last; ++W.nc; }
	if (lane == 0) n_cigar[tid] = W.nc;
	if (W.nc <= cap) {
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		for (int k = lane; k < W.nc >> 1; k += 64) {
			const uint32_t t0 = cg[k];
			cg[k] = cg[W.nc - 1 - k], cg[W.nc - 1 - k] = t0;
		}
	}
}

__global__ __launch_bounds__(256) void ksw_backtrack_wave_kernel(const KswTask *__restrict__ tasks, int n,
                                                                 const uint8_t *__restrict__ bt,
                                                                 const int32_t *__restrict__ status,
                                                                 int32_t *__restrict__ score, int32_t *__restrict__ n_cigar,
                                                                 uint32_t *__restrict__ cigar, const int32_t *__restrict__ task_ids)
{
	const int lane = threadIdx.x & 63;
	int tid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
	if (tid >= n) return;
	if (task_ids) { tid = __builtin_amdgcn_readfirstlane(task_ids[tid]); if (tid < 0) return; }
	const int st = __builtin_amdgcn_readfirstlane(status[tid]);
	if (st == GD_ST_EXACT || st == GD_ST_TRACED) return;
	if (st != GD_ST_DONE) {
		if (lane == 0) {
			n_cigar[tid] = 0;
			if (st == GD_ST_ZDROPPED) score[tid] = GD_NEG_INF;
		}
		return;
	}
	gd_bt_wave_walk(tasks[tid], tid, bt, n_cigar, cigar, lane);
}
