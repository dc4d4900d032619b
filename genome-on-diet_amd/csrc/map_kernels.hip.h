// Device executors of the seeding / voting stages (S1-S7, V1, V3 and the first half of G1): one read per thread,
// running the host/device-shared code of map_stages.h against the device-resident index.  These stages are ~2 % of
// the reference's per-read time on the HiFi configuration; thread-per-read keeps them bit-identical
// to the host build of the same source, which tests/test_map_host.py checks against the reference.
//   map_seed_kernel : sketch2 -> get_shift -> sketch3 -> mz_flt -> collect_matches2        (per-read seed list, n_a)
//   map_vote_kernel : seed hits -> sort -> vote / density+vt_f filters / vote_2             (<= vt_nb_loc+2 candidates)
//   map_gather_kernel: DP boxes -> packed nt4 query / target windows for the ksw batch       (one wavefront per box)
#pragma once
#include <hip/hip_runtime.h>
#include "map_stages.h"
#include "map_post.h"

struct MapDevOpt { // uniform per batch
	int32_t k, w;
	float max_seeds, q_occ_frac;
	int32_t mid_occ, max_max_occ, occ_dist;
	uint32_t max_nb_seeds; // cap of mm_sketch3 (UINT32_MAX unless frag mode)
	int64_t flag;
	GdLrVoteOpt vote;
	GdSrVoteOpt sr;    // ShortReads variant (flag & MM_F_SR)
	int32_t is_sr;
	int32_t sort_cap;  // hashes of one read the wave seed kernel sorts in LDS (a power of two; the launch provides 8 B each)
	uint32_t seed_lds; // dynamic LDS bytes of the wave seed kernel's launch
	int32_t prio;      // s_setprio of the wave seed / vote kernels' wavefronts (0-3): they run beside the DP wavefronts of another batch
	GdPattern pat;
};
__device__ __forceinline__ void map_set_prio(int p)
{
	if (p == 1) __builtin_amdgcn_s_setprio(1);
	else if (p == 2) __builtin_amdgcn_s_setprio(2);
	else if (p >= 3) __builtin_amdgcn_s_setprio(3);
}

struct MapReadScratch { // per-read slices of the batch scratch arena (element offsets)
	uint64_t mv_off;     // GdMini[mv_cap]
	uint64_t u64_off;    // uint64[2*mv_cap]
	uint64_t seed_off;   // GdSeed[mv_cap]
	uint32_t mv_cap;
	uint32_t pad;
};

struct MapSeedOut {
	int32_t n_seeds;  // kept seeds (< 0: scratch overflow)
	int32_t shift;
	uint32_t tel;     // tmp_extracted_len
	uint32_t n_mv;    // mv.n after mm_seed_mz_flt (the ShortReads vote thresholds scale with it, SR/map.c:667-676)
	int64_t n_a;      // total occurrences of the kept seeds
};

// mm_sketch2 / mm_sketch3 of one read by one thread with the winnowing window in LDS (entry j of this thread's window at win[j * 64]): the
// bodies of gd_sketch2 / gd_sketch3 (map_stages.h) around the EXT_WIN form of the automaton.  As a local array the window is 1 KB of
// scratch memory per thread and every window access a scratch load / store.
__device__ __forceinline__ unsigned map_sketch2_lds(const uint8_t *str, int len, int w, int k, const GdPattern &P, float max_seeds, GdMini *out, unsigned max_out,
                                                    uint32_t *shift_n, GdMini *win)
{
	unsigned len_crop, total = 0;
	uint32_t cap;
	if (max_seeds < 1) len_crop = (unsigned)((float)max_seeds * len), cap = UINT32_MAX;
	else len_crop = len, cap = (uint32_t)max_seeds;
	for (int shift = 0; shift < P.W; ++shift) {
		const unsigned dl = gd_diet_len(P, len_crop, shift);
		GdEmitCount e = {out + total, 0, cap, max_out - total, false};
		gd_sketch_range<GdEmitCount, true>(str, 0, 0, dl, 0, true, w, k, 0, (unsigned)shift, P, true, e, win, 64);
		if (e.overflow) return ~0u;
		shift_n[shift] = e.n;
		total += e.n;
		if (cap == UINT32_MAX) len_crop = len, cap = e.n;
	}
	return total;
}
__device__ __forceinline__ unsigned map_sketch3_lds(const uint8_t *str, unsigned len, int w, int k, const GdPattern &P, int shift, uint32_t max_nb_seeds, GdMini *out,
                                                    unsigned max_out, unsigned *n_out, GdMini *win)
{
	if (shift < 0) shift = 0;
	const unsigned dl = gd_diet_len(P, len, (unsigned)shift);
	GdEmitCap e = {out, 0, max_nb_seeds, max_out, len, false};
	gd_sketch_range<GdEmitCap, true>(str, 0, 0, dl, 0, true, w, k, 0, (unsigned)shift, P, true, e, win, 64);
	*n_out = e.overflow ? ~0u : e.n;
	return e.ret;
}

__global__ __launch_bounds__(64) void map_seed_kernel(int n_reads, const uint8_t *__restrict__ reads, const int64_t *__restrict__ roff,
                                                      GdIdxView I, MapDevOpt O, const MapReadScratch *__restrict__ sc, GdMini *__restrict__ mv_arena,
                                                      uint64_t *__restrict__ u64_arena, GdSeed *__restrict__ seed_arena, MapSeedOut *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t seed_thread_lds[]; // the 64 threads' winnowing windows, w x 64 entries, thread-interleaved
	GdMini *win = reinterpret_cast<GdMini *>(seed_thread_lds) + threadIdx.x;
	const int rid = blockIdx.x * blockDim.x + threadIdx.x;
	if (rid >= n_reads) return;
	const uint8_t *str = reads + roff[rid];
	const int len = (int)(roff[rid + 1] - roff[rid]);
	MapSeedOut o;
	o.n_seeds = 0, o.shift = 0, o.tel = (uint32_t)len, o.n_mv = 0, o.n_a = 0;
	if (len <= 0) { out[rid] = o; return; }
	const MapReadScratch S = sc[rid];
	GdMini *mv = mv_arena + S.mv_off;
	uint32_t shift_n[64];
	const unsigned tot = map_sketch2_lds(str, len, O.w, O.k, O.pat, O.max_seeds, mv, S.mv_cap, shift_n, win);
	if (tot == ~0u) { o.n_seeds = -1; out[rid] = o; return; }
	o.shift = (int32_t)gd_get_shift(I, mv, shift_n, O.pat.W);
	unsigned n_mv = 0;
	o.tel = map_sketch3_lds(str, (unsigned)len, O.w, O.k, O.pat, o.shift, O.max_nb_seeds, mv, S.mv_cap, &n_mv, win);
	if (n_mv == ~0u) { o.n_seeds = -1; out[rid] = o; return; }
	if (O.q_occ_frac > 0.0f) n_mv = gd_mz_flt(mv, n_mv, O.mid_occ, O.q_occ_frac, u64_arena + S.u64_off);
	o.n_mv = n_mv;
	o.n_seeds = gd_collect_matches2(I, mv, n_mv, len, O.mid_occ, O.max_max_occ, O.occ_dist, seed_arena + S.seed_off, &o.n_a);
	out[rid] = o;
}

struct MapVoteOut {
	uint32_t n_cand, pad;
	GdVt cand[GDM_MAX_VT];
};

__global__ __launch_bounds__(64) void map_vote_kernel(int n_reads, const int64_t *__restrict__ roff, GdIdxView I, MapDevOpt O,
                                                      const MapReadScratch *__restrict__ sc, const GdSeed *__restrict__ seed_arena,
                                                      const MapSeedOut *__restrict__ seeds, const int64_t *__restrict__ hit_off,
                                                      GdLoc *__restrict__ hits /* 3 slices per read: for, rev, tmp */, MapVoteOut *__restrict__ out,
                                                      int spread)
{
	int rid = blockIdx.x * blockDim.x + threadIdx.x;
	if (spread) { // one read per wavefront, lane 0 only (see ksw_backtrack_kernel)
		if (threadIdx.x & 63) return;
		rid >>= 6;
	}
	if (rid >= n_reads) return;
	MapVoteOut &o = out[rid];
	o.n_cand = 0, o.pad = 0;
	const MapSeedOut so = seeds[rid];
	if (so.n_seeds <= 0) return;
	const int len = (int)(roff[rid + 1] - roff[rid]);
	const int64_t na = so.n_a;
	GdLoc *a_for = hits + 3 * hit_off[rid], *a_rev = a_for + na, *tmp = a_rev + na;
	unsigned nf = 0, nr = 0;
	gd_seed_hits(I, seed_arena + sc[rid].seed_off, so.n_seeds, O.flag, so.tel, a_for, a_rev, &nf, &nr);
	// the merge sort ping-pongs between the array and tmp; sort the forward hits first, park them, then the reverse hits
	GdLoc *sf = gd_sort_locs(a_for, tmp, nf);
	if (sf != a_for) for (unsigned i = 0; i < nf; ++i) a_for[i] = sf[i];
	GdLoc *sr = gd_sort_locs(a_rev, tmp, nr);
	if (O.is_sr) o.n_cand = gd_sr_candidates(a_for, nf, sr, nr, (uint32_t)len, so.tel, so.n_mv, O.sr, o.cand);
	else o.n_cand = gd_lr_candidates(a_for, nf, sr, nr, (uint32_t)len, (int32_t)so.tel, O.vote, o.cand);
}

// ---- wave-parallel form of map_vote_kernel: one 64-lane wavefront per read ----------------------------------------------------
// The occurrences of 64 seeds are expanded side by side (ballot + prefix count give every hit its slot on its strand), each
// strand's hits are sorted by target with a wavefront bitonic sort in LDS, and the whole wavefront runs the (wave-uniform) vote
// scans in lockstep on a 64-hits-per-round-trip view of the sorted arrays.
// A strand with more than MAP_VOTE_CAP hits is sorted in LDS-sized runs that are then merged in global memory by the whole wavefront.
#define MAP_VOTE_CAP 4096 // (the array bound; the launches use gd_vote_cap_max(): 1024 by default)

// Front-to-back view of a sorted hit array for the vote scans, executed by ALL lanes of the wavefront in lockstep: the lanes
// load 64 consecutive hits at once and hand them out through v_readlane, so the (wave-uniform, hence scalar) scan pays one LDS
// round trip per 64 hits instead of one per hit.
struct MapWaveLocs {
	const GdLoc *base;
	unsigned n;
	mutable unsigned chunk;
	mutable uint64_t t;
	mutable uint32_t q;
	__device__ GdLoc operator[](unsigned i) const
	{
		const unsigned c = i & ~63u;
		if (c != chunk) {
			chunk = c;
			const unsigned j = c + (threadIdx.x & 63);
			t = j < n ? base[j].target : ~0ull, q = j < n ? base[j].query : 0u;
		}
		const int l = (int)(i & 63);
		GdLoc r;
		r.target = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(t >> 32), l) << 32;
		r.query = (uint32_t)__builtin_amdgcn_readlane((int)q, l), r.pad = 0;
		return r;
	}
};

__device__ __forceinline__ void map_bitonic_locs(GdLoc *a, unsigned P2, unsigned lane)
{
	for (unsigned kk = 2; kk <= P2; kk <<= 1)
		for (unsigned jj = kk >> 1; jj > 0; jj >>= 1) {
			for (unsigned i = lane; i < P2; i += 64) {
				const unsigned ixj = i ^ jj;
				if (ixj > i) {
					const GdLoc x = a[i], y = a[ixj];
					const bool up = (i & kk) == 0;
					if ((x.target > y.target) == up) a[i] = y, a[ixj] = x;
				}
			}
			__syncthreads();
		}
}

// The sort buffer is DYNAMIC shared memory of vote_cap entries (a power of two <= MAP_VOTE_CAP, sized by the host from the largest
// hit count of the batch: 16 KB for HiFi reads instead of 64).  With a static 64 KB array the compiler, knowing that the LDS allows
// one wavefront per SIMD, padded the kernel's register count from 46 to 264 (the count that pins that occupancy): a wavefront
// that cannot start next to the DP kernel's wavefronts of another batch in flight until three of them have left the same SIMD.
// (Measured in the pipeline: no change of the HiFi step time either way -- the DP kernel leaves little to share -- but the
// kernel no longer depends on the tail of a DP kernel to run.)
__global__ __launch_bounds__(64) void map_vote_wave_kernel(int n_reads, const int64_t *__restrict__ roff, GdIdxView I, MapDevOpt O,
                                                           const MapReadScratch *__restrict__ sc, const GdSeed *__restrict__ seed_arena,
                                                           const MapSeedOut *__restrict__ seeds, const int64_t *__restrict__ hit_off,
                                                           GdLoc *__restrict__ hits, MapVoteOut *__restrict__ out, unsigned vote_cap)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t vote_lds[];
	map_set_prio(O.prio);
	GdLoc *s_buf = reinterpret_cast<GdLoc *>(vote_lds); // one strand at a time
	const int rid = blockIdx.x;
	if (rid >= n_reads) return;
	const unsigned lane = threadIdx.x;
	MapVoteOut &o = out[rid];
	const MapSeedOut so = seeds[rid];
	if (so.n_seeds <= 0) { if (lane == 0) o.n_cand = 0, o.pad = 0; return; }
	const int len = (int)(roff[rid + 1] - roff[rid]);
	const int64_t na = so.n_a;
	GdLoc *a_for = hits + 3 * hit_off[rid], *a_rev = a_for + na, *tmp = a_rev + na;
	const GdSeed *m = seed_arena + sc[rid].seed_off;
	// S6: expansion, 64 seeds at a time (gd_seed_hits; the order inside a strand is irrelevant: it is sorted next)
	unsigned nf = 0, nr = 0;
	for (int base = 0; base < so.n_seeds; base += 64) {
		const int si = base + (int)lane;
		GdSeed q;
		q.n = 0, q.q_pos = 0, q.start = 0, q.flt = 0;
		if (si < so.n_seeds) q = m[si];
		unsigned mx = q.n;
		for (int d = 32; d > 0; d >>= 1) { const unsigned v = __shfl_xor(mx, d); mx = v > mx ? v : mx; }
		for (unsigned k = 0; k < mx; ++k) {
			bool is_for = false, is_rev = false;
			GdLoc L;
			L.target = 0, L.query = 0, L.pad = 0;
			if (k < q.n) {
				const uint64_t r = I.pos[(uint64_t)q.start + k];
				bool skip = false;
				if (O.flag & (GDM_F_FOR_ONLY | GDM_F_REV_ONLY)) { // skip_seed, LR/map.c:724-730
					if ((r & 1) == (q.q_pos & 1)) skip = (O.flag & GDM_F_REV_ONLY) != 0;
					else skip = (O.flag & GDM_F_FOR_ONLY) != 0;
				}
				if (!skip) {
					const uint32_t qpos = q.q_pos >> 1;
					const unsigned str = (unsigned)((r & 1) ^ (q.q_pos & 1));
					uint32_t loc = (uint32_t)r >> 1;
					loc = str ? loc + qpos : loc + so.tel - qpos;
					L.target = (r >> 32) << 32 | loc, L.query = qpos;
					is_rev = str != 0, is_for = !is_rev;
				}
			}
			const uint64_t bf = __ballot(is_for), br = __ballot(is_rev);
			const uint64_t below = (1ull << lane) - 1;
			if (is_for) a_for[nf + __popcll(bf & below)] = L;
			if (is_rev) a_rev[nr + __popcll(br & below)] = L;
			nf += __popcll(bf), nr += __popcll(br);
		}
	}
	__syncthreads();
	// S7: one strand at a time through LDS: padded with +inf to a power of two, bitonic sort by target, back to its global array.
	// A strand with more hits than the LDS buffer holds (long ONT reads: ~9 seeds per 100 bases) is sorted in runs of vote_cap
	// and the runs are merged pairwise in global memory, every lane merging one slice of the output (merge path); the sequential
	// sort by one lane this replaces cost tens of milliseconds for such a read.
	GdLoc inf;
	inf.target = UINT64_MAX, inf.query = 0, inf.pad = 0;
	for (int strand = 0; strand < 2; ++strand) {
		GdLoc *g = strand ? a_rev : a_for;
		const unsigned cnt = strand ? nr : nf;
		if (cnt < 2) continue;
		for (unsigned c0 = 0; c0 < cnt; c0 += vote_cap) {
			const unsigned m_ = cnt - c0 < vote_cap ? cnt - c0 : vote_cap;
			unsigned P2 = 64;
			while (P2 < m_) P2 <<= 1;
			for (unsigned i = lane; i < P2; i += 64) s_buf[i] = i < m_ ? g[c0 + i] : inf;
			__syncthreads();
			map_bitonic_locs(s_buf, P2, lane);
			for (unsigned i = lane; i < m_; i += 64) g[c0 + i] = s_buf[i];
			__syncthreads();
		}
		if (cnt > vote_cap) {
			GdLoc *src = g, *dst = tmp; // tmp holds n_a entries: room for either strand
			for (unsigned width = vote_cap; width < cnt; width <<= 1) {
				for (unsigned s0 = 0; s0 < cnt; s0 += 2 * width) {
					const unsigned a_n = cnt - s0 < width ? cnt - s0 : width, b_n = cnt - s0 - a_n < width ? cnt - s0 - a_n : width;
					const GdLoc *A = src + s0, *B = A + a_n;
					GdLoc *out_ = dst + s0;
					const unsigned tot = a_n + b_n, per = (tot + 63) / 64;
					const unsigned lo = lane * per < tot ? lane * per : tot, hi = lo + per < tot ? lo + per : tot;
					// how many of the first `lo` merged elements come from A (ties: A first)
					unsigned x0 = lo > b_n ? lo - b_n : 0, x1 = lo < a_n ? lo : a_n;
					while (x0 < x1) {
						const unsigned mid = (x0 + x1) >> 1;
						if (A[mid].target <= B[lo - mid - 1].target) x0 = mid + 1;
						else x1 = mid;
					}
					unsigned i = x0, j = lo - x0;
					for (unsigned k = lo; k < hi; ++k) {
						const bool take_a = j >= b_n || (i < a_n && A[i].target <= B[j].target);
						out_[k] = take_a ? A[i++] : B[j++];
					}
				}
				__threadfence_block();
				__syncthreads();
				GdLoc *t_ = src;
				src = dst, dst = t_;
			}
			if (src != g) {
				for (unsigned i = lane; i < cnt; i += 64) g[i] = src[i];
				__threadfence_block();
				__syncthreads();
			}
		}
	}
	// V1 / V2 / V3 / G1a: the scans are sequential in the hits but wave-uniform, so every lane runs them in lockstep on a
	// prefetching view of the sorted arrays (MapWaveLocs); the candidate list lives in LDS, lane by lane identical
	__shared__ GdVt s_cand[GDM_MAX_VT];
	__shared__ unsigned s_ncand;
	const MapWaveLocs vf = {a_for, nf, ~0u, 0, 0}, vr = {a_rev, nr, ~0u, 0, 0}; // the sorted global arrays, 64 hits per round trip
	unsigned nc;
	if (O.is_sr) nc = gd_sr_candidates(vf, nf, vr, nr, (uint32_t)len, so.tel, so.n_mv, O.sr, s_cand);
	else nc = gd_lr_candidates(vf, nf, vr, nr, (uint32_t)len, (int32_t)so.tel, O.vote, s_cand);
	if (lane == 0) s_ncand = nc;
	__syncthreads();
	if (lane == 0) o.n_cand = s_ncand, o.pad = 0;
	for (unsigned i = lane; i < s_ncand; i += 64) o.cand[i] = s_cand[i];
}

// one DP box: where its query / target windows come from and where they go in the packed ksw batch buffers
#define GD_NEG_INF_SCORE_DEV (-0x40000000)
struct MapBox {
	int64_t read_off;   // offset of the read in the nt4 read buffer
	int64_t q_dst, t_dst; // destination offsets in the packed query / target buffers
	uint64_t t_src;     // absolute base offset in S (contig offset + target_start)
	uint32_t read_len, qseq_off, qlen, tlen, t_avail; // t_avail: bases that exist (window clipped at the contig end)
	uint32_t rev;       // query window taken from the reverse-complemented read
};

__global__ __launch_bounds__(64) void map_gather_kernel(int n_box, const MapBox *__restrict__ boxes, const uint8_t *__restrict__ reads,
                                                        const uint32_t *__restrict__ S, uint8_t *__restrict__ qbuf, uint8_t *__restrict__ tbuf)
{
	__builtin_amdgcn_s_setprio(3); // (a short kernel between a batch's voting and its DP stage, beside another batch's DP wavefronts)
	const int b = blockIdx.x;
	if (b >= n_box) return;
	const MapBox B = boxes[b];
	const uint8_t *rd = reads + B.read_off;
	for (uint32_t i = threadIdx.x; i < B.qlen; i += blockDim.x) {
		const uint32_t j = B.qseq_off + i;
		// forward: encoded read; reverse: qs_rev[len-1-p] = enc[p] ^ 3 (an N becomes 7, LR/map.c:1634)
		qbuf[B.q_dst + i] = B.rev ? (uint8_t)(rd[B.read_len - 1 - j] ^ 3) : rd[j];
	}
	for (uint32_t i = threadIdx.x; i < B.tlen; i += blockDim.x) {
		const uint64_t o = B.t_src + i;
		tbuf[B.t_dst + i] = i < B.t_avail ? (uint8_t)(S[o >> 3] >> ((o & 7) << 2) & 0xf) : (uint8_t)0;
	}
}

// ---- G2 on the device (ShortReads variant): the voted diagonals of every read become DP boxes, MapBox records and
// the offset tables of the DP batch without the candidates visiting the host first.  Per 262 144 reads the host stages this replaces
// took 12 ms of a 16-thread pool (boxes 5.7, offsets 2.0, window table 4.8) -- as long as the GPU needs for the whole batch.
//   map_sr_box_kernel : per read, gd_sr_box_core (SR/map.c:779-839) on each of its candidates -> its boxes in a slot of its own, their
//                       number and the sum of their lengths (a box is len x len); a read with a degenerate box gives none and is counted
//   (two exclusive scans: first box index and first window offset of every read)
//   map_sr_fill_kernel: per read, its boxes into the dense tables: MapBox (what map_gather_kernel reads), GdCandBox (what the host's
//                       record stage reads), window / CIGAR offsets, band width and exact-match score per box (what the DP planner reads)
struct MapSrTotals { int32_t n_failed, last_failed; };
__global__ __launch_bounds__(64) void map_sr_box_kernel(int n_reads, const int64_t *__restrict__ roff, const MapVoteOut *__restrict__ vo, const uint32_t *__restrict__ seq_len,
                                                        uint32_t n_seq, int k, int a, int slots, int fault_read, GdCandBox *__restrict__ sbox, int32_t *__restrict__ cnt,
                                                        int64_t *__restrict__ sumlen, MapSrTotals *__restrict__ tot)
{
	const int rid = blockIdx.x * blockDim.x + threadIdx.x;
	if (rid >= n_reads) return;
	const uint32_t rl = (uint32_t)(roff[rid + 1] - roff[rid]);
	const unsigned nc = vo[rid].n_cand;
	GdCandBox *out = sbox + (size_t)rid * slots;
	int m = 0;
	int64_t sl = 0;
	bool bad = rid == fault_read && nc > 0;
	for (unsigned j = 0; j < nc && m < slots; ++j) {
		const GdVt v = vo[rid].cand[j];
		GdCandBox b;
		if (!gd_sr_box_core(v, k, a, rl, v.chrom_id < n_seq ? (int32_t)seq_len[v.chrom_id] : 0, b)) continue;
		bad |= b.qlen == 0 || b.tlen == 0 || b.qlen > rl || b.qseq_off + b.qlen > rl || b.tlen > 8u * rl + 100000u;
		out[m++] = b, sl += b.qlen;
	}
	if (bad) { m = 0, sl = 0; atomicAdd(&tot->n_failed, 1), atomicMax(&tot->last_failed, rid); }
	cnt[rid] = m, sumlen[rid] = sl;
}

struct MapSrFillOut { // dense per-box tables, all on the device (the host gets copies of cand, qoff, coff, bw, ex)
	MapBox *boxes;
	GdCandBox *cand;
	int64_t *qoff, *coff; // nb + 1 entries; the window of a box is len bases of the read and len of the reference: toff == qoff
	int32_t *bw, *ex;
};
__global__ __launch_bounds__(64) void map_sr_fill_kernel(int n_reads, const int64_t *__restrict__ roff, const int32_t *__restrict__ cnt, const int32_t *__restrict__ box_first,
                                                         const int64_t *__restrict__ len_first, const GdCandBox *__restrict__ sbox, int slots,
                                                         const uint32_t *__restrict__ seq_len, const uint64_t *__restrict__ seq_off, uint32_t n_seq, GdSrVoteOpt sr, MapSrFillOut O)
{
	const int rid = blockIdx.x * blockDim.x + threadIdx.x;
	if (rid > n_reads) return;
	if (rid == n_reads) { // the closing entries of the offset tables
		O.qoff[box_first[n_reads]] = len_first[n_reads], O.coff[box_first[n_reads]] = 2 * len_first[n_reads];
		return;
	}
	const uint32_t rl = (uint32_t)(roff[rid + 1] - roff[rid]);
	const int m = cnt[rid];
	int64_t off = len_first[rid];
	const int32_t bw = (int32_t)gd_sr_bw((int)rl, sr); // SR/map.c:624-631,925
	for (int j = 0; j < m; ++j) {
		const GdCandBox c = sbox[(size_t)rid * slots + j];
		const int b = box_first[rid] + j;
		MapBox M;
		M.read_off = roff[rid], M.read_len = rl, M.qseq_off = c.qseq_off, M.qlen = c.qlen, M.tlen = c.tlen, M.rev = c.v.str;
		uint32_t avail = 0;
		uint64_t src = 0;
		if (c.target_id < n_seq && c.target_start < seq_len[c.target_id]) { // (a window hanging off its contig: the missing part is zero-filled)
			const uint32_t left = seq_len[c.target_id] - c.target_start;
			avail = c.tlen < left ? c.tlen : left, src = seq_off[c.target_id] + c.target_start;
		}
		M.t_avail = avail, M.t_src = src, M.q_dst = off, M.t_dst = off;
		O.boxes[b] = M, O.cand[b] = c, O.qoff[b] = off, O.coff[b] = 2 * off, O.bw[b] = bw, O.ex[b] = c.exact_score;
		off += c.qlen;
	}
}

// compact the CIGARs of a ksw batch (each sits at the start of a qlen+tlen sized slot) into one contiguous array
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) void map_pack_cigar_kernel(int nb, const uint32_t *__restrict__ cig, const int64_t *__restrict__ coff,
                                                            const int64_t *__restrict__ poff, uint32_t *__restrict__ packed)
{
	__builtin_amdgcn_s_setprio(3); // (see map_post_wave_kernel)
	const int b = blockIdx.x;
	if (b >= nb) return;
	const int64_t n = poff[b + 1] - poff[b];
	for (int64_t i = threadIdx.x; i < n; i += blockDim.x) packed[poff[b] + i] = cig[coff[b] + i];
}

// P1 on the device: mm_fix_cigar + mm_update_extra of every alignment of the batch, one thread per alignment,
// right behind the backtrack.  The CIGAR is rewritten in its slot, n_cigar updated, and the scalars the host needs for the record
// (shifts of qs / qe / rs, mlen, blen, n_ambi, dp_max) land in post[b].  The windows are the ones the DP read (qbuf / tbuf): the
// host then needs neither the 4-bit reference nor the reverse-complemented read for P1.
struct MapPostOpt { int8_t mat[25]; int8_t q, e; int32_t log_gap; };
__global__ __launch_bounds__(64) void map_post_kernel(int nb, const MapBox *__restrict__ boxes, const uint8_t *__restrict__ qbuf, const uint8_t *__restrict__ tbuf,
                                                      const int64_t *__restrict__ coff, uint32_t *__restrict__ cig, int32_t *__restrict__ n_cigar,
                                                      const int32_t *__restrict__ score, MapPostOpt O, GdPostOut *__restrict__ post,
                                                      int32_t *__restrict__ x_score, int32_t *__restrict__ x_ncig /* both given: post, x_score, x_ncig are page-locked HOST memory */)
{
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nb) return;
	GdPostOut P;
	P.qshift = P.tshift = P.mlen = P.blen = P.dp_max = 0, P.n_ambi = 0;
	const int32_t n0 = n_cigar[b];
	if (score[b] != GD_NEG_INF_SCORE_DEV && n0 > 0 && (int64_t)n0 <= coff[b + 1] - coff[b]) { // (an overflowing CIGAR is reported by the host)
		uint32_t n = (uint32_t)n0;
		gdp_update_extra(cig + coff[b], &n, qbuf + boxes[b].q_dst, tbuf + boxes[b].t_dst, O.mat, O.q, O.e, O.log_gap, &P);
		n_cigar[b] = (int32_t)n;
	}
	post[b] = P;
	if (x_score) x_score[b] = score[b], x_ncig[b] = n_cigar[b]; // the batch's results leave with this kernel: no copy afterwards (map_pipeline.hip.h)
}

// ---- wave-parallel form of map_post_kernel: one 64-lane wavefront per alignment ---------------------------------------------------
// One thread per alignment leaves a HiFi batch with 147 wavefronts, each lane in a 15 000-step loop of dependent loads: 22 ms per
// 9 400 alignments, a quarter of a single batch's latency.  Here mm_fix_cigar is a kernel of its own (map_fix_cigar_kernel: a few hundred
// operations per alignment), and the walk over the bases -- mlen / blen / n_ambi and the running score of mm_update_extra -- is done
// by all 64 lanes of a wavefront per alignment.
// The running score  s <- max(s + a_i, 0),  mx <- max(mx, s)  (LR/align.c:286-310; the reference updates mx only where s stays >= 0,
// and not at gaps: the same thing, as s <= mx always holds and mx starts at 0) is a composition of maps  s -> max(s + A, B)  together
// with the largest state seen  max(s + P, Q); for two consecutive segments x, y:
//     A = Ax + Ay,  B = max(Bx + Ay, By),  P = max(Px, Ax + Py),  Q = max(Qx, Bx + Py, Qy)
// which is associative, so a segment of bases can be reduced in any bracketing (every lane four bases, then an ordered tree over the
// lanes).  It is also EXACT: the scores of bases are integers and a gap's penalty q + e * mg_log2(1 + len) -- a float times a small
// integer -- is a multiple of 2^-30, so every sum the reference forms in its double accumulator is exactly representable (|s| < 2^20,
// 50 bits), no addition ever rounds, and the order of additions cannot matter.  The state is kept as integer part + fraction in units of 2^-30.
#define GDP_FRAC 30
struct GdpSeg { int32_t A, B, P, Q; }; // in units of 1 (base scores are integers); B / Q = GDP_NONE: no clamp inside the segment
#define GDP_NONE (-0x20000000)
__device__ __forceinline__ GdpSeg gdp_seg_join(const GdpSeg &x, const GdpSeg &y)
{
	GdpSeg r;
	r.A = x.A + y.A;
	const int32_t bx = x.B == GDP_NONE ? GDP_NONE : x.B + y.A;
	r.B = bx > y.B ? bx : y.B;
	const int32_t p2 = y.P == GDP_NONE ? GDP_NONE : x.A + y.P;
	r.P = x.P > p2 ? x.P : p2;
	const int32_t q2 = (x.B == GDP_NONE || y.P == GDP_NONE) ? GDP_NONE : x.B + y.P;
	int32_t q = x.Q > y.Q ? x.Q : y.Q;
	r.Q = q > q2 ? q : q2;
	return r;
}

// At most 32 VGPRs (checked at build time by __graft_entry__.build through the compiler's resource remarks): the 64-lane DP kernel of the
// next batch in flight holds 5 x 96 of a SIMD's 512 registers, so a wavefront of this size is the largest that can start BESIDE a full
// house of DP wavefronts instead of waiting for one of them to retire (with 52 registers the kernel's 9 400 wavefronts took as long as
// the DP kernel next to them, the batch's results arrived one DP kernel late and the pipelined step went from 88.8 to 97.6 ms).
// first half of P1 for the wave-parallel form: mm_fix_cigar alone, one alignment per thread (a few hundred CIGAR operations each; the few
// wavefronts this needs find room at once).  Leaves the fixed CIGAR and its length in place and the two shifts in post[b].
__global__ __launch_bounds__(64) void map_fix_cigar_kernel(int nb, const MapBox *__restrict__ boxes, const uint8_t *__restrict__ qbuf, const uint8_t *__restrict__ tbuf,
                                                           const int64_t *__restrict__ coff, uint32_t *__restrict__ cig, int32_t *__restrict__ n_cigar,
                                                           const int32_t *__restrict__ score, GdPostOut *__restrict__ post)
{
	__builtin_amdgcn_s_setprio(3); // (see map_post_wave_kernel)
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nb) return;
	GdPostOut P;
	P.qshift = P.tshift = P.mlen = P.blen = P.dp_max = 0, P.n_ambi = 0;
	const int32_t n0 = n_cigar[b];
	if (score[b] != GD_NEG_INF_SCORE_DEV && n0 > 0 && (int64_t)n0 <= coff[b + 1] - coff[b]) {
		uint32_t n = (uint32_t)n0;
		gdp_fix_cigar(cig + coff[b], &n, qbuf + boxes[b].q_dst, tbuf + boxes[b].t_dst, &P.qshift, &P.tshift);
		n_cigar[b] = (int32_t)n;
	}
	post[b] = P;
}

__global__ __launch_bounds__(64) void map_post_wave_kernel(int nb, const MapBox *__restrict__ boxes, const uint8_t *__restrict__ qbuf, const uint8_t *__restrict__ tbuf,
                                                           const int64_t *__restrict__ coff, uint32_t *__restrict__ cig, int32_t *__restrict__ n_cigar,
                                                           const int32_t *__restrict__ score, MapPostOpt O, GdPostOut *__restrict__ post,
                                                           int32_t *__restrict__ x_score, int32_t *__restrict__ x_ncig)
{
#pragma clang fp contract(off)
	// These wavefronts run beside a full house of DP wavefronts of the next batch (older, VALU-bound, five per SIMD): at the default
	// priority they get the issue slots those leave and the kernel's 2 ms of work took 40 ms -- the tail of every batch's latency.
	__builtin_amdgcn_s_setprio(3);
	const int b = blockIdx.x;
	if (b >= nb) return;
	const unsigned lane = threadIdx.x;
	__shared__ int8_t s_mat[32];
	if (lane < 25) s_mat[lane] = O.mat[lane];
	if (lane >= 25 && lane < 32) s_mat[lane] = 0; // index past the matrix (query byte 7 against a target N): taken as 0, as gdp_update_extra does
	GdPostOut P;
	P.qshift = P.tshift = P.mlen = P.blen = P.dp_max = 0, P.n_ambi = 0;
	const int32_t n0 = __builtin_amdgcn_readfirstlane(n_cigar[b]);
	const int32_t sc_b = __builtin_amdgcn_readfirstlane(score[b]);
	const int64_t c0 = coff[b], c1 = coff[b + 1];
	const bool live = sc_b != GD_NEG_INF_SCORE_DEV && n0 > 0 && (int64_t)n0 <= c1 - c0;
	const int32_t n_out = n0;
	if (live) {
		// (everything about the alignment as a whole is wave-uniform and is kept on the scalar unit: the per-lane state is a segment, two
		// counters and the loaded bytes)
		uint32_t *cg = (uint32_t *)gdw_uniform_ptr((const uint8_t *)(cig + c0), 0);
		const uint8_t *qseq = gdw_uniform_ptr(qbuf + boxes[b].q_dst, 0), *tseq = gdw_uniform_ptr(tbuf + boxes[b].t_dst, 0);
		const uint32_t n = (uint32_t)n0; // (mm_fix_cigar has run: map_fix_cigar_kernel)
		P.qshift = __builtin_amdgcn_readfirstlane(post[b].qshift), P.tshift = __builtin_amdgcn_readfirstlane(post[b].tshift);
		__syncthreads(); // (s_mat)
		qseq += P.qshift, tseq += P.tshift;
		// the running score S and its maximum MX as (integer part, fraction in units of 2^-GDP_FRAC, 0 <= fraction < 2^GDP_FRAC): base
		// scores are integers, only gap penalties carry a fraction -- all scalar 32-bit arithmetic, nothing of it in vector registers
		int32_t Si = 0, Sf = 0, MXi = 0, MXf = 0;
		int32_t qoff = 0, toff = 0, blen = 0, mlen = 0;
		uint32_t n_ambi_tot = 0;
		for (uint32_t k = 0; k < n; ++k) {
			const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cg[k]), op = c & 0xf, len = c >> 4;
			if (op == 0) {
				for (uint32_t base = 0; base < len; base += 64) { // one base per lane, then an ordered tree over the lanes
					GdpSeg g = {0, GDP_NONE, GDP_NONE, GDP_NONE};
					int cnt2 = 0; // n_ambi | n_diff << 16
					const uint32_t l = base + lane;
					if (l < len) {
						const int cq = qseq[qoff + l], ct = tseq[toff + l];
						if (ct > 3 || cq > 3) cnt2 = 1;
						else if (ct != cq) cnt2 = 1 << 16;
						const int idx = ct * 5 + cq;
						const int32_t a = idx < 25 ? (int32_t)s_mat[idx] : 0;
						g.A = a, g.B = 0, g.P = a, g.Q = 0; // s -> max(s + a, 0); the state after it is max(s + a, 0) as well
					}
#pragma unroll 1 // (not unrolled: the kernel must stay within 32 VGPRs, see above)
					for (int d = 1; d < 64; d <<= 1) { // lane i: segment [i, i + 2d) = its own [i, i + d) followed by lane i + d's
						GdpSeg o;
						o.A = __shfl_down(g.A, d), o.B = __shfl_down(g.B, d), o.P = __shfl_down(g.P, d), o.Q = __shfl_down(g.Q, d);
						const int oc = __shfl_down(cnt2, d);
						if (lane + d < 64) g = gdp_seg_join(g, o), cnt2 += oc;
					}
					const int32_t A = __builtin_amdgcn_readfirstlane(g.A), Bc = __builtin_amdgcn_readfirstlane(g.B), Pm = __builtin_amdgcn_readfirstlane(g.P),
					              Qm = __builtin_amdgcn_readfirstlane(g.Q);
					const int c2 = __builtin_amdgcn_readfirstlane(cnt2), na = c2 & 0xffff, nd = c2 >> 16;
					if (Pm != GDP_NONE && (Si + Pm > MXi || (Si + Pm == MXi && Sf > MXf))) MXi = Si + Pm, MXf = Sf; // MX = max(MX, S + P)
					if (Qm != GDP_NONE && (Qm > MXi)) MXi = Qm, MXf = 0;                                              // MX = max(MX, Q)
					Si += A;
					if (Bc != GDP_NONE && Si < Bc) Si = Bc, Sf = 0; // S = max(S + A, B): with 0 <= fraction < 1, S + A < B iff its integer part is
					const uint32_t cnt = len - base < 64 ? len - base : 64;
					blen += (int32_t)cnt - na, mlen += (int32_t)cnt - (na + nd), n_ambi_tot += (uint32_t)na;
				}
				toff += len, qoff += len;
			} else if (op == 1 || op == 2) {
				int na = 0; // (counted with ballots: wave-uniform, no vector registers for a butterfly)
				const uint8_t *src = op == 1 ? qseq + qoff : tseq + toff;
				for (uint32_t l0 = 0; l0 < len; l0 += 64) na += __popcll(__ballot(l0 + lane < len && src[l0 + lane] > 3));
				blen += (int32_t)len - na, n_ambi_tot += (uint32_t)na;
				double tot;
				if (O.log_gap) {
					const double pen = (double)O.e * (double)gdp_mg_log2(1.0f + (float)len);
					tot = (double)O.q + pen;
				} else tot = (double)(O.q + O.e);
				// tot >= 0 is a multiple of 2^-30 below 2^12: integer part and fraction, both exact
				const int32_t Ti = __builtin_amdgcn_readfirstlane((int32_t)tot);
				const int32_t Tf = __builtin_amdgcn_readfirstlane((int32_t)((tot - (double)Ti) * (double)(1 << GDP_FRAC)));
				Sf -= Tf, Si -= Ti;
				if (Sf < 0) Sf += 1 << GDP_FRAC, Si -= 1;
				if (Si < 0) Si = 0, Sf = 0; // (s < 0 iff its integer part is)
				if (op == 1) qoff += len; else toff += len;
			} else if (op == 3) toff += len;
		}
		P.mlen = mlen, P.blen = blen, P.n_ambi = n_ambi_tot;
		P.dp_max = (int32_t)((double)MXi + (double)MXf / (double)(1 << GDP_FRAC) + .499);
	}
	if (lane == 0) {
		post[b] = P;
		if (x_score) x_score[b] = sc_b, x_ncig[b] = n_out;
	}
}

// ---- wave-parallel form of map_seed_kernel: one 64-lane wavefront per read -------------------------------------------
// The winnowing automaton is sequential, but exact slices of it can be produced independently (gd_sketch_slice), so the 64
// lanes sketch 64 slices of the read, compact their minimizers in read order with a wavefront prefix sum, probe the index
// in parallel, and leave only the short sequential parts (query-occurrence filter, high-occurrence seed selection) to lane 0.
#define MAP_SORT_CAP 2048      // hashes of one read the seed kernel sorts in LDS: at least (16 KB) ...
#define MAP_SORT_CAP_MAX 16384 // ... and at most (128 KB: ONT reads of up to ~180 kbp), chosen per batch from its longest read

#ifdef GD_SEED_PROF
__device__ unsigned long long gd_seed_prof[8];
#define GD_PROF_T(i) do { const unsigned long long t_ = wall_clock64(); if (lane == 0) atomicAdd(&gd_seed_prof[i], t_ - prof_t); prof_t = t_; } while (0)
#else
#define GD_PROF_T(i) do { } while (0)
#endif
struct GdEmitLane { // per-lane emission list in scratch
	GdMini *out;
	unsigned n, cap;
	__device__ bool operator()(const GdMini &m)
	{
		if (n < cap) out[n] = m;
		++n;
		return false;
	}
};

// parallel sketch of `dl` sparsified bases; the first min(total, cap) minimizers in read order go to dst[0..).
// Returns (uniform) the count, or ~0u when a scratch list or dst overflowed.
// the steps [a, b) of a sketch over dl sparsified bases, 64 exact slices; their minimizers are appended in read order to dst[have..)
// as long as the list stays below `cap` (0: no cap).  Returns (uniform) the new length, or ~0u when a scratch list or dst overflowed.
__device__ __forceinline__ unsigned map_par_sketch_range(const uint8_t *str, unsigned dl, unsigned a, unsigned b, int w, int k, unsigned shift, const GdPattern &P,
                                                         GdMini *tmp, unsigned R, GdMini *dst, unsigned dst_cap, uint32_t cap, unsigned have, GdMini *win)
{
	const unsigned lane = threadIdx.x & 63;
	const unsigned chunk = (b - a + 63) / 64;
	const unsigned i0 = a + lane * chunk, i1 = i0 + chunk < b ? i0 + chunk : b;
	GdEmitLane e = {tmp + (size_t)lane * R, 0, R};
#ifdef GD_SEED_PROF
	unsigned long long prof_t = wall_clock64();
#endif
	if (chunk && i0 < b) gd_sketch_slice<GdEmitLane, true>(str, dl, i0, i1, w, k, 0, shift, P, true, e, win + lane, 64);
	GD_PROF_T(6);
	unsigned incl = e.n;
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned v = __shfl_up(incl, d);
		if ((int)lane >= d) incl += v;
	}
	const unsigned total = have + __shfl(incl, 63), off = have + incl - e.n;
	const unsigned eff = (cap > 0 && total >= cap) ? cap : total;
	bool bad = e.n > R;
	for (unsigned j = 0; j < e.n && !bad; ++j) {
		const unsigned p = off + j;
		if (p >= eff) break;
		if (p < dst_cap) dst[p] = e.out[j];
		else bad = true;
	}
	GD_PROF_T(7);
	return __any(bad) ? ~0u : eff;
}

// parallel sketch of `dl` sparsified bases; the first min(total, cap) minimizers in read order go to dst[0..).
// Returns (uniform) the count, or ~0u on overflow.  With a cap the sequential reference stops at the cap-th minimizer; slices
// cannot stop each other, so a prefix that should hold it (the expected 2 / (w + 1) minimizers per base, 25 % margin) is sketched
// first and the rest of the read only if the prefix fell short -- the slices being exact, the first `cap` minimizers are the same
// either way.  (Phase >= 1 of mm_sketch2 is capped at phase 0's count, found in the first 20 % of a HiFi read: without this the
// capped pass cost as much as an uncapped one.)
__device__ __forceinline__ unsigned map_par_sketch(const uint8_t *str, unsigned dl, int w, int k, unsigned shift, const GdPattern &P, GdMini *tmp,
                                                   unsigned R, GdMini *dst, unsigned dst_cap, uint32_t cap, GdMini *win /* LDS: w x 64 entries, lane-interleaved */)
{
	if (cap == 0) return map_par_sketch_range(str, dl, 0, dl, w, k, shift, P, tmp, R, dst, dst_cap, 0, 0, win);
	uint64_t l1 = (uint64_t)cap * (uint64_t)(w + 1) / 2;
	l1 += l1 / 4 + (unsigned)(w + k);
	if (l1 >= dl) return map_par_sketch_range(str, dl, 0, dl, w, k, shift, P, tmp, R, dst, dst_cap, cap, 0, win);
	const unsigned n1 = map_par_sketch_range(str, dl, 0, (unsigned)l1, w, k, shift, P, tmp, R, dst, dst_cap, cap, 0, win);
	if (n1 == ~0u || n1 >= cap) return n1;
	__syncthreads(); // (the lanes' scratch lists and LDS windows are reused)
	return map_par_sketch_range(str, dl, (unsigned)l1, dl, w, k, shift, P, tmp, R, dst, dst_cap, cap, n1, win);
}

__global__ __launch_bounds__(64) void map_seed_wave_kernel(int n_reads, const uint8_t *__restrict__ reads, const int64_t *__restrict__ roff,
                                                           GdIdxView I, MapDevOpt O, const MapReadScratch *__restrict__ sc, GdMini *__restrict__ mv_arena,
                                                           uint64_t *__restrict__ u64_arena, GdSeed *__restrict__ seed_arena, MapSeedOut *__restrict__ out,
                                                           const int32_t *__restrict__ ids /* the reads of this launch (null: 0 .. n_reads - 1) */)
{
	map_set_prio(O.prio);
	if ((int)blockIdx.x >= n_reads) return;
	const int rid = ids ? ids[blockIdx.x] : (int)blockIdx.x;
	const unsigned lane = threadIdx.x;
	const uint8_t *str = reads + roff[rid];
	const int len = (int)(roff[rid + 1] - roff[rid]);
	MapSeedOut o;
	o.n_seeds = 0, o.shift = 0, o.tel = (uint32_t)len, o.n_mv = 0, o.n_a = 0;
	if (len <= 0) { if (lane == 0) out[rid] = o; return; }
	const MapReadScratch S = sc[rid];
	GdMini *mv = mv_arena + S.mv_off;
	GdMini *tmp = (GdMini *)(u64_arena + S.u64_off); // 2*mv_cap uint64 = mv_cap GdMini
	GdSeed *seeds = seed_arena + S.seed_off;
	const unsigned R = S.mv_cap / 64;
	// dynamic LDS: the winnowing windows of the 64 lanes (w x 64 entries, lane-interleaved), reused by the hash sort of S4
	extern __shared__ __attribute__((aligned(16))) uint8_t seed_lds[];
	GdMini *win = reinterpret_cast<GdMini *>(seed_lds);
#ifdef GD_SEED_PROF
	unsigned long long prof_t = wall_clock64();
#endif
	// S1: mm_sketch2 -- every phase, phase 0 on the cropped read, later phases capped at phase 0's count (LR/sketch.c:2174-2223)
	unsigned len_crop, total = 0;
	uint32_t cap;
	if (O.max_seeds < 1) len_crop = (unsigned)((float)O.max_seeds * len), cap = UINT32_MAX;
	else len_crop = (unsigned)len, cap = (uint32_t)O.max_seeds;
	uint32_t shift_n[64];
	bool bad = false;
	for (int shift = 0; shift < O.pat.W; ++shift) {
		const unsigned dl = gd_diet_len(O.pat, len_crop, (unsigned)shift);
		const unsigned n = map_par_sketch(str, dl, O.w, O.k, (unsigned)shift, O.pat, tmp, R, mv + total, S.mv_cap - total, cap == UINT32_MAX ? 0u : cap, win);
		__syncthreads();
		if (n == ~0u) { bad = true; break; }
		shift_n[shift] = n, total += n;
		if (cap == UINT32_MAX) len_crop = (unsigned)len, cap = n;
	}
	if (bad) { o.n_seeds = -1; if (lane == 0) out[rid] = o; return; }
	GD_PROF_T(0);
	// S3: mm_get_shift -- probes in parallel, one sum per phase
	{
		unsigned best = 0, base = 0;
		for (int i = 0; i < O.pat.W; ++i) {
			unsigned cur = 0;
			for (unsigned j = lane; j < shift_n[i]; j += 64) {
				uint64_t st;
				cur += gd_idx_get(I, mv[base + j].x >> 8, &st);
			}
			for (int d = 32; d > 0; d >>= 1) cur += __shfl_xor(cur, d);
			if (cur > best) o.shift = i, best = cur;
			base += shift_n[i];
		}
	}
	__syncthreads();
	GD_PROF_T(1);
	// S2: mm_sketch3 at the chosen phase
	unsigned n_mv;
	{
		const unsigned dl = gd_diet_len(O.pat, (unsigned)len, (unsigned)o.shift);
		n_mv = map_par_sketch(str, dl, O.w, O.k, (unsigned)o.shift, O.pat, tmp, R, mv, S.mv_cap, O.max_nb_seeds == UINT32_MAX ? 0u : O.max_nb_seeds, win);
		__syncthreads();
		if (n_mv == ~0u) { o.n_seeds = -1; if (lane == 0) out[rid] = o; return; }
		if (O.max_nb_seeds != UINT32_MAX && O.max_nb_seeds > 0 && n_mv == O.max_nb_seeds) o.tel = (uint32_t)(mv[n_mv - 1].y >> 1); // :2010-2012
	}
	GD_PROF_T(2);
	// S4: mm_seed_mz_flt.  It only ever drops something when one hash occurs more than mid_occ times in the read, which a
	// wavefront bitonic sort of the hashes in LDS decides in a few microseconds (a run longer than mid_occ <=> s[i] == s[i + mid_occ]
	// for some i); only then -- practically never -- does lane 0 run the sequential filter.  Lists too long for the LDS buffer
	// (O.sort_cap, sized by the host for the longest read of the batch, at most MAP_SORT_CAP_MAX) take the sequential path directly.
	if (O.q_occ_frac > 0.0f && (int64_t)n_mv > (int64_t)O.mid_occ && O.mid_occ > 0) {
		uint64_t *srt = reinterpret_cast<uint64_t *>(seed_lds); // the window storage is idle now
		bool need = true;
		{ // cheap exact pre-check: count the hashes in buckets (LDS atomics); no bucket above mid_occ => no hash above mid_occ => nothing to drop
			const unsigned nb = (unsigned)(O.seed_lds / sizeof(uint32_t)) >= 8192u ? 8192u : 4096u; // (the launch provides at least 16 KB)
			uint32_t *cnt = reinterpret_cast<uint32_t *>(seed_lds);
			for (unsigned i = lane; i < nb; i += 64) cnt[i] = 0;
			__syncthreads();
			for (unsigned i = lane; i < n_mv; i += 64) atomicAdd(&cnt[(unsigned)((mv[i].x >> 8) * 0x9E3779B97F4A7C15ull >> 40) & (nb - 1)], 1u);
			__syncthreads();
			bool over = false;
			for (unsigned i = lane; i < nb; i += 64) over |= cnt[i] > (uint32_t)O.mid_occ;
			need = __any(over);
			__syncthreads();
		}
		if (need && n_mv <= (unsigned)O.sort_cap) {
			unsigned P2 = 64;
			while (P2 < n_mv) P2 <<= 1;
			for (unsigned i = lane; i < P2; i += 64) srt[i] = i < n_mv ? mv[i].x : UINT64_MAX;
			__syncthreads();
			for (unsigned kk = 2; kk <= P2; kk <<= 1)
				for (unsigned jj = kk >> 1; jj > 0; jj >>= 1) {
					for (unsigned i = lane; i < P2; i += 64) {
						const unsigned ixj = i ^ jj;
						if (ixj > i) {
							const uint64_t a = srt[i], b = srt[ixj];
							const bool up = (i & kk) == 0;
							if ((a > b) == up) srt[i] = b, srt[ixj] = a;
						}
					}
					__syncthreads();
				}
			bool hit = false;
			for (unsigned i = lane; i + (unsigned)O.mid_occ < n_mv; i += 64) hit |= srt[i] == srt[i + (unsigned)O.mid_occ];
			need = __any(hit);
		}
		if (need) {
			unsigned nn = 0;
			if (lane == 0) nn = gd_mz_flt(mv, n_mv, O.mid_occ, O.q_occ_frac, u64_arena + S.u64_off);
			__syncthreads();
			n_mv = __shfl(nn, 0);
		}
	}
	GD_PROF_T(3);
	// S5: probes in parallel, then the sequential selection (mm_seed_select's streak logic and the two compactions: three passes of
	// lane 0 over the seeds, each a dependent load-store chain).  The seeds live in LDS for it when they fit (the sort buffer is idle
	// by now): ~30 ns per access instead of the ~500 ns of a round trip to global memory -- that chain was most of the kernel's
	// latency (4.5 k seeds of a 50 kbp ONT read: ~7 ms).  The kept seeds go to their global array in one parallel copy.
	const bool in_lds = (size_t)n_mv * sizeof(GdSeed) <= (size_t)O.seed_lds;
	GdSeed *work = in_lds ? reinterpret_cast<GdSeed *>(seed_lds) : seeds;
	for (unsigned j = lane; j < n_mv; j += 64) gd_collect_probe(I, mv[j], work[j]);
	__syncthreads();
	GD_PROF_T(4);
	// gd_collect_finish with its two compactions done by the whole wavefront (ballot + prefix count, in place: a chunk is read into
	// registers before anything of it is overwritten, and writes never pass the chunk's own positions); only mm_seed_select's streak
	// logic stays on lane 0, and only for reads that have a seed above max_occ at all
	int n_m0 = 0;
	bool any_high = false;
	for (unsigned b0 = 0; b0 < n_mv; b0 += 64) { // drop the minimizers absent from the index
		const unsigned j = b0 + lane;
		GdSeed sd;
		sd.n = 0;
		if (j < n_mv) sd = work[j];
		const bool keep = j < n_mv && sd.n != 0;
		const uint64_t m = __ballot(keep);
		any_high |= keep && (int32_t)sd.n > O.mid_occ;
		__syncthreads();
		if (keep) work[n_m0 + __popcll(m & ((1ull << lane) - 1))] = sd;
		n_m0 += __popcll(m);
		__syncthreads();
	}
	if (__any(any_high)) {
		if (O.occ_dist > 0 && O.max_max_occ > O.mid_occ) {
			if (lane == 0) gd_seed_select(n_m0, work, len, O.mid_occ, O.max_max_occ, O.occ_dist);
		} else
			for (int j = (int)lane; j < n_m0; j += 64)
				if ((int32_t)work[j].n > O.mid_occ) work[j].flt = 1;
		__syncthreads();
	}
	int n_kept = 0;
	int64_t n_a = 0;
	for (int b0 = 0; b0 < n_m0; b0 += 64) { // keep the unfiltered, count their occurrences
		const int j = b0 + (int)lane;
		GdSeed sd;
		sd.n = 0, sd.flt = 1;
		if (j < n_m0) sd = work[j];
		const bool keep = j < n_m0 && !sd.flt;
		const uint64_t m = __ballot(keep);
		if (keep) n_a += sd.n;
		__syncthreads();
		if (keep) work[n_kept + __popcll(m & ((1ull << lane) - 1))] = sd;
		n_kept += __popcll(m);
		__syncthreads();
	}
	for (int d = 32; d > 0; d >>= 1) n_a += __shfl_xor(n_a, d);
	if (lane == 0) {
		o.n_mv = n_mv, o.n_seeds = n_kept, o.n_a = n_a;
		out[rid] = o;
	}
	if (in_lds)
		for (int j = (int)lane; j < n_kept; j += 64) seeds[j] = work[j];
	GD_PROF_T(5);
}
