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

struct MapDevOpt { // uniform per batch
	int32_t k, w;
	float max_seeds, q_occ_frac;
	int32_t mid_occ, max_max_occ, occ_dist;
	uint32_t max_nb_seeds; // cap of mm_sketch3 (UINT32_MAX unless frag mode)
	int64_t flag;
	GdLrVoteOpt vote;
	GdPattern pat;
};

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
	uint32_t pad;
	int64_t n_a;      // total occurrences of the kept seeds
};

__global__ __launch_bounds__(64) void map_seed_kernel(int n_reads, const uint8_t *__restrict__ reads, const int64_t *__restrict__ roff,
                                                      GdIdxView I, MapDevOpt O, const MapReadScratch *__restrict__ sc, GdMini *__restrict__ mv_arena,
                                                      uint64_t *__restrict__ u64_arena, GdSeed *__restrict__ seed_arena, MapSeedOut *__restrict__ out)
{
	const int rid = blockIdx.x * blockDim.x + threadIdx.x;
	if (rid >= n_reads) return;
	const uint8_t *str = reads + roff[rid];
	const int len = (int)(roff[rid + 1] - roff[rid]);
	MapSeedOut o;
	o.n_seeds = 0, o.shift = 0, o.tel = (uint32_t)len, o.pad = 0, o.n_a = 0;
	if (len <= 0) { out[rid] = o; return; }
	const MapReadScratch S = sc[rid];
	GdMini *mv = mv_arena + S.mv_off;
	uint32_t shift_n[64];
	const unsigned tot = gd_sketch2(str, len, O.w, O.k, O.pat, O.max_seeds, mv, S.mv_cap, shift_n);
	if (tot == ~0u) { o.n_seeds = -1; out[rid] = o; return; }
	o.shift = (int32_t)gd_get_shift(I, mv, shift_n, O.pat.W);
	unsigned n_mv = 0;
	o.tel = gd_sketch3(str, (unsigned)len, O.w, O.k, O.pat, o.shift, O.max_nb_seeds, mv, S.mv_cap, &n_mv);
	if (n_mv == ~0u) { o.n_seeds = -1; out[rid] = o; return; }
	if (O.q_occ_frac > 0.0f) n_mv = gd_mz_flt(mv, n_mv, O.mid_occ, O.q_occ_frac, u64_arena + S.u64_off);
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
                                                      GdLoc *__restrict__ hits /* 3 slices per read: for, rev, tmp */, MapVoteOut *__restrict__ out)
{
	const int rid = blockIdx.x * blockDim.x + threadIdx.x;
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
	o.n_cand = gd_lr_candidates(a_for, nf, sr, nr, (uint32_t)len, (int32_t)so.tel, O.vote, o.cand);
}

// one DP box: where its query / target windows come from and where they go in the packed ksw batch buffers
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

// compact the CIGARs of a ksw batch (each sits at the start of a qlen+tlen sized slot) into one contiguous array
__global__ __launch_bounds__(64) void map_pack_cigar_kernel(int nb, const uint32_t *__restrict__ cig, const int64_t *__restrict__ coff,
                                                            const int64_t *__restrict__ poff, uint32_t *__restrict__ packed)
{
	const int b = blockIdx.x;
	if (b >= nb) return;
	const int64_t n = poff[b + 1] - poff[b];
	for (int64_t i = threadIdx.x; i < n; i += blockDim.x) packed[poff[b] + i] = cig[coff[b] + i];
}
