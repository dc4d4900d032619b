// Host lock-step emulator of the register-resident ksw_extd2 wavefront kernel (test infrastructure).
// It drives genome-on-diet_amd/csrc/ksw_wave_core.h -- the very per-lane code the GPU kernel compiles -- with 64
// (or 16) emulated lanes, exchanging registers between lanes where the GPU uses DPP, and compares score and CIGAR
// with the CPU oracle (oracle/gdo_ksw2.c) on seeded random pairs.  Exit code 0 = all pairs identical.
//
//   g++ -O2 -I genome-on-diet_amd/csrc -I oracle tests/emul/wave_emul.cpp oracle/gdo_ksw2.c -o wave_emul
//   ./wave_emul <seed> <n_pairs> <lanes>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>
#define __host__
#define __device__
#include "ksw_wave_core.h"
#include "gdo_ksw2.h"

struct EmuResult { int score; std::vector<uint32_t> cigar; };

// lock-step emulation of ksw_extd2_wave_kernel<LANES>; mirrors the device row loop statement by statement
static long g_fast_chunks = 0; // chunks of the half-block cone that took the interior fast path
static bool g_single = false; // argv[4] == "single": the single-affine (ksw_extz2) form of the kernel against the extz2 oracle

static EmuResult emulate(int LANES, const uint8_t *query, int qlen, const uint8_t *target, int tlen, int w, const KswConst &C)
{
	WaveK K;
	if (!gdw_make_consts(C, K)) { fprintf(stderr, "consts rejected\n"); exit(2); }
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;
	std::vector<WaveLane> L(LANES);
	std::vector<uint8_t> bt((size_t)(rend + 1) * LANES * 16, 0xEE);
	for (int l = 0; l < LANES; ++l) gdw_load_block(L[l], K, l, 0, query, qlen, target, tlen);
	int prev_st_ = 0, prev_st0 = -1, prev_up = -1, prev_en0 = -1, have_f = 0, Rf = 0;
	int rA, rS; // the paired rows in the middle of a long alignment (the device's pair_row): band limits, selectors and the events of a row from m alone
	gdw_steady_rows(qlen, tlen, w, rA, rS);
	const int nblkA = (w - 1 + 16) >> 4, nblkB = (w + 16) >> 4;
	for (int r = 0; r <= rend; ++r) {
		WaveRow W;
		W.r = r;
		gd_band(r, qlen, tlen, w, W.st0, W.en0);
		if (W.st0 > W.en0) { fprintf(stderr, "empty band in wave kernel\n"); exit(2); }
		W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
		W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
		int advanced = W.st_ > prev_st_;
		W.use_array = advanced;
		W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open;
		W.set_tr = (W.en0 | 15) >= r;
		W.ukey = gdw_edge_key(K, r);
		const bool paired = r >= rA && r < rS, row_a = paired && ((r - rA) & 1) == 0;
		if (paired) { // what pair_row derives from m must be what the generic row computes
			const int m = (r - w + 1) >> 1;
			WaveRow P;
			P.r = r, P.st0 = m, P.en0 = row_a ? m + w - 1 : m + w, P.st_ = m >> 4, P.en_ = P.en0 >> 4, P.up = m + ((row_a ? nblkA : nblkB) << 4);
			const int adv = row_a && (m & 15) == 0;
			P.use_array = adv, P.v1key = K.key_open, P.set_tr = 0, P.ukey = 0;
			if (((r - w + 1) & 1) != (row_a ? 0 : 1) || P.st0 != W.st0 || P.en0 != W.en0 || P.up != W.up || adv != advanced || P.st_ - adv != prev_st_ || W.set_tr || W.v1key != K.key_open ||
			    W.en0 == tlen - 1 || (row_a && W.en0 != prev_en0) || (!row_a && W.st0 != prev_st0)) { fprintf(stderr, "paired row %d: band bookkeeping differs\n", r); exit(2); }
			W = P, advanced = adv;
		}
		// (1) cross-lane exchange of row r-1 values (DPP wave_ror:1 on the GPU), before anything is modified
		std::vector<u32> pX(LANES), pV(LANES), pX2(LANES), pQ(LANES);
		for (int l = 0; l < LANES; ++l) {
			const int p = (l + LANES - 1) % LANES;
			pX[l] = L[p].X[7], pV[l] = L[p].V[7], pX2[l] = L[p].X2[7], pQ[l] = L[p].Qc[3];
		}
		// groups of 10 / 8 lanes (several alignments per wavefront, no ring): the first lane of a group receives another
		// alignment's values on the GPU -- whatever they are, they must not matter
		if (LANES == 10 || LANES == 8) pX[0] = 0xDEADBEEFu ^ (u32)r, pV[0] = 0xBADC0FFEu + (u32)r, pX2[0] = 0x13579BDFu * (u32)(r + 1), pQ[0] = 0xA5A5A5A5u;
		// (2) query window advance / block retirement
		int reloaded = 0;
		for (int l = 0; l < LANES; ++l) {
			if (r > 0) gdw_shift_query(L[l], pQ[l], L[l].blk == prev_st_, gdw_qbyte(query, qlen, r - (prev_st_ << 4)));
			if (L[l].blk < W.st_) gdw_load_block(L[l], K, L[l].blk + LANES, r, query, qlen, target, tlen), reloaded = 1;
		}
		// (3) per-row scalar fix-ups and the score row
		const int remask = W.st0 != prev_st0 || W.up != prev_up || reloaded;
		bool any_tn = false; // the device takes this from a wavefront ballot
		for (int l = 0; l < LANES; ++l) any_tn |= L[l].tn != 0;
		for (int l = 0; l < LANES; ++l) {
			if (W.set_tr) gdw_reset_tr(L[l], K, W);
			if (paired) {
				if (row_a || nblkA != nblkB) {
					u32 lo[4], hi[4], want[4];
					gdw_sel_uniform(W.st0 & 15, lo, hi);
					u32 m_lo;
					gdw_pick_sel(L[l], W.st_, W.up >> 4, (W.up >> 4) - W.st_ >= LANES - 1, lo, hi, m_lo);
					if (m_lo != (L[l].blk == W.st_ ? ~0u : 0u)) { fprintf(stderr, "lane mask\n"); exit(2); }
					memcpy(want, L[l].SEL, 16);
					gdw_make_sel(L[l], W.st0, W.up);
					if (memcmp(want, L[l].SEL, 16)) { fprintf(stderr, "paired row %d lane %d: selectors differ\n", r, l); exit(2); }
				} else if (remask) { fprintf(stderr, "paired row %d: selectors change on a second row\n", r); exit(2); }
			} else if (remask) gdw_make_sel(L[l], W.st0, W.up);
			gdw_update_scores(L[l], K, any_tn);
		}
		// (4) the DP cells of the active lanes
		for (int l = 0; l < LANES; ++l)
			if (L[l].blk <= W.en_) {
				u32 out[4];
				if (g_single) gdw_compute<false>(L[l], K, W, pX[l], pV[l], pX2[l], out);
				else gdw_compute<true>(L[l], K, W, pX[l], pV[l], pX2[l], out);
				memcpy(&bt[((size_t)r * LANES + l) * 16], out, 16);
			}
		// (5) score trackers
		if (r == 0) L[0].R = gdw_lo(L[0].V[0]) - K.B1 - K.qe8;
		else for (int l = 0; l < LANES; ++l) L[l].R += gdw_lo(L[l].V[0]) - K.B1;
		if (paired && row_a && W.en0 != prev_en0) { fprintf(stderr, "paired row %d: en0 moved on a first row\n", r); exit(2); }
		if (r > 0 && W.en0 != prev_en0 && (W.en0 & 15) == 0) {
			const int m = W.en0 >> 4;
			const int h = gdw_track_handoff(L[(m - 1) % LANES]);
			L[m % LANES].R = h + gdw_lo(L[m % LANES].U[0]);
		}
		if (W.en0 == tlen - 1) {
			WaveLane &F = L[mlast % LANES];
			if (!have_f) Rf = gdw_track_to_slot(F, sl), have_f = 1;
			else Rf += gdw_cell(F.V, sl) - K.B1;
		}
		prev_st_ = W.st_, prev_st0 = W.st0, prev_up = W.up, prev_en0 = W.en0;
	}
	EmuResult res;
	if (Rf % 8) { fprintf(stderr, "tracker not a multiple of 8\n"); exit(2); }
	res.score = Rf / 8;
	// backtrack: convert the wave layout to the reference layout and run the oracle's backtrack on it
	const int ncol = gd_ncol16(qlen, tlen, w);
	std::vector<uint8_t> p((size_t)(rend + 1) * ncol * 16 + 16, 0);
	std::vector<int> off(2 * (rend + 1));
	for (int r = 0; r <= rend; ++r) {
		int st0, en0;
		gd_band(r, qlen, tlen, w, st0, en0);
		const int st = st0 & ~15, en = en0 | 15;
		off[r] = st, off[rend + 1 + r] = en;
		for (int i = st; i <= en; ++i) {
			const int c = i & 15, g = (c & 7) >> 1, h = (c & 1) | ((c >> 3) << 1);
			const uint8_t b = bt[((size_t)r * LANES + ((i >> 4) % LANES)) * 16 + 4 * g + h];
			const uint8_t nb = (uint8_t)~b; // (4-d) | nY2<<4 | nX2<<5 | nY<<6 | nX<<7 -> d | cX<<3 | cY<<4 | cX2<<5 | cY2<<6
			const uint8_t ref = (uint8_t)((4 - (b & 7)) | ((nb >> 4) & 0x08) | ((nb >> 2) & 0x10) | (nb & 0x20) | ((nb << 2) & 0x40));
			p[(size_t)r * ncol * 16 + (i - st)] = ref;
		}
	}
	int m_cigar = 0, n_cigar = 0;
	uint32_t *cigar = 0;
	gdo_backtrack(0, 0, p.data(), off.data(), off.data() + rend + 1, ncol * 16, tlen - 1, qlen - 1, &m_cigar, &n_cigar, &cigar);
	res.cigar.assign(cigar, cigar + n_cigar);
	free(cigar);
	return res;
}

// ---- the checkpointed wide-band form (ksw_extd2_wave128c_kernel): pass 1 = the 128-position ring without a backtrace, a snapshot
// of all positions every CK rows; pass 2 = per chunk, from the last one down, the CONE of the walk recomputed by 64 lanes holding the
// blocks [b0, b0 + 63] (gdw_cone_restore / gdw_cone_row on the device) and consumed by a resumable walk.  A cell the walk reads
// that the cone pass did not compute is an error.
// pass 1 on the 96-block ring (ksw_extd2_wave96c_kernel / gdw96_row, statement by statement): 64 lanes, each one block (F) and one
// half block (H); snapshots in the 128-position record format (gdw96_save)
static int pass1_ring96(const uint8_t *query, int qlen, const uint8_t *target, int tlen, int w, const WaveK &K, int CK, std::vector<std::vector<WaveLane>> &snap)
{
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;
	std::vector<WaveLane> F(64);
	std::vector<WaveHalf> H(64);
	for (int l = 0; l < 64; ++l) gdw_load_block(F[l], K, l, 0, query, qlen, target, tlen), gdw_load_half(H[l], K, 64 + (l >> 1), l & 1, 0, query, qlen, target, tlen);
	int prev_st_ = 0, prev_st0 = -1, prev_up = -1, prev_en0 = -1, have_f = 0, Rf = 0;
	auto snapshot = [&](const int r) {
		if (r % CK == 0) { // gdw96_save
			std::vector<WaveLane> rec(128);
			for (auto &x : rec) memset(&x, 0xEE, sizeof(x)); // (records nobody writes hold whatever was there)
			for (int l = 0; l < 64; ++l) {
				rec[F[l].blk & 127] = F[l];
				WaveLane &d = rec[H[l].blk & 127];
				const int h = H[l].half;
				for (int k = 0; k < 8; ++k) {
					const u32 m = h ? 0x0000ffffu : 0xffff0000u;
					const int sh = h ? 16 : 0;
					d.U[k] = (d.U[k] & m) | gdw_half_cell16(H[l].U, k) << sh, d.V[k] = (d.V[k] & m) | gdw_half_cell16(H[l].V, k) << sh;
					d.X[k] = (d.X[k] & m) | gdw_half_cell16(H[l].X, k) << sh, d.Y[k] = (d.Y[k] & m) | gdw_half_cell16(H[l].Y, k) << sh;
					d.X2[k] = (d.X2[k] & m) | gdw_half_cell16(H[l].X2, k) << sh, d.Y2[k] = (d.Y2[k] & m) | gdw_half_cell16(H[l].Y2, k) << sh;
				}
				for (int g = 0; g < 2; ++g) d.Sb[2 * h + g] = H[l].Sb[g], d.Tb[2 * h + g] = H[l].Tb[g], d.Qc[2 * h + g] = H[l].Qc[g], d.SEL[2 * h + g] = H[l].SEL[g];
				d.blk = H[l].blk;
				if (h == 0) d.R = H[l].R;
			}
			for (int l = 0; l < 32; ++l) rec[(prev_st_ + 96 + l) & 127].blk = -1;
			snap.push_back(rec);
		}
	};
	auto generic_row = [&](const int r) { // gdw96_row
		WaveRow W;
		W.r = r;
		gd_band(r, qlen, tlen, w, W.st0, W.en0);
		W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
		W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
		const int advanced = W.st_ > prev_st_;
		W.use_array = advanced, W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open, W.set_tr = (W.en0 | 15) >= r, W.ukey = gdw_edge_key(K, r);
		u32 fX[64], fV[64], fX2[64], fQ[64], hX[64], hV[64], hX2[64], hQ[64];
		for (int l = 0; l < 64; ++l) {
			const int p = (l + 63) % 64;
			const u32 aX = F[p].X[7], aV = F[p].V[7], aX2 = F[p].X2[7], aQ = F[p].Qc[3], bX = H[p].X[3], bV = H[p].V[3], bX2 = H[p].X2[3], bQ = H[p].Qc[1];
			const bool l0 = l == 0;
			fX[l] = l0 ? bX : aX, fV[l] = l0 ? bV : aV, fX2[l] = l0 ? bX2 : aX2, fQ[l] = l0 ? bQ : aQ;
			hX[l] = l0 ? aX : bX, hV[l] = l0 ? aV : bV, hX2[l] = l0 ? aX2 : bX2, hQ[l] = l0 ? aQ : bQ;
		}
		bool reloaded = false;
		for (int l = 0; l < 64; ++l) {
			if (r > 0) {
				const u32 seam = gdw_qbyte(query, qlen, r - (prev_st_ << 4));
				gdw_shift_query(F[l], fQ[l], F[l].blk == prev_st_, seam);
				gdw_shift_query_half(H[l], hQ[l], H[l].blk == prev_st_ && H[l].half == 0, seam);
			}
			if (F[l].blk < W.st_) gdw_load_block(F[l], K, F[l].blk + 96, r, query, qlen, target, tlen), reloaded = true;
			if (H[l].blk < W.st_) gdw_load_half(H[l], K, H[l].blk + 96, H[l].half, r, query, qlen, target, tlen), reloaded = true;
		}
		if (reloaded != (advanced != 0) && reloaded) { fprintf(stderr, "ring96: reload without an advance\n"); exit(2); }
		bool any_tn = false;
		for (int l = 0; l < 64; ++l) any_tn |= (F[l].tn | H[l].tn) != 0;
		const int remask = W.st0 != prev_st0 || W.up != prev_up || advanced;
		for (int l = 0; l < 64; ++l) {
			if (W.set_tr) gdw_reset_tr(F[l], K, W), gdw_reset_tr_half(H[l], K, W);
			if (remask) gdw_make_sel(F[l], W.st0, W.up), gdw_make_sel_half(H[l], W.st0, W.up);
			gdw_update_scores(F[l], K, any_tn);
			gdw_update_scores_half(H[l], K, any_tn);
		}
		for (int l = 0; l < 64; ++l) {
			if (F[l].blk <= W.en_) {
				u32 out[4];
				gdw_compute<true>(F[l], K, W, fX[l], fV[l], fX2[l], out);
			}
			if (H[l].blk <= W.en_) gdw_compute_half(H[l], K, W, hX[l], hV[l], hX2[l]);
		}
		for (int l = 0; l < 64; ++l) {
			if (r == 0) F[l].R = gdw_lo(F[l].V[0]) - K.B1 - K.qe8, H[l].R = gdw_lo(H[l].V[0]) - K.B1 - K.qe8;
			else F[l].R += gdw_lo(F[l].V[0]) - K.B1, H[l].R += gdw_lo(H[l].V[0]) - K.B1;
		}
		if (r > 0 && W.en0 != prev_en0 && (W.en0 & 7) == 0) {
			int ha[64], hb[64];
			for (int l = 0; l < 64; ++l) ha[l] = gdw_track_handoff(F[(l + 63) % 64]), hb[l] = gdw_track_handoff_half(H[(l + 63) % 64]);
			for (int l = 0; l < 64; ++l) {
				const int hf = l == 0 ? hb[l] : ha[l], hh = l == 0 ? ha[l] : hb[l];
				if ((W.en0 & 15) == 0) {
					if (F[l].blk == W.en_) F[l].R = hf + gdw_lo(F[l].U[0]);
					if (H[l].blk == W.en_ && H[l].half == 0) H[l].R = hh + gdw_lo(H[l].U[0]);
				} else if (H[l].blk == W.en_ && H[l].half == 1) H[l].R = hh + gdw_lo(H[l].U[0]);
			}
		}
		if (W.en0 == tlen - 1) {
			for (int l = 0; l < 64; ++l) {
				if (F[l].blk == mlast) {
					if (!have_f) Rf = gdw_track_to_slot(F[l], sl);
					else Rf += gdw_cell(F[l].V, sl) - K.B1;
				}
				if (H[l].blk == mlast && H[l].half == (sl >> 3)) {
					if (!have_f) Rf = gdw_track_to_slot_half(H[l], sl & 7);
					else Rf += gdw_cell_half(H[l].V, sl & 7) - K.B1;
				}
			}
			have_f = 1;
		}
		prev_st_ = W.st_, prev_st0 = W.st0, prev_up = W.up, prev_en0 = W.en0;
	};
	// gdw96_pair_row: the rows in the middle of a long alignment as pairs (lane masks m_lowF / m_lowH per lane)
	std::vector<u32> m_lowF(64), m_lowH(64);
	const int nblkA = (w - 1 + 16) >> 4, nblkB = (w + 16) >> 4;
	auto pair_row = [&](const int r, const int m, const bool ROW_A) {
		WaveRow W;
		W.r = r, W.st0 = m, W.en0 = ROW_A ? m + w - 1 : m + w;
		W.st_ = m >> 4, W.en_ = W.en0 >> 4;
		W.up = m + ((ROW_A ? nblkA : nblkB) << 4);
		{ // the closed forms must be the band of the reference
			int st0, en0;
			gd_band(r, qlen, tlen, w, st0, en0);
			if (st0 != W.st0 || en0 != W.en0 || W.up != st0 + (((en0 - st0 + 16) >> 4) << 4) || (en0 | 15) >= r || st0 < 16 || en0 >= tlen - 1) { fprintf(stderr, "ring96 pair row %d: band mismatch\n", r); exit(2); }
		}
		const int advanced = ROW_A && (m & 15) == 0;
		const int pst_ = W.st_ - advanced;
		if (pst_ != prev_st_) { fprintf(stderr, "ring96 pair row %d: pst_ %d != prev_st_ %d\n", r, pst_, prev_st_); exit(2); }
		W.use_array = advanced, W.v1key = K.key_open, W.set_tr = 0, W.ukey = 0;
		u32 fX[64], fV[64], fX2[64], fQ[64], hX[64], hV[64], hX2[64], hQ[64];
		for (int l = 0; l < 64; ++l) {
			const int p = (l + 63) % 64;
			const u32 aX = F[p].X[7], aV = F[p].V[7], aX2 = F[p].X2[7], aQ = F[p].Qc[3], bX = H[p].X[3], bV = H[p].V[3], bX2 = H[p].X2[3], bQ = H[p].Qc[1];
			const bool l0 = l == 0;
			fX[l] = l0 ? bX : aX, fV[l] = l0 ? bV : aV, fX2[l] = l0 ? bX2 : aX2, fQ[l] = l0 ? bQ : aQ;
			hX[l] = l0 ? aX : bX, hV[l] = l0 ? aV : bV, hX2[l] = l0 ? aX2 : bX2, hQ[l] = l0 ? aQ : bQ;
		}
		const u32 seam = gdw_qbyte(query, qlen, r - (pst_ << 4));
		for (int l = 0; l < 64; ++l) {
			gdw_shift_query_m(F[l], fQ[l], m_lowF[l], seam);
			gdw_shift_query_half_m(H[l], hQ[l], m_lowH[l], seam);
			if (advanced) {
				if (F[l].blk < W.st_) gdw_load_block(F[l], K, F[l].blk + 96, r, query, qlen, target, tlen);
				if (H[l].blk < W.st_) gdw_load_half(H[l], K, H[l].blk + 96, H[l].half, r, query, qlen, target, tlen);
			} else if (F[l].blk < W.st_ || H[l].blk < W.st_) { fprintf(stderr, "ring96 pair row %d: a block below the window\n", r); exit(2); }
		}
		bool any_tn = false;
		for (int l = 0; l < 64; ++l) any_tn |= (F[l].tn | H[l].tn) != 0;
		if (ROW_A || nblkA != nblkB) {
			u32 lo[4], hi[4];
			gdw_sel_uniform(m & 15, lo, hi);
			for (int l = 0; l < 64; ++l) {
				gdw_pick_sel(F[l], W.st_, W.up >> 4, (W.up >> 4) - W.st_ >= 96 - 1, lo, hi, m_lowF[l]);
				gdw_make_sel_half(H[l], W.st0, W.up);
				m_lowH[l] = (H[l].blk == W.st_ && H[l].half == 0) ? ~0u : 0u;
			}
		}
		for (int l = 0; l < 64; ++l) {
			gdw_update_scores(F[l], K, any_tn);
			gdw_update_scores_half(H[l], K, any_tn);
		}
		for (int l = 0; l < 64; ++l) {
			W.m_first_valid = 1, W.m_first = m_lowF[l], W.m_first_h = m_lowH[l];
			if (F[l].blk <= W.en_) {
				u32 out[4];
				gdw_compute<true>(F[l], K, W, fX[l], fV[l], fX2[l], out);
			}
			if (H[l].blk <= W.en_) gdw_compute_half(H[l], K, W, hX[l], hV[l], hX2[l]);
		}
		for (int l = 0; l < 64; ++l) F[l].R += gdw_lo(F[l].V[0]), H[l].R += gdw_lo(H[l].V[0]);
		if (!ROW_A && (W.en0 & 7) == 0) {
			int ha[64], hb[64];
			for (int l = 0; l < 64; ++l) ha[l] = gdw_track_handoff(F[(l + 63) % 64]), hb[l] = gdw_track_handoff_half(H[(l + 63) % 64]);
			for (int l = 0; l < 64; ++l) {
				const int hf = l == 0 ? hb[l] : ha[l], hh = l == 0 ? ha[l] : hb[l];
				if ((W.en0 & 15) == 0) {
					if (F[l].blk == W.en_) F[l].R = hf + gdw_lo(F[l].U[0]);
					if (H[l].blk == W.en_ && H[l].half == 0) H[l].R = hh + gdw_lo(H[l].U[0]);
				} else if (H[l].blk == W.en_ && H[l].half == 1) H[l].R = hh + gdw_lo(H[l].U[0]);
			}
		}
		prev_st_ = W.st_;
	};
	int rA, rS, r = 0;
	gdw_steady_rows(qlen, tlen, w, rA, rS);
	for (; r <= rend && r < rA; ++r) snapshot(r), generic_row(r);
	if (r == rA && rS > rA) {
		int m = (rA - w + 1) >> 1;
		for (int l = 0; l < 64; ++l) m_lowF[l] = F[l].blk == prev_st_ ? ~0u : 0u, m_lowH[l] = (H[l].blk == prev_st_ && H[l].half == 0) ? ~0u : 0u;
		for (; r < rS; r += 2, ++m) {
			snapshot(r), pair_row(r, m, true);
			snapshot(r + 1), pair_row(r + 1, m, false);
		}
		for (int l = 0; l < 64; ++l) F[l].R -= (rS - rA) * K.B1, H[l].R -= (rS - rA) * K.B1;
		--m;
		prev_st_ = m >> 4, prev_st0 = m, prev_up = m + (nblkB << 4), prev_en0 = m + w;
	}
	for (; r <= rend; ++r) snapshot(r), generic_row(r);
	return Rf;
}

static EmuResult emulate_ckpt(const uint8_t *query, int qlen, const uint8_t *target, int tlen, int w, const KswConst &C, int CK, bool ring96 = false)
{
	const int LANES = 128;
	WaveK K;
	if (!gdw_make_consts(C, K)) { fprintf(stderr, "consts rejected\n"); exit(2); }
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int rend = qlen + tlen - 2, mlast = (tlen - 1) >> 4, sl = (tlen - 1) & 15;
	std::vector<WaveLane> L(LANES);
	std::vector<std::vector<WaveLane>> snap;
	for (int l = 0; l < LANES; ++l) gdw_load_block(L[l], K, l, 0, query, qlen, target, tlen);
	int prev_st_ = 0, prev_st0 = -1, prev_up = -1, prev_en0 = -1, have_f = 0, Rf = 0;
	if (ring96) Rf = pass1_ring96(query, qlen, target, tlen, w, K, CK, snap);
	for (int r = 0; r <= rend && !ring96; ++r) { // pass 1 (the row loop of emulate(), nothing stored)
		if (r % CK == 0) snap.push_back(L);
		WaveRow W;
		W.r = r;
		gd_band(r, qlen, tlen, w, W.st0, W.en0);
		W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
		W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
		const int advanced = W.st_ > prev_st_;
		W.use_array = advanced, W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open, W.set_tr = (W.en0 | 15) >= r, W.ukey = gdw_edge_key(K, r);
		std::vector<u32> pX(LANES), pV(LANES), pX2(LANES), pQ(LANES);
		for (int l = 0; l < LANES; ++l) {
			const int p = (l + LANES - 1) % LANES;
			pX[l] = L[p].X[7], pV[l] = L[p].V[7], pX2[l] = L[p].X2[7], pQ[l] = L[p].Qc[3];
		}
		int reloaded = 0;
		for (int l = 0; l < LANES; ++l) {
			if (r > 0) gdw_shift_query(L[l], pQ[l], L[l].blk == prev_st_, gdw_qbyte(query, qlen, r - (prev_st_ << 4)));
			if (L[l].blk < W.st_) gdw_load_block(L[l], K, L[l].blk + LANES, r, query, qlen, target, tlen), reloaded = 1;
		}
		const int remask = W.st0 != prev_st0 || W.up != prev_up || reloaded;
		bool any_tn = false;
		for (int l = 0; l < LANES; ++l) any_tn |= L[l].tn != 0;
		for (int l = 0; l < LANES; ++l) {
			if (W.set_tr) gdw_reset_tr(L[l], K, W);
			if (remask) gdw_make_sel(L[l], W.st0, W.up);
			gdw_update_scores(L[l], K, any_tn);
		}
		for (int l = 0; l < LANES; ++l)
			if (L[l].blk <= W.en_) {
				u32 out[4];
				gdw_compute<true>(L[l], K, W, pX[l], pV[l], pX2[l], out);
			}
		if (r == 0) L[0].R = gdw_lo(L[0].V[0]) - K.B1 - K.qe8;
		else for (int l = 0; l < LANES; ++l) L[l].R += gdw_lo(L[l].V[0]) - K.B1;
		if (r > 0 && W.en0 != prev_en0 && (W.en0 & 15) == 0) {
			const int m = W.en0 >> 4;
			const int h = gdw_track_handoff(L[(m - 1) % LANES]);
			L[m % LANES].R = h + gdw_lo(L[m % LANES].U[0]);
		}
		if (W.en0 == tlen - 1) {
			WaveLane &F = L[mlast % LANES];
			if (!have_f) Rf = gdw_track_to_slot(F, sl), have_f = 1;
			else Rf += gdw_cell(F.V, sl) - K.B1;
		}
		prev_st_ = W.st_, prev_st0 = W.st0, prev_up = W.up, prev_en0 = W.en0;
	}
	EmuResult res;
	res.score = Rf / 8;
	// pass 2
	int wi = tlen - 1, wj = qlen - 1, state = 0;
	std::vector<uint32_t> cg; // reversed
	auto push = [&](uint32_t op, uint32_t len) {
		if (!cg.empty() && (cg.back() & 0xf) == op) cg.back() += len << 4;
		else cg.push_back(len << 4 | op);
	};
	for (int k = (int)snap.size() - 1; k >= 0 && wi >= 0 && wj >= 0 && ring96; --k) { // the cone with one HALF block per lane (gdw_cone_restore_half / gdw_cone_row_half)
		const int r0 = k * CK, rtop = wi + wj;
		if (rtop < r0) continue;
		const int r1 = rtop < r0 + CK - 1 ? rtop : r0 + CK - 1, hb0 = (wi >> 3) - 63;
		std::vector<WaveHalf> Cn(64);
		for (int l = 0; l < 64; ++l) {
			const int hidx = hb0 + l, blk = hidx >> 1, half = hidx & 1;
			bool have = false;
			if (blk >= 0 && snap[k][blk & 127].blk == blk) gdw_half_from_lane(snap[k][blk & 127], half, Cn[l]), have = true;
			if (!have) gdw_fresh_half(Cn[l], K, blk, half, r0 > 0 ? r0 - 1 : 0, query, qlen, target, tlen);
		}
		bool any_tn = false;
		for (int l = 0; l < 64; ++l) any_tn |= Cn[l].tn != 0;
		int pst_ = 0, pst0 = -1, pup = -1;
		if (r0 > 0) {
			int st0, en0;
			gd_band(r0 - 1, qlen, tlen, w, st0, en0);
			pst_ = st0 >> 4, pst0 = st0, pup = st0 + (((en0 - st0 + 16) >> 4) << 4);
		}
		std::vector<uint8_t> buf((size_t)(r1 - r0 + 1) * 512, 0xEE), valid((size_t)(r1 - r0 + 1) * 64, 0);
		const bool interior = gdw_cone_interior_half(r0, r1, hb0, qlen, tlen, w); // gdw_cone_row_half_fast
		if (interior) ++g_fast_chunks;
		for (int r = r0; r <= r1 && interior; ++r) {
			if (r == r0) for (int l = 0; l < 64; ++l) Cn[l].SEL[0] = Cn[l].SEL[1] = 0x07060504u;
			WaveRow W;
			W.r = r, W.use_array = 1;
			u32 pX[64], pV[64], pX2[64], pQ[64];
			for (int l = 0; l < 64; ++l) {
				const int p = (l + 63) % 64;
				pX[l] = Cn[p].X[3], pV[l] = Cn[p].V[3], pX2[l] = Cn[p].X2[3], pQ[l] = Cn[p].Qc[1];
			}
			for (int l = 0; l < 64; ++l) {
				gdw_shift_query_half(Cn[l], pQ[l], l == 0, gdw_qbyte(query, qlen, r - hb0 * 8));
				gdw_update_scores_half(Cn[l], K, any_tn);
			}
			for (int l = 0; l < 64; ++l) {
				u32 out[2];
				gdw_compute_half<true>(Cn[l], K, W, pX[l], pV[l], pX2[l], out);
				memcpy(&buf[(size_t)(r - r0) * 512 + l * 8], out, 8);
				valid[(size_t)(r - r0) * 64 + l] = 1;
			}
		}
		for (int r = r0; r <= r1 && !interior; ++r) {
			WaveRow W;
			W.r = r;
			gd_band(r, qlen, tlen, w, W.st0, W.en0);
			W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
			W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
			const int advanced = W.st_ > pst_;
			W.use_array = advanced, W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open, W.set_tr = (W.en0 | 15) >= r, W.ukey = gdw_edge_key(K, r);
			u32 pX[64], pV[64], pX2[64], pQ[64];
			for (int l = 0; l < 64; ++l) {
				const int p = (l + 63) % 64;
				pX[l] = Cn[p].X[3], pV[l] = Cn[p].V[3], pX2[l] = Cn[p].X2[3], pQ[l] = Cn[p].Qc[1];
			}
			for (int l = 0; l < 64; ++l) {
				if (r > 0) gdw_shift_query_half(Cn[l], pQ[l], l == 0, gdw_qbyte(query, qlen, r - hb0 * 8));
				if (W.set_tr) gdw_reset_tr_half(Cn[l], K, W);
				if (W.st0 != pst0 || W.up != pup || advanced) gdw_make_sel_half(Cn[l], W.st0, W.up);
				gdw_update_scores_half(Cn[l], K, any_tn);
			}
			for (int l = 0; l < 64; ++l)
				if (Cn[l].blk >= W.st_ && Cn[l].blk <= W.en_) {
					u32 out[2];
					gdw_compute_half<true>(Cn[l], K, W, pX[l], pV[l], pX2[l], out);
					memcpy(&buf[(size_t)(r - r0) * 512 + l * 8], out, 8);
					valid[(size_t)(r - r0) * 64 + l] = 1;
				}
			pst_ = W.st_, pst0 = W.st0, pup = W.up;
		}
		while (wi >= 0 && wj >= 0 && wi + wj >= r0) {
			const int r = wi + wj;
			int st0, en0, force_state = -1;
			gd_band(r, qlen, tlen, w, st0, en0);
			const int off = st0 & ~15, off_end = en0 | 15;
			if (wi < off) force_state = 2;
			if (wi > off_end) force_state = 1;
			uint32_t tmp = 0;
			if (force_state < 0) {
				const int b = (wi >> 3) - hb0, c = wi & 7, g = (c & 3) >> 1, h = (c & 1) | ((c >> 2) << 1);
				if (r > r1 || b < 0 || b > 63 || !valid[(size_t)(r - r0) * 64 + b]) { fprintf(stderr, "walk left the half-block cone: r=%d i=%d chunk [%d,%d] hb0=%d\n", r, wi, r0, r1, hb0); exit(2); }
				const uint8_t bb = buf[(size_t)(r - r0) * 512 + b * 8 + 4 * g + h], nb = (uint8_t)~bb;
				tmp = (uint8_t)((4 - (bb & 7)) | ((nb >> 4) & 0x08) | ((nb >> 2) & 0x10) | (nb & 0x20) | ((nb << 2) & 0x40));
			}
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) push(0, 1), --wi, --wj;
			else if (state == 1 || state == 3) push(2, 1), --wi;
			else push(1, 1), --wj;
		}
	}
	for (int k = (int)snap.size() - 1; k >= 0 && wi >= 0 && wj >= 0 && !ring96; --k) {
		const int r0 = k * CK, rtop = wi + wj;
		if (rtop < r0) continue;
		const int r1 = rtop < r0 + CK - 1 ? rtop : r0 + CK - 1, b0 = (wi >> 4) - 63;
		std::vector<WaveLane> Cn(64);
		for (int l = 0; l < 64; ++l) {
			const int blk = b0 + l;
			bool have = false;
			if (blk >= 0) {
				Cn[l] = snap[k][blk & 127], have = Cn[l].blk == blk;
				Cn[l].tn = (Cn[l].Tb[0] | Cn[l].Tb[1] | Cn[l].Tb[2] | Cn[l].Tb[3]) & 0x04040404u; // (as gdw_cone_restore)
			}
			if (!have) gdw_fresh_block(Cn[l], K, blk, r0 > 0 ? r0 - 1 : 0, query, qlen, target, tlen);
		}
		bool any_tn = false;
		for (int l = 0; l < 64; ++l) any_tn |= Cn[l].tn != 0;
		int pst_ = 0, pst0 = -1, pup = -1;
		if (r0 > 0) {
			int st0, en0;
			gd_band(r0 - 1, qlen, tlen, w, st0, en0);
			pst_ = st0 >> 4, pst0 = st0, pup = st0 + (((en0 - st0 + 16) >> 4) << 4);
		}
		std::vector<uint8_t> buf((size_t)(r1 - r0 + 1) * 1024, 0xEE);
		std::vector<uint8_t> valid((size_t)(r1 - r0 + 1) * 64, 0);
		for (int r = r0; r <= r1; ++r) { // gdw_cone_row, statement by statement
			WaveRow W;
			W.r = r;
			gd_band(r, qlen, tlen, w, W.st0, W.en0);
			W.st_ = W.st0 >> 4, W.en_ = W.en0 >> 4;
			W.up = W.st0 + (((W.en0 - W.st0 + 16) >> 4) << 4);
			const int advanced = W.st_ > pst_;
			W.use_array = advanced, W.v1key = W.st_ == 0 ? gdw_edge_key(K, r) : K.key_open, W.set_tr = (W.en0 | 15) >= r, W.ukey = gdw_edge_key(K, r);
			u32 pX[64], pV[64], pX2[64], pQ[64];
			for (int l = 0; l < 64; ++l) {
				const int p = (l + 63) % 64;
				pX[l] = Cn[p].X[7], pV[l] = Cn[p].V[7], pX2[l] = Cn[p].X2[7], pQ[l] = Cn[p].Qc[3];
			}
			for (int l = 0; l < 64; ++l) {
				if (r > 0) gdw_shift_query(Cn[l], pQ[l], l == 0, gdw_qbyte(query, qlen, r - b0 * 16));
				if (W.set_tr) gdw_reset_tr(Cn[l], K, W);
				if (W.st0 != pst0 || W.up != pup || advanced) gdw_make_sel(Cn[l], W.st0, W.up);
				gdw_update_scores(Cn[l], K, any_tn);
			}
			for (int l = 0; l < 64; ++l)
				if (Cn[l].blk >= W.st_ && Cn[l].blk <= W.en_) {
					u32 out[4];
					gdw_compute<true>(Cn[l], K, W, pX[l], pV[l], pX2[l], out);
					memcpy(&buf[(size_t)(r - r0) * 1024 + l * 16], out, 16);
					valid[(size_t)(r - r0) * 64 + l] = 1;
				}
			pst_ = W.st_, pst0 = W.st0, pup = W.up;
		}
		while (wi >= 0 && wj >= 0 && wi + wj >= r0) { // the walk (SR/ksw2.h:131-163) over the rows of this chunk
			const int r = wi + wj;
			int st0, en0, force_state = -1;
			gd_band(r, qlen, tlen, w, st0, en0);
			const int off = st0 & ~15, off_end = en0 | 15;
			if (wi < off) force_state = 2;
			if (wi > off_end) force_state = 1;
			uint32_t tmp = 0;
			if (force_state < 0) {
				const int b = (wi >> 4) - b0, c = wi & 15, g = (c & 7) >> 1, h = (c & 1) | ((c >> 3) << 1);
				if (r > r1 || b < 0 || b > 63 || !valid[(size_t)(r - r0) * 64 + b]) { fprintf(stderr, "walk left the cone: r=%d i=%d chunk [%d,%d] b0=%d\n", r, wi, r0, r1, b0); exit(2); }
				const uint8_t bb = buf[(size_t)(r - r0) * 1024 + b * 16 + 4 * g + h], nb = (uint8_t)~bb;
				tmp = (uint8_t)((4 - (bb & 7)) | ((nb >> 4) & 0x08) | ((nb >> 2) & 0x10) | (nb & 0x20) | ((nb << 2) & 0x40));
			}
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) push(0, 1), --wi, --wj;
			else if (state == 1 || state == 3) push(2, 1), --wi;
			else push(1, 1), --wj;
		}
	}
	if (wi >= 0) push(2, wi + 1);
	if (wj >= 0) push(1, wj + 1);
	res.cigar.assign(cg.rbegin(), cg.rend());
	return res;
}

static void mutate(std::mt19937 &g, const std::vector<uint8_t> &t, std::vector<uint8_t> &q, double sub, double ins, double del, double nfrac)
{
	std::uniform_real_distribution<double> U(0, 1);
	q.clear();
	for (uint8_t c : t) {
		double r = U(g);
		if (r < del) continue;
		if (r < del + ins) q.push_back(g() & 3);
		if (U(g) < sub) c = (c + 1 + g() % 3) & 3;
		if (U(g) < nfrac) c = 4;
		q.push_back(c);
	}
	if (q.empty()) q.push_back(0);
}

int main(int argc, char **argv)
{
	const unsigned seed = argc > 1 ? atoi(argv[1]) : 1;
	const int n = argc > 2 ? atoi(argv[2]) : 200, LANES = argc > 3 ? atoi(argv[3]) : 64;
	g_single = argc > 4 && !strcmp(argv[4], "single");
	const bool ring96 = argc > 5 && !strcmp(argv[4], "ckpt96"); // ./wave_emul <seed> <n> 96 ckpt96 <rows per chunk>
	const int ckpt = argc > 5 && (!strcmp(argv[4], "ckpt") || ring96) ? atoi(argv[5]) : 0; // ./wave_emul <seed> <n> 128 ckpt <rows per chunk>
	std::mt19937 g(seed);
	const int presets[3][6] = {{2, 8, 12, 2, 24, 1}, {1, 4, 6, 2, 26, 1}, {2, 4, 4, 2, 24, 1}};
	int n_run = 0, n_bad = 0, n_skip = 0;
	for (int it = 0; it < n; ++it) {
		const int *P = presets[it % 3];
		int tlen, w;
		double sub = 0.01, ins = 0.003, del = 0.003, nfrac = (it % 7 == 0) ? 0.02 : 0.0;
		if (LANES == 16) tlen = 100 + g() % 120, w = 32 + g() % 130;
		else if (LANES == 10) tlen = (it % 4 == 0) ? 150 : 100 + g() % 61, w = (it % 4 == 0) ? 150 : 32 + g() % 130; // targets of <= 160 bases
		else if (LANES == 8) tlen = 30 + g() % 99, w = 20 + g() % 130;                                                // <= 128
		else if (LANES == 128) tlen = 1500 + g() % 3000, w = (it % 3 == 0) ? 1300 : 1010 + g() % 1000, sub = 0.03, ins = 0.02, del = 0.02; // ONT bands
		else if (LANES == 96) tlen = 1500 + g() % 3000, w = (it % 3 == 0) ? 1300 : 1010 + g() % 480, sub = 0.03, ins = 0.02, del = 0.02; // ... that fit 96 blocks
		else {
			switch (it % 5) {
			case 0: tlen = 150, w = 150; break;
			case 1: tlen = 300 + g() % 2500, w = 40 + g() % 400; break;
			case 2: tlen = 1200 + g() % 1800, w = 1000; break;
			case 3: tlen = 64 + g() % 400, w = 32 + g() % 100, sub = 0.05, ins = 0.03, del = 0.03; break;
			default: tlen = 500 + g() % 1500, w = 17 + g() % 985, sub = 0.03, ins = 0.02, del = 0.02; break;
			}
		}
		std::vector<uint8_t> t(tlen), q;
		for (auto &c : t) c = g() & 3;
		if (nfrac > 0) for (auto &c : t) if ((g() % 1000) < 20) c = 4;
		mutate(g, t, q, sub, ins, del, nfrac);
		if (it % 11 == 3 && (int)q.size() > 200) { // a long indel near the band edge
			int sz = (int)(w * 0.45), pos = 50 + g() % (q.size() - 100);
			if (it & 1) q.insert(q.begin() + pos, sz, (uint8_t)(g() & 3));
			else if (pos + sz < (int)q.size()) q.erase(q.begin() + pos, q.begin() + pos + sz);
		}
		if (it % 4 == 1 && !g_single) for (auto &c : q) if (c == 4 || (g() % 400) == 0) c = 7; // N of a reverse-complemented read (LR/map.c:1634)
		const int qlen = (int)q.size();
		KswConst C;
		C.q = P[2], C.e = P[3], C.q2 = g_single ? P[2] : P[4], C.e2 = g_single ? P[3] : P[5];
		if (C.q2 + C.e2 < C.q + C.e) std::swap(C.q, C.q2), std::swap(C.e, C.e2);
		C.sc_mch = P[0], C.sc_mis = -P[1], C.sc_N = -C.e2;
		C.long_thres = C.e != C.e2 ? (C.q2 - C.q) / (C.e - C.e2) - 1 : 0;
		if (C.q2 + C.e2 + C.long_thres * C.e2 > C.q + C.e + C.long_thres * C.e) ++C.long_thres;
		C.long_diff = C.long_thres * (C.e - C.e2) - (C.q2 - C.q) - C.e2;
		// (the library's rule for the 10- / 8-lane groups: whatever the 16-lane form takes, if the target has at most 16 * lanes bases)
		if (LANES == 10 || LANES == 8 ? !(gd_wave_geometry_ok(qlen, tlen, w, 16) && tlen <= 16 * LANES) : !gd_wave_geometry_ok(qlen, tlen, w, LANES)) { ++n_skip; continue; }
		int8_t mat[25];
		for (int i = 0; i < 25; ++i) mat[i] = (i / 5 == 4 || i % 5 == 4) ? 0 : (i / 5 == i % 5 ? P[0] : -P[1]);
		gdo_extz_t ez;
		memset(&ez, 0, sizeof(ez));
		// (the extz2 oracle has the SSE score rule only: identical to the AVX-512 table except for query byte 7, which the single runs avoid)
		if (g_single) gdo_ksw_extz2(qlen, q.data(), tlen, t.data(), 5, mat, P[2], P[3], w, -1, 0, GDO_EZ_APPROX_MAX, &ez);
		else gdo_ksw_extd2(qlen, q.data(), tlen, t.data(), 5, mat, P[2], P[3], P[4], P[5], w, -1, 0, GDO_EZ_APPROX_MAX | GDO_EZ_AVX512_SC, &ez);
		EmuResult e = ckpt ? emulate_ckpt(q.data(), qlen, t.data(), tlen, w, C, ckpt, ring96) : emulate(LANES, q.data(), qlen, t.data(), tlen, w, C);
		++n_run;
		bool ok = e.score == ez.score && (int)e.cigar.size() == ez.n_cigar && (ez.n_cigar == 0 || !memcmp(e.cigar.data(), ez.cigar, 4 * ez.n_cigar));
		if (!ok) {
			++n_bad;
			if (n_bad <= 10) fprintf(stderr, "MISMATCH it=%d qlen=%d tlen=%d w=%d preset=%d score emu=%d oracle=%d ncig %zu/%d\n", it, qlen, tlen, w, it % 3, e.score, ez.score, e.cigar.size(), ez.n_cigar);
		}
		free(ez.cigar);
	}
	printf("wave_emul lanes=%d pairs_run=%d skipped=%d mismatches=%d fast_cone_chunks=%ld\n", LANES, n_run, n_skip, n_bad, g_fast_chunks);
	return n_bad ? 1 : (n_run ? 0 : 3);
}
