// K1 (register-resident form): the per-lane arithmetic of the wavefront ksw_extd2 kernel, written once for the
// device kernel (ksw_wave.hip.h) and for the host lock-step emulator used by tests/test_wave_emulator.py.
//
// Mapping (HiFi geometry, w = 1000): one 64-lane wavefront per alignment.  The reference's 16-lane SSE block
// [16m, 16m+15] of target positions is ONE GPU lane; lane = m mod 64, so the 64 lanes always hold the 64
// consecutive blocks [st_, st_+63] that can be touched by an anti-diagonal (n_col_ <= 64).  A lane therefore is
// either wholly inside the reference's 16-aligned computed window [st, en] or wholly outside it, which makes the
// reference's padding-cell behaviour (SURVEY Notes K1.1-K1.3) fall out of plain lane predication.
//
// Per lane the 16 cells are held as 8 packed registers per state array: register k = (cell k | cell k+8 << 16),
// so "row r-1, cell t-1" is register k-1 (no instruction) except for k = 0, which takes one DPP rotate from the
// previous lane plus one v_alignbit.  Arithmetic is packed 16-bit (v_pk_add/sub/max/min_i16): two cells per VALU
// instruction.  Values are kept as KEYS  8*value + tie + bias:
//     U,V: 8u+B1, 8v+B1      X: 8x+3+B1   Y: 8y+2+B1   X2: 8x2+1+B2   Y2: 8y2+B2      S: 8s+4+2*B1
//     B1 = 8*(q+e-1), B2 = 8*(q2+e2-1)
//   * a = X+V, b = Y+U carry bias 2*B1, a2 = X2+V, b2 = Y2+U carry B1+B2; with c2 = B1-B2 added to max(a2,b2) all
//     five candidates carry bias 2*B1 and tie codes 4 (S),3,2,1,0, so a chain of packed max gives both z (key & ~7)
//     and the reference's direction d = 4 - (key & 7): the strict-'>' priority order z,a,b,a2,b2 of
//     SR/ksw2_extd2_sse.c:235-242 is "largest value, then largest tie code".
//   * with B1 = 8(q+e-1) the no-continuation value x = -(q+e) is key -5 (y: -6) and every continuation value is
//     >= 0 (x2 = -(q2+e2): key -7, y2: -8 thanks to B2), so all four E/F continuation flags (:263-272) are the sign
//     bits of the stored X/Y/X2/Y2 registers.
// 8-bit wrap-around of the reference cannot occur for parameters accepted by gd_wave_supported() (all keys stay
// far inside int16), so 16-bit arithmetic reproduces the int8 results exactly; the emulator test and the GPU
// parity tests check that against the oracle.
//
// The score (approximate-max walk, :367-383) is obtained from the identity H(r,t)-H(r-1,t) = v(r,t),
// H(r,t+1)-H(r,t) = u(r,t+1)-v(r,t): the sum along ANY monotone in-band path from (0,0) to the last cell equals
// the reference's H0 (each unit square between two such paths contributes u+v'=v+u' identically, because both
// u(r,t) and v(r,t) are z minus the stored neighbours).  Each lane tracks 8*H at its cell 0; see gdw_track_*.
#pragma once
#include <stdint.h>
#include "ksw_common.h"

#if defined(__HIPCC__)
#define GDW_HD __host__ __device__ __forceinline__
#else
#define GDW_HD static inline
#endif

typedef uint32_t u32;

// ---- packed 16-bit helpers ---------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef short gdw_v2s __attribute__((ext_vector_type(2)));
GDW_HD gdw_v2s gdw_as_v(u32 a) { return __builtin_bit_cast(gdw_v2s, a); }
GDW_HD u32 gdw_as_u(gdw_v2s a) { return __builtin_bit_cast(u32, a); }
GDW_HD u32 pk_add(u32 a, u32 b) { return gdw_as_u(gdw_as_v(a) + gdw_as_v(b)); }
GDW_HD u32 pk_sub(u32 a, u32 b) { return gdw_as_u(gdw_as_v(a) - gdw_as_v(b)); }
GDW_HD u32 pk_max(u32 a, u32 b) { return gdw_as_u(__builtin_elementwise_max(gdw_as_v(a), gdw_as_v(b))); }
GDW_HD u32 pk_min(u32 a, u32 b) { return gdw_as_u(__builtin_elementwise_min(gdw_as_v(a), gdw_as_v(b))); }
GDW_HD u32 gdw_perm(u32 s0, u32 s1, u32 sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
GDW_HD u32 gdw_alignbit(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
GDW_HD u32 gdw_alignbyte(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
#else
GDW_HD u32 pk_add(u32 a, u32 b) { return ((a + b) & 0xffffu) | ((((a >> 16) + (b >> 16)) & 0xffffu) << 16); }
GDW_HD u32 pk_sub(u32 a, u32 b) { return ((a - b) & 0xffffu) | ((((a >> 16) - (b >> 16)) & 0xffffu) << 16); }
GDW_HD u32 pk_max(u32 a, u32 b)
{
	int16_t al = (int16_t)a, bl = (int16_t)b, ah = (int16_t)(a >> 16), bh = (int16_t)(b >> 16);
	return (u32)(uint16_t)(al > bl ? al : bl) | (u32)(uint16_t)(ah > bh ? ah : bh) << 16;
}
GDW_HD u32 pk_min(u32 a, u32 b)
{
	int16_t al = (int16_t)a, bl = (int16_t)b, ah = (int16_t)(a >> 16), bh = (int16_t)(b >> 16);
	return (u32)(uint16_t)(al < bl ? al : bl) | (u32)(uint16_t)(ah < bh ? ah : bh) << 16;
}
GDW_HD u32 gdw_perm(u32 s0, u32 s1, u32 sel) // v_perm_b32: bytes 0-3 = s1, 4-7 = s0, 12 -> 0x00, >= 13 -> 0xff
{
	uint64_t src = (uint64_t)s0 << 32 | s1;
	u32 out = 0;
	for (int i = 0; i < 4; ++i) {
		u32 s = (sel >> (8 * i)) & 0xff, b;
		if (s <= 7) b = (u32)(src >> (8 * s)) & 0xff;
		else if (s == 12) b = 0;
		else if (s >= 13) b = 0xff;
		else b = ((src >> (16 * (s - 8) + 15)) & 1) ? 0xff : 0; // 8..11: sign of bytes 1,3,5,7 (unused here)
		out |= b << (8 * i);
	}
	return out;
}
GDW_HD u32 gdw_alignbit(u32 hi, u32 lo, u32 sh) { return (u32)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
GDW_HD u32 gdw_alignbyte(u32 hi, u32 lo, u32 sh) { return (u32)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
#endif
GDW_HD u32 gdw_bfi(u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }
// the same with a wave-uniform mask (a constant, or derived from the anti-diagonal index): one v_bfi_b32 with the mask in an SGPR.
// The compiler splits the C form into v_and + v_and + v_or once the mask is a constant (~mask folds into a second constant).
#if defined(__HIP_DEVICE_COMPILE__)
GDW_HD u32 gdw_bfi_u(u32 mask, u32 a, u32 b)
{
	u32 d;
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
	return d;
}
#else
GDW_HD u32 gdw_bfi_u(u32 mask, u32 a, u32 b) { return gdw_bfi(mask, a, b); }
#endif
GDW_HD u32 gdw_pack2(int v) { return ((u32)v & 0xffffu) | ((u32)v << 16); }
GDW_HD int gdw_lo(u32 a) { return (int)(int16_t)(a & 0xffffu); }
GDW_HD int gdw_hi(u32 a) { return (int)(int16_t)(a >> 16); }

// ---- uniform constants -------------------------------------------------------------------------------------
struct WaveK {
	int32_t B1;                // 8*(q+e-1)
	u32 cx, cy, cx2, cy2;      // packed no-continuation / initial keys of X, Y, X2, Y2 (-5, -6, -7, -8)
	u32 c2;                    // packed B1 - B2: lifts max(a2, b2) to the bias of the other candidates
	u32 uv0;                   // packed initial key of U and V (value -(q+e))
	u32 zmax;                  // packed 8*sc_mch + 2*B1
	u32 te, te2;               // packed: tE = z8 - te (te = 8(q-1)); tE2 = z8 - te2 (te2 = B1 - 8*e2)
	u32 lut_lo, lut_hi;        // S-key bytes indexed by (T^Q)|TN: [match,mis,mis,mis] / [N,N,N,N]
	u32 s0;                    // S-key byte of score 0, replicated (the reference's zero-filled s[])
	int32_t key_open, key_e, key_ld, key_e2; // boundary keys of v1 / u[r]: -(q+e), -e, long_diff, -e2 (SR/ksw2_extd2_sse.c:158,162)
	int32_t long_thres;
	int32_t qe8;               // 8*(q+e)
};

static inline bool gdw_make_consts(const KswConst &C, WaveK &K)
{
	const int qe = C.q + C.e, qe2 = C.q2 + C.e2;
	K.B1 = 8 * (qe - 1);
	if (qe < 1) return false;
	K.cx = gdw_pack2(8 * -qe + 3 + K.B1), K.cy = gdw_pack2(8 * -qe + 2 + K.B1);
	const int B2 = 8 * (qe2 - 1);
	K.cx2 = gdw_pack2(8 * -qe2 + 1 + B2), K.cy2 = gdw_pack2(8 * -qe2 + B2);
	K.c2 = gdw_pack2(K.B1 - B2);
	K.uv0 = gdw_pack2(8 * -qe + K.B1);
	K.zmax = gdw_pack2(8 * C.sc_mch + 2 * K.B1);
	K.te = gdw_pack2(8 * (C.q - 1)), K.te2 = gdw_pack2(K.B1 - 8 * C.e2);
	const int km = 8 * C.sc_mch + 4 + 2 * K.B1, kx = 8 * C.sc_mis + 4 + 2 * K.B1, kn = 8 * C.sc_N + 4 + 2 * K.B1, k0 = 4 + 2 * K.B1;
	if (km < 0 || km > 255 || kx < 0 || kx > 255 || kn < 0 || kn > 255 || k0 > 255) return false; // S keys must fit a byte
	K.lut_lo = (u32)km | (u32)kx << 8 | (u32)kx << 16 | (u32)kx << 24;
	K.lut_hi = (u32)kn * 0x01010101u;
	K.s0 = (u32)k0 * 0x01010101u;
	K.key_open = 8 * -qe + K.B1, K.key_e = 8 * -C.e + K.B1, K.key_ld = 8 * C.long_diff + K.B1, K.key_e2 = 8 * -C.e2 + K.B1;
	K.long_thres = C.long_thres;
	K.qe8 = 8 * qe;
	// every key must stay well inside int16: |value| <= 127 => |key| <= 8*127 + 7 + 2*B1
	if (8 * 127 + 7 + 2 * K.B1 > 30000) return false;
	return true;
}

// boundary key of v1 (st == 0) and of u[r] (SR/ksw2_extd2_sse.c:158,162)
GDW_HD int gdw_edge_key(const WaveK &K, int r)
{
	return r == 0 ? K.key_open : r < K.long_thres ? K.key_e : r == K.long_thres ? K.key_ld : K.key_e2;
}

// ---- per-lane state ----------------------------------------------------------------------------------------
struct WaveLane {
	u32 U[8], V[8], X[8], Y[8], X2[8], Y2[8]; // packed keys, register k = cells (k, k+8) of the lane's block
	u32 Sb[4];   // S-key bytes, dword g = cells 4g..4g+3 (persistent: stale cells keep their old score)
	u32 Tb[4];   // target nt4 bytes of the block (0 beyond tlen)
	u32 Qc[4];   // query nt4 bytes facing the cells on the current anti-diagonal: cell i <-> query[r - (tb+i)]
	u32 SEL[4];  // v_perm selectors blending fresh scores into Sb (per byte: 4+i = take fresh, i = keep)
	u32 tn;      // != 0 iff the block's target bytes contain an N (4): only then the score index needs its N fix-up
	int32_t blk; // block index m held by the lane (m mod lanes == lane id)
	int32_t R;   // 8*H(r, 16m): score tracker at cell 0 of the block
};

// query byte facing cell t on anti-diagonal r; outside [0,qlen) the reference reads zeroed memory
GDW_HD u32 gdw_qbyte(const uint8_t *query, int qlen, int j) { return (j >= 0 && j < qlen) ? query[j] : 0u; }

// (re)load a lane for block m at anti-diagonal r: reference's initial fill (SR/ksw2_extd2_sse.c:107,111-116)
GDW_HD void gdw_load_block(WaveLane &L, const WaveK &K, int m, int r, const uint8_t *query, int qlen,
                           const uint8_t *target, int tlen)
{
	L.blk = m;
	const int tb = m << 4;
#pragma unroll
	for (int k = 0; k < 8; ++k) L.U[k] = K.uv0, L.V[k] = K.uv0, L.X[k] = K.cx, L.Y[k] = K.cy, L.X2[k] = K.cx2, L.Y2[k] = K.cy2;
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		u32 tw = 0, qw = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int t = tb + 4 * g + b;
			tw |= (t < tlen ? (u32)target[t] : 0u) << (8 * b);
			qw |= gdw_qbyte(query, qlen, r - t) << (8 * b);
		}
		L.Tb[g] = tw, L.Qc[g] = qw, L.Sb[g] = K.s0, L.SEL[g] = 0x03020100u;
	}
	L.tn = (L.Tb[0] | L.Tb[1] | L.Tb[2] | L.Tb[3]) & 0x04040404u;
	L.R = 0;
}

// the same for any block index (also below block 0 or beyond the target: the cone pass holds 64 consecutive blocks whatever they
// are); the query bytes are those facing the cells on anti-diagonal rq
GDW_HD void gdw_fresh_block(WaveLane &L, const WaveK &K, int blk, int rq, const uint8_t *query, int qlen, const uint8_t *target, int tlen)
{
	const int tb = blk * 16;
	L.blk = blk;
#pragma unroll
	for (int k = 0; k < 8; ++k) L.U[k] = K.uv0, L.V[k] = K.uv0, L.X[k] = K.cx, L.Y[k] = K.cy, L.X2[k] = K.cx2, L.Y2[k] = K.cy2;
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		u32 tw = 0, qw = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int t = tb + 4 * g + b;
			tw |= (t >= 0 && t < tlen ? (u32)target[t] : 0u) << (8 * b);
			qw |= gdw_qbyte(query, qlen, rq - t) << (8 * b);
		}
		L.Tb[g] = tw, L.Qc[g] = qw, L.Sb[g] = K.s0, L.SEL[g] = 0x03020100u;
	}
	L.tn = (L.Tb[0] | L.Tb[1] | L.Tb[2] | L.Tb[3]) & 0x04040404u;
	L.R = 0;
}

// uniform description of one anti-diagonal
struct WaveRow {
	int r, st0, en0, st_, en_, up;
	int use_array;   // boundary x1/x21/v1 come from the previous lane's row r-1 values (block st_-1 just retired)
	int v1key;       // otherwise: key of v1 (x1/x21 are always the no-continuation keys then)
	int set_tr;      // en >= r: cell t == r gets u/y/y2 reset (:160-163)
	int ukey;        // key of u[r] for that reset
	int m_first_valid = 0; // the caller already holds the lane mask "this lane's block is st_" (paired rows)
	u32 m_first = 0, m_first_h = 0; // (m_first_h: the same for a half block -- "the first half of block st_")
};

// selectors for the score blend of one lane: cells [st0, up) are rewritten (:166-180)
GDW_HD void gdw_make_sel(WaveLane &L, int st0, int up)
{
	const int tb = L.blk << 4;
	int lo = st0 - tb, hi = up - tb;
	lo = lo < 0 ? 0 : lo > 16 ? 16 : lo;
	hi = hi < 0 ? 0 : hi > 16 ? 16 : hi;
	const u32 bm = hi > lo ? (((1u << (hi - lo)) - 1u) << lo) : 0u; // bit i: cell i is rewritten
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const u32 n = (bm >> (4 * g)) & 15u;
		const u32 spread = (n * 0x00204081u) & 0x01010101u; // bit i of n -> bit 0 of byte i
		L.SEL[g] = 0x03020100u + (spread << 2);
	}
}

// The same selectors for the rows in the middle of a long alignment, where they depend on the lane only through its role: with
// d0 = st0 & 15, the lane of the lowest block st_ rewrites its cells >= d0, the lane of block top = up >> 4 its cells < d0 (up and
// st0 are congruent mod 16), the blocks between them all 16 cells and the blocks above top none.  The two selector sets are
// wave-uniform (scalar unit); a lane only picks.
GDW_HD void gdw_sel_uniform(int d0, u32 lo[4], u32 hi[4])
{
	const uint64_t C = 0x0404040404040404ull; // byte c of the 16-byte mask (f1:f0) = 4 iff cell c >= d0
	const uint64_t f0 = d0 < 8 ? C << (8 * d0) : 0ull, f1 = d0 < 8 ? C : C << (8 * (d0 - 8));
	lo[0] = 0x03020100u + (u32)f0, lo[1] = 0x03020100u + (u32)(f0 >> 32), lo[2] = 0x03020100u + (u32)f1, lo[3] = 0x03020100u + (u32)(f1 >> 32);
#pragma unroll
	for (int g = 0; g < 4; ++g) hi[g] = 0x0a080604u - lo[g]; // 0x03020100 + (0x04040404 - mask)
}
// (a select between two scalar values costs a copy into a vector register plus the select -- one scalar operand per instruction --
// so the picks are v_bfi_b32 with the scalar value as data and a 0 / ~0 lane mask; all_inside: no lane holds a block above top)
#if defined(__HIP_DEVICE_COMPILE__)
GDW_HD u32 gdw_bfi_s(u32 mask, u32 a_uniform, u32 b)
{
	u32 d;
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "s"(a_uniform), "v"(b));
	return d;
}
#else
GDW_HD u32 gdw_bfi_s(u32 mask, u32 a_uniform, u32 b) { return gdw_bfi(mask, a_uniform, b); }
#endif
GDW_HD void gdw_pick_sel(WaveLane &L, int st_, int top, bool all_inside, const u32 lo[4], const u32 hi[4], u32 &m_lo)
{
	m_lo = L.blk == st_ ? ~0u : 0u;
	const u32 m_top = L.blk == top ? ~0u : 0u;
	u32 base = 0x07060504u;
	if (!all_inside) base = L.blk < top ? 0x07060504u : 0x03020100u;
#pragma unroll
	for (int g = 0; g < 4; ++g) L.SEL[g] = gdw_bfi_s(m_lo, lo[g], gdw_bfi_s(m_top, hi[g], base));
}

// advance the query window by one anti-diagonal: cell i now faces what cell i-1 faced.  Cell 0 takes the byte that cell 15
// of the block below faced on the previous anti-diagonal, i.e. the top byte of the previous lane's Qc[3] (`below`, fetched
// with the same DPP rotate as the DP state) -- the query streams through the lanes and no lane touches memory.  Only the
// lane holding the lowest block has nobody below it: it takes `seam`, the one fresh byte query[r - 16*lowest_block].
GDW_HD void gdw_shift_query(WaveLane &L, u32 below, bool is_lowest, u32 seam)
{
	const u32 in = is_lowest ? (seam << 24) : below;
	L.Qc[3] = gdw_alignbyte(L.Qc[3], L.Qc[2], 3);
	L.Qc[2] = gdw_alignbyte(L.Qc[2], L.Qc[1], 3);
	L.Qc[1] = gdw_alignbyte(L.Qc[1], L.Qc[0], 3);
	L.Qc[0] = gdw_alignbyte(L.Qc[0], in, 3);
}

// rewrite the persistent score bytes of this lane (every lane, active or not: the rewritten range may spill into
// the first block above the computed window).  any_tn: some lane of the wavefront holds a target N (uniform).
GDW_HD void gdw_update_scores(WaveLane &L, const WaveK &K, bool any_tn)
{
	// index of the AVX-512 score table (SR/ksw2_extd2_avx.c:183-209) folded into 3 bits: 0 match, 1-3 mismatch,
	// 4-7 sc_N.  target ^ query is already right except when both are N (4 ^ 4 = 0): set bit 2 when the target is N
	// and the query byte is not the reverse-complemented N (7 = 4 ^ 3, which the table scores as a mismatch vs N).
	// Without a target N in the block the fix-up term is zero, so wavefronts without any (almost all) skip it.
	if (any_tn) {
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const u32 x = (L.Tb[g] ^ L.Qc[g]) | (L.Tb[g] & ~(L.Qc[g] << 1) & 0x04040404u);
			L.Sb[g] = gdw_perm(gdw_perm(K.lut_hi, K.lut_lo, x), L.Sb[g], L.SEL[g]);
		}
	} else {
#pragma unroll
		for (int g = 0; g < 4; ++g) L.Sb[g] = gdw_perm(gdw_perm(K.lut_hi, K.lut_lo, L.Tb[g] ^ L.Qc[g]), L.Sb[g], L.SEL[g]);
	}
}

// reset of cell t == r (u, y, y2), only while en >= r (:160-163)
GDW_HD void gdw_reset_tr(WaveLane &L, const WaveK &K, const WaveRow &W)
{
	if (L.blk != (W.r >> 4)) return;
	const int k = W.r & 7;
	const u32 mask = (W.r & 8) ? 0xffff0000u : 0x0000ffffu;
	const u32 uk = gdw_pack2(W.ukey);
#pragma unroll
	for (int kk = 0; kk < 8; ++kk)
		if (kk == k) {
			L.U[kk] = gdw_bfi(mask, uk, L.U[kk]);
			L.Y[kk] = gdw_bfi(mask, K.cy, L.Y[kk]);
			L.Y2[kk] = gdw_bfi(mask, K.cy2, L.Y2[kk]);
		}
}

// One anti-diagonal for one ACTIVE lane.  pX/pV/pX2: register 7 of X/V/X2 of the previous lane (row r-1 values).
// bt: the lane's 16 backtrace bytes of this row (4 dwords; byte 4g+h = cell 2g+(h&1)+8*(h>>1)).
// backtrace byte = (4 - d) | nY2<<4 | nX2<<5 | nY<<6 | nX<<7, n* = "no continuation" (inverse of :263-272); bit 3 is undefined.
// DUAL = false: the single-affine recurrence of ksw_extz2 (K3).  With both gap models equal the second pair (a2, b2) can never
// win the priority chain (equal value, lower tie code) and its continuation flags are never read by the backtrack, so the whole
// X2 / Y2 half is dropped; the two flag bits are stored as "no continuation".
template <bool DUAL = true>
GDW_HD void gdw_compute(WaveLane &L, const WaveK &K, const WaveRow &W, u32 pX, u32 pV, u32 pX2, u32 bt[4])
{
	if (!W.use_array) { // first computed block: boundary scalars x1, v1, x21 (:149-159) instead of what the lane below handed over
		const u32 first = W.m_first_valid ? W.m_first : (L.blk == W.st_ ? ~0u : 0u); // (only the upper halves of the incoming dwords are looked at)
		pX = gdw_bfi_s(first, K.cx, pX), pV = gdw_bfi_s(first, gdw_pack2(W.v1key), pV);
		if (DUAL) pX2 = gdw_bfi_s(first, K.cx2, pX2);
	}
	const u32 inX = gdw_alignbit(L.X[7], pX, 16), inV = gdw_alignbit(L.V[7], pV, 16), inX2 = DUAL ? gdw_alignbit(L.X2[7], pX2, 16) : 0u;
	u32 zk_hi = 0;
#pragma unroll
	for (int k = 7; k >= 0; --k) {
		const u32 xin = k ? L.X[k - 1] : inX, vin = k ? L.V[k - 1] : inV, x2in = DUAL ? (k ? L.X2[k - 1] : inX2) : 0u;
		const u32 sk = gdw_perm(L.Sb[2 + (k >> 2)], L.Sb[k >> 2], 0x0c000c00u | (u32)(k & 3) | (u32)(4 + (k & 3)) << 16);
		const u32 a = pk_add(xin, vin), b = pk_add(L.Y[k], L.U[k]);
		u32 zk = pk_max(pk_max(sk, a), b), a2 = 0, b2 = 0;
		if (DUAL) {
			a2 = pk_add(x2in, vin), b2 = pk_add(L.Y2[k], L.U[k]);
			zk = pk_max(zk, pk_add(pk_max(a2, b2), K.c2));
		}
		const u32 z8 = pk_min(zk & 0xfff8fff8u, K.zmax);
		const u32 nV = pk_sub(z8, L.U[k]), nU = pk_sub(z8, vin); // (this order: U[k] is last read by nV, so nU can take its register)
		const u32 tE = pk_sub(z8, K.te);
		L.X[k] = pk_max(pk_sub(a, tE), K.cx), L.Y[k] = pk_max(pk_sub(b, tE), K.cy);
		if (DUAL) {
			const u32 tE2 = pk_sub(z8, K.te2);
			L.X2[k] = pk_max(pk_sub(a2, tE2), K.cx2), L.Y2[k] = pk_max(pk_sub(b2, tE2), K.cy2);
		}
		L.U[k] = nU, L.V[k] = nV;
		if (k & 1) zk_hi = zk;
		else {
			// flags of register pairs k (cells k, k+8) and k+1 (cells k+1, k+9) together: v_perm's sign selectors (8..11) turn the
			// sign of every 16-bit key -- "no continuation" -- into a 0x00 / 0xff byte, already in backtrace byte order, so the four
			// arrays merge with three v_bfi and no shift; the low nibble comes from the priority-chain key (bit 3 is undefined)
			const u32 mX = gdw_perm(L.X[k + 1], L.X[k], 0x0b090a08u), mY = gdw_perm(L.Y[k + 1], L.Y[k], 0x0b090a08u);
			u32 f;
			if (DUAL) {
				const u32 mX2 = gdw_perm(L.X2[k + 1], L.X2[k], 0x0b090a08u), mY2 = gdw_perm(L.Y2[k + 1], L.Y2[k], 0x0b090a08u);
				f = gdw_bfi_u(0x80808080u, mX, gdw_bfi_u(0x40404040u, mY, gdw_bfi_u(0x20202020u, mX2, mY2)));
			} else f = gdw_bfi_u(0x80808080u, mX, mY) | 0x30303030u;
			bt[k >> 1] = gdw_bfi_u(0xf0f0f0f0u, f, gdw_perm(zk_hi, zk, 0x06020400u));
		}
	}
}

// Rows [rA, rS), rS - rA even, of a long alignment for which, with m = (r - w + 1) >> 1:
//   r - w + 1 even:  st0 = m, en0 = m + w - 1        r - w + 1 odd:  st0 = m, en0 = m + w
// and neither the first block, the cell t == r, nor the last target column is involved (st0 >= 16, (en0 | 15) < r, en0 < tlen - 1):
//   st0 = max(0, r - qlen + 1, (r - w + 1) >> 1) is its last term iff r <= 2 qlen - w - 2 (and r >= w - 1),
//   en0 = min(tlen - 1, r, (r + w) >> 1) is its last term, and below tlen - 1, iff r < 2 (tlen - 1) - w (and r >= w).
GDW_HD void gdw_steady_rows(int qlen, int tlen, int w, int &rA, int &rS)
{
	rA = w + 48;
	if ((rA - w + 1) & 1) ++rA;
	const int a = 2 * (tlen - 1) - w, b = 2 * qlen - w - 1;
	rS = a < b ? a : b;
	rS = rS > rA ? rA + ((rS - rA) & ~1) : rA;
}

// Is the chunk [r0, r1] of the half-block cone [hb0, hb0 + 63] INTERIOR: every row in the middle part of the alignment
// (gdw_steady_rows) and all 512 cells of the cone, on every row, inside the cells whose scores are rewritten ([st0, up)) and inside the
// computed blocks (<= en_)?  Then no lane ever is the first block, resets a cell or keeps a stale score, and the rows need no band
// arithmetic at all (gdw_cone_row_half_fast).  st0 and en0 never decrease, and up >= st0 + 16 * nblkA on both rows of a pair.
GDW_HD bool gdw_cone_interior_half(int r0, int r1, int hb0, int qlen, int tlen, int w)
{
	int rA, rS;
	gdw_steady_rows(qlen, tlen, w, rA, rS);
	if (r0 < rA || r1 >= rS || r0 <= 0) return false;
	int st0a, en0a, st0b, en0b;
	gd_band(r0, qlen, tlen, w, st0a, en0a);
	gd_band(r1, qlen, tlen, w, st0b, en0b);
	const int nblkA = (w - 1 + 16) >> 4;
	return 8 * hb0 >= st0b && 8 * (hb0 + 64) <= st0a + 16 * nblkA && ((hb0 + 63) >> 1) <= (en0a >> 4);
}
// ---- score tracking ----------------------------------------------------------------------------------------
// sum of the 16 V keys / of the U keys of cells 1..15 of a lane (horizontal step across a block)
GDW_HD int gdw_sum16(const u32 A[8])
{
	u32 s = pk_add(pk_add(pk_add(A[0], A[1]), pk_add(A[2], A[3])), pk_add(pk_add(A[4], A[5]), pk_add(A[6], A[7])));
	return gdw_lo(s) + gdw_hi(s);
}
// what the predecessor lane contributes to the tracker of the block activated above it on this anti-diagonal:
//   8*H(r,16m) = 8*H(r,16(m-1)) + sum_{i=1..15} U_i - sum_{i=0..15} V_i  [+ U_0 of the new block, added there]
GDW_HD int gdw_track_handoff(const WaveLane &L) { return L.R + gdw_sum16(L.U) - gdw_lo(L.U[0]) - gdw_sum16(L.V); }

// key of cell `slot` (0..15) of a packed array
GDW_HD int gdw_cell(const u32 A[8], int slot)
{
	u32 v = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k)
		if (k == (slot & 7)) v = A[k];
	return (slot & 8) ? gdw_hi(v) : gdw_lo(v);
}
// 8*H(r, 16m+sl) from 8*H(r,16m): partial horizontal walk inside one block
GDW_HD int gdw_track_to_slot(const WaveLane &L, int sl)
{
	int acc = L.R;
	for (int i = 0; i < sl; ++i) acc += gdw_cell(L.U, i + 1) - gdw_cell(L.V, i);
	return acc;
}

// ---- host-side admission test --------------------------------------------------------------------------------
// The wave kernel takes an alignment iff (a) every anti-diagonal fits `lanes` blocks incl. the score-row spill,
// (b) the band never empties, (c) each block's tracker can be seeded from the block below when it enters the band
// and the final cell can be reached from cell 0 of the last block (the band must be >= 17 cells wide there).
// The definition: condition (c) block by block (~5.6 us for a 50 kbp alignment -- 70 ms of one thread per ONT mini-batch when the planner
// asked it for every alignment).  Kept as the reference form of gd_wave_geometry_ok below; tests/emul/plan_test.cpp compares the two on
// millions of random geometries.
static inline bool gd_wave_geometry_ok_loop(int qlen, int tlen, int w, int lanes)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	if (qlen < 1 || tlen < 1) return false;
	if (gd_ncol16(qlen, tlen, w) > lanes) return false;
	// (b): for w >= 1 the band of every anti-diagonal is non-empty iff the corner is inside the band
	if (w < 1 || tlen - qlen > w || qlen - tlen > w) return false;
	// (c)
	const int mlast = (tlen - 1) >> 4;
	for (int m = 1; m <= mlast; ++m) {
		int rs = 16 * m > 32 * m - w ? 16 * m : 32 * m - w; // first anti-diagonal with en0 >= 16m
		int s0, e0;
		gd_band(rs, qlen, tlen, w, s0, e0);
		if (e0 != 16 * m || s0 > 16 * (m - 1)) return false;
	}
	{
		int rb = tlen - 1 > 2 * (tlen - 1) - w ? tlen - 1 : 2 * (tlen - 1) - w; // first anti-diagonal with en0 == tlen-1
		int s0, e0;
		gd_band(rb, qlen, tlen, w, s0, e0);
		if (e0 != tlen - 1 || s0 > 16 * mlast) return false;
	}
	return true;
}

// The same in O(1): as functions of the block index m, rs = max(16 m, 32 m - w), the three terms of en0 = min(tlen - 1, rs, (rs + w) >> 1)
// and the three of st0 = max(0, rs - qlen + 1, (rs - w + 1) >> 1) are (floors of) linear functions with breakpoints only where two of them
// cross -- m = w / 16 (rs changes its form), rs = 2 qlen - w - 1 (the two non-constant terms of st0) -- so each of the two conditions,
// an inequality between such a function and 16 m / 16 (m - 1), can only change its truth value next to a breakpoint or at an end of
// [1, mlast]: it is enough to test a few m around each.
static inline bool gd_wave_geometry_ok(int qlen, int tlen, int w, int lanes)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	if (qlen < 1 || tlen < 1) return false;
	if (gd_ncol16(qlen, tlen, w) > lanes) return false;
	if (w < 1 || tlen - qlen > w || qlen - tlen > w) return false;
	const int mlast = (tlen - 1) >> 4;
	auto block_ok = [&](int m) -> bool {
		if (m < 1 || m > mlast) return true;
		const int rs = 16 * m > 32 * m - w ? 16 * m : 32 * m - w;
		int s0, e0;
		gd_band(rs, qlen, tlen, w, s0, e0);
		return e0 == 16 * m && s0 <= 16 * (m - 1);
	};
	const int rx = 2 * qlen - w - 1; // the anti-diagonal on which r - qlen + 1 overtakes (r - w + 1) >> 1
	const int around[4] = {w >> 4, rx > 0 ? (rx + w) / 32 : 0, rx > 0 ? rx / 16 : 0, mlast - 1};
	for (int a = 0; a < 4; ++a)
		for (int d = -2; d <= 2; ++d)
			if (!block_ok(around[a] + d)) return false;
	for (int m = 1; m <= 3; ++m)
		if (!block_ok(m)) return false;
	{
		int rb = tlen - 1 > 2 * (tlen - 1) - w ? tlen - 1 : 2 * (tlen - 1) - w; // first anti-diagonal with en0 == tlen-1
		int s0, e0;
		gd_band(rb, qlen, tlen, w, s0, e0);
		if (e0 != tlen - 1 || s0 > 16 * mlast) return false;
	}
	return true;
}

// ---- half blocks (wide bands on a 96-block ring: ksw_extd2_wave96c_kernel) -------------------------------------------------------
// A band of w = 1300 has 83 blocks in flight.  On the 128-position ring (two blocks per lane) 36 % of the lanes idle; here a lane
// holds ONE block of ring positions 0..63 plus HALF a block (8 cells) of positions 64..95 -- twelve packed registers per state
// array instead of sixteen, 86 % of them busy.  A half block keeps its 8 cells as 4 packed registers, register j = cells (j, j+4), so
// "cell t-1" is again the register before (j = 0: the incoming dword of the half / block below).  Both halves of a block are inside
// or outside the reference's 16-aligned window together; everything that is defined per cell (which scores are rewritten, the
// reset of cell t == r, the boundary scalars of the first computed cell) carries over with the half's first target position
// tb = 16 * blk + 8 * half.
struct WaveHalf {
	u32 U[4], V[4], X[4], Y[4], X2[4], Y2[4];
	u32 Sb[2], Tb[2], Qc[2], SEL[2];
	u32 tn;
	int32_t blk, half;
	int32_t R; // 8*H(r, tb): score tracker at the half's first cell
};

GDW_HD void gdw_load_half(WaveHalf &H, const WaveK &K, int m, int half, int r, const uint8_t *query, int qlen, const uint8_t *target, int tlen)
{
	H.blk = m, H.half = half;
	const int tb = (m << 4) + (half << 3);
#pragma unroll
	for (int k = 0; k < 4; ++k) H.U[k] = K.uv0, H.V[k] = K.uv0, H.X[k] = K.cx, H.Y[k] = K.cy, H.X2[k] = K.cx2, H.Y2[k] = K.cy2;
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		u32 tw = 0, qw = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int t = tb + 4 * g + b;
			tw |= (t < tlen ? (u32)target[t] : 0u) << (8 * b);
			qw |= gdw_qbyte(query, qlen, r - t) << (8 * b);
		}
		H.Tb[g] = tw, H.Qc[g] = qw, H.Sb[g] = K.s0, H.SEL[g] = 0x03020100u;
	}
	H.tn = (H.Tb[0] | H.Tb[1]) & 0x04040404u;
	H.R = 0;
}

GDW_HD void gdw_make_sel_half(WaveHalf &H, int st0, int up)
{
	const int tb = (H.blk << 4) + (H.half << 3);
	int lo = st0 - tb, hi = up - tb;
	lo = lo < 0 ? 0 : lo > 8 ? 8 : lo;
	hi = hi < 0 ? 0 : hi > 8 ? 8 : hi;
	const u32 bm = hi > lo ? (((1u << (hi - lo)) - 1u) << lo) : 0u;
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		const u32 n = (bm >> (4 * g)) & 15u;
		const u32 spread = (n * 0x00204081u) & 0x01010101u;
		H.SEL[g] = 0x03020100u + (spread << 2);
	}
}

// the same with the choice of the lowest block's lane as a 0 / ~0 lane mask (one v_bfi with the scalar seam byte as data)
GDW_HD void gdw_shift_query_m(WaveLane &L, u32 below, u32 m_lowest, u32 seam)
{
	const u32 in = gdw_bfi_s(m_lowest, seam << 24, below);
	L.Qc[3] = gdw_alignbyte(L.Qc[3], L.Qc[2], 3);
	L.Qc[2] = gdw_alignbyte(L.Qc[2], L.Qc[1], 3);
	L.Qc[1] = gdw_alignbyte(L.Qc[1], L.Qc[0], 3);
	L.Qc[0] = gdw_alignbyte(L.Qc[0], in, 3);
}

GDW_HD void gdw_shift_query_half(WaveHalf &H, u32 below, bool is_lowest, u32 seam)
{
	const u32 in = is_lowest ? (seam << 24) : below;
	H.Qc[1] = gdw_alignbyte(H.Qc[1], H.Qc[0], 3);
	H.Qc[0] = gdw_alignbyte(H.Qc[0], in, 3);
}

GDW_HD void gdw_shift_query_half_m(WaveHalf &H, u32 below, u32 m_lowest, u32 seam) // (the lowest half's lane as a 0 / ~0 mask)
{
	const u32 in = gdw_bfi_s(m_lowest, seam << 24, below);
	H.Qc[1] = gdw_alignbyte(H.Qc[1], H.Qc[0], 3);
	H.Qc[0] = gdw_alignbyte(H.Qc[0], in, 3);
}

GDW_HD void gdw_update_scores_half(WaveHalf &H, const WaveK &K, bool any_tn)
{
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		u32 x = H.Tb[g] ^ H.Qc[g];
		if (any_tn) x |= H.Tb[g] & ~(H.Qc[g] << 1) & 0x04040404u;
		H.Sb[g] = gdw_perm(gdw_perm(K.lut_hi, K.lut_lo, x), H.Sb[g], H.SEL[g]);
	}
}

GDW_HD void gdw_reset_tr_half(WaveHalf &H, const WaveK &K, const WaveRow &W)
{
	if (H.blk != (W.r >> 4) || H.half != ((W.r >> 3) & 1)) return;
	const int c = W.r & 7, k = c & 3;
	const u32 mask = (c & 4) ? 0xffff0000u : 0x0000ffffu;
	const u32 uk = gdw_pack2(W.ukey);
#pragma unroll
	for (int kk = 0; kk < 4; ++kk)
		if (kk == k) {
			H.U[kk] = gdw_bfi(mask, uk, H.U[kk]);
			H.Y[kk] = gdw_bfi(mask, K.cy, H.Y[kk]);
			H.Y2[kk] = gdw_bfi(mask, K.cy2, H.Y2[kk]);
		}
}

// one anti-diagonal of an ACTIVE half block.  pX / pV / pX2: the last packed register of X / V / X2 of the (half) block below, row r-1
// values.  BT = false: state and score only (the first pass of the checkpointed form); BT = true: bt[2] receives the half's 8
// backtrace bytes, coded as in gdw_compute, byte 4g + h = cell 2g + (h & 1) + 4 (h >> 1).
template <bool BT = false>
GDW_HD void gdw_compute_half(WaveHalf &H, const WaveK &K, const WaveRow &W, u32 pX, u32 pV, u32 pX2, u32 *bt = nullptr)
{
	if (!W.use_array) {
		const u32 first = W.m_first_valid ? W.m_first_h : ((H.blk == W.st_ && H.half == 0) ? ~0u : 0u);
		pX = gdw_bfi_s(first, K.cx, pX), pV = gdw_bfi_s(first, gdw_pack2(W.v1key), pV), pX2 = gdw_bfi_s(first, K.cx2, pX2);
	}
	const u32 inX = gdw_alignbit(H.X[3], pX, 16), inV = gdw_alignbit(H.V[3], pV, 16), inX2 = gdw_alignbit(H.X2[3], pX2, 16);
	u32 zk_hi = 0;
#pragma unroll
	for (int k = 3; k >= 0; --k) {
		const u32 xin = k ? H.X[k - 1] : inX, vin = k ? H.V[k - 1] : inV, x2in = k ? H.X2[k - 1] : inX2;
		const u32 sk = gdw_perm(H.Sb[1], H.Sb[0], 0x0c000c00u | (u32)k | (u32)(4 + k) << 16);
		const u32 a = pk_add(xin, vin), b = pk_add(H.Y[k], H.U[k]);
		const u32 a2 = pk_add(x2in, vin), b2 = pk_add(H.Y2[k], H.U[k]);
		const u32 zk = pk_max(pk_max(pk_max(sk, a), b), pk_add(pk_max(a2, b2), K.c2));
		const u32 z8 = pk_min(zk & 0xfff8fff8u, K.zmax);
		const u32 nV = pk_sub(z8, H.U[k]), nU = pk_sub(z8, vin);
		const u32 tE = pk_sub(z8, K.te), tE2 = pk_sub(z8, K.te2);
		H.X[k] = pk_max(pk_sub(a, tE), K.cx), H.Y[k] = pk_max(pk_sub(b, tE), K.cy);
		H.X2[k] = pk_max(pk_sub(a2, tE2), K.cx2), H.Y2[k] = pk_max(pk_sub(b2, tE2), K.cy2);
		H.U[k] = nU, H.V[k] = nV;
		if (BT) {
			if (k & 1) zk_hi = zk;
			else {
				const u32 mX = gdw_perm(H.X[k + 1], H.X[k], 0x0b090a08u), mY = gdw_perm(H.Y[k + 1], H.Y[k], 0x0b090a08u);
				const u32 mX2 = gdw_perm(H.X2[k + 1], H.X2[k], 0x0b090a08u), mY2 = gdw_perm(H.Y2[k + 1], H.Y2[k], 0x0b090a08u);
				const u32 f = gdw_bfi_u(0x80808080u, mX, gdw_bfi_u(0x40404040u, mY, gdw_bfi_u(0x20202020u, mX2, mY2)));
				bt[k >> 1] = gdw_bfi_u(0xf0f0f0f0u, f, gdw_perm(zk_hi, zk, 0x06020400u));
			}
		}
	}
}

// a half block out of a block's full record (register k = cells (k, k+8)): cells 8h + j and 8h + j + 4 are the 16-bit halves h of
// registers j and j + 4
GDW_HD u32 gdw_half_pick(u32 rj, u32 rj4, int h) { return h ? (rj >> 16) | (rj4 & 0xffff0000u) : (rj & 0xffffu) | (rj4 << 16); }
GDW_HD void gdw_half_from_lane(const WaveLane &L, int h, WaveHalf &H)
{
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		H.U[j] = gdw_half_pick(L.U[j], L.U[j + 4], h), H.V[j] = gdw_half_pick(L.V[j], L.V[j + 4], h);
		H.X[j] = gdw_half_pick(L.X[j], L.X[j + 4], h), H.Y[j] = gdw_half_pick(L.Y[j], L.Y[j + 4], h);
		H.X2[j] = gdw_half_pick(L.X2[j], L.X2[j + 4], h), H.Y2[j] = gdw_half_pick(L.Y2[j], L.Y2[j + 4], h);
	}
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		H.Sb[g] = h ? L.Sb[2 + g] : L.Sb[g], H.Tb[g] = h ? L.Tb[2 + g] : L.Tb[g];
		H.Qc[g] = h ? L.Qc[2 + g] : L.Qc[g], H.SEL[g] = h ? L.SEL[2 + g] : L.SEL[g];
	}
	H.tn = (H.Tb[0] | H.Tb[1]) & 0x04040404u;
	H.blk = L.blk, H.half = h, H.R = 0;
}
// the pristine half block of any index (also below block 0 or beyond the target), its query bytes those of anti-diagonal rq
GDW_HD void gdw_fresh_half(WaveHalf &H, const WaveK &K, int blk, int half, int rq, const uint8_t *query, int qlen, const uint8_t *target, int tlen)
{
	const int tb = blk * 16 + half * 8;
	H.blk = blk, H.half = half;
#pragma unroll
	for (int k = 0; k < 4; ++k) H.U[k] = K.uv0, H.V[k] = K.uv0, H.X[k] = K.cx, H.Y[k] = K.cy, H.X2[k] = K.cx2, H.Y2[k] = K.cy2;
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		u32 tw = 0, qw = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int t = tb + 4 * g + b;
			tw |= (t >= 0 && t < tlen ? (u32)target[t] : 0u) << (8 * b);
			qw |= gdw_qbyte(query, qlen, rq - t) << (8 * b);
		}
		H.Tb[g] = tw, H.Qc[g] = qw, H.Sb[g] = K.s0, H.SEL[g] = 0x03020100u;
	}
	H.tn = (H.Tb[0] | H.Tb[1]) & 0x04040404u;
	H.R = 0;
}

GDW_HD int gdw_sum8(const u32 A[4])
{
	const u32 s = pk_add(pk_add(A[0], A[1]), pk_add(A[2], A[3]));
	return gdw_lo(s) + gdw_hi(s);
}
// what a half block contributes to the tracker of the (half) block above it: 8*H(r, tb + 8) = 8*H(r, tb) + sum_{i=1..7} U_i - sum_{i=0..7} V_i [+ U_0 there]
GDW_HD int gdw_track_handoff_half(const WaveHalf &H) { return H.R + gdw_sum8(H.U) - gdw_lo(H.U[0]) - gdw_sum8(H.V); }
// key of cell `slot` (0..7) of a half's packed array
GDW_HD int gdw_cell_half(const u32 A[4], int slot)
{
	u32 v = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k)
		if (k == (slot & 3)) v = A[k];
	return (slot & 4) ? gdw_hi(v) : gdw_lo(v);
}
GDW_HD int gdw_track_to_slot_half(const WaveHalf &H, int sl)
{
	int acc = H.R;
	for (int i = 0; i < sl; ++i) acc += gdw_cell_half(H.U, i + 1) - gdw_cell_half(H.V, i);
	return acc;
}
// The snapshots keep the 16-cell layout of WaveLane (register k = cells (k, k+8)) whatever held the block: a half block provides
// the low (half 0) or high (half 1) 16 bits of the eight registers of each array -- cell k of the half is this value.
GDW_HD u32 gdw_half_cell16(const u32 A[4], int k) { return (k & 4) ? (A[k & 3] >> 16) : (A[k & 3] & 0xffffu); }
