// micro-benchmark 2: steady-state issue cost of VALU instruction classes on gfx950 (long kernels, so that the clock has settled).
// hipcc --offload-arch=gfx950 -O3 tools/ub_valu2.hip -o /tmp/ub_valu2 && /tmp/ub_valu2
// Per instruction template: 8 independent dependency chains (registers a0..a7), 16 instructions per loop iteration, `wps` waves
// per SIMD on every SIMD of the chip.  Printed: shader cycles (s_memtime) and ns of wall time per wave-instruction PER SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define DEF_KERNEL(NAME, ASM)                                                                              \
__global__ __launch_bounds__(256) void NAME(unsigned *out, long long *cyc, int iters, unsigned seed)         \
{                                                                                                          \
	unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
	unsigned b = a0 ^ 0x5a5a, c = a0 | 0x1111;                                                                \
	unsigned sg = __builtin_amdgcn_readfirstlane(seed * 77u + 0x80808080u);                                   \
	long long t0 = __builtin_amdgcn_s_memtime();                                                            \
	for (int i = 0; i < iters; ++i) {                                                                        \
		asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sg));  \
	}                                                                                                      \
	long long t1 = __builtin_amdgcn_s_memtime();                                                            \
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                        \
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
}

#define T(n, fmt) fmt
// templates: %n = chain register, %8 / %9 = loop-invariant VGPRs, %10 = SGPR
#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_ADD_S(n) "v_add_u32 %" #n ", %10, %" #n "\n"
#define A_ADD64(n) "v_add_u32_e64 %" #n ", %" #n ", %8\n"
#define A_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_SUBREV(n) "v_subrev_u32 %" #n ", %" #n ", %8\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_ANDK(n) "v_and_b32 %" #n ", 0xfff8fff8, %" #n "\n"
#define A_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_MAXI32(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define A_MAXU32(n) "v_max_u32 %" #n ", %" #n ", %8\n"
#define A_MINU32(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define A_ASHR(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_MAXI16(n) "v_max_i16 %" #n ", %" #n ", %8\n"
#define A_MAXU16(n) "v_max_u16 %" #n ", %" #n ", %8\n"
#define A_ADDU16(n) "v_add_u16 %" #n ", %" #n ", %8\n"
#define A_MAXI16_SDWA(n) "v_max_i16_sdwa %" #n ", %" #n ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define A_ADD_SDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define A_ADD_DPP(n) "v_add_u32_dpp %" #n ", %8, %" #n " wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define A_MOV_DPP(n) "v_mov_b32_dpp %" #n ", %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define A_MOV_DPPROW(n) "v_mov_b32_dpp %" #n ", %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define A_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define A_PKSUB(n) "v_pk_sub_u16 %" #n ", %" #n ", %8\n"
#define A_PKMAXI(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define A_PKMAXU(n) "v_pk_max_u16 %" #n ", %" #n ", %8\n"
#define A_PKMINU(n) "v_pk_min_u16 %" #n ", %" #n ", %8\n"
#define A_PKMAXS(n) "v_pk_max_i16 %" #n ", %" #n ", %10\n"
#define A_PKADDF16(n) "v_pk_add_f16 %" #n ", %" #n ", %8\n"
#define A_PKMAXF16(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n"
#define A_PKMAX3F16(n) "v_pk_maximum3_f16 %" #n ", %" #n ", %8, %9\n"
#define A_PKFMAF16(n) "v_pk_fma_f16 %" #n ", %" #n ", %8, %9\n"
#define A_PKMAD(n) "v_pk_mad_u16 %" #n ", %" #n ", %8, %9\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_PERM_S(n) "v_perm_b32 %" #n ", %" #n ", %8, %10\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define A_BFI_S(n) "v_bfi_b32 %" #n ", %10, %" #n ", %9\n"
#define A_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0xd8\n"
#define A_BITOP3_S(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %10 bitop3:0xd8\n"
#define A_BITOP3_16(n) "v_bitop3_b16 %" #n ", %" #n ", %8, %9 bitop3:0xd8\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 16\n"
#define A_ALIGNBYTE(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 3\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %9\n"
#define A_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n"
#define A_MAX3I32(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n"
#define A_MAX3I16(n) "v_max3_i16 %" #n ", %" #n ", %8, %9\n"
#define A_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define A_MADI24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define A_MULU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_SADU16(n) "v_sad_u16 %" #n ", %" #n ", %8, %9\n"
#define A_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define A_PKFMAF32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
// mixtures (two instructions per slot: the printed cost is per PAIR)
#define A_MIX_PKADD_ADD(n) "v_pk_max_i16 %" #n ", %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %9\n"
#define A_MIX_PK_PK(n) "v_pk_max_i16 %" #n ", %" #n ", %8\nv_pk_add_u16 %" #n ", %" #n ", %9\n"
#define A_MIX_ADD_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\nv_sub_u32 %" #n ", %" #n ", %9\n"
#define A_MIX_PK_BITOP(n) "v_pk_max_i16 %" #n ", %" #n ", %8\nv_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0xd8\n"
#define A_MIX_PK_PERM(n) "v_pk_max_i16 %" #n ", %" #n ", %8\nv_perm_b32 %" #n ", %" #n ", %8, %9\n"

#define F1(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define S1(n) "v_pk_max_i16 %" #n ", %" #n ", %9\n"
// runs across the 8 chains: RUNxy = x fast ops (chains 0..x-1 ...) -- built below as whole-iteration bodies via the slot macro
#define A_RUN_F(n) F1(n)
#define A_RUN_S(n) S1(n)
#define A_F2S2(n) F1(n) F1(n) S1(n) S1(n)
#define A_F4S4(n) F1(n) F1(n) F1(n) F1(n) S1(n) S1(n) S1(n) S1(n)
#define A_F1S3(n) F1(n) S1(n) S1(n) S1(n)
#define A_F3S1(n) F1(n) F1(n) F1(n) S1(n)
#define KLIST(X) \
	X(k_add, A_ADD, "v_add_u32 (VOP2)", 1) X(k_add_s, A_ADD_S, "v_add_u32 sgpr src", 1) X(k_add64, A_ADD64, "v_add_u32_e64 (VOP3)", 1) \
	X(k_sub, A_SUB, "v_sub_u32", 1) X(k_subrev, A_SUBREV, "v_subrev_u32", 1) X(k_and, A_AND, "v_and_b32", 1) X(k_andk, A_ANDK, "v_and_b32 literal", 1) \
	X(k_or, A_OR, "v_or_b32", 1) X(k_xor, A_XOR, "v_xor_b32", 1) X(k_maxi32, A_MAXI32, "v_max_i32", 1) X(k_maxu32, A_MAXU32, "v_max_u32", 1) \
	X(k_minu32, A_MINU32, "v_min_u32", 1) X(k_lshl, A_LSHL, "v_lshlrev_b32", 1) X(k_lshr, A_LSHR, "v_lshrrev_b32", 1) X(k_ashr, A_ASHR, "v_ashrrev_i32", 1) \
	X(k_mov, A_MOV, "v_mov_b32", 1) X(k_cnd, A_CNDMASK, "v_cndmask_b32", 1) X(k_maxi16, A_MAXI16, "v_max_i16 (VOP2)", 1) X(k_maxu16, A_MAXU16, "v_max_u16 (VOP2)", 1) \
	X(k_addu16, A_ADDU16, "v_add_u16 (VOP2)", 1) X(k_maxi16s, A_MAXI16_SDWA, "v_max_i16_sdwa WORD_1", 1) X(k_addsdwa, A_ADD_SDWA, "v_add_u32_sdwa", 1) \
	X(k_adddpp, A_ADD_DPP, "v_add_u32_dpp wave_ror:1", 1) X(k_movdpp, A_MOV_DPP, "v_mov_b32_dpp wave_ror:1", 1) X(k_movdppr, A_MOV_DPPROW, "v_mov_b32_dpp row_ror:1", 1) \
	X(k_pkadd, A_PKADD, "v_pk_add_u16", 1) X(k_pksub, A_PKSUB, "v_pk_sub_u16", 1) X(k_pkmaxi, A_PKMAXI, "v_pk_max_i16", 1) X(k_pkmaxu, A_PKMAXU, "v_pk_max_u16", 1) \
	X(k_pkminu, A_PKMINU, "v_pk_min_u16", 1) X(k_pkmaxs, A_PKMAXS, "v_pk_max_i16 sgpr src", 1) X(k_pkaddf, A_PKADDF16, "v_pk_add_f16", 1) X(k_pkmaxf, A_PKMAXF16, "v_pk_max_f16", 1) \
	X(k_pkmax3f, A_PKMAX3F16, "v_pk_maximum3_f16", 1) X(k_pkfmaf, A_PKFMAF16, "v_pk_fma_f16", 1) X(k_pkmad, A_PKMAD, "v_pk_mad_u16", 1) \
	X(k_perm, A_PERM, "v_perm_b32", 1) X(k_perms, A_PERM_S, "v_perm_b32 sgpr sel", 1) X(k_bfi, A_BFI, "v_bfi_b32", 1) X(k_bfis, A_BFI_S, "v_bfi_b32 sgpr mask", 1) \
	X(k_bitop3, A_BITOP3, "v_bitop3_b32", 1) X(k_bitop3s, A_BITOP3_S, "v_bitop3_b32 sgpr src", 1) X(k_bitop16, A_BITOP3_16, "v_bitop3_b16", 1) \
	X(k_alignbit, A_ALIGNBIT, "v_alignbit_b32", 1) X(k_alignbyte, A_ALIGNBYTE, "v_alignbyte_b32", 1) X(k_add3, A_ADD3, "v_add3_u32", 1) X(k_lshladd, A_LSHLADD, "v_lshl_add_u32", 1) \
	X(k_xad, A_XAD, "v_xad_u32", 1) X(k_max3i32, A_MAX3I32, "v_max3_i32", 1) X(k_max3i16, A_MAX3I16, "v_max3_i16", 1) X(k_med3, A_MED3, "v_med3_i32", 1) \
	X(k_madi24, A_MADI24, "v_mad_i32_i24", 1) X(k_mulu24, A_MULU24, "v_mul_u32_u24", 1) X(k_sadu16, A_SADU16, "v_sad_u16", 1) X(k_fma, A_FMA, "v_fma_f32", 1) \
	X(k_fmac, A_FMAC, "v_fmac_f32 (VOP2)", 1) X(k_addf32, A_PKFMAF32, "v_add_f32", 1) \
	X(k_mix1, A_MIX_PKADD_ADD, "PAIR pk_max_i16 + v_add_u32", 2) X(k_mix2, A_MIX_PK_PK, "PAIR pk_max_i16 + pk_add_u16", 2) X(k_mix3, A_MIX_ADD_ADD, "PAIR v_add_u32 + v_sub_u32", 2) \
	X(k_f2s2, A_F2S2, "QUAD 2 add_u32 + 2 pk_max (same chain)", 4) X(k_f4s4, A_F4S4, "OCT 4 add_u32 + 4 pk_max (same chain)", 8) \
	X(k_f1s3, A_F1S3, "QUAD 1 add_u32 + 3 pk_max", 4) X(k_f3s1, A_F3S1, "QUAD 3 add_u32 + 1 pk_max", 4) \
	X(k_mix4, A_MIX_PK_BITOP, "PAIR pk_max_i16 + v_bitop3", 2) X(k_mix5, A_MIX_PK_PERM, "PAIR pk_max_i16 + v_perm", 2)

#define X(name, asm_, label, n) DEF_KERNEL(name, asm_)
KLIST(X)
#undef X

typedef void (*kfn)(unsigned *, long long *, int, unsigned);

int main(int argc, char **argv)
{
	struct KD { const char *name; kfn f; int n; };
#define X(name, asm_, label, n) {label, name, n},
	KD K[] = {KLIST(X)};
#undef X
	const int iters = argc > 1 ? atoi(argv[1]) : 60000, per_iter = 16;
	unsigned *d_out; long long *d_cyc;
	CHK(hipMalloc(&d_out, 256 * 4 * 8 * 256 * 4 * 2));
	CHK(hipMalloc(&d_cyc, 256 * 4 * 8 * 8 * 8));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	// warm the clock up
	for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(K[0].f, dim3(256 * 8), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
	CHK(hipDeviceSynchronize());
	printf("%-32s %16s %16s %16s   (shader cycles / ns per wave-instruction per SIMD; PAIR rows: per pair)\n", "instruction", "2 waves/SIMD", "5 waves/SIMD", "8 waves/SIMD");
	for (auto &k : K) {
		printf("%-32s", k.name);
		for (int wps : {2, 5, 8}) {
			const int blocks = 256 * wps; // 256 threads = 4 waves = one per SIMD of a CU
			hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters / 4, 1u);
			CHK(hipDeviceSynchronize());
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
			CHK(hipEventRecord(e1));
			CHK(hipDeviceSynchronize());
			std::vector<long long> c(blocks * 4);
			CHK(hipMemcpy(c.data(), d_cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
			double s = 0; for (auto v : c) s += v;
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
			printf("   %5.2f c %5.2f ns", s / c.size() / ((double)iters * per_iter) / wps, ms * 1e6 / ((double)iters * per_iter * wps));
		}
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
