// micro-benchmark: issue cost of the VALU instructions the ksw wave kernel is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/ub_valu.hip -o /tmp/ub_valu && /tmp/ub_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define DEF_KERNEL(NAME, ASM)                                                                              \
__global__ __launch_bounds__(256) void NAME(unsigned *out, long long *cyc, int iters, unsigned seed)         \
{                                                                                                          \
	unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
	unsigned b = a0 ^ 0x5a5a, c = a0 | 0x1111;                                                                \
	long long t0 = clock64();                                                                               \
	for (int i = 0; i < iters; ++i) {                                                                        \
		asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));  \
	}                                                                                                      \
	long long t1 = clock64();                                                                               \
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                        \
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
}

#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define A_PKSUB(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define A_PKMAX(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define A_PKMIN(n) "v_pk_min_i16 %" #n ", %" #n ", %8\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 16\n"
#define A_ALIGNBYTE(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 3\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define A_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 8, %9\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define A_DPP(n) "v_mov_b32_dpp %" #n ", %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define A_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0xd8\n"
#define A_MAX3(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n"
#define A_PKMAD(n) "v_pk_mad_u16 %" #n ", %" #n ", %8, %9\n"
#define A_PKLSHR(n) "v_pk_lshrrev_b16 %" #n ", 15, %" #n "\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_BFE(n) "v_bfe_i32 %" #n ", %" #n ", 0, 16\n"

DEF_KERNEL(k_add, A_ADD)
DEF_KERNEL(k_pkadd, A_PKADD)
DEF_KERNEL(k_pksub, A_PKSUB)
DEF_KERNEL(k_pkmax, A_PKMAX)
DEF_KERNEL(k_pkmin, A_PKMIN)
DEF_KERNEL(k_perm, A_PERM)
DEF_KERNEL(k_alignbit, A_ALIGNBIT)
DEF_KERNEL(k_alignbyte, A_ALIGNBYTE)
DEF_KERNEL(k_bfi, A_BFI)
DEF_KERNEL(k_andor, A_ANDOR)
DEF_KERNEL(k_lshlor, A_LSHLOR)
DEF_KERNEL(k_and, A_AND)
DEF_KERNEL(k_lshr, A_LSHR)
DEF_KERNEL(k_dpp, A_DPP)
DEF_KERNEL(k_bitop3, A_BITOP3)
DEF_KERNEL(k_max3, A_MAX3)
DEF_KERNEL(k_pkmad, A_PKMAD)
DEF_KERNEL(k_pklshr, A_PKLSHR)
DEF_KERNEL(k_add3, A_ADD3)
DEF_KERNEL(k_bfe, A_BFE)

typedef void (*kfn)(unsigned *, long long *, int, unsigned);

int main()
{
	struct { const char *name; kfn f; } K[] = {{"v_add_u32", k_add}, {"v_pk_add_u16", k_pkadd}, {"v_pk_sub_i16", k_pksub}, {"v_pk_max_i16", k_pkmax},
		{"v_pk_min_i16", k_pkmin}, {"v_perm_b32", k_perm}, {"v_alignbit_b32", k_alignbit}, {"v_alignbyte_b32", k_alignbyte}, {"v_bfi_b32", k_bfi},
		{"v_and_or_b32", k_andor}, {"v_lshl_or_b32", k_lshlor}, {"v_and_b32", k_and}, {"v_lshrrev_b32", k_lshr}, {"v_mov_b32_dpp wave_ror:1", k_dpp},
		{"v_bitop3_b32", k_bitop3}, {"v_max3_i32", k_max3}, {"v_pk_mad_u16", k_pkmad}, {"v_pk_lshrrev_b16", k_pklshr}, {"v_add3_u32", k_add3}, {"v_bfe_i32", k_bfe}};
	const int iters = 4000, per_iter = 16;
	unsigned *d_out; long long *d_cyc;
	CHK(hipMalloc(&d_out, 256 * 4 * 8 * 256 * 4 * 2));
	CHK(hipMalloc(&d_cyc, 256 * 4 * 8 * 8 * 8));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	printf("%-28s %10s %10s %10s %10s %10s %10s  (cycles per wave-instruction seen by ONE wave; x waves/SIMD = SIMD cost)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "5", "6", "8");
	for (auto &k : K) {
		printf("%-28s", k.name);
		for (int wps : {1, 2, 4, 5, 6, 8}) {
			const int blocks = 256 * wps; // 256 threads = 4 waves = one per SIMD of a CU
			hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 10, 1u);
			CHK(hipDeviceSynchronize());
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
			CHK(hipEventRecord(e1));
			CHK(hipDeviceSynchronize());
			std::vector<long long> c(blocks * 4);
			CHK(hipMemcpy(c.data(), d_cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
			double s = 0; for (auto v : c) s += v;
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
			printf(" %5.2f/%4.2fns", s / c.size() / ((double)iters * per_iter) / wps, ms * 1e6 / ((double)iters * per_iter * wps));
		}
		printf("\n");
	}
	return 0;
}
