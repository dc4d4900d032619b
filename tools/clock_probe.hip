// What do s_memtime / s_memrealtime count on gfx950, and what shader clock does the chip sustain under the DP kernel's kind of load?
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
// Every wavefront stamps both counters at its start and end; the host brackets the launch with HIP events.  s_memrealtime is the
// constant 100 MHz reference clock (10 ns per tick: checked against the event time below), so
//     ticks(s_memtime) / ticks(s_memrealtime) x 100 MHz  =  the frequency of whatever s_memtime counts.
// Loads: (a) one wavefront per CU sleeping (s_sleep: the chip is idle), (b) one wavefront per CU in a dependent v_add_u32 chain,
// (c) every SIMD with 5 wavefronts of a v_pk_max_i16 / v_pk_add_u16 / v_perm_b32 stream (the DP kernel's instruction class),
// (d) the same at 8 wavefronts per SIMD, (e) v_add_u32 (first class) at 5 per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long mt0, rt0, mt1, rt1; unsigned hw_id, xcc_id; };

template <int MODE> __global__ __launch_bounds__(256) void probe(Stamp *st, unsigned *out, int iters)
{
	unsigned a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19, b = a0 ^ 0x5a5a, c = a0 | 0x01020304;
	const unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
	for (int i = 0; i < iters; ++i) {
		if (MODE == 0) __builtin_amdgcn_s_sleep(64);
		else if (MODE == 1) asm volatile("v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n" : "+v"(a0) : "v"(b));
		else if (MODE == 2)
			asm volatile("v_pk_max_i16 %0, %0, %8\nv_pk_add_u16 %1, %1, %8\nv_perm_b32 %2, %2, %8, %9\nv_pk_max_i16 %3, %3, %8\nv_pk_sub_u16 %4, %4, %8\nv_pk_max_i16 %5, %5, %8\n"
			             "v_bfi_b32 %6, %8, %6, %9\nv_pk_add_u16 %7, %7, %8\nv_pk_max_i16 %0, %0, %8\nv_pk_add_u16 %1, %1, %8\nv_perm_b32 %2, %2, %8, %9\nv_pk_max_i16 %3, %3, %8\n"
			             "v_pk_sub_u16 %4, %4, %8\nv_pk_max_i16 %5, %5, %8\nv_bfi_b32 %6, %8, %6, %9\nv_pk_add_u16 %7, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
		else
			asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\n"
			             "v_add_u32 %7, %7, %8\nv_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\n"
			             "v_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
	}
	const unsigned long long mt1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
	unsigned hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\ns_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc)); // where the wavefront ran: SIMD [5:4], CU [11:8], SH [12], SE [15:13]; XCC [3:0]
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if ((threadIdx.x & 63) == 0) { Stamp s = {mt0, rt0, mt1, rt1, hw, xcc}; st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s; }
}

typedef void (*kfn)(Stamp *, unsigned *, int);
int main()
{
	Stamp *d_st; unsigned *d_out;
	const int max_waves = 256 * 4 * 8 * 2;
	CHK(hipMalloc(&d_st, sizeof(Stamp) * max_waves));
	CHK(hipMalloc(&d_out, sizeof(unsigned) * max_waves * 64));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	struct Run { const char *what; kfn f; int blocks, threads, iters, per_iter; } runs[] = {
		{"(a) idle: 1 wavefront / CU, s_sleep", probe<0>, 256, 64, 200000, 0},
		{"(b) light: 1 wavefront / CU, dependent v_add_u32", probe<1>, 256, 64, 4000000, 4},
		{"(c) DP-class stream, 5 wavefronts / SIMD, all CUs", probe<2>, 256 * 5, 256, 1500000, 16},
		{"(d) DP-class stream, 8 wavefronts / SIMD, all CUs", probe<2>, 256 * 8, 256, 1000000, 16},
		{"(e) v_add_u32 stream, 5 wavefronts / SIMD, all CUs", probe<3>, 256 * 5, 256, 1500000, 16},
		{"(f) DP-class stream, 5120 one-wavefront workgroups", probe<2>, 5120, 64, 1500000, 16},
		{"(g) DP-class stream, 8192 one-wavefront workgroups", probe<2>, 8192, 64, 1000000, 16},
		{"(h) v_add_u32 stream, 8192 one-wavefront workgroups", probe<3>, 8192, 64, 1000000, 16},
		{"(i) DP-class stream, 1024 one-wavefront workgroups", probe<2>, 1024, 64, 3000000, 16},
		{"(j) DP-class stream, 2048 one-wavefront workgroups", probe<2>, 2048, 64, 3000000, 16},
	};
	for (auto &R : runs) {
		for (int rep = 0; rep < 2; ++rep) { // the second launch is the measurement (clock settled on the load)
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL(R.f, dim3(R.blocks), dim3(R.threads), 0, 0, d_st, d_out, R.iters);
			CHK(hipEventRecord(e1));
			CHK(hipDeviceSynchronize());
		}
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		const int nw = R.blocks * (R.threads / 64);
		std::vector<Stamp> st(nw);
		CHK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
		std::vector<double> f, rt_ms;
		unsigned long long rt_lo = ~0ull, rt_hi = 0;
		for (auto &s : st) {
			f.push_back((double)(s.mt1 - s.mt0) / (double)(s.rt1 - s.rt0) * 100.0);
			rt_ms.push_back((double)(s.rt1 - s.rt0) * 1e-5);
			rt_lo = std::min(rt_lo, s.rt0), rt_hi = std::max(rt_hi, s.rt1);
		}
		std::sort(f.begin(), f.end()); std::sort(rt_ms.begin(), rt_ms.end());
		const double insts = R.per_iter ? (double)R.iters * R.per_iter : 0;
		printf("%-52s events %8.2f ms | s_memrealtime span %8.2f ms (x 10 ns) | s_memtime / s_memrealtime: min %7.1f median %7.1f max %7.1f MHz", R.what, ms,
		       (double)(rt_hi - rt_lo) * 1e-5, f.front(), f[f.size() / 2], f.back());
		if (insts > 0) {
			// the robust figure: all instructions of the launch over its duration, per SIMD (assumes nothing about where the wavefronts ran)
			const double ns_total = (double)ms * 1e6 / (insts * nw / 1024.0);
			printf(" | all instructions / kernel time: %.2f ns = %.2f cycles per wave-instruction per SIMD", ns_total, ns_total * f[f.size() / 2] * 1e-3);
			// where the wavefronts ran, and for how long each
			std::vector<int> per_simd(8 * 8 * 2 * 16 * 4, 0);
			for (auto &s : st) per_simd[(((s.xcc_id & 7) * 8 + ((s.hw_id >> 13) & 7)) * 2 + ((s.hw_id >> 12) & 1)) * 64 + ((s.hw_id >> 8) & 15) * 4 + ((s.hw_id >> 4) & 3)]++;
			int used = 0, mx = 0; std::vector<int> hist(64, 0);
			for (int v : per_simd) if (v) { ++used; mx = std::max(mx, v); hist[std::min(v, 63)]++; }
			printf("\n      wavefront duration p10 / p50 / p90 / max: %.1f / %.1f / %.1f / %.1f ms; SIMDs used %d; wavefronts per SIMD (over the whole launch):", rt_ms[rt_ms.size() / 10], rt_ms[rt_ms.size() / 2],
			       rt_ms[rt_ms.size() * 9 / 10], rt_ms.back(), used);
			for (int v = 1; v <= mx; ++v) if (hist[v]) printf(" %d x%d", v, hist[v]);
		}
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
