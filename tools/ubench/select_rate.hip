// Micro-benchmark: cost of the compare+select chain used to map candidate slots to runs (compiler-generated code).
#include <hip/hip_runtime.h>
#include <cstdio>
struct P { unsigned p[8]; unsigned d[9]; };
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned* out, P prm, int iters)
{
	unsigned slot = threadIdx.x + blockIdx.x, acc = 0;
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) {          // 8 x (v_cmp + v_cndmask), scalar operands
			unsigned d = prm.d[0];
			#pragma unroll
			for (int r = 0; r < 8; r++) d = slot >= prm.p[r] ? prm.d[r + 1] : d;
			acc += d; slot += acc & 7;
		}
		else if (MODE == 1) {     // arithmetic alternative: d = d0 + sum (slot >= p_r) * (d_r+1 - d_r)  via v_cmp + v_cndmask(0, diff) + add
			unsigned d = prm.d[0];
			#pragma unroll
			for (int r = 0; r < 8; r++) d += (slot >= prm.p[r]) ? (prm.d[r + 1] - prm.d[r]) : 0u;
			acc += d; slot += acc & 7;
		}
		else if (MODE == 2) {     // min/max formulation: count of thresholds passed via v_min/v_sub... : d = sum max(0, min(1, slot - p_r + 1))*diff (integer, 3 ops each)
			unsigned d = prm.d[0];
			#pragma unroll
			for (int r = 0; r < 8; r++) { const int s = (int)slot - (int)prm.p[r]; const unsigned ge = (unsigned)(~s) >> 31; d += ge * (prm.d[r + 1] - prm.d[r]); }
			acc += d; slot += acc & 7;
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, unsigned* d, P prm)
{
	const int blocks = 256 * 8, iters = 2048;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, prm, iters);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, prm, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double chains_per_simd = (double)blocks * 4 / (256.0 * 4) * iters;
	printf("%-40s %8.3f ms -> %.1f cycles per 8-run chain per wave per SIMD @2.4GHz\n", name, ms, ms * 1e6 / chains_per_simd * 2.4);
}
int main()
{
	unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
	P prm; for (int i = 0; i < 8; i++) prm.p[i] = 40 * (i + 1); for (int i = 0; i < 9; i++) prm.d[i] = 1000 * i + 7;
	run<0>("select chain (cmp+cndmask)", d, prm); run<1>("add chain (cmp+cndmask0+add)", d, prm); run<2>("shift/mul chain (sub,not,lshr,mad)", d, prm);
	return 0;
}
