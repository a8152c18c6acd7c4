// Micro-benchmark: do VALU and SALU instructions of co-resident waves overlap on a gfx950 SIMD, or do they share issue slots?
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed, int sseed)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const float c = seed * 0.5f;
	int s0 = sseed, s1 = sseed + 1, s2 = sseed + 2, s3 = sseed + 3, s4 = sseed + 4, s5 = sseed + 5, s6 = sseed + 6, s7 = sseed + 7;
	for (int i = 0; i < ITERS; i++) {
		if (MODE == 0) {        // 8 VALU
			asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
		else if (MODE == 1) {   // 8 SALU
			asm volatile("s_add_i32 %0, %0, 3\n s_add_i32 %1, %1, 3\n s_add_i32 %2, %2, 3\n s_add_i32 %3, %3, 3\n s_add_i32 %4, %4, 3\n s_add_i32 %5, %5, 3\n s_add_i32 %6, %6, 3\n s_add_i32 %7, %7, 3"
			             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
		}
		else if (MODE == 2) {   // 8 VALU + 8 SALU interleaved
			asm volatile("v_mul_f32 %0, %0, %16\n s_add_i32 %8, %8, 3\n v_mul_f32 %1, %1, %16\n s_add_i32 %9, %9, 3\n v_mul_f32 %2, %2, %16\n s_add_i32 %10, %10, 3\n v_mul_f32 %3, %3, %16\n s_add_i32 %11, %11, 3\n"
			             "v_mul_f32 %4, %4, %16\n s_add_i32 %12, %12, 3\n v_mul_f32 %5, %5, %16\n s_add_i32 %13, %13, 3\n v_mul_f32 %6, %6, %16\n s_add_i32 %14, %14, 3\n v_mul_f32 %7, %7, %16\n s_add_i32 %15, %15, 3"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
			               "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(c) : "scc");
		}
		else if (MODE == 3) {   // 8 slow VALU (v_cmp -> sgpr) + 8 SALU
			asm volatile("v_cmp_ge_f32 s[40:41], %0, %16\n s_add_i32 %8, %8, 3\n v_cmp_ge_f32 s[42:43], %1, %16\n s_add_i32 %9, %9, 3\n v_cmp_ge_f32 s[44:45], %2, %16\n s_add_i32 %10, %10, 3\n v_cmp_ge_f32 s[46:47], %3, %16\n s_add_i32 %11, %11, 3\n"
			             "v_cmp_ge_f32 s[48:49], %4, %16\n s_add_i32 %12, %12, 3\n v_cmp_ge_f32 s[50:51], %5, %16\n s_add_i32 %13, %13, 3\n v_cmp_ge_f32 s[52:53], %6, %16\n s_add_i32 %14, %14, 3\n v_cmp_ge_f32 s[54:55], %7, %16\n s_add_i32 %15, %15, 3"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
			               "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(c)
			             : "scc", "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55");
		}
		else if (MODE == 4) {   // 8 pk VALU + 8 SALU
			asm volatile("v_pk_mul_f32 %0, %0, %0\n s_add_i32 %4, %4, 3\n s_add_i32 %5, %5, 3\n v_pk_mul_f32 %1, %1, %1\n s_add_i32 %6, %6, 3\n s_add_i32 %7, %7, 3\n v_pk_mul_f32 %2, %2, %2\n s_add_i32 %8, %8, 3\n s_add_i32 %9, %9, 3\n v_pk_mul_f32 %3, %3, %3\n s_add_i32 %10, %10, 3\n s_add_i32 %11, %11, 3"
			             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6),
			               "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
}
template <int MODE> void run(const char* name, float* d, int blocks_per_cu)
{
	const int blocks = 256 * blocks_per_cu;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double iters_per_simd = (double)blocks * 4 / (256.0 * 4) * ITERS;   // loop iterations executed per SIMD
	printf("%-44s waves/SIMD %d  %8.3f ms -> %.1f cycles per loop iteration per SIMD @2.4GHz\n", name, blocks_per_cu, ms, ms * 1e6 / iters_per_simd * 2.4);
}
int main()
{
	float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
	for (int w : {8, 4, 1}) {
		if (w == 8) { run<0>("8 v_mul", d, 8); run<1>("8 s_add", d, 8); run<2>("8 v_mul + 8 s_add", d, 8); run<3>("8 v_cmp->sgpr + 8 s_add", d, 8); run<4>("4 v_pk_mul + 8 s_add", d, 8); }
		if (w == 4) { run<0>("8 v_mul", d, 4); run<1>("8 s_add", d, 4); run<2>("8 v_mul + 8 s_add", d, 4); }
		if (w == 1) { run<0>("8 v_mul", d, 1); run<1>("8 s_add", d, 1); run<2>("8 v_mul + 8 s_add", d, 1); }
	}
	return 0;
}
