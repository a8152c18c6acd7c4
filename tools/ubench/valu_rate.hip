// Micro-benchmark: issue cost of the VALU instructions the query kernel is made of (gfx950), 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	v2f b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = b0 + 1.f, b5 = b1 + 1.f, b6 = b2 + 1.f, b7 = b3 + 1.f;
	const float c = seed * 0.5f;
	unsigned cnt = 0;
	for (int i = 0; i < ITERS; i++) {
		if (MODE == 0) {  // 8 x v_mul_f32
			asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
		else if (MODE == 1) {  // 8 x v_pk_mul_f32
			asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
			             : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(b0));
		}
		else if (MODE == 2) {  // 8 x v_fma_f32
			asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
		else if (MODE == 3) {  // 8 x v_pk_fma_f32
			asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
			             : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(b0));
		}
		else if (MODE == 4) {  // 8 x v_cmp_ge_f32 -> sgpr pair
			asm volatile("v_cmp_ge_f32 s[20:21], %0, %8\n v_cmp_ge_f32 s[22:23], %1, %8\n v_cmp_ge_f32 s[24:25], %2, %8\n v_cmp_ge_f32 s[26:27], %3, %8\n v_cmp_ge_f32 s[28:29], %4, %8\n v_cmp_ge_f32 s[30:31], %5, %8\n v_cmp_ge_f32 s[32:33], %6, %8\n v_cmp_ge_f32 s[34:35], %7, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
		}
		else if (MODE == 5) {  // 8 x v_pk_add_f32 with sgpr-pair broadcast operand (op_sel_hi:[0,1]) as in the query kernel
			asm volatile("v_pk_add_f32 %0, s[20:21], %0 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, s[20:21], %1 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, s[20:21], %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, s[20:21], %3 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %4, s[20:21], %4 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5, s[20:21], %5 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %6, s[20:21], %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7, s[20:21], %7 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
			             : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : : "s20", "s21");
		}
		else if (MODE == 6) {  // 8 x v_sub_f32 with sgpr operand
			asm volatile("v_sub_f32 %0, s20, %0\n v_sub_f32 %1, s20, %1\n v_sub_f32 %2, s20, %2\n v_sub_f32 %3, s20, %3\n v_sub_f32 %4, s20, %4\n v_sub_f32 %5, s20, %5\n v_sub_f32 %6, s20, %6\n v_sub_f32 %7, s20, %7"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20");
		}
		else if (MODE == 7) {  // 8 x v_cmp_ge_f32 e32 -> vcc
			asm volatile("v_cmp_ge_f32 vcc, %0, %8\n v_cmp_ge_f32 vcc, %1, %8\n v_cmp_ge_f32 vcc, %2, %8\n v_cmp_ge_f32 vcc, %3, %8\n v_cmp_ge_f32 vcc, %4, %8\n v_cmp_ge_f32 vcc, %5, %8\n v_cmp_ge_f32 vcc, %6, %8\n v_cmp_ge_f32 vcc, %7, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
		}
		else if (MODE == 8) {  // 4 x (v_mbcnt_lo + v_mbcnt_hi) with sgpr masks
			asm volatile("v_mbcnt_lo_u32_b32 %0, s20, 0\n v_mbcnt_hi_u32_b32 %0, s21, %0\n v_mbcnt_lo_u32_b32 %1, s20, 0\n v_mbcnt_hi_u32_b32 %1, s21, %1\n v_mbcnt_lo_u32_b32 %2, s20, 0\n v_mbcnt_hi_u32_b32 %2, s21, %2\n v_mbcnt_lo_u32_b32 %3, s20, 0\n v_mbcnt_hi_u32_b32 %3, s21, %3"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21");
		}
		else if (MODE == 9) {  // 8 x v_cndmask_b32 with sgpr data operand, vcc select
			asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
		}
		else if (MODE == 10) {  // 8 x v_readlane_b32
			asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20","s21","s22","s23","s24","s25","s26","s27");
		}
		else if (MODE == 11) {  // 8 x v_add_u32 (int)
			asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
		else if (MODE == 13) {  // 8 x v_addc_co_u32 with an sgpr-pair carry-in (per-lane hit counting)
			asm volatile("v_addc_co_u32 %0, vcc, 0, %0, s[20:21]\n v_addc_co_u32 %1, vcc, 0, %1, s[20:21]\n v_addc_co_u32 %2, vcc, 0, %2, s[20:21]\n v_addc_co_u32 %3, vcc, 0, %3, s[20:21]\n v_addc_co_u32 %4, vcc, 0, %4, s[20:21]\n v_addc_co_u32 %5, vcc, 0, %5, s[20:21]\n v_addc_co_u32 %6, vcc, 0, %6, s[20:21]\n v_addc_co_u32 %7, vcc, 0, %7, s[20:21]"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc", "s20", "s21");
		}
		else if (MODE == 14) {  // 8 x v_add_u32 dpp row_shr:1 (scan step)
			asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			             "v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %0, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		}
		else if (MODE == 15) {  // 8 x v_cmpx_ge_f32 (writes exec) + restore
			asm volatile("v_cmpx_ge_f32 s[20:21], %0, %8\n s_mov_b64 exec, -1\n v_cmpx_ge_f32 s[22:23], %1, %8\n s_mov_b64 exec, -1\n v_cmpx_ge_f32 s[24:25], %2, %8\n s_mov_b64 exec, -1\n v_cmpx_ge_f32 s[26:27], %3, %8\n s_mov_b64 exec, -1\n"
			             "v_cmpx_ge_f32 s[28:29], %4, %8\n s_mov_b64 exec, -1\n v_cmpx_ge_f32 s[30:31], %5, %8\n s_mov_b64 exec, -1\n v_cmpx_ge_f32 s[32:33], %6, %8\n s_mov_b64 exec, -1\n v_cmpx_ge_f32 s[34:35], %7, %8\n s_mov_b64 exec, -1"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
		}
		else if (MODE == 16) {  // 8 x v_sub_f32 vgpr + v_alignbit (sign collect)
			asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
		else if (MODE == 17) {  // 8 x v_writelane_b32 (sgpr data, m0 lane)
			asm volatile("s_mov_b32 m0, 5\n v_writelane_b32 %0, s20, m0\n v_writelane_b32 %1, s20, m0\n v_writelane_b32 %2, s20, m0\n v_writelane_b32 %3, s20, m0\n v_writelane_b32 %4, s20, m0\n v_writelane_b32 %5, s20, m0\n v_writelane_b32 %6, s20, m0\n v_writelane_b32 %7, s20, m0"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20");
		}
		else if (MODE == 18) {  // 8 x v_pk_mul_f32 with VGPR-pair broadcast operand (op_sel_hi:[0,1]): the query in vector registers
			asm volatile("v_pk_add_f32 %0, %8, %0 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %8, %1 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %8, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %8, %3 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
			             "v_pk_add_f32 %4, %8, %4 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5, %8, %5 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %6, %8, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7, %8, %7 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
			             : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(b0));
		}
		else if (MODE == 19) {  // 8 x v_cndmask_b32 e64 with an sgpr-pair mask
			asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s20", "s21");
		}
		else if (MODE == 20) {  // 8 x v_cndmask_b32 vcc on INDEPENDENT destinations (mode 9 chains nothing either, but reuses its sources)
			asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
			             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(c), "v"(b0.x) : "vcc");
		}
		else if (MODE == 21) {  // 4 x (v_cmp -> vcc, v_cndmask vcc): the compare + select pair of a clamp
			asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");
		}
		else if (MODE == 22) {  // 8 x v_min_f32 (select-free clamp)
			asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
		else if (MODE == 23) {  // 8 x ds_bpermute_b32
			asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
		else if (MODE == 12) {  // 8 x v_add_f32 vgpr operands
			asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b1.y + b2.x + b3.y + b4.x + b5.y + b6.x + b7.y + cnt;
}
template <int MODE> void run(const char* name, float* d)
{
	const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves/SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double instr_per_simd = (double)blocks * 4 / (256.0 * 4) * ITERS * 8;   // wave-instructions per SIMD
	printf("%-34s %8.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
	float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
	run<0>("v_mul_f32", d); run<1>("v_pk_mul_f32", d); run<2>("v_fma_f32", d); run<3>("v_pk_fma_f32", d);
	run<4>("v_cmp_ge_f32 -> sgpr", d); run<5>("v_pk_add_f32 sgpr bcast operand", d); run<6>("v_sub_f32 sgpr operand", d);
	run<7>("v_cmp_ge_f32 e32 -> vcc", d); run<8>("v_mbcnt_lo/hi (per instr)", d); run<9>("v_cndmask_b32 vcc", d); run<10>("v_readlane_b32", d); run<11>("v_add_u32", d); run<12>("v_add_f32 vgpr", d);
	run<13>("v_addc_co_u32 sgpr carry-in", d); run<14>("v_add_u32_dpp row_shr:1", d); run<15>("v_cmpx_ge_f32 + s_mov exec", d); run<16>("v_alignbit_b32", d);
	run<17>("v_writelane_b32", d); run<18>("v_pk_add_f32 vgpr bcast operand", d);
	run<19>("v_cndmask_b32_e64 sgpr mask", d); run<20>("v_cndmask_b32 vcc, indep dst", d); run<21>("v_cmp->vcc + v_cndmask (per pair /2)", d); run<22>("v_min_f32", d); run<23>("ds_bpermute_b32", d);
	return 0;
}
