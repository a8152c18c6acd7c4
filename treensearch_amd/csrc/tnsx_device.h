// Small wave64 device helpers shared by the gfx950 kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tnsx {

static constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
// number of set bits of m in lanes below the calling lane
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint32_t readfirstlane_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// orders this wave's LDS traffic (all lanes of a wave execute an LDS instruction together, in program order)
__device__ __forceinline__ void wave_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// cell coordinate of a point along one axis: the one quantisation every kernel uses (build, query), so a point is always found
// in the cell it was sorted into
__device__ __forceinline__ int bin_coord(float p, float o, float inv_h, int n)
{
	// fp32 sub, mul, truncate, clamp -- the quantisation form of TreeNSearch.cpp:713-715
	const float f = __fmul_rn(__fsub_rn(p, o), inv_h);
	int c = (int)f;
	c = c < 0 ? 0 : c;
	return c > n - 1 ? n - 1 : c;
}
// sparse grid: {first, one past last} sorted position of the cell with this key, or (0, 0) -- occ is the key-ordered list of occupied cells with a sentinel
// behind it, blk[b] the first entry whose key is >= b << shift (tnsx_build.hip, k_sparse_blocks)
__device__ __forceinline__ uint2 sparse_find(const uint2* __restrict__ occ, const uint32_t* __restrict__ blk, int shift, uint32_t key)
{
	const uint32_t b = key >> shift;
	uint32_t lo = blk[b], hi = blk[b + 1u];
	while (lo < hi) {                       // lower bound of key in occ[lo, hi)
		const uint32_t mid = (lo + hi) >> 1;
		if (occ[mid].y < key) lo = mid + 1u; else hi = mid;
	}
	const uint2 e = occ[lo], nx = occ[lo + 1u];   // (lo <= n_occ: the sentinel's key is 0xffffffff; the slot behind it is allocated)
	return e.y == key ? make_uint2(e.x, nx.x) : make_uint2(0u, 0u);
}
// spreads the low 21 bits of v to every third bit (libmorton's 3-D encoding: x -> bit 0, y -> bit 1, z -> bit 2)
__device__ __forceinline__ uint64_t spread3(uint64_t v)
{
	v &= 0x1fffffull;
	v = (v | (v << 32)) & 0x1f00000000ffffull;
	v = (v | (v << 16)) & 0x1f0000ff0000ffull;
	v = (v | (v << 8)) & 0x100f00f00f00f00full;
	v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
	v = (v | (v << 2)) & 0x1249249249249249ull;
	return v;
}
// Wave-wide bounding box: min of (x0,y0,z0) and max of (x1,y1,z1) over the 64 lanes, returned wave-uniform.  Six values are
// reduced together, one DPP step at a time, so that the five instructions between a write and its DPP read cover the two
// wait states gfx9 needs there (the compiler does not look into inline asm).  v_min/v_max ignore NaN operands.
__device__ __forceinline__ void wave_bbox(float& x0, float& y0, float& z0, float& x1, float& y1, float& z1)
{
#define TNSX_DPP_STEP(pre, ctl)                                                                                          \
	asm volatile(pre "v_min_f32_dpp %0, %0, %0 " ctl "\n\tv_min_f32_dpp %1, %1, %1 " ctl "\n\tv_min_f32_dpp %2, %2, %2 " ctl "\n\t" \
	                 "v_max_f32_dpp %3, %3, %3 " ctl "\n\tv_max_f32_dpp %4, %4, %4 " ctl "\n\tv_max_f32_dpp %5, %5, %5 " ctl          \
	             : "+v"(x0), "+v"(y0), "+v"(z0), "+v"(x1), "+v"(y1), "+v"(z1))
	TNSX_DPP_STEP("s_nop 1\n\t", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
	TNSX_DPP_STEP("", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
	TNSX_DPP_STEP("", "row_half_mirror row_mask:0xf bank_mask:0xf");
	TNSX_DPP_STEP("", "row_mirror row_mask:0xf bank_mask:0xf");
	TNSX_DPP_STEP("", "row_bcast:15 row_mask:0xa bank_mask:0xf");
	TNSX_DPP_STEP("", "row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef TNSX_DPP_STEP
	asm volatile("s_nop 1" ::: );
	x0 = readlane_f32(x0, 63); y0 = readlane_f32(y0, 63); z0 = readlane_f32(z0, 63);
	x1 = readlane_f32(x1, 63); y1 = readlane_f32(y1, 63); z1 = readlane_f32(z1, 63);
}
// the same over lanes 0..15 only (a cell's query points when there are at most sixteen of them: the usual case): four DPP steps instead of six
__device__ __forceinline__ void wave_bbox16(float& x0, float& y0, float& z0, float& x1, float& y1, float& z1)
{
#define TNSX_DPP_STEP(pre, ctl)                                                                                          \
	asm volatile(pre "v_min_f32_dpp %0, %0, %0 " ctl "\n\tv_min_f32_dpp %1, %1, %1 " ctl "\n\tv_min_f32_dpp %2, %2, %2 " ctl "\n\t" \
	                 "v_max_f32_dpp %3, %3, %3 " ctl "\n\tv_max_f32_dpp %4, %4, %4 " ctl "\n\tv_max_f32_dpp %5, %5, %5 " ctl          \
	             : "+v"(x0), "+v"(y0), "+v"(z0), "+v"(x1), "+v"(y1), "+v"(z1))
	TNSX_DPP_STEP("s_nop 1\n\t", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
	TNSX_DPP_STEP("", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
	TNSX_DPP_STEP("", "row_half_mirror row_mask:0xf bank_mask:0xf");
	TNSX_DPP_STEP("", "row_mirror row_mask:0xf bank_mask:0xf");
#undef TNSX_DPP_STEP
	asm volatile("s_nop 1" ::: );
	x0 = readlane_f32(x0, 0); y0 = readlane_f32(y0, 0); z0 = readlane_f32(z0, 0);
	x1 = readlane_f32(x1, 0); y1 = readlane_f32(y1, 0); z1 = readlane_f32(z1, 0);
}
// sum of v over the lanes 0..31 whose bit is set in `lanes` (wave-uniform result): one select, four DPP steps inside the rows of 16, two readlanes
__device__ __forceinline__ uint32_t wave_sum32_masked(uint32_t v, uint64_t lanes)
{
	uint32_t t;
	asm volatile("v_cndmask_b32 %0, 0, %1, %2\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "=&v"(t) : "v"(v), "s"(lanes));
	return readlane_u32(t, 0) + readlane_u32(t, 16);
}
// wave-wide maximum of one value (same scheme; s_nop between the dependent DPP steps)
__device__ __forceinline__ float wave_max_dpp(float v)
{
	asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
	             : "+v"(v));
	return readlane_f32(v, 63);
}
}  // namespace tnsx
