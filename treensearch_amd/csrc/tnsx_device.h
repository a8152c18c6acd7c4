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
}  // namespace tnsx
