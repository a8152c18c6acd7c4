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

}  // namespace tnsx
