// Per-wave bump allocation from the record pool (pool passes of the query kernels); shared by tnsx_query.hip and tools/ubench/tnsx_query_group.hip.
#pragma once
#include "tnsx_kernels.h"
#include "tnsx_device.h"

namespace tnsx {

// Per-wave bump allocator over the record pool (MODE_POOL): a wave owns a slab of POOL_SLAB ints at a time and takes a
// new one with ONE atomic when the next record does not fit.  Slab remainders stay unused, so the pool has holes; every
// record is still contiguous and exact.
// The pool is cut into POOL_REGIONS regions, one per XCD, each with its own cursor on its own cache line, for the fast tier (whose
// cells are split among the XCDs by position in the cell list: what an XCD produces changes slowly from run to run), plus one
// common region for the two heavy tiers (their worklists are appended to in any order) that a fast-tier wave also falls back to
// when its XCD's region is full.  One cursor for everybody was the bottleneck of
// every launch that is not huge: the L2 serialises the atomics of a line (~88 per microsecond), 8192 waves that each take
// ~8 slabs are 65 k atomics = 0.74 ms whatever the problem size (C3's fluid->boundary pair: 0.66 ms for 69 k cells; a
// 1 M point query: 0.84 ms instead of 0.28).  Region capacities follow the payload every XCD produced in the previous run.
// All fast-path bookkeeping is 32-bit scalar work: gfx9 has no 64-bit scalar magnitude compare, so a `cur + len > end`
// test on 64-bit values would be done on the VALU (with copies back and forth) for every single query.
struct PoolState {
	uint32_t cur_lo, cur_hi;   // next free int of the wave's slab
	uint32_t left;             // ints left in the slab
	uint32_t ok;               // 1 when the whole slab lies inside the pool (else: count, but do not write)
	uint32_t waste;            // ints of abandoned slab remainders so far (payload of an XCD = what it asked for - what it wasted)
};

// rare path, deliberately out of line so that the per-query fast path stays a handful of scalar instructions.
// Returns the first int of the new slab, or POOL_NONE when it may not be written (valid in lane 0).
static constexpr unsigned long long POOL_NONE = ~0ull;
static __device__ __attribute__((noinline)) unsigned long long pool_take_slab(unsigned long long* cursors, const unsigned long long* regions, uint32_t sz, uint32_t heavy_tier)
{
	unsigned long long first = POOL_NONE;
	if (lane_id() == 0) {
		const uint32_t r = heavy_tier ? (uint32_t)POOL_OVERFLOW : (blockIdx.x & 7u);
		const unsigned long long old = atomicAdd(cursors + (size_t)r * POOL_CURSOR_STRIDE, (unsigned long long)sz);
		if (old + sz <= regions[2 * r + 1]) first = regions[2 * r] + old;
		else if (r != (uint32_t)POOL_OVERFLOW) {
			const unsigned long long cap_o = regions[2 * POOL_OVERFLOW + 1];
			if (cap_o != 0ull) {   // (0: dry pass, nothing is written anywhere)
				const unsigned long long old_o = atomicAdd(cursors + (size_t)POOL_OVERFLOW * POOL_CURSOR_STRIDE, (unsigned long long)sz);
				if (old_o + sz <= cap_o) first = regions[2 * POOL_OVERFLOW] + old_o;
			}
		}
	}
	return first;
}

// end of a wave's work: its neighbour count and its unused ints -> the counters of its region
template <bool HEAVY>
__device__ __forceinline__ void pool_wave_done(const QueryArgs& a, const PoolState& ps, uint32_t wave_hits, int lane)
{
	if (lane == 0) {
		const uint32_t r = HEAVY ? (uint32_t)POOL_OVERFLOW : (blockIdx.x & 7u);
		unsigned long long* line = a.pool_cursor + (size_t)r * POOL_CURSOR_STRIDE;
		const unsigned long long waste = (unsigned long long)ps.waste + ps.left;
		if (wave_hits) atomicAdd(line + POOL_HITS_WORD, (unsigned long long)wave_hits);
		if (waste) atomicAdd(line + POOL_WASTE_WORD, waste);
	}
}

__device__ __forceinline__ void pool_waste(PoolState& ps, uint32_t left)
{
	ps.waste += left;
}

template <bool HEAVY>
__device__ __forceinline__ uint64_t pool_alloc(const QueryArgs& a, PoolState& ps, uint32_t len, int lane, bool& ok)
{
	(void)lane;
	if (len > ps.left) {
		pool_waste(ps, ps.left);
		const uint32_t slab = HEAVY ? a.pool_slab_heavy : a.pool_slab;
		const uint32_t sz = len > slab ? len : slab;
		const unsigned long long first = pool_take_slab(a.pool_cursor, a.pool_regions, sz, HEAVY ? 1u : 0u);
		ps.cur_lo = readfirstlane_u32((uint32_t)first);
		ps.cur_hi = readfirstlane_u32((uint32_t)(first >> 32));
		ps.left = sz;
		ps.ok = (ps.cur_lo & ps.cur_hi) != 0xffffffffu ? 1u : 0u;
		if (ps.ok == 0u) { ps.cur_lo = 0u; ps.cur_hi = 0u; }
	}
	const uint64_t off = ((uint64_t)ps.cur_hi << 32) | ps.cur_lo;
	const uint64_t nxt = off + len;
	ps.cur_lo = (uint32_t)nxt;
	ps.cur_hi = (uint32_t)(nxt >> 32);
	ps.left -= len;
	ok = ps.ok != 0u;
	return off;
}

}  // namespace tnsx
