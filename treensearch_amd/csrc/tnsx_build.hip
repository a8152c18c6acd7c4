// gfx950 kernels: the FAST build of the search structure -- a counting sort by cell.
//
//   k_bin_count    key(p) and rank(p) = atomicAdd(count[key], 1)            12 B read + 8 B written per point
//   scan           start = exclusive scan of count over the grid cells       (tnsx_kernels.hip)
//   k_bin_scatter  xyzi[start[key] + rank] = (x, y, z, original index)       20 B read + 16 B written per point
//   k_cells_from_counts  table[c] = (start[c], start[c+1]) and the KEY-ORDERED list of occupied cells
//
// Compared with the radix-sort build (cell keys -> 3 LSD passes over (key, idx) pairs -> random gather -> cell table) this
// moves every point twice instead of eight times.  The price: the order of the points INSIDE one cell (and therefore the
// order of the indices inside a neighbour list) is the arrival order of the atomics and differs from run to run.  The
// neighbour SETS do not depend on it.  `exact_layout = 1` selects the radix build, which is stable and reproducible.
#include "tnsx_kernels.h"
#include "tnsx_device.h"

namespace tnsx {

__device__ __forceinline__ int bin_coord(float p, float o, float inv_h, int n)
{
	// identical to cell_coord() of tnsx_kernels.hip: fp32 sub, mul, truncate, clamp
	const float f = __fmul_rn(__fsub_rn(p, o), inv_h);
	int c = (int)f;
	c = c < 0 ? 0 : c;
	return c > n - 1 ? n - 1 : c;
}

__global__ void __launch_bounds__(256) k_bin_count(const float* __restrict__ xyz, int n, GridParams g, uint32_t* __restrict__ count,
                                                  uint2* __restrict__ keyrank)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int ix = bin_coord(xyz[3 * (size_t)i], g.ox, g.inv_h, g.nx);
	const int iy = bin_coord(xyz[3 * (size_t)i + 1], g.oy, g.inv_h, g.ny);
	const int iz = bin_coord(xyz[3 * (size_t)i + 2], g.oz, g.inv_h, g.nz);
	const uint32_t key = (uint32_t)((iz * g.ny + iy) * g.nx + ix);
	const uint32_t rank = atomicAdd(count + key, 1u);
	keyrank[i] = make_uint2(key, rank);
}
void launch_bin_count(const float* xyz, int n, GridParams g, uint32_t* count, uint2* keyrank, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_bin_count, dim3((n + 255) / 256), dim3(256), 0, s, xyz, n, g, count, keyrank);
}

__global__ void __launch_bounds__(256) k_bin_scatter(const float* __restrict__ xyz, const float* __restrict__ radii, const uint2* __restrict__ keyrank,
                                                    const uint32_t* __restrict__ start, int n, float4* __restrict__ xyzi, float* __restrict__ r2)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const uint2 kr = keyrank[i];
	const uint32_t pos = start[kr.x] + kr.y;
	float4 v;
	v.x = xyz[3 * (size_t)i]; v.y = xyz[3 * (size_t)i + 1]; v.z = xyz[3 * (size_t)i + 2];
	v.w = __uint_as_float((uint32_t)i);
	xyzi[pos] = v;
	if (radii) { const float r = radii[i]; r2[pos] = __fmul_rn(r, r); }   // radii_sq = r*r in fp32, TreeNSearch.cpp:2352
}
void launch_bin_scatter(const float* xyz, const float* radii, const uint2* keyrank, const uint32_t* start, int n, float4* xyzi, float* r2, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_bin_scatter, dim3((n + 255) / 256), dim3(256), 0, s, xyz, radii, keyrank, start, n, xyzi, r2);
}

// table + occupied-cell list from the scanned counts.  One block per 4096 cells; blocks append their occupied cells in
// key order, the order between blocks follows the (nearly monotone) order of their atomics.
static constexpr int CC_THREADS = 256;
static constexpr int CC_ITEMS = 16;
static constexpr int CC_TILE = CC_THREADS * CC_ITEMS;

__global__ void __launch_bounds__(CC_THREADS) k_cells_from_counts(const uint32_t* __restrict__ start, uint32_t n_cells, uint2* __restrict__ table,
                                                                 uint2* __restrict__ occ, uint32_t* __restrict__ n_occ)
{
	__shared__ uint32_t wcnt[CC_ITEMS * (CC_THREADS / WAVE)];
	__shared__ uint32_t block_base;
	const int w = threadIdx.x / WAVE;
	const uint32_t base = blockIdx.x * CC_TILE;
	uint32_t first[CC_ITEMS];
	uint32_t flags = 0;
	#pragma unroll
	for (int i = 0; i < CC_ITEMS; i++) {
		const uint32_t c = base + (uint32_t)i * CC_THREADS + threadIdx.x;
		bool occupied = false;
		first[i] = 0;
		if (c < n_cells) {
			const uint32_t s0 = start[c], s1 = start[c + 1];
			table[c] = make_uint2(s0, s1);
			first[i] = s0;
			occupied = s1 > s0;
		}
		flags |= (occupied ? 1u : 0u) << i;
		const uint64_t m = __builtin_amdgcn_ballot_w64(occupied);
		if (lane_id() == 0) wcnt[i * (CC_THREADS / WAVE) + w] = (uint32_t)__popcll(m);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t s = 0;
		for (int q = 0; q < CC_ITEMS * (CC_THREADS / WAVE); q++) { const uint32_t t = wcnt[q]; wcnt[q] = s; s += t; }
		block_base = s ? atomicAdd(n_occ, s) : 0u;
	}
	__syncthreads();
	const uint32_t bb = block_base;
	#pragma unroll
	for (int i = 0; i < CC_ITEMS; i++) {
		const bool occupied = (flags >> i) & 1u;
		const uint64_t m = __builtin_amdgcn_ballot_w64(occupied);
		if (occupied) {
			const uint32_t c = base + (uint32_t)i * CC_THREADS + threadIdx.x;
			occ[bb + wcnt[i * (CC_THREADS / WAVE) + w] + mbcnt64(m)] = make_uint2(first[i], c);
		}
	}
}
void launch_cells_from_counts(const uint32_t* start, uint32_t n_cells, uint2* table, uint2* occ, uint32_t* n_occ, hipStream_t s)
{
	if (n_cells == 0) return;
	hipLaunchKernelGGL(k_cells_from_counts, dim3((n_cells + CC_TILE - 1) / CC_TILE), dim3(CC_THREADS), 0, s, start, n_cells, table, occ, n_occ);
}

}  // namespace tnsx
