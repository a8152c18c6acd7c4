// gfx950 kernels: the BUILD of the search structure -- binning of one point set by grid cell, then the cell table.
//
// An LSD radix sort on the cell key that moves THE POINT ITSELF (x, y, z, original index [, r*r]) and nothing else: no key
// array and no index array exist.  Every kernel recomputes the key of a point from its position (six fp32 ops and two
// integer multiply-adds) -- cheaper than the traffic of a key array, and MUCH cheaper than its scattered 4-byte stores:
// measured on MI355X (tools/ubench/cellsort_bench.hip, 10 M points, two passes) the scattered 4-byte key stores cost 0.17 ms
// of a 0.45 ms sort, the scattered 16-byte point stores 0.05 ms.
//
//   k_cs_hist<FIRST>     per-tile histogram of one digit               12 B (user xyz) or 16 B read per point
//   k_cs_strip_sums/scan exclusive scan of the tile histograms over the tiles (per digit value) + the totals
//   k_cs_scatter<FIRST>  ranked, stable scatter of the points          12|16 B read, 16 B written per point
//   k_cell_table         table[key] = (first, one past last) sorted position, list of occupied cells
//
// Digits are up to 11 bits wide, so the 20-bit keys of a 10 M-point cloud need TWO passes: every point is moved twice.
// prepare_zsort uses the same kernels with the Morton code of the reference grid as the key (sort_key<true>).
// Stable: wave w of a workgroup owns CS_ITEMS*64 consecutive elements and walks them in rounds of 64, so (wave, round, lane)
// order is index order; the point order inside a cell is therefore the input order -- reproducible from run to run.
#include "tnsx_kernels.h"
#include "tnsx_device.h"

namespace tnsx {

#ifndef TNSX_CS_THREADS
#define TNSX_CS_THREADS 256
#endif
static constexpr int CS_THREADS = TNSX_CS_THREADS;
static constexpr int CS_WAVES = CS_THREADS / WAVE;
static constexpr int CS_ITEMS = 16;
static constexpr int CS_TILE = CS_THREADS * CS_ITEMS;

static int cs_num_tiles(int n) { return (n + CS_TILE - 1) / CS_TILE; }
// Workgroup b runs on XCD b % 8.  Tiles are handed out so that every XCD owns a CONTIGUOUS range of tiles: the output runs of
// neighbouring tiles are adjacent in memory (same digit, next tile), so the cache lines they share are completed inside one
// XCD's L2 instead of leaving two XCDs as partial-line writes (measured: -20 % on the scatter).
static int cs_grid(int ntiles) { return ((ntiles + 7) / 8) * 8; }
__device__ __forceinline__ int cs_tile_of_block(int ntiles) { return (int)(blockIdx.x & 7u) * ((ntiles + 7) / 8) + (int)(blockIdx.x >> 3); }

CellSortPlan cell_sort_plan(int key_bits)
{
	CellSortPlan p{};
	if (key_bits < 1) key_bits = 1;
	p.passes = (key_bits + CS_MAX_BITS - 1) / CS_MAX_BITS;
	int left = key_bits;
	for (int i = 0; i < p.passes; i++) {
		const int b = (left + (p.passes - i) - 1) / (p.passes - i);   // spread the bits evenly
		p.bits[i] = b < 8 ? 8 : b;
		left -= b;
	}
	return p;
}
size_t cell_sort_temp_bytes(int n)
{
	const size_t hist_elems = ((size_t)1 << CS_MAX_BITS) * (size_t)cs_num_tiles(n > 0 ? n : 1);
	const size_t nstrips = ((size_t)cs_num_tiles(n > 0 ? n : 1) + 31) / 32;
	return ((hist_elems * sizeof(uint32_t) + 255) / 256) * 256 + ((size_t)1 << CS_MAX_BITS) * sizeof(uint32_t) * (1 + nstrips) + 256;
}

struct F3 { float x, y, z; };   // 12-byte AoS point of the user array (4-byte aligned)

// (bin_coord: tnsx_device.h -- the query kernels recompute cell coordinates with the very same operations)
// row-major cell key, x fastest: the three x-neighbours of a row are contiguous in sorted order.
// A point whose x is NaN is NO POINT (the padding rows of a fixed-capacity ghost message, treensearch_amd/multi.py): it gets the
// key one past the last cell, is sorted behind everything and enters no cell; every comparison with it is false anyway.
__device__ __forceinline__ uint32_t cell_key(float x, float y, float z, const GridParams& g)
{
	const int ix = bin_coord(x, g.ox, g.inv_h, g.nx);
	const int iy = bin_coord(y, g.oy, g.inv_h, g.ny);
	const int iz = bin_coord(z, g.oz, g.inv_h, g.nz);
	const uint32_t key = ((uint32_t)iz * (uint32_t)g.ny + (uint32_t)iy) * (uint32_t)g.nx + (uint32_t)ix;   // (unsigned: a sparse grid may have up to 2^32 - 16 cells)
	return x != x ? (uint32_t)g.nx * (uint32_t)g.ny * (uint32_t)g.nz : key;
}
// the two sort keys: MORTON = false the cell key of the search grid, MORTON = true the Morton code of the cell on the reference's
// grid (prepare_zsort; up to 63 bits)
template <bool MORTON>
__device__ __forceinline__ uint64_t sort_key(float x, float y, float z, const GridParams& g)
{
	if (!MORTON) return cell_key(x, y, z, g);
	const uint64_t ux = (uint64_t)bin_coord(x, g.ox, g.inv_h, g.nx), uy = (uint64_t)bin_coord(y, g.oy, g.inv_h, g.ny), uz = (uint64_t)bin_coord(z, g.oz, g.inv_h, g.nz);
	return x != x ? ~0ull : (spread3(ux) | (spread3(uy) << 1) | (spread3(uz) << 2));   // NaN: behind everything
}

// ---- run-time validation of what tnsx_run speculates on (both ride on the first histogram pass, which reads every point anyway):
//   guard     the search grid of the previous run is reused without looking at the bounds first; a point outside the box the
//             grid was laid out for (or a radius above the one its cell edge covers) raises *flag and the host repeats the run
//   checksum  order-sensitive 64-bit sum over the raw bits of all points (and radii): a set whose checksum, pointer and size did
//             not change keeps its sorted arrays and cell table (the static boundary of an SPH scene)
__device__ __forceinline__ unsigned long long point_hash(uint32_t i, float x, float y, float z, float r)
{
	// two 32-bit mixes of the four words (odd multipliers, rotations), joined to 64 bits and weighted with the odd number 2 i + 1
	// (a permutation of the points, or a change of one of them, changes the sum): ~14 integer instructions per point
	const uint32_t ux = __float_as_uint(x), uy = __float_as_uint(y), uz = __float_as_uint(z), ur = __float_as_uint(r);
	const uint32_t m1 = (ux * 0x9E3779B1u) ^ __funnelshift_l(uy, uy, 13) ^ (uz * 0x85EBCA77u) ^ __funnelshift_l(ur, ur, 7);
	const uint32_t m2 = (uy * 0xC2B2AE3Du) ^ __funnelshift_l(uz, uz, 17) ^ (ur * 0x27D4EB2Fu) ^ __funnelshift_l(ux, ux, 5) ^ i;
	const unsigned long long h = ((unsigned long long)m1 << 32) | m2;
	return h * (unsigned long long)(2u * i + 1u) + (unsigned long long)m1 * m2;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
	return v;
}
// ---- per-tile histogram of one digit -----------------------------------------------------------------------------
template <int BITS, bool FIRST, bool MORTON>
__global__ void __launch_bounds__(CS_THREADS) k_cs_hist(const float* __restrict__ xyz, const float4* __restrict__ xyzi, int n, GridParams g, int shift,
                                                        uint32_t* __restrict__ hist, int ntiles, const float* __restrict__ radii, BuildGuard gd)
{
	constexpr int RADIX = 1 << BITS;
	__shared__ uint32_t h[RADIX];
	for (int b = threadIdx.x; b < RADIX; b += CS_THREADS) h[b] = 0;
	__syncthreads();
	const size_t base = (size_t)blockIdx.x * CS_TILE;
	bool bad = false;
	float mn[3] = { gd.hi[0], gd.hi[1], gd.hi[2] }, mx[3] = { gd.lo[0], gd.lo[1], gd.lo[2] };   // (guard: this thread's bounds, compared once at the end)
	unsigned long long chk = 0;
	uint32_t n_outside = 0;
	#pragma unroll 8
	for (int i = 0; i < CS_ITEMS; i++) {
		const size_t e = base + (size_t)i * CS_THREADS + threadIdx.x;
		if (e < (size_t)n) {
			uint64_t key;
			if (FIRST) {
				const F3 q = reinterpret_cast<const F3*>(xyz)[e];
				key = sort_key<MORTON>(q.x, q.y, q.z, g);
				if (!MORTON) {
					if (gd.flag) {
						// v_min / v_max drop a NaN operand: a NaN x ("no point") never counts, NaN in y or z is caught below
						// (a NaN x is NO POINT: whatever its y and z hold -- the rows of a ghost message past the real count -- counts for nothing)
						const bool pt = q.x == q.x;
						const float qy = pt ? q.y : q.x, qz = pt ? q.z : q.x;
						mn[0] = fminf(mn[0], q.x); mx[0] = fmaxf(mx[0], q.x);
						mn[1] = fminf(mn[1], qy); mx[1] = fmaxf(mx[1], qy);
						mn[2] = fminf(mn[2], qz); mx[2] = fmaxf(mx[2], qz);
						bad |= pt & ((q.y != q.y) | (q.z != q.z));
						if (gd.outside) n_outside += pt & ((q.x < gd.soft_lo[0]) | (q.x > gd.soft_hi[0]) | (q.y < gd.soft_lo[1]) | (q.y > gd.soft_hi[1]) |
						                                   (q.z < gd.soft_lo[2]) | (q.z > gd.soft_hi[2]));
					}
					if (gd.checksum) chk += point_hash((uint32_t)e, q.x, q.y, q.z, radii ? radii[e] : 0.0f);
				}
			}
			else { const float4 q = xyzi[e]; key = sort_key<MORTON>(q.x, q.y, q.z, g); }
			atomicAdd(&h[(uint32_t)(key >> shift) & (RADIX - 1)], 1u);
		}
	}
	if (FIRST && !MORTON) {
		if (gd.flag) bad |= mn[0] < gd.lo[0] || mn[1] < gd.lo[1] || mn[2] < gd.lo[2] || mx[0] > gd.hi[0] || mx[1] > gd.hi[1] || mx[2] > gd.hi[2];
		if (gd.flag && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane_id() == 0) atomicOr(gd.flag, 1u);
		if (gd.flag && gd.outside && __builtin_amdgcn_ballot_w64(n_outside != 0u) != 0ull) {   // (rare by construction: a handful of outliers)
			const unsigned long long w = wave_sum_u64((unsigned long long)n_outside);
			if (lane_id() == 0) atomicAdd(gd.outside, w);
		}
		// (partial sums spread over CHK_SLOTS cache lines: the L2 serialises atomics on one line, ~88 per microsecond, and ten thousand
		//  waves adding to ONE word cost 0.12 ms at 10 M points)
		if (gd.checksum) { chk = wave_sum_u64(chk); if (lane_id() == 0 && chk) atomicAdd(gd.checksum + (blockIdx.x % CHK_SLOTS) * CHK_STRIDE, chk); }
	}
	__syncthreads();
	for (int b = threadIdx.x; b < RADIX; b += CS_THREADS) hist[(size_t)blockIdx.x * RADIX + b] = h[b];   // row = tile: coalesced
}
// the checksum alone (a set that is taken to be static skips its build; this verifies the assumption)
__global__ void __launch_bounds__(256) k_set_checksum(const float* __restrict__ xyz, const float* __restrict__ radii, int n, unsigned long long* __restrict__ out)
{
	unsigned long long chk = 0;
	for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)n; e += (size_t)gridDim.x * 256) {
		const F3 q = reinterpret_cast<const F3*>(xyz)[e];
		chk += point_hash((uint32_t)e, q.x, q.y, q.z, radii ? radii[e] : 0.0f);
	}
	chk = wave_sum_u64(chk);
	if (lane_id() == 0 && chk) atomicAdd(out + (blockIdx.x % CHK_SLOTS) * CHK_STRIDE, chk);
}
void launch_set_checksum(const float* xyz, const float* radii, int n, unsigned long long* out, hipStream_t s)
{
	if (n <= 0) return;
	int blocks = (n + 256 * 16 - 1) / (256 * 16);
	blocks = blocks > 4096 ? 4096 : blocks;
	hipLaunchKernelGGL(k_set_checksum, dim3(blocks), dim3(256), 0, s, xyz, radii, n, out);
}

// ---- scan of the tile histograms down the columns: hist[tile][value] -> exclusive prefix over the tiles, totals[value].
//      The table is stored row = tile (the histogram and scatter kernels then read and write whole rows; written column-wise
//      the 4-byte stores of the histogram kernel cost +25 us per pass).  Two small kernels over strips of SB_STRIP tiles:
//      column sums of every strip, then every strip adds the sums of the strips before it and scans its own rows in place.
static constexpr int SB_THREADS = 256;
static constexpr int SB_STRIP = 32;
template <int BITS>
__global__ void __launch_bounds__(SB_THREADS) k_cs_strip_sums(const uint32_t* __restrict__ hist, int ntiles, uint32_t* __restrict__ strip_sums)
{
	constexpr int RADIX = 1 << BITS;
	const int t0 = blockIdx.x * SB_STRIP, t1 = min(t0 + SB_STRIP, ntiles);
	const int b = blockIdx.y * SB_THREADS + threadIdx.x;   // grid.y = RADIX / SB_THREADS
	uint32_t acc = 0;
	#pragma unroll 16
	for (int t = t0; t < t1; t++) acc += hist[(size_t)t * RADIX + b];
	strip_sums[(size_t)blockIdx.x * RADIX + b] = acc;
}
template <int BITS>
__global__ void __launch_bounds__(SB_THREADS) k_cs_strip_scan(uint32_t* __restrict__ hist, int ntiles, const uint32_t* __restrict__ strip_sums,
                                                              uint32_t* __restrict__ totals)
{
	constexpr int RADIX = 1 << BITS;
	const int t0 = blockIdx.x * SB_STRIP, t1 = min(t0 + SB_STRIP, ntiles);
	const int b = blockIdx.y * SB_THREADS + threadIdx.x;   // grid.y = RADIX / SB_THREADS
	uint32_t base = 0;
	#pragma unroll 16
	for (int st = 0; st < (int)blockIdx.x; st++) base += strip_sums[(size_t)st * RADIX + b];
	for (int t = t0; t < t1; t += 16) {
		uint32_t v[16];
		#pragma unroll
		for (int k = 0; k < 16; k++) v[k] = (t + k < t1) ? hist[(size_t)(t + k) * RADIX + b] : 0u;
		#pragma unroll
		for (int k = 0; k < 16; k++) { if (t + k < t1) hist[(size_t)(t + k) * RADIX + b] = base; base += v[k]; }
	}
	if (blockIdx.x == gridDim.x - 1) totals[b] = base;
}

// ---- ranked scatter of the points --------------------------------------------------------------------------------
// FIRST: the point comes from the user's array (xyz AoS) and gets its original index attached; otherwise it comes from the
// previous pass.  VARIABLE (last pass only): r2 = r*r in fp32 (TreeNSearch.cpp:2352) is written next to the point, the radius
// picked up by original index -- carrying it through every pass costs a scattered 4-byte store per point and pass, the expensive
// kind (measured at 20 M points, two passes: sort 0.77 ms carried, 0.68 ms gathered).
template <int BITS, bool FIRST, bool VARIABLE, bool MORTON>
// (three waves per SIMD allowed: at four, the sixteen register-resident points of a thread + their ranks leave the compiler 16 - 52 bytes of scratch per lane in some
//  instantiations; the kernel waits for its scattered stores, not for issue slots -- round 5: scratch 0 for every instantiation)
__global__ void __launch_bounds__(CS_THREADS) __attribute__((amdgpu_waves_per_eu(3, 4)))
k_cs_scatter(const float* __restrict__ xyz, const float* __restrict__ radii, const float4* __restrict__ xyzi_in, const float* __restrict__ r2_in,
             float4* __restrict__ xyzi_out, float* __restrict__ r2_out, int n, GridParams g, int shift, const uint32_t* __restrict__ hist_scanned,
             const uint32_t* __restrict__ totals, int ntiles, const int* __restrict__ ids, uint32_t* __restrict__ orig_out, BuildGuard gd)
{
	constexpr int RADIX = 1 << BITS;
	constexpr int PER = RADIX / CS_THREADS;   // digit values per thread in the prefix steps
	static_assert(RADIX % CS_THREADS == 0 && PER >= 1, "digit values must divide among the threads");
	__shared__ uint32_t wcount[CS_WAVES][RADIX];
	__shared__ uint32_t gbase[RADIX];
	__shared__ uint32_t wsum[CS_WAVES];
	const int w = (int)readfirstlane_u32(threadIdx.x / WAVE), lane = lane_id();
	const int tile = cs_tile_of_block(ntiles);
	if (tile >= ntiles) return;

	float px[CS_ITEMS], py[CS_ITEMS], pz[CS_ITEMS], pw[CS_ITEMS];   // (scalar arrays: a float4 array ends up in scratch)
	bool bad_r = false;
	// this wave's CS_ITEMS*64 consecutive elements: wave-uniform 64-bit base + 32-bit lane offsets (scalar-base addressing)
	const size_t wbase = (size_t)tile * CS_TILE + (size_t)w * (CS_ITEMS * WAVE);
	const uint32_t rem = wbase < (size_t)n ? (uint32_t)((size_t)n - wbase < (size_t)(CS_ITEMS * WAVE) ? (size_t)n - wbase : (size_t)(CS_ITEMS * WAVE)) : 0u;
	const size_t lbase = rem ? wbase : 0;   // (waves past the end load element 0 and drop it)
	const uint32_t lclamp = rem ? rem - 1u : 0u;
	// all loads of the tile up front, branch-free (clamped index), so that they are in flight during the set-up below
	#pragma unroll
	for (int i = 0; i < CS_ITEMS; i++) {
		const uint32_t li = (uint32_t)(i * WAVE + lane);
		const uint32_t lc = li < lclamp ? li : lclamp;
		if (FIRST) {
			const F3 q = (reinterpret_cast<const F3*>(xyz) + lbase)[lc];
			px[i] = q.x; py[i] = q.y; pz[i] = q.z; pw[i] = __uint_as_float((uint32_t)wbase + li);
		}
		else {
			const float4 q = (xyzi_in + lbase)[lc];
			px[i] = q.x; py[i] = q.y; pz[i] = q.z; pw[i] = q.w;
		}
	}

	// global base of every digit value for this tile = (exclusive scan of the totals) + (scanned tile count).  Thread t owns the
	// PER consecutive values [t*PER, t*PER + PER).
	{
		uint32_t tot[PER], sum = 0;
		#pragma unroll
		for (int k = 0; k < PER; k++) { tot[k] = totals[threadIdx.x * PER + k]; sum += tot[k]; }
		uint32_t inc = sum;
		#pragma unroll
		for (int o = 1; o < WAVE; o <<= 1) { const uint32_t u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
		if (lane == WAVE - 1) wsum[w] = inc;
		#pragma unroll
		for (int k = 0; k < PER; k++) {
			#pragma unroll
			for (int ww = 0; ww < CS_WAVES; ww++) wcount[ww][threadIdx.x * PER + k] = 0;
		}
		__syncthreads();
		uint32_t ex = inc - sum;
		#pragma unroll
		for (int ww = 0; ww < CS_WAVES; ww++) if (ww < w) ex += wsum[ww];
		#pragma unroll
		for (int k = 0; k < PER; k++) {
			const int b = threadIdx.x * PER + k;
			gbase[b] = ex + hist_scanned[(size_t)tile * RADIX + b];
			ex += tot[k];
		}
	}

	uint32_t dig_rank[CS_ITEMS];   // digit | rank in the wave's sub-tile << 16 (both < 2^16)
	#pragma unroll
	for (int i = 0; i < CS_ITEMS; i++) {
		const bool valid = (uint32_t)(i * WAVE + lane) < rem;
		const uint32_t d = (uint32_t)(sort_key<MORTON>(px[i], py[i], pz[i], g) >> shift) & (RADIX - 1);
		uint64_t peers = __ballot(valid);
		#pragma unroll
		for (int b = 0; b < BITS; b++) {
			const bool bit = (d >> b) & 1u;
			const uint64_t m = __ballot(valid && bit);
			peers &= bit ? m : ~m;
		}
		const uint32_t r = mbcnt64(peers);                 // peers in lower lanes
		const uint32_t cnt = (uint32_t)__popcll(peers);
		uint32_t prev = 0;
		if (valid) prev = wcount[w][d];
		wave_lds_fence();
		if (valid && r == 0) wcount[w][d] = prev + cnt;
		wave_lds_fence();
		dig_rank[i] = d | ((prev + r) << 16);
	}
	__syncthreads();
	#pragma unroll
	for (int k = 0; k < PER; k++) {
		const int b = threadIdx.x * PER + k;
		uint32_t s = gbase[b];
		#pragma unroll
		for (int ww = 0; ww < CS_WAVES; ww++) { const uint32_t t = wcount[ww][b]; wcount[ww][b] = s; s += t; }
	}
	__syncthreads();
	// The radii are picked up HERE, eight at a time, by original index (the radii array of a 50 M-point set is 200 MB: it stays in the 256 MB Infinity Cache; a single
	// pass: index = position, a coalesced load) -- not with the points at the top of the kernel: sixteen more registers alive through the ranking were what made
	// every VARIABLE instantiation spill 150 - 200 bytes per lane at four waves per SIMD (round 5: scratch 0).
	constexpr int RB = 8;
	#pragma unroll
	for (int i0 = 0; i0 < CS_ITEMS; i0 += RB) {
		float rr[RB];
		if (VARIABLE) {
			#pragma unroll
			for (int u = 0; u < RB; u++) {
				const int i = i0 + u;
				const uint32_t li = (uint32_t)(i * WAVE + lane);
				const float r = FIRST ? (radii + lbase)[li < lclamp ? li : lclamp] : radii[__float_as_uint(pw[i])];
				rr[u] = __fmul_rn(r, r);
				bad_r |= (r > gd.r_max) & (px[i] == px[i]);   // (the radius of a NaN x -- no point -- counts for nothing)
			}
		}
		#pragma unroll
		for (int u = 0; u < RB; u++) {
			const int i = i0 + u;
			if ((uint32_t)(i * WAVE + lane) < rem) {
				const uint32_t pos = wcount[w][dig_rank[i] & 0xffffu] + (dig_rank[i] >> 16);
				float wv = pw[i];
				if (ids) {
					// last pass of a set with user ids (tnsx_set_point_ids): the point carries its ID from here on -- that is what the
					// query emits -- and its original index goes to a side array (the query needs it for the queries only)
					const uint32_t o = __float_as_uint(wv);
					wv = __int_as_float(ids[o]);
					orig_out[pos] = o;
				}
				xyzi_out[pos] = make_float4(px[i], py[i], pz[i], wv);
				if (VARIABLE) r2_out[pos] = rr[u];
			}
		}
	}
	// (speculated grid: a radius above the one the cell edge was chosen for -> the host repeats the run with fresh bounds)
	if (VARIABLE && gd.flag && __builtin_amdgcn_ballot_w64(bad_r) != 0ull && lane == 0) atomicOr(gd.flag, 1u);
}

template <int BITS, bool MORTON>
static void cs_hist(bool first, const float* xyz, const float4* xyzi, int n, const GridParams& g, int shift, uint32_t* hist, int ntiles, const float* radii,
                    const BuildGuard& gd, hipStream_t s)
{
	if (first) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_hist<BITS, true, MORTON>), dim3(ntiles), dim3(CS_THREADS), 0, s, xyz, xyzi, n, g, shift, hist, ntiles, radii, gd);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_hist<BITS, false, MORTON>), dim3(ntiles), dim3(CS_THREADS), 0, s, xyz, xyzi, n, g, shift, hist, ntiles, radii, gd);
}
template <int BITS>
static void cs_scan(uint32_t* hist, int ntiles, uint32_t* strip_sums, uint32_t* totals, hipStream_t s)
{
	const int nstrips = (ntiles + SB_STRIP - 1) / SB_STRIP;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_strip_sums<BITS>), dim3(nstrips, (1u << BITS) / SB_THREADS), dim3(SB_THREADS), 0, s, hist, ntiles, strip_sums);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_strip_scan<BITS>), dim3(nstrips, (1u << BITS) / SB_THREADS), dim3(SB_THREADS), 0, s, hist, ntiles, strip_sums, totals);
}
template <int BITS, bool MORTON>
static void cs_scatter(bool first, bool variable, const float* xyz, const float* radii, const float4* xyzi_in, const float* r2_in, float4* xyzi_out,
                       float* r2_out, int n, const GridParams& g, int shift, const uint32_t* hs, const uint32_t* totals, int ntiles, const int* ids,
                       uint32_t* orig_out, const BuildGuard& gd, hipStream_t s)
{
#define TNSX_CS_GO(F, V)                                                                                                                       \
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_scatter<BITS, F, V, MORTON>), dim3(cs_grid(ntiles)), dim3(CS_THREADS), 0, s, xyz, radii, xyzi_in, r2_in, xyzi_out, \
	                   r2_out, n, g, shift, hs, totals, ntiles, ids, orig_out, gd)
	if (MORTON) { if (first) TNSX_CS_GO(true, false); else TNSX_CS_GO(false, false); }   // the z-order carries no radii
	else if (first) { if (variable) TNSX_CS_GO(true, true); else TNSX_CS_GO(true, false); }
	else            { if (variable) TNSX_CS_GO(false, true); else TNSX_CS_GO(false, false); }
#undef TNSX_CS_GO
}

#define TNSX_CS_DISPATCH(bits, call)                  \
	switch (bits) {                                   \
	case 8:  { constexpr int B = 8;  call; } break;   \
	case 9:  { constexpr int B = 9;  call; } break;   \
	case 10: { constexpr int B = 10; call; } break;   \
	default: { constexpr int B = 11; call; } break;   \
	}

template <bool MORTON>
static int point_sort(const float* xyz, const float* radii, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, const int* ids,
                      uint32_t* orig_out, const BuildGuard& gd, hipStream_t s)
{
	const CellSortPlan plan = cell_sort_plan(key_bits);
	if (n <= 0) return plan.passes & 1;
	const int ntiles = cs_num_tiles(n);
	const size_t hist_cap = ((size_t)1 << CS_MAX_BITS) * (size_t)ntiles;
	uint32_t* hist = (uint32_t*)temp;
	uint32_t* totals = (uint32_t*)((char*)temp + ((hist_cap * sizeof(uint32_t) + 255) / 256) * 256);
	uint32_t* strip_sums = totals + ((size_t)1 << CS_MAX_BITS);
	const bool variable = radii != nullptr;
	int cur = 0, shift = 0;
	for (int p = 0; p < plan.passes; p++) {
		const int bits = plan.bits[p];
		TNSX_CS_DISPATCH(bits, (cs_hist<B, MORTON>(p == 0, xyz, b.xyzi[cur], n, g, shift, hist, ntiles, radii, gd, s)));
		TNSX_CS_DISPATCH(bits, cs_scan<B>(hist, ntiles, strip_sums, totals, s));
		const bool last = p == plan.passes - 1;
		TNSX_CS_DISPATCH(bits, (cs_scatter<B, MORTON>(p == 0, variable && last, xyz, radii, b.xyzi[cur], b.r2[cur], b.xyzi[cur ^ 1], b.r2[cur ^ 1], n, g, shift, hist,
		                                              totals, ntiles, last ? ids : nullptr, orig_out, gd, s)));
		cur ^= 1;
		shift += bits;
	}
	return cur;
}
int launch_cell_sort(const float* xyz, const float* radii, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, const int* ids,
                     uint32_t* orig_sorted, const BuildGuard& gd, hipStream_t s)
{
	return point_sort<false>(xyz, radii, n, g, key_bits, b, temp, ids, orig_sorted, gd, s);
}

__global__ void __launch_bounds__(256) k_extract_order(const float4* __restrict__ xyzi, int n, int* __restrict__ order)
{
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p < n) order[p] = (int)__float_as_uint(xyzi[p].w);
}
// z-order in TWO passes (round 4; keys of 16 .. 24 bits, i.e. the cell-level order after a run()): the stable ranked pass on the high digit, then
// one workgroup per bucket that counts its points per low digit in LDS, scans, and writes the ORDER directly -- the point's original index at its
// final rank, 4 bytes -- instead of moving the point a third time and extracting the index column afterwards.  The order inside a cell of the
// reference grid is whatever the LDS atomics give (the reference keeps the order of its own cell lists there, which is just as unspecified).
static constexpr int BP_THREADS = 1024;
static constexpr int BP_KEEP = 8, BP_UNROLL = 4;
// the low 30 bits of a point's Morton code in 32-bit arithmetic (the digit a bucket is sorted by has at most 13): a third of the instructions of sort_key<true>
__device__ __forceinline__ uint32_t spread3_10(uint32_t v)
{
	v &= 0x3ffu;
	v = (v | (v << 16)) & 0x030000ffu;
	v = (v | (v << 8)) & 0x0300f00fu;
	v = (v | (v << 4)) & 0x030c30c3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}
__device__ __forceinline__ uint32_t morton_low(float x, float y, float z, const GridParams& g)
{
	const uint32_t ux = (uint32_t)bin_coord(x, g.ox, g.inv_h, g.nx), uy = (uint32_t)bin_coord(y, g.oy, g.inv_h, g.ny), uz = (uint32_t)bin_coord(z, g.oz, g.inv_h, g.nz);
	return x != x ? 0xffffffffu : (spread3_10(ux) | (spread3_10(uy) << 1) | (spread3_10(uz) << 2));   // (NaN: all ones, like the full key)
}
__global__ void __launch_bounds__(BP_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_morton_place(const float4* __restrict__ in, int* __restrict__ order_out, GridParams g, int lo_bits,
                                                             const uint32_t* __restrict__ totals)
{
	extern __shared__ uint32_t bp_h[];
	__shared__ uint32_t red[BP_THREADS / WAVE];
	const int RADIX = 1 << lo_bits;
	const int b = (int)blockIdx.x, lane = lane_id(), w = (int)threadIdx.x / WAVE;
	uint32_t part = 0;
	for (int k = (int)threadIdx.x; k < b; k += BP_THREADS) part += totals[k];
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
	if (lane == 0) red[w] = part;
	const uint32_t count = totals[b];
	for (int k = (int)threadIdx.x; k < RADIX; k += BP_THREADS) bp_h[k] = 0u;
	__syncthreads();
	uint32_t start = 0;
	#pragma unroll
	for (int k = 0; k < BP_THREADS / WAVE; k++) start += red[k];
	if (count == 0u) return;
	in += start;
	const uint32_t mask = (uint32_t)(RADIX - 1);
	// The first BP_KEEP * BP_THREADS points of the bucket (all of them, usually) are read ONCE: their digit and their index stay in registers between the two
	// sweeps, and the loads are issued BP_UNROLL at a time (a workgroup with a few points per thread is a chain of latencies, not of bytes).  Measured on the 10 M
	// points of a dam break kept in z-order, interleaved in one process (tools/zsort_ab.py): prepare_zsort 0.385 -> 0.317 ms with this and the 32-bit digit.  Adding
	// the length of a run of equal digits in consecutive lanes once (the lanes of a wave hold a handful of distinct digits) instead of one LDS atomic per lane was
	// built and measured SLOWER (0.358 ms): the two cross-lane moves it needs cost more than the conflicts they avoid.
	uint32_t kd[BP_KEEP], ko[BP_KEEP];
	#pragma unroll
	for (int u0 = 0; u0 < BP_KEEP; u0 += BP_UNROLL) {
		float4 q[BP_UNROLL];
		#pragma unroll
		for (int u = 0; u < BP_UNROLL; u++) { const uint32_t i = (uint32_t)(u0 + u) * BP_THREADS + threadIdx.x; q[u] = in[i < count ? i : count - 1u]; }
		#pragma unroll
		for (int u = 0; u < BP_UNROLL; u++) {
			kd[u0 + u] = morton_low(q[u].x, q[u].y, q[u].z, g) & mask;
			ko[u0 + u] = __float_as_uint(q[u].w);
			if ((uint32_t)(u0 + u) * BP_THREADS + threadIdx.x < count) atomicAdd(&bp_h[kd[u0 + u]], 1u);
		}
	}
	for (uint32_t i0 = BP_KEEP * BP_THREADS; i0 < count; i0 += BP_THREADS * BP_UNROLL) {   // (a bucket larger than that: read twice)
		float4 q[BP_UNROLL];
		#pragma unroll
		for (int u = 0; u < BP_UNROLL; u++) { const uint32_t i = i0 + (uint32_t)u * BP_THREADS + threadIdx.x; q[u] = in[i < count ? i : count - 1u]; }
		#pragma unroll
		for (int u = 0; u < BP_UNROLL; u++)
			if (i0 + (uint32_t)u * BP_THREADS + threadIdx.x < count) atomicAdd(&bp_h[morton_low(q[u].x, q[u].y, q[u].z, g) & mask], 1u);
	}
	__syncthreads();
	// exclusive scan of the counts: thread t owns PER consecutive digits
	const int PER = RADIX >= BP_THREADS ? RADIX / BP_THREADS : 1;
	const int mine = (int)threadIdx.x * PER < RADIX ? PER : 0;
	uint32_t sum = 0;
	for (int k = 0; k < mine; k++) sum += bp_h[threadIdx.x * PER + k];
	uint32_t inc = sum;
	#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) { const uint32_t u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
	__syncthreads();
	if (lane == WAVE - 1) red[w] = inc;
	__syncthreads();
	uint32_t ex = inc - sum;
	#pragma unroll
	for (int k = 0; k < BP_THREADS / WAVE; k++) if (k < w) ex += red[k];
	for (int k = 0; k < mine; k++) { const uint32_t c = bp_h[threadIdx.x * PER + k]; bp_h[threadIdx.x * PER + k] = ex; ex += c; }
	__syncthreads();
	#pragma unroll
	for (int u = 0; u < BP_KEEP; u++) {
		const bool valid = (uint32_t)u * BP_THREADS + threadIdx.x < count;
		const uint32_t pos = valid ? atomicAdd(&bp_h[kd[u]], 1u) : 0u;
		if (valid) order_out[start + pos] = (int)ko[u];
	}
	for (uint32_t i0 = BP_KEEP * BP_THREADS; i0 < count; i0 += BP_THREADS * BP_UNROLL) {
		float4 q[BP_UNROLL];
		#pragma unroll
		for (int u = 0; u < BP_UNROLL; u++) { const uint32_t i = i0 + (uint32_t)u * BP_THREADS + threadIdx.x; q[u] = in[i < count ? i : count - 1u]; }
		#pragma unroll
		for (int u = 0; u < BP_UNROLL; u++) {
			const bool valid = i0 + (uint32_t)u * BP_THREADS + threadIdx.x < count;
			const uint32_t pos = valid ? atomicAdd(&bp_h[morton_low(q[u].x, q[u].y, q[u].z, g) & mask], 1u) : 0u;
			if (valid) order_out[start + pos] = (int)__float_as_uint(q[u].w);
		}
	}
}
// =====================================================================================================
// Z-order of WIDE keys (25 .. 30 bits: the cell-level order of a 50 M-point set, 512 cells per axis) -- round 5: {key, index} PAIRS through
// SINGLE-PASS digit sorts ("onesweep": one upfront histogram read for all digits, decoupled look-back instead of a histogram pass + scan
// kernels per digit).  Against the three point-moving LSD passes this path replaces (3 x (k_cs_hist + two scan kernels + k_cs_scatter) + k_extract_order,
// 2.0 ms at 50 M points):
//   * the z-order needs the ORDER, not the points: an element is 8 bytes {30-bit Morton key, original index}, not 16 -- and the key is computed once;
//   * k_zs_keys reads the points ONCE, writes the pairs and counts every digit of every pass on the way (per-workgroup partial histograms in LDS,
//     a small reduction kernel turns them into the global digit bases); no pass reads its input twice;
//   * k_zs_pass sorts a tile of 8192 pairs by one digit: stable rank by ballot matching per wave (as k_cs_scatter), tile totals per digit published to a
//     status word {flag, count}, the totals of the tiles before it collected by LOOK-BACK (tiles take their number from a ticket, so every tile a
//     workgroup waits for is already running), the tile sorted in LDS and written out run by run: a digit's run of a tile is 16 pairs = 128 contiguous
//     bytes on average -- full-line writes instead of the 50 M partial-line requests per pass that bound the scattered stores of the LSD passes;
//   * the last pass writes the index column only: the order.
// =====================================================================================================
static constexpr int ZS_THREADS = 512, ZS_WAVES = ZS_THREADS / WAVE, ZS_ITEMS = 16, ZS_TILE = ZS_THREADS * ZS_ITEMS;
static constexpr int ZS_MAX_PASSES = 4, ZS_MAX_BITS = 10;
static constexpr int ZS_KEY_THREADS = 256, ZS_KEY_ITEMS = 16, ZS_KEY_TILE = ZS_KEY_THREADS * ZS_KEY_ITEMS, ZS_KEY_GRID = 1024;
static constexpr uint32_t ZS_FLAG_AGG = 1u << 30, ZS_FLAG_INC = 2u << 30, ZS_VALUE = (1u << 30) - 1u;
struct ZsPlan { int passes; int bits[ZS_MAX_PASSES]; int shift[ZS_MAX_PASSES]; };
static ZsPlan zs_plan(int key_bits)
{
	ZsPlan p{};
	p.passes = (key_bits + ZS_MAX_BITS - 1) / ZS_MAX_BITS;
	int left = key_bits, sh = 0;
	for (int i = 0; i < p.passes; i++) {
		int b = (left + (p.passes - i) - 1) / (p.passes - i);
		b = b < 8 ? 8 : b;
		p.bits[i] = b; p.shift[i] = sh; sh += b; left -= b;
	}
	return p;
}
static int zs_key_grid(int n) { const int t = (n + ZS_KEY_TILE - 1) / ZS_KEY_TILE; return t < ZS_KEY_GRID ? (t < 1 ? 1 : t) : ZS_KEY_GRID; }
static int zs_tiles(int n) { return (n + ZS_TILE - 1) / ZS_TILE; }
// temp: [partial histograms: grid x passes x 2^ZS_MAX_BITS][digit bases: passes x 2^ZS_MAX_BITS][tickets: ZS_MAX_PASSES][status: passes x tiles x 2^bits]
#ifndef TNSX_ZS_PAIRS_FROM
#define TNSX_ZS_PAIRS_FROM 25   // keys of this many bits and more take the pair sort; 16 .. 24 bits: one ranked pass + k_morton_place (measured against each other: profiles/r5_zsort.txt)
#endif
// the pair sort's look-back status word is {2-bit flag, 30-bit count} and its inclusive per-digit prefixes reach n: sets of 2^30 points and more keep the LSD passes
static bool zs_uses_pairs(int n, int key_bits) { return n >= (1 << 16) && n < (1 << 30) && key_bits >= TNSX_ZS_PAIRS_FROM && key_bits <= 30; }
size_t zsort_temp_bytes(int n, int key_bits)
{
	if (!zs_uses_pairs(n, key_bits)) return 0;   // (the other paths live in cell_sort_temp_bytes)
	const size_t R = (size_t)1 << ZS_MAX_BITS;
	// the status words only of the passes this key takes (round-5 advice: four passes' worth was 400 MB at 200 M points, for every key width)
	return ((size_t)ZS_KEY_GRID * ZS_MAX_PASSES * R + (size_t)ZS_MAX_PASSES * R + 64 + (size_t)zs_plan(key_bits).passes * (size_t)zs_tiles(n) * R) * sizeof(uint32_t) + 256;
}
struct ZsPlanDev { int passes; int bits[ZS_MAX_PASSES]; int shift[ZS_MAX_PASSES]; };
__global__ void __launch_bounds__(ZS_KEY_THREADS) k_zs_keys(const float* __restrict__ xyz, int n, GridParams g, ZsPlanDev plan, uint2* __restrict__ pairs, uint32_t* __restrict__ part)
{
	__shared__ uint32_t h[ZS_MAX_PASSES << ZS_MAX_BITS];
	constexpr int R = 1 << ZS_MAX_BITS;
	for (int k = threadIdx.x; k < plan.passes * R; k += ZS_KEY_THREADS) h[k] = 0u;
	__syncthreads();
	const int ntiles = (n + ZS_KEY_TILE - 1) / ZS_KEY_TILE;
	for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		const size_t base = (size_t)tile * ZS_KEY_TILE;
		const uint32_t rem = (uint32_t)((size_t)n - base < (size_t)ZS_KEY_TILE ? (size_t)n - base : (size_t)ZS_KEY_TILE);
		F3 q[ZS_KEY_ITEMS];
		#pragma unroll
		for (int i = 0; i < ZS_KEY_ITEMS; i++) {   // all loads up front, clamped
			const uint32_t li = (uint32_t)i * ZS_KEY_THREADS + threadIdx.x;
			q[i] = (reinterpret_cast<const F3*>(xyz) + base)[li < rem ? li : rem - 1u];
		}
		#pragma unroll
		for (int i = 0; i < ZS_KEY_ITEMS; i++) {
			const uint32_t li = (uint32_t)i * ZS_KEY_THREADS + threadIdx.x;
			const uint32_t key = morton_low(q[i].x, q[i].y, q[i].z, g);   // (30 bits; a NaN x: all ones -- behind everything, like the full key)
			if (li < rem) pairs[base + li] = make_uint2(key, (uint32_t)(base + li));
			const uint64_t live = __ballot(li < rem);
			if (live != 0ull) {
				const int first = __builtin_ctzll(live);
				for (int p = 0; p < plan.passes; p++) {
					const uint32_t d = (key >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u);
					const uint32_t d0 = readlane_u32(d, first);
					if (__ballot(li < rem && d == d0) == live) { if (lane_id() == first) atomicAdd(&h[p * R + d0], (uint32_t)__popcll(live)); }
					else if (li < rem) atomicAdd(&h[p * R + d], 1u);
				}
			}
		}
	}
	__syncthreads();
	for (int k = threadIdx.x; k < plan.passes * R; k += ZS_KEY_THREADS) part[(size_t)blockIdx.x * (ZS_MAX_PASSES * R) + k] = h[k];
}
// digit totals of every pass: column sums of the partial histograms (the exclusive scan over the digit values is done by every tile of k_zs_pass itself: 512
// values).  A workgroup owns 64 digit values of one pass, sixteen threads share a column.
__global__ void __launch_bounds__(1024) k_zs_totals(const uint32_t* __restrict__ part, int n_part, uint32_t* __restrict__ totals)
{
	constexpr int R = 1 << ZS_MAX_BITS;
	__shared__ uint32_t red[16][64];
	const int p = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
	uint32_t c = 0;
	for (int k = kg; k < n_part; k += 16) c += part[(size_t)k * (ZS_MAX_PASSES * R) + p * R + d];
	red[kg][threadIdx.x & 63] = c;
	__syncthreads();
	if (kg == 0) {
		uint32_t t = 0;
		#pragma unroll
		for (int q = 0; q < 16; q++) t += red[q][threadIdx.x];
		totals[p * R + d] = t;
	}
}
template <int BITS, bool LAST>
__global__ void __launch_bounds__(ZS_THREADS) k_zs_pass(const uint2* __restrict__ in, uint2* __restrict__ out, int* __restrict__ order_out, int n, int shift,
                                                        const uint32_t* __restrict__ totals, uint32_t* __restrict__ status, uint32_t* __restrict__ ticket)
{
	constexpr int RADIX = 1 << BITS;
	constexpr int PER = RADIX / ZS_THREADS > 0 ? RADIX / ZS_THREADS : 1;   // digit values per thread (1 or 2)
	static_assert(RADIX <= 2 * ZS_THREADS, "at most two digit values per thread");
	__shared__ uint16_t wcount[ZS_WAVES][RADIX];   // (16 bits: a tile has 8192 elements -- with 32-bit counters the 9-bit pass is 2 KB over what lets two workgroups share a CU)
	__shared__ uint32_t gdelta[RADIX];
	__shared__ uint2 stage[ZS_TILE];
	__shared__ uint32_t wsum[2 * ZS_WAVES];
	__shared__ uint32_t s_tile;
	const int w = (int)readfirstlane_u32(threadIdx.x / WAVE), lane = lane_id();
	if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
	for (int k = threadIdx.x; k < ZS_WAVES * RADIX; k += ZS_THREADS) (&wcount[0][0])[k] = 0;
	__syncthreads();
	const uint32_t tile = readfirstlane_u32(s_tile);
	const size_t tbase = (size_t)tile * ZS_TILE;
	const uint32_t trem = (uint32_t)((size_t)n - tbase < (size_t)ZS_TILE ? (size_t)n - tbase : (size_t)ZS_TILE);   // elements of this tile
	// ---- this wave's ZS_ITEMS * 64 consecutive elements, all loads up front
	const uint32_t woff = (uint32_t)w * (ZS_ITEMS * WAVE);
	const uint32_t rem = trem > woff ? (trem - woff < (uint32_t)(ZS_ITEMS * WAVE) ? trem - woff : (uint32_t)(ZS_ITEMS * WAVE)) : 0u;
	uint2 e[ZS_ITEMS];
	#pragma unroll
	for (int i = 0; i < ZS_ITEMS; i++) {
		const uint32_t li = (uint32_t)(i * WAVE + lane);
		e[i] = in[tbase + (rem ? woff + (li < rem ? li : rem - 1u) : 0u)];
	}
	// ---- stable rank inside the wave's sub-tile (rounds of 64 in index order; ballot matching of the digit's bits)
	uint32_t dig_rank[ZS_ITEMS];
	#pragma unroll
	for (int i = 0; i < ZS_ITEMS; i++) {
		const bool valid = (uint32_t)(i * WAVE + lane) < rem;
		const uint32_t d = (e[i].x >> shift) & (uint32_t)(RADIX - 1);
		uint64_t peers = __ballot(valid);
		#pragma unroll
		for (int b = 0; b < BITS; b++) {
			const bool bit = (d >> b) & 1u;
			const uint64_t m = __ballot(valid && bit);
			peers &= bit ? m : ~m;
		}
		const uint32_t r = mbcnt64(peers);
		const uint32_t cnt = (uint32_t)__popcll(peers);
		uint32_t prev = 0;
		if (valid) prev = wcount[w][d];
		wave_lds_fence();
		if (valid && r == 0) wcount[w][d] = (uint16_t)(prev + cnt);
		wave_lds_fence();
		dig_rank[i] = d | ((prev + r) << 16);
	}
	__syncthreads();
	// ---- per digit value: the tile's count, its start inside the sorted tile, and -- by look-back -- the count of all tiles before this one
	uint32_t cnt_d[PER], sum = 0;
	#pragma unroll
	for (int k = 0; k < PER; k++) {
		const int d = (int)threadIdx.x * PER + k;
		uint32_t c = 0;
		if (d < RADIX) {
			#pragma unroll
			for (int ww = 0; ww < ZS_WAVES; ww++) c += wcount[ww][d];
			// (published before anything else: the tiles behind this one are waiting for it)
			__hip_atomic_store(status + (size_t)tile * RADIX + d, (tile == 0u ? ZS_FLAG_INC : ZS_FLAG_AGG) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		cnt_d[k] = c; sum += c;
	}
	// (two scans over the digit values at once: the tile's counts -> where a digit starts inside the sorted tile; the set's totals -> where it starts in the output)
	uint32_t tot_d[PER], gsum = 0;
	#pragma unroll
	for (int k = 0; k < PER; k++) { const int d = (int)threadIdx.x * PER + k; tot_d[k] = d < RADIX ? totals[d] : 0u; gsum += tot_d[k]; }
	uint32_t inc = sum, ginc = gsum;
	#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) { const uint32_t u = __shfl_up(inc, o, WAVE), gu = __shfl_up(ginc, o, WAVE); if (lane >= o) { inc += u; ginc += gu; } }
	if (lane == WAVE - 1) { wsum[w] = inc; wsum[ZS_WAVES + w] = ginc; }
	__syncthreads();
	uint32_t tstart = inc - sum, gstart = ginc - gsum;
	#pragma unroll
	for (int ww = 0; ww < ZS_WAVES; ww++) if (ww < w) { tstart += wsum[ww]; gstart += wsum[ZS_WAVES + ww]; }
	#pragma unroll
	for (int k = 0; k < PER; k++) {
		const int d = (int)threadIdx.x * PER + k;
		if (d < RADIX) {
			uint32_t before = 0;
			if (tile != 0u) {
				for (uint32_t t = tile - 1u;; t--) {
					uint32_t v;
					do { v = __hip_atomic_load(status + (size_t)t * RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((v >> 30) == 0u);
					before += v & ZS_VALUE;
					if ((v >> 30) == 2u) break;
				}
				__hip_atomic_store(status + (size_t)tile * RADIX + d, ZS_FLAG_INC | (before + cnt_d[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			gdelta[d] = gstart + before - tstart;   // global position of an element = its index in the sorted tile + this
			// where each wave's elements of this digit start inside the sorted tile
			uint32_t s0 = tstart;
			#pragma unroll
			for (int ww = 0; ww < ZS_WAVES; ww++) { const uint32_t c = wcount[ww][d]; wcount[ww][d] = (uint16_t)s0; s0 += c; }
			tstart += cnt_d[k]; gstart += tot_d[k];
		}
	}
	__syncthreads();
	// ---- the tile, sorted by this digit, in LDS
	#pragma unroll
	for (int i = 0; i < ZS_ITEMS; i++)
		if ((uint32_t)(i * WAVE + lane) < rem) stage[wcount[w][dig_rank[i] & 0xffffu] + (dig_rank[i] >> 16)] = e[i];
	__syncthreads();
	// ---- and out: consecutive threads write consecutive elements of a digit's run
	#pragma unroll
	for (int i = 0; i < ZS_ITEMS; i++) {
		const uint32_t j = (uint32_t)i * ZS_THREADS + threadIdx.x;
		if (j < trem) {
			const uint2 v = stage[j];
			const uint32_t pos = j + gdelta[(v.x >> shift) & (uint32_t)(RADIX - 1)];
			if (LAST) order_out[pos] = (int)v.y; else out[pos] = v;
		}
	}
}
template <int BITS>
static void zs_pass(bool last, const uint2* in, uint2* out, int* order_out, int n, int shift, const uint32_t* base, uint32_t* status, uint32_t* ticket, hipStream_t s)
{
	if (last) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zs_pass<BITS, true>), dim3(zs_tiles(n)), dim3(ZS_THREADS), 0, s, in, out, order_out, n, shift, base, status, ticket);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zs_pass<BITS, false>), dim3(zs_tiles(n)), dim3(ZS_THREADS), 0, s, in, out, order_out, n, shift, base, status, ticket);
}
static void launch_zsort_pairs(const float* xyz, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, int* order_out, hipStream_t s)
{
	constexpr size_t R = (size_t)1 << ZS_MAX_BITS;
	const ZsPlan plan = zs_plan(key_bits);
	ZsPlanDev pd{};
	pd.passes = plan.passes;
	for (int i = 0; i < plan.passes; i++) { pd.bits[i] = plan.bits[i]; pd.shift[i] = plan.shift[i]; }
	const int kgrid = zs_key_grid(n), tiles = zs_tiles(n);
	uint32_t* part = (uint32_t*)temp;
	uint32_t* base = part + (size_t)ZS_KEY_GRID * ZS_MAX_PASSES * R;
	uint32_t* ticket = base + (size_t)ZS_MAX_PASSES * R;
	uint32_t* status = ticket + 64;
	// the status words of all passes and the tickets: zero (one fill; a tile's word becomes non-zero exactly once)
	(void)hipMemsetAsync(ticket, 0, (64 + (size_t)plan.passes * (size_t)tiles * R) * sizeof(uint32_t), s);
	uint2* buf[2] = { reinterpret_cast<uint2*>(b.xyzi[0]), reinterpret_cast<uint2*>(b.xyzi[1]) };   // (the ping-pong arrays of the search structure: 16 bytes per point each)
	hipLaunchKernelGGL(k_zs_keys, dim3(kgrid), dim3(ZS_KEY_THREADS), 0, s, xyz, n, g, pd, buf[0], part);
	hipLaunchKernelGGL(k_zs_totals, dim3((unsigned)(R / 64), (unsigned)plan.passes), dim3(1024), 0, s, part, kgrid, base);
	int cur = 0;
	for (int p = 0; p < plan.passes; p++) {
		const bool last = p == plan.passes - 1;
		uint32_t* st = status + (size_t)p * (size_t)tiles * R;
		switch (plan.bits[p]) {
		case 8:  zs_pass<8>(last, buf[cur], buf[cur ^ 1], order_out, n, plan.shift[p], base + p * R, st, ticket + p, s); break;
		case 9:  zs_pass<9>(last, buf[cur], buf[cur ^ 1], order_out, n, plan.shift[p], base + p * R, st, ticket + p, s); break;
		default: zs_pass<10>(last, buf[cur], buf[cur ^ 1], order_out, n, plan.shift[p], base + p * R, st, ticket + p, s); break;
		}
		cur ^= 1;
	}
}

int launch_morton_sort(const float* xyz, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, int* order_out, hipStream_t s)
{
	if (n >= (1 << 16) && key_bits >= 16 && key_bits <= 24 && key_bits < TNSX_ZS_PAIRS_FROM) {
		int lo_bits = key_bits - CS_MAX_BITS;
		lo_bits = lo_bits < 8 ? 8 : lo_bits;
		const int hi_bits = key_bits - lo_bits;   // 8 .. 11
		const int ntiles = cs_num_tiles(n);
		const size_t hist_cap = ((size_t)1 << CS_MAX_BITS) * (size_t)ntiles;
		uint32_t* hist = (uint32_t*)temp;
		uint32_t* totals = (uint32_t*)((char*)temp + ((hist_cap * sizeof(uint32_t) + 255) / 256) * 256);
		uint32_t* strip_sums = totals + ((size_t)1 << CS_MAX_BITS);
		const BuildGuard none{};
		TNSX_CS_DISPATCH(hi_bits, (cs_hist<B, true>(true, xyz, b.xyzi[0], n, g, lo_bits, hist, ntiles, nullptr, none, s)));
		TNSX_CS_DISPATCH(hi_bits, cs_scan<B>(hist, ntiles, strip_sums, totals, s));
		TNSX_CS_DISPATCH(hi_bits, (cs_scatter<B, true>(true, false, xyz, nullptr, b.xyzi[0], b.r2[0], b.xyzi[1], b.r2[1], n, g, lo_bits, hist, totals, ntiles, nullptr, nullptr, none, s)));
		const size_t lds = ((size_t)1 << lo_bits) * sizeof(uint32_t);
		hipLaunchKernelGGL(k_morton_place, dim3(1 << hi_bits), dim3(BP_THREADS), lds, s, b.xyzi[1], order_out, g, lo_bits, totals);
		return 1;
	}
	if (zs_uses_pairs(n, key_bits)) {   // round 5: {key, index} pairs through single-pass digit sorts (above)
		launch_zsort_pairs(xyz, n, g, key_bits, b, temp, order_out, s);
		return 1;
	}
	const int res = point_sort<true>(xyz, nullptr, n, g, key_bits, b, temp, nullptr, nullptr, BuildGuard{}, s);
	if (n > 0) hipLaunchKernelGGL(k_extract_order, dim3((n + 255) / 256), dim3(256), 0, s, b.xyzi[res], n, order_out);
	return res;
}

// =====================================================================================================
// cell table + list of occupied cells from the sorted points.  One block per tile of 4096 sorted points, ONE atomic per block.
// =====================================================================================================
static constexpr int CT_THREADS = 256;
static constexpr int CT_ITEMS = 16;
static constexpr int CT_TILE = CT_THREADS * CT_ITEMS;

__global__ void __launch_bounds__(CT_THREADS) k_cell_table(const float4* __restrict__ xyzi, int n, GridParams g, uint2* __restrict__ table,
                                                          uint2* __restrict__ occ, uint32_t* __restrict__ n_occ)
{
	__shared__ uint32_t sk[CT_TILE + 2];                        // keys of the tile, one halo entry on either side
	__shared__ uint32_t wcnt[CT_ITEMS * (CT_THREADS / WAVE)];   // [round][wave] -> exclusive prefix
	__shared__ uint32_t block_base;
	const int w = threadIdx.x / WAVE;
	const size_t base = (size_t)blockIdx.x * CT_TILE;
	const uint32_t n_cells = (uint32_t)(g.nx * g.ny * g.nz);
	#pragma unroll
	for (int i = 0; i < CT_ITEMS; i++) {
		const size_t p = base + (size_t)i * CT_THREADS + threadIdx.x;
		if (p < (size_t)n) { const float4 q = xyzi[p]; sk[1 + i * CT_THREADS + threadIdx.x] = cell_key(q.x, q.y, q.z, g); }
	}
	if (threadIdx.x == 0 && base > 0) { const float4 q = xyzi[base - 1]; sk[0] = cell_key(q.x, q.y, q.z, g); }
	if (threadIdx.x == 64 && base + CT_TILE < (size_t)n) { const float4 q = xyzi[base + CT_TILE]; sk[CT_TILE + 1] = cell_key(q.x, q.y, q.z, g); }
	__syncthreads();
	uint32_t flags = 0;
	#pragma unroll
	for (int i = 0; i < CT_ITEMS; i++) {
		const int t = i * CT_THREADS + (int)threadIdx.x;
		const size_t p = base + (size_t)t;
		bool is_start = false;
		if (p < (size_t)n) {
			const uint32_t k = sk[1 + t];
			if (k < n_cells) {                           // (k == n_cells: NaN points, behind all cells; they enter no cell)
				is_start = (p == 0) || (sk[t] != k);
				const bool is_end = (p == (size_t)n - 1) || (sk[2 + t] != k);
				if (is_start) table[k].x = (uint32_t)p;
				if (is_end) table[k].y = (uint32_t)p + 1u;
			}
		}
		flags |= (is_start ? 1u : 0u) << i;
		const uint64_t m = __builtin_amdgcn_ballot_w64(is_start);
		if (lane_id() == 0) wcnt[i * (CT_THREADS / WAVE) + w] = (uint32_t)__popcll(m);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t s = 0;
		for (int q = 0; q < CT_ITEMS * (CT_THREADS / WAVE); q++) { const uint32_t t = wcnt[q]; wcnt[q] = s; s += t; }
		block_base = s ? atomicAdd(n_occ, s) : 0u;
	}
	__syncthreads();
	const uint32_t bb = block_base;
	#pragma unroll
	for (int i = 0; i < CT_ITEMS; i++) {
		const bool is_start = (flags >> i) & 1u;
		const uint64_t m = __builtin_amdgcn_ballot_w64(is_start);
		if (is_start) {
			const int t = i * CT_THREADS + (int)threadIdx.x;
			occ[bb + wcnt[i * (CT_THREADS / WAVE) + w] + mbcnt64(m)] = make_uint2((uint32_t)(base + (size_t)t), sk[1 + t]);
		}
	}
}
void launch_cell_table(const float4* xyzi_sorted, int n, GridParams g, uint2* table, uint2* occ, uint32_t* n_occ, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_cell_table, dim3((n + CT_TILE - 1) / CT_TILE), dim3(CT_THREADS), 0, s, xyzi_sorted, n, g, table, occ, n_occ);
}

// =====================================================================================================
// SPARSE grids (round 4): no dense cell table at all.  A cloud that is sparse everywhere -- a sheet, a filament -- has a bounding box of billions of
// cells of one search radius; rounds 1-3 coarsened the cells until a dense table fitted (exact, but every query then tests 8x the candidates per doubling).
// Here the cells keep their edge (any grid of up to 2^32 - 16 cells: 32-bit keys) and the search structure is the list of OCCUPIED cells in key order,
// {first sorted position, key} with a sentinel behind it (a cell ends where the next one begins), plus a block index: blk[b] = first list entry whose key
// is >= b << shift, one entry per 2^shift keys (at most 4 M entries).  A lookup is blk[b], blk[b + 1] and a binary search over the handful of entries between
// them (tnsx_device.h sparse_find): three or four dependent loads instead of one -- slower per cell than the dense table, and far cheaper than coarser cells.
//   k_sparse_cells<false>  cell starts per tile of 4096 sorted points        -> counts[tile]
//   k_sparse_scan          exclusive scan of the tile counts (one workgroup)  -> counts[tile] = first list entry of the tile, *n_occ
//   k_sparse_cells<true>   the entries, in position (= key) order; the sentinel behind the last one
//   k_sparse_blocks        the block index
// NaN-x points (key == n_cells: no points) sort behind every cell; the sentinel starts where they start.
// =====================================================================================================
template <bool WRITE>
__global__ void __launch_bounds__(CT_THREADS) k_sparse_cells(const float4* __restrict__ xyzi, int n, GridParams g, uint32_t* __restrict__ counts, uint2* __restrict__ occ)
{
	__shared__ uint32_t sk[CT_TILE + 1];                        // keys of the tile, one halo entry in front
	__shared__ uint32_t wcnt[CT_ITEMS * (CT_THREADS / WAVE) + 1];
	const int w = threadIdx.x / WAVE;
	const size_t base = (size_t)blockIdx.x * CT_TILE;
	const uint32_t n_cells = (uint32_t)g.nx * (uint32_t)g.ny * (uint32_t)g.nz;
	#pragma unroll
	for (int i = 0; i < CT_ITEMS; i++) {
		const size_t p = base + (size_t)i * CT_THREADS + threadIdx.x;
		if (p < (size_t)n) { const float4 q = xyzi[p]; sk[1 + i * CT_THREADS + threadIdx.x] = cell_key(q.x, q.y, q.z, g); }
	}
	if (threadIdx.x == 0 && base > 0) { const float4 q = xyzi[base - 1]; sk[0] = cell_key(q.x, q.y, q.z, g); }
	__syncthreads();
	uint32_t flags = 0;       // bit i: a cell starts at this thread's point of round i (real cells only)
	uint32_t nan_start = 0;   // bit i: the NaN points start there (at most one such point in the whole array)
	#pragma unroll
	for (int i = 0; i < CT_ITEMS; i++) {
		const int t = i * CT_THREADS + (int)threadIdx.x;
		const size_t p = base + (size_t)t;
		bool is_start = false;
		if (p < (size_t)n) {
			const uint32_t k = sk[1 + t];
			const bool first_of_key = (p == 0) || (sk[t] != k);
			is_start = first_of_key && k < n_cells;
			nan_start |= (first_of_key && k >= n_cells ? 1u : 0u) << i;
		}
		flags |= (is_start ? 1u : 0u) << i;
		const uint64_t m = __builtin_amdgcn_ballot_w64(is_start);
		if (lane_id() == 0) wcnt[i * (CT_THREADS / WAVE) + w] = (uint32_t)__popcll(m);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t s = 0;
		for (int q = 0; q < CT_ITEMS * (CT_THREADS / WAVE); q++) { const uint32_t t = wcnt[q]; wcnt[q] = s; s += t; }
		wcnt[CT_ITEMS * (CT_THREADS / WAVE)] = s;
		if (!WRITE) counts[blockIdx.x] = s;
	}
	if (!WRITE) return;
	__syncthreads();
	const uint32_t bb = counts[blockIdx.x];   // (scanned: the first list entry of this tile)
	#pragma unroll
	for (int i = 0; i < CT_ITEMS; i++) {
		const bool is_start = (flags >> i) & 1u;
		const uint64_t m = __builtin_amdgcn_ballot_w64(is_start);
		const int t = i * CT_THREADS + (int)threadIdx.x;
		const uint32_t rank = bb + wcnt[i * (CT_THREADS / WAVE) + w] + mbcnt64(m);
		if (is_start) occ[rank] = make_uint2((uint32_t)(base + (size_t)t), sk[1 + t]);
		else if ((nan_start >> i) & 1u) occ[rank] = make_uint2((uint32_t)(base + (size_t)t), 0xffffffffu);   // the sentinel: every real start lies in front of it
	}
	// no NaN point at all: the sentinel stands behind the last point
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
		const float4 q = xyzi[n - 1];
		if (cell_key(q.x, q.y, q.z, g) < n_cells) occ[bb + wcnt[CT_ITEMS * (CT_THREADS / WAVE)]] = make_uint2((uint32_t)n, 0xffffffffu);
	}
}
__global__ void __launch_bounds__(1024) k_sparse_scan(uint32_t* __restrict__ counts, int ntiles, uint32_t* __restrict__ n_occ)
{
	__shared__ uint32_t red[1024 / WAVE];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0) carry = 0u;
	__syncthreads();
	for (int t0 = 0; t0 < ntiles; t0 += 1024) {
		const int t = t0 + (int)threadIdx.x;
		const uint32_t v = t < ntiles ? counts[t] : 0u;
		uint32_t inc = v;
		#pragma unroll
		for (int o = 1; o < WAVE; o <<= 1) { const uint32_t u = __shfl_up(inc, o, WAVE); if (lane_id() >= o) inc += u; }
		if (lane_id() == WAVE - 1) red[threadIdx.x / WAVE] = inc;
		__syncthreads();
		uint32_t before = carry;
		for (int k = 0; k < (int)(threadIdx.x / WAVE); k++) before += red[k];
		if (t < ntiles) counts[t] = before + inc - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry = before + inc;
		__syncthreads();
	}
	if (threadIdx.x == 0) *n_occ = carry;
}
// blk[b] = first entry of occ (n_occ entries + sentinel) whose key is >= b << shift, b = 0 .. n_blocks (blk[n_blocks] = n_occ): one lower bound per
// block (a thread per entry filling the gap in front of it would be cheaper on average and arbitrarily unbalanced on a cloud that sits in a few cells)
__global__ void __launch_bounds__(256) k_sparse_blocks(const uint2* __restrict__ occ, const uint32_t* __restrict__ n_occ_p, int shift, uint32_t n_blocks, uint32_t* __restrict__ blk)
{
	const uint32_t n_occ = *n_occ_p;
	for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b <= n_blocks; b += gridDim.x * 256u) {
		uint32_t lo = 0, hi = n_occ;
		if (b < n_blocks) {
			const uint32_t key = b << shift;
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (occ[mid].y < key) lo = mid + 1u; else hi = mid; }
		}
		else lo = n_occ;
		blk[b] = lo;
	}
}
void launch_sparse_cells(const float4* xyzi_sorted, int n, GridParams g, int shift, uint32_t n_blocks, void* temp, uint2* occ, uint32_t* n_occ, uint32_t* blk, hipStream_t s)
{
	if (n <= 0) return;
	const int ntiles = (n + CT_TILE - 1) / CT_TILE;
	uint32_t* counts = (uint32_t*)temp;   // (the sort is done with its tables: ntiles words)
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sparse_cells<false>), dim3(ntiles), dim3(CT_THREADS), 0, s, xyzi_sorted, n, g, counts, occ);
	hipLaunchKernelGGL(k_sparse_scan, dim3(1), dim3(1024), 0, s, counts, ntiles, n_occ);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sparse_cells<true>), dim3(ntiles), dim3(CT_THREADS), 0, s, xyzi_sorted, n, g, counts, occ);
	const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n_blocks + 256) / 256, 8192);
	hipLaunchKernelGGL(k_sparse_blocks, dim3(blocks), dim3(256), 0, s, occ, n_occ, shift, n_blocks, blk);
}

// =====================================================================================================
// The build in TWO passes with the cell table for free (round 3): bucket pass + bucket-local counting sort.
//   pass A  the ordinary histogram / scan / ranked-scatter pass on the HIGH digit of the cell key (it also carries the run-time
//           checks of the speculation): the points land grouped by bucket = 2^lo consecutive cells, i.e. a few x-rows of the grid
//   pass B  k_bucket_sort: ONE workgroup per bucket.  Sweep 1 counts the bucket's points per cell in LDS (the low digit IS the cell
//           inside the bucket); the scan of the counts is the cell table of the bucket -- first / one-past-last position of every
//           cell and the list of occupied cells fall out of it, no key comparison between neighbouring points, no second read of
//           the sorted array; sweep 2 re-reads the bucket (it is a few hundred KB: L2) and drops every point at the cursor of
//           its cell (ds_add_rtn).  Its stores stay inside the bucket's own window of the output, where the L2 completes them
//           to full lines.
// Against the two LSD passes + k_cell_table this saves one histogram pass over all points, the scan kernels of the second
// pass, the 16 N bytes k_cell_table reads, and the BITS ballots per point the ranked scatter spends on a stable rank.
// The order of the points INSIDE a cell is whatever the LDS atomics give (the neighbour SETS do not depend on it; the exact
// two-pass layout, which promises a reproducible order, keeps the stable LSD sort), with one exception that the engine relies on:
// points that get lists (original index < query_limit) come before the candidates-only points of their cell (the ghosts of a
// slab) -- the former fill a cell from the front, the latter from the back.
// =====================================================================================================
#ifndef TNSX_NT_BUCKET_LOADS
#define TNSX_NT_BUCKET_LOADS 1
#endif
#if TNSX_NT_BUCKET_LOADS
__device__ __forceinline__ float4 ld_bucket_nt(const float4* p) { const float* f = reinterpret_cast<const float*>(p); typedef float v4 __attribute__((ext_vector_type(4))); const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(f)); return make_float4(v.x, v.y, v.z, v.w); }
#define TNSX_LD_BUCKET(p) ld_bucket_nt(p)
#else
#define TNSX_LD_BUCKET(p) (*(p))
#endif
static constexpr int BS_THREADS = 1024;
static constexpr int BS_KEEP = 12;     // points per thread that stay in registers between the two sweeps (12 K points per bucket: all of them, usually);
                                       // their loads are issued back to back -- with one load in flight per thread the sweeps are latency-bound
static constexpr int BS_UNROLL = 4;    // loads in flight per thread for the points beyond that
template <bool VARIABLE, bool SPLIT>
__global__ void __launch_bounds__(BS_THREADS) k_bucket_sort(const float4* __restrict__ in, float4* __restrict__ out, float* __restrict__ r2_out,
                                                            const float* __restrict__ radii, GridParams g, int lo_bits, const uint32_t* __restrict__ totals, int tstride,
                                                            const uint2* __restrict__ win, uint2* __restrict__ table, uint2* __restrict__ occ_tmp, uint2* __restrict__ bucket_info,
                                                            uint32_t* __restrict__ n_occ, const int* __restrict__ ids, uint32_t* __restrict__ orig_out,
                                                            uint32_t query_limit, BuildGuard gd)
{
	extern __shared__ uint32_t bs_lds[];   // front cursor per cell of the bucket [+ back cursor per cell (SPLIT)]
	__shared__ uint32_t red[2 * (BS_THREADS / WAVE) + 2];
	const int RADIX = 1 << lo_bits;
	uint32_t* const h = bs_lds;
	uint32_t* const hb = bs_lds + RADIX;
	const int b = (int)blockIdx.x, lane = lane_id(), w = (int)threadIdx.x / WAVE;
	const uint32_t n_cells = (uint32_t)(g.nx * g.ny * g.nz);
	// ---- where the bucket lies in the output of pass A: the counts of the buckets before it
	uint32_t part = 0;
	// (totals: the dense totals of the histogram pass, or -- tstride > 1 -- the cursors the one-read bucket pass left behind)
	for (int k = (int)threadIdx.x; k < b; k += BS_THREADS) part += totals[(size_t)k * tstride];
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
	if (lane == 0) red[w] = part;
	// (one-read pass: the cursor holds what the tiles ASKED the window for.  More than it holds means that the window overflowed, the guard flag
	//  is up and this attempt will be thrown away -- but it must not read past the window, which for the last buckets is the end of the array)
	const uint32_t count = win ? min(totals[(size_t)b * tstride], win[b].y) : totals[(size_t)b * tstride];
	for (int k = (int)threadIdx.x; k < RADIX; k += BS_THREADS) h[k] = 0u;
	// One-read pass whose guard is already up (a window overflowed in k_bucket_scatter, earlier on this stream): the window has slots that nobody wrote -- stale
	// points of a run with another n, or memory that was never written -- and their w component must not be used as an index into radii[] / ids[].  The attempt
	// is thrown away by the host anyway: the bucket reports no cells and leaves.  (One thread reads the word for everybody: other workgroups of this launch may
	// raise it meanwhile, and the whole workgroup has to take the same side of the barriers below.)
	if (threadIdx.x == 0) red[2 * (BS_THREADS / WAVE) + 1] = (win && gd.flag) ? __hip_atomic_load(gd.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
	__syncthreads();
	uint32_t start = 0;
	#pragma unroll
	for (int k = 0; k < BS_THREADS / WAVE; k++) start += red[k];
	if (count == 0u || red[2 * (BS_THREADS / WAVE) + 1] != 0u) { if (threadIdx.x == 0) bucket_info[b] = make_uint2(0u, 0u); return; }
	// where the bucket's points are: packed behind the buckets before it (histogram pass), or in the bucket's own window (one-read pass)
	if (win) in += win[b].x; else in += start;
	// ---- the bucket's points -> registers (all loads in flight at once), sweep 1: points per cell
	float4 keep[BS_KEEP];
	#pragma unroll
	for (int u = 0; u < BS_KEEP; u++) { const uint32_t i = (uint32_t)u * BS_THREADS + threadIdx.x; keep[u] = TNSX_LD_BUCKET(in + (i < count ? i : count - 1u)); }
	#pragma unroll
	for (int u = 0; u < BS_KEEP; u++) {
		const uint32_t i = (uint32_t)u * BS_THREADS + threadIdx.x;
		if (i < count) atomicAdd(&h[cell_key(keep[u].x, keep[u].y, keep[u].z, g) & (uint32_t)(RADIX - 1)], 1u);
	}
	for (uint32_t i0 = BS_KEEP * BS_THREADS; i0 < count; i0 += BS_THREADS * BS_UNROLL) {   // (a bucket larger than the registers hold: read twice)
		float4 q[BS_UNROLL];
		#pragma unroll
		for (int u = 0; u < BS_UNROLL; u++) { const uint32_t i = i0 + (uint32_t)u * BS_THREADS + threadIdx.x; q[u] = TNSX_LD_BUCKET(in + (i < count ? i : count - 1u)); }
		#pragma unroll
		for (int u = 0; u < BS_UNROLL; u++) {
			const uint32_t i = i0 + (uint32_t)u * BS_THREADS + threadIdx.x;
			if (i < count) atomicAdd(&h[cell_key(q[u].x, q[u].y, q[u].z, g) & (uint32_t)(RADIX - 1)], 1u);
		}
	}
	__syncthreads();
	// ---- scan of the counts = the cell table of the bucket.  Thread t owns the PER consecutive cells [t * PER, t * PER + PER).
	const int PER = RADIX >= BS_THREADS ? RADIX / BS_THREADS : 1;   // (fewer cells than threads: the upper threads own none)
	const int my_bins = (int)threadIdx.x * PER < RADIX ? PER : 0;
	const uint32_t key0 = ((uint32_t)b << lo_bits) + (uint32_t)threadIdx.x * (uint32_t)PER;
	uint32_t sum = 0, nz = 0;
	for (int k = 0; k < my_bins; k++) { const uint32_t c = h[threadIdx.x * PER + k]; sum += c; nz += (c != 0u && key0 + (uint32_t)k < n_cells) ? 1u : 0u; }
	uint32_t inc = sum, ninc = nz;
	#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) { const uint32_t u = __shfl_up(inc, o, WAVE), v = __shfl_up(ninc, o, WAVE); if (lane >= o) { inc += u; ninc += v; } }
	__syncthreads();   // (red is read above by everybody)
	if (lane == WAVE - 1) { red[w] = inc; red[BS_THREADS / WAVE + w] = ninc; }
	__syncthreads();
	uint32_t ex = inc - sum, nex = ninc - nz, nz_total = 0;
	#pragma unroll
	for (int k = 0; k < BS_THREADS / WAVE; k++) { if (k < w) { ex += red[k]; nex += red[BS_THREADS / WAVE + k]; } nz_total += red[BS_THREADS / WAVE + k]; }
	if (threadIdx.x == 0) {
		const uint32_t base = nz_total ? atomicAdd(n_occ, nz_total) : 0u;   // ONE atomic per bucket; k_occ_reorder puts the buckets' pieces into key order
		red[2 * (BS_THREADS / WAVE)] = base;
		bucket_info[b] = make_uint2(base, nz_total);
	}
	__syncthreads();
	const uint32_t obase = red[2 * (BS_THREADS / WAVE)];
	for (int k = 0; k < my_bins; k++) {
		const int bin = (int)threadIdx.x * PER + k;
		const uint32_t c = h[bin];
		h[bin] = ex;                       // front cursor
		if (SPLIT) hb[bin] = ex + c;       // back cursor (one past)
		if (c != 0u && key0 + (uint32_t)k < n_cells) {   // (key == n_cells: NaN points, behind all cells; they enter no cell)
			table[key0 + (uint32_t)k] = make_uint2(start + ex, start + ex + c);
			occ_tmp[obase + nex] = make_uint2(start + ex, key0 + (uint32_t)k);
			nex++;
		}
		ex += c;
	}
	__syncthreads();
	// ---- sweep 2: every point to the cursor of its cell
	bool bad_r = false;
	auto place = [&](const float4 q, const float r) {
		const uint32_t low = cell_key(q.x, q.y, q.z, g) & (uint32_t)(RADIX - 1);
		const uint32_t o = __float_as_uint(q.w);
		uint32_t pos;
		if (SPLIT && o >= query_limit) pos = atomicSub(&hb[low], 1u) - 1u;
		else pos = atomicAdd(&h[low], 1u);
		pos += start;
		float wv = q.w;
		if (ids) { wv = __int_as_float(ids[o]); orig_out[pos] = o; }   // (tnsx_set_point_ids: the point carries its id from here on)
		out[pos] = make_float4(q.x, q.y, q.z, wv);
		if (VARIABLE) { r2_out[pos] = __fmul_rn(r, r); bad_r |= (r > gd.r_max) & (q.x == q.x); }
	};
	{
		// gather of the radii by original index (the radii array of a 50 M-point set is 200 MB: it stays in the 256 MB Infinity Cache)
		float rr[BS_KEEP];
		#pragma unroll
		for (int u = 0; u < BS_KEEP; u++) rr[u] = VARIABLE ? radii[__float_as_uint(keep[u].w)] : 0.0f;
		#pragma unroll
		for (int u = 0; u < BS_KEEP; u++) if ((uint32_t)u * BS_THREADS + threadIdx.x < count) place(keep[u], rr[u]);
	}
	for (uint32_t i0 = BS_KEEP * BS_THREADS; i0 < count; i0 += BS_THREADS * BS_UNROLL) {
		float4 qq[BS_UNROLL];
		float rr[BS_UNROLL];
		#pragma unroll
		for (int u = 0; u < BS_UNROLL; u++) { const uint32_t i = i0 + (uint32_t)u * BS_THREADS + threadIdx.x; qq[u] = TNSX_LD_BUCKET(in + (i < count ? i : count - 1u)); }
		#pragma unroll
		for (int u = 0; u < BS_UNROLL; u++) rr[u] = VARIABLE ? radii[__float_as_uint(qq[u].w)] : 0.0f;
		#pragma unroll
		for (int u = 0; u < BS_UNROLL; u++) if (i0 + (uint32_t)u * BS_THREADS + threadIdx.x < count) place(qq[u], rr[u]);
	}
	// (speculated grid: a radius above the one the cell edge was chosen for -> the host repeats the run with fresh bounds)
	if (VARIABLE && gd.flag && __builtin_amdgcn_ballot_w64(bad_r) != 0ull && lane == 0) atomicOr(gd.flag, 1u);
}
// the buckets' pieces of the occupied-cell list -> key order (what the query's XCD-contiguous work split wants).
// Also, for the NEXT run's one-read bucket pass: win_next[b] = {first slot, capacity} of bucket b's window of the intermediate array, the capacity
// this run's count + 1/8 + 64 (points move by a fraction of a cell per step: the counts of a bucket of several rows change slowly)
__host__ __device__ __forceinline__ uint32_t bucket_window_cap(uint32_t count) { return (count + (count >> 3) + 64u + 7u) & ~7u; }
__global__ void __launch_bounds__(BS_THREADS) k_occ_reorder(const uint2* __restrict__ occ_tmp, uint2* __restrict__ occ, const uint2* __restrict__ bucket_info,
                                                            const uint32_t* __restrict__ totals, int tstride, uint2* __restrict__ win_next)
{
	__shared__ uint32_t red[2 * (BS_THREADS / WAVE)];
	const int b = (int)blockIdx.x;
	const uint2 me = bucket_info[b];
	uint32_t part = 0, wpart = 0;
	for (int k = (int)threadIdx.x; k < b; k += BS_THREADS) { part += bucket_info[k].y; if (win_next) wpart += bucket_window_cap(totals[(size_t)k * tstride]); }
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { part += __shfl_xor(part, o, WAVE); wpart += __shfl_xor(wpart, o, WAVE); }
	if (lane_id() == 0) { red[threadIdx.x / WAVE] = part; red[BS_THREADS / WAVE + threadIdx.x / WAVE] = wpart; }
	__syncthreads();
	uint32_t prefix = 0, wprefix = 0;
	#pragma unroll
	for (int k = 0; k < BS_THREADS / WAVE; k++) { prefix += red[k]; wprefix += red[BS_THREADS / WAVE + k]; }
	if (win_next && threadIdx.x == 0) win_next[b] = make_uint2(wprefix, bucket_window_cap(totals[(size_t)b * tstride]));
	for (uint32_t i = threadIdx.x; i < me.y; i += BS_THREADS) occ[prefix + i] = occ_tmp[me.x + i];
}

// =====================================================================================================
// Pass A in ONE read (round 4).  The histogram pass + its two scan kernels exist only to tell the scatter where every bucket starts and how
// much of it the tiles before this one fill.  A steady-state step knows both well enough from the previous run: every bucket gets a WINDOW of
// the intermediate array (last run's count + 1/8 + 64 slots, k_occ_reorder), and a tile reserves its piece of a window with one returning atomic
// per bucket it touches on a cursor that starts at zero (k_run_begin).  The order of the tiles inside a window is whatever the atomics give --
// nothing downstream needs it (k_bucket_sort re-sorts the bucket by cell; the exact layout keeps the stable LSD passes).  A window that
// overflows raises the run's guard flag: the attempt is thrown away and repeated with the histogram pass, like any other failed assumption
// of a speculative run.  Tiles are large (1024 threads x 16 points) so that the ~1000 atomics of a tile of points in random order are one per
// 16 points (measured: tools/ubench/atomic_scatter.hip, 0.6 M returning atomics on cursors 128 bytes apart take 25 us on their own) and a
// tile's piece of a window is 256 contiguous bytes.  The rank of a point inside its (tile, bucket) piece is an LDS atomic.
// Carries the run-time checks of the speculation like k_cs_hist does (box guard, checksum).
// =====================================================================================================
#ifndef TNSX_B1_THREADS
#define TNSX_B1_THREADS 512
#endif
#ifndef TNSX_B1_ITEMS
#define TNSX_B1_ITEMS 16
#endif
#ifndef TNSX_NT_BUILD_LOADS
#define TNSX_NT_BUILD_LOADS 1
#endif
static constexpr int B1_THREADS = TNSX_B1_THREADS, B1_ITEMS = TNSX_B1_ITEMS, B1_TILE = B1_THREADS * B1_ITEMS;
__global__ void __launch_bounds__(B1_THREADS) k_bucket_scatter(const float* __restrict__ xyz, int n, GridParams g, int lo_bits, int n_buckets, const uint2* __restrict__ win,
                                                               uint32_t* __restrict__ cursors, float4* __restrict__ out, const float* __restrict__ radii, BuildGuard gd)
{
	extern __shared__ uint32_t b1_h[];   // per bucket: points of this tile, then where the tile's piece starts
	for (int b = threadIdx.x; b < n_buckets; b += B1_THREADS) b1_h[b] = 0u;
	__syncthreads();
	const size_t base = (size_t)blockIdx.x * B1_TILE;
	float px[B1_ITEMS], py[B1_ITEMS], pz[B1_ITEMS];
	uint32_t dr[B1_ITEMS];   // bucket | rank inside the tile's piece << 16 (a tile holds 2^14 points; at most 2^11 buckets)
	const uint32_t rem = base < (size_t)n ? (uint32_t)((size_t)n - base < (size_t)B1_TILE ? (size_t)n - base : (size_t)B1_TILE) : 0u;
	#pragma unroll
	for (int i = 0; i < B1_ITEMS; i++) {   // all loads up front, branch-free (clamped)
		const uint32_t li = (uint32_t)i * B1_THREADS + threadIdx.x;
#if TNSX_NT_BUILD_LOADS
		const float* qp = xyz + 3 * (base + (li < rem ? li : rem - 1u));
		px[i] = __builtin_nontemporal_load(qp); py[i] = __builtin_nontemporal_load(qp + 1); pz[i] = __builtin_nontemporal_load(qp + 2);
#else
		const F3 q = (reinterpret_cast<const F3*>(xyz) + base)[li < rem ? li : rem - 1u];
		px[i] = q.x; py[i] = q.y; pz[i] = q.z;
#endif
	}
	bool bad = false;
	float mn[3] = { gd.hi[0], gd.hi[1], gd.hi[2] }, mx[3] = { gd.lo[0], gd.lo[1], gd.lo[2] };
	unsigned long long chk = 0;
	uint32_t n_outside = 0;
	#pragma unroll
	for (int i = 0; i < B1_ITEMS; i++) {
		const uint32_t li = (uint32_t)i * B1_THREADS + threadIdx.x;
		if (li < rem) {
			const float x = px[i], y = py[i], z = pz[i];
			if (gd.flag) {   // (see k_cs_hist: a NaN x is no point and counts for nothing)
				const bool pt = x == x;
				const float qy = pt ? y : x, qz = pt ? z : x;
				mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
				mn[1] = fminf(mn[1], qy); mx[1] = fmaxf(mx[1], qy);
				mn[2] = fminf(mn[2], qz); mx[2] = fmaxf(mx[2], qz);
				bad |= pt & ((y != y) | (z != z));
				if (gd.outside) n_outside += pt & ((x < gd.soft_lo[0]) | (x > gd.soft_hi[0]) | (y < gd.soft_lo[1]) | (y > gd.soft_hi[1]) | (z < gd.soft_lo[2]) | (z > gd.soft_hi[2]));
			}
			if (gd.checksum) chk += point_hash((uint32_t)(base + li), x, y, z, radii ? radii[base + li] : 0.0f);
			// A NaN x is NO POINT (the padding rows of a speculative ghost message): it enters no cell and nothing ever reads it, so it is simply
			// left out here -- their number changes from step to step with the fill of the messages, and a window sized for last step's would
			// overflow for nothing.  (The sorted array then has unused slots behind its last cell.)
			dr[i] = 0xffffffffu;
			if (x == x) {
				const uint32_t d = cell_key(x, y, z, g) >> lo_bits;
				dr[i] = d | (atomicAdd(&b1_h[d], 1u) << 16);
			}
		}
	}
	if (gd.flag) bad |= mn[0] < gd.lo[0] || mn[1] < gd.lo[1] || mn[2] < gd.lo[2] || mx[0] > gd.hi[0] || mx[1] > gd.hi[1] || mx[2] > gd.hi[2];
	if (gd.flag && gd.outside && __builtin_amdgcn_ballot_w64(n_outside != 0u) != 0ull) {
		const unsigned long long w = wave_sum_u64((unsigned long long)n_outside);
		if (lane_id() == 0) atomicAdd(gd.outside, w);
	}
	if (gd.checksum) { chk = wave_sum_u64(chk); if (lane_id() == 0 && chk) atomicAdd(gd.checksum + (blockIdx.x % CHK_SLOTS) * CHK_STRIDE, chk); }
	__syncthreads();
	// ---- this tile's piece of every window it touches
	for (int b = threadIdx.x; b < n_buckets; b += B1_THREADS) {
		const uint32_t c = b1_h[b];
		if (c == 0u) continue;
		const uint2 w = win[b];
		const uint32_t first = atomicAdd(&cursors[(size_t)b * BUCKET_CURSOR_STRIDE], c);
		if (first + c > w.y) { bad = true; b1_h[b] = 0xffffffffu; }   // the window is full: nothing of this piece is written, the run is repeated
		else b1_h[b] = w.x + first;
	}
	if (gd.flag && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane_id() == 0) atomicOr(gd.flag, 1u);
	__syncthreads();
	#pragma unroll
	for (int i = 0; i < B1_ITEMS; i++) {
		const uint32_t li = (uint32_t)i * B1_THREADS + threadIdx.x;
		if (li < rem && dr[i] != 0xffffffffu) {
			const uint32_t p0 = b1_h[dr[i] & 0xffffu];
			if (p0 != 0xffffffffu) out[p0 + (dr[i] >> 16)] = make_float4(px[i], py[i], pz[i], __uint_as_float((uint32_t)(base + li)));
		}
	}
}

// bits of the two digits of the bucket build, or {0, 0} when the key is too wide for it (the LSD passes + k_cell_table then)
static bool bucket_plan(int key_bits, int n, int min_points, int& hi_bits, int& lo_bits)
{
	if (min_points < 0) return false;
	if (key_bits > 24 || n < (min_points > 0 ? min_points : (1 << 16))) return false;
	lo_bits = key_bits / 2;
	lo_bits = lo_bits < 8 ? 8 : (lo_bits > 13 ? 13 : lo_bits);
	hi_bits = key_bits - lo_bits;
	if (hi_bits < 8) hi_bits = 8;
	if (hi_bits > CS_MAX_BITS) { hi_bits = CS_MAX_BITS; lo_bits = key_bits - hi_bits; }
	return lo_bits >= 8 && lo_bits <= 13;
}
size_t cell_build_temp_bytes(int n)
{
	// the sort's tables, then occ_tmp (n entries) and bucket_info (2^CS_MAX_BITS entries) of the bucket build
	return ((cell_sort_temp_bytes(n) + 255) / 256) * 256 + (size_t)(n > 0 ? n : 1) * sizeof(uint2) + ((size_t)1 << CS_MAX_BITS) * sizeof(uint2) + 256;
}
bool cell_build_uses_buckets(int n, int key_bits, bool stable_order, int bucket_min_points, int* n_buckets)
{
	int hi_bits = 0, lo_bits = 0;
	if (stable_order || !bucket_plan(key_bits, n, bucket_min_points, hi_bits, lo_bits)) return false;
	if (n_buckets) *n_buckets = 1 << hi_bits;
	return true;
}
size_t bucket_window_slots(int n, int n_buckets) { return (size_t)n + (size_t)n / 8 + (size_t)n_buckets * 72 + 64; }   // >= the sum of bucket_window_cap over any counts that sum to n
int launch_cell_build(const float* xyz, const float* radii, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, const int* ids,
                      uint32_t* orig_sorted, const BuildGuard& gd, uint32_t query_limit, bool stable_order, int bucket_min_points, uint2* table, uint2* occ,
                      uint32_t* n_occ, int* passes_out, const BucketWindows& bw, const SparseCells& sp, hipStream_t s)
{
	int hi_bits = 0, lo_bits = 0;
	if (sp.blk) {
		// sparse grid: stable LSD passes, then the key-ordered list of occupied cells and its block index instead of a table
		const int res = launch_cell_sort(xyz, radii, n, g, key_bits, b, temp, ids, orig_sorted, gd, s);
		launch_sparse_cells(b.xyzi[res], n, g, sp.shift, sp.n_blocks, temp, occ, n_occ, sp.blk, s);
		if (passes_out) *passes_out = cell_sort_plan(key_bits).passes;
		return res;
	}
	if (stable_order || !bucket_plan(key_bits, n, bucket_min_points, hi_bits, lo_bits)) {
		const int res = launch_cell_sort(xyz, radii, n, g, key_bits, b, temp, ids, orig_sorted, gd, s);
		launch_cell_table(b.xyzi[res], n, g, table, occ, n_occ, s);
		if (passes_out) *passes_out = cell_sort_plan(key_bits).passes;
		return res;
	}
	if (passes_out) *passes_out = 2;
	const int ntiles = cs_num_tiles(n);
	const size_t hist_cap = ((size_t)1 << CS_MAX_BITS) * (size_t)ntiles;
	uint32_t* hist = (uint32_t*)temp;
	uint32_t* totals = (uint32_t*)((char*)temp + ((hist_cap * sizeof(uint32_t) + 255) / 256) * 256);
	uint32_t* strip_sums = totals + ((size_t)1 << CS_MAX_BITS);
	uint2* occ_tmp = (uint2*)((char*)temp + ((cell_sort_temp_bytes(n) + 255) / 256) * 256);
	uint2* bucket_info = occ_tmp + (size_t)n;
	const int n_buckets = 1 << hi_bits;
	// ---- pass A: bucket = high digit.  (radii and ids are picked up by original index in pass B)
	const uint32_t* counts = totals;
	int cstride = 1;
	const uint2* win = nullptr;
	if (bw.use && bw.win && bw.cursors) {
		// one read: windows from the previous run, one returning atomic per tile and bucket (k_bucket_scatter)
		hipLaunchKernelGGL(k_bucket_scatter, dim3((n + B1_TILE - 1) / B1_TILE), dim3(B1_THREADS), (size_t)n_buckets * sizeof(uint32_t), s, xyz, n, g, lo_bits, n_buckets, bw.win,
		                   bw.cursors, b.xyzi[1], radii, gd);
		counts = bw.cursors; cstride = BUCKET_CURSOR_STRIDE; win = bw.win;
	}
	else {
		BuildGuard gda = gd;
		TNSX_CS_DISPATCH(hi_bits, (cs_hist<B, false>(true, xyz, b.xyzi[0], n, g, lo_bits, hist, ntiles, radii, gda, s)));
		TNSX_CS_DISPATCH(hi_bits, cs_scan<B>(hist, ntiles, strip_sums, totals, s));
		TNSX_CS_DISPATCH(hi_bits, (cs_scatter<B, false>(true, false, xyz, nullptr, b.xyzi[0], b.r2[0], b.xyzi[1], b.r2[1], n, g, lo_bits, hist, totals, ntiles, nullptr,
		                                               nullptr, gda, s)));
	}
	// ---- pass B: one workgroup per bucket
	const bool variable = radii != nullptr, split = query_limit < (uint32_t)n;
	const size_t lds = ((size_t)1 << lo_bits) * sizeof(uint32_t) * (split ? 2 : 1);
#define TNSX_BS_GO(V, SP) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bucket_sort<V, SP>), dim3(n_buckets), dim3(BS_THREADS), lds, s, b.xyzi[1], b.xyzi[0], b.r2[0], radii, g, lo_bits, \
	                                         counts, cstride, win, table, occ_tmp, bucket_info, n_occ, ids, orig_sorted, query_limit, gd)
	if (lds > 48u * 1024u) {   // (a 13-bit low digit with two cursors per cell: above the default limit of dynamic LDS)
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_sort<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_sort<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	}
	if (variable) { if (split) TNSX_BS_GO(true, true); else TNSX_BS_GO(true, false); }
	else          { if (split) TNSX_BS_GO(false, true); else TNSX_BS_GO(false, false); }
#undef TNSX_BS_GO
	// (the windows of the next run: from this run's counts, whichever pass A produced them; written behind pass B, which reads this run's)
	hipLaunchKernelGGL(k_occ_reorder, dim3(n_buckets), dim3(BS_THREADS), 0, s, occ_tmp, occ, bucket_info, counts, cstride, bw.win);
	return 0;
}

__global__ void __launch_bounds__(256) k_table_clear(const uint2* __restrict__ occ, uint32_t n_occ, uint2* __restrict__ table)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n_occ) table[occ[i].y] = make_uint2(0u, 0u);
}
void launch_table_clear(const uint2* occ, uint32_t n_occ, uint2* table, hipStream_t s)
{
	if (n_occ == 0) return;
	hipLaunchKernelGGL(k_table_clear, dim3((n_occ + 255u) / 256u), dim3(256), 0, s, occ, n_occ, table);
}

}  // namespace tnsx
