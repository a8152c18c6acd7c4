// gfx950 (MI355X, CDNA4) kernels of the fixed-radius neighbour search engine.
//
// Everything here is sort / bin / compare-and-compact work bounded by HBM bandwidth and VALU issue; no MFMA.
// Wave = 64 lanes everywhere.  Compiled with -ffp-contract=off: the distance arithmetic is spelled with
// explicit round-to-nearest intrinsics so that the neighbour predicate is bit-identical to the reference
// (TreeNSearch.cpp:2478-2486 / BruteforceNSearch.cpp:88) in either arithmetic mode.
#include "tnsx_kernels.h"
#include "tnsx_device.h"

#include <algorithm>
#include <cfloat>

namespace tnsx {

// =====================================================================================================
// f64 -> f32 staging (TreeNSearch.cpp:277-296: `(float)` cast, round to nearest even)
// =====================================================================================================
__global__ void __launch_bounds__(256) k_f64_to_f32(const double* __restrict__ in, float* __restrict__ out, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = __double2float_rn(in[i]);
}
void launch_f64_to_f32(const double* in, float* out, size_t n, hipStream_t s)
{
	if (!n) return;
	const int blocks = (int)((n + 256 * 8 - 1) / (256 * 8) < 4096 ? (n + 256 * 8 - 1) / (256 * 8) : 4096);
	hipLaunchKernelGGL(k_f64_to_f32, dim3(blocks), dim3(256), 0, s, in, out, n);
}

// =====================================================================================================
// bounds: tight AABB (TreeNSearch.cpp:432-472) + min/max radius (TreeNSearch.cpp:304-313, :831-834)
// =====================================================================================================
static constexpr int BOUNDS_THREADS = 256;
static constexpr int BOUNDS_MAX_BLOCKS = 1024;
int bounds_num_blocks(int n)
{
	const int b = (n + BOUNDS_THREADS * 16 - 1) / (BOUNDS_THREADS * 16);
	return b < 1 ? 1 : (b > BOUNDS_MAX_BLOCKS ? BOUNDS_MAX_BLOCKS : b);
}
__device__ __forceinline__ float wave_min(float v)
{
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
	return v;
}
__device__ __forceinline__ float wave_max(float v)
{
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
	return v;
}
__device__ void block_reduce_write8(float (&v)[8], float* out)
{
	// v[0..2] min xyz, v[3..5] max xyz, v[6] min r, v[7] max r
	__shared__ float sh[BOUNDS_THREADS / WAVE][8];
	#pragma unroll
	for (int k = 0; k < 8; k++) v[k] = (k < 3 || k == 6) ? wave_min(v[k]) : wave_max(v[k]);
	const int w = threadIdx.x / WAVE;
	if (lane_id() == 0) {
		#pragma unroll
		for (int k = 0; k < 8; k++) sh[w][k] = v[k];
	}
	__syncthreads();
	if (threadIdx.x < 8) {
		const int k = threadIdx.x;
		float r = sh[0][k];
		for (int ww = 1; ww < BOUNDS_THREADS / WAVE; ww++) r = (k < 3 || k == 6) ? fminf(r, sh[ww][k]) : fmaxf(r, sh[ww][k]);
		out[k] = r;
	}
}
// VEC: the xyz array is 16-byte aligned and is read as float4 -- three of them hold four points: [x y z x | y z x y | z x y z]
template <bool VEC>
__global__ void __launch_bounds__(BOUNDS_THREADS) k_bounds_partial(const float* __restrict__ xyz, const float* __restrict__ radii, int n,
                                                                   float* __restrict__ partials)
{
	float v[8] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX };
	// (fminf / fmaxf drop a NaN operand.  A NaN x is NO POINT -- the rows of a ghost message past the real count -- and then its y, z and radius,
	//  whatever they hold, count for nothing either: a stale radius there once made the cells five times too wide)
	auto take = [&](float x, float y, float z) {
		const bool pt = x == x;
		y = pt ? y : x; z = pt ? z : x;
		v[0] = fminf(v[0], x); v[1] = fminf(v[1], y); v[2] = fminf(v[2], z);
		v[3] = fmaxf(v[3], x); v[4] = fmaxf(v[4], y); v[5] = fmaxf(v[5], z);
	};
	const int stride = gridDim.x * BOUNDS_THREADS, t0 = blockIdx.x * BOUNDS_THREADS + threadIdx.x;
	int first_scalar = 0;
	if (VEC) {
		const int n4 = n / 4;   // groups of four points = three float4
		const float4* q = reinterpret_cast<const float4*>(xyz);
		#pragma unroll 2
		for (int gi = t0; gi < n4; gi += stride) {
			const float4 a = q[3 * (size_t)gi], b = q[3 * (size_t)gi + 1], c = q[3 * (size_t)gi + 2];
			take(a.x, a.y, a.z); take(a.w, b.x, b.y); take(b.z, b.w, c.x); take(c.y, c.z, c.w);
		}
		first_scalar = n4 * 4;
	}
	for (int i = first_scalar + t0; i < n; i += stride) take(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
	if (radii) {
		for (int i = t0; i < n; i += stride) { const float x = xyz[3 * (size_t)i]; const float r = x == x ? radii[i] : x; v[6] = fminf(v[6], r); v[7] = fmaxf(v[7], r); }
	}
	block_reduce_write8(v, partials + 8 * (size_t)blockIdx.x);
}
__global__ void __launch_bounds__(BOUNDS_THREADS) k_bounds_final(const float* __restrict__ partials, int n_partials, float* __restrict__ out8)
{
	float v[8] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX };
	for (int i = threadIdx.x; i < n_partials; i += BOUNDS_THREADS) {
		#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = (k < 3 || k == 6) ? fminf(v[k], partials[8 * (size_t)i + k]) : fmaxf(v[k], partials[8 * (size_t)i + k]);
	}
	block_reduce_write8(v, out8);
}
void launch_bounds_partial(const float* xyz, const float* radii, int n, float* partials, hipStream_t s)
{
	if (((uintptr_t)xyz & 15u) == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bounds_partial<true>), dim3(bounds_num_blocks(n)), dim3(BOUNDS_THREADS), 0, s, xyz, radii, n, partials);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bounds_partial<false>), dim3(bounds_num_blocks(n)), dim3(BOUNDS_THREADS), 0, s, xyz, radii, n, partials);
}
void launch_bounds_final(const float* partials, int n_partials, float* out8, hipStream_t s)
{
	hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(BOUNDS_THREADS), 0, s, partials, n_partials, out8);
}

// =====================================================================================================
// exclusive scan (reduce -> spine -> apply), tiles of 4096 elements, 16-byte vector loads
// =====================================================================================================
static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_TILE = 4096;   // 4 sub-tiles of 1024 (uint4 per thread)

size_t scan_temp_bytes(size_t n) { return ((n + SCAN_TILE - 1) / SCAN_TILE + 2) * sizeof(uint64_t); }

__device__ __forceinline__ uint4 load4_guarded(const uint32_t* in, size_t e, size_t n)
{
	if (e + 3 < n) return *reinterpret_cast<const uint4*>(in + e);
	uint4 v = { 0, 0, 0, 0 };
	if (e < n) v.x = in[e];
	if (e + 1 < n) v.y = in[e + 1];
	if (e + 2 < n) v.z = in[e + 2];
	return v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const uint32_t* __restrict__ in, size_t n, uint64_t* __restrict__ sums)
{
	__shared__ uint64_t sh[SCAN_THREADS / WAVE];
	const size_t base = (size_t)blockIdx.x * SCAN_TILE;
	uint64_t acc = 0;
	#pragma unroll
	for (int it = 0; it < 4; it++) {
		const uint4 v = load4_guarded(in, base + (size_t)it * 1024 + threadIdx.x * 4, n);
		acc += (uint64_t)v.x + v.y + v.z + v.w;
	}
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
	if (lane_id() == 0) sh[threadIdx.x / WAVE] = acc;
	__syncthreads();
	if (threadIdx.x == 0) sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// single block: exclusive scan of sums[nb] in place, total -> sums[nb]
__global__ void __launch_bounds__(1024) k_scan_spine(uint64_t* __restrict__ sums, int nb)
{
	__shared__ uint64_t wsum[16];
	__shared__ uint64_t carry_sh;
	if (threadIdx.x == 0) carry_sh = 0;
	__syncthreads();
	for (int base = 0; base < nb; base += 1024) {
		const int i = base + threadIdx.x;
		const uint64_t v = i < nb ? sums[i] : 0;
		uint64_t inc = v;
		#pragma unroll
		for (int o = 1; o < WAVE; o <<= 1) { const uint64_t t = __shfl_up(inc, o, WAVE); if (lane_id() >= o) inc += t; }
		if (lane_id() == WAVE - 1) wsum[threadIdx.x / WAVE] = inc;
		__syncthreads();
		uint64_t woff = 0;
		for (int w = 0; w < (int)(threadIdx.x / WAVE); w++) woff += wsum[w];
		const uint64_t carry = carry_sh;
		if (i < nb) sums[i] = carry + woff + inc - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry_sh = carry + woff + inc;
		__syncthreads();
	}
	if (threadIdx.x == 0) sums[nb] = carry_sh;
}

template <typename TOut>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(const uint32_t* __restrict__ in, TOut* __restrict__ out, size_t n,
                                                            const uint64_t* __restrict__ sums, int nb, int write_total)
{
	__shared__ uint64_t wsum[SCAN_THREADS / WAVE];
	const size_t base = (size_t)blockIdx.x * SCAN_TILE;
	uint64_t carry = sums[blockIdx.x];
	#pragma unroll 1
	for (int it = 0; it < 4; it++) {
		const size_t e = base + (size_t)it * 1024 + threadIdx.x * 4;
		const uint4 v = load4_guarded(in, e, n);
		const uint64_t t = (uint64_t)v.x + v.y + v.z + v.w;
		uint64_t inc = t;
		#pragma unroll
		for (int o = 1; o < WAVE; o <<= 1) { const uint64_t u = __shfl_up(inc, o, WAVE); if (lane_id() >= o) inc += u; }
		__syncthreads();   // wsum reuse
		if (lane_id() == WAVE - 1) wsum[threadIdx.x / WAVE] = inc;
		__syncthreads();
		uint64_t woff = 0, tile_total = 0;
		#pragma unroll
		for (int w = 0; w < SCAN_THREADS / WAVE; w++) { if (w < (int)(threadIdx.x / WAVE)) woff += wsum[w]; tile_total += wsum[w]; }
		const uint64_t ex = carry + woff + inc - t;
		if (e < n) out[e] = (TOut)ex;
		if (e + 1 < n) out[e + 1] = (TOut)(ex + v.x);
		if (e + 2 < n) out[e + 2] = (TOut)(ex + v.x + v.y);
		if (e + 3 < n) out[e + 3] = (TOut)(ex + v.x + v.y + v.z);
		carry += tile_total;
	}
	if (write_total && blockIdx.x == 0 && threadIdx.x == 0) out[n] = (TOut)sums[nb];
}

template <typename TOut>
static void scan_impl(const uint32_t* in, TOut* out, size_t n, void* temp, hipStream_t s, int write_total)
{
	uint64_t* sums = (uint64_t*)temp;
	if (n == 0) {
		if (write_total) (void)hipMemsetAsync(out, 0, sizeof(TOut), s);
		return;
	}
	const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
	hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, s, in, n, sums);
	hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, sums, nb);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<TOut>), dim3(nb), dim3(SCAN_THREADS), 0, s, in, out, n, sums, nb, write_total);
}
void exclusive_scan_u32_to_u64(const uint32_t* in, uint64_t* out, size_t n, void* temp, hipStream_t s) { scan_impl<uint64_t>(in, out, n, temp, s, 1); }

// =====================================================================================================
// Gap-free copy of a pair's records in POINT order (round 5: the host mirror).  The record pool has holes (the unused ends of the waves' slabs: 10 % at C2) and
// its records lie in the order the cells were served; the link to the host is what bounds the drop-in mode, so what crosses it is compacted first:
// len[p] = count + 1 -> exclusive scan = the mirror's offsets -> every record copied to its place.  A wave owns 64 consecutive points; it walks their records
// four at a time (the loads of four records in flight), 64 ints per lane round.
// =====================================================================================================
// skip = 0: whole records, `[count, j...]` (the host mirror);  skip = 1: the indices only (a standard CSR: tnsx_pair_csr_device)
__global__ void __launch_bounds__(256) k_record_lengths(const int* __restrict__ records, const uint64_t* __restrict__ offs, int n, uint32_t* __restrict__ len, uint32_t skip)
{
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p < n) len[p] = (uint32_t)records[offs[p]] + 1u - skip;
}
__global__ void __launch_bounds__(256) k_compact_records(const int* __restrict__ records, const uint64_t* __restrict__ offs, const uint64_t* __restrict__ new_offs, int n,
                                                         int* __restrict__ out, uint32_t skip)
{
	const int lane = lane_id();
	const size_t wave = (size_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
	const size_t p0 = wave * WAVE;
	if (p0 >= (size_t)n) return;
	const size_t p = p0 + (size_t)lane < (size_t)n ? p0 + (size_t)lane : (size_t)n - 1;
	const uint64_t src = offs[p] + skip, dst = new_offs[p];
	const uint32_t len = (uint32_t)(new_offs[p + 1] - dst);
	const int cnt = (int)((size_t)n - p0 < (size_t)WAVE ? (size_t)n - p0 : (size_t)WAVE);
	for (int t0 = 0; t0 < cnt; t0 += 4) {
		int v[4];
		uint64_t d[4];
		uint32_t l[4];
		#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int t = t0 + u < cnt ? t0 + u : cnt - 1;
			const uint64_t s_t = ((uint64_t)readlane_u32((uint32_t)(src >> 32), t) << 32) | readlane_u32((uint32_t)src, t);
			d[u] = ((uint64_t)readlane_u32((uint32_t)(dst >> 32), t) << 32) | readlane_u32((uint32_t)dst, t);
			l[u] = t0 + u < cnt ? readlane_u32(len, t) : 0u;
			v[u] = (uint32_t)lane < l[u] ? records[s_t + (uint32_t)lane] : 0;
		}
		#pragma unroll
		for (int u = 0; u < 4; u++) {
			if ((uint32_t)lane < l[u]) out[d[u] + (uint32_t)lane] = v[u];
			if (l[u] > (uint32_t)WAVE) {   // (a record longer than 64 ints: the rest, 64 at a time)
				const int t = t0 + u;
				const uint64_t s_t = ((uint64_t)readlane_u32((uint32_t)(src >> 32), t) << 32) | readlane_u32((uint32_t)src, t);
				for (uint32_t k = (uint32_t)WAVE + (uint32_t)lane; k < l[u]; k += (uint32_t)WAVE) out[d[u] + k] = records[s_t + k];
			}
		}
	}
}
void launch_record_lengths(const int* records, const uint64_t* offs, int n, uint32_t* len, bool indices_only, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(k_record_lengths, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, records, offs, n, len, indices_only ? 1u : 0u);
}
void launch_compact_records(const int* records, const uint64_t* offs, const uint64_t* new_offs, int n, int* out, bool indices_only, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(k_compact_records, dim3((unsigned)(((size_t)n + 255) / 256)), dim3(256), 0, s, records, offs, new_offs, n, out, indices_only ? 1u : 0u);
}

// =====================================================================================================
// permutation of fixed-size byte records (device-side apply_zsort, TreeNSearch.h:465-480)
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) k_permute_t(const T* __restrict__ in, T* __restrict__ out, const int* __restrict__ perm, int n, int words)
{
	const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t total = (size_t)n * words;
	if (g >= total) return;
	const size_t rec = g / words, wd = g % words;
	out[g] = in[(size_t)perm[rec] * words + wd];
}
// records of 1, 2, 3 or 4 words (radii, ids, xyz, float4 -- what apply_zsort is called with in an SPH step): a RECORD per thread, one read of the permutation and one
// (wide) load / store per record, four records in flight per thread.  (round 5: the word-per-thread kernel above read perm[] once per word and kept one load in flight)
template <int WORDS>
__global__ void __launch_bounds__(256) k_permute_rec(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const int* __restrict__ perm, int n)
{
	struct Rec { uint32_t w[WORDS]; };
	constexpr int PER = 4;
	const size_t base = (size_t)blockIdx.x * (256 * PER) + threadIdx.x;
	int src[PER];
	#pragma unroll
	for (int k = 0; k < PER; k++) { const size_t r = base + (size_t)k * 256; src[k] = perm[r < (size_t)n ? r : (size_t)n - 1]; }
	Rec v[PER];
	#pragma unroll
	for (int k = 0; k < PER; k++) v[k] = reinterpret_cast<const Rec*>(in)[src[k]];
	#pragma unroll
	for (int k = 0; k < PER; k++) { const size_t r = base + (size_t)k * 256; if (r < (size_t)n) reinterpret_cast<Rec*>(out)[r] = v[k]; }
}
void launch_permute_bytes(const void* in, void* out, const int* new_to_old, int n, size_t rec_bytes, hipStream_t s)
{
	if (n <= 0 || rec_bytes == 0) return;
	const bool aligned4 = (rec_bytes % 4 == 0) && (((uintptr_t)in | (uintptr_t)out) % 4 == 0);
	if (aligned4 && rec_bytes <= 16) {
		const dim3 grid((unsigned)(((size_t)n + 1023) / 1024));
		const uint32_t* i4 = (const uint32_t*)in; uint32_t* o4 = (uint32_t*)out;
		switch (rec_bytes / 4) {
		case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_permute_rec<1>), grid, dim3(256), 0, s, i4, o4, new_to_old, n); break;
		case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_permute_rec<2>), grid, dim3(256), 0, s, i4, o4, new_to_old, n); break;
		case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_permute_rec<3>), grid, dim3(256), 0, s, i4, o4, new_to_old, n); break;
		default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_permute_rec<4>), grid, dim3(256), 0, s, i4, o4, new_to_old, n); break;
		}
	}
	else if (aligned4) {
		const int words = (int)(rec_bytes / 4);
		const size_t total = (size_t)n * words;
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_permute_t<uint32_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const uint32_t*)in, (uint32_t*)out,
		                   new_to_old, n, words);
	}
	else {
		const int words = (int)rec_bytes;
		const size_t total = (size_t)n * words;
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_permute_t<unsigned char>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const unsigned char*)in,
		                   (unsigned char*)out, new_to_old, n, words);
	}
}

// =====================================================================================================
// ghost-halo selection (multi-GPU slabs): one pass over x, wave-aggregated append to the two send buffers
// =====================================================================================================
// One workgroup per tile of 4096 points and ONE atomic per workgroup and side (a single counter takes ~88 atomics per
// microsecond; per-wave appends would cost 1.8 ms at 10 M points).
static constexpr int HP_ITEMS = 16;
template <bool WITH_R>
__global__ void __launch_bounds__(256) k_halo_pack(const float* __restrict__ xyz, const float* __restrict__ radii, const long long* __restrict__ gids, int n,
                                                   float left_cut, float right_cut, float* __restrict__ out_left, float* __restrict__ out_right,
                                                   unsigned long long cap_left, unsigned long long cap_right, unsigned int* __restrict__ counts)
{
	constexpr int COLS = WITH_R ? 6 : 5;
	__shared__ uint32_t wcnt[2][HP_ITEMS * 4];   // [side][round * 4 + wave] -> exclusive prefix inside the tile
	__shared__ uint32_t bbase[2];
	const int w = threadIdx.x / WAVE;
	const size_t base = (size_t)blockIdx.x * (256 * HP_ITEMS);
	uint32_t flags = 0;   // bit 2i: point i goes left, bit 2i+1: right
	#pragma unroll
	for (int i = 0; i < HP_ITEMS; i++) {
		const size_t p = base + (size_t)i * 256 + threadIdx.x;
		const float x = p < (size_t)n ? xyz[3 * p] : 0.0f;
		const bool tl = out_left != nullptr && p < (size_t)n && x < left_cut;
		const bool tr = out_right != nullptr && p < (size_t)n && x >= right_cut;
		flags |= (tl ? 1u : 0u) << (2 * i) | (tr ? 2u : 0u) << (2 * i);
		const uint64_t ml = __ballot(tl), mr = __ballot(tr);
		if (lane_id() == 0) { wcnt[0][i * 4 + w] = (uint32_t)__popcll(ml); wcnt[1][i * 4 + w] = (uint32_t)__popcll(mr); }
	}
	__syncthreads();
	if (threadIdx.x < 2) {
		const int side = threadIdx.x;
		uint32_t s = 0;
		for (int q = 0; q < HP_ITEMS * 4; q++) { const uint32_t t = wcnt[side][q]; wcnt[side][q] = s; s += t; }
		bbase[side] = s ? atomicAdd(counts + side, s) : 0u;
	}
	__syncthreads();
	#pragma unroll
	for (int i = 0; i < HP_ITEMS; i++) {
		const size_t p = base + (size_t)i * 256 + threadIdx.x;
		#pragma unroll
		for (int side = 0; side < 2; side++) {
			const bool take = (flags >> (2 * i + side)) & 1u;
			const uint64_t m = __ballot(take);
			const unsigned long long row = (unsigned long long)bbase[side] + wcnt[side][i * 4 + w] + mbcnt64(m);
			if (take && row < (side == 0 ? cap_left : cap_right)) {
				float* o = (side == 0 ? out_left : out_right) + row * COLS;
				o[0] = xyz[3 * p]; o[1] = xyz[3 * p + 1]; o[2] = xyz[3 * p + 2];
				if (WITH_R) o[3] = radii[p];
				const long long g = gids[p];
				o[COLS - 2] = __uint_as_float((uint32_t)g);
				o[COLS - 1] = __uint_as_float((uint32_t)((unsigned long long)g >> 32));
			}
		}
	}
}
void launch_halo_pack(const float* xyz, const float* radii, const long long* gids, int n, float left_cut, float right_cut, float* out_left,
                      float* out_right, unsigned long long cap_left, unsigned long long cap_right, unsigned int* counts, hipStream_t s)
{
	if (n <= 0 || (!out_left && !out_right)) return;
	const dim3 grid((n + 256 * HP_ITEMS - 1) / (256 * HP_ITEMS)), block(256);
	if (radii) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_halo_pack<true>), grid, block, 0, s, xyz, radii, gids, n, left_cut, right_cut, out_left, out_right, cap_left, cap_right, counts);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_halo_pack<false>), grid, block, 0, s, xyz, radii, gids, n, left_cut, right_cut, out_left, out_right, cap_left, cap_right, counts);
}

// =====================================================================================================
// slab layer: received halo rows -> the [owned | ghosts] arrays of a set; 64-bit ids -> the 32-bit indices of the lists; checks
// =====================================================================================================
__global__ void __launch_bounds__(256) k_slab_unpack(const float* __restrict__ rows, uint32_t n_rows, const unsigned int* __restrict__ count, int W,
                                                     float* __restrict__ xyz, float* __restrict__ radii, int* __restrict__ ids, unsigned int* __restrict__ flag)
{
	const uint32_t have = count ? *count : n_rows;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_rows; i += (size_t)gridDim.x * 256) {
		const float* r = rows + i * (size_t)W;
		const bool valid = i < (size_t)have;
		xyz[3 * i] = valid ? r[0] : __uint_as_float(0x7fc00000u);     // x = NaN: no point
		xyz[3 * i + 1] = valid ? r[1] : 0.0f;
		xyz[3 * i + 2] = valid ? r[2] : 0.0f;
		if (radii) radii[i] = valid ? r[3] : 0.0f;
		const uint32_t lo = __float_as_uint(r[W - 2]), hi = __float_as_uint(r[W - 1]);
		ids[i] = valid ? (int)lo : -1;
		if (valid && (hi != 0u || lo > 0x7fffffffu)) *flag = 1u;
	}
}
__global__ void __launch_bounds__(256) k_slab_ids(const long long* __restrict__ gids, int n, int* __restrict__ ids, unsigned int* __restrict__ flag)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) {
		const long long g = gids[i];
		ids[i] = (int)g;
		if (g < 0 || g > 0x7fffffffll) *flag = 1u;
	}
}
__global__ void __launch_bounds__(256) k_slab_flag_gt(const float* __restrict__ v, int n, float limit, unsigned int* __restrict__ flag)
{
	bool any = false;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) any = any || v[i] > limit;
	if (__ballot(any) != 0ull && lane_id() == 0) *flag = 1u;
}
__global__ void __launch_bounds__(256) k_slab_x_range(const float* __restrict__ xyz, int n, float* __restrict__ minmax)
{
	float lo = FLT_MAX, hi = -FLT_MAX;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) {
		const float x = xyz[3 * i];
		if (x == x) { lo = fminf(lo, x); hi = fmaxf(hi, x); }
	}
	for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, WAVE)); hi = fmaxf(hi, __shfl_xor(hi, o, WAVE)); }
	if (lane_id() == 0) {
		// float min / max through integer atomics on the ordered encoding (sign-magnitude -> two's complement order)
		auto enc = [](float f) { const int b = __float_as_int(f); return b >= 0 ? b : (int)(0x80000000u - (uint32_t)b); };
		auto dec = [](int e) { return __int_as_float(e >= 0 ? e : (int)(0x80000000u - (uint32_t)e)); };
		int* mm = reinterpret_cast<int*>(minmax);
		int old = mm[0];
		while (enc(lo) < enc(__int_as_float(old))) { const int seen = atomicCAS(mm, old, __float_as_int(lo)); if (seen == old) break; old = seen; }
		old = mm[1];
		while (enc(hi) > enc(__int_as_float(old))) { const int seen = atomicCAS(mm + 1, old, __float_as_int(hi)); if (seen == old) break; old = seen; }
		(void)dec;
	}
}
void launch_slab_unpack(const float* rows, uint32_t n_rows, const unsigned int* count, int W, float* xyz, float* radii, int* ids, unsigned int* flag, hipStream_t s)
{
	if (!n_rows) return;
	const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n_rows + 255) / 256, 4096);
	hipLaunchKernelGGL(k_slab_unpack, dim3(blocks), dim3(256), 0, s, rows, n_rows, count, W, xyz, radii, ids, flag);
}
void launch_slab_ids(const long long* gids, int n, int* ids, unsigned int* flag, hipStream_t s)
{
	if (n <= 0) return;
	const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n + 1023) / 1024, 4096);
	hipLaunchKernelGGL(k_slab_ids, dim3(blocks), dim3(256), 0, s, gids, n, ids, flag);
}
void launch_slab_flag_gt(const float* v, int n, float limit, unsigned int* flag, hipStream_t s)
{
	if (n <= 0) return;
	const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n + 2047) / 2048, 2048);
	hipLaunchKernelGGL(k_slab_flag_gt, dim3(blocks), dim3(256), 0, s, v, n, limit, flag);
}
void launch_slab_x_range(const float* xyz, int n, float* minmax, hipStream_t s)
{
	if (n <= 0) return;
	const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n + 4095) / 4096, 1024);
	hipLaunchKernelGGL(k_slab_x_range, dim3(blocks), dim3(256), 0, s, xyz, n, minmax);
}

// =====================================================================================================
// slab layer, redistribution (the one all-to-all of a decomposition: every point goes to the slab that owns its x).
//   k_slab_dest_rows<COUNT = true>   counts[d] += points with cuts[d] <= x < cuts[d + 1] (LDS-aggregated; NaN x: no point, goes nowhere)
//   k_slab_dest_rows<COUNT = false>  the same points as rows [x, y, z, (r,) gid_lo, gid_hi] at rows[(first[d] + k) * W], k from the cursor of d
//   k_slab_rows_to_points            rows -> xyz / radii / 64-bit ids
// The slab of a point is found by counting the interior cuts at or below x (world <= 64: a handful of compares on wave-uniform values).
// =====================================================================================================
static constexpr int RD_MAX_WORLD = 64;
struct SlabCuts { float c[RD_MAX_WORLD + 1]; int world; };
template <bool COUNT>
__global__ void __launch_bounds__(256) k_slab_dest_rows(const float* __restrict__ xyz, const float* __restrict__ radii, const long long* __restrict__ gids, int n, SlabCuts cuts,
                                                        unsigned int* __restrict__ counters, const unsigned int* __restrict__ first, float* __restrict__ rows, int W)
{
	__shared__ unsigned int h[RD_MAX_WORLD], base[RD_MAX_WORLD];
	if (threadIdx.x < RD_MAX_WORLD) h[threadIdx.x] = 0u;
	__syncthreads();
	constexpr int ITEMS = 8;
	int dest[ITEMS];
	unsigned int rank[ITEMS];
	const size_t b0 = (size_t)blockIdx.x * (256 * ITEMS);
	#pragma unroll
	for (int i = 0; i < ITEMS; i++) {
		const size_t p = b0 + (size_t)i * 256 + threadIdx.x;
		dest[i] = -1; rank[i] = 0;
		if (p < (size_t)n) {
			const float x = xyz[3 * p];
			if (x == x) {
				int d = 0;
				for (int k = 1; k < cuts.world; k++) d += x >= cuts.c[k] ? 1 : 0;   // cuts ascending: the number of interior cuts at or below x
				dest[i] = d;
				rank[i] = atomicAdd(&h[d], 1u);
			}
		}
	}
	__syncthreads();
	if ((int)threadIdx.x < cuts.world) { const unsigned int c = h[threadIdx.x]; base[threadIdx.x] = c ? atomicAdd(counters + threadIdx.x, c) : 0u; }
	if (COUNT) return;
	__syncthreads();
	#pragma unroll
	for (int i = 0; i < ITEMS; i++) {
		if (dest[i] < 0) continue;
		const size_t p = b0 + (size_t)i * 256 + threadIdx.x;
		float* r = rows + ((size_t)first[dest[i]] + base[dest[i]] + rank[i]) * (size_t)W;
		r[0] = xyz[3 * p]; r[1] = xyz[3 * p + 1]; r[2] = xyz[3 * p + 2];
		if (radii) r[3] = radii[p];
		const unsigned long long g = (unsigned long long)gids[p];
		r[W - 2] = __uint_as_float((uint32_t)g); r[W - 1] = __uint_as_float((uint32_t)(g >> 32));
	}
}
void launch_slab_dest_rows(bool count_only, const float* xyz, const float* radii, const long long* gids, int n, const float* cuts, int world, unsigned int* counters,
                           const unsigned int* first, float* rows, int W, hipStream_t s)
{
	if (n <= 0) return;
	SlabCuts c;
	c.world = world;
	for (int k = 0; k <= world && k <= RD_MAX_WORLD; k++) c.c[k] = cuts[k];
	const unsigned blocks = (unsigned)(((size_t)n + 2047) / 2048);
	if (count_only) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slab_dest_rows<true>), dim3(blocks), dim3(256), 0, s, xyz, radii, gids, n, c, counters, first, rows, W);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slab_dest_rows<false>), dim3(blocks), dim3(256), 0, s, xyz, radii, gids, n, c, counters, first, rows, W);
}
__global__ void __launch_bounds__(256) k_slab_rows_to_points(const float* __restrict__ rows, size_t n_rows, int W, float* __restrict__ xyz, float* __restrict__ radii,
                                                             long long* __restrict__ gids)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (size_t)gridDim.x * 256) {
		const float* r = rows + i * (size_t)W;
		xyz[3 * i] = r[0]; xyz[3 * i + 1] = r[1]; xyz[3 * i + 2] = r[2];
		if (radii) radii[i] = r[3];
		gids[i] = (long long)(((unsigned long long)__float_as_uint(r[W - 1]) << 32) | __float_as_uint(r[W - 2]));
	}
}
void launch_slab_rows_to_points(const float* rows, size_t n_rows, int W, float* xyz, float* radii, long long* gids, hipStream_t s)
{
	if (!n_rows) return;
	const unsigned blocks = (unsigned)std::min<size_t>((n_rows + 255) / 256, 8192);
	hipLaunchKernelGGL(k_slab_rows_to_points, dim3(blocks), dim3(256), 0, s, rows, n_rows, W, xyz, radii, gids);
}

// =====================================================================================================
// x-plane histogram (multi-GPU slabs: balanced cuts).  hist[b] += points with clamp(trunc((x - x0) * inv_dx), 0, n_bins - 1) == b.
// Every block keeps a private copy of the bins in LDS (n_bins <= XH_LDS_BINS) and adds its non-zero bins to the global
// histogram at the end; wider histograms go straight to global atomics.
// =====================================================================================================
static constexpr int XH_LDS_BINS = 8192;
template <bool LDS>
__global__ void __launch_bounds__(256) k_x_histogram(const float* __restrict__ xyz, int n, float x0, float inv_dx, int n_bins, unsigned int* __restrict__ hist)
{
	__shared__ unsigned int h[LDS ? XH_LDS_BINS : 1];
	if (LDS) {
		for (int b = threadIdx.x; b < n_bins; b += 256) h[b] = 0u;
		__syncthreads();
	}
	for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)n; p += (size_t)gridDim.x * 256) {
		const float x = xyz[3 * p];
		if (x != x) continue;                                   // NaN: no point
		int b = (int)__fmul_rn(__fsub_rn(x, x0), inv_dx);
		b = b < 0 ? 0 : (b > n_bins - 1 ? n_bins - 1 : b);
		atomicAdd(LDS ? &h[b] : &hist[b], 1u);
	}
	if (LDS) {
		__syncthreads();
		for (int b = threadIdx.x; b < n_bins; b += 256) { const unsigned int v = h[b]; if (v) atomicAdd(&hist[b], v); }
	}
}
void launch_x_histogram(const float* xyz, int n, float x0, float inv_dx, int n_bins, unsigned int* hist, hipStream_t s)
{
	if (n <= 0 || n_bins <= 0) return;
	int blocks = (n + 256 * 16 - 1) / (256 * 16);
	blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
	if (n_bins <= XH_LDS_BINS) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_x_histogram<true>), dim3(blocks), dim3(256), 0, s, xyz, n, x0, inv_dx, n_bins, hist);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_x_histogram<false>), dim3(blocks), dim3(256), 0, s, xyz, n, x0, inv_dx, n_bins, hist);
}

// =====================================================================================================
// neighbour-id translation: every list entry j of the first n_query records becomes id_map[j] (in place).  One wave per
// record at a time (a record is ~60 consecutive ints: one coalesced read-modify-write), waves grid-stride over the points.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_translate_records(int* __restrict__ records, const uint64_t* __restrict__ offs_by_orig, int n_query,
                                                          const int* __restrict__ id_map)
{
	const int lane = lane_id();
	const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) / WAVE, n_waves = (size_t)gridDim.x * (256 / WAVE);
	for (size_t p = wave; p < (size_t)n_query; p += n_waves) {
		int* rec = records + offs_by_orig[p];
		const int cnt = rec[0];
		for (int k = lane; k < cnt; k += WAVE) rec[1 + k] = id_map[rec[1 + k]];
	}
}
void launch_translate_records(int* records, const uint64_t* offs_by_orig, int n_query, const int* id_map, int n_cus, hipStream_t s)
{
	if (n_query <= 0) return;
	long long blocks = ((long long)n_query + 3) / 4;
	const long long cap = (long long)n_cus * 32;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL(k_translate_records, dim3((unsigned)blocks), dim3(256), 0, s, records, offs_by_orig, n_query, id_map);
}

// =====================================================================================================
// start of a pool pass: the region table of the pass (first int and capacity of every pool region, tnsx_query.hip PoolState) goes to
// the device words the query kernels read; for a pair of two DIFFERENT sets every offset is also pointed at the shared empty record
// (int 0 of the pool, count 0) -- one launch instead of a copy and three memsets
// =====================================================================================================
struct PoolRegionTable { unsigned long long v[2 * (POOL_REGIONS + 1)]; };
__device__ __forceinline__ void pool_begin_block(const unsigned long long* __restrict__ regions, uint32_t* __restrict__ ctrl)
{
	// the control block of the pass: the hot words of every slot (cursor + neighbour / waste counters: 18 64-bit words; ticket counters and
	// worklist lengths: the first word) start at zero -- 141 x 144 bytes instead of a memset of the whole 600 KB block
	if (threadIdx.x < CTRL_SLOTS) {
		uint32_t* slot = ctrl + (size_t)threadIdx.x * CTRL_STRIDE_U32;
		for (int w = 0; w < 2 * POOL_CTRL_WORDS; w++) slot[w] = 0u;
	}
	__syncthreads();
	unsigned long long* table = reinterpret_cast<unsigned long long*>(ctrl + (size_t)CTRL_REGIONS * CTRL_STRIDE_U32);
	if (threadIdx.x < 2 * (POOL_REGIONS + 1)) table[threadIdx.x] = regions[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_pool_begin(PoolRegionTable t, uint32_t* __restrict__ ctrl, uint64_t* __restrict__ offs, size_t n, int* __restrict__ records)
{
	if (blockIdx.x == 0) pool_begin_block(t.v, ctrl);
	if (blockIdx.x == 0 && threadIdx.x == 0) records[0] = 0;   // the empty record of the pool
	if (n == 0) return;
	ulonglong2* o2 = reinterpret_cast<ulonglong2*>(offs);
	const size_t n2 = n / 2;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) o2[i] = make_ulonglong2(0ull, 0ull);
	if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) offs[n - 1] = 0ull;
}
// query points that entered no cell (x is NaN: no point) were never visited and have no offset: point them at the pool's empty record (int 0)
__global__ void __launch_bounds__(256) k_point_nan_offsets(const float* __restrict__ xyz, int n, uint64_t* __restrict__ offs)
{
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p < n) { const float x = xyz[3 * (size_t)p]; if (x != x) offs[p] = 0ull; }
}
void launch_point_nan_offsets(const float* xyz, int n, uint64_t* offs, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(k_point_nan_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xyz, n, offs);
}
// The same in the exact (two-pass) layout.  There the NaN-x points sit behind all cells of the SORTED array (the stable sort keys them behind everything) and no
// query kernel visits them: phase 0 (before the count pass) gives each of them that is a query a record of one int -- its count word -- in the scan, phase 1 (after
// the fill pass) writes that count word (0) and the offset.
__global__ void __launch_bounds__(256) k_exact_nan(const float4* __restrict__ xyzi, const uint32_t* __restrict__ orig_sorted, int n, uint32_t query_limit, uint32_t* __restrict__ counts,
                                                   const uint64_t* __restrict__ offs_sorted, int* __restrict__ records, uint64_t* __restrict__ offs_by_orig, int phase)
{
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n) return;
	const float x = xyzi[p].x;
	if (x == x) return;
	const uint32_t orig = orig_sorted ? orig_sorted[p] : __float_as_uint(xyzi[p].w);
	const bool is_query = orig < query_limit;
	if (phase == 0) counts[p] = is_query ? 1u : 0u;
	else if (is_query) { const uint64_t o = offs_sorted[p]; records[o] = 0; offs_by_orig[orig] = o; }
}
void launch_exact_nan(const float4* xyzi, const uint32_t* orig_sorted, int n, uint32_t query_limit, uint32_t* counts, const uint64_t* offs_sorted, int* records,
                      uint64_t* offs_by_orig, int phase, hipStream_t s)
{
	if (n > 0) hipLaunchKernelGGL(k_exact_nan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xyzi, orig_sorted, n, query_limit, counts, offs_sorted, records, offs_by_orig, phase);
}
// every stride-th entry of a cell list (the middle one of each group) -> dst, *n_dst = their number: the worklist of a sampled count-only pass
__global__ void __launch_bounds__(256) k_sample_cells(const uint2* __restrict__ src, const uint32_t* __restrict__ n_src_p, uint32_t stride, uint2* __restrict__ dst, uint32_t* __restrict__ n_dst)
{
	const uint32_t n_src = *n_src_p, n = (n_src + stride - 1u) / stride;
	if (blockIdx.x == 0 && threadIdx.x == 0) *n_dst = n;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const uint32_t j = i * stride + stride / 2u;
		dst[i] = src[j < n_src ? j : n_src - 1u];
	}
}
void launch_sample_cells(const uint2* src, const uint32_t* n_src, uint32_t stride, uint2* dst, uint32_t* n_dst, size_t max_dst, hipStream_t s)
{
	size_t blocks = (max_dst + 255) / 256;
	blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
	hipLaunchKernelGGL(k_sample_cells, dim3((unsigned)blocks), dim3(256), 0, s, src, n_src, stride, dst, n_dst);
}
void launch_pool_begin(const unsigned long long* regions, uint32_t* ctrl, uint64_t* offs, size_t n_shared_empty, int* records, hipStream_t s)
{
	static_assert(CTRL_SLOTS <= 256, "one thread per slot");
	PoolRegionTable t;
	for (int k = 0; k < 2 * (POOL_REGIONS + 1); k++) t.v[k] = regions[k];
	size_t blocks = (n_shared_empty / 2 + 256 * 8 - 1) / (256 * 8);
	blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
	hipLaunchKernelGGL(k_pool_begin, dim3((unsigned)blocks), dim3(256), 0, s, t, ctrl, offs, n_shared_empty, records);
}

// end of a run: everything the host needs to judge it -- per pool pass the cursor / neighbour / waste words of every region and the length of its
// candidate-presence (or group) worklist, the occupied-cell counts, the guard and checksum words -- goes straight into pinned host memory from ONE
// small kernel (three copy commands with the bubbles between them were ~30 us of a 2 ms run)
__global__ void __launch_bounds__(256) k_run_end(RunEndArgs a)
{
	const uint32_t tid = threadIdx.x;
	for (int j = 0; j < a.n_jobs; j++) {
		const RunEndJob& jb = a.job[j];
		for (uint32_t t = tid; t < (uint32_t)((POOL_REGIONS + 1) * POOL_CTRL_WORDS); t += 256u) {
			const uint32_t r = t / (uint32_t)POOL_CTRL_WORDS, w = t % (uint32_t)POOL_CTRL_WORDS;
			jb.h_ctrl[t] = reinterpret_cast<const unsigned long long*>(jb.ctrl_cursor + (size_t)r * CTRL_STRIDE_U32)[w];
		}
		if (tid == 0 && jb.h_count) *jb.h_count = *jb.d_count;
		if (tid == 1 && jb.h_heavy) *jb.h_heavy = *jb.d_heavy;
	}
	for (uint32_t t = tid; t < (uint32_t)a.n_sets; t += 256u) a.h_nocc[t] = a.n_occ[t];
	for (size_t t = tid; t < a.n_words; t += 256u) a.h_words[t] = a.words[t];
}
void launch_run_end(const RunEndArgs& a, hipStream_t s)
{
	hipLaunchKernelGGL(k_run_end, dim3(1), dim3(256), 0, s, a);
}

// start of a run, ONE launch: the words the build kernels add to (guard flag, partial checksums: `words`, 64-bit), the occupied-cell counts
// of the sets that are built in this run (bit si of `sets`; at most 64 sets) and the cursors of the one-read bucket pass start at zero; the
// table entries the previous run set are cleared (k_table_clear's job); every pool pass of the run gets its control block (k_pool_begin's job).
// Round 4: these were three to four launches of a steady-state step, ~5 us of dispatch each, for a few microseconds of work.
__global__ void __launch_bounds__(256) k_run_begin(const RunBeginArgs a)
{
	const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
	for (size_t i = gtid; i < a.n_words; i += gsz) a.words[i] = 0ull;
	if (blockIdx.x == 0 && threadIdx.x < 64 && ((a.sets >> threadIdx.x) & 1ull)) a.n_occ[threadIdx.x] = 0u;
	for (int k = 0; k < a.n_zero; k++) for (size_t i = gtid; i < (size_t)a.n_zero_words[k]; i += gsz) a.zero[k][i] = 0u;
	for (int k = 0; k < a.n_clear; k++) {
		const RunBeginClear& c = a.clear[k];
		for (size_t i = gtid; i < (size_t)c.n; i += gsz) c.table[c.occ[i].y] = make_uint2(0u, 0u);
	}
	for (int k = 0; k < a.n_pool; k++) {
		const RunBeginPool& p = a.pool[k];
		if (blockIdx.x == (unsigned)k % gridDim.x) pool_begin_block(p.regions, p.ctrl);   // (uniform per block: the barrier inside is safe)
		if (gtid == 0) p.records[0] = 0;   // the empty record of the pool
		if (p.n_shared_empty == 0) continue;
		ulonglong2* o2 = reinterpret_cast<ulonglong2*>(p.offs);
		const size_t n2 = p.n_shared_empty / 2;
		for (size_t i = gtid; i < n2; i += gsz) o2[i] = make_ulonglong2(0ull, 0ull);
		if (gtid == 0 && (p.n_shared_empty & 1)) p.offs[p.n_shared_empty - 1] = 0ull;
	}
}
void launch_run_begin(const RunBeginArgs& a, hipStream_t s)
{
	static_assert(CTRL_SLOTS <= 256, "one thread per slot");
	size_t work = a.n_words;
	for (int k = 0; k < a.n_zero; k++) work = std::max<size_t>(work, a.n_zero_words[k]);
	for (int k = 0; k < a.n_clear; k++) work = std::max<size_t>(work, a.clear[k].n);
	for (int k = 0; k < a.n_pool; k++) work = std::max<size_t>(work, a.pool[k].n_shared_empty / 16);   // (16 bytes per store, 8 stores per thread)
	const size_t blocks = std::min<size_t>(2048, std::max<size_t>((work + 255) / 256, (size_t)std::max(a.n_pool, 1)));
	hipLaunchKernelGGL(k_run_begin, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// =====================================================================================================
// ascending neighbour lists (tnsx_options.sorted_lists).  The reference's lists are ascending by construction
// (TreeNSearch.cpp:2474-2500 emits in cell order over z-sorted input; BruteforceNSearch.cpp:135-137 sorts before comparing); the
// single-pass query writes them in lane order.  One wave per record, in place:
//   <= 64 entries    one entry per lane, bitonic network over the lanes (shuffles)
//   <= 2048 entries  bitonic network in the wave's LDS slice
//   longer           the same network directly on the record in global memory (rare: thousands of neighbours)
// Entries past the end act as +infinity and are never written.
// =====================================================================================================
static constexpr int SL_LDS = 2048;
__device__ __forceinline__ void bitonic_mem(int* a, uint32_t cnt, int lane)
{
	// `a` (LDS or global), any length.  The network form in which EVERY compare-exchange is ascending (the first stage of a merge
	// pairs i with its mirror image inside the block, i ^ (k - 1), the later ones with i ^ j): the absent elements past the end
	// then behave like +infinity that already sits in place, and a pair whose upper index is past the end is simply skipped.
	uint32_t n2 = 1;
	while (n2 < cnt) n2 <<= 1;
	auto stage = [&](uint32_t mask) {
		for (uint32_t i = (uint32_t)lane; i < cnt; i += WAVE) {
			const uint32_t l = i ^ mask;
			if (l > i && l < cnt) {
				const int x = a[i], y = a[l];
				if (x > y) { a[i] = y; a[l] = x; }
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};
	for (uint32_t k = 2; k <= n2; k <<= 1) {
		stage(k - 1u);
		for (uint32_t j = k >> 2; j > 0; j >>= 1) stage(j);
	}
}
__global__ void __launch_bounds__(256) k_sort_records(int* __restrict__ records, const uint64_t* __restrict__ offs_by_orig, int n_query)
{
	__shared__ int lds[(256 / WAVE) * SL_LDS];
	const int lane = lane_id();
	int* const my = lds + (threadIdx.x / WAVE) * SL_LDS;
	const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) / WAVE, n_waves = (size_t)gridDim.x * (256 / WAVE);
	for (size_t p = wave; p < (size_t)n_query; p += n_waves) {
		int* rec = records + offs_by_orig[p];
		const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(rec[0]);
		rec += 1;
		if (cnt <= 1u) continue;
		if (cnt <= (uint32_t)WAVE) {
			int v = (uint32_t)lane < cnt ? rec[lane] : 0x7fffffff;
			#pragma unroll
			for (int k = 2; k <= WAVE; k <<= 1) {
				#pragma unroll
				for (int j = k >> 1; j > 0; j >>= 1) {
					const int o = __shfl_xor(v, j, WAVE);
					const bool up = (lane & k) == 0, low = (lane & j) == 0;
					const int mn = v < o ? v : o, mx = v < o ? o : v;
					v = (low == up) ? mn : mx;
				}
			}
			if ((uint32_t)lane < cnt) rec[lane] = v;
		}
		else if (cnt <= (uint32_t)SL_LDS) {
			for (uint32_t i = (uint32_t)lane; i < cnt; i += WAVE) my[i] = rec[i];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			bitonic_mem(my, cnt, lane);
			for (uint32_t i = (uint32_t)lane; i < cnt; i += WAVE) rec[i] = my[i];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
		else {
			bitonic_mem(rec, cnt, lane);
		}
	}
}
void launch_sort_records(int* records, const uint64_t* offs_by_orig, int n_query, int n_cus, hipStream_t s)
{
	if (n_query <= 0) return;
	long long blocks = ((long long)n_query + 3) / 4;
	const long long cap = (long long)n_cus * 32;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL(k_sort_records, dim3((unsigned)blocks), dim3(256), 0, s, records, offs_by_orig, n_query);
}

}  // namespace tnsx
