// Stand-alone timing + validation harness of the build stage (cell sort, tnsx_build.hip) on synthetic uniform points.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I include -I treensearch_amd/csrc -o tools/ubench/cellsort_bench \
//         tools/ubench/cellsort_bench.hip treensearch_amd/csrc/tnsx_build.hip treensearch_amd/csrc/tnsx_kernels.hip
//   tools/ubench/cellsort_bench [n_points] [cells_per_axis] [iterations]
#include "tnsx_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_points(float* xyz, int n, uint64_t seed)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	for (int c = 0; c < 3; c++) {
		uint64_t z = seed + (uint64_t)(3 * (uint64_t)i + c) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
		xyz[3 * (size_t)i + c] = (float)(z >> 40) * (1.0f / 16777216.0f);
	}
}

int main(int argc, char** argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 10000000;
	const int cells = argc > 2 ? atoi(argv[2]) : 89;
	const int iters = argc > 3 ? atoi(argv[3]) : 10;
	tnsx::GridParams g{ 0.f, 0.f, 0.f, (float)cells, cells, cells, cells };
	int key_bits = 1; while ((1ull << key_bits) < (uint64_t)cells * cells * cells) key_bits++;
	const tnsx::CellSortPlan plan = tnsx::cell_sort_plan(key_bits);
	printf("n=%d grid=%d^3 key_bits=%d passes=%d bits=%d,%d,%d\n", n, cells, key_bits, plan.passes, plan.bits[0], plan.bits[1], plan.bits[2]);

	float* xyz; CK(hipMalloc(&xyz, (size_t)n * 12));
	hipLaunchKernelGGL(k_points, dim3((n + 255) / 256), dim3(256), 0, 0, xyz, n, 12345ull);
	tnsx::CellSortBuffers b{};
	for (int k = 0; k < 2; k++) { CK(hipMalloc(&b.xyzi[k], (size_t)n * 16)); b.r2[k] = nullptr; }
	uint2* table; uint2* occ; uint32_t* n_occ;
	CK(hipMalloc(&table, (size_t)cells * cells * cells * 8)); CK(hipMalloc(&occ, (size_t)n * 8)); CK(hipMalloc(&n_occ, 4));
	void* temp; CK(hipMalloc(&temp, tnsx::cell_sort_temp_bytes(n)));
	hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
	int res = 0;
	float t_sort = 0, t_table = 0;
	for (int it = 0; it < iters + 2; it++) {
		CK(hipMemsetAsync(table, 0, (size_t)cells * cells * cells * 8, 0));
		CK(hipMemsetAsync(n_occ, 0, 4, 0));
		CK(hipEventRecord(e0, 0));
		res = tnsx::launch_cell_sort(xyz, nullptr, n, g, key_bits, b, temp, 0);
		CK(hipEventRecord(e1, 0));
		tnsx::launch_cell_table(b.xyzi[res], n, g, table, occ, n_occ, 0);
		CK(hipEventRecord(e2, 0));
		CK(hipEventSynchronize(e2));
		float a, c; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&c, e1, e2));
		if (it >= 2) { t_sort += a; t_table += c; }
	}
	printf("cell sort %.4f ms   cell table %.4f ms   total %.4f ms\n", t_sort / iters, t_table / iters, (t_sort + t_table) / iters);

	// validation: keys non-decreasing, every point carries its own key, every original index exactly once, stable inside a cell
	std::vector<uint32_t> keys(n); std::vector<float> pts((size_t)n * 4); std::vector<float> in((size_t)n * 3);
	CK(hipMemcpy(pts.data(), b.xyzi[res], (size_t)n * 16, hipMemcpyDeviceToHost));
	CK(hipMemcpy(in.data(), xyz, (size_t)n * 12, hipMemcpyDeviceToHost));
	std::vector<char> seen(n, 0);
	size_t bad = 0;
	auto coord = [&](float p) { int c = (int)((p - 0.f) * (float)cells); c = c < 0 ? 0 : c; return c > cells - 1 ? cells - 1 : c; };
	for (int p = 0; p < n; p++) {
		uint32_t idx; memcpy(&idx, &pts[4 * (size_t)p + 3], 4);
		if (idx >= (uint32_t)n || seen[idx]) { bad++; continue; }
		seen[idx] = 1;
		const uint32_t k = (uint32_t)((coord(in[3 * (size_t)idx + 2]) * cells + coord(in[3 * (size_t)idx + 1])) * cells + coord(in[3 * (size_t)idx]));
		keys[p] = k;
		if (pts[4 * (size_t)p] != in[3 * (size_t)idx] || pts[4 * (size_t)p + 2] != in[3 * (size_t)idx + 2]) bad++;
		if (p > 0) {
			uint32_t pidx; memcpy(&pidx, &pts[4 * (size_t)p - 1], 4);
			if (keys[p - 1] > keys[p] || (keys[p - 1] == keys[p] && pidx > idx)) bad++;
		}
	}
	// cell table against the sorted keys
	std::vector<uint32_t> tab((size_t)cells * cells * cells * 2);
	CK(hipMemcpy(tab.data(), table, tab.size() * 4, hipMemcpyDeviceToHost));
	uint32_t nocc = 0; CK(hipMemcpy(&nocc, n_occ, 4, hipMemcpyDeviceToHost));
	uint32_t distinct = 0;
	for (int p = 0; p < n; p++) {
		if (p == 0 || keys[p] != keys[p - 1]) { distinct++; if (tab[2 * (size_t)keys[p]] != (uint32_t)p) bad++; }
		if (p == n - 1 || keys[p] != keys[p + 1]) { if (tab[2 * (size_t)keys[p] + 1] != (uint32_t)p + 1) bad++; }
	}
	if (distinct != nocc) bad++;
	printf("validation: %zu bad of %d (occupied cells %u)\n", bad, n, nocc);
	return bad ? 1 : 0;
}
