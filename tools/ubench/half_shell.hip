// Micro-benchmark (MI355X), round 5: what would the SECOND append of a half-shell symmetric evaluation cost?  (test every pair once, in the
// "upper" half of the 27 cells, and write j into i's list AND i into j's.)  The first append is the cell kernels' own compaction on half as many hits.  The second one
// lands in the record of a point that belongs to another cell -- another wave's block, already flushed or not yet built -- so it is a returning atomic on that record's
// count word plus a scattered 4-byte store, per hit.  This kernel does exactly that and nothing else, with the locality the real thing would have: thread = query i in
// sorted order, its 30 "upper" neighbours spread over the runs of the 13 upper cells (a run of 3 cells ~ 42 consecutive sorted points; rows one grid row apart, layers one
// grid layer apart, as in the 93^3 grid of C2).  Records of fixed capacity 64 ints (the real thing would also need a count pass to size them).
//   hipcc --offload-arch=gfx950 -O3 -o half_shell half_shell.hip && ./half_shell
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_reverse_append(int* __restrict__ rec, int n, int per, int row, int layer, int mode)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t h = (uint32_t)i * 2654435761u;
	for (int k = 0; k < per; k++) {
		h = h * 1664525u + 1013904223u;
		// one of the 13 upper cells' runs: same row ahead (1 run), next row (3 runs over 3 x-cells = 1 run of 42), next layer (9 cells = 3 runs)
		const int which = (int)((h >> 8) % 5u);
		const long long base = which == 0 ? i + 1 : which == 1 ? (long long)i + row - 21 : (long long)i + layer - 21 + (which - 3) * (long long)row;
		long long j = base + (long long)((h >> 16) % 42u);
		j = j < 0 ? 0 : (j >= n ? n - 1 : j);
		if (mode == 0) { const int pos = atomicAdd(&rec[(size_t)j * 64], 1); rec[(size_t)j * 64 + 1 + (pos & 62)] = i; }   // returning atomic + scattered store
		else if (mode == 1) { atomicAdd(&rec[(size_t)j * 64], 1); }                                                          // the atomic alone (not returning)
		else { rec[(size_t)j * 64 + 1 + (k & 62)] = i; }                                                                     // the scattered store alone
	}
}
int main()
{
	const int n = 10000000, per = 30, row = 14 * 93, layer = 14 * 93 * 93;
	int* rec; CHK(hipMalloc(&rec, (size_t)n * 64 * 4));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const char* names[3] = { "returning atomic on the count word + scattered 4-byte store", "atomic alone (no return value)", "scattered 4-byte store alone" };
	for (int mode = 0; mode < 3; mode++) {
		CHK(hipMemset(rec, 0, (size_t)n * 64 * 4));
		hipLaunchKernelGGL(k_reverse_append, dim3((n + 255) / 256), dim3(256), 0, 0, rec, n, per, row, layer, mode);
		CHK(hipMemset(rec, 0, (size_t)n * 64 * 4));
		CHK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(k_reverse_append, dim3((n + 255) / 256), dim3(256), 0, 0, rec, n, per, row, layer, mode);
		CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("%d queries x %d reverse appends, %-62s %8.3f ms  (%.1f G per s)\n", n, per, names[mode], ms, (double)n * per / (ms * 1e-3) / 1e9);
	}
	printf("for comparison: the whole first query tier of C2 (every pair tested twice, both lists written) takes 1.39 ms; halving its tests would save at most ~0.35 ms\n");
	return 0;
}
