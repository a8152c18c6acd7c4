// Micro-benchmark (MI355X): what does it cost to let every tile of a one-read bucket pass reserve its output range per bucket with ONE
// returning global atomic per (tile, bucket) -- T tiles x 1024 buckets, cursors `stride` words apart?  And, beside it, the empty-kernel
// launch-to-launch period on one stream (what a steady-state step pays per extra kernel).
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scatter atomic_scatter.hip && ./atomic_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_reserve(uint32_t* cursors, int stride, int per, uint32_t* sink, int nb)
{
	// thread t of tile b owns buckets (t * per + k + b * 37) % nb: every tile touches per * 256 buckets, neighbours start at different ones
	uint32_t acc = 0;
	for (int k = 0; k < per; k++) {
		const uint32_t d = (uint32_t)(threadIdx.x * per + k + blockIdx.x * 37) % (uint32_t)nb;
		acc += atomicAdd(&cursors[(size_t)d * stride], 4u);
	}
	if (acc == 0xffffffffu) sink[0] = acc;
}
__global__ void k_empty(uint32_t* p) { if (p && threadIdx.x == 999) p[0] = 1; }

int main()
{
	uint32_t *cur, *sink;
	CHK(hipMalloc(&cur, 4096 * 64 * 4)); CHK(hipMalloc(&sink, 64));
	hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int cfg[][4] = { { 2442, 4, 1, 1024 }, { 2442, 4, 32, 1024 }, { 610, 4, 1, 1024 }, { 610, 4, 32, 1024 }, { 2442, 8, 32, 2048 }, { 2442, 1, 32, 256 }, { 9768, 4, 32, 1024 } };
	for (auto& c : cfg) {
		CHK(hipMemsetAsync(cur, 0, 4096 * 64 * 4, st));
		for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_reserve, dim3(c[0]), dim3(256), 0, st, cur, c[2], c[1], sink, c[3]);
		CHK(hipEventRecord(e0, st));
		for (int rep = 0; rep < 10; rep++) hipLaunchKernelGGL(k_reserve, dim3(c[0]), dim3(256), 0, st, cur, c[2], c[1], sink, c[3]);
		CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("tiles %5d x %4d atomics/thread x 256 threads, %4d cursors, stride %2d words: %8.1f us per launch, %.2f G atomics/s\n", c[0], c[1], c[3], c[2], ms * 100.0,
		       (double)c[0] * 256 * c[1] / (ms / 10 * 1e-3) / 1e9);
	}
	// launch-to-launch period of tiny dependent kernels in one stream
	for (int n : { 1, 10, 100 }) {
		for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (uint32_t*)nullptr);
		CHK(hipStreamSynchronize(st));
		CHK(hipEventRecord(e0, st));
		for (int rep = 0; rep < n; rep++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (uint32_t*)nullptr);
		CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("%3d empty kernels back to back: %.2f us each\n", n, ms * 1e3 / n);
	}
	// the same captured in a graph
	{
		hipGraph_t g; hipGraphExec_t ge;
		CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
		for (int rep = 0; rep < 12; rep++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (uint32_t*)nullptr);
		CHK(hipStreamEndCapture(st, &g));
		CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		for (int rep = 0; rep < 3; rep++) CHK(hipGraphLaunch(ge, st));
		CHK(hipStreamSynchronize(st));
		CHK(hipEventRecord(e0, st));
		for (int rep = 0; rep < 10; rep++) CHK(hipGraphLaunch(ge, st));
		CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("graph of 12 empty kernels: %.2f us per graph launch = %.2f us per kernel\n", ms * 100.0, ms * 100.0 / 12);
	}
	return 0;
}
