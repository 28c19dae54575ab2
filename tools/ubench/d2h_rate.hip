// d2h_rate.hip -- how fast does a 4K frame (33 MB) reach pinned host memory while the next
// frame's kernels run?  Per frame: a compute kernel of ~0.42 ms (the trace + blur of a 4K frame:
// 1280 workgroups, 5 per CU) on one stream; the copy of that frame on another, behind an event.
//   mode 0      hipMemcpyAsync (SDMA; blit kernels with HSA_ENABLE_SDMA=0)
//   mode 1      hipMemcpyAsync in two halves on two streams
//   mode 2..    a copy kernel storing straight into the pinned buffer (optionally s_setprio 3)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/d2h_rate tools/ubench/d2h_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template<int PRIO>
__global__ void __launch_bounds__(256) copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
	if(PRIO) __builtin_amdgcn_s_setprio(3);
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t step = (size_t)gridDim.x * blockDim.x;
	for(; i + 3 * step < n16; i += 4 * step)
	{
		u32x4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
		__builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + step);
		__builtin_nontemporal_store(c, dst + i + 2 * step); __builtin_nontemporal_store(d, dst + i + 3 * step);
	}
	for(; i < n16; i += step) __builtin_nontemporal_store(src[i], dst + i);
}
__global__ void __launch_bounds__(256) busy_kernel(float *out, int iters)
{
	float a = threadIdx.x * 0.001f, b = 1.0001f;
	for(int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 0.00001f; }
	if(a == 12345.0f) out[0] = a + b;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
	const size_t bytes = 3840ull * 2160 * 4;
	const int reps = 60, NS = 3;
	u32x4 *d[NS]; void *h[NS]; float *dj;
	for(int i = 0; i < NS; i++) { CK(hipMalloc((void **)&d[i], bytes)); CK(hipMemset(d[i], 1 + i, bytes)); CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault)); }
	CK(hipMalloc((void **)&dj, 64));
	hipStream_t sc, sc2, sk;
	CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc2, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
	hipEvent_t ev[reps], ev2[reps];
	for(int i = 0; i < reps; i++) { CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev2[i], hipEventDisableTiming)); }
	// calibrate the compute kernel to 0.42 ms
	int iters = 20000;
	for(int k = 0; k < 4; k++)
	{
		CK(hipDeviceSynchronize());
		double t0 = now();
		for(int r = 0; r < 20; r++) hipLaunchKernelGGL(busy_kernel, dim3(1280), dim3(256), 0, sk, dj, iters);
		CK(hipDeviceSynchronize());
		double ms = (now() - t0) / 20 * 1e3;
		iters = (int)(iters * 0.42 / ms);
	}
	printf("compute kernel: %d iterations for 0.42 ms\n", iters);
	const int modes[][3] = { {0,0,0}, {1,0,0}, {2,16,0}, {2,32,0}, {2,64,0}, {2,16,1}, {2,32,1}, {2,64,1}, {2,128,1} };
	for(int busy = 0; busy < 2; busy++)
	for(size_t m = 0; m < sizeof(modes) / sizeof(modes[0]); m++)
	{
		const int mode = modes[m][0], grid = modes[m][1], prio = modes[m][2];
		CK(hipDeviceSynchronize());
		double t0 = now();
		for(int r = 0; r < reps; r++)
		{
			const int s = r % NS;
			if(busy) hipLaunchKernelGGL(busy_kernel, dim3(1280), dim3(256), 0, sk, dj, iters);
			CK(hipEventRecord(ev[r], sk));
			CK(hipStreamWaitEvent(sc, ev[r], 0));
			if(mode == 0) CK(hipMemcpyAsync(h[s], d[s], bytes, hipMemcpyDeviceToHost, sc));
			else if(mode == 1)
			{
				CK(hipStreamWaitEvent(sc2, ev[r], 0));
				CK(hipMemcpyAsync(h[s], d[s], bytes / 2, hipMemcpyDeviceToHost, sc));
				CK(hipMemcpyAsync((char *)h[s] + bytes / 2, (char *)d[s] + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, sc2));
			}
			else if(prio) hipLaunchKernelGGL(copy_kernel<1>, dim3(grid), dim3(256), 0, sc, (const u32x4 *)d[s], (u32x4 *)h[s], bytes / 16);
			else hipLaunchKernelGGL(copy_kernel<0>, dim3(grid), dim3(256), 0, sc, (const u32x4 *)d[s], (u32x4 *)h[s], bytes / 16);
		}
		CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(sc2));
		double t1 = now();
		CK(hipStreamSynchronize(sk));
		double t2 = now();
		char name[64];
		if(mode == 0) snprintf(name, sizeof(name), "hipMemcpyAsync");
		else if(mode == 1) snprintf(name, sizeof(name), "hipMemcpyAsync x2");
		else snprintf(name, sizeof(name), "copy kernel %3d wg%s", grid, prio ? " prio3" : "");
		printf("compute=%d %-26s %6.3f ms/frame %6.1f GB/s  (compute stream done after %.3f ms/frame)\n", busy, name,
			(t1 - t0) / reps * 1e3, bytes * reps / (t1 - t0) / 1e9, (t2 - t0) / reps * 1e3);
	}
	unsigned *p = (unsigned *)h[0];
	printf("check: %08x\n", p[12345]);
	return 0;
}
