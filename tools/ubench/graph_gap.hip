// graph_gap.hip -- what do dependent dispatches cost between two kernels of a frame, and does a hipGraph change it?
// Two stand-in kernels (every wave spins on the 100 MHz clock: A 300 us on a full persistent grid with 27 KB of LDS
// per workgroup like the trace kernel, B 40 us on 2048 workgroups of 1024 threads like the blur) launched as frames
//   mode 0: A, B on one stream, nothing else                      mode 1: ... plus one event record per frame
//   mode 2: ... plus a wait for an event of another stream        mode 3: a captured graph of 8 frames, replayed
//   mode 6 / 7: A as a fixed amount of WORK (~300 us), B as before; 7 adds a small kernel on another stream that depends on
//           nothing and so runs in the middle of A (what an upload or an exchange beside the trace grid is)
//   mode 4: like 1, but the event rides on B's own dispatch packet (hipExtLaunchKernelGGL's stopEvent) and the host
//           waits for it every eighth frame; mode 5: mode 1 with the same host waits (the control for 4)
//   hipcc --offload-arch=gfx950 -O2 -o graph_gap graph_gap.hip && ./graph_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void __launch_bounds__(256) spin_a(unsigned ticks, unsigned *sink)
{
	extern __shared__ unsigned lds[];
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	lds[threadIdx.x] = threadIdx.x;
	while(__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
	if(lds[threadIdx.x] == 0xffffffffu) *sink = 1;
}
__global__ void __launch_bounds__(1024) spin_b(unsigned ticks, unsigned *sink)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while(__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
	if(ticks == 0xffffffffu) *sink = 1;
}
// a fixed amount of work instead of a fixed time (for mode 6: is the work disturbed?)
__global__ void __launch_bounds__(256) work_a(unsigned iters, unsigned *sink)
{
	extern __shared__ unsigned lds[];
	lds[threadIdx.x] = threadIdx.x;
	float a = (float)threadIdx.x, b = 1.0001f;
	for(unsigned i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
	if(__float_as_uint(a) == iters || lds[(threadIdx.x + iters) & 255] == 0xffffffffu) *sink = 1;
}
__global__ void tiny(unsigned *sink) { if(threadIdx.x == 1000) *sink = 2; }

int main()
{
	unsigned *sink;
	CHECK(hipMalloc(&sink, 4));
	hipStream_t s, s2;
	CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	CHECK(hipFuncSetAttribute((const void *)spin_a, hipFuncAttributeMaxDynamicSharedMemorySize, 27 * 1024));
	CHECK(hipFuncSetAttribute((const void *)work_a, hipFuncAttributeMaxDynamicSharedMemorySize, 27 * 1024));
	hipEvent_t ev[8], ev2;
	for(int i = 0; i < 8; i++) CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
	CHECK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
	const unsigned TA = 30000, TB = 2000;      // 300 us, 20 us per resident round of B (two rounds: 2048 workgroups on 256 CUs x 2... a few rounds)
	const int frames = 400;
	auto frame = [&](int mode, int i)
	{
		if(mode == 2)
		{
			hipLaunchKernelGGL(tiny, dim3(8), dim3(256), 0, s2, sink);
			CHECK(hipEventRecord(ev2, s2));
			CHECK(hipStreamWaitEvent(s, ev2, 0));
		}
		if(mode >= 6) hipLaunchKernelGGL(work_a, dim3(1280), dim3(256), 27 * 1024, s, 6400u, sink);
		else hipLaunchKernelGGL(spin_a, dim3(1280), dim3(256), 27 * 1024, s, TA, sink);
		if(mode == 7) hipLaunchKernelGGL(tiny, dim3(8), dim3(256), 0, s2, sink);
		if(mode == 4) hipExtLaunchKernelGGL(spin_b, dim3(512), dim3(1024), 0, s, NULL, ev[i & 7], 0, TB, sink);
		else hipLaunchKernelGGL(spin_b, dim3(512), dim3(1024), 0, s, TB, sink);
		if(mode == 1 || mode == 2 || mode == 5) CHECK(hipEventRecord(ev[i & 7], s));
		if((mode == 4 || mode == 5) && i >= 3 && (i & 7) == 7) CHECK(hipEventSynchronize(ev[(i - 3) & 7]));
	};
	for(int mode = 0; mode < 8; mode++)
	{
		hipGraph_t g = NULL; hipGraphExec_t ge = NULL;
		if(mode == 3)
		{
			CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
			for(int i = 0; i < 8; i++) frame(0, i);
			CHECK(hipStreamEndCapture(s, &g));
			CHECK(hipGraphInstantiate(&ge, g, NULL, NULL, 0));
		}
		for(int rep = 0; rep < 3; rep++)
		{
			CHECK(hipDeviceSynchronize());
			const auto t0 = std::chrono::steady_clock::now();
			if(mode == 3) for(int i = 0; i < frames / 8; i++) CHECK(hipGraphLaunch(ge, s));
			else for(int i = 0; i < frames; i++) frame(mode, i);
			const auto t1 = std::chrono::steady_clock::now();
			CHECK(hipStreamSynchronize(s));
			const auto t2 = std::chrono::steady_clock::now();
			printf("mode %d: %.2f us per frame (host enqueue %.2f us per frame)\n", mode,
				std::chrono::duration<double, std::micro>(t2 - t0).count() / frames,
				std::chrono::duration<double, std::micro>(t1 - t0).count() / frames);
		}
	}
	return 0;
}
