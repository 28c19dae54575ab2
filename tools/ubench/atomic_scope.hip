// atomic_scope.hip -- what does a returning atomic on a work-queue counter cost on MI355X, by memory scope?
// 1280 workgroups x 4 waves (the trace kernel's grid); every wave draws ITER tickets from a counter, each draw
// dependent on the one before (the wave waits for the value, like a wave that draws its next unit when it is
// done with the current one).  Counters: 64, 128 B apart.
//   mode 0  agent scope (what the trace kernel uses): counter = (wave number) % 64 -- shared by all XCDs
//   mode 1  agent scope, counter = 8 * XCC_ID + (wave % 8) -- every counter used by ONE XCD only
//   mode 2  workgroup scope on the per-XCD counters: the read-modify-write may stay in that XCD's L2
//   mode 3  wavefront scope on the per-XCD counters
// Checks: every ticket of every counter is drawn exactly once (a histogram on the host), i.e. the weaker scopes are
// still atomic among the waves of one XCD.
//   hipcc --offload-arch=gfx950 -O2 -o atomic_scope atomic_scope.hip && ./atomic_scope
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define ITER 64
#define NQ 64
#define STRIDE 32

template<int MODE>
__global__ void __launch_bounds__(256) draw(unsigned *counters, unsigned *got, unsigned long long *ticks, int work)
{
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const unsigned gw = blockIdx.x * 4 + wave;
	unsigned xcc = 0;
	// HW_REG_XCC_ID = 20 on gfx94x / gfx950: bits 3:0 the XCC
	xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
	const unsigned q = MODE == 0 ? gw % NQ : (8u * (xcc & 7u) + (gw % 8u));
	unsigned *c = counters + q * STRIDE;
	float a = (float)threadIdx.x;
	unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	for(int i = 0; i < ITER; i++)
	{
		unsigned t = 0;
		if(lane == 0)
		{
			if(MODE <= 1) t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else if(MODE == 2) t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		}
		t = __builtin_amdgcn_readfirstlane(t);
		if(lane == 0) got[(size_t)gw * ITER + i] = (q << 24) | t;
		// some work between draws (work = 0: back to back)
		for(int k = 0; k < work; k++) a = a * 1.0001f + 0.5f;
	}
	unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
	if(lane == 0) ticks[gw] = t1 - t0;
	if(a == 12345.0f) counters[0] = 1;
}

template<int MODE> static void run(const char *name, unsigned *d_c, unsigned *d_got, unsigned long long *d_t, int grid, int work)
{
	hipMemset(d_c, 0, NQ * STRIDE * 4);
	hipDeviceSynchronize();
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(draw<MODE>, dim3(grid), dim3(256), 0, 0, d_c, d_got, d_t, work);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const size_t n = (size_t)grid * 4 * ITER;
	std::vector<unsigned> got(n);
	std::vector<unsigned long long> t((size_t)grid * 4);
	hipMemcpy(got.data(), d_got, n * 4, hipMemcpyDeviceToHost);
	hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost);
	// every (queue, ticket) exactly once and tickets of a queue dense from 0
	std::sort(got.begin(), got.end());
	size_t dup = 0, gaps = 0;
	for(size_t i = 1; i < n; i++)
	{
		if(got[i] == got[i - 1]) dup++;
		else if((got[i] >> 24) == (got[i - 1] >> 24) && (got[i] & 0xffffff) != (got[i - 1] & 0xffffff) + 1) gaps++;
	}
	std::sort(t.begin(), t.end());
	printf("%-44s work %4d: kernel %.3f ms, per draw (wave's own clock): median %.2f us, p90 %.2f us; duplicates %zu, gaps %zu\n",
		name, work, ms, (double)t[t.size() / 2] / 100.0 / ITER, (double)t[t.size() * 9 / 10] / 100.0 / ITER, dup, gaps);
}

int main()
{
	int grid = 1280;
	unsigned *d_c, *d_got; unsigned long long *d_t;
	hipMalloc(&d_c, NQ * STRIDE * 4);
	hipMalloc(&d_got, (size_t)grid * 4 * ITER * 4);
	hipMalloc(&d_t, (size_t)grid * 4 * 8);
	for(int work = 0; work <= 20000; work = work ? work * 10 : 200)
	{
		run<0>("agent scope, 64 counters shared by all XCDs", d_c, d_got, d_t, grid, work);
		run<1>("agent scope, 8 counters per XCD", d_c, d_got, d_t, grid, work);
		run<2>("workgroup scope, 8 counters per XCD", d_c, d_got, d_t, grid, work);
		run<3>("wavefront scope, 8 counters per XCD", d_c, d_got, d_t, grid, work);
	}
	return 0;
}
