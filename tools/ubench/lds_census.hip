// lds_census.hip -- how many 256-thread workgroups with a given dynamic-LDS size are
// really resident per CU on this chip, against what the occupancy API says.
// Every block stamps s_memrealtime at start and end of a ~200 us spin; a grid of
// 256 CUs x k blocks is "resident" if every block started before the first one ended.
//   hipcc --offload-arch=gfx950 -O2 -o lds_census lds_census.hip && ./lds_census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) spin(unsigned long long *t, int ticks)
{
	extern __shared__ unsigned char lds[];
	unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	if(threadIdx.x == 0) lds[0] = 1;
	while(__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
	if(threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	int ncu = p.multiProcessorCount;
	printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.gcnArchName, ncu,
		(size_t)p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
	hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	unsigned long long *d; hipMalloc(&d, 16 * ncu * 16);
	const int sizes[] = { 16384, 20480, 24576, 25792, 26624, 27824, 28672, 30016, 30720, 32768, 36864, 40960, 49152, 54000, 65536 };
	for(int lds : sizes)
	{
		int api = 0;
		hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, spin, 256, lds);
		int best = 0;
		for(int k = 1; k <= 10; k++)
		{
			int nb = ncu * k;
			hipMemset(d, 0, 16 * nb);
			hipLaunchKernelGGL(spin, dim3(nb), dim3(256), lds, 0, d, 20000);   // 200 us at 100 MHz
			if(hipDeviceSynchronize() != hipSuccess) break;
			std::vector<unsigned long long> h(2 * nb);
			hipMemcpy(h.data(), d, 16 * nb, hipMemcpyDeviceToHost);
			unsigned long long first_end = ~0ull, last_start = 0;
			for(int i = 0; i < nb; i++) { first_end = std::min(first_end, h[2 * i + 1]); last_start = std::max(last_start, h[2 * i]); }
			if(last_start < first_end) best = k; else break;
		}
		printf("dynamic LDS %6d B: resident blocks/CU %d, API says %d, floor(160K/lds) = %d, floor(128K/lds) = %d\n",
			lds, best, api, 163840 / lds, 131072 / lds);
	}
	return 0;
}
