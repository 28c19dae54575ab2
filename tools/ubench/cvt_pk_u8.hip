// cvt_pk_u8.hip -- does v_cvt_pk_u8_f32 agree with the reference's cvtps2dq (RNE) + packs_epi32 + packus_epi16 chain
// (util.h:48-59, dev_math.h ftoint_lane) on EVERY fp32 input?  Exhaustive: 2^32 bit patterns.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/cvt_pk_u8 tools/ubench/cvt_pk_u8.hip && tools/ubench/cvt_pk_u8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
__device__ uint32_t ref_lane(float s)
{
	int i = (int)rintf(s);
	i = min(max(i, 0), 255);
	return (s < 2147483648.0f) ? (uint32_t)i : 0u;
}
__global__ void k(unsigned long long *bad, uint32_t *first, uint32_t hi)
{
	uint32_t bits = (hi << 24) | (blockIdx.x * 256u + threadIdx.x);       // 2^24 patterns per launch
	float s = __uint_as_float(bits);
	uint32_t got;
	asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(got) : "v"(s));
	uint32_t want = ref_lane(s);
	if((got & 0xffu) != want)
	{
		unsigned long long n = atomicAdd(bad, 1ull);
		if(n < 16) { first[2 * n] = bits; first[2 * n + 1] = got; }
	}
}
int main()
{
	unsigned long long *bad; uint32_t *first;
	hipMalloc(&bad, 8); hipMalloc(&first, 16 * 8); hipMemset(bad, 0, 8); hipMemset(first, 0, 128);
	for(uint32_t hi = 0; hi < 256; hi++) hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, bad, first, hi);
	hipDeviceSynchronize();
	unsigned long long nb; uint32_t f[32];
	hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 128, hipMemcpyDeviceToHost);
	printf("v_cvt_pk_u8_f32 vs the reference's pack over all 2^32 inputs: %llu differ\n", nb);
	for(int i = 0; i < 16 && i < (int)nb; i++) { float x; memcpy(&x, &f[2 * i], 4); printf("  bits %08x (%g): got %u want %u\n", f[2 * i], x, f[2 * i + 1] & 0xff, 0u); }
	return 0;
}
