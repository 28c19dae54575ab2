/*
 * tools/check_libm.c -- pin oracle/pwn_libm.h against the container's glibc
 * (the libm the compiled reference links) on EVERY float bit pattern:
 * sinf and cosf on all finite inputs, expf on all non-NaN inputs <= 88,
 * in the default MXCSR mode and in FTZ|DAZ (what a -ffast-math executable
 * such as the reference runs with).
 *
 * gcc -O2 -fopenmp -ffp-contract=off -Ioracle tools/check_libm.c -o /tmp/check_libm -lm
 */
#include <stdio.h>
#include <math.h>
#include <xmmintrin.h>
#include "pwn_libm.h"

int main(void)
{
	int rc = 0;
	for(int mode = 0; mode < 2; mode++)
	{
		long long bs = 0, bc = 0, be = 0, ns = 0, ne = 0;
#pragma omp parallel reduction(+:bs,bc,be,ns,ne)
		{
			unsigned csr = _mm_getcsr();
			_mm_setcsr(mode ? (csr | 0x8040) : (csr & ~0x8040u));
#pragma omp for schedule(dynamic, 1 << 20)
			for(long long i = 0; i < (1LL << 32); i++)
			{
				uint32_t b = (uint32_t)i;
				float x; memcpy(&x, &b, 4);
				if(((b >> 23) & 0xff) == 0xff) continue;
				ns++;
				if(pwn_asuint(sinf(x)) != pwn_asuint(pwn_sinf(x))) bs++;
				if(pwn_asuint(cosf(x)) != pwn_asuint(pwn_cosf(x))) bc++;
				if(x <= 88.0f) { ne++; if(pwn_asuint(expf(x)) != pwn_asuint(pwn_expf(x))) be++; }
			}
			_mm_setcsr(csr);
		}
		printf("mode %-8s sinf %lld/%lld cosf %lld/%lld expf %lld/%lld mismatches\n",
			mode ? "FTZ|DAZ" : "default", bs, ns, bc, ns, be, ne);
		if(bs || bc || be) rc = 1;
	}
	return rc;
}
