/*
 * tools/gen_approx_tables.c -- capture the host CPU's RCPPS / RSQRTPS
 * approximation tables and verify the table emulation against the hardware
 * instruction over ALL 2^32 float bit patterns.
 *
 * The reference computes 1/|ray| with _mm_rcp_ps (trace.h:231) and
 * normalises with _mm_rsqrt_ps (util.h:43).  Both are vendor specific
 * 12-bit approximations; on the Intel survey/build host they are pure table
 * functions of (top 11 mantissa bits) resp. (exponent parity, top 10 mantissa
 * bits) -- SURVEY.md App. B2.  This tool must therefore be run on that host.
 *
 *   gen_approx_tables gen   <rcp.u16> <rsqrt.u16>   write 2 x 2048 u16 tables
 *   gen_approx_tables check <rcp.u16> <rsqrt.u16>   exhaustive emulation check
 *
 * Table entry: bit 12 = "off" (result exponent is one lower than the
 * power-of-two reciprocal), bits 11..0 = the 12 result mantissa bits
 * (float mantissa bits 22..11; bits 10..0 of every result are zero).
 *
 * gcc -O2 -fopenmp tools/gen_approx_tables.c -o /tmp/gen_approx_tables
 */
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <xmmintrin.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline uint32_t hw_rcp(uint32_t b) { return f2u(_mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(u2f(b))))); }
static inline uint32_t hw_rsqrt(uint32_t b) { return f2u(_mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(u2f(b))))); }

static uint16_t rcp_tab[2048], rsq_tab[2048];

/* the emulation under test; the same logic lives in oracle/approx_tables.h
   and pwnfps_amd/csrc/dev_math.h */
static inline uint32_t emu_rcp(uint32_t b)
{
	uint32_t sign = b & 0x80000000u;
	uint32_t e = (b >> 23) & 0xff;
	uint32_t m = b & 0x7fffffu;
	if(e == 0) return sign | 0x7f800000u;          /* zero/denormal -> inf */
	if(e == 255) return m ? (b | 0x00400000u) : sign; /* nan quieted, inf -> 0 */
	uint32_t t = rcp_tab[m >> 12];
	int re = 254 - (int)e - (int)(t >> 12);
	if(re <= 0) return sign;                       /* tiny -> flushed to zero */
	return sign | ((uint32_t)re << 23) | ((t & 0xfffu) << 11);
}

static inline uint32_t emu_rsqrt(uint32_t b)
{
	uint32_t sign = b & 0x80000000u;
	uint32_t e = (b >> 23) & 0xff;
	uint32_t m = b & 0x7fffffu;
	if(e == 255 && m) return b | 0x00400000u;      /* nan quieted */
	if(e == 0) return sign | 0x7f800000u;          /* +-0/denormal -> +-inf */
	if(sign) return 0xffc00000u;                   /* negative -> indefinite */
	if(e == 255) return 0;                         /* +inf -> +0 */
	int E = (int)e - 127;
	int par = E & 1;
	uint32_t t = rsq_tab[(par << 10) | (m >> 13)];
	int re = 127 - (int)(t >> 12) - ((E - par) >> 1);
	return ((uint32_t)re << 23) | ((t & 0xfffu) << 11);
}

static int gen(const char *f1, const char *f2)
{
	for(int i = 0; i < 2048; i++)
	{
		uint32_t r = hw_rcp(0x3f800000u | ((uint32_t)i << 12));
		if(r & 0x7ff) { fprintf(stderr, "rcp: low bits set at %d\n", i); return 1; }
		int off = 127 - (int)((r >> 23) & 0xff);
		if(off < 0 || off > 1) { fprintf(stderr, "rcp: exponent out of range at %d\n", i); return 1; }
		rcp_tab[i] = (uint16_t)((off << 12) | ((r >> 11) & 0xfff));
	}
	for(int i = 0; i < 2048; i++)
	{
		int par = i >> 10;
		uint32_t r = hw_rsqrt(((uint32_t)(127 + par) << 23) | ((uint32_t)(i & 1023) << 13));
		if(r & 0x7ff) { fprintf(stderr, "rsqrt: low bits set at %d\n", i); return 1; }
		int off = 127 - (int)((r >> 23) & 0xff);
		if(off < 0 || off > 1) { fprintf(stderr, "rsqrt: exponent out of range at %d\n", i); return 1; }
		rsq_tab[i] = (uint16_t)((off << 12) | ((r >> 11) & 0xfff));
	}
	FILE *fp = fopen(f1, "wb"); fwrite(rcp_tab, 2, 2048, fp); fclose(fp);
	fp = fopen(f2, "wb"); fwrite(rsq_tab, 2, 2048, fp); fclose(fp);
	printf("rcp(1)=%08x rcp(3)=%08x rsq(1)=%08x rsq(2)=%08x\n",
		hw_rcp(0x3f800000u), hw_rcp(0x40400000u), hw_rsqrt(0x3f800000u), hw_rsqrt(0x40000000u));
	return 0;
}

static int check(const char *f1, const char *f2)
{
	FILE *fp = fopen(f1, "rb"); if(!fp || fread(rcp_tab, 2, 2048, fp) != 2048) return 2; fclose(fp);
	fp = fopen(f2, "rb"); if(!fp || fread(rsq_tab, 2, 2048, fp) != 2048) return 2; fclose(fp);
	long long bad_rcp = 0, bad_rsq = 0;
	uint32_t first_rcp = 0, first_rsq = 0;
	/* the reference executable runs with FTZ|DAZ (crtfastmath.o); check in
	   that mode and in the default mode */
	for(int mode = 0; mode < 2; mode++)
	{
#pragma omp parallel reduction(+:bad_rcp,bad_rsq)
		{
			unsigned csr = _mm_getcsr();
			_mm_setcsr(mode ? (csr | 0x8040) : (csr & ~0x8040u));
#pragma omp for schedule(static)
			for(long long i = 0; i < (1LL << 32); i++)
			{
				uint32_t b = (uint32_t)i;
				uint32_t h = hw_rcp(b), e = emu_rcp(b);
				if(h != e) { if(!bad_rcp) first_rcp = b; bad_rcp++; }
				h = hw_rsqrt(b); e = emu_rsqrt(b);
				if(h != e) { if(!bad_rsq) first_rsq = b; bad_rsq++; }
			}
			_mm_setcsr(csr);
		}
		printf("mode %s: rcp mismatches %lld (first %08x hw %08x emu %08x), rsqrt mismatches %lld (first %08x hw %08x emu %08x)\n",
			mode ? "FTZ|DAZ" : "default",
			bad_rcp, first_rcp, hw_rcp(first_rcp), emu_rcp(first_rcp),
			bad_rsq, first_rsq, hw_rsqrt(first_rsq), emu_rsqrt(first_rsq));
	}
	return (bad_rcp || bad_rsq) ? 1 : 0;
}

int main(int argc, char **argv)
{
	if(argc == 4 && !strcmp(argv[1], "gen")) return gen(argv[2], argv[3]);
	if(argc == 4 && !strcmp(argv[1], "check")) return check(argv[2], argv[3]);
	fprintf(stderr, "usage: %s gen|check rcp.u16 rsqrt.u16\n", argv[0]);
	return 2;
}
