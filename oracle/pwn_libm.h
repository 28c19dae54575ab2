/*
 * oracle/pwn_libm.h -- TEST INFRASTRUCTURE (oracle side).
 *
 * The reference calls libm's sinf/cosf (trace.h:42-46; gcc merges the
 * same-argument pair into sincosf) and expf (trace.h:97).  libm is a
 * third-party dependency that is NOT under /root/reference: the reference
 * binary gets whatever the host's glibc provides.  The survey/build host has
 * glibc 2.35 (Ubuntu 2.35-0ubuntu3.11), whose float sin/cos/exp are the
 * ARM "optimized-routines" algorithms (sysdeps/ieee754/flt-32/s_sinf.c,
 * s_cosf.c, s_sincosf.h, s_sincosf_data.c, e_expf.c, e_exp2f_data.c),
 * dispatched on x86-64 to the -mfma -mavx2 builds (multiarch *_fma ifuncs),
 * i.e. every a+b*c in the double-precision kernels is one fused multiply-add.
 *
 * This header restates that published algorithm so that the oracle (and,
 * separately, the HIP device code) does not depend on the host's libm.
 * Pinning: tools/check_libm.c compares these functions with the container's
 * glibc on every float in the ranges the renderer can produce; the committed
 * KAT fixture tests/golden/libm_kat.bin holds (x, sinf, cosf, expf) samples
 * produced by that glibc for the GPU-side check.
 *
 * Out of range by construction and therefore not modelled: NaN/Inf inputs
 * to sinf/cosf (return NaN), expf overflow (x > 88.7: the renderer only
 * passes x = -0.6*fog <= 0).
 */
#ifndef PWN_ORACLE_LIBM_H
#define PWN_ORACLE_LIBM_H
#include <stdint.h>
#include <string.h>

#ifndef PWN_FMA
#define PWN_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

static inline uint32_t pwn_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline double pwn_asdouble(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t pwn_asuint64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline uint32_t pwn_abstop12(float x) { return (pwn_asuint(x) >> 20) & 0x7ff; }

/* polynomial data: [0] for quadrants where cos keeps its sign, [1] negated */
static const double pwn_sc_c[2][5] = {
	{ 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16 },
	{ -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16 },
};
static const double pwn_sc_s[3] = { -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 };
static const double pwn_sc_sign[4] = { 1.0, -1.0, -1.0, 1.0 };
#define PWN_HPI_INV 0x1.45F306DC9C883p+23 /* 2/pi * 2^24 */
#define PWN_HPI 0x1.921FB54442D18p0
#define PWN_PI63 0x1.921FB54442D18p-62

/* 4/pi as overlapping 32-bit windows, 8 bits apart */
static const uint32_t pwn_inv_pio4[24] = {
	0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
	0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd,
	0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43,
	0x993c4390, 0x3c439041,
};

/* sin if n even, cos if n odd, of x in [-pi/4,pi/4] (x2 = x*x) */
static inline float pwn_sinf_poly(double x, double x2, int tbl, int n)
{
	if((n & 1) == 0)
	{
		double x3 = x * x2;
		double s1 = PWN_FMA(x2, pwn_sc_s[2], pwn_sc_s[1]);
		double x7 = x3 * x2;
		double s = PWN_FMA(x3, pwn_sc_s[0], x);
		return (float)PWN_FMA(x7, s1, s);
	}
	else
	{
		const double *c = pwn_sc_c[tbl];
		double x4 = x2 * x2;
		double c2 = PWN_FMA(x2, c[4], c[3]);
		double c1 = PWN_FMA(x2, c[1], c[0]);
		double x6 = x4 * x2;
		double cc = PWN_FMA(x4, c[2], c1);
		return (float)PWN_FMA(x6, c2, cc);
	}
}

static inline double pwn_reduce_fast(double x, int *np)
{
	double r = x * PWN_HPI_INV;
	int n = ((int32_t)r + 0x800000) >> 24;
	*np = n;
	return PWN_FMA(-(double)n, PWN_HPI, x);
}

static inline double pwn_reduce_large(uint32_t xi, int *np)
{
	const uint32_t *arr = &pwn_inv_pio4[(xi >> 26) & 15];
	int shift = (xi >> 23) & 7;
	uint64_t n, res0, res1, res2;
	xi = (xi & 0xffffff) | 0x800000;
	xi <<= shift;
	res0 = xi * arr[0];
	res1 = (uint64_t)xi * arr[4];
	res2 = (uint64_t)xi * arr[8];
	res0 = (res2 >> 32) | (res0 << 32);
	res0 += res1;
	n = (res0 + (1ULL << 61)) >> 62;
	res0 -= n << 62;
	double x = (double)(int64_t)res0;
	*np = (int)n;
	return x * PWN_PI63;
}

/* which = 0: sinf, 1: cosf */
static inline float pwn_sincosf1(float y, int which)
{
	double x = y;
	int n;
	if(pwn_abstop12(y) < pwn_abstop12(0x1.921FB6p-1f)) /* |y| < pi/4 */
	{
		double x2 = x * x;
		if(pwn_abstop12(y) < pwn_abstop12(0x1p-12f))
			return which ? 1.0f : y;
		return pwn_sinf_poly(x, x2, 0, which);
	}
	else if(pwn_abstop12(y) < pwn_abstop12(120.0f))
	{
		x = pwn_reduce_fast(x, &n);
		double s = pwn_sc_sign[n & 3];
		int tbl = (n & 2) ? 1 : 0;
		return pwn_sinf_poly(x * s, x * x, tbl, n ^ which);
	}
	else if(pwn_abstop12(y) < pwn_abstop12(__builtin_inff()))
	{
		uint32_t xi = pwn_asuint(y);
		int sign = xi >> 31;
		x = pwn_reduce_large(xi, &n);
		double s = pwn_sc_sign[(n + sign) & 3];
		int tbl = ((n + sign) & 2) ? 1 : 0;
		return pwn_sinf_poly(x * s, x * x, tbl, n ^ which);
	}
	return __builtin_nanf("");
}

static inline float pwn_sinf(float x) { return pwn_sincosf1(x, 0); }
static inline float pwn_cosf(float x) { return pwn_sincosf1(x, 1); }

/* 2^(i/32) bit patterns minus (i << 47): table of e_exp2f_data.c (N = 32) */
static const uint64_t pwn_exp2f_tab[32] = {
	0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
	0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
	0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
	0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
	0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
	0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
	0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
	0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
#define PWN_EXP_N 32
#define PWN_EXP_INVLN2N (0x1.71547652b82fep+0 * PWN_EXP_N)
#define PWN_EXP_SHIFT 0x1.8p+52
#define PWN_EXP_C0 (0x1.c6af84b912394p-5 / PWN_EXP_N / PWN_EXP_N / PWN_EXP_N)
#define PWN_EXP_C1 (0x1.ebfce50fac4f3p-3 / PWN_EXP_N / PWN_EXP_N)
#define PWN_EXP_C2 (0x1.62e42ff0c52d6p-1 / PWN_EXP_N)

static inline float pwn_expf(float x)
{
	double xd = (double)x;
	uint32_t abstop = (pwn_asuint(x) >> 20) & 0x7ff;
	if(abstop >= ((pwn_asuint(88.0f) >> 20) & 0x7ff))
	{
		if(pwn_asuint(x) == pwn_asuint(-__builtin_inff())) return 0.0f;
		if(abstop >= ((pwn_asuint(__builtin_inff()) >> 20) & 0x7ff)) return x + x;
		if(x > 0x1.62e42ep6f) return __builtin_inff();
		if(x < -0x1.9fe368p6f) return 0.0f;
	}
	/* z = InvLn2N*xd only feeds an add and a subtract, so the -mfma build of
	   glibc fuses it into both (no separately rounded z exists) */
	double kd = PWN_FMA(PWN_EXP_INVLN2N, xd, PWN_EXP_SHIFT);
	uint64_t ki = pwn_asuint64(kd);
	kd -= PWN_EXP_SHIFT;
	double r = PWN_FMA(PWN_EXP_INVLN2N, xd, -kd);
	double z;
	uint64_t t = pwn_exp2f_tab[ki % PWN_EXP_N];
	t += ki << (52 - 5);
	double s = pwn_asdouble(t);
	z = PWN_FMA(PWN_EXP_C0, r, PWN_EXP_C1);
	double r2 = r * r;
	double y = PWN_FMA(PWN_EXP_C2, r, 1.0);
	y = PWN_FMA(z, r2, y);
	y = y * s;
	return (float)y;
}

#endif
