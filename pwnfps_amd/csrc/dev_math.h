// dev_math.h -- gfx950 device arithmetic for the ray-march kernels.
//
// Everything here exists to reproduce, bit for bit, what the reference's
// SSE2 + glibc build computes:
//   rcp / rsqrt  : _mm_rcp_ps (trace.h:231) / _mm_rsqrt_ps (util.h:43) are
//                  12-bit table functions on the reference host; the two
//                  2048-entry tables live in LDS (approx_tables.inc).
//   sinf/cosf/expf: glibc 2.35's float kernels (double-precision polynomials,
//                  FMA-contracted as in its x86-64 *_fma builds), trace.h:42-46,97.
//   v4 helpers   : util.h:18-46 (4-lane dot incl. w, summed (x+z)+(y+w)).
//   LCG          : util.h:1-16.   colour pack: util.h:48-59.
// The translation unit is compiled with -ffp-contract=off: every fp32
// multiply/add below is rounded separately, like the reference's non-FMA SSE2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The libm restatements are always inlined.  As real (noinline) device functions
// they were correct in isolation, but the 4-lane trace variants then rendered wrong
// pixels on ~10 % of random scenes, and whether they did depended on the LAYOUT of
// the code object (identical kernel code; appending an unrelated kernel made it go
// away; -O1 too).  The root cause was not isolated - the observations fit a callee
// clobbering a register in which the caller keeps a value (hipcc 7.2 keeps SGPR
// spills such as a non-leaf callee's return address in VGPR lanes via v_writelane).
// With no calls there is nothing of that kind to get wrong (and no scratch);
// tools/fuzz_parity.py is the regression check.
#ifndef PWN_LIBM_ATTR
#define PWN_LIBM_ATTR __forceinline__
#endif

struct v4 { float x, y, z, w; };

__device__ __forceinline__ v4 v4_set(float x, float y, float z, float w) { v4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
__device__ __forceinline__ v4 v4_add(v4 a, v4 b) { return v4_set(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ v4 v4_sub(v4 a, v4 b) { return v4_set(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ v4 v4_mul(v4 a, v4 b) { return v4_set(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ v4 v4_scale(float s, v4 a) { return v4_set(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float v4_dot(v4 a, v4 b)
{
	return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w);
}

// ---- RCPPS / RSQRTPS emulation; tab points into LDS -----------------------
// Both are pure table functions on the reference host:
//   rcp:   index = mantissa bits 22..12;  result = 2^(253-e+flag) * 1.m12
//   rsqrt: index = (e odd ? 1024 : 0) + mantissa bits 22..13 (i.e. bits 23..13
//          of the input as they lie);     result = 2^(190-h+flag) * 1.m12,
//          h = (e+1)>>1
// An LDS entry is the 13 bits flag<<12 | m12 (u16, built by pack_blob() from
// approx_tables.inc); BASE + (entry << 11) is the result for e = 0 / h = 0 as an
// fp32 bit pattern (BASE = 253<<23 resp. 190<<23), so a positive normal input
// costs an index, one ds_read_u16, one shift-add and a subtract in the
// exponent field.  Everything else takes the general path.
#include "tables.h"
// (TP: a pointer to the table -- an LDS-address-space pointer at a constant offset in the trace kernels, see
// trace_common.h, a plain one in the probe kernel)
template<class TP> __device__ __forceinline__ uint32_t rcp_entry(TP tab, uint32_t a)
{
	return PWN_RCP_BASE + ((uint32_t)tab[(a >> 12) & 2047u] << 11);
}
template<class TP> __device__ __forceinline__ float tab_rcp(TP tab, float x)
{
	uint32_t b = __float_as_uint(x);
	uint32_t sign = b & 0x80000000u, a = b & 0x7fffffffu;
	uint32_t r;
	if(a - 0x00800000u < 0x7e000000u)          // 1 <= e <= 252: result is a normal number
		r = sign | (rcp_entry(tab, a) - (a & 0x7f800000u));
	else
	{
//@SLOW
		uint32_t e = a >> 23, m = a & 0x7fffffu;
		if(e == 0u) r = sign | 0x7f800000u;      // zero / denormal (DAZ) -> inf
		else if(e == 255u) r = m ? (b | 0x00400000u) : sign;
		else
		{
			// e = 253, 254: the result is denormal (flushed) unless the entry's exponent allows it
			uint32_t t = rcp_entry(tab, a);
			int re = (int)(t >> 23) - (int)e;
			r = re <= 0 ? sign : (sign | (t - (e << 23)));
//@FAST
		}
	}
	return __uint_as_float(r);
}
// x >= EPSILON is known (ray components after the clamp of trace.h:220-222)
template<class TP> __device__ __forceinline__ float tab_rcp_pos(TP tab, float x)
{
	uint32_t a = __float_as_uint(x);
	return __uint_as_float(rcp_entry(tab, a) - (a & 0x7f800000u));
}

// (in two halves, so that a caller can put work of its own between the table read and its use)
template<class TP> __device__ __forceinline__ uint32_t tab_rsqrt_entry(TP tab, uint32_t b)
{
	// every lane takes the table path (the index is in range whatever the bits are)
	return (uint32_t)tab[(b >> 13) & 2047u];
}
__device__ __forceinline__ float tab_rsqrt_finish(uint32_t b, uint32_t entry)
{
	uint32_t r;
	// ... the rest is patched in behind ONE wave-uniform branch
	r = (PWN_RSQ_BASE + (entry << 11)) - (((b + 0x00800000u) >> 1) & 0x7f800000u);
	const bool special = !(b - 0x00800000u < 0x7f000000u);
	if(__builtin_expect(__ballot(special) != 0ull, 0))
	{
//@SLOW
		uint32_t sign = b & 0x80000000u, e = (b >> 23) & 0xffu, m = b & 0x7fffffu;
		uint32_t q;
		if(e == 255u && m) q = b | 0x00400000u;
		else if(e == 0u) q = sign | 0x7f800000u;
		else if(sign) q = 0xffc00000u;
		else q = 0u;                               // +inf
		r = special ? q : r;
//@FAST
	}
	return __uint_as_float(r);
}
template<class TP> __device__ __forceinline__ float tab_rsqrt(TP tab, float x)
{
	const uint32_t b = __float_as_uint(x);
	return tab_rsqrt_finish(b, tab_rsqrt_entry(tab, b));
}

template<class TP> __device__ __forceinline__ v4 v4_normalise(TP rsq, v4 a)
{
	return v4_scale(tab_rsqrt(rsq, v4_dot(a, a)), a);
}

// ---- LCG -------------------------------------------------------------------
__device__ __forceinline__ uint32_t lcg_next(uint32_t &s)
{
	s = (s * 25739u + 4u) & 0x7FFFFFFFu;
	return s;
}
__device__ __forceinline__ float lcg_fs(uint32_t &s)
{
	// (s % 3759) / 3759.0f compiled as a multiply by the float reciprocal
	// under the reference's -ffast-math
	const float inv = 1.0f / 3759.0f;
	float u = (float)(lcg_next(s) % 3759u) * inv;
	// u * 2 - 1 as one fused operation: doubling is exact, so there is still exactly one rounding
	return __builtin_fmaf(u, 2.0f, -1.0f);
}

// The same generator on the DOUBLED state t = 2 s (util.h:1-16 of the reference): the "& 0x7FFFFFFF" of every step is then the
// wrap of 32-bit arithmetic (2 ((25739 s + 4) mod 2^31) = (25739 t + 8) mod 2^32), the quotient s / 3759 is the same
// multiply-high (t M >> 44 = s M >> 43, M = ceil(2^43 / 3759), exact for every s < 2^31: checked for all 2^31 states on the
// CPU), the remainder comes out doubled from one 24-bit multiply-add (the quotient has 20 bits), and (float)(2 r) * (inv / 2)
// rounds exactly as (float)r * inv.  Two instructions per draw fewer than lcg_fs, one fewer than lcg_next; same bits.
// A state is doubled once (t = s << 1: bit 31 of s never mattered, the first step masks it away) and stays doubled.
__device__ __forceinline__ void lcg2_next(uint32_t &t) { t = t * 25739u + 8u; }
__device__ __forceinline__ float lcg2_fs(uint32_t &t)
{
	t = t * 25739u + 8u;
	const uint32_t q = __umulhi(t, 0x8b79b351u) >> 12;
	const uint32_t r2 = (uint32_t)(__mul24((int)q, -7518) + (int)t);
	const float u = (float)r2 * (0.5f * (1.0f / 3759.0f));
	return __builtin_fmaf(u, 2.0f, -1.0f);
}

// ---- colour pack: cvtps2dq (RNE) + packs_epi32 + packus_epi16 --------------
__device__ __forceinline__ uint32_t ftoint_lane(float f)
{
	// cvtps2dq gives 0x80000000 ("integer indefinite") for NaN and for anything
	// outside int32; the two saturating packs then clamp to [-32768,32767] and
	// [0,255], i.e. to [0,255], and the indefinite value ends as 0.
	// v_cvt_i32_f32 saturates instead, which differs only for s >= 2^31.
	float s = f * 255.0f;
	int i = (int)rintf(s);
	i = min(max(i, 0), 255);
	return (s < 2147483648.0f) ? (uint32_t)i : 0u;
}
// The four lanes of a colour at once.  v_cvt_pk_u8_f32 (round to nearest even, clamp to [0,255], insert as byte k) IS
// that chain for every fp32 input except s >= 2^31 and +inf, where it gives 255 and the reference's "integer
// indefinite" ends as 0: exhaustive over all 2^32 bit patterns, tools/ubench/cvt_pk_u8.hip.  Those are patched in
// behind one wave-uniform branch (v_max ignores a NaN lane, which both forms send to 0).
__device__ __forceinline__ uint32_t col_pack4(float x, float y, float z, float w)
{
	const float sx = x * 255.0f, sy = y * 255.0f, sz = z * 255.0f, sw = w * 255.0f;
	uint32_t pk = 0u;
	asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(pk) : "v"(sx));
	asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(pk) : "v"(sy));
	asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(pk) : "v"(sz));
	asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(pk) : "v"(sw));
	const bool over = fmaxf(fmaxf(sx, sy), fmaxf(sz, sw)) >= 2147483648.0f;
	if(__builtin_expect(__ballot(over) != 0ull, 0))
	{
//@SLOW
		if(sx >= 2147483648.0f) pk &= ~0x000000ffu;
		if(sy >= 2147483648.0f) pk &= ~0x0000ff00u;
		if(sz >= 2147483648.0f) pk &= ~0x00ff0000u;
		if(sw >= 2147483648.0f) pk &= ~0xff000000u;
//@FAST
	}
	return pk;
}
__device__ __forceinline__ uint32_t col_pack(v4 c) { return col_pack4(c.x, c.y, c.z, c.w); }

// ---- glibc 2.35 sinf / cosf ------------------------------------------------
// Published algorithm of sysdeps/ieee754/flt-32/s_sincosf.h (ARM optimized
// routines).  fma() here is v_fma_f64; the placement of the fused operations
// follows glibc's x86-64 FMA build, which is what the reference host runs.
#define PWN_HPI_INV 0x1.45F306DC9C883p+23
#define PWN_HPI     0x1.921FB54442D18p0
#define PWN_PI63    0x1.921FB54442D18p-62

__device__ __forceinline__ uint32_t abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }

// sign = +1/-1 flips the cosine polynomial (glibc's second table)
__device__ __forceinline__ float sincos_poly(double x, double x2, double csign, int n)
{
	if((n & 1) == 0)
	{
		double x3 = x * x2;
		double s1 = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
		double x7 = x3 * x2;
		double s = fma(x3, -0x1.555545995a603p-3, x);
		return (float)fma(x7, s1, s);
	}
	else
	{
		double x4 = x2 * x2;
		double c2 = fma(x2, csign * 0x1.99343027bf8c3p-16, csign * -0x1.6c087e89a359dp-10);
		double c1 = fma(x2, csign * -0x1.ffffffd0c621cp-2, csign * 0x1p0);
		double x6 = x4 * x2;
		double c = fma(x4, csign * 0x1.55553e1068f19p-5, c1);
		return (float)fma(x6, c2, c);
	}
}

__device__ __forceinline__ uint32_t inv_pio4_word(int i)
{
	// 4/pi as overlapping 32-bit windows, 8 bits apart
	const uint32_t t[24] = {
		0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
		0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd,
		0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43,
		0x993c4390, 0x3c439041 };
	return t[i];
}

// The two polynomials without the sign: with s, p = +-1 (they are), sincos_poly(x * s, x2, p, n) is
// s * sin_poly_pos(x, x2) for even n and p * cos_poly_pos(x2) for odd n exactly -- every product and every
// fused operation above is odd in s resp. p, and rounding is symmetric -- so the sign is one xor on the result
// instead of five f64 multiplies and two f64 selects per call.
__device__ __forceinline__ float sin_poly_pos(double x, double x2)
{
	double x3 = x * x2;
	double s1 = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
	double x7 = x3 * x2;
	double s = fma(x3, -0x1.555545995a603p-3, x);
	return (float)fma(x7, s1, s);
}
__device__ __forceinline__ float cos_poly_pos(double x2)
{
	double x4 = x2 * x2;
	double c2 = fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
	double c1 = fma(x2, -0x1.ffffffd0c621cp-2, 0x1p0);
	double x6 = x4 * x2;
	double c = fma(x4, 0x1.55553e1068f19p-5, c1);
	return (float)fma(x6, c2, c);
}
// 2^-12 <= |y| < 120 (top 12 bits): glibc's middle path.  It also reproduces its |y| < pi/4 path bit for bit
// (there n = 0: x = fma(-0.0, hpi, x) = x, s = p = 1), so only |y| < 2^-12, |y| >= 120, inf and NaN are left
// for the general code, and the callers test that once per wave.
__device__ __forceinline__ bool sincosf_is_mid(float y)
{
	return abstop12(y) - abstop12(0x1p-12f) < abstop12(120.0f) - abstop12(0x1p-12f);
}
// s = -1 for n & 3 in {1, 2}; p = -1 for n & 2 (glibc's __sincosf_table signs)
__device__ __forceinline__ uint32_t sincosf_sign_s(int n) { return (uint32_t)((n ^ (n >> 1)) & 1) << 31; }
__device__ __forceinline__ uint32_t sincosf_sign_p(int n) { return (uint32_t)((n >> 1) & 1) << 31; }

// which = 0: sinf, 1: cosf -- every input
//@SLOW
__device__ PWN_LIBM_ATTR float glibc_sincosf_general(float y, int which)
{
	double x = (double)y;
	int n;
	double s;
	if(abstop12(y) < abstop12(0x1.921FB6p-1f))
	{
		if(abstop12(y) < abstop12(0x1p-12f)) return which ? 1.0f : y;
		return sincos_poly(x, x * x, 1.0, which);
	}
	if(abstop12(y) < abstop12(120.0f))
	{
		double r = x * PWN_HPI_INV;
		n = ((int)r + 0x800000) >> 24;
		x = fma(-(double)n, PWN_HPI, x);
		s = (n & 3) == 0 || (n & 3) == 3 ? 1.0 : -1.0;
		return sincos_poly(x * s, x * x, (n & 2) ? -1.0 : 1.0, n ^ which);
	}
	if(abstop12(y) < abstop12(__builtin_inff()))
	{
		uint32_t xi = __float_as_uint(y);
		int sign = (int)(xi >> 31);
		int base = (int)((xi >> 26) & 15u);
		int shift = (int)((xi >> 23) & 7u);
		xi = (xi & 0xffffffu) | 0x800000u;
		xi <<= shift;
		uint64_t res0 = (uint64_t)(uint32_t)(xi * inv_pio4_word(base));
		uint64_t res1 = (uint64_t)xi * inv_pio4_word(base + 4);
		uint64_t res2 = (uint64_t)xi * inv_pio4_word(base + 8);
		res0 = (res2 >> 32) | (res0 << 32);
		res0 += res1;
		uint64_t nn = (res0 + (1ULL << 61)) >> 62;
		res0 -= nn << 62;
		x = (double)(int64_t)res0 * PWN_PI63;
		n = (int)nn;
		int q = (n + sign) & 3;
		s = (q == 0 || q == 3) ? 1.0 : -1.0;
		return sincos_poly(x * s, x * x, ((n + sign) & 2) ? -1.0 : 1.0, n ^ which);
	}
	return __builtin_nanf("");
//@FAST
}

// which = 0: sinf, 1: cosf
__device__ PWN_LIBM_ATTR float glibc_sincosf(float y, int which)
{
	if(__builtin_expect(__ballot(!sincosf_is_mid(y)) == 0ull, 1))
	{
		double x = (double)y;
		double r = x * PWN_HPI_INV;
		const int n = ((int)r + 0x800000) >> 24;
		x = fma(-(double)n, PWN_HPI, x);
		const int k = n ^ which;
		const double x2 = x * x;
		float pv; uint32_t sign;
		if(k & 1) { pv = cos_poly_pos(x2); sign = sincosf_sign_p(n); }
		else { pv = sin_poly_pos(x, x2); sign = sincosf_sign_s(n); }
		return __uint_as_float(__float_as_uint(pv) ^ sign);
	}
	return glibc_sincosf_general(y, which);
}

// sinf(y) and cosf(y) of the same argument (trace.h:45-46: the rippled floor
// normal): one range reduction, both polynomials evaluated once and handed out
// by quadrant.  Bit-identical to glibc_sincosf(y,0) / glibc_sincosf(y,1): in
// sinf_poly(x*s, x*x, p, n) and sinf_poly(x*s, x*x, p, n^1) everything but the
// choice of polynomial is shared.  x = sin, y = cos.
__device__ PWN_LIBM_ATTR float2 glibc_sincosf_both(float y)
{
	if(__builtin_expect(__ballot(!sincosf_is_mid(y)) == 0ull, 1))
	{
		double x = (double)y;
		double r = x * PWN_HPI_INV;
		const int n = ((int)r + 0x800000) >> 24;
		x = fma(-(double)n, PWN_HPI, x);
		const double x2 = x * x;
		const float ps = __uint_as_float(__float_as_uint(sin_poly_pos(x, x2)) ^ sincosf_sign_s(n));
		const float pc = __uint_as_float(__float_as_uint(cos_poly_pos(x2)) ^ sincosf_sign_p(n));
		return (n & 1) ? make_float2(pc, ps) : make_float2(ps, pc);
	}
	return make_float2(glibc_sincosf_general(y, 0), glibc_sincosf_general(y, 1));
}

// ---- glibc 2.35 expf (e_expf.c, N = 32) -------------------------------------
__device__ __forceinline__ uint64_t exp2f_tab(int i)
{
	const uint64_t t[32] = PWN_EXP2F_TAB_INIT;
	return t[i];
}

// lds_tab: the 32-entry table in LDS (tables.h PWN_T_EXP2), or NULL for the copy in constant memory (a
// per-lane global load: fine for the probe kernel, a long stall in the trace kernels)
template<class TP> __device__ PWN_LIBM_ATTR float glibc_expf_t(float x, TP lds_tab, bool have_tab)
{
	const double N = 32.0;
	const double InvLn2N = 0x1.71547652b82fep+0 * N;
	const double SHIFT = 0x1.8p+52;
	const double C0 = 0x1.c6af84b912394p-5 / N / N / N;
	double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
	const double C2 = 0x1.62e42ff0c52d6p-1 / N;
	// materialised here, at the (rare) call: hoisted to the top of the kernel this constant pair ended
	// in scratch memory (20 B per lane written by every wave of every launch: 6.5 MB per 4K frame)
	asm volatile("" : "+s"(C1));
	double xd = (double)x;
	uint32_t at = (__float_as_uint(x) >> 20) & 0x7ffu;
	// z = InvLn2N*xd is fused into both of its uses in glibc's FMA build
	double kd = fma(InvLn2N, xd, SHIFT);
	uint64_t ki = (uint64_t)__double_as_longlong(kd);
	kd -= SHIFT;
	double r = fma(InvLn2N, xd, -kd);
	uint64_t t = have_tab ? (uint64_t)lds_tab[ki & 31u] : exp2f_tab((int)(ki & 31u));
	t += ki << (52 - 5);
	double s = __longlong_as_double((long long)t);
	double z = fma(C0, r, C1);
	double r2 = r * r;
	double yv = fma(C2, r, 1.0);
	yv = fma(z, r2, yv);
	yv = yv * s;
	float res = (float)yv;
	// |x| >= 88, inf, NaN (e_expf.c's early returns): every lane computes the above -- harmless whatever x
	// is -- and these are patched in behind ONE wave-uniform branch
	const bool big = at >= ((__float_as_uint(88.0f) >> 20) & 0x7ffu);
	if(__builtin_expect(__ballot(big) != 0ull, 0))
	{
		if(big)
		{
//@SLOW
			if(__float_as_uint(x) == 0xff800000u) res = 0.0f;
			else if(at >= 0x7f8u) res = x + x;
			else if(x > 0x1.62e42ep6f) res = __builtin_inff();
			else if(x < -0x1.9fe368p6f) res = 0.0f;
//@FAST
		}
	}
	return res;
}
__device__ PWN_LIBM_ATTR float glibc_expf(float x) { return glibc_expf_t<const uint64_t *>(x, nullptr, false); }
template<class TP> __device__ PWN_LIBM_ATTR float glibc_expf(float x, TP lds_tab) { return glibc_expf_t<TP>(x, lds_tab, true); }
