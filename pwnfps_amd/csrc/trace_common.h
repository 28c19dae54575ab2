// trace_common.h -- types and helpers shared by the two trace kernels
// (trace_kernel.hip: a wave64 traces 16x4-pixel units in step; trace_refill.hip: lanes are
// refilled with new rays by ballot + prefix rank while the others walk on).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "tables.h"

#define EPS 0.0000000000001f      // defs.h:1
#define REFLECT_BLUR_F 0.03f      // defs.h:5
#define REFLECT_MAX 2             // defs.h:7
enum { FXP = 0, FZP, FXN, FZN, FYP, FYN };   // defs.h:25-33

// workgroup = PWN_BLOCK threads sharing one copy of the blob in LDS; a wave's unit of work is
// 16 x 4 pixels (half the width of the 32-pixel tile of screen.h:6-7, one DPP row per pixel row)
#ifndef PWN_BLOCK
#define PWN_BLOCK 256
#endif
#define TILE_W 16
#define TILE_H 4

// min waves per SIMD the register allocator must leave room for (Makefile MINW)
#ifndef PWN_MIN_WAVES
#define PWN_MIN_WAVES 3
#endif

// The tables of the blob sit at CONSTANT offsets of the workgroup's LDS (tables.h; the kernels have no static
// __shared__ data, so the dynamic allocation they are copied into starts at LDS address 0 -- the kernels check
// that once).  They are addressed through LDS-address-space pointers made from those constants, so a table
// access is a ds_read with an immediate offset; through the `extern __shared__` symbol every access adds that
// symbol's address (0, but known too late to fold): 21 `v_add_u32 v, 0, v` in the kernel, one per cell step.
#define PWN_LDS __attribute__((address_space(3)))
template<class T> __device__ __forceinline__ const PWN_LDS T *lds_at(uint32_t byte) { return (const PWN_LDS T *)(uintptr_t)byte; }
// (16-byte LDS loads as a built-in vector type: HIP's float4 class has no constructor from another address space)
typedef float pwn_f4 __attribute__((ext_vector_type(4)));
struct Lds
{
	const PWN_LDS uint32_t *cellinfo;
	const PWN_LDS uint16_t *rcp, *rsq;
	const PWN_LDS uint32_t *pmap;
	const PWN_LDS uint16_t *binidx;
	const PWN_LDS uint16_t *recsph;       // inline sphere records (tables.h): which sphere a record is of
	const PWN_LDS float *sph;
	const PWN_LDS uint64_t *exp2;         // tables.h PWN_T_EXP2
	const PWN_LDS pwn_f4 *faces;           // tables.h PWN_T_FACES: [0..4) wall colours, [4 + 2 * face ..] face constants
};
__device__ __forceinline__ Lds lds_tables(uint32_t off_sph, uint32_t off_recsph = 0u)
{
	Lds L;
	L.cellinfo = lds_at<uint32_t>(PWN_T_CELLINFO);
	L.rcp = lds_at<uint16_t>(PWN_T_RCP);
	L.rsq = lds_at<uint16_t>(PWN_T_RSQ);
	L.pmap = lds_at<uint32_t>(PWN_T_PMAP);
	L.binidx = lds_at<uint16_t>(PWN_T_BINIDX);
	L.faces = lds_at<pwn_f4>(PWN_T_FACES);
	L.exp2 = lds_at<uint64_t>(PWN_T_EXP2);
	L.sph = lds_at<float>(off_sph);
	L.recsph = lds_at<uint16_t>(off_recsph);
	return L;
}

// util.h:151-158 (per-axis clamp to 0) -> the packed cell word.  The table has
// 65 rows / columns; index 64 repeats index 0 (tables.h), so the clamp is a min.
__device__ __forceinline__ uint32_t cellword_at(const Lds &L, int cx, int cz)
{
	uint32_t ux = min((uint32_t)cx, 64u), uz = min((uint32_t)cz, 64u);
	return L.cellinfo[uz * PWN_GRID_PITCH + ux];
}

// The walk keeps the cell coordinates as two signed 16-bit fields of ONE register (x low, z high; a ray
// makes at most 1000 steps from a cell of the 64 x 64 grid, so 16 bits hold them), and its per-axis steps
// (gx, 0) / (0, gz) the same way.  A cell step is then one packed add, get_cell's per-axis clamp one
// packed unsigned min (a negative coordinate is a large unsigned one: 64, the repeat of index 0) and
// the byte offset x*4 + z*260 one v_dot2_u32_u16: 4 VALU where separate ints took 9.
typedef unsigned short pwn_us2 __attribute__((ext_vector_type(2)));
typedef short pwn_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cxz_pack(int cx, int cz) { return ((uint32_t)cx & 0xffffu) | ((uint32_t)cz << 16); }
// The cell a segment STARTS in comes from a float position, which can be anything -- a ray that came back from
// 10^13 units away, an infinity (the conversion saturates to INT_MIN / INT_MAX): 16 bits would fold such a cell
// back into or near the grid.  A start further out than 16383 cells is outside the grid for all of the
// segment's <= 1000 steps whatever its exact number, so it is pinned there.  (Found by lattice scenes of
// tools/fuzz_parity.py, seeds 9002 / 9004: tests/golden/far_starts.npz.)
__device__ __forceinline__ uint32_t cxz_pack_start(int cx, int cz)
{
	return cxz_pack(max(min(cx, 16383), -16384), max(min(cz, 16383), -16384));
}
__device__ __forceinline__ int cxz_x(uint32_t c) { return (int)(int16_t)(uint16_t)(c & 0xffffu); }
__device__ __forceinline__ int cxz_z(uint32_t c) { return (int)c >> 16; }
__device__ __forceinline__ uint32_t cxz_add(uint32_t c, uint32_t step)
{
	return __builtin_bit_cast(uint32_t, (pwn_s2)(__builtin_bit_cast(pwn_s2, c) + __builtin_bit_cast(pwn_s2, step)));
}
__device__ __forceinline__ uint32_t cellword_pk(const Lds &L, uint32_t cxz)
{
	const pwn_us2 lim = { 64, 64 }, pitch = { 4, (unsigned short)(PWN_GRID_PITCH * 4u) };
	const pwn_us2 c = __builtin_elementwise_min(__builtin_bit_cast(pwn_us2, cxz), lim);
	const uint32_t byte = __builtin_amdgcn_udot2(c, pitch, 0u, false);
	return *(const PWN_LDS uint32_t *)((const PWN_LDS unsigned char *)L.cellinfo + byte);
}

// HAS_W = false: the camera rows x,y,z carry w = 0 and the position w = 1
// (mat4_iden + rotations, main.c:61-64).  Then every ray has w = +-0 and every
// position w = 1, sphere-relative vectors have w = 0, and each 4-lane dot
// (x+z)+(y+w) of util.h:18-30 equals (x+z)+y bit for bit (a product of zeros
// adds +0; the only sign-of-zero effect is on a dot that is itself +-0, which
// the code only compares with 0 or squares).  The w lanes are dropped.
template<bool HAS_W> struct Vec { float x, y, z, w; };

template<bool W> __device__ __forceinline__ float dot3(const Vec<W> &a, const Vec<W> &b)
{
	if constexpr(W) return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w);
	else return (a.x * b.x + a.z * b.z) + a.y * b.y;
}
template<bool W> __device__ __forceinline__ Vec<W> vscale(float s, const Vec<W> &a)
{
	Vec<W> r; r.x = s * a.x; r.y = s * a.y; r.z = s * a.z;
	if constexpr(W) r.w = s * a.w; else r.w = 0.0f;
	return r;
}
template<bool W> __device__ __forceinline__ Vec<W> vadd(const Vec<W> &a, const Vec<W> &b)
{
	Vec<W> r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z;
	if constexpr(W) r.w = a.w + b.w; else r.w = 0.0f;
	return r;
}
template<bool W> __device__ __forceinline__ Vec<W> vsub(const Vec<W> &a, const Vec<W> &b)
{
	Vec<W> r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z;
	if constexpr(W) r.w = a.w - b.w; else r.w = 0.0f;
	return r;
}
// util.h:32-46
template<bool W> __device__ __forceinline__ Vec<W> vnormalise(const PWN_LDS uint16_t *rsq, const Vec<W> &a)
{
	return vscale<W>(tab_rsqrt(rsq, dot3<W>(a, a)), a);
}

// lane j of each 16-lane DPP row reads lane j-1; lane 0 reads 0.0f
__device__ __forceinline__ float dpp_row_shr1(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xf, 0xf, true));
}

enum { EV_NONE = 0, EV_WALL, EV_SPHERE, EV_EXHAUSTED };
// BASE_ROOM_Y: the ray left a room through its floor / ceiling (trace.h:323-329,373-379);
// the face and the colour follow from the ray's y sign after the walk
enum { BASE_CEIL = 0, BASE_FLOOR, BASE_WALL, BASE_MAGENTA, BASE_ROOM_Y };

// Regions of the kernel for the issue model (tools/issue_model.py): `//@R name` comments in the sources mark their
// extent, RG(k) counts -- in the counting variants -- how often a wave64 enters one with at least one lane
// (pwn_stats.regions).  The walk's own paths are the older wave_paths counters (WAVE_PATH) and wave_steps.
enum { RG_SEG = 0, RG_SETUP_SLOW, RG_EXHAUSTED, RG_WALL, RG_SPHERE, RG_FLOOR, RG_SPHREFL, RG_JITTER, RG_COMP1, RG_COMP1_FOG,
	RG_COMP2, RG_COMP2_FOG, RG_HELP, RG_UNIT, RG_SPHTEST, RG_SPHUPD, RG_ELSE,
	RG_UNIT_HALF, RG_HC_R2, RG_HC_OUT, RG_PORTAL_WALL, RG_PORTAL_GO, RG_PORTAL_ODD, RG_PORTAL_ROT2, RG_N };
struct Counters { uint32_t rays, steps, portals, tests, exhausted, wsteps, wp[8], apasses, apass_lanes, rg[RG_N]; };
// one count per wave64 that enters a code path with at least one lane (pwn_stats.wave_paths)
#define WAVE_PATH(k) do { if(COUNT && (__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) cnt.wp[k]++; } while(0)
#define RG(k) do { if(COUNT && (__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) cnt.rg[k]++; } while(0)
