// tables.h -- layout of the per-level table blob shared by the host packer
// (pwn_api.cpp) and the kernels.  The blob is built once per level / sphere
// upload, lives in HBM, and is copied verbatim into LDS by every workgroup.
//
//   [0      .. 4096)   rcp      u16 [2048]        RCPPS table     (trace.h:231), see dev_math.h
//   [4096   .. 8192)   rsqrt    u16 [2048]        RSQRTPS table   (util.h:43)
//   [8192   .. 8448)   faces    f32 [4][4] + [6][8]  constants of the shading step, looked up per lane:
//                       [base] = colour of a wall class, b g r - (trace.h:108-154; BASE_* of trace_common.h);
//                       [face] = what hitting face FXP..FYN does (defs.h:25-33): sign masks that mirror the
//                       ray, its reflectivity; the 0.001 step off the surface per axis (-0.0f where the
//                       axis is not the face's: x + -0 = x), the sign mask of the diffuse term
//   [8448   .. 8704)   exp2     u64 [32]          glibc expf's 2^(i/32) table (dev_math.h), for the fog composites
//   [8704   .. 25604)  cellinfo u32 [65][65]     ONE word per cell for the walk loop.
//                       Row/column 64 repeat row/column 0 WITHOUT the sphere bit:
//                       get_cell's per-axis clamp-to-0 (util.h:151-158) becomes
//                       min(c, 64) and the in-bounds test in front of the sphere
//                       loop (trace.h:252) is folded into the word.
//                       bits 0..7   cell type char        level.data   (defs.h:105)
//                       bit  8      PWN_C_ROOM   ; $ "  #  &   (1-high or 2-high room)
//                       bit  9      PWN_C_ROOM2  # &
//                       bit  10     PWN_C_FOG    $ &
//                       bit  11     PWN_C_DQ     "
//                       bit  12     PWN_C_RAMP   > < , ^
//                       bit  13     PWN_C_RAMPX  > <            (tilt along x)
//                       bit  14     PWN_C_RAMPM  > ,            (ray.y -= ramp*tilt)
//                       bit  15     PWN_C_PORTAL A..Z
//                       bits 16..30 first entry of this cell's sphere list in binidx
//                       bit  31     PWN_C_SPH    the cell holds >= 1 sphere
//   [25616  .. 25824)  pmap     2 x u32 [26]      portals         (defs.h:87-94)
//                       word0 = x1 | z1<<8 | x2<<16 | z2<<24   (0xff = -1)
//                       word1 = rot12 | c1<<8 | c2<<16
//   [25824  .. +2*nbin pad 16)  binidx u16        per-cell sphere lists (level.h:64-81),
//                                                 object order, as BYTE offsets into the sphere
//                                                 array (index * 32), each list closed by 0xffff
//   [...    .. +32*nsph)        spheres 8 x f32   x, y, z, r*r | refl, cb, cg, cr: what a test reads is ONE 16-byte
//                                                 load (r only ever enters as r*r, trace.h:262,270; the host squares it
//                                                 in fp32 like the reference does)
#pragma once
#include <stdint.h>

#define PWN_GRID_PITCH 65u
#define PWN_T_RCP      0u
#define PWN_T_RSQ      4096u
#define PWN_T_FACES    8192u
#define PWN_T_EXP2     8448u
#define PWN_T_CELLINFO 8704u
#define PWN_T_PMAP     25616u
#define PWN_T_BINIDX   25824u

// table entry -> fp32 pattern of the result for a zero exponent field (dev_math.h):
// entry = (1 - exponent offset) << 12 | result mantissa bits 22..11
#define PWN_RCP_BASE   0x7e800000u   /* 253 << 23 */
#define PWN_RSQ_BASE   0x5f000000u   /* 190 << 23 */

#define PWN_C_ROOM   0x0100u
#define PWN_C_ROOM2  0x0200u
#define PWN_C_FOG    0x0400u
#define PWN_C_DQ     0x0800u
#define PWN_C_RAMP   0x1000u
#define PWN_C_RAMPX  0x2000u
#define PWN_C_RAMPM  0x4000u
#define PWN_C_PORTAL 0x8000u
#define PWN_C_SPH    0x80000000u
#define PWN_LIST_END 0xffffu

static inline uint32_t pwn_t_sph_offset(uint32_t nbin)
{
	return PWN_T_BINIDX + ((nbin * 2u + 15u) & ~15u);
}
static inline uint32_t pwn_t_total(uint32_t nbin, uint32_t nsph)
{
	return pwn_t_sph_offset(nbin) + nsph * 32u;
}
// The per-cell lists with the sphere records INLINE (round 5; the unit scheduler's kernels, where the blob still leaves five
// workgroups per CU): at PWN_T_BINIDX nrec records of 16 bytes -- x, y, z, r*r of the sphere, one per (cell, sphere) pair in
// object order, the LAST record of a cell's list with the sign bit of r*r set (r*r is never negative) -- then nrec u16 "which
// sphere" (its byte offset in the sphere array: read when a hit is shaded, not per test), then the spheres as before.  A test is
// then ONE LDS read whose address does not depend on another read (level.txt: all 14 spheres sit in one cell, a ray through it
// makes 14 tests in a row).  The cell word's offset field counts records.
static inline uint32_t pwn_t_recsph_offset(uint32_t nrec) { return PWN_T_BINIDX + nrec * 16u; }
static inline uint32_t pwn_t_sph_offset_inl(uint32_t nrec) { return pwn_t_recsph_offset(nrec) + ((nrec * 2u + 15u) & ~15u); }
static inline uint32_t pwn_t_total_inl(uint32_t nrec, uint32_t nsph) { return pwn_t_sph_offset_inl(nrec) + nsph * 32u; }

// glibc 2.35 e_expf.c / exp2f_data: 2^(i/32) as double bit patterns with the exponent adjusted (N = 32)
#define PWN_EXP2F_TAB_INIT { \
	0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, \
	0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, \
	0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, \
	0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, \
	0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13, \
	0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, \
	0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, \
	0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540 }

// the constant shading tables at PWN_T_FACES and PWN_T_EXP2 (host: pack_blob)
static inline void pwn_fill_faces(float *f)
{
	// wall class -> colour factors (BASE_CEIL, BASE_FLOOR, BASE_WALL, BASE_MAGENTA of trace_common.h)
	static const float col[4][4] = { { 30.0f, 30.0f, 0.0f, 0.0f }, { 1.0f, 1.0f, 1.0f, 0.0f }, { 0.8f, 0.8f, 1.0f, 0.0f }, { 5.0f, 0.0f, 5.0f, 0.0f } };
	for(int i = 0; i < 16; i++) f[i] = col[i >> 2][i & 3];
	// face (FXP, FZP, FXN, FZN, FYP, FYN) -> mirror masks x y z, reflectivity | step x y z, diffuse sign mask
	uint32_t *u = (uint32_t *)(f + 16);
	const uint32_t S = 0x80000000u, P = 0x3a83126fu /* 0.001f */, N = 0xba83126fu, Z = 0x80000000u /* -0.0f */;
	const uint32_t R25 = 0x3e800000u /* 0.25f */, R70 = 0x3f333333u /* 0.7f */;
	const uint32_t t[6][8] = {
		{ S, 0, 0, R25,  N, Z, Z, 0 },        // FXP: ray.x = -ray.x, pos.x -= 0.001 (trace.h:50-75)
		{ 0, 0, S, R25,  Z, Z, N, 0 },        // FZP
		{ S, 0, 0, R25,  P, Z, Z, S },        // FXN: pos.x += 0.001, diffuse = -ray.x
		{ 0, 0, S, R25,  Z, Z, P, S },        // FZN
		{ 0, S, 0, R25,  Z, N, Z, 0 },        // FYP (ceiling)
		{ 0, 0, 0, R70,  Z, N, Z, S },        // FYN (floor): pos.y -= 0.001, then the rippled normal mirrors the ray
	};
	for(int i = 0; i < 48; i++) u[i] = t[i >> 3][i & 7];
	static const uint64_t e2[32] = PWN_EXP2F_TAB_INIT;
	uint64_t *e = (uint64_t *)(f + (PWN_T_EXP2 - PWN_T_FACES) / 4u);
	for(int i = 0; i < 32; i++) e[i] = e2[i];
}

// class bits of a cell type (trace.h:300-666 switch labels)
static inline uint32_t pwn_cell_class(uint32_t c)
{
	switch(c)
	{
		case ';': return PWN_C_ROOM;
		case '$': return PWN_C_ROOM | PWN_C_FOG;
		case '"': return PWN_C_ROOM | PWN_C_DQ;
		case '#': return PWN_C_ROOM | PWN_C_ROOM2;
		case '&': return PWN_C_ROOM | PWN_C_ROOM2 | PWN_C_FOG;
		case '>': return PWN_C_RAMP | PWN_C_RAMPX | PWN_C_RAMPM;
		case '<': return PWN_C_RAMP | PWN_C_RAMPX;
		case ',': return PWN_C_RAMP | PWN_C_RAMPM;
		case '^': return PWN_C_RAMP;
	}
	return (c >= 'A' && c <= 'Z') ? PWN_C_PORTAL : 0u;
}

// kernel arguments of one trace launch (rows [y0,y1) of a w x h frame)
struct pwn_trace_params
{
	float rayb[4], rdx[4], rdy[4], from[4];   // screen.h:43-57
	float sec_current;                        // defs.h:23
	int w, h, y0, y1;
	int tiles_x, tiles_total;                 // 16 x 4 pixel units (one wave64 each): per row, in all
	uint32_t ux_magic; int ux_shift;          // unit / tiles_x = (unit * ux_magic >> 32) >> ux_shift for unit < 2^31; ux_shift < 0: divide
	uint32_t blob_bytes, off_sph;
	uint32_t *sbuf;                           // full frame, pitch w
	float *zbuf;                              // full frame, pitch w
	const uint32_t *blob;
	unsigned long long *counters;             // 24 x u64 (pwn_stats counters, wave_paths, residency) or NULL
	unsigned long long *wave_log;             // counting builds, diagnosis: (start, end) of every wave, 100 MHz ticks; or NULL
	int has_w;                                // camera has w components (general 4-lane path)
	int scheduler;                            // PWN_SCHED_* (pwnhip.h)
	int refill_limit;                         // PWN_SCHED_REFILL: walk on while more lanes than this walk (trace_refill.hip)
	// work queues of the wave scheduler (trace_kernel.hip): PWN_QUEUES counters, one per 128 B,
	// for this launch; the set of the next launch, which this one clears
	uint32_t *tickets, *tickets_next;
	uint32_t *cost_word;                      // NULL, or a word this launch adds the sum of its waves' lifetimes to (100 MHz ticks): what
	                                          // the rows cost, for the row tiling's moving cuts (pwn_tiled.cpp)
	uint16_t *unit_cost;                      // NULL, or one entry per unit: how long the wave that traced it spent on it, in 40 ns (four ticks of the
	                                          // 100 MHz clock, saturating) -- what the NEXT launch of this geometry is ordered by (PWN_OPT_UNIT_ORDER,
	                                          // post_kernels.hip pwn_order_kernel), and what tools/unit_order_sim.py replays
	const uint32_t *perm;                     // NULL (units in arithmetic order: rows from the frame's middle outwards), or the hand-out order of
	uint32_t perm_cap;                        // every queue: perm[q * perm_cap + ticket] = the unit that ticket of queue q stands for
	uint32_t *clear_word;                     // NULL, or a word this launch sets to 0 (the row tiling's miss word of the frame:
	                                          // the blur of the same frame, behind this launch on the stream, counts in it)
	uint32_t off_recsph;                      // != 0: the blob holds the per-cell lists with inline sphere records (above); where their "which sphere" array is
};

#ifndef PWN_QUEUES
#define PWN_QUEUES 64u                       /* a power of two <= 64: one lane of a wave looks at each */
#endif
#define PWN_QUEUE_STRIDE 32u                  /* uint32 words between two counters (128 B) */
