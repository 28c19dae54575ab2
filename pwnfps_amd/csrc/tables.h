// tables.h -- layout of the per-level table blob shared by the host packer
// (pwn_api.cpp) and the kernels.  The blob is built once per level / sphere
// upload, lives in HBM, and is copied verbatim into LDS by every workgroup.
//
//   [0      .. 4096)   rcp      u16 [2048]        RCPPS table     (trace.h:231), see dev_math.h
//   [4096   .. 8192)   rsqrt    u16 [2048]        RSQRTPS table   (util.h:43)
//   [8192   .. 25104)  cellinfo u32 [65][65]     ONE word per cell for the walk loop.
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
//   [25104  .. 25312)  pmap     2 x u32 [26]      portals         (defs.h:87-94)
//                       word0 = x1 | z1<<8 | x2<<16 | z2<<24   (0xff = -1)
//                       word1 = rot12 | c1<<8 | c2<<16
//   [25312  .. +2*nbin pad 16)  binidx u16        per-cell sphere lists (level.h:64-81),
//                                                 object order, each list closed by 0xffff
//   [...    .. +32*nsph)        spheres 8 x f32   r, refl, x, y, z, cb, cg, cr
#pragma once
#include <stdint.h>

#define PWN_GRID_PITCH 65u
#define PWN_T_RCP      0u
#define PWN_T_RSQ      4096u
#define PWN_T_CELLINFO 8192u
#define PWN_T_PMAP     25104u
#define PWN_T_BINIDX   25312u

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
};

#ifndef PWN_QUEUES
#define PWN_QUEUES 64u                       /* a power of two <= 64: one lane of a wave looks at each */
#endif
#define PWN_QUEUE_STRIDE 32u                  /* uint32 words between two counters (128 B) */
