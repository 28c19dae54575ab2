// tables.h -- layout of the per-level table blob shared by the host packer
// (pwn_api.cpp) and the kernels.  The blob is built once per level / sphere
// upload, lives in HBM, and is copied verbatim into LDS by every workgroup.
//
//   [0      .. 16384)  cellinfo u32 [64][64]      per cell, ONE word for the walk loop:
//                       bits 0..7   cell type char        level.data   (defs.h:105)
//                       bits 8..15  min(#spheres, 255)    parts_num    (defs.h:108)
//                       bits 16..31 first entry in binidx (level.h:64-81 lists as CSR)
//   [16384  .. 20480)  rcp      u16 [2048]        RCPPS table     (trace.h:231)
//   [20480  .. 24576)  rsqrt    u16 [2048]        RSQRTPS table   (util.h:43)
//   [24576  .. 24784)  pmap     2 x u32 [26]      portals         (defs.h:87-94)
//                       word0 = x1 | z1<<8 | x2<<16 | z2<<24   (0xff = -1)
//                       word1 = rot12 | c1<<8 | c2<<16
//   [24784  .. 32992)  binoff   u16 [4104]        CSR offsets (only read when a cell
//                                                 holds >= 255 spheres)
//   [32992  .. +2*nbin pad 16)  binidx u16        sphere indices, object order
//   [...    .. +32*nsph)        spheres 8 x f32   r, refl, x, y, z, cb, cg, cr
#pragma once
#include <stdint.h>

#define PWN_T_CELLINFO 0u
#define PWN_T_RCP      16384u
#define PWN_T_RSQ      20480u
#define PWN_T_PMAP     24576u
#define PWN_T_BINOFF   24784u
#define PWN_T_BINIDX   32992u

static inline uint32_t pwn_t_sph_offset(uint32_t nbin)
{
	return PWN_T_BINIDX + ((nbin * 2u + 15u) & ~15u);
}
static inline uint32_t pwn_t_total(uint32_t nbin, uint32_t nsph)
{
	return pwn_t_sph_offset(nbin) + nsph * 32u;
}

// kernel arguments of one trace launch (rows [y0,y1) of a w x h frame)
struct pwn_trace_params
{
	float rayb[4], rdx[4], rdy[4], from[4];   // screen.h:43-57
	float sec_current;                        // defs.h:23
	int w, h, y0, y1;
	int tiles_x, tiles_total;
	uint32_t blob_bytes, off_sph;
	uint32_t *sbuf;                           // full frame, pitch w
	float *zbuf;                              // full frame, pitch w
	const uint32_t *blob;
	unsigned long long *counters;             // 5 x u64 or NULL
	int has_w;                                // camera has w components (general 4-lane path)
};
