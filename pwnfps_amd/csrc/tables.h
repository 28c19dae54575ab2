// tables.h -- layout of the per-level table blob shared by the host packer
// (pwn_api.cpp) and the kernels.  The blob is built once per level / sphere
// upload, lives in HBM, and is copied verbatim into LDS by every workgroup.
//
//   [0      .. 4096 )  cells    u8  [64][64]      level.data      (defs.h:105)
//   [4096   .. 8192 )  rcp      u16 [2048]        RCPPS table     (trace.h:231)
//   [8192   .. 12288)  rsqrt    u16 [2048]        RSQRTPS table   (util.h:43)
//   [12288  .. 12496)  pmap     2 x u32 [26]      portals         (defs.h:87-94)
//                       word0 = x1 | z1<<8 | x2<<16 | z2<<24   (0xff = -1)
//                       word1 = rot12 | c1<<8 | c2<<16
//   [12496  .. 20704)  binoff   u16 [4104]        CSR offsets per cell (level.h:64-81)
//   [20704  .. +2*nbin pad 16)  binidx u16        sphere indices, object order
//   [...    .. +32*nsph)        spheres 8 x f32   r, refl, x, y, z, cb, cg, cr
#pragma once
#include <stdint.h>

#define PWN_T_CELLS   0u
#define PWN_T_RCP     4096u
#define PWN_T_RSQ     8192u
#define PWN_T_PMAP    12288u
#define PWN_T_BINOFF  12496u
#define PWN_T_BINIDX  20704u

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
};
