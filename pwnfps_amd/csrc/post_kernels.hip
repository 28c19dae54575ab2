// post_kernels.hip -- framebuffer-side kernels: depth-driven blur
// (screen.h:69-123), SDL-sink upscale (screen.h:126-149), primitive probes.
#include <hip/hip_runtime.h>
#include <mutex>
#include "dev_math.h"
#include "tables.h"

// ---------------------------------------------------------------- blur -----
// The reference walks each row left to right, 4 pixels at a time, drawing 32
// LCG values per group (x then y offset for tap i of pixel j, i-major) from a
// per-row seed cy*cy+415135.  One thread takes one 4-pixel group; its seed is
// the row seed advanced by 32*g draws, done with the precomputed affine
// skip-ahead (A_g, C_g): s -> (A_g*s + C_g) mod 2^31  (the LCG is affine mod
// 2^31 because of the & 0x7FFFFFFF).  Reads: 16 B of depth, 16 scattered 4-B
// taps from the pre-blur frame; write: one 16-B store.
struct pwn_blur_params
{
	int w, h, y0, y1;
	int groups;                    // w / 4 (screen.h:91: cx < dimx-3)
	const uint32_t *pre;           // full pre-blur frame ("tsbuf")
	const float *zbuf;             // full frame depth
	uint32_t *out;                 // full frame
	const uint2 *skip;             // groups x (A_g, C_g)
	// row tiling with a bounded exchange: only rows [avail_y0, avail_y1) of `pre` hold this
	// frame; a tap that lands outside them makes the kernel add to *miss (the caller then
	// repeats the strip with the whole frame present).  miss == NULL: every row is valid.
	int avail_y0, avail_y1;
	uint32_t *miss;
	int tile_h, tile_w, batch;     // a workgroup's tile of output pixels (rows, columns) and the form of its staging loop: pwn_i_launch_blur picks
	                               // them by the size of the launch, pwn_launch_blur has the instantiations
	// row tiling with moving cuts: the trace launch in front of this one on the stream added up what its strip cost
	// in *cost_acc (pwn_trace_params.cost_word); this launch moves the sum to *cost_out, the word that travels with
	// the frame, and clears the accumulator for the stream's next trace.  Both NULL otherwise.
	uint32_t *cost_acc, *cost_out;
	uint32_t cost_mul, cost_div;   // ... scaled on the way: the resident grid over the grid the trace ran with (PWN_OPT_TRACE_ROOM), so that ranks with and without room compare
};

__device__ __forceinline__ uint32_t avg_u8x4(uint32_t a, uint32_t b)
{
	// _mm_avg_epu8: per byte (p + q + 1) >> 1 -- which is what v_lerp_u8 computes with a round bit of 1 in
	// every byte of its third operand (one instruction; as (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7f) it was five)
	return __builtin_amdgcn_lerp(a, b, 0x01010101u);
}

// screen.h:101-106: a tap coordinate goes float -> int with cvttss2si (INT_MIN for NaN and for anything
// outside int32) and is then clamped to [0, hi - 1].  v_cvt_i32_f32 truncates the same way but saturates (and
// gives 0 for NaN); the one input class where that ends differently is f >= 2^31 (INT_MAX, which the clamp
// would send to hi - 1 where the reference ends at 0).  Adding 1 wraps INT_MAX to INT_MIN; the clamp is then
// done one up, [1, hi], as one v_med3_i32.
__device__ __forceinline__ int blur_coord(float f, int hi)
{
	int t, r;
	asm("v_cvt_i32_f32 %0, %1" : "=v"(t) : "v"(f));
	t = (int)((uint32_t)t + 1u);
	asm("v_med3_i32 %0, %1, 1, %2" : "=v"(r) : "v"(t), "v"(hi));
	return r - 1;
}

// The same, left one up: [1, hi].  The tiled kernel folds the "- 1" into the constants the coordinate meets.
__device__ __forceinline__ int blur_coord1(float f, int hi)
{
	int t, r;
	asm("v_cvt_i32_f32 %0, %1" : "=v"(t) : "v"(f));
	t = (int)((uint32_t)t + 1u);
	asm("v_med3_i32 %0, %1, 1, %2" : "=v"(r) : "v"(t), "v"(hi));
	return r;
}

// (the row LCG runs on the doubled state: lcg2_fs, dev_math.h)

__global__ void __launch_bounds__(256)
pwn_blur_kernel(pwn_blur_params P)
{
	int g = blockIdx.x * blockDim.x + threadIdx.x;
	int cy = P.y0 + blockIdx.y;
	if(g >= P.groups || cy >= P.y1) return;

	uint32_t seed = (uint32_t)cy * (uint32_t)cy + 415135u;
	uint2 ac = P.skip[g];
	if(g > 0) seed = (ac.x * seed + ac.y) & 0x7FFFFFFFu;

	const float fstr = 0.002f * (float)P.h;
	int cx = g * 4;
	size_t row = (size_t)cy * (size_t)P.w;
	float4 zv = *(const float4 *)(P.zbuf + row + cx);
	float z[4] = { zv.x - 1.0f, zv.y - 1.0f, zv.z - 1.0f, zv.w - 1.0f };
	uint32_t tap[4][4];
#pragma unroll
	for(int i = 0; i < 4; i++)
	{
#pragma unroll
		for(int j = 0; j < 4; j++)
		{
			// screen.h:101-102: int + float, truncated toward zero on assignment
			float fx = (float)(cx + j) + (lcg_fs(seed) * fstr) * z[j];
			float fy = (float)cy + (lcg_fs(seed) * fstr) * z[j];
			// cvttss2si yields INT_MIN for NaN / out-of-range; then the clamps
			// of screen.h:103-106 send it to 0
			int x = (fx >= -2147483648.0f && fx < 2147483648.0f) ? (int)fx : INT32_MIN;
			int y = (fy >= -2147483648.0f && fy < 2147483648.0f) ? (int)fy : INT32_MIN;
			x = max(x, 0); y = max(y, 0);
			x = min(x, P.w - 1); y = min(y, P.h - 1);
			tap[i][j] = P.pre[(size_t)y * (size_t)P.w + (size_t)x];
		}
	}
	uint4 o;
	o.x = avg_u8x4(avg_u8x4(tap[0][0], tap[1][0]), avg_u8x4(tap[2][0], tap[3][0]));
	o.y = avg_u8x4(avg_u8x4(tap[0][1], tap[1][1]), avg_u8x4(tap[2][1], tap[3][1]));
	o.z = avg_u8x4(avg_u8x4(tap[0][2], tap[1][2]), avg_u8x4(tap[2][2], tap[3][2]));
	o.w = avg_u8x4(avg_u8x4(tap[0][3], tap[1][3]), avg_u8x4(tap[2][3], tap[3][3]));
	*(uint4 *)(P.out + row + cx) = o;
}

// Tiled form (the one launched): a workgroup owns BLUR_TW x BLUR_TH output
// pixels and first copies that rectangle plus a BLUR_HALO border of the
// pre-blur frame into LDS (coalesced 16-B loads).  A tap that lands inside the
// staged rectangle - at 4K on the reference level most do: |offset| <=
// 0.002*h*|depth-1|, 82 % of the pixels stay within 16 - is an LDS read; any other tap
// falls back to the global load of the plain kernel above, so the result does
// not depend on the tile shape.  Why: 16 scattered 4-byte taps per thread keep
// the vector L1 busy with one cache-line lookup per lane (the plain kernel is
// bound by that, not by HBM); LDS serves the same gather an order of
// magnitude faster and the staging adds only (1 + 2*HALO/TW)(1 + 2*HALO/TH)
// coalesced reads per output pixel.
// The tile (BLUR_TW x BLUR_TH) and the form of the staging loop (BATCH) are template parameters; pwn_i_launch_blur
// (pwn_api.cpp) picks them.  Rounds 1-2 ran 128 x 32 tiles (1024 threads, 42 KB of LDS: the least staging per output
// pixel and the fastest launch BY ITSELF); judged by the frame rate with the next frame's trace grid on the chip
// beside it, small workgroups win -- 32 x 32 (256 threads, 17 KB) on wide frames, 128 x 16 on narrow ones
// (profiles/r3_blur_sweep.txt).
#ifndef BLUR_HALO
#define BLUR_HALO 16        // measured 8 / 16 / 24 / 32: 45.6 / 45.4 / 46.6 / 48.0 us at 4K (less staging beats fewer fall-backs)
#endif
#define BLUR_LW (BLUR_TW + 2 * BLUR_HALO)          // staged columns
#define BLUR_LH (BLUR_TH + 2 * BLUR_HALO)          // staged rows
#define BLUR_PITCH (BLUR_LW + (BATCH == 2 ? 0 : 4)) // words; +4 keeps rows 16-B aligned and off one bank (direct-to-LDS staging: rows back to back)
#define BLUR_THREADS (BLUR_TW / 4 * BLUR_TH)       // one thread per 4-pixel group

template<bool CHECK, int BLUR_TW, int BLUR_TH, int BATCH>
__global__ void __launch_bounds__(BLUR_THREADS)
pwn_blur_tiled_kernel(pwn_blur_params P)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
	// Workgroup w runs on XCD w % 8 (round-robin dispatch) and every XCD has its
	// own L2.  Neighbouring tiles share their halos, so XCD k takes the k-th
	// contiguous eighth of the tiles in row-major order: the halo re-reads then
	// hit that XCD's L2 instead of each going out to the Infinity Cache.
	// (Placement only changes speed, never results.)
	if(P.cost_acc != NULL && blockIdx.x == 0 && threadIdx.x == 0) { *P.cost_out = (uint32_t)((unsigned long long)*P.cost_acc * P.cost_mul / P.cost_div); *P.cost_acc = 0u; }
	const int tiles_x = (P.w + BLUR_TW - 1) / BLUR_TW;
	const int ntiles = tiles_x * ((P.y1 - P.y0 + BLUR_TH - 1) / BLUR_TH);
	const int per_xcd = (ntiles + 7) >> 3;
	const int t = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
	if(t >= ntiles) return;
	const int x0 = (t % tiles_x) * BLUR_TW, y0 = P.y0 + (t / tiles_x) * BLUR_TH;
	const int lx0 = x0 - BLUR_HALO, ly0 = y0 - BLUR_HALO;

	// this thread's 4-pixel group
	const int g = (x0 >> 2) + (int)(threadIdx.x % (BLUR_TW / 4));
	const int cy = y0 + (int)(threadIdx.x / (BLUR_TW / 4));
	const bool mine = g < P.groups && cy < P.y1;
	uint2 ac;
	float4 zv;
	constexpr int NV = BLUR_LH * (BLUR_LW / 4);                  // uint4s of the staged rectangle
	// stage: uint4 = 4 pixels; w % 4 == 0 and lx0 % 4 == 0, so a uint4 is inside the frame or outside
	if constexpr(BATCH == 0)
	{
		for(int i = threadIdx.x; i < BLUR_LH * (BLUR_LW / 4); i += BLUR_THREADS)
		{
			const int row = i / (BLUR_LW / 4), c4 = i - row * (BLUR_LW / 4);
			const int gy = ly0 + row, gx = lx0 + c4 * 4;
			if((unsigned)gy < (unsigned)P.h && (unsigned)gx < (unsigned)P.w)
				*(uint4 *)(tile + row * BLUR_PITCH + c4 * 4) = *(const uint4 *)(P.pre + (size_t)gy * (size_t)P.w + (size_t)gx);
		}
		__syncthreads();
		if(!mine) return;
		ac = P.skip[g];
		zv = *(const float4 *)(P.zbuf + (size_t)cy * (size_t)P.w + (size_t)(g * 4));
	}
	else if constexpr(BATCH == 2)
	{
		// (sweep builds only: measured, not shipped -- the launch by itself is as fast as with the loads through registers, the
		// frame rate on two streams 5.5 % LOWER at 4K, profiles/r3_blur_sweep.txt)
		// Staging without registers: global_load_lds_dwordx4 (gfx950) writes a wave's 64 x 16 bytes straight into 1 KB of LDS,
		// lane after lane from M0's base, from per-lane global addresses -- the tile's rows lie back to back for it (pitch =
		// the staged width), uint4 number i of the tile comes from lane i % 64 of load i / 64.  Addresses are clamped into
		// the frame; what lies outside is loaded from there and never read (tap coordinates are clamped into the frame).
		const int gq = mine ? g : 0, cyq = mine ? cy : P.y0;
		ac = P.skip[gq];
		zv = *(const float4 *)(P.zbuf + (size_t)cyq * (size_t)P.w + (size_t)(gq * 4));
		constexpr int NL = (NV + 63) / 64;                         // wave-loads in all
		constexpr int NW = BLUR_THREADS / 64;
		const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
#pragma unroll
		for(int k = 0; k < (NL + NW - 1) / NW; k++)
		{
			const int l = wave + k * NW;                           // wave-uniform
			if(l < NL)
			{
				const int i = min(l * 64 + lane, NV - 1);
				const int row = i / (BLUR_LW / 4), c4 = i - row * (BLUR_LW / 4);
				const int gyc = min(max(ly0 + row, 0), P.h - 1), gxc = min(max(lx0 + c4 * 4, 0), P.w - 4);
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(P.pre + (size_t)gyc * (size_t)P.w + (size_t)gxc),
					(__attribute__((address_space(3))) void *)(tile + l * 256), 16, 0, 0);
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		asm volatile("" : "+v"(ac.x), "+v"(ac.y), "+v"(zv.x), "+v"(zv.y), "+v"(zv.z), "+v"(zv.w));
		__syncthreads();
		if(!mine) return;
	}
	else
	{
		// The shipped form: (1) the thread's own two loads from HBM (skip-ahead constants, depths) are asked for first and
		// used behind the barrier, and (2) all of its staging loads are in flight together, from an address clamped into the
		// frame (round 5: what lies outside IS stored, as the pixel its coordinates clamp to) -- as the plain loop above the compiler waits for every load before it
		// issues the next, and a workgroup has nothing else to do until its tile is there (profiles/r3_blur_staging.txt,
		// r3_blur_sweep.txt: batch 1 against batch 0).
		const int gq = mine ? g : 0, cyq = mine ? cy : P.y0;
		ac = P.skip[gq];
		zv = *(const float4 *)(P.zbuf + (size_t)cyq * (size_t)P.w + (size_t)(gq * 4));
		typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
		constexpr int NT = (NV + BLUR_THREADS - 1) / BLUR_THREADS;
		static_assert(NT >= 1 && NT <= 8, "the barrier below names its operands");
		u32x4 r[NT];
		int dst[NT];
		int edge[NT];
#pragma unroll
		for(int k = 0; k < NT; k++)
		{
			const int i = (int)threadIdx.x + k * BLUR_THREADS;
			const int row = i / (BLUR_LW / 4), c4 = i - row * (BLUR_LW / 4);
			const int gy = ly0 + row, gx = lx0 + c4 * 4;
			// every cell of the rectangle is stored: one outside the frame holds the pixel its coordinates CLAMP to (screen.h:103-106
			// clamp x and y one by one), so that a tap is looked up by its unclamped coordinates and clamped only where it leaves
			// the rectangle
			dst[k] = i < NV ? row * BLUR_PITCH + c4 * 4 : -4;
			const int gyc = min(max(gy, 0), P.h - 1), gxc = min(max(gx, 0), P.w - 4);
			r[k] = *(const u32x4 *)(P.pre + (size_t)gyc * (size_t)P.w + (size_t)gxc);
			edge[k] = gx < 0 ? 1 : (gx >= P.w ? 2 : 0);
		}
		// (all of them live at one point: the loads cannot be sunk to their stores one by one)
		if constexpr(NT == 2) asm volatile("" : "+v"(r[0]), "+v"(r[1]));
		if constexpr(NT == 3) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]));
		if constexpr(NT == 4) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
		if constexpr(NT == 5) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]));
		if constexpr(NT == 6) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]));
		if constexpr(NT == 7) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]));
		if constexpr(NT == 8) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
		for(int k = 0; k < NT; k++)
		{
			if(edge[k] == 1) r[k] = (u32x4){ r[k].x, r[k].x, r[k].x, r[k].x };
			if(edge[k] == 2) r[k] = (u32x4){ r[k].w, r[k].w, r[k].w, r[k].w };
		}
#pragma unroll
		for(int k = 0; k < NT; k++)
			if(dst[k] >= 0) *(u32x4 *)__builtin_assume_aligned(tile + dst[k], 16) = r[k];
		asm volatile("" : "+v"(ac.x), "+v"(ac.y), "+v"(zv.x), "+v"(zv.y), "+v"(zv.z), "+v"(zv.w));     // (asked for above, not down here)
		__syncthreads();
		if(!mine) return;
	}

	uint32_t seed = (uint32_t)cy * (uint32_t)cy + 415135u;
	if(g > 0) seed = (ac.x * seed + ac.y) & 0x7FFFFFFFu;

	const float fstr = 0.002f * (float)P.h;
	const int cx = g * 4;
	const size_t row = (size_t)cy * (size_t)P.w;
	const float z[4] = { zv.x - 1.0f, zv.y - 1.0f, zv.z - 1.0f, zv.w - 1.0f };
	const float fcx[4] = { (float)cx, (float)(cx + 1), (float)(cx + 2), (float)(cx + 3) };
	const float fcy = (float)cy;
	// Tap coordinates are kept ONE UP, as blur_coord1 leaves them ([1, w] x [1, h]); the constants they meet
	// absorb the difference: the staged rectangle's origin, the rows this rank holds, and the frame's base
	// address (moved back by one row and one pixel -- as an integer, no pointer outside the allocation is formed
	// until the offset of a real tap, >= w + 1, has been added).
	int lx1 = lx0 + 1, ly1 = ly0 + 1;
	asm volatile("" : "+s"(lx1), "+s"(ly1));       // (one scalar each: left to itself the compiler subtracts the tile's origin and adds the constant per tap)
	const uint32_t w1 = (uint32_t)P.w + 1u;
	const uintptr_t pre1 = (uintptr_t)P.pre - (uintptr_t)w1 * 4u;
	uint32_t t2 = seed << 1;                               // the LCG state doubled (lcg2_fs, dev_math.h)
	// (the frame's width and height once in vector registers: blur_coord1's v_med3_i32 takes its bound from one, and handed a
	// scalar the compiler copied it into a fresh register in front of every one of the 32 clamps)
	int vw = P.w, vh = P.h;
	asm volatile("" : "+v"(vw), "+v"(vh));
	uint32_t tap[4][4];
	bool missed = false;
	// Round 4: no "+ 1" per coordinate where it is not needed.  The one-up form exists for ONE input class, f >= 2^31 (above:
	// v_cvt_i32_f32 saturates to INT_MAX where cvttss2si gives INT_MIN), and a coordinate is column / row + r * fstr * (depth - 1)
	// with |r| <= 1 and fstr <= 66: while every depth of the wave's groups is below 10^6 in magnitude no coordinate comes near
	// 2^31, the conversion needs no correction, and the clamp is [0, hi - 1] directly (NaN and -inf end at 0 both ways).  One
	// wave-uniform test buys 32 additions per thread (4 % of the kernel's vector instructions).
	const bool tame = fabsf(z[0]) < 1.0e6f && fabsf(z[1]) < 1.0e6f && fabsf(z[2]) < 1.0e6f && fabsf(z[3]) < 1.0e6f;
	if(__builtin_expect(__ballot(!tame) == 0ull, 1))
	{
		const uint32_t w0 = (uint32_t)P.w;
		const uintptr_t pre0 = (uintptr_t)P.pre;
		int vw1 = P.w - 1, vh1 = P.h - 1;
		asm volatile("" : "+v"(vw1), "+v"(vh1));
		int lx = lx0, ly = ly0;
		asm volatile("" : "+s"(lx), "+s"(ly));
#pragma unroll
		for(int i = 0; i < 4; i++)
		{
#pragma unroll
			for(int j = 0; j < 4; j++)
			{
				// screen.h:101-106
				const float fx = fcx[j] + (lcg2_fs(t2) * fstr) * z[j];
				const float fy = fcy + (lcg2_fs(t2) * fstr) * z[j];
				if constexpr(BATCH == 1)
				{
					int cx_, cy_;
					asm("v_cvt_i32_f32 %0, %1" : "=v"(cx_) : "v"(fx));
					asm("v_cvt_i32_f32 %0, %1" : "=v"(cy_) : "v"(fy));
					if(CHECK)
					{
						int y0c;
						asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y0c) : "v"(cy_), "v"(vh1));
						missed |= (unsigned)(y0c - P.avail_y0) >= (unsigned)(P.avail_y1 - P.avail_y0);
					}
					// (unclamped: the rectangle's cells outside the frame hold what the clamp would have fetched)
					const unsigned tx = (unsigned)(cx_ - lx), ty = (unsigned)(cy_ - ly);
					const uint32_t *p = tile + (ty * BLUR_PITCH + tx);
					bool inside;
					if constexpr(BLUR_LW == 64 && BLUR_LH == 64) inside = (tx | ty) < 64u;
					else inside = tx < (unsigned)BLUR_LW && ty < (unsigned)BLUR_LH;
					if(!inside)
					{
						asm volatile("");
						int x0c, y0c;
						asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x0c) : "v"(cx_), "v"(vw1));
						asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y0c) : "v"(cy_), "v"(vh1));
						p = (const uint32_t *)(pre0 + ((uintptr_t)__umul24((unsigned)y0c, w0) + (uintptr_t)(unsigned)x0c) * 4u);
					}
					tap[i][j] = *p;
					continue;
				}
				int x0c, y0c, cx_, cy_;
				asm("v_cvt_i32_f32 %0, %1" : "=v"(cx_) : "v"(fx));
				asm("v_cvt_i32_f32 %0, %1" : "=v"(cy_) : "v"(fy));
				asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x0c) : "v"(cx_), "v"(vw1));
				asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y0c) : "v"(cy_), "v"(vh1));
				if(CHECK) missed |= (unsigned)(y0c - P.avail_y0) >= (unsigned)(P.avail_y1 - P.avail_y0);
				const unsigned tx = (unsigned)(x0c - lx), ty = (unsigned)(y0c - ly);
				const uint32_t *p = tile + (ty * BLUR_PITCH + tx);
				if(!(tx < (unsigned)BLUR_LW && ty < (unsigned)BLUR_LH))
				{
					asm volatile("");
					p = (const uint32_t *)(pre0 + ((uintptr_t)__umul24((unsigned)y0c, w0) + (uintptr_t)(unsigned)x0c) * 4u);
				}
				tap[i][j] = *p;
			}
		}
	}
	else
	{
#pragma unroll
	for(int i = 0; i < 4; i++)
	{
#pragma unroll
		for(int j = 0; j < 4; j++)
		{
			// screen.h:101-106
			const float fx = fcx[j] + (lcg2_fs(t2) * fstr) * z[j];
			const float fy = fcy + (lcg2_fs(t2) * fstr) * z[j];
			const int x1 = blur_coord1(fx, vw), y1 = blur_coord1(fy, vh);
			if(CHECK) missed |= (unsigned)(y1 - (P.avail_y0 + 1)) >= (unsigned)(P.avail_y1 - P.avail_y0);
			const unsigned tx = (unsigned)(x1 - lx1), ty = (unsigned)(y1 - ly1);
			// ONE load through a generic pointer: into the staged rectangle, or -- behind a real branch around the address
			// arithmetic of the rare case (left alone the compiler computes both addresses and selects) -- into the frame.
			// hipcc emits a flat load for it, through the LDS aperture or a global address.  (Forcing a ds_read plus a separate
			// global load serialises the taps on their waits: 46.2 against 45.5 us at 4K with 128 x 32 tiles; without the
			// branch -- every lane loading from the frame as well, then a select -- 5 % fewer vector instructions, all 32
			// loads in flight, and 1-2 us SLOWER.  Round 3: the LDS index is no longer clamped for the lanes that do not use
			// it -- one v_min_u32 per tap.)
			const uint32_t *p = tile + (ty * BLUR_PITCH + tx);
			if(!(tx < (unsigned)BLUR_LW && ty < (unsigned)BLUR_LH))
			{
				asm volatile("");
				p = (const uint32_t *)(pre1 + ((uintptr_t)__umul24((unsigned)y1, (unsigned)P.w) + (uintptr_t)(unsigned)x1) * 4u);
			}
			const uint32_t v = *p;
			tap[i][j] = v;
		}
	}
	}
	uint4 o;
	o.x = avg_u8x4(avg_u8x4(tap[0][0], tap[1][0]), avg_u8x4(tap[2][0], tap[3][0]));
	o.y = avg_u8x4(avg_u8x4(tap[0][1], tap[1][1]), avg_u8x4(tap[2][1], tap[3][1]));
	o.z = avg_u8x4(avg_u8x4(tap[0][2], tap[1][2]), avg_u8x4(tap[2][2], tap[3][2]));
	o.w = avg_u8x4(avg_u8x4(tap[0][3], tap[1][3]), avg_u8x4(tap[2][3], tap[3][3]));
	*(uint4 *)(P.out + row + cx) = o;
	if(CHECK) { if(missed) atomicAdd(P.miss, 1u); }
}

template<bool CHECK, int TW, int TH, int BATCH>
static hipError_t launch_blur_variant(const pwn_blur_params *P, hipStream_t stream)
{
	const int BLUR_TW = TW, BLUR_TH = TH;
	// (direct-to-LDS staging writes whole 1 KB pieces: room for the last one)
	const size_t lds = BATCH == 2 ? ((size_t)BLUR_LH * (BLUR_LW / 4) + 63) / 64 * 1024 : (size_t)BLUR_PITCH * BLUR_LH * sizeof(uint32_t);
	static bool lds_mark[64];
	static std::mutex lds_lock;      // contexts of several threads share the per-function attribute
	int dev = 0;
	(void)hipGetDevice(&dev);
	{
		std::lock_guard<std::mutex> g(lds_lock);
		bool &lds_set = lds_mark[dev & 63];
		if(!lds_set)
		{
			hipError_t e = hipFuncSetAttribute((const void *)pwn_blur_tiled_kernel<CHECK, TW, TH, BATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			if(e != hipSuccess) return e;
			lds_set = true;
		}
	}
	const int ntiles = ((P->w + BLUR_TW - 1) / BLUR_TW) * ((P->y1 - P->y0 + BLUR_TH - 1) / BLUR_TH);
	dim3 grid(((ntiles + 7) / 8) * 8);
	hipLaunchKernelGGL((pwn_blur_tiled_kernel<CHECK, TW, TH, BATCH>), grid, dim3(BLUR_THREADS), lds, stream, *P);
	return hipGetLastError();
}

template<int TW, int TH, int BATCH>
static hipError_t launch_blur_check(const pwn_blur_params *P, hipStream_t stream)
{
	return P->miss != NULL ? launch_blur_variant<true, TW, TH, BATCH>(P, stream) : launch_blur_variant<false, TW, TH, BATCH>(P, stream);
}

extern "C" hipError_t pwn_launch_blur(const pwn_blur_params *P, hipStream_t stream)
{
	if(P->groups <= 0 || P->y1 <= P->y0) return hipSuccess;
#ifdef PWN_BLUR_PLAIN
	dim3 grid((P->groups + 255) / 256, P->y1 - P->y0);
	hipLaunchKernelGGL(pwn_blur_kernel, grid, dim3(256), 0, stream, *P);
	return hipGetLastError();
#else
	// tile shape (tile_w x tile_h output pixels per workgroup) and staging form: chosen by pwn_i_launch_blur
	const int key = P->tile_w * 1000 + P->tile_h * 10 + P->batch;
	switch(key)
	{
#ifdef PWN_BLUR_SWEEP
		case 128162: return launch_blur_check<128, 16, 2>(P, stream);
		case 32322: return launch_blur_check<32, 32, 2>(P, stream);
		case 64322: return launch_blur_check<64, 32, 2>(P, stream);
		case 128320: return launch_blur_check<128, 32, false>(P, stream);
		case 128321: return launch_blur_check<128, 32, true>(P, stream);
		case 128160: return launch_blur_check<128, 16, false>(P, stream);
		case 128080: return launch_blur_check<128, 8, false>(P, stream);
		case 128081: return launch_blur_check<128, 8, true>(P, stream);
		case 64320: return launch_blur_check<64, 32, false>(P, stream);
		case 64160: return launch_blur_check<64, 16, false>(P, stream);
		case 64161: return launch_blur_check<64, 16, true>(P, stream);
		case 64081: return launch_blur_check<64, 8, true>(P, stream);
		case 64321: return launch_blur_check<64, 32, true>(P, stream);
		case 32161: return launch_blur_check<32, 16, true>(P, stream);
		case 16321: return launch_blur_check<16, 32, true>(P, stream);
		case 32641: return launch_blur_check<32, 64, true>(P, stream);
		case 64641: return launch_blur_check<64, 64, true>(P, stream);
		case 16641: return launch_blur_check<16, 64, true>(P, stream);
#endif
		case 128161: return launch_blur_check<128, 16, true>(P, stream);
		case 32321: return launch_blur_check<32, 32, true>(P, stream);
		default: return hipErrorInvalidValue;
	}
#endif
}

// ------------------------------------------------------ the units' order ----
// PWN_OPT_UNIT_ORDER: the hand-out order of the NEXT trace launch of this geometry from what every unit cost in the
// one that just ran (pwn_trace_params.unit_cost: 40 ns per count, written by the wave that traced the unit).  Queue q
// of the trace kernel holds the units a = q (mod 64); this kernel puts each queue's units in order of falling cost
// and writes perm[q * cap + t] = the unit that ticket t of queue q stands for.  One 256-thread workgroup per queue,
// a counting sort over 256 cost classes of 0.32 us (everything above 82 us in the dearest): a histogram in LDS, its
// prefix sums, a scatter -- units of one class come out in whatever order their atomics land, which is as good as any
// (a unit's cost is its wave's wall time, good to ~10 %).  (The first form, a bitonic sort of the whole queue in LDS,
// took 80 us at 4K -- 66 passes of a barrier each -- against the ~20 us the order can win: profiles/r4/unit_order_ab.txt.)
// Whatever the costs are, the output is a permutation of the queue's units, so the order can only change speed, never
// a pixel.  Replaces: the static schedule of screen.h:63-64.
__device__ __forceinline__ uint32_t order_class(uint32_t cost)
{
	const uint32_t b = cost >> 3;
	return 255u - (b > 255u ? 255u : b);              // class 0 = dearest
}

__global__ void __launch_bounds__(256)
pwn_order_kernel(const uint16_t *cost, uint32_t units, uint32_t cap, uint32_t *perm)
{
	__shared__ uint32_t hist[256], base[256];
	const uint32_t q = blockIdx.x, t = threadIdx.x;
	const uint32_t n = (units + PWN_QUEUES - 1u - q) / PWN_QUEUES;          // units of this queue: q, q + 64, ...
	hist[t] = 0u;
	__syncthreads();
	for(uint32_t i = t; i < n; i += 256u) atomicAdd(&hist[order_class(cost[i * PWN_QUEUES + q])], 1u);
	__syncthreads();
	// exclusive prefix sums over the 256 classes (Hillis-Steele, eight rounds)
	uint32_t v = hist[t];
	base[t] = v;
	__syncthreads();
	for(uint32_t d = 1u; d < 256u; d <<= 1)
	{
		const uint32_t add = t >= d ? base[t - d] : 0u;
		__syncthreads();
		base[t] += add;
		__syncthreads();
	}
	const uint32_t excl = base[t] - v;
	__syncthreads();
	base[t] = excl;
	__syncthreads();
	for(uint32_t i = t; i < n; i += 256u)
	{
		const uint32_t pos = atomicAdd(&base[order_class(cost[i * PWN_QUEUES + q])], 1u);
		perm[q * cap + pos] = i * PWN_QUEUES + q;
	}
}

extern "C" hipError_t pwn_launch_order(const uint16_t *cost, uint32_t units, uint32_t cap, uint32_t *perm, hipStream_t stream)
{
	hipLaunchKernelGGL(pwn_order_kernel, dim3(PWN_QUEUES), dim3(256), 0, stream, cost, units, cap, perm);
	return hipGetLastError();
}

// -------------------------------------------------------------- upscale ----
// screen.h:126-149: every source pixel becomes a scale x scale block.  The
// reference's destination pointer advances w*scale per source row plus
// pitch*(scale-1) (screen.h:132,138-139), so source row py starts at word
// py*(w*scale + pitch*(scale-1)); that equals py*scale*pitch when
// pitch == w*scale (what SDL gives it) and packs rows tighter otherwise.
// Kept as is.  One thread per destination pixel, stores coalesced along x.
__global__ void __launch_bounds__(256)
pwn_upscale_kernel(const uint32_t *src, uint32_t *dst, int w, int h, int scale, int pitch)
{
	int dx = blockIdx.x * blockDim.x + threadIdx.x;
	if(dx >= w * scale) return;
	size_t rowadv = (size_t)w * scale + (size_t)pitch * (scale - 1);
	// grid.y is capped at 65535 rows; taller surfaces are walked in that stride
	for(int dy = blockIdx.y; dy < h * scale; dy += gridDim.y)
	{
		int py = dy / scale, y = dy - py * scale;
		dst[(size_t)py * rowadv + (size_t)y * pitch + dx] = src[(size_t)py * (size_t)w + (dx / scale)];
	}
}

extern "C" hipError_t pwn_launch_upscale(const uint32_t *src, uint32_t *dst, int w, int h, int scale, int pitch, hipStream_t stream)
{
	const long long rows = (long long)h * scale;
	dim3 grid((w * scale + 255) / 256, (unsigned)(rows > 65535 ? 65535 : rows));
	hipLaunchKernelGGL(pwn_upscale_kernel, grid, dim3(256), 0, stream, src, dst, w, h, scale, pitch);
	return hipGetLastError();
}

// ---------------------------------------------------------- table upload ----
// The per-frame table upload (level_prepare_render, main.c:95: ~18 KB) as a kernel that reads
// the pinned staging buffer over PCIe itself.  A hipMemcpyAsync would queue on a DMA engine
// behind the 33 MB copy of the previous frame to the host, and the next trace launch waits
// for its tables: measured 0.83 ms per 4K frame with frames in flight against 0.60 with this.
__global__ void __launch_bounds__(256)
pwn_upload_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16)
{
	for(int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

extern "C" hipError_t pwn_launch_upload(const void *h_pinned_src, void *d_dst, size_t bytes, hipStream_t stream)
{
	const int n16 = (int)((bytes + 15) / 16);
	if(n16 <= 0) return hipSuccess;
	hipLaunchKernelGGL(pwn_upload_kernel, dim3((n16 + 255) / 256 > 8 ? 8 : (n16 + 255) / 256), dim3(256), 0, stream,
		(const uint4 *)h_pinned_src, (uint4 *)d_dst, n16);
	return hipGetLastError();
}

// The two words of every rank's strip (pwn_tiled.cpp: the miss word and the cost word) from where the exchange left them
// to pinned host memory, by a kernel: two DMA copies of 8 bytes each cost the comm stream 20-50 us per frame, and the
// host, which reads the words when it takes the frame, waited for them (profiles/r4/host_bound.txt).
__global__ void pwn_words_kernel(const uint32_t *__restrict__ all, const uint32_t *__restrict__ own, uint32_t *h, int world)
{
	for(int i = threadIdx.x; i < 2 * world + 2; i += 64) h[i] = i < 2 * world ? all[i] : own[i - 2 * world];
}

extern "C" hipError_t pwn_launch_words(const uint32_t *d_all, const uint32_t *d_own, uint32_t *h_pinned_dst, int world, hipStream_t stream)
{
	hipLaunchKernelGGL(pwn_words_kernel, dim3(1), dim3(64), 0, stream, d_all, d_own, h_pinned_dst, world);
	return hipGetLastError();
}

// --------------------------------------------------------------- probes ----
// Device-side known-answer access to the arithmetic primitives (pwnhip.h
// PWN_PROBE_*).  tabs = the rcp+rsqrt part of the blob (2 x 2048 u16).
__global__ void pwn_probe_kernel(int op, const uint32_t *in, uint32_t *out, int n, const uint16_t *tabs)
{
	__shared__ uint16_t t[4096];
	for(int i = threadIdx.x; i < 4096; i += blockDim.x) t[i] = tabs[i];
	__syncthreads();
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	const uint16_t *rcp = t, *rsq = t + 2048;
	switch(op)
	{
		case 0: out[i] = __float_as_uint(tab_rcp(rcp, __uint_as_float(in[i]))); break;
		case 1: out[i] = __float_as_uint(tab_rsqrt(rsq, __uint_as_float(in[i]))); break;
		case 2: out[i] = __float_as_uint(glibc_sincosf(__uint_as_float(in[i]), 0)); break;
		case 3: out[i] = __float_as_uint(glibc_sincosf(__uint_as_float(in[i]), 1)); break;
		case 4: out[i] = __float_as_uint(glibc_expf(__uint_as_float(in[i]))); break;
		case 5: out[i] = __float_as_uint(sqrtf(__uint_as_float(in[i]))); break;
		case 6: out[i] = __float_as_uint(__uint_as_float(in[2 * i]) / __uint_as_float(in[2 * i + 1])); break;
		case 7: out[i] = col_pack(v4_set(__uint_as_float(in[4 * i]), __uint_as_float(in[4 * i + 1]),
			__uint_as_float(in[4 * i + 2]), __uint_as_float(in[4 * i + 3]))); break;
		case 8: { uint32_t t = in[i] << 1; out[i] = __float_as_uint(lcg2_fs(t)); break; }        // (the form the kernels use: doubled state)
		case 9: out[i] = __float_as_uint(glibc_sincosf_both(__uint_as_float(in[i])).x); break;
		case 10: out[i] = __float_as_uint(glibc_sincosf_both(__uint_as_float(in[i])).y); break;
		default: out[i] = 0; break;
	}
}

extern "C" hipError_t pwn_launch_probe(int op, const uint32_t *in, uint32_t *out, int n, const uint16_t *tabs, hipStream_t stream)
{
	hipLaunchKernelGGL(pwn_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, in, out, n, tabs);
	return hipGetLastError();
}
