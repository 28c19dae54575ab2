// trace_kernel.hip -- the per-pixel portal ray-march on gfx950 (CDNA4).
//
// Replaces trace_ray_prelude (screen.h:1-28) + trace_ray / trace_ray_through /
// trace_hit_wall / trace_hit_bounce (trace.h) of the reference.
//
// Shape: persistent 256-thread workgroups (4 wave64), exactly as many as are
// resident at once.  Each workgroup copies the level blob (rcp/rsqrt tables,
// per-cell word, portals, per-cell sphere lists, spheres: tables.h) HBM -> LDS
// once; then every wave pulls 16x4-pixel units of the row strip from work
// queues until none are left (rays differ in cost by an order of magnitude
// across a frame; see the kernel body), one thread per pixel.
// The reference's recursion (depth <= REFLECT) is a loop over at most three ray
// segments; the composites of trace.h:91-101 are applied on unwinding.
// Output: BGRA8 colour + fp32 depth, row-major, 4-byte stores.
//
// No MFMA: this is a branchy DDA, not a contraction.  HBM traffic is the two
// output planes only (8 B / pixel); everything the inner loop reads is in LDS.
// Its time follows the number of wave-instructions issued, so the code
// is organised for few instructions per cell step and few scalar mask sequences:
//   - one walk loop with one exit; the cell class is a bit test on ONE LDS
//     word per step (class flags + sphere-list offset, 65x65 clamp-free grid),
//   - the room body (1-high and 2-high share it) is written with selects,
//   - the per-tile ray add-chain is built systolically with DPP row shifts,
//   - rcp/rsqrt table hits cost an index, a ds_read_u16 and a subtract,
//   - cameras without w components (the usual case) take a 3-lane path that
//     is arithmetically identical to the 4-lane SSE code (see HAS_W below),
//   - no device function calls (dev_math.h) and no SLP packing (Makefile).
#include <hip/hip_runtime.h>
#include <mutex>
#include "trace_common.h"


// One pixel = trace_ray(0, ...) of screen.h:22-24 with the recursion unrolled.
template<bool COUNT, bool HAS_W, bool INL>
__device__ __forceinline__ void trace_pixel(const Lds &L, float sec_current, uint32_t seed,
	Vec<HAS_W> from, Vec<HAS_W> iray, float &out_x, float &out_y, float &out_z, float &out_w,
	float *zpix, Counters &cnt)
{
	typedef Vec<HAS_W> V;
	// icol (screen.h:24).  Its w lane, and the w lane of every surface colour, is
	// x * 0.0f (COL_* have a = 0, defs.h:17-19; spheres get b,g,r only, script.h:30-32):
	// +-0 for any finite input, and the sign of a zero never reaches a pixel, so the
	// w lanes of icol and of the composite stack are not kept.
	// The composite stack (reflectivity, fog, colour of the surfaces the ray bounced off), two entries, as a
	// shift register: a bounce moves the top entry down and writes the new one on top (plain moves; indexing by
	// the segment number made ten selects of it).  The top entry's colour IS the next segment's icol
	// (trace.h:90), so it starts as 1,1,1 and icol needs no registers of its own.
	// (an entry is written when its surface is met and read only by a lane that bounced that often: the entries start as whatever
	// their registers hold -- an asm statement without instructions, volatile so that two of them are not taken for one value)
	float st_refl0, st_refl1, st_fog0, st_fog1;
	float sc0x = 1.0f, sc0y = 1.0f, sc0z = 1.0f;
	float sc1x, sc1y, sc1z;
	asm volatile("" : "=v"(st_refl0)); asm volatile("" : "=v"(st_refl1)); asm volatile("" : "=v"(st_fog0)); asm volatile("" : "=v"(st_fog1));
	asm volatile("" : "=v"(sc1x)); asm volatile("" : "=v"(sc1y)); asm volatile("" : "=v"(sc1z));
#define icx sc0x
#define icy sc0y
#define icz sc0z
	// Every lane still in the segment loop is on the same segment, so the segment number `seg` is one scalar
	// for the wave (tests on it are scalar branches); `depth`, the number of surfaces a pixel's ray bounced off,
	// is per lane and set where the lane leaves the loop.
	int depth, seg = 0;
	asm volatile("" : "=v"(depth));
	float vx, vy, vz, vw;
	asm volatile("" : "=v"(vx)); asm volatile("" : "=v"(vy)); asm volatile("" : "=v"(vz));
	// The colour's w lane is not carried: every surface colour has w = 0 (defs.h:16-18, spheres'
	// col.w is never set), so col.w = diffuse * (icol.w * 0) is 0 -- unless the shading factor is
	// not finite (a ray that went through 1/0 in a ramp, trace.h:461), when it is NaN and so is
	// everything composited from it.  One accumulator stands for that lane: w_acc += factor * 0 stays
	// +0 until a factor is NaN or inf and is NaN from then on (a register rather than a flag: a
	// lane-divergent bool carried across the walk loop costs three mask updates per iteration).
	float w_acc = 0.0f;

	// (carried from segment to segment: the position -- a segment starts where the one before ended, trace.h:86-89 -- and the
	// sphere candidate's fields behind aux_dist, which alone is reset per segment)
	V pos = from;
	float aux_diff = 0.0f;
	uint32_t aux_idx = 0;
	V aux_pos, aux_norm;
	aux_pos.x = aux_pos.y = aux_pos.z = aux_pos.w = 0.0f;
	aux_norm = aux_pos;
#pragma unroll 1
	for(;;)
	{
		//@R p_setup
		RG(RG_SEG);
		seg = __builtin_amdgcn_readfirstlane(seg);
		// ------------------------------------------------ trace.h:186-248
		float cdist = 0.0f, fog = 0.0f;
		// nearest sphere candidate (trace.h:193-199): distance, hit point, which sphere and
		// its diffuse factor; normal, colour and reflectivity are rebuilt from these when
		// the hit is committed
		// aux_dist: the reference's "none yet" value -1 (trace.h:200) is kept as +inf here, so that
		// "a candidate exists and lies behind us" is one comparison; a candidate whose distance is
		// exactly -1.0f counts as none there and is stored as +inf here too
		// (the candidate's other fields are read only behind aux_dist: they keep what the segment before left in them -- declared in
		// front of the loop -- instead of five moves per ray)
		float aux_dist = __builtin_inff();
		if(COUNT) cnt.rays++;

		// The set-up's table reads (1/sqrt, the first cell's word, three reciprocals) are each issued ahead of work that does not
		// need them -- the compiler leaves an LDS read where the source has it, directly in front of its use, and sinks one that
		// only a branch uses into that branch: -0.3 % launch time at 4K, -0.5 % on synth64 (profiles/r5/sphere_lists_ab.txt).
		const uint32_t lb = __float_as_uint(dot3<HAS_W>(iray, iray));
		const uint32_t rsq_e = tab_rsqrt_entry(L.rsq, lb);
		int cx = (int)pos.x, cz = (int)pos.z;
		// signs of the UN-normalised input (trace.h:225-227)
		int gx = (iray.x < 0.0f ? -1 : 1);
		int gz = (iray.z < 0.0f ? -1 : 1);
		const bool gyp = !(iray.y < 0.0f);          // gy > 0
		// cell coordinates and steps in the packed form the walk uses (trace_common.h)
		uint32_t cxz = cxz_pack_start(cx, cz), sx = (uint32_t)gx & 0xffffu, sz = (uint32_t)gz << 16;
		uint32_t cw = cellword_pk(L, cxz);
		float wx = pos.x - (float)cx, wy = pos.y, wz = pos.z - (float)cz;
		const int ldy = gyp ? FYP : FYN;
		int ldx = (gx < 0 ? FXN : FXP), ldz = (gz < 0 ? FZN : FZP);
		// util.h:32-46
		V ray = vscale<HAS_W>(tab_rsqrt_finish(lb, rsq_e), iray);
		// trace.h:220-222 clamp |ray| to EPSILON, trace.h:230-231 take the three reciprocals.  A normalised ray
		// has all three magnitudes in [EPSILON, 2^126) unless it is degenerate: ONE test on the bit patterns
		// (a NaN's is above every number's) and one wave-uniform branch; then nothing is clamped and all three
		// reciprocals are the one-subtract table path (one LDS round trip, under way while the fractions are turned:
		// a table index is in range whatever the bits are)
		float iax, iaz, iay_;
		{
			const uint32_t EPSB = __float_as_uint(EPS);
			const uint32_t bx = __float_as_uint(ray.x) & 0x7fffffffu, by = __float_as_uint(ray.y) & 0x7fffffffu,
				bz = __float_as_uint(ray.z) & 0x7fffffffu;
			uint32_t ex = rcp_entry(L.rcp, bx), ey = rcp_entry(L.rcp, by), ez = rcp_entry(L.rcp, bz);
			const bool plain = max(max(bx - EPSB, by - EPSB), bz - EPSB) < 0x7e800000u - EPSB;
			if(ray.x >= 0.0f) wx = 1.0f - wx;
			if(ray.y >= 0.0f) wy = 1.0f - wy;
			if(ray.z >= 0.0f) wz = 1.0f - wz;
			// (statements the compiler may not reorder: the fractions first, then the wait for the table.  Not in the 4-lane
			// variant: its ordered form would need 12 bytes of scratch per lane for them)
			if constexpr(!HAS_W)
			{
				asm volatile("" : "+v"(wx), "+v"(wy), "+v"(wz));
				asm volatile("" : "+v"(ex), "+v"(ey), "+v"(ez));
			}
			if(__builtin_expect(__ballot(!plain) == 0ull, 1))
			{
				iax = __uint_as_float(ex - (bx & 0x7f800000u)); iay_ = __uint_as_float(ey - (by & 0x7f800000u));
				iaz = __uint_as_float(ez - (bz & 0x7f800000u));
			}
			else
			{
				//@R p_setup_slow
				RG(RG_SETUP_SLOW);
				// (the fractions above were turned by the unclamped signs: the clamp keeps ">= 0" as it was -- -0 counts as +)
				if(fabsf(ray.x) < EPS) ray.x = (ray.x < 0.0f ? -EPS : EPS);
				if(fabsf(ray.y) < EPS) ray.y = (ray.y < 0.0f ? -EPS : EPS);
				if(fabsf(ray.z) < EPS) ray.z = (ray.z < 0.0f ? -EPS : EPS);
				iax = tab_rcp(L.rcp, fabsf(ray.x)); iay_ = tab_rcp(L.rcp, fabsf(ray.y)); iaz = tab_rcp(L.rcp, fabsf(ray.z));
			}
		}
		//@R p_setup
		const float iay = iay_;
		wx *= iax; wy *= iay; wz *= iaz;
		// the "-part of a two-level room shifts the floor by one: wy moves by -+iay
		// (trace.h:345-349,381-385); iay_dn is the amount added when stepping DOWN into it
		const float iay_dn = gyp ? iay : -iay;
		uint32_t iay_up_bits = gyp ? __float_as_uint(iay) : 0u;         // +iay when looking up, else +0
		asm volatile("" : "+v"(iay_up_bits));        // keep it a register, not a select on gyp per step
		int ldir = FYN;
		int ev = EV_NONE, base = BASE_ROOM_Y;

		// ------------------------------------------------ trace.h:250-675 (trace_walk.inc)
		//@R p_walk_ctl
		// How the walk loop ends: ev per step -- two selects in the room body (sphere hit / floor-ceiling / neither), the
		// step limit (trace.h:250) folded into it with three more VALU instructions, one compare of ev for the exit.
		// Two other forms were built and measured in round 3 (profiles/r3_walk_exit.txt; the code is in commit 2b36426):
		// the step limit as a scalar count with a branch of its own (-3 VALU, +2 scalar per step: +3.9 % time at 4K), and a
		// lane mask `done` = hit || ymin with WHICH of the two read off cdist against aux_dist after the walk (-4 VALU,
		// +10 scalar mask instructions per step as compiled: +5.3 %).  Scalar instructions are not free here.
		int maxsteps = 1000;
#pragma unroll 1
		do
		{
#include "trace_walk.inc"
			// trace.h:250,677: out of steps
			if(--maxsteps == 0 && ev == 0) ev = EV_EXHAUSTED;
		} while(ev == 0);
		// what the ray ended on is read back from the register: without this the compiler keeps
		// "ev == EV_EXHAUSTED" as a lane mask that it updates in every iteration of the walk
		// (5 of ~85 instructions per step)
		asm volatile("" : "+v"(ev));

		//@R p_post
		if(ev == EV_EXHAUSTED)
		{
			//@R p_exhausted
			RG(RG_EXHAUSTED);
			// trace.h:677-678: out of steps -- the walked ray is the colour
			if(COUNT) cnt.exhausted++;
			vx = ray.x; vy = ray.y; vz = ray.z; vw = HAS_W ? ray.w : 0.0f;
			depth = seg;
			break;
		}
		//@R p_post
		if(ev == EV_WALL && base == BASE_ROOM_Y) { ldir = ldy; base = (gyp ? BASE_CEIL : BASE_FLOOR); }
		// zbuf = the PRIMARY ray's hit distance (trace.h:102-105); a primary ray that ran out of steps
		// leaves the old depth in place (trace.h:677)
		if(seg == 0) *zpix = (ev == EV_SPHERE ? aux_dist : cdist);

		// (the segment's colour is computed into the pixel's own registers: a segment that ends the ray has nothing to copy)
		float refl;
#define colx vx
#define coly vy
#define colz vz
		if(ev == EV_WALL)
		{
			//@R p_wall
			RG(RG_WALL);
			// trace.h:108-154, and the axis-aligned mirrors of trace.h:50-75.  Colour by wall class and what
			// the face does to the ray are two constant tables in LDS (tables.h PWN_T_FACES): three 16-byte
			// reads instead of two switch trees (which the compiler builds out of lane masks and branches)
			const pwn_f4 wc = L.faces[base];
			const pwn_f4 fa = L.faces[4 + 2 * ldir], fb = L.faces[5 + 2 * ldir];
			float diffuse = (ldir & 1) ? ray.z : ray.x;
			diffuse = ldir >= FYP ? ray.y : diffuse;
			diffuse = __uint_as_float(__float_as_uint(diffuse) ^ __float_as_uint(fb.w));      // -ray.c on the N faces
			if(diffuse < 0.0f) diffuse = 0.0f;
			const float amb = 0.1f;
			diffuse = (1.0f - amb) * diffuse + amb;
			colx = diffuse * (icx * wc.x); coly = diffuse * (icy * wc.y); colz = diffuse * (icz * wc.z);
			w_acc = __builtin_fmaf(diffuse, 0.0f, w_acc);
			refl = fa.w;
			// the mirror: flip the ray component along the face normal, step 0.001 off the surface (the other
			// axes add -0.0f, which changes nothing); the floor (FYN) takes the step here and its ray from
			// the rippled normal below
			ray.x = __uint_as_float(__float_as_uint(ray.x) ^ __float_as_uint(fa.x));
			ray.y = __uint_as_float(__float_as_uint(ray.y) ^ __float_as_uint(fa.y));
			ray.z = __uint_as_float(__float_as_uint(ray.z) ^ __float_as_uint(fa.z));
			pos.x += fb.x; pos.y += fb.y; pos.z += fb.z;
		}
		else
		{
			//@R p_sphere
			RG(RG_SPHERE);
			// trace.h:283-291 for the committed sphere
			// (inline records: aux_idx is the record's LDS address; which sphere it is of -- a byte offset -- is looked up here, once per hit)
			if constexpr(INL) aux_idx = L.recsph[(aux_idx - PWN_T_BINIDX) >> 4];
			const PWN_LDS pwn_f4 *sp = (const PWN_LDS pwn_f4 *)((const PWN_LDS unsigned char *)L.sph + aux_idx);      // (a byte offset)
			const pwn_f4 s0 = sp[0], s1 = sp[1];
			V d;
			d.x = aux_pos.x - s0.x; d.y = aux_pos.y - s0.y; d.z = aux_pos.z - s0.z;
			if constexpr(HAS_W) d.w = aux_pos.w - 1.0f; else d.w = 0.0f;
			aux_norm = vnormalise<HAS_W>(L.rsq, d);
			colx = aux_diff * s1.y; coly = aux_diff * s1.z; colz = aux_diff * s1.w;
			w_acc = __builtin_fmaf(aux_diff, 0.0f, w_acc);
			refl = s1.x;
			ldir = -1;
			pos = aux_pos;
		}

		//@R p_post
		// trace.h:3-7
		if(seg >= REFLECT_MAX || refl == 0.0f) { vw = 0.0f; depth = seg; break; }

		// trace.h:9-75
		if(ldir == FYN)
		{
			//@R p_floor
			RG(RG_FLOOR);
			const float pi = (float)3.14159265358979323846;
			float ang = (pi * 2.0f) * (
				(glibc_sincosf((pi * 0.5f) * pos.x, 0) + glibc_sincosf((pi * 0.5f) * pos.z, 1))
				+ sec_current);
			const float2 sc = glibc_sincosf_both(ang);
			V n; n.x = sc.x; n.y = 38.0f; n.z = sc.y; n.w = 0.0f;
			n = vnormalise<HAS_W>(L.rsq, n);
			float rmul = -2.0f * ((ray.x * n.x + ray.y * n.y) + ray.z * n.z);
			ray = vnormalise<HAS_W>(L.rsq, vadd<HAS_W>(vscale<HAS_W>(rmul, n), ray));
		}
		else if(ldir < 0)
		{
			//@R p_sphrefl
			RG(RG_SPHREFL);
			pos = vsub<HAS_W>(pos, vscale<HAS_W>(0.001f, ray));
			float rmul = -2.0f * ((ray.x * aux_norm.x + ray.y * aux_norm.y) + ray.z * aux_norm.z);
			ray = vnormalise<HAS_W>(L.rsq, vadd<HAS_W>(vscale<HAS_W>(rmul, aux_norm), ray));
		}

		//@R p_jitter
		RG(RG_JITTER);
		// trace.h:77-84: five draws, two discarded
		// (straight into the next segment's direction; the entry below the top is moved down only once there is one)
		iray.x = ray.x + lcg2_fs(seed) * REFLECT_BLUR_F;
		iray.y = ray.y + lcg2_fs(seed) * REFLECT_BLUR_F;
		lcg2_next(seed);
		iray.z = ray.z + lcg2_fs(seed) * REFLECT_BLUR_F;
		lcg2_next(seed);
		if constexpr(HAS_W) iray.w = ray.w;

		if(seg != 0) { st_refl1 = st_refl0; st_fog1 = st_fog0; sc1x = sc0x; sc1y = sc0y; sc1z = sc0z; }
		st_refl0 = refl; st_fog0 = fog; sc0x = colx; sc0y = coly; sc0z = colz;
#undef colx
#undef coly
#undef colz
		seg++;
	}

	//@R p_comp
	// trace.h:91-101, innermost first
	// the top of the stack is the last surface the ray bounced off, the entry below it the one before
	if(depth >= 1)
	{
		//@R p_comp1
		RG(RG_COMP1);
		const float r0 = st_refl0, q0 = 1.0f - st_refl0;
		vx = r0 * vx + q0 * sc0x; vy = r0 * vy + q0 * sc0y; vz = r0 * vz + q0 * sc0z; vw = r0 * vw;
		if(st_fog0 != 0.0f)
		{
			//@R p_comp1_fog
			RG(RG_COMP1_FOG);
			float f = glibc_expf(-0.6f * st_fog0, L.exp2), g = 1.0f - f;
			vx = f * vx + g; vy = f * vy + g; vz = f * vz + g; vw = f * vw + g;
		}
	}
	//@R p_comp
	if(depth >= 2)
	{
		//@R p_comp2
		RG(RG_COMP2);
		const float r1 = st_refl1, q1 = 1.0f - st_refl1;
		vx = r1 * vx + q1 * sc1x; vy = r1 * vy + q1 * sc1y; vz = r1 * vz + q1 * sc1z; vw = r1 * vw;
		if(st_fog1 != 0.0f)
		{
			//@R p_comp2_fog
			RG(RG_COMP2_FOG);
			float f = glibc_expf(-0.6f * st_fog1, L.exp2), g = 1.0f - f;
			vx = f * vx + g; vy = f * vy + g; vz = f * vz + g; vw = f * vw + g;
		}
	}
#undef icx
#undef icy
#undef icz
	//@R p_comp
	out_x = vx; out_y = vy; out_z = vz; out_w = vw + w_acc;
}

// Rounds 1..15 of the add chain of screen.h:12-18 (see the unit loop): in round k the lanes k..15 of every 16-lane row add
// rdx once more, so lane j ends with j adds on top of the row's start value -- the same sequence of fp32 additions for
// every pixel as the reference's "+= rdx" per pixel of the tile.  Written with the execution mask set by hand: a
// v_add_f32 under a mask issues at full rate, the DPP form of the same systolic chain (v[j] = v[j-1] + rdx, round 2) at
// half rate, and this is 45 of a unit's ~170 vector instructions.  All 64 lanes are active on entry (wave-uniform
// control flow); rdx is wave-uniform (kernel argument).
//@R k_unit
#define PWN_CHAIN_ROUND3(m) "s_mov_b32 exec_lo, " m "\n\ts_mov_b32 exec_hi, " m "\n\tv_add_f32 %0, %4, %0\n\tv_add_f32 %1, %5, %1\n\tv_add_f32 %2, %6, %2\n\t"
#define PWN_CHAIN_ROUND4(m) "s_mov_b32 exec_lo, " m "\n\ts_mov_b32 exec_hi, " m "\n\tv_add_f32 %0, %5, %0\n\tv_add_f32 %1, %6, %1\n\tv_add_f32 %2, %7, %2\n\tv_add_f32 %3, %8, %3\n\t"
#define PWN_CHAIN_ALL(R) R("0xfffefffe") R("0xfffcfffc") R("0xfff8fff8") R("0xfff0fff0") R("0xffe0ffe0") R("0xffc0ffc0") R("0xff80ff80") \
	R("0xff00ff00") R("0xfe00fe00") R("0xfc00fc00") R("0xf800f800") R("0xf000f000") R("0xe000e000") R("0xc000c000") R("0x80008000")
template<bool HAS_W> __device__ __forceinline__ void chain_rounds(Vec<HAS_W> &v, const Vec<HAS_W> &rdx)
{
	unsigned long long saved;
	if constexpr(HAS_W)
		asm volatile("s_mov_b64 %4, exec\n\t" PWN_CHAIN_ALL(PWN_CHAIN_ROUND4) "s_mov_b64 exec, %4"
			: "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "=&s"(saved) : "s"(rdx.x), "s"(rdx.y), "s"(rdx.z), "s"(rdx.w));
	else
		asm volatile("s_mov_b64 %3, exec\n\t" PWN_CHAIN_ALL(PWN_CHAIN_ROUND3) "s_mov_b64 exec, %3"
			: "+v"(v.x), "+v"(v.y), "+v"(v.z), "=&s"(saved) : "s"(rdx.x), "s"(rdx.y), "s"(rdx.z));
}

// ORDER: the launch writes what every unit cost its wave and / or hands its units out by a table (PWN_OPT_UNIT_ORDER, the
// wave log).  A template parameter, not a test of the two pointers: as dormant code -- two wave-uniform branches and a
// clock read per unit -- it cost launches that do not use it 2.5-3 % (profiles/r4/unit_order_dormant_cost.txt).
template<bool COUNT, bool HAS_W, bool ORDER, bool INL>
__global__ void __launch_bounds__(PWN_BLOCK, PWN_MIN_WAVES)
pwn_trace_kernel(pwn_trace_params P)
{
	//@R k_prologue
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

	// HBM -> LDS, 16 B per lane per trip
	{
		const uint4 *src = (const uint4 *)P.blob;
		uint4 *dst = (uint4 *)lds_raw;
		int n16 = (int)(P.blob_bytes >> 4);
		for(int i = threadIdx.x; i < n16; i += PWN_BLOCK) dst[i] = src[i];
		// one word behind the tables: the workgroup's share of pwn_trace_params.cost_word (bottom of the kernel)
		if(threadIdx.x == 0) *(uint32_t *)(lds_raw + ((P.blob_bytes + 15u) & ~15u)) = 0u;
	}
	__syncthreads();

	// (the tables are addressed from LDS address 0 on, trace_common.h; the launcher checks that this kernel has
	// no static LDS in front of the dynamic allocation)
	const Lds L = lds_tables(P.off_sph, P.off_recsph);

	typedef Vec<HAS_W> V;
	V rayb, rdx, rdy, from;
	rayb.x = P.rayb[0]; rayb.y = P.rayb[1]; rayb.z = P.rayb[2]; rayb.w = HAS_W ? P.rayb[3] : 0.0f;
	rdx.x = P.rdx[0]; rdx.y = P.rdx[1]; rdx.z = P.rdx[2]; rdx.w = HAS_W ? P.rdx[3] : 0.0f;
	rdy.x = P.rdy[0]; rdy.y = P.rdy[1]; rdy.z = P.rdy[2]; rdy.w = HAS_W ? P.rdy[3] : 0.0f;
	from.x = P.from[0]; from.y = P.from[1]; from.z = P.from[2]; from.w = HAS_W ? P.from[3] : 1.0f;

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l16 = lane & 15;

	Counters cnt = {};
	// PWN_OPT_WAVE_LOG: when was this wave resident (the GPU's constant 100 MHz clock)
	unsigned long long t_begin = 0ull;
	if(P.wave_log != NULL || P.cost_word != NULL) t_begin = __builtin_amdgcn_s_memrealtime();

	// Work distribution.  A unit is one wave64's 16 x 4 pixels (lane & 15 = column inside one
	// half of the 32-wide tile of screen.h:6-7 = one DPP row, lane >> 4 = row).  Rays differ in
	// cost by an order of magnitude from one part of a frame to another, and with a static split
	// (unit k to wave k mod #waves) the slowest wave ran 1.5x (level.txt) to 2.7x (synth256) as long as
	// the average one: the chip idled a third of the kernel's time.  So waves pull units:
	// PWN_QUEUES counters 128 B apart (same-address atomics serialise at ~4 ns; one counter for a
	// 4K frame's 130 k units would be the bottleneck, as an earlier attempt showed; 8 / 16 / 32 / 64
	// queues measured 0.402 / 0.394 / 0.388 / 0.383 ms at 4K), queue q holds
	// the units u = q (mod PWN_QUEUES); a wave drains its home queue, then helps with the others,
	// and asks for its next unit before it starts on the current one (the ~2 us round trip of
	// the atomic hides behind ~15 us of tracing).  The counters of the NEXT launch of this context
	// are cleared here (launches of a context are stream-ordered, include/pwnhip.h).
	const uint32_t units_x = ((uint32_t)P.w + 15u) >> 4;
	const uint32_t units = units_x * (((uint32_t)(P.y1 - P.y0) + 3u) >> 2);
	if(blockIdx.x == 0 && threadIdx.x < PWN_QUEUES) P.tickets_next[threadIdx.x * PWN_QUEUE_STRIDE] = 0u;
	if(blockIdx.x == 0 && threadIdx.x == PWN_QUEUES && P.clear_word != NULL) *P.clear_word = 0u;
	uint32_t q = (blockIdx.x * (PWN_BLOCK / 64) + (uint32_t)wave) % PWN_QUEUES;
	uint32_t ticket;
	// The first ticket of a wave is its place among the home waves of its queue: nobody draws it, and the counters
	// hand out the tickets after those (QBASE).  With a drawn first ticket every wave of the grid waits for a
	// returning atomic at the start of the launch, 80 of them per counter at once on a full grid (a draw that a wave
	// waits for costs it 0.7 us on average at 4K and 2.7 us in a strip of an 8-way tiling, tools/r3/draw_probe.py);
	// static first tickets measured +1.3 % at 4K, +2.5 % at 720p, -0.8 % on the strips' kernel time
	// (profiles/r3_strips/static_first.txt).
	const uint32_t nwaves_all = gridDim.x * (PWN_BLOCK / 64);
#define QBASE(qq) ((nwaves_all + PWN_QUEUES - 1u - (qq)) / PWN_QUEUES)
	ticket = (blockIdx.x * (PWN_BLOCK / 64) + (uint32_t)wave) / PWN_QUEUES;
	// A wave that keeps finding queues empty although they looked open stops helping after a
	// few rounds: every queue is drained by its home waves anyway (a wave leaves its home queue
	// only when that is empty), so this costs parallelism at the very end at worst and makes
	// sure every wave's loop ends whatever the loads return.
	int misses = 0;
#ifdef PWN_DRAW_PROBE
	unsigned long long probe_ticks = 0ull, probe_n = 0ull;
#endif
	// Tickets are drawn two at a time when the launch is long (>= 16 units per wave): the returning atomic is a
	// 32-byte write at the memory side, 4 MB per 4K frame with one per unit, 2 MB with pairs.  Strips and small
	// frames keep single tickets for the balance of their tail; three per draw measured 2.5 % slower at 4K (the
	// tail) for another 0.6 MB.  `left` = tickets in hand after the current one (wave-uniform).
	const uint32_t draw_n = units >= 16u * (PWN_BLOCK / 64u) * gridDim.x ? 2u : 1u;
	uint32_t left = 0u;
	for(;;)
	{
		//@R k_unit
		// units of queue q: q, q + Q, ...  below `units`
		const uint32_t qlen = (units + PWN_QUEUES - 1u - q) / PWN_QUEUES;
		if(ticket >= qlen)
		{
			//@R k_help
			RG(RG_HELP);
			if(++misses > 2 * (int)PWN_QUEUES) break;
			left = 0u;
			// this queue is empty: find one that is not (plain loads; a stale value can only
			// look fuller than the queue is, and then the atomic below says so)
			uint32_t seen = 0xffffffffu;
			// (the lane number recomputed and made opaque here: otherwise the address and the queue length
			// below are computed once at the top of the kernel and live in scratch memory until this rare
			// block: 20 B per lane written by every wave of every launch, 6.5 MB per 4K frame)
			uint32_t ql = 0u;
			asm volatile("" : "+v"(ql));
			ql = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, ql));     // = lane
			if(ql < PWN_QUEUES)
				seen = __hip_atomic_load(&P.tickets[ql * PWN_QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t len_l = (units + PWN_QUEUES - 1u - (ql & (PWN_QUEUES - 1u))) / PWN_QUEUES;
			const unsigned long long open = __ballot(ql < PWN_QUEUES && seen < len_l - min(len_l, QBASE(ql & (PWN_QUEUES - 1u))));
			if(open == 0ull) break;
			// the next open queue after q, cyclically: bit i of the shifted double mask is queue q+1+i
			static_assert(PWN_QUEUES <= 64u && (PWN_QUEUES & (PWN_QUEUES - 1u)) == 0u, "a power of two, one lane per queue");
			if constexpr(PWN_QUEUES == 64u)
			{
				const uint32_t rot = q + 1u;                 // 1..64
				const unsigned long long r = rot == 64u ? open : ((open >> rot) | (open << (64u - rot)));
				q = (q + 1u + (uint32_t)__builtin_ctzll(r)) & 63u;
			}
			else
				q = (q + 1u + (uint32_t)__builtin_ctzll((open | (open << (PWN_QUEUES & 31u))) >> (q + 1u))) & (PWN_QUEUES - 1u);
			uint32_t t = 0;
			if(lane == 0) t = atomicAdd(&P.tickets[q * PWN_QUEUE_STRIDE], 1u);
			ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + QBASE(q);
			continue;
		}
		//@R k_unit
		RG(RG_UNIT);
		misses = 0;
		// Which unit a ticket stands for: ticket * 64 + q in arithmetic order, or -- PWN_OPT_UNIT_ORDER, off by default -- what
		// the table says: every queue's units sorted by what they cost in the last launch of this geometry, dearest first
		// (pwn_order_kernel), so that the units handed out last are the cheap ones.  The reference's answer to uneven rows
		// is OpenMP's static schedule (screen.h:63-64).  Never changes a pixel: any permutation of the units does.
		// (the 4-lane variant is out of registers: loop-invariant lane values -- the lane number, "am I lane 0" -- end in
		// scratch memory there unless they are made afresh per unit, from mbcnt behind an opaque zero)
		uint32_t ln;
		if constexpr(HAS_W)
		{
			ln = 0u;
			asm volatile("" : "+v"(ln));
			ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, ln));
		}
		else ln = (uint32_t)lane;
		uint32_t unit = ticket * PWN_QUEUES + q;
		// (a SCALAR load: the address is the same for the whole wave, and the scalar cache answers in a fraction of the
		// microsecond a vector load takes here -- every unit waits for this word before it can do anything)
		if(ORDER && P.perm != NULL)
		{
			const uint32_t *pp = P.perm + (uint32_t)__builtin_amdgcn_readfirstlane((int)(q * P.perm_cap + ticket));
			asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(unit) : "s"(pp) : "memory");
		}
		const bool draw = left == 0u;
		// The next unit is asked for BEFORE this one is traced, which commits the wave to two units -- near the end of a
		// launch that is the tail: the last ticket of a queue goes to a wave that still has a whole unit in front of it
		// while its neighbours find the queues empty and leave.  Drawing only when a unit is done, throughout or for the
		// last tickets of a queue, was built and measured in round 3 (profiles/r3_strips/late_draws.txt, commit 5a68fef):
		// slower everywhere (a strip of an 8-way tiling 64 -> 65..77 us, 720p 58 -> 62..75 us, 4K equal), also with the
		// draw issued in front of the unit's colour store.  tools/r3/draw_probe.py times the wait in place: 0.7 us per
		// draw at 4K (4.7 % of a wave's life), 2.7 us in a strip of an 8-way tiling (13.8 %: its 5 120 waves draw
		// in step, 80 per counter), profiles/r3_strips/draw_probe.txt.
		uint32_t next_raw = ticket + 1u;
#ifndef PWN_DRAW_PROBE
		if(draw && ln == 0u) next_raw = atomicAdd(&P.tickets[q * PWN_QUEUE_STRIDE], draw_n) + QBASE(q);
#endif
		// rows from the middle outwards: the horizon band, where rays run longest,
		// is started first and the cheap top and bottom edges make up the tail
		// (this arithmetic is the same for the whole wave, but the compiler does it per lane because q starts
		// from the wave number, which it derives from threadIdx.  Declaring q uniform and dividing by a multiply-high
		// with a host-computed reciprocal moves ~35 VALU instructions per unit to the scalar unit: measured 0.8 %
		// SLOWER at 4K, three runs -- a wave's scalar instructions issue one at a time and in order)
		// unit / units_x by the host's reciprocal (pwn_trace_params.ux_magic: exact for every unit < 2^31, pwn_api.cpp
		// unit_div_magic): two instructions where the compiler's division takes thirteen; frames one unit wide divide
		uint32_t k;
		// (ux_shift < 0 only for units_x == 1, pwn_api.cpp unit_div_magic: then k = unit.  A real division here had its
		// reciprocal hoisted to the top of the kernel and, in the 4-lane variant, parked in scratch memory)
		if(P.ux_shift >= 0) k = __umulhi(unit, P.ux_magic) >> P.ux_shift;
		else k = unit;
		const uint32_t ux = unit - k * units_x;
		const uint32_t rows_u = ((uint32_t)(P.y1 - P.y0) + 3u) >> 2;
		// ... of the FRAME: a strip of a row tiling starts at its rows nearest the frame's middle row (the strip of
		// the whole frame at its own middle), not at its own middle
		const int hrow = ((P.h >> 1) - P.y0) >> 2;
		const uint32_t mid = (uint32_t)min(max(hrow, 0), (int)rows_u - 1);
		// k = 0,1,2,3,... -> mid, mid-1, mid+1, mid-2, ... while there are rows on both sides (a above, b below),
		// then the rest of the longer side in order
		uint32_t uy;
		{
			const uint32_t a = mid, b = rows_u - 1u - mid, m = min(a, b);
			const uint32_t j = (k + 1u) >> 1;
			if(k <= 2u * m) uy = (k & 1u) ? mid - j : mid + j;
			else uy = a > b ? mid - (k - b) : mid + (k - a);
		}
		const int half = (int)(ux & 1u);                  // left / right half of the 32-wide tile
		const int cx0 = (int)(ux >> 1) * 32;              // the 32-pixel tile of screen.h:6-7 this wave is in
		const int x = (int)ux * 16 + (HAS_W ? (int)(ln & 15u) : l16), y = P.y0 + (int)uy * 4 + (HAS_W ? (int)(ln >> 4) : (lane >> 4));

		// screen.h:12-18, in the order the reference build evaluates it:
		// rayl = (cx*rdx + rayb) + y*rdy, then one "+= rdx" per pixel of the
		// 32-wide tile up to and including this one.  The chain is sequential
		// in fp32, but every pixel of a row walks the SAME chain, so the lanes of
		// a DPP row build it systolically: after k rounds of
		//     v[j] = v[j-1] + rdx      (lane 0 of the row keeps its value)
		// lanes 0..k hold their final value.  All 64 lanes take part (also those
		// outside the frame), so this sits in front of the bounds test.
		V rayl = vadd<HAS_W>(vadd<HAS_W>(vscale<HAS_W>((float)cx0, rdx), rayb), vscale<HAS_W>((float)y, rdy));
		if(half)
		{
			//@R k_unit_half
			RG(RG_UNIT_HALF);
#pragma unroll
			for(int k = 0; k < 16; k++) rayl = vadd<HAS_W>(rayl, rdx);
		}
		//@R k_unit
		rayl = vadd<HAS_W>(rayl, rdx);
#ifndef PWN_CHAIN_DPP
		chain_rounds<HAS_W>(rayl, rdx);
#else
		{
			const bool first = (l16 == 0);
			V add;
			add.x = first ? rayl.x : rdx.x; add.y = first ? rayl.y : rdx.y; add.z = first ? rayl.z : rdx.z;
			add.w = HAS_W ? (first ? rayl.w : rdx.w) : 0.0f;
#pragma unroll
			for(int k = 1; k < 16; k++)
			{
				rayl.x = dpp_row_shr1(rayl.x) + add.x;
				rayl.y = dpp_row_shr1(rayl.y) + add.y;
				rayl.z = dpp_row_shr1(rayl.z) + add.z;
				if constexpr(HAS_W) rayl.w = dpp_row_shr1(rayl.w) + add.w;
			}
		}
#endif

		unsigned long long u_begin = 0ull;
		if(ORDER && P.unit_cost != NULL) u_begin = __builtin_amdgcn_s_memrealtime();
		if(x < P.w && y < P.y1)
		{
			// screen.h:19-21 (uint32 wrap-around)
			uint32_t seed = (uint32_t)x + (uint32_t)y * (uint32_t)y * ((uint32_t)P.w + 1u);
			seed *= seed * seed;
			seed *= seed * seed;
			seed <<= 1;                               // the generator runs on the doubled state (lcg2_fs, dev_math.h)

			float ox, oy, oz, ow;
			const uint32_t o = __umul24((uint32_t)y, (uint32_t)P.w) + (uint32_t)x;      // w, h <= 32768 (pwn_init)
			trace_pixel<COUNT, HAS_W, INL>(L, P.sec_current, seed, from, rayl, ox, oy, oz, ow, P.zbuf + o, cnt);
			P.sbuf[o] = col_pack4(ox, oy, oz, ow);
		}
		// what this unit cost its wave (the add chain and the ticket arithmetic in front of it are the same for every unit)
		if(ORDER && P.unit_cost != NULL && ln == 0u)
		{
			const unsigned long long d = (__builtin_amdgcn_s_memrealtime() - u_begin) >> 2;
			P.unit_cost[unit] = (uint16_t)(d > 65535ull ? 65535ull : d);
		}
#ifdef PWN_DRAW_PROBE
		// experiment build (tools/r3/draw_probe.py): the draw AFTER the unit, and how long the wave waits for it
		{
			const unsigned long long p0 = __builtin_amdgcn_s_memrealtime();
			if(draw && lane == 0) next_raw = atomicAdd(&P.tickets[q * PWN_QUEUE_STRIDE], draw_n) + QBASE(q);
			ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_raw);
			asm volatile("" : "+s"(ticket));
			probe_ticks += __builtin_amdgcn_s_memrealtime() - p0;
			probe_n++;
		}
#else
		ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_raw);
#endif
		left = draw ? draw_n - 1u : left - 1u;
	}

	//@R k_epilogue
	if(COUNT)
	{
		// wave reduce, one atomic per wave and counter
		unsigned long long v[16 + RG_N] = { cnt.rays, cnt.steps, cnt.portals, cnt.tests, cnt.exhausted, cnt.wsteps,
			cnt.wp[0], cnt.wp[1], cnt.wp[2], cnt.wp[3], cnt.wp[4], cnt.wp[5], cnt.wp[6], cnt.wp[7], cnt.apasses, cnt.apass_lanes };
		for(int i = 0; i < RG_N; i++) v[16 + i] = cnt.rg[i];
		for(int i = 0; i < 16 + RG_N; i++)
		{
			unsigned long long s = v[i];
			for(int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
			if(lane == 0 && s) atomicAdd(&P.counters[i], s);
		}
		// (waves of this launch: the prologue and the epilogue of the issue model)
		if(lane == 0) atomicAdd(&P.counters[16 + RG_N], 1ull);
	}
	// PWN_OPT_WAVE_LOG: every wave's lifetime (pwn_stats.wave_time ..., tools/wave_log.py)
	// (Which wave of the workgroup this is comes from the hardware: the four waves of a 256-thread workgroup
	// sit on the four SIMDs of their CU, HW_ID.simd_id is bits 5:4 of hardware register 4; the lane number
	// comes from mbcnt.  Keeping threadIdx.x alive to the end of the kernel costs a scratch slot per lane,
	// and a shared append counter serialises the waves' exits and stretches the very tail it measures.)
	if(P.wave_log != NULL && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
	{
		static_assert(PWN_BLOCK == 256, "one wave per SIMD: simd_id tells the waves of a workgroup apart");
		const unsigned simd = __builtin_amdgcn_s_getreg(4 | (4 << 6) | ((2 - 1) << 11));
		const size_t wid = 1u + (size_t)blockIdx.x * 4u + simd;
#ifdef PWN_DRAW_PROBE
		P.wave_log[2 * wid] = probe_ticks | (probe_n << 40); P.wave_log[2 * wid + 1] = __builtin_amdgcn_s_memrealtime() - t_begin;
#else
		P.wave_log[2 * wid] = t_begin; P.wave_log[2 * wid + 1] = __builtin_amdgcn_s_memrealtime();
#endif
	}
	// Row tiling with moving cuts (pwn_tiled.cpp): what this strip COST, as the sum of its waves' lifetimes in ticks of
	// the constant 100 MHz clock -- a wave lives exactly as long as it finds units, so the sum is the strip's work
	// in wave-time, whatever the tail of the launch looked like.  The four waves of a workgroup add up in LDS (one
	// ds_add_rtn: lifetime in the low 28 bits, a count in the high four) and the last one to leave adds the
	// workgroup's sum to the word: ~1300 no-return atomics per launch, spread over its tail.
	if(P.cost_word != NULL && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
	{
		static_assert(PWN_BLOCK == 256, "four waves per workgroup");
		const uint32_t life = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_begin) & 0x03ffffffu;
		PWN_LDS uint32_t *wg = (PWN_LDS uint32_t *)(uintptr_t)((P.blob_bytes + 15u) & ~15u);
		const uint32_t old = __hip_atomic_fetch_add(wg, life + (1u << 28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if((old >> 28) == 3u) (void)__hip_atomic_fetch_add(P.cost_word, (old + life) & 0x0fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

template<bool COUNT, bool HAS_W, bool ORDER, bool INL>
static hipError_t launch_variant(const pwn_trace_params *P, int grid, size_t lds_bytes, hipStream_t stream)
{
	// the dynamic-LDS limit is a per-function attribute: raise it only when the blob grew
	// (high-water mark per device and variant; contexts of several threads share it, so the
	// check and the raise happen under a lock and the mark only ever grows)
	static size_t lds_mark[64];
	static std::mutex lds_lock;
	int dev = 0;
	(void)hipGetDevice(&dev);
	{
		std::lock_guard<std::mutex> g(lds_lock);
		size_t &lds_set = lds_mark[dev & 63];
		if(lds_bytes > lds_set)
		{
			// the kernel addresses its tables from LDS address 0 (trace_common.h): that holds while it has no
			// static LDS, which would be laid out in front of the dynamic allocation
			hipFuncAttributes fa;
			hipError_t e = hipFuncGetAttributes(&fa, (const void *)pwn_trace_kernel<COUNT, HAS_W, ORDER, INL>);
			if(e != hipSuccess) return e;
			if(fa.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;
			e = hipFuncSetAttribute((const void *)pwn_trace_kernel<COUNT, HAS_W, ORDER, INL>,
				hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
			if(e != hipSuccess) return e;
			lds_set = lds_bytes;
		}
	}
	hipLaunchKernelGGL((pwn_trace_kernel<COUNT, HAS_W, ORDER, INL>), dim3(grid), dim3(PWN_BLOCK), lds_bytes, stream, *P);
	return hipGetLastError();
}

template<bool ORDER, bool INL>
static hipError_t launch_ordered(const pwn_trace_params *P, int grid, size_t lds_bytes, bool count, hipStream_t stream)
{
	if(count) return P->has_w ? launch_variant<true, true, ORDER, INL>(P, grid, lds_bytes, stream) : launch_variant<true, false, ORDER, INL>(P, grid, lds_bytes, stream);
	return P->has_w ? launch_variant<false, true, ORDER, INL>(P, grid, lds_bytes, stream) : launch_variant<false, false, ORDER, INL>(P, grid, lds_bytes, stream);
}

extern "C" hipError_t pwn_launch_trace(const pwn_trace_params *P, int grid, size_t lds_bytes, bool count, hipStream_t stream)
{
	// (the blob says which form its per-cell lists have: pack_blob, pwn_api.cpp)
	if(P->off_recsph != 0u)
	{
		if(P->perm != NULL || P->unit_cost != NULL) return launch_ordered<true, true>(P, grid, lds_bytes, count, stream);
		return launch_ordered<false, true>(P, grid, lds_bytes, count, stream);
	}
	if(P->perm != NULL || P->unit_cost != NULL) return launch_ordered<true, false>(P, grid, lds_bytes, count, stream);
	return launch_ordered<false, false>(P, grid, lds_bytes, count, stream);
}

// resident 256-thread workgroups per CU for this variant and LDS size
extern "C" int pwn_trace_tile_h(void) { return TILE_H; }
extern "C" int pwn_trace_tile_w(void) { return TILE_W; }
// LDS a workgroup needs beyond the table blob
extern "C" unsigned pwn_trace_lds_extra(void)
{
	return 16u;        // the workgroup's cost word (pwn_trace_params.cost_word)
}

extern "C" int pwn_trace_blocks_per_cu(size_t lds_bytes, bool count, bool has_w)
{
	int n = 0;
	hipError_t e;
	if(count) e = has_w ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_kernel<true, true, false, false>, PWN_BLOCK, lds_bytes)
	                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_kernel<true, false, false, false>, PWN_BLOCK, lds_bytes);
	else e = has_w ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_kernel<false, true, false, false>, PWN_BLOCK, lds_bytes)
	               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_kernel<false, false, false, false>, PWN_BLOCK, lds_bytes);
	if(e != hipSuccess || n < 1) n = 2;
	return n;
}
