// trace_refill.hip -- the portal ray-march with wave64 ballot / prefix refill of live rays.
//
// Same arithmetic as trace_kernel.hip (trace_ray_prelude, screen.h:1-28, and trace.h; the cell
// step is the same text, trace_walk.inc), different schedule.  There a wave64 takes a 16x4-pixel
// unit and its 64 lanes run ray set-up, walk and shading in step, three times per pixel
// (REFLECT = 2): a lane whose ray ended waits until the slowest ray of the wave has ended, and
// a lane whose pixel is finished waits for the whole unit.  Rays differ a lot -- portal chains,
// mirror halls -- so the walk loop ran with 0.89 (level.txt), 0.88 (synth64) and 0.74
// (synth256) of its lanes active.
//
// Here a wave is one persistent walk loop.  Every lane carries its own pixel (output offset,
// bounce depth, composite stack, seed) and its own ray.  The loop runs
//
//   A  for the lanes whose ray has ended:  shade; then either bounce (a new ray for the same
//      pixel) or finish the pixel (composite, store) and take a NEW pixel: the lanes that need
//      one are found with a ballot, each takes pixel `handed_out + mbcnt(ballot)` of the wave's
//      current 16x4 unit (prefix rank = compaction of the requests onto consecutive pixels), and
//      the wave pulls the next unit from the work queues when the current one is used up;
//      ray set-up for every lane that got a new ray;
//   B  cell steps for the lanes that walk, until the rays set up in the last pass have all
//      ended, or until the ended ones have waited P.refill_limit lane-steps in sum: rays still
//      walking then are stragglers, and nobody waits for a straggler twice.
//
// Phase A is one pass of straight code whatever the number of lanes that need it (and it is
// 43 % of the kernel's time), so it is entered when a batch of lanes is ready, not per ray.
//
// The per-tile ray add-chain (screen.h:12-18) is still built systolically with DPP by all 64
// lanes when a unit is taken, and parked in LDS (3 or 4 x 64 floats per wave) where the lane that
// is handed pixel p reads entry p.
#include <hip/hip_runtime.h>
#include <mutex>
#include "trace_common.h"

enum { EV_IDLE = 4, EV_SETUP = 5, EV_DONE = 6 };      // lane states beyond EV_NONE (walking) .. EV_EXHAUSTED

#define RTAB_FLOATS(HAS_W) (((HAS_W) ? 4 : 3) * 64)

template<bool COUNT, bool HAS_W>
__global__ void __launch_bounds__(PWN_BLOCK, PWN_MIN_WAVES)
pwn_trace_refill_kernel(pwn_trace_params P)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

	// HBM -> LDS, 16 B per lane per trip
	{
		const uint4 *src = (const uint4 *)P.blob;
		uint4 *dst = (uint4 *)lds_raw;
		int n16 = (int)(P.blob_bytes >> 4);
		for(int i = threadIdx.x; i < n16; i += PWN_BLOCK) dst[i] = src[i];
	}
	__syncthreads();

	// (the tables are addressed from LDS address 0 on, trace_common.h; the launcher checks that this kernel has
	// no static LDS in front of the dynamic allocation)
	const Lds L = lds_tables(P.off_sph);

	typedef Vec<HAS_W> V;
	V rayb, rdx, rdy, cam_from;
	rayb.x = P.rayb[0]; rayb.y = P.rayb[1]; rayb.z = P.rayb[2]; rayb.w = HAS_W ? P.rayb[3] : 0.0f;
	rdx.x = P.rdx[0]; rdx.y = P.rdx[1]; rdx.z = P.rdx[2]; rdx.w = HAS_W ? P.rdx[3] : 0.0f;
	rdy.x = P.rdy[0]; rdy.y = P.rdy[1]; rdy.z = P.rdy[2]; rdy.w = HAS_W ? P.rdy[3] : 0.0f;
	cam_from.x = P.from[0]; cam_from.y = P.from[1]; cam_from.z = P.from[2]; cam_from.w = HAS_W ? P.from[3] : 1.0f;

	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int l16 = lane & 15;
	// this wave's ray table: component-major, one float per pixel of the current unit
	float *rtab = (float *)(lds_raw + ((P.blob_bytes + 15u) & ~15u)) + wave * RTAB_FLOATS(HAS_W);

	Counters cnt = {};
	// PWN_OPT_WAVE_LOG: when was this wave resident (the GPU's constant 100 MHz clock)
	unsigned long long t_begin = 0ull;
	if(P.wave_log != NULL) t_begin = __builtin_amdgcn_s_memrealtime();

	// ---- work queues (as in trace_kernel.hip: PWN_QUEUES counters 128 B apart, queue q holds the
	// units u = q (mod PWN_QUEUES); a wave drains its home queue, then helps with the others, and
	// asks for its next unit when it takes one, so the round trip of the atomic hides behind the
	// unit's 64 pixels).  The counters of the NEXT launch of this context are cleared here.
	const uint32_t units_x = ((uint32_t)P.w + 15u) >> 4;
	const uint32_t rows_u = ((uint32_t)(P.y1 - P.y0) + 3u) >> 2;
	const uint32_t units = units_x * rows_u;
	if(blockIdx.x == 0 && threadIdx.x < PWN_QUEUES) P.tickets_next[threadIdx.x * PWN_QUEUE_STRIDE] = 0u;
	if(blockIdx.x == 0 && threadIdx.x == PWN_QUEUES && P.clear_word != NULL) *P.clear_word = 0u;
	uint32_t q = (blockIdx.x * (PWN_BLOCK / 64) + (uint32_t)wave) % PWN_QUEUES;
	uint32_t ticket_v = 0;                 // lane 0: the ticket drawn ahead for queue q
	if(lane == 0) ticket_v = atomicAdd(&P.tickets[q * PWN_QUEUE_STRIDE], 1u);
	int misses = 0;
	bool more_units = true;
	uint32_t handed = 64u;                 // pixels of the current unit already handed out (64: none left)
	int unit_x0 = 0, unit_y0 = 0;          // its origin in the frame

	// ---- per lane: the pixel
	int ev = EV_IDLE;
	uint32_t o = 0u, seed = 0u;
	int depth = 0;
	// icol (screen.h:24; trace.h:90 for a bounced ray) is the colour of the composite stack's top entry (trace_kernel.hip)
	float st_refl0 = 0.0f, st_refl1 = 0.0f, st_fog0 = 0.0f, st_fog1 = 0.0f;
	float sc0x = 1.0f, sc0y = 1.0f, sc0z = 1.0f, sc1x = 0.0f, sc1y = 0.0f, sc1z = 0.0f;
#define icx sc0x
#define icy sc0y
#define icz sc0z
	float w_acc = 0.0f;                                 // stand-in for the colour's w lane (trace_kernel.hip)
	// ---- per lane: the ray (trace.h:186-248)
	V pos, ray, aux_pos;
	pos.x = pos.y = pos.z = pos.w = 0.0f; ray = pos; aux_pos = pos;
	float cdist = 0.0f, fog = 0.0f, aux_dist = __builtin_inff(), aux_diff = 0.0f;
	uint32_t aux_idx = 0u;
	uint32_t cxz = 0u, sx = 1u, sz = 1u << 16;         // cell x | z << 16 and the steps (gx, 0), (0, gz): trace_common.h
	int ldx = FXP, ldz = FZP, ldy = FYP, ldir = FYN, base = BASE_ROOM_Y, maxsteps = 0;
	float wx = 0.0f, wy = 0.0f, wz = 0.0f, iax = 0.0f, iay = 0.0f, iaz = 0.0f, iay_dn = 0.0f;
	uint32_t iay_up_bits = 0u, cw = 0u;

	const int limit = P.refill_limit;                   // lane-steps the ended rays of a batch wait for the others in sum

#pragma unroll 1
	for(;;)
	{
		// =========================================================== A.1: rays that ended
		if(ev >= EV_WALL && ev <= EV_EXHAUSTED)
		{
			if(COUNT) { cnt.apass_lanes++; if((__ffsll((long long)__ballot(1)) - 1) == lane) cnt.apasses++; }
			float vx, vy, vz, vw;
			bool finished;
			if(ev == EV_EXHAUSTED)
			{
				// trace.h:677-678: out of steps -- the walked ray is the colour; a primary ray leaves the old depth in place
				if(COUNT) cnt.exhausted++;
				vx = ray.x; vy = ray.y; vz = ray.z; vw = HAS_W ? ray.w : 0.0f;
				finished = true;
			}
			else
			{
				if(ev == EV_WALL && base == BASE_ROOM_Y) { ldir = ldy; base = (ldy == FYP ? BASE_CEIL : BASE_FLOOR); }
				// zbuf = the PRIMARY ray's hit distance (trace.h:102-105)
				if(depth == 0) P.zbuf[o] = (ev == EV_SPHERE ? aux_dist : cdist);

				float colx, coly, colz, refl;
				V aux_norm;
				aux_norm.x = aux_norm.y = aux_norm.z = aux_norm.w = 0.0f;
				if(ev == EV_WALL)
				{
					// trace.h:108-154 and the axis-aligned mirrors of trace.h:50-75 from the constant tables in LDS
					// (tables.h PWN_T_FACES; trace_kernel.hip has the same block)
					// (one table entry at a time: this kernel keeps every lane's walk state live across phase A)
					float diffuse = (ldir & 1) ? ray.z : ray.x;
					diffuse = ldir >= FYP ? ray.y : diffuse;
					{
						const pwn_f4 fb = L.faces[5 + 2 * ldir];
						diffuse = __uint_as_float(__float_as_uint(diffuse) ^ __float_as_uint(fb.w));      // -ray.c on the N faces
						pos.x += fb.x; pos.y += fb.y; pos.z += fb.z;
					}
					if(diffuse < 0.0f) diffuse = 0.0f;
					const float amb = 0.1f;
					diffuse = (1.0f - amb) * diffuse + amb;
					{
						const pwn_f4 wc = L.faces[base];
						colx = diffuse * (icx * wc.x); coly = diffuse * (icy * wc.y); colz = diffuse * (icz * wc.z);
					}
					w_acc = __builtin_fmaf(diffuse, 0.0f, w_acc);
					{
						const pwn_f4 fa = L.faces[4 + 2 * ldir];
						refl = fa.w;
						ray.x = __uint_as_float(__float_as_uint(ray.x) ^ __float_as_uint(fa.x));
						ray.y = __uint_as_float(__float_as_uint(ray.y) ^ __float_as_uint(fa.y));
						ray.z = __uint_as_float(__float_as_uint(ray.z) ^ __float_as_uint(fa.z));
					}
				}
				else
				{
					// trace.h:283-291 for the committed sphere
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

				// trace.h:3-7
				if(depth >= REFLECT_MAX || refl == 0.0f) { vx = colx; vy = coly; vz = colz; vw = 0.0f; finished = true; }
				else
				{
					// trace.h:9-75
					if(ldir == FYN)
					{
						const float pi = (float)3.14159265358979323846;
						float ang = (pi * 2.0f) * (
							(glibc_sincosf((pi * 0.5f) * pos.x, 0) + glibc_sincosf((pi * 0.5f) * pos.z, 1))
							+ P.sec_current);
						const float2 sc = glibc_sincosf_both(ang);
						V n; n.x = sc.x; n.y = 38.0f; n.z = sc.y; n.w = 0.0f;
						n = vnormalise<HAS_W>(L.rsq, n);
						float rmul = -2.0f * ((ray.x * n.x + ray.y * n.y) + ray.z * n.z);
						ray = vnormalise<HAS_W>(L.rsq, vadd<HAS_W>(vscale<HAS_W>(rmul, n), ray));
					}
					else if(ldir < 0)
					{
						pos = vsub<HAS_W>(pos, vscale<HAS_W>(0.001f, ray));
						float rmul = -2.0f * ((ray.x * aux_norm.x + ray.y * aux_norm.y) + ray.z * aux_norm.z);
						ray = vnormalise<HAS_W>(L.rsq, vadd<HAS_W>(vscale<HAS_W>(rmul, aux_norm), ray));
					}

					// trace.h:77-84: five draws, two discarded
					ray.x += lcg2_fs(seed) * REFLECT_BLUR_F;
					ray.y += lcg2_fs(seed) * REFLECT_BLUR_F;
					lcg2_next(seed);
					ray.z += lcg2_fs(seed) * REFLECT_BLUR_F;
					lcg2_next(seed);

					// the composite stack as a shift register; its top entry's colour is the next segment's icol
					st_refl1 = st_refl0; st_fog1 = st_fog0; sc1x = sc0x; sc1y = sc0y; sc1z = sc0z;
					st_refl0 = refl; st_fog0 = fog; sc0x = colx; sc0y = coly; sc0z = colz;
					depth++;
					// the bounced ray starts at pos with direction ray (trace.h:90): set up below
					ev = EV_SETUP;
					finished = false;
					vx = vy = vz = vw = 0.0f;
				}
			}
			if(finished)
			{
				// trace.h:91-101, innermost first: the top of the stack, then the entry below it
				if(depth >= 1)
				{
					const float r0 = st_refl0, q0 = 1.0f - st_refl0;
					vx = r0 * vx + q0 * sc0x; vy = r0 * vy + q0 * sc0y; vz = r0 * vz + q0 * sc0z; vw = r0 * vw;
					if(st_fog0 != 0.0f)
					{
						float f = glibc_expf(-0.6f * st_fog0, L.exp2), g = 1.0f - f;
						vx = f * vx + g; vy = f * vy + g; vz = f * vz + g; vw = f * vw + g;
					}
				}
				if(depth >= 2)
				{
					const float r1 = st_refl1, q1 = 1.0f - st_refl1;
					vx = r1 * vx + q1 * sc1x; vy = r1 * vy + q1 * sc1y; vz = r1 * vz + q1 * sc1z; vw = r1 * vw;
					if(st_fog1 != 0.0f)
					{
						float f = glibc_expf(-0.6f * st_fog1, L.exp2), g = 1.0f - f;
						vx = f * vx + g; vy = f * vy + g; vz = f * vz + g; vw = f * vw + g;
					}
				}
				// screen.h:22 (col_ftoint, util.h:48-59)
				P.sbuf[o] = col_pack4(vx, vy, vz, vw + w_acc);
				ev = EV_IDLE;
			}
		}

		// =========================================================== A.2: new pixels
		// Lanes without a pixel take the next ones of the wave's unit: the k-th requesting lane
		// (k = number of requesting lanes below it: mbcnt of the ballot) takes pixel handed + k.
		{
			unsigned long long need = __ballot(ev == EV_IDLE);
			while(need != 0ull)
			{
				if(handed >= 64u)
				{
					// ---- the next unit (everything here is wave-uniform; all 64 lanes take part)
					bool got = false;
					uint32_t unit = 0u;
					if(more_units)
					{
						uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket_v);
						for(;;)
						{
							// units of queue q: q, q + Q, ...  below `units`
							const uint32_t qlen = (units + PWN_QUEUES - 1u - q) / PWN_QUEUES;
							if(ticket < qlen) { got = true; unit = ticket * PWN_QUEUES + q; break; }
							// A wave that keeps finding queues empty although they looked open stops helping:
							// every queue is drained by its home waves anyway, and this bounds the loop
							// whatever the loads return.
							if(++misses > 2 * (int)PWN_QUEUES) break;
							uint32_t seen = 0xffffffffu;
							// = lane, opaque: keeps the address below out of the kernel's prologue (trace_kernel.hip)
							uint32_t ql = 0u;
							asm volatile("" : "+v"(ql));
							ql = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, ql));
							if(ql < PWN_QUEUES)
								seen = __hip_atomic_load(&P.tickets[ql * PWN_QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							const uint32_t len_l = (units + PWN_QUEUES - 1u - (ql & (PWN_QUEUES - 1u))) / PWN_QUEUES;
							const unsigned long long open = __ballot(ql < PWN_QUEUES && seen < len_l);
							if(open == 0ull) break;
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
							ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
						}
					}
					if(!got)
					{
						// no units left: the requesting lanes are done for good
						more_units = false;
						if(ev == EV_IDLE) ev = EV_DONE;
						break;
					}
					misses = 0;
					if(lane == 0) ticket_v = atomicAdd(&P.tickets[q * PWN_QUEUE_STRIDE], 1u);     // the next one, drawn ahead
					// rows from the middle of the strip outwards: the horizon band, where rays run longest,
					// is started first and the cheap top and bottom edges make up the tail
					const uint32_t ux = unit % units_x, k = unit / units_x;
					const uint32_t mid = rows_u >> 1;
					uint32_t uy;
					{
						const uint32_t d = (k + 1u) >> 1;
						const bool down = (k & 1u) != 0u;            // odd: above the middle
						int cand = down ? (int)mid - (int)d : (int)mid + (int)d;
						if(cand < 0) cand = (int)mid + (int)(k - mid);           // ran past the top: the remaining rows are at the bottom
						else if(cand >= (int)rows_u) cand = (int)mid - (int)(k - (rows_u - 1u - mid));   // ran past the bottom
						uy = (uint32_t)cand;
					}
					const int half = (int)(ux & 1u);                  // left / right half of the 32-wide tile
					const int cx0 = (int)(ux >> 1) * 32;              // the 32-pixel tile of screen.h:6-7 this unit is in
					unit_x0 = (int)ux * 16; unit_y0 = P.y0 + (int)uy * 4;
					const int y = unit_y0 + (lane >> 4);
					// screen.h:12-18 in the reference build's evaluation order, the add-chain built systolically
					// over the 16 lanes of a DPP row (see trace_kernel.hip)
					V rayl = vadd<HAS_W>(vadd<HAS_W>(vscale<HAS_W>((float)cx0, rdx), rayb), vscale<HAS_W>((float)y, rdy));
					if(half)
					{
#pragma unroll
						for(int j = 0; j < 16; j++) rayl = vadd<HAS_W>(rayl, rdx);
					}
					rayl = vadd<HAS_W>(rayl, rdx);
					{
						const bool first = (l16 == 0);
						V add;
						add.x = first ? rayl.x : rdx.x; add.y = first ? rayl.y : rdx.y; add.z = first ? rayl.z : rdx.z;
						add.w = HAS_W ? (first ? rayl.w : rdx.w) : 0.0f;
#pragma unroll
						for(int j = 1; j < 16; j++)
						{
							rayl.x = dpp_row_shr1(rayl.x) + add.x;
							rayl.y = dpp_row_shr1(rayl.y) + add.y;
							rayl.z = dpp_row_shr1(rayl.z) + add.z;
							if constexpr(HAS_W) rayl.w = dpp_row_shr1(rayl.w) + add.w;
						}
					}
					rtab[lane] = rayl.x; rtab[64 + lane] = rayl.y; rtab[128 + lane] = rayl.z;
					if constexpr(HAS_W) rtab[192 + lane] = rayl.w;
					handed = 0u;
				}
				// ---- hand out: prefix rank among the requesting lanes
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
				const uint32_t p = handed + rank;
				if(ev == EV_IDLE && p < 64u)
				{
					const int x = unit_x0 + (int)(p & 15u), y = unit_y0 + (int)(p >> 4);
					// (a pixel beyond the frame's edge is used up and the lane asks again)
					if(x < P.w && y < P.y1)
					{
						pos = cam_from;
						ray.x = rtab[p]; ray.y = rtab[64u + p]; ray.z = rtab[128u + p];
						if constexpr(HAS_W) ray.w = rtab[192u + p]; else ray.w = 0.0f;
						o = (uint32_t)y * (uint32_t)P.w + (uint32_t)x;
						// screen.h:19-21 (uint32 wrap-around)
						seed = (uint32_t)x + (uint32_t)y * (uint32_t)y * ((uint32_t)P.w + 1u);
						seed *= seed * seed;
						seed *= seed * seed;
						seed <<= 1;                   // the generator runs on the doubled state (lcg2_fs, dev_math.h)
						depth = 0; sc0x = sc0y = sc0z = 1.0f; w_acc = 0.0f;
						ev = EV_SETUP;
					}
				}
				handed = min(64u, handed + (uint32_t)__builtin_popcountll(need));
				need = __ballot(ev == EV_IDLE);
			}
		}

		// =========================================================== A.3: ray set-up (trace.h:186-248)
		unsigned long long fresh = __ballot(ev == EV_SETUP);
		if(ev == EV_SETUP)
		{
			cdist = 0.0f; fog = 0.0f;
			// aux_dist: the reference's "none yet" value -1 (trace.h:200) is kept as +inf (trace_kernel.hip)
			aux_dist = __builtin_inff(); aux_diff = 0.0f; aux_idx = 0u;
			aux_pos.x = aux_pos.y = aux_pos.z = aux_pos.w = 0.0f;
			if(COUNT) cnt.rays++;
			const V iray = ray;
			ray = vnormalise<HAS_W>(L.rsq, iray);
			const int cx = (int)pos.x, cz = (int)pos.z;
			// signs of the UN-normalised input (trace.h:225-227)
			const int gx = (iray.x < 0.0f ? -1 : 1);
			const int gz = (iray.z < 0.0f ? -1 : 1);
			const bool gyp = !(iray.y < 0.0f);          // gy > 0
			// trace.h:220-222 and 230-231: one test on the bit patterns, one wave-uniform branch (trace_kernel.hip)
			{
				const uint32_t EPSB = __float_as_uint(EPS);
				const uint32_t bx = __float_as_uint(ray.x) & 0x7fffffffu, by = __float_as_uint(ray.y) & 0x7fffffffu,
					bz = __float_as_uint(ray.z) & 0x7fffffffu;
				const bool plain = max(max(bx - EPSB, by - EPSB), bz - EPSB) < 0x7e800000u - EPSB;
				if(__builtin_expect(__ballot(!plain) == 0ull, 1))
				{
					iax = tab_rcp_pos(L.rcp, __uint_as_float(bx)); iay = tab_rcp_pos(L.rcp, __uint_as_float(by));
					iaz = tab_rcp_pos(L.rcp, __uint_as_float(bz));
				}
				else
				{
					if(fabsf(ray.x) < EPS) ray.x = (ray.x < 0.0f ? -EPS : EPS);
					if(fabsf(ray.y) < EPS) ray.y = (ray.y < 0.0f ? -EPS : EPS);
					if(fabsf(ray.z) < EPS) ray.z = (ray.z < 0.0f ? -EPS : EPS);
					iax = tab_rcp(L.rcp, fabsf(ray.x)); iay = tab_rcp(L.rcp, fabsf(ray.y)); iaz = tab_rcp(L.rcp, fabsf(ray.z));
				}
			}
			wx = pos.x - (float)cx; wy = pos.y; wz = pos.z - (float)cz;
			if(ray.x >= 0.0f) wx = 1.0f - wx;
			if(ray.y >= 0.0f) wy = 1.0f - wy;
			if(ray.z >= 0.0f) wz = 1.0f - wz;
			wx *= iax; wy *= iay; wz *= iaz;
			iay_dn = gyp ? iay : -iay;
			ldy = gyp ? FYP : FYN;
			iay_up_bits = gyp ? __float_as_uint(iay) : 0u;         // +iay when looking up, else +0
			cxz = cxz_pack_start(cx, cz); sx = (uint32_t)gx & 0xffffu; sz = (uint32_t)gz << 16;
			cw = cellword_pk(L, cxz);
			ldx = (gx < 0 ? FXN : FXP); ldz = (gz < 0 ? FZN : FZP);
			ldir = FYN; base = BASE_ROOM_Y;
			maxsteps = 1000;
			ev = EV_NONE;
		}

		// =========================================================== B: the walk (trace.h:250-675)
		// When to stop walking and make a pass of phase A for the rays that have ended?  Rays of one
		// wave are coherent: started together they mostly end within a few steps of each other, and a
		// ray that ended early is best left waiting for them, so that the pass (one stretch of code
		// whatever the number of lanes in it, 43 % of the kernel's time) runs once for all 64.  Leaving
		// at the first ended ray, or with a fixed number of lanes or steps, splits every such group in
		// two that then wait for each other: measured +60 % time.  What must not hold a wave up is the
		// ray that walks ten or a hundred times as far (portal chains, mirror halls).  So:
		//   - walk while a YOUNG ray (set up in the pass just made) is walking;
		//   - but stop once the lanes whose ray has ended have waited `limit` lane-steps in sum: a pass
		//     costs about as much as four cell steps of the whole wave, so waiting longer than that for
		//     the rest of a batch is what a pass would have bought (the ski-rental bound);
		//   - rays still walking then are stragglers; they stay in their lanes and walk along with the
		//     next batches, and nobody waits for a straggler a second time.
		const unsigned long long walking0 = __ballot(ev == EV_NONE);
		if(walking0 == 0ull) break;           // every lane is EV_DONE: set-up leaves no other state behind
		if(fresh == 0ull) fresh = walking0;   // nothing was set up in this pass: the walkers are the batch
		int waited = 0;
		if(ev == EV_NONE)
		{
			asm volatile("" : "+v"(iay_up_bits));        // keep it a register, not a select per step (trace_kernel.hip)
			bool on;
#pragma unroll 1
			do
			{
			constexpr bool INL = false;          // (this scheduler reads the indexed lists: pack_blob packs those for it)
#include "trace_walk.inc"
				// wave-uniform: a young ray still walks, and the ended ones have not waited too long
				const unsigned long long w = __ballot(ev == EV_NONE);
				waited += (int)__builtin_popcountll(walking0 & ~w);
				on = (w & fresh) != 0ull && waited < limit;
				maxsteps--;
			} while(ev == EV_NONE && maxsteps != 0 && on);
			asm volatile("" : "+v"(ev), "+v"(maxsteps));
			// trace.h:250,677: out of steps
			if(ev == EV_NONE && maxsteps == 0) ev = EV_EXHAUSTED;
			// what the ray ended on is read back from the register (see trace_kernel.hip)
			asm volatile("" : "+v"(ev));
		}
	}

	if(COUNT)
	{
		// wave reduce, one atomic per wave and counter
		unsigned long long v[16] = { cnt.rays, cnt.steps, cnt.portals, cnt.tests, cnt.exhausted, cnt.wsteps,
			cnt.wp[0], cnt.wp[1], cnt.wp[2], cnt.wp[3], cnt.wp[4], cnt.wp[5], cnt.wp[6], cnt.wp[7], cnt.apasses, cnt.apass_lanes };
		for(int i = 0; i < 16; i++)
		{
			unsigned long long s = v[i];
			for(int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
			if(lane == 0 && s) atomicAdd(&P.counters[i], s);
		}
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
		P.wave_log[2 * wid] = t_begin; P.wave_log[2 * wid + 1] = __builtin_amdgcn_s_memrealtime();
	}
}

template<bool COUNT, bool HAS_W>
static hipError_t launch_variant(const pwn_trace_params *P, int grid, size_t lds_bytes, hipStream_t stream)
{
	// the dynamic-LDS limit is a per-function attribute: raise it only when the blob grew
	// (high-water mark per device and variant, under a lock: contexts of several threads share it)
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
			hipError_t e = hipFuncGetAttributes(&fa, (const void *)pwn_trace_refill_kernel<COUNT, HAS_W>);
			if(e != hipSuccess) return e;
			if(fa.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;
			e = hipFuncSetAttribute((const void *)pwn_trace_refill_kernel<COUNT, HAS_W>,
				hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
			if(e != hipSuccess) return e;
			lds_set = lds_bytes;
		}
	}
	hipLaunchKernelGGL((pwn_trace_refill_kernel<COUNT, HAS_W>), dim3(grid), dim3(PWN_BLOCK), lds_bytes, stream, *P);
	return hipGetLastError();
}

extern "C" hipError_t pwn_launch_trace_refill(const pwn_trace_params *P, int grid, size_t lds_bytes, bool count, hipStream_t stream)
{
	if(count) return P->has_w ? launch_variant<true, true>(P, grid, lds_bytes, stream) : launch_variant<true, false>(P, grid, lds_bytes, stream);
	return P->has_w ? launch_variant<false, true>(P, grid, lds_bytes, stream) : launch_variant<false, false>(P, grid, lds_bytes, stream);
}

// LDS a workgroup needs beyond the table blob: the waves' ray tables
extern "C" unsigned pwn_trace_refill_lds_extra(bool has_w)
{
	return (unsigned)((PWN_BLOCK / 64) * RTAB_FLOATS(has_w) * sizeof(float));
}

extern "C" int pwn_trace_refill_blocks_per_cu(size_t lds_bytes, bool count, bool has_w)
{
	int n = 0;
	hipError_t e;
	if(count) e = has_w ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_refill_kernel<true, true>, PWN_BLOCK, lds_bytes)
	                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_refill_kernel<true, false>, PWN_BLOCK, lds_bytes);
	else e = has_w ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_refill_kernel<false, true>, PWN_BLOCK, lds_bytes)
	               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwn_trace_refill_kernel<false, false>, PWN_BLOCK, lds_bytes);
	if(e != hipSuccess || n < 1) n = 2;
	return n;
}
