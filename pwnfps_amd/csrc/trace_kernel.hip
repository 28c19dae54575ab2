// trace_kernel.hip -- the per-pixel portal ray-march on gfx950 (CDNA4).
//
// Replaces trace_ray_prelude (screen.h:1-28) + trace_ray / trace_ray_through /
// trace_hit_wall / trace_hit_bounce (trace.h) of the reference.
//
// Shape: persistent 256-thread workgroups (4 wave64).  Each workgroup copies
// the level blob (cells, rcp/rsqrt tables, portals, per-cell sphere lists,
// spheres: tables.h) HBM -> LDS once, then walks 32x8-pixel tiles of its row
// strip, one thread per pixel.  The reference's recursion (depth <= REFLECT)
// is a loop over at most three ray segments with the composites of
// trace.h:91-101 applied on unwinding.  Output: BGRA8 colour + fp32 depth,
// row-major, one 4-byte store each per pixel.
//
// No MFMA: this is a branchy DDA, not a contraction.  HBM traffic is the two
// output planes only (8 B / pixel); everything the inner loop reads is in LDS.
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "tables.h"

#define EPS 0.0000000000001f      // defs.h:1
#define REFLECT_BLUR_F 0.03f      // defs.h:5
#define REFLECT_MAX 2             // defs.h:7
enum { FXP = 0, FZP, FXN, FZN, FYP, FYN };   // defs.h:25-33

#define TILE_W 32
#define TILE_H 8
#define WAVE_W 16                 // a wave64 covers WAVE_W x (64/WAVE_W) pixels

struct Lds
{
	const uint8_t *cells;
	const uint16_t *rcp, *rsq;
	const uint32_t *pmap;
	const uint16_t *binoff, *binidx;
	const float *sph;
};

__device__ __forceinline__ int cell_at(const Lds &L, int cx, int cz)
{
	// util.h:151-158: per-axis clamp to 0
	if(cx < 0 || cx >= 64) cx = 0;
	if(cz < 0 || cz >= 64) cz = 0;
	return L.cells[cz * 64 + cx];
}

enum { EV_EXHAUSTED = 0, EV_WALL, EV_SPHERE };

struct Hit
{
	int ev, ldir;
	v4 ray, pos, norm, col;
	float refl, fog, dist;
};

struct Counters { uint32_t rays, steps, portals, tests, exhausted; };

template<bool COUNT>
__device__ __forceinline__ void walk(const Lds &L, v4 from, v4 iray, Hit &h, Counters &cnt)
{
	float cdist = 0.0f, fog = 0.0f, fogbeg = 0.0f;
	float aux_dist = -1.0f, aux_refl = 0.25f;
	v4 aux_pos = v4_set(0, 0, 0, 0), aux_norm = v4_set(0, 0, 0, 0), aux_col = v4_set(1, 1, 1, 1);
	if(COUNT) cnt.rays++;

	// trace.h:212-241
	v4 pos = from;
	v4 ray = v4_normalise(L.rsq, iray);
	int cx = (int)from.x, cz = (int)from.z;
	if(ray.x > -EPS && ray.x < EPS) ray.x = (ray.x < 0.0f ? -EPS : EPS);
	if(ray.y > -EPS && ray.y < EPS) ray.y = (ray.y < 0.0f ? -EPS : EPS);
	if(ray.z > -EPS && ray.z < EPS) ray.z = (ray.z < 0.0f ? -EPS : EPS);
	int gx = (iray.x < 0.0f ? -1 : 1);
	int gy = (iray.y < 0.0f ? -1 : 1);
	int gz = (iray.z < 0.0f ? -1 : 1);
	float iax = tab_rcp(L.rcp, fabsf(ray.x));
	float iay = tab_rcp(L.rcp, fabsf(ray.y));
	float iaz = tab_rcp(L.rcp, fabsf(ray.z));
	float wx = pos.x - (float)cx, wy = pos.y, wz = pos.z - (float)cz;
	if(ray.x >= 0.0f) wx = 1.0f - wx;
	if(ray.y >= 0.0f) wy = 1.0f - wy;
	if(ray.z >= 0.0f) wz = 1.0f - wz;
	wx *= iax; wy *= iay; wz *= iaz;

	int cell = cell_at(L, cx, cz);
	int ldir = FYN;

#define AUX_HIT() (aux_dist != -1.0f && cdist > aux_dist)
#define RET_SPHERE() do { h.ev = EV_SPHERE; h.ray = ray; h.pos = aux_pos; h.norm = aux_norm; \
	h.ldir = -1; h.refl = aux_refl; h.fog = fog; h.dist = aux_dist; h.col = aux_col; return; } while(0)
#define RET_WALL(c) do { h.ev = EV_WALL; h.ray = ray; h.pos = pos; h.ldir = ldir; \
	h.fog = fog; h.dist = cdist; h.col = (c); return; } while(0)
// trace.h:156-184
#define THROUGH(gxa) do { float t_; \
	if(wy < wx && wy < wz) { t_ = wy; ldir = (gy < 0 ? FYN : FYP); } \
	else if(wx < wz) { t_ = wx; ldir = ((gxa) < 0 ? FXN : FXP); } \
	else { t_ = wz; ldir = (gz < 0 ? FZN : FZP); } \
	cdist += t_; pos = v4_add(v4_scale(t_, ray), pos); } while(0)
// fog sums below are written in the reference build's operation order:
// (fog - fogbeg) + cdist and (fog + aux_dist) - fogbeg
// trace.h:331-340
#define ADVANCE_XZ() do { \
	if(ldir == FXN || ldir == FXP) { wy -= wx; wz -= wx; wx = iax; cx += gx; } \
	else { wx -= wz; wy -= wz; wz = iaz; cz += gz; } } while(0)
#define COL_CEIL  v4_set(30.0f, 30.0f, 0.0f, 0.0f)
#define COL_FLOOR v4_set(1.0f, 1.0f, 1.0f, 0.0f)
#define COL_WALL  v4_set(0.8f, 0.8f, 1.0f, 0.0f)

	for(int maxsteps = 1000; maxsteps > 0; maxsteps--)
	{
		if(COUNT) cnt.steps++;

		// trace.h:252-296: spheres binned to this cell
		if((unsigned)cx < 64u && (unsigned)cz < 64u)
		{
			int c = cz * 64 + cx;
			int k1 = L.binoff[c + 1];
			for(int k = L.binoff[c]; k < k1; k++)
			{
				const float *sp = L.sph + 8 * (int)L.binidx[k];
				if(COUNT) cnt.tests++;
				float sr = sp[0];
				v4 spos = v4_set(sp[2], sp[3], sp[4], 1.0f);
				float rad2 = sr * sr;
				v4 rel = v4_sub(spos, pos);
				float d2 = v4_dot(rel, rel);
				float dt = v4_dot(rel, ray);
				if(dt > 0.0f)
				{
					float calc = d2 - dt * dt;
					if(calc < rad2)
					{
						float sd2 = 1.0f - calc / rad2;
						float sdist = sqrtf(d2) - sqrtf(sd2);
						if(aux_dist == -1.0f || sdist + cdist < aux_dist)
						{
							aux_dist = sdist + cdist;
							aux_pos = v4_add(pos, v4_scale(sdist, ray));
							aux_norm = v4_normalise(L.rsq, v4_sub(aux_pos, spos));
							float diff = -v4_dot(ray, aux_norm);
							if(diff < 0.0f) diff = 0.0f;
							const float amb = 0.2f;
							aux_refl = sp[1];
							diff = amb + (1.0f - amb) * diff;
							aux_col = v4_scale(diff, v4_set(sp[5], sp[6], sp[7], 0.0f));
						}
					}
				}
			}
		}

		int this_cell = cell;
		if(this_cell == ';' || this_cell == '$' || this_cell == '"')
		{
			// trace.h:302-352: 1-high room
			if(this_cell == '$') fogbeg = cdist;
			THROUGH(gx);
			if(AUX_HIT())
			{
				if(this_cell == '$' && aux_dist > fogbeg) fog = (fog + aux_dist) - fogbeg;
				RET_SPHERE();
			}
			if(this_cell == '$') fog = (fog - fogbeg) + cdist;
			if(ldir == FYN || ldir == FYP) RET_WALL(gy > 0 ? COL_CEIL : COL_FLOOR);
			ADVANCE_XZ();
			cell = cell_at(L, cx, cz);
			if(this_cell == '"' && (cell == '#' || cell == '&'))
			{
				pos.y += 1.0f;
				if(gy < 0) wy += iay; else wy -= iay;
			}
		}
		else if(this_cell == '#' || this_cell == '&')
		{
			// trace.h:354-441: 2-high room
			if(gy > 0) wy += iay;
			if(this_cell == '&') fogbeg = cdist;
			THROUGH(gx);
			if(AUX_HIT())
			{
				if(this_cell == '&' && aux_dist > fogbeg) fog = (fog + aux_dist) - fogbeg;
				RET_SPHERE();
			}
			if(this_cell == '&') fog = (fog - fogbeg) + cdist;
			if(ldir == FYN || ldir == FYP) RET_WALL(gy > 0 ? COL_CEIL : COL_FLOOR);
			ADVANCE_XZ();
			if(gy > 0) wy -= iay;
			cell = cell_at(L, cx, cz);
			if(cell == '"')
			{
				pos.y -= 1.0f;
				if(gy > 0) wy += iay; else wy -= iay;
			}
			int xcell = cell;
			if(xcell >= 'A' && xcell <= 'Z')
			{
				// trace.h:404-413: look through a portal at the cell type behind it
				uint32_t p0 = L.pmap[2 * (xcell - 'A')], p1 = L.pmap[2 * (xcell - 'A') + 1];
				int x1 = (int)(int8_t)(p0 & 0xff), z1 = (int)(int8_t)((p0 >> 8) & 0xff);
				int x2 = (int)(int8_t)((p0 >> 16) & 0xff), z2 = (int)(int8_t)(p0 >> 24);
				if(x1 == cx && z1 == cz) xcell = (int)((p1 >> 16) & 0xff);
				else if(x2 == cx && z2 == cz) xcell = (int)((p1 >> 8) & 0xff);
			}
			if(pos.y < 0.0f || pos.y > 1.0f)
			{
				if(!(xcell == '#' || xcell == '&'))
				{
					if(xcell == '"')
					{
						pos.y += 1.0f;
						if(gy > 0) wy -= iay; else wy += iay;
					}
					RET_WALL(COL_WALL);
				}
			}
		}
		else if(this_cell == '>' || this_cell == '<' || this_cell == ',' || this_cell == '^')
		{
			// trace.h:443-505: ramps
			const float ramp = 0.5f;
			float tilt = (this_cell == '>' || this_cell == '<') ? ray.x : ray.z;
			bool minus = (this_cell == '>' || this_cell == ',');
			if(minus) ray.y -= ramp * tilt; else ray.y += ramp * tilt;
			wy = pos.y;
			if(ray.y >= 0.0f) wy = 1.0f - wy;
			wy *= 1.0f / (ray.y < 0.0f ? -ray.y : ray.y);
			if(AUX_HIT()) RET_SPHERE();
			THROUGH(gy); // sic: trace.h:470 passes gy for gx
			if(ldir == FYN || ldir == FYP)
			{
				ldir = (ray.y < 0.0f ? FYN : FYP);
				RET_WALL(ray.y >= 0.0f ? COL_CEIL : COL_FLOOR);
			}
			else if(ldir == FXN || ldir == FXP)
			{
				ldir = (ray.x < 0.0f ? FXN : FXP);
				wy -= wx; wz -= wx; wx = iax; cx += gx;
			}
			else
			{
				ldir = (ray.z < 0.0f ? FZN : FZP);
				wx -= wz; wy -= wz; wz = iaz; cz += gz;
			}
			tilt = (this_cell == '>' || this_cell == '<') ? ray.x : ray.z;
			if(minus) ray.y += ramp * tilt; else ray.y -= ramp * tilt;
			wy = pos.y;
			if(ray.y >= 0.0f) wy = 1.0f - wy;
			wy *= iay;
			cell = cell_at(L, cx, cz);
		}
		else if(this_cell >= 'A' && this_cell <= 'Z')
		{
			// trace.h:508-650: portal
			uint32_t p0 = L.pmap[2 * (this_cell - 'A')], p1 = L.pmap[2 * (this_cell - 'A') + 1];
			int x1 = (int)(int8_t)(p0 & 0xff), z1 = (int)(int8_t)((p0 >> 8) & 0xff);
			int x2 = (int)(int8_t)((p0 >> 16) & 0xff), z2 = (int)(int8_t)(p0 >> 24);
			int rot12 = (int)(p1 & 0xff);
			int rot;
			if(x2 == -1)
			{
				if(AUX_HIT()) RET_SPHERE();
				RET_WALL(COL_WALL);
			}
			if(x1 == cx && z1 == cz)
			{
				cx = x2; cz = z2;
				pos.x += (float)(x2 - x1);
				pos.z += (float)(z2 - z1);
				rot = (-rot12) & 3;
			}
			else if(x2 == cx && z2 == cz)
			{
				cx = x1; cz = z1;
				pos.x -= (float)(x2 - x1);
				pos.z -= (float)(z2 - z1);
				rot = rot12 & 3;
			}
			else
			{
				if(AUX_HIT()) RET_SPHERE();
				RET_WALL(v4_set(5.0f, 0.0f, 5.0f, 0.0f));
			}
			if(COUNT) cnt.portals++;

			// trace.h:561-622.  The operation order is the one the reference
			// build executes (its -ffast-math cancels the +-0.5 terms).
			float trx = pos.x, trz = pos.z, trvx = ray.x, trvz = ray.z;
			int tgx = gx, tgz = gz;
			float fcx = (float)cx, fcz = (float)cz, t;
			ldir = (ldir - rot) & 3;
			if(rot == 1)
			{
				pos.x = (trz + fcx) - fcz;
				pos.z = (1.0f - trx) + (fcx + fcz);
				ray.x = trvz; ray.z = -trvx;
				gx = tgz; gz = -tgx;
				t = wx; wx = wz; wz = t;
				t = iax; iax = iaz; iaz = t;
			}
			else if(rot == 2)
			{
				pos.x = (fcx + 0.5f) * 2.0f - trx;
				pos.z = (fcz + 0.5f) * 2.0f - trz;
				ray.x = -trvx; ray.z = -trvz;
				gx = -tgx; gz = -tgz;
			}
			else if(rot == 3)
			{
				pos.x = (1.0f - trz) + (fcx + fcz);
				pos.z = (fcz + trx) - fcx;
				ray.x = -trvz; ray.z = trvx;
				gx = -tgz; gz = tgx;
				t = wx; wx = wz; wz = t;
				t = iax; iax = iaz; iaz = t;
			}
			// trace.h:624-647: step out of the far endpoint
			if(ldir == FZP) { cz++; pos.z += 1.0f; }
			else if(ldir == FXN) { cx--; pos.x -= 1.0f; }
			else if(ldir == FZN) { cz--; pos.z -= 1.0f; }
			else { cx++; pos.x += 1.0f; }
			cell = cell_at(L, cx, cz);
		}
		else
		{
			// trace.h:651-664: solid
			if(AUX_HIT()) RET_SPHERE();
			RET_WALL(ldir == FYP ? COL_CEIL : COL_WALL);
		}

		// trace.h:668-673
		if(AUX_HIT()) RET_SPHERE();
	}

	// trace.h:677-678: out of steps -- the walked ray is the colour
	if(COUNT) cnt.exhausted++;
	h.ev = EV_EXHAUSTED;
	h.ray = ray;
#undef AUX_HIT
#undef RET_SPHERE
#undef RET_WALL
#undef THROUGH
#undef ADVANCE_XZ
}

// trace_ray(0, ...) of screen.h:22-24 with the recursion unrolled
template<bool COUNT>
__device__ __forceinline__ v4 trace_pixel(const Lds &L, float sec_current, uint32_t seed,
	v4 from, v4 iray, float &dist, bool &have_dist, Counters &cnt)
{
	v4 icol = v4_set(1.0f, 1.0f, 1.0f, 1.0f);
	float st_refl[REFLECT_MAX], st_fog[REFLECT_MAX];
	v4 st_col[REFLECT_MAX];
	int depth = 0;
	v4 value;
	have_dist = false;

#pragma unroll 1
	for(;;)
	{
		Hit h;
		h.norm = v4_set(0, 0, 0, 0);
		walk<COUNT>(L, from, iray, h, cnt);
		if(h.ev == EV_EXHAUSTED) { value = h.ray; break; }
		if(depth == 0) { dist = h.dist; have_dist = true; }

		v4 col;
		float refl;
		if(h.ev == EV_WALL)
		{
			// trace.h:108-154
			float diffuse;
			col = v4_mul(icol, h.col);
			switch(h.ldir)
			{
				case FYP: diffuse = h.ray.y; break;
				case FZP: diffuse = h.ray.z; break;
				case FXN: diffuse = -h.ray.x; break;
				case FYN: diffuse = -h.ray.y; break;
				case FZN: diffuse = -h.ray.z; break;
				default:  diffuse = h.ray.x; break;
			}
			if(diffuse < 0.0f) diffuse = 0.0f;
			const float amb = 0.1f;
			diffuse = (1.0f - amb) * diffuse + amb;
			col = v4_scale(diffuse, col);
			refl = (h.ldir == FYN ? 0.7f : 0.25f);
		}
		else
		{
			col = h.col;
			refl = h.refl;
		}

		// trace.h:3-7
		if(depth >= REFLECT_MAX || refl == 0.0f) { value = col; break; }

		// trace.h:9-75
		v4 ray = h.ray, pos = h.pos;
		if(h.ldir == FYN)
		{
			pos.y -= 0.001f;
			const float pi = (float)3.14159265358979323846;
			float ang = (pi * 2.0f) * (
				(glibc_sincosf((pi * 0.5f) * pos.x, 0) + glibc_sincosf((pi * 0.5f) * pos.z, 1))
				+ sec_current);
			v4 norm = v4_normalise(L.rsq, v4_set(glibc_sincosf(ang, 0), 38.0f, glibc_sincosf(ang, 1), 0.0f));
			float rmul = -2.0f * ((ray.x * norm.x + ray.y * norm.y) + ray.z * norm.z);
			ray = v4_normalise(L.rsq, v4_add(v4_scale(rmul, norm), ray));
		}
		else if(h.ldir < 0)
		{
			pos = v4_sub(pos, v4_scale(0.001f, ray));
			v4 norm = h.norm;
			float rmul = -2.0f * ((ray.x * norm.x + ray.y * norm.y) + ray.z * norm.z);
			ray = v4_normalise(L.rsq, v4_add(v4_scale(rmul, norm), ray));
		}
		else if(h.ldir == FXP) { ray.x = -ray.x; pos.x -= 0.001f; }
		else if(h.ldir == FXN) { ray.x = -ray.x; pos.x += 0.001f; }
		else if(h.ldir == FZP) { ray.z = -ray.z; pos.z -= 0.001f; }
		else if(h.ldir == FZN) { ray.z = -ray.z; pos.z += 0.001f; }
		else { ray.y = -ray.y; pos.y -= 0.001f; }

		// trace.h:77-84: five draws, two discarded
		ray.x += lcg_fs(seed) * REFLECT_BLUR_F;
		ray.y += lcg_fs(seed) * REFLECT_BLUR_F;
		lcg_next(seed);
		ray.z += lcg_fs(seed) * REFLECT_BLUR_F;
		lcg_next(seed);

		st_refl[depth] = refl; st_fog[depth] = h.fog; st_col[depth] = col;
		depth++;
		icol = col;
		from = pos;
		iray = ray;
	}

	// trace.h:91-101, innermost first
#pragma unroll
	for(int d = REFLECT_MAX - 1; d >= 0; d--)
	{
		if(d < depth)
		{
			float refl = st_refl[d];
			value = v4_add(v4_scale(refl, value), v4_scale(1.0f - refl, st_col[d]));
			if(st_fog[d] != 0.0f)
			{
				float f = glibc_expf(-0.6f * st_fog[d]);
				float g = 1.0f - f;
				value = v4_add(v4_scale(f, value), v4_set(g, g, g, g));
			}
		}
	}
	return value;
}

template<bool COUNT>
__global__ void __launch_bounds__(256)
pwn_trace_kernel(pwn_trace_params P)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

	// HBM -> LDS, 16 B per lane per trip
	{
		const uint4 *src = (const uint4 *)P.blob;
		uint4 *dst = (uint4 *)lds_raw;
		int n16 = (int)(P.blob_bytes >> 4);
		for(int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
	}
	__syncthreads();

	Lds L;
	L.cells = lds_raw + PWN_T_CELLS;
	L.rcp = (const uint16_t *)(lds_raw + PWN_T_RCP);
	L.rsq = (const uint16_t *)(lds_raw + PWN_T_RSQ);
	L.pmap = (const uint32_t *)(lds_raw + PWN_T_PMAP);
	L.binoff = (const uint16_t *)(lds_raw + PWN_T_BINOFF);
	L.binidx = (const uint16_t *)(lds_raw + PWN_T_BINIDX);
	L.sph = (const float *)(lds_raw + P.off_sph);

	const v4 rayb = v4_set(P.rayb[0], P.rayb[1], P.rayb[2], P.rayb[3]);
	const v4 rdx = v4_set(P.rdx[0], P.rdx[1], P.rdx[2], P.rdx[3]);
	const v4 rdy = v4_set(P.rdy[0], P.rdy[1], P.rdy[2], P.rdy[3]);
	const v4 from = v4_set(P.from[0], P.from[1], P.from[2], P.from[3]);

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// wave footprint WAVE_W x (64/WAVE_W); waves tile the 32x8 block
	const int waves_x = TILE_W / WAVE_W;
	const int lx = (wave % waves_x) * WAVE_W + (lane % WAVE_W);
	const int ly = (wave / waves_x) * (64 / WAVE_W) + (lane / WAVE_W);

	Counters cnt = { 0, 0, 0, 0, 0 };

	for(int tile = blockIdx.x; tile < P.tiles_total; tile += gridDim.x)
	{
		int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
		int cx0 = tx * TILE_W;
		int x = cx0 + lx, y = P.y0 + ty * TILE_H + ly;
		if(x < P.w && y < P.y1)
		{
			// screen.h:12-18, in the order the reference build evaluates it:
			// rayl = (cx*rdx + rayb) + y*rdy, then one "+= rdx" per pixel of
			// the 32-wide tile up to and including this one
			v4 rayl = v4_add(v4_add(v4_scale((float)cx0, rdx), rayb), v4_scale((float)y, rdy));
			for(int k = 0; k <= lx; k++) rayl = v4_add(rayl, rdx);

			// screen.h:19-21 (uint32 wrap-around)
			uint32_t seed = (uint32_t)x + (uint32_t)y * (uint32_t)y * ((uint32_t)P.w + 1u);
			seed *= seed * seed;
			seed *= seed * seed;

			float dist = 0.0f;
			bool have_dist;
			v4 c = trace_pixel<COUNT>(L, P.sec_current, seed, from, rayl, dist, have_dist, cnt);
			size_t o = (size_t)y * (size_t)P.w + (size_t)x;
			P.sbuf[o] = col_pack(c);
			if(have_dist) P.zbuf[o] = dist;
		}
	}

	if(COUNT)
	{
		// wave reduce, one atomic per wave and counter
		unsigned long long v[5] = { cnt.rays, cnt.steps, cnt.portals, cnt.tests, cnt.exhausted };
		for(int i = 0; i < 5; i++)
		{
			unsigned long long s = v[i];
			for(int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
			if(lane == 0 && s) atomicAdd(&P.counters[i], s);
		}
	}
}

extern "C" hipError_t pwn_launch_trace(const pwn_trace_params *P, int grid, size_t lds_bytes, bool count, hipStream_t stream)
{
	if(count)
	{
		hipError_t e = hipFuncSetAttribute((const void *)pwn_trace_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		if(e != hipSuccess) return e;
		hipLaunchKernelGGL(pwn_trace_kernel<true>, dim3(grid), dim3(256), lds_bytes, stream, *P);
	}
	else
	{
		hipError_t e = hipFuncSetAttribute((const void *)pwn_trace_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		if(e != hipSuccess) return e;
		hipLaunchKernelGGL(pwn_trace_kernel<false>, dim3(grid), dim3(256), lds_bytes, stream, *P);
	}
	return hipGetLastError();
}
