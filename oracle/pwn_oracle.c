/*
 * oracle/pwn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See pwn_oracle.h.  Every function cites the reference lines it restates.
 *
 * The reference recurses trace_ray -> trace_hit_wall -> trace_hit_bounce ->
 * trace_ray (depth <= REFLECT = 2).  Here one pixel is a loop over at most
 * three ray segments; each bounce pushes (refl, base colour, fog) and the
 * composite of trace.h:91-101 is applied while unwinding.
 *
 * Arithmetic rules that make the pixels match the reference build
 * (gcc -O3 -ffast-math, no FMA): fp32, every multiply and add rounded
 * separately (built with -ffp-contract=off), 4-lane dot products summed as
 * (x+z)+(y+w) (util.h:18-30), RCPPS/RSQRTPS via the captured tables, libm via
 * pwn_libm.h.  Places where -ffast-math reassociated the reference's source
 * expression are marked "assoc:" with the compiled order.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <xmmintrin.h>
#include <omp.h>
#include "pwn_oracle.h"
#include "approx_tables.h"
#include "pwn_libm.h"

/* defs.h:1,5,7 */
#define EPS 0.0000000000001f
#define REFLECT_BLUR_F 0.03f
#define REFLECT_MAX 2
/* defs.h:25-33 */
enum { FXP = 0, FZP, FXN, FZN, FYP, FYN };

typedef struct v4 { float x, y, z, w; } v4;

static inline v4 v4_set(float x, float y, float z, float w) { v4 r = { x, y, z, w }; return r; }
static inline v4 v4_add(v4 a, v4 b) { return v4_set(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 v4_sub(v4 a, v4 b) { return v4_set(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline v4 v4_mul(v4 a, v4 b) { return v4_set(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 v4_scale(float s, v4 a) { return v4_set(s * a.x, s * a.y, s * a.z, s * a.w); }
/* util.h:18-30 */
static inline float v4_dot(v4 a, v4 b)
{
	v4 p = v4_mul(a, b);
	return (p.x + p.z) + (p.y + p.w);
}
/* util.h:32-46: v * rsqrtps(|v|^2), no Newton step */
static inline v4 v4_normalise(v4 a)
{
	return v4_scale(pwn_tab_rsqrt(v4_dot(a, a)), a);
}

/* util.h:1-16; the divide by 3759.0f is a multiply by the float reciprocal
   under the reference's -freciprocal-math */
static inline uint32_t lcg_next(uint32_t *seed)
{
	*seed = (*seed * 25739u + 4u) & 0x7FFFFFFFu;
	return *seed;
}
static inline float lcg_fu(uint32_t *seed)
{
	const float inv = 1.0f / 3759.0f;
	return (float)(lcg_next(seed) % 3759u) * inv;
}
static inline float lcg_fs(uint32_t *seed) { return lcg_fu(seed) * 2.0f - 1.0f; }

/* util.h:48-59: cvtps2dq (RNE, indefinite 0x80000000), packs, packus */
static inline uint32_t ftoint_lane(float f)
{
	float s = f * 255.0f;
	int32_t i;
	if(!(s >= -2147483648.0f && s < 2147483648.0f)) i = INT32_MIN;
	else i = (int32_t)__builtin_rintf(s);
	if(i < -32768) i = -32768;
	if(i > 32767) i = 32767;
	if(i < 0) i = 0;
	if(i > 255) i = 255;
	return (uint32_t)i;
}
static inline uint32_t col_pack(v4 c)
{
	return ftoint_lane(c.x) | (ftoint_lane(c.y) << 8) | (ftoint_lane(c.z) << 16) | (ftoint_lane(c.w) << 24);
}

/* util.h:151-158 */
static inline int cell_at(const pwno_level *lv, int cx, int cz)
{
	if(cx < 0 || cx >= 64) cx = 0;
	if(cz < 0 || cz >= 64) cz = 0;
	return lv->data[cz][cx];
}

/* ------------------------------------------------------------- level I/O */

pwno_level *pwno_level_new(void)
{
	pwno_level *lv = calloc(1, sizeof(*lv));
	if(lv == NULL) return NULL;
	pwno_level_set_tables(lv, NULL, NULL);
	return lv;
}

void pwno_level_free(pwno_level *lv)
{
	if(lv == NULL) return;
	free(lv->spheres);
	free(lv->bin_idx);
	free(lv);
}

/* level.h:85-105 (level_new) */
static void level_reset(pwno_level *lv)
{
	memset(lv->data, '.', sizeof(lv->data));
	for(int i = 0; i < 26; i++)
	{
		pwno_portal *pm = &lv->pmap[i];
		pm->x1 = pm->x2 = -1; pm->z1 = pm->z2 = -1;
		pm->rot12 = 0; pm->c1 = ';'; pm->c2 = ';';
	}
	lv->sx = lv->sz = 0;
}

int pwno_level_set_tables(pwno_level *lv, const uint8_t *data4096, const int32_t *pmap26x7)
{
	level_reset(lv);
	if(data4096 != NULL) memcpy(lv->data, data4096, 4096);
	if(pmap26x7 != NULL)
	for(int i = 0; i < 26; i++)
	{
		pwno_portal *pm = &lv->pmap[i];
		pm->x1 = pmap26x7[i*7+0]; pm->z1 = pmap26x7[i*7+1];
		pm->x2 = pmap26x7[i*7+2]; pm->z2 = pmap26x7[i*7+3];
		pm->rot12 = pmap26x7[i*7+4]; pm->c1 = pmap26x7[i*7+5]; pm->c2 = pmap26x7[i*7+6];
	}
	return 0;
}

/* util.h:128-138 */
static int cell_is_free(int c)
{
	return c == ';' || c == '$' || c == '"' || c == '#' || c == '&'
		|| c == '>' || c == '<' || c == '^' || c == ',';
}

/* the reference indexes data[z][x+-1] / data[z+-1][x] unguarded
   (util.h:140-149); inside the 4096-byte array that aliases the neighbouring
   row, which is reproduced; outside the array the cell counts as not free */
static int cell_flat(const pwno_level *lv, int x, int z)
{
	int i = z*64 + x;
	if(i < 0 || i >= 4096) return '.';
	return ((const uint8_t *)lv->data)[i];
}

static int free_dir(const pwno_level *lv, int x, int z)
{
	if(cell_is_free(cell_flat(lv, x+1, z))) return FXP;
	if(cell_is_free(cell_flat(lv, x, z+1))) return FZP;
	if(cell_is_free(cell_flat(lv, x-1, z))) return FXN;
	if(cell_is_free(cell_flat(lv, x, z-1))) return FZN;
	return FXP;
}

static int neighbour(const pwno_level *lv, int x, int z, int d)
{
	switch(d)
	{
		case FXP: return cell_flat(lv, x+1, z);
		case FZP: return cell_flat(lv, x, z+1);
		case FXN: return cell_flat(lv, x-1, z);
		default:  return cell_flat(lv, x, z-1);
	}
}

/* level.h:107-228 (level_load) */
int pwno_level_load_mem(pwno_level *lv, const char *text, int len)
{
	level_reset(lv);
	int pos = 0;
	for(int z = 0; z < 64; z++)
	{
		int x = 0;
		while(x < 64)
		{
			int c = pos < len ? (unsigned char)text[pos++] : -1;
			if(c == -1) goto done;
			if(c == '\r' || c == '\n')
			{
				if(x == 0) continue; /* level.h:129-131: blank/leftover line end */
				break;               /* level.h:132-134: end of row */
			}
			if(c == '*') { c = ';'; lv->sx = x; lv->sz = z; }
			if(c >= 'a' && c <= 'z' - 1)
			{
				/* level.h:144-161: lowercase registers, then becomes NEXT letter */
				pwno_portal *pm = &lv->pmap[c - 'a'];
				if(pm->x1 == -1) { pm->x1 = x; pm->z1 = z; }
				else if(pm->x2 == -1) { pm->x2 = x; pm->z2 = z; }
				c = (c - 'a') + 'A' + 1;
			}
			if(c >= 'A' && c <= 'Z')
			{
				pwno_portal *pm = &lv->pmap[c - 'A'];
				if(pm->x1 == -1) { pm->x1 = x; pm->z1 = z; }
				else if(pm->x2 == -1) { pm->x2 = x; pm->z2 = z; }
			}
			lv->data[z][x] = (uint8_t)c;
			x++;
		}
	}
done:
	for(int i = 0; i < 26; i++)
	{
		pwno_portal *pm = &lv->pmap[i];
		if(pm->x2 == -1) continue;
		int d1 = free_dir(lv, pm->x1, pm->z1);
		int d2 = free_dir(lv, pm->x2, pm->z2);
		pm->rot12 = (d2 - d1 + 2) & 3;
		pm->c1 = neighbour(lv, pm->x1, pm->z1, d1);
		pm->c2 = neighbour(lv, pm->x2, pm->z2, d2);
	}
	return 0;
}

int pwno_level_load_file(pwno_level *lv, const char *path)
{
	FILE *fp = fopen(path, "rb");
	if(fp == NULL) return -1;
	char *buf = malloc(1 << 20);
	int n = (int)fread(buf, 1, 1 << 20, fp);
	fclose(fp);
	int r = pwno_level_load_mem(lv, buf, n);
	free(buf);
	return r;
}

/* level.h:1-39,64-81: every live object is appended, in object order, to each
   cell of [(int)(x-r)..(int)(x+r)] x [(int)(z-r)..(int)(z+r)].  The reference
   does not bounds-check (out-of-grid cells are undefined behaviour there);
   here they are skipped. */
int pwno_level_set_spheres(pwno_level *lv, const pwno_sphere *s, int n)
{
	if(n < 0) return -1;
	free(lv->spheres); lv->spheres = NULL;
	free(lv->bin_idx); lv->bin_idx = NULL;
	lv->nspheres = n;
	if(n > 0)
	{
		lv->spheres = malloc(sizeof(*s) * n);
		memcpy(lv->spheres, s, sizeof(*s) * n);
	}
	int *cnt = calloc(4097, sizeof(int));
	for(int pass = 0; pass < 2; pass++)
	{
		if(pass == 1)
		{
			int acc = 0;
			for(int i = 0; i < 4096; i++) { lv->bin_off[i] = acc; acc += cnt[i]; cnt[i] = 0; }
			lv->bin_off[4096] = acc;
			lv->bin_cap = acc;
			lv->bin_idx = malloc(sizeof(int32_t) * (acc > 0 ? acc : 1));
		}
		for(int i = 0; i < n; i++)
		{
			int cx1 = (int)(s[i].x - s[i].r), cz1 = (int)(s[i].z - s[i].r);
			int cx2 = (int)(s[i].x + s[i].r), cz2 = (int)(s[i].z + s[i].r);
			for(int z = cz1; z <= cz2; z++)
			for(int x = cx1; x <= cx2; x++)
			{
				if(x < 0 || x >= 64 || z < 0 || z >= 64) continue;
				int c = z*64 + x;
				if(pass == 1) lv->bin_idx[lv->bin_off[c] + cnt[c]] = i;
				cnt[c]++;
			}
		}
	}
	free(cnt);
	return 0;
}

/* ------------------------------------------------------------- the trace */

typedef struct frame_ctx
{
	const pwno_level *lv;
	float sec_current;
	pwno_stats st;
	int dbg;
	uint16_t *smap;   /* this pixel's slots of the step map, or NULL */
} frame_ctx;

/* analysis aid (tools/unit_shapes.py): pwno_step_map(m) makes the next pwno_trace_rows calls store the
   walk-loop iterations of every ray segment, m[(y*w + x)*3 + depth]; NULL switches it off */
static uint16_t *step_map = NULL;
void pwno_step_map(uint16_t *m) { step_map = m; }

/* debugging aid for oracle-vs-reference hunts: pwno_debug_pixel(x,y) makes the
   next renders print the event chain of that pixel to stderr */
static int dbg_x = -1, dbg_y = -1;
void pwno_debug_pixel(int x, int y) { dbg_x = x; dbg_y = y; }
#define DBG(fc, ...) do { if((fc)->dbg) fprintf(stderr, __VA_ARGS__); } while(0)

enum { EV_EXHAUSTED = 0, EV_WALL, EV_SPHERE };

typedef struct hit
{
	int ev;
	v4 ray, pos;      /* walked ray and hit position (wall) or aux_pos (sphere) */
	v4 norm;          /* sphere normal */
	int ldir;         /* wall face; -1 for a sphere */
	float refl;       /* sphere reflectivity */
	float fog;
	float dist;
	v4 col;           /* wall base colour (before icol/diffuse) or sphere colour */
} hit;

#define COL_CEIL  v4_set(30.0f, 30.0f, 0.0f, 0.0f)
#define COL_FLOOR v4_set(1.0f, 1.0f, 1.0f, 0.0f)
#define COL_WALL  v4_set(0.8f, 0.8f, 1.0f, 0.0f)

/* trace.h:186-679 without the tail calls: walk until something is hit */
static void walk(frame_ctx *fc, v4 from, v4 iray, hit *h)
{
	const pwno_level *lv = fc->lv;
	float cdist = 0.0f, fog = 0.0f, fogbeg = 0.0f;
	float aux_dist = -1.0f, aux_refl = 0.25f;
	v4 aux_pos = v4_set(0, 0, 0, 0), aux_norm = v4_set(0, 0, 0, 0), aux_col = v4_set(1, 1, 1, 1);

	fc->st.rays++;

	/* trace.h:212-222 */
	v4 pos = from;
	v4 ray = v4_normalise(iray);
	int cx = (int)from.x;
	int cz = (int)from.z;
	if(ray.x > -EPS && ray.x < EPS) ray.x = (ray.x < 0.0f ? -EPS : EPS);
	if(ray.y > -EPS && ray.y < EPS) ray.y = (ray.y < 0.0f ? -EPS : EPS);
	if(ray.z > -EPS && ray.z < EPS) ray.z = (ray.z < 0.0f ? -EPS : EPS);

	/* trace.h:225-227: signs of the UN-normalised input */
	int gx = (iray.x < 0 ? -1 : 1);
	int gy = (iray.y < 0 ? -1 : 1);
	int gz = (iray.z < 0 ? -1 : 1);

	/* trace.h:230-241 */
	float iax = pwn_tab_rcp(__builtin_fabsf(ray.x));
	float iay = pwn_tab_rcp(__builtin_fabsf(ray.y));
	float iaz = pwn_tab_rcp(__builtin_fabsf(ray.z));
	float wx = pos.x - (float)cx;
	float wy = pos.y - 0.0f;
	float wz = pos.z - (float)cz;
	if(ray.x >= 0.0f) wx = 1.0f - wx;
	if(ray.y >= 0.0f) wy = 1.0f - wy;
	if(ray.z >= 0.0f) wz = 1.0f - wz;
	wx *= iax; wy *= iay; wz *= iaz;

	int cell = cell_at(lv, cx, cz);
	int ldir = FYN;

#define AUX_HIT() (aux_dist != -1.0f && cdist > aux_dist)
#define RET_SPHERE() do { h->ev = EV_SPHERE; h->ray = ray; h->pos = aux_pos; h->norm = aux_norm; \
	h->ldir = -1; h->refl = aux_refl; h->fog = fog; h->dist = aux_dist; h->col = aux_col; return; } while(0)
#define RET_WALL(c) do { h->ev = EV_WALL; h->ray = ray; h->pos = pos; h->ldir = ldir; \
	h->fog = fog; h->dist = cdist; h->col = (c); return; } while(0)
/* trace.h:156-184 */
#define THROUGH(gxa) do { float t_; \
	if(wy < wx && wy < wz) { t_ = wy; ldir = (gy < 0 ? FYN : FYP); } \
	else if(wx < wz) { t_ = wx; ldir = ((gxa) < 0 ? FXN : FXP); } \
	else { t_ = wz; ldir = (gz < 0 ? FZN : FZP); } \
	cdist += t_; pos = v4_add(v4_scale(t_, ray), pos); } while(0)
/* assoc (fog): the reference build evaluates  fog += cdist - fogbeg  as
   (fog - fogbeg) + cdist  and  fog += aux_dist - fogbeg  as
   (fog + aux_dist) - fogbeg  (both the inlined and the stand-alone copy) */
/* trace.h:331-340 */
#define ADVANCE_XZ() do { \
	if(ldir == FXN || ldir == FXP) { wy -= wx; wz -= wx; wx = iax; cx += gx; } \
	else { wx -= wz; wy -= wz; wz = iaz; cz += gz; } } while(0)

	for(int maxsteps = 1000; maxsteps > 0; maxsteps--)
	{
		fc->st.steps++;

		/* trace.h:252-296 */
		if(cx >= 0 && cx < 64 && cz >= 0 && cz < 64)
		{
			int c = cz*64 + cx;
			for(int k = lv->bin_off[c]; k < lv->bin_off[c+1]; k++)
			{
				const pwno_sphere *sp = &lv->spheres[lv->bin_idx[k]];
				fc->st.sphere_tests++;
				v4 spos = v4_set(sp->x, sp->y, sp->z, 1.0f);
				float rad2 = sp->r * sp->r;
				v4 rel = v4_sub(spos, pos);
				float d2 = v4_dot(rel, rel);
				float dt = v4_dot(rel, ray);
				if(dt > 0.0f)
				{
					float calc = d2 - dt*dt;
					if(calc < rad2)
					{
						float sd2 = 1.0f - calc/rad2;
						float sdist = __builtin_sqrtf(d2) - __builtin_sqrtf(sd2);
						if(aux_dist == -1.0f || sdist + cdist < aux_dist)
						{
							aux_dist = sdist + cdist;
							aux_pos = v4_add(pos, v4_scale(sdist, ray));
							aux_norm = v4_normalise(v4_sub(aux_pos, spos));
							float diff = -v4_dot(ray, aux_norm);
							if(diff < 0.0f) diff = 0.0f;
							const float amb = 0.2f;
							aux_refl = sp->refl;
							diff = amb + (1.0f - amb)*diff;
							aux_col = v4_scale(diff, v4_set(sp->cb, sp->cg, sp->cr, 0.0f));
						}
					}
				}
			}
		}

		int this_cell = cell;
		DBG(fc, "  step cell '%c' (%d,%d) pos %a %a %a ray %a %a %a w %a %a %a cdist %a aux %a ldir %d\n", this_cell, cx, cz,
			pos.x, pos.y, pos.z, ray.x, ray.y, ray.z, wx, wy, wz, cdist, aux_dist, ldir);
		switch(this_cell)
		{
			case ';': case '$': case '"':
				/* trace.h:302-352 */
				if(this_cell == '$') fogbeg = cdist;
				THROUGH(gx);
				if(AUX_HIT())
				{
					if(this_cell == '$' && aux_dist > fogbeg) fog = (fog + aux_dist) - fogbeg;
					RET_SPHERE();
				}
				if(this_cell == '$') fog = (fog - fogbeg) + cdist;
				if(ldir == FYN || ldir == FYP)
					RET_WALL(gy > 0 ? COL_CEIL : COL_FLOOR);
				ADVANCE_XZ();
				cell = cell_at(lv, cx, cz);
				if(this_cell == '"' && (cell == '#' || cell == '&'))
				{
					pos.y += 1.0f;
					if(gy < 0) wy += iay; else wy -= iay;
				}
				break;

			case '#': case '&':
			{
				/* trace.h:354-441 */
				if(gy > 0) wy += iay;
				if(this_cell == '&') fogbeg = cdist;
				THROUGH(gx);
				if(AUX_HIT())
				{
					if(this_cell == '&' && aux_dist > fogbeg) fog = (fog + aux_dist) - fogbeg;
					RET_SPHERE();
				}
				if(this_cell == '&') fog = (fog - fogbeg) + cdist;
				if(ldir == FYN || ldir == FYP)
					RET_WALL(gy > 0 ? COL_CEIL : COL_FLOOR);
				ADVANCE_XZ();
				if(gy > 0) wy -= iay;
				cell = cell_at(lv, cx, cz);
				if(cell == '"')
				{
					pos.y -= 1.0f;
					if(gy > 0) wy += iay; else wy -= iay;
				}
				int xcell = cell;
				if(xcell >= 'A' && xcell <= 'Z')
				{
					const pwno_portal *pm = &lv->pmap[xcell - 'A'];
					if(pm->x1 == cx && pm->z1 == cz) xcell = pm->c2;
					else if(pm->x2 == cx && pm->z2 == cz) xcell = pm->c1;
				}
				if(pos.y < 0.0f || pos.y > 1.0f)
				{
					if(xcell == '#' || xcell == '&') { /* open above: carry on */ }
					else
					{
						if(xcell == '"')
						{
							pos.y += 1.0f;
							if(gy > 0) wy -= iay; else wy += iay;
						}
						RET_WALL(COL_WALL);
					}
				}
				break;
			}

			case '>': case '<': case ',': case '^':
			{
				/* trace.h:443-505 */
				const float ramp = 0.5f;
				switch(this_cell)
				{
					case '>': ray.y -= ramp * ray.x; break;
					case '<': ray.y += ramp * ray.x; break;
					case ',': ray.y -= ramp * ray.z; break;
					default:  ray.y += ramp * ray.z; break;
				}
				wy = pos.y;
				if(ray.y >= 0.0f) wy = 1.0f - wy;
				wy *= 1.0f / (ray.y < 0.0f ? -ray.y : ray.y);
				if(AUX_HIT()) RET_SPHERE();
				THROUGH(gy); /* sic: trace.h:470 passes gy for gx */
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
				switch(this_cell)
				{
					case '>': ray.y += ramp * ray.x; break;
					case '<': ray.y -= ramp * ray.x; break;
					case ',': ray.y += ramp * ray.z; break;
					default:  ray.y -= ramp * ray.z; break;
				}
				wy = pos.y;
				if(ray.y >= 0.0f) wy = 1.0f - wy;
				wy *= iay;
				cell = cell_at(lv, cx, cz);
				break;
			}

			default:
				/* trace.h:507-664 */
				if(cell >= 'A' && cell <= 'Z')
				{
					const pwno_portal *pm = &lv->pmap[cell - 'A'];
					int rot;
					if(pm->x2 == -1)
					{
						if(AUX_HIT()) RET_SPHERE();
						RET_WALL(COL_WALL);
					}
					if(pm->x1 == cx && pm->z1 == cz)
					{
						cx = pm->x2; cz = pm->z2;
						pos.x += (float)(pm->x2 - pm->x1);
						pos.z += (float)(pm->z2 - pm->z1);
						rot = (-pm->rot12) & 3;
					}
					else if(pm->x2 == cx && pm->z2 == cz)
					{
						cx = pm->x1; cz = pm->z1;
						pos.x -= (float)(pm->x2 - pm->x1);
						pos.z -= (float)(pm->z2 - pm->z1);
						rot = pm->rot12 & 3;
					}
					else
					{
						if(AUX_HIT()) RET_SPHERE();
						RET_WALL(v4_set(5.0f, 0.0f, 5.0f, 0.0f));
					}
					fc->st.portals++;

					/* trace.h:561-622 */
					float trx = pos.x, trz = pos.z, trvx = ray.x, trvz = ray.z;
					int tgx = gx, tgz = gz;
					float t;
					/* assoc: source is (c+.5) +- (t - (c'+.5)); the compiled reference
					   (both the inlined primary copy and the stand-alone trace_ray)
					   cancels the halves: see the orders below.  rot 2 is as written. */
					float fcx = (float)cx, fcz = (float)cz;
					float ccx = fcx + 0.5f, ccz = fcz + 0.5f;
					ldir = (ldir - rot) & 3;
					switch(rot)
					{
						case 1:
							pos.x = (trz + fcx) - fcz;
							pos.z = (1.0f - trx) + (fcx + fcz);
							ray.x = trvz; ray.z = -trvx;
							gx = tgz; gz = -tgx;
							t = wx; wx = wz; wz = t;
							t = iax; iax = iaz; iaz = t;
							break;
						case 2:
							pos.x = ccx*2.0f - trx;
							pos.z = ccz*2.0f - trz;
							ray.x = -trvx; ray.z = -trvz;
							gx = -tgx; gz = -tgz;
							break;
						case 3:
							pos.x = (1.0f - trz) + (fcx + fcz);
							pos.z = (fcz + trx) - fcx;
							ray.x = -trvz; ray.z = trvx;
							gx = -tgz; gz = tgx;
							t = wx; wx = wz; wz = t;
							t = iax; iax = iaz; iaz = t;
							break;
						default: break;
					}
					/* trace.h:624-647 */
					switch(ldir)
					{
						case FZP: cz++; pos.z += 1.0f; break;
						case FXN: cx--; pos.x -= 1.0f; break;
						case FZN: cz--; pos.z -= 1.0f; break;
						default:  cx++; pos.x += 1.0f; break;
					}
					cell = cell_at(lv, cx, cz);
					break;
				}
				if(AUX_HIT()) RET_SPHERE();
				RET_WALL(ldir == FYP ? COL_CEIL : COL_WALL);
		}

		/* trace.h:668-673 */
		if(AUX_HIT()) RET_SPHERE();
	}

	/* trace.h:677-678: the walked ray is returned as the colour */
	fc->st.exhausted++;
	h->ev = EV_EXHAUSTED;
	h->ray = ray;
#undef AUX_HIT
#undef RET_SPHERE
#undef RET_WALL
#undef THROUGH
#undef ADVANCE_XZ
}

/* one pixel: trace_ray(0,...) of screen.h:22-24 with its recursion unrolled.
   Returns colour; *dist written only if the primary ray hit something. */
static v4 trace_pixel(frame_ctx *fc, uint32_t *seed, v4 from, v4 iray, float *dist)
{
	v4 icol = v4_set(1.0f, 1.0f, 1.0f, 1.0f);
	float st_refl[REFLECT_MAX], st_fog[REFLECT_MAX];
	v4 st_col[REFLECT_MAX];
	int depth = 0;
	v4 value;

	for(;;)
	{
		hit h;
		memset(&h, 0, sizeof(h));
		DBG(fc, " segment %d from %a %a %a %a dir %a %a %a %a\n", depth, from.x, from.y, from.z, from.w, iray.x, iray.y, iray.z, iray.w);
		const int64_t steps0 = fc->st.steps;
		walk(fc, from, iray, &h);
		if(fc->smap != NULL) fc->smap[depth] = (uint16_t)(fc->st.steps - steps0);
		DBG(fc, " -> ev %d ldir %d dist %a fog %a pos %a %a %a col %a %a %a refl %a\n", h.ev, h.ldir, h.dist, h.fog, h.pos.x, h.pos.y, h.pos.z, h.col.x, h.col.y, h.col.z, h.refl);
		if(h.ev == EV_EXHAUSTED) { value = h.ray; break; }
		if(depth == 0) *dist = h.dist;

		v4 col;
		float refl;
		if(h.ev == EV_WALL)
		{
			/* trace.h:108-154 */
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
			diffuse = (1.0f - amb)*diffuse + amb;
			col = v4_scale(diffuse, col);
			refl = (h.ldir == FYN ? 0.7f : 0.25f);
		}
		else
		{
			col = h.col;
			refl = h.refl;
		}

		/* trace.h:3-7 */
		if(depth >= REFLECT_MAX || refl == 0.0f) { value = col; break; }

		/* trace.h:9-75 */
		v4 ray = h.ray, pos = h.pos;
		switch(h.ldir)
		{
			case FXP: ray.x = -ray.x; pos.x -= 0.001f; break;
			case FXN: ray.x = -ray.x; pos.x += 0.001f; break;
			case FZP: ray.z = -ray.z; pos.z -= 0.001f; break;
			case FZN: ray.z = -ray.z; pos.z += 0.001f; break;
			case FYP: ray.y = -ray.y; pos.y -= 0.001f; break;
			case FYN:
			{
				pos.y -= 0.001f;
				const float pi = (float)3.14159265358979323846;
				float ang = (pi*2.0f)*(
					(pwn_sinf((pi*0.5f)*pos.x) + pwn_cosf((pi*0.5f)*pos.z))
					+ fc->sec_current);
				v4 norm = v4_normalise(v4_set(pwn_sinf(ang), 38.0f, pwn_cosf(ang), 0.0f));
				float rmul = -2.0f * ((ray.x*norm.x + ray.y*norm.y) + ray.z*norm.z);
				ray = v4_normalise(v4_add(v4_scale(rmul, norm), ray));
				break;
			}
			default:
			{
				pos = v4_sub(pos, v4_scale(0.001f, ray));
				v4 norm = h.norm;
				float rmul = -2.0f * ((ray.x*norm.x + ray.y*norm.y) + ray.z*norm.z);
				ray = v4_normalise(v4_add(v4_scale(rmul, norm), ray));
				break;
			}
		}
		/* trace.h:77-84 */
		ray.x += lcg_fs(seed) * REFLECT_BLUR_F;
		ray.y += lcg_fs(seed) * REFLECT_BLUR_F;
		lcg_fs(seed);
		ray.z += lcg_fs(seed) * REFLECT_BLUR_F;
		lcg_fs(seed);

		st_refl[depth] = refl; st_fog[depth] = h.fog; st_col[depth] = col;
		depth++;
		icol = col;
		from = pos;
		iray = ray;
	}

	/* trace.h:91-101, innermost first */
	while(depth > 0)
	{
		depth--;
		float refl = st_refl[depth];
		value = v4_add(v4_scale(refl, value), v4_scale(1.0f - refl, st_col[depth]));
		if(st_fog[depth] != 0.0f)
		{
			float f = pwn_expf(-0.6f * st_fog[depth]);
			float g = 1.0f - f;
			value = v4_add(v4_scale(f, value), v4_set(g, g, g, g));
		}
	}
	return value;
}

/* screen.h:43-57.  assoc: the compiled reference forms
   rayb = (cam.x + cam.z) + (-yrat)*cam.y */
void pwno_frame_setup(int w, int h, const float cam[16], float rayb[4], float rdx[4], float rdy[4])
{
	float dimx = (float)w, dimy = (float)h;
	float yrat = (-dimy) / dimx;
	float xsrat = -2.0f / dimx;
	float ysrat = (yrat + yrat) / dimy;
	for(int i = 0; i < 4; i++)
	{
		rayb[i] = (cam[0+i] + cam[8+i]) + (-yrat) * cam[4+i];
		rdx[i] = xsrat * cam[0+i];
		rdy[i] = ysrat * cam[4+i];
	}
}

/* screen.h:19-21 */
uint32_t pwno_pixel_seed(int x, int y, int rwidth)
{
	uint32_t s = (uint32_t)x + (uint32_t)y*(uint32_t)y*((uint32_t)rwidth + 1u);
	s *= s * s;
	s *= s * s;
	return s;
}

static unsigned fast_math_on(void)
{
	/* the reference executable is linked with crtfastmath.o: FTZ|DAZ */
	unsigned csr = _mm_getcsr();
	_mm_setcsr(csr | 0x8040);
	return csr;
}

int pwno_trace_rows(const pwno_level *lv, int w, int h, int y0, int y1,
	const float cam[16], float sec_current, int nthreads,
	uint32_t *sbuf, float *zbuf, pwno_stats *stats)
{
	if(lv == NULL || w <= 0 || h <= 0 || y0 < 0 || y1 > h || y0 > y1) return -1;
	float rb[4], dx[4], dy[4];
	pwno_frame_setup(w, h, cam, rb, dx, dy);
	v4 rayb = v4_set(rb[0], rb[1], rb[2], rb[3]);
	v4 rdx = v4_set(dx[0], dx[1], dx[2], dx[3]);
	v4 rdy = v4_set(dy[0], dy[1], dy[2], dy[3]);
	v4 from = v4_set(cam[12], cam[13], cam[14], cam[15]);
	pwno_stats tot = { 0, 0, 0, 0, 0 };
	if(nthreads <= 0) nthreads = omp_get_max_threads();

#pragma omp parallel num_threads(nthreads)
	{
		unsigned csr = fast_math_on();
		frame_ctx fc;
		fc.lv = lv; fc.sec_current = sec_current; fc.dbg = 0; fc.smap = NULL;
		memset(&fc.st, 0, sizeof(fc.st));
#pragma omp for schedule(dynamic, 4)
		for(int y = y0; y < y1; y++)
		{
			/* screen.h:6-26; assoc: rayl = (cx*rdx + rayb) + y*rdy */
			for(int cx = 0; cx < w; cx += 32)
			{
				v4 rayl = v4_add(v4_add(v4_scale((float)cx, rdx), rayb), v4_scale((float)y, rdy));
				for(int x = cx; x < cx + 32 && x < w; x++)
				{
					rayl = v4_add(rayl, rdx);
					uint32_t seed = pwno_pixel_seed(x, y, w);
					float dist = 0.0f;
					float *dp = zbuf != NULL ? &zbuf[(size_t)y*w + x] : &dist;
					fc.dbg = (x == dbg_x && y == dbg_y);
					fc.smap = step_map != NULL ? step_map + ((size_t)y*w + x)*(REFLECT_MAX + 1) : NULL;
					v4 c = trace_pixel(&fc, &seed, from, rayl, dp);
					DBG(&fc, "pixel %d,%d value %a %a %a %a -> %08x\n", x, y, c.x, c.y, c.z, c.w, col_pack(c));
					sbuf[(size_t)y*w + x] = col_pack(c);
				}
			}
		}
#pragma omp critical
		{
			tot.rays += fc.st.rays; tot.steps += fc.st.steps; tot.portals += fc.st.portals;
			tot.sphere_tests += fc.st.sphere_tests; tot.exhausted += fc.st.exhausted;
		}
		_mm_setcsr(csr);
	}
	if(stats != NULL) *stats = tot;
	return 0;
}

static inline uint32_t avg_u8x4(uint32_t a, uint32_t b)
{
	/* _mm_avg_epu8: per byte (p+q+1)>>1 */
	uint32_t r = 0;
	for(int k = 0; k < 32; k += 8)
		r |= ((((a >> k) & 0xff) + ((b >> k) & 0xff) + 1) >> 1) << k;
	return r;
}

/* screen.h:77-121 */
int pwno_blur_rows(int w, int h, int y0, int y1, int nthreads,
	const uint32_t *tsbuf, const float *zbuf, uint32_t *sbuf)
{
	if(w <= 0 || h <= 0 || y0 < 0 || y1 > h || y0 > y1) return -1;
	if(nthreads <= 0) nthreads = omp_get_max_threads();
	const float fstr = 0.002f * (float)h;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
	for(int cy = y0; cy < y1; cy++)
	{
		unsigned csr = fast_math_on();
		uint32_t seed = (uint32_t)cy*(uint32_t)cy + 415135u;
		for(int cx = 0; cx < w - 3; cx += 4)
		{
			uint32_t tap[4][4];
			for(int i = 0; i < 4; i++)
			for(int j = 0; j < 4; j++)
			{
				float z = zbuf[(size_t)cy*w + cx + j] - 1.0f;
				/* x = cx + j + randfs*fstr*z : int + float, then truncated */
				int x = (int)((float)(cx + j) + (lcg_fs(&seed) * fstr) * z);
				int y = (int)((float)cy + (lcg_fs(&seed) * fstr) * z);
				if(x < 0) x = 0;
				if(y < 0) y = 0;
				if(x >= w) x = w - 1;
				if(y >= h) y = h - 1;
				tap[i][j] = tsbuf[(size_t)y*w + x];
			}
			for(int j = 0; j < 4; j++)
				sbuf[(size_t)cy*w + cx + j] = avg_u8x4(avg_u8x4(tap[0][j], tap[1][j]), avg_u8x4(tap[2][j], tap[3][j]));
		}
		_mm_setcsr(csr);
	}
	return 0;
}

uint32_t pwno_blur_seed_at(int cy, int groups)
{
	uint32_t seed = (uint32_t)cy*(uint32_t)cy + 415135u;
	for(int i = 0; i < groups*32; i++) lcg_next(&seed);
	return seed;
}

int pwno_render(const pwno_level *lv, int w, int h, const float cam[16], float sec_current,
	int blur_passes, int nthreads, uint32_t *sbuf, float *zbuf, pwno_stats *stats)
{
	if(blur_passes < 0) return -1;
	float *ztmp = NULL;
	if(zbuf == NULL && blur_passes > 0)
	{
		ztmp = calloc((size_t)w*h, 4);
		zbuf = ztmp;
	}
	int r = pwno_trace_rows(lv, w, h, 0, h, cam, sec_current, nthreads, sbuf, zbuf, stats);
	if(r == 0 && blur_passes > 0)
	{
		uint32_t *ts = malloc((size_t)w*h*4);
		for(int p = 0; p < blur_passes && r == 0; p++)
		{
			memcpy(ts, sbuf, (size_t)w*h*4);
			r = pwno_blur_rows(w, h, 0, h, nthreads, ts, zbuf, sbuf);
		}
		free(ts);
	}
	free(ztmp);
	return r;
}

/* screen.h:126-149.  The destination pointer advances by w*scale per source
   row plus pitch*(scale-1) (screen.h:132,138-139), i.e. source row py starts
   at py*(w*scale + pitch*(scale-1)) words: equal to py*scale*pitch only when
   pitch == w*scale, which is what SDL hands the reference.  Reproduced as is. */
int pwno_upscale(const uint32_t *src, int w, int h, int scale, int pitch_bytes, uint32_t *dst)
{
	if(scale <= 0 || pitch_bytes < w*scale*4) return -1;
	size_t pitch = (size_t)pitch_bytes / 4;
	size_t rowadv = (size_t)w*scale + pitch*(size_t)(scale - 1);
	for(int py = 0; py < h; py++)
	for(int px = 0; px < w; px++)
	{
		uint32_t v = src[(size_t)py*w + px];
		for(int y = 0; y < scale; y++)
		for(int x = 0; x < scale; x++)
			dst[(size_t)py*rowadv + (size_t)y*pitch + (size_t)px*scale + x] = v;
	}
	return 0;
}

/* ------------------------------------------------------ KAT entry points */

uint32_t pwno_col_ftoint(const float v[4]) { return col_pack(v4_set(v[0], v[1], v[2], v[3])); }
void pwno_normalise(const float in[4], float out[4])
{
	v4 r = v4_normalise(v4_set(in[0], in[1], in[2], in[3]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
float pwno_dot(const float a[4], const float b[4])
{
	return v4_dot(v4_set(a[0], a[1], a[2], a[3]), v4_set(b[0], b[1], b[2], b[3]));
}
float pwno_rcp(float x) { return pwn_tab_rcp(x); }
float pwno_rsqrt(float x) { return pwn_tab_rsqrt(x); }
float pwno_sinf(float x) { return pwn_sinf(x); }
float pwno_cosf(float x) { return pwn_cosf(x); }
float pwno_expf(float x) { return pwn_expf(x); }
uint32_t pwno_randi(uint32_t *seed) { return lcg_next(seed); }
float pwno_randfu(uint32_t *seed) { return lcg_fu(seed); }
float pwno_randfs(uint32_t *seed) { return lcg_fs(seed); }

/* SURVEY.md App. B6 */
uint64_t pwno_fnv64(const uint32_t *p, int64_t n)
{
	uint64_t hh = 1469598103934665603ULL;
	for(int64_t i = 0; i < n; i++) { hh ^= p[i]; hh *= 1099511628211ULL; }
	return hh;
}

/* util.h:95-110 / 79-93 with this library's own sinf/cosf */
void pwno_mat4_roty(float m[16], float ang)
{
	float vs = pwn_sinf(ang), vc = pwn_cosf(ang);
	float vxx = m[0], vxz = m[2], vzx = m[8], vzz = m[10];
	m[0] = vc*vxx + vs*vxz; m[2] = vc*vxz - vs*vxx;
	m[8] = vc*vzx + vs*vzz; m[10] = vc*vzz - vs*vzx;
}
void pwno_mat4_rotx(float m[16], float ang)
{
	float vs = pwn_sinf(ang), vc = pwn_cosf(ang);
	float vyy = m[5], vyz = m[6], vzy = m[9], vzz = m[10];
	m[5] = vc*vyy + vs*vyz; m[6] = vc*vyz - vs*vyy;
	m[9] = vc*vzy + vs*vzz; m[10] = vc*vzz - vs*vzy;
}
