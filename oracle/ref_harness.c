/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Builds the reference's OWN hot-path sources (defs.h, util.h, trace.h,
 * screen.h, level.h) unmodified, from where they lie under /root/reference,
 * behind a small C API that tests and tools can call through ctypes.
 * Nothing from the reference is copied here: the headers are #included by
 * path (-I/root/reference) and the resulting .so lands in oracle/_ref/
 * (git-ignored).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * This file takes the role of the reference's main.c for the render path:
 * main.c itself needs SDL 1.2 and Lua 5.1 (absent here) and is NOT built.
 * It supplies
 *   - the system includes of main.c:1-17 minus SDL/Lua,
 *   - the globals of main.c:26-34 (rwidth, rheight, rscale, screen, sbuf,
 *     tsbuf, zbuf, lvroot).  `screen` is only ever dereferenced as
 *     screen->pitch and screen->pixels (screen.h:129,136), so it is declared
 *     here as a pointer to this harness's own two-member surface struct,
 *   - the header order of main.c:24,36-39 (script.h is skipped: Lua).
 *
 * Build variants (oracle/Makefile):
 *   libpwnref_hw.so   reference flags, native _mm_rcp_ps/_mm_rsqrt_ps
 *                     (result depends on the host CPU's approximation tables)
 *   libpwnref_tab.so  -DPWNREF_TABLES: the two intrinsics are redirected to
 *                     the 2048-entry table emulation captured from the Intel
 *                     survey/build host, so the output is host independent
 *   -DPWNREF_BLUR_RUNTIME makes POSTPROC_BLUR (defs.h:9) a run-time pass
 *                     count so one library yields pre- and post-blur frames.
 */
#include <string.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <errno.h>
#include <assert.h>
#include <math.h>
#include <sys/types.h>
#include <mmintrin.h>
#include <xmmintrin.h>
#include <emmintrin.h>
#include <omp.h>

#ifdef PWNREF_TABLES
#include "approx_tables.h"
static inline __m128 pwnref_tab_rcp_ps(__m128 v)
{
	float a[4] __attribute__((aligned(16)));
	_mm_store_ps(a, v);
	for(int i = 0; i < 4; i++) a[i] = pwn_tab_rcp(a[i]);
	return _mm_load_ps(a);
}
static inline __m128 pwnref_tab_rsqrt_ps(__m128 v)
{
	float a[4] __attribute__((aligned(16)));
	_mm_store_ps(a, v);
	for(int i = 0; i < 4; i++) a[i] = pwn_tab_rsqrt(a[i]);
	return _mm_load_ps(a);
}
#define _mm_rcp_ps(x) pwnref_tab_rcp_ps(x)
#define _mm_rsqrt_ps(x) pwnref_tab_rsqrt_ps(x)
#endif

#ifdef PWNREF_COUNTERS
/* rays, cell steps, portal crossings, sphere tests, maxsteps exhaustions;
   the increments are spliced into a throw-away copy of trace.h by the Makefile */
long long pwnref_cnt[8];
#define PWNREF_CNT(i) __atomic_fetch_add(&pwnref_cnt[i], 1, __ATOMIC_RELAXED)
#endif

/* the harness's own sink type; see header comment */
typedef struct pwnref_surface { int pitch; void *pixels; } pwnref_surface;

#include "defs.h"

#ifdef PWNREF_BLUR_RUNTIME
int pwnref_blur_passes = 1;
#undef POSTPROC_BLUR
#define POSTPROC_BLUR pwnref_blur_passes
#endif

/* main.c:26-34 */
int rwidth = DEF_RWIDTH;
int rheight = DEF_RHEIGHT;
int rscale = DEF_SCALE;
pwnref_surface *screen = NULL;
uint32_t *sbuf = NULL;
uint32_t *tsbuf = NULL;
float *zbuf = NULL;
level *lvroot = NULL;

#include "util.h"
#include "trace.h"
#include "screen.h"
#include "level.h"

/* ------------------------------------------------------------------ API */

typedef struct pwnref_sphere { float r, refl, x, y, z, cb, cg, cr; } pwnref_sphere;

static unsigned pwnref_set_fast_math(unsigned on, unsigned old)
{
	/* what crtfastmath.o does for a -ffast-math executable: FTZ|DAZ */
	unsigned cur = _mm_getcsr();
	if(on) _mm_setcsr(cur | 0x8040);
	else _mm_setcsr((cur & ~0x8040u) | (old & 0x8040u));
	return cur;
}

int pwnref_variant(void)
{
	int v = 0;
#ifdef PWNREF_TABLES
	v |= 1;
#endif
#ifdef PWNREF_BLUR_RUNTIME
	v |= 2;
#endif
#ifdef PWNREF_COUNTERS
	v |= 4;
#endif
	return v;
}

int pwnref_load_level(const char *path)
{
	fflush(stdout);
	FILE *save = stdout;
	/* level_load chats on stdout (level.h:220,225); keep test logs clean */
	FILE *nul = fopen("/dev/null", "w");
	if(nul != NULL) stdout = nul;
	lvroot = level_load(path);
	if(nul != NULL) { fflush(nul); stdout = save; fclose(nul); }
	return lvroot == NULL ? -1 : 0;
}

/* pmap_out: 26 x {x1,z1,x2,z2,rot12,c1,c2} as int32 */
int pwnref_get_level(uint8_t *data_out, int32_t *pmap_out, int32_t *spawn_out)
{
	if(lvroot == NULL) return -1;
	memcpy(data_out, lvroot->data, 64*64);
	for(int i = 0; i < 26; i++)
	{
		portal *pm = &lvroot->pmap[i];
		pmap_out[i*7+0] = pm->x1; pmap_out[i*7+1] = pm->z1;
		pmap_out[i*7+2] = pm->x2; pmap_out[i*7+3] = pm->z2;
		pmap_out[i*7+4] = (pm->x2 == -1 ? 0 : pm->rot12);
		pmap_out[i*7+5] = pm->c1; pmap_out[i*7+6] = pm->c2;
		/* z1/z2 of never-seen endpoints are uninitialised in the reference
		   (level_new only sets x1,x2,c1,c2: level.h:94-99); normalise */
		if(pm->x1 == -1) pmap_out[i*7+1] = -1;
		if(pm->x2 == -1) pmap_out[i*7+3] = -1;
	}
	spawn_out[0] = lvroot->sx; spawn_out[1] = lvroot->sz;
	return 0;
}

/* install tables produced elsewhere (e.g. by the build's own loader) */
int pwnref_set_level(const uint8_t *data_in, const int32_t *pmap_in)
{
	lvroot = level_new();
	memcpy(lvroot->data, data_in, 64*64);
	for(int i = 0; i < 26; i++)
	{
		portal *pm = &lvroot->pmap[i];
		pm->x1 = pmap_in[i*7+0]; pm->z1 = pmap_in[i*7+1];
		pm->x2 = pmap_in[i*7+2]; pm->z2 = pmap_in[i*7+3];
		pm->rot12 = pmap_in[i*7+4];
		pm->c1 = (char)pmap_in[i*7+5]; pm->c2 = (char)pmap_in[i*7+6];
	}
	for(int z = 0; z < 64; z++)
	for(int x = 0; x < 64; x++)
	{
		/* lvbase is a static singleton: keep any bins already allocated */
		lvroot->parts_num[z][x] = 0;
	}
	return 0;
}

/* objects exactly as script.h:20-32 fills them (col.a stays 0: lvbase is static) */
int pwnref_set_spheres(const pwnref_sphere *s, int n)
{
	if(lvroot == NULL) return -1;
	if(n < 0 || n > OBJ_MAX) return -2;
	lvroot->objs_num = 0;
	for(int i = 0; i < n; i++)
	{
		part *pt = level_obj_new(lvroot);
		if(pt == NULL) return -3;
		memset(pt, 0, sizeof(*pt));
		pt->typ = P_SPHERE;
		pt->sph.r = s[i].r;
		pt->sph.refl = s[i].refl;
		pt->sph.pos.v.x = s[i].x;
		pt->sph.pos.v.y = s[i].y;
		pt->sph.pos.v.z = s[i].z;
		pt->sph.pos.v.w = 1.0f;
		pt->sph.col.c.b = s[i].cb;
		pt->sph.col.c.g = s[i].cg;
		pt->sph.col.c.r = s[i].cr;
	}
	return 0;
}

/* per-cell sphere bins after level_prepare_render: counts[4096], then for
   each cell the object indices in list order, up to cap entries total */
int pwnref_get_bins(uint16_t *counts, int32_t *idx, int cap)
{
	if(lvroot == NULL) return -1;
	level_prepare_render(lvroot);
	int k = 0;
	for(int z = 0; z < 64; z++)
	for(int x = 0; x < 64; x++)
	{
		counts[z*64+x] = lvroot->parts_num[z][x];
		for(int i = 0; i < lvroot->parts_num[z][x]; i++)
		{
			if(k >= cap) return -2;
			idx[k++] = (int32_t)(lvroot->parts[z][x][i] - lvroot->objs);
		}
	}
	return k;
}

#ifdef PWNREF_COUNTERS
void pwnref_get_counters(long long *out) { memcpy(out, pwnref_cnt, sizeof(pwnref_cnt)); }
void pwnref_reset_counters(void) { memset(pwnref_cnt, 0, sizeof(pwnref_cnt)); }
#endif

/*
 * One frame exactly as mainloop does it (main.c:95,107):
 * level_prepare_render + trace_screen_centred(lv,0,0,w,h,&cam).
 * cam = 16 floats, rows x,y,z,w (defs.h:46-52).  zbuf is zero-filled first
 * (the reference leaves it uninitialised).  blur_passes is honoured only by
 * PWNREF_BLUR_RUNTIME builds (else it must equal the compiled POSTPROC_BLUR=1).
 */
int pwnref_render(int w, int h, const float *cam16, float sec, int nthreads,
	int blur_passes, uint32_t *sbuf_out, float *zbuf_out)
{
	if(lvroot == NULL) return -1;
	if(w <= 0 || h <= 0) return -2;
#ifdef PWNREF_BLUR_RUNTIME
	pwnref_blur_passes = blur_passes;
#else
	if(blur_passes != 1) return -4;
#endif
	if(blur_passes > 0 && (w & 3) != 0) return -5; /* screen.h:88,117 aligned store */

	size_t n = (size_t)w*(size_t)h;
	uint32_t *sb = NULL, *tb = NULL; float *zb = NULL;
	if(posix_memalign((void **)&sb, 64, n*4+64)) return -3;
	if(posix_memalign((void **)&tb, 64, n*4+64)) { free(sb); return -3; }
	if(posix_memalign((void **)&zb, 64, n*4+64)) { free(sb); free(tb); return -3; }
	memset(sb, 0, n*4); memset(tb, 0, n*4); memset(zb, 0, n*4);

	rwidth = w; rheight = h; rscale = 1;
	sbuf = sb; tsbuf = tb; zbuf = zb;
	sec_current = sec;

	mat4 cam;
	memcpy(&cam, cam16, sizeof(cam));

	if(nthreads > 0) omp_set_num_threads(nthreads);
	int nt = omp_get_max_threads();
	unsigned *old = calloc(nt, sizeof(unsigned));
	unsigned oldmain = pwnref_set_fast_math(1, 0);
#pragma omp parallel num_threads(nt)
	{ old[omp_get_thread_num()] = pwnref_set_fast_math(1, 0); }

	level_prepare_render(lvroot);
	trace_screen_centred(lvroot, 0, 0, w, h, &cam);

#pragma omp parallel num_threads(nt)
	{ pwnref_set_fast_math(0, old[omp_get_thread_num()]); }
	pwnref_set_fast_math(0, oldmain);
	free(old);

	memcpy(sbuf_out, sb, n*4);
	if(zbuf_out != NULL) memcpy(zbuf_out, zb, n*4);
	free(sb); free(tb); free(zb);
	sbuf = tsbuf = NULL; zbuf = NULL;
	return 0;
}

/* screen_upscale (screen.h:126-149) into a caller surface */
int pwnref_upscale(const uint32_t *src, int w, int h, int scale, int pitch_bytes, uint32_t *dst)
{
	pwnref_surface surf = { pitch_bytes, dst };
	rwidth = w; rheight = h; rscale = scale;
	sbuf = (uint32_t *)src;
	screen = &surf;
	screen_upscale();
	screen = NULL; sbuf = NULL;
	return 0;
}

/* known-answer probes of the helpers (util.h) */
uint32_t pwnref_col_ftoint(const float *v4) { return col_ftoint(_mm_loadu_ps(v4)); }
void pwnref_normalise(const float *in4, float *out4) { _mm_storeu_ps(out4, v_normalise(_mm_loadu_ps(in4))); }
float pwnref_dot(const float *a4, const float *b4) { return v_dot(_mm_loadu_ps(a4), _mm_loadu_ps(b4)); }
float pwnref_rcp(float x) { return _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(x))); }
float pwnref_rsqrt(float x) { return _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(x))); }
float pwnref_randfs(uint32_t *seed) { return randfs(seed); }
float pwnref_randfu(uint32_t *seed) { return randfu(seed); }
uint32_t pwnref_randi(uint32_t *seed) { return randi(seed); }
int pwnref_get_cell(int cx, int cz) { return lvroot == NULL ? -1 : get_cell(lvroot, cx, cz); }
float pwnref_sinf(float x) { return sinf(x); }
float pwnref_cosf(float x) { return cosf(x); }
float pwnref_expf(float x) { return expf(x); }
void pwnref_mat4_roty(float *m16, float ang) { mat4 m; memcpy(&m, m16, 64); mat4_roty(&m, ang); memcpy(m16, &m, 64); }
void pwnref_mat4_rotx(float *m16, float ang) { mat4 m; memcpy(&m, m16, 64); mat4_rotx(&m, ang); memcpy(m16, &m, 64); }
