/*
 * oracle/pwn_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement ("port") of the reference's render hot path, written from
 * SURVEY.md section 8 and the reference's behaviour, in scalar C with
 *   - table emulation of RCPPS/RSQRTPS (approx_tables.h),
 *   - its own sinf/cosf/expf (pwn_libm.h),
 * so that it produces the reference's pixels on any x86-64 host.
 * Pinned against the compiled reference (oracle/_ref) by tests/test_oracle_vs_ref.py
 * and against the committed golden frames/hashes in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library; the product (pwnfps_amd/) never links or loads it.
 */
#ifndef PWN_ORACLE_H
#define PWN_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one endpoint pair, reference `portal` (defs.h:87-94) without double1/2 */
typedef struct pwno_portal { int32_t x1, z1, x2, z2, rot12, c1, c2; } pwno_portal;
/* script.h:20-32 arguments: r, refl, pos, colour b,g,r */
typedef struct pwno_sphere { float r, refl, x, y, z, cb, cg, cr; } pwno_sphere;

typedef struct pwno_level
{
	uint8_t data[64][64];       /* [z][x], defs.h:105 */
	pwno_portal pmap[26];       /* defs.h:103 */
	int32_t sx, sz;             /* spawn cell, defs.h:101 */
	int32_t nspheres;
	pwno_sphere *spheres;
	/* per-cell sphere lists in object order (level.h:1-39,64-81) as CSR */
	int32_t bin_off[4097];
	int32_t *bin_idx;
	int32_t bin_cap;
} pwno_level;

typedef struct pwno_stats
{
	int64_t rays, steps, portals, sphere_tests, exhausted;
} pwno_stats;

pwno_level *pwno_level_new(void);
void pwno_level_free(pwno_level *lv);
/* level.h:107-228 */
int pwno_level_load_file(pwno_level *lv, const char *path);
int pwno_level_load_mem(pwno_level *lv, const char *text, int len);
int pwno_level_set_tables(pwno_level *lv, const uint8_t *data4096, const int32_t *pmap26x7);
/* level.h:64-81 (level_prepare_render) */
int pwno_level_set_spheres(pwno_level *lv, const pwno_sphere *s, int n);

/* screen.h:31-67 (trace only): rows [y0,y1) of a w x h frame. sbuf/zbuf are
   FULL-frame pointers (pitch w).  zbuf entries of rays that exhaust maxsteps
   are left untouched (trace.h:677). */
int pwno_trace_rows(const pwno_level *lv, int w, int h, int y0, int y1,
	const float cam16[16], float sec_current, int nthreads,
	uint32_t *sbuf, float *zbuf, pwno_stats *stats);
/* screen.h:69-123: blur rows [y0,y1): reads tsbuf (full pre-blur frame) and
   zbuf, writes sbuf rows */
int pwno_blur_rows(int w, int h, int y0, int y1, int nthreads,
	const uint32_t *tsbuf, const float *zbuf, uint32_t *sbuf);
/* whole trace_screen_centred(lv,0,0,w,h,cam): trace + blur_passes blurs */
int pwno_render(const pwno_level *lv, int w, int h, const float cam16[16], float sec_current,
	int blur_passes, int nthreads, uint32_t *sbuf, float *zbuf, pwno_stats *stats);
/* screen.h:126-149 */
int pwno_upscale(const uint32_t *src, int w, int h, int scale, int pitch_bytes, uint32_t *dst);

/* helpers exposed for known-answer tests */
uint32_t pwno_col_ftoint(const float v[4]);
void pwno_normalise(const float in[4], float out[4]);
float pwno_dot(const float a[4], const float b[4]);
float pwno_rcp(float x);
float pwno_rsqrt(float x);
float pwno_sinf(float x);
float pwno_cosf(float x);
float pwno_expf(float x);
uint32_t pwno_randi(uint32_t *seed);
float pwno_randfu(uint32_t *seed);
float pwno_randfs(uint32_t *seed);
uint32_t pwno_pixel_seed(int x, int y, int rwidth);
/* blur row seed after `groups` 4-pixel groups (32 draws each): LCG skip-ahead check */
uint32_t pwno_blur_seed_at(int cy, int groups);
void pwno_frame_setup(int w, int h, const float cam16[16], float rayb[4], float rdx[4], float rdy[4]);
uint64_t pwno_fnv64(const uint32_t *p, int64_t n);
void pwno_mat4_roty(float m16[16], float ang);
void pwno_mat4_rotx(float m16[16], float ang);

#ifdef __cplusplus
}
#endif
#endif
