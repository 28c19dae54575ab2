// CPU stand-ins for libpwnhip's kernels, for the sanitizer builds of its host logic (tools/sanitize/README.txt).  They do NOT
// compute the product's pixels.  What they keep is what the choreography depends on:
//   trace   writes a value per pixel that depends on (x, y, sec, camera, the tables' checksum): a stale table or a stale row shows;
//           depth is small except in a band of rows that FAKE_DEEP_ROWS moves through the frame, where it is large
//   blur    out(x, y) mixes the pixel with the one `reach(depth)` rows below it, like the reference's taps reach 0.002*h*|depth-1| rows
//           (screen.h:100-102): a missing halo row shows as a wrong pixel, a tap outside the rows it was given is counted in *miss
// so that a frame row-tiled over N members equals the frame of one context, bit for bit, exactly when strips, halo rows, the
// repeat after a missed halo, the gather and the delivery are right.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include "pwn_internal.h"

static uint32_t mix(uint32_t a, uint32_t b) { a ^= b + 0x9e3779b9u + (a << 6) + (a >> 2); return a * 2654435761u; }

extern "C" hipError_t pwn_launch_trace(const pwn_trace_params *P, int, size_t, bool count, hipStream_t)
{
	uint32_t tab = 0;
	for(uint32_t i = 0; i < P->blob_bytes / 4; i += 7) tab = mix(tab, P->blob[i]);
	uint32_t secbits, cambits = 0;
	memcpy(&secbits, &P->sec_current, 4);
	for(int i = 0; i < 4; i++) { uint32_t u; memcpy(&u, &P->rayb[i], 4); cambits = mix(cambits, u); memcpy(&u, &P->from[i], 4); cambits = mix(cambits, u); }
	const int deep0 = (int)(fabsf(P->sec_current) * 37.0f) % (P->h > 8 ? P->h - 8 : 1);
	for(int y = P->y0; y < P->y1; y++)
		for(int x = 0; x < P->w; x++)
		{
			const size_t o = (size_t)y * P->w + x;
			P->sbuf[o] = mix(mix(mix((uint32_t)x, (uint32_t)y), secbits ^ tab), cambits);
			// (like a ray that runs out of steps, trace.h:677: every 97th pixel keeps the depth it had)
			if((x + 3 * y) % 97 != 0) P->zbuf[o] = (P->sec_current >= 100.0f && y >= deep0 && y < deep0 + 8) ? 400.0f : 1.0f + (float)((x + y + (secbits >> 20)) % 9);
		}
	if(P->clear_word) *P->clear_word = 0u;
	if(P->cost_word) *P->cost_word += (uint32_t)(P->y1 - P->y0) * 10u + (uint32_t)(P->y0 % 7);
	if(P->tickets_next) for(unsigned q = 0; q < PWN_QUEUES; q++) P->tickets_next[q * PWN_QUEUE_STRIDE] = 0u;
	if(count && P->counters) { P->counters[0] += (unsigned long long)(P->y1 - P->y0) * P->w * 3ull; P->counters[1] += (unsigned long long)(P->y1 - P->y0) * P->w * 12ull; }
	return hipSuccess;
}
extern "C" hipError_t pwn_launch_trace_refill(const pwn_trace_params *P, int g, size_t l, bool c, hipStream_t s) { return pwn_launch_trace(P, g, l, c, s); }
extern "C" unsigned pwn_trace_refill_lds_extra(bool) { return 16u; }
extern "C" int pwn_trace_refill_blocks_per_cu(size_t, bool, bool) { return 4; }
extern "C" int pwn_trace_blocks_per_cu(size_t, bool, bool) { return 5; }
extern "C" int pwn_trace_tile_h(void) { return 4; }
extern "C" int pwn_trace_tile_w(void) { return 16; }
extern "C" unsigned pwn_trace_lds_extra(void) { return 16u; }

extern "C" hipError_t pwn_launch_blur(const pwn_blur_params *B, hipStream_t)
{
	if(B->cost_acc != NULL) { *B->cost_out = (uint32_t)((unsigned long long)*B->cost_acc * B->cost_mul / B->cost_div); *B->cost_acc = 0u; }
	unsigned missed = 0;
	for(int y = B->y0; y < B->y1; y++)
		for(int x = 0; x < B->w; x++)
		{
			const size_t o = (size_t)y * B->w + x;
			const float z = B->zbuf[o];
			int reach = (int)(0.002f * (float)B->h * fabsf(z - 1.0f));
			int ty = y + ((x & 1) ? reach : -reach);
			ty = ty < 0 ? 0 : (ty >= B->h ? B->h - 1 : ty);
			if(B->miss != NULL && (unsigned)(ty - B->avail_y0) >= (unsigned)(B->avail_y1 - B->avail_y0)) missed++;
			B->out[o] = mix(B->pre[o], B->pre[(size_t)ty * B->w + x]);
		}
	if(B->miss != NULL && missed) *B->miss += missed;
	return hipSuccess;
}
extern "C" hipError_t pwn_launch_order(const uint16_t *, uint32_t, uint32_t, uint32_t *, hipStream_t) { return hipSuccess; }
extern "C" hipError_t pwn_launch_upscale(const uint32_t *src, uint32_t *dst, int w, int h, int scale, int pitch, hipStream_t)
{
	const size_t rowadv = (size_t)w * scale + (size_t)pitch * (scale - 1);
	for(int dy = 0; dy < h * scale; dy++) for(int dx = 0; dx < w * scale; dx++)
		dst[(size_t)(dy / scale) * rowadv + (size_t)(dy % scale) * pitch + dx] = src[(size_t)(dy / scale) * w + dx / scale];
	return hipSuccess;
}
extern "C" hipError_t pwn_launch_upload(const void *src, void *dst, size_t bytes, hipStream_t) { memcpy(dst, src, bytes); return hipSuccess; }
extern "C" hipError_t pwn_launch_words(const uint32_t *all, const uint32_t *own, uint32_t *h, int world, hipStream_t)
{
	for(int i = 0; i < 2 * world + 2; i++) h[i] = i < 2 * world ? all[i] : own[i - 2 * world];
	return hipSuccess;
}
extern "C" hipError_t pwn_launch_probe(int, const uint32_t *in, uint32_t *out, int n, const uint16_t *, hipStream_t) { for(int i = 0; i < n; i++) out[i] = in[i]; return hipSuccess; }
