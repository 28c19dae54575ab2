// pwn_api.cpp -- the C ABI of libpwnhip.so (include/pwnhip.h) over the HIP
// runtime: context, table packing/upload, kernel launches, D2H.
// Host code only; compiled with -ffp-contract=off because the camera set-up
// of screen.h:43-57 is part of the bit-exact path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <new>
#include <vector>

#include "pwnhip.h"
#include "pwn_internal.h"
#include "level_host.h"
#include "approx_tables.inc"

extern "C" const char *pwn_strerror(int code)
{
	switch(code)
	{
		case PWN_OK: return "ok";
		case PWN_EINVAL: return "invalid argument";
		case PWN_ENODEV: return "no usable HIP device";
		case PWN_ENOMEM: return "out of memory";
		case PWN_EIO: return "level file could not be read";
		case PWN_EHIP: return "HIP runtime error";
		case PWN_ENOLEVEL: return "no level uploaded";
		case PWN_ETOOBIG: return "sphere tables exceed the LDS budget";
		case PWN_EBUSY: return "frame slot still in flight";
		case PWN_ENOTSUP: return "not available (RCCL missing, or not configured)";
		case PWN_ETIMEDOUT: return "the row tiling's deadline passed waiting for a peer";
	}
	return "unknown error";
}

extern "C" const char *pwn_last_error(pwn_ctx *ctx) { return ctx ? ctx->err : "null context"; }

static int ensure_scratch(pwn_ctx *c, size_t bytes)
{
	if(bytes <= c->scratch_cap) return PWN_OK;
	if(c->d_scratch) { (void)hipFree(c->d_scratch); c->d_scratch = NULL; c->scratch_cap = 0; }
	HIPCHK(c, hipMalloc((void **)&c->d_scratch, bytes));
	c->scratch_cap = bytes;
	return PWN_OK;
}

static void expand_tables(uint16_t *rcp, uint16_t *rsq);
static void frames_release(pwn_ctx *c);

// pwn_init_multi's handle (pwn_group.cpp): the call goes to the group, or to its member 0 where it is about the one
// level and object table, or is refused
#define GRP_HEAD(c) ((c) != NULL && (c)->grp != NULL && (c)->grp_head)
#define GRP_M0(c) pwn_group_member((c), 0)
#define GRP_REFUSE(c, what) do { if(GRP_HEAD(c)) { snprintf((c)->err, sizeof((c)->err), "%s is not available on a group's handle", what); return PWN_ENOTSUP; } } while(0)

extern "C" int pwn_init(pwn_ctx **out, int device, int width, int height)
{
	if(out == NULL || width <= 0 || height <= 0 || width > 32768 || height > 32768) return PWN_EINVAL;
	*out = NULL;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PWN_ENODEV;
	pwn_ctx *c = new(std::nothrow) pwn_ctx();
	if(c == NULL) return PWN_ENOMEM;
	c->device = device; c->w = width; c->h = height;
	c->blur_passes = 1; c->counters_on = 0; c->scheduler = PWN_SCHED_DEFAULT; c->refill_limit = PWN_REFILL_LIMIT_DEFAULT; c->have_level = false; c->cell_base_ok = false;
	// experiments and the test suite pick the trace scheduler for every context of a process
	if(const char *e = getenv("PWN_SCHEDULER")) c->scheduler = (strcmp(e, "refill") == 0 || strcmp(e, "1") == 0) ? PWN_SCHED_REFILL : PWN_SCHED_UNITS;
	if(const char *e = getenv("PWN_REFILL_LIMIT")) { int v = atoi(e); if(v >= 1 && v <= 64000) c->refill_limit = v; }
	// test / experiment hooks, read once: every camera through the general 4-lane variant (tests/test_gpu_fuzz.py);
	// workgroups per CU of the persistent grid
	c->dbg_force_hasw = getenv("PWN_DBG_FORCE_HASW") != NULL;
	c->dbg_blocks_per_cu = 0;
	c->dbg_blur_th = 0;
	if(const char *e = getenv("PWN_DBG_BLUR_TH")) c->dbg_blur_th = atoi(e);
	c->dbg_sphere_lists = 0; c->off_recsph = 0;
	if(const char *e = getenv("PWN_SPHERE_LISTS")) c->dbg_sphere_lists = strcmp(e, "indexed") == 0 ? 1 : (strcmp(e, "inline") == 0 ? 2 : 0);
	c->dbg_blur_tw = 0; c->dbg_blur_batch = -1;          // (sweeps with a -DPWN_BLUR_SWEEP build: tools/r3/run_w.sh)
	if(const char *e = getenv("PWN_DBG_BLUR_TW")) c->dbg_blur_tw = atoi(e);
	if(const char *e = getenv("PWN_DBG_BLUR_BATCH")) if(*e) c->dbg_blur_batch = atoi(e);
	if(const char *e = getenv("PWN_DBG_BLOCKS_PER_CU")) { int v = atoi(e); if(v > 0) c->dbg_blocks_per_cu = v; }
	c->blob_cur = 0; c->blob_dirty = true; c->off_sph = 0; c->stage_next = 0; c->up_stream = NULL;
	for(int i = 0; i < PWN_NBLOB; i++)
	{
		c->d_blob[i] = NULL; c->blob_has_static[i] = false;
		c->ev_tables[i] = NULL; c->tables_in_use[i] = false; c->tables_wait[i] = NULL;
		c->ev_upload[i] = NULL; c->upload_pending[i] = false;
	}
	for(int i = 0; i < PWN_NSTAGE; i++) { c->h_stage[i] = NULL; c->ev_stage[i] = NULL; c->stage_used[i] = false; }
	c->d_pre = c->d_out = NULL; c->d_z = NULL; c->d_skip = NULL; c->d_counters = NULL; c->d_tickets = NULL; c->ticket_set = 0; c->launch_rot = 2; c->launch_waits = 0;
	c->trace_clear_word = NULL; c->trace_cost_word = NULL; c->trace_tables_event = NULL; c->grid_reserve = 0;
	memset(&c->room, 0, sizeof(c->room)); c->room.mode = -1; c->launch_room = 0;
	c->cost_mul = c->cost_div = 1u; c->blur_cost_mul = c->blur_cost_div = 0u;
	if(const char *e = getenv("PWN_TRACE_ROOM")) if(*e) c->room.mode = atoi(e) < 0 ? -1 : atoi(e);      // (the option's default for every context of a process)
	c->d_scratch = NULL; c->scratch_cap = 0;
	for(int i = 0; i < 8; i++) { c->occ_lds[i] = 0; c->occ_blocks[i] = 0; }
	c->stream = NULL; c->copy_stream = NULL; c->copy_stream2 = NULL; c->stream2 = NULL; c->d_pre2 = NULL;
	c->frame_overlap = 1; c->last_frame_done = NULL; c->last_frame_stream = NULL;
	if(const char *e = getenv("PWN_FRAME_OVERLAP")) c->frame_overlap = atoi(e) != 0;
	c->unit_order = 0;             // (measured: a loss except on short launches that run alone, profiles/r4/unit_order_ab.txt)
	if(const char *e = getenv("PWN_UNIT_ORDER")) c->unit_order = atoi(e) != 0;
	c->tiled_comms = PWN_TILED_COMMS_ONE;
	if(const char *e = getenv("PWN_TILED_COMMS")) c->tiled_comms = (strcmp(e, "perstream") == 0 || strcmp(e, "2") == 0) ? PWN_TILED_COMMS_PER_STREAM : PWN_TILED_COMMS_ONE;
	c->tiled_streams = PWN_TILED_STREAMS_DEFAULT;
	if(const char *e = getenv("PWN_TILED_STREAMS")) { const int v = atoi(e); if(v == 2 || v == 3) c->tiled_streams = v; }
	c->tiled_choreo = PWN_TILED_CHOREO_INSTREAM;
	if(const char *e = getenv("PWN_TILED_CHOREO")) c->tiled_choreo = (strcmp(e, "split") == 0 || strcmp(e, "1") == 0) ? PWN_TILED_CHOREO_SPLIT : PWN_TILED_CHOREO_INSTREAM;
	memset(c->ev, 0, sizeof(c->ev));
	memset(&c->stats, 0, sizeof(c->stats));
	c->strip_reach = 0;
	c->call_strips = -1; c->strip_ev_n = 0; c->d_strip_miss = NULL; c->h_strip_miss = NULL; c->strip_backoff = 0;
	c->strip_calls = c->strip_redone = 0; c->strips_last = 1; c->host_regs_n = 0;
	memset(c->strip_ev, 0, sizeof(c->strip_ev));
	c->strip_copy_streams = 0; c->strip_calib_n = 0;
	if(const char *e = getenv("PWN_CALL_COPY_STREAMS")) if(atoi(e) == 1 || atoi(e) == 2) c->strip_copy_streams = atoi(e);
	if(const char *e = getenv("PWN_CALL_STRIPS")) if(*e) { const int v = atoi(e); if(v >= -1 && v <= PWN_CALL_STRIPS_MAX && v != 1) c->call_strips = v; }
	c->frame_timing = 1; c->wave_log_on = 0; c->d_wave_log = NULL; c->wave_log_cap = 0;
	c->nslots = 0; c->frame_flags = 0; c->frame_scale = 1; c->frame_pitch = 0; c->frame_seq = 0;
	memset(c->slot, 0, sizeof(c->slot));
	c->tiled = NULL; c->grp = NULL; c->grp_head = false; c->hub = NULL;
	c->err[0] = 0;
	pwn_level_clear(c->cells, c->pmap, c->spawn);
	c->bin_off.assign(4097, 0);
	expand_tables(c->tabs, c->tabs + 2048);

	int rc = PWN_OK;
	do
	{
		hipDeviceProp_t prop;
		if(hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { rc = PWN_ENODEV; break; }
		if(strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		{
			snprintf(c->err, sizeof(c->err), "device %d is %s; libpwnhip is built for gfx950 only", device, prop.gcnArchName);
			rc = PWN_ENODEV; break;
		}
		c->num_cus = prop.multiProcessorCount;
		size_t n = (size_t)width * (size_t)height;
		if(hipMalloc((void **)&c->d_pre, n * 4) != hipSuccess ||
		   hipMalloc((void **)&c->d_out, n * 4) != hipSuccess ||
		   hipMalloc((void **)&c->d_z, n * 4) != hipSuccess ||
		   hipMalloc((void **)&c->d_counters, PWN_NCOUNTERS * sizeof(unsigned long long)) != hipSuccess ||
		   hipMalloc((void **)&c->d_tickets, PWN_TICKET_SETS * PWN_QUEUES * PWN_QUEUE_STRIDE * sizeof(uint32_t)) != hipSuccess ||
		   hipMemset(c->d_tickets, 0, PWN_TICKET_SETS * PWN_QUEUES * PWN_QUEUE_STRIDE * sizeof(uint32_t)) != hipSuccess ||
		   hipMalloc((void **)&c->d_skip, sizeof(uint2) * (size_t)(width / 4 + 1)) != hipSuccess) { rc = PWN_ENOMEM; break; }
		for(int i = 0; i < PWN_NBLOB && rc == PWN_OK; i++)
			if(hipMalloc((void **)&c->d_blob[i], PWN_BLOB_MAX) != hipSuccess) rc = PWN_ENOMEM;
		for(int i = 0; i < PWN_NSTAGE && rc == PWN_OK; i++)
			if(hipHostMalloc((void **)&c->h_stage[i], PWN_BLOB_MAX, hipHostMallocDefault) != hipSuccess) rc = PWN_ENOMEM;
		if(rc != PWN_OK) break;
		if(hipMemset(c->d_z, 0, n * 4) != hipSuccess || hipMemset(c->d_pre, 0, n * 4) != hipSuccess ||
		   hipMemset(c->d_out, 0, n * 4) != hipSuccess) { rc = PWN_EHIP; break; }
		if(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
		   hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
		   hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) != hipSuccess) { rc = PWN_EHIP; break; }
		// The second compute stream has to sit on another HARDWARE queue than the first, or the two only take turns.
		// The runtime hands its hardware queues (four per priority level by default, GPU_MAX_HW_QUEUES) to streams as
		// they are first used, and with PyTorch's streams, the copy and the upload stream in the same process the two
		// compute streams of a context landed on the same one: frames "on two streams" measured exactly like frames
		// on one (0.0823 / 0.0823 ms at 3840 x 272; 0.0823 / 0.0655 with GPU_MAX_HW_QUEUES=8).  Queues are pooled per
		// priority, so a stream of another priority level is certain to have a queue of its own.
		{
			int least = 0, greatest = 0;
			(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
			int prio = greatest < 0 ? greatest : -1;          // (numerically lower = more urgent; 0 is the default level)
			if(const char *e = getenv("PWN_DBG_STREAM2_PRIORITY")) if(*e) prio = atoi(e);
			if(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio) != hipSuccess)
			{
				// (a runtime without stream priorities: an ordinary stream -- the frames then overlap only if it happens
				// to get a hardware queue of its own)
				(void)hipGetLastError();
				if(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { rc = PWN_EHIP; break; }
			}
		}
		for(int i = 0; i < 4; i++) if(hipEventCreate(&c->ev[i]) != hipSuccess) { rc = PWN_EHIP; break; }
		for(int i = 0; i < PWN_NBLOB && rc == PWN_OK; i++)
			if(hipEventCreateWithFlags(&c->ev_tables[i], hipEventDisableTiming) != hipSuccess ||
			   hipEventCreateWithFlags(&c->ev_upload[i], hipEventDisableTiming) != hipSuccess) rc = PWN_EHIP;
		for(int i = 0; i < PWN_NSTAGE && rc == PWN_OK; i++)
			if(hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming) != hipSuccess) rc = PWN_EHIP;
		if(rc != PWN_OK) break;

		// blur LCG skip-ahead: entry g maps the row seed to the seed in front
		// of group g, i.e. 32*g draws later (screen.h:82,95-109; util.h:1-6)
		std::vector<uint2> skip((size_t)(width / 4 + 1));
		uint32_t A = 1, C = 0;
		for(size_t g = 0; g < skip.size(); g++)
		{
			skip[g].x = A; skip[g].y = C;
			for(int k = 0; k < 32; k++) { A = (A * 25739u) & 0x7FFFFFFFu; C = (C * 25739u + 4u) & 0x7FFFFFFFu; }
		}
		if(hipMemcpy(c->d_skip, skip.data(), skip.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) { rc = PWN_EHIP; break; }
		// (the memsets above ran on the null stream; the context's streams are non-blocking and do not wait for it)
		if(hipDeviceSynchronize() != hipSuccess) { rc = PWN_EHIP; break; }
	} while(0);

	if(rc != PWN_OK) { pwn_destroy(c); return rc; }
	*out = c;
	return PWN_OK;
}

extern "C" void pwn_destroy(pwn_ctx *c)
{
	if(GRP_HEAD(c)) { pwn_group_destroy(c); return; }
	if(c == NULL) return;
	(void)hipSetDevice(c->device);
	if(c->tiled) pwn_tiled_destroy(c);
	if(c->stream) (void)hipStreamSynchronize(c->stream);
	if(c->stream2) (void)hipStreamSynchronize(c->stream2);
	if(c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
	if(c->up_stream) (void)hipStreamSynchronize(c->up_stream);
	(void)hipDeviceSynchronize();        // strip forms run on the caller's streams
	frames_release(c);
	for(int i = 0; i < 4; i++) if(c->ev[i]) (void)hipEventDestroy(c->ev[i]);
	for(int i = 0; i < c->strip_ev_n; i++) if(c->strip_ev[i]) (void)hipEventDestroy(c->strip_ev[i]);
	(void)hipFree(c->d_strip_miss);
	if(c->h_strip_miss) (void)hipHostFree(c->h_strip_miss);
	for(int i = 0; i < c->host_regs_n; i++) (void)hipHostUnregister(c->host_regs[i].base);
	for(int i = 0; i < PWN_NBLOB; i++)
	{
		if(c->ev_tables[i]) (void)hipEventDestroy(c->ev_tables[i]);
		if(c->ev_upload[i]) (void)hipEventDestroy(c->ev_upload[i]);
		(void)hipFree(c->d_blob[i]);
	}
	for(int i = 0; i < PWN_NSTAGE; i++)
	{
		if(c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]);
		if(c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
	}
	if(c->stream) (void)hipStreamDestroy(c->stream);
	if(c->stream2) (void)hipStreamDestroy(c->stream2);
	if(c->copy_stream2) { (void)hipStreamSynchronize(c->copy_stream2); (void)hipStreamDestroy(c->copy_stream2); }
	if(c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
	if(c->up_stream) (void)hipStreamDestroy(c->up_stream);
	(void)hipFree(c->d_pre); (void)hipFree(c->d_out); (void)hipFree(c->d_z); (void)hipFree(c->d_pre2);
	(void)hipFree(c->d_wave_log);
	for(int i = 0; i < 4; i++) { (void)hipFree(c->order[i].d_cost); (void)hipFree(c->order[i].d_perm); }
	(void)hipFree(c->d_skip); (void)hipFree(c->d_counters); (void)hipFree(c->d_tickets); (void)hipFree(c->d_scratch);
	delete c;
}

// ---- PWN_OPT_TRACE_ROOM (pwn_internal.h) ----
#define ROOM_WINDOW 24        // delivered frames timed per setting
#define ROOM_SKIP 6           // ... after this many that are not (frames in flight were launched with the old setting)
#define ROOM_HOLD 480         // frames the better setting is kept before the next look
int pwn_room_for_launch(pwn_ctx *c)
{
	if(c->room.mode >= 0) return c->room.mode;
	return c->room.arm ? c->num_cus : 0;
}

void pwn_room_frame_done(pwn_ctx *c)
{
	pwn_room_ctl &r = c->room;
	if(r.mode >= 0) return;
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	const double now = (double)ts.tv_sec + (double)ts.tv_nsec * 1e-9, dt = now - r.t_prev;
	const bool first = r.t_prev == 0.0;
	r.t_prev = now;
	if(first) { r.skip = ROOM_SKIP; return; }
	if(r.hold > 0) { if(r.hold > 1) r.hold--; else if(now - r.t_hold > 0.5) { r.hold = 0; r.arm = 1 - r.best; r.skip = ROOM_SKIP; r.sum[0] = r.sum[1] = 0.0; r.cnt[0] = r.cnt[1] = 0; r.switches++; } return; }
	if(r.skip > 0) { r.skip--; return; }
	// (a host that stops between frames -- a debugger, a vsync -- is not the GPU's time: intervals far above the window's mean are left out)
	if(r.cnt[r.arm] >= 4 && dt > 4.0 * r.sum[r.arm] / r.cnt[r.arm]) return;
	r.sum[r.arm] += dt; r.cnt[r.arm]++;
	if(r.cnt[r.arm] < ROOM_WINDOW) return;
	if(r.cnt[1 - r.arm] < ROOM_WINDOW) { r.arm = 1 - r.arm; r.skip = ROOM_SKIP; r.switches++; return; }
	// both windows are in: the better one (the one in use stays on a tie within 0.5 %)
	const double m0 = r.sum[0] / r.cnt[0], m1 = r.sum[1] / r.cnt[1];
	int best = m1 < m0 ? 1 : 0;
	if(r.looks > 0 && best != r.best && (best ? m1 > 0.995 * m0 : m0 > 0.995 * m1)) best = r.best;
	if(best != r.arm) { r.skip = ROOM_SKIP; r.switches++; }
	r.best = best; r.arm = best; r.hold = ROOM_HOLD; r.t_hold = now; r.looks++;       // (kept for ROOM_HOLD frames and at least half a second)
}

extern "C" int pwn_trace_room_state(pwn_ctx *c, int out[4])
{
	if(GRP_HEAD(c)) return pwn_trace_room_state(GRP_M0(c), out);
	if(c == NULL || out == NULL) return PWN_EINVAL;
	out[0] = c->room.mode; out[1] = pwn_room_for_launch(c); out[2] = (int)c->room.looks; out[3] = (int)c->room.switches;
	return PWN_OK;
}

extern "C" int pwn_set_option(pwn_ctx *c, int option, int value)
{
	if(GRP_HEAD(c)) return pwn_group_set_option(c, option, value);
	if(c == NULL) return PWN_EINVAL;
	switch(option)
	{
		case PWN_OPT_BLUR_PASSES:
			if(value < 0 || value > 16) return PWN_EINVAL;
			// the row tiling fixed its halo, planes and choreography on this at pwn_tiled_init; frames in flight were
			// enqueued with it
			if(c->tiled != NULL) return PWN_EBUSY;
			for(int i = 0; i < c->nslots; i++) if(c->slot[i].in_flight) return PWN_EBUSY;
			c->blur_passes = value; return PWN_OK;
		case PWN_OPT_COUNTERS: c->counters_on = value ? 1 : 0; return PWN_OK;
		case PWN_OPT_SCHEDULER:
			if(value < 0 || value > PWN_SCHED_REFILL) return PWN_EINVAL;
			if(value != c->scheduler) c->blob_dirty = true;          // (the form of the per-cell sphere lists goes with the scheduler: pack_blob)
			c->scheduler = value; return PWN_OK;
		case PWN_OPT_WAVE_LOG: c->wave_log_on = value ? 1 : 0; return PWN_OK;
		case PWN_OPT_FRAME_TIMING: if(value < 0) return PWN_EINVAL; c->frame_timing = value; return PWN_OK;
		case PWN_OPT_REFILL_LIMIT: if(value < 1 || value > 64000) return PWN_EINVAL; c->refill_limit = value; return PWN_OK;
		case PWN_OPT_UNIT_ORDER: if(value != 0 && value != 1) return PWN_EINVAL; c->unit_order = value; for(int i = 0; i < 4; i++) c->order[i].perm_valid = false; return PWN_OK;
		case PWN_OPT_TILED_CHOREO:
			if(value != PWN_TILED_CHOREO_INSTREAM && value != PWN_TILED_CHOREO_SPLIT) return PWN_EINVAL;
			if(c->tiled != NULL) return PWN_EBUSY;
			c->tiled_choreo = value; return PWN_OK;
		case PWN_OPT_TILED_COMMS:
			if(value != PWN_TILED_COMMS_ONE && value != PWN_TILED_COMMS_PER_STREAM) return PWN_EINVAL;
			if(c->tiled != NULL) return PWN_EBUSY;
			c->tiled_comms = value; return PWN_OK;
		case PWN_OPT_TILED_STREAMS:
			if(value != 2 && value != 3) return PWN_EINVAL;
			if(c->tiled != NULL) return PWN_EBUSY;
			c->tiled_streams = value; return PWN_OK;
		case PWN_OPT_TRACE_ROOM: if(value < -1 || value > 4096) return PWN_EINVAL; memset(&c->room, 0, sizeof(c->room)); c->room.mode = value; return PWN_OK;
		case PWN_OPT_CALL_STRIPS:
			if(value < -1 || value == 1 || value > PWN_CALL_STRIPS_MAX) return PWN_EINVAL;
			c->call_strips = value; c->strip_backoff = 0; c->strip_reach = 0; return PWN_OK;
		case PWN_OPT_FRAME_OVERLAP:
			for(int i = 0; i < c->nslots; i++) if(c->slot[i].in_flight) return PWN_EBUSY;
			c->frame_overlap = value ? 1 : 0; return PWN_OK;
	}
	return PWN_EINVAL;
}

// ---- level / spheres --------------------------------------------------------

// RCPPS / RSQRTPS tables in the form the kernels read (dev_math.h, tables.h):
// entry = (1 - exponent offset) << 12 | result mantissa bits 22..11.
// approx_tables.inc entry: bit 12 = exponent offset, bits 11..0 = that mantissa.
static void expand_tables(uint16_t *rcp, uint16_t *rsq)
{
	for(int i = 0; i < 2048; i++)
	{
		rcp[i] = (uint16_t)(pwn_host_rcp_tab[i] ^ 0x1000u);
		// kernel index = input bits 23..13: bit 10 = low exponent bit; the generated
		// table is indexed by parity of (e - 127), i.e. with that bit flipped
		rsq[i] = (uint16_t)(pwn_host_rsqrt_tab[i ^ 0x400] ^ 0x1000u);
	}
}

static int pack_blob(pwn_ctx *c)
{
	// per-cell sphere lists, each closed by PWN_LIST_END; only non-empty cells get one
	uint32_t nsph = (uint32_t)c->spheres.size();
	uint32_t nbin = 0;
	for(int i = 0; i < 4096; i++)
	{
		uint32_t cnt = (uint32_t)(c->bin_off[i + 1] - c->bin_off[i]);
		if(cnt) nbin += cnt + 1u;
	}
	if(nbin > 32767u || nsph * 32u >= PWN_LIST_END) return PWN_ETOOBIG;      // list entries are byte offsets (index * 32) in 16 bits
	uint32_t total = (pwn_t_total(nbin, nsph) + 15u) & ~15u;
	if(total > PWN_BLOB_MAX) return PWN_ETOOBIG;
	// Which form the per-cell lists take (tables.h): inline sphere records -- one LDS read per test instead of two dependent ones --
	// for the unit scheduler's kernels wherever the bigger blob leaves as many workgroups per CU as the indexed one (2 KiB granules of
	// 152 KiB: pwn_i_launch_trace); level.txt 26.5 against 26.3 KB, synth64 30.0 against 28.3 (both five), synth256 32.8 against
	// 30.5 (four against five: indexed).  PWN_SPHERE_LISTS=indexed|inline in the environment forces one (experiments, tests).
	uint32_t nrec = (uint32_t)c->bin_off[4096];
	bool inl = false;
	{
		const uint32_t total_inl = (pwn_t_total_inl(nrec, nsph) + 15u) & ~15u;
		const uint32_t extra = pwn_trace_lds_extra();
		const uint32_t fit_idx = 155648u / ((total + extra + 2047u) & ~2047u), fit_inl = 155648u / ((total_inl + extra + 2047u) & ~2047u);
		inl = c->scheduler == PWN_SCHED_UNITS && nrec > 0u && total_inl <= PWN_BLOB_MAX && fit_inl >= (fit_idx < 5u ? fit_idx : 5u);
		if(c->dbg_sphere_lists == 1) inl = false;
		if(c->dbg_sphere_lists == 2) inl = c->scheduler == PWN_SCHED_UNITS && nrec > 0u && total_inl <= PWN_BLOB_MAX;
		if(inl) total = total_inl;
	}
	c->blob.assign(total, 0);        // (nothing has changed up to here: a failed call leaves the context as it was)
	uint8_t *b = c->blob.data();
	uint32_t *ci = (uint32_t *)(b + PWN_T_CELLINFO);
	uint16_t *bi = (uint16_t *)(b + PWN_T_BINIDX);
	// the level's part of the cell words (char + class bits) is the same until the level changes: built once,
	// copied per upload and patched where a cell has spheres
	if(!c->cell_base_ok)
	{
		uint32_t *cb = c->cell_base;
		memset(cb, 0, sizeof(c->cell_base));
		for(int z = 0; z < 64; z++)
		for(int x = 0; x < 64; x++)
			cb[z * PWN_GRID_PITCH + x] = (uint32_t)c->cells[z * 64 + x] | pwn_cell_class(c->cells[z * 64 + x]);
		// row / column 64 = what get_cell returns outside the grid on that axis (util.h:151-158),
		// never with spheres (the sphere loop runs for in-grid cells only, trace.h:252)
		for(int z = 0; z < 64; z++) cb[z * PWN_GRID_PITCH + 64] = cb[z * PWN_GRID_PITCH];
		for(int x = 0; x <= 64; x++) cb[64 * PWN_GRID_PITCH + x] = cb[x];
		c->cell_base_ok = true;
	}
	memcpy(ci, c->cell_base, sizeof(c->cell_base));
	uint32_t at = 0;
	for(int i = 0; i < 4096 && !inl; i++)
	{
		int32_t k0 = c->bin_off[i], k1 = c->bin_off[i + 1];
		if(k1 > k0)
		{
			ci[(i >> 6) * PWN_GRID_PITCH + (i & 63)] |= PWN_C_SPH | (at << 16);
			for(int32_t k = k0; k < k1; k++) bi[at++] = (uint16_t)(c->bin_idx[k] * 32);
			bi[at++] = (uint16_t)PWN_LIST_END;
		}
	}
	if(inl)
	{
		// records in the lists' order (bin_idx is that order already: cell by cell, object order inside a cell), the last of a cell signed
		float *rec = (float *)(b + PWN_T_BINIDX);
		uint16_t *which = (uint16_t *)(b + pwn_t_recsph_offset(nrec));
		for(int i = 0; i < 4096; i++)
		{
			const int32_t k0 = c->bin_off[i], k1 = c->bin_off[i + 1];
			if(k1 <= k0) continue;
			ci[(i >> 6) * PWN_GRID_PITCH + (i & 63)] |= PWN_C_SPH | ((uint32_t)k0 << 16);
			for(int32_t k = k0; k < k1; k++)
			{
				const pwn_sphere &q = c->spheres[(size_t)c->bin_idx[k]];
				float r2 = q.r * q.r;
				if(r2 < 1.17549435e-38f) r2 = 0.0f;          // (as below: the reference build's flush to zero)
				rec[4 * k + 0] = q.x; rec[4 * k + 1] = q.y; rec[4 * k + 2] = q.z;
				uint32_t wbits;
				memcpy(&wbits, &r2, 4);
				if(k == k1 - 1) wbits |= 0x80000000u;
				memcpy(&rec[4 * k + 3], &wbits, 4);
				which[k] = (uint16_t)(c->bin_idx[k] * 32);
			}
		}
	}
	memcpy(b + PWN_T_RCP, c->tabs, 4096);
	memcpy(b + PWN_T_RSQ, c->tabs + 2048, 4096);
	pwn_fill_faces((float *)(b + PWN_T_FACES));
	uint32_t *pm = (uint32_t *)(b + PWN_T_PMAP);
	for(int i = 0; i < 26; i++)
	{
		const pwn_portal &p = c->pmap[i];
		pm[2 * i] = (uint32_t)(p.x1 & 0xff) | ((uint32_t)(p.z1 & 0xff) << 8) | ((uint32_t)(p.x2 & 0xff) << 16) | ((uint32_t)(p.z2 & 0xff) << 24);
		pm[2 * i + 1] = (uint32_t)(p.rot12 & 0xff) | ((uint32_t)(p.c1 & 0xff) << 8) | ((uint32_t)(p.c2 & 0xff) << 16);
	}
	c->off_sph = inl ? pwn_t_sph_offset_inl(nrec) : pwn_t_sph_offset(nbin);
	c->off_recsph = inl ? pwn_t_recsph_offset(nrec) : 0u;
	// the kernels' layout of a sphere (tables.h): position and r*r in one 16-byte half, the rest in the other
	{
		float *sp = (float *)(b + c->off_sph);
		for(uint32_t i = 0; i < nsph; i++, sp += 8)
		{
			const pwn_sphere &q = c->spheres[i];
			float r2 = q.r * q.r;                 // trace.h:262 `rad*rad`, one fp32 multiply
			if(r2 < 1.17549435e-38f) r2 = 0.0f;   // ... whose result the device flushes like the reference build (FTZ)
			sp[0] = q.x; sp[1] = q.y; sp[2] = q.z; sp[3] = r2;
			sp[4] = q.refl; sp[5] = q.cb; sp[6] = q.cg; sp[7] = q.cr;
		}
	}

	// Upload into the copy no launch in flight reads.  The rcp / rsqrt head of the blob never
	// changes: once a copy has it, only [cellinfo, end) travels (cell words carry the sphere-list
	// offsets, so they change with the spheres).  Nothing here blocks on the GPU except re-use
	// of a staging buffer that is still being read, four uploads later.
	const int nb = (c->blob_cur + 1) % PWN_NBLOB;
	const uint32_t from = c->blob_has_static[nb] ? PWN_T_CELLINFO : 0u;
	const unsigned st = c->stage_next++ % PWN_NSTAGE;
	if(c->stage_used[st]) HIPCHK(c, hipEventSynchronize(c->ev_stage[st]));
	memcpy(c->h_stage[st], b + from, total - from);
	// launches still reading that copy (two uploads ago) finish first -- a wait between streams
	if(c->tables_in_use[nb]) HIPCHK(c, hipStreamWaitEvent(c->up_stream, c->tables_wait[nb], 0));
	// (a kernel that reads the pinned buffer, not a DMA copy: see pwn_upload_kernel; total and from are multiples of 16)
	HIPCHK(c, pwn_launch_upload(c->h_stage[st], c->d_blob[nb] + from, total - from, c->up_stream));
	HIPCHK(c, hipEventRecord(c->ev_stage[st], c->up_stream));
	HIPCHK(c, hipEventRecord(c->ev_upload[nb], c->up_stream));
	c->stage_used[st] = true;
	c->upload_pending[nb] = true;
	c->blob_has_static[nb] = true;
	c->blob_cur = nb;
	c->blob_dirty = false;
	return PWN_OK;
}

static int valid_portals(const pwn_portal *pm)
{
	for(int i = 0; i < 26; i++)
	{
		const int32_t v[4] = { pm[i].x1, pm[i].z1, pm[i].x2, pm[i].z2 };
		for(int k = 0; k < 4; k++) if(v[k] < -1 || v[k] > 63) return 0;
	}
	return 1;
}

extern "C" int pwn_upload_level(pwn_ctx *c, const uint8_t data[4096], const pwn_portal pmap[26])
{
	if(GRP_HEAD(c)) return (data == NULL || pmap == NULL) ? PWN_EINVAL : pwn_group_upload_level(c, data, pmap);
	if(c == NULL || data == NULL || pmap == NULL || !valid_portals(pmap)) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	memcpy(c->cells, data, 4096);
	memcpy(c->pmap, pmap, sizeof(c->pmap));
	c->have_level = true;
	c->cell_base_ok = false;
	c->blob_dirty = true;
	return pack_blob(c);
}

extern "C" int pwn_level_load_mem(pwn_ctx *c, const char *text, int len)
{
	if(GRP_HEAD(c)) return (text == NULL || len < 0) ? PWN_EINVAL : pwn_group_level_mem(c, text, len);
	if(c == NULL || text == NULL || len < 0) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	if(pwn_parse_level(text, len, c->cells, c->pmap, c->spawn) != 0) return PWN_EINVAL;
	c->have_level = true;
	c->cell_base_ok = false;
	c->blob_dirty = true;
	return pack_blob(c);
}

extern "C" int pwn_level_load(pwn_ctx *c, const char *path)
{
	if(c == NULL || path == NULL) return PWN_EINVAL;
	FILE *fp = fopen(path, "rb");
	if(fp == NULL) { snprintf(c->err, sizeof(c->err), "cannot open %s", path); return PWN_EIO; }
	std::vector<char> buf(1 << 20);
	size_t n = fread(buf.data(), 1, buf.size(), fp);
	fclose(fp);
	return pwn_level_load_mem(c, buf.data(), (int)n);
}

extern "C" int pwn_get_level(pwn_ctx *c, uint8_t data[4096], pwn_portal pmap[26], int32_t spawn[2])
{
	if(GRP_HEAD(c)) return pwn_get_level(GRP_M0(c), data, pmap, spawn);
	if(c == NULL) return PWN_EINVAL;
	if(data) memcpy(data, c->cells, 4096);
	if(pmap) memcpy(pmap, c->pmap, sizeof(c->pmap));
	if(spawn) { spawn[0] = c->spawn[0]; spawn[1] = c->spawn[1]; }
	return PWN_OK;
}

enum { OBJ_INVAL = 0, OBJ_FREE = 1, OBJ_SPHERE = 2 };   // defs.h:55-60

// level_prepare_render's binning + upload for a compact list of live spheres
static int upload_live(pwn_ctx *c, const pwn_sphere *s, int n);

extern "C" int pwn_upload_spheres(pwn_ctx *c, const pwn_sphere *s, int n)
{
	if(GRP_HEAD(c)) return (n < 0 || n > PWN_OBJ_MAX || (n > 0 && s == NULL)) ? PWN_EINVAL : pwn_group_upload_spheres(c, s, n);
	if(c == NULL || n < 0 || n > PWN_OBJ_MAX || (n > 0 && s == NULL)) return PWN_EINVAL;
	int rc = upload_live(c, s, n);
	if(rc == PWN_OK)
	{
		c->objs.assign(s, s + n);
		c->obj_typ.assign((size_t)n, OBJ_SPHERE);
	}
	return rc;
}

extern "C" int pwn_obj_new(pwn_ctx *c)
{
	if(GRP_HEAD(c)) { const int r = pwn_obj_new(GRP_M0(c)); if(r < 0) snprintf(c->err, sizeof(c->err), "%s", GRP_M0(c)->err); return r; }
	if(c == NULL) return PWN_EINVAL;
	for(size_t i = 0; i < c->obj_typ.size(); i++)
		if(c->obj_typ[i] == OBJ_FREE) { c->obj_typ[i] = OBJ_INVAL; return (int)i; }
	if(c->obj_typ.size() >= (size_t)PWN_OBJ_MAX)
	{
		snprintf(c->err, sizeof(c->err), "obj_new: all %d object slots are taken", PWN_OBJ_MAX);
		return PWN_ENOMEM;
	}
	c->objs.push_back(pwn_sphere());
	c->obj_typ.push_back(OBJ_INVAL);
	return (int)c->obj_typ.size() - 1;
}

// A handle is the index of a slot that level_obj_new once handed out.  Like the reference's
// part pointers it stays usable after obj_free: obj_set on a freed slot makes it a sphere again
// (script.h:24 sets pt->typ whatever it was), obj_free on it is a no-op (script.h:48).
static bool obj_ok(pwn_ctx *c, int obj, const char *who)
{
	if(c == NULL) return false;
	if(obj >= 0 && (size_t)obj < c->obj_typ.size()) return true;
	snprintf(c->err, sizeof(c->err), "%s: %d is not an object", who, obj);
	return false;
}

extern "C" int pwn_obj_set_sphere(pwn_ctx *c, int obj, double r, double refl, double x, double y, double z,
	double cb, double cg, double cr)
{
	if(GRP_HEAD(c)) { const int q = pwn_obj_set_sphere(GRP_M0(c), obj, r, refl, x, y, z, cb, cg, cr); if(q < 0) snprintf(c->err, sizeof(c->err), "%s", GRP_M0(c)->err); return q; }
	if(!obj_ok(c, obj, "obj_set")) return PWN_EINVAL;
	pwn_sphere &s = c->objs[(size_t)obj];
	s.r = (float)r; s.refl = (float)refl;
	s.x = (float)x; s.y = (float)y; s.z = (float)z;
	s.cb = (float)cb; s.cg = (float)cg; s.cr = (float)cr;
	c->obj_typ[(size_t)obj] = OBJ_SPHERE;
	return PWN_OK;
}

extern "C" int pwn_obj_free(pwn_ctx *c, int obj)
{
	if(GRP_HEAD(c)) { const int q = pwn_obj_free(GRP_M0(c), obj); if(q < 0) snprintf(c->err, sizeof(c->err), "%s", GRP_M0(c)->err); return q; }
	if(!obj_ok(c, obj, "obj_free")) return PWN_EINVAL;
	c->obj_typ[(size_t)obj] = OBJ_FREE;
	return PWN_OK;
}

extern "C" int pwn_level_get(pwn_ctx *c, int cx, int cz)
{
	if(GRP_HEAD(c)) return pwn_level_get(GRP_M0(c), cx, cz);
	if(c == NULL) return PWN_EINVAL;
	if(!c->have_level) return PWN_ENOLEVEL;
	if(cx < 0 || cx >= 64) cx = 0;
	if(cz < 0 || cz >= 64) cz = 0;
	return c->cells[cz * 64 + cx];
}

static int live_objects(pwn_ctx *c, std::vector<pwn_sphere> &live)
{
	for(size_t i = 0; i < c->obj_typ.size(); i++)
	{
		if(c->obj_typ[i] == OBJ_FREE) continue;
		if(c->obj_typ[i] != OBJ_SPHERE)
		{
			snprintf(c->err, sizeof(c->err), "object %d was created but never set", (int)i);
			return PWN_EINVAL;
		}
		live.push_back(c->objs[i]);
	}
	return PWN_OK;
}

extern "C" int pwn_prepare_render(pwn_ctx *c)
{
	if(GRP_HEAD(c)) return pwn_group_prepare_render(c);
	if(c == NULL) return PWN_EINVAL;
	std::vector<pwn_sphere> live;
	int rc = live_objects(c, live);
	if(rc != PWN_OK) return rc;
	return upload_live(c, live.data(), (int)live.size());
}

extern "C" int pwn_get_objects(pwn_ctx *c, pwn_sphere *out, int cap)
{
	if(GRP_HEAD(c)) { const int q = pwn_get_objects(GRP_M0(c), out, cap); if(q < 0) snprintf(c->err, sizeof(c->err), "%s", GRP_M0(c)->err); return q; }
	if(c == NULL || cap < 0 || (cap > 0 && out == NULL)) return PWN_EINVAL;
	std::vector<pwn_sphere> live;
	int rc = live_objects(c, live);
	if(rc != PWN_OK) return rc;
	for(size_t i = 0; i < live.size() && i < (size_t)cap; i++) out[i] = live[i];
	return (int)live.size();
}

// (a group, pwn_group.cpp: the list is binned once and every member uploads it)
int pwn_i_bin_spheres(const pwn_sphere *s, int n, pwn_binned *out)
{
	out->off.assign(4097, 0);
	const int nb = pwn_bin_spheres(s, n, out->off.data(), NULL, 0);
	if(nb < 0) return PWN_ENOMEM;
	out->idx.assign((size_t)(nb > 0 ? nb : 1), 0);
	if(pwn_bin_spheres(s, n, out->off.data(), out->idx.data(), nb) != nb) return PWN_ENOMEM;
	out->s.assign(s, s + n);
	// what pack_blob would refuse
	uint32_t nbin = 0;
	for(int i = 0; i < 4096; i++) { const uint32_t cnt = (uint32_t)(out->off[i + 1] - out->off[i]); if(cnt) nbin += cnt + 1u; }
	if(nbin > 32767u || (uint32_t)n * 32u >= PWN_LIST_END || ((pwn_t_total(nbin, (uint32_t)n) + 15u) & ~15u) > PWN_BLOB_MAX) return PWN_ETOOBIG;
	return PWN_OK;
}

int pwn_i_upload_binned(pwn_ctx *c, const pwn_binned &b)
{
	(void)hipSetDevice(c->device);
	c->spheres = b.s; c->bin_off = b.off; c->bin_idx = b.idx;
	c->blob_dirty = true;
	return pack_blob(c);
}

void pwn_i_set_object_table(pwn_ctx *c, const pwn_sphere *s, int n)
{
	c->objs.assign(s, s + n);
	c->obj_typ.assign((size_t)n, OBJ_SPHERE);
}

static int upload_live(pwn_ctx *c, const pwn_sphere *s, int n)
{
	(void)hipSetDevice(c->device);
	std::vector<int32_t> off(4097, 0);
	int nb = pwn_bin_spheres(s, n, off.data(), NULL, 0);
	if(nb < 0) return PWN_ENOMEM;
	std::vector<int32_t> idx((size_t)(nb > 0 ? nb : 1));
	if(pwn_bin_spheres(s, n, off.data(), idx.data(), nb) != nb) return PWN_ENOMEM;
	std::vector<pwn_sphere> keep_s(c->spheres);
	std::vector<int32_t> keep_o(c->bin_off), keep_i(c->bin_idx);
	c->spheres.assign(s, s + n);
	c->bin_off.swap(off);
	c->bin_idx.swap(idx);
	c->blob_dirty = true;
	int rc = pack_blob(c);
	if(rc != PWN_OK)
	{
		// keep the previous, working set
		c->spheres.swap(keep_s); c->bin_off.swap(keep_o); c->bin_idx.swap(keep_i);
		// too big: pack_blob gave up before it changed anything; a failed upload is tried again
		if(rc == PWN_ETOOBIG) c->blob_dirty = false; else (void)pack_blob(c);
	}
	return rc;
}

extern "C" int pwn_get_bins(pwn_ctx *c, uint16_t counts[4096], int32_t *idx, int cap)
{
	if(GRP_HEAD(c)) { const int rc = pwn_group_sync(c); return rc != PWN_OK ? rc : pwn_get_bins(GRP_M0(c), counts, idx, cap); }      // (the lists are the members' threads' work)
	if(c == NULL || counts == NULL) return PWN_EINVAL;
	for(int i = 0; i < 4096; i++) counts[i] = (uint16_t)(c->bin_off[i + 1] - c->bin_off[i]);
	int n = c->bin_off[4096];
	if(idx != NULL)
	{
		if(n > cap) return PWN_EINVAL;
		for(int i = 0; i < n; i++) idx[i] = c->bin_idx[i];
	}
	return n;
}

// ---- frame ------------------------------------------------------------------

// screen.h:43-57 in the reference build's operation order:
// rayb = (cam.x + cam.z) + (-yrat)*cam.y

// The trace kernel turns a unit number into (row of units, unit in the row): a division by the units per row, which
// is the same number for every unit of a launch.  For d >= 2 and s the largest shift with 2^s < d, M = ceil(2^(32+s) / d)
// fits 32 bits, and with 2^(32+s) = M d - e, 0 <= e < d: n M / 2^(32+s) = n / d + n e / (d 2^(32+s)), whose floor is
// floor(n / d) while n e < 2^(32+s); e < d <= 2^(s+1), so every n < 2^31 divides exactly (a frame has at most 2^24 units).
// d == 1 (frames up to 16 pixels wide): shift -1, the kernel divides.
static void unit_div_magic(uint32_t d, uint32_t *magic, int *shift)
{
	*magic = 0u; *shift = -1;
	if(d < 2u) return;
	int s = 0;
	while((2u << s) < d) s++;                  // 2^s < d <= 2^(s+1)
	const unsigned long long two = 1ull << (32 + s);
	const unsigned long long M = (two + d - 1u) / d;        // < 2^32: 2^s < d
	*magic = (uint32_t)M; *shift = s;
}

static void frame_setup(int w, int h, const float cam[16], pwn_trace_params *P)
{
	float dimx = (float)w, dimy = (float)h;
	float yrat = (-dimy) / dimx;
	float xsrat = -2.0f / dimx;
	float ysrat = (yrat + yrat) / dimy;
	for(int i = 0; i < 4; i++)
	{
		P->rayb[i] = (cam[0 + i] + cam[8 + i]) + (-yrat) * cam[4 + i];
		P->rdx[i] = xsrat * cam[0 + i];
		P->rdy[i] = ysrat * cam[4 + i];
		P->from[i] = cam[12 + i];
	}
}

// the unit-order entry of a stream (pwn_internal.h); NULL: none to spare right now
static pwn_ctx::unit_order_state *order_entry(pwn_ctx *c, hipStream_t stream, bool make)
{
	if(stream == c->stream) { c->order[0].key = stream; c->order[0].used = true; return &c->order[0]; }
	if(c->stream2 != NULL && stream == c->stream2) { c->order[1].key = stream; c->order[1].used = true; return &c->order[1]; }
	for(int i = 2; i < 4; i++) if(c->order[i].used && c->order[i].key == stream) { c->order[i].stamp = ++c->order_stamp; return &c->order[i]; }
	if(!make) return NULL;
	int pick = -1;
	for(int i = 2; i < 4; i++) if(!c->order[i].used) { pick = i; break; }
	if(pick < 0)
	{
		// both spare entries belong to other streams of the caller, whose kernels may still be using them: the older one is
		// taken over once the device is idle (a host that rotates over many streams pays for it; two are free)
		pick = c->order[2].stamp < c->order[3].stamp ? 2 : 3;
		if(hipDeviceSynchronize() != hipSuccess) return NULL;
	}
	pwn_ctx::unit_order_state &e = c->order[pick];
	e.key = stream; e.used = true; e.stamp = ++c->order_stamp; e.perm_valid = e.cost_fresh = false;
	return &e;
}

// (the events of the launch history belong to frame slots / the row tiling: forget them where those go, with the
// compute streams idle)
void pwn_launch_history_clear(pwn_ctx *c)
{
	for(int i = 0; i < 3; i++) { c->launch_stream[i] = NULL; c->launch_event[i] = NULL; }
}

// How many streams successive trace launches rotate over (pwn_tiled.cpp: 3 for a tiling on three compute streams, else 2).
// The work-queue counters are used in turn, 2R sets: a change starts them afresh, with the device idle.
int pwn_i_set_launch_rotation(pwn_ctx *c, int rot)
{
	if(rot != 2 && rot != 3) return PWN_EINVAL;
	if(c->launch_rot == rot) return PWN_OK;
	HIPCHK(c, hipDeviceSynchronize());
	HIPCHK(c, hipMemset(c->d_tickets, 0, PWN_TICKET_SETS * PWN_QUEUES * PWN_QUEUE_STRIDE * sizeof(uint32_t)));
	c->ticket_set = 0; c->launch_rot = rot;
	pwn_launch_history_clear(c);
	return PWN_OK;
}

int pwn_i_launch_trace(pwn_ctx *c, const float cam[16], float sec, int y0, int y1,
	uint32_t *d_sbuf, float *d_zbuf, hipStream_t stream)
{
	uint32_t *clear_word = c->trace_clear_word;            // for this launch only
	c->trace_clear_word = NULL;
	uint32_t *cost_word = c->trace_cost_word;
	c->trace_cost_word = NULL;
	hipEvent_t caller_event = c->trace_tables_event;       // (pwn_internal.h)
	c->trace_tables_event = NULL;
	if(!c->have_level) return PWN_ENOLEVEL;
	if(c->blob_dirty) { int rc = pack_blob(c); if(rc != PWN_OK) return rc; }
	if(y1 == y0) return PWN_OK;
	pwn_trace_params P;
	memset(&P, 0, sizeof(P));
	frame_setup(c->w, c->h, cam, &P);
	P.sec_current = sec;
	P.w = c->w; P.h = c->h; P.y0 = y0; P.y1 = y1;
	const int tw = pwn_trace_tile_w();
	P.tiles_x = (c->w + tw - 1) / tw;
	const int th = pwn_trace_tile_h();
	P.tiles_total = P.tiles_x * ((y1 - y0 + th - 1) / th);
	unit_div_magic((uint32_t)P.tiles_x, &P.ux_magic, &P.ux_shift);
	// (the kernel takes unit / tiles_x = unit where there is no shift: true for one unit per row only)
	if(P.ux_shift < 0 && P.tiles_x != 1) { snprintf(c->err, sizeof(c->err), "no division constant for %d units per row", P.tiles_x); return PWN_EINVAL; }
	P.blob_bytes = (uint32_t)c->blob.size();
	P.off_sph = c->off_sph;
	P.off_recsph = c->off_recsph;
	P.sbuf = d_sbuf; P.zbuf = d_zbuf;
	const int cur = c->blob_cur;
	P.blob = (const uint32_t *)c->d_blob[cur];
	// the upload of these tables runs on its own stream: this launch comes after it
	if(c->upload_pending[cur])
	{
		if(hipEventQuery(c->ev_upload[cur]) == hipSuccess) c->upload_pending[cur] = false;
		else HIPCHK(c, hipStreamWaitEvent(stream, c->ev_upload[cur], 0));
	}
	P.counters = c->d_counters;
	P.clear_word = clear_word;
	P.cost_word = cost_word;
	P.wave_log = NULL;
	// the kernel's work queues: 2R sets, R = launch_rot; launch n counts in set n mod 2R and clears set (n + R) mod 2R,
	// the one of the launch R launches on.  Launches of a context are stream-ordered (include/pwnhip.h) -- on ONE stream,
	// or rotating over the R = 2 compute streams of the frames in flight (pwn_submit_frame) or the R = 2 or 3 of a row
	// tiling, where launch n + R is behind launch n on its stream and the launches between may run beside it with sets
	// of their own.
	const unsigned rot = (unsigned)c->launch_rot, nsets = 2u * rot;
	P.tickets = c->d_tickets + (c->ticket_set % nsets) * PWN_QUEUES * PWN_QUEUE_STRIDE;
	P.tickets_next = c->d_tickets + ((c->ticket_set + rot) % nsets) * PWN_QUEUES * PWN_QUEUE_STRIDE;
	// ordinary cameras (rows x,y,z with w = 0, position w = 1: mat4_iden + rotations,
	// main.c:61-64) never put anything but 0 / 1 into the w lanes; the kernel has a
	// 3-lane specialisation for them that is arithmetically identical
	P.has_w = !(cam[3] == 0.0f && cam[7] == 0.0f && cam[11] == 0.0f && cam[15] == 1.0f);
	// test hook (tests/test_gpu_fuzz.py): send every camera through the general variant
	if(c->dbg_force_hasw) P.has_w = 1;
	if(c->counters_on) HIPCHK(c, hipMemsetAsync(c->d_counters, 0, PWN_NCOUNTERS * sizeof(unsigned long long), stream));
	// persistent grid: as many workgroups as are resident at once, each striding over tiles
	const bool refill = c->scheduler == PWN_SCHED_REFILL;
	const size_t lds_bytes = ((P.blob_bytes + 15u) & ~15u) + (refill ? pwn_trace_refill_lds_extra(P.has_w != 0) : pwn_trace_lds_extra());
	// resident workgroups per CU depend on (LDS bytes, kernel variant) only: ask once per combination
	P.scheduler = c->scheduler;
	P.refill_limit = c->refill_limit;
	const int variant = (refill ? 4 : 0) | (c->counters_on ? 2 : 0) | (P.has_w ? 1 : 0);
	if(c->occ_lds[variant] != lds_bytes)
	{
		c->occ_blocks[variant] = refill ? pwn_trace_refill_blocks_per_cu(lds_bytes, c->counters_on != 0, P.has_w != 0)
		                                : pwn_trace_blocks_per_cu(lds_bytes, c->counters_on != 0, P.has_w != 0);
		c->occ_lds[variant] = lds_bytes;
	}
	int per_cu = c->occ_blocks[variant];
	// The API counts LDS as 160 KiB / bytes; a census (tools/ubench/lds_census.hip) shows one
	// workgroup fewer resident at some sizes (5 x 32 KiB, 3 x 54000 B).  A persistent grid
	// that is one workgroup per CU too large runs its surplus at the end at a fraction of the
	// occupancy (+40 % frame time), so stay on the safe side: 2 KiB granules in 152 KiB.
	{
		const int lds_fit = (int)(155648u / (((uint32_t)lds_bytes + 2047u) & ~2047u));
		if(per_cu > lds_fit) per_cu = lds_fit;
	}
	if(per_cu < 1) per_cu = 1;
	if(c->dbg_blocks_per_cu > 0) per_cu = c->dbg_blocks_per_cu;                                      // experiments
	int grid = c->num_cus * per_cu;
	// row tiling over RCCL: a persistent grid that fills every CU leaves RCCL's send / recv kernels no registers
	// to start with (5 waves x 96 VGPRs of 512 per SIMD), and the exchange would only run in the gaps between the
	// kernels; a few workgroups fewer leave room on some CUs (pwn_tiled.cpp sets the number)
	// ... and frames on two compute streams: room for the other stream's kernels (PWN_OPT_TRACE_ROOM; the caller says how much)
	{
		if(c->room.mode >= 0) c->launch_room = c->room.mode;        // (a host that set a number gets it for every launch)
		const int reserve = c->grid_reserve > c->launch_room ? c->grid_reserve : c->launch_room;
		c->launch_room = 0;
		c->cost_mul = c->cost_div = (uint32_t)grid;
		if(reserve > 0 && grid > 2 * reserve) grid -= reserve;
		c->cost_div = (uint32_t)grid;
	}
	// fewer units than resident waves (a 320 x 240 frame is 1200 units for 5120 waves): one unit per wave, a
	// workgroup per four of them -- a workgroup whose waves find nothing still copies the tables into LDS
	{
		const int wg_waves = 4;          // PWN_BLOCK / 64
		if(!refill && grid > (P.tiles_total + wg_waves - 1) / wg_waves) grid = (P.tiles_total + wg_waves - 1) / wg_waves;
		if(grid > P.tiles_total) grid = P.tiles_total;
	}
	// PWN_OPT_WAVE_LOG: every wave of this launch writes its start and end time; entry 0 is unused, entry
	// 1 + 4 * workgroup + SIMD is a wave's (the buffer follows the grid of the launch)
	if(c->wave_log_on)
	{
		const size_t entries = (size_t)grid * 4 + 1;
		if(entries > c->wave_log_cap)
		{
			// (a kernel of an earlier launch that writes the old buffer may be on either compute stream)
			if(c->d_wave_log) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(c->d_wave_log); c->d_wave_log = NULL; c->wave_log_cap = 0; }
			HIPCHK(c, hipMalloc((void **)&c->d_wave_log, entries * 16));
			c->wave_log_cap = entries;
		}
		HIPCHK(c, hipMemsetAsync(c->d_wave_log, 0, c->wave_log_cap * 16, stream));
		P.wave_log = c->d_wave_log;
	}
	// PWN_OPT_UNIT_ORDER (and the wave log, for tools/unit_order_sim.py): this launch writes what every unit cost its wave,
	// and hands its units out in the order sorted from the last launch of the same rows on this stream
	// (with the wave log only where the dump is asked for, PWN_DBG_UNIT_COST: the kernel variant that writes the costs is 2-3 % slower,
	// and the wave log's span and residency are figures of the ordinary launch)
	const bool want_cost = !refill && (c->unit_order || (c->wave_log_on && getenv("PWN_DBG_UNIT_COST") != NULL));
	pwn_ctx::unit_order_state *uop = order_entry(c, stream, want_cost);
	if(uop != NULL && want_cost)
	{
		pwn_ctx::unit_order_state &uo = *uop;
		const size_t units = (size_t)P.tiles_total;
		const uint32_t qcap = (uint32_t)((units + PWN_QUEUES - 1u) / PWN_QUEUES);
		if(units > uo.cost_cap || (size_t)qcap * PWN_QUEUES > uo.perm_cap)
		{
			// (kernels that use the old buffers may still run, on either stream)
			if(uo.d_cost || uo.d_perm) HIPCHK(c, hipDeviceSynchronize());
			(void)hipFree(uo.d_cost); (void)hipFree(uo.d_perm);
			uo.d_cost = NULL; uo.d_perm = NULL; uo.cost_cap = uo.perm_cap = 0; uo.perm_valid = uo.cost_fresh = false;
			HIPCHK(c, hipMalloc((void **)&uo.d_cost, units * 2));
			HIPCHK(c, hipMalloc((void **)&uo.d_perm, (size_t)qcap * PWN_QUEUES * 4));
			uo.cost_cap = units; uo.perm_cap = (size_t)qcap * PWN_QUEUES;
		}
		P.unit_cost = uo.d_cost;
		if(c->unit_order && uo.perm_valid && uo.perm_units == (uint32_t)units && uo.perm_y0 == y0 && uo.perm_y1 == y1)
		{
			P.perm = uo.d_perm; P.perm_cap = qcap;
			c->order_used++;
		}
		uo.units = (uint32_t)units; uo.qcap = qcap; uo.y0 = y0; uo.y1 = y1; uo.cost_fresh = true;
	}
	else if(uop != NULL) uop->cost_fresh = false;
	// This launch clears the ticket set that the launch R before it drew from (above): it has to come after that one.
	// On one stream and in the rotation over R it does by itself; a launch that leaves the pattern waits.
	if(c->launch_event[rot - 1] != NULL && c->launch_stream[rot - 1] != stream)
	{
		HIPCHK(c, hipStreamWaitEvent(stream, c->launch_event[rot - 1], 0));
		c->launch_waits++;
	}
	if(refill) HIPCHK(c, pwn_launch_trace_refill(&P, grid, lds_bytes, c->counters_on != 0, stream));
	else HIPCHK(c, pwn_launch_trace(&P, grid, lds_bytes, c->counters_on != 0, stream));
	c->ticket_set++;                     // only a launch that went out has cleared the other set
	c->launch_stream[2] = c->launch_stream[1]; c->launch_event[2] = c->launch_event[1];
	c->launch_stream[1] = c->launch_stream[0]; c->launch_event[1] = c->launch_event[0];
	c->launch_stream[0] = stream; c->launch_event[0] = caller_event != NULL ? caller_event : c->ev_tables[cur];
	if(caller_event != NULL)
	{
		// The caller is about to record this event again.  Its last record was behind a frame the caller has since
		// waited for on the host (a free slot), so a copy of the tables still guarded by that record is not in
		// use any more -- and must stop pointing at the event, or its next upload would wait for THIS frame,
		// the newest one in flight (four copies with three slots did exactly that: 20 us of idle GPU per frame).
		for(int i = 0; i < PWN_NBLOB; i++)
			if(c->tables_in_use[i] && c->tables_wait[i] == caller_event) c->tables_in_use[i] = false;
		c->tables_wait[cur] = caller_event;
	}
	else
	{
		HIPCHK(c, hipEventRecord(c->ev_tables[cur], stream));
		c->tables_wait[cur] = c->ev_tables[cur];
	}
	c->tables_in_use[cur] = true;
	return PWN_OK;
}

// Behind the last kernel of a frame on `stream`: the costs its trace launch wrote become the hand-out order of the next
// launch of the same rows on that stream.  ~10 us of 64 workgroups at 4K, which the caller puts where nobody waits for
// it (behind the frame's "done" event / its copies to the host).
int pwn_i_launch_order(pwn_ctx *c, hipStream_t stream)
{
	pwn_ctx::unit_order_state *uop = order_entry(c, stream, false);
	if(uop == NULL) return PWN_OK;
	pwn_ctx::unit_order_state &uo = *uop;
	if(!c->unit_order || !uo.cost_fresh || uo.units < 4u * PWN_QUEUES) return PWN_OK;
	uo.cost_fresh = false;
	HIPCHK(c, pwn_launch_order(uo.d_cost, uo.units, uo.qcap, uo.d_perm, stream));
	uo.perm_valid = true; uo.perm_units = uo.units; uo.perm_y0 = uo.y0; uo.perm_y1 = uo.y1;
	c->order_sorts++;
	return PWN_OK;
}

// the sort by itself, host arrays in and out (tests): perm_out has 64 * ceil(units / 64) entries, queue q's at [q * cap, q * cap + its length)
extern "C" int pwn_unit_order_probe(pwn_ctx *c, const uint16_t *cost, uint32_t units, uint32_t *perm_out)
{
	if(GRP_HEAD(c)) return pwn_group_on_member0(c, [cost, units, perm_out](pwn_ctx *m0) { return pwn_unit_order_probe(m0, cost, units, perm_out); });
	if(c == NULL || cost == NULL || perm_out == NULL || units == 0u) return PWN_EINVAL;
	const uint32_t cap = (units + PWN_QUEUES - 1u) / PWN_QUEUES;
	(void)hipSetDevice(c->device);
	uint16_t *d_cost = NULL; uint32_t *d_perm = NULL;
	int rc = PWN_OK;
	if(hipMalloc((void **)&d_cost, (size_t)units * 2) != hipSuccess || hipMalloc((void **)&d_perm, (size_t)cap * PWN_QUEUES * 4) != hipSuccess) rc = PWN_ENOMEM;
	if(rc == PWN_OK && (hipMemcpy(d_cost, cost, (size_t)units * 2, hipMemcpyHostToDevice) != hipSuccess ||
	   hipMemset(d_perm, 0xff, (size_t)cap * PWN_QUEUES * 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) rc = PWN_EHIP;
	if(rc == PWN_OK && pwn_launch_order(d_cost, units, cap, d_perm, c->stream) != hipSuccess) rc = PWN_EHIP;
	if(rc == PWN_OK && (hipStreamSynchronize(c->stream) != hipSuccess ||
	   hipMemcpy(perm_out, d_perm, (size_t)cap * PWN_QUEUES * 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = PWN_EHIP;
	(void)hipFree(d_cost); (void)hipFree(d_perm);
	return rc;
}

extern "C" int pwn_launch_order_waits(pwn_ctx *c, unsigned long long *out)
{
	if(GRP_HEAD(c)) return pwn_launch_order_waits(GRP_M0(c), out);
	if(c == NULL || out == NULL) return PWN_EINVAL;
	*out = c->launch_waits;
	return PWN_OK;
}

extern "C" int pwn_unit_order_state(pwn_ctx *c, unsigned long long out[4])
{
	if(GRP_HEAD(c)) return pwn_unit_order_state(GRP_M0(c), out);
	if(c == NULL || out == NULL) return PWN_EINVAL;
	out[0] = (unsigned long long)c->unit_order; out[1] = c->order_used; out[2] = c->order_sorts;
	out[3] = c->order[0].perm_valid ? c->order[0].perm_units : 0ull;
	return PWN_OK;
}

int pwn_i_launch_blur(pwn_ctx *c, int y0, int y1, const uint32_t *d_pre, const float *d_z, uint32_t *d_out, hipStream_t stream,
	int avail_y0, int avail_y1, uint32_t *d_miss, uint32_t *d_cost_acc, uint32_t *d_cost_out)
{
	if((c->w & 3) != 0) return PWN_EINVAL; // screen.h:88,117: aligned 16-B store per group
	pwn_blur_params B;
	B.w = c->w; B.h = c->h; B.y0 = y0; B.y1 = y1;
	B.groups = c->w / 4;
	B.pre = d_pre; B.zbuf = d_z; B.out = d_out; B.skip = c->d_skip;
	B.avail_y0 = avail_y0; B.avail_y1 = avail_y1; B.miss = d_miss;
	B.cost_acc = d_cost_acc; B.cost_out = d_cost_out;
	B.cost_mul = c->blur_cost_mul ? c->blur_cost_mul : 1u; B.cost_div = c->blur_cost_div ? c->blur_cost_div : 1u;
	c->blur_cost_mul = c->blur_cost_div = 0u;
	// The workgroup's tile of output pixels (post_kernels.hip), chosen by what the frame rate on two streams said
	// (profiles/r3_blur_sweep.txt): 32 x 32 for wide frames and their strips (256-thread workgroups with 17 KB of LDS find room
	// beside the trace grid's workgroups that a 1024-thread one with 42 KB does not: 4K +1.9 %, 8K +5.3 %, the strips of an
	// 8-way 4K tiling -3.5 % kernel time against the 128 x 32 of rounds 1-2), 128 x 16 for narrow ones (720p +4.8 %, 1080p
	// +3.4 %; 32 x 32 loses 6 % at 720p).  Any shape gives the same pixels.
	{
		// With room beside the trace grid (PWN_OPT_TRACE_ROOM, what level.txt-like scenes settle on) 32 x 32 wins on narrow frames
		// too: 720p 16.4-17.5 -> 19.4 Gpixels/s (profiles/r3_blur_sweep.txt, last block).
		B.tile_w = 32; B.tile_h = 32; B.batch = 1;
		if(c->w < 2560 && pwn_room_for_launch(c) == 0) { B.tile_w = 128; B.tile_h = 16; }
		if(c->dbg_blur_th > 0) B.tile_h = c->dbg_blur_th;          // (a shape without an instantiation: the launch fails with hipErrorInvalidValue)
		if(c->dbg_blur_tw > 0) B.tile_w = c->dbg_blur_tw;
		if(c->dbg_blur_batch >= 0) B.batch = c->dbg_blur_batch;
	}
	HIPCHK(c, pwn_launch_blur(&B, stream));
	return PWN_OK;
}

extern "C" int pwn_trace_rows_device(pwn_ctx *c, const float cam[16], float sec, int y0, int y1,
	void *d_sbuf, void *d_zbuf, void *stream)
{
	GRP_REFUSE(c, "pwn_trace_rows_device");
	if(c == NULL || cam == NULL || d_sbuf == NULL || d_zbuf == NULL || y0 < 0 || y1 > c->h || y0 > y1) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	return pwn_i_launch_trace(c, cam, sec, y0, y1, (uint32_t *)d_sbuf, (float *)d_zbuf, (hipStream_t)stream);
}

extern "C" int pwn_blur_rows_device(pwn_ctx *c, int y0, int y1, const void *d_pre, const void *d_zbuf, void *d_out, void *stream)
{
	GRP_REFUSE(c, "pwn_blur_rows_device");
	if(c == NULL || d_pre == NULL || d_zbuf == NULL || d_out == NULL || y0 < 0 || y1 > c->h || y0 > y1 || d_pre == d_out) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	int rc = pwn_i_launch_blur(c, y0, y1, (const uint32_t *)d_pre, (const float *)d_zbuf, (uint32_t *)d_out, (hipStream_t)stream, 0, 0, NULL, NULL, NULL);
	// PWN_OPT_UNIT_ORDER: a strip's trace and blur on one stream of the caller's -- the order of that stream's next trace of the same rows
	if(rc == PWN_OK) rc = pwn_i_launch_order(c, (hipStream_t)stream);
	return rc;
}

extern "C" int pwn_blur_rows_device_bounded(pwn_ctx *c, int y0, int y1, const void *d_pre, const void *d_zbuf, void *d_out,
	int avail_y0, int avail_y1, void *d_miss, void *stream)
{
	GRP_REFUSE(c, "pwn_blur_rows_device_bounded");
	if(c == NULL || d_pre == NULL || d_zbuf == NULL || d_out == NULL || d_miss == NULL || y0 < 0 || y1 > c->h || y0 > y1 ||
	   d_pre == d_out || avail_y0 > avail_y1) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	int rc = pwn_i_launch_blur(c, y0, y1, (const uint32_t *)d_pre, (const float *)d_zbuf, (uint32_t *)d_out, (hipStream_t)stream,
		avail_y0, avail_y1, (uint32_t *)d_miss, NULL, NULL);
	if(rc == PWN_OK) rc = pwn_i_launch_order(c, (hipStream_t)stream);
	return rc;
}

// ---- the blocking call in row strips (PWN_OPT_CALL_STRIPS, include/pwnhip.h) ----
// One pwn_trace_screen_centred of a 4K frame is 0.32 ms of trace, 0.03 ms of blur and 0.64 ms of PCIe (33 MB at the
// link's 52 GB/s): one after the other the GPU idles for two thirds of the call.  In strips the copy starts when the
// first strips are blurred and runs beside the kernels of the rest: the call is the copy plus the first strips.
// Strips grow from the top -- the first ones small so that the copy starts early, each later one 1.2 times the one before.

// rows a blur tap of depth <= `depth` reaches (screen.h:100-102): 24 is the row tiling's halo; the strips of a blocking call begin with 8
// (level.txt from its spawn pose: taps reach 32 rows at 4K, depth 8 covers 36) and go to 24 after the first frame whose taps went further
static int blur_reach_rows(int h, double depth = 24.0) { return (int)(0.002 * h * depth) + 2; }

static int strip_cuts(const pwn_ctx *c, int want, int *cuts)
{
	const int h = c->h;
	int n = 0;
	cuts[0] = 0;
	if(want >= 2)
	{
		const int per = ((h + want - 1) / want + 31) & ~31;           // (the blur works in 32-row tiles)
		while(cuts[n] < h && n < PWN_CALL_STRIPS_MAX) { cuts[n + 1] = cuts[n] + per < h ? cuts[n] + per : h; n++; }
		cuts[n] = h;
		return n;
	}
	// (measured at 4K, profiles/r5/call_strips.txt: first strip 96 / 128 / 160 / 192 rows x growth 1.2 / 1.4 / 1.6 -- 128 x 1.2, eight
	// strips, was the fastest; 1.6 costs 10 %: the last chunk's copy starts when everything else is done.  With the round's final trace
	// kernel 160 rows, seven strips, is 0.5-0.8 % faster at 4K on one copy stream and on two, 128 against 96 rows 2.8 % at 1440p,
	// 320 against 256 equal at 8K: the first strip is 2/27 of the frame)
	double grow = 1.2;
	int rows = (h * 2 / 27 + 31) & ~31;
	if(const char *e = getenv("PWN_DBG_STRIP_FIRST")) if(atoi(e) >= 8) rows = (atoi(e) + 7) & ~7;        // (experiments)
	if(const char *e = getenv("PWN_DBG_STRIP_GROW")) if(atof(e) >= 1.0) grow = atof(e);
	while(cuts[n] < h)
	{
		int y = cuts[n] + rows;
		if(n == PWN_CALL_STRIPS_MAX - 1 || h - y < rows / 2) y = h;     // (a rest of less than half a strip goes with this one)
		cuts[++n] = y;
		rows = ((int)((double)rows * grow) + 31) & ~31;
	}
	return n;
}

// registered (pwn_host_register, hipHostRegister) or allocated pinned: a copy into it is DMA that does not hold the caller
static bool host_is_pinned(const void *p, size_t bytes)
{
	if(p == NULL) return true;
	const void *ends[2] = { p, (const char *)p + bytes - 1 };
	for(int i = 0; i < 2; i++)
	{
		hipPointerAttribute_t a;
		memset(&a, 0, sizeof(a));
		if(hipPointerGetAttributes(&a, ends[i]) != hipSuccess) { (void)hipGetLastError(); return false; }
		if(a.type != hipMemoryTypeHost) return false;
	}
	return true;
}

extern "C" int pwn_host_register(pwn_ctx *c, void *base, size_t bytes)
{
	if(GRP_HEAD(c)) return (base == NULL || bytes == 0) ? PWN_EINVAL : pwn_group_host_register(c, base, bytes);
	if(c == NULL || base == NULL || bytes == 0) return PWN_EINVAL;
	for(int i = 0; i < c->host_regs_n; i++) if(c->host_regs[i].base == base) return c->host_regs[i].bytes >= bytes ? PWN_OK : PWN_EINVAL;
	if(c->host_regs_n >= PWN_HOST_REGS_MAX) return PWN_EBUSY;
	(void)hipSetDevice(c->device);
	const hipError_t e = hipHostRegister(base, bytes, hipHostRegisterPortable);
	if(e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return PWN_OK; }       // (somebody else's registration: theirs to undo)
	if(e != hipSuccess) { (void)hipGetLastError(); snprintf(c->err, sizeof(c->err), "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e)); return PWN_EHIP; }
	c->host_regs[c->host_regs_n].base = base; c->host_regs[c->host_regs_n].bytes = bytes; c->host_regs_n++;
	return PWN_OK;
}

extern "C" int pwn_host_unregister(pwn_ctx *c, void *base)
{
	if(GRP_HEAD(c)) return base == NULL ? PWN_EINVAL : pwn_group_host_unregister(c, base);
	if(c == NULL || base == NULL) return PWN_EINVAL;
	for(int i = 0; i < c->host_regs_n; i++)
		if(c->host_regs[i].base == base)
		{
			(void)hipSetDevice(c->device);
			// (a copy of an earlier call into it is complete: the blocking call returns behind its copies)
			const hipError_t e = hipHostUnregister(base);
			c->host_regs[i] = c->host_regs[--c->host_regs_n];
			if(e != hipSuccess) { (void)hipGetLastError(); snprintf(c->err, sizeof(c->err), "hipHostUnregister: %s", hipGetErrorString(e)); return PWN_EHIP; }
			return PWN_OK;
		}
	return PWN_EINVAL;
}

extern "C" int pwn_call_strips_state(pwn_ctx *c, unsigned long long out[6])
{
	if(GRP_HEAD(c)) return pwn_call_strips_state(GRP_M0(c), out);
	if(c == NULL || out == NULL) return PWN_EINVAL;
	out[0] = (unsigned long long)(long long)c->call_strips; out[1] = (unsigned long long)c->strips_last; out[2] = c->strip_calls; out[3] = c->strip_redone;
	out[4] = (unsigned long long)c->strip_copy_streams; out[5] = c->strip_reach ? 24ull : 8ull;
	return PWN_OK;
}

static int call_in_strips(pwn_ctx *c, const float cam[16], float sec, uint32_t *sbuf, float *zbuf, const int *cuts, int K)
{
	// Chunks go to the host on ONE copy stream or on TWO in turn.  A DMA copy behind another on one stream starts ~11 us after that one
	// ended, and small copies run at 35-47 GB/s; on two streams the next copy's start-up hides behind the current one's bytes -- on
	// some boxes: 0.745-0.755 against 0.77-0.79 ms per 4K call on three of them, 0.805 against 0.773 on a fourth
	// (profiles/r5/call_strips.txt).  So a context finds out once: its first 8 calls in strips use one stream, the next 8 two, and the
	// faster (by the median of the calls' own wall times, the first of each eight left out) is kept.  PWN_CALL_COPY_STREAMS=1|2 in the
	// environment fixes it.
	if(c->copy_stream2 == NULL && hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->copy_stream2 = NULL; }
	int ncopy = c->strip_copy_streams;
	if(ncopy == 0) ncopy = c->strip_calib_n < 8 ? 1 : 2;               // (finding out)
	if(c->copy_stream2 == NULL) ncopy = 1;
	hipStream_t cs = c->copy_stream, cpy[2] = { c->copy_stream, ncopy == 2 ? c->copy_stream2 : c->copy_stream };
	struct timespec t_call0;
	clock_gettime(CLOCK_MONOTONIC, &t_call0);
	// Successive strips' trace launches alternate between the context's two compute streams (the pattern the work-queue
	// counters are made for, include/pwnhip.h): strip k + 1's grid moves onto the CUs as strip k's waves run out of units.
	// On one stream every launch waits for the tail of the one before it: 8 strips of a 4K frame took 0.61 ms where the
	// whole frame takes 0.35 (two streams: 0.43).
	hipStream_t st[2] = { c->stream, (c->frame_overlap && c->stream2 != NULL) ? c->stream2 : c->stream };
	if(const char *e = getenv("PWN_DBG_STRIP_STREAMS")) if(atoi(e) == 1) st[1] = st[0];
	const bool two = st[1] != st[0];
	const size_t W = (size_t)c->w, n = W * (size_t)c->h;
	const bool blur = c->blur_passes == 1;
	for(; c->strip_ev_n < 2 * K; c->strip_ev_n++) HIPCHK(c, hipEventCreateWithFlags(&c->strip_ev[c->strip_ev_n], hipEventDisableTiming));
	if(c->d_strip_miss == NULL)
	{
		HIPCHK(c, hipMalloc((void **)&c->d_strip_miss, 64));
		HIPCHK(c, hipMemset(c->d_strip_miss, 0, 64));
		HIPCHK(c, hipHostMalloc((void **)&c->h_strip_miss, 64, hipHostMallocDefault));
		HIPCHK(c, hipDeviceSynchronize());
	}
	*c->h_strip_miss = 0u;
	uint32_t *pre = c->d_pre, *fin = blur ? c->d_out : c->d_pre;
	// The blur is cut apart from the trace: behind strip k's trace the rows [0, cuts[k + 1]) are there, and every row whose
	// taps of depth <= 8 (24 once a frame's taps went further) stay inside them is blurred and sent -- chunk k = rows [sent, cuts[k + 1] - H), the last one to the end.
	// So the first bytes leave behind the FIRST strip's trace, not the second's.
	int H = blur ? blur_reach_rows(c->h, c->strip_reach ? 24.0 : 8.0) : 0;
	if(const char *e = getenv("PWN_DBG_STRIP_REACH")) if(*e && blur) H = atoi(e);
	// A copy into memory the device does not know is staged by the runtime and holds this thread until it is through:
	// two strips of kernels are then enqueued ahead of every copy, so that the GPU has work while the host waits in it.
	// (Measured and dropped, profiles/r5/call_strips.txt: the blur storing its pixels into the registered buffer itself, and a
	// copy kernel per chunk in place of the DMA copy -- stores to host memory that come from the CUs back up into the trace
	// grids' own stores: the strips' traces of a 4K frame went from 0.43 to 0.9 ms.)
	const bool pinned = host_is_pinned(sbuf, n * 4) && host_is_pinned(zbuf, n * 4);
	int ahead = pinned ? 0 : 2;
	if(const char *e = getenv("PWN_DBG_STRIP_AHEAD")) if(*e) ahead = atoi(e);
	// (workgroups the grids leave free for the other stream's kernels: the blurs find a place at once.  One per CU was the best of
	// 0 / 128 / 256 / 512 / 768 when the plan was made; with the round's final kernels 384 is 2 % faster than 256 on two copy streams
	// (0.732 -> 0.716 ms per 4K call), 512 nearly so, and equal on one: tools/r5/strip_plan_ab.py, profiles/r5/call_strips.txt)
	int room = two ? c->num_cus + c->num_cus / 2 : 0;
	if(const char *e = getenv("PWN_DBG_STRIP_ROOM")) if(*e) room = atoi(e);
	int traced = 0, chunks = 0, copied = 0, rc;
	int chunk_y[PWN_CALL_STRIPS_MAX + 1];          // chunk j = rows [chunk_y[j], chunk_y[j + 1]), blurred behind the trace of strip chunk_k[j]
	chunk_y[0] = 0;
	// PWN_DBG_STRIP_TIMELINE: when every strip's trace, every chunk's blur and copy ended (experiments: timing events between the kernels cost a few us each)
	static hipEvent_t tl[4 * PWN_CALL_STRIPS_MAX];
	const bool timeline = getenv("PWN_DBG_STRIP_TIMELINE") != NULL;
	if(timeline && tl[0] == NULL) for(int i = 0; i < 4 * PWN_CALL_STRIPS_MAX; i++) HIPCHK(c, hipEventCreate(&tl[i]));
#define STRIPS_FAIL(code) do { (void)hipStreamSynchronize(st[0]); (void)hipStreamSynchronize(st[1]); (void)hipStreamSynchronize(cpy[0]); (void)hipStreamSynchronize(cpy[1]); return (code); } while(0)
	HIPCHK(c, hipEventRecord(c->ev[0], st[0]));
	while(traced < K || copied < chunks)
	{
		if(traced < K)
		{
			const int k = traced;
			hipStream_t ts = st[k & 1];
			if(k == 0 && blur) c->trace_clear_word = c->d_strip_miss;      // (cleared by the first strip's launch, in front of every blur)
			c->trace_tables_event = c->strip_ev[2 * k];                   // recorded right behind the launch; its previous record is a call that returned
			c->launch_room = room;
			rc = pwn_i_launch_trace(c, cam, sec, cuts[k], cuts[k + 1], pre, c->d_z, ts);
			if(rc != PWN_OK) { (void)hipEventRecord(c->strip_ev[2 * k], ts); STRIPS_FAIL(rc); }
			HIPCHK(c, hipEventRecord(c->strip_ev[2 * k], ts));
			if(timeline) HIPCHK(c, hipEventRecord(tl[4 * k], ts));
			traced++;
			if(traced == K) HIPCHK(c, hipEventRecord(c->ev[1], ts));
			// the chunk this trace completes, on its stream: behind it and -- one wait between the streams -- behind the newest
			// trace of the other stream (older ones are in front of those)
			int to = traced == K ? c->h : ((cuts[traced] - H) & ~31);
			if(to > chunk_y[chunks] || traced == K)
			{
				const int j = chunks, y0 = chunk_y[j];
				if(to < y0) to = y0;
				if(blur && to > y0)
				{
					if(two && k >= 1) HIPCHK(c, hipStreamWaitEvent(ts, c->strip_ev[2 * (k - 1)], 0));
					const bool whole = traced == K;
					rc = pwn_i_launch_blur(c, y0, to, pre, c->d_z, fin, ts, 0, whole ? 0 : cuts[traced], whole ? NULL : c->d_strip_miss, NULL, NULL);
					if(rc != PWN_OK) STRIPS_FAIL(rc);
				}
				else if(!blur && two && k >= 1) HIPCHK(c, hipStreamWaitEvent(ts, c->strip_ev[2 * (k - 1)], 0));      // (the chunk's rows may be the other stream's)
				HIPCHK(c, hipEventRecord(c->strip_ev[2 * j + 1], ts));
				if(timeline) HIPCHK(c, hipEventRecord(tl[4 * j + 1], ts));
				chunk_y[j + 1] = to;
				chunks++;
				if(traced == K) HIPCHK(c, hipEventRecord(c->ev[2], ts));
			}
		}
		while(copied < chunks && (traced == K || traced - copied > ahead))
		{
			const int j = copied;
			const size_t off = (size_t)chunk_y[j] * W, cnt = (size_t)(chunk_y[j + 1] - chunk_y[j]) * W * 4;
			hipStream_t xs = cpy[j & 1];
			HIPCHK(c, hipStreamWaitEvent(xs, c->strip_ev[2 * j + 1], 0));
			if(timeline) HIPCHK(c, hipEventRecord(tl[4 * j + 2], xs));
			if(cnt != 0)
			{
				if(zbuf != NULL) HIPCHK(c, hipMemcpyAsync(zbuf + off, c->d_z + off, cnt, hipMemcpyDeviceToHost, xs));
				HIPCHK(c, hipMemcpyAsync(sbuf + off, fin + off, cnt, hipMemcpyDeviceToHost, xs));
			}
			if(timeline) HIPCHK(c, hipEventRecord(tl[4 * j + 3], xs));
			copied++;
		}
	}
	if(cpy[1] != cpy[0])
	{
		// (the first copy stream behind the second's last copy; strip_ev[0] -- strip 0's trace, long over -- is free for it)
		HIPCHK(c, hipEventRecord(c->strip_ev[0], cpy[1]));
		HIPCHK(c, hipStreamWaitEvent(cs, c->strip_ev[0], 0));
	}
	if(blur && chunks > 0) HIPCHK(c, hipStreamWaitEvent(cs, c->strip_ev[2 * (chunks - 1) + 1], 0));       // (the last chunk's blur)
	if(blur) HIPCHK(c, hipMemcpyAsync(c->h_strip_miss, c->d_strip_miss, 4, hipMemcpyDeviceToHost, cs));
	HIPCHK(c, hipEventRecord(c->ev[3], cs));
	HIPCHK(c, hipEventSynchronize(c->ev[3]));
	if(c->strip_copy_streams == 0)
	{
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		c->strip_calib_ms[c->strip_calib_n++] = (double)(t1.tv_sec - t_call0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t_call0.tv_nsec) * 1e-6;
		if(c->strip_calib_n == 16)
		{
			double med[2];
			for(int m = 0; m < 2; m++)
			{
				double v[7];
				for(int i = 0; i < 7; i++) v[i] = c->strip_calib_ms[8 * m + 1 + i];
				for(int i = 1; i < 7; i++) for(int k = i; k > 0 && v[k] < v[k - 1]; k--) { const double t = v[k]; v[k] = v[k - 1]; v[k - 1] = t; }
				med[m] = v[3];
			}
			c->strip_copy_streams = med[1] < med[0] ? 2 : 1;
		}
	}
	if(timeline)
	{
		fprintf(stderr, "strips %d, chunks %d:", K, chunks);
		for(int k = 0; k < K; k++) { float a = 0; (void)hipEventElapsedTime(&a, c->ev[0], tl[4 * k]); fprintf(stderr, " T%d(%d rows) %.3f", k, cuts[k + 1] - cuts[k], a); }
		for(int j = 0; j < chunks; j++)
		{
			float b = 0, d = 0, e = 0;
			(void)hipEventElapsedTime(&b, c->ev[0], tl[4 * j + 1]); (void)hipEventElapsedTime(&d, c->ev[0], tl[4 * j + 2]); (void)hipEventElapsedTime(&e, c->ev[0], tl[4 * j + 3]);
			fprintf(stderr, "\n   chunk %d rows %d: blur end %.3f copy %.3f..%.3f", j, chunk_y[j + 1] - chunk_y[j], b, d, e);
		}
		fprintf(stderr, "\n");
	}
	c->strip_calls++;
	if(blur && *c->h_strip_miss != 0u)
	{
		// a tap landed below the rows that were traced when its chunk was blurred: the whole pre-blur frame is there now, so
		// the pass again in one piece, and the frame again to the host (depth is what it was); the next calls in one piece
		c->strip_redone++;
		// (the short reach was not enough for this view: the long one from now on; that one too: one-piece calls for a while)
		if(c->strip_reach == 0) c->strip_reach = 1; else c->strip_backoff = 64;
		rc = pwn_i_launch_blur(c, 0, c->h, pre, c->d_z, fin, st[0], 0, 0, NULL, NULL, NULL);
		if(rc != PWN_OK) return rc;
		HIPCHK(c, hipEventRecord(c->ev[2], st[0]));
		HIPCHK(c, hipMemcpyAsync(sbuf, fin, n * 4, hipMemcpyDeviceToHost, st[0]));
		HIPCHK(c, hipEventRecord(c->ev[3], st[0]));
		HIPCHK(c, hipEventSynchronize(c->ev[3]));
	}
#undef STRIPS_FAIL
	return PWN_OK;
}

extern "C" int pwn_trace_screen_centred(pwn_ctx *c, const float cam[16], float sec, uint32_t *sbuf, float *zbuf)
{
	if(GRP_HEAD(c)) return pwn_group_trace_screen_centred(c, cam, sec, sbuf, zbuf);
	if(c == NULL || cam == NULL || sbuf == NULL) return PWN_EINVAL;
	if(c->blur_passes > 0 && (c->w & 3) != 0) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	size_t n = (size_t)c->w * (size_t)c->h;
	hipStream_t s = c->stream;
	// behind the frames in flight, whichever compute stream their kernels are on
	if(c->last_frame_done != NULL && c->last_frame_stream != s) HIPCHK(c, hipStreamWaitEvent(s, c->last_frame_done, 0));
	if(c->last_frame_done != NULL && c->stream2 != NULL && c->last_frame_stream != c->stream2) HIPCHK(c, hipStreamWaitEvent(c->stream2, c->last_frame_done, 0));      // (strips use both)
	c->last_frame_done = NULL;       // (this call ends with the stream empty)
	// ---- in row strips, the copies beside the kernels (PWN_OPT_CALL_STRIPS)
	{
		int cuts[PWN_CALL_STRIPS_MAX + 1], K = 1;
		const bool can = c->blur_passes <= 1 && !c->counters_on && !c->wave_log_on && !c->unit_order && c->have_level;
		if(can && c->call_strips >= 2) K = strip_cuts(c, c->call_strips, cuts);
		// by frame size (profiles/r5/call_strips.txt): into registered buffers strips pay from 2560 x 1440 on (0.43 against 0.51 ms; 4K
		// 0.76 against 0.99, 8K 2.66 against 3.93) and not at 1080p (0.32 / 0.31); into pageable ones, where every chunk's copy holds
		// the caller, from 4K on (0.88 against 1.02 ms; 1440p equal, 1080p 0.44 against 0.33)
		else if(can && c->call_strips < 0 && c->h >= 256 &&
		        n >= ((host_is_pinned(sbuf, n * 4) && host_is_pinned(zbuf, n * 4)) ? 3000000u : 6000000u))
		{
			if(c->strip_backoff > 0) c->strip_backoff--;
			else K = strip_cuts(c, 0, cuts);
		}
		c->strips_last = K >= 2 ? K : 1;
		if(K >= 2)
		{
			const int rc = call_in_strips(c, cam, sec, sbuf, zbuf, cuts, K);
			if(rc != PWN_OK) return rc;
			if(c->blur_passes == 0) { uint32_t *t = c->d_pre; c->d_pre = c->d_out; c->d_out = t; }      // (the final frame stays addressable as d_out)
			(void)hipEventElapsedTime(&c->stats.trace_ms, c->ev[0], c->ev[1]);
			(void)hipEventElapsedTime(&c->stats.blur_ms, c->ev[1], c->ev[2]);
			(void)hipEventElapsedTime(&c->stats.total_ms, c->ev[0], c->ev[3]);
			return PWN_OK;
		}
	}
	HIPCHK(c, hipEventRecord(c->ev[0], s));
	// trace into d_pre; with blur on, d_pre plays tsbuf and d_out plays sbuf
	// (the memcpy of screen.h:75 becomes a pointer swap per pass)
	uint32_t *cur = c->d_pre, *other = c->d_out;
	int rc = pwn_i_launch_trace(c, cam, sec, 0, c->h, cur, c->d_z, s);
	if(rc != PWN_OK) return rc;
	HIPCHK(c, hipEventRecord(c->ev[1], s));
	for(int p = 0; p < c->blur_passes; p++)
	{
		rc = pwn_i_launch_blur(c, 0, c->h, cur, c->d_z, other, s, 0, 0, NULL, NULL, NULL);
		if(rc != PWN_OK) return rc;
		uint32_t *t = cur; cur = other; other = t;
	}
	HIPCHK(c, hipEventRecord(c->ev[2], s));
	HIPCHK(c, hipMemcpyAsync(sbuf, cur, n * 4, hipMemcpyDeviceToHost, s));
	if(zbuf != NULL) HIPCHK(c, hipMemcpyAsync(zbuf, c->d_z, n * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipEventRecord(c->ev[3], s));
	// (the units' order for the next call: behind the copies, and this call does not wait for it)
	rc = pwn_i_launch_order(c, s);
	if(rc != PWN_OK) return rc;
	HIPCHK(c, hipEventSynchronize(c->ev[3]));
	// keep the final frame addressable as d_out for pwn_screen_upscale(NULL,...)
	if(cur != c->d_out) { c->d_pre = c->d_out; c->d_out = cur; }
	(void)hipEventElapsedTime(&c->stats.trace_ms, c->ev[0], c->ev[1]);
	(void)hipEventElapsedTime(&c->stats.blur_ms, c->ev[1], c->ev[2]);
	(void)hipEventElapsedTime(&c->stats.total_ms, c->ev[0], c->ev[3]);
	return PWN_OK;
}

// ---- frames in flight ---------------------------------------------------------
// The reference's loop presents every frame on the host (main.c:107-109).  Over PCIe that
// copy takes longer than the kernels of a 4K frame, so a host that wants throughput keeps
// two or three frames in flight: while frame i travels to its pinned host buffers on the
// copy stream, the kernels of frame i+1 run on the compute stream.

static void slot_release(pwn_slot &sl)
{
	(void)hipFree(sl.d_out); (void)hipFree(sl.d_z); (void)hipFree(sl.d_surface);
	if(sl.h_sbuf) (void)hipHostFree(sl.h_sbuf);
	if(sl.h_zbuf) (void)hipHostFree(sl.h_zbuf);
	if(sl.h_surface) (void)hipHostFree(sl.h_surface);
	for(int k = 0; k < 4; k++) if(sl.ev_k[k]) (void)hipEventDestroy(sl.ev_k[k]);
	if(sl.ev_done) (void)hipEventDestroy(sl.ev_done);
	memset(&sl, 0, sizeof(sl));
}

static void frames_release(pwn_ctx *c)
{
	// A slot's "kernels done" event may be what an upload would wait for (pwn_ctx.trace_tables_event): no frame
	// is in flight here, so once the compute stream is empty no copy of the tables is in use any more
	if(c->nslots > 0)
	{
		if(c->stream) (void)hipStreamSynchronize(c->stream);
		if(c->stream2) (void)hipStreamSynchronize(c->stream2);
		for(int i = 0; i < PWN_NBLOB; i++) c->tables_in_use[i] = false;
	}
	c->last_frame_done = NULL;       // (a slot's event, destroyed below)
	pwn_launch_history_clear(c);
	for(int i = 0; i < PWN_MAX_SLOTS; i++) slot_release(c->slot[i]);
	c->nslots = 0;
}

extern "C" int pwn_frames_config(pwn_ctx *c, int nslots, int flags, int scale, int pitch_bytes)
{
	if(GRP_HEAD(c)) return pwn_group_frames_config(c, nslots, flags, scale, pitch_bytes);
	if(c == NULL || nslots < 0 || nslots > PWN_MAX_SLOTS || (flags & ~(PWN_FRAME_SBUF | PWN_FRAME_ZBUF | PWN_FRAME_SURFACE)) != 0) return PWN_EINVAL;
	if(flags & PWN_FRAME_SURFACE)
	{
		if(scale <= 0) return PWN_EINVAL;
		if(pitch_bytes == 0) pitch_bytes = c->w * scale * 4;
		if((pitch_bytes & 3) != 0 || (long long)pitch_bytes < (long long)c->w * scale * 4) return PWN_EINVAL;
	}
	else { scale = 1; pitch_bytes = 0; }
	(void)hipSetDevice(c->device);
	for(int i = 0; i < c->nslots; i++) if(c->slot[i].in_flight) return PWN_EBUSY;
	if(pwn_tiled_busy(c)) return PWN_EBUSY;            // (frames_release drops the guards of the tables their launches read)
	frames_release(c);
	const size_t n = (size_t)c->w * (size_t)c->h;
	const size_t surf_bytes = (size_t)pitch_bytes * (size_t)c->h * (size_t)scale;
	int rc = PWN_OK;
	for(int i = 0; i < nslots && rc == PWN_OK; i++)
	{
		pwn_slot &sl = c->slot[i];
		// every slot has its own final colour and depth planes: they are copied out (or looked at
		// by the caller) while the next frames are traced
		if(hipMalloc((void **)&sl.d_out, n * 4) != hipSuccess || hipMalloc((void **)&sl.d_z, n * 4) != hipSuccess) rc = PWN_ENOMEM;
		// like the context's depth plane: zero, and kept at pixels whose primary ray runs out of steps
		else if(hipMemset(sl.d_z, 0, n * 4) != hipSuccess || hipMemset(sl.d_out, 0, n * 4) != hipSuccess) rc = PWN_EHIP;
		if(rc == PWN_OK && (flags & PWN_FRAME_SBUF) && hipHostMalloc((void **)&sl.h_sbuf, n * 4, hipHostMallocDefault) != hipSuccess) rc = PWN_ENOMEM;
		if(rc == PWN_OK && (flags & PWN_FRAME_ZBUF) && hipHostMalloc((void **)&sl.h_zbuf, n * 4, hipHostMallocDefault) != hipSuccess) rc = PWN_ENOMEM;
		if(rc == PWN_OK && (flags & PWN_FRAME_SURFACE))
		{
			if(hipMalloc((void **)&sl.d_surface, surf_bytes) != hipSuccess || hipHostMalloc((void **)&sl.h_surface, surf_bytes, hipHostMallocDefault) != hipSuccess) rc = PWN_ENOMEM;
			else if(hipMemset(sl.d_surface, 0, surf_bytes) != hipSuccess) rc = PWN_EHIP;   // bytes between rows of a padded pitch stay 0
		}
		for(int k = 0; k < 4 && rc == PWN_OK; k++) if(hipEventCreate(&sl.ev_k[k]) != hipSuccess) rc = PWN_EHIP;
		if(rc == PWN_OK && hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) != hipSuccess) rc = PWN_EHIP;
	}
	// the pre-blur plane of the frames on the second compute stream (PWN_OPT_FRAME_OVERLAP)
	if(rc == PWN_OK && nslots >= 2 && c->d_pre2 == NULL)
	{
		if(hipMalloc((void **)&c->d_pre2, n * 4) != hipSuccess) rc = PWN_ENOMEM;
		else if(hipMemset(c->d_pre2, 0, n * 4) != hipSuccess) rc = PWN_EHIP;
	}
	// (the memsets above run on the null stream, which the non-blocking streams of the frames do not wait for)
	if(rc == PWN_OK && hipDeviceSynchronize() != hipSuccess) rc = PWN_EHIP;
	if(rc != PWN_OK) { frames_release(c); return rc; }
	c->nslots = nslots; c->frame_flags = flags; c->frame_scale = scale; c->frame_pitch = pitch_bytes;
	return PWN_OK;
}

extern "C" int pwn_submit_frame(pwn_ctx *c, const float cam[16], float sec, int slot)
{
	if(GRP_HEAD(c)) return pwn_group_submit_frame(c, cam, sec, slot);
	if(c == NULL || cam == NULL || slot < 0 || slot >= c->nslots) return PWN_EINVAL;
	if(c->blur_passes > 0 && (c->w & 3) != 0) return PWN_EINVAL;
	pwn_slot &sl = c->slot[slot];
	if(sl.in_flight) return PWN_EBUSY;
	(void)hipSetDevice(c->device);
	const size_t n = (size_t)c->w * (size_t)c->h;
	// The kernels of a frame go one after the other on a compute stream, the copies to the host on the copy
	// stream behind their frame's last kernel.  Frames ALTERNATE between two compute streams (PWN_OPT_FRAME_OVERLAP,
	// the default): the trace grid of frame f+1 moves onto the CUs as the waves of frame f's grid run out of units
	// and leave, and the blur of frame f runs beside it -- what a single queue leaves idle in the tail of every
	// launch (mean wave residency 0.93 at 4K, 0.65 on a strip of an 8-way tiling).  For that a frame has its own
	// pre-blur plane (by parity), its own set of work-queue counters (four sets, pwn_i_launch_trace) and its own
	// copy of the tables if they changed (PWN_NBLOB).  (Blur and sink of frame i on a stream of their own beside the
	// trace of frame i+1 -- the kernels of ONE frame split over two queues -- measured nothing: 0.4433 against
	// 0.4429 ms per 4K frame.)  Every event between two kernels costs a few microseconds of pipeline, so only the
	// ones somebody reads are recorded.
	const bool counted = c->counters_on || c->wave_log_on;           // one set of counters: no second grid beside a counted one
	const bool overlap = c->frame_overlap && c->nslots >= 2 && !counted && c->blur_passes <= 1 && (c->d_pre2 != NULL || c->blur_passes == 0);
	const int par = overlap ? (int)(c->frame_seq & 1u) : 0;
	hipStream_t s = par ? c->stream2 : c->stream;
	uint32_t *pre = par ? c->d_pre2 : c->d_pre;
	// timing events: on every frame_timing-th frame (each event between two kernels is a few microseconds of
	// pipeline: 0.428 against 0.417 ms per 4K frame with all frames timed).  With two compute streams the
	// durations are those of kernels that share the chip with the neighbour frames' kernels: a host that wants
	// the duration of a launch by itself times frames with PWN_OPT_FRAME_OVERLAP 0 (bench.py's roofline leg).
	// (Making a timed frame run alone -- its stream waits for the frame before it, the next frame's for it -- was
	// tried: with the per-frame table upload in the picture the two waits between the streams cost 50 us per
	// frame at 4K, 0.4335 against 0.3629 ms.)
	const bool timing = c->frame_timing > 0 && (c->frame_seq % (uint64_t)c->frame_timing) == 0;
	// a frame on the other stream than the one before it is ordered behind that one only where the streams are
	// not meant to run side by side
	if(c->last_frame_done != NULL && c->last_frame_stream != s && !overlap) HIPCHK(c, hipStreamWaitEvent(s, c->last_frame_done, 0));
	if(timing) HIPCHK(c, hipEventRecord(sl.ev_k[0], s));
	// the last pass writes into the slot's own plane, which is what the copy stream reads while
	// the next frame's kernels reuse the context's d_pre / d_out
	uint32_t *cur = c->blur_passes > 0 ? pre : sl.d_out;
	c->trace_tables_event = sl.ev_k[2];        // recorded below, behind the frame's last kernel
	c->launch_room = overlap ? pwn_room_for_launch(c) : 0;
	int rc = pwn_i_launch_trace(c, cam, sec, 0, c->h, cur, sl.d_z, s);
	if(rc != PWN_OK) return rc;
	if(timing) HIPCHK(c, hipEventRecord(sl.ev_k[1], s));
	for(int p = 0; p < c->blur_passes; p++)
	{
		uint32_t *dst = (p == c->blur_passes - 1) ? sl.d_out : (cur == c->d_pre ? c->d_out : c->d_pre);      // (several passes: one stream)
		rc = pwn_i_launch_blur(c, 0, c->h, cur, sl.d_z, dst, s, 0, 0, NULL, NULL, NULL);
		if(rc != PWN_OK) { (void)hipEventRecord(sl.ev_k[2], s); return rc; }     // (the trace launch counts on this event)
		cur = dst;
	}
	HIPCHK(c, hipEventRecord(sl.ev_k[2], s));
	c->last_frame_done = sl.ev_k[2]; c->last_frame_stream = s;
	hipEvent_t last = sl.ev_k[2];
	if(!(c->frame_flags & PWN_FRAME_SURFACE)) { rc = pwn_i_launch_order(c, s); if(rc != PWN_OK) return rc; }      // (behind the frame's "kernels done": nobody waits for it)
	if(c->frame_flags & PWN_FRAME_SURFACE)
	{
		HIPCHK(c, pwn_launch_upscale(sl.d_out, sl.d_surface, c->w, c->h, c->frame_scale, c->frame_pitch / 4, s));
		HIPCHK(c, hipEventRecord(sl.ev_k[3], s));
		last = sl.ev_k[3];
		rc = pwn_i_launch_order(c, s);
		if(rc != PWN_OK) return rc;
	}
	if(c->frame_flags != 0)
	{
		HIPCHK(c, hipStreamWaitEvent(c->copy_stream, last, 0));
		if(c->frame_flags & PWN_FRAME_SBUF) HIPCHK(c, hipMemcpyAsync(sl.h_sbuf, sl.d_out, n * 4, hipMemcpyDeviceToHost, c->copy_stream));
		if(c->frame_flags & PWN_FRAME_ZBUF) HIPCHK(c, hipMemcpyAsync(sl.h_zbuf, sl.d_z, n * 4, hipMemcpyDeviceToHost, c->copy_stream));
		if(c->frame_flags & PWN_FRAME_SURFACE)
			HIPCHK(c, hipMemcpyAsync(sl.h_surface, sl.d_surface, (size_t)c->frame_pitch * (size_t)c->h * (size_t)c->frame_scale,
				hipMemcpyDeviceToHost, c->copy_stream));
		HIPCHK(c, hipEventRecord(sl.ev_done, c->copy_stream));
	}
	sl.in_flight = true; sl.sec = sec; sl.seq = ++c->frame_seq; sl.timed = timing; sl.beside = overlap;
	return PWN_OK;
}

extern "C" int pwn_read_plane(pwn_ctx *c, const void *d_src, void *dst, size_t bytes)
{
	if(GRP_HEAD(c)) return pwn_group_on_member0(c, [d_src, dst, bytes](pwn_ctx *m0) { return pwn_read_plane(m0, d_src, dst, bytes); });       // (a frame that stayed on the devices was gathered on member 0's)
	if(c == NULL || d_src == NULL || dst == NULL) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	HIPCHK(c, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
	return PWN_OK;
}

extern "C" int pwn_frame_ready(pwn_ctx *c, int slot)
{
	if(GRP_HEAD(c)) return pwn_group_frame_ready(c, slot);
	if(c == NULL || slot < 0 || slot >= c->nslots) return PWN_EINVAL;
	if(!c->slot[slot].in_flight) return 1;
	(void)hipSetDevice(c->device);
	hipError_t e = hipEventQuery(c->frame_flags != 0 ? c->slot[slot].ev_done : c->slot[slot].ev_k[2]);
	if(e == hipSuccess) return 1;
	if(e == hipErrorNotReady) return 0;
	snprintf(c->err, sizeof(c->err), "hipEventQuery: %s", hipGetErrorString(e));
	return PWN_EHIP;
}

extern "C" int pwn_wait_frame(pwn_ctx *c, int slot, pwn_frame *out)
{
	if(GRP_HEAD(c)) return pwn_group_wait_frame(c, slot, out);
	if(c == NULL || slot < 0 || slot >= c->nslots) return PWN_EINVAL;
	pwn_slot &sl = c->slot[slot];
	if(sl.seq == 0) return PWN_EINVAL;            // nothing was ever submitted here
	(void)hipSetDevice(c->device);
	if(sl.in_flight)
	{
		HIPCHK(c, hipEventSynchronize(c->frame_flags != 0 ? sl.ev_done : sl.ev_k[2]));
		sl.in_flight = false;
		if(sl.beside) pwn_room_frame_done(c);          // PWN_OPT_TRACE_ROOM: one more delivered frame of a two-stream sequence
	}
	if(out != NULL)
	{
		memset(out, 0, sizeof(*out));
		out->sbuf = sl.h_sbuf; out->zbuf = sl.h_zbuf; out->surface = sl.h_surface;
		out->surface_pitch_bytes = c->frame_pitch;
		out->d_sbuf = sl.d_out; out->d_zbuf = sl.d_z; out->d_surface = sl.d_surface;
		out->sec_current = sl.sec; out->seq = sl.seq;
		if(sl.timed)
		{
			(void)hipEventElapsedTime(&out->trace_ms, sl.ev_k[0], sl.ev_k[1]);
			(void)hipEventElapsedTime(&out->blur_ms, sl.ev_k[1], sl.ev_k[2]);
			if(c->frame_flags & PWN_FRAME_SURFACE) (void)hipEventElapsedTime(&out->sink_ms, sl.ev_k[2], sl.ev_k[3]);
			c->stats.trace_ms = out->trace_ms; c->stats.blur_ms = out->blur_ms;
		}
		out->timed = sl.timed ? 1 : 0;
	}
	return PWN_OK;
}

extern "C" int pwn_get_stats(pwn_ctx *c, pwn_stats *out)
{
	if(GRP_HEAD(c)) return out == NULL ? PWN_EINVAL : pwn_group_get_stats(c, out);
	if(c == NULL || out == NULL) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	if(c->counters_on)
	{
		unsigned long long v[PWN_NCOUNTERS];
		HIPCHK(c, hipMemcpy(v, c->d_counters, sizeof(v), hipMemcpyDeviceToHost));
		for(int i = 0; i < 32; i++) c->stats.regions[i] = v[16 + i];
		c->stats.rays = v[0]; c->stats.steps = v[1]; c->stats.portals = v[2];
		c->stats.sphere_tests = v[3]; c->stats.exhausted = v[4]; c->stats.wave_steps = v[5];
		for(int i = 0; i < 8; i++) c->stats.wave_paths[i] = v[6 + i];
		c->stats.phase_passes = v[14]; c->stats.phase_lanes = v[15];
	}
	if(c->wave_log_on && c->d_wave_log != NULL)
	{
		std::vector<unsigned long long> log(c->wave_log_cap * 2);
		HIPCHK(c, hipMemcpy(log.data(), c->d_wave_log, log.size() * 8, hipMemcpyDeviceToHost));
		unsigned long long sum = 0, first = ~0ull, last = 0, n = 0;
		for(size_t i = 2; i + 1 < log.size(); i += 2)
		{
			if(log[i + 1] == 0) continue;
			sum += log[i + 1] - log[i]; n++;
			if(log[i] < first) first = log[i];
			if(log[i + 1] > last) last = log[i + 1];
		}
		c->stats.wave_time = sum; c->stats.waves = n; c->stats.kernel_span = n ? last - first : 0;
		if(const char *path = getenv("PWN_DBG_WAVE_LOG"))        // tools/wave_log.py: the raw log
			if(FILE *fp = fopen(path, "wb")) { fwrite(log.data(), 8, log.size(), fp); fclose(fp); }
		if(const char *path = getenv("PWN_DBG_UNIT_COST"))       // tools/unit_order_sim.py: u16 per unit (arithmetic numbering), 40 ns each
			if(c->order[0].d_cost != NULL && c->order[0].units > 0)
			{
				std::vector<uint16_t> uc(c->order[0].units);
				HIPCHK(c, hipMemcpy(uc.data(), c->order[0].d_cost, uc.size() * 2, hipMemcpyDeviceToHost));
				if(FILE *fp = fopen(path, "wb")) { fwrite(uc.data(), 2, uc.size(), fp); fclose(fp); }
			}
	}
	*out = c->stats;
	return PWN_OK;
}

// ---- sink -------------------------------------------------------------------

extern "C" int pwn_upscale_device(pwn_ctx *c, const void *d_src, int scale, int pitch_bytes, void *d_dst, void *stream)
{
	GRP_REFUSE(c, "pwn_upscale_device");
	if(c == NULL || d_src == NULL || d_dst == NULL || scale <= 0 || (pitch_bytes & 3) != 0 ||
	   (long long)pitch_bytes < (long long)c->w * scale * 4) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	HIPCHK(c, pwn_launch_upscale((const uint32_t *)d_src, (uint32_t *)d_dst, c->w, c->h, scale, pitch_bytes / 4, (hipStream_t)stream));
	return PWN_OK;
}

extern "C" int pwn_screen_upscale(pwn_ctx *c, const uint32_t *sbuf, int scale, int pitch_bytes, uint32_t *pixels)
{
	if(GRP_HEAD(c)) return pwn_group_screen_upscale(c, sbuf, scale, pitch_bytes, pixels);
	if(c == NULL || pixels == NULL || scale <= 0 || (pitch_bytes & 3) != 0 ||
	   (long long)pitch_bytes < (long long)c->w * scale * 4) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	size_t n = (size_t)c->w * (size_t)c->h;
	size_t dst_bytes = (size_t)pitch_bytes * (size_t)c->h * (size_t)scale;
	int rc = ensure_scratch(c, dst_bytes + n * 4);
	if(rc != PWN_OK) return rc;
	uint32_t *d_dst = c->d_scratch;
	const uint32_t *d_src = c->d_out;
	if(sbuf != NULL)
	{
		uint32_t *d_in = (uint32_t *)((uint8_t *)c->d_scratch + dst_bytes);
		HIPCHK(c, hipMemcpyAsync(d_in, sbuf, n * 4, hipMemcpyHostToDevice, c->stream));
		d_src = d_in;
	}
	// With pitch == w*scale*4 every byte of the surface is written.  Otherwise
	// the reference leaves gaps untouched (and packs rows, see the kernel):
	// start from the caller's surface so that untouched bytes survive.
	bool padded = (long long)pitch_bytes != (long long)c->w * scale * 4;
	if(padded) HIPCHK(c, hipMemcpyAsync(d_dst, pixels, dst_bytes, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, pwn_launch_upscale(d_src, d_dst, c->w, c->h, scale, pitch_bytes / 4, c->stream));
	HIPCHK(c, hipMemcpyAsync(pixels, d_dst, dst_bytes, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return PWN_OK;
}

// ---- probes -----------------------------------------------------------------

extern "C" int pwn_probe(pwn_ctx *c, int op, const uint32_t *in, uint32_t *out, int n)
{
	if(GRP_HEAD(c)) return pwn_group_on_member0(c, [op, in, out, n](pwn_ctx *m0) { return pwn_probe(m0, op, in, out, n); });
	if(c == NULL || in == NULL || out == NULL || n < 0 || op < 0 || op > PWN_PROBE_COS_OF_PAIR) return PWN_EINVAL;
	if(n == 0) return PWN_OK;
	(void)hipSetDevice(c->device);
	size_t per = (op == PWN_PROBE_DIV) ? 2 : (op == PWN_PROBE_FTOINT ? 4 : 1);
	size_t in_bytes = (size_t)n * per * 4, out_bytes = (size_t)n * 4;
	int rc = ensure_scratch(c, in_bytes + out_bytes + 8192);
	if(rc != PWN_OK) return rc;
	uint8_t *base = (uint8_t *)c->d_scratch;
	uint16_t *d_tabs = (uint16_t *)base;
	uint32_t *d_in = (uint32_t *)(base + 8192), *d_out = (uint32_t *)(base + 8192 + in_bytes);
	HIPCHK(c, hipMemcpy(d_tabs, c->tabs, 8192, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, pwn_launch_probe(op, d_in, d_out, n, d_tabs, c->stream));
	HIPCHK(c, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return PWN_OK;
}
