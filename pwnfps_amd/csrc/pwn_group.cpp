// pwn_group.cpp -- several GPUs of ONE process behind one handle (pwn_init_multi, include/pwnhip.h).
//
// The reference's host is one process with one loop (main.c:93-109); that its frame is computed by many cores is the
// business of two OpenMP pragmas inside trace_screen_centred (screen.h:63-67,77).  A group is the same for GPUs: the host
// keeps its one loop and its one handle, and every call of include/pwnhip.h on that handle fans out here --
//   * level and object calls act on member 0's tables (the handle's game state exists once); what a frame needs on the
//     devices -- the packed tables -- is made per member from that one state (pwn_prepare_render: one list of live
//     spheres, binned and uploaded by every member's thread on its own device);
//   * a frame is the row tiling of pwn_tiled.cpp, unchanged: every member is a rank with a full-size context on its
//     device, the members' threads make the same pwn_tiled_submit / _wait calls a process per GPU would make, and the
//     exchange between them is RCCL (one communicator rank per device, brought up in-process under the deadline) or, for
//     members that share a device and on request, PWN_TRANSPORT_LOCAL.  The frame goes to the host over N PCIe links at once:
//     every member copies its strip into the caller's sbuf (pwn_trace_screen_centred) or into the group's pinned frames
//     (frames in flight); with nothing to deliver the strips are gathered on member 0's device.
// One library-owned thread per member beyond the first (the caller's thread drives member 0): a single thread enqueuing
// for eight GPUs would spend 8 x 30 us per frame where a device needs 50 us for its strip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <new>
#include <atomic>
#include <array>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "pwnhip.h"
#include "pwn_internal.h"

#define MAXM PWN_TILED_MAX_WORLD
#define GROUP_RING 32u
enum { MODE_NONE = 0, MODE_SINK = 1, MODE_RESIDENT = 2 };

struct pwn_group
{
	int n, transport;
	bool transport_fixed;                        // PWN_GROUP_TRANSPORT named it: no falling back
	char note[160];                              // why RCCL is not what carries the rows (pwn_group_info.note), or empty
	pwn_ctx *head, *m[MAXM];
	int devices[MAXM];
	pwn_hub hub;
	// The members' threads, one per member.  A job is a function of the member's index; the caller posts it for all members and
	// either goes on (the tables of the next frame, a frame's submission: the members run ahead of the host loop like the ranks
	// of a process-per-GPU run do) or waits for it (a frame's delivery, everything rare).  Jobs run in the order they were
	// posted, on every member.  The first failure marks the group broken: the members skip what is queued behind it, and the
	// next call that waits reports it.
	std::thread th[MAXM];
	struct job_t { std::function<int(int)> fn; bool always; } jobs[GROUP_RING];
	std::atomic<unsigned long long> posted;                 // jobs 1..posted exist (job k in jobs[k % GROUP_RING])
	std::atomic<unsigned long long> done[MAXM];             // ... and member i is through jobs 1..done[i]
	std::atomic<bool> quit, broken;
	std::mutex err_mu; int err_rc, err_member; char err_text[200];
	// PWN_DBG_GROUP_PROF: where the time goes (printed when the group goes): how long the caller waits per job it waits for, how
	// long the members take per job
	bool prof; double p_join, p_job[MAXM]; unsigned long long p_jobs, p_joins;
	// the tiling in force
	int mode;
	int init_ms, wait_ms;
	// frames in flight
	int nslots, flags;
	uint32_t *h_sbuf[PWN_MAX_SLOTS]; float *h_zbuf[PWN_MAX_SLOTS]; uint32_t *h_surface[PWN_MAX_SLOTS];
	int surf_scale, surf_pitch;                  // PWN_FRAME_SURFACE as configured; the tiling in force upscales with these (tiling_surf_*)
	int tiling_surf_scale, tiling_surf_pitch;
	bool in_flight[PWN_MAX_SLOTS], delivered[PWN_MAX_SLOTS];
	pwn_frame fdone[PWN_MAX_SLOTS];
	float sec[PWN_MAX_SLOTS];
	int fifo[PWN_MAX_SLOTS + 1], fifo_n;        // slots in submission order, oldest first
	uint64_t frame_seq;
	const uint32_t *last_sbuf;                   // where the last blocking call delivered (pwn_screen_upscale(NULL, ...), main.c:108)
	unsigned long long calls; int stall_member, stall_ms; unsigned long long stall_call;      // PWN_DBG_GROUP_STALL (pwn_group_trace_screen_centred)
	pwn_tiled_frame tf[MAXM];
};

static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

// ---- the members' meeting point (pwn_internal.h).  Every member's thread arrives; the last one opens the next round.
int pwn_hub_meet(pwn_hub *h, int wait_ms)
{
	if(h->failed.load(std::memory_order_relaxed)) return PWN_EHIP;
	const unsigned gen = h->bar_gen.load(std::memory_order_acquire);
	if(h->bar_count.fetch_add(1u, std::memory_order_acq_rel) + 1u == (unsigned)h->world)
	{
		h->bar_count.store(0u, std::memory_order_relaxed);
		h->bar_gen.store(gen + 1u, std::memory_order_release);
		return PWN_OK;
	}
	const double t0 = now_ms();
	for(unsigned long long spins = 0; h->bar_gen.load(std::memory_order_acquire) == gen; spins++)
	{
		if(h->failed.load(std::memory_order_relaxed)) return PWN_EHIP;
		if(spins > 4000)
		{
			if(now_ms() - t0 > (double)wait_ms) { h->failed.store(1); return PWN_ETIMEDOUT; }
			if(spins > 40000) { struct timespec ts = { 0, 50 * 1000 }; nanosleep(&ts, NULL); }
		}
	}
	return PWN_OK;
}

// ---- jobs
static void worker(pwn_group *g, int i)
{
	(void)hipSetDevice(g->devices[i]);
	unsigned long long seen = 0;
	for(;;)
	{
		// a frame loop posts jobs back to back: look for the next one for a while (~0.2 ms) before going to sleep
		unsigned long long now = g->posted.load(std::memory_order_acquire);
		for(unsigned spins = 0; now == seen && spins < 100000u && !g->quit.load(std::memory_order_relaxed); spins++) now = g->posted.load(std::memory_order_acquire);
		// (asleep in slices that grow to 0.2 ms: an idle group costs its threads a few thousand wake-ups a second, and a job posted
		// to a sleeping group starts that much late -- once)
		for(long slice = 20 * 1000; now == seen && !g->quit.load(std::memory_order_relaxed); slice = slice < 200 * 1000 ? slice * 2 : slice)
		{
			struct timespec ts = { 0, slice };
			nanosleep(&ts, NULL);
			now = g->posted.load(std::memory_order_acquire);
		}
		if(now == seen) { if(g->quit.load()) return; continue; }
		while(seen < now)
		{
			pwn_group::job_t &j = g->jobs[(seen + 1) % GROUP_RING];
			const double t0 = g->prof ? now_ms() : 0.0;
			if(j.always || !g->broken.load(std::memory_order_acquire))
			{
				const int rc = j.fn(i);
				if(rc != PWN_OK)
				{
					std::lock_guard<std::mutex> lk(g->err_mu);
					if(g->err_rc == PWN_OK)
					{
						g->err_rc = rc; g->err_member = i;
						snprintf(g->err_text, sizeof(g->err_text), "%.190s", g->m[i]->err[0] ? g->m[i]->err : pwn_strerror(rc));
					}
					g->broken.store(true, std::memory_order_release);
					g->hub.failed.store(1);             // (nobody waits for this member any longer)
				}
			}
			if(g->prof) g->p_job[i] += now_ms() - t0;
			seen++;
			g->done[i].store(seen, std::memory_order_release);
		}
	}
}

// post `fn` for every member -- member i runs fn(i) on its own thread, behind everything posted before -- and go on
static unsigned long long post(pwn_group *g, std::function<int(int)> fn, bool always = false)
{
	const unsigned long long id = g->posted.load(std::memory_order_relaxed) + 1;
	// (the slot's last job, GROUP_RING jobs back, has to be through on every member)
	if(id > GROUP_RING)
		for(int i = 0; i < g->n; i++)
			for(unsigned long long spins = 0; g->done[i].load(std::memory_order_acquire) + GROUP_RING < id; spins++)
				if(spins > 100000) { struct timespec ts = { 0, 20 * 1000 }; nanosleep(&ts, NULL); }
	g->jobs[id % GROUP_RING].fn = std::move(fn);
	g->jobs[id % GROUP_RING].always = always;
	g->posted.store(id, std::memory_order_release);
	if(g->prof) g->p_jobs++;
	return id;
}

// wait until every member is through job `id`; the first failure since the group was last in order, if any
static int join(pwn_group *g, unsigned long long id)
{
	const double t0 = g->prof ? now_ms() : 0.0;
	for(int i = 0; i < g->n; i++)
		for(unsigned long long spins = 0; g->done[i].load(std::memory_order_acquire) < id; spins++)
			if(spins > 200000) { struct timespec ts = { 0, 20 * 1000 }; nanosleep(&ts, NULL); }
	if(g->prof) { g->p_join += now_ms() - t0; g->p_joins++; }
	if(g->broken.load(std::memory_order_acquire))
	{
		std::lock_guard<std::mutex> lk(g->err_mu);
		snprintf(g->head->err, sizeof(g->head->err), "member %d (device %d): %s", g->err_member, g->devices[g->err_member], g->err_text);
		return g->err_rc != PWN_OK ? g->err_rc : PWN_EHIP;
	}
	return PWN_OK;
}

static int run_all(pwn_group *g, std::function<int(int)> fn, bool always = false) { return join(g, post(g, std::move(fn), always)); }

// the group is in order again (the caller has dealt with the failure: the tiling is down)
static void mend(pwn_group *g)
{
	std::lock_guard<std::mutex> lk(g->err_mu);
	g->err_rc = PWN_OK; g->err_member = 0; g->err_text[0] = 0;
	g->broken.store(false, std::memory_order_release);
}

pwn_ctx *pwn_group_member(pwn_ctx *h, int i) { return (h != NULL && h->grp != NULL && i >= 0 && i < h->grp->n) ? h->grp->m[i] : NULL; }

static void frames_free(pwn_group *g)
{
	for(int i = 0; i < PWN_MAX_SLOTS; i++)
	{
		if(g->h_sbuf[i]) (void)hipHostFree(g->h_sbuf[i]);
		if(g->h_zbuf[i]) (void)hipHostFree(g->h_zbuf[i]);
		if(g->h_surface[i]) (void)hipHostFree(g->h_surface[i]);
		g->h_sbuf[i] = NULL; g->h_zbuf[i] = NULL; g->h_surface[i] = NULL; g->in_flight[i] = g->delivered[i] = false;
	}
	g->nslots = 0; g->flags = 0; g->fifo_n = 0;
}

// the tiling of the members, in the mode the next frame needs: delivered to the host by every member, or gathered on member 0
static int tiling_down(pwn_group *g)
{
	if(g->mode == MODE_NONE) return PWN_OK;
	(void)run_all(g, [g](int i) { pwn_tiled_shutdown(g->m[i]); return PWN_OK; }, true);
	g->mode = MODE_NONE;
	g->tiling_surf_scale = g->tiling_surf_pitch = 0;
	mend(g);
	g->hub.failed.store(0);
	g->hub.bar_count.store(0);
	for(int i = 0; i < g->n * g->n; i++) { g->hub.boxes[i].posted.store(0); g->hub.boxes[i].copied.store(0); g->hub.boxes[i].consumed.store(0); }
	return PWN_OK;
}

static int tiling_up(pwn_group *g, int mode)
{
	const int sscale = (mode == MODE_SINK && (g->flags & PWN_FRAME_SURFACE)) ? g->surf_scale : 0, spitch = sscale ? g->surf_pitch : 0;
	if(g->mode == mode && (sscale == 0 || (sscale == g->tiling_surf_scale && spitch == g->tiling_surf_pitch))) return PWN_OK;
	tiling_down(g);
	unsigned char id[PWN_TILED_ID_BYTES];
	memset(id, 0, sizeof(id));
	if(g->transport == PWN_TRANSPORT_RCCL)
	{
		const int rc = pwn_tiled_unique_id(id, PWN_TRANSPORT_RCCL);
		if(rc != PWN_OK) { snprintf(g->head->err, sizeof(g->head->err), "librccl: no unique id (%s)", pwn_strerror(rc)); return rc; }
	}
	std::vector<unsigned char> idv(id, id + sizeof(id));
	const int rc = run_all(g, [g, idv, mode, sscale, spitch](int i)
	{
		const unsigned char *id = idv.data();
		pwn_ctx *c = g->m[i];
		int r = pwn_tiled_set_timeouts(c, g->init_ms ? g->init_ms : 0, g->wait_ms ? g->wait_ms : 0);
		if(r == PWN_OK) r = pwn_tiled_init(c, i, g->n, id, g->transport, -1);
		if(r == PWN_OK && mode == MODE_SINK) r = pwn_i_tiled_sink(c);
		if(r == PWN_OK && sscale) r = pwn_i_tiled_surface(c, sscale, spitch);
		return r;
	});
	if(rc != PWN_OK)
	{
		char keep[256];
		memcpy(keep, g->head->err, sizeof(keep));
		g->mode = mode;               // (whatever came up goes down again)
		tiling_down(g);
		memcpy(g->head->err, keep, sizeof(keep));
		// RCCL did not come up between the members (it has never had to, on this pool): the rows can travel as copies between the
		// members' planes instead -- in one process nothing else is needed.  Once, and said in pwn_group_info.note; a host that
		// named the transport (PWN_GROUP_TRANSPORT) gets the error.
		if(g->transport == PWN_TRANSPORT_RCCL && !g->transport_fixed)
		{
			snprintf(g->note, sizeof(g->note), "RCCL did not come up (%s: %.100s): copies between the members instead", pwn_strerror(rc), keep);
			g->transport = PWN_TRANSPORT_LOCAL;
			return tiling_up(g, mode);
		}
		return rc;
	}
	g->mode = mode;
	g->tiling_surf_scale = sscale; g->tiling_surf_pitch = spitch;
	return PWN_OK;
}

// a frame call failed on some member: the tiling is not in a state to go on with (a transport that was aborted, frames half
// enqueued): take it down; the next frame call sets it up afresh
static int frame_failed(pwn_group *g, int rc)
{
	char keep[256];
	memcpy(keep, g->head->err, sizeof(keep));
	tiling_down(g);
	for(int i = 0; i < PWN_MAX_SLOTS; i++) g->in_flight[i] = g->delivered[i] = false;
	g->fifo_n = 0;
	memcpy(g->head->err, keep, sizeof(keep));
	return rc;
}

extern "C" int pwn_init_multi(pwn_ctx **out, const int *devices, int ndev, int width, int height)
{
	if(out == NULL || devices == NULL || ndev < 1 || ndev > MAXM) return PWN_EINVAL;
	*out = NULL;
	if(ndev == 1) return pwn_init(out, devices[0], width, height);
	pwn_group *g = new(std::nothrow) pwn_group();
	pwn_ctx *h = new(std::nothrow) pwn_ctx();
	if(g == NULL || h == NULL) { delete g; delete h; return PWN_ENOMEM; }
	g->n = ndev; g->head = h; g->mode = MODE_NONE; g->posted.store(0); g->quit.store(false); g->broken.store(false);
	g->err_rc = PWN_OK; g->err_member = 0; g->err_text[0] = 0;
	g->calls = 0; g->stall_member = -1; g->stall_ms = 0; g->stall_call = 0;
	if(const char *e = getenv("PWN_DBG_GROUP_STALL")) { int a = -1, c = 0; unsigned long long b = 0; if(sscanf(e, "%d:%llu:%d", &a, &b, &c) == 3) { g->stall_member = a; g->stall_call = b; g->stall_ms = c; } }
	g->prof = getenv("PWN_DBG_GROUP_PROF") != NULL; g->p_join = 0.0; g->p_jobs = g->p_joins = 0;
	for(int i = 0; i < MAXM; i++) { g->p_job[i] = 0.0; g->done[i].store(0); }
	g->init_ms = g->wait_ms = 0; g->nslots = 0; g->flags = 0; g->fifo_n = 0; g->frame_seq = 0; g->last_sbuf = NULL;
	for(int i = 0; i < PWN_MAX_SLOTS; i++) { g->h_sbuf[i] = NULL; g->h_zbuf[i] = NULL; g->h_surface[i] = NULL; g->in_flight[i] = g->delivered[i] = false; }
	g->surf_scale = g->surf_pitch = g->tiling_surf_scale = g->tiling_surf_pitch = 0;
	for(int i = 0; i < MAXM; i++) { g->m[i] = NULL; g->devices[i] = -1; g->hub.member[i] = NULL; }
	g->hub.world = ndev; g->hub.failed.store(0); g->hub.bar_count.store(0); g->hub.bar_gen.store(0); g->hub.boxes = NULL;
	// the handle: no device of its own, the frame's size and the group
	h->device = devices[0]; h->w = width; h->h = height; h->grp = g; h->grp_head = true; h->hub = NULL; h->tiled = NULL;
	h->blur_passes = 1; h->err[0] = 0; h->nslots = 0; h->frame_flags = 0;
	memset(&h->stats, 0, sizeof(h->stats));
	int rc = PWN_OK;
	g->hub.boxes = new(std::nothrow) pwn_hub_box[(size_t)ndev * (size_t)ndev];
	if(g->hub.boxes == NULL) rc = PWN_ENOMEM;
	for(int i = 0; i < ndev * ndev && rc == PWN_OK; i++) { g->hub.boxes[i].posted.store(0); g->hub.boxes[i].copied.store(0); g->hub.boxes[i].consumed.store(0); }
	bool distinct = true;
	for(int i = 0; i < ndev && rc == PWN_OK; i++)
	{
		g->devices[i] = devices[i];
		for(int k = 0; k < i; k++) if(devices[k] == devices[i]) distinct = false;
		rc = pwn_init(&g->m[i], devices[i], width, height);
		if(rc != PWN_OK) { snprintf(h->err, sizeof(h->err), "member %d: pwn_init on device %d: %s", i, devices[i], pwn_strerror(rc)); break; }
		g->m[i]->grp = g; g->m[i]->grp_head = false; g->m[i]->hub = &g->hub;
		g->hub.member[i] = g->m[i];
	}
	if(rc != PWN_OK)
	{
		for(int i = 0; i < ndev; i++) if(g->m[i]) { g->m[i]->grp = NULL; g->m[i]->hub = NULL; pwn_destroy(g->m[i]); }
		delete[] g->hub.boxes; delete g; delete h;
		return rc;
	}
	// the exchange between the members: RCCL where every member has a device of its own and the library loads, else copies
	// between the members' planes behind events; PWN_GROUP_TRANSPORT=local|rccl in the environment says which
	g->transport = PWN_TRANSPORT_LOCAL;
	g->note[0] = 0;
	const char *want = getenv("PWN_GROUP_TRANSPORT");
	g->transport_fixed = want != NULL && (strcmp(want, "local") == 0 || strcmp(want, "rccl") == 0);
	// (PWN_DBG_GROUP_TRY_RCCL: RCCL is tried although two members share a device -- its bring-up then fails, which is how the tests
	// see the fall-back without a second GPU)
	const bool try_rccl = distinct || getenv("PWN_DBG_GROUP_TRY_RCCL") != NULL;
	if(try_rccl && !(want != NULL && strcmp(want, "local") == 0))
	{
		unsigned char id[PWN_TILED_ID_BYTES];
		if(pwn_tiled_unique_id(id, PWN_TRANSPORT_RCCL) == PWN_OK) g->transport = PWN_TRANSPORT_RCCL;
		else snprintf(g->note, sizeof(g->note), "librccl did not load: copies between the members instead");
	}
	else if(!distinct) snprintf(g->note, sizeof(g->note), "members share a device: copies between the members");
	// (peer-to-peer copies between two devices go over xGMI directly where each may address the other's memory)
	for(int a = 0; a < ndev && distinct; a++)
		for(int b = 0; b < ndev; b++)
		{
			if(a == b) continue;
			int can = 0;
			if(hipDeviceCanAccessPeer(&can, devices[a], devices[b]) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
			(void)hipSetDevice(devices[a]);
			if(hipDeviceEnablePeerAccess(devices[b], 0) != hipSuccess) (void)hipGetLastError();          // (already enabled: fine)
		}
	for(int i = 0; i < ndev; i++) g->th[i] = std::thread(worker, g, i);
	*out = h;
	return PWN_OK;
}

void pwn_group_destroy(pwn_ctx *h)
{
	pwn_group *g = h->grp;
	tiling_down(g);
	(void)join(g, g->posted.load());
	g->quit.store(true);
	for(int i = 0; i < g->n; i++) if(g->th[i].joinable()) g->th[i].join();
	frames_free(g);
	if(g->prof && g->p_jobs)
	{
		fprintf(stderr, "group of %d: %llu jobs, %llu of them waited for: %.1f us per wait;", g->n, g->p_jobs, g->p_joins, g->p_joins ? g->p_join / g->p_joins * 1e3 : 0.0);
		for(int i = 0; i < g->n; i++) fprintf(stderr, " member %d %.1f us per job;", i, g->p_job[i] / g->p_jobs * 1e3);
		fprintf(stderr, "\n");
	}
	for(int i = 0; i < g->n; i++) if(g->m[i]) { g->m[i]->grp = NULL; g->m[i]->hub = NULL; pwn_destroy(g->m[i]); }
	delete[] g->hub.boxes;
	delete g;
	h->grp = NULL;
	delete h;
}

extern "C" int pwn_group_info_get(pwn_ctx *h, pwn_group_info *out)
{
	if(h == NULL || out == NULL || h->grp == NULL || !h->grp_head) return PWN_EINVAL;
	pwn_group *g = h->grp;
	memset(out, 0, sizeof(*out));
	out->members = g->n; out->transport = g->transport;
	snprintf(out->note, sizeof(out->note), "%s", g->note);
	for(int i = 0; i < g->n; i++) out->devices[i] = g->devices[i];
	pwn_tiled_info ti;
	(void)join(g, g->posted.load());          // (the members' threads are through what was posted: their tilings are at rest)
	if(g->mode != MODE_NONE && pwn_tiled_get_info(g->m[0], &ti) == PWN_OK)
	{
		(void)pwn_tiled_get_cuts(g->m[0], out->cuts, NULL);
		out->halo_rows = ti.halo_rows; out->host_sink = ti.host_sink;
		out->frames = ti.frames; out->frames_redone = ti.frames_redone; out->recuts = ti.recuts;
	}
	return PWN_OK;
}

// ---- options, level, objects ---------------------------------------------------
int pwn_group_set_option(pwn_ctx *h, int option, int value)
{
	pwn_group *g = h->grp;
	for(int i = 0; i < PWN_MAX_SLOTS; i++) if(g->in_flight[i]) return PWN_EBUSY;
	// (what a tiling fixes when it is set up: it goes down and comes up again with the next frame)
	if(option == PWN_OPT_BLUR_PASSES || option == PWN_OPT_FRAME_OVERLAP || option == PWN_OPT_TILED_CHOREO || option == PWN_OPT_TILED_COMMS ||
	   option == PWN_OPT_TILED_STREAMS) tiling_down(g);
	if(option == PWN_OPT_BLUR_PASSES && value > 1) { snprintf(h->err, sizeof(h->err), "a group renders with POSTPROC_BLUR 0 or 1"); return PWN_EINVAL; }
	const int rc = run_all(g, [g, option, value](int i) { return pwn_set_option(g->m[i], option, value); }, true);
	if(rc != PWN_OK) { mend(g); return rc; }        // (a refused option is no failure of the group)
	if(option == PWN_OPT_BLUR_PASSES) h->blur_passes = value;
	return PWN_OK;
}

int pwn_group_level_mem(pwn_ctx *h, const char *text, int len)
{
	pwn_group *g = h->grp;
	return run_all(g, [g, text, len](int i) { return pwn_level_load_mem(g->m[i], text, len); });
}

int pwn_group_upload_level(pwn_ctx *h, const uint8_t *data, const pwn_portal *pmap)
{
	pwn_group *g = h->grp;
	return run_all(g, [g, data, pmap](int i) { return pwn_upload_level(g->m[i], data, pmap); });
}

// level_prepare_render (level.h:64-81) for every device: the handle's ONE object table (member 0's, touched by the caller's thread
// only) gives the list of live spheres; it is binned once, here, and every member's thread packs and uploads the lists on its own
// device -- behind whatever it is still doing: the call does not wait
static int post_spheres(pwn_group *g, const pwn_sphere *s, int n)
{
	std::shared_ptr<pwn_binned> b = std::make_shared<pwn_binned>();
	const int rc = pwn_i_bin_spheres(s, n, b.get());
	if(rc != PWN_OK) { snprintf(g->head->err, sizeof(g->head->err), "%s", pwn_strerror(rc)); return rc; }
	(void)post(g, [g, b](int i) { return pwn_i_upload_binned(g->m[i], *b); });
	return PWN_OK;
}

int pwn_group_upload_spheres(pwn_ctx *h, const pwn_sphere *s, int n)
{
	pwn_group *g = h->grp;
	const int rc = post_spheres(g, s, n);
	if(rc == PWN_OK) pwn_i_set_object_table(g->m[0], s, n);
	return rc;
}

int pwn_group_prepare_render(pwn_ctx *h)
{
	pwn_group *g = h->grp;
	const int n = pwn_get_objects(g->m[0], NULL, 0);
	if(n < 0) { snprintf(h->err, sizeof(h->err), "%s", g->m[0]->err); return n; }
	std::vector<pwn_sphere> live((size_t)(n > 0 ? n : 1));
	const int got = pwn_get_objects(g->m[0], live.data(), n);
	if(got < 0) return got;
	return post_spheres(g, live.data(), n);
}

// everything posted so far is through (what reads a member's tables from the caller's thread comes behind this)
int pwn_group_sync(pwn_ctx *h) { return join(h->grp, h->grp->posted.load()); }

// a call about member 0's device (the sink, a plane, the probes), on member 0's thread
int pwn_group_on_member0(pwn_ctx *h, const std::function<int(pwn_ctx *)> &fn)
{
	pwn_group *g = h->grp;
	const int rc = run_all(g, [g, &fn](int i) { return i == 0 ? fn(g->m[0]) : PWN_OK; }, true);
	if(rc != PWN_OK && !g->broken.load()) snprintf(h->err, sizeof(h->err), "%s", g->m[0]->err);
	return rc;
}

int pwn_group_host_register(pwn_ctx *h, void *base, size_t bytes)
{
	const int rc = pwn_group_on_member0(h, [base, bytes](pwn_ctx *m0) { return pwn_host_register(m0, base, bytes); });
	if(rc != PWN_OK) mend(h->grp);
	return rc;
}
int pwn_group_host_unregister(pwn_ctx *h, void *base)
{
	const int rc = pwn_group_on_member0(h, [base](pwn_ctx *m0) { return pwn_host_unregister(m0, base); });
	if(rc != PWN_OK) mend(h->grp);
	return rc;
}

int pwn_group_set_timeouts(pwn_ctx *h, int init_ms, int wait_ms)
{
	pwn_group *g = h->grp;
	if(init_ms != 0) g->init_ms = init_ms > 0 ? init_ms : 0;
	if(wait_ms != 0) g->wait_ms = wait_ms > 0 ? wait_ms : 0;
	(void)run_all(g, [g, init_ms, wait_ms](int i) { return pwn_tiled_set_timeouts(g->m[i], init_ms, wait_ms); }, true);
	return PWN_OK;
}

// ---- frames --------------------------------------------------------------------
static void note_times(pwn_group *g, pwn_frame *f)
{
	// the slowest member's kernels (every member times its own strip)
	f->trace_ms = f->blur_ms = 0.0f; f->timed = 0;
	for(int i = 0; i < g->n; i++)
	{
		if(!g->tf[i].timed) continue;
		f->timed = 1;
		if(g->tf[i].trace_ms > f->trace_ms) f->trace_ms = g->tf[i].trace_ms;
		if(g->tf[i].blur_ms > f->blur_ms) f->blur_ms = g->tf[i].blur_ms;
	}
}

// trace_screen_centred (screen.h:31-124, main.c:107) over the group: every member traces its strip, exchanges the border rows,
// blurs, and copies strip and depth strip straight into the caller's buffers; the call returns when every strip has landed
int pwn_group_trace_screen_centred(pwn_ctx *h, const float cam[16], float sec, uint32_t *sbuf, float *zbuf)
{
	pwn_group *g = h->grp;
	if(cam == NULL || sbuf == NULL) return PWN_EINVAL;
	if(h->blur_passes > 0 && (h->w & 3) != 0) return PWN_EINVAL;
	for(int i = 0; i < PWN_MAX_SLOTS; i++) if(g->in_flight[i]) { snprintf(h->err, sizeof(h->err), "a group's blocking call needs its frames in flight waited for first"); return PWN_EBUSY; }
	const double t0 = now_ms();
	int rc = tiling_up(g, MODE_SINK);
	if(rc != PWN_OK) return rc;
	std::array<float, 16> cm;
	memcpy(cm.data(), cam, sizeof(float) * 16);
	const unsigned long long call = ++g->calls;
	rc = run_all(g, [g, cm, sec, sbuf, zbuf, call](int i)
	{
		// test hook (tests/test_gpu_group.py): PWN_DBG_GROUP_STALL=member:call:milliseconds -- that member's thread is late to that
		// blocking call by so long (a device that stops answering): the others must come back with PWN_ETIMEDOUT at the deadline
		if(g->stall_member == i && g->stall_call == call) { struct timespec ts = { g->stall_ms / 1000, (long)(g->stall_ms % 1000) * 1000000L }; nanosleep(&ts, NULL); }
		// (depth plane PWN_MAX_SLOTS: the blocking calls' own -- every call's depth lives in the same plane, so that a pixel whose
		// primary ray runs out of steps keeps the previous call's value, trace.h:677, as the context's plane does on one device)
		int r = pwn_i_tiled_submit(g->m[i], cm.data(), sec, sbuf, zbuf, PWN_MAX_SLOTS, NULL);
		if(r == PWN_OK) r = pwn_tiled_wait(g->m[i], 0, &g->tf[i]);
		return r;
	});
	if(rc != PWN_OK) return frame_failed(g, rc);
	pwn_frame f;
	memset(&f, 0, sizeof(f));
	note_times(g, &f);
	h->stats.trace_ms = f.trace_ms; h->stats.blur_ms = f.blur_ms; h->stats.total_ms = (float)(now_ms() - t0);
	g->last_sbuf = sbuf;
	return PWN_OK;
}

int pwn_group_frames_config(pwn_ctx *h, int nslots, int flags, int scale, int pitch_bytes)
{
	pwn_group *g = h->grp;
	if(nslots < 0 || nslots > PWN_MAX_SLOTS || (flags & ~(PWN_FRAME_SBUF | PWN_FRAME_ZBUF | PWN_FRAME_SURFACE)) != 0) return PWN_EINVAL;
	if(flags & PWN_FRAME_SURFACE)
	{
		if(scale <= 0) return PWN_EINVAL;
		if(pitch_bytes == 0) pitch_bytes = h->w * scale * 4;
		if((pitch_bytes & 3) != 0 || (long long)pitch_bytes < (long long)h->w * scale * 4) return PWN_EINVAL;
	}
	else { scale = 1; pitch_bytes = 0; }
	for(int i = 0; i < PWN_MAX_SLOTS; i++) if(g->in_flight[i]) return PWN_EBUSY;
	frames_free(g);
	const size_t bytes = (size_t)h->w * (size_t)h->h * 4, surf_bytes = (size_t)pitch_bytes * (size_t)h->h * (size_t)scale;
	(void)hipSetDevice(g->devices[0]);
	for(int i = 0; i < nslots; i++)
	{
		// (portable: every member's device copies into them.  Whatever is delivered, the members' strips land in a host frame)
		if(flags != 0 && hipHostMalloc((void **)&g->h_sbuf[i], bytes, hipHostMallocPortable) != hipSuccess) { frames_free(g); return PWN_ENOMEM; }
		if((flags & PWN_FRAME_ZBUF) && hipHostMalloc((void **)&g->h_zbuf[i], bytes, hipHostMallocPortable) != hipSuccess) { frames_free(g); return PWN_ENOMEM; }
		if(flags & PWN_FRAME_SURFACE)
		{
			if(hipHostMalloc((void **)&g->h_surface[i], surf_bytes, hipHostMallocPortable) != hipSuccess) { frames_free(g); return PWN_ENOMEM; }
			memset(g->h_surface[i], 0, surf_bytes);           // bytes between the rows of a wider pitch read 0
		}
	}
	g->nslots = nslots; g->flags = flags; g->surf_scale = (flags & PWN_FRAME_SURFACE) ? scale : 0; g->surf_pitch = (flags & PWN_FRAME_SURFACE) ? pitch_bytes : 0;
	h->nslots = nslots; h->frame_flags = flags; h->frame_pitch = pitch_bytes; h->frame_scale = scale;
	return PWN_OK;
}

int pwn_group_submit_frame(pwn_ctx *h, const float cam[16], float sec, int slot)
{
	pwn_group *g = h->grp;
	if(cam == NULL || slot < 0 || slot >= g->nslots) return PWN_EINVAL;
	if(h->blur_passes > 0 && (h->w & 3) != 0) return PWN_EINVAL;
	if(g->in_flight[slot]) return PWN_EBUSY;
	const bool sink = g->flags != 0;
	int rc = tiling_up(g, sink ? MODE_SINK : MODE_RESIDENT);
	if(rc != PWN_OK) return rc;
	uint32_t *hs = sink ? g->h_sbuf[slot] : NULL;
	float *hz = (g->flags & PWN_FRAME_ZBUF) ? g->h_zbuf[slot] : NULL;
	uint32_t *hsurf = (g->flags & PWN_FRAME_SURFACE) ? g->h_surface[slot] : NULL;
	// (posted, not waited for: the members enqueue the frame while the host goes on; a failure shows at the frame's pwn_wait_frame)
	std::array<float, 16> cm;
	memcpy(cm.data(), cam, sizeof(float) * 16);
	// (depth plane `slot`: a frame slot's depth carries from the slot's previous frame, as a one-device context's slot planes do)
	(void)post(g, [g, cm, sec, hs, hz, slot, hsurf](int i) { return pwn_i_tiled_submit(g->m[i], cm.data(), sec, hs, hz, slot, hsurf); });
	g->in_flight[slot] = true; g->delivered[slot] = false; g->sec[slot] = sec;
	g->fifo[g->fifo_n++] = slot;
	memset(&g->fdone[slot], 0, sizeof(pwn_frame));
	g->fdone[slot].seq = ++g->frame_seq;
	return PWN_OK;
}

// the oldest frame in flight, on every member
static int deliver_oldest(pwn_group *g)
{
	const int slot = g->fifo[0];
	const int rc = run_all(g, [g](int i) { return pwn_tiled_wait(g->m[i], 0, &g->tf[i]); });
	if(rc != PWN_OK) return frame_failed(g, rc);
	for(int i = 1; i < g->fifo_n; i++) g->fifo[i - 1] = g->fifo[i];
	g->fifo_n--;
	pwn_frame &f = g->fdone[slot];
	note_times(g, &f);
	f.sec_current = g->sec[slot];
	f.sbuf = (g->flags & PWN_FRAME_SBUF) ? g->h_sbuf[slot] : NULL;
	f.zbuf = (g->flags & PWN_FRAME_ZBUF) ? g->h_zbuf[slot] : NULL;
	f.surface = (g->flags & PWN_FRAME_SURFACE) ? g->h_surface[slot] : NULL;
	f.surface_pitch_bytes = g->surf_pitch;
	f.d_sbuf = g->tf[0].d_sbuf;               // nothing delivered: the frame as gathered on member 0's device
	g->delivered[slot] = true;
	return PWN_OK;
}

int pwn_group_wait_frame(pwn_ctx *h, int slot, pwn_frame *out)
{
	pwn_group *g = h->grp;
	if(slot < 0 || slot >= g->nslots || g->fdone[slot].seq == 0) return PWN_EINVAL;
	while(g->in_flight[slot] && !g->delivered[slot])
	{
		const int rc = deliver_oldest(g);
		if(rc != PWN_OK) return rc;
	}
	g->in_flight[slot] = false;
	if(out != NULL) *out = g->fdone[slot];
	return PWN_OK;
}

int pwn_group_frame_ready(pwn_ctx *h, int slot)
{
	pwn_group *g = h->grp;
	if(slot < 0 || slot >= g->nslots) return PWN_EINVAL;
	if(!g->in_flight[slot] || g->delivered[slot]) return 1;
	// every member's share of the slot's frame -- the k-th oldest in flight -- is through on its device (a job on the members'
	// threads, behind the submissions that were posted: it waits for those to be ENQUEUED, not for the GPUs)
	int k = -1;
	for(int i = 0; i < g->fifo_n; i++) if(g->fifo[i] == slot) k = i;
	if(k < 0) return 0;
	std::atomic<int> *all = new std::atomic<int>(1);
	const int rc = run_all(g, [g, k, all](int i) { const int r = pwn_i_tiled_ready(g->m[i], k); if(r < 0) return r; if(r == 0) all->store(0); return PWN_OK; });
	const int ready = all->load();
	delete all;
	if(rc != PWN_OK) return frame_failed(g, rc);
	return ready;
}

int pwn_group_get_stats(pwn_ctx *h, pwn_stats *out)
{
	pwn_group *g = h->grp;
	std::vector<pwn_stats> st((size_t)g->n);
	pwn_stats *sp = st.data();
	const int rc = run_all(g, [g, sp](int i) { return pwn_get_stats(g->m[i], &sp[i]); });
	if(rc != PWN_OK) return rc;
	pwn_stats sum = h->stats;
	bool first = true;
	for(int i = 0; i < g->n; i++)
	{
		if(!g->m[i]->counters_on && !g->m[i]->wave_log_on) continue;
		// counted frames: every member counts its strip
		const pwn_stats &s = st[(size_t)i];
		if(first)
		{
			sum.rays = sum.steps = sum.portals = sum.sphere_tests = sum.exhausted = sum.wave_steps = 0;
			memset(sum.wave_paths, 0, sizeof(sum.wave_paths)); memset(sum.regions, 0, sizeof(sum.regions));
			sum.phase_passes = sum.phase_lanes = sum.wave_time = sum.waves = sum.kernel_span = 0;
			first = false;
		}
		sum.rays += s.rays; sum.steps += s.steps; sum.portals += s.portals; sum.sphere_tests += s.sphere_tests; sum.exhausted += s.exhausted; sum.wave_steps += s.wave_steps;
		for(int k = 0; k < 8; k++) sum.wave_paths[k] += s.wave_paths[k];
		for(int k = 0; k < 32; k++) sum.regions[k] += s.regions[k];
		sum.phase_passes += s.phase_passes; sum.phase_lanes += s.phase_lanes; sum.wave_time += s.wave_time; sum.waves += s.waves;
		if(s.kernel_span > sum.kernel_span) sum.kernel_span = s.kernel_span;
	}
	*out = sum;
	return PWN_OK;
}

// screen_upscale (screen.h:126-149, main.c:108) of a delivered frame: on member 0's device.  sbuf == NULL: the frame the last
// pwn_trace_screen_centred delivered (the host still holds it: it is the host's own sbuf)
int pwn_group_screen_upscale(pwn_ctx *h, const uint32_t *sbuf, int scale, int pitch_bytes, uint32_t *pixels)
{
	pwn_group *g = h->grp;
	if(sbuf == NULL) sbuf = g->last_sbuf;
	if(sbuf == NULL) { snprintf(h->err, sizeof(h->err), "no frame was delivered yet"); return PWN_EINVAL; }
	return pwn_group_on_member0(h, [sbuf, scale, pitch_bytes, pixels](pwn_ctx *m0) { return pwn_screen_upscale(m0, sbuf, scale, pitch_bytes, pixels); });
}
