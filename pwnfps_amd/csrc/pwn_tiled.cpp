// pwn_tiled.cpp -- one frame row-tiled over the GPUs of a node, behind the C ABI.
//
// One process per GPU, each with its own pwn_ctx; rank r owns the strip of rows
// [r * rows_per, min((r + 1) * rows_per, h)), rows_per = ceil(h / world) rounded up to 8
// (the reference parallelises the same two loops with OpenMP over rows, screen.h:63,77).
// The trace pass needs no exchange.  The blur does: its taps reach 0.002*h*(depth-1) rows
// (screen.h:100-102), unbounded in depth.  Per submitted frame f (slot s = f mod 6) a rank enqueues, IN ORDER ON THE
// FRAME'S OWN COMPUTE STREAM (frames alternate between two):
//
//   trace strip f -> pre[s], z[s]
//   G1(f)   one grouped exchange, a single RCCL launch: the H border rows of strip f to / from the neighbour strips,
//           straight out of / into the full-frame plane pre[s] (or, without a halo, every strip to everybody: an
//           all-gather by send / recv)
//   blur strip f from pre rows [y0-H, y1+H) -> out; taps outside those rows are counted in the rank's miss word
//   G2(f)   the second grouped exchange: the FINISHED strip of frame f to the frame's root (the gather) and the rank's
//           two words to every rank (see below)
//   a kernel that stores every rank's words into pinned host memory, and the event pwn_tiled_wait waits for
//
// While one stream's frame is exchanged the other stream's frame is traced: a frame costs max(kernels, exchange), not
// their sum, and no event crosses a queue.  (PWN_TILED_CHOREO_SPLIT, the form of rounds 2-3, is kept as an option: the
// exchanges on a third stream tied to the kernels by four events per frame, blur f enqueued by submit f+1 behind the
// next trace, G2(f) by submit f+2 in front of G1(f+2).  Measured on one GPU, round 4: those queue-to-queue waits cost a
// rank 0.11-0.19 ms per frame whatever the frame's size -- a 64 x 32 frame, a strip of an 8-way tiled 4K frame whose
// kernels take 0.05 ms -- against 0.044 / 0.059 ms in-stream, 0.088 with RCCL carrying a real rank's bytes to itself;
// profiles/r4/host_bound.txt.)  At most five frames are in flight, six buffer sets.
// Every rank holds every rank's miss word of a delivered frame f: if one is non-zero the bounded halo was
// not enough for that frame and ALL ranks, having the same words, repeat its exchange with whole
// strips, its blur and its gather before it is delivered, and use whole strips from then on.  A
// delivered frame is always exact; the host synchronises only at delivery, behind its submissions.
//
// Host sink (pwn_tiled_host_sink): frames are delivered to ONE frame buffer in host memory that every rank has
// mapped and registered.  There is then no gather: behind its blur every rank copies its strip into the frame on a
// copy stream of its own (N PCIe links in parallel), and G2 is one word per pair of ranks, sent behind the
// sender's copy on that stream -- when a rank has received every other rank's word of frame f, every strip of f
// is in host memory.  The word is the miss word, so the exactness protocol is unchanged.
//
// The transport is a small interface: RCCL (ncclSend / ncclRecv in a group, loaded with dlopen so
// that libpwnhip.so has no link-time dependency on it), or -- for tests on a box with one GPU,
// where RCCL cannot run two ranks -- POSIX shared memory through the host.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include <rccl/rccl.h>          // types only: the entry points are resolved at run time
#include "pwn_internal.h"

// ---------------------------------------------------------------- transports ----
struct pwn_transport
{
	virtual ~pwn_transport() {}
	// a group = sends and receives that progress together; stream-ordered on `stream`
	virtual int begin(hipStream_t stream) = 0;
	virtual int send(const void *d_src, size_t bytes, int peer) = 0;
	virtual int recv(void *d_dst, size_t bytes, int peer) = 0;
	virtual int end() = 0;
	virtual const char *name() const = 0;
	// Deadlines (pwn_tiled_set_timeouts).  A transport never waits for a peer longer than wait_ms inside end(); `alive`
	// asks it for an error that arrived asynchronously (a peer that died, a link that went down) while the caller polls
	// an event; `abort` tears the connection down so that whatever it has on the device leaves -- after it the
	// transport is dead (every later call fails) and its destructor frees nothing that the abort already freed.
	virtual int alive() { return PWN_OK; }
	virtual void abort() { dead = true; }
	int wait_ms = 60000;
	bool dead = false;
	char err[200];
};

static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static int env_ms(const char *name, int dflt)
{
	const char *e = getenv(name);
	if(e == NULL || *e == 0) return dflt;
	const long v = strtol(e, NULL, 10);
	return (v > 0 && v < 86400000L) ? (int)v : dflt;
}
#define PWN_INIT_TIMEOUT_DEFAULT_MS 120000
#define PWN_WAIT_TIMEOUT_DEFAULT_MS 60000

// ---- RCCL over xGMI
struct rccl_api
{
	void *lib;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *);
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*GroupStart)(void);
	ncclResult_t (*GroupEnd)(void);
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	const char *(*GetErrorString)(ncclResult_t);
	// present in every RCCL this was built against, but optional here: without them the non-blocking mode is refused and a
	// deadline that passes cannot abort, only report
	ncclResult_t (*CommInitRankConfig)(ncclComm_t *, int, ncclUniqueId, int, ncclConfig_t *);
	ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *);
	ncclResult_t (*CommAbort)(ncclComm_t);
	ncclResult_t (*GetVersion)(int *);
	char path[256];                 // the file dlopen resolved (dladdr of ncclGetUniqueId)
};

static rccl_api *rccl_load(char *err, size_t errlen)
{
	static rccl_api api;
	static int state = 0;          // 0 untried, 1 ok, -1 failed
	if(state == 1) return &api;
	if(state == -1) { snprintf(err, errlen, "librccl could not be loaded"); return NULL; }
	// a process that already has an RCCL (PyTorch ships its own) gets that one: same soname
	const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	for(size_t i = 0; i < sizeof(names) / sizeof(names[0]) && api.lib == NULL; i++) api.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
	if(api.lib == NULL) { state = -1; snprintf(err, errlen, "dlopen(librccl): %s", dlerror()); return NULL; }
#define SYM(field, name) do { *(void **)&api.field = dlsym(api.lib, name); if(api.field == NULL) { state = -1; \
	snprintf(err, errlen, "librccl has no %s", name); return NULL; } } while(0)
	SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
	SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
	SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
	*(void **)&api.CommInitRankConfig = dlsym(api.lib, "ncclCommInitRankConfig");
	*(void **)&api.CommGetAsyncError = dlsym(api.lib, "ncclCommGetAsyncError");
	*(void **)&api.CommAbort = dlsym(api.lib, "ncclCommAbort");
	*(void **)&api.GetVersion = dlsym(api.lib, "ncclGetVersion");
	{
		Dl_info di;
		api.path[0] = 0;
		if(dladdr((void *)api.GetUniqueId, &di) != 0 && di.dli_fname != NULL) snprintf(api.path, sizeof(api.path), "%s", di.dli_fname);
	}
	state = 1;
	return &api;
}

// How the communicator is driven (PWN_TILED_RCCL_MODE):
//   blocking (default)  ncclCommInitRank and the first exchange with every peer -- the calls that wait for OTHER
//                       processes -- run on a helper thread under the bring-up deadline; afterwards a grouped send / recv
//                       launch only enqueues a kernel and returns, as in rounds 2-3, and what can still hang is that kernel
//                       on the device: pwn_tiled_wait polls its events against the wait deadline.
//   nonblocking         ncclCommInitRankConfig with blocking = 0: every call may return ncclInProgress and is polled with
//                       ncclCommGetAsyncError against the deadline (a grouped launch is then handed to a thread of RCCL's
//                       and this one waits for it).  Costs host time per group; `tiling.sweep` of bench.py times both.
// On expiry, either way: ncclCommAbort, PWN_ETIMEDOUT, the transport is dead.
struct rccl_transport : pwn_transport
{
	// `comm` is the communicator of the group in hand: comms[0], or -- PWN_OPT_TILED_COMMS -- the one bound to the group's
	// stream (a communicator runs its launches in the order they were made, whatever their streams: with one per compute
	// stream a frame's halo rows do not wait for the other stream's gather)
	rccl_api *api; ncclComm_t comm; hipStream_t stream; bool nb; int rank, world;
	ncclComm_t comms[3]; hipStream_t keys[3]; int ncomm;
	rccl_transport() : api(NULL), comm(NULL), stream(NULL), nb(false), rank(0), world(1), ncomm(0) { err[0] = 0; for(int i = 0; i < 3; i++) { comms[i] = NULL; keys[i] = NULL; } }
	~rccl_transport() { for(int i = 0; i < ncomm; i++) if(comms[i] && !dead) (void)api->CommDestroy(comms[i]); }
	void abort()
	{
		for(int i = 0; i < ncomm; i++) if(!dead && comms[i] && api->CommAbort) (void)api->CommAbort(comms[i]);
		dead = true;
	}
	// nonblocking: the call that returned ncclInProgress is complete when the communicator's state is ncclSuccess again
	int settle(const char *what)
	{
		const double t0 = now_ms();
		for(unsigned spins = 0;; spins++)
		{
			ncclResult_t st = ncclSuccess;
			const ncclResult_t q = api->CommGetAsyncError(comm, &st);
			if(q != ncclSuccess) { snprintf(err, sizeof(err), "rank %d: ncclCommGetAsyncError after %s: %s", rank, what, api->GetErrorString(q)); dead = true; return PWN_EHIP; }
			if(st == ncclSuccess) return PWN_OK;
			if(st != ncclInProgress) { snprintf(err, sizeof(err), "rank %d: %s: %s", rank, what, api->GetErrorString(st)); abort(); return PWN_EHIP; }
			if(now_ms() - t0 > (double)wait_ms)
			{
				abort();
				snprintf(err, sizeof(err), "rank %d of %d: %s still in progress after %d ms; communicator aborted", rank, world, what, wait_ms);
				return PWN_ETIMEDOUT;
			}
			if(spins > 500) { struct timespec ts = { 0, 20 * 1000 }; nanosleep(&ts, NULL); }
		}
	}
	int chk(ncclResult_t r, const char *what, bool wait_here)
	{
		if(r == ncclSuccess) return PWN_OK;
		if(r == ncclInProgress && nb) return wait_here ? settle(what) : PWN_OK;
		snprintf(err, sizeof(err), "rank %d: %s: %s", rank, what, api->GetErrorString(r));
		return PWN_EHIP;
	}
	int begin(hipStream_t s)
	{
		if(dead) { snprintf(err, sizeof(err), "rank %d: the communicator is gone (aborted earlier)", rank); return PWN_ETIMEDOUT; }
		stream = s; comm = comms[0];
		for(int i = 1; i < ncomm; i++) if(keys[i] == s) comm = comms[i];
		return chk(api->GroupStart(), "ncclGroupStart", true);
	}
	int send(const void *p, size_t n, int peer) { return chk(api->Send(p, n, ncclUint8, peer, comm, stream), "ncclSend", false); }
	int recv(void *p, size_t n, int peer) { return chk(api->Recv(p, n, ncclUint8, peer, comm, stream), "ncclRecv", false); }
	int end() { return chk(api->GroupEnd(), "ncclGroupEnd", true); }
	int alive()
	{
		if(dead) return PWN_ETIMEDOUT;
		if(api->CommGetAsyncError == NULL) return PWN_OK;
		for(int i = 0; i < ncomm; i++)
		{
			ncclResult_t st = ncclSuccess;
			if(api->CommGetAsyncError(comms[i], &st) != ncclSuccess) continue;
			if(st == ncclSuccess || st == ncclInProgress) continue;
			snprintf(err, sizeof(err), "rank %d: communicator %d reports %s", rank, i, api->GetErrorString(st));
			return PWN_EHIP;
		}
		return PWN_OK;
	}
	const char *name() const { return "rccl"; }
};

// ---- bring-up of the RCCL communicator under a deadline: ncclCommInitRank, then one word to and from EVERY other
// rank (to itself when alone) in one grouped launch, so that every connection a frame will use -- RCCL connects a pair
// of ranks the first time they talk -- exists when pwn_tiled_init returns, and a peer that is not there is found out
// here, under init_ms, not in the middle of the first frame.  Self-contained (its own stream and four-byte buffers, owned
// by the shared state): in blocking mode it runs on a helper thread that the caller may have to leave behind.
struct rccl_bringup
{
	std::mutex m; std::condition_variable cv; bool done = false;
	std::atomic<int> stage{0};                     // 0 not started, 1 in ncclCommInitRank, 2 in the first exchange, 3 through
	std::atomic<ncclComm_t> comm{nullptr};
	std::atomic<bool> abandon{false};             // the deadline passed: whatever still runs gives up at its next look
	int rc = PWN_OK; char err[200] = { 0 };
	hipStream_t s = NULL; uint32_t *d = NULL;      // owned: freed by the last holder
	int device = 0;
	~rccl_bringup() { (void)hipSetDevice(device); if(d) (void)hipFree(d); if(s) (void)hipStreamDestroy(s); }
};
static const char *bringup_stage(int st)
{
	return st <= 0 ? "before ncclCommInitRank" : st == 1 ? "inside ncclCommInitRank (a rank that never called it, or no route to it)" :
	       st == 2 ? "in the first exchange with every other rank (a connection that does not come up, or a peer that left)" : "done";
}
static void rccl_bringup_run(rccl_api *api, std::shared_ptr<rccl_bringup> st, ncclUniqueId uid, int world, int rank, bool nb, double deadline)
{
	rccl_bringup &b = *st;
#define FAIL(code, ...) do { snprintf(b.err, sizeof(b.err), __VA_ARGS__); b.rc = (code); return; } while(0)
	if(hipSetDevice(b.device) != hipSuccess) FAIL(PWN_EHIP, "rank %d: hipSetDevice(%d) failed", rank, b.device);
	// nonblocking: wait for the communicator to settle; with a deadline of its own (no helper thread in this mode)
	auto settle = [&](ncclComm_t cm, const char *what) -> int
	{
		for(unsigned spins = 0;; spins++)
		{
			ncclResult_t a = ncclSuccess;
			const ncclResult_t q = api->CommGetAsyncError(cm, &a);
			if(q != ncclSuccess) { snprintf(b.err, sizeof(b.err), "rank %d: ncclCommGetAsyncError after %s: %s", rank, what, api->GetErrorString(q)); return PWN_EHIP; }
			if(a == ncclSuccess) return PWN_OK;
			if(a != ncclInProgress) { snprintf(b.err, sizeof(b.err), "rank %d: %s: %s", rank, what, api->GetErrorString(a)); return PWN_EHIP; }
			if(now_ms() > deadline || b.abandon.load()) { snprintf(b.err, sizeof(b.err), "rank %d of %d: %s still in progress at the bring-up deadline", rank, world, what); return PWN_ETIMEDOUT; }
			if(spins > 200) { struct timespec ts = { 0, 50 * 1000 }; nanosleep(&ts, NULL); }
		}
	};
	b.stage = 1;
	ncclComm_t cm = NULL;
	if(nb)
	{
		ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
		cfg.blocking = 0;
		const ncclResult_t r = api->CommInitRankConfig(&cm, world, uid, rank, &cfg);
		b.comm = cm;
		if(r != ncclSuccess && r != ncclInProgress) FAIL(PWN_EHIP, "rank %d: ncclCommInitRankConfig: %s", rank, api->GetErrorString(r));
		const int rc = settle(cm, "ncclCommInitRankConfig");
		if(rc != PWN_OK) { b.rc = rc; return; }
	}
	else
	{
		const ncclResult_t r = api->CommInitRank(&cm, world, uid, rank);
		b.comm = cm;
		if(r != ncclSuccess) FAIL(PWN_EHIP, "rank %d: ncclCommInitRank: %s", rank, api->GetErrorString(r));
	}
	if(b.abandon.load())
	{
		// the caller gave up while this thread sat in ncclCommInitRank: whoever takes the handle out of the state aborts it
		ncclComm_t mine = b.comm.exchange(nullptr);
		if(mine && api->CommAbort) (void)api->CommAbort(mine);
		FAIL(PWN_ETIMEDOUT, "rank %d: abandoned after ncclCommInitRank", rank);
	}
	b.stage = 2;
	// d[0] = this rank's word, d[1 + r] = what rank r sent
	if(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking) != hipSuccess) FAIL(PWN_EHIP, "rank %d: hipStreamCreate failed", rank);
	if(hipMalloc((void **)&b.d, (size_t)(world + 1) * 4) != hipSuccess) FAIL(PWN_ENOMEM, "rank %d: hipMalloc failed", rank);
	std::vector<uint32_t> h((size_t)world + 1, 0u);
	h[0] = 0x50574e00u + (uint32_t)rank;
	if(hipMemcpy(b.d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(PWN_EHIP, "rank %d: hipMemcpy failed", rank);
	{
		ncclResult_t r = api->GroupStart();
		if(r != ncclSuccess && !(nb && r == ncclInProgress)) FAIL(PWN_EHIP, "rank %d: ncclGroupStart: %s", rank, api->GetErrorString(r));
		for(int k = 0; k < world; k++)
		{
			const int peer = world == 1 ? 0 : k;
			if(world > 1 && peer == rank) continue;
			r = api->Send(b.d, 4, ncclUint8, peer, cm, b.s);
			if(r != ncclSuccess && !(nb && r == ncclInProgress)) { (void)api->GroupEnd(); FAIL(PWN_EHIP, "rank %d: ncclSend to %d: %s", rank, peer, api->GetErrorString(r)); }
			r = api->Recv(b.d + 1 + peer, 4, ncclUint8, peer, cm, b.s);
			if(r != ncclSuccess && !(nb && r == ncclInProgress)) { (void)api->GroupEnd(); FAIL(PWN_EHIP, "rank %d: ncclRecv from %d: %s", rank, peer, api->GetErrorString(r)); }
			if(world == 1) break;
		}
		r = api->GroupEnd();
		if(nb && (r == ncclInProgress || r == ncclSuccess)) { const int rc = settle(cm, "the first ncclGroupEnd"); if(rc != PWN_OK) { b.rc = rc; return; } }
		else if(r != ncclSuccess) FAIL(PWN_EHIP, "rank %d: ncclGroupEnd: %s", rank, api->GetErrorString(r));
	}
	// the launch is on the stream: it ends when every peer's word is here
	for(unsigned spins = 0;; spins++)
	{
		const hipError_t e = hipStreamQuery(b.s);
		if(e == hipSuccess) break;
		if(e != hipErrorNotReady) FAIL(PWN_EHIP, "rank %d: the first exchange failed on the device: %s", rank, hipGetErrorString(e));
		if(now_ms() > deadline || b.abandon.load()) FAIL(PWN_ETIMEDOUT, "rank %d of %d: the first exchange did not complete before the bring-up deadline", rank, world);
		if(api->CommGetAsyncError != NULL && (spins & 63u) == 63u)
		{
			ncclResult_t a = ncclSuccess;
			if(api->CommGetAsyncError(cm, &a) == ncclSuccess && a != ncclSuccess && a != ncclInProgress)
				FAIL(PWN_EHIP, "rank %d: the first exchange: %s", rank, api->GetErrorString(a));
		}
		struct timespec ts = { 0, 100 * 1000 };
		nanosleep(&ts, NULL);
	}
	(void)hipGetLastError();
	if(hipMemcpy(h.data(), b.d, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) FAIL(PWN_EHIP, "rank %d: hipMemcpy failed", rank);
	for(int k = 0; k < world; k++)
	{
		if(world > 1 && k == rank) continue;
		if(h[1 + (size_t)k] != 0x50574e00u + (uint32_t)k)
			FAIL(PWN_EHIP, "rank %d: the test word from rank %d arrived as %08x", rank, k, h[1 + (size_t)k]);
	}
	b.stage = 3;
#undef FAIL
}

// returns PWN_OK with *out = the communicator, or an error with `err` filled; never hangs longer than init_ms (+ 5 s)
static int rccl_bringup_bounded(rccl_api *api, int device, const ncclUniqueId &uid, int world, int rank, bool nb, int init_ms,
	ncclComm_t *out, char *err, size_t errlen)
{
	std::shared_ptr<rccl_bringup> st = std::make_shared<rccl_bringup>();
	st->device = device;
	const double deadline = now_ms() + (double)init_ms;
	*out = NULL;
	if(nb)
	{
		// every wait inside polls against the deadline itself
		rccl_bringup_run(api, st, uid, world, rank, true, deadline);
		ncclComm_t cm = st->comm.load();
		if(st->rc != PWN_OK)
		{
			st->comm = nullptr;
			if(cm && api->CommAbort) (void)api->CommAbort(cm);
			snprintf(err, errlen, "%s [stage: %s; limit %d ms; nonblocking]", st->err, bringup_stage(st->stage.load()), init_ms);
			return st->rc;
		}
		*out = cm;
		return PWN_OK;
	}
	std::thread th([api, st, uid, world, rank, deadline]()
	{
		rccl_bringup_run(api, st, uid, world, rank, false, deadline);
		{ std::lock_guard<std::mutex> g(st->m); st->done = true; }
		st->cv.notify_all();
	});
	bool through;
	{
		std::unique_lock<std::mutex> lk(st->m);
		through = st->cv.wait_for(lk, std::chrono::milliseconds(init_ms), [&] { return st->done; });
	}
	if(!through)
	{
		// The helper sits in a call that waits for another process.  With a communicator in hand ncclCommAbort makes that
		// call return; inside ncclCommInitRank there is none yet, and the thread is left behind (it owns everything it uses).
		st->abandon = true;
		{ struct timespec ts = { 0, 2 * 1000 * 1000 }; nanosleep(&ts, NULL); }      // (a helper that polls sees the flag before the handle goes)
		ncclComm_t cm = st->comm.exchange(nullptr);
		const int stage = st->stage.load();
		if(cm && api->CommAbort) (void)api->CommAbort(cm);
		{
			std::unique_lock<std::mutex> lk(st->m);
			through = st->cv.wait_for(lk, std::chrono::milliseconds(5000), [&] { return st->done; });
		}
		if(through) th.join(); else th.detach();
		snprintf(err, errlen, "rank %d of %d: the RCCL bring-up did not finish within %d ms; it was %s%s", rank, world, init_ms,
			bringup_stage(stage), cm ? "; communicator aborted" : "");
		return PWN_ETIMEDOUT;
	}
	th.join();
	if(st->rc != PWN_OK)
	{
		ncclComm_t cm = st->comm.exchange(nullptr);
		if(cm && st->rc == PWN_ETIMEDOUT && api->CommAbort) (void)api->CommAbort(cm);
		else if(cm) (void)api->CommDestroy(cm);
		snprintf(err, errlen, "%s [stage: %s; limit %d ms]", st->err, bringup_stage(st->stage.load()), init_ms);
		return st->rc;
	}
	*out = st->comm.load();
	return PWN_OK;
}

// A further communicator of the same ranks needs an id of its own: rank 0 makes it and sends it to every rank over the
// communicator that is up (every pair is connected: the bring-up's first exchange).  Bounded by `ms`.
static int rccl_share_id(rccl_api *api, ncclComm_t cm, int world, int rank, ncclUniqueId *id, int ms, char *err, size_t errlen)
{
	if(rank == 0 && api->GetUniqueId(id) != ncclSuccess) { snprintf(err, errlen, "ncclGetUniqueId failed"); return PWN_EHIP; }
	if(world == 1) return PWN_OK;
	hipStream_t s = NULL; unsigned char *d = NULL;
	int rc = PWN_OK;
	do
	{
		if(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&d, sizeof(ncclUniqueId)) != hipSuccess) { rc = PWN_EHIP; break; }
		if(rank == 0 && hipMemcpy(d, id, sizeof(ncclUniqueId), hipMemcpyHostToDevice) != hipSuccess) { rc = PWN_EHIP; break; }
		ncclResult_t r = api->GroupStart();
		for(int k = 1; k < world && r == ncclSuccess && rank == 0; k++) r = api->Send(d, sizeof(ncclUniqueId), ncclUint8, k, cm, s);
		if(r == ncclSuccess && rank != 0) r = api->Recv(d, sizeof(ncclUniqueId), ncclUint8, 0, cm, s);
		const ncclResult_t e = api->GroupEnd();
		if(r != ncclSuccess || e != ncclSuccess) { snprintf(err, errlen, "rank %d: sending a further communicator's id: %s", rank, api->GetErrorString(r != ncclSuccess ? r : e)); rc = PWN_EHIP; break; }
		const double t0 = now_ms();
		for(;;)
		{
			const hipError_t q = hipStreamQuery(s);
			if(q == hipSuccess) break;
			if(q != hipErrorNotReady) { rc = PWN_EHIP; break; }
			if(now_ms() - t0 > (double)ms) { snprintf(err, errlen, "rank %d of %d: a further communicator's id did not arrive within %d ms", rank, world, ms); rc = PWN_ETIMEDOUT; break; }
			struct timespec ts = { 0, 100 * 1000 }; nanosleep(&ts, NULL);
		}
		if(rc != PWN_OK) break;
		if(rank != 0 && hipMemcpy(id, d, sizeof(ncclUniqueId), hipMemcpyDeviceToHost) != hipSuccess) rc = PWN_EHIP;
	} while(0);
	if(rc == PWN_EHIP && err[0] == 0) snprintf(err, errlen, "rank %d: a HIP call failed while sending a further communicator's id", rank);
	if(rc != PWN_ETIMEDOUT) { if(d) (void)hipFree(d); if(s) (void)hipStreamDestroy(s); }      // (on expiry the stream still holds the launch: left to the abort)
	return rc;
}

// ---- shared memory through the host (tests: several ranks on ONE GPU).  One mailbox per
// ordered pair of ranks: a ring of SHM_SLOTS messages of at most `slot_bytes`; a sender copies
// device -> mailbox and publishes the message number, the receiver waits for that number,
// copies mailbox -> device and frees the slot.  Messages between two ranks are matched in
// order, like RCCL's.  Host-synchronous by design.
#define SHM_SLOTS 4
struct shm_box { volatile unsigned long long sent, taken; unsigned long long pad[6]; };   // 64 B
struct shm_transport : pwn_transport
{
	int rank, world; size_t slot_bytes, total; char shm_name[64]; bool owner;
	unsigned char *base; hipStream_t stream;
	struct op { bool is_send; const void *src; void *dst; size_t bytes; int peer; };
	std::vector<op> ops;
	shm_transport() : rank(0), world(1), slot_bytes(0), total(0), owner(false), base(NULL), stream(NULL) { err[0] = 0; shm_name[0] = 0; }
	~shm_transport()
	{
		if(base) munmap(base, total);
		if(owner && shm_name[0]) shm_unlink(shm_name);
	}
	shm_box *box(int src, int dst) { return (shm_box *)(base + ((size_t)src * world + dst) * sizeof(shm_box)); }
	unsigned char *slot(int src, int dst, unsigned long long n)
	{
		return base + (size_t)world * world * sizeof(shm_box) + (((size_t)src * world + dst) * SHM_SLOTS + (size_t)(n % SHM_SLOTS)) * slot_bytes;
	}
	int open_region(const char *name, int rank_, int world_, size_t slot_bytes_)
	{
		rank = rank_; world = world_; slot_bytes = (slot_bytes_ + 63) & ~(size_t)63;
		snprintf(shm_name, sizeof(shm_name), "%s", name);
		total = (size_t)world * world * sizeof(shm_box) + (size_t)world * world * SHM_SLOTS * slot_bytes;
		int fd = -1;
		if(rank == 0)
		{
			fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
			owner = fd >= 0;
			if(fd >= 0 && ftruncate(fd, (off_t)total) != 0) { close(fd); fd = -1; }
		}
		else
		{
			// rank 0 creates it and sizes it; wait for both
			for(int tries = 0; tries < 3000; tries++)
			{
				fd = shm_open(shm_name, O_RDWR, 0600);
				struct stat sb;
				if(fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= total) break;
				if(fd >= 0) { close(fd); fd = -1; }
				struct timespec ts = { 0, 10 * 1000 * 1000 };
				nanosleep(&ts, NULL);
			}
		}
		if(fd < 0) { snprintf(err, sizeof(err), "shm_open(%s): %s", shm_name, strerror(errno)); return PWN_EIO; }
		base = (unsigned char *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if(base == (unsigned char *)MAP_FAILED) { base = NULL; snprintf(err, sizeof(err), "mmap(%s): %s", shm_name, strerror(errno)); return PWN_ENOMEM; }
		return PWN_OK;
	}
	int begin(hipStream_t s)
	{
		if(dead) { snprintf(err, sizeof(err), "rank %d: the transport is dead (a deadline passed earlier)", rank); return PWN_ETIMEDOUT; }
		stream = s; ops.clear(); return PWN_OK;
	}
	int send(const void *p, size_t n, int peer) { op o = { true, p, NULL, n, peer }; ops.push_back(o); return n <= slot_bytes ? PWN_OK : PWN_EINVAL; }
	int recv(void *p, size_t n, int peer) { op o = { false, NULL, p, n, peer }; ops.push_back(o); return n <= slot_bytes ? PWN_OK : PWN_EINVAL; }
	int wait_until(volatile unsigned long long *word, unsigned long long at_least)
	{
		// a peer that died must not hang its neighbours: give up at the deadline (pwn_tiled_set_timeouts)
		const double t0 = now_ms();
		for(unsigned long long spins = 0; __atomic_load_n(word, __ATOMIC_ACQUIRE) < at_least; spins++)
		{
			if(spins > 2000)
			{
				if(now_ms() - t0 > (double)wait_ms)
				{
					dead = true;
					snprintf(err, sizeof(err), "rank %d of %d: a peer did not answer within %d ms", rank, world, wait_ms);
					return PWN_ETIMEDOUT;
				}
				struct timespec ts = { 0, 100 * 1000 }; nanosleep(&ts, NULL);
			}
		}
		return PWN_OK;
	}
	int end()
	{
		// what is sent was produced on `stream`
		if(hipStreamSynchronize(stream) != hipSuccess) { snprintf(err, sizeof(err), "hipStreamSynchronize failed"); return PWN_EHIP; }
		// all sends first (rings have room for a group's messages), then the receives
		for(size_t i = 0; i < ops.size(); i++)
		{
			const op &o = ops[i];
			if(!o.is_send) continue;
			shm_box *b = box(rank, o.peer);
			const unsigned long long n = b->sent;
			int rc = wait_until(&b->taken, n + 1 >= SHM_SLOTS ? n + 1 - SHM_SLOTS : 0);
			if(rc != PWN_OK) return rc;
			if(o.bytes && hipMemcpy(slot(rank, o.peer, n), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) { snprintf(err, sizeof(err), "D2H failed"); return PWN_EHIP; }
			__atomic_store_n(&b->sent, n + 1, __ATOMIC_RELEASE);
		}
		for(size_t i = 0; i < ops.size(); i++)
		{
			const op &o = ops[i];
			if(o.is_send) continue;
			shm_box *b = box(o.peer, rank);
			const unsigned long long n = b->taken;
			int rc = wait_until(&b->sent, n + 1);
			if(rc != PWN_OK) return rc;
			if(o.bytes && hipMemcpy(o.dst, slot(o.peer, rank, n), o.bytes, hipMemcpyHostToDevice) != hipSuccess) { snprintf(err, sizeof(err), "H2D failed"); return PWN_EHIP; }
			__atomic_store_n(&b->taken, n + 1, __ATOMIC_RELEASE);
		}
		return PWN_OK;
	}
	const char *name() const { return "shm"; }
};

// ---- inside one process (pwn_init_multi: the members of a group, one thread each; PWN_TRANSPORT_LOCAL).  A send is an entry in
// the mailbox of the ordered pair (pwn_hub_box, pwn_internal.h): the receiver posts the place, the sender copies there on its own
// stream -- device to device: a peer-to-peer DMA over xGMI between two GPUs, a plain copy on one -- and the receiver's stream
// waits for the sender's event.  Messages between two ranks are matched in order, like RCCL's.  Nothing here is a kernel, so nothing
// competes with the persistent trace grids for CUs.  A member that waits for a peer's entry gives up at the deadline
// (pwn_tiled_set_timeouts) or as soon as the group says that a member has failed.
struct local_transport : pwn_transport
{
	pwn_hub *hub; int rank, world, device; hipStream_t stream;
	hipEvent_t evs[PWN_HUB_RING]; unsigned ev_next;
	struct op { bool is_send; const void *src; void *dst; size_t bytes; int peer; };
	std::vector<op> ops;
	local_transport() : hub(NULL), rank(0), world(1), device(0), stream(NULL), ev_next(0) { err[0] = 0; for(int i = 0; i < PWN_HUB_RING; i++) evs[i] = NULL; }
	~local_transport() { (void)hipSetDevice(device); for(int i = 0; i < PWN_HUB_RING; i++) if(evs[i]) (void)hipEventDestroy(evs[i]); }
	int open(pwn_hub *h, int rank_, int world_, int device_)
	{
		hub = h; rank = rank_; world = world_; device = device_;
		for(int i = 0; i < PWN_HUB_RING; i++) if(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming) != hipSuccess) { snprintf(err, sizeof(err), "hipEventCreate failed"); return PWN_EHIP; }
		return PWN_OK;
	}
	pwn_hub_box *box(int from, int to) { return &hub->boxes[(size_t)from * (size_t)world + (size_t)to]; }
	int begin(hipStream_t s)
	{
		if(dead) { snprintf(err, sizeof(err), "rank %d: the transport is dead (a deadline passed earlier)", rank); return PWN_ETIMEDOUT; }
		stream = s; ops.clear(); return PWN_OK;
	}
	int send(const void *p, size_t n, int peer) { op o = { true, p, NULL, n, peer }; ops.push_back(o); return PWN_OK; }
	int recv(void *p, size_t n, int peer) { op o = { false, NULL, p, n, peer }; ops.push_back(o); return PWN_OK; }
	int wait_until(std::atomic<unsigned long long> *word, unsigned long long at_least, const char *what, int peer)
	{
		const double t0 = now_ms();
		for(unsigned long long spins = 0; word->load(std::memory_order_acquire) < at_least; spins++)
		{
			if(hub->failed.load(std::memory_order_relaxed)) { dead = true; snprintf(err, sizeof(err), "rank %d of %d: another member of the group has failed", rank, world); return PWN_ETIMEDOUT; }
			if(spins > 2000)
			{
				if(now_ms() - t0 > (double)wait_ms)
				{
					dead = true; hub->failed.store(1);
					snprintf(err, sizeof(err), "rank %d of %d: member %d did not %s within %d ms", rank, world, peer, what, wait_ms);
					return PWN_ETIMEDOUT;
				}
				if(spins > 20000) { struct timespec ts = { 0, 50 * 1000 }; nanosleep(&ts, NULL); }
			}
		}
		return PWN_OK;
	}
	int end()
	{
		// 1. every receive of the group is posted: the place.  (No event "the place is free": what the tiling receives into -- halo rows
		//    and foreign strips of a slot's pre-blur plane, a root's assembled frame -- was last used by the frame NSLOT frames back, which
		//    was delivered, i.e. waited for on the host, before the slot was handed out again.  An event here would be a second wait
		//    between two queues per message: 20-50 us each beside a running trace grid.)
		const hipEvent_t ev_free = NULL;
		for(size_t i = 0; i < ops.size(); i++)
		{
			const op &o = ops[i];
			if(o.is_send) continue;
			pwn_hub_box *b = box(o.peer, rank);
			const unsigned long long n = b->posted.load(std::memory_order_relaxed);
			int rc = wait_until(&b->consumed, n + 1 >= PWN_HUB_RING ? n + 1 - PWN_HUB_RING : 0, "get through its earlier messages", o.peer);       // (room in the ring: this rank's own progress)
			if(rc != PWN_OK) return rc;
			pwn_hub_post &m = b->post[n % PWN_HUB_RING];
			m.dst = o.dst; m.bytes = o.bytes; m.ev_free = ev_free; m.device = device;
			b->posted.store(n + 1, std::memory_order_release);
		}
		// 2. every send: wait for the peer's post (it posts before it waits for anything), copy on THIS stream behind the peer's event,
		//    one event behind the group's copies
		bool sent = false;
		size_t first_send = ops.size();
		for(size_t i = 0; i < ops.size(); i++)
		{
			const op &o = ops[i];
			if(!o.is_send) continue;
			pwn_hub_box *b = box(rank, o.peer);
			unsigned long long n = b->copied.load(std::memory_order_relaxed);
			for(size_t k = 0; k < i; k++) if(ops[k].is_send && ops[k].peer == o.peer) n++;        // (earlier sends of this group to the same peer: published below)
			int rc = wait_until(&b->posted, n + 1, "post its receive", o.peer);
			if(rc != PWN_OK) return rc;
			const pwn_hub_post m = b->post[n % PWN_HUB_RING];
			if(m.bytes != o.bytes) { snprintf(err, sizeof(err), "rank %d: member %d expects %zu bytes where %zu are sent", rank, o.peer, m.bytes, o.bytes); hub->failed.store(1); return PWN_EINVAL; }
			hipError_t he = m.ev_free != NULL ? hipStreamWaitEvent(stream, m.ev_free, 0) : hipSuccess;
			if(he == hipSuccess && o.bytes)
				he = m.device == device ? hipMemcpyAsync(m.dst, o.src, o.bytes, hipMemcpyDeviceToDevice, stream)
				                        : hipMemcpyPeerAsync(m.dst, m.device, o.src, device, o.bytes, stream);
			if(he != hipSuccess) { snprintf(err, sizeof(err), "rank %d: copy to member %d: %s", rank, o.peer, hipGetErrorString(he)); hub->failed.store(1); return PWN_EHIP; }
			if(!sent) { sent = true; first_send = i; }
		}
		if(sent)
		{
			hipEvent_t e = evs[ev_next++ % PWN_HUB_RING];
			if(hipEventRecord(e, stream) != hipSuccess) { snprintf(err, sizeof(err), "hipEventRecord failed"); return PWN_EHIP; }
			for(size_t i = first_send; i < ops.size(); i++)
			{
				const op &o = ops[i];
				if(!o.is_send) continue;
				pwn_hub_box *b = box(rank, o.peer);
				const unsigned long long n = b->copied.load(std::memory_order_relaxed);
				b->done[n % PWN_HUB_RING] = e;
				b->copied.store(n + 1, std::memory_order_release);
			}
		}
		// 3. every receive: this stream behind the sender's event
		for(size_t i = 0; i < ops.size(); i++)
		{
			const op &o = ops[i];
			if(o.is_send) continue;
			pwn_hub_box *b = box(o.peer, rank);
			const unsigned long long n = b->consumed.load(std::memory_order_relaxed);
			int rc = wait_until(&b->copied, n + 1, "send", o.peer);
			if(rc != PWN_OK) return rc;
			const hipEvent_t e = b->done[n % PWN_HUB_RING];
			const hipError_t he = hipStreamWaitEvent(stream, e, 0);
			b->consumed.store(n + 1, std::memory_order_release);
			if(he != hipSuccess) { snprintf(err, sizeof(err), "rank %d: waiting for member %d's copy: %s", rank, o.peer, hipGetErrorString(he)); hub->failed.store(1); return PWN_EHIP; }
		}
		return PWN_OK;
	}
	int alive()
	{
		if(dead) return PWN_ETIMEDOUT;
		if(hub->failed.load(std::memory_order_relaxed)) { snprintf(err, sizeof(err), "rank %d: another member of the group has failed", rank); return PWN_EHIP; }
		return PWN_OK;
	}
	void abort() { dead = true; hub->failed.store(1); }
	const char *name() const { return "local"; }
};

// ---------------------------------------------------------------- state ----
#define NSLOT 6          // buffer sets: at most five frames in flight and the one being reused
#define MAXW PWN_TILED_MAX_WORLD
struct pwn_tiled
{
	int rank, world, per;
	int max_rows;                       // the tallest strip a rank may be given (what the shared-memory mailboxes are sized for)
	// Strip r = rows [cuts[r], cuts[r + 1]): `cuts` for the next submitted frame, `fcuts[s]` as used for the frame in
	// slot s -- the cuts move (pwn_tiled_balance), and a frame is exchanged, blurred, gathered and, after a missed
	// halo, repeated with the cuts it was traced with.  Every rank holds the same numbers for the same frame.
	int cuts[MAXW + 1], fcuts[NSLOT][MAXW + 1];
	int want_halo;                      // the halo asked for at init; moving cuts keep every strip at least this tall
	int halo;                           // rows exchanged with each neighbour; 0 = whole strips to everybody
	int fhalo[NSLOT];                   // ... as used for the frame in that slot (the mode changes after a miss)
	uint32_t fcost_mul[NSLOT], fcost_div[NSLOT];   // what the slot's trace launch says its cost word is to be scaled by (pwn_ctx.cost_mul / _div)
	int root_mode, froot[NSLOT];     // pwn_tiled_gather_root: PWN_TILED_ROOT_*; the rank the slot's frame is gathered on
	int balance_every;                  // re-cut every this many delivered frames from the ranks' cost words; 0 = never
	uint32_t last_cost[MAXW];           // the cost words of the last delivered frame (pwn_tiled_get_cuts)
	// what the re-cut works from: per rank the SMALLEST cost among the delivered frames that were traced with acc_cuts
	// (a launch that shared its GPU with something else for a moment -- another process, a clock dip -- reports a
	// cost too high, never one too low), reset when a frame with other cuts is delivered
	int acc_cuts[MAXW + 1]; uint32_t acc_cost[MAXW]; int acc_frames;
	pwn_transport *tp;
	hipStream_t comm;
	// Compute: the kernels of frame f (trace f, later blur f) on cs[f & 1] -- two streams, so that the trace grid of
	// frame f+1 moves onto the CUs while frame f's runs out of units (a strip of an 8-way tiling of a 4K frame is
	// three units per wave: the mean wave is resident for 2/3 of such a launch) and a stream that waits for its
	// frame's halo rows does not hold up the other.  Without PWN_OPT_FRAME_OVERLAP both are the context's one stream.
	hipStream_t cs[3]; int ncs;         // ncs = 1 (no PWN_OPT_FRAME_OVERLAP), 2, or 3 (PWN_OPT_TILED_STREAMS, in-stream only): frame f on cs[f % ncs]
	hipStream_t cs3;                    // the third compute stream, the tiling's own
	uint32_t *cost_acc;                 // device, a word per buffer set, 64 B apart: what the slot's trace launch cost so far (tables.h cost_word)
	uint32_t *pre[NSLOT], *out[NSLOT], *fin[NSLOT]; float *z[NSLOT];     // full-frame planes per frame slot (fin: rank 0)
	// the two words a rank says about a frame: [0] taps that left its halo (miss), [1] what its strip cost (the sum of
	// its trace waves' lifetimes, 100 MHz ticks); missw = this rank's, missv = every rank's (world x 2)
	uint32_t *missw[NSLOT], *missv[NSLOT];
	uint32_t *h_missv;                  // pinned: per slot, every rank's two words of the slot's frame (world x 2) and then this rank's own
	uint32_t *h_frame;                  // pinned host copy of a delivered frame (rank 0, PWN_TILED_HOST)
	// per slot: behind the frame's trace; behind the group G(f) it opened; behind its blur; behind the
	// group that carried its gather (that is G(f+2)'s event or the drain group's)
	hipEvent_t ev_t[NSLOT], ev_x[NSLOT], ev_b[NSLOT], ev_d[NSLOT];
	hipEvent_t ev_k0[NSLOT], ev_k1[NSLOT], ev_k2[NSLOT], ev_k3[NSLOT];      // timing (PWN_OPT_FRAME_TIMING): around the trace, around the blur
	hipEvent_t ev_g0[NSLOT], ev_g1[NSLOT], ev_g2[NSLOT], ev_g3[NSLOT];      // ... on the comm stream: around the gather group (g0, g1), around the halo group (g2, g3)
	hipStream_t fstream[NSLOT];         // the compute stream the slot's frame is on: cs[f & 1], or cs[0] for a counted frame (one set of counters)
	// PWN_TILED_SELF=1 with ONE rank over RCCL (measurement only: profiles/r4/rccl_self_exchange.txt): the rank sends itself what a rank
	// of a real tiling sends -- H border rows behind every trace, its finished strip and its two words behind every blur, two sends and
	// two receives to the same peer in each grouped launch -- into a scratch plane, so that RCCL's kernels, the room they find beside the
	// persistent trace grid and the host's cost per grouped launch can be measured on a box with one GPU
	bool self_exchange; uint32_t *self_buf;
	// The choreography (PWN_OPT_TILED_CHOREO, fixed at pwn_tiled_init).  IN-STREAM (default): everything of frame f -- trace, halo rows,
	// blur, gather, words -- is enqueued by pwn_tiled_submit(f) on the frame's own compute stream, in order; the exchange of frame f
	// overlaps the other stream's trace f+1.  No event crosses a queue.  SPLIT (rounds 2-3): the exchanges on a third stream, tied
	// to the kernels by four events per frame, blur f enqueued by submit f+1 and the gather of f by submit f+2.  Measured on one
	// GPU (profiles/r4/host_bound.txt): the split form's queue-to-queue waits cost a rank 150-200 us per frame whatever the frame's
	// size, three times what the kernels of a strip of an 8-way tiled 4K frame take.
	bool instream;
	hipEvent_t gathered_by[NSLOT];      // which of the above marks the slot's gather as done
	// host sink: NSLOT whole frames in host memory shared by the ranks; this rank's copies on their own stream
	uint8_t *host_base; void *host_registered;   // the frames; what this context registered with the device (or NULL)
	bool sink;                          // frames are delivered to the host by every rank (pwn_tiled_host_sink, or a group's own: pwn_i_tiled_sink)
	// a group (pwn_init_multi): where the slot's frame goes -- the caller's own sbuf / zbuf (main.c:31,33) or the group's pinned
	// frames -- instead of a place in host_base; and what the members share
	uint32_t *hdst[NSLOT]; float *hzdst[NSLOT];
	// a group's frames with the SDL sink (PWN_FRAME_SURFACE; screen_upscale, screen.h:126-149): every member upscales its own strip on
	// its device -- a strip of source rows [y0, y1) is the surface's words [y0 * rowadv, y1 * rowadv) -- and copies that into the
	// host surface of the slot's frame; dsurf: one device plane per frame slot of the group's host (pwn_i_tiled_surface)
	uint32_t *hsurf[NSLOT], *fsurf[NSLOT], *dsurf[PWN_MAX_SLOTS];
	int surf_scale, surf_pitch_words;
	float *fz[NSLOT];                   // the depth plane of the slot's frame: z[s], or -- a group -- the plane the group names: one per frame slot of ITS
	                                    // host and one for the blocking calls, so that a pixel whose primary ray runs out of steps keeps the depth of the
	                                    // slot's previous frame / of the previous call (trace.h:677), exactly as a one-device context's planes do
	pwn_hub *hub;
	hipStream_t copy;
	hipEvent_t ev_h[NSLOT];             // behind the copy of the slot's strip into the host frame
	bool timed[NSLOT]; bool timed_g2[NSLOT];
	float enqueue_us[NSLOT];            // host time spent inside pwn_tiled_submit for the slot's frame
	// frames: submitted (traced, group opened), blurred (blur enqueued), gathered (gather in a group), delivered
	unsigned long long submitted, blurred, gathered, delivered;
	pwn_tiled_info info;
	// PWN_DBG_TILED_PROF: host time per frame inside pwn_tiled_submit (the trace launch; the halo group; the blur; the gather group)
	// and pwn_tiled_wait (its events; the group's meeting), printed when the tiling goes
	bool prof; double p_trace, p_halo, p_blur, p_gather, p_events, p_meet, p_rest; unsigned long long p_frames;
};

#define GRP_HEAD(c) ((c) != NULL && (c)->grp != NULL && (c)->grp_head)
#define GRP_REFUSE(c) do { if(GRP_HEAD(c)) { snprintf((c)->err, sizeof((c)->err), "a group's handle runs its own tiling (pwn_init_multi)"); return PWN_ENOTSUP; } } while(0)
#define TPCHK(c, call) do { int rc_ = (call); if(rc_ != PWN_OK) { snprintf((c)->err, sizeof((c)->err), "%s transport: %s", \
	(c)->tiled->tp->name(), (c)->tiled->tp->err); return rc_; } } while(0)

static int init_timeout_ms(const pwn_ctx *c) { return c->tiled_init_ms > 0 ? c->tiled_init_ms : env_ms("PWN_TILED_INIT_TIMEOUT_MS", PWN_INIT_TIMEOUT_DEFAULT_MS); }
static int wait_timeout_ms(const pwn_ctx *c) { return c->tiled_wait_ms > 0 ? c->tiled_wait_ms : env_ms("PWN_TILED_WAIT_TIMEOUT_MS", PWN_WAIT_TIMEOUT_DEFAULT_MS); }
static bool rccl_mode_nonblocking(void)
{
	const char *e = getenv("PWN_TILED_RCCL_MODE");
	return e != NULL && (strcmp(e, "nonblocking") == 0 || strcmp(e, "nb") == 0);
}

extern "C" int pwn_tiled_set_timeouts(pwn_ctx *c, int init_ms, int wait_ms)
{
	if(GRP_HEAD(c)) return pwn_group_set_timeouts(c, init_ms, wait_ms);
	if(c == NULL) return PWN_EINVAL;
	if(init_ms != 0) c->tiled_init_ms = init_ms > 0 ? init_ms : 0;
	if(wait_ms != 0) c->tiled_wait_ms = wait_ms > 0 ? wait_ms : 0;
	if(c->tiled != NULL)
	{
		if(c->tiled->tp != NULL) c->tiled->tp->wait_ms = wait_timeout_ms(c);
		c->tiled->info.init_timeout_ms = init_timeout_ms(c); c->tiled->info.wait_timeout_ms = wait_timeout_ms(c);
	}
	return PWN_OK;
}

// What a first multi-GPU run wants on record before it starts (pwnhip.h).  No tiling needed.
extern "C" int pwn_tiled_preflight(pwn_ctx *c, char *json, size_t n)
{
	if(GRP_HEAD(c)) return pwn_tiled_preflight(pwn_group_member(c, 0), json, n);
	if(c == NULL || json == NULL || n < 2) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	int count = 0;
	if(hipGetDeviceCount(&count) != hipSuccess) count = 0;
	char peers[512]; size_t pl = 0;
	peers[0] = 0;
	for(int d = 0; d < count && pl + 8 < sizeof(peers); d++)
	{
		int can = d == c->device ? 1 : 0;
		if(d != c->device && hipDeviceCanAccessPeer(&can, c->device, d) != hipSuccess) { (void)hipGetLastError(); can = -1; }
		pl += (size_t)snprintf(peers + pl, sizeof(peers) - pl, "%s%d", d ? "," : "", can);
	}
	char err[200]; err[0] = 0;
	rccl_api *api = rccl_load(err, sizeof(err));
	// (what goes into the JSON text below as strings comes from the environment, the loader and the driver: quotes, backslashes and
	// control characters would end the string early -- in the very case the report exists for, a dlopen that failed)
	auto json_safe = [](char *dst, size_t n, const char *src)
	{
		size_t k = 0;
		for(; src != NULL && *src && k + 1 < n; src++) dst[k++] = (*src == '"' || *src == '\\' || (unsigned char)*src < 0x20) ? '?' : *src;
		dst[k] = 0;
	};
	char err_s[200], path_s[256], pci_s[32], ipc_s[64];
	json_safe(err_s, sizeof(err_s), err);
	json_safe(path_s, sizeof(path_s), api ? api->path : "");
	int ver = 0;
	if(api != NULL && api->GetVersion != NULL) (void)api->GetVersion(&ver);
	char pci[32]; pci[0] = 0;
	if(hipDeviceGetPCIBusId(pci, (int)sizeof(pci), c->device) != hipSuccess) { (void)hipGetLastError(); pci[0] = 0; }
	const char *ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
	json_safe(pci_s, sizeof(pci_s), pci);
	json_safe(ipc_s, sizeof(ipc_s), ipc);
	const int len = snprintf(json, n,
		"{\"device\": %d, \"pci\": \"%s\", \"devices_visible\": %d, \"can_access_peer\": [%s], "
		"\"librccl\": %s%s%s, \"rccl_version\": %d, \"rccl_has_nonblocking_api\": %s, \"rccl_has_abort\": %s, "
		"\"rccl_mode\": \"%s\", \"choreography\": \"%s\", \"init_timeout_ms\": %d, \"wait_timeout_ms\": %d, \"HSA_ENABLE_IPC_MODE_LEGACY\": %s%s%s%s%s%s}",
		c->device, pci_s, count, peers,
		api ? "\"" : "", api ? path_s : "null", api ? "\"" : "", ver,
		(api && api->CommInitRankConfig && api->CommGetAsyncError) ? "true" : "false", (api && api->CommAbort) ? "true" : "false",
		rccl_mode_nonblocking() ? "nonblocking" : "blocking", c->tiled_choreo == PWN_TILED_CHOREO_SPLIT ? "split" : "instream", init_timeout_ms(c), wait_timeout_ms(c),
		ipc ? "\"" : "", ipc ? ipc_s : "null", ipc ? "\"" : "",
		api ? "" : ", \"librccl_error\": \"", api ? "" : err_s, api ? "" : "\"");
	if(len < 0) return PWN_EINVAL;
	return len < (int)n ? len : (int)n - 1;
}

static int strip_rows(int h, int world)
{
	int per = (h + world - 1) / world;
	return (per + 7) / 8 * 8;
}

static void equal_cuts(int h, int world, int *cuts)
{
	const int per = strip_rows(h, world);
	for(int r = 0; r <= world; r++) cuts[r] = r * per < h ? r * per : h;
	cuts[world] = h;
}

// Cuts for the next frames from what the strips of a delivered frame cost.  The cost of a row is taken as constant
// inside a strip (cost[r] / rows of r), which makes the cost up to row y piecewise linear; the new cut k sits where
// that reaches k / world of the total, rounded to the 8 rows the kernels tile by.  Rows are not equally expensive
// inside a strip either (the horizon band is 3x the floor), so one step does not land on the balance point; a few
// re-cuts do, each from the costs measured with the cuts before.  Every strip keeps at least min_rows rows (the halo
// it has to send) and at most max_rows.  All ranks run this on the same numbers and get the same cuts.
static bool recut(const int *old, const uint32_t *cost, int world, int h, int min_rows, int max_rows, int *out)
{
	double total = 0.0, top = 0.0;
	for(int r = 0; r < world; r++)
	{
		if(cost[r] == 0u || old[r + 1] <= old[r]) return false;      // a rank that does not measure, an empty strip: leave it
		total += (double)cost[r];
		if((double)cost[r] > top) top = (double)cost[r];
	}
	if(top * world < total * 1.02) return false;                     // within 2 % of the mean already
	int cut[MAXW + 1];
	cut[0] = 0; cut[world] = h;
	double acc = 0.0;
	int r = 0;
	for(int k = 1; k < world; k++)
	{
		const double want = total * (double)k / (double)world;
		while(r < world - 1 && acc + (double)cost[r] < want) { acc += (double)cost[r]; r++; }
		double y = (double)old[r] + (want - acc) / (double)cost[r] * (double)(old[r + 1] - old[r]);
		// (three quarters of the way: the rows next to a cut are the ones whose cost the strip's mean misrepresents most)
		y = (double)old[k] + 0.75 * (y - (double)old[k]);
		int y8 = (int)(y / 8.0 + 0.5) * 8;
		cut[k] = y8;
	}
	// every strip at least min_rows and at most max_rows tall: push the cuts apart from the top, then from the bottom
	min_rows = (min_rows + 7) / 8 * 8; if(min_rows < 8) min_rows = 8;
	if((long long)min_rows * world > h || (long long)max_rows * world < h) return false;
	for(int k = 1; k < world; k++) { if(cut[k] < cut[k - 1] + min_rows) cut[k] = cut[k - 1] + min_rows; if(cut[k] > cut[k - 1] + max_rows) cut[k] = cut[k - 1] + max_rows; }
	for(int k = world - 1; k >= 1; k--) { if(cut[k] > cut[k + 1] - min_rows) cut[k] = (cut[k + 1] - min_rows) / 8 * 8; if(cut[k] < cut[k + 1] - max_rows) cut[k] = (cut[k + 1] - max_rows + 7) / 8 * 8; }
	for(int k = 1; k <= world; k++) if(cut[k] - cut[k - 1] < min_rows || cut[k] - cut[k - 1] > max_rows) return false;
	bool moved = false;
	for(int k = 0; k <= world; k++) { if(cut[k] != old[k]) moved = true; out[k] = cut[k]; }
	return moved;
}

// (the rule by itself, for hosts and tests that want to know what the library would do)
extern "C" int pwn_tiled_recut(const int *cuts, const uint32_t *cost, int world, int h, int min_rows, int max_rows, int *out)
{
	if(cuts == NULL || cost == NULL || out == NULL || world < 1 || world > MAXW || h < 1) return PWN_EINVAL;
	if(recut(cuts, cost, world, h, min_rows, max_rows, out)) return 1;
	memcpy(out, cuts, sizeof(int) * (size_t)(world + 1));
	return 0;
}

static bool valid_cuts(const pwn_tiled *t, int h, const int *cuts, int min_rows)
{
	if(cuts[0] != 0 || cuts[t->world] != h) return false;
	for(int r = 0; r < t->world; r++)
	{
		const int rows = cuts[r + 1] - cuts[r];
		if(rows < min_rows || rows < 1 || rows > t->max_rows) return false;
		if(r + 1 < t->world && (cuts[r + 1] & 7) != 0) return false;
	}
	return true;
}

extern "C" int pwn_tiled_unique_id(void *id, int transport)
{
	if(id == NULL) return PWN_EINVAL;
	memset(id, 0, PWN_TILED_ID_BYTES);
	if(transport == PWN_TRANSPORT_SHM)
	{
		struct timespec ts;
		clock_gettime(CLOCK_REALTIME, &ts);
		snprintf((char *)id, PWN_TILED_ID_BYTES, "/pwn_tiled_%d_%lx", (int)getpid(), (unsigned long)ts.tv_nsec);
		return PWN_OK;
	}
	if(transport != PWN_TRANSPORT_RCCL) return PWN_EINVAL;
	char err[200];
	rccl_api *api = rccl_load(err, sizeof(err));
	if(api == NULL) return PWN_ENOTSUP;
	static_assert(NSLOT == PWN_TILED_SLOTS, "the host sink holds one frame per buffer set");
	static_assert(PWN_MAX_SLOTS < NSLOT, "a group names a depth plane per frame slot of its host and one more for its blocking calls");
	static_assert(sizeof(ncclUniqueId) == PWN_TILED_ID_BYTES, "the id is an ncclUniqueId");
	return api->GetUniqueId((ncclUniqueId *)id) == ncclSuccess ? PWN_OK : PWN_EHIP;
}

void pwn_tiled_destroy(pwn_ctx *c)
{
	pwn_tiled *t = c->tiled;
	if(t == NULL) return;
	if(t->prof && t->p_frames)
		fprintf(stderr, "tiling rank %d of %d, %llu frames, host us per frame: submit = trace launch %.1f + halo group %.1f + blur %.1f + gather group %.1f; wait = events %.1f + meeting %.1f + rest %.1f\n",
			t->rank, t->world, t->p_frames, t->p_trace / t->p_frames, t->p_halo / t->p_frames, t->p_blur / t->p_frames, t->p_gather / t->p_frames,
			t->p_events / t->p_frames, t->p_meet / t->p_frames, t->p_rest / t->p_frames);
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	// (the device is idle: no copy of the tables is in use, and the events that said so -- ev_t, handed to the trace
	// launches as pwn_ctx.trace_tables_event -- are destroyed below)
	for(int i = 0; i < PWN_NBLOB; i++) c->tables_in_use[i] = false;
	pwn_launch_history_clear(c);
	delete t->tp;
	for(int s = 0; s < NSLOT; s++)
	{
		(void)hipFree(t->pre[s]); (void)hipFree(t->out[s]); (void)hipFree(t->fin[s]); (void)hipFree(t->z[s]);
		(void)hipFree(t->missw[s]); (void)hipFree(t->missv[s]);
		hipEvent_t *evs[] = { &t->ev_t[s], &t->ev_x[s], &t->ev_b[s], &t->ev_d[s], &t->ev_k0[s], &t->ev_k1[s], &t->ev_k2[s], &t->ev_k3[s],
			&t->ev_g0[s], &t->ev_g1[s], &t->ev_g2[s], &t->ev_g3[s], &t->ev_h[s] };
		for(size_t i = 0; i < sizeof(evs) / sizeof(evs[0]); i++) if(*evs[i]) (void)hipEventDestroy(*evs[i]);
	}
	(void)hipFree(t->cost_acc); (void)hipFree(t->self_buf);
	for(int i = 0; i < PWN_MAX_SLOTS; i++) (void)hipFree(t->dsurf[i]);
	if(t->copy) (void)hipStreamDestroy(t->copy);
	if(t->host_registered) (void)hipHostUnregister(t->host_registered);
	if(t->h_missv) (void)hipHostFree(t->h_missv);
	if(t->h_frame) (void)hipHostFree(t->h_frame);
	if(t->comm) (void)hipStreamDestroy(t->comm);
	if(t->cs3) (void)hipStreamDestroy(t->cs3);
	delete t;
	c->tiled = NULL;
	c->grid_reserve = 0;
	(void)pwn_i_set_launch_rotation(c, 2);
}

extern "C" void pwn_tiled_shutdown(pwn_ctx *c) { if(c != NULL && !GRP_HEAD(c)) pwn_tiled_destroy(c); }

bool pwn_tiled_busy(pwn_ctx *c) { return c->tiled != NULL && c->tiled->submitted != c->tiled->delivered; }

extern "C" int pwn_tiled_init(pwn_ctx *c, int rank, int world, const void *id, int transport, int halo_rows)
{
	GRP_REFUSE(c);
	if(c == NULL || id == NULL || world < 1 || world > MAXW || rank < 0 || rank >= world) return PWN_EINVAL;
	if(c->tiled != NULL) return PWN_EBUSY;
	for(int i = 0; i < c->nslots; i++) if(c->slot[i].in_flight) return PWN_EBUSY;
	if(c->blur_passes > 1) { snprintf(c->err, sizeof(c->err), "row tiling supports POSTPROC_BLUR 0 or 1"); return PWN_EINVAL; }
	if(c->blur_passes > 0 && (c->w & 3) != 0) return PWN_EINVAL;
	(void)hipSetDevice(c->device);
	// (frames of the frames-in-flight API may have used the second compute stream)
	if(c->stream2 && hipStreamSynchronize(c->stream2) != hipSuccess) return PWN_EHIP;
	c->last_frame_done = NULL;
	pwn_tiled *t = new(std::nothrow) pwn_tiled();
	if(t == NULL) return PWN_ENOMEM;
	memset(t, 0, sizeof(*t));
	t->rank = rank; t->world = world;
	t->per = strip_rows(c->h, world);
	equal_cuts(c->h, world, t->cuts);
	for(int s = 0; s < NSLOT; s++) memcpy(t->fcuts[s], t->cuts, sizeof(t->cuts));
	// a rank's strip may grow to one and a half equal strips when the cuts move
	t->max_rows = (t->per + t->per / 2 + 7) / 8 * 8;
	if(t->max_rows > c->h) t->max_rows = c->h;
	// bounded halo: H rows per neighbour, possible when every strip has at least H rows; the default
	// covers depth 24 (taps reach 0.002 * h * (depth - 1) rows, screen.h:100-102)
	int H = halo_rows < 0 ? (int)(0.002 * c->h * 24.0) + 2 : halo_rows;
	int shortest = c->h;
	for(int r = 0; r < world; r++) if(t->cuts[r + 1] - t->cuts[r] < shortest) shortest = t->cuts[r + 1] - t->cuts[r];
	if(world == 1 || c->blur_passes == 0 || H > shortest || H <= 0) H = 0;
	t->halo = H; t->want_halo = H;
	// moving cuts: on unless a strip of the equal split is empty (more ranks than 8-row bands); PWN_TILED_BALANCE=k
	// sets the period in delivered frames, 0 switches it off; pwn_tiled_balance() does the same from the host
	t->balance_every = (world > 1 && shortest >= 16 && c->blur_passes == 1) ? 8 : 0;
	if(const char *e = getenv("PWN_TILED_BALANCE")) { int v = atoi(e); if(v >= 0 && v <= 100000 && (v == 0 || t->balance_every > 0)) t->balance_every = v; }
	t->cs[0] = c->stream; t->cs[1] = (c->frame_overlap && c->stream2 != NULL) ? c->stream2 : c->stream;
	t->ncs = t->cs[1] != t->cs[0] ? 2 : 1;
	t->instream = c->tiled_choreo != PWN_TILED_CHOREO_SPLIT;
	t->prof = getenv("PWN_DBG_TILED_PROF") != NULL;
	c->tiled = t;

	int rc = PWN_OK;
	const size_t n = (size_t)c->w * (size_t)c->h;
	const size_t strip_bytes = (size_t)c->w * (size_t)t->max_rows * 4;
	do
	{
		if(transport == PWN_TRANSPORT_RCCL)
		{
			rccl_api *api = rccl_load(c->err, sizeof(c->err));
			if(api == NULL) { rc = PWN_ENOTSUP; break; }
			const bool nb = rccl_mode_nonblocking();
			if(nb && (api->CommInitRankConfig == NULL || api->CommGetAsyncError == NULL))
			{
				snprintf(c->err, sizeof(c->err), "PWN_TILED_RCCL_MODE=nonblocking: this librccl (%s) has no ncclCommInitRankConfig / ncclCommGetAsyncError", api->path);
				rc = PWN_ENOTSUP; break;
			}
			rccl_transport *rt = new(std::nothrow) rccl_transport();
			if(rt == NULL) { rc = PWN_ENOMEM; break; }
			rt->api = api; rt->nb = nb; rt->rank = rank; rt->world = world; rt->wait_ms = wait_timeout_ms(c);
			t->tp = rt;
			ncclUniqueId uid;
			memcpy(&uid, id, sizeof(uid));
			// the communicator and one word to and from every other rank, under the bring-up deadline
			rc = rccl_bringup_bounded(api, c->device, uid, world, rank, nb, init_timeout_ms(c), &rt->comm, c->err, sizeof(c->err));
			if(rc != PWN_OK) { rt->dead = true; break; }
			rt->comms[0] = rt->comm; rt->keys[0] = t->cs[0]; rt->ncomm = 1;
			t->info.rccl_nonblocking = nb ? 1 : 0;
		}
		else if(transport == PWN_TRANSPORT_SHM)
		{
			shm_transport *st = new(std::nothrow) shm_transport();
			if(st == NULL) { rc = PWN_ENOMEM; break; }
			t->tp = st;
			st->wait_ms = wait_timeout_ms(c);
			char name[PWN_TILED_ID_BYTES];
			memcpy(name, id, sizeof(name)); name[sizeof(name) - 1] = 0;
			rc = st->open_region(name, rank, world, strip_bytes);
			if(rc != PWN_OK) { snprintf(c->err, sizeof(c->err), "%s", st->err); break; }
		}
		else if(transport == PWN_TRANSPORT_LOCAL)
		{
			// (the id is the group's: only members of a group have a hub)
			if(c->hub == NULL || c->hub->world != world) { snprintf(c->err, sizeof(c->err), "PWN_TRANSPORT_LOCAL is the transport of a group's members (pwn_init_multi)"); rc = PWN_EINVAL; break; }
			local_transport *lt = new(std::nothrow) local_transport();
			if(lt == NULL) { rc = PWN_ENOMEM; break; }
			t->tp = lt;
			lt->wait_ms = wait_timeout_ms(c);
			rc = lt->open(c->hub, rank, world, c->device);
			if(rc != PWN_OK) { snprintf(c->err, sizeof(c->err), "%s", lt->err); break; }
		}
		else { rc = PWN_EINVAL; break; }
		t->hub = c->hub;

		// the exchange on a hardware queue of its own (queues are pooled per priority level: pwn_init on stream2) -- on a
		// queue shared with a compute stream the transport's kernels would wait behind that stream's trace launches
		{
			int least = 0, greatest = 0;
			(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
			if(hipStreamCreateWithPriority(&t->comm, hipStreamNonBlocking, greatest < 0 ? greatest : -1) != hipSuccess)
			{
				(void)hipGetLastError();
				if(hipStreamCreateWithFlags(&t->comm, hipStreamNonBlocking) != hipSuccess) { rc = PWN_EHIP; break; }
			}
		}
		// a third compute stream (in-stream choreography): frame f on cs[f mod 3], so that two frames' exchanges and blurs
		// overlap a third frame's trace.  On the hardware-queue pool of stream2 and comm (pwn_init)
		if(t->instream && t->ncs == 2 && c->tiled_streams == 3)
		{
			int least = 0, greatest = 0;
			(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
			if(hipStreamCreateWithPriority(&t->cs3, hipStreamNonBlocking, greatest < 0 ? greatest : -1) != hipSuccess)
			{
				(void)hipGetLastError();
				if(hipStreamCreateWithFlags(&t->cs3, hipStreamNonBlocking) != hipSuccess) { rc = PWN_EHIP; break; }
			}
			t->cs[2] = t->cs3; t->ncs = 3;
		}
		// PWN_OPT_TILED_COMMS: a communicator per compute stream (in-stream choreography, blocking mode).  Each comes up like the
		// first -- ncclCommInitRank and a word with every peer under the bring-up deadline -- with an id that rank 0 sends round
		// over the first.
		if(transport == PWN_TRANSPORT_RCCL && t->instream && t->ncs > 1 && c->tiled_comms == PWN_TILED_COMMS_PER_STREAM && !rccl_mode_nonblocking())
		{
			rccl_transport *rt = (rccl_transport *)t->tp;
			for(int k = 1; k < t->ncs && rc == PWN_OK; k++)
			{
				ncclUniqueId id2;
				memset(&id2, 0, sizeof(id2));
				c->err[0] = 0;
				rc = rccl_share_id(rt->api, rt->comms[0], world, rank, &id2, init_timeout_ms(c), c->err, sizeof(c->err));
				if(rc != PWN_OK) break;
				ncclComm_t cm = NULL;
				rc = rccl_bringup_bounded(rt->api, c->device, id2, world, rank, false, init_timeout_ms(c), &cm, c->err, sizeof(c->err));
				if(rc != PWN_OK) break;
				rt->comms[k] = cm; rt->keys[k] = t->cs[k]; rt->ncomm = k + 1;
			}
			if(rc != PWN_OK) { rt->abort(); break; }
		}
		rc = pwn_i_set_launch_rotation(c, t->ncs == 3 ? 3 : 2);
		if(rc != PWN_OK) break;
		if(hipMalloc((void **)&t->cost_acc, 64 * NSLOT) != hipSuccess) { rc = PWN_ENOMEM; break; }
		if(hipMemset(t->cost_acc, 0, 64 * NSLOT) != hipSuccess) { rc = PWN_EHIP; break; }
		for(int s = 0; s < NSLOT && rc == PWN_OK; s++)
		{
			if(hipMalloc((void **)&t->pre[s], n * 4) != hipSuccess || hipMalloc((void **)&t->out[s], n * 4) != hipSuccess ||
			   hipMalloc((void **)&t->z[s], n * 4) != hipSuccess || hipMalloc((void **)&t->missw[s], 64) != hipSuccess ||
			   hipMalloc((void **)&t->missv[s], (size_t)world * 8 + 64) != hipSuccess ||
			   (rank == 0 && hipMalloc((void **)&t->fin[s], n * 4) != hipSuccess)) { rc = PWN_ENOMEM; break; }
			if(hipMemset(t->pre[s], 0, n * 4) != hipSuccess || hipMemset(t->out[s], 0, n * 4) != hipSuccess ||
			   hipMemset(t->z[s], 0, n * 4) != hipSuccess || hipMemset(t->missw[s], 0, 64) != hipSuccess ||
			   hipMemset(t->missv[s], 0, (size_t)world * 8 + 64) != hipSuccess ||
			   (rank == 0 && hipMemset(t->fin[s], 0, n * 4) != hipSuccess)) { rc = PWN_EHIP; break; }
			if(hipEventCreateWithFlags(&t->ev_t[s], hipEventDisableTiming) != hipSuccess ||
			   hipEventCreateWithFlags(&t->ev_x[s], hipEventDisableTiming) != hipSuccess ||
			   hipEventCreateWithFlags(&t->ev_b[s], hipEventDisableTiming) != hipSuccess ||
			   hipEventCreateWithFlags(&t->ev_d[s], hipEventDisableTiming) != hipSuccess ||
			   hipEventCreate(&t->ev_k0[s]) != hipSuccess || hipEventCreate(&t->ev_k1[s]) != hipSuccess ||
			   hipEventCreate(&t->ev_k2[s]) != hipSuccess || hipEventCreate(&t->ev_k3[s]) != hipSuccess ||
			   hipEventCreate(&t->ev_g0[s]) != hipSuccess || hipEventCreate(&t->ev_g1[s]) != hipSuccess ||
			   hipEventCreate(&t->ev_g2[s]) != hipSuccess || hipEventCreate(&t->ev_g3[s]) != hipSuccess) { rc = PWN_EHIP; break; }
		}
		if(rc != PWN_OK) break;
		if(hipHostMalloc((void **)&t->h_missv, (size_t)NSLOT * ((size_t)world + 1) * 8, hipHostMallocDefault) != hipSuccess) { rc = PWN_ENOMEM; break; }
		memset(t->h_missv, 0, (size_t)NSLOT * ((size_t)world + 1) * 8);
		// one round trip through the transport before the first frame depends on it: every rank sends a
		// word to its right-hand neighbour (to itself when alone) and checks what arrives from the left
		// (the RCCL bring-up above has exchanged a word with EVERY rank already)
		if(transport != PWN_TRANSPORT_RCCL)
		{
			const int to = (rank + 1) % world, from = (rank + world - 1) % world;
			const uint32_t mine = 0x50574e00u + (uint32_t)rank, want = 0x50574e00u + (uint32_t)from;
			uint32_t got = 0;
			if(hipMemcpy(t->missw[0], &mine, 4, hipMemcpyHostToDevice) != hipSuccess) { rc = PWN_EHIP; break; }
			int trc = t->tp->begin(t->comm);
			if(trc == PWN_OK) trc = t->tp->send(t->missw[0], 4, to);
			if(trc == PWN_OK) trc = t->tp->recv(t->missv[0], 4, from);
			if(trc == PWN_OK) trc = t->tp->end();
			if(trc != PWN_OK)
			{
				// (a peer that is not there: PWN_ETIMEDOUT after the wait deadline, with the rank in the text)
				snprintf(c->err, sizeof(c->err), "%s transport: %s", t->tp->name(), t->tp->err); rc = trc == PWN_ETIMEDOUT ? trc : PWN_EHIP; break;
			}
			if(hipStreamSynchronize(t->comm) != hipSuccess || hipMemcpy(&got, t->missv[0], 4, hipMemcpyDeviceToHost) != hipSuccess ||
			   hipMemset(t->missw[0], 0, 4) != hipSuccess || hipMemset(t->missv[0], 0, 4) != hipSuccess) { rc = PWN_EHIP; break; }
			if(got != want)
			{
				snprintf(c->err, sizeof(c->err), "%s transport: the test word from rank %d arrived as %08x", t->tp->name(), from, got);
				rc = PWN_EHIP; break;
			}
		}
	} while(0);
	// (the memsets above ran on the null stream, which the non-blocking streams of the frames do not wait for)
	if(rc == PWN_OK && hipDeviceSynchronize() != hipSuccess) rc = PWN_EHIP;
	if(rc != PWN_OK) { char keep[256]; memcpy(keep, c->err, sizeof(keep)); pwn_tiled_destroy(c); memcpy(c->err, keep, sizeof(keep)); return rc; }
	// Room for RCCL's kernels beside the persistent trace grid (pwn_api.cpp): 16 workgroups of ~1280, i.e. one less
	// on 16 CUs, 1.25 % of the grid.  Not measurable without several GPUs; PWN_TILED_RESERVE=n overrides it
	// (0 = fill every CU), pwn_tiled_set_reserve() changes it between frames: bench.py --gpus N sweeps it.
	c->grid_reserve = 0;
	if(world == 1 && transport == PWN_TRANSPORT_RCCL && getenv("PWN_TILED_SELF") != NULL && atoi(getenv("PWN_TILED_SELF")) != 0)
	{
		if(hipMalloc((void **)&t->self_buf, n * 4 + 64) != hipSuccess) { pwn_tiled_destroy(c); return PWN_ENOMEM; }
		t->self_exchange = true;
	}
	if((world > 1 || t->self_exchange) && transport == PWN_TRANSPORT_RCCL)
	{
		c->grid_reserve = 16;
		if(const char *e = getenv("PWN_TILED_RESERVE")) { int v = atoi(e); if(v >= 0 && v <= 512) c->grid_reserve = v; }
	}
	t->info.rank = rank; t->info.world = world; t->info.y0 = t->cuts[rank]; t->info.y1 = t->cuts[rank + 1]; t->info.rows_per_rank = t->per;
	t->info.halo_rows = t->halo; t->info.transport = transport;
	t->info.max_rows = t->max_rows; t->info.grid_reserve = c->grid_reserve; t->info.two_streams = t->ncs > 1; t->info.compute_streams = t->ncs; t->info.choreography = t->instream ? PWN_TILED_CHOREO_INSTREAM : PWN_TILED_CHOREO_SPLIT;
	t->info.communicators = transport == PWN_TRANSPORT_RCCL ? ((rccl_transport *)t->tp)->ncomm : 0;
	t->info.init_timeout_ms = init_timeout_ms(c); t->info.wait_timeout_ms = wait_timeout_ms(c);
	return PWN_OK;
}

extern "C" int pwn_tiled_balance(pwn_ctx *c, int every_frames)
{
	GRP_REFUSE(c);
	if(c == NULL || c->tiled == NULL || every_frames < 0) return PWN_EINVAL;
	c->tiled->balance_every = every_frames;
	return PWN_OK;
}

extern "C" int pwn_tiled_set_reserve(pwn_ctx *c, int workgroups)
{
	GRP_REFUSE(c);
	if(c == NULL || c->tiled == NULL || workgroups < 0 || workgroups > 512) return PWN_EINVAL;
	c->grid_reserve = workgroups;
	c->tiled->info.grid_reserve = workgroups;
	return PWN_OK;
}

extern "C" int pwn_tiled_gather_root(pwn_ctx *c, int mode)
{
	GRP_REFUSE(c);
	if(c == NULL || c->tiled == NULL || (mode != PWN_TILED_ROOT_FIXED && mode != PWN_TILED_ROOT_ROTATE)) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	if(t->submitted != t->delivered) return PWN_EBUSY;
	if(mode == PWN_TILED_ROOT_ROTATE && !t->sink)
	{
		// every rank is the root of some frames: the buffers a root assembles frames in
		(void)hipSetDevice(c->device);
		const size_t n = (size_t)c->w * (size_t)c->h;
		for(int s = 0; s < NSLOT; s++)
			if(t->fin[s] == NULL)
			{
				if(hipMalloc((void **)&t->fin[s], n * 4) != hipSuccess) return PWN_ENOMEM;
				if(hipMemset(t->fin[s], 0, n * 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return PWN_EHIP;
			}
	}
	t->root_mode = mode;
	t->info.gather_root = mode;
	return PWN_OK;
}

extern "C" int pwn_tiled_set_cuts(pwn_ctx *c, const int *cuts, int n)
{
	GRP_REFUSE(c);
	if(c == NULL || c->tiled == NULL || cuts == NULL) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	if(n != t->world + 1) return PWN_EINVAL;
	// (a strip shorter than the halo it has to send would mean whole strips from now on: not through this call)
	if(!valid_cuts(t, c->h, cuts, t->halo > 0 ? t->halo : 1))
	{
		snprintf(c->err, sizeof(c->err), "cuts: 0 = c[0] < ... < c[%d] = %d in multiples of 8, every strip %d..%d rows", t->world, c->h,
			t->halo > 0 ? t->halo : 1, t->max_rows);
		return PWN_EINVAL;
	}
	memcpy(t->cuts, cuts, sizeof(int) * (size_t)n);
	t->info.y0 = t->cuts[t->rank]; t->info.y1 = t->cuts[t->rank + 1];
	return PWN_OK;
}

extern "C" int pwn_tiled_get_cuts(pwn_ctx *c, int *cuts, uint32_t *cost)
{
	GRP_REFUSE(c);
	if(c == NULL || c->tiled == NULL || cuts == NULL) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	memcpy(cuts, t->cuts, sizeof(int) * (size_t)(t->world + 1));
	if(cost != NULL) memcpy(cost, t->last_cost, sizeof(uint32_t) * (size_t)t->world);
	return t->world + 1;
}

extern "C" int pwn_tiled_host_sink(pwn_ctx *c, void *base, size_t bytes)
{
	GRP_REFUSE(c);
	if(c == NULL || c->tiled == NULL || base == NULL) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	if(t->submitted != 0 || t->host_base != NULL) return PWN_EBUSY;
	const size_t frame = (size_t)c->w * (size_t)c->h * 4;
	if(bytes < frame * NSLOT) { snprintf(c->err, sizeof(c->err), "a host sink holds %d frames: %zu bytes", NSLOT, frame * NSLOT); return PWN_EINVAL; }
	(void)hipSetDevice(c->device);
	// memory of the caller (shared between the rank processes): make it known to this device.  Already pinned by
	// this process (hipHostMalloc in a one-process test) is fine too.
	hipError_t e = hipHostRegister(base, frame * NSLOT, hipHostRegisterDefault);
	if(e == hipSuccess) t->host_registered = base;
	else if(e == hipErrorHostMemoryAlreadyRegistered) (void)hipGetLastError();
	else { snprintf(c->err, sizeof(c->err), "hipHostRegister: %s", hipGetErrorString(e)); return PWN_EHIP; }
	if(hipStreamCreateWithFlags(&t->copy, hipStreamNonBlocking) != hipSuccess) return PWN_EHIP;
	for(int s = 0; s < NSLOT; s++) if(hipEventCreateWithFlags(&t->ev_h[s], hipEventDisableTiming) != hipSuccess) return PWN_EHIP;
	t->host_base = (uint8_t *)base;
	t->sink = true;
	t->info.host_sink = 1;
	return PWN_OK;
}

// ... and, with the sink of screen.h:126-149 wanted, upscales on every member: the device planes, zeroed (bytes between the rows of a
// wider pitch read 0).  Behind pwn_i_tiled_sink, before the first frame.
int pwn_i_tiled_surface(pwn_ctx *c, int scale, int pitch_bytes)
{
	if(c == NULL || c->tiled == NULL || scale < 1 || (pitch_bytes & 3) != 0 || (long long)pitch_bytes < (long long)c->w * scale * 4) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	if(t->submitted != 0 || !t->sink) return PWN_EBUSY;
	(void)hipSetDevice(c->device);
	const size_t bytes = (size_t)pitch_bytes * (size_t)c->h * (size_t)scale;
	for(int i = 0; i < PWN_MAX_SLOTS; i++)
	{
		if(hipMalloc((void **)&t->dsurf[i], bytes) != hipSuccess) return PWN_ENOMEM;
		if(hipMemset(t->dsurf[i], 0, bytes) != hipSuccess) return PWN_EHIP;
	}
	if(hipDeviceSynchronize() != hipSuccess) return PWN_EHIP;
	t->surf_scale = scale; t->surf_pitch_words = pitch_bytes / 4;
	return PWN_OK;
}

// A group's tiling delivers to the host too, into the buffers handed over with every frame (pwn_i_tiled_submit): every
// member copies its strip there over its own PCIe link.  Before the first frame.
int pwn_i_tiled_sink(pwn_ctx *c)
{
	if(c == NULL || c->tiled == NULL) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	if(t->submitted != 0 || t->sink) return PWN_EBUSY;
	(void)hipSetDevice(c->device);
	if(hipStreamCreateWithFlags(&t->copy, hipStreamNonBlocking) != hipSuccess) return PWN_EHIP;
	for(int s = 0; s < NSLOT; s++) if(hipEventCreateWithFlags(&t->ev_h[s], hipEventDisableTiming) != hipSuccess) return PWN_EHIP;
	t->sink = true;
	t->info.host_sink = 1;
	return PWN_OK;
}

extern "C" int pwn_tiled_get_info(pwn_ctx *c, pwn_tiled_info *out)
{
	GRP_REFUSE(c);
	if(c == NULL || out == NULL || c->tiled == NULL) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	t->info.halo_rows = t->halo;
	t->info.y0 = t->cuts[t->rank]; t->info.y1 = t->cuts[t->rank + 1];
	t->info.balance_every = t->balance_every;
	t->info.dead = (t->tp != NULL && t->tp->dead) ? 1 : 0;
	*out = t->info;
	return PWN_OK;
}

// strip r of the frame in slot s
static inline void rows_of(const pwn_tiled *t, int s, int r, int *a, int *b) { *a = t->fcuts[s][r]; *b = t->fcuts[s][r + 1]; }

// the two words of every rank to every rank
static int add_words(pwn_ctx *c, pwn_tiled *t, int s)
{
	// (a group's members read each other's words from pinned memory when they meet in pwn_tiled_wait)
	if(t->hub != NULL) return PWN_OK;
	for(int r = 0; r < t->world; r++)
	{
		if(r == t->rank) continue;
		TPCHK(c, t->tp->send(t->missw[s], 8, r));
		TPCHK(c, t->tp->recv(t->missv[s] + 2 * r, 8, r));
	}
	return PWN_OK;
}

// the second half of a group: frame `g`'s finished strips to rank 0, its words to everybody
static int add_gather(pwn_ctx *c, pwn_tiled *t, unsigned long long g)
{
	const int s = (int)(g % NSLOT);
	const size_t w4 = (size_t)c->w * 4;
	uint32_t *mine = c->blur_passes ? t->out[s] : t->pre[s];
	int y0, y1; rows_of(t, s, t->rank, &y0, &y1);
	// host sink: no strips; the words go out behind this rank's copy to the host
	if(t->sink) return add_words(c, t, s);
	const int root = t->froot[s];
	if(t->rank == root)
	{
		for(int r = 0; r < t->world; r++)
		{
			if(r == root) continue;
			int a, b; rows_of(t, s, r, &a, &b);
			if(b > a) { TPCHK(c, t->tp->recv(t->fin[s] + (size_t)a * c->w, (size_t)(b - a) * w4, r)); t->info.bytes_received += (unsigned long long)(b - a) * w4; }
		}
	}
	else if(y1 > y0)
	{
		TPCHK(c, t->tp->send(mine + (size_t)y0 * c->w, (size_t)(y1 - y0) * w4, root));
		t->info.bytes_sent += (unsigned long long)(y1 - y0) * w4;
	}
	return add_words(c, t, s);
}

// PWN_TILED_SELF: what a rank of a real tiling puts into the gather group, to itself
static int add_self_gather(pwn_ctx *c, pwn_tiled *t, unsigned long long g)
{
	const int s = (int)(g % NSLOT);
	const size_t w4 = (size_t)c->w * 4;
	const uint32_t *mine = t->fin[s];          // (one rank is its own root: blur -- or, without one, the trace -- wrote straight into the assembled frame)
	int y0, y1; rows_of(t, s, 0, &y0, &y1);
	TPCHK(c, t->tp->send(mine + (size_t)y0 * c->w, (size_t)(y1 - y0) * w4, 0));
	TPCHK(c, t->tp->recv(t->self_buf, (size_t)(y1 - y0) * w4, 0));
	TPCHK(c, t->tp->send(t->missw[s], 8, 0));
	TPCHK(c, t->tp->recv(t->missv[s], 8, 0));
	t->info.bytes_sent += (unsigned long long)(y1 - y0) * w4 + 8; t->info.bytes_received += (unsigned long long)(y1 - y0) * w4 + 8;
	return PWN_OK;
}

// behind a group that carried frame g's words: bring them (and this rank's own) to pinned host memory on the comm
// stream, so that pwn_tiled_wait reads them there instead of making two blocking copies per frame.  By a kernel that
// stores into the pinned buffer (one launch instead of two asynchronous copies of 8 (x world) bytes).
static int fetch_words(pwn_ctx *c, pwn_tiled *t, unsigned long long g, hipStream_t xs)
{
	const int s = (int)(g % NSLOT);
	uint32_t *h = t->h_missv + (size_t)s * ((size_t)t->world + 1) * 2;
	HIPCHK(c, pwn_launch_words(t->missv[s], t->missw[s], h, t->world, xs));
	return PWN_OK;
}

// whole strips of pre[s] to everybody (an all-gather by send / recv, in place in the full-frame plane)
static int add_allgather(pwn_ctx *c, pwn_tiled *t, int s)
{
	const size_t w4 = (size_t)c->w * 4;
	int y0, y1; rows_of(t, s, t->rank, &y0, &y1);
	for(int r = 0; r < t->world; r++)
	{
		if(r == t->rank) continue;
		int a, b; rows_of(t, s, r, &a, &b);
		if(y1 > y0) { TPCHK(c, t->tp->send(t->pre[s] + (size_t)y0 * c->w, (size_t)(y1 - y0) * w4, r)); t->info.bytes_sent += (unsigned long long)(y1 - y0) * w4; }
		if(b > a) { TPCHK(c, t->tp->recv(t->pre[s] + (size_t)a * c->w, (size_t)(b - a) * w4, r)); t->info.bytes_received += (unsigned long long)(b - a) * w4; }
	}
	return PWN_OK;
}

// host sink: this rank's finished strip of the slot's frame into the host frame, behind its blur
static int copy_strip_to_host(pwn_ctx *c, pwn_tiled *t, int s)
{
	int y0, y1; rows_of(t, s, t->rank, &y0, &y1);
	if(y1 <= y0) { HIPCHK(c, hipEventRecord(t->ev_h[s], t->copy)); return PWN_OK; }
	const size_t w4 = (size_t)c->w * 4, frame = w4 * (size_t)c->h;
	const uint32_t *src = c->blur_passes ? t->out[s] : t->pre[s];
	HIPCHK(c, hipStreamWaitEvent(t->copy, t->ev_b[s], 0));
	uint8_t *frame_at = t->hdst[s] != NULL ? (uint8_t *)t->hdst[s] : t->host_base + (size_t)s * frame;
	if(t->hzdst[s] != NULL)          // (zbuf, main.c:33: the depth strip as the trace left it)
	{
		HIPCHK(c, hipMemcpyAsync(t->hzdst[s] + (size_t)y0 * c->w, t->fz[s] + (size_t)y0 * c->w, (size_t)(y1 - y0) * w4, hipMemcpyDeviceToHost, t->copy));
		t->info.bytes_to_host += (unsigned long long)(y1 - y0) * w4;
	}
	HIPCHK(c, hipMemcpyAsync(frame_at + (size_t)y0 * w4, src + (size_t)y0 * c->w,
		(size_t)(y1 - y0) * w4, hipMemcpyDeviceToHost, t->copy));
	if(t->hsurf[s] != NULL)
	{
		// screen_upscale of this strip (the kernel of pwn_screen_upscale on the strip's rows: a source row's words start at row * rowadv),
		// then its words to the host surface -- on the copy stream, behind the strip's colour
		const size_t rowadv = (size_t)c->w * (size_t)t->surf_scale + (size_t)t->surf_pitch_words * (size_t)(t->surf_scale - 1);
		HIPCHK(c, pwn_launch_upscale(src + (size_t)y0 * c->w, t->fsurf[s] + (size_t)y0 * rowadv, c->w, y1 - y0, t->surf_scale, t->surf_pitch_words, t->copy));
		HIPCHK(c, hipMemcpyAsync(t->hsurf[s] + (size_t)y0 * rowadv, t->fsurf[s] + (size_t)y0 * rowadv, (size_t)(y1 - y0) * rowadv * 4, hipMemcpyDeviceToHost, t->copy));
		t->info.bytes_to_host += (unsigned long long)(y1 - y0) * rowadv * 4ull;
	}
	HIPCHK(c, hipEventRecord(t->ev_h[s], t->copy));
	t->info.bytes_to_host += (unsigned long long)(y1 - y0) * w4;
	return PWN_OK;
}

// the blur of frame k on that frame's compute stream, behind the group that brought its halo rows
static int enqueue_blur(pwn_ctx *c, pwn_tiled *t, unsigned long long k)
{
	const int s = (int)(k % NSLOT);
	hipStream_t cs = t->fstream[s];
	int y0, y1; rows_of(t, s, t->rank, &y0, &y1);
	if(c->blur_passes)
	{
		if(!t->instream) HIPCHK(c, hipStreamWaitEvent(cs, t->ev_x[s], 0));          // (in-stream: the halo rows came in front of this on cs)
		if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_k2[s], cs));
		uint32_t *dst = (t->rank == t->froot[s] && !t->sink) ? t->fin[s] : t->out[s];
		// the trace of this frame, in front of this launch on the stream, added up what the strip cost: the blur
		// moves that into the frame's second word and clears the accumulator for the stream's next trace
		uint32_t *acc = t->cost_acc + 16 * s;
		int rc;
		c->blur_cost_mul = t->fcost_mul[s]; c->blur_cost_div = t->fcost_div[s];
		if(t->fhalo[s])
		{
			const int H = t->fhalo[s];
			const int a0 = t->rank > 0 ? y0 - H : 0, a1 = (t->rank < t->world - 1 && y1 < c->h) ? y1 + H : c->h;
			rc = pwn_i_launch_blur(c, y0, y1, t->pre[s], t->fz[s], dst, cs, a0, a1, t->missw[s], acc, t->missw[s] + 1);
		}
		else rc = pwn_i_launch_blur(c, y0, y1, t->pre[s], t->fz[s], dst, cs, 0, 0, NULL, acc, t->missw[s] + 1);
		if(rc != PWN_OK) return rc;
		if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_k3[s], cs));
	}
	if(!t->instream || t->sink) HIPCHK(c, hipEventRecord(t->ev_b[s], cs));       // (in-stream: what follows the blur follows it on cs; only the copy to the host waits for it)
	// PWN_OPT_UNIT_ORDER: the strip's unit costs (written by the frame's trace, in front of this on the stream) sorted into
	// the order of the stream's next trace of the same rows; behind ev_b, which is what the exchange waits for
	{
		const int rc = pwn_i_launch_order(c, cs);
		if(rc != PWN_OK) return rc;
	}
	if(t->sink) return copy_strip_to_host(c, t, s);
	return PWN_OK;
}

// G2 of the frames [from, to) as ONE grouped launch on `gs` (behind their blurs -- and copies to the host -- which the caller
// has put in front on that stream, or which `gs` is made to wait for here), their words to pinned memory, the event
// pwn_tiled_wait waits for
static int gather_frames(pwn_ctx *c, pwn_tiled *t, unsigned long long from, unsigned long long to, hipStream_t gs)
{
	int rc;
	const int ls = (int)((to - 1) % NSLOT);
	for(unsigned long long g = from; g < to; g++)
	{
		const int s = (int)(g % NSLOT);
		if(t->sink)
		{
			if(gs != t->copy) HIPCHK(c, hipStreamWaitEvent(gs, t->ev_h[s], 0));       // the word goes out behind the copy (which is behind the blur)
		}
		else if(gs != t->fstream[s]) HIPCHK(c, hipStreamWaitEvent(gs, t->ev_b[s], 0));
	}
	if(t->timed[ls]) HIPCHK(c, hipEventRecord(t->ev_g0[ls], gs));
	if((t->world > 1 && !(t->hub != NULL && t->sink)) || t->self_exchange)          // (a group's sink: no strips travel, and no words)
	{
		TPCHK(c, t->tp->begin(gs));
		for(unsigned long long g = from; g < to; g++) { rc = t->self_exchange ? add_self_gather(c, t, g) : add_gather(c, t, g); if(rc != PWN_OK) return rc; }
		TPCHK(c, t->tp->end());
		t->info.groups++;
	}
	if(t->timed[ls]) { HIPCHK(c, hipEventRecord(t->ev_g1[ls], gs)); t->timed_g2[ls] = true; }
	for(unsigned long long g = from; g < to; g++) { rc = fetch_words(c, t, g, gs); if(rc != PWN_OK) return rc; }
	HIPCHK(c, hipEventRecord(t->ev_d[ls], gs));
	for(unsigned long long g = from; g < to; g++) t->gathered_by[g % NSLOT] = t->ev_d[ls];
	t->gathered = to;
	return PWN_OK;
}

// the pre-blur rows of the frame in slot s that the neighbours' blurs read (or, without a bounded halo, whole strips to
// everybody), as one grouped launch on `xs`, behind the frame's trace
static int exchange_halo(pwn_ctx *c, pwn_tiled *t, int s, hipStream_t xs, int y0, int y1)
{
	const size_t w4 = (size_t)c->w * 4;
	int rc;
	if(t->self_exchange && c->blur_passes)
	{
		// (to itself: the rows a middle rank exchanges with its two neighbours, out of this frame's pre-blur plane into the scratch plane)
		const int H = (int)(0.002 * c->h * 24.0) + 2 < (y1 - y0) ? (int)(0.002 * c->h * 24.0) + 2 : (y1 - y0);
		TPCHK(c, t->tp->begin(xs));
		TPCHK(c, t->tp->send(t->pre[s] + (size_t)y0 * c->w, (size_t)H * w4, 0));
		TPCHK(c, t->tp->recv(t->self_buf, (size_t)H * w4, 0));
		TPCHK(c, t->tp->send(t->pre[s] + (size_t)(y1 - H) * c->w, (size_t)H * w4, 0));
		TPCHK(c, t->tp->recv(t->self_buf + (size_t)H * c->w, (size_t)H * w4, 0));
		TPCHK(c, t->tp->end());
		t->info.groups++;
		t->info.bytes_sent += 2ull * H * w4; t->info.bytes_received += 2ull * H * w4;
	}
	if(t->world > 1 && c->blur_passes)
	{
		TPCHK(c, t->tp->begin(xs));
		if(t->halo)
		{
			const int H = t->halo;
			if(t->rank > 0)
			{
				TPCHK(c, t->tp->send(t->pre[s] + (size_t)y0 * c->w, (size_t)H * w4, t->rank - 1));
				TPCHK(c, t->tp->recv(t->pre[s] + (size_t)(y0 - H) * c->w, (size_t)H * w4, t->rank - 1));
				t->info.bytes_sent += (unsigned long long)H * w4; t->info.bytes_received += (unsigned long long)H * w4;
			}
			if(t->rank < t->world - 1 && y1 < c->h)
			{
				TPCHK(c, t->tp->send(t->pre[s] + (size_t)(y1 - H) * c->w, (size_t)H * w4, t->rank + 1));
				TPCHK(c, t->tp->recv(t->pre[s] + (size_t)y1 * c->w, (size_t)H * w4, t->rank + 1));
				t->info.bytes_sent += (unsigned long long)H * w4; t->info.bytes_received += (unsigned long long)H * w4;
			}
		}
		else { rc = add_allgather(c, t, s); if(rc != PWN_OK) return rc; }
		TPCHK(c, t->tp->end());
		t->info.groups++;
	}
	return PWN_OK;
}

static double now_us(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

extern "C" int pwn_tiled_submit(pwn_ctx *c, const float cam[16], float sec) { return pwn_i_tiled_submit(c, cam, sec, NULL, NULL, -1, NULL); }

int pwn_i_tiled_submit(pwn_ctx *c, const float cam[16], float sec, uint32_t *host_sbuf, float *host_zbuf, int zplane, uint32_t *host_surface)
{
	GRP_REFUSE(c);
	if(c == NULL || cam == NULL || c->tiled == NULL) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	if(t->sink && host_sbuf == NULL && t->host_base == NULL) return PWN_EINVAL;
	if(t->submitted - t->delivered >= NSLOT - 1) return PWN_EBUSY;
	if(t->tp->dead) { snprintf(c->err, sizeof(c->err), "%s transport: dead (%s)", t->tp->name(), t->tp->err); return PWN_ETIMEDOUT; }
	(void)hipSetDevice(c->device);
	const double t_in = now_us();
	const unsigned long long f = t->submitted;
	const int s = (int)(f % NSLOT);
	// A counted frame (PWN_OPT_COUNTERS, PWN_OPT_WAVE_LOG) goes on cs[0] whatever its parity: there is ONE set of
	// counters and one wave log, which a launch clears at its start -- two counted grids side by side would clear
	// each other's (and a wave log that grows would be freed under the other stream's kernel).  pwn_i_launch_trace
	// orders a launch that leaves the alternating pattern behind the launch two before it (the ticket sets).
	const bool counted = c->counters_on || c->wave_log_on;
	hipStream_t cs = counted ? t->cs[0] : t->cs[f % (unsigned long long)t->ncs];
	t->fstream[s] = cs;
	t->hdst[s] = host_sbuf; t->hzdst[s] = host_zbuf;
	t->hsurf[s] = NULL; t->fsurf[s] = NULL;
	if(host_surface != NULL)
	{
		if(t->surf_scale < 1 || zplane < 0 || zplane >= PWN_MAX_SLOTS || t->dsurf[zplane] == NULL) return PWN_EINVAL;
		t->hsurf[s] = host_surface; t->fsurf[s] = t->dsurf[zplane];
	}
	t->fz[s] = (zplane >= 0 && zplane < NSLOT) ? t->z[zplane] : t->z[s];        // (the caller's frames in flight name different planes)
	// (a counted frame right behind uncounted ones on the OTHER streams: wait for those frames' traces, so that the
	// counters and the wave log are this launch's alone)
	for(int back = 1; counted && back < t->ncs && (unsigned long long)back <= f; back++)
		if(t->fstream[(f - back) % NSLOT] != cs) HIPCHK(c, hipStreamWaitEvent(cs, t->ev_t[(f - back) % NSLOT], 0));
	t->fhalo[s] = t->halo;
	t->froot[s] = t->root_mode == PWN_TILED_ROOT_ROTATE ? (int)(f % (unsigned long long)t->world) : 0;
	memcpy(t->fcuts[s], t->cuts, sizeof(t->cuts));
	const int y0 = t->cuts[t->rank], y1 = t->cuts[t->rank + 1];
	// The slot's buffers were frame f-NSLOT's.  Its blur ran on this stream (NSLOT is even); its strips left in
	// G(f-NSLOT) and in the group that carried its gather, and the frame was delivered (NSLOT-1 in flight at most), which
	// waited for that group on the host: nothing to wait for here.
	uint32_t *plane = c->blur_passes ? t->pre[s] : ((t->rank == t->froot[s] && !t->sink) ? t->fin[s] : t->pre[s]);
	t->timed[s] = c->frame_timing > 0 && (f % (unsigned long long)c->frame_timing) == 0;
	t->timed_g2[s] = false;
	if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_k0[s], cs));
	c->trace_clear_word = t->missw[s];           // the frame's miss word is cleared by its trace launch (no memset between the kernels)
	c->trace_cost_word = c->blur_passes ? t->cost_acc + 16 * s : NULL;      // (the blur moves it on: enqueue_blur)
	c->trace_tables_event = t->ev_t[s];          // ... and ev_t, recorded right behind it, also tells when its tables are free again
	c->launch_room = t->ncs > 1 ? pwn_room_for_launch(c) : 0;          // PWN_OPT_TRACE_ROOM
	int rc = pwn_i_launch_trace(c, cam, sec, y0, y1, plane, t->fz[s], cs);
	if(rc != PWN_OK) { (void)hipEventRecord(t->ev_t[s], cs); return rc; }
	t->fcost_mul[s] = c->cost_mul; t->fcost_div[s] = c->cost_div;
	if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_k1[s], cs));
	HIPCHK(c, hipEventRecord(t->ev_t[s], cs));
	double tp0 = 0.0, tp1 = 0.0, tp2 = 0.0;
	if(t->prof) { tp0 = now_us(); t->p_trace += tp0 - t_in; }

	if(t->instream)
	{
		// ---- everything else of frame f behind its trace, on its stream: halo rows, blur, gather, words
		if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_g2[s], cs));
		rc = exchange_halo(c, t, s, cs, y0, y1);
		if(rc != PWN_OK) return rc;
		if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_g3[s], cs));
		if(t->prof) { tp1 = now_us(); t->p_halo += tp1 - tp0; }
		// Host sink: the words of a frame go out behind this rank's copy of its strip to the host, on the copy's stream (the
		// compute stream does not wait for PCIe) -- and one submit late, IN FRONT of this frame's copy: the transport runs
		// its launches in the order they were made, so G1(f+1) would otherwise wait for G2(f) and with it for copy f.
		if(t->sink && t->gathered < f)
		{
			rc = gather_frames(c, t, t->gathered, f, t->copy);
			if(rc != PWN_OK) return rc;
		}
		rc = enqueue_blur(c, t, f);
		if(rc != PWN_OK) return rc;
		if(t->prof) { tp2 = now_us(); t->p_blur += tp2 - tp1; }
		t->blurred = f + 1;
		if(!t->sink)
		{
			rc = gather_frames(c, t, f, f + 1, cs);
			if(rc != PWN_OK) return rc;
		}
		if(t->prof) { t->p_gather += now_us() - tp2; t->p_frames++; }
		t->submitted = f + 1;
		t->enqueue_us[s] = (float)(now_us() - t_in);
		return PWN_OK;
	}

	// ---- the blur of the frames before this one (normally just f-1), each on its frame's stream.  With one
	// compute stream that puts the blur of f-1 BEHIND the trace of f, so that the stream does not sit waiting for
	// the halo rows of f-1 while it could trace; with two, the other stream waits for them and this one traces.
	for(; t->blurred < f; t->blurred++)
	{
		rc = enqueue_blur(c, t, t->blurred);
		if(rc != PWN_OK) return rc;
	}

	// ---- SPLIT form.  On the comm stream, two grouped launches.  First the second half of the frames that are blurred
	// and not gathered yet except the newest blur (its kernel was enqueued a moment ago: next time), i.e. normally of
	// frame f-2: it waits for that frame's blur only, NOT for the trace enqueued above.  (As ONE group with the halo
	// rows below, which need trace f, every delivery waited for the newest trace, round 2.  Gathering the newest blur
	// too, round 4, made the chain blur -> words -> halo event -> next blur the limit: 0.28 instead of 0.21 ms per
	// strip-sized frame, profiles/r4/host_bound.txt.)
	const unsigned long long g_end = f >= 1 ? f - 1 : 0;        // gather frames [gathered, g_end)
	if(g_end > t->gathered)
	{
		rc = gather_frames(c, t, t->gathered, g_end, t->comm);
		if(rc != PWN_OK) return rc;
	}
	// ---- then this frame's pre-blur rows, behind its trace
	HIPCHK(c, hipStreamWaitEvent(t->comm, t->ev_t[s], 0));
	if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_g2[s], t->comm));
	rc = exchange_halo(c, t, s, t->comm, y0, y1);
	if(rc != PWN_OK) return rc;
	if(t->timed[s]) HIPCHK(c, hipEventRecord(t->ev_g3[s], t->comm));       // (ev_x carries no time stamp)
	HIPCHK(c, hipEventRecord(t->ev_x[s], t->comm));
	t->submitted = f + 1;
	t->enqueue_us[s] = (float)(now_us() - t_in);
	return PWN_OK;
}

// Block until `ev` has happened -- but not for longer than the wait deadline, and not past an error the transport
// reports meanwhile.  What a frame waits for may sit behind a kernel of the transport that itself waits for a peer: a
// peer that never sends would hold hipEventSynchronize for ever.  On expiry the transport is aborted (its kernels
// leave the device) and the tiling is dead.
static int wait_event(pwn_ctx *c, pwn_tiled *t, hipEvent_t ev, const char *what, unsigned long long frame)
{
	hipError_t e = hipEventQuery(ev);
	if(e == hipSuccess) return PWN_OK;
	const double t0 = now_ms();
	double look = t0 + 2.0;
	for(unsigned spins = 0;; spins++)
	{
		e = hipEventQuery(ev);
		if(e == hipSuccess) { (void)hipGetLastError(); return PWN_OK; }
		if(e != hipErrorNotReady) { snprintf(c->err, sizeof(c->err), "rank %d: frame %llu, %s: %s", t->rank, frame, what, hipGetErrorString(e)); return PWN_EHIP; }
		if(spins < 4000u) continue;              // the usual wait is a fraction of a millisecond: no clock, no sleep
		const double now = now_ms();
		if(now >= look)
		{
			look = now + 5.0;
			const int rc = t->tp->alive();
			if(rc != PWN_OK && !t->tp->dead)
			{
				t->tp->abort();
				snprintf(c->err, sizeof(c->err), "%s transport: %s (frame %llu, waiting for %s); aborted", t->tp->name(), t->tp->err, frame, what);
				return rc;
			}
		}
		if(now - t0 > (double)t->tp->wait_ms)
		{
			t->tp->abort();
			snprintf(c->err, sizeof(c->err), "rank %d of %d: frame %llu: %s not reached within %d ms (submitted %llu, blurred %llu, gathered %llu, "
				"delivered %llu); %s transport aborted", t->rank, t->world, frame, what, t->tp->wait_ms, t->submitted, t->blurred, t->gathered,
				t->delivered, t->tp->name());
			snprintf(t->tp->err, sizeof(t->tp->err), "deadline passed at frame %llu (%s)", frame, what);
			return PWN_ETIMEDOUT;
		}
		struct timespec ts = { 0, 20 * 1000 };
		nanosleep(&ts, NULL);
	}
}

// Without blocking: is the frame `ahead` behind the oldest one in flight through on this rank's device -- its strip blurred and copied
// to the host (sink) or its gather's group finished?  1 / 0 / PWN_E*.  (A hint for pwn_frame_ready on a group's handle: a frame
// whose taps left the halo still has its repeat in front of it, which pwn_tiled_wait runs.)
int pwn_i_tiled_ready(pwn_ctx *c, int ahead)
{
	if(c == NULL || c->tiled == NULL || ahead < 0) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	const unsigned long long d = t->delivered + (unsigned long long)ahead;
	if(d >= t->submitted) return PWN_EINVAL;
	if(t->tp->dead) return PWN_ETIMEDOUT;
	(void)hipSetDevice(c->device);
	const int s = (int)(d % NSLOT);
	hipEvent_t ev = t->sink ? t->ev_h[s] : (t->gathered > d ? t->gathered_by[s] : NULL);
	if(ev == NULL) return 0;
	const hipError_t e = hipEventQuery(ev);
	if(e == hipSuccess) return 1;
	(void)hipGetLastError();
	return e == hipErrorNotReady ? 0 : PWN_EHIP;
}

extern "C" int pwn_tiled_wait(pwn_ctx *c, int flags, pwn_tiled_frame *out)
{
	GRP_REFUSE(c);
	if(c == NULL || c->tiled == NULL) return PWN_EINVAL;
	pwn_tiled *t = c->tiled;
	if(t->delivered >= t->submitted) return PWN_EINVAL;          // nothing in flight
	if(t->tp->dead) { snprintf(c->err, sizeof(c->err), "%s transport: dead (%s)", t->tp->name(), t->tp->err); return PWN_ETIMEDOUT; }
	(void)hipSetDevice(c->device);
	const unsigned long long d = t->delivered;
	const int s = (int)(d % NSLOT);
	const size_t n = (size_t)c->w * (size_t)c->h;
	int rc;
	const double tw0 = t->prof ? now_us() : 0.0;
	double tw1 = 0.0, tw2 = 0.0;
	// no newer frame has enqueued this one's blur / carried its gather: do both now
	for(; t->blurred <= d; t->blurred++)
	{
		rc = enqueue_blur(c, t, t->blurred);
		if(rc != PWN_OK) return rc;
	}
	if(t->gathered <= d)
	{
		// (in-stream, which leaves only a host sink's words for later: behind the copies, on their stream)
		rc = gather_frames(c, t, t->gathered, d + 1, t->instream ? (t->sink ? t->copy : t->fstream[s]) : t->comm);
		if(rc != PWN_OK) return rc;
	}
	rc = wait_event(c, t, t->gathered_by[s], "the group that carries its strips and words", d);
	if(rc != PWN_OK) return rc;
	if(!t->instream || t->sink)         // (in-stream the gather's event is behind the blur on the frame's stream)
	{
		rc = wait_event(c, t, t->ev_b[s], "its blur (behind the halo rows of its neighbours)", d);               // (world 1, and rank 0's own strip)
		if(rc != PWN_OK) return rc;
	}
	if(t->sink) { rc = wait_event(c, t, t->ev_h[s], "the copy of its strip into the host frame", d); if(rc != PWN_OK) return rc; }      // this rank's own strip is in the host frame

	if(t->prof) { tw1 = now_us(); t->p_events += tw1 - tw0; }
	// ---- a group's members meet here: every member's kernels and copies of this frame are through, its two words are in ITS pinned
	// memory (fetch_words), and everybody reads everybody's from there
	if(t->hub != NULL)
	{
		rc = pwn_hub_meet(t->hub, t->tp->wait_ms);
		if(rc != PWN_OK)
		{
			t->tp->abort();
			snprintf(c->err, sizeof(c->err), "rank %d of %d: frame %llu: the group's members did not all arrive with it within %d ms (or one of them failed)", t->rank, t->world, d, t->tp->wait_ms);
			return rc;
		}
	}
	if(t->prof) { tw2 = now_us(); t->p_meet += tw2 - tw1; }
	// ---- the ranks' words of this frame (they came to pinned memory behind the group that carried them: fetch_words)
	const uint32_t *h = t->h_missv + (size_t)s * ((size_t)t->world + 1) * 2;
	uint32_t cost[MAXW];
	bool miss = false;
	for(int r = 0; r < t->world; r++)
	{
		const uint32_t *wr = r == t->rank ? h + 2 * t->world : h + 2 * r;
		if(t->hub != NULL && r != t->rank) wr = t->hub->member[r]->tiled->h_missv + ((size_t)s * ((size_t)t->world + 1) + (size_t)t->world) * 2;
		if(t->fhalo[s] && wr[0] != 0u) miss = true;           // was the bounded halo enough for this frame, on every rank?
		cost[r] = wr[1];
	}
	memcpy(t->last_cost, cost, sizeof(uint32_t) * (size_t)t->world);
	if(miss)
	{
		// Every rank sees the same words and comes here together: the frame's exchange again with
		// whole strips (pre[s] and z[s] still hold this frame), blur, gather; whole strips from now on.
		// Behind the groups of the newer frames that are already enqueued (split form: on the comm stream; in-stream: on
		// this frame's stream, the transport keeping the order of its own launches).
		t->info.frames_redone++;
		t->halo = 0; t->fhalo[s] = 0;
		hipStream_t cs = t->fstream[s];
		hipStream_t xs = t->instream ? cs : t->comm;          // the stream of the exchanges (in-stream: waits on its own events cost nothing)
		int y0, y1; rows_of(t, s, t->rank, &y0, &y1);
		TPCHK(c, t->tp->begin(xs));
		rc = add_allgather(c, t, s);
		if(rc != PWN_OK) return rc;
		TPCHK(c, t->tp->end());
		HIPCHK(c, hipEventRecord(t->ev_d[s], xs));
		HIPCHK(c, hipStreamWaitEvent(cs, t->ev_d[s], 0));
		uint32_t *dst = (t->rank == t->froot[s] && !t->sink) ? t->fin[s] : t->out[s];
		// (this stream's cost accumulator may hold the trace of frame d+2 by now: it is left alone, the frame's
		// cost word was moved by its first blur)
		rc = pwn_i_launch_blur(c, y0, y1, t->pre[s], t->fz[s], dst, cs, 0, 0, NULL, NULL, NULL);
		if(rc != PWN_OK) return rc;
		HIPCHK(c, hipEventRecord(t->ev_b[s], cs));
		HIPCHK(c, hipStreamWaitEvent(xs, t->ev_b[s], 0));
		if(t->sink)
		{
			rc = copy_strip_to_host(c, t, s);                     // the strip again, and the words behind it
			if(rc != PWN_OK) return rc;
			HIPCHK(c, hipStreamWaitEvent(xs, t->ev_h[s], 0));
		}
		TPCHK(c, t->tp->begin(xs));
		rc = add_gather(c, t, d);
		if(rc != PWN_OK) return rc;
		TPCHK(c, t->tp->end());
		t->info.groups += 2;
		// (everything of the repeat is in front of this event on the comm stream: the whole strips, the blur behind them --
		// the comm stream waited for ev_b -- the copy to the host, the gather)
		HIPCHK(c, hipEventRecord(t->ev_d[s], xs));
		rc = wait_event(c, t, t->ev_d[s], "its repeat with whole strips", d);
		if(rc != PWN_OK) return rc;
		HIPCHK(c, hipStreamSynchronize(cs));
		if(t->sink) HIPCHK(c, hipStreamSynchronize(t->copy));
	}
	// ---- moving cuts: every balance_every delivered frames, new cuts from what this frame's strips cost.  Every
	// rank has the same words and the same cuts of this frame, so every rank computes the same new cuts, and they
	// take effect with the same frame: the next one submitted.
	if(t->acc_frames == 0 || memcmp(t->acc_cuts, t->fcuts[s], sizeof(int) * (size_t)(t->world + 1)) != 0)
	{
		memcpy(t->acc_cuts, t->fcuts[s], sizeof(int) * (size_t)(t->world + 1));
		memcpy(t->acc_cost, cost, sizeof(uint32_t) * (size_t)t->world);
		t->acc_frames = 1;
	}
	else
	{
		for(int r = 0; r < t->world; r++) if(cost[r] < t->acc_cost[r]) t->acc_cost[r] = cost[r];
		t->acc_frames++;
	}
	if(t->balance_every > 0 && t->world > 1 && ((d + 1) % (unsigned long long)t->balance_every) == 0)
	{
		int nc[MAXW + 1];
		if(recut(t->acc_cuts, t->acc_cost, t->world, c->h, t->halo > 0 ? t->halo : 8, t->max_rows, nc))
		{
			memcpy(t->cuts, nc, sizeof(int) * (size_t)(t->world + 1));
			t->info.recuts++;
		}
	}
	t->delivered = d + 1;
	t->info.frames++;
	if(t->prof) t->p_rest += now_us() - tw2;
	if(t->ncs > 1) pwn_room_frame_done(c);
	if(out != NULL)
	{
		memset(out, 0, sizeof(*out));
		out->seq = d + 1;
		out->redone = miss ? 1 : 0;
		out->timed = t->timed[s] ? 1 : 0;
		out->y0 = t->fcuts[s][t->rank]; out->y1 = t->fcuts[s][t->rank + 1];
		out->cost = cost[t->rank];
		out->root = t->sink ? -1 : t->froot[s];
		out->enqueue_us = t->enqueue_us[s];
		if(t->timed[s])
		{
			(void)hipEventElapsedTime(&out->trace_ms, t->ev_k0[s], t->ev_k1[s]);
			if(c->blur_passes)
			{
				(void)hipEventElapsedTime(&out->blur_ms, t->ev_k2[s], t->ev_k3[s]);
				(void)hipEventElapsedTime(&out->frame_ms, t->ev_k0[s], t->ev_k3[s]);      // trace .. blur: waiting for the halo rows in between
				(void)hipEventElapsedTime(&out->halo_ms, t->ev_g2[s], t->ev_g3[s]);
			}
			if(t->timed_g2[s]) (void)hipEventElapsedTime(&out->gather_ms, t->ev_g0[s], t->ev_g1[s]);
			if(hipGetLastError() != hipSuccess) { /* (an event without timing data: the figure stays 0) */ }
		}
		if(t->hdst[s] != NULL) out->sbuf = t->hdst[s];
		else if(t->host_base != NULL) out->sbuf = (const uint32_t *)(t->host_base + (size_t)s * n * 4);
		else if(t->rank == t->froot[s])
		{
			out->d_sbuf = t->fin[s];
			if(flags & PWN_TILED_HOST)
			{
				if(t->h_frame == NULL) HIPCHK(c, hipHostMalloc((void **)&t->h_frame, n * 4, hipHostMallocDefault));
				HIPCHK(c, hipMemcpy(t->h_frame, t->fin[s], n * 4, hipMemcpyDeviceToHost));
				out->sbuf = t->h_frame;
			}
		}
	}
	return PWN_OK;
}
