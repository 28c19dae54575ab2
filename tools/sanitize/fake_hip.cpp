// tools/sanitize/fakehip/hip/hip_runtime.h: the stand-in's few lines of state (which host ranges are pinned; a clock for events)
#include <hip/hip_runtime.h>
#include <time.h>
#include <map>
#include <mutex>

struct fakehip_stream { int dummy; };
struct fakehip_event { double t; bool recorded; };
static std::mutex g_mu;
static std::map<const char *, size_t> g_pinned;
static thread_local int t_device = 0;
static int g_devices = 8;

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

extern "C" {
hipError_t hipGetDeviceCount(int *n) { if(const char *e = getenv("FAKEHIP_DEVICES")) g_devices = atoi(e); *n = g_devices; return hipSuccess; }
hipError_t hipSetDevice(int d) { t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 4; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *s, int len, int d) { snprintf(s, (size_t)len, "0000:%02x:00.0", d); return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
	*p = malloc(n ? n : 1);
	if(*p == NULL) return hipErrorOutOfMemory;
	std::lock_guard<std::mutex> g(g_mu); g_pinned[(const char *)*p] = n; return hipSuccess;
}
hipError_t hipHostFree(void *p) { { std::lock_guard<std::mutex> g(g_mu); g_pinned.erase((const char *)p); } free(p); return hipSuccess; }
hipError_t hipHostRegister(void *p, size_t n, unsigned)
{
	std::lock_guard<std::mutex> g(g_mu);
	if(g_pinned.count((const char *)p)) return hipErrorHostMemoryAlreadyRegistered;
	g_pinned[(const char *)p] = n; return hipSuccess;
}
hipError_t hipHostUnregister(void *p) { std::lock_guard<std::mutex> g(g_mu); return g_pinned.erase((const char *)p) ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
	std::lock_guard<std::mutex> g(g_mu);
	memset(a, 0, sizeof(*a));
	auto it = g_pinned.upper_bound((const char *)p);
	if(it != g_pinned.begin()) { --it; if((const char *)p < it->first + it->second) { a->type = hipMemoryTypeHost; return hipSuccess; } }
	a->type = hipMemoryTypeUnregistered;
	return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = new fakehip_stream(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new fakehip_stream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = new fakehip_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new fakehip_event(); (*e)->t = 0; (*e)->recorded = false; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
// (an event may be recorded by one member's thread and looked at by another's: the stand-in keeps that race-free with the lock)
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { std::lock_guard<std::mutex> g(g_mu); e->t = now_ms(); e->recorded = true; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { std::lock_guard<std::mutex> g(g_mu); *ms = (float)(b->t - a->t); return hipSuccess; }
}
