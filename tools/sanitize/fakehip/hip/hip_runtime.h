// A stand-in for the HIP runtime, for CPU sanitizer builds of libpwnhip's HOST logic only (tools/sanitize/README.txt): pwn_api.cpp,
// pwn_tiled.cpp and pwn_group.cpp compiled with g++ -fsanitize=thread / address,undefined against this header, with CPU stand-ins for
// the kernels (fake_kernels.cpp).  Test infrastructure: nothing under pwnfps_amd/ includes it, and nothing here computes a pixel of
// the product.  Streams are synchronous: a launch or a copy has run when the call returns, an event has happened when it was
// recorded -- what the sanitizers look at is what the host threads do to each other's memory, and the choreography's bookkeeping.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorHostMemoryAlreadyRegistered = 712,
               hipErrorInvalidConfiguration = 9 } hipError_t;
typedef struct fakehip_stream *hipStream_t;
typedef struct fakehip_event *hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 } hipMemoryType;
typedef struct { hipMemoryType type; int device; void *devicePointer; void *hostPointer; } hipPointerAttribute_t;
typedef struct { char gcnArchName[256]; int multiProcessorCount; } hipDeviceProp_t;
struct uint2 { unsigned x, y; };
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
#define hipHostMallocDefault 0u
#define hipHostMallocPortable 1u
#define hipHostRegisterDefault 0u
#define hipHostRegisterPortable 1u

#ifdef __cplusplus
extern "C" {
#endif
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize(void);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipDeviceCanAccessPeer(int *can, int a, int b);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipDeviceGetPCIBusId(char *s, int len, int d);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p);
hipError_t hipMemset(void *p, int v, size_t n);
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpyPeerAsync(void *d, int dd, const void *s, int sd, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int prio);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
#ifdef __cplusplus
}
#endif
