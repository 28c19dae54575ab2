// the few RCCL TYPES pwn_tiled.cpp names (its entry points are resolved with dlopen at run time and are not used by the sanitizer
// runs: PWN_GROUP_TRANSPORT=local, PWN_TRANSPORT_SHM); see ../hip/hip_runtime.h
#pragma once
#include <hip/hip_runtime.h>
typedef struct fake_nccl_comm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct { size_t size; unsigned magic; unsigned version; int blocking; int cgaClusterSize; int minCTAs; int maxCTAs; const char *netName; int splitShare; } ncclConfig_t;
#define NCCL_CONFIG_INITIALIZER { sizeof(ncclConfig_t), 0xcafebeef, 0, 1, 0, 0, 0, NULL, 0 }
