// Stand-in for <rccl/rccl.h> for the CPU ThreadSanitizer run of kq_fanout.cpp (tests/tsan/Makefile): the handful of
// types, enumerators (RCCL's values) and prototypes that file uses.  Test infrastructure only; the product is compiled
// against the real header of /opt/rocm.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

typedef struct mock_nccl_comm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t s);
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s);
const char *ncclGetErrorString(ncclResult_t e);
ncclResult_t ncclGetVersion(int *v);
ncclResult_t ncclCommCount(const ncclComm_t comm, int *n);
}
