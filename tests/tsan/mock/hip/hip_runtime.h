// Device-less stand-in for <hip/hip_runtime.h>, for the ThreadSanitizer build of the library's OWN host protocol code
// (tests/tsan/Makefile): kq_compat.cpp and kq_radio.cpp compile against it unchanged.  Test infrastructure only.
//
// What it models of the device: memory is host memory, and a stream is an in-order queue -- every operation that would
// go through a stream runs to completion under one mutex (mock_stream_mutex), which is the ordering the real stream
// gives the kernels and copies that the compat surface queues on its single stream.  Nothing else of HIP is here.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>

struct float2 {
  float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
typedef struct mock_stream_t *hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };

inline std::mutex &mock_stream_mutex() {
  static std::mutex m;
  return m;
}
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock error"; }
inline hipError_t hipGetDeviceCount(int *n) {
  *n = 1;
  return hipSuccess;
}
inline hipError_t hipGetDevice(int *d) {
  *d = 0;
  return hipSuccess;
}
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
  *s = reinterpret_cast<hipStream_t>(new char);
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());  // everything queued so far has run
  return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) {
  *p = calloc(1, n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void *p) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());  // hipFree waits for the device
  free(p);
  return hipSuccess;
}
inline hipError_t hipMemset(void *p, int v, size_t n) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  memset(p, v, n);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  memcpy(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  memcpy(d, s, n);
  return hipSuccess;
}
