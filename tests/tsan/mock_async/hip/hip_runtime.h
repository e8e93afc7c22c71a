// Device-less stand-in for <hip/hip_runtime.h> with ASYNCHRONOUS streams, for the ThreadSanitizer run of the library's
// multi-GPU fan-out (kq_fanout.cpp compiles against it unchanged; tests/tsan/Makefile, fanout_harness.cpp).
// Test infrastructure only.
//
// What it models of the device, and only that:
//   * a "device" is a thread-local index (hipSetDevice); memory is host memory;
//   * a stream is an in-order queue drained by its own worker thread: an operation queued on a stream runs later, on
//     that thread, after everything queued on the same stream before it -- and in no order at all with respect to other
//     streams or to the host thread, unless an event says so;
//   * hipEventRecord queues a marker; hipStreamWaitEvent captures the event's most recent record AT THE CALL and queues
//     a wait for that record (a never-recorded event is no wait at all -- HIP's rule); hipEventQuery / Synchronize /
//     ElapsedTime look at the most recent record.
// The only happens-before edges between threads are therefore the ones the real runtime gives: queue order on one
// stream, event record -> wait, and the synchronising calls.  A slot overwritten before its consumer released it, or
// read before its batch landed, is a data race on plain memory that ThreadSanitizer reports.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

struct float2 {
  float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };

struct mock_stream_t {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  uint64_t submitted = 0, completed = 0;
  bool stop = false;
  std::thread worker;
  mock_stream_t() {
    worker = std::thread([this] {
      for (;;) {
        std::function<void()> op;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [this] { return stop || !q.empty(); });
          if (q.empty()) return;
          op = std::move(q.front());
          q.pop_front();
        }
        op();
        {
          std::lock_guard<std::mutex> lk(m);
          completed++;
        }
        cv.notify_all();
      }
    });
  }
  void enqueue(std::function<void()> op) {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(std::move(op));
      submitted++;
    }
    cv.notify_all();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(m);
    uint64_t const upto = submitted;
    cv.wait(lk, [&] { return completed >= upto; });
  }
  ~mock_stream_t() {
    drain();
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    worker.join();
  }
};
typedef mock_stream_t *hipStream_t;

struct mock_event_t {
  std::mutex m;
  std::condition_variable cv;
  uint64_t recorded = 0, completed = 0;  // generation of the latest hipEventRecord / of the latest marker reached
  std::chrono::steady_clock::time_point when;
  bool timing = true;
};
typedef mock_event_t *hipEvent_t;

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline int &mock_current_device() {
  static thread_local int d = 0;
  return d;
}
// failure injection for the tests: the n-th hipMalloc from now on THIS thread fails (0: never)
inline int &mock_fail_malloc_in() {
  static thread_local int n = 0;
  return n;
}
inline hipError_t hipGetDeviceCount(int *n) {
  *n = 64;
  return hipSuccess;
}
inline hipError_t hipGetDevice(int *d) {
  *d = mock_current_device();
  return hipSuccess;
}
inline hipError_t hipSetDevice(int d) {
  mock_current_device() = d;
  return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
  *s = new mock_stream_t();
  return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) {
  delete s;
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s) {
  if (s) s->drain();
  return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) {
  int &f = mock_fail_malloc_in();
  if (f > 0 && --f == 0) {
    *p = nullptr;
    return hipErrorOutOfMemory;
  }
  *p = calloc(1, n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void *p) {
  free(p);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
  memcpy(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemset(void *p, int v, size_t n) {
  memset(p, v, n);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st) {
  if (!st) return hipErrorInvalidValue;
  st->enqueue([=] { memcpy(d, s, n); });
  return hipSuccess;
}
typedef void *hipDeviceptr_t;
inline hipError_t hipMemsetD32Async(hipDeviceptr_t p, int v, size_t count, hipStream_t st) {
  if (!st) return hipErrorInvalidValue;
  st->enqueue([=] {
    for (size_t i = 0; i < count; i++) static_cast<int *>(p)[i] = v;
  });
  return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags) {
  *e = new mock_event_t();
  (*e)->timing = !(flags & hipEventDisableTiming);
  return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
  if (!e || !st) return hipErrorInvalidValue;
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lk(e->m);
    gen = ++e->recorded;
  }
  st->enqueue([e, gen] {
    {
      std::lock_guard<std::mutex> lk(e->m);
      if (gen > e->completed) {
        e->completed = gen;
        e->when = std::chrono::steady_clock::now();
      }
    }
    e->cv.notify_all();
  });
  return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned) {
  if (!e || !st) return hipErrorInvalidValue;
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lk(e->m);
    gen = e->recorded;  // the record the wait refers to is fixed now; later records do not move it
  }
  if (gen == 0) return hipSuccess;
  st->enqueue([e, gen] {
    std::unique_lock<std::mutex> lk(e->m);
    e->cv.wait(lk, [&] { return e->completed >= gen; });
  });
  return hipSuccess;
}
inline hipError_t hipEventQuery(hipEvent_t e) {
  std::lock_guard<std::mutex> lk(e->m);
  return e->completed >= e->recorded ? hipSuccess : hipErrorNotReady;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> lk(e->m);
  uint64_t const gen = e->recorded;
  e->cv.wait(lk, [&] { return e->completed >= gen; });
  return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  std::scoped_lock lk(a->m, b->m);
  if (!a->timing || !b->timing || a->completed == 0 || b->completed == 0) return hipErrorInvalidValue;
  if (a->completed < a->recorded || b->completed < b->recorded) return hipErrorNotReady;
  *ms = std::chrono::duration<float, std::milli>(b->when - a->when).count();
  return hipSuccess;
}
