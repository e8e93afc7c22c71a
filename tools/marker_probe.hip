// marker_probe.hip -- what an event record between two kernels of one stream costs on the device.
//   hipcc -O3 --offload-arch=gfx950 tools/marker_probe.hip -o tools/marker_probe.bin ;  gpurun -- tools/marker_probe.bin
// A step = one long kernel (writes `mb` MB, ~`us` microseconds of arithmetic), K event records, one short kernel.
// Steps are queued back to back; the time per step minus the K = 0 time is what the K markers cost.  Variants: event
// flags (default / disable-timing / + release-to-device), and a second stream waiting on the first marker (as the
// demodulator stream does on ev_filter_done).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__global__ __launch_bounds__(256) void k_long(float4 *out, size_t n4, int spin) {
  size_t const i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float4 v = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
  for (int k = 0; k < spin; k++) {
    v.x = __builtin_fmaf(v.x, 1.0001f, v.y);
    v.y = __builtin_fmaf(v.y, 0.9999f, v.z);
    v.z = __builtin_fmaf(v.z, 1.0002f, v.w);
    v.w = __builtin_fmaf(v.w, 0.9998f, v.x);
  }
  if (i < n4) out[i] = v;
}
__global__ void k_short(float *p) {
  if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
}

int main(int argc, char **argv) {
  int const steps = 200;
  size_t const n4 = (size_t)17 * 1024 * 1024 / 16;  // 17 MB of dirty lines behind the long kernel
  float4 *big;
  float *small;
  CHECK(hipMalloc(&big, n4 * 16));
  CHECK(hipMalloc(&small, 4096));
  CHECK(hipMemset(small, 0, 4096));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned const grid = (unsigned)((n4 + 255) / 256) * 4;  // 4 x more workgroups than stores: ~1 ms with the spin below
  int const spin = argc > 1 ? atoi(argv[1]) : 2200;
  struct Variant {
    const char *name;
    unsigned flags;
    bool cross;
  } const variants[] = {{"default events", hipEventDefault, false},
                        {"disable-timing", hipEventDisableTiming, false},
                        {"disable-timing + release-to-device", hipEventDisableTiming | hipEventReleaseToDevice, false},
                        {"disable-timing, 2nd stream waits on the 1st marker", hipEventDisableTiming, true}};
  for (Variant const &v : variants) {
    for (int K = 0; K <= 4; K++) {
      std::vector<hipEvent_t> ev(steps * (K ? K : 1));
      for (auto &e : ev) CHECK(hipEventCreateWithFlags(&e, v.flags));
      hipEvent_t t0, t1;
      CHECK(hipEventCreate(&t0));
      CHECK(hipEventCreate(&t1));
      for (int pass = 0; pass < 2; pass++) {  // pass 0 warms up
        CHECK(hipEventRecord(t0, s1));
        for (int i = 0; i < steps; i++) {
          hipLaunchKernelGGL(k_long, dim3(grid), dim3(256), 0, s1, big, n4, spin);
          for (int k = 0; k < K; k++) CHECK(hipEventRecord(ev[i * K + k], s1));
          if (v.cross && K > 0) {
            CHECK(hipStreamWaitEvent(s2, ev[i * K], 0));
            hipLaunchKernelGGL(k_short, dim3(64), dim3(64), 0, s2, small + 512);
          }
          hipLaunchKernelGGL(k_short, dim3(64), dim3(64), 0, s1, small);
        }
        CHECK(hipEventRecord(t1, s1));
        CHECK(hipEventSynchronize(t1));
        CHECK(hipStreamSynchronize(s2));
      }
      float ms;
      CHECK(hipEventElapsedTime(&ms, t0, t1));
      printf("%-52s K=%d  %8.2f us per step\n", v.name, K, ms / steps * 1e3);
      fflush(stdout);
      for (auto &e : ev) CHECK(hipEventDestroy(e));
    }
  }
  return 0;
}
