// side_probe.hip -- a side-stream kernel with the footprint of an RCCL broadcast, for tools/bcast_probe.py:
// `nwg` workgroups of `threads` threads (RCCL: one workgroup per channel, 256 threads) stream `bytes` from src to dst
// with 16-byte accesses, hold `lds` bytes of LDS and ~120 vector registers each, and stay resident for at least `hold_us`
// microseconds (a receiver's kernel lives as long as the link takes to deliver, whatever the local CUs do).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/side_probe.hip -o tools/side_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void k_side(const uint4 *src, uint4 *dst, size_t n16, unsigned long long hold_ticks) {
  extern __shared__ uint4 lds[];
  unsigned long long const t0 = __builtin_readcyclecounter();  // s_memtime: 100 MHz-class constant clock on gfx9
  unsigned long long const w0 = wall_clock64();
  size_t const stride = (size_t)gridDim.x * blockDim.x;
  uint4 keep[24];  // ~96 registers of payload in flight, as a copy loop with deep unrolling holds
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * 24) {
#pragma unroll
    for (int u = 0; u < 24; u++) keep[u] = i + u * stride < n16 ? src[i + u * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 24; u++)
      if (i + u * stride < n16) dst[i + u * stride] = keep[u];
  }
  if (hold_ticks) {
    if (threadIdx.x == 0) lds[0] = make_uint4((unsigned)t0, 0, 0, 0);
    while (wall_clock64() - w0 < hold_ticks) __builtin_amdgcn_s_sleep(32);
  }
}

extern "C" int side_probe_launch(void *stream, const void *src, void *dst, size_t bytes, int nwg, int threads, int lds, double hold_us) {
  static int set = 0;
  if (lds > 64 * 1024 && !set) {
    if (hipFuncSetAttribute((const void *)k_side, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
    set = 1;
  }
  int rate_khz = 100000;  // wall_clock64 ticks at 100 MHz on gfx9
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  unsigned long long const ticks = (unsigned long long)(hold_us * 1e-6 * rate_khz * 1e3);
  hipLaunchKernelGGL(k_side, dim3(nwg), dim3(threads), lds, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst, bytes / 16, ticks);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
