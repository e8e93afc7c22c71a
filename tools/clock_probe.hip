// clock_probe.hip -- the rate of the shader's cycle counter (s_memtime) against the constant 100 MHz counter while
// something else loads the device.  Measured on MI355X: 2.39-2.41 GHz on every box, idle or under the filter kernel --
// whatever makes the same build 7 % slower on some boxes of the pool than on others does not show here
// (GRBM_GUI_ACTIVE from rocprofv3 --pmc over the kernel's duration gives 2.2-2.35 GHz).
//   hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o /tmp/clock_probe
//   python bench.py --steps 4000 --no-cpu-baseline & sleep 15; /tmp/clock_probe 40
// One wave spins for 100 ms of the constant 100 MHz counter (s_memrealtime) and reports how far the shader-clock counter
// (s_memtime) moved meanwhile; repeated `n` times.  It occupies one wave slot of one CU and no LDS, so it runs beside
// the kernels of another process.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void probe(unsigned long long *out, unsigned long long ticks) {
  unsigned long long const r0 = wall_clock64(), c0 = clock64();
  unsigned long long r = r0;
  while (r - r0 < ticks) {
    __builtin_amdgcn_s_sleep(64);
    r = wall_clock64();
  }
  out[0] = clock64() - c0;
  out[1] = r - r0;
}

int main(int argc, char **argv) {
  int const n = argc > 1 ? atoi(argv[1]) : 20;
  unsigned long long *d, h[2];
  if (hipMalloc(&d, 16) != hipSuccess) return 1;
  for (int i = 0; i < n; i++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 10000000ull);
    if (hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("%.0f MHz\n", (double)h[0] / ((double)h[1] / 100e6) / 1e6);
    fflush(stdout);
  }
  return 0;
}
