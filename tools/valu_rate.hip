// valu_rate.hip -- issue rate of the vector instructions the FFT kernels are made of, per SIMD, on the device at hand.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Every wave runs a loop of 8 independent instructions of one kind (no dependent issue stalls); one workgroup per CU with
// 1, 2 or 4 waves per SIMD.  Cycles are counted by the shader's own clock counter (s_memtime) inside every wave -- mean
// (min-max) over the waves of [cycles the wave ran / instructions its SIMD issued meanwhile] -- and the frequency the
// counter ran at comes from the constant 100 MHz counter (s_memrealtime) beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float seed, unsigned long long *clk) {
  // shader-clock (s_memtime) and 100 MHz (s_memrealtime) counters around the loop: their ratio is the clock the shader
  // really ran at, whatever hipDeviceProp_t::clockRate says
  unsigned long long const c0 = clock64(), r0 = wall_clock64();
  v2f a[8], b = {seed, 1.0f + seed}, c = {0.5f, 0.25f};
  v2f sconst = {1.0001f, 0.9999f};
  for (int i = 0; i < 8; i++) a[i] = (v2f){seed + i, seed - i};
  float f[8];
  for (int i = 0; i < 8; i++) f[i] = seed + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b.x), "v"(c.x));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(b));
      if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sconst), "v"(c));
      if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));   // three different VGPR pairs
      if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b.x));
      if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(f[i]) : "v"(b.x));
      if (KIND == 9) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 3) & 7]));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + f[i];
  if ((threadIdx.x & 63) == 0) {
    int const wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    clk[2 * wave] = clock64() - c0;
    clk[2 * wave + 1] = wall_clock64() - r0;
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
void run(const char *name) {
  int dev = 0;
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, dev);
  int const cus = p.multiProcessorCount, iters = 100000;
  float *out;
  unsigned long long *clk;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * 16 * cus);
  printf("%-46s", name);
  for (int w = 1; w <= 4; w *= 2) {   // waves per SIMD: one workgroup of 4 w waves per CU
    std::vector<unsigned long long> h(2 * 4 * w * cus);
    dim3 grid(cus), block(256 * w);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, 1000, 1.0f, clk);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters, 1.0f, clk);
    (void)hipMemcpy(h.data(), clk, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0, mhz = 0, lo = 1e30, hi = 0;
    int const nw = 4 * w * cus;
    for (int i = 0; i < nw; i++) {
      double const c = (double)h[2 * i] / ((double)iters * 8 * w);   // shader cycles the wave ran / instructions its SIMD issued meanwhile
      cyc += c / nw;
      lo = c < lo ? c : lo;
      hi = c > hi ? c : hi;
      mhz += (double)h[2 * i] / ((double)h[2 * i + 1] / 100e6) / 1e6 / nw;
    }
    printf("  %d/SIMD: %.2f (%.2f-%.2f) @%4.0f MHz", w, cyc, lo, hi, mhz);
  }
  printf("\n");
  (void)hipFree(out);
  (void)hipFree(clk);
}

int main() {
  run<0>("v_fma_f32 v,v,v");
  run<7>("v_add_f32 v,v");
  run<8>("v_mov_b32");
  run<1>("v_pk_fma_f32 acc,b,c (acc in place)");
  run<2>("v_pk_fma_f32 with op_sel / neg_lo (cmul form)");
  run<6>("v_pk_fma_f32 three different VGPR pairs");
  run<5>("v_pk_fma_f32 with an SGPR-pair operand");
  run<3>("v_pk_mul_f32 op_sel_hi (cmul form)");
  run<4>("v_pk_add_f32 acc,b");
  run<9>("v_pk_add_f32 two other VGPR pairs");
  return 0;
}
