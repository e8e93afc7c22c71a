// valu_rate.hip -- issue rate of the vector instructions the FFT kernels are made of, per SIMD, on the device at hand.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Every wave runs a loop of 8 independent instructions of one kind (no dependent issue stalls), 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed) {
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
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
double run(const char *name) {
  int dev = 0;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, dev);
  int const cus = p.multiProcessorCount, iters = 20000;
  float *out;
  hipMalloc(&out, 4096);
  dim3 grid(cus * 4), block(512);   // 4 workgroups of 8 waves per CU = 8 waves per SIMD
  hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, 100, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double const wave_instrs_per_simd = 8.0 * iters * 8;    // 8 waves x iters x 8 instructions
  double const clk = p.clockRate * 1e3;                   // Hz
  double const cyc = ms * 1e-3 * clk / wave_instrs_per_simd;
  printf("%-46s %.3f ms  %.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", name, ms, cyc, clk / 1e6);
  hipFree(out);
  return cyc;
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
