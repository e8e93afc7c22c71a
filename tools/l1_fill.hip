// l1_fill.hip -- how fast one CU pulls an L2-resident window through its vector L1, by load width.
//   hipcc -O3 --offload-arch=gfx950 tools/l1_fill.hip -o /tmp/l1_fill && /tmp/l1_fill
// Every workgroup (512 threads, 2 per CU like k_filter_full16k) reads the same 128 KiB window, 32 x 8 bytes or 16 x 16
// bytes or 8 x 32 bytes (two dwordx4) per thread, all loads issued back to back, and reports the shader cycles from the
// first issue to the last arrival (mean over workgroups of wave 0's count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WIDTH>
__global__ __launch_bounds__(512, 4) void k(const float *__restrict__ win, float *out, unsigned long long *cyc, int rounds) {
  __shared__ float pad[16 * 1024];   // 64 KiB: two workgroups per CU, as in the filter kernel
  int const t = threadIdx.x;
  float acc = 0;
  unsigned long long total = 0;
  for (int r = 0; r < rounds; r++) {
    __syncthreads();
    const float *w = win + ((r & 3) << 15);   // four windows in turn (else the loads are hoisted out of the loop)
    unsigned long long const c0 = clock64();
    if (WIDTH == 8) {
      v2f x[32];
#pragma unroll
      for (int i = 0; i < 32; i++) x[i] = *reinterpret_cast<const v2f *>(w + 2 * (512 * i + t));
#pragma unroll
      for (int i = 0; i < 32; i++) asm volatile("" ::"v"(x[i]));   // every value has arrived before the clock is read
#pragma unroll
      for (int i = 0; i < 32; i++) acc += x[i].x * x[i].y;
    } else {
      v4f x[16];
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = *reinterpret_cast<const v4f *>(w + 4 * (512 * i + t));
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("" ::"v"(x[i]));
#pragma unroll
      for (int i = 0; i < 16; i++) acc += x[i].x * x[i].y + x[i].z * x[i].w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long const c1 = clock64();
    asm volatile("" : "+v"(acc));
    total += c1 - c0;
    // something to do between the rounds, so that the two workgroups of a CU drift apart as real ones do
    for (int j = 0; j < 200 + 37 * (blockIdx.x & 7); j++) acc = acc * 1.0001f + 0.5f;
    pad[t] = acc;
  }
  if (t == 0) cyc[blockIdx.x] = total;
  if (acc == 12345.f) out[t] = acc + pad[(t * 7) & 1023];
}

template <int WIDTH>
void run(const char *name, const float *win, int cus) {
  float *out;
  unsigned long long *cyc;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&cyc, 8 * 2 * cus);
  int const rounds = 200;
  hipLaunchKernelGGL(k<WIDTH>, dim3(2 * cus), dim3(512), 0, 0, win, out, cyc, 20);
  hipLaunchKernelGGL(k<WIDTH>, dim3(2 * cus), dim3(512), 0, 0, win, out, cyc, rounds);
  std::vector<unsigned long long> h(2 * cus);
  (void)hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
  double m = 0;
  for (auto v : h) m += (double)v / rounds / h.size();
  printf("%-24s %7.0f cycles per 128 KiB window = %.1f bytes per cycle per workgroup\n", name, m, 131072.0 / m);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  float *win;
  (void)hipMalloc(&win, 1 << 20);
  (void)hipMemset(win, 0, 1 << 20);
  run<8>("32 x dwordx2 per thread", win, p.multiProcessorCount);
  run<16>("16 x dwordx4 per thread", win, p.multiProcessorCount);
  return 0;
}
