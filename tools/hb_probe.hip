// hb_probe.hip -- what the LOAD STRUCTURE of the half-band group kernel (kq_decim.hip) costs with no filter in it:
// read n complex samples (8 B each), write n/8 of them, through the alternatives the kernel could use.
//   hipcc -O3 --offload-arch=gfx950 tools/hb_probe.hip -o tools/hb_probe.bin ;  gpurun -- tools/hb_probe.bin
// Variants (all read 512 MiB and write 64 MiB per launch at the default size):
//   copy      grid of one-shot workgroups, 8 x 16 B per thread added up, one 16-B store        (the byte mix's ceiling)
//   copy NNNthr xU [nt]   the same with U loads in flight per thread, plain or nontemporal
//   tile ... nolds        persistent workgroups walking 32 KiB tiles (+ a misaligning halo in front) as k_hb_group does,
//                         next tile's loads issued before the current one is consumed, registers only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ in, float4 *__restrict__ out) {
  size_t const o = (size_t)blockIdx.x * 256 + threadIdx.x;
  const float4 *p = in + (size_t)blockIdx.x * 2048 + threadIdx.x;
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = p[i * 256];
  float4 s = v[0];
#pragma unroll
  for (int i = 1; i < 8; i++) {
    s.x += v[i].x;
    s.y += v[i].y;
    s.z += v[i].z;
    s.w += v[i].w;
  }
  out[o] = s;
}

// the same byte mix with U loads of 16 B in flight per thread, optionally nontemporal, T threads per workgroup
template <int T, int U, bool NT>
__global__ __launch_bounds__(T) void k_copy_u(const float4 *__restrict__ in, float4 *__restrict__ out) {
  const float4 *p = in + (size_t)blockIdx.x * (T * U) + threadIdx.x;
  float4 v[U];
#pragma unroll
  for (int i = 0; i < U; i++) {
    if constexpr (NT) {
      typedef float vf4 __attribute__((ext_vector_type(4)));
      vf4 const w = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p + i * T));
      v[i] = make_float4(w.x, w.y, w.z, w.w);
    } else
      v[i] = p[i * T];
  }
#pragma unroll
  for (int g = 0; g < U / 8; g++) {
    float4 s = v[8 * g];
#pragma unroll
    for (int i = 1; i < 8; i++) {
      s.x += v[8 * g + i].x;
      s.y += v[8 * g + i].y;
      s.z += v[8 * g + i].z;
      s.w += v[8 * g + i].w;
    }
    out[((size_t)blockIdx.x * (U / 8) + g) * T + threadIdx.x] = s;
  }
}

// T threads, tile of TILE float4 (TILE * 16 bytes), HALO float4 in front (misaligns the start as the real halo does)
template <int T, int TILE, int HALO, bool PERSIST, bool LDS, int NBAR>
__global__ __launch_bounds__(T) void k_tile(const float4 *__restrict__ in, float4 *__restrict__ out, long long ntiles) {
  extern __shared__ float4 lds[];
  constexpr int LEN = TILE + HALO;
  constexpr int IT = (LEN + T - 1) / T;
  constexpr int FULL = TILE / T;
  float4 v[IT];
  int const tid = threadIdx.x;
  auto fetch = [&](long long t) {
    const float4 *src = in + t * TILE + (HALO ? (TILE - HALO) : 0);  // never before the buffer: shifted one tile up
#pragma unroll
    for (int it = 0; it < IT; it++) {
      int const i = it < FULL ? it * T + tid : min(it * T + tid, LEN - 1);
      v[it] = src[i];
    }
  };
  long long t = blockIdx.x;
  long long const step = PERSIST ? gridDim.x : ntiles;
  if (t < ntiles) fetch(t);
  for (; t < ntiles; t += step) {
    if constexpr (LDS) {
#pragma unroll
      for (int it = 0; it < IT; it++)
        if (it < FULL || it * T + tid < LEN) lds[it * T + tid] = v[it];
      __syncthreads();
      if (t + step < ntiles) fetch(t + step);
#pragma unroll
      for (int b = 0; b < NBAR; b++) {
        // a token LDS round trip per stage so that the barriers are not folded away
        if (tid == b) lds[LEN + b] = lds[tid];
        __syncthreads();
      }
      if (tid < TILE / 8) {
        float4 s = lds[HALO + tid];
#pragma unroll
        for (int i = 1; i < 8; i++) {
          float4 const w = lds[HALO + i * (TILE / 8) + tid];
          s.x += w.x;
          s.y += w.y;
          s.z += w.z;
          s.w += w.w;
        }
        out[t * (TILE / 8) + tid] = s;
      }
      __syncthreads();
    } else {
      float4 s = v[0];
#pragma unroll
      for (int it = 1; it < IT; it++) {
        s.x += v[it].x;
        s.y += v[it].y;
        s.z += v[it].z;
        s.w += v[it].w;
      }
      if (t + step < ntiles) fetch(t + step);
      if (tid < TILE / 8) out[t * (TILE / 8) + tid] = s;
    }
  }
}

template <class F>
static float time_it(const char *name, F launch, double bytes, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 300; i++) launch();
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-34s %8.4f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
  fflush(stdout);
  return ms;
}

int main(int argc, char **argv) {
  size_t const n4 = (size_t)(argc > 1 ? atoll(argv[1]) : (64ll << 20)) / 2;  // float4 = 2 complex samples
  int const reps = 100;
  float4 *in, *out;
  CHECK(hipMalloc(&in, (n4 + 4096) * 16));
  CHECK(hipMalloc(&out, n4 / 8 * 16 + 4096));
  CHECK(hipMemset(in, 0, (n4 + 4096) * 16));
  double const bytes = n4 * 16.0 + n4 / 8 * 16.0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  unsigned const cus = prop.multiProcessorCount;

  time_it("copy (one-shot WGs)", [&] { hipLaunchKernelGGL(k_copy, dim3(n4 / 2048), dim3(256), 0, 0, in, out); }, bytes, reps);

#define COPY_VARIANT(name, T, U, NT) \
  time_it(name, [&] { hipLaunchKernelGGL((k_copy_u<T, U, NT>), dim3(n4 / ((T) * (U))), dim3(T), 0, 0, in, out); }, bytes, reps);
  COPY_VARIANT("copy 256thr x8", 256, 8, false)
  COPY_VARIANT("copy 256thr x8 nt", 256, 8, true)
  COPY_VARIANT("copy 256thr x16", 256, 16, false)
  COPY_VARIANT("copy 256thr x16 nt", 256, 16, true)
  COPY_VARIANT("copy 512thr x8", 512, 8, false)
  COPY_VARIANT("copy 1024thr x8", 1024, 8, false)
  COPY_VARIANT("copy 512thr x16 nt", 512, 16, true)
  COPY_VARIANT("copy 64thr x8", 64, 8, false)
  COPY_VARIANT("copy 64thr x16 nt", 64, 16, true)

#define TILE_VARIANT(name, T, TILE, HALO, PERSIST, LDS, NBAR, WGS_PER_CU)                                         \
  {                                                                                                                \
    long long const ntiles = (long long)(n4 / (TILE)) - 1;                                                         \
    size_t const ldsb = 160 * 1024 / (WGS_PER_CU) - 2048; /* pins the occupancy, used or not */                   \
    unsigned const grid = (PERSIST) ? (unsigned)std::min<long long>(ntiles, (long long)cus * (WGS_PER_CU)) : (unsigned)ntiles; \
    auto kern = k_tile<T, TILE, HALO, PERSIST, LDS, NBAR>;                                                         \
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));        \
    time_it(name, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(T), ldsb, 0, in, out, ntiles); }, bytes, reps);  \
  }
  // 32 KiB tiles = 2048 float4; the real kernel's halo is 98 samples = 49 float4; it has 50 KB of LDS -> 3 WGs per CU.
  // Registers only (the LDS staging of the real kernel is not what this probe is about)
  TILE_VARIANT("tile 512thr halo49 nolds x3", 512, 2048, 49, true, false, 0, 3)
  TILE_VARIANT("tile 512thr halo49 nolds x4", 512, 2048, 49, true, false, 0, 4)
  return 0;
}
