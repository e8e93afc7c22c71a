// handover_probe.hip -- what would the hand-over of a SHARED first radix-4 stage cost at N = 65536 (cfg 5)?
//   hipcc -O3 --offload-arch=gfx950 tools/handover_probe.hip -o /tmp/handover_probe && /tmp/handover_probe
//
// Today each of a channel-block's four sibling workgroups reads the whole 512 KiB window (64 sixteen-byte loads per thread,
// all L2 hits: the window is shared by every channel) and forms its own residue class.  The arrangement VERDICT r5 #1 asks
// for: a sibling loads a quarter of the rows (16 loads), forms all four classes of those rows and hands three of them to
// its siblings through memory (12 sixteen-byte stores), then collects its own class's other rows from them (12 loads) --
// software-pipelined over the blocks of a channel (produce block b + 1, consume block b, so nobody waits for a producer),
// the four siblings on ONE XCD (workgroup ids congruent mod 8) so that the hand-over can stay in that XCD's L2.
// This program runs only the MEMORY side of both arrangements, with a stand-in for the 16384-point body (a chain of packed
// multiply-adds sized to the body's instruction count), at cfg 5's shape (512 channels x 16 blocks), two 512-thread
// workgroups of 68 KiB LDS per CU as the filter kernel has:
//   mode 0  body only
//   mode 1  body + today's front: 64 sixteen-byte loads per thread of the shared window
//   mode 2  body + the pipelined hand-over: 16 window loads, 12 stores into the channel's pool, a counter, 12 loads of what the
//           siblings stored one block earlier -- in the form MI355X_MICROARCH.md measures as valid between workgroups on any
//           XCDs: `sc1` stores, every storing wave's vmcnt(0), a workgroup barrier, ONE relaxed agent-scope flag store; the
//           consumer polls with relaxed agent-scope loads (as kq_full16k.hip's sibling_exchange does), a workgroup barrier,
//           `sc1` loads of the rows
//   mode 3  the same through the XCD's own L2 (the siblings ARE on one XCD): plain stores (the vector L1 writes through), loads
//           that bypass the consumer's L1 (`sc0`); nothing the memory model promises, the best case of "the hand-over lives in L2"
// (The first version of this probe released with __threadfence() and polled with ACQUIRE loads: every poll invalidates the
// poller's L2, and 6 000 pollers took the launch from 1 ms to 14 -- the guide's "255 pollers cut chip bandwidth 37-71 %".)
// Prints the time per launch of each.  (mode 2) - (mode 0) against (mode 1) - (mode 0) is what the hand-over costs
// against what it saves on the memory side; the vector instructions a shared stage saves (15 -> 6.5 packed per output)
// are not in here and are credited separately.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kT = 512, kBlocks = 16, kRowBytes = 512 * 16;  // a row pair: 512 threads x 16 bytes
constexpr size_t kWinBytes = 512 * 1024;                      // one block's window (N = 65536 complex)
constexpr size_t kClassBytes = 128 * 1024;                    // one residue class of one channel-block

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__device__ __forceinline__ v4f body(v4f acc, int n) {
  // the 16384-point body's vector work: ~1 100 instructions per wave, dependent enough not to be collapsed
  v2f a = (v2f){acc.x, acc.y}, b = (v2f){acc.z, acc.w};
  v2f const k1 = (v2f){1.0001f, 0.9999f}, k2 = (v2f){0.25f, -0.25f};
  for (int i = 0; i < n; i++) {
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(k1), "v"(k2));
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(k1), "v"(k2));
  }
  return (v4f){a.x, a.y, b.x, b.y};
}

template <int MODE>
__global__ __launch_bounds__(kT, 4) void k_probe(const char *__restrict__ win, char *__restrict__ pool, unsigned *__restrict__ flags,
                                                 float *__restrict__ out, int nchan, int body_n, unsigned base, unsigned *__restrict__ lost) {
  extern __shared__ float lds[];  // 68 KiB: two workgroups per CU
  int const t = threadIdx.x;
  // siblings of a channel: workgroup ids congruent mod 8 (one XCD).  id = 8 * (4 * (chan / 8) + sibling) + chan % 8
  int const wg = blockIdx.x, xcd = wg & 7, local = wg >> 3, S = local & 3, chan = (local >> 2) * 8 + xcd;
  if (chan >= nchan) return;
  v4f acc = (v4f){1.f, 2.f, 3.f, 4.f};
  char *const mypool = pool + (size_t)chan * 2 * 4 * kClassBytes;   // [slot][class][16 row pairs][512][16 B]
  unsigned *const myflags = flags + (size_t)chan * 2 * 4;           // [slot][sibling]: launch tag + block, relaxed agent-scope words
  // (compiler builtins, not inline assembly: the compiler then counts the loads in flight itself.  aux bit 4 = sc1, bit 0 = sc0)
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  rsrc_t const pr = __builtin_amdgcn_make_buffer_rsrc(mypool, 0, (int)(2 * 4 * kClassBytes), 0x00020000);
  auto st16 = [&](unsigned off, v4f v) {
    if (MODE == 2)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), pr, (int)off, 0, 16);
    else
      *reinterpret_cast<v4f *>(mypool + off) = v;
  };
  auto ld16 = [&](unsigned off) {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(pr, (int)off, 0, MODE == 2 ? 16 : 1));
  };
  auto produce = [&](int b) {  // rows 4 S .. 4 S + 3 (row pairs) of all four quarters of block b's window; three classes out
    const char *w = win + (size_t)(b & (kBlocks - 1)) * kWinBytes;
    v4f x[16];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) x[4 * j + i] = *reinterpret_cast<const v4f *>(w + (size_t)(16 * j + 4 * S + i) * kRowBytes + t * 16);
    v4f y[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {  // a radix-4 worth of adds per row pair
      v4f const p = x[i] + x[8 + i], q = x[4 + i] + x[12 + i], r = x[i] - x[8 + i], s = x[4 + i] - x[12 + i];
      acc += p + q;
      y[i] = p - q;
      y[4 + i] = r + s;
      y[8 + i] = r - s;
    }
    unsigned const slot = (unsigned)(b & 1) * 4u * (unsigned)kClassBytes;
#pragma unroll
    for (int n = 0; n < 3; n++) {
      int const cls = n + (n >= S ? 1 : 0);  // the three classes that are not this sibling's
#pragma unroll
      for (int i = 0; i < 4; i++) st16(slot + (unsigned)cls * (unsigned)kClassBytes + (unsigned)(4 * S + i) * kRowBytes + t * 16, y[4 * n + i]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, before the barrier its signalling lane joins
    __syncthreads();
    if (t == 0) __hip_atomic_store(myflags + (b & 1) * 4 + S, base + (unsigned)b + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto consume = [&](int b) {  // this class's rows from the three siblings
    if (t < 4 && t != S) {  // relaxed polls (the product's sibling_exchange): no acquire, nothing invalidated
      unsigned const want = base + (unsigned)b + 1u;
      int it = 0;
      while (__hip_atomic_load(myflags + (b & 1) * 4 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want && ++it < (1 << 17))
        __builtin_amdgcn_s_sleep(8);
      if (it >= (1 << 17)) atomicAdd(lost, 1u);  // (a sibling that never came: counted, never waited for without end)
    }
    __syncthreads();
    unsigned const slot = (unsigned)(b & 1) * 4u * (unsigned)kClassBytes + (unsigned)S * (unsigned)kClassBytes;
    v4f x[12];
#pragma unroll
    for (int n = 0; n < 3; n++) {
      int const sib = n + (n >= S ? 1 : 0);
#pragma unroll
      for (int i = 0; i < 4; i++) x[4 * n + i] = ld16(slot + (unsigned)(4 * sib + i) * kRowBytes + t * 16);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) acc += x[i];
  };
  if (MODE >= 2) produce(0);
  for (int b = 0; b < kBlocks; b++) {
    if (MODE == 1) {  // today: the whole window, 64 loads
      const char *w = win + (size_t)b * kWinBytes;
#pragma unroll
      for (int half = 0; half < 4; half++) {
        v4f x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = *reinterpret_cast<const v4f *>(w + (size_t)(16 * half + i) * kRowBytes + t * 16);
#pragma unroll
        for (int i = 0; i < 16; i++) acc += x[i];
      }
    }
    if (MODE >= 2) {
      // consume first: a sibling can then never store block b + 2 into the slot block b still sits in (its produce(b + 2)
      // follows its consume(b + 1), which needs everybody's produce(b + 1), which follows everybody's consume(b)) -- two
      // slots are enough, and what is consumed here was stored a whole body ago
      consume(b);
      if (b + 1 < kBlocks) produce(b + 1);
    }
    acc = body(acc, body_n);
    lds[t] = acc.x;
    __syncthreads();
    acc.y += lds[(t + 64) & (kT - 1)];
    __syncthreads();
  }
  if (acc.x == 12345.678f) out[wg * kT + t] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char **argv) {
  int const nchan = argc > 1 ? atoi(argv[1]) : 512;
  int const body_n = argc > 2 ? atoi(argv[2]) : 280;  // 2 x 280 packed FMAs ~ the body's issue time (packed = two issue slots)
  char *win, *pool;
  unsigned *flags;
  float *out;
  unsigned *lost;
  CHECK(hipMalloc(&lost, sizeof(unsigned)));
  CHECK(hipMemset(lost, 0, sizeof(unsigned)));
  CHECK(hipMalloc(&win, kBlocks * kWinBytes));
  CHECK(hipMalloc(&pool, (size_t)nchan * 2 * 4 * kClassBytes));
  CHECK(hipMalloc(&flags, (size_t)nchan * 2 * 4 * sizeof(unsigned)));
  CHECK(hipMalloc(&out, (size_t)nchan * 4 * kT * sizeof(float)));
  CHECK(hipMemset(win, 0, kBlocks * kWinBytes));
  CHECK(hipMemset(pool, 0, (size_t)nchan * 2 * 4 * kClassBytes));
  CHECK(hipMemset(flags, 0, (size_t)nchan * 2 * 4 * sizeof(unsigned)));
  size_t const lds = 68 * 1024;
  CHECK(hipFuncSetAttribute((const void *)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute((const void *)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute((const void *)k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute((const void *)k_probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  unsigned base = 0;  // launch tag of the flag words
  int const grid = ((nchan + 7) / 8) * 8 * 4;
  auto run = [&](int mode, int reps) {
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps + 3; r++) {
      CHECK(hipEventRecord(e0, 0));
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(kT), lds, 0, win, pool, flags, out, nchan, body_n, base, lost);
      if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(kT), lds, 0, win, pool, flags, out, nchan, body_n, base, lost);
      if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(grid), dim3(kT), lds, 0, win, pool, flags, out, nchan, body_n, base, lost);
      if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(grid), dim3(kT), lds, 0, win, pool, flags, out, nchan, body_n, base, lost);
      if (mode >= 2) base += 64u;
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 3) {
        sum += ms;
        if (ms < best) best = ms;
      }
    }
    printf("mode %d: %.4f ms mean, %.4f best per launch (%d channels x %d blocks x 4 siblings, body %d)\n", mode, sum / reps, best, nchan,
           kBlocks, body_n);
    return sum / reps;
  };
  for (int r = 0; r < 20; r++) run(1, 1);  // clocks up
  float const t0 = run(0, 20), t1 = run(1, 20), t2 = run(2, 20), t3 = run(3, 20);
  unsigned nlost = 0;
  CHECK(hipMemcpy(&nlost, lost, sizeof nlost, hipMemcpyDeviceToHost));
  printf("waits that ran out (must be 0): %u\n", nlost);
  printf("front of today (64 window loads per thread):   +%.4f ms over the body alone\n", t1 - t0);
  printf("pipelined hand-over, sc1 stores and loads:      +%.4f ms over the body alone\n", t2 - t0);
  printf("pipelined hand-over through the XCD's L2:       +%.4f ms over the body alone\n", t3 - t0);
  printf("hand-over bytes through the pool per launch: %.2f GB written, as much read; pool footprint %.0f MiB, live per XCD at a time "
         "(16 channel-blocks in flight): %.1f MiB against 4 MiB of L2\n",
         (double)nchan * kBlocks * 3 * kClassBytes / 1e9, (double)nchan * 2 * 4 * kClassBytes / 1048576.0, 16.0 * 2 * 3 * kClassBytes / 1048576.0);
  return 0;
}
