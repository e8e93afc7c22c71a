// kq_decim.hip -- half-band decimator cascade of the front-end daemons on gfx950.
//
// Replaces decimate.c:108-160 (hb15_block / hb3_block, portable branch) as hackrf.c:260-330 drives them: Fs/4
// rotation of the raw samples, Log_decimate half-band stages (the first ones, j >= stage_threshold, with the 1-2-1
// filter, the rest with the 15-tap Goodman/Carey F8), scaling by Filter_atten, conversion to int16 and the output
// energy.  I and Q go through the same real filter, so a complex sample is one float2 element.
//
// Every stage is a FIR, so instead of the reference's per-stage shift registers the carried state is the tail of
// each kernel's INPUT (zero at start, exactly like the zeroed states of hackrf.c:211-216): a workgroup re-derives
// the few intermediate samples it needs from that halo.  Up to four stages (decimate by 16) are fused in one
// kernel, staged through LDS, so HBM sees the input once plus 1/16 + 1/16 of it between kernels.  The arithmetic
// keeps the reference's operand order with FMA contraction switched off, so the
// result is bit-identical to a scalar C evaluation of decimate.c.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "ka9q_hip.h"
#include "kq_device.hpp"

// hipcc contracts a*b+c into an FMA by default (and its __fmul_rn/__fadd_rn are plain operators), which would change
// the rounding with respect to decimate.c built for x86-64: switch contraction off for this translation unit.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }

constexpr int kMaxFuse = 4;       // stages per kernel
// outputs of a fused group per workgroup: the level-0 tile is 4096 samples (+ halo) either way, i.e. 50 KB of LDS with
// level 1 and three workgroups per CU
__host__ __device__ constexpr int tile_out(int nstages) { return nstages >= 4 ? 256 : 512; }
constexpr int kThreads = 512;     // (256 until round 3: the same 50 KB tile shared by eight waves instead of four, at most
                                  //  80 registers each so that three workgroups fit a CU, is 4 % faster; 1024 threads 9 % slower)

struct GroupArgs {
  const float2 *in;    // n_in = n_out << nstages complex samples
  const float2 *hist;  // the `halo` samples preceding in[0]
  float2 *out;         // n_out complex samples (scaled by `scale` when final)
  int16_t *out16;      // final group only, may be null: interleaved I,Q int16
  unsigned long long *partial;  // final group only: per-workgroup energy, tagged: epoch << 32 | float bits
  float *energy_out;   // final group only, may be null: the call's output energy, added up by the edge workgroup
  int *err;            // pinned host word, set when a workgroup's energy never arrived
  unsigned epoch;      // this call's tag
  long long n_out;
  int nstages;
  unsigned hb15_mask;  // bit s set: stage s of this group is the 15-tap filter, else 1-2-1
  int halo;            // input history needed by output 0
  int rot_step;        // Fs/4 rotation (first group only): phase(i) = (rot_phase0 + i*rot_step) & 3
  int rot_phase0;
  int rotate;
  int in_once;         // `in` is the caller's buffer, read once: nontemporal loads
  int final;
  float scale;
  float c0, c1, c2, c3;
};

// The caller's samples are read once (plus a tile's 98-sample halo): nontemporal loads, which leave the cache to the
// lines the stores are being merged in.  Measured on the byte mix of the first group (tools/hb_probe.hip): 6.35 against
// 5.4 TB/s.  The later groups read what the group before them has just written (an eighth of the size, normally still in
// the Infinity Cache) with plain loads: nontemporal there was 2 % slower on the whole call.
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 load_once(const float4 *p) {
  vf4 const w = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p));
  return make_float4(w.x, w.y, w.z, w.w);
}
__device__ __forceinline__ float2 load_once(const float2 *p) {
  vf2 const w = __builtin_nontemporal_load(reinterpret_cast<const vf2 *>(p));
  return make_float2(w.x, w.y);
}

__device__ __forceinline__ float2 rot90(float2 v, int phase) {
  // hackrf.c:272-289: multiply by j^phase -- an exact swap and sign flip, done without branches
  bool const sw = phase & 1;
  unsigned const ax = __float_as_uint(sw ? v.y : v.x), ay = __float_as_uint(sw ? v.x : v.y);
  unsigned const negx = ((phase + 1) & 2) << 30;  // phases 1, 2
  unsigned const negy = (phase & 2) << 30;        // phases 2, 3
  return make_float2(__uint_as_float(ax ^ negx), __uint_as_float(ay ^ negy));
}

__device__ __forceinline__ float hb15_tap(float acc, float a, float b, float c) {
  // decimate.c:128: result += (odd[i] + old_odd[i]) * coeffs[i], unfused
  return add_rn(acc, mul_rn(add_rn(a, b), c));
}

// A level lives in LDS de-interleaved: local sample i is plane[i & 1][i >> 1] (local index 0 = first history sample).
// Both filters then read with unit stride across lanes: output k of the 15-tap filter needs the odd-plane samples
// k..k+7 and the even-plane sample k+4, the 1-2-1 filter even k, odd k, even k+1.  A thread produces the output pair
// (2p, 2p+1), which shares 7 of its 8 odd taps and lands on element p of each plane of the next level.
struct Level {
  float2 *even, *odd;
};

struct Taps15 {
  float4 o[5], e;
};
struct Taps3 {
  float2 e0, o0, e1, o1, e2;
};

__device__ __forceinline__ Taps15 read15(Level src, int p) {
  Taps15 t;
  const float4 *o4 = reinterpret_cast<const float4 *>(src.odd + 2 * p);
#pragma unroll
  for (int m = 0; m < 5; m++) t.o[m] = o4[m];
  t.e = *reinterpret_cast<const float4 *>(src.even + 2 * p + 4);
  return t;
}

__device__ __forceinline__ void filter15(const Taps15 &t, const GroupArgs &a, float2 &r0, float2 &r1) {
  float2 o[10];
#pragma unroll
  for (int m = 0; m < 5; m++) {
    o[2 * m] = make_float2(t.o[m].x, t.o[m].y);
    o[2 * m + 1] = make_float2(t.o[m].z, t.o[m].w);
  }
  float2 r[2] = {make_float2(t.e.x, t.e.y), make_float2(t.e.z, t.e.w)};
#pragma unroll
  for (int j = 0; j < 2; j++) {
    // output k = 2p+j: o[k-i] is o[j+7-i], o[k-7+i] is o[j+i]
    r[j].x = hb15_tap(r[j].x, o[j + 7].x, o[j].x, a.c0);
    r[j].y = hb15_tap(r[j].y, o[j + 7].y, o[j].y, a.c0);
    r[j].x = hb15_tap(r[j].x, o[j + 6].x, o[j + 1].x, a.c1);
    r[j].y = hb15_tap(r[j].y, o[j + 6].y, o[j + 1].y, a.c1);
    r[j].x = hb15_tap(r[j].x, o[j + 5].x, o[j + 2].x, a.c2);
    r[j].y = hb15_tap(r[j].y, o[j + 5].y, o[j + 2].y, a.c2);
    r[j].x = hb15_tap(r[j].x, o[j + 4].x, o[j + 3].x, a.c3);
    r[j].y = hb15_tap(r[j].y, o[j + 4].y, o[j + 3].y, a.c3);
  }
  r0 = r[0];
  r1 = r[1];
}

__device__ __forceinline__ Taps3 read3(Level src, int p) {
  return Taps3{src.even[2 * p], src.odd[2 * p], src.even[2 * p + 1], src.odd[2 * p + 1], src.even[2 * p + 2]};
}

__device__ __forceinline__ void filter3(const Taps3 &t, float2 &r0, float2 &r1) {
  // decimate.c:155: 2*in[2k] + in[2k+1] + in[2k-1]; in[2k-1] is local sample 2k
  r0.x = add_rn(add_rn(mul_rn(2.f, t.o0.x), t.e1.x), t.e0.x);
  r0.y = add_rn(add_rn(mul_rn(2.f, t.o0.y), t.e1.y), t.e0.y);
  r1.x = add_rn(add_rn(mul_rn(2.f, t.o1.x), t.e2.x), t.e1.x);
  r1.y = add_rn(add_rn(mul_rn(2.f, t.o1.y), t.e2.y), t.e1.y);
}

// One stage: every thread takes output pairs p, p + kThreads, ... two at a time, reading the taps of both before
// filtering either so that the second LDS round trip hides behind the arithmetic of the first.
template <bool HB15, class Emit>
__device__ __forceinline__ void run_stage(Level src, int n_prod, const GroupArgs &a, Emit emit) {
  int const npairs = (n_prod + 1) >> 1;
  for (int p = threadIdx.x; p < npairs; p += kThreads) {
    float2 r0, r1;
    if constexpr (HB15) {
      Taps15 const ta = read15(src, p);
      filter15(ta, a, r0, r1);
    } else {
      Taps3 const ta = read3(src, p);
      filter3(ta, r0, r1);
    }
    emit(p, r0, r1);
  }
}

// plane capacity (float2 elements, multiple of 2 so that every plane stays 16-byte aligned) for a level of n samples;
// the slack covers the over-read of the last output pair
__host__ __device__ constexpr int plane_cap(int n) { return ((n + 1) / 2 + 12) & ~1; }

template <int N>
using ic = std::integral_constant<int, N>;

// Persistent workgroups: each walks tiles blockIdx.x, +gridDim.x, ... and issues the global loads of its next tile
// (held in registers) before it starts filtering the current one, so HBM requests stay in flight during the LDS phases.
// G = stages in this group; FINAL = last group (Filter_atten, int16, energy); ROT = Fs/4 rotation on the way in.
// A thread loads two neighbouring samples (16 bytes) that land on the same element of the two planes.  That needs an
// even number of history samples in front of the tile: when the group's halo is odd (its first stage is then the 1-2-1
// filter) the tile starts one sample earlier, which swaps the roles of the planes for stage 0 (`pad` below).
// EDGE = this workgroup takes the tiles that need care -- the first (history buffer) and a ragged last one -- with
// clamped indices and a pointer select; the other workgroups share the full tiles in between and need neither.
template <int G, bool FINAL, bool ROT, bool EDGE>
__device__ __forceinline__ void hb_group_body(const GroupArgs &a, float2 *lds) {
  // halo_s: history (in level-s samples) that level s must hold ahead of a tile's first output
  int halo[G + 1];
  halo[G] = 0;
#pragma unroll
  for (int s = G - 1; s >= 0; s--) halo[s] = 2 * halo[s + 1] + ((a.hb15_mask >> s) & 1 ? 14 : 1);

  constexpr int kTileOut = tile_out(G);
  int const pad = halo[0] & 1;
  int const cap0 = plane_cap((kTileOut << G) + halo[0] + pad);
  int const cap1 = G > 1 ? plane_cap((kTileOut << (G - 1)) + halo[G > 1 ? 1 : 0]) : 0;
  Level const lvA{lds, lds + cap0};                        // level 0 as loaded (and level 2)
  Level const lvB{lds + 2 * cap0, lds + 2 * cap0 + cap1};  // level 1 (and level 3)
  // level 0 as stage 0 sees it, local sample 0 = the first history sample it needs: with the pad sample in front, the
  // even samples sit in the odd plane and the odd ones in the even plane, one element up
  Level const lv0 = pad ? Level{lvA.odd, lvA.even + 1} : lvA;

  long long const ntiles = (a.n_out + kTileOut - 1) / kTileOut;
  constexpr int kMaxLen0 = (kTileOut << G) + 14 * ((1 << G) - 1);
  constexpr int kPer = 2;
  constexpr int kLoadIters = (kMaxLen0 + kPer * kThreads - 1) / (kPer * kThreads);
  constexpr int kFullIters = (kTileOut << G) / (kPer * kThreads);  // always inside a full tile, whatever the halo
  using LoadT = float4;
  LoadT v[kLoadIters];
  int const tid = threadIdx.x;

  // Level 0 of tile t as loaded: global input index = t * kTileOut * 2^G - halo[0] - pad + i, i < (tile << G) + halo[0]
  // + pad (an even count from an even index).  All loads of a thread are issued back to back.
  auto fetch = [&](long long t) {
    long long const first_out = t * kTileOut;
    long long const lo0 = (first_out << G) - halo[0] - pad;
    if constexpr (!EDGE) {
      int const len0 = (kTileOut << G) + halo[0] + pad;
      const LoadT *src = reinterpret_cast<const LoadT *>(a.in + lo0) + tid;
#pragma unroll
      for (int it = 0; it < kLoadIters; it++) {
        if (it < kFullIters)
          v[it] = a.in_once ? load_once(src + it * kThreads) : src[it * kThreads];
        else {
          const LoadT *q = reinterpret_cast<const LoadT *>(a.in + lo0 + min((it * kThreads + tid) * kPer, len0 - kPer));
          v[it] = a.in_once ? load_once(q) : *q;
        }
      }
    } else {
      int const tile = (int)min((long long)kTileOut, a.n_out - first_out);
      int const len0 = (tile << G) + halo[0] + pad;
#pragma unroll
      for (int it = 0; it < kLoadIters; it++) {
        int const i = min((it * kThreads + tid) * kPer, len0 - kPer);
        long long const gi = lo0 + i;
        const float2 *src = gi >= 0 ? a.in + gi : a.hist + (a.halo + gi);  // hist[-1] exists (the pad sample, unused)
        v[it] = *reinterpret_cast<const LoadT *>(src);
      }
    }
  };

  // tile walk: the edge workgroup does tile 0 and then the ragged tile (if any); the others stride over 1..nfull-1
  long long const nfull = a.n_out / kTileOut;
  long long const first = EDGE ? 0 : (long long)blockIdx.x;  // blockIdx.x >= 1 here
  long long const step = EDGE ? (nfull > 0 && nfull < ntiles ? nfull : ntiles) : (long long)gridDim.x - 1;
  long long const end = EDGE ? ntiles : nfull;
  long long t = first;
  if (t < end) fetch(t);
  float energy = 0.f;  // FINAL: this thread's share over all of the workgroup's tiles
  for (; t < end; t += step) {
    long long const first_out = t * kTileOut;
    int const tile = (int)min((long long)kTileOut, a.n_out - first_out);
    int len[G + 1];
#pragma unroll
    for (int s = 0; s <= G; s++) len[s] = (tile << (G - s)) + halo[s];
    long long const lo0 = (first_out << G) - halo[0] - pad;
    // rotation phase of this thread's first sample; later iterations are a multiple of 4 samples further on
    int const ph = a.rot_phase0 + (int)((lo0 + tid * kPer) & 3) * a.rot_step;
    bool const full = !EDGE;
#pragma unroll
    for (int it = 0; it < kLoadIters; it++) {
      int const i = (it * kThreads + tid) * kPer;
      if ((it < kFullIters && full) || i < len[0] + pad) {
        float2 w0 = make_float2(v[it].x, v[it].y), w1 = make_float2(v[it].z, v[it].w);
        if constexpr (ROT) {
          w0 = rot90(w0, ph);
          w1 = rot90(w1, ph + a.rot_step);
        }
        lvA.even[it * kThreads + tid] = w0;
        lvA.odd[it * kThreads + tid] = w1;
      }
    }
    __syncthreads();
    if (t + step < end) fetch(t + step);

    auto do_stage = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < G) {
        constexpr bool last = s == G - 1;
        Level const src = s == 0 ? lv0 : (s & 1) ? lvB : lvA;
        Level const dst = (s & 1) ? lvA : lvB;
        int const n_prod = len[s + 1];
        auto emit = [&](int p, float2 r0, float2 r1) {
          if constexpr (!last) {
            dst.even[p] = r0;  // the element past an odd n_prod is slack
            dst.odd[p] = r1;
          } else {
            long long const go = first_out + 2 * p;
            bool const two = 2 * p + 1 < n_prod;
            if constexpr (FINAL) {
              // hackrf.c:307-311: s = sample * Filter_atten; energy += s*s; (short)round(32767 * s)
              r0.x = mul_rn(r0.x, a.scale);
              r0.y = mul_rn(r0.y, a.scale);
              r1.x = mul_rn(r1.x, a.scale);
              r1.y = mul_rn(r1.y, a.scale);
              energy = add_rn(energy, add_rn(mul_rn(r0.x, r0.x), mul_rn(r0.y, r0.y)));
              if (two) energy = add_rn(energy, add_rn(mul_rn(r1.x, r1.x), mul_rn(r1.y, r1.y)));
              if (a.out16) {
                short4 q;
                q.x = (int16_t)(int)roundf(mul_rn(32767.f, r0.x));
                q.y = (int16_t)(int)roundf(mul_rn(32767.f, r0.y));
                q.z = (int16_t)(int)roundf(mul_rn(32767.f, r1.x));
                q.w = (int16_t)(int)roundf(mul_rn(32767.f, r1.y));
                if (two)
                  *reinterpret_cast<short4 *>(a.out16 + 2 * go) = q;
                else
                  *reinterpret_cast<short2 *>(a.out16 + 2 * go) = make_short2(q.x, q.y);
              }
            }
            if (two)
              *reinterpret_cast<float4 *>(a.out + go) = make_float4(r0.x, r0.y, r1.x, r1.y);
            else
              a.out[go] = r0;
          }
        };
        if ((a.hb15_mask >> s) & 1)
          run_stage<true>(src, n_prod, a, emit);
        else
          run_stage<false>(src, n_prod, a, emit);
        // the last stage of an even group reads level G-1 in lvB and writes to memory: the next tile's level 0 (lvA) can
        // go in beside it, and the barrier behind that comes before anything writes lvB again
        if constexpr (!last || (s & 1) == 0) __syncthreads();
      }
    };
    do_stage(ic<0>{});
    do_stage(ic<1>{});
    do_stage(ic<2>{});
    do_stage(ic<3>{});
  }

  if constexpr (FINAL) {
    // The workgroup's output energy, once per launch (per tile it was one more barrier on every tile's critical path):
    // lanes by xor-shuffle, then waves in order.  Which tiles a workgroup takes depends only on the grid, so the sum is
    // reproducible on a given device.  Value and "this call's" go out in one word, so that the edge workgroup can take it
    // without a release fence here (which would write back the L2 under every workgroup's output stores) and without a
    // launch boundary.  Every workgroup publishes, also one that found no tile.
    __shared__ float wsum[kThreads / 64];
    for (int off = 32; off; off >>= 1) energy += __shfl_xor(energy, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = energy;
    __syncthreads();
    if (threadIdx.x == 0) {
      float e = 0;
      for (int w = 0; w < kThreads / 64; w++) e += wsum[w];
      __hip_atomic_store(a.partial + blockIdx.x, ((unsigned long long)a.epoch << 32) | __float_as_uint(e), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  if constexpr (EDGE) {
    // Carry for the next call: the last `halo` samples of hist ++ in replace the history, in place -- this
    // workgroup is the only one that reads it (tile 0), and it has.
    long long const n_in = a.n_out << G;
    float2 hv = make_float2(0.f, 0.f);
    if (tid < a.halo) {
      long long const gi = n_in - a.halo + tid;
      hv = gi >= 0 ? a.in[gi] : a.hist[a.halo + gi];
    }
    __syncthreads();
    if (tid < a.halo) const_cast<float2 *>(a.hist)[tid] = hv;

    if constexpr (FINAL) {
      // Output energy (hackrf.c:308,325): the workgroups' partials in a fixed order, whenever they arrive.  Nobody waits for this workgroup, so the others finish regardless and the wait below ends; the bound
      // only guards against a workgroup that died (a kernel cannot run longer than HBM lasts: 60 ms).
      if (a.energy_out) {
        __shared__ double dsum[kThreads / 64];
        double acc = 0;
        bool lost = false;
        for (int i = tid; i < (int)gridDim.x; i += kThreads) {
          unsigned long long w;
          int polls = 0;
          for (;;) {
            w = __hip_atomic_load(a.partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == a.epoch || ++polls > (1 << 20)) break;
            __builtin_amdgcn_s_sleep(8);
          }
          if ((unsigned)(w >> 32) != a.epoch) {
            lost = true;
            w = 0x7fc00000u;  // NaN
          }
          acc += (double)__uint_as_float((unsigned)w);
        }
        if (lost) *a.err = 1;
        for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
        if ((tid & 63) == 0) dsum[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
          double e = 0;
          for (int w = 0; w < kThreads / 64; w++) e += dsum[w];
          *a.energy_out = (float)e;
        }
      }
    }
  }
}

// six waves per SIMD = three workgroups per CU, which is what the 50 KB tile allows: the FINAL instances otherwise take
// 86-91 registers and run two (log_decimate 3 and 4: -5 % and -10 %)
template <int G, bool FINAL, bool ROT>
__global__ __launch_bounds__(kThreads, 6) void k_hb_group(GroupArgs a) {
  extern __shared__ float4 lds4[];
  float2 *lds = reinterpret_cast<float2 *>(lds4);
  if (blockIdx.x == 0)
    hb_group_body<G, FINAL, ROT, true>(a, lds);
  else
    hb_group_body<G, FINAL, ROT, false>(a, lds);
}

using GroupKernel = void (*)(GroupArgs);

template <int G>
GroupKernel group_kernel_g(bool final, bool rotate) {
  if (final) return rotate ? k_hb_group<G, true, true> : k_hb_group<G, true, false>;
  return rotate ? k_hb_group<G, false, true> : k_hb_group<G, false, false>;
}

GroupKernel group_kernel(int nstages, bool final, bool rotate) {
  switch (nstages) {
    case 1: return group_kernel_g<1>(final, rotate);
    case 2: return group_kernel_g<2>(final, rotate);
    case 3: return group_kernel_g<3>(final, rotate);
    default: return group_kernel_g<4>(final, rotate);
  }
}

struct Group {
  int nstages = 0;
  unsigned mask = 0;
  int halo = 0;
  int shift_in = 0;  // log2(input rate / final output rate)
  float2 *hist = nullptr;  // the `halo` input samples preceding the next call; hist[-2], hist[-1] are allocated (pad)
  float2 *out = nullptr;  // intermediate buffer (null for the last group)
  unsigned resident = 0;  // workgroups of this group's kernel that fit on a CU (for resident_for bytes of LDS)
  size_t resident_for = 0;
};

}  // namespace

struct kq_decimator {
  kq_decim_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<Group> groups;
  int rot_phase = 0;
  float atten = 1;
  float coeffs[4];
  float2 *in_dev = nullptr;   // staging for host-resident input
  float2 *out_dev = nullptr;  // staging for host-resident output
  int16_t *out16_dev = nullptr;
  unsigned long long *partial = nullptr;  // tagged per-workgroup energies of the last group's launch
  float *energy_dev = nullptr;
  int *err = nullptr;  // pinned host word the kernel sets when a tile's energy never arrived
  unsigned epoch = 0;
  size_t n_partial = 0;
  unsigned num_cus = 256;
  int max_fuse = kMaxFuse;
};

void kq_internal_set_error(const char *fmt, ...);

#define DEC_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      kq_internal_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                              \
    }                                                                                         \
  } while (0)

static int decim_alloc(kq_decimator *d) {
  kq_decim_config const &c = d->cfg;
  kq::DeviceScope dev_scope_(c.device);  // the caller's current device is restored on return
  {
    hipDeviceProp_t prop;
    DEC_TRY(hipGetDeviceProperties(&prop, c.device));
    d->num_cus = (unsigned)prop.multiProcessorCount;
  }
  if (c.stream)
    d->stream = (hipStream_t)c.stream;
  else {
    DEC_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    d->own_stream = true;
  }
  // stages in processing order: j = log_decimate-1 .. 0, 1-2-1 while j >= stage_threshold (hackrf.c:297-300)
  int const S = c.log_decimate;
  int done = 0;
  while (done < S) {
    d->groups.emplace_back();  // in the list before it owns anything, so that a failure below still frees it
    Group &g = d->groups.back();
    g.nstages = std::min(d->max_fuse, S - done);
    g.shift_in = S - done;
    for (int s = 0; s < g.nstages; s++) {
      int const j = S - 1 - (done + s);
      if (j < c.stage_threshold) g.mask |= 1u << s;
    }
    int need = 0;
    for (int s = g.nstages - 1; s >= 0; s--) need = 2 * need + ((g.mask >> s) & 1 ? 14 : 1);
    g.halo = need;
    done += g.nstages;
    float2 *h = nullptr;
    DEC_TRY(hipMalloc(&h, sizeof(float2) * (need + 2)));
    g.hist = h + 2;
    DEC_TRY(hipMemsetAsync(h, 0, sizeof(float2) * (need + 2), d->stream));
    if (done < S) DEC_TRY(hipMalloc(&g.out, sizeof(float2) * (c.max_out << (S - done))));
  }
  d->n_partial = (size_t)d->num_cus * 16 + 1;  // one per workgroup of the last launch
  DEC_TRY(hipMalloc(&d->partial, sizeof(unsigned long long) * d->n_partial));
  DEC_TRY(hipMemsetAsync(d->partial, 0, sizeof(unsigned long long) * d->n_partial, d->stream));  // epoch 0 = never written
  DEC_TRY(hipHostMalloc((void **)&d->err, sizeof(int), hipHostMallocDefault));
  *d->err = 0;
  DEC_TRY(hipMalloc(&d->energy_dev, sizeof(float)));
  DEC_TRY(hipStreamSynchronize(d->stream));
  return 0;
}

// after a synchronisation: did the edge workgroup of some call give up waiting for a tile's energy?
static bool decim_lost(kq_decimator *d) {
  if (!d->err || !*d->err) return false;
  *d->err = 0;
  kq_internal_set_error("kq_decim: a workgroup's output energy never arrived (device fault?); the energy of that call is NaN");
  return true;
}

extern "C" {

kq_decimator *kq_decim_create(const kq_decim_config *cfg) {
  if (!cfg || cfg->log_decimate < 1 || cfg->log_decimate > 16 || cfg->max_out == 0) {
    kq_internal_set_error("kq_decim_create: log_decimate must be 1..16 and max_out > 0");
    return nullptr;
  }
  kq_decimator *d = new kq_decimator;
  d->cfg = *cfg;
  d->atten = cfg->filter_atten != 0 ? cfg->filter_atten : powf(.5f, (float)cfg->log_decimate);  // hackrf.c:469
  // hackrf.c:229-238 -- [3] is next to the centre tap, [0] on the tails
  d->coeffs[3] = 490. / 802;
  d->coeffs[2] = -116. / 802;
  d->coeffs[1] = 33. / 802;
  d->coeffs[0] = -6. / 802;
  if (const char *e = getenv("KQ_DECIM_FUSE")) d->max_fuse = std::max(1, std::min(kMaxFuse, atoi(e)));  // diagnostic
  if (decim_alloc(d) != 0) {
    kq_decim_destroy(d);
    return nullptr;
  }
  return d;
}

int kq_decim_destroy(kq_decimator *d) {
  kq::DeviceScope dev_scope_(d ? d->cfg.device : -1);
  if (!d) return -1;
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  for (Group &g : d->groups) {
    if (g.hist) (void)hipFree(g.hist - 2);
    (void)hipFree(g.out);
  }
  (void)hipFree(d->in_dev);
  (void)hipFree(d->out_dev);
  (void)hipFree(d->out16_dev);
  (void)hipFree(d->partial);
  (void)hipFree(d->energy_dev);
  if (d->err) (void)hipHostFree(d->err);
  if (d->own_stream) (void)hipStreamDestroy(d->stream);
  delete d;
  return 0;
}

int kq_decim_set_coeffs(kq_decimator *d, const float coeffs[4]) {
  kq::DeviceScope dev_scope_(d ? d->cfg.device : -1);
  if (!d || !coeffs) return -1;
  for (int i = 0; i < 4; i++) d->coeffs[i] = coeffs[i];
  return 0;
}

int kq_decim_process(kq_decimator *d, const float *iq_in, int on_device, size_t n_out, float *out_cf32,
                     int16_t *out_s16, float *out_energy) {
  kq::DeviceScope dev_scope_(d ? d->cfg.device : -1);
  if (!d || !iq_in || !out_cf32) {
    kq_internal_set_error("kq_decim_process: null argument");
    return -1;
  }
  if (n_out == 0) return 0;
  if (n_out > d->cfg.max_out) {
    kq_internal_set_error("kq_decim_process: n_out %zu exceeds max_out %zu", n_out, d->cfg.max_out);
    return -1;
  }
  int const S = d->cfg.log_decimate;
  size_t const n_in = n_out << S;
  const float2 *src = (const float2 *)iq_in;
  float2 *final_out = (float2 *)out_cf32;
  int16_t *final16 = out_s16;
  if (!on_device) {
    if (!d->in_dev) {
      DEC_TRY(hipMalloc(&d->in_dev, sizeof(float2) * (d->cfg.max_out << S)));
      DEC_TRY(hipMalloc(&d->out_dev, sizeof(float2) * d->cfg.max_out));
      DEC_TRY(hipMalloc(&d->out16_dev, sizeof(int16_t) * 2 * d->cfg.max_out));
    }
    DEC_TRY(hipMemcpyAsync(d->in_dev, iq_in, sizeof(float2) * n_in, hipMemcpyHostToDevice, d->stream));
    src = d->in_dev;
    final_out = d->out_dev;
    final16 = out_s16 ? d->out16_dev : nullptr;
  }
  d->epoch++;
  if (d->epoch == 0) d->epoch = 1;  // 0 is what a never-written word holds (the bank's tags skip it the same way)
  size_t n_g_in = n_in;
  for (size_t gi = 0; gi < d->groups.size(); gi++) {
    Group &g = d->groups[gi];
    bool const last = gi + 1 == d->groups.size();
    GroupArgs a{};
    a.in = src;
    a.hist = g.hist;
    a.out = last ? final_out : g.out;
    a.out16 = last ? final16 : nullptr;
    a.partial = last ? d->partial : nullptr;
    a.energy_out = last && out_energy ? (on_device ? out_energy : d->energy_dev) : nullptr;
    a.err = d->err;
    a.epoch = d->epoch;
    a.n_out = (long long)(n_g_in >> g.nstages);
    a.nstages = g.nstages;
    a.hb15_mask = g.mask;
    a.halo = g.halo;
    a.rotate = gi == 0 && (d->cfg.offset & 3) != 0;
    a.rot_step = d->cfg.offset & 3;
    a.rot_phase0 = d->rot_phase;
    a.in_once = gi == 0;
    a.final = last;
    a.scale = d->atten;
    a.c0 = d->coeffs[0];
    a.c1 = d->coeffs[1];
    a.c2 = d->coeffs[2];
    a.c3 = d->coeffs[3];
    int const kTileOut = tile_out(g.nstages);
    unsigned const ntiles = (unsigned)((a.n_out + kTileOut - 1) / kTileOut);
    // two planes each for level 0 and level 1
    int const h1 = (g.halo - ((g.mask & 1) ? 14 : 1)) / 2;
    size_t const lds_elems = 2 * (size_t)plane_cap((kTileOut << g.nstages) + g.halo + (g.halo & 1)) +
                             (g.nstages > 1 ? 2 * (size_t)plane_cap((kTileOut << (g.nstages - 1)) + h1) : 0);
    // persistent workgroups: exactly as many as fit on the device at once (registers, wave slots and LDS all count -- an
    // estimate from the LDS alone once launched half as many again as could run, and the stragglers ran alone)
    GroupKernel const kern = group_kernel(g.nstages, last, a.rotate != 0);
    size_t const lds_bytes = sizeof(float2) * lds_elems;
    if (g.resident_for != lds_bytes) {
      int nb = 0;
      DEC_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, kThreads, lds_bytes));
      g.resident = (unsigned)std::max(1, nb);
      g.resident_for = lds_bytes;
    }
    unsigned const grid = std::min(ntiles, d->num_cus * g.resident) + 1;  // + the edge workgroup
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds_bytes, d->stream, a);
    src = g.out;
    n_g_in >>= g.nstages;
  }
  DEC_TRY(hipGetLastError());
  d->rot_phase = (int)((d->rot_phase + (long long)(n_in & 3) * (d->cfg.offset & 3)) & 3);
  if (out_energy && !on_device)
    DEC_TRY(hipMemcpyAsync(out_energy, d->energy_dev, sizeof(float), hipMemcpyDeviceToHost, d->stream));
  if (!on_device) {
    DEC_TRY(hipMemcpyAsync(out_cf32, d->out_dev, sizeof(float2) * n_out, hipMemcpyDeviceToHost, d->stream));
    if (out_s16)
      DEC_TRY(hipMemcpyAsync(out_s16, d->out16_dev, sizeof(int16_t) * 2 * n_out, hipMemcpyDeviceToHost, d->stream));
    DEC_TRY(hipStreamSynchronize(d->stream));
    if (decim_lost(d)) return -1;
  }
  return 0;
}

int kq_decim_sync(kq_decimator *d) {
  kq::DeviceScope dev_scope_(d ? d->cfg.device : -1);
  if (!d) return -1;
  DEC_TRY(hipStreamSynchronize(d->stream));
  return decim_lost(d) ? -1 : 0;
}

int kq_decim_reset(kq_decimator *d) {
  kq::DeviceScope dev_scope_(d ? d->cfg.device : -1);
  if (!d) return -1;
  for (Group &g : d->groups)
    DEC_TRY(hipMemsetAsync(g.hist, 0, sizeof(float2) * g.halo, d->stream));
  d->rot_phase = 0;
  return 0;
}

}  // extern "C"
