// kq_pruned.hip -- pruned forward path of the pre-detection filter for gfx950.
//
// execute_filter_output reads only N_dec = N/D of the N master bins (filter.c:206-227): bins
// 0..N_dec/2 and N-N_dec/2+1..N-1, i.e. signed bins k in (-N_dec/2, N_dec/2].  With N = N_dec*R and
// n = R*a + b the mixed spectrum at those bins is
//
//   Y[k] = P0 * sum_{b<R} T_b[k] * F_b[k mod N_dec],   F_b[q] = sum_{a<N_dec} x[R a + b] A[a] W_Ndec^{a q}
//
// with the NCO exp(j 2 pi (phi + f n)) split into P0 = exp(j 2 pi phi), A[a] = exp(j 2 pi f R a) and
// T_b[k] = exp(j 2 pi b (f - k/N)).  Same values as the full N-point FFT at those bins up to float
// rounding, for (5 N log2 N_dec + ~14 N) flops instead of 5 N log2 N.
//
// Mapping (N_dec = 64): one wave per channel-block, one lane per column b (mod 64), the 64-point
// column FFT in registers as two 32-point halves (even / odd output bins), per-channel twiddles from
// small tables read through the scalar cache, columns b, b+64, b+128, ... accumulated in registers,
// the remaining 64-lane sum done as a cross-lane reduce-scatter, then response multiply, CROSS_CONJ
// and the 64-point inverse FFT across lanes.  The N-sample window is staged once in LDS and shared by
// all waves of the workgroup and by several channels per wave.
#include <cstdlib>

#include "kq_device.hpp"

namespace kq {

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// a + b*c
__device__ __forceinline__ float2 cfma(float2 b, float2 c, float2 a) {
  return make_float2(fmaf(-b.y, c.y, fmaf(b.x, c.x, a.x)), fmaf(b.y, c.x, fmaf(b.x, c.y, a.y)));
}

// compile-time twiddles exp(-2 pi i k / 64)
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cx_cos(double x) {  // |x| <= pi/2 after reduction below; Taylor to 1e-17
  double const x2 = x * x;
  double term = 1, sum = 1;
  for (int n = 1; n < 14; n++) {
    term *= -x2 / ((2 * n - 1) * (2 * n));
    sum += term;
  }
  return sum;
}
constexpr double cx_sin(double x) {
  double const x2 = x * x;
  double term = x, sum = x;
  for (int n = 1; n < 14; n++) {
    term *= -x2 / ((2 * n) * (2 * n + 1));
    sum += term;
  }
  return sum;
}
// real / imaginary part of exp(-2 pi i k / 64) for 0 <= k < 32, exact symmetries used for accuracy
constexpr float w64_re(int k) {
  return k == 0 ? 1.f : k == 16 ? 0.f : (k < 16 ? (float)cx_cos(2 * kPi * k / 64) : (float)(-cx_sin(2 * kPi * (k - 16) / 64)));
}
constexpr float w64_im(int k) {
  return k == 0 ? 0.f : k == 16 ? -1.f : (k < 16 ? (float)(-cx_sin(2 * kPi * k / 64)) : (float)(-cx_cos(2 * kPi * (k - 16) / 64)));
}

constexpr int bitrev5(int i) {
  return ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4);
}

// 32-point forward FFT in registers, decimation in time, radix 2, fully unrolled, compile-time twiddles.
// In: sample a must have been stored at v[bitrev5(a)].  Out: bin q' in v[q'] (natural order).
// Butterflies are FMA-fused: u = a + w b costs 4 fma, the other output is 2a - u (2 fma): 6 ops instead of the
// 8 of multiply-then-add/sub.
__device__ __forceinline__ void fft32_dit(float2 (&v)[32]) {
#pragma unroll
  for (int len = 2; len <= 32; len <<= 1) {
    int const half = len / 2;
    int const tstep = 64 / len;  // exp(-2 pi i j / len) = w64[j * 64/len]
#pragma unroll
    for (int base = 0; base < 32; base += len) {
#pragma unroll
      for (int j = 0; j < half; j++) {
        float2 const a = v[base + j], b = v[base + j + half];
        int const t = j * tstep;  // 0..31
        if (t == 0) {
          v[base + j] = cadd(a, b);
          v[base + j + half] = csub(a, b);
        } else if (t == 16) {  // w = -i: w b = (b.y, -b.x)
          v[base + j] = make_float2(a.x + b.y, a.y - b.x);
          v[base + j + half] = make_float2(a.x - b.y, a.y + b.x);
        } else {
          float const wr = w64_re(t), wi = w64_im(t);
          float2 u;
          u.x = fmaf(wr, b.x, fmaf(-wi, b.y, a.x));
          u.y = fmaf(wr, b.y, fmaf(wi, b.x, a.y));
          v[base + j] = u;
          v[base + j + half] = make_float2(fmaf(2.f, a.x, -u.x), fmaf(2.f, a.y, -u.y));
        }
      }
    }
  }
}

// Table layout per channel (floats), built by k_pruned_tables:
//   A  : [2 passes][32] float4  = (A0.re, A0.im, A1.re, A1.im)
//   J  : [3][2 passes][32] float2  column-group twiddles, j = 1..3, bin q = 2 q' + pass
//   Lv : [6][64] float2            cross-lane levels, lane bit i, natural q
// A and J (2560 B) are copied into a wave-private LDS slot per channel-block and read back with
// wave-uniform ds_read_b128 (broadcast): LDS reads retire in order, so they pipeline with the column
// reads.  (Scalar loads were tried first: they return out of order, every use needs s_waitcnt lgkmcnt(0),
// and that also drains the LDS column reads -- 48 % of wave time was spent waiting.)
constexpr int kTabA = 2 * 32 * 4;
constexpr int kTabJ = 3 * 64 * 2;
constexpr int kTabL = 6 * 64 * 2;
constexpr int kTabFloats = kTabA + kTabJ + kTabL;
constexpr int kWaveTabF4 = (kTabA + kTabJ) / 4;  // 160 float4 per wave

__device__ __forceinline__ int signed_bin(int q) { return q <= 32 ? q : q - 64; }

__device__ __forceinline__ void swap32(float &a, float &b) {
  // v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b
  auto const r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

}  // namespace

// Per-channel twiddle tables for the call (the NCO step f0 is constant over a call for unswept channels)
__global__ void k_pruned_tables(Geom g, ChanDev ch, float *__restrict__ tab, int nchan) {
  int const c = blockIdx.x;
  if (c >= nchan) return;
  double const f = ch.lo_freq[c];
  int const R = g.D;  // N = N_dec * R
  float *t = tab + (size_t)c * kTabFloats;
  for (int i = threadIdx.x; i < 2 * 32; i += blockDim.x) {
    int const pass = i >> 5, a = i & 31;
    // even bins: u[a] = x[a] A[a] + x[a+32] A[a+32]
    // odd bins : v[a] = (x[a] A[a] - x[a+32] A[a+32]) * W64^a
    double t0 = f * (double)R * a, t1 = f * (double)R * (a + 32);
    if (pass) {
      t0 -= a / 64.0;
      t1 -= a / 64.0;
      t1 += 0.5;  // the minus sign
    }
    t0 -= rint(t0);
    t1 -= rint(t1);
    float s0, c0, s1, c1;
    sincospif(2.f * (float)t0, &s0, &c0);
    sincospif(2.f * (float)t1, &s1, &c1);
    float *o = t + (size_t)i * 4;
    o[0] = c0;
    o[1] = s0;
    o[2] = c1;
    o[3] = s1;
  }
  for (int i = threadIdx.x; i < 3 * 64 + 6 * 64; i += blockDim.x) {
    int const row = i >> 6, rem = i & 63;
    // rows 0..2: column groups j=1..3 (offset 64 j), stored [pass][q']; rows 3..8: lane bit (offset 2^bit), natural q
    int const q = row < 3 ? 2 * (rem & 31) + (rem >> 5) : rem;
    double const off = row < 3 ? 64.0 * (row + 1) : (double)(1 << (row - 3));
    double turns = off * f;
    turns -= rint(turns);
    turns -= off * (double)signed_bin(q) / (double)g.N;
    turns -= rint(turns);
    float s, co;
    sincospif(2.f * (float)turns, &s, &co);
    float *o = t + kTabA + (size_t)i * 2;
    o[0] = co;
    o[1] = s;
  }
}

// grid (channel groups, blocks); block = NWAVES waves; dynamic LDS = N float2 + NWAVES * 2560 B
template <int NWAVES, int CPW, int R>
__global__ void __launch_bounds__(NWAVES * 64) k_filter_pruned64(Geom g, ChanDev ch, Planes pl,
                                                                 const float2 *__restrict__ window,
                                                                 const float *__restrict__ tab, int nchan) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const blk = blockIdx.y;
  constexpr int N = 64 * R;  // R = decimation ratio = number of columns; compile-time so LDS offsets are immediates
  {
    const float4 *src = reinterpret_cast<const float4 *>(window + (size_t)blk * g.L);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int i = threadIdx.x; i < N / 2; i += NWAVES * 64) dst[i] = src[i];
  }
  __syncthreads();
  int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int groups = R / 64;  // column groups of 64 lanes
  float4 *wtab = reinterpret_cast<float4 *>(lds + N) + wave * kWaveTabF4;
  int const b0 = lane & 1;

  for (int ci = 0; ci < CPW; ci++) {
    int const c = __builtin_amdgcn_readfirstlane((blockIdx.x * NWAVES + wave) * CPW + ci);
    if (c >= nchan) break;
    const float *tc = tab + (size_t)c * kTabFloats;
    const float2 *tL = reinterpret_cast<const float2 *>(tc + kTabA + kTabJ);
    {
      const float4 *src = reinterpret_cast<const float4 *>(tc);
      for (int i = lane; i < kWaveTabF4; i += 64) wtab[i] = src[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const float2 *wJ = reinterpret_cast<const float2 *>(wtab + 64);

    float2 ypass[2];
    // pass 0 = even bins, pass 1 = odd bins (first radix-2 DIF stage of the 64-point column FFT fused into the
    // premultiply).  Each pass: accumulate the column groups, then reduce over the 64 lanes.
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      float2 acc[32];  // acc[q'] = bin q = 2 q' + pass
#pragma unroll 1
      for (int j = 0; j < groups; j++) {
        const float2 *col = lds + 64 * j + lane;
        const float4 *tAj = wtab + pass * 32;
        float2 v[32];
        // Software pipeline in chunks of CH samples: the LDS reads of chunk k+1 (2 column samples and one
        // broadcast twiddle pair per row) are issued before the arithmetic of chunk k; sched_barriers pin
        // that order, otherwise the scheduler hoists all 96 reads of the pass and spills.
        constexpr int CH = 4;
        float4 tw_[2][CH];
        float2 xa_[2][CH], xb_[2][CH];
#pragma unroll
        for (int i = 0; i < CH; i++) {
          tw_[0][i] = tAj[i];
          xa_[0][i] = col[R * i];
          xb_[0][i] = col[R * (i + 32)];
        }
#pragma unroll
        for (int k = 0; k < 32 / CH; k++) {
          int const cur = k & 1, nxt = cur ^ 1;
          if (k + 1 < 32 / CH) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
              int const a = (k + 1) * CH + i;
              tw_[nxt][i] = tAj[a];
              xa_[nxt][i] = col[R * a];
              xb_[nxt][i] = col[R * (a + 32)];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < CH; i++) {
            float4 const t = tw_[cur][i];
            float2 const x0 = xa_[cur][i], x1 = xb_[cur][i];
            float2 r = make_float2(x0.x * t.x - x0.y * t.y, x0.x * t.y + x0.y * t.x);
            r.x = fmaf(x1.x, t.z, fmaf(-x1.y, t.w, r.x));
            r.y = fmaf(x1.x, t.w, fmaf(x1.y, t.z, r.y));
            v[bitrev5(k * CH + i)] = r;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        fft32_dit(v);
        __builtin_amdgcn_sched_barrier(0);
        if (j == 0) {
#pragma unroll
          for (int i = 0; i < 32; i++) acc[i] = v[i];  // natural bin order
        } else {
          const float4 *tj = reinterpret_cast<const float4 *>(wJ + ((j - 1) * 2 + pass) * 32);
#pragma unroll
          for (int k = 0; k < 4; k++) {  // 8 bins per chunk, twiddles as 4 broadcast ds_read_b128
            float4 w4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) w4[i] = tj[k * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; i++) {
              int const q0 = k * 8 + 2 * i;
              acc[q0] = cfma(make_float2(w4[i].x, w4[i].y), v[q0], acc[q0]);
              acc[q0 + 1] = cfma(make_float2(w4[i].z, w4[i].w), v[q0 + 1], acc[q0 + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      // ---- reduce-scatter over lane bits 5..1, then a pair sum over bit 0
      float2 z[16];
      int qlow;
      {  // lane bit 5 through v_permlane32_swap: no LDS, no selects
        int const bit = (lane >> 5) & 1;
        qlow = bit;
        const float2 *tl = tL + 5 * 64;
#pragma unroll
        for (int m = 0; m < 16; m++) {
          float2 e = acc[2 * m], o = acc[2 * m + 1];
          swap32(e.x, o.x);
          swap32(e.y, o.y);
          // lower lanes: e = own even, o = partner's even; upper lanes: e = partner's odd, o = own odd
          z[m] = cfma(tl[2 * (2 * m + bit) + pass], o, e);
        }
      }
#pragma unroll
      for (int t = 1; t < 5; t++) {
        int const i = 5 - t;
        int const bit = (lane >> i) & 1;
        int const cnt = 16 >> t;
        const float2 *tl = tL + (size_t)i * 64;
#pragma unroll
        for (int m = 0; m < cnt; m++) {
          float2 const e = z[2 * m], o = z[2 * m + 1];
          float2 const keep = bit ? o : e, send = bit ? e : o;
          float2 recv;
          recv.x = __shfl_xor(send.x, 1 << i, 64);
          recv.y = __shfl_xor(send.y, 1 << i, 64);
          float2 const lo = bit ? recv : keep, hi = bit ? keep : recv;
          int const qp = (((2 * m + bit) << t) | qlow);
          z[m] = cfma(tl[2 * qp + pass], hi, lo);
        }
        qlow |= bit << t;
      }
      {  // lane bit 0: both lanes of a pair end with the full sum
        float2 recv;
        recv.x = __shfl_xor(z[0].x, 1, 64);
        recv.y = __shfl_xor(z[0].y, 1, 64);
        float2 const lo = b0 ? recv : z[0], hi = b0 ? z[0] : recv;
        ypass[pass] = cfma(tL[2 * qlow + pass], hi, lo);
      }
    }
    // lane now holds bin 2*rev5(lane>>1) + pass for both passes; keep pass = lane bit 0, then move bin
    // bitrev6(lane) into each lane for the decimation-in-time inverse transform
    float2 y = b0 ? ypass[1] : ypass[0];
    int const q = (int)(__brev((unsigned)lane) >> 26);  // bin this lane must hold
    {
      int const src = (int)((__brev((unsigned)(q >> 1)) >> 27) << 1) | (q & 1);
      float2 t2;
      t2.x = __shfl(y.x, src, 64);
      t2.y = __shfl(y.y, src, 64);
      y = t2;
    }

    // ---- P0, response multiply (filter.c:206-227), CROSS_CONJ (filter.c:239-249)
    {
      double const m0 = (double)blk * g.L;
      double turns = ch.lo_phase[c] + ch.lo_freq[c] * m0;
      turns -= rint(turns);
      float s, co;
      sincospif(2.f * (float)turns, &s, &co);
      y = cmul(y, make_float2(co, s));
      y = cmul(y, ch.resp[(size_t)c * 64 + q]);
      if (ch.flags[c] & FLAG_ISB) {
        int const qp = (64 - q) & 63;
        int const partner = (int)(__brev((unsigned)qp) >> 26);
        float2 o;
        o.x = __shfl(y.x, partner, 64);
        o.y = __shfl(y.y, partner, 64);
        if (q != 0 && q != 32) y = (q < 32) ? cadd(y, cconj(o)) : csub(y, cconj(o));
      }
    }
    // ---- 64-point inverse FFT across lanes: decimation in time on bit-reversed input
#pragma unroll
    for (int s = 0; s < 6; s++) {
      int const half = 1 << s;
      int const bit = (lane >> s) & 1;
      int const jj = lane & (half - 1);
      float sw, cw;
      sincospif((float)jj / (float)half, &sw, &cw);  // exp(+i pi jj / half) = exp(+2 pi i jj / (2 half))
      float2 const v = bit ? cmul(y, make_float2(cw, sw)) : y;
      float2 r;
      r.x = __shfl_xor(v.x, half, 64);
      r.y = __shfl_xor(v.y, half, 64);
      y = bit ? csub(r, v) : cadd(v, r);
    }
    // lane m holds sample m of the N_dec-point block; the last olen are the output (filter.c:131)
    int const first = 64 - g.olen;
    if (lane >= first) pl.filt[((size_t)c * g.max_blocks + blk) * g.olen + (lane - first)] = y;
  }
}

bool pruned_supported(const Geom &g) { return g.Ndec == 64 && (g.D == 64 || g.D == 128 || g.D == 256); }
size_t pruned_table_elems(const Geom &) { return (size_t)kTabFloats / 2; }

void launch_pruned_tables(hipStream_t s, const Geom &g, const ChanDev &ch, float2 *chan_tw, int nchan) {
  hipLaunchKernelGGL(k_pruned_tables, dim3(nchan), dim3(256), 0, s, g, ch, reinterpret_cast<float *>(chan_tw), nchan);
}

namespace {
template <int NWAVES, int CPW, int R>
void launch_r(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window, const float *tab,
              int nchan, int nblocks) {
  size_t const lds_bytes = (size_t)g.N * sizeof(float2) + (size_t)NWAVES * kWaveTabF4 * sizeof(float4);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void *)k_filter_pruned64<NWAVES, CPW, R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    configured = true;
  }
  int const per_wg = NWAVES * CPW;
  hipLaunchKernelGGL((k_filter_pruned64<NWAVES, CPW, R>), dim3((nchan + per_wg - 1) / per_wg, nblocks), dim3(NWAVES * 64),
                     lds_bytes, s, g, ch, pl, window, tab, nchan);
}
template <int NWAVES, int CPW>
void launch_variant(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window, const float *tab,
                    int nchan, int nblocks) {
  if (g.D == 256) return launch_r<NWAVES, CPW, 256>(s, g, ch, pl, window, tab, nchan, nblocks);
  if (g.D == 128) return launch_r<NWAVES, CPW, 128>(s, g, ch, pl, window, tab, nchan, nblocks);
  return launch_r<NWAVES, CPW, 64>(s, g, ch, pl, window, tab, nchan, nblocks);
}
}  // namespace

void launch_filter_pruned(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                          const float2 *, const float2 *chan_tw, int nchan, int nblocks) {
  // tuning knobs (waves per workgroup, channels per wave); defaults chosen on MI355X
  static int waves = 0, cpw = 0;
  if (!waves) {
    const char *e = getenv("KQ_PRUNED_WAVES");
    waves = e ? atoi(e) : 8;
    e = getenv("KQ_PRUNED_CPW");
    cpw = e ? atoi(e) : 4;
  }
  const float *tab = reinterpret_cast<const float *>(chan_tw);
  if (waves == 12 && cpw == 2) return launch_variant<12, 2>(s, g, ch, pl, window, tab, nchan, nblocks);
  if (waves == 12) return launch_variant<12, 4>(s, g, ch, pl, window, tab, nchan, nblocks);
  if (cpw == 2) return launch_variant<8, 2>(s, g, ch, pl, window, tab, nchan, nblocks);
  return launch_variant<8, 4>(s, g, ch, pl, window, tab, nchan, nblocks);
}

}  // namespace kq
