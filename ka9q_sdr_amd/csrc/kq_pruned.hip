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

// 32-point forward FFT in registers, decimation in frequency, radix 2, fully unrolled.
// In: v[0..31] natural order.  Out: bin q' is left in v[bitrev5(q')].
__device__ __forceinline__ void fft32_dif(float2 (&v)[32]) {
#pragma unroll
  for (int len = 32; len >= 2; len >>= 1) {
    int const half = len / 2;
    int const tstep = 64 / len;  // twiddle exp(-2 pi i j / len) = w64[j * 64/len]
#pragma unroll
    for (int base = 0; base < 32; base += len) {
#pragma unroll
      for (int j = 0; j < half; j++) {
        float2 const a = v[base + j], b = v[base + j + half];
        v[base + j] = cadd(a, b);
        float2 const d = csub(a, b);
        int const t = j * tstep;  // 0..31
        if (t == 0) {
          v[base + j + half] = d;
        } else if (t == 16) {
          v[base + j + half] = make_float2(d.y, -d.x);  // times -i
        } else {
          float const wr = w64_re(t), wi = w64_im(t);
          v[base + j + half] = make_float2(d.x * wr - d.y * wi, d.x * wi + d.y * wr);
        }
      }
    }
  }
}

// Table layout per channel (floats): see launch_pruned_tables
//   A  : [2 passes][32] float4  = (A0.re, A0.im, A1.re, A1.im)
//   J  : [3][64] float2          column-group twiddles, j = 1..3, natural q
//   Lv : [6][64] float2          cross-lane levels, lane bit i, natural q
constexpr int kTabA = 2 * 32 * 4;
constexpr int kTabJ = 3 * 64 * 2;
constexpr int kTabL = 6 * 64 * 2;
constexpr int kTabFloats = kTabA + kTabJ + kTabL;

__device__ __forceinline__ int signed_bin(int q) { return q <= 32 ? q : q - 64; }

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
    int const row = i >> 6, q = i & 63;
    // rows 0..2: column groups j=1..3 (offset 64 j); rows 3..8: lane bit i (offset 2^i)
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

// grid (channel groups, blocks); block = 512 threads = 8 waves; dynamic LDS = N float2
template <int CPW>
__global__ void __launch_bounds__(512) k_filter_pruned64(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ window,
                                                         const float *__restrict__ tab, int nchan) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const blk = blockIdx.y;
  int const N = g.N;
  int const R = g.D;
  {
    const float4 *src = reinterpret_cast<const float4 *>(window + (size_t)blk * g.L);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int i = threadIdx.x; i < N / 2; i += 512) dst[i] = src[i];
  }
  __syncthreads();
  int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int const groups = R / 64;  // column groups of 64 lanes

  for (int ci = 0; ci < CPW; ci++) {
    int const c = __builtin_amdgcn_readfirstlane((blockIdx.x * 8 + wave) * CPW + ci);
    if (c >= nchan) break;
    const float *tc = tab + (size_t)c * kTabFloats;
    const float4 *tA = reinterpret_cast<const float4 *>(tc);
    const float2 *tJ = reinterpret_cast<const float2 *>(tc + kTabA);
    const float2 *tL = reinterpret_cast<const float2 *>(tc + kTabA + kTabJ);

    float2 acc[2][32];  // [pass][bitrev5(q')]  with q = 2 q' + pass
    // pass 0 = even bins, pass 1 = odd bins: pass outermost so that only one 32-point work set and the
    // two accumulator sets are ever live (3 x 64 VGPRs); the window is re-read from LDS per pass
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll 1
      for (int j = 0; j < groups; j++) {
        const float2 *col = lds + 64 * j + lane;
        // the A table does not depend on j: hide that from LICM or all its scalars are hoisted and spilled
        int aoff = pass * 32;
        asm volatile("" : "+s"(aoff));
        const float4 *tAj = tA + aoff;
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; a++) {
          float4 const t = tAj[a];  // wave-uniform: scalar load
          float2 const x0 = col[(size_t)R * a], x1 = col[(size_t)R * (a + 32)];
          float2 r = make_float2(x0.x * t.x - x0.y * t.y, x0.x * t.y + x0.y * t.x);
          r.x = fmaf(x1.x, t.z, fmaf(-x1.y, t.w, r.x));
          r.y = fmaf(x1.x, t.w, fmaf(x1.y, t.z, r.y));
          v[a] = r;
        }
        fft32_dif(v);
        if (j == 0) {
#pragma unroll
          for (int i = 0; i < 32; i++) acc[pass][i] = v[i];
        } else {
          const float2 *tj = tJ + (size_t)(j - 1) * 64;
#pragma unroll
          for (int qp = 0; qp < 32; qp++) {
            float2 const w = tj[2 * qp + pass];  // wave-uniform
            acc[pass][bitrev5(qp)] = cfma(w, v[bitrev5(qp)], acc[pass][bitrev5(qp)]);
          }
        }
      }
    }

    // ---- cross-lane reduce-scatter over the 6 lane bits; lane ends with bin q = bitrev6(lane)
    // level t = 0..5 handles lane bit i = 5 - t and halves the per-lane bin set
    float2 z[32];
    int qlow;
    {
      int const bit = (lane >> 5) & 1;
      qlow = bit;
      const float2 *tl = tL + 5 * 64;
#pragma unroll
      for (int m = 0; m < 32; m++) {
        float2 const e = acc[0][bitrev5(m)], o = acc[1][bitrev5(m)];  // q = 2m, 2m+1
        float2 const keep = bit ? o : e, send = bit ? e : o;
        float2 recv;
        recv.x = __shfl_xor(send.x, 32, 64);
        recv.y = __shfl_xor(send.y, 32, 64);
        float2 const lo = bit ? recv : keep, hi = bit ? keep : recv;
        z[m] = cfma(tl[2 * m + bit], hi, lo);
      }
    }
#pragma unroll
    for (int t = 1; t < 6; t++) {
      int const i = 5 - t;
      int const bit = (lane >> i) & 1;
      int const cnt = 32 >> t;  // elements kept after this level
      const float2 *tl = tL + (size_t)i * 64;
#pragma unroll
      for (int m = 0; m < cnt; m++) {
        float2 const e = z[2 * m], o = z[2 * m + 1];
        float2 const keep = bit ? o : e, send = bit ? e : o;
        float2 recv;
        recv.x = __shfl_xor(send.x, 1 << i, 64);
        recv.y = __shfl_xor(send.y, 1 << i, 64);
        float2 const lo = bit ? recv : keep, hi = bit ? keep : recv;
        int const q = (((2 * m + bit) << t) | qlow);
        z[m] = cfma(tl[q], hi, lo);
      }
      qlow |= bit << t;
    }
    int const q = qlow;  // == bitrev6(lane)
    float2 y = z[0];

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

bool pruned_supported(const Geom &g) { return g.Ndec == 64 && g.D >= 64 && g.D % 64 == 0 && (size_t)g.N * 8 <= 128 * 1024; }
size_t pruned_table_elems(const Geom &) { return (size_t)kTabFloats / 2; }

void launch_pruned_tables(hipStream_t s, const Geom &g, const ChanDev &ch, float2 *chan_tw, int nchan) {
  hipLaunchKernelGGL(k_pruned_tables, dim3(nchan), dim3(256), 0, s, g, ch, reinterpret_cast<float *>(chan_tw), nchan);
}

void launch_filter_pruned(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                          const float2 *, const float2 *chan_tw, int nchan, int nblocks) {
  constexpr int CPW = 4;
  size_t const lds_bytes = (size_t)g.N * sizeof(float2);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void *)k_filter_pruned64<CPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    configured = true;
  }
  int const per_wg = 8 * CPW;
  hipLaunchKernelGGL(k_filter_pruned64<CPW>, dim3((nchan + per_wg - 1) / per_wg, nblocks), dim3(512), lds_bytes, s, g, ch, pl,
                     window, reinterpret_cast<const float *>(chan_tw), nchan);
}

}  // namespace kq
