// kq_ldsfft.hpp -- device helpers shared by the kernels: complex arithmetic, wave/block reductions and the
// in-LDS power-of-two FFT (unnormalised, sign -1 forward / +1 backward: the convention of fftwf_plan_dft_1d that
// filter.c:84,133 plans with).
#pragma once
#include <type_traits>

#include "kq_device.hpp"
#include "kq_lane.hpp"

namespace kq {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float cnrm(float2 a) { return a.x * a.x + a.y * a.y; }

__device__ __forceinline__ unsigned bitrev(unsigned i, int bits) { return bits ? (__brev(i) >> (32 - bits)) : 0u; }

// Unit phasor exp(j*2*pi*turns) from a double phase in turns
__device__ __forceinline__ float2 phasor_turns(double turns) {
  turns -= rint(turns);
  float s, c;
  sincospif(2.0f * (float)turns, &s, &c);
  return make_float2(c, s);
}

// Butterfly reductions over the 64 lanes (every lane gets the result) by DPP / v_permlane exchanges (kq_lane.hpp):
// same pairing order as the ds_bpermute versions they replace (32, 16, ..., 1), so the same rounding.
template <class T, class Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
  int const lane = threadIdx.x & 63;
  auto x = [&](auto m) {
    constexpr int M = decltype(m)::value;
    if constexpr (sizeof(T) == 4 && std::is_same<T, float>::value)
      return lane_xor<M>(v, lane);
    else
      return (T)lane_xor_i<M>((int)v, lane);
  };
  v = op(v, x(std::integral_constant<int, 32>{}));
  v = op(v, x(std::integral_constant<int, 16>{}));
  v = op(v, x(std::integral_constant<int, 8>{}));
  v = op(v, x(std::integral_constant<int, 4>{}));
  v = op(v, x(std::integral_constant<int, 2>{}));
  v = op(v, x(std::integral_constant<int, 1>{}));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_reduce(v, [](float a, float b) { return a + b; });
}
// Sum over the 64 lanes delivered to lane 63 only (the other lanes hold partial sums): seven v_add_f32 with DPP
// operands -- shifts by 1, 2, 3 inside each row of 16, by 4 and 8 into the upper banks, then the row totals broadcast
// into the rows above -- a third of the instructions of the butterfly that leaves the sum in every lane.  All lanes
// must be active.  (The s_nop are the two wait states a DPP read needs behind the write of its source register.)
__device__ __forceinline__ float wave_sum_to63(float v) {
  float r;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "=&v"(r)
      : "v"(v));
  return r;
}
__device__ __forceinline__ int wave_sum_i(int v) {
  return wave_reduce(v, [](int a, int b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_min(float v) {
  return wave_reduce(v, [](float a, float b) { return fminf(a, b); });
}

// Block-wide sum of (float, int) pairs; red_f / red_i hold one slot per wave (<= 16 waves)
__device__ __forceinline__ void block_sum_fi(float &f, int &i, float *red_f, int *red_i) {
  f = wave_sum(f);
  i = wave_sum_i(i);
  int const w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red_f[w] = f;
    red_i[w] = i;
  }
  __syncthreads();
  float tf = 0;
  int ti = 0;
  for (int k = 0; k < nw; k++) {
    tf += red_f[k];
    ti += red_i[k];
  }
  f = tf;
  i = ti;
}

// In-place FFT of 2^log2n points held in LDS in BIT-REVERSED order; result in natural order.
// Radix-2^2 decimation in time (plus one radix-2 stage when log2n is odd).  tw[k] = exp(-2*pi*i*k/T),
// T = 1 << tw_log2 >= n, k < T/2.  SIGN -1 forward / +1 backward, unnormalised like FFTW.
template <int SIGN>
__device__ inline void lds_fft(float2 *s, int log2n, const float2 *__restrict__ tw, int tw_log2) {
  int const n = 1 << log2n;
  int stage = 0;
  __syncthreads();
  if (log2n & 1) {
    for (int i = threadIdx.x; i < n / 2; i += blockDim.x) {
      float2 const a = s[2 * i], b = s[2 * i + 1];
      s[2 * i] = cadd(a, b);
      s[2 * i + 1] = csub(a, b);
    }
    stage = 1;
    __syncthreads();
  }
  for (; stage < log2n; stage += 2) {
    int const m = 1 << stage;
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x) {
      int const j = i & (m - 1);
      int const base = ((i >> stage) << (stage + 2)) + j;
      float2 w2 = tw[(size_t)j << (tw_log2 - stage - 1)];
      float2 w4 = tw[(size_t)j << (tw_log2 - stage - 2)];
      if (SIGN > 0) {
        w2.y = -w2.y;
        w4.y = -w4.y;
      }
      float2 const a0 = s[base], a1 = cmul(s[base + m], w2);
      float2 const a2 = s[base + 2 * m], a3 = cmul(s[base + 3 * m], w2);
      float2 const b0 = cadd(a0, a1), b1 = csub(a0, a1), b2 = cadd(a2, a3), b3 = csub(a2, a3);
      float2 const c2 = cmul(b2, w4);
      float2 c3 = cmul(b3, w4);
      c3 = (SIGN < 0) ? make_float2(c3.y, -c3.x) : make_float2(-c3.y, c3.x);  // times exp(-+ i*pi/2)
      s[base] = cadd(b0, c2);
      s[base + 2 * m] = csub(b0, c2);
      s[base + m] = cadd(b1, c3);
      s[base + 3 * m] = csub(b1, c3);
    }
    __syncthreads();
  }
}

// The oscillators of retune transitions before the last one that still have samples in a window (ChanDev::hist2_*): `shift`
// = b L takes the counts from the call's first window to block b's; `any` = the window has old samples at all.  The first
// level -- a second transition in a window, the usual case -- is held in registers; the few samples that lie before a third
// one fetch their oscillator where they need it (the whole list in registers, 28 of them, put the per-sample variants of
// k_filter_full16k 52 bytes into scratch; this form leaves them without, as they were).
struct OlderOsc {
  const int *len;      // the channel's kOldLevels counts
  const double *osc;   // ... and (phase, step, sweep) triples
  int shift;
  int n0, n1;          // samples of this window before the transition before the last / before the one before that
  double p0, f0, r0;   // level 0's oscillator (the usual case of a second transition: in registers)
};
__device__ __forceinline__ void load_older(const ChanDev &ch, int c, int shift, bool any, OlderOsc &o) {
  size_t const k = (size_t)c * kOldLevels;
  o.len = ch.hist2_len + k;
  o.osc = ch.hist2_osc + 3 * k;
  o.shift = shift;
  o.n0 = any ? o.len[0] - shift : 0;
  o.n1 = o.n0 > 0 ? o.len[1] - shift : 0;
  o.p0 = o.n0 > 0 ? o.osc[0] : 0.0;
  o.f0 = o.n0 > 0 ? o.osc[1] : 0.0;
  o.r0 = o.n0 > 0 ? o.osc[2] : 0.0;
}
// sample i of the window: the oldest transition it lies before decides (the counts fall with the level)
__device__ __forceinline__ void pick_older(const OlderOsc &o, int i, double &ph, double &ff, double &rr) {
  bool const l0 = i < o.n0;
  ph = l0 ? o.p0 : ph;
  ff = l0 ? o.f0 : ff;
  rr = l0 ? o.r0 : rr;
  if (i < o.n1) {
    for (int l = 1; l < kOldLevels; l++) {
      if (i >= o.len[l] - o.shift) break;
      ph = o.osc[3 * l];
      ff = o.osc[3 * l + 1];
      rr = o.osc[3 * l + 2];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Sizes with a factor 3, 5 or 7 (round 6: FFTW plans whatever N = L + M - 1 and N / decimate come out, filter.c:78,132, and
// decimate = samprate / 48000 is 5 at 240 kHz, radio_status.c:266).  n = f[0] f[1] ... f[nf-1], radices 2, 3, 4, 5, 7, applied
// in that order as in-place decimation-in-time stages: the stage of radix r over span m (the product of the radices before
// it) takes s[base + j + q m], q < r, times W_{m r}^{j q}, through an r-point transform back into the same places.  The
// input goes in DIGIT-REVERSED: natural index i sits at rev[i] (table built by the host with the plan: position
// p = sum_k d_k prod_{j<k} f[j] holds index i = sum_k d_k n / prod_{j<=k} f[j]); the result comes out in natural order,
// as from lds_fft.  Twiddles from the bank's full-circle table twc[k] = exp(-2 pi i k / tw_n), n | tw_n.
// The fast kernels stay with powers of two; this is the generic path's transform (k_filter_full, k_fm_audio, k_design,
// the compat kernels).  log2n >= 0 says the size IS a power of two: fft_pos / fft_any then are bitrev / lds_fft.
__device__ __forceinline__ unsigned fft_pos(unsigned i, const FftDim &d) { return d.log2n >= 0 ? bitrev(i, d.log2n) : d.rev[i]; }

template <int SIGN>
__device__ inline void lds_fft_mixed(float2 *s, const FftDim &d) {
  int const n = d.n;
  int m = 1;
  __syncthreads();
  for (int st = 0; st < d.nf; st++) {
    int const r = d.f[st];
    int const tstep = d.tw_n / (m * r);  // W_{m r}^{1} = twc[tstep]
    for (int i = threadIdx.x; i < n / r; i += blockDim.x) {
      int const j = i % m, base = (i / m) * m * r + j;
      float2 a[7];
#pragma unroll
      for (int q = 0; q < 7; q++) {
        if (q < r) {
          float2 v = s[base + q * m];
          if (q && j) {
            float2 w = d.twc[(size_t)j * q * tstep];
            if (SIGN > 0) w.y = -w.y;
            v = cmul(v, w);
          }
          a[q] = v;
        }
      }
      // r-point transforms, forward kernel exp(-2 pi i q t / r); SIGN > 0 conjugates the constants
      float const sg = SIGN < 0 ? 1.f : -1.f;
      if (r == 2) {
        s[base] = cadd(a[0], a[1]);
        s[base + m] = csub(a[0], a[1]);
      } else if (r == 4) {
        float2 const b0 = cadd(a[0], a[2]), b1 = csub(a[0], a[2]), b2 = cadd(a[1], a[3]), b3 = csub(a[1], a[3]);
        float2 const b3r = make_float2(sg * b3.y, -sg * b3.x);  // b3 times -i (forward) / +i (backward)
        s[base] = cadd(b0, b2);
        s[base + m] = cadd(b1, b3r);
        s[base + 2 * m] = csub(b0, b2);
        s[base + 3 * m] = csub(b1, b3r);
      } else if (r == 3) {
        float const c1 = -0.5f, s1 = sg * -0.86602540378443864676f;  // exp(-+2 pi i / 3)
        float2 const t1 = cadd(a[1], a[2]), t2 = csub(a[1], a[2]);
        float2 const u = make_float2(a[0].x + c1 * t1.x, a[0].y + c1 * t1.y);
        float2 const v = make_float2(-s1 * t2.y, s1 * t2.x);  // i s1 t2
        s[base] = cadd(a[0], t1);
        s[base + m] = cadd(u, v);
        s[base + 2 * m] = csub(u, v);
      } else if (r == 5) {
        float const c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        float const s1 = sg * -0.95105651629515357212f, s2 = sg * -0.58778525229247312917f;  // sin(-+2 pi / 5), sin(-+4 pi / 5)
        float2 const t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
        float2 const u1 = make_float2(a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y);
        float2 const u2 = make_float2(a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y);
        float2 const w1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
        float2 const w2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
        float2 const v1 = make_float2(-w1.y, w1.x), v2 = make_float2(-w2.y, w2.x);  // i w
        s[base] = cadd(a[0], cadd(t1, t2));
        s[base + m] = cadd(u1, v1);
        s[base + 4 * m] = csub(u1, v1);
        s[base + 2 * m] = cadd(u2, v2);
        s[base + 3 * m] = csub(u2, v2);
      } else {  // r == 7: X_k = a0 + sum_j [t_j cos(2 pi j k / 7) -+ i d_j sin(2 pi j k / 7)], t_j = a_j + a_{7-j}, d_j = a_j - a_{7-j}
        float const c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
        float const s1 = sg * -0.78183148246802980871f, s2 = sg * -0.97492791218182360702f, s3 = sg * -0.43388373911755812048f;
        float2 const t1 = cadd(a[1], a[6]), t2 = cadd(a[2], a[5]), t3 = cadd(a[3], a[4]);
        float2 const d1 = csub(a[1], a[6]), d2 = csub(a[2], a[5]), d3 = csub(a[3], a[4]);
        // j k mod 7 for k = 1: 1 2 3; k = 2: 2 4 6 = 2 -3 -1; k = 3: 3 6 2 = 3 -1 2 (cosine even, sine odd)
        float2 const u1 = make_float2(a[0].x + c1 * t1.x + c2 * t2.x + c3 * t3.x, a[0].y + c1 * t1.y + c2 * t2.y + c3 * t3.y);
        float2 const u2 = make_float2(a[0].x + c2 * t1.x + c3 * t2.x + c1 * t3.x, a[0].y + c2 * t1.y + c3 * t2.y + c1 * t3.y);
        float2 const u3 = make_float2(a[0].x + c3 * t1.x + c1 * t2.x + c2 * t3.x, a[0].y + c3 * t1.y + c1 * t2.y + c2 * t3.y);
        float2 const w1 = make_float2(s1 * d1.x + s2 * d2.x + s3 * d3.x, s1 * d1.y + s2 * d2.y + s3 * d3.y);
        float2 const w2 = make_float2(s2 * d1.x - s3 * d2.x - s1 * d3.x, s2 * d1.y - s3 * d2.y - s1 * d3.y);
        float2 const w3 = make_float2(s3 * d1.x - s1 * d2.x + s2 * d3.x, s3 * d1.y - s1 * d2.y + s2 * d3.y);
        float2 const v1 = make_float2(-w1.y, w1.x), v2 = make_float2(-w2.y, w2.x), v3 = make_float2(-w3.y, w3.x);  // i w
        s[base] = cadd(cadd(a[0], t1), cadd(t2, t3));
        s[base + m] = cadd(u1, v1);
        s[base + 6 * m] = csub(u1, v1);
        s[base + 2 * m] = cadd(u2, v2);
        s[base + 5 * m] = csub(u2, v2);
        s[base + 3 * m] = cadd(u3, v3);
        s[base + 4 * m] = csub(u3, v3);
      }
    }
    m *= r;
    __syncthreads();
  }
}
// either kind: `tw` / `tw_log2` are lds_fft's half-circle table (used for powers of two only)
template <int SIGN>
__device__ __forceinline__ void fft_any(float2 *s, const FftDim &d, const float2 *__restrict__ tw, int tw_log2) {
  if (d.log2n >= 0)
    lds_fft<SIGN>(s, d.log2n, tw, tw_log2);
  else
    lds_fft_mixed<SIGN>(s, d);
}

}  // namespace kq
