// kq_regfft.hpp -- small FFTs held entirely in a thread's registers (compile-time twiddles, fully unrolled), the
// building block of the register-resident transforms (kq_pruned.hip, kq_full16k.hip).  Forward sign (-1),
// unnormalised, like fftwf_plan_dft_1d(FFTW_FORWARD) in filter.c:84.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace kq {
namespace rfft {

// ---- compile-time twiddles exp(-2 pi i k / n), 0 <= k < n/2
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cx_cos(double x) {  // |x| <= pi/4
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
constexpr double turn_cos(double t) {  // cos(2 pi t), 0 <= t < 0.5, octant symmetries keep exact zeros exact
  return t <= 0.125 ? cx_cos(2 * kPi * t) : t <= 0.375 ? -cx_sin(2 * kPi * (t - 0.25)) : -cx_cos(2 * kPi * (0.5 - t));
}
constexpr double turn_sin(double t) {
  return t <= 0.125 ? cx_sin(2 * kPi * t) : t <= 0.375 ? cx_cos(2 * kPi * (t - 0.25)) : cx_sin(2 * kPi * (0.5 - t));
}
constexpr float tw_re(int k, int n) { return (float)turn_cos((double)k / n); }
constexpr float tw_im(int k, int n) { return (float)(-turn_sin((double)k / n)); }

constexpr int bitrev5(int i) {
  return ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4);
}
constexpr int bitrev4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

// One radix-2 butterfly of stage LEN: pair index I selects group base = (I / half) * LEN and offset j = I % half.
// FMA-fused: u = a + w b costs 4 fma, the other output is 2a - u (2 fma).
template <int NP, int LEN, int I>
__device__ __forceinline__ void bfly(float2 (&v)[NP]) {
  constexpr int half = LEN / 2, base = (I / half) * LEN, j = I % half;
  constexpr int t = j * (64 / LEN);  // twiddle exp(-2 pi i t / 64), 0..31
  float2 const a = v[base + j], b = v[base + j + half];
  if constexpr (t == 0) {
    v[base + j] = make_float2(a.x + b.x, a.y + b.y);
    v[base + j + half] = make_float2(a.x - b.x, a.y - b.y);
  } else if constexpr (t == 16) {  // w = -i
    v[base + j] = make_float2(a.x + b.y, a.y - b.x);
    v[base + j + half] = make_float2(a.x - b.y, a.y + b.x);
  } else {
    constexpr float wr = tw_re(t, 64), wi = tw_im(t, 64);
    float2 u;
    u.x = fmaf(wr, b.x, fmaf(-wi, b.y, a.x));
    u.y = fmaf(wr, b.y, fmaf(wi, b.x, a.y));
    v[base + j] = u;
    v[base + j + half] = make_float2(fmaf(2.f, a.x, -u.x), fmaf(2.f, a.y, -u.y));
  }
}

template <int NP, int LEN, int... I>
__device__ __forceinline__ void stage(float2 (&v)[NP], std::integer_sequence<int, I...>) {
  if constexpr (LEN <= NP) (bfly<NP, LEN, I>(v), ...);
}

// NP-point forward FFT in registers (NP <= 32), decimation in time, radix 2, unrolled by construction.
// In: sample a stored at v[bitrev(a)].  Out: bin q in v[q].
template <int NP>
__device__ __forceinline__ void fft_dit(float2 (&v)[NP]) {
  using pairs = std::make_integer_sequence<int, NP / 2>;
  stage<NP, 2>(v, pairs{});
  stage<NP, 4>(v, pairs{});
  stage<NP, 8>(v, pairs{});
  stage<NP, 16>(v, pairs{});
  stage<NP, 32>(v, pairs{});
}

// ---- packed variant: a complex value is one 2-vector, so that a butterfly is three v_pk_fma_f32 (two for the
// trivial twiddles) instead of six scalar FMAs.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// a * b, complex, for compile-time or scalar-register operands (the compiler picks the encodings)
__device__ __forceinline__ v2f pk_cmul_any(v2f a, v2f b) { return pk_fma((v2f){-a.y, a.y}, b.yx, a.xx * b); }
// a * b, complex, both in vector registers: exactly two instructions.  The half selects and the sign of a.y ride on
// the VOP3P operand modifiers (op_sel / op_sel_hi / neg_lo); written out because the compiler, given the vector form
// above, materialises {-a.y, a.y} with a v_xor_b32 and a v_mov_b32 whenever `a` has a second use.
//   t = (a.x b.x, a.x b.y);  d = (-a.y b.y + t.x, a.y b.x + t.y)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b) {
  v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
  return d;
}
// a * conj(b)
__device__ __forceinline__ v2f pk_cmul_conj(v2f a, v2f b) {
  v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (a.x b.x, -a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(t));  // (+a.y b.y, +a.y b.x)
  return d;
}

// c + a * b, complex, all in vector registers: two instructions (the product's first half accumulates onto c)
__device__ __forceinline__ v2f pk_cmadd(v2f a, v2f b, v2f c) {
  v2f t, d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(b), "v"(c));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
  return d;
}

template <int NP, int LEN, int I>
__device__ __forceinline__ void bfly_pk(v2f (&v)[NP]) {
  constexpr int half = LEN / 2, base = (I / half) * LEN, j = I % half;
  constexpr int t = j * (64 / LEN);
  v2f const a = v[base + j], b = v[base + j + half];
  if constexpr (t == 0) {
    v[base + j] = a + b;
    v[base + j + half] = a - b;
  } else if constexpr (t == 16) {  // w = -i: a +- (b.y, -b.x).  The swap and the sign ride on the operand modifiers; given the
    // vector expression the compiler builds (b.y, -b.x) in registers first (a v_xor_b32 and a v_mov_b32 per butterfly)
    v2f x, y;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(x) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(y) : "v"(a), "v"(b));
    v[base + j] = x;
    v[base + j + half] = y;
  } else {
    constexpr float wr = tw_re(t, 64), wi = tw_im(t, 64);
    v2f const u = pk_fma((v2f){-wi, wi}, b.yx, pk_fma((v2f){wr, wr}, b, a));
    v[base + j] = u;
    v[base + j + half] = pk_fma((v2f){2.f, 2.f}, a, -u);
  }
}

template <int NP, int LEN, int... I>
__device__ __forceinline__ void stage_pk(v2f (&v)[NP], std::integer_sequence<int, I...>) {
  if constexpr (LEN <= NP) (bfly_pk<NP, LEN, I>(v), ...);
}

template <int NP>
__device__ __forceinline__ void fft_dit_pk(v2f (&v)[NP]) {
  using pairs = std::make_integer_sequence<int, NP / 2>;
  stage_pk<NP, 2>(v, pairs{});
  stage_pk<NP, 4>(v, pairs{});
  stage_pk<NP, 8>(v, pairs{});
  stage_pk<NP, 16>(v, pairs{});
  stage_pk<NP, 32>(v, pairs{});
}

}  // namespace rfft
}  // namespace kq
