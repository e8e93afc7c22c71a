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
// Mapping: one wave per channel-block, one lane per column b (mod 64).  The N_dec-point column FFT runs
// entirely in the lane's registers as P = N_dec/32 passes of a 32-point FFT (pass p yields bins q = P q' + p;
// the first radix-P stage is fused into the premultiply by A).  Columns b, b+64, ... are accumulated in
// registers by Horner's rule with one wave-uniform twiddle T[k] = exp(j 2 pi 64 (f - k/N)); the remaining sum
// over the 64 lanes is a reduce-scatter over the lane bits (v_permlane32_swap for bit 5), followed in the same
// wave by P0, the response multiply, CROSS_CONJ and the N_dec-point inverse transform.
// Per-channel twiddles live in a wave-private LDS slot and are read back as wave-uniform (broadcast)
// ds_reads: LDS reads retire in order, so they pipeline with the column reads.  (Scalar loads were tried
// first: they return out of order, every use needs s_waitcnt lgkmcnt(0), which also drains the LDS column
// reads -- 48 % of wave time was spent waiting.)
//
// Two kernels:
//   k_pruned_resident  N_dec = 64, N <= 16384: the whole window is staged once in LDS and shared by all waves
//                      of the workgroup and several channels per wave.
//   k_pruned_stream    N_dec = 128, N = 65536: the window is streamed through LDS in slices of 64 columns
//                      with global_load_lds (no VGPR staging), double buffered.
// Swept NCO (set_doppler rate != 0, osc.c:43-47): phase(n) = phi + f n + r n(n-1)/2.  With n = R a + b the
// quadratic term splits into a part in a (folded into A), a part in b (negligible, checked on the host) and the
// cross term r R a b, applied per sample to first order (x *= 1 + j 2 pi r R a b).  The SWEPT kernels rebuild
// their tables per channel-block from the block's instantaneous step.
#include <cstdlib>
#include <type_traits>

#include "kq_device.hpp"
#include "kq_lane.hpp"
#include "kq_regfft.hpp"

namespace kq {

namespace {

// Complex products and butterflies go through 2-vectors so that they become v_pk_fma_f32 / v_pk_mul_f32
__device__ __forceinline__ rfft::v2f as_v2f(float2 a) { return (rfft::v2f){a.x, a.y}; }
__device__ __forceinline__ float2 as_f2(rfft::v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return as_f2(rfft::pk_cmul(as_v2f(a), as_v2f(b))); }
// the same left to the compiler: for operands that are compile-time rotations or wave-uniform (the asm form would
// force them into fresh vector registers)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return as_f2(rfft::pk_cmul_any(as_v2f(a), as_v2f(b))); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// a + b*c
// the same with every operand in a vector register (table entry times lane value): exactly two packed instructions
__device__ __forceinline__ float2 cfma_v(float2 b, float2 c, float2 a) { return as_f2(rfft::pk_cmadd(as_v2f(b), as_v2f(c), as_v2f(a))); }
__device__ __forceinline__ float2 cfma(float2 b, float2 c, float2 a) {
  rfft::v2f const bb = as_v2f(b), cc = as_v2f(c);
  return as_f2(rfft::pk_fma((rfft::v2f){-bb.y, bb.y}, cc.yx, rfft::pk_fma(bb.xx, cc, as_v2f(a))));
}

using rfft::bitrev5;
using rfft::kPi;
using rfft::tw_im;
using rfft::tw_re;
// 32-point forward FFT in registers (kq_regfft.hpp): sample a goes in at v[bitrev5(a)], bin q' comes out in v[q']
__device__ __forceinline__ void fft32_dit(float2 (&v)[32]) {
  rfft::v2f w[32];
#pragma unroll
  for (int i = 0; i < 32; i++) w[i] = as_v2f(v[i]);
  rfft::fft_dit_pk<32>(w);
#pragma unroll
  for (int i = 0; i < 32; i++) v[i] = as_f2(w[i]);
}

// ---- per-channel tables (floats).  ND = 64: A is merged per pass, [2][32] float4 = (A[a] w, A[a+32] w') with the
// pass twiddle folded in.  ND = 128: A only, [32][4] float2 = A[a' + 32 s]; the radix-4 factor (-i)^{s p} is a
// compile-time rotation and the residual twiddle W_128^{a' p} comes from a workgroup-shared LDS table.
// ND = 256: fully merged, [8 passes][32][8] float2 = A[a' + 32 s] W_8^{s p} W_256^{a' p}.  T: [P][32] float2 Horner twiddle of bin q = P q' + p.
// Lv: [6][ND] float2 cross-lane level twiddles, natural q.
template <int ND>
struct Tab {
  static constexpr int P = ND / 32;
  static constexpr int kA = (ND == 64) ? 2 * 32 * 4 : (ND == 128) ? 32 * P * 2 : P * 32 * P * 2;
  static constexpr int kT = ND * 2;
  static constexpr int kL = 6 * ND * 2;
  static constexpr int kFloats = kA + kT + kL;
  // float4 copied into the wave's LDS slot: A and the Horner twiddle T.  N/D = 256 keeps its (16 KiB, fully merged)
  // A table in HBM and reads it with wave-uniform vector loads; its wave slot is only the epilogue scratch.
  static constexpr int kWaveF4 = (ND == 256) ? ND / 2 : (kA + kT) / 4;
};

template <int ND>
__device__ __forceinline__ int signed_bin(int q) {
  return q <= ND / 2 ? q : q - ND;
}

__device__ __forceinline__ float2 unit(double turns) {
  turns -= rint(turns);
  float s, c;
  sincospif(2.f * (float)turns, &s, &c);
  return make_float2(c, s);
}

// Table entry i of the A + T part (i indexes float2 units), for instantaneous step f, sweep r, R columns
template <int ND>
__device__ __forceinline__ float2 table_entry_AT(int i, double f, double r, int R, int N) {
  constexpr int P = ND / 32;
  constexpr int nA = Tab<ND>::kA / 2;
  if (i < nA) {
    double a, extra = 0;
    if (ND == 64) {
      // entry = [pass][a'][half]: A[a' + 32 half], times W64^{a'} (and a minus sign on the second) in pass 1
      int const pass = i >> 6, ap = (i >> 1) & 31, hf = i & 1;
      a = ap + 32 * hf;
      if (pass) extra = -(double)ap / 64.0 + (hf ? 0.5 : 0.0);
    } else if (ND == 128) {
      int const ap = i / P, s = i % P;  // [a'][s]
      a = ap + 32 * s;
    } else {
      int const pass = i / (32 * P), ap = (i / P) % 32, s = i % P;  // [pass][a'][s]
      a = ap + 32 * s;
      extra = -(double)(s * pass) / 8.0 - (double)(ap * pass) / 256.0;
    }
    double const Ra = (double)R * a;
    return unit(f * Ra + r * (0.5 * Ra * (Ra - 1.0)) + extra);
  }
  int const t = i - nA;  // T: [p][q'] -> bin q = P q' + p
  int const p = t >> 5, qp = t & 31;
  int const q = P * qp + p;
  double turns = 64.0 * f;
  turns -= rint(turns);
  turns -= 64.0 * (double)signed_bin<ND>(q) / (double)N;
  return unit(turns);
}

__device__ __forceinline__ void swap32(float &a, float &b) {
  // v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b
  auto const r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

__device__ __forceinline__ void swap16(float &a, float &b) {
  // v_permlane16_swap: the odd rows (of 16 lanes) of a <-> the even rows of b
  auto const r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// One level of the reduce-scatter across lane bit I (3 or 2) without selects: on return `e` holds, in every lane, the
// value of the pair's LOWER lane (its own e, or the partner's o) and `o` that of the UPPER lane (the partner's e, or
// its own o) -- two DPP moves each, their bank masks picking the lanes with the bit clear or set.
template <int I>
__device__ __forceinline__ void pair_dpp(float &e, float &o) {
  static_assert(I == 3 || I == 2, "lane bits 3 and 2: whole banks of four lanes");
  int const ei = __float_as_int(e), oi = __float_as_int(o);
  int lo, up;
  if constexpr (I == 3) {
    lo = __builtin_amdgcn_update_dpp(ei, oi, 0x128, 0xF, 0xC, false);  // row_ror:8 into lanes 8..15 of each row
    up = __builtin_amdgcn_update_dpp(oi, ei, 0x128, 0xF, 0x3, false);  //            into lanes 0..7
  } else {
    lo = __builtin_amdgcn_update_dpp(ei, oi, 0x114, 0xF, 0xA, false);  // row_shr:4 into banks 1 and 3 (bit 2 set: lane - 4)
    up = __builtin_amdgcn_update_dpp(oi, ei, 0x104, 0xF, 0x5, false);  // row_shl:4 into banks 0 and 2 (bit 2 clear: lane + 4)
  }
  e = __int_as_float(lo);
  o = __int_as_float(up);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- one column group, one pass: premultiply + radix-P fold + 32-point FFT.
// col[STRIDE * a] is row a of this lane's column; tab is the wave's LDS slot (A part).
template <int ND, int PASS, int STRIDE, bool SWEPT>
__device__ __forceinline__ void column_pass(const float2 *col, const float4 *tab, const float2 *wsh, float kappa,
                                            float2 (&v)[32]) {
  constexpr int P = ND / 32;
  constexpr int CH = (ND == 64) ? 4 : (ND == 128) ? 2 : 1;  // rows a' per software-pipeline stage
  // Software pipeline in chunks of CH rows: the LDS reads of chunk k+1 are issued before the arithmetic of chunk
  // k; sched_barriers pin that order, otherwise the scheduler hoists every read of the pass and spills.
  constexpr int TW4 = (ND == 64) ? 1 : P / 2;  // float4 of twiddles per row a'
  float2 xs[2][CH][P];
  float4 tw[2][CH][TW4];
#pragma unroll
  for (int i = 0; i < CH; i++) {
#pragma unroll
    for (int s = 0; s < P; s++) xs[0][i][s] = col[STRIDE * (i + 32 * s)];
#pragma unroll
    for (int t = 0; t < TW4; t++) tw[0][i][t] = (ND == 64) ? tab[PASS * 32 + i] : tab[TW4 * i + t];
  }
#pragma unroll
  for (int k = 0; k < 32 / CH; k++) {
    int const cur = k & 1, nxt = cur ^ 1;
    if (k + 1 < 32 / CH) {
#pragma unroll
      for (int i = 0; i < CH; i++) {
        int const ap = (k + 1) * CH + i;
#pragma unroll
        for (int s = 0; s < P; s++) xs[nxt][i][s] = col[STRIDE * (ap + 32 * s)];
#pragma unroll
        for (int t = 0; t < TW4; t++) tw[nxt][i][t] = (ND == 64) ? tab[PASS * 32 + ap] : tab[TW4 * ap + t];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < CH; i++) {
      int const ap = k * CH + i;
      float2 x[P];
#pragma unroll
      for (int s = 0; s < P; s++) {
        x[s] = xs[cur][i][s];
        if (SWEPT) {  // cross term of the sweep, first order: x *= 1 + j kappa a
          float const ph = kappa * (float)(ap + 32 * s);
          x[s] = make_float2(fmaf(-ph, x[s].y, x[s].x), fmaf(ph, x[s].x, x[s].y));
        }
      }
      float2 r;
      if (ND == 64) {
        float4 const t = tw[cur][i][0];
        r = make_float2(x[0].x * t.x - x[0].y * t.y, x[0].x * t.y + x[0].y * t.x);
        r.x = fmaf(x[1].x, t.z, fmaf(-x[1].y, t.w, r.x));
        r.y = fmaf(x[1].x, t.w, fmaf(x[1].y, t.z, r.y));
      } else {
        // sum_s x_s A_s W_P^{s PASS}.  W_P^e = (-i)^(e8 >> 1) * W_8^(e8 & 1) with e8 the exponent on the 8th-root
        // circle: the power of -i is a compile-time component swap / sign of A_s; terms with an odd e8 go to a
        // second accumulator that is rotated by W_8 = (1 - i)/sqrt(2) once at the end.
        float2 re = make_float2(0.f, 0.f), ro = make_float2(0.f, 0.f);
        bool first_e = true, first_o = true;
#pragma unroll
        for (int s = 0; s < P; s++) {
          float4 const t4 = tw[cur][i][s >> 1];
          float2 const A = (s & 1) ? make_float2(t4.z, t4.w) : make_float2(t4.x, t4.y);
          int const e8 = ((8 / P) * s * PASS) & 7;
          int const rot = e8 >> 1;
          float2 const As = rot == 0   ? A
                            : rot == 1 ? make_float2(A.y, -A.x)
                            : rot == 2 ? make_float2(-A.x, -A.y)
                                       : make_float2(-A.y, A.x);
          if (e8 & 1) {
            ro = first_o ? cmulc(x[s], As) : cfma(x[s], As, ro);
            first_o = false;
          } else {
            re = first_e ? cmulc(x[s], As) : cfma(x[s], As, re);
            first_e = false;
          }
        }
        r = re;
        if (!first_o) {
          float const h = 0.70710678118654752440f;
          r.x += (ro.x + ro.y) * h;
          r.y += (ro.y - ro.x) * h;
        }
        if (PASS != 0 && ap != 0) {
          // times W_ND^{a' PASS}: read from a workgroup-shared LDS table (wave-uniform address).  As immediates
          // the distinct constants of the later passes do not fit the register file and spill.
          r = cmul(r, wsh[(ap * PASS) % ND]);
        }
      }
      v[bitrev5(ap)] = r;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  fft32_dit(v);
  __builtin_amdgcn_sched_barrier(0);
}

// acc = v + T * acc (Horner over the column groups), T from the wave's LDS slot, 8 bins per chunk
__device__ __forceinline__ void horner(float2 (&acc)[32], const float2 (&v)[32], const float4 *tT) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float4 w4[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w4[i] = tT[k * 4 + i];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int const q0 = k * 8 + 2 * i;
      acc[q0] = cfma(make_float2(w4[i].x, w4[i].y), acc[q0], v[q0]);
      acc[q0 + 1] = cfma(make_float2(w4[i].z, w4[i].w), acc[q0 + 1], v[q0 + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Reduce acc[q'] (bin q = P q' + PASS) over the 64 lanes: reduce-scatter over lane bits 5..1, pair sum over bit 0.
// Returns the bin with q' = rev5(lane >> 1), identical in both lanes of a pair.  th[i]: extra uniform factor of
// level i (swept NCO), multiplied into the level twiddle.
template <int ND, int PASS, bool SWEPT>
__device__ __forceinline__ float2 lane_reduce(float2 (&acc)[32], const float2 *tL, const float2 (&th)[6], int lane) {
  constexpr int P = ND / 32;
  float2 z[16];
  int qlow;
  {
    int const bit = (lane >> 5) & 1;
    qlow = bit;
    const float2 *tl = tL + 5 * ND;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      float2 e = acc[2 * m], o = acc[2 * m + 1];
      swap32(e.x, o.x);
      swap32(e.y, o.y);
      // lower lanes: e = own even, o = partner's even; upper lanes: e = partner's odd, o = own odd
      float2 w = tl[P * (2 * m + bit) + PASS];
      if (SWEPT) w = cmul(w, th[5]);
      z[m] = cfma_v(w, o, e);
    }
  }
#pragma unroll
  for (int t = 1; t < 5; t++) {
    int const i = 5 - t;
    int const bit = (lane >> i) & 1;
    int const cnt = 16 >> t;
    const float2 *tl = tL + (size_t)i * ND;
#pragma unroll
    for (int m = 0; m < cnt; m++) {
      float2 lo, hi;  // the pair's lower lane's even entry, its upper lane's odd entry
      if (i >= 2) {   // exchanges without selects: lane-half swap (bit 4), bank-masked DPP moves (bits 3 and 2)
        lo = z[2 * m];
        hi = z[2 * m + 1];
        if (i == 4) {
          swap16(lo.x, hi.x);
          swap16(lo.y, hi.y);
        } else if (i == 3) {
          pair_dpp<3>(lo.x, hi.x);
          pair_dpp<3>(lo.y, hi.y);
        } else {
          pair_dpp<2>(lo.x, hi.x);
          pair_dpp<2>(lo.y, hi.y);
        }
      } else {
        float2 const e = z[2 * m], o = z[2 * m + 1];
        float2 const keep = bit ? o : e, send = bit ? e : o;
        float2 recv;
        recv.x = __shfl_xor(send.x, 1 << i, 64);
        recv.y = __shfl_xor(send.y, 1 << i, 64);
        lo = bit ? recv : keep;
        hi = bit ? keep : recv;
      }
      int const qp = (((2 * m + bit) << t) | qlow);
      float2 w = tl[P * qp + PASS];
      if (SWEPT) w = cmul(w, th[i]);
      z[m] = cfma_v(w, hi, lo);
    }
    qlow |= bit << t;
  }
  float2 recv;
  recv.x = __shfl_xor(z[0].x, 1, 64);
  recv.y = __shfl_xor(z[0].y, 1, 64);
  int const b0 = lane & 1;
  float2 const lo = b0 ? recv : z[0], hi = b0 ? z[0] : recv;
  float2 w = tL[P * qlow + PASS];
  if (SWEPT) w = cmul(w, th[0]);
  return cfma_v(w, hi, lo);
}

// lane_reduce with the pass as a run-time value (rolled pass loop of the N/D = 256 kernel), unswept only
template <int ND>
__device__ __forceinline__ float2 lane_reduce_rt(float2 (&acc)[32], const float2 *tL, int pass, int lane) {
  constexpr int P = ND / 32;
  float2 z[16];
  int qlow;
  {
    int const bit = (lane >> 5) & 1;
    qlow = bit;
    const float2 *tl = tL + 5 * ND + pass;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      float2 e = acc[2 * m], o = acc[2 * m + 1];
      swap32(e.x, o.x);
      swap32(e.y, o.y);
      z[m] = cfma_v(tl[P * (2 * m + bit)], o, e);
    }
  }
#pragma unroll
  for (int t = 1; t < 5; t++) {
    int const i = 5 - t;
    int const bit = (lane >> i) & 1;
    int const cnt = 16 >> t;
    const float2 *tl = tL + (size_t)i * ND + pass;
#pragma unroll
    for (int m = 0; m < cnt; m++) {
      float2 lo, hi;
      if (i >= 2) {
        lo = z[2 * m];
        hi = z[2 * m + 1];
        if (i == 4) {
          swap16(lo.x, hi.x);
          swap16(lo.y, hi.y);
        } else if (i == 3) {
          pair_dpp<3>(lo.x, hi.x);
          pair_dpp<3>(lo.y, hi.y);
        } else {
          pair_dpp<2>(lo.x, hi.x);
          pair_dpp<2>(lo.y, hi.y);
        }
      } else {
        float2 const e = z[2 * m], o = z[2 * m + 1];
        float2 const keep = bit ? o : e, send = bit ? e : o;
        float2 recv;
        recv.x = __shfl_xor(send.x, 1 << i, 64);
        recv.y = __shfl_xor(send.y, 1 << i, 64);
        lo = bit ? recv : keep;
        hi = bit ? keep : recv;
      }
      int const qp = (((2 * m + bit) << t) | qlow);
      z[m] = cfma_v(tl[P * qp], hi, lo);
    }
    qlow |= bit << t;
  }
  float2 recv;
  recv.x = __shfl_xor(z[0].x, 1, 64);
  recv.y = __shfl_xor(z[0].y, 1, 64);
  int const b0 = lane & 1;
  float2 const lo = b0 ? recv : z[0], hi = b0 ? z[0] : recv;
  return cfma_v(tL[P * qlow + pass], hi, lo);
}

// Fill the wave's LDS slot (A and T) and the per-level sweep factors for channel c, block blk
template <int ND, bool SWEPT>
__device__ __forceinline__ void fill_tables(float4 *wtab, const float *tc, double f_blk, double df, double r, int R, int N,
                                            float2 (&th)[6], int lane) {
  if (SWEPT) {
    float2 *w2 = reinterpret_cast<float2 *>(wtab);
    for (int i = lane; i < Tab<ND>::kWaveF4 * 2; i += 64) w2[i] = table_entry_AT<ND>(i, f_blk, r, R, N);
    // the level tables in HBM were built for the call's first block: rotate them by the step difference
#pragma unroll
    for (int i = 0; i < 6; i++) th[i] = unit((double)(1 << i) * df);
  } else {
    const float4 *src = reinterpret_cast<const float4 *>(tc);
    for (int i = lane; i < Tab<ND>::kWaveF4; i += 64) wtab[i] = src[i];
#pragma unroll
    for (int i = 0; i < 6; i++) th[i] = make_float2(1.f, 0.f);
  }
  wave_lds_sync();
}

// Epilogue for N_dec > 64: each lane owns the bins q = P q' + p with q' = rev5(lane >> 1) and p = b0, b0+2, ...
// (P/2 bins per lane).  P0 and the response multiply (filter.c:206-227) are applied on the way into the wave's LDS
// scratch, which then serves CROSS_CONJ (filter.c:239-249) and the bit-reversed order of the N_dec-point inverse
// transform: V = N_dec/64 positions per lane, log2(V) in-lane stages, 6 cross-lane stages.
template <int ND>
__device__ __forceinline__ void epilogue_from_scratch(const float2 *scratch, bool isb, float2 *out, int olen, int lane);

template <int ND>
__device__ __forceinline__ void epilogue_lds(float2 *scratch, const float2 (&ypass)[ND / 32], float2 p0, const float2 *H,
                                             bool isb, float2 *out, int olen, int lane) {
  constexpr int P = ND / 32, V = ND / 64;
  int const b0 = lane & 1;
  int const qp = (int)(__brev((unsigned)(lane >> 1)) >> 27);
#pragma unroll
  for (int i = 0; i < V; i++) {
    float2 const y = b0 ? ypass[2 * i + 1] : ypass[2 * i];
    int const q = P * qp + 2 * i + b0;
    scratch[q] = cmul(cmul(y, p0), H[q]);
  }
  wave_lds_sync();
  epilogue_from_scratch<ND>(scratch, isb, out, olen, lane);
}

// CROSS_CONJ and the inverse transform, reading the response-weighted bins G[q] (natural order) from the scratch
template <int ND>
__device__ __forceinline__ void epilogue_from_scratch(const float2 *scratch, bool isb, float2 *out, int olen, int lane) {
  constexpr int V = ND / 64;
  constexpr int LOGN = (ND == 128) ? 7 : 8, LOGV = (ND == 128) ? 1 : 2;
  float2 z[V];
#pragma unroll
  for (int e = 0; e < V; e++) {
    int const q = (int)(__brev((unsigned)(V * lane + e)) >> (32 - LOGN));
    float2 gq = scratch[q];
    if (isb && q != 0 && q != ND / 2) {
      float2 const o = scratch[ND - q];
      gq = (q < ND / 2) ? cadd(gq, cconj(o)) : csub(gq, cconj(o));
    }
    z[e] = gq;
  }
  // decimation in time on bit-reversed positions V*lane + e
#pragma unroll
  for (int s = 0; s < LOGV; s++) {  // in-lane stages: twiddles exp(+i pi jj / half) with jj = e mod half
    int const half = 1 << s;
#pragma unroll
    for (int e = 0; e < V; e++) {
      if (e & half) continue;
      int const jj = e & (half - 1);
      float2 const a = z[e];
      float2 b = z[e + half];
      if (jj != 0) b = make_float2(-b.y, b.x);  // only half = 2, jj = 1: times +i
      z[e] = cadd(a, b);
      z[e + half] = csub(a, b);
    }
  }
#pragma unroll
  for (int s = LOGV; s < LOGN; s++) {
    int const half = 1 << s;
    int const bit = (lane >> (s - LOGV)) & 1;
#pragma unroll
    for (int e = 0; e < V; e++) {
      int const jj = (V * lane + e) & (half - 1);
      float sw, cw;
      sincospif((float)jj / (float)half, &sw, &cw);
      float2 const v = bit ? cmul(z[e], make_float2(cw, sw)) : z[e];
      float2 rr;
      rr.x = __shfl_xor(v.x, half / V, 64);
      rr.y = __shfl_xor(v.y, half / V, 64);
      z[e] = bit ? csub(rr, v) : cadd(v, rr);
    }
  }
  int const first = ND - olen;  // filter.c:131
#pragma unroll
  for (int e = 0; e < V; e++) {
    int const pos = V * lane + e;
    if (pos >= first) out[pos - first] = z[e];
  }
}

}  // namespace

// Per-channel twiddle tables for the call (step f0 at the first window; constant over a call for unswept channels)
template <int ND>
__global__ void k_pruned_tables(Geom g, ChanDev ch, float *__restrict__ tab, int nchan) {
  int const c = blockIdx.x;
  if (c >= nchan) return;
  double const f = ch.lo_freq[c];
  float2 *t = reinterpret_cast<float2 *>(tab + (size_t)c * Tab<ND>::kFloats);
  constexpr int nAT = (Tab<ND>::kA + Tab<ND>::kT) / 2;
  for (int i = threadIdx.x; i < nAT; i += blockDim.x) t[i] = table_entry_AT<ND>(i, f, 0.0, g.D, g.N);
  for (int i = threadIdx.x; i < 6 * ND; i += blockDim.x) {
    int const lvl = i / ND, q = i % ND;
    double const off = (double)(1 << lvl);
    double turns = off * f;
    turns -= rint(turns);
    turns -= off * (double)signed_bin<ND>(q) / (double)g.N;
    t[nAT + i] = unit(turns);
  }
}

// ------------------------------------------------------------------ N_dec = 64, window resident in LDS
// grid (channel groups, blocks); block = NWAVES waves; dynamic LDS = N float2 + NWAVES wave slots
template <int NWAVES, int CPW, int R, bool SWEPT>
__global__ void __launch_bounds__(NWAVES * 64) k_pruned_resident(Geom g, ChanDev ch, Planes pl,
                                                                 const float2 *__restrict__ window,
                                                                 const float *__restrict__ tab, int nchan,
                                                                 const int *__restrict__ chan_list, IirArgs iir) {
  constexpr int ND = 64;
  constexpr int N = ND * R;  // compile time, so that LDS offsets are immediates
  constexpr int groups = R / 64;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const blk = blockIdx.y;
  {
    const float4 *src = reinterpret_cast<const float4 *>(window + (size_t)blk * g.L);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int i = threadIdx.x; i < N / 2; i += NWAVES * 64) dst[i] = src[i];
  }
  // twiddles of the 64-point inverse transform, exp(+j pi k / 32): behind the wave slots
  float2 *itw = lds + N + NWAVES * Tab<ND>::kWaveF4 * 2;
  if (threadIdx.x < 32) {
    float sw, cw;
    sincospif((float)threadIdx.x / 32.f, &sw, &cw);
    itw[threadIdx.x] = make_float2(cw, sw);
  }
  __syncthreads();
  int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 *wtab = reinterpret_cast<float4 *>(lds + N) + wave * Tab<ND>::kWaveF4;
  int const b0 = lane & 1;
  double const m0 = (double)blk * g.L;

  for (int ci = 0; ci < CPW; ci++) {
    int const idx = __builtin_amdgcn_readfirstlane((blockIdx.x * NWAVES + wave) * CPW + ci);
    if (idx >= nchan) break;
    int const c = chan_list ? __builtin_amdgcn_readfirstlane(chan_list[idx]) : idx;  // slots emptied by remove_channel are skipped
    const float *tc = tab + (size_t)c * Tab<ND>::kFloats;
    const float2 *tL = reinterpret_cast<const float2 *>(tc + Tab<ND>::kA + Tab<ND>::kT);
    double const f0 = ch.lo_freq[c], r = SWEPT ? ch.lo_rate[c] : 0.0;
    double const f_blk = f0 + r * m0;
    float2 th[6];
    fill_tables<ND, SWEPT>(wtab, tc, f_blk, r * m0, r, R, N, th, lane);
    const float4 *wT = wtab + Tab<ND>::kA / 4;
    float const kap_lane = SWEPT ? (float)(2.0 * kPi * r * (double)R) : 0.f;  // times b = 64 j + lane below

    float2 y_even = make_float2(0.f, 0.f), y_odd = make_float2(0.f, 0.f);  // two named values: an array here ends up in scratch
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      float2 acc[32];
#pragma unroll 1
      for (int j = groups - 1; j >= 0; j--) {
        const float2 *col = lds + 64 * j + lane;
        float const kappa = kap_lane * (float)(64 * j + lane);
        float2 v[32];
        if (pass == 0)
          column_pass<ND, 0, R, SWEPT>(col, wtab, nullptr, kappa, v);
        else
          column_pass<ND, 1, R, SWEPT>(col, wtab, nullptr, kappa, v);
        if (j == groups - 1) {
#pragma unroll
          for (int i = 0; i < 32; i++) acc[i] = v[i];
        } else {
          horner(acc, v, wT + pass * 16);
        }
      }
      if (pass == 0)
        y_even = lane_reduce<ND, 0, SWEPT>(acc, tL, th, lane);
      else
        y_odd = lane_reduce<ND, 1, SWEPT>(acc, tL, th, lane);
    }
    // lane holds bin 2*rev5(lane>>1) + pass for both passes; keep pass = lane bit 0, then move bin bitrev6(lane)
    // into each lane for the decimation-in-time inverse transform
    float2 y = make_float2(b0 ? y_odd.x : y_even.x, b0 ? y_odd.y : y_even.y);
    int const q = (int)(__brev((unsigned)lane) >> 26);
    {
      int const src = (int)((__brev((unsigned)(q >> 1)) >> 27) << 1) | (q & 1);
      float2 t2;
      t2.x = __shfl(y.x, src, 64);
      t2.y = __shfl(y.y, src, 64);
      y = t2;
    }
    // ---- P0, response multiply (filter.c:206-227), CROSS_CONJ (filter.c:239-249)
    {
      double turns = ch.lo_phase[c] + f0 * m0;
      if (SWEPT) turns += r * (0.5 * m0 * (m0 - 1.0));
      y = cmul(y, unit(turns));
      y = cmul(y, ch.resp[(size_t)c * 64 + q]);
      if (ch.fflags[c] & FLAG_ISB) {
        int const qp = (64 - q) & 63;
        int const partner = (int)(__brev((unsigned)qp) >> 26);
        float2 o;
        o.x = __shfl(y.x, partner, 64);
        o.y = __shfl(y.y, partner, 64);
        if (q != 0 && q != 32) y = (q < 32) ? cadd(y, cconj(o)) : csub(y, cconj(o));
      }
    }
    // ---- 64-point inverse FFT across lanes: decimation in time on bit-reversed input
    // twiddle exp(+j pi (lane & (half - 1)) / half) from the staged table (the same sincospif values); partners by DPP
    // / permlane exchanges instead of the LDS crossbar
    auto stage = [&](auto hc) {
      constexpr int half = decltype(hc)::value;
      constexpr int sh = (half == 1) ? 5 : (half == 2) ? 4 : (half == 4) ? 3 : (half == 8) ? 2 : (half == 16) ? 1 : 0;
      bool const bit = (lane & half) != 0;
      float2 v = y;
      if (half > 1) {
        float2 const w = itw[(lane & (half - 1)) << sh];
        v = bit ? cmul(y, w) : y;
      }
      float2 rr;
      rr.x = lane_xor<half>(v.x, lane);
      rr.y = lane_xor<half>(v.y, lane);
      y = bit ? csub(rr, v) : cadd(v, rr);
    };
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 8>{});
    stage(std::integral_constant<int, 16>{});
    stage(std::integral_constant<int, 32>{});
    int const first = 64 - g.olen;  // the last olen samples are the output (filter.c:131)
    if (lane >= first) pl.filt[((size_t)c * g.max_blocks + blk) * g.olen + (lane - first)] = y;
  }
  // side job of one wave of the launch, once its own channels are done: the call's IF-power recurrence (kq_energy.hpp; as
  // in k_filter_full16k -- the launch of its own stood between the filter pass and the demodulators)
  if (iir.sums != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == NWAVES - 1)
    block_energy_iir_wave(iir.sums, iir.split, iir.update, iir.nblocks, iir.L, iir.state, iir.if_power, lane);
}

// ------------------------------------------------------------------ N_dec = 128, window streamed in column slices
// grid (channel groups, blocks); block = NWAVES waves = NWAVES channels.
// LDS: 2 slices of [128 rows][64 columns] float2 (64 KiB each) + per wave: A (1 KiB), T (1 KiB), scratch (1 KiB)
// + one shared table exp(-2 pi i k/128).
template <int NWAVES, int R, bool SWEPT>
__global__ void __launch_bounds__(NWAVES * 64) k_pruned_stream(Geom g, ChanDev ch, Planes pl,
                                                               const float2 *__restrict__ window,
                                                               const float *__restrict__ tab, int nchan,
                                                                 const int *__restrict__ chan_list) {
  constexpr int ND = 128, P = 4;
  constexpr int N = ND * R;
  constexpr int groups = R / 64;
  constexpr int kSlice = ND * 64;  // float2 per slice
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const blk = blockIdx.y;
  int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 *wtab = reinterpret_cast<float4 *>(lds + 2 * kSlice) + wave * (Tab<ND>::kWaveF4 + ND / 2);
  float2 *scratch = reinterpret_cast<float2 *>(wtab + Tab<ND>::kWaveF4);  // ND float2
  float2 *wsh = reinterpret_cast<float2 *>(reinterpret_cast<float4 *>(lds + 2 * kSlice) + NWAVES * (Tab<ND>::kWaveF4 + ND / 2));
  for (int i = threadIdx.x; i < 128; i += NWAVES * 64) wsh[i] = unit(-(double)i / 128.0);  // exp(-2 pi i k / 128)
  const float2 *win = window + (size_t)blk * g.L;
  double const m0 = (double)blk * g.L;

  int const idx = __builtin_amdgcn_readfirstlane(min((int)(blockIdx.x * NWAVES + wave), nchan - 1));
  int const c = chan_list ? __builtin_amdgcn_readfirstlane(chan_list[idx]) : idx;
  bool const live = (int)(blockIdx.x * NWAVES + wave) < nchan;  // idle waves still help loading the slices
  const float *tc = tab + (size_t)c * Tab<ND>::kFloats;
  const float2 *tL = reinterpret_cast<const float2 *>(tc + Tab<ND>::kA + Tab<ND>::kT);
  double const f0 = ch.lo_freq[c], r = SWEPT ? ch.lo_rate[c] : 0.0;
  double const f_blk = f0 + r * m0;
  float2 th[6];
  fill_tables<ND, SWEPT>(wtab, tc, f_blk, r * m0, r, R, N, th, lane);
  const float4 *wT = wtab + Tab<ND>::kA / 4;
  float const kap_lane = SWEPT ? (float)(2.0 * kPi * r * (double)R) : 0.f;

  // slice j -> LDS buffer: each wave copies ND/NWAVES rows, two 512-byte rows per global_load_lds instruction.
  // Address = wave-uniform base (SGPR pair) + one per-lane byte offset, so no per-load address registers pile up.
  unsigned const lane_off = (unsigned)(lane >> 5) * (unsigned)(R * sizeof(float2)) + (unsigned)(lane & 31) * 16u;
  auto stage = [&](int j, int buf) {
    constexpr int rows_per_wave = ND / NWAVES;
    const char *slice0 = reinterpret_cast<const char *>(win + (size_t)(wave * rows_per_wave) * R + 64 * j);
#pragma unroll
    for (int i = 0; i < rows_per_wave / 2; i++) {
      const char *ubase = slice0 + (size_t)(2 * i) * R * sizeof(float2);  // wave-uniform
      float2 *dst = lds + buf * kSlice + (wave * rows_per_wave + 2 * i) * 64;  // wave-uniform; hardware adds lane*16
      __builtin_amdgcn_global_load_lds(ubase + lane_off, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
  };

  float2 ypass[P];
  int step = 0;
  stage(groups - 1, 0);
  // The four passes are spelled out through a compile-time index: left as a loop the compiler unrolls it only
  // partially, `pass` becomes a run-time value and every per-pass array lands in scratch.
  auto do_pass = [&](auto pass_c) {
    constexpr int pass = decltype(pass_c)::value;
    float2 acc[32];
#pragma unroll 1
    for (int j = groups - 1; j >= 0; j--, step++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // slice (pass, j) has landed; everyone is done with the other buffer
      if (j > 0)
        stage(j - 1, (step + 1) & 1);
      else if (pass + 1 < P)
        stage(groups - 1, (step + 1) & 1);
      const float2 *col = lds + (step & 1) * kSlice + lane;
      float const kappa = kap_lane * (float)(64 * j + lane);
      float2 v[32];
      column_pass<ND, pass, 64, SWEPT>(col, wtab, wsh, kappa, v);
      if (j == groups - 1) {
#pragma unroll
        for (int i = 0; i < 32; i++) acc[i] = v[i];
      } else {
        horner(acc, v, wT + pass * 16);
      }
    }
    ypass[pass] = lane_reduce<ND, pass, SWEPT>(acc, tL, th, lane);
  };
  do_pass(std::integral_constant<int, 0>{});
  do_pass(std::integral_constant<int, 1>{});
  do_pass(std::integral_constant<int, 2>{});
  do_pass(std::integral_constant<int, 3>{});
  if (!live) return;
  {
    double turns = ch.lo_phase[c] + f0 * m0;
    if (SWEPT) turns += r * (0.5 * m0 * (m0 - 1.0));
    epilogue_lds<ND>(scratch, ypass, unit(turns), ch.resp + (size_t)c * ND, (ch.fflags[c] & FLAG_ISB) != 0,
                     pl.filt + ((size_t)c * g.max_blocks + blk) * g.olen, g.olen, lane);
  }
}

// ------------------------------------------------------------------ N_dec = 256, N = 16384 (R = 64: one column group)
// Window resident in LDS like k_pruned_resident.  8 passes of the 32-point column FFT; the radix-8 first stage, the
// premultiply and the pass twiddle are merged into one table A[p][a'][s] (16 KiB per channel, in HBM, read with
// wave-uniform vector loads that retire in order under vmcnt), which keeps the pass loop rolled.  No Horner step.
// After each pass the owning lane of a pair drops its bin, already times P0 and H, into the wave's LDS scratch.
template <int NWAVES, int CPW>
__global__ void __launch_bounds__(NWAVES * 64) k_pruned_resident256(Geom g, ChanDev ch, Planes pl,
                                                                    const float2 *__restrict__ window,
                                                                    const float *__restrict__ tab, int nchan,
                                                                 const int *__restrict__ chan_list) {
  constexpr int ND = 256, P = 8, R = 64;
  constexpr int N = ND * R;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const blk = blockIdx.y;
  {
    const float4 *src = reinterpret_cast<const float4 *>(window + (size_t)blk * g.L);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int i = threadIdx.x; i < N / 2; i += NWAVES * 64) dst[i] = src[i];
  }
  __syncthreads();
  int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2 *scratch = lds + N + wave * ND;
  double const m0 = (double)blk * g.L;
  int const vzero = __builtin_amdgcn_mbcnt_lo(0u, 0u);  // a zero the compiler cannot prove uniform: vector loads
  int const b0 = lane & 1;
  int const qp = (int)(__brev((unsigned)(lane >> 1)) >> 27);
  float2 th[6];
#pragma unroll
  for (int i = 0; i < 6; i++) th[i] = make_float2(1.f, 0.f);

  for (int ci = 0; ci < CPW; ci++) {
    int const idx = __builtin_amdgcn_readfirstlane((blockIdx.x * NWAVES + wave) * CPW + ci);
    if (idx >= nchan) break;
    int const c = chan_list ? __builtin_amdgcn_readfirstlane(chan_list[idx]) : idx;  // slots emptied by remove_channel are skipped
    const float *tc = tab + (size_t)c * Tab<ND>::kFloats;
    const float2 *tL = reinterpret_cast<const float2 *>(tc + Tab<ND>::kA + Tab<ND>::kT);
    const float2 *H = ch.resp + (size_t)c * ND;
    float2 const p0 = unit(ch.lo_phase[c] + ch.lo_freq[c] * m0);
    const float2 *col = lds + lane;
#pragma unroll 1
    for (int pass = 0; pass < P; pass++) {
      const float4 *tp = reinterpret_cast<const float4 *>(tc) + (size_t)pass * 32 * 4 + vzero;
      float2 v[32];
      float2 xs[2][P];
      float4 tw[2][4];
#pragma unroll
      for (int s = 0; s < P; s++) xs[0][s] = col[R * (32 * s)];
#pragma unroll
      for (int t = 0; t < 4; t++) tw[0][t] = tp[t];
#pragma unroll
      for (int ap = 0; ap < 32; ap++) {
        int const cur = ap & 1, nxt = cur ^ 1;
        if (ap + 1 < 32) {
#pragma unroll
          for (int s = 0; s < P; s++) xs[nxt][s] = col[R * (ap + 1 + 32 * s)];
#pragma unroll
          for (int t = 0; t < 4; t++) tw[nxt][t] = tp[4 * (ap + 1) + t];
        }
        __builtin_amdgcn_sched_barrier(0);
        float2 r = cmul(xs[cur][0], make_float2(tw[cur][0].x, tw[cur][0].y));
        r = cfma(xs[cur][1], make_float2(tw[cur][0].z, tw[cur][0].w), r);
#pragma unroll
        for (int t = 1; t < 4; t++) {
          r = cfma(xs[cur][2 * t], make_float2(tw[cur][t].x, tw[cur][t].y), r);
          r = cfma(xs[cur][2 * t + 1], make_float2(tw[cur][t].z, tw[cur][t].w), r);
        }
        v[bitrev5(ap)] = r;
        __builtin_amdgcn_sched_barrier(0);
      }
      fft32_dit(v);
      __builtin_amdgcn_sched_barrier(0);
      float2 const y = lane_reduce_rt<ND>(v, tL, pass, lane);
      if (b0 == (pass & 1)) {
        int const q = P * qp + pass;
        scratch[q] = cmul(cmul(y, p0), H[q]);
      }
    }
    wave_lds_sync();
    epilogue_from_scratch<ND>(scratch, (ch.fflags[c] & FLAG_ISB) != 0, pl.filt + ((size_t)c * g.max_blocks + blk) * g.olen,
                              g.olen, lane);
    wave_lds_sync();
  }
}

// ------------------------------------------------------------------ host side
bool pruned_supported(const Geom &g) {
  if (g.Ndec == 64) return g.D == 64 || g.D == 128 || g.D == 256;
  if (g.Ndec == 256) return g.D == 64;
  return g.Ndec == 128 && g.D == 512;
}
size_t pruned_table_elems(const Geom &g) {
  return (size_t)(g.Ndec == 64 ? Tab<64>::kFloats : g.Ndec == 128 ? Tab<128>::kFloats : Tab<256>::kFloats) / 2;
}

void launch_pruned_tables(hipStream_t s, const Geom &g, const ChanDev &ch, float2 *chan_tw, int nchan) {
  if (g.Ndec == 64)
    hipLaunchKernelGGL(k_pruned_tables<64>, dim3(nchan), dim3(256), 0, s, g, ch, reinterpret_cast<float *>(chan_tw), nchan);
  else if (g.Ndec == 128)
    hipLaunchKernelGGL(k_pruned_tables<128>, dim3(nchan), dim3(256), 0, s, g, ch, reinterpret_cast<float *>(chan_tw), nchan);
  else
    hipLaunchKernelGGL(k_pruned_tables<256>, dim3(nchan), dim3(256), 0, s, g, ch, reinterpret_cast<float *>(chan_tw), nchan);
}

namespace {
template <int R, bool SWEPT>
void launch_resident(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window, const float *tab,
                     int nchan, int nblocks, const int *chan_list, const IirArgs &iir) {
  constexpr int NWAVES = 8, CPW = 4;
  size_t const lds_bytes = (size_t)64 * R * sizeof(float2) + (size_t)NWAVES * Tab<64>::kWaveF4 * sizeof(float4) +
                           32 * sizeof(float2);  // window, wave slots, inverse-transform twiddles
  ensure_dynamic_lds((const void *)k_pruned_resident<NWAVES, CPW, R, SWEPT>, (size_t)(lds_bytes));
  int const per_wg = NWAVES * CPW;
  hipLaunchKernelGGL((k_pruned_resident<NWAVES, CPW, R, SWEPT>), dim3((nchan + per_wg - 1) / per_wg, nblocks),
                     dim3(NWAVES * 64), lds_bytes, s, g, ch, pl, window, tab, nchan, chan_list, iir);
}

void launch_resident256(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                        const float *tab, int nchan, int nblocks, const int *chan_list) {
  constexpr int NWAVES = 8, CPW = 4;
  size_t const lds_bytes = (size_t)16384 * sizeof(float2) + (size_t)NWAVES * 256 * sizeof(float2);
  ensure_dynamic_lds((const void *)k_pruned_resident256<NWAVES, CPW>, (size_t)(lds_bytes));
  int const per_wg = NWAVES * CPW;
  hipLaunchKernelGGL((k_pruned_resident256<NWAVES, CPW>), dim3((nchan + per_wg - 1) / per_wg, nblocks),
                     dim3(NWAVES * 64), lds_bytes, s, g, ch, pl, window, tab, nchan, chan_list);
}

template <bool SWEPT>
void launch_stream(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window, const float *tab,
                   int nchan, int nblocks, const int *chan_list) {
  constexpr int NWAVES = 8, R = 512;
  size_t const lds_bytes = (size_t)2 * 128 * 64 * sizeof(float2) + (size_t)NWAVES * (Tab<128>::kWaveF4 + 64) * sizeof(float4) +
                           128 * sizeof(float2);
  ensure_dynamic_lds((const void *)k_pruned_stream<NWAVES, R, SWEPT>, (size_t)(lds_bytes));
  hipLaunchKernelGGL((k_pruned_stream<NWAVES, R, SWEPT>), dim3((nchan + NWAVES - 1) / NWAVES, nblocks), dim3(NWAVES * 64),
                     lds_bytes, s, g, ch, pl, window, tab, nchan, chan_list);
}
}  // namespace

bool pruned_carries_iir(const Geom &g) { return g.Ndec == 64 && (g.D == 256 || g.D == 128 || g.D == 64); }

void launch_filter_pruned(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                          const float2 *chan_tw, int nchan, int nblocks, bool swept, const int *chan_list, const IirArgs &iir) {
  const float *tab = reinterpret_cast<const float *>(chan_tw);
  if (g.Ndec == 256) {  // unswept only (the host routes swept channels at this geometry to the full path)
    launch_resident256(s, g, ch, pl, window, tab, nchan, nblocks, chan_list);
    return;
  }
  if (g.Ndec == 128) {
    if (swept)
      launch_stream<true>(s, g, ch, pl, window, tab, nchan, nblocks, chan_list);
    else
      launch_stream<false>(s, g, ch, pl, window, tab, nchan, nblocks, chan_list);
    return;
  }
#define KQ_RES(RR)                                                           \
  if (g.D == RR) {                                                           \
    if (swept)                                                               \
      launch_resident<RR, true>(s, g, ch, pl, window, tab, nchan, nblocks, chan_list, iir);  \
    else                                                                     \
      launch_resident<RR, false>(s, g, ch, pl, window, tab, nchan, nblocks, chan_list, iir); \
    return;                                                                  \
  }
  KQ_RES(256)
  KQ_RES(128)
  KQ_RES(64)
#undef KQ_RES
}

}  // namespace kq
