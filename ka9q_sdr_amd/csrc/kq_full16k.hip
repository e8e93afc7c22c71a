// kq_full16k.hip -- full-spectrum pre-detection filter for N = 16384, register-resident forward transform.
//
// Same contract as k_filter_full (kq_kernels.hip): per (channel, block) NCO mix (radio.c:132-139), the N-point forward
// transform of execute_filter_input (filter.c:151), compute_n0 over all N bins (radio.c:383-425), response multiply,
// CROSS_CONJ and the N/D-point inverse transform of execute_filter_output (filter.c:206-250).  It exists because
// compute_n0 -- which the reference's demodulator threads run on every block -- needs every bin, so the pruned
// kernels cannot serve it.
//
// N = 32 * 32 * 16.  With n = 512 n1 + 16 n2 + n3 and k = k1 + 32 k2 + 1024 k3
//   X[k] = sum_n3 W16^{n3 k3} W512^{n3 k2} sum_n2 W32^{n2 k2} W_N^{(16 n2 + n3) k1} sum_n1 W32^{n1 k1} x[n]
// 512 threads; thread t = 16 n2 + n3 loads its 32 samples x[512 n1 + t] (coalesced), mixes them and runs the
// 32-point transform over n1 in registers; the twiddle W_N^{t k1} = W^{(k1 & 3) t} W^{(k1 & ~3) t} is one product of
// two entries of a small per-thread table (coalesced loads).  Complex values are 2-vectors: the arithmetic is
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.
// Two transposes through LDS (each in two half rounds, so the buffer is 68 KiB and two workgroups share a CU) feed the 32-point transform over n2 and the two 16-point transforms over n3.  Every bin ends up in a register:
// compute_n0 is two block reductions, and the N/D bins the slave reads are dropped into LDS for the existing
// multiply / inverse-transform epilogue.
//
// NCO: without sweep the phasor of sample 512 n1 + t is P_t S^{n1}; S^{n1} is built from S, S^2, S^4, S^8, S^16
// (each from the double-precision phase), at most four products deep.  Swept channels and the first block after a
// retune (history still on the old oscillator) evaluate the closed-form phase per sample, as k_filter_full does.
#include "kq_device.hpp"
#include "kq_ldsfft.hpp"
#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kq_lane.hpp"
#include "kq_regfft.hpp"

namespace kq {

namespace {

constexpr int kN = 16384, kT = 512;
// Both transposes move complex (8-byte) elements in two half rounds, so the buffer holds half of the data.
constexpr int kRow1 = 528;    // transpose 1: [k1 & 15][16 n2 + n3]; rows 4224 B apart alternate 128-byte bank halves
constexpr int kCol2 = 33;     // transpose 2: [n3][33 (k1 & 15) + k2], rows kRow2 apart: 16 n3 x 2 k1 lanes hit
constexpr int kRow2 = 546;    //   32 distinct 8-byte bank pairs (546 = 2 mod 32, 33 = 1 mod 32)
constexpr int kXchElems = 16 * kRow2;  // float2 elements; 16 * kRow1 fits as well

using rfft::pk_cmul;
using rfft::v2f;

__device__ __forceinline__ v2f ld2(const float2 *p) {
  float2 const f = *p;
  return (v2f){f.x, f.y};
}

// Twiddle tables (float2), computed in double on the host.  Pass 1, thread t: lo1[l-1][t] = W_N^{l t}, l = 1..3;
// hi1[h-1][t] = W_N^{4 h t}, h = 1..7.  Pass 2, n3 = t & 15: lo2[l-1][n3] = W_N^{32 l n3}, hi2[h-1][n3] = W_N^{128 h n3}.
// (Loading all 31 twiddles of a pass directly -- 31 coalesced loads, no products -- was measured 20 % slower: the
// loads' latency is exposed, the 24 extra products are not.)
constexpr int kTabLo1 = 0, kTabHi1 = 3 * kT, kTabLo2 = 10 * kT, kTabHi2 = 10 * kT + 3 * 16, kTabSize = 10 * kT + 10 * 16;

// v[q] *= W^{q}, q = 0..31, W^{q} = lo[q & 3] * hi[q >> 2]; lo/hi rows are `stride` entries apart
__device__ __forceinline__ void twiddle32(v2f (&v)[32], const float2 *__restrict__ lo_tab, const float2 *__restrict__ hi_tab,
                                          int stride) {
  v2f lo[4];
#pragma unroll
  for (int l = 1; l < 4; l++) lo[l] = ld2(lo_tab + (l - 1) * stride);
#pragma unroll
  for (int l = 1; l < 4; l++) v[l] = pk_cmul(v[l], lo[l]);
#pragma unroll
  for (int h = 1; h < 8; h++) {
    v2f const hi = ld2(hi_tab + (h - 1) * stride);
    v[4 * h] = pk_cmul(v[4 * h], hi);
#pragma unroll
    for (int l = 1; l < 4; l++) v[4 * h + l] = pk_cmul(v[4 * h + l], pk_cmul(hi, lo[l]));
  }
}

__device__ __forceinline__ v2f phasor2(double turns) {
  float2 const p = phasor_turns(turns);
  return (v2f){p.x, p.y};
}

}  // namespace

// grid (channel, block); dynamic LDS = kXchElems float2 (the epilogue's 2 * N_dec float2 fit in it)
__global__ __launch_bounds__(kT, 4) void k_filter_full16k(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ window,
                                                       const float2 *__restrict__ tw, const float2 *__restrict__ tab,
                                                       int compute_n0, float2 *__restrict__ spec_dump, int spec_ch,
                                                       const int *__restrict__ chan_list) {
  extern __shared__ __attribute__((aligned(16))) float2 xch[];
  __shared__ float red_f[2][kT / 64];  // compute_n0: one slot per wave and pass
  __shared__ int red_i[2][kT / 64];
  int const c = chan_list ? chan_list[blockIdx.x] : (int)blockIdx.x, b = blockIdx.y;
  int const t = threadIdx.x;
  int const Ndec = g.Ndec;

  // ---------------- load + NCO mix (radio.c:132-139), samples n = 512 n1 + t into v[bitrev5(n1)]
  v2f v[32];
  {
    double const ph0 = ch.lo_phase[c], f0 = ch.lo_freq[c], r = ch.lo_rate[c];
    double const hp0 = ch.hist_phase[c], hf0 = ch.hist_freq[c], hr = ch.hist_rate[c];
    const float2 *x = window + (size_t)b * g.L + t;
    double const mbase = (double)b * g.L;
    bool const retuned = b == 0 && (hp0 != ph0 || hf0 != f0 || hr != r);
    if (r == 0.0 && !retuned) {
      v2f const pt = phasor2(ph0 + f0 * (mbase + t));
      // S^(2^j), j = 0..4, S = exp(j 2 pi 512 f0): lane j of every wave evaluates one, the wave shares them
      v2f const sj = phasor2(f0 * (double)(512 << (t & 7)));
      auto rl = [](float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); };
      auto bc = [&](int lane) { return (v2f){rl(sj.x, lane), rl(sj.y, lane)}; };
      v2f const s1 = bc(0), s2 = bc(1), s4 = bc(2), s8 = bc(3), s16 = bc(4);
      v2f lo[4];
      lo[0] = pt;
      lo[1] = pk_cmul(pt, s1);
      lo[2] = pk_cmul(pt, s2);
      lo[3] = pk_cmul(lo[2], s1);
#pragma unroll
      for (int h = 0; h < 8; h++) {
        // S^{4h} from s4, s8, s16
        v2f hi = (v2f){1.f, 0.f};
        if (h & 1) hi = s4;
        if (h & 2) hi = (h & 1) ? pk_cmul(hi, s8) : s8;
        if (h & 4) hi = (h & 3) ? pk_cmul(hi, s16) : s16;
#pragma unroll
        for (int l = 0; l < 4; l++) {
          int const n1 = 4 * h + l;
          v2f const p = h ? pk_cmul(lo[l], hi) : lo[l];
          v[rfft::bitrev5(n1)] = pk_cmul(ld2(x + 512 * n1), p);
        }
      }
    } else {
#pragma unroll
      for (int n1 = 0; n1 < 32; n1++) {
        int const i = 512 * n1 + t;
        double const m = mbase + i;
        bool const old = (b == 0) && i < g.M - 1;  // history of the call's first block: pre-retune oscillator
        double const rr = old ? hr : r;
        double turns = old ? hp0 + hf0 * m : ph0 + f0 * m;
        if (rr != 0.0) turns += rr * (0.5 * m * (m - 1.0));
        v[rfft::bitrev5(n1)] = pk_cmul(ld2(x + 512 * n1), phasor2(turns));
      }
    }
  }

  // ---------------- pass 1: 32-point transforms over n1, twiddle W_N^{t k1}
  rfft::fft_dit_pk<32>(v);
  twiddle32(v, tab + kTabLo1 + t, tab + kTabHi1 + t, kT);

  // ---------------- transpose 1: [k1][t] -> thread (k1 = t >> 4, n3 = t & 15) gathers n2 = 0..31.
  // Half round A carries k1 < 16 (read by threads t < 256), half round B the rest.
  v2f u[32];
  {
    int const rd = ((t >> 4) & 15) * kRow1 + (t & 15);
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) xch[k1 * kRow1 + t] = make_float2(v[16 * half + k1].x, v[16 * half + k1].y);
      __syncthreads();
      if ((t >> 8) == half) {
#pragma unroll
        for (int n2 = 0; n2 < 32; n2++) u[rfft::bitrev5(n2)] = ld2(xch + rd + 16 * n2);
      }
      __syncthreads();
    }
  }

  // ---------------- pass 2: 32-point transforms over n2, twiddle W_512^{n3 k2} = W_N^{32 n3 k2}
  rfft::fft_dit_pk<32>(u);
  twiddle32(u, tab + kTabLo2 + (t & 15), tab + kTabHi2 + (t & 15), 16);

  // ---------------- transpose 2: [n3][k1][k2] -> thread (k1 = t >> 5 (+16), k2 = t & 31) gathers n3 = 0..15.
  // Half round A is written by the threads holding k1 < 16 (t < 256) and yields ya, half round B yields yb.
  v2f ya[16], yb[16];
  {
    int const wr = (t & 15) * kRow2 + ((t >> 4) & 15) * kCol2;
    int const rd = (t >> 5) * kCol2 + (t & 31);
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if ((t >> 8) == half) {
#pragma unroll
        for (int k2 = 0; k2 < 32; k2++) xch[wr + k2] = make_float2(u[k2].x, u[k2].y);
      }
      __syncthreads();
#pragma unroll
      for (int n3 = 0; n3 < 16; n3++) (half ? yb : ya)[rfft::bitrev4(n3)] = ld2(xch + n3 * kRow2 + rd);
      __syncthreads();
    }
  }

  // ---------------- pass 3: 16-point transforms over n3.  ya[k3] = X[ka + 1024 k3], yb[k3] = X[kb + 1024 k3]
  rfft::fft_dit_pk<16>(ya);
  rfft::fft_dit_pk<16>(yb);
  int const ka = (t >> 5) + 32 * (t & 31), kb = ka + 16;  // k1 + 32 k2 with k1 = t >> 5 and 16 + (t >> 5)

  if (spec_dump != nullptr && c == spec_ch) {
    float2 *o = spec_dump + (size_t)b * kN;
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      o[ka + 1024 * k3] = make_float2(ya[k3].x, ya[k3].y);
      o[kb + 1024 * k3] = make_float2(yb[k3].x, yb[k3].y);
    }
  }

  // ---------------- compute_n0 (radio.c:383-425), status only
  if (compute_n0) {
    float const low = ch.low[c], high = ch.high[c];
    unsigned incl_a = 0, incl_b = 0;  // bit k3: bin outside the passband
    float pa[16], pb[16];
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      pa[k3] = ya[k3].x * ya[k3].x + ya[k3].y * ya[k3].y;
      pb[k3] = yb[k3].x * yb[k3].x + yb[k3].y * yb[k3].y;
    }
    if (ch.n0mask) {
      // precomputed per channel on the host (kq_bank.cpp upload_n0mask): it depends only on the filter edges
      unsigned const m = ch.n0mask[(size_t)c * kT + t];
      incl_a = m & 0xffffu;
      incl_b = m >> 16;
    } else {
      // The reference forms k*samprate in int (radio.c:407,409) with k the signed bin: keep its 32-bit wrap.
      // n*samprate - (n > N/2 ? N*samprate : 0) modulo 2^32, built by additions from the thread's first bin.
      unsigned const sr = (unsigned)g.samprate;
      unsigned const prod_a0 = (unsigned)ka * sr, prod_b0 = (unsigned)kb * sr;
#pragma unroll
      for (int k3 = 0; k3 < 16; k3++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
          // n = k + 1024 k3 <= N/2  <=>  k3 < 8, or k3 == 8 and k == 0 (only ka can be 0)
          bool const neg = k3 > 8 || (k3 == 8 && (half || ka != 0));
          unsigned const prod = (half ? prod_b0 : prod_a0) + (unsigned)(1024 * k3) * sr - (neg ? (unsigned)kN * sr : 0u);
          float const f = (float)(int)prod / kN;
          if (!(f >= low && f <= high)) (half ? incl_b : incl_a) |= 1u << k3;
        }
      }
    }
    // bins inside the passband never count: +inf fails both passes' `< thr` (as do the NaN / inf bins the reference's
    // comparison drops)
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      pa[k3] = ((incl_a >> k3) & 1) ? pa[k3] : INFINITY;
      pb[k3] = ((incl_b >> k3) & 1) ? pb[k3] : INFINITY;
    }
    float avg = INFINITY;
    // Both passes run the same loop body.  (Left to itself the compiler peels the first one and, with thr = inf known,
    // counts bins with `p != inf` -- which a NaN bin would pass -- while still summing with an ordered compare.)
    asm volatile("" : "+v"(avg));
    for (int iter = 0; iter < 2; iter++) {
      float acc = 0;
      int wave_bins = 0;  // counted on the scalar unit from the comparison masks
      float const thr = avg * 2;
#pragma unroll
      for (int k3 = 0; k3 < 16; k3++) {
        bool const ta = pa[k3] < thr, tb = pb[k3] < thr;
        acc += ta ? pa[k3] : 0.f;
        acc += tb ? pb[k3] : 0.f;
        wave_bins += __popcll(__ballot(ta)) + __popcll(__ballot(tb));
      }
      acc = wave_sum(acc);
      if ((t & 63) == 0) {
        red_f[iter][t >> 6] = acc;
        red_i[iter][t >> 6] = wave_bins;
      }
      __syncthreads();
      float tf = 0;
      int bins = 0;
#pragma unroll
      for (int k = 0; k < kT / 64; k++) {
        tf += red_f[iter][k];
        bins += red_i[iter][k];
      }
      avg = tf / bins;
    }
    if (t == 0) pl.n0raw[(size_t)c * g.max_blocks + b] = (float)(avg / (2.0 * kN * g.samprate));
  }

  // ---------------- slave (filter.c:206-250): the N/D bins it reads go to LDS as Xs[p], p = k mod N_dec
  float2 *Xs = xch;
  float2 *G = Xs + Ndec;
  auto to_f2 = [](v2f a) { return make_float2(a.x, a.y); };
  if (Ndec <= 1024) {
    // the slave's bins all lie in the first and the last 1024: rows k3 = 0 and k3 = 15 only
    if (ka <= Ndec / 2) Xs[ka] = to_f2(ya[0]);
    if (kb <= Ndec / 2) Xs[kb] = to_f2(yb[0]);
    if (ka + 15 * 1024 > kN - Ndec / 2) Xs[ka - 1024 + Ndec] = to_f2(ya[15]);
    if (kb + 15 * 1024 > kN - Ndec / 2) Xs[kb - 1024 + Ndec] = to_f2(yb[15]);
  } else {
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      // bins of this k3 lie in [1024 k3, 1024 k3 + 1023]: skip the rows that cannot hold a bin the slave reads
      if (1024 * k3 > Ndec / 2 && 1024 * k3 + 1023 <= kN - Ndec / 2) continue;
#pragma unroll
      for (int half = 0; half < 2; half++) {
        int const n = (half ? kb : ka) + 1024 * k3;
        float2 const val = to_f2(half ? yb[k3] : ya[k3]);
        if (n <= Ndec / 2)
          Xs[n] = val;
        else if (n > kN - Ndec / 2)
          Xs[n - kN + Ndec] = val;
      }
    }
  }
  __syncthreads();
  const float2 *H = ch.resp + (size_t)c * Ndec;
  bool const isb = (ch.flags[c] & FLAG_ISB) != 0;
  if (Ndec == 64) {
    // cfg 3 / 4: one wave multiplies and runs the 64-point inverse transform in its registers (lane exchanges, no
    // barriers); the other seven are done
    if (t >= 64) return;
    int const q = (int)(__brev((unsigned)t) >> 26);  // decimation in time: bit-reversed in, natural out
    float2 z = cmul(H[q], Xs[q]);
    if (isb && q != 0 && q != 32) {  // filter.c:242-248
      float2 const other = cmul(H[64 - q], Xs[64 - q]);
      z = q < 32 ? cadd(z, cconj(other)) : csub(z, cconj(other));
    }
    auto xch2 = [&](float2 v, auto m) {
      return make_float2(lane_xor<decltype(m)::value>(v.x, t), lane_xor<decltype(m)::value>(v.y, t));
    };
    auto stage = [&](auto m) {
      constexpr int half = decltype(m)::value;
      float sn, cs;
      sincospif((float)(t & (half - 1)) / (float)half, &sn, &cs);
      bool const up = (t & half) != 0;
      float2 const v = up ? cmul(z, make_float2(cs, sn)) : z;
      float2 const r = xch2(v, m);
      z = up ? csub(r, v) : cadd(v, r);
    };
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 8>{});
    stage(std::integral_constant<int, 16>{});
    stage(std::integral_constant<int, 32>{});
    float2 *o = pl.filt + ((size_t)c * g.max_blocks + b) * g.olen;
    if (t >= 64 - g.olen) o[t - (64 - g.olen)] = z;  // filter.c:131
    return;
  }
  for (int p = t; p <= Ndec / 2; p += kT) {
    float2 gp = cmul(H[p], Xs[p]);
    if (p > 0 && p < Ndec / 2) {
      int const k = Ndec - p;
      float2 gn = cmul(H[k], Xs[k]);
      if (isb) {
        float2 const pos = gp, neg = gn;
        gp = cadd(pos, cconj(neg));
        gn = csub(neg, cconj(pos));
      }
      G[bitrev((unsigned)k, g.log2Ndec)] = gn;
    }
    G[bitrev((unsigned)p, g.log2Ndec)] = gp;
  }
  lds_fft<+1>(G, g.log2Ndec, tw, g.tw_log2);  // filter.c:250

  float2 *o = pl.filt + ((size_t)c * g.max_blocks + b) * g.olen;
  for (int i = t; i < g.olen; i += kT) o[i] = G[Ndec - g.olen + i];  // filter.c:131
}

bool full16k_supported(const Geom &g) {
  // the epilogue keeps Xs[N_dec] and G[N_dec] in the exchange buffer
  return g.N == kN && 2 * g.Ndec <= kXchElems && g.Ndec >= 4;
}

// The twiddle tables depend on nothing but N: one copy per device, built on first use.
static const float2 *twiddle_tables() {
  static std::mutex mu;
  static float2 *tabs[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (tabs[dev]) return tabs[dev];
  std::vector<float2> h(kTabSize);
  auto w = [](long long e) {
    double const ang = -2.0 * M_PI * (double)(e % kN) / kN;
    return make_float2((float)cos(ang), (float)sin(ang));
  };
  for (int t = 0; t < kT; t++) {
    for (int l = 1; l < 4; l++) h[kTabLo1 + (l - 1) * kT + t] = w((long long)l * t);
    for (int hh = 1; hh < 8; hh++) h[kTabHi1 + (hh - 1) * kT + t] = w(4LL * hh * t);
  }
  for (int n3 = 0; n3 < 16; n3++) {
    for (int l = 1; l < 4; l++) h[kTabLo2 + (l - 1) * 16 + n3] = w(32LL * l * n3);
    for (int hh = 1; hh < 8; hh++) h[kTabHi2 + (hh - 1) * 16 + n3] = w(128LL * hh * n3);
  }
  float2 *d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(float2)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return nullptr;
  }
  tabs[dev] = d;
  return d;
}

void launch_filter_full16k(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                           const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                           const int *chan_list) {
  size_t const lds_bytes = (size_t)kXchElems * sizeof(float2);
  ensure_dynamic_lds((const void *)k_filter_full16k, (size_t)(lds_bytes));
  const float2 *tab = twiddle_tables();
  if (!tab) {  // cannot happen short of an allocation failure: fall back to the LDS kernel rather than fail the block
    launch_filter_full(s, g, ch, pl, window, tw, nchan, nblocks, compute_n0, spec_dump, spec_ch, chan_list);
    return;
  }
  hipLaunchKernelGGL(k_filter_full16k, dim3(nchan, nblocks), dim3(kT), lds_bytes, s, g, ch, pl, window, tw, tab,
                     compute_n0, spec_dump, spec_ch, chan_list);
}

}  // namespace kq
