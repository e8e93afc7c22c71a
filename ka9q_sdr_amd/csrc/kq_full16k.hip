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
// 32-point transform over n1 in registers; the twiddle W_N^{t k1} is one product of two table entries per element.
// Two transposes through LDS (real and imaginary planes in turn, so the buffer is 66 KiB and two workgroups share a
// CU) feed the 32-point transform over n2 and the two 16-point transforms over n3.  Every bin ends up in a register:
// compute_n0 is two block reductions, and the N/D bins the slave reads are dropped into LDS for the existing
// multiply / inverse-transform epilogue.
//
// NCO: without sweep the phasor of sample 512 n1 + t is P_t S^{n1}; S^{n1} is built from S, S^2, S^4, S^8, S^16
// (each from the double-precision phase), at most four products deep.  Swept channels and the first block after a
// retune (history still on the old oscillator) evaluate the closed-form phase per sample, as k_filter_full does.
#include "kq_device.hpp"
#include "kq_ldsfft.hpp"
#include "kq_regfft.hpp"

namespace kq {

namespace {

constexpr int kN = 16384, kLog2N = 14, kT = 512;
constexpr int kRow1 = 528;    // exchange 1: [k1][16 n2 + n3], row padded so that 4 rows x 16 lanes hit 64 banks
constexpr int kRow2 = 1025;   // exchange 2: [n3][32 k1 + k2], row padded against the 1024-float stride
constexpr int kPlane = 32 * kRow1;  // floats; 16 * kRow2 = 16400 fits as well

// exp(-2 pi i idx / N) from the half-period table tw (period 1 << tw_log2)
__device__ __forceinline__ float2 twN(const float2 *__restrict__ tw, int sh, int idx) {
  idx &= kN - 1;
  float2 w = tw[(size_t)(idx & (kN / 2 - 1)) << sh];
  if (idx >= kN / 2) w = make_float2(-w.x, -w.y);
  return w;
}

// v[q] *= W_N^{base q}, q = 0..31: W^{q} = W^{q & 3} * W^{q & ~3}, both factors straight from the table
__device__ __forceinline__ void twiddle32(float2 (&v)[32], const float2 *__restrict__ tw, int sh, int base) {
  float2 lo[4];
#pragma unroll
  for (int l = 1; l < 4; l++) lo[l] = twN(tw, sh, base * l);
#pragma unroll
  for (int h = 0; h < 8; h++) {
    if (h == 0) {
#pragma unroll
      for (int l = 1; l < 4; l++) v[l] = cmul(v[l], lo[l]);
    } else {
      float2 const hi = twN(tw, sh, base * 4 * h);
      v[4 * h] = cmul(v[4 * h], hi);
#pragma unroll
      for (int l = 1; l < 4; l++) v[4 * h + l] = cmul(v[4 * h + l], cmul(hi, lo[l]));
    }
  }
}

}  // namespace

// grid (channel, block); dynamic LDS = kPlane floats (+ room for the epilogue: 2 * N_dec float2 <= that)
__global__ __launch_bounds__(kT, 4) void k_filter_full16k(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ window,
                                                       const float2 *__restrict__ tw, int compute_n0,
                                                       float2 *__restrict__ spec_dump, int spec_ch,
                                                       const int *__restrict__ chan_list) {
  extern __shared__ __attribute__((aligned(16))) float xch[];
  __shared__ float red_f[16];
  __shared__ int red_i[16];
  int const c = chan_list ? chan_list[blockIdx.x] : (int)blockIdx.x, b = blockIdx.y;
  int const t = threadIdx.x;
  int const sh = g.tw_log2 - kLog2N;
  int const Ndec = g.Ndec;

  // ---------------- load + NCO mix (radio.c:132-139), samples n = 512 n1 + t into v[bitrev5(n1)]
  float2 v[32];
  {
    double const ph0 = ch.lo_phase[c], f0 = ch.lo_freq[c], r = ch.lo_rate[c];
    double const hp0 = ch.hist_phase[c], hf0 = ch.hist_freq[c], hr = ch.hist_rate[c];
    const float2 *x = window + (size_t)b * g.L + t;
    double const mbase = (double)b * g.L;
    bool const retuned = b == 0 && (hp0 != ph0 || hf0 != f0 || hr != r);
    if (r == 0.0 && !retuned) {
      float2 const pt = phasor_turns(ph0 + f0 * (mbase + t));
      float2 const s1 = phasor_turns(f0 * 512.0), s2 = phasor_turns(f0 * 1024.0), s4 = phasor_turns(f0 * 2048.0),
                   s8 = phasor_turns(f0 * 4096.0), s16 = phasor_turns(f0 * 8192.0);
      float2 lo[4];
      lo[0] = pt;
      lo[1] = cmul(pt, s1);
      lo[2] = cmul(pt, s2);
      lo[3] = cmul(lo[2], s1);
#pragma unroll
      for (int h = 0; h < 8; h++) {
        // S^{4h} from s4, s8, s16
        float2 hi = make_float2(1.f, 0.f);
        if (h & 1) hi = s4;
        if (h & 2) hi = (h & 1) ? cmul(hi, s8) : s8;
        if (h & 4) hi = (h & 3) ? cmul(hi, s16) : s16;
#pragma unroll
        for (int l = 0; l < 4; l++) {
          int const n1 = 4 * h + l;
          float2 const p = h ? cmul(lo[l], hi) : lo[l];
          v[rfft::bitrev5(n1)] = cmul(x[512 * n1], p);
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
        v[rfft::bitrev5(n1)] = cmul(x[512 * n1], phasor_turns(turns));
      }
    }
  }

  // ---------------- pass 1: 32-point transforms over n1, twiddle W_N^{t k1}
  rfft::fft_dit<32>(v);
  twiddle32(v, tw, sh, t);

  // ---------------- transpose 1: [k1][t] -> thread (k1 = t >> 4, n3 = t & 15) gathers n2 = 0..31
  float2 u[32];
  {
    int const rd = (t >> 4) * kRow1 + (t & 15);
#pragma unroll
    for (int k1 = 0; k1 < 32; k1++) xch[k1 * kRow1 + t] = v[k1].x;
    __syncthreads();
#pragma unroll
    for (int n2 = 0; n2 < 32; n2++) u[rfft::bitrev5(n2)].x = xch[rd + 16 * n2];
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 32; k1++) xch[k1 * kRow1 + t] = v[k1].y;
    __syncthreads();
#pragma unroll
    for (int n2 = 0; n2 < 32; n2++) u[rfft::bitrev5(n2)].y = xch[rd + 16 * n2];
    __syncthreads();
  }

  // ---------------- pass 2: 32-point transforms over n2, twiddle W_512^{n3 k2} = W_N^{32 n3 k2}
  rfft::fft_dit<32>(u);
  twiddle32(u, tw, sh, 32 * (t & 15));

  // ---------------- transpose 2: [n3][32 k1 + k2] -> thread (k1 = t >> 5 (+16), k2 = t & 31) gathers n3 = 0..15
  float2 ya[16], yb[16];
  {
    int const wr = (t & 15) * kRow2 + (t >> 4) * 32;
#pragma unroll
    for (int k2 = 0; k2 < 32; k2++) xch[wr + k2] = u[k2].x;
    __syncthreads();
#pragma unroll
    for (int n3 = 0; n3 < 16; n3++) {
      ya[rfft::bitrev4(n3)].x = xch[n3 * kRow2 + t];
      yb[rfft::bitrev4(n3)].x = xch[n3 * kRow2 + t + 512];
    }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 32; k2++) xch[wr + k2] = u[k2].y;
    __syncthreads();
#pragma unroll
    for (int n3 = 0; n3 < 16; n3++) {
      ya[rfft::bitrev4(n3)].y = xch[n3 * kRow2 + t];
      yb[rfft::bitrev4(n3)].y = xch[n3 * kRow2 + t + 512];
    }
    __syncthreads();
  }

  // ---------------- pass 3: 16-point transforms over n3.  ya[k3] = X[ka + 1024 k3], yb[k3] = X[kb + 1024 k3]
  rfft::fft_dit<16>(ya);
  rfft::fft_dit<16>(yb);
  int const ka = (t >> 5) + 32 * (t & 31), kb = ka + 16;  // k1 + 32 k2 with k1 = t >> 5 and 16 + (t >> 5)

  if (spec_dump != nullptr && c == spec_ch) {
    float2 *o = spec_dump + (size_t)b * kN;
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      o[ka + 1024 * k3] = ya[k3];
      o[kb + 1024 * k3] = yb[k3];
    }
  }

  // ---------------- compute_n0 (radio.c:383-425), status only
  if (compute_n0) {
    float const low = ch.low[c], high = ch.high[c];
    unsigned incl_a = 0, incl_b = 0;  // bit k3: bin outside the passband
    float pa[16], pb[16];
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      pa[k3] = cnrm(ya[k3]);
      pb[k3] = cnrm(yb[k3]);
#pragma unroll
      for (int half = 0; half < 2; half++) {
        int const n = (half ? kb : ka) + 1024 * k3;
        int const k = (n <= kN / 2) ? n : n - kN;
        // the reference forms k*samprate in int (radio.c:407,409): keep its 32-bit wrap
        int const prod = (int)((unsigned)k * (unsigned)g.samprate);
        float const f = (float)prod / kN;
        if (!(f >= low && f <= high)) (half ? incl_b : incl_a) |= 1u << k3;
      }
    }
    float avg = INFINITY;
    for (int iter = 0; iter < 2; iter++) {
      float acc = 0;
      int bins = 0;
      float const thr = avg * 2;
#pragma unroll
      for (int k3 = 0; k3 < 16; k3++) {
        if (((incl_a >> k3) & 1) && pa[k3] < thr) {
          acc += pa[k3];
          bins++;
        }
        if (((incl_b >> k3) & 1) && pb[k3] < thr) {
          acc += pb[k3];
          bins++;
        }
      }
      block_sum_fi(acc, bins, red_f, red_i);
      avg = acc / bins;
    }
    if (t == 0) pl.n0raw[(size_t)c * g.max_blocks + b] = (float)(avg / (2.0 * kN * g.samprate));
  }

  // ---------------- slave (filter.c:206-250): the N/D bins it reads go to LDS as Xs[p], p = k mod N_dec
  float2 *Xs = reinterpret_cast<float2 *>(xch);
  float2 *G = Xs + Ndec;
#pragma unroll
  for (int k3 = 0; k3 < 16; k3++) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      int const n = (half ? kb : ka) + 1024 * k3;
      float2 const val = half ? yb[k3] : ya[k3];
      if (n <= Ndec / 2)
        Xs[n] = val;
      else if (n > kN - Ndec / 2)
        Xs[n - kN + Ndec] = val;
    }
  }
  __syncthreads();
  const float2 *H = ch.resp + (size_t)c * Ndec;
  bool const isb = (ch.flags[c] & FLAG_ISB) != 0;
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
  return g.N == kN && (size_t)2 * g.Ndec * sizeof(float2) <= (size_t)kPlane * sizeof(float) && g.Ndec >= 4;
}

void launch_filter_full16k(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                           const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                           const int *chan_list) {
  size_t const lds_bytes = (size_t)kPlane * sizeof(float);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void *)k_filter_full16k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    configured = true;
  }
  hipLaunchKernelGGL(k_filter_full16k, dim3(nchan, nblocks), dim3(kT), lds_bytes, s, g, ch, pl, window, tw, compute_n0,
                     spec_dump, spec_ch, chan_list);
}

}  // namespace kq
