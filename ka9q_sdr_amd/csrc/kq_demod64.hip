// kq_demod64.hip -- demodulators specialised for N/D = 64, olen = 32 (BASELINE configs 3 and 4).
//
// One launch serves all three demodulator types (workgroup ranges by type) so they overlap on the chip.
//   FM     : one wave per channel, one lane per sample; everything stays in registers: amplitude statistics
//            by wave reductions, the "hold last good sample" rule of fm.c:128-144 through a ballot mask
//            (previous valid sample = highest set bit below the lane), and the REAL->REAL de-emphasis
//            overlap-save (fm.c:162-171) as 64-point transforms across the 64 lanes (history in lanes 0-31,
//            the new block in lanes 32-63).  No LDS, no barriers.
//   AM/lin : the AGC is a sequential recurrence (am.c:55-75, linear.c:251-281): one lane per channel, the
//            block's 32 samples fetched up front so the loop is pure ALU.
// Blocks of one channel are processed in sequence with the state carried in registers, and written back
// to HBM at the end of the launch.
#include "kq_device.hpp"

namespace kq {

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float cnrm(float2 a) { return a.x * a.x + a.y * a.y; }
__device__ __forceinline__ float2 shfl2(float2 v, int src) {
  return make_float2(__shfl(v.x, src, 64), __shfl(v.y, src, 64));
}
__device__ __forceinline__ float2 shfl2_xor(float2 v, int m) {
  return make_float2(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64));
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wmin(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int bitrev6(int i) { return (int)(__brev((unsigned)i) >> 26); }

__device__ __forceinline__ void put_status(kq_chan_status &st, const Geom &g, const ChanDev &ch, const Planes &pl, int c, int b,
                                           int compute_n0, float n0_rate, float &n0) {
  st.if_power = pl.if_power[b];
  st.noise_gain = ch.noise_gain[c];
  if (compute_n0) {
    float const fresh = pl.n0raw[(size_t)c * g.max_blocks + b];
    if (isnan(n0))
      n0 = fresh;
    else
      n0 += n0_rate * (fresh - n0);
    st.n0 = n0;
  } else {
    st.n0 = NAN;
  }
}

__device__ void fm_channel(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  int const lane = threadIdx.x & 63;
  bool const upper = lane >= 32;
  int const n = lane - 32;
  bool const flat = (ch.flags[c] & FLAG_FLAT) != 0;
  float const gain = ch.fm_gain[c];
  int const kbin = bitrev6(lane);
  float2 const HA = (!flat && kbin <= 32) ? ch.aresp[(size_t)c * 33 + kbin] : make_float2(0.f, 0.f);
  int const herm_src = bitrev6((64 - kbin) & 63);

  // per-lane stage twiddles: forward DIF (half = 32..1) and inverse DIT (half = 1..32)
  float2 wf[6], wi[6];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    int const half = 1 << s;
    float sn, cs;
    sincospif((float)(lane & (half - 1)) / (float)half, &sn, &cs);
    wf[s] = make_float2(cs, -sn);  // exp(-2 pi i j / (2 half))
    wi[s] = make_float2(cs, sn);
  }

  float2 state = ch.fm_state[c];
  float lastaudio = ch.lastaudio[c];
  int sq = ch.sq_count[c];
  float foffset = ch.foffset[c], pdev = ch.pdev[c];
  float n0 = ch.n0[c];
  float hist = upper ? 0.f : ch.ahist[(size_t)c * 32 + lane];

  const float2 *in = pl.filt + (size_t)c * g.max_blocks * 32;
  float2 s_next = upper ? in[n] : make_float2(0.f, 0.f);
  for (int b = 0; b < nblocks; b++) {
    float2 const S = s_next;
    if (b + 1 < nblocks) s_next = upper ? in[(size_t)(b + 1) * 32 + n] : make_float2(0.f, 0.f);
    float const t = upper ? cnrm(S) : 0.f;
    float const sum_t = wsum(t), sum_a = wsum(sqrtf(t));
    float const bb = sum_t / 64.f;                                   // / (2*olen), fm.c:99
    float const amp = (float)((double)sum_a / (M_SQRT2 * 32));       // fm.c:100
    float const variance = bb - amp * amp;
    float snr = amp * amp / (2 * variance) - 1;
    snr = (0.0f > snr) ? 0.0f : snr;
    if (snr > 2) {
      sq = 0;
    } else if (++sq > 1000) {
      sq = 1000;
    }
    float out = 0.f;
    int blanked = 0;
    if (sq < 2) {
      float const thr = (float)(0.55 * 0.55 * amp * amp);
      bool const valid = upper && t > thr;
      unsigned long long const mask = __ballot(valid);
      unsigned long long const below = mask & ((1ull << lane) - 1ull);
      unsigned long long const upto = mask & ((2ull << lane) - 1ull);
      int const pv = below ? 63 - __clzll((long long)below) : -1;
      int const lv = upto ? 63 - __clzll((long long)upto) : -1;
      float2 const sp = shfl2(S, pv >= 0 ? pv : 0);
      float2 const st = pv >= 0 ? cconj(sp) : state;
      float2 const pr = cmul(S, st);
      float const y = valid ? atan2f(pr.y, pr.x) : 0.f;
      float const yl = __shfl(y, lv >= 0 ? lv : 0, 64);
      out = upper ? (lv >= 0 ? yl : lastaudio) : 0.f;
      float const sum_y = wsum(out);
      float const vmax = wmax((valid && n > 0) ? y : -INFINITY);
      float const vmin = wmin((valid && n > 0) ? y : INFINITY);
      float const y0 = __shfl(y, 32, 64);
      float const seed = ((mask >> 32) & 1ull) ? y0 : 0.f;
      blanked = 32 - __popcll(mask);
      if (mask) {
        int const last = 63 - __clzll((long long)mask);
        state = cconj(shfl2(S, last));
        lastaudio = __shfl(y, last, 64);
      }
      float pdev_pos = fmaxf(seed, vmax), pdev_neg = fminf(seed, vmin);
      float const avg_f = sum_y / 32.f;
      if (sq < 1) {
        foffset = (float)(g.dsamprate * avg_f * (0.5 * M_1_PI));
        pdev_pos -= avg_f;
        pdev_neg -= avg_f;
        float const mx = (pdev_pos > -pdev_neg) ? pdev_pos : -pdev_neg;
        pdev = (float)(g.dsamprate * mx * (0.5 * M_1_PI));
      }
    } else {
      state = make_float2(0.f, 0.f);
      lastaudio = 0.f;
    }
    // post-detection overlap-save: [history | block] across the 64 lanes
    float audio = out;
    if (!flat) {
      float2 z = make_float2(upper ? out : hist, 0.f);
#pragma unroll
      for (int s = 5; s >= 0; s--) {  // forward, decimation in frequency: natural in, bit-reversed out
        int const half = 1 << s;
        float2 const r = shfl2_xor(z, half);
        z = ((lane >> s) & 1) ? cmul(csub(r, z), wf[s]) : cadd(z, r);
      }
      float2 gk = cmul(HA, z);  // bins 0..32 (filter.c:206-208); zero elsewhere
      if (kbin == 0 || kbin == 32) gk.y = 0.f;
      float2 const mirror = shfl2(gk, herm_src);
      if (kbin > 32) gk = cconj(mirror);  // Hermitian extension of the c2r transform
      z = gk;
#pragma unroll
      for (int s = 0; s < 6; s++) {  // backward, decimation in time: bit-reversed in, natural out
        int const half = 1 << s;
        int const bit = (lane >> s) & 1;
        float2 const v = bit ? cmul(z, wi[s]) : z;
        float2 const r = shfl2_xor(v, half);
        z = bit ? csub(r, v) : cadd(v, r);
      }
      audio = z.x * gain;  // fm.c:169-170
    }
    if (upper) pl.audio[((size_t)c * g.max_blocks + b) * 64 + n] = audio;
    hist = __shfl(out, lane + 32, 64);  // lanes 0..31 take this block as the next history (filter.c:168)
    if (lane == 0) {
      kq_chan_status st;
      put_status(st, g, ch, pl, c, b, compute_n0, .01f, n0);
      st.bb_power = bb;
      st.snr = snr;
      st.foffset = foffset;
      st.pdeviation = pdev;
      st.agc_gain = 0;
      st.squelch_count = sq;
      st.hangcount = 0;
      st.blanked = blanked;
      st.nout = 32;
      pl.status[(size_t)c * g.max_blocks + b] = st;
    }
  }
  if (!upper) ch.ahist[(size_t)c * 32 + lane] = hist;
  if (lane == 0) {
    ch.fm_state[c] = state;
    ch.lastaudio[c] = lastaudio;
    ch.sq_count[c] = sq;
    ch.foffset[c] = foffset;
    ch.pdev[c] = pdev;
    ch.n0[c] = n0;
  }
}

template <bool LINEAR>
__device__ void agc_channel(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  float const headroom = ch.headroom[c], recovery = ch.recovery[c];
  int const hangmax = ch.hangmax[c];
  bool const stereo = LINEAR && (ch.flags[c] & FLAG_STEREO) != 0;
  double const sh_ph = LINEAR ? ch.sh_phase[c] : 0.0, sh_f = LINEAR ? ch.sh_freq[c] : 0.0;
  float gain = ch.gain[c], dc = LINEAR ? 0.f : ch.dc[c];
  int hang = ch.hang[c];
  float n0 = ch.n0[c];
  for (int b = 0; b < nblocks; b++) {
    const float4 *in = reinterpret_cast<const float4 *>(pl.filt + ((size_t)c * g.max_blocks + b) * 32);
    float4 buf[16];
#pragma unroll
    for (int i = 0; i < 16; i++) buf[i] = in[i];
    float *aud = pl.audio + ((size_t)c * g.max_blocks + b) * 64;
    float signal = 0, noise = 0;
    float o[64];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      float2 s = (i & 1) ? make_float2(buf[i >> 1].z, buf[i >> 1].w) : make_float2(buf[i >> 1].x, buf[i >> 1].y);
      if (LINEAR) {
        float const rp = s.x * s.x, ip = s.y * s.y;
        signal += rp;
        noise += ip;
        float const amplitude = sqrtf(rp + ip);
        if (isnan(gain)) {
          gain = headroom / amplitude;
        } else if (amplitude * gain > headroom) {
          gain = headroom / amplitude;
          hang = hangmax;
        } else if (hang != 0) {
          hang--;
        } else {
          gain *= recovery;
        }
        s = make_float2(s.x * gain, s.y * gain);
        if (sh_f != 0.0) {
          double turns = sh_ph + sh_f * ((double)b * 32 + i);
          turns -= rint(turns);
          float sn, cs;
          sincospif(2.f * (float)turns, &sn, &cs);
          s = cmul(s, make_float2(cs, sn));
        }
        o[2 * i] = s.x;
        o[2 * i + 1] = s.y;
      } else {
        float const sq = cnrm(s);
        signal += sq;
        float const samp = sqrtf(sq);
        dc += 0.0001f * (samp - dc);
        if (isnan(gain)) {
          gain = headroom / dc;
        } else if (gain * dc > headroom) {
          gain = headroom / dc;
          hang = hangmax;
        } else if (hang != 0) {
          hang--;
        } else {
          gain *= recovery;
        }
        o[i] = (samp - dc) * gain;
      }
    }
    if (LINEAR) {
      if (stereo) {
        float4 *a4 = reinterpret_cast<float4 *>(aud);
#pragma unroll
        for (int i = 0; i < 16; i++) a4[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
      } else {
        float4 *a4 = reinterpret_cast<float4 *>(aud);
#pragma unroll
        for (int i = 0; i < 8; i++) a4[i] = make_float4(o[8 * i], o[8 * i + 2], o[8 * i + 4], o[8 * i + 6]);
      }
    } else {
      float4 *a4 = reinterpret_cast<float4 *>(aud);
#pragma unroll
      for (int i = 0; i < 8; i++) a4[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
    kq_chan_status st;
    put_status(st, g, ch, pl, c, b, compute_n0, .001f, n0);
    st.bb_power = (signal + noise) / 64.f;
    st.snr = LINEAR ? NAN : 0.f;
    st.foffset = 0;
    st.pdeviation = 0;
    st.agc_gain = gain;
    st.squelch_count = 0;
    st.hangcount = hang;
    st.blanked = 0;
    st.nout = stereo ? 64 : 32;
    pl.status[(size_t)c * g.max_blocks + b] = st;
  }
  ch.gain[c] = gain;
  if (!LINEAR) ch.dc[c] = dc;
  ch.hang[c] = hang;
  ch.n0[c] = n0;
}

}  // namespace

// grid = n_fm + ceil(n_am/64) + ceil(n_lin/64) workgroups of one wave
__global__ void __launch_bounds__(64) k_demod64(Geom g, ChanDev ch, Planes pl, const int *__restrict__ list_fm, int n_fm,
                                                const int *__restrict__ list_am, int n_am,
                                                const int *__restrict__ list_lin, int n_lin, int nblocks, int compute_n0) {
  int wg = blockIdx.x;
  if (wg < n_fm) {
    fm_channel(g, ch, pl, list_fm[wg], nblocks, compute_n0);
    return;
  }
  wg -= n_fm;
  int const am_wgs = (n_am + 63) / 64;
  if (wg < am_wgs) {
    int const i = wg * 64 + threadIdx.x;
    if (i < n_am) agc_channel<false>(g, ch, pl, list_am[i], nblocks, compute_n0);
    return;
  }
  wg -= am_wgs;
  int const i = wg * 64 + threadIdx.x;
  if (i < n_lin) agc_channel<true>(g, ch, pl, list_lin[i], nblocks, compute_n0);
}

bool demod64_supported(const Geom &g) { return g.Ndec == 64 && g.olen == 32 && g.Mdec == 33; }

void launch_demod64(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const int *list_fm, int n_fm,
                    const int *list_am, int n_am, const int *list_lin, int n_lin, int nblocks, int compute_n0) {
  int const wgs = n_fm + (n_am + 63) / 64 + (n_lin + 63) / 64;
  if (wgs == 0) return;
  hipLaunchKernelGGL(k_demod64, dim3(wgs), dim3(64), 0, s, g, ch, pl, list_fm, n_fm, list_am, n_am, list_lin, n_lin, nblocks,
                     compute_n0);
}

}  // namespace kq
