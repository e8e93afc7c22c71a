// kq_pll.hip -- linear demodulator with carrier tracking (linear.c:129-246): modes CAM, AME, CISB (opt.pll) and the
// squaring loop of DSB (opt.square).
//
// One workgroup per PLL channel, blocks in sequence.  Per block, as the reference:
//   1. append the filter output (squared in the squaring loop) to a 65536-sample ring           linear.c:131-153
//   2. lock detector with hysteresis on the PREVIOUS block's SNR                                linear.c:157-170
//   3. while unlocked, at most every 32768 new samples: transform the ring and steer the coarse NCO to the strongest
//      bin within +-300 Hz.  Only those bins are needed, so the 65536-point transform is done as four 16384-point
//      transforms in LDS (decimation in time) that are combined for the search window only     linear.c:173-200
//   4. spin the block by coarse*fine, sum it (or its square) -> carrier phase                   linear.c:208-223
//   5. second-order loop filter once per block -> fine NCO                                     linear.c:228-245
//   6. hang AGC, shift NCO, mono / stereo out, SNR = sum(I^2)/sum(Q^2) - 1                      linear.c:248-309
// The two NCOs are kept as (phase, step) pairs; the reference's sample-by-sample double phasor recurrences are
// evaluated in closed form inside the block.
#include "kq_device.hpp"

namespace kq {

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float cnrm(float2 a) { return a.x * a.x + a.y * a.y; }
__device__ __forceinline__ unsigned bitrev(unsigned i, int bits) { return __brev(i) >> (32 - bits); }
__device__ __forceinline__ float2 unit(double turns) {
  turns -= rint(turns);
  float s, c;
  sincospif(2.f * (float)turns, &s, &c);
  return make_float2(c, s);
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 16384-point forward FFT in LDS (bit-reversed in, natural out), radix 2^2 like kq_kernels.hip
__device__ void fft16k(float2 *s, const float2 *__restrict__ tw, int tw_log2) {
  constexpr int LOG = 14, n = 1 << LOG;
  __syncthreads();
  for (int stage = 0; stage < LOG; stage += 2) {
    int const m = 1 << stage;
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x) {
      int const j = i & (m - 1);
      int const base = ((i >> stage) << (stage + 2)) + j;
      float2 const w2 = tw[(size_t)j << (tw_log2 - stage - 1)];
      float2 const w4 = tw[(size_t)j << (tw_log2 - stage - 2)];
      float2 const a0 = s[base], a1 = cmul(s[base + m], w2);
      float2 const a2 = s[base + 2 * m], a3 = cmul(s[base + 3 * m], w2);
      float2 const b0 = cadd(a0, a1), b1 = csub(a0, a1), b2 = cadd(a2, a3), b3 = csub(a2, a3);
      float2 const c2 = cmul(b2, w4);
      float2 c3 = cmul(b3, w4);
      c3 = make_float2(c3.y, -c3.x);
      s[base] = cadd(b0, c2);
      s[base + 2 * m] = csub(b0, c2);
      s[base + m] = cadd(b1, c3);
      s[base + 3 * m] = csub(b1, c3);
    }
    __syncthreads();
  }
}

}  // namespace

// dynamic LDS: 16384 float2 (search transform) + olen float2 (the block)
__global__ void __launch_bounds__(1024) k_demod_linear_pll(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ tw,
                                                           const int *__restrict__ list, const PllChunk *__restrict__ chunks,
                                                           const int *__restrict__ slot_of, int nblocks, int compute_n0) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  __shared__ float red_a[16], red_b[16];
  __shared__ int red_i[16];
  constexpr int FS = 1 << 16;  // linear.c:43
  int const c = list[blockIdx.x], slot = slot_of[c];  // the channel's slot: its for as long as it tracks a carrier
  PllChunk const ck = chunks[slot / kPllChunk];
  PllState *const states = ck.state;
  int const sl = slot % kPllChunk;
  int const tid = threadIdx.x, nthr = blockDim.x, nw = nthr >> 6;
  int const olen = g.olen;
  float2 *S = lds + 16384;
  float2 *ring = ck.rings + (size_t)sl * FS;
  float2 *side = ck.side + (size_t)sl * 4096;
  PllState st = states[sl];

  bool const square = (ch.flags[c] & FLAG_SQUARE) != 0;
  bool const stereo = (ch.flags[c] & FLAG_STEREO) != 0;
  float const headroom = ch.headroom[c], recovery = ch.recovery[c];
  int const hangmax = ch.hangmax[c];
  double const sh_ph = ch.sh_phase[c], sh_f = ch.sh_freq[c];
  float gain = ch.gain[c];
  int hang = ch.hang[c];
  float n0 = ch.n0[c];

  // constants of linear.c:26-67, same float arithmetic
  float const samptime = (float)g.D / (float)g.samprate;
  float const blocktime = samptime * g.L;
  float const snrthresh = powf(10.f, (float)(3. / 10));
  int const lock_limit = (int)round(1 / samptime);
  float const binsize = (float)(1. / (FS * samptime));
  int const sq = square ? 2 : 1;
  int const lowlimit = (int)round(sq * -300.f / binsize), highlimit = (int)round(sq * 300.f / binsize);
  float const vcogain = (float)(2 * M_PI), natfreq = (float)(1 * 2 * M_PI);
  float const tau1 = vcogain * 1.f / (natfreq * natfreq);
  float const integrator_gain = 1 / tau1;
  float const tau2 = (float)(2 * M_SQRT1_2 / natfreq);
  float const prop_gain = tau2 / tau1;

  for (int b = 0; b < nblocks; b++) {
    const float2 *in = pl.filt + ((size_t)c * g.max_blocks + b) * olen;
    for (int i = tid; i < olen; i += nthr) {
      float2 const s = in[i];
      S[i] = s;
      ring[(st.fft_ptr + i) & (FS - 1)] = square ? cmul(s, s) : s;  // linear.c:135-152
    }
    st.fft_ptr = (st.fft_ptr + olen) & (FS - 1);
    st.fft_samples = min(st.fft_samples + olen, FS);
    // lock detector (linear.c:157-170)
    if (st.snr < snrthresh)
      st.lock_count -= olen;
    else
      st.lock_count += olen;
    if (st.lock_count >= lock_limit) {
      st.lock_count = lock_limit;
      st.pll_lock = 1;
    }
    if (st.lock_count <= -lock_limit) {
      st.lock_count = -lock_limit;
      st.pll_lock = 0;
    }
    if (!st.pll_lock && st.fft_samples > FS / 2) {  // carrier search (linear.c:173-200)
      st.fft_samples = 0;
      int const nbins = highlimit - lowlimit + 1;
      __threadfence_block();
      __syncthreads();  // ring writes of this block are visible to the workgroup
      for (int i = tid; i < nbins; i += nthr) side[i] = make_float2(0.f, 0.f);
      for (int s4 = 0; s4 < 4; s4++) {
        __syncthreads();
        for (int i = tid; i < 16384; i += nthr) lds[bitrev((unsigned)i, 14)] = ring[4 * i + s4];
        fft16k(lds, tw, g.tw_log2);
        for (int i = tid; i < nbins; i += nthr) {
          int const k = lowlimit + i;                              // signed bin of the 65536-point transform
          int const src = k & 16383;                               // k mod 16384
          int const idx = (int)(((long long)s4 * k) & (FS - 1));   // W_65536^{s4 k}
          float2 w = tw[(size_t)(idx & (FS / 2 - 1)) << (g.tw_log2 - 16)];
          if (idx >= FS / 2) w = make_float2(-w.x, -w.y);
          side[i] = cadd(side[i], cmul(w, lds[src]));
        }
      }
      __syncthreads();
      // first bin, scanning lowlimit..highlimit, holding the largest energy (linear.c:181-189)
      float pe = 0;
      int pb = 1 << 30;
      for (int i = tid; i < nbins; i += nthr) {
        float const e = cnrm(side[i]);
        if (e > pe) {
          pe = e;
          pb = i;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float const oe = __shfl_xor(pe, o, 64);
        int const ob = __shfl_xor(pb, o, 64);
        if (oe > pe || (oe == pe && ob < pb)) {
          pe = oe;
          pb = ob;
        }
      }
      if ((tid & 63) == 0) {
        red_a[tid >> 6] = pe;
        red_i[tid >> 6] = pb;
      }
      __syncthreads();
      pe = 0;
      pb = 1 << 30;
      for (int k = 0; k < nw; k++)
        if (red_a[k] > pe || (red_a[k] == pe && red_i[k] < pb)) {
          pe = red_a[k];
          pb = red_i[k];
        }
      __syncthreads();
      if (pe > 0) {
        int const maxbin = lowlimit + pb;
        double new_delta_f = (double)binsize * maxbin;
        if (square) new_delta_f /= 2;
        if (new_delta_f != (double)st.delta_f) {
          st.delta_f = (float)new_delta_f;
          st.integrator = 0;
          st.c_freq = (double)(-samptime * st.delta_f);  // set_osc(&coarse, -samptime * delta_f, 0.0)
        }
      }
    }
    __syncthreads();
    // spin by coarse*fine and gather the carrier phase (linear.c:208-223)
    float ax = 0, ay = 0;
    for (int i = tid; i < olen; i += nthr) {
      float2 const rot = unit(st.c_phase + st.f_phase + (st.c_freq + st.f_freq) * (double)i);
      float2 const s = cmul(S[i], rot);
      S[i] = s;
      float2 const ss = square ? cmul(s, s) : s;
      ax += ss.x;
      ay += ss.y;
    }
    ax = wsum(ax);
    ay = wsum(ay);
    if ((tid & 63) == 0) {
      red_a[tid >> 6] = ax;
      red_b[tid >> 6] = ay;
    }
    __syncthreads();
    ax = 0;
    ay = 0;
    for (int k = 0; k < nw; k++) {
      ax += red_a[k];
      ay += red_b[k];
    }
    st.c_phase += st.c_freq * olen;  // frozen when the step is zero (osc.c:43)
    st.c_phase -= rint(st.c_phase);
    st.f_phase += st.f_freq * olen;
    st.f_phase -= rint(st.f_phase);
    float cphase = atan2f(ay, ax);
    if (isnan(cphase)) cphase = 0;
    if (square) cphase /= 2;
    st.cphase = cphase;
    // loop filter (linear.c:228-245); ramprate is 0 in the reference, so the sweep generator never moves
    st.integrator += cphase * blocktime + 0.f;
    float const feedback = integrator_gain * st.integrator + prop_gain * cphase;
    st.f_freq = (double)(-feedback * samptime);
    if (isnan(st.foffset))
      st.foffset = feedback + st.delta_f;
    else
      st.foffset += (float)(0.001 * (feedback + st.delta_f - st.foffset));
    __syncthreads();
    // hang AGC (linear.c:251-281): serial recurrence, one lane walks the block in LDS
    if (tid == 0) {
      float signal = 0, noise = 0;
      for (int i = 0; i < olen; i++) {
        float2 const s = S[i];
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
        S[i] = make_float2(s.x * gain, s.y * gain);
      }
      red_a[0] = signal;
      red_b[0] = noise;
      red_a[1] = gain;
      red_i[0] = hang;
    }
    __syncthreads();
    float const signal = red_a[0], noise = red_b[0];
    gain = red_a[1];
    hang = red_i[0];
    float *aud = pl.audio + ((size_t)c * g.max_blocks + b) * (2 * (size_t)olen);
    for (int i = tid; i < olen; i += nthr) {
      float2 s = S[i];
      if (sh_f != 0.0) s = cmul(s, unit(sh_ph + sh_f * ((double)b * olen + i)));  // linear.c:283-289
      if (stereo) {
        aud[2 * i] = s.x;
        aud[2 * i + 1] = s.y;
      } else {
        aud[i] = s.x;
      }
    }
    if (noise != 0) {  // linear.c:304-309
      st.snr = signal / noise - 1;
      if (st.snr < 0) st.snr = 0;
    } else {
      st.snr = NAN;
    }
    if (tid == 0) {
      kq_chan_status o;
      o.if_power = pl.if_power[b];
      o.noise_gain = ch.noise_gain[c];
      o.plfreq = NAN;
      if (compute_n0) {
        float const fresh = pl.n0raw[(size_t)c * g.max_blocks + b];
        n0 = isnan(n0) ? fresh : (float)((double)n0 + .001 * (double)(fresh - n0));  // linear.c:124, double literal
        o.n0 = n0;
      } else {
        o.n0 = NAN;
      }
      o.bb_power = (signal + noise) / (2 * olen);
      o.snr = st.snr;
      o.foffset = st.foffset;
      o.pdeviation = 0;
      o.agc_gain = gain;
      o.cphase = st.cphase;
      o.pll_lock = st.pll_lock;
      o.lock_count = st.lock_count;
      o.squelch_count = 0;
      o.hangcount = hang;
      o.blanked = 0;
      o.nout = stereo ? 2 * olen : olen;
      pl.status[(size_t)c * g.max_blocks + b] = o;
    }
    __syncthreads();
  }
  if (tid == 0) {
    states[sl] = st;
    ch.gain[c] = gain;
    ch.hang[c] = hang;
    ch.n0[c] = n0;
  }
}

void launch_demod_pll(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *tw, const int *list_pll,
                      int n_pll, const PllChunk *chunks, const int *slot_of, int nblocks, int compute_n0) {
  if (n_pll <= 0) return;
  size_t const lds_bytes = ((size_t)16384 + g.olen) * sizeof(float2);
  ensure_dynamic_lds((const void *)k_demod_linear_pll, lds_bytes);
  hipLaunchKernelGGL(k_demod_linear_pll, dim3(n_pll), dim3(1024), lds_bytes, s, g, ch, pl, tw, list_pll, chunks, slot_of,
                     nblocks, compute_n0);
}

}  // namespace kq
