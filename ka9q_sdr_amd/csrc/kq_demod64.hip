// kq_demod64.hip -- demodulators specialised for N/D = 64, olen = 32 (BASELINE configs 3 and 4).
//
// One launch serves all three demodulator types (workgroup ranges by type) so they overlap on the chip.
//   FM     : one wave per channel, one lane per sample, two blocks per iteration; everything stays in registers:
//            amplitude statistics by half-wave reductions, the "hold last good sample" rule of fm.c:128-144
//            through a ballot mask (previous valid sample = highest set bit below the lane), and the REAL->REAL
//            de-emphasis overlap-save (fm.c:162-171) of both blocks as ONE complex 64-point transform pair
//            across the 64 lanes.  No LDS, no barriers; lane exchanges by DPP / v_permlane (kq_lane.hpp).
//   AM/lin : one wave per channel, one lane per sample (two blocks per iteration); only the AGC recurrence
//            (am.c:55-75, linear.c:251-281) is serial, wave-uniform through v_readlane.
// Blocks of one channel are processed in sequence with the state carried in registers, and written back
// to HBM at the end of the launch.
#include "kq_device.hpp"
#include <cstdlib>
#include "kq_lane.hpp"

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
template <int M>
__device__ __forceinline__ float2 xor2(float2 v, int lane) {
  return make_float2(lane_xor<M>(v.x, lane), lane_xor<M>(v.y, lane));
}
// lane ^ (1 << s) with s a constant after unrolling
__device__ __forceinline__ float2 xor2_pow(float2 v, int s, int lane) {
  switch (s) {
    case 0: return xor2<1>(v, lane);
    case 1: return xor2<2>(v, lane);
    case 2: return xor2<4>(v, lane);
    case 3: return xor2<8>(v, lane);
    case 4: return xor2<16>(v, lane);
    default: return xor2<32>(v, lane);
  }
}
// value of a wave-uniform lane
__device__ __forceinline__ float rdlane(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ int bitrev6(int i) { return (int)(__brev((unsigned)i) >> 26); }

__device__ __forceinline__ void put_status(kq_chan_status &st, const Geom &g, const ChanDev &ch, const Planes &pl, int c, int b,
                                           int compute_n0, double n0_rate, float &n0) {
  st.if_power = pl.if_power[b];
  st.noise_gain = ch.noise_gain[c];
  st.plfreq = NAN;
  st.cphase = 0;
  st.pll_lock = 0;
  st.lock_count = 0;  // N/D = 64: the PL slave would have 2 points (fm.c:203), measurement off
  if (compute_n0) {
    float const fresh = pl.n0raw[(size_t)c * g.max_blocks + b];
    if (isnan(n0))
      n0 = fresh;
    else
      n0 = (float)((double)n0 + n0_rate * (double)(fresh - n0));  // the reference's rate is a double literal
    st.n0 = n0;
  } else {
    st.n0 = NAN;
  }
}

// reduction over the 32 lanes of each half of the wave (lane bit 5 selects the half), every lane gets its half's result
template <class Op>
__device__ __forceinline__ float hreduce(float v, Op op) {
  int const lane = threadIdx.x & 63;
  v = op(v, lane_xor<16>(v, lane));
  v = op(v, lane_xor<8>(v, lane));
  v = op(v, lane_xor<4>(v, lane));
  v = op(v, lane_xor<2>(v, lane));
  v = op(v, lane_xor<1>(v, lane));
  return v;
}
__device__ __forceinline__ float hsum(float v) {
  return hreduce(v, [](float a, float b) { return a + b; });
}

// Two consecutive blocks per iteration: block b in lanes 0-31, block b+1 in lanes 32-63.
//  * Discriminator (fm.c:91-160): the "previous valid sample" search of fm.c:128-144 runs over the 64-bit ballot, so a
//    sample of block b+1 finds its predecessor in block b exactly as the carried state of the reference would hand
//    it over; per-block quantities (amplitude statistics, threshold, squelch, deviation) are reductions over a half.
//    A squelched block b clears its half of the mask and hands zeros on, as fm.c:156-160 resets the state.
//  * De-emphasis (fm.c:162-171): the two overlap-save windows [hist | b] and [b | b+1] are real, so they go through
//    ONE complex 64-point transform as its real and imaginary parts, are separated by the Hermitian symmetry,
//    multiplied by the response and come back through ONE inverse transform whose real part is the audio of block
//    b and whose imaginary part is the audio of block b+1.
// Half the instructions per block of the one-block-per-iteration version, which is what matters with one wave per SIMD.
template <bool FLAT>
__device__ void fm_channel_pair(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  int const lane = threadIdx.x & 63;
  int const h = lane >> 5, n = lane & 31;
  float const gain = ch.fm_gain[c];
  float const noise_gain = ch.noise_gain[c];
  int const kbin = bitrev6(lane);
  int const herm_src = bitrev6((64 - kbin) & 63);
  // response on every bin: HA[k] up to Nyquist, conj(HA[64-k]) above (the Hermitian extension of the c2r transform)
  float2 HAf = make_float2(0.f, 0.f);
  if (!FLAT) {
    float2 const t = ch.aresp[(size_t)c * 33 + (kbin <= 32 ? kbin : 64 - kbin)];
    HAf = kbin <= 32 ? t : cconj(t);
  }
  bool const real_bin = kbin == 0 || kbin == 32;

  float2 wf[6], wi[6];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    int const half = 1 << s;
    float sn, cs;
    sincospif((float)(lane & (half - 1)) / (float)half, &sn, &cs);
    wf[s] = make_float2(cs, -sn);
    wi[s] = make_float2(cs, sn);
  }

  float2 state = ch.fm_state[c];
  float lastaudio = ch.lastaudio[c];
  int sq = ch.sq_count[c];
  float foffset = ch.foffset[c], pdev = ch.pdev[c];
  float n0 = ch.n0[c];
  float hist = h ? 0.f : ch.ahist[(size_t)c * 32 + lane];  // lanes 0-31: the block before b
  float ifp_v = 0.f, n0raw_v = 0.f;
  const float2 *in = pl.filt + (size_t)c * g.max_blocks * 32;

  float2 s_next = (h < nblocks) ? in[lane] : make_float2(0.f, 0.f);
  for (int b = 0; b < nblocks; b += 2) {
    bool const have1 = b + 1 < nblocks;
    float2 const S = s_next;
    if (b + 2 < nblocks) s_next = (b + 2 + h < nblocks) ? in[(size_t)(b + 2) * 32 + lane] : make_float2(0.f, 0.f);
    if ((b & 63) == 0) {  // per-block status inputs, lane i holds block b + i
      int const bb = b + lane;
      ifp_v = bb < nblocks ? pl.if_power[bb] : 0.f;
      n0raw_v = (compute_n0 && bb < nblocks) ? pl.n0raw[(size_t)c * g.max_blocks + bb] : 0.f;
    }

    // ---- amplitude statistics and squelch (fm.c:91-114), per half
    float const t = cnrm(S);
    float const sum_t = hsum(t), sum_a = hsum(sqrtf(t));
    float const bbp = sum_t / 64.f;                                  // / (2*olen), fm.c:99
    float const amp = (float)((double)sum_a / (M_SQRT2 * 32));       // fm.c:100
    float const variance = bbp - amp * amp;
    float snr = amp * amp / (2 * variance) - 1;
    snr = (0.0f > snr) ? 0.0f : snr;
    float const snr0 = rdlane(snr, 0), snr1 = rdlane(snr, 32);
    int nsq = sq + 1;
    nsq = nsq > 1000 ? 1000 : nsq;
    int const sq0 = (snr0 > 2) ? 0 : nsq;
    nsq = sq0 + 1;
    nsq = nsq > 1000 ? 1000 : nsq;
    int const sq1 = have1 ? ((snr1 > 2) ? 0 : nsq) : sq0;
    bool const open0 = sq0 < 2, open1 = have1 && sq1 < 2;
    bool const my_open = h ? open1 : open0;

    // ---- discriminator with hold (fm.c:117-144)
    float const thr = (float)(0.55 * 0.55 * amp * amp);
    unsigned long long const raw = __ballot(t > thr);
    unsigned long long const m_lo = open0 ? (raw & 0xffffffffull) : 0ull;
    unsigned long long const m_hi = open1 ? (raw >> 32) : 0ull;
    unsigned long long const mask = m_lo | (m_hi << 32);
    bool const valid = (mask >> lane) & 1ull;
    unsigned long long const below = mask & ((1ull << lane) - 1ull);
    unsigned long long const upto = mask & ((2ull << lane) - 1ull);
    int const pv = below ? 63 - __clzll((long long)below) : -1;
    int const lv = upto ? 63 - __clzll((long long)upto) : -1;
    // with no predecessor in the mask: block b falls back on the carried state, block b+1 on what block b leaves
    // behind when it has no valid sample either -- the carried state if b is open, zero if it is squelched
    float2 const state_fb = (h && !open0) ? make_float2(0.f, 0.f) : state;
    float const la_fb = (h && !open0) ? 0.f : lastaudio;
    float2 const sp = shfl2(S, pv >= 0 ? pv : 0);
    float2 const stc = pv >= 0 ? cconj(sp) : state_fb;
    float2 const pr = cmul(S, stc);
    float const y = valid ? atan2f(pr.y, pr.x) : 0.f;
    float const yl = __shfl(y, lv >= 0 ? lv : 0, 64);
    float const out = my_open ? (lv >= 0 ? yl : la_fb) : 0.f;

    // ---- frequency offset and peak deviation (fm.c:125-154), per half
    float const sum_y = hsum(out);
    float const vmax = hreduce((valid && n > 0) ? y : -INFINITY, [](float a, float b2) { return fmaxf(a, b2); });
    float const vmin = hreduce((valid && n > 0) ? y : INFINITY, [](float a, float b2) { return fminf(a, b2); });
    float const y_first = h ? rdlane(y, 32) : rdlane(y, 0);
    float const seed = ((mask >> (32 * h)) & 1ull) ? y_first : 0.f;
    float pdev_pos = fmaxf(seed, vmax), pdev_neg = fminf(seed, vmin);
    float const avg_f = sum_y / 32.f;
    pdev_pos -= avg_f;
    pdev_neg -= avg_f;
    float const mx = (pdev_pos > -pdev_neg) ? pdev_pos : -pdev_neg;
    float const fo_new = (float)(g.dsamprate * avg_f * (0.5 * M_1_PI));
    float const pd_new = (float)(g.dsamprate * mx * (0.5 * M_1_PI));
    float const fo0 = (sq0 < 1) ? rdlane(fo_new, 0) : foffset, pd0 = (sq0 < 1) ? rdlane(pd_new, 0) : pdev;
    float const fo1 = (have1 && sq1 < 1) ? rdlane(fo_new, 32) : fo0, pd1 = (have1 && sq1 < 1) ? rdlane(pd_new, 32) : pd0;

    // ---- carried state after the pair (fm.c:133-144, 156-160)
    {
      int const last0 = m_lo ? 63 - __clzll((long long)m_lo) : 0;
      int const last1 = m_hi ? 95 - __clzll((long long)m_hi) : 0;
      float2 const sl0 = cconj(make_float2(rdlane(S.x, last0), rdlane(S.y, last0)));
      float2 const sl1 = cconj(make_float2(rdlane(S.x, last1), rdlane(S.y, last1)));
      float const yl0 = rdlane(y, last0), yl1 = rdlane(y, last1);
      float2 st0 = open0 ? (m_lo ? sl0 : state) : make_float2(0.f, 0.f);
      float la0 = open0 ? (m_lo ? yl0 : lastaudio) : 0.f;
      if (have1) {
        st0 = open1 ? (m_hi ? sl1 : st0) : make_float2(0.f, 0.f);
        la0 = open1 ? (m_hi ? yl1 : la0) : 0.f;
      }
      state = st0;
      lastaudio = la0;
    }
    sq = sq1;
    foffset = fo1;
    pdev = pd1;

    // ---- de-emphasis (fm.c:162-171): both windows through one complex transform
    float const xo = lane_xor<32>(out, lane);  // lanes 0-31: block b+1, lanes 32-63: block b
    float a0 = out, a1 = xo;                   // flat: lanes 0-31 hold block b, and (after the exchange) block b+1
    if (!FLAT) {
      float2 z = make_float2(h ? xo : hist, out);  // real: [hist | b], imaginary: [b | b+1]
#pragma unroll
      for (int s = 5; s >= 0; s--) {  // forward, decimation in frequency: natural in, bit-reversed out
        float2 const r = xor2_pow(z, s, lane);
        z = ((lane >> s) & 1) ? cmul(csub(r, z), wf[s]) : cadd(z, r);
      }
      float2 const zm = cconj(shfl2(z, herm_src));  // conj(Z[64 - k])
      float2 const z1 = make_float2(0.5f * (z.x + zm.x), 0.5f * (z.y + zm.y));     // spectrum of the real part
      float2 const z2 = make_float2(0.5f * (z.y - zm.y), -0.5f * (z.x - zm.x));    // spectrum of the imaginary part
      float2 g1 = cmul(HAf, z1), g2 = cmul(HAf, z2);  // filter.c:206-208 on both
      if (real_bin) g1.y = g2.y = 0.f;                // the c2r transform ignores these imaginary parts
      z = make_float2(g1.x - g2.y, g1.y + g2.x);      // g1 + j g2
#pragma unroll
      for (int s = 0; s < 6; s++) {  // backward, decimation in time: bit-reversed in, natural out
        int const bit = (lane >> s) & 1;
        float2 const v = bit ? cmul(z, wi[s]) : z;
        float2 const r = xor2_pow(v, s, lane);
        z = bit ? csub(r, v) : cadd(v, r);
      }
      // lanes 32-63 hold the kept halves: real part block b, imaginary part block b+1 (fm.c:169-170)
      a0 = z.x * gain;
      a1 = z.y * gain;
      if (h) {
        pl.audio[((size_t)c * g.max_blocks + b) * 64 + n] = a0;
        if (have1) pl.audio[((size_t)c * g.max_blocks + b + 1) * 64 + n] = a1;
      }
    } else {
      if (!h) {
        pl.audio[((size_t)c * g.max_blocks + b) * 64 + n] = a0;
        if (have1) pl.audio[((size_t)c * g.max_blocks + b + 1) * 64 + n] = a1;
      }
    }
    hist = have1 ? xo : out;  // lanes 0-31: the last block processed becomes the history (filter.c:168)

    // ---- status records: lane 0 writes block b, lane 32 block b+1
    float const ifp = h ? rdlane(ifp_v, (b + 1) & 63) : rdlane(ifp_v, b & 63);
    float const fresh0 = rdlane(n0raw_v, b & 63), fresh1 = rdlane(n0raw_v, (b + 1) & 63);
    float n0a = n0, n0b = n0;
    if (compute_n0) {  // fm.c:79-82, block after block
      n0a = isnan(n0) ? fresh0 : (float)((double)n0 + .01 * (double)(fresh0 - n0));
      n0b = have1 ? (isnan(n0a) ? fresh1 : (float)((double)n0a + .01 * (double)(fresh1 - n0a))) : n0a;
      n0 = n0b;
    }
    if (n == 0 && (h == 0 || have1)) {
      kq_chan_status rec;
      rec.if_power = ifp;
      rec.noise_gain = noise_gain;
      rec.plfreq = NAN;  // N/D = 64: the PL slave would have 2 points (fm.c:203), measurement off
      rec.cphase = 0;
      rec.pll_lock = 0;
      rec.lock_count = 0;
      rec.n0 = compute_n0 ? (h ? n0b : n0a) : NAN;
      rec.bb_power = bbp;
      rec.snr = snr;
      rec.foffset = h ? fo1 : fo0;
      rec.pdeviation = h ? pd1 : pd0;
      rec.agc_gain = 0;
      rec.squelch_count = h ? sq1 : sq0;
      rec.hangcount = 0;
      rec.blanked = my_open ? 32 - __popcll((mask >> (32 * h)) & 0xffffffffull) : 0;
      rec.nout = 32;
      pl.status[(size_t)c * g.max_blocks + b + h] = rec;
    }
  }
  if (!h) ch.ahist[(size_t)c * 32 + lane] = hist;
  if (lane == 0) {
    ch.fm_state[c] = state;
    ch.lastaudio[c] = lastaudio;
    ch.sq_count[c] = sq;
    ch.foffset[c] = foffset;
    ch.pdev[c] = pdev;
    ch.n0[c] = n0;
  }
}

// The same demodulator on FOUR waves per channel (de-emphasised channels), a pipeline over the block pairs: in one
// iteration wave 0 runs statistics and squelch of pair i, wave 1 the discriminator of pair i - 1, wave 2 deviation and
// the forward half of the de-emphasis filter of pair i - 2, wave 3 its inverse half, the audio store and the status
// records of pair i - 3; each hand-over goes through LDS behind the iteration's one barrier.  A wave issues at most one
// vector instruction per 8 cycles (tools/valu_rate.hip), and with one channel per SIMD that rate is all this kernel gets:
// splitting the ~800 dependent instructions of a pair four ways is worth what the longest quarter takes.  Every value
// is computed by the same expressions as in fm_channel_pair.
__device__ void fm_channel_four_waves(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  // [parity][...]; scalars per half: see the stages
  __shared__ float2 s_S[2][64];                 // 0 -> 1: the pair's samples
  __shared__ float s_a[2][2][4];                // 0 -> 1: amplitude, bb_power, snr, squelch_count
  __shared__ float s_out[2][64], s_y[2][64];    // 1 -> 2: the discriminator's output after / before the hold (fm.c:128-144)
  __shared__ float s_b[2][2][5];                // 1 -> 2: bb_power, snr, mask of valid samples, squelch_count, blanked
  __shared__ float2 s_z[2][64];                 // 2 -> 3: filtered spectra of both windows, g1 + j g2
  __shared__ float s_c[2][2][6];                // 2 -> 3: bb_power, snr, foffset, pdeviation, squelch_count, blanked
  int const lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  int const h = lane >> 5, n = lane & 31;
  int const npairs = (nblocks + 1) / 2, niter = npairs + 3;

  if (role == 0) {  // ---------------- statistics and squelch (fm.c:91-114), per half
    int sq = ch.sq_count[c];
    const float2 *in = pl.filt + (size_t)c * g.max_blocks * 32;
    float2 s_next = (h < nblocks) ? in[lane] : make_float2(0.f, 0.f);
    for (int it = 0; it < niter; it++) {
      if (it < npairs) {
        int const b = 2 * it, par = it & 1;
        bool const have1 = b + 1 < nblocks;
        float2 const S = s_next;
        if (b + 2 < nblocks) s_next = (b + 2 + h < nblocks) ? in[(size_t)(b + 2) * 32 + lane] : make_float2(0.f, 0.f);
        float const t = cnrm(S);
        float const sum_t = hsum(t), sum_a = hsum(sqrtf(t));
        float const bbp = sum_t / 64.f;                                  // / (2*olen), fm.c:99
        float const amp = (float)((double)sum_a / (M_SQRT2 * 32));       // fm.c:100
        float const variance = bbp - amp * amp;
        float snr = amp * amp / (2 * variance) - 1;
        snr = (0.0f > snr) ? 0.0f : snr;
        float const snr0 = rdlane(snr, 0), snr1 = rdlane(snr, 32);
        int nsq = sq + 1;
        nsq = nsq > 1000 ? 1000 : nsq;
        int const sq0 = (snr0 > 2) ? 0 : nsq;
        nsq = sq0 + 1;
        nsq = nsq > 1000 ? 1000 : nsq;
        int const sq1 = have1 ? ((snr1 > 2) ? 0 : nsq) : sq0;
        sq = sq1;
        s_S[par][lane] = S;
        if (n == 0) {
          float *st = s_a[par][h];
          st[0] = amp;
          st[1] = bbp;
          st[2] = snr;
          st[3] = __int_as_float(h ? sq1 : sq0);
        }
      }
      __syncthreads();
    }
    if (lane == 0) ch.sq_count[c] = sq;
    return;
  }

  if (role == 1) {  // ---------------- discriminator with hold (fm.c:117-144)
    float2 state = ch.fm_state[c];
    float lastaudio = ch.lastaudio[c];
    for (int it = 0; it < niter; it++) {
      if (it >= 1 && it - 1 < npairs) {
        int const b = 2 * (it - 1), par = (it - 1) & 1;
        bool const have1 = b + 1 < nblocks;
        float2 const S = s_S[par][lane];
        float const amp = s_a[par][h][0];
        int const sq0 = __float_as_int(s_a[par][0][3]), sq1 = __float_as_int(s_a[par][1][3]);
        bool const open0 = sq0 < 2, open1 = have1 && sq1 < 2;
        bool const my_open = h ? open1 : open0;
        float const t = cnrm(S);
        float const thr = (float)(0.55 * 0.55 * amp * amp);
        unsigned long long const raw = __ballot(t > thr);
        unsigned long long const m_lo = open0 ? (raw & 0xffffffffull) : 0ull;
        unsigned long long const m_hi = open1 ? (raw >> 32) : 0ull;
        unsigned long long const mask = m_lo | (m_hi << 32);
        bool const valid = (mask >> lane) & 1ull;
        unsigned long long const below = mask & ((1ull << lane) - 1ull);
        unsigned long long const upto = mask & ((2ull << lane) - 1ull);
        int const pv = below ? 63 - __clzll((long long)below) : -1;
        int const lv = upto ? 63 - __clzll((long long)upto) : -1;
        // with no predecessor in the mask: block b falls back on the carried state, block b+1 on what block b leaves
        // behind when it has no valid sample either -- the carried state if b is open, zero if it is squelched
        float2 const state_fb = (h && !open0) ? make_float2(0.f, 0.f) : state;
        float const la_fb = (h && !open0) ? 0.f : lastaudio;
        float2 const sp = shfl2(S, pv >= 0 ? pv : 0);
        float2 const stc = pv >= 0 ? cconj(sp) : state_fb;
        float2 const pr = cmul(S, stc);
        float const y = valid ? atan2f(pr.y, pr.x) : 0.f;
        float const yl = __shfl(y, lv >= 0 ? lv : 0, 64);
        float const out = my_open ? (lv >= 0 ? yl : la_fb) : 0.f;
        // carried state after the pair (fm.c:133-144, 156-160)
        {
          int const last0 = m_lo ? 63 - __clzll((long long)m_lo) : 0;
          int const last1 = m_hi ? 95 - __clzll((long long)m_hi) : 0;
          float2 const sl0 = cconj(make_float2(rdlane(S.x, last0), rdlane(S.y, last0)));
          float2 const sl1 = cconj(make_float2(rdlane(S.x, last1), rdlane(S.y, last1)));
          float const yl0 = rdlane(y, last0), yl1 = rdlane(y, last1);
          float2 st0 = open0 ? (m_lo ? sl0 : state) : make_float2(0.f, 0.f);
          float la0 = open0 ? (m_lo ? yl0 : lastaudio) : 0.f;
          if (have1) {
            st0 = open1 ? (m_hi ? sl1 : st0) : make_float2(0.f, 0.f);
            la0 = open1 ? (m_hi ? yl1 : la0) : 0.f;
          }
          state = st0;
          lastaudio = la0;
        }
        s_out[par][lane] = out;
        s_y[par][lane] = y;
        if (n == 0) {
          float *st = s_b[par][h];
          st[0] = s_a[par][h][1];
          st[1] = s_a[par][h][2];
          st[2] = __int_as_float((int)(unsigned)(mask >> (32 * h)));   // this half's mask of valid samples
          st[3] = __int_as_float(h ? sq1 : sq0);
          st[4] = __int_as_float(my_open ? 32 - __popcll((mask >> (32 * h)) & 0xffffffffull) : 0);
        }
      }
      __syncthreads();
    }
    if (lane == 0) {
      ch.fm_state[c] = state;
      ch.lastaudio[c] = lastaudio;
    }
    return;
  }

  int const kbin = bitrev6(lane);
  if (role == 2) {  // ---------------- frequency offset and deviation (fm.c:125-154); de-emphasis, forward half (fm.c:162-171)
    int const herm_src = bitrev6((64 - kbin) & 63);
    float2 HAf;
    {
      float2 const t = ch.aresp[(size_t)c * 33 + (kbin <= 32 ? kbin : 64 - kbin)];
      HAf = kbin <= 32 ? t : cconj(t);
    }
    bool const real_bin = kbin == 0 || kbin == 32;
    float2 wf[6];
#pragma unroll
    for (int s = 0; s < 6; s++) {
      int const half = 1 << s;
      float sn, cs;
      sincospif((float)(lane & (half - 1)) / (float)half, &sn, &cs);
      wf[s] = make_float2(cs, -sn);
    }
    float foffset = ch.foffset[c], pdev = ch.pdev[c];
    float hist = h ? 0.f : ch.ahist[(size_t)c * 32 + lane];  // lanes 0-31: the block before b
    for (int it = 0; it < niter; it++) {
      if (it >= 2 && it - 2 < npairs) {
        int const b = 2 * (it - 2), par = (it - 2) & 1;
        bool const have1 = b + 1 < nblocks;
        float const out = s_out[par][lane], y = s_y[par][lane];
        unsigned const mask_h = (unsigned)__float_as_int(s_b[par][h][2]);
        int const sq0 = __float_as_int(s_b[par][0][3]), sq1 = __float_as_int(s_b[par][1][3]);
        bool const valid = (mask_h >> n) & 1u;
        float const sum_y = hsum(out);
        float const vmax = hreduce((valid && n > 0) ? y : -INFINITY, [](float a, float b2) { return fmaxf(a, b2); });
        float const vmin = hreduce((valid && n > 0) ? y : INFINITY, [](float a, float b2) { return fminf(a, b2); });
        float const y_first = h ? rdlane(y, 32) : rdlane(y, 0);
        float const seed = (mask_h & 1u) ? y_first : 0.f;
        float pdev_pos = fmaxf(seed, vmax), pdev_neg = fminf(seed, vmin);
        float const avg_f = sum_y / 32.f;
        pdev_pos -= avg_f;
        pdev_neg -= avg_f;
        float const mx = (pdev_pos > -pdev_neg) ? pdev_pos : -pdev_neg;
        float const fo_new = (float)(g.dsamprate * avg_f * (0.5 * M_1_PI));
        float const pd_new = (float)(g.dsamprate * mx * (0.5 * M_1_PI));
        float const fo0 = (sq0 < 1) ? rdlane(fo_new, 0) : foffset, pd0 = (sq0 < 1) ? rdlane(pd_new, 0) : pdev;
        float const fo1 = (have1 && sq1 < 1) ? rdlane(fo_new, 32) : fo0, pd1 = (have1 && sq1 < 1) ? rdlane(pd_new, 32) : pd0;
        foffset = fo1;
        pdev = pd1;

        float const xo = lane_xor<32>(out, lane);    // lanes 0-31: block b+1, lanes 32-63: block b
        float2 z = make_float2(h ? xo : hist, out);  // real: [hist | b], imaginary: [b | b+1]
#pragma unroll
        for (int s = 5; s >= 0; s--) {  // forward, decimation in frequency: natural in, bit-reversed out
          float2 const r = xor2_pow(z, s, lane);
          z = ((lane >> s) & 1) ? cmul(csub(r, z), wf[s]) : cadd(z, r);
        }
        float2 const zm = cconj(shfl2(z, herm_src));  // conj(Z[64 - k])
        float2 const z1 = make_float2(0.5f * (z.x + zm.x), 0.5f * (z.y + zm.y));     // spectrum of the real part
        float2 const z2 = make_float2(0.5f * (z.y - zm.y), -0.5f * (z.x - zm.x));    // spectrum of the imaginary part
        float2 g1 = cmul(HAf, z1), g2 = cmul(HAf, z2);  // filter.c:206-208 on both
        if (real_bin) g1.y = g2.y = 0.f;                // the c2r transform ignores these imaginary parts
        s_z[par][lane] = make_float2(g1.x - g2.y, g1.y + g2.x);  // g1 + j g2
        hist = have1 ? xo : out;  // lanes 0-31: the last block processed becomes the history (filter.c:168)
        if (n == 0) {
          float *st = s_c[par][h];
          st[0] = s_b[par][h][0];
          st[1] = s_b[par][h][1];
          st[2] = h ? fo1 : fo0;
          st[3] = h ? pd1 : pd0;
          st[4] = s_b[par][h][3];
          st[5] = s_b[par][h][4];
        }
      }
      __syncthreads();
    }
    if (!h) ch.ahist[(size_t)c * 32 + lane] = hist;
    if (lane == 0) {
      ch.foffset[c] = foffset;
      ch.pdev[c] = pdev;
    }
    return;
  }

  // ---------------- wave 3: de-emphasis, inverse half; audio; status records
  float const gain = ch.fm_gain[c];
  float const noise_gain = ch.noise_gain[c];
  float2 wi[6];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    int const half = 1 << s;
    float sn, cs;
    sincospif((float)(lane & (half - 1)) / (float)half, &sn, &cs);
    wi[s] = make_float2(cs, sn);
  }
  float n0 = ch.n0[c];
  float ifp_v = 0.f, n0raw_v = 0.f;
  for (int it = 0; it < niter; it++) {
    if (it >= 3) {
      int const b = 2 * (it - 3), par = (it - 3) & 1;
      bool const have1 = b + 1 < nblocks;
      if ((b & 63) == 0) {  // per-block status inputs, lane i holds block b + i
        int const bb = b + lane;
        ifp_v = bb < nblocks ? pl.if_power[bb] : 0.f;
        n0raw_v = (compute_n0 && bb < nblocks) ? pl.n0raw[(size_t)c * g.max_blocks + bb] : 0.f;
      }
      float2 z = s_z[par][lane];
#pragma unroll
      for (int s = 0; s < 6; s++) {  // backward, decimation in time: bit-reversed in, natural out
        int const bit = (lane >> s) & 1;
        float2 const v = bit ? cmul(z, wi[s]) : z;
        float2 const r = xor2_pow(v, s, lane);
        z = bit ? csub(r, v) : cadd(v, r);
      }
      // lanes 32-63 hold the kept halves: real part block b, imaginary part block b+1 (fm.c:169-170)
      float const a0 = z.x * gain, a1 = z.y * gain;
      if (h) {
        pl.audio[((size_t)c * g.max_blocks + b) * 64 + n] = a0;
        if (have1) pl.audio[((size_t)c * g.max_blocks + b + 1) * 64 + n] = a1;
      }
      // status records: lane 0 writes block b, lane 32 block b+1
      float const ifp = h ? rdlane(ifp_v, (b + 1) & 63) : rdlane(ifp_v, b & 63);
      float const fresh0 = rdlane(n0raw_v, b & 63), fresh1 = rdlane(n0raw_v, (b + 1) & 63);
      float n0a = n0, n0b = n0;
      if (compute_n0) {  // fm.c:79-82, block after block
        n0a = isnan(n0) ? fresh0 : (float)((double)n0 + .01 * (double)(fresh0 - n0));
        n0b = have1 ? (isnan(n0a) ? fresh1 : (float)((double)n0a + .01 * (double)(fresh1 - n0a))) : n0a;
        n0 = n0b;
      }
      if (n == 0 && (h == 0 || have1)) {
        const float *st = s_c[par][h];
        kq_chan_status rec;
        rec.if_power = ifp;
        rec.noise_gain = noise_gain;
        rec.plfreq = NAN;  // N/D = 64: the PL slave would have 2 points (fm.c:203), measurement off
        rec.cphase = 0;
        rec.pll_lock = 0;
        rec.lock_count = 0;
        rec.n0 = compute_n0 ? (h ? n0b : n0a) : NAN;
        rec.bb_power = st[0];
        rec.snr = st[1];
        rec.foffset = st[2];
        rec.pdeviation = st[3];
        rec.agc_gain = 0;
        rec.squelch_count = __float_as_int(st[4]);
        rec.hangcount = 0;
        rec.blanked = __float_as_int(st[5]);
        rec.nout = 32;
        pl.status[(size_t)c * g.max_blocks + b + h] = rec;
      }
    }
    __syncthreads();
  }
  if (lane == 0) ch.n0[c] = n0;
}

__device__ void fm_channel(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  if (ch.flags[c] & FLAG_FLAT)
    fm_channel_pair<true>(g, ch, pl, c, nblocks, compute_n0);
  else
    fm_channel_pair<false>(g, ch, pl, c, nblocks, compute_n0);
}

// AM / linear: one wave per channel, one lane per sample, two consecutive blocks per iteration (lanes 0-31 and
// 32-63).  Square roots, the attack gains headroom/level (IEEE divisions), the shift NCO and all loads/stores are
// lane-parallel and coalesced; only the AGC recurrence itself (am.c:64-74, linear.c:269-279: one multiply, one
// compare, two selects per sample) runs serially, wave-uniform, reading its per-sample inputs with v_readlane.
// The arithmetic per sample is exactly the reference's, in the reference's order.
template <bool LINEAR, int OLEN, bool SPLIT>
__device__ void agc_channel(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  int const lane = threadIdx.x & 63;
  constexpr int BPI = 64 / OLEN;  // blocks per iteration
  int const half = lane / OLEN, n = lane % OLEN;
  float const headroom = ch.headroom[c], recovery = ch.recovery[c];
  int const hangmax = ch.hangmax[c];
  bool const stereo = LINEAR && (ch.flags[c] & FLAG_STEREO) != 0;
  double const sh_ph = LINEAR ? ch.sh_phase[c] : 0.0, sh_f = LINEAR ? ch.sh_freq[c] : 0.0;
  float gain = ch.gain[c], dc = LINEAR ? 0.f : ch.dc[c];
  int hang = ch.hang[c];
  float n0 = ch.n0[c];
  const float2 *in = pl.filt + (size_t)c * g.max_blocks * OLEN;
  float2 s_next = (half < nblocks) ? in[lane] : make_float2(0.f, 0.f);
  // SPLIT: two waves per channel.  Wave 0 runs the front of iteration i -- envelope, power sums, the carrier filter's
  // recurrence, the attack gains -- while wave 1 runs the AGC recurrence, the audio and the status of iteration i - 1;
  // the six per-sample values cross in LDS behind one barrier per iteration.  A wave issues one instruction per 8
  // cycles (tools/valu_rate.hip) and the two recurrences are what an AM channel's time consists of.
  __shared__ float s_hand[SPLIT ? 2 : 1][7][64];
  int const role = SPLIT ? (int)(threadIdx.x >> 6) : 0;
  int const niter = (nblocks + BPI - 1) / BPI;
  for (int it = 0; it < niter + (SPLIT ? 1 : 0); it++) {
    float2 S = make_float2(0.f, 0.f);
    float level = 0.f, env = 0.f, sig = 0.f, noi = 0.f, inv = 0.f;
    if (role == 0 && it < niter) {
      int const b0 = it * BPI;
      int const blk = b0 + half;
      int const nsamp = (b0 + BPI <= nblocks) ? 64 : OLEN;  // wave-uniform
      S = s_next;
      if (b0 + BPI < nblocks) s_next = (blk + BPI < nblocks) ? in[(size_t)(b0 + BPI) * OLEN + lane] : make_float2(0.f, 0.f);
      float const rp = S.x * S.x, ip = S.y * S.y;
      level = sqrtf(LINEAR ? rp + ip : S.x * S.x + S.y * S.y);  // amplitude (linear.c:260) / envelope (am.c:58)
      env = level;
      // per-block power sums over each OLEN-lane group
      sig = LINEAR ? rp : rp + ip;
      noi = LINEAR ? ip : 0.f;
      if (!LINEAR) sig = S.x * S.x + S.y * S.y;
#pragma unroll
      for (int o = OLEN / 2; o > 0; o >>= 1) {
        sig += __shfl_xor(sig, o, 64);
        noi += __shfl_xor(noi, o, 64);
      }
      if (!LINEAR) {
        // carrier tracking (am.c:62), serial; lane i keeps the value after sample i
        float dc_mine = dc;
#pragma unroll 32
        for (int i = 0; i < nsamp; i++) {
          float const e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, env), i));
          dc += 0.0001f * (e - dc);
          dc_mine = (lane == i) ? dc : dc_mine;
        }
        level = dc_mine;
      }
      inv = headroom / level;
      if (SPLIT) {
        float(*hb)[64] = s_hand[it & 1];
        hb[0][lane] = S.x;
        hb[1][lane] = S.y;
        hb[2][lane] = level;
        hb[3][lane] = env;
        hb[4][lane] = sig;
        hb[5][lane] = noi;
        hb[6][lane] = inv;
      }
    }
    if (SPLIT) {
      if (role == 1 && it >= 1) {
        float(*hb)[64] = s_hand[(it - 1) & 1];
        S = make_float2(hb[0][lane], hb[1][lane]);
        level = hb[2][lane];
        env = hb[3][lane];
        sig = hb[4][lane];
        noi = hb[5][lane];
        inv = hb[6][lane];
      }
      __syncthreads();
      if (role == 0 || it == 0) continue;
    }
    int const b0 = (SPLIT ? it - 1 : it) * BPI;
    int const blk = b0 + half;
    bool const active = blk < nblocks;
    int const nsamp = (b0 + BPI <= nblocks) ? 64 : OLEN;  // wave-uniform
    float g_mine = gain;
    float gain_end[2] = {gain, gain};
    int hang_end[2] = {hang, hang};
    // While the hang counter outlasts the group and no sample attacks, the gain does not move (am.c:67-70,
    // linear.c:273-275): the 64 comparisons are made at once and the counter drops by the group length.  This is
    // the usual state of an SSB channel (hang time 1.1 s).
    bool const held = !isnan(gain) && hang >= nsamp &&
                      __ballot(lane < nsamp && (LINEAR ? level * gain > headroom : gain * level > headroom)) == 0ull;
    if (held) {
      hang_end[0] = hang - OLEN;
      hang -= nsamp;
    }
    // Coasting: the counter runs out inside (or before) the group and still no sample attacks, so the gain only holds
    // and then recovers, gain *= recovery per sample once the counter is at zero (am.c:71-73, linear.c:276-278).  The
    // products are formed in sequence, as the reference forms them; every lane keeps the value after its own sample,
    // takes the one before it from its neighbour, and one ballot checks that indeed nothing attacks.  Tried only for
    // channels with a hang time (for the others a group without an attack is the exception).
    bool coast = false;
    if (!held && hangmax != 0 && hang < nsamp && !isnan(gain)) {
      float g = gain, after = gain;
#pragma unroll 32
      for (int i = 0; i < nsamp; i++) {
        g = (i >= hang) ? g * recovery : g;
        after = (lane == i) ? g : after;
      }
      float before = __shfl_up(after, 1, 64);
      before = (lane == 0) ? gain : before;
      coast = __ballot(lane < nsamp && (LINEAR ? level * before > headroom : before * level > headroom)) == 0ull;
      if (coast) {
        g_mine = after;
        gain_end[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, after), OLEN - 1));
        hang_end[0] = hang > OLEN ? hang - OLEN : 0;
        gain = g;
        hang = 0;
      }
    }
    // No hang time (the AM entry of modes.txt) and no NaN in sight: the counter stays at zero and the NaN test of
    // the gain cannot fire (the gain is either an attack value of this group or a product with the recovery
    // factor), which leaves multiply / compare / select per sample.
    bool const plain = !held && hangmax == 0 && hang == 0 && !isnan(gain) && __ballot(lane < nsamp && isnan(inv)) == 0ull;
    if (plain) {
#pragma unroll 32
      for (int i = 0; i < nsamp; i++) {
        float const lv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, level), i));
        float const iv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), i));
        bool const attack = LINEAR ? lv * gain > headroom : gain * lv > headroom;
        gain = attack ? iv : gain * recovery;
        g_mine = (lane == i) ? gain : g_mine;
        if (i == OLEN - 1) gain_end[0] = gain;
      }
    }
#pragma unroll 32
    for (int i = 0; i < ((held || plain || coast) ? 0 : nsamp); i++) {
      float const lv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, level), i));
      float const iv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), i));
      bool const nan_gain = isnan(gain);
      bool const attack = nan_gain || (LINEAR ? lv * gain > headroom : gain * lv > headroom);
      float const rec = (hang != 0) ? gain : gain * recovery;
      int const hdec = (hang != 0) ? hang - 1 : 0;
      hang = (attack && !nan_gain) ? hangmax : (attack ? hang : hdec);
      gain = attack ? iv : rec;
      g_mine = (lane == i) ? gain : g_mine;
      if (i == OLEN - 1) {
        gain_end[0] = gain;
        hang_end[0] = hang;
      }
    }
    gain_end[1] = gain;
    hang_end[1] = hang;
    float *aud = pl.audio + ((size_t)c * g.max_blocks + blk) * (2 * OLEN);
    if (active) {
      if (LINEAR) {
        float2 sv = make_float2(S.x * g_mine, S.y * g_mine);
        if (sh_f != 0.0) {  // linear.c:283-289
          double turns = sh_ph + sh_f * ((double)blk * OLEN + n);
          turns -= rint(turns);
          float sn, cs;
          sincospif(2.f * (float)turns, &sn, &cs);
          sv = cmul(sv, make_float2(cs, sn));
        }
        if (stereo)
          reinterpret_cast<float2 *>(aud)[n] = sv;
        else
          aud[n] = sv.x;
      } else {
        aud[n] = (env - level) * g_mine;
      }
    }
    // status: lane 0 of each half, in block order so the smoothed n0 follows the sequence
#pragma unroll
    for (int h = 0; h < BPI; h++) {
      if (b0 + h < nblocks) {
        float const sg = __shfl(sig, h * OLEN, 64), nz = __shfl(noi, h * OLEN, 64);
        if (lane == 0) {
          kq_chan_status st;
          put_status(st, g, ch, pl, c, b0 + h, compute_n0, .001, n0);
          st.bb_power = (sg + nz) / (2.f * OLEN);
          st.snr = LINEAR ? NAN : 0.f;
          st.foffset = 0;
          st.pdeviation = 0;
          st.agc_gain = gain_end[BPI == 1 ? 1 : h];
          st.squelch_count = 0;
          st.hangcount = hang_end[BPI == 1 ? 1 : h];
          st.blanked = 0;
          st.nout = stereo ? 2 * OLEN : OLEN;
          pl.status[(size_t)c * g.max_blocks + b0 + h] = st;
        }
      }
    }
  }
  if (lane == 0) {
    if (!SPLIT || role == 1) {
      ch.gain[c] = gain;
      ch.hang[c] = hang;
      ch.n0[c] = n0;
    }
    if (!LINEAR && role == 0) ch.dc[c] = dc;
  }
}

// The same for any block length (the generic geometries: N/D = 128 ... 16384, mixed radix included): one wave per
// channel, a block in groups of 64 samples (the last one may be short), gain / hang counter / carrier estimate carried
// from group to group in wave-uniform registers.  Where the 64-sample form keeps a block's end state per half, a block
// here ends with a group, so the state after the group is the block's.  k_demod_am / k_demod_linear (one LANE per
// channel, kq_kernels.hip) did this work before round 6: 1024 channels were 16 waves on a device with 1024 SIMDs.
// FULL: the group has 64 samples and the serial loops unroll; the short last group of a block runs them as loops.

// carrier tracking (am.c:62), serial; lane i keeps the value after sample i
template <bool FULL>
__device__ __forceinline__ float carrier_group(float env, int nsamp, int lane, float &dc) {
  float dc_mine = dc;
  int const ns = FULL ? 64 : nsamp;
  constexpr int UNR = FULL ? 32 : 1;  // a runtime trip count and v_readlane (convergent) do not unroll
#pragma unroll UNR
  for (int i = 0; i < ns; i++) {
    float const e = rdlane(env, i);
    dc += 0.0001f * (e - dc);
    dc_mine = (lane == i) ? dc : dc_mine;
  }
  return dc_mine;
}

// the AGC recurrence over one group in agc_channel's four forms (held / coasting / plain / general); returns the gain
// each lane's sample is scaled by
template <bool LINEAR, bool FULL>
__device__ __forceinline__ float agc_group(float level, float inv, int nsamp, int lane, float headroom, float recovery,
                                           int hangmax, float &gain, int &hang) {
  int const ns = FULL ? 64 : nsamp;
  constexpr int UNR = FULL ? 32 : 1;
  bool const mine = lane < ns;
  float g_mine = gain;
  bool const held = !isnan(gain) && hang >= ns &&
                    __ballot(mine && (LINEAR ? level * gain > headroom : gain * level > headroom)) == 0ull;
  if (held) {
    hang -= ns;
    return g_mine;
  }
  if (hangmax != 0 && hang < ns && !isnan(gain)) {
    float gg = gain, after = gain;
#pragma unroll UNR
    for (int i = 0; i < ns; i++) {
      gg = (i >= hang) ? gg * recovery : gg;
      after = (lane == i) ? gg : after;
    }
    float before = __shfl_up(after, 1, 64);
    before = (lane == 0) ? gain : before;
    if (__ballot(mine && (LINEAR ? level * before > headroom : before * level > headroom)) == 0ull) {
      gain = gg;
      hang = 0;
      return after;
    }
  }
  if (hangmax == 0 && hang == 0 && !isnan(gain) && __ballot(mine && isnan(inv)) == 0ull) {
#pragma unroll UNR
    for (int i = 0; i < ns; i++) {
      float const lv = rdlane(level, i), iv = rdlane(inv, i);
      bool const attack = LINEAR ? lv * gain > headroom : gain * lv > headroom;
      gain = attack ? iv : gain * recovery;
      g_mine = (lane == i) ? gain : g_mine;
    }
    return g_mine;
  }
#pragma unroll UNR
  for (int i = 0; i < ns; i++) {
    float const lv = rdlane(level, i), iv = rdlane(inv, i);
    bool const nan_gain = isnan(gain);
    bool const attack = nan_gain || (LINEAR ? lv * gain > headroom : gain * lv > headroom);
    float const rec = (hang != 0) ? gain : gain * recovery;
    int const hdec = (hang != 0) ? hang - 1 : 0;
    hang = (attack && !nan_gain) ? hangmax : (attack ? hang : hdec);
    gain = attack ? iv : rec;
    g_mine = (lane == i) ? gain : g_mine;
  }
  return g_mine;
}

template <bool LINEAR>
__device__ void agc_channel_any(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  int const lane = threadIdx.x & 63;
  int const olen = g.olen;
  float const headroom = ch.headroom[c], recovery = ch.recovery[c];
  int const hangmax = ch.hangmax[c];
  bool const stereo = LINEAR && (ch.flags[c] & FLAG_STEREO) != 0;
  double const sh_ph = LINEAR ? ch.sh_phase[c] : 0.0, sh_f = LINEAR ? ch.sh_freq[c] : 0.0;
  float gain = ch.gain[c], dc = LINEAR ? 0.f : ch.dc[c];
  int hang = ch.hang[c];
  float n0 = ch.n0[c];
  int const ngroups = (olen + 63) / 64;
  const float2 *in = pl.filt + (size_t)c * g.max_blocks * olen;  // the call's blocks are contiguous
  float2 s_next = (lane < olen && nblocks > 0) ? in[lane] : make_float2(0.f, 0.f);
  for (int b = 0; b < nblocks; b++) {
    float *aud = pl.audio + ((size_t)c * g.max_blocks + b) * (2 * (size_t)olen);
    float sig = 0.f, noi = 0.f;
    for (int gr = 0; gr < ngroups; gr++) {
      int const off = gr * 64;
      int const nsamp = olen - off < 64 ? olen - off : 64;  // wave-uniform
      float2 const S = s_next;
      {  // the next group's samples travel while this one's recurrence runs
        int nb = b, noff = off + 64;
        if (noff >= olen) {
          nb++;
          noff = 0;
        }
        s_next = (nb < nblocks && noff + lane < olen) ? in[(size_t)nb * olen + noff + lane] : make_float2(0.f, 0.f);
      }
      float const rp = S.x * S.x, ip = S.y * S.y;
      float const env = sqrtf(LINEAR ? rp + ip : S.x * S.x + S.y * S.y);  // amplitude (linear.c:260) / envelope (am.c:58)
      float level = env;
      sig += LINEAR ? rp : S.x * S.x + S.y * S.y;  // lanes past the block's end hold zeros
      noi += LINEAR ? ip : 0.f;
      if (!LINEAR) level = nsamp == 64 ? carrier_group<true>(env, 64, lane, dc) : carrier_group<false>(env, nsamp, lane, dc);
      float const inv = headroom / level;
      float const g_mine = nsamp == 64 ? agc_group<LINEAR, true>(level, inv, 64, lane, headroom, recovery, hangmax, gain, hang)
                                       : agc_group<LINEAR, false>(level, inv, nsamp, lane, headroom, recovery, hangmax, gain, hang);
      if (lane < nsamp) {
        int const n = off + lane;
        if (LINEAR) {
          float2 sv = make_float2(S.x * g_mine, S.y * g_mine);
          if (sh_f != 0.0) {  // linear.c:283-289
            double turns = sh_ph + sh_f * ((double)b * olen + n);
            turns -= rint(turns);
            float sn, cs;
            sincospif(2.f * (float)turns, &sn, &cs);
            sv = cmul(sv, make_float2(cs, sn));
          }
          if (stereo)
            reinterpret_cast<float2 *>(aud)[n] = sv;
          else
            aud[n] = sv.x;
        } else {
          aud[n] = (env - level) * g_mine;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sig += __shfl_xor(sig, o, 64);
      noi += __shfl_xor(noi, o, 64);
    }
    if (lane == 0) {
      kq_chan_status st;
      put_status(st, g, ch, pl, c, b, compute_n0, .001, n0);
      st.bb_power = (sig + noi) / (2.f * olen);  // am.c:78, linear.c:302
      st.snr = LINEAR ? NAN : 0.f;               // linear.c:309
      st.foffset = 0;
      st.pdeviation = 0;
      st.agc_gain = gain;
      st.squelch_count = 0;
      st.hangcount = hang;
      st.blanked = 0;
      st.nout = stereo ? 2 * olen : olen;
      pl.status[(size_t)c * g.max_blocks + b] = st;
    }
  }
  if (lane == 0) {
    ch.gain[c] = gain;
    ch.hang[c] = hang;
    ch.n0[c] = n0;
    if (!LINEAR) ch.dc[c] = dc;
  }
}

}  // namespace

// grid = n_am + n_lin workgroups of one wave, one channel each, any block length
__global__ void __launch_bounds__(64) k_demod_agc_any(Geom g, ChanDev ch, Planes pl, const int *__restrict__ list_am, int n_am,
                                                      const int *__restrict__ list_lin, int n_lin, int nblocks, int compute_n0) {
  int const wg = blockIdx.x;
  if (wg < n_am)
    agc_channel_any<false>(g, ch, pl, list_am[wg], nblocks, compute_n0);
  else if (wg - n_am < n_lin)
    agc_channel_any<true>(g, ch, pl, list_lin[wg - n_am], nblocks, compute_n0);
}

// grid = n_fm + n_am + n_lin workgroups, one channel each: one wave, or four (THREADS = 256) of which the other three
// join in on de-emphasised FM channels and leave at once everywhere else
template <int OLEN, int THREADS>
__global__ void __launch_bounds__(THREADS) k_demod64(Geom g, ChanDev ch, Planes pl, const int *__restrict__ list_fm, int n_fm,
                                                     const int *__restrict__ list_am, int n_am,
                                                     const int *__restrict__ list_lin, int n_lin, int nblocks, int compute_n0) {
  int wg = blockIdx.x;
  if (OLEN == 32) {
    if (wg < n_fm) {
      int const c = list_fm[wg];
      if (THREADS == 256 && !(ch.flags[c] & FLAG_FLAT)) {
        fm_channel_four_waves(g, ch, pl, c, nblocks, compute_n0);
        return;
      }
      if (threadIdx.x >= 64) return;
      fm_channel(g, ch, pl, c, nblocks, compute_n0);
      return;
    }
    wg -= n_fm;
  }
  if (wg < n_am) {
    if (THREADS == 256) {  // two of the four waves share an AM channel's two recurrences
      if (threadIdx.x < 128) agc_channel<false, OLEN, true>(g, ch, pl, list_am[wg], nblocks, compute_n0);
      return;
    }
    if (threadIdx.x >= 64) return;
    agc_channel<false, OLEN, false>(g, ch, pl, list_am[wg], nblocks, compute_n0);
    return;
  }
  if (threadIdx.x >= 64) return;
  wg -= n_am;
  if (wg < n_lin) agc_channel<true, OLEN, false>(g, ch, pl, list_lin[wg], nblocks, compute_n0);
}

// Register-resident demodulators exist for olen = 32 (all three types) and olen = 64 (AM / linear)
bool demod64_supported(const Geom &g) { return g.Ndec == 64 && g.olen == 32 && g.Mdec == 33; }
// wave-per-channel AM / linear: every block length (KQ_AGC_WAVE=0: the one-lane-per-channel kernels for the lengths other
// than 32 and 64 -- an A/B switch for tools/bench_mixed.py only: their in-sequence power sums miss the 2e-6 float64 bar of
// tests/test_gpu_parity.py's keyed-signal case at 600 samples per block, which the per-lane partial sums here meet)
bool demod_agc_wave_supported(const Geom &g) {
  static bool const off = getenv("KQ_AGC_WAVE") && atoi(getenv("KQ_AGC_WAVE")) == 0;
  return g.olen == 64 || g.olen == 32 || !off;
}

void launch_demod64(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const int *list_fm, int n_fm,
                    const int *list_am, int n_am, const int *list_lin, int n_lin, int nblocks, int compute_n0) {
  if (g.olen == 32) {
    int const wgs = n_fm + n_am + n_lin;
    if (wgs == 0) return;
    // KQ_DEMOD_ONE_WAVE=1 / 0: the one-wave forms of the FM and AM demodulators always / never (A/B switch).  Otherwise by
    // the number of channels: the multi-wave pipelines exist to fill a device that has one wave per SIMD to run (1024
    // channels); with more than two waves' worth of channels per SIMD the one-wave forms do the same work in fewer
    // instructions, without the LDS hand-overs and the pipeline's fill and drain (rocprofv3, 32768 channels x 2 blocks:
    // 82 us against 157; 8192 x 8: 46 against 61; 1024 x 64: 79 against 47)
    static int const forced = getenv("KQ_DEMOD_ONE_WAVE") ? atoi(getenv("KQ_DEMOD_ONE_WAVE")) : -1;
    bool const one_wave = forced >= 0 ? forced != 0 : wgs > 2048;
    if (n_fm + n_am > 0 && !one_wave)
      hipLaunchKernelGGL((k_demod64<32, 256>), dim3(wgs), dim3(256), 0, s, g, ch, pl, list_fm, n_fm, list_am, n_am, list_lin,
                         n_lin, nblocks, compute_n0);
    else
      hipLaunchKernelGGL((k_demod64<32, 64>), dim3(wgs), dim3(64), 0, s, g, ch, pl, list_fm, n_fm, list_am, n_am, list_lin, n_lin,
                         nblocks, compute_n0);
  } else if (g.olen != 64) {  // any other length: AM / linear by the group (FM stays on the generic kernels)
    int const wgs = n_am + n_lin;
    if (wgs == 0) return;
    hipLaunchKernelGGL(k_demod_agc_any, dim3(wgs), dim3(64), 0, s, g, ch, pl, list_am, n_am, list_lin, n_lin, nblocks, compute_n0);
  } else {  // olen = 64: AM / linear only; FM stays on the generic kernel (launch_demods)
    int const wgs = n_am + n_lin;
    if (wgs == 0) return;
    hipLaunchKernelGGL((k_demod64<64, 64>), dim3(wgs), dim3(64), 0, s, g, ch, pl, list_fm, 0, list_am, n_am, list_lin, n_lin,
                       nblocks, compute_n0);
  }
}

}  // namespace kq
