// kq_demod64.hip -- demodulators specialised for N/D = 64, olen = 32 (BASELINE configs 3 and 4).
//
// One launch serves all three demodulator types (workgroup ranges by type) so they overlap on the chip.
//   FM     : one wave per channel, one lane per sample; everything stays in registers: amplitude statistics
//            by wave reductions, the "hold last good sample" rule of fm.c:128-144 through a ballot mask
//            (previous valid sample = highest set bit below the lane), and the REAL->REAL de-emphasis
//            overlap-save (fm.c:162-171) as 64-point transforms across the 64 lanes (history in lanes 0-31,
//            the new block in lanes 32-63).  No LDS, no barriers.
//   AM/lin : one wave per channel, one lane per sample (two blocks per iteration); only the AGC recurrence
//            (am.c:55-75, linear.c:251-281) is serial, wave-uniform through v_readlane.
// Blocks of one channel are processed in sequence with the state carried in registers, and written back
// to HBM at the end of the launch.
#include "kq_device.hpp"
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
// butterfly reductions over the 64 lanes, every lane gets the result
template <class Op>
__device__ __forceinline__ float wreduce(float v, Op op) {
  int const lane = threadIdx.x & 63;
  v = op(v, lane_xor<32>(v, lane));
  v = op(v, lane_xor<16>(v, lane));
  v = op(v, lane_xor<8>(v, lane));
  v = op(v, lane_xor<4>(v, lane));
  v = op(v, lane_xor<2>(v, lane));
  v = op(v, lane_xor<1>(v, lane));
  return v;
}
__device__ __forceinline__ float wsum(float v) {
  return wreduce(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wmax(float v) {
  return wreduce(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wmin(float v) {
  return wreduce(v, [](float a, float b) { return fminf(a, b); });
}
__device__ __forceinline__ int bitrev6(int i) { return (int)(__brev((unsigned)i) >> 26); }

__device__ __forceinline__ void put_status(kq_chan_status &st, const Geom &g, const ChanDev &ch, const Planes &pl, int c, int b,
                                           int compute_n0, float n0_rate, float &n0) {
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
      n0 += n0_rate * (fresh - n0);
    st.n0 = n0;
  } else {
    st.n0 = NAN;
  }
}

// What the discriminator half of a block (fm.c:91-160) leaves behind for the status record
struct FmStats {
  float bb, snr, foffset, pdev;
  int sq, blanked;
};

// Software-pipelined over the blocks of the channel: the loop body holds the discriminator of block b+1 and the
// de-emphasis filter of block b.  The only loop-carried chain (squelch counter, previous sample, last good audio
// value) runs through the discriminator; the filter hangs off it.  The discriminator is written without branches
// (the squelch decision selects its results) so that both halves sit in one basic block and the scheduler can
// interleave their shuffle chains -- with one wave per SIMD there is nothing else to hide that latency behind.
template <bool FLAT>
__device__ void fm_channel_t(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  int const lane = threadIdx.x & 63;
  bool const upper = lane >= 32;
  int const n = lane - 32;
  float const gain = ch.fm_gain[c];
  int const kbin = bitrev6(lane);
  float2 const HA = (!FLAT && kbin <= 32) ? ch.aresp[(size_t)c * 33 + kbin] : make_float2(0.f, 0.f);
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
  float const noise_gain = ch.noise_gain[c];
  float ifp_v = 0.f, n0raw_v = 0.f;
  const float2 *in = pl.filt + (size_t)c * g.max_blocks * 32;

  // discriminator of one block (fm.c:91-160): S = this lane's sample (upper lanes), returns the audio sample before
  // de-emphasis and updates the carried state
  auto discriminate = [&](float2 S, FmStats &st) -> float {
    float const t = upper ? cnrm(S) : 0.f;
    float const sum_t = wsum(t), sum_a = wsum(sqrtf(t));
    float const bb = sum_t / 64.f;                                   // / (2*olen), fm.c:99
    float const amp = (float)((double)sum_a / (M_SQRT2 * 32));       // fm.c:100
    float const variance = bb - amp * amp;
    float snr = amp * amp / (2 * variance) - 1;
    snr = (0.0f > snr) ? 0.0f : snr;
    int nsq = sq + 1;                                                // fm.c:108-114
    nsq = nsq > 1000 ? 1000 : nsq;
    sq = (snr > 2) ? 0 : nsq;
    bool const open = sq < 2;
    // fm.c:117-154, evaluated unconditionally; `open` selects
    float const thr = (float)(0.55 * 0.55 * amp * amp);
    bool const valid = upper && t > thr;
    unsigned long long const mask = __ballot(valid);
    unsigned long long const below = mask & ((1ull << lane) - 1ull);
    unsigned long long const upto = mask & ((2ull << lane) - 1ull);
    int const pv = below ? 63 - __clzll((long long)below) : -1;
    int const lv = upto ? 63 - __clzll((long long)upto) : -1;
    float2 const sp = shfl2(S, pv >= 0 ? pv : 0);
    float2 const stc = pv >= 0 ? cconj(sp) : state;
    float2 const pr = cmul(S, stc);
    float const y = valid ? atan2f(pr.y, pr.x) : 0.f;
    float const yl = __shfl(y, lv >= 0 ? lv : 0, 64);
    float const out_open = upper ? (lv >= 0 ? yl : lastaudio) : 0.f;
    float const sum_y = wsum(out_open);
    float const vmax = wmax((valid && n > 0) ? y : -INFINITY);
    float const vmin = wmin((valid && n > 0) ? y : INFINITY);
    float const y0 = rdlane(y, 32);
    float const seed = ((mask >> 32) & 1ull) ? y0 : 0.f;
    int const last = mask ? 63 - __clzll((long long)mask) : 0;
    float2 const s_last = cconj(make_float2(rdlane(S.x, last), rdlane(S.y, last)));
    float const y_last = rdlane(y, last);
    // carried state: open and something valid -> last valid sample; open and nothing valid -> unchanged;
    // squelched -> reset (fm.c:156-160)
    bool const any = mask != 0;
    state = open ? (any ? s_last : state) : make_float2(0.f, 0.f);
    lastaudio = open ? (any ? y_last : lastaudio) : 0.f;
    float pdev_pos = fmaxf(seed, vmax), pdev_neg = fminf(seed, vmin);
    float const avg_f = sum_y / 32.f;
    pdev_pos -= avg_f;
    pdev_neg -= avg_f;
    float const mx = (pdev_pos > -pdev_neg) ? pdev_pos : -pdev_neg;
    bool const upd = sq < 1;                                          // fm.c:146
    foffset = upd ? (float)(g.dsamprate * avg_f * (0.5 * M_1_PI)) : foffset;
    pdev = upd ? (float)(g.dsamprate * mx * (0.5 * M_1_PI)) : pdev;
    st.bb = bb;
    st.snr = snr;
    st.foffset = foffset;
    st.pdev = pdev;
    st.sq = sq;
    st.blanked = open ? 32 - __popcll(mask) : 0;
    return open ? out_open : 0.f;
  };

  // de-emphasis overlap-save of one block (fm.c:162-171): [history | block] across the 64 lanes
  auto deemphasize = [&](float out, int b, const FmStats &st) {
    float audio = out;
    if (!FLAT) {
      float2 z = make_float2(upper ? out : hist, 0.f);
#pragma unroll
      for (int s = 5; s >= 0; s--) {  // forward, decimation in frequency: natural in, bit-reversed out
        float2 const r = xor2_pow(z, s, lane);
        z = ((lane >> s) & 1) ? cmul(csub(r, z), wf[s]) : cadd(z, r);
      }
      float2 gk = cmul(HA, z);  // bins 0..32 (filter.c:206-208); zero elsewhere
      if (kbin == 0 || kbin == 32) gk.y = 0.f;
      float2 const mirror = shfl2(gk, herm_src);
      if (kbin > 32) gk = cconj(mirror);  // Hermitian extension of the c2r transform
      z = gk;
#pragma unroll
      for (int s = 0; s < 6; s++) {  // backward, decimation in time: bit-reversed in, natural out
        int const bit = (lane >> s) & 1;
        float2 const v = bit ? cmul(z, wi[s]) : z;
        float2 const r = xor2_pow(v, s, lane);
        z = bit ? csub(r, v) : cadd(v, r);
      }
      audio = z.x * gain;  // fm.c:169-170
    }
    if (upper) pl.audio[((size_t)c * g.max_blocks + b) * 64 + n] = audio;
    hist = lane_xor<32>(out, lane);  // lanes 0..31 take this block as the next history (filter.c:168)
    // per-block inputs of the status record come out of registers (lane b & 63 holds block b's values): a global
    // load here would stall this in-order wave for a full memory round trip per block
    if ((b & 63) == 0) {
      int const bb = b + lane;
      ifp_v = bb < nblocks ? pl.if_power[bb] : 0.f;
      n0raw_v = (compute_n0 && bb < nblocks) ? pl.n0raw[(size_t)c * g.max_blocks + bb] : 0.f;
    }
    float const ifp = rdlane(ifp_v, b & 63), fresh = rdlane(n0raw_v, b & 63);
    if (compute_n0) n0 = isnan(n0) ? fresh : n0 + .01f * (fresh - n0);  // fm.c:79-82
    if (lane == 0) {
      kq_chan_status rec;
      rec.if_power = ifp;
      rec.noise_gain = noise_gain;
      rec.plfreq = NAN;  // N/D = 64: the PL slave would have 2 points (fm.c:203), measurement off
      rec.cphase = 0;
      rec.pll_lock = 0;
      rec.lock_count = 0;
      rec.n0 = compute_n0 ? n0 : NAN;
      rec.bb_power = st.bb;
      rec.snr = st.snr;
      rec.foffset = st.foffset;
      rec.pdeviation = st.pdev;
      rec.agc_gain = 0;
      rec.squelch_count = st.sq;
      rec.hangcount = 0;
      rec.blanked = st.blanked;
      rec.nout = 32;
      pl.status[(size_t)c * g.max_blocks + b] = rec;
    }
  };

  if (nblocks > 0) {
    FmStats st_cur, st_next;
    float2 s_next = (upper && nblocks > 1) ? in[32 + n] : make_float2(0.f, 0.f);
    float out_cur = discriminate(upper ? in[n] : make_float2(0.f, 0.f), st_cur);
    for (int b = 0; b + 1 < nblocks; b++) {
      float2 const S = s_next;
      if (b + 2 < nblocks) s_next = upper ? in[(size_t)(b + 2) * 32 + n] : make_float2(0.f, 0.f);
      float const out_next = discriminate(S, st_next);
      deemphasize(out_cur, b, st_cur);
      out_cur = out_next;
      st_cur = st_next;
    }
    deemphasize(out_cur, nblocks - 1, st_cur);
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

__device__ void fm_channel(const Geom &g, const ChanDev &ch, const Planes &pl, int c, int nblocks, int compute_n0) {
  if (ch.flags[c] & FLAG_FLAT)
    fm_channel_t<true>(g, ch, pl, c, nblocks, compute_n0);
  else
    fm_channel_t<false>(g, ch, pl, c, nblocks, compute_n0);
}

// AM / linear: one wave per channel, one lane per sample, two consecutive blocks per iteration (lanes 0-31 and
// 32-63).  Square roots, the attack gains headroom/level (IEEE divisions), the shift NCO and all loads/stores are
// lane-parallel and coalesced; only the AGC recurrence itself (am.c:64-74, linear.c:269-279: one multiply, one
// compare, two selects per sample) runs serially, wave-uniform, reading its per-sample inputs with v_readlane.
// The arithmetic per sample is exactly the reference's, in the reference's order.
template <bool LINEAR, int OLEN>
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
  for (int b0 = 0; b0 < nblocks; b0 += BPI) {
    int const blk = b0 + half;
    bool const active = blk < nblocks;
    int const nsamp = (b0 + BPI <= nblocks) ? 64 : OLEN;  // wave-uniform
    float2 const S = s_next;
    if (b0 + BPI < nblocks) s_next = (blk + BPI < nblocks) ? in[(size_t)(b0 + BPI) * OLEN + lane] : make_float2(0.f, 0.f);
    float const rp = S.x * S.x, ip = S.y * S.y;
    float level = sqrtf(LINEAR ? rp + ip : S.x * S.x + S.y * S.y);  // amplitude (linear.c:260) / envelope (am.c:58)
    float const env = level;
    // per-block power sums over each OLEN-lane group
    float sig = LINEAR ? rp : rp + ip, noi = LINEAR ? ip : 0.f;
    if (!LINEAR) sig = S.x * S.x + S.y * S.y;
#pragma unroll
    for (int o = OLEN / 2; o > 0; o >>= 1) {
      sig += __shfl_xor(sig, o, 64);
      noi += __shfl_xor(noi, o, 64);
    }
    if (!LINEAR) {
      // carrier tracking (am.c:62), serial; lane i keeps the value after sample i
      float dc_mine = dc;
      for (int i = 0; i < nsamp; i++) {
        float const e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, env), i));
        dc += 0.0001f * (e - dc);
        dc_mine = (lane == i) ? dc : dc_mine;
      }
      level = dc_mine;
    }
    float const inv = headroom / level;
    float g_mine = gain;
    float gain_end[2] = {gain, gain};
    int hang_end[2] = {hang, hang};
    for (int i = 0; i < nsamp; i++) {
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
          put_status(st, g, ch, pl, c, b0 + h, compute_n0, .001f, n0);
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
    ch.gain[c] = gain;
    if (!LINEAR) ch.dc[c] = dc;
    ch.hang[c] = hang;
    ch.n0[c] = n0;
  }
}

}  // namespace

// grid = n_fm + n_am + n_lin workgroups of one wave (one channel each)
template <int OLEN>
__global__ void __launch_bounds__(64) k_demod64(Geom g, ChanDev ch, Planes pl, const int *__restrict__ list_fm, int n_fm,
                                                const int *__restrict__ list_am, int n_am,
                                                const int *__restrict__ list_lin, int n_lin, int nblocks, int compute_n0) {
  int wg = blockIdx.x;
  if (OLEN == 32) {
    if (wg < n_fm) {
      fm_channel(g, ch, pl, list_fm[wg], nblocks, compute_n0);
      return;
    }
    wg -= n_fm;
  }
  if (wg < n_am) {
    agc_channel<false, OLEN>(g, ch, pl, list_am[wg], nblocks, compute_n0);
    return;
  }
  wg -= n_am;
  if (wg < n_lin) agc_channel<true, OLEN>(g, ch, pl, list_lin[wg], nblocks, compute_n0);
}

// Register-resident demodulators exist for olen = 32 (all three types) and olen = 64 (AM / linear)
bool demod64_supported(const Geom &g) { return g.Ndec == 64 && g.olen == 32 && g.Mdec == 33; }
bool demod_agc_wave_supported(const Geom &g) { return g.olen == 64 || g.olen == 32; }

void launch_demod64(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const int *list_fm, int n_fm,
                    const int *list_am, int n_am, const int *list_lin, int n_lin, int nblocks, int compute_n0) {
  if (g.olen == 32) {
    int const wgs = n_fm + n_am + n_lin;
    if (wgs == 0) return;
    hipLaunchKernelGGL(k_demod64<32>, dim3(wgs), dim3(64), 0, s, g, ch, pl, list_fm, n_fm, list_am, n_am, list_lin, n_lin,
                       nblocks, compute_n0);
  } else {  // olen = 64: AM / linear only; FM stays on the generic kernel (launch_demods)
    int const wgs = n_am + n_lin;
    if (wgs == 0) return;
    hipLaunchKernelGGL(k_demod64<64>, dim3(wgs), dim3(64), 0, s, g, ch, pl, list_fm, 0, list_am, n_am, list_lin, n_lin,
                       nblocks, compute_n0);
  }
}

}  // namespace kq
