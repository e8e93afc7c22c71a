// kq_kernels.hip -- gfx950 kernels of the ka9q-radio per-channel DSP hot path.
//
//   k_ingest            int16/int8/float I/Q -> float2 ring, scaled           (radio.c:110-122)
//   k_block_energy      IF power, halving accumulator                          (radio.c:123,143-145)
//   k_filter_full       per (channel, block): NCO mix -> N-point FFT in LDS -> [compute_n0] ->
//                       response multiply / CROSS_CONJ -> N/D-point IFFT -> last olen samples
//                       (radio.c:132-139, filter.c:151, radio.c:383-425, filter.c:206-250)
//   k_demod_fm/am/lin   demodulators with state carried across calls in HBM (fm.c, am.c, linear.c)
//
// The pruned forward path lives in kq_pruned.hip.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "kq_device.hpp"
#include "kq_ctl.hpp"
#include "kq_energy.hpp"
#include "kq_ldsfft.hpp"

namespace kq {

void ensure_dynamic_lds(const void *kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> limit;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(mu);
  size_t &cur = limit[{kernel, dev}];
  if (bytes > cur) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    cur = bytes;
  }
}

// ---------------------------------------------------------------- ingest
__global__ void k_ingest(const void *__restrict__ src, int format, float2 *__restrict__ dst, size_t n, float gain) {
  size_t const stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float2 v;
    if (format == KQ_IQ_S16) {
      short2 const q = reinterpret_cast<const short2 *>(src)[i];
      float const sc = 1.f / 32767.f;  // SCALE16, radio.c:38
      v = make_float2(q.x * sc, q.y * sc);
    } else if (format == KQ_IQ_S8) {
      char2 const q = reinterpret_cast<const char2 *>(src)[i];
      float const sc = 1.f / 127.f;  // SCALE8, radio.c:39
      v = make_float2(q.x * sc, q.y * sc);
    } else {
      v = reinterpret_cast<const float2 *>(src)[i];
    }
    dst[i] = make_float2(v.x * gain, v.y * gain);  // radio.c:122
  }
}

void launch_ingest(hipStream_t s, const void *src, int format, float2 *dst, size_t nsamples, float gain) {
  if (nsamples == 0) return;
  int const threads = 256;
  size_t blocks = (nsamples + threads - 1) / threads;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_ingest, dim3((unsigned)blocks), dim3(threads), 0, s, src, format, dst, nsamples, gain);
}

// Sum |s|^2 over the L new samples of each block (radio.c:123), `split` workgroups per block (a block alone is 15 trips
// of one workgroup at cfg 4: latency, with 32 of 256 CUs busy), each taking every split-th trip of 1024 samples and
// leaving its partial sum in sums[block * split + part].  The workgroups behind the summing ones carry
// the call's parameter block (oscillator planes + update flags) from the pinned host staging slot into device
// memory, 8 bytes per thread straight over the bus: a hipMemcpyAsync in front of this kernel cost ~25 us of idle
// stream per call (copy-engine start-up), this costs nothing.
// Workgroups 0 .. nblocks*split-1: energy; then the copy of the call's staged parameters; then
// (paired != null) the window's history rows.  `paired`: the samples written out once more with their 512-sample rows
// interleaved in pairs, for k_filter_full16k's 16-byte loads (kq_full16k.hip: out[1024 r + 2 c + e] = in[512 (2 r + e)
// + c]) -- the kernel reads every new sample anyway.  L and hist are multiples of 1024 then.
// `prev` != null, the steady state (no oscillator has been set, no channel has come or gone since the call before): the
// eight oscillator planes are not carried over the bus at all but ADVANCED on the device from the planes of the call
// before -- phase(n + adv) = phase + f adv + r adv (adv - 1) / 2, f + r adv (osc.c:39-51 in closed form, as the host's
// Osc::rebase does it), the history's oscillator = the current one, the shift oscillator advanced by adv_out output
// samples -- and only the per-block flag bytes come from the host.  At 32768 channels the planes are 2 MiB per call and
// took 144 us of the call's 1.6 ms to fetch over the link, 8 bytes per thread.
__device__ __forceinline__ double frac_turns(double ph, double f, double n) {
  // ph + f n modulo one turn: the product split exactly (hi + lo = f n to 106 bits), its whole turns dropped before the sum
  double const hi = f * n, lo = __fma_rn(f, n, -hi);
  double const p = ph + ((hi - floor(hi)) + lo);
  return p - floor(p);
}
__global__ void k_block_energy_sum(const float2 *__restrict__ x, int L, float *__restrict__ sums, int nblocks, int split,
                                   const unsigned long long *__restrict__ params_host,
                                   unsigned long long *__restrict__ params_dev, unsigned nwords, int copy_wgs,
                                   float2 *__restrict__ paired, int hist, const double *__restrict__ prev, unsigned nchan,
                                   unsigned cmax, double adv, double adv_out, int hist_wgs,
                                   const unsigned long long *__restrict__ patch_rec, int npatch,
                                   const unsigned long long *__restrict__ patch_bits) {
  int const pcol = 2 * (threadIdx.x & 511) + (threadIdx.x >> 9);  // place of sample (row parity, column) within its pair of rows
  int const nsum = nblocks * split;
  if ((int)blockIdx.x >= nsum + copy_wgs + hist_wgs) {
    // the channels retuned since the last call (kq_bank.cpp: patch_list): their planes as the host staged them, records of
    // (channel index, eight values) in pinned memory.  The advancing threads below leave exactly these channels alone
    // (patch_bits: one bit per channel, staged with the records), so the two never write the same place.
    int const j = ((int)blockIdx.x - nsum - copy_wgs - hist_wgs) * (int)blockDim.x + (int)threadIdx.x;
    if (j >= npatch) return;
    const unsigned long long *r = patch_rec + (size_t)j * 9;
    unsigned const c = (unsigned)r[0];
    if (c >= cmax) return;
    double *planes = reinterpret_cast<double *>(params_dev);
#pragma unroll
    for (int k = 0; k < 8; k++) planes[(size_t)k * cmax + c] = __longlong_as_double((long long)r[1 + k]);
    return;
  }
  if ((int)blockIdx.x >= nsum + copy_wgs) {  // history: copy only, 8192 samples per workgroup
    int const base = ((int)blockIdx.x - nsum - copy_wgs) * 8192;
    for (int j = 0; j < 8 && base + 1024 * j < hist; j++) paired[base + 1024 * j + pcol] = (x - hist)[base + 1024 * j + threadIdx.x];
    return;
  }
  if ((int)blockIdx.x >= nsum) {
    unsigned const i = (blockIdx.x - nsum) * blockDim.x + threadIdx.x;
    if (!prev) {
      if (i < nwords) params_dev[i] = params_host[i];
      return;
    }
    if (i < nchan) {
      if (patch_bits && ((patch_bits[i >> 6] >> (i & 63)) & 1ull)) return;  // a patched channel: the patch role writes its planes
      double *out = reinterpret_cast<double *>(params_dev);
      double const ph = prev[i], f = prev[cmax + i], r = prev[2 * (size_t)cmax + i];
      double const sp = prev[3 * (size_t)cmax + i], sf = prev[4 * (size_t)cmax + i];
      double p = frac_turns(ph, f, adv);
      if (r != 0.0) {
        p += r * (0.5 * adv * (adv - 1.0));
        p -= floor(p);
      }
      double const f2 = f + r * adv;
      out[i] = p;
      out[cmax + i] = f2;
      out[2 * (size_t)cmax + i] = r;
      out[3 * (size_t)cmax + i] = frac_turns(sp, sf, adv_out);
      out[4 * (size_t)cmax + i] = sf;
      out[5 * (size_t)cmax + i] = p;
      out[6 * (size_t)cmax + i] = f2;
      out[7 * (size_t)cmax + i] = r;
    } else if (i - nchan < nwords - 8 * cmax) {  // the per-block flags behind the planes
      params_dev[8 * (size_t)cmax + (i - nchan)] = params_host[8 * (size_t)cmax + (i - nchan)];
    }
    return;
  }
  __shared__ float red_f[16];
  __shared__ int red_i[16];
  int const blk = (int)blockIdx.x / split, part = (int)blockIdx.x % split;
  const float2 *p = x + (size_t)blk * L;
  float acc = 0;
  int dummy = 0;
  int const first = part * (int)blockDim.x + (int)threadIdx.x, stride = split * (int)blockDim.x;
  // four trips' loads in flight at a time (a workgroup has three or four trips in all: one memory latency, not one each)
  float2 *o = paired ? paired + hist + (size_t)blk * L : nullptr;
  for (int i0 = first; i0 < L; i0 += 4 * stride) {  // blockDim.x = 1024 = one pair of rows per trip
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = i0 + j * stride < L ? p[i0 + j * stride] : make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int const i = i0 + j * stride;
      if (i < L) {
        acc += cnrm(v[j]);
        if (o) o[(i - (int)threadIdx.x) + pcol] = v[j];
      }
    }
  }
  block_sum_fi(acc, dummy, red_f, red_i);
  if (threadIdx.x == 0) sums[blockIdx.x] = acc;
}
// the IF-power recurrence as a launch of its own (kq_energy.hpp; the full-spectrum filter kernel runs it on the side instead)
__global__ void k_block_energy_iir(const float *__restrict__ sums, int split, const unsigned char *__restrict__ update,
                                   int nblocks, int L, float *__restrict__ state, float *__restrict__ if_power) {
  block_energy_iir_wave(sums, split, update, nblocks, L, state, if_power, (int)threadIdx.x);
}

int block_energy_split(int L) {
  int const trips = (L + 1023) / 1024;
  return std::max(1, std::min(kEnergySplitMax, (trips + 2) / 3));
}

void launch_block_energy_sum(hipStream_t s, const float2 *newsamples, int L, int nblocks, float *sums, const void *params_host,
                             void *params_dev, size_t params_bytes, float2 *paired, int hist, const double *prev_planes,
                             unsigned nchan, unsigned cmax, double adv, double adv_out, const void *patch_records_host, int npatch,
                             const void *patch_bits_host) {
  unsigned const nwords = (unsigned)((params_bytes + 7) / 8);
  // steady state: one thread per channel advances its planes, then the flag words behind the planes are copied
  unsigned const work = prev_planes ? nchan + (nwords - 8 * cmax) : nwords;
  int const copy_wgs = (int)((work + 1023) / 1024);
  int const hist_wgs = paired ? (hist + 8191) / 8192 : 0;
  int const split = block_energy_split(L);
  if (!prev_planes || !patch_records_host || !patch_bits_host) npatch = 0;  // (patches exist in the steady state only)
  int const patch_wgs = (npatch + 1023) / 1024;
  hipLaunchKernelGGL(k_block_energy_sum, dim3(nblocks * split + copy_wgs + hist_wgs + patch_wgs), dim3(1024), 0, s, newsamples, L, sums,
                     nblocks, split, static_cast<const unsigned long long *>(params_host),
                     static_cast<unsigned long long *>(params_dev), nwords, copy_wgs, paired, hist, prev_planes, nchan, cmax, adv,
                     adv_out, hist_wgs, static_cast<const unsigned long long *>(patch_records_host), npatch,
                     npatch ? static_cast<const unsigned long long *>(patch_bits_host) : nullptr);
}

// Control-plane writes (kq_bank.cpp CtlQueue): what kq_bank_set_filter / add_channel / set_mode ... change on the device,
// gathered by the host in pinned memory since the last call and applied here in ONE launch -- as separate small copies each
// cost the stream 10-20 us of switching between kernel and copy packets (a kq_bank_set_filter came to 0.8 ms of pipeline
// time on a bank at real time, tools/soak_realtime.py).  Record r (32 bytes at the front of the buffer): destination, byte
// count (a multiple of 4), then the offset of its payload in the buffer (kind 0), a 32-bit fill value (kind 1) or the
// device address to copy from (kind 2: a value an earlier kernel of the call left on the device -- the noise gain of a
// response designed in front of the filter pass, kq_design.hip design_launch).
__global__ void __launch_bounds__(256) k_ctl_apply(const unsigned char *__restrict__ q) { ctl_apply_record(q, blockIdx.x, threadIdx.x, 256); }
void launch_ctl_apply(hipStream_t s, const void *queue_host, int nrec) {
  if (nrec > 0) hipLaunchKernelGGL(k_ctl_apply, dim3(nrec), dim3(256), 0, s, static_cast<const unsigned char *>(queue_host));
}

void launch_block_energy_iir(hipStream_t s, const float *sums, int L, int nblocks, const unsigned char *update, float *energy_state,
                             float *if_power) {
  hipLaunchKernelGGL(k_block_energy_iir, dim3(1), dim3(64), 0, s, sums, block_energy_split(L), update, nblocks, L, energy_state,
                     if_power);
}

// Device planes -> pinned host memory, 16 bytes per thread and trip, by a handful of workgroups that stay resident
// (kq_bank_pull_planes_async).  hipMemcpyAsync was measured first: with the filter kernel of the next call filling
// every CU the runtime served these device-to-host copies with its own blit kernel, whose workgroups queued behind the
// filter's -- 1.1 ms for 17 MB, and the filter kernel disturbed (rocprofv3 --memory-copy-trace: __amd_rocclr_copyBuffer,
// no SDMA transfer).  These few workgroups are launched the moment the demodulators finish, take their slots before the
// next filter launch fills the rest, and stream at the link's rate (the stores are posted writes over PCIe).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// audio: rows of `row` floats per channel-block of which the first status.nout hold samples (mono: half of a row) -- only
// those travel; 16 lanes take one row, 16 bytes per lane and trip.  status: the whole plane.  rows = channels * max_blocks.
__global__ void __launch_bounds__(256) k_copy_to_host(const float *__restrict__ audio, float *__restrict__ haudio, int row,
                                                      const kq_chan_status *__restrict__ status, u32x4 *__restrict__ hstatus,
                                                      size_t rows, size_t status16) {
  size_t const stride = (size_t)gridDim.x * blockDim.x;
  size_t const tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (haudio) {
    int const sub = (int)(tid & 15);
    for (size_t r = tid >> 4; r < rows; r += stride >> 4) {
      int const n = min(status[r].nout, row);
      const float *src = audio + r * row;
      float *dst = haudio + r * row;
      for (int i = 4 * sub; i < n; i += 64) {  // row and nout are multiples of 4 here (the caller checks), rows 16-byte aligned
        u32x4 const v = *reinterpret_cast<const u32x4 *>(src + i);
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst + i));
      }
    }
  }
  const u32x4 *sp = reinterpret_cast<const u32x4 *>(status);
  if (hstatus)
    for (size_t i = tid; i < status16; i += stride) __builtin_nontemporal_store(sp[i], hstatus + i);
}
void launch_copy_to_host(hipStream_t s, const float *audio, float *haudio, int row, const kq_chan_status *status, void *hstatus,
                         size_t rows) {
  // As few workgroups as the transfer needs: one pushes ~4.4 GB/s over the link whatever it has in flight, and every wave
  // of this kernel displaces a workgroup of the filter pass running beside it (with_host_io at cfg 4, 12.6 MB per call:
  // 1.59 ms per step with 32 workgroups, 1.48 with 4-8, 1.64 with 2, which no longer finish inside the step).  One per
  // 2 MiB is 26 GB/s at cfg 4.
  size_t const bytes = rows * ((haudio ? (size_t)row * sizeof(float) : 0) + (hstatus ? sizeof(kq_chan_status) : 0));
  unsigned wgs = (unsigned)std::min<size_t>(64, std::max<size_t>(4, (bytes + (2u << 20) - 1) >> 21));
  static int const forced = getenv("KQ_COPY_WGS") ? atoi(getenv("KQ_COPY_WGS")) : 0;  // diagnostic (tools/ab_env_hostio.sh)
  if (forced > 0) wgs = (unsigned)forced;
  hipLaunchKernelGGL(k_copy_to_host, dim3(wgs), dim3(256), 0, s, audio, haudio, row, status, (u32x4 *)hstatus, rows,
                     rows * sizeof(kq_chan_status) / 16);
}

// The same delivery in the reference's own output format: float -> clipped int16 in network byte order (scaleclip,
// audio.c:22-28; htons at audio.c:48,98 -- the arithmetic of k_pcm below) on the way out, half the bytes over the link.
// 16 lanes take one row, 8 words (two 16-byte loads, one 16-byte store) per lane and trip; 480 = 60 x 8, so a lane's 8
// words never straddle one of the 480-word chunks whose all-zero test decides whether the reference sends the packet
// (audio.c:49,99,105): mask bit k = chunk k of the row is all zero.
__device__ __forceinline__ unsigned pcm_word_be(float x) {
  int v;
  if (x >= 1.0f)
    v = 32767;
  else if (x <= -1.0f)
    v = -32768;
  else
    v = (int)(32767.f * x);  // truncation, as the (short) cast of audio.c:27
  unsigned const h = (unsigned)v & 0xffffu;
  return ((h << 8) | (h >> 8)) & 0xffffu;
}
// COMPACT: instead of the whole 64-byte status records, the 24 bytes a receiver reads per block (kq_chan_status_compact:
// bb_power, n0, snr, FM foffset / AM + linear agc.gain, FM squelch counter / AM + linear hang counter, nout) -- at real
// time the status plane is half of the PCM delivery's bytes.  A workgroup packs 256 records into LDS and stores them as
// whole 16-byte pieces (6144 bytes per trip, 16-byte aligned whatever the trip).
template <bool COMPACT>
__global__ void __launch_bounds__(256) k_copy_pcm_to_host(const float *__restrict__ audio, short *__restrict__ hpcm,
                                                          unsigned *__restrict__ hmask, int row,
                                                          const kq_chan_status *__restrict__ status, u32x4 *__restrict__ hstatus,
                                                          size_t rows, size_t status16, const int *__restrict__ mode,
                                                          int max_blocks) {
  size_t const stride = (size_t)gridDim.x * blockDim.x;
  size_t const tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int const sub = (int)(tid & 15);
  for (size_t r = tid >> 4; r < rows; r += stride >> 4) {
    int const n = min(status[r].nout, row);
    const float *src = audio + r * row;
    short *dst = hpcm + r * row;
    unsigned nonzero = 0;  // bit k: this lane saw a non-zero word in chunk k
    for (int i = 8 * sub; i < n; i += 128) {  // row and nout are multiples of 8 here (the caller checks)
      float4 const a = *reinterpret_cast<const float4 *>(src + i), c = *reinterpret_cast<const float4 *>(src + i + 4);
      u32x4 v;
      v.x = pcm_word_be(a.x) | (pcm_word_be(a.y) << 16);
      v.y = pcm_word_be(a.z) | (pcm_word_be(a.w) << 16);
      v.z = pcm_word_be(c.x) | (pcm_word_be(c.y) << 16);
      v.w = pcm_word_be(c.z) | (pcm_word_be(c.w) << 16);
      if ((v.x | v.y | v.z | v.w) != 0) nonzero |= 1u << (i / 480);
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst + i));
    }
    // (the 16 lanes of a row run the same number of trips +-1 and meet again here: rows are handed out 16 lanes at a time)
    for (int m = 1; m < 16; m <<= 1) nonzero |= __shfl_xor(nonzero, m, 16);
    if (sub == 0 && hmask) {
      int const chunks = (n + 479) / 480;
      hmask[r] = ~nonzero & (chunks >= 32 ? 0xffffffffu : ((1u << chunks) - 1u));
    }
  }
  if (!hstatus) return;
  if constexpr (COMPACT) {
    __shared__ __attribute__((aligned(16))) unsigned rec[256 * 6];
    size_t const trips = (rows + 255) / 256;
    for (size_t trip = blockIdx.x; trip < trips; trip += gridDim.x) {  // (uniform per workgroup: the barriers are safe)
      size_t const r0 = trip * 256, r = r0 + threadIdx.x;
      if (r < rows) {
        kq_chan_status const st = status[r];
        bool const fm = mode[r / (size_t)max_blocks] == KQ_FM_DEMOD;
        unsigned *o = rec + 6 * threadIdx.x;
        o[0] = __float_as_uint(st.bb_power);
        o[1] = __float_as_uint(st.n0);
        o[2] = __float_as_uint(st.snr);
        o[3] = __float_as_uint(fm ? st.foffset : st.agc_gain);
        o[4] = (unsigned)(fm ? st.squelch_count : st.hangcount);
        o[5] = (unsigned)st.nout;
      }
      __syncthreads();
      size_t const nrec = min((size_t)256, rows - r0);
      unsigned const words = (unsigned)nrec * 6, full = words / 4;
      u32x4 *dst = reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned *>(hstatus) + r0 * 6);  // r0 * 24 bytes: a multiple of 16
      for (unsigned i = threadIdx.x; i < full; i += 256)
        __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(rec)[i], dst + i);
      if (threadIdx.x < words - 4 * full)  // the last trip's odd record: its trailing 8 bytes
        reinterpret_cast<unsigned *>(dst)[4 * full + threadIdx.x] = rec[4 * full + threadIdx.x];
      __syncthreads();
    }
  } else {
    const u32x4 *sp = reinterpret_cast<const u32x4 *>(status);
    for (size_t i = tid; i < status16; i += stride) __builtin_nontemporal_store(sp[i], hstatus + i);
  }
}
void launch_copy_pcm_to_host(hipStream_t s, const float *audio, short *hpcm, unsigned *hmask, int row,
                             const kq_chan_status *status, void *hstatus, size_t rows, const int *mode_compact, int max_blocks) {
  size_t const sbytes = hstatus ? (mode_compact ? sizeof(kq_chan_status_compact) : sizeof(kq_chan_status)) : 0;
  size_t const bytes = rows * ((size_t)row * sizeof(short) + 4 + sbytes);
  unsigned wgs = (unsigned)std::min<size_t>(64, std::max<size_t>(4, (bytes + (2u << 20) - 1) >> 21));
  static int const forced = getenv("KQ_COPY_WGS") ? atoi(getenv("KQ_COPY_WGS")) : 0;
  if (forced > 0) wgs = (unsigned)forced;
  if (mode_compact)
    hipLaunchKernelGGL(k_copy_pcm_to_host<true>, dim3(wgs), dim3(256), 0, s, audio, hpcm, hmask, row, status, (u32x4 *)hstatus, rows,
                       rows * sizeof(kq_chan_status) / 16, mode_compact, max_blocks);
  else
    hipLaunchKernelGGL(k_copy_pcm_to_host<false>, dim3(wgs), dim3(256), 0, s, audio, hpcm, hmask, row, status, (u32x4 *)hstatus, rows,
                       rows * sizeof(kq_chan_status) / 16, mode_compact, max_blocks);
}

// ---------------------------------------------------------------- full-FFT pre-detection filter
// grid (channel, block); dynamic LDS = N float2.
__global__ void k_filter_full(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ window,
                              const float2 *__restrict__ tw, int compute_n0, float2 *__restrict__ spec_dump, int spec_ch,
                              const int *__restrict__ chan_list) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  __shared__ float red_f[16];
  __shared__ int red_i[16];
  int const c = chan_list ? chan_list[blockIdx.x] : (int)blockIdx.x, b = blockIdx.y;
  int const N = g.N, Ndec = g.Ndec;

  // --- NCO mix (radio.c:132-139): closed form of the phasor recurrence of osc.c:39-51
  double const ph0 = ch.lo_phase[c], f0 = ch.lo_freq[c], r = ch.lo_rate[c];
  double const hp0 = ch.hist_phase[c], hf0 = ch.hist_freq[c], hr = ch.hist_rate[c];
  const float2 *x = window + (size_t)b * g.L;
  double const mbase = (double)b * g.L;
  // One oscillator over the whole window and no sweep (every block but the first one after a retune, every channel that is
  // not Doppler-tracked): a thread's samples are blockDim apart, so its phasor advances by one constant step; evaluated afresh
  // in double every fourth sample, three float products in between (3 x 6e-8 of rounding against the 1e-5 of the parity bar).
  // The per-sample evaluation in double was 800 of this kernel's 1 770 vector instructions per wave (tools/pmc_sq.sh).
  // (samples of an old oscillator: the first hist_len[c] of the call's first window, ChanDev -- block b's window has those
  //  that lie beyond its start, if the history planes differ from the current ones at all)
  bool const same_osc = hr == r && hp0 == ph0 && hf0 == f0;
  int const n_old = same_osc ? 0 : ch.hist_len[c] - b * g.L;
  // (... and of the oscillators before that one, where a channel was retuned again inside M - 1 samples: hist2_*)
  OlderOsc older;
  load_older(ch, c, b * g.L, n_old > 0, older);
  bool const one_osc = r == 0.0 && n_old <= 0;
  if (one_osc) {
    float2 const step = phasor_turns(f0 * (double)blockDim.x);
    float2 lo = make_float2(1.f, 0.f);
    int k = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x, k++) {
      lo = (k & 3) ? cmul(lo, step) : phasor_turns(ph0 + f0 * (mbase + i));
      lds[fft_pos((unsigned)i, g.dN)] = cmul(x[i], lo);
    }
  } else {
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      double const m = mbase + i;
      bool const old = i < n_old;  // mixed before the retune took effect: pre-retune oscillator(s)
      double pp = old ? hp0 : ph0, ff = old ? hf0 : f0, rr = old ? hr : r;
      pick_older(older, i, pp, ff, rr);
      double turns = pp + ff * m;
      if (rr != 0.0) turns += rr * (0.5 * m * (m - 1.0));
      float2 const lo = phasor_turns(turns);
      lds[fft_pos((unsigned)i, g.dN)] = cmul(x[i], lo);
    }
  }
  fft_any<-1>(lds, g.dN, tw, g.tw_log2);  // filter.c:151

  if (spec_dump != nullptr && c == spec_ch) {
    float2 *o = spec_dump + (size_t)b * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) o[i] = lds[i];
  }

  // --- compute_n0 (radio.c:383-425), status only
  if (compute_n0) {
    float const low = ch.low[c], high = ch.high[c];
    float avg = INFINITY;
    for (int iter = 0; iter < 2; iter++) {
      float acc = 0;
      int bins = 0;
      for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int const k = (n <= N / 2) ? n : n - N;
        // the reference forms k*samprate in int (radio.c:407,409): keep its 32-bit wrap
        int const prod = (int)((unsigned)k * (unsigned)g.samprate);
        float const f = (float)prod / N;
        if (f >= low && f <= high) continue;
        float const p = cnrm(lds[n]);
        if (p < avg * 2) {
          acc += p;
          bins++;
        }
      }
      block_sum_fi(acc, bins, red_f, red_i);
      avg = acc / bins;
    }
    if (threadIdx.x == 0) pl.n0raw[(size_t)c * g.max_blocks + b] = (float)(avg / (2.0 * N * g.samprate));
  }

  // --- slave: response multiply (filter.c:206-227), CROSS_CONJ (filter.c:239-249)
  // G goes to the unused middle of the spectrum buffer: bins N_dec/2+1 .. N-N_dec/2 are never read
  // (decimate 1: there is no such middle -- every bin is read -- and the launch has asked for a second buffer behind the first)
  float2 *G = Ndec == N ? lds + N : lds + (Ndec / 2 + 1);
  const float2 *H = ch.resp + (size_t)c * Ndec;
  bool const isb = (ch.fflags[c] & FLAG_ISB) != 0;
  for (int p = threadIdx.x; p <= Ndec / 2; p += blockDim.x) {
    float2 gp = cmul(H[p], lds[p]);
    if (p > 0 && p < Ndec / 2) {
      int const k = Ndec - p;
      float2 gn = cmul(H[k], lds[N - p]);
      if (isb) {
        float2 const pos = gp, neg = gn;
        gp = cadd(pos, cconj(neg));
        gn = csub(neg, cconj(pos));
      }
      G[fft_pos((unsigned)k, g.dNdec)] = gn;
    }
    G[fft_pos((unsigned)p, g.dNdec)] = gp;
  }
  fft_any<+1>(G, g.dNdec, tw, g.tw_log2);  // filter.c:250

  float2 *o = pl.filt + ((size_t)c * g.max_blocks + b) * g.olen;
  for (int i = threadIdx.x; i < g.olen; i += blockDim.x) o[i] = G[Ndec - g.olen + i];  // filter.c:131
}

void launch_filter_full(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                        const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                        const int *chan_list) {
  size_t const lds_bytes = (size_t)g.N * sizeof(float2) * (g.Ndec == g.N ? 2 : 1);
  ensure_dynamic_lds((const void *)k_filter_full, lds_bytes);
  static int const forced = getenv("KQ_FULL_THREADS") ? atoi(getenv("KQ_FULL_THREADS")) : 0;  // A/B switch (tools/bench_mixed.py)
  // By how many workgroups a CU's 160 KiB of LDS hold: 256 threads where there are four or more of them, 512 where two or
  // three, 1024 where one workgroup has the CU to itself (tools/bench_mixed.py and a sweep over N with KQ_FULL_THREADS, filter
  // kernel ms for 1024 channels x 8 blocks at 1024 / 512 / 256 threads: N = 2048 0.33 / 0.16 / 0.13, 4096 0.53 / 0.29 / 0.28,
  // 6144 0.94 / 0.51 / 0.66, 8192 0.90 / 0.64 / 0.74, 9600 1.20 / 0.87 / 1.32, 10240 1.37 / 1.64 / 2.61, 15360 1.74 / 2.10 / 3.44;
  // until round 6 it was 1024 from N = 4096 on)
  int const threads = forced > 0 ? forced : lds_bytes <= 40 * 1024 ? 256 : 2 * lds_bytes + 512 <= 160 * 1024 ? 512 : 1024;
  hipLaunchKernelGGL(k_filter_full, dim3(nchan, nblocks), dim3(threads), lds_bytes, s, g, ch, pl, window, tw, compute_n0,
                     spec_dump, spec_ch, chan_list);
}

// ---------------------------------------------------------------- full path for N beyond one LDS block
// N = S * N1 with N1 * 8 B <= 128 KiB.  Decimation in time over s: F_s = FFT_N1{ xm[S m + s] } and
// X[k] = sum_s W_N^{s k} F_s[k mod N1]; only the N/D bins the slave reads (filter.c:206-227) are combined,
// into a small side buffer.  Same mix, response multiply, CROSS_CONJ and inverse transform as k_filter_full.
// compute_n0 needs every bin of the N-point spectrum and is not available on this path.
// Round 6: N1 and N/D may carry factors 3 and 5 (d1 / g.dNdec; twN = the full-circle table of N points then), so that a
// bank takes N = 19200, 24000, 38400, 48000 ... as well as 32768.
// grid (channel, block); dynamic LDS = (N1 + N_dec) float2.
__global__ void k_filter_split(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ window,
                               const float2 *__restrict__ tw, int S, FftDim d1, const float2 *__restrict__ twN,
                               const int *__restrict__ chan_list) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const c = chan_list ? chan_list[blockIdx.x] : (int)blockIdx.x, b = blockIdx.y;
  int const N = g.N, Ndec = g.Ndec, N1 = d1.n;
  float2 *side = lds + N1;  // X at signed bin k, stored at index k mod N_dec
  for (int i = threadIdx.x; i < Ndec; i += blockDim.x) side[i] = make_float2(0.f, 0.f);

  double const ph0 = ch.lo_phase[c], f0 = ch.lo_freq[c], r = ch.lo_rate[c];
  double const hp0 = ch.hist_phase[c], hf0 = ch.hist_freq[c], hr = ch.hist_rate[c];
  const float2 *x = window + (size_t)b * g.L;
  double const mbase = (double)b * g.L;
  int const n_old = (hr == r && hp0 == ph0 && hf0 == f0) ? 0 : ch.hist_len[c] - b * g.L;  // as in k_filter_full
  OlderOsc older;
  load_older(ch, c, b * g.L, n_old > 0, older);
  for (int s = 0; s < S; s++) {
    __syncthreads();
    for (int i = threadIdx.x; i < N1; i += blockDim.x) {
      int const n = S * i + s;
      double const m = mbase + n;
      bool const old = n < n_old;
      double pp = old ? hp0 : ph0, ff = old ? hf0 : f0, rr = old ? hr : r;
      pick_older(older, n, pp, ff, rr);
      double turns = pp + ff * m;
      if (rr != 0.0) turns += rr * (0.5 * m * (m - 1.0));
      lds[fft_pos((unsigned)i, d1)] = cmul(x[n], phasor_turns(turns));
    }
    fft_any<-1>(lds, d1, tw, g.tw_log2);
    for (int q = threadIdx.x; q < Ndec; q += blockDim.x) {
      int const k = (q <= Ndec / 2) ? q : q - Ndec;            // signed bin
      int const src = (k >= 0) ? k : N1 + k;                   // k mod N1
      int idx = (int)(((long long)s * k) % N);                 // W_N^{s k}
      if (idx < 0) idx += N;
      float2 w;
      if (twN) {
        w = twN[idx];
      } else {
        w = tw[(size_t)(idx & (N / 2 - 1)) << (g.tw_log2 - g.log2N)];
        if (idx >= N / 2) w = make_float2(-w.x, -w.y);
      }
      side[q] = cadd(side[q], cmul(w, lds[src]));
    }
  }
  __syncthreads();
  const float2 *H = ch.resp + (size_t)c * Ndec;
  bool const isb = (ch.fflags[c] & FLAG_ISB) != 0;
  float2 *G = lds;
  for (int p = threadIdx.x; p <= Ndec / 2; p += blockDim.x) {
    float2 gp = cmul(H[p], side[p]);
    if (p > 0 && p < Ndec / 2) {
      int const k = Ndec - p;
      float2 gn = cmul(H[k], side[k]);
      if (isb) {
        float2 const pos = gp, neg = gn;
        gp = cadd(pos, cconj(neg));
        gn = csub(neg, cconj(pos));
      }
      G[fft_pos((unsigned)k, g.dNdec)] = gn;
    }
    G[fft_pos((unsigned)p, g.dNdec)] = gp;
  }
  fft_any<+1>(G, g.dNdec, tw, g.tw_log2);
  float2 *o = pl.filt + ((size_t)c * g.max_blocks + b) * g.olen;
  for (int i = threadIdx.x; i < g.olen; i += blockDim.x) o[i] = G[Ndec - g.olen + i];
}

// the split N = S * N1: the smallest S whose N1 is a size the LDS transform takes (a power of two: S = N / 16384)
static int split_factor(const Geom &g) {
  if (g.dN.log2n >= 0) return g.N > 16384 ? g.N >> 14 : 0;
  for (int S = 2; S <= 64; S++)
    if (g.N % S == 0 && g.N / S <= 16384 && fft_size_ok(g.N / S)) return S;
  return 0;
}
bool split_supported(const Geom &g) {
  int const S = g.N > 16384 && g.N <= 65536 ? split_factor(g) : 0;
  return S > 0 && (size_t)g.Ndec * 8 + (size_t)(g.N / S) * 8 <= 160 * 1024 - 256 && g.Ndec <= g.N / S;
}

void launch_filter_split(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                         const float2 *tw, int nchan, int nblocks, const int *chan_list) {
  int const S = split_factor(g);
  if (S <= 0) return;
  bool ok = false;
  FftDim const d1 = fft_dim(g.N / S, &ok);  // (cached since the bank was created)
  if (!ok) return;
  size_t const lds_bytes = ((size_t)d1.n + g.Ndec) * sizeof(float2);
  ensure_dynamic_lds((const void *)k_filter_split, lds_bytes);
  hipLaunchKernelGGL(k_filter_split, dim3(nchan, nblocks), dim3(1024), lds_bytes, s, g, ch, pl, window, tw, S, d1,
                     g.dN.log2n >= 0 ? (const float2 *)nullptr : g.dN.twc, chan_list);
}

// ---------------------------------------------------------------- demodulators
__device__ __forceinline__ void status_common(kq_chan_status &st, const Geom &g, const ChanDev &ch, const Planes &pl, int c,
                                              int b, int compute_n0, double n0_rate) {
  st.if_power = pl.if_power[b];
  st.noise_gain = ch.noise_gain[c];
  st.plfreq = NAN;
  st.cphase = 0;
  st.pll_lock = 0;
  st.lock_count = 0;
  if (compute_n0) {
    float const fresh = pl.n0raw[(size_t)c * g.max_blocks + b];
    float n0 = ch.n0[c];
    if (isnan(n0))
      n0 = fresh;  // fm.c:79-80
    else
      n0 = (float)((double)n0 + n0_rate * (double)(fresh - n0));  // fm.c:82 / am.c:47 / linear.c:124: double literal
    ch.n0[c] = n0;
    st.n0 = n0;
  } else {
    st.n0 = NAN;
  }
}

// FM, generic geometry, in two kernels.
//   k_demod_fm   amplitude statistics, squelch, discriminator with the hold rule, frequency offset / deviation
//                (fm.c:91-160); the detected samples of every block go to `fmout`.
//   k_fm_audio   the REAL->REAL de-emphasis overlap-save and the PL slave (fm.c:162-171, 219-234), one wave per
//                (channel, block).
// k_demod_fm.  fm.c walks the blocks of a channel in sequence, but what one block hands to the next is small: the
// squelch counter, the last strong sample (conjugated) and the last good audio value, and the offset / deviation
// readings that are only refreshed while the squelch is fully open.  So a workgroup takes one channel and 64 blocks
// at a time in four phases, with W waves sharing the blocks in the two heavy ones:
//   A  (per block)   amplitude statistics -> bb, snr, threshold; the last two strong samples of the block
//   B  (wave 0)      lanes = blocks: squelch counters from the snr flags; for every block the state and audio value it
//                    starts from, found at the nearest earlier block that defines them (squelched: zeros; open with a
//                    strong sample: that sample and its discriminator output)
//   C  (per block)   discriminator, hold rule, sums -> fmout and the block's own offset / deviation
//   D  (wave 0)      lanes = blocks: offset / deviation carried from the nearest block that measured them, the n0
//                    smoother, status records
// Every per-block expression and reduction order is that of the sequential loop, so results do not depend on W.
// Dynamic LDS: per wave  S float2[olen] | Y float[olen].
namespace {
__device__ __forceinline__ void wave_sync() {  // LDS written by one lane of this wave, read by another
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int top_bit(unsigned long long m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ unsigned long long bits_upto(int k) { return (2ull << k) - 1ull; }   // bits 0..k
__device__ __forceinline__ unsigned long long bits_below(int k) { return (1ull << k) - 1ull; }  // bits 0..k-1
}  // namespace

__global__ void __launch_bounds__(1024) k_demod_fm(Geom g, ChanDev ch, Planes pl, float *__restrict__ fmout,
                                                   const int *__restrict__ list, int nblocks, int compute_n0) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  // per-block records of the current 64-block chunk
  __shared__ float r_bb[64], r_snr[64], r_amp[64], r_la_out[64], r_la_in[64], r_foff[64], r_pdev[64];
  __shared__ int r_carry[64], r_pvc[64], r_sq[64], r_blanked[64];
  __shared__ float2 r_sc[64], r_sp[64], r_st_out[64], r_st_in[64];
  int const c = list[blockIdx.x];
  int const lane = threadIdx.x & 63;
  int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
  int const olen = g.olen;
  float2 *S = lds + (size_t)wave * olen;
  float *Y = reinterpret_cast<float *>(lds + (size_t)W * olen) + (size_t)wave * olen;
  // carried from block to block (fm.c:26,68-69 and struct demod); only wave 0 uses them
  float2 c_state = ch.fm_state[c];
  float c_la = ch.lastaudio[c];
  int c_sq = ch.sq_count[c];
  float c_foff = ch.foffset[c], c_pdev = ch.pdev[c];
  float c_n0 = ch.n0[c];
  float const noise_gain = ch.noise_gain[c];

  for (int b0 = 0; b0 < nblocks; b0 += 64) {
    int const nb = min(64, nblocks - b0);
    // ---- A: amplitude statistics (fm.c:91-103) and the last two strong samples of each block
    for (int k = wave; k < nb; k += W) {
      const float2 *in = pl.filt + ((size_t)c * g.max_blocks + b0 + k) * olen;
      float sum_t = 0, sum_a = 0;
      for (int n = lane; n < olen; n += 64) {
        float2 const v = in[n];
        S[n] = v;
        float const t = cnrm(v);
        sum_t += t;
        sum_a += sqrtf(t);
      }
      sum_t = wave_sum(sum_t);
      sum_a = wave_sum(sum_a);
      float const bb = sum_t / (2 * olen);
      float const amp = (float)((double)sum_a / (M_SQRT2 * olen));
      float const variance = bb - amp * amp;
      float snr = amp * amp / (2 * variance) - 1;
      snr = (0.0f > snr) ? 0.0f : snr;  // misc.h max(): NaN propagates
      float const thr = (float)(0.55 * 0.55 * amp * amp);  // fm.c:121
      int carry = -1, pvc = -1;  // last strong sample and the one before it
      for (int cb = 0; cb < olen; cb += 64) {
        int const n = cb + lane;
        bool const valid = n < olen && cnrm(S[n < olen ? n : 0]) > thr;
        unsigned long long const m = __ballot(valid);
        if (m) {
          int const top = top_bit(m);
          unsigned long long const rest = m & ~(1ull << top);
          pvc = rest ? cb + top_bit(rest) : carry;
          carry = cb + top;
        }
      }
      wave_sync();
      if (lane == 0) {
        r_bb[k] = bb;
        r_snr[k] = snr;
        r_amp[k] = amp;
        r_carry[k] = carry;
        r_pvc[k] = pvc;
        r_sc[k] = carry >= 0 ? S[carry] : make_float2(0.f, 0.f);
        r_sp[k] = pvc >= 0 ? S[pvc] : make_float2(0.f, 0.f);
      }
      wave_sync();
    }
    __syncthreads();
    // ---- B: squelch counters and what every block starts from
    if (wave == 0) {
      bool const act = lane < nb;
      bool const reset = act && r_snr[lane] > 2;  // fm.c:108-114
      unsigned long long const rm = __ballot(reset), rl = rm & bits_upto(lane);
      int const sq = rl ? lane - top_bit(rl) : min(c_sq + lane + 1, 1000);
      bool const open = sq < 2;
      int const carry = act ? r_carry[lane] : -1;
      // a squelched block leaves zeros behind (fm.c:156-160), an open one with a strong sample leaves that sample
      bool const def = act && (!open || carry >= 0);
      float2 const sc = r_sc[lane];
      r_st_out[lane] = open ? cconj(sc) : make_float2(0.f, 0.f);
      unsigned long long const dm = __ballot(def), dl = dm & bits_below(lane);
      int const j = dl ? top_bit(dl) : -1;
      wave_sync();
      float2 const st_in = j >= 0 ? r_st_out[j] : c_state;
      float ylast = 0;
      if (open && carry >= 0) {  // the discriminator output at the block's last strong sample (fm.c:130-132)
        float2 const st = r_pvc[lane] >= 0 ? cconj(r_sp[lane]) : st_in;
        float2 const pr = cmul(sc, st);
        ylast = atan2f(pr.y, pr.x);
      }
      r_la_out[lane] = ylast;
      wave_sync();
      r_la_in[lane] = j >= 0 ? r_la_out[j] : c_la;
      r_st_in[lane] = st_in;
      r_sq[lane] = sq;
      if (dm) {
        int const jl = top_bit(dm);
        c_state = r_st_out[jl];
        c_la = r_la_out[jl];
      }
      c_sq = __shfl(sq, nb - 1, 64);
    }
    __syncthreads();
    // ---- C: discriminator and hold rule (fm.c:116-160)
    for (int k = wave; k < nb; k += W) {
      const float2 *in = pl.filt + ((size_t)c * g.max_blocks + b0 + k) * olen;
      float *fo = fmout + ((size_t)c * g.max_blocks + b0 + k) * olen;
      int const sq = r_sq[k];
      int blanked = 0;
      float foff = 0, pdev = 0;
      if (sq < 2) {
        float const amp = r_amp[k];
        float const thr = (float)(0.55 * 0.55 * amp * amp);
        float2 const st_in = r_st_in[k];
        float const la_in = r_la_in[k];
        for (int n = lane; n < olen; n += 64) S[n] = in[n];
        wave_sync();
        int carry = -1;
        for (int cb = 0; cb < olen; cb += 64) {
          int const n = cb + lane;
          float2 const v = S[n < olen ? n : 0];
          bool const valid = n < olen && cnrm(v) > thr;
          unsigned long long const m = __ballot(valid), ml = m & bits_below(lane);
          if (valid) {  // arg(s_n * conj(previous strong sample)), fm.c:130-132
            int const pv = ml ? cb + top_bit(ml) : carry;
            float2 const st = pv >= 0 ? cconj(S[pv]) : st_in;
            float2 const pr = cmul(v, st);
            Y[n] = atan2f(pr.y, pr.x);
          }
          if (m) carry = cb + top_bit(m);
        }
        wave_sync();
        // weak samples repeat the last good audio value (fm.c:141)
        float sum_y = 0, vmax = -INFINITY, vmin = INFINITY;
        bool first_valid = false;
        carry = -1;
        for (int cb = 0; cb < olen; cb += 64) {
          int const n = cb + lane;
          bool const valid = n < olen && cnrm(S[n < olen ? n : 0]) > thr;
          unsigned long long const m = __ballot(valid), mu = m & bits_upto(lane);
          if (cb == 0) first_valid = (m & 1ull) != 0;
          if (n < olen) {
            int const lv = mu ? cb + top_bit(mu) : carry;
            float const y = lv >= 0 ? Y[lv] : la_in;
            fo[n] = y;
            sum_y += y;
            if (valid) {
              if (n > 0) {
                vmax = fmaxf(vmax, y);
                vmin = fminf(vmin, y);
              }
            } else {
              blanked++;
            }
          }
          if (m) carry = cb + top_bit(m);
        }
        sum_y = wave_sum(sum_y);
        vmax = wave_max(vmax);
        vmin = wave_min(vmin);
        blanked = wave_sum_i(blanked);
        // peak-deviation seeds: sample 0 seeds both only when it is strong (fm.c:125-139)
        float const seed = first_valid ? Y[0] : 0.0f;
        float pdev_pos = fmaxf(seed, vmax), pdev_neg = fminf(seed, vmin);
        float const avg_f = sum_y / olen;
        if (sq < 1) {  // fm.c:146-154
          foff = (float)(g.dsamprate * avg_f * (0.5 * M_1_PI));
          pdev_pos -= avg_f;
          pdev_neg -= avg_f;
          float const mx = (pdev_pos > -pdev_neg) ? pdev_pos : -pdev_neg;
          pdev = (float)(g.dsamprate * mx * (0.5 * M_1_PI));
        }
        wave_sync();  // S and Y are reused by this wave's next block
      } else {
        for (int n = lane; n < olen; n += 64) fo[n] = 0;  // fm.c:156-160
      }
      if (lane == 0) {
        r_blanked[k] = blanked;
        r_foff[k] = foff;
        r_pdev[k] = pdev;
      }
    }
    __syncthreads();
    // ---- D: carried readings and the status records
    if (wave == 0) {
      bool const act = lane < nb;
      int const sq = r_sq[lane];
      bool const own = act && sq < 1;
      unsigned long long const om = __ballot(own), ol = om & bits_upto(lane);
      int const jo = ol ? top_bit(ol) : -1;
      float const foffset = jo >= 0 ? r_foff[jo] : c_foff;
      float const pdev = jo >= 0 ? r_pdev[jo] : c_pdev;
      if (om) {
        int const jl = top_bit(om);
        c_foff = r_foff[jl];
        c_pdev = r_pdev[jl];
      }
      float n0_mine = NAN;
      if (compute_n0) {  // fm.c:79-82: a chain in double through the blocks
        float const fresh_v = act ? pl.n0raw[(size_t)c * g.max_blocks + b0 + lane] : 0.f;
        for (int k = 0; k < nb; k++) {
          float const fresh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fresh_v), k));
          c_n0 = isnan(c_n0) ? fresh : (float)((double)c_n0 + .01 * (double)(fresh - c_n0));
          if (lane == k) n0_mine = c_n0;
        }
      }
      if (act) {
        kq_chan_status st;
        st.if_power = pl.if_power[b0 + lane];
        st.noise_gain = noise_gain;
        st.plfreq = NAN;
        st.cphase = 0;
        st.pll_lock = 0;
        st.lock_count = 0;
        st.n0 = n0_mine;
        st.bb_power = r_bb[lane];
        st.snr = r_snr[lane];
        st.foffset = foffset;
        st.pdeviation = pdev;
        st.agc_gain = 0;
        st.squelch_count = sq;
        st.hangcount = 0;
        st.blanked = r_blanked[lane];
        st.nout = olen;
        pl.status[(size_t)c * g.max_blocks + b0 + lane] = st;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ch.n0[c] = c_n0;
    ch.fm_state[c] = c_state;
    ch.lastaudio[c] = c_la;
    ch.sq_count[c] = c_sq;
    ch.foffset[c] = c_foff;
    ch.pdev[c] = c_pdev;
  }
}

// De-emphasis overlap-save and PL slave of one (channel, block): REAL -> REAL (fm.c:162-171, 219-234;
// filter.c:151,206-208,250).  The filter input is the channel's stream of detected samples: `hist_in` holds the
// AM-1 samples that precede block 0 of this call, `fmout` the blocks of the call.  The workgroup of the last block
// writes the AM-1 samples that will precede the next call into `hist_out` (a different buffer: every block-0
// workgroup of this launch is still reading hist_in).
// Dynamic LDS carve:  F float2[AN] | AIN float[AN] | PLB float2[pl_n] | TWL float2[AN/2]
__global__ void __launch_bounds__(256) k_fm_audio(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ tw,
                                                 const float *__restrict__ fmout, const float *__restrict__ hist_in,
                                                 float *__restrict__ hist_out, const int *__restrict__ list, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const c = list[blockIdx.x], b = blockIdx.y;
  int const lane = threadIdx.x, nthr = blockDim.x;  // 64 ... 256 threads (launch_demods): every loop strides by the workgroup
  int const AN = g.Ndec, AM = g.Mdec, AL = g.olen;
  float2 *F = lds;
  float *AIN = reinterpret_cast<float *>(F + AN);
  float2 *PLB = reinterpret_cast<float2 *>(AIN + AN);
  float2 *TWL = PLB + g.pl_n;  // exp(-2 pi i k / AN), k < AN/2
  // AN with a factor 3, 5 or 7 (kq_ldsfft.hpp lds_fft_mixed): twiddles from the plan's own table; where TWL would sit, a second
  // buffer of AN bins takes the products in digit-reversed order (that permutation is no involution: no swapping in place)
  bool const mixed = g.dNdec.log2n < 0;
  float2 *F2 = TWL;
  bool const pl_on = g.pl_n > 0 && pl.plout != nullptr;
  bool const flat = (ch.flags[c] & FLAG_FLAT) != 0;
  const float *stream = fmout + (size_t)c * g.max_blocks * AL;  // detected samples of this call, block after block
  const float *hin = hist_in + (size_t)c * (AM - 1);
  // sample j of the stream, j >= -(AM-1)
  auto sample = [&](long long j) { return j >= 0 ? stream[j] : hin[(AM - 1) + j]; };
  if (b == nblocks - 1) {
    float *ho = hist_out + (size_t)c * (AM - 1);
    for (int i = lane; i < AM - 1; i += nthr) ho[i] = sample((long long)nblocks * AL - (AM - 1) + i);
  }
  float *aud = pl.audio + ((size_t)c * g.max_blocks + b) * (2 * (size_t)AL);
  if (flat && !pl_on) {
    for (int n = lane; n < AL; n += nthr) aud[n] = stream[(size_t)b * AL + n];
    return;
  }
  if (!mixed)
    for (int k = lane; k < AN / 2; k += nthr) TWL[k] = tw[(size_t)k << (g.tw_log2 - g.log2Ndec)];
  for (int i = lane; i < AN; i += nthr) {
    float const v = sample((long long)b * AL - (AM - 1) + i);
    AIN[i] = v;
    F[fft_pos((unsigned)i, g.dNdec)] = make_float2(v, 0.f);
  }
  // forward transform of the audio master (fm.c:162, filter.c:151)
  if (mixed)
    lds_fft_mixed<-1>(F, g.dNdec);
  else
    lds_fft<-1>(F, g.log2Ndec, TWL, g.log2Ndec);
  if (pl_on) {
    // PL slave: REAL -> REAL, decimate 32 (fm.c:219,234; filter.c:206-208 then c2r of pl_n points)
    int const PN = g.pl_n;
    int log2pl = 0;
    while ((1 << log2pl) < PN) log2pl++;
    for (int k = lane; k <= PN / 2; k += nthr) {
      float2 gk = cmul(ch.plresp[k], F[k]);
      if (k == 0 || k == PN / 2) {
        gk.y = 0.f;
      } else {
        PLB[fft_pos((unsigned)(PN - k), g.dPl)] = cconj(gk);
      }
      PLB[fft_pos((unsigned)k, g.dPl)] = gk;
    }
    if (mixed)
      lds_fft_mixed<+1>(PLB, g.dPl);
    else
      lds_fft<+1>(PLB, log2pl, TWL, g.log2Ndec);
    float *po = pl.plout + ((size_t)c * g.max_blocks + b) * g.pl_l;
    for (int n = lane; n < g.pl_l; n += nthr) po[n] = PLB[PN - g.pl_l + n].x;  // filter.c:140
    __syncthreads();
  }
  if (flat) {
    for (int n = lane; n < AL; n += nthr) aud[n] = AIN[AM - 1 + n];
    return;
  }
  // multiply DC..Nyquist (filter.c:206-208) and Hermitian-extend for the c2r transform, which ignores the
  // imaginary parts of DC and Nyquist.  Lane k touches only F[k] and F[AN-k].
  const float2 *HA = ch.aresp + (size_t)c * (AN / 2 + 1);
  if (mixed) {
    for (int k = lane; k <= AN / 2; k += nthr) {
      float2 const gk = cmul(HA[k], F[k]);
      if (k == 0 || k == AN / 2) {
        F2[fft_pos((unsigned)k, g.dNdec)] = make_float2(gk.x, 0.f);
      } else {
        F2[fft_pos((unsigned)k, g.dNdec)] = gk;
        F2[fft_pos((unsigned)(AN - k), g.dNdec)] = cconj(gk);
      }
    }
    lds_fft_mixed<+1>(F2, g.dNdec);
    float const gain = ch.fm_gain[c];
    for (int n = lane; n < AL; n += nthr) aud[n] = F2[AN - AL + n].x * gain;  // fm.c:169-170
    return;
  }
  for (int k = lane; k <= AN / 2; k += nthr) {
    float2 const gk = cmul(HA[k], F[k]);
    if (k == 0 || k == AN / 2) {
      F[k] = make_float2(gk.x, 0.f);
    } else {
      F[k] = gk;
      F[AN - k] = cconj(gk);
    }
  }
  __syncthreads();
  for (int i = lane; i < AN; i += nthr) {  // bit-reverse in place, then backward transform
    unsigned const r = bitrev((unsigned)i, g.log2Ndec);
    if (r > (unsigned)i) {
      float2 const t = F[i];
      F[i] = F[r];
      F[r] = t;
    }
  }
  lds_fft<+1>(F, g.log2Ndec, TWL, g.log2Ndec);
  float const gain = ch.fm_gain[c];
  for (int n = lane; n < AL; n += nthr) aud[n] = F[AN - AL + n].x * gain;  // fm.c:169-170
}

// The same de-emphasis overlap-save for AN = 256 (AL = 128, AM = 129: BASELINE cfg 2's geometry) with the PL measurement off:
// one wave per PAIR of blocks, registers and lane exchanges only (k_demod64's scheme for 64 points, four values per lane).
// The two real windows [b-1 | b] and [b | b+1] are the real and imaginary part of ONE complex 256-point sequence z; both are
// filtered by the same response, so Y0 + i Y1 = HAf . (W0 + i W1) = HAf . Z with HAf the response's Hermitian extension
// (real at DC and Nyquist, whose imaginary parts the c2r transform ignores, filter.c:250) -- no separation of the two
// spectra is needed -- and one inverse transform returns block b's audio as its real part, block b+1's as its imaginary part.
// 256 = 4 x 64: i = m + 64 a, k = 4 q + r (lane m, register a or r).  Forward, decimation in frequency: in-lane radix-4
// over a, twiddle W_256^{m r}, then four 64-point transforms across the lanes (natural in, bit-reversed out: lane l holds
// q = bitrev6(l)); inverse the other way round (bit-reversed in, natural out).  fm.c:162-171, filter.c:151,206-208,250.
__global__ void __launch_bounds__(64) k_fm_audio256(Geom g, ChanDev ch, Planes pl, const float *__restrict__ fmout,
                                                    const float *__restrict__ hist_in, float *__restrict__ hist_out,
                                                    const int *__restrict__ list, int nblocks) {
  constexpr int AL = 128, AN = 256;
  int const c = list[blockIdx.x], b0 = 2 * (int)blockIdx.y, lane = threadIdx.x;
  bool const have1 = b0 + 1 < nblocks;
  const float *stream = fmout + (size_t)c * g.max_blocks * AL;  // detected samples of this call, block after block
  const float *pm = b0 > 0 ? stream + (size_t)(b0 - 1) * AL : hist_in + (size_t)c * AL;  // AM - 1 = 128: exactly one block
  float const p0 = pm[lane], p1 = pm[lane + 64];
  float const c0 = stream[(size_t)b0 * AL + lane], c1 = stream[(size_t)b0 * AL + lane + 64];
  float const n0 = have1 ? stream[(size_t)(b0 + 1) * AL + lane] : 0.f, n1 = have1 ? stream[(size_t)(b0 + 1) * AL + lane + 64] : 0.f;
  if (b0 + 2 >= nblocks) {  // the call's last block precedes the next call (filter.c:164)
    float *ho = hist_out + (size_t)c * AL;
    ho[lane] = have1 ? n0 : c0;
    ho[lane + 64] = have1 ? n1 : c1;
  }
  float *aud0 = pl.audio + ((size_t)c * g.max_blocks + b0) * (2 * (size_t)AL);
  float *aud1 = aud0 + 2 * AL;
  bool const pl_on = g.pl_n == 8 && pl.plout != nullptr;  // the PL slave (fm.c:201-234) reads bins 0..4 of the same transform
  bool const flat = (ch.flags[c] & FLAG_FLAT) != 0;
  if (flat) {  // fm.c:164-172: no filter, no gain
    aud0[lane] = c0;
    aud0[lane + 64] = c1;
    if (have1) {
      aud1[lane] = n0;
      aud1[lane + 64] = n1;
    }
    if (!pl_on) return;
  }
  // response on this lane's four bins k = 4 bitrev6(lane) + r, Hermitian-extended
  const float2 *HA = ch.aresp + (size_t)c * (AN / 2 + 1);
  int const q = (int)(__brev((unsigned)lane) >> 26);
  float2 hf[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int const k = 4 * q + r;
    float2 const t = HA[k <= AN / 2 ? k : AN - k];
    hf[r] = k <= AN / 2 ? t : cconj(t);
    if (k == 0 || k == AN / 2) hf[r].y = 0.f;
  }
  // lane-exchange twiddles of the 64-point transforms and the radix-4 twiddles W_256^{lane r}
  float2 wf[6], wi[6], w4[3];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    int const half = 1 << s;
    float sn, cs;
    sincospif((float)(lane & (half - 1)) / (float)half, &sn, &cs);
    wf[s] = make_float2(cs, -sn);
    wi[s] = make_float2(cs, sn);
  }
#pragma unroll
  for (int r = 1; r < 4; r++) {
    float sn, cs;
    sincospif((float)(lane * r) / 128.f, &sn, &cs);
    w4[r - 1] = make_float2(cs, -sn);
  }
  auto xor_pow = [&](float2 v, int s) {
    switch (s) {
      case 0: return make_float2(lane_xor<1>(v.x, lane), lane_xor<1>(v.y, lane));
      case 1: return make_float2(lane_xor<2>(v.x, lane), lane_xor<2>(v.y, lane));
      case 2: return make_float2(lane_xor<4>(v.x, lane), lane_xor<4>(v.y, lane));
      case 3: return make_float2(lane_xor<8>(v.x, lane), lane_xor<8>(v.y, lane));
      case 4: return make_float2(lane_xor<16>(v.x, lane), lane_xor<16>(v.y, lane));
      default: return make_float2(lane_xor<32>(v.x, lane), lane_xor<32>(v.y, lane));
    }
  };
  auto muli = [](float2 a) { return make_float2(-a.y, a.x); };  // i a
  // z[m + 64 a]: real part the window [b0-1 | b0], imaginary part [b0 | b0+1]
  float2 const z0 = make_float2(p0, c0), z1 = make_float2(p1, c1), z2 = make_float2(c0, n0), z3 = make_float2(c1, n1);
  float2 u[4], xf[4];
  {
    float2 const t0 = cadd(z0, z2), t1 = csub(z0, z2), t2 = cadd(z1, z3), t3 = csub(z1, z3);
    u[0] = cadd(t0, t2);
    u[2] = cmul(csub(t0, t2), w4[1]);
    u[1] = cmul(csub(t1, muli(t3)), w4[0]);
    u[3] = cmul(cadd(t1, muli(t3)), w4[2]);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float2 z = u[r];
#pragma unroll
    for (int s = 5; s >= 0; s--) {  // forward, decimation in frequency: natural in, bit-reversed out
      float2 const o = xor_pow(z, s);
      z = ((lane >> s) & 1) ? cmul(csub(o, z), wf[s]) : cadd(z, o);
    }
    xf[r] = z;           // X[4 bitrev6(lane) + r] of the pair's packed transform (the PL slave reads a few of them)
    z = cmul(hf[r], z);  // filter.c:206-208 on both windows at once
#pragma unroll
    for (int s = 0; s < 6; s++) {  // backward, decimation in time: bit-reversed in, natural out
      int const bit = (lane >> s) & 1;
      float2 const v = bit ? cmul(z, wi[s]) : z;
      float2 const o = xor_pow(v, s);
      z = bit ? csub(o, v) : cadd(v, o);
    }
    u[r] = r ? cmul(z, cconj(w4[r - 1])) : z;
  }
  if (pl_on) {
    // PL slave of both blocks (fm.c:219,234: REAL -> REAL, decimate 32, 8 points, the last 4 kept): it needs bins 0..4 of each
    // window's own transform, W0[k] = (X[k] + conj X[256 - k]) / 2 and W1[k] = (X[k] - conj X[256 - k]) / 2i -- nine values
    // of the packed transform, held by lanes 0 (bins 0..3), 32 (bin 4) and 63 (bins 252..255).  Lane j < 8 forms output
    // n = 4 + (j & 3) of window j >> 2: y[n] = G0 + (-1)^n G4 + 2 Re sum_{k=1..3} G[k] e^{2 pi i k n / 8}, G = plresp . W.
    auto rd = [&](float2 v, int src) { return make_float2(__shfl(v.x, src, 64), __shfl(v.y, src, 64)); };
    float2 const X0 = rd(xf[0], 0), X1 = rd(xf[1], 0), X2 = rd(xf[2], 0), X3 = rd(xf[3], 0), X4 = rd(xf[0], 32);
    float2 const Xm1 = rd(xf[3], 63), Xm2 = rd(xf[2], 63), Xm3 = rd(xf[1], 63), Xm4 = rd(xf[0], 63);
    float2 const Xk[5] = {X0, X1, X2, X3, X4}, Xn[5] = {X0, Xm1, Xm2, Xm3, Xm4};
    int const w = (lane >> 2) & 1, n = 4 + (lane & 3);
    float y = 0.f;
#pragma unroll
    for (int k = 0; k <= 4; k++) {
      float2 const a = Xk[k], bc = cconj(Xn[k]);
      // window 0: (a + b) / 2; window 1: (a - b) / 2i = -i (a - b) / 2
      float2 const d = csub(a, bc);
      float2 const W = w ? make_float2(0.5f * d.y, -0.5f * d.x) : make_float2(0.5f * (a.x + bc.x), 0.5f * (a.y + bc.y));
      float2 gk = cmul(ch.plresp[k], W);
      if (k == 0 || k == 4) {
        y += (k == 4 && (n & 1)) ? -gk.x : gk.x;  // the c2r transform ignores the imaginary parts of DC and Nyquist
      } else {
        float sn, cs;
        sincospif((float)(k * n) * 0.25f, &sn, &cs);
        y += 2.f * (gk.x * cs - gk.y * sn);
      }
    }
    if (lane < 8 && (w == 0 || have1)) pl.plout[((size_t)c * g.max_blocks + b0 + w) * g.pl_l + (lane & 3)] = y;  // filter.c:140
    if (flat) return;
  }
  // inverse radix-4 over r, outputs i = 128 + m (a = 2) and 192 + m (a = 3) only: the samples the slave keeps (filter.c:140)
  float2 const y2 = csub(cadd(u[0], u[2]), cadd(u[1], u[3]));
  float2 const y3 = csub(csub(u[0], u[2]), muli(csub(u[1], u[3])));
  float const gain = ch.fm_gain[c];
  aud0[lane] = y2.x * gain;  // fm.c:169-170
  aud0[lane + 64] = y3.x * gain;
  if (have1) {
    aud1[lane] = y2.y * gain;
    aud1[lane + 64] = y3.y * gain;
  }
}

// The whole FM demodulator of one channel in ONE launch for N/D = 256 (cfg 2's geometry: 128 samples per block, de-emphasis
// filter of 129 taps) without the PL measurement: k_demod_fm's four phases and k_fm_audio256's overlap-save on the same
// 16-wave workgroup, the call's blocks held in LDS from the first load to the audio store.  What the two-kernel form pays
// and this does not: the second launch, the detected samples' round trip through memory (8 MB per call at cfg 2), and one
// exposed memory latency per block and phase -- a wave asks for all its blocks' samples at once here, and phase C finds
// them in LDS.  Per-block expressions and reduction orders are those of k_demod_fm / k_fm_audio256 (and so of the
// sequential loop of fm.c:91-171): the results are theirs bit for bit.
// Static LDS: S[64][128] float2 (64 KiB) | FO[65][128] float (row 0 = the block before the chunk) | Y[16][128] float.
namespace {
struct Audio256 {  // k_fm_audio256's transform pair, set up once per wave
  // wf / wi: the lane-exchange stages' twiddles as the lane applies them -- the stage's twiddle in the upper lane of a
  // butterfly pair, 1 in the lower one -- and sg: -1 / +1 likewise.  A stage is then the same six instructions in every
  // lane (round 6; until then both arms of `bit ? (o - z) w : z + o` were computed and one selected: twelve).  The values
  // are the old form's bit for bit: o - z and z + o round once either way, and a product with (1, 0) is exact.
  float2 hf[4], wf[6], wi[6], w4[3];
  float sg[6];
  int lane;
  __device__ __forceinline__ void init(int lane_, const float2 *HA) {
    lane = lane_;
    int const q = (int)(__brev((unsigned)lane) >> 26);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int const k = 4 * q + r;
      float2 const t = HA[k <= 128 ? k : 256 - k];
      hf[r] = k <= 128 ? t : cconj(t);
      if (k == 0 || k == 128) hf[r].y = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 6; s++) {
      int const half = 1 << s;
      float sn, cs;
      sincospif((float)(lane & (half - 1)) / (float)half, &sn, &cs);
      bool const up = (lane >> s) & 1;
      wf[s] = up ? make_float2(cs, -sn) : make_float2(1.f, 0.f);
      wi[s] = up ? make_float2(cs, sn) : make_float2(1.f, 0.f);
      sg[s] = up ? -1.f : 1.f;
    }
#pragma unroll
    for (int r = 1; r < 4; r++) {
      float sn, cs;
      sincospif((float)(lane * r) / 128.f, &sn, &cs);
      w4[r - 1] = make_float2(cs, -sn);
    }
  }
  __device__ __forceinline__ float2 xor_pow(float2 v, int s) const {
    switch (s) {
      case 0: return make_float2(lane_xor<1>(v.x, lane), lane_xor<1>(v.y, lane));
      case 1: return make_float2(lane_xor<2>(v.x, lane), lane_xor<2>(v.y, lane));
      case 2: return make_float2(lane_xor<4>(v.x, lane), lane_xor<4>(v.y, lane));
      case 3: return make_float2(lane_xor<8>(v.x, lane), lane_xor<8>(v.y, lane));
      case 4: return make_float2(lane_xor<16>(v.x, lane), lane_xor<16>(v.y, lane));
      default: return make_float2(lane_xor<32>(v.x, lane), lane_xor<32>(v.y, lane));
    }
  }
  // z[m + 64 a] = (window [b-1 | b], window [b | b+1]) -> the filtered samples 128 + m and 192 + m of both (real / imaginary part)
  __device__ __forceinline__ void run(float2 z0, float2 z1, float2 z2, float2 z3, float2 &y2, float2 &y3) const {
    auto muli = [](float2 a) { return make_float2(-a.y, a.x); };  // i a
    float2 u[4];
    {
      float2 const t0 = cadd(z0, z2), t1 = csub(z0, z2), t2 = cadd(z1, z3), t3 = csub(z1, z3);
      u[0] = cadd(t0, t2);
      u[2] = cmul(csub(t0, t2), w4[1]);
      u[1] = cmul(csub(t1, muli(t3)), w4[0]);
      u[3] = cmul(cadd(t1, muli(t3)), w4[2]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float2 z = u[r];
#pragma unroll
      for (int s = 5; s >= 0; s--) {  // forward, decimation in frequency: upper lane (o - z) w, lower lane z + o
        float2 const o = xor_pow(z, s);
        z = cmul(make_float2(fmaf(z.x, sg[s], o.x), fmaf(z.y, sg[s], o.y)), wf[s]);
      }
      z = cmul(hf[r], z);
#pragma unroll
      for (int s = 0; s < 6; s++) {  // backward, decimation in time: v = z w (upper) / z (lower); upper o - v, lower v + o
        float2 const v = cmul(z, wi[s]);
        float2 const o = xor_pow(v, s);
        z = make_float2(fmaf(v.x, sg[s], o.x), fmaf(v.y, sg[s], o.y));
      }
      u[r] = r ? cmul(z, cconj(w4[r - 1])) : z;
    }
    y2 = csub(cadd(u[0], u[2]), cadd(u[1], u[3]));
    y3 = csub(csub(u[0], u[2]), muli(csub(u[1], u[3])));
  }
};
}  // namespace

// W waves per channel: 8 (two per SIMD, 173 registers) or 16 (four per SIMD: the 128 registers that leaves spill 19 since the
// exchange stages of round 6 -- 112 before, when 16 was first tried and dropped); KQ_FM256_WAVES picks, see launch_demods
template <int W>
__global__ void __launch_bounds__(64 * W) k_demod_fm256(Geom g, ChanDev ch, Planes pl, const float *__restrict__ hist_in,
                                                         float *__restrict__ hist_out, const int *__restrict__ list, int nblocks,
                                                         int compute_n0) {
  constexpr int olen = 128;
  __shared__ __attribute__((aligned(16))) float2 S[64 * olen];
  __shared__ __attribute__((aligned(16))) float FO[65 * olen];
  __shared__ float Yall[W * olen];
  __shared__ float r_bb[64], r_snr[64], r_amp[64], r_la_out[64], r_la_in[64], r_foff[64], r_pdev[64];
  __shared__ int r_carry[64], r_pvc[64], r_sq[64], r_blanked[64];
  __shared__ float2 r_sc[64], r_sp[64], r_st_out[64], r_st_in[64];
  int const c = list[blockIdx.x];
  int const lane = threadIdx.x & 63;
  int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Y = Yall + wave * olen;
  float2 c_state = ch.fm_state[c];
  float c_la = ch.lastaudio[c];
  int c_sq = ch.sq_count[c];
  float c_foff = ch.foffset[c], c_pdev = ch.pdev[c];
  float c_n0 = ch.n0[c];
  float const noise_gain = ch.noise_gain[c];
  bool const flat = (ch.flags[c] & FLAG_FLAT) != 0;
  float const gain = ch.fm_gain[c];
  if (threadIdx.x < olen) FO[threadIdx.x] = hist_in[(size_t)c * olen + threadIdx.x];  // the block before the call (AM - 1 = 128)

  for (int b0 = 0; b0 < nblocks; b0 += 64) {
    int const nb = min(64, nblocks - b0);
    // ---- A: all of this wave's blocks asked for at once, then statistics block by block (fm.c:91-103)
    constexpr int kPer = 64 / W;
    float2 va[kPer], vb[kPer];
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      int const k = wave + W * i;
      if (k < nb) {
        const float2 *in = pl.filt + ((size_t)c * g.max_blocks + b0 + k) * olen;
        va[i] = in[lane];
        vb[i] = in[lane + 64];
      }
    }
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      int const k = wave + W * i;
      if (k >= nb) break;
      float2 *Sk = S + k * olen;
      Sk[lane] = va[i];
      Sk[lane + 64] = vb[i];
      float sum_t = 0, sum_a = 0;
      {  // (n = lane, then n = lane + 64: the accumulation order of k_demod_fm's loop)
        float const t0 = cnrm(va[i]);
        sum_t += t0;
        sum_a += sqrtf(t0);
        float const t1 = cnrm(vb[i]);
        sum_t += t1;
        sum_a += sqrtf(t1);
      }
      sum_t = wave_sum(sum_t);
      sum_a = wave_sum(sum_a);
      float const bb = sum_t / (2 * olen);
      float const amp = (float)((double)sum_a / (M_SQRT2 * olen));
      float const variance = bb - amp * amp;
      float snr = amp * amp / (2 * variance) - 1;
      snr = (0.0f > snr) ? 0.0f : snr;  // misc.h max(): NaN propagates
      float const thr = (float)(0.55 * 0.55 * amp * amp);  // fm.c:121
      int carry = -1, pvc = -1;  // last strong sample and the one before it
#pragma unroll
      for (int h = 0; h < 2; h++) {
        bool const valid = cnrm(h ? vb[i] : va[i]) > thr;
        unsigned long long const m = __ballot(valid);
        if (m) {
          int const top = top_bit(m);
          unsigned long long const rest = m & ~(1ull << top);
          pvc = rest ? 64 * h + top_bit(rest) : carry;
          carry = 64 * h + top;
        }
      }
      wave_sync();
      if (lane == 0) {
        r_bb[k] = bb;
        r_snr[k] = snr;
        r_amp[k] = amp;
        r_carry[k] = carry;
        r_pvc[k] = pvc;
        r_sc[k] = carry >= 0 ? Sk[carry] : make_float2(0.f, 0.f);
        r_sp[k] = pvc >= 0 ? Sk[pvc] : make_float2(0.f, 0.f);
      }
    }
    __syncthreads();
    // ---- B: squelch counters and what every block starts from (k_demod_fm's phase B)
    if (wave == 0) {
      bool const act = lane < nb;
      bool const reset = act && r_snr[lane] > 2;  // fm.c:108-114
      unsigned long long const rm = __ballot(reset), rl = rm & bits_upto(lane);
      int const sq = rl ? lane - top_bit(rl) : min(c_sq + lane + 1, 1000);
      bool const open = sq < 2;
      int const carry = act ? r_carry[lane] : -1;
      bool const def = act && (!open || carry >= 0);
      float2 const sc = r_sc[lane];
      r_st_out[lane] = open ? cconj(sc) : make_float2(0.f, 0.f);
      unsigned long long const dm = __ballot(def), dl = dm & bits_below(lane);
      int const j = dl ? top_bit(dl) : -1;
      wave_sync();
      float2 const st_in = j >= 0 ? r_st_out[j] : c_state;
      float ylast = 0;
      if (open && carry >= 0) {  // the discriminator output at the block's last strong sample (fm.c:130-132)
        float2 const st = r_pvc[lane] >= 0 ? cconj(r_sp[lane]) : st_in;
        float2 const pr = cmul(sc, st);
        ylast = atan2f(pr.y, pr.x);
      }
      r_la_out[lane] = ylast;
      wave_sync();
      r_la_in[lane] = j >= 0 ? r_la_out[j] : c_la;
      r_st_in[lane] = st_in;
      r_sq[lane] = sq;
      if (dm) {
        int const jl = top_bit(dm);
        c_state = r_st_out[jl];
        c_la = r_la_out[jl];
      }
      c_sq = __shfl(sq, nb - 1, 64);
    }
    __syncthreads();
    // ---- C: discriminator and hold rule (fm.c:116-160), samples from LDS, detected samples into FO[k + 1]
    for (int k = wave; k < nb; k += W) {
      const float2 *Sk = S + k * olen;
      float *fo = FO + (k + 1) * olen;
      int const sq = r_sq[k];
      int blanked = 0;
      float foff = 0, pdev = 0;
      if (sq < 2) {
        float const amp = r_amp[k];
        float const thr = (float)(0.55 * 0.55 * amp * amp);
        float2 const st_in = r_st_in[k];
        float const la_in = r_la_in[k];
        int carry = -1;
        for (int cb = 0; cb < olen; cb += 64) {
          int const n = cb + lane;
          float2 const v = Sk[n];
          bool const valid = cnrm(v) > thr;
          unsigned long long const m = __ballot(valid), ml = m & bits_below(lane);
          if (valid) {  // arg(s_n * conj(previous strong sample)), fm.c:130-132
            int const pv = ml ? cb + top_bit(ml) : carry;
            float2 const st = pv >= 0 ? cconj(Sk[pv]) : st_in;
            float2 const pr = cmul(v, st);
            Y[n] = atan2f(pr.y, pr.x);
          }
          if (m) carry = cb + top_bit(m);
        }
        wave_sync();
        float sum_y = 0, vmax = -INFINITY, vmin = INFINITY;
        bool first_valid = false;
        carry = -1;
        for (int cb = 0; cb < olen; cb += 64) {
          int const n = cb + lane;
          bool const valid = cnrm(Sk[n]) > thr;
          unsigned long long const m = __ballot(valid), mu = m & bits_upto(lane);
          if (cb == 0) first_valid = (m & 1ull) != 0;
          int const lv = mu ? cb + top_bit(mu) : carry;
          float const y = lv >= 0 ? Y[lv] : la_in;  // weak samples repeat the last good audio value (fm.c:141)
          fo[n] = y;
          sum_y += y;
          if (valid) {
            if (n > 0) {
              vmax = fmaxf(vmax, y);
              vmin = fminf(vmin, y);
            }
          } else {
            blanked++;
          }
          if (m) carry = cb + top_bit(m);
        }
        sum_y = wave_sum(sum_y);
        vmax = wave_max(vmax);
        vmin = wave_min(vmin);
        blanked = wave_sum_i(blanked);
        float const seed = first_valid ? Y[0] : 0.0f;  // fm.c:125-139
        float pdev_pos = fmaxf(seed, vmax), pdev_neg = fminf(seed, vmin);
        float const avg_f = sum_y / olen;
        if (sq < 1) {  // fm.c:146-154
          foff = (float)(g.dsamprate * avg_f * (0.5 * M_1_PI));
          pdev_pos -= avg_f;
          pdev_neg -= avg_f;
          float const mx = (pdev_pos > -pdev_neg) ? pdev_pos : -pdev_neg;
          pdev = (float)(g.dsamprate * mx * (0.5 * M_1_PI));
        }
        wave_sync();  // Y is reused by this wave's next block
      } else {
        fo[lane] = 0;  // fm.c:156-160
        fo[lane + 64] = 0;
      }
      if (lane == 0) {
        r_blanked[k] = blanked;
        r_foff[k] = foff;
        r_pdev[k] = pdev;
      }
    }
    __syncthreads();
    // ---- D (wave 0): carried readings and the status records; the other waves start on the audio filter meanwhile
    if (wave == 0) {
      bool const act = lane < nb;
      int const sq = r_sq[lane];
      bool const own = act && sq < 1;
      unsigned long long const om = __ballot(own), ol = om & bits_upto(lane);
      int const jo = ol ? top_bit(ol) : -1;
      float const foffset = jo >= 0 ? r_foff[jo] : c_foff;
      float const pdev = jo >= 0 ? r_pdev[jo] : c_pdev;
      if (om) {
        int const jl = top_bit(om);
        c_foff = r_foff[jl];
        c_pdev = r_pdev[jl];
      }
      float n0_mine = NAN;
      if (compute_n0) {  // fm.c:79-82: a chain in double through the blocks
        float const fresh_v = act ? pl.n0raw[(size_t)c * g.max_blocks + b0 + lane] : 0.f;
        for (int k = 0; k < nb; k++) {
          float const fresh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fresh_v), k));
          c_n0 = isnan(c_n0) ? fresh : (float)((double)c_n0 + .01 * (double)(fresh - c_n0));
          if (lane == k) n0_mine = c_n0;
        }
      }
      if (act) {
        kq_chan_status st;
        st.if_power = pl.if_power[b0 + lane];
        st.noise_gain = noise_gain;
        st.plfreq = NAN;
        st.cphase = 0;
        st.pll_lock = 0;
        st.lock_count = 0;
        st.n0 = n0_mine;
        st.bb_power = r_bb[lane];
        st.snr = r_snr[lane];
        st.foffset = foffset;
        st.pdeviation = pdev;
        st.agc_gain = 0;
        st.squelch_count = sq;
        st.hangcount = 0;
        st.blanked = r_blanked[lane];
        st.nout = olen;
        pl.status[(size_t)c * g.max_blocks + b0 + lane] = st;
      }
    }
    // ---- audio: REAL -> REAL de-emphasis overlap-save on pairs of blocks (fm.c:162-171), FO row k + 1 = block b0 + k
    // (the transform's constants are formed here and not at the top: held across phases A - C they spill)
    Audio256 af;
    if (!flat) af.init(lane, ch.aresp + (size_t)c * 129);
    for (int pr = (wave + W - 1) % W; pr < (nb + 1) / 2; pr += W) {  // (wave 1 takes pair 0: wave 0 is busy with phase D)
      int const k = 2 * pr;
      bool const have1 = k + 1 < nb;
      const float *pm = FO + k * olen, *cm = pm + olen, *nm = cm + olen;
      float const p0 = pm[lane], p1 = pm[lane + 64], c0 = cm[lane], c1 = cm[lane + 64];
      float const n0 = have1 ? nm[lane] : 0.f, n1 = have1 ? nm[lane + 64] : 0.f;
      float *aud0 = pl.audio + ((size_t)c * g.max_blocks + b0 + k) * (2 * (size_t)olen);
      float *aud1 = aud0 + 2 * olen;
      if (flat) {  // fm.c:164-172: no filter, no gain
        aud0[lane] = c0;
        aud0[lane + 64] = c1;
        if (have1) {
          aud1[lane] = n0;
          aud1[lane + 64] = n1;
        }
        continue;
      }
      float2 y2, y3;
      af.run(make_float2(p0, c0), make_float2(p1, c1), make_float2(c0, n0), make_float2(c1, n1), y2, y3);
      aud0[lane] = y2.x * gain;  // fm.c:169-170
      aud0[lane + 64] = y3.x * gain;
      if (have1) {
        aud1[lane] = y2.y * gain;
        aud1[lane + 64] = y3.y * gain;
      }
    }
    __syncthreads();
    // the chunk's last block precedes the next chunk (and, after the last chunk, the next call: filter.c:164)
    if (threadIdx.x < olen) {
      float const v = FO[nb * olen + threadIdx.x];
      FO[threadIdx.x] = v;
      if (b0 + 64 >= nblocks) hist_out[(size_t)c * olen + threadIdx.x] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ch.n0[c] = c_n0;
    ch.fm_state[c] = c_state;
    ch.lastaudio[c] = c_la;
    ch.sq_count[c] = c_sq;
    ch.foffset[c] = c_foff;
    ch.pdev[c] = c_pdev;
  }
}

// AM: envelope, carrier removal, hang AGC -- a strictly sequential recurrence per channel
// (am.c:55-75), so one lane per channel and channels across lanes.
__global__ void k_demod_am(Geom g, ChanDev ch, Planes pl, const int *__restrict__ list, int nchan, int nblocks,
                           int compute_n0) {
  int const t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchan) return;
  int const c = list[t];
  int const olen = g.olen;
  float const headroom = ch.headroom[c], recovery = ch.recovery[c];
  int const hangmax = ch.hangmax[c];
  float gain = ch.gain[c], dc = ch.dc[c];
  int hang = ch.hang[c];
  for (int b = 0; b < nblocks; b++) {
    const float2 *in = pl.filt + ((size_t)c * g.max_blocks + b) * olen;
    float *aud = pl.audio + ((size_t)c * g.max_blocks + b) * (2 * (size_t)olen);
    float signal = 0;
    for (int n = 0; n < olen; n++) {
      float const sq = cnrm(in[n]);
      signal += sq;
      float const samp = sqrtf(sq);
      dc += 0.0001f * (samp - dc);  // am.c:34,62
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
      aud[n] = (samp - dc) * gain;
    }
    kq_chan_status st;
    status_common(st, g, ch, pl, c, b, compute_n0, .001);
    st.bb_power = signal / (2 * olen);  // am.c:78
    st.snr = 0;
    st.foffset = 0;
    st.pdeviation = 0;
    st.agc_gain = gain;
    st.squelch_count = 0;
    st.hangcount = hang;
    st.blanked = 0;
    st.nout = olen;
    pl.status[(size_t)c * g.max_blocks + b] = st;
  }
  ch.gain[c] = gain;
  ch.dc[c] = dc;
  ch.hang[c] = hang;
}

// Linear (SSB/CW/IQ/ISB without carrier PLL): hang AGC on |s|, optional shift NCO, mono = Re or
// stereo = I/Q (linear.c:251-300).
__global__ void k_demod_linear(Geom g, ChanDev ch, Planes pl, const int *__restrict__ list, int nchan, int nblocks,
                               int compute_n0) {
  int const t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchan) return;
  int const c = list[t];
  int const olen = g.olen;
  float const headroom = ch.headroom[c], recovery = ch.recovery[c];
  int const hangmax = ch.hangmax[c];
  bool const stereo = (ch.flags[c] & FLAG_STEREO) != 0;
  double const sh_ph = ch.sh_phase[c], sh_f = ch.sh_freq[c];
  float gain = ch.gain[c];
  int hang = ch.hang[c];
  for (int b = 0; b < nblocks; b++) {
    const float2 *in = pl.filt + ((size_t)c * g.max_blocks + b) * olen;
    float *aud = pl.audio + ((size_t)c * g.max_blocks + b) * (2 * (size_t)olen);
    float signal = 0, noise = 0;
    for (int n = 0; n < olen; n++) {
      float2 s = in[n];
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
      if (sh_f != 0.0) {  // linear.c:283-289
        double const j = (double)b * olen + n;
        s = cmul(s, phasor_turns(sh_ph + sh_f * j));
      }
      if (stereo) {
        aud[2 * n] = s.x;
        aud[2 * n + 1] = s.y;
      } else {
        aud[n] = s.x;
      }
    }
    kq_chan_status st;
    status_common(st, g, ch, pl, c, b, compute_n0, .001);
    st.bb_power = (signal + noise) / (2 * olen);  // linear.c:302
    st.snr = NAN;                                 // linear.c:309
    st.foffset = 0;
    st.pdeviation = 0;
    st.agc_gain = gain;
    st.squelch_count = 0;
    st.hangcount = hang;
    st.blanked = 0;
    st.nout = stereo ? 2 * olen : olen;
    pl.status[(size_t)c * g.max_blocks + b] = st;
  }
  ch.gain[c] = gain;
  ch.hang[c] = hang;
}

// PL tone tracker (fm.c:236-277): per FM channel, blocks in sequence: append the PL filter output to the
// 16384-sample ring; after every >= 512 new samples transform the ring (in storage order, as the reference
// does) and pick the peak bin.  One workgroup per channel; the ring transform runs in LDS (128 KiB).
__global__ void __launch_bounds__(1024) k_pl_track(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ tw,
                                                   const int *__restrict__ list, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  __shared__ float red_e[16];
  __shared__ float red_p[16];
  __shared__ int red_i[16];
  constexpr int FS = 16384;  // (1 << 19) / 32, fm.c:225
  int const c = list[blockIdx.x];
  float *ring = ch.plring + (size_t)c * FS;
  int ptr = ch.pl_ptr[c], last = ch.pl_last[c];
  float plfreq = ch.plfreq[c];
  float const pl_samprate = g.dsamprate / 32.f;
  // The blocks between two transforms are taken together: their PL samples (contiguous in plout) go into the ring in one
  // sweep and their status records in another -- block by block this was a chain of 64 tiny dependent steps per call.
  for (int b = 0; b < nblocks;) {
    int need = (512 - last + g.pl_l - 1) / g.pl_l;  // blocks until fm.c:251's count is reached
    if (need < 1) need = 1;
    int const nb = min(need, nblocks - b);
    const float *src = pl.plout + ((size_t)c * g.max_blocks + b) * g.pl_l;
    for (int i = threadIdx.x; i < nb * g.pl_l; i += blockDim.x) ring[(ptr + i) & (FS - 1)] = src[i];
    ptr = (ptr + nb * g.pl_l) & (FS - 1);
    last += nb * g.pl_l;
    float const before = plfreq;  // what the blocks in front of the one that completes the count report
    if (last >= 512) {  // fm.c:251
      last = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < FS; i += blockDim.x) lds[bitrev((unsigned)i, 14)] = make_float2(ring[i], 0.f);
      lds_fft<-1>(lds, 14, tw, g.tw_log2);
      float tot = 0, pe = 0;
      int pb = -1;
      for (int n = 1 + threadIdx.x; n < FS / 2; n += blockDim.x) {  // skip DC (fm.c:260)
        float const e = cnrm(lds[n]);
        tot += e;
        if (e > pe) {
          pe = e;
          pb = n;
        }
      }
      // block reduction: total energy, and the first bin holding the maximum energy
      tot = wave_sum(tot);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float const oe = __shfl_xor(pe, o, 64);
        int const ob = __shfl_xor(pb, o, 64);
        if (oe > pe || (oe == pe && ob >= 0 && (pb < 0 || ob < pb))) {
          pe = oe;
          pb = ob;
        }
      }
      int const w = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) {
        red_e[w] = tot;
        red_p[w] = pe;
        red_i[w] = pb;
      }
      __syncthreads();
      tot = 0;
      pe = 0;
      pb = -1;
      for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
        tot += red_e[k];
        if (red_p[k] > pe || (red_p[k] == pe && red_i[k] >= 0 && (pb < 0 || red_i[k] < pb))) {
          pe = red_p[k];
          pb = red_i[k];
        }
      }
      if (pb > 0 && pe > 0.01f * tot) {  // fm.c:271-276
        float const f = (float)pb * pl_samprate / FS;
        if (f > 67 && f < 255) plfreq = f;
      } else {
        plfreq = NAN;
      }
      __syncthreads();
    }
    for (int j = threadIdx.x; j < nb; j += blockDim.x)
      pl.status[(size_t)c * g.max_blocks + b + j].plfreq = j == nb - 1 ? plfreq : before;
    b += nb;
  }
  if (threadIdx.x == 0) {
    ch.pl_ptr[c] = ptr;
    ch.pl_last[c] = last;
    ch.plfreq[c] = plfreq;
  }
}

void launch_pl_track(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *tw, const int *list_fm,
                     int n_fm, int nblocks) {
  if (n_fm <= 0 || g.pl_n <= 0) return;
  ensure_dynamic_lds((const void *)k_pl_track, (size_t)(16384 * 8));
  hipLaunchKernelGGL(k_pl_track, dim3(n_fm), dim3(1024), 16384 * 8, s, g, ch, pl, tw, list_fm, nblocks);
}

// dynamic LDS of the generic FM demodulator: k_demod_fm's samples and discriminator outputs per wave; k_fm_audio's
// audio master, its transform, the PL slave and the twiddles
// waves per channel of k_demod_fm: as many as fit 96 KiB of LDS at 12 bytes a sample, at most 16 or one per block
static int fm_disc_waves(const Geom &g, int nblocks) {
  int const fit = (int)((96u * 1024u) / (12u * (unsigned)g.olen));
  return std::max(1, std::min({16, fit, nblocks}));
}
static size_t fm_disc_lds_bytes(const Geom &g, int waves = 1) { return (size_t)g.olen * 12 * waves; }
static size_t fm_audio_lds_bytes(const Geom &g) {
  return (size_t)g.Ndec * (8 + 4) + (size_t)g.pl_n * 8 + (size_t)(g.dNdec.log2n < 0 ? g.Ndec : g.Ndec / 2) * 8;
}
size_t demod_fm_lds_bytes(const Geom &g) { return std::max(fm_disc_lds_bytes(g), fm_audio_lds_bytes(g)); }

void launch_demods(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *tw, const int *list_fm,
                   int n_fm, const int *list_am, int n_am, const int *list_lin, int n_lin, int nblocks, int compute_n0,
                   float *fmout, const float *fm_hist_in, float *fm_hist_out) {
  // cfg 2's geometry without the PL measurement: one fused launch (KQ_FM_FUSED=0: the two-kernel form, A/B switch)
  static bool const fused_off = getenv("KQ_FM_FUSED") && atoi(getenv("KQ_FM_FUSED")) == 0;
  bool const fused = g.Ndec == 256 && g.olen == 128 && g.Mdec == 129 && g.pl_n == 0 && !fused_off;
  if (n_fm > 0 && fused) {
    // (8 against 16 waves per channel, tools/ab_fm256.sh, three alternating rounds on one box: 35.3 / 35.2 / 35.7 us against
    //  35.3 / 35.6 / 35.4 -- the launch is bound by its vector instruction count, 55 000 issue cycles per SIMD either way)
    static int const waves = getenv("KQ_FM256_WAVES") ? atoi(getenv("KQ_FM256_WAVES")) : 8;
    if (waves != 16)
      hipLaunchKernelGGL(k_demod_fm256<8>, dim3(n_fm), dim3(512), 0, s, g, ch, pl, fm_hist_in, fm_hist_out, list_fm, nblocks,
                         compute_n0);
    else
      hipLaunchKernelGGL(k_demod_fm256<16>, dim3(n_fm), dim3(1024), 0, s, g, ch, pl, fm_hist_in, fm_hist_out, list_fm, nblocks,
                         compute_n0);
  } else if (n_fm > 0) {
    int const waves = fm_disc_waves(g, nblocks);
    size_t const lds_a = fm_disc_lds_bytes(g, waves), lds_b = fm_audio_lds_bytes(g);
    ensure_dynamic_lds((const void *)k_demod_fm, lds_a);
    ensure_dynamic_lds((const void *)k_fm_audio, lds_b);
    hipLaunchKernelGGL(k_demod_fm, dim3(n_fm), dim3(64 * waves), lds_a, s, g, ch, pl, fmout, list_fm, nblocks, compute_n0);
    if (g.Ndec == 256 && g.olen == 128 && g.Mdec == 129 && (g.pl_n == 0 || (g.pl_n == 8 && g.pl_l == 4)))  // cfg 2's geometry: registers only
      hipLaunchKernelGGL(k_fm_audio256, dim3(n_fm, (nblocks + 1) / 2), dim3(64), 0, s, g, ch, pl, fmout, fm_hist_in, fm_hist_out,
                         list_fm, nblocks);
    else
    {
      // one wave per block up to a 512-point audio master; four from there on (the transform's passes are loops over the
      // workgroup with a barrier each: tools/bench_mixed.py, KQ_FM_AUDIO_THREADS)
      static int const forced = getenv("KQ_FM_AUDIO_THREADS") ? atoi(getenv("KQ_FM_AUDIO_THREADS")) : 0;
      int const thr = forced > 0 ? forced : g.Ndec >= 1024 ? 256 : 64;
      hipLaunchKernelGGL(k_fm_audio, dim3(n_fm, nblocks), dim3(thr), lds_b, s, g, ch, pl, tw, fmout, fm_hist_in, fm_hist_out,
                         list_fm, nblocks);
    }
  }
  if (n_am > 0)
    hipLaunchKernelGGL(k_demod_am, dim3((n_am + 63) / 64), dim3(64), 0, s, g, ch, pl, list_am, n_am, nblocks, compute_n0);
  if (n_lin > 0)
    hipLaunchKernelGGL(k_demod_linear, dim3((n_lin + 63) / 64), dim3(64), 0, s, g, ch, pl, list_lin, n_lin, nblocks,
                       compute_n0);
}

// ---------------------------------------------------------------- PCM output stage
// float -> clipped int16 in network byte order (audio.c:22-28, htons at audio.c:48,98) and the all-zero test per
// 480-word chunk that decides whether the reference sends the packet (audio.c:49,99,105).  One wave per channel-block.
__global__ void __launch_bounds__(64) k_pcm(Geom g, Planes pl, short *__restrict__ pcm, unsigned *__restrict__ mask,
                                            const int *__restrict__ chan_list) {
  int const c = chan_list ? chan_list[blockIdx.x] : (int)blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  size_t const cb = (size_t)c * g.max_blocks + b;
  int const nout = pl.status[cb].nout;
  const float *a = pl.audio + cb * 2 * (size_t)g.olen;
  short *o = pcm + cb * 2 * (size_t)g.olen;
  unsigned m = 0;
  int chunk_id = 0;
  for (int base = 0; base < nout; base += 480, chunk_id++) {
    int const len = min(480, nout - base);
    int any = 0;
    for (int i = lane; i < len; i += 64) {
      float const x = a[base + i];
      int v;
      if (x >= 1.0f)
        v = 32767;
      else if (x <= -1.0f)
        v = -32768;
      else
        v = (int)(32767.f * x);  // truncation, as the (short) cast of audio.c:27
      unsigned const h = (unsigned)v & 0xffffu;
      unsigned const be = ((h << 8) | (h >> 8)) & 0xffffu;
      o[base + i] = (short)be;
      any |= (int)be;
    }
    if (__ballot(any != 0) == 0ull) m |= 1u << chunk_id;
  }
  if (lane == 0) mask[cb] = m;
}

void launch_pcm(hipStream_t s, const Geom &g, const Planes &pl, short *pcm, unsigned *mask, int nchan, int nblocks,
                const int *chan_list) {
  hipLaunchKernelGGL(k_pcm, dim3(nchan, nblocks), dim3(64), 0, s, g, pl, pcm, mask, chan_list);
}

// ---------------------------------------------------------------- single transforms (compat surface)
__global__ void k_fft_single(const float2 *__restrict__ in, float2 *__restrict__ out, FftDim d, int sign,
                             const float2 *__restrict__ tw, int tw_log2) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const n = d.n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[fft_pos((unsigned)i, d)] = in[i];
  if (sign < 0)
    fft_any<-1>(lds, d, tw, tw_log2);
  else
    fft_any<+1>(lds, d, tw, tw_log2);
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = lds[i];
}

// n: a power of two (tw / tw_log2: the half-circle table lds_fft reads) or any 2^a 3^b 5^c 7^d the caller has a plan for
void launch_fft_single(hipStream_t s, const float2 *in, float2 *out, const FftDim &d, int sign, const float2 *tw, int tw_log2) {
  size_t const lds_bytes = sizeof(float2) * (size_t)d.n;
  ensure_dynamic_lds((const void *)k_fft_single, lds_bytes);
  int const threads = d.n >= 4096 ? 1024 : 256;
  hipLaunchKernelGGL(k_fft_single, dim3(1), dim3(threads), lds_bytes, s, in, out, d, sign, tw, tw_log2);
}

// Transforms beyond the 16384 points one workgroup holds in LDS (compat masters up to 2^22 points): N = Na * Nb through
// global memory.  With n = Nb n1 + n2 and k = k1 + Na k2:
//   X[k1 + Na k2] = sum_n2 W_Nb^{n2 k2} W_N^{n2 k1} sum_n1 x[Nb n1 + n2] W_Na^{n1 k1}
// k_fft_cols: one workgroup per n2 runs the Na-point transform over n1, applies W_N^{n2 k1}, stores tmp[k1][n2];
// k_fft_rows: one workgroup per k1 runs the Nb-point transform over n2 and scatters to out[k1 + Na k2].
// (powers of two: da / db carry log2 and the W_N^{n2 k1} factors come from the half-circle table; otherwise twN is the
// full-circle table of N points and n2 k1 < N indexes it directly)
__global__ void k_fft_cols(const float2 *__restrict__ in, float2 *__restrict__ tmp, FftDim da, FftDim db, int sign,
                           const float2 *__restrict__ tw, int tw_log2, const float2 *__restrict__ twN) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const na = da.n, nb = db.n, n2 = blockIdx.x;
  for (int i = threadIdx.x; i < na; i += blockDim.x) lds[fft_pos((unsigned)i, da)] = in[(size_t)nb * i + n2];
  if (sign < 0)
    fft_any<-1>(lds, da, tw, tw_log2);
  else
    fft_any<+1>(lds, da, tw, tw_log2);
  if (twN) {
    for (int k1 = threadIdx.x; k1 < na; k1 += blockDim.x) {
      float2 w = twN[(size_t)n2 * k1];
      if (sign > 0) w.y = -w.y;
      tmp[(size_t)k1 * nb + n2] = cmul(lds[k1], w);
    }
    return;
  }
  unsigned const half = 1u << (tw_log2 - 1), shift = (unsigned)(tw_log2 - da.log2n - db.log2n);
  for (int k1 = threadIdx.x; k1 < na; k1 += blockDim.x) {
    unsigned e = ((unsigned)n2 * (unsigned)k1) << shift;  // exponent on the table's period, < 2^tw_log2
    float2 w = tw[e & (half - 1)];
    if (e & half) w = make_float2(-w.x, -w.y);
    if (sign > 0) w.y = -w.y;
    tmp[(size_t)k1 * nb + n2] = cmul(lds[k1], w);
  }
}
__global__ void k_fft_rows(const float2 *__restrict__ tmp, float2 *__restrict__ out, FftDim da, FftDim db, int sign,
                           const float2 *__restrict__ tw, int tw_log2) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  int const na = da.n, nb = db.n, k1 = blockIdx.x;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) lds[fft_pos((unsigned)i, db)] = tmp[(size_t)k1 * nb + i];
  if (sign < 0)
    fft_any<-1>(lds, db, tw, tw_log2);
  else
    fft_any<+1>(lds, db, tw, tw_log2);
  for (int k2 = threadIdx.x; k2 < nb; k2 += blockDim.x) out[(size_t)k1 + (size_t)na * k2] = lds[k2];
}

// N beyond one LDS block: a power of two up to 2^22, or 2^a 3^b 5^c 7^d up to 65536 (the full-circle table's reach)
int launch_fft_large(hipStream_t s, const float2 *in, float2 *out, float2 *tmp, int N, int sign, const float2 *tw, int tw_log2) {
  bool ok = false;
  FftDim const dn = fft_dim(N, &ok);
  if (!ok) return -1;
  int na, nb;
  if (dn.log2n >= 0) {
    int const log2na = (dn.log2n + 1) / 2;
    na = 1 << log2na;
    nb = N >> log2na;
  } else {  // split the radices so that both sides fit one LDS block and stay even where they can
    na = 1;
    for (int k = 0; k < dn.nf; k++)
      if ((long long)na * na < N) na *= dn.f[k];
    nb = N / na;
    if (na > 16384 || nb > 16384) return -1;
  }
  bool oka = false, okb = false;
  FftDim const da = fft_dim(na, &oka), db = fft_dim(nb, &okb);
  if (!oka || !okb) return -1;
  size_t const lds_a = sizeof(float2) * (size_t)na, lds_b = sizeof(float2) * (size_t)nb;
  ensure_dynamic_lds((const void *)k_fft_cols, lds_a);
  ensure_dynamic_lds((const void *)k_fft_rows, lds_b);
  int const ta = na >= 1024 ? 256 : 64, tb = nb >= 1024 ? 256 : 64;
  hipLaunchKernelGGL(k_fft_cols, dim3((unsigned)nb), dim3(ta), lds_a, s, in, tmp, da, db, sign, tw, tw_log2,
                     dn.log2n >= 0 ? (const float2 *)nullptr : dn.twc);
  hipLaunchKernelGGL(k_fft_rows, dim3((unsigned)na), dim3(tb), lds_b, s, tmp, out, da, db, sign, tw, tw_log2);
  return 0;
}

// compute_n0 (radio.c:383-425) on one resident master spectrum (the compat surface and the demodulator entry points)
__global__ void k_n0_single(const float2 *__restrict__ X, int N, int samprate, float low, float high, float *__restrict__ out) {
  __shared__ float red_f[16];
  __shared__ int red_i[16];
  float avg = INFINITY;
  for (int iter = 0; iter < 2; iter++) {
    float acc = 0;
    int bins = 0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      int const k = (n <= N / 2) ? n : n - N;
      int const prod = (int)((unsigned)k * (unsigned)samprate);  // the reference's 32-bit product, wrap included
      float const f = (float)prod / N;
      if (f >= low && f <= high) continue;
      float const p = cnrm(X[n]);
      if (p < avg * 2) {
        acc += p;
        bins++;
      }
    }
    block_sum_fi(acc, bins, red_f, red_i);
    avg = acc / bins;
  }
  if (threadIdx.x == 0) *out = (float)(avg / (2.0 * N * samprate));
}

void launch_n0_single(hipStream_t s, const float2 *fdomain, int N, int samprate, float low, float high, float *out) {
  hipLaunchKernelGGL(k_n0_single, dim3(1), dim3(1024), 0, s, fdomain, N, samprate, low, high, out);
}

// One slave execution on a resident master spectrum: all four in/out type combinations of
// filter.c:206-250.  out: N_dec float2 (complex out) or N_dec floats packed in float2[N_dec/2] (real out).
__global__ void k_slave_single(const float2 *__restrict__ X, const float2 *__restrict__ H, float2 *__restrict__ out, int N,
                               FftDim dd, int in_real, int out_type, const float2 *__restrict__ tw, int tw_log2) {
  extern __shared__ __attribute__((aligned(16))) float2 G[];
  int const Ndec = dd.n;
  bool const out_real = out_type == 3;
  for (int p = threadIdx.x; p <= Ndec / 2; p += blockDim.x) {
    float2 gp = cmul(H[p], X[p]);
    bool const interior = p > 0 && p < Ndec / 2;
    int const k = Ndec - p;
    float2 gn = make_float2(0, 0);
    if (interior) {
      if (in_real) {
        if (!out_real) gn = cmul(H[k], cconj(X[p]));  // filter.c:214-216
      } else if (!out_real) {
        gn = cmul(H[k], X[N - p]);  // filter.c:225-227
      } else {
        gp = cadd(gp, cconj(cmul(H[k], X[N - p])));  // filter.c:232-234
      }
      if (out_type == 2) {  // CROSS_CONJ, filter.c:239-249
        float2 const pos = gp, neg = gn;
        gp = cadd(pos, cconj(neg));
        gn = csub(neg, cconj(pos));
      }
      if (out_real) gn = cconj(gp);  // c2r Hermitian extension
      G[fft_pos((unsigned)k, dd)] = gn;
    } else if (out_real) {
      gp.y = 0;  // c2r ignores the imaginary parts of DC and Nyquist
    }
    G[fft_pos((unsigned)p, dd)] = gp;
  }
  fft_any<+1>(G, dd, tw, tw_log2);
  if (out_real) {
    float *o = reinterpret_cast<float *>(out);
    for (int i = threadIdx.x; i < Ndec; i += blockDim.x) o[i] = G[i].x;
  } else {
    for (int i = threadIdx.x; i < Ndec; i += blockDim.x) out[i] = G[i];
  }
}

// The bank's slave on a spectrum handed in from outside (kq_bank_process_spectrum): COMPLEX in, COMPLEX or CROSS_CONJ out,
// the last `olen` of the N_dec outputs (filter.c:131) to `out`.
__global__ void k_slave_bank(const float2 *__restrict__ X, const float2 *__restrict__ H, float2 *__restrict__ out, int N, FftDim dd,
                             int olen, int out_type, const float2 *__restrict__ tw, int tw_log2) {
  extern __shared__ __attribute__((aligned(16))) float2 G[];
  int const Ndec = dd.n;
  for (int p = threadIdx.x; p <= Ndec / 2; p += blockDim.x) {
    float2 gp = cmul(H[p], X[p]);
    if (p > 0 && p < Ndec / 2) {
      int const k = Ndec - p;
      float2 gn = cmul(H[k], X[N - p]);  // filter.c:225-227
      if (out_type == 2) {               // CROSS_CONJ, filter.c:239-249
        float2 const pos = gp, neg = gn;
        gp = cadd(pos, cconj(neg));
        gn = csub(neg, cconj(pos));
      }
      G[fft_pos((unsigned)k, dd)] = gn;
    }
    G[fft_pos((unsigned)p, dd)] = gp;
  }
  fft_any<+1>(G, dd, tw, tw_log2);
  for (int i = threadIdx.x; i < olen; i += blockDim.x) out[i] = G[Ndec - olen + i];
}

void launch_slave_bank(hipStream_t s, const float2 *fdomain, const float2 *resp, float2 *out, int N, int Ndec, int olen,
                       int out_type, const float2 *tw, int tw_log2) {
  bool ok = false;
  FftDim const dd = fft_dim(Ndec, &ok);  // (cached: the bank / the compat slave made the plan when it was created)
  if (!ok) return;
  size_t const lds_bytes = sizeof(float2) * (size_t)Ndec;
  ensure_dynamic_lds((const void *)k_slave_bank, lds_bytes);
  int const threads = Ndec >= 4096 ? 1024 : 256;
  hipLaunchKernelGGL(k_slave_bank, dim3(1), dim3(threads), lds_bytes, s, fdomain, resp, out, N, dd, olen, out_type, tw, tw_log2);
}

void launch_slave_single(hipStream_t s, const float2 *fdomain, const float2 *resp, float2 *out, int N, int Ndec, int in_real,
                         int out_type, const float2 *tw, int tw_log2) {
  bool ok = false;
  FftDim const dd = fft_dim(Ndec, &ok);
  if (!ok) return;
  size_t const lds_bytes = sizeof(float2) * (size_t)Ndec;
  ensure_dynamic_lds((const void *)k_slave_single, lds_bytes);
  int const threads = Ndec >= 4096 ? 1024 : 256;
  hipLaunchKernelGGL(k_slave_single, dim3(1), dim3(threads), lds_bytes, s, fdomain, resp, out, N, dd, in_real, out_type, tw, tw_log2);
}

}  // namespace kq
