// kq_afsk.hip -- AFSK-1200 / HDLC packet decoder for a bank of audio sessions on gfx950.
//
// Replaces, per session, what packet.c does between the PCM words of an RTP packet and a decoded AX.25 frame:
//   packet.c:201-212  samples into the REAL master filter (AL = 1000 new samples, AM = 1049 taps, N = 2048)
//   packet.c:272-273  slave, COMPLEX out, decimate 1, set_filter(+100 Hz, +4000 Hz, 3.0): analytic band-limited signal
//   packet.c:276-284  mark / space replica oscillators at -1200 / -2200 Hz
//   packet.c:302-410  correlators, on-time and mid-bit integrators, Gardner-style bit clock, NRZI, HDLC deframing
//   ax25.c:138-156    crc_good
//
// One workgroup per session walks the session's blocks in order.  The filter (forward transform of the real window,
// response multiply, inverse transform) and the two spin-down products are data parallel across the workgroup; the
// bit clock and the deframer are the reference's serial recurrence and run on one lane from LDS.  The two replica
// oscillators are exact periodic tables (periods 40 and 240 samples at 48 kHz) instead of the recurrence of osc.c.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ka9q_hip.h"
#include "kq_design.hpp"
#include "kq_ldsfft.hpp"

void kq_internal_set_error(const char *fmt, ...);

namespace {

constexpr int AL = 1000, AM = 1049, AN = 2048, LOG2AN = 11;  // packet.c:41-45
constexpr int SAMPPBIT = 40;                                  // packet.c:48
constexpr int FRAME_MAX = 1024;                               // packet.c:294
constexpr int kThreads = 256;
constexpr int MARK_PERIOD = 40, SPACE_PERIOD = 240;           // 48000/1200, 48000/gcd(48000,2200)

// Carried decoder state of one session (packet.c:286-299)
struct AfskState {
  float2 mark_accum, space_accum, mark_offset_accum, space_offset_accum;
  float last_val, mid_val;
  int symphase, frame_bit, flagsync, ones;
  int osc_phase;  // samples since start modulo SPACE_PERIOD (a multiple of MARK_PERIOD)
  int decoded_packets;
};

struct AfskArgs {
  const void *src;       // [nsessions][stride] new samples
  int format;            // kq_pcm_format
  size_t stride;         // elements between sessions in src
  size_t nnew;           // new samples per session
  int fill;              // samples already pending per session
  float *pend;           // [S][AL] pending partial block
  float *hist;           // [S][AM-1] filter history
  const float2 *resp;    // [AN] response
  const float2 *tw;      // twiddles, period AN
  const double2 *mark_tab, *space_tab;
  AfskState *state;      // [S]
  unsigned char *hdlc;   // [S][FRAME_MAX] frame under assembly
  unsigned char *frames; // [S][max_frames][FRAME_MAX]
  int *frame_len;        // [S][max_frames]
  int *nframes;          // [S]
  int *dropped;          // [S] frames lost because the arena was full
  int max_frames;
  float2 *last_out;      // [S][AL] filter output of the last block (parity / diagnostics)
};

__device__ __forceinline__ float load_sample(const void *src, int format, size_t idx) {
  if (format == KQ_PCM_S16BE) {
    // packet.c:207: ntohs(*samples++) * SCALE -- ntohs() is unsigned, negative words arrive as 32768..65535
    const unsigned char *p = reinterpret_cast<const unsigned char *>(src) + 2 * idx;
    unsigned const v = ((unsigned)p[0] << 8) | p[1];
    return v * (1.f / 32768.f);
  }
  return reinterpret_cast<const float *>(src)[idx];
}

__device__ int crc_good(const unsigned char *frame, int length) {
  // ax25.c:138-156
  unsigned crc = 0xffff;
  while (length-- > 0) {
    unsigned byte = *frame++;
    for (int i = 0; i < 8; i++) {
      unsigned const feedback = ((crc ^ byte) & 1) ? 0x8408u : 0u;
      crc = (crc >> 1) ^ feedback;
      byte >>= 1;
    }
  }
  return crc == 0xf0b8;
}

__global__ __launch_bounds__(kThreads) void k_afsk(AfskArgs a) {
  __shared__ float2 buf[AN];
  __shared__ float2 sm[AL], sp[AL];
  __shared__ float hist[AM - 1];
  __shared__ unsigned char frame[FRAME_MAX];
  int const s = blockIdx.x, tid = threadIdx.x;
  size_t const total = (size_t)a.fill + a.nnew;
  int const nblk = (int)(total / AL);
  const char *src = reinterpret_cast<const char *>(a.src);
  size_t const esize = a.format == KQ_PCM_S16BE ? 2 : 4;
  const void *my_src = src + (size_t)s * a.stride * esize;
  float *my_pend = a.pend + (size_t)s * AL;
  auto stream_sample = [&](size_t j) -> float {  // j-th sample of pending ++ new
    return j < (size_t)a.fill ? my_pend[j] : load_sample(my_src, a.format, j - a.fill);
  };

  if (nblk > 0) {
    for (int i = tid; i < AM - 1; i += kThreads) hist[i] = a.hist[(size_t)s * (AM - 1) + i];
    for (int i = tid; i < FRAME_MAX; i += kThreads) frame[i] = a.hdlc[(size_t)s * FRAME_MAX + i];
  }
  AfskState st = a.state[s];
  __syncthreads();

  for (int b = 0; b < nblk; b++) {
    // window [history | AL new] in bit-reversed order, imaginary part zero (REAL master, filter.c:163-167)
    for (int i = tid; i < AN; i += kThreads) {
      float const v = i < AM - 1 ? hist[i] : stream_sample((size_t)b * AL + (i - (AM - 1)));
      buf[kq::bitrev(i, LOG2AN)] = make_float2(v, 0.f);
    }
    __syncthreads();
    // next block's history = last AM-1 samples of this window (filter.c:169-170)
    float hnew[(AM - 1 + kThreads - 1) / kThreads];
#pragma unroll
    for (int k = 0; k < (AM - 1 + kThreads - 1) / kThreads; k++) {
      int const i = tid + k * kThreads;
      if (i < AM - 1) hnew[k] = buf[kq::bitrev(i + AL, LOG2AN)].x;
    }
    kq::lds_fft<-1>(buf, LOG2AN, a.tw, LOG2AN);
#pragma unroll
    for (int k = 0; k < (AM - 1 + kThreads - 1) / kThreads; k++) {
      int const i = tid + k * kThreads;
      if (i < AM - 1) hist[i] = hnew[k];
    }
    // response multiply (filter.c:206-216; the spectrum of a real window is conjugate symmetric, so the
    // "negative bins from the conjugate of the positive ones" rule is the plain product), stored bit-reversed
    for (int k = tid; k < AN; k += kThreads) {
      int const r = kq::bitrev(k, LOG2AN);
      if (k <= r) {
        float2 const xk = buf[k], xr = buf[r];
        float2 const gk = kq::cmul(a.resp[k], xk), gr = kq::cmul(a.resp[r], xr);
        buf[r] = gk;
        buf[k] = gr;
      }
    }
    kq::lds_fft<+1>(buf, LOG2AN, a.tw, LOG2AN);  // filter.c:250; output.c = last AL samples
    // spin down by the mark and space replicas (packet.c:309,313): float complex x double complex, rounded to float
    for (int n = tid; n < AL; n += kThreads) {
      float2 const y = buf[AM - 1 + n];
      int const ph = (st.osc_phase + n) % SPACE_PERIOD;
      double2 const m = a.mark_tab[ph % MARK_PERIOD], q = a.space_tab[ph];
      double const yr = y.x, yi = y.y;
      sm[n] = make_float2((float)(yr * m.x - yi * m.y), (float)(yr * m.y + yi * m.x));
      sp[n] = make_float2((float)(yr * q.x - yi * q.y), (float)(yr * q.y + yi * q.x));
      if (b == nblk - 1) a.last_out[(size_t)s * AL + n] = y;
    }
    __syncthreads();
    st.osc_phase = (st.osc_phase + AL) % SPACE_PERIOD;

    if (tid == 0) {
      // packet.c:304-409, one lane: integrators, bit clock, NRZI, HDLC
      for (int n = 0; n < AL; n++) {
        float2 const m = sm[n], q = sp[n];
        st.mark_accum.x += m.x;
        st.mark_accum.y += m.y;
        st.mark_offset_accum.x += m.x;
        st.mark_offset_accum.y += m.y;
        st.space_accum.x += q.x;
        st.space_accum.y += q.y;
        st.space_offset_accum.x += q.x;
        st.space_offset_accum.y += q.y;
        if (++st.symphase == SAMPPBIT / 2) {
          st.mid_val = kq::cnrm(st.mark_offset_accum) - kq::cnrm(st.space_offset_accum);
          st.mark_offset_accum = st.space_offset_accum = make_float2(0.f, 0.f);
        }
        if (st.symphase < SAMPPBIT) continue;
        st.symphase = 0;
        float const cur_val = kq::cnrm(st.mark_accum) - kq::cnrm(st.space_accum);
        st.mark_accum = st.space_accum = make_float2(0.f, 0.f);
        if (cur_val * st.last_val < 0) {
          st.symphase += ((cur_val - st.last_val) * st.mid_val) > 0 ? +1 : -1;
          if (st.ones == 6) {
            if (st.flagsync) {
              st.frame_bit -= 7;
              int const bytes = st.frame_bit / 8;
              if (bytes > 0 && bytes <= FRAME_MAX && crc_good(frame, bytes)) {
                int const slot = a.nframes[s];
                if (slot < a.max_frames) {
                  unsigned char *dst = a.frames + ((size_t)s * a.max_frames + slot) * FRAME_MAX;
                  for (int i = 0; i < bytes; i++) dst[i] = frame[i];
                  a.frame_len[(size_t)s * a.max_frames + slot] = bytes;
                  a.nframes[s] = slot + 1;
                } else {
                  a.dropped[s]++;
                }
                st.decoded_packets++;
              }
            }
            for (int i = 0; i < FRAME_MAX; i++) frame[i] = 0;
            st.frame_bit = 0;
            st.flagsync = 1;
          } else if (st.ones < 5) {
            if (st.flagsync) st.frame_bit++;
          }
          st.ones = 0;
        } else {
          if (++st.ones == 7) {
            for (int i = 0; i < FRAME_MAX; i++) frame[i] = 0;
            st.frame_bit = 0;
            st.flagsync = 0;
          } else if (st.flagsync) {
            // packet.c:401 has no bound on hdlc_frame[1024]; past it the reference's write is undefined, here dropped
            if (st.frame_bit >= 0 && st.frame_bit < 8 * FRAME_MAX) frame[st.frame_bit / 8] |= 1 << (st.frame_bit % 8);
            st.frame_bit++;
          }
        }
        st.last_val = cur_val;
      }
    }
    __syncthreads();
  }

  // carry: leftover samples, history, frame under assembly, decoder state
  size_t const used = (size_t)nblk * AL;
  int const left = (int)(total - used);
  float keep[(AL + kThreads - 1) / kThreads];
#pragma unroll
  for (int k = 0; k < (AL + kThreads - 1) / kThreads; k++) {
    int const i = tid + k * kThreads;
    if (i < left) keep[k] = stream_sample(used + i);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < (AL + kThreads - 1) / kThreads; k++) {
    int const i = tid + k * kThreads;
    if (i < left) my_pend[i] = keep[k];
  }
  if (nblk > 0) {
    for (int i = tid; i < AM - 1; i += kThreads) a.hist[(size_t)s * (AM - 1) + i] = hist[i];
    for (int i = tid; i < FRAME_MAX; i += kThreads) a.hdlc[(size_t)s * FRAME_MAX + i] = frame[i];
    if (tid == 0) a.state[s] = st;
  }
}

}  // namespace

struct kq_afsk_bank {
  kq_afsk_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int fill = 0;
  float *pend = nullptr, *hist = nullptr;
  float2 *resp = nullptr, *tw = nullptr, *last_out = nullptr;
  double2 *mark_tab = nullptr, *space_tab = nullptr;
  AfskState *state = nullptr;
  unsigned char *hdlc = nullptr, *frames = nullptr;
  int *frame_len = nullptr, *nframes = nullptr, *dropped = nullptr;
  void *staging = nullptr;
  size_t staging_bytes = 0;
  uint64_t blocks = 0;
};

#define AF_TRY(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      kq_internal_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                                      \
    }                                                                                                 \
  } while (0)

static int afsk_alloc(kq_afsk_bank *b) {
  kq_afsk_config const &c = b->cfg;
  size_t const S = c.max_sessions;
  kq::DeviceScope dev_scope_(c.device);  // the caller's current device is restored on return
  if (c.stream)
    b->stream = (hipStream_t)c.stream;
  else {
    AF_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    b->own_stream = true;
  }
  AF_TRY(hipMalloc(&b->pend, S * AL * sizeof(float)));
  AF_TRY(hipMalloc(&b->hist, S * (AM - 1) * sizeof(float)));
  AF_TRY(hipMalloc(&b->resp, AN * sizeof(float2)));
  AF_TRY(hipMalloc(&b->tw, (AN / 2) * sizeof(float2)));
  AF_TRY(hipMalloc(&b->last_out, S * AL * sizeof(float2)));
  AF_TRY(hipMalloc(&b->mark_tab, MARK_PERIOD * sizeof(double2)));
  AF_TRY(hipMalloc(&b->space_tab, SPACE_PERIOD * sizeof(double2)));
  AF_TRY(hipMalloc(&b->state, S * sizeof(AfskState)));
  AF_TRY(hipMalloc(&b->hdlc, S * FRAME_MAX));
  AF_TRY(hipMalloc(&b->frames, S * c.max_frames * FRAME_MAX));
  AF_TRY(hipMalloc(&b->frame_len, S * c.max_frames * sizeof(int)));
  AF_TRY(hipMalloc(&b->nframes, S * sizeof(int)));
  AF_TRY(hipMalloc(&b->dropped, S * sizeof(int)));
  AF_TRY(hipMemsetAsync(b->pend, 0, S * AL * sizeof(float), b->stream));
  AF_TRY(hipMemsetAsync(b->hist, 0, S * (AM - 1) * sizeof(float), b->stream));
  AF_TRY(hipMemsetAsync(b->last_out, 0, S * AL * sizeof(float2), b->stream));
  AF_TRY(hipMemsetAsync(b->state, 0, S * sizeof(AfskState), b->stream));
  AF_TRY(hipMemsetAsync(b->hdlc, 0, S * FRAME_MAX, b->stream));
  AF_TRY(hipMemsetAsync(b->nframes, 0, S * sizeof(int), b->stream));
  AF_TRY(hipMemsetAsync(b->dropped, 0, S * sizeof(int), b->stream));

  // packet.c:273 set_filter(filter, +100/Samprate, +4000/Samprate, 3.0) on a slave with decimate 1
  std::vector<kq::cfloat> const r = kq::design_response(AN, AL, AM, kq::FT_COMPLEX, 100.f / 48000.f, 4000.f / 48000.f, 3.0f);
  if ((int)r.size() != AN) {
    kq_internal_set_error("kq_afsk_create: response design returned %zu bins", r.size());
    return -1;
  }
  AF_TRY(hipMemcpyAsync(b->resp, r.data(), AN * sizeof(float2), hipMemcpyHostToDevice, b->stream));
  std::vector<float2> tw(AN / 2);
  for (int k = 0; k < AN / 2; k++) {
    double const ang = -2.0 * M_PI * k / AN;
    tw[k] = make_float2((float)cos(ang), (float)sin(ang));
  }
  AF_TRY(hipMemcpyAsync(b->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice, b->stream));
  // replica phasors exp(-j 2 pi f t / 48000), f = 1200 and 2200 (packet.c:279,284), exact periods 40 and 240
  std::vector<double2> mt(MARK_PERIOD), stb(SPACE_PERIOD);
  for (int t = 0; t < MARK_PERIOD; t++) {
    double const ang = -2.0 * M_PI * t / MARK_PERIOD;
    mt[t] = make_double2(cos(ang), sin(ang));
  }
  for (int t = 0; t < SPACE_PERIOD; t++) {
    double const ang = -2.0 * M_PI * ((11 * t) % SPACE_PERIOD) / SPACE_PERIOD;
    stb[t] = make_double2(cos(ang), sin(ang));
  }
  AF_TRY(hipMemcpyAsync(b->mark_tab, mt.data(), mt.size() * sizeof(double2), hipMemcpyHostToDevice, b->stream));
  AF_TRY(hipMemcpyAsync(b->space_tab, stb.data(), stb.size() * sizeof(double2), hipMemcpyHostToDevice, b->stream));
  AF_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

extern "C" {

kq_afsk_bank *kq_afsk_create(const kq_afsk_config *cfg) {
  if (!cfg || cfg->max_sessions == 0 || cfg->max_frames == 0) {
    kq_internal_set_error("kq_afsk_create: max_sessions and max_frames must be positive");
    return nullptr;
  }
  kq_afsk_bank *b = new kq_afsk_bank;
  b->cfg = *cfg;
  if (afsk_alloc(b) != 0) {
    kq_afsk_destroy(b);
    return nullptr;
  }
  return b;
}

int kq_afsk_destroy(kq_afsk_bank *b) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b) return -1;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  void *ptrs[] = {b->pend, b->hist, b->resp, b->tw, b->last_out, b->mark_tab, b->space_tab, b->state,
                  b->hdlc, b->frames, b->frame_len, b->nframes, b->dropped, b->staging};
  for (void *p : ptrs) (void)hipFree(p);
  if (b->own_stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return 0;
}

int kq_afsk_push(kq_afsk_bank *b, const void *samples, int format, unsigned nsessions, size_t nsamples,
                 size_t session_stride, int on_device) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b || (!samples && nsamples)) {
    kq_internal_set_error("kq_afsk_push: null argument");
    return -1;
  }
  if (format != KQ_PCM_F32 && format != KQ_PCM_S16BE) {
    kq_internal_set_error("kq_afsk_push: unknown sample format %d", format);
    return -1;
  }
  if (nsessions != b->cfg.max_sessions) {
    kq_internal_set_error("kq_afsk_push: every call carries all %u sessions (got %u)", b->cfg.max_sessions, nsessions);
    return -1;
  }
  if (session_stride < nsamples) {
    kq_internal_set_error("kq_afsk_push: session_stride %zu < nsamples %zu", session_stride, nsamples);
    return -1;
  }
  if (nsamples == 0) return 0;
  size_t const esize = format == KQ_PCM_S16BE ? 2 : 4;
  const void *src = samples;
  size_t stride = session_stride;
  if (!on_device) {
    size_t const need = (size_t)nsessions * nsamples * esize;
    if (need > b->staging_bytes) {
      AF_TRY(hipStreamSynchronize(b->stream));
      (void)hipFree(b->staging);
      b->staging = nullptr;
      AF_TRY(hipMalloc(&b->staging, need));
      b->staging_bytes = need;
    }
    AF_TRY(hipMemcpy2DAsync(b->staging, nsamples * esize, samples, session_stride * esize, nsamples * esize, nsessions,
                            hipMemcpyHostToDevice, b->stream));
    src = b->staging;
    stride = nsamples;
  }
  AfskArgs a{};
  a.src = src;
  a.format = format;
  a.stride = stride;
  a.nnew = nsamples;
  a.fill = b->fill;
  a.pend = b->pend;
  a.hist = b->hist;
  a.resp = b->resp;
  a.tw = b->tw;
  a.mark_tab = b->mark_tab;
  a.space_tab = b->space_tab;
  a.state = b->state;
  a.hdlc = b->hdlc;
  a.frames = b->frames;
  a.frame_len = b->frame_len;
  a.nframes = b->nframes;
  a.dropped = b->dropped;
  a.max_frames = (int)b->cfg.max_frames;
  a.last_out = b->last_out;
  hipLaunchKernelGGL(k_afsk, dim3(nsessions), dim3(kThreads), 0, b->stream, a);
  AF_TRY(hipGetLastError());
  size_t const total = (size_t)b->fill + nsamples;
  int const nblk = (int)(total / AL);
  b->fill = (int)(total % AL);
  b->blocks += nblk;
  if (!on_device) AF_TRY(hipStreamSynchronize(b->stream));  // the staging buffer is reused by the next call
  return nblk;
}

int kq_afsk_sync(kq_afsk_bank *b) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b) return -1;
  AF_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

int kq_afsk_num_frames(kq_afsk_bank *b, unsigned session) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b || session >= b->cfg.max_sessions) return -1;
  int n = 0;
  AF_TRY(hipStreamSynchronize(b->stream));
  AF_TRY(hipMemcpy(&n, b->nframes + session, sizeof n, hipMemcpyDeviceToHost));
  return n;
}

int kq_afsk_dropped_frames(kq_afsk_bank *b, unsigned session) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b || session >= b->cfg.max_sessions) return -1;
  int n = 0;
  AF_TRY(hipStreamSynchronize(b->stream));
  AF_TRY(hipMemcpy(&n, b->dropped + session, sizeof n, hipMemcpyDeviceToHost));
  return n;
}

int kq_afsk_pull_frame(kq_afsk_bank *b, unsigned session, unsigned index, unsigned char *dst, size_t cap) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b || !dst || session >= b->cfg.max_sessions) return -1;
  int const n = kq_afsk_num_frames(b, session);
  if (n < 0 || index >= (unsigned)n) {
    kq_internal_set_error("kq_afsk_pull_frame: session %u has %d frames", session, n);
    return -1;
  }
  int len = 0;
  size_t const slot = (size_t)session * b->cfg.max_frames + index;
  AF_TRY(hipMemcpy(&len, b->frame_len + slot, sizeof len, hipMemcpyDeviceToHost));
  size_t const take = std::min(cap, (size_t)len);
  AF_TRY(hipMemcpy(dst, b->frames + slot * FRAME_MAX, take, hipMemcpyDeviceToHost));
  return len;
}

int kq_afsk_clear_frames(kq_afsk_bank *b) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b) return -1;
  AF_TRY(hipMemsetAsync(b->nframes, 0, b->cfg.max_sessions * sizeof(int), b->stream));
  AF_TRY(hipMemsetAsync(b->dropped, 0, b->cfg.max_sessions * sizeof(int), b->stream));
  return 0;
}

int kq_afsk_pull_filter_output(kq_afsk_bank *b, unsigned session, float *dst_re_im, size_t cap_complex) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b || !dst_re_im || session >= b->cfg.max_sessions || cap_complex < (size_t)AL) return -1;
  AF_TRY(hipStreamSynchronize(b->stream));
  AF_TRY(hipMemcpy(dst_re_im, b->last_out + (size_t)session * AL, AL * sizeof(float2), hipMemcpyDeviceToHost));
  return AL;
}

int kq_afsk_pull_state(kq_afsk_bank *b, unsigned session, kq_afsk_state *out) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);
  if (!b || !out || session >= b->cfg.max_sessions) return -1;
  AfskState st;
  AF_TRY(hipStreamSynchronize(b->stream));
  AF_TRY(hipMemcpy(&st, b->state + session, sizeof st, hipMemcpyDeviceToHost));
  out->symphase = st.symphase;
  out->frame_bit = st.frame_bit;
  out->flagsync = st.flagsync;
  out->ones = st.ones;
  out->last_val = st.last_val;
  out->mid_val = st.mid_val;
  out->decoded_packets = st.decoded_packets;
  out->pending_samples = b->fill;
  out->blocks = b->blocks;
  return 0;
}

}  // extern "C"
