// kq_design.hip -- filter response design on the device (control plane: runs per retune, never per block).
//
// What the reference does on the CPU with FFTW in set_filter / window_filter / window_rfilter / make_kaiser / noise_gain
// (filter.c:282-546) and in the FM demodulator's prologue (fm.c:54-66): a target spectrum -- a brick wall between two
// edges, the 300 Hz / -6 dB-per-octave de-emphasis curve, or one the caller supplies -- is taken to the time domain,
// rotated so that the impulse response sits in the first M taps, shaped by a Kaiser window, zero padded and taken
// back.  Here one workgroup designs one response in LDS: target spectrum -> inverse transform -> taps -> forward
// transform -> response + sum |H|^2 for noise_gain; a launch designs a batch (every channel of a bank retuning at
// once costs one launch).  Arithmetic is float, like the reference's; the responses agree with the oracle's to a few
// 1e-12 absolute (response peak 1/N ~ 6e-5).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include "kq_design.hpp"
#include "kq_device.hpp"
#include "kq_ldsfft.hpp"
#include "kq_ctl.hpp"

void kq_internal_set_error(const char *fmt, ...);

namespace kq {

namespace {

enum { SPEC_GIVEN = 0, SPEC_BAND = 1, SPEC_DEEMPH = 2 };

// modified Bessel function of the first kind, order 0: power series sum_k (x^2/4)^k / (k!)^2 in float, cut off where
// a term drops below 1e-12 of the sum or after 40 terms (the truncation rule of filter.c:282-293)
__device__ float bessel_i0(float x) {
#pragma clang fp contract(off)  // one rounding per operation, as the host compiler evaluates the reference's float code
  float const q = 0.25f * x * x;
  float term = q, sum = 1.f + q;
  for (int k = 2; k < 40; k++) {
    term *= q / (float)(k * k);
    sum += term;
    if (term < 1e-12f * sum) break;
  }
  return sum;
}

// Kaiser window tap n of M (filter.c:337-357): symmetric about the middle, which is exactly 1 for odd M
__device__ float kaiser_tap(int n, int M, float beta) {
#pragma clang fp contract(off)
  int const m = min(n, M - 1 - n);
  if ((M & 1) && m == (M - 1) / 2) return 1.f;
  float const arg = 3.14159265358979323846f * beta;
  float const pos = (2.0f / (float)(M - 1)) * (float)m - 1.f;  // -1 at the edge .. 0 at the middle
  return bessel_i0(arg * sqrtf(1.f - pos * pos)) * (1.f / bessel_i0(arg));
}

__global__ void k_kaiser(float *__restrict__ w, int M, float beta) {
  int const n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < M) w[n] = kaiser_tap(n, M, beta);
}

// grid: one workgroup per response.  REAL: the taps are real (window_rfilter): the spectrum is N/2+1 bins in and out.
// dynamic LDS: N float2
template <bool REAL>
__global__ void k_design(FftDim dim, int M, int spec, const DesignJob *__restrict__ jobs, const float2 *__restrict__ given,
                         float2 *__restrict__ out, float2 *__restrict__ scratch, float *__restrict__ sumsq, int nsum,
                         const float2 *__restrict__ tw, int tw_log2, const DesignTarget *__restrict__ targets,
                         const unsigned char *__restrict__ ctlq, unsigned njobs) {
  // a launch on a bank's stream may take the filter side's control records along: the workgroups behind the design jobs
  if (ctlq != nullptr && blockIdx.x >= njobs) {
    ctl_apply_record(ctlq, blockIdx.x - njobs, threadIdx.x, blockDim.x);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  __shared__ float red_f[16];
  __shared__ int red_i[16];
  int const N = dim.n, nbins = REAL ? N / 2 + 1 : N;
  DesignJob const job = jobs[blockIdx.x];
  const float2 *in = given ? given + (size_t)blockIdx.x * nbins : nullptr;
  float2 *res = targets ? static_cast<float2 *>(targets[blockIdx.x].resp) : out + (size_t)blockIdx.x * nbins;
  if (!res) return;  // a job withdrawn after it was queued (the whole workgroup leaves)
  float2 *taps = scratch + (size_t)blockIdx.x * N;

  // ---- target spectrum, bin i of N (both halves: the inverse transform below is complex)
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    float2 s = make_float2(0.f, 0.f);
    int const j = i <= N / 2 ? i : N - i;  // the bin a real signal's Hermitian half mirrors to
    if (spec == SPEC_BAND) {               // filter.c:525-535: gain where low <= f <= high, f the signed bin over N
      float const f = i <= N / 2 ? (float)i / (float)N : (float)(i - N) / (float)N;
      if (f >= job.low && f <= job.high) s.x = job.gain;
    } else if (spec == SPEC_DEEMPH) {      // fm.c:59-63: gain * 300 / f between 300 and 6000 Hz (job.low = rate / N)
      float const f = (float)j * job.low;
      if (f >= 300.f && f <= 6000.f) s.x = job.gain * 300.f / f;
    } else if (REAL) {                     // c2r semantics: Hermitian extension, DC and Nyquist taken as real
      float2 const v = in[j];
      s = (j == 0 || j == N / 2) ? make_float2(v.x, 0.f) : (i <= N / 2 ? v : make_float2(v.x, -v.y));
    } else {
      s = in[i];
    }
    lds[fft_pos((unsigned)i, dim)] = s;
  }
  fft_any<+1>(lds, dim, tw, tw_log2);

  // ---- taps: rotate by M/2 so the response is causal, window, scale by 1/N (filter.c:377-386); zero beyond M
  float const inv_n = 1.0f / (float)N;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    float2 t = make_float2(0.f, 0.f);
    if (n < M) {
      int const src = (n - M / 2 + N) % N;
      float2 x = lds[src];
      // The reference rotates IN PLACE, n descending (filter.c:389-390, 445-446): for n < M/2 the source slot N - M/2 + n
      // lies above every slot still to be written only while L > M/2; with a longer impulse response (M >= 2 L) the slots
      // up to M - 1 among them have been written already -- from their own source, which at that time was untouched -- and
      // the reference's taps carry that.  Same here (one level deep: the source of a source is below it).
      if (n < M / 2 && src < M) {
        float2 const y = lds[src - M / 2];
        float const ws = kaiser_tap(src, M, job.beta);
        x = REAL ? make_float2(y.x * ws * inv_n, 0.f) : make_float2(y.x * ws * inv_n, y.y * ws * inv_n);
      }
      float const w = kaiser_tap(n, M, job.beta);
      t = REAL ? make_float2(x.x * w * inv_n, 0.f) : make_float2(x.x * w * inv_n, x.y * w * inv_n);
    }
    taps[n] = t;
  }
  __threadfence_block();
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {  // read back what other threads of the workgroup just wrote
    const volatile float *q = reinterpret_cast<const volatile float *>(taps + n);
    lds[fft_pos((unsigned)n, dim)] = make_float2(q[0], q[1]);
  }
  fft_any<-1>(lds, dim, tw, tw_log2);

  // ---- response, and the sum of |H|^2 noise_gain is built from (filter.c:472-497: nsum bins)
  float acc = 0;
  int dummy = 0;
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
    float2 const h = lds[k];
    res[k] = h;
    if (k < nsum) acc += h.x * h.x + h.y * h.y;
  }
  block_sum_fi(acc, dummy, red_f, red_i);
  if (threadIdx.x == 0 && sumsq) sumsq[blockIdx.x] = acc;
  if (threadIdx.x == 0 && targets && targets[blockIdx.x].noise_gain) *targets[blockIdx.x].noise_gain = targets[blockIdx.x].ng_scale * acc;
}

// twiddle tables exp(-2 pi i k / T), k < T/2, per device and size (built in double, rounded once)
const float2 *design_twiddles(int log2T) {
  static std::mutex mu;
  static std::map<std::pair<int, int>, float2 *> tabs;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  auto it = tabs.find({dev, log2T});
  if (it != tabs.end()) return it->second;
  size_t const T = (size_t)1 << log2T;
  std::vector<float2> h(T / 2 ? T / 2 : 1);
  for (size_t k = 0; k < T / 2; k++) {
    double const a = -2.0 * M_PI * (double)k / (double)T;
    h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  float2 *d = nullptr;
  if (hipMalloc((void **)&d, h.size() * sizeof(float2)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return nullptr;
  }
  tabs[{dev, log2T}] = d;
  return d;
}

}  // namespace

bool fft_size_ok(int n) {
  if (n < 2 || n > 65536 || (n & 1)) return false;
  for (int p : {2, 3, 5, 7})
    while (n % p == 0) n /= p;
  return n == 1;
}

// The plan of an n-point transform on the generic path, tables on the current device (cached per device and size)
FftDim fft_dim(int n, bool *ok) {
  FftDim d{};
  d.n = n;
  d.log2n = -1;
  if (ok) *ok = false;
  if (n >= 1 && (n & (n - 1)) == 0) {
    d.log2n = 0;
    while ((1 << d.log2n) < n) d.log2n++;
    if (ok) *ok = true;
    return d;
  }
  {  // (odd sizes are fine here -- a factor of a two-pass transform may be one; fft_size_ok's evenness is the filters' rule)
    int m = n;
    for (int p : {2, 3, 5, 7})
      while (m % p == 0) m /= p;
    if (n < 2 || n > 65536 || m != 1) return d;
  }
  {  // radices: 4s first (fewest passes), then 2, 3s, 5s, 7s
    int m = n;
    while (m % 4 == 0) d.f[d.nf++] = 4, m /= 4;
    while (m % 2 == 0) d.f[d.nf++] = 2, m /= 2;
    while (m % 3 == 0) d.f[d.nf++] = 3, m /= 3;
    while (m % 5 == 0) d.f[d.nf++] = 5, m /= 5;
    while (m % 7 == 0) d.f[d.nf++] = 7, m /= 7;
  }
  static std::mutex mu;
  static std::map<std::pair<int, int>, std::pair<const unsigned short *, const float2 *>> tabs;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return d;
  std::lock_guard<std::mutex> lk(mu);
  auto it = tabs.find({dev, n});
  if (it == tabs.end()) {
    std::vector<unsigned short> rev(n);
    for (int p = 0; p < n; p++) {  // position p = sum_k d_k prod_{j<k} f_j holds index i = sum_k d_k n / prod_{j<=k} f_j
      int rest = p, weight = n, i = 0;
      for (int k = 0; k < d.nf; k++) {
        weight /= d.f[k];
        i += (rest % d.f[k]) * weight;
        rest /= d.f[k];
      }
      rev[i] = (unsigned short)p;
    }
    std::vector<float2> tw(n);
    for (int k = 0; k < n; k++) {
      double const a = -2.0 * M_PI * (double)k / (double)n;
      tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    unsigned short *drev = nullptr;
    float2 *dtw = nullptr;
    if (hipMalloc((void **)&drev, n * sizeof(unsigned short)) != hipSuccess || hipMalloc((void **)&dtw, n * sizeof(float2)) != hipSuccess ||
        hipMemcpy(drev, rev.data(), n * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dtw, tw.data(), n * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
      if (drev) (void)hipFree(drev);
      if (dtw) (void)hipFree(dtw);
      return d;
    }
    it = tabs.emplace(std::make_pair(dev, n), std::make_pair((const unsigned short *)drev, (const float2 *)dtw)).first;
  }
  d.rev = it->second.first;
  d.twc = it->second.second;
  d.tw_n = n;
  if (ok) *ok = true;
  return d;
}

namespace {

// Workspace of the design kernels: per device, grown on demand, never freed.  (hipMalloc / hipFree around every design --
// round 1's form -- made each retune a device-wide synchronisation: hipFree waits for everything in flight, so
// kq_bank_set_filter on a bank running at real time stalled the host for the four calls it had queued, 6 ms at 32768
// channels.  tools/churn_probe.py.)  The kernels run on the NULL stream, as they always did, and wait for that stream only:
// the bank's streams are non-blocking, so nothing of theirs is waited for.  (A stream of the workspace's own was tried first:
// streams are dealt onto a handful of hardware queues in creation order, and one more stream created before the bank's copy
// streams moved those onto queues they then shared with the kernels -- with_host_io went from 0.99 to 0.87 of the resident
// step, found by bisecting the round's commits.)
struct Workspace {
  hipStream_t stream = nullptr;  // always the null stream
  void *buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[5] = {0, 0, 0, 0, 0};
  void *host = nullptr;  // pinned: job list in, results out
  size_t host_cap = 0;
  bool reserve(int i, size_t bytes) {
    if (bytes <= cap[i]) return true;
    // growing frees the old buffer -- a synchronising call, but only until the largest batch has been seen once
    if (buf[i]) (void)hipFree(buf[i]);
    buf[i] = nullptr;
    cap[i] = 0;
    size_t const want = bytes < 4096 ? 4096 : bytes;
    if (hipMalloc(&buf[i], want) != hipSuccess) return false;
    cap[i] = want;
    return true;
  }
  bool reserve_host(size_t bytes) {
    if (bytes <= host_cap) return true;
    if (host) (void)hipHostFree(host);
    host = nullptr;
    host_cap = 0;
    size_t const want = bytes < 65536 ? 65536 : bytes;
    if (hipHostMalloc(&host, want, hipHostMallocDefault) != hipSuccess) return false;
    host_cap = want;
    return true;
  }
};
std::mutex g_design_mu;  // one design at a time per process (control plane)
Workspace *workspace() {
  static std::map<int, Workspace> ws;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  return &ws[dev];
}

}  // namespace

// Designs jobs.size() responses of N = L + M - 1 points on the current device; `given`: jobs.size() * nbins target
// bins (SPEC_GIVEN) or null.  Returns 0 and fills `out` (jobs.size() * nbins) and, when asked, the |H|^2 sums.
// Runs on the null stream and waits for that stream only: whatever the (non-blocking) streams of a bank are doing goes on.
int design_batch(int L, int M, bool real_taps, int spec, const std::vector<DesignJob> &jobs, const cfloat *given,
                 std::vector<cfloat> &out, std::vector<float> *sumsq, int nsum) {
  int const N = L + M - 1;
  bool dim_ok = false;
  FftDim const dim = (N >= 2 && N <= 16384) ? fft_dim(N, &dim_ok) : FftDim{};
  int const log2n = dim.log2n >= 0 ? dim.log2n : 1;  // (the half-circle table is only read for powers of two)
  if (!dim_ok || M < 1 || jobs.empty()) {
    kq_internal_set_error("response design: N = %d must be 2^a 3^b 5^c 7^d in 2..16384", N);
    return -1;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    kq_internal_set_error("response design: no HIP device (the design kernels are the only path)");
    return -1;
  }
  size_t const count = jobs.size(), nbins = real_taps ? (size_t)N / 2 + 1 : (size_t)N;
  const float2 *tw = design_twiddles(log2n);
  std::lock_guard<std::mutex> lk(g_design_mu);
  Workspace *w = workspace();
  size_t const b_jobs = count * sizeof(DesignJob), b_in = given ? count * nbins * sizeof(float2) : 0,
               b_out = count * nbins * sizeof(float2), b_sum = count * sizeof(float);
  // pinned host block: [jobs | given | out | sums], each part 16-byte aligned
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t const o_in = al(b_jobs), o_out = o_in + al(b_in), o_sum = o_out + al(b_out), h_total = o_sum + al(b_sum);
  if (!tw || !w || !w->reserve(0, b_jobs) || !w->reserve(1, b_in) || !w->reserve(2, b_out) ||
      !w->reserve(3, count * (size_t)N * sizeof(float2)) || !w->reserve(4, b_sum) || !w->reserve_host(h_total)) {
    kq_internal_set_error("response design: device allocation failed");
    return -1;
  }
  char *hp = static_cast<char *>(w->host);
  memcpy(hp, jobs.data(), b_jobs);
  if (given) memcpy(hp + o_in, given, b_in);
  hipStream_t const st = w->stream;
  bool ok = hipMemcpyAsync(w->buf[0], hp, b_jobs, hipMemcpyHostToDevice, st) == hipSuccess;
  if (ok && given) ok = hipMemcpyAsync(w->buf[1], hp + o_in, b_in, hipMemcpyHostToDevice, st) == hipSuccess;
  if (ok) {
    size_t const lds_bytes = (size_t)N * sizeof(float2);
    int const threads = N >= 1024 ? 256 : 64;
    auto go = [&](auto kernel) {
      ensure_dynamic_lds((const void *)kernel, lds_bytes);
      hipLaunchKernelGGL(kernel, dim3((unsigned)count), dim3(threads), lds_bytes, st, dim, M, spec, (const DesignJob *)w->buf[0],
                         (const float2 *)(given ? w->buf[1] : nullptr), (float2 *)w->buf[2], (float2 *)w->buf[3], (float *)w->buf[4],
                         nsum, tw, log2n, (const DesignTarget *)nullptr, (const unsigned char *)nullptr, (unsigned)count);
    };
    real_taps ? go(k_design<true>) : go(k_design<false>);
    ok = hipGetLastError() == hipSuccess;
  }
  if (ok) ok = hipMemcpyAsync(hp + o_out, w->buf[2], b_out, hipMemcpyDeviceToHost, st) == hipSuccess;
  if (ok && sumsq) ok = hipMemcpyAsync(hp + o_sum, w->buf[4], b_sum, hipMemcpyDeviceToHost, st) == hipSuccess;
  if (ok) ok = hipStreamSynchronize(st) == hipSuccess;
  out.resize(count * nbins);
  if (ok) {
    memcpy((void *)out.data(), hp + o_out, b_out);
    if (sumsq) {
      sumsq->resize(count);
      memcpy(sumsq->data(), hp + o_sum, b_sum);
    }
  }
  if (!ok) kq_internal_set_error("response design: %s", hipGetErrorString(hipGetLastError()));
  return ok ? 0 : -1;
}

int make_kaiser(float *window, unsigned M, float beta) {
  if (M == 0) return 0;
  std::lock_guard<std::mutex> lk(g_design_mu);
  Workspace *w = workspace();
  if (!w || !w->reserve(2, M * sizeof(float)) || !w->reserve_host(M * sizeof(float))) return -1;
  hipLaunchKernelGGL(k_kaiser, dim3((M + 255) / 256), dim3(256), 0, w->stream, (float *)w->buf[2], (int)M, beta);
  if (hipMemcpyAsync(w->host, w->buf[2], M * sizeof(float), hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
      hipStreamSynchronize(w->stream) != hipSuccess)
    return -1;
  memcpy(window, w->host, M * sizeof(float));
  return 0;
}

int window_filter(int L, int M, std::vector<cfloat> &response, float beta) {
  if ((int)response.size() != L + M - 1) return -1;
  std::vector<cfloat> out;
  if (design_batch(L, M, false, SPEC_GIVEN, {DesignJob{0, 0, beta, 0}}, response.data(), out, nullptr, 0)) return -1;
  response.swap(out);
  return 0;
}

int window_rfilter(int L, int M, std::vector<cfloat> &response, float beta) {
  if ((int)response.size() != (L + M - 1) / 2 + 1) return -1;
  std::vector<cfloat> out;
  if (design_batch(L, M, true, SPEC_GIVEN, {DesignJob{0, 0, beta, 0}}, response.data(), out, nullptr, 0)) return -1;
  response.swap(out);
  return 0;
}

// set_filter's design step for a batch of slaves of one geometry (filter.c:500-546): unity passband between the
// edges (cycles per output sample), scaled 1/N for the unnormalised transforms and by a further 1/sqrt(2) where two
// sidebands add (REAL and CROSS_CONJ outputs, filter.c:518-522); noise gains as filter.c:472-497 (complex master).
int design_responses(int N, int L_dec, int M_dec, int out_type, const std::vector<BandEdges> &edges, std::vector<cfloat> &responses,
                     std::vector<float> &noise_gains) {
  bool const two_sided = out_type == FT_REAL || out_type == FT_CROSS_CONJ;
  float gain = 1.0f / (float)N;
  if (two_sided) gain *= (float)M_SQRT1_2;
  std::vector<DesignJob> jobs;
  for (BandEdges const &e : edges) jobs.push_back(DesignJob{e.low, e.high, e.beta, gain});
  std::vector<float> sums;
  int const n_dec = L_dec + M_dec - 1;
  if (design_batch(L_dec, M_dec, false, SPEC_BAND, jobs, nullptr, responses, &sums, n_dec)) return -1;
  noise_gains.resize(sums.size());
  for (size_t i = 0; i < sums.size(); i++) noise_gains[i] = (two_sided ? 2.f : 1.f) * (float)N * sums[i];
  return 0;
}

// The same design, queued on a stream of the caller and left on the device: job i's response goes to targets[i].resp
// and its noise gain to targets[i].noise_gain -- no copy back, nothing waits.  `jobs` and `targets` are read by the
// kernel where they lie (pinned host memory the caller keeps until the launch is over), `scratch` is count * (L_dec +
// M_dec - 1) float2 of device memory.  The twiddle table of this size must exist (design_prepare, which may block).
int design_prepare(int L_dec, int M_dec) {
  int const N = L_dec + M_dec - 1;
  if (N < 2 || N > 16384) return -1;
  bool ok = false;
  FftDim const dim = fft_dim(N, &ok);
  if (!ok) return -1;
  return design_twiddles(dim.log2n >= 0 ? dim.log2n : 1) ? 0 : -1;
}
int design_launch(void *stream, int L_dec, int M_dec, const DesignJob *jobs, const DesignTarget *targets, unsigned count,
                  void *scratch, const void *ctl_queue, unsigned ctl_records) {
  int const N = L_dec + M_dec - 1;
  bool dim_ok = false;
  FftDim const dim = fft_dim(N, &dim_ok);  // (cached since design_prepare: no allocation here)
  int const log2n = dim.log2n >= 0 ? dim.log2n : 1;
  hipStream_t const st = static_cast<hipStream_t>(stream);
  const float2 *tw = design_twiddles(log2n);
  if (!tw || !dim_ok || count == 0) return (tw && dim_ok) ? 0 : -1;
  size_t const lds_bytes = (size_t)N * sizeof(float2);
  ensure_dynamic_lds((const void *)k_design<false>, lds_bytes);
  if (!ctl_queue) ctl_records = 0;
  hipLaunchKernelGGL(k_design<false>, dim3(count + ctl_records), dim3(N >= 1024 ? 256 : 64), lds_bytes, st, dim, M_dec, (int)SPEC_BAND,
                     jobs, (const float2 *)nullptr, (float2 *)nullptr, (float2 *)scratch, (float *)nullptr, N, tw, log2n, targets,
                     static_cast<const unsigned char *>(ctl_queue), count);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// gain and noise-gain factor of a slave's design (what design_responses applies on the host)
void design_scales(int N, int out_type, float *gain, float *ng_scale) {
  bool const two_sided = out_type == FT_REAL || out_type == FT_CROSS_CONJ;
  *gain = 1.0f / (float)N;
  if (two_sided) *gain *= (float)M_SQRT1_2;
  *ng_scale = (two_sided ? 2.f : 1.f) * (float)N;
}

std::vector<cfloat> design_response(int N, int L_dec, int M_dec, int out_type, float low, float high, float beta,
                                    float *noise_gain_out) {
  std::vector<cfloat> r;
  std::vector<float> ng;
  if (design_responses(N, L_dec, M_dec, out_type, {BandEdges{low, high, beta}}, r, ng)) return {};
  if (noise_gain_out) *noise_gain_out = ng[0];
  return r;
}

std::vector<cfloat> design_fm_audio_response(int AL, int AM, float dsamprate, float beta) {
  int const AN = AL + AM - 1;
  std::vector<cfloat> r;
  // fm.c:42: 10 / AN brings the subjective level up; job.low carries the bin spacing in Hz
  if (design_batch(AL, AM, true, SPEC_DEEMPH, {DesignJob{dsamprate / (float)AN, 0, beta, 10.0f / (float)AN}}, nullptr, r, nullptr, 0))
    return {};
  return r;
}

}  // namespace kq
