// kq_compat.cpp -- the reference's one-channel filter / osc / dsp API on top of the gfx950 kernels
// (include/ka9q_hip_compat.h).  Blocking, one block per call, host buffers at the boundary exactly
// as the reference's callers expect (radio.c:139-142 fills input.c[]; linear.c:211 reads output.c[]).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/ka9q_hip_radio.h"
#include "kq_design.hpp"
#include "kq_device.hpp"

namespace {

struct DevCtx {
  bool ok = false;
  int device = 0;
  hipStream_t stream = nullptr;
  float *d_scalar = nullptr;  // one float of device scratch (kq_compat_compute_n0)
  std::map<int, float2 *> tw;  // log2(T) -> table of T/2 twiddles
  std::mutex mu;
};

DevCtx &ctx() {
  static DevCtx c;
  return c;
}

bool ctx_init() {
  DevCtx &c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.ok) return true;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    fprintf(stderr, "ka9q_hip: no HIP device; the filter API has no CPU fallback\n");
    return false;
  }
  if (hipGetDevice(&c.device) != hipSuccess) return false;
  if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) return false;
  if (hipMalloc((void **)&c.d_scalar, sizeof(float)) != hipSuccess) return false;
  c.ok = true;
  return true;
}

float2 *twiddles(int log2T) {
  DevCtx &c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  auto it = c.tw.find(log2T);
  if (it != c.tw.end()) return it->second;
  size_t const T = (size_t)1 << log2T;
  std::vector<float2> h(T / 2 ? T / 2 : 1);
  for (size_t k = 0; k < T / 2; k++) {
    double const a = -2.0 * M_PI * (double)k / (double)T;
    h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  float2 *d = nullptr;
  if (hipMalloc((void **)&d, h.size() * sizeof(float2)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  c.tw[log2T] = d;
  return d;
}

int ilog2(unsigned v) {
  int l = 0;
  while ((1u << l) < v) l++;
  return l;
}

struct MasterDev {
  int N, log2N;
  kq::FftDim dim;
  float2 *d_in = nullptr, *d_fdomain = nullptr;
  float2 *d_tmp = nullptr;  // N > 16384: scratch of the two-pass transform
  std::vector<float2> stage;
  float2 *tw = nullptr;
  // which block's samples d_in holds, as of the work queued so far: the upload and this number move together under
  // in_mu, so a reader that queues its copy under the same lock knows which block it will get (compat_snapshot_window)
  std::mutex in_mu;
  unsigned in_block = 0;
};

struct SlaveDev {
  int Ndec;
  float2 *d_resp = nullptr, *d_out = nullptr;
};

inline float re(kq_cfloat z) { return __real__ z; }
inline float im(kq_cfloat z) { return __imag__ z; }

}  // namespace

namespace kq {

int compat_master_device(void) { return ctx().ok ? ctx().device : -1; }

int compat_snapshot_window(struct filter_in *m, float2 *dst, unsigned *block) {
  if (!m || !m->fwd_plan || !dst) return -1;
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  MasterDev *d = (MasterDev *)m->fwd_plan;
  hipStream_t s = ctx().stream;
  {
    // The master may already have queued the next block's upload (it does not wait for its consumers, filter.c:146-172):
    // the copy lands in stream order, so the block it carries is the one recorded with the last upload queued.
    std::lock_guard<std::mutex> lk(d->in_mu);
    if (hipMemcpyAsync(dst, d->d_in, (size_t)d->N * sizeof(float2), hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
    if (block) *block = d->in_block;
  }
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  return d->N;
}

// The master's spectrum of the block last transformed (N bins, complex input): what execute_filter_output and
// compute_n0 read in the reference (filter.c:206-227, radio.c:396), copied so that the master may go on to its next block.
int compat_snapshot_spectrum(struct filter_in *m, float2 *dst, unsigned *block) {
  if (!m || !m->fwd_plan || !dst || m->in_type != COMPLEX) return -1;
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  MasterDev *d = (MasterDev *)m->fwd_plan;
  hipStream_t s = ctx().stream;
  {
    std::lock_guard<std::mutex> lk(d->in_mu);
    if (hipMemcpyAsync(dst, d->d_fdomain, (size_t)d->N * sizeof(float2), hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
    if (block) *block = d->in_block;
  }
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  return d->N;
}

}  // namespace kq

extern "C" {

float Kaiser_beta = 3.0;

float kq_compat_compute_n0(struct filter_in *m, int samprate, float low, float high) {
  if (!m || !m->fwd_plan || m->in_type != COMPLEX || samprate <= 0) return NAN;
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  MasterDev *d = (MasterDev *)m->fwd_plan;
  DevCtx &c = ctx();
  float r = NAN;
  std::lock_guard<std::mutex> lk(c.mu);  // one scratch float
  kq::launch_n0_single(c.stream, d->d_fdomain, d->N, samprate, low, high, c.d_scalar);
  if (hipMemcpyAsync(&r, c.d_scalar, sizeof r, hipMemcpyDeviceToHost, c.stream) != hipSuccess) return NAN;
  if (hipStreamSynchronize(c.stream) != hipSuccess) return NAN;
  return r;
}


struct filter_in *create_filter_input(unsigned int L, unsigned int M, enum filtertype in_type) {
  unsigned const N = L + M - 1;
  // FFTW plans any N (filter.c:78); here: a power of two up to 2^22, or 2^a 3^b 5^c 7^d (even) up to 65536
  bool const pow2 = (N & (N - 1)) == 0;
  if (L == 0 || M == 0 || N < 4 || (pow2 ? N > (1u << 22) : !kq::fft_size_ok((int)N))) {
    fprintf(stderr, "ka9q_hip: create_filter_input: N=%u must be a power of two in 4..4194304 or an even 2^a 3^b 5^c 7^d up to 65536\n", N);
    return NULL;
  }
  if (!ctx_init()) return NULL;
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  struct filter_in *m = (struct filter_in *)calloc(1, sizeof(*m));
  if (!m) return NULL;
  pthread_mutex_init(&m->filter_mutex, NULL);
  pthread_cond_init(&m->filter_cond, NULL);
  if (in_type != REAL && in_type != COMPLEX) {
    fprintf(stderr, "Filter input type %d, assuming complex\n", in_type);  // filter.c:69-71
    in_type = COMPLEX;
  }
  m->in_type = in_type;
  m->ilen = L;
  m->impulse_length = M;
  MasterDev *d = new MasterDev();
  d->N = (int)N;
  d->log2N = ilog2(N);
  d->tw = twiddles(d->log2N);  // (half-circle table of the next power of two: lds_fft's; also what the slaves are handed)
  bool dim_ok = false;
  d->dim = kq::fft_dim((int)N, &dim_ok);
  if (!dim_ok) d->tw = nullptr;
  d->stage.resize(N);
  if (!d->tw || hipMalloc((void **)&d->d_in, N * sizeof(float2)) != hipSuccess ||
      hipMalloc((void **)&d->d_fdomain, N * sizeof(float2)) != hipSuccess ||
      (N > 16384 && hipMalloc((void **)&d->d_tmp, N * sizeof(float2)) != hipSuccess)) {
    (void)hipFree(d->d_in);
    (void)hipFree(d->d_fdomain);
    delete d;
    free(m);
    return NULL;
  }
  m->fwd_plan = d;
  if (in_type == COMPLEX) {
    m->fdomain = (kq_cfloat *)calloc(N, sizeof(kq_cfloat));
    m->input_buffer.c = (kq_cfloat *)calloc(N, sizeof(kq_cfloat));  // history cleared: filter.c:76
    m->input.c = m->input_buffer.c + (M - 1);
  } else {
    m->fdomain = (kq_cfloat *)calloc(N / 2 + 1, sizeof(kq_cfloat));
    m->input_buffer.r = (float *)calloc(N, sizeof(float));
    m->input.r = m->input_buffer.r + (M - 1);
  }
  return m;
}

int execute_filter_input(struct filter_in *m) {
  if (m == NULL) return -1;  // filter.c:148-149
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  MasterDev *d = (MasterDev *)m->fwd_plan;
  hipStream_t s = ctx().stream;
  int const N = d->N;
  const void *src;
  if (m->in_type == REAL) {
    for (int i = 0; i < N; i++) d->stage[i] = make_float2(m->input_buffer.r[i], 0.f);
    src = d->stage.data();
  } else {
    src = m->input_buffer.c;
  }
  {
    // upload and transform are queued under one lock: a consumer's snapshot (queued under the same lock) then finds
    // window, spectrum and block number of ONE block, whichever side of this pair it lands on
    std::lock_guard<std::mutex> lk(d->in_mu);
    if (hipMemcpyAsync(d->d_in, src, N * sizeof(float2), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    d->in_block = m->blocknum + 1;  // only this thread moves blocknum (below, once the transform is done)
    if (N > 16384) {
      if (kq::launch_fft_large(s, d->d_in, d->d_fdomain, d->d_tmp, N, -1, d->tw, d->log2N)) return -1;
    } else
      kq::launch_fft_single(s, d->d_in, d->d_fdomain, d->dim, -1, d->tw, d->log2N);
  }
  size_t const bins = (m->in_type == REAL) ? N / 2 + 1 : N;
  if (hipMemcpyAsync(m->fdomain, d->d_fdomain, bins * sizeof(float2), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;

  pthread_mutex_lock(&m->filter_mutex);  // filter.c:154-157
  m->blocknum++;
  pthread_cond_broadcast(&m->filter_cond);
  pthread_mutex_unlock(&m->filter_mutex);

  if (m->in_type == REAL)
    memmove(m->input_buffer.r, m->input_buffer.r + m->ilen, (m->impulse_length - 1) * sizeof(float));
  else
    memmove(m->input_buffer.c, m->input_buffer.c + m->ilen, (m->impulse_length - 1) * sizeof(kq_cfloat));
  return 0;
}

int delete_filter_input(struct filter_in *m) {
  if (m == NULL) return 0;
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  MasterDev *d = (MasterDev *)m->fwd_plan;
  if (d) {
    (void)hipFree(d->d_in);
    (void)hipFree(d->d_fdomain);
    (void)hipFree(d->d_tmp);
    delete d;
  }
  free(m->input_buffer.c);  // same storage either way (union), as filter.c:259
  free(m->fdomain);
  free(m);
  return 0;
}

float noise_gain(struct filter_out const *f) {
  if (f == NULL) return NAN;
  struct filter_in const *m = f->master;
  int const N = (int)(m->ilen + m->impulse_length - 1);
  int const nd = N / (int)f->decimate;
  int const count = (m->in_type == REAL && f->out_type == REAL) ? nd / 2 + 1 : nd;
  float sum = 0;
  for (int i = 0; i < count; i++) sum += re(f->response[i]) * re(f->response[i]) + im(f->response[i]) * im(f->response[i]);
  if (f->out_type == REAL || f->out_type == CROSS_CONJ) return 2 * N * sum;
  return N * sum;
}

struct filter_out *create_filter_output(struct filter_in *master, kq_cfloat *response, unsigned int decimate,
                                        enum filtertype out_type) {
  if (master == NULL || decimate == 0) return NULL;  // filter.c:99-100
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  int const N = (int)(master->ilen + master->impulse_length - 1);
  int const nd = N / (int)decimate;
  if ((N % decimate) != 0) fprintf(stderr, "Warning: FFT size %d is not divisible by decimation ratio %u\n", N, decimate);
  {
    bool dim_ok = false;
    if (nd >= 4 && nd <= 16384) (void)kq::fft_dim(nd, &dim_ok);  // (makes and caches the plan the slave's kernel will ask for)
    if (!dim_ok) {
      fprintf(stderr, "ka9q_hip: create_filter_output: N/decimate=%d must be an even 2^a 3^b 5^c 7^d in 4..16384\n", nd);
      return NULL;
    }
  }
  struct filter_out *s = (struct filter_out *)calloc(1, sizeof(*s));
  if (!s) return NULL;
  pthread_mutex_init(&s->response_mutex, NULL);
  s->master = master;
  s->out_type = out_type;
  s->decimate = decimate;
  s->olen = master->ilen / decimate;
  s->response = response;
  s->noise_gain = response ? noise_gain(s) : NAN;
  SlaveDev *d = new SlaveDev();
  d->Ndec = nd;
  if (hipMalloc((void **)&d->d_resp, nd * sizeof(float2)) != hipSuccess ||
      hipMalloc((void **)&d->d_out, nd * sizeof(float2)) != hipSuccess ||
      hipMemset(d->d_resp, 0, nd * sizeof(float2)) != hipSuccess) {
    (void)hipFree(d->d_resp);
    (void)hipFree(d->d_out);
    delete d;
    free(s);
    return NULL;
  }
  s->rev_plan = d;
  if (out_type == REAL) {
    s->output_buffer.r = (float *)calloc(nd, sizeof(float));
    s->output.r = s->output_buffer.r + nd - s->olen;  // filter.c:140
  } else {
    s->output_buffer.c = (kq_cfloat *)calloc(nd, sizeof(kq_cfloat));
    s->output.c = s->output_buffer.c + nd - s->olen;  // filter.c:131
  }
  return s;
}

int execute_filter_output(struct filter_out *s) {
  if (s == NULL) return -1;
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  struct filter_in *m = s->master;
  MasterDev *md = (MasterDev *)m->fwd_plan;
  SlaveDev *sd = (SlaveDev *)s->rev_plan;
  hipStream_t st = ctx().stream;

  pthread_mutex_lock(&m->filter_mutex);  // filter.c:195-199: wait for a new block
  while (s->blocknum == m->blocknum) pthread_cond_wait(&m->filter_cond, &m->filter_mutex);
  s->blocknum = m->blocknum;
  pthread_mutex_unlock(&m->filter_mutex);

  int const nd = sd->Ndec;
  bool const real_out = s->out_type == REAL;
  // REAL in / REAL out reads N_dec/2+1 response bins (filter.c:209-212); every other combination reads all N_dec,
  // COMPLEX in / REAL out included: it folds in H[N_dec - p] X[N - p] (filter.c:232-234).  The same rule as noise_gain().
  size_t const rbins = (real_out && m->in_type == REAL) ? nd / 2 + 1 : nd;
  pthread_mutex_lock(&s->response_mutex);  // filter.c:201
  if (s->response == NULL) {
    pthread_mutex_unlock(&s->response_mutex);
    return -1;
  }
  hipError_t e = hipMemcpyAsync(sd->d_resp, s->response, rbins * sizeof(float2), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  pthread_mutex_unlock(&s->response_mutex);
  if (e != hipSuccess) return -1;

  kq::launch_slave_single(st, md->d_fdomain, sd->d_resp, sd->d_out, md->N, nd, m->in_type == REAL, (int)s->out_type, md->tw,
                          md->log2N);
  size_t const obytes = real_out ? nd * sizeof(float) : nd * sizeof(float2);
  if (hipMemcpyAsync(s->output_buffer.c, sd->d_out, obytes, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
  if (hipStreamSynchronize(st) != hipSuccess) return -1;
  return 0;
}

int delete_filter_output(struct filter_out *s) {
  if (s == NULL) return 0;
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  SlaveDev *d = (SlaveDev *)s->rev_plan;
  if (d) {
    (void)hipFree(d->d_resp);
    (void)hipFree(d->d_out);
    delete d;
  }
  pthread_mutex_destroy(&s->response_mutex);
  free(s->output_buffer.c);
  free(s->response);
  free(s);
  return 0;
}

int make_kaiser(float *window, unsigned int M, float beta) {
  if (window == NULL) return -1;
  if (!ctx_init()) return -1;
  kq::DeviceScope dev_scope_(ctx().device);
  return kq::make_kaiser(window, M, beta);
}

int window_filter(int L, int M, kq_cfloat *response, float beta) {
  if (response == NULL) return -1;
  if (!ctx_init()) return -1;
  kq::DeviceScope dev_scope_(ctx().device);
  int const N = L + M - 1;
  std::vector<kq::cfloat> r(N);
  memcpy((void *)r.data(), response, N * sizeof(kq_cfloat));
  if (kq::window_filter(L, M, r, beta)) return -1;
  memcpy(response, r.data(), N * sizeof(kq_cfloat));
  return 0;
}

int window_rfilter(int L, int M, kq_cfloat *response, float beta) {
  if (response == NULL) return -1;
  if (!ctx_init()) return -1;
  kq::DeviceScope dev_scope_(ctx().device);
  int const N = L + M - 1;
  std::vector<kq::cfloat> r(N / 2 + 1);
  memcpy((void *)r.data(), response, r.size() * sizeof(kq_cfloat));
  if (kq::window_rfilter(L, M, r, beta)) return -1;
  memcpy(response, r.data(), r.size() * sizeof(kq_cfloat));
  return 0;
}

int set_filter(struct filter_out *s, float low, float high, float kaiser_beta) {
  if (s == NULL) return -1;
  if (std::isnan(low) || std::isnan(high)) return -1;  // filter.c:504-505
  struct filter_in *m = s->master;
  int const L_dec = (int)s->olen;
  int const M_dec = (int)((m->impulse_length - 1) / s->decimate + 1);
  int const N = (int)(m->ilen + m->impulse_length - 1);
  kq::DeviceScope dev_scope_(ctx().ok ? ctx().device : -1);
  std::vector<kq::cfloat> r = kq::design_response(N, L_dec, M_dec, (int)s->out_type, low, high, kaiser_beta);
  if (r.empty()) return -1;
  kq_cfloat *fresh = (kq_cfloat *)malloc(r.size() * sizeof(kq_cfloat));
  if (!fresh) return -1;
  memcpy(fresh, r.data(), r.size() * sizeof(kq_cfloat));
  pthread_mutex_lock(&s->response_mutex);  // hot swap: filter.c:538-543
  kq_cfloat *old = s->response;
  s->response = fresh;
  s->noise_gain = noise_gain(s);
  pthread_mutex_unlock(&s->response_mutex);
  free(old);
  return 0;
}

// ---- NCO (host scalar API; inside the bank the kernels evaluate the same sequence in closed form) ----
int is_phasor_init(kq_cdouble x) {
  double const a = __real__ x, b = __imag__ x;
  if (std::isnan(a) || std::isnan(b) || a * a + b * b < 0.9) return 0;  // osc.c:14-18
  return 1;
}

static kq_cdouble unit_pi(double x) {
  kq_cdouble z;
  __real__ z = std::cos(x * M_PI);
  __imag__ z = std::sin(x * M_PI);
  return z;
}

void set_osc(struct osc *o, double f, double r) {
  pthread_mutex_lock(&o->mutex);
  if (!is_phasor_init(o->phasor)) {  // osc.c:24-27
    __real__ o->phasor = 1;
    __imag__ o->phasor = 0;
    o->steps = 0;
  }
  o->freq = f;
  o->rate = r;
  o->phasor_step = unit_pi(2 * f);
  if (r != 0) {
    o->phasor_step_step = unit_pi(2 * r);
  } else {
    __real__ o->phasor_step_step = 1;
    __imag__ o->phasor_step_step = 0;
  }
  pthread_mutex_unlock(&o->mutex);
}

static inline kq_cdouble zmul(kq_cdouble a, kq_cdouble b) {
  kq_cdouble z;
  __real__ z = __real__ a * __real__ b - __imag__ a * __imag__ b;
  __imag__ z = __real__ a * __imag__ b + __imag__ a * __real__ b;
  return z;
}

void renorm_osc(struct osc *o) {
  o->steps = 0;
  double const mag = std::hypot(__real__ o->phasor, __imag__ o->phasor);
  __real__ o->phasor /= mag;
  __imag__ o->phasor /= mag;
  if (o->rate != 0) {
    double const ms = std::hypot(__real__ o->phasor_step, __imag__ o->phasor_step);
    __real__ o->phasor_step /= ms;
    __imag__ o->phasor_step /= ms;
  }
}

kq_cdouble step_osc(struct osc *o) {
  kq_cdouble const now = o->phasor;
  if (o->freq != 0) {  // osc.c:43-47
    o->phasor = zmul(o->phasor, o->phasor_step);
    if (o->rate != 0) o->phasor_step = zmul(o->phasor_step, o->phasor_step_step);
  }
  if (++o->steps == 16384) renorm_osc(o);  // Renorm_rate, osc.c:11
  return now;
}

// ---- dsp.h helpers ----
kq_cfloat csincosf(float x) {
  kq_cfloat z;
  __real__ z = cosf(x);
  __imag__ z = sinf(x);
  return z;
}
kq_cfloat csincospif(float x) { return csincosf(x * (float)M_PI); }
kq_cdouble csincos(double x) {
  kq_cdouble z;
  __real__ z = std::cos(x);
  __imag__ z = std::sin(x);
  return z;
}
kq_cdouble csincospi(double x) { return csincos(x * M_PI); }
// filter.c:551-571.  Mixed float/double complex arithmetic as C evaluates it: the products with the double phasor
// are formed in double and rounded to float on assignment.
struct notchfilter *notch_create(double f, float bw) {
  struct notchfilter *nf = (struct notchfilter *)calloc(1, sizeof(struct notchfilter));
  if (!nf) return nullptr;
  __real__ nf->osc_phase = 1;
  __imag__ nf->osc_phase = 0;
  nf->osc_step = csincospi(2 * f);
  __real__ nf->dcstate = 0;
  __imag__ nf->dcstate = 0;
  nf->bw = bw;
  return nf;
}

kq_cfloat notch(struct notchfilter *nf, kq_cfloat s) {
  kq_cfloat r;
  if (!nf) {
    __real__ r = NAN;
    __imag__ r = 0;
    return r;
  }
  double const pr = __real__ nf->osc_phase, pi = __imag__ nf->osc_phase;
  double const sr = __real__ s, si = __imag__ s;
  // s = s * conj(osc_phase) - dcstate
  float const dr = (float)((sr * pr + si * pi) - (double)__real__ nf->dcstate);
  float const di = (float)((si * pr - sr * pi) - (double)__imag__ nf->dcstate);
  // dcstate += bw * s
  __real__ nf->dcstate = __real__ nf->dcstate + nf->bw * dr;
  __imag__ nf->dcstate = __imag__ nf->dcstate + nf->bw * di;
  // s *= osc_phase
  __real__ r = (float)((double)dr * pr - (double)di * pi);
  __imag__ r = (float)((double)dr * pi + (double)di * pr);
  // osc_phase *= osc_step
  double const tr = __real__ nf->osc_step, ti = __imag__ nf->osc_step;
  __real__ nf->osc_phase = pr * tr - pi * ti;
  __imag__ nf->osc_phase = pr * ti + pi * tr;
  return r;
}

float cnrmf(kq_cfloat x) { return __real__ x * __real__ x + __imag__ x * __imag__ x; }
double cnrm(kq_cdouble x) { return __real__ x * __real__ x + __imag__ x * __imag__ x; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// The FFTW entry points the reference calls OUTSIDE filter.c (include/ka9q_hip_fftw.h): fm.c:226-228,255,281-283 (the PL
// tone's 16384-point r2c transform), linear.c:90-92,178,313-317 (the carrier search's 65536-point transform), the
// allocators of fm.c:56,208 / modulate.c:115 (responses handed to create_filter_output, which frees them with free():
// these allocate with aligned_alloc), and main.c:102-103,183-184 (wisdom / threads: nothing to do).  With them
// `fm.o` / `linear.o` / `main.o` link against this library alone -- no libfftw3f on the link line.
// A plan is (size, kind, the caller's two buffers); fftwf_execute moves one transform over the link and back.
struct kq_fftwf_plan_s {
  int n, kind;  // kind 0: c2c (sign), 1: r2c, 2: c2r
  int sign;
  void *in, *out;
  int log2T;  // the half-circle table handed to the kernels (powers of two read it)
  kq::FftDim dim;
  float2 *d_in = nullptr, *d_out = nullptr, *d_tmp = nullptr;
  std::vector<float2> stage;
};

static kq_fftwf_plan_s *fftw_plan_make(int n, int kind, int sign, void *in, void *out) {
  bool const pow2 = n > 0 && (n & (n - 1)) == 0;
  if (n < 2 || !in || !out || (pow2 ? n > (1 << 22) : !kq::fft_size_ok(n))) {
    fprintf(stderr, "ka9q_hip: fftwf_plan: size %d must be a power of two up to 4194304 or an even 2^a 3^b 5^c 7^d up to 65536\n", n);
    return nullptr;
  }
  if (!ctx_init()) return nullptr;
  kq::DeviceScope dev_scope_(ctx().device);
  auto *p = new kq_fftwf_plan_s();
  p->n = n;
  p->kind = kind;
  p->sign = sign;
  p->in = in;
  p->out = out;
  p->log2T = ilog2((unsigned)n);
  bool ok = false;
  p->dim = kq::fft_dim(n, &ok);
  if (!ok || !twiddles(p->log2T) || hipMalloc((void **)&p->d_in, (size_t)n * sizeof(float2)) != hipSuccess ||
      hipMalloc((void **)&p->d_out, (size_t)n * sizeof(float2)) != hipSuccess ||
      (n > 16384 && hipMalloc((void **)&p->d_tmp, (size_t)n * sizeof(float2)) != hipSuccess)) {
    (void)hipFree(p->d_in);
    (void)hipFree(p->d_out);
    delete p;
    return nullptr;
  }
  if (kind != 0) p->stage.resize(n);
  return p;
}

extern "C" {

void *fftwf_malloc(size_t n) { return aligned_alloc(64, (n + 63) & ~(size_t)63); }
float *fftwf_alloc_real(size_t n) { return static_cast<float *>(fftwf_malloc(n * sizeof(float))); }
kq_cfloat *fftwf_alloc_complex(size_t n) { return static_cast<kq_cfloat *>(fftwf_malloc(n * sizeof(kq_cfloat))); }
void fftwf_free(void *p) { free(p); }

kq_fftwf_plan_s *fftwf_plan_dft_1d(int n, kq_cfloat *in, kq_cfloat *out, int sign, unsigned) {
  return fftw_plan_make(n, 0, sign < 0 ? -1 : +1, in, out);
}
kq_fftwf_plan_s *fftwf_plan_dft_r2c_1d(int n, float *in, kq_cfloat *out, unsigned) { return fftw_plan_make(n, 1, -1, in, out); }
kq_fftwf_plan_s *fftwf_plan_dft_c2r_1d(int n, kq_cfloat *in, float *out, unsigned) { return fftw_plan_make(n, 2, +1, in, out); }

void fftwf_execute(const kq_fftwf_plan_s *cp) {
  auto *p = const_cast<kq_fftwf_plan_s *>(cp);
  if (!p) return;
  kq::DeviceScope dev_scope_(ctx().device);
  DevCtx &c = ctx();
  int const n = p->n;
  const void *src = p->in;
  if (p->kind == 1) {  // real samples in
    const float *x = static_cast<const float *>(p->in);
    for (int i = 0; i < n; i++) p->stage[i] = make_float2(x[i], 0.f);
    src = p->stage.data();
  } else if (p->kind == 2) {  // n/2 + 1 bins in: Hermitian extension, DC and Nyquist taken as real (FFTW's c2r)
    const float2 *X = static_cast<const float2 *>(p->in);
    p->stage[0] = make_float2(X[0].x, 0.f);
    p->stage[n / 2] = make_float2(X[n / 2].x, 0.f);
    for (int k = 1; k < n / 2; k++) {
      p->stage[k] = X[k];
      p->stage[n - k] = make_float2(X[k].x, -X[k].y);
    }
    src = p->stage.data();
  }
  float2 *tw = twiddles(p->log2T);
  std::lock_guard<std::mutex> lk(c.mu);  // one transform at a time on the context's stream
  if (hipMemcpyAsync(p->d_in, src, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, c.stream) != hipSuccess) return;
  if (n > 16384) {
    if (kq::launch_fft_large(c.stream, p->d_in, p->d_out, p->d_tmp, n, p->sign, tw, p->log2T)) return;
  } else {
    kq::launch_fft_single(c.stream, p->d_in, p->d_out, p->dim, p->sign, tw, p->log2T);
  }
  if (p->kind == 2) {
    if (hipMemcpyAsync(p->stage.data(), p->d_out, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, c.stream) != hipSuccess) return;
    if (hipStreamSynchronize(c.stream) != hipSuccess) return;
    float *y = static_cast<float *>(p->out);
    for (int i = 0; i < n; i++) y[i] = p->stage[i].x;
    return;
  }
  size_t const bins = p->kind == 1 ? (size_t)n / 2 + 1 : (size_t)n;
  if (hipMemcpyAsync(p->out, p->d_out, bins * sizeof(float2), hipMemcpyDeviceToHost, c.stream) != hipSuccess) return;
  (void)hipStreamSynchronize(c.stream);
}

void fftwf_destroy_plan(kq_fftwf_plan_s *p) {
  if (!p) return;
  kq::DeviceScope dev_scope_(ctx().device);
  (void)hipFree(p->d_in);
  (void)hipFree(p->d_out);
  (void)hipFree(p->d_tmp);
  delete p;
}

// main.c:102-103,183-184: FFTW's wisdom and threading have no counterpart here
int fftwf_import_system_wisdom(void) { return 1; }
void fftwf_make_planner_thread_safe(void) {}
int fftwf_init_threads(void) { return 1; }
void fftwf_plan_with_nthreads(int) {}

}  // extern "C"
