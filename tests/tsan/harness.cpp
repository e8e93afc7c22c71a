// ThreadSanitizer drive of the compat surface's threading protocol: the library's own
// kq_compat.cpp and kq_radio.cpp, compiled for the CPU against tests/tsan/mock/hip/hip_runtime.h, with mock transforms
// and a mock channel bank, used the way radio.c / display.c use the reference's filter.c:
//   one producer          execute_filter_input per block (radio.c:140-146)
//   three consumers       the demodulator thread entry points demod_fm / demod_am / demod_linear (radio.h:235-237): slave
//                         creation, set_filter, the blocking wait of execute_filter_output (filter.c:195-199), the poll of
//                         demod->terminate once per block, the hand-off
//   one "user interface"  set_filter on demod->filter.out while the threads run (display.c:161-177, filter.c:538-543),
//                         set_shift-like field updates
// then terminate + join as set_mode does (radio.c:336-338).  Test infrastructure only: nothing here is product code.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/ka9q_hip.h"
#include "../../include/ka9q_hip_radio.h"
#include "../../ka9q_sdr_amd/csrc/kq_design.hpp"
#include "../../ka9q_sdr_amd/csrc/kq_device.hpp"

// ---- mock device work: each launcher is one operation of the in-order stream
namespace kq {
void launch_fft_single(hipStream_t, const float2 *in, float2 *out, const FftDim &d, int, const float2 *, int) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  for (int i = 0; i < d.n; i++) out[i] = in[i];
}
int launch_fft_large(hipStream_t, const float2 *in, float2 *out, float2 *tmp, int N, int, const float2 *, int) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  for (int i = 0; i < N; i++) tmp[i] = in[i];
  for (int i = 0; i < N; i++) out[i] = tmp[i];
  return 0;
}
FftDim fft_dim(int n, bool *ok) {  // (the mock transforms copy: a plan is its size)
  FftDim d{};
  d.n = n;
  d.log2n = -1;
  if (n >= 1 && (n & (n - 1)) == 0)
    for (d.log2n = 0; (1 << d.log2n) < n;) d.log2n++;
  if (ok) *ok = true;
  return d;
}
bool fft_size_ok(int n) { return n >= 2 && (n & 1) == 0; }
void launch_n0_single(hipStream_t, const float2 *X, int N, int, float, float, float *out) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  float acc = 0;
  for (int i = 0; i < N; i++) acc += X[i].x * X[i].x + X[i].y * X[i].y;
  *out = acc / N;
}
void launch_slave_single(hipStream_t, const float2 *X, const float2 *H, float2 *out, int, int Ndec, int, int, const float2 *, int) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  for (int i = 0; i < Ndec; i++) out[i] = make_float2(X[i].x * H[i].x - X[i].y * H[i].y, X[i].x * H[i].y + X[i].y * H[i].x);
}
int make_kaiser(float *w, unsigned M, float) {
  for (unsigned i = 0; i < M; i++) w[i] = 1.f;
  return 0;
}
int window_filter(int, int, std::vector<cfloat> &, float) { return 0; }
int window_rfilter(int, int, std::vector<cfloat> &, float) { return 0; }
std::vector<cfloat> design_response(int N, int L_dec, int M_dec, int, float low, float high, float, float *ng) {
  std::vector<cfloat> r(L_dec + M_dec - 1);
  for (size_t i = 0; i < r.size(); i++) r[i] = cfloat((low + high) / N, 0.f);
  if (ng) *ng = 1.f;
  return r;
}
}  // namespace kq

// ---- mock channel bank of one (include/ka9q_hip.h): reads the window it is handed as the kernels would
struct kq_bank {
  kq_bank_config cfg;
  float acc = 0;
};
static thread_local std::string g_err;
extern "C" {
const char *kq_last_error(void) { return g_err.c_str(); }
kq_bank *kq_bank_create(const kq_bank_config *c) {
  kq_bank *b = new kq_bank();
  b->cfg = *c;
  return b;
}
int kq_bank_destroy(kq_bank *b) {
  delete b;
  return 0;
}
int kq_bank_add_channel(kq_bank *, const kq_channel_config *) { return 0; }
int kq_bank_set_filter(kq_bank *, int, float, float, float) { return 0; }
int kq_bank_set_n0(kq_bank *, int, float) { return 0; }
int kq_bank_set_shift(kq_bank *, int, double) { return 0; }
int kq_bank_set_linear_options(kq_bank *, int, int, int) { return 0; }
int kq_bank_process_resident(kq_bank *b, const void *iq, unsigned nblocks) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  const float2 *x = static_cast<const float2 *>(iq);
  unsigned const N = b->cfg.L + b->cfg.M - 1;
  float a = 0;
  for (unsigned i = 0; i < N; i++) a += x[i].x;
  b->acc = a;
  return (int)nblocks;
}
int kq_bank_process_spectrum(kq_bank *b, const void *spec, unsigned nblocks) {
  std::lock_guard<std::mutex> lk(mock_stream_mutex());
  const float2 *x = static_cast<const float2 *>(spec);
  unsigned const N = b->cfg.L + b->cfg.M - 1;
  float a = 0;
  for (unsigned i = 0; i < N; i++) a += x[i].x;
  b->acc = a;
  return (int)nblocks;
}
int kq_bank_pull_audio(kq_bank *b, int, unsigned, float *dst, size_t cap, size_t *n) {
  size_t const olen = b->cfg.L / b->cfg.decimate;
  for (size_t i = 0; i < cap; i++) dst[i] = b->acc;
  if (n) *n = olen;
  return 0;
}
int kq_bank_pull_status(kq_bank *b, int, unsigned, kq_chan_status *st) {
  memset(st, 0, sizeof *st);
  st->nout = (int)(b->cfg.L / b->cfg.decimate);
  st->n0 = 1e-9f;
  st->agc_gain = 1.f;
  return 0;
}
}
void kq_internal_set_error(const char *fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ---- the host program's side: audio hand-off (audio.c:32,82)
static std::atomic<long> g_mono{0}, g_stereo{0};
extern "C" int send_mono_output(struct demod *, const float *buf, int n) {
  float a = 0;
  for (int i = 0; i < n; i++) a += buf[i];
  g_mono += (a == a);
  return 0;
}
extern "C" int send_stereo_output(struct demod *, const float *buf, int n) {
  float a = 0;
  for (int i = 0; i < 2 * n; i++) a += buf[i];
  g_stereo += (a == a);
  return 0;
}

static std::atomic<int> g_running{0};
struct Thread {
  struct demod *d;
  void *(*entry)(void *);
};
static void *run_demod(void *arg) {
  Thread *t = static_cast<Thread *>(arg);
  g_running++;
  t->entry(t->d);
  g_running--;
  return nullptr;
}

int main() {
  unsigned const L = 256, M = 257, D = 4;
  struct filter_in *master = create_filter_input(L, M, COMPLEX);
  if (!master) return 2;
  struct demod *dm = static_cast<struct demod *>(calloc(3, sizeof(struct demod)));
  void *(*entries[3])(void *) = {demod_fm, demod_am, demod_linear};
  Thread th[3];
  pthread_t tid[3], ui;
  for (int k = 0; k < 3; k++) {
    struct demod *d = &dm[k];
    d->input.samprate = 192000;
    d->filter.in = master;
    d->filter.L = (int)L;
    d->filter.M = (int)M;
    d->filter.decimate = (int)D;
    d->filter.low = -5000;
    d->filter.high = 5000;
    d->filter.kaiser_beta = 3;
    d->agc.headroom = 0.17f;
    d->output.channels = k == 2 ? 2 : 1;
    d->sig.n0 = NAN;
    pthread_mutex_init(&d->shift.mutex, nullptr);
    pthread_mutex_init(&d->second_LO.mutex, nullptr);
    pthread_mutex_init(&d->doppler.mutex, nullptr);
    th[k] = Thread{d, entries[k]};
    pthread_create(&tid[k], nullptr, run_demod, &th[k]);
  }
  // the user interface: retunes the filters of running demodulators (display.c:161-177)
  static std::atomic<bool> ui_stop{false};
  struct Ui {
    struct demod *dm;
  } uiarg{dm};
  pthread_create(
      &ui, nullptr,
      [](void *p) -> void * {
        struct demod *dm = static_cast<Ui *>(p)->dm;
        for (int it = 0; !ui_stop.load(); it++) {
          for (int k = 0; k < 3; k++) {
            struct filter_out *out = __atomic_load_n(&dm[k].filter.out, __ATOMIC_ACQUIRE);
            if (!out) continue;
            float const lo = -4000.f - (it % 7) * 100.f, hi = 4000.f + (it % 5) * 100.f;
            __atomic_store(&dm[k].filter.low, &lo, __ATOMIC_RELAXED);
            __atomic_store(&dm[k].filter.high, &hi, __ATOMIC_RELAXED);
            set_filter(out, lo / 48000.f, hi / 48000.f, 3.0f);
            (void)noise_gain(out);  // radio_status.c:171 reads it from yet another thread; here after our own set_filter
          }
          usleep(200);
        }
        return nullptr;
      },
      &uiarg);
  // the producer (radio.c:106-147): fill the user area, run the master
  long blocks = 0;
  for (; blocks < 600; blocks++) {
    for (unsigned i = 0; i < L; i++) master->input.c[i] = (float)(blocks + i) * 1e-3f;
    if (execute_filter_input(master)) return 3;
    if (blocks % 8 == 0) usleep(100);  // sometimes the consumers keep up, sometimes the master runs ahead of them
  }
  ui_stop = true;
  pthread_join(ui, nullptr);
  for (int k = 0; k < 3; k++) {
    int one = 1;
    __atomic_store(&dm[k].terminate, &one, __ATOMIC_RELEASE);  // set_mode, radio.c:336-338 (after the last touch of the slave)
  }
  while (g_running.load() > 0) {  // the threads look at terminate once per block: keep the blocks coming
    for (unsigned i = 0; i < L; i++) master->input.c[i] = 0;
    if (execute_filter_input(master)) return 3;
    blocks++;
    usleep(100);
  }
  for (int k = 0; k < 3; k++) pthread_join(tid[k], nullptr);
  delete_filter_input(master);
  printf("tsan harness: %ld blocks produced, %ld mono and %ld stereo hand-offs\n", blocks, g_mono.load(), g_stereo.load());
  free(dm);
  return (g_mono.load() > 100 && g_stereo.load() > 50) ? 0 : 4;
}
