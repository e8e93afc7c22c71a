// kq_radio.cpp -- the demodulator thread entry points of radio.h:235-237 (include/ka9q_hip_radio.h).
//
// What the reference's threads do per block (fm.c:72-174, am.c:43-79, linear.c:114-311) is done by a channel bank
// of one channel: the bank takes the master's device-resident input window of the block (already mixed by
// proc_samples, radio.c:132-139, so its own oscillators stay at rest), runs slave + compute_n0 + demodulator on the
// GPU and hands back audio and status.  The thread keeps the reference's shell: a slave made with
// create_filter_output (so that demod->filter.out, its response, noise_gain and output.c stay what callers of the
// reference find there), the blocking wait in execute_filter_output, the poll of demod->terminate once per block,
// the hand-off through the host program's send_mono_output / send_stereo_output.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/ka9q_hip.h"
#include "../../include/ka9q_hip_radio.h"
#include "kq_device.hpp"

// The host program provides these two (audio.c); weak so that the library loads without them (tests that never start a
// demodulator thread, the Python bindings).
extern "C" int send_mono_output(struct demod *, const float *, int) __attribute__((weak));
extern "C" int send_stereo_output(struct demod *, const float *, int) __attribute__((weak));

namespace {

// struct demod is shared with the host program's other threads the way the reference shares it -- plain fields, no lock
// (display.c:161 writes filter.low / high, set_mode writes terminate, radio.c:336): the fields those threads write are
// read here with relaxed atomic loads, and the slave pointer they pick up is published with a release store, so that the
// protocol is defined behaviour (and clean under ThreadSanitizer, tests/tsan) without changing the layout.  `terminate`
// is read with acquire: the thread frees its slave once it has seen the flag, and whatever the thread that set the flag
// did to that slave before (set_filter, noise_gain) must be over by then -- which a release store of the flag hands on.
template <class T>
inline T shared_load(const T &v) {
  T t;
  __atomic_load(const_cast<T *>(&v), &t, __ATOMIC_RELAXED);
  return t;
}
inline int told_to_stop(const struct demod *demod) { return __atomic_load_n(const_cast<int *>(&demod->terminate), __ATOMIC_ACQUIRE); }
inline void publish_slave(struct demod *demod, struct filter_out *s) { __atomic_store_n(&demod->filter.out, s, __ATOMIC_RELEASE); }

struct Session {
  struct demod *demod = nullptr;
  struct filter_out *slave = nullptr;
  kq_bank *bank = nullptr;
  float2 *d_spectrum = nullptr;
  int dev = -1;
  unsigned olen = 0;
  // what the bank channel was last told
  float low = 0, high = 0, beta = 0;
  int isb = 0, channels = 1;
  double shift_freq = 0;
  int type = KQ_LINEAR_DEMOD;  // which of the three threads this is
  bool host_n0 = false;  // N = 32768: the bank's split path has no compute_n0; the single-spectrum kernel supplies it
  bool have_block = false;  // next_block: the master's block number of the window demodulated last
  unsigned last_block = 0;

  ~Session() {
    if (bank) kq_bank_destroy(bank);
    if (d_spectrum) {
      kq::DeviceScope scope(dev);
      (void)hipFree(d_spectrum);
    }
    if (slave) delete_filter_output(slave);
  }
};

void fail(struct demod *demod, const char *what) {
  fprintf(stderr, "ka9q_hip: demodulator thread: %s (%s)\n", what, kq_last_error());
  publish_slave(demod, NULL);
}

// Prologue common to the three threads: slave (for the caller-visible state) + bank of one
bool start(Session &s, struct demod *demod, int demod_type, enum filtertype out_type, float edge_scale) {
  s.demod = demod;
  s.type = demod_type;
  struct filter_in *m = demod->filter.in;
  if (!m || demod->filter.decimate <= 0 || demod->input.samprate <= 0) {
    fail(demod, "demod->filter.in, filter.decimate or input.samprate not set");
    return false;
  }
  if (!send_mono_output || !send_stereo_output) {
    fail(demod, "the host program does not define send_mono_output / send_stereo_output");
    return false;
  }
  s.slave = create_filter_output(m, NULL, (unsigned)demod->filter.decimate, out_type);  // fm.c:33, am.c:39, linear.c:77
  if (!s.slave) {
    fail(demod, "create_filter_output failed");
    return false;
  }
  publish_slave(demod, s.slave);
  float const low0 = shared_load(demod->filter.low), high0 = shared_load(demod->filter.high), beta0 = shared_load(demod->filter.kaiser_beta);
  set_filter(s.slave, edge_scale * low0, edge_scale * high0, beta0);
  s.olen = s.slave->olen;
  s.dev = kq::compat_master_device();

  kq_bank_config bc;
  memset(&bc, 0, sizeof bc);
  bc.samprate = (unsigned)demod->input.samprate;
  bc.L = m->ilen;
  bc.M = m->impulse_length;
  bc.decimate = (unsigned)demod->filter.decimate;
  bc.max_channels = 1;
  bc.max_blocks = 1;
  bc.device = s.dev;
  bc.gain_factor = 1.f;
  {
    size_t const Nm = (size_t)m->ilen + m->impulse_length - 1;
    s.host_n0 = Nm > 16384 && Nm != 65536;  // the bank computes it up to 16384 points and at 65536 (kq_full16k.hip)
  }
  bc.compute_n0 = s.host_n0 ? 0 : 1;
  bc.fwd_mode = KQ_FWD_FULL;
  s.bank = kq_bank_create(&bc);
  if (!s.bank) {
    fail(demod, "kq_bank_create failed");
    return false;
  }
  kq_channel_config cc;
  memset(&cc, 0, sizeof cc);
  cc.demod_type = demod_type;
  cc.flat = demod->opt.flat;
  cc.isb = demod_type == KQ_LINEAR_DEMOD ? demod->filter.isb : 0;
  cc.channels = demod->output.channels == 2 ? 2 : 1;
  cc.low = low0;  // (an edge the user interface moves from here on is picked up by next_block's comparison)
  cc.high = high0;
  cc.kaiser_beta = beta0;
  cc.headroom = demod->agc.headroom;
  cc.hangtime = demod->agc.hangtime;
  cc.recovery_rate = demod->agc.recovery_rate;
  // the post-detection shift oscillator runs at the output rate: f = shift * D / Fs cycles per sample (radio.c:309)
  s.shift_freq = demod->shift.freq;
  cc.shift = s.shift_freq * demod->input.samprate / demod->filter.decimate;
  cc.pll = demod_type == KQ_LINEAR_DEMOD ? demod->opt.pll : 0;
  cc.square = demod_type == KQ_LINEAR_DEMOD ? demod->opt.square : 0;
  if (kq_bank_add_channel(s.bank, &cc) != 0) {
    fail(demod, "kq_bank_add_channel failed");
    return false;
  }
  // sig.n0 lives in struct demod, not in the thread: a thread started by set_mode goes on smoothing from the value
  // the last one left (fm.c:78-82, am.c:46-49, linear.c:123-126 test it for NaN, nothing ever resets it)
  if (!s.host_n0 && kq_bank_set_n0(s.bank, 0, demod->sig.n0)) {
    fail(demod, "kq_bank_set_n0 failed");
    return false;
  }
  s.low = cc.low;
  s.high = cc.high;
  s.beta = cc.kaiser_beta;
  s.isb = cc.isb;
  s.channels = cc.channels;
  kq::DeviceScope scope(s.dev);
  size_t const N = (size_t)m->ilen + m->impulse_length - 1;
  if (hipMalloc((void **)&s.d_spectrum, N * sizeof(float2)) != hipSuccess) {
    fail(demod, "device allocation failed");
    return false;
  }
  return true;
}

// One block: wait for the master (filter.c:195-199), run the bank on that block's window.  false: stop the thread.
bool next_block(Session &s, kq_chan_status *st, std::vector<float> &audio, size_t *nout) {
  struct demod *demod = s.demod;
  // filter edges may change under us (display.c:163, 950 call set_filter on demod->filter.out after updating these)
  float const low = shared_load(demod->filter.low), high = shared_load(demod->filter.high), beta = shared_load(demod->filter.kaiser_beta);
  if (low != s.low || high != s.high || beta != s.beta) {
    s.low = low;
    s.high = high;
    s.beta = beta;
    if (kq_bank_set_filter(s.bank, 0, s.low, s.high, s.beta)) return false;
  }
  // The master does not wait for its consumers (filter.c:146-172): by the time this thread wakes it may have queued the
  // next block already, and the spectrum copied here is then that one.  A block is demodulated once: when the copy
  // turns out to be the block done last time round, wait for the next.  (A thread that falls further behind skips
  // blocks, as the reference's equality test on blocknum does, filter.c:195-199.)
  for (;;) {
    if (execute_filter_output(s.slave)) return false;  // blocks until the master has a new block; refreshes output.c
    if (told_to_stop(demod)) return false;
    unsigned blk = 0;
    if (kq::compat_snapshot_spectrum(demod->filter.in, s.d_spectrum, &blk) < 0) return false;
    if (s.have_block && blk == s.last_block) continue;
    s.have_block = true;
    s.last_block = blk;
    break;
  }
  // slave, compute_n0 and demodulator on the master's own transform (filter.c:206-250, radio.c:383-425): the forward
  // transform is done once, by execute_filter_input, as in the reference
  if (kq_bank_process_spectrum(s.bank, s.d_spectrum, 1) != 1) return false;
  audio.resize(2 * (size_t)s.olen);
  if (kq_bank_pull_audio(s.bank, 0, 0, audio.data(), audio.size(), nout)) return false;
  if (kq_bank_pull_status(s.bank, 0, 0, st)) return false;
  if (s.host_n0) {
    // radio.c:383-425 on the master's resident spectrum, then the demodulator's smoothing (fm.c:78-82: 0.01;
    // am.c:46-49, linear.c:123-126: 0.001; the first value is taken as it comes)
    float const raw = kq_compat_compute_n0(demod->filter.in, demod->input.samprate, s.low, s.high);
    float const k = s.type == KQ_FM_DEMOD ? 0.01f : 0.001f;
    st->n0 = std::isnan(demod->sig.n0) ? raw : demod->sig.n0 + k * (raw - demod->sig.n0);
  }
  return true;
}

void finish(Session &s) {
  struct demod *demod = s.demod;
  publish_slave(demod, NULL);  // fm.c:182, am.c:81, linear.c:320 (the Session's destructor deletes the slave)
}

}  // namespace

extern "C" {

// fm.c:21-186
void *demod_fm(void *arg) {
  struct demod *const demod = (struct demod *)arg;
  if (!demod) return NULL;
  Session s;
  float const dsamprate = (float)demod->input.samprate / demod->filter.decimate;  // fm.c:27
  demod->sig.pdeviation = 0;                                                      // fm.c:28-30
  demod->sig.foffset = 0;
  demod->output.channels = 1;
  if (!start(s, demod, KQ_FM_DEMOD, COMPLEX, 1.f / dsamprate)) return NULL;
  std::vector<float> audio;
  kq_chan_status st;
  size_t n = 0;
  while (!told_to_stop(demod)) {
    if (!next_block(s, &st, audio, &n)) {
      if (!told_to_stop(demod)) fail(demod, "block failed");  // (told to stop while waiting: not a failure)
      break;
    }
    demod->sig.n0 = st.n0;  // fm.c:78-82 (smoothed on the device with the same recurrence)
    demod->sig.bb_power = st.bb_power;
    demod->sig.snr = st.snr;
    if (st.squelch_count < 1) {  // fm.c:145-154: only while the squelch is fully open
      demod->sig.foffset = st.foffset;
      demod->sig.pdeviation = st.pdeviation;
    }
    demod->sig.plfreq = st.plfreq;  // pltask, fm.c:276-281
    send_mono_output(demod, audio.data(), (int)s.olen);
  }
  finish(s);
  return NULL;
}

// am.c:15-83
void *demod_am(void *arg) {
  struct demod *const demod = (struct demod *)arg;
  if (!demod) return NULL;
  Session s;
  float const samptime = demod->filter.decimate / (float)demod->input.samprate;  // am.c:21
  demod->agc.gain = powf(10.f, 80.f / 20.f);                                     // am.c:30
  demod->output.channels = 1;
  if (!start(s, demod, KQ_AM_DEMOD, COMPLEX, samptime)) return NULL;
  std::vector<float> audio;
  kq_chan_status st;
  size_t n = 0;
  while (!told_to_stop(demod)) {
    if (!next_block(s, &st, audio, &n)) {
      if (!told_to_stop(demod)) fail(demod, "block failed");  // (told to stop while waiting: not a failure)
      break;
    }
    demod->sig.n0 = st.n0;  // am.c:46-49
    demod->agc.gain = st.agc_gain;
    send_mono_output(demod, audio.data(), (int)s.olen);
    demod->sig.bb_power = st.bb_power;  // am.c:78, after the hand-off
  }
  finish(s);
  return NULL;
}

// linear.c:21-322
void *demod_linear(void *arg) {
  struct demod *const demod = (struct demod *)arg;
  if (!demod) return NULL;
  Session s;
  demod->opt.loop_bw = 1;                                                                // linear.c:26
  float const samptime = (float)demod->filter.decimate / (float)demod->input.samprate;  // linear.c:29
  demod->agc.gain = powf(10.f, 100.f / 20.f);                                            // linear.c:39
  demod->sig.snr = 0;                                                                    // linear.c:74
  if (!start(s, demod, KQ_LINEAR_DEMOD, demod->filter.isb ? CROSS_CONJ : COMPLEX, samptime)) return NULL;
  std::vector<float> audio;
  kq_chan_status st;
  size_t n = 0;
  while (!told_to_stop(demod)) {
    // linear.c:117-120: the ISB flag is copied to the slave before every block; output.channels is read after it
    int const isb = demod->filter.isb != 0, channels = demod->output.channels == 2 ? 2 : 1;
    if (isb != s.isb || channels != s.channels) {
      s.slave->out_type = isb ? CROSS_CONJ : COMPLEX;
      if (kq_bank_set_linear_options(s.bank, 0, isb, channels)) {
        fail(demod, "kq_bank_set_linear_options failed");
        break;
      }
      s.isb = isb;
      s.channels = channels;
    }
    if (demod->shift.freq != s.shift_freq) {  // set_shift() (radio.c:304-311) since the last block
      s.shift_freq = demod->shift.freq;
      if (kq_bank_set_shift(s.bank, 0, s.shift_freq * demod->input.samprate / demod->filter.decimate)) {
        fail(demod, "kq_bank_set_shift failed");
        break;
      }
    }
    if (!next_block(s, &st, audio, &n)) {
      if (!told_to_stop(demod)) fail(demod, "block failed");  // (told to stop while waiting: not a failure)
      break;
    }
    demod->sig.n0 = st.n0;  // linear.c:123-126
    if (demod->opt.pll) {   // linear.c:157-246
      demod->sig.cphase = st.cphase;
      demod->sig.foffset = st.foffset;
      demod->sig.pll_lock = st.pll_lock;
      demod->sig.lock_timer = (float)st.lock_count;
    }
    demod->agc.gain = st.agc_gain;
    if (demod->shift.freq != 0) {
      // the shift oscillator advanced olen steps on the device (linear.c:283-289): keep the caller's copy in step
      pthread_mutex_lock(&demod->shift.mutex);
      double const turns = demod->shift.freq * (double)s.olen;
      double const a = 2 * M_PI * (turns - std::floor(turns));
      double const pr = __real__ demod->shift.phasor, pi = __imag__ demod->shift.phasor;
      __real__ demod->shift.phasor = pr * std::cos(a) - pi * std::sin(a);
      __imag__ demod->shift.phasor = pr * std::sin(a) + pi * std::cos(a);
      pthread_mutex_unlock(&demod->shift.mutex);
    }
    if (channels == 1)
      send_mono_output(demod, audio.data(), (int)s.olen);  // linear.c:291-296
    else
      send_stereo_output(demod, audio.data(), (int)s.olen);  // linear.c:297-300
    demod->sig.bb_power = st.bb_power;                       // linear.c:302
    demod->sig.snr = st.snr;                                 // linear.c:304-309
  }
  finish(s);
  return NULL;
}

}  // extern "C"
