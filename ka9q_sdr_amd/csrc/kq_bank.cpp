// kq_bank.cpp -- host side of the channel bank: the C ABI of include/ka9q_hip.h.
//
// Host responsibilities (control plane, once per call or per retune -- never per sample):
//   * NCO bookkeeping in closed form.  The reference advances a complex-double phasor one sample at
//     a time (osc.c:39-51); here each oscillator is (phase, step, sweep) at a reference sample index
//     and the kernels evaluate phase(n) = phase + step*k + sweep*k*(k-1)/2 themselves.
//     In the steady state (nothing set, added or removed since the call before) the device advances every channel's
//     planes itself and the host touches no per-channel state; a retuned channel travels as one 72-byte patch record.
//   * the control plane: set_filter / set_mode / add / remove / set_n0 ... never touch the device.  They gather write
//     records (CtlQueue: filter side, demodulator side) and design jobs (DesignQueue) in pinned memory; the next call
//     applies them with one launch each, in front of its own kernels, behind the calls in flight.  A new filter's
//     response is designed on the bank's stream by kq_design.hip's kernel, straight into the channel's row.
//   * ring management, kernel sequencing, streaming host I/O (copy streams, pinned planes), RTP in and out
//   * one lock per handle (every entry point; let go of while an entry point waits for the device)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kq_design.hpp"
#include "kq_device.hpp"

namespace {

thread_local std::string g_err;

void set_err(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

// launch errors are sticky until read: name the launch group that failed
#define LAUNCH_CHECK(what)                                                          \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      set_err("kernel launch failed in %s: %s", what, hipGetErrorString(e_));       \
      return -1;                                                                    \
    }                                                                               \
  } while (0)

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                       \
    }                                                                                  \
  } while (0)

// One NCO in closed form.  `frozen` mirrors osc.c:43: an oscillator whose set frequency is zero
// never advances, whatever its sweep rate.
struct Osc {
  bool init = false;
  bool frozen = true;
  double phase = 0;  // turns at sample n_ref
  double f = 0;      // cycles/sample applied between n_ref and n_ref+1
  double r = 0;      // cycles/sample^2
  double set_f = 0;  // value last passed to set() -- what osc->freq holds in the reference
  int64_t n_ref = 0;

  double phase_at(int64_t n) const {
    if (frozen) return phase;
    double const k = (double)(n - n_ref);
    return phase + f * k + r * (0.5 * k * (k - 1.0));
  }
  double step_at(int64_t n) const { return frozen ? 0.0 : f + r * (double)(n - n_ref); }
  double sweep() const { return frozen ? 0.0 : r; }
  // set_osc (osc.c:22-36): keeps the phase when already initialised
  void set(double freq, double rate, int64_t now) {
    if (init) {
      phase = phase_at(now);
      phase -= std::floor(phase);
    } else {
      phase = 0;
      init = true;
    }
    n_ref = now;
    set_f = freq;
    frozen = (freq == 0);
    f = freq;
    r = rate;
  }
  // move the reference point forward so k stays small (no change of the generated sequence)
  void rebase(int64_t now) {
    if (!init || frozen) {
      n_ref = now;
      return;
    }
    double const p = phase_at(now);
    f = step_at(now);
    phase = p - std::floor(p);
    n_ref = now;
  }
};

struct HostChan {
  kq_channel_config cfg;
  Osc lo2, dop, shift;
  // oscillators as they were before a retune that has not reached the kernels yet: the M-1 history samples of
  // the next block were mixed with these (radio.c:132-139)
  Osc lo2_old, dop_old;    // the oscillators before the last retune (the history planes) ...
  // ... and the ones before the retunes before that, while samples of theirs are still in the history: [0] the transition
  // before the last, [l + 1] the one before [l]
  Osc lo2_oldx[kq::kOldLevels], dop_oldx[kq::kOldLevels];
  bool retuned = false;
  // ... and how many samples from the start of the NEXT call's first window still carry the old oscillators (ChanDev::hist_len):
  // M - 1 when the retune happens; a call of n blocks takes n L off it; the channel stays `retuned` while any are left
  int64_t hist_old = 0;
  int64_t hist_oldx[kq::kOldLevels] = {};  // the same for lo2_oldx / dop_oldx (older: fewer samples; 0 ends the list)
  int hist_dev = -1;   // what hist_len[c] on the device was last told
  // ... and hist2_len[kOldLevels c + l]; -1 = never (the words of a slot taken over from a removed channel are whatever that
  // one left: the first retune writes every level)
  int histx_dev[kq::kOldLevels] = {-1, -1, -1, -1};
  static_assert(kq::kOldLevels == 4, "histx_dev's initialiser");
  bool active = true;  // false: a hole left by kq_bank_remove_channel, reused by the next kq_bank_add_channel
  kq_out_rtp_state out_rtp{};  // demod->output.rtp + output.silent (audio.c:32-132)
  int out_type;
  std::vector<kq::cfloat> resp, aresp;
  float noise_gain;
  int pll_slot = -1;  // carrier-tracking channels: the slot of the loop's state and ring (pll_acquire)
  int n0slot = -1;  // which of the bank's compute_n0 mask sets this channel uses (shared by all channels with its edges)
  // where the channel stands in the bank's lists (kq_bank: list_host[lk][lpos], list_active_host[apos]); lk = 3: on the
  // carrier-loop list, -1: on none
  int lk = -1, lpos = -1, apos = -1;
  bool patched = false;  // an oscillator of this channel has been set since the last call (it is on the bank's patch list)
  double r_eff = 0;      // sweep of its input oscillators as the launch decisions last saw it (cycles / sample^2)
};

struct EventPair {
  hipEvent_t a, b;
};

}  // namespace

struct kq_bank {
  // One lock per handle, taken by every entry point: a receiver thread in its process / push / pull loop and an operator's
  // thread changing filters, modes and frequencies (display.c / radio_status.c beside the demodulator threads of the
  // reference) may share a bank.  The entry points that wait for the device to catch up (kq_bank_pull_wait, _host_io_wait,
  // _sync) let go of it while they wait.  Recursive: some entry points are built from others.
  std::recursive_mutex mu;
  kq_bank_config cfg;
  kq::Geom g;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int fwd_mode = KQ_FWD_FULL;
  // The demodulators are latency-bound and independent of the next batch's filter pass, so they run on a
  // second stream: filter(k+1) overlaps demod(k).  Planes the two stages hand over are double buffered.
  hipStream_t stream2 = nullptr;   // demodulators of a call that overlaps the next call's filter pass (== stream: never)
  int overlap_mode = -1;           // KQ_DEMOD_OVERLAP: 0 never, 1 always, unset (-1) per call, see run_blocks
  bool demod_overlapped[2] = {false, false};  // by call parity: ev_demod_done[parity] was recorded on stream2
  bool pulled_since_call = false;  // kq_bank_pull_planes_async since the last call: the host streams planes out
  // front-end packet bookkeeping (struct rtp_state + demod->input.samples)
  kq_rtp_counters rtp{};
  bool rtp_init = false;
  bool rtp_retry = false;          // the last datagram was sent back with -2: the same one comes again
  uint16_t rtp_retry_seq = 0;
  uint32_t rtp_retry_ts = 0;
  hipEvent_t ev_demod_done[2] = {nullptr, nullptr};
  kq::Planes pl2[2];
  double *osc_dev2[2] = {nullptr, nullptr};
  // generic FM path: detected samples of a call [C][B][olen] and the de-emphasis filter's history [C][Mdec-1],
  // double buffered by call parity (read by every block-0 workgroup while the last block writes the next one)
  float *fmout = nullptr;
  float *fm_hist[2] = {nullptr, nullptr};
  int fm_hist_cur = 0;
  uint64_t calls = 0;

  float2 *ring[2] = {nullptr, nullptr};
  int cur = 0;
  size_t ring_cap = 0;   // samples, including the M-1 history
  size_t pending = 0;    // new samples in the ring beyond the history
  std::vector<unsigned char> zero_tail;  // per pending block: 1 if its last sample came from a zero fill
  size_t zero_run = 0;                   // trailing zero-fill samples of the partially filled block
  bool partial_ends_in_zero = false;

  float2 *tw = nullptr;
  float2 *chan_tw = nullptr;  // pruned path: per-channel twiddle tables
  bool chan_tw_dirty = true;
  kq::ChanDev chd;
  kq::Planes pl;
  int *list_dev[3] = {nullptr, nullptr, nullptr};  // fm, am, linear (without PLL)
  // carrier-tracking linear channels (linear.c:129-246): own list, 65536-sample search ring per channel
  // carrier-tracking channels: a slot each (state + 65536-sample ring + search scratch), handed out from chunks of
  // kq::kPllChunk that are allocated as the count grows; a slot stays with its channel until the channel leaves the set, so
  // adding or removing one moves nothing and waits for nothing (rounds 1-5: slot = rank, 64 at most, synchronous moves)
  static constexpr int kMaxPllChunks = 1024;
  int *list_pll_dev = nullptr;         // [max_channels]
  int *pll_slot_dev = nullptr;         // [max_channels] channel -> slot
  kq::PllChunk *pll_chunks_dev = nullptr;  // [kMaxPllChunks]
  std::vector<kq::PllChunk> pll_chunks;
  std::vector<int> pll_free;           // slots not in use, lowest last
  int *list_active_dev = nullptr;      // the active channels, for the filter launch, when remove_channel has left holes
  int *list_active_ds_dev = nullptr;   // the same list as the PCM stage reads it, on the demodulators' stream
  // Every active channel, in no particular order (the lists follow the channels' coming and going incrementally: a channel
  // that leaves is replaced by the list's last entry, one that comes is appended -- one or two 4-byte writes to the device's
  // copy instead of the list: at 32 768 channels a rebuilt list was 128 KiB over the link per change).  Used by the launches
  // only while there are holes (fewer entries than slots); a bank whose channels have ALL been removed never launches.
  std::vector<int> list_active_host;
  std::vector<int> list_pll_host;
  std::vector<int> list_host[3];
  bool lists_dirty = true;
  float *energy_state = nullptr;
  float2 *win_paired = nullptr;  // row-paired copy of a call's samples for k_filter_full16k (full16k_paired_supported)
  // N = 65536 full-spectrum path (four sibling workgroups per channel-block, kq_full16k.hip): what the siblings hand to
  // each other and to k_epilogue64k; big.err is pinned host memory the kernel writes when a sibling never showed up
  bool use64k = false;
  kq::Big64 big{};
  // per-call parameters (5 double planes of max_channels + max_blocks update flags) travel through
  // pinned staging slots so kq_bank_process never has to synchronise the stream
  static constexpr int kSlots = 4;
  unsigned char *stage_host[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  // One marker per call on the main stream, recorded behind the filter launch(es) of the call that used the slot: the
  // demodulator stream waits for it, the host waits for it before it refills the slot four calls later, and with
  // kq_bank_enable_timing it closes the filter's time interval, which stage_t0 opened (a marker costs the stream ~5 us
  // behind a long kernel, tools/marker_probe.hip; there were four per call)
  hipEvent_t stage_ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_t0[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  bool stage_timed[kSlots] = {false, false, false, false};
  // the host's side of that interval: when the opening marker was queued, how long until the closing one was, which call
  std::chrono::steady_clock::time_point stage_h0[kSlots];
  double stage_submit_ms[kSlots] = {0, 0, 0, 0};
  uint64_t stage_launch[kSlots] = {0, 0, 0, 0};
  int stage_next = 0;
  size_t stage_bytes = 0;
  float2 *spec_dump = nullptr;
  int spec_ch = -1;
  bool pcm_on = false;
  short *pcm = nullptr;       // [C][max_blocks][2*olen] int16, network byte order
  unsigned *pcm_mask = nullptr;  // [C][max_blocks]
  void *stage_dev = nullptr;  // staging for host-side raw I/Q before conversion
  size_t stage_cap = 0;
  // streaming host I/O (kq_bank_push_iq_async / kq_bank_pull_planes_async): copy streams of their own, two input staging
  // buffers, events that order them against the kernels
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  void *in_stage[2] = {nullptr, nullptr};
  size_t in_stage_cap[2] = {0, 0};
  bool in_used[2] = {false, false};  // in_ready[k] has been recorded at least once
  hipEvent_t in_ready[2] = {nullptr, nullptr}, in_free[2] = {nullptr, nullptr};
  int in_next = 0;
  // kq_bank_push_rtp's payloads gather in pinned host memory and travel as ONE asynchronous copy + conversion per run of
  // packets (flushed by kq_bank_process and by whatever else touches the ring): a datagram no longer costs a host wait
  // for everything the stream has queued -- with process calls of 1.5 ms in flight that wait was the end of real time
  unsigned char *acc_pin[2] = {nullptr, nullptr};
  size_t acc_cap = 0;        // bytes, each buffer
  int acc_cur = 0, acc_fmt = -1;
  size_t acc_n = 0;          // samples gathered in acc_pin[acc_cur]
  size_t acc_ring_off = 0;   // where in the ring (samples) the run starts
  hipEvent_t acc_read[2] = {nullptr, nullptr};  // the copy engine has read the buffer
  bool acc_read_set[2] = {false, false};
  hipEvent_t out_ready = nullptr;
  // one marker per queued plane copy, a ring of them: the next call's demodulators wait for the newest on the device, and a
  // streaming host waits for the one `lag` deliveries back (kq_bank_pull_wait) while newer calls are in flight
  static constexpr int kPullRing = 8;
  hipEvent_t pull_done[kPullRing] = {};
  uint64_t pulls = 0;        // plane copies queued so far; the newest one's marker is pull_done[(pulls - 1) % kPullRing]
  bool out_pending = false;  // a plane copy is queued that the next call's demodulators must wait for

  // compute_n0's lane masks depend on a channel's filter edges only, and a receiver's channels mostly share a handful
  // of filters: one mask set (2 KiB; N = 65536: 8 KiB) per distinct pair of edges, counted references, so that the masks
  // of tens of thousands of channels stay in the L2 instead of streaming 70 MB per block from memory
  std::map<std::pair<float, float>, int> n0slot_of;
  std::vector<int> n0slot_refs;                       // per slot; 0 = free
  std::vector<std::pair<float, float>> n0slot_key;    // per slot
  // Control-plane writes (per-channel parameters, responses, carried-state resets, channel lists) do not touch the device
  // when they are made: they gather in pinned host memory, in two queues, and the next process call applies each queue with
  // ONE small launch (k_ctl_apply) at the place in the stream order where its readers expect it --
  //   FILTER side: what the filter kernels read (responses, compute_n0 masks, the ISB flag, the filter launch's lists):
  //                on the main stream in front of the call's first kernel, behind the filter passes in flight;
  //   DEMOD side:  what the demodulators read and carry (gains, flags, squelch / AGC / filter state, their lists): on
  //                whichever stream the call's demodulators run, in front of them, behind the demodulators in flight.
  // The calls in flight keep the values they were queued with, nothing waits on the host or across streams, and a
  // change costs the device a few microseconds (as separate small copies on the stream each change cost 0.5-1 ms of
  // pipeline time at 32768 channels, tools/soak_realtime.py --only filter).
  struct CtlQueue {
    static constexpr size_t kBytes = 1u << 20, kMaxRec = 4096;
    static constexpr int kDepth = 4;  // the host runs up to three calls ahead of the device (kq_bank_pull_wait's lag + 1)
    unsigned char *buf[kDepth] = {};  // pinned; [records (32 B each, kMaxRec of them) | payloads]
    hipEvent_t applied[kDepth] = {};
    bool applied_set[kDepth] = {};
    int cur = 0;
    unsigned nrec = 0;
    size_t used = 0;  // payload bytes
    // The records of one launch are applied concurrently, one workgroup each: two writes to one place must not both be in
    // it.  A later write to a destination already in the queue replaces the earlier one on the host (destination -> record).
    std::unordered_map<unsigned long long, unsigned> at;
  };
  CtlQueue ctl[2];  // 0 filter side, 1 demod side
  // Responses are designed where they are used: kq_bank_set_filter / add_channel / set_mode gather design jobs, and the
  // next call launches ONE design kernel for them on the main stream in front of its filter pass, which writes each
  // response into its channel's row (kq_design.hip design_launch).  No copy back, no wait: on a bank at real time the
  // round trip of a design on a stream of its own came to 2.0-2.4 ms of host time per operation (its packets queue
  // behind the copy kernels that share its hardware queue; tools/soak_realtime.py).  The noise gain a design yields is
  // demodulator-side state: the kernel leaves it in ng_next[epoch parity][channel] and a device-to-device record of the
  // DEMOD queue moves it over in front of the call's demodulators.  The host's copy of a response (kq_bank_get_response)
  // is fetched when asked for.
  struct DesignQueue {
    static constexpr unsigned kMax = 1024;
    static constexpr int kDepth = 4;
    unsigned char *pin[kDepth] = {};  // pinned: [kMax jobs | kMax targets]
    hipEvent_t read[kDepth] = {};     // the launch that read pin[k] is over
    bool read_set[kDepth] = {};
    int cur = 0;
    std::vector<kq::DesignJob> jobs;
    std::vector<kq::DesignTarget> targets;
    std::unordered_map<int, unsigned> at;  // channel -> job: the later design of a channel replaces the earlier one
    unsigned max_jobs = kMax;              // what the scratch holds
    float2 *scratch = nullptr;             // max_jobs * Ndec
    float *ng_next = nullptr;              // [2][max_channels]
    hipEvent_t ng_moved[2] = {};           // the DEMOD-side records that read ng_next[p] have been applied
    bool ng_moved_set[2] = {false, false};
    int ng_to_record = -1;                 // parity whose records the next DEMOD flush applies
    unsigned long long epoch = 0;
  };
  DesignQueue dq;
  std::map<float, std::vector<kq::cfloat>> aresp_cache;  // FM audio response by Kaiser beta (fm.c:54-66: geometry fixed per bank)
  std::vector<HostChan> chans;
  // Steady state of the oscillators: nothing has been set, added or removed since the call before, so the per-call planes
  // follow from that call's on the device (k_block_energy_sum) and the host touches no per-channel state at all.
  // osc_dirty = false promises: the planes of the call before are valid for its window start (planes_n_w, planes_out_abs),
  // no channel has `retuned` set, and the cached launch decisions below still hold.
  bool osc_dirty = true;
  int64_t planes_n_w = 0, planes_out_abs = 0, rebased_at = 0;
  size_t refresh_next = 0;  // the channel whose closed forms the next steady call re-references first
  std::vector<int> ret_host;  // the channels whose windows of the call being staged hold samples of an old oscillator
  bool cache_any = false;
  // Retunes (kq_bank_set_second_lo / _doppler / _shift) leave the steady state intact: the channels touched since the last
  // call are on patch_list, and the next call advances everybody on the device as usual and then overwrites just those
  // channels' planes from a few records staged by the host (the patch role of k_block_energy_sum) -- a receiver that tracks Doppler on thousands
  // of channels retunes some of them before almost every call, and staging all channels for that cost 55 us and 0.4 ms of
  // host time per call at 32768 channels.  Beyond kMaxPatch channels per call the whole bank is staged as before.
  static constexpr int kMaxPatch = 1024;
  static constexpr size_t kPatchBytes = 72;  // one record: channel index (8 bytes), then the channel's eight plane values
  std::vector<int> patch_list;
  size_t patch_off = 0;                       // of the patch records inside a staging slot
  size_t bits_off = 0;                        // of the patched-channel bitmap (one bit per channel) behind them: the threads of
                                              // k_block_energy_sum that advance the planes skip the channels its patch role writes
  std::vector<unsigned> slot_bit_words[4];    // which words of a slot's bitmap are not zero (cleared when the slot comes round)
  int n_swept = 0, n_fast = 0;                // active channels with a sweep / with one beyond the table path's reach (N = 65536
                                              // and N = 16384: full64k_sweep_limit / full16k_sweep_limit)
  int n_active = 0;
  // N = 16384, some channels swept (satellite passes in a bank of fixed-frequency channels): the unswept ones still run the
  // steady-state variant of the kernel (16-byte loads from the row-paired copy, no per-sample oscillator path), the swept
  // ones the general variant, as two launches over two channel lists -- one swept channel used to cost the whole bank 6 %
  int *list_unswept_dev = nullptr, *list_swept_dev = nullptr;
  std::vector<int> list_unswept_host, list_swept_host;
  bool sweep_lists_dirty = true;
  int64_t n_abs = 0;        // absolute index of the first new (not yet processed) sample
  int64_t out_abs = 0;      // absolute index of the next output sample
  unsigned last_blocks = 0;

  int timing = 0;  // 0 off, 1 filter kernel only, >= 2 every scope
  kq_host_timing host_acc = {};  // the host's own time inside the process calls (always on: three clock reads per call)
  const char *worst_holder = "";  // the entry point behind host_acc.ctl_hold_max_ms
  std::vector<EventPair> ev_filter, ev_demod, ev_ingest;
  size_t ev_used[3] = {0, 0, 0};
  kq_timing acc = {};
};

namespace {

template <typename T>
int dev_alloc(T **p, size_t n) {
  HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
  HIP_TRY(hipMemset(*p, 0, n * sizeof(T)));
  return 0;
}

int ilog2(unsigned v) {
  int l = 0;
  while ((1u << l) < v) l++;
  return l;
}

int sync_all(kq_bank *b);

// every entry point taking a handle: the handle's device made current, the handle's lock held.
// kq_host_timing's lock figures are kept here: a `receiver` scope (the process calls) records how long it WAITED for the
// lock, every other scope how long it HELD it (device waits taken with the lock let go -- Unlocked -- not counted): the
// worst of the second is the longest the receiver thread can have been kept out by the control plane.
struct BankScope {
  using clock = std::chrono::steady_clock;
  kq::DeviceScope dev;
  std::unique_lock<std::recursive_mutex> lk;
  kq_bank *bank = nullptr;
  bool receiver = false;
  const char *who;  // the entry point (its function name, taken where the scope is declared)
  clock::time_point t_acq;
  double unlocked_ms = 0;
  explicit BankScope(kq_bank *b, bool receiver_ = false, const char *fn = __builtin_FUNCTION())
      : dev(b ? b->cfg.device : -1), bank(b), receiver(receiver_), who(fn) {
    if (!b) return;
    auto const t0 = clock::now();
    lk = std::unique_lock<std::recursive_mutex>(b->mu);
    t_acq = clock::now();
    if (receiver) {
      double const w = std::chrono::duration<double, std::milli>(t_acq - t0).count();
      b->host_acc.lock_wait_ms += w;
      if (w > b->host_acc.lock_wait_max_ms) b->host_acc.lock_wait_max_ms = w;
    }
  }
  explicit BankScope(const kq_bank *b, const char *fn = __builtin_FUNCTION()) : BankScope(const_cast<kq_bank *>(b), false, fn) {}
  ~BankScope() {
    if (!bank || receiver || !lk.owns_lock()) return;
    double const h = std::chrono::duration<double, std::milli>(clock::now() - t_acq).count() - unlocked_ms;
    if (h > bank->host_acc.ctl_hold_max_ms) {
      bank->host_acc.ctl_hold_max_ms = h;
      bank->worst_holder = who;
    }
  }
};
// a wait for the device inside an entry point: the lock is let go for its duration (one level: an entry point called from
// another keeps the outer one's)
struct Unlocked {
  BankScope &scope;
  std::unique_lock<std::recursive_mutex> &lk;
  BankScope::clock::time_point t0;
  explicit Unlocked(BankScope &s) : scope(s), lk(s.lk), t0(BankScope::clock::now()) {
    if (lk.owns_lock()) lk.unlock();
  }
  ~Unlocked() {
    if (lk.mutex() && !lk.owns_lock()) lk.lock();
    scope.unlocked_ms += std::chrono::duration<double, std::milli>(BankScope::clock::now() - t0).count();
  }
};

enum { CTL_FILTER = 0, CTL_DEMOD = 1 };
struct CtlRecHost {  // kq_kernels.hip CtlRec
  unsigned long long dst;
  unsigned nbytes, fill, value, payload_off;  // fill: 0 payload, 1 fill with `value`, 2 copy from device address `src`
  unsigned long long src;
};
static_assert(sizeof(CtlRecHost) == 32, "control record layout");

int ctl_flush(kq_bank *b, int side, hipStream_t st);
int ctl_flush_now(kq_bank *b);

// room for one more record with `bytes` of payload in queue `side`; a full queue is applied early, on the main stream
// behind the demodulators in flight (never in practice: a megabyte of parameters between two calls)
int ctl_room(kq_bank *b, int side, size_t bytes) {
  kq_bank::CtlQueue &q = b->ctl[side];
  size_t const payload_cap = kq_bank::CtlQueue::kBytes - kq_bank::CtlQueue::kMaxRec * sizeof(CtlRecHost);
  if (bytes > payload_cap) {
    set_err("control-plane write of %zu bytes exceeds the queue", bytes);
    return -1;
  }
  if (q.nrec >= kq_bank::CtlQueue::kMaxRec || q.used + bytes > payload_cap) return ctl_flush_now(b);
  return 0;
}
// queue a copy of `bytes` (a multiple of 4) from host memory to device memory / a 32-bit fill of device memory.
// A destination that is already in the queue: the earlier record is rewritten in place (same or larger extent) or
// cancelled and replaced (smaller extent) -- every field of the control plane has one destination and one extent, so a
// partial overlap with a different start does not occur.
int ctl_put(kq_bank *b, int side, void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return 0;
  // a long payload (a channel list of a large bank: 128 KiB) as records of 4 KiB: a record is one workgroup's work, and its
  // loads cross the link -- as ONE record a list took a workgroup ~0.25 ms, which the next call's kernels waited for
  size_t const kChunk = 4096;
  if (bytes > kChunk) {
    for (size_t off = 0; off < bytes; off += kChunk)
      if (ctl_put(b, side, static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, std::min(kChunk, bytes - off)))
        return -1;
    return 0;
  }
  kq_bank::CtlQueue &q = b->ctl[side];
  unsigned long long const key = (unsigned long long)(uintptr_t)dst;
  auto it = q.at.find(key);
  if (it != q.at.end()) {
    CtlRecHost *old = reinterpret_cast<CtlRecHost *>(q.buf[q.cur]) + it->second;
    if (!old->fill && old->nbytes >= bytes) {
      memcpy(q.buf[q.cur] + old->payload_off, src, bytes);
      return 0;
    }
    if (old->nbytes <= bytes) old->nbytes = 0;  // covered by the new record: cancelled
    // (an earlier, LARGER fill under a smaller copy -- no caller does that -- stays in the launch beside the copy: avoided
    //  by applying what has gathered first)
    else if (ctl_flush_now(b)) return -1;
  }
  if (ctl_room(b, side, bytes)) return -1;
  size_t const off = kq_bank::CtlQueue::kMaxRec * sizeof(CtlRecHost) + q.used;
  memcpy(q.buf[q.cur] + off, src, bytes);
  CtlRecHost const r{key, (unsigned)bytes, 0u, 0u, (unsigned)off, 0ull};
  memcpy(q.buf[q.cur] + (size_t)q.nrec * sizeof r, &r, sizeof r);
  q.at[key] = q.nrec;
  q.nrec++;
  q.used += (bytes + 15) & ~(size_t)15;
  return 0;
}
int ctl_fill(kq_bank *b, int side, void *dst, unsigned value, size_t bytes) {
  if (bytes == 0) return 0;
  kq_bank::CtlQueue &q = b->ctl[side];
  unsigned long long const key = (unsigned long long)(uintptr_t)dst;
  auto it = q.at.find(key);
  if (it != q.at.end()) {
    CtlRecHost *old = reinterpret_cast<CtlRecHost *>(q.buf[q.cur]) + it->second;
    if (old->nbytes <= bytes)
      old->nbytes = 0;  // covered: cancelled
    else if (ctl_flush_now(b))
      return -1;
  }
  if (ctl_room(b, side, 0)) return -1;
  CtlRecHost const r{key, (unsigned)bytes, 1u, value, 0u, 0ull};
  memcpy(q.buf[q.cur] + (size_t)q.nrec * sizeof r, &r, sizeof r);
  q.at[key] = q.nrec;
  q.nrec++;
  return 0;
}
// queue a copy of `bytes` from device memory `src`, read when the queue is applied
int ctl_copy_dev(kq_bank *b, int side, void *dst, const void *src, size_t bytes) {
  kq_bank::CtlQueue &q = b->ctl[side];
  unsigned long long const key = (unsigned long long)(uintptr_t)dst;
  auto it = q.at.find(key);
  if (it != q.at.end()) {
    CtlRecHost *old = reinterpret_cast<CtlRecHost *>(q.buf[q.cur]) + it->second;
    if (old->nbytes <= bytes)
      old->nbytes = 0;
    else if (ctl_flush_now(b))
      return -1;
  }
  if (ctl_room(b, side, 0)) return -1;
  CtlRecHost const r{key, (unsigned)bytes, 2u, 0u, 0u, (unsigned long long)(uintptr_t)src};
  memcpy(q.buf[q.cur] + (size_t)q.nrec * sizeof r, &r, sizeof r);
  q.at[key] = q.nrec;
  q.nrec++;
  return 0;
}
// withdraw a queued write to `dst` (something else is going to write there in front of the same call)
void ctl_cancel(kq_bank *b, int side, void *dst) {
  kq_bank::CtlQueue &q = b->ctl[side];
  auto it = q.at.find((unsigned long long)(uintptr_t)dst);
  if (it == q.at.end()) return;
  (reinterpret_cast<CtlRecHost *>(q.buf[q.cur]) + it->second)->nbytes = 0;
  q.at.erase(it);
}
// withdraw every queued write that starts inside [dst, dst + bytes): a bulk rewrite of that range follows, and the records
// of one launch are applied concurrently (a 4-byte list entry queued earlier must not land beside the chunk covering it)
void ctl_cancel_range(kq_bank *b, int side, const void *dst, size_t bytes) {
  kq_bank::CtlQueue &q = b->ctl[side];
  unsigned long long const lo = (unsigned long long)(uintptr_t)dst, hi = lo + bytes;
  for (auto it = q.at.begin(); it != q.at.end();) {
    if (it->first >= lo && it->first < hi) {
      (reinterpret_cast<CtlRecHost *>(q.buf[q.cur]) + it->second)->nbytes = 0;
      it = q.at.erase(it);
    } else
      ++it;
  }
}

// the design jobs gathered since the last call: one launch on the main stream
// `ctl_queue` != null: that many write records of the filter side's queue ride in the same launch (*took_records set)
int design_flush(kq_bank *b, const void *ctl_queue = nullptr, unsigned ctl_records = 0, bool *took_records = nullptr) {
  kq_bank::DesignQueue &d = b->dq;
  if (took_records) *took_records = false;
  if (d.jobs.empty()) return 0;
  int const p = (int)(d.epoch & 1);
  unsigned const n = (unsigned)d.jobs.size();
  if (d.read_set[d.cur]) HIP_TRY(hipEventSynchronize(d.read[d.cur]));  // (kDepth launches ago)
  unsigned char *pin = d.pin[d.cur];
  memcpy(pin, d.jobs.data(), n * sizeof(kq::DesignJob));
  memcpy(pin + kq_bank::DesignQueue::kMax * sizeof(kq::DesignJob), d.targets.data(), n * sizeof(kq::DesignTarget));
  // ng_next[p] was last written two design launches ago; the records that moved those values on ran on the demodulators' stream
  if (d.ng_moved_set[p]) HIP_TRY(hipStreamWaitEvent(b->stream, d.ng_moved[p], 0));
  if (kq::design_launch(b->stream, b->g.olen, b->g.Mdec, reinterpret_cast<const kq::DesignJob *>(pin),
                        reinterpret_cast<const kq::DesignTarget *>(pin + kq_bank::DesignQueue::kMax * sizeof(kq::DesignJob)), n,
                        d.scratch, ctl_queue, ctl_records)) {
    set_err("response design launch failed");
    return -1;
  }
  if (took_records) *took_records = ctl_queue != nullptr && ctl_records > 0;
  HIP_TRY(hipEventRecord(d.read[d.cur], b->stream));
  d.read_set[d.cur] = true;
  d.cur = (d.cur + 1) % kq_bank::DesignQueue::kDepth;
  d.ng_to_record = p;
  d.epoch++;
  d.jobs.clear();
  d.targets.clear();
  d.at.clear();
  return 0;
}

// apply what has gathered in queue `side` with one launch on `st`
int ctl_flush(kq_bank *b, int side, hipStream_t st) {
  kq_bank::CtlQueue &q = b->ctl[side];
  // (the filter side is always applied on the main stream: with design jobs waiting, its records ride in their launch --
  //  nothing a record writes is read or written by a design job, kq_bank::DesignQueue / ctl_cancel see to that)
  bool applied = false;
  if (side == CTL_FILTER && design_flush(b, q.nrec ? q.buf[q.cur] : nullptr, q.nrec, &applied)) return -1;
  if (q.nrec == 0) return 0;
  if (!applied) kq::launch_ctl_apply(st, q.buf[q.cur], (int)q.nrec);
  HIP_TRY(hipEventRecord(q.applied[q.cur], st));
  q.applied_set[q.cur] = true;
  if (side == CTL_DEMOD && b->dq.ng_to_record >= 0) {
    int const p = b->dq.ng_to_record;
    HIP_TRY(hipEventRecord(b->dq.ng_moved[p], st));
    b->dq.ng_moved_set[p] = true;
    b->dq.ng_to_record = -1;
  }
  q.cur = (q.cur + 1) % kq_bank::CtlQueue::kDepth;
  q.nrec = 0;
  q.used = 0;
  q.at.clear();
  // the buffer gathered into next was handed to the device kDepth flushes ago: long applied
  if (q.applied_set[q.cur]) HIP_TRY(hipEventSynchronize(q.applied[q.cur]));
  return 0;
}
// both queues applied now, on the main stream behind the demodulators in flight (for the rare paths that go on to touch
// the device synchronously: carrier-loop slots, batched channel set-up)
int ctl_flush_now(kq_bank *b) {
  if (b->ctl[0].nrec == 0 && b->ctl[1].nrec == 0 && b->dq.jobs.empty()) return 0;
  if (b->calls > 0) {
    int const last = (int)((b->calls - 1) & 1);
    if (b->demod_overlapped[last]) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_demod_done[last], 0));
  }
  if (ctl_flush(b, CTL_FILTER, b->stream) || ctl_flush(b, CTL_DEMOD, b->stream)) return -1;
  return 0;
}

// compute_n0's passband exclusion (radio.c:405-411) depends only on the channel's filter edges: one bit per bin, as
// 64-bit lane masks in the order k_filter_full16k holds the bins (kq_device.hpp ChanDev::n0lane).  Same arithmetic as
// the reference, int wrap of k * samprate included (radio.c:407,409).  N = 65536: sub-transform r holds bins 4 q + r.
void build_n0mask(const kq_bank *b, float low, float high, std::vector<unsigned long long> &m, std::vector<unsigned> &meta) {
  kq::Geom const &g = b->g;
  int const nsub = b->use64k ? 4 : 1;
  m.assign((size_t)nsub * 256, 0ull);
  meta.assign(nsub, 0u);
  for (int r = 0; r < nsub; r++)
    for (int t = 0; t < 512; t++) {
      int const ka = kq::full16k_bin(t);
      for (int half = 0; half < 2; half++)
        for (int k3 = 0; k3 < 16; k3++) {
          int const n = nsub * (ka + kq::kFull16kHalf * half + 1024 * k3) + r;
          int const k = (n <= g.N / 2) ? n : n - g.N;
          int const prod = (int)((unsigned)k * (unsigned)g.samprate);
          float const f = (float)prod / g.N;
          if (!(f >= low && f <= high)) {
            m[((size_t)r * 8 + (t >> 6)) * 32 + 16 * half + k3] |= 1ull << (t & 63);
            meta[r]++;
          }
        }
    }
}

// the mask set for these edges: an existing one, or a free slot filled now (*fresh).  Takes a reference.  -1: no slot
// left (cannot happen while every reference belongs to a channel and a channel gives its old slot back BEFORE it asks
// for a new one -- upload_n0mask; the planes hold exactly max_channels sets, so a slot past them is refused, never
// written)
int acquire_n0slot(kq_bank *b, float low, float high, bool *fresh) {
  auto const key = std::make_pair(low, high);
  auto it = b->n0slot_of.find(key);
  *fresh = it == b->n0slot_of.end();
  int slot;
  if (!*fresh) {
    slot = it->second;
  } else {
    slot = -1;
    for (size_t k = 0; k < b->n0slot_refs.size(); k++)
      if (b->n0slot_refs[k] == 0) {
        slot = (int)k;
        break;
      }
    if (slot < 0) {
      if (b->n0slot_refs.size() >= (size_t)b->cfg.max_channels) {
        set_err("compute_n0 mask slots exhausted (%zu sets for %u channels)", b->n0slot_refs.size(), b->cfg.max_channels);
        return -1;
      }
      slot = (int)b->n0slot_refs.size();
      b->n0slot_refs.push_back(0);
      b->n0slot_key.push_back(key);
    }
    b->n0slot_key[slot] = key;
    b->n0slot_of[key] = slot;
  }
  b->n0slot_refs[slot]++;
  return slot;
}
void release_n0slot(kq_bank *b, int slot) {
  if (slot < 0 || (size_t)slot >= b->n0slot_refs.size() || b->n0slot_refs[slot] <= 0) return;
  if (--b->n0slot_refs[slot] == 0) b->n0slot_of.erase(b->n0slot_key[slot]);
}

int upload_n0mask(kq_bank *b, int c) {
  if (!b->chd.n0lane) return 0;
  int const nsub = b->use64k ? 4 : 1;
  bool fresh = false;
  int const old = b->chans[c].n0slot;
  auto const key = std::make_pair(b->chans[c].cfg.low, b->chans[c].cfg.high);
  if (old >= 0 && (size_t)old < b->n0slot_refs.size() && b->n0slot_refs[old] > 0 && b->n0slot_key[old] == key) {
    // unchanged edges keep their slot (and its reference)
    return ctl_put(b, CTL_FILTER, b->chd.n0slot + c, &old, sizeof(int)) ? -1 : 0;
  }
  // the old set goes back FIRST: in a full bank of distinct edges it is the only free one (asking first ran
  // one set past the planes).  Reusing it in place is safe: the mask write below is queued behind the filter passes in
  // flight, and no other channel refers to a slot whose count reached zero
  release_n0slot(b, old);
  b->chans[c].n0slot = -1;
  int const slot = acquire_n0slot(b, key.first, key.second, &fresh);
  if (slot < 0) return -1;
  b->chans[c].n0slot = slot;
  std::vector<unsigned long long> m;
  std::vector<unsigned> meta;
  if (fresh) {
    build_n0mask(b, b->chans[c].cfg.low, b->chans[c].cfg.high, m, meta);
    if (ctl_put(b, CTL_FILTER, b->chd.n0lane + (size_t)slot * nsub * 256, m.data(), m.size() * sizeof(m[0]))) return -1;
    if (ctl_put(b, CTL_FILTER, b->chd.n0meta + (size_t)slot * nsub, meta.data(), meta.size() * sizeof(unsigned))) return -1;
  }
  if (ctl_put(b, CTL_FILTER, b->chd.n0slot + c, &slot, sizeof(int))) return -1;
  return 0;
}

// The constants each demodulator thread derives in its prologue (fm.c:86; am.c:21-30; linear.c:29-39)
struct Derived {
  int mode, flags, hangmax;
  float fm_gain, recovery, init_gain;
};
Derived derive(const kq::Geom &g, const kq_channel_config &k) {
  Derived d;
  d.mode = k.demod_type;
  d.flags = 0;
  if (k.flat) d.flags |= kq::FLAG_FLAT;
  if (k.isb && d.mode == KQ_LINEAR_DEMOD) d.flags |= kq::FLAG_ISB;
  if (k.channels == 2 && d.mode == KQ_LINEAR_DEMOD) d.flags |= kq::FLAG_STEREO;
  if (k.square && d.mode == KQ_LINEAR_DEMOD) d.flags |= kq::FLAG_SQUARE;
  float const samptime = (float)g.D / (float)g.samprate;  // am.c:21, linear.c:29
  float const rec_db = k.recovery_rate * samptime;
  d.recovery = powf(10.f, (float)((double)rec_db / 20.));  // dB2voltage, dsp.h:38
  d.hangmax = (int)(k.hangtime / samptime);                  // am.c:29, linear.c:38
  d.fm_gain = (float)((k.headroom * M_1_PI * g.dsamprate) / fabsf(k.low - k.high));  // fm.c:86
  d.init_gain = (d.mode == KQ_AM_DEMOD) ? powf(10.f, (float)(80. / 20.)) : powf(10.f, (float)(100.0 / 20.));
  return d;
}

// Derived per-channel constants, as each demod thread computes them in its prologue
// fresh = false: a new demodulator thread on an existing channel (set_mode): what struct demod keeps (sig.n0,
// sig.foffset, sig.pdeviation) is left alone
int upload_channel(kq_bank *b, int c, bool fresh = true) {
  // nothing here touches the device or waits: the writes gather in the control queues and the next call applies them --
  // what the filter kernels read in front of its filter pass, what the demodulators read and carry in front of its
  // demodulators, each behind the calls in flight
  HostChan &h = b->chans[c];
  kq::Geom const &g = b->g;
  kq_channel_config const &k = h.cfg;
  Derived const dv = derive(g, k);
  int const mode = dv.mode, flags = dv.flags, hangmax = dv.hangmax;
  float const recovery = dv.recovery, fm_gain = dv.fm_gain, init_gain = dv.init_gain;
  float const nan = NAN;
  float2 const one = make_float2(1.f, 0.f);  // fm.c:26
  auto const D = [&](void *dst, const void *src, size_t n) { return ctl_put(b, CTL_DEMOD, dst, src, n); };
  auto const Z = [&](void *dst, size_t n) { return ctl_fill(b, CTL_DEMOD, dst, 0u, n); };

  if (ctl_put(b, CTL_FILTER, b->chd.fflags + c, &flags, sizeof(int))) return -1;
  if (ctl_put(b, CTL_FILTER, b->chd.low + c, &k.low, sizeof(float))) return -1;
  if (ctl_put(b, CTL_FILTER, b->chd.high + c, &k.high, sizeof(float))) return -1;
  if (D(b->chd.mode + c, &mode, sizeof(int)) || D(b->chd.flags + c, &flags, sizeof(int)) ||
      D(b->chd.fm_gain + c, &fm_gain, sizeof(float)) || D(b->chd.headroom + c, &k.headroom, sizeof(float)) ||
      D(b->chd.recovery + c, &recovery, sizeof(float)) || D(b->chd.hangmax + c, &hangmax, sizeof(int)) ||
      D(b->chd.gain + c, &init_gain, sizeof(float)) || (fresh && D(b->chd.n0 + c, &nan, sizeof(float))) ||
      D(b->chd.fm_state + c, &one, sizeof(float2)))
    return -1;
  // thread-local state of the demodulators at their prologue values (fm.c:26,68-69; am.c:26,33; linear.c:33)
  if (Z(b->chd.lastaudio + c, sizeof(float)) || Z(b->chd.sq_count + c, sizeof(int)) || Z(b->chd.hang + c, sizeof(int)) ||
      Z(b->chd.dc + c, sizeof(float)))
    return -1;
  if (g.Mdec > 1) {
    if (Z(b->chd.ahist + (size_t)c * (g.Mdec - 1), sizeof(float) * (g.Mdec - 1))) return -1;
    for (int kk = 0; kk < 2; kk++)
      if (b->fm_hist[kk] && Z(b->fm_hist[kk] + (size_t)c * (g.Mdec - 1), sizeof(float) * (g.Mdec - 1))) return -1;
  }
  if (fresh && (Z(b->chd.foffset + c, sizeof(float)) || Z(b->chd.pdev + c, sizeof(float)))) return -1;
  if (g.pl_n > 0) {
    if (Z(b->chd.plring + (size_t)c * 16384, sizeof(float) * 16384) || Z(b->chd.pl_ptr + c, sizeof(*b->chd.pl_ptr)) ||
        Z(b->chd.pl_last + c, sizeof(*b->chd.pl_last)))
      return -1;
  }
  if (D(b->chd.plfreq + c, &nan, sizeof(float))) return -1;
  if (upload_n0mask(b, c)) return -1;
  return 0;
}

// Pre-detection response: set_filter with edges normalised to the output rate
// (fm.c:35: low/dsamprate; am.c:41, linear.c:81: samptime*low)
// `runtime`: a change made while running goes through display.c:161-177, which scales by samptime whatever the mode
// The design is queued for the next call's design launch (kq_bank::DesignQueue); the FM audio response (designed once, in
// the demodulator's prologue, fm.c:54-66) comes from the bank's cache by Kaiser beta -- designed, and waited for, the
// first time a beta is seen.
int queue_design(kq_bank *b, int c, bool runtime = false) {
  HostChan &h = b->chans[c];
  kq::Geom const &g = b->g;
  kq_bank::DesignQueue &d = b->dq;
  if (g.Ndec > 16384) {
    set_err("response design: N / decimate = %d exceeds 16384", g.Ndec);
    return -1;
  }
  float lo_n, hi_n;
  if (h.cfg.demod_type == KQ_FM_DEMOD && !runtime) {
    lo_n = h.cfg.low / g.dsamprate;
    hi_n = h.cfg.high / g.dsamprate;
  } else {
    float const samptime = (float)g.D / (float)g.samprate;
    lo_n = samptime * h.cfg.low;
    hi_n = samptime * h.cfg.high;
  }
  if (!runtime) {
    if (h.cfg.demod_type == KQ_FM_DEMOD && !h.cfg.flat) {
      if (std::isnan(h.cfg.kaiser_beta)) {  // (not a key an ordered map can hold; the design takes it as the reference's does)
        h.aresp = kq::design_fm_audio_response(g.olen, g.Mdec, g.dsamprate, h.cfg.kaiser_beta);
        if (h.aresp.empty()) return -1;
      } else {
        auto it = b->aresp_cache.find(h.cfg.kaiser_beta);
        if (it == b->aresp_cache.end()) {
          std::vector<kq::cfloat> a = kq::design_fm_audio_response(g.olen, g.Mdec, g.dsamprate, h.cfg.kaiser_beta);
          if (a.empty()) return -1;
          if (b->aresp_cache.size() >= 64) b->aresp_cache.clear();
          it = b->aresp_cache.emplace(h.cfg.kaiser_beta, std::move(a)).first;
        }
        h.aresp = it->second;
      }
      if (ctl_put(b, CTL_DEMOD, b->chd.aresp + (size_t)c * (g.Ndec / 2 + 1), h.aresp.data(), sizeof(float2) * (g.Ndec / 2 + 1)))
        return -1;
    } else {
      h.aresp.clear();
    }
  }
  auto at = d.at.find(c);
  if (at == d.at.end() && d.jobs.size() >= d.max_jobs && ctl_flush_now(b)) return -1;  // (applies what has gathered, early)
  int const p = (int)(d.epoch & 1);
  float gain, ng_scale;
  kq::design_scales(g.N, h.out_type, &gain, &ng_scale);
  float2 *const row = b->chd.resp + (size_t)c * g.Ndec;
  float *const ng = d.ng_next + (size_t)p * b->cfg.max_channels + c;
  kq::DesignJob const job{lo_n, hi_n, h.cfg.kaiser_beta, gain};
  kq::DesignTarget const target{row, ng, ng_scale, 0.f};
  at = d.at.find(c);
  if (at != d.at.end()) {
    d.jobs[at->second] = job;
    d.targets[at->second] = target;
  } else {
    d.at[c] = (unsigned)d.jobs.size();
    d.jobs.push_back(job);
    d.targets.push_back(target);
  }
  ctl_cancel(b, CTL_FILTER, row);  // a response queued from the host for this row: the design runs in front of the queue
  h.resp.clear();                  // the host's copy: fetched when asked for
  return ctl_copy_dev(b, CTL_DEMOD, b->chd.noise_gain + c, ng, sizeof(float));
}
// the host's copy of a channel's response
int fetch_response(kq_bank *b, int c) {
  HostChan &h = b->chans[c];
  if (!h.resp.empty()) return 0;
  if (ctl_flush_now(b) || sync_all(b)) return -1;
  h.resp.resize(b->g.Ndec);
  HIP_TRY(hipMemcpy((void *)h.resp.data(), b->chd.resp + (size_t)c * b->g.Ndec, sizeof(float2) * b->g.Ndec, hipMemcpyDeviceToHost));
  return 0;
}

int ensure_events(std::vector<EventPair> &v, size_t need) {
  while (v.size() < need) {
    EventPair p;
    HIP_TRY(hipEventCreate(&p.a));
    HIP_TRY(hipEventCreate(&p.b));
    v.push_back(p);
  }
  return 0;
}

// the N = 65536 kernel's "a sibling never showed up" flag (pinned host memory): reported once, by whichever of
// kq_bank_sync / kq_bank_host_io_wait the host uses to wait -- a streaming host never calls the former
int report_lost_sibling(kq_bank *b) {
  if (b->big.err && *b->big.err) {
    *b->big.err = 0;
    set_err("N = 65536 filter: a sibling workgroup's compute_n0 sum never arrived (n0 of that call is NaN)");
    return -1;
  }
  return 0;
}

int sync_all(kq_bank *b) {
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream2));
  // the streaming copies too: "everything issued so far" (kq_bank_sync) includes a plane copy still in flight
  if (b->copy_in) HIP_TRY(hipStreamSynchronize(b->copy_in));
  if (b->copy_out) HIP_TRY(hipStreamSynchronize(b->copy_out));
  return 0;
}

// the filter interval of the call that last used the slot (its closing marker has completed)
int harvest_slot(kq_bank *b, int slot) {
  if (!b->stage_timed[slot]) return 0;
  b->stage_timed[slot] = false;
  float ms = 0;
  // (a call that failed between the two records leaves an interval that does not exist: dropped, not an error of this call)
  if (hipEventElapsedTime(&ms, b->stage_t0[slot], b->stage_ev[slot]) == hipSuccess && ms > 0) {
    b->acc.filter_ms += ms;
    if (ms > b->acc.filter_max_ms) {
      b->acc.filter_max_ms = ms;
      b->acc.filter_max_submit_ms = b->stage_submit_ms[slot];
      b->acc.filter_max_launch = b->stage_launch[slot];
    }
  } else
    (void)hipGetLastError();
  return 0;
}

int drain_timing(kq_bank *b) {
  if (sync_all(b)) return -1;
  for (int k = 0; k < kq_bank::kSlots; k++)
    if (harvest_slot(b, k)) return -1;
  std::vector<EventPair> *sets[3] = {&b->ev_filter, &b->ev_demod, &b->ev_ingest};
  double *dst[3] = {&b->acc.filter_ms, &b->acc.demod_ms, &b->acc.ingest_ms};
  for (int k = 0; k < 3; k++) {
    for (size_t i = 0; i < b->ev_used[k]; i++) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, (*sets[k])[i].a, (*sets[k])[i].b));
      *dst[k] += ms;
    }
    b->ev_used[k] = 0;
  }
  return 0;
}

struct Scope {
  kq_bank *b;
  int kind;
  EventPair *p = nullptr;
  hipStream_t st;
  Scope(kq_bank *bank, int k, hipStream_t stream) : b(bank), kind(k), st(stream) {
    if (!b->timing || (kind != 0 && b->timing < 2)) return;
    std::vector<EventPair> &v = kind == 0 ? b->ev_filter : kind == 1 ? b->ev_demod : b->ev_ingest;
    if (b->ev_used[kind] >= 512) drain_timing(b);
    if (ensure_events(v, b->ev_used[kind] + 1)) return;
    p = &v[b->ev_used[kind]++];
    (void)hipEventRecord(p->a, st);
  }
  ~Scope() {
    if (p) (void)hipEventRecord(p->b, st);
  }
};

// the eight plane values of one channel for a call whose first window starts at absolute sample n_w:
// out = {phase, step, sweep, shift phase, shift step, history phase, history step, history sweep}
void eval_planes(const kq_bank *b, const HostChan &h, int64_t n_w, double out[8]) {
  double p = h.lo2.phase_at(n_w), f = h.lo2.step_at(n_w), r = h.lo2.sweep();
  if (h.dop.set_f != 0) {  // radio.c:135: the Doppler NCO is applied only while its frequency is non-zero
    p += h.dop.phase_at(n_w);
    f += h.dop.step_at(n_w);
    r += h.dop.sweep();
  }
  out[0] = p - std::floor(p);
  out[1] = f;
  out[2] = r;
  double const q = h.shift.phase_at(b->out_abs);
  out[3] = q - std::floor(q);
  out[4] = h.shift.step_at(b->out_abs);
  if (h.retuned) {  // history of the first block keeps the pre-retune oscillators
    double q2 = h.lo2_old.phase_at(n_w), g2 = h.lo2_old.step_at(n_w), r2 = h.lo2_old.sweep();
    if (h.dop_old.set_f != 0) {
      q2 += h.dop_old.phase_at(n_w);
      g2 += h.dop_old.step_at(n_w);
      r2 += h.dop_old.sweep();
    }
    out[5] = q2 - std::floor(q2);
    out[6] = g2;
    out[7] = r2;
  } else {
    out[5] = out[0];
    out[6] = f;
    out[7] = r;
  }
}

// an older transition inside the history (note_retune): its oscillators at the window start, as the history planes
void eval_older(const HostChan &h, int l, int64_t n_w, double out[3]) {
  double q = h.lo2_oldx[l].phase_at(n_w), f = h.lo2_oldx[l].step_at(n_w), r = h.lo2_oldx[l].sweep();
  if (h.dop_oldx[l].set_f != 0) {
    q += h.dop_oldx[l].phase_at(n_w);
    f += h.dop_oldx[l].step_at(n_w);
    r += h.dop_oldx[l].sweep();
  }
  out[0] = q - std::floor(q);
  out[1] = f;
  out[2] = r;
}

// Per-call parameters: oscillator phase/step/sweep for a call whose first window starts at absolute
// sample n_w, the shift oscillator at the first output sample, and the IF-power update flags.
// Filled into the next pinned staging slot (returned in *slot_out); the first kernel of the call copies it to the device.
// steady: the planes are advanced on the device from the call before; only the flags, the retune list and the patch
// records of the channels on b->patch_list travel (*npatch_out of them).
int stage_call_params(kq_bank *b, int64_t n_w, const unsigned char *update, unsigned nblocks, int *slot_out, int *nret_out,
                      bool steady, int *npatch_out) {
  size_t const C = b->chans.size(), Cmax = b->cfg.max_channels;
  int const slot = b->stage_next;
  b->stage_next = (slot + 1) % kq_bank::kSlots;
  auto const tw0 = std::chrono::steady_clock::now();
  HIP_TRY(hipEventSynchronize(b->stage_ev[slot]));  // the call that last read this slot has got past its filter
  auto const tw1 = std::chrono::steady_clock::now();
  b->host_acc.slot_wait_ms += std::chrono::duration<double, std::milli>(tw1 - tw0).count();
  if (harvest_slot(b, slot)) return -1;
  double *pl = reinterpret_cast<double *>(b->stage_host[slot]);
  unsigned char *flags = b->stage_host[slot] + 8 * Cmax * sizeof(double);
  memcpy(flags, update, nblocks);
  // the per-block flags sit right behind the eight oscillator planes, in the staging slot and on the device; behind
  // them the channels retuned since the last call (their first block is redone on the per-sample path)
  int *ret = reinterpret_cast<int *>(flags + ((b->cfg.max_blocks + 7) & ~7u));
  int nret = 0, npatch = 0;
  b->ret_host.clear();
  {  // the bitmap of the slot's last use goes back to zero
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(b->stage_host[slot] + b->bits_off);
    for (unsigned w : b->slot_bit_words[slot]) bits[w] = 0;
    b->slot_bit_words[slot].clear();
  }
  if (steady) {
    unsigned char *rec = b->stage_host[slot] + b->patch_off;
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(b->stage_host[slot] + b->bits_off);
    for (int c : b->patch_list) {
      if ((size_t)c >= b->chans.size()) continue;  // (removed since, and dropped from the end)
      HostChan const &h = b->chans[c];
      if (!h.active) continue;
      long long const idx = c;
      memcpy(rec, &idx, sizeof idx);
      double v[8];
      eval_planes(b, h, n_w, v);
      memcpy(rec + 8, v, sizeof v);
      rec += kq_bank::kPatchBytes;
      npatch++;
      bits[c >> 6] |= 1ull << (c & 63);
      b->slot_bit_words[slot].push_back((unsigned)(c >> 6));
      if (h.retuned) {
        ret[nret++] = c;
        b->ret_host.push_back(c);
      }
    }
  } else {
    for (size_t c = 0; c < C; c++) {
      double v[8];
      eval_planes(b, b->chans[c], n_w, v);
      for (int k = 0; k < 8; k++) pl[(size_t)k * Cmax + c] = v[k];
    }
    for (size_t c = 0; c < C; c++)
      if (b->chans[c].active && b->chans[c].retuned) {
        ret[nret++] = (int)c;
        b->ret_host.push_back((int)c);
      }
  }
  *nret_out = nret;
  *npatch_out = npatch;
  *slot_out = slot;
  b->host_acc.stage_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count();
  return 0;
}

int list_kind(const HostChan &h) {
  int const m = h.cfg.demod_type;
  if (m == KQ_LINEAR_DEMOD && h.cfg.pll) return 3;
  return m == KQ_FM_DEMOD ? 0 : m == KQ_AM_DEMOD ? 1 : 2;
}
// the lists from scratch (set-up, batched adds, anything that touches the carrier-loop list): ascending channel order
int upload_lists(kq_bank *b) {
  for (int k = 0; k < 3; k++) b->list_host[k].clear();
  b->list_pll_host.clear();
  b->list_active_host.clear();
  for (size_t c = 0; c < b->chans.size(); c++) {
    HostChan &h = b->chans[c];
    h.lk = h.lpos = h.apos = -1;
    if (!h.active) continue;
    h.apos = (int)b->list_active_host.size();
    b->list_active_host.push_back((int)c);
    h.lk = list_kind(h);
    std::vector<int> &l = h.lk == 3 ? b->list_pll_host : b->list_host[h.lk];  // (carrier loops: slot = position = order of creation)
    h.lpos = (int)l.size();
    l.push_back((int)c);
  }
  // the filter launch's list on the filter side; the demodulators' lists, and the PCM stage's copy of the active list,
  // on the demodulator side (the last call's demodulators may still be walking the old ones)
  size_t const n = b->list_active_host.size() * sizeof(int);
  {  // single entries queued while the lists were current (lists_add / lists_remove) give way to the rebuild
    size_t const whole = (size_t)b->cfg.max_channels * sizeof(int);
    ctl_cancel_range(b, CTL_FILTER, b->list_active_dev, whole);
    ctl_cancel_range(b, CTL_DEMOD, b->list_active_ds_dev, whole);
    if (b->list_pll_dev) ctl_cancel_range(b, CTL_DEMOD, b->list_pll_dev, whole);
    for (int k = 0; k < 3; k++) ctl_cancel_range(b, CTL_DEMOD, b->list_dev[k], whole);
  }
  if (ctl_put(b, CTL_FILTER, b->list_active_dev, b->list_active_host.data(), n)) return -1;
  if (ctl_put(b, CTL_DEMOD, b->list_active_ds_dev, b->list_active_host.data(), n)) return -1;
  if (!b->list_pll_host.empty())
    if (ctl_put(b, CTL_DEMOD, b->list_pll_dev, b->list_pll_host.data(), b->list_pll_host.size() * sizeof(int))) return -1;
  for (int k = 0; k < 3; k++)
    if (!b->list_host[k].empty())
      if (ctl_put(b, CTL_DEMOD, b->list_dev[k], b->list_host[k].data(), b->list_host[k].size() * sizeof(int))) return -1;
  b->lists_dirty = false;
  return 0;
}
// one channel onto / off the lists as they stand (nothing to do while a rebuild is pending; a carrier-loop channel asks for one)
int lists_add(kq_bank *b, int c) {
  if (b->lists_dirty) return 0;
  HostChan &h = b->chans[c];
  int const k = list_kind(h);
  if (k == 3) {
    b->lists_dirty = true;
    return 0;
  }
  h.apos = (int)b->list_active_host.size();
  b->list_active_host.push_back(c);
  if (ctl_put(b, CTL_FILTER, b->list_active_dev + h.apos, &c, sizeof(int))) return -1;
  if (ctl_put(b, CTL_DEMOD, b->list_active_ds_dev + h.apos, &c, sizeof(int))) return -1;
  h.lk = k;
  h.lpos = (int)b->list_host[k].size();
  b->list_host[k].push_back(c);
  return ctl_put(b, CTL_DEMOD, b->list_dev[k] + h.lpos, &c, sizeof(int));
}
int lists_remove(kq_bank *b, int c, bool from_active);
// after a mode change: onto the new mode's list if that is another one
int lists_retype(kq_bank *b, int c) {
  if (b->lists_dirty) return 0;
  HostChan &h = b->chans[c];
  int const k = list_kind(h);
  if (k == 3 || h.lk == 3 || h.lk < 0) {
    b->lists_dirty = true;
    return 0;
  }
  if (k == h.lk) return 0;
  if (lists_remove(b, c, false)) return -1;
  h.lk = k;
  h.lpos = (int)b->list_host[k].size();
  b->list_host[k].push_back(c);
  return ctl_put(b, CTL_DEMOD, b->list_dev[k] + h.lpos, &c, sizeof(int));
}
int lists_remove(kq_bank *b, int c, bool from_active = true) {
  if (b->lists_dirty) return 0;
  HostChan &h = b->chans[c];
  if (h.lk == 3 || h.lk < 0 || h.lpos < 0 || h.apos < 0) {
    b->lists_dirty = true;
    return 0;
  }
  {
    std::vector<int> &l = b->list_host[h.lk];
    int const last = l.back();
    l[h.lpos] = last;
    b->chans[last].lpos = h.lpos;
    l.pop_back();
    if (last != c && ctl_put(b, CTL_DEMOD, b->list_dev[h.lk] + h.lpos, &last, sizeof(int))) return -1;
  }
  if (from_active) {
    std::vector<int> &l = b->list_active_host;
    int const last = l.back();
    l[h.apos] = last;
    b->chans[last].apos = h.apos;
    l.pop_back();
    if (last != c) {
      if (ctl_put(b, CTL_FILTER, b->list_active_dev + h.apos, &last, sizeof(int))) return -1;
      if (ctl_put(b, CTL_DEMOD, b->list_active_ds_dev + h.apos, &last, sizeof(int))) return -1;
    }
    h.apos = -1;
  }
  h.lk = h.lpos = -1;
  return 0;
}

// The kernels of one call over `nblocks` blocks whose first window starts at `window`
// `spectrum` != null: the master's transform has been done elsewhere (execute_filter_input of the compat surface) and
// `spectrum` holds its N bins per block -- slave, compute_n0 and demodulators only (kq_bank_process_spectrum)
int run_blocks_timed(kq_bank *b, const float2 *window, unsigned nblocks, const unsigned char *update_host,
                     const float2 *spectrum);
void note_patch(kq_bank *b, int ch);
void note_retune(kq_bank *b, int ch);

// kq_bank_get_host_timing: the host's wall time inside one call, kernels only queued (call_ms includes slot_wait_ms)
int run_blocks(kq_bank *b, const float2 *window, unsigned nblocks, const unsigned char *update_host,
               const float2 *spectrum = nullptr) {
  auto const t0 = std::chrono::steady_clock::now();
  int const rc = run_blocks_timed(b, window, nblocks, update_host, spectrum);
  // A call that failed may have left the lists' bulk records queued and unapplied; a 4-byte record of lists_add /
  // lists_remove beside them in one launch would race with the chunk that covers it.  With the lists marked
  // stale those entry points queue nothing, and the next call's rebuild rewrites the queued chunks in place.
  if (rc < 0) b->lists_dirty = true;
  b->host_acc.call_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  b->host_acc.calls++;
  return rc;
}

int run_blocks_timed(kq_bank *b, const float2 *window, unsigned nblocks, const unsigned char *update_host,
                     const float2 *spectrum) {
  kq::Geom const &g = b->g;
  int const C = (int)b->chans.size();
  // Per-channel decisions are taken afresh only when something about the channels or their oscillators has changed since
  // the call before (osc_dirty); a steady call walks no per-channel state on the host.
  const char *const steady_env = getenv("KQ_STEADY");  // A/B switch, read per call: 0 = stage every call on the host
  bool const steady_off = steady_env && atoi(steady_env) == 0;
  bool const steady = !b->osc_dirty && !spectrum && b->calls > 0 && !steady_off;
  if (!steady) {
    bool any = false;
    for (HostChan const &h : b->chans) any = any || h.active;
    b->cache_any = any;
  }  // (steady: kept up to date by kq_bank_add_channel / kq_bank_remove_channel)
  if (!b->cache_any) {
    set_err("no channels in bank");
    return -1;
  }
  if (b->lists_dirty && upload_lists(b)) return -1;
  // Launch decisions that depend on the channels' sweeps: counted over all channels when the bank is staged afresh, kept up
  // to date from the patch list otherwise.
  auto const sweep_of = [](HostChan const &h) { return h.lo2.sweep() + (h.dop.set_f != 0 ? h.dop.sweep() : 0.0); };
  auto const sweep_ok = [&](double r) {  // the pruned forward path's limits on a swept oscillator
    if (b->fwd_mode != KQ_FWD_PRUNED || r == 0) return true;
    // The pruned kernels take the sweep's cross term r*R*a*b to first order and drop its b^2 part:
    // both must stay far below the 1e-5 parity budget over one window.
    double const cross = 2 * M_PI * std::fabs(r) * (double)g.N * g.D, quad = std::fabs(r) * (double)g.D * g.D * 0.5;
    if (g.Ndec == 256) {
      set_err("swept NCO at N/D = 256 is only available on the full forward path: use KQ_FWD_FULL");
      return false;
    }
    if (cross > 3e-4 || quad > 2e-7) {
      set_err("sweep rate too large for the pruned forward path (cross term %.3g rad): use KQ_FWD_FULL", cross);
      return false;
    }
    return true;
  };
  double const fast_limit = b->use64k ? kq::full64k_sweep_limit() : kq::full16k_sweep_limit();
  if (!steady) {
    int n_swept = 0, n_fast = 0, n_active = 0;
    for (HostChan &h : b->chans) {
      if (!h.active) continue;
      double const r = sweep_of(h);
      if (!sweep_ok(r)) return -1;
      h.r_eff = r;
      n_active++;
      n_swept += r != 0;
      n_fast += std::fabs(r) > fast_limit;
    }
    b->n_swept = n_swept;
    b->n_fast = n_fast;
    b->n_active = n_active;
    b->sweep_lists_dirty = true;
  } else {
    // Rolling re-reference.  The device advances the oscillator planes from call to call; the host's closed forms, which a
    // retune is patched from, are referred to the sample they were last staged at and lose digits as that recedes (f k at
    // k = 2^24: 1e-9 turns).  Until round 5 the whole bank was therefore staged afresh every 2^24 samples -- at 32768
    // channels a call with 0.4-3 ms more host time and 2 MB over the link every 1.68 s: the periodic stall of a paced
    // receiver (tools/rt_stall_diag.sh: long delivery intervals at multiples of 1.68 s).  Now every call moves a few
    // channels' reference up and re-stages just those through the patch records (33 of 32768 per 2-block call), so that
    // each channel comes round at least once per 2^24 samples and no call is different from the others.
    {
      size_t const Cn = b->chans.size();
      double const span = (double)nblocks * (double)g.L;
      size_t want = (size_t)((double)Cn * span / (double)(1 << 24)) + 1;
      want = std::min({want, Cn, (size_t)kq_bank::kMaxPatch / 2});
      for (size_t i = 0; i < want && (int)b->patch_list.size() < kq_bank::kMaxPatch / 2; i++) {
        size_t const c = b->refresh_next < Cn ? b->refresh_next : 0;
        b->refresh_next = c + 1;
        HostChan &h = b->chans[c];
        if (!h.active || h.patched) continue;  // (a channel the operator has just set is staged from fresh forms anyway)
        h.lo2.rebase(b->n_abs);
        h.dop.rebase(b->n_abs);
        h.shift.rebase(b->out_abs);
        note_patch(b, (int)c);
      }
    }
    for (int c : b->patch_list) {
      if ((size_t)c >= b->chans.size()) continue;
      HostChan &h = b->chans[c];
      if (!h.active) continue;
      double const r = sweep_of(h);
      if (!sweep_ok(r)) return -1;
      if ((r != 0) != (h.r_eff != 0)) b->sweep_lists_dirty = true;
      b->n_swept += (r != 0) - (h.r_eff != 0);
      b->n_fast += (std::fabs(r) > fast_limit) - (std::fabs(h.r_eff) > fast_limit);
      h.r_eff = r;
    }
  }
  bool const swept = b->fwd_mode == KQ_FWD_PRUNED && b->n_swept > 0;
  if (swept) b->chan_tw_dirty = true;  // the step changes from call to call
  int const pp = (int)(b->calls & 1);
  size_t const Cmax = b->cfg.max_channels;
  kq::Planes pl = b->pl2[pp];
  kq::ChanDev chd = b->chd;
  chd.lo_phase = b->osc_dev2[pp];
  chd.lo_freq = chd.lo_phase + Cmax;
  chd.lo_rate = chd.lo_phase + 2 * Cmax;
  chd.sh_phase = chd.lo_phase + 3 * Cmax;
  chd.sh_freq = chd.lo_phase + 4 * Cmax;
  chd.hist_phase = chd.lo_phase + 5 * Cmax;
  chd.hist_freq = chd.lo_phase + 6 * Cmax;
  chd.hist_rate = chd.lo_phase + 7 * Cmax;
  // this parity's hand-over planes were last read by the demodulators two calls ago
  // (asked of the host first: a wait on another stream's event costs the stream a barrier packet, several microseconds
  // of idle device even when the event fired long ago -- and two calls later it always has)
  if (b->demod_overlapped[pp] && hipEventQuery(b->ev_demod_done[pp]) != hipSuccess)
    HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_demod_done[pp], 0));
  int slot = 0, nret = 0, npatch = 0;
  int64_t const n_w = b->n_abs - (g.M - 1);
  if (stage_call_params(b, n_w, update_host, nblocks, &slot, &nret, steady, &npatch)) return -1;
  // how far into the call's windows those channels' old samples reach (ChanDev::hist_len; M - 1 right after the retune), and
  // how many leading blocks the kernels that mix a whole window with one oscillator therefore leave to the per-sample variant
  unsigned nredo = 1;
  {
    int64_t deepest = 0;
    for (int c : b->ret_host) {
      HostChan &h = b->chans[c];
      deepest = std::max(deepest, h.hist_old);
      int const len = (int)std::min<int64_t>(h.hist_old, INT32_MAX);
      if (h.hist_dev != len) {
        if (ctl_put(b, CTL_FILTER, b->chd.hist_len + c, &len, sizeof len)) return -1;
        h.hist_dev = len;
      }
      for (int l = 0; l < kq::kOldLevels; l++) {
        size_t const k = (size_t)c * kq::kOldLevels + l;
        int const lenx = (int)std::min<int64_t>(h.hist_oldx[l], INT32_MAX);
        if (lenx > 0) {  // (its phase is the window start's: written for every call it lasts)
          double v[3];
          eval_older(h, l, n_w, v);
          if (ctl_put(b, CTL_FILTER, b->chd.hist2_osc + 3 * k, v, sizeof v)) return -1;
        }
        if (h.histx_dev[l] != lenx) {
          if (ctl_put(b, CTL_FILTER, b->chd.hist2_len + k, &lenx, sizeof lenx)) return -1;
          h.histx_dev[l] = lenx;
        }
      }
    }
    nredo = (unsigned)std::min<int64_t>(nblocks, std::max<int64_t>(1, (deepest + g.L - 1) / g.L));
  }
  size_t const ret_off = 8 * Cmax * sizeof(double) + ((b->cfg.max_blocks + 7) & ~7u);
  const int *retune_list = reinterpret_cast<const int *>(reinterpret_cast<const unsigned char *>(b->osc_dev2[pp]) + ret_off);
  // full-spectrum path: the register-resident N = 16384 kernel where it applies (KQ_FULL_LDS=1 forces the LDS one)
  static bool const lds_only = getenv("KQ_FULL_LDS") && atoi(getenv("KQ_FULL_LDS")) != 0;
  bool const use16k = !lds_only && kq::full16k_supported(g);
  // No sweep anywhere: the register-resident kernel runs without its per-sample oscillator path; the first block of
  // a channel retuned since the last call (history still on the old oscillator) is then redone below with the
  // general variant, as the pruned path does.
  // (N = 65536: the steady-state variant takes sweeps up to full64k_sweep_limit() itself)
  // (N = 16384: swept channels inside full16k_sweep_limit() have a steady-state variant of their own, `swept_steady`)
  bool const plain = b->use64k ? b->n_fast == 0 : b->n_swept == 0;
  static bool const swept_steady_off = getenv("KQ_SWEPT_STEADY") && atoi(getenv("KQ_SWEPT_STEADY")) == 0;  // A/B switch
  bool const swept_steady = use16k && !b->use64k && b->fwd_mode != KQ_FWD_PRUNED && !spectrum && b->n_swept > 0 && b->n_fast == 0 &&
                            !swept_steady_off;
  bool const swept64k = b->use64k && b->n_swept > 0;
  // N = 16384 with swept and unswept channels side by side: two launches over two lists (see list_unswept_dev)
  bool const mixed = use16k && !b->use64k && b->fwd_mode != KQ_FWD_PRUNED && !spectrum && b->n_swept > 0 && b->n_swept < b->n_active;
  if (mixed && b->sweep_lists_dirty) {
    b->list_unswept_host.clear();
    b->list_swept_host.clear();
    for (size_t c = 0; c < b->chans.size(); c++)
      if (b->chans[c].active) (b->chans[c].r_eff != 0 ? b->list_swept_host : b->list_unswept_host).push_back((int)c);
    if (ctl_put(b, CTL_FILTER, b->list_unswept_dev, b->list_unswept_host.data(), b->list_unswept_host.size() * sizeof(int))) return -1;
    if (ctl_put(b, CTL_FILTER, b->list_swept_dev, b->list_swept_host.data(), b->list_swept_host.size() * sizeof(int))) return -1;
    b->sweep_lists_dirty = false;
  }
  // That steady-state variant loads its samples 16 bytes at a time from a copy of the call's samples whose 512-sample
  // rows are interleaved in pairs; the IF-power kernel, which reads every new sample anyway, writes it
  float2 *const paired = ((use16k || b->use64k) && (plain || mixed || swept_steady) && b->win_paired && b->fwd_mode != KQ_FWD_PRUNED) ? b->win_paired : nullptr;
  // the control plane's filter-side writes since the last call, in front of this call's first kernel
  if (ctl_flush(b, CTL_FILTER, b->stream)) return -1;
  {
    Scope t(b, 2, b->stream);
    // the partial sums live behind the plane's max_blocks if_power values
    // (spectrum mode: no samples to sum -- the launch only carries the call's parameter block to the device)
    kq::launch_block_energy_sum(b->stream, spectrum ? nullptr : window + (g.M - 1), g.L, spectrum ? 0 : (int)nblocks,
                                pl.if_power + b->cfg.max_blocks,
                                b->stage_host[slot], b->osc_dev2[pp],
                                nret ? ret_off + nret * sizeof(int) : 8 * Cmax * sizeof(double) + nblocks,
                                spectrum ? nullptr : paired, (int)(g.M - 1), steady ? b->osc_dev2[pp ^ 1] : nullptr, (unsigned)C,
                                (unsigned)Cmax, (double)(n_w - b->planes_n_w), (double)(b->out_abs - b->planes_out_abs),
                                // the channels retuned since the last call: their planes, staged by the host, written by a
                                // few more workgroups of the same launch (the advancing threads skip those channels)
                                b->stage_host[slot] + b->patch_off, npatch, b->stage_host[slot] + b->bits_off);
  }
  LAUNCH_CHECK("IF power");
  if (b->timing) {
    HIP_TRY(hipEventRecord(b->stage_t0[slot], b->stream));
    b->stage_timed[slot] = true;
    b->stage_h0[slot] = std::chrono::steady_clock::now();
    b->stage_launch[slot] = b->acc.filter_launches;
  }
  // The IF-power recurrence rides in the first full-spectrum launch of the call (one wave of its first workgroup,
  // kq_full16k.hip) where there is one; KQ_IIR_IN_FILTER=0: as a launch of its own in front of the demodulators, as before
  static bool const iir_in_filter_off = getenv("KQ_IIR_IN_FILTER") && atoi(getenv("KQ_IIR_IN_FILTER")) == 0;
  bool iir_done = false;
  auto const iir_arm = [&]() {  // before a launch of k_filter_full16k: hand it the job once per call
    b->big.iir = kq::IirArgs{};
    if (iir_done || iir_in_filter_off || spectrum) return;
    b->big.iir.sums = pl.if_power + b->cfg.max_blocks;
    b->big.iir.update = reinterpret_cast<const unsigned char *>(b->osc_dev2[pp] + 8 * Cmax);
    b->big.iir.state = b->energy_state;
    b->big.iir.if_power = pl.if_power;
    b->big.iir.split = kq::block_energy_split(g.L);
    b->big.iir.nblocks = (int)nblocks;
    b->big.iir.L = g.L;
    iir_done = true;
  };
  {
    // `redo`: the list names channels retuned since the last call, which need the general variant
    // (as_plain: the steady-state variant -- the host vouches that no channel of THIS launch sweeps)
    auto const full_launch = [&](hipStream_t st, const kq::Geom &gg, const kq::ChanDev &cd, const kq::Planes &pp, const float2 *win,
                                 const float2 *twp, int nch, int nbl, int n0, float2 *dump, int dump_ch, const int *list,
                                 bool redo = true, bool as_plain = false, bool as_swept_steady = false) {
      // 0: the general variant; 1: steady state, no channel of the launch sweeps; 2: steady state, every one does
      int const pv = redo ? 0 : as_swept_steady ? 2 : (plain || as_plain) ? 1 : 0;
      if (use16k && b->fwd_mode != KQ_FWD_PRUNED) iir_arm();  // (not the pruned path's redo launches: they come second)
      if (use16k)
        kq::launch_filter_full16k(st, gg, cd, pp, win, twp, nch, nbl, n0, dump, dump_ch, list, pv, pv ? paired : nullptr, b->big);
      else
        kq::launch_filter_full(st, gg, cd, pp, win, twp, nch, nbl, n0, dump, dump_ch, list);
    };
    if (spectrum) {
      // execute_filter_output (filter.c:206-250) and compute_n0 (radio.c:383-425) on the spectrum handed in: one small
      // launch each per channel-block (this is the one-channel, one-block path of the demodulator thread entry points)
      for (int c = 0; c < C; c++) {
        if (!b->chans[c].active) continue;
        bool const isb = b->chans[c].cfg.demod_type == KQ_LINEAR_DEMOD && b->chans[c].cfg.isb;
        for (unsigned k = 0; k < nblocks; k++) {
          const float2 *X = spectrum + (size_t)k * g.N;
          size_t const cb = (size_t)c * g.max_blocks + k;
          if (b->cfg.compute_n0)
            kq::launch_n0_single(b->stream, X, g.N, g.samprate, b->chans[c].cfg.low, b->chans[c].cfg.high, pl.n0raw + cb);
          kq::launch_slave_bank(b->stream, X, chd.resp + (size_t)c * g.Ndec, pl.filt + cb * g.olen, g.N, g.Ndec, g.olen, isb ? 2 : 1,
                                b->tw, g.tw_log2);
        }
      }
    } else if (b->fwd_mode == KQ_FWD_PRUNED) {
      if (b->chan_tw_dirty) {  // the tables depend only on each channel's LO step: rebuild after a retune
        kq::launch_pruned_tables(b->stream, g, chd, b->chan_tw, C);
        b->chan_tw_dirty = false;
      }
      {  // slots emptied by remove_channel are skipped: the launch goes over the list of active channels then
        bool const holes = b->list_active_host.size() != b->chans.size();
        if (kq::pruned_carries_iir(g)) iir_arm();
        kq::launch_filter_pruned(b->stream, g, chd, pl, window, b->chan_tw, holes ? (int)b->list_active_host.size() : C,
                                 (int)nblocks, swept, holes ? b->list_active_dev : nullptr, b->big.iir);
        b->big.iir = kq::IirArgs{};
      }
      // The pruned kernels assume one oscillator over the whole window.  For the first block after a retune the
      // history half still carries the old one: redo just those channel-blocks on the per-sample path.
      if (nret > 0) {
        if (g.N > 16384)
          kq::launch_filter_split(b->stream, g, chd, pl, window, b->tw, nret, (int)nredo, retune_list);
        else
          full_launch(b->stream, g, chd, pl, window, b->tw, nret, (int)nredo, 0, nullptr, -1, retune_list);
      }
    } else if (b->use64k) {
      bool const holes = b->list_active_host.size() != b->chans.size();
      auto const launch64k = [&](int nch, int nbl, const int *list, bool steady) {
        iir_arm();
        kq::Big64 big = b->big;
        big.epoch = ++b->big.epoch;  // never 0: the words start out zeroed
        if (big.epoch == 0) big.epoch = ++b->big.epoch;
        kq::launch_filter_full64k(b->stream, g, chd, pl, window, b->tw, nch, nbl, b->cfg.compute_n0, b->spec_dump, b->spec_ch,
                                  list, steady, swept64k, steady ? paired : nullptr, big);
      };
      launch64k(holes ? (int)b->list_active_host.size() : C, (int)nblocks, holes ? b->list_active_dev : nullptr, plain);
      // the steady-state variant mixes a whole window with one oscillator: the first block of a channel retuned since
      // the last call (history still on the old one) is redone with the per-sample variant
      if (plain && nret > 0) launch64k(nret, (int)nredo, retune_list, false);
    } else if (g.N > 16384) {
      kq::launch_filter_split(b->stream, g, chd, pl, window, b->tw, C, (int)nblocks, nullptr);
    } else {
      // slots emptied by remove_channel are skipped: the launch goes over the list of active channels then
      bool const holes = b->list_active_host.size() != b->chans.size();
      if (mixed) {
        full_launch(b->stream, g, chd, pl, window, b->tw, (int)b->list_unswept_host.size(), (int)nblocks, b->cfg.compute_n0,
                    b->spec_dump, b->spec_ch, b->list_unswept_dev, false, true);
        full_launch(b->stream, g, chd, pl, window, b->tw, (int)b->list_swept_host.size(), (int)nblocks, b->cfg.compute_n0,
                    b->spec_dump, b->spec_ch, b->list_swept_dev, false, false, swept_steady);
      } else {  // (nobody sweeps, or everybody does)
        full_launch(b->stream, g, chd, pl, window, b->tw, holes ? (int)b->list_active_host.size() : C, (int)nblocks,
                    b->cfg.compute_n0, b->spec_dump, b->spec_ch, holes ? b->list_active_dev : nullptr, false, false, swept_steady);
      }
      if (use16k && (plain || mixed || swept_steady) && nret > 0)
        full_launch(b->stream, g, chd, pl, window, b->tw, nret, (int)nredo, b->cfg.compute_n0, b->spec_dump, b->spec_ch, retune_list);
    }
    LAUNCH_CHECK("pre-detection filter");
    b->acc.filter_launches++;
    b->acc.channel_blocks += (uint64_t)C * nblocks;
  }
  HIP_TRY(hipEventRecord(b->stage_ev[slot], b->stream));
  if (b->timing) b->stage_submit_ms[slot] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b->stage_h0[slot]).count();
  // Where the demodulators of this call run.  On a second stream they overlap the next call's filter pass -- but that
  // kernel leaves no room beside it (500 of a SIMD's 512 registers), so what they take it loses, and at N = 65536 a
  // displaced sibling workgroup stalls the three that wait for it.  Measured on one box (tools/ab_env_rows.sh,
  // tools/ab_hostio.sh; ms per step overlapped / not): cfg 4 1.432 / 1.432-1.439, pruned 0.821 / 0.819, cfg 2 0.475 /
  // 0.469, cfg 5 1.31 / 1.15 -- but cfg 3 1.464 / 1.488 (the AGC recurrence of an AM or SSB channel is one long thin wave)
  // and, with the planes streamed to the host after every call, 1.54 / 1.60-1.76 (the copy kernel then has the whole next
  // filter pass to hide under).  KQ_DEMOD_OVERLAP=0 / 1 forces either.
  bool const agc_channels = !b->list_host[1].empty() || !b->list_host[2].empty();
  // (round 5, cfg 2 with its fused FM demodulator: overlapped 0.4175-0.4196 ms per step against 0.4194-0.4195 on one box,
  //  0.4427 against 0.4475 on a slower one, with the filter launch itself 9 % longer -- left where it is)
  bool const overlap = b->stream2 != b->stream &&
                       (b->overlap_mode == 1 || (b->overlap_mode < 0 && !b->use64k && (b->pulled_since_call || agc_channels)));
  b->pulled_since_call = false;
  hipStream_t const ds = overlap ? b->stream2 : b->stream;
  int const prev = pp ^ 1;
  if (overlap)
    HIP_TRY(hipStreamWaitEvent(ds, b->stage_ev[slot], 0));  // behind this call's filter, hence behind the last call's demodulators if they ran on the main stream
  else if (b->calls > 0 && b->demod_overlapped[prev])
    HIP_TRY(hipStreamWaitEvent(ds, b->ev_demod_done[prev], 0));  // the channel state they carry
  if (b->out_pending) {  // kq_bank_pull_planes_async is still reading the audio / status planes of the last call
    HIP_TRY(hipStreamWaitEvent(ds, b->pull_done[(b->pulls - 1) % kq_bank::kPullRing], 0));
    b->out_pending = false;
  }
  // the IF-power recurrence over the call's blocks: one wave, consumed by the demodulators only.  Where the call has a
  // full-spectrum launch it has ridden in that (iir_arm above: -5..6 us per step at cfg 4 and cfg 5, the filter kernel's own
  // time unchanged; tools/ab_libs.sh against the commit before); otherwise it runs here, with the demodulators and
  // not in front of the filter (folded into the sum's launch -- its last workgroup taking tagged partial sums as they
  // arrive -- it saved nothing measurable: 1.422 against 1.421 ms per step; on the second stream beside the filter pass,
  // with an event each way, it cost 7-9 us per step where the demodulators run on the main stream -- a wait on another
  // queue's event is dearer than the 6.5 us kernel: cfg 2 0.4326 -> 0.4416, cfg 4 1.5050 -> 1.5124, tools/ab_env.sh)
  // the control plane's demodulator-side writes, on the stream this call's demodulators run on, in front of them
  if (ctl_flush(b, CTL_DEMOD, ds)) return -1;
  if (spectrum)  // the IF power belongs to whoever fed the master (radio.c:123,143-145): status.if_power = 0, not what a
                 // normal call two calls back left in this parity's plane
    HIP_TRY(hipMemsetAsync(pl.if_power, 0, nblocks * sizeof(float), ds));
  else if (!iir_done)
    kq::launch_block_energy_iir(ds, pl.if_power + b->cfg.max_blocks, g.L, (int)nblocks,
                                reinterpret_cast<const unsigned char *>(b->osc_dev2[pp] + 8 * Cmax), b->energy_state, pl.if_power);
  {
    Scope t(b, 1, ds);
    int const nfm = (int)b->list_host[0].size(), nam = (int)b->list_host[1].size(), nlin = (int)b->list_host[2].size();
    if (kq::demod64_supported(g)) {
      kq::launch_demod64(ds, g, chd, pl, b->list_dev[0], nfm, b->list_dev[1], nam, b->list_dev[2], nlin, (int)nblocks,
                         b->cfg.compute_n0);
    } else if (kq::demod_agc_wave_supported(g)) {  // wave-per-channel AM / linear, generic FM
      kq::launch_demod64(ds, g, chd, pl, b->list_dev[0], 0, b->list_dev[1], nam, b->list_dev[2], nlin, (int)nblocks,
                         b->cfg.compute_n0);
      kq::launch_demods(ds, g, chd, pl, b->tw, b->list_dev[0], nfm, b->list_dev[1], 0, b->list_dev[2], 0, (int)nblocks,
                        b->cfg.compute_n0, b->fmout, b->fm_hist[b->fm_hist_cur], b->fm_hist[b->fm_hist_cur ^ 1]);
      if (nfm > 0) b->fm_hist_cur ^= 1;
    } else {
      kq::launch_demods(ds, g, chd, pl, b->tw, b->list_dev[0], nfm, b->list_dev[1], nam, b->list_dev[2], nlin,
                        (int)nblocks, b->cfg.compute_n0, b->fmout, b->fm_hist[b->fm_hist_cur], b->fm_hist[b->fm_hist_cur ^ 1]);
      if (nfm > 0) b->fm_hist_cur ^= 1;
    }
  }
  LAUNCH_CHECK("demodulators");
  if (!b->list_pll_host.empty())
    kq::launch_demod_pll(ds, g, chd, pl, b->tw, b->list_pll_dev, (int)b->list_pll_host.size(), b->pll_chunks_dev, b->pll_slot_dev,
                         (int)nblocks, b->cfg.compute_n0);
  if (g.pl_n > 0 && !b->list_host[0].empty())
    kq::launch_pl_track(ds, g, chd, pl, b->tw, b->list_dev[0], (int)b->list_host[0].size(), (int)nblocks);
  if (b->pcm_on) {
    bool const holes = b->list_active_host.size() != b->chans.size();
    kq::launch_pcm(ds, g, pl, b->pcm, b->pcm_mask, holes ? (int)b->list_active_host.size() : C, (int)nblocks,
                   holes ? b->list_active_ds_dev : nullptr);
  }
  if (overlap) HIP_TRY(hipEventRecord(b->ev_demod_done[pp], ds));
  b->demod_overlapped[pp] = overlap;
  LAUNCH_CHECK("PLL / PL tone / PCM stage");
  b->pl = pl;  // what the pull functions read
  b->calls++;
  b->planes_n_w = n_w;  // what this parity's planes on the device describe
  b->planes_out_abs = b->out_abs;
  b->n_abs += (int64_t)nblocks * g.L;
  b->out_abs += (int64_t)nblocks * g.olen;
  for (int c : b->patch_list)
    if ((size_t)c < b->chans.size()) b->chans[c].patched = false;
  b->patch_list.clear();
  // the channels that went through this call with samples of an old oscillator in their windows: the call has moved the
  // windows on by nblocks L; whoever has old samples left (M - 1 > L and a short call) goes through the next call the same way
  for (int c : b->ret_host) {
    if ((size_t)c >= b->chans.size()) continue;
    HostChan &h = b->chans[c];
    if (!h.retuned) continue;
    h.hist_old -= (int64_t)nblocks * g.L;
    for (int64_t &n : h.hist_oldx) n = std::max<int64_t>(0, n - (int64_t)nblocks * g.L);
    if (h.hist_old > 0 && h.active) {
      note_patch(b, c);
    } else {
      h.retuned = false;
      h.hist_old = 0;
      for (int64_t &n : h.hist_oldx) n = 0;
    }
  }
  b->ret_host.clear();
  if (!steady) {
    for (HostChan &h : b->chans) {
      h.lo2.rebase(b->n_abs);
      h.dop.rebase(b->n_abs);
      h.shift.rebase(b->out_abs);
    }
    b->rebased_at = b->n_abs;
    b->osc_dirty = spectrum != nullptr;  // (a spectrum call stages no oscillators the next call could advance)
  }  // (steady calls keep the host's closed forms fresh a few channels at a time: the rolling re-reference above)
  b->last_blocks = nblocks;
  return (int)nblocks;
}

bool valid_ch(const kq_bank *b, int ch) { return b && ch >= 0 && (size_t)ch < b->chans.size() && b->chans[ch].active; }

// The second LO or the Doppler oscillator of channel `ch` is about to be set: the samples mixed so far keep what they were
// mixed with (radio.c:132-139 mixes sample by sample, osc.c:22-36 changes only what follows), and the next M - 1 of them are
// the history of the windows to come.  Several settings between two calls are one transition (nothing was mixed with the
// ones in between).  A setting while an EARLIER transition is still inside the history -- possible only where M - 1 > L: a
// channel retuned before every block at the reference's default -L 3840 -M 4353 -- moves that one, and any before it, one
// level down the list the kernels know per window (hist2_*: kOldLevels of them beyond the last); one more than that inside
// the same M - 1 samples (M - 1 > 5 L and a retune before every block) drops the oldest.
void note_retune(kq_bank *b, int ch) {
  HostChan &h = b->chans[ch];
  int64_t const hist = (int64_t)b->g.M - 1;
  if (h.retuned && h.hist_old == hist) return;  // set again before anything was mixed with the setting in between
  if (h.retuned) {  // the transition before this one still has samples in the history: it and its elders move one level down
    for (int l = kq::kOldLevels - 1; l > 0; l--) {
      h.lo2_oldx[l] = h.lo2_oldx[l - 1];
      h.dop_oldx[l] = h.dop_oldx[l - 1];
      h.hist_oldx[l] = h.hist_oldx[l - 1];
    }
    h.lo2_oldx[0] = h.lo2_old;
    h.dop_oldx[0] = h.dop_old;
    h.hist_oldx[0] = h.hist_old;
  }
  h.lo2_old = h.lo2;
  h.dop_old = h.dop;
  h.retuned = true;
  h.hist_old = hist;
}

// an oscillator of channel `ch` has been set (or the channel is new): the next call patches its planes -- or, with too
// many of them, stages the whole bank
void note_patch(kq_bank *b, int ch) {
  HostChan &h = b->chans[ch];
  if (h.patched) return;
  if ((int)b->patch_list.size() >= kq_bank::kMaxPatch) {
    b->osc_dirty = true;
    return;
  }
  h.patched = true;
  b->patch_list.push_back(ch);
}

}  // namespace

// error reporting for the other translation units of the library (kq_decim.hip)
void kq_internal_set_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

extern "C" {

const char *kq_last_error(void) { return g_err.c_str(); }
const char *kq_version(void) { return "ka9q_hip 0.1 (gfx950)"; }
int kq_abi_version(void) { return KQ_ABI_VERSION; }

int kq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

void *kq_host_alloc(size_t bytes) {
  void *p = nullptr;
  hipError_t const e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
  if (e != hipSuccess) {
    set_err("hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}
void kq_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

kq_bank *kq_bank_create(const kq_bank_config *cfg) {
  if (!cfg) {
    set_err("NULL config");
    return nullptr;
  }
  unsigned const N = cfg->L + cfg->M - 1;
  // FFTW plans any N (filter.c:78); here a power of two, or -- on the generic kernels, one LDS block -- an even 2^a 3^b 5^c 7^d
  // up to 16384 (a front end whose rate is not 48 kHz x 2^k: 240 kHz gives decimate 5, radio_status.c:266)
  bool const n_pow2 = (N & (N - 1)) == 0;
  if (cfg->L == 0 || cfg->M < 2 || N < 16 || (!n_pow2 && (!kq::fft_size_ok((int)N) || N > 65536))) {
    set_err("L+M-1 = %u must be a power of two >= 16, or an even 2^a 3^b 5^c 7^d in 16..65536", N);
    return nullptr;
  }
  if (cfg->decimate < 1 || N % cfg->decimate != 0 || cfg->L % cfg->decimate != 0 || (cfg->M - 1) % cfg->decimate != 0) {
    set_err("decimate %u must be >= 1 and divide N, L and M-1", cfg->decimate);
    return nullptr;
  }
  // decimate = samprate / 48000 = 1 (radio_status.c:266: a 48 kHz front end, a sound-card receiver): the slave's transform is
  // as long as the master's and needs a buffer of its own beside it in LDS (k_filter_full)
  if (cfg->decimate == 1 && N > 8192) {
    set_err("decimate 1 needs L+M-1 = %u <= 8192 (master and slave transform side by side in LDS)", N);
    return nullptr;
  }
  unsigned const Ndec = N / cfg->decimate;
  if (Ndec < 4 || ((Ndec & (Ndec - 1)) != 0 && (!kq::fft_size_ok((int)Ndec) || Ndec > 16384))) {
    set_err("N/decimate = %u must be a power of two >= 4, or an even 2^a 3^b 5^c 7^d in 4..16384", Ndec);
    return nullptr;
  }
  if (cfg->max_channels == 0 || cfg->max_blocks == 0 || cfg->samprate <= 0) {
    set_err("max_channels, max_blocks and samprate must be positive");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err("no HIP device available (the gfx950 kernels are the only compute path)");
    return nullptr;
  }
  if (cfg->device < 0 || cfg->device >= ndev) {
    set_err("device %d out of range (%d visible)", cfg->device, ndev);
    return nullptr;
  }
  // the handle lives on cfg->device; the calling thread's current device is put back on the way out, error paths
  // included, like every other entry point (kq_device.hpp DeviceScope)
  kq::DeviceScope dev_scope_(cfg->device);
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) {
      set_err("hipSetDevice(%d) failed", cfg->device);
      return nullptr;
    }
  }

  kq_bank *b = new kq_bank();
  b->cfg = *cfg;
  b->chans.reserve(cfg->max_channels);  // the elements never move (kq_bank_rtp_from_planes keeps an address past the lock)
  kq::Geom &g = b->g;
  g.N = (int)N;
  g.L = (int)cfg->L;
  g.M = (int)cfg->M;
  g.D = (int)cfg->decimate;
  g.Ndec = (int)Ndec;
  g.olen = (int)(cfg->L / cfg->decimate);           // filter.c:116
  g.Mdec = (int)((cfg->M - 1) / cfg->decimate + 1); // filter.c:514
  g.log2N = ilog2(N);
  g.log2Ndec = ilog2(Ndec);
  g.samprate = cfg->samprate;
  g.tw_log2 = g.log2N < 16 ? 16 : g.log2N;  // the PL tracker (16384) and the PLL carrier search (65536) need these periods
  {
    // pltask geometry (fm.c:201-205): decimate 32 from the audio master; needs a usable transform size
    int const pn = g.Ndec / 32, plen = g.olen / 32;
    // (where 32 does not divide N/decimate or the block, create_filter_output warns and truncates both, filter.c:103-107,116:
    //  the slave then resamples by N_dec / PL_N instead of 32 and the tone reads that much off -- what the reference does, and
    //  what happens here; only a PL_N this library has no transform for -- odd, or with a prime factor beyond 7 -- has no PL slave)
    bool const ok = pn >= 4 && plen >= 1 && !cfg->pl_tone_off && ((pn & (pn - 1)) == 0 || kq::fft_size_ok(pn));
    g.pl_n = ok ? pn : 0;
    g.pl_l = ok ? plen : 0;
  }
  {  // the generic path's transform plans (powers of two: no tables)
    bool ok_n = false, ok_d = false, ok_p = true;
    g.dN = kq::fft_dim(g.N, &ok_n);
    g.dNdec = kq::fft_dim(g.Ndec, &ok_d);
    g.dPl = g.pl_n > 0 ? kq::fft_dim(g.pl_n, &ok_p) : kq::FftDim{};
    if (!ok_n || !ok_d || !ok_p) {
      set_err("transform plan for N = %d / N/decimate = %d failed", g.N, g.Ndec);
      delete b;
      return nullptr;
    }
  }
  g.max_blocks = (int)cfg->max_blocks;
  g.dsamprate = (float)cfg->samprate / cfg->decimate;

  // N = 65536 with compute_n0 (cfg 5 as the reference runs it), or the full path asked for by name: the 16384-point
  // register kernel in four sibling workgroups per channel-block (KQ_FULL_SPLIT=1 keeps the older split kernel, no n0)
  static bool const split_only = getenv("KQ_FULL_SPLIT") && atoi(getenv("KQ_FULL_SPLIT")) != 0;
  bool const can64k = kq::full64k_supported(g) && !split_only;
  bool const can_prune = kq::pruned_supported(g) && !cfg->compute_n0;
  if (cfg->fwd_mode == KQ_FWD_PRUNED) {
    if (!can_prune) {
      set_err("pruned forward path unavailable for this geometry or with compute_n0 enabled");
      delete b;
      return nullptr;
    }
    b->fwd_mode = KQ_FWD_PRUNED;
  } else if (cfg->fwd_mode == KQ_FWD_FULL) {
    b->fwd_mode = KQ_FWD_FULL;
  } else {
    // N/D = 256 (cfg 2): the pruned kernel is correct but slower than the full-spectrum kernel (2.24 vs 0.45 ms per
    // 16384 channel-blocks), so AUTO keeps the full path there; KQ_FWD_PRUNED still selects it explicitly
    b->fwd_mode = (can_prune && g.Ndec != 256) ? KQ_FWD_PRUNED : KQ_FWD_FULL;
  }
  b->use64k = b->fwd_mode == KQ_FWD_FULL && can64k;
  if (b->fwd_mode == KQ_FWD_FULL && N > 16384 && !b->use64k && (!kq::split_supported(g) || cfg->compute_n0)) {
    set_err("N = %u: the full path beyond 16384 points needs N = 65536, or compute_n0 off and a split N = S x N1 with N1 <= 16384 a "
            "size of its own (N = 32768; 19200, 24000, 38400, 48000 ...) and N/D small enough to sit beside it in LDS", N);
    delete b;
    return nullptr;
  }

  if (cfg->stream) {
    b->stream = (hipStream_t)cfg->stream;
  } else {
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
      set_err("hipStreamCreate failed");
      delete b;
      return nullptr;
    }
    b->own_stream = true;
  }

  size_t const C = cfg->max_channels, B = cfg->max_blocks;
  // max_blocks blocks plus L - 1 samples of slack: a partly filled block never stands in the way of a batch that fits the
  // ring as such (kq_bank_push_rtp); kq_bank_process still takes at most max_blocks blocks per call
  b->ring_cap = (size_t)(g.M - 1) + B * (size_t)g.L + (size_t)(g.L - 1);
  int rc = 0;
  rc |= dev_alloc(&b->ring[0], b->ring_cap);
  rc |= dev_alloc(&b->ring[1], b->ring_cap);
  rc |= dev_alloc(&b->tw, (size_t)1 << (g.tw_log2 - 1));
  rc |= dev_alloc(&b->chd.mode, C);
  rc |= dev_alloc(&b->chd.flags, C);
  rc |= dev_alloc(&b->chd.low, C);
  rc |= dev_alloc(&b->chd.high, C);
  rc |= dev_alloc(&b->chd.resp, C * g.Ndec);
  rc |= dev_alloc(&b->chd.aresp, C * (g.Ndec / 2 + 1));
  rc |= dev_alloc(&b->chd.fm_gain, C);
  rc |= dev_alloc(&b->chd.headroom, C);
  rc |= dev_alloc(&b->chd.recovery, C);
  rc |= dev_alloc(&b->chd.hangmax, C);
  rc |= dev_alloc(&b->chd.noise_gain, C);
  rc |= dev_alloc(&b->chd.hist_len, C);
  rc |= dev_alloc(&b->chd.hist2_len, C * kq::kOldLevels);
  rc |= dev_alloc(&b->chd.hist2_osc, 3 * C * kq::kOldLevels);
  b->chd.n0lane = nullptr;
  b->chd.n0meta = nullptr;
  b->chd.n0slot = nullptr;
  if (b->cfg.compute_n0 && kq::full16k_supported(g)) {
    rc |= dev_alloc(&b->chd.n0lane, C * 256);
    rc |= dev_alloc(&b->chd.n0meta, C);
    rc |= dev_alloc(&b->chd.n0slot, C);
  }
  if (b->use64k) {
    if (b->cfg.compute_n0) {
      rc |= dev_alloc(&b->chd.n0lane, C * 4 * 256);
      rc |= dev_alloc(&b->chd.n0meta, C * 4);
      rc |= dev_alloc(&b->chd.n0slot, C);
    }
    rc |= dev_alloc(&b->big.sync, C * cfg->max_blocks * 12);
    rc |= dev_alloc(&b->big.n0part, C * cfg->max_blocks * 4);
    rc |= dev_alloc(&b->big.xs, C * cfg->max_blocks * (size_t)g.Ndec);
    if (hipHostMalloc((void **)&b->big.err, sizeof(int), hipHostMallocDefault) != hipSuccess) {
      set_err("pinned allocation failed");
      rc = -1;
    } else {
      *b->big.err = 0;
    }
  }
  // eight oscillator planes + the per-block IF-power flags of one call
  // eight oscillator planes | the per-block IF-power flags | the list of channels retuned since the last call
  for (int k = 0; k < 2; k++)
    rc |= dev_alloc(&b->osc_dev2[k], 8 * C + (B + sizeof(double) - 1) / sizeof(double) + (C * sizeof(int) + 7) / 8);
  b->chd.lo_phase = b->chd.lo_freq = b->chd.lo_rate = b->chd.sh_phase = b->chd.sh_freq = nullptr;  // set per call
  b->chd.hist_phase = b->chd.hist_freq = b->chd.hist_rate = nullptr;
  // the demodulators' own stream (run_blocks says when it is used); KQ_DEMOD_OVERLAP=0 leaves it out
  const char *ov = getenv("KQ_DEMOD_OVERLAP");
  b->overlap_mode = ov ? (atoi(ov) != 0 ? 1 : 0) : -1;
  bool const overlap = b->overlap_mode != 0;
  if (!overlap) b->stream2 = b->stream;
  // (its stream at high priority -- so that the demodulators' few workgroups are placed ahead of the next filter launch's --
  // was measured in round 4: with_host_io 0.50 -> 0.78-1.25 ms at cfg 2, 1.45 -> 1.72 at cfg 4; at the LOWEST priority --
  // so that the next call's first kernels are not dispatched behind the demodulators' waiting workgroups -- round 5:
  // 1.483 -> 1.692 ms per call at 32768 channels x 2 blocks, the demodulators starve until the filter pass ends and the
  // call after next waits for them; default priority it is)
  if ((overlap && hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking) != hipSuccess) ||
      hipEventCreateWithFlags(&b->ev_demod_done[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&b->ev_demod_done[1], hipEventDisableTiming) != hipSuccess) {
    set_err("second stream / event creation failed");
    rc = -1;
  }
  b->stage_bytes = 8 * C * sizeof(double) + ((B + 7) & ~(size_t)7) + ((C * sizeof(int) + 7) & ~(size_t)7);  // copied in 8-byte words
  b->patch_off = b->stage_bytes;               // the retune patches' records (never copied as a whole: the patch role of k_block_energy_sum reads them)
  b->stage_bytes += (size_t)kq_bank::kMaxPatch * kq_bank::kPatchBytes;
  b->bits_off = b->stage_bytes;
  b->stage_bytes += ((C + 63) / 64) * sizeof(unsigned long long);
  static_assert(kq_bank::kSlots == 4, "slot_bit_words");
  for (int k = 0; k < kq_bank::kSlots && !rc; k++) {
    if (hipHostMalloc((void **)&b->stage_host[k], b->stage_bytes, hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&b->stage_ev[k]) != hipSuccess || hipEventCreate(&b->stage_t0[k]) != hipSuccess) {
      set_err("pinned staging allocation failed");
      rc = -1;
    } else {
      memset(b->stage_host[k] + b->bits_off, 0, b->stage_bytes - b->bits_off);
    }
  }
  rc |= dev_alloc(&b->chd.fm_state, C);
  rc |= dev_alloc(&b->chd.lastaudio, C);
  rc |= dev_alloc(&b->chd.sq_count, C);
  rc |= dev_alloc(&b->chd.ahist, C * (size_t)(g.Mdec > 1 ? g.Mdec - 1 : 1));
  if (!kq::demod64_supported(g)) {
    rc |= dev_alloc(&b->fmout, C * B * (size_t)g.olen);
    for (int k = 0; k < 2; k++) rc |= dev_alloc(&b->fm_hist[k], C * (size_t)(g.Mdec > 1 ? g.Mdec - 1 : 1));
  }
  rc |= dev_alloc(&b->chd.foffset, C);
  rc |= dev_alloc(&b->chd.pdev, C);
  rc |= dev_alloc(&b->chd.gain, C);
  rc |= dev_alloc(&b->chd.hang, C);
  rc |= dev_alloc(&b->chd.dc, C);
  rc |= dev_alloc(&b->chd.n0, C);
  b->chd.plresp = nullptr;
  b->chd.plring = nullptr;
  rc |= dev_alloc(&b->chd.pl_ptr, C);
  rc |= dev_alloc(&b->chd.pl_last, C);
  rc |= dev_alloc(&b->chd.plfreq, C);
  if (g.pl_n > 0) {
    rc |= dev_alloc(&b->chd.plresp, (size_t)g.pl_n / 2 + 1);
    rc |= dev_alloc(&b->chd.plring, C * 16384);
  }
  rc |= dev_alloc(&b->pl.audio, C * B * 2 * (size_t)g.olen);
  rc |= dev_alloc(&b->pl.status, C * B);
  for (int k = 0; k < 2; k++) {  // filter -> demod hand-over planes, one set per call parity
    b->pl2[k].audio = b->pl.audio;
    b->pl2[k].status = b->pl.status;
    rc |= dev_alloc(&b->pl2[k].filt, C * B * g.olen);
    rc |= dev_alloc(&b->pl2[k].n0raw, C * B);
    rc |= dev_alloc(&b->pl2[k].if_power, B * (1 + kq::kEnergySplitMax));  // + the partial sums of k_block_energy_sum
    b->pl2[k].plout = nullptr;
    if (g.pl_n > 0) rc |= dev_alloc(&b->pl2[k].plout, C * B * g.pl_l);
  }
  b->pl = b->pl2[0];
  rc |= dev_alloc(&b->energy_state, 2);
  if (kq::full16k_paired_supported(b->g)) rc |= dev_alloc(&b->win_paired, (size_t)(b->g.M - 1) + (size_t)B * b->g.L);
  for (int k = 0; k < 3; k++) rc |= dev_alloc(&b->list_dev[k], C);
  for (kq_bank::CtlQueue &q : b->ctl)
    for (int k = 0; k < kq_bank::CtlQueue::kDepth; k++)
      if (hipHostMalloc((void **)&q.buf[k], kq_bank::CtlQueue::kBytes, hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&q.applied[k], hipEventDisableTiming) != hipSuccess) {
        set_err("pinned control queue allocation failed");
        rc = -1;
      }
  {
    kq_bank::DesignQueue &d = b->dq;
    size_t const per_job = (size_t)g.Ndec * sizeof(float2);
    d.max_jobs = (unsigned)std::max<size_t>(1, std::min<size_t>(kq_bank::DesignQueue::kMax, ((size_t)64 << 20) / per_job));
    rc |= dev_alloc(&d.scratch, (size_t)d.max_jobs * g.Ndec);
    rc |= dev_alloc(&d.ng_next, 2 * C);
    for (int k = 0; k < kq_bank::DesignQueue::kDepth; k++)
      if (hipHostMalloc((void **)&d.pin[k], kq_bank::DesignQueue::kMax * (sizeof(kq::DesignJob) + sizeof(kq::DesignTarget)),
                        hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&d.read[k], hipEventDisableTiming) != hipSuccess)
        rc = -1;
    for (int k = 0; k < 2; k++)
      if (hipEventCreateWithFlags(&d.ng_moved[k], hipEventDisableTiming) != hipSuccess) rc = -1;
    if (g.Ndec <= 16384 && kq::design_prepare(g.olen, g.Mdec)) rc = -1;  // (the twiddle table of the design kernel, built now)
  }
  rc |= dev_alloc(&b->list_active_dev, C);
  rc |= dev_alloc(&b->list_active_ds_dev, C);
  rc |= dev_alloc(&b->chd.fflags, C);
  rc |= dev_alloc(&b->list_unswept_dev, C);
  rc |= dev_alloc(&b->list_swept_dev, C);
  if (b->fwd_mode == KQ_FWD_PRUNED) rc |= dev_alloc(&b->chan_tw, C * kq::pruned_table_elems(g));
  if (rc) {
    kq_bank_destroy(b);
    return nullptr;
  }
  if (g.pl_n > 0) {
    // PL low-pass: bins with 0 < f < 300 Hz, Kaiser beta 2.0 (fm.c:207-218)
    int const PL_M = g.pl_n - g.pl_l + 1;
    std::vector<kq::cfloat> r(g.pl_n / 2 + 1, kq::cfloat(0, 0));
    for (int j = 0; j <= g.pl_n / 2; j++) {
      float const f = (float)j * g.dsamprate / g.Ndec;
      if (f > 0 && f < 300) r[j] = 1;
    }
    if (kq::window_rfilter(g.pl_l, PL_M, r, 2.0)) {
      kq_bank_destroy(b);
      return nullptr;
    }
    (void)hipMemcpy(b->chd.plresp, r.data(), r.size() * sizeof(float2), hipMemcpyHostToDevice);
  }
  {
    std::vector<float> nanv(C, NAN);
    (void)hipMemcpy(b->chd.plfreq, nanv.data(), C * sizeof(float), hipMemcpyHostToDevice);
  }
  // twiddles exp(-2*pi*i*k/T) in double, rounded once
  {
    size_t const T = (size_t)1 << g.tw_log2;
    std::vector<float2> tw(T / 2);
    for (size_t k = 0; k < T / 2; k++) {
      double const a = -2.0 * M_PI * (double)k / (double)T;
      tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    if (hipMemcpy(b->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
      set_err("twiddle upload failed");
      kq_bank_destroy(b);
      return nullptr;
    }
  }
  return b;
}

int kq_bank_destroy(kq_bank *b) {
  kq::DeviceScope dev_scope_(b ? b->cfg.device : -1);  // (not the handle's lock: nobody else may be using a handle that is being destroyed)
  if (!b) return 0;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->stream2 && b->stream2 != b->stream) (void)hipStreamSynchronize(b->stream2);
  for (hipStream_t st : {b->copy_in, b->copy_out})  // before any plane they read or write is freed
    if (st) (void)hipStreamSynchronize(st);
  void *ptrs[] = {b->ring[0], b->ring[1], b->tw, b->chan_tw, b->chd.mode, b->chd.flags, b->chd.low, b->chd.high, b->chd.resp,
                  b->chd.aresp, b->chd.fm_gain, b->chd.headroom, b->chd.recovery, b->chd.hangmax, b->chd.noise_gain, b->chd.hist_len, b->chd.hist2_len, b->chd.hist2_osc, b->chd.n0lane, b->chd.n0meta, b->chd.n0slot, b->fmout, b->fm_hist[0], b->fm_hist[1],
                  b->osc_dev2[0], b->osc_dev2[1], b->chd.fm_state,
                  b->chd.lastaudio, b->chd.sq_count, b->chd.ahist, b->chd.foffset, b->chd.pdev, b->chd.gain, b->chd.hang,
                  b->chd.dc, b->chd.n0, b->chd.plresp, b->chd.plring, b->chd.pl_ptr, b->chd.pl_last, b->chd.plfreq,
                  b->pl2[0].plout, b->pl2[1].plout, b->pl.audio, b->pl.status, b->pl2[0].filt, b->pl2[0].n0raw, b->pl2[0].if_power,
                  b->pl2[1].filt, b->pl2[1].n0raw, b->pl2[1].if_power, b->energy_state, b->win_paired,
                  b->big.sync, b->big.n0part, b->big.xs,
                  b->list_dev[0], b->list_dev[1], b->list_dev[2], b->list_active_dev, b->list_active_ds_dev, b->chd.fflags, b->list_unswept_dev, b->list_swept_dev, b->spec_dump, b->stage_dev, b->pcm, b->pcm_mask, b->list_pll_dev, b->pll_slot_dev, b->pll_chunks_dev,
                  b->dq.scratch, b->dq.ng_next};
  for (kq::PllChunk const &ck : b->pll_chunks) {
    if (ck.state) (void)hipFree(ck.state);
    if (ck.rings) (void)hipFree(ck.rings);
    if (ck.side) (void)hipFree(ck.side);
  }
  b->pll_chunks.clear();
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  for (auto *v : {&b->ev_filter, &b->ev_demod, &b->ev_ingest})
    for (EventPair &p : *v) {
      (void)hipEventDestroy(p.a);
      (void)hipEventDestroy(p.b);
    }
  if (b->big.err) (void)hipHostFree(b->big.err);
  for (kq_bank::CtlQueue &q : b->ctl)
    for (int k = 0; k < kq_bank::CtlQueue::kDepth; k++) {
      if (q.buf[k]) (void)hipHostFree(q.buf[k]);
      if (q.applied[k]) (void)hipEventDestroy(q.applied[k]);
    }
  for (int k = 0; k < kq_bank::DesignQueue::kDepth; k++) {
    if (b->dq.pin[k]) (void)hipHostFree(b->dq.pin[k]);
    if (b->dq.read[k]) (void)hipEventDestroy(b->dq.read[k]);
  }
  for (int k = 0; k < 2; k++)
    if (b->dq.ng_moved[k]) (void)hipEventDestroy(b->dq.ng_moved[k]);
  for (hipStream_t st : {b->copy_in, b->copy_out})
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  for (int k = 0; k < 2; k++) {
    if (b->in_stage[k]) (void)hipFree(b->in_stage[k]);
    if (b->in_ready[k]) (void)hipEventDestroy(b->in_ready[k]);
    if (b->in_free[k]) (void)hipEventDestroy(b->in_free[k]);
  }
  if (b->out_ready) (void)hipEventDestroy(b->out_ready);
  for (int k = 0; k < 2; k++) {
    if (b->acc_pin[k]) (void)hipHostFree(b->acc_pin[k]);
    if (b->acc_read[k]) (void)hipEventDestroy(b->acc_read[k]);
  }
  for (hipEvent_t e : b->pull_done)
    if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < kq_bank::kSlots; k++) {
    if (b->stage_host[k]) (void)hipHostFree(b->stage_host[k]);
    if (b->stage_ev[k]) (void)hipEventDestroy(b->stage_ev[k]);
    if (b->stage_t0[k]) (void)hipEventDestroy(b->stage_t0[k]);
  }
  for (int k = 0; k < 2; k++)
    if (b->ev_demod_done[k]) (void)hipEventDestroy(b->ev_demod_done[k]);
  if (b->stream2 && b->stream2 != b->stream) (void)hipStreamDestroy(b->stream2);
  if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return 0;
}

namespace {
static bool is_pll(const kq_channel_config &c) { return c.demod_type == KQ_LINEAR_DEMOD && c.pll; }

// checks before a channel becomes a PLL channel (linear.c:51-56: the carrier search window is +-300 Hz, x2 when squaring,
// in bins of the 65536-point transform)
static int pll_admit(kq_bank *b, const kq_channel_config &m) {
  float const samptime = (float)b->g.D / (float)b->g.samprate;
  float const binsize = (float)(1. / (65536 * samptime));
  int const nbins = 2 * (int)round((m.square ? 2 : 1) * 300.f / binsize) + 1;
  if (nbins > 4096) {
    set_err("output rate too low for the PLL search window (%d bins > 4096)", nbins);
    return -1;
  }
  return 0;
}

// A slot for channel c's carrier loop, started afresh (linear.c:97-112): state and ring are zeroed by fill records of the
// demodulator side's queue -- applied in front of the next call's demodulators, behind the ones in flight, which may
// still be running the slot's previous owner.  Nothing here waits for the device; a chunk of slots is allocated when the
// free list runs out (33 MiB per 64 channels).
static int pll_acquire(kq_bank *b, int c) {
  size_t const Cmax = b->cfg.max_channels;
  if (!b->list_pll_dev) {
    if (dev_alloc(&b->list_pll_dev, Cmax) || dev_alloc(&b->pll_slot_dev, Cmax) ||
        dev_alloc(&b->pll_chunks_dev, (size_t)kq_bank::kMaxPllChunks))
      return -1;
  }
  if (b->pll_free.empty()) {
    if ((int)b->pll_chunks.size() >= kq_bank::kMaxPllChunks) {
      set_err("at most %d carrier-tracking (pll) channels per bank", kq_bank::kMaxPllChunks * kq::kPllChunk);
      return -1;
    }
    kq::PllChunk ck{};
    if (dev_alloc(&ck.state, (size_t)kq::kPllChunk) || dev_alloc(&ck.rings, (size_t)kq::kPllChunk * 65536) ||
        dev_alloc(&ck.side, (size_t)kq::kPllChunk * 4096)) {
      if (ck.state) (void)hipFree(ck.state);
      if (ck.rings) (void)hipFree(ck.rings);
      if (ck.side) (void)hipFree(ck.side);
      return -1;
    }
    int const k = (int)b->pll_chunks.size();
    b->pll_chunks.push_back(ck);
    if (ctl_put(b, CTL_DEMOD, b->pll_chunks_dev + k, &ck, sizeof ck)) return -1;
    for (int s = kq::kPllChunk - 1; s >= 0; s--) b->pll_free.push_back(k * kq::kPllChunk + s);
  }
  int const slot = b->pll_free.back();
  b->pll_free.pop_back();
  kq::PllChunk const &ck = b->pll_chunks[slot / kq::kPllChunk];
  int const sl = slot % kq::kPllChunk;
  if (ctl_fill(b, CTL_DEMOD, ck.state + sl, 0u, sizeof(kq::PllState)) ||
      ctl_fill(b, CTL_DEMOD, ck.rings + (size_t)sl * 65536, 0u, sizeof(float2) * 65536) ||
      ctl_put(b, CTL_DEMOD, b->pll_slot_dev + c, &slot, sizeof(int))) {
    b->pll_free.push_back(slot);
    return -1;
  }
  b->chans[c].pll_slot = slot;
  return 0;
}
static void pll_release(kq_bank *b, int c) {
  int &slot = b->chans[c].pll_slot;
  if (slot >= 0) b->pll_free.push_back(slot);
  slot = -1;
}
}  // namespace

int kq_bank_add_channel(kq_bank *b, const kq_channel_config *cfg) {
  BankScope dev_scope_(b);
  if (!b || !cfg) {
    set_err("NULL argument");
    return -1;
  }
  // the lowest hole a removed channel left, else a new slot at the end
  int c = (int)b->chans.size();
  for (int k = 0; k < (int)b->chans.size(); k++)
    if (!b->chans[k].active) {
      c = k;
      break;
    }
  if ((size_t)c >= b->cfg.max_channels) {
    set_err("bank is full (%u channels)", b->cfg.max_channels);
    return -1;
  }
  if (cfg->demod_type < KQ_LINEAR_DEMOD || cfg->demod_type > KQ_FM_DEMOD) {
    set_err("unknown demod_type %d", cfg->demod_type);
    return -1;
  }
  if (std::isnan(cfg->low) || std::isnan(cfg->high)) {  // filter.c:504-505
    set_err("NaN filter edge");
    return -1;
  }
  if (cfg->demod_type == KQ_FM_DEMOD && !kq::demod64_supported(b->g) && kq::demod_fm_lds_bytes(b->g) > 160 * 1024) {
    // the FM kernels keep one block of samples / the N/D-point audio master in LDS (N/D <= 8192)
    set_err("FM working set of %zu bytes exceeds the 160 KiB of LDS at this geometry", kq::demod_fm_lds_bytes(b->g));
    return -1;
  }
  if (is_pll(*cfg) && pll_admit(b, *cfg)) return -1;
  HostChan h;
  h.cfg = *cfg;
  h.out_type = (cfg->demod_type == KQ_LINEAR_DEMOD && cfg->isb) ? kq::FT_CROSS_CONJ : kq::FT_COMPLEX;
  double const fs = b->g.samprate;
  // oscillator setter scalings: radio.c:299, radio.c:182, radio.c:309
  h.lo2.set(cfg->second_lo == 0 ? 0.0 : cfg->second_lo / fs, 0.0, b->n_abs);
  h.dop.set(-cfg->doppler / fs, -cfg->doppler_rate / (fs * fs), b->n_abs);
  h.shift.set(cfg->shift == 0 ? 0.0 : cfg->shift * b->g.D / fs, 0.0, b->out_abs);
  bool const appended = c == (int)b->chans.size();
  if (appended)
    b->chans.push_back(h);
  else
    b->chans[c] = h;
  if ((is_pll(*cfg) && pll_acquire(b, c)) || upload_channel(b, c) || queue_design(b, c)) {
    pll_release(b, c);  // (the slot it may have been given)
    release_n0slot(b, b->chans[c].n0slot);  // (the mask set it may have been given)
    b->chans[c].n0slot = -1;
    if (appended)
      b->chans.pop_back();
    else
      b->chans[c].active = false;
    return -1;
  }
  if (lists_add(b, c)) return -1;
  b->chan_tw_dirty = true;
  // a channel more leaves the steady state of the others alone: its planes are patched in by the next call
  b->chans[c].r_eff = 0;
  b->n_active++;
  b->cache_any = true;
  b->sweep_lists_dirty = true;
  note_patch(b, c);
  return c;
}

// Many channels at once (a receiver bank of tens of thousands of channels is set up in one go): the same result as
// kq_bank_add_channel called n times, but every distinct response is designed once (one launch for all of them), every
// distinct compute_n0 mask is built once, and each per-channel plane is uploaded with one copy instead of n.
int kq_bank_add_channels(kq_bank *b, const kq_channel_config *cfgs, unsigned n, int *indices) {
  BankScope dev_scope_(b);
  if (!b || (!cfgs && n)) {
    set_err("NULL argument");
    return -1;
  }
  if (n == 0) return 0;
  bool holes = false, any_pll = false, any_nan = false;
  for (HostChan const &h : b->chans) holes = holes || !h.active;
  for (unsigned i = 0; i < n; i++) any_pll = any_pll || is_pll(cfgs[i]);
  // a NaN beta is not a key the ordered maps below can hold (it breaks their strict weak ordering): such a batch takes
  // the one-by-one path, whose design takes the value as the reference's does (queue_design)
  for (unsigned i = 0; i < n; i++) any_nan = any_nan || std::isnan(cfgs[i].kaiser_beta);
  if (holes || any_pll || any_nan || n < 4) {  // slot reuse and carrier-loop slots: one by one; all or nothing
    std::vector<int> got;
    for (unsigned i = 0; i < n; i++) {
      int const c = kq_bank_add_channel(b, &cfgs[i]);
      if (c < 0) {
        std::string const why = g_err;
        for (size_t k = got.size(); k-- > 0;) (void)kq_bank_remove_channel(b, got[k]);
        g_err = why;
        return -1;
      }
      got.push_back(c);
      if (indices) indices[i] = c;
    }
    return (int)n;
  }
  kq::Geom const &g = b->g;
  size_t const c0 = b->chans.size();
  if (c0 + n > b->cfg.max_channels) {
    set_err("bank is full (%u channels): %zu present, %u more asked for", b->cfg.max_channels, c0, n);
    return -1;
  }
  for (unsigned i = 0; i < n; i++) {
    kq_channel_config const &k = cfgs[i];
    if (k.demod_type < KQ_LINEAR_DEMOD || k.demod_type > KQ_FM_DEMOD) {
      set_err("unknown demod_type %d (entry %u)", k.demod_type, i);
      return -1;
    }
    if (std::isnan(k.low) || std::isnan(k.high)) {  // filter.c:504-505
      set_err("NaN filter edge (entry %u)", i);
      return -1;
    }
    if (k.demod_type == KQ_FM_DEMOD && !kq::demod64_supported(g) && kq::demod_fm_lds_bytes(g) > 160 * 1024) {
      set_err("FM working set of %zu bytes exceeds the 160 KiB of LDS at this geometry", kq::demod_fm_lds_bytes(g));
      return -1;
    }
  }
  // responses: every distinct (out_type, edges, beta) once, one launch per out_type
  std::vector<HostChan> hs(n);
  double const fs = g.samprate;
  struct Key {
    float lo, hi, beta;
    bool operator<(Key const &o) const { return lo != o.lo ? lo < o.lo : hi != o.hi ? hi < o.hi : beta < o.beta; }
  };
  for (int ot : {(int)kq::FT_COMPLEX, (int)kq::FT_CROSS_CONJ}) {
    std::map<Key, int> job;
    std::vector<kq::BandEdges> edges;
    std::vector<int> which(n, -1);
    for (unsigned i = 0; i < n; i++) {
      kq_channel_config const &k = cfgs[i];
      int const out_type = (k.demod_type == KQ_LINEAR_DEMOD && k.isb) ? kq::FT_CROSS_CONJ : kq::FT_COMPLEX;
      if (out_type != ot) continue;
      float lo_n, hi_n;  // as queue_design: fm.c:35 divides by the output rate, am.c:41 / linear.c:81 multiply by samptime
      if (k.demod_type == KQ_FM_DEMOD) {
        lo_n = k.low / g.dsamprate;
        hi_n = k.high / g.dsamprate;
      } else {
        float const samptime = (float)g.D / (float)g.samprate;
        lo_n = samptime * k.low;
        hi_n = samptime * k.high;
      }
      Key const key{lo_n, hi_n, k.kaiser_beta};
      auto it = job.find(key);
      if (it == job.end()) {
        it = job.emplace(key, (int)edges.size()).first;
        edges.push_back(kq::BandEdges{lo_n, hi_n, k.kaiser_beta});
      }
      which[i] = it->second;
    }
    if (edges.empty()) continue;
    std::vector<kq::cfloat> resp;
    std::vector<float> ng;
    if (kq::design_responses(g.N, g.olen, g.Mdec, ot, edges, resp, ng) || resp.size() != edges.size() * (size_t)g.Ndec) {
      set_err("response design failed");
      return -1;
    }
    for (unsigned i = 0; i < n; i++)
      if (which[i] >= 0) {
        hs[i].out_type = ot;
        hs[i].resp.assign(resp.begin() + (size_t)which[i] * g.Ndec, resp.begin() + (size_t)(which[i] + 1) * g.Ndec);
        hs[i].noise_gain = ng[which[i]];
      }
  }
  std::map<float, std::vector<kq::cfloat>> aresp_by_beta;  // fm.c:54-66 depends on the geometry and beta only
  for (unsigned i = 0; i < n; i++) {
    kq_channel_config const &k = cfgs[i];
    HostChan &h = hs[i];
    h.cfg = k;
    // oscillator setter scalings: radio.c:299, radio.c:182, radio.c:309
    h.lo2.set(k.second_lo == 0 ? 0.0 : k.second_lo / fs, 0.0, b->n_abs);
    h.dop.set(-k.doppler / fs, -k.doppler_rate / (fs * fs), b->n_abs);
    h.shift.set(k.shift == 0 ? 0.0 : k.shift * g.D / fs, 0.0, b->out_abs);
    if (k.demod_type == KQ_FM_DEMOD && !k.flat) {
      auto it = aresp_by_beta.find(k.kaiser_beta);
      if (it == aresp_by_beta.end()) {
        std::vector<kq::cfloat> a = kq::design_fm_audio_response(g.olen, g.Mdec, g.dsamprate, k.kaiser_beta);
        if (a.empty()) {
          set_err("FM audio response design failed");
          return -1;
        }
        it = aresp_by_beta.emplace(k.kaiser_beta, std::move(a)).first;
      }
      h.aresp = it->second;
    }
  }
  if (ctl_flush_now(b) || sync_all(b)) return -1;  // (what the control plane has queued goes first: the copies below are immediate)
  // per-channel planes of the new range, one copy each
  auto put = [&](auto *dst, auto const &v) -> int {
    HIP_TRY(hipMemcpy(dst + c0, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    return 0;
  };
  {
    std::vector<int> mode(n), flags(n), hangmax(n);
    std::vector<float> low(n), high(n), fm_gain(n), headroom(n), recovery(n), gain(n), ngain(n), nanv(n, NAN);
    std::vector<float2> one(n, make_float2(1.f, 0.f));  // fm.c:26
    for (unsigned i = 0; i < n; i++) {
      Derived const d = derive(g, cfgs[i]);
      mode[i] = d.mode;
      flags[i] = d.flags;
      hangmax[i] = d.hangmax;
      low[i] = cfgs[i].low;
      high[i] = cfgs[i].high;
      fm_gain[i] = d.fm_gain;
      headroom[i] = cfgs[i].headroom;
      recovery[i] = d.recovery;
      gain[i] = d.init_gain;
      ngain[i] = hs[i].noise_gain;
    }
    if (put(b->chd.mode, mode) || put(b->chd.flags, flags) || put(b->chd.fflags, flags) || put(b->chd.hangmax, hangmax) || put(b->chd.low, low) ||
        put(b->chd.high, high) || put(b->chd.fm_gain, fm_gain) || put(b->chd.headroom, headroom) ||
        put(b->chd.recovery, recovery) || put(b->chd.gain, gain) || put(b->chd.noise_gain, ngain) || put(b->chd.n0, nanv) ||
        put(b->chd.plfreq, nanv) || put(b->chd.fm_state, one))
      return -1;
  }
  // demodulator state at its prologue values (fm.c:26,68-69; am.c:26,33; linear.c:33)
  HIP_TRY(hipMemset(b->chd.lastaudio + c0, 0, n * sizeof(float)));
  HIP_TRY(hipMemset(b->chd.sq_count + c0, 0, n * sizeof(int)));
  HIP_TRY(hipMemset(b->chd.hang + c0, 0, n * sizeof(int)));
  HIP_TRY(hipMemset(b->chd.dc + c0, 0, n * sizeof(float)));
  HIP_TRY(hipMemset(b->chd.foffset + c0, 0, n * sizeof(float)));
  HIP_TRY(hipMemset(b->chd.pdev + c0, 0, n * sizeof(float)));
  if (g.Mdec > 1) {
    size_t const w = (size_t)(g.Mdec - 1);
    HIP_TRY(hipMemset(b->chd.ahist + c0 * w, 0, n * w * sizeof(float)));
    for (int k = 0; k < 2; k++)
      if (b->fm_hist[k]) HIP_TRY(hipMemset(b->fm_hist[k] + c0 * w, 0, n * w * sizeof(float)));
  }
  if (g.pl_n > 0) {
    HIP_TRY(hipMemset(b->chd.plring + c0 * 16384, 0, (size_t)n * 16384 * sizeof(float)));
    HIP_TRY(hipMemset(b->chd.pl_ptr + c0, 0, n * sizeof(*b->chd.pl_ptr)));
    HIP_TRY(hipMemset(b->chd.pl_last + c0, 0, n * sizeof(*b->chd.pl_last)));
  }
  {  // responses
    std::vector<float2> resp((size_t)n * g.Ndec);
    for (unsigned i = 0; i < n; i++) memcpy(&resp[(size_t)i * g.Ndec], hs[i].resp.data(), sizeof(float2) * g.Ndec);
    HIP_TRY(hipMemcpy(b->chd.resp + c0 * g.Ndec, resp.data(), resp.size() * sizeof(float2), hipMemcpyHostToDevice));
    size_t const na = (size_t)g.Ndec / 2 + 1;
    std::vector<float2> ar((size_t)n * na, make_float2(0.f, 0.f));
    bool any = false;
    for (unsigned i = 0; i < n; i++)
      if (!hs[i].aresp.empty()) {
        memcpy(&ar[(size_t)i * na], hs[i].aresp.data(), sizeof(float2) * na);
        any = true;
      }
    if (any) HIP_TRY(hipMemcpy(b->chd.aresp + c0 * na, ar.data(), ar.size() * sizeof(float2), hipMemcpyHostToDevice));
  }
  if (b->chd.n0lane) {  // compute_n0's lane masks: one set per distinct pair of edges, shared
    int const nsub = b->use64k ? 4 : 1;
    std::vector<int> slots(n);
    unsigned taken = 0;
    auto fill = [&]() -> int {
      for (unsigned i = 0; i < n; i++) {
        bool fresh = false;
        int const slot = acquire_n0slot(b, cfgs[i].low, cfgs[i].high, &fresh);
        hs[i].n0slot = slots[i] = slot;
        if (slot < 0) return -1;
        taken = i + 1;
        if (fresh) {
          std::vector<unsigned long long> m;
          std::vector<unsigned> meta;
          build_n0mask(b, cfgs[i].low, cfgs[i].high, m, meta);
          HIP_TRY(hipMemcpy(b->chd.n0lane + (size_t)slot * nsub * 256, m.data(), m.size() * sizeof(m[0]), hipMemcpyHostToDevice));
          HIP_TRY(hipMemcpy(b->chd.n0meta + (size_t)slot * nsub, meta.data(), meta.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        }
      }
      HIP_TRY(hipMemcpy(b->chd.n0slot + c0, slots.data(), n * sizeof(int), hipMemcpyHostToDevice));
      return 0;
    };
    if (fill()) {  // all or nothing: the references taken so far go back
      for (unsigned i = 0; i < taken; i++) release_n0slot(b, slots[i]);
      return -1;
    }
  }
  for (unsigned i = 0; i < n; i++) {
    b->chans.push_back(std::move(hs[i]));
    if (indices) indices[i] = (int)(c0 + i);
  }
  b->lists_dirty = true;
  b->osc_dirty = true;
  b->chan_tw_dirty = true;
  return (int)n;
}

// close_chan equivalent: the demodulator thread is joined and its struct demod freed (radio.c:335-337 does the join
// for a mode change).  Channel numbers of the others do not change; the slot is a hole until an add reuses it.
int kq_bank_remove_channel(kq_bank *b, int ch) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch)) {
    set_err("bad channel");
    return -1;
  }
  HostChan &h = b->chans[ch];
  pll_release(b, ch);  // (a carrier loop's slot goes back on the free list: its next owner starts it afresh)
  // (nothing on the device changes: the calls in flight still carry the channel, the next call's lists do not)
  if (lists_remove(b, ch)) return -1;
  h.active = false;
  h.retuned = false;
  h.hist_old = 0;
  for (int64_t &n : h.hist_oldx) n = 0;
  h.patched = false;  // (its entry on the patch list, if any, is skipped: the next call stages the whole bank)
  release_n0slot(b, h.n0slot);
  h.n0slot = -1;
  // (the others' steady state is untouched: the launch decisions' counters lose this channel, the lists are redone)
  b->n_active--;
  b->n_swept -= h.r_eff != 0;
  b->n_fast -= std::fabs(h.r_eff) > (b->use64k ? kq::full64k_sweep_limit() : kq::full16k_sweep_limit());
  b->cache_any = b->n_active > 0;
  b->sweep_lists_dirty = true;
  h.r_eff = 0;
  h.lo2 = h.dop = h.shift = h.lo2_old = h.dop_old = Osc{};
  for (int l = 0; l < kq::kOldLevels; l++) h.lo2_oldx[l] = h.dop_oldx[l] = Osc{};
  h.out_rtp = kq_out_rtp_state{};
  while (!b->chans.empty() && !b->chans.back().active) b->chans.pop_back();  // holes at the end just go
  return 0;
}

int kq_bank_channel_active(const kq_bank *b, int ch) {
  if (!b) return 0;
  std::lock_guard<std::recursive_mutex> lk(const_cast<kq_bank *>(b)->mu);
  return valid_ch(b, ch) ? 1 : 0;
}

unsigned kq_bank_num_channels(const kq_bank *b) {
  if (!b) return 0;
  std::lock_guard<std::recursive_mutex> lk(const_cast<kq_bank *>(b)->mu);
  return (unsigned)b->chans.size();
}

int kq_bank_set_mode(kq_bank *b, int ch, const kq_channel_config *m) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !m) {
    set_err("bad channel or NULL mode");
    return -1;
  }
  if (m->demod_type < KQ_LINEAR_DEMOD || m->demod_type > KQ_FM_DEMOD) {
    set_err("unknown demod_type %d", m->demod_type);
    return -1;
  }
  if (std::isnan(m->low) || std::isnan(m->high)) {
    set_err("NaN filter edge");
    return -1;
  }
  if (m->demod_type == KQ_FM_DEMOD && !kq::demod64_supported(b->g) && kq::demod_fm_lds_bytes(b->g) > 160 * 1024) {
    set_err("FM working set exceeds the LDS at this geometry");
    return -1;
  }
  // pthread_join of the old demodulator thread (radio.c:335-337): the new state is written on the main stream behind the
  // last call's demodulators (upload_channel); only carrier-loop slots, moved by synchronous copies, need the device idle
  HostChan &h = b->chans[ch];
  bool const was = is_pll(h.cfg), now = is_pll(*m);
  if (now && pll_admit(b, *m)) return -1;
  {  // a fresh loop either way (linear.c:97-112).  The new slot is taken BEFORE the old one goes back: a failure (the
     // allocation of another chunk) then leaves the channel as it was, loop and all
    int const old_slot = h.pll_slot;
    if (now) {
      h.pll_slot = -1;
      if (pll_acquire(b, ch)) {
        h.pll_slot = old_slot;
        if (old_slot >= 0 && ctl_put(b, CTL_DEMOD, b->pll_slot_dev + ch, &old_slot, sizeof(int))) return -1;
        return -1;
      }
    }
    if (was && old_slot >= 0) {
      b->pll_free.push_back(old_slot);
      if (!now) h.pll_slot = -1;
    }
  }
  // the mode table entry (radio.c:341-363); the input oscillators are not touched
  h.cfg.demod_type = m->demod_type;
  h.cfg.low = m->low > m->high ? m->high : m->low;  // radio.c:343-349
  h.cfg.high = m->low > m->high ? m->low : m->high;
  h.cfg.flat = m->flat;
  h.cfg.isb = m->isb;
  h.cfg.channels = m->channels;
  h.cfg.pll = m->pll;
  h.cfg.square = m->square;
  h.cfg.recovery_rate = m->recovery_rate;
  h.cfg.hangtime = m->hangtime;
  h.cfg.kaiser_beta = m->kaiser_beta;
  h.cfg.headroom = m->headroom;
  h.cfg.shift = m->shift;
  h.out_type = (m->demod_type == KQ_LINEAR_DEMOD && m->isb) ? kq::FT_CROSS_CONJ : kq::FT_COMPLEX;
  h.shift.set(m->shift == 0 ? 0.0 : m->shift * b->g.D / (double)b->g.samprate, 0.0, b->out_abs);  // radio.c:367
  if (upload_channel(b, ch, false) || queue_design(b, ch)) return -1;
  if (lists_retype(b, ch)) return -1;
  note_patch(b, ch);  // the shift oscillator (radio.c:367)
  return 0;
}

// linear.c:117-120 copies demod->filter.isb into the slave's out_type before every block, and linear.c:291-300 looks
// at demod->output.channels after it: both may change while the demodulator runs, without touching its AGC or the
// response (which keeps the gain it was designed with until the next set_filter, as in the reference).
int kq_bank_set_linear_options(kq_bank *b, int ch, int isb, int channels) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || (channels != 1 && channels != 2)) {
    set_err("bad channel, or channels not 1 or 2");
    return -1;
  }
  HostChan &h = b->chans[ch];
  if (h.cfg.demod_type != KQ_LINEAR_DEMOD) {
    set_err("not a linear channel");
    return -1;
  }
  if ((h.cfg.isb != 0) == (isb != 0) && h.cfg.channels == channels) return 0;
  h.cfg.isb = isb != 0;
  h.cfg.channels = channels;
  h.out_type = h.cfg.isb ? kq::FT_CROSS_CONJ : kq::FT_COMPLEX;
  int flags = 0;
  if (h.cfg.flat) flags |= kq::FLAG_FLAT;
  if (h.cfg.isb) flags |= kq::FLAG_ISB;
  if (h.cfg.channels == 2) flags |= kq::FLAG_STEREO;
  if (h.cfg.square) flags |= kq::FLAG_SQUARE;
  // (filter.out->out_type for the slave, demod->output.channels for the hand-off: each side from its next block on)
  if (ctl_put(b, CTL_FILTER, b->chd.fflags + ch, &flags, sizeof(int))) return -1;
  if (ctl_put(b, CTL_DEMOD, b->chd.flags + ch, &flags, sizeof(int))) return -1;
  return 0;
}

int kq_bank_set_second_lo(kq_bank *b, int ch, double hz) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || std::isnan(hz)) {
    set_err("bad channel or NaN");
    return -1;
  }
  note_retune(b, ch);
  b->chans[ch].cfg.second_lo = hz;
  b->chans[ch].lo2.set(hz == 0 ? 0.0 : hz / b->g.samprate, 0.0, b->n_abs);
  b->chan_tw_dirty = true;
  note_patch(b, ch);
  return 0;
}

int kq_bank_set_doppler(kq_bank *b, int ch, double hz, double hz_per_s) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || std::isnan(hz) || std::isnan(hz_per_s)) {
    set_err("bad channel or NaN");
    return -1;
  }
  double const fs = b->g.samprate;
  note_retune(b, ch);
  b->chans[ch].cfg.doppler = hz;
  b->chans[ch].cfg.doppler_rate = hz_per_s;
  b->chans[ch].dop.set(-hz / fs, -hz_per_s / (fs * fs), b->n_abs);
  b->chan_tw_dirty = true;
  note_patch(b, ch);
  return 0;
}

int kq_bank_set_shift(kq_bank *b, int ch, double hz) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || std::isnan(hz)) {
    set_err("bad channel or NaN");
    return -1;
  }
  b->chans[ch].cfg.shift = hz;
  b->chans[ch].shift.set(hz == 0 ? 0.0 : hz * b->g.D / (double)b->g.samprate, 0.0, b->out_abs);
  note_patch(b, ch);
  return 0;
}

int kq_bank_set_n0(kq_bank *b, int ch, float n0) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch)) {
    set_err("bad channel");
    return -1;
  }
  // the demodulators of a call in flight own the state: written behind them, in front of the next call's
  if (ctl_put(b, CTL_DEMOD, b->chd.n0 + ch, &n0, sizeof n0)) return -1;
  return 0;
}

int kq_bank_set_filter(kq_bank *b, int ch, float low, float high, float beta) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch)) {
    set_err("bad channel");
    return -1;
  }
  if (std::isnan(low) || std::isnan(high)) {  // filter.c:504-505
    set_err("NaN filter edge");
    return -1;
  }
  HostChan &h = b->chans[ch];
  h.cfg.low = low;
  h.cfg.high = high;
  h.cfg.kaiser_beta = beta;
  float const fm_gain = (float)((h.cfg.headroom * M_1_PI * b->g.dsamprate) / fabsf(low - high));
  // the new response takes effect from the next call on (filter.c:538-543 swaps it under the mutex between two blocks):
  // queued for that call; the host does not wait
  if (ctl_put(b, CTL_FILTER, b->chd.low + ch, &low, sizeof(float))) return -1;
  if (ctl_put(b, CTL_FILTER, b->chd.high + ch, &high, sizeof(float))) return -1;
  if (ctl_put(b, CTL_DEMOD, b->chd.fm_gain + ch, &fm_gain, sizeof(float))) return -1;
  if (upload_n0mask(b, ch)) return -1;
  return queue_design(b, ch, true);
}

namespace {
static int acc_flush(kq_bank *b);
static int acc_append(kq_bank *b, const void *src, size_t nsamples, int format);
}  // namespace

int kq_bank_push_iq(kq_bank *b, const void *iq, size_t nsamples, int format, int is_device) {
  BankScope dev_scope_(b);
  if (!b || (!iq && nsamples)) {
    set_err("NULL argument");
    return -1;
  }
  if (acc_flush(b)) return -1;  // packet payloads gathered by kq_bank_push_rtp go first
  if (format < KQ_IQ_CF32 || format > KQ_IQ_S8) {
    set_err("unknown I/Q format %d", format);
    return -1;
  }
  kq::Geom const &g = b->g;
  size_t const used = (size_t)(g.M - 1) + b->pending;
  if (used + nsamples > b->ring_cap) {
    set_err("ring overflow: %zu pending + %zu pushed > %zu", b->pending, nsamples, b->ring_cap - (g.M - 1));
    return -1;
  }
  size_t const bps = format == KQ_IQ_CF32 ? 8 : format == KQ_IQ_S16 ? 4 : 2;
  const void *src = iq;
  if (!is_device) {
    if (b->stage_cap < nsamples * bps) {
      if (b->stage_dev) (void)hipFree(b->stage_dev);
      b->stage_cap = nsamples * bps;
      HIP_TRY(hipMalloc(&b->stage_dev, b->stage_cap));
    }
    HIP_TRY(hipMemcpyAsync(b->stage_dev, iq, nsamples * bps, hipMemcpyHostToDevice, b->stream));
    src = b->stage_dev;
  }
  {
    Scope t(b, 2, b->stream);
    kq::launch_ingest(b->stream, src, format, b->ring[b->cur] + used, nsamples, b->cfg.gain_factor);
  }
  if (!is_device) HIP_TRY(hipStreamSynchronize(b->stream));  // the caller may reuse iq
  // block completion bookkeeping for the IF-power rule
  size_t fill = b->pending % g.L;
  size_t left = nsamples;
  while (left) {
    size_t const take = std::min(left, (size_t)g.L - fill);
    fill += take;
    left -= take;
    if (fill == (size_t)g.L) {
      b->zero_tail.push_back(0);
      fill = 0;
    }
  }
  b->pending += nsamples;
  return 0;
}

namespace {
static int host_io_setup(kq_bank *b) {
  if (b->copy_in) return 0;
  // Streams share a handful of hardware queues, handed out in creation order, and a queue runs in order: the two copy
  // streams can land on one queue.  Then an input copy queued BEHIND an output copy waits with it for that call's
  // demodulators, and every step runs input copy, kernels and output copy one after the other (rocprofv3 timeline,
  // tools/hostio_trace.sh) -- hence the call order the header asks for: push batch k+1 before pulling the planes of
  // batch k.  (Streams of different priority come from different queue pools, but with a high-priority output stream
  // the filter kernel itself ran 40 % slower for the whole step, measured.)
  HIP_TRY(hipStreamCreateWithFlags(&b->copy_in, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&b->copy_out, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) {
    HIP_TRY(hipEventCreateWithFlags(&b->in_ready[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&b->in_free[k], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreateWithFlags(&b->out_ready, hipEventDisableTiming));
  for (int k = 0; k < kq_bank::kPullRing; k++) HIP_TRY(hipEventCreateWithFlags(&b->pull_done[k], hipEventDisableTiming));
  return 0;
}
// block completion bookkeeping for the IF-power rule (radio.c:140-146 against radio.c:94-98)
static void note_pushed(kq_bank *b, size_t nsamples, unsigned char zero) {
  size_t fill = b->pending % b->g.L, left = nsamples;
  while (left) {
    size_t const take = std::min(left, (size_t)b->g.L - fill);
    fill += take;
    left -= take;
    if (fill == (size_t)b->g.L) {
      b->zero_tail.push_back(zero);
      fill = 0;
    }
  }
  b->pending += nsamples;
}
}  // namespace

namespace {
// `nsamples` samples of `format` in pinned host memory -> staging buffer (copy stream) -> conversion kernel (bank's stream)
// into the ring at sample offset `ring_off`.  Nothing waits on the host but the reuse of a staging buffer two copies later.
static int queue_input_copy(kq_bank *b, const void *iq, size_t nsamples, int format, size_t ring_off) {
  if (host_io_setup(b)) return -1;
  size_t const bps = format == KQ_IQ_CF32 ? 8 : format == KQ_IQ_S16 ? 4 : 2;
  int const k = b->in_next;
  b->in_next ^= 1;
  if (b->in_stage_cap[k] < nsamples * bps) {
    HIP_TRY(hipStreamSynchronize(b->copy_in));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->in_stage[k]) (void)hipFree(b->in_stage[k]);
    b->in_stage_cap[k] = nsamples * bps;
    HIP_TRY(hipMalloc(&b->in_stage[k], b->in_stage_cap[k]));
  }
  // The header's promise -- `iq` may be reused once two more pushes have been queued -- is kept here: the push two
  // back used this staging index, and its host-to-device copy must have READ the caller's buffer before this call
  // returns (a device-side wait alone would let the host run ahead of the copy engine).  Normally long done.
  if (b->in_used[k]) HIP_TRY(hipEventSynchronize(b->in_ready[k]));
  b->in_used[k] = true;
  HIP_TRY(hipStreamWaitEvent(b->copy_in, b->in_free[k], 0));  // the conversion kernel that last read this buffer
  HIP_TRY(hipMemcpyAsync(b->in_stage[k], iq, nsamples * bps, hipMemcpyHostToDevice, b->copy_in));
  HIP_TRY(hipEventRecord(b->in_ready[k], b->copy_in));
  HIP_TRY(hipStreamWaitEvent(b->stream, b->in_ready[k], 0));
  {
    Scope t(b, 2, b->stream);
    kq::launch_ingest(b->stream, b->in_stage[k], format, b->ring[b->cur] + ring_off, nsamples, b->cfg.gain_factor);
  }
  HIP_TRY(hipEventRecord(b->in_free[k], b->stream));
  return 0;
}

// the gathered run of packet payloads goes out (see acc_pin)
static int acc_flush(kq_bank *b) {
  if (b->acc_n == 0) return 0;
  int const j = b->acc_cur;
  if (queue_input_copy(b, b->acc_pin[j], b->acc_n, b->acc_fmt, b->acc_ring_off)) return -1;
  HIP_TRY(hipEventRecord(b->acc_read[j], b->copy_in));
  b->acc_read_set[j] = true;
  b->acc_cur ^= 1;
  b->acc_n = 0;
  // the buffer gathered into next was handed to the copy engine two flushes ago
  if (b->acc_read_set[b->acc_cur]) HIP_TRY(hipEventSynchronize(b->acc_read[b->acc_cur]));
  return 0;
}

// one packet's payload (host memory, any alignment) joins the run; the bookkeeping of the ring moves at once
static int acc_append(kq_bank *b, const void *src, size_t nsamples, int format) {
  if (nsamples == 0) return 0;
  size_t const bps = format == KQ_IQ_S16 ? 4 : format == KQ_IQ_S8 ? 2 : 8;
  if (!b->acc_pin[0]) {
    b->acc_cap = b->ring_cap * 8;  // the whole ring in the widest format
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipHostMalloc((void **)&b->acc_pin[k], b->acc_cap, hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&b->acc_read[k], hipEventDisableTiming));
    }
  }
  if (b->acc_n && (format != b->acc_fmt || (b->acc_n + nsamples) * bps > b->acc_cap) && acc_flush(b)) return -1;
  if (b->acc_n == 0) {
    b->acc_fmt = format;
    b->acc_ring_off = (size_t)(b->g.M - 1) + b->pending;
  }
  memcpy(b->acc_pin[b->acc_cur] + b->acc_n * bps, src, nsamples * bps);
  b->acc_n += nsamples;
  note_pushed(b, nsamples, 0);
  return 0;
}
}  // namespace

int kq_bank_push_iq_async(kq_bank *b, const void *iq, size_t nsamples, int format) {
  BankScope dev_scope_(b);
  if (!b || (!iq && nsamples)) {
    set_err("NULL argument");
    return -1;
  }
  if (format < KQ_IQ_CF32 || format > KQ_IQ_S8) {
    set_err("unknown I/Q format %d", format);
    return -1;
  }
  kq::Geom const &g = b->g;
  size_t const used = (size_t)(g.M - 1) + b->pending;
  if (used + nsamples > b->ring_cap) {
    set_err("ring overflow: %zu pending + %zu pushed > %zu", b->pending, nsamples, b->ring_cap - (g.M - 1));
    return -1;
  }
  if (nsamples == 0) return 0;
  if (acc_flush(b)) return -1;
  if (queue_input_copy(b, iq, nsamples, format, used)) return -1;
  note_pushed(b, nsamples, 0);
  return 0;
}

namespace {
// the copy stream gets behind the last call's demodulators: their own marker when they ran on their own stream, else one
// on the main stream
static int pull_prologue(kq_bank *b) {
  if (!b || b->calls == 0) {
    set_err("nothing processed yet");
    return -1;
  }
  if (host_io_setup(b)) return -1;
  int const last = (int)((b->calls - 1) & 1);
  if (b->demod_overlapped[last]) {
    HIP_TRY(hipStreamWaitEvent(b->copy_out, b->ev_demod_done[last], 0));
  } else {
    HIP_TRY(hipEventRecord(b->out_ready, b->stream));
    HIP_TRY(hipStreamWaitEvent(b->copy_out, b->out_ready, 0));
  }
  b->pulled_since_call = true;
  return 0;
}
static int pull_epilogue(kq_bank *b) {
  HIP_TRY(hipEventRecord(b->pull_done[b->pulls % kq_bank::kPullRing], b->copy_out));
  b->pulls++;
  b->out_pending = true;
  return 0;
}
}  // namespace

int kq_bank_pull_planes_async(kq_bank *b, float *audio, kq_chan_status *status) {
  BankScope dev_scope_(b);
  if (pull_prologue(b)) return -1;
  size_t const n = b->chans.size() * (size_t)b->g.max_blocks;
  // of every channel-block's 2 * olen floats only the status.nout that hold samples travel (mono: half): the kernel
  // moves 16 bytes per lane, so olen must be a multiple of 4 for it -- other geometries take the plain copy
  size_t const sbytes = n * sizeof(kq_chan_status), s16 = sbytes & ~(size_t)15;
  // (hipHostMalloc'ed memory is page aligned; anything less than 16 bytes takes the plain copies)
  bool const aligned = (reinterpret_cast<uintptr_t>(audio) & 15) == 0 && (reinterpret_cast<uintptr_t>(status) & 15) == 0;
  bool const rows_ok = b->g.olen % 4 == 0 && aligned;
  if (!aligned) {
    if (audio)
      HIP_TRY(hipMemcpyAsync(audio, b->pl.audio, n * 2 * (size_t)b->g.olen * sizeof(float), hipMemcpyDeviceToHost, b->copy_out));
    if (status) HIP_TRY(hipMemcpyAsync(status, b->pl.status, sbytes, hipMemcpyDeviceToHost, b->copy_out));
    return pull_epilogue(b);
  }
  kq::launch_copy_to_host(b->copy_out, b->pl.audio, rows_ok ? audio : nullptr, 2 * b->g.olen, b->pl.status, status, n);
  LAUNCH_CHECK("plane copy");
  if (audio && !rows_ok)
    HIP_TRY(hipMemcpyAsync(audio, b->pl.audio, n * 2 * (size_t)b->g.olen * sizeof(float), hipMemcpyDeviceToHost, b->copy_out));
  if (status && sbytes > s16)
    HIP_TRY(hipMemcpyAsync((char *)status + s16, (const char *)b->pl.status + s16, sbytes - s16, hipMemcpyDeviceToHost, b->copy_out));
  return pull_epilogue(b);
}

// The reference's real output format (audio.c:22-28, 45-50, 95-100): clipped int16 in network byte order, half the bytes
// of the float plane.  The conversion runs inside the copy kernel (the same arithmetic as k_pcm, the stage behind
// kq_bank_enable_pcm, which this call does not need).
static int pull_pcm_planes(kq_bank *b, int16_t *pcm, uint32_t *silent_mask, void *status, bool compact) {
  BankScope dev_scope_(b);
  if (!pcm) {
    set_err("NULL pcm plane");
    return -1;
  }
  bool const aligned = (reinterpret_cast<uintptr_t>(pcm) & 15) == 0 && (reinterpret_cast<uintptr_t>(status) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(silent_mask) & 3) == 0;
  if (b && (b->g.olen % 8 != 0 || !aligned || 2 * (size_t)b->g.olen > 32 * 480)) {
    set_err("kq_bank_pull_pcm_planes_async: olen must be a multiple of 8 and at most 7680, the planes 16-byte aligned");
    return -1;
  }
  if (pull_prologue(b)) return -1;
  size_t const n = b->chans.size() * (size_t)b->g.max_blocks;
  size_t const sbytes = n * sizeof(kq_chan_status), s16 = sbytes & ~(size_t)15;
  kq::launch_copy_pcm_to_host(b->copy_out, b->pl.audio, pcm, silent_mask, 2 * b->g.olen, b->pl.status, status, n,
                              compact && status ? b->chd.mode : nullptr, b->g.max_blocks);
  LAUNCH_CHECK("PCM plane copy");
  if (status && !compact && sbytes > s16)
    HIP_TRY(hipMemcpyAsync((char *)status + s16, (const char *)b->pl.status + s16, sbytes - s16, hipMemcpyDeviceToHost, b->copy_out));
  return pull_epilogue(b);
}
int kq_bank_pull_pcm_planes_async(kq_bank *b, int16_t *pcm, uint32_t *silent_mask, kq_chan_status *status) {
  return pull_pcm_planes(b, pcm, silent_mask, status, false);
}
// ... with the 24 bytes of every status record a receiver reads per block (include/ka9q_hip.h kq_chan_status_compact)
int kq_bank_pull_pcm_planes_compact_async(kq_bank *b, int16_t *pcm, uint32_t *silent_mask, kq_chan_status_compact *status) {
  return pull_pcm_planes(b, pcm, silent_mask, status, true);
}

int kq_bank_pull_wait(kq_bank *b, unsigned lag) {
  BankScope dev_scope_(b);
  if (!b) return -1;
  if (lag >= (unsigned)kq_bank::kPullRing) {
    set_err("kq_bank_pull_wait: lag %u, at most %d deliveries are remembered", lag, kq_bank::kPullRing - 1);
    return -1;
  }
  if (b->pulls <= lag) return 0;  // nothing that far back was ever queued
  hipEvent_t const ev = b->pull_done[(b->pulls - 1 - lag) % kq_bank::kPullRing];
  {
    Unlocked u(dev_scope_);  // (the ring holds kPullRing deliveries: the event is not recorded again before this one is long over)
    HIP_TRY(hipEventSynchronize(ev));
  }
  return report_lost_sibling(b);
}

int kq_bank_host_io_wait(kq_bank *b) {
  BankScope dev_scope_(b);
  if (!b) return -1;
  hipStream_t const cin = b->copy_in, cout = b->copy_out;  // (read under the lock: host_io_setup may be creating them)
  {
    Unlocked u(dev_scope_);
    if (cin) HIP_TRY(hipStreamSynchronize(cin));
    if (cout) HIP_TRY(hipStreamSynchronize(cout));
  }
  return report_lost_sibling(b);  // the planes just landed come from kernels that have finished
}

int kq_bank_push_zeros(kq_bank *b, size_t nsamples) {
  BankScope dev_scope_(b);
  if (!b) {
    set_err("NULL bank");
    return -1;
  }
  kq::Geom const &g = b->g;
  size_t const used = (size_t)(g.M - 1) + b->pending;
  if (used + nsamples > b->ring_cap) {
    set_err("ring overflow");
    return -1;
  }
  if (acc_flush(b)) return -1;
  HIP_TRY(hipMemsetAsync(b->ring[b->cur] + used, 0, nsamples * sizeof(float2), b->stream));
  size_t fill = b->pending % g.L;
  size_t left = nsamples;
  while (left) {
    size_t const take = std::min(left, (size_t)g.L - fill);
    fill += take;
    left -= take;
    if (fill == (size_t)g.L) {
      b->zero_tail.push_back(1);  // completed inside radio.c:88-99
      fill = 0;
    }
  }
  b->pending += nsamples;
  return 0;
}

static inline size_t g_M1(const kq_bank *b) { return (size_t)(b->g.M - 1); }

int kq_bank_push_rtp(kq_bank *b, const void *datagram, size_t size) {
  BankScope dev_scope_(b);
  if (!b || !datagram) {
    set_err("NULL argument");
    return -1;
  }
  const unsigned char *p = static_cast<const unsigned char *>(datagram);
  auto be16 = [](const unsigned char *q) { return (unsigned)((q[0] << 8) | q[1]); };
  auto be32 = [](const unsigned char *q) { return ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; };
  if (size < 12) return 0;  // RTP_MIN_SIZE, main.c:315-316
  // RTP header (multicast.c:242-277)
  bool const pad = (p[0] >> 5) & 1, ext = (p[0] >> 4) & 1;
  unsigned const cc = p[0] & 0xf;
  unsigned const type = p[1] & 0x7f;
  uint16_t const seq = (uint16_t)be16(p + 2);
  uint32_t const ts = be32(p + 4), ssrc = be32(p + 8);
  size_t hdr = 12 + 4 * (size_t)cc;
  if (ext) {
    if (hdr + 4 > size) return 0;
    hdr += 4 + 4 + be16(p + hdr + 2);  // type, length, and the reference's "4 + length" bytes
  }
  if (hdr > size) return 0;
  size_t len = size - hdr;
  if (pad && len > 0) {  // main.c:324-328
    unsigned const npad = p[size - 1];
    if (npad > len) return 0;
    len -= npad;
  }
  if (type != 97 && type != 98) return 0;  // IQ_PT / IQ_PT8, main.c:329-330
  if (len < 24) return 0;
  hdr += 24;  // obsolete status block, main.c:338-341
  len -= 24;
  int const sampcount = (int)(type == 97 ? len / 4 : len / 2);  // radio.c:64-72

  // proc_samples + rtp_process (radio.c:73-104, multicast.c:305-340)
  kq_rtp_counters &r = b->rtp;
  if (!b->rtp_init || ssrc != r.ssrc) {
    r.samples = 0;  // radio.c:73-77 (a fresh state has ssrc 0, so the first packet lands here as well)
    r.ssrc = ssrc;
    r.packets = 0;
    r.next_seq = seq;
    r.next_timestamp = ts;
    r.dupes = 0;
    r.drops = 0;
    b->rtp_init = true;
    b->rtp_retry = false;
  }
  // A datagram handed in again after -2 is the same packet, not a new one: it is counted once.
  bool const retry = b->rtp_retry && seq == b->rtp_retry_seq && ts == b->rtp_retry_ts;
  b->rtp_retry = false;
  short const seq_step = (short)(seq - r.next_seq);
  if (seq_step < 0) {
    r.packets++;
    r.dupes++;
    return 0;
  }
  int const time_step = (int)(ts - r.next_timestamp);
  if (time_step < 0 || time_step > 192000) {  // old samples (multicast.c:334-336) / a jump too far to fill (radio.c:79-82)
    if (!retry) r.packets++;
    r.drops += seq_step;
    r.next_seq = (uint16_t)(seq + 1);
    if (time_step >= 0) r.next_timestamp = ts + (uint32_t)sampcount;
    return 0;
  }
  // Room.  The ring takes max_blocks * L samples plus L - 1 of slack, so that a partly filled block never stands in the
  // way of a packet that fits the ring as such.  What does not fit now:
  //  * whole blocks are waiting (pending >= L): nothing moves, -2 -- kq_bank_process frees them, then the same datagram
  //    fits or falls under the next case;
  //  * no whole block is waiting, so the zero fill of the gap (radio.c:83-100) is itself larger than the ring: as many
  //    zeros as fit go in now and the timestamp moves past them, -2 -- every retry after a kq_bank_process brings the
  //    gap a ring closer to its end, with the oscillators running through it sample by sample as in the reference;
  //  * the gap is filled and the payload alone is larger than the ring (max_blocks * L below one packet): -1, the
  //    sequence number moves on, the timestamp does not, so the next packet fills these samples with zeros.
  size_t const room = b->ring_cap - (size_t)(g_M1(b)) - b->pending;
  size_t const payload = (size_t)(sampcount > 0 ? sampcount : 0);
  if ((size_t)time_step + payload > room) {
    auto remember = [&]() {
      b->rtp_retry = true;
      b->rtp_retry_seq = seq;
      b->rtp_retry_ts = ts;
    };
    if (b->pending >= (size_t)b->g.L) {  // nothing has moved: not the sequence number, not the timestamp, not a counter
      b->rtp_retry = retry;
      set_err("ring full: %zu samples pending, the packet brings %zu (zero fill %d): run kq_bank_process, then push it again",
              b->pending, (size_t)time_step + payload, time_step);
      return -2;
    }
    if (time_step > 0) {
      size_t const z = std::min((size_t)time_step, room);
      if (kq_bank_push_zeros(b, z)) return -1;
      if (!retry) r.packets++;
      r.samples += (int)z;
      r.next_timestamp += (uint32_t)z;
      r.drops += seq_step;
      r.next_seq = seq;  // the retry is in sequence
      remember();
      set_err("the gap's zero fill is larger than the ring: %zu of %d samples in, run kq_bank_process, then push it again", z,
              time_step);
      return -2;
    }
    if (!retry) r.packets++;
    r.drops += seq_step;
    r.next_seq = (uint16_t)(seq + 1);
    set_err("a packet of %zu samples does not fit a ring of %zu: raise max_blocks", payload, b->ring_cap - (size_t)g_M1(b));
    return -1;
  }
  if (!retry) r.packets++;
  r.drops += seq_step;
  r.next_seq = (uint16_t)(seq + 1);
  r.next_timestamp = ts + (uint32_t)sampcount;
  if (time_step > 0) {
    if (kq_bank_push_zeros(b, (size_t)time_step)) return -1;  // radio.c:83-100
    r.samples += time_step;
  }
  r.samples += sampcount;
  // the payload joins the run gathered in pinned memory: one asynchronous copy and one conversion per run, not per packet
  if (sampcount > 0 && acc_append(b, p + hdr, (size_t)sampcount, type == 97 ? KQ_IQ_S16 : KQ_IQ_S8)) return -1;
  return time_step + sampcount;
}

int kq_bank_rtp_counters(const kq_bank *b, kq_rtp_counters *out) {
  BankScope dev_scope_(b);
  if (!b || !out) return -1;
  *out = b->rtp;
  return 0;
}

unsigned kq_bank_blocks_ready(const kq_bank *b) {
  if (!b) return 0;
  std::lock_guard<std::recursive_mutex> lk(const_cast<kq_bank *>(b)->mu);
  return (unsigned)(b->pending / b->g.L);
}

// The back-pressure wait of a process call (the staging slot it is about to fill was last read four calls ago), taken
// BEFORE the call looks at any state and with the handle's lock let go: the operator's thread is not kept out for the
// length of a device wait (tools/soak_realtime.py showed 10 ms set_filter calls that were this wait).  Only the
// receiver thread advances stage_next, so the slot is still the next one when the lock is back.
int slot_prewait(BankScope &scope, kq_bank *b) {
  hipEvent_t const ev = b->stage_ev[b->stage_next];
  auto const t0 = std::chrono::steady_clock::now();
  {
    Unlocked u(scope);
    HIP_TRY(hipEventSynchronize(ev));
  }
  double const w = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  b->host_acc.slot_wait_ms += w;
  b->host_acc.call_ms += w;
  return 0;
}

int kq_bank_process(kq_bank *b) {
  BankScope dev_scope_(b, true);
  if (!b) {
    set_err("NULL bank");
    return -1;
  }
  if (slot_prewait(dev_scope_, b)) return -1;
  kq::Geom const &g = b->g;
  if (acc_flush(b)) return -1;  // what kq_bank_push_rtp has gathered goes into the ring now
  unsigned nb = (unsigned)(b->pending / g.L);
  if (nb > b->cfg.max_blocks) nb = b->cfg.max_blocks;
  if (nb == 0) return 0;
  std::vector<unsigned char> upd(nb);
  for (unsigned i = 0; i < nb; i++) upd[i] = b->zero_tail[i] ? 0 : 1;
  int const done = run_blocks(b, b->ring[b->cur], nb, upd.data());
  if (done < 0) return -1;
  b->zero_tail.erase(b->zero_tail.begin(), b->zero_tail.begin() + nb);
  // overlap-save history (filter.c:164) plus any unprocessed tail moves to the other ring buffer
  size_t const consumed = (size_t)nb * g.L;
  size_t const keep = (size_t)(g.M - 1) + (b->pending - consumed);
  HIP_TRY(hipMemcpyAsync(b->ring[b->cur ^ 1], b->ring[b->cur] + consumed, keep * sizeof(float2), hipMemcpyDeviceToDevice,
                         b->stream));
  b->cur ^= 1;
  b->pending -= consumed;
  return done;
}

int kq_bank_process_resident(kq_bank *b, const void *iq_dev, unsigned nblocks) {
  BankScope dev_scope_(b, true);
  if (!b || !iq_dev) {
    set_err("NULL argument");
    return -1;
  }
  if (nblocks == 0 || nblocks > b->cfg.max_blocks) {
    set_err("nblocks %u out of range 1..%u", nblocks, b->cfg.max_blocks);
    return -1;
  }
  if (slot_prewait(dev_scope_, b)) return -1;
  std::vector<unsigned char> upd(nblocks, 1);
  return run_blocks(b, (const float2 *)iq_dev, nblocks, upd.data());
}

int kq_bank_process_spectrum(kq_bank *b, const void *spectrum_dev, unsigned nblocks) {
  BankScope dev_scope_(b, true);
  if (!b || !spectrum_dev) {
    set_err("NULL argument");
    return -1;
  }
  if (nblocks == 0 || nblocks > b->cfg.max_blocks) {
    set_err("nblocks %u out of range 1..%u", nblocks, b->cfg.max_blocks);
    return -1;
  }
  for (HostChan const &h : b->chans) {
    if (!h.active) continue;
    // the spectrum is what it is: a channel that still has an oscillator to apply cannot be served from it
    if (h.lo2.set_f != 0 || h.dop.set_f != 0) {
      set_err("kq_bank_process_spectrum: a channel has a second LO or Doppler set; the mix belongs in front of the master");
      return -1;
    }
  }
  std::vector<unsigned char> upd(nblocks, 0);
  return run_blocks(b, nullptr, nblocks, upd.data(), (const float2 *)spectrum_dev);
}

void *kq_bank_stream(kq_bank *b) { return b ? (void *)b->stream : nullptr; }


int kq_bank_join(kq_bank *b) {
  BankScope dev_scope_(b);
  if (!b) return -1;
  if (b->calls == 0) return 0;
  int const last = (int)((b->calls - 1) & 1);
  if (b->demod_overlapped[last]) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_demod_done[last], 0));
  return 0;
}

int kq_bank_sync(kq_bank *b) {
  BankScope dev_scope_(b);
  if (!b) return -1;
  // "everything issued so far" includes what the control plane has queued for the next call: applied now
  if (ctl_flush_now(b)) return -1;
  hipStream_t const st[4] = {b->stream, b->stream2, b->copy_in, b->copy_out};  // (read under the lock)
  {
    Unlocked u(dev_scope_);
    for (hipStream_t x : st)
      if (x) HIP_TRY(hipStreamSynchronize(x));
  }
  return report_lost_sibling(b);
}

unsigned kq_bank_olen(const kq_bank *b) { return b ? (unsigned)b->g.olen : 0; }
unsigned kq_bank_last_blocks(const kq_bank *b) { return b ? b->last_blocks : 0; }

int kq_bank_pull_status(kq_bank *b, int ch, unsigned blk, kq_chan_status *st) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !st || blk >= b->last_blocks) {
    set_err("bad channel/block");
    return -1;
  }
  if (sync_all(b)) return -1;
  HIP_TRY(hipMemcpy(st, b->pl.status + (size_t)ch * b->g.max_blocks + blk, sizeof(*st), hipMemcpyDeviceToHost));
  return 0;
}

int kq_bank_pull_audio(kq_bank *b, int ch, unsigned blk, float *dst, size_t cap, size_t *n) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !dst || blk >= b->last_blocks) {
    set_err("bad channel/block");
    return -1;
  }
  kq_chan_status st;
  if (kq_bank_pull_status(b, ch, blk, &st)) return -1;
  if ((size_t)st.nout > cap) {
    set_err("audio buffer too small: %d > %zu", st.nout, cap);
    return -1;
  }
  HIP_TRY(hipMemcpy(dst, b->pl.audio + ((size_t)ch * b->g.max_blocks + blk) * 2 * (size_t)b->g.olen, st.nout * sizeof(float),
                    hipMemcpyDeviceToHost));
  if (n) *n = (size_t)st.nout;
  return 0;
}

int kq_bank_enable_pcm(kq_bank *b, int on) {
  BankScope dev_scope_(b);
  if (!b) return -1;
  if (on && 2 * (size_t)b->g.olen > 32 * 480) {
    // the silent-packet mask of kq_bank_pull_pcm is 32 bits: one per 480-word packet of a block
    set_err("PCM stage: %d output samples per block make more than 32 packets", b->g.olen);
    return -1;
  }
  if (on && !b->pcm) {
    size_t const CB = (size_t)b->cfg.max_channels * b->cfg.max_blocks;
    if (dev_alloc(&b->pcm, CB * 2 * (size_t)b->g.olen) || dev_alloc(&b->pcm_mask, CB)) return -1;
  }
  b->pcm_on = on != 0;
  return 0;
}

int kq_bank_pull_pcm(kq_bank *b, int ch, unsigned blk, int16_t *dst, size_t cap, size_t *nwords, uint32_t *silent_mask) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !dst || blk >= b->last_blocks || !b->pcm_on) {
    set_err("bad channel/block, or PCM stage not enabled");
    return -1;
  }
  kq_chan_status st;
  if (kq_bank_pull_status(b, ch, blk, &st)) return -1;
  if ((size_t)st.nout > cap) {
    set_err("PCM buffer too small");
    return -1;
  }
  size_t const cb = (size_t)ch * b->g.max_blocks + blk;
  HIP_TRY(hipMemcpy(dst, b->pcm + cb * 2 * (size_t)b->g.olen, st.nout * sizeof(int16_t), hipMemcpyDeviceToHost));
  uint32_t m = 0;
  HIP_TRY(hipMemcpy(&m, b->pcm_mask + cb, sizeof(m), hipMemcpyDeviceToHost));
  if (nwords) *nwords = (size_t)st.nout;
  if (silent_mask) *silent_mask = m;
  return 0;
}

int kq_bank_set_output_ssrc(kq_bank *b, int ch, uint32_t ssrc) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch)) {
    set_err("bad channel");
    return -1;
  }
  b->chans[ch].out_rtp.ssrc = ssrc;
  return 0;
}

int kq_bank_output_rtp_state(const kq_bank *b, int ch, kq_out_rtp_state *out) {
  BankScope dev_scope_(b);
  if (!out || !valid_ch(b, ch)) return -1;
  *out = b->chans[ch].out_rtp;
  return 0;
}

namespace {
// send_mono_output / send_stereo_output (audio.c:32-132) on the words of one channel-block: 480-word chunks, all-zero
// chunks skipped while the timestamp still advances, marker bit on the first packet after silence, sequence numbers on
// sent packets only.  `w`: nwords int16 in network byte order.  Packets back to back as [2-byte LE length][bytes].
int packetize_block(kq_out_rtp_state &o, const unsigned char *w, size_t nwords, bool stereo, unsigned char *dst, size_t cap,
                    size_t *used) {
  size_t pos = 0, left = nwords;
  int packets = 0;
  while (left > 0) {
    size_t const chunk = std::min<size_t>(480, left);  // PCM_BUFSIZE words, audio.c:19,44,94
    bool not_silent = false;
    for (size_t i = 0; i < 2 * chunk; i++) not_silent |= w[i] != 0;
    uint32_t const ts = o.timestamp;
    o.timestamp += (uint32_t)(stereo ? chunk / 2 : chunk);  // audio.c:52-53,103-104: advances even when nothing is sent
    if (not_silent) {
      o.packets++;
      o.bytes += (int64_t)(2 * chunk);
      int marker = 0;
      if (o.silent) {  // audio.c:57-61,109-113
        o.silent = 0;
        marker = 1;
      }
      uint16_t const seq = o.seq++;
      size_t const len = 12 + 2 * chunk;
      if (pos + 2 + len > cap) {
        set_err("packet buffer too small");
        return -1;
      }
      unsigned char *dp = dst + pos;
      dp[0] = (unsigned char)len;
      dp[1] = (unsigned char)(len >> 8);
      dp += 2;
      dp[0] = 2 << 6;  // RTP version 2; no padding, extension or CSRCs (multicast.c:285)
      dp[1] = (unsigned char)((marker << 7) | (stereo ? 10 : 11));
      dp[2] = (unsigned char)(seq >> 8);
      dp[3] = (unsigned char)seq;
      for (int k = 0; k < 4; k++) dp[4 + k] = (unsigned char)(ts >> (24 - 8 * k));
      for (int k = 0; k < 4; k++) dp[8 + k] = (unsigned char)(o.ssrc >> (24 - 8 * k));
      memcpy(dp + 12, w, 2 * chunk);
      pos += 2 + len;
      packets++;
    } else {
      o.silent = 1;
    }
    w += 2 * chunk;
    left -= chunk;
  }
  if (used) *used = pos;
  return packets;
}
}  // namespace

int kq_bank_pull_rtp_audio(kq_bank *b, int ch, unsigned blk, unsigned char *dst, size_t cap, size_t *used) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !dst) {
    set_err("bad channel or NULL buffer");
    return -1;
  }
  std::vector<int16_t> words(2 * (size_t)b->g.olen);
  size_t nwords = 0;
  if (kq_bank_pull_pcm(b, ch, blk, words.data(), words.size(), &nwords, nullptr)) return -1;
  bool const stereo = nwords == 2 * (size_t)b->g.olen;  // what the demodulator passed to send_stereo_output
  return packetize_block(b->chans[ch].out_rtp, reinterpret_cast<const unsigned char *>(words.data()), nwords, stereo, dst, cap, used);
}

// The same datagrams from planes the host already holds (kq_bank_pull_pcm_planes_async): no device access, no wait.
int kq_bank_rtp_from_planes(kq_bank *b, int ch, unsigned blk, const int16_t *pcm_plane, const kq_chan_status *status_plane,
                            unsigned char *dst, size_t cap, size_t *used) {
  if (!b) {
    set_err("NULL bank");
    return -1;
  }
  // Host work only, and meant to be spread over the host's threads by channel range: the bank's lock is held just long
  // enough to check the channel and take the address of its RTP state (b->chans is reserved for max_channels at create:
  // its elements never move).  One thread per channel at a time -- the caller's partition -- owns that state; a channel
  // removed or restarted while its packetiser runs is the caller's race, as two threads in audio.c:82 would be.
  kq_out_rtp_state *o = nullptr;
  int olen = 0, max_blocks = 0;
  {
    std::lock_guard<std::recursive_mutex> lk(b->mu);
    if (!valid_ch(b, ch) || !dst || !pcm_plane || !status_plane || blk >= (unsigned)b->g.max_blocks) {
      set_err("bad channel / block or NULL plane");
      return -1;
    }
    o = &b->chans[ch].out_rtp;
    olen = b->g.olen;
    max_blocks = b->g.max_blocks;
  }
  size_t const cb = (size_t)ch * max_blocks + blk;
  int const nout = status_plane[cb].nout;
  if (nout < 0 || nout > 2 * olen) {
    set_err("status plane: nout %d out of range", nout);
    return -1;
  }
  return packetize_block(*o, reinterpret_cast<const unsigned char *>(pcm_plane + cb * 2 * (size_t)olen), (size_t)nout,
                         nout == 2 * olen, dst, cap, used);
}

int kq_bank_pull_filter_output(kq_bank *b, int ch, unsigned blk, float *dst, size_t cap) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !dst || blk >= b->last_blocks || cap < (size_t)b->g.olen) {
    set_err("bad channel/block/capacity");
    return -1;
  }
  if (sync_all(b)) return -1;
  HIP_TRY(hipMemcpy(dst, b->pl.filt + ((size_t)ch * b->g.max_blocks + blk) * b->g.olen, b->g.olen * sizeof(float2),
                    hipMemcpyDeviceToHost));
  return 0;
}

int kq_bank_pull_spectrum(kq_bank *b, int ch, unsigned blk, float *dst, size_t cap) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !dst || cap < (size_t)b->g.N) {
    set_err("bad channel/capacity");
    return -1;
  }
  if (b->fwd_mode != KQ_FWD_FULL) {
    set_err("master spectrum exists only in KQ_FWD_FULL mode");
    return -1;
  }
  // The dump is armed for one channel at a time: the first pull after (re)arming returns -1 with a
  // hint; spectra are captured by the next kq_bank_process call.
  if (b->spec_ch != ch || !b->spec_dump) {
    if (!b->spec_dump && dev_alloc(&b->spec_dump, (size_t)b->g.max_blocks * b->g.N)) return -1;
    b->spec_ch = ch;
    set_err("spectrum capture armed for channel %d; it is filled by the next process call", ch);
    return -1;
  }
  if (blk >= b->last_blocks) {
    set_err("bad block");
    return -1;
  }
  if (sync_all(b)) return -1;
  HIP_TRY(hipMemcpy(dst, b->spec_dump + (size_t)blk * b->g.N, b->g.N * sizeof(float2), hipMemcpyDeviceToHost));
  return 0;
}

int kq_bank_get_response(kq_bank *b, int ch, float *dst, size_t cap) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !dst || cap < (size_t)b->g.Ndec) {
    set_err("bad channel/capacity");
    return -1;
  }
  if (fetch_response(b, ch)) return -1;
  memcpy(dst, b->chans[ch].resp.data(), sizeof(float2) * b->g.Ndec);
  return 0;
}

int kq_bank_get_audio_response(kq_bank *b, int ch, float *dst, size_t cap) {
  BankScope dev_scope_(b);
  if (!valid_ch(b, ch) || !dst) {
    set_err("bad channel");
    return -1;
  }
  auto const &a = b->chans[ch].aresp;
  if (a.empty() || cap < a.size()) {
    set_err("no audio response (not FM, or flat) or capacity too small");
    return -1;
  }
  memcpy(dst, a.data(), sizeof(float2) * a.size());
  return 0;
}

void *kq_bank_audio_device_ptr(kq_bank *b) { return b ? b->pl.audio : nullptr; }
void *kq_bank_status_device_ptr(kq_bank *b) { return b ? b->pl.status : nullptr; }

int kq_bank_enable_timing(kq_bank *b, int on) {
  BankScope dev_scope_(b);
  if (!b) return -1;
  if (!on && b->timing) drain_timing(b);
  b->timing = on;  // 0 off, 1 filter kernel only, >= 2 every scope
  return 0;
}

int kq_bank_get_timing(kq_bank *b, kq_timing *t, int reset) {
  BankScope dev_scope_(b);
  if (!b || !t) return -1;
  if (drain_timing(b)) return -1;
  *t = b->acc;
  if (reset) b->acc = kq_timing{};
  return 0;
}

int kq_bank_get_host_timing(kq_bank *b, kq_host_timing *t, int reset) {
  if (!b || !t) return -1;
  std::lock_guard<std::recursive_mutex> lk(b->mu);
  *t = b->host_acc;
  if (reset) {
    b->host_acc = kq_host_timing{};
    b->worst_holder = "";
  }
  return 0;
}

const char *kq_bank_worst_lock_holder(kq_bank *b) {
  if (!b) return "";
  std::lock_guard<std::recursive_mutex> lk(b->mu);
  return b->worst_holder;  // (a function name: static storage)
}

int kq_bank_fwd_mode(const kq_bank *b) { return b ? b->fwd_mode : -1; }

}  // extern "C"
