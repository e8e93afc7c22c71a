// kq_device.hpp -- device-side data layout shared by the kernels and the host bank.
//
// HBM layout (all planes allocated once at bank creation):
//   ring      float2 [M-1 + max_blocks*L]        front-end I/Q, one copy shared by every channel
//   resp      float2 [C][N_dec]                  pre-detection responses (filter.out->response)
//   aresp     float2 [C][N_dec/2+1]              FM de-emphasis responses
//   filt      float2 [C][max_blocks][olen]       filter.out->output.c per channel-block
//   audio     float  [C][max_blocks][2*olen]     demodulated audio
//   status    kq_chan_status [C][max_blocks]
//   per-channel parameter / carried-state vectors (SoA, indexed by channel)
#pragma once
#include "kq_energy.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ka9q_hip.h"

struct filter_in;  // include/ka9q_hip_compat.h

namespace kq {

enum { FLAG_FLAT = 1, FLAG_ISB = 2, FLAG_STEREO = 4, FLAG_SQUARE = 8 };

// Carried state of one carrier-tracking (PLL) linear channel, linear.c:97-112
struct PllState {
  double c_phase, c_freq;  // coarse (FFT-steered) oscillator: turns, cycles per output sample
  double f_phase, f_freq;  // fine (loop-steered) oscillator
  float integrator, delta_f, snr, foffset, cphase;
  int lock_count, pll_lock, fft_samples, fft_ptr;
};

// A transform size on the generic path (kq_ldsfft.hpp lds_fft_mixed): n = f[0] f[1] ... f[nf-1] with radices 2..5, or a
// power of two (log2n >= 0: then rev / twc are not used and the kernels run lds_fft as ever).  fft_dim() builds one on the
// current device: the tables are cached per (device, n) and never freed.  ok = false: n has a prime factor beyond 7, or
// is beyond 65536 (rev is 16 bits wide), or the allocation failed.
struct FftDim {
  int n;
  int log2n;                  // -1: not a power of two
  int nf;
  unsigned char f[12];
  const unsigned short *rev;  // [n] digit reversal: natural index i goes to position rev[i]; null for powers of two
  const float2 *twc;          // [tw_n] exp(-2 pi i k / tw_n)
  int tw_n;
};
FftDim fft_dim(int n, bool *ok);
bool fft_size_ok(int n);      // n = 2^a 3^b 5^c 7^d, 2 <= n <= 65536, even

// Carrier-tracking channels keep their loop state, 65536-sample ring and search scratch in a SLOT of their own for as long
// as they exist (kq_bank.cpp pll_acquire): storage grows by chunks of kPllChunk slots, nothing ever moves.
constexpr int kPllChunk = 64;
struct PllChunk {
  PllState *state;  // [kPllChunk]
  float2 *rings;    // [kPllChunk][65536]
  float2 *side;     // [kPllChunk][4096]
};

struct Geom {
  int N, L, M, D;
  FftDim dN, dNdec, dPl;  // the transforms of the generic path: master, slave / audio master, PL slave
  int Ndec, olen, Mdec;
  int log2N, log2Ndec;
  int samprate;
  int tw_log2;      // twiddle table period = 1 << tw_log2 (>= N)
  int max_blocks;
  float dsamprate;  // fm.c:27
  int pl_n, pl_l;   // PL slave geometry N_dec/32, olen/32 (0: PL measurement off)
};

constexpr int kOldLevels = 4;  // retune transitions a window may hold beyond the last one (ChanDev::hist2_*)

struct ChanDev {
  // configuration
  int *mode;            // enum kq_demod_type
  int *flags;           // as the demodulators see them (written on their stream, kq_bank.cpp ctl queues)
  int *fflags;          // the same word as the filter kernels see it (FLAG_ISB): written on the main stream
  float *low, *high;    // Hz, for compute_n0's passband exclusion
  float2 *resp;
  float2 *aresp;
  float *fm_gain;       // fm.c:86
  float *headroom;
  float *recovery;      // am.c:27 / linear.c:34 recovery_factor
  int *hangmax;
  float *noise_gain;
  // compute_n0's passband exclusion for k_filter_full16k (null: compute_n0 off): per (wave w, bin slot s = 16 half + k3) one
  // 64-bit lane mask, bit l set = thread t = 64 w + l's bin full16k_bin(t) + kFull16kHalf half + 1024 k3 lies OUTSIDE the
  // passband.  [C][8][32]; N = 65536: [C][4][8][32], one set per sub-transform (bins 4 q + r)
  unsigned long long *n0lane;
  unsigned *n0meta;     // [slot] number of bins outside the passband (N = 65536: [slot][4], per sub-transform)
  int *n0slot;          // [C] the channel's mask set in n0lane / n0meta (channels with the same filter edges share one)
  // oscillator phase (turns), step (cycles/sample) and sweep (cycles/sample^2) at sample 0 of the
  // first window of the call; second LO and Doppler already summed (they multiply, so phases add)
  double *lo_phase, *lo_freq, *lo_rate;
  // the same for the M-1 history samples of the call's first block: they were mixed before a retune took effect
  // and keep the old oscillator (radio.c:132-139 mixes sample by sample; osc.c:22-36 only changes what follows)
  double *hist_phase, *hist_freq, *hist_rate;
  // ... and how many samples from the start of the call's first window still carry it: sample i of block b's window is an old
  // one while b L + i < hist_len[c].  M - 1 at the first call after a retune; where M - 1 > L (the reference's default
  // -L 3840 -M 4353) the second block's window still reaches back that far, and with short calls so does the next call's.
  // Only read where the history planes differ from the current ones: the word of a channel that was not retuned is stale.
  int *hist_len;
  // A channel retuned again while samples of oscillators BEFORE the last one are still in the history (M - 1 > L and a retune
  // before every block): level l = 0 is the transition before the last, l + 1 the one before that.  The first
  // hist2_len[kOldLevels c + l] samples of the call's first window -- fewer with every level -- carry
  // hist2_osc[3 (kOldLevels c + l) ...] (phase, step, sweep at the window start).  Read only where hist_len says there are old
  // samples at all; a level of 0 samples ends the list.  (kq_ldsfft.hpp: load_older / pick_older)
  int *hist2_len;
  double *hist2_osc;
  // post-detection shift oscillator at output sample 0 of the call
  double *sh_phase, *sh_freq;
  // carried demodulator state
  float2 *fm_state;
  float *lastaudio;
  int *sq_count;
  float *ahist;         // [C][Mdec-1] FM audio overlap-save history
  float *foffset, *pdev;
  float *gain;          // agc.gain
  int *hang;
  float *dc;            // AM DC_filter
  float *n0;            // smoothed noise density
  // PL (CTCSS) tone measurement, fm.c:189-285 (only when N/D >= 128)
  float2 *plresp;       // [PL_N/2+1] shared: geometry only
  float *plring;        // [C][16384]
  int *pl_ptr, *pl_last;
  float *plfreq;
};

struct Planes {
  float2 *filt;
  float *audio;
  kq_chan_status *status;
  float *n0raw;         // [C][max_blocks] unsmoothed compute_n0 results
  float *if_power;      // [max_blocks]
  float *plout;         // [C][max_blocks][PL_L] PL filter outputs of the call (null when PL is off)
};

// Every entry point that takes a handle runs on the handle's device whatever the calling thread's current device is,
// and leaves the thread's device as it found it.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
};

// Raises the kernel's dynamic-LDS limit on the current device when `bytes` exceeds what was set before (the limit is
// kept per kernel and device: a process may drive banks on several devices).
void ensure_dynamic_lds(const void *kernel, size_t bytes);

// launchers (kq_kernels.hip)
// output planes to pinned host memory (a null host pointer skips its plane): of every audio row of `row` floats the
// first status.nout, and the status plane up to its last whole 16 bytes
void launch_copy_to_host(hipStream_t s, const float *audio, float *haudio, int row, const kq_chan_status *status, void *hstatus,
                         size_t rows);
void launch_copy_pcm_to_host(hipStream_t s, const float *audio, short *hpcm, unsigned *hmask, int row,
                             const kq_chan_status *status, void *hstatus, size_t rows, const int *mode_compact = nullptr,
                             int max_blocks = 1);
void launch_ingest(hipStream_t s, const void *src, int format, float2 *dst, size_t nsamples, float scale);
// IF power in two launches.  _sum (in front of the filter, whose row-paired samples it also writes): per-block partial
// sums of |s|^2 into sums[nblocks * block_energy_split(L)], and the call's parameter block from `params_host` (pinned,
// device-visible) into `params_dev`.  _iir (wherever the demodulators run, behind _sum): the recurrence over the blocks;
// `update` points into the device copy of the parameter block.
constexpr int kEnergySplitMax = 16;
int block_energy_split(int L);
void launch_block_energy_sum(hipStream_t s, const float2 *newsamples, int L, int nblocks, float *sums, const void *params_host,
                             void *params_dev, size_t params_bytes, float2 *paired, int hist, const double *prev_planes = nullptr,
                             unsigned nchan = 0, unsigned cmax = 0, double adv = 0, double adv_out = 0, const void *patch_records_host = nullptr,
                             int npatch = 0, const void *patch_bits_host = nullptr);
// control-plane writes of a call, gathered by the host in pinned memory (kq_bank.cpp CtlQueue): nrec records {dst, nbytes,
// payload offset | fill value}; one workgroup per record copies or fills 4-byte words
void launch_ctl_apply(hipStream_t s, const void *queue_host, int nrec);
void launch_block_energy_iir(hipStream_t s, const float *sums, int L, int nblocks, const unsigned char *update, float *energy_state,
                             float *if_power);
void launch_filter_full(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                        const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                        const int *chan_list);
// N = 65536 on the same kernel: four sibling workgroups per channel-block, each the 16384-point transform of one residue
// class of bins (k = 4 q + r); what they share travels through these planes (kq_full16k.hip)
struct Big64 {  // (N = 65536's hand-over places -- and, for every full-spectrum launch, what its side job needs: `iir`)
  unsigned long long *sync;  // [C][max_blocks][3][4] tagged words: first-pass sums of compute_n0 handed between the siblings
  float2 *n0part;            // [C][max_blocks][4] second pass: (sum, count) per sub-transform
  float2 *xs;                // [C][max_blocks][N_dec] the bins the slave reads, index k mod N_dec
  int *err;                  // set when a sibling's word never arrived
  unsigned epoch;            // tag of this launch
  IirArgs iir;               // sums != null: the last wave of workgroup (0, 0) runs the call's IF-power recurrence first
};
// register-resident N = 16384 variant of the same (kq_full16k.hip)
bool full16k_supported(const Geom &g);
// where k_filter_full16k leaves the spectrum: thread t holds bins full16k_bin(t) + kFull16kHalf * half + 1024 * k3
// (half = 0, 1; k3 = 0..15); ChanDev::n0lane is laid out to match
constexpr int kFull16kHalf = 16;
constexpr int full16k_bin(int t) { return (t >> 5) + 32 * (t & 31); }
// plain: no channel of the launch has a sweep rate or a retune pending (a leaner kernel variant serves that case);
// window_paired: the same samples with their 512-sample rows interleaved in pairs (launch_block_energy_sum writes it), which
// the plain variant loads 16 bytes at a time; null: it reads `window`
void launch_filter_full16k(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                           const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                           const int *chan_list, int plain, const float2 *window_paired, const Big64 &big);
bool full16k_paired_supported(const Geom &g);
bool full64k_supported(const Geom &g);
// plain: no channel of the launch was retuned since the last call and every sweep rate is inside full64k_sweep_limit();
// swept: some channel sweeps
void launch_filter_full64k(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                           const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                           const int *chan_list, bool plain, bool swept, const float2 *window_paired, const Big64 &big);
double full16k_sweep_limit();  // the same for the N = 16384 steady-state variant of swept channels (plain == 2)
double full64k_sweep_limit();  // |rate| in cycles per sample^2 up to which the table path's first-order cross term holds
bool split_supported(const Geom &g);
void launch_filter_split(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                         const float2 *tw, int nchan, int nblocks, const int *chan_list);
bool pruned_supported(const Geom &g);
void launch_filter_pruned(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                          const float2 *chan_tw, int nchan, int nblocks, bool swept, const int *chan_list, const IirArgs &iir = IirArgs{});
// the geometries whose pruned kernel runs the IF-power recurrence handed to it in `iir` (the others ignore it)
bool pruned_carries_iir(const Geom &g);
void launch_pruned_tables(hipStream_t s, const Geom &g, const ChanDev &ch, float2 *chan_tw, int nchan);
size_t demod_fm_lds_bytes(const Geom &g);
void launch_demods(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *tw,
                   const int *list_fm, int n_fm, const int *list_am, int n_am, const int *list_lin, int n_lin,
                   int nblocks, int compute_n0, float *fmout, const float *fm_hist_in, float *fm_hist_out);
bool demod64_supported(const Geom &g);
bool demod_agc_wave_supported(const Geom &g);
void launch_demod64(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const int *list_fm, int n_fm,
                    const int *list_am, int n_am, const int *list_lin, int n_lin, int nblocks, int compute_n0);
void launch_demod_pll(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *tw, const int *list_pll,
                      int n_pll, const PllChunk *chunks, const int *slot_of, int nblocks, int compute_n0);
// chan_list (both): the active channels when kq_bank_remove_channel has left holes (nchan = its length), else null
void launch_pcm(hipStream_t s, const Geom &g, const Planes &pl, short *pcm, unsigned *mask, int nchan, int nblocks,
                const int *chan_list);
void launch_pl_track(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *tw, const int *list_fm,
                     int n_fm, int nblocks);
// single transforms for the compat surface
void launch_fft_single(hipStream_t s, const float2 *in, float2 *out, const FftDim &d, int sign, const float2 *tw, int tw_log2);
// the same for 2^15 .. 2^22 points, through global memory (`tmp`: N elements of scratch)
int launch_fft_large(hipStream_t s, const float2 *in, float2 *out, float2 *tmp, int N, int sign, const float2 *tw, int tw_log2);
void launch_n0_single(hipStream_t s, const float2 *fdomain, int N, int samprate, float low, float high, float *out);
void launch_slave_single(hipStream_t s, const float2 *fdomain, const float2 *resp, float2 *out, int N, int Ndec,
                         int in_real, int out_type, const float2 *tw, int tw_log2);
void launch_slave_bank(hipStream_t s, const float2 *fdomain, const float2 *resp, float2 *out, int N, int Ndec, int olen,
                       int out_type, const float2 *tw, int tw_log2);
size_t pruned_table_elems(const Geom &g);


// ---- compat surface internals shared with the demodulator entry points (kq_compat.cpp, kq_radio.cpp)
// Copies the master's device-resident input window (N samples: M-1 history, L new) of the block last transformed to
// `dst` (device), ordered behind that transform on the compat stream, and waits for it.  Returns N, or -1.
// *block: the master's block number the copied window belongs to (the master may be one block ahead of its consumers)
int compat_snapshot_window(struct filter_in *master, float2 *dst, unsigned *block);
// the same for the master's device-resident SPECTRUM (N bins) of the block last transformed
int compat_snapshot_spectrum(struct filter_in *master, float2 *dst, unsigned *block);
int compat_master_device(void);  // device the compat surface runs on (the calling thread's current device at first use)

}  // namespace kq
