/* ka9q_hip.h -- C ABI of libka9q_hip.so: the MI355X (gfx950) implementation of ka9q-radio's
 * per-channel DSP hot path (NCO mix -> overlap-save filter/decimator -> FM / AM / linear demod).
 *
 * Plain C: pointers, sizes, ints and doubles only.  No HIP or torch types appear here; device
 * pointers and streams cross the boundary as void*.
 *
 * Two surfaces:
 *   1. kq_bank_*   -- the batched "channel bank": many receiver channels share one front-end I/Q
 *                     stream.  This is what a multi-channel `radio` binds.  Each
 *                     entry point cites the reference interface it replaces (paths relative to
 *                     the reference tree).
 *   2. compat      -- the reference's own filter.h / osc.h symbol names, one channel per object,
 *                     in ka9q_hip_compat.h.
 *
 * Error convention follows the reference (filter.c:148-149, 504-505): int functions return 0 on
 * success and -1 on a NULL / NaN / out-of-range argument; constructors return NULL on failure.
 * kq_last_error() adds a human-readable reason (the reference has none).
 */
#ifndef KA9Q_HIP_H
#define KA9Q_HIP_H 1

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum demod_type of radio.h:20-24 */
enum kq_demod_type { KQ_LINEAR_DEMOD = 0, KQ_AM_DEMOD = 1, KQ_FM_DEMOD = 2 };

/* I/Q sample formats accepted by kq_bank_push_iq:
 *   KQ_IQ_CF32 : complex float (re,im)           -- the synthetic configs of BASELINE.json
 *   KQ_IQ_S16  : int16 little-endian I,Q  (RTP payload type IQ_PT  = 97, multicast.h:19; radio.c:113-114)
 *   KQ_IQ_S8   : int8 I,Q                 (RTP payload type IQ_PT8 = 98, multicast.h:20; radio.c:117-118) */
enum kq_iq_format { KQ_IQ_CF32 = 0, KQ_IQ_S16 = 1, KQ_IQ_S8 = 2 };

/* Forward-transform strategy of the pre-detection filter */
enum kq_fwd_mode {
  KQ_FWD_AUTO = 0,    /* pruned when the geometry allows and n0 is off, else full */
  KQ_FWD_FULL = 1,    /* N-point FFT per channel in LDS, all N bins (needed by compute_n0) */
  KQ_FWD_PRUNED = 2   /* only the N/D bins execute_filter_output reads (filter.c:206-227) */
};

typedef struct kq_bank kq_bank;   /* opaque */
/* Threads: a handle carries one lock, taken by every kq_bank_* entry point, so a receiver thread (process / push / pull) and
 * an operator's thread (set_filter, set_mode, the frequency setters, add / remove -- display.c and radio_status.c beside the
 * demodulator threads of the reference) may share a bank.  The entry points whose job is to wait for the device --
 * kq_bank_pull_wait, kq_bank_host_io_wait, kq_bank_sync -- let go of the lock while they wait.  kq_last_error is per thread.
 * kq_bank_destroy wants the handle to itself.  Buffers handed to the asynchronous pulls are the caller's to keep apart. */

/* Bank geometry: replaces the arguments of create_filter_input(L, M, COMPLEX) (filter.c:54,
 * called from main.c:232) plus demod->filter.decimate and demod->input.samprate
 * (radio_status.c:264-267), which the reference keeps per process.
 * Limits (deviations from the reference, which hands any L / M to FFTW -- main.c:160-170, filter.c:78,132):
 *   - N = L + M - 1 and N / decimate: powers of two (the reference's default, N = 8192, is one), or -- since round 6, on
 *     the generic kernels -- any even 2^a 3^b 5^c 7^d up to 16384 (up to 65536 without compute_n0, where N splits into a few
 *     transforms of at most 16384 points: 24000, 38400, 48000 ...): the sizes of a front end whose rate is not 48 kHz x 2^k
 *     (decimate = samprate / 48000, radio_status.c:266: 5 at 240 kHz with L = 4800, M = 4801).  A prime factor beyond 7 is
 *     refused.  decimate must divide N, L and M - 1 (the reference only warns when it does not, filter.c:106-107); decimate 1
 *     (a 48 kHz front end) needs N <= 8192.  The fast
 *     kernels (N = 16384 / 65536 full-spectrum, the pruned ones, N/D = 64 and 256 demodulators) are power-of-two only;
 *   - the full-spectrum forward path (compute_n0 = 1) serves N <= 16384 and N = 65536; N = 32768 only without compute_n0.
 *     At N = 65536 the four workgroups of a channel-block wait for each other (bounded: a lost one gives NaN and an error
 *     from kq_bank_sync, never a hung device): run such banks from ONE process per GPU -- two processes' launches in flight on
 *     one device can starve each other's workgroups until the waits run out;
 *   - carrier-tracking channels (kq_channel_config.pll, linear.c:129-246): as many as the bank has channels, up to 65536;
 *     each keeps a 65536-sample search ring (544 KiB with its state), allocated 64 channels at a time as the count grows;
 *   - the PL-tone measurement (fm.c:201-205 decimates by 32) runs where N / decimate / 32 is at least 4 and a size this
 *     library has a transform for -- even, no prime factor beyond 7; where 32 does not divide the sizes they are truncated as
 *     create_filter_output truncates them (filter.c:103-107,116), the tone reading that much off, as in the reference;
 *   - FM channels need N / decimate <= 8192 (the post-detection filter lives in one CU's LDS).
 * Nothing limits max_channels but memory (~3 KiB per channel + the planes): banks of 34560 channels run in the bench. */
typedef struct kq_bank_config {
  int device;              /* HIP device ordinal.  A handle stays on its device: every call taking the handle runs
                            * there and restores the calling thread's current device, so one process can drive
                            * banks on several GPUs (kq_bank_create included). */
  int samprate;            /* front-end complex sample rate, Hz */
  unsigned L;              /* new samples per block            (demod->filter.L) */
  unsigned M;              /* impulse response length          (demod->filter.M); N = L+M-1: see Limits */
  unsigned decimate;       /* D: output rate = samprate / D    (demod->filter.decimate), >= 1 (1 -- a 48 kHz front end -- with
                              L + M - 1 <= 8192) */
  unsigned max_channels;
  unsigned max_blocks;     /* largest number of blocks one kq_bank_process call may take */
  float gain_factor;       /* demod->sdr.gain_factor (radio.c:122); 1.0 for synthetic float input */
  int compute_n0;          /* run the status-only compute_n0 (radio.c:383-425) every block */
  int fwd_mode;            /* enum kq_fwd_mode */
  void *stream;            /* hipStream_t to launch on, or NULL for the bank's own stream */
  int pl_tone_off;         /* 0 (default): pltask's PL / CTCSS tone measurement (fm.c:189-285) runs for every FM channel whenever
                            * its slave has a usable size (N/D >= 128), as every demod_fm of the reference starts it;
                            * 1: never (status.plfreq = NaN) -- SURVEY 8d measures the synthetic configs 2-5 without it */
} kq_bank_config;

/* One receiver channel: the fields of struct demod / struct modetab that the path reads
 * (radio.h:64-193, radio.h:35-50; defaults in modes.txt:25-39, main.c:113-117). */
typedef struct kq_channel_config {
  int demod_type;          /* enum kq_demod_type */
  int flat;                /* opt.flat : FM without de-emphasis filter and gain (fm.c:55, 164-172) */
  int isb;                 /* filter.isb : CROSS_CONJ slave (linear.c:78-79, filter.c:239-249) */
  int channels;            /* output.channels : 1 mono, 2 stereo (linear.c:291-300) */
  float low, high;         /* filter.low / filter.high, Hz */
  float kaiser_beta;       /* filter.kaiser_beta */
  float headroom;          /* agc.headroom, amplitude ratio (main.c:117: 10^(-15/20)) */
  float hangtime;          /* agc.hangtime, s */
  float recovery_rate;     /* agc.recovery_rate, dB/s */
  double second_lo;        /* set_second_LO() argument, Hz (radio.c:290-301); 0 freezes the NCO */
  double doppler;          /* set_doppler() arguments, Hz and Hz/s (radio.c:180-184) */
  double doppler_rate;
  double shift;            /* set_shift() argument, Hz (radio.c:304-311) */
  int pll;                 /* opt.pll : carrier-tracking PLL in linear mode (linear.c:129-246; modes CAM, AME, CISB) */
  int square;              /* opt.square : squaring loop for suppressed-carrier DSB / BPSK (mode DSB) */
} kq_channel_config;

/* Per channel, per block signal status: the demod->sig.* / agc.gain values the path produces
 * (radio.h:164-175), plus the integer decision state that must match the reference exactly. */
typedef struct kq_chan_status {
  float if_power;          /* radio.c:143-145 (halving accumulator, shared by all channels) */
  float bb_power;          /* fm.c:99, am.c:78, linear.c:302 */
  float n0;                /* smoothed noise density (fm.c:78-82, am.c:46-49); NaN when compute_n0 = 0 */
  float snr;               /* fm.c:102-103; NaN for linear without PLL (linear.c:309); 0 for AM */
  float foffset;           /* fm.c:148 */
  float pdeviation;        /* fm.c:153 */
  float agc_gain;          /* am.c / linear.c agc.gain after the block */
  float noise_gain;        /* filter.out->noise_gain (filter.c:472-497) */
  float plfreq;            /* fm.c:189-285 CTCSS (PL) tone estimate, Hz; NaN: none, not FM, or N/D < 128 */
  float cphase;            /* linear.c:219-223 carrier phase of the block, rad (PLL modes; else 0) */
  int32_t pll_lock;        /* linear.c:162-169 */
  int32_t lock_count;      /* linear.c:157-170 (sig.lock_timer) */
  int32_t squelch_count;   /* fm.c:70 snr_below_threshold after the block */
  int32_t hangcount;       /* am.c:26 / linear.c:33 hangcount after the block */
  int32_t blanked;         /* FM samples held at lastaudio this block (fm.c:141) */
  int32_t nout;            /* floats of audio this block: olen (mono) or 2*olen (stereo) */
} kq_chan_status;

/* The part of kq_chan_status a receiver reads with every block (kq_bank_pull_pcm_planes_compact_async): at real time the
 * 64-byte record is half of what a PCM delivery moves over the link, and most of it changes at the rate of the status
 * packets (radio_status.c), not of the blocks.  The whole record stays available through kq_bank_pull_status and the
 * full-plane pulls. */
typedef struct kq_chan_status_compact {
  float bb_power;          /* as kq_chan_status */
  float n0;
  float snr;
  float aux;               /* FM: foffset (fm.c:148); AM and linear: agc_gain (agc.gain after the block) */
  int32_t state;           /* FM: squelch_count (fm.c:70); AM and linear: hangcount */
  int32_t nout;            /* words of audio this block */
} kq_chan_status_compact;

/* Per-kernel device time accumulated since the last reset (HIP events on the bank's stream) */
typedef struct kq_timing {
  double filter_ms;        /* pre-detection filter kernel (mix + forward FFT + response + IFFT) */
  double demod_ms;         /* FM + AM + linear demodulator kernels */
  double ingest_ms;        /* format conversion + IF power */
  uint64_t filter_launches;
  uint64_t channel_blocks; /* channel-blocks processed by the filter kernel */
  double filter_max_ms;    /* the longest single filter INTERVAL of a call since the last reset (timing level 1: from the marker
                              queued in front of the call's filter kernels to the one behind them, as the device's queue
                              processed them).  It holds the kernels' run time, anything that kept the queue from starting them,
                              and the host's own delay between queueing the two markers: */
  double filter_max_submit_ms; /* ... the host's part of that longest interval: wall time between queueing its two markers.
                              Small (tens of microseconds) = the time went by on the device's side; about the interval = the
                              calling thread was off its core in the middle of the call */
  uint64_t filter_max_launch;  /* ... and which filter pass since the last reset it was (0-based; to find it in a kernel trace) */
} kq_timing;

const char *kq_last_error(void);
const char *kq_version(void);
/* The structs of this header carry no size fields; the library counts their revisions instead.  A host built against
 * this header checks kq_abi_version() == KQ_ABI_VERSION once at start-up: a mismatch means a struct (kq_bank_config,
 * kq_fanout_info ...) has gained a field since the host was compiled. */
#define KQ_ABI_VERSION 6
int kq_abi_version(void);
/* Number of visible HIP devices, or -1 when the HIP runtime cannot be initialised */
int kq_device_count(void);
/* Page-locked host memory for the streaming entry points (kq_bank_push_iq_async, kq_bank_pull_planes_async,
 * kq_bank_pull_pcm_planes_async): what a host would otherwise take from hipHostMalloc, so that a C, cgo or JNI host links
 * this library alone.  Usable with every device of the process.  NULL on failure (kq_last_error says why). */
void *kq_host_alloc(size_t bytes);
void kq_host_free(void *p);

/* --- lifetime --- */
kq_bank *kq_bank_create(const kq_bank_config *cfg);               /* main.c:232 create_filter_input */
int kq_bank_destroy(kq_bank *bank);                               /* filter.c:254 delete_filter_input */

/* --- channels: what set_mode() + the demod thread prologue do (radio.c:322-374; fm.c:27-67,
 *     am.c:21-41, linear.c:29-81): creates the slave, designs its response, arms the demodulator.
 *     Returns the channel index (>= 0) or -1. --- */
int kq_bank_add_channel(kq_bank *bank, const kq_channel_config *cfg);
/* n channels at once -- the same result as n calls of kq_bank_add_channel, but every distinct response is designed once
 * (one launch for all), every distinct compute_n0 mask is built once and every per-channel plane is uploaded with one copy:
 * a bank of tens of thousands of channels is set up in a second instead of a minute.  indices (may be NULL) receives the
 * channel numbers.  All or nothing: returns n, or -1 with no channel added.  Meant for set-up: its copies are immediate, so it
 * waits for the calls in flight first -- on a running bank kq_bank_add_channel per channel does not (0.02 ms of host time each). */
int kq_bank_add_channels(kq_bank *bank, const kq_channel_config *cfgs, unsigned n, int *indices);
/* The demodulator thread's epilogue once demod->terminate is set and it has been joined (fm.c:177-182, am.c:80,
 * linear.c:319: delete_filter_output on its slaves).  The other channels keep their numbers; the slot is a hole that
 * the next kq_bank_add_channel reuses (lowest hole first, with prologue state), and holes at the end are dropped.
 * The pre-detection filter launch still spans holes, so a bank with many long-lived ones is better rebuilt.
 * Every per-channel call on a removed channel fails with -1. */
int kq_bank_remove_channel(kq_bank *bank, int ch);
/* 1 when `ch` names a live channel, 0 for a hole or an index out of range */
int kq_bank_channel_active(const kq_bank *bank, int ch);
/* One past the highest live channel number (holes included) */
unsigned kq_bank_num_channels(const kq_bank *bank);
/* set_mode (radio.c:322-374) on a running channel: the demodulator is torn down and started afresh with the mode's
 * demod_type, flat, isb, channels, pll, square, recovery_rate, hangtime, low / high (swapped when low > high),
 * kaiser_beta, headroom and shift -- new slave and response, demodulator state back at its prologue values (AGC gain,
 * squelch, FM state, audio filter history, carrier loop).  What struct demod keeps survives: both input
 * oscillators keep running (second_lo / doppler of `mode` are ignored), the shift oscillator keeps its phase,
 * sig.n0, sig.foffset and sig.pdeviation keep their values.  (The reference then re-runs set_freq to pull the second
 * LO back into range, radio.c:370 -- control plane, left to the caller.) */
int kq_bank_set_mode(kq_bank *bank, int ch, const kq_channel_config *mode);

/* --- tuning, all phase continuous and effective from the next block (osc.c:22-36) ---
 * None of the calls of this section, nor kq_bank_add_channel / remove_channel / set_mode / set_filter / set_n0, waits for
 * the device: what they change gathers in pinned queues and the next kq_bank_process applies it on the device, behind the
 * calls already queued (which keep the values they were queued with) and in front of its own kernels -- a new filter
 * response included, which is designed on the bank's stream where it is used (kq_bank_get_response then fetches it).  A
 * bank running at real time with several calls in flight is not stalled by its control plane (0.001-0.05 ms of host time
 * per operation at 32768 channels; carrier-tracking channels included since round 6: a loop's slot is its own and is
 * started afresh by the queue.  Only kq_bank_get_response of a channel whose filter has just been set waits, and a
 * carrier-tracking channel that needs a new chunk of 64 slots allocated). */
/* demod->filter.isb and demod->output.channels of a running linear channel (linear.c:117-120, 291-300): the slave's
 * out_type and the mono / stereo hand-off change from the next block on; AGC state and response are left alone. */
int kq_bank_set_linear_options(kq_bank *bank, int ch, int isb, int channels);
/* Second LO and Doppler: the new setting mixes the samples pushed from now on; the M - 1 samples before them, the history of
 * the windows to come, keep what they were mixed with (radio.c:132-139 mixes sample by sample).  That holds for every block
 * the old samples reach into -- one where M - 1 <= L, two with the reference's default -L 3840 -M 4353, across calls if the
 * calls are short.  A channel may be set again while old samples of its previous settings are still inside its history
 * (M - 1 > L and a retune before every block): up to five transitions per window are kept apart; a sixth inside the same
 * M - 1 samples (only possible where M - 1 > 5 L) treats the oldest samples as mixed with the oscillator after theirs. */
int kq_bank_set_second_lo(kq_bank *bank, int ch, double hz);                 /* radio.c:290 set_second_LO */
int kq_bank_set_doppler(kq_bank *bank, int ch, double hz, double hz_per_s);  /* radio.c:180 set_doppler */
int kq_bank_set_shift(kq_bank *bank, int ch, double hz);                     /* radio.c:304 set_shift */
int kq_bank_set_filter(kq_bank *bank, int ch, float low_hz, float high_hz, float kaiser_beta); /* filter.c:500 set_filter */
/* demod->sig.n0 as the demodulator finds it when it starts: NaN = the first compute_n0 result is taken as it comes,
 * anything else = the smoothing goes on from there (fm.c:78-82, am.c:46-49, linear.c:123-126).  A fresh channel has NaN. */
int kq_bank_set_n0(kq_bank *bank, int ch, float n0);

/* --- data path --- */
/* Append nsamples complex samples of the given format to the bank's input ring.  `iq` is a host
 * pointer, or a device pointer when is_device != 0.  Conversion/scaling as radio.c:110-122. */
int kq_bank_push_iq(kq_bank *bank, const void *iq, size_t nsamples, int format, int is_device);
/* One front-end datagram exactly as `radio` takes it off the wire: RTP header incl. CSRCs, extension
 * and padding (multicast.c:242-277, main.c:318-328), payload type IQ_PT = 97 (int16 I/Q) or IQ_PT8 = 98 (int8 I/Q)
 * (multicast.h:19-20; anything else is ignored, main.c:329-330), the obsolete 24-byte status block skipped
 * (main.c:338-341, sdr.h:18-48), then proc_samples' bookkeeping (radio.c:62-104): rtp_process (multicast.c:305-340)
 * decides whether the packet is a duplicate / stale (dropped), in sequence, or ahead with a timestamp gap -- in which
 * case the lost samples are injected as zeros with the LOs kept running (kq_bank_push_zeros) before the payload is
 * converted and appended (kq_bank_push_iq).  Packets are taken in arrival order; the reference's small
 * sort-by-sequence queue (main.c:347-357) stays with the caller.
 * Returns the number of samples appended by this call (zeros + payload), 0 for an ignored or dropped datagram, -1 on
 * error, and -2 when the ring has no room right now: run kq_bank_process and hand the SAME datagram in again, until it
 * is taken --   while ((n = kq_bank_push_rtp(b, pkt, len)) == -2) kq_bank_process(b);
 * The ring holds max_blocks * L samples plus L - 1 of slack, so a partly filled block never stands in the way of a packet
 * that fits the ring as such.  With whole blocks waiting, -2 leaves sequence, timestamp and counters untouched.  With
 * none waiting the gap's zero fill (up to 192000 samples, radio.c:79-100) is itself larger than the ring: each call then
 * puts in as many zeros as fit and moves the timestamp past them before it returns -2, so every round brings the gap a
 * ring closer to its end, sample for sample as the reference fills it; the packet is counted once.  A payload larger
 * than the whole ring is an error (-1; its samples become a gap that the next packet fills with zeros).
 * The call does not wait for the device: payloads gather in pinned host memory and go to the ring as one asynchronous
 * copy + conversion per run of packets, queued by the next kq_bank_process (or by any other call that touches the ring);
 * kq_bank_blocks_ready counts them at once. */
int kq_bank_push_rtp(kq_bank *bank, const void *datagram, size_t size);
typedef struct kq_rtp_counters {   /* struct rtp_state (multicast.h:41-50) + demod->input.samples */
  uint32_t ssrc;
  uint16_t next_seq;
  uint32_t next_timestamp;
  int64_t packets, drops, dupes;
  int64_t samples;
} kq_rtp_counters;
int kq_bank_rtp_counters(const kq_bank *bank, kq_rtp_counters *out);
/* Lost-sample zero fill (radio.c:81-100): append `nsamples` zeros; LOs keep running. */
int kq_bank_push_zeros(kq_bank *bank, size_t nsamples);
/* Number of complete blocks waiting in the ring */
unsigned kq_bank_blocks_ready(const kq_bank *bank);
/* Run mix + filter + demod for up to max_blocks waiting blocks on every channel
 * (radio.c:140-146 execute_filter_input, then each demod thread's loop body).
 * Asynchronous on the bank's stream.  Returns the number of blocks processed, or -1. */
int kq_bank_process(kq_bank *bank);
/* Convenience for resident-input benchmarks: process `nblocks` blocks reading the window
 * [M-1 history | nblocks*L] straight from `iq_dev` (device, complex float), no ring copy. */
int kq_bank_process_resident(kq_bank *bank, const void *iq_dev, unsigned nblocks);
/* The part of the path that follows the master (what a demodulator thread of the reference does with
 * filter.in->fdomain): `spectrum_dev` = nblocks x N complex float bins in device memory, the master's forward
 * transform of each block (execute_filter_input, filter.c:151) -- execute_filter_output (filter.c:206-250),
 * compute_n0 (radio.c:383-425) and the demodulators run on it; nothing is mixed or transformed forward.  Channels must
 * have no second LO / Doppler set (that mix sits in front of the master, radio.c:132-139); status.if_power is left 0
 * (radio.c:143-145 belongs to whoever fed the master).  The demodulator thread entry points use this on the compat
 * master's resident spectrum. */
int kq_bank_process_spectrum(kq_bank *bank, const void *spectrum_dev, unsigned nblocks);
/* The demodulators of a call may run on a second internal stream, overlapping the next call's filter pass (the bank
 * decides per call: after kq_bank_pull_planes_async, or with AM / SSB channels; KQ_DEMOD_OVERLAP=0 / 1 forces it).
 * kq_bank_join makes the bank's main stream wait (on the device) for the last call's demodulators wherever they ran;
 * kq_bank_sync applies what the control plane has queued since the last call and blocks the host until everything
 * issued so far has finished. */
int kq_bank_join(kq_bank *bank);
int kq_bank_sync(kq_bank *bank);
/* The hipStream_t the bank launches on, as void*: kq_bank_config.stream, or the bank's own stream when that was NULL.
 * A C host without the HIP headers hands it to kq_fanout_acquire / kq_fanout_release as the consumer stream. */
void *kq_bank_stream(kq_bank *bank);

/* --- streaming host I/O ---
 * Both ends of the path are host buffers in the reference: a packet's samples in (radio.c:106-147), one float buffer
 * per block out (audio.c:82 send_mono_output(demod, samples, olen)).  These three move whole batches between PINNED host
 * memory (hipHostMalloc / hipHostRegister) and the bank on copy streams of their own, under the kernels of the
 * neighbouring calls; nothing here blocks the host but kq_bank_host_io_wait.
 *  kq_bank_push_iq_async:     kq_bank_push_iq from host memory without the wait: the samples travel on the input copy
 *                             stream into one of two staging buffers, the conversion kernel follows them on the bank's
 *                             stream.  `iq` must stay unchanged until kq_bank_host_io_wait or until two more pushes
 *                             have been queued.
 *  kq_bank_pull_planes_async: queues the copy of the last call's outputs -- audio [channels][max_blocks][2 * olen] float
 *                             (of each channel-block the first status.nout floats are written, the rest of its row is
 *                             left as it was) and status [channels][max_blocks] (either may be NULL) -- on the output copy stream;
 *                             the next call's demodulators wait for it on the device before they overwrite the planes.
 *  kq_bank_pull_pcm_planes_async: the same delivery in the reference's own output format (audio.c:22-28, 45-50, 95-100):
 *                             pcm [channels][max_blocks][2 * olen] clipped int16 in network byte order (status.nout words
 *                             per channel-block), silent_mask [channels][max_blocks] (may be NULL; bit i set: the i-th
 *                             480-word chunk is all zero, the packet send_mono_output would not send) and the status
 *                             plane -- half the bytes of the float plane; the conversion rides in the copy kernel and
 *                             does not need kq_bank_enable_pcm.  olen a multiple of 8.
 *  kq_bank_pull_pcm_planes_compact_async: the same with kq_chan_status_compact records (24 bytes per channel-block
 *                             instead of 64; the plane 16-byte aligned) -- a third fewer bytes per delivery at N/D = 64.
 *  kq_bank_pull_wait:         blocks until the delivery queued `lag` deliveries before the newest one has landed (0: the
 *                             newest; at most 7) -- a streaming host takes the planes of call k-2 in hand while calls
 *                             k-1 and k are in flight.
 *  kq_bank_host_io_wait:      blocks until every queued copy has landed.
 * Call order for full overlap: process batch k, push batch k+1, then pull the planes of batch k --
 *     push(0); for k: { process(); push(k + 1); pull_planes(k); }
 * an input copy queued behind an output copy may share its hardware queue and then waits with it for the demodulators. */
int kq_bank_push_iq_async(kq_bank *bank, const void *iq_pinned, size_t nsamples, int format);
int kq_bank_pull_planes_async(kq_bank *bank, float *audio_pinned, kq_chan_status *status_pinned);
int kq_bank_pull_pcm_planes_async(kq_bank *bank, int16_t *pcm_pinned, uint32_t *silent_mask_pinned, kq_chan_status *status_pinned);
int kq_bank_pull_pcm_planes_compact_async(kq_bank *bank, int16_t *pcm_pinned, uint32_t *silent_mask_pinned,
                                          kq_chan_status_compact *status_pinned);
int kq_bank_pull_wait(kq_bank *bank, unsigned lag);
int kq_bank_host_io_wait(kq_bank *bank);

/* --- results of the last kq_bank_process call (replace send_mono_output/send_stereo_output,
 *     audio.c:82 / audio.c:32, and the sig.* fields read by radio_status.c:170-203) --- */
unsigned kq_bank_olen(const kq_bank *bank);                       /* filter.c:116 */
unsigned kq_bank_last_blocks(const kq_bank *bank);
/* Audio of channel ch for block blk of the last call: status.nout floats. Synchronises. */
int kq_bank_pull_audio(kq_bank *bank, int ch, unsigned blk, float *dst, size_t cap, size_t *n);
int kq_bank_pull_status(kq_bank *bank, int ch, unsigned blk, kq_chan_status *st);
/* PCM output stage (audio.c:22-28 scaleclip, audio.c:45-50 / 95-100): once enabled, every process call
 * also converts the audio plane to clipped int16 in network byte order on the device.  kq_bank_pull_pcm returns the
 * status.nout words of one channel-block and a mask whose bit i is set when the i-th 480-word chunk is all zero --
 * send_mono_output / send_stereo_output skip such a packet but still advance the RTP timestamp (audio.c:101-104).
 * Geometries with more than 32 packets per block (2 * olen > 15360 words) are refused. */
int kq_bank_enable_pcm(kq_bank *bank, int on);
int kq_bank_pull_pcm(kq_bank *bank, int ch, unsigned blk, int16_t *dst_be, size_t cap_words, size_t *nwords,
                     uint32_t *silent_mask);
/* The datagrams send_mono_output / send_stereo_output (audio.c:32-132) would hand to send() for one channel-block,
 * built on the host from the PCM plane: RTP header (multicast.c:282-294; payload type PCM_MONO_PT = 11 or
 * PCM_STEREO_PT = 10, multicast.h:22-23), 480-word chunks, all-zero chunks skipped while the timestamp still
 * advances, marker bit on the first packet after silence, sequence numbers on sent packets only.  Packets are written
 * back to back as [2-byte little-endian length][bytes].  The per-channel state (demod->output.rtp, output.silent)
 * advances, so call it once per channel-block, in order.  Returns the number of packets, or -1. */
typedef struct kq_out_rtp_state {
  uint32_t ssrc;
  uint16_t seq;
  uint32_t timestamp;
  int32_t silent;
  int64_t packets, bytes;
} kq_out_rtp_state;
int kq_bank_set_output_ssrc(kq_bank *bank, int ch, uint32_t ssrc);
int kq_bank_pull_rtp_audio(kq_bank *bank, int ch, unsigned blk, unsigned char *dst, size_t cap, size_t *used);
/* The same datagrams built from planes the host already holds -- pcm [channels][max_blocks][2 * olen] and status
 * [channels][max_blocks] as kq_bank_pull_pcm_planes_async delivered them: host work only, no device access, no wait; the
 * per-channel RTP state advances exactly as with kq_bank_pull_rtp_audio (call it once per channel-block, in order; do
 * not mix the two on one channel-block).  Channels are independent: a host with tens of thousands of them spreads the
 * channel range over its threads. */
int kq_bank_rtp_from_planes(kq_bank *bank, int ch, unsigned blk, const int16_t *pcm_plane, const kq_chan_status *status_plane,
                            unsigned char *dst, size_t cap, size_t *used);
int kq_bank_output_rtp_state(const kq_bank *bank, int ch, kq_out_rtp_state *out);
/* Pre-detection filter output (filter.out->output.c, olen complex) before demodulation */
int kq_bank_pull_filter_output(kq_bank *bank, int ch, unsigned blk, float *dst_re_im, size_t cap_complex);
/* Master spectrum fdomain[N] of one channel/block (only in KQ_FWD_FULL mode; radio.c:396) */
int kq_bank_pull_spectrum(kq_bank *bank, int ch, unsigned blk, float *dst_re_im, size_t cap_complex);
/* Designed responses (filter.out->response, N/D complex; FM audio response N/D/2+1).  The pre-detection response of a
 * channel added or changed one at a time lives on the device (it is designed there, on the bank's stream): the first
 * kq_bank_get_response after such a change applies what is queued, waits for the device and fetches it. */
int kq_bank_get_response(kq_bank *bank, int ch, float *dst_re_im, size_t cap_complex);
int kq_bank_get_audio_response(kq_bank *bank, int ch, float *dst_re_im, size_t cap_complex);
/* Device-resident result planes for zero-copy consumers:
 *   audio  : float [max_channels][max_blocks][2*olen]
 *   status : kq_chan_status [max_channels][max_blocks] */
void *kq_bank_audio_device_ptr(kq_bank *bank);
void *kq_bank_status_device_ptr(kq_bank *bank);

/* --- measurement --- */
/* on = 0: off; 1: HIP events around the filter kernel only (two stream operations per call); 2: around the ingest /
 * IF-power and demodulator kernels as well */
int kq_bank_enable_timing(kq_bank *bank, int on);
int kq_bank_get_timing(kq_bank *bank, kq_timing *t, int reset);
/* The HOST's own time inside kq_bank_process / _resident / _spectrum since the last reset -- what scales with the number of
 * channels on the host side of a call (the kernels are only queued there).  A receiver at 1.0 x real time with tens of
 * thousands of channels and a batch of a few blocks has about a millisecond per call: this is the figure to watch.
 * No reference analogue (each `radio` process steps its own oscillators per sample, radio.c:132-136). */
typedef struct kq_host_timing {
  double call_ms;          /* wall time inside the process calls, everything included */
  double stage_ms;         /* of that: evaluating and staging the per-channel oscillator parameters of the call */
  double slot_wait_ms;     /* of that: blocked because the device was four calls behind (back-pressure, not work); taken
                              with the handle's lock let go, so control-plane calls from another thread run meanwhile */
  uint64_t calls;
  /* The handle's lock between the receiver thread (the process calls) and whoever drives the control plane (set_* / add /
   * remove from another thread; main.c's UI and status threads beside proc_samples): */
  double lock_wait_ms;     /* total the process calls waited to get the lock (not part of call_ms) */
  double lock_wait_max_ms; /* the worst single wait: what a control-plane call cost the stream at most */
  double ctl_hold_max_ms;  /* the longest any OTHER entry point held the lock (device waits with the lock let go excluded) */
} kq_host_timing;
int kq_bank_get_host_timing(kq_bank *bank, kq_host_timing *t, int reset);
/* The entry point behind ctl_hold_max_ms (its name, e.g. "kq_bank_set_filter"; "" when nothing has held the lock since the
 * last reset).  Read it before the kq_bank_get_host_timing call that resets. */
const char *kq_bank_worst_lock_holder(kq_bank *bank);
/* Which forward path the bank resolved to (enum kq_fwd_mode, never AUTO) */
int kq_bank_fwd_mode(const kq_bank *bank);

/* ---- multi-GPU: channel shards and the front-end fan-out ------------------------------------------------------------
 * The reference runs one `radio` process per channel and fans the front-end I/Q stream out to them by UDP multicast
 * (multicast.c:143-237, README.md:470-477).  Here one process (or thread) per GPU owns a contiguous range of the
 * channels and every batch of front-end samples is broadcast from the ingest rank over RCCL / xGMI (ncclBroadcast on a
 * side stream, two slots so that batch k+1 travels while batch k is processed).  Channels are independent: this is
 * the only exchange on the path.  librccl is loaded on first use.
 *
 *   rank `root`:  kq_fanout_unique_id(id)  -> hand the 128 bytes to the other ranks (file, socket, MPI ...)
 *   every rank:   f = kq_fanout_create(device, rank, world, root, id, samples_per_batch);
 *                 COLLECTIVE: ncclCommInitRank inside returns only once all `world` ranks have called it with the same
 *                 id -- one thread or process per rank, all calling concurrently (one thread creating the fan-outs of
 *                 several ranks one after the other would wait on itself for ever).  A rank whose own set-up fails
 *                 (memory, stream) still enters the communicator, and the ranks then agree through a one-word
 *                 ncclAllReduce, issued by every rank that holds a communicator whatever else went wrong on it: either
 *                 every rank gets its fan-out or every rank gets NULL (kq_last_error says which side it was on).
 *                 What this cannot cover: ncclCommInitRank itself failing, or librccl missing, on a SUBSET of the ranks --
 *                 the others then wait inside RCCL, and only the caller's own timeout ends that.
 *                 KQ_RCCL_LIB in the environment names the librccl to dlopen.
 *                 kq_shard_range(total_channels, world, rank, &first, &count) -> add channels first .. first+count-1
 *   per batch k:  slot = k & 1
 *                 kq_fanout_post(f, slot, iq, n, is_device)      queue the broadcast (iq: read on the root rank only)
 *                 p = kq_fanout_acquire(f, slot, bank_stream, &n)                  bank_stream waits for the batch
 *                 kq_bank_process_resident(bank, p, nblocks)                       (window layout: M-1 history + blocks)
 *                 kq_fanout_release(f, slot, bank_stream)                          the slot may be overwritten after this
 * `bank_stream` is the hipStream_t the bank was created on (kq_bank_config.stream); it must outlive the fan-out's next
 * kq_fanout_post of that slot (a world of one without a communicator records the release on it only then, and only if
 * that post copies something into the slot). */
#define KQ_FANOUT_ID_BYTES 128
typedef struct kq_fanout kq_fanout;
/* contiguous, balanced range of `rank`: the first total % world ranks hold one channel more */
int kq_shard_range(unsigned total, unsigned world, unsigned rank, unsigned *first, unsigned *count);
int kq_fanout_unique_id(void *id128);
kq_fanout *kq_fanout_create(int device, int rank, int world, int root, const void *id128, size_t max_samples);
int kq_fanout_destroy(kq_fanout *f);
int kq_fanout_post(kq_fanout *f, int slot, const void *iq_cf32, size_t nsamples, int src_is_device);
const void *kq_fanout_acquire(kq_fanout *f, int slot, void *consumer_stream, size_t *nsamples);
int kq_fanout_release(kq_fanout *f, int slot, void *consumer_stream);
/* what the fan-out has done so far: the world as RCCL itself counts it (ncclCommCount; 0 without a communicator), the
 * library's version code (ncclGetVersion) and the time the broadcasts took on the side stream (HIP events around
 * ncclBroadcast; a broadcast still in flight when its slot was posted again is not counted).  Waits for the side stream. */
typedef struct kq_fanout_info {
  int world, rank;
  int rccl_ranks, rccl_version;
  unsigned long long broadcasts;
  double broadcast_ms;
  /* consumer side: kq_fanout_acquire calls so far; of those, the ones that found their batch not yet landed while
   * kq_fanout_enable_timing was on (`waits`), and the time the consumer stream then stood still for it (`wait_ms`, HIP
   * events on that stream around the wait: a stalled step shows here, a slow broadcast that still arrives in time does not) */
  unsigned long long acquires, waits;
  double wait_ms;
  /* acquires that had to wait but went untimed because the slot's previous timed wait was still in flight (the host several
   * steps ahead of a stalled device): wait_ms / waits describe the timed ones only */
  unsigned long long waits_dropped;
} kq_fanout_info;
int kq_fanout_stats(kq_fanout *f, kq_fanout_info *out);
/* Path of the shared object ncclBroadcast was bound from (dladdr), "" before librccl has been loaded or when it could not
 * be: a process that also holds torch has torch's own copy of librccl mapped, and the dynamic loader hands this library
 * whichever `librccl.so.1` it finds first -- one line of a bench or log says which one carried the broadcasts. */
const char *kq_fanout_rccl_path(void);
/* on != 0: time the consumer's waits (two event records on the consumer stream per acquire that has to wait, ~5 us each
 * behind a long kernel -- a diagnostic, off by default). */
int kq_fanout_enable_timing(kq_fanout *f, int on);

/* --- front-end half-band decimator cascade ---------------------------------------------------
 * What hackrf.c:260-330 does to every block of raw A/D samples before they reach the channel filter: rotate by
 * +offset*Fs/4 (hackrf.c:272-291), run log_decimate half-band stages -- hb3_block (decimate.c:146-160) while the
 * stage index j >= stage_threshold, hb15_block (decimate.c:108-144) below it (hackrf.c:295-300) -- multiply by
 * Filter_atten (hackrf.c:469), convert to int16 (hackrf.c:307-311) and sum the output energy.  The filter state
 * (struct hb15_state / hb3state, hackrf.c:211-216) is carried inside the handle across calls. */
typedef struct kq_decimator kq_decimator;
typedef struct kq_decim_config {
  int device;
  int log_decimate;     /* hackrf.c:64 Log_decimate: decimation ratio is 2^log_decimate */
  int stage_threshold;  /* hackrf.c:76: stages j >= this use the 1-2-1 filter */
  int offset;           /* hackrf.c:68 Offset: rotate_phase increment per input sample, 0 = no rotation */
  float filter_atten;   /* hackrf.c:65,469 Filter_atten; 0 selects 0.5^log_decimate */
  size_t max_out;       /* largest n_out of one process call */
  void *stream;         /* hipStream_t to run on, NULL = private stream */
} kq_decim_config;
kq_decimator *kq_decim_create(const kq_decim_config *cfg);
int kq_decim_destroy(kq_decimator *d);
/* hb15 coefficients, order as struct hb15_state.coeffs (decimate.h:5); default is hackrf.c:229-238 */
int kq_decim_set_coeffs(kq_decimator *d, const float coeffs[4]);
/* iq_in: n_out << log_decimate interleaved complex float samples.  out_cf32: n_out complex samples after
 * Filter_atten; out_s16 (may be NULL): the same as (short)round(32767*s), interleaved I,Q; out_energy (may be
 * NULL): sum of s*s over the call (hackrf.c:308,325 output_energy), added up inside the last kernel of the call in an
 * order fixed by the device (same input, same device: same bits).  on_device != 0: every pointer is device
 * memory and the call is asynchronous on the handle's stream; otherwise host memory, synchronous.
 * kq_decim_sync (and a host-memory call) returns -1 if a workgroup's share of the energy never arrived (a device fault;
 * that call's energy is NaN, its samples are unaffected). */
int kq_decim_process(kq_decimator *d, const float *iq_in, int on_device, size_t n_out, float *out_cf32,
                     int16_t *out_s16, float *out_energy);
int kq_decim_sync(kq_decimator *d);
int kq_decim_reset(kq_decimator *d);

/* --- AFSK-1200 / HDLC packet decoder -----------------------------------------------------------
 * What the `packet` program does per RTP session between its PCM input and a decoded AX.25 frame: the REAL master
 * filter of 1000 new samples / 1049 taps (packet.c:41-45,190), the analytic 100..4000 Hz slave (packet.c:272-273),
 * the mark / space correlators with on-time and mid-bit integrators, the Gardner-style bit clock, NRZI and HDLC
 * deframing (packet.c:276-410) and the frame check (ax25.c:138-156 crc_good).  Constants are the reference's: 48 kHz
 * audio, 1200 bit/s, 40 samples per bit.  A bank holds `max_sessions` independent sessions that advance in lock step;
 * decoded frames (body + the two FCS bytes, exactly what packet.c:356 sends) collect in a per-session arena. */
typedef struct kq_afsk_bank kq_afsk_bank;
enum kq_pcm_format {
  KQ_PCM_F32 = 0,   /* float samples, what packet.c:207 stores into input.r[] */
  KQ_PCM_S16BE = 1  /* 16-bit big-endian PCM words as they arrive in the RTP payload, converted as packet.c:207 does:
                       ntohs() is unsigned, so a negative word w becomes (w + 65536) / 32768 (quirk kept) */
};
typedef struct kq_afsk_config {
  int device;
  unsigned max_sessions;
  unsigned max_frames;   /* arena slots per session between kq_afsk_clear_frames calls */
  void *stream;          /* hipStream_t, NULL = private stream */
} kq_afsk_config;
typedef struct kq_afsk_state {   /* packet.c:286-299 after the last processed block */
  int symphase, frame_bit, flagsync, ones;
  float last_val, mid_val;
  int decoded_packets;           /* packet.c:34 */
  int pending_samples;           /* packet.c:31 input_pointer */
  uint64_t blocks;
} kq_afsk_state;
kq_afsk_bank *kq_afsk_create(const kq_afsk_config *cfg);
int kq_afsk_destroy(kq_afsk_bank *bank);
/* Append `nsamples` samples to every session (session i at samples + i * session_stride elements) and decode every
 * block of 1000 that completes (packet.c:204-211).  Returns the number of blocks decoded per session, or -1.
 * on_device != 0: `samples` is device memory and the call is asynchronous on the bank's stream. */
int kq_afsk_push(kq_afsk_bank *bank, const void *samples, int format, unsigned nsessions, size_t nsamples,
                 size_t session_stride, int on_device);
int kq_afsk_sync(kq_afsk_bank *bank);
int kq_afsk_num_frames(kq_afsk_bank *bank, unsigned session);
int kq_afsk_dropped_frames(kq_afsk_bank *bank, unsigned session);   /* good frames lost to a full arena */
/* Copies frame `index` of `session` (at most cap bytes) and returns its length (<= 1024), or -1 */
int kq_afsk_pull_frame(kq_afsk_bank *bank, unsigned session, unsigned index, unsigned char *dst, size_t cap);
int kq_afsk_clear_frames(kq_afsk_bank *bank);
/* filter.out->output.c of the last decoded block (1000 complex) and the decoder state, for parity checks */
int kq_afsk_pull_filter_output(kq_afsk_bank *bank, unsigned session, float *dst_re_im, size_t cap_complex);
int kq_afsk_pull_state(kq_afsk_bank *bank, unsigned session, kq_afsk_state *out);

#ifdef __cplusplus
}
#endif
#endif
