/* kq_oracle.h -- CPU restatement ("oracle") of the ka9q-radio per-channel DSP hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped library (libka9q_hip.so)
 * never links or calls anything in oracle/.
 *
 * Every function restates, in fresh single-threaded C, the algorithm of the reference
 * function whose file:line is cited next to it (paths relative to the reference tree).
 * The reference's FFTs come from FFTW3 single precision (un-vendored dependency,
 * libfftw3f >= 3.3.5, INSTALLING.md:12-24); here they are replaced by the plain
 * power-of-two float FFT in kq_fft.c (same unnormalised forward -1 / backward +1
 * convention as FFTW's fftwf_plan_dft_1d / r2c / c2r).
 *
 * PINNING STATUS
 *   osc/dsp (NCO)   : pinned against the reference itself -- oracle/_ref/libref_osc.so is
 *                     built from /root/reference/{osc.c,dsp.c} unmodified (oracle/Makefile).
 *   decimate.c      : pinned likewise (oracle/_ref/libref_decimate.so).
 *   ax25.c crc_good : pinned likewise (oracle/_ref/libref_ax25.so).
 *   filter/radio/fm/am/linear/packet : PARITY UNPINNED.  Those reference files include <fftw3.h>
 *                     (filter.h:12) which this image lacks, and the reference ships no tests,
 *                     golden vectors or fixtures.  The restatement is
 *                     instead cross-checked against independent float64 numpy/scipy
 *                     formulations (direct convolution, scipy.signal.windows.kaiser, ...).
 */
#ifndef KQ_ORACLE_H
#define KQ_ORACLE_H 1

#include <complex.h>
#include <stdint.h>

#ifdef __cplusplus
#error "oracle is plain C"
#endif

/* ---------- FFT (stands in for FFTW3f; kq_fft.c) ---------- */
typedef struct kqo_fft kqo_fft;
kqo_fft *kqo_fft_create(unsigned n);           /* n = 2^a 3^b 5^c 7^d >= 1 (NULL otherwise) */
int kqo_fft_size_ok(unsigned n);               /* even and of that form */
void kqo_fft_destroy(kqo_fft *p);
/* 1: kqo_fft_c2c (and r2c / c2r through it) use the radix-4 autosort variant -- CPU-baseline timing only
 * (tests/test_oracle_filter.py::test_fast_transform_equals_the_plain_one).  A plan carries its own scratch buffers, which
 * both variants write through the const handle: one plan per thread -- a plan shared between threads races. */
void kqo_fft_set_fast(int on);
/* out-of-place or in-place (in == out) complex transform; sign -1 forward, +1 backward */
void kqo_fft_c2c(const kqo_fft *p, const float complex *in, float complex *out, int sign);
/* real -> n/2+1 complex bins (forward) */
void kqo_fft_r2c(const kqo_fft *p, const float *in, float complex *out);
/* n/2+1 complex bins -> n reals (backward, unnormalised) */
void kqo_fft_c2r(const kqo_fft *p, const float complex *in, float *out);

/* ---------- NCO: osc.c:14-59, dsp.c:38-50 ---------- */
typedef struct {
  double freq;                 /* cycles/sample */
  double rate;                 /* cycles/sample^2 */
  double complex phasor;
  double complex phasor_step;
  double complex phasor_step_step;
  int steps;                   /* since last renormalisation */
} kqo_osc;
#define KQO_RENORM_RATE 16384  /* osc.c:11 */
int kqo_is_phasor_init(double complex x);
void kqo_set_osc(kqo_osc *o, double f, double r);
double complex kqo_step_osc(kqo_osc *o);
void kqo_renorm_osc(kqo_osc *o);

/* ---------- Fast-convolution filter: filter.c:54-546 ---------- */
enum kqo_ftype { KQO_NONE = 0, KQO_COMPLEX = 1, KQO_CROSS_CONJ = 2, KQO_REAL = 3 }; /* filter.h:17-22 */

typedef struct kqo_filter_in {
  int in_type;
  unsigned ilen;               /* L */
  unsigned impulse_length;     /* M */
  unsigned n;                  /* L+M-1 */
  float complex *fdomain;      /* n (complex in) or n/2+1 (real in) bins */
  float complex *inbuf_c;      /* n samples, user area at +M-1 */
  float *inbuf_r;
  float complex *input_c;      /* = inbuf_c + M-1 */
  float *input_r;
  unsigned blocknum;
  kqo_fft *plan;
} kqo_filter_in;

typedef struct kqo_filter_out {
  kqo_filter_in *master;
  int out_type;
  float complex *response;     /* n_dec (or n_dec/2+1 for real out) bins; owned */
  float complex *f_fdomain;
  float noise_gain;
  float complex *outbuf_c;     /* n_dec */
  float *outbuf_r;
  float complex *output_c;     /* last olen of outbuf */
  float *output_r;
  unsigned decimate, olen, n_dec;
  unsigned blocknum;
  kqo_fft *plan;
} kqo_filter_out;

kqo_filter_in *kqo_create_filter_input(unsigned L, unsigned M, int in_type);
kqo_filter_out *kqo_create_filter_output(kqo_filter_in *m, float complex *response, unsigned decimate, int out_type);
int kqo_execute_filter_input(kqo_filter_in *m);
int kqo_execute_filter_output(kqo_filter_out *s);
int kqo_delete_filter_input(kqo_filter_in *m);
int kqo_delete_filter_output(kqo_filter_out *s);
int kqo_make_kaiser(float *window, unsigned M, float beta);
int kqo_window_filter(int L, int M, float complex *response, float beta);
int kqo_window_rfilter(int L, int M, float complex *response, float beta);
int kqo_set_filter(kqo_filter_out *s, float low, float high, float beta);
float kqo_noise_gain(const kqo_filter_out *s);

/* experimental IIR complex notch (filter.h:95-105, filter.c:549-571) */
typedef struct {
  double complex osc_phase, osc_step;
  float complex dcstate;
  float bw;
} kqo_notch;
kqo_notch *kqo_notch_create(double f, float bw);
float complex kqo_notch_step(kqo_notch *nf, float complex s);
void kqo_notch_run(kqo_notch *nf, const float complex *in, float complex *out, int n);

/* ---------- One receiver channel: radio.c:41-150,383-425; fm.c; am.c; linear.c ---------- */
enum kqo_demod { KQO_LINEAR = 0, KQO_AM = 1, KQO_FM = 2 };   /* radio.h:20-24 */

typedef struct {
  int samprate;        /* demod->input.samprate */
  unsigned L, M, D;    /* demod->filter.{L,M,decimate} */
  int demod_type;      /* enum kqo_demod */
  int flat;            /* opt.flat (FM) */
  int isb;             /* filter.isb (linear, CROSS_CONJ) */
  int channels;        /* output.channels 1|2 (linear) */
  float low, high;     /* Hz */
  float kaiser_beta;
  float headroom;      /* agc.headroom (amplitude ratio) */
  float hangtime;      /* s */
  float recovery_rate; /* dB/s */
  float gain_factor;   /* sdr.gain_factor */
  double lo2_hz;       /* set_second_LO argument (Hz; 0 = frozen) */
  double doppler_hz;   /* set_doppler arguments */
  double doppler_rate; /* Hz/s */
  double shift_hz;     /* set_shift argument */
  int compute_n0;      /* 0: skip the status-only noise estimate */
  int pll;             /* opt.pll: carrier tracking in linear mode (linear.c:129-246) */
  int square;          /* opt.square: squaring loop for suppressed-carrier DSB / BPSK */
} kqo_chan_cfg;

typedef struct {
  float if_power, bb_power, n0, snr, foffset, pdeviation, agc_gain;
  float plfreq;        /* fm.c:189-285 CTCSS tone estimate; NaN when none / not FM / geometry too small */
  float cphase;        /* linear.c:219-223 carrier phase of the block (PLL modes) */
  int pll_lock;        /* linear.c:162-169 */
  int lock_count;      /* linear.c:157-170 (sig.lock_timer) */
  int squelch_count;   /* fm.c snr_below_threshold */
  int hangcount;       /* am.c / linear.c hangcount */
  int blanked;         /* FM samples replaced by lastaudio this block */
  int nout;            /* floats written to audio this block */
  long long samples;   /* input samples consumed so far */
} kqo_status;

typedef struct kqo_chan kqo_chan;
kqo_chan *kqo_chan_create(const kqo_chan_cfg *cfg);
void kqo_chan_destroy(kqo_chan *c);
/* Retune while running (phase continuous, osc.c:24-27) */
void kqo_chan_set_lo2(kqo_chan *c, double lo2_hz);
void kqo_chan_set_doppler(kqo_chan *c, double hz, double rate);
/* set_mode (radio.c:322-374): fresh demodulator thread with the new mode, oscillators and sig.n0 carried over */
int kqo_chan_set_mode(kqo_chan *c, const kqo_chan_cfg *mode);
void kqo_chan_set_shift(kqo_chan *c, double shift_hz);                       /* radio.c:304-311 */
void kqo_chan_set_filter(kqo_chan *c, float low, float high, float beta);    /* display.c:161-177 */
/* Feed exactly L complex-float samples (re,im interleaved), run the whole chain for one block.
 * audio: olen*channels floats.  filt (optional): olen complex pre-detection filter outputs,
 * captured right after execute_filter_output (i.e. before linear.c:280 scales them in place).
 * spectrum (optional): N master bins.  Returns 0. */
int kqo_chan_block(kqo_chan *c, const float *iq, float *audio, kqo_status *st,
                   float *filt, float *spectrum);
/* int16 / int8 interleaved I/Q ingest (radio.c:110-122) -> same as above */
int kqo_chan_block_i16(kqo_chan *c, const int16_t *iq, float *audio, kqo_status *st);
int kqo_chan_block_i8(kqo_chan *c, const int8_t *iq, float *audio, kqo_status *st);
/* One packet payload of `count` int16 / int8 I/Q samples through proc_samples' loop (radio.c:104-147); returns the
 * number of blocks completed (each demodulated; audio / st hold count/L + 1 blocks) */
int kqo_chan_push_raw(kqo_chan *c, const void *iq, int count, int fmt, float *audio, kqo_status *st);
/* Lost-sample zero fill (radio.c:81-100): inject `count` zero samples, LOs keep running. Any
 * blocks completed meanwhile are demodulated; audio must hold ceil(count/L)+1 blocks.
 * Returns number of blocks completed. */
int kqo_chan_zero_fill(kqo_chan *c, int count, float *audio, kqo_status *st);
/* test helper for channels that join a running master (see kq_chan.c); iq: M-1 raw samples */
void kqo_chan_prime_history(kqo_chan *c, const float *iq);
unsigned kqo_chan_olen(const kqo_chan *c);
float kqo_chan_noise_gain(const kqo_chan *c);
const float complex *kqo_chan_response(const kqo_chan *c, unsigned *n);
const float complex *kqo_chan_audio_response(const kqo_chan *c, unsigned *n);

/* Half-band decimators of the front-end daemons (decimate.c:108-160, decimate.h:4-9) */
typedef struct {
  float coeffs[4];
  float even_samples[4];
  float odd_samples[4];
  float old_odd_samples[4];
} kqo_hb15_state;
void kqo_hb15_init(kqo_hb15_state *st);
void kqo_hb15_block(kqo_hb15_state *st, float *output, const float *input, int cnt);
void kqo_hb3_block(float *state, float *output, const float *input, int cnt);

/* I/Q packet ingest: RTP header, payload types, sequence / timestamp rules (multicast.c:242-277, 305-340;
 * main.c:315-341; radio.c:62-104) */
enum { KQO_IQ_S16 = 1, KQO_IQ_S8 = 2 };
typedef struct {
  int version;
  uint8_t type;
  uint16_t seq;
  uint32_t timestamp;
  uint32_t ssrc;
  int marker, pad, extension, cc;
} kqo_rtp_header;
typedef struct {                     /* multicast.h:41-52 */
  uint32_t ssrc;
  int init;
  uint16_t seq;
  uint32_t timestamp;
  long long packets, drops, dupes;
} kqo_rtp_state;
typedef struct {
  kqo_rtp_state rtp;
  long long samples;                 /* demod->input.samples */
} kqo_iq_ingest;
int kqo_ntoh_rtp(kqo_rtp_header *rtp, const unsigned char *data);
int kqo_rtp_process(kqo_rtp_state *state, const kqo_rtp_header *rtp, int sampcnt);
int kqo_iq_packet(kqo_iq_ingest *in, const unsigned char *packet, int size, int *zeros, int *offset, int *count, int *format);

/* AFSK-1200 / HDLC packet decoder (packet.c:36-48, 201-212, 267-414; ax25.c:138-156) */
#define KQO_AFSK_AL 1000        /* packet.c:42 */
#define KQO_AFSK_AM 1049        /* packet.c:44 */
#define KQO_AFSK_SAMPRATE 48000.f
#define KQO_AFSK_SAMPPBIT 40    /* packet.c:48 */
#define KQO_AFSK_FRAME_MAX 1024 /* packet.c:294 hdlc_frame[] */
typedef struct kqo_afsk kqo_afsk;
int kqo_crc_good(const unsigned char *frame, int length);
kqo_afsk *kqo_afsk_create(void);
void kqo_afsk_destroy(kqo_afsk *a);
void kqo_afsk_push(kqo_afsk *a, const float *samples, int n);
void kqo_afsk_push_pcm_be(kqo_afsk *a, const unsigned char *be, int nwords);
int kqo_afsk_nframes(const kqo_afsk *a);
int kqo_afsk_frame(const kqo_afsk *a, int i, unsigned char *dst, int cap);
const float complex *kqo_afsk_filter_output(const kqo_afsk *a);
void kqo_afsk_state(const kqo_afsk *a, int *symphase, int *frame_bit, int *flagsync, int *ones, float *last_val,
                    float *mid_val);

/* PCM output stage (audio.c:22-28, 45-50, 95-100): float -> clipped int16, network byte order, in chunks of at
 * most 480 words; bit i of *silent_mask is set when chunk i is all zero (the reference then skips the packet but
 * still advances the RTP timestamp).  Returns the number of chunks. */
int kqo_pcm_block(const float *audio, int nwords, int16_t *pcm_be, uint32_t *silent_mask);
/* send_mono_output / send_stereo_output as datagram builders (audio.c:32-132); state = demod->output.{rtp,silent} */
typedef struct {
  uint32_t ssrc;
  uint16_t seq;
  uint32_t timestamp;
  int silent;
  long long packets, bytes;
} kqo_out_rtp;
int kqo_pcm_rtp(kqo_out_rtp *o, const float *audio, int nfloats, int stereo, unsigned char *dst, int cap, int *used);

/* compute_n0 on a bare spectrum (radio.c:383-425) */
float kqo_compute_n0(const float complex *fdomain, unsigned N, int samprate, float low, float high);

/* Multi-channel CPU baseline (used only by bench.py cpu_baseline): nchan channels over `nthreads` pthreads; set-up and
 * `warm` blocks per channel untimed, then `timed` blocks per channel between two barriers; the nblocks_avail blocks of
 * `iq` are cycled.  fast_fft: time with the radix-4 autosort transform instead of the parity path's radix-2 one.
 * Returns the timed wall seconds. */
double kqo_bench_channels(const kqo_chan_cfg *cfgs, int nchan, const float *iq, int nblocks_avail, int warm, int timed,
                          int nthreads, int fast_fft, double *checksum);

#endif
