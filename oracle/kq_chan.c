/* kq_chan.c -- oracle restatement of one `radio` receiver channel, block at a time
 * (test infrastructure only; PARITY UNPINNED for everything but the NCO, see kq_oracle.h).
 *
 * The reference runs this as three threads (procsamp / demod / pl); the oracle runs the same
 * arithmetic in program order for one block of L input samples:
 *   ingest + mix          radio.c:106-147   (zero fill: radio.c:81-100)
 *   noise estimate        radio.c:383-425
 *   FM                    fm.c:21-174, PL-tone measurement fm.c:189-285 (run in lockstep after each block)
 *   AM                    am.c:15-83
 *   linear                linear.c:21-322 incl. the carrier PLL / squaring loop (linear.c:129-246)
 *   oscillator setters    radio.c:180-184, 290-311
 */
#define _GNU_SOURCE 1
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <limits.h>
#include <pthread.h>
#include <time.h>
#include "kq_oracle.h"

#define DB2VOLTAGE(x) (powf(10., (x) / 20.))   /* dsp.h:38 */
#define M_1_2PI (0.5 * M_1_PI)                 /* dsp.h:11 */
#define MAXF(x, y) ((x) > (y) ? (x) : (y))     /* misc.h:17-21: a NaN second argument propagates */

struct kqo_chan {
  kqo_chan_cfg cfg;
  kqo_filter_in *master;
  kqo_filter_out *slave;
  kqo_osc second_lo, doppler, shift;
  /* ingest state (radio.c:47-48) */
  float block_energy;
  int in_cnt;
  long long samples;
  float if_power;
  /* signal status (radio.h:164-175) */
  float bb_power, n0, snr, foffset, pdeviation, agc_gain;
  /* FM (fm.c:26,39-70) */
  float complex fm_state;
  float lastaudio;
  int snr_below_threshold;
  kqo_filter_in *audio_master;
  kqo_filter_out *audio_filter;
  int blanked;
  /* PL tone measurement (fm.c:189-285) */
  kqo_filter_out *pl_filter;
  float *pl_input;
  float complex *pl_spectrum;
  kqo_fft *pl_plan;
  int pl_fft_ptr, pl_last_fft;
  float pl_samprate, plfreq;
  /* AM / linear (am.c:26-34, linear.c:33-39) */
  int hangcount, hangmax;
  float recovery_factor;
  float dc_filter;
  float samptime, dsamprate;
  /* linear carrier PLL (linear.c:26-112) */
  kqo_osc pll_fine, pll_coarse;
  float pll_integrator, pll_delta_f, pll_ramp, cphase;
  int pll_lock_count, pll_lock, pll_fft_samples, pll_fft_ptr;
  float complex *pll_fftin, *pll_fftout;
  kqo_fft *pll_plan;
  float *cap_filt;             /* test hook: pre-detection filter output captured right after the slave runs */
};

static inline float norm2f(float complex z){
  return crealf(z) * crealf(z) + cimagf(z) * cimagf(z);
}

/* radio.c:290-301 */
void kqo_chan_set_lo2(kqo_chan *c, double lo2_hz){
  if(lo2_hz == 0)
    kqo_set_osc(&c->second_lo, 0.0, 0.0);
  else
    kqo_set_osc(&c->second_lo, lo2_hz / c->cfg.samprate, 0.0);
}
/* radio.c:180-184 */
void kqo_chan_set_doppler(kqo_chan *c, double hz, double rate){
  double const fs = c->cfg.samprate;
  kqo_set_osc(&c->doppler, -hz / fs, -rate / (fs * fs));
}
/* radio.c:304-311 */
static void chan_set_shift(kqo_chan *c, double shift){
  if(shift == 0)
    kqo_set_osc(&c->shift, 0.0, 0.0);
  else
    kqo_set_osc(&c->shift, shift * c->cfg.D / (double)c->cfg.samprate, 0.0);
}

void kqo_chan_set_shift(kqo_chan *c, double shift_hz){                    /* radio.c:304-311 */
  c->cfg.shift_hz = shift_hz;
  chan_set_shift(c, shift_hz);
}

/* Run-time filter change as the UI does it (display.c:161-177): new edges into demod->filter.{low,high,kaiser_beta},
 * then set_filter(filter.out, samptime*low, samptime*high, beta) -- samptime scaling whatever the mode.  The new
 * response is picked up by the next execute_filter_output (filter.c:538-543). */
void kqo_chan_set_filter(kqo_chan *c, float low, float high, float beta){
  c->cfg.low = low;
  c->cfg.high = high;
  c->cfg.kaiser_beta = beta;
  kqo_set_filter(c->slave, c->samptime * low, c->samptime * high, beta);
}

/* radio.c:383-425.  The reference forms n*samprate in int (radio.c:407,409), which overflows
 * for N/2*samprate >= 2^31 (all the multi-MS/s configs); the oracle reproduces the wrapped
 * 32-bit product that gcc emits, so passband exclusion matches the compiled reference. */
float kqo_compute_n0(const float complex *fdomain, unsigned N, int samprate, float low, float high){
  int const n_bins = (int)N;
  float *power = malloc(sizeof(float) * N);
  for(int n = 0; n < n_bins; n++)
    power[n] = norm2f(fdomain[n]);
  float avg = INFINITY;
  for(int iter = 0; iter < 2; iter++){
    int bins = 0;
    float acc = 0;
    for(int n = 0; n < n_bins; n++){
      int32_t const k = (n <= n_bins / 2) ? n : n - n_bins;
      int32_t const prod = (int32_t)((uint32_t)k * (uint32_t)samprate);   /* wraps like the int product */
      float const f = (float)prod / n_bins;
      if(f >= low && f <= high)
        continue;
      float const s = power[n];
      if(s < avg * 2){
        acc += s;
        bins++;
      }
    }
    avg = acc / bins;
  }
  free(power);
  return avg / (2.0 * N * samprate);
}

/* What a demodulator thread owns and builds in its prologue (fm.c:21-70, am.c:15-41, linear.c:21-112): the slave
 * with its response, the audio / PL filters, and the thread-local state.  What lives in struct demod (oscillators,
 * sig.n0, sig.foffset, sig.pdeviation) is not touched here. */
static int demod_start(kqo_chan *c){
  kqo_chan_cfg const *cfg = &c->cfg;
  int out_type = KQO_COMPLEX;
  if(cfg->demod_type == KQO_LINEAR && cfg->isb)
    out_type = KQO_CROSS_CONJ;                                             /* linear.c:78-79 */
  c->slave = kqo_create_filter_output(c->master, NULL, cfg->D, out_type);
  if(!c->slave)
    return -1;
  switch(cfg->demod_type){
  case KQO_FM:{
    kqo_set_filter(c->slave, cfg->low / c->dsamprate, cfg->high / c->dsamprate, cfg->kaiser_beta); /* fm.c:35 */
    int const AL = cfg->L / cfg->D;                                        /* fm.c:39-41 */
    int const AM = (cfg->M - 1) / cfg->D + 1;
    int const AN = AL + AM - 1;
    float const filter_gain = 10. / AN;                                    /* fm.c:42 */
    c->audio_master = kqo_create_filter_input(AL, AM, KQO_REAL);
    if(!cfg->flat){
      float complex *ar = calloc(AN / 2 + 1, sizeof(float complex));
      for(int j = 0; j <= AN / 2; j++){                                    /* fm.c:59-63 */
        float const f = (float)j * c->dsamprate / AN;
        if(f >= 300 && f <= 6000)
          ar[j] = filter_gain * 300. / f;
      }
      kqo_window_rfilter(AL, AM, ar, cfg->kaiser_beta);
      c->audio_filter = kqo_create_filter_output(c->audio_master, ar, 1, KQO_REAL);
    }
    {
      /* pltask set-up, fm.c:196-229.  PL_N = AN/32 must leave a usable transform (the reference runs it
       * regardless; below 4 points it is meaningless and the oracle leaves plfreq NaN).  Where 32 does not divide AN or AL,
       * create_filter_output warns and truncates (filter.c:103-107,116) and so does kqo_create_filter_output; a PL_N this
       * oracle's FFT has no plan for (odd, or a prime factor beyond 7) leaves plfreq NaN. */
      int const PL_decimate = 32;
      int const PL_N = AN / PL_decimate, PL_L = AL / PL_decimate, PL_M = PL_N - PL_L + 1;
      c->plfreq = NAN;
      c->pl_fft_ptr = c->pl_last_fft = 0;
      if(PL_N >= 4 && PL_L >= 1 && ((PL_N & (PL_N - 1)) == 0 || kqo_fft_size_ok((unsigned)PL_N))){
        c->pl_samprate = c->dsamprate / PL_decimate;
        float complex *plr = calloc(PL_N / 2 + 1, sizeof(float complex));
        for(int j = 0; j <= PL_N / 2; j++){
          float const f = (float)j * c->dsamprate / AN;                    /* fm.c:214 */
          if(f > 0 && f < 300)
            plr[j] = 1;
        }
        kqo_window_rfilter(PL_L, PL_M, plr, 2.0);                          /* fm.c:218 */
        c->pl_filter = kqo_create_filter_output(c->audio_master, plr, PL_decimate, KQO_REAL);
        int const pl_fft_size = (1 << 19) / PL_decimate;                   /* fm.c:225 */
        c->pl_input = calloc(pl_fft_size, sizeof(float));
        c->pl_spectrum = calloc(pl_fft_size / 2 + 1, sizeof(float complex));
        c->pl_plan = kqo_fft_create(pl_fft_size);
      }
    }
    c->hangcount = 0;                                                      /* no such state in the FM thread */
    c->fm_state = 1;                                                       /* fm.c:26 */
    c->lastaudio = 0;                                                      /* fm.c:68-69 */
    c->snr_below_threshold = 0;
    break;
  }
  case KQO_AM:
    kqo_set_filter(c->slave, c->samptime * cfg->low, c->samptime * cfg->high, cfg->kaiser_beta); /* am.c:41 */
    c->hangcount = 0;                                                      /* am.c:26 */
    c->recovery_factor = DB2VOLTAGE(cfg->recovery_rate * c->samptime);     /* am.c:27 */
    c->hangmax = cfg->hangtime / c->samptime;                              /* am.c:29 */
    c->agc_gain = DB2VOLTAGE(80.);                                         /* am.c:30 */
    c->dc_filter = 0;                                                      /* am.c:33 */
    break;
  default:
    kqo_set_filter(c->slave, c->samptime * cfg->low, c->samptime * cfg->high, cfg->kaiser_beta); /* linear.c:81 */
    c->hangcount = 0;                                                      /* linear.c:33 */
    c->recovery_factor = DB2VOLTAGE(cfg->recovery_rate * c->samptime);     /* linear.c:34 */
    c->hangmax = cfg->hangtime / c->samptime;                              /* linear.c:38 */
    c->agc_gain = DB2VOLTAGE(100.0);                                       /* linear.c:39 */
    c->snr = 0;                                                            /* linear.c:75 */
    if(cfg->pll){
      c->pll_fftin = calloc(1 << 16, sizeof(float complex));               /* linear.c:89-93 */
      c->pll_fftout = calloc(1 << 16, sizeof(float complex));
      c->pll_plan = kqo_fft_create(1 << 16);
      memset(&c->pll_fine, 0, sizeof c->pll_fine);
      memset(&c->pll_coarse, 0, sizeof c->pll_coarse);
      c->pll_fine.phasor = 1;                                              /* linear.c:97-105 */
      kqo_set_osc(&c->pll_fine, 0.0, 0.0);
      c->pll_coarse.phasor = 1;
      kqo_set_osc(&c->pll_coarse, 0.0, 0.0);
      c->pll_integrator = c->pll_delta_f = c->pll_ramp = c->cphase = 0;
      c->pll_lock_count = c->pll_lock = c->pll_fft_samples = c->pll_fft_ptr = 0;
    }
    break;
  }
  return 0;
}

/* the thread's exit path (fm.c:175-183, am.c:80-82, linear.c:313-321) */
static void demod_stop(kqo_chan *c){
  free(c->pll_fftin);
  free(c->pll_fftout);
  kqo_fft_destroy(c->pll_plan);
  kqo_delete_filter_output(c->pl_filter);
  free(c->pl_input);
  free(c->pl_spectrum);
  kqo_fft_destroy(c->pl_plan);
  kqo_delete_filter_output(c->audio_filter);
  kqo_delete_filter_input(c->audio_master);
  kqo_delete_filter_output(c->slave);
  c->pll_fftin = c->pll_fftout = NULL;
  c->pll_plan = NULL;
  c->pl_filter = NULL;
  c->pl_input = NULL;
  c->pl_spectrum = NULL;
  c->pl_plan = NULL;
  c->audio_filter = NULL;
  c->audio_master = NULL;
  c->slave = NULL;
}

kqo_chan *kqo_chan_create(const kqo_chan_cfg *cfg){
  kqo_chan *c = calloc(1, sizeof(*c));
  c->cfg = *cfg;
  c->master = kqo_create_filter_input(cfg->L, cfg->M, KQO_COMPLEX);       /* main.c:232 */
  if(!c->master){
    free(c);
    return NULL;
  }
  c->n0 = NAN;
  c->snr = 0;
  c->samptime = (float)cfg->D / (float)cfg->samprate;                      /* am.c:21, linear.c:29 */
  c->dsamprate = (float)cfg->samprate / cfg->D;                            /* fm.c:27 */
  /* struct osc zero-initialised => phasor not initialised => first set_osc sets it to 1 */
  kqo_chan_set_lo2(c, cfg->lo2_hz);
  kqo_chan_set_doppler(c, cfg->doppler_hz, cfg->doppler_rate);
  chan_set_shift(c, cfg->shift_hz);
  c->pdeviation = 0;
  c->foffset = 0;
  if(demod_start(c) != 0){
    kqo_delete_filter_input(c->master);
    free(c);
    return NULL;
  }
  return c;
}

/* set_mode (radio.c:322-374): the running demodulator thread is joined and a fresh one started with the new mode's
 * parameters.  From `mode`: demod_type, flat, isb, channels, pll, square, recovery_rate, hangtime, low / high (swapped
 * when low > high, radio.c:343-349), shift (set_shift keeps the shift oscillator's phase), kaiser_beta, headroom.
 * The input oscillators keep running; sig.n0, sig.foffset and sig.pdeviation keep their values. */
int kqo_chan_set_mode(kqo_chan *c, const kqo_chan_cfg *mode){
  demod_stop(c);
  c->cfg.demod_type = mode->demod_type;
  if(mode->low > mode->high){
    c->cfg.low = mode->high;
    c->cfg.high = mode->low;
  } else {
    c->cfg.low = mode->low;
    c->cfg.high = mode->high;
  }
  c->cfg.shift_hz = mode->shift_hz;
  c->cfg.flat = mode->flat;
  c->cfg.isb = mode->isb;
  c->cfg.channels = mode->channels;
  c->cfg.pll = mode->pll;
  c->cfg.square = mode->square;
  c->cfg.recovery_rate = mode->recovery_rate;
  c->cfg.hangtime = mode->hangtime;
  c->cfg.kaiser_beta = mode->kaiser_beta;
  c->cfg.headroom = mode->headroom;
  chan_set_shift(c, c->cfg.shift_hz);                                      /* radio.c:367 */
  return demod_start(c);
}

void kqo_chan_destroy(kqo_chan *c){
  if(!c)
    return;
  demod_stop(c);
  kqo_delete_filter_input(c->master);
  free(c);
}

unsigned kqo_chan_olen(const kqo_chan *c){ return c->slave->olen; }
float kqo_chan_noise_gain(const kqo_chan *c){ return c->slave->noise_gain; }
const float complex *kqo_chan_response(const kqo_chan *c, unsigned *n){
  if(n)
    *n = c->slave->n_dec;
  return c->slave->response;
}
const float complex *kqo_chan_audio_response(const kqo_chan *c, unsigned *n){
  if(!c->audio_filter)
    return NULL;
  if(n)
    *n = c->audio_filter->n_dec / 2 + 1;
  return c->audio_filter->response;
}

/* --- demodulators, one block each --- */

static void capture_filt(kqo_chan *c){
  if(c->cap_filt)
    memcpy(c->cap_filt, c->slave->output_c, sizeof(float complex) * c->slave->olen);
}

/* rate is the double literal of the reference (.01 in fm.c:82, .001 in am.c:47 and linear.c:124): the update is
 * evaluated in double and rounded to float on assignment */
static void update_n0(kqo_chan *c, double rate){
  if(!c->cfg.compute_n0)
    return;
  float const fresh = kqo_compute_n0(c->master->fdomain, c->master->n, c->cfg.samprate, c->cfg.low, c->cfg.high);
  if(isnan(c->n0))
    c->n0 = fresh;                                                         /* fm.c:79-80, am.c:49 */
  else
    c->n0 += rate * (fresh - c->n0);                                       /* fm.c:82, am.c:47: float += double * float */
}

static int fm_block(kqo_chan *c, float *audio){
  kqo_filter_out *flt = c->slave;
  kqo_filter_in *am = c->audio_master;
  int const olen = (int)flt->olen;
  kqo_execute_filter_output(flt);
  capture_filt(c);
  update_n0(c, .01);

  float const gain = (c->cfg.headroom * M_1_PI * c->dsamprate) / fabsf(c->cfg.low - c->cfg.high); /* fm.c:86 */

  float avg_amp = 0;
  c->bb_power = 0;
  for(int n = 0; n < olen; n++){                                           /* fm.c:93-97 */
    float const t = norm2f(flt->output_c[n]);
    c->bb_power += t;
    avg_amp += sqrtf(t);
  }
  c->bb_power /= 2 * olen;
  avg_amp /= M_SQRT2 * olen;
  float const variance = c->bb_power - avg_amp * avg_amp;
  c->snr = avg_amp * avg_amp / (2 * variance) - 1;
  c->snr = MAXF(0.0f, c->snr);                                             /* fm.c:103 (misc.h max) */

  if(c->snr > 2){                                                          /* fm.c:108-114 */
    c->snr_below_threshold = 0;
  } else if(++c->snr_below_threshold > 1000){
    c->snr_below_threshold = 1000;
  }
  c->blanked = 0;
  if(c->snr_below_threshold < 2){
    float const min_ampl = 0.55 * 0.55 * avg_amp * avg_amp;                /* fm.c:121 */
    float pdev_pos = 0, pdev_neg = 0, avg_f = 0;
    for(int n = 0; n < olen; n++){                                         /* fm.c:128-144 */
      float complex const samp = flt->output_c[n];
      if(norm2f(samp) > min_ampl){
        c->lastaudio = audio[n] = am->input_r[n] = cargf(samp * c->fm_state);
        c->fm_state = conjf(samp);
        if(n == 0)
          pdev_pos = pdev_neg = c->lastaudio;
        else if(c->lastaudio > pdev_pos)
          pdev_pos = c->lastaudio;
        else if(c->lastaudio < pdev_neg)
          pdev_neg = c->lastaudio;
      } else {
        audio[n] = am->input_r[n] = c->lastaudio;
        c->blanked++;
      }
      avg_f += c->lastaudio;
    }
    avg_f /= olen;
    if(c->snr_below_threshold < 1){                                        /* fm.c:146-154 */
      c->foffset = c->dsamprate * avg_f * M_1_2PI;
      pdev_pos -= avg_f;
      pdev_neg -= avg_f;
      c->pdeviation = c->dsamprate * MAXF(pdev_pos, -pdev_neg) * M_1_2PI;
    }
  } else {
    c->fm_state = 0;                                                       /* fm.c:156-160 */
    c->lastaudio = 0;
    memset(audio, 0, sizeof(float) * am->ilen);
    memset(am->input_r, 0, sizeof(float) * am->ilen);
  }
  kqo_execute_filter_input(am);                                            /* fm.c:162 */
  if(c->pl_filter){
    /* one iteration of pltask's loop, fm.c:233-277 */
    int const pl_fft_size = (1 << 19) / 32;
    kqo_execute_filter_output(c->pl_filter);
    int remain = (int)c->pl_filter->olen;
    c->pl_last_fft += remain;
    float const *data = c->pl_filter->output_r;
    while(remain != 0){
      int chunk = pl_fft_size - c->pl_fft_ptr;
      if(chunk > remain)
        chunk = remain;
      memcpy(c->pl_input + c->pl_fft_ptr, data, sizeof(float) * chunk);
      c->pl_fft_ptr += chunk;
      data += chunk;
      remain -= chunk;
      if(c->pl_fft_ptr >= pl_fft_size)
        c->pl_fft_ptr -= pl_fft_size;
    }
    if(c->pl_last_fft >= 512){                                             /* fm.c:251 */
      c->pl_last_fft = 0;
      kqo_fft_r2c(c->pl_plan, c->pl_input, c->pl_spectrum);
      int peakbin = -1;
      float peakenergy = 0, totenergy = 0;
      for(int n = 1; n < pl_fft_size / 2; n++){                            /* fm.c:260-267 */
        float const energy = norm2f(c->pl_spectrum[n]);
        totenergy += energy;
        if(energy > peakenergy){
          peakenergy = energy;
          peakbin = n;
        }
      }
      if(peakbin > 0 && peakenergy > 0.01 * totenergy){                    /* fm.c:271-276 */
        float const f = (float)peakbin * c->pl_samprate / pl_fft_size;
        if(f > 67 && f < 255)
          c->plfreq = f;
      } else
        c->plfreq = NAN;
    }
  }
  if(c->audio_filter){
    kqo_execute_filter_output(c->audio_filter);
    for(int n = 0; n < (int)c->audio_filter->olen; n++)                    /* fm.c:169-170 */
      audio[n] = c->audio_filter->output_r[n] * gain;
  }
  return (int)am->ilen;
}

static int am_block(kqo_chan *c, float *audio){
  kqo_filter_out *flt = c->slave;
  int const olen = (int)flt->olen;
  kqo_execute_filter_output(flt);
  capture_filt(c);
  update_n0(c, .001);
  float signal = 0, noise = 0;
  float const dc_coeff = .0001;                                            /* am.c:34 */
  for(int n = 0; n < olen; n++){                                           /* am.c:55-75 */
    float const sq = norm2f(flt->output_c[n]);
    signal += sq;
    float const samp = sqrtf(sq);
    c->dc_filter += dc_coeff * (samp - c->dc_filter);
    if(isnan(c->agc_gain)){
      c->agc_gain = c->cfg.headroom / c->dc_filter;
    } else if(c->agc_gain * c->dc_filter > c->cfg.headroom){
      c->agc_gain = c->cfg.headroom / c->dc_filter;
      c->hangcount = c->hangmax;
    } else if(c->hangcount != 0){
      c->hangcount--;
    } else {
      c->agc_gain *= c->recovery_factor;
    }
    audio[n] = (samp - c->dc_filter) * c->agc_gain;
  }
  c->bb_power = (signal + noise) / (2 * olen);                             /* am.c:78 */
  return olen;
}

static int linear_block(kqo_chan *c, float *audio){
  kqo_filter_out *flt = c->slave;
  int const olen = (int)flt->olen;
  flt->out_type = c->cfg.isb ? KQO_CROSS_CONJ : KQO_COMPLEX;               /* linear.c:117-120 */
  kqo_execute_filter_output(flt);
  capture_filt(c);
  update_n0(c, .001);
  if(c->cfg.pll){                                                          /* linear.c:129-246 */
    float const samptime = c->samptime;
    float const blocktime = samptime * c->cfg.L;                           /* linear.c:30 (L is the undecimated block) */
    int const fftsize = 1 << 16;
    float const snrthresh = powf(10, 3. / 10);                             /* linear.c:42,49 */
    int const lock_limit = round(1 / samptime);                            /* linear.c:45,50 */
    float const binsize = 1. / (fftsize * samptime);
    int const sq = c->cfg.square ? 2 : 1;
    int const lowlimit = round(sq * -300.f / binsize);                     /* linear.c:53-56 */
    int const highlimit = round(sq * 300.f / binsize);
    float const vcogain = 2 * M_PI, pdgain = 1, damping = M_SQRT1_2;       /* linear.c:59-65 */
    float const natfreq = 1 * 2 * M_PI;                                    /* loop_bw = 1, linear.c:26 */
    float const tau1 = vcogain * pdgain / (natfreq * natfreq);
    float const integrator_gain = 1 / tau1;
    float const tau2 = 2 * damping / natfreq;
    float const prop_gain = tau2 / tau1;
    float const ramprate = 0;                                              /* linear.c:67 */

    c->pll_fft_samples += olen;                                            /* linear.c:132-152 */
    if(c->pll_fft_samples > fftsize)
      c->pll_fft_samples = fftsize;
    for(int i = 0; i < olen; i++){
      float complex const s = flt->output_c[i];
      c->pll_fftin[c->pll_fft_ptr++] = c->cfg.square ? s * s : s;
      if(c->pll_fft_ptr >= fftsize)
        c->pll_fft_ptr -= fftsize;
    }
    if(c->snr < snrthresh)                                                 /* linear.c:157-170: uses the PREVIOUS block's snr */
      c->pll_lock_count -= olen;
    else
      c->pll_lock_count += olen;
    if(c->pll_lock_count >= lock_limit){
      c->pll_lock_count = lock_limit;
      c->pll_lock = 1;
    }
    if(c->pll_lock_count <= -lock_limit){
      c->pll_lock_count = -lock_limit;
      c->pll_lock = 0;
    }
    if(!c->pll_lock){                                                      /* linear.c:173-203 */
      if(c->pll_fft_samples > fftsize / 2){
        c->pll_fft_samples = 0;
        kqo_fft_c2c(c->pll_plan, c->pll_fftin, c->pll_fftout, -1);
        int maxbin = 0;
        float maxenergy = 0;
        for(int n = lowlimit; n <= highlimit; n++){
          float const e = norm2f(c->pll_fftout[n < 0 ? n + fftsize : n]);
          if(e > maxenergy){
            maxenergy = e;
            maxbin = n;
          }
        }
        if(maxenergy > 0){
          double new_delta_f = binsize * maxbin;
          if(c->cfg.square)
            new_delta_f /= 2;
          if(new_delta_f != c->pll_delta_f){
            c->pll_delta_f = new_delta_f;
            c->pll_integrator = 0;
            kqo_set_osc(&c->pll_coarse, -samptime * c->pll_delta_f, 0.0);
          }
        }
      }
      if(c->pll_ramp == 0)
        c->pll_ramp = ramprate;
    } else {
      c->pll_ramp = 0;
    }
    float complex accum = 0;                                               /* linear.c:208-223 */
    for(int n = 0; n < olen; n++){
      flt->output_c[n] *= kqo_step_osc(&c->pll_coarse) * kqo_step_osc(&c->pll_fine);
      float complex ss = flt->output_c[n];
      if(c->cfg.square)
        ss *= ss;
      accum += ss;
    }
    c->cphase = cargf(accum);
    if(isnan(c->cphase))
      c->cphase = 0;
    if(c->cfg.square)
      c->cphase /= 2;
    float const carrier_phase = c->cphase;                                 /* linear.c:228-245 */
    c->pll_integrator += carrier_phase * blocktime + c->pll_ramp;
    float const feedback = integrator_gain * c->pll_integrator + prop_gain * carrier_phase;
    kqo_set_osc(&c->pll_fine, -feedback * samptime, 0.0);
    if((feedback >= binsize) && (c->pll_ramp > 0))
      c->pll_ramp = -ramprate;
    else if((feedback <= binsize) && (c->pll_ramp < 0))
      c->pll_ramp = ramprate;
    if(isnan(c->foffset))
      c->foffset = feedback + c->pll_delta_f;
    else
      c->foffset += 0.001 * (feedback + c->pll_delta_f - c->foffset);
  }
  float signal = 0, noise = 0;
  for(int n = 0; n < olen; n++){                                           /* linear.c:251-281 */
    float complex const s = flt->output_c[n];
    float const rp = crealf(s) * crealf(s);
    float const ip = cimagf(s) * cimagf(s);
    signal += rp;
    noise += ip;
    float const amplitude = sqrtf(rp + ip);
    if(isnan(c->agc_gain)){
      c->agc_gain = c->cfg.headroom / amplitude;
    } else if(amplitude * c->agc_gain > c->cfg.headroom){
      c->agc_gain = c->cfg.headroom / amplitude;
      c->hangcount = c->hangmax;
    } else if(c->hangcount != 0){
      c->hangcount--;
    } else {
      c->agc_gain *= c->recovery_factor;
    }
    flt->output_c[n] *= c->agc_gain;
  }
  if(c->shift.freq != 0){                                                  /* linear.c:283-289 */
    for(int n = 0; n < olen; n++)
      flt->output_c[n] *= kqo_step_osc(&c->shift);
  }
  int nout;
  if(c->cfg.channels == 1){                                                /* linear.c:291-300 */
    for(int n = 0; n < olen; n++)
      audio[n] = crealf(flt->output_c[n]);
    nout = olen;
  } else {
    memcpy(audio, flt->output_c, sizeof(float complex) * olen);
    nout = 2 * olen;
  }
  c->bb_power = (signal + noise) / (2 * olen);                             /* linear.c:302 */
  if(noise != 0 && c->cfg.pll){                                            /* linear.c:304-309 */
    c->snr = (signal / noise) - 1;
    if(c->snr < 0)
      c->snr = 0;
  } else
    c->snr = NAN;
  return nout;
}

static int demod_block(kqo_chan *c, float *audio, kqo_status *st, float *filt, float *spectrum){
  int nout;
  if(spectrum)
    memcpy(spectrum, c->master->fdomain, sizeof(float complex) * c->master->n);
  c->cap_filt = filt;
  switch(c->cfg.demod_type){
  case KQO_FM: nout = fm_block(c, audio); break;
  case KQO_AM: nout = am_block(c, audio); break;
  default:     nout = linear_block(c, audio); break;
  }
  c->cap_filt = NULL;
  if(st){
    st->if_power = c->if_power;
    st->bb_power = c->bb_power;
    st->n0 = c->n0;
    st->snr = c->snr;
    st->foffset = c->foffset;
    st->pdeviation = c->pdeviation;
    st->agc_gain = c->agc_gain;
    st->squelch_count = c->snr_below_threshold;
    st->hangcount = c->hangcount;
    st->blanked = c->blanked;
    st->plfreq = (c->cfg.demod_type == KQO_FM) ? c->plfreq : NAN;
    st->cphase = c->cphase;
    st->pll_lock = c->pll_lock;
    st->lock_count = c->pll_lock_count;
    st->nout = nout;
    st->samples = c->samples;
  }
  return nout;
}

/* One sample through radio.c:122-146; returns 1 when a block was completed */
static inline int ingest_sample(kqo_chan *c, float complex samp){
  c->block_energy += norm2f(samp);                                         /* radio.c:123 */
  samp *= kqo_step_osc(&c->second_lo);                                     /* radio.c:132: double product, rounded to float */
  if(c->doppler.freq != 0)
    samp *= kqo_step_osc(&c->doppler);                                     /* radio.c:135-136 */
  c->master->input_c[c->in_cnt++] = samp;
  if(c->in_cnt == (int)c->master->ilen){
    kqo_execute_filter_input(c->master);
    c->block_energy *= 0.5;                                                /* halved, never cleared: radio.c:143 */
    c->if_power = c->block_energy / c->in_cnt;
    c->in_cnt = 0;
    return 1;
  }
  return 0;
}

/* Not a reference function.  The reference runs one demodulator per master, created together (main.c:232); the GPU
 * bank shares one master among channels that come and go.  A channel that joins a running master sees the master's
 * M-1 history samples in its first block: this loads them (`iq`: M-1 raw samples, oldest first) as if this
 * channel's oscillators had been running before its creation and reach their initial phase at its first sample. */
void kqo_chan_prime_history(kqo_chan *c, const float *iq){
  int const h = (int)c->master->impulse_length - 1;
  for(int i = 0; i < h; i++){
    double const n = (double)(i - h);
    float complex samp = (iq[2 * i] + iq[2 * i + 1] * _Complex_I) * c->cfg.gain_factor;
    kqo_osc const *o = &c->second_lo;
    samp *= o->freq == 0 ? o->phasor : o->phasor * cexp(2 * M_PI * _Complex_I * (o->freq * n + o->rate * (0.5 * n * (n - 1.0))));
    o = &c->doppler;
    if(o->freq != 0)
      samp *= o->phasor * cexp(2 * M_PI * _Complex_I * (o->freq * n + o->rate * (0.5 * n * (n - 1.0))));
    c->master->inbuf_c[i] = samp;
  }
}

int kqo_chan_block(kqo_chan *c, const float *iq, float *audio, kqo_status *st, float *filt, float *spectrum){
  unsigned const L = c->master->ilen;
  int done = 0;
  c->samples += L;
  for(unsigned i = 0; i < L; i++){
    float complex const samp = (iq[2 * i] + iq[2 * i + 1] * _Complex_I) * c->cfg.gain_factor; /* radio.c:122 */
    done += ingest_sample(c, samp);
  }
  if(done)
    demod_block(c, audio, st, filt, spectrum);
  return 0;
}

int kqo_chan_block_i16(kqo_chan *c, const int16_t *iq, float *audio, kqo_status *st){
  float const scale16 = 1. / SHRT_MAX;                                     /* radio.c:38 */
  unsigned const L = c->master->ilen;
  int done = 0;
  c->samples += L;
  for(unsigned i = 0; i < L; i++){
    float const si = iq[2 * i] * scale16, sq = iq[2 * i + 1] * scale16;    /* radio.c:113-114 */
    done += ingest_sample(c, (si + sq * _Complex_I) * c->cfg.gain_factor);
  }
  if(done)
    demod_block(c, audio, st, NULL, NULL);
  return 0;
}

int kqo_chan_block_i8(kqo_chan *c, const int8_t *iq, float *audio, kqo_status *st){
  float const scale8 = 1. / 127;                                           /* radio.c:39 */
  unsigned const L = c->master->ilen;
  int done = 0;
  c->samples += L;
  for(unsigned i = 0; i < L; i++){
    float const si = iq[2 * i] * scale8, sq = iq[2 * i + 1] * scale8;      /* radio.c:117-118 */
    done += ingest_sample(c, (si + sq * _Complex_I) * c->cfg.gain_factor);
  }
  if(done)
    demod_block(c, audio, st, NULL, NULL);
  return 0;
}

/* proc_samples' per-sample loop over one packet payload of `count` samples (radio.c:104-147): fmt KQO_IQ_S16 or
 * KQO_IQ_S8.  Every block that completes is demodulated; audio / st must hold count/L + 1 blocks.  Returns the
 * number of blocks completed. */
int kqo_chan_push_raw(kqo_chan *c, const void *iq, int count, int fmt, float *audio, kqo_status *st){
  float const scale16 = 1. / SHRT_MAX, scale8 = 1. / 127;                  /* radio.c:38-39 */
  unsigned const olen_max = 2 * c->slave->olen;
  const int16_t *p16 = iq;
  const int8_t *p8 = iq;
  int blocks = 0;
  c->samples += count;                                                     /* radio.c:104 */
  for(int i = 0; i < count; i++){
    float si, sq;
    if(fmt == KQO_IQ_S8){
      si = p8[2 * i] * scale8;
      sq = p8[2 * i + 1] * scale8;
    } else {
      si = p16[2 * i] * scale16;
      sq = p16[2 * i + 1] * scale16;
    }
    if(ingest_sample(c, (si + sq * _Complex_I) * c->cfg.gain_factor)){
      demod_block(c, audio + (size_t)blocks * olen_max, st ? st + blocks : NULL, NULL, NULL);
      blocks++;
    }
  }
  return blocks;
}

int kqo_chan_zero_fill(kqo_chan *c, int count, float *audio, kqo_status *st){
  int blocks = 0;
  unsigned const olen_max = 2 * c->slave->olen;
  c->samples += count;                                                     /* radio.c:87 */
  for(int i = 0; i < count; i++){                                          /* radio.c:88-99 */
    c->master->input_c[c->in_cnt++] = 0;
    (void)kqo_step_osc(&c->second_lo);
    (void)kqo_step_osc(&c->doppler);
    if(c->in_cnt == (int)c->master->ilen){
      kqo_execute_filter_input(c->master);
      c->in_cnt = 0;
      demod_block(c, audio + (size_t)blocks * olen_max, st ? st + blocks : NULL, NULL, NULL);
      blocks++;
    }
  }
  return blocks;
}

/* ---- PCM output stage: audio.c:22-28 (scaleclip), 45-50 / 95-100 (htons, silence detection per 480-word chunk) ---- */
static short scaleclip(float x){
  if(x >= 1.0)
    return SHRT_MAX;
  else if(x <= -1.0)
    return SHRT_MIN;
  return (short)(SHRT_MAX * x);
}

int kqo_pcm_block(const float *audio, int nwords, int16_t *pcm_be, uint32_t *silent_mask){
  int chunks = 0;
  uint32_t mask = 0;
  while(nwords > 0){
    int const chunk = nwords < 480 ? nwords : 480;                          /* PCM_BUFSIZE, audio.c:19 */
    int not_silent = 0;
    for(int i = 0; i < chunk; i++){
      uint16_t const h = (uint16_t)scaleclip(*audio++);
      uint16_t const be = (uint16_t)((h << 8) | (h >> 8));                   /* htons on a little-endian host */
      *pcm_be++ = (int16_t)be;
      not_silent |= be;
    }
    if(!not_silent)
      mask |= 1u << chunks;
    chunks++;
    nwords -= chunk;
  }
  if(silent_mask)
    *silent_mask = mask;
  return chunks;
}

/* ---- PCM RTP packetising: audio.c:32-79 (stereo) and 82-132 (mono); header layout multicast.c:282-294 ----
 * Emits the datagrams send_*_output would pass to send(): 480-word chunks, all-zero chunks skipped while the
 * timestamp still advances, marker bit on the first packet after silence, sequence numbers on sent packets only.
 * Packets are written back to back as [2-byte little-endian length][bytes].  Returns the number of packets. */
int kqo_pcm_rtp(kqo_out_rtp *o, const float *audio, int nfloats, int stereo, unsigned char *dst, int cap, int *used){
  int packets = 0, pos = 0;
  int size = stereo ? nfloats / 2 : nfloats;                               /* frames */
  while(size > 0){
    int not_silent = 0;
    int const chunk = stereo ? (480 < 2 * size ? 480 : 2 * size) : (480 < size ? 480 : size);
    unsigned char words[2 * 480];
    for(int i = 0; i < chunk; i++){
      uint16_t const h = (uint16_t)scaleclip(*audio++);
      words[2 * i] = (unsigned char)(h >> 8);                                /* htons */
      words[2 * i + 1] = (unsigned char)h;
      not_silent |= h;
    }
    uint32_t const ts = o->timestamp;
    o->timestamp += stereo ? chunk / 2 : chunk;
    if(not_silent){
      o->packets++;
      o->bytes += 2 * chunk;
      int marker = 0;
      if(o->silent){
        o->silent = 0;
        marker = 1;
      }
      uint16_t const seq = o->seq++;
      int const len = 12 + 2 * chunk;
      if(pos + 2 + len > cap)
        return -1;
      unsigned char *dp = dst + pos;
      dp[0] = (unsigned char)len;
      dp[1] = (unsigned char)(len >> 8);
      dp += 2;
      dp[0] = 2 << 6;                                                        /* RTP_VERS, no pad / extension / CSRC */
      dp[1] = (unsigned char)((marker << 7) | (stereo ? 10 : 11));           /* PCM_STEREO_PT / PCM_MONO_PT */
      dp[2] = (unsigned char)(seq >> 8);
      dp[3] = (unsigned char)seq;
      dp[4] = (unsigned char)(ts >> 24);
      dp[5] = (unsigned char)(ts >> 16);
      dp[6] = (unsigned char)(ts >> 8);
      dp[7] = (unsigned char)ts;
      dp[8] = (unsigned char)(o->ssrc >> 24);
      dp[9] = (unsigned char)(o->ssrc >> 16);
      dp[10] = (unsigned char)(o->ssrc >> 8);
      dp[11] = (unsigned char)o->ssrc;
      memcpy(dp + 12, words, 2 * chunk);
      pos += 2 + len;
      packets++;
    } else
      o->silent = 1;
    size -= stereo ? chunk / 2 : chunk;
  }
  if(used)
    *used = pos;
  return packets;
}

/* ---- multi-channel CPU baseline (bench.py cpu_baseline leg only) ----
 * The reference runs one `radio` process per channel; here `nthreads` workers share the channels and walk them block
 * by block, as a real-time receiver would.  Channel set-up (response design, transform plans) and `warm` blocks per
 * channel happen before the clock starts; `timed` blocks per channel are timed between two barriers.  The input
 * holds nblocks_avail blocks and is cycled. */
struct bench_arg {
  const kqo_chan_cfg *cfgs;
  int first, last, nblocks_avail, warm, timed;
  const float *iq;
  double checksum;
  pthread_barrier_t *bar;
};

static void *bench_worker(void *p){
  struct bench_arg *a = p;
  int const n = a->last - a->first;
  kqo_chan **c = calloc((size_t)(n > 0 ? n : 1), sizeof(*c));
  unsigned L = 0, omax = 0;
  for(int i = 0; i < n; i++){
    c[i] = kqo_chan_create(&a->cfgs[a->first + i]);
    L = a->cfgs[a->first + i].L;
    if(kqo_chan_olen(c[i]) > omax)
      omax = kqo_chan_olen(c[i]);
  }
  float *audio = malloc(sizeof(float) * 2 * (omax + 1));
  double sum = 0;
  pthread_barrier_wait(a->bar);                         /* everyone is set up */
  for(int b = 0; b < a->warm + a->timed; b++){
    if(b == a->warm)
      pthread_barrier_wait(a->bar);                     /* the clock starts behind this one */
    const float *blk = a->iq + (size_t)2 * L * (size_t)(b % a->nblocks_avail);
    for(int i = 0; i < n; i++){
      kqo_status st;
      kqo_chan_block(c[i], blk, audio, &st, NULL, NULL);
      for(int k = 0; k < st.nout; k++)
        sum += audio[k];
    }
  }
  pthread_barrier_wait(a->bar);                         /* ... and stops behind this one */
  free(audio);
  for(int i = 0; i < n; i++)
    kqo_chan_destroy(c[i]);
  free(c);
  a->checksum = sum;
  return NULL;
}

double kqo_bench_channels(const kqo_chan_cfg *cfgs, int nchan, const float *iq, int nblocks_avail, int warm, int timed,
                          int nthreads, int fast_fft, double *checksum){
  if(nthreads < 1)
    nthreads = 1;
  if(nthreads > nchan)
    nthreads = nchan;
  if(nblocks_avail < 1 || timed < 1)
    return -1;
  kqo_fft_set_fast(fast_fft);
  pthread_t *tid = malloc(sizeof(pthread_t) * nthreads);
  struct bench_arg *args = calloc(nthreads, sizeof(*args));
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1);
  for(int t = 0; t < nthreads; t++){
    args[t].cfgs = cfgs;
    args[t].first = (int)((long long)nchan * t / nthreads);
    args[t].last = (int)((long long)nchan * (t + 1) / nthreads);
    args[t].nblocks_avail = nblocks_avail;
    args[t].warm = warm;
    args[t].timed = timed;
    args[t].iq = iq;
    args[t].bar = &bar;
    pthread_create(&tid[t], NULL, bench_worker, &args[t]);
  }
  struct timespec t0, t1;
  pthread_barrier_wait(&bar);
  pthread_barrier_wait(&bar);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_barrier_wait(&bar);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  double sum = 0;
  for(int t = 0; t < nthreads; t++){
    pthread_join(tid[t], NULL);
    sum += args[t].checksum;
  }
  kqo_fft_set_fast(0);
  pthread_barrier_destroy(&bar);
  if(checksum)
    *checksum = sum;
  free(tid);
  free(args);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
