/* kq_filter.c -- oracle restatement of the overlap-save fast-convolution filter
 * (test infrastructure only; PARITY UNPINNED, see kq_oracle.h).
 *
 * Follows filter.c of the reference:
 *   master  : create 54-91, execute 146-172, delete 254-263
 *   slave   : create 97-145, execute 175-252, delete 264-275
 *   design  : i0 282-293, make_kaiser 337-357, window_filter 365-415,
 *             window_rfilter 420-469, noise_gain 472-497, set_filter 500-546
 * Threading (condvar block counter, response mutex) is deliberately absent: the oracle is
 * single-threaded and the caller alternates input/output executions.
 */
#define _GNU_SOURCE 1
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include "kq_oracle.h"

static inline float norm2f(float complex z){
  return crealf(z) * crealf(z) + cimagf(z) * cimagf(z);      /* dsp.c:45-47 */
}

/* ---- master half ---- */
kqo_filter_in *kqo_create_filter_input(unsigned L, unsigned M, int in_type){
  kqo_filter_in *m = calloc(1, sizeof(*m));
  if(!m)
    return NULL;
  m->ilen = L;
  m->impulse_length = M;
  m->n = L + M - 1;
  m->blocknum = 0;
  m->plan = kqo_fft_create(m->n);
  if(!m->plan){
    fprintf(stderr, "kq oracle: FFT size %u has a prime factor beyond 7\n", m->n);
    free(m);
    return NULL;
  }
  if(in_type != KQO_REAL)
    in_type = KQO_COMPLEX;                                     /* filter.c:69-71 fall-through */
  m->in_type = in_type;
  if(in_type == KQO_COMPLEX){
    m->fdomain = calloc(m->n, sizeof(float complex));
    m->inbuf_c = calloc(m->n, sizeof(float complex));          /* history zeroed: filter.c:76 */
    m->input_c = m->inbuf_c + (M - 1);                         /* filter.c:77 */
  } else {
    m->fdomain = calloc(m->n / 2 + 1, sizeof(float complex));  /* filter.c:81 */
    m->inbuf_r = calloc(m->n, sizeof(float));
    m->input_r = m->inbuf_r + (M - 1);
  }
  return m;
}

int kqo_execute_filter_input(kqo_filter_in *m){
  if(!m)
    return -1;
  unsigned const keep = m->impulse_length - 1;
  if(m->in_type == KQO_COMPLEX){
    kqo_fft_c2c(m->plan, m->inbuf_c, m->fdomain, -1);          /* filter.c:151 */
    m->blocknum++;
    memmove(m->inbuf_c, m->inbuf_c + m->ilen, keep * sizeof(float complex)); /* filter.c:164 */
  } else {
    kqo_fft_r2c(m->plan, m->inbuf_r, m->fdomain);
    m->blocknum++;
    memmove(m->inbuf_r, m->inbuf_r + m->ilen, keep * sizeof(float));
  }
  return 0;
}

int kqo_delete_filter_input(kqo_filter_in *m){
  if(!m)
    return 0;
  kqo_fft_destroy(m->plan);
  free(m->inbuf_c);
  free(m->inbuf_r);
  free(m->fdomain);
  free(m);
  return 0;
}

/* ---- slave half ---- */
kqo_filter_out *kqo_create_filter_output(kqo_filter_in *m, float complex *response, unsigned decimate, int out_type){
  if(!m)
    return NULL;
  unsigned const n_dec = m->n / decimate;
  if(m->n % decimate != 0)
    fprintf(stderr, "Warning: FFT size %u is not divisible by decimation ratio %u\n", m->n, decimate); /* filter.c:106 */
  kqo_filter_out *s = calloc(1, sizeof(*s));
  if(!s)
    return NULL;
  s->master = m;
  s->out_type = out_type;
  s->decimate = decimate;
  s->n_dec = n_dec;
  s->olen = m->ilen / decimate;                                /* filter.c:116 */
  s->response = response;
  s->noise_gain = response ? kqo_noise_gain(s) : NAN;
  s->plan = kqo_fft_create(n_dec);
  if(!s->plan){
    fprintf(stderr, "kq oracle: decimated FFT size %u has a prime factor beyond 7\n", n_dec);
    free(s);
    return NULL;
  }
  if(out_type == KQO_REAL){
    s->f_fdomain = calloc(n_dec / 2 + 1, sizeof(float complex));
    s->outbuf_r = calloc(n_dec, sizeof(float));
    s->output_r = s->outbuf_r + (n_dec - s->olen);             /* filter.c:140 */
  } else {
    s->f_fdomain = calloc(n_dec, sizeof(float complex));
    s->outbuf_c = calloc(n_dec, sizeof(float complex));
    s->output_c = s->outbuf_c + (n_dec - s->olen);             /* filter.c:131 */
  }
  return s;
}

int kqo_execute_filter_output(kqo_filter_out *s){
  if(!s)
    return -1;
  kqo_filter_in const *m = s->master;
  int const N = (int)m->n;
  int const nd = (int)s->n_dec;
  float complex const *X = m->fdomain;
  float complex const *H = s->response;
  float complex *G = s->f_fdomain;

  s->blocknum = m->blocknum;                                   /* filter.c:198 (no waiting here) */

  for(int p = 0; p <= nd / 2; p++)                             /* filter.c:206-208 */
    G[p] = H[p] * X[p];

  if(m->in_type == KQO_REAL){
    if(s->out_type != KQO_REAL){
      /* real in, complex out: negative bins from the conjugate of the positive ones (filter.c:214-216) */
      for(int p = 1, k = nd - 1; k > nd / 2; p++, k--)
        G[k] = H[k] * conjf(X[p]);
    }
  } else if(s->out_type != KQO_REAL){
    /* complex in, complex out: top nd/2-1 bins of the N-point spectrum (filter.c:225-227) */
    for(int n = N - 1, k = nd - 1; k > nd / 2; n--, k--)
      G[k] = H[k] * X[n];
  } else {
    /* complex in, real out: fold conjugated negative side onto the positive side (filter.c:232-234) */
    for(int n = N - 1, p = 1, k = nd - 1; p < nd / 2; p++, n--, k--)
      G[p] += conjf(H[k] * X[n]);
  }

  if(s->out_type == KQO_CROSS_CONJ){                           /* filter.c:239-249 */
    for(int p = 1, k = nd - 1; p < nd / 2; p++, k--){
      float complex const pos = G[p], neg = G[k];
      G[p] = pos + conjf(neg);
      G[k] = neg - conjf(pos);
    }
  }
  if(s->out_type == KQO_REAL)
    kqo_fft_c2r(s->plan, G, s->outbuf_r);                      /* filter.c:250 */
  else
    kqo_fft_c2c(s->plan, G, s->outbuf_c, +1);
  return 0;
}

int kqo_delete_filter_output(kqo_filter_out *s){
  if(!s)
    return 0;
  kqo_fft_destroy(s->plan);
  free(s->outbuf_c);
  free(s->outbuf_r);
  free(s->response);
  free(s->f_fdomain);
  free(s);
  return 0;
}

/* ---- response design ---- */

/* Modified Bessel I0 by its power series, float arithmetic throughout (filter.c:282-293) */
static float bessel_i0(float x){
  float const t = 0.25 * x * x;
  float sum = 1 + t;
  float term = t;
  for(int k = 2; k < 40; k++){
    term *= t / (k * k);
    sum += term;
    if(term < 1e-12 * sum)
      break;
  }
  return sum;
}

int kqo_make_kaiser(float *window, unsigned M, float beta){
  if(!window)
    return -1;
  float const arg = M_PI * beta;                               /* filter.c:342 */
  float const scale = 1. / bessel_i0(arg);
  float const pc = 2.0 / (M - 1);
  for(unsigned n = 0; n < M / 2; n++){                         /* filter.c:348-351 */
    float const p = pc * n - 1;
    float const w = bessel_i0(arg * sqrtf(1 - p * p)) * scale;
    window[n] = w;
    window[M - 1 - n] = w;
  }
  if(M & 1)
    window[(M - 1) / 2] = 1;
  return 0;
}

int kqo_window_filter(int L, int M, float complex *response, float beta){
  if(!response)
    return -1;
  int const N = L + M - 1;
  kqo_fft *plan = kqo_fft_create((unsigned)N);
  if(!plan)
    return -1;
  float complex *buf = malloc(sizeof(float complex) * N);
  float *win = malloc(sizeof(float) * M);
  kqo_fft_c2c(plan, response, buf, +1);                        /* to time domain: filter.c:377-378 */
  kqo_make_kaiser(win, (unsigned)M, beta);
  float const gain = 1. / N;                                   /* filter.c:387 */
  /* in place, n descending, as filter.c:389-390 does it: no slot is read after it was written as long as L > M/2; with a
   * longer impulse response the first taps are formed from slots the loop has already written, here as there */
  for(int n = M - 1; n >= 0; n--)
    buf[n] = buf[(n - M / 2 + N) % N] * win[n] * gain;
  for(int n = M; n < N; n++)
    buf[n] = 0;
  kqo_fft_c2c(plan, buf, response, -1);                        /* filter.c:401,412 */
  free(win);
  free(buf);
  kqo_fft_destroy(plan);
  return 0;
}

int kqo_window_rfilter(int L, int M, float complex *response, float beta){
  if(!response)
    return -1;
  int const N = L + M - 1;
  kqo_fft *plan = kqo_fft_create((unsigned)N);
  if(!plan)
    return -1;
  float *tb = malloc(sizeof(float) * N);
  float *win = malloc(sizeof(float) * M);
  kqo_fft_c2r(plan, response, tb);                             /* filter.c:436-437 */
  kqo_make_kaiser(win, (unsigned)M, beta);
  float const gain = 1. / N;
  for(int n = M - 1; n >= 0; n--)                              /* filter.c:445-446 */
    tb[n] = tb[(n - M / 2 + N) % N] * win[n] * gain;
  for(int n = M; n < N; n++)
    tb[n] = 0;
  kqo_fft_r2c(plan, tb, response);                             /* filter.c:457,466 */
  free(win);
  free(tb);
  kqo_fft_destroy(plan);
  return 0;
}

float kqo_noise_gain(const kqo_filter_out *s){
  if(!s)
    return NAN;
  kqo_filter_in const *m = s->master;
  int const N = (int)m->n;
  int const nd = N / (int)s->decimate;
  int const count = (m->in_type == KQO_REAL && s->out_type == KQO_REAL) ? nd / 2 + 1 : nd; /* filter.c:481-487 */
  float sum = 0;
  for(int i = 0; i < count; i++)
    sum += norm2f(s->response[i]);
  if(s->out_type == KQO_REAL || s->out_type == KQO_CROSS_CONJ)  /* filter.c:493-496 */
    return 2 * N * sum;
  return N * sum;
}

int kqo_set_filter(kqo_filter_out *s, float low, float high, float beta){
  if(!s)
    return -1;
  if(isnan(low) || isnan(high))
    return -1;                                                 /* filter.c:504-505 */
  kqo_filter_in const *m = s->master;
  int const L_dec = (int)s->olen;
  int const M_dec = (int)((m->impulse_length - 1) / s->decimate + 1);  /* filter.c:514 */
  int const N_dec = L_dec + M_dec - 1;
  int const N = (int)m->n;

  float gain = 1. / ((float)N);                                /* filter.c:518 */
  if(s->out_type == KQO_REAL || s->out_type == KQO_CROSS_CONJ)
    gain *= M_SQRT1_2;
  float complex *resp = calloc((size_t)N_dec, sizeof(float complex));
  for(int n = 0; n < N_dec; n++){                              /* filter.c:525-535 */
    float const f = (n <= N_dec / 2) ? (float)n / N_dec : (float)(n - N_dec) / N_dec;
    resp[n] = (f >= low && f <= high) ? gain : 0;
  }
  kqo_window_filter(L_dec, M_dec, resp, beta);
  float complex *old = s->response;
  s->response = resp;
  s->noise_gain = kqo_noise_gain(s);
  free(old);
  return 0;
}

/* ---- experimental IIR complex notch: filter.c:549-571 ---- */
kqo_notch *kqo_notch_create(double f, float bw){
  kqo_notch *nf = calloc(1, sizeof *nf);
  if(!nf)
    return NULL;
  nf->osc_phase = 1;
  nf->osc_step = cos(2 * f * M_PI) + I * sin(2 * f * M_PI);   /* csincospi(2*f), dsp.c:45-50 */
  nf->dcstate = 0;
  nf->bw = bw;
  return nf;
}

float complex kqo_notch_step(kqo_notch *nf, float complex s){
  if(!nf)
    return NAN;
  s = s * conj(nf->osc_phase) - nf->dcstate;
  nf->dcstate += nf->bw * s;
  s *= nf->osc_phase;
  nf->osc_phase *= nf->osc_step;
  return s;
}

void kqo_notch_run(kqo_notch *nf, const float complex *in, float complex *out, int n){
  for(int i = 0; i < n; i++)
    out[i] = kqo_notch_step(nf, in[i]);
}
