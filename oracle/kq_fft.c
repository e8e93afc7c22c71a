/* kq_fft.c -- power-of-two single-precision FFT for the oracle (test infrastructure only).
 *
 * Stands in for the FFTW3f calls of the reference (filter.c:78,87,132,141,373-374,430-432;
 * fm.c:228; linear.c:92).  Same conventions as FFTW: unnormalised, forward kernel
 * exp(-2*pi*i*jk/n), backward exp(+2*pi*i*jk/n); r2c returns n/2+1 bins; c2r consumes
 * n/2+1 bins and ignores the imaginary parts of DC and Nyquist.
 * Arithmetic is float (twiddles computed in double, rounded once), iterative radix-2
 * decimation in time after a bit-reversal permutation.
 */
#define _GNU_SOURCE 1
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "kq_oracle.h"

struct kqo_fft {
  unsigned n, log2n;
  float complex *tw;      /* tw[k] = exp(-2*pi*i*k/n), k < n/2 */
  unsigned *rev;          /* bit reversal */
  float complex *scratch; /* n, for r2c/c2r and in-place safety */
};

kqo_fft *kqo_fft_create(unsigned n){
  if(n == 0 || (n & (n - 1)) != 0)
    return NULL;
  kqo_fft *p = calloc(1, sizeof(*p));
  p->n = n;
  while((1u << p->log2n) < n)
    p->log2n++;
  p->tw = malloc(sizeof(float complex) * (n / 2 + 1));
  p->rev = malloc(sizeof(unsigned) * n);
  p->scratch = malloc(sizeof(float complex) * n);
  for(unsigned k = 0; k < n / 2 + 1; k++){
    double s, c;
    sincos(-2.0 * M_PI * (double)k / (double)n, &s, &c);
    p->tw[k] = (float)c + (float)s * _Complex_I;
  }
  for(unsigned i = 0; i < n; i++){
    unsigned r = 0;
    for(unsigned b = 0; b < p->log2n; b++)
      if(i & (1u << b))
        r |= 1u << (p->log2n - 1 - b);
    p->rev[i] = r;
  }
  return p;
}

void kqo_fft_destroy(kqo_fft *p){
  if(!p)
    return;
  free(p->tw);
  free(p->rev);
  free(p->scratch);
  free(p);
}

static inline float complex cmulf_(float complex a, float complex b){
  float ar = crealf(a), ai = cimagf(a), br = crealf(b), bi = cimagf(b);
  return (ar * br - ai * bi) + (ar * bi + ai * br) * _Complex_I;
}

void kqo_fft_c2c(const kqo_fft *p, const float complex *in, float complex *out, int sign){
  unsigned const n = p->n;
  if(in == out){
    for(unsigned i = 0; i < n; i++){
      unsigned r = p->rev[i];
      if(r > i){
        float complex t = out[i];
        out[i] = out[r];
        out[r] = t;
      }
    }
  } else {
    for(unsigned i = 0; i < n; i++)
      out[p->rev[i]] = in[i];
  }
  for(unsigned len = 2; len <= n; len <<= 1){
    unsigned const half = len >> 1;
    unsigned const stride = n / len;
    for(unsigned base = 0; base < n; base += len){
      for(unsigned j = 0; j < half; j++){
        float complex w = p->tw[j * stride];
        if(sign > 0)
          w = conjf(w);
        float complex const u = out[base + j];
        float complex const v = cmulf_(out[base + j + half], w);
        out[base + j] = u + v;
        out[base + j + half] = u - v;
      }
    }
  }
}

void kqo_fft_r2c(const kqo_fft *p, const float *in, float complex *out){
  unsigned const n = p->n;
  float complex *s = p->scratch;
  for(unsigned i = 0; i < n; i++)
    s[i] = in[i];
  kqo_fft_c2c(p, s, s, -1);
  memcpy(out, s, sizeof(float complex) * (n / 2 + 1));
}

void kqo_fft_c2r(const kqo_fft *p, const float complex *in, float *out){
  unsigned const n = p->n;
  float complex *s = p->scratch;
  s[0] = crealf(in[0]);
  if(n > 1)
    s[n / 2] = crealf(in[n / 2]);
  for(unsigned k = 1; k < n / 2; k++){
    s[k] = in[k];
    s[n - k] = conjf(in[k]);
  }
  kqo_fft_c2c(p, s, s, +1);
  for(unsigned i = 0; i < n; i++)
    out[i] = crealf(s[i]);
}
