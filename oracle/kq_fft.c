/* kq_fft.c -- single-precision FFT for the oracle (test infrastructure only): powers of two, and since round 6 every
 * n = 2^a 3^b 5^c 7^d (the sizes a front end whose rate is not 48 kHz x 2^k needs: filter.c:78,102-107,132 plan whatever
 * N and N / decimate come out, radio_status.c:266 gives decimate = samprate / 48000 -- 240 kHz: 5).
 *
 * Stands in for the FFTW3f calls of the reference (filter.c:78,87,132,141,373-374,430-432;
 * fm.c:228; linear.c:92).  Same conventions as FFTW: unnormalised, forward kernel
 * exp(-2*pi*i*jk/n), backward exp(+2*pi*i*jk/n); r2c returns n/2+1 bins; c2r consumes
 * n/2+1 bins and ignores the imaginary parts of DC and Nyquist.
 * Arithmetic is float (twiddles computed in double, rounded once), iterative radix-2
 * decimation in time after a bit-reversal permutation.  Sizes with a factor 3, 5 or 7 take a recursive mixed-radix
 * decimation in time (radices 4, 2, 3, 5, 7; the power-of-two path is left exactly as it was: its rounding is what the
 * committed vectors were generated with).  Checked against numpy's float64 transforms in tests/test_oracle_filter.py.
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
  float *work;            /* 4 n floats: split re / im ping-pong buffers of the fast variant */
  float *twr, *twi;       /* n floats each: exp(-2*pi*i*k/n) split, k < n (the radix-4 pass reads up to 3 j m < n) */
  unsigned nf, fac[32];   /* mixed sizes (log2n == 0 and n > 1): the radices, outermost first */
  float complex *twf;     /* mixed sizes: exp(-2*pi*i*k/n), k < n */
};

/* n = 2^a 3^b 5^c 7^d ?  fills the radix list (4s first, then 2, 3s, 5s, 7s) */
static int factorise(unsigned n, unsigned *fac, unsigned *nf){
  *nf = 0;
  while(n % 4 == 0){ fac[(*nf)++] = 4; n /= 4; }
  while(n % 2 == 0){ fac[(*nf)++] = 2; n /= 2; }
  while(n % 3 == 0){ fac[(*nf)++] = 3; n /= 3; }
  while(n % 5 == 0){ fac[(*nf)++] = 5; n /= 5; }
  while(n % 7 == 0){ fac[(*nf)++] = 7; n /= 7; }
  return n == 1;
}

/* an even n = 2^a 3^b 5^c 7^d: the sizes the real transforms below are used at */
int kqo_fft_size_ok(unsigned n){
  unsigned fac[32], nf;
  return n >= 2 && (n & 1) == 0 && factorise(n, fac, &nf);
}

kqo_fft *kqo_fft_create(unsigned n){
  if(n == 0)
    return NULL;
  if((n & (n - 1)) != 0){       /* a factor 3, 5 or 7: the mixed-radix plan */
    unsigned fac[32], nf;
    if(!factorise(n, fac, &nf))
      return NULL;
    kqo_fft *p = calloc(1, sizeof(*p));
    p->n = n;
    p->nf = nf;
    memcpy(p->fac, fac, sizeof fac);
    p->twf = malloc(sizeof(float complex) * n);
    p->scratch = malloc(sizeof(float complex) * 2 * n);
    for(unsigned k = 0; k < n; k++){
      double sn, cs;
      sincos(-2.0 * M_PI * (double)k / (double)n, &sn, &cs);
      p->twf[k] = (float)cs + (float)sn * _Complex_I;
    }
    return p;
  }
  kqo_fft *p = calloc(1, sizeof(*p));
  p->n = n;
  while((1u << p->log2n) < n)
    p->log2n++;
  p->tw = malloc(sizeof(float complex) * (n / 2 + 1));
  p->rev = malloc(sizeof(unsigned) * n);
  p->scratch = malloc(sizeof(float complex) * n);
  p->work = malloc(sizeof(float) * 4 * n);
  p->twr = malloc(sizeof(float) * n);
  p->twi = malloc(sizeof(float) * n);
  for(unsigned k = 0; k < n; k++){
    double s, c;
    sincos(-2.0 * M_PI * (double)k / (double)n, &s, &c);
    p->twr[k] = (float)c;
    p->twi[k] = (float)s;
  }
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
  free(p->work);
  free(p->twr);
  free(p->twi);
  free(p->twf);
  free(p);
}

static inline float complex cmulf_(float complex a, float complex b){
  float ar = crealf(a), ai = cimagf(a), br = crealf(b), bi = cimagf(b);
  return (ar * br - ai * bi) + (ar * bi + ai * br) * _Complex_I;
}

/* ---- fast variant, used by the CPU-baseline leg of bench.py only (kqo_fft_set_fast(1)) ----
 * The parity path keeps the plain radix-2 transform above and below: its rounding is what the committed oracle
 * vectors were generated with.  The baseline times the same chain with a radix-4 (plus one radix-2 pass when
 * log2 n is odd) Stockham autosort transform: no bit-reversal pass, unit-stride inner loops over split re / im
 * arrays that the compiler vectorises (function clones for AVX2 / AVX-512 are picked at load time).  Same
 * conventions, results equal to the plain transform to float rounding
 * (tests/test_oracle_filter.py::test_fast_transform_equals_the_plain_one).  Writes the plan's scratch: one plan per thread. */
static int Fast;
void kqo_fft_set_fast(int on){ Fast = on; }

#define KQO_CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))

/* one radix-4 pass: n = 4 * l * m; x[(4 j + q) * m + k] -> y[(j + q l) * m + k], twiddles w^{j m q} */
KQO_CLONES static void pass4(unsigned l, unsigned m, const float *restrict xr, const float *restrict xi, float *restrict yr,
                             float *restrict yi, const float *restrict twr, const float *restrict twi, unsigned tstride, float sgn){
  for(unsigned j = 0; j < l; j++){
    float const w1r = twr[j * tstride], w1i = sgn * twi[j * tstride];
    float const w2r = twr[2 * j * tstride], w2i = sgn * twi[2 * j * tstride];
    float const w3r = twr[3 * j * tstride], w3i = sgn * twi[3 * j * tstride];
    const float *ar = xr + (size_t)4 * j * m, *ai = xi + (size_t)4 * j * m;
    float *o0r = yr + (size_t)j * m, *o0i = yi + (size_t)j * m;
    float *o1r = o0r + (size_t)l * m, *o1i = o0i + (size_t)l * m;
    float *o2r = o1r + (size_t)l * m, *o2i = o1i + (size_t)l * m;
    float *o3r = o2r + (size_t)l * m, *o3i = o2i + (size_t)l * m;
    for(unsigned k = 0; k < m; k++){
      float const a0r = ar[k], a0i = ai[k];
      float const b1r = ar[m + k], b1i = ai[m + k], b2r = ar[2 * m + k], b2i = ai[2 * m + k], b3r = ar[3 * m + k], b3i = ai[3 * m + k];
      float const a1r = b1r * w1r - b1i * w1i, a1i = b1r * w1i + b1i * w1r;
      float const a2r = b2r * w2r - b2i * w2i, a2i = b2r * w2i + b2i * w2r;
      float const a3r = b3r * w3r - b3i * w3i, a3i = b3r * w3i + b3i * w3r;
      float const s02r = a0r + a2r, s02i = a0i + a2i, d02r = a0r - a2r, d02i = a0i - a2i;
      float const s13r = a1r + a3r, s13i = a1i + a3i, d13r = a1r - a3r, d13i = a1i - a3i;
      o0r[k] = s02r + s13r;
      o0i[k] = s02i + s13i;
      o2r[k] = s02r - s13r;
      o2i[k] = s02i - s13i;
      /* times -i (forward) or +i (backward): sgn = +1 forward */
      o1r[k] = d02r + sgn * d13i;
      o1i[k] = d02i - sgn * d13r;
      o3r[k] = d02r - sgn * d13i;
      o3i[k] = d02i + sgn * d13r;
    }
  }
}
KQO_CLONES static void pass2(unsigned l, unsigned m, const float *restrict xr, const float *restrict xi, float *restrict yr,
                             float *restrict yi, const float *restrict twr, const float *restrict twi, unsigned tstride, float sgn){
  for(unsigned j = 0; j < l; j++){
    float const wr = twr[j * tstride], wi = sgn * twi[j * tstride];
    const float *ar = xr + (size_t)2 * j * m, *ai = xi + (size_t)2 * j * m;
    float *o0r = yr + (size_t)j * m, *o0i = yi + (size_t)j * m, *o1r = o0r + (size_t)l * m, *o1i = o0i + (size_t)l * m;
    for(unsigned k = 0; k < m; k++){
      float const br = ar[m + k] * wr - ai[m + k] * wi, bi = ar[m + k] * wi + ai[m + k] * wr;
      o0r[k] = ar[k] + br;
      o0i[k] = ai[k] + bi;
      o1r[k] = ar[k] - br;
      o1i[k] = ai[k] - bi;
    }
  }
}

/* decimation in time, Stockham: after the pass with sub-transform count l the data holds l-point transforms of the
 * m = n / (r l) interleaved sub-sequences */
static void fft_fast(const kqo_fft *p, const float complex *in, float complex *out, int sign){
  unsigned const n = p->n;
  float *ar = p->work, *ai = ar + n, *br = ai + n, *bi = br + n;
  for(unsigned i = 0; i < n; i++){
    ar[i] = crealf(in[i]);
    ai[i] = cimagf(in[i]);
  }
  float const sgn = sign > 0 ? -1.f : 1.f;   /* the table holds the forward twiddles */
  unsigned l = 1, m = n;
  /* x[n1*m' + k]: at each pass split the current length-m sub-sequences by the radix r: m' = m / r */
  while(m > 1){
    unsigned const r = (m % 4 == 0) ? 4 : 2;
    m /= r;
    /* twiddle exponent: w_{r l}^{j q} = W_n^{j q n / (r l)} = W_n^{j q m} */
    if(r == 4)
      pass4(l, m, ar, ai, br, bi, p->twr, p->twi, m, sgn);
    else
      pass2(l, m, ar, ai, br, bi, p->twr, p->twi, m, sgn);
    float *t;
    t = ar; ar = br; br = t;
    t = ai; ai = bi; bi = t;
    l *= r;
  }
  for(unsigned i = 0; i < n; i++)
    out[i] = ar[i] + ai[i] * _Complex_I;
}

/* ---- mixed radix, decimation in time: n = r m; the r sub-sequences in[q + r j] are transformed (recursively) into
 * out[q m .. q m + m - 1], then for every k < m the r values out[k + q m] W_n^{q k} go through an r-point transform.
 * `stride` = distance of consecutive elements of `in`; tw = exp(-+2 pi i k / ntot) (ntot / n = tstride). */
static void mixed_rec(const kqo_fft *p, unsigned level, unsigned n, const float complex *in, unsigned stride, float complex *out,
                      int sign){
  if(n == 1){
    out[0] = in[0];
    return;
  }
  unsigned const r = p->fac[level], m = n / r, tstride = p->n / n;
  for(unsigned q = 0; q < r; q++)
    mixed_rec(p, level + 1, m, in + (size_t)q * stride, stride * r, out + (size_t)q * m, sign);
  for(unsigned k = 0; k < m; k++){
    float complex a[7], b[7];
    for(unsigned q = 0; q < r; q++){
      float complex w = p->twf[(size_t)q * k * tstride];
      if(sign > 0)
        w = conjf(w);
      a[q] = q ? cmulf_(out[k + q * m], w) : out[k];
    }
    for(unsigned t = 0; t < r; t++){            /* the r-point transform, W_r^{q t} from the same table */
      float complex acc = a[0];
      for(unsigned q = 1; q < r; q++){
        float complex w = p->twf[(size_t)((q * t) % r) * (p->n / r)];
        if(sign > 0)
          w = conjf(w);
        acc += cmulf_(a[q], w);
      }
      b[t] = acc;
    }
    for(unsigned t = 0; t < r; t++)
      out[k + t * m] = b[t];
  }
}

void kqo_fft_c2c(const kqo_fft *p, const float complex *in, float complex *out, int sign){
  unsigned const n = p->n;
  if(p->twf){                                   /* a size with a factor 3, 5 or 7 */
    float complex *src = p->scratch + n;         /* (r2c / c2r hand in p->scratch itself) */
    memcpy(src, in, sizeof(float complex) * n);
    mixed_rec(p, 0, n, src, 1, out, sign);
    return;
  }
  if(Fast && n >= 4){
    fft_fast(p, in, out, sign);
    return;
  }
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
