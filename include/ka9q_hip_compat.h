/* ka9q_hip_compat.h -- the reference's own one-channel API, served by libka9q_hip.so.
 *
 * Same function names, argument meaning, struct field names and error behaviour as the
 * reference's filter.h / osc.h / dsp.h, so that radio.c, fm.c, am.c, linear.c, packet.c and
 * modulate.c compile against this header in place of theirs.  What changes
 * underneath: every FFT runs on the GPU (execute_filter_input = H2D + N-point forward transform +
 * D2H of fdomain; execute_filter_output = response multiply + N/D-point inverse on the resident
 * spectrum + D2H of the output block).  FFTW is not used anywhere.
 *
 * Differences a maintainer must know:
 *   - `fwd_plan` / `rev_plan` are opaque device contexts (void *), not fftwf_plan.
 *   - response arrays handed to create_filter_output() must come from malloc()/calloc() or from this library's
 *     fftwf_alloc_complex (ka9q_hip_fftw.h: what fm.c:56 calls, once -lfftw3f has left the link line); the reference frees
 *     them with fftwf_free (filter.c:271), here it is free() -- which releases either.
 *   - N = L+M-1: a power of two, 4 <= N <= 2^22, or an even 2^a 3^b 5^c 7^d up to 65536 (240 kHz front ends: L = 4800,
 *     M = 4801, decimate 5); past 16384 points the master's transform runs in two passes through device memory.
 *     N/decimate: the same kinds of size, 4 <= N/decimate <= 16384.  A prime factor beyond 7 returns NULL (FFTW would
 *     plan it, filter.c:78,132).
 *   - This surface moves one block over PCIe per call; it exists for drop-in correctness.  The
 *     throughput path is the channel bank in ka9q_hip.h.
 */
#ifndef KA9Q_HIP_COMPAT_H
#define KA9Q_HIP_COMPAT_H 1

#include <pthread.h>

#ifdef __cplusplus
typedef float _Complex kq_cfloat;
typedef double _Complex kq_cdouble;
extern "C" {
#else
#include <complex.h>
typedef float _Complex kq_cfloat;
typedef double _Complex kq_cdouble;
#endif

/* filter.h:17-22 */
enum filtertype { NONE, COMPLEX, CROSS_CONJ, REAL };

/* filter.h:25-28 */
union rc {
  float *r;
  kq_cfloat *c;
};

/* Master half -- field names of filter.h:54-66 */
struct filter_in {
  enum filtertype in_type;
  unsigned int ilen;            /* L */
  unsigned int impulse_length;  /* M */
  kq_cfloat *fdomain;           /* N bins (complex in) or N/2+1 (real in), host copy refreshed every block */
  union rc input_buffer;        /* N samples */
  union rc input;               /* user area: input_buffer + M-1 */
  void *fwd_plan;               /* opaque device context */
  unsigned int blocknum;
  pthread_mutex_t filter_mutex;
  pthread_cond_t filter_cond;
};

/* Slave half -- field names of filter.h:67-80 */
struct filter_out {
  struct filter_in *master;
  enum filtertype out_type;
  kq_cfloat *response;
  pthread_mutex_t response_mutex;
  kq_cfloat *f_fdomain;         /* kept for source compatibility; not filled (the product stays on the device) */
  float noise_gain;
  union rc output_buffer;       /* N/decimate samples */
  union rc output;              /* last olen of output_buffer */
  void *rev_plan;               /* opaque device context */
  unsigned int decimate;
  unsigned int olen;
  unsigned int blocknum;
};

/* filter.h:81-92 */
int window_filter(int L, int M, kq_cfloat *response, float beta);
int window_rfilter(int L, int M, kq_cfloat *response, float beta);
struct filter_in *create_filter_input(unsigned int L, unsigned int M, enum filtertype in_type);
struct filter_out *create_filter_output(struct filter_in *master, kq_cfloat *response, unsigned int decimate,
                                        enum filtertype out_type);
int execute_filter_input(struct filter_in *);
int execute_filter_output(struct filter_out *);
int delete_filter_input(struct filter_in *);
int delete_filter_output(struct filter_out *);
int make_kaiser(float *window, unsigned int M, float beta);
int set_filter(struct filter_out *, float low, float high, float kaiser_beta);
float noise_gain(struct filter_out const *);
extern float Kaiser_beta;       /* filter.c:279 */

/* osc.h:9-24 */
struct osc {
  double freq;
  double rate;
  kq_cdouble phasor;
  kq_cdouble phasor_step;
  kq_cdouble phasor_step_step;
  pthread_mutex_t mutex;
  int steps;
};
void set_osc(struct osc *osc, double f, double r);
kq_cdouble step_osc(struct osc *osc);
void renorm_osc(struct osc *osc);
int is_phasor_init(kq_cdouble x);

/* filter.h:95-105, filter.c:549-571: experimental IIR complex notch (its only call site, radio.c, is #if 0'd in the
 * reference; host scalar code here as there) */
struct notchfilter {
  kq_cdouble osc_phase; /* phase of the local complex mixer */
  kq_cdouble osc_step;  /* mixer phase increment */
  kq_cfloat dcstate;    /* average signal at the mixer frequency */
  float bw;             /* relative bandwidth of the notch */
};
struct notchfilter *notch_create(double f, float bw);
#define notch_delete(x) free(x)
kq_cfloat notch(struct notchfilter *nf, kq_cfloat s);

/* dsp.h:20-31 */
kq_cfloat csincosf(float x);
kq_cfloat csincospif(float x);
kq_cdouble csincos(double x);
kq_cdouble csincospi(double x);
float cnrmf(kq_cfloat x);
double cnrm(kq_cdouble x);

#ifdef __cplusplus
}
#endif
#endif
