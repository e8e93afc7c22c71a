/* ka9q_hip_fftw.h -- the FFTW3f entry points ka9q-radio calls OUTSIDE its filter API, served by libka9q_hip.so.
 *
 * filter.c is replaced as a whole by ka9q_hip_compat.h.  Three more places of the hot path's files go to FFTW directly:
 *   fm.c:226-228,255,281-283   pltask: fftwf_plan_dft_r2c_1d of 16384 points, executed every 512 samples
 *   linear.c:90-92,178,313-317 carrier search: fftwf_plan_dft_1d of 65536 points
 *   fm.c:56,208, modulate.c:115 fftwf_alloc_complex for responses that create_filter_output() takes ownership of
 *   main.c:102-103,183-184     fftwf_import_system_wisdom, fftwf_make_planner_thread_safe, fftwf_init_threads,
 *                              fftwf_plan_with_nthreads
 * With the definitions below `fm.o`, `linear.o` and `main.o` -- compiled against the system's own <fftw3.h>, unchanged --
 * link against this library alone: -lfftw3f and -lfftw3f_threads leave the link line.  The prototypes are
 * FFTW 3.3's (fftw3.h); a program that includes <fftw3.h> does not need this header, it documents what the library exports
 * and serves hosts without FFTW's headers.  Every transform runs on the GPU: fftwf_execute moves one transform over the link
 * and back (16384 points: ~0.1 ms), like the rest of the compat surface.
 *
 * Differences: sizes are powers of two up to 2^22 or even 2^a 3^b 5^c 7^d up to 65536 (NULL plan otherwise: FFTW plans any n);
 * `flags` are ignored; plans are not thread-safe against each other's execution beyond one transform at a time (a mutex);
 * fftwf_alloc_* memory is released by fftwf_free OR free() -- delete_filter_output's free() of a response from
 * fftwf_alloc_complex (fm.c:56, filter.c:271) is therefore fine, which ka9q_hip_compat.h used to list as a difference.
 */
#ifndef KA9Q_HIP_FFTW_H
#define KA9Q_HIP_FFTW_H 1

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef FFTW_FORWARD
#define FFTW_FORWARD (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_ESTIMATE (1U << 6)
typedef float fftwf_complex[2];
typedef struct kq_fftwf_plan_s *fftwf_plan;
#endif

void *fftwf_malloc(size_t n);
float *fftwf_alloc_real(size_t n);
fftwf_complex *fftwf_alloc_complex(size_t n);
void fftwf_free(void *p);

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);  /* linear.c:92 */
fftwf_plan fftwf_plan_dft_r2c_1d(int n, float *in, fftwf_complex *out, unsigned flags);                /* fm.c:228 */
fftwf_plan fftwf_plan_dft_c2r_1d(int n, fftwf_complex *in, float *out, unsigned flags);
void fftwf_execute(const fftwf_plan p);                                                                /* fm.c:255, linear.c:178 */
void fftwf_destroy_plan(fftwf_plan p);                                                                 /* fm.c:281, linear.c:317 */

int fftwf_import_system_wisdom(void);       /* main.c:102: returns 1 */
void fftwf_make_planner_thread_safe(void);  /* main.c:103: nothing to do */
int fftwf_init_threads(void);               /* main.c:183: returns 1 */
void fftwf_plan_with_nthreads(int nthreads); /* main.c:184: ignored */

#ifdef __cplusplus
}
#endif
#endif
