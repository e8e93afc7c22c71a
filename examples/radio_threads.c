/* radio_threads.c -- the reference's thread structure on top of libka9q_hip.so, from plain C.
 *
 * What main.c / radio.c do around one channel, reduced to the calls that touch the library: create the master
 * (main.c:232), fill a `struct demod`, start the mode's demodulator thread (radio.c:372), then per block mix the
 * input with the second LO sample by sample into filter.in->input.c (radio.c:132-139) and run execute_filter_input
 * (radio.c:142).  The demodulator thread hands audio back through send_mono_output / send_stereo_output
 * (audio.c:82, 32), defined here as capture functions.
 *
 * By default the thread is the library's own demod_fm / demod_am / demod_linear (include/ka9q_hip_radio.h).  With
 * --ref FILE the entry point is taken from that shared object instead: oracle/Makefile builds
 * oracle/_ref/libref_am.so from the reference's am.c, compiled where it lies and unmodified against
 * ka9q_hip_compat.h -- the reference's demodulator loop running on this library's create_filter_output /
 * set_filter / execute_filter_output (tests/test_gpu_dropin.py).
 *
 *   gcc -std=gnu11 -O2 -Iinclude examples/radio_threads.c -Lka9q_sdr_amd/lib -lka9q_hip \
 *       -Wl,-rpath,$PWD/ka9q_sdr_amd/lib -rdynamic -ldl -lpthread -lm -o radio_threads
 *   radio_threads am 192000 3840 3841 4 -5000 5000 6 in.cf32 out.bin [--lo HZ] [--ref libref_am.so]
 *              [--stereo] [--isb] [--flat] [--shift HZ] [--hang S] [--recovery DBPS] [--time]
 *
 * --time prints, per block (mean over the run, the first four blocks left out): the host's mix loop, execute_filter_input
 * (upload, N-point transform on the GPU, spectrum back to filter.in->fdomain), the demodulator thread's block (from
 * execute_filter_input's return to the hand-off: slave, compute_n0, demodulator, audio back) and the real-time factor
 * block duration / their sum.
 *
 * out.bin, per block: int32 n, float audio[n], then 8 floats read at the hand-off: bb_power, n0, snr, foffset,
 * pdeviation, agc.gain, plfreq, noise_gain.  (FM sets its status before the hand-off, AM and linear after it:
 * fm.c:92-154 / am.c:76-78 / linear.c:302-309, so for those two the record of block b shows block b-1's bb_power.)
 */
#define _GNU_SOURCE 1
#include <complex.h>
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <semaphore.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ka9q_hip_radio.h"

static double now_us(void){
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static FILE *Out;
static sem_t Block_done;

static void record(struct demod *demod, const float *buf, int n){
  int32_t const cnt = n;
  float const st[8] = { demod->sig.bb_power, demod->sig.n0, demod->sig.snr, demod->sig.foffset, demod->sig.pdeviation,
                        demod->agc.gain, demod->sig.plfreq, demod->filter.out ? demod->filter.out->noise_gain : NAN };
  fwrite(&cnt, sizeof cnt, 1, Out);
  fwrite(buf, sizeof *buf, (size_t)n, Out);
  fwrite(st, sizeof st, 1, Out);
  sem_post(&Block_done);
}
/* audio.c:82 and audio.c:32 take `size` samples per channel */
int send_mono_output(struct demod *demod, const float *buffer, int size){
  record(demod, buffer, size);
  return 0;
}
int send_stereo_output(struct demod *demod, const float *buffer, int size){
  record(demod, buffer, 2 * size);
  return 0;
}
/* radio.c:383-425 stays with the host program in the reference; here it asks the library, which evaluates it on the
 * master's device-resident spectrum (the reference's am.c calls it by this name) */
float const compute_n0(struct demod const *demod){
  return kq_compat_compute_n0(demod->filter.in, demod->input.samprate, demod->filter.low, demod->filter.high);
}

int main(int argc, char **argv){
  if(argc < 11){
    fprintf(stderr, "usage: %s fm|am|linear samprate L M D low high nblocks in.cf32 out.bin [options]\n", argv[0]);
    return 2;
  }
  const char *mode = argv[1];
  int const samprate = atoi(argv[2]);
  unsigned const L = (unsigned)atoi(argv[3]), M = (unsigned)atoi(argv[4]);
  int const D = atoi(argv[5]);
  float const low = (float)atof(argv[6]), high = (float)atof(argv[7]);
  int const nblocks = atoi(argv[8]);
  const char *in_path = argv[9], *out_path = argv[10], *ref = NULL;
  double lo = 0, shift = 0;
  float hang = 0, recovery = 0;
  int stereo = 0, isb = 0, flat = 0, timing = 0;
  for(int i = 11; i < argc; i++){
    if(!strcmp(argv[i], "--lo") && i + 1 < argc) lo = atof(argv[++i]);
    else if(!strcmp(argv[i], "--ref") && i + 1 < argc) ref = argv[++i];
    else if(!strcmp(argv[i], "--shift") && i + 1 < argc) shift = atof(argv[++i]);
    else if(!strcmp(argv[i], "--hang") && i + 1 < argc) hang = (float)atof(argv[++i]);
    else if(!strcmp(argv[i], "--recovery") && i + 1 < argc) recovery = (float)atof(argv[++i]);
    else if(!strcmp(argv[i], "--stereo")) stereo = 1;
    else if(!strcmp(argv[i], "--isb")) isb = 1;
    else if(!strcmp(argv[i], "--flat")) flat = 1;
    else if(!strcmp(argv[i], "--time")) timing = 1;
    else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  void *(*entry)(void *) = !strcmp(mode, "fm") ? demod_fm : !strcmp(mode, "am") ? demod_am : demod_linear;
  if(ref){
    void *h = dlopen(ref, RTLD_NOW | RTLD_GLOBAL);
    if(!h){ fprintf(stderr, "dlopen %s: %s\n", ref, dlerror()); return 1; }
    char name[32];
    snprintf(name, sizeof name, "demod_%s", mode);
    entry = (void *(*)(void *))dlsym(h, name);      /* the object's own definition, not the library's */
    if(!entry){ fprintf(stderr, "%s has no %s\n", ref, name); return 1; }
  }
  FILE *in = fopen(in_path, "rb");
  Out = fopen(out_path, "wb");
  if(!in || !Out){ perror("open"); return 1; }
  sem_init(&Block_done, 0, 0);

  struct demod *demod = calloc(1, sizeof *demod);   /* main.c:111-126: the few fields the path reads */
  demod->input.samprate = samprate;
  demod->filter.L = (int)L;
  demod->filter.M = (int)M;
  demod->filter.decimate = D;
  demod->filter.interpolate = 1;
  demod->filter.low = low;
  demod->filter.high = high;
  demod->filter.kaiser_beta = 3.0f;
  demod->filter.isb = isb;
  demod->opt.flat = flat;
  demod->agc.headroom = powf(10.f, -15.f / 20.f);   /* main.c:117 */
  demod->agc.hangtime = hang;
  demod->agc.recovery_rate = recovery;
  demod->output.channels = stereo ? 2 : 1;
  demod->sig.n0 = NAN;
  demod->sig.plfreq = NAN;
  pthread_mutex_init(&demod->second_LO.mutex, NULL);
  pthread_mutex_init(&demod->shift.mutex, NULL);
  pthread_mutex_init(&demod->doppler.mutex, NULL);
  set_osc(&demod->second_LO, lo / samprate, 0.0);                         /* radio.c:299 */
  if(shift != 0) set_osc(&demod->shift, shift * D / (double)samprate, 0.0); /* radio.c:309 */

  demod->filter.in = create_filter_input(L, M, COMPLEX);                  /* main.c:232 */
  if(!demod->filter.in){ fprintf(stderr, "create_filter_input failed\n"); return 1; }
  pthread_create(&demod->demod_thread, NULL, entry, demod);               /* radio.c:372 */

  float complex *blk = malloc(L * sizeof *blk);
  int rc = 0;
  double t_mix = 0, t_master = 0, t_thread = 0;
  int timed = 0;
  for(int b = 0; b <= nblocks; b++){
    if(b == nblocks){
      demod->terminate = 1;                 /* radio.c:335: the thread leaves after one more block */
      memset(blk, 0, L * sizeof *blk);
    } else if(fread(blk, sizeof *blk, L, in) != L){
      fprintf(stderr, "short input\n");
      rc = 1;
      break;
    }
    double const t0 = now_us();
    for(unsigned i = 0; i < L; i++){        /* radio.c:132-139: the product is formed in double */
      double complex const s = (double complex)blk[i] * step_osc(&demod->second_LO);
      demod->filter.in->input.c[i] = (float complex)s;
    }
    double const t1 = now_us();
    execute_filter_input(demod->filter.in);
    double const t2 = now_us();
    if(b < nblocks){
      struct timespec ts;
      clock_gettime(CLOCK_REALTIME, &ts);
      ts.tv_sec += 60;
      if(sem_timedwait(&Block_done, &ts)){   /* the demodulator thread gave up: its message is on stderr */
        fprintf(stderr, "no output for block %d\n", b);
        rc = 1;
        demod->terminate = 1;
        break;
      }
      if(b >= 4){
        t_mix += t1 - t0;
        t_master += t2 - t1;
        t_thread += now_us() - t2;
        timed++;
      }
    }
  }
  if(timing && timed > 0){
    double const block_us = 1e6 * L / samprate, sum = (t_mix + t_master + t_thread) / timed;
    printf("timing: N %u D %d: mix %.1f us, execute_filter_input %.1f us, demodulator thread %.1f us per block of %.1f us: "
           "%.1f x real time (%d blocks)\n", L + M - 1, D, t_mix / timed, t_master / timed, t_thread / timed, block_us,
           block_us / sum, timed);
  }
  if(rc == 0)
    pthread_join(demod->demod_thread, NULL);
  /* the state the thread left behind (AM / linear write bb_power after the hand-off) */
  int32_t const tail = -1;
  float const st[8] = { demod->sig.bb_power, demod->sig.n0, demod->sig.snr, demod->sig.foffset, demod->sig.pdeviation,
                        demod->agc.gain, demod->sig.plfreq, NAN };
  fwrite(&tail, sizeof tail, 1, Out);
  fwrite(st, sizeof st, 1, Out);
  fclose(Out);
  if(rc == 0 && demod->filter.out != NULL){
    fprintf(stderr, "the thread left demod->filter.out set\n");
    rc = 1;
  }
  delete_filter_input(demod->filter.in);
  puts(rc == 0 ? "ok" : "failed");
  return rc;
}
