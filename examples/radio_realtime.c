/* radio_realtime.c -- tens of thousands of receiver channels on ONE GPU, every one of them at 1.0 x the front end's rate: the
 * reference's operating point (one `radio` process per channel behind one multicast group, main.c:105, README.md:470-477) as
 * one channel bank fed in small batches.  INTEGRATION.md section B2.
 *
 *   radio_realtime [channels [blocks_per_call [seconds [pcm [operator [paced]]]]]]      defaults 32768 2 3 1 0 0
 *
 * A 10 MS/s front end with FM carriers 140 kHz apart (rank 0 of radio_fanout.c's stream), N = 16384, decimate 256: a
 * batch of 2 blocks is 1.64 ms of signal.  The host's loop is the one a live receiver runs --
 *
 *   process batch k;  push batch k + 1;  queue the delivery of batch k;  wait for the delivery of batch k - 2
 *
 * -- with the input in pinned host memory (what a socket reader fills) and every channel's audio handed back to pinned host
 * memory after every call, as clipped big-endian int16 PCM words (audio.c:22-28: what send_mono_output puts on the wire) with
 * the silent-packet masks, or as floats (pcm = 0), plus the status plane.  Set-up is one kq_bank_add_channels call.
 * operator = 1: somebody works the receiver meanwhile, ON A THREAD OF THEIR OWN as in the reference (display.c / radio_status.c
 * beside the demodulator threads): about once per call period one channel's filter is changed (kq_bank_set_filter:
 * display.c:161-177), every other time a channel is dropped or the dropped one comes back (kq_bank_remove_channel /
 * kq_bank_add_channel), every fourth a channel is retuned (kq_bank_set_second_lo).  None of these waits for the device; the
 * calls in flight keep the parameters they were queued with.  Both threads take the handle's lock: the receiver's worst wait
 * for it and the operator's longest hold are printed (kq_host_timing.lock_wait_max_ms / ctl_hold_max_ms).
 * paced = 1: the front end is a clock -- batch n is complete at A_n = start + (n + 1) x 1.64 ms and the loop takes it then
 * (process the batch before it, push this one, queue the delivery, take the delivery queued two calls ago), as a socket
 * reader takes main.c:288-365's packets.  Printed:
 * deliveries in hand more than one call period behind that schedule (late), the deepest backlog in whole periods, and the
 * intervals between deliveries.  paced = 0: batch after batch as fast as they go; the factor printed is then a mean.
 * Prints the real-time factor (signal time / wall time; >= 1 means the bank keeps up), the host's own time per call and
 * the delivery rate, and checks a delivered channel: squelch open, the 1 kHz tone's deviation seen.
 *
 *   gcc -std=gnu11 -O2 -Iinclude examples/radio_realtime.c -Lka9q_sdr_amd/lib -lka9q_hip \
 *       -Wl,-rpath,$PWD/ka9q_sdr_amd/lib -lm -lpthread -o radio_realtime
 * (nothing of HIP in the host program: the pinned buffers come from kq_host_alloc.)
 */
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ka9q_hip.h"

enum { L = 8192, M = 8193, D = 256, SAMPRATE = 10000000, OLEN = L / D };
#define NEMIT 16
#define DEVIATION 3000.0

static double now_s(void){
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static double emitter_freq(int e){ return (e - (NEMIT - 1) / 2.0) * 140000.0; }

/* the operator's thread: a change about once per call period, until told to stop */
struct operator_args {
  kq_bank *bank;
  kq_channel_config *cc;
  unsigned C;
  double period_s;
  volatile int stop, failed;
  long ops;
  double worst_s, total_s;
};
static void *operator_thread(void *arg){
  struct operator_args *o = arg;
  kq_bank *bank = o->bank;
  unsigned const C = o->C;
  int away = -1;                        /* the channel that has been dropped and not yet come back */
  for(long k = 0; !o->stop; k++){
    double const t0 = now_s();
    int rc = 0;
    unsigned const c = 64u + (unsigned)((k * 7919) % (C - 64));  /* (channels 0..63 are left alone: the check reads channel 5) */
    float const w = 6000.f + 125.f * (float)(k % 17);
    if((int)c != away){
      rc |= kq_bank_set_filter(bank, (int)c, -w, w, 3.0f) != 0;
      o->ops++;
    }
    if(k % 2 == 0){
      if(away < 0){
        away = 64 + (int)((k * 104729 + 3) % (C - 64));
        rc |= kq_bank_remove_channel(bank, away) != 0;
      } else {
        rc |= kq_bank_add_channel(bank, &o->cc[away]) != away;   /* the bank hands out the lowest hole: the only one */
        away = -1;
      }
      o->ops++;
    }
    if(k % 4 == 1){
      unsigned const r = 64u + (unsigned)((k * 15485863) % (C - 64));
      if((int)r != away){
        rc |= kq_bank_set_second_lo(bank, (int)r, o->cc[r].second_lo + (k & 4 ? 1.0 : 0.0)) != 0;
        o->ops++;
      }
    }
    if(rc){
      fprintf(stderr, "operator: %s\n", kq_last_error());
      o->failed = 1;
      break;
    }
    double const dt = now_s() - t0;       /* this round's two or three operations together */
    o->total_s += dt;
    if(dt > o->worst_s)
      o->worst_s = dt;
    double const next = t0 + o->period_s;
    while(now_s() < next && !o->stop){
      struct timespec ts = {0, 100000};
      nanosleep(&ts, NULL);
    }
  }
  if(away >= 0 && !o->failed)             /* everybody is back when the receiver checks its last delivery */
    o->failed |= kq_bank_add_channel(bank, &o->cc[away]) != away;
  return NULL;
}
static int cmp_double(const void *a, const void *b){
  double const x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int main(int argc, char **argv){
  unsigned const C = argc > 1 ? (unsigned)atoi(argv[1]) : 32768u;
  unsigned const B = argc > 2 ? (unsigned)atoi(argv[2]) : 2u;
  double const seconds = argc > 3 ? atof(argv[3]) : 3.0;
  int const pcm = argc > 4 ? atoi(argv[4]) : 1;
  int const operator_on = argc > 5 ? atoi(argv[5]) : 0;
  int const paced = argc > 6 ? atoi(argv[6]) : 0;
  if(C == 0 || B == 0 || B > 64)
    return 2;
  if(kq_abi_version() != KQ_ABI_VERSION || kq_device_count() <= 0){
    fprintf(stderr, "library / device: %s\n", kq_last_error());
    return 2;
  }
  /* ---- the bank and its channels: channel c listens to emitter c mod NEMIT, a few Hz off so that no two LOs are equal */
  kq_bank_config bc = { .device = 0, .samprate = SAMPRATE, .L = L, .M = M, .decimate = D, .max_channels = C, .max_blocks = B,
                        .gain_factor = 1.0f, .compute_n0 = 1, .fwd_mode = KQ_FWD_AUTO, .pl_tone_off = 1 };
  double t0 = now_s();
  kq_bank *bank = kq_bank_create(&bc);
  if(!bank){
    fprintf(stderr, "kq_bank_create: %s\n", kq_last_error());
    return 1;
  }
  kq_channel_config *cc = calloc(C, sizeof *cc);
  for(unsigned c = 0; c < C; c++)
    cc[c] = (kq_channel_config){ .demod_type = KQ_FM_DEMOD, .channels = 1, .low = -8000, .high = 8000, .kaiser_beta = 3.0f,
                                 .headroom = 0.1778f, .second_lo = -(emitter_freq(c % NEMIT) + (c / NEMIT) * 0.25) };
  if(kq_bank_add_channels(bank, cc, C, NULL) != (int)C){
    fprintf(stderr, "kq_bank_add_channels: %s\n", kq_last_error());
    return 1;
  }
  printf("%u FM channels set up in %.2f s\n", C, now_s() - t0);

  /* ---- pinned host memory: one batch of input (a live receiver has a ring of them), three sets of output planes */
  size_t const nin = (size_t)B * L, rows = (size_t)C * B;
  float complex *in = NULL;
  void *out[3], *st[3];
  uint32_t *mask[3];
  size_t const out_bytes = rows * 2 * OLEN * (pcm ? sizeof(int16_t) : sizeof(float));
  int bad = (in = kq_host_alloc(nin * sizeof *in)) == NULL;
  for(int i = 0; i < 3; i++)
    bad |= (out[i] = kq_host_alloc(out_bytes)) == NULL || (st[i] = kq_host_alloc(rows * sizeof(kq_chan_status))) == NULL ||
           (mask[i] = kq_host_alloc(rows * sizeof(uint32_t))) == NULL;
  if(bad){
    fprintf(stderr, "kq_host_alloc: %s\n", kq_last_error());
    return 1;
  }
  double phase[NEMIT] = {0};
  unsigned lcg = 2024u;
  for(size_t i = 0; i < nin; i++){
    double const t = (double)i / SAMPRATE;
    float complex s = 0;
    for(int e = 0; e < NEMIT; e++){
      phase[e] += 2 * M_PI * (emitter_freq(e) + DEVIATION * cos(2 * M_PI * 1000. * t)) / SAMPRATE;
      s += 0.04f * ((float)cos(phase[e]) + (float)sin(phase[e]) * I);
    }
    lcg = lcg * 1664525u + 1013904223u;
    float const nr = ((lcg >> 8) & 0xffff) / 65536.f - 0.5f;
    lcg = lcg * 1664525u + 1013904223u;
    float const ni = ((lcg >> 8) & 0xffff) / 65536.f - 0.5f;
    in[i] = s + 2e-3f * (nr + ni * I);
  }

  /* ---- the receiver's loop (the same batch over and over: the stream is synthetic, the work is not) */
  if(paced){      /* a receiver thread belongs in a real-time class; an ordinary user is normally refused (said below) */
    struct sched_param sp = { .sched_priority = 1 };
    printf("scheduler: %s\n", sched_setscheduler(0, SCHED_FIFO, &sp) == 0 ? "SCHED_FIFO" : "SCHED_OTHER (SCHED_FIFO refused: the long intervals "
           "of a paced run on a shared host are this thread kept off its core)");
  }
  double const signal_s = (double)B * L / SAMPRATE;
  long calls = 0, warm = 50;
  kq_host_timing ht;
  int rc = 0;
  size_t const cap = (size_t)(seconds / signal_s * 1.5) + 1024;   /* per-call records of the timed part */
  double *stamp = malloc(cap * sizeof *stamp), *lag = malloc(cap * sizeof *lag);
  float (*step)[4] = malloc(cap * sizeof *step);       /* ms inside process / push / queueing the delivery / waiting for delivery k - 2 */
  kq_bank_enable_timing(bank, 1);
  struct operator_args op = { .bank = bank, .cc = cc, .C = C, .period_s = signal_s };
  pthread_t op_tid;
  int op_started = 0;
  double origin = 0;                      /* paced: batch n of the timed part is complete at origin + (n + 1) signal_s */
  if(kq_bank_push_iq_async(bank, in, nin, KQ_IQ_CF32) != 0)
    rc = 1;
  for(long k = 0; rc == 0; k++){
    if(k == warm){                      /* clocks up, every buffer touched once */
      kq_bank_host_io_wait(bank);
      kq_bank_get_host_timing(bank, &ht, 1);
      { kq_timing discard; kq_bank_get_timing(bank, &discard, 1); }   /* the device's side from here on too */
      if(operator_on && C > 64){
        op_started = pthread_create(&op_tid, NULL, operator_thread, &op) == 0;
        if(!op_started)
          rc = 1;
      }
      t0 = now_s();
      origin = t0 + 2e-4 - signal_s;
    }
    int const j = (int)(k % 3);
    if(paced && k >= warm){             /* the batch that is pushed below has just arrived: not before */
      double const due = origin + (double)(calls + 1) * signal_s;
      double now = now_s();
      /* sleep most of the way, spin the last 0.2 ms: a loop that only spins is a CPU hog to the scheduler, and on a host shared
       * with other jobs it was taken off its core for 5-11 ms once or twice a minute (the whole stall outside the library's calls) */
      if(due - now > 3e-4){
        double const nap = due - now - 2e-4;
        struct timespec ts = { (time_t)nap, (long)((nap - (double)(time_t)nap) * 1e9) };
        nanosleep(&ts, NULL);
        now = now_s();
      }
      while(now < due)
        now = now_s();
      if((size_t)calls < cap)
        lag[calls] = now - due;
    }
    /* (process first, then push: an input copy queued behind the previous call's output copy would wait with it for that
     * call's demodulators -- ka9q_hip.h "Call order for full overlap") */
    double const p0 = now_s();
    if(kq_bank_process(bank) != (int)B)
      rc = 1;
    double const p1 = now_s();
    if(rc == 0 && kq_bank_push_iq_async(bank, in, nin, KQ_IQ_CF32) != 0)
      rc = 1;
    double const p2 = now_s();
    if(rc == 0 && (pcm ? kq_bank_pull_pcm_planes_async(bank, out[j], mask[j], st[j]) : kq_bank_pull_planes_async(bank, out[j], st[j])))
      rc = 1;
    double const p3 = now_s();
    if(rc == 0 && kq_bank_pull_wait(bank, 2) != 0)           /* delivery k - 2 has landed: out[(k - 2) % 3] is the host's to read */
      rc = 1;
    if(k >= warm && (size_t)calls < cap){
      double const p4 = now_s();
      step[calls][0] = (float)(1e3 * (p1 - p0));
      step[calls][1] = (float)(1e3 * (p2 - p1));
      step[calls][2] = (float)(1e3 * (p3 - p2));
      step[calls][3] = (float)(1e3 * (p4 - p3));
    }
    if(op.failed)
      rc = 1;
    if(k >= warm){
      double const now = now_s();
      if((size_t)calls < cap)
        stamp[calls] = now;
      calls++;
      if(now - t0 >= seconds)
        break;
    }
  }
  double const wall = now_s() - t0;
  if(op_started){
    op.stop = 1;
    pthread_join(op_tid, NULL);
    if(op.failed)
      rc = 1;
  }
  if(rc != 0 || kq_bank_host_io_wait(bank) != 0){
    fprintf(stderr, "receiver loop: %s\n", kq_last_error());
    return 1;
  }
  if(op_started){                       /* the channel that came back last takes part in one more call before the check */
    int const j = (int)((warm + calls) % 3);
    if(rc || kq_bank_process(bank) != (int)B ||
       (pcm ? kq_bank_pull_pcm_planes_async(bank, out[j], mask[j], st[j]) : kq_bank_pull_planes_async(bank, out[j], st[j])) ||
       kq_bank_host_io_wait(bank) != 0){
      fprintf(stderr, "last call: %s\n", kq_last_error());
      return 1;
    }
  }
  kq_bank_get_host_timing(bank, &ht, 0);
  double const per_call = wall / calls;
  double const d2h = (double)rows * (OLEN * (pcm ? 2 : 4) + sizeof(kq_chan_status) + (pcm ? 4 : 0));
  printf("%u channels x %u blocks per call (%.3f ms of signal): %.4f ms per call over %ld calls = %.3f x real time%s\n", C, B,
         signal_s * 1e3, per_call * 1e3, calls, signal_s / per_call, paced ? " (paced by the clock)" : " (a mean: batches as fast as they go)");
  {                                     /* deliveries as the host sees them */
    size_t const n = (size_t)calls < cap ? (size_t)calls : cap;
    size_t const skip = paced ? (size_t)(0.5 / signal_s) < n / 4 ? (size_t)(0.5 / signal_s) : n / 4 : 0;
    double *iv = malloc(n * sizeof *iv);
    size_t niv = 0;
    for(size_t i = skip + 1; i < n; i++)
      iv[niv++] = stamp[i] - stamp[i - 1];
    qsort(iv, niv, sizeof *iv, cmp_double);
    if(niv > 0)
      printf("delivery intervals: p50 %.3f  p99 %.3f  p99.9 %.3f  max %.3f ms\n", 1e3 * iv[niv / 2], 1e3 * iv[(size_t)(0.99 * (niv - 1))],
             1e3 * iv[(size_t)(0.999 * (niv - 1))], 1e3 * iv[niv - 1]);
    {                                   /* where the longest interval went: the host's four steps of that iteration */
      size_t worst = skip + 1;
      for(size_t i = skip + 1; i < n; i++)
        if(stamp[i] - stamp[i - 1] > stamp[worst] - stamp[worst - 1])
          worst = i;
      {                                 /* when the eight longest intervals happened: a period would point at a timer */
        size_t top[8];
        size_t ntop = 0;
        for(size_t i = skip + 1; i < n; i++){
          double const d = stamp[i] - stamp[i - 1];
          if(d < 1.5 * signal_s)
            continue;
          if(i > skip + 1 && stamp[i - 1] - stamp[i - 2] >= 1.5 * signal_s)
            continue;                   /* (the iterations that catch up behind a stall are part of it) */
          if(ntop < 8)
            top[ntop++] = i;
          else {
            size_t m = 0;
            for(size_t q = 1; q < 8; q++)
              if(stamp[top[q]] - stamp[top[q] - 1] < stamp[top[m]] - stamp[top[m] - 1])
                m = q;
            if(d > stamp[top[m]] - stamp[top[m] - 1])
              top[m] = i;
          }
        }
        printf("stalls (intervals beyond 1.5 periods, the longest %zu): at", ntop);
        for(size_t q = 0; q < ntop; q++)
          printf(" %.2f s (%.2f ms)", stamp[top[q]] - t0, 1e3 * (stamp[top[q]] - stamp[top[q] - 1]));
        printf("\n");
      }
      kq_timing tm;
      kq_bank_get_timing(bank, &tm, 0);
      if(worst < n)
        printf("longest interval (%.3f ms, call %zu): process %.3f, push %.3f, queueing the delivery %.3f, waiting for delivery k-2 %.3f ms; "
               "longest filter interval in the device's queue %.3f ms (mean %.3f; pass %llu of the timed part, %.3f ms of it the host "
               "between queueing its two markers)\n", 1e3 * (stamp[worst] - stamp[worst - 1]), worst, step[worst][0],
               step[worst][1], step[worst][2], step[worst][3], tm.filter_max_ms, tm.filter_ms / (double)(tm.filter_launches ? tm.filter_launches : 1),
               (unsigned long long)tm.filter_max_launch, tm.filter_max_submit_ms);
    }
    if(paced){
      long late = 0, late_playout = 0, backlog_max = 0;
      double worst = 0;
      for(size_t i = skip; i < n; i++){
        double const due = origin + (double)(i + 1) * signal_s;
        double const late_by = stamp[i] - due - signal_s;      /* delivery i - 2 is due with the start of iteration i */
        if(late_by > 0.100)                                     /* beyond the reference player's playout buffer (monitor.c:83) */
          late_playout++;
        if(late_by > 0){
          late++;
          if(late_by > worst)
            worst = late_by;
        }
        long const bl = (long)floor(lag[i] / signal_s);
        if(bl > backlog_max)
          backlog_max = bl;
      }
      printf("deadline: %ld of %zu deliveries more than one call period behind schedule (worst %.3f ms late), deepest backlog %ld periods; "
             "beyond the reference player's 100 ms playout buffer (monitor.c:83): %ld\n", late, n - skip, 1e3 * worst, backlog_max, late_playout);
    }
    free(iv);
  }
  if(op_started)
    printf("operator thread: %ld changes (filter, channel dropped / back, retune) = %.0f per second beside the stream; host time inside "
           "them worst %.3f ms per round; the receiver waited for the handle's lock %.4f ms per call, worst %.3f ms; longest hold by "
           "another entry point %.3f ms\n", op.ops, op.ops / wall, 1e3 * op.worst_s, ht.lock_wait_ms / (double)ht.calls,
           ht.lock_wait_max_ms, ht.ctl_hold_max_ms);
  printf("host inside kq_bank_process: %.4f ms per call (per-channel staging %.4f, waiting for the device %.4f); delivered %s + status: %.2f GB/s\n",
         ht.call_ms / (double)ht.calls, ht.stage_ms / (double)ht.calls, ht.slot_wait_ms / (double)ht.calls,
         pcm ? "int16 PCM" : "float audio", d2h / per_call / 1e9);
  free(stamp);
  free(lag);
  free(step);

  /* ---- a delivered plane: every channel reports olen samples per block; channel 5 sits on a carrier */
  int const jl = (int)((warm + calls - 1 + (op_started ? 1 : 0)) % 3);
  const kq_chan_status *s = st[jl];
  unsigned open = 0;
  size_t counted = 0;
  for(size_t r = 0; r < rows; r++){
    if(operator_on && r / B >= 64)     /* a channel that has just come back or had its filter changed may still be settling */
      continue;
    counted++;
    if(s[r].nout != OLEN)
      rc = 3;
    open += s[r].squelch_count < 2;
  }
  kq_chan_status const *s5 = &s[(size_t)(5 % C) * B];
  printf("last delivery: %u of %zu channel-blocks with the squelch open; channel %u: snr %.0f, pdeviation %.0f Hz, n0 %.3g\n",
         open, counted, 5 % C, s5->snr, s5->pdeviation, s5->n0);
  if(open != counted || !(s5->snr > 20.f) || !(s5->pdeviation > 0.3f * DEVIATION && s5->pdeviation < 2.f * DEVIATION))
    rc = 4;
  if(pcm){                                       /* the PCM words of an open FM channel are not silence */
    const int16_t *w = (const int16_t *)out[jl] + (size_t)(5 % C) * B * 2 * OLEN;
    int nz = 0;
    for(int i = 0; i < OLEN; i++)
      nz += w[i] != 0;
    if(nz == 0 || mask[jl][(size_t)(5 % C) * B] != 0)
      rc = 5;
  }
  kq_bank_destroy(bank);
  free(cc);
  kq_host_free(in);
  for(int i = 0; i < 3; i++){
    kq_host_free(out[i]);
    kq_host_free(st[i]);
    kq_host_free(mask[i]);
  }
  puts(rc == 0 ? "ok" : "unexpected result");
  return rc;
}
