/* radio_fanout.c -- several GPUs from one plain-C process: one thread per GPU, channels sharded, every batch of front-end
 * I/Q broadcast from rank 0 by the library's fan-out (kq_fanout_*: ncclBroadcast over RCCL / xGMI on a side stream, two
 * slots) -- what the reference does with one `radio` process per channel behind a UDP multicast group
 * (multicast.c:143-237, README.md:470-477).  INTEGRATION.md section F.
 *
 *   radio_fanout [world [batches [channels_total [timed_steps]]]]      world = number of GPUs (threads), default 1
 *
 * timed_steps > 0 adds a throughput phase behind the checked batches -- the two slots re-broadcast in place, no results
 * pulled -- and prints the per-rank table that `bench.py --gpus N` carries as per_rank: step time, filter-kernel time,
 * broadcast time and the time the rank's stream stood still waiting for a batch.  A C-only host reads a sub-linear scaling
 * result from it: one slow rank (kernel_ms), a slow link (bcast_ms), or ranks starved by the root (wait_ms).
 *
 * Rank 0 synthesises a 10 MS/s stream with FM carriers 140 kHz apart; the channels (FM, +-8 kHz) are dealt to the ranks with
 * kq_shard_range.  Every rank sees the same front-end samples, so every rank must report the same IF power (radio.c:143-145)
 * for every block -- compared bit for bit across the ranks at the end -- and a channel on a carrier must see it (fm.c:100-103's
 * SNR estimate; a block of 32 output samples is shorter than the 1 kHz tone's period, so its peak deviation moves around 3 kHz).
 *
 *   gcc -std=gnu11 -O2 -pthread -Iinclude examples/radio_fanout.c -Lka9q_sdr_amd/lib -lka9q_hip \
 *       -Wl,-rpath,$PWD/ka9q_sdr_amd/lib -lm -o radio_fanout
 */
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ka9q_hip.h"

enum { L = 8192, M = 8193, D = 256, NBLOCKS = 4, SAMPRATE = 10000000 };
#define NEMIT 16
#define DEVIATION 3000.0

struct shared {
  int world, batches, timed;
  pthread_barrier_t start;   /* the timed phase begins on every rank at once */
  unsigned channels_total;
  unsigned char id[KQ_FANOUT_ID_BYTES];
  float complex *stream;     /* M-1 zeros, then batches * NBLOCKS * L samples */
};
struct rank_ctx {
  struct shared *sh;
  int rank, rc;
  pthread_t thread;
  float *if_power;           /* [batches][NBLOCKS] of this rank's first channel */
  float pdev_first, snr_first; /* deviation and SNR measured by this rank's first channel in the last block */
  unsigned first, count;
  kq_fanout_info info;
  /* timed phase */
  double ms_per_step, kernel_ms, bcast_ms, wait_ms_per_step;
  unsigned long long waits, waits_untimed;
  char err[256];
};

static double now_ms(void){
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec;
}

static double emitter_freq(int e){ return (e - (NEMIT - 1) / 2.0) * 140000.0; }

static void *rank_main(void *arg){
  struct rank_ctx *r = arg;
  struct shared *sh = r->sh;
  size_t const nwin = (size_t)(M - 1) + (size_t)NBLOCKS * L, nnew = (size_t)NBLOCKS * L;
  int const ndev = kq_device_count();
  int const device = ndev > 0 ? r->rank % ndev : 0;   /* one GPU per rank when there are enough (RCCL refuses to share one) */
  kq_fanout *fan = NULL;
  kq_bank *bank = NULL;
  int met = 0;               /* this rank has been to the timed phase's barrier */
  r->rc = 1;
  /* COLLECTIVE: every rank's thread is in here at the same time; either all of them get a fan-out or none does */
  fan = kq_fanout_create(device, r->rank, sh->world, 0, sh->world > 1 ? sh->id : NULL, nwin);
  if(!fan){
    snprintf(r->err, sizeof r->err, "kq_fanout_create: %s", kq_last_error());
    goto done;
  }
  if(kq_shard_range(sh->channels_total, sh->world, r->rank, &r->first, &r->count) != 0 || r->count == 0){
    snprintf(r->err, sizeof r->err, "kq_shard_range: %s", kq_last_error());
    goto done;
  }
  kq_bank_config bc = { .device = device, .samprate = SAMPRATE, .L = L, .M = M, .decimate = D, .max_channels = r->count,
                        .max_blocks = NBLOCKS, .gain_factor = 1.0f, .compute_n0 = 1, .fwd_mode = KQ_FWD_AUTO };
  bank = kq_bank_create(&bc);
  if(!bank){
    snprintf(r->err, sizeof r->err, "kq_bank_create: %s", kq_last_error());
    goto done;
  }
  for(unsigned c = r->first; c < r->first + r->count; c++){       /* channel c listens to emitter c mod NEMIT */
    kq_channel_config cc = { .demod_type = KQ_FM_DEMOD, .channels = 1, .low = -8000, .high = 8000, .kaiser_beta = 3.0f,
                             .headroom = 0.1778f, .second_lo = -emitter_freq(c % NEMIT) };
    if(kq_bank_add_channel(bank, &cc) < 0){
      snprintf(r->err, sizeof r->err, "kq_bank_add_channel: %s", kq_last_error());
      goto done;
    }
  }
  void *bs = kq_bank_stream(bank);
  /* batch k = the window [M-1 history | NBLOCKS * L new samples]; only rank 0 has the samples */
#define WINDOW(k) (r->rank == 0 ? (const void *)(sh->stream + (size_t)(k) * nnew) : NULL)
  for(int k = 0; k < 2 && k < sh->batches; k++)
    if(kq_fanout_post(fan, k, WINDOW(k), nwin, 0) != 0){
      snprintf(r->err, sizeof r->err, "kq_fanout_post: %s", kq_last_error());
      goto done;
    }
  for(int k = 0; k < sh->batches; k++){
    int const slot = k & 1;
    size_t got = 0;
    const void *win = kq_fanout_acquire(fan, slot, bs, &got);         /* the bank's stream waits for the batch */
    if(!win || got != nwin || kq_bank_process_resident(bank, win, NBLOCKS) != NBLOCKS
       || kq_fanout_release(fan, slot, bs) != 0){                      /* the slot may be overwritten once this is reached */
      snprintf(r->err, sizeof r->err, "batch %d: %s", k, kq_last_error());
      goto done;
    }
    if(k + 2 < sh->batches && kq_fanout_post(fan, slot, WINDOW(k + 2), nwin, 0) != 0){   /* travels under batch k + 1 */
      snprintf(r->err, sizeof r->err, "kq_fanout_post: %s", kq_last_error());
      goto done;
    }
    /* results of batch k (this waits for the bank; the broadcast of batch k + 2 is already on its way) */
    for(unsigned b = 0; b < NBLOCKS; b++){
      kq_chan_status st;
      if(kq_bank_pull_status(bank, 0, b, &st) != 0){
        snprintf(r->err, sizeof r->err, "kq_bank_pull_status: %s", kq_last_error());
        goto done;
      }
      r->if_power[k * NBLOCKS + b] = st.if_power;
      r->pdev_first = st.pdeviation;
      r->snr_first = st.snr;
    }
  }
  if(kq_bank_sync(bank) != 0 || kq_fanout_stats(fan, &r->info) != 0){
    snprintf(r->err, sizeof r->err, "sync: %s", kq_last_error());
    goto done;
  }
  if(sh->timed > 0){
    /* throughput: both slots hold a batch (rank 0 re-broadcasts each in place while the other is consumed) */
    kq_fanout_info before = r->info, after;
    kq_timing tm;
    const void *held[2] = { NULL, NULL };
    int bad = 0;
    for(int k = 0; k < 2; k++)
      bad |= kq_fanout_post(fan, k, WINDOW(0), nwin, 0);
    bad |= kq_bank_enable_timing(bank, 1) | kq_bank_get_timing(bank, &tm, 1) | kq_fanout_enable_timing(fan, 1);
    pthread_barrier_wait(&sh->start);
    met = 1;
    double const t0 = now_ms();
    for(int k = 0; k < sh->timed && !bad; k++){
      int const slot = k & 1;
      size_t got = 0;
      held[slot] = kq_fanout_acquire(fan, slot, bs, &got);
      bad |= !held[slot] || kq_bank_process_resident(bank, held[slot], NBLOCKS) != NBLOCKS || kq_fanout_release(fan, slot, bs) != 0;
      if(!bad)
        bad |= kq_fanout_post(fan, slot, held[slot], nwin, 1);    /* refill this slot for step k + 2 while step k + 1 computes */
    }
    bad |= kq_bank_sync(bank);
    double const t1 = now_ms();
    bad |= kq_bank_get_timing(bank, &tm, 1) | kq_fanout_stats(fan, &after) | kq_fanout_enable_timing(fan, 0) | kq_bank_enable_timing(bank, 0);
    if(bad){
      snprintf(r->err, sizeof r->err, "timed phase: %s", kq_last_error());
      goto done;
    }
    unsigned long long const nb = after.broadcasts - before.broadcasts;
    r->ms_per_step = (t1 - t0) / sh->timed;
    r->kernel_ms = tm.filter_launches ? tm.filter_ms / (double)tm.filter_launches : 0;
    r->bcast_ms = nb ? (after.broadcast_ms - before.broadcast_ms) / (double)nb : 0;
    r->wait_ms_per_step = (after.wait_ms - before.wait_ms) / sh->timed;
    r->waits = after.waits - before.waits;
    r->waits_untimed = after.waits_dropped - before.waits_dropped;
  }
  r->rc = 0;
done:
  if(sh->timed > 0 && !met)                    /* a rank that failed on the way still meets the others: nobody waits for ever */
    pthread_barrier_wait(&sh->start);
  if(bank)
    kq_bank_destroy(bank);
  kq_fanout_destroy(fan);
  return NULL;
}

int main(int argc, char **argv){
  struct shared sh = { .world = argc > 1 ? atoi(argv[1]) : 1, .batches = argc > 2 ? atoi(argv[2]) : 4,
                       .channels_total = argc > 3 ? (unsigned)atoi(argv[3]) : 0, .timed = argc > 4 ? atoi(argv[4]) : 0 };
  if(sh.world < 1 || sh.world > 64 || sh.batches < 1 || sh.timed < 0)
    return 2;
  if(kq_abi_version() != KQ_ABI_VERSION){
    fprintf(stderr, "libka9q_hip.so has ABI revision %d, this program was built for %d\n", kq_abi_version(), KQ_ABI_VERSION);
    return 2;
  }
  pthread_barrier_init(&sh.start, NULL, (unsigned)sh.world);
  if(sh.channels_total == 0)
    sh.channels_total = 24u * sh.world + 3;          /* uneven on purpose: the first ranks hold one channel more */
  if(kq_device_count() <= 0){
    fprintf(stderr, "no HIP device: %s\n", kq_last_error());
    return 2;
  }
  if(sh.world > 1 && kq_fanout_unique_id(sh.id) != 0){   /* one identifier for the world; threads share it through memory */
    fprintf(stderr, "kq_fanout_unique_id: %s\n", kq_last_error());
    return 1;
  }
  size_t const nnew = (size_t)sh.batches * NBLOCKS * L;
  sh.stream = calloc((M - 1) + nnew, sizeof *sh.stream);
  double phase[NEMIT] = {0};
  unsigned lcg = 2024u;
  for(size_t i = 0; i < nnew; i++){
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
    sh.stream[(M - 1) + i] = s + 2e-3f * (nr + ni * I);
  }
  struct rank_ctx *ranks = calloc(sh.world, sizeof *ranks);
  for(int r = 0; r < sh.world; r++){
    ranks[r].sh = &sh;
    ranks[r].rank = r;
    ranks[r].if_power = calloc((size_t)sh.batches * NBLOCKS, sizeof(float));
    pthread_create(&ranks[r].thread, NULL, rank_main, &ranks[r]);
  }
  int rc = 0;
  for(int r = 0; r < sh.world; r++){
    pthread_join(ranks[r].thread, NULL);
    if(ranks[r].rc != 0){
      fprintf(stderr, "rank %d: %s\n", r, ranks[r].err);
      rc = 1;
    }
  }
  for(int r = 0; r < sh.world && rc == 0; r++){
    struct rank_ctx *x = &ranks[r];
    printf("rank %d: channels %u..%u  rccl ranks %d  if_power[last] %.6g  first channel: snr %.0f, pdeviation %.0f Hz\n", r, x->first,
           x->first + x->count - 1, x->info.rccl_ranks, x->if_power[sh.batches * NBLOCKS - 1], x->snr_first, x->pdev_first);
    if(memcmp(x->if_power, ranks[0].if_power, sizeof(float) * sh.batches * NBLOCKS) != 0){
      printf("rank %d saw different front-end samples than rank 0\n", r);
      rc = 3;
    }
    if(x->info.rccl_ranks != (sh.world > 1 ? sh.world : 0))
      rc = 4;
    if(!(x->if_power[sh.batches * NBLOCKS - 1] > 0))
      rc = 5;
    if(!(x->snr_first > 20.f) || !(x->pdev_first > 0.3f * (float)DEVIATION && x->pdev_first < 2.f * (float)DEVIATION))
      rc = 6;
  }
  if(rc == 0 && sh.timed > 0){
    double slowest = 0;
    printf("timed: %d steps of %d blocks, librccl %s\n", sh.timed, NBLOCKS, sh.world > 1 ? kq_fanout_rccl_path() : "not used (world 1)");
    printf("rank  ms_per_step  kernel_ms  bcast_ms  wait_ms_per_step  waits  waits_untimed\n");
    for(int r = 0; r < sh.world; r++){
      struct rank_ctx *x = &ranks[r];
      printf("%4d  %11.4f  %9.4f  %8.4f  %16.4f  %5llu  %13llu\n", r, x->ms_per_step, x->kernel_ms, x->bcast_ms, x->wait_ms_per_step,
             x->waits, x->waits_untimed);
      if(x->ms_per_step > slowest)
        slowest = x->ms_per_step;
    }
    printf("all ranks: %.1f M channel-samples/s (%u channels x %d samples per step / %.4f ms, the slowest rank's step)\n",
           sh.channels_total * (double)NBLOCKS * L / slowest / 1e3, sh.channels_total, NBLOCKS * L, slowest);
  }
  puts(rc == 0 ? "ok" : "unexpected result");
  return rc;
}
