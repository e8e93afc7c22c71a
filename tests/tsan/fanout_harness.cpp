// ThreadSanitizer drive of the multi-GPU fan-out with a world of MORE THAN ONE rank: the library's own
// kq_fanout.cpp, compiled unchanged for the CPU against mock_async/hip/hip_runtime.h (asynchronous streams and events)
// and loading mock_rccl.cpp's thread-based librccl through KQ_RCCL_LIB.  One thread per rank, as one process (or thread)
// per GPU would run, each with its own "device", consumer stream and fan-out, in the call order bench.py and
// examples/radio_fanout.c use:
//     post(0); post(1);  for k: { p = acquire(k & 1); consume(p); release(k & 1); post(k & 1) for batch k + 2 }
// Checked: every rank reads the root's batch k at step k (contents), no slot is overwritten before its release and no
// slot is read before its batch has landed (either is a data race on plain memory: ThreadSanitizer fails the run),
// kq_fanout_stats reports the world RCCL holds, a rank whose set-up fails takes every rank's create down with it and
// nobody waits for ever.  Replaces multicast.c:143-237's fan-out; test infrastructure only.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ka9q_hip.h"

static thread_local std::string g_err;
void kq_internal_set_error(const char *fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
extern "C" void mock_rccl_fail_init_of_rank(int rank);  // resolved from the stand-in library at run time (dlsym below)

static std::atomic<long> g_bad{0}, g_checked{0};
#define CHECK(c)                                                              \
  do {                                                                        \
    if (!(c)) {                                                               \
      fprintf(stderr, "CHECK failed %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, g_err.c_str()); \
      g_bad++;                                                                \
    }                                                                         \
  } while (0)

struct Run {
  int world, steps;
  size_t n;
  std::vector<float2> batches;  // root's source: batch k at k * n, sample j = (k, j)
  char id[KQ_FANOUT_ID_BYTES];
  int fail_malloc_rank = -1;    // that rank's second hipMalloc inside kq_fanout_create fails
  bool expect_null = false;
};

static void rank_main(Run *r, int rank, kq_fanout_info *info_out) {
  hipSetDevice(rank);
  hipStream_t cs;
  hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (rank == r->fail_malloc_rank) mock_fail_malloc_in() = 2;
  kq_fanout *f = kq_fanout_create(rank, rank, r->world, 0, r->world > 1 ? r->id : nullptr, r->n);
  mock_fail_malloc_in() = 0;
  if (r->expect_null) {
    CHECK(f == nullptr);
    if (f) kq_fanout_destroy(f);
    hipStreamDestroy(cs);
    return;
  }
  CHECK(f != nullptr);
  if (!f) {
    hipStreamDestroy(cs);
    return;
  }
  if (rank & 1) CHECK(kq_fanout_enable_timing(f, 1) == 0);  // half the ranks time their waits
  std::mt19937 rng(1234 + rank);
  auto src = [&](int k) -> const void * { return rank == 0 ? (const void *)(r->batches.data() + (size_t)k * r->n) : nullptr; };
  for (int k = 0; k < 2 && k < r->steps; k++) CHECK(kq_fanout_post(f, k, src(k), r->n, 0) == 0);
  for (int k = 0; k < r->steps; k++) {
    int const slot = k & 1;
    size_t n = 0;
    const float2 *p = static_cast<const float2 *>(kq_fanout_acquire(f, slot, cs, &n));
    CHECK(p != nullptr && n == r->n);
    // the consumer: what kq_bank_process_resident queues on the bank's stream -- reads the whole slot, takes its time
    unsigned const busy_us = (rng() % 16 == 0) ? 300 + rng() % 1500 : rng() % 120;
    size_t const cnt = r->n;
    cs->enqueue([p, cnt, k, busy_us] {
      long bad = 0;
      for (size_t j = 0; j < cnt; j++) bad += (p[j].x != (float)k) | (p[j].y != (float)j);
      if (busy_us) usleep(busy_us);
      for (size_t j = 0; j < cnt; j += 7) bad += p[j].x != (float)k;  // still batch k when the "kernel" ends
      if (bad) g_bad += bad;
      g_checked++;
    });
    CHECK(kq_fanout_release(f, slot, cs) == 0);
    if (k + 2 < r->steps) CHECK(kq_fanout_post(f, slot, src(k + 2), r->n, 0) == 0);
    if (rng() % 64 == 0) usleep(rng() % 400);  // the host thread of a rank stalls now and then
  }
  hipStreamSynchronize(cs);
  kq_fanout_info info;
  CHECK(kq_fanout_stats(f, &info) == 0);
  *info_out = info;
  CHECK(kq_fanout_destroy(f) == 0);
  hipStreamDestroy(cs);
}

static void run_world(Run &r, std::vector<kq_fanout_info> &infos) {
  if (r.world > 1) {
    if (kq_fanout_unique_id(r.id) != 0) {
      fprintf(stderr, "kq_fanout_unique_id: %s\n", g_err.c_str());
      exit(2);
    }
  }
  infos.assign(r.world, kq_fanout_info{});
  std::vector<std::thread> th;
  for (int rank = 0; rank < r.world; rank++) th.emplace_back(rank_main, &r, rank, &infos[rank]);
  for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
  int const world = argc > 1 ? atoi(argv[1]) : 8, steps = argc > 2 ? atoi(argv[2]) : 2000;
  std::vector<kq_fanout_info> infos;
  Run r;
  r.world = world;
  r.steps = steps;
  r.n = 96;
  r.batches.resize((size_t)steps * r.n);
  for (int k = 0; k < steps; k++)
    for (size_t j = 0; j < r.n; j++) r.batches[(size_t)k * r.n + j] = make_float2((float)k, (float)j);

  // 1. the steady state: `world` ranks, `steps` batches, jittered consumers
  run_world(r, infos);
  unsigned long long bc = 0, waits = 0;
  for (int rank = 0; rank < world; rank++) {
    CHECK(infos[rank].rccl_ranks == world && infos[rank].world == world && infos[rank].rank == rank);
    CHECK(infos[rank].acquires == (unsigned long long)steps);
    bc += infos[rank].broadcasts;
    waits += infos[rank].waits;
    if (!(rank & 1)) CHECK(infos[rank].waits == 0);
  }
  CHECK(g_checked.load() == (long)world * steps);
  CHECK(bc > 0);
  printf("world %d: %ld consumer passes checked, %llu timed broadcasts, %llu timed waits\n", world, g_checked.load(), bc, waits);

  // 2. a world of one without a communicator (the single-GPU host): in-place re-posts move nothing, posts from a source copy
  {
    Run one = r;
    one.world = 1;
    one.steps = steps < 200 ? steps : 200;
    g_checked = 0;
    run_world(one, infos);
    CHECK(infos[0].rccl_ranks == 0 && infos[0].broadcasts == 0);
    CHECK(g_checked.load() == one.steps);
  }

  // 3. one rank's own set-up fails (its second hipMalloc): it still enters the communicator, the ranks agree, and EVERY
  //    rank's kq_fanout_create returns NULL -- nobody is left waiting in ncclCommInitRank or holding a dead fan-out
  {
    Run bad = r;
    bad.steps = 0;
    bad.fail_malloc_rank = world - 1;
    bad.expect_null = true;
    run_world(bad, infos);
  }

  // 4. the communicator itself cannot be formed on one rank: every rank errors out
  if (world > 1) {
    Run bad = r;
    bad.steps = 0;
    bad.expect_null = true;
    mock_rccl_fail_init_of_rank(world / 2);
    run_world(bad, infos);
  }

  // 5. and after all that a fresh world still works (nothing was left behind)
  {
    Run again = r;
    again.steps = steps < 100 ? steps : 100;
    g_checked = 0;
    run_world(again, infos);
    CHECK(g_checked.load() == (long)world * again.steps);
  }
  if (g_bad.load()) {
    fprintf(stderr, "%ld failures\n", g_bad.load());
    return 1;
  }
  printf("fan-out protocol: ok\n");
  return 0;
}
