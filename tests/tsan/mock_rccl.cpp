// Thread-based stand-in for librccl, loaded by the library's own kq_fanout.cpp through KQ_RCCL_LIB in the CPU
// ThreadSanitizer run (tests/tsan/Makefile, fanout_harness.cpp): ranks are threads of one process, a communicator is a
// rendezvous among them, and a collective is -- as in RCCL -- an operation QUEUED ON THE CALLER'S STREAM that completes
// when the data has moved.  Test infrastructure only.
//
// ncclBroadcast here is deliberately loose about timing, so that the fan-out's own ordering (its events) is what keeps
// the slots apart: the root's operation copies its buffer into a staging entry and completes at once (RCCL's root also
// runs ahead of its receivers, by its FIFO depth); a receiver's operation waits for entry #seq and copies it out.  At most
// kDepth entries are outstanding; the root waits when the receivers lag that far behind.
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {
constexpr int kDepth = 4;
constexpr auto kTimeout = std::chrono::seconds(30);  // a rank that never shows up is an error, not a hung test

struct Entry {
  std::vector<char> data;
  int readers_left = 0;
  // all-reduce
  int arrived = 0, left = 0;
  long long acc = 0;
};
struct World {
  std::mutex m;
  std::condition_variable cv;
  int nranks = 0, joined = 0, alive = 0;
  bool init_failed = false;
  std::map<uint64_t, Entry> ops;
  std::atomic<int> errors{0};
};
std::mutex g_m;
std::map<uint64_t, std::shared_ptr<World>> g_worlds;
std::atomic<uint64_t> g_next_id{1};
std::atomic<int> g_fail_init_rank{-1};
// (system_clock deadline = pthread_cond_timedwait, which this libtsan intercepts; wait_for's steady clock goes through
// pthread_cond_clockwait, which it does not, and reports a double lock that is not there)
template <class Pred>
bool wait_bounded(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, Pred p) {
  return cv.wait_until(lk, std::chrono::system_clock::now() + kTimeout, p);
}
size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
}  // namespace

struct mock_nccl_comm {
  std::shared_ptr<World> w;
  int rank = 0;
  uint64_t seq = 0;  // operations issued on this communicator so far (host side, in call order)
};

extern "C" {

// test control: the next ncclCommInitRank of `rank` fails, after it has met the others (-1: nobody)
void mock_rccl_fail_init_of_rank(int rank) { g_fail_init_rank = rank; }
int mock_rccl_errors(ncclComm_t c) { return c->w->errors.load(); }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  uint64_t const v = g_next_id++;
  memcpy(id->internal, &v, sizeof v);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  uint64_t key;
  memcpy(&key, id.internal, sizeof key);
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_m);
    auto &slot = g_worlds[key];
    if (!slot) {
      slot = std::make_shared<World>();
      slot->nranks = nranks;
    }
    w = slot;
  }
  std::unique_lock<std::mutex> lk(w->m);
  if (w->nranks != nranks) return ncclInvalidArgument;
  if (g_fail_init_rank.load() == rank) {
    g_fail_init_rank = -1;
    w->init_failed = true;  // RCCL's bootstrap takes the whole world down with one rank
  }
  w->joined++;
  w->cv.notify_all();
  if (!wait_bounded(w->cv, lk, [&] { return w->joined >= w->nranks; })) return ncclSystemError;
  if (w->init_failed) return ncclSystemError;
  w->alive++;
  *comm = new mock_nccl_comm{w, rank, 0};
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclInvalidArgument;
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) {
  if (!c || !s || root < 0 || root >= c->w->nranks) return ncclInvalidArgument;
  uint64_t const seq = c->seq++;
  size_t const bytes = count * type_bytes(t);
  std::shared_ptr<World> w = c->w;
  int const rank = c->rank;
  s->enqueue([=] {
    std::unique_lock<std::mutex> lk(w->m);
    if (rank == root) {
      if (!wait_bounded(w->cv, lk, [&] { return (int)w->ops.size() < kDepth; })) {
        w->errors++;
        return;
      }
      Entry &e = w->ops[seq];
      e.readers_left = w->nranks - 1;
      lk.unlock();
      std::vector<char> tmp(bytes);
      memcpy(tmp.data(), send, bytes);  // reads the root's slot: must come after whatever filled it, on this stream
      if (recv != send) memcpy(recv, send, bytes);
      lk.lock();
      Entry &e2 = w->ops[seq];
      e2.data = std::move(tmp);
      if (e2.readers_left == 0) w->ops.erase(seq);
      lk.unlock();
      w->cv.notify_all();
    } else {
      if (!wait_bounded(w->cv, lk, [&] {
            auto it = w->ops.find(seq);
            return it != w->ops.end() && !it->second.data.empty();
          })) {
        w->errors++;
        return;
      }
      Entry &e = w->ops[seq];
      if (e.data.size() != bytes) w->errors++;  // the ranks disagree about the batch size
      std::vector<char> const &d = e.data;
      size_t const n = d.size() < bytes ? d.size() : bytes;
      // (copied under the lock: the entry is erased by the last reader; batches in the test are small)
      memcpy(recv, d.data(), n);  // writes this rank's slot: must come after its last consumer released it
      if (--e.readers_left == 0) w->ops.erase(seq);
      lk.unlock();
      w->cv.notify_all();
    }
  });
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (!c || !s || t != ncclInt32 || count != 1) return ncclInvalidArgument;  // all the fan-out asks for
  uint64_t const seq = c->seq++ | (1ull << 63);
  std::shared_ptr<World> w = c->w;
  s->enqueue([=] {
    int const mine = *static_cast<const int *>(send);
    std::unique_lock<std::mutex> lk(w->m);
    Entry &e = w->ops[seq];
    if (e.arrived == 0) {
      e.acc = mine;
      e.left = w->nranks;
    } else {
      e.acc = op == ncclMin ? (mine < e.acc ? mine : e.acc) : op == ncclMax ? (mine > e.acc ? mine : e.acc) : e.acc + mine;
    }
    e.arrived++;
    w->cv.notify_all();
    if (!wait_bounded(w->cv, lk, [&] { return w->ops[seq].arrived >= w->nranks; })) {
      w->errors++;
      return;
    }
    Entry &e2 = w->ops[seq];
    *static_cast<int *>(recv) = (int)e2.acc;
    if (--e2.left == 0) w->ops.erase(seq);
    lk.unlock();
    w->cv.notify_all();
  });
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t e) { return e == ncclSuccess ? "no error" : e == ncclSystemError ? "mock system error" : "mock error"; }
ncclResult_t ncclGetVersion(int *v) {
  *v = 99999;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) {
  *n = c->w->nranks;
  return ncclSuccess;
}
}
