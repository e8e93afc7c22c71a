// kq_fanout.cpp -- front-end I/Q fan-out for a C host on several GPUs (include/ka9q_hip.h, kq_fanout_*).
//
// The reference fans the front-end stream out to its one-process-per-channel receivers by UDP multicast
// (multicast.c:143-237).  Here the channels are sharded over one process (or thread) per GPU and every batch of
// front-end samples is broadcast from the ingest rank with ncclBroadcast -- RCCL over xGMI -- on a side stream,
// double buffered: batch k+1 travels while batch k is processed.  Channels are independent, so this is the only
// exchange on the path.  Same protocol as ka9q_sdr_amd/shard.py (which bench.py drives through torch.distributed).
//
// librccl is loaded on first use (dlopen), so that a single-GPU host never maps it; KQ_RCCL_LIB names another build of
// it (a path for dlopen) -- tests/tsan/ runs this file, unchanged, against a thread-based stand-in that way.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/ka9q_hip.h"
#include "kq_device.hpp"

void kq_internal_set_error(const char *fmt, ...);

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  bool ok = false;
  char path[512] = "";  // the shared object ncclBroadcast was bound from (dladdr), for kq_fanout_rccl_path
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    if (const char *named = getenv("KQ_RCCL_LIB"); named && *named) {
      r.handle = dlopen(named, RTLD_NOW | RTLD_GLOBAL);  // and nothing else: a misspelt path must not fall back silently
    } else {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
      }
    }
    if (!r.handle) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.handle, "ncclBroadcast");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.handle, "ncclGetVersion");
    r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Broadcast && r.AllReduce && r.GetErrorString;
    Dl_info info;
    if (r.Broadcast && dladdr((void *)r.Broadcast, &info) && info.dli_fname) snprintf(r.path, sizeof r.path, "%s", info.dli_fname);
  });
  return r;
}

}  // namespace

struct kq_fanout {
  int device = 0, rank = 0, world = 1, root = 0;
  size_t max_samples = 0;
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;
  float2 *buf[2] = {nullptr, nullptr};
  hipEvent_t ready[2] = {nullptr, nullptr}, freed[2] = {nullptr, nullptr};
  size_t count[2] = {0, 0};
  hipStream_t release_stream[2] = {nullptr, nullptr};  // no communicator: a release whose marker has not been recorded yet
  bool release_pending[2] = {false, false};
  // broadcast timing: events around ncclBroadcast on the side stream, read back when the slot is posted again
  hipEvent_t t0[2] = {nullptr, nullptr}, t1[2] = {nullptr, nullptr};
  bool timed[2] = {false, false};
  double bcast_ms = 0;
  unsigned long long bcasts = 0;
  // consumer-side wait (kq_fanout_enable_timing): events on the consumer's stream around the wait for `ready`
  bool time_waits = false;
  hipEvent_t w0[2] = {nullptr, nullptr}, w1[2] = {nullptr, nullptr};
  bool waited[2] = {false, false};
  double wait_ms = 0;
  unsigned long long waits = 0, acquires = 0, waits_dropped = 0;
};

namespace {
int fail(const char *what, hipError_t e) {
  kq_internal_set_error("%s: %s", what, hipGetErrorString(e));
  return -1;
}
// collects the slot's last broadcast time if it has completed (never waits)
void harvest(kq_fanout *f, int slot) {
  if (!f->timed[slot]) return;
  float ms = 0;
  if (hipEventElapsedTime(&ms, f->t0[slot], f->t1[slot]) == hipSuccess) {
    f->bcast_ms += ms;
    f->bcasts++;
  }
  f->timed[slot] = false;
}
// the same for the slot's last timed wait of the consumer
void harvest_wait(kq_fanout *f, int slot) {
  if (!f->waited[slot]) return;
  float ms = 0;
  if (hipEventElapsedTime(&ms, f->w0[slot], f->w1[slot]) == hipSuccess) {
    f->wait_ms += ms;
    f->waits++;
  }
  f->waited[slot] = false;
}
}  // namespace

extern "C" {

int kq_shard_range(unsigned total, unsigned world, unsigned rank, unsigned *first, unsigned *count) {
  if (world < 1 || rank >= world || !first || !count) {
    kq_internal_set_error("kq_shard_range: bad world / rank");
    return -1;
  }
  unsigned const base = total / world, extra = total % world;
  *first = rank * base + (rank < extra ? rank : extra);
  *count = base + (rank < extra ? 1u : 0u);
  return 0;
}

int kq_fanout_unique_id(void *id128) {
  if (!id128) return -1;
  Rccl &r = rccl();
  if (!r.ok) {
    kq_internal_set_error("librccl not available: %s", dlerror());
    return -1;
  }
  ncclUniqueId id;
  ncclResult_t const e = r.GetUniqueId(&id);
  if (e != ncclSuccess) {
    kq_internal_set_error("ncclGetUniqueId: %s", r.GetErrorString(e));
    return -1;
  }
  static_assert(sizeof id == KQ_FANOUT_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof id);
  return 0;
}

kq_fanout *kq_fanout_create(int device, int rank, int world, int root, const void *id128, size_t max_samples) {
  if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || max_samples == 0 || (world > 1 && !id128)) {
    kq_internal_set_error("kq_fanout_create: bad arguments");
    return nullptr;
  }
  kq::DeviceScope scope(device);
  kq_fanout *f = new kq_fanout();
  f->device = device;
  f->rank = rank;
  f->world = world;
  f->root = root;
  f->max_samples = max_samples;
  // every failure names its step (kq_last_error); nothing below is overwritten by a later, vaguer message
  hipError_t e = hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking);
  const char *step = "kq_fanout_create: hipStreamCreate";
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    step = "kq_fanout_create: slot allocation (hipMalloc / hipEventCreate)";
    e = hipMalloc((void **)&f->buf[i], max_samples * sizeof(float2));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ready[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->freed[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&f->t0[i]);
    if (e == hipSuccess) e = hipEventCreate(&f->t1[i]);
  }
  bool ok = e == hipSuccess;
  if (!ok) fail(step, e);
  if (world > 1 || id128 != nullptr) {  // with an id even a world of one goes through RCCL (tests)
    // The communicator is entered even when this rank's own set-up has already failed: ncclCommInitRank is collective,
    // and the other ranks would wait in theirs for ever.  Then the ranks AGREE: a one-word all-reduce (minimum of "my
    // set-up worked") tells every rank whether every rank made it; if one did not, all of them give the communicator
    // back and return NULL -- nobody is left holding a fan-out whose first broadcast can never complete.
    Rccl &r = rccl();
    // the word the ranks agree through, made BEFORE the collective set-up: once the communicator exists the all-reduce is
    // issued whatever else has failed (a rank that skipped it left the others waiting in theirs).  Slot 0 of
    // the fan-out is not in use yet and serves when it could be allocated; else four bytes of their own.
    int *flag = f->buf[0] ? reinterpret_cast<int *>(f->buf[0]) : nullptr, *own_flag = nullptr;
    if (!flag && hipMalloc((void **)&own_flag, sizeof(int)) == hipSuccess) flag = own_flag;
    if (!r.ok) {
      if (ok) kq_internal_set_error("kq_fanout_create: librccl not available%s", getenv("KQ_RCCL_LIB") ? " (KQ_RCCL_LIB)" : "");
      ok = false;  // (no communicator can exist on this rank: a world whose other ranks do load librccl waits in its
                   //  ncclCommInitRank for this one -- see the header, a subset failing there needs the caller's timeout)
    } else {
      ncclUniqueId id;
      memcpy(&id, id128, sizeof id);
      ncclResult_t const ne = r.CommInitRank(&f->comm, world, id, rank);
      if (ne != ncclSuccess) {
        if (ok) kq_internal_set_error("kq_fanout_create: ncclCommInitRank(rank %d of %d): %s", rank, world, r.GetErrorString(ne));
        f->comm = nullptr;
        ok = false;
      } else {
        int all = 0;
        // a flag that could not be allocated or set counts as "my set-up failed", but the collective is still entered
        // whenever there is memory to run it on
        bool set = flag && hipMemsetD32Async((hipDeviceptr_t)flag, ok ? 1 : 0, 1, f->side) == hipSuccess;
        if (flag && !set) {  // the word must hold a KNOWN value before the reduction reads it: second try from the host
          static const int zero = 0;
          set = hipMemcpyAsync(flag, &zero, sizeof(int), hipMemcpyHostToDevice, f->side) == hipSuccess;
          if (ok) kq_internal_set_error("kq_fanout_create: the agreement word could not be set on rank %d", rank);
          ok = false;  // (either way this rank's vote is "failed")
        }
        // with the word still undefined the collective is entered all the same (the peers are blocked in it) and its
        // result is not believed: min() over garbage could read as a yes
        bool const agreed = flag && r.AllReduce(flag, flag, 1, ncclInt32, ncclMin, f->comm, f->side) == ncclSuccess &&
                            hipMemcpyAsync(&all, flag, sizeof(int), hipMemcpyDeviceToHost, f->side) == hipSuccess &&
                            hipStreamSynchronize(f->side) == hipSuccess && set;
        if (!agreed) {
          if (ok) kq_internal_set_error("kq_fanout_create: the ranks' agreement (ncclAllReduce) failed on rank %d", rank);
          ok = false;
        } else if (!all) {
          if (ok) kq_internal_set_error("kq_fanout_create: another rank's set-up failed; every rank gives up (rank %d)", rank);
          ok = false;
        }
      }
    }
    if (own_flag) (void)hipFree(own_flag);
  }
  if (!ok) {
    kq_fanout_destroy(f);
    return nullptr;
  }
  return f;
}

int kq_fanout_destroy(kq_fanout *f) {
  if (!f) return 0;
  kq::DeviceScope scope(f->device);
  if (f->side) (void)hipStreamSynchronize(f->side);
  if (f->comm) (void)rccl().CommDestroy(f->comm);
  for (int i = 0; i < 2; i++) {
    if (f->buf[i]) (void)hipFree(f->buf[i]);
    if (f->ready[i]) (void)hipEventDestroy(f->ready[i]);
    if (f->freed[i]) (void)hipEventDestroy(f->freed[i]);
    if (f->t0[i]) (void)hipEventDestroy(f->t0[i]);
    if (f->t1[i]) (void)hipEventDestroy(f->t1[i]);
    if (f->w0[i]) (void)hipEventDestroy(f->w0[i]);
    if (f->w1[i]) (void)hipEventDestroy(f->w1[i]);
  }
  if (f->side) (void)hipStreamDestroy(f->side);
  delete f;
  return 0;
}

int kq_fanout_post(kq_fanout *f, int slot, const void *src, size_t nsamples, int src_is_device) {
  if (!f || slot < 0 || slot > 1 || nsamples == 0 || nsamples > f->max_samples || (f->rank == f->root && !src)) {
    kq_internal_set_error("kq_fanout_post: bad arguments");
    return -1;
  }
  kq::DeviceScope scope(f->device);
  harvest(f, slot);
  bool const copies = f->rank == f->root && src != f->buf[slot];
  if (!f->comm && !copies) {  // a world of one refilling its slot in place: nothing moves, nothing to order
    f->count[slot] = nsamples;
    return 0;
  }
  // the slot's previous consumer (kq_fanout_release) must be done before it is overwritten
  if (f->release_pending[slot]) {  // (a world of one puts its marker on the consumer's stream only now that it is needed)
    f->release_pending[slot] = false;
    if (hipError_t e = hipEventRecord(f->freed[slot], f->release_stream[slot]); e != hipSuccess)
      return fail("kq_fanout_post: hipEventRecord", e);
  }
  if (hipError_t e = hipStreamWaitEvent(f->side, f->freed[slot], 0); e != hipSuccess) return fail("kq_fanout_post: hipStreamWaitEvent", e);
  if (copies) {
    hipMemcpyKind const kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (hipError_t e = hipMemcpyAsync(f->buf[slot], src, nsamples * sizeof(float2), kind, f->side); e != hipSuccess)
      return fail("kq_fanout_post: copy into the slot", e);
  }
  if (f->comm) {
    (void)hipEventRecord(f->t0[slot], f->side);
    ncclResult_t const e =
        rccl().Broadcast(f->buf[slot], f->buf[slot], 2 * nsamples, ncclFloat32, f->root, f->comm, f->side);
    if (e != ncclSuccess) {
      kq_internal_set_error("kq_fanout_post: ncclBroadcast: %s", rccl().GetErrorString(e));
      return -1;
    }
    f->timed[slot] = hipEventRecord(f->t1[slot], f->side) == hipSuccess;
  }
  f->count[slot] = nsamples;
  if (hipError_t e = hipEventRecord(f->ready[slot], f->side); e != hipSuccess) return fail("kq_fanout_post: hipEventRecord", e);
  return 0;
}

const void *kq_fanout_acquire(kq_fanout *f, int slot, void *consumer_stream, size_t *nsamples) {
  if (!f || slot < 0 || slot > 1) {
    kq_internal_set_error("kq_fanout_acquire: bad arguments");
    return nullptr;
  }
  kq::DeviceScope scope(f->device);
  // a batch that has already landed needs no wait on the device (a cross-stream wait is a barrier packet: microseconds
  // of idle consumer stream even for an event that fired long ago)
  f->acquires++;
  if (hipEventQuery(f->ready[slot]) != hipSuccess) {
    hipStream_t const cs = (hipStream_t)consumer_stream;
    bool timed = f->time_waits && f->w0[slot] && f->w1[slot];
    if (timed) {
      // the slot's event pair may still be in flight (the host runs several steps ahead of a stalled device -- exactly the
      // case this diagnostic exists for): re-recording it would lose that sample, so this acquire goes untimed instead
      // and is counted
      if (f->waited[slot] && hipEventQuery(f->w1[slot]) != hipSuccess) {
        f->waits_dropped++;
        timed = false;
      } else {
        harvest_wait(f, slot);
        (void)hipEventRecord(f->w0[slot], cs);
      }
    }
    if (hipError_t e = hipStreamWaitEvent(cs, f->ready[slot], 0); e != hipSuccess) {
      fail("kq_fanout_acquire: hipStreamWaitEvent", e);
      return nullptr;
    }
    if (timed) f->waited[slot] = hipEventRecord(f->w1[slot], cs) == hipSuccess;
  }
  if (nsamples) *nsamples = f->count[slot];
  return f->buf[slot];
}

int kq_fanout_release(kq_fanout *f, int slot, void *consumer_stream) {
  if (!f || slot < 0 || slot > 1) return -1;
  kq::DeviceScope scope(f->device);
  if (!f->comm) {
    // Without a communicator the slot is only ever overwritten by a copy the same host thread posts: the marker (~5 us of
    // the consumer's stream behind a long kernel, tools/marker_probe.hip) is left until such a post needs it.  It then
    // lands behind whatever else the consumer stream has been given meanwhile, which only waits longer.
    f->release_stream[slot] = (hipStream_t)consumer_stream;
    f->release_pending[slot] = true;
    return 0;
  }
  if (hipError_t e = hipEventRecord(f->freed[slot], (hipStream_t)consumer_stream); e != hipSuccess)
    return fail("kq_fanout_release: hipEventRecord", e);
  return 0;
}

int kq_fanout_stats(kq_fanout *f, kq_fanout_info *out) {
  if (!f || !out) return -1;
  kq::DeviceScope scope(f->device);
  (void)hipStreamSynchronize(f->side);
  harvest(f, 0);
  harvest(f, 1);
  memset(out, 0, sizeof *out);
  out->world = f->world;
  out->rank = f->rank;
  out->rccl_ranks = 0;
  if (f->comm) {
    Rccl &r = rccl();
    if (r.CommCount) (void)r.CommCount(f->comm, &out->rccl_ranks);  // what RCCL itself holds the world to be
    if (r.GetVersion) (void)r.GetVersion(&out->rccl_version);
  }
  out->broadcasts = f->bcasts;
  out->broadcast_ms = f->bcast_ms;
  // (a wait still in flight on the consumer's stream is left for the next call: this one never waits for that stream)
  for (int i = 0; i < 2; i++)
    if (f->waited[i] && hipEventQuery(f->w1[i]) == hipSuccess) harvest_wait(f, i);
  out->acquires = f->acquires;
  out->waits = f->waits;
  out->wait_ms = f->wait_ms;
  out->waits_dropped = f->waits_dropped;
  return 0;
}

const char *kq_fanout_rccl_path(void) {
  Rccl &r = rccl();
  return r.handle ? r.path : "";
}

int kq_fanout_enable_timing(kq_fanout *f, int on) {
  if (!f) return -1;
  kq::DeviceScope scope(f->device);
  for (int i = 0; i < 2 && on; i++) {
    if (!f->w0[i] && hipEventCreate(&f->w0[i]) != hipSuccess) return fail("kq_fanout_enable_timing: hipEventCreate", hipGetLastError());
    if (!f->w1[i] && hipEventCreate(&f->w1[i]) != hipSuccess) return fail("kq_fanout_enable_timing: hipEventCreate", hipGetLastError());
  }
  f->time_waits = on != 0;
  return 0;
}

}  // extern "C"
