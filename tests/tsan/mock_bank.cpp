// A stand-in channel bank for the CPU ThreadSanitizer run of examples/radio_fanout.c with a world of eight
// (tests/tsan/Makefile `example`): the entry points of include/ka9q_hip.h that program calls, on the asynchronous mock
// streams of mock_async/hip/hip_runtime.h.  kq_bank_process_resident queues ONE operation on the bank's stream that
// reads the whole window it was handed (as the filter kernel does) and leaves the reference's IF power
// (radio.c:143-145: E <- (E + sum |s|^2) / 2, E / L) in the status plane.  Test infrastructure only.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
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

struct kq_bank {
  kq_bank_config cfg;
  hipStream_t stream = nullptr;
  unsigned channels = 0;
  float energy = 0;
  std::vector<kq_chan_status> status;  // [max_blocks], written on the stream, read after a drain
};

extern "C" {
const char *kq_last_error(void) { return g_err.c_str(); }
int kq_device_count(void) { return 64; }
kq_bank *kq_bank_create(const kq_bank_config *c) {
  kq_bank *b = new kq_bank();
  b->cfg = *c;
  hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  b->status.resize(c->max_blocks);
  return b;
}
int kq_bank_destroy(kq_bank *b) {
  if (!b) return 0;
  hipStreamDestroy(b->stream);
  delete b;
  return 0;
}
int kq_bank_add_channel(kq_bank *b, const kq_channel_config *) { return (int)b->channels++; }
void *kq_bank_stream(kq_bank *b) { return b->stream; }
int kq_bank_process_resident(kq_bank *b, const void *iq, unsigned nblocks) {
  if (nblocks > b->cfg.max_blocks) return -1;
  const float2 *x = static_cast<const float2 *>(iq);
  b->stream->enqueue([b, x, nblocks] {
    unsigned const L = b->cfg.L, M = b->cfg.M;
    for (unsigned k = 0; k < nblocks; k++) {
      float e = 0;
      for (unsigned i = 0; i < L; i++) {
        float2 const s = x[(size_t)(M - 1) + (size_t)k * L + i];
        e += s.x * s.x + s.y * s.y;
      }
      float h = 0;  // the history is read too
      for (unsigned i = 0; i < M - 1; i += 64) h += x[(size_t)k * L + i].x;
      b->energy = 0.5f * (b->energy + e) + 0.f * h;
      kq_chan_status st;
      memset(&st, 0, sizeof st);
      st.if_power = b->energy / (float)L;
      st.pdeviation = 3000.f;
      st.snr = 100.f;
      st.nout = (int)(L / b->cfg.decimate);
      b->status[k] = st;
    }
  });
  return (int)nblocks;
}
int kq_abi_version(void) { return KQ_ABI_VERSION; }
int kq_bank_enable_timing(kq_bank *, int) { return 0; }
int kq_bank_get_timing(kq_bank *, kq_timing *t, int) {
  memset(t, 0, sizeof *t);
  return 0;
}
int kq_bank_sync(kq_bank *b) {
  hipStreamSynchronize(b->stream);
  return 0;
}
int kq_bank_pull_status(kq_bank *b, int, unsigned blk, kq_chan_status *st) {
  hipStreamSynchronize(b->stream);
  *st = b->status[blk];
  return 0;
}
}
