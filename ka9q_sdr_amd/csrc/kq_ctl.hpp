// kq_ctl.hpp -- one write record of the bank's control queues (kq_bank.cpp CtlQueue) and the workgroup-wide routine that applies it:
// shared by k_ctl_apply (kq_kernels.hip) and by the response-design kernel (kq_design.hip), whose launch takes the filter side's
// records along when a call has both (one launch instead of two in front of the call's kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace kq {

// Record r (32 bytes at the front of the queue buffer): destination, byte count (a multiple of 4), then the offset of its payload in
// the buffer (kind 0), a 32-bit fill value (kind 1) or the device address to copy from (kind 2: a value an earlier kernel of the call
// left on the device -- the noise gain of a response designed in front of the filter pass).
struct CtlRec {
  unsigned long long dst;
  unsigned nbytes, kind, value, payload_off;
  unsigned long long src;
};

// all `nthreads` threads of a workgroup call this for record `rec` of queue buffer `q` (pinned host memory)
__device__ __forceinline__ void ctl_apply_record(const unsigned char *__restrict__ q, unsigned rec, unsigned tid, unsigned nthreads) {
  const CtlRec *r = reinterpret_cast<const CtlRec *>(q) + rec;
  unsigned *dst = reinterpret_cast<unsigned *>(r->dst);
  unsigned const n = r->nbytes >> 2;
  if (r->kind == 1) {
    unsigned const v = r->value;
    for (unsigned i = tid; i < n; i += nthreads) dst[i] = v;
    return;
  }
  // (a payload lies in host memory: every load is a trip over the link, so as few and as wide as the alignment allows --
  //  the host cuts long payloads into records of 4 KiB, one trip per thread)
  const unsigned *src = r->kind == 2 ? reinterpret_cast<const unsigned *>(r->src) : reinterpret_cast<const unsigned *>(q + r->payload_off);
  if ((((unsigned long long)(uintptr_t)dst | (unsigned long long)(uintptr_t)src | r->nbytes) & 15ull) == 0) {
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    for (unsigned i = tid; i < (n >> 2); i += nthreads) d4[i] = s4[i];
  } else {
    for (unsigned i = tid; i < n; i += nthreads) dst[i] = src[i];
  }
}

}  // namespace kq
