// kq_energy.hpp -- the IF-power recurrence over a call's blocks (radio.c:143-145), one wave: shared by the stand-alone
// kernel k_block_energy_iir (kq_kernels.hip) and by the full-spectrum filter kernel, one wave of whose first workgroup runs
// it on the side (kq_full16k.hip: the stand-alone launch was 6.6 us between every call's filter pass and its demodulators).
#pragma once
#include <hip/hip_runtime.h>

namespace kq {

// what the wave needs: the partial sums k_block_energy_sum left (`split` per block, in order), the per-block update flags,
// the accumulator carried from call to call (state[0] = E, state[1] = the last if_power) and the plane of results
struct IirArgs {
  const float *sums = nullptr;  // null: nothing to do
  const unsigned char *update = nullptr;
  float *state = nullptr;
  float *if_power = nullptr;
  int split = 0, nblocks = 0, L = 0;
};

#ifdef __HIPCC__  // (the struct above is also seen by host-only builds: tests/tsan compiles kq_device.hpp with g++)
// E <- 0.5*(E + sum); if_power = E / L  (the accumulator is halved, never cleared: radio.c:143-145).
// A block whose last sample came from the lost-packet zero fill completes inside radio.c:94-98,
// which runs the filter but leaves block_energy and if_power alone: update[b] == 0 marks those.
// Called by all 64 lanes of ONE wave (lane = 0..63).
__device__ __forceinline__ void block_energy_iir_wave(const float *__restrict__ sums, int split, const unsigned char *__restrict__ update,
                                                      int nblocks, int L, float *__restrict__ state, float *__restrict__ if_power,
                                                      int lane) {
  // fetch 64 blocks' sums and flags at a time in parallel, run the (inherently serial) recurrence out of registers via
  // readlane -- two additions' worth per block -- and divide once, in parallel, at the end
  float e = state[0], last = state[1];
  for (int base = 0; base < nblocks; base += 64) {
    int const i = base + lane;
    float sm = 0.f;
    if (i < nblocks)
      for (int k = 0; k < split; k++) sm += sums[i * split + k];  // the parts in order
    int const up = i < nblocks ? update[i] : 0;
    float e_upd = 0.f, mine_e = 0.f;  // the accumulator as the last updating block left it; the same as of this lane's block
    bool any = false, mine_any = false;
    int const cnt = min(64, nblocks - base);
    for (int k = 0; k < cnt; k++) {  // k is wave-uniform: v_readlane, not a trip through the LDS crossbar
      e += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), k));
      if (__builtin_amdgcn_readlane(up, k)) {
        e *= 0.5f;
        e_upd = e;
        any = true;
      }
      if (lane == k) {
        mine_e = e_upd;
        mine_any = any;
      }
    }
    float const mine = mine_any ? mine_e / L : last;  // blocks before the chunk's first update keep what came before
    if (i < nblocks) if_power[i] = mine;
    if (any) last = e_upd / L;
  }
  if (lane == 0) {
    state[0] = e;
    state[1] = last;
  }
}
#endif  // __HIPCC__

}  // namespace kq
