// kq_lane.hpp -- cross-lane exchange lane <-> lane ^ M of a wave64 without the LDS crossbar: DPP for M < 16
// (quad permutes, row shifts with bank masks, row rotate), v_permlane16_swap / v_permlane32_swap (gfx950) above.
// A ds_bpermute costs an LDS round trip; in the single-wave demodulators forty of them in a dependent chain were
// the whole block latency.
#pragma once
#include <hip/hip_runtime.h>

namespace kq {

template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ int dpp_mov(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANK_MASK, false);
}

// value of lane (lane ^ M); `lane` is the caller's lane id (only read for M >= 16)
template <int M>
__device__ __forceinline__ int lane_xor_i(int v, int lane) {
  static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "power of two below 64");
  if constexpr (M == 1) {
    return dpp_mov<0xB1>(v, v);  // quad_perm [1,0,3,2]
  } else if constexpr (M == 2) {
    return dpp_mov<0x4E>(v, v);  // quad_perm [2,3,0,1]
  } else if constexpr (M == 4) {
    int r = dpp_mov<0x104, 0x5>(v, v);  // row_shl:4 into banks 0 and 2 (lanes with bit 2 clear take lane + 4)
    return dpp_mov<0x114, 0xA>(r, v);   // row_shr:4 into banks 1 and 3
  } else if constexpr (M == 8) {
    return dpp_mov<0x128>(v, v);  // row_ror:8
  } else if constexpr (M == 16) {
    auto const r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    // r[0] = [row0, row0, row2, row2], r[1] = [row1, row1, row3, row3] of v
    return (lane & 16) ? (int)r[0] : (int)r[1];
  } else {
    auto const r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    // r[0] = [low half, low half], r[1] = [high half, high half]
    return (lane & 32) ? (int)r[0] : (int)r[1];
  }
}

template <int M>
__device__ __forceinline__ float lane_xor(float v, int lane) {
  return __int_as_float(lane_xor_i<M>(__float_as_int(v), lane));
}

}  // namespace kq
