// kq_full16k.hip -- full-spectrum pre-detection filter for N = 16384, register-resident forward transform.
//
// Same contract as k_filter_full (kq_kernels.hip): per (channel, block) NCO mix (radio.c:132-139), the N-point forward
// transform of execute_filter_input (filter.c:151), compute_n0 over all N bins (radio.c:383-425), response multiply,
// CROSS_CONJ and the N/D-point inverse transform of execute_filter_output (filter.c:206-250).  It exists because
// compute_n0 -- which the reference's demodulator threads run on every block -- needs every bin, so the pruned
// kernels cannot serve it.
//
// N = 32 * 32 * 16.  With n = 512 n1 + 16 n2 + n3 and k = k1 + 32 k2 + 1024 k3
//   X[k] = sum_n3 W16^{n3 k3} W512^{n3 k2} sum_n2 W32^{n2 k2} W_N^{(16 n2 + n3) k1} sum_n1 W32^{n1 k1} x[n]
// 512 threads; thread t = 16 n2 + n3 loads its 32 samples x[512 n1 + t] (coalesced), mixes them and runs the
// 32-point transform over n1 in registers; the twiddle W_N^{t k1} = W^{(k1 & 3) t} W^{(k1 & ~3) t} is one product of
// two entries of a small per-thread table (coalesced loads).  Complex values are 2-vectors: the arithmetic is
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.
// Two transposes through LDS (each in two half rounds, so the buffer is 68 KiB and two workgroups share a CU) feed the 32-point transform over n2 and the two 16-point transforms over n3.  Every bin ends up in a register:
// compute_n0 is two block reductions, and the N/D bins the slave reads are dropped into LDS for the existing
// multiply / inverse-transform epilogue.
//
// NCO: without sweep the phasor of sample 512 n1 + t is P_t S^{n1}; S^{n1} is built from S, S^2, S^4, S^8, S^16
// (each from the double-precision phase), at most four products deep.  Swept channels and the first block after a
// retune (history still on the old oscillator) evaluate the closed-form phase per sample, as k_filter_full does.
//
// N = 65536 (BIG; cfg 5: compute_n0 needs every bin there too, linear.c:123-126).  64 Ki complex values are 512 KiB -- more
// than a CU's LDS, as much as its whole register file -- so the transform is split by decimation in frequency over the
// first radix-4 stage, ACROSS four sibling workgroups: bins k = 4 q + r are the 16384-point transform of
//   z_r[m] = W_N^{r m} sum_j (-i)^{r j} xm[m + 16384 j],   xm = samples times the oscillator,
// and the four residue classes need nothing from each other afterwards.  Workgroup (channel, block, r) forms z_r while it
// loads -- every sample meets its row's phasor T[j][n1] (oscillator at the row start, (-i)^{r j} and the row part of
// W_N^{r m} in one table entry per row, evaluated in double per wave); the column part rides on pass 1's twiddles with
// P_t as before -- and then runs the body below unchanged.  A sweep adds the cross term rate * (row start) * (column):
// its wave part goes into the wave's table, the lane part (below 1e-3 rad inside full64k_sweep_limit()) is applied to
// first order on the row sum.  compute_n0's first pass needs the sum over all four classes: each sibling publishes its
// part in a tagged 64-bit word (agent-scope atomic store), reads the other three (bounded spin: siblings are adjacent
// workgroup ids, dispatched together) and adds the four in a fixed order, so all four use the same threshold; the second
// pass' (sum, count) and the bins the slave reads go to global memory, where k_epilogue64k finishes them.
#include "kq_device.hpp"
#include "kq_ldsfft.hpp"
#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kq_lane.hpp"
#include "kq_regfft.hpp"

// -DKQ_TIMELINE (never in a shipped build): wave 0..7 of 64 sampled workgroups stamp the shader clock between the phases;
// tools/timeline.py reads the stamps back through kq_debug_timeline.
#ifdef KQ_TIMELINE
__device__ unsigned long long kq_timeline[64][8][12];
#define KQ_STAMP(i)                                                                        \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (blockIdx.y == (gridDim.y > 40 ? 40 : gridDim.y / 2) && (blockIdx.x & 15) == 0 && (blockIdx.x >> 4) < 64 &&      \
        (threadIdx.x & 63) == 0)                                                           \
      kq_timeline[blockIdx.x >> 4][threadIdx.x >> 6][i] = clock64();                       \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
extern "C" int kq_debug_timeline(unsigned long long *dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(kq_timeline), sizeof(kq_timeline)) == hipSuccess ? 0 : -1;
}
#else
#define KQ_STAMP(i)
#endif

namespace kq {

namespace {

constexpr int kN = 16384, kT = 512;
// Both transposes move complex (8-byte) elements in two half rounds, so the buffer holds half of the data.
constexpr int kRow1 = 528;    // transpose 1: [k1 & 15][16 n2 + n3]; rows 4224 B apart alternate 128-byte bank halves
constexpr int kCol2 = 32;     // transpose 2: [n3][32 (k1 & 15) + k2], rows kRow2 apart.  A ds_write_b64 is served in groups of
constexpr int kRow2 = 513;    //   16 consecutive lanes over 32 banks (16 eight-byte slots): the 16 n3 of a group need an odd pitch;
                              //   the reads are 32 consecutive elements per lane group whatever the pitches
constexpr int kXchElems = 16 * kRow1;  // float2 elements; 16 * kRow2 fits as well

using rfft::pk_cmul;
using rfft::v2f;

__device__ __forceinline__ v2f ld2(const float2 *p) {
  float2 const f = *p;
  return (v2f){f.x, f.y};
}

// Buffer loads: address = base (scalar registers) + lane offset (one VGPR, 32 bit) + row offset (scalar / literal).
// The 32 window rows and the twiddle rows are 4 KiB and more apart, beyond the immediate offset of a global load,
// which the compiler then serves with a 64-bit vector add (v_add_co / v_addc) per row; here no vector instruction
// is spent on addresses at all.  Reads past `bytes` return zero.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ v2f buf_ld2(rsrc_t r, unsigned lane_off, unsigned row_off) {
  return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_off, (int)row_off, 0));
}
// two adjacent complex values in one 16-byte load
__device__ __forceinline__ void buf_ld4(v2f &a, v2f &b, rsrc_t r, unsigned lane_off, unsigned row_off) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f const q = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)row_off, 0));
  a = (v2f){q.x, q.y};
  b = (v2f){q.z, q.w};
}

// Twiddle tables (float2), computed in double on the host.  Pass 1, thread t: ten entries e = 0..9 -- W_N^{l t}, l = 1..3,
// then W_N^{4 h t}, h = 1..7 -- stored in pairs, element kTabP1 + 2 (512 j + t) + (e & 1), j = e >> 1, so that one
// 16-byte load per lane fetches two of them (the memory pipeline's cost is per load instruction, tools/l1_fill.hip:
// -1.0 % on the kernel against ten 8-byte loads).
// (Loading all 31 twiddles of pass 1 directly -- no products -- was measured 20 % slower in round 1, and would hold 62
// registers through the transform.  Pass 2's depend on n3 = t & 15 only: all 512 of them sit in LDS, kTabTw2.)
constexpr int kTabP1 = 0;
// N/D = 64 epilogue (one wave, lane t): epi[s-1][t], s = 1..5 = twiddle of inverse-transform stage s (span 2^s) as lane t
// applies it: exp(+j pi (t mod 2^s) / 2^s) in the upper lane of a butterfly pair (t & 2^s), 1 in the lower one
constexpr int kTabEpi = kTabP1 + 10 * kT;
// all of pass 2's twiddles: tw2[n3][k2] = W_N^{32 n3 k2} (n3 < 16, k2 < 32), copied into LDS by every workgroup
constexpr int kTabTw2 = kTabEpi + 5 * 64, kTabSize = kTabTw2 + 512;
constexpr int kTw2Pitch = 34;  // LDS row pitch (elements): lane n3 reads 16 bytes at 272 n3 + 8 k2, distinct 16-byte slots

// Pass 1's twiddles.  The common factor c is the thread's NCO phasor P_t: folded into the four `lo` values here it costs
// 11 products more than the bare twiddles, where multiplying it into the 31 row phasors S^{n1} before the transform
// cost 31.
// v[q] *= c W^{q}, q = 0..31, W^{q} = lo[q & 3] * hi[q >> 2]; tlo = W^1..W^3, thi = W^4, W^8 .. W^28 of the thread's column.
__device__ __forceinline__ void twiddle32(v2f (&v)[32], const v2f (&tlo)[3], const v2f (&thi)[7], v2f c) {
  v2f lo[4];
  lo[0] = c;
#pragma unroll
  for (int l = 1; l < 4; l++) lo[l] = pk_cmul(c, tlo[l - 1]);
#pragma unroll
  for (int l = 0; l < 4; l++) v[l] = pk_cmul(v[l], lo[l]);
#pragma unroll
  for (int h = 1; h < 8; h++) {
#pragma unroll
    for (int l = 0; l < 4; l++) v[4 * h + l] = pk_cmul(v[4 * h + l], pk_cmul(thi[h - 1], lo[l]));
  }
}
// its ten table entries: requested right behind the window loads -- asked for where they are used, after the transform,
// each batch is a full trip to L2 with nothing to overlap it
__device__ __forceinline__ void twiddle32_fetch(v2f (&tlo)[3], v2f (&thi)[7], rsrc_t tab, int t) {
  v2f e[10];
#pragma unroll
  for (int j = 0; j < 5; j++) buf_ld4(e[2 * j], e[2 * j + 1], tab, (unsigned)t * 16u, (unsigned)(kTabP1 + 2 * kT * j) * 8u);
#pragma unroll
  for (int l = 0; l < 3; l++) tlo[l] = e[l];
#pragma unroll
  for (int h = 0; h < 7; h++) thi[h] = e[3 + h];
}

__device__ __forceinline__ v2f phasor2(double turns) {
  float2 const p = phasor_turns(turns);
  return (v2f){p.x, p.y};
}

// N = 65536: largest sweep (cycles per sample^2) the table path takes.  The lane part of the cross term,
// theta = 2 pi rate R lane with R <= 65024 and lane <= 63, is applied as 1 + i theta: its error theta^2 / 2 stays below 5e-8.
constexpr double kSweepLimit64k = 1.1e-11;
// N = 16384, the steady-state variant for swept channels (PLAIN == 2): theta = 2 pi rate (512 n1) lane, n1 <= 31, lane <= 63,
// applied as 1 + i theta; up to 3e-4 rad (error 4.5e-8) -- 4.8 kHz/s at 10 MS/s, 190 Hz/s at 2 MS/s
constexpr double kSweepLimit16k = 4.8e-11;

// N = 65536: hands one 32-bit value to the three sibling workgroups of the channel-block and collects all four into
// out[0..3] (LDS), in sub-transform order.  `slots` = this channel-block's four words of one exchange round; a word is
// (launch tag << 32 | value), written and read with agent-scope atomics (the siblings sit on other XCDs: other L2s).
// The siblings are adjacent workgroup ids of one launch and are dispatched together; the spin is bounded all the same,
// so that a lost sibling shows as an error flag and a NaN, never as a hung device.  Two barriers; every thread calls.
__device__ __forceinline__ void sibling_exchange(unsigned long long *slots, int r4, unsigned value, unsigned tag, float *out,
                                                 int *err) {
  int const t = threadIdx.x;
  if (t == 0)
    __hip_atomic_store(slots + r4, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t < 4) {
    unsigned long long w = 0;
    int it = 0;
    for (;;) {
      w = __hip_atomic_load(slots + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(w >> 32) == tag || ++it > (1 << 18)) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if ((unsigned)(w >> 32) != tag) {
      *err = 1;
      w = 0x7fc00000u;  // NaN
    }
    out[t] = __uint_as_float((unsigned)w);
  }
  __syncthreads();
}

}  // namespace

// grid (channel, block); dynamic LDS = kXchElems float2 (the epilogue's 2 * N_dec float2 fit in it).
// N0: compute_n0 on every block (needs ch.n0lane / ch.n0meta); DUMP: copy one channel's spectra out (tests); PLAIN: the
// host vouches that no channel of the launch sweeps or was retuned since the last call -- the steady state of a
// receiver -- so the per-sample oscillator path (2000 instructions of double arithmetic) is left out and the window
// loads can be issued before anything else (with that path in the kernel they cost it 24 spilled registers); PLAIN == 2
// (N = 16384): the same promise about retunes, but EVERY channel of the launch sweeps, inside kSweepLimit16k -- the sweep
// rides in the row phasors and in a first-order lane term (see the mix).
// BIG: 0 = N 16384; 1 = N 65536 (four sibling workgroups per channel-block, blockIdx.x = 4 * channel + r); 2 = the same,
// PLAIN launch with swept channels
// EPI: which slave epilogue the instance carries -- 1: N/D = 64 only (one wave, registers); 2: N/D = 128 .. 512 only (N/D / 64
// waves); 0: all of them, chosen at run time (other N/D, the DUMP instances, N = 65536 which has none).  The headline
// instance is EPI = 1: with the other epilogues compiled in it took 4 registers more and ran 0.4 % slower.
template <bool N0, bool DUMP, int PLAIN, bool PAIRED, int BIG, int EPI>
__global__ __launch_bounds__(kT, 4) void k_filter_full16k(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ window,
                                                       const float2 *__restrict__ tw, const float2 *__restrict__ tab,
                                                       float2 *__restrict__ spec_dump, int spec_ch,
                                                       const int *__restrict__ chan_list, Big64 big) {
  extern __shared__ __attribute__((aligned(16))) float2 xch[];
  // per wave: S^{n1}, see the mix (BIG: the 128 row phasors T[j][n1])
  __shared__ __attribute__((aligned(16))) float2 stab[(kT / 64) * (BIG ? 128 : 32)];
  __shared__ float sib_f[BIG ? 12 : 1];  // BIG: what the siblings published, see sibling_exchange
  constexpr int kNfull = BIG ? 4 * kN : kN;
  int const r4 = BIG ? (int)(blockIdx.x & 3) : 0;  // residue class of this workgroup's bins
  __shared__ __attribute__((aligned(16))) float2 tw2[16 * kTw2Pitch];   // pass 2's twiddles W_N^{32 n3 k2} at [n3][k2]
  __shared__ float red_f[2][kT / 64];  // compute_n0: one slot per wave and pass
  __shared__ float red_c[kT / 64];     //   bins counted by the fast second pass (exact in float: at most 16384)
  __shared__ int red_i[2][kT / 64];
  int const t = threadIdx.x;
  int const Ndec = g.Ndec;
  // ---------------- load: samples n = 512 n1 + t into v[bitrev5(n1)].  The 32 window loads need nothing but the kernel
  // arguments, so they go out before the channel's parameters are even asked for: a workgroup's first microsecond is
  // otherwise two memory latencies in a row (parameters, then samples) with nothing to compute.
  KQ_STAMP(0);
  // A young workgroup's first instructions -- addresses, the loads, the oscillator table -- are few, and everything it
  // does later waits for what they start; the CU's other workgroup is older and would be served first.  So the waves
  // run at raised priority until their samples are mixed (measured: -1.8 %; dropped earlier or later, the
  // instruction's side effect on the compiler's schedule costs more than the priority gains).
  __builtin_amdgcn_s_setprio(3);
  v2f v[32];
  rsrc_t const xr = make_rsrc(window + (size_t)blockIdx.y * g.L, kNfull * (unsigned)sizeof(float2));
  unsigned const toff = (unsigned)t * (unsigned)sizeof(float2);
  auto load_rows = [&](int first, int last) {
    if constexpr (PAIRED) {
      // `window` is the row-paired copy (written by k_block_energy_sum): rows 2r and 2r + 1 of the window interleaved sample
      // by sample, so that one 16-byte load fetches the thread's samples of both.  The memory pipeline's cost is per
      // instruction (tools/l1_fill.hip): the 128 KiB of a window arrive in half the time (-3.4 % on the kernel).
#pragma unroll
      for (int n1 = first; n1 < last; n1 += 2)
        buf_ld4(v[rfft::bitrev5(n1)], v[rfft::bitrev5(n1 + 1)], xr, 2 * toff, (unsigned)(n1 / 2) * (unsigned)(1024 * sizeof(float2)));
    } else {
#pragma unroll
      for (int n1 = first; n1 < last; n1++) v[rfft::bitrev5(n1)] = buf_ld2(xr, toff, (unsigned)n1 * (unsigned)(512 * sizeof(float2)));
    }
  };
  auto load_window = [&]() { load_rows(0, 32); };
  rsrc_t const tabr = make_rsrc(tab, kTabSize * (unsigned)sizeof(float2));
  v2f tlo[3], thi[7];
  int const cidx = BIG ? (int)(blockIdx.x >> 2) : (int)blockIdx.x;
  int const c = chan_list ? chan_list[cidx] : cidx, b = blockIdx.y;
  v2f pt = (v2f){1.f, 0.f};
  if constexpr (BIG != 0) {
    // ---------------- N = 65536: z_r[m] = sum_j T[j][n1] x[16384 j + 512 n1 + t] (1 + i theta), m = 512 n1 + t
    double const ph0 = ch.lo_phase[c], f0 = ch.lo_freq[c], rs = ch.lo_rate[c];
    double const hp0 = ch.hist_phase[c], hf0 = ch.hist_freq[c], hr = ch.hist_rate[c];
    double const mbase = (double)b * g.L;
    // samples of an old oscillator in this window: those of the call's first hist_len[c] that lie beyond its start (ChanDev)
    int const n_old = (hp0 != ph0 || hf0 != f0 || hr != rs) ? ch.hist_len[c] - b * g.L : 0;
    bool const retuned = n_old > 0;
    if (PLAIN || (!retuned && fabs(rs) <= kSweepLimit64k)) {
      constexpr bool kSwept = !PLAIN || BIG == 2;
      // staging for two steps of loads (a step = the row pair 2 i, 2 i + 1 of all four quarters): sa = even row, sb = odd
      v2f sa[2][4], sb[2][4];
      auto issue = [&](int i, int slot) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if constexpr (PAIRED) {
            buf_ld4(sa[slot][j], sb[slot][j], xr, 2 * toff, (unsigned)(16 * j + i) * (unsigned)(1024 * sizeof(float2)));
          } else {
            sa[slot][j] = buf_ld2(xr, toff, (unsigned)(32 * j + 2 * i) * (unsigned)(512 * sizeof(float2)));
            sb[slot][j] = buf_ld2(xr, toff, (unsigned)(32 * j + 2 * i + 1) * (unsigned)(512 * sizeof(float2)));
          }
        }
      };
      issue(0, 0);
      issue(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      // Row phasors, per wave (lane e and e + 64 evaluate entries e, e + 64; entry 32 j + n1 belongs to the row that starts
      // at R = 16384 j + 512 n1).  With u = mbase + R + t the sample's index in the call:
      //   phase(u) = [ph0 + (f0 + rs mbase) t + rs t (t - 1) / 2]  +  [f0 U + rs U (U - 1) / 2]  +  rs R t,   U = mbase + R
      // first bracket = P_t, second = the row's entry; of the cross term rs R t, t = 64 w + lane, the wave's share
      // rs R 64 w goes into the entry as well.  DIF: W_N^{r m} (-i)^{r j} = exp(-2 pi i (r (512 n1 + t) / N + r j / 4)).
      float2 *const sw = stab + (t >> 6) * 128;
      {
        int const lane = t & 63;
        double const wv = (double)(64 * (t >> 6));
#pragma unroll
        for (int e2 = 0; e2 < 2; e2++) {
          int const e = lane + 64 * e2, j = e >> 5, n1 = e & 31;
          double const R = (double)(16384 * j + 512 * n1), U = mbase + R;
          double turns = f0 * U - (double)(r4 * n1) * (1.0 / 128.0) - 0.25 * (double)(r4 * j);
          if (kSwept) turns += rs * (0.5 * U * (U - 1.0) + R * wv);
          sw[e] = phasor_turns(turns);
        }
      }
      {
        double const td = (double)t;
        double turns = ph0 + f0 * td - (double)(r4 * t) * (1.0 / 65536.0);
        if (kSwept) turns += rs * (mbase * td + 0.5 * td * (td - 1.0));
        pt = phasor2(turns);
      }
      // first-order lane part of the cross term: theta = 2 pi rs R lane = kb (16384 j + 512 n1) / 16384
      float const kbf = kSwept ? (float)(2.0 * M_PI * 16384.0 * rs * (double)(t & 63)) : 0.f;
      v2f const kb = (v2f){kbf, kbf};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float4 *const tp = reinterpret_cast<const float4 *>(sw);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int const slot = i & 1;
        float4 tt[4];
#pragma unroll
        for (int j = 0; j < 4; j++) tt[j] = tp[16 * j + i];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          int const n1 = 2 * i + e;
          v2f x[4], w[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            x[j] = e ? sb[slot][j] : sa[slot][j];
            w[j] = e ? (v2f){tt[j].z, tt[j].w} : (v2f){tt[j].x, tt[j].y};
          }
          v2f z;
          if constexpr (kSwept) {
            // sum_j y_j (1 + i kappa R_j), R_j = 16384 j + 512 n1:  S + i kb (W + n1 / 32 S), W = y1 + 2 y2 + 3 y3
            v2f const y0 = pk_cmul(x[0], w[0]), y1 = pk_cmul(x[1], w[1]), y2 = pk_cmul(x[2], w[2]), y3 = pk_cmul(x[3], w[3]);
            v2f const a = y1 + y3, bb = y2 + y3;
            v2f const S = (y0 + y2) + a;
            v2f const W = rfft::pk_fma(bb, (v2f){2.f, 2.f}, a);
            v2f const Uu = n1 ? rfft::pk_fma(S, (v2f){n1 / 32.f, n1 / 32.f}, W) : W;
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(z) : "v"(Uu), "v"(kb), "v"(S));
          } else {
            z = pk_cmul(x[0], w[0]);
            z = rfft::pk_cmadd(x[1], w[1], z);
            z = rfft::pk_cmadd(x[2], w[2], z);
            z = rfft::pk_cmadd(x[3], w[3], z);
          }
          v[rfft::bitrev5(n1)] = z;
        }
        if (i + 2 < 16) issue(i + 2, slot);
      }
    } else {
      // first block after a retune (history still on the old oscillator), or a sweep beyond the table path's reach:
      // closed-form phase per sample
      // (and of the oscillators before that one, where the channel was retuned again inside M - 1 samples: ChanDev::hist2_*)
      OlderOsc older;
      load_older(ch, c, b * g.L, n_old > 0, older);
#pragma unroll
      for (int n1 = 0; n1 < 32; n1++) {
        v2f acc = (v2f){0.f, 0.f};
        for (int j = 0; j < 4; j++) {
          int const i = 16384 * j + 512 * n1 + t;
          double const m = mbase + i;
          bool const old = i < n_old;
          double pp = old ? hp0 : ph0, ff = old ? hf0 : f0, rr = old ? hr : rs;
          pick_older(older, i, pp, ff, rr);
          double turns = pp + ff * m;
          if (rr != 0.0) turns += rr * (0.5 * m * (m - 1.0));
          turns -= (double)(r4 * (512 * n1 + t)) * (1.0 / 65536.0) + 0.25 * (double)(r4 * j);
          acc = rfft::pk_cmadd(buf_ld2(xr, toff, (unsigned)(32 * j + n1) * (unsigned)(512 * sizeof(float2))), phasor2(turns), acc);
        }
        v[rfft::bitrev5(n1)] = acc;
      }
    }
  }
  if constexpr (PLAIN && BIG == 0) {
    // The channel's two parameters are asked for before the samples, and the oscillator's table and phasor -- two
    // double-precision evaluations, the only arithmetic a wave has before its samples arrive -- sit between the batches
    // of loads: the memory pipeline takes the workgroup's 350 load instructions at about one per 14 cycles and a wave
    // cannot run ahead of a load it cannot issue yet, so they cost nothing there (-1 % against parameters, table and
    // phasor behind the last load).  The phasor of sample 512 n1 + t is P_t S^{n1}, S = exp(j 2 pi 512 f0): lane n1 of
    // each wave evaluates S^{n1} from the double-precision phase into the wave's own LDS slot.
    // PLAIN == 2: every channel of the launch sweeps (rate inside kSweepLimit16k), none was retuned since the last call.
    // With u = A + R, A = b L + t and R = 512 n1, the phase ph0 + f0 u + rs u (u - 1) / 2 is
    //   [ph0 + f0 A + rs A (A - 1) / 2]  +  [f0 R + rs R (R - 1) / 2 + rs R (b L + 64 w)]  +  rs R lane
    // = P_t (rides on pass 1's twiddles as ever) + the wave's table entry n1 + a lane part that is applied to first order
    // in the mix below (as the N = 65536 path does).
    double const ph0 = ch.lo_phase[c], f0 = ch.lo_freq[c];
    double const rs = PLAIN == 2 ? ch.lo_rate[c] : 0.0;
    load_rows(0, 12);
    __builtin_amdgcn_sched_barrier(0);
    float2 *const sw = stab + (t >> 6) * 32;
    if ((t & 63) < 32) {
      double const R = (double)(512 * (t & 63));
      double turns = f0 * R;
      if constexpr (PLAIN == 2) turns += rs * (0.5 * R * (R - 1.0) + R * ((double)b * g.L + (double)(t & ~63)));
      sw[t & 63] = phasor_turns(turns);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_rows(12, 22);
    __builtin_amdgcn_sched_barrier(0);
    {
      double const A = (double)b * g.L + t;
      double turns = ph0 + f0 * A;
      if constexpr (PLAIN == 2) turns += rs * (0.5 * A * (A - 1.0));
      pt = phasor2(turns);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_rows(22, 32);
  }
  twiddle32_fetch(tlo, thi, tabr, t);
  // pass 2's twiddle table goes into LDS: requested here, stored on the way into transpose 1
  v2f const tw2_mine = buf_ld2(tabr, toff, (unsigned)kTabTw2 * 8u);
  KQ_STAMP(10);  // loads issued

  // ---------------- NCO mix (radio.c:132-139)
  if constexpr (BIG == 0) {
    double const ph0 = ch.lo_phase[c], f0 = ch.lo_freq[c], r = ch.lo_rate[c];
    double const hp0 = ch.hist_phase[c], hf0 = ch.hist_freq[c], hr = ch.hist_rate[c];
    double const mbase = (double)b * g.L;
    int const n_old = (hp0 != ph0 || hf0 != f0 || hr != r) ? ch.hist_len[c] - b * g.L : 0;  // (as above)
    bool const retuned = n_old > 0;
    if (PLAIN || (r == 0.0 && !retuned)) {
      float2 *const sw = stab + (t >> 6) * 32;
      if constexpr (!PLAIN) {
        load_window();
        if ((t & 63) < 32) sw[t & 63] = phasor_turns(f0 * (double)(512 * (t & 63)));
        pt = phasor2(ph0 + f0 * (mbase + t));
      }
      // One phasor evaluation and one product per sample, each power exact to float rounding; every lane reads entry n1
      // of its wave's table (a broadcast read).  P_t, the same for the thread's 32 samples, commutes with the transform
      // over n1: it rides on pass 1's twiddles below.
      KQ_STAMP(11);  // oscillator table and P_t evaluated
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if constexpr (PLAIN == 2) {
        // y (1 + i theta), theta = kb n1: one packed add for theta, one packed fma with the operand swizzle of the N = 65536 path
        float const kbf = (float)(2.0 * M_PI * 512.0 * r * (double)(t & 63));
        v2f const kb = (v2f){kbf, kbf};
        v2f th = (v2f){0.f, 0.f};
#pragma unroll
        for (int n1 = 1; n1 < 32; n1++) {
          v2f const y = pk_cmul(v[rfft::bitrev5(n1)], ld2(sw + n1));
          th = th + kb;
          v2f z;
          asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(z) : "v"(y), "v"(th), "v"(y));
          v[rfft::bitrev5(n1)] = z;
        }
      } else {
#pragma unroll
        for (int n1 = 1; n1 < 32; n1++) v[rfft::bitrev5(n1)] = pk_cmul(v[rfft::bitrev5(n1)], ld2(sw + n1));
      }
    } else if constexpr (!PLAIN) {
      // swept channels, and the first block after a retune (history still on the old oscillator): closed-form phase
      // per sample
      // (and of the oscillators before that one, where the channel was retuned again inside M - 1 samples: ChanDev::hist2_*)
      OlderOsc older;
      load_older(ch, c, b * g.L, n_old > 0, older);
#pragma unroll
      for (int n1 = 0; n1 < 32; n1++) {
        int const i = 512 * n1 + t;
        double const m = mbase + i;
        bool const old = i < n_old;  // mixed before the retune took effect: pre-retune oscillator(s)
        double pp = old ? hp0 : ph0, ff = old ? hf0 : f0, rr = old ? hr : r;
        pick_older(older, i, pp, ff, rr);
        double turns = pp + ff * m;
        if (rr != 0.0) turns += rr * (0.5 * m * (m - 1.0));
        v[rfft::bitrev5(n1)] = pk_cmul(buf_ld2(xr, toff, (unsigned)n1 * (unsigned)(512 * sizeof(float2))), phasor2(turns));
      }
    }
  }

  KQ_STAMP(1);  // mixed: every window load has landed
  __builtin_amdgcn_s_setprio(0);
  // ---------------- pass 1: 32-point transforms over n1, twiddle W_N^{t k1}
  rfft::fft_dit_pk<32>(v);
  twiddle32(v, tlo, thi, pt);

  KQ_STAMP(2);
  // ---------------- transpose 1: [k1][t] -> thread (k1 = t >> 4, n3 = t & 15) gathers n2 = 0..31.
  // Half round A carries k1 < 16 (read by threads t < 256), half round B the rest.
  v2f u[32];
  tw2[(t >> 5) * kTw2Pitch + (t & 31)] = make_float2(tw2_mine.x, tw2_mine.y);  // read behind transpose 1's barriers
#if defined(KQ_ABLATE) && (KQ_ABLATE & 1)
  // ABLATION (tools/ablate.sh; results are garbage by design): transpose 1 without its LDS traffic and barriers -- what the
  // kernel would cost if the exchange were free
#pragma unroll
  for (int n2 = 0; n2 < 32; n2++) u[n2] = v[n2];
#else
  {
    int const rd = ((t >> 4) & 15) * kRow1 + (t & 15);
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) xch[k1 * kRow1 + t] = make_float2(v[16 * half + k1].x, v[16 * half + k1].y);
#if !(defined(KQ_ABLATE) && (KQ_ABLATE & 4))   // bit 2: the LDS traffic without the barriers (racy, timing only)
      __syncthreads();
#endif
      if ((t >> 8) == half) {
#pragma unroll
        for (int n2 = 0; n2 < 32; n2++) u[rfft::bitrev5(n2)] = ld2(xch + rd + 16 * n2);
      }
#if !(defined(KQ_ABLATE) && (KQ_ABLATE & 4))
      __syncthreads();
#endif
    }
  }
#endif

  KQ_STAMP(3);
  // ---------------- pass 2: 32-point transforms over n2, twiddle W_512^{n3 k2} = W_N^{32 n3 k2}
  rfft::fft_dit_pk<32>(u);
  {
    // all 31 from the workgroup's LDS copy of the table (two per 16-byte read): no products of table entries
    const float4 *row = reinterpret_cast<const float4 *>(tw2 + (t & 15) * kTw2Pitch);
#pragma unroll
    for (int k2 = 0; k2 < 32; k2 += 2) {
      float4 const w = row[k2 / 2];
      if (k2) u[k2] = pk_cmul(u[k2], (v2f){w.x, w.y});
      u[k2 + 1] = pk_cmul(u[k2 + 1], (v2f){w.z, w.w});
    }
  }

  KQ_STAMP(4);
  // ---------------- transpose 2: [n3][k1][k2] -> thread (k1 = t >> 5 (+16), k2 = t & 31) gathers n3 = 0..15.
  // (A wave holds four k1 with all their n3 before this exchange; laid out so that it also does afterwards, the
  // exchange needs no workgroup barrier at all -- measured 1.6 % slower than this version: the waves drift apart and
  // the workgroup, which holds its LDS until its last wave is done, lives longer.)
  // Half round A is written by the threads holding k1 < 16 (t < 256) and yields ya, half round B yields yb.
  v2f ya[16], yb[16];
#if defined(KQ_ABLATE) && (KQ_ABLATE & 2)
#pragma unroll
  for (int n3 = 0; n3 < 16; n3++) {  // ABLATION: transpose 2 without its LDS traffic and barriers
    ya[n3] = u[n3];
    yb[n3] = u[16 + n3];
  }
#else
  {
    int const wr = (t & 15) * kRow2 + ((t >> 4) & 15) * kCol2;
    int const rd = (t >> 5) * kCol2 + (t & 31);
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if ((t >> 8) == half) {
#pragma unroll
        for (int k2 = 0; k2 < 32; k2++) xch[wr + k2] = make_float2(u[k2].x, u[k2].y);
      }
#if !(defined(KQ_ABLATE) && (KQ_ABLATE & 4))
      __syncthreads();
#endif
#pragma unroll
      for (int n3 = 0; n3 < 16; n3++) (half ? yb : ya)[rfft::bitrev4(n3)] = ld2(xch + n3 * kRow2 + rd);
#if !(defined(KQ_ABLATE) && (KQ_ABLATE & 4))
      __syncthreads();
#endif
    }
  }
#endif

  KQ_STAMP(5);
  // ---------------- pass 3: 16-point transforms over n3.  ya[k3] = X[ka + 1024 k3], yb[k3] = X[kb + 1024 k3]
  rfft::fft_dit_pk<16>(ya);
  rfft::fft_dit_pk<16>(yb);
  int const ka = full16k_bin(t), kb = ka + kFull16kHalf;  // k1 + 32 k2 with k1 = t >> 5 and 16 + (t >> 5), k2 = t & 31

  if (DUMP && c == spec_ch) {
    float2 *o = spec_dump + (size_t)b * kNfull;
    constexpr int kStep = BIG ? 4 : 1;  // BIG: bin 4 q + r4
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      o[kStep * (ka + 1024 * k3) + r4] = make_float2(ya[k3].x, ya[k3].y);
      o[kStep * (kb + 1024 * k3) + r4] = make_float2(yb[k3].x, yb[k3].y);
    }
  }

  KQ_STAMP(6);
  // ---------------- slave (filter.c:206-250): the N/D bins it reads go to LDS as Xs[p], p = k mod N_dec
  // (the exchange buffer is free since the last barrier of transpose 2; done first so that ya / yb die before compute_n0)
  float2 *Xs = xch;
  float2 *G = Xs + Ndec;
  auto to_f2 = [](v2f a) { return make_float2(a.x, a.y); };
  if constexpr (BIG != 0) {
    // bins k = 4 q + r4 of this sub-transform that the slave reads, to global memory at k mod N_dec (k_epilogue64k)
    float2 *const Xg = big.xs + ((size_t)c * g.max_blocks + b) * Ndec;
    auto put = [&](int q, v2f val) {
      int const k = 4 * q + r4;
      if (k <= Ndec / 2)
        Xg[k] = to_f2(val);
      else if (k > kNfull - Ndec / 2)
        Xg[k - kNfull + Ndec] = to_f2(val);
    };
    if (Ndec <= 4096) {  // sub-transform rows k3 = 0 and k3 = 15 only
      put(ka, ya[0]);
      put(kb, yb[0]);
      put(ka + 15 * 1024, ya[15]);
      put(kb + 15 * 1024, yb[15]);
    } else {
#pragma unroll
      for (int k3 = 0; k3 < 16; k3++) {
        put(ka + 1024 * k3, ya[k3]);
        put(kb + 1024 * k3, yb[k3]);
      }
    }
  } else if (Ndec <= 1024) {
    // the slave's bins all lie in the first and the last 1024: rows k3 = 0 and k3 = 15 only
    if (ka <= Ndec / 2) Xs[ka] = to_f2(ya[0]);
    if (kb <= Ndec / 2) Xs[kb] = to_f2(yb[0]);
    if (ka + 15 * 1024 > kN - Ndec / 2) Xs[ka - 1024 + Ndec] = to_f2(ya[15]);
    if (kb + 15 * 1024 > kN - Ndec / 2) Xs[kb - 1024 + Ndec] = to_f2(yb[15]);
  } else {
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      // bins of this k3 lie in [1024 k3, 1024 k3 + 1023]: skip the rows that cannot hold a bin the slave reads
      if (1024 * k3 > Ndec / 2 && 1024 * k3 + 1023 <= kN - Ndec / 2) continue;
#pragma unroll
      for (int half = 0; half < 2; half++) {
        int const n = (half ? kb : ka) + 1024 * k3;
        float2 const val = to_f2(half ? yb[k3] : ya[k3]);
        if (n <= Ndec / 2)
          Xs[n] = val;
        else if (n > kN - Ndec / 2)
          Xs[n - kN + Ndec] = val;
      }
    }
  }
  // N/D = 64: the wave that will run the inverse transform fetches what it needs now, under compute_n0
  const float2 *H = ch.resp + (size_t)c * Ndec;
  bool const isb = (ch.fflags[c] & FLAG_ISB) != 0;
  v2f epi_h = {0.f, 0.f}, epi_h2 = {0.f, 0.f}, epi_w[5] = {};
  bool const epi64 = BIG == 0 && (EPI == 1 || (EPI == 0 && Ndec == 64));
  if (epi64 && t < 64) {
    int const q = (int)(__brev((unsigned)t) >> 26);
    epi_h = ld2(H + q);
    if (isb) epi_h2 = ld2(H + ((64 - q) & 63));
#pragma unroll
    for (int st = 0; st < 5; st++) epi_w[st] = buf_ld2(tabr, (unsigned)t * 8u, (unsigned)(kTabEpi + st * 64) * 8u);
  }
  // N/D = 128, 256, 512 (cfg 2: 256): R = N_dec / 64 waves share the inverse transform (below); each fetches its
  // response bins, the lane-exchange twiddles and its output twiddle here, under compute_n0
  int const epiR = (BIG == 0 && (EPI == 2 || (EPI == 0 && Ndec >= 128 && Ndec <= 512))) ? Ndec >> 6 : 0;
  v2f epi_post = {1.f, 0.f};
  if (epiR && t < 64 * epiR) {
    int const r = t >> 6, l = t & 63;
    int const q = (int)(__brev((unsigned)l) >> 26);
    int const k = epiR * q + r;  // this lane's bin, and its CROSS_CONJ partner N_dec - k
    epi_h = ld2(H + k);
    if (isb) epi_h2 = ld2(H + ((Ndec - k) & (Ndec - 1)));
#pragma unroll
    for (int st = 0; st < 5; st++) epi_w[st] = buf_ld2(tabr, (unsigned)l * 8u, (unsigned)(kTabEpi + st * 64) * 8u);
    float sn, cs;
    sincospif(2.f * (float)(r * l) / (float)Ndec, &sn, &cs);  // exp(+2 pi i r m / N_dec) for output m = lane
    epi_post = (v2f){cs, sn};
  }

  // ---------------- compute_n0 (radio.c:383-425), status only.  pp[k3] = (|X[ka + 1024 k3]|^2, |X[kb + 1024 k3]|^2)
  bool n0_fast = false;
  int outside = 0;  // this transform's bins outside the passband
  if constexpr (N0) {
    v2f pp[16];
    // (read through the constant address space: nothing writes the masks while a kernel runs, and only then may the
    // compiler fetch them with scalar loads this late in the kernel, behind global stores it cannot tell apart from them)
    typedef const unsigned long long __attribute__((address_space(4))) *lane_masks;
    int const ns = __builtin_amdgcn_readfirstlane(ch.n0slot[c]);  // channels with the same filter edges share one mask set
    lane_masks const lm = (lane_masks)(
        ch.n0lane + ((size_t)(ns * (BIG ? 4 : 1) + r4) * (kT / 64) + __builtin_amdgcn_readfirstlane(t >> 6)) * 32);
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++) {
      // spelled out (given `x*x + y*y` on both halves the compiler pairs the additions into one v_pk_add_f32 behind three
      // register moves).  Unpacked on purpose: a packed instruction occupies the SIMD twice as long as a plain one
      // (tools/valu_rate.hip), so v_mul + v_fma per bin beats v_pk_mul + v_add.
      float p0, p1;
      asm("v_mul_f32 %0, %1, %1" : "=v"(p0) : "v"(ya[k3].x));
      asm("v_mul_f32 %0, %1, %1" : "=v"(p1) : "v"(yb[k3].x));
      asm("v_fma_f32 %0, %1, %1, %0" : "+v"(p0) : "v"(ya[k3].y));
      asm("v_fma_f32 %0, %1, %1, %0" : "+v"(p1) : "v"(yb[k3].y));
      // Passband exclusion (radio.c:405-411): the bin counts as 0 from here on.  Which bins lie inside the passband
      // depends only on the channel's filter edges and is precomputed on the host (kq_bank.cpp upload_n0mask) as one
      // 64-bit LANE mask per (wave, bin slot): scalar loads, and one v_cndmask with the mask as its condition per bin.
      // (With the reference's 32-bit wrap of k * samprate the "passband" is scattered over the whole spectrum at
      // 10 MS/s -- 6 % of all bins -- so there are no rows to skip; a bit mask per thread took and / compare / select
      // per bin and was applied twice: 192 instructions per wave where these are 32.)
      asm("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(p0) : "s"(lm[k3]));
      asm("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(p1) : "s"(lm[16 + k3]));
      pp[k3] = (v2f){p0, p1};
    }
    outside = (int)ch.n0meta[ns * (BIG ? 4 : 1) + r4];
    // First pass: avg_n = inf, so `s < avg_n * 2` keeps every finite bin.  Sum them all (the passband's count as 0); when
    // the total comes out finite no bin was inf or NaN (powers are >= 0, nothing cancels) and the count is the
    // precomputed one.
    v2f s2;
    {
      v2f const a0 = pp[0] + pp[1], a1 = pp[2] + pp[3], a2 = pp[4] + pp[5], a3 = pp[6] + pp[7];
      v2f const a4 = pp[8] + pp[9], a5 = pp[10] + pp[11], a6 = pp[12] + pp[13], a7 = pp[14] + pp[15];
      s2 = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
    float const tot = wave_sum_to63(s2.x + s2.y);
    if ((t & 63) == 63) red_f[0][t >> 6] = tot;
    __syncthreads();
    float total = 0;
#pragma unroll
    for (int k = 0; k < kT / 64; k++) total += red_f[0][k];
    int bins1 = outside;
    unsigned long long *const slots = BIG ? big.sync + ((size_t)c * g.max_blocks + b) * 12 : nullptr;
    if constexpr (BIG != 0) {
      // the mean runs over the bins of all four sub-transforms: same four numbers, same order in every sibling
      sibling_exchange(slots, r4, __float_as_uint(total), big.epoch, sib_f, big.err);
      total = (sib_f[0] + sib_f[1]) + (sib_f[2] + sib_f[3]);
      const unsigned *const mt = ch.n0meta + ns * 4;
      bins1 = (int)(mt[0] + mt[1] + mt[2] + mt[3]);
    }
    float const thr = (total / bins1) * 2;
    // Second pass: sum and count of the bins with s < thr.  For a normal, finite threshold the comparison is done in
    // packed arithmetic: with scale = 2^(40 - exponent(thr)), clamp(thr * scale - s * scale) is exactly 1 for every
    // float s < thr (the difference is at least 2^16 after scaling; the fused multiply-add rounds once, so its sign is
    // that of thr - s), exactly 0 for s >= thr.  A passband bin, held as 0, adds nothing to the sum and 1 to the count: the
    // count is put right by their known number at the end.  Three packed instructions per pair of
    // bins instead of compare / select / add per bin.  Anything else -- a NaN or inf bin, an all-zero or denormal
    // spectrum, an empty bin set -- takes the loop that spells the reference's comparisons out.
    n0_fast = total < INFINITY && thr < INFINITY && thr >= 1e-26f;
    if (n0_fast) {
      int const e = (__float_as_int(thr) >> 23) & 0xff;  // biased exponent, 1..254 here
      float const scale = __int_as_float((127 + 40 + 127 - e) << 23);
      float const cs = thr * scale;  // exact: a power-of-two scaling inside the normal range
      v2f const nsc = (v2f){-scale, -scale}, c2 = (v2f){cs, cs};
      v2f acc_e = (v2f){0.f, 0.f}, acc_o = acc_e, cnt_e = acc_e, cnt_o = acc_e;
#pragma unroll
      for (int k3 = 0; k3 < 16; k3++) {
        v2f keep;
        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(keep) : "v"(nsc), "v"(pp[k3]), "v"(c2));
        if (k3 & 1) {
          acc_o = rfft::pk_fma(keep, pp[k3], acc_o);
          cnt_o += keep;
        } else {
          acc_e = rfft::pk_fma(keep, pp[k3], acc_e);
          cnt_e += keep;
        }
      }
      v2f const acc2 = acc_e + acc_o, cnt2 = cnt_e + cnt_o;
      float const acc = wave_sum_to63(acc2.x + acc2.y), cnt = wave_sum_to63(cnt2.x + cnt2.y);
      if ((t & 63) == 63) {
        red_f[1][t >> 6] = acc;
        red_c[t >> 6] = cnt;
      }
      // the sums meet behind the barrier that also publishes the slave's bins, below
    } else {
      // the passband's bins become inf: fails both passes' `< thr`, as do the NaN / inf bins the reference's comparison drops
      {
        float const inf = INFINITY;
#pragma unroll
        for (int k3 = 0; k3 < 16; k3++) {
          asm("v_cndmask_b32_e64 %0, %2, %0, %1" : "+v"(pp[k3].x) : "s"(lm[k3]), "v"(inf));
          asm("v_cndmask_b32_e64 %0, %2, %0, %1" : "+v"(pp[k3].y) : "s"(lm[16 + k3]), "v"(inf));
        }
      }
      float avg = INFINITY;
      float tf_last = 0;
      int bins_last = 0;
      // Both passes run the same loop body.  (Left to itself the compiler peels the first one and, with thr = inf
      // known, counts bins with `p != inf` -- which a NaN bin would pass -- while still summing with an ordered compare.)
      asm volatile("" : "+v"(avg));
      for (int iter = 0; iter < 2; iter++) {
        float acc = 0;
        int wave_bins = 0;  // counted on the scalar unit from the comparison masks
        float const th = avg * 2;
#pragma unroll
        for (int k3 = 0; k3 < 16; k3++) {
          bool const ta = pp[k3].x < th, tb = pp[k3].y < th;
          acc += ta ? pp[k3].x : 0.f;
          acc += tb ? pp[k3].y : 0.f;
          wave_bins += __popcll(__ballot(ta)) + __popcll(__ballot(tb));
        }
        acc = wave_sum(acc);
        __syncthreads();  // the previous round's slots have been read
        if ((t & 63) == 0) {
          red_f[iter][t >> 6] = acc;
          red_i[iter][t >> 6] = wave_bins;
        }
        __syncthreads();
        float tf = 0;
        int bins = 0;
#pragma unroll
        for (int k = 0; k < kT / 64; k++) {
          tf += red_f[iter][k];
          bins += red_i[iter][k];
        }
        if (BIG != 0 && iter == 0) {  // first pass: sum and count over all four sub-transforms
          sibling_exchange(slots + 4, r4, __float_as_uint(tf), big.epoch, sib_f + 4, big.err);
          sibling_exchange(slots + 8, r4, (unsigned)bins, big.epoch, sib_f + 8, big.err);
          tf = (sib_f[4] + sib_f[5]) + (sib_f[6] + sib_f[7]);
          bins = (int)(__float_as_uint(sib_f[8]) + __float_as_uint(sib_f[9]) + __float_as_uint(sib_f[10]) + __float_as_uint(sib_f[11]));
        }
        avg = tf / bins;
        tf_last = tf;
        bins_last = bins;
      }
      if constexpr (BIG != 0) {
        if (t == 0) big.n0part[((size_t)c * g.max_blocks + b) * 4 + r4] = make_float2(tf_last, (float)bins_last);
      } else {
        if (t == 0) pl.n0raw[(size_t)c * g.max_blocks + b] = (float)(avg / (2.0 * kN * g.samprate));
      }
    }
  }

  KQ_STAMP(7);
  // ---------------- slave, continued: response multiply, CROSS_CONJ, inverse transform
  __syncthreads();
  KQ_STAMP(8);
  // (the last wave's first lane: wave 0 has the inverse transform to run, and this division -- float, then double -- would
  // sit in front of it on the workgroup's critical path)
  if (N0 && n0_fast && t == kT - 64) {
    float tf = 0, bins = 0;
#pragma unroll
    for (int k = 0; k < kT / 64; k++) {
      tf += red_f[1][k];
      bins += red_c[k];
    }
    bins -= (float)(kN - outside);  // the passband's bins, held as 0, were counted as kept
    if constexpr (BIG != 0) {
      big.n0part[((size_t)c * g.max_blocks + b) * 4 + r4] = make_float2(tf, bins);  // summed over the siblings in k_epilogue64k
    } else {
      float const avg = tf / bins;  // new_avg_n /= noisebins, radio.c:421
      pl.n0raw[(size_t)c * g.max_blocks + b] = (float)(avg / (2.0 * kN * g.samprate));
    }
  }
  // Side job of ONE wave of the launch: the IF-power recurrence over the call's blocks (kq_energy.hpp), which the
  // demodulators behind this launch read.  As a launch of its own it stood between every filter pass and its demodulators.
  // Here, behind compute_n0, the wave has nothing left to do in the one-wave and R-wave epilogues; in FRONT of the loads the
  // same lines cost the kernel 4 % (the compiler's schedule of the whole load phase changed: 1.459 -> 1.518 ms, tools/ab_libs.sh).
  if (big.iir.sums != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && t >= kT - 64)
    block_energy_iir_wave(big.iir.sums, big.iir.split, big.iir.update, big.iir.nblocks, big.iir.L, big.iir.state, big.iir.if_power,
                          t & 63);
  if constexpr (BIG != 0) return;  // response multiply and inverse transform: k_epilogue64k
  if (epi64) {
    // cfg 3 / 4: one wave multiplies and runs the 64-point inverse transform in its registers (lane exchanges, no
    // barriers); the other seven are done.  Its response bins and stage twiddles were fetched before compute_n0
    // (epi_*), so nothing here waits for memory: the workgroup's slot on the CU is held by this tail alone.
    if (t >= 64) return;
    int const q = (int)(__brev((unsigned)t) >> 26);  // decimation in time: bit-reversed in, natural out
    v2f z = pk_cmul(epi_h, ld2(Xs + q));
    if (isb && q != 0 && q != 32) {  // filter.c:242-248
      v2f const other = pk_cmul(epi_h2, ld2(Xs + 64 - q));
      v2f const oc = (v2f){other.x, -other.y};
      z = q < 32 ? z + oc : z - oc;
    }
    auto stage = [&](auto mm, v2f w, bool first) {
      constexpr int half = decltype(mm)::value;
      v2f const v = first ? z : pk_cmul(z, w);  // w = 1 in the lower lane of a pair
      v2f const r = (v2f){lane_xor<half>(v.x, t), lane_xor<half>(v.y, t)};
      float const sg = __int_as_float(0x3f800000 | ((t & half) ? 0x80000000 : 0));  // z = up ? r - v : r + v
      z = rfft::pk_fma(v, (v2f){sg, sg}, r);
    };
    stage(std::integral_constant<int, 1>{}, z, true);
    stage(std::integral_constant<int, 2>{}, epi_w[0], false);
    stage(std::integral_constant<int, 4>{}, epi_w[1], false);
    stage(std::integral_constant<int, 8>{}, epi_w[2], false);
    stage(std::integral_constant<int, 16>{}, epi_w[3], false);
    stage(std::integral_constant<int, 32>{}, epi_w[4], false);
    float2 *o = pl.filt + ((size_t)c * g.max_blocks + b) * g.olen;
    if (t >= 64 - g.olen) o[t - (64 - g.olen)] = make_float2(z.x, z.y);  // filter.c:131
    KQ_STAMP(9);
    return;
  }
  if (epiR) {
    // N_dec = 64 R.  With k = R q + r and n = m + 64 j:
    //   y[m + 64 j] = sum_r e^{2 pi i r j / R} [ e^{2 pi i r m / N_dec} sum_q G[R q + r] e^{2 pi i q m / 64} ]
    // wave r runs the 64-point inverse transform of its bins across its lanes exactly as the N/D = 64 epilogue does
    // (registers and lane exchanges, no barrier), twiddles it and leaves Z_r[m] in LDS; one barrier; then lane m of wave 0
    // finishes with the R-point transform over r in its registers.  The shared LDS transform this replaces ran its passes on
    // one wave, behind a barrier each, with 64 threads busy: 4200 cycles of a 30 000-cycle workgroup at cfg 2
    // (tools/timeline.py) where this takes a third of that.
    float2 *Z = G;  // [R][64], behind the bins
    if (t < 64 * epiR) {
      int const r = t >> 6, l = t & 63;
      int const q = (int)(__brev((unsigned)l) >> 26);
      int const k = epiR * q + r;
      v2f z = pk_cmul(epi_h, ld2(Xs + k));
      if (isb && k != 0 && 2 * k != Ndec) {  // filter.c:242-248
        v2f const other = pk_cmul(epi_h2, ld2(Xs + Ndec - k));
        v2f const oc = (v2f){other.x, -other.y};
        z = 2 * k < Ndec ? z + oc : z - oc;
      }
      auto stage = [&](auto mm, v2f w, bool first) {
        constexpr int half = decltype(mm)::value;
        v2f const v = first ? z : pk_cmul(z, w);
        v2f const rr = (v2f){lane_xor<half>(v.x, l), lane_xor<half>(v.y, l)};
        float const sg = __int_as_float(0x3f800000 | ((l & half) ? 0x80000000 : 0));
        z = rfft::pk_fma(v, (v2f){sg, sg}, rr);
      };
      stage(std::integral_constant<int, 1>{}, z, true);
      stage(std::integral_constant<int, 2>{}, epi_w[0], false);
      stage(std::integral_constant<int, 4>{}, epi_w[1], false);
      stage(std::integral_constant<int, 8>{}, epi_w[2], false);
      stage(std::integral_constant<int, 16>{}, epi_w[3], false);
      stage(std::integral_constant<int, 32>{}, epi_w[4], false);
      z = pk_cmul(z, epi_post);
      Z[64 * r + l] = make_float2(z.x, z.y);
    }
    __syncthreads();
    if (t >= 64) return;
    // lane m: the R-point inverse transform over r of Z_r[m] in registers (conjugate in, forward transform, conjugate
    // out), then the samples the slave keeps (filter.c:131): n = m + 64 j >= N_dec - olen
    float2 *o = pl.filt + ((size_t)c * g.max_blocks + b) * g.olen;
    int const first = Ndec - g.olen;
    auto combine = [&](auto rc) {
      constexpr int R = decltype(rc)::value;
      constexpr int bits = R == 2 ? 1 : R == 4 ? 2 : 3;
      float2 v[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        float2 const zz = Z[64 * r + t];
        v[(int)(__brev((unsigned)r) >> (32 - bits))] = make_float2(zz.x, -zz.y);
      }
      rfft::fft_dit<R>(v);
#pragma unroll
      for (int j = 0; j < R; j++) {
        int const i = t + 64 * j - first;
        if (i >= 0 && i < g.olen) o[i] = make_float2(v[j].x, -v[j].y);
      }
    };
    if (epiR == 2)
      combine(std::integral_constant<int, 2>{});
    else if (epiR == 4)
      combine(std::integral_constant<int, 4>{});
    else
      combine(std::integral_constant<int, 8>{});
    KQ_STAMP(9);
    return;
  }
  if constexpr (EPI != 0) return;  // (not reached: the host picks the instance by N/D)
  for (int p = t; p <= Ndec / 2; p += kT) {
    float2 gp = cmul(H[p], Xs[p]);
    if (p > 0 && p < Ndec / 2) {
      int const k = Ndec - p;
      float2 gn = cmul(H[k], Xs[k]);
      if (isb) {
        float2 const pos = gp, neg = gn;
        gp = cadd(pos, cconj(neg));
        gn = csub(neg, cconj(pos));
      }
      G[bitrev((unsigned)k, g.log2Ndec)] = gn;
    }
    G[bitrev((unsigned)p, g.log2Ndec)] = gp;
  }
  lds_fft<+1>(G, g.log2Ndec, tw, g.tw_log2);  // filter.c:250

  float2 *o = pl.filt + ((size_t)c * g.max_blocks + b) * g.olen;
  for (int i = t; i < g.olen; i += kT) o[i] = G[Ndec - g.olen + i];  // filter.c:131
  KQ_STAMP(9);
}

// N = 65536: what is left of a channel-block once its four sibling workgroups are done -- the response multiply,
// CROSS_CONJ and the N/D-point inverse transform of execute_filter_output (filter.c:206-250) on the bins they dropped
// at big.xs, and the division that ends compute_n0 (radio.c:421-424) on their second-pass sums.
// grid (channel, block); dynamic LDS = N_dec float2.
__global__ void k_epilogue64k(Geom g, ChanDev ch, Planes pl, const float2 *__restrict__ tw, Big64 big,
                              const int *__restrict__ chan_list, int compute_n0) {
  extern __shared__ __attribute__((aligned(16))) float2 G[];
  int const t = threadIdx.x, Ndec = g.Ndec;
  int const c = chan_list ? chan_list[blockIdx.x] : (int)blockIdx.x, b = blockIdx.y;
  size_t const cb = (size_t)c * g.max_blocks + b;
  if (compute_n0 && t == 0) {
    const float2 *p = big.n0part + cb * 4;
    float const tf = (p[0].x + p[1].x) + (p[2].x + p[3].x), bins = (p[0].y + p[1].y) + (p[2].y + p[3].y);
    float const avg = tf / bins;  // new_avg_n /= noisebins, radio.c:421
    pl.n0raw[cb] = (float)(avg / (2.0 * 65536.0 * g.samprate));
  }
  const float2 *Xs = big.xs + cb * Ndec;
  const float2 *H = ch.resp + (size_t)c * Ndec;
  bool const isb = (ch.fflags[c] & FLAG_ISB) != 0;
  for (int p = t; p <= Ndec / 2; p += blockDim.x) {
    float2 gp = cmul(H[p], Xs[p]);
    if (p > 0 && p < Ndec / 2) {
      int const k = Ndec - p;
      float2 gn = cmul(H[k], Xs[k]);
      if (isb) {  // filter.c:242-248
        float2 const pos = gp, neg = gn;
        gp = cadd(pos, cconj(neg));
        gn = csub(neg, cconj(pos));
      }
      G[bitrev((unsigned)k, g.log2Ndec)] = gn;
    }
    G[bitrev((unsigned)p, g.log2Ndec)] = gp;
  }
  lds_fft<+1>(G, g.log2Ndec, tw, g.tw_log2);  // filter.c:250
  float2 *o = pl.filt + cb * g.olen;
  for (int i = t; i < g.olen; i += blockDim.x) o[i] = G[Ndec - g.olen + i];  // filter.c:131
}

bool full16k_supported(const Geom &g) {
  // the epilogue keeps Xs[N_dec] and G[N_dec] in the exchange buffer
  return g.N == kN && 2 * g.Ndec <= kXchElems && g.Ndec >= 4;
}

// The twiddle tables depend on nothing but N: one copy per device, built on first use.
static const float2 *twiddle_tables() {
  static std::mutex mu;
  static float2 *tabs[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (tabs[dev]) return tabs[dev];
  std::vector<float2> h(kTabSize);
  auto w = [](long long e) {
    double const ang = -2.0 * M_PI * (double)(e % kN) / kN;
    return make_float2((float)cos(ang), (float)sin(ang));
  };
  for (int t = 0; t < kT; t++)
    for (int e = 0; e < 10; e++)
      h[kTabP1 + 2 * (kT * (e >> 1) + t) + (e & 1)] = e < 3 ? w((long long)(e + 1) * t) : w(4LL * (e - 2) * t);
  for (int n3 = 0; n3 < 16; n3++)
    for (int k2 = 0; k2 < 32; k2++) h[kTabTw2 + 32 * n3 + k2] = w(32LL * n3 * k2);
  for (int st = 1; st <= 5; st++)
    for (int t = 0; t < 64; t++) {
      int const half = 1 << st;
      double const ang = M_PI * (double)(t & (half - 1)) / half;
      h[kTabEpi + (st - 1) * 64 + t] = (t & half) ? make_float2((float)cos(ang), (float)sin(ang)) : make_float2(1.f, 0.f);
    }
  float2 *d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(float2)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return nullptr;
  }
  tabs[dev] = d;
  return d;
}

// the row-paired copy (see PAIRED) needs whole pairs of rows in every block; k_block_energy_sum writes it
bool full16k_paired_supported(const Geom &g) {
  return (full16k_supported(g) || full64k_supported(g)) && g.L % 1024 == 0 && (g.M - 1) % 1024 == 0;
}

bool full64k_supported(const Geom &g) { return g.N == 4 * kN && g.Ndec >= 4 && g.Ndec <= 16384; }
double full64k_sweep_limit() { return kSweepLimit64k; }
double full16k_sweep_limit() { return kSweepLimit16k; }

void launch_filter_full16k(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                           const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                           const int *chan_list, int plain, const float2 *window_paired, const Big64 &big) {
  size_t const lds_bytes = (size_t)kXchElems * sizeof(float2);
  const float2 *tab = twiddle_tables();
  if (!tab) {  // cannot happen short of an allocation failure: fall back to the LDS kernel rather than fail the block
    launch_filter_full(s, g, ch, pl, window, tw, nchan, nblocks, compute_n0, spec_dump, spec_ch, chan_list);
    if (big.iir.sums)  // the side job this launch was to carry
      launch_block_energy_iir(s, big.iir.sums, big.iir.L, big.iir.nblocks, big.iir.update, big.iir.state, big.iir.if_power);
    return;
  }
  bool const n0 = compute_n0 && ch.n0lane && ch.n0meta;  // the bank uploads both whenever it was created with compute_n0
  bool const dump = spec_dump != nullptr;
  bool const paired = plain && window_paired != nullptr;  // only the steady-state variant reads the row-paired copy
  auto go = [&](auto kernel) {
    ensure_dynamic_lds((const void *)kernel, lds_bytes);
    hipLaunchKernelGGL(kernel, dim3(nchan, nblocks), dim3(kT), lds_bytes, s, g, ch, pl, paired ? window_paired : window, tw, tab,
                       spec_dump, spec_ch, chan_list, big);
  };
  int const epi = dump ? 0 : g.Ndec == 64 ? 1 : (g.Ndec >= 128 && g.Ndec <= 512) ? 2 : 0;
  auto pick3 = [&](auto n0c, auto dumpc, auto epic) {
    constexpr bool kN0 = decltype(n0c)::value, kDump = decltype(dumpc)::value;
    constexpr int kEpi = decltype(epic)::value;
    if (plain == 2)  // steady state, every channel of the launch swept
      paired ? go(k_filter_full16k<kN0, kDump, 2, true, 0, kEpi>) : go(k_filter_full16k<kN0, kDump, 2, false, 0, kEpi>);
    else if (paired)
      go(k_filter_full16k<kN0, kDump, 1, true, 0, kEpi>);
    else
      plain ? go(k_filter_full16k<kN0, kDump, 1, false, 0, kEpi>) : go(k_filter_full16k<kN0, kDump, 0, false, 0, kEpi>);
  };
  auto pick = [&](auto n0c, auto dumpc) {
    if constexpr (decltype(dumpc)::value) {
      pick3(n0c, dumpc, std::integral_constant<int, 0>{});
    } else {
      if (epi == 1)
        pick3(n0c, dumpc, std::integral_constant<int, 1>{});
      else if (epi == 2)
        pick3(n0c, dumpc, std::integral_constant<int, 2>{});
      else
        pick3(n0c, dumpc, std::integral_constant<int, 0>{});
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  if (n0)
    dump ? pick(T{}, T{}) : pick(T{}, F{});
  else
    dump ? pick(F{}, T{}) : pick(F{}, F{});
}

// N = 65536: four workgroups per channel-block (blockIdx.x = 4 * channel + sub-transform: siblings are adjacent ids, which
// sibling_exchange relies on), then the epilogue kernel.
void launch_filter_full64k(hipStream_t s, const Geom &g, const ChanDev &ch, const Planes &pl, const float2 *window,
                           const float2 *tw, int nchan, int nblocks, int compute_n0, float2 *spec_dump, int spec_ch,
                           const int *chan_list, bool plain, bool swept, const float2 *window_paired, const Big64 &big) {
  size_t const lds_bytes = (size_t)kXchElems * sizeof(float2);
  const float2 *tab = twiddle_tables();
  if (!tab) {  // allocation failure: the caller's launch check reports it; the side job this launch was to carry still runs
    if (big.iir.sums)
      launch_block_energy_iir(s, big.iir.sums, big.iir.L, big.iir.nblocks, big.iir.update, big.iir.state, big.iir.if_power);
    return;
  }
  bool const n0 = compute_n0 && ch.n0lane && ch.n0meta;
  bool const dump = spec_dump != nullptr;
  bool const paired = plain && window_paired != nullptr;
  auto go = [&](auto kernel) {
    ensure_dynamic_lds((const void *)kernel, lds_bytes);
    hipLaunchKernelGGL(kernel, dim3(4 * nchan, nblocks), dim3(kT), lds_bytes, s, g, ch, pl, paired ? window_paired : window, tw,
                       tab, spec_dump, spec_ch, chan_list, big);
  };
  auto pick = [&](auto n0c, auto dumpc) {
    constexpr bool kN0 = decltype(n0c)::value, kDump = decltype(dumpc)::value;
    if (!plain)
      go(k_filter_full16k<kN0, kDump, 0, false, 1, 0>);
    else if (paired)
      swept ? go(k_filter_full16k<kN0, kDump, 1, true, 2, 0>) : go(k_filter_full16k<kN0, kDump, 1, true, 1, 0>);
    else
      swept ? go(k_filter_full16k<kN0, kDump, 1, false, 2, 0>) : go(k_filter_full16k<kN0, kDump, 1, false, 1, 0>);
  };
  using T = std::true_type;
  using F = std::false_type;
  if (n0)
    dump ? pick(T{}, T{}) : pick(T{}, F{});
  else
    dump ? pick(F{}, T{}) : pick(F{}, F{});
  int const threads = g.Ndec >= 1024 ? 256 : 64;
  size_t const epi_lds = (size_t)g.Ndec * sizeof(float2);
  ensure_dynamic_lds((const void *)k_epilogue64k, epi_lds);
  hipLaunchKernelGGL(k_epilogue64k, dim3(nchan, nblocks), dim3(threads), epi_lds, s, g, ch, pl, tw, big, chan_list, n0 ? 1 : 0);
}

}  // namespace kq
