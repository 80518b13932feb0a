// Persistent multi-CU GRU recurrence for gfx950 (forward and backward).
//
// Reference semantics: torch.nn.GRU (batch_first, gate order r,z,n) as built by
// models/enc2dec/base_modules.py:42 and driven by encoders.py:59-62 / decoders.py:182:
//     r = s(W_ir x + b_ir + W_hr h + b_hr)      z = s(W_iz x + b_iz + W_hz h + b_hz)
//     n = tanh(W_in x + b_in + r * (W_hn h + b_hn))          h' = (1 - z) * n + z * h
//
// Design.  The input projections (W_i* x + b_i*) for all time steps are hoisted into ONE bf16 MFMA
// GEMM (gemm_bf16.hip) and arrive here as `xproj`.  The serial part, h_{t-1} @ W_hh^T, runs as ONE
// persistent launch: workgroup (g, s) owns batch rows [32g, 32g+32) and hidden units [U s, U s + U)
// (U = 16 or 32) and keeps its slice of W_hh (3U rows x H, bf16) resident in LDS for the whole
// sequence.  Wave (mt, ut) owns the 16 rows x 16 units tile (mt, ut) and computes ALL THREE gate
// pre-activations of that tile over the full K, so the r/z/n accumulators of one (row, unit) sit in the
// same lane and the gate math runs straight on the MFMA output registers (no cross-wave reduction).
// Per step the critical path is: poll the group's arrival counter -> barrier -> 16-byte sc1 loads of
// h_{t-1} into MFMA A fragments -> 3 x (H/32) MFMAs against the LDS-resident weights -> gates ->
// bf16 h_t through a 2 KB LDS tile -> ONE 16-byte write-through store per thread -> drain -> barrier ->
// one arrival.  Everything that is not needed by the next step (fp32 outputs, the saved gates for the
// backward pass) is stored AFTER the arrival and drains while the next step waits.
//
// Inter-workgroup hand-off (cdna guide, Guideline 16; MI355X_MICROARCH "Valid forms", table row 1):
// every payload byte is stored sc1 (write-through) with 16-byte stores; every storing wave drains
// vmcnt(0); workgroup barrier; ONE lane publishes the arrival: it stores the step number into the workgroup's own flag word
// (default) or adds to the group's arrival counter (agent-scope atomic; PV_GRU_FLAGS=0).
// Consumer: one wave polls the group's flags (one lane: the counter) with sc1 loads, workgroup barrier, then EVERY load of the
// handed-off bytes is a 16-byte `buffer_load ... sc1` to registers -- no L1 invalidate is needed because
// no plain load ever touches those bytes.  Batch groups are independent, so a barrier spans only the S
// workgroups of one group.  Nothing depends on dispatch order or XCD placement; every spin is bounded
// (wall clock) and reports through `err` instead of hanging.  The grid is sized by the host so that all
// workgroups are co-resident (1 per CU; <= the device's CU count).
#include "common.h"
#include "../../include/pevae_hip.h"
#include <stdlib.h>

namespace {

constexpr int ROWS = 32;                    // batch rows per group = 2 MFMA row tiles
constexpr unsigned long long SPIN_LIMIT = 300000000ull;   // 3 s of s_memrealtime (100 MHz)

typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;                 // cache-policy bits of the raw buffer builtins: sc1

// Sticky, process-level: set beside a launch's own error word whenever a bounded wait gives up, and cleared by no launch.
// A launch zeroes ITS sync buffer (error word included) when it starts, so under HIP-graph replay a step that timed out
// is wiped by the next replay before anybody has looked; this word survives (pv_rnn_sync_error reads it).
__device__ unsigned g_rnn_err;

__device__ __forceinline__ u32 ld_relaxed(const u32* p) {      // global_load_dword ... sc1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One lane waits for *ctr >= want; returns false on timeout / error already flagged.  The counter line lives at the
// memory side (atomics are executed there), so every poll is a ~0.5 us round trip: four polls are kept in flight
// and checked in issue order, which cuts the detection delay to a quarter of a round trip.  The error word and the
// wall clock are looked at only every 64 polls.
__device__ __forceinline__ bool wait_ge(const u32* ctr, u32 want, u32* err) {
  u32 v0 = ld_relaxed(ctr);
  if (v0 >= want) return true;
  __builtin_amdgcn_s_sleep(1);
  u32 v1 = ld_relaxed(ctr);
  __builtin_amdgcn_s_sleep(1);
  u32 v2 = ld_relaxed(ctr);
  __builtin_amdgcn_s_sleep(1);
  u32 v3 = ld_relaxed(ctr);
  const unsigned long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    if (v0 >= want) return true;
    v0 = ld_relaxed(ctr);
    if (v1 >= want) return true;
    v1 = ld_relaxed(ctr);
    if (v2 >= want) return true;
    v2 = ld_relaxed(ctr);
    if (v3 >= want) return true;
    v3 = ld_relaxed(ctr);
    if ((it & 15) == 15) {
      if (ld_relaxed(err) != 0) return false;
      if (wall_clock64() - t0 > SPIN_LIMIT) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g_rnn_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

// Flag form of the hand-off (the default; PV_GRU_FLAGS=0 selects the arrival counters): slice s of a row group stores the
// number of steps it has published into its own word, and the waiting WAVE reads the group's S words with one load per
// lane -- no read-modify-write on one address by 16 workgroups (those serialise at the memory side: the backward kernel's
// step went from 5.6 to 4.7 us, the forward kernel's from 4.2 to 4.0).  All 64 lanes must call it.
constexpr int FLAG_STRIDE = 64;             // words per row group (S <= 64)
constexpr int FLAGS_AT = 256;               // first flag word of the sync buffer
__device__ __forceinline__ bool wait_flags_ge(const u32* flags, int S, u32 want, u32* err, int lane) {
  const unsigned long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    const u32 v = lane < S ? ld_relaxed(flags + lane) : want;
    if (__all(v >= want)) return true;
    if ((it & 15) == 15) {
      if (ld_relaxed(err) != 0) return false;
      if (wall_clock64() - t0 > SPIN_LIMIT) {
        if (lane == 0) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&g_rnn_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Which (row group, unit slice) a workgroup owns.  The S workgroups of a group exchange h_t every step, and a hand-off
// between CUs of one XCD (one L2) is several times faster than one across XCDs: with the XCD map, workgroups whose
// blockIdx.x has the same low 3 bits -- one XCD under the round-robin dispatch of an idle chip -- form the groups
// (group g on XCD g % 8).  Measured at B=256, H=512, L=50: 437 -> 204 us forward.  Placement is speed only: nothing
// in the hand-off protocol depends on it.  The grid is 8 * ceil(G/8) * S; workgroups of groups >= G leave at once.
__device__ __forceinline__ bool group_slice(int xmap, int G, int S, int& g, int& s) {
  if (!xmap) {
    g = blockIdx.x / S;
    s = blockIdx.x % S;
    return true;
  }
  const int j = blockIdx.x >> 3;
  g = (blockIdx.x & 7) + 8 * (j / S);
  s = j % S;
  return g < G;
}

__device__ __forceinline__ int swz_mask_for(int nchunks) {
  // largest power of two <= 16 dividing nchunks, minus 1
  int m = 16;
  while (nchunks % m) m >>= 1;
  return m - 1;
}

// v_rcp_f32 (1 ulp), not the correctly rounded division sequence __frcp_rn expands to (10 instructions per reciprocal:
// a third of the gate phase of a recurrence step)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// Diagnostic build only (-DPV_GRU_STAMPS): wave 0 accumulates wall-clock (100 MHz) time per segment of the
// step into words 64.. of the sync buffer.  Never compiled into the shipped library.
#ifdef PV_GRU_STAMPS
#define STAMP(i)                                                    \
  do {                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    const unsigned long long now__ = wall_clock64();                \
    stamp_acc[i] += (unsigned)(now__ - stamp_t);                    \
    stamp_t = now__;                                                \
  } while (0)
#else
#define STAMP(i)
#endif

struct GruFwdArgs {
  const float* xprojT;    // [3H][ldx]: (W_ih x + b_ih)[gate*H+u] of (time t, batch row b) at column t*Bpad + b
  long ldx;
  const bf16_t* whh;      // [3H][Hk] bf16, columns >= H zero
  const float* bhh;       // [3H]
  bf16_t* hseq;           // [(L+1)][Bpad][Hk]; slot i = state before step i (slot 0 pre-filled)
  float* gsaveT;          // optional [L][4][H][Bpad]: r, z, n, (W_hn h + b_hn)   (batch row fastest)
  float* hstate;          // [B][H]: ends as h_n
  float* hall;            // [(L+1)][Bpad][H] fp32 states in processing order (slot 0 pre-filled with h0)
  u32* counters;          // [NG] zeroed
  u32* err;
  u32* stamps;            // diagnostic build only
  const int* lens;        // optional [B]: row b is frozen (state kept, z saved as 1) at time steps t >= lens[b]
  float* cstate;          // LSTM only: [B][H] cell state, ends as c_n
  float* call;            // LSTM only: [(L+1)][Bpad][H] fp32 cell states (slot 0 pre-filled with c0)
  int L, B, H, Hk, Bpad, NG, G, S, reverse, a_lds, xmap;
  unsigned hseq_bytes;
  int dbg;                // timing ablation bits (env PV_GRU_DBG; 0 in normal use, results are wrong otherwise)
  u32* flags;             // optional [NG][FLAG_STRIDE] zeroed: per-slice arrival flags instead of the counters
};

// Forward kernel.  512 threads = 8 waves with fixed roles:
//   waves 0-3  compute: wave (mt, ut) owns the 16 rows x 16 units tile (mt, ut): A staging, MFMAs, gates, publish;
//   wave  4    poller : its lane 0 spins on the group's arrival counter; it never stores, so its sc1 poll
//                       loads never queue behind stores (vmcnt returns in issue order);
//   waves 5-7  storers: copy r, z, n, hn (for the backward pass) and the fp32 h_t from an LDS staging area to
//                       global memory one step behind, so no compute wave ever has a slow store in front of
//                       a load it waits for.
// NKS_C: number of 32-deep k-steps (Hk / 32) when known at compile time, 0 = run-time.  With a compile-time trip
// count the k loop is straight-line and the LDS fragment reads are software-pipelined one 4-k-step chunk ahead
// of the MFMAs; the run-time variant branches per k-step and is LDS-latency bound (kept for uncommon sizes).
constexpr int NT_F = 512, NSTORE = 192;

// MT: 16-row MFMA tiles per batch group (group = 16*MT rows).  MT = 1 halves a group's work so that a workgroup
// can own TWO groups and process them alternately: the hand-off latency of one hides behind the other's work.
// NGT: gates per unit.  3 = GRU (r, z, n; base_modules.py:42 with rnn_cell 'gru'), 4 = LSTM (i, f, g, o; rnn_cell
// 'lstm'): same tiling, hand-off and staging; the LSTM additionally carries the fp32 cell state c (never
// exchanged: c of a unit is only ever needed by the wave that owns the unit) and saves i, f, g, o in gsaveT.
template <int UT, int NKS_C, int MT, int NGT>
__global__ __launch_bounds__(NT_F, 2) void gru_fwd_kernel(GruFwdArgs a) {
  constexpr int U = UT * 16, RW = MT * 16;
  constexpr bool LSTM = NGT == 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int mt = MT == 2 ? (w & 1) : 0, ut = MT == 2 ? ((w >> 1) & 3) : (w & 3);
  const bool compute = w < 4, active = compute && ut < UT;   // wave owns tile (mt, ut)
  const bool storer = w >= 5;
  const int sid = tid - (NT_F - NSTORE);
  int g, s;
  if (!group_slice(a.xmap, a.G, a.S, g, s)) return;
  const int u0 = s * U;
  const int Hk = NKS_C ? NKS_C * 32 : a.Hk, H = a.H, nch = Hk / 8, mask = swz_mask_for(nch), NKS = Hk / 32;
  char* wl = smem;                                            // [NGT*U][Hk] bf16, swizzled
  bf16_t* tile = (bf16_t*)(smem + (size_t)NGT * U * Hk * 2);  // [RW][U] bf16: h_t of this slice
  // [4][U][SGP] fp32: r, z, n, hn | i, f, g, o (row fastest).  SGP = RW + 4: the 16 lanes of a gate write (one unit each, 16
  // bytes) would all start in the same bank at a row pitch of RW floats
  constexpr int SGP = RW + 4;
  float* sg = (float*)(tile + RW * U);
  float* shv = sg + 4 * U * SGP;                            // [RW][U] fp32: h_t
  float* scv = shv + RW * U;                                // [RW][U] fp32: c_t (LSTM only)
  char* al = (char*)(scv + (LSTM ? RW * U : 0));            // [RW][Hk] bf16, swizzled: h_{t-1} of the group (a_lds)

  for (int i = tid; i < NGT * U * nch; i += NT_F) {           // resident weight slice
    const int n = i / nch, ch = i % nch, gate = n / U, u = n % U;
    const uint4 v = *(const uint4*)(a.whh + (size_t)(gate * H + u0 + u) * Hk + ch * 8);
    *(uint4*)(wl + (size_t)n * Hk * 2 + ((ch ^ (n & mask)) * 16)) = v;
  }
  const int utc = active ? ut : 0;
  const int unit = u0 + utc * 16 + c16;
  float bh[NGT];
#pragma unroll
  for (int gt = 0; gt < NGT; ++gt) bh[gt] = a.bhh[gt * H + unit];
  // LDS fragment offsets: 16-byte chunk c of row n sits at chunk (c ^ (n & mask)); with c = ks*4 + q4 the XOR only
  // touches the low 4 bits, so offset = row base + (ks>>2)*256 + off[ks&3]  (immediates in the unrolled loop)
  const int m16 = c16 & mask;
  int boff[NGT][4], aoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int sw = ((j * 4 + q4) ^ m16) * 16;
#pragma unroll
    for (int gt = 0; gt < NGT; ++gt) boff[gt][j] = (gt * U + utc * 16 + c16) * Hk * 2 + sw;
    aoff[j] = (mt * 16 + c16) * Hk * 2 + sw;
  }
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.hseq, 0, a.hseq_bytes, 0x00020000);
  const bool single = a.NG <= a.G;                            // one row group per workgroup: state stays in registers
  float hp[4] = {0.f, 0.f, 0.f, 0.f}, cp[4] = {0.f, 0.f, 0.f, 0.f};
  if (single && active) {                                     // the state lives in registers: read slot 0 once, here
    const int rb0 = g * RW + mt * 16 + q4 * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hp[j] = a.hall[((size_t)rb0 + j) * H + unit];
      if constexpr (LSTM) cp[j] = a.call[((size_t)rb0 + j) * H + unit];
    }
  }
  __syncthreads();
#ifdef PV_GRU_STAMPS
  unsigned stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_t = wall_clock64();
#endif

  for (int step = 0; step < a.L; ++step) {
    const int t = a.reverse ? a.L - 1 - step : step;
    for (int rg = g; rg < a.NG; rg += a.G) {
      const int row0 = rg * RW, rbase = row0 + mt * 16 + q4 * 4;
      // (a) independent of h_{t-1}: this lane's input projections, 4 batch rows per 16-byte load.  The loads are only
      //     issued here: xg is left undefined for waves that do not use it, because merging it with a default value
      //     made the compiler wait for the loads at the merge (2.2 us per step, serial, on every workgroup).
      f32x4 xg[NGT];
      if (active) {
        const float* xp = a.xprojT + (size_t)unit * a.ldx + (size_t)t * a.Bpad + rbase;
#pragma unroll
        for (int gt = 0; gt < NGT; ++gt) xg[gt] = *(const f32x4*)(xp + (size_t)gt * H * a.ldx);
        if (!single) {                                       // several row groups per workgroup: state in memory
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hp[j] = rbase + j < a.B ? a.hstate[(size_t)(rbase + j) * H + unit] : 0.f;
            if constexpr (LSTM) cp[j] = rbase + j < a.B ? a.cstate[(size_t)(rbase + j) * H + unit] : 0.f;
          }
          // wait here, inside the branch: otherwise the compiler waits for every load of the step where the two paths
          // merge, on the common single-group path too
          __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
        }
      }
      STAMP(0);
      // (b) wait for the group's h_{step-1}
      if (step > 0 && w == 4 && !(a.dbg & 1)) {             // wave-uniform branch: only the poller wave enters
        if (a.flags) wait_flags_ge(a.flags + rg * FLAG_STRIDE, a.S, (u32)step, a.err, lane);
        else if (lane == 0) wait_ge(a.counters + rg, (u32)(a.S * step), a.err);
      }
      STAMP(1);
      __syncthreads();
      STAMP(2);
      // (c) three gate tiles over the full K
      f32x4 acc[NGT];
#pragma unroll
      for (int gt = 0; gt < NGT; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.a_lds) {
        // the group's h_{t-1} (RW x Hk bf16) is fetched ONCE per workgroup with sc1 loads and shared through LDS
        if (compute && !(a.dbg & 2)) {
          const unsigned gbase = (unsigned)(((size_t)step * a.Bpad + row0) * Hk * 2);
          for (int i0 = 0; i0 < RW * nch; i0 += 8 * 256) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int idx = i0 + i * 256 + tid;
              if (idx < RW * nch) v[i] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, gbase + idx * 16, 0, AUX_SC1);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int idx = i0 + i * 256 + tid;
              if (idx < RW * nch) {
                const int row = idx / nch, ch = idx % nch;
                *(u32x4*)(al + (size_t)row * Hk * 2 + ((ch ^ (row & mask)) * 16)) = v[i];
              }
            }
          }
        }
        __syncthreads();
        STAMP(3);
        if (active && !(a.dbg & 4)) {
          if constexpr (NKS_C > 0) {
            // fragments of 2 k-steps per chunk, read one chunk ahead of the MFMAs that consume them
            constexpr int NC = NKS_C / 2;
            static_assert(NKS_C % 4 == 0, "compile-time k-step count must be a multiple of 4");
            bf16x8 fa[2][2], fb[2][NGT][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              fa[0][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(al + aoff[j]));
#pragma unroll
              for (int gt = 0; gt < NGT; ++gt) fb[0][gt][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(wl + boff[gt][j]));
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int cur = c & 1, nxt = cur ^ 1;
              if (c + 1 < NC) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  const int ks = (c + 1) * 2 + j, o = (ks >> 2) * 256;
                  fa[nxt][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(al + aoff[ks & 3] + o));
#pragma unroll
                  for (int gt = 0; gt < NGT; ++gt)
                    fb[nxt][gt][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(wl + boff[gt][ks & 3] + o));
                }
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int gt = 0; gt < NGT; ++gt)
                  acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][j], fb[cur][gt][j], acc[gt], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
            for (int ks = 0; ks < NKS; ++ks) {
              const int o = (ks >> 2) * 256;
              const bf16x8 av = __builtin_bit_cast(bf16x8, *(const uint4*)(al + aoff[ks & 3] + o));
#pragma unroll
              for (int gt = 0; gt < NGT; ++gt) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *(const uint4*)(wl + boff[gt][ks & 3] + o));
                acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b, acc[gt], 0, 0, 0);
              }
            }
          }
        }
      } else if (NKS_C == 0 && active) {
        // LDS too small for the shared copy (H > 512): A fragments straight from the published states
        const unsigned abase = (unsigned)((((size_t)step * a.Bpad + row0 + mt * 16 + c16) * Hk + q4 * 8) * 2);
        for (int ks0 = 0; ks0 < NKS; ks0 += 8) {
          u32x4 av[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk)
            if (ks0 + kk < NKS) av[kk] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, abase + (ks0 + kk) * 64, 0, AUX_SC1);
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            if (ks0 + kk < NKS) {
              const int o = ks0 * 64 + (kk >> 2) * 256;
#pragma unroll
              for (int gt = 0; gt < NGT; ++gt) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *(const uint4*)(wl + boff[gt][kk & 3] + o));
                acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[kk]), b, acc[gt], 0, 0, 0);
              }
            }
          }
        }
      }
      STAMP(4);
      // (d) gate math on the accumulator layout: lane (c16, q4) holds rows rbase..rbase+3 of unit `unit`
      float hv[4], cv[4] = {0.f, 0.f, 0.f, 0.f};
      if (active) {
        // rr, zz, nn, hn: the four saved per-step quantities (GRU r, z, n, W_hn h + b_hn | LSTM i, f, g, o)
        float rr[4], zz[4], nn[4], hn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (LSTM) {
            rr[j] = fast_sigmoid(xg[0][j] + acc[0][j] + bh[0]);       // input gate
            zz[j] = fast_sigmoid(xg[1][j] + acc[1][j] + bh[1]);       // forget gate
            nn[j] = fast_tanh(xg[2][j] + acc[2][j] + bh[2]);          // cell candidate
            hn[j] = fast_sigmoid(xg[NGT - 1][j] + acc[NGT - 1][j] + bh[NGT - 1]);   // output gate
            const bool live = rbase + j < a.B;
            cv[j] = live ? zz[j] * cp[j] + rr[j] * nn[j] : 0.f;
            hv[j] = live ? hn[j] * fast_tanh(cv[j]) : 0.f;
            if (a.lens && live && t >= a.lens[rbase + j]) {
              // packed-sequence semantics (encoders.py:55-65): past its length a row keeps (h, c); saving i = g = o = 0
              // and f = 1 zeroes every gate gradient of the step and passes dL/dc through, the backward kernel carries
              // dL/dh across it
              hv[j] = hp[j];
              cv[j] = cp[j];
              zz[j] = 1.f;
              rr[j] = nn[j] = hn[j] = 0.f;
            }
            cp[j] = cv[j];
            hp[j] = hv[j];
            continue;
          }
          hn[j] = acc[2][j] + bh[2];
          rr[j] = fast_sigmoid(xg[0][j] + acc[0][j] + bh[0]);
          zz[j] = fast_sigmoid(xg[1][j] + acc[1][j] + bh[1]);
          nn[j] = fast_tanh(xg[2][j] + rr[j] * hn[j]);
          hv[j] = (rbase + j < a.B) ? nn[j] + zz[j] * (hp[j] - nn[j]) : 0.f;
          if (a.lens && rbase + j < a.B && t >= a.lens[rbase + j]) {
            // packed-sequence semantics (encoders.py:55-65): past its length a row keeps its state; saving z = 1 and
            // zero gates makes the unchanged backward kernel pass dL/dh straight through this step
            hv[j] = hp[j];
            zz[j] = 1.f;
            rr[j] = nn[j] = hn[j] = 0.f;
          }
          hp[j] = hv[j];
        }
        const int lu = ut * 16 + c16, lr = mt * 16 + q4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tile[(lr + j) * U + lu] = f2bf(hv[j]);
          shv[(lr + j) * U + lu] = hv[j];
          if constexpr (LSTM) scv[(lr + j) * U + lu] = cv[j];
        }
        if (a.gsaveT) {
          *(float4*)(sg + (0 * U + lu) * SGP + lr) = make_float4(rr[0], rr[1], rr[2], rr[3]);
          *(float4*)(sg + (1 * U + lu) * SGP + lr) = make_float4(zz[0], zz[1], zz[2], zz[3]);
          *(float4*)(sg + (2 * U + lu) * SGP + lr) = make_float4(nn[0], nn[1], nn[2], nn[3]);
          *(float4*)(sg + (3 * U + lu) * SGP + lr) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        }
      }
      STAMP(9);
      __syncthreads();
      STAMP(5);
      // (e) publish h_step: one 16-byte write-through store per thread, drain, barrier, one arrival.
      //     (The storer waves copy the staged values out after the arrival; the staging area is not rewritten
      //     before they have passed the next two barriers.)
      constexpr int NG4 = 4 * U * RW / 4, IT_G = (NG4 + NSTORE - 1) / NSTORE;
      constexpr int NH4 = RW * U / 4, IT_H = (NH4 + NSTORE - 1) / NSTORE;
      if (tid < RW * U / 8 && !(a.dbg & 8)) {
        const int row = tid / (U / 8), ch = tid % (U / 8);
        const u32x4 v = *(const u32x4*)(tile + row * U + ch * 8);
        const unsigned off = (unsigned)((((size_t)(step + 1) * a.Bpad + row0 + row) * Hk + u0 + ch * 8) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, hrsrc, off, 0, AUX_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(6);
      __syncthreads();
      if (tid == 0 && !(a.dbg & 16)) {
        if (a.flags) __hip_atomic_store(a.flags + rg * FLAG_STRIDE + s, (u32)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(a.counters + rg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("" ::: "memory");
      STAMP(7);
      // (f) everything the next step does not wait for
      if (storer && !(a.dbg & 32)) {
        if (a.gsaveT) {
#pragma unroll
          for (int k = 0; k < IT_G; ++k) {
            const int i = sid + k * NSTORE;          // float4 index into sg: [(q*U + unit)][row/4]
            if (i < NG4) {
              const int qu = i / (RW / 4), r4 = i % (RW / 4), q = qu / U, lu = qu % U;
              *(float4*)(a.gsaveT + (((size_t)t * 4 + q) * H + u0 + lu) * a.Bpad + row0 + r4 * 4) = *(const float4*)(sg + qu * SGP + r4 * 4);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < IT_H; ++k) {
          const int i = sid + k * NSTORE;            // float4 index into shv: [row][unit/4]
          if (i < NH4) {
            const int row = i / (U / 4), c4 = i % (U / 4);
            const size_t o = ((size_t)(step + 1) * a.Bpad + row0 + row) * H + u0 + c4 * 4;
            *(float4*)(a.hall + o) = *(const float4*)(shv + i * 4);
            if constexpr (LSTM) *(float4*)(a.call + o) = *(const float4*)(scv + i * 4);
          }
        }
      } else if (active && (!single || step == a.L - 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (rbase + j < a.B) {
            a.hstate[(size_t)(rbase + j) * H + unit] = hv[j];
            if constexpr (LSTM) a.cstate[(size_t)(rbase + j) * H + unit] = cv[j];
          }
      }
      STAMP(8);
    }
  }
#ifdef PV_GRU_STAMPS
  if (tid == 0 && blockIdx.x < 16)
    for (int i = 0; i < 10; ++i) a.stamps[blockIdx.x * 10 + i] = stamp_acc[i];
#endif
}

struct GruBwdArgs {
  const float* dout;      // optional dL/d out: dout[b*sb + t*st + u]
  long dout_sb, dout_st;
  const float* dhn;       // optional [B][H]
  const float* gsaveT;    // [L][4][H][Bpad]  (batch row fastest)
  const float* hall;      // [(L+1)][Bpad][H] fp32 states in processing order (slot i = state before step i)
  const bf16_t* whhT;     // [H][K3] bf16: whhT[u][k] = W_hh[k][u], columns >= 3H zero
  bf16_t* dgi;            // [L][Bpad][K3] bf16: (dr, dz, dn)       rows >= B zero
  bf16_t* dgh;            // [L][Bpad][K3] bf16: (dr, dz, dn * r)   exchange buffer (when dgx is null) and GEMM operand
  float* dcarry;          // [B][H] fp32 scratch; ends as dL/dh0 (GRU) | dL/dc0 (LSTM: the directly carried gradient is dc)
  const float* call;      // LSTM only: [(L+1)][Bpad][H] fp32 cell states
  const float* dcn;       // LSTM only, optional [B][H]: dL/dc_n
  float* dh0;             // LSTM only: [B][H] dL/dh0
  float* dbias_part;      // [G][4][H]: sums over rows & time of dr, dz, dn, dn*r
  u32* counters;          // [NG] zeroed
  u32* err;
  int L, B, H, K3, Bpad, NG, G, S, reverse, xmap;
  unsigned dg_bytes;
  u32* stamps;            // diagnostic build only
  const int* lens;        // LSTM only: rows frozen at t >= lens[b] (the GRU encodes a frozen step in its saved gates)
  bf16_t* dgx;            // GRU, optional: the exchanged copy of dgh in MFMA A-fragment order (see FX below)
  u32* flags;             // optional [NG][FLAG_STRIDE] zeroed: per-slice arrival flags instead of the counters
};

// Backward through time.  Per step: rec = dgh[previously handled step] @ W_hh (K = 3H, split over the 4
// waves, partials combined through LDS), then the elementwise gate backward on the MFMA output layout
// (wave (mt, ut) owns tile (mt, ut) as in the forward kernel), then the bf16 gate gradients go through an
// LDS tile so that the exchanged copy (dgh) leaves as 16-byte write-through stores; the same hand-off
// protocol as the forward kernel.
// NGT = 4 (LSTM): one gate-gradient image (di, df, dg, do) serves both the exchange and the weight-gradient
// GEMMs (dgi is not written), K = 4H, and the carried gradient is dL/dc while dL/dh arrives only through
// the recurrent term.
// MT: 16-row MFMA tiles per row group (2: 32-row groups, 1: 16-row groups -- twice the workgroups, each fetching half
// the exchanged bytes per step; the step is bound by that fetch).
// FX: the exchanged gate gradients travel through a.dgx, an image in MFMA A-fragment order -- [t][16-row tile][k-step]
// [q4][row in tile][8 bf16], the 1 KB a wave needs for one MFMA is one contiguous block -- and the row-major dgh (the
// operand of the dW_hh GEMM) is written after the arrival, with the copies nobody waits for.  A fragment load from the
// row-major image touches 16 rows x 64 bytes, one cache-line look-up per lane group: the step's fetch of the group's
// 48 KB took 2.4 us that way (stamps), against 0.7 us at the L2 -> CU rate.
template <int UT, int NKS_C, int NGT, int MT, bool FX>
__global__ __launch_bounds__(256, 1) void gru_bwd_kernel(GruBwdArgs a) {
  constexpr int U = UT * 16, RW = MT * 16, NTILE = MT * UT;
  constexpr bool LSTM = NGT == 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int mt = MT == 2 ? (w & 1) : 0, ut = MT == 2 ? (w >> 1) : w;
  const bool active = ut < UT;
  int g, s;
  if (!group_slice(a.xmap, a.G, a.S, g, s)) return;
  const int u0 = s * U;
  const int K3 = NKS_C ? NKS_C * 32 : a.K3, nch = K3 / 8, mask = swz_mask_for(nch), NKS = K3 / 32, H = a.H;
  char* wl = smem;                                                   // [U][K3] bf16 swizzled (W_hh^T slice)
  f32x4* part = (f32x4*)(smem + (size_t)U * K3 * 2);                 // [4 waves][NTILE][64 lanes]
  bf16_t* tgh = (bf16_t*)(part + 4 * NTILE * 64);                    // [RW][NGT][U] bf16: dr, dz, dn*r | di, df, dg, do
  bf16_t* tgi = tgh + RW * NGT * U;                                  // [RW][3][U] bf16: dr, dz, dn (GRU only)
  float* bred = (float*)(tgi + (LSTM ? 0 : RW * NGT * U));           // [4*MT][4][U] bias-grad reduction

  for (int i = tid; i < U * nch; i += 256) {
    const int n = i / nch, ch = i % nch;
    const uint4 v = *(const uint4*)(a.whhT + (size_t)(u0 + n) * K3 + ch * 8);
    *(uint4*)(wl + (size_t)n * K3 * 2 + ((ch ^ (n & mask)) * 16)) = v;
  }
  const int unit = u0 + (active ? ut : 0) * 16 + c16;
  // wave w takes k-steps ks = w + 4*i: chunk ks*4 + q4 has constant low bits (w*4 + q4), so the swizzled
  // LDS offset is a per-lane base + i*256
  int boff[UT];
#pragma unroll
  for (int j = 0; j < UT; ++j) boff[j] = (j * 16 + c16) * K3 * 2 + ((((w & 3) * 4 + q4) ^ (c16 & mask)) * 16);
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(FX ? a.dgx : a.dgh), 0, a.dg_bytes, 0x00020000);
  const bool single = a.NG <= a.G;
  float carry[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  float gx[4] = {0.f, 0.f, 0.f, 0.f};                    // dL/dh_n (and the LSTM's dL/dc_n in carry): enter at step 0
  float hcarry[4] = {0.f, 0.f, 0.f, 0.f};                // LSTM with lengths: dL/dh carried across a row's frozen steps
  if (single) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bclamp = min(g * RW + mt * 16 + q4 * 4 + j, a.B - 1);
      if (a.dhn) gx[j] = a.dhn[(size_t)bclamp * H + unit];
      if constexpr (LSTM) carry[j] = a.dcn ? a.dcn[(size_t)bclamp * H + unit] : 0.f;
    }
  }
  __syncthreads();
#ifdef PV_GRU_STAMPS
  unsigned stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_t = wall_clock64();
#endif

  // steps run from the last processed time step back to the first; one extra pass (step == L)
  // only folds the recurrent term of the first step into dL/dh0.
  for (int step = 0; step <= a.L; ++step) {
    const int ti = a.L - 1 - step;                       // processing index whose gradients we form
    const int t = a.reverse ? a.L - 1 - ti : ti;         // time index of that step
    for (int rg = g; rg < a.NG; rg += a.G) {
      const int row0 = rg * RW, rbase = row0 + mt * 16 + q4 * 4;
      // (b) wait for the group's gate gradients of the previously handled step and REQUEST this wave's fragments of them
      //     first; the step's own operands (a) are requested behind them: the poll is then not queued behind a dozen HBM
      //     loads (vmcnt retires in order), and those loads' latency passes under the MFMAs instead of in front of the wait
      const int tprev = a.reverse ? a.L - 1 - (ti + 1) : ti + 1;
      const unsigned abase = FX ? (unsigned)(((size_t)tprev * (a.Bpad >> 4) + (row0 >> 4)) * NKS * 1024 + lane * 16)
                                : (unsigned)((((size_t)tprev * a.Bpad + row0 + c16) * K3 + q4 * 8) * 2);
      u32x4 av[12][MT];                                      // wave w takes k-steps w, w+4, ... (12 per pass)
      if (step > 0) {
        if (a.flags) {
          if (w == 0) wait_flags_ge(a.flags + rg * FLAG_STRIDE, a.S, (u32)step, a.err, lane);
        } else if (tid == 0) wait_ge(a.counters + rg, (u32)(a.S * step), a.err);
        STAMP(1);
        __syncthreads();
        STAMP(2);
#pragma unroll
        for (int kk = 0; kk < 12; ++kk)
          if (w + kk * 4 < NKS) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
              av[kk][i] = __builtin_amdgcn_raw_buffer_load_b128(
                  grsrc, FX ? abase + (unsigned)((i * NKS + w + kk * 4) * 1024) : abase + (unsigned)(i * 16 * K3 * 2) + (w + kk * 4) * 64,
                  0, AUX_SC1);
          }
      }
      STAMP(9);
      // (a) independent of the exchange: saved gates / states / upstream gradients of this lane's 4 rows.  Everything is
      //     loaded unconditionally from clamped (always valid) addresses and masked where it is used: loads under
      //     data-dependent branches made the compiler wait for each one at the branch merge -- five serial round trips,
      //     2.6 us per step.  (Rows up to Bpad exist in gsaveT / hall / call; dout has B rows only, hence the clamp.)
      float gr[4], gz[4], gn[4], ghn[4], ghp[4], gdo[4], gtc[4];   // LSTM: ghp = c_{t-1}, gtc = tanh(c_t)
      {
        const int tc = step < a.L ? t : (a.reverse ? a.L - 1 : 0), tic = step < a.L ? ti : 0;
        const float* gs = a.gsaveT + ((size_t)tc * 4 * H + unit) * a.Bpad + rbase;
        const size_t gq = (size_t)H * a.Bpad;
        const float4 v0 = *(const float4*)gs, v1 = *(const float4*)(gs + gq), v2 = *(const float4*)(gs + 2 * gq),
                     v3 = *(const float4*)(gs + 3 * gq);
        gr[0] = v0.x; gr[1] = v0.y; gr[2] = v0.z; gr[3] = v0.w;
        gz[0] = v1.x; gz[1] = v1.y; gz[2] = v1.z; gz[3] = v1.w;
        gn[0] = v2.x; gn[1] = v2.y; gn[2] = v2.z; gn[3] = v2.w;
        ghn[0] = v3.x; ghn[1] = v3.y; ghn[2] = v3.z; ghn[3] = v3.w;
        const float* dsrc = a.dout ? a.dout : a.hall;            // a readable dummy when there is no dL/d out
        const long dsb = a.dout ? a.dout_sb : 0, dst_ = a.dout ? a.dout_st : 0;
        const float dscale = a.dout ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int brow = rbase + j, bclamp = min(brow, a.B - 1);
          if constexpr (LSTM) {
            ghp[j] = a.call[((size_t)tic * a.Bpad + brow) * H + unit];
            gtc[j] = a.call[((size_t)(tic + 1) * a.Bpad + brow) * H + unit];    // tanh applied at use
          } else {
            ghp[j] = a.hall[((size_t)tic * a.Bpad + brow) * H + unit];
            gtc[j] = 0.f;
          }
          gdo[j] = dsrc[bclamp * dsb + tc * dst_ + unit] * dscale;
        }
        if (!single) {                                           // several row groups per workgroup: state in memory
          if (step == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int bclamp = min(rbase + j, a.B - 1);
              gx[j] = a.dhn ? a.dhn[(size_t)bclamp * H + unit] : 0.f;
              if constexpr (LSTM) carry[j] = a.dcn ? a.dcn[(size_t)bclamp * H + unit] : 0.f;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) carry[j] = a.dcarry[(size_t)min(rbase + j, a.B - 1) * H + unit];
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0) inside the branch (see the forward kernel)
        }
      }
      // recurrent term from the previously handled step: dgh[prev] @ W_hh  (K = 3H)
      float rec[4] = {0.f, 0.f, 0.f, 0.f};
      STAMP(0);
      if (step > 0) {
        f32x4 acc[MT][UT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < UT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks0 = w; ks0 < NKS; ks0 += 48) {
          if (ks0 != w) {                                   // later passes (K > 1536): load here
#pragma unroll
            for (int kk = 0; kk < 12; ++kk)
              if (ks0 + kk * 4 < NKS) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                  av[kk][i] = __builtin_amdgcn_raw_buffer_load_b128(
                      grsrc, FX ? abase + (unsigned)((i * NKS + ks0 + kk * 4) * 1024) : abase + (unsigned)(i * 16 * K3 * 2) + (ks0 + kk * 4) * 64,
                      0, AUX_SC1);
              }
          }
#pragma unroll
          for (int kk = 0; kk < 12; ++kk) {
            const int ks = ks0 + kk * 4;
            if (ks < NKS) {
#pragma unroll
              for (int j = 0; j < UT; ++j) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *(const uint4*)(wl + boff[j] + (ks0 >> 2) * 256 + kk * 256));
#pragma unroll
                for (int i = 0; i < MT; ++i)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[kk][i]), b, acc[i][j], 0, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < UT; ++j) part[(w * NTILE + i * UT + j) * 64 + lane] = acc[i][j];
        STAMP(3);
        __syncthreads();
        STAMP(4);
        if (active) {
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) {
            const f32x4 p = part[(ww * NTILE + mt * UT + ut) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) rec[j] += p[j];
          }
        }
      }
      // (c) elementwise backward of step ti
      if (step == a.L) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int brow = rbase + j;
          if (active && brow < a.B) {
            const float prev = single ? carry[j] : a.dcarry[(size_t)brow * H + unit];
            if constexpr (LSTM) {
              a.dcarry[(size_t)brow * H + unit] = prev;
              a.dh0[(size_t)brow * H + unit] = rec[j] + hcarry[j];
            } else {
              a.dcarry[(size_t)brow * H + unit] = prev + rec[j];
            }
          }
        }
        __syncthreads();          // the partial buffer is reused by the next row group
        continue;
      }
      float dr[4], dz[4], dn[4], dnr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = active && rbase + j < a.B;
        if constexpr (LSTM) {
          // saved slots: gr = i, gz = f, gn = g, ghn = o.  (dr, dz, dn, dnr) carry (di, df, dg, do) pre-activation
          // a frozen step (t >= lens[row]; the forward kernel saved i = g = o = 0, f = 1): every gate gradient is zero,
          // dL/dc passes through (carry = dc * f) and dL/dh is handed to the step before through hcarry
          const bool frozen = a.lens != nullptr && live && t >= a.lens[rbase + j];
          const float dh = rec[j] + gdo[j] + (step == 0 ? gx[j] : 0.f) + hcarry[j];
          hcarry[j] = frozen ? dh : 0.f;
          gtc[j] = fast_tanh(gtc[j]);
          const float dc = dh * ghn[j] * (1.f - gtc[j] * gtc[j]) + carry[j];
          dr[j] = dc * gn[j] * gr[j] * (1.f - gr[j]);
          dz[j] = dc * ghp[j] * gz[j] * (1.f - gz[j]);
          dn[j] = dc * gr[j] * (1.f - gn[j] * gn[j]);
          dnr[j] = dh * gtc[j] * ghn[j] * (1.f - ghn[j]);
          carry[j] = dc * gz[j];
        } else {
          const float dh = rec[j] + gdo[j] + (step > 0 ? carry[j] : gx[j]);
          dn[j] = dh * (1.f - gz[j]) * (1.f - gn[j] * gn[j]);
          dz[j] = dh * (ghp[j] - gn[j]) * gz[j] * (1.f - gz[j]);
          dr[j] = dn[j] * ghn[j] * gr[j] * (1.f - gr[j]);
          dnr[j] = dn[j] * gr[j];
          carry[j] = dh * gz[j];
        }
        if (!live) dn[j] = dz[j] = dr[j] = dnr[j] = 0.f;
        bsum[0] += dr[j];
        bsum[1] += dz[j];
        bsum[2] += dn[j];
        bsum[3] += dnr[j];
        if (active) {
          const int o = (mt * 16 + q4 * 4 + j) * NGT * U + ut * 16 + c16;
          const bf16_t br = f2bf(dr[j]), bz = f2bf(dz[j]);
          tgh[o] = br;
          tgh[o + U] = bz;
          if constexpr (LSTM) {
            tgh[o + 2 * U] = f2bf(dn[j]);
            tgh[o + (NGT - 1) * U] = f2bf(dnr[j]);
          } else {
            tgh[o + 2 * U] = f2bf(dnr[j]);
            tgi[o] = br;
            tgi[o + U] = bz;
            tgi[o + 2 * U] = f2bf(dn[j]);
          }
        }
      }
      __syncthreads();
      STAMP(5);
      // (d) publish dgh (write-through), drain, barrier, arrive; then the copies nobody waits for
      constexpr int NCH = RW * NGT * U / 8;                // 16-byte chunks per image
      for (int c = tid; c < NCH; c += 256) {
        if constexpr (FX) {                                 // row fastest: 16 consecutive threads fill 256 contiguous bytes
          const int row = c % RW, rem = c / RW, gt = rem / (U / 8), ch = rem % (U / 8), k = gt * H + u0 + ch * 8;
          const u32x4 v = *(const u32x4*)(tgh + row * NGT * U + gt * U + ch * 8);
          const unsigned off = (unsigned)((((size_t)t * (a.Bpad >> 4) + ((row0 + row) >> 4)) * NKS + (k >> 5)) * 1024 +
                                          ((k >> 3) & 3) * 256 + ((row0 + row) & 15) * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, grsrc, off, 0, AUX_SC1);
        } else {
          const int row = c / (NGT * U / 8), rem = c % (NGT * U / 8), gt = rem / (U / 8), ch = rem % (U / 8);
          const u32x4 v = *(const u32x4*)(tgh + row * NGT * U + gt * U + ch * 8);
          const unsigned off = (unsigned)((((size_t)t * a.Bpad + row0 + row) * K3 + gt * H + u0 + ch * 8) * 2);
          __builtin_amdgcn_raw_buffer_store_b128(v, grsrc, off, 0, AUX_SC1);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(6);
      __syncthreads();
      if (tid == 0) {
        if (a.flags) __hip_atomic_store(a.flags + rg * FLAG_STRIDE + s, (u32)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(a.counters + rg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("" ::: "memory");
      STAMP(7);
      if constexpr (!LSTM) {
        for (int c = tid; c < NCH; c += 256) {
          const int row = c / (3 * U / 8), rem = c % (3 * U / 8), gt = rem / (U / 8), ch = rem % (U / 8);
          const uint4 v = *(const uint4*)(tgi + row * 3 * U + gt * U + ch * 8);
          *(uint4*)(a.dgi + ((size_t)t * a.Bpad + row0 + row) * K3 + gt * H + u0 + ch * 8) = v;
          if constexpr (FX)
            *(uint4*)(a.dgh + ((size_t)t * a.Bpad + row0 + row) * K3 + gt * H + u0 + ch * 8) =
                *(const uint4*)(tgh + row * 3 * U + gt * U + ch * 8);
        }
      }
      if (!single) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (active && rbase + j < a.B) a.dcarry[(size_t)(rbase + j) * H + unit] = carry[j];
      }
      STAMP(8);
    }
  }
#ifdef PV_GRU_STAMPS
  if (tid == 0 && blockIdx.x < 16)
    for (int i = 0; i < 10; ++i) a.stamps[blockIdx.x * 10 + i] = stamp_acc[i];
#endif
  // bias-gradient partials: reduce this workgroup's rows in a fixed order
  {
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bred[((mt * 4 + q4) * 4 + i) * U + ut * 16 + c16] = bsum[i];
    }
    __syncthreads();
    for (int i = tid; i < 4 * U; i += 256) {
      float sacc = 0.f;
      for (int r = 0; r < 4 * MT; ++r) sacc += bred[r * 4 * U + i];
      a.dbias_part[((size_t)g * 4 + i / U) * H + u0 + i % U] = sacc;
    }
  }
}

// hseq slot 0 / hstate <- h0 (or zeros); also used to zero padded rows
__global__ void gru_pack_h0_kernel(const float* __restrict__ h0, int B, int H, int Hk, bf16_t* __restrict__ hseq0,
                                   float* __restrict__ hstate, float* __restrict__ hall0, u32* __restrict__ sync,
                                   int sync_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < sync_words) sync[i] = 0;                       // arrival counters and error word of the launch that follows
  if (i >= B * H) return;
  const int b = i / H, u = i % H;
  const float v = h0 ? h0[i] : 0.f;
  hseq0[(size_t)b * Hk + u] = f2bf(v);
  hstate[i] = v;
  if (hall0) hall0[i] = v;
}

// the backward launch's zero-initialised buffers in one pass (three to four memsets before)
__global__ void rnn_bwd_zero_kernel(u32* __restrict__ sync, int sync_words, float* __restrict__ dh0, float* __restrict__ dc0,
                                    long n_state, float* __restrict__ dbias_part, long n_part) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < sync_words) sync[i] = 0;
  if (i < n_state) {
    dh0[i] = 0.f;
    if (dc0) dc0[i] = 0.f;
  }
  if (i < n_part) dbias_part[i] = 0.f;
}

__global__ void gru_bias_reduce_kernel(const float* __restrict__ part, int G, int H, int lstm, float* __restrict__ db_ih,
                                       float* __restrict__ db_hh) {
  // part [G][4][H] -> GRU: db_ih [3H] = (dr, dz, dn), db_hh [3H] = (dr, dz, dn*r);  LSTM: db_ih = db_hh = all four
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * H) return;
  float sacc = 0.f;
  for (int gidx = 0; gidx < G; ++gidx) sacc += part[(size_t)gidx * 4 * H + i];
  const int gt = i / H, u = i % H;
  if (lstm) {
    db_ih[i] = sacc;
    db_hh[i] = sacc;
    return;
  }
  if (gt < 3) db_ih[gt * H + u] = sacc;
  if (gt < 2) db_hh[gt * H + u] = sacc;
  if (gt == 3) db_hh[2 * H + u] = sacc;
}

// raise the dynamic-LDS limit of a kernel only when a larger request than any earlier one comes in
template <class K>
int ensure_lds(K kernel, size_t bytes, size_t* seen) {
  if (bytes > *seen) {
    PV_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    *seen = bytes;
  }
  return PV_OK;
}

int g_cu_count = 0;
int cu_count() {
  if (!g_cu_count) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      g_cu_count = n;
    else
      g_cu_count = 64;
  }
  return g_cu_count;
}

// c0 -> cell-state slot 0 and the running cell state (LSTM)
__global__ void lstm_pack_c0_kernel(const float* __restrict__ c0, int B, int H, float* __restrict__ cstate,
                                    float* __restrict__ call0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const float v = c0 ? c0[i] : 0.f;
  cstate[i] = v;
  call0[i] = v;
}

// groups mapped onto XCDs (group_slice) when every XCD can hold its share of workgroups, one per CU
int xcd_map(int G, int S, int cap) {
  if (const char* e = getenv("PV_GRU_XCD")) if (atoi(e) == 0) return 0;
  return cdiv(G, 8) * S <= cap / 8;
}

int rnn_geometry(int ngt, int B, int H, int* Hk, int* KG, int* Bpad, int* NG) {
  PV_REQUIRE(B > 0 && H > 0 && H % 16 == 0, "pv_gru/pv_lstm: hidden size %d must be a positive multiple of 16", H);
  PV_REQUIRE(H <= 1024, "pv_gru/pv_lstm: hidden size %d > 1024 not supported (LDS-resident weight slice)", H);
  // sync buffer layout: word NG = error flag, then up to 2 NG arrival counters, and the per-slice arrival flags from word
  // FLAGS_AT: beyond this batch the error word / counters would run into the flags
  if (3 * cdiv(B, ROWS) + 1 > FLAGS_AT) {
    pv_set_error("pv_gru/pv_lstm: batch %d exceeds the sync buffer layout's limit of %d rows", B, (FLAGS_AT - 1) / 3 * ROWS);
    return PV_ERR_LIMIT;
  }
  if (Hk) *Hk = round_up(H, 64);      // also the K of GEMMs that consume the bf16 state sequence
  if (KG) *KG = round_up(ngt * H, 64);
  if (Bpad) *Bpad = round_up(B, ROWS);
  if (NG) *NG = cdiv(B, ROWS);
  return PV_OK;
}

}  // namespace

extern "C" {

int pv_gru_geometry(int B, int H, int* Hk, int* K3, int* Bpad, int* NG) { return rnn_geometry(3, B, H, Hk, K3, Bpad, NG); }
int pv_lstm_geometry(int B, int H, int* Hk, int* K4, int* Bpad, int* NG) { return rnn_geometry(4, B, H, Hk, K4, Bpad, NG); }

}  // extern "C"
namespace {
__global__ void rnn_err_inject_kernel() { __hip_atomic_store(&g_rnn_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace
extern "C" {

int pv_rnn_sync_error_inject(void* stream) {
  hipLaunchKernelGGL(rnn_err_inject_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream);
  PV_LAUNCH_CHECK("rnn_err_inject_kernel");
  return PV_OK;
}

int pv_rnn_sync_error(int* flag, int clear) {
  PV_REQUIRE(flag, "pv_rnn_sync_error: null argument");
  unsigned v = 0;
  PV_HIP(hipDeviceSynchronize());
  PV_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_rnn_err), sizeof(v)));
  *flag = (int)v;
  if (v && clear) {
    const unsigned zero = 0;
    PV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rnn_err), &zero, sizeof(zero)));
  }
  return PV_OK;
}

size_t pv_gru_sync_bytes(int B) {
  // [NG]: unused, [NG]: error flag, then up to 2*NG arrival counters (diagnostic build: stamps from word 64); from word
  // FLAGS_AT: 2*NG row groups x FLAG_STRIDE arrival flags
  return ((size_t)FLAGS_AT + (size_t)2 * cdiv(B, ROWS) * FLAG_STRIDE) * sizeof(u32);
}

}  // extern "C"

namespace {

int rnn_fwd(int ngt, const float* xprojT, long ldx, const void* whh_bf16, const float* bhh, const float* h0,
            const float* c0, const int* lengths, int L, int B, int H, int reverse, void* hseq_bf16, float* gsaveT,
            float* hall, float* call, float* hn, float* cn, void* sync, int max_wgs, void* stream) {
  int Hk, K3, Bpad, NG;
  int rc = rnn_geometry(ngt, B, H, &Hk, &K3, &Bpad, &NG);
  if (rc) return rc;
  PV_REQUIRE(xprojT && whh_bf16 && bhh && hseq_bf16 && hall && hn && sync && L > 0, "pv_gru_fwd/pv_lstm_fwd: null/empty argument");
  PV_REQUIRE(ngt == 3 || (call && cn), "pv_lstm_fwd: null cell-state buffer");
  PV_REQUIRE(((uintptr_t)whh_bf16 & 15) == 0 && ((uintptr_t)hseq_bf16 & 15) == 0, "pv_gru_fwd: bf16 buffers must be 16-byte aligned");
  PV_REQUIRE(ldx >= (long)L * Bpad && ldx % 4 == 0 && ((uintptr_t)xprojT & 15) == 0 && ((uintptr_t)gsaveT & 15) == 0,
             "pv_gru_fwd: xprojT needs ldx >= L*Bpad = %ld, ldx %% 4 == 0 and 16-byte alignment (also gsaveT)", (long)L * Bpad);
  hipStream_t st = (hipStream_t)stream;
  const size_t hseq_bytes = (size_t)(L + 1) * Bpad * Hk * 2;
  PV_REQUIRE(hseq_bytes < (1ull << 31), "pv_gru_fwd: state sequence of %zu bytes exceeds the 2 GiB buffer-addressing limit", hseq_bytes);
  int UT = (H % 32 == 0) ? 2 : 1;
  // the weight slice, the staging areas and (preferably) the shared h_{t-1} copy must fit the 160 KiB LDS
  const size_t nstage = ngt == 4 ? 6 : 5;
  if (UT == 2 && (size_t)ngt * 32 * Hk * 2 + ROWS * 32 * 2 + nstage * 32 * ROWS * 4 + 4 * 32 * 4 * 4 + (ngt == 4 ? (size_t)ROWS * Hk * 2 : 0) > 160 * 1024) UT = 1;
  const int U = UT * 16, S = H / U;
  const int cap = max_wgs > 0 ? max_wgs : cu_count();
  PV_REQUIRE(S <= cap, "pv_gru_fwd: %d unit slices exceed the %d co-resident workgroups available", S, cap);
  // Group shape: 32-row groups, one per workgroup when they all fit.  (16-row groups, two per workgroup processed
  // alternately, were measured slower at B=256, H=512 -- 544 vs 430 us -- because the per-group cost is dominated by
  // fixed barrier / latency terms, not by rows; kept selectable with PV_GRU_MT=1 for experiments.)
  int MT = 2;
  if (const char* e = getenv("PV_GRU_MT")) MT = (atoi(e) == 1 && ngt == 3) ? 1 : 2;
  const int RW = MT * 16, NGk = Bpad / RW;
  int G = NGk < cap / S ? NGk : cap / S;
  if (MT == 1 && G > (NGk + 1) / 2) G = (NGk + 1) / 2;         // two groups per workgroup
  if (const char* e = getenv("PV_GRU_G")) { const int v = atoi(e); if (v >= 1 && v <= G) G = v; }
  const int xmap = xcd_map(G, S, cap), grid = xmap ? 8 * cdiv(G, 8) * S : G * S;
  u32* err = (u32*)sync + NG;
  u32* counters = err + 1;
  if (Hk != H || Bpad != B) PV_HIP(hipMemsetAsync(hseq_bf16, 0, (size_t)(L + 1) * Bpad * Hk * 2, st));
  if (Bpad != B) {
    PV_HIP(hipMemsetAsync(hall, 0, (size_t)Bpad * H * sizeof(float), st));      // slot 0 rows >= B are read as states
    if (ngt == 4) PV_HIP(hipMemsetAsync(call, 0, (size_t)Bpad * H * sizeof(float), st));
  }
  const int sync_words = (int)(pv_gru_sync_bytes(B) / sizeof(u32));       // the grid must cover them too (small B * H)
  hipLaunchKernelGGL(gru_pack_h0_kernel, dim3(cdiv(B * H > sync_words ? B * H : sync_words, 256)), dim3(256), 0, st, h0, B, H, Hk, (bf16_t*)hseq_bf16, hn, hall,
                     (u32*)sync, (int)(pv_gru_sync_bytes(B) / sizeof(u32)));
  PV_LAUNCH_CHECK("gru_pack_h0_kernel");
  if (ngt == 4) {
    hipLaunchKernelGGL(lstm_pack_c0_kernel, dim3(cdiv(B * H, 256)), dim3(256), 0, st, c0, B, H, cn, call);
    PV_LAUNCH_CHECK("lstm_pack_c0_kernel");
  }
  const size_t smem_base = (size_t)ngt * U * Hk * 2 + (size_t)RW * U * 2 + nstage * U * RW * 4 + (size_t)4 * U * 4 * 4,   // (+ the pitch padding of sg)
               smem_a = (size_t)RW * Hk * 2;
  const int a_lds = smem_base + smem_a <= 160 * 1024;       // share the group's h_{t-1} through LDS when it fits
  const size_t smem = smem_base + (a_lds ? smem_a : 0);
  GruFwdArgs a{xprojT, ldx, (const bf16_t*)whh_bf16, bhh, (bf16_t*)hseq_bf16, gsaveT, hn, hall,
               counters, err, (u32*)sync + 64, lengths, cn, call, L, B, H, Hk, Bpad, NGk, G, S, reverse, a_lds, xmap,
               (unsigned)hseq_bytes, 0, nullptr};
  if (const char* e = getenv("PV_GRU_DBG")) a.dbg = atoi(e);
  static const int use_flags = getenv("PV_GRU_FLAGS") ? atoi(getenv("PV_GRU_FLAGS")) : 1;   // 0: the counter form
  if (use_flags && S <= FLAG_STRIDE) a.flags = (u32*)sync + FLAGS_AT;
#define PV_GRU_LAUNCH_F4(UTv, NKv, MTv, NGv)                                                          \
  do {                                                                                                \
    static size_t seen = 0;                                                                           \
    if ((rc = ensure_lds(gru_fwd_kernel<UTv, NKv, MTv, NGv>, smem, &seen))) return rc;                \
    hipLaunchKernelGGL((gru_fwd_kernel<UTv, NKv, MTv, NGv>), dim3(grid), dim3(NT_F), smem, st, a);   \
  } while (0)
#define PV_GRU_LAUNCH_F(UTv, NKv, MTv) PV_GRU_LAUNCH_F4(UTv, NKv, MTv, 3)
  if (ngt == 4) {
    if (UT == 1 && Hk == 512 && a_lds) PV_GRU_LAUNCH_F4(1, 16, 2, 4);
    else if (UT == 2 && Hk == 256 && a_lds) PV_GRU_LAUNCH_F4(2, 8, 2, 4);
    else if (UT == 2) PV_GRU_LAUNCH_F4(2, 0, 2, 4);
    else PV_GRU_LAUNCH_F4(1, 0, 2, 4);
  } else if (MT == 1) {
    if (UT == 2 && Hk == 512 && a_lds) PV_GRU_LAUNCH_F(2, 16, 1);
    else if (UT == 2 && Hk == 256 && a_lds) PV_GRU_LAUNCH_F(2, 8, 1);
    else if (UT == 2) PV_GRU_LAUNCH_F(2, 0, 1);
    else PV_GRU_LAUNCH_F(1, 0, 1);
  } else {
    if (UT == 2 && Hk == 512 && a_lds) PV_GRU_LAUNCH_F(2, 16, 2);
    else if (UT == 2 && Hk == 256 && a_lds) PV_GRU_LAUNCH_F(2, 8, 2);
    else if (UT == 2) PV_GRU_LAUNCH_F(2, 0, 2);
    else PV_GRU_LAUNCH_F(1, 0, 2);
  }
#undef PV_GRU_LAUNCH_F
#undef PV_GRU_LAUNCH_F4
  PV_LAUNCH_CHECK("gru_fwd_kernel");
  return PV_OK;
}

int rnn_bwd(int ngt, const float* dout, long dout_sb, long dout_st, const float* dhn, const float* dcn,
            const float* gsaveT, const float* hall, const float* call, const void* whhT_bf16, int L, int B, int H,
            int reverse, void* dgi_bf16, void* dgh_bf16, void* dgx_bf16, float* dh0, float* dc0, float* dbias_part,
            float* db_ih, float* db_hh, const int* lengths, void* sync, int max_wgs, void* stream) {
  int Hk, K3, Bpad, NG;
  int rc = rnn_geometry(ngt, B, H, &Hk, &K3, &Bpad, &NG);
  if (rc) return rc;
  PV_REQUIRE(gsaveT && hall && whhT_bf16 && dgh_bf16 && dh0 && dbias_part && db_ih && db_hh && sync && L > 0,
             "pv_gru_bwd/pv_lstm_bwd: null/empty argument");
  PV_REQUIRE(ngt == 4 ? (call && dc0) : (dgi_bf16 != nullptr), "pv_gru_bwd/pv_lstm_bwd: null/empty argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t dg_bytes = (size_t)L * Bpad * K3 * 2;
  PV_REQUIRE(dg_bytes < (1ull << 31), "pv_gru_bwd: gate-gradient buffer of %zu bytes exceeds the 2 GiB buffer-addressing limit", dg_bytes);
  auto lds_need = [&](int ut) {
    const size_t u = (size_t)ut * 16;
    return u * K3 * 2 + (size_t)4 * 2 * ut * 64 * 16 + (ngt == 4 ? 1 : 2) * (size_t)ROWS * ngt * u * 2 + (size_t)8 * 4 * u * 4;
  };                                                     // sized for 32-row groups; 16-row groups need less
  int UT = (H % 32 == 0) ? 2 : 1;
  if (UT == 2 && lds_need(2) > 160 * 1024) UT = 1;
  const int U = UT * 16, S = H / U;
  const int cap = max_wgs > 0 ? max_wgs : cu_count();
  PV_REQUIRE(S <= cap, "pv_gru_bwd: %d unit slices exceed the %d co-resident workgroups available", S, cap);
  // Group shape: 16-row groups when all of them fit the chip at one workgroup per CU (B=256, H=512: 16 x 16 = 256
  // workgroups instead of 8 x 16), because a step is bound by fetching the group's exchanged gate gradients
  // (rows x 3H bf16 per workgroup) and that halves; 32-row groups otherwise.  PV_GRU_BWD_MT=2 forces the latter.
  int MT = (Bpad / 16) * S <= cap ? 1 : 2;
  if (const char* e = getenv("PV_GRU_BWD_MT")) MT = atoi(e) == 1 ? MT : 2;
  const int NGk = Bpad / (MT * 16);
  const int G = NGk < cap / S ? NGk : cap / S;
  PV_REQUIRE(!lengths || NGk <= G, "pv_lstm_bwd: variable lengths need one row group per workgroup (B=%d too large)", B);
  const int xmap = xcd_map(G, S, cap), grid = xmap ? 8 * cdiv(G, 8) * S : G * S;
  const size_t smem = lds_need(UT);
  PV_REQUIRE(smem <= 160 * 1024, "pv_gru_bwd: LDS need %zu exceeds 160 KiB", smem);
  u32* err = (u32*)sync + NG;
  u32* counters = err + 1;
  {
    const long n_state = (long)B * H, n_part = (long)2 * NG * 4 * H, n_sync = (long)(pv_gru_sync_bytes(B) / sizeof(u32));
    const long n = n_state > n_part ? (n_state > n_sync ? n_state : n_sync) : (n_part > n_sync ? n_part : n_sync);
    hipLaunchKernelGGL(rnn_bwd_zero_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (u32*)sync,
                       (int)(pv_gru_sync_bytes(B) / sizeof(u32)), dh0, ngt == 4 ? dc0 : nullptr, n_state, dbias_part, n_part);
    PV_LAUNCH_CHECK("rnn_bwd_zero_kernel");
  }
  if (K3 != ngt * H || Bpad != B) {
    if (dgi_bf16) PV_HIP(hipMemsetAsync(dgi_bf16, 0, (size_t)L * Bpad * K3 * 2, st));
    PV_HIP(hipMemsetAsync(dgh_bf16, 0, (size_t)L * Bpad * K3 * 2, st));
    if (dgx_bf16) PV_HIP(hipMemsetAsync(dgx_bf16, 0, (size_t)L * Bpad * K3 * 2, st));
  }
  GruBwdArgs a{dout, dout_sb, dout_st, dhn, gsaveT, hall, (const bf16_t*)whhT_bf16, (bf16_t*)dgi_bf16,
               (bf16_t*)dgh_bf16, ngt == 4 ? dc0 : dh0, call, dcn, dh0, dbias_part, counters, err, L, B, H, K3, Bpad,
               NGk, G, S, reverse, xmap, (unsigned)dg_bytes, (u32*)sync + 64, lengths, (bf16_t*)dgx_bf16, nullptr};
  static const int use_flags = getenv("PV_GRU_FLAGS") ? atoi(getenv("PV_GRU_FLAGS")) : 1;   // 0: the counter form
  if (use_flags && S <= FLAG_STRIDE) a.flags = (u32*)sync + FLAGS_AT;
  PV_REQUIRE(!dgx_bf16 || (ngt == 3 && ((uintptr_t)dgx_bf16 & 15) == 0), "pv_gru_bwd: dgx_bf16 is a GRU-only, 16-byte aligned buffer");
#define PV_GRU_LAUNCH_B5(UTv, NKv, NGv, MTv)                                                            \
  do {                                                                                                  \
    static size_t seen = 0;                                                                             \
    if (NGv == 3 && a.dgx) {                                                                            \
      static size_t seen_x = 0;                                                                         \
      if ((rc = ensure_lds(gru_bwd_kernel<UTv, NKv, 3, MTv, true>, smem, &seen_x))) return rc;          \
      hipLaunchKernelGGL((gru_bwd_kernel<UTv, NKv, 3, MTv, true>), dim3(grid), dim3(256), smem, st, a); \
      break;                                                                                            \
    }                                                                                                   \
    if ((rc = ensure_lds(gru_bwd_kernel<UTv, NKv, NGv, MTv, false>, smem, &seen))) return rc;           \
    hipLaunchKernelGGL((gru_bwd_kernel<UTv, NKv, NGv, MTv, false>), dim3(grid), dim3(256), smem, st, a); \
  } while (0)
#define PV_GRU_LAUNCH_B4(UTv, NKv, NGv)                     \
  do {                                                      \
    if (MT == 1) PV_GRU_LAUNCH_B5(UTv, NKv, NGv, 1);        \
    else PV_GRU_LAUNCH_B5(UTv, NKv, NGv, 2);                \
  } while (0)
#define PV_GRU_LAUNCH_B(UTv, NKv) PV_GRU_LAUNCH_B4(UTv, NKv, 3)
  if (ngt == 4) {
    if (UT == 2 && K3 == 2048) PV_GRU_LAUNCH_B4(2, 64, 4);
    else if (UT == 2) PV_GRU_LAUNCH_B4(2, 0, 4);
    else PV_GRU_LAUNCH_B4(1, 0, 4);
  } else if (UT == 2 && K3 == 1536) PV_GRU_LAUNCH_B(2, 48);
  else if (UT == 2 && K3 == 768) PV_GRU_LAUNCH_B(2, 24);
  else if (UT == 2) PV_GRU_LAUNCH_B(2, 0);
  else PV_GRU_LAUNCH_B(1, 0);
#undef PV_GRU_LAUNCH_B
#undef PV_GRU_LAUNCH_B4
#undef PV_GRU_LAUNCH_B5
  PV_LAUNCH_CHECK("gru_bwd_kernel");
  hipLaunchKernelGGL(gru_bias_reduce_kernel, dim3(cdiv(4 * H, 256)), dim3(256), 0, st, dbias_part, G, H, ngt == 4 ? 1 : 0,
                     db_ih, db_hh);
  PV_LAUNCH_CHECK("gru_bias_reduce_kernel");
  return PV_OK;
}

}  // namespace

extern "C" {

int pv_gru_fwd(const float* xprojT, long ldx, const void* whh_bf16, const float* bhh, const float* h0,
               const int* lengths, int L, int B, int H, int reverse, void* hseq_bf16, float* gsaveT, float* hall,
               float* hn, void* sync, int max_wgs, void* stream) {
  return rnn_fwd(3, xprojT, ldx, whh_bf16, bhh, h0, nullptr, lengths, L, B, H, reverse, hseq_bf16, gsaveT, hall, nullptr,
                 hn, nullptr, sync, max_wgs, stream);
}

int pv_lstm_fwd(const float* xprojT, long ldx, const void* whh_bf16, const float* bhh, const float* h0,
                const float* c0, const int* lengths, int L, int B, int H, int reverse, void* hseq_bf16, float* gsaveT,
                float* hall, float* call, float* hn, float* cn, void* sync, int max_wgs, void* stream) {
  return rnn_fwd(4, xprojT, ldx, whh_bf16, bhh, h0, c0, lengths, L, B, H, reverse, hseq_bf16, gsaveT, hall, call, hn, cn,
                 sync, max_wgs, stream);
}

int pv_gru_bwd(const float* dout, long dout_sb, long dout_st, const float* dhn, const float* gsaveT,
               const float* hall, const void* whhT_bf16, int L, int B, int H, int reverse, void* dgi_bf16,
               void* dgh_bf16, void* dgx_bf16, float* dh0, float* dbias_part, float* db_ih, float* db_hh, void* sync,
               int max_wgs, void* stream) {
  return rnn_bwd(3, dout, dout_sb, dout_st, dhn, nullptr, gsaveT, hall, nullptr, whhT_bf16, L, B, H, reverse, dgi_bf16,
                 dgh_bf16, dgx_bf16, dh0, nullptr, dbias_part, db_ih, db_hh, nullptr, sync, max_wgs, stream);
}

int pv_lstm_bwd(const float* dout, long dout_sb, long dout_st, const float* dhn, const float* dcn,
                const float* gsaveT, const float* hall, const float* call, const void* whhT_bf16, int L, int B, int H,
                int reverse, void* dg_bf16, float* dh0, float* dc0, float* dbias_part, float* db_ih, float* db_hh,
                const int* lengths, void* sync, int max_wgs, void* stream) {
  return rnn_bwd(4, dout, dout_sb, dout_st, dhn, dcn, gsaveT, hall, call, whhT_bf16, L, B, H, reverse, nullptr, dg_bf16,
                 nullptr, dh0, dc0, dbias_part, db_ih, db_hh, lengths, sync, max_wgs, stream);
}

}  // extern "C"
