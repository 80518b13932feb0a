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
// vmcnt(0); workgroup barrier; ONE lane adds to the group's arrival counter (agent-scope atomic).
// Consumer: one lane polls the counter with sc1 loads, workgroup barrier, then EVERY load of the
// handed-off bytes is a 16-byte `buffer_load ... sc1` to registers -- no L1 invalidate is needed because
// no plain load ever touches those bytes.  Batch groups are independent, so a barrier spans only the S
// workgroups of one group.  Nothing depends on dispatch order or XCD placement; every spin is bounded
// (wall clock) and reports through `err` instead of hanging.  The grid is sized by the host so that all
// workgroups are co-resident (1 per CU; <= the device's CU count).
#include "common.h"
#include "../../include/pevae_hip.h"

namespace {

constexpr int ROWS = 32;                    // batch rows per group = 2 MFMA row tiles
constexpr unsigned long long SPIN_LIMIT = 300000000ull;   // 3 s of s_memrealtime (100 MHz)

typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;                 // cache-policy bits of the raw buffer builtins: sc1

__device__ __forceinline__ u32 ld_relaxed(const u32* p) {      // global_load_dword ... sc1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane waits for *ctr >= want; returns false on timeout / error already flagged
__device__ __forceinline__ bool wait_ge(const u32* ctr, u32 want, u32* err) {
  if (ld_relaxed(ctr) >= want) return true;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    __builtin_amdgcn_s_sleep(1);
    if (ld_relaxed(ctr) >= want) return true;
    if (ld_relaxed(err) != 0) return false;
    if (wall_clock64() - t0 > SPIN_LIMIT) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

__device__ __forceinline__ int swz_mask_for(int nchunks) {
  // largest power of two <= 16 dividing nchunks, minus 1
  int m = 16;
  while (nchunks % m) m >>= 1;
  return m - 1;
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f); }

struct GruFwdArgs {
  const float* xproj;     // [L][Bpad][3H]  (W_ih x + b_ih), time-major
  const bf16_t* whh;      // [3H][Hk] bf16, columns >= H zero
  const float* bhh;       // [3H]
  float* out;             // optional: out[b*out_sb + t*out_st + u]
  long out_sb, out_st;
  bf16_t* hseq;           // [(L+1)][Bpad][Hk]; slot i = state before step i (slot 0 pre-filled)
  float* gsave;           // optional [L][Bpad][4][H]: r, z, n, (W_hn h + b_hn)
  float* hstate;          // [B][H] running state (pre-filled with h0), ends as h_n
  float* hall;            // optional [(L+1)][Bpad][H] fp32 states in processing order (slot 0 pre-filled)
  u32* counters;          // [NG] zeroed
  u32* err;
  int L, B, H, Hk, Bpad, NG, G, S, reverse;
  unsigned hseq_bytes;
};

template <int UT>
__global__ __launch_bounds__(256, 1) void gru_fwd_kernel(GruFwdArgs a) {
  constexpr int U = UT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int mt = w & 1, ut = w >> 1;
  const bool active = ut < UT;                                // wave owns tile (mt, ut)
  const int g = blockIdx.x / a.S, s = blockIdx.x % a.S, u0 = s * U;
  const int Hk = a.Hk, H = a.H, nch = Hk / 8, mask = swz_mask_for(nch), NKS = Hk / 32;
  char* wl = smem;                                            // [3U][Hk] bf16, swizzled
  bf16_t* tile = (bf16_t*)(smem + (size_t)3 * U * Hk * 2);    // [ROWS][U] bf16: h_t of this slice

  for (int i = tid; i < 3 * U * nch; i += 256) {              // resident weight slice
    const int n = i / nch, ch = i % nch, gate = n / U, u = n % U;
    const uint4 v = *(const uint4*)(a.whh + (size_t)(gate * H + u0 + u) * Hk + ch * 8);
    *(uint4*)(wl + (size_t)n * Hk * 2 + ((ch ^ (n & mask)) * 16)) = v;
  }
  const int unit = u0 + (active ? ut : 0) * 16 + c16;
  const float bh_r = a.bhh[unit], bh_z = a.bhh[H + unit], bh_n = a.bhh[2 * H + unit];
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.hseq, 0, a.hseq_bytes, 0x00020000);
  const bool single = a.NG <= a.G;                            // one row group per workgroup: state stays in registers
  float hp[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int step = 0; step < a.L; ++step) {
    const int t = a.reverse ? a.L - 1 - step : step;
    for (int rg = g; rg < a.NG; rg += a.G) {
      const int row0 = rg * ROWS, rbase = row0 + mt * 16 + q4 * 4;
      // (a) independent of h_{t-1}: this lane's input projections (and state, when it is not register resident)
      float xr[4], xz[4], xn[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xr[j] = xz[j] = xn[j] = 0.f;
        if (active && rbase + j < a.B) {
          const float* xp = a.xproj + ((size_t)t * a.Bpad + rbase + j) * 3 * H + unit;
          xr[j] = xp[0];
          xz[j] = xp[H];
          xn[j] = xp[2 * H];
          if (!single || step == 0) hp[j] = a.hstate[(size_t)(rbase + j) * H + unit];
        }
      }
      // (b) wait for the group's h_{step-1}
      if (step > 0) {
        if (tid == 0) wait_ge(a.counters + rg, (u32)(a.S * step), a.err);
        __syncthreads();
      }
      // (c) three gate tiles over the full K; A fragments straight from the published bf16 states (sc1)
      f32x4 acc[3];
#pragma unroll
      for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (active) {
        const unsigned abase = (unsigned)((((size_t)step * a.Bpad + row0 + mt * 16 + c16) * Hk + q4 * 8) * 2);
        for (int ks0 = 0; ks0 < NKS; ks0 += 16) {
          u32x4 av[16];
#pragma unroll
          for (int kk = 0; kk < 16; ++kk)
            if (ks0 + kk < NKS) av[kk] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, abase + (ks0 + kk) * 64, 0, AUX_SC1);
#pragma unroll
          for (int kk = 0; kk < 16; ++kk) {
            const int ks = ks0 + kk;
            if (ks < NKS) {
#pragma unroll
              for (int gt = 0; gt < 3; ++gt) {
                const int n = gt * U + ut * 16 + c16;
                const bf16x8 b = __builtin_bit_cast(
                    bf16x8, *(const uint4*)(wl + (size_t)n * Hk * 2 + (((ks * 4 + q4) ^ (n & mask)) * 16)));
                acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[kk]), b, acc[gt], 0, 0, 0);
              }
            }
          }
        }
      }
      // (d) gate math on the accumulator layout: lane (c16, q4) holds rows rbase..rbase+3 of unit `unit`
      float rr[4], zz[4], nn[4], hn[4], hv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = active && rbase + j < a.B;
        hn[j] = acc[2][j] + bh_n;
        rr[j] = fast_sigmoid(xr[j] + acc[0][j] + bh_r);
        zz[j] = fast_sigmoid(xz[j] + acc[1][j] + bh_z);
        nn[j] = fast_tanh(xn[j] + rr[j] * hn[j]);
        hv[j] = live ? (1.f - zz[j]) * nn[j] + zz[j] * hp[j] : 0.f;
        hp[j] = hv[j];
        if (active) tile[(mt * 16 + q4 * 4 + j) * U + ut * 16 + c16] = f2bf(hv[j]);
      }
      __syncthreads();
      // (e) publish h_step: one 16-byte write-through store per thread, drain, barrier, one arrival
      if (tid < ROWS * U / 8) {
        const int row = tid / (U / 8), ch = tid % (U / 8);
        const u32x4 v = *(const u32x4*)(tile + row * U + ch * 8);
        const unsigned off = (unsigned)((((size_t)(step + 1) * a.Bpad + row0 + row) * Hk + u0 + ch * 8) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, hrsrc, off, 0, AUX_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.counters + rg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::: "memory");
      // (f) everything the next step does not wait for
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int brow = rbase + j;
        if (active && brow < a.B) {
          if (a.gsave) {
            float* gs = a.gsave + (((size_t)t * a.Bpad + brow) * 4) * H + unit;
            gs[0] = rr[j];
            gs[H] = zz[j];
            gs[2 * H] = nn[j];
            gs[3 * H] = hn[j];
          }
          if (!single || step == a.L - 1) a.hstate[(size_t)brow * H + unit] = hv[j];
          if (a.hall) a.hall[((size_t)(step + 1) * a.Bpad + brow) * H + unit] = hv[j];
          if (a.out) a.out[brow * a.out_sb + t * a.out_st + unit] = hv[j];
        }
      }
    }
  }
}

struct GruBwdArgs {
  const float* dout;      // optional dL/d out: dout[b*sb + t*st + u]
  long dout_sb, dout_st;
  const float* dhn;       // optional [B][H]
  const float* gsave;     // [L][Bpad][4][H]
  const float* hall;      // [(L+1)][Bpad][H] fp32 states in processing order (slot i = state before step i)
  const bf16_t* whhT;     // [H][K3] bf16: whhT[u][k] = W_hh[k][u], columns >= 3H zero
  bf16_t* dgi;            // [L][Bpad][K3] bf16: (dr, dz, dn)       rows >= B zero
  bf16_t* dgh;            // [L][Bpad][K3] bf16: (dr, dz, dn * r)   exchange buffer
  float* dcarry;          // [B][H] fp32 scratch; ends as dL/dh0
  float* dbias_part;      // [G][4][H]: sums over rows & time of dr, dz, dn, dn*r
  u32* counters;          // [NG] zeroed
  u32* err;
  int L, B, H, K3, Bpad, NG, G, S, reverse;
  unsigned dg_bytes;
};

// Backward through time.  Per step: rec = dgh[previously handled step] @ W_hh (K = 3H, split over the 4
// waves, partials combined through LDS), then the elementwise gate backward on the MFMA output layout
// (wave (mt, ut) owns tile (mt, ut) as in the forward kernel), then the bf16 gate gradients go through an
// LDS tile so that the exchanged copy (dgh) leaves as 16-byte write-through stores; the same hand-off
// protocol as the forward kernel.
template <int UT>
__global__ __launch_bounds__(256, 1) void gru_bwd_kernel(GruBwdArgs a) {
  constexpr int U = UT * 16, NTILE = 2 * UT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int mt = w & 1, ut = w >> 1;
  const bool active = ut < UT;
  const int g = blockIdx.x / a.S, s = blockIdx.x % a.S, u0 = s * U;
  const int K3 = a.K3, nch = K3 / 8, mask = swz_mask_for(nch), NKS = K3 / 32, H = a.H;
  char* wl = smem;                                                   // [U][K3] bf16 swizzled (W_hh^T slice)
  f32x4* part = (f32x4*)(smem + (size_t)U * K3 * 2);                 // [4 waves][NTILE][64 lanes]
  bf16_t* tgh = (bf16_t*)(part + 4 * NTILE * 64);                    // [ROWS][3][U] bf16: dr, dz, dn*r
  bf16_t* tgi = tgh + ROWS * 3 * U;                                  // [ROWS][3][U] bf16: dr, dz, dn
  float* bred = (float*)(tgi + ROWS * 3 * U);                        // [8][4][U] bias-grad reduction

  for (int i = tid; i < U * nch; i += 256) {
    const int n = i / nch, ch = i % nch;
    const uint4 v = *(const uint4*)(a.whhT + (size_t)(u0 + n) * K3 + ch * 8);
    *(uint4*)(wl + (size_t)n * K3 * 2 + ((ch ^ (n & mask)) * 16)) = v;
  }
  const int unit = u0 + (active ? ut : 0) * 16 + c16;
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dgh, 0, a.dg_bytes, 0x00020000);
  const bool single = a.NG <= a.G;
  float carry[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // steps run from the last processed time step back to the first; one extra pass (step == L)
  // only folds the recurrent term of the first step into dL/dh0.
  for (int step = 0; step <= a.L; ++step) {
    const int ti = a.L - 1 - step;                       // processing index whose gradients we form
    const int t = a.reverse ? a.L - 1 - ti : ti;         // time index of that step
    for (int rg = g; rg < a.NG; rg += a.G) {
      const int row0 = rg * ROWS, rbase = row0 + mt * 16 + q4 * 4;
      // (a) independent of the exchange: saved gates / states / upstream gradients of this lane's 4 rows
      float gr[4], gz[4], gn[4], ghn[4], ghp[4], gdo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gr[j] = gz[j] = gn[j] = ghn[j] = ghp[j] = gdo[j] = 0.f;
        const int brow = rbase + j;
        if (active && brow < a.B && step < a.L) {
          const float* gs = a.gsave + (((size_t)t * a.Bpad + brow) * 4) * H + unit;
          gr[j] = gs[0];
          gz[j] = gs[H];
          gn[j] = gs[2 * H];
          ghn[j] = gs[3 * H];
          ghp[j] = a.hall[((size_t)ti * a.Bpad + brow) * H + unit];
          if (a.dout) gdo[j] = a.dout[brow * a.dout_sb + t * a.dout_st + unit];
          if (step == 0) {
            if (a.dhn) gdo[j] += a.dhn[(size_t)brow * H + unit];
          } else if (!single) {
            carry[j] = a.dcarry[(size_t)brow * H + unit];
          }
        }
      }
      // (b) recurrent term from the previously handled step: dgh[prev] @ W_hh  (K = 3H)
      float rec[4] = {0.f, 0.f, 0.f, 0.f};
      if (step > 0) {
        if (tid == 0) wait_ge(a.counters + rg, (u32)(a.S * step), a.err);
        __syncthreads();
        const int tprev = a.reverse ? a.L - 1 - (ti + 1) : ti + 1;
        f32x4 acc[2][UT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < UT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned abase = (unsigned)((((size_t)tprev * a.Bpad + row0 + c16) * K3 + q4 * 8) * 2);
        for (int ks0 = w; ks0 < NKS; ks0 += 48) {           // wave w takes k-steps w, w+4, ... (12 per pass)
          u32x4 av[12][2];
#pragma unroll
          for (int kk = 0; kk < 12; ++kk)
            if (ks0 + kk * 4 < NKS) {
#pragma unroll
              for (int i = 0; i < 2; ++i)
                av[kk][i] = __builtin_amdgcn_raw_buffer_load_b128(
                    grsrc, abase + (unsigned)(i * 16 * K3 * 2) + (ks0 + kk * 4) * 64, 0, AUX_SC1);
            }
#pragma unroll
          for (int kk = 0; kk < 12; ++kk) {
            const int ks = ks0 + kk * 4;
            if (ks < NKS) {
#pragma unroll
              for (int j = 0; j < UT; ++j) {
                const int n = j * 16 + c16;
                const bf16x8 b = __builtin_bit_cast(
                    bf16x8, *(const uint4*)(wl + (size_t)n * K3 * 2 + (((ks * 4 + q4) ^ (n & mask)) * 16)));
#pragma unroll
                for (int i = 0; i < 2; ++i)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[kk][i]), b, acc[i][j], 0, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < UT; ++j) part[(w * NTILE + i * UT + j) * 64 + lane] = acc[i][j];
        __syncthreads();
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
            a.dcarry[(size_t)brow * H + unit] = prev + rec[j];
          }
        }
        __syncthreads();          // the partial buffer is reused by the next row group
        continue;
      }
      float dr[4], dz[4], dn[4], dnr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = active && rbase + j < a.B;
        const float dh = rec[j] + gdo[j] + (step > 0 ? carry[j] : 0.f);
        dn[j] = dh * (1.f - gz[j]) * (1.f - gn[j] * gn[j]);
        dz[j] = dh * (ghp[j] - gn[j]) * gz[j] * (1.f - gz[j]);
        dr[j] = dn[j] * ghn[j] * gr[j] * (1.f - gr[j]);
        dnr[j] = dn[j] * gr[j];
        if (!live) dn[j] = dz[j] = dr[j] = dnr[j] = 0.f;
        carry[j] = dh * gz[j];
        bsum[0] += dr[j];
        bsum[1] += dz[j];
        bsum[2] += dn[j];
        bsum[3] += dnr[j];
        if (active) {
          const int o = (mt * 16 + q4 * 4 + j) * 3 * U + ut * 16 + c16;
          const bf16_t br = f2bf(dr[j]), bz = f2bf(dz[j]);
          tgh[o] = br;
          tgh[o + U] = bz;
          tgh[o + 2 * U] = f2bf(dnr[j]);
          tgi[o] = br;
          tgi[o + U] = bz;
          tgi[o + 2 * U] = f2bf(dn[j]);
        }
      }
      __syncthreads();
      // (d) publish dgh (write-through), drain, barrier, arrive; then the copies nobody waits for
      constexpr int NCH = ROWS * 3 * U / 8;                  // 16-byte chunks per image
      for (int c = tid; c < NCH; c += 256) {
        const int row = c / (3 * U / 8), rem = c % (3 * U / 8), gt = rem / (U / 8), ch = rem % (U / 8);
        const u32x4 v = *(const u32x4*)(tgh + row * 3 * U + gt * U + ch * 8);
        const unsigned off = (unsigned)((((size_t)t * a.Bpad + row0 + row) * K3 + gt * H + u0 + ch * 8) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, grsrc, off, 0, AUX_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.counters + rg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::: "memory");
      for (int c = tid; c < NCH; c += 256) {
        const int row = c / (3 * U / 8), rem = c % (3 * U / 8), gt = rem / (U / 8), ch = rem % (U / 8);
        const uint4 v = *(const uint4*)(tgi + row * 3 * U + gt * U + ch * 8);
        *(uint4*)(a.dgi + ((size_t)t * a.Bpad + row0 + row) * K3 + gt * H + u0 + ch * 8) = v;
      }
      if (!single) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (active && rbase + j < a.B) a.dcarry[(size_t)(rbase + j) * H + unit] = carry[j];
      }
    }
  }
  // bias-gradient partials: reduce this workgroup's rows in a fixed order
  {
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bred[((mt * 4 + q4) * 4 + i) * U + ut * 16 + c16] = bsum[i];
    }
    __syncthreads();
    for (int i = tid; i < 4 * U; i += 256) {
      float sacc = 0.f;
      for (int r = 0; r < 8; ++r) sacc += bred[r * 4 * U + i];
      a.dbias_part[((size_t)g * 4 + i / U) * H + u0 + i % U] = sacc;
    }
  }
}

// hseq slot 0 / hstate <- h0 (or zeros); also used to zero padded rows
__global__ void gru_pack_h0_kernel(const float* __restrict__ h0, int B, int H, int Hk, bf16_t* __restrict__ hseq0,
                                   float* __restrict__ hstate, float* __restrict__ hall0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, u = i % H;
  const float v = h0 ? h0[i] : 0.f;
  hseq0[(size_t)b * Hk + u] = f2bf(v);
  hstate[i] = v;
  if (hall0) hall0[i] = v;
}

__global__ void gru_bias_reduce_kernel(const float* __restrict__ part, int G, int H, float* __restrict__ db_ih,
                                       float* __restrict__ db_hh) {
  // part [G][4][H] -> db_ih [3H] = (dr, dz, dn), db_hh [3H] = (dr, dz, dn*r)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * H) return;
  float sacc = 0.f;
  for (int gidx = 0; gidx < G; ++gidx) sacc += part[(size_t)gidx * 4 * H + i];
  const int gt = i / H, u = i % H;
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

}  // namespace

extern "C" {

int pv_gru_geometry(int B, int H, int* Hk, int* K3, int* Bpad, int* NG) {
  PV_REQUIRE(B > 0 && H > 0 && H % 16 == 0, "pv_gru: hidden size %d must be a positive multiple of 16", H);
  PV_REQUIRE(H <= 1024, "pv_gru: hidden size %d > 1024 not supported (LDS-resident weight slice)", H);
  if (Hk) *Hk = round_up(H, 64);      // also the K of GEMMs that consume the bf16 state sequence
  if (K3) *K3 = round_up(3 * H, 64);
  if (Bpad) *Bpad = round_up(B, ROWS);
  if (NG) *NG = cdiv(B, ROWS);
  return PV_OK;
}

size_t pv_gru_sync_bytes(int B) { return ((size_t)cdiv(B, ROWS) + 1) * sizeof(u32); }

int pv_gru_fwd(const float* xproj, const void* whh_bf16, const float* bhh, const float* h0, int L, int B,
               int H, int reverse, float* out, long out_sb, long out_st, void* hseq_bf16, float* gsave,
               float* hall, float* hn, void* sync, int max_wgs, void* stream) {
  int Hk, K3, Bpad, NG;
  int rc = pv_gru_geometry(B, H, &Hk, &K3, &Bpad, &NG);
  if (rc) return rc;
  PV_REQUIRE(xproj && whh_bf16 && bhh && hseq_bf16 && hn && sync && L > 0, "pv_gru_fwd: null/empty argument");
  PV_REQUIRE(((uintptr_t)whh_bf16 & 15) == 0 && ((uintptr_t)hseq_bf16 & 15) == 0, "pv_gru_fwd: bf16 buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const size_t hseq_bytes = (size_t)(L + 1) * Bpad * Hk * 2;
  PV_REQUIRE(hseq_bytes < (1ull << 31), "pv_gru_fwd: state sequence of %zu bytes exceeds the 2 GiB buffer-addressing limit", hseq_bytes);
  int UT = (H % 32 == 0) ? 2 : 1;
  if (UT == 2 && (size_t)3 * 32 * Hk * 2 + ROWS * 32 * 2 > 160 * 1024) UT = 1;    // weight slice must fit the 160 KiB LDS
  const int U = UT * 16, S = H / U;
  const int cap = max_wgs > 0 ? max_wgs : cu_count();
  PV_REQUIRE(S <= cap, "pv_gru_fwd: %d unit slices exceed the %d co-resident workgroups available", S, cap);
  const int G = NG < cap / S ? NG : cap / S;
  u32* counters = (u32*)sync;
  u32* err = counters + NG;
  PV_HIP(hipMemsetAsync(sync, 0, pv_gru_sync_bytes(B), st));
  if (Hk != H || Bpad != B) PV_HIP(hipMemsetAsync(hseq_bf16, 0, (size_t)(L + 1) * Bpad * Hk * 2, st));
  hipLaunchKernelGGL(gru_pack_h0_kernel, dim3(cdiv(B * H, 256)), dim3(256), 0, st, h0, B, H, Hk, (bf16_t*)hseq_bf16, hn, hall);
  PV_LAUNCH_CHECK("gru_pack_h0_kernel");
  GruFwdArgs a{xproj, (const bf16_t*)whh_bf16, bhh, out, out_sb, out_st, (bf16_t*)hseq_bf16, gsave, hn, hall,
               counters, err, L, B, H, Hk, Bpad, NG, G, S, reverse, (unsigned)hseq_bytes};
  const size_t smem = (size_t)3 * U * Hk * 2 + (size_t)ROWS * U * 2;
  static size_t seen_f2 = 0, seen_f1 = 0;
  if (UT == 2) {
    if ((rc = ensure_lds(gru_fwd_kernel<2>, smem, &seen_f2))) return rc;
    hipLaunchKernelGGL(gru_fwd_kernel<2>, dim3(G * S), dim3(256), smem, st, a);
  } else {
    if ((rc = ensure_lds(gru_fwd_kernel<1>, smem, &seen_f1))) return rc;
    hipLaunchKernelGGL(gru_fwd_kernel<1>, dim3(G * S), dim3(256), smem, st, a);
  }
  PV_LAUNCH_CHECK("gru_fwd_kernel");
  return PV_OK;
}

int pv_gru_bwd(const float* dout, long dout_sb, long dout_st, const float* dhn, const float* gsave,
               const float* hall, const void* whhT_bf16, int L, int B, int H, int reverse, void* dgi_bf16,
               void* dgh_bf16, float* dh0, float* dbias_part, float* db_ih, float* db_hh, void* sync,
               int max_wgs, void* stream) {
  int Hk, K3, Bpad, NG;
  int rc = pv_gru_geometry(B, H, &Hk, &K3, &Bpad, &NG);
  if (rc) return rc;
  PV_REQUIRE(gsave && hall && whhT_bf16 && dgi_bf16 && dgh_bf16 && dh0 && dbias_part && db_ih && db_hh && sync && L > 0,
             "pv_gru_bwd: null/empty argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t dg_bytes = (size_t)L * Bpad * K3 * 2;
  PV_REQUIRE(dg_bytes < (1ull << 31), "pv_gru_bwd: gate-gradient buffer of %zu bytes exceeds the 2 GiB buffer-addressing limit", dg_bytes);
  auto lds_need = [&](int ut) {
    const size_t u = (size_t)ut * 16;
    return u * K3 * 2 + (size_t)4 * 2 * ut * 64 * 16 + 2 * (size_t)ROWS * 3 * u * 2 + (size_t)8 * 4 * u * 4;
  };
  int UT = (H % 32 == 0) ? 2 : 1;
  if (UT == 2 && lds_need(2) > 160 * 1024) UT = 1;
  const int U = UT * 16, S = H / U;
  const int cap = max_wgs > 0 ? max_wgs : cu_count();
  PV_REQUIRE(S <= cap, "pv_gru_bwd: %d unit slices exceed the %d co-resident workgroups available", S, cap);
  const int G = NG < cap / S ? NG : cap / S;
  const size_t smem = lds_need(UT);
  PV_REQUIRE(smem <= 160 * 1024, "pv_gru_bwd: LDS need %zu exceeds 160 KiB", smem);
  u32* counters = (u32*)sync;
  u32* err = counters + NG;
  PV_HIP(hipMemsetAsync(sync, 0, pv_gru_sync_bytes(B), st));
  PV_HIP(hipMemsetAsync(dh0, 0, (size_t)B * H * sizeof(float), st));
  PV_HIP(hipMemsetAsync(dbias_part, 0, (size_t)NG * 4 * H * sizeof(float), st));
  if (K3 != 3 * H || Bpad != B) {
    PV_HIP(hipMemsetAsync(dgi_bf16, 0, (size_t)L * Bpad * K3 * 2, st));
    PV_HIP(hipMemsetAsync(dgh_bf16, 0, (size_t)L * Bpad * K3 * 2, st));
  }
  GruBwdArgs a{dout, dout_sb, dout_st, dhn, gsave, hall, (const bf16_t*)whhT_bf16, (bf16_t*)dgi_bf16,
               (bf16_t*)dgh_bf16, dh0, dbias_part, counters, err, L, B, H, K3, Bpad, NG, G, S, reverse,
               (unsigned)dg_bytes};
  static size_t seen_b2 = 0, seen_b1 = 0;
  if (UT == 2) {
    if ((rc = ensure_lds(gru_bwd_kernel<2>, smem, &seen_b2))) return rc;
    hipLaunchKernelGGL(gru_bwd_kernel<2>, dim3(G * S), dim3(256), smem, st, a);
  } else {
    if ((rc = ensure_lds(gru_bwd_kernel<1>, smem, &seen_b1))) return rc;
    hipLaunchKernelGGL(gru_bwd_kernel<1>, dim3(G * S), dim3(256), smem, st, a);
  }
  PV_LAUNCH_CHECK("gru_bwd_kernel");
  hipLaunchKernelGGL(gru_bias_reduce_kernel, dim3(cdiv(4 * H, 256)), dim3(256), 0, st, dbias_part, G, H, db_ih, db_hh);
  PV_LAUNCH_CHECK("gru_bias_reduce_kernel");
  return PV_OK;
}

}  // extern "C"
