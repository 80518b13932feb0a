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
// sequence.  Every step it (1) waits until the S workgroups of ITS batch group have published
// h_{t-1}, (2) pulls the group's h_{t-1} rows (bf16) straight into MFMA A fragments (K split over the
// 4 waves), (3) multiplies against the LDS-resident weights with v_mfma_f32_16x16x32_bf16, (4) reduces
// the 4 K-partials through LDS and applies the gate math in-register (the 3 gates of a unit live in
// one thread), and (5) publishes its slice of h_t.
//
// Inter-workgroup hand-off (cdna guide, Guideline 16, recipe R1): payload = 8-byte write-through
// (sc1) stores; every storing wave drains vmcnt(0); barrier; ONE lane bumps the group's arrival
// counter (agent-scope atomic).  Consumer: one lane polls the counter relaxed, ONE agent-scope acquire,
// vmcnt(0), barrier, then plain loads.  Batch groups are independent, so a barrier spans only the S
// workgroups of one group.  Nothing depends on dispatch order or XCD placement; every spin is bounded
// (wall clock) and reports through `err` instead of hanging.  The grid is sized by the host so that all
// workgroups are co-resident (1 per CU; <= the device's CU count).
#include "common.h"
#include "../../include/pevae_hip.h"

namespace {

constexpr int MT = 2;                       // 16-row MFMA tiles per batch group  -> 32 rows
constexpr int ROWS = MT * 16;
constexpr unsigned long long SPIN_LIMIT = 300000000ull;   // 3 s of s_memrealtime (100 MHz)

typedef unsigned int u32;

__device__ __forceinline__ u32 ld_relaxed(const u32* p) {
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
};

template <int UT>
__global__ __launch_bounds__(256, 1) void gru_fwd_kernel(GruFwdArgs a) {
  constexpr int U = UT * 16, NT3 = 3 * UT, PPT = MT * UT;   // PPT consecutive units per thread
  constexpr int PLD = 3 * U + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, q4 = lane >> 4;
  const int g = blockIdx.x / a.S, s = blockIdx.x % a.S, u0 = s * U;
  const int Hk = a.Hk, nch = Hk / 8, mask = swz_mask_for(nch), NKS = Hk / 32;
  char* wl = smem;                                            // [3U][Hk] bf16, swizzled
  float* part = (float*)(smem + (size_t)3 * U * Hk * 2);      // [4][ROWS][PLD]

  for (int i = tid; i < 3 * U * nch; i += 256) {              // resident weight slice
    const int n = i / nch, ch = i % nch, gate = n / U, u = n % U;
    const uint4 v = *(const uint4*)(a.whh + (size_t)(gate * a.H + u0 + u) * Hk + ch * 8);
    *(uint4*)(wl + (size_t)n * Hk * 2 + ((ch ^ (n & mask)) * 16)) = v;
  }
  __syncthreads();

  const int prow = (tid * PPT) / U, pu = (tid * PPT) % U;     // this thread's (row, first unit)
  bool ok = true;
  for (int step = 0; step < a.L; ++step) {
    const int t = a.reverse ? a.L - 1 - step : step;
    for (int rg = g; rg < a.NG; rg += a.G) {
      const int row0 = rg * ROWS;
      const int brow = row0 + prow;
      const bool live = brow < a.B;
      // (a) independent of h: this thread's input projections and previous state
      float xr[PPT], xz[PPT], xn[PPT], hp[PPT];
      if (live) {
        const float* xp = a.xproj + ((size_t)t * a.Bpad + brow) * 3 * a.H + u0 + pu;
        const float* hs = a.hstate + (size_t)brow * a.H + u0 + pu;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          xr[k] = xp[k];
          xz[k] = xp[a.H + k];
          xn[k] = xp[2 * a.H + k];
          hp[k] = hs[k];
        }
      }
      // (b) wait for the group's h_{step-1}
      if (step > 0) {
        if (tid == 0) {
          ok = wait_ge(a.counters + rg, (u32)(a.S * step), a.err);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
      }
      // (c) K-split MFMA: wave w takes k-steps w, w+4, ...
      f32x4 acc[MT][NT3];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bf16_t* hsrc = a.hseq + ((size_t)step * a.Bpad + row0) * Hk;
      for (int ks0 = w; ks0 < NKS; ks0 += 16) {
        uint4 av[4][MT];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int ks = ks0 + kk * 4;
          if (ks < NKS) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
              av[kk][i] = *(const uint4*)(hsrc + (size_t)(i * 16 + c16) * Hk + ks * 32 + q4 * 8);
          }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int ks = ks0 + kk * 4;
          if (ks < NKS) {
#pragma unroll
            for (int j = 0; j < NT3; ++j) {
              const int n = j * 16 + c16;
              const bf16x8 b = __builtin_bit_cast(
                  bf16x8, *(const uint4*)(wl + (size_t)n * Hk * 2 + (((ks * 4 + q4) ^ (n & mask)) * 16)));
#pragma unroll
              for (int i = 0; i < MT; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[kk][i]), b,
                                                                    acc[i][j], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT3; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            part[((size_t)w * ROWS + i * 16 + q4 * 4 + r) * PLD + j * 16 + c16] = acc[i][j][r];
      __syncthreads();
      // (d) gate math: PPT consecutive units of one row per thread
      u32 packed[PPT / 2];
      {
        float hnew[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          float hr = 0.f, hz = 0.f, hn = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) {
            const float* pp = part + ((size_t)ww * ROWS + prow) * PLD + pu + k;
            hr += pp[0];
            hz += pp[U];
            hn += pp[2 * U];
          }
          const int uu = u0 + pu + k;
          hr += a.bhh[uu];
          hz += a.bhh[a.H + uu];
          hn += a.bhh[2 * a.H + uu];
          float r = 0.f, z = 0.f, n = 0.f, hv = 0.f;
          if (live) {
            r = 1.f / (1.f + expf(-(xr[k] + hr)));
            z = 1.f / (1.f + expf(-(xz[k] + hz)));
            n = tanhf(xn[k] + r * hn);
            hv = (1.f - z) * n + z * hp[k];
            if (a.gsave) {
              float* gs = a.gsave + (((size_t)t * a.Bpad + brow) * 4) * a.H + uu;
              gs[0] = r;
              gs[a.H] = z;
              gs[2 * a.H] = n;
              gs[3 * a.H] = hn;
            }
            a.hstate[(size_t)brow * a.H + uu] = hv;
            if (a.hall) a.hall[((size_t)(step + 1) * a.Bpad + brow) * a.H + uu] = hv;
            if (a.out) a.out[brow * a.out_sb + t * a.out_st + uu] = hv;
          }
          hnew[k] = hv;
        }
#pragma unroll
        for (int k = 0; k < PPT / 2; ++k) packed[k] = pack2bf(hnew[2 * k], hnew[2 * k + 1]);
      }
      // (e) publish h_step (write-through stores), then arrive
      {
        bf16_t* dst = a.hseq + ((size_t)(step + 1) * a.Bpad + brow) * Hk + u0 + pu;
        if constexpr (PPT == 2) {
          __hip_atomic_store((u32*)dst, packed[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
          for (int k = 0; k < PPT / 4; ++k) {
            const unsigned long long v = (unsigned long long)packed[2 * k] | ((unsigned long long)packed[2 * k + 1] << 32);
            __hip_atomic_store((unsigned long long*)(dst + 4 * k), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.counters + rg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  (void)ok;
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
  float* dbias_part;      // [NG][4][H]: sums over rows & time of dr, dz, dn, dn*r
  u32* counters;          // [NG] zeroed
  u32* err;
  int L, B, H, K3, Bpad, NG, G, S, reverse;
};

template <int UT>
__global__ __launch_bounds__(256, 1) void gru_bwd_kernel(GruBwdArgs a) {
  constexpr int U = UT * 16, PPT = MT * UT;
  constexpr int PLD = U + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, q4 = lane >> 4;
  const int g = blockIdx.x / a.S, s = blockIdx.x % a.S, u0 = s * U;
  const int K3 = a.K3, nch = K3 / 8, mask = swz_mask_for(nch), NKS = K3 / 32, H = a.H;
  char* wl = smem;                                           // [U][K3] bf16 swizzled (W_hh^T slice)
  float* part = (float*)(smem + (size_t)U * K3 * 2);         // [4][ROWS][PLD]
  float* bred = part + 4 * ROWS * PLD;                       // [ROWS][4][U] bias-grad reduction

  for (int i = tid; i < U * nch; i += 256) {
    const int n = i / nch, ch = i % nch;
    const uint4 v = *(const uint4*)(a.whhT + (size_t)(u0 + n) * K3 + ch * 8);
    *(uint4*)(wl + (size_t)n * K3 * 2 + ((ch ^ (n & mask)) * 16)) = v;
  }
  __syncthreads();

  const int prow = (tid * PPT) / U, pu = (tid * PPT) % U;
  float bsum[4][PPT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < PPT; ++k) bsum[i][k] = 0.f;

  // steps run from the last processed time step back to the first; one extra pass (step == L)
  // only folds the recurrent term of the first step into dL/dh0.
  for (int step = 0; step <= a.L; ++step) {
    const int ti = a.L - 1 - step;                       // processing index whose gradients we form
    const int t = a.reverse ? a.L - 1 - ti : ti;         // time index of that step
    for (int rg = g; rg < a.NG; rg += a.G) {
      const int row0 = rg * ROWS, brow = row0 + prow;
      const bool live = brow < a.B;
      // (a) recurrent term from the previously handled step: dgh[prev] @ W_hh  (K = 3H)
      float rec[PPT];
#pragma unroll
      for (int k = 0; k < PPT; ++k) rec[k] = 0.f;
      if (step > 0) {
        if (tid == 0) {
          wait_ge(a.counters + rg, (u32)(a.S * step), a.err);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int tprev = a.reverse ? a.L - 1 - (ti + 1) : ti + 1;
        const bf16_t* src = a.dgh + ((size_t)tprev * a.Bpad + row0) * K3;
        f32x4 acc[MT][UT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < UT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks0 = w; ks0 < NKS; ks0 += 16) {
          uint4 av[4][MT];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int ks = ks0 + kk * 4;
            if (ks < NKS) {
#pragma unroll
              for (int i = 0; i < MT; ++i)
                av[kk][i] = *(const uint4*)(src + (size_t)(i * 16 + c16) * K3 + ks * 32 + q4 * 8);
            }
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int ks = ks0 + kk * 4;
            if (ks < NKS) {
#pragma unroll
              for (int j = 0; j < UT; ++j) {
                const int n = j * 16 + c16;
                const bf16x8 b = __builtin_bit_cast(
                    bf16x8, *(const uint4*)(wl + (size_t)n * K3 * 2 + (((ks * 4 + q4) ^ (n & mask)) * 16)));
#pragma unroll
                for (int i = 0; i < MT; ++i)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[kk][i]), b,
                                                                      acc[i][j], 0, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < UT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              part[((size_t)w * ROWS + i * 16 + q4 * 4 + r) * PLD + j * 16 + c16] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PPT; ++k)
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) rec[k] += part[((size_t)ww * ROWS + prow) * PLD + pu + k];
      }
      // (b) elementwise backward of step ti
      if (step == a.L) {
        if (live) {
#pragma unroll
          for (int k = 0; k < PPT; ++k) a.dcarry[(size_t)brow * H + u0 + pu + k] += rec[k];
        }
        // the partial buffer is reused by the next row group
        __syncthreads();
        continue;
      }
      u32 pgi[3][PPT / 2], pgh[PPT / 2];
      {
        float vgi[3][PPT], vgh[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int uu = u0 + pu + k;
          float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f;
          if (live) {
            const float* gs = a.gsave + (((size_t)t * a.Bpad + brow) * 4) * H + uu;
            const float r = gs[0], z = gs[H], n = gs[2 * H], hn = gs[3 * H];
            const float hp = a.hall[((size_t)ti * a.Bpad + brow) * H + uu];
            float dh = rec[k];
            if (step == 0) {
              if (a.dhn) dh += a.dhn[(size_t)brow * H + uu];
            } else {
              dh += a.dcarry[(size_t)brow * H + uu];
            }
            if (a.dout) dh += a.dout[brow * a.dout_sb + t * a.dout_st + uu];
            dn = dh * (1.f - z) * (1.f - n * n);
            dz = dh * (hp - n) * z * (1.f - z);
            dr = dn * hn * r * (1.f - r);
            dnr = dn * r;
            a.dcarry[(size_t)brow * H + uu] = dh * z;
          }
          vgi[0][k] = dr;
          vgi[1][k] = dz;
          vgi[2][k] = dn;
          vgh[k] = dnr;
          bsum[0][k] += dr;
          bsum[1][k] += dz;
          bsum[2][k] += dn;
          bsum[3][k] += dnr;
        }
#pragma unroll
        for (int k = 0; k < PPT / 2; ++k) {
#pragma unroll
          for (int gt = 0; gt < 3; ++gt) pgi[gt][k] = pack2bf(vgi[gt][2 * k], vgi[gt][2 * k + 1]);
          pgh[k] = pack2bf(vgh[2 * k], vgh[2 * k + 1]);
        }
      }
      {
        const size_t base = ((size_t)t * a.Bpad + brow) * K3 + u0 + pu;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
          bf16_t* d1 = a.dgi + base + (size_t)gt * H;
          bf16_t* d2 = a.dgh + base + (size_t)gt * H;
          if constexpr (PPT == 2) {
            *(u32*)d1 = pgi[gt][0];
            __hip_atomic_store((u32*)d2, gt == 2 ? pgh[0] : pgi[gt][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
#pragma unroll
            for (int k = 0; k < PPT / 4; ++k) {
              const unsigned long long vi = (unsigned long long)pgi[gt][2 * k] | ((unsigned long long)pgi[gt][2 * k + 1] << 32);
              const unsigned long long vh = gt == 2 ? ((unsigned long long)pgh[2 * k] | ((unsigned long long)pgh[2 * k + 1] << 32)) : vi;
              *(unsigned long long*)(d1 + 4 * k) = vi;
              __hip_atomic_store((unsigned long long*)(d2 + 4 * k), vh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.counters + rg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // bias-gradient partials: reduce this workgroup's rows in a fixed order
  {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < PPT; ++k) bred[((size_t)prow * 4 + i) * U + pu + k] = bsum[i][k];
    __syncthreads();
    for (int i = tid; i < 4 * U; i += 256) {
      const int gt = i / U, u = i % U;
      float sacc = 0.f;
      for (int r = 0; r < ROWS; ++r) sacc += bred[((size_t)r * 4 + gt) * U + u];
      a.dbias_part[((size_t)g * 4 + gt) * H + u0 + u] = sacc;
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
  const int UT = (H % 32 == 0) ? 2 : 1, U = UT * 16, S = H / U;
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
               counters, err, L, B, H, Hk, Bpad, NG, G, S, reverse};
  const size_t smem = (size_t)3 * U * Hk * 2 + (size_t)4 * ROWS * (3 * U + 4) * 4;
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
  const int UT = (H % 32 == 0) ? 2 : 1, U = UT * 16, S = H / U;
  const int cap = max_wgs > 0 ? max_wgs : cu_count();
  PV_REQUIRE(S <= cap, "pv_gru_bwd: %d unit slices exceed the %d co-resident workgroups available", S, cap);
  const int G = NG < cap / S ? NG : cap / S;
  const size_t smem = (size_t)U * K3 * 2 + (size_t)4 * ROWS * (U + 4) * 4 + (size_t)ROWS * 4 * U * 4;
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
               (bf16_t*)dgh_bf16, dh0, dbias_part, counters, err, L, B, H, K3, Bpad, NG, G, S, reverse};
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
