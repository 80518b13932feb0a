// bf16 MFMA "NT" GEMM core for gfx950 with pluggable epilogues.
//
//   C[M,N] = A[M,K] * B[N,K]^T        A, B bf16 row-major (K contiguous), fp32 accumulate
//
// One 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile; each wave owns 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 accumulators.  A/B K-tiles (64 deep) are staged global->LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a 2-deep ring; the LDS image is lane-linear, so
// the bank-conflict XOR swizzle (16-B chunk ^= (row>>1)&7) is applied on the per-lane SOURCE address
// and again on the ds_read_b128 fragment reads.  The block id is remapped so that the 8 XCDs (one L2
// each) work on compact groups of neighbouring tiles.
//
// Epilogues (all fused, accumulators never leave registers un-reduced):
//   EpiStore     : C (+bias) -> fp32 and/or bf16                    (GRU input projections, dH, dW ...)
//   EpiSoftmax   : per-row (max, sum-exp) partials per 64-column slab + the target logit
//                  -> vocabulary softmax NLL without materialising logits   (decoders.py:191-193 +
//                  criterions.py:55-64 of the reference, fused)
//   EpiProb      : P = (softmax - onehot) * g as bf16, written both as P[T,V] and P^T[V,T] (LDS
//                  transposed), + column sums for the bias gradient          (backward of the above)
//   EpiLogProb   : dense log-softmax rows (API-compat path for callers that want [B,L,V])
#include "common.h"
#include "../../include/pevae_hip.h"
#include <mutex>

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;       // 64 KiB -> 2 workgroups per CU

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  int M, N, K, lda, ldb;
  int splits;         // split-K: every tile is computed `splits` times over disjoint k-tile ranges (1 = no split)
  int adaptive_slots; // > 0 with m_live: the kernel itself lowers `splits` to max(1, adaptive_slots / live tiles), so
                      // that a launch whose row count is only known on the device still fills the chip once
  // Device-side bounds (NULL = none), read at run time so that a captured graph follows the batch:
  const int* m_live;  // tiles whose first row is >= *m_live exit at once (their rows are ignored tokens)
  const int* k_live;  // the K loop stops at round_up(*k_live, 64) (<= K); with split-K the live k-tiles are divided
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) bf16x4* ltr_t;

// Tile walk.  The grid is a multiple of 8 workgroups; workgroups b and b+8 share an XCD (one L2), so XCD x owns the
// contiguous range [xbeg, xbeg+xcnt) of tile ids and its gridDim.x/8 workgroups stride through it.  Tile ids follow
// a grouped ordering: 8 tile-rows at a time, column-major inside the group, so tiles in flight share A/B panels.
// With a grid of round_up(ntiles, 8) every workgroup has at most one tile; a smaller grid makes the workgroups
// persistent, which is what the 15.7k-tile vocabulary launches want (workgroup dispatch alone costs ~16 ns each).
__host__ __device__ __forceinline__ int live_splits(int splits, int adaptive_slots, int ntiles) {
  if (adaptive_slots <= 0) return splits;
  const int fit = adaptive_slots / (ntiles > 0 ? ntiles : 1);
  return fit < 1 ? 1 : (fit < splits ? fit : splits);
}

__device__ __forceinline__ void tile_of(int id, int nbm, int nbn, int& pm, int& pn) {
  const int width = 8 * nbn, grp = id / width, first_m = grp * 8;
  const int gsz = min(nbm - first_m, 8);
  pm = first_m + (id % width) % gsz;
  pn = (id % width) / gsz;
}

// WM = 64-row waves per tile column: 2 -> the 128x128 tile, 4 waves, 2 LDS stages, 2 workgroups per CU;
//                                   4 -> a 256x128 tile, 8 waves, 3 LDS stages (144 KiB), 1 workgroup per CU: 85 instead of
//                                        64 flop per byte staged and two K-tiles in flight, for the K = 512 kernels.
// Every wave owns a 64x64 sub-tile either way, so the epilogues are the same.
// TN (WM = 2 only): both operands are K-MAJOR -- A [K][lda] holds A^T (element (m, k) at A[k * lda + m]), B [K][ldb]
// likewise: C = A^T B with K the leading (row) index of both, the form of a weight gradient dy^T x whose operands sit in
// memory token-major.  A stage is then two [64 k][128 x] images of 256-byte rows; the MFMA fragments (8 consecutive k of
// one row / column) come out of them through ds_read_b64_tr_b16 -- a 16-lane group reads a 4 (k) x 16 (x) block and every
// lane receives one x's four k -- so no transposed copy of either operand is ever made (round 2 spent 12 launches per
// step on them).  Swizzle: 16-byte chunk c of row r sits at c ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3)): the 8 rows a
// half-wave's transposed read touches land in 8 different 32-byte bank groups.
template <class Epi, int WM, bool TN = false>
__global__ __launch_bounds__(WM * 128, WM == 2 ? 2 : 1) void gemm_nt_kernel(GemmArgs g, Epi epi) {
  static_assert(!TN || WM == 2, "the K-major operand form is built for the 128 x 128 tile");
  constexpr int BMT = WM * 64, NW = WM * 2, NST = WM == 2 ? 2 : 3, PB = 16 / NW, STAGE = (BMT + BN) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int c16 = lane & 15, q4 = lane >> 4;
  // rows at and past *m_live carry ignored tokens: their tiles are not walked at all
  const int m_rows = g.m_live ? min(g.M, *g.m_live) : g.M;
  const int nbm = (m_rows + BMT - 1) / BMT, nbn = (g.N + BN - 1) / BN, ntiles = nbm * nbn;
  const int splits = live_splits(g.splits, g.adaptive_slots, ntiles);
  const int nitems = ntiles * splits;               // work items: (split, tile), split-major
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int tq = nitems >> 3, tr = nitems & 7;
  const int xbeg = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, xcnt = tq + (xcd < tr ? 1 : 0);

  // fragment read offsets (bytes) inside a stage: row*128 + ((ks*4+q4) ^ ((row>>1)&7))*16
  int offA[4], offB[4], swz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + c16, rb = wn * 64 + i * 16 + c16;
    offA[i] = ra * 128;
    offB[i] = BMT * BK * 2 + rb * 128;
    swz[i] = (c16 >> 1) & 7;   // (row>>1)&7 : rows are 16-aligned + c16, so only c16 matters
    if constexpr (TN) {
      // transposed read of the 4 x 16 block (k rows 8 q4 + q [+ 4], columns x0 + 16 i ..): lane (q = c16 >> 2, p = c16 & 3)
      // addresses 4 columns of row q; k-step ks adds 32 rows, the fragment's second half 4 rows
      const int q = c16 >> 2, pp = c16 & 3, sw = (q << 1) | ((q4 & 1) << 3);
      offA[i] = (8 * q4 + q) * 256 + (((wm * 8 + i * 2 + (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1);
      offB[i] = BMT * BK * 2 + (8 * q4 + q) * 256 + (((wn * 8 + i * 2 + (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1);
    }
  }
  const int nk_all = g.k_live ? min(g.K / BK, (*g.k_live + BK - 1) / BK) : g.K / BK;
  const int per_split = (nk_all + splits - 1) / splits;

  for (int tc = blockIdx.x >> 3; tc < xcnt; tc += per_xcd) {
  const int item = xbeg + tc, split = item / ntiles;
  const int kt0 = split * per_split, nk = min(nk_all, kt0 + per_split);   // an empty range stores zeros
  int pm, pn;
  tile_of(item - split * ntiles, nbm, nbn, pm, pn);
  const int m0 = pm * BMT, n0 = pn * BN;

  // --- per-lane source pointers for the 4+4 LDS-DMA pieces this wave issues per K-tile ----------
  const bf16_t* srcA[4];
  const bf16_t* srcB[PB];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (TN) {
      const int r = (w * 4 + i) * 4 + (lane >> 4);        // k row of the stage this lane fills (a piece = 4 rows of 256 B)
      const int c = (lane & 15) ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
      srcA[i] = g.A + (size_t)r * g.lda + min(m0 + c * 8, g.lda - 8);   // (columns past M feed rows past M only)
      srcB[i] = g.B + (size_t)r * g.ldb + min(n0 + c * 8, g.ldb - 8);
    } else {
      const int r = (w * 4 + i) * 8 + (lane >> 3);          // tile row this lane fills
      const int c = (lane & 7) ^ ((r >> 1) & 7);            // logical 16-B chunk stored at phys lane&7
      srcA[i] = g.A + (size_t)min(m0 + r, g.M - 1) * g.lda + c * 8;
    }
  }
  if constexpr (!TN) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int r = (w * PB + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      srcB[i] = g.B + (size_t)min(n0 + r, g.N - 1) * g.ldb + c * 8;
    }
  }
  auto stage = [&](int s, int kt) {
    char* base = smem + s * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(TN ? srcA[i] + (size_t)kt * BK * g.lda : srcA[i] + kt * BK),
                                       (lptr_t)(base + (w * 4 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(TN ? srcB[i] + (size_t)kt * BK * g.ldb : srcB[i] + kt * BK),
                                       (lptr_t)(base + BMT * BK * 2 + (w * PB + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // what the epilogue reads from global memory (bias, targets, row statistics) is requested now, a whole main loop
  // ahead of its use: at K = 512 a tile's main loop is ~4 us and an exposed load round trip in the epilogue ~1-2 us
  const typename Epi::Pre pre = epi.prefetch(m0 + wm * 64, n0 + wn * 64, lane);
  // NST-deep ring, NST-1 K-tiles requested ahead.  vmcnt retires in issue order, so "at most the 4 + PB LDS-DMA
  // instructions of the younger stage outstanding" means stage kt has landed.
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (kt0 + p < nk) stage((kt0 + p) % NST, kt0 + p);
  for (int kt = kt0; kt < nk; ++kt) {
    const int s = kt % NST;
    if (NST == 3 && kt + 1 < nk) {
      if constexpr (PB == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (kt + NST - 1 < nk) stage((kt + NST - 1) % NST, kt + NST - 1);   // the slot every wave finished reading before this barrier
    const char* base = smem + s * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (TN) {
          const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(base + offA[i] + ks * 8192));
          const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(base + offA[i] + ks * 8192 + 1024));
          const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(base + offB[i] + ks * 8192));
          const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(base + offB[i] + ks * 8192 + 1024));
          af[i] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          bfr[i] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        } else {
          const int ch = ((ks * 4 + q4) ^ swz[i]) * 16;
          af[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(base + offA[i] + ch));
          bfr[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(base + offB[i] + ch));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();   // all waves done with the ring: epilogues may reuse smem
  epi(acc, pre, m0 + wm * 64, n0 + wn * 64, w, lane, smem, split);
  __syncthreads();   // ... and the next tile's first stage may overwrite what the epilogue kept there
  }
}

// -------------------------------------------------------------------------------------------------
// epilogues.  acc[mi][ni][j] is element (row0 + mi*16 + (lane>>4)*4 + j, col0 + ni*16 + (lane&15)).
// -------------------------------------------------------------------------------------------------

struct EpiStore {
  float* C;
  bf16_t* Cb;
  const float* bias;
  int M, N, ldc, ldcb, accumulate, bias_per_row;
  long split_stride;   // split-K: slab `split` of the fp32 output starts at C + split * split_stride
  int transposed;      // fp32 only: store C^T, C[(col) * ldc + row] -- a lane's 4 accumulator rows are 16 contiguous bytes
  struct Pre {};
  __device__ Pre prefetch(int, int, int) const { return Pre{}; }
  __device__ void operator()(f32x4 (&acc)[4][4], const Pre&, int row0, int col0, int w, int lane, char* smem, int split) const {
    const int c16 = lane & 15, q4 = lane >> 4;
    float* C = this->C ? this->C + (long)split * split_stride : nullptr;
    if (transposed) {          // host guarantees M % 4 == 0, ldc % 4 == 0, 16-byte aligned C, no accumulate, bias per column
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = col0 + ni * 16 + c16;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int row = row0 + mi * 16 + q4 * 4;
          if (row < M) *(f32x4*)(C + (size_t)col * ldc + row) = acc[mi][ni] + bv;
        }
      }
      return;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = col0 + ni * 16 + c16;
      if (col >= N) continue;
      const float bv = (bias && !bias_per_row) ? bias[col] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row0 + mi * 16 + q4 * 4 + j;
          if (row >= M) continue;
          float v = acc[mi][ni][j] + bv;
          if (bias && bias_per_row) v += bias[row];
          if (C) {
            if (accumulate) v += C[(size_t)row * ldc + col];
            C[(size_t)row * ldc + col] = v;
          }
          if (Cb) Cb[(size_t)row * ldcb + col] = f2bf(v);
        }
    }
  }
};

// reductions over the 16 lanes that share lane >> 4, on the DPP cross-lane path (full-rate VALU; a ds_bpermute shuffle
// is a ~100-cycle LDS round trip, and 128 of them in a dependent chain made this epilogue 40 % of the kernel)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov<0x141>(v));     // row_half_mirror
  return fmaxf(v, dpp_mov<0x140>(v));  // row_mirror
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  return v + dpp_mov<0x140>(v);
}

struct EpiSoftmax {
  const float* bias;        // [V]
  const int64_t* target;    // [T]
  float2* part;             // [nslab][Tpad]  (max, sumexp) of the scaled logits of a 64-column slab
  float* tgt_logit;         // [T]
  int T, V, Tpad;
  float inv_temp;
  struct Pre {
    float bv[4];
    int tgt[16];
  };
  __device__ Pre prefetch(int row0, int col0, int lane) const {
    Pre p;
    const int c16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) p.bv[ni] = bias[min(col0 + ni * 16 + c16, V - 1)];
#pragma unroll
    for (int r = 0; r < 16; ++r) p.tgt[r] = (int)target[min(row0 + (r >> 2) * 16 + q4 * 4 + (r & 3), T - 1)];
    return p;
  }
  __device__ void operator()(f32x4 (&acc)[4][4], const Pre& pre, int row0, int col0, int w, int lane, char* smem, int split) const {
    if (col0 >= V) return;
    const int c16 = lane & 15, q4 = lane >> 4;
    float bv[4];
    bool live[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = col0 + ni * 16 + c16;
      live[ni] = col < V;
      bv[ni] = live[ni] ? pre.bv[ni] : 0.f;
    }
    // the 16 rows of this lane in two halves of 8: 8 independent reduction chains each
    float my_m = 0.f, my_s = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int tc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = row0 + (2 * h + (r >> 2)) * 16 + q4 * 4 + (r & 3);
        tc[r] = row < T ? pre.tgt[8 * h + r] - col0 : -1;
      }
      float m[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int mi = 2 * h + (r >> 2), j = r & 3;
        float x[4];
        float mx = -INFINITY;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          x[ni] = live[ni] ? (acc[mi][ni][j] + bv[ni]) * inv_temp : -INFINITY;
          mx = fmaxf(mx, x[ni]);
        }
        if (tc[r] >= 0 && tc[r] < 64 && (tc[r] & 15) == c16) {
          const int tn = tc[r] >> 4;
          tgt_logit[row0 + mi * 16 + q4 * 4 + j] = tn == 0 ? x[0] : tn == 1 ? x[1] : tn == 2 ? x[2] : x[3];
        }
        m[r] = mx;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) m[r] = row16_max(m[r]);
      float sm[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int mi = 2 * h + (r >> 2), j = r & 3;
        float t = 0.f;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          t += live[ni] ? __expf((acc[mi][ni][j] + bv[ni]) * inv_temp - m[r]) : 0.f;
        sm[r] = t;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) sm[r] = row16_sum(sm[r]);
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (c16 == 8 * h + r) {
          my_m = m[r];
          my_s = sm[r];
        }
    }
    // lane (q4, c16) publishes row  (c16>>2)*16 + q4*4 + (c16&3)
    const int row = row0 + (c16 >> 2) * 16 + q4 * 4 + (c16 & 3);
    if (row < Tpad) part[(size_t)(col0 >> 6) * Tpad + row] = make_float2(my_m, my_s);
  }
};

// Greedy decoding (decoders.py:252-254: scores.topk(1)): per row and 64-column slab the maximum scaled logit, its column
// and sum exp(x - max): arg-max and the log-probability of the chosen symbol without a [B, V] tensor.  `mask_col`
// (decoders.py:144-147,247-248: mask_unk) is excluded from the arg-max but stays in the softmax denominator, as adding
// a -inf one-hot to LOG-PROBABILITIES does in the reference.
struct ArgPart {
  float m, s, best;     // slab max over ALL columns, sum exp(x - m), best value over the allowed columns
  int idx;              // its column (lowest on ties), -1 if none
};

template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

struct EpiArgmax {
  const float* bias;
  ArgPart* part;            // [nslab][Tpad]
  int T, V, Tpad, mask_col;
  float inv_temp;
  struct Pre {
    float bv[4];
  };
  __device__ Pre prefetch(int row0, int col0, int lane) const {
    Pre p;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) p.bv[ni] = bias[min(col0 + ni * 16 + (lane & 15), V - 1)];
    return p;
  }
  __device__ void operator()(f32x4 (&acc)[4][4], const Pre& pre, int row0, int col0, int w, int lane, char* smem, int split) const {
    if (col0 >= V) return;
    const int c16 = lane & 15, q4 = lane >> 4;
    ArgPart mine{0.f, 0.f, 0.f, -1};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mi = r >> 2, j = r & 3;
      float x[4], mx = -INFINITY, bst = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = col0 + ni * 16 + c16;
        x[ni] = col < V ? (acc[mi][ni][j] + pre.bv[ni]) * inv_temp : -INFINITY;
        mx = fmaxf(mx, x[ni]);
        if (col < V && col != mask_col && x[ni] > bst) {      // ascending columns: the first maximum wins
          bst = x[ni];
          bi = col;
        }
      }
      const float m = row16_max(mx), b = row16_max(bst);
      float sacc = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) sacc += __expf(x[ni] - m);
      sacc = row16_sum(sacc);
      int cand = (bst == b && b > -INFINITY) ? bi : 0x7fffffff;   // lowest column among the lanes that hold the maximum
      cand = min(cand, dpp_movi<0xB1>(cand));
      cand = min(cand, dpp_movi<0x4E>(cand));
      cand = min(cand, dpp_movi<0x141>(cand));
      cand = min(cand, dpp_movi<0x140>(cand));
      if (c16 == r) mine = ArgPart{m, sacc, b, cand == 0x7fffffff ? -1 : cand};
    }
    const int row = row0 + (c16 >> 2) * 16 + q4 * 4 + (c16 & 3);
    if (row < Tpad) part[(size_t)(col0 >> 6) * Tpad + row] = mine;
  }
};

// `sample` decoding (decoders.py:255-256 -> base_modules.py:54-59, torch.multinomial(exp(log-probs), 1)) without a [B, V]
// tensor: one draw from softmax(x) IS arg-max_v (x_v - log q_v) with q_v ~ Exp(1) (the sampler torch uses for one draw
// per row: arg-max of p / q), so the sampling step is the greedy epilogue on perturbed logits.  q comes from `noise`
// ([T][noise_ld], parity tests: the same draws go to the dense path / the reference) or from Philox keyed by (seed,
// step, site, row, column).  The record's `best` is the perturbed value (compared across slabs); the unperturbed
// logit of that column goes to `xbest` for the emitted symbol's log-probability.
struct EpiSample {
  const float* bias;
  ArgPart* part;            // [nslab][Tpad]
  float* xbest;             // [nslab][Tpad]
  const float* noise;
  long noise_ld;
  const RngState* rng;
  uint32_t site;
  int T, V, Tpad, mask_col;
  float inv_temp;
  struct Pre {
    float bv[4];
  };
  __device__ Pre prefetch(int row0, int col0, int lane) const {
    Pre p;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) p.bv[ni] = bias[min(col0 + ni * 16 + (lane & 15), V - 1)];
    return p;
  }
  __device__ void operator()(f32x4 (&acc)[4][4], const Pre& pre, int row0, int col0, int w, int lane, char* smem, int split) const {
    if (col0 >= V) return;
    const int c16 = lane & 15, q4 = lane >> 4;
    ArgPart mine{0.f, 0.f, 0.f, -1};
    float mine_x = 0.f;
    RngState st{0, 0};
    if (!noise) st = *rng;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mi = r >> 2, j = r & 3, row = row0 + mi * 16 + q4 * 4 + j;
      float x[4], mx = -INFINITY, bst = -INFINITY, bx = 0.f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = col0 + ni * 16 + c16;
        x[ni] = col < V ? (acc[mi][ni][j] + pre.bv[ni]) * inv_temp : -INFINITY;
        mx = fmaxf(mx, x[ni]);
        if (col < V && col != mask_col && row < T) {
          float q;
          if (noise) {
            q = noise[(size_t)row * noise_ld + col];
          } else {
            uint32_t rnd[4];
            philox4x32_10((uint32_t)col, (uint32_t)row, site, (uint32_t)st.step ^ (uint32_t)(st.step >> 32) * 0x9E3779B9u,
                          (uint32_t)st.seed, (uint32_t)(st.seed >> 32), rnd);
            q = -__logf(((float)(rnd[0] >> 8) + 0.5f) * (1.0f / 16777216.0f));     // Exp(1), u in (0, 1)
          }
          const float xs = x[ni] - __logf(q);
          if (xs > bst) {                                      // ascending columns: the first maximum wins
            bst = xs;
            bx = x[ni];
            bi = col;
          }
        }
      }
      const float m = row16_max(mx), b = row16_max(bst);
      float sacc = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) sacc += __expf(x[ni] - m);
      sacc = row16_sum(sacc);
      int cand = (bst == b && b > -INFINITY) ? bi : 0x7fffffff;   // lowest column among the lanes that hold the maximum
      cand = min(cand, dpp_movi<0xB1>(cand));
      cand = min(cand, dpp_movi<0x4E>(cand));
      cand = min(cand, dpp_movi<0x141>(cand));
      cand = min(cand, dpp_movi<0x140>(cand));
      const float xw = row16_max((bi == cand && cand != 0x7fffffff) ? bx : -INFINITY);
      if (c16 == r) {
        mine = ArgPart{m, sacc, b, cand == 0x7fffffff ? -1 : cand};
        mine_x = xw;
      }
    }
    const int row = row0 + (c16 >> 2) * 16 + q4 * 4 + (c16 & 3);
    if (row < Tpad) {
      part[(size_t)(col0 >> 6) * Tpad + row] = mine;
      xbest[(size_t)(col0 >> 6) * Tpad + row] = mine_x;
    }
  }
};

// 16 threads per row, 16 rows per workgroup: thread (row r, chunk c) folds slabs c, c+16, ... in ascending order, then
// the 16 chunk results of a row are folded in chunk order through LDS.  Fixed order (run-to-run identical); among equal
// maxima the smallest column wins, as a left-to-right scan would choose.  (One thread per row walking all 313 slabs of
// the Yelp vocabulary in a dependent chain took 89 us per decoded token -- a third of the greedy decode.)
// xbest (sampling): the chosen column's unperturbed logit per record; NULL (greedy): the record's `best` is that logit.
__global__ __launch_bounds__(256) void vocab_argmax_reduce_kernel(const ArgPart* __restrict__ part, const float* __restrict__ xbest,
                                                                   int T, int Tpad, int nslab, int64_t* __restrict__ ids,
                                                                   float* __restrict__ logp) {
  __shared__ float sm_m[256], sm_s[256], sm_b[256], sm_x[256];
  __shared__ int sm_i[256];
  const int r = threadIdx.x & 15, c = threadIdx.x >> 4;
  const int row = min(blockIdx.x * 16 + r, Tpad - 1);
  float m = -INFINITY, s = 0.f, best = -INFINITY, xb = 0.f;
  int idx = -1;
  for (int k = c; k < nslab; k += 16) {
    const ArgPart p = part[(size_t)k * Tpad + row];
    const float mn = fmaxf(m, p.m);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (p.m == -INFINITY ? 0.f : p.s * __expf(p.m - mn));
    m = mn;
    if (p.idx >= 0 && (p.best > best || (p.best == best && (idx < 0 || p.idx < idx)))) {
      best = p.best;
      idx = p.idx;
      xb = xbest ? xbest[(size_t)k * Tpad + row] : p.best;
    }
  }
  sm_m[threadIdx.x] = m;
  sm_s[threadIdx.x] = s;
  sm_b[threadIdx.x] = best;
  sm_x[threadIdx.x] = xb;
  sm_i[threadIdx.x] = idx;
  __syncthreads();
  if (c != 0 || blockIdx.x * 16 + r >= T) return;
  for (int k = 1; k < 16; ++k) {
    const float pm = sm_m[k * 16 + r], ps = sm_s[k * 16 + r], pb = sm_b[k * 16 + r];
    const int pi = sm_i[k * 16 + r];
    const float mn = fmaxf(m, pm);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (pm == -INFINITY ? 0.f : ps * __expf(pm - mn));
    m = mn;
    if (pi >= 0 && (pb > best || (pb == best && (idx < 0 || pi < idx)))) {
      best = pb;
      idx = pi;
      xb = sm_x[k * 16 + r];
    }
  }
  ids[blockIdx.x * 16 + r] = idx;
  logp[blockIdx.x * 16 + r] = xb - (m + logf(s));
}

constexpr int EPI_LD = 144;            // bytes per staged 64-element bf16 row (128 + 16 pad)
constexpr int EPI_WAVE_BYTES = 64 * EPI_LD;

struct EpiProb {
  const float* bias;        // [V]
  const int64_t* target;    // [T]
  const float* lse;         // [T]
  const float* gscale;      // [T]  dLoss/dnll_tok (0 for ignored tokens); inv_temp applied here
  bf16_t* P;                // [Tpad][ldp]   ldp  >= round_up(V,64)
  bf16_t* PT;               // [Vpad][ldpt]  ldpt >= Tpad = round_up(T,64)
  float* dbias_part;        // [Tpad/64][ldp]
  int T, V, Tpad, Vpad, ldp, ldpt;
  float inv_temp;
  struct Pre {
    float bv[4];
    int tgt[16];
    float lse[16], gs[16];
  };
  __device__ Pre prefetch(int row0, int col0, int lane) const {
    Pre p;
    const int c16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) p.bv[ni] = bias[min(col0 + ni * 16 + c16, V - 1)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(row0 + (r >> 2) * 16 + q4 * 4 + (r & 3), T - 1);
      p.tgt[r] = (int)target[row];
      p.lse[r] = lse[row];
      p.gs[r] = gscale[row];
    }
    return p;
  }
  __device__ void operator()(f32x4 (&acc)[4][4], const Pre& pre, int row0, int col0, int w, int lane, char* smem, int split) const {
    const int c16 = lane & 15, q4 = lane >> 4;
    const bool slab_live = (col0 < Vpad) && (row0 < Tpad);
    char* my = smem + w * EPI_WAVE_BYTES;
    float bv[4];
    bool live[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = col0 + ni * 16 + c16;
      live[ni] = col < V;
      bv[ni] = live[ni] ? pre.bv[ni] : 0.f;
    }
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = row0 + mi * 16 + q4 * 4 + j;
        const bool rl = row < T;
        const float l = rl ? pre.lse[mi * 4 + j] : 0.f;
        const float gs = rl ? pre.gs[mi * 4 + j] * inv_temp : 0.f;
        const int tc = rl ? pre.tgt[mi * 4 + j] - col0 : -1;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float p = 0.f;
          if (live[ni] && gs != 0.f) {
            p = __expf((acc[mi][ni][j] + bv[ni]) * inv_temp - l);
            if (tc == ni * 16 + c16) p -= 1.f;
            p *= gs;
          }
          acc[mi][ni][j] = p;
          colsum[ni] += p;
        }
      }
    if (!slab_live) {     // uniform per wave; the barriers below are workgroup-wide, keep them reachable
      __syncthreads();
      __syncthreads();
      __syncthreads();
      return;
    }
    // bias-gradient partials: reduce the 4 row-groups (lanes l, l^16, l^32, l^48)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float s = colsum[ni];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (q4 == 0) dbias_part[(size_t)(row0 >> 6) * ldp + col0 + ni * 16 + c16] = s;
    }
    // ---- P tile, row-major image [64 rows][64 cols] ------------------------------------------
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(bf16_t*)(my + (mi * 16 + q4 * 4 + j) * EPI_LD + (ni * 16 + c16) * 2) = f2bf(acc[mi][ni][j]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 8 + (lane >> 3), ch = lane & 7;
      const uint4 v = *(const uint4*)(my + r * EPI_LD + ch * 16);
      *(uint4*)(P + (size_t)(row0 + r) * ldp + col0 + ch * 8) = v;
    }
    __syncthreads();
    // ---- P^T tile: image [64 cols][64 rows]; a lane's 4 j-values are 4 consecutive rows = 8 bytes
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        uint2 v;
        v.x = pack2bf(acc[mi][ni][0], acc[mi][ni][1]);
        v.y = pack2bf(acc[mi][ni][2], acc[mi][ni][3]);
        *(uint2*)(my + (ni * 16 + c16) * EPI_LD + (mi * 16 + q4 * 4) * 2) = v;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 8 + (lane >> 3), ch = lane & 7;
      const uint4 v = *(const uint4*)(my + r * EPI_LD + ch * 16);
      *(uint4*)(PT + (size_t)(col0 + r) * ldpt + row0 + ch * 8) = v;
    }
  }
};

struct EpiLogProb {
  const float* bias;
  const float* lse;   // [T]
  float* out;         // [T][V] dense log-probs
  int T, V;
  float inv_temp;
  struct Pre {};
  __device__ Pre prefetch(int, int, int) const { return Pre{}; }
  __device__ void operator()(f32x4 (&acc)[4][4], const Pre&, int row0, int col0, int w, int lane, char* smem, int split) const {
    const int c16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = col0 + ni * 16 + c16;
      if (col >= V) continue;
      const float bv = bias[col];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row0 + mi * 16 + q4 * 4 + j;
          if (row < T) out[(size_t)row * V + col] = (acc[mi][ni][j] + bv) * inv_temp - lse[row];
        }
    }
  }
};

// combine the per-slab (max, sumexp) partials: 16 rows per block, 16 slab-strided lanes per row (a lane's dependent
// chain is nslab/16 = 20 links at the Yelp vocabulary), then the 16 lane results of a row in lane order through LDS
__global__ __launch_bounds__(256) void vocab_reduce_kernel(const float2* __restrict__ part,
                                                            const float* __restrict__ tgt_logit,
                                                            const int64_t* __restrict__ target, int T,
                                                            int Tpad, int nslab, int64_t ignore_index,
                                                            const int* __restrict__ live_rows,
                                                            float* __restrict__ lse,
                                                            float* __restrict__ nll) {
  __shared__ float sm[16][16], ss[16][16];
  const int r = threadIdx.x & 15, y = threadIdx.x >> 4;
  const int row = blockIdx.x * 16 + r;
  if (live_rows) {                                   // tiles past the live rows were never computed
    const int lim = (*live_rows + 127) / 128 * 128;
    if ((int)blockIdx.x * 16 >= lim) {
      if (y == 0 && row < T) {
        lse[row] = 0.f;
        nll[row] = 0.f;
      }
      return;
    }
  }
  float m = -INFINITY, s = 0.f;
  if (row < T) {
    for (int k = y; k < nslab; k += 16) {
      const float2 p = part[(size_t)k * Tpad + row];
      const float mn = fmaxf(m, p.x);
      s = s * __expf(m - mn) + p.y * __expf(p.x - mn);
      m = mn;
    }
  }
  sm[y][r] = m;
  ss[y][r] = s;
  __syncthreads();
  if (y == 0 && row < T) {
    float M = sm[0][r];
#pragma unroll
    for (int k = 1; k < 16; ++k) M = fmaxf(M, sm[k][r]);
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) S += sm[k][r] == -INFINITY ? 0.f : ss[k][r] * __expf(sm[k][r] - M);
    const float l = M + logf(S);
    lse[row] = l;
    const int64_t t = target[row];
    nll[row] = (t == ignore_index) ? 0.f : l - tgt_logit[row];
  }
}

// ---- ignored-token compaction ------------------------------------------------------------------------------------
// Rows whose target is the ignore index (padding: 44 % of the 12 800 token rows of the bench batch) contribute exactly
// zero to the loss (NLLLoss(ignore_index), criterions.py:38-41) and to every gradient.  A stable partition moves the
// live rows to the front; the GEMMs take the live count from device memory (GemmArgs::m_live / k_live), so the launch
// geometry is static (graph-capturable) while the work follows the batch.
// perm[new] = old row; live rows first in their original order, then the ignored rows; count[0] = live rows.
__global__ __launch_bounds__(1024) void vocab_compact_kernel(const int64_t* __restrict__ target, int T, int64_t ignore,
                                                              int64_t* __restrict__ perm, int* __restrict__ count) {
  // one workgroup of 16 waves; position t = base + it*1024 + tid, so every load is coalesced and a wave's 64 flags are
  // one ballot.  Live rank = live tokens before the chunk + (iteration, wave) prefix + popcount of the lower lanes.
  __shared__ int red[16];
  __shared__ int wc[256 + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int c = 0;
  for (int t = tid; t < T; t += 1024) c += target[t] != ignore;
  for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) red[w] = c;
  __syncthreads();
  int total = 0;
  for (int i = 0; i < 16; ++i) total += red[i];
  int live_before = 0;
  for (int base = 0; base < T; base += 16 * 1024) {
    unsigned long long m[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int t = base + it * 1024 + tid;
      m[it] = __ballot(t < T && target[t] != ignore);
      if (lane == 0) wc[it * 16 + w] = __builtin_popcountll(m[it]);
    }
    __syncthreads();
    if (w == 0) {                                   // exclusive scan of the 256 (iteration, wave) counts
      int v[4], sum = 0;
      for (int j = 0; j < 4; ++j) { v[j] = wc[lane * 4 + j]; sum += v[j]; }
      int incl = sum;
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      int run = incl - sum;
      for (int j = 0; j < 4; ++j) { wc[lane * 4 + j] = run; run += v[j]; }
      if (lane == 63) wc[256] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int t = base + it * 1024 + tid;
      if (t < T) {
        const int rank = live_before + wc[it * 16 + w] + __builtin_popcountll(m[it] & ((1ull << lane) - 1));
        if ((m[it] >> lane) & 1) perm[rank] = t;
        else perm[total + (t - rank)] = t;          // dead tokens follow the live ones, in order too
      }
    }
    live_before += wc[256];
    __syncthreads();
  }
  if (tid == 0) count[0] = total;
}

// h_c[new][:] = h[perm[new]][:] (bf16 rows, 16-byte chunks), target_c[new] = target[perm[new]]
__global__ void vocab_gather_kernel(const bf16_t* __restrict__ h, int ldh, const int64_t* __restrict__ target,
                                    const int64_t* __restrict__ perm, int T, int chunks, bf16_t* __restrict__ hc, int ldc,
                                    int64_t* __restrict__ target_c) {
  const long n = (long)T * chunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / chunks), ch = (int)(i % chunks);
    const long src = perm[row];
    *(uint4*)(hc + (size_t)row * ldc + ch * 8) = *(const uint4*)(h + (size_t)src * ldh + ch * 8);
    if (ch == 0) target_c[row] = target[src];
  }
}

// dst[perm[new]][:] = new < count ? src[new][:] : 0     (fp32 rows of `cols` floats)
__global__ void scatter_rows_kernel(const float* __restrict__ src, long lds_, const int64_t* __restrict__ perm,
                                    const int* __restrict__ count, int T, int cols, float* __restrict__ dst, long ldd) {
  const long n = (long)T * cols;
  const int live = *count;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cols), c = (int)(i % cols);
    dst[perm[row] * ldd + c] = row < live ? src[(size_t)row * lds_ + c] : 0.f;
  }
}

int check_gemm(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const char* who) {
  PV_REQUIRE(A && B, "%s: null operand", who);
  PV_REQUIRE(M > 0 && N > 0 && K > 0, "%s: empty problem M=%d N=%d K=%d", who, M, N, K);
  PV_REQUIRE(K % BK == 0, "%s: K=%d must be a multiple of %d (pad with zeros)", who, K, BK);
  PV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
             "%s: lda=%d ldb=%d must be multiples of 8 and >= K=%d", who, lda, ldb, K);
  PV_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "%s: operands must be 16-byte aligned", who);
  return PV_OK;
}

// C[m][n] (+)= sum over splits of ws[s][m][n]   (fixed order: bitwise reproducible)
// With m_live the rows past round_up(*m_live, 128) were never computed and are left alone, and the split count is
// the one the GEMM kernel chose for that row count.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long slab, int M,
                                                             int N, float* __restrict__ C, int ldc, int accumulate,
                                                             const int* __restrict__ m_live, int adaptive_slots) {
  if (m_live) {
    const int nbm = (min(M, *m_live) + BM - 1) / BM;
    M = min(M, nbm * BM);
    splits = live_splits(splits, adaptive_slots, nbm * ((N + BN - 1) / BN));
  }
  const long total = (long)M * N;
  if ((N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0) {       // 16-byte path: four columns per thread
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    f32x4 s = *(const f32x4*)(ws + i);
    for (int k = 1; k < splits; ++k) s += *(const f32x4*)(ws + k * slab + i);
    f32x4* dst = (f32x4*)(C + (i / N) * ldc + i % N);
    *dst = accumulate ? *dst + s : s;
    return;
  }
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4, e = min(total, i + 4); i < e; ++i) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[k * slab + i];
    float* dst = C + (i / N) * ldc + i % N;
    *dst = accumulate ? *dst + s : s;
  }
}

// Beam search's step without a [rows, V] matrix (decoders.py:263-270: topk(beam) over cum_sum + log-softmax): the k best
// symbols of every row and their log-probabilities.  The arg-max epilogue above already leaves, per 64-column slab, the
// slab's best allowed value: the k best entries of a row lie inside the k slabs with the largest such values (any other slab
// is dominated by k slabs that each hold an entry at least as large).  One workgroup per row: the log-sum-exp from the slab
// partials, the k best slabs, their 64 k logits recomputed from the row's state (one wave per column: a coalesced 16-byte
// read of the weight row per lane, fp32 accumulation), then the k best of those, lowest column first on ties.
constexpr int TOPK_MAX = 16;
__global__ __launch_bounds__(256) void vocab_topk_kernel(const ArgPart* __restrict__ part, int Tpad, int nslab,
                                                         const bf16_t* __restrict__ h, int ldh, const bf16_t* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias, int V, int H, float inv_temp, int mask_col,
                                                         int k, int64_t* __restrict__ ids, float* __restrict__ logp) {
  extern __shared__ float tk_sm[];                    // [nslab] slab bests, then [H] the row's state, then [k * 64] candidates
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ float red_m[4], red_s[4];
  __shared__ int sel[TOPK_MAX];
  float* sbest = tk_sm;
  float* hrow = tk_sm + nslab;
  float* cand = hrow + H;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // ---- log-sum-exp over the slabs; the slabs' best allowed values -------------------------------------------------------
  float m = -INFINITY, sacc = 0.f;
  for (int sl = tid; sl < nslab; sl += 256) {
    const ArgPart p = part[(size_t)sl * Tpad + row];
    const float mn = fmaxf(m, p.m);
    sacc = (m == -INFINITY ? 0.f : sacc * __expf(m - mn)) + (p.m == -INFINITY ? 0.f : p.s * __expf(p.m - mn));
    m = mn;
    sbest[sl] = p.idx >= 0 ? p.best : -INFINITY;
  }
  for (int c = tid; c < H; c += 256) hrow[c] = bf2f(h[(size_t)row * ldh + c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(sacc, o);
    const float mn = fmaxf(m, m2);
    sacc = (m == -INFINITY ? 0.f : sacc * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
  }
  if (lane == 0) { red_m[w] = m; red_s[w] = sacc; }
  __syncthreads();
  float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3])), S = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) S += red_m[i] == -INFINITY ? 0.f : red_s[i] * __expf(red_m[i] - M);
  const float lse = M + __logf(S);
  // block-wide arg-max over arr[0, n): largest value, lowest index on ties; the winner is struck out
  auto pick = [&](float* arr, int n, float& best_v) -> int {
    float v = -INFINITY;
    int at = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
      const float x = arr[i];
      if (x > v) { v = x; at = i; }                  // ascending i: the first maximum wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(v, o);
      const int a2 = __shfl_xor(at, o);
      if (v2 > v || (v2 == v && a2 < at)) { v = v2; at = a2; }
    }
    __syncthreads();                                   // (red_* of the round before have been read)
    if (lane == 0) { red_v[w] = v; red_i[w] = at; }
    __syncthreads();
    v = red_v[0];
    at = red_i[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (red_v[i] > v || (red_v[i] == v && red_i[i] < at)) { v = red_v[i]; at = red_i[i]; }
    best_v = v;
    if (tid == 0 && at != 0x7fffffff) arr[at] = -INFINITY;
    __syncthreads();
    return at;
  };
  const int ks = min(k, nslab);
  for (int j = 0; j < ks; ++j) {
    float bv;
    const int at = pick(sbest, nslab, bv);
    if (tid == 0) sel[j] = (at == 0x7fffffff || bv == -INFINITY) ? -1 : at;
  }
  __syncthreads();
  // ---- the candidates: 64 columns of each chosen slab, one wave per column -------------------------------------------------
  for (int c = w; c < ks * 64; c += 4) {
    const int sl = sel[c >> 6], col = sl * 64 + (c & 63);
    float x = -INFINITY;
    if (sl >= 0 && col < V && col != mask_col) {       // (wave-uniform)
      const bf16_t* wr = W + (size_t)col * ldw;
      float acc = 0.f;
      for (int p8 = lane * 8; p8 < H; p8 += 512) {
        const bf16x8 wv = *(const bf16x8*)(wr + p8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)wv[e], hrow[p8 + e], acc);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      x = (acc + bias[col]) * inv_temp;
    }
    if (lane == 0) cand[c] = x;
  }
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    float bv;
    const int at = pick(cand, ks * 64, bv);
    if (tid == 0) {
      const bool ok = at != 0x7fffffff && bv > -INFINITY;
      ids[(size_t)row * k + j] = ok ? (int64_t)sel[at >> 6] * 64 + (at & 63) : 0;
      logp[(size_t)row * k + j] = ok ? bv - lse : -INFINITY;
    }
  }
}

// ---- scratch of pv_vocab_argmax_noalloc: one array per GPU, held until the process ends ----------------------------
constexpr size_t ARGMAX_SCRATCH_BYTES = 4u << 20;      // e.g. 256 rows x 65 536 symbols, or 1 024 rows x 16 384
static __device__ __align__(256) unsigned char scratch[ARGMAX_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
static void* scratch_dev_ptr[MAX_DEVICES];             // device address of `scratch` on each GPU (0 = not looked up yet)
static std::mutex scratch_mutex;

int argmax_scratch_for_current_device(void** out) {
  int dev = 0;
  PV_HIP(hipGetDevice(&dev));
  PV_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "pv_vocab_argmax_noalloc: device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lock(scratch_mutex);
  if (!scratch_dev_ptr[dev]) {
    void* p = nullptr;
    PV_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));       // with `dev` current: that GPU's copy
    PV_REQUIRE(p && ((uintptr_t)p & 255) == 0, "pv_vocab_argmax_noalloc: bad scratch address %p", p);
    scratch_dev_ptr[dev] = p;
  }
  *out = scratch_dev_ptr[dev];
  return PV_OK;
}

int persistent_grid() {
  static int grid = 0;
  if (!grid) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (const char* e = getenv("PEVAE_GEMM_WGS_PER_CU")) grid = (cus * atoi(e) + 7) & ~7;
    else grid = (cus * 2 + 7) & ~7;
  }
  return grid;
}

template <class Epi, int WM = 2, bool TN = false>
int launch(const GemmArgs& g, const Epi& epi, hipStream_t st, const char* who) {
  constexpr int BMT = WM * 64, LDS = (WM == 2 ? 2 : 3) * (BMT + BN) * BK * 2;
  static bool attr_done = false;
  if (!attr_done) {
    PV_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<Epi, WM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done = true;
  }
  // one work item per workgroup while that fills the chip about once; beyond it the resident workgroups (two per CU
  // for the 128-row tile, one for the 256-row tile) walk the items
  const int gx = std::min((cdiv(g.M, BMT) * cdiv(g.N, BN) * g.splits + 7) & ~7, persistent_grid() * 2 / WM);
  hipLaunchKernelGGL((gemm_nt_kernel<Epi, WM, TN>), dim3(gx), dim3(WM * 128), LDS, st, g, epi);
  PV_LAUNCH_CHECK(who);
  return PV_OK;
}

// The 256-row tile for the two K = 512 vocabulary kernels is an experiment, off unless PEVAE_GEMM_BIG_TILE=1: measured
// SLOWER than the 128-row tile (forward 464 vs 400 us, probabilities 579 vs 483 us at T = 12800) although it stages a
// third fewer bytes per flop and keeps two K-tiles in flight -- with one 8-wave workgroup per CU nothing overlaps a
// tile's epilogue, whereas two 4-wave workgroups per CU run one's epilogue under the other's main loop.
bool big_tile(int M, int N, int K) {
  const char* e = getenv("PEVAE_GEMM_BIG_TILE");
  if (!e || atoi(e) != 1) return false;
  return K <= 1024 && (long)cdiv(M, 256) * cdiv(N, BN) >= 4L * persistent_grid();
}

}  // namespace

extern "C" {

int pv_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                    const float* bias, int bias_per_row, float* C, int ldc, void* Cb, int ldcb, int accumulate,
                    const int* m_live, const int* k_live, void* stream) {
  int rc = check_gemm(A, B, M, N, K, lda, ldb, "pv_gemm_bf16_nt");
  if (rc) return rc;
  PV_REQUIRE(C || Cb, "pv_gemm_bf16_nt: no output given");
  PV_REQUIRE((!C || ldc >= N) && (!Cb || ldcb >= N), "pv_gemm_bf16_nt: output leading dimension < N");
  GemmArgs g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, 1, 0, m_live, k_live};
  PV_REQUIRE(!accumulate || C, "pv_gemm_bf16_nt: accumulate needs the fp32 output");
  EpiStore e{C, (bf16_t*)Cb, bias, M, N, ldc, ldcb, accumulate, bias_per_row, 0, 0};
  return launch(g, e, (hipStream_t)stream, "pv_gemm_bf16_nt");
}

/* C^T (+bias per column of C) in fp32: Ct[n * ldct + m] = sum_k A[m][k] B[n][k] + bias[n].  The accumulator layout gives
 * every lane four consecutive m of one n, so the transposed image leaves as 16-byte stores (the row-major epilogue
 * stores 4 bytes per lane).  Used for the recurrence's input projections, which the kernels read unit-major:
 * xprojT [3H][tokens] = (x W_ih^T + b_ih)^T with A = x, B = W_ih. */
int pv_gemm_bf16_nt_ct(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias, float* Ct,
                       int ldct, void* stream) {
  int rc = check_gemm(A, B, M, N, K, lda, ldb, "pv_gemm_bf16_nt_ct");
  if (rc) return rc;
  PV_REQUIRE(Ct && M % 4 == 0 && ldct % 4 == 0 && ldct >= M && ((uintptr_t)Ct & 15) == 0,
             "pv_gemm_bf16_nt_ct: M and ldct must be multiples of 4, ldct >= M, Ct 16-byte aligned");
  GemmArgs g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, 1, 0, nullptr, nullptr};
  EpiStore e{Ct, nullptr, bias, M, N, ldct, 0, 0, 0, 0, 1};
  return launch(g, e, (hipStream_t)stream, "pv_gemm_bf16_nt_ct");
}

#ifdef PV_GEMM_DEBUG
// diagnostic build only: the GEMM main loop with a (nearly) empty epilogue, to time epilogues by difference
struct EpiNone {
  float* sink;
  struct Pre {};
  __device__ Pre prefetch(int, int, int) const { return Pre{}; }
  __device__ void operator()(f32x4 (&acc)[4][4], const Pre&, int row0, int col0, int w, int lane, char* smem, int split) const {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 1234.5678f) sink[0] = s;
  }
};
}
extern "C" int pv_gemm_bf16_nt_noepi(const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* sink, void* stream) {
  GemmArgs g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, 1, 0, nullptr, nullptr};
  EpiNone e{sink};
  return launch(g, e, (hipStream_t)stream, "noepi");
}
extern "C" {
#endif

int pv_gemm_splitk_plan(int M, int N, int K, int target_wgs, int* splits) {
  PV_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && splits, "pv_gemm_splitk_plan: bad argument");
  const int tiles = cdiv(M, BM) * cdiv(N, BN), nk = K / BK;
  int s = target_wgs / tiles;
  if (s < 1) {
    // More tiles than resident workgroups: the launch runs in rounds and the last one may be nearly empty (628 tiles on
    // 512 slots = 2 rounds at 61 %).  A small split evens the rounds out; it pays only when the gain clearly exceeds the
    // slab traffic (measured at M=20004, N=512, K=12800: 523 -> 410 us with 3 splits; M=12800: no split is best).
    s = 1;
    const double e1 = (double)tiles / target_wgs / cdiv(tiles, target_wgs);
    for (int c = 2; c <= 4; ++c) {
      const double ec = (double)tiles * c / target_wgs / cdiv(tiles * c, target_wgs);
      if (ec >= 0.9 && ec >= 1.3 * e1) {
        s = c;
        break;
      }
    }
  }
  if (s > nk / 4) s = nk / 4 > 0 ? nk / 4 : 1;         // keep >= 4 k-tiles per split
  const int per = cdiv(nk, s);
  *splits = cdiv(nk, per);                             // no empty split
  return PV_OK;
}

size_t pv_gemm_splitk_ws_bytes(int M, int N, int splits) { return (size_t)splits * M * N * sizeof(float); }

int pv_gemm_bf16_nt_splitk(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int splits,
                           void* workspace, float* C, int ldc, int accumulate, const int* m_live, const int* k_live,
                           void* stream) {
  int rc = check_gemm(A, B, M, N, K, lda, ldb, "pv_gemm_bf16_nt_splitk");
  if (rc) return rc;
  PV_REQUIRE(C && workspace && ldc >= N && splits >= 1 && splits <= K / BK, "pv_gemm_bf16_nt_splitk: bad argument");
  // with a device-side row bound `splits` is an upper limit: the kernel lowers it until live tiles x splits fits the
  // resident workgroups once
  const int slots = m_live ? persistent_grid() : 0;
  GemmArgs g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, splits, slots, m_live, k_live};
  EpiStore e{(float*)workspace, nullptr, nullptr, M, N, N, 0, 0, 0, (long)M * N, 0};
  hipStream_t st = (hipStream_t)stream;
  rc = launch(g, e, st, "pv_gemm_bf16_nt_splitk");
  if (rc) return rc;
  const long n = (long)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, (const float*)workspace,
                     splits, n, M, N, C, ldc, accumulate, m_live, slots);
  PV_LAUNCH_CHECK("splitk_reduce_kernel");
  return PV_OK;
}

int pv_gemm_bf16_tn_splitk(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int splits,
                           void* workspace, float* C, int ldc, int accumulate, void* stream) {
  PV_REQUIRE(A && B && C, "pv_gemm_bf16_tn_splitk: null argument");
  PV_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0, "pv_gemm_bf16_tn_splitk: K=%d must be a positive multiple of %d", K, BK);
  PV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N && lda >= 8 && ldb >= 8 && ldc >= N,
             "pv_gemm_bf16_tn_splitk: lda=%d ldb=%d must be multiples of 8 and >= M=%d / N=%d, ldc=%d >= N", lda, ldb, M, N, ldc);
  PV_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "pv_gemm_bf16_tn_splitk: operands must be 16-byte aligned");
  PV_REQUIRE(splits >= 1 && splits <= K / BK && (splits == 1 || workspace), "pv_gemm_bf16_tn_splitk: bad split count %d", splits);
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, splits, 0, nullptr, nullptr};
  if (splits == 1) {
    EpiStore e{C, nullptr, nullptr, M, N, ldc, 0, accumulate, 0, 0, 0};
    return launch<EpiStore, 2, true>(g, e, st, "pv_gemm_bf16_tn_splitk");
  }
  EpiStore e{(float*)workspace, nullptr, nullptr, M, N, N, 0, 0, 0, (long)M * N, 0};
  int rc = launch<EpiStore, 2, true>(g, e, st, "pv_gemm_bf16_tn_splitk");
  if (rc) return rc;
  const long n = (long)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, (const float*)workspace,
                     splits, n, M, N, C, ldc, accumulate, (const int*)nullptr, 0);
  PV_LAUNCH_CHECK("splitk_reduce_kernel");
  return PV_OK;
}

int pv_vocab_compact(const int64_t* target, int T, int64_t ignore_index, int64_t* perm, int* count, void* stream) {
  PV_REQUIRE(target && perm && count && T > 0, "pv_vocab_compact: bad argument");
  hipLaunchKernelGGL(vocab_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, target, T, ignore_index, perm, count);
  PV_LAUNCH_CHECK("vocab_compact_kernel");
  return PV_OK;
}

int pv_vocab_gather(const void* h, int ldh, const int64_t* target, const int64_t* perm, int T, int H, void* h_c, int ldc,
                    int64_t* target_c, void* stream) {
  PV_REQUIRE(h && target && perm && h_c && target_c && T > 0 && H > 0 && H % 8 == 0 && ldh % 8 == 0 && ldc % 8 == 0 &&
                 ldh >= H && ldc >= H, "pv_vocab_gather: bad argument (H and the leading dimensions must be multiples of 8)");
  PV_REQUIRE(((uintptr_t)h & 15) == 0 && ((uintptr_t)h_c & 15) == 0, "pv_vocab_gather: rows must be 16-byte aligned");
  const long n = (long)T * (H / 8);
  const long gx = (n + 255) / 256;
  hipLaunchKernelGGL(vocab_gather_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)h, ldh, target, perm, T, H / 8, (bf16_t*)h_c, ldc, target_c);
  PV_LAUNCH_CHECK("vocab_gather_kernel");
  return PV_OK;
}

// the same gather with the targets also as int32 (what the fused vocabulary kernels stage through LDS); perm == NULL:
// identity order (h_c may then be NULL: only the int32 targets are produced)
__global__ void vocab_gather32_kernel(const bf16_t* __restrict__ h, int ldh, const int64_t* __restrict__ target,
                                      const int64_t* __restrict__ perm, int T, int chunks, bf16_t* __restrict__ hc, int ldc,
                                      int64_t* __restrict__ target_c, int* __restrict__ target32) {
  const long n = (long)T * chunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / chunks), ch = (int)(i % chunks);
    const long src = perm ? perm[row] : row;
    if (hc) *(uint4*)(hc + (size_t)row * ldc + ch * 8) = *(const uint4*)(h + (size_t)src * ldh + ch * 8);
    if (ch == 0) {
      const int64_t t = target[src];
      if (target_c) target_c[row] = t;
      target32[row] = (t < -2147483647 || t > 2147483647) ? -2147483647 : (int)t;
    }
  }
}

extern "C" int pv_vocab_gather32(const void* h, int ldh, const int64_t* target, const int64_t* perm, int T, int H, void* h_c,
                                 int ldc, int64_t* target_c, int* target32, void* stream) {
  PV_REQUIRE(target && target32 && T > 0 && H > 0 && H % 8 == 0, "pv_vocab_gather32: bad argument");
  PV_REQUIRE(!h_c || (h && ldh % 8 == 0 && ldc % 8 == 0 && ldh >= H && ldc >= H && ((uintptr_t)h & 15) == 0 &&
                      ((uintptr_t)h_c & 15) == 0), "pv_vocab_gather32: rows must be 16-byte aligned, leading dimensions multiples of 8");
  const long n = (long)T * (H / 8);
  const long gx = (n + 255) / 256;
  hipLaunchKernelGGL(vocab_gather32_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)h, ldh, target, perm, T, H / 8, (bf16_t*)h_c, ldc, target_c, target32);
  PV_LAUNCH_CHECK("vocab_gather32_kernel");
  return PV_OK;
}

int pv_scatter_rows_f32(const float* src, long lds, const int64_t* perm, const int* count, int T, int cols, float* dst,
                        long ldd, void* stream) {
  PV_REQUIRE(src && perm && count && dst && T > 0 && cols > 0 && lds >= cols && ldd >= cols, "pv_scatter_rows_f32: bad argument");
  const long n = (long)T * cols;
  const long gx = (n + 255) / 256;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream, src, lds,
                     perm, count, T, cols, dst, ldd);
  PV_LAUNCH_CHECK("scatter_rows_kernel");
  return PV_OK;
}

size_t pv_vocab_nll_fwd_ws_bytes(int T, int V) {
  return (size_t)cdiv(V, 64) * round_up(T, 64) * sizeof(float2) + (size_t)round_up(T, 64) * sizeof(float);
}

int pv_vocab_nll_fwd(const void* h, int ldh, const void* W, int ldw, const float* bias,
                     const int64_t* target, int T, int V, int H, float inv_temp, int64_t ignore_index,
                     void* workspace, float* lse, float* nll, const int* live_rows, void* stream) {
  int rc = check_gemm(h, W, T, V, H, ldh, ldw, "pv_vocab_nll_fwd");
  if (rc) return rc;
  PV_REQUIRE(bias && target && workspace && lse && nll, "pv_vocab_nll_fwd: null argument");
  const int Tpad = round_up(T, 64), nslab = cdiv(V, 64);
  float2* part = (float2*)workspace;
  float* tgt_logit = (float*)(part + (size_t)nslab * Tpad);
  hipStream_t st = (hipStream_t)stream;
  // rows whose target is out of range (e.g. ignore_index < 0) never get a target logit written
  PV_HIP(hipMemsetAsync(tgt_logit, 0, (size_t)Tpad * sizeof(float), st));
  GemmArgs g{(const bf16_t*)h, (const bf16_t*)W, T, V, H, ldh, ldw, 1, 0, live_rows, nullptr};
  EpiSoftmax e{bias, target, part, tgt_logit, T, V, Tpad, inv_temp};
  rc = big_tile(T, V, H) ? launch<decltype(e), 4>(g, e, st, "pv_vocab_nll_fwd") : launch(g, e, st, "pv_vocab_nll_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(vocab_reduce_kernel, dim3(cdiv(T, 16)), dim3(256), 0, st, part, tgt_logit, target, T,
                     Tpad, nslab, ignore_index, live_rows, lse, nll);
  PV_LAUNCH_CHECK("vocab_reduce_kernel");
  return PV_OK;
}

int pv_vocab_nll_bwd_prob(const void* h, int ldh, const void* W, int ldw, const float* bias,
                          const int64_t* target, const float* lse, const float* gscale, int T, int V,
                          int H, float inv_temp, void* P, int ldp, void* PT, int ldpt,
                          float* dbias_part, float* dbias, const int* live_rows, void* stream) {
  int rc = check_gemm(h, W, T, V, H, ldh, ldw, "pv_vocab_nll_bwd_prob");
  if (rc) return rc;
  const int Tpad = round_up(T, 64), Vpad = round_up(V, 64);
  PV_REQUIRE(bias && target && lse && gscale && P && PT && dbias_part && dbias, "pv_vocab_nll_bwd_prob: null argument");
  PV_REQUIRE(ldp >= Vpad && ldp % 8 == 0 && ldpt >= Tpad && ldpt % 8 == 0,
             "pv_vocab_nll_bwd_prob: need ldp >= %d and ldpt >= %d (multiples of 8)", Vpad, Tpad);
  PV_REQUIRE(((uintptr_t)P & 15) == 0 && ((uintptr_t)PT & 15) == 0, "pv_vocab_nll_bwd_prob: P/PT must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (live_rows) PV_HIP(hipMemsetAsync(dbias_part, 0, (size_t)(Tpad / 64) * ldp * sizeof(float), st));   // skipped tiles
  GemmArgs g{(const bf16_t*)h, (const bf16_t*)W, T, V, H, ldh, ldw, 1, 0, live_rows, nullptr};
  EpiProb e{bias, target, lse, gscale, (bf16_t*)P, (bf16_t*)PT, dbias_part, T, V, Tpad, Vpad, ldp, ldpt, inv_temp};
  rc = big_tile(T, V, H) ? launch<decltype(e), 4>(g, e, st, "pv_vocab_nll_bwd_prob") : launch(g, e, st, "pv_vocab_nll_bwd_prob");
  if (rc) return rc;
  // dbias[v] = sum over row-slabs of dbias_part[slab][v]   (fixed order: bitwise reproducible)
  return pv_colsum_f32(dbias_part, Tpad / 64, ldp, V, dbias, 0, stream);
}

size_t pv_vocab_argmax_ws_bytes(int T, int V) { return (size_t)cdiv(V, 64) * round_up(T, 64) * sizeof(ArgPart); }

int pv_vocab_argmax(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H, float inv_temp,
                    int mask_col, void* workspace, int64_t* ids, float* logp, void* stream) {
  int rc = check_gemm(h, W, T, V, H, ldh, ldw, "pv_vocab_argmax");
  if (rc) return rc;
  PV_REQUIRE(bias && workspace && ids && logp, "pv_vocab_argmax: null argument");
  const int Tpad = round_up(T, 64), nslab = cdiv(V, 64);
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g{(const bf16_t*)h, (const bf16_t*)W, T, V, H, ldh, ldw, 1, 0, nullptr, nullptr};
  EpiArgmax e{bias, (ArgPart*)workspace, T, V, Tpad, mask_col, inv_temp};
  rc = launch(g, e, st, "pv_vocab_argmax");
  if (rc) return rc;
  hipLaunchKernelGGL(vocab_argmax_reduce_kernel, dim3(cdiv(T, 16)), dim3(256), 0, st, (const ArgPart*)workspace,
                     (const float*)nullptr, T, Tpad, nslab, ids, logp);
  PV_LAUNCH_CHECK("vocab_argmax_reduce_kernel");
  return PV_OK;
}

int pv_vocab_topk(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H, float inv_temp,
                  int mask_col, int k, void* workspace, int64_t* ids, float* logp, void* stream) {
  int rc = check_gemm(h, W, T, V, H, ldh, ldw, "pv_vocab_topk");
  if (rc) return rc;
  PV_REQUIRE(bias && workspace && ids && logp, "pv_vocab_topk: null argument");
  PV_REQUIRE(k >= 1 && k <= TOPK_MAX, "pv_vocab_topk: k = %d outside 1..%d", k, TOPK_MAX);
  PV_REQUIRE(H % 8 == 0 && ldw % 8 == 0, "pv_vocab_topk: H and ldw must be multiples of 8 (16-byte weight reads)");
  const int Tpad = round_up(T, 64), nslab = cdiv(V, 64);
  const size_t lds = ((size_t)nslab + H + (size_t)k * 64) * sizeof(float);
  PV_REQUIRE(lds <= 60 * 1024, "pv_vocab_topk: V = %d, H = %d need %zu bytes of LDS (limit 60 KiB)", V, H, lds);
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g{(const bf16_t*)h, (const bf16_t*)W, T, V, H, ldh, ldw, 1, 0, nullptr, nullptr};
  EpiArgmax e{bias, (ArgPart*)workspace, T, V, Tpad, mask_col, inv_temp};
  rc = launch(g, e, st, "pv_vocab_topk");
  if (rc) return rc;
  hipLaunchKernelGGL(vocab_topk_kernel, dim3(T), dim3(256), lds, st, (const ArgPart*)workspace, Tpad, nslab, (const bf16_t*)h, ldh,
                     (const bf16_t*)W, ldw, bias, V, H, inv_temp, mask_col, k, ids, logp);
  PV_LAUNCH_CHECK("vocab_topk_kernel");
  return PV_OK;
}

size_t pv_vocab_sample_ws_bytes(int T, int V) {
  return pv_vocab_argmax_ws_bytes(T, V) + (size_t)cdiv(V, 64) * round_up(T, 64) * sizeof(float);
}

int pv_vocab_sample(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H, float inv_temp,
                    int mask_col, const float* noise_q, long noise_ld, const void* rng_state, int site,
                    void* workspace, int64_t* ids, float* logp, void* stream) {
  int rc = check_gemm(h, W, T, V, H, ldh, ldw, "pv_vocab_sample");
  if (rc) return rc;
  PV_REQUIRE(bias && workspace && ids && logp, "pv_vocab_sample: null argument");
  PV_REQUIRE(noise_q ? noise_ld >= V : rng_state != nullptr, "pv_vocab_sample: needs noise_q with noise_ld >= V, or rng_state");
  const int Tpad = round_up(T, 64), nslab = cdiv(V, 64);
  hipStream_t st = (hipStream_t)stream;
  float* xbest = (float*)((char*)workspace + pv_vocab_argmax_ws_bytes(T, V));
  GemmArgs g{(const bf16_t*)h, (const bf16_t*)W, T, V, H, ldh, ldw, 1, 0, nullptr, nullptr};
  EpiSample e{bias, (ArgPart*)workspace, xbest, noise_q, noise_ld, (const RngState*)rng_state, (uint32_t)site, T, V, Tpad, mask_col, inv_temp};
  rc = launch(g, e, st, "pv_vocab_sample");
  if (rc) return rc;
  hipLaunchKernelGGL(vocab_argmax_reduce_kernel, dim3(cdiv(T, 16)), dim3(256), 0, st, (const ArgPart*)workspace,
                     (const float*)xbest, T, Tpad, nslab, ids, logp);
  PV_LAUNCH_CHECK("vocab_argmax_reduce_kernel");
  return PV_OK;
}

/* The same greedy step for callers that cannot allocate (HIP graph capture, tight serving loops): the per-slab partials
 * live in a file-scope device array instead of a caller workspace.  Each GPU holds its own copy of that array at its own
 * address, looked up once per device with hipGetSymbolAddress (the array's NAME is only a host-side stand-in, never a
 * device address); every call on a GPU shares it, so calls that overlap on one GPU race.  Nothing has to start at zero:
 * the GEMM epilogue writes every (slab, row) record the reduce kernel reads. */
int pv_vocab_argmax_noalloc(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H,
                            float inv_temp, int mask_col, int64_t* ids, float* logp, void* stream) {
  int rc = check_gemm(h, W, T, V, H, ldh, ldw, "pv_vocab_argmax_noalloc");
  if (rc) return rc;
  PV_REQUIRE(bias && ids && logp, "pv_vocab_argmax_noalloc: null argument");
  const size_t need = pv_vocab_argmax_ws_bytes(T, V);
  if (need > ARGMAX_SCRATCH_BYTES) {
    pv_set_error("pv_vocab_argmax_noalloc: ceil(V/64) * round_up(T,64) * 16 = %zu bytes exceeds the built-in scratch of "
                 "%zu bytes (T=%d, V=%d); use pv_vocab_argmax with a workspace", need, (size_t)ARGMAX_SCRATCH_BYTES, T, V);
    return PV_ERR_LIMIT;
  }
  void* ws = nullptr;
  rc = argmax_scratch_for_current_device(&ws);
  if (rc) return rc;
  return pv_vocab_argmax(h, ldh, W, ldw, bias, T, V, H, inv_temp, mask_col, ws, ids, logp, stream);
}

size_t pv_vocab_argmax_noalloc_max_bytes(void) { return ARGMAX_SCRATCH_BYTES; }

int pv_vocab_logprob_dense(const void* h, int ldh, const void* W, int ldw, const float* bias,
                           const float* lse, int T, int V, int H, float inv_temp, float* out, void* stream) {
  int rc = check_gemm(h, W, T, V, H, ldh, ldw, "pv_vocab_logprob_dense");
  if (rc) return rc;
  PV_REQUIRE(bias && lse && out, "pv_vocab_logprob_dense: null argument");
  GemmArgs g{(const bf16_t*)h, (const bf16_t*)W, T, V, H, ldh, ldw, 1, 0, nullptr, nullptr};
  EpiLogProb e{bias, lse, out, T, V, inv_temp};
  return launch(g, e, (hipStream_t)stream, "pv_vocab_logprob_dense");
}

}  // extern "C"
