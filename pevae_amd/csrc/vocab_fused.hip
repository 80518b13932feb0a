// Fused vocabulary projection + softmax + NLL, forward AND backward, for H = 512 on gfx950 -- no [T, V] matrix of
// logits or probabilities ever reaches HBM (decoders.py:191-193 + criterions.py:55-64 of the reference, and their
// backward).  Two flash-attention-shaped kernels, each two chained MFMA products per (32 rows x 32 columns) block:
//
//   vocab_fwd_dh_kernel   S^T = W h^T  ->  online softmax over the vocabulary  ->  O^T += W^T P^T      (per 128-row group)
//                         gives lse[t], the target logit, and U[t] = sum_v softmax[t, v] W[v] - W[y_t], the gradient of the
//                         NLL with respect to h[t] for a unit upstream gradient: dH needs no pass of its own in backward
//   vocab_dw_kernel       S = h W^T    ->  P = g (softmax - onehot)          ->  Z += P^T h            (per 128-column tile)
//                         gives dW and dbias
//
// Structure (both): one 256-thread workgroup per CU, ONE wave per SIMD with the whole 512-register file: a wave owns a
// 32 x 512 fp32 accumulator (256 registers) and keeps its stationary operand -- its 32 rows of h, or its 32 rows of W -- as
// 32 MFMA fragments in another 128 registers; the streamed operand (32 x 512 bf16 tiles of W, or of h) is staged
// global -> LDS with global_load_lds_dwordx4 into a 2-deep ring and is read twice: by rows (ds_read_b128, the first
// product) and by columns (ds_read_b64_tr_b16, the second product) from ONE image -- four 128-column panels of 256-byte
// rows with the XOR swizzle that serves both kinds of read.  The first product's accumulator IS the second product's
// operand (v_mfma_f32_32x32x16_bf16: the accumulator's row index is the next product's k index), so probabilities go from
// registers to registers.  Staged bytes per flop: 1/256 (the 128 x 128 tile GEMMs this replaces: 1/64).
//
// Work split: "stream-K" over the linear item space (unit-major: row group / column tile; item: one 32-wide slice of the
// streamed operand): workgroup w of G takes items [w N / G, (w+1) N / G), so every CU gets the same number of items
// whatever the live row count is (a device scalar: the launch geometry is static, HIP-graph capturable).  A workgroup's
// share of a unit is a SEGMENT; every segment leaves a partial result in a slab of the workspace and a second kernel
// folds a unit's segments in workgroup order (fixed order: bitwise reproducible, no atomics).
#include "common.h"
#include "dropout.h"
#include "../../include/pevae_hip.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4* ltr_t;

constexpr int HD = 512;                 // hidden width the kernels are built for (16 column blocks of 32)
constexpr int NCB = HD / 32;            // 16 accumulator blocks per wave
constexpr int NKS = HD / 16;            // 32 k-steps of the first product
constexpr int TILE_BYTES = 32 * HD * 2; // one streamed tile: 32 rows x 512 bf16 = 32 KiB, four panels of 8 KiB
constexpr int AUX_BYTES = 1024;         // per-tile side data (bias slice, or lse / gscale / targets of 32 rows)
constexpr int BUF_BYTES = TILE_BYTES + AUX_BYTES;
constexpr int SMEM_BYTES = 2 * BUF_BYTES;
constexpr int UNIT = 128;               // rows (forward) / columns (dW) a workgroup's accumulators cover: 4 waves x 32
constexpr float NEG_BIG = -1e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 28.f;     // log2 units: a row's maximum may exceed the one in use by this much before O is rescaled

// Diagnostic build only (-DPV_VOCAB_STAMPS): every wave accumulates shader-clock cycles per section of its item loop
// (s_memtime; the stamp drains the LDS queue, so read the SHARES, never this build's run time) into a table of its own
// that no kernel reads; tools/vocab_stamps.py copies it out with pv_vocab_stamps_copy.
#ifdef PV_VOCAB_STAMPS
constexpr int NSTAMP = 8;
__device__ unsigned long long g_vstamps[2][256][4][NSTAMP];
#define PV_STAMP_DECL unsigned long long st_acc[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0; \
  { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define PV_STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
  st_acc[i] += t_ - st_last; st_last = t_; }
#define PV_STAMP_FLUSH(k) if (lane == 0 && blockIdx.x < 256) { _Pragma("unroll") for (int i_ = 0; i_ < NSTAMP; ++i_) g_vstamps[k][blockIdx.x][w][i_] = st_acc[i_]; }
#else
#define PV_STAMP_DECL
#define PV_STAMP(i)
#define PV_STAMP_FLUSH(k)
#endif

struct Segs {                            // a workgroup's item range and how units map to its slabs
  long beg, end;
};
__host__ __device__ __forceinline__ long seg_beg(long w, long N, int G) { return w * N / G; }

// Which stream-K range a workgroup takes.  Ranges are contiguous in the (unit, item) space, so range k starts at item
// phase(k) = seg_beg(k) mod `period` of its unit and walks the streamed operand cyclically from there.  Left in launch
// order, the 32 workgroups of an XCD start at 32 phases spread over the whole operand (7 MB of h, 20 MB of W) and every
// tile is fetched into that XCD's 4 MB L2 once per workgroup: FETCH_SIZE showed 1.13 GB per dW launch, 157 column tiles x
// 7.2 MB.  Here the ranges are sorted by phase and dealt out in runs of G/8 to the XCDs (workgroups with equal blockIdx.x
// % 8 share an XCD under round-robin dispatch): an XCD's workgroups then walk within period/8 items of each other and a
// tile is fetched once per XCD and pass.  A permutation only: the split, the slabs and the fold order are unchanged, and
// placement is speed, not correctness.  `scratch`: G + 1 ints of LDS, free again on return.
__device__ __forceinline__ int phase_sorted_range(long N, int G, int period, int* scratch) {
  if ((G & 7) || N < G || period <= 0) return blockIdx.x;
  const int tid = threadIdx.x, want = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  for (int t = tid; t < G; t += blockDim.x) scratch[t] = (int)(seg_beg(t, N, G) % period);
  __syncthreads();
  for (int t = tid; t < G; t += blockDim.x) {
    const int ph = scratch[t];
    int rank = 0;
    for (int u = 0; u < G; ++u) {
      const int pu = scratch[u];
      rank += (pu < ph || (pu == ph && u < t)) ? 1 : 0;
    }
    if (rank == want) scratch[G] = t;
  }
  __syncthreads();
  const int k = scratch[G];
  __syncthreads();
  return k;
}

// ---- staging: wave w fills panel w (32 rows x 256 B) of a tile with 8 LDS-DMA pieces of 4 rows ------------------------
// image: off(panel, row, chunk) = panel * 8192 + row * 256 + 16 * (chunk ^ swz(row)), swz(row) = ((row & 3) << 2) | ((row >> 2) & 3)
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, long ld, int row0, int max_row, char* buf, int w,
                                           int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + (lane >> 4);
    const int ch = (lane & 15) ^ (((lane >> 4) << 2) | (i & 3));
    const bf16_t* p = src + (long)min(row0 + row, max_row) * ld + w * 128 + ch * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(buf + w * 8192 + i * 1024), 16, 0, 0);
  }
}

// row read for k-step ks (A operand of the first product): lane (r = lane & 31, hh = lane >> 5) gets row r, k = 16 ks + 8 hh ..
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int ks, int rbase, int swh) {
  const int off = (ks >> 3) * 8192 + rbase + 16 * (((2 * ks) & 15) ^ swh);
  return __builtin_bit_cast(bf16x8, *(const uint4*)(tile + off));
}

// column read for column block c, k-step s (operand of the second product): the 8 tile rows
// 16 s + 8 (j >> 2) + 4 hh + (j & 3), j = 0..7, of column 32 c + (lane & 31) -- the k order in which an accumulator of the
// first product hands its rows to the next MFMA
struct TrAddr {
  int base[2];      // per part u: 256 * (8 u + 4 hh + q) + 8 * (p & 1)
  int sw[2];        // per part u: (q << 2) | (2 u + hh)
  int lane_part;    // 2 cg + (p >> 1)
};
__device__ __forceinline__ TrAddr make_tr(int lane) {
  const int g = lane >> 4, hh = g >> 1, cg = g & 1, q = (lane & 15) >> 2, p = lane & 3;
  TrAddr t;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    t.base[u] = 256 * (8 * u + 4 * hh + q) + 8 * (p & 1);
    t.sw[u] = (q << 2) | (2 * u + hh);
  }
  t.lane_part = 2 * cg + (p >> 1);
  return t;
}
__device__ __forceinline__ bf16x8 col_frag(const char* tile, const TrAddr& t, int c, int s) {
  bf16x4 part[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int off = (c >> 2) * 8192 + 4096 * s + t.base[u] + 16 * (((4 * (c & 3)) | t.lane_part) ^ t.sw[u]);
    part[u] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(tile + off));
  }
  return bf16x8{part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// First product, X += rows(tile) x stat[ks] over the 32 k-steps.  The LDS reads run one batch of four fragments ahead of
// the MFMAs that consume them (an LDS round trip is about four 32-cycle MFMAs long); left to itself hipcc emits
// read -> wait -> MFMA with one fragment register set and the chain runs at LDS latency.  ROWS_ARE_A: the streamed rows
// are the A operand (forward kernel: rows of W against h fragments give S^T) or the B operand ... both kernels stream A.
#define PV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void first_product(const char* tile, int rbase, int swh, const bf16x8 (&stat)[NKS], f32x16& X) {
  bf16x8 fa[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = row_frag(tile, i, rbase, swh);
#pragma unroll
  for (int b = 0; b < NKS / 4; b += 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = row_frag(tile, 4 * (b + 1) + i, rbase, swh);
    PV_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 4; ++i) X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], stat[4 * b + i], X, 0, 0, 0);
    PV_SCHED_FENCE();
    if (b + 2 < NKS / 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = row_frag(tile, 4 * (b + 2) + i, rbase, swh);
    }
    PV_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 4; ++i) X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], stat[4 * (b + 1) + i], X, 0, 0, 0);
    PV_SCHED_FENCE();
  }
}

// Second product for k-step s: ACC[c] += (COLS_ARE_A ? cols(tile, c, s) x pf : pf x cols(tile, c, s)), c = 0..15, with the
// transposed reads one batch of four column fragments ahead.
template <bool COLS_ARE_A>
__device__ __forceinline__ void second_product(const char* tile, const TrAddr& tr, int s, const bf16x8& pf, f32x16 (&ACC)[NCB]) {
  bf16x8 fa[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = col_frag(tile, tr, i, s);
#pragma unroll
  for (int b = 0; b < NCB / 4; b += 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = col_frag(tile, tr, 4 * (b + 1) + i, s);
    PV_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ACC[4 * b + i] = COLS_ARE_A ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], pf, ACC[4 * b + i], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, fa[i], ACC[4 * b + i], 0, 0, 0);
    PV_SCHED_FENCE();
    if (b + 2 < NCB / 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = col_frag(tile, tr, 4 * (b + 2) + i, s);
    }
    PV_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ACC[4 * (b + 1) + i] = COLS_ARE_A ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], pf, ACC[4 * (b + 1) + i], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, fb[i], ACC[4 * (b + 1) + i], 0, 0, 0);
    PV_SCHED_FENCE();
  }
}

// =====================================================================================================================
// forward + dH:  per row t   lse, target logit, O[t] = sum_v exp(s_tv - m) W[v]  (partial over a vocabulary range)
// =====================================================================================================================
struct FwdArgs {
  const bf16_t* h;      // [T][ldh]   compact rows (live first)
  const bf16_t* W;      // [V][ldw]
  const float* bias;    // [V]
  const int* target;    // [T] int32, compact order
  const int* live;      // device scalar: rows to compute (NULL: all T)
  int T, V, ldh, ldw, max_seg;
  float inv_temp;
  float* o_slab;        // [G * max_seg][HD][UNIT]
  float4* stats;        // [G * max_seg][UNIT]  (m, l, target logit, -)
};

__global__ __launch_bounds__(256, 1) void vocab_fwd_dh_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r32 = lane & 31, hh = lane >> 5;
  const int G = gridDim.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NVI = (a.V + 31) / 32, NRG = (nlive + UNIT - 1) / UNIT;
  const long N = (long)NRG * NVI;
  const int rid = phase_sorted_range(N, G, NVI, (int*)smem);
  const long beg = seg_beg(rid, N, G), end = seg_beg(rid + 1, N, G);
  const int rbase = 256 * r32, swh = (((r32 & 3) << 2) | ((r32 >> 2) & 3)) ^ hh;
  const TrAddr tr = make_tr(lane);
  int seg = 0;
  PV_STAMP_DECL
  for (long at = beg; at < end; ++seg) {
    const int rg = (int)(at / NVI), vi0 = (int)(at % NVI);
    const int nvi = (int)min((long)NVI - vi0, end - at);
    at += nvi;
    // ---- this wave's 32 rows of h as 32 B-operand fragments (lane: row r32, k = 16 ks + 8 hh ..) ----------------------
    const int t_mine = rg * UNIT + 32 * w + r32;
    const bf16_t* hrow = a.h + (long)min(t_mine, a.T - 1) * a.ldh + 8 * hh;
    bf16x8 hf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) hf[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(hrow + 16 * ks));
    const int y = a.target[min(t_mine, a.T - 1)];
    f32x16 O[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) O[c][j] = 0.f;
    float m_used = NEG_BIG, l_sum = 0.f, tl = 0.f;       // running maximum and target logit in log2 units
    const float it_l2e = a.inv_temp * LOG2E;

    auto stage = [&](int k) {
      char* buf = smem + (k & 1) * BUF_BYTES;
      const int v0 = (vi0 + k) * 32;
      stage_tile(a.W, a.ldw, v0, a.V - 1, buf, w, lane);
      if (w == 0)
        __builtin_amdgcn_global_load_lds((gptr_t)(a.bias + min(v0 + r32, a.V - 1)), (lptr_t)(buf + TILE_BYTES), 4, 0, 0);
    };
    stage(0);
    PV_STAMP(0)
    for (int k = 0; k < nvi; ++k) {
      __syncthreads();                       // tile k has landed (every wave waited for its own pieces), tile k-1 is done with
      PV_STAMP(1)
      if (k + 1 < nvi) stage(k + 1);
      PV_STAMP(2)
      const char* tile = smem + (k & 1) * BUF_BYTES;
      const float* bias_t = (const float*)(tile + TILE_BYTES);
      const int v0 = (vi0 + k) * 32;
      // ---- S^T block [32 v][32 t]: rows of W (LDS) x this wave's h fragments -----------------------------------------
      f32x16 X;
#pragma unroll
      for (int j = 0; j < 16; ++j) X[j] = 0.f;
      first_product(tile, rbase, swh, hf, X);
      PV_STAMP(3)
      // ---- scaled logits in LOG2 units (exp is v_exp_f32 = 2^x: no multiply per element), running maximum (shared by the
      // two lane halves of a row), target logit.  The column mask and the target pick are needed by one tile in hundreds:
      // both sit behind wave-uniform branches.
      float sv[16], mx = NEG_BIG;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 b4 = *(const f32x4*)(bias_t + 8 * g4 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[4 * g4 + i] = (X[4 * g4 + i] + b4[i]) * it_l2e;
      }
      if (v0 + 32 > a.V) {                                 // the last tile: columns past V
#pragma unroll
        for (int j = 0; j < 16; ++j) sv[j] = v0 + 8 * (j >> 2) + 4 * hh + (j & 3) < a.V ? sv[j] : NEG_BIG;
      }
      if (__any((unsigned)(y - v0) < 32u)) {               // some row of this wave has its target in this tile
#pragma unroll
        for (int j = 0; j < 16; ++j) tl += v0 + 8 * (j >> 2) + 4 * hh + (j & 3) == y ? sv[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, sv[j]);
      mx = fmaxf(mx, xor32(mx));
      if (k == 0) {
        m_used = mx;                                       // O and l are still zero: nothing to rescale
      } else if (__builtin_expect(!__all(mx - m_used <= RESCALE_THR), 0)) {
        // Rare (a row's maximum grew by more than RESCALE_THR since the segment's first tile): bring O and l to the new
        // maximum.  The accumulators live in AGPRs; written as plain C++ the 256 multiplies pull them into VGPRs and the
        // whole kernel spills (944 bytes of scratch per lane), so each one is read / scaled / written back in place.
        // MFMA results need up to 18 wait states before v_accvgpr_read may see them; the assembler pads nothing inside asm.
        const float m_new = fmaxf(m_used, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_used - m_new);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            float tmp;
            asm volatile("v_accvgpr_read_b32 %1, %0\n\ts_nop 1\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\tv_accvgpr_write_b32 %0, %1"
                         : "+a"(O[c][j]), "=&v"(tmp) : "v"(alpha));
          }
        asm volatile("s_nop 7" ::: "memory");
        l_sum *= alpha;
        m_used = m_new;
      }
      bf16x8 pf[2];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = __builtin_amdgcn_exp2f(sv[j] - m_used);   // <= 2^RESCALE_THR; masked columns: 2^(-1e30 - m) = 0
        l_sum += p;
        pf[j >> 3][j & 7] = (__bf16)p;
      }
      PV_STAMP(4)
      // ---- O^T[hc][t] += W^T[hc][v] P^T[v][t]: columns of the same W tile x the probabilities, register to register ----
      second_product<true>(tile, tr, 0, pf[0], O);
      second_product<true>(tile, tr, 1, pf[1], O);
      PV_STAMP(5)
    }
    // ---- partial result of this segment: O^T as [hc][t] (128-byte runs), (m, l, target logit) per row -----------------
    const long slab = (long)rid * a.max_seg + seg;
    float* o = a.o_slab + slab * HD * UNIT + 32 * w + r32;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) o[(long)(32 * c + (j & 3) + 8 * (j >> 2) + 4 * hh) * UNIT] = O[c][j];
    l_sum += xor32(l_sum);
    tl += xor32(tl);
    if (hh == 0) a.stats[slab * UNIT + 32 * w + r32] = make_float4(m_used * LN2, l_sum, tl * LN2, 0.f);
    __syncthreads();                           // the ring restarts with the next segment
    PV_STAMP(6)
  }
  PV_STAMP_FLUSH(0)
}

// fold a row group's segments: lse, nll, U[t][:] = sum_seg O_seg e^(m_seg - M) / L - W[y_t]
struct FwdCombineArgs {
  const float* o_slab;
  const float4* stats;
  const bf16_t* W;
  const int* target;
  const int* live;
  int T, V, ldw, max_seg, G;
  long ignore_index;
  float* lse;           // [T]
  float* nll;           // [T]
  float* U;             // [T][HD]
};

__global__ __launch_bounds__(256) void vocab_fwd_combine_kernel(FwdCombineArgs a) {
  __shared__ float tile[32][UNIT + 1];
  __shared__ float rowM[UNIT];
  __shared__ float invL[UNIT];
  const int rg = blockIdx.x, hc0 = blockIdx.y * 32, tid = threadIdx.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NVI = (a.V + 31) / 32, NRG = (nlive + UNIT - 1) / UNIT;
  if (rg >= NRG) {                             // rows nobody computed (ignored tokens behind the live ones): zeros
    for (int i = tid; i < UNIT * 32; i += 256) {
      const int t = rg * UNIT + (i >> 5);
      if (t < a.T) a.U[(long)t * HD + hc0 + (i & 31)] = 0.f;
    }
    if (blockIdx.y == 0 && tid < UNIT && rg * UNIT + tid < a.T) {
      a.lse[rg * UNIT + tid] = 0.f;
      a.nll[rg * UNIT + tid] = 0.f;
    }
    return;
  }
  const long N = (long)NRG * NVI, x0 = (long)rg * NVI, x1 = x0 + NVI;
  // the workgroups whose item range meets [x0, x1) are consecutive: w0 .. w1 - 1 (some of them may be empty when there
  // are fewer items than workgroups); workgroup w keeps this unit in slab w * max_seg + (rg - first unit of w)
  int w0 = (int)(x0 * a.G / N);
  while (w0 > 0 && seg_beg(w0, N, a.G) > x0) --w0;
  while (seg_beg(w0 + 1, N, a.G) <= x0) ++w0;
  int w1 = w0 + 1;
  while (w1 < a.G && seg_beg(w1, N, a.G) < x1) ++w1;
  auto slab_of = [&](int w) -> long {
    const long b = seg_beg(w, N, a.G), e = seg_beg(w + 1, N, a.G);
    return (e <= x0 || e == b) ? -1 : (long)w * a.max_seg + (rg - (int)(b / NVI));
  };
  if (tid < UNIT) {
    float M = NEG_BIG;
    for (int w = w0; w < w1; ++w) {
      const long sl = slab_of(w);
      if (sl >= 0) M = fmaxf(M, a.stats[sl * UNIT + tid].x);
    }
    float L = 0.f, tl = 0.f;
    for (int w = w0; w < w1; ++w) {
      const long sl = slab_of(w);
      if (sl < 0) continue;
      const float4 st = a.stats[sl * UNIT + tid];
      L += st.y * __expf(st.x - M);
      tl += st.z;
    }
    rowM[tid] = M;
    invL[tid] = 1.f / L;
    const int t = rg * UNIT + tid;
    if (blockIdx.y == 0 && t < a.T) {
      const bool is_live = t < nlive && (long)a.target[t] != a.ignore_index;
      const float lse = M + logf(L);
      a.lse[t] = t < nlive ? lse : 0.f;
      a.nll[t] = is_live ? lse - tl : 0.f;
    }
  }
  __syncthreads();
  const int t = tid & (UNIT - 1), hsel = tid >> 7;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int w = w0; w < w1; ++w) {                  // segments in workgroup order: a fixed order of additions
    const long sl = slab_of(w);
    if (sl < 0) continue;
    const float sc = __expf(a.stats[sl * UNIT + t].x - rowM[t]);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += a.o_slab[(sl * HD + hc0 + hsel + 2 * i) * UNIT + t] * sc;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[hsel + 2 * i][t] = acc[i] * invL[t];
  __syncthreads();
  for (int i = tid; i < UNIT * 32; i += 256) {
    const int tl_ = i >> 5, c = i & 31, tg = rg * UNIT + tl_;
    if (tg >= a.T) continue;
    float u = 0.f;
    if (tg < nlive) {
      const int y = a.target[tg];
      const float wy = (y >= 0 && y < a.V && (long)y != a.ignore_index) ? bf2f(a.W[(long)y * a.ldw + hc0 + c]) : 0.f;
      u = ((long)y != a.ignore_index) ? tile[c][tl_] - wy : 0.f;
    }
    a.U[(long)tg * HD + hc0 + c] = u;
  }
}

// =====================================================================================================================
// dW / dbias:  per vocabulary column tile   Z[v][:] = sum_t P[t, v] h[t],  P = gs[t] (exp(s_tv - lse_t) - [v == y_t])
// =====================================================================================================================
struct DwArgs {
  const bf16_t* h;
  const bf16_t* W;
  const float* bias;
  const int* target;
  const float* lse;     // [T] compact
  const float* gs;      // [T] compact: dL/dnll, 0 for ignored rows (inv_temp is applied here)
  const int* live;
  int T, V, ldh, ldw, max_seg;
  float inv_temp;
  float* z_slab;        // [G * max_seg][UNIT][HD]
  float* b_slab;        // [G * max_seg][UNIT]
};

__global__ __launch_bounds__(256, 1) void vocab_dw_kernel(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r32 = lane & 31, hh = lane >> 5;
  const int G = gridDim.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NCT = (a.V + UNIT - 1) / UNIT, NRI = (nlive + 31) / 32;
  const long N = (long)NCT * NRI;
  const int rid = phase_sorted_range(N, G, NRI, (int*)smem);
  const long beg = seg_beg(rid, N, G), end = seg_beg(rid + 1, N, G);
  const int rbase = 256 * r32, swh = (((r32 & 3) << 2) | ((r32 >> 2) & 3)) ^ hh;
  const TrAddr tr = make_tr(lane);
  int seg = 0;
  PV_STAMP_DECL
  for (long at = beg; at < end; ++seg) {
    const int ct = (int)(at / NRI), ri0 = (int)(at % NRI);
    const int nri = (int)min((long)NRI - ri0, end - at);
    at += nri;
    // ---- this wave's 32 rows of W as 32 B-operand fragments ------------------------------------------------------------
    const int v_mine = ct * UNIT + 32 * w + r32;
    const bool v_ok = v_mine < a.V;
    const bf16_t* wrow = a.W + (long)min(v_mine, a.V - 1) * a.ldw + 8 * hh;
    bf16x8 wf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(wrow + 16 * ks));
    const float bias_v = a.bias[min(v_mine, a.V - 1)];
    f32x16 Z[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) Z[c][j] = 0.f;
    float bsum = 0.f;
    const float it_l2e = a.inv_temp * LOG2E, cb = bias_v * it_l2e;

    auto stage = [&](int k) {
      char* buf = smem + (k & 1) * BUF_BYTES;
      const int t0 = (ri0 + k) * 32;
      stage_tile(a.h, a.ldh, t0, a.T - 1, buf, w, lane);
      const int t = min(t0 + r32, a.T - 1);
      if (w == 0)        // aux: lse[32] | gs[32] | target[32]
        __builtin_amdgcn_global_load_lds((gptr_t)(hh ? a.gs + t : a.lse + t), (lptr_t)(buf + TILE_BYTES), 4, 0, 0);
      if (w == 1)
        __builtin_amdgcn_global_load_lds((gptr_t)(a.target + t), (lptr_t)(buf + TILE_BYTES + 256), 4, 0, 0);
    };
    stage(0);
    PV_STAMP(0)
    for (int k = 0; k < nri; ++k) {
      __syncthreads();
      PV_STAMP(1)
      if (k + 1 < nri) stage(k + 1);
      PV_STAMP(2)
      const char* tile = smem + (k & 1) * BUF_BYTES;
      const float* lse_t = (const float*)(tile + TILE_BYTES);
      const float* gs_t = lse_t + 32;
      const int* tgt_t = (const int*)(tile + TILE_BYTES + 256);
      const int t0 = (ri0 + k) * 32;
      // ---- S block [32 t][32 v]: rows of h (LDS) x this wave's W fragments ---------------------------------------------
      f32x16 X;
#pragma unroll
      for (int j = 0; j < 16; ++j) X[j] = 0.f;
      first_product(tile, rbase, swh, wf, X);
      PV_STAMP(3)
      // ---- P = gs (softmax - onehot); its column sums are the bias gradient.  exp(s - lse) = 2^(X it_l2e + cb - lse log2e).
      // The row mask (rows past the live count: the last item only) and the onehot (a target inside this wave's 32 columns:
      // one item in twenty) sit behind a wave-uniform branch.  Columns past V need no mask: nothing reads their Z rows.
      bf16x8 pf[2];
      const bool edge = t0 + 32 > nlive;
      const bool hit = __any((unsigned)(tgt_t[r32] - (ct * UNIT + 32 * w)) < 32u);
      if (__builtin_expect(edge || hit, 0)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 l4 = *(const f32x4*)(lse_t + 8 * g4 + 4 * hh);
          const f32x4 g_4 = *(const f32x4*)(gs_t + 8 * g4 + 4 * hh);
          const int4 y4 = *(const int4*)(tgt_t + 8 * g4 + 4 * hh);
          const int yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int tloc = 8 * g4 + 4 * hh + i;
            const float gsv = t0 + tloc < nlive ? g_4[i] * a.inv_temp : 0.f;
            // (the clamp keeps 2^x finite where lse is stale, e.g. the 0 of a dead row: its gsv is 0 and 0 * inf is NaN)
            float p = __builtin_amdgcn_exp2f(fminf(fmaf(X[4 * g4 + i], it_l2e, fmaf(l4[i], -LOG2E, cb)), 126.f));
            p = (yy[i] == v_mine ? p - 1.f : p) * gsv;
            bsum += p;
            pf[(4 * g4 + i) >> 3][(4 * g4 + i) & 7] = (__bf16)p;
          }
        }
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 l4 = *(const f32x4*)(lse_t + 8 * g4 + 4 * hh);
          const f32x4 g_4 = *(const f32x4*)(gs_t + 8 * g4 + 4 * hh);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float p = __builtin_amdgcn_exp2f(fminf(fmaf(X[4 * g4 + i], it_l2e, fmaf(l4[i], -LOG2E, cb)), 126.f)) *
                            (g_4[i] * a.inv_temp);
            bsum += p;
            pf[(4 * g4 + i) >> 3][(4 * g4 + i) & 7] = (__bf16)p;
          }
        }
      }
      // ---- Z[v][hc] += P^T[v][t] h[t][hc]: the probabilities x columns of the same h tile -----------------------------
      PV_STAMP(4)
      second_product<false>(tile, tr, 0, pf[0], Z);
      second_product<false>(tile, tr, 1, pf[1], Z);
      PV_STAMP(5)
    }
    const long slab = (long)rid * a.max_seg + seg;
    float* z = a.z_slab + slab * UNIT * HD + r32;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) z[(long)(32 * w + (j & 3) + 8 * (j >> 2) + 4 * hh) * HD + 32 * c] = Z[c][j];
    bsum += xor32(bsum);
    if (hh == 0) a.b_slab[slab * UNIT + 32 * w + r32] = bsum;
    __syncthreads();
    PV_STAMP(6)
  }
  PV_STAMP_FLUSH(1)
}

struct DwReduceArgs {
  const float* z_slab;
  const float* b_slab;
  const int* live;
  int T, V, max_seg, G, lddw, accumulate, accumulate_bias;
  float* dW;            // [V][lddw]
  float* dbias;         // [V]
};

__global__ __launch_bounds__(256) void vocab_dw_reduce_kernel(DwReduceArgs a) {
  const int ct = blockIdx.x, tid = threadIdx.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NCT = (a.V + UNIT - 1) / UNIT, NRI = (nlive + 31) / 32;
  const long N = (long)NCT * NRI, x0 = (long)ct * NRI, x1 = x0 + NRI;
  int w0 = 0, w1 = 0;
  if (N > 0) {
    w0 = (int)(x0 * a.G / N);
    while (w0 > 0 && seg_beg(w0, N, a.G) > x0) --w0;
    while (seg_beg(w0 + 1, N, a.G) <= x0) ++w0;
    w1 = w0 + 1;
    while (w1 < a.G && seg_beg(w1, N, a.G) < x1) ++w1;
  }
  auto slab_of = [&](int w) -> long {
    const long b = seg_beg(w, N, a.G), e = seg_beg(w + 1, N, a.G);
    return (e <= x0 || e == b) ? -1 : (long)w * a.max_seg + (ct - (int)(b / NRI));
  };
  // rows blockIdx.y * 16 .. + 16 of the tile, 4 columns per thread and step; segments in workgroup order
  for (int i = tid; i < 16 * (HD / 4); i += 256) {
    const int vl = blockIdx.y * 16 + i / (HD / 4), c4 = (i % (HD / 4)) * 4, v = ct * UNIT + vl;
    if (v >= a.V) continue;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int w = w0; w < w1; ++w) {
      const long sl = slab_of(w);
      if (sl >= 0) acc += *(const f32x4*)(a.z_slab + (sl * UNIT + vl) * HD + c4);
    }
    f32x4* dst = (f32x4*)(a.dW + (long)v * a.lddw + c4);
    *dst = a.accumulate ? *dst + acc : acc;
  }
  if (tid < 16) {
    const int vl = blockIdx.y * 16 + tid, v = ct * UNIT + vl;
    if (v < a.V) {
      float acc = 0.f;
      for (int w = w0; w < w1; ++w) {
        const long sl = slab_of(w);
        if (sl >= 0) acc += a.b_slab[sl * UNIT + vl];
      }
      a.dbias[v] = a.accumulate_bias ? a.dbias[v] + acc : acc;
    }
  }
}

// dst[perm[new]][:] = new < count ? src[new][:] * scale[new] : 0
__global__ void scatter_rows_scaled_kernel(const float* __restrict__ src, const float* __restrict__ scale, float factor,
                                           const int64_t* __restrict__ perm, const int* __restrict__ count, int T, int cols4,
                                           float* __restrict__ dst, long ldd) {
  const long n = (long)T * cols4;
  const int live = count ? *count : T;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cols4), c = (int)(i % cols4) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < live) v = *(const f32x4*)(src + (long)row * cols4 * 4 + c) * (scale[row] * factor);
    *(f32x4*)(dst + (perm ? perm[row] : row) * ldd + c) = v;
  }
}

// the same with the backward of a dropout on the destination folded in: destination row r = i0 * n1p + i1 holds the
// logical elements i0 * m0 + i1 * m1 + c (pv_dropout_f32's index convention); rows with i1 >= n1 are padding
struct RowDrop {
  const uint8_t* mask;
  const RngState* rng;
  uint32_t site, thresh;
  float scale;
  int n1, n1p;
  long m0, m1;
};

__global__ void scatter_rows_scaled_drop_kernel(const float* __restrict__ src, const float* __restrict__ scale, float factor,
                                                const int64_t* __restrict__ perm, const int* __restrict__ count, int T, int cols4,
                                                float* __restrict__ dst, long ldd, RowDrop d) {
  const long n = (long)T * cols4;
  const int live = count ? *count : T;
  RngState st{0, 0};
  if (!d.mask) st = *d.rng;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cols4), c = (int)(i % cols4) * 4;
    const long drow = perm ? perm[row] : row;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int i0 = (int)(drow / d.n1p), i1 = (int)(drow % d.n1p);
    if (row < live && i1 < d.n1) {
      const f32x4 x = *(const f32x4*)(src + (long)row * cols4 * 4 + c) * (scale[row] * factor);
      v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      drop4_at(d.mask, st, d.site, d.thresh, d.scale, i0 * d.m0 + i1 * d.m1 + c, v);
    }
    *(f32x4*)(dst + drow * ldd + c) = f32x4{v[0], v[1], v[2], v[3]};
  }
}

int grid_wgs() {
  static int g = 0;
  if (!g) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    g = cus;
  }
  return g;
}

int max_seg_fwd(int T, int V, int G) {
  const long NVI = (V + 31) / 32, NRG = (T + UNIT - 1) / UNIT;
  const long per = (NRG * NVI + G - 1) / G;
  return (int)(per / NVI) + 2;
}
int max_seg_dw(int V, int G) { return ((V + UNIT - 1) / UNIT + G - 1) / G + 1; }

int check_common(const void* h, int ldh, const void* W, int ldw, int T, int V, int H, const char* who) {
  PV_REQUIRE(h && W && T > 0 && V > 0, "%s: null / empty operand", who);
  PV_REQUIRE(H == HD, "%s: built for H = %d (got %d); use the tile-GEMM path", who, HD, H);
  PV_REQUIRE(ldh >= HD && ldw >= HD && ldh % 8 == 0 && ldw % 8 == 0, "%s: leading dimensions must be multiples of 8 and >= %d", who, HD);
  PV_REQUIRE(((uintptr_t)h & 15) == 0 && ((uintptr_t)W & 15) == 0, "%s: operands must be 16-byte aligned", who);
  return PV_OK;
}

template <class K>
int set_smem(K kernel) {
  PV_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
  return PV_OK;
}

}  // namespace

extern "C" {

#ifdef PV_VOCAB_STAMPS
int pv_vocab_stamps_copy(void* host_dst, size_t bytes) {
  PV_HIP(hipDeviceSynchronize());
  PV_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_vstamps), bytes < sizeof(g_vstamps) ? bytes : sizeof(g_vstamps)));
  return PV_OK;
}
#endif

int pv_vocab_fused_supported(int H) { return H == HD ? 1 : 0; }

size_t pv_vocab_fused_fwd_ws_bytes(int T, int V) {
  const int G = grid_wgs();
  const size_t slabs = (size_t)G * max_seg_fwd(T, V, G);
  return slabs * ((size_t)HD * UNIT * sizeof(float) + UNIT * sizeof(float4));
}

int pv_vocab_fused_fwd(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, int T, int V,
                       int H, float inv_temp, int64_t ignore_index, void* workspace, float* lse, float* nll, float* U,
                       const int* live_rows, void* stream) {
  int rc = check_common(h, ldh, W, ldw, T, V, H, "pv_vocab_fused_fwd");
  if (rc) return rc;
  PV_REQUIRE(bias && target32 && workspace && lse && nll && U, "pv_vocab_fused_fwd: null argument");
  PV_REQUIRE(((uintptr_t)U & 15) == 0 && ((uintptr_t)workspace & 15) == 0, "pv_vocab_fused_fwd: U / workspace must be 16-byte aligned");
  static bool attr = false;
  if (!attr) {
    rc = set_smem(vocab_fwd_dh_kernel);
    if (rc) return rc;
    attr = true;
  }
  const int G = grid_wgs(), ms = max_seg_fwd(T, V, G);
  float* o_slab = (float*)workspace;
  float4* stats = (float4*)(o_slab + (size_t)G * ms * HD * UNIT);
  hipStream_t st = (hipStream_t)stream;
  FwdArgs a{(const bf16_t*)h, (const bf16_t*)W, bias, target32, live_rows, T, V, ldh, ldw, ms, inv_temp, o_slab, stats};
  hipLaunchKernelGGL(vocab_fwd_dh_kernel, dim3(G), dim3(256), SMEM_BYTES, st, a);
  PV_LAUNCH_CHECK("vocab_fwd_dh_kernel");
  FwdCombineArgs c{o_slab, stats, (const bf16_t*)W, target32, live_rows, T, V, ldw, ms, G, (long)ignore_index, lse, nll, U};
  hipLaunchKernelGGL(vocab_fwd_combine_kernel, dim3(cdiv(T, UNIT), HD / 32), dim3(256), 0, st, c);
  PV_LAUNCH_CHECK("vocab_fwd_combine_kernel");
  return PV_OK;
}

size_t pv_vocab_fused_dw_ws_bytes(int V) {
  const int G = grid_wgs();
  return (size_t)G * max_seg_dw(V, G) * ((size_t)UNIT * HD + UNIT) * sizeof(float);
}

int pv_vocab_fused_dw(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, const float* lse,
                      const float* gscale, int T, int V, int H, float inv_temp, void* workspace, float* dW, int lddw,
                      int accumulate, float* dbias, const int* live_rows, void* stream) {
  return pv_vocab_fused_dw_wgs(h, ldh, W, ldw, bias, target32, lse, gscale, T, V, H, inv_temp, workspace, dW, lddw, accumulate,
                               dbias, 0, live_rows, 0, stream);
}

int pv_vocab_fused_dw_wgs(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, const float* lse,
                          const float* gscale, int T, int V, int H, float inv_temp, void* workspace, float* dW, int lddw,
                          int accumulate, float* dbias, int accumulate_bias, const int* live_rows, int wgs, void* stream) {
  int rc = check_common(h, ldh, W, ldw, T, V, H, "pv_vocab_fused_dw");
  if (rc) return rc;
  PV_REQUIRE(bias && target32 && lse && gscale && workspace && dW && dbias, "pv_vocab_fused_dw: null argument");
  PV_REQUIRE(lddw >= HD && lddw % 4 == 0 && ((uintptr_t)dW & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
             "pv_vocab_fused_dw: dW must be 16-byte aligned with lddw a multiple of 4 and >= %d", HD);
  static bool attr = false;
  if (!attr) {
    rc = set_smem(vocab_dw_kernel);
    if (rc) return rc;
    attr = true;
  }
  PV_REQUIRE(wgs >= 0, "pv_vocab_fused_dw_wgs: wgs = %d", wgs);
  const int G = wgs > 0 && wgs < grid_wgs() ? wgs : grid_wgs(), ms = max_seg_dw(V, G);
  PV_REQUIRE((size_t)G * ms * ((size_t)UNIT * HD + UNIT) * sizeof(float) <= pv_vocab_fused_dw_ws_bytes(V),
             "pv_vocab_fused_dw_wgs: %d workgroups need more than pv_vocab_fused_dw_ws_bytes", G);
  float* z_slab = (float*)workspace;
  float* b_slab = z_slab + (size_t)G * ms * UNIT * HD;
  hipStream_t st = (hipStream_t)stream;
  DwArgs a{(const bf16_t*)h, (const bf16_t*)W, bias, target32, lse, gscale, live_rows, T, V, ldh, ldw, ms, inv_temp, z_slab, b_slab};
  hipLaunchKernelGGL(vocab_dw_kernel, dim3(G), dim3(256), SMEM_BYTES, st, a);
  PV_LAUNCH_CHECK("vocab_dw_kernel");
  DwReduceArgs r{z_slab, b_slab, live_rows, T, V, ms, G, lddw, accumulate, accumulate_bias, dW, dbias};
  hipLaunchKernelGGL(vocab_dw_reduce_kernel, dim3(cdiv(V, UNIT), UNIT / 16), dim3(256), 0, st, r);
  PV_LAUNCH_CHECK("vocab_dw_reduce_kernel");
  return PV_OK;
}

int pv_scatter_rows_scaled_f32(const float* src, const float* scale, float factor, const int64_t* perm, const int* count, int T,
                               int cols, float* dst, long ldd, void* stream) {
  PV_REQUIRE(src && scale && dst && T > 0 && cols > 0 && cols % 4 == 0 && ldd >= cols && ldd % 4 == 0,
             "pv_scatter_rows_scaled_f32: bad argument (cols and ldd must be multiples of 4)");
  PV_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pv_scatter_rows_scaled_f32: buffers must be 16-byte aligned");
  const long n = (long)T * (cols / 4);
  const long gx = (n + 255) / 256;
  hipLaunchKernelGGL(scatter_rows_scaled_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream, src,
                     scale, factor, perm, count, T, cols / 4, dst, ldd);
  PV_LAUNCH_CHECK("scatter_rows_scaled_kernel");
  return PV_OK;
}

int pv_scatter_rows_scaled_drop_f32(const float* src, const float* scale, float factor, const int64_t* perm, const int* count,
                                    int T, int cols, float* dst, long ldd, int n1, int n1p, long m0, long m1, float p,
                                    const void* mask, const void* rng_state, int site, void* stream) {
  PV_REQUIRE(src && scale && dst && T > 0 && cols > 0 && cols % 4 == 0 && ldd >= cols && ldd % 4 == 0,
             "pv_scatter_rows_scaled_drop_f32: bad argument (cols and ldd must be multiples of 4)");
  PV_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pv_scatter_rows_scaled_drop_f32: buffers must be 16-byte aligned");
  PV_REQUIRE(n1 > 0 && n1p >= n1 && T % n1p == 0, "pv_scatter_rows_scaled_drop_f32: T=%d rows are not whole groups of n1p=%d", T, n1p);
  PV_REQUIRE(p >= 0.f && p < 1.f && (mask || rng_state), "pv_scatter_rows_scaled_drop_f32: p in [0, 1) and a keep mask or a generator state");
  RowDrop d{(const uint8_t*)mask, (const RngState*)rng_state, (uint32_t)site, drop_threshold(p), 1.f / (1.f - p), n1, n1p, m0, m1};
  const long n = (long)T * (cols / 4);
  const long gx = (n + 255) / 256;
  hipLaunchKernelGGL(scatter_rows_scaled_drop_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream,
                     src, scale, factor, perm, count, T, cols / 4, dst, ldd, d);
  PV_LAUNCH_CHECK("scatter_rows_scaled_drop_kernel");
  return PV_OK;
}

}  // extern "C"
