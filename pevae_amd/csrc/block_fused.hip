// One launch per Block forward, three per Block backward (dialog_models.py:253-271).
//
// A Block is  out = translate(LayerNorm(x + c4(gelu(c3(gelu(c2(gelu(c1(gelu(x))))))))))  with five spectral-norm
// linears.  At batch 256 every one of its GEMMs is a few hundred kFLOP..MFLOP: run as separate launches the step
// spends its time in launch boundaries and dependent memory round trips (~600 launches of 3-10 us for the whole DLS
// latent stack).  Every operation of the Block is ROW-independent once 1/sigma of each weight is known (the
// multi-tensor power iteration runs before, pv_spectral_norm_multi), so a workgroup that owns 16 batch rows can walk
// the whole Block without talking to any other workgroup: activations stay in LDS, weights stream from L2 straight
// into MFMA B-operand registers (16 workgroups share them), and the only global traffic besides the weights is the
// activations the backward pass needs.
//
//   block_fwd_kernel   : x -> out, saving a0 = gelu(x), y1, a1, y2, a2, y3, a3, y4, o = LN(x + y4), mean, rstd
//   block_bwd_dx_kernel: dL/dout -> dL/dx, storing dy1..dy4 for the weight gradients and, per workgroup, partial
//                        column sums (bias / LayerNorm parameter gradients) and partial <dy, y - b> (spectral term)
//   (+ pv_linear_f32_group: the five weight-gradient GEMMs dy^T a with the spectral rank-1 epilogue, one launch; the
//    fixed-order sums of the per-workgroup partials into the bias / gamma / beta gradients ride on the same launch as
//    ColReduceJob riders in extra workgroups)
//
// Arithmetic: exact fp32 (v_mfma_f32_16x16x4_f32), like the unfused path; sums run in a different order.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "linear_group.h"
#include "../../include/pevae_hip.h"

namespace {

constexpr int RB = 16;          // batch rows per workgroup = one MFMA row tile
constexpr int NTC = 256;        // compute threads: waves 0-3
constexpr int NT = 320;         // + wave 4, the storer

struct BlockDesc {
  const float* W[5];            // c1, c2, c3, c4, translate: weight_orig [out][in] row-major
  const float* bias[5];
  const float* inv_sigma[5];    // device scalars (1 / sigma)
  const float* gamma;
  const float* beta;
  int Din, mid, Dout;
  float eps;
};

// erf to ~1 ulp, branch-free (both ranges are evaluated and selected): at one wave per SIMD a library erff is a
// ~100-deep dependent chain with data-dependent branches, which made GELU the largest cost of these kernels; straight-
// line code lets the compiler interleave the chains of a lane's independent elements.  Two minimax polynomials
// (|x| <= 0.927734375: odd polynomial in x; above: 1 - exp(p(|x|))), the split and degrees of the widely used
// single-precision erf of N. Juffa.
__device__ __forceinline__ float erf_fast(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  r = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  q = fmaf(q, a, a);
  return t > 0.927734375f ? r : q;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
  return 0.5f * (1.f + erf_fast(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for this wave's global stores (the saved
// activations), a write round trip per layer that nothing here depends on.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// sum over the 16 lanes that share lane >> 4 (every lane gets the result)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Diagnostic build only (-DPV_BLOCK_STAMPS, `make stamps`): wave 0 of workgroup 0 records the wall clock (100 MHz) at
// phase boundaries; read back with pv_block_stamps_read.  Never compiled into the shipped library.
#ifdef PV_BLOCK_STAMPS
__device__ unsigned long long g_block_stamps[2][32];
#define BSTAMP_DECL(ptr) unsigned long long* bst = (unsigned long long*)(ptr);
#define BSTAMP(k, i)                                                  \
  do {                                                                \
    if (k == 0 && threadIdx.x == 0) bst[i] = wall_clock64();          \
  } while (0)
#define BSTAMP_FLUSH(k)                                                                 \
  do {                                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0)                                            \
      for (int i__ = 0; i__ < 16; ++i__) g_block_stamps[k][i__] = bst[i__];             \
  } while (0)
#else
#define BSTAMP_DECL(ptr)
#define BSTAMP(k, i)
#define BSTAMP_FLUSH(k)
#endif

// offsets (floats per row) of the saved activations
struct SaveLayout {
  int a0, y1, a1, y2, a2, y3, a3, y4, o, width;
  __host__ __device__ SaveLayout(int Din, int mid) {
    a0 = 0; y1 = Din; a1 = 2 * Din; y2 = 3 * Din; a2 = y2 + mid; y3 = a2 + mid; a3 = y3 + mid; y4 = a3 + mid;
    o = y4 + Din; width = o + Din;
  }
};
// dY row: dy1 (Din) | dy2 (mid) | dy3 (mid) | dy4 (Din)
struct GradLayout {
  int d1, d2, d3, d4, width;
  __host__ __device__ GradLayout(int Din, int mid) { d1 = 0; d2 = Din; d3 = Din + mid; d4 = Din + 2 * mid; width = 2 * Din + 2 * mid; }
};
// per-workgroup partial row: db1 (Din) | db2 (mid) | db3 (mid) | db4 (Din) | dbt (Dout) | dgamma (Din) | dbeta (Din)
struct PartLayout {
  int b1, b2, b3, b4, bt, dg, dbe, width;
  __host__ __device__ PartLayout(int Din, int mid, int Dout) {
    b1 = 0; b2 = Din; b3 = b2 + mid; b4 = b3 + mid; bt = b4 + Din; dg = bt + Dout; dbe = dg + Din; width = dbe + Din;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// One layer of the calling wave:  out[16][N] = in[16][K] * Wop  with, FWD: Wop[k][n] = W[n*K + k] (y = a W^T), else
// Wop[k][n] = W[k*N + n] (dx = dy W).  A wave works on 64-column super-tiles n0 = 64*w, 64*(w+4), ...; the lane
// (c16, q4) owns columns n0 + 4*c16 + e (e = 0..3, one accumulator each) and, inside a 16-deep k chunk, supplies
// k = 4*q4 + j at MFMA step j for BOTH operands (a k permutation common to A and B), so every operand access is one
// 16-byte load:
//   FWD: p[e] + ch*16       -> W[col e][16*ch + 4*q4 .. +3]          (element j = step j)
//   BWD: p[j] + ch*16*N     -> W[16*ch + 4*q4 + j][n0 + 4*c16 .. +3] (element e = accumulator e)
// Columns past N use clamped (valid) addresses; their results are dropped by the epilogue.
//
// Weights stream global -> VGPR through a ring of RD chunks: the slot a chunk's 16 MFMAs have just read is refilled
// with the chunk RD ahead (clamped to the last one: no load sits under a condition), so ~4000 MFMA cycles of loads are
// in flight (measured: with 4 chunks the layers ran at 2.2x their MFMA time, bound by HBM latency x bytes in flight).  After the last k loop of a layer the ring is refilled with the NEXT layer's first chunks, whose
// latency then hides behind the epilogue and the barrier.  The code is kept SMALL on purpose (one copy of this function
// per kernel, layers walked by a run-time loop): a first version with one inlined copy per layer was instruction-fetch
// bound (a workgroup runs through its code once).
// ---------------------------------------------------------------------------------------------------------------
#ifndef PV_BLOCK_ABL
#define PV_BLOCK_ABL 0           // timing ablations of the diagnostic build: 1 = no weight refills, 2 = no MFMAs
#endif
constexpr int RD = 4;            // ring depth in chunks: 16 loads of 16 bytes per lane in flight (8 measured no faster)

// Where a lane finds its B-operand values: Wop[k][n] = W[k*N + n] (a [K][N] row-major matrix: the weight itself for
// dx = dy W, its transposed copy for y = a W^T).  p[j] + ch*16*N -> W[16*ch + 4*q4 + j][n0 + 4*c16 .. +3]: a 16-lane
// group reads 256 contiguous bytes (element e = accumulator e).
struct WPtr {
  const float* p[4];
  long step;
};

__device__ __forceinline__ WPtr wptr(const float* __restrict__ W, int N, int n0, int c16, int q4) {
  WPtr s;
#pragma unroll
  for (int j = 0; j < 4; ++j) s.p[j] = W + (size_t)(4 * q4 + j) * N + min(n0 + 4 * c16, N - 4);
  s.step = (long)16 * N;
  return s;
}

__device__ __forceinline__ void ring_fill(const WPtr& s, int nch, f32x4 (&ring)[RD][4]) {
#pragma unroll
  for (int t = 0; t < RD; ++t) {
    const long o = (long)min(t, nch - 1) * s.step;
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[t][i] = *(const f32x4*)(s.p[i] + o);
  }
}

// One layer of the calling compute wave:  out[16][N] = in[16][K] * Wop,  Wop [K][N] row-major.  `in` is the LDS image
// [16][LD] of the layer input.  A wave works on 64-column super-tiles n0 = 64*w, 64*(w+4), ...; the lane (c16, q4) owns
// columns n0 + 4*c16 + e (e = 0..3, one accumulator each) and, inside a 16-deep k chunk, supplies k = 4*q4 + j at MFMA
// step j for BOTH operands (a k permutation common to A and B), so every operand access is one 16-byte load.  Columns
// past N use clamped (valid) addresses; their results are dropped by the epilogue.
//
// Weights stream global -> VGPR through a ring of RD chunks: the slot a chunk's 16 MFMAs have just read is refilled
// with the chunk RD ahead (clamped to the last one: no load sits under a condition).  On entry the ring holds chunks
// 0..RD-1 of super-tile n0 = 64*w (requested by the previous layer / the prologue); after the last k loop it is
// refilled with the NEXT layer's first chunks (Wn, Nn, Kn; Wn == nullptr: none), whose latency hides behind the
// epilogue and the barrier.  pre(n0) runs before a super-tile's k loop (issue the epilogue's global loads there);
// epi(acc, n0): acc[e][j] = out[row 4*q4 + j][col n0 + 4*c16 + e].
// The code is kept SMALL on purpose (one copy per kernel, layers walked by a run-time loop).
template <class Pre, class Epi>
__device__ __forceinline__ void dense16(const float* __restrict__ in, const int LD, const float* __restrict__ W, const int N,
                                        const int K, const float* __restrict__ Wn, const int Nn, const int Kn, const int w,
                                        const int lane, f32x4 (&ring)[RD][4], Pre pre, Epi epi) {
  const int c16 = lane & 15, q4 = lane >> 4;
  const float* ap = in + c16 * LD + 4 * q4;
  const int nch = K / 16;
  for (int n0 = w * 64; n0 < N; n0 += 256) {
    const WPtr s = wptr(W, N, n0, c16, q4);
    if (n0 != w * 64) ring_fill(s, nch, ring);
    pre(n0);
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av = *(const f32x4*)ap;
    for (int ch = 0; ch < nch; ch += RD) {
#pragma unroll
      for (int t = 0; t < RD; ++t) {
        const int c = ch + t;
        const f32x4 avn = *(const f32x4*)(ap + min(c + 1, nch - 1) * 16);
        if (c < nch) {
#if PV_BLOCK_ABL == 2
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ring[t][i]), "v"(av));
#else
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], ring[t][j][e], acc[e], 0, 0, 0);
#endif
        }
#if PV_BLOCK_ABL != 1
        const long o = (long)min(c + RD, nch - 1) * s.step;
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[t][i] = *(const f32x4*)(s.p[i] + o);
#endif
        av = avn;
      }
    }
    if (n0 + 256 >= N && Wn) ring_fill(wptr(Wn, Nn, w * 64, c16, q4), Kn / 16, ring);
    epi(acc, n0);
  }
  if (w * 64 >= N && Wn) ring_fill(wptr(Wn, Nn, w * 64, c16, q4), Kn / 16, ring);   // idle wave this layer
}

// ---- the storer wave's helpers (wave 4: every global store of saved state leaves from here, so that no compute wave
// ever has a slow store queued in front of a weight load it waits for: vmcnt retires in issue order) ----------------
__device__ __forceinline__ void wave_copy_out(const float* __restrict__ buf, int LD, float* __restrict__ g, size_t gs,
                                              int ncols, int r0, int M, int lane) {
  const int nq = RB * ncols / 4, per = ncols / 4;
  for (int i = lane; i < nq; i += 64) {
    const int row = i / per, c4 = (i % per) * 4;
    if (r0 + row < M) *(f32x4*)(g + (size_t)(r0 + row) * gs + c4) = *(const f32x4*)(buf + row * LD + c4);
  }
}
// column sums over the workgroup's 16 rows of an LDS image (rows past M hold zeros)
__device__ __forceinline__ void wave_colsum(const float* __restrict__ buf, int LD, int N, float* __restrict__ dst, int lane) {
  for (int c = lane * 4; c < N; c += 256) {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RB; ++r) sum += *(const f32x4*)(buf + r * LD + c);
    *(f32x4*)(dst + c) = sum;
  }
}
// biases / gamma / beta -> LDS (PartLayout offsets), all loads of the wave issued before the first LDS write
__device__ __forceinline__ void wave_stage_params(const BlockDesc& d, const PartLayout& P, float* par, int lane) {
  const int nq = P.width / 4;                     // <= 704 for widths <= 512
  const float* src[7] = {d.bias[0], d.bias[1], d.bias[2], d.bias[3], d.bias[4], d.gamma, d.beta};
  const int beg[8] = {P.b1, P.b2, P.b3, P.b4, P.bt, P.dg, P.dbe, P.width};
  f32x4 v[11];
#pragma unroll
  for (int u = 0; u < 11; ++u) {
    const int c = min(lane + 64 * u, nq - 1) * 4;
    const float* ptr = src[0] + c;
#pragma unroll
    for (int k = 1; k < 7; ++k)
      if (c >= beg[k]) ptr = src[k] + (c - beg[k]);
    v[u] = *(const f32x4*)ptr;
  }
#pragma unroll
  for (int u = 0; u < 11; ++u) {
    const int i = lane + 64 * u;
    if (i < nq) *(f32x4*)(par + i * 4) = v[u];
  }
  if (lane < 5) par[P.width + lane] = d.inv_sigma[lane][0];
}

// ---------------------------------------------------------------------------------------------------------------
// Single-workgroup slabs (every Block of the launch narrower than 512): 64-column wave tiles, a fifth *storer* wave that
// writes the state the backward pass needs one layer behind, no hand-offs.  (Round-1 kernel; the cluster kernel below
// also walks G = 1 Blocks, but its 32-column tiles and its LDS staging cost the small Blocks ~10 %.)
// ---------------------------------------------------------------------------------------------------------------
struct BlockFwdArgs1 {
  const float* x;
  long ldx;
  int M;
  BlockDesc d;      // d.W: the TRANSPOSED effective-layout copies W_i^T [in][out] (from pv_spectral_norm_multi)
  float* save;      // [M][SaveLayout.width]
  float* mean;      // [M]
  float* rstd;      // [M]
  float* out;       // [M][ldo]
  long ldo;
  int LD;           // LDS row stride (floats) = max width + 4
};

// Independent Blocks (the two pyramids' Blocks of one level, the three Blocks of a DLS_Auto level) share a launch:
// blockIdx.y picks the Block, blockIdx.x its 16-row slab.  Each Block alone occupies 16 of the 256 CUs.
struct BlockFwdGroup1 {
  BlockFwdArgs1 a[PV_BLOCK_GROUP_MAX];
};

__global__ __launch_bounds__(NT) void block_fwd_kernel(BlockFwdGroup1 grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BlockFwdArgs1& a = grp.a[blockIdx.y];
  if ((int)blockIdx.x * RB >= a.M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, q4 = lane >> 4;
  const bool compute = w < 4;
  const int Din = a.d.Din, mid = a.d.mid, Dout = a.d.Dout, LD = a.LD, M = a.M;
  const int r0 = blockIdx.x * RB;
  float* buf0 = lds;
  float* buf1 = lds + RB * LD;
  float* yb0 = lds + 2 * RB * LD;                 // the layer's pre-activation y, parked for the storer wave
  float* yb1 = lds + 3 * RB * LD;
  float* par = lds + 4 * RB * LD;
  const SaveLayout L(Din, mid);
  const PartLayout P(Din, mid, Dout);
  const float* isg = par + P.width;
  float* rowstat = par + P.width + 8;             // mean[16], rstd[16]
  const size_t SW = L.width;
  BSTAMP_DECL(par + P.width + 8 + 2 * RB)
  BSTAMP(0, 0);
#ifdef PV_BLOCK_STAMPS
  if (threadIdx.x == 0) bst[14] = clock64();
#endif

  f32x4 ring[RD][4];
  if (compute) {
    ring_fill(wptr(a.d.W[0], Din, w * 64, c16, q4), Din / 16, ring);
    // x rows: a0 = gelu(x) -> buf0 (all of a thread's loads are issued first: widths <= 512 -> at most 8 per thread)
    const int nq = RB * Din / 4, per = Din / 4;
    f32x4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(tid + u * NTC, nq - 1);
      xv[u] = *(const f32x4*)(a.x + (size_t)min(r0 + i / per, M - 1) * a.ldx + (i % per) * 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * NTC;
      if (i < nq) {
        const f32x4 v = xv[u];
        *(f32x4*)(buf0 + (i / per) * LD + (i % per) * 4) = f32x4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
      }
    }
  } else {
    wave_stage_params(a.d, P, par, lane);
  }
  lds_barrier();
  BSTAMP(0, 1);

  // layers c1, c2, c3 (y -> yb, gelu(y) -> next input), c4 (y -> yb, x + y -> LayerNorm input), translate (-> out);
  // the storer wave runs one layer behind: it writes out what the previous layer left in LDS.
  for (int li = 0; li < 5; ++li) {
    const int N = li == 0 || li == 3 ? Din : (li == 4 ? Dout : mid);
    const int K = li == 0 || li == 1 || li == 4 ? Din : mid;
    const int Nn = li == 0 || li == 1 ? mid : (li == 2 ? Din : Dout), Kn = li == 0 || li == 3 ? Din : mid;
    const float* in = li == 4 ? buf1 : ((li & 1) ? buf1 : buf0);
    float* outb = (li & 1) ? buf0 : buf1;
    float* yb = (li & 1) ? yb1 : yb0;
    if (compute) {
      const float* Wn = li < 4 ? a.d.W[li + 1] : nullptr;
      const int off_b = li == 0 ? P.b1 : li == 1 ? P.b2 : li == 2 ? P.b3 : li == 3 ? P.b4 : P.bt;
      const float is = isg[li];
      f32x4 xs[4];
      dense16(in, LD, a.d.W[li], N, K, Wn, Nn, Kn, w, lane, ring,
              [&](int n0) __attribute__((always_inline)) {
                if (li != 3) return;                  // the residual input, requested before the k loop
                const int col = min(n0 + 4 * c16, N - 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  xs[j] = *(const f32x4*)(a.x + (size_t)min(r0 + q4 * 4 + j, M - 1) * a.ldx + col);
              },
              [&](f32x4 (&acc)[4], int n0) __attribute__((always_inline)) {
                const int col = n0 + 4 * c16;
                if (col >= N) return;
                const f32x4 bv = *(const f32x4*)(par + off_b + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int row = q4 * 4 + j, gr = r0 + row;
                  const f32x4 y = {acc[0][j] * is + bv[0], acc[1][j] * is + bv[1], acc[2][j] * is + bv[2],
                                   acc[3][j] * is + bv[3]};
                  if (li < 3) {
                    *(f32x4*)(yb + row * LD + col) = y;
                    *(f32x4*)(outb + row * LD + col) = f32x4{gelu_f(y[0]), gelu_f(y[1]), gelu_f(y[2]), gelu_f(y[3])};
                  } else if (li == 3) {
                    *(f32x4*)(yb + row * LD + col) = y;
                    *(f32x4*)(buf0 + row * LD + col) = xs[j] + y;
                  } else if (gr < M) {
                    *(f32x4*)(a.out + (size_t)gr * a.ldo + col) = y;
                  }
                }
              });
    } else {
      if (li == 0) {
        wave_copy_out(buf0, LD, a.save + L.a0, SW, Din, r0, M, lane);
      } else if (li < 4) {
        // outputs of layer li - 1: y in the other yb, gelu(y) in this layer's input
        const int pN = li == 1 ? Din : mid;
        const int off_y = li == 1 ? L.y1 : li == 2 ? L.y2 : L.y3, off_a = li == 1 ? L.a1 : li == 2 ? L.a2 : L.a3;
        wave_copy_out((li & 1) ? yb0 : yb1, LD, a.save + off_y, SW, pN, r0, M, lane);
        wave_copy_out(in, LD, a.save + off_a, SW, pN, r0, M, lane);
      } else {
        wave_copy_out(buf1, LD, a.save + L.o, SW, Din, r0, M, lane);
        if (lane < RB && r0 + lane < M) {
          a.mean[r0 + lane] = rowstat[lane];
          a.rstd[r0 + lane] = rowstat[RB + lane];
        }
      }
    }
    BSTAMP(0, 2 + 2 * li);
    if (li == 4) break;
    lds_barrier();
    BSTAMP(0, 3 + 2 * li);
    if (li == 3) {
      if (compute) {
        // LayerNorm of s = x + y4 (buf0; biased variance, two passes as nn.LayerNorm) -> buf1: 16 lanes per row, the
        // wave's four rows side by side
        const int row = w * 4 + q4;
        const float* srow = buf0 + row * LD;
        float sm = 0.f;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 v = *(const f32x4*)(srow + c);
          sm += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float mu = group16_sum(sm) / Din;
        float q = 0.f;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 v = *(const f32x4*)(srow + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) q += (v[e] - mu) * (v[e] - mu);
        }
        const float rs = rsqrtf(group16_sum(q) / Din + a.d.eps);
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 v = *(const f32x4*)(srow + c), gm = *(const f32x4*)(par + P.dg + c), bt = *(const f32x4*)(par + P.dbe + c);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu) * rs * gm[e] + bt[e];
          *(f32x4*)(buf1 + row * LD + c) = o;
        }
        if (c16 == 0) {
          rowstat[row] = mu;
          rowstat[RB + row] = rs;
        }
      } else {
        wave_copy_out(yb1, LD, a.save + L.y4, SW, Din, r0, M, lane);
      }
      lds_barrier();
      BSTAMP(0, 12);
    }
  }
#ifdef PV_BLOCK_STAMPS
  if (threadIdx.x == 0) bst[15] = clock64();
#endif
  BSTAMP_FLUSH(0);
}

// ---------------------------------------------------------------------------------------------------------------
// Forward: a CLUSTER of G workgroups (1, 2 or 4) shares one 16-row slab and splits the columns of every layer.
//
// A slab's whole Block on one CU costs 72 us of f32 MFMA and 52 us of weight streaming for the 512-wide Blocks (every
// workgroup streams all 2.9 MB of weights through its own CU; measured with the diagnostic build's ablations), and a
// Block has only M / 16 = 16 slabs: 16 of 256 CUs busy.  With the columns of each layer split over G workgroups both
// terms divide by G; the price is that layer l + 1 needs ALL columns of layer l's output, i.e. one exchange per layer
// among the G workgroups of a slab.  The exchange rides on stores the kernel makes anyway: y_l and a_l = gelu(y_l) go
// to the `save` tensor for the backward pass; the cluster's members store their column slices write-through (sc1),
// count themselves on the slab's arrival counter and fetch the other members' slices from `save` with sc1 loads once
// the counter shows everybody (the hand-off protocol of gru.hip).  Members of a cluster sit on one XCD (blockIdx.x =
// slab + slab_stride * member, slab_stride a multiple of 8) so the hand-off stays inside one L2; placement is speed
// only.  Waves work on 32-column tiles here (two accumulators per lane), so that a member's share of a 512-wide layer
// (128 columns) still feeds all four compute waves.  Spins are bounded (wall clock) and report through g_block_err.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
constexpr int TW = 32;           // columns per wave tile (forward)
constexpr int RD2 = 8;           // ring depth in chunks: 32 loads of 8 bytes per lane in flight
constexpr int GMAX = 4;          // largest cluster
constexpr int SLAB_MAX = 64;     // slabs per Block the sync table covers (M <= 1024; larger batches run with G = 1)
constexpr unsigned long long BLOCK_SPIN_LIMIT = 200000000ull;    // 2 s of the 100 MHz wall clock

__device__ unsigned g_fwd_sync[PV_BLOCK_GROUP_MAX][SLAB_MAX][2];   // [block of the launch][slab][arrivals, departures]
__device__ unsigned g_block_err;                                   // set by a spin that gave up

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;      // cache-policy bits of the raw buffer builtins: sc1
// The hand-off moves 16 bytes per lane: an sc1 store of 8 bytes costs 2.7x a 16-byte one per byte, an 8-byte sc1 load runs
// at 0.54-0.70x the 16-byte rate (MI355X_MICROARCH, cache-policy table); a member's slice is therefore written to LDS by
// the epilogue and leaves the CU in a cooperative copy.
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ f32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX_SC1));
}

// One lane: count this workgroup on the slab's arrival counter, then wait until `want` arrivals are there.
__device__ __forceinline__ void cluster_arrive_wait(unsigned* ctr, unsigned want) {
  __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
    __builtin_amdgcn_s_sleep(1);
    if ((it & 63) == 63 && wall_clock64() - t0 > BLOCK_SPIN_LIMIT) {
      __hip_atomic_store(&g_block_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}
// One lane, after the workgroup's last wait: the last member to leave zeroes the slab's counters for the next launch
// (every member has passed its last wait by then: it counts itself out only afterwards).
__device__ __forceinline__ void cluster_depart(unsigned* ctr, unsigned G) {
  if (__hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
    __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct WPtr2 {
  const float* p[4];
  long step;
};
__device__ __forceinline__ WPtr2 wptr2(const float* __restrict__ W, int N, int n0, int c16, int q4) {
  WPtr2 s;
#pragma unroll
  for (int j = 0; j < 4; ++j) s.p[j] = W + (size_t)(4 * q4 + j) * N + min(n0 + 2 * c16, N - 2);
  s.step = (long)16 * N;
  return s;
}
__device__ __forceinline__ void ring2_fill(const WPtr2& s, int nch, f32x2 (&ring)[RD2][4]) {
#pragma unroll
  for (int t = 0; t < RD2; ++t) {
    const long o = (long)min(t, nch - 1) * s.step;
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[t][i] = *(const f32x2*)(s.p[i] + o);
  }
}

// dense16 on 32-column tiles restricted to the columns [nbeg, nend) of the layer: the wave's tiles are
// n0 = nbeg + 32 w, + 128, ...; lane (c16, q4) owns columns n0 + 2 c16 + e, e = 0, 1.  On entry the ring holds the
// first chunks of tile nbeg + 32 w; after the last k loop it is refilled for the next layer (Wn, Nn, Kn, columns from
// nbeg_n).  epi(acc, n0): acc[e][j] = out[row 4 q4 + j][col n0 + 2 c16 + e].
template <class Pre, class Epi>
__device__ __forceinline__ void dense16s(const float* __restrict__ in, const int LD, const float* __restrict__ W, const int N,
                                         const int K, const int nbeg, const int nend, const float* __restrict__ Wn,
                                         const int Nn, const int Kn, const int nbeg_n, const int w, const int lane,
                                         f32x2 (&ring)[RD2][4], Pre pre, Epi epi) {
  const int c16 = lane & 15, q4 = lane >> 4;
  const float* ap = in + c16 * LD + 4 * q4;
  const int nch = K / 16;
  for (int n0 = nbeg + w * TW; n0 < nend; n0 += 4 * TW) {
    const WPtr2 s = wptr2(W, N, n0, c16, q4);
    if (n0 != nbeg + w * TW) ring2_fill(s, nch, ring);
    pre(n0);
    f32x4 acc[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av = *(const f32x4*)ap;
    for (int ch = 0; ch < nch; ch += RD2) {
#pragma unroll
      for (int t = 0; t < RD2; ++t) {
        const int c = ch + t;
        const f32x4 avn = *(const f32x4*)(ap + min(c + 1, nch - 1) * 16);
        if (c < nch) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e)
              acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], ring[t][j][e], acc[e], 0, 0, 0);
        }
        const long o = (long)min(c + RD2, nch - 1) * s.step;
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[t][i] = *(const f32x2*)(s.p[i] + o);
        av = avn;
      }
    }
    if (n0 + 4 * TW >= nend && Wn) ring2_fill(wptr2(Wn, Nn, nbeg_n + w * TW, c16, q4), Kn / 16, ring);
    epi(acc, n0);
  }
  if (nbeg + w * TW >= nend && Wn) ring2_fill(wptr2(Wn, Nn, nbeg_n + w * TW, c16, q4), Kn / 16, ring);   // idle wave this layer
}

// the workgroup's column share [beg, end) of an LDS image -> global rows (plain stores; 8 bytes per lane)
__device__ __forceinline__ void wave_copy_cols(const float* __restrict__ buf, int LD, float* __restrict__ g, size_t gs,
                                               int beg, int end, int r0, int M, int lane) {
  const int per = (end - beg) / 2, nq = RB * per;
  for (int i = lane; i < nq; i += 64) {
    const int row = i / per, c = beg + (i % per) * 2;
    if (r0 + row < M) *(f32x2*)(g + (size_t)(r0 + row) * gs + c) = *(const f32x2*)(buf + row * LD + c);
  }
}

struct BlockFwdArgs {
  const float* x;
  long ldx;
  int M;
  BlockDesc d;      // d.W: the TRANSPOSED effective-layout copies W_i^T [in][out] (from pv_spectral_norm_multi)
  float* save;      // [M][SaveLayout.width]
  float* mean;      // [M]
  float* rstd;      // [M]
  float* out;       // [M][ldo]
  long ldo;
  int LD;           // LDS row stride (floats) = max width + 4
  int G;            // cluster size of this Block: 1, 2 or 4 (every layer width is a multiple of 32 G)
  unsigned save_bytes;
};

// Independent Blocks (the two pyramids' Blocks of one level, the three Blocks of a DLS_Auto level) share a launch:
// blockIdx.y picks the Block, blockIdx.x = slab + slab_stride * member.
struct BlockFwdGroup {
  BlockFwdArgs a[PV_BLOCK_GROUP_MAX];
  int slab_stride;
};

__global__ __launch_bounds__(NT) void block_fwd_cluster_kernel(BlockFwdGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BlockFwdArgs& a = grp.a[blockIdx.y];
  const int slab = blockIdx.x % grp.slab_stride, g = blockIdx.x / grp.slab_stride, G = a.G;
  if (g >= G || slab * RB >= a.M) return;          // the same answer for every member of a cluster
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, q4 = lane >> 4;
  const bool compute = w < 4;
  const int Din = a.d.Din, mid = a.d.mid, Dout = a.d.Dout, LD = a.LD, M = a.M;
  const int r0 = slab * RB;
  float* buf0 = lds;
  float* buf1 = lds + RB * LD;
  float* yb = lds + 2 * RB * LD;                  // y4 of this member's columns, parked for the hand-off
  float* par = lds + 3 * RB * LD;
  const SaveLayout L(Din, mid);
  const PartLayout P(Din, mid, Dout);
  const float* isg = par + P.width;
  float* rowstat = par + P.width + 8;             // mean[16], rstd[16]
  const size_t SW = L.width;
  unsigned* ctr = &g_fwd_sync[blockIdx.y][slab][0];
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.save, 0, a.save_bytes, 0x00020000);
  BSTAMP_DECL(par + P.width + 8 + 2 * RB)
  BSTAMP(0, 0);
#ifdef PV_BLOCK_STAMPS
  if (threadIdx.x == 0) bst[14] = clock64();
#endif

  f32x2 ring[RD2][4];
  if (compute) {
    ring2_fill(wptr2(a.d.W[0], Din, g * (Din / G) + w * TW, c16, q4), Din / 16, ring);
    // x rows: a0 = gelu(x) -> buf0, all Din columns in every member (all of a thread's loads are issued first:
    // widths <= 512 -> at most 8 per thread)
    const int nq = RB * Din / 4, per = Din / 4;
    f32x4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(tid + u * NTC, nq - 1);
      xv[u] = *(const f32x4*)(a.x + (size_t)min(r0 + i / per, M - 1) * a.ldx + (i % per) * 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * NTC;
      if (i < nq) {
        const f32x4 v = xv[u];
        *(f32x4*)(buf0 + (i / per) * LD + (i % per) * 4) = f32x4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
      }
    }
  } else {
    wave_stage_params(a.d, P, par, lane);
  }
  lds_barrier();
  BSTAMP(0, 1);

  // layers c1, c2, c3 (y and gelu(y) -> save, gelu(y) -> next input), c4 (y -> save, x + y -> LayerNorm input),
  // translate (-> out); each member computes its column share and, for G > 1, fetches the others' after the hand-off.
  for (int li = 0; li < 5; ++li) {
    const int N = li == 0 || li == 3 ? Din : (li == 4 ? Dout : mid);
    const int K = li == 0 || li == 1 || li == 4 ? Din : mid;
    const int Nn = li == 0 || li == 1 ? mid : (li == 2 ? Din : Dout), Kn = li == 0 || li == 3 ? Din : mid;
    const int Nw = N / G, nbeg = g * Nw, nend = nbeg + Nw;
    const float* in = li == 4 ? buf1 : ((li & 1) ? buf1 : buf0);
    float* outb = li == 3 ? buf0 : ((li & 1) ? buf0 : buf1);
    const int off_y = li == 0 ? L.y1 : li == 1 ? L.y2 : li == 2 ? L.y3 : L.y4;
    const int off_a = li == 0 ? L.a1 : li == 1 ? L.a2 : L.a3;
    if (compute) {
      const float* Wn = li < 4 ? a.d.W[li + 1] : nullptr;
      const int off_b = li == 0 ? P.b1 : li == 1 ? P.b2 : li == 2 ? P.b3 : li == 3 ? P.b4 : P.bt;
      const float is = isg[li];
      f32x2 xs[4];
      dense16s(in, LD, a.d.W[li], N, K, nbeg, nend, Wn, Nn, Kn, g * (Nn / G), w, lane, ring,
               [&](int n0) __attribute__((always_inline)) {
                 if (li != 3) return;                  // the residual input, requested before the k loop
                 const int col = min(n0 + 2 * c16, N - 2);
#pragma unroll
                 for (int j = 0; j < 4; ++j)
                   xs[j] = *(const f32x2*)(a.x + (size_t)min(r0 + q4 * 4 + j, M - 1) * a.ldx + col);
               },
               [&](f32x4 (&acc)[2], int n0) __attribute__((always_inline)) {
                 const int col = n0 + 2 * c16;
                 if (col >= nend) return;              // widths are multiples of 16: half of the last tile may be past N
                 const f32x2 bv = *(const f32x2*)(par + off_b + col);
#pragma unroll
                 for (int j = 0; j < 4; ++j) {
                   const int row = q4 * 4 + j, gr = r0 + row;
                   const f32x2 y = {acc[0][j] * is + bv[0], acc[1][j] * is + bv[1]};
                   if (li < 3) {
                     const f32x2 ga = {gelu_f(y[0]), gelu_f(y[1])};
                     *(f32x2*)(outb + row * LD + col) = ga;
                     if (gr < M) {
                       *(f32x2*)(a.save + (size_t)gr * SW + off_y + col) = y;              // read by the backward pass only
                       if (G == 1) *(f32x2*)(a.save + (size_t)gr * SW + off_a + col) = ga;   // (G > 1: the hand-off copy below)
                     }
                   } else if (li == 3) {
                     *(f32x2*)(buf0 + row * LD + col) = xs[j] + y;
                     if (G > 1) *(f32x2*)(yb + row * LD + col) = y;
                     else if (gr < M) *(f32x2*)(a.save + (size_t)gr * SW + off_y + col) = y;
                   } else if (gr < M) {
                     *(f32x2*)(a.out + (size_t)gr * a.ldo + col) = y;
                   }
                 }
               });
    } else if (li == 0) {
      wave_copy_cols(buf0, LD, a.save + L.a0, SW, g * (Din / G), (g + 1) * (Din / G), r0, M, lane);
    } else if (li == 4) {
      wave_copy_cols(buf1, LD, a.save + L.o, SW, g * (Din / G), (g + 1) * (Din / G), r0, M, lane);
      if (g == 0 && lane < RB && r0 + lane < M) {
        a.mean[r0 + lane] = rowstat[lane];
        a.rstd[r0 + lane] = rowstat[RB + lane];
      }
    }
    BSTAMP(0, 2 + 2 * li);
    if (li == 4) break;
    lds_barrier();
    if (G > 1) {
      // this member's slice of a_l = gelu(y_l) (layers c1..c3) or y4 (c4) -> `save`, 16 bytes per lane, write-through:
      // the backward pass reads it later and the other members do now
      const float* own = li < 3 ? outb : yb;
      const unsigned off_s = (unsigned)(li < 3 ? off_a : off_y);
      const int per = Nw / 4, nq = RB * per;
      for (int i = tid; i < nq; i += NT) {
        const int row = i / per, col = nbeg + (i % per) * 4;
        if (r0 + row < M)
          st16_sc1(srsrc, (unsigned)(((size_t)(r0 + row) * SW + off_s + col) * sizeof(float)), *(const f32x4*)(own + row * LD + col));
      }
      // hand-off: the slice has left the CU (vmcnt) -> count this member -> wait for the others -> fetch their slices
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (tid == 0) cluster_arrive_wait(ctr, (unsigned)(G * (li + 1)));
      lds_barrier();
      const int pero = (N - Nw) / 4, nqo = RB * pero;
      for (int i = tid; i < nqo; i += NT) {
        const int row = i / pero, cc = (i % pero) * 4, col = cc < nbeg ? cc : cc + Nw;
        const int gr = min(r0 + row, M - 1);
        f32x4 v = ld16_sc1(srsrc, (unsigned)(((size_t)gr * SW + off_s + col) * sizeof(float)));
        if (li == 3) v += *(const f32x4*)(a.x + (size_t)gr * a.ldx + col);
        *(f32x4*)(outb + row * LD + col) = v;
      }
      lds_barrier();
    }
    BSTAMP(0, 3 + 2 * li);
    if (li == 3) {
      if (compute) {
        // LayerNorm of s = x + y4 (buf0; biased variance, two passes as nn.LayerNorm) -> buf1: 16 lanes per row, the
        // wave's four rows side by side (every member normalises the whole row: it needs all of it for the translate layer)
        const int row = w * 4 + q4;
        const float* srow = buf0 + row * LD;
        float sm = 0.f;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 v = *(const f32x4*)(srow + c);
          sm += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float mu = group16_sum(sm) / Din;
        float q = 0.f;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 v = *(const f32x4*)(srow + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) q += (v[e] - mu) * (v[e] - mu);
        }
        const float rs = rsqrtf(group16_sum(q) / Din + a.d.eps);
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 v = *(const f32x4*)(srow + c), gm = *(const f32x4*)(par + P.dg + c), bt = *(const f32x4*)(par + P.dbe + c);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu) * rs * gm[e] + bt[e];
          *(f32x4*)(buf1 + row * LD + c) = o;
        }
        if (c16 == 0) {
          rowstat[row] = mu;
          rowstat[RB + row] = rs;
        }
      }
      lds_barrier();
      BSTAMP(0, 12);
    }
  }
  if (G > 1 && tid == 0) cluster_depart(ctr, (unsigned)G);
#ifdef PV_BLOCK_STAMPS
  if (threadIdx.x == 0) bst[15] = clock64();
#endif
  BSTAMP_FLUSH(0);
}

struct BlockBwdArgs1 {
  const float* dout;   // [M][lddo]   dL/d out
  long lddo;
  const float* x;      // [M][ldx]
  long ldx;
  const float* out;    // [M][ldo]    forward output (for the translate layer's spectral term)
  long ldo;
  int M;
  BlockDesc d;         // d.W: weight_orig [out][in]
  const float* save;
  const float* mean;
  const float* rstd;
  float* dx;           // [M][lddx]
  long lddx;
  float* dY;           // [M][GradLayout.width]
  float* cpart;        // [5][nwg]   partial <dy_i, y_i - b_i>
  float* ppart;        // [nwg][PartLayout.width]
  int LD;
};

struct BlockBwdGroup1 {
  BlockBwdArgs1 a[PV_BLOCK_GROUP_MAX];
};

__global__ __launch_bounds__(NT) void block_bwd_dx_kernel(BlockBwdGroup1 grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BlockBwdArgs1& a = grp.a[blockIdx.y];
  if ((int)blockIdx.x * RB >= a.M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, q4 = lane >> 4;
  const bool compute = w < 4;
  const int Din = a.d.Din, mid = a.d.mid, Dout = a.d.Dout, LD = a.LD, M = a.M;
  const int r0 = blockIdx.x * RB, nwg = (a.M + RB - 1) / RB;
  float* buf0 = lds;
  float* buf1 = lds + RB * LD;
  float* buf2 = lds + 2 * RB * LD;
  float* sbuf = lds + 3 * RB * LD;                              // s = x + y4 (the LayerNorm input)
  float* par = lds + 4 * RB * LD;
  const SaveLayout L(Din, mid);
  const GradLayout G(Din, mid);
  const PartLayout P(Din, mid, Dout);
  const float* isg = par + P.width;
  float* rowstat = par + P.width + 8;                           // mean[16], rstd[16]
  float (*cred)[5] = (float (*)[5])(par + P.width + 8 + 2 * RB);   // [4 waves][5]: <dy_i, y_i - b_i>, i = c1..c4, translate
  volatile int* flag = (volatile int*)(par + P.width + 8 + 2 * RB + 20);   // storer -> compute: LayerNorm column sums read
  const size_t SW = L.width, GW = G.width;
  float* prow = a.ppart + (size_t)blockIdx.x * P.width;

  f32x4 ring[RD][4];
  if (compute) {
    ring_fill(wptr(a.d.W[4], Din, w * 64, c16, q4), Dout / 16, ring);
    // dL/dout rows -> buf0 (zero past M) with the translate layer's spectral term; s = x + y4 -> sbuf.  Loads first.
    const int nqo = RB * Dout / 4, pero = Dout / 4, nqi = RB * Din / 4, peri = Din / 4;
    f32x4 dv[8], ov[8], bv[8], xv[8], yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(tid + u * NTC, nqo - 1), gr = min(r0 + i / pero, M - 1), c4 = (i % pero) * 4;
      dv[u] = *(const f32x4*)(a.dout + (size_t)gr * a.lddo + c4);
      ov[u] = *(const f32x4*)(a.out + (size_t)gr * a.ldo + c4);
      bv[u] = *(const f32x4*)(a.d.bias[4] + c4);
      const int k = min(tid + u * NTC, nqi - 1), gk = min(r0 + k / peri, M - 1), k4 = (k % peri) * 4;
      xv[u] = *(const f32x4*)(a.x + (size_t)gk * a.ldx + k4);
      yv[u] = *(const f32x4*)(a.save + (size_t)gk * SW + L.y4 + k4);
    }
    float c4s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * NTC;
      if (i < nqo) {
        const int row = i / pero;
        f32x4 v = dv[u];
        if (r0 + row >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) c4s += v[e] * (ov[u][e] - bv[u][e]);
        *(f32x4*)(buf0 + row * LD + (i % pero) * 4) = v;
      }
      if (i < nqi) *(f32x4*)(sbuf + (i / peri) * LD + (i % peri) * 4) = xv[u] + yv[u];
    }
    c4s = wave_sum(c4s);
    if (lane == 0) cred[w][4] = c4s;
  } else {
    wave_stage_params(a.d, P, par, lane);
    if (lane < RB) {
      const int grc = min(r0 + lane, M - 1);
      rowstat[lane] = a.mean[grc];
      rowstat[RB + lane] = a.rstd[grc];
    }
    if (lane == 0) *flag = 0;
  }
  lds_barrier();

  // step 0: do = dout W_t / sigma_t -> buf1, then the LayerNorm backward (ds = dy4 -> buf2);
  // steps 1..3: dy_i = (dy_{i+1} W_{i+1} / sigma_{i+1}) * gelu'(y_i) for y3, y2, y1 (buffers 2 -> 0 -> 1 -> 0);
  // step 4: dx = ds + (dy1 W_1 / sigma_1) * gelu'(x), and the c4 spectral term (y4 - b4 = s - x - b4).
  // The storer wave runs one step behind: gradients of the inner linears -> dY, column sums -> this workgroup's
  // partial row of the bias / LayerNorm parameter gradients.
  for (int st = 0; st < 5; ++st) {
    const int li = 4 - st;                                  // the linear whose weight this step multiplies by
    const int N = st == 0 || st >= 3 ? Din : mid;
    const int K = st == 0 ? Dout : (st == 1 || st == 4 ? Din : mid);
    const int Nn = st == 0 || st == 1 ? mid : Din, Kn = st == 0 || st == 3 ? Din : mid;
    const float* in = st == 0 ? buf0 : st == 1 ? buf2 : st == 2 ? buf0 : st == 3 ? buf1 : buf0;
    float* outb = st == 0 ? buf1 : st == 1 ? buf0 : st == 2 ? buf1 : buf0;
    if (compute) {
      const float* Wn = st < 4 ? a.d.W[li - 1] : nullptr;
      const int off_y = st == 1 ? L.y3 : st == 2 ? L.y2 : L.y1;
      const int off_b = st == 1 ? P.b3 : st == 2 ? P.b2 : st == 3 ? P.b1 : P.b4;
      // what the epilogue multiplies by gelu'(.): saved y_i (steps 1..3) or x (step 4), requested before the k loop
      const float* pbase = st == 4 ? a.x : a.save + off_y;
      const long pstride = st == 4 ? a.ldx : (long)SW;
      const float is = isg[li];
      float csum = 0.f;
      f32x4 ys[4];
      dense16(in, LD, a.d.W[li], N, K, Wn, Nn, Kn, w, lane, ring,
              [&](int n0) __attribute__((always_inline)) {
                if (st == 0) return;
                const int col = min(n0 + 4 * c16, N - 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  ys[j] = *(const f32x4*)(pbase + (size_t)min(r0 + q4 * 4 + j, M - 1) * pstride + col);
              },
              [&](f32x4 (&acc)[4], int n0) __attribute__((always_inline)) {
                const int col = n0 + 4 * c16;
                if (col >= N) return;
                if (st == 0) {
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    *(f32x4*)(outb + (q4 * 4 + j) * LD + col) = f32x4{acc[0][j] * is, acc[1][j] * is, acc[2][j] * is, acc[3][j] * is};
                  return;
                }
                if (st == 1)                                 // buf0 still feeds the storer's dgamma column sums
                  while (*flag == 0) __builtin_amdgcn_s_sleep(1);
                const f32x4 bv = *(const f32x4*)(par + off_b + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int row = q4 * 4 + j, gr = r0 + row;
                  if (st < 4) {
                    f32x4 dy = {0.f, 0.f, 0.f, 0.f};
                    if (gr < M) {
#pragma unroll
                      for (int e = 0; e < 4; ++e) {
                        dy[e] = acc[e][j] * is * gelu_df(ys[j][e]);
                        csum += dy[e] * (ys[j][e] - bv[e]);
                      }
                    }
                    *(f32x4*)(outb + row * LD + col) = dy;
                  } else if (gr < M) {
                    const f32x4 ds = *(const f32x4*)(buf2 + row * LD + col), sv = *(const f32x4*)(sbuf + row * LD + col);
                    f32x4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      r[e] = ds[e] + acc[e][j] * is * gelu_df(ys[j][e]);
                      csum += ds[e] * (sv[e] - ys[j][e] - bv[e]);
                    }
                    *(f32x4*)(a.dx + (size_t)gr * a.lddx + col) = r;
                  }
                }
              });
      if (st > 0) {
        csum = wave_sum(csum);
        if (lane == 0) cred[w][st == 4 ? 3 : 3 - st] = csum;
      }
    } else {
      if (st == 0) {
        wave_colsum(buf0, LD, Dout, prow + P.bt, lane);
      } else if (st == 1) {
        wave_copy_out(buf2, LD, a.dY + G.d4, GW, Din, r0, M, lane);
        wave_colsum(buf0, LD, Din, prow + P.dg, lane);
        wave_colsum(buf1, LD, Din, prow + P.dbe, lane);
        wave_colsum(buf2, LD, Din, prow + P.b4, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *flag = 1;
      } else {
        // dy of the previous step sits in this step's input
        const int pN = st == 4 ? Din : mid;
        const int off_d = st == 2 ? G.d3 : st == 3 ? G.d2 : G.d1, off_pb = st == 2 ? P.b3 : st == 3 ? P.b2 : P.b1;
        wave_copy_out(in, LD, a.dY + off_d, GW, pN, r0, M, lane);
        wave_colsum(in, LD, pN, prow + off_pb, lane);
      }
    }
    lds_barrier();
    if (st == 0) {
      if (compute) {
        // LayerNorm backward: ds = rstd (g - mean(g) - xhat mean(g xhat)), g = do * gamma.  buf0 <- do * xhat (for
        // dgamma), buf1 keeps do (for dbeta), buf2 <- ds = dy4 (also the residual branch of dx)
        const int row = w * 4 + q4;
        const float mu = rowstat[row], rs = rowstat[RB + row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 sv = *(const f32x4*)(sbuf + row * LD + c), dv = *(const f32x4*)(buf1 + row * LD + c),
                      gm = *(const f32x4*)(par + P.dg + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = dv[e] * gm[e];
            s1 += g;
            s2 += g * ((sv[e] - mu) * rs);
          }
        }
        s1 = group16_sum(s1) / Din;
        s2 = group16_sum(s2) / Din;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 sv = *(const f32x4*)(sbuf + row * LD + c), dv = *(const f32x4*)(buf1 + row * LD + c),
                      gm = *(const f32x4*)(par + P.dg + c);
          f32x4 dsv, dxh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (sv[e] - mu) * rs;
            dsv[e] = rs * (dv[e] * gm[e] - s1 - xh * s2);       // rows past M: do = 0 -> ds = 0
            dxh[e] = dv[e] * xh;
          }
          *(f32x4*)(buf0 + row * LD + c) = dxh;
          *(f32x4*)(buf2 + row * LD + c) = dsv;
        }
      }
      lds_barrier();
    }
  }
  if (tid < 5) a.cpart[tid * nwg + blockIdx.x] = cred[0][tid] + cred[1][tid] + cred[2][tid] + cred[3][tid];
}

struct BlockBwdArgs {
  const float* dout;   // [M][lddo]   dL/d out
  long lddo;
  const float* x;      // [M][ldx]
  long ldx;
  const float* out;    // [M][ldo]    forward output (for the translate layer's spectral term)
  long ldo;
  int M;
  BlockDesc d;         // d.W: weight_orig [out][in]
  const float* save;
  const float* mean;
  const float* rstd;
  float* dx;           // [M][lddx]
  long lddx;
  float* dY;           // [M][GradLayout.width]
  float* cpart;        // [5][nwg * G]   partial <dy_i, y_i - b_i>, one per member
  float* ppart;        // [nwg][PartLayout.width]
  int LD;
  int G;               // cluster size (block_cluster), as in the forward kernel
  unsigned dx_bytes, dY_bytes;
};

struct BlockBwdGroup {
  BlockBwdArgs a[PV_BLOCK_GROUP_MAX];
  int slab_stride;
};

__device__ unsigned g_bwd_sync[PV_BLOCK_GROUP_MAX][SLAB_MAX][2];   // as g_fwd_sync

// column sums over the workgroup's 16 rows of columns [beg, end) of an LDS image (rows past M hold zeros)
__device__ __forceinline__ void wave_colsum_cols(const float* __restrict__ buf, int LD, int beg, int end, float* __restrict__ dst,
                                                 int lane) {
  for (int c = beg + lane * 4; c < end; c += 256) {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RB; ++r) sum += *(const f32x4*)(buf + r * LD + c);
    *(f32x4*)(dst + c) = sum;
  }
}

// The dx chain with the forward kernel's clusters: the members of a slab split the columns of every step's output.
// The LayerNorm backward needs whole rows of do = dout W_t / sigma_t, and every later step whole rows of the dy before
// it, so each step ends in a hand-off of the members' slices: do travels through the rows of `dx` (free until the last
// step writes them), dy3 / dy2 / dy1 through the rows of `dY` where the weight-gradient GEMMs read them anyway.
__global__ __launch_bounds__(NT) void block_bwd_dx_cluster_kernel(BlockBwdGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BlockBwdArgs& a = grp.a[blockIdx.y];
  const int slab = blockIdx.x % grp.slab_stride, g = blockIdx.x / grp.slab_stride, G = a.G;
  if (g >= G || slab * RB >= a.M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, q4 = lane >> 4;
  const bool compute = w < 4;
  const int Din = a.d.Din, mid = a.d.mid, Dout = a.d.Dout, LD = a.LD, M = a.M;
  const int r0 = slab * RB, nwg = (a.M + RB - 1) / RB;
  float* buf0 = lds;
  float* buf1 = lds + RB * LD;
  float* buf2 = lds + 2 * RB * LD;
  float* sbuf = lds + 3 * RB * LD;                              // s = x + y4 (the LayerNorm input)
  float* par = lds + 4 * RB * LD;
  const SaveLayout L(Din, mid);
  const GradLayout GL(Din, mid);
  const PartLayout P(Din, mid, Dout);
  const float* isg = par + P.width;
  float* rowstat = par + P.width + 8;                           // mean[16], rstd[16]
  float (*cred)[5] = (float (*)[5])(par + P.width + 8 + 2 * RB);   // [4 waves][5]: <dy_i, y_i - b_i>, i = c1..c4, translate
  volatile int* flag = (volatile int*)(par + P.width + 8 + 2 * RB + 20);   // storer -> compute: LayerNorm column sums read
  const size_t SW = L.width, GW = GL.width;
  float* prow = a.ppart + (size_t)slab * P.width;
  unsigned* ctr = &g_bwd_sync[blockIdx.y][slab][0];
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, a.dx_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, a.dY_bytes, 0x00020000);
  const int din_b = g * (Din / G), din_e = din_b + Din / G;     // this member's share of a Din-wide row

  f32x2 ring[RD2][4];
  if (compute) {
    ring2_fill(wptr2(a.d.W[4], Din, din_b + w * TW, c16, q4), Dout / 16, ring);
    // dL/dout rows -> buf0 (zero past M) with the translate layer's spectral term; s = x + y4 -> sbuf.  Loads first.
    const int nqo = RB * Dout / 4, pero = Dout / 4, nqi = RB * Din / 4, peri = Din / 4;
    f32x4 dv[8], ov[8], bv[8], xv[8], yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(tid + u * NTC, nqo - 1), gr = min(r0 + i / pero, M - 1), c4 = (i % pero) * 4;
      dv[u] = *(const f32x4*)(a.dout + (size_t)gr * a.lddo + c4);
      ov[u] = *(const f32x4*)(a.out + (size_t)gr * a.ldo + c4);
      bv[u] = *(const f32x4*)(a.d.bias[4] + c4);
      const int k = min(tid + u * NTC, nqi - 1), gk = min(r0 + k / peri, M - 1), k4 = (k % peri) * 4;
      xv[u] = *(const f32x4*)(a.x + (size_t)gk * a.ldx + k4);
      yv[u] = *(const f32x4*)(a.save + (size_t)gk * SW + L.y4 + k4);
    }
    float c4s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * NTC;
      if (i < nqo) {
        const int row = i / pero;
        f32x4 v = dv[u];
        if (r0 + row >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) c4s += v[e] * (ov[u][e] - bv[u][e]);
        *(f32x4*)(buf0 + row * LD + (i % pero) * 4) = v;
      }
      if (i < nqi) *(f32x4*)(sbuf + (i / peri) * LD + (i % peri) * 4) = xv[u] + yv[u];
    }
    c4s = wave_sum(c4s);
    if (lane == 0) cred[w][4] = g == 0 ? c4s : 0.f;             // every member holds all of dout: one of them counts it
  } else {
    wave_stage_params(a.d, P, par, lane);
    if (lane < RB) {
      const int grc = min(r0 + lane, M - 1);
      rowstat[lane] = a.mean[grc];
      rowstat[RB + lane] = a.rstd[grc];
    }
    if (lane == 0) *flag = 0;
  }
  lds_barrier();

  // step 0: do = dout W_t / sigma_t -> buf1, then the LayerNorm backward (ds = dy4 -> buf2);
  // steps 1..3: dy_i = (dy_{i+1} W_{i+1} / sigma_{i+1}) * gelu'(y_i) for y3, y2, y1 (buffers 2 -> 0 -> 1 -> 0);
  // step 4: dx = ds + (dy1 W_1 / sigma_1) * gelu'(x), and the c4 spectral term (y4 - b4 = s - x - b4).
  // The storer wave runs one step behind: column sums of this member's columns -> the slab's partial row of the bias /
  // LayerNorm parameter gradients (and, for G = 1, the dy copies to dY; a cluster stores them in its hand-off).
  for (int st = 0; st < 5; ++st) {
    const int li = 4 - st;                                  // the linear whose weight this step multiplies by
    const int N = st == 0 || st >= 3 ? Din : mid;
    const int K = st == 0 ? Dout : (st == 1 || st == 4 ? Din : mid);
    const int Nn = st == 0 || st == 1 ? mid : Din, Kn = st == 0 || st == 3 ? Din : mid;
    const int Nw = N / G, nbeg = g * Nw, nend = nbeg + Nw;
    const float* in = st == 0 ? buf0 : st == 1 ? buf2 : st == 2 ? buf0 : st == 3 ? buf1 : buf0;
    float* outb = st == 0 ? buf1 : st == 1 ? buf0 : st == 2 ? buf1 : buf0;
    if (compute) {
      const float* Wn = st < 4 ? a.d.W[li - 1] : nullptr;
      const int off_y = st == 1 ? L.y3 : st == 2 ? L.y2 : L.y1;
      const int off_b = st == 1 ? P.b3 : st == 2 ? P.b2 : st == 3 ? P.b1 : P.b4;
      // what the epilogue multiplies by gelu'(.): saved y_i (steps 1..3) or x (step 4), requested before the k loop
      const float* pbase = st == 4 ? a.x : a.save + off_y;
      const long pstride = st == 4 ? a.ldx : (long)SW;
      const float is = isg[li];
      float csum = 0.f;
      f32x2 ys[4];
      dense16s(in, LD, a.d.W[li], N, K, nbeg, nend, Wn, Nn, Kn, g * (Nn / G), w, lane, ring,
               [&](int n0) __attribute__((always_inline)) {
                 if (st == 0) return;
                 const int col = min(n0 + 2 * c16, N - 2);
#pragma unroll
                 for (int j = 0; j < 4; ++j)
                   ys[j] = *(const f32x2*)(pbase + (size_t)min(r0 + q4 * 4 + j, M - 1) * pstride + col);
               },
               [&](f32x4 (&acc)[2], int n0) __attribute__((always_inline)) {
                 const int col = n0 + 2 * c16;
                 if (col >= nend) return;
                 if (st == 0) {
#pragma unroll
                   for (int j = 0; j < 4; ++j)
                     *(f32x2*)(outb + (q4 * 4 + j) * LD + col) = f32x2{acc[0][j] * is, acc[1][j] * is};
                   return;
                 }
                 if (st == 1)                                 // buf0 still feeds the storer's dgamma column sums
                   while (*flag == 0) __builtin_amdgcn_s_sleep(1);
                 const f32x2 bv = *(const f32x2*)(par + off_b + col);
#pragma unroll
                 for (int j = 0; j < 4; ++j) {
                   const int row = q4 * 4 + j, gr = r0 + row;
                   if (st < 4) {
                     f32x2 dy = {0.f, 0.f};
                     if (gr < M) {
#pragma unroll
                       for (int e = 0; e < 2; ++e) {
                         dy[e] = acc[e][j] * is * gelu_df(ys[j][e]);
                         csum += dy[e] * (ys[j][e] - bv[e]);
                       }
                     }
                     *(f32x2*)(outb + row * LD + col) = dy;
                   } else if (gr < M) {
                     const f32x2 ds = *(const f32x2*)(buf2 + row * LD + col), sv = *(const f32x2*)(sbuf + row * LD + col);
                     f32x2 r;
#pragma unroll
                     for (int e = 0; e < 2; ++e) {
                       r[e] = ds[e] + acc[e][j] * is * gelu_df(ys[j][e]);
                       csum += ds[e] * (sv[e] - ys[j][e] - bv[e]);
                     }
                     *(f32x2*)(a.dx + (size_t)gr * a.lddx + col) = r;
                   }
                 }
               });
      if (st > 0) {
        csum = wave_sum(csum);
        if (lane == 0) cred[w][st == 4 ? 3 : 3 - st] = csum;
      }
    } else {
      if (st == 0) {
        wave_colsum_cols(buf0, LD, g * (Dout / G), (g + 1) * (Dout / G), prow + P.bt, lane);
      } else if (st == 1) {
        wave_copy_cols(buf2, LD, a.dY + GL.d4, GW, din_b, din_e, r0, M, lane);
        wave_colsum_cols(buf0, LD, din_b, din_e, prow + P.dg, lane);
        wave_colsum_cols(buf1, LD, din_b, din_e, prow + P.dbe, lane);
        wave_colsum_cols(buf2, LD, din_b, din_e, prow + P.b4, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *flag = 1;
      } else {
        // dy of the previous step sits in this step's input
        const int pN = st == 4 ? Din : mid;
        const int off_d = st == 2 ? GL.d3 : st == 3 ? GL.d2 : GL.d1, off_pb = st == 2 ? P.b3 : st == 3 ? P.b2 : P.b1;
        const int pb = g * (pN / G), pe = pb + pN / G;
        if (G == 1) wave_copy_cols(in, LD, a.dY + off_d, GW, pb, pe, r0, M, lane);
        wave_colsum_cols(in, LD, pb, pe, prow + off_pb, lane);
      }
    }
    lds_barrier();
    if (G > 1 && st < 4) {
      // hand-off of this step's output: do through the rows of dx, dy3 / dy2 / dy1 through dY (16-byte sc1 stores of this
      // member's slice -> drained -> arrival -> the others' slices with sc1 loads; rows past M are zeros)
      const __amdgpu_buffer_rsrc_t rs = st == 0 ? xrsrc : yrsrc;
      const size_t rstride = st == 0 ? (size_t)a.lddx : GW;
      const unsigned off_s = st == 0 ? 0u : (unsigned)(st == 1 ? GL.d3 : st == 2 ? GL.d2 : GL.d1);
      const int per = Nw / 4, nq = RB * per;
      for (int i = tid; i < nq; i += NT) {
        const int row = i / per, col = nbeg + (i % per) * 4;
        if (r0 + row < M)
          st16_sc1(rs, (unsigned)(((size_t)(r0 + row) * rstride + off_s + col) * sizeof(float)), *(const f32x4*)(outb + row * LD + col));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (tid == 0) cluster_arrive_wait(ctr, (unsigned)(G * (st + 1)));
      lds_barrier();
      const int pero = (N - Nw) / 4, nqo = RB * pero;
      for (int i = tid; i < nqo; i += NT) {
        const int row = i / pero, cc = (i % pero) * 4, col = cc < nbeg ? cc : cc + Nw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < M) v = ld16_sc1(rs, (unsigned)(((size_t)(r0 + row) * rstride + off_s + col) * sizeof(float)));
        *(f32x4*)(outb + row * LD + col) = v;
      }
      lds_barrier();
    }
    if (st == 0) {
      if (compute) {
        // LayerNorm backward: ds = rstd (g - mean(g) - xhat mean(g xhat)), g = do * gamma.  buf0 <- do * xhat (for
        // dgamma), buf1 keeps do (for dbeta), buf2 <- ds = dy4 (also the residual branch of dx); whole rows in every member
        const int row = w * 4 + q4;
        const float mu = rowstat[row], rs = rowstat[RB + row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 sv = *(const f32x4*)(sbuf + row * LD + c), dv = *(const f32x4*)(buf1 + row * LD + c),
                      gm = *(const f32x4*)(par + P.dg + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gg = dv[e] * gm[e];
            s1 += gg;
            s2 += gg * ((sv[e] - mu) * rs);
          }
        }
        s1 = group16_sum(s1) / Din;
        s2 = group16_sum(s2) / Din;
        for (int c = c16 * 4; c < Din; c += 64) {
          const f32x4 sv = *(const f32x4*)(sbuf + row * LD + c), dv = *(const f32x4*)(buf1 + row * LD + c),
                      gm = *(const f32x4*)(par + P.dg + c);
          f32x4 dsv, dxh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (sv[e] - mu) * rs;
            dsv[e] = rs * (dv[e] * gm[e] - s1 - xh * s2);       // rows past M: do = 0 -> ds = 0
            dxh[e] = dv[e] * xh;
          }
          *(f32x4*)(buf0 + row * LD + c) = dxh;
          *(f32x4*)(buf2 + row * LD + c) = dsv;
        }
      }
      lds_barrier();
    }
  }
  if (G > 1 && tid == 0) cluster_depart(ctr, (unsigned)G);
  if (tid < 5) a.cpart[(size_t)tid * nwg * G + slab * G + g] = cred[0][tid] + cred[1][tid] + cred[2][tid] + cred[3][tid];
}

#include "block_rows.h"

struct ParamReduceArgs {
  const float* ppart;    // [nwg][width]
  int nwg, width;
  int seg_end[7];        // exclusive ends of db1, db2, db3, db4, dbt, dgamma, dbeta inside a partial row
  float* dst[7];
  int accumulate;
};

struct ParamReduceGroup {
  ParamReduceArgs r[PV_BLOCK_GROUP_MAX];
};

int check_dims(int M, int Din, int mid, int Dout, const char* who) {
  PV_REQUIRE(M > 0 && Din > 0 && mid > 0 && Dout > 0, "%s: empty shape", who);
  PV_REQUIRE(Din % 16 == 0 && mid % 16 == 0 && Dout % 16 == 0, "%s: widths %d/%d/%d must be multiples of 16", who, Din, mid, Dout);
  PV_REQUIRE(Din <= 512 && mid <= 512 && Dout <= 512, "%s: widths %d/%d/%d exceed 512 (16-row LDS images)", who, Din, mid, Dout);
  return PV_OK;
}

int fill_desc(BlockDesc& d, const pv_block_call& c, bool transposed, const char* who) {
  for (int i = 0; i < 5; ++i) {
    const float* W = transposed ? c.Wt[i] : c.W[i];
    PV_REQUIRE(W && c.bias[i] && c.inv_sigma[i], "%s: null parameter of linear %d", who, i);
    PV_REQUIRE(((uintptr_t)W & 15) == 0 && ((uintptr_t)c.bias[i] & 15) == 0, "%s: parameters must be 16-byte aligned", who);
    d.W[i] = W;
    d.bias[i] = c.bias[i];
    d.inv_sigma[i] = c.inv_sigma[i];
  }
  PV_REQUIRE(c.gamma && (c.beta || !transposed), "%s: null LayerNorm parameter", who);
  PV_REQUIRE(((uintptr_t)c.gamma & 15) == 0 && ((uintptr_t)c.beta & 15) == 0, "%s: parameters must be 16-byte aligned", who);
  d.gamma = c.gamma;
  d.beta = c.beta ? c.beta : c.gamma;
  d.Din = c.Din;
  d.mid = c.mid;
  d.Dout = c.Dout;
  d.eps = c.ln_eps;
  return PV_OK;
}

int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// Which kernels walk a launch's Blocks: 4-row slabs (block_rows.h) unless PEVAE_BLOCK_ROWS=16 asks for the 16-row kernels
// (single workgroups below 512 columns, clusters of four at 512).  PEVAE_BLOCK_ROWS_WIDE=0 keeps the clusters for launches
// with a 512-wide Block (a workgroup of the 4-row kernel streams all 2.9 MB of such a Block through its own L1, ~25 us,
// against 55-70 us for the cluster launch with its hand-offs: 2.759 vs 2.771 ms per step, twice).  Read once: forward, dx
// chain and parameter-gradient launch of a process must agree on the number of per-workgroup partial rows.
// Which kernels walk a launch's Blocks (pv_block_rows; environment at first use): 4-row slabs (block_rows.h) unless 16 is
// asked for (single workgroups below 512 columns, clusters of four at 512).  wide = 0 keeps the clusters for launches
// with a 512-wide Block: a workgroup of the 4-row kernel streams all 2.9 MB of such a Block through its own L1, ~25 us,
// against 55-70 us for the cluster launch with its hand-offs -- 2.759 vs 2.771 ms per step, twice, so wide = 1 by default.
int g_rows = -1, g_rows_wide = -1;
void rows_init() {
  if (g_rows < 0) {
    const char* e = getenv("PEVAE_BLOCK_ROWS");
    g_rows = e && atoi(e) == 16 ? 16 : 4;
  }
  if (g_rows_wide < 0) {
    const char* f = getenv("PEVAE_BLOCK_ROWS_WIDE");
    g_rows_wide = f ? (atoi(f) != 0) : 1;
  }
}
bool rows_mode(int gmax) {
  rows_init();
  return g_rows == 4 && (gmax == 1 || g_rows_wide != 0);
}

// Cluster size of a Block's forward launch: as many members as leave every one of them at least 64 columns of the
// narrowest layer and whole 32-column tiles of every layer; only Blocks wide enough to be bound by one CU's MFMA and
// weight-stream rate are split.
int block_cluster(int M, int Din, int mid, int Dout) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("PEVAE_BLOCK_CLUSTER");
    forced = e ? atoi(e) : 0;
  }
  if (cdiv(M, RB) > SLAB_MAX) return 1;
  const int wmax = max3(Din, mid, Dout);
  int G = wmax >= 512 ? 4 : 1;      // (256-wide Blocks: two members measured no faster than one, 35.7 vs 36.0 us)
  if (forced == 1 || forced == 2 || forced == 4) G = forced;
  while (G > 1 && (Din % (TW * G) || mid % (TW * G) || Dout % (TW * G))) G >>= 1;
  return G;
}

template <class K>
int set_lds(K kernel, size_t bytes, size_t* seen) {
  if (bytes > *seen) {
    PV_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    *seen = bytes;
  }
  return PV_OK;
}

}  // namespace

extern "C" {

#ifdef PV_BLOCK_STAMPS
int pv_block_stamps_read(unsigned long long* host64) {
  PV_HIP(hipDeviceSynchronize());
  PV_HIP(hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_block_stamps), sizeof(unsigned long long) * 64));
  return PV_OK;
}
#endif

int pv_block_layout(int Din, int mid, int Dout, int* save_width, int* grad_width, int* part_width) {
  int rc = check_dims(1, Din, mid, Dout, "pv_block_layout");
  if (rc) return rc;
  if (save_width) *save_width = SaveLayout(Din, mid).width;
  if (grad_width) *grad_width = GradLayout(Din, mid).width;
  if (part_width) *part_width = PartLayout(Din, mid, Dout).width;
  return PV_OK;
}

int pv_block_fwd(const pv_block_call* calls, int n, void* stream) {
  PV_REQUIRE(calls && n >= 1 && n <= PV_BLOCK_GROUP_MAX, "pv_block_fwd: 1..%d blocks per launch", PV_BLOCK_GROUP_MAX);
  BlockFwdGroup g;
  size_t smem = 0;
  int nslab = 0, gmax = 1;
  for (int b = 0; b < n; ++b) {
    const pv_block_call& c = calls[b];
    int rc = check_dims(c.M, c.Din, c.mid, c.Dout, "pv_block_fwd");
    if (rc) return rc;
    PV_REQUIRE(c.x && c.save && c.mean && c.rstd && c.out && c.ldx >= c.Din && c.ldo >= c.Dout && c.ldx % 4 == 0 &&
                   c.ldo % 4 == 0, "pv_block_fwd: bad argument (block %d)", b);
    PV_REQUIRE(((uintptr_t)c.x & 15) == 0 && ((uintptr_t)c.save & 15) == 0 && ((uintptr_t)c.out & 15) == 0,
               "pv_block_fwd: buffers must be 16-byte aligned");
    BlockFwdArgs& a = g.a[b];
    if ((rc = fill_desc(a.d, c, true, "pv_block_fwd"))) return rc;
    a.x = c.x; a.ldx = c.ldx; a.M = c.M; a.save = c.save; a.mean = c.mean; a.rstd = c.rstd; a.out = c.out; a.ldo = c.ldo;
    a.LD = max3(c.Din, c.mid, c.Dout) + 4;
    a.G = block_cluster(c.M, c.Din, c.mid, c.Dout);
    a.save_bytes = (unsigned)((size_t)c.M * SaveLayout(c.Din, c.mid).width * sizeof(float));
    const size_t need = ((size_t)3 * RB * a.LD + PartLayout(c.Din, c.mid, c.Dout).width + 8 + 2 * RB + 32) * sizeof(float);
    if (need > smem) smem = need;
    if (cdiv(c.M, RB) > nslab) nslab = cdiv(c.M, RB);
    if (a.G > gmax) gmax = a.G;
  }
  for (int b = n; b < PV_BLOCK_GROUP_MAX; ++b) g.a[b] = g.a[0];
  if (rows_mode(gmax)) {                        // 4-row slabs: cdiv(M, 4) workgroups per Block
    BlockFwdGroup1 g1;
    size_t smem4 = 0;
    int nslab4 = 0;
    for (int b = 0; b < PV_BLOCK_GROUP_MAX; ++b) {
      const BlockFwdArgs& a = g.a[b];
      g1.a[b] = BlockFwdArgs1{a.x, a.ldx, a.M, a.d, a.save, a.mean, a.rstd, a.out, a.ldo, a.LD};
      const size_t need = (size_t)rows_fwd_lds_floats(a.LD, PartLayout(a.d.Din, a.d.mid, a.d.Dout).width) * sizeof(float);
      if (need > smem4) smem4 = need;
      if (cdiv(a.M, RR) > nslab4) nslab4 = cdiv(a.M, RR);
    }
    bool wide = false;
    for (int b = 0; b < n; ++b) wide = wide || max3(calls[b].Din, calls[b].mid, calls[b].Dout) > 256;
    static size_t seen4 = 0, seen4w = 0;
    int rc = wide ? set_lds(block_fwd_rows_kernel<true>, smem4, &seen4w) : set_lds(block_fwd_rows_kernel<false>, smem4, &seen4);
    if (rc) return rc;
    if (wide) hipLaunchKernelGGL(block_fwd_rows_kernel<true>, dim3(nslab4, n), dim3(NTR), smem4, (hipStream_t)stream, g1);
    else hipLaunchKernelGGL(block_fwd_rows_kernel<false>, dim3(nslab4, n), dim3(NTR), smem4, (hipStream_t)stream, g1);
    PV_LAUNCH_CHECK("block_fwd_rows_kernel");
    return PV_OK;
  }
  if (gmax == 1) {                              // nothing to split: the single-workgroup kernel (64-column tiles)
    BlockFwdGroup1 g1;
    size_t smem1 = 0;
    for (int b = 0; b < PV_BLOCK_GROUP_MAX; ++b) {
      const BlockFwdArgs& a = g.a[b];
      g1.a[b] = BlockFwdArgs1{a.x, a.ldx, a.M, a.d, a.save, a.mean, a.rstd, a.out, a.ldo, a.LD};
      const size_t need = ((size_t)4 * RB * a.LD + PartLayout(a.d.Din, a.d.mid, a.d.Dout).width + 8 + 2 * RB + 32) * sizeof(float);
      if (need > smem1) smem1 = need;
    }
    static size_t seen1 = 0;
    int rc = set_lds(block_fwd_kernel, smem1, &seen1);
    if (rc) return rc;
    hipLaunchKernelGGL(block_fwd_kernel, dim3(nslab, n), dim3(NT), smem1, (hipStream_t)stream, g1);
    PV_LAUNCH_CHECK("block_fwd_kernel");
    return PV_OK;
  }
  g.slab_stride = round_up(nslab, 8);           // members of a cluster: equal blockIdx.x mod 8 -> one XCD
  static size_t seen = 0;
  int rc = set_lds(block_fwd_cluster_kernel, smem, &seen);
  if (rc) return rc;
  hipLaunchKernelGGL(block_fwd_cluster_kernel, dim3(g.slab_stride * gmax, n), dim3(NT), smem, (hipStream_t)stream, g);
  PV_LAUNCH_CHECK("block_fwd_cluster_kernel");
  return PV_OK;
}

int pv_block_rows(int rows, int wide) {
  PV_REQUIRE(rows < 0 || rows == 4 || rows == 16, "pv_block_rows: rows is 4 or 16 (negative: unchanged)");
  rows_init();
  if (rows > 0) g_rows = rows;
  if (wide >= 0) g_rows_wide = wide != 0;
  return g_rows;
}

int pv_block_sync_error(int* flag) {
  PV_REQUIRE(flag, "pv_block_sync_error: null argument");
  unsigned v = 0;
  PV_HIP(hipDeviceSynchronize());
  PV_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_block_err), sizeof(v)));
  *flag = (int)v;
  return PV_OK;
}

int pv_block_bwd(const pv_block_call* calls, int n, void* stream) { return pv_block_bwd_phase(calls, n, 3, stream); }

int pv_block_bwd_phase(const pv_block_call* calls, int n, int phase, void* stream) {
  PV_REQUIRE(calls && n >= 1 && n <= PV_BLOCK_GROUP_MAX, "pv_block_bwd: 1..%d blocks per launch", PV_BLOCK_GROUP_MAX);
  PV_REQUIRE(phase >= 1 && phase <= 3, "pv_block_bwd_phase: phase is a mask of 1 (dx chain) and 2 (parameter gradients)");
  hipStream_t st = (hipStream_t)stream;
  BlockBwdGroup g;
  ParamReduceGroup pr;
  LinGroupItem it[5 * PV_BLOCK_GROUP_MAX];
  size_t smem = 0, smem4 = 0;
  int nwg_max = 0, width_max = 0, gmax = 1;
  for (int b = 0; b < n; ++b) {
    const int G = block_cluster(calls[b].M, calls[b].Din, calls[b].mid, calls[b].Dout);
    if (G > gmax) gmax = G;
  }
  const bool rows4 = rows_mode(gmax);            // 4-row slabs: cdiv(M, 4) partial rows per Block, no clusters
  for (int b = 0; b < n; ++b) {
    const pv_block_call& c = calls[b];
    int rc = check_dims(c.M, c.Din, c.mid, c.Dout, "pv_block_bwd");
    if (rc) return rc;
    PV_REQUIRE(c.dout && c.x && c.out && c.save && c.mean && c.rstd && c.dx && c.dY && c.cpart && c.ppart && c.dgamma && c.dbeta,
               "pv_block_bwd: null argument (block %d)", b);
    PV_REQUIRE(c.lddo >= c.Dout && c.ldx >= c.Din && c.ldo >= c.Dout && c.lddx >= c.Din && c.lddo % 4 == 0 && c.ldx % 4 == 0 &&
                   c.ldo % 4 == 0 && c.lddx % 4 == 0, "pv_block_bwd: bad leading dimension");
    PV_REQUIRE(((uintptr_t)c.dout & 15) == 0 && ((uintptr_t)c.x & 15) == 0 && ((uintptr_t)c.out & 15) == 0 &&
                   ((uintptr_t)c.save & 15) == 0 && ((uintptr_t)c.dx & 15) == 0 && ((uintptr_t)c.dY & 15) == 0 &&
                   ((uintptr_t)c.ppart & 15) == 0, "pv_block_bwd: buffers must be 16-byte aligned");
    BlockBwdArgs& a = g.a[b];
    if ((rc = fill_desc(a.d, c, false, "pv_block_bwd"))) return rc;
    a.dout = c.dout; a.lddo = c.lddo; a.x = c.x; a.ldx = c.ldx; a.out = c.out; a.ldo = c.ldo; a.M = c.M; a.save = c.save;
    a.mean = c.mean; a.rstd = c.rstd; a.dx = c.dx; a.lddx = c.lddx; a.dY = c.dY; a.cpart = c.cpart; a.ppart = c.ppart;
    a.LD = max3(c.Din, c.mid, c.Dout) + 4;
    a.G = rows4 ? 1 : block_cluster(c.M, c.Din, c.mid, c.Dout);
    a.dx_bytes = (unsigned)((size_t)c.M * c.lddx * sizeof(float));
    a.dY_bytes = (unsigned)((size_t)c.M * GradLayout(c.Din, c.mid).width * sizeof(float));
    const int nwg = cdiv(c.M, rows4 ? RR : RB);
    const PartLayout P(c.Din, c.mid, c.Dout);
    const size_t need = ((size_t)4 * RB * a.LD + P.width + 8 + 2 * RB + 24 + 8) * sizeof(float);
    if (need > smem) smem = need;
    const size_t need4 = (size_t)rows_bwd_lds_floats(a.LD, P.width) * sizeof(float);
    if (need4 > smem4) smem4 = need4;
    if (nwg > nwg_max) nwg_max = nwg;
    if (P.width > width_max) width_max = P.width;

    ParamReduceArgs& r = pr.r[b];
    r.ppart = c.ppart; r.nwg = nwg; r.width = P.width; r.accumulate = c.accumulate;
    const int ends[7] = {P.b2, P.b3, P.b4, P.bt, P.dg, P.dbe, P.width};
    float* dsts[7] = {c.db[0], c.db[1], c.db[2], c.db[3], c.db[4], c.dgamma, c.dbeta};
    for (int i = 0; i < 7; ++i) {
      PV_REQUIRE(dsts[i], "pv_block_bwd: null parameter-gradient destination %d", i);
      r.seg_end[i] = ends[i];
      r.dst[i] = dsts[i];
    }
    // weight gradients: dW_i (+)= (dy_i^T a_{i-1} - <dy_i, y_i - b_i> u_i v_i^T) / sigma_i
    const SaveLayout L(c.Din, c.mid);
    const GradLayout G(c.Din, c.mid);
    const long SW = L.width, GW = G.width;
    const int outs[5] = {c.Din, c.mid, c.mid, c.Din, c.Dout}, ins[5] = {c.Din, c.Din, c.mid, c.mid, c.Din};
    const float* dys[5] = {c.dY + G.d1, c.dY + G.d2, c.dY + G.d3, c.dY + G.d4, c.dout};
    const long dyld[5] = {GW, GW, GW, GW, c.lddo};
    const float* acts[5] = {c.save + L.a0, c.save + L.a1, c.save + L.a2, c.save + L.a3, c.save + L.o};
    for (int i = 0; i < 5; ++i) {
      PV_REQUIRE(c.dW[i] && c.u[i] && c.v[i], "pv_block_bwd: null weight-gradient argument %d", i);
      it[5 * b + i] = LinGroupItem{dys[i], 1, dyld[i], acts[i], SW, 1, c.dW[i], ins[i], outs[i], ins[i], c.M,
                                   c.inv_sigma[i], c.accumulate, c.cpart + (size_t)i * nwg * a.G, nwg * a.G, c.u[i], c.v[i]};
    }
  }
  for (int b = n; b < PV_BLOCK_GROUP_MAX; ++b) {
    g.a[b] = g.a[0];
    pr.r[b] = pr.r[0];
  }
  if ((phase & 1) && rows4) {
    bool wide = false;
    for (int b = 0; b < n; ++b) wide = wide || max3(calls[b].Din, calls[b].mid, calls[b].Dout) > 256;
    static size_t seen4 = 0, seen4w = 0;
    int rc = wide ? set_lds(block_bwd_dx_rows_kernel<true>, smem4, &seen4w) : set_lds(block_bwd_dx_rows_kernel<false>, smem4, &seen4);
    if (rc) return rc;
    BlockBwdGroup1 g1;
    for (int b = 0; b < PV_BLOCK_GROUP_MAX; ++b) {
      const BlockBwdArgs& a = g.a[b];
      g1.a[b] = BlockBwdArgs1{a.dout, a.lddo, a.x, a.ldx, a.out, a.ldo, a.M, a.d, a.save, a.mean, a.rstd, a.dx, a.lddx, a.dY,
                              a.cpart, a.ppart, a.LD};
    }
    if (wide) hipLaunchKernelGGL(block_bwd_dx_rows_kernel<true>, dim3(nwg_max, n), dim3(NTR), smem4, st, g1);
    else hipLaunchKernelGGL(block_bwd_dx_rows_kernel<false>, dim3(nwg_max, n), dim3(NTR), smem4, st, g1);
    PV_LAUNCH_CHECK("block_bwd_dx_rows_kernel");
  } else if (phase & 1) {
    static size_t seen = 0, seen_c = 0;
    int rc = gmax == 1 ? set_lds(block_bwd_dx_kernel, smem, &seen) : set_lds(block_bwd_dx_cluster_kernel, smem, &seen_c);
    if (rc) return rc;
    if (gmax == 1) {
      BlockBwdGroup1 g1;
      for (int b = 0; b < PV_BLOCK_GROUP_MAX; ++b) {
        const BlockBwdArgs& a = g.a[b];
        g1.a[b] = BlockBwdArgs1{a.dout, a.lddo, a.x, a.ldx, a.out, a.ldo, a.M, a.d, a.save, a.mean, a.rstd, a.dx, a.lddx, a.dY,
                                a.cpart, a.ppart, a.LD};
      }
      hipLaunchKernelGGL(block_bwd_dx_kernel, dim3(nwg_max, n), dim3(NT), smem, st, g1);
      PV_LAUNCH_CHECK("block_bwd_dx_kernel");
    } else {
      g.slab_stride = round_up(nwg_max, 8);
      hipLaunchKernelGGL(block_bwd_dx_cluster_kernel, dim3(g.slab_stride * gmax, n), dim3(NT), smem, st, g);
      PV_LAUNCH_CHECK("block_bwd_dx_cluster_kernel");
    }
  }
  if (!(phase & 2)) return PV_OK;
  // the 5 n weight-gradient GEMMs in one launch, the bias / LayerNorm-parameter reductions riding on it
  ColReduceJob jobs[PV_BLOCK_GROUP_MAX];
  for (int b = 0; b < n; ++b) {
    const ParamReduceArgs& r = pr.r[b];
    jobs[b].part = r.ppart; jobs[b].nrows = r.nwg; jobs[b].width = r.width; jobs[b].accumulate = r.accumulate;
    for (int i = 0; i < 7; ++i) { jobs[b].seg_end[i] = r.seg_end[i]; jobs[b].dst[i] = r.dst[i]; }
  }
  return pv_linear_f32_group(it, 5 * n, st, jobs, n);
}

}  // extern "C"
