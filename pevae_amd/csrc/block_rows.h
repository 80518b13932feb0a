// 4-row slabs for the fused Block kernels (included by block_fused.hip inside its anonymous namespace; round 3).
//
// The 16-row kernels above are bound twice over by ONE compute unit: a 256 x 256 layer of a 16-row slab is 3.4 us of
// v_mfma_f32_16x16x4_f32 on its four matrix cores AND 256 KB of weights through its L1 (3.7 us), and at batch 256 only
// 16 of the chip's 256 CUs work on a Block.  Nothing in a Block couples batch rows, so the slab can shrink as far as the
// matrix core allows: v_mfma_f32_4x4x1_16b_f32 multiplies sixteen independent 4 x 4 blocks per instruction at the same
// 64 FLOP per clock -- fed the SAME four activations in every block it is a 4-row x 64-column outer-product step, still
// exact fp32.  A workgroup then owns 4 batch rows (64 workgroups per Block at batch 256: four times the matrix cores and
// four times the L1 ports on the same work), and what a layer costs its CU is the weight stream alone:
//
//   * the four compute waves split K: wave w multiplies rows [w K/4, (w+1) K/4) of Wop [K][N] -- one 1 KiB row segment
//     per load instruction (lane l: columns 4l .. 4l+3 of a 256-column super-tile), 16 of them in flight per wave, each
//     feeding four MFMAs (operand A: a[row l % 4][k] from the LDS image of the layer input, one 16-byte read per 4 k);
//   * the four partial results (4 rows x N each) meet in LDS and are summed in wave order by the thread that runs the
//     epilogue of that element (bias, 1/sigma, GELU / GELU', residual): two workgroup barriers per layer;
//   * the storer wave (wave 4) and the saved-state layout are those of the 16-row kernels, so the parameter-gradient
//     launch (pv_linear_f32_group with its ColReduceJob riders) is unchanged: it just sees M/4 partial rows.
//
// lane l of a 4x4x1 MFMA: A = value for (block l / 4, row l % 4), B = value for (block l / 4, column l % 4),
// D[vgpr r][lane l] = A[lane 4 (l / 4) + r] * B[lane l]   (tools/probe_mfma_4x4x1.hip, run on the MI355X).
#pragma once
// (included by block_fused.hip INSIDE its anonymous namespace: no #include here -- <type_traits> comes from there)

#ifndef PV_ROWS_ABL
#define PV_ROWS_ABL 0            // timing ablations (diagnostic builds only): 1 = no weight loads, 2 = no MFMAs, 3 = neither
#endif
constexpr int RR = 4;            // batch rows per workgroup
#ifndef PV_ROWS_NCW
#define PV_ROWS_NCW 4
#endif
// compute waves (they split K); wave NCW is the storer.  8 waves (twice the loads in flight) need <= 168 registers per lane:
// hipcc spills 41 of them around the k loop and the step is SLOWER (2.858 vs 2.744 ms, A/B on one box), so 4 it is.
constexpr int NCW = PV_ROWS_NCW;
constexpr int NTR = (NCW + 1) * 64, NTCR = NCW * 64;
constexpr int UR = (128 * RR + NTCR - 1) / NTCR;   // elements (of 4 columns) per compute thread of a 4 x 512 image
constexpr int RDK = 16;          // weight-row segments (1 KiB per wave each) in flight per wave (24: 256 registers, spills, slower)

// Wave w's share of K: [w kper, w kper + kw), kper = K / NCW rounded up to a multiple of 4 (kw may be short or zero)
__device__ __forceinline__ int rows_kper(int K) { return ((K + 4 * NCW - 1) / (4 * NCW)) * 4; }
__device__ __forceinline__ int rows_kw(int K, int w) { return max(0, min(rows_kper(K), K - w * rows_kper(K))); }

// Segment s of wave w's share of Wop [K][N] (N > 256: two 256-column super-tiles per k, interleaved): the first RDK of them
template <bool WIDE>
__device__ __forceinline__ void ring4_fill(const float* __restrict__ W, int N, int K, int w, int lane, f32x4 (&ring)[RDK]) {
  const int kw = rows_kw(K, w), sh = WIDE && N > 256 ? 1 : 0, nseq = kw << sh;     // (sh: log2 of the super-tile count)
  if (nseq == 0) return;
  const float* base = W + (size_t)(w * rows_kper(K)) * N;
  const int c0 = min(4 * lane, N - 4), c1 = min(256 + 4 * lane, N - 4);
#pragma unroll
  for (int t = 0; t < RDK; ++t) {
    const int s = min(t, nseq - 1);
#if PV_ROWS_ABL & 1
    ring[t] = f32x4{(float)s, 1.f, 2.f, 3.f};
#else
    ring[t] = *(const f32x4*)(base + (size_t)(s >> sh) * N + ((s & sh) ? c1 : c0));
#endif
  }
}

// acc[st][e][r] += sum over wave w's k of in[r][k] * Wop[k][256 st + 4 lane + e].  On entry the ring holds the first RDK
// segments; the slot an MFMA group has read is refilled with the segment RDK ahead (clamped: no load under a condition).
template <int NST>
__device__ __forceinline__ void kloop4(const float* __restrict__ in, int LD, const float* __restrict__ W, int N, int K, int w,
                                       int lane, f32x4 (&ring)[RDK], f32x4 (&acc)[2][4]) {
  const int kper = rows_kw(K, w), nseq = kper * NST, k0 = w * rows_kper(K);
  if (kper == 0) return;
  const float* base = W + (size_t)k0 * N;
  const int c0 = min(4 * lane, N - 4), c1 = min(256 + 4 * lane, N - 4);
  const float* ap = in + (lane & 3) * LD + k0;
  constexpr int KPB = RDK / NST;          // k per unrolled body
  // one body = RDK segments; `refill`: request the segment RDK ahead into the slot just read (the last body requests nothing:
  // the ring is about to be refilled with the next layer's segments, and re-reading the last row RDK times cost the short
  // layers half of their load traffic)
  auto body = [&](int kb, auto refill) __attribute__((always_inline)) {
    f32x4 av[KPB / 4];
#pragma unroll
    for (int q = 0; q < KPB / 4; ++q) av[q] = *(const f32x4*)(ap + min(kb + 4 * q, kper - 4));
#pragma unroll
    for (int t = 0; t < RDK; ++t) {
      const int kq = t / NST, st = t % NST;
      if (kb + kq < kper) {
        const float a = av[kq >> 2][kq & 3];
#if PV_ROWS_ABL & 2
        asm volatile("" ::"v"(ring[t]), "v"(a));
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[st][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, ring[t][e], acc[st][e], 0, 0, 0);
#endif
      }
#if !(PV_ROWS_ABL & 1)
      if constexpr (decltype(refill)::value) {
        const int sn = min(kb * NST + t + RDK, nseq - 1);
        ring[t] = *(const f32x4*)(base + (size_t)(sn / NST) * N + ((sn % NST) ? c1 : c0));
        // keep the request HERE: left alone, the scheduler sinks all RDK loads below the body's MFMAs, the ring drains
        // once per body and the load latency adds to the MFMA time instead of hiding behind it
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
  };
  int kb = 0;
  for (; kb + KPB < kper; kb += KPB) body(kb, std::true_type{});
  body(kb, std::false_type{});
}

// One layer's products of the calling compute wave, its partial result parked in red[w][RR][LD], and the next layer's
// first segments requested (Wn == nullptr: none).
template <bool WIDE>
__device__ __forceinline__ void dense4(const float* __restrict__ in, int LD, const float* __restrict__ W, int N, int K,
                                       const float* __restrict__ Wn, int Nn, int Kn, int w, int lane, f32x4 (&ring)[RDK],
                                       float* __restrict__ red) {
  f32x4 acc[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[s][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  // WIDE = false: a launch without a layer wider than 256 columns runs a kernel without the two-super-tile loop (these
  // kernels run through most of their code once: instruction fetch is part of a launch's fixed cost)
  if (WIDE && N > 256) kloop4<2>(in, LD, W, N, K, w, lane, ring, acc);
  else kloop4<1>(in, LD, W, N, K, w, lane, ring, acc);
  if (Wn) ring4_fill<WIDE>(Wn, Nn, Kn, w, lane, ring);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int col = 256 * s + 4 * lane;
    if (col < N) {
#pragma unroll
      for (int r = 0; r < RR; ++r)
        *(f32x4*)(red + (w * RR + r) * LD + col) = f32x4{acc[s][0][r], acc[s][1][r], acc[s][2][r], acc[s][3][r]};
    }
  }
}
// the layer's result for (row, col .. col + 3): the compute waves' partials in wave order
__device__ __forceinline__ f32x4 red4(const float* __restrict__ red, int LD, int row, int col) {
  f32x4 v = *(const f32x4*)(red + row * LD + col);
#pragma unroll
  for (int q = 1; q < NCW; ++q) v += *(const f32x4*)(red + (q * RR + row) * LD + col);
  return v;
}

// row of element i of a [RR][per] image (i < RR * per): three compares instead of an integer division (a division by a
// run-time value is ~40 instructions, and these kernels run through their code once)
__device__ __forceinline__ int row_of(int i, int per) { return (i >= per) + (i >= 2 * per) + (i >= 3 * per); }
static_assert(RR == 4, "row_of() is written for four rows");

// biases / gamma / beta -> LDS (PartLayout offsets) + the five 1/sigma: every load of the wave issued before the first LDS
// write (one memory round trip), segment by segment -- a segment is at most 512 floats = two 16-byte pieces per lane --
// instead of block_fused.hip's chunk-major walk with a seven-way source select per piece (530 instructions there, 18 % of
// this kernel's code, which a workgroup runs through once: instruction fetch is part of the launch's fixed cost)
__device__ __forceinline__ void wave_stage_params_rows(const BlockDesc& d, const PartLayout& P, float* par, int lane) {
  const float* src[7] = {d.bias[0], d.bias[1], d.bias[2], d.bias[3], d.bias[4], d.gamma, d.beta};
  const int beg[8] = {P.b1, P.b2, P.b3, P.b4, P.bt, P.dg, P.dbe, P.width};
  f32x4 v[7][2];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int n4 = (beg[k + 1] - beg[k]) >> 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) v[k][j] = *(const f32x4*)(src[k] + 4 * min(lane + 64 * j, n4 - 1));
  }
  float isv = 0.f;
  if (lane < 5) isv = d.inv_sigma[lane][0];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int n4 = (beg[k + 1] - beg[k]) >> 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (lane + 64 * j < n4) *(f32x4*)(par + beg[k] + 4 * (lane + 64 * j)) = v[k][j];
  }
  if (lane < 5) par[P.width + lane] = isv;
}

__device__ __forceinline__ void wave_copy_rows(const float* __restrict__ buf, int LD, float* __restrict__ g, size_t gs,
                                               int ncols, int r0, int M, int lane) {
  const int per = ncols / 4, nq = RR * per;
  for (int i = lane; i < nq; i += 64) {
    const int row = row_of(i, per), c4 = (i - row * per) * 4;
    if (r0 + row < M) *(f32x4*)(g + (size_t)(r0 + row) * gs + c4) = *(const f32x4*)(buf + row * LD + c4);
  }
}
__device__ __forceinline__ void wave_colsum_rows(const float* __restrict__ buf, int LD, int N, float* __restrict__ dst, int lane) {
  for (int c = lane * 4; c < N; c += 256) {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RR; ++r) sum += *(const f32x4*)(buf + r * LD + c);
    *(f32x4*)(dst + c) = sum;
  }
}

// LDS floats of the two kernels (host and device agree through these)
__host__ __device__ inline int rows_fwd_lds_floats(int LD, int part_width) { return (5 + NCW) * RR * LD + part_width + 8 + 2 * RR + 32 + 8; }
__host__ __device__ inline int rows_bwd_lds_floats(int LD, int part_width) { return (7 + NCW) * RR * LD + part_width + 8 + 2 * RR + 5 * NCW + 8; }

// (Tried and dropped: the storer wave touching one dword per 128-byte line of the coming layers' weights, as LDS-DMA into a
// dummy area, to warm the L2 and the TLBs ahead of the compute waves -- every touch pulls its whole line through the same
// L1 the weight stream is bound by: Block(256,128,128) forward 20.3 -> 27.6 us.)
template <bool WIDE>
__global__ __launch_bounds__(NTR) void block_fwd_rows_kernel(BlockFwdGroup1 grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BlockFwdArgs1& a = grp.a[blockIdx.y];
  if ((int)blockIdx.x * RR >= a.M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Din = a.d.Din, mid = a.d.mid, Dout = a.d.Dout, LD = a.LD, M = a.M;
  const int r0 = blockIdx.x * RR;
  float* buf0 = lds;
  float* buf1 = lds + RR * LD;
  float* yb0 = lds + 2 * RR * LD;                 // the layer's pre-activation y, parked for the storer wave
  float* yb1 = lds + 3 * RR * LD;
  float* xb = lds + 4 * RR * LD;                  // the raw input rows (residual)
  float* red = lds + 5 * RR * LD;                 // [NCW waves][RR][LD] partial results
  float* par = lds + (5 + NCW) * RR * LD;
  const SaveLayout L(Din, mid);
  const PartLayout P(Din, mid, Dout);
  const float* isg = par + P.width;
  float* rowstat = par + P.width + 8;             // mean[RR], rstd[RR]
  const size_t SW = L.width;
  BSTAMP_DECL(par + P.width + 8 + 2 * RR)
  BSTAMP(0, 0);
#ifdef PV_BLOCK_STAMPS
  if (threadIdx.x == 0) bst[14] = clock64();
#endif
  // per layer: N (outputs), K (inputs)
  auto Nof = [&](int li) { return li == 0 || li == 3 ? Din : (li == 4 ? Dout : mid); };
  auto Kof = [&](int li) { return li == 0 || li == 1 || li == 4 ? Din : mid; };

  if (w == NCW) {
    // ---- the storer wave: parameters -> LDS, L2 warming, and every global store of saved state, one layer behind ----
    wave_stage_params_rows(a.d, P, par, lane);
    lds_barrier();
    for (int li = 0; li < 5; ++li) {
      const float* in = (li & 1) || li == 4 ? buf1 : buf0;
      if (li == 0) {
        wave_copy_rows(buf0, LD, a.save + L.a0, SW, Din, r0, M, lane);
      } else if (li < 4) {
        const int pN = li == 1 ? Din : mid;
        const int off_y = li == 1 ? L.y1 : li == 2 ? L.y2 : L.y3, off_a = li == 1 ? L.a1 : li == 2 ? L.a2 : L.a3;
        wave_copy_rows((li & 1) ? yb0 : yb1, LD, a.save + off_y, SW, pN, r0, M, lane);
        wave_copy_rows(in, LD, a.save + off_a, SW, pN, r0, M, lane);
      } else {
        wave_copy_rows(buf1, LD, a.save + L.o, SW, Din, r0, M, lane);
        if (lane < RR && r0 + lane < M) {
          a.mean[r0 + lane] = rowstat[lane];
          a.rstd[r0 + lane] = rowstat[RR + lane];
        }
      }
      lds_barrier();                               // (the compute waves' partial results are in LDS)
      if (li == 4) break;
      lds_barrier();                               // (their epilogue has written the next layer's input)
      if (li == 3) {
        wave_copy_rows(yb1, LD, a.save + L.y4, SW, Din, r0, M, lane);
        lds_barrier();
      }
    }
    return;
  }

  // ---- the four compute waves ----
  f32x4 ring[RDK];
  ring4_fill<WIDE>(a.d.W[0], Din, Din, w, lane, ring);
  {
    const int per = Din / 4, nq = RR * per;       // <= 512: UR per thread
    f32x4 xv[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = min(tid + u * NTCR, nq - 1), row = row_of(i, per);
      xv[u] = *(const f32x4*)(a.x + (size_t)min(r0 + row, M - 1) * a.ldx + (i - row * per) * 4);
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = tid + u * NTCR;
      if (i < nq) {
        const int row = row_of(i, per), c4 = (i - row * per) * 4;
        const f32x4 v = xv[u];
        *(f32x4*)(xb + row * LD + c4) = v;
        *(f32x4*)(buf0 + row * LD + c4) = f32x4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
      }
    }
  }
  lds_barrier();
  BSTAMP(0, 1);

  for (int li = 0; li < 5; ++li) {
    const int N = Nof(li), K = Kof(li);
    const float* in = (li & 1) || li == 4 ? buf1 : buf0;
    float* outb = (li & 1) ? buf0 : buf1;
    float* yb = (li & 1) ? yb1 : yb0;
    dense4<WIDE>(in, LD, a.d.W[li], N, K, li < 4 ? a.d.W[li + 1] : nullptr, li < 4 ? Nof(li + 1) : 0, li < 4 ? Kof(li + 1) : 0, w, lane,
           ring, red);
    BSTAMP(0, 2 + 2 * li);
    lds_barrier();                                 // the four partial results are in LDS
    {
      const int off_b = li == 0 ? P.b1 : li == 1 ? P.b2 : li == 2 ? P.b3 : li == 3 ? P.b4 : P.bt;
      const float is = isg[li];
      const int per = N >> 2, nit = RR * per;
      for (int i = tid; i < nit; i += NTCR) {
        const int row = row_of(i, per), col = (i - row * per) * 4, gr = r0 + row;
        const f32x4 y = red4(red, LD, row, col) * is + *(const f32x4*)(par + off_b + col);
        if (li < 3) {
          *(f32x4*)(yb + row * LD + col) = y;
          *(f32x4*)(outb + row * LD + col) = f32x4{gelu_f(y[0]), gelu_f(y[1]), gelu_f(y[2]), gelu_f(y[3])};
        } else if (li == 3) {
          *(f32x4*)(yb + row * LD + col) = y;
          *(f32x4*)(buf0 + row * LD + col) = *(const f32x4*)(xb + row * LD + col) + y;
        } else if (gr < M) {
          *(f32x4*)(a.out + (size_t)gr * a.ldo + col) = y;
        }
      }
    }
    if (li == 4) break;
    lds_barrier();
    BSTAMP(0, 3 + 2 * li);
    if (li == 3) {
      // LayerNorm of s = x + y4 (buf0; biased variance, two passes as nn.LayerNorm) -> buf1: wave w, row w
      if (w < RR) {
      const float* srow = buf0 + w * LD;
      float sm = 0.f;
      for (int c = lane * 4; c < Din; c += 256) {
        const f32x4 v = *(const f32x4*)(srow + c);
        sm += (v[0] + v[1]) + (v[2] + v[3]);
      }
      const float mu = wave_sum(sm) / Din;
      float q = 0.f;
      for (int c = lane * 4; c < Din; c += 256) {
        const f32x4 v = *(const f32x4*)(srow + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) q += (v[e] - mu) * (v[e] - mu);
      }
      const float rs = rsqrtf(wave_sum(q) / Din + a.d.eps);
      for (int c = lane * 4; c < Din; c += 256) {
        const f32x4 v = *(const f32x4*)(srow + c), gm = *(const f32x4*)(par + P.dg + c), bt = *(const f32x4*)(par + P.dbe + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu) * rs * gm[e] + bt[e];
        *(f32x4*)(buf1 + w * LD + c) = o;
      }
      if (lane == 0) {
        rowstat[w] = mu;
        rowstat[RR + w] = rs;
      }
      }
      lds_barrier();
      BSTAMP(0, 12);
    }
  }
  BSTAMP(0, 11);
#ifdef PV_BLOCK_STAMPS
  if (threadIdx.x == 0) bst[15] = clock64();
#endif
  BSTAMP_FLUSH(0);
}

template <bool WIDE>
__global__ __launch_bounds__(NTR) void block_bwd_dx_rows_kernel(BlockBwdGroup1 grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BlockBwdArgs1& a = grp.a[blockIdx.y];
  if ((int)blockIdx.x * RR >= a.M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Din = a.d.Din, mid = a.d.mid, Dout = a.d.Dout, LD = a.LD, M = a.M;
  const int r0 = blockIdx.x * RR, nwg = (a.M + RR - 1) / RR;
  float* g0 = lds;                                 // dL/dout rows
  float* bdo = lds + RR * LD;                      // do = dout W_t / sigma_t (LayerNorm output gradient)
  float* bdxh = lds + 2 * RR * LD;                 // do * xhat (dgamma)
  float* bds = lds + 3 * RR * LD;                  // ds = dy4 (also the residual branch of dx)
  float* pA = lds + 4 * RR * LD;                   // dy3, dy1
  float* pB = lds + 5 * RR * LD;                   // dy2
  float* sbuf = lds + 6 * RR * LD;                 // s = x + y4 (the LayerNorm input)
  float* red = lds + 7 * RR * LD;
  float* par = lds + (7 + NCW) * RR * LD;
  const SaveLayout L(Din, mid);
  const GradLayout G(Din, mid);
  const PartLayout P(Din, mid, Dout);
  const float* isg = par + P.width;
  float* rowstat = par + P.width + 8;              // mean[RR], rstd[RR]
  float (*cred)[5] = (float (*)[5])(par + P.width + 8 + 2 * RR);   // [NCW waves][5]: <dy_i, y_i - b_i>, i = c1..c4, translate
  const size_t SW = L.width, GW = G.width;
  float* prow = a.ppart + (size_t)blockIdx.x * P.width;
  // step st multiplies by the weight of linear 4 - st ([K][N] = weight_orig [out][in])
  auto Nof = [&](int st) { return st == 0 || st >= 3 ? Din : mid; };
  auto Kof = [&](int st) { return st == 0 ? Dout : (st == 1 || st == 4 ? Din : mid); };
  auto inof = [&](int st) { return st == 0 ? g0 : st == 1 ? bds : st == 2 ? pA : st == 3 ? pB : pA; };

  if (w == NCW) {
    // ---- the storer wave: one step behind, gradients of the inner linears -> dY, column sums -> this workgroup's partial row
    wave_stage_params_rows(a.d, P, par, lane);
    if (lane < RR) {
      const int grc = min(r0 + lane, M - 1);
      rowstat[lane] = a.mean[grc];
      rowstat[RR + lane] = a.rstd[grc];
    }
    lds_barrier();
    for (int st = 0; st < 5; ++st) {
      const float* in = inof(st);
      if (st == 0) {
        wave_colsum_rows(g0, LD, Dout, prow + P.bt, lane);
      } else if (st == 1) {
        wave_copy_rows(bds, LD, a.dY + G.d4, GW, Din, r0, M, lane);
        wave_colsum_rows(bdxh, LD, Din, prow + P.dg, lane);
        wave_colsum_rows(bdo, LD, Din, prow + P.dbe, lane);
        wave_colsum_rows(bds, LD, Din, prow + P.b4, lane);
      } else {
        // dy of the previous step sits in this step's input
        const int pN = st == 4 ? Din : mid;
        const int off_d = st == 2 ? G.d3 : st == 3 ? G.d2 : G.d1, off_pb = st == 2 ? P.b3 : st == 3 ? P.b2 : P.b1;
        wave_copy_rows(in, LD, a.dY + off_d, GW, pN, r0, M, lane);
        wave_colsum_rows(in, LD, pN, prow + off_pb, lane);
      }
      lds_barrier();
      lds_barrier();
      if (st == 0) lds_barrier();
    }
    return;
  }

  // ---- the four compute waves ----
  f32x4 ring[RDK];
  ring4_fill<WIDE>(a.d.W[4], Din, Dout, w, lane, ring);
  {
    const int pero = Dout / 4, nqo = RR * pero, peri = Din / 4, nqi = RR * peri;
    f32x4 dv[UR], ov[UR], bv[UR], xv[UR], yv[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = min(tid + u * NTCR, nqo - 1), ro = row_of(i, pero), gr = min(r0 + ro, M - 1), c4 = (i - ro * pero) * 4;
      dv[u] = *(const f32x4*)(a.dout + (size_t)gr * a.lddo + c4);
      ov[u] = *(const f32x4*)(a.out + (size_t)gr * a.ldo + c4);
      bv[u] = *(const f32x4*)(a.d.bias[4] + c4);
      const int k = min(tid + u * NTCR, nqi - 1), rk = row_of(k, peri), gk = min(r0 + rk, M - 1), k4 = (k - rk * peri) * 4;
      xv[u] = *(const f32x4*)(a.x + (size_t)gk * a.ldx + k4);
      yv[u] = *(const f32x4*)(a.save + (size_t)gk * SW + L.y4 + k4);
    }
    float c4s = 0.f;
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = tid + u * NTCR;
      if (i < nqo) {
        const int row = row_of(i, pero);
        f32x4 v = dv[u];
        if (r0 + row >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) c4s += v[e] * (ov[u][e] - bv[u][e]);
        *(f32x4*)(g0 + row * LD + (i - row * pero) * 4) = v;
      }
      if (i < nqi) {
        const int row = row_of(i, peri);
        *(f32x4*)(sbuf + row * LD + (i - row * peri) * 4) = xv[u] + yv[u];
      }
    }
    c4s = wave_sum(c4s);
    if (lane == 0) cred[w][4] = c4s;
  }
  lds_barrier();

  // step 0: do = dout W_t / sigma_t, then the LayerNorm backward (ds = dy4);
  // steps 1..3: dy_i = (dy_{i+1} W_{i+1} / sigma_{i+1}) * gelu'(y_i) for y3, y2, y1;
  // step 4: dx = ds + (dy1 W_1 / sigma_1) * gelu'(x), and the c4 spectral term (y4 - b4 = s - x - b4).
  for (int st = 0; st < 5; ++st) {
    const int li = 4 - st;                                  // the linear whose weight this step multiplies by
    const int N = Nof(st), K = Kof(st);
    const float* in = inof(st);
    float* outb = st == 0 ? bdo : st == 1 ? pA : st == 2 ? pB : pA;
    const int per = N >> 2, nit = RR * per;
    // what the epilogue multiplies by gelu'(.): saved y_i (steps 1..3) or x (step 4), requested before the k loop
    const int off_y = st == 1 ? L.y3 : st == 2 ? L.y2 : L.y1;
    const float* pbase = st == 4 ? a.x : a.save + off_y;
    const long pstride = st == 4 ? a.ldx : (long)SW;
    f32x4 ys[UR];
    if (st > 0) {
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int i = min(tid + u * NTCR, nit - 1), row = row_of(i, per);
        ys[u] = *(const f32x4*)(pbase + (size_t)min(r0 + row, M - 1) * pstride + (i - row * per) * 4);
      }
    }
    dense4<WIDE>(in, LD, a.d.W[li], N, K, st < 4 ? a.d.W[li - 1] : nullptr, st < 4 ? Nof(st + 1) : 0, st < 4 ? Kof(st + 1) : 0, w, lane,
           ring, red);
    lds_barrier();                               // the four partial results are in LDS
    const int off_b = st == 1 ? P.b3 : st == 2 ? P.b2 : st == 3 ? P.b1 : P.b4;
    const float is = isg[li];
    float csum = 0.f;
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = tid + u * NTCR;
      if (i >= nit) continue;
      const int row = row_of(i, per), col = (i - row * per) * 4, gr = r0 + row;
      const f32x4 acc = red4(red, LD, row, col) * is;
      if (st == 0) {
        *(f32x4*)(outb + row * LD + col) = acc;
        continue;
      }
      const f32x4 bv = *(const f32x4*)(par + off_b + col);
      if (st < 4) {
        f32x4 dy = {0.f, 0.f, 0.f, 0.f};
        if (gr < M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dy[e] = acc[e] * gelu_df(ys[u][e]);
            csum += dy[e] * (ys[u][e] - bv[e]);
          }
        }
        *(f32x4*)(outb + row * LD + col) = dy;
      } else if (gr < M) {
        const f32x4 ds = *(const f32x4*)(bds + row * LD + col), sv = *(const f32x4*)(sbuf + row * LD + col);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r[e] = ds[e] + acc[e] * gelu_df(ys[u][e]);
          csum += ds[e] * (sv[e] - ys[u][e] - bv[e]);
        }
        *(f32x4*)(a.dx + (size_t)gr * a.lddx + col) = r;
      }
    }
    if (st > 0) {
      csum = wave_sum(csum);
      if (lane == 0) cred[w][st == 4 ? 3 : 3 - st] = csum;
    }
    lds_barrier();
    if (st == 0) {
      // LayerNorm backward, wave w on row w: ds = rstd (g - mean(g) - xhat mean(g xhat)), g = do * gamma
      if (w < RR) {
      const float mu = rowstat[w], rs = rowstat[RR + w];
      float s1 = 0.f, s2 = 0.f;
      for (int c = lane * 4; c < Din; c += 256) {
        const f32x4 sv = *(const f32x4*)(sbuf + w * LD + c), dv = *(const f32x4*)(bdo + w * LD + c),
                    gm = *(const f32x4*)(par + P.dg + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = dv[e] * gm[e];
          s1 += g;
          s2 += g * ((sv[e] - mu) * rs);
        }
      }
      s1 = wave_sum(s1) / Din;
      s2 = wave_sum(s2) / Din;
      for (int c = lane * 4; c < Din; c += 256) {
        const f32x4 sv = *(const f32x4*)(sbuf + w * LD + c), dv = *(const f32x4*)(bdo + w * LD + c),
                    gm = *(const f32x4*)(par + P.dg + c);
        f32x4 dsv, dxh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (sv[e] - mu) * rs;
          dsv[e] = rs * (dv[e] * gm[e] - s1 - xh * s2);       // rows past M: do = 0 -> ds = 0
          dxh[e] = dv[e] * xh;
        }
        *(f32x4*)(bdxh + w * LD + c) = dxh;
        *(f32x4*)(bds + w * LD + c) = dsv;
      }
      }
      lds_barrier();
    }
  }
  if (tid < 5) {
    float c = cred[0][tid];
#pragma unroll
    for (int q = 1; q < NCW; ++q) c += cred[q][tid];
    a.cpart[tid * nwg + blockIdx.x] = c;
  }
}
