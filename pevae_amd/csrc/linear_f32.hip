// Small fp32 strided GEMM (f32-input MFMA) for the latent / connector linears of the DLS_VAE step (M = batch = 256,
// K,N <= a few hundred: Hidden2Gaussian nn_lib.py:198-199, LinearConnector nn_lib.py:143, the Block MLPs
// dialog_models.py:258-263).  These are launch-latency bound, not FLOP bound, and they feed KL terms that are
// compared with the reference at fp32 tolerance, so they stay in fp32 FMA (no bf16 rounding).
//
//   C[m*ldc + n] (+)= alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]  (+ bias[n])
//
// Arbitrary element strides make forward (x W^T), data-grad (dy W) and weight-grad (dy^T x) the same kernel;
// `batch` + batch strides give the bmm forms of the attention block (decoders.py:85-107).
#include "common.h"
#include "linear_group.h"
#include "../../include/pevae_hip.h"

namespace {

// 64x64 output tile per 256-thread workgroup, 4 waves as 2x2, each wave 32x32 = 2x2 v_mfma_f32_16x16x4_f32
// accumulators (exact fp32 FMA chain in k order).  K is walked 32 deep; the next K-tile's global loads are
// issued BEFORE the current tile's MFMAs and written to the other LDS buffer after them (one barrier per
// K-tile), so the global latency of these tiny, launch/latency-bound GEMMs overlaps the math.
// The LDS image is [k][m] (row stride 80 floats: the 4 k-rows of one MFMA fragment land on disjoint banks),
// filled from 16-byte global loads along whichever direction has unit stride.
constexpr int TK = 32;
constexpr int MODE_SCALAR = 0, MODE_XFAST = 1, MODE_KFAST = 2;

struct LinArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  int M, N, K;
  long sam, sak, sbk, sbn, ldc;
  long bsa, bsb, bsc;   // batch strides (0 = shared)
  float alpha;
  const float* alpha_dev;   // optional device scalar multiplied into alpha (1/sigma of a spectral-norm weight)
  int accumulate;
  int modeA, modeB;
  float* gelu_out;          // optional [M][N] (ld = N): also write gelu(C) (the next Block linear's input, erf form)
  const float* gelu_grad;   // optional [M][N] (ld = N): C = result * gelu'(gelu_grad[m][n])  (backward through the GELU)
  const float* r1_c;        // optional rank-1 correction (spectral-norm weight gradient):
  const float* r1_u;        //   C -= (sum of r1_c[0..r1_nc)) * alpha_dev[0] * r1_u[m] * r1_v[n]
  const float* r1_v;
  int r1_nc;
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}


// P[x*sx + k*sk] for x in [x0, x0+T), k in [k0, k0+32) -> T/8 registers per thread (T = tile edge, 64 or 32).
// No load sits under a per-lane condition (hipcc would branch around each one and wait for it: dependent L2 round
// trips per tile): interior tiles (a workgroup-uniform test) use unconditional 16-byte loads, edge tiles load from
// clamped, always-valid addresses and zero the out-of-range values afterwards.
template <int mode, int T>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, long sx, long sk, int X, int K, int x0, int k0,
                                          int tid, float (&r)[T / 8]) {
  constexpr int NR = T / 8;
  const bool interior = x0 + T <= X && k0 + TK <= K;
  if (mode == MODE_XFAST && interior) {
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      const int e = tid + p * 256, x = x0 + (e % (T / 4)) * 4, k = k0 + e / (T / 4);
      const float4 v = *(const float4*)(P + x + (long)k * sk);
      r[p * 4 + 0] = v.x; r[p * 4 + 1] = v.y; r[p * 4 + 2] = v.z; r[p * 4 + 3] = v.w;
    }
  } else if (mode == MODE_KFAST && interior) {
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      const int e = tid + p * 256, x = x0 + e % T, k = k0 + (e / T) * 4;
      const float4 v = *(const float4*)(P + (long)x * sx + k);
      r[p * 4 + 0] = v.x; r[p * 4 + 1] = v.y; r[p * 4 + 2] = v.z; r[p * 4 + 3] = v.w;
    }
  } else {
    // element i of this thread, in the register order tile_store expects for `mode`
    float v[NR];
    bool ok[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      int x, k;
      if (mode == MODE_XFAST) {
        const int e = tid + (i >> 2) * 256;
        x = x0 + (e % (T / 4)) * 4 + (i & 3);
        k = k0 + e / (T / 4);
      } else if (mode == MODE_KFAST) {
        const int e = tid + (i >> 2) * 256;
        x = x0 + e % T;
        k = k0 + (e / T) * 4 + (i & 3);
      } else {
        const int e = tid + i * 256;
        x = x0 + e % T;
        k = k0 + e / T;
      }
      ok[i] = x < X && k < K;
      v[i] = P[(long)min(x, X - 1) * sx + (long)min(k, K - 1) * sk];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = ok[i] ? v[i] : 0.f;
  }
}

template <int mode, int T>
__device__ __forceinline__ void tile_store(float* __restrict__ S, int tid, const float (&r)[T / 8]) {
  constexpr int NR = T / 8, LS = T + 16;
  if (mode == MODE_XFAST) {
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      const int e = tid + p * 256;
      *(float4*)(S + (e / (T / 4)) * LS + (e % (T / 4)) * 4) = make_float4(r[p * 4], r[p * 4 + 1], r[p * 4 + 2], r[p * 4 + 3]);
    }
  } else if (mode == MODE_KFAST) {
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      const int e = tid + p * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) S[((e / T) * 4 + q) * LS + e % T] = r[p * 4 + q];
    }
  } else {
#pragma unroll
    for (int p = 0; p < NR; ++p) {
      const int e = tid + p * 256;
      S[(e / T) * LS + e % T] = r[p];
    }
  }
}

// MA / MB: load pattern of the A / B operand (compile time: each instantiation is straight-line staging code).
// T: tile edge.  T x T outputs per 256-thread workgroup, 4 waves as 2x2, each wave (T/2)x(T/2) in 16x16
// v_mfma_f32_16x16x4_f32 accumulators (exact fp32 FMA chain in k order).  The f32 MFMA runs at 1/16 of the bf16 rate,
// so these batch-256 GEMMs are bound by how many CUs they occupy: T = 32 quadruples the workgroup count and is used
// whenever 64x64 tiles would leave most of the chip idle.
template <int MA, int MB, int T>
__device__ __forceinline__ void linear_tile(const LinArgs& g, const int m0, const int n0, const int z,
                                            float (&lds)[2][2][TK * (T + 16)]) {
  constexpr int LS = T + 16, NR = T / 8, NF = T / 32;       // NF x NF 16x16 tiles per wave
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  const float* A = g.A + (long)z * g.bsa;
  const float* B = g.B + (long)z * g.bsb;
  float* C = g.C + (long)z * g.bsc;
  f32x4 acc[NF][NF];
#pragma unroll
  for (int t = 0; t < NF; ++t)
#pragma unroll
    for (int u = 0; u < NF; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  // FOUR K-tiles of global loads stay in flight (these GEMMs are a handful of dependent memory round trips long, so
  // latency, not bandwidth, sets their time): tile t lives in register set t % 4 from its request until it is written
  // to LDS buffer t % 2 one iteration before it is multiplied.
  float ra0[NR], rb0[NR], ra1[NR], rb1[NR], ra2[NR], rb2[NR], ra3[NR], rb3[NR];
  const int nk = (g.K + TK - 1) / TK;
  auto fetch = [&](int kt, float (&ra)[NR], float (&rb)[NR]) {
    if (kt < nk) {
      tile_load<MA, T>(A, g.sam, g.sak, g.M, g.K, m0, kt * TK, tid, ra);
      tile_load<MB, T>(B, g.sbn, g.sbk, g.N, g.K, n0, kt * TK, tid, rb);
    }
  };
  auto stash = [&](int kt, float (&ra)[NR], float (&rb)[NR]) {      // registers -> LDS buffer kt & 1
    if (kt < nk) {
      tile_store<MA, T>(lds[kt & 1][0], tid, ra);
      tile_store<MB, T>(lds[kt & 1][1], tid, rb);
    }
  };
  auto multiply = [&](int kt) {
    const float* As = lds[kt & 1][0] + wm * (T / 2) + i16;
    const float* Bs = lds[kt & 1][1] + wn * (T / 2) + i16;
#pragma unroll
    for (int ks = 0; ks < TK / 4; ++ks) {
      const int kr = (ks * 4 + kq) * LS;
      float a[NF], b[NF];
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        a[t] = As[kr + 16 * t];
        b[t] = Bs[kr + 16 * t];
      }
#pragma unroll
      for (int t = 0; t < NF; ++t)
#pragma unroll
        for (int u = 0; u < NF; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
    }
  };
  fetch(0, ra0, rb0);
  fetch(1, ra1, rb1);
  fetch(2, ra2, rb2);
  fetch(3, ra3, rb3);
  stash(0, ra0, rb0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 4) {
    fetch(kt + 4, ra0, rb0);          // set 0 is free: tile kt already sits in LDS
    multiply(kt);
    stash(kt + 1, ra1, rb1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    fetch(kt + 5, ra1, rb1);
    multiply(kt + 1);
    stash(kt + 2, ra2, rb2);
    __syncthreads();
    if (kt + 2 >= nk) break;
    fetch(kt + 6, ra2, rb2);
    multiply(kt + 2);
    stash(kt + 3, ra3, rb3);
    __syncthreads();
    if (kt + 3 >= nk) break;
    fetch(kt + 7, ra3, rb3);
    multiply(kt + 3);
    stash(kt + 4, ra0, rb0);
    __syncthreads();
  }
  const float alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
  float r1 = 0.f;
  if (g.r1_c) {
    // sum of the per-workgroup partials, the same fixed order in every wave of every tile: lane l adds partials l, l + 64, ...,
    // then a butterfly over the lanes (64 partials per Block with 4-row slabs: a serial loop of dependent loads cost ~10 us)
    for (int i = tid & 63; i < g.r1_nc; i += 64) r1 += g.r1_c[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r1 += __shfl_xor(r1, o, 64);
    r1 *= g.alpha_dev ? g.alpha_dev[0] : 1.f;
  }
#pragma unroll
  for (int u = 0; u < NF; ++u) {
    const int n = n0 + wn * (T / 2) + u * 16 + i16;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < NF; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * (T / 2) + t * 16 + kq * 4 + r;
        if (m >= g.M) continue;
        float v = alpha * acc[t][u][r] + bv;
        if (g.gelu_grad) v *= gelu_df(g.gelu_grad[(long)m * g.N + n]);
        if (g.r1_c) v -= r1 * g.r1_u[m] * g.r1_v[n];
        float* dst = C + (long)m * g.ldc + n;
        if (g.accumulate) v += *dst;
        *dst = v;
        if (g.gelu_out) g.gelu_out[(long)m * g.N + n] = gelu_f(v);
      }
  }
}

template <int MA, int MB, int T>
__global__ __launch_bounds__(256) void linear_f32_kernel(LinArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][TK * (T + 16)];      // [buffer][A|B][k][x]
  linear_tile<MA, MB, T>(g, blockIdx.y * T, blockIdx.x * T, blockIdx.z, lds);
}

// Several independent GEMMs of the x-fast / x-fast operand form (the weight gradients dy^T a of one Block) in ONE
// launch: blockIdx.x walks the 32x32 tiles of item 0, then item 1, ...
constexpr int GROUP_MAX = 15;
struct LinGroup {
  LinArgs g[GROUP_MAX];
  int tile_end[GROUP_MAX];     // exclusive prefix of tile counts
  int ntn[GROUP_MAX];          // tiles along N
  int n;
  ColReduceJob jobs[COLREDUCE_MAX];
  int job_end[COLREDUCE_MAX];  // exclusive prefix of the riders' workgroup counts, after the GEMM tiles
  int njobs;
};

__global__ __launch_bounds__(256) void linear_f32_group_kernel(LinGroup G) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][TK * 48];
  const int gemm_tiles = G.tile_end[G.n - 1];
  if ((int)blockIdx.x >= gemm_tiles) {                  // a rider: 256 columns of one column-reduce job
    int j = 0;
    while (j + 1 < G.njobs && (int)blockIdx.x - gemm_tiles >= G.job_end[j]) ++j;
    const ColReduceJob& a = G.jobs[j];
    const int c = ((int)blockIdx.x - gemm_tiles - (j ? G.job_end[j - 1] : 0)) * 256 + threadIdx.x;
    if (c >= a.width) return;
    // fixed order: eight interleaved partial sums (independent loads in flight), then their sum
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int g = 0;
    for (; g + 8 <= a.nrows; g += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] += a.part[(size_t)(g + j) * a.width + c];
    }
    for (int j = 0; g < a.nrows; ++g, ++j) p[j] += a.part[(size_t)g * a.width + c];
    const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    int seg = 0;
    while (c >= a.seg_end[seg]) ++seg;
    float* d = a.dst[seg] + (c - (seg ? a.seg_end[seg - 1] : 0));
    *d = a.accumulate ? *d + s : s;
    return;
  }
  int i = 0;
  while (i + 1 < G.n && (int)blockIdx.x >= G.tile_end[i]) ++i;
  const int local = blockIdx.x - (i ? G.tile_end[i - 1] : 0);
  linear_tile<MODE_XFAST, MODE_XFAST, 32>(G.g[i], (local / G.ntn[i]) * 32, (local % G.ntn[i]) * 32, 0, lds);
}

void launch_linear(const LinArgs& g, int batch, hipStream_t st) {
  // 32x32 tiles when 64x64 tiles would occupy fewer than ~half of the CUs
  const bool small = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * batch < 128;
  const int T = small ? 32 : 64;
  const dim3 grid(cdiv(g.N, T), cdiv(g.M, T), batch);
#define PV_LIN_CASE(a, b)                                                                           \
  if (g.modeA == a && g.modeB == b) {                                                               \
    if (small) hipLaunchKernelGGL((linear_f32_kernel<a, b, 32>), grid, dim3(256), 0, st, g);        \
    else hipLaunchKernelGGL((linear_f32_kernel<a, b, 64>), grid, dim3(256), 0, st, g);              \
    return;                                                                                         \
  }
  PV_LIN_CASE(0, 0) PV_LIN_CASE(0, 1) PV_LIN_CASE(0, 2)
  PV_LIN_CASE(1, 0) PV_LIN_CASE(1, 1) PV_LIN_CASE(1, 2)
  PV_LIN_CASE(2, 0) PV_LIN_CASE(2, 1) PV_LIN_CASE(2, 2)
#undef PV_LIN_CASE
}

// which 16-byte load pattern an operand P[x*sx + k*sk] admits
int pick_mode(const float* P, long sx, long sk, long bs, int batch) {
  const bool al = ((uintptr_t)P & 15) == 0 && (batch == 1 || bs % 4 == 0);
  if (al && sx == 1 && sk % 4 == 0) return MODE_XFAST;
  if (al && sk == 1 && sx % 4 == 0) return MODE_KFAST;
  return MODE_SCALAR;
}

}  // namespace

int pv_linear_f32_group(const LinGroupItem* it, int n, hipStream_t st, const ColReduceJob* jobs, int njobs) {
  PV_REQUIRE(it && n > 0 && n <= GROUP_MAX, "pv_linear_f32_group: bad item count %d", n);
  PV_REQUIRE(njobs >= 0 && njobs <= COLREDUCE_MAX && (njobs == 0 || jobs), "pv_linear_f32_group: bad rider count %d", njobs);
  LinGroup G;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const LinGroupItem& t = it[i];
    PV_REQUIRE(t.A && t.B && t.C && t.M > 0 && t.N > 0 && t.K > 0 && t.ldc >= t.N, "pv_linear_f32_group: bad item %d", i);
    PV_REQUIRE(pick_mode(t.A, t.sam, t.sak, 0, 1) == MODE_XFAST && pick_mode(t.B, t.sbn, t.sbk, 0, 1) == MODE_XFAST,
               "pv_linear_f32_group: item %d operands must be 16-byte aligned with unit stride along M / N", i);
    G.g[i] = LinArgs{t.A, t.B, nullptr, t.C, t.M, t.N, t.K, t.sam, t.sak, t.sbk, t.sbn, t.ldc, 0, 0, 0, 1.f, t.alpha_dev,
                     t.accumulate, MODE_XFAST, MODE_XFAST, nullptr, nullptr, t.r1_c, t.r1_u, t.r1_v, t.r1_nc};
    G.ntn[i] = cdiv(t.N, 32);
    tiles += cdiv(t.M, 32) * G.ntn[i];
    G.tile_end[i] = tiles;
  }
  for (int i = n; i < GROUP_MAX; ++i) { G.g[i] = G.g[0]; G.tile_end[i] = tiles; G.ntn[i] = 1; }
  G.n = n;
  int riders = 0;
  for (int j = 0; j < njobs; ++j) {
    G.jobs[j] = jobs[j];
    riders += cdiv(jobs[j].width, 256);
    G.job_end[j] = riders;
  }
  G.njobs = njobs;
  hipLaunchKernelGGL(linear_f32_group_kernel, dim3(tiles + riders), dim3(256), 0, st, G);
  PV_LAUNCH_CHECK("linear_f32_group_kernel");
  return PV_OK;
}

extern "C" {

int pv_linear_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias,
                  float* C, long ldc, int M, int N, int K, float alpha, const float* alpha_dev, int accumulate,
                  int batch, long bsa, long bsb, long bsc, void* stream) {
  PV_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && ldc >= N, "pv_linear_f32: bad argument");
  LinArgs g{A, B, bias, C, M, N, K, sam, sak, sbk, sbn, ldc, bsa, bsb, bsc, alpha, alpha_dev, accumulate,
            pick_mode(A, sam, sak, bsa, batch), pick_mode(B, sbn, sbk, bsb, batch), nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  launch_linear(g, batch, (hipStream_t)stream);
  PV_LAUNCH_CHECK("linear_f32_kernel");
  return PV_OK;
}

// The same GEMM with the Block fusions (dialog_models.py:258-269): gelu_out also receives gelu(C) (the activation the
// next linear consumes, computed once per element here instead of once per consuming tile); gelu_grad multiplies the
// result by gelu'(gelu_grad[m][n]) (backward through that GELU); r1_* subtract the spectral-norm rank-1 term from a
// weight gradient (see pv_spectral_norm_reduce).
int pv_linear_f32_fused(const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias,
                        float* C, long ldc, int M, int N, int K, float alpha, const float* alpha_dev, int accumulate,
                        float* gelu_out, const float* gelu_grad, const float* r1_c, int r1_nc, const float* r1_u,
                        const float* r1_v, void* stream) {
  PV_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N, "pv_linear_f32_fused: bad argument");
  PV_REQUIRE(!r1_c || (r1_u && r1_v && r1_nc > 0), "pv_linear_f32_fused: rank-1 term needs c, u and v");
  LinArgs g{A, B, bias, C, M, N, K, sam, sak, sbk, sbn, ldc, 0, 0, 0, alpha, alpha_dev, accumulate,
            pick_mode(A, sam, sak, 0, 1), pick_mode(B, sbn, sbk, 0, 1), gelu_out, gelu_grad, r1_c, r1_u, r1_v, r1_nc};
  launch_linear(g, 1, (hipStream_t)stream);
  PV_LAUNCH_CHECK("linear_f32_kernel");
  return PV_OK;
}

}  // extern "C"
