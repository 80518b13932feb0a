// Small exact-fp32 strided GEMM for the latent / connector linears of the DLS_VAE step (M = batch = 256,
// K,N <= a few hundred: Hidden2Gaussian nn_lib.py:198-199, LinearConnector nn_lib.py:143, the Block MLPs
// dialog_models.py:258-263).  These are launch-latency bound, not FLOP bound, and they feed KL terms that are
// compared with the reference at fp32 tolerance, so they stay in fp32 FMA (no bf16 rounding).
//
//   C[m*ldc + n] (+)= alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]  (+ bias[n])
//
// Arbitrary element strides make forward (x W^T), data-grad (dy W) and weight-grad (dy^T x) the same kernel;
// `batch` + batch strides give the bmm forms of the attention block (decoders.py:85-107).
#include "common.h"
#include "../../include/pevae_hip.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

struct LinArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  int M, N, K;
  long sam, sak, sbk, sbn, ldc;
  long bsa, bsb, bsc;   // batch strides (0 = shared)
  float alpha;
  int accumulate;
};

__global__ __launch_bounds__(256) void linear_f32_kernel(LinArgs g) {
  __shared__ float As[TK][TM + 1];
  __shared__ float Bs[TK][TN + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const float* A = g.A + (long)blockIdx.z * g.bsa;
  const float* B = g.B + (long)blockIdx.z * g.bsb;
  float* C = g.C + (long)blockIdx.z * g.bsc;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // loader mapping: prefer the unit-stride direction as the fast thread index
  const bool a_kfast = g.sak == 1, b_nfast = g.sbn == 1;
  for (int k0 = 0; k0 < g.K; k0 += TK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      int mm, kk;
      if (a_kfast) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
      const int m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < g.M && k < g.K) ? A[m * g.sam + k * g.sak] : 0.f;
      int nn, kb;
      if (b_nfast) { nn = e & 63; kb = e >> 6; } else { kb = e & 15; nn = e >> 4; }
      const int n = n0 + nn, k2 = k0 + kb;
      Bs[kb][nn] = (n < g.N && k2 < g.K) ? B[k2 * g.sbk + n * g.sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx + 16 * j;
      if (n >= g.N) continue;
      float v = g.alpha * acc[i][j];
      if (g.bias) v += g.bias[n];
      float* dst = C + m * g.ldc + n;
      if (g.accumulate) v += *dst;
      *dst = v;
    }
  }
}

}  // namespace

extern "C" {

int pv_linear_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias,
                  float* C, long ldc, int M, int N, int K, float alpha, int accumulate, int batch, long bsa,
                  long bsb, long bsc, void* stream) {
  PV_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && ldc >= N, "pv_linear_f32: bad argument");
  LinArgs g{A, B, bias, C, M, N, K, sam, sak, sbk, sbn, ldc, bsa, bsb, bsc, alpha, accumulate};
  hipLaunchKernelGGL(linear_f32_kernel, dim3(cdiv(N, TN), cdiv(M, TM), batch), dim3(256), 0, (hipStream_t)stream, g);
  PV_LAUNCH_CHECK("linear_f32_kernel");
  return PV_OK;
}

}  // extern "C"
