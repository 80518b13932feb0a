// Fused additive ('cat') attention scores, row softmax and tanh for the decoder attention block
// (models/enc2dec/decoders.py:89-118).  The reference tiles the mapped decoder outputs against the mapped
// context into a [B, Lout, Lin, D] fp32 tensor (decoders.py:92-94: 1.31 GB at B=256, L=50, D=512) before the
// tanh; here score[b,i,j] = qw . tanh(k[b,j,:] + q[b,i,:]) + qb is reduced in registers and the 4-D tensor
// never exists, forward or backward (the backward recomputes tanh).
#include "common.h"
#include "../../include/pevae_hip.h"

namespace {

// block = (b, i): 4 waves stride over j; lanes stride over d
__global__ __launch_bounds__(256) void additive_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ qw, const float* __restrict__ qb,
                                                            int Lo, int Li, int D, float* __restrict__ score) {
  extern __shared__ float sh[];          // q row [D], qw [D]
  const int b = blockIdx.x / Lo, i = blockIdx.x % Lo;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* qrow = q + ((size_t)b * Lo + i) * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    sh[d] = qrow[d];
    sh[D + d] = qw[d];
  }
  __syncthreads();
  const float bias = qb ? qb[0] : 0.f;
  for (int j = w; j < Li; j += 4) {
    const float* krow = k + ((size_t)b * Li + j) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += sh[D + d] * tanhf(krow[d] + sh[d]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) score[((size_t)b * Lo + i) * Li + j] = s + bias;
  }
}

// which = 0: block (b,i) -> dq[b,i,:] and dqw partial;  which = 1: block (b,j) -> dk[b,j,:]
__global__ __launch_bounds__(256) void additive_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ qw, const float* __restrict__ ds,
                                                            int Lo, int Li, int D, int which, float* __restrict__ dout,
                                                            float* __restrict__ dqw_part) {
  const int n_outer = which == 0 ? Lo : Li, n_inner = which == 0 ? Li : Lo;
  const int b = blockIdx.x / n_outer, o = blockIdx.x % n_outer;
  const float* own = (which == 0 ? q : k) + ((size_t)b * n_outer + o) * D;
  const float* other = (which == 0 ? k : q) + (size_t)b * n_inner * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float x = own[d], wgt = qw[d];
    float acc = 0.f, accw = 0.f;
    for (int r = 0; r < n_inner; ++r) {
      const float g = which == 0 ? ds[((size_t)b * Lo + o) * Li + r] : ds[((size_t)b * Lo + r) * Li + o];
      const float t = tanhf(other[(size_t)r * D + d] + x);
      acc += g * (1.f - t * t);
      accw += g * t;
    }
    dout[((size_t)b * n_outer + o) * D + d] = acc * wgt;
    if (which == 0 && dqw_part) dqw_part[(size_t)blockIdx.x * D + d] = accw;
  }
}

// one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int rows, int n,
                                                            float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * n;
  float m = -INFINITY;
  for (int c = lane; c < n; c += 64) m = fmaxf(m, xr[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += expf(xr[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  for (int c = lane; c < n; c += 64) y[(size_t)r * n + c] = expf(xr[c] - m) / s;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                int rows, int n, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < n; c += 64) dot += y[(size_t)r * n + c] * dy[(size_t)r * n + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  for (int c = lane; c < n; c += 64) dx[(size_t)r * n + c] = y[(size_t)r * n + c] * (dy[(size_t)r * n + c] - dot);
}

// op 0: y = tanh(x);  op 1: dx = dy * (1 - y^2);  op 2: y = gelu(x) (erf form);  op 3: dx = dy * gelu'(x)
__global__ void unary_kernel(const float* __restrict__ a, const float* __restrict__ bsrc, long n, int op,
                             float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = a[i];
    float r;
    if (op == 0) r = tanhf(x);
    else if (op == 1) r = bsrc[i] * (1.f - x * x);
    else if (op == 2) r = 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    else r = bsrc[i] * (0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x));
    out[i] = r;
  }
}

}  // namespace

extern "C" {

int pv_additive_scores_fwd(const float* q, const float* k, const float* qw, const float* qb, int B, int Lo, int Li,
                           int D, float* score, void* stream) {
  PV_REQUIRE(q && k && qw && score && B > 0 && Lo > 0 && Li > 0 && D > 0, "pv_additive_scores_fwd: bad argument");
  PV_REQUIRE(D <= 8192, "pv_additive_scores_fwd: D=%d > 8192", D);
  hipLaunchKernelGGL(additive_fwd_kernel, dim3(B * Lo), dim3(256), 2 * D * sizeof(float), (hipStream_t)stream, q, k, qw,
                     qb, Lo, Li, D, score);
  PV_LAUNCH_CHECK("additive_fwd_kernel");
  return PV_OK;
}

int pv_additive_scores_bwd(const float* q, const float* k, const float* qw, const float* dscore, int B, int Lo,
                           int Li, int D, float* dq, float* dk, float* dqw_part, float* dqw, void* stream) {
  PV_REQUIRE(q && k && qw && dscore && dq && dk && dqw_part && dqw, "pv_additive_scores_bwd: null argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(additive_bwd_kernel, dim3(B * Lo), dim3(256), 0, st, q, k, qw, dscore, Lo, Li, D, 0, dq, dqw_part);
  PV_LAUNCH_CHECK("additive_bwd_kernel(dq)");
  hipLaunchKernelGGL(additive_bwd_kernel, dim3(B * Li), dim3(256), 0, st, q, k, qw, dscore, Lo, Li, D, 1, dk,
                     (float*)nullptr);
  PV_LAUNCH_CHECK("additive_bwd_kernel(dk)");
  return pv_colsum_f32(dqw_part, B * Lo, D, D, dqw, 0, stream);
}

int pv_softmax_rows(const float* x, int rows, int n, float* y, void* stream) {
  PV_REQUIRE(x && y && rows > 0 && n > 0, "pv_softmax_rows: bad argument");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, n, y);
  PV_LAUNCH_CHECK("softmax_rows_kernel");
  return PV_OK;
}

int pv_softmax_rows_bwd(const float* y, const float* dy, int rows, int n, float* dx, void* stream) {
  PV_REQUIRE(y && dy && dx && rows > 0 && n > 0, "pv_softmax_rows_bwd: bad argument");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, dy, rows, n, dx);
  PV_LAUNCH_CHECK("softmax_rows_bwd_kernel");
  return PV_OK;
}

int pv_unary_f32(const float* a, const float* b, long n, int op, float* out, void* stream) {
  PV_REQUIRE(a && out && n > 0 && op >= 0 && op <= 3 && (op % 2 == 0 || b), "pv_unary_f32: bad argument");
  long g = (n + 255) / 256;
  hipLaunchKernelGGL(unary_kernel, dim3((int)(g > 2048 ? 2048 : g)), dim3(256), 0, (hipStream_t)stream, a, b, n, op, out);
  PV_LAUNCH_CHECK("unary_kernel");
  return PV_OK;
}

}  // extern "C"
