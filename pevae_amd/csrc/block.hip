// Kernels for the DLS latent pyramid (models/dialog_models.py:253-271 Block, :1835-1870 DlsRec/DlsPri,
// :3161-3352 DLS_Auto): LayerNorm over a residual sum, spectral-norm power iteration and its weight-gradient
// correction, row L2 normalisation and the symmetric InfoNCE loss of DLS_Auto.cons_loss (:3266-3272).
// All of these are tiny (batch x <=1536) and latency bound: one wave per row, fixed-order reductions.
#include "common.h"
#include "linear_group.h"
#include "../../include/pevae_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 256 threads of a block (result valid in every thread); `red` = 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}

// out[r] = LayerNorm(x[r] + y[r]) * gamma + beta   (nn.LayerNorm, biased variance; dialog_models.py:269)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int rows, int D, float eps, float* __restrict__ out,
                                                             float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * D;
  const float* yr = y ? y + (size_t)r * D : nullptr;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c] + (yr ? yr[c] : 0.f);
  const float mu = wave_sum(s) / D;
  float q = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = xr[c] + (yr ? yr[c] : 0.f) - mu;
    q += d * d;
  }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  for (int c = lane; c < D; c += 64)
    out[(size_t)r * D + c] = (xr[c] + (yr ? yr[c] : 0.f) - mu) * rs * gamma[c] + beta[c];
  if (lane == 0) {
    mean[r] = mu;
    rstd[r] = rs;
  }
}

// ds[r] = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dout * gamma   (gradient w.r.t. the summed input)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ dout,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             int rows, int D, float* __restrict__ ds) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * D;
  const float* yr = y ? y + (size_t)r * D : nullptr;
  const float mu = mean[r], rs = rstd[r];
  float a = 0.f, b = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float xh = (xr[c] + (yr ? yr[c] : 0.f) - mu) * rs, g = dout[(size_t)r * D + c] * gamma[c];
    a += g;
    b += g * xh;
  }
  a = wave_sum(a) / D;
  b = wave_sum(b) / D;
  for (int c = lane; c < D; c += 64) {
    const float xh = (xr[c] + (yr ? yr[c] : 0.f) - mu) * rs, g = dout[(size_t)r * D + c] * gamma[c];
    ds[(size_t)r * D + c] = rs * (g - a - xh * b);
  }
}

// dgamma[c] (+)= sum_r dout * xhat, dbeta[c] (+)= sum_r dout.  Block = 64 columns x 4 row phases, the phases are
// added in a fixed order (reproducible).
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                    const float* __restrict__ dout,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, int rows, int D,
                                                                    int accumulate, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta) {
  __shared__ float pg[4][64], pb[4][64];
  const int cl = threadIdx.x & 63, yy = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float g = 0.f, b = 0.f;
  if (c < D) {
    int r = yy;
    for (; r + 12 < rows; r += 16) {
      float sv[4], dv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t o = (size_t)(r + 4 * k) * D + c;
        sv[k] = x[o] + (y ? y[o] : 0.f);
        dv[k] = dout[o];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g += dv[k] * (sv[k] - mean[r + 4 * k]) * rstd[r + 4 * k];
        b += dv[k];
      }
    }
    for (; r < rows; r += 4) {
      const size_t o = (size_t)r * D + c;
      const float d = dout[o];
      g += d * (x[o] + (y ? y[o] : 0.f) - mean[r]) * rstd[r];
      b += d;
    }
  }
  pg[yy][cl] = g;
  pb[yy][cl] = b;
  __syncthreads();
  if (yy == 0 && c < D) {
    const float tg = ((pg[0][cl] + pg[1][cl]) + pg[2][cl]) + pg[3][cl];
    const float tb = ((pb[0][cl] + pb[1][cl]) + pb[2][cl]) + pb[3][cl];
    dgamma[c] = accumulate ? dgamma[c] + tg : tg;
    dbeta[c] = accumulate ? dbeta[c] + tb : tb;
  }
}

// torch.nn.utils.spectral_norm (hook form, one power iteration per training forward, dialog_models.py:258-263):
//   training: v = normalize(W^T u), u = normalize(W v) (buffers updated in place);  sigma = u . (W v)
// One block per matrix; writes 1/sigma.  W [out][in] row-major.
struct SnDesc {             // one spectral-norm weight (device array for the multi-tensor launch)
  const float* W;
  float* u;
  float* v;
  float* inv_sigma;
  long n_out, n_in;
  float* Wt;                // optional [n_in][n_out]: W^T, rewritten every call (operand layout of pv_block_fwd)
};

__device__ __forceinline__ void spectral_iter_body(const float* __restrict__ W, float* __restrict__ u,
                                                   float* __restrict__ v, int n_out, int n_in, float eps,
                                                   int do_iter, float* __restrict__ inv_sigma) {
  extern __shared__ float sh[];           // u [n_out], v [n_in], wv [n_out], red [4]
  float* su = sh;
  float* sv = sh + n_out;
  float* swv = sv + n_in;
  float* red = swv + n_out;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < n_out; i += 256) su[i] = u[i];
  for (int j = tid; j < n_in; j += 256) sv[j] = v[j];
  __syncthreads();
  if (do_iter) {
    float nn = 0.f;
    for (int j = tid; j < n_in; j += 256) {        // v = W^T u   (coalesced over j, 8 rows in flight)
      float s = 0.f;
      int i = 0;
      for (; i + 8 <= n_out; i += 8) {
        float wv8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wv8[k] = W[(size_t)(i + k) * n_in + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += wv8[k] * su[i + k];
      }
      for (; i < n_out; ++i) s += W[(size_t)i * n_in + j] * su[i];
      sv[j] = s;
      nn += s * s;
    }
    nn = block_sum(nn, red);
    const float inv = 1.f / fmaxf(sqrtf(nn), eps);
    for (int j = tid; j < n_in; j += 256) sv[j] *= inv;
    __syncthreads();
  }
  for (int i0 = w * 4; i0 < n_out; i0 += 16) {     // wv = W v   (one wave per 4 rows: 4 independent row sums)
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < n_in; j += 64) {
      const float vj = sv[j];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i0 + k < n_out) s4[k] += W[(size_t)(i0 + k) * n_in + j] * vj;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float t = wave_sum(s4[k]);
      if (lane == 0 && i0 + k < n_out) swv[i0 + k] = t;
    }
  }
  __syncthreads();
  if (do_iter) {
    float nn = 0.f;
    for (int i = tid; i < n_out; i += 256) nn += swv[i] * swv[i];
    nn = block_sum(nn, red);
    const float inv = 1.f / fmaxf(sqrtf(nn), eps);
    for (int i = tid; i < n_out; i += 256) su[i] = swv[i] * inv;
    __syncthreads();
    for (int i = tid; i < n_out; i += 256) u[i] = su[i];
    for (int j = tid; j < n_in; j += 256) v[j] = sv[j];
  }
  float d = 0.f;
  for (int i = tid; i < n_out; i += 256) d += su[i] * swv[i];
  d = block_sum(d, red);
  if (tid == 0) inv_sigma[0] = 1.f / d;
}

__global__ __launch_bounds__(256) void spectral_iter_kernel(const float* __restrict__ W, float* __restrict__ u,
                                                             float* __restrict__ v, int n_out, int n_in, float eps,
                                                             int do_iter, float* __restrict__ inv_sigma) {
  spectral_iter_body(W, u, v, n_out, n_in, eps, do_iter, inv_sigma);
}
// one workgroup per weight: every spectral-norm matrix of the model advances in ONE launch
__global__ __launch_bounds__(256) void spectral_iter_multi_kernel(const SnDesc* __restrict__ desc, float eps, int do_iter) {
  const SnDesc d = desc[blockIdx.x];
  spectral_iter_body(d.W, d.u, d.v, (int)d.n_out, (int)d.n_in, eps, do_iter, d.inv_sigma);
}

// ---- the multi-tensor pre-pass, spread over the chip ------------------------------------------------------------
// One workgroup per weight reads a 512x512 matrix twice on a single CU (~220 us on the critical path of every step).
// The power iteration is two mat-vecs with a norm in between, so it is split at the norms into three launches whose
// workgroups each own a 64-column / 64-row slab of one weight:
//   phase 1 (training): v <- W^T u (unnormalised, in place), pn1[weight][slab] = partial |v|^2
//   phase 2           : wv = W (v / |v|) -> workspace, pn2[weight][slab] = partial |wv|^2
//   phase 3           : training: u <- wv / |wv|, v <- v / |v|;   1/sigma = 1 / (u . wv)
// workspace per weight: SN_WS_HDR floats of partials + max_out floats of wv.  Partials are summed in slab order.
constexpr int SN_SLABS = 16;               // up to 1024 rows / columns
constexpr int SN_WS_HDR = 2 * SN_SLABS;

__device__ __forceinline__ float sn_inv_norm(const float* part, int nslab, float eps) {
  float nn = 0.f;
  for (int i = 0; i < nslab; ++i) nn += part[i];
  return 1.f / fmaxf(sqrtf(nn), eps);
}

__global__ __launch_bounds__(256) void sn_phase1_kernel(const SnDesc* __restrict__ desc, float* __restrict__ ws, int ws_stride) {
  __shared__ float part[4][64];
  __shared__ float red[4];
  const SnDesc d = desc[blockIdx.y];
  const int n_out = (int)d.n_out, n_in = (int)d.n_in, j0 = blockIdx.x * 64;
  if (j0 >= n_in) return;
  const int c = threadIdx.x & 63, rq = threadIdx.x >> 6, j = min(j0 + c, n_in - 1);
  float sacc = 0.f;
  int i = rq;
  for (; i + 28 < n_out; i += 32) {                  // 8 independent loads in flight
    float wv8[8], u8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      wv8[k] = d.W[(size_t)(i + 4 * k) * n_in + j];
      u8[k] = d.u[i + 4 * k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sacc += wv8[k] * u8[k];
  }
  for (; i < n_out; i += 4) sacc += d.W[(size_t)i * n_in + j] * d.u[i];
  part[rq][c] = sacc;
  __syncthreads();
  float t = 0.f;
  if (rq == 0) {
    t = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
    if (j0 + c < n_in) d.v[j0 + c] = t; else t = 0.f;
  }
  const float nn = block_sum(t * t, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.y * ws_stride + blockIdx.x] = nn;
}

__global__ __launch_bounds__(256) void sn_phase2_kernel(const SnDesc* __restrict__ desc, float* __restrict__ ws, int ws_stride,
                                                         float eps, int do_iter) {
  extern __shared__ float sv[];                      // v / |v|  [n_in]
  __shared__ float red[4];
  const SnDesc d = desc[blockIdx.y];
  const int n_out = (int)d.n_out, n_in = (int)d.n_in, i0 = blockIdx.x * 64;
  if (i0 >= n_out) return;
  float* wsw = ws + (size_t)blockIdx.y * ws_stride;
  const float inv = do_iter ? sn_inv_norm(wsw, (n_in + 63) / 64, eps) : 1.f;
  for (int j = threadIdx.x; j < n_in; j += 256) sv[j] = d.v[j] * inv;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float nn = 0.f;
  for (int r = 0; r < 16; r += 4) {                  // wave w: rows i0 + 16 w + r .. + 3, four row sums side by side
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < n_in; j += 64) {
      const float vj = sv[j];
#pragma unroll
      for (int k = 0; k < 4; ++k) s4[k] += d.W[(size_t)min(i0 + 16 * w + r + k, n_out - 1) * n_in + j] * vj;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float t = wave_sum(s4[k]);
      const int i = i0 + 16 * w + r + k;
      if (i < n_out) {
        if (lane == 0) wsw[SN_WS_HDR + i] = t;
        nn += t * t;                                  // identical in every lane
      }
    }
  }
  __syncthreads();
  if (lane == 0) red[w] = nn;
  __syncthreads();
  if (threadIdx.x == 0) wsw[SN_SLABS + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void sn_phase3_kernel(const SnDesc* __restrict__ desc, const float* __restrict__ ws,
                                                         int ws_stride, float eps, int do_iter) {
  __shared__ float red[4];
  const SnDesc d = desc[blockIdx.x];
  const int n_out = (int)d.n_out, n_in = (int)d.n_in;
  const float* wsw = ws + (size_t)blockIdx.x * ws_stride;
  float dot = 0.f;
  if (do_iter) {
    const float inv_v = sn_inv_norm(wsw, (n_in + 63) / 64, eps), inv_u = sn_inv_norm(wsw + SN_SLABS, (n_out + 63) / 64, eps);
    for (int j = threadIdx.x; j < n_in; j += 256) d.v[j] *= inv_v;
    for (int i = threadIdx.x; i < n_out; i += 256) {
      const float wv = wsw[SN_WS_HDR + i], un = wv * inv_u;
      d.u[i] = un;
      dot += un * wv;
    }
  } else {
    for (int i = threadIdx.x; i < n_out; i += 256) dot += d.u[i] * wsw[SN_WS_HDR + i];
  }
  dot = block_sum(dot, red);
  if (threadIdx.x == 0) d.inv_sigma[0] = 1.f / dot;
}

// W^T of every described weight (32x32 tiles through LDS; blockIdx.y = weight, blockIdx.x walks its tiles)
__global__ __launch_bounds__(256) void sn_transpose_multi_kernel(const SnDesc* __restrict__ desc) {
  __shared__ float tile[32][33];
  const SnDesc d = desc[blockIdx.y];
  if (!d.Wt) return;
  const int n_out = (int)d.n_out, n_in = (int)d.n_in;
  const int tx = (n_in + 31) / 32, ty = (n_out + 31) / 32;
  const int cx = threadIdx.x & 31, cy = threadIdx.x >> 5;           // 32 x 8
  for (int t = blockIdx.x; t < tx * ty; t += gridDim.x) {
    const int i0 = (t / tx) * 32, j0 = (t % tx) * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + cy + 8 * r, j = j0 + cx;
      if (i < n_out && j < n_in) tile[cy + 8 * r][cx] = d.W[(size_t)i * n_in + j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + cy + 8 * r, i = i0 + cx;
      if (i < n_out && j < n_in) d.Wt[(size_t)j * n_out + i] = tile[cx][cy + 8 * r];
    }
    __syncthreads();
  }
}

// One pass over dy for a spectral-norm linear's backward: db[n] (+)= sum_m dy[m][n] and, per 64-column block,
// cpart[block] = sum_{m, n in block} dy[m][n] * (y[m][n] - bias[n]); the sum of cpart is <dL/dW_eff, W_eff>.
__global__ __launch_bounds__(256) void sn_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ bias, int M, int N,
                                                         float* __restrict__ db, int accumulate_db,
                                                         float* __restrict__ cpart) {
  __shared__ float part[4][64];
  __shared__ float red[4];
  const int cl = threadIdx.x & 63, yy = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float s = 0.f, q = 0.f;
  if (c < N) {
    const float b = bias ? bias[c] : 0.f;
    int r = yy;
    for (; r + 28 < M; r += 32) {                    // 16 independent loads in flight
      float d[8], v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        d[k] = dy[(size_t)(r + 4 * k) * N + c];
        v[k] = y[(size_t)(r + 4 * k) * N + c];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s += d[k];
        q += d[k] * (v[k] - b);
      }
    }
    for (; r < M; r += 4) {
      const float d = dy[(size_t)r * N + c];
      s += d;
      q += d * (y[(size_t)r * N + c] - b);
    }
  }
  part[yy][cl] = s;
  q = block_sum(q, red);
  if (threadIdx.x == 0) cpart[blockIdx.x] = q;
  if (yy == 0 && c < N && db) {
    const float t = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
    db[c] = accumulate_db ? db[c] + t : t;
  }
}

// c = sum_{m,n} dy[m][n] * (y[m][n] - bias[n])  =  <dL/dW_eff, W_eff>   (one block, fixed order)
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ bias, long M, int N, float* __restrict__ c) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = threadIdx.x; i < M * N; i += 256) s += dy[i] * (y[i] - (bias ? bias[i % N] : 0.f));
  s = block_sum(s, red);
  if (threadIdx.x == 0) c[0] = s;
}

// dW[i][j] -= c * inv_sigma * u[i] * v[j]
__global__ void sn_rank1_kernel(float* __restrict__ dW, const float* __restrict__ u, const float* __restrict__ v,
                                const float* __restrict__ c, const float* __restrict__ inv_sigma, int n_out, int n_in) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n_out * n_in) return;
  dW[i] -= c[0] * inv_sigma[0] * u[i / n_in] * v[i % n_in];
}

// a[r] = x[r] / ||x[r]||  (dialog_models.py:3269-3270; no epsilon, like the reference)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int rows, int D,
                                                          float* __restrict__ a, float* __restrict__ nrm) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float q = 0.f;
  for (int c = lane; c < D; c += 64) q += x[(size_t)r * D + c] * x[(size_t)r * D + c];
  const float n = sqrtf(wave_sum(q));
  for (int c = lane; c < D; c += 64) a[(size_t)r * D + c] = x[(size_t)r * D + c] / n;
  if (lane == 0) nrm[r] = n;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ a, const float* __restrict__ nrm,
                                                          const float* __restrict__ da, int rows, int D,
                                                          float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float d = 0.f;
  for (int c = lane; c < D; c += 64) d += a[(size_t)r * D + c] * da[(size_t)r * D + c];
  d = wave_sum(d);
  for (int c = lane; c < D; c += 64) dx[(size_t)r * D + c] = (da[(size_t)r * D + c] - a[(size_t)r * D + c] * d) / nrm[r];
}

// which = 0: lse over row i of S;  which = 1: lse over column i of S.   S [B][B]
__global__ __launch_bounds__(256) void infonce_lse_kernel(const float* __restrict__ S, int B, float* __restrict__ lse_r,
                                                           float* __restrict__ lse_c) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6), which = blockIdx.y;
  if (i >= B) return;
  const long st = which ? B : 1;
  const float* p = S + (which ? i : (long)i * B);
  float m = -INFINITY;
  for (int j = lane; j < B; j += 64) m = fmaxf(m, p[j * st]);
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < B; j += 64) s += expf(p[j * st] - m);
  s = wave_sum(s);
  if (lane == 0) (which ? lse_c : lse_r)[i] = m + logf(s);
}
// loss = ( mean_i(lse_r[i] - S_ii) + mean_i(lse_c[i] - S_ii) ) / 2      (one block)
__global__ __launch_bounds__(256) void infonce_loss_kernel(const float* __restrict__ S, const float* __restrict__ lse_r,
                                                            const float* __restrict__ lse_c, int B, float* __restrict__ loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += lse_r[i] + lse_c[i] - 2.f * S[(long)i * B + i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = s / (2.f * B);
}
// dS[i][j] = g/(2B) * (exp(S - lse_r[i]) + exp(S - lse_c[j]) - 2 [i == j])
__global__ void infonce_bwd_kernel(const float* __restrict__ S, const float* __restrict__ lse_r,
                                   const float* __restrict__ lse_c, const float* __restrict__ g, int B,
                                   float* __restrict__ dS) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (long)B * B) return;
  const int i = (int)(k / B), j = (int)(k % B);
  const float s = S[k];
  dS[k] = g[0] / (2.f * B) * (expf(s - lse_r[i]) + expf(s - lse_c[j]) - (i == j ? 2.f : 0.f));
}

// ---- DLS_Auto.cons_loss for several (v1, v2) pairs at once (dialog_models.py:3266-3272, called once per level :3307
// and once on the outputs :3284): every pair of a step depends only on the pyramids' activations, so the 5 pairs of the
// Yelp model are one launch per stage (normalise, scores, log-sum-exps, loss | dS, dn, dv) instead of 5 x 10 launches.
constexpr int CONS_MAX = 7;                    // 2 * CONS_MAX weight... items must fit the grouped linear launch (15)
struct ConsGroup {
  const float* x[2 * CONS_MAX];                // v1_p at 2p, v2_p at 2p + 1
  float* a[2 * CONS_MAX];                      // normalised rows [B][D]
  float* aT[2 * CONS_MAX];                     // ... and their transposes [D][B] (the grouped GEMM wants K-major operands)
  float* nrm[2 * CONS_MAX];
  float* S[CONS_MAX];                          // [B][B] scores
  float* lse_r[CONS_MAX];
  float* lse_c[CONS_MAX];
  float* dS[CONS_MAX];
  float* dST[CONS_MAX];                        // dS^T
  float* da[2 * CONS_MAX];                     // gradient of the normalised rows
  float* dx[2 * CONS_MAX];                     // outputs of the backward pass
  int D[CONS_MAX];
  int Dp[CONS_MAX];                            // leading dimension of a / da: round_up(D, 4)
  int n, B, Bp;                                // Bp = round_up(B, 4): leading dimension of aT, S, dS, dST (any batch size
};                                             // and width keeps every grouped-GEMM operand row 16-byte aligned)

__global__ __launch_bounds__(256) void cons_l2norm_fwd_kernel(ConsGroup g, float alpha, float* __restrict__ alpha_dev) {
  const int m = blockIdx.y, D = g.D[m >> 1], Dp = g.Dp[m >> 1];
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && m == 0 && threadIdx.x == 0) alpha_dev[0] = alpha;
  if (r >= g.B) return;
  const float* x = g.x[m] + (size_t)r * D;
  float q = 0.f;
  for (int c = lane; c < D; c += 64) q += x[c] * x[c];
  const float n = sqrtf(wave_sum(q));
  for (int c = lane; c < D; c += 64) {
    const float v = x[c] / n;
    g.a[m][(size_t)r * Dp + c] = v;
    g.aT[m][(size_t)c * g.Bp + r] = v;
  }
  if (lane == 0) g.nrm[m][r] = n;
}
__global__ __launch_bounds__(256) void cons_l2norm_bwd_kernel(ConsGroup g) {
  const int m = blockIdx.y, D = g.D[m >> 1], Dp = g.Dp[m >> 1];
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= g.B) return;
  const float* a = g.a[m] + (size_t)r * Dp;
  const float* da = g.da[m] + (size_t)r * Dp;
  float d = 0.f;
  for (int c = lane; c < D; c += 64) d += a[c] * da[c];
  d = wave_sum(d);
  const float inv = 1.f / g.nrm[m][r];
  for (int c = lane; c < D; c += 64) g.dx[m][(size_t)r * D + c] = (da[c] - a[c] * d) * inv;
}
// blockIdx.y = 2p + which (0: rows, 1: columns)
__global__ __launch_bounds__(256) void cons_lse_kernel(ConsGroup g) {
  const int p = blockIdx.y >> 1, which = blockIdx.y & 1, B = g.B;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const long st = which ? g.Bp : 1;
  const float* q = g.S[p] + (which ? i : (long)i * g.Bp);
  float m = -INFINITY;
  for (int j = lane; j < B; j += 64) m = fmaxf(m, q[j * st]);
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < B; j += 64) s += expf(q[j * st] - m);
  s = wave_sum(s);
  if (lane == 0) (which ? g.lse_c[p] : g.lse_r[p])[i] = m + logf(s);
}
// loss[0] = sum_p ( mean_i(lse_r - S_ii) + mean_i(lse_c - S_ii) ) / 2, pairs added in index order (one block)
__global__ __launch_bounds__(256) void cons_loss_kernel(ConsGroup g, float* __restrict__ loss) {
  __shared__ float red[4];
  float total = 0.f;
  for (int p = 0; p < g.n; ++p) {
    float s = 0.f;
    for (int i = threadIdx.x; i < g.B; i += 256) s += g.lse_r[p][i] + g.lse_c[p][i] - 2.f * g.S[p][(long)i * g.Bp + i];
    s = block_sum(s, red);
    total += s / (2.f * g.B);
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = total;
}
__global__ void cons_bwd_kernel(ConsGroup g, const float* __restrict__ gout) {
  const int p = blockIdx.y, B = g.B;
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (long)B * B) return;
  const int i = (int)(k / B), j = (int)(k % B);
  const float s = g.S[p][(long)i * g.Bp + j];
  const float d = gout[0] / (2.f * B) * (expf(s - g.lse_r[p][i]) + expf(s - g.lse_c[p][j]) - (i == j ? 2.f : 0.f));
  g.dS[p][(long)i * g.Bp + j] = d;
  g.dST[p][(long)j * g.Bp + i] = d;
}

// workspace carve-up shared by the two entry points (floats): alpha | per pair: a1 a2 nrm1 nrm2 S lse_r lse_c dS da1 da2
size_t cons_fill(ConsGroup& g, const int* D, int n, int B, float* ws) {
  size_t o = 4;
  g.n = n;
  g.B = B;
  g.Bp = (B + 3) & ~3;
  for (int p = 0; p < n; ++p) {
    g.D[p] = D[p];
    g.Dp[p] = (D[p] + 3) & ~3;
    const size_t bd = (size_t)g.Bp * g.Dp[p], bb = (size_t)B * g.Bp;
    auto take = [&](size_t k) { float* q = ws ? ws + o : nullptr; o += (k + 3) & ~(size_t)3; return q; };
    g.a[2 * p] = take(bd); g.a[2 * p + 1] = take(bd);
    g.aT[2 * p] = take(bd); g.aT[2 * p + 1] = take(bd);
    g.nrm[2 * p] = take(B); g.nrm[2 * p + 1] = take(B);
    g.S[p] = take(bb); g.lse_r[p] = take(B); g.lse_c[p] = take(B);
    g.dS[p] = take(bb); g.dST[p] = take(bb);
    g.da[2 * p] = take(bd); g.da[2 * p + 1] = take(bd);
  }
  return o;
}

}  // namespace

extern "C" {

size_t pv_cons_loss_ws_floats(const int* D, int n, int B) {
  ConsGroup g;
  if (!D || n < 1 || n > CONS_MAX || B < 1) return 0;
  return cons_fill(g, D, n, B, nullptr);
}

int pv_cons_loss_fwd(const float* const* v1, const float* const* v2, const int* D, int n, int B, float alpha, float* ws,
                     float* loss, void* stream) {
  PV_REQUIRE(v1 && v2 && D && ws && loss && n >= 1 && n <= CONS_MAX && B > 0, "pv_cons_loss_fwd: bad argument (1..%d pairs)", CONS_MAX);
  hipStream_t st = (hipStream_t)stream;
  ConsGroup g{};
  cons_fill(g, D, n, B, ws);
  LinGroupItem it[CONS_MAX];
  for (int p = 0; p < n; ++p) {
    PV_REQUIRE(v1[p] && v2[p] && D[p] > 0, "pv_cons_loss_fwd: bad pair %d", p);
    g.x[2 * p] = v1[p];
    g.x[2 * p + 1] = v2[p];
    // S = alpha a1 a2^T from the K-major copies: A(m, k) = a1T[k][m], B(k, n) = a2T[k][n]
    it[p] = LinGroupItem{g.aT[2 * p], 1, g.Bp, g.aT[2 * p + 1], g.Bp, 1, g.S[p], g.Bp, B, B, D[p], ws, 0, nullptr, 0, nullptr, nullptr};
  }
  hipLaunchKernelGGL(cons_l2norm_fwd_kernel, dim3(cdiv(B, 4), 2 * n), dim3(256), 0, st, g, alpha, ws);
  PV_LAUNCH_CHECK("cons_l2norm_fwd_kernel");
  int rc = pv_linear_f32_group(it, n, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cons_lse_kernel, dim3(cdiv(B, 4), 2 * n), dim3(256), 0, st, g);
  PV_LAUNCH_CHECK("cons_lse_kernel");
  hipLaunchKernelGGL(cons_loss_kernel, dim3(1), dim3(256), 0, st, g, loss);
  PV_LAUNCH_CHECK("cons_loss_kernel");
  return PV_OK;
}

int pv_cons_loss_bwd(const float* gout, const int* D, int n, int B, float* ws, float* const* dv1, float* const* dv2,
                     void* stream) {
  PV_REQUIRE(gout && D && ws && dv1 && dv2 && n >= 1 && n <= CONS_MAX && B > 0, "pv_cons_loss_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  ConsGroup g{};
  cons_fill(g, D, n, B, ws);
  LinGroupItem it[2 * CONS_MAX];
  for (int p = 0; p < n; ++p) {
    PV_REQUIRE(dv1[p] && dv2[p], "pv_cons_loss_bwd: null output %d", p);
    g.dx[2 * p] = dv1[p];
    g.dx[2 * p + 1] = dv2[p];
    // da1 = alpha dS a2 : A(m, k) = dST[k][m] ;  da2 = alpha dS^T a1 : A(m, k) = dS[k][m] ;  B operands row-major [B][D]
    it[2 * p] = LinGroupItem{g.dST[p], 1, g.Bp, g.a[2 * p + 1], g.Dp[p], 1, g.da[2 * p], g.Dp[p], B, D[p], B, ws, 0, nullptr, 0, nullptr, nullptr};
    it[2 * p + 1] = LinGroupItem{g.dS[p], 1, g.Bp, g.a[2 * p], g.Dp[p], 1, g.da[2 * p + 1], g.Dp[p], B, D[p], B, ws, 0, nullptr, 0, nullptr, nullptr};
  }
  const long nn = (long)B * B;
  hipLaunchKernelGGL(cons_bwd_kernel, dim3((unsigned)((nn + 255) / 256), n), dim3(256), 0, st, g, gout);
  PV_LAUNCH_CHECK("cons_bwd_kernel");
  int rc = pv_linear_f32_group(it, 2 * n, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cons_l2norm_bwd_kernel, dim3(cdiv(B, 4), 2 * n), dim3(256), 0, st, g);
  PV_LAUNCH_CHECK("cons_l2norm_bwd_kernel");
  return PV_OK;
}

int pv_layernorm_fwd(const float* x, const float* y, const float* gamma, const float* beta, int rows, int D, float eps,
                     float* out, float* mean, float* rstd, void* stream) {
  PV_REQUIRE(x && gamma && beta && out && mean && rstd && rows > 0 && D > 0, "pv_layernorm_fwd: bad argument");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta,
                     rows, D, eps, out, mean, rstd);
  PV_LAUNCH_CHECK("layernorm_fwd_kernel");
  return PV_OK;
}

int pv_layernorm_bwd(const float* x, const float* y, const float* gamma, const float* dout, const float* mean,
                     const float* rstd, int rows, int D, float* ds, float* dgamma, float* dbeta, int accumulate_params,
                     void* stream) {
  PV_REQUIRE(x && gamma && dout && mean && rstd && ds && dgamma && dbeta && rows > 0 && D > 0, "pv_layernorm_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, y, gamma, dout, mean, rstd, rows, D, ds);
  PV_LAUNCH_CHECK("layernorm_bwd_kernel");
  hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3(cdiv(D, 64)), dim3(256), 0, st, x, y, dout, mean, rstd, rows, D,
                     accumulate_params, dgamma, dbeta);
  PV_LAUNCH_CHECK("layernorm_bwd_params_kernel");
  return PV_OK;
}

int pv_spectral_norm(const float* W, float* u, float* v, int n_out, int n_in, float eps, int do_iter, float* inv_sigma,
                     void* stream) {
  PV_REQUIRE(W && u && v && inv_sigma && n_out > 0 && n_in > 0, "pv_spectral_norm: bad argument");
  const size_t smem = (size_t)(2 * n_out + n_in + 4) * sizeof(float);
  PV_REQUIRE(smem <= 64 * 1024, "pv_spectral_norm: %d x %d exceeds the 64 KiB LDS staging of the power iteration", n_out, n_in);
  hipLaunchKernelGGL(spectral_iter_kernel, dim3(1), dim3(256), smem, (hipStream_t)stream, W, u, v, n_out, n_in, eps,
                     do_iter, inv_sigma);
  PV_LAUNCH_CHECK("spectral_iter_kernel");
  return PV_OK;
}

/* descs: device array of {W, u, v, inv_sigma (pointers), n_out, n_in (int64), Wt (pointer or 0)} records, 56 bytes each */
size_t pv_spectral_norm_multi_ws_bytes(int count, int max_out) {
  return (size_t)count * (SN_WS_HDR + (size_t)round_up(max_out > 0 ? max_out : 1, 4)) * sizeof(float);
}

int pv_spectral_norm_multi(const void* descs, int count, int max_out, int max_in, float eps, int do_iter, int transpose,
                           void* workspace, void* stream) {
  if (workspace) {
    PV_REQUIRE(descs && count > 0 && max_out > 0 && max_in > 0, "pv_spectral_norm_multi: bad argument");
    PV_REQUIRE(max_out <= 64 * SN_SLABS && max_in <= 64 * SN_SLABS, "pv_spectral_norm_multi: %d x %d exceeds %d rows / columns",
               max_out, max_in, 64 * SN_SLABS);
    hipStream_t st = (hipStream_t)stream;
    const SnDesc* dd = (const SnDesc*)descs;
    float* ws = (float*)workspace;
    const int stride = SN_WS_HDR + round_up(max_out, 4);
    if (do_iter) {
      hipLaunchKernelGGL(sn_phase1_kernel, dim3(cdiv(max_in, 64), count), dim3(256), 0, st, dd, ws, stride);
      PV_LAUNCH_CHECK("sn_phase1_kernel");
    }
    hipLaunchKernelGGL(sn_phase2_kernel, dim3(cdiv(max_out, 64), count), dim3(256), (size_t)max_in * sizeof(float), st, dd, ws,
                       stride, eps, do_iter);
    PV_LAUNCH_CHECK("sn_phase2_kernel");
    hipLaunchKernelGGL(sn_phase3_kernel, dim3(count), dim3(256), 0, st, dd, (const float*)ws, stride, eps, do_iter);
    PV_LAUNCH_CHECK("sn_phase3_kernel");
    if (transpose) {
      const int tiles = cdiv(max_out, 32) * cdiv(max_in, 32);
      hipLaunchKernelGGL(sn_transpose_multi_kernel, dim3(tiles < 64 ? tiles : 64, count), dim3(256), 0, st, dd);
      PV_LAUNCH_CHECK("sn_transpose_multi_kernel");
    }
    return PV_OK;
  }
  PV_REQUIRE(descs && count > 0 && max_out > 0 && max_in > 0, "pv_spectral_norm_multi: bad argument");
  const size_t smem = (size_t)(2 * max_out + max_in + 4) * sizeof(float);
  PV_REQUIRE(smem <= 64 * 1024, "pv_spectral_norm_multi: %d x %d exceeds the 64 KiB LDS staging", max_out, max_in);
  hipLaunchKernelGGL(spectral_iter_multi_kernel, dim3(count), dim3(256), smem, (hipStream_t)stream, (const SnDesc*)descs,
                     eps, do_iter);
  PV_LAUNCH_CHECK("spectral_iter_multi_kernel");
  if (transpose) {
    const int tiles = cdiv(max_out, 32) * cdiv(max_in, 32);
    hipLaunchKernelGGL(sn_transpose_multi_kernel, dim3(tiles < 64 ? tiles : 64, count), dim3(256), 0, (hipStream_t)stream,
                       (const SnDesc*)descs);
    PV_LAUNCH_CHECK("sn_transpose_multi_kernel");
  }
  return PV_OK;
}

/* db (optional) and the cdiv(N,64) partial sums of <dy, y - bias> in one pass over dy */
int pv_spectral_norm_reduce(const float* dy, const float* y, const float* bias, int M, int N, float* db,
                            int accumulate_db, float* cpart, void* stream) {
  PV_REQUIRE(dy && y && cpart && M > 0 && N > 0, "pv_spectral_norm_reduce: bad argument");
  hipLaunchKernelGGL(sn_reduce_kernel, dim3(cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, dy, y, bias, M, N, db,
                     accumulate_db, cpart);
  PV_LAUNCH_CHECK("sn_reduce_kernel");
  return PV_OK;
}

int pv_spectral_norm_bwd(const float* dy, const float* y, const float* bias, long M, int n_out, int n_in, const float* u,
                         const float* v, const float* inv_sigma, float* scratch1, float* dW, void* stream) {
  PV_REQUIRE(dy && y && u && v && inv_sigma && scratch1 && dW && M > 0, "pv_spectral_norm_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_dot_kernel, dim3(1), dim3(256), 0, st, dy, y, bias, M, n_out, scratch1);
  PV_LAUNCH_CHECK("sn_dot_kernel");
  const long n = (long)n_out * n_in;
  hipLaunchKernelGGL(sn_rank1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dW, u, v, scratch1, inv_sigma,
                     n_out, n_in);
  PV_LAUNCH_CHECK("sn_rank1_kernel");
  return PV_OK;
}

int pv_l2norm_rows_fwd(const float* x, int rows, int D, float* a, float* nrm, void* stream) {
  PV_REQUIRE(x && a && nrm && rows > 0 && D > 0, "pv_l2norm_rows_fwd: bad argument");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, D, a, nrm);
  PV_LAUNCH_CHECK("l2norm_fwd_kernel");
  return PV_OK;
}

int pv_l2norm_rows_bwd(const float* a, const float* nrm, const float* da, int rows, int D, float* dx, void* stream) {
  PV_REQUIRE(a && nrm && da && dx && rows > 0 && D > 0, "pv_l2norm_rows_bwd: bad argument");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, nrm, da, rows, D, dx);
  PV_LAUNCH_CHECK("l2norm_bwd_kernel");
  return PV_OK;
}

int pv_infonce_fwd(const float* S, int B, float* lse_r, float* lse_c, float* loss, void* stream) {
  PV_REQUIRE(S && lse_r && lse_c && loss && B > 0, "pv_infonce_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(infonce_lse_kernel, dim3(cdiv(B, 4), 2), dim3(256), 0, st, S, B, lse_r, lse_c);
  PV_LAUNCH_CHECK("infonce_lse_kernel");
  hipLaunchKernelGGL(infonce_loss_kernel, dim3(1), dim3(256), 0, st, S, lse_r, lse_c, B, loss);
  PV_LAUNCH_CHECK("infonce_loss_kernel");
  return PV_OK;
}

int pv_infonce_bwd(const float* S, const float* lse_r, const float* lse_c, const float* gout, int B, float* dS,
                   void* stream) {
  PV_REQUIRE(S && lse_r && lse_c && gout && dS && B > 0, "pv_infonce_bwd: bad argument");
  const long n = (long)B * B;
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, lse_r,
                     lse_c, gout, B, dS);
  PV_LAUNCH_CHECK("infonce_bwd_kernel");
  return PV_OK;
}

}  // extern "C"
