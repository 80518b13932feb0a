// Fused elementwise / row-reduce passes of the DLS_VAE step for gfx950 (HBM-bound, one pass each):
//   * layout helpers: fp32 -> bf16 cast-and-pack (K zero-padded), LDS-tiled transposes
//   * reparameterisation + Gaussian KL, forward and backward   (dialog_models.py:191-203, :323-325,
//     :3206-3207; criterions.py:145-155; dialog_models.py:3215-3221)
//   * Gumbel-softmax with straight-through one-hot              (nn_lib.py:226-265)
//   * categorical KL / entropy rows                             (criterions.py:157-175, :196-211)
//   * masked sum (loss assembly), dense log-softmax backward
#include "common.h"
#include <atomic>
#include "../../include/pevae_hip.h"

namespace {

// dst[(i0*n1p+i1)*ldd + c] = bf16(src[i0*s0 + i1*s1 + c*sc]), c < C; zero for C <= c < ldd  (i1 < n1 <= n1p)
__global__ void cast_pack_kernel(const float* __restrict__ src, int n0, int n1, int n1p, int C, long s0, long s1,
                                 long sc, bf16_t* __restrict__ dst, int ldd) {
  const long total = (long)n0 * n1 * ldd;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldd);
    const long r = i / ldd;
    const int i1 = (int)(r % n1);
    const long i0 = r / n1;
    dst[(i0 * n1p + i1) * ldd + c] = c < C ? f2bf(src[i0 * s0 + i1 * s1 + c * sc]) : (bf16_t)0;
  }
}

// the same for unit column stride and 4-element alignment: one 16-byte load and one 8-byte store per thread
__global__ void cast_pack_vec4_kernel(const float* __restrict__ src, int n0, int n1, int n1p, int C, long s0, long s1,
                                      bf16_t* __restrict__ dst, int ldd) {
  const int q = ldd / 4;
  const long total = (long)n0 * n1 * q;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % q) * 4;
    const long r = i / q;
    const int i1 = (int)(r % n1);
    const long i0 = r / n1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < C) v = *(const f32x4*)(src + i0 * s0 + i1 * s1 + c);
    ushort4 o;
    o.x = f2bf(v[0]); o.y = f2bf(v[1]); o.z = f2bf(v[2]); o.w = f2bf(v[3]);
    *(ushort4*)(dst + (i0 * n1p + i1) * ldd + c) = o;
  }
}

// dst[c*ldd + r] = bf16(src[r*lds + c]) for r < R, c < C; dst columns R..Rpad zero-filled.  64x64 LDS tiles.
template <typename TS>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const TS* __restrict__ src, int R, int C, long lds_,
                                                              bf16_t* __restrict__ dst, long ldd, int Rpad) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    bf16_t v = 0;
    if (r < R && c < C) {
      if constexpr (sizeof(TS) == 4) v = f2bf((float)src[(size_t)r * lds_ + c]);
      else v = (bf16_t)src[(size_t)r * lds_ + c];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) dst[(size_t)c * ldd + r] = tile[tx][i];
  }
}

// The same with 16-byte global accesses on both sides (source rows read as 8 bf16 / 4 fp32 per lane, destination rows
// written as 8 bf16 per lane); the element-wise kernel above moved 128 bytes per wave instruction.  Needs C % 8 == 0 for
// nothing (columns are clamped), but 16-byte aligned rows on both sides: see pv_transpose_to_bf16.
template <typename TS>
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(const TS* __restrict__ src, int R, int C, long lds_,
                                                                  bf16_t* __restrict__ dst, long ldd, int Rpad) {
  constexpr int EPL = 16 / sizeof(TS);                 // source elements per 16-byte load: 8 (bf16) or 4 (fp32)
  constexpr int CPR = 64 / EPL;                        // loads per 64-column tile row
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];   // [column][row], 144-byte pitch
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int it = 0; it < 64 * CPR / 256; ++it) {
    const int idx = threadIdx.x + it * 256, r = idx / CPR, ch = idx % CPR, c = c0 + ch * EPL;
    bf16_t v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = 0;
    if (r0 + r < R && c < C) {                         // C is a multiple of EPL (checked by the host): whole chunk valid
      const TS* p = src + (size_t)(r0 + r) * lds_ + c;
      if constexpr (sizeof(TS) == 4) {
        const f32x4 q = *(const f32x4*)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f2bf(q[e]);
      } else {
        const uint4 q = *(const uint4*)p;
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)(w[e >> 1] >> (16 * (e & 1)));
      }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) tile[ch * EPL + e][r] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + it * 256, cc = idx >> 3, rch = idx & 7;
    if (c0 + cc < C && r0 + rch * 8 < Rpad)
      *(uint4*)(dst + (size_t)(c0 + cc) * ldd + r0 + rch * 8) = *(const uint4*)(&tile[cc][rch * 8]);
  }
}

// mode 0: logvar = log sigma^2 (NormKLLoss);  mode 1: inputs are log sigma (gaussian_analytical_kl).
// z = mu_q + exp(lv_q) * eps in both (the reference's std = exp(logvar), dialog_models.py:194).
__global__ void reparam_kl_fwd_kernel(const float* __restrict__ mq, const float* __restrict__ lq,
                                      const float* __restrict__ mp, const float* __restrict__ lp,
                                      const float* __restrict__ eps, int B, int Z, int mode, float* __restrict__ z,
                                      float* __restrict__ kl_row) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float acc = 0.f;
  for (int d = lane; d < Z; d += 64) {
    const size_t i = (size_t)b * Z + d;
    const float a = mq[i], la = lq[i];
    if (z) z[i] = eps ? a + expf(la) * eps[i] : a;
    if (kl_row) {
      const float c = mp[i], lc = lp[i];
      const float diff = c - a;
      if (mode == 0) acc += 1.f + (la - lc) - diff * diff * expf(-lc) - expf(la - lc);
      else acc += -0.5f + lc - la + 0.5f * (expf(2.f * la) + diff * diff) * expf(-2.f * lc);
    }
  }
  if (kl_row) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) kl_row[b] = mode == 0 ? -0.5f * acc : acc;
  }
}

__global__ void reparam_kl_bwd_kernel(const float* __restrict__ mq, const float* __restrict__ lq,
                                      const float* __restrict__ mp, const float* __restrict__ lp,
                                      const float* __restrict__ eps, const float* __restrict__ dz,
                                      const float* __restrict__ dkl_row, int B, int Z, int mode,
                                      float* __restrict__ dmq, float* __restrict__ dlq, float* __restrict__ dmp,
                                      float* __restrict__ dlp) {
  const long n = (long)B * Z;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / Z);
    const float a = mq[i], la = lq[i];
    float gmq = 0.f, glq = 0.f, gmp = 0.f, glp = 0.f;
    if (dz) {
      gmq += dz[i];
      if (eps) glq += dz[i] * expf(la) * eps[i];
    }
    if (dkl_row) {
      const float g = dkl_row[b], c = mp[i], lc = lp[i], diff = c - a;
      if (mode == 0) {
        const float ip = expf(-lc), ratio = expf(la - lc);
        // kl = -0.5 * sum(1 + la - lc - diff^2*ip - ratio)
        gmq += g * (-0.5f) * (2.f * diff * ip);
        gmp += g * (-0.5f) * (-2.f * diff * ip);
        glq += g * (-0.5f) * (1.f - ratio);
        glp += g * (-0.5f) * (-1.f + diff * diff * ip + ratio);
      } else {
        const float ip2 = expf(-2.f * lc), e2a = expf(2.f * la);
        // kl = sum(-0.5 + lc - la + 0.5*(e2a + diff^2)*ip2)
        gmq += g * (-diff * ip2);
        gmp += g * (diff * ip2);
        glq += g * (-1.f + e2a * ip2);
        glp += g * (1.f - (e2a + diff * diff) * ip2);
      }
    }
    if (dmq) dmq[i] = gmq;
    if (dlq) dlq[i] = glq;
    if (dmp) dmp[i] = gmp;
    if (dlp) dlp[i] = glp;
  }
}

// one thread per row (k is small: 5 in the reference config, <= 64 supported)
__global__ void gumbel_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ u, int rows, int k,
                                  float inv_tau, int hard, float* __restrict__ y_soft, float* __restrict__ y,
                                  int64_t* __restrict__ ids) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* lg = logits + (size_t)r * k;
  const float* ur = u + (size_t)r * k;
  float m = -INFINITY;
  for (int j = 0; j < k; ++j) {
    const float gn = -logf(-logf(ur[j] + 1e-20f) + 1e-20f);    // nn_lib.py:232
    m = fmaxf(m, (lg[j] + gn) * inv_tau);
  }
  float s = 0.f;
  for (int j = 0; j < k; ++j) {
    const float gn = -logf(-logf(ur[j] + 1e-20f) + 1e-20f);
    s += expf((lg[j] + gn) * inv_tau - m);
  }
  int best = 0;
  float bestv = -1.f;
  for (int j = 0; j < k; ++j) {
    const float gn = -logf(-logf(ur[j] + 1e-20f) + 1e-20f);
    const float p = expf((lg[j] + gn) * inv_tau - m) / s;
    y_soft[(size_t)r * k + j] = p;
    if (p > bestv) {           // first maximum, as torch.max
      bestv = p;
      best = j;
    }
  }
  if (ids) ids[r] = best;
  for (int j = 0; j < k; ++j) {
    const float p = y_soft[(size_t)r * k + j];
    // straight-through value (onehot - y).detach() + y, evaluated in the reference's order (nn_lib.py:256)
    y[(size_t)r * k + j] = hard ? ((j == best ? 1.f : 0.f) - p) + p : p;
  }
}

__global__ void softmax_bwd_rows_kernel(const float* __restrict__ y_soft, const float* __restrict__ dy, int rows,
                                        int k, float scale, float* __restrict__ dlogits) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float dot = 0.f;
  for (int j = 0; j < k; ++j) dot += dy[(size_t)r * k + j] * y_soft[(size_t)r * k + j];
  for (int j = 0; j < k; ++j)
    dlogits[(size_t)r * k + j] = scale * y_soft[(size_t)r * k + j] * (dy[(size_t)r * k + j] - dot);
}

// rows of sum_j q*(log q - log p)  (mode 0, CatKLLoss) or  sum_j -q*log q (mode 1, Entropy);  log_p may be a
// single broadcast scalar (lp_n == 1: log(1/k), dialog_models.py:3426) or a full [rows,k] matrix.
__global__ void cat_rows_fwd_kernel(const float* __restrict__ log_q, const float* __restrict__ log_p, long lp_n,
                                    int rows, int k, int mode, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int j = 0; j < k; ++j) {
    const float lq = log_q[(size_t)r * k + j], q = expf(lq);
    if (mode == 0) s += q * (lq - (lp_n == 1 ? log_p[0] : log_p[(size_t)r * k + j]));
    else s += -lq * q;
  }
  out[r] = s;
}

__global__ void cat_rows_bwd_kernel(const float* __restrict__ log_q, const float* __restrict__ log_p, long lp_n,
                                    const float* __restrict__ drow, int rows, int k, int mode,
                                    float* __restrict__ dlog_q, float* __restrict__ dlog_p) {
  const long n = (long)rows * k;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float lq = log_q[i], q = expf(lq), g = drow[i / k];
  if (mode == 0) {
    const float lp = lp_n == 1 ? log_p[0] : log_p[i];
    dlog_q[i] = g * q * (lq - lp + 1.f);
    if (dlog_p) dlog_p[i] = -g * q;
  } else {
    dlog_q[i] = -g * q * (lq + 1.f);
  }
}

// out[0] = sum_i x[i] * (mask ? mask[i] != ignore : 1),  out[1] = count of unmasked.  Single block, fixed order.
__global__ __launch_bounds__(1024) void masked_sum_kernel(const float* __restrict__ x, const int64_t* __restrict__ mask,
                                                           int64_t ignore, long n, float* __restrict__ out) {
  __shared__ float ss[1024];
  __shared__ float sc[1024];
  float s = 0.f, c = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const bool on = !mask || mask[i] != ignore;
    if (on) {
      s += x[i];
      c += 1.f;
    }
  }
  ss[threadIdx.x] = s;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      ss[threadIdx.x] += ss[threadIdx.x + o];
      sc[threadIdx.x] += sc[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = ss[0];
    out[1] = sc[0];
  }
}

// backward of a dense log-softmax row: g = (dlogp - exp(logp) * sum(dlogp)) * inv_temp, emitted as bf16 in both
// P[T][ldp] and (through the caller's transpose) P^T.  One workgroup per row.
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float* __restrict__ logp,
                                                              const float* __restrict__ dlogp, int T, int V,
                                                              float inv_temp, bf16_t* __restrict__ P, int ldp,
                                                              float* __restrict__ rowsum_out) {
  __shared__ float red[256];
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) s += dlogp[(size_t)r * V + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float tot = red[0];
  if (threadIdx.x == 0 && rowsum_out) rowsum_out[r] = tot;
  for (int c = threadIdx.x; c < ldp; c += 256) {
    float g = 0.f;
    if (c < V) g = (dlogp[(size_t)r * V + c] - expf(logp[(size_t)r * V + c]) * tot) * inv_temp;
    P[(size_t)r * ldp + c] = f2bf(g);
  }
}

// column sums of a bf16 matrix [R][ld] over rows, fp32 accumulate (bias gradient of the dense path)
__global__ void colsum_bf16_kernel(const bf16_t* __restrict__ P, int R, int ld, int n, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += bf2f(P[(size_t)r * ld + c]);
  out[c] = s;
}

// out[c] (+)= sum_r X[r][c].  Block = 64 columns x 4 row phases; each thread keeps 8 independent loads in flight
// over its rows (r = y, y+4, ...), then the 4 phases are added in a fixed order: bitwise reproducible.
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ X, int R, long ld, int n,
                                                          float* __restrict__ out, int accumulate) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, y = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < n) {
    int r = y;
    for (; r + 28 < R; r += 32) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = X[(size_t)(r + 4 * k) * ld + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; r < R; r += 4) s += X[(size_t)r * ld + c];
  }
  part[y][cl] = s;
  __syncthreads();
  if (y == 0 && c < n) {
    const float t = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// dense-input NLL pick: out[t] = -logp[t][target[t]] (0 where target == ignore)   (criterions.py:57-63)
__global__ void pick_nll_kernel(const float* __restrict__ logp, const int64_t* __restrict__ target, int64_t ignore,
                                int T, int V, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t c = target[t];
  out[t] = (c == ignore || c < 0 || c >= V) ? 0.f : -logp[(size_t)t * V + c];
}

__global__ void pick_nll_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ target,
                                    int64_t ignore, int T, int V, float* __restrict__ dlogp) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t c = target[t];
  if (!(c == ignore || c < 0 || c >= V)) dlogp[(size_t)t * V + c] = -dout[t];
}

inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int pv_cast_pack_bf16(const float* src, int n0, int n1, int n1p, int C, long s0, long s1, long sc, void* dst,
                      int ldd, void* stream) {
  PV_REQUIRE(src && dst && n0 > 0 && n1 > 0 && n1p >= n1 && C > 0 && ldd >= C, "pv_cast_pack_bf16: bad argument");
  if (sc == 1 && C % 4 == 0 && ldd % 4 == 0 && s0 % 4 == 0 && s1 % 4 == 0 && ((uintptr_t)src & 15) == 0 &&
      ((uintptr_t)dst & 7) == 0) {
    hipLaunchKernelGGL(cast_pack_vec4_kernel, dim3(grid_for((long)n0 * n1 * (ldd / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, src, n0, n1, n1p, C, s0, s1, (bf16_t*)dst, ldd);
    PV_LAUNCH_CHECK("cast_pack_vec4_kernel");
    return PV_OK;
  }
  hipLaunchKernelGGL(cast_pack_kernel, dim3(grid_for((long)n0 * n1 * ldd, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, n0, n1, n1p, C, s0, s1, sc, (bf16_t*)dst, ldd);
  PV_LAUNCH_CHECK("cast_pack_kernel");
  return PV_OK;
}

int pv_transpose_to_bf16(const void* src, int src_is_f32, int R, int C, long lds_, void* dst, long ldd, int Rpad,
                         void* stream) {
  PV_REQUIRE(src && dst && R > 0 && C > 0 && lds_ >= C && Rpad >= R && ldd >= Rpad, "pv_transpose_to_bf16: bad argument");
  dim3 grid(cdiv(C, 64), cdiv(Rpad, 64));
  const int epl = src_is_f32 ? 4 : 8;
  if (C % epl == 0 && lds_ % epl == 0 && ((uintptr_t)src & 15) == 0 && Rpad % 8 == 0 && ldd % 8 == 0 &&
      ((uintptr_t)dst & 15) == 0) {
    if (src_is_f32)
      hipLaunchKernelGGL(transpose_bf16_vec_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, R, C,
                         lds_, (bf16_t*)dst, ldd, Rpad);
    else
      hipLaunchKernelGGL(transpose_bf16_vec_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                         R, C, lds_, (bf16_t*)dst, ldd, Rpad);
    PV_LAUNCH_CHECK("transpose_bf16_vec_kernel");
    return PV_OK;
  }
  if (src_is_f32)
    hipLaunchKernelGGL(transpose_bf16_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, R, C,
                       lds_, (bf16_t*)dst, ldd, Rpad);
  else
    hipLaunchKernelGGL(transpose_bf16_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, R,
                       C, lds_, (bf16_t*)dst, ldd, Rpad);
  PV_LAUNCH_CHECK("transpose_bf16_kernel");
  return PV_OK;
}

int pv_reparam_kl_fwd(const float* mu_q, const float* lv_q, const float* mu_p, const float* lv_p, const float* eps,
                      int B, int Z, int mode, float* z, float* kl_row, void* stream) {
  PV_REQUIRE(mu_q && lv_q && B > 0 && Z > 0 && (z || kl_row), "pv_reparam_kl_fwd: bad argument");
  PV_REQUIRE(!kl_row || (mu_p && lv_p), "pv_reparam_kl_fwd: KL needs the prior");
  PV_REQUIRE(mode == 0 || mode == 1, "pv_reparam_kl_fwd: mode must be 0 (log sigma^2) or 1 (log sigma)");
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, mu_q, lv_q, mu_p,
                     lv_p, eps, B, Z, mode, z, kl_row);
  PV_LAUNCH_CHECK("reparam_kl_fwd_kernel");
  return PV_OK;
}

int pv_reparam_kl_bwd(const float* mu_q, const float* lv_q, const float* mu_p, const float* lv_p, const float* eps,
                      const float* dz, const float* dkl_row, int B, int Z, int mode, float* dmu_q, float* dlv_q,
                      float* dmu_p, float* dlv_p, void* stream) {
  PV_REQUIRE(mu_q && lv_q && B > 0 && Z > 0, "pv_reparam_kl_bwd: bad argument");
  PV_REQUIRE(!dkl_row || (mu_p && lv_p), "pv_reparam_kl_bwd: KL needs the prior");
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(grid_for((long)B * Z, 256)), dim3(256), 0, (hipStream_t)stream, mu_q,
                     lv_q, mu_p, lv_p, eps, dz, dkl_row, B, Z, mode, dmu_q, dlv_q, dmu_p, dlv_p);
  PV_LAUNCH_CHECK("reparam_kl_bwd_kernel");
  return PV_OK;
}

}  // extern "C"

namespace {

// ---- one level of DLS_Auto's train branch (dialog_models.py:3285-3311) as two elementwise launches -----------------
// The level's glue -- chunk(2), reparameterise, + act2, + layer(output), two detached-direction KLs, their batch means
// and weights -- was ~15 four-microsecond launches forward and as many backward.
//   Q = dis_layer output  [B][2n] = (mean | logsigma) of the posterior;  Pr = dis_layer2 output [B][2n] = prior
//   ls  = mean + exp(logsigma) * eps + a2                      (last_sample, :3299-3300)
//   out = lay + ls                                             (:3310)
//   kl  = (ew + cw) * mean_b sum_d KL(prior || posterior)      (:3305-3306; as a VALUE both detached-direction terms are
//         this number, their gradients differ: ew scales the prior's, cw the posterior's)
// Workgroup g handles a contiguous run of elements and leaves its KL partial in part[g]; the workgroup that arrives LAST
// at the launch's counter adds the partials in index order (fixed-order reduction: run-to-run identical whichever
// workgroup that is) and resets the counter for the next launch.  One workgroup alone took 56 us at the widest level
// (B * n = 131072); the fold used to be a second launch (5 us + a launch gap on the latent stack's chain, three times
// per step).  Partials and counter travel with agent-scope accesses: the workgroups sit on different XCDs.
constexpr int DLS_WG_ELEMS = 2048, DLS_MAX_PART = 256, DLS_COUNTERS = 16;
__device__ unsigned g_dls_counter[DLS_COUNTERS];            // zero when the library is loaded; every launch leaves its own at zero

__global__ __launch_bounds__(256) void dls_level_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Pr,
                                                             const float* __restrict__ a2, const float* __restrict__ eps,
                                                             const float* __restrict__ lay, int B, int n, float ew, float cw,
                                                             float* __restrict__ ls, float* __restrict__ out,
                                                             float* __restrict__ part, int per_wg,
                                                             const float* __restrict__ a1n, const float* __restrict__ a2n, int wn,
                                                             float* __restrict__ q_in, float* __restrict__ p_in,
                                                             unsigned* __restrict__ counter, float scale,
                                                             float* __restrict__ kl, const float* __restrict__ kl_add) {
  __shared__ float red[256];
  __shared__ int last;
  float acc = 0.f;
  const int lo = blockIdx.x * per_wg, hi = min(B * n, lo + per_wg);
  const int Wn = wn + n;
  if (q_in) {            // the next level's inputs (:3289-3297): q_in = [a1' + a2' | ls], p_in = [a2' | t_mean]
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * wn; i += gridDim.x * 256) {
      const int b = i / wn, c = i % wn;
      const float x2 = a2n[i];
      q_in[(size_t)b * Wn + c] = a1n[i] + x2;
      p_in[(size_t)b * Wn + c] = x2;
    }
  }
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const int b = i / n, d = i % n;
    const float m = Q[(size_t)b * 2 * n + d], l = Q[(size_t)b * 2 * n + n + d];
    const float tm = Pr[(size_t)b * 2 * n + d], tl = Pr[(size_t)b * 2 * n + n + d];
    const float s = m + expf(l) * eps[i] + a2[i];
    ls[i] = s;
    out[i] = lay[i] + s;
    if (q_in) {
      q_in[(size_t)b * Wn + wn + d] = s;
      p_in[(size_t)b * Wn + wn + d] = tm;
    }
    const float diff = tm - m;
    acc += -0.5f + l - tl + 0.5f * (expf(2.f * tl) + diff * diff) * expf(-2.f * l);     // KL(t || q), log-sigma inputs
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(part + blockIdx.x, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = old == gridDim.x - 1;
  }
  __syncthreads();
  if (!last || threadIdx.x >= 64) return;
  float s = 0.f;
  const int nparts = gridDim.x;
  for (int i0 = 0; i0 < nparts; i0 += 64) {                 // 64 partials per load, added in index order
    const float v = i0 + (int)threadIdx.x < nparts
                        ? __hip_atomic_load(part + i0 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    for (int k = 0; k < 64 && i0 + k < nparts; ++k) s += __shfl(v, k);
  }
  if (threadIdx.x == 0) {
    kl[0] = kl_add ? kl_add[0] + scale * s : scale * s;      // (kl_add: the running loss this term joins)
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// d_ls / d_out / g_kl: gradients of the three outputs (d_ls, d_out may be NULL).  dQ, dPr [B][2n]; da2, dlay [B][n].
__global__ void dls_level_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ Pr, const float* __restrict__ eps,
                                     const float* __restrict__ d_ls, const float* __restrict__ d_out,
                                     const float* __restrict__ g_kl, int B, int n, float ew, float cw,
                                     float* __restrict__ dQ, float* __restrict__ dPr, float* __restrict__ da2,
                                     float* __restrict__ dlay, const float* __restrict__ dq_in, const float* __restrict__ dp_in,
                                     int wn, float* __restrict__ da1n, float* __restrict__ da2n) {
  const float g = g_kl ? g_kl[0] / B : 0.f;
  const int Wn = wn + n;
  if (da1n) {            // through the next level's inputs: q_in = [a1' + a2' | ls], p_in = [a2' | t_mean]
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * wn; i += gridDim.x * blockDim.x) {
      const int b = i / wn, c = i % wn;
      const float gq = dq_in ? dq_in[(size_t)b * Wn + c] : 0.f, gp = dp_in ? dp_in[(size_t)b * Wn + c] : 0.f;
      da1n[i] = gq;
      da2n[i] = gq + gp;
    }
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * n; i += gridDim.x * blockDim.x) {
    const int b = i / n, d = i % n;
    const size_t im = (size_t)b * 2 * n + d, il = im + n;
    const float m = Q[im], l = Q[il], tm = Pr[im], tl = Pr[il];
    const float gls = (d_ls ? d_ls[i] : 0.f) + (da1n && dq_in ? dq_in[(size_t)b * Wn + wn + d] : 0.f);
    const float gtm = da1n && dp_in ? dp_in[(size_t)b * Wn + wn + d] : 0.f;
    const float go = d_out ? d_out[i] : 0.f, gs = gls + go;
    const float diff = tm - m, ip2 = expf(-2.f * l), e2t = expf(2.f * tl);
    // posterior (m, l) is "2" of KL(1 || 2), weight cw; prior (tm, tl) is "1", weight ew
    dQ[im] = gs + g * cw * (-diff * ip2);
    dQ[il] = gs * expf(l) * eps[i] + g * cw * (1.f - (e2t + diff * diff) * ip2);
    dPr[im] = gtm + g * ew * (diff * ip2);
    dPr[il] = g * ew * (-1.f + e2t * ip2);
    da2[i] = gs;
    dlay[i] = go;
  }
}

// q_in = [a1 + a2 | ls], p_in = [a2 | lt]   (:3289-3291, :3295-3297): one launch instead of an add and two cats
__global__ void dls_inputs_fwd_kernel(const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ ls,
                                      const float* __restrict__ lt, int B, int w, int e, float* __restrict__ q_in,
                                      float* __restrict__ p_in) {
  const int W = w + e;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * W; i += gridDim.x * blockDim.x) {
    const int b = i / W, c = i % W;
    if (c < w) {
      const float x2 = a2[(size_t)b * w + c];
      q_in[i] = a1[(size_t)b * w + c] + x2;
      p_in[i] = x2;
    } else {
      q_in[i] = ls[(size_t)b * e + c - w];
      p_in[i] = lt[(size_t)b * e + c - w];
    }
  }
}
__global__ void dls_inputs_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ dp, int B, int w, int e,
                                      float* __restrict__ da1, float* __restrict__ da2, float* __restrict__ dls,
                                      float* __restrict__ dlt) {
  const int W = w + e;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * W; i += gridDim.x * blockDim.x) {
    const int b = i / W, c = i % W;
    const float gq = dq ? dq[i] : 0.f, gp = dp ? dp[i] : 0.f;
    if (c < w) {
      da1[(size_t)b * w + c] = gq;
      da2[(size_t)b * w + c] = gq + gp;
    } else {
      dls[(size_t)b * e + c - w] = gq;
      dlt[(size_t)b * e + c - w] = gp;
    }
  }
}

}  // namespace

extern "C" {

int pv_dls_level_fwd(const float* Q, const float* Pr, const float* a2, const float* eps, const float* lay, int B, int n,
                     float ew, float cw, float* ls, float* out, float* kl, float* kl_part, void* stream) {
  return pv_dls_level_next_fwd(Q, Pr, a2, eps, lay, B, n, ew, cw, ls, out, kl, kl_part, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

int pv_dls_level_next_fwd(const float* Q, const float* Pr, const float* a2, const float* eps, const float* lay, int B, int n,
                          float ew, float cw, float* ls, float* out, float* kl, float* kl_part, const float* a1n,
                          const float* a2n, int wn, float* q_in, float* p_in, void* stream) {
  return pv_dls_level_next_add_fwd(Q, Pr, a2, eps, lay, B, n, ew, cw, ls, out, kl, kl_part, a1n, a2n, wn, q_in, p_in, nullptr,
                                   stream);
}

int pv_dls_level_next_add_fwd(const float* Q, const float* Pr, const float* a2, const float* eps, const float* lay, int B, int n,
                              float ew, float cw, float* ls, float* out, float* kl, float* kl_part, const float* a1n,
                              const float* a2n, int wn, float* q_in, float* p_in, const float* kl_add, void* stream) {
  PV_REQUIRE(Q && Pr && a2 && eps && lay && ls && out && kl && kl_part && B > 0 && n > 0, "pv_dls_level_fwd: bad argument");
  PV_REQUIRE(!q_in || (a1n && a2n && p_in && wn > 0), "pv_dls_level_next_fwd: the next level's inputs need a1n, a2n, p_in, wn");
  const long total = (long)B * n;
  PV_REQUIRE(total < (1l << 31), "pv_dls_level_fwd: B * n too large");
  int wgs = (int)((total + DLS_WG_ELEMS - 1) / DLS_WG_ELEMS);
  if (wgs > DLS_MAX_PART) wgs = DLS_MAX_PART;
  const int per = (int)((total + wgs - 1) / wgs);
  // a counter of its own per launch in flight: 16 in rotation (launches that could share one are 16 launches apart on
  // streams that a step orders anyway)
  static std::atomic<unsigned> next_counter{0};
  static std::atomic<unsigned*> counters_of[64];            // per device: each GPU holds its own copy of the array
  int dev = 0;
  PV_HIP(hipGetDevice(&dev));
  PV_REQUIRE(dev >= 0 && dev < 64, "pv_dls_level_fwd: device ordinal %d out of range", dev);
  unsigned* counters = counters_of[dev].load();
  if (!counters) {
    PV_HIP(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_dls_counter)));
    counters_of[dev].store(counters);
  }
  unsigned* counter = counters + next_counter.fetch_add(1) % DLS_COUNTERS;
  hipLaunchKernelGGL(dls_level_fwd_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, Q, Pr, a2, eps, lay, B, n, ew, cw, ls,
                     out, kl_part, per, a1n, a2n, wn, q_in, p_in, counter, (ew + cw) / B, kl, kl_add);
  PV_LAUNCH_CHECK("dls_level_fwd_kernel");
  return PV_OK;
}

int pv_dls_level_bwd(const float* Q, const float* Pr, const float* eps, const float* d_ls, const float* d_out,
                     const float* g_kl, int B, int n, float ew, float cw, float* dQ, float* dPr, float* da2, float* dlay,
                     void* stream) {
  return pv_dls_level_next_bwd(Q, Pr, eps, d_ls, d_out, g_kl, B, n, ew, cw, dQ, dPr, da2, dlay, nullptr, nullptr, 0, nullptr,
                               nullptr, stream);
}

int pv_dls_level_next_bwd(const float* Q, const float* Pr, const float* eps, const float* d_ls, const float* d_out,
                          const float* g_kl, int B, int n, float ew, float cw, float* dQ, float* dPr, float* da2, float* dlay,
                          const float* dq_in, const float* dp_in, int wn, float* da1n, float* da2n, void* stream) {
  PV_REQUIRE(Q && Pr && eps && dQ && dPr && da2 && dlay && B > 0 && n > 0, "pv_dls_level_bwd: bad argument");
  PV_REQUIRE(!da1n || (da2n && wn > 0), "pv_dls_level_next_bwd: the next level's input gradients need da2n and wn");
  hipLaunchKernelGGL(dls_level_bwd_kernel, dim3(grid_for((long)B * (n > wn ? n : wn), 256)), dim3(256), 0, (hipStream_t)stream,
                     Q, Pr, eps, d_ls, d_out, g_kl, B, n, ew, cw, dQ, dPr, da2, dlay, dq_in, dp_in, wn, da1n, da2n);
  PV_LAUNCH_CHECK("dls_level_bwd_kernel");
  return PV_OK;
}

int pv_dls_inputs_fwd(const float* a1, const float* a2, const float* ls, const float* lt, int B, int w, int e, float* q_in,
                      float* p_in, void* stream) {
  PV_REQUIRE(a1 && a2 && ls && lt && q_in && p_in && B > 0 && w > 0 && e > 0, "pv_dls_inputs_fwd: bad argument");
  hipLaunchKernelGGL(dls_inputs_fwd_kernel, dim3(grid_for((long)B * (w + e), 256)), dim3(256), 0, (hipStream_t)stream, a1, a2,
                     ls, lt, B, w, e, q_in, p_in);
  PV_LAUNCH_CHECK("dls_inputs_fwd_kernel");
  return PV_OK;
}

int pv_dls_inputs_bwd(const float* dq, const float* dp, int B, int w, int e, float* da1, float* da2, float* dls, float* dlt,
                      void* stream) {
  PV_REQUIRE(da1 && da2 && dls && dlt && B > 0 && w > 0 && e > 0, "pv_dls_inputs_bwd: bad argument");
  hipLaunchKernelGGL(dls_inputs_bwd_kernel, dim3(grid_for((long)B * (w + e), 256)), dim3(256), 0, (hipStream_t)stream, dq, dp,
                     B, w, e, da1, da2, dls, dlt);
  PV_LAUNCH_CHECK("dls_inputs_bwd_kernel");
  return PV_OK;
}

int pv_gumbel_softmax_fwd(const float* logits, const float* u, int rows, int k, float tau, int hard, float* y_soft,
                          float* y, int64_t* ids, void* stream) {
  PV_REQUIRE(logits && u && y_soft && y && rows > 0 && k > 0 && tau > 0.f, "pv_gumbel_softmax_fwd: bad argument");
  hipLaunchKernelGGL(gumbel_fwd_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, logits, u, rows, k,
                     1.f / tau, hard, y_soft, y, ids);
  PV_LAUNCH_CHECK("gumbel_fwd_kernel");
  return PV_OK;
}

int pv_softmax_bwd_rows(const float* y_soft, const float* dy, int rows, int k, float scale, float* dlogits,
                        void* stream) {
  PV_REQUIRE(y_soft && dy && dlogits && rows > 0 && k > 0, "pv_softmax_bwd_rows: bad argument");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, y_soft, dy,
                     rows, k, scale, dlogits);
  PV_LAUNCH_CHECK("softmax_bwd_rows_kernel");
  return PV_OK;
}

int pv_cat_rows_fwd(const float* log_q, const float* log_p, long log_p_numel, int rows, int k, int mode, float* out,
                    void* stream) {
  PV_REQUIRE(log_q && out && rows > 0 && k > 0 && (mode == 1 || log_p), "pv_cat_rows_fwd: bad argument");
  PV_REQUIRE(mode == 1 || log_p_numel == 1 || log_p_numel == (long)rows * k, "pv_cat_rows_fwd: log_p must be scalar or [rows,k]");
  hipLaunchKernelGGL(cat_rows_fwd_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, log_q, log_p,
                     log_p_numel, rows, k, mode, out);
  PV_LAUNCH_CHECK("cat_rows_fwd_kernel");
  return PV_OK;
}

int pv_cat_rows_bwd(const float* log_q, const float* log_p, long log_p_numel, const float* drow, int rows, int k,
                    int mode, float* dlog_q, float* dlog_p, void* stream) {
  PV_REQUIRE(log_q && drow && dlog_q && rows > 0 && k > 0 && (mode == 1 || log_p), "pv_cat_rows_bwd: bad argument");
  PV_REQUIRE(!dlog_p || log_p_numel == (long)rows * k, "pv_cat_rows_bwd: dlog_p needs a full log_p");
  hipLaunchKernelGGL(cat_rows_bwd_kernel, dim3(cdiv(rows * k, 256)), dim3(256), 0, (hipStream_t)stream, log_q, log_p,
                     log_p_numel, drow, rows, k, mode, dlog_q, dlog_p);
  PV_LAUNCH_CHECK("cat_rows_bwd_kernel");
  return PV_OK;
}

}  // extern "C"

namespace {
// backward of pv_masked_sum: dx[i] = g[0] where mask[i] != ignore (everywhere without a mask), else 0
__global__ void masked_sum_bwd_kernel(const int64_t* __restrict__ mask, int64_t ignore, const float* __restrict__ g, long n,
                                      float* __restrict__ dx) {
  const float gv = g[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = (!mask || mask[i] != ignore) ? gv : 0.f;
}
// row scale of the vocabulary backward: gscale[t] = dnll[perm ? perm[t] : t] * (target[t] != ignore)
__global__ void vocab_gscale_kernel(const float* __restrict__ dnll, const int64_t* __restrict__ perm,
                                    const int64_t* __restrict__ target, int64_t ignore, int T, float* __restrict__ gscale) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x)
    gscale[t] = target[t] != ignore ? dnll[perm ? perm[t] : t] : 0.f;
}
}  // namespace

extern "C" {

int pv_masked_sum_bwd(const int64_t* mask, int64_t ignore, const float* g, long n, float* dx, void* stream) {
  PV_REQUIRE(g && dx && n >= 0, "pv_masked_sum_bwd: bad argument");
  if (n == 0) return PV_OK;
  hipLaunchKernelGGL(masked_sum_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, mask, ignore, g, n, dx);
  PV_LAUNCH_CHECK("masked_sum_bwd_kernel");
  return PV_OK;
}

int pv_vocab_gscale(const float* dnll, const int64_t* perm, const int64_t* target, int64_t ignore, int T, float* gscale,
                    void* stream) {
  PV_REQUIRE(dnll && target && gscale && T > 0, "pv_vocab_gscale: bad argument");
  hipLaunchKernelGGL(vocab_gscale_kernel, dim3(grid_for(T, 256)), dim3(256), 0, (hipStream_t)stream, dnll, perm, target,
                     ignore, T, gscale);
  PV_LAUNCH_CHECK("vocab_gscale_kernel");
  return PV_OK;
}

// Gradient sums at the fan-out points of the latent stack: dst_i = sum of up to PV_SUM_SRC addends, up to PV_SUM_GROUP
// tensors per launch (the autograd engine adds the gradients of a tensor's consumers one launch at a time otherwise).
// Addends are added left to right (fixed order).
static __global__ void sum_group_kernel(pv_sum_group_desc g) {
  const int i = blockIdx.y;
  const long n4 = g.n[i] >> 2;
  float4* dst = (float4*)g.dst[i];
  const float4* s0 = (const float4*)g.src[i][0];
  const float4* s1 = (const float4*)g.src[i][1];
  const float4* s2 = (const float4*)g.src[i][2];
  const float4* s3 = (const float4*)g.src[i][3];
  for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < n4; k += (long)gridDim.x * blockDim.x) {
    float4 v = s0[k];
    float4 w1, w2, w3;
    if (s1) w1 = s1[k];
    if (s2) w2 = s2[k];
    if (s3) w3 = s3[k];
    if (s1) { v.x += w1.x; v.y += w1.y; v.z += w1.z; v.w += w1.w; }
    if (s2) { v.x += w2.x; v.y += w2.y; v.z += w2.z; v.w += w2.w; }
    if (s3) { v.x += w3.x; v.y += w3.y; v.z += w3.z; v.w += w3.w; }
    dst[k] = v;
  }
}

// NLLEntropy(avg_type 'seq') and Perplexity of one step from the pair {sum of the token NLL, live tokens} in one launch
// each way (criterions.py:64 loss / batch, :137 exp(loss / n_words)): out2 = {total / batch, exp(total / count)};
// backward: d total = g_nll / batch + g_ppl * ppl / count (count carries no gradient).
static __global__ void nll_head_fwd_kernel(const float* __restrict__ sums, float inv_batch, float* __restrict__ out2) {
  const float total = sums[0], count = sums[1];
  out2[0] = total * inv_batch;
  out2[1] = __expf(total / count);
}
static __global__ void nll_head_bwd_kernel(const float* __restrict__ sums, const float* __restrict__ out2, const float* __restrict__ g_nll,
                                    const float* __restrict__ g_ppl, float inv_batch, float* __restrict__ dsums) {
  float d = g_nll ? g_nll[0] * inv_batch : 0.f;
  if (g_ppl) d += g_ppl[0] * out2[1] / sums[1];
  dsums[0] = d;
  dsums[1] = 0.f;
}

int pv_masked_sum(const float* x, const int64_t* mask, int64_t ignore, long n, float* out2, void* stream) {
  PV_REQUIRE(x && out2 && n >= 0, "pv_masked_sum: bad argument");
  hipLaunchKernelGGL(masked_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, mask, ignore, n, out2);
  PV_LAUNCH_CHECK("masked_sum_kernel");
  return PV_OK;
}

int pv_logsoftmax_bwd_bf16(const float* logp, const float* dlogp, int T, int V, float inv_temp, void* P, int ldp,
                           float* dbias, void* stream) {
  PV_REQUIRE(logp && dlogp && P && T > 0 && V > 0 && ldp >= V, "pv_logsoftmax_bwd_bf16: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3(T), dim3(256), 0, st, logp, dlogp, T, V, inv_temp, (bf16_t*)P, ldp,
                     (float*)nullptr);
  PV_LAUNCH_CHECK("logsoftmax_bwd_kernel");
  if (dbias) {
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(V, 256)), dim3(256), 0, st, (const bf16_t*)P, T, ldp, V, dbias);
    PV_LAUNCH_CHECK("colsum_bf16_kernel");
  }
  return PV_OK;
}

int pv_pick_nll_fwd(const float* logp, const int64_t* target, int64_t ignore, int T, int V, float* out,
                    void* stream) {
  PV_REQUIRE(logp && target && out && T > 0 && V > 0, "pv_pick_nll_fwd: bad argument");
  hipLaunchKernelGGL(pick_nll_kernel, dim3(cdiv(T, 256)), dim3(256), 0, (hipStream_t)stream, logp, target, ignore, T,
                     V, out);
  PV_LAUNCH_CHECK("pick_nll_kernel");
  return PV_OK;
}

int pv_pick_nll_bwd(const float* dout, const int64_t* target, int64_t ignore, int T, int V, float* dlogp,
                    void* stream) {
  PV_REQUIRE(dout && target && dlogp && T > 0 && V > 0, "pv_pick_nll_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  PV_HIP(hipMemsetAsync(dlogp, 0, (size_t)T * V * sizeof(float), st));
  hipLaunchKernelGGL(pick_nll_bwd_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, dout, target, ignore, T, V, dlogp);
  PV_LAUNCH_CHECK("pick_nll_bwd_kernel");
  return PV_OK;
}

int pv_colsum_f32(const float* X, int R, long ld, int n, float* out, int accumulate, void* stream) {
  PV_REQUIRE(X && out && R > 0 && n > 0 && ld >= n, "pv_colsum_f32: bad argument");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream, X, R, ld, n, out,
                     accumulate);
  PV_LAUNCH_CHECK("colsum_f32_kernel");
  return PV_OK;
}

int pv_nll_head_fwd(const float* sums2, int batch, float* out2, void* stream) {
  PV_REQUIRE(sums2 && out2 && batch > 0, "pv_nll_head_fwd: bad argument");
  hipLaunchKernelGGL(nll_head_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums2, 1.0f / (float)batch, out2);
  PV_LAUNCH_CHECK("nll_head_fwd_kernel");
  return PV_OK;
}

int pv_nll_head_bwd(const float* sums2, const float* out2, const float* g_nll, const float* g_ppl, int batch, float* dsums2,
                    void* stream) {
  PV_REQUIRE(sums2 && out2 && dsums2 && batch > 0, "pv_nll_head_bwd: bad argument");
  hipLaunchKernelGGL(nll_head_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums2, out2, g_nll, g_ppl,
                     1.0f / (float)batch, dsums2);
  PV_LAUNCH_CHECK("nll_head_bwd_kernel");
  return PV_OK;
}

int pv_sum_group(const pv_sum_group_desc* g, void* stream) {
  PV_REQUIRE(g && g->count > 0 && g->count <= PV_SUM_GROUP, "pv_sum_group: 1..%d tensors per launch", PV_SUM_GROUP);
  long nmax = 0;
  for (int i = 0; i < g->count; ++i) {
    PV_REQUIRE(g->dst[i] && g->src[i][0] && g->n[i] > 0 && g->n[i] % 4 == 0, "pv_sum_group: tensor %d: null pointer or length %ld not a multiple of 4", i, g->n[i]);
    PV_REQUIRE((((uintptr_t)g->dst[i] | (uintptr_t)g->src[i][0] | (uintptr_t)g->src[i][1] | (uintptr_t)g->src[i][2] | (uintptr_t)g->src[i][3]) & 15) == 0,
               "pv_sum_group: tensor %d is not 16-byte aligned", i);
    nmax = g->n[i] > nmax ? g->n[i] : nmax;
  }
  const int bx = (int)((nmax / 4 + 255) / 256);
  hipLaunchKernelGGL(sum_group_kernel, dim3(bx < 64 ? bx : 64, g->count), dim3(256), 0, (hipStream_t)stream, *g);
  PV_LAUNCH_CHECK("sum_group_kernel");
  return PV_OK;
}

}  // extern "C"
