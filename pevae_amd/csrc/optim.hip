// Loss-assembly tail of the training step for gfx950: global-norm clip, the reference's skip-step guards and
// Adam with constant-with-warmup LR, all on device over ONE flat parameter buffer (no host sync per step).
//
// Reference: clip_grad_norm_(params, grad_clip, 2) models/main_pretrain.py:207; skip on inf/nan (<= 5 times
// in a row) :213-217; skip on a norm spike > 3x the running mean :219-222; running-mean update :225;
// torch.optim.Adam(lr=init_lr) + transformers.get_constant_schedule_with_warmup, dialog_models.py:122-131.
// The host never reads the norm: `guard_kernel` writes {skip, clip_coef, lr, bias corrections} into a small
// device state block that `adam_kernel` consumes, so DP ranks (which all see the same all-reduced gradient)
// decide identically and the step can live inside a HIP graph.
#include "common.h"
#include "../../include/pevae_hip.h"

namespace {

struct OptState {       // 16 x 4 bytes, lives in device memory (pv_optim_state_bytes())
  float avg_norm;       // running mean of the gradient norm (main_pretrain.py:210-225)
  int have_avg;
  int jump_cnt;         // consecutive non-finite skips
  int batch_cnt;        // applied steps (the reference's batch_cnt)
  int adam_step;        // Adam's t
  float last_norm;      // pre-clip total norm of the last call
  float clip_coef;
  int skip;             // 1 = this step's update is dropped
  float lr_eff;
  float bc1, bc2;       // 1 - beta^t
  int skipped_total;
  float lr_base;        // byte 48: the base learning rate when pv_grad_guard is called with lr < 0 (written by the host)
  float pad[3];
};

constexpr int NPART = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  const long n4 = n / 4;
  const float4* g4 = (const float4*)g;
  for (long i = blockIdx.x * 256l + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void guard_kernel(const float* __restrict__ partial, int npart, OptState* st,
                                                     float max_norm, float lr, int warmup, float beta1, float beta2,
                                                     int guards, float grad_scale) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < npart; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  // grad_scale: the buffer holds grad_scale^-1 times the gradient (a data-parallel SUM over `world` ranks with
  // grad_scale = 1/world): norm, clip and update all see the scaled gradient without a pass over the buffer
  const float norm = sqrtf(red[0]) * grad_scale;
  OptState o = *st;
  o.last_norm = norm;
  o.skip = 0;
  if (guards) {
    if (!o.have_avg) {
      o.avg_norm = norm;
      o.have_avg = 1;
    }
    const bool bad = isinf(norm) || isnan(norm);
    if (bad && o.jump_cnt < 5) {
      o.jump_cnt += 1;
      o.skip = 1;
    } else {
      o.jump_cnt = 0;
      if (o.avg_norm * 3.f < norm) {
        o.avg_norm = o.avg_norm / 20.f * 19.f + norm / 20.f;
        o.skip = 1;
      } else {
        o.avg_norm = o.avg_norm / (float)(o.batch_cnt + 1) * (float)o.batch_cnt + norm / (float)(o.batch_cnt + 1);
      }
    }
  }
  if (o.skip) {
    o.skipped_total += 1;
  } else {
    const float coef = max_norm / (norm + 1e-6f);          // torch clip_grad_norm_
    o.clip_coef = (max_norm > 0.f ? fminf(coef, 1.f) : 1.f) * grad_scale;
    // LambdaLR: the k-th optimizer.step() (k = 0,1,..) runs with lr * min(1, k / warmup)
    const int k = o.adam_step;
    // lr < 0: the base rate is a DEVICE scalar (lr_base), so that a replayed HIP graph follows the host's plateau decay
    const float lr_now = lr < 0.f ? o.lr_base : lr;
    o.lr_eff = (warmup > 0 && k < warmup) ? lr_now * (float)k / fmaxf(1.f, (float)warmup) : lr_now;
    o.adam_step = k + 1;
    o.batch_cnt += 1;
    o.bc1 = 1.f - powf(beta1, (float)o.adam_step);
    o.bc2 = 1.f - powf(beta2, (float)o.adam_step);
  }
  *st = o;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    const OptState* __restrict__ st, float beta1, float beta2,
                                                    float eps, float weight_decay) {
  if (st->skip) return;
  const float coef = st->clip_coef, step_size = st->lr_eff / st->bc1, rbc2 = rsqrtf(st->bc2);
  const float decay = 1.f - st->lr_eff * weight_decay;      // torch.optim.AdamW: p *= 1 - lr * wd before the update
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256l + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pv = ((float4*)p)[i], mv = ((float4*)m)[i], vv = ((float4*)v)[i];
    const float4 gv = ((const float4*)g)[i];
    float* pp = (float*)&pv;
    float* mm = (float*)&mv;
    float* vx = (float*)&vv;
    const float* gg = (const float*)&gv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gg[k] * coef;
      mm[k] = beta1 * mm[k] + (1.f - beta1) * gr;
      vx[k] = beta2 * vx[k] + (1.f - beta2) * gr * gr;
      pp[k] = pp[k] * decay - step_size * mm[k] / (sqrtf(vx[k]) * rbc2 + eps);
    }
    ((float4*)p)[i] = pv;
    ((float4*)m)[i] = mv;
    ((float4*)v)[i] = vv;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * coef;
      m[i] = beta1 * m[i] + (1.f - beta1) * gr;
      v[i] = beta2 * v[i] + (1.f - beta2) * gr * gr;
      p[i] = p[i] * decay - step_size * m[i] / (sqrtf(v[i]) * rbc2 + eps);
    }
}

}  // namespace

extern "C" {

size_t pv_optim_state_bytes(void) { return sizeof(OptState) + NPART * sizeof(float); }

int pv_grad_guard(const float* grads, long n, void* state, float max_norm, float lr, int warmup_steps, float beta1,
                  float beta2, int guards, float grad_scale, void* stream) {
  PV_REQUIRE(grads && state && n > 0, "pv_grad_guard: bad argument");
  PV_REQUIRE(((uintptr_t)grads & 15) == 0, "pv_grad_guard: flat gradient buffer must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)((char*)state + sizeof(OptState));
  long blocks = (n / 4 + 255) / 256;
  const int nb = (int)(blocks < 1 ? 1 : (blocks > NPART ? NPART : blocks));
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, grads, n, partial);
  PV_LAUNCH_CHECK("sumsq_kernel");
  hipLaunchKernelGGL(guard_kernel, dim3(1), dim3(256), 0, st, partial, nb, (OptState*)state, max_norm, lr,
                     warmup_steps, beta1, beta2, guards, grad_scale);
  PV_LAUNCH_CHECK("guard_kernel");
  return PV_OK;
}

int pv_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, const void* state,
                 float beta1, float beta2, float eps, float weight_decay, void* stream) {
  PV_REQUIRE(params && grads && exp_avg && exp_avg_sq && state && n > 0, "pv_adam_step: bad argument");
  PV_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
             "pv_adam_step: flat buffers must be 16-byte aligned");
  long blocks = (n / 4 + 255) / 256;
  const int nb = (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
  hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                     (const OptState*)state, beta1, beta2, eps, weight_decay);
  PV_LAUNCH_CHECK("adam_kernel");
  return PV_OK;
}

}  // extern "C"
