// Dropout for the DLS_VAE step on gfx950: nn.Dropout of EncoderRNN / DecoderRNN (models/enc2dec/encoders.py:54,
// decoders.py:175,189) as our own kernels -- a counter-based generator (Philox4x32-10) keyed by (seed, step, call site,
// logical element index), so the backward pass REGENERATES the keep decisions instead of storing a mask, a replayed HIP
// graph draws fresh masks every step (the step counter lives in device memory and is advanced by a one-thread kernel), and
// a caller-supplied mask (parity tests: the same mask goes to the oracle) overrides the generator.
//
//   pv_dropout_f32        y = x * keep / (1 - p) on a strided [n0][n1][C] fp32 view (forward and backward are the same call)
//   pv_dropout_pack_bf16  the same, written as the zero-padded bf16 GEMM operand [n0*n1p][ldd] (decoder states -> the
//                         vocabulary projection's A operand in one pass instead of contiguous + dropout + cast)
// HBM-bound: 4 B read + 2 or 4 B written per element; 16-byte accesses when C % 4 == 0.
#include "common.h"
#include "dropout.h"
#include "../../include/pevae_hip.h"

namespace {

struct DropArgs {
  const float* src;
  int n0, n1, n1p, C;
  long s0, s1, sc;             // source strides (elements)
  long m0, m1;                 // logical-index strides of (i0, i1); the column stride is 1
  const uint8_t* mask;         // optional caller-supplied keep mask, indexed by the logical index
  const RngState* rng;         // used when mask == NULL
  uint32_t site, thresh;
  float scale;                 // 1 / (1 - p)
};

__device__ __forceinline__ void drop4(const DropArgs& a, long midx, const RngState& st, float (&v)[4]) {
  drop4_at(a.mask, st, a.site, a.thresh, a.scale, midx, v);
}

// one thread per 4 consecutive columns of one (i0, i1) row; BF16OUT: dst rows i0*n1p + i1 of `ldd` bf16, zero padded
template <bool BF16OUT>
__global__ __launch_bounds__(256) void dropout_kernel(DropArgs a, void* dst_, long d0, long d1, int ldd) {
  const int cg = BF16OUT ? ldd / 4 : (a.C + 3) / 4;              // column groups per row (bf16: including the padding)
  const long rows = (long)a.n0 * (BF16OUT ? a.n1p : a.n1);
  const long total = rows * cg;
  RngState st{0, 0};
  if (!a.mask) st = *a.rng;
  for (long i = blockIdx.x * 256l + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / cg;
    const int c = (int)(i % cg) * 4;
    const int i0 = (int)(row / (BF16OUT ? a.n1p : a.n1)), i1 = (int)(row % (BF16OUT ? a.n1p : a.n1));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool live = i1 < a.n1 && c < a.C;
    if (live) {
      const float* s = a.src + i0 * a.s0 + i1 * a.s1 + (long)c * a.sc;
      if (a.sc == 1 && c + 3 < a.C && (((uintptr_t)s) & 15) == 0) {
        const f32x4 x = *(const f32x4*)s;
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c + k < a.C ? s[(long)k * a.sc] : 0.f;
      }
      const long midx = i0 * a.m0 + i1 * a.m1 + c;
      if (a.mask && c + 3 >= a.C) {            // ragged tail with a mask: stay inside the mask's row
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (c + k < a.C && a.mask[midx + k]) ? v[k] * a.scale : 0.f;
      } else {
        drop4(a, midx, st, v);
      }
    }
    if constexpr (BF16OUT) {
      uint2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      *(uint2*)((bf16_t*)dst_ + row * ldd + c) = o;
    } else {
      float* d = (float*)dst_ + i0 * d0 + i1 * d1 + c;
      if (c + 3 < a.C && (((uintptr_t)d) & 15) == 0) {
        *(f32x4*)d = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k < a.C) d[k] = v[k];
      }
    }
  }
}

__global__ void rng_advance_kernel(RngState* st) { st->step += 1; }

int fill_args(DropArgs& a, const float* src, int n0, int n1, int n1p, int C, long s0, long s1, long sc, long m0, long m1,
              float p, const void* mask, const void* rng, int site, const char* who) {
  PV_REQUIRE(src && n0 > 0 && n1 > 0 && C > 0 && n1p >= n1, "%s: bad shape", who);
  PV_REQUIRE(p >= 0.f && p < 1.f, "%s: p=%f must be in [0, 1)", who, (double)p);
  PV_REQUIRE(mask || rng, "%s: need a keep mask or a generator state", who);
  a = DropArgs{src, n0, n1, n1p, C, s0, s1, sc, m0, m1, (const uint8_t*)mask, (const RngState*)rng, (uint32_t)site,
               drop_threshold(p), 1.f / (1.f - p)};
  return PV_OK;
}

int grid_for(long total) {
  const long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" {

size_t pv_rng_state_bytes(void) { return sizeof(RngState); }

int pv_rng_advance(void* state, void* stream) {
  PV_REQUIRE(state, "pv_rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (RngState*)state);
  PV_LAUNCH_CHECK("rng_advance_kernel");
  return PV_OK;
}

int pv_dropout_f32(const float* src, int n0, int n1, int C, long s0, long s1, long sc, long m0, long m1, float p,
                   const void* mask, const void* rng_state, int site, float* dst, long d0, long d1, void* stream) {
  DropArgs a;
  int rc = fill_args(a, src, n0, n1, n1, C, s0, s1, sc, m0, m1, p, mask, rng_state, site, "pv_dropout_f32");
  if (rc) return rc;
  PV_REQUIRE(dst, "pv_dropout_f32: null output");
  const long total = (long)n0 * n1 * ((C + 3) / 4);
  hipLaunchKernelGGL(dropout_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, (void*)dst, d0, d1, 0);
  PV_LAUNCH_CHECK("dropout_kernel<f32>");
  return PV_OK;
}

int pv_dropout_pack_bf16(const float* src, int n0, int n1, int n1p, int C, long s0, long s1, long sc, long m0, long m1,
                         float p, const void* mask, const void* rng_state, int site, void* dst, int ldd, void* stream) {
  DropArgs a;
  int rc = fill_args(a, src, n0, n1, n1p, C, s0, s1, sc, m0, m1, p, mask, rng_state, site, "pv_dropout_pack_bf16");
  if (rc) return rc;
  PV_REQUIRE(dst && ldd >= C && ldd % 4 == 0 && ((uintptr_t)dst & 7) == 0,
             "pv_dropout_pack_bf16: ldd=%d must be a multiple of 4 and >= C=%d, dst 8-byte aligned", ldd, C);
  const long total = (long)n0 * n1p * (ldd / 4);
  hipLaunchKernelGGL(dropout_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, dst, 0l, 0l, ldd);
  PV_LAUNCH_CHECK("dropout_kernel<bf16>");
  return PV_OK;
}

}  // extern "C"
