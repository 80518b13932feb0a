// Shared helpers for the pevae_amd HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_t;   // raw bf16 bits in memory

#define PV_OK 0
#define PV_ERR_ARG -1      // invalid argument (shape / alignment / null pointer)
#define PV_ERR_HIP -2      // a HIP runtime call or launch failed
#define PV_ERR_LIMIT -3    // size beyond a documented limit
#define PV_ERR_TIMEOUT -4  // a bounded in-kernel spin gave up (persistent GRU)

void pv_set_error(const char* fmt, ...);

#define PV_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      pv_set_error(__VA_ARGS__);              \
      return PV_ERR_ARG;                      \
    }                                         \
  } while (0)

#define PV_LAUNCH_CHECK(name)                                             \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      pv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return PV_ERR_HIP;                                                  \
    }                                                                     \
  } while (0)

#define PV_HIP(call)                                                      \
  do {                                                                    \
    hipError_t e__ = (call);                                              \
    if (e__ != hipSuccess) {                                              \
      pv_set_error("%s failed: %s", #call, hipGetErrorString(e__));       \
      return PV_ERR_HIP;                                                  \
    }                                                                     \
  } while (0)

__device__ __forceinline__ bf16_t f2bf(float x) {
  __bf16 b = (__bf16)x;   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
  return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); accurate to ~1e-7 abs, saturates cleanly
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f / (e + 1.0f);
}

// Counter-based generator shared by the dropout kernels (dropout.hip) and the sampling epilogue (gemm_bf16.hip).
struct RngState {        // 16 bytes of device memory
  unsigned long long seed;
  unsigned long long step;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
