// Internal (not part of the C ABI): several small fp32 GEMMs in one launch, used by the fused Block backward.
#pragma once
#include <hip/hip_runtime.h>

// C[m*ldc + n] (+)= alpha_dev[0] * (sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]  -  (sum r1_c[0..r1_nc)) * r1_u[m] * r1_v[n])
// Operands must admit 16-byte loads along M (sam == 1) and N (sbn == 1).
struct LinGroupItem {
  const float* A;
  long sam, sak;
  const float* B;
  long sbk, sbn;
  float* C;
  long ldc;
  int M, N, K;
  const float* alpha_dev;
  int accumulate;
  const float* r1_c;
  int r1_nc;
  const float* r1_u;
  const float* r1_v;
};

int pv_linear_f32_group(const LinGroupItem* items, int n, hipStream_t st);
