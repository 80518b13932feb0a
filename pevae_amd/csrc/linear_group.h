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

// Optional rider of a grouped launch: column sums of per-workgroup partial rows, dst_seg[c - seg_begin] (+)= sum_g
// part[g][c] in row order (the bias / LayerNorm-parameter gradients of a fused Block), done by extra workgroups of the
// same launch instead of a launch of their own.
struct ColReduceJob {
  const float* part;     // [nrows][width]
  int nrows, width;
  int seg_end[7];        // exclusive ends of the 7 destination segments inside a row
  float* dst[7];
  int accumulate;
};
constexpr int COLREDUCE_MAX = 3;

int pv_linear_f32_group(const LinGroupItem* items, int n, hipStream_t st, const ColReduceJob* jobs = nullptr, int njobs = 0);
