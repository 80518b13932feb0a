// Keep decisions of the step's dropout sites (dropout.hip; also applied by kernels that fold a dropout backward into
// their own store pass): Philox4x32-10 keyed by (seed, step, call site, logical element index / 4), or a supplied mask.
#pragma once
#include "common.h"

// keep decisions of the 4 logical elements 4*q .. 4*q+3 of call site `site` at the state's current step
__device__ __forceinline__ void keep4(const RngState& st, uint32_t site, unsigned long long q, uint32_t thresh, bool (&keep)[4]) {
  uint32_t r[4];
  philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), site, (uint32_t)st.step ^ (uint32_t)(st.step >> 32) * 0x9E3779B9u,
                (uint32_t)st.seed, (uint32_t)(st.seed >> 32), r);
#pragma unroll
  for (int k = 0; k < 4; ++k) keep[k] = r[k] >= thresh;        // P(keep) = 1 - p
}

// v[0..3] <- dropout of the 4 logical elements midx .. midx+3 (mask when given, else the generator)
__device__ __forceinline__ void drop4_at(const uint8_t* mask, const RngState& st, uint32_t site, uint32_t thresh, float scale,
                                         long midx, float (&v)[4]) {
  if (mask) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = mask[midx + k] ? v[k] * scale : 0.f;
    return;
  }
  if ((midx & 3) == 0) {
    bool keep[4];
    keep4(st, site, (unsigned long long)midx >> 2, thresh, keep);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = keep[k] ? v[k] * scale : 0.f;
  } else {                                     // group straddles two counters
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bool keep[4];
      keep4(st, site, (unsigned long long)(midx + k) >> 2, thresh, keep);
      v[k] = keep[(midx + k) & 3] ? v[k] * scale : 0.f;
    }
  }
}

inline uint32_t drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return (uint32_t)(t >= 4294967295.0 ? 4294967295.0 : t);
}
