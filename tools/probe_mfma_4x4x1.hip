// Probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 blocks, k = 1).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_4x4x1.hip -o tools/build/probe_mfma_4x4x1
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  // A value encodes its lane as 1 + l, B value as 100 + l: D = A * B identifies which lanes met
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), (float)(100 + l), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[r * 64 + l] = acc[r];
}
int main() {
  float* d; float h[256];
  hipMalloc(&d, sizeof(h));
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int r = 0; r < 4; ++r)
    for (int l = 0; l < 64; ++l) {
      // hypothesis: D[vgpr r][lane l] = A[lane 4*(l/4) + r] * B[lane l]
      const float want = (float)(1 + 4 * (l / 4) + r) * (float)(100 + l);
      if (h[r * 64 + l] != want) { ok = 0; if (l < 8) printf("r%d l%d got %g want %g\n", r, l, h[r * 64 + l], want); }
    }
  printf("hypothesis D[r][l] = A[4*(l/4)+r] * B[l]: %s\n", ok ? "HOLDS" : "FAILS");
  for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l], h[64 + l], h[128 + l], h[192 + l]);
  return 0;
}
