// Probe (gfx950): what does buffer_load_dwordx4 ... lds write for lanes whose offset is outside the descriptor's range?
// The fused vocabulary kernels rely on zeros (rows past the matrix are staged without clamping the address).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_lds_dma_oob.hip -o tools/build/probe_lds_dma_oob ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const unsigned* src, unsigned bytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = 0x7fc0dead;      // NaN pattern
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // piece 0: lanes 0..31 in range, lanes 32..63 past the end; piece 1: every lane past the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds, 16, lane * 16 + (int)bytes - 512, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(lds + 256), 16, lane * 16 + (int)bytes + 4096, 0, 0, 0);
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  const unsigned n = 4096;
  std::vector<unsigned> h(n);
  for (unsigned i = 0; i < n; ++i) h[i] = 0x10000000u + i;
  unsigned *d, *o;
  hipMalloc(&d, n * 4 + 65536);                    // memory behind the descriptor's end exists and holds a marker
  hipMalloc(&o, 512 * 4);
  hipMemset(d, 0xab, n * 4 + 65536);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n * 4, o);
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  int ok_in = 0, zero_out = 0, other = 0;
  for (int i = 0; i < 128; ++i) ok_in += r[i] == 0x10000000u + (n - 128) + i;
  for (int i = 128; i < 512; ++i) { zero_out += r[i] == 0; other += (r[i] != 0); }
  printf("in-range words correct: %d/128; out-of-range words zero: %d/384, not zero: %d (first: %08x)\n", ok_in, zero_out, other, r[128]);
  return (ok_in == 128 && zero_out == 384) ? 0 : 1;
}
