// Probe of ds_read_b64_tr_b16 (gfx950): which (row, col) of a 4 x 16 block of 16-bit elements lands in which lane / element.
// hipcc --offload-arch=gfx950 tools/tr_probe.hip -o tools/_build/tr_probe && tools/_build/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;      // value = row * 64 + col
  __syncthreads();
  int l = threadIdx.x & 15, q = l >> 2, p = l & 3, g = threadIdx.x >> 4;
  // group g reads the block rows 4g .. 4g+3, cols 16 .. 31
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + (4 * g + q) * 64 + 16 + 4 * p));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t = 0; t < 64; ++t) { printf("lane %2d:", t); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[t * 4 + e] / 64, h[t * 4 + e] % 64); printf("\n"); }
  return 0;
}
