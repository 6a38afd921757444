#include <hip/hip_runtime.h>
__global__ void k(int* g, int* out) {
  __shared__ int s[64];
  g[threadIdx.x] = 5;
  s[threadIdx.x] = 1;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  out[threadIdx.x] = s[63 - threadIdx.x];
}
__global__ void k2(int* g, int* out) {
  __shared__ int s[64];
  g[threadIdx.x] = 5;
  s[threadIdx.x] = 1;
  __builtin_amdgcn_wave_barrier();
  out[threadIdx.x] = s[63 - threadIdx.x];
}
