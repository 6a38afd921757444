// Probe (round 4): the XCD a workgroup runs on, read with s_getreg_b32 HW_REG_XCC_ID, against blockIdx % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  // SIMM16 = size-1 [15:11] | offset [10:6] | hwreg id [5:0]; HW_REG_XCC_ID = 20, bits 3:0 = XCC_ID
  const int v = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  k<<<4096, 64>>>(d);
  int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[16] = {0}, consistent = 0;
  for (int b = 0; b < 4096; b++) { hist[h[b] & 15]++; if ((h[b] & 7) == ((h[0] + b) & 7)) consistent++; }
  printf("first 16:"); for (int b = 0; b < 16; b++) printf(" %d", h[b]); printf("\nhist:"); for (int i = 0; i < 16; i++) printf(" %d", hist[i]);
  printf("\nblocks whose id follows blockIdx round-robin from block 0: %d of 4096\n", consistent);
  return 0;
}
