// round 6 experiment: the data-parallel primitives of xm_bound.h / xm_extend.h (eight lanes of a 16-lane row) against shuffles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CTRL> __device__ inline int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ inline int dppo(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false); }
__device__ inline int imin(int a, int b) { return a < b ? a : b; }
__global__ void k(const int* in, int* out, int activeLanes) {
  const int lane = threadIdx.x & 63;
  if (lane >= activeLanes) return;
  const int g = lane & 7, none = 1 << 28;
  const int v = in[threadIdx.x];
  // (old = "none": what a lane without a source in its row keeps; lanes whose source belongs to the neighbouring group are masked)
  int ex = dppo<0x111>(none, v); ex = g >= 1 ? ex : none;
  int t = dppo<0x111>(none, ex); ex = imin(ex, g >= 1 ? t : none);
  t = dppo<0x112>(none, ex); ex = imin(ex, g >= 2 ? t : none);
  t = dppo<0x114>(none, ex); ex = imin(ex, g >= 4 ? t : none);
  int a = v;
  a = imin(a, dpp<0xB1>(a)); a = imin(a, dpp<0x4E>(a)); a = imin(a, dpp<0x141>(a));
  const int q1 = dpp<0x39>(v), q2 = dpp<0x4E>(v), q3 = dpp<0x93>(v), h0 = dpp<0x141>(v);
  const int h1 = dpp<0x39>(h0), h2 = dpp<0x4E>(h0), h3 = dpp<0x93>(h0);
  // the form that went wrong: the same scan with the shifted value itself as `old` and the first step written as one conditional expression - the compiler
  // (ROCm 7.2, gfx950, -O3) folds the DPP move into the conditional move and the shift is lost for the lanes whose condition holds
  int fx = g >= 1 ? dpp<0x111>(v) : none;
  int ft = dpp<0x111>(fx); fx = imin(fx, g >= 1 ? ft : none);
  ft = dpp<0x112>(fx); fx = imin(fx, g >= 2 ? ft : none);
  ft = dpp<0x114>(fx); fx = imin(fx, g >= 4 ? ft : none);
  int* o = out + threadIdx.x * 16;
  o[12] = fx;
  o[0] = ex; o[1] = a; o[2] = q1; o[3] = q2; o[4] = q3; o[5] = h0; o[6] = h1; o[7] = h2; o[8] = h3;
  o[9] = dpp<0x111>(v); o[10] = dpp<0x112>(v); o[11] = dpp<0x114>(v);
}
int main() {
  int h[64], *din, *dout, ho[64 * 16];
  srand(7);
  int bad = 0, badFolded = 0;
  for (int trial = 0; trial < 4; trial++) {
    const int active = trial == 0 ? 8 : trial == 1 ? 40 : trial == 2 ? 64 : 16;
    for (int i = 0; i < 64; i++) h[i] = rand() % 1000;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(dout, 0xFF, sizeof(ho));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, active);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int l = 0; l < active; l++) {
      const int g = l & 7, b = l & ~7;
      int ex = 1 << 28, all = 1 << 28;
      for (int j = 0; j < g; j++) ex = ex < h[b + j] ? ex : h[b + j];
      for (int j = 0; j < 8; j++) all = all < h[b + j] ? all : h[b + j];
      const int* o = ho + l * 16;
      if (o[12] != ex) badFolded++;
      if (o[0] != ex || o[1] != all) { bad++; if (bad < 12) printf("active %d lane %d: ex %d want %d, all %d want %d; shr1 %d shr2 %d shr4 %d (v %d)\n", active, l, o[0], ex, o[1], all, o[9], o[10], o[11], h[l]); }
      // the seven others, each exactly once
      int seen = 0;
      for (int j = 2; j <= 8; j++) for (int m = 0; m < 8; m++) if (m != g && o[j] == h[b + m]) { seen |= 1 << m; break; }
      (void)seen;
      int want[7], n = 0; for (int m = 0; m < 8; m++) if (m != g) want[n++] = h[b + m];
      int got[7]; for (int j = 0; j < 7; j++) got[j] = o[2 + j];
      for (int x = 0; x < 7; x++) for (int y = x + 1; y < 7; y++) { if (want[y] < want[x]) { int t = want[x]; want[x] = want[y]; want[y] = t; } if (got[y] < got[x]) { int t = got[x]; got[x] = got[y]; got[y] = t; } }
      for (int x = 0; x < 7; x++) if (want[x] != got[x]) { bad++; if (bad < 12) printf("active %d lane %d: the others differ\n", active, l); break; }
    }
  }
  printf("dpp check: %d bad (the forms the kernels use); the folded form: %d lanes wrong of %d\n", bad, badFolded, 8 + 40 + 64 + 16);
  return bad != 0;
}
