// developer micro-test: whole-wavefront shift by one lane with DPP (wave_shr:1 / wave_shl:1, GFX9)
// against __shfl_up / __shfl_down (ds_bpermute)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double dpp_up1(double x) {      // lane n <- lane n-1 (lane 0 keeps its own)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_dn1(double x) {      // lane n <- lane n+1 (lane 63 keeps its own)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double *o) {
  const int t = threadIdx.x;
  const double x = 1000. + t + 0.25 * blockIdx.x;
  o[(blockIdx.x * 64 + t) * 4 + 0] = dpp_up1(x);
  o[(blockIdx.x * 64 + t) * 4 + 1] = __shfl_up(x, 1, 64);
  o[(blockIdx.x * 64 + t) * 4 + 2] = dpp_dn1(x);
  o[(blockIdx.x * 64 + t) * 4 + 3] = __shfl_down(x, 1, 64);
}
int main() {
  double *d; hipMalloc(&d, 2 * 64 * 4 * 8);
  hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, d);
  double h[2 * 64 * 4]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int n = 0; n < 128; n++) { if (h[n * 4] != h[n * 4 + 1]) bad++; if (h[n * 4 + 2] != h[n * 4 + 3]) bad++; }
  printf("mismatches: %d   lane0 up: %g %g   lane63 down: %g %g   lane5: %g %g %g %g\n", bad, h[0], h[1], h[63 * 4 + 2], h[63 * 4 + 3], h[20], h[21], h[22], h[23]);
  return bad != 0;
}
