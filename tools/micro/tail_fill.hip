// developer tool: do two kernels on two streams fill each other's LAST round of workgroups on gfx950?
// A 3-D kernel on a 194-row tile is ~3.3 rounds of CU-filling workgroups and takes the time of 4 (T = ceil(rounds) x
// round time fits 8-, 4-, 2-tile and full-grid launches of k_advct_col to 3 %).  If the idle CUs of a last round take another kernel's
// workgroups, two such kernels side by side on two streams cost ceil(6.6) = 7 round times instead of 8.
//   hipcc --offload-arch=gfx950 -O3 -o tail_fill tail_fill.hip && ./tail_fill
// The kernel: 512-thread workgroups that hold ~250 VGPRs (one workgroup per CU, like k_profq / the row-sharing kernels) and run a fixed
// chain of dependent FMAs (issue-bound, no memory): its duration per workgroup does not depend on what else runs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define NR 110
__global__ void __launch_bounds__(512) k_busy(double *out, int iters, double seed) {
  double r[NR];
#pragma unroll
  for (int n = 0; n < NR; n++) r[n] = seed + n + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int n = 0; n < NR; n++) r[n] = fma(r[n], 1.0000001, r[(n + 1) % NR] * 1e-9);
  }
  double s = 0.;
#pragma unroll
  for (int n = 0; n < NR; n++) s += r[n];
  if (s == 12345.678) out[blockIdx.x] = s;                   // never true: keeps the chain alive
}
static double ms_between(hipEvent_t a, hipEvent_t b) { float m; hipEventElapsedTime(&m, a, b); return m; }
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  double *out; hipMalloc(&out, 1 << 20);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e0, e1, ef; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ef, hipEventDisableTiming);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k_busy);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_busy, 512, 0);
  printf("%d CUs; k_busy: %d VGPRs, %d workgroup(s) of 512 threads per CU\n", ncu, fa.numRegs, occ);
  const int iters = 400;
  auto run = [&](int g1, int g2, int mode) {                  // mode 0: g1 alone; 1: g1 then g2 on one stream; 2: g1 on s1, g2 on s2
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s1);
      if (mode == 2) { hipEventRecord(ef, s1); hipStreamWaitEvent(s2, ef, 0); }
      hipLaunchKernelGGL(k_busy, dim3(g1), dim3(512), 0, s1, out, iters, 1.0);
      if (mode == 1) hipLaunchKernelGGL(k_busy, dim3(g2), dim3(512), 0, s1, out, iters, 2.0);
      if (mode == 2) { hipLaunchKernelGGL(k_busy, dim3(g2), dim3(512), 0, s2, out, iters, 2.0); hipEventRecord(ef, s2); hipStreamWaitEvent(s1, ef, 0); }
      hipEventRecord(e1, s1);
      hipEventSynchronize(e1);
      const double m = ms_between(e0, e1);
      if (m < best) best = m;
    }
    return best;
  };
  const double round = run(ncu * occ, 0, 0);
  printf("one full round (%d workgroups): %.3f ms\n", ncu * occ, round);
  printf("%-34s %10s %10s %10s %10s\n", "workgroups per kernel (rounds)", "alone", "2 in turn", "2 streams", "ideal 2");
  const double fr[] = {0.32, 1.32, 3.32, 3.9, 6.5};
  for (double f : fr) {
    const int g = (int)(f * ncu * occ);
    const double a = run(g, 0, 0), b = run(g, g, 1), c = run(g, g, 2);
    printf("%6d (%.2f)                      %10.3f %10.3f %10.3f %10.3f   in round times: %.2f %.2f %.2f\n", g, f, a, b, c, 2 * f * round, a / round, b / round, c / round);
  }
  return 0;
}
