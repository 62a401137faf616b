// developer tool: does ONE wavefront saturate a SIMD's fp64 pipe?  The same loop of independent v_fma_f64 / v_mul_f64 chains
// (tools/micro/f64_rates.hip) with 1, 2 and 4 wavefronts per SIMD of one CU: cycles per instruction and wavefront.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o f64_occupancy f64_occupancy.hip && ./f64_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 8
#define REP 512
template <int OP> __global__ void k(double *io, long long *cyc, double a, double b) {
  double x[N];
  for (int n = 0; n < N; n++) x[n] = io[threadIdx.x * N + n];
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int n = 0; n < N; n++) {
      if (OP == 0) x[n] = __builtin_fma(x[n], a, b);
      if (OP == 1) x[n] = x[n] * a;
      if (OP == 2) { x[n] = __builtin_fma(x[n], a, b); asm volatile("v_mov_b32 %0, %0" : "+v"(io) ); }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  for (int n = 0; n < N; n++) io[threadIdx.x * N + n] = x[n];
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
int main() {
  double *io; long long *cyc;
  hipMalloc(&io, 1024 * N * sizeof(double)); hipMalloc(&cyc, 16 * sizeof(long long));
  hipMemset(io, 0, 1024 * N * sizeof(double));
  for (int op = 0; op < 2; op++)
    for (int threads = 256; threads <= 1024; threads *= 2) {
      if (op == 0) k<0><<<1, threads>>>(io, cyc, 1.0000001, 1e-9); else k<1><<<1, threads>>>(io, cyc, 1.0000001, 1e-9);
      hipDeviceSynchronize();
      long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      long long mx = 0; for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? h[w] : mx;
      printf("%s  %d wavefront(s) per SIMD: %.2f cycles per instruction and wavefront, %.2f per instruction on the SIMD\n", op ? "v_mul_f64" : "v_fma_f64",
             threads / 256, (double)mx / (REP * N), (double)mx / (REP * N) / (threads / 256));
    }
  return 0;
}
