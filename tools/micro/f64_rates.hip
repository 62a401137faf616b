// developer tool: issue cost of the fp64 vector instructions the column kernels are made of (one wave on one SIMD,
// eight independent chains), and of a/b by three routes: the compiler's macro (v_div_scale x2, v_rcp, 5 fma,
// v_div_fmas, v_div_fixup), and the invariant-divisor form used by the kernels (pomgpu_internal.hpp: div_inv).
// Also checks div_inv against the hardware quotient on random operands (prints the number of mismatches).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o f64_rates f64_rates.hip && ./f64_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#define N 8
#define REP 256
template <int OP> __global__ void k_rate(double *io, long long *cyc, double a, double b) {
  double x[N];
  for (int n = 0; n < N; n++) x[n] = io[threadIdx.x * N + n];
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int n = 0; n < N; n++) {
      if (OP == 0) x[n] = __builtin_fma(x[n], a, b);
      if (OP == 1) x[n] = x[n] * a;
      if (OP == 2) x[n] = x[n] + a;
      if (OP == 3) x[n] = __builtin_amdgcn_rcp(x[n]);
      if (OP == 4) x[n] = __builtin_sqrt(x[n]);                 // full macro
      if (OP == 5) x[n] = x[n] / a;                             // full macro, invariant divisor
      if (OP == 6) { const double q = x[n] * b; const double e = __builtin_fma(-a, q, x[n]); x[n] = __builtin_fma(e, b, q); }   // 1 correction
      if (OP == 7) { double q = x[n] * b; double e = __builtin_fma(-a, q, x[n]); q = __builtin_fma(e, b, q); e = __builtin_fma(-a, q, x[n]); x[n] = __builtin_fma(e, b, q); }   // 2 corrections
      if (OP == 8) x[n] = __builtin_amdgcn_div_fixup(x[n], a, b);
      if (OP == 9) x[n] = __builtin_amdgcn_div_fmas(x[n], a, b, true);
      if (OP == 10) x[n] = __builtin_amdgcn_rsq(x[n]);
      if (OP == 11) x[n] = __builtin_fmax(x[n], a);
      if (OP == 12) x[n] = __builtin_fabs(x[n]) * a;
      if (OP == 13) x[n] = __builtin_amdgcn_sqrt(x[n]);         // v_sqrt_f64 alone
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  for (int n = 0; n < N; n++) io[threadIdx.x * N + n] = x[n];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// hardware quotient vs the invariant-divisor form: y = 1/b (IEEE), q = a*y, two fma corrections
__device__ __forceinline__ double div_inv2(double a, double b, double y) {      // = divi() of pomgpu_internal.hpp, the zero's sign included
  const double q0 = a * y;
  double e = __builtin_fma(-b, q0, a);
  double q = __builtin_fma(e, y, q0);
  e = __builtin_fma(-b, q, a);
  q = __builtin_fma(e, y, q);
  return a == 0. ? q0 : q;
}
__device__ __forceinline__ double div_inv1(double a, double b, double y) {
  const double q = a * y;
  const double e = __builtin_fma(-b, q, a);
  return __builtin_fma(e, y, q);
}
__device__ uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
__global__ void k_check(unsigned long long seed, int mode, unsigned long long *bad1, unsigned long long *bad2, int iters) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long b1 = 0, b2 = 0;
  for (int it = 0; it < iters; it++) {
    uint64_t r1 = mix(seed + g * 2654435761ull + (uint64_t)it * 0x100000001b3ull), r2 = mix(r1 ^ 0xabcdef12345ull);
    double a, b;
    if (mode == 0) {          // full random significands, exponents within +-40 of 1.0 (no overflow / underflow / denormals on the way)
      a = __longlong_as_double((long long)((r1 & 0x800fffffffffffffull) | ((uint64_t)(1023 - 40 + (r1 >> 52) % 81) << 52)));
      b = __longlong_as_double((long long)((r2 & 0x800fffffffffffffull) | ((uint64_t)(1023 - 40 + (r2 >> 52) % 81) << 52)));
    } else if (mode == 1) {   // divisors with few significant bits / near powers of two, numerators near the hard cases
      b = __longlong_as_double((long long)((r2 & 0x000ff00000000fffull) | (1023ull << 52)));
      a = __longlong_as_double((long long)((r1 & 0x000fffffffffffffull) | (1023ull << 52)));
    } else if (mode == 2) {   // significand of b all ones on top (worst case for the reciprocal), a random
      b = __longlong_as_double((long long)((0x000fffffffff0000ull | (r2 & 0xffff)) | (1023ull << 52)));
      a = __longlong_as_double((long long)((r1 & 0x000fffffffffffffull) | (1024ull << 52)));
    } else if (mode == 3) {   // zero numerators of both signs (masked fluxes), divisors of both signs: the QUOTIENT'S SIGN must survive
      b = __longlong_as_double((long long)((r2 & 0x800fffffffffffffull) | ((uint64_t)(1023 - 40 + (r2 >> 52) % 81) << 52)));
      a = (r1 & 1) ? -0.0 : 0.0;
    } else {                  // numerators whose residual a - b*q is subnormal (|a| in 2^-1074 .. 2^-960): documented as not covered, counted here
      b = __longlong_as_double((long long)((r2 & 0x000fffffffffffffull) | ((uint64_t)(1023 + (r2 >> 52) % 4) << 52)));
      a = __longlong_as_double((long long)((r1 & 0x800fffffffffffffull) | ((uint64_t)((r1 >> 52) % 64) << 52)));
    }
    const double y = 1.0 / b, q = a / b;
    if (__double_as_longlong(div_inv1(a, b, y)) != __double_as_longlong(q)) b1++;      // bitwise: -0.0 is not +0.0 here
    if (__double_as_longlong(div_inv2(a, b, y)) != __double_as_longlong(q)) b2++;
  }
  if (b1) atomicAdd(bad1, b1);
  if (b2) atomicAdd(bad2, b2);
}
int main() {
  double *io; long long *cyc; unsigned long long *bad;
  hipMalloc(&io, 64 * N * sizeof(double)); hipMalloc(&cyc, 8); hipMalloc(&bad, 16);
  double h[64 * N];
  const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "sqrt() macro", "a/b macro", "a/b inv, 1 corr (3 instr)", "a/b inv, 2 corr (5 instr)",
                         "v_div_fixup_f64", "v_div_fmas_f64", "v_rsq_f64", "v_max_f64", "abs+mul", "v_sqrt_f64"};
#define RUN(OP) { for (int n = 0; n < 64 * N; n++) h[n] = 1.0 + 0.001 * n; hipMemcpy(io, h, sizeof h, hipMemcpyHostToDevice); \
    k_rate<OP><<<1, 64>>>(io, cyc, 1.0000001, 0.9999999); k_rate<OP><<<1, 64>>>(io, cyc, 1.0000001, 0.9999999); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-28s %7.2f cycles per op per wave (%lld cycles / %d)\n", names[OP], (double)c / (REP * N), c, REP * N); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
  for (int mode = 0; mode < 5; mode++) {
    hipMemset(bad, 0, 16);
    k_check<<<4096, 256>>>(12345 + mode, mode, bad, bad + 1, 1024);
    unsigned long long hb[2]; hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
    printf("division check mode %d%s: %llu quotients compared bit for bit; mismatches with 1 correction: %llu, with 2 corrections: %llu\n", mode,
           mode == 3 ? " (signed zero numerators)" : mode == 4 ? " (numerators in the subnormal range: NOT covered by divi, documented)" : "", 4096ull * 256 * 1024, hb[0], hb[1]);
  }
  return 0;
}
