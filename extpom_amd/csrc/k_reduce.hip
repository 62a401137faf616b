// k_reduce.hip -- check_velocity (advance.f:611-641): vamax = max |vaf| with the LAST position
// (j outer, i inner scan; `.ge.` at :623) at which the maximum is attained.
//
// A (value, linear index) pair is reduced with the total order "larger value wins, ties go to the
// larger scan index", which is associative and commutative, so the wavefront shuffle tree gives
// exactly the reference's sequential answer.  Two stages: up to 1024 workgroups each reduce a strided
// share (4 wavefronts x __shfl_down tree -> LDS -> first wavefront), one workgroup reduces the partials.
#include "pomgpu_internal.hpp"

struct VelMax { double v; long long n; };
__device__ __forceinline__ VelMax vm_best(VelMax a, VelMax b) {
  return (b.v > a.v || (b.v == a.v && b.n > a.n)) ? b : a;
}
__device__ __forceinline__ VelMax vm_wave(VelMax x) {
  for (int off = 32; off > 0; off >>= 1) {
    VelMax y;
    y.v = __shfl_down(x.v, off, 64);
    y.n = __shfl_down(x.n, off, 64);
    x = vm_best(x, y);
  }
  return x;
}
__device__ __forceinline__ VelMax vm_block(VelMax best, VelMax *part) {   // all threads call; valid in wave 0
  best = vm_wave(best);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = best;
  __syncthreads();
  VelMax x = (lane < (int)(blockDim.x >> 6)) ? part[lane] : best;
  return vm_wave(x);
}
// stage 1: every workgroup scans a strided share of the plane; partial (value, index) -> pv, pn
__global__ void __launch_bounds__(256) k_check_velocity(KP P, double *pv, double *pn) {
  __shared__ VelMax part[4];
  const long long total = (long long)P.im * P.jm;
  VelMax best;
  best.v = 0.;            // vamax starts at 0 and `>=` lets the last zero win
  best.n = -1;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(n / P.im) + 1, i = (int)(n % P.im) + 1;
    VelMax c;
    c.v = fabs(F2(vaf, i, j));
    c.n = n;
    if (c.v >= best.v) best = vm_best(best, c);
  }
  best = vm_block(best, part);
  if (threadIdx.x == 0) { pv[blockIdx.x] = best.v; pn[blockIdx.x] = (double)best.n; }
}
// stage 2: out[0] = vamax, out[1] = imax, out[2] = jmax (as doubles); err: device error flag
__global__ void __launch_bounds__(256) k_check_velocity_fin(KP P, const double *pv, const double *pn, int nb, double *out, int *err) {
  __shared__ VelMax part[4];
  VelMax best;
  best.v = 0.;
  best.n = -1;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    VelMax c;
    c.v = pv[b];
    c.n = (long long)pn[b];
    if (c.v >= best.v) best = vm_best(best, c);
  }
  best = vm_block(best, part);
  if (threadIdx.x == 0) {
    out[0] = best.v;
    out[1] = (best.n >= 0) ? (double)(best.n % P.im + 1) : 0.;
    out[2] = (best.n >= 0) ? (double)(best.n / P.im + 1) : 0.;
    if (best.v > P.vmaxl) *err = 1;
  }
}
void launch_check_velocity(pomgpu_ctx *c) {
  const long long total = (long long)c->P.im * c->P.jm;
  int nb = (int)((total + 256 * 8 - 1) / (256 * 8));
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  LAUNCH(c, k_check_velocity, dim3(nb), dim3(256), c->P, c->P.s2[5], c->P.s2[6]);
  LAUNCH(c, k_check_velocity_fin, dim3(1), dim3(256), c->P, (const double *)c->P.s2[5], (const double *)c->P.s2[6], nb, c->d_vel, c->d_err);
}
