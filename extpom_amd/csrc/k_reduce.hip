// k_reduce.hip -- check_velocity (advance.f:611-641): vamax = max |vaf| with the LAST position
// (j outer, i inner scan; `.ge.` at :623) at which the maximum is attained.
//
// A (value, linear index) pair is reduced with the total order "larger value wins, ties go to the
// larger scan index", which is associative and commutative, so the wavefront shuffle tree gives
// exactly the reference's sequential answer.  One 1024-thread workgroup strides over the plane:
// 16 wavefronts x __shfl_down tree (64 lanes) -> LDS -> first wavefront.
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
// out[0] = vamax, out[1] = imax, out[2] = jmax (as doubles); err: device error flag
__global__ void __launch_bounds__(1024) k_check_velocity(KP P, double *out, int *err) {
  __shared__ VelMax part[16];
  const long long total = (long long)P.im * P.jm;
  VelMax best;
  best.v = 0.;            // vamax starts at 0 and `>=` lets the last zero win
  best.n = -1;
  for (long long n = threadIdx.x; n < total; n += blockDim.x) {
    const int j = (int)(n / P.im) + 1, i = (int)(n % P.im) + 1;
    VelMax c;
    c.v = fabs(F2(vaf, i, j));
    c.n = n;
    if (c.v >= best.v) best = vm_best(best, c);
  }
  best = vm_wave(best);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = best;
  __syncthreads();
  if (wave == 0) {
    VelMax x = (lane < (int)(blockDim.x >> 6)) ? part[lane] : best;
    x = vm_wave(x);
    if (lane == 0) {
      out[0] = x.v;
      out[1] = (x.n >= 0) ? (double)(x.n % P.im + 1) : 0.;
      out[2] = (x.n >= 0) ? (double)(x.n / P.im + 1) : 0.;
      if (x.v > P.vmaxl) *err = 1;
    }
  }
}
void launch_check_velocity(pomgpu_ctx *c) {
  LAUNCH(c, k_check_velocity, dim3(1), dim3(1024), c->P, c->d_vel, c->d_err);
}
