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
    if (best.v > P.vmaxl) atomicOr(err, POMGPU_DERR_VELOCITY);
  }
}
// ---------------------------------------------------------------------------------------------
// domain_stats (advance.f:644-756): seven masked sums -- vtot, atot, mtot, stot, sum(tb*dvol), sum(et*darea),
// ekin -- as a fixed-shape reduction: a thread sums its water columns level by level, a wavefront
// shuffle tree and the 4 wavefronts of a workgroup follow, one workgroup adds the partials in index
// order.  Same launch, same bits (no atomics); against the reference's SUM intrinsic parity is to
// rounding (its order of additions is the compiler's).  dvol is assigned on 2:imm1 x 2:jmm1 only and
// zero elsewhere (:692-695), so the 3-D sums -- edge additions included -- cover the interior; atot and
// the elevation sum include the physical edges without corners (:666-677).
#define NSTAT 7
__device__ __forceinline__ double ds_wave(double x) {
  for (int off = 32; off > 0; off >>= 1) x = x + __shfl_down(x, off, 64);
  return x;
}
__device__ __forceinline__ void ds_block(double (&a)[NSTAT], double (*part)[NSTAT]) {   // result in thread 0
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NSTAT; q++) a[q] = ds_wave(a[q]);
  if (lane == 0)
    for (int q = 0; q < NSTAT; q++) part[wave][q] = a[q];
  __syncthreads();
  if (threadIdx.x == 0)
    for (int q = 0; q < NSTAT; q++) a[q] = part[0][q] + part[1][q] + part[2][q] + part[3][q];
}
__global__ void __launch_bounds__(256) k_domain_stats(KP P, double *partial) {
  __shared__ double part[4][NSTAT];
  double a[NSTAT] = {0., 0., 0., 0., 0., 0., 0.};           // vtot atot mtot stot tsum esum ekin
  const long long total = (long long)P.im * P.jm;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(n / P.im) + 1, i = (int)(n % P.im) + 1;
    const bool ii = (i >= 2 && i <= P.imm1), jj = (j >= 2 && j <= P.jmm1);
    const bool edge = (jj && ((P.W && i == 1) || (P.E && i == P.im))) || (ii && ((P.S && j == 1) || (P.N && j == P.jm)));
    if (!(ii && jj) && !edge) continue;
    const double darea = F2(dx, i, j) * F2(dy, i, j) * F2(fsm, i, j);
    a[1] = a[1] + darea;
    a[5] = a[5] + F2(et, i, j) * darea;
    if (ii && jj) {
      const double dtc = F2(dt, i, j);
      for (int k = 1; k <= P.kbm1; k++) {
        const double dvol = darea * dtc * F1(dz, k);
        const double dmass = dvol * (F3(rho, i, j, k) * P.rhoref + 1000.);
        a[0] = a[0] + dvol;
        a[2] = a[2] + dmass;
        a[3] = a[3] + F3(sb, i, j, k) * dvol;
        a[4] = a[4] + F3(tb, i, j, k) * dvol;
        const double uu = F3(u, i, j, k), vv = F3(v, i, j, k);
        a[6] = a[6] + .5 * (dmass * (uu * uu + vv * vv));
      }
    }
  }
  ds_block(a, part);
  if (threadIdx.x == 0)
    for (int q = 0; q < NSTAT; q++) partial[(size_t)blockIdx.x * NSTAT + q] = a[q];
}
__global__ void __launch_bounds__(256) k_domain_stats_fin(const double *partial, int nb, double *out) {
  __shared__ double part[4][NSTAT];
  double a[NSTAT] = {0., 0., 0., 0., 0., 0., 0.};
  for (int b = threadIdx.x; b < nb; b += blockDim.x)
    for (int q = 0; q < NSTAT; q++) a[q] = a[q] + partial[(size_t)b * NSTAT + q];
  ds_block(a, part);
  if (threadIdx.x == 0)
    for (int q = 0; q < NSTAT; q++) out[q] = a[q];
}
void launch_domain_stats(pomgpu_ctx *c, double *out_dev) {
  const long long total = (long long)c->P.im * c->P.jm;
  int nb = (int)((total + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  LAUNCH(c, k_domain_stats, dim3(nb), dim3(256), c->P, c->P.s2[7]);          // 2048 x 7 partials fit one 2-D scratch plane (>= 25 cells checked at create)
  LAUNCH(c, k_domain_stats_fin, dim3(1), dim3(256), (const double *)c->P.s2[7], nb, out_dev);
}
void launch_check_velocity(pomgpu_ctx *c) {
  const long long total = (long long)c->P.im * c->P.jm;
  int nb = (int)((total + 256 * 8 - 1) / (256 * 8));
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  LAUNCH(c, k_check_velocity, dim3(nb), dim3(256), c->P, c->P.s2[5], c->P.s2[6]);
  LAUNCH(c, k_check_velocity_fin, dim3(1), dim3(256), c->P, (const double *)c->P.s2[5], (const double *)c->P.s2[6], nb, c->d_vel, c->d_err);
}
