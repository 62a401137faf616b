// tests/emu/emu_support.cpp -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h in this directory).
#include "pomgpu_internal.hpp"
thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;

// check_velocity without the wavefront shuffles of k_reduce.hip (which is not emulated)
void launch_check_velocity(pomgpu_ctx *c) {
  const KP &P = c->P;
  double vamax = 0.;
  int imax = 0, jmax = 0;
  for (int j = 1; j <= P.jm; j++)
    for (int i = 1; i <= P.im; i++) {
      const double a = fabs(F2(vaf, i, j));
      if (a >= vamax) { vamax = a; imax = i; jmax = j; }
    }
  c->d_vel[0] = vamax; c->d_vel[1] = imax; c->d_vel[2] = jmax;
  if (vamax > P.vmaxl) *c->d_err = 1;
}

// domain_stats: the reduction kernels of k_reduce.hip are not emulated (GPU tests cover them)
void launch_domain_stats(pomgpu_ctx *, double *out_dev) {
  for (int q = 0; q < 7; q++) out_dev[q] = 0.;
}
