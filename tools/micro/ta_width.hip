// developer micro-benchmark: is a 9-point stencil over several 2-D arrays bound by the number of
// wavefront load instructions (L1/TA) rather than bytes?  A: one cell per lane, 8-byte loads, every
// neighbour a separate load.  B: two cells per lane, 16-byte loads for the rows, i+-1 by lane shuffle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NA 6
struct Arrs { const double *a[NA]; };
__global__ void kA(Arrs A, double *out, int im, int jm) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (i < 1 || i >= im - 1 || j < 1 || j >= jm - 1) return;
  double s = 0.;
#pragma unroll
  for (int n = 0; n < NA; n++) {
    const double *p = A.a[n] + (size_t)j * im + i;
    s += p[-im - 1] + p[-im] + p[-im + 1] + p[-1] + p[0] + p[1] + p[im - 1] + p[im] + p[im + 1];
  }
  out[(size_t)j * im + i] = s;
}
__global__ void kB(Arrs A, double *out, int im, int jm) {
  const int lane = threadIdx.x;
  const int c0 = 2 * ((int)blockIdx.x * 62 + lane - 1);          // first of this lane's two columns; lanes 0, 63 are halo lanes
  const int j = blockIdx.y * 4 + threadIdx.y;
  if (j < 1 || j >= jm - 1) return;
  const int c = c0 < 0 ? 0 : (c0 > im - 2 ? im - 2 : c0);
  double s0 = 0., s1 = 0.;
#pragma unroll
  for (int n = 0; n < NA; n++) {
    const double *p = A.a[n] + (size_t)j * im + c;
    const double2 r0 = *(const double2 *)(p - im), r1 = *(const double2 *)p, r2 = *(const double2 *)(p + im);
    const double cs0 = r0.x + r1.x + r2.x, cs1 = r0.y + r1.y + r2.y;    // column sums
    const double w = __shfl_up(cs1, 1, 64), e = __shfl_down(cs0, 1, 64);
    s0 += w + cs0 + cs1;
    s1 += cs0 + cs1 + e;
  }
  if (lane >= 1 && lane <= 62 && c0 >= 0 && c0 + 1 < im) {
    if (c0 >= 1 && c0 < im - 1) out[(size_t)j * im + c0] = s0;
    if (c0 + 1 < im - 1) out[(size_t)j * im + c0 + 1] = s1;
  }
}
int main() {
  const int im = 2048, jm = 1536; const size_t n = (size_t)im * jm;
  Arrs A; double *out;
  std::vector<double> h(n);
  for (size_t k = 0; k < n; k++) h[k] = (double)(k % 1000) * 1e-3;
  for (int a = 0; a < NA; a++) { double *d; hipMalloc(&d, n * 8); hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice); A.a[a] = d; }
  hipMalloc(&out, n * 8); hipMemset(out, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(kA, dim3(im / 64, jm / 4), dim3(64, 4), 0, 0, A, out, im, jm);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("A (1 cell/lane, 54 x 8B loads): %.1f us/launch\n", ms / 20 * 1e3);
    std::vector<double> ra(n); hipMemcpy(ra.data(), out, n * 8, hipMemcpyDeviceToHost); hipMemset(out, 0, n * 8);
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(kB, dim3((im / 2 + 61) / 62, jm / 4), dim3(64, 4), 0, 0, A, out, im, jm);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("B (2 cells/lane, 18 x 16B loads): %.1f us/launch\n", ms / 20 * 1e3);
    std::vector<double> rb(n); hipMemcpy(rb.data(), out, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t k = 0; k < n; k++) if (fabs(ra[k] - rb[k]) > 1e-9 * (1 + fabs(ra[k]))) bad++;
    printf("mismatching cells: %zu\n", bad);
  }
  return 0;
}
