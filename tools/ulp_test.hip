// developer tool: are fp64 sqrt and divide on gfx950 (as hipcc lowers them at -O3 -ffp-contract=off)
// bit-identical to the host's IEEE results?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const double *a, const double *b, double *s, double *q, double *e, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  s[t] = sqrt(a[t]);
  q[t] = a[t] / b[t];
  e[t] = exp(-a[t]);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> a(n), b(n), s(n), q(n), e(n);
  srand48(7);
  for (int t = 0; t < n; t++) { a[t] = drand48() * pow(10., (t % 40) - 20); b[t] = (drand48() + 1e-3) * pow(10., (t % 23) - 11); }
  double *da, *db, *ds, *dq, *de;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&de, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, ds, dq, de, n);
  hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost);
  long bs = 0, bq = 0, be = 0;
  for (int t = 0; t < n; t++) { if (s[t] != sqrt(a[t])) bs++; if (q[t] != a[t] / b[t]) bq++; if (e[t] != exp(-a[t])) be++; }
  printf("of %d: sqrt mismatches %ld, divide mismatches %ld, exp mismatches %ld\n", n, bs, bq, be);
  return 0;
}
