// developer check (CPU, gcc + libquadmath): proft_rad_q (extpom_amd/csrc/dd_exp.h, double-double) against the expression as the
// reference evaluates it in REAL(16) (solver.f:1608-1611; the oracle's restatement, oracle/pom_oracle.c:806-812, uses the same
// libquadmath expq) on arguments of the ranges proft produces.  Prints the number of results that differ.
//   g++ -O2 -ffp-contract=off -o /tmp/check_dd_exp tools/check_dd_exp.cpp -lquadmath && /tmp/check_dd_exp [n]
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include "../extpom_amd/csrc/dd_exp.h"
static uint64_t s = 0x9E3779B97F4A7C15ull;
static double u01() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0); }
int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 20000000;
  const double R[5] = {.58, .62, .67, .77, .78}, AD1[5] = {.35, .60, 1.0, 1.5, 1.4}, AD2[5] = {23., 20., 17., 14., 7.9};   // Jerlov types (initialize.f)
  long bad = 0, worst_ulp = 0;
  for (long t = 0; t < n; t++) {
    const int ntp = (int)(u01() * 5) % 5;
    const double z = -u01(), dh = 5. + u01() * (t % 3 == 0 ? 60. : 6000.), swrad = -(1e-6 + u01() * 3e-4);
    const double r = R[ntp], omr = 1. - r;
    const double x1 = z * dh / AD1[ntp], x2 = z * dh / AD2[ntp];
    const __float128 q = (__float128)swrad * ((__float128)r * expq((__float128)x1) + (__float128)omr * expq((__float128)x2));
    const double ref = (double)q, got = proft_rad_q(swrad, r, omr, x1, x2);
    if (memcmp(&ref, &got, 8)) {
      bad++;
      int64_t a, b; memcpy(&a, &ref, 8); memcpy(&b, &got, 8);
      long d = labs((long)(a - b)); if (d > worst_ulp) worst_ulp = d;
      if (bad <= 5) printf("differs: x1 %.17g x2 %.17g swrad %.17g  ref %.17g got %.17g\n", x1, x2, swrad, ref, got);
    }
  }
  printf("%ld arguments, %ld results differ from the REAL(16) evaluation (largest distance %ld ulp)\n", n, bad, worst_ulp);
  return bad != 0;
}
