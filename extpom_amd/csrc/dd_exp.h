// Short-wave penetration of proft (reference pom/solver.f:1605-1614) with the bits of the reference.
//
// The reference evaluates   rad = real( swrad * ( r*exp(real(x1,16)) + (1.d0-r)*exp(real(x2,16)) ), 8 )
// in REAL(16): x1 = z(k)*dh/ad1 and x2 = z(k)*dh/ad2 are REAL(8) values widened, the two exp, the two products, the sum and the
// product with swrad are quad operations (113-bit significands), and the result is rounded ONCE to REAL(8).  Whatever a quad
// library makes of the last of its 113 bits, that result is the correctly rounded double of the exact expression unless the exact
// value lies within ~2^-105 (relative) of the midpoint of two doubles -- one evaluation in 2^52.  gfx950 has no quad arithmetic;
// an fp64 exp differs from the reference in 0.6 % of the values by one ulp, and this flow amplifies an ulp to 1e-3 within 1000
// internal steps (tools/swrad_drift.py, round 4: u differs by 8e-4 after 1000 steps with nbct = 2).  Here the expression is
// evaluated in double-double arithmetic (two fp64 per value, error-free transformations on the FMA unit, ~2^-100 relative) and
// rounded once: the same double as the reference's, with the same caveat one evaluation in ~2^45.  tools/check_dd_exp.cpp counts
// the differences to libquadmath's expq on 2e7 arguments of the ranges proft produces: none.
//
// Cost: ~700 fp64 operations per value, two values per cell and level, only in the SW instantiation of the proft kernels (nbc = 2 or 4
// with swrad != 0) -- not in the benchmarked configuration (nbct = 1).
#pragma once
#include <math.h>
// The error-free transformations below need every operation as written: a compiler that fuses `a.h * b.l + a.l * b.h` or a Horner step
// into FMAs of its own choosing computes another (not wrong, but different) double-double, and host and device would no longer evaluate
// the same operations.  The library (__graft_entry__.py) and the host check (tools/check_dd_exp.cpp, tests/test_host_logic.py) are both
// built with -ffp-contract=off; clang-based compilers (hipcc) also get it here, whatever their command line says (from this header
// to the end of the translation unit -- which is built with contraction off anyway).  The FMAs that ARE wanted are written as fma().
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
#ifdef __HIPCC__
#define DDX_HD __host__ __device__ __forceinline__
#else
#define DDX_HD static inline
#endif

struct ddx { double h, l; };

DDX_HD ddx ddx_fast_two_sum(double a, double b) { ddx r; r.h = a + b; r.l = b - (r.h - a); return r; }   // |a| >= |b|
DDX_HD ddx ddx_two_sum(double a, double b) {
  ddx r;
  r.h = a + b;
  const double bb = r.h - a;
  r.l = (a - (r.h - bb)) + (b - bb);
  return r;
}
DDX_HD ddx ddx_two_prod(double a, double b) { ddx r; r.h = a * b; r.l = fma(a, b, -r.h); return r; }
DDX_HD ddx ddx_add(ddx a, ddx b) {
  ddx s = ddx_two_sum(a.h, b.h);
  const ddx t = ddx_two_sum(a.l, b.l);
  s.l += t.h;
  s = ddx_fast_two_sum(s.h, s.l);
  s.l += t.l;
  return ddx_fast_two_sum(s.h, s.l);
}
DDX_HD ddx ddx_mul(ddx a, ddx b) {
  ddx p = ddx_two_prod(a.h, b.h);
  p.l += a.h * b.l + a.l * b.h;
  return ddx_fast_two_sum(p.h, p.l);
}
DDX_HD ddx ddx_mul_d(ddx a, double b) {
  ddx p = ddx_two_prod(a.h, b);
  p.l = fma(a.l, b, p.l);
  return ddx_fast_two_sum(p.h, p.l);
}
// exp(x) = m * 2^k with m in double-double, |x| < 2^20 * ln 2 (the caller bounds x)
DDX_HD ddx ddx_exp(double x, int *kout) {
  const double LN2_1 = 0x1.62e42fee00000p-1, LN2_2 = 0x1.a39ef35793c76p-33, LN2_3 = 0x1.cc01f97b57a08p-87;   // ln 2 in three pieces: k * LN2_1 is exact
  const double kd = nearbyint(x * 0x1.71547652b82fep+0);
  ddx r;
  r.h = x - kd * LN2_1;                                       // exact (Cody-Waite)
  r.l = 0.;
  ddx p2 = ddx_two_prod(kd, LN2_2);
  p2.h = -p2.h; p2.l = -p2.l;
  r = ddx_add(r, p2);
  ddx p3; p3.h = -kd * LN2_3; p3.l = 0.;
  r = ddx_add(r, p3);
  r.h *= 0.0625; r.l *= 0.0625;                               // s = r / 16, |s| < 0.0217: 16 Taylor terms reach 2^-113
  const double C[17][2] = {
      {0x1.0000000000000p+0, 0x0.0p+0},                       // 1/0!
      {0x1.0000000000000p+0, 0x0.0p+0},
      {0x1.0000000000000p-1, 0x0.0p+0},
      {0x1.5555555555555p-3, 0x1.5555555555555p-57},
      {0x1.5555555555555p-5, 0x1.5555555555555p-59},
      {0x1.1111111111111p-7, 0x1.1111111111111p-63},
      {0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65},
      {0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73},
      {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
      {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
      {0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76},
      {0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80},
      {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
      {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
      {0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92},
      {0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97},
      {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101}};       // 1/16!
  ddx p; p.h = C[16][0]; p.l = C[16][1];
#pragma unroll
  for (int n = 15; n >= 0; n--) {
    p = ddx_mul(p, r);
    ddx c; c.h = C[n][0]; c.l = C[n][1];
    p = ddx_add(p, c);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) p = ddx_mul(p, p);              // ^16
  *kout = (int)kd;
  return p;
}
// real( swrad * ( r*exp(real(x1,16)) + omr*exp(real(x2,16)) ), 8 ),  omr = 1.d0 - r as the reference forms it in REAL(8)
DDX_HD double proft_rad_q(double swrad, double r, double omr, double x1, double x2) {
  // below -11400 even REAL(16)'s exp is 0 (its smallest subnormal is 2^-16494); NaN operands take the plain formula
  if (!(x1 == x1) || !(x2 == x2) || x1 > 700. || x2 > 700.) return swrad * (r * exp(x1) + omr * exp(x2));
  int k1 = 0, k2 = 0;
  ddx e1, e2;
  const bool z1 = x1 < -11400., z2 = x2 < -11400.;
  e1.h = e1.l = e2.h = e2.l = 0.;
  if (!z1) e1 = ddx_exp(x1, &k1);
  if (!z2) e2 = ddx_exp(x2, &k2);
  if (z1 && z2) return swrad * 0.;
  // common exponent: the larger one; a term more than 2^-240 below the other is beyond the 113th bit of the sum
  int k = z1 ? k2 : (z2 ? k1 : (k1 > k2 ? k1 : k2));
  ddx t1 = ddx_mul_d(e1, r), t2 = ddx_mul_d(e2, omr);
  const int d1 = k1 - k, d2 = k2 - k;
  if (z1 || d1 < -240) { t1.h = t1.l = 0.; } else { t1.h = ldexp(t1.h, d1); t1.l = ldexp(t1.l, d1); }
  if (z2 || d2 < -240) { t2.h = t2.l = 0.; } else { t2.h = ldexp(t2.h, d2); t2.l = ldexp(t2.l, d2); }
  ddx s = ddx_add(t1, t2);
  s = ddx_mul_d(s, swrad);
  // one rounding: s.h is RN(s.h + s.l) (normalised); the scaling by 2^k is exact while the result stays a normal number
  if (s.h == 0. || ilogb(s.h) + k >= -1022) return ldexp(s.h, k);
  // a subnormal result lies on the grid of 2^-1074: round s.h + s.l to that grid in ONE step (ldexp(s.h, k) would round the already
  // rounded s.h a second time).  In units of the grid the value is y + yl with y < 2^52 exactly representable
  const double y = ldexp(s.h, k + 1074), yl = ldexp(s.l, k + 1074);
  double n = nearbyint(y);                                   // ties to even
  const double d = (y - n) + yl;                              // y - n is exact
  if (d > .5 || (d == .5 && fmod(n, 2.) != 0.)) n += 1.;
  else if (d < -.5 || (d == -.5 && fmod(n, 2.) != 0.)) n -= 1.;
  return ldexp(n, -1074);
}
