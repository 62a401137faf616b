// k_ext.hip -- the external (2-D barotropic) mode: advave, mode_external, the 2-D parts of
// mode_interaction.  Every kernel is one thread per water column, threadIdx.x along i (coalesced
// 512-byte row segments per wavefront); each of them is a handful of 2-D array passes, i.e.
// HBM/L2-bound at ~10 flop per 8-byte word.
//
// Fusion rule used throughout: all statements of the reference between two halo-exchange points
// become ONE kernel; neighbour values of intermediates that the reference stored in scratch arrays
// (fluxua, fluxva) are recomputed in registers where that needs no data from beyond the tile.
#include <type_traits>
#include "pomgpu_internal.hpp"

// ua, va, d, el, elb are read through KP.x2 and written through KP.y2 in this file: the fused external
// step (k_ext_step) reads one generation and writes the next into a second set of buffers, the host
// swaps the two sets after every substep (pomgpu_api.hip, ext_parity).  Everywhere else x2 == y2 ==
// the blk2d arrays and these are plain in-place accessors.
#define d_(i, j) P.x2[X2_d][IX2(i, j)]
#define ua_(i, j) P.x2[X2_ua][IX2(i, j)]
#define va_(i, j) P.x2[X2_va][IX2(i, j)]
#define el_(i, j) P.x2[X2_el][IX2(i, j)]
#define elb_(i, j) P.x2[X2_elb][IX2(i, j)]
#define Y2(name, i, j) P.y2[X2_##name][IX2(i, j)]
#define dx_(i, j) F2(dx, i, j)
#define dy_(i, j) F2(dy, i, j)
#define uab_(i, j) P.x2[X2_uab][IX2(i, j)]
#define vab_(i, j) P.x2[X2_vab][IX2(i, j)]
#define aam2d_(i, j) F2(aam2d, i, j)

// ---------------------------------------------------------------------------------------------
// advave, u half: fluxua, fluxva (to scratch s2[0], s2[1]) and tps   -- solver.f:16-58
__global__ void k_advave_a(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  double fu = 0., fv = 0.;
  if (j >= 2 && j <= P.jm && i >= 2 && i <= P.im) {
    if (i <= P.imm1) {
      fu = .125 * ((d_(i + 1, j) + d_(i, j)) * ua_(i + 1, j) + (d_(i, j) + d_(i - 1, j)) * ua_(i, j)) *
           (ua_(i + 1, j) + ua_(i, j));                                                   // :20-26
      fu = fu - d_(i, j) * 2. * aam2d_(i, j) * (uab_(i + 1, j) - uab_(i, j)) / dx_(i, j); // :37-43
    }
    fv = .125 * ((d_(i, j) + d_(i, j - 1)) * va_(i, j) + (d_(i - 1, j) + d_(i - 1, j - 1)) * va_(i - 1, j)) *
         (ua_(i, j) + ua_(i, j - 1));                                                     // :28-34
    const double t = .25 * (d_(i, j) + d_(i - 1, j) + d_(i, j - 1) + d_(i - 1, j - 1)) *
                     (aam2d_(i, j) + aam2d_(i, j - 1) + aam2d_(i - 1, j) + aam2d_(i - 1, j - 1)) *
                     ((uab_(i, j) - uab_(i, j - 1)) / (dy_(i, j) + dy_(i - 1, j) + dy_(i, j - 1) + dy_(i - 1, j - 1)) +
                      (vab_(i, j) - vab_(i - 1, j)) / (dx_(i, j) + dx_(i - 1, j) + dx_(i, j - 1) + dx_(i - 1, j - 1)));
    F2(tps, i, j) = t;                                                                    // :47-53
    fu = fu * dy_(i, j);                                                                  // :54
    fv = (fv - t) * .25 * (dx_(i, j) + dx_(i - 1, j) + dx_(i, j - 1) + dx_(i - 1, j - 1)); // :55-56
  }
  G2(P.s2[0], i, j) = fu;
  G2(P.s2[1], i, j) = fv;
}

// advave: advua from the exchanged u-half fluxes, then the v-half fluxes -- solver.f:63-109
__global__ void k_advave_b(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const double *fu1 = P.s2[0], *fv1 = P.s2[1];
  double adv = 0.;
  if (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1)
    adv = G2(fu1, i, j) - G2(fu1, i - 1, j) + G2(fv1, i, j + 1) - G2(fv1, i, j);          // :63-68
  F2(advua, i, j) = adv;
  double fu = 0., fv = 0.;
  if (j >= 2 && j <= P.jm && i >= 2 && i <= P.im) {
    fu = .125 * ((d_(i, j) + d_(i - 1, j)) * ua_(i, j) + (d_(i, j - 1) + d_(i - 1, j - 1)) * ua_(i, j - 1)) *
         (va_(i - 1, j) + va_(i, j));                                                     // :78-84
    if (j <= P.jmm1) {
      fv = .125 * ((d_(i, j + 1) + d_(i, j)) * va_(i, j + 1) + (d_(i, j) + d_(i, j - 1)) * va_(i, j)) *
           (va_(i, j + 1) + va_(i, j));                                                   // :86-92
      fv = fv - d_(i, j) * 2. * aam2d_(i, j) * (vab_(i, j + 1) - vab_(i, j)) / dy_(i, j); // :95-101
    }
    fv = fv * dx_(i, j);                                                                  // :105
    fu = (fu - F2(tps, i, j)) * .25 * (dy_(i, j) + dy_(i - 1, j) + dy_(i, j - 1) + dy_(i - 1, j - 1)); // :106-107
  }
  F2(fluxua, i, j) = fu;
  F2(fluxva, i, j) = fv;
}

// advave: advva -- solver.f:114-119
__global__ void k_advave_c(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  double adv = 0.;
  if (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1)
    adv = F2(fluxua, i + 1, j) - F2(fluxua, i, j) + F2(fluxva, i, j) - F2(fluxva, i, j - 1);
  F2(advva, i, j) = adv;
}

// ---------------------------------------------------------------------------------------------
// advave, single tile, mode != 2: advua and advva in ONE pass -- solver.f:16-121.
// With all neighbours -1 the six exchanges inside advave are no-ops, so the flux arrays need not
// exist: every flux is a function of (i,j) evaluated where it is needed, in exactly the index ranges
// in which the reference writes its zero-initialised arrays; the (i-1) / (i+1) instances come from
// the neighbour lane (halo-lane wavefronts, pomgpu_internal.hpp).  fluxua, fluxva, tps (pure scratch in the reference) are not materialised.
// 8 reads + 2 writes instead of the 31 array passes of the three kernels above.
__device__ __forceinline__ double advave_tps(const KP &P, int i, int j) {                   // :47-53; 2<=i<=im, 2<=j<=jm
  return .25 * (d_(i, j) + d_(i - 1, j) + d_(i, j - 1) + d_(i - 1, j - 1)) *
         (aam2d_(i, j) + aam2d_(i, j - 1) + aam2d_(i - 1, j) + aam2d_(i - 1, j - 1)) *
         ((uab_(i, j) - uab_(i, j - 1)) / K2(DY4, i, j) + (vab_(i, j) - vab_(i - 1, j)) / K2(DX4, i, j));
}
__device__ __forceinline__ double advave_fu(const KP &P, int i, int j) {                    // fluxua, u half; 2<=j<=jm
  if (i < 2 || i > P.imm1) return 0.;
  double f = .125 * ((d_(i + 1, j) + d_(i, j)) * ua_(i + 1, j) + (d_(i, j) + d_(i - 1, j)) * ua_(i, j)) * (ua_(i + 1, j) + ua_(i, j));
  f = f - d_(i, j) * 2. * aam2d_(i, j) * (uab_(i + 1, j) - uab_(i, j)) / dx_(i, j);
  return f * dy_(i, j);
}
__device__ __forceinline__ double advave_fv(const KP &P, int i, int j, double tps) {        // fluxva, u half; 2<=i<=im, 2<=j<=jm
  const double f = .125 * ((d_(i, j) + d_(i, j - 1)) * va_(i, j) + (d_(i - 1, j) + d_(i - 1, j - 1)) * va_(i - 1, j)) *
                   (ua_(i, j) + ua_(i, j - 1));
  return (f - tps) * .25 * K2(DX4, i, j);
}
__device__ __forceinline__ double advave_gu(const KP &P, int i, int j, double tps) {        // fluxua, v half; 2<=i<=im, 2<=j<=jm
  const double f = .125 * ((d_(i, j) + d_(i - 1, j)) * ua_(i, j) + (d_(i, j - 1) + d_(i - 1, j - 1)) * ua_(i, j - 1)) *
                   (va_(i - 1, j) + va_(i, j));
  return (f - tps) * .25 * K2(DY4, i, j);
}
__device__ __forceinline__ double advave_gv(const KP &P, int i, int j) {                    // fluxva, v half; 2<=i<=im
  if (j < 2 || j > P.jmm1) return 0.;
  double f = .125 * ((d_(i, j + 1) + d_(i, j)) * va_(i, j + 1) + (d_(i, j) + d_(i, j - 1)) * va_(i, j)) * (va_(i, j + 1) + va_(i, j));
  f = f - d_(i, j) * 2. * aam2d_(i, j) * (vab_(i, j + 1) - vab_(i, j)) / dy_(i, j);
  return f * dx_(i, j);
}
__global__ void k_advave_fused(KP P) {
  const int lane = HALO_LANE, i0 = HALO_COL, j = TID_J;
  if (j > P.jml) return;                                   // a whole wavefront (one row)
  const bool out = (lane >= 1 && lane <= 62 && i0 <= P.iml);
#ifdef POMGPU_EMU
  if (!out) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.im ? P.im : i0);      // halo / padding lanes shadow a valid column
  const bool row = (j >= 2 && j <= P.jmm1);                // rows on which advua/advva are formed
  const bool in = out && row && (i0 >= 2 && i0 <= P.imm1);
  double fu = 0., gu = 0., tps = 0.;
  if (row) {
    fu = advave_fu(P, i, j);
    if (i >= 2) { tps = advave_tps(P, i, j); gu = advave_gu(P, i, j, tps); }
  }
  const double fu_w = halo_w(fu, [&] { return (row && i >= 2) ? advave_fu(P, i - 1, j) : 0.; });
  const double gu_e = halo_e(gu, [&] { return (row && i + 1 <= P.im) ? advave_gu(P, i + 1, j, advave_tps(P, i + 1, j)) : 0.; });
  double au = 0., av = 0.;
  if (in) {
    au = fu - fu_w + advave_fv(P, i, j + 1, advave_tps(P, i, j + 1)) - advave_fv(P, i, j, tps);      // :65-66
    av = gu_e - gu + advave_gv(P, i, j) - advave_gv(P, i, j - 1);                                   // :116-117
  }
  if (out) {
    F2(advua, i0, j) = au;                                 // advua = 0., advva = 0. elsewhere (:16,:73)
    F2(advva, i0, j) = av;
  }
}

// ---------------------------------------------------------------------------------------------
// advave, single tile, TWO columns per lane.  PMC of the kernel above: 56 wavefront loads per cell,
// 74 % L1/TA busy, 124 us per launch for 0.3 GB -- bound by the NUMBER of load instructions (an
// 8-byte-per-lane load occupies the texture-address path as long as a 16-byte one; micro-benchmark
// tools/micro/ta_width.hip: 65 us -> 37 us for a 9-point stencil over 6 arrays).  Here a lane owns
// the columns (i, i+1), i odd: every operand row is ONE aligned 16-byte load, all i-1 / i+1 operands
// come from the neighbour lane (halo-lane wavefronts: lanes 1..62 own 124 output columns), and the
// four rows j-2..j+1 are loaded once: 27 loads per two cells.  Needs an even leading dimension.
struct AdvaveCell {   // operands of one cell: C(entre), W(est), E(ast) at rows m2 = j-2, m1 = j-1, 0 = j, p1 = j+1
  double d_m2, d_m1, d_0, d_p1, dW_m1, dW_0, dW_p1, dE_0;
  double ua_m1, ua_0, ua_p1, uaE_0, va_m1, va_0, va_p1, vaW_0, vaW_p1;
  double uab_m1, uab_0, uab_p1, uabE_0, vab_m1, vab_0, vab_p1, vabW_0, vabW_p1;
  double am_m1, am_0, am_p1, amW_m1, amW_0, amW_p1;
  double dx_0, dx_m1, dy_0, dy_m1, DX4_0, DX4_p1, DY4_0, DY4_p1;
};
struct AdvaveOut { double fu, gu, fv0, fvP, gv0, gvM; };
// the fluxes of one cell -- the expressions of advave_fu/tps/fv/gu/gv above on register operands.
// fu_on: 2 <= i <= imm1; w_on: i >= 2 (corner quantities exist); sm1: row j-1 lies in 2..jmm1
__device__ __forceinline__ AdvaveOut advave_cell(const AdvaveCell &c, bool fu_on, bool w_on, bool sm1) {
  AdvaveOut r;
  r.fu = r.gu = r.fv0 = r.fvP = r.gv0 = r.gvM = 0.;
  if (fu_on) {
    double f = .125 * ((c.dE_0 + c.d_0) * c.uaE_0 + (c.d_0 + c.dW_0) * c.ua_0) * (c.uaE_0 + c.ua_0);
    f = f - c.d_0 * 2. * c.am_0 * (c.uabE_0 - c.uab_0) / c.dx_0;
    r.fu = f * c.dy_0;
  }
  if (w_on) {
    const double tps0 = .25 * (c.d_0 + c.dW_0 + c.d_m1 + c.dW_m1) * (c.am_0 + c.am_m1 + c.amW_0 + c.amW_m1) *
                        ((c.uab_0 - c.uab_m1) / c.DY4_0 + (c.vab_0 - c.vabW_0) / c.DX4_0);
    const double tpsP = .25 * (c.d_p1 + c.dW_p1 + c.d_0 + c.dW_0) * (c.am_p1 + c.am_0 + c.amW_p1 + c.amW_0) *
                        ((c.uab_p1 - c.uab_0) / c.DY4_p1 + (c.vab_p1 - c.vabW_p1) / c.DX4_p1);
    const double g = .125 * ((c.d_0 + c.dW_0) * c.ua_0 + (c.d_m1 + c.dW_m1) * c.ua_m1) * (c.vaW_0 + c.va_0);
    r.gu = (g - tps0) * .25 * c.DY4_0;
    const double f0 = .125 * ((c.d_0 + c.d_m1) * c.va_0 + (c.dW_0 + c.dW_m1) * c.vaW_0) * (c.ua_0 + c.ua_m1);
    r.fv0 = (f0 - tps0) * .25 * c.DX4_0;
    const double fP = .125 * ((c.d_p1 + c.d_0) * c.va_p1 + (c.dW_p1 + c.dW_0) * c.vaW_p1) * (c.ua_p1 + c.ua_0);
    r.fvP = (fP - tpsP) * .25 * c.DX4_p1;
    double gv0 = .125 * ((c.d_p1 + c.d_0) * c.va_p1 + (c.d_0 + c.d_m1) * c.va_0) * (c.va_p1 + c.va_0);
    gv0 = gv0 - c.d_0 * 2. * c.am_0 * (c.vab_p1 - c.vab_0) / c.dy_0;
    r.gv0 = gv0 * c.dx_0;
    if (sm1) {
      double gvM = .125 * ((c.d_0 + c.d_m1) * c.va_0 + (c.d_m1 + c.d_m2) * c.va_m1) * (c.va_0 + c.va_m1);
      gvM = gvM - c.d_m1 * 2. * c.am_m1 * (c.vab_0 - c.vab_m1) / c.dy_m1;
      r.gvM = gvM * c.dx_m1;
    }
  }
  return r;
}
#define LD2(ptr, ii, jj) (*(const double2 *)&(ptr)[IX2(ii, jj)])
#define A2(name) (P.b2 + (size_t)P2_##name * P.n2)
__global__ void __launch_bounds__(256) k_advave_pair(KP P) {
  const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int lane = g & 63;
  const int ia0 = 2 * ((g >> 6) * 62 + lane - 1) + 1;       // first (odd, 1-based) column of this lane; -1 for the first halo lane
  const int j = TID_J;
  if (j > P.jml) return;                                   // a whole wavefront (one row)
  const bool out = (lane >= 1 && lane <= 62 && ia0 <= P.iml);
#ifdef POMGPU_EMU
  if (!out) return;
#endif
  const int ia = ia0 < 1 ? 1 : (ia0 > P.iml - 1 ? P.iml - 1 : ia0), ib = ia + 1;   // halo / padding lanes shadow a valid pair
  const int iw = ia > 1 ? ia - 1 : 1, ie = ib < P.iml ? ib + 1 : P.iml;
  const int jm2 = j > 2 ? j - 2 : 1, jm1 = j > 1 ? j - 1 : 1, jp1 = j < P.jml ? j + 1 : P.jml;
  const bool row = (j >= 2 && j <= P.jmm1);
  double2 au = {0., 0.}, av = {0., 0.};
  if (row) {                                               // wave-uniform
    const double2 d_m2 = LD2(P.x2[X2_d], ia, jm2), d_m1 = LD2(P.x2[X2_d], ia, jm1), d_0 = LD2(P.x2[X2_d], ia, j), d_p1 = LD2(P.x2[X2_d], ia, jp1);
    const double2 ua_m1 = LD2(P.x2[X2_ua], ia, jm1), ua_0 = LD2(P.x2[X2_ua], ia, j), ua_p1 = LD2(P.x2[X2_ua], ia, jp1);
    const double2 va_m1 = LD2(P.x2[X2_va], ia, jm1), va_0 = LD2(P.x2[X2_va], ia, j), va_p1 = LD2(P.x2[X2_va], ia, jp1);
    const double2 uab_m1 = LD2(P.x2[X2_uab], ia, jm1), uab_0 = LD2(P.x2[X2_uab], ia, j), uab_p1 = LD2(P.x2[X2_uab], ia, jp1);
    const double2 vab_m1 = LD2(P.x2[X2_vab], ia, jm1), vab_0 = LD2(P.x2[X2_vab], ia, j), vab_p1 = LD2(P.x2[X2_vab], ia, jp1);
    const double2 am_m1 = LD2(A2(aam2d), ia, jm1), am_0 = LD2(A2(aam2d), ia, j), am_p1 = LD2(A2(aam2d), ia, jp1);
    const double2 dx_m1 = LD2(A2(dx), ia, jm1), dx_0 = LD2(A2(dx), ia, j), dy_m1 = LD2(A2(dy), ia, jm1), dy_0 = LD2(A2(dy), ia, j);
    const double2 DX4_0 = LD2(P.c2[C2_DX4], ia, j), DX4_p1 = LD2(P.c2[C2_DX4], ia, jp1);
    const double2 DY4_0 = LD2(P.c2[C2_DY4], ia, j), DY4_p1 = LD2(P.c2[C2_DY4], ia, jp1);
    // west operands of column ia (= the east column of the lane to the west), east operands of column ib
#define WV(v2, name, jj) halo_w((v2).y, [&] { return name##_(iw, jj); })
#define EV(v2, name, jj) halo_e((v2).x, [&] { return name##_(ie, jj); })
    AdvaveCell a, b;
    a.d_m2 = d_m2.x; a.d_m1 = d_m1.x; a.d_0 = d_0.x; a.d_p1 = d_p1.x;
    b.d_m2 = d_m2.y; b.d_m1 = d_m1.y; b.d_0 = d_0.y; b.d_p1 = d_p1.y;
    a.dW_m1 = WV(d_m1, d, jm1); a.dW_0 = WV(d_0, d, j); a.dW_p1 = WV(d_p1, d, jp1); a.dE_0 = d_0.y;
    b.dW_m1 = d_m1.x; b.dW_0 = d_0.x; b.dW_p1 = d_p1.x; b.dE_0 = EV(d_0, d, j);
    a.ua_m1 = ua_m1.x; a.ua_0 = ua_0.x; a.ua_p1 = ua_p1.x; a.uaE_0 = ua_0.y;
    b.ua_m1 = ua_m1.y; b.ua_0 = ua_0.y; b.ua_p1 = ua_p1.y; b.uaE_0 = EV(ua_0, ua, j);
    a.va_m1 = va_m1.x; a.va_0 = va_0.x; a.va_p1 = va_p1.x; a.vaW_0 = WV(va_0, va, j); a.vaW_p1 = WV(va_p1, va, jp1);
    b.va_m1 = va_m1.y; b.va_0 = va_0.y; b.va_p1 = va_p1.y; b.vaW_0 = va_0.x; b.vaW_p1 = va_p1.x;
    a.uab_m1 = uab_m1.x; a.uab_0 = uab_0.x; a.uab_p1 = uab_p1.x; a.uabE_0 = uab_0.y;
    b.uab_m1 = uab_m1.y; b.uab_0 = uab_0.y; b.uab_p1 = uab_p1.y; b.uabE_0 = EV(uab_0, uab, j);
    a.vab_m1 = vab_m1.x; a.vab_0 = vab_0.x; a.vab_p1 = vab_p1.x; a.vabW_0 = WV(vab_0, vab, j); a.vabW_p1 = WV(vab_p1, vab, jp1);
    b.vab_m1 = vab_m1.y; b.vab_0 = vab_0.y; b.vab_p1 = vab_p1.y; b.vabW_0 = vab_0.x; b.vabW_p1 = vab_p1.x;
    a.am_m1 = am_m1.x; a.am_0 = am_0.x; a.am_p1 = am_p1.x;
    a.amW_m1 = WV(am_m1, aam2d, jm1); a.amW_0 = WV(am_0, aam2d, j); a.amW_p1 = WV(am_p1, aam2d, jp1);
    b.am_m1 = am_m1.y; b.am_0 = am_0.y; b.am_p1 = am_p1.y; b.amW_m1 = am_m1.x; b.amW_0 = am_0.x; b.amW_p1 = am_p1.x;
    a.dx_0 = dx_0.x; a.dx_m1 = dx_m1.x; a.dy_0 = dy_0.x; a.dy_m1 = dy_m1.x;
    b.dx_0 = dx_0.y; b.dx_m1 = dx_m1.y; b.dy_0 = dy_0.y; b.dy_m1 = dy_m1.y;
    a.DX4_0 = DX4_0.x; a.DX4_p1 = DX4_p1.x; a.DY4_0 = DY4_0.x; a.DY4_p1 = DY4_p1.x;
    b.DX4_0 = DX4_0.y; b.DX4_p1 = DX4_p1.y; b.DY4_0 = DY4_0.y; b.DY4_p1 = DY4_p1.y;
#undef WV
#undef EV
    const bool sm1 = (j - 1 >= 2);
    const AdvaveOut fa = advave_cell(a, ia >= 2 && ia <= P.imm1, ia >= 2 && ia <= P.im, sm1);
    const AdvaveOut fb = advave_cell(b, ib >= 2 && ib <= P.imm1, ib <= P.im, sm1);
    const double fu_w = halo_w(fb.fu, [&] { return ia >= 2 ? advave_fu(P, ia - 1, j) : 0.; });
    const double gu_e = halo_e(fa.gu, [&] { return ib + 1 <= P.im ? advave_gu(P, ib + 1, j, advave_tps(P, ib + 1, j)) : 0.; });
    if (ia0 >= 2 && ia0 <= P.imm1) {
      au.x = fa.fu - fu_w + fa.fvP - fa.fv0;               // :65-66
      av.x = fb.gu - fa.gu + fa.gv0 - fa.gvM;              // :116-117
    }
    if (ia0 + 1 >= 2 && ia0 + 1 <= P.imm1) {
      au.y = fb.fu - fa.fu + fb.fvP - fb.fv0;
      av.y = gu_e - fb.gu + fb.gv0 - fb.gvM;
    }
  }
  if (out) {
    *(double2 *)&F2(advua, ia0, j) = au;                    // advua = 0., advva = 0. outside the interior (:16,:73)
    *(double2 *)&F2(advva, ia0, j) = av;
  }
}
#undef LD2
#undef A2

// advave, mode==2 only: bottom stress and curvature terms -- solver.f:123-195
__global__ void k_advave_m2a(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.im || j > P.jm) return;
  double cv = 0.;
  if (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1) {
    F2(wubot, i, j) = -0.5 * (F2(cbc, i, j) + F2(cbc, i - 1, j)) *
                      sqrt(sq(uab_(i, j)) + sq(.25 * (vab_(i, j) + vab_(i, j + 1) + vab_(i - 1, j) + vab_(i - 1, j + 1)))) *
                      uab_(i, j);
    F2(wvbot, i, j) = -0.5 * (F2(cbc, i, j) + F2(cbc, i, j - 1)) *
                      sqrt(sq(vab_(i, j)) + sq(.25 * (uab_(i, j) + uab_(i + 1, j) + uab_(i, j - 1) + uab_(i + 1, j - 1)))) *
                      vab_(i, j);
    cv = .25 * ((va_(i, j + 1) + va_(i, j)) * (dy_(i + 1, j) - dy_(i - 1, j)) -
                (ua_(i + 1, j) + ua_(i, j)) * (dx_(i, j + 1) - dx_(i, j - 1))) / (dx_(i, j) * dy_(i, j));
  }
  G2(P.s2[2], i, j) = cv;   // curv2d
}
__global__ void k_advave_m2b(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.imm1 || j > P.jmm1 || i < 2 || j < 2) return;
  const double *cv = P.s2[2];
  if (i >= (P.W ? 3 : 2))
    F2(advua, i, j) = F2(advua, i, j) - F2(aru, i, j) * .25 *
                      (G2(cv, i, j) * d_(i, j) * (va_(i, j + 1) + va_(i, j)) +
                       G2(cv, i - 1, j) * d_(i - 1, j) * (va_(i - 1, j + 1) + va_(i - 1, j)));
  if (j >= (P.S ? 3 : 2))
    F2(advva, i, j) = F2(advva, i, j) + F2(arv, i, j) * .25 *
                      (G2(cv, i, j) * d_(i, j) * (ua_(i + 1, j) + ua_(i, j)) +
                       G2(cv, i, j - 1) * d_(i, j - 1) * (ua_(i + 1, j - 1) + ua_(i, j - 1)));
}

// ---------------------------------------------------------------------------------------------
// mode_interaction, vertical integrals -- advance.f:152-168.  One thread per column, k in a
// register loop; reads 5 3-D arrays once (coalesced planes), writes 5 2-D arrays.
// only_aam: adx2d, ady2d, drx2d, dry2d were left by k_advct_col / k_baropg (sum2d) in this step
// ghost (with only_aam; pomgpu_api.hip "rim rounds"): +1 = every cell but the lines a neighbour tile owns -- their aam arrives with a message round that
// runs beside this kernel --, -1 = those lines alone, on the side stream behind that round
__global__ void k_vint(KP P, int only_aam, int ghost) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  if (ghost) {
    const bool gl = (i == 1 && !P.W) || (i == P.im && !P.E) || (j == 1 && !P.S) || (j == P.jm && !P.N);
    if ((ghost > 0) == gl) return;
  }
  double ax = 0., ay = 0., rx = 0., ry = 0., am = 0.;
  if (only_aam) {
    if (i <= P.im && j <= P.jm)
      for (int k = 1; k <= P.kbm1; k++) am = am + F3(aam, i, j, k) * F1(dz, k);
    F2(aam2d, i, j) = am;
    return;
  }
  if (i <= P.im && j <= P.jm) {
    for (int k = 1; k <= P.kbm1; k++) {
      const double dzk = F1(dz, k);
      ax = ax + F3(advx, i, j, k) * dzk;
      ay = ay + F3(advy, i, j, k) * dzk;
      rx = rx + F3(drhox, i, j, k) * dzk;
      ry = ry + F3(drhoy, i, j, k) * dzk;
      am = am + F3(aam, i, j, k) * dzk;
    }
  }
  F2(adx2d, i, j) = ax;
  F2(ady2d, i, j) = ay;
  F2(drx2d, i, j) = rx;
  F2(dry2d, i, j) = ry;
  F2(aam2d, i, j) = am;
}

// mode_interaction tail -- advance.f:172-196
__global__ void k_modeint_tail(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.im || j > P.jm) return;
  if (P.mode != 2) {
    F2(adx2d, i, j) = F2(adx2d, i, j) - F2(advua, i, j);
    F2(ady2d, i, j) = F2(ady2d, i, j) - F2(advva, i, j);
  }
  F2(egf, i, j) = el_(i, j) * P.ispi;
  if (i >= 2) F2(utf, i, j) = ua_(i, j) * (d_(i, j) + d_(i - 1, j)) * P.isp2i;
  if (j >= 2) F2(vtf, i, j) = va_(i, j) * (d_(i, j) + d_(i, j - 1)) * P.isp2i;
}

// ---------------------------------------------------------------------------------------------
// mode_external, continuity + bcond(1) -- advance.f:211-231, bounds_forcing.f:18-41.
// fluxua/fluxva are recomputed in registers; the zero-gradient edge copy becomes a clamped
// source index (the value at the clamped interior point is the same arithmetic on the same data).
__device__ __forceinline__ double flux_ua(const KP &P, int i, int j) {
  return .25 * (d_(i, j) + d_(i - 1, j)) * (dy_(i, j) + dy_(i - 1, j)) * ua_(i, j);
}
__device__ __forceinline__ double flux_va(const KP &P, int i, int j) {
  return .25 * (d_(i, j) + d_(i, j - 1)) * (dx_(i, j) + dx_(i, j - 1)) * va_(i, j);
}
// the new surface elevation of cell (i,j), i <= iml, j <= jml, including bcond(1) (clamp + mask)
__device__ __forceinline__ double elf_at(const KP &P, int i, int j) {
  const int ii = (P.W && i == 1) ? 2 : ((P.E && i == P.im) ? P.imm1 : i);
  const int jj = (P.S && j == 1) ? 2 : ((P.N && j == P.jm) ? P.jmm1 : j);
  double v;
  if (ii >= 2 && ii <= P.imm1 && jj >= 2 && jj <= P.jmm1 && i <= P.im && j <= P.jm)
    v = elb_(ii, jj) +
        P.dte2 * (-(flux_ua(P, ii + 1, jj) - flux_ua(P, ii, jj) + flux_va(P, ii, jj + 1) - flux_va(P, ii, jj)) /
                      F2(art, ii, jj) -
                  F2(vfluxf, ii, jj));
  else
    v = F2(elf, i, j);
  return v * F2(fsm, i, j);
}
__global__ void k_ext_elf(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  F2(elf, i, j) = elf_at(P, i, j);
}

// bcond(1) alone (for the stand-alone entry point) -- bounds_forcing.f:18-41
__global__ void k_bcond1(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const int ii = (P.W && i == 1) ? 2 : ((P.E && i == P.im) ? P.imm1 : i);
  const int jj = (P.S && j == 1) ? 2 : ((P.N && j == P.jm) ? P.jmm1 : j);
  G2(P.s2[3], i, j) = F2(elf, ii, jj) * F2(fsm, i, j);
}
__global__ void k_copy2(KP P, double *dst, const double *src) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  G2(dst, i, j) = G2(src, i, j);
}

// mode_external, momentum + bcond(2) -- advance.f:237-290, bounds_forcing.f:43-83
// ec / ew / es: the new elevation elf at (i,j), (i-1,j), (i,j-1)
__device__ __forceinline__ double uaf_interior(const KP &P, int i, int j, double ec, double ew, double adu) {
  double v = F2(adx2d, i, j) + adu -
             F2(aru, i, j) * .25 *
                 (F2(cor, i, j) * d_(i, j) * (va_(i, j + 1) + va_(i, j)) +
                  F2(cor, i - 1, j) * d_(i - 1, j) * (va_(i - 1, j + 1) + va_(i - 1, j))) +
             .25 * P.grav * (dy_(i, j) + dy_(i - 1, j)) * (d_(i, j) + d_(i - 1, j)) *
                 ((1. - 2. * P.alpha) * (el_(i, j) - el_(i - 1, j)) +
                  P.alpha * (elb_(i, j) - elb_(i - 1, j) + ec - ew) +
                  F2(e_atmos, i, j) - F2(e_atmos, i - 1, j)) +
             F2(drx2d, i, j) + F2(aru, i, j) * (F2(wusurf, i, j) - F2(wubot, i, j));          // :239-250
  v = ((F2(h, i, j) + elb_(i, j) + F2(h, i - 1, j) + elb_(i - 1, j)) * F2(aru, i, j) * uab_(i, j) -
       4. * P.dte * v) /
      ((F2(h, i, j) + ec + F2(h, i - 1, j) + ew) * F2(aru, i, j));                            // :256-260
  return v;
}
__device__ __forceinline__ double vaf_interior(const KP &P, int i, int j, double ec, double es, double adv) {
  double v = F2(ady2d, i, j) + adv +
             F2(arv, i, j) * .25 *
                 (F2(cor, i, j) * d_(i, j) * (ua_(i + 1, j) + ua_(i, j)) +
                  F2(cor, i, j - 1) * d_(i, j - 1) * (ua_(i + 1, j - 1) + ua_(i, j - 1))) +
             .25 * P.grav * (dx_(i, j) + dx_(i, j - 1)) * (d_(i, j) + d_(i, j - 1)) *
                 ((1. - 2. * P.alpha) * (el_(i, j) - el_(i, j - 1)) +
                  P.alpha * (elb_(i, j) - elb_(i, j - 1) + ec - es) +
                  F2(e_atmos, i, j) - F2(e_atmos, i, j - 1)) +
             F2(dry2d, i, j) + F2(arv, i, j) * (F2(wvsurf, i, j) - F2(wvbot, i, j));          // :266-276
  v = ((F2(h, i, j) + elb_(i, j) + F2(h, i, j - 1) + elb_(i, j - 1)) * F2(arv, i, j) * vab_(i, j) -
       4. * P.dte * v) /
      ((F2(h, i, j) + ec + F2(h, i, j - 1) + es) * F2(arv, i, j));                            // :282-286
  return v;
}
// the open-boundary values of bcond(2); `interior` = 0 skips the advance.f formulas (bcond alone)
// adu, adv: advua(i,j), advva(i,j)
__device__ __forceinline__ void uvaf_cell(const KP &P, int i, int j, int interior, double ec, double ew, double es, double adu, double adv,
                                          double &uo, double &vo) {
  double u = F2(uaf, i, j), v = F2(vaf, i, j);
  const bool jin = (j >= 2 && j <= P.jmm1), iin = (i >= 2 && i <= P.imm1);
  if (interior) {
    if (i >= 2 && i <= P.im && jin) u = uaf_interior(P, i, j, ec, ew, adu);
    if (iin && j >= 2 && j <= P.jm) v = vaf_interior(P, i, j, ec, es, adv);
  }
  if (P.W && jin && (i == 1 || i == 2)) {                                                 // :47-53
    if (i == 1) v = BD1(vabw, j);
    u = BD1(uabw, j) - P.rfw * sqrt(P.grav / d_(2, j)) * (el_(2, j) - BD1(elw, j));
    u = P.ramp * u;
  }
  if (P.E && jin && i == P.im) {                                                          // :56-61
    u = BD1(uabe, j) + P.rfe * sqrt(P.grav / d_(P.imm1, j)) * (el_(P.imm1, j) - BD1(ele, j));
    u = P.ramp * u;
    v = BD1(vabe, j);
  }
  if (P.S && iin && (j == 1 || j == 2)) {                                                 // :64-70
    if (j == 1) u = BD1(uabs, i);
    v = BD1(vabs, i) - P.rfs * sqrt(P.grav / d_(i, 2)) * (el_(i, 2) - BD1(els, i));
    v = P.ramp * v;
  }
  if (P.N && iin && j == P.jm) {                                                          // :73-78
    v = BD1(vabn, i) + P.rfn * sqrt(P.grav / d_(i, P.jmm1)) * (el_(i, P.jmm1) - BD1(eln, i));
    v = P.ramp * v;
    u = BD1(uabn, i);
  }
  uo = u;
  vo = v;
}
__global__ void k_ext_uvaf(KP P, int interior) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  double u, v;
  if (i <= P.im && j <= P.jm) {
    const double ec = F2(elf, i, j), ew = i >= 2 ? F2(elf, i - 1, j) : 0., es = j >= 2 ? F2(elf, i, j - 1) : 0.;
    uvaf_cell(P, i, j, interior, ec, ew, es, F2(advua, i, j), F2(advva, i, j), u, v);
  } else { u = F2(uaf, i, j); v = F2(vaf, i, j); }
  F2(uaf, i, j) = u * F2(dum, i, j);                                                      // :80-81
  F2(vaf, i, j) = v * F2(dvm, i, j);
}

// mode_external, etf weights + Asselin filter + time rotation + accumulation -- advance.f:295-350
// elf, uaf, vaf: the new values at (i,j); ew, es: elf at (i-1,j), (i,j-1).  Reads generation x2, writes y2.
__device__ __forceinline__ void ext_update_cell(const KP &P, int i, int j, bool act, double elf, double uaf, double vaf, double ew, double es) {
  if (act) {                                                                              // :295-318
    if (P.iext == P.isplit - 2) F2(etf, i, j) = .25 * P.smoth * elf;
    else if (P.iext == P.isplit - 1) F2(etf, i, j) = F2(etf, i, j) + .5 * (1. - .5 * P.smoth) * elf;
    else if (P.iext == P.isplit) F2(etf, i, j) = (F2(etf, i, j) + .5 * elf) * F2(fsm, i, j);
  }
  const double ua = ua_(i, j), va = va_(i, j), el = el_(i, j);
  Y2(uab, i, j) = ua + .5 * P.smoth * (uab_(i, j) - 2. * ua + uaf);                       // :321,327
  Y2(vab, i, j) = va + .5 * P.smoth * (vab_(i, j) - 2. * va + vaf);                       // :322,329
  Y2(elb, i, j) = el + .5 * P.smoth * (elb_(i, j) - 2. * el + elf);                       // :323-324
  Y2(el, i, j) = elf;                                                                     // :325
  const double dn = F2(h, i, j) + elf;
  Y2(d, i, j) = dn;                                                                       // :326
  Y2(ua, i, j) = uaf;                                                                     // :328
  Y2(va, i, j) = vaf;                                                                     // :330
  if (P.iext != P.isplit && act) {                                                        // :332-347
    F2(egf, i, j) = F2(egf, i, j) + elf * P.ispi;
    if (i >= 2) F2(utf, i, j) = F2(utf, i, j) + uaf * (dn + (F2(h, i - 1, j) + ew)) * P.isp2i;
    if (j >= 2) F2(vtf, i, j) = F2(vtf, i, j) + vaf * (dn + (F2(h, i, j - 1) + es)) * P.isp2i;
  }
}
__global__ void k_ext_update(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const bool act = (i <= P.im && j <= P.jm), acc = (act && P.iext != P.isplit);
  ext_update_cell(P, i, j, act, F2(elf, i, j), F2(uaf, i, j), F2(vaf, i, j), (acc && i >= 2) ? F2(elf, i - 1, j) : 0.,
                  (acc && j >= 2) ? F2(elf, i, j - 1) : 0.);
}

// ---- branch-free forms for the rim cells of the fused substep kernel ------------------------------------------------
// A rim thread walks elf at three cells, advave, the momentum formulas and the update.  Written with range checks
// around each piece, every piece's loads wait for the arithmetic of the one before (in-order issue): ~15-20 us of
// dependent latency that bounds the whole launch on small grids (256x256: 23 us per substep, 14 of them this chain).
// Here every operand is loaded from a clamped, always valid index and the result selected afterwards, so the loads
// of all pieces can be in flight together.  Same formulas, same operands where the result is used.
__device__ __forceinline__ double elf_at_nb(const KP &P, int i, int j) {                     // i or j < 1: 0 (the callers' guards)
  const bool cell = (i >= 1 && j >= 1);
  const int ci = i < 1 ? 1 : i, cj = j < 1 ? 1 : j;
  const int ii = (P.W && ci == 1) ? 2 : ((P.E && ci == P.im) ? P.imm1 : ci);
  const int jj = (P.S && cj == 1) ? 2 : ((P.N && cj == P.jm) ? P.jmm1 : cj);
  const bool ok = ii >= 2 && ii <= P.imm1 && jj >= 2 && jj <= P.jmm1 && ci <= P.im && cj <= P.jm;
  const int si = ok ? ii : 2, sj = ok ? jj : 2;
  const double v1 = elb_(si, sj) +
                    P.dte2 * (-(flux_ua(P, si + 1, sj) - flux_ua(P, si, sj) + flux_va(P, si, sj + 1) - flux_va(P, si, sj)) /
                                  F2(art, si, sj) -
                              F2(vfluxf, si, sj));
  const double v2 = F2(elf, ci, cj);
  const double r = (ok ? v1 : v2) * F2(fsm, ci, cj);
  return cell ? r : 0.;
}
__device__ __forceinline__ double advave_fu_raw(const KP &P, int i, int j) {                 // advave_fu without its range guard
  double f = .125 * ((d_(i + 1, j) + d_(i, j)) * ua_(i + 1, j) + (d_(i, j) + d_(i - 1, j)) * ua_(i, j)) * (ua_(i + 1, j) + ua_(i, j));
  f = f - d_(i, j) * 2. * aam2d_(i, j) * (uab_(i + 1, j) - uab_(i, j)) / dx_(i, j);
  return f * dy_(i, j);
}
__device__ __forceinline__ double advave_gv_raw(const KP &P, int i, int j) {                 // advave_gv without its range guard
  double f = .125 * ((d_(i, j + 1) + d_(i, j)) * va_(i, j + 1) + (d_(i, j) + d_(i, j - 1)) * va_(i, j)) * (va_(i, j + 1) + va_(i, j));
  f = f - d_(i, j) * 2. * aam2d_(i, j) * (vab_(i, j + 1) - vab_(i, j)) / dy_(i, j);
  return f * dx_(i, j);
}
__device__ __forceinline__ void advave_at_nb(const KP &P, int i, int j, double &au, double &av) {
  const bool in = (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1);
  const int si = in ? i : 2, sj = in ? j : 2;                 // 2 <= si <= imm1, 2 <= sj <= jmm1
  const double tps = advave_tps(P, si, sj);
  const int wi = si >= 3 ? si - 1 : 2, mj = sj >= 3 ? sj - 1 : 2;
  const double fu_c = advave_fu_raw(P, si, sj), fu_w = si >= 3 ? advave_fu_raw(P, wi, sj) : 0.;                      // fluxua(1,j) = 0
  const double a = fu_c - fu_w + advave_fv(P, si, sj + 1, advave_tps(P, si, sj + 1)) - advave_fv(P, si, sj, tps);   // :65-66
  const double gv_c = advave_gv_raw(P, si, sj), gv_s = sj >= 3 ? advave_gv_raw(P, si, mj) : 0.;                      // fluxva(i,1) = 0
  const double b = advave_gu(P, si + 1, sj, advave_tps(P, si + 1, sj)) - advave_gu(P, si, sj, tps) + gv_c - gv_s;   // :116-117
  au = in ? a : 0.;
  av = in ? b : 0.;
}
// uvaf_cell with the advance.f formulas evaluated at a clamped cell and selected; the open-boundary branches as they are
__device__ __forceinline__ void uvaf_cell_nb(const KP &P, int i, int j, double ec, double ew, double es, double adu, double adv, double &uo,
                                             double &vo) {
  const bool jin = (j >= 2 && j <= P.jmm1), iin = (i >= 2 && i <= P.imm1);
  const bool ucalc = (i >= 2 && i <= P.im && jin), vcalc = (iin && j >= 2 && j <= P.jm);
  const double u0 = F2(uaf, i, j), v0 = F2(vaf, i, j);
  const double u1 = uaf_interior(P, ucalc ? i : 2, ucalc ? j : 2, ec, ew, adu);
  const double v1 = vaf_interior(P, vcalc ? i : 2, vcalc ? j : 2, ec, es, adv);
  double u = ucalc ? u1 : u0, v = vcalc ? v1 : v0;
  if (P.W && jin && (i == 1 || i == 2)) {                                                 // :47-53
    if (i == 1) v = BD1(vabw, j);
    u = BD1(uabw, j) - P.rfw * sqrt(P.grav / d_(2, j)) * (el_(2, j) - BD1(elw, j));
    u = P.ramp * u;
  }
  if (P.E && jin && i == P.im) {                                                          // :56-61
    u = BD1(uabe, j) + P.rfe * sqrt(P.grav / d_(P.imm1, j)) * (el_(P.imm1, j) - BD1(ele, j));
    u = P.ramp * u;
    v = BD1(vabe, j);
  }
  if (P.S && iin && (j == 1 || j == 2)) {                                                 // :64-70
    if (j == 1) u = BD1(uabs, i);
    v = BD1(vabs, i) - P.rfs * sqrt(P.grav / d_(i, 2)) * (el_(i, 2) - BD1(els, i));
    v = P.ramp * v;
  }
  if (P.N && iin && j == P.jm) {                                                          // :73-78
    v = BD1(vabn, i) + P.rfn * sqrt(P.grav / d_(i, P.jmm1)) * (el_(i, P.jmm1) - BD1(eln, i));
    v = P.ramp * v;
    u = BD1(uabn, i);
  }
  uo = u;
  vo = v;
}
// One cell of a substep by the general formulas (any cell: rim cells with bcond(1) / bcond(2), interior cells alike), from memory.
// fuse_adv: advua, advva are formed here (advave is part of this substep's kernel) instead of read
__device__ __forceinline__ void ext_cell_general(const KP &P, int i, int j, int store_f, int fuse_adv) {
  double ec, ew, es, u, v, adu, adv;
  if (fuse_adv) {                                           // the fused kernel's rim: branch-free forms, loads in flight together
    ec = elf_at_nb(P, i, j); ew = elf_at_nb(P, i - 1, j); es = elf_at_nb(P, i, j - 1);
    advave_at_nb(P, i, j, adu, adv);
    uvaf_cell_nb(P, i, j, ec, ew, es, adu, adv, u, v);
  } else {
    ec = elf_at(P, i, j); ew = i >= 2 ? elf_at(P, i - 1, j) : 0.; es = j >= 2 ? elf_at(P, i, j - 1) : 0.;
    adu = F2(advua, i, j); adv = F2(advva, i, j);
    uvaf_cell(P, i, j, 1, ec, ew, es, adu, adv, u, v);
  }
  u = u * F2(dum, i, j);
  v = v * F2(dvm, i, j);
  if (store_f) {
    F2(elf, i, j) = ec; F2(uaf, i, j) = u; F2(vaf, i, j) = v;
    if (fuse_adv) { F2(advua, i, j) = adu; F2(advva, i, j) = adv; }
  }
  ext_update_cell(P, i, j, true, ec, u, v, ew, es);
}
// the three outermost lines on every side, with bcond(1) and bcond(2); t numbers their cells.
__device__ __forceinline__ void ext_rim_cell(const KP &P, int t, int store_f, int fuse_adv) {
  const int im = P.im, jm = P.jm, ncol = jm - 3;          // rows 1, 2, jm in full; columns 1, 2, im for j = 3..jmm1
  int i, j;
  if (t < 3 * im) { const int r = t / im; i = t - r * im + 1; j = r == 0 ? 1 : (r == 1 ? 2 : jm); }
  else {
    const int q = t - 3 * im;
    if (ncol <= 0 || q >= 3 * ncol) return;
    const int r = q / ncol; j = 3 + (q - r * ncol); i = r == 0 ? 1 : (r == 1 ? 2 : im);
  }
  ext_cell_general(P, i, j, store_f, fuse_adv);
}
// A ring of cells: lines 1..nlo and the last nhi ones on every side (rows in full, columns for the rows between); t numbers its
// cells.  With (nlo, nhi) = (2, 1) it is the rim; k_ext_march2 leaves (4, 2) to it.  advave is formed in place.
__device__ __forceinline__ int ext_ring_count(const KP &P, int nlo, int nhi) {
  const int nl = nlo + nhi, mid = P.jm - nl;
  return nl * P.im + (mid > 0 ? nl * mid : 0);
}
__device__ __forceinline__ void ext_ring_cell(const KP &P, int t, int store_f, int nlo, int nhi) {
  const int im = P.im, jm = P.jm, nl = nlo + nhi, mid = jm - nl;
  int i, j;
  if (t < nl * im) { const int r = t / im; i = t - r * im + 1; j = r < nlo ? r + 1 : jm - nl + r + 1; }
  else {
    const int q = t - nl * im;
    if (mid <= 0 || q >= nl * mid) return;
    const int r = q / mid; j = nlo + 1 + (q - r * mid); i = r < nlo ? r + 1 : im - nl + r + 1;
  }
  ext_cell_general(P, i, j, store_f, 1);
}
// ---------------------------------------------------------------------------------------------
// mode_external, single tile: ONE kernel per substep.  With nothing to exchange between the
// continuity, momentum and filter phases (advance.f:233, :292-293) the three kernels above fuse: elf
// at (i,j), (i-1,j), (i,j-1) is evaluated in registers, uaf/vaf follow, and the time rotation writes
// the NEXT generation of ua, va, d, el, elb into the second buffer set (the neighbours of those five
// are still being read by other threads).  elf, uaf, vaf reach memory only on the last substep
// (nothing reads them in between).  Per substep 42 array passes instead of 70.
// k_ext_step: cells 3..imm1 x 3..jmm1 (no clamp, no open-boundary formula);  k_ext_step_rim: the
// three outermost lines on every side, with bcond(1) and bcond(2).  Array padding is not touched.
// Register-operand form: every array row is loaded once per cell (50 loads instead of ~95 through the
// cell functions above), the (i-1) / (i+1) operands, the two x-fluxes of the east neighbour and the
// west neighbour's new elevation come from the neighbour lane (halo-lane wavefronts).
// rim_rows > 0: the first rim_rows block-rows of the grid do the rim cells (ext_rim_cell) instead of a launch of
// their own -- a few thousand threads on a long chain of dependent loads, ~15 us whatever the grid size, 30 times
// per internal step; as the FIRST workgroups of this launch they run beside the interior ones.  They read the
// same generation of ua, va, d, el, elb and write other cells.
// FUSE_ADV: advave (solver.f:16-121, re-evaluated every substep, advance.f:235 with ispadv = 1) rides along: its
// fluxes need the same rows of d, ua, va this kernel holds already, plus uab, vab, aam2d and four metric arrays --
// one launch and ~9 of 54 array passes per substep less than k_advave_pair + this kernel.  advua, advva reach memory
// on the last substep only.  uab, vab are then read at neighbour cells while this kernel rewrites them: they are
// double-buffered like ua, va, d, el, elb.
template <int FUSE_ADV>
__device__ __forceinline__ void ext_step_body(const KP &P, int store_f, int rim_rows) {   // the substep is P.iext
  if ((int)blockIdx.y < rim_rows) {
    ext_rim_cell(P, (int)((blockIdx.y * blockDim.y + threadIdx.y) * (gridDim.x * blockDim.x) + blockIdx.x * blockDim.x + threadIdx.x), store_f, FUSE_ADV);
    return;
  }
  const int lane = HALO_LANE, i0 = HALO_COL, j = (int)((blockIdx.y - rim_rows) * blockDim.y + threadIdx.y + 1);
  if (j < 3 || j > P.jmm1) return;                         // a whole wavefront (one row)
  const bool out = (lane >= 1 && lane <= 62 && i0 >= 3 && i0 <= P.imm1);
#ifdef POMGPU_EMU
  if (!out) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);    // halo / padding lanes shadow a valid column
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
#define WS(x, expr) halo_w(x, [&] { return expr; })
#define ES(x, expr) halo_e(x, [&] { return expr; })
  // ---- operands: rows m2 = j-2, m1 = j-1, 0 = j, p1 = j+1 of this column
  const double d_m2 = d_(i, j - 2), d_m1 = d_(i, j - 1), d_0 = d_(i, j), d_p1 = d_(i, j + 1);
  const double ua_m1 = ua_(i, j - 1), ua_0 = ua_(i, j);
  const double va_m1 = va_(i, j - 1), va_0 = va_(i, j), va_p1 = va_(i, j + 1);
  // metric sums: dy(i,q)+dy(i-1,q), dx(i,q)+dx(i,q-1) -- read as derived arrays, or (FUSE_ADV: dx, dy are loaded anyway)
  // formed from the rows of dx, dy and the west lane's values in the order k_coef_static sums them
  double dysx_m1, dysx_0, dxsy_m1, dxsy_0, dxsy_p1;
  double dx_m1 = 0., dx_0 = 0., dx_p1 = 0., dy_m1 = 0., dy_0 = 0., dy_p1 = 0., dxW_m1 = 0., dxW_0 = 0., dxW_p1 = 0., dyW_m1 = 0., dyW_0 = 0., dyW_p1 = 0.;
  if (FUSE_ADV) {
    const double dx_m2 = dx_(i, j - 2);
    dx_m1 = dx_(i, j - 1); dx_0 = dx_(i, j); dx_p1 = dx_(i, j + 1);
    dy_m1 = dy_(i, j - 1); dy_0 = dy_(i, j); dy_p1 = dy_(i, j + 1);
    dxW_m1 = WS(dx_m1, dx_(iw, j - 1)); dxW_0 = WS(dx_0, dx_(iw, j)); dxW_p1 = WS(dx_p1, dx_(iw, j + 1));
    dyW_m1 = WS(dy_m1, dy_(iw, j - 1)); dyW_p1 = WS(dy_p1, dy_(iw, j + 1));
    dyW_0 = lane_w(dy_0, [&] { return dy_(iw, j); });       // feeds ec, which the east neighbour takes as its ew: true on the west halo lane too
    dysx_m1 = dy_m1 + dyW_m1; dysx_0 = dy_0 + dyW_0;
    dxsy_m1 = dx_m1 + dx_m2; dxsy_0 = dx_0 + dx_m1; dxsy_p1 = dx_p1 + dx_0;
  } else {
    dysx_m1 = K2(DYSX, i, j - 1); dysx_0 = K2(DYSX, i, j);
    dxsy_m1 = K2(DXSY, i, j - 1); dxsy_0 = K2(DXSY, i, j); dxsy_p1 = K2(DXSY, i, j + 1);
  }
  const double elb_m1 = elb_(i, j - 1), elb_0 = elb_(i, j), el_m1 = el_(i, j - 1), el_0 = el_(i, j);
  const double art_m1 = F2(art, i, j - 1), art_0 = F2(art, i, j), vfl_m1 = F2(vfluxf, i, j - 1), vfl_0 = F2(vfluxf, i, j);
  double fsm_m1, fsm_0, dum_0 = 0., dvm_0 = 0.;
  if (FUSE_ADV) {                                          // masks from the byte array (KP.m8)
    const unsigned mk_m1 = P.m8[IX2(i, j - 1)], mk_0 = P.m8[IX2(i, j)];
    fsm_m1 = (double)(mk_m1 & 1u); fsm_0 = (double)(mk_0 & 1u); dum_0 = (double)((mk_0 >> 1) & 1u); dvm_0 = (double)((mk_0 >> 2) & 1u);
  } else { fsm_m1 = F2(fsm, i, j - 1); fsm_0 = F2(fsm, i, j); }
  const double ea_m1 = F2(e_atmos, i, j - 1), ea_0 = F2(e_atmos, i, j), h_m1 = F2(h, i, j - 1), h_0 = F2(h, i, j);
  const double cor_m1 = F2(cor, i, j - 1), cor_0 = F2(cor, i, j);
  const double adx2d = F2(adx2d, i, j), aru = F2(aru, i, j), drx2d = F2(drx2d, i, j);
  const double wusurf = F2(wusurf, i, j), wubot = F2(wubot, i, j), uab = uab_(i, j);
  const double ady2d = F2(ady2d, i, j), arv = F2(arv, i, j), dry2d = F2(dry2d, i, j);
  double advua = 0., advva = 0.;
  if (!FUSE_ADV) { advua = F2(advua, i, j); advva = F2(advva, i, j); }
  const double wvsurf = F2(wvsurf, i, j), wvbot = F2(wvbot, i, j), vab = vab_(i, j);
  // ---- neighbour-lane operands
  // d(i-1) feeds this lane's own x-flux and through it ec, which the east neighbour takes as its ew:
  // the west halo lane needs the true value too, so lane 0 loads it (lane_w) instead of shuffling
  const double dW_0 = lane_w(d_0, [&] { return d_(iw, j); }), dW_m1 = lane_w(d_m1, [&] { return d_(iw, j - 1); });
  const double corW_0 = WS(cor_0, F2(cor, iw, j)), vaW_0 = WS(va_0, va_(iw, j)), vaW_p1 = WS(va_p1, va_(iw, j + 1));
  const double elW_0 = WS(el_0, el_(iw, j)), elbW_0 = WS(elb_0, elb_(iw, j)), eaW_0 = WS(ea_0, F2(e_atmos, iw, j)), hW_0 = WS(h_0, F2(h, iw, j));
  const double uaE_0 = ES(ua_0, ua_(ie, j)), uaE_m1 = ES(ua_m1, ua_(ie, j - 1));
  // ---- continuity (advance.f:211-231) at (i,j) and (i,j-1); the west value from the neighbour lane
  const double fua_0 = .25 * (d_0 + dW_0) * dysx_0 * ua_0, fua_m1 = .25 * (d_m1 + dW_m1) * dysx_m1 * ua_m1;
  const double fuaE_0 = ES(fua_0, flux_ua(P, ie, j)), fuaE_m1 = ES(fua_m1, flux_ua(P, ie, j - 1));
  const double fva_m1 = .25 * (d_m1 + d_m2) * dxsy_m1 * va_m1, fva_0 = .25 * (d_0 + d_m1) * dxsy_0 * va_0,
               fva_p1 = .25 * (d_p1 + d_0) * dxsy_p1 * va_p1;
  const double ec = (elb_0 + P.dte2 * (-(fuaE_0 - fua_0 + fva_p1 - fva_0) / art_0 - vfl_0)) * fsm_0;
  const double es = (elb_m1 + P.dte2 * (-(fuaE_m1 - fua_m1 + fva_0 - fva_m1) / art_m1 - vfl_m1)) * fsm_m1;
  const double ew = WS(ec, elf_at(P, iw, j));
  if (FUSE_ADV) {                                          // advave on register operands, as k_advave_pair (advave_cell)
    AdvaveCell a;
    a.d_m2 = d_m2; a.d_m1 = d_m1; a.d_0 = d_0; a.d_p1 = d_p1;
    a.dW_m1 = dW_m1; a.dW_0 = dW_0; a.dW_p1 = WS(d_p1, d_(iw, j + 1)); a.dE_0 = ES(d_0, d_(ie, j));
    a.ua_m1 = ua_m1; a.ua_0 = ua_0; a.ua_p1 = ua_(i, j + 1); a.uaE_0 = uaE_0;
    a.va_m1 = va_m1; a.va_0 = va_0; a.va_p1 = va_p1; a.vaW_0 = vaW_0; a.vaW_p1 = vaW_p1;
    a.uab_m1 = uab_(i, j - 1); a.uab_0 = uab; a.uab_p1 = uab_(i, j + 1); a.uabE_0 = ES(uab, uab_(ie, j));
    a.vab_m1 = vab_(i, j - 1); a.vab_0 = vab; a.vab_p1 = vab_(i, j + 1);
    a.vabW_0 = WS(a.vab_0, vab_(iw, j)); a.vabW_p1 = WS(a.vab_p1, vab_(iw, j + 1));
    a.am_m1 = aam2d_(i, j - 1); a.am_0 = aam2d_(i, j); a.am_p1 = aam2d_(i, j + 1);
    a.amW_m1 = WS(a.am_m1, aam2d_(iw, j - 1)); a.amW_0 = WS(a.am_0, aam2d_(iw, j)); a.amW_p1 = WS(a.am_p1, aam2d_(iw, j + 1));
    a.dx_0 = dx_0; a.dx_m1 = dx_m1; a.dy_0 = dy_0; a.dy_m1 = dy_m1;
    a.DX4_0 = dx_0 + dxW_0 + dx_m1 + dxW_m1; a.DX4_p1 = dx_p1 + dxW_p1 + dx_0 + dxW_0;      // as k_coef_static: (i,j)+(i-1,j)+(i,j-1)+(i-1,j-1)
    a.DY4_0 = dy_0 + dyW_0 + dy_m1 + dyW_m1; a.DY4_p1 = dy_p1 + dyW_p1 + dy_0 + dyW_0;
    // the flux ranges of the reference: fluxua 2..imm1, the corner quantities 2..im; row j-1 >= 2 here
    const AdvaveOut f = advave_cell(a, i0 >= 2 && i0 <= P.imm1, i0 >= 2 && i0 <= P.im, true);
    const double fu_w = WS(f.fu, advave_fu(P, i - 1, j));
    const double gu_e = ES(f.gu, (i + 1 <= P.im) ? advave_gu(P, i + 1, j, advave_tps(P, i + 1, j)) : 0.);
    advua = f.fu - fu_w + f.fvP - f.fv0;                   // :65-66
    advva = gu_e - f.gu + f.gv0 - f.gvM;                   // :116-117
  }
#undef WS
#undef ES
  if (!out) return;
  // ---- momentum (:237-290), as uaf_interior / vaf_interior
  double u = adx2d + advua - aru * .25 * (cor_0 * d_0 * (va_p1 + va_0) + corW_0 * dW_0 * (vaW_p1 + vaW_0)) +
             .25 * P.grav * dysx_0 * (d_0 + dW_0) *
                 ((1. - 2. * P.alpha) * (el_0 - elW_0) + P.alpha * (elb_0 - elbW_0 + ec - ew) + ea_0 - eaW_0) +
             drx2d + aru * (wusurf - wubot);
  u = ((h_0 + elb_0 + hW_0 + elbW_0) * aru * uab - 4. * P.dte * u) / ((h_0 + ec + hW_0 + ew) * aru);
  double v = ady2d + advva + arv * .25 * (cor_0 * d_0 * (uaE_0 + ua_0) + cor_m1 * d_m1 * (uaE_m1 + ua_m1)) +
             .25 * P.grav * dxsy_0 * (d_0 + d_m1) *
                 ((1. - 2. * P.alpha) * (el_0 - el_m1) + P.alpha * (elb_0 - elb_m1 + ec - es) + ea_0 - ea_m1) +
             dry2d + arv * (wvsurf - wvbot);
  v = ((h_0 + elb_0 + h_m1 + elb_m1) * arv * vab - 4. * P.dte * v) / ((h_0 + ec + h_m1 + es) * arv);
  u = u * (FUSE_ADV ? dum_0 : F2(dum, i, j));
  v = v * (FUSE_ADV ? dvm_0 : F2(dvm, i, j));
  if (store_f) {
    F2(elf, i, j) = ec; F2(uaf, i, j) = u; F2(vaf, i, j) = v;
    if (FUSE_ADV) { F2(advua, i, j) = advua; F2(advva, i, j) = advva; }
  }
  // ---- etf weights, Asselin filter, time rotation, accumulation (:295-347), as ext_update_cell
  if (P.iext == P.isplit - 2) F2(etf, i, j) = .25 * P.smoth * ec;
  else if (P.iext == P.isplit - 1) F2(etf, i, j) = F2(etf, i, j) + .5 * (1. - .5 * P.smoth) * ec;
  else if (P.iext == P.isplit) F2(etf, i, j) = (F2(etf, i, j) + .5 * ec) * fsm_0;
  Y2(uab, i, j) = ua_0 + .5 * P.smoth * (uab - 2. * ua_0 + u);
  Y2(vab, i, j) = va_0 + .5 * P.smoth * (vab - 2. * va_0 + v);
  Y2(elb, i, j) = el_0 + .5 * P.smoth * (elb_0 - 2. * el_0 + ec);
  Y2(el, i, j) = ec;
  const double dn = h_0 + ec;
  Y2(d, i, j) = dn;
  Y2(ua, i, j) = u;
  Y2(va, i, j) = v;
  if (P.iext != P.isplit) {
    F2(egf, i, j) = F2(egf, i, j) + ec * P.ispi;
    F2(utf, i, j) = F2(utf, i, j) + u * (dn + (hW_0 + ew)) * P.isp2i;
    F2(vtf, i, j) = F2(vtf, i, j) + v * (dn + (h_m1 + es)) * P.isp2i;
  }
}
template <int FUSE_ADV>
__global__ void __launch_bounds__(256) k_ext_step(KP P, int store_f, int rim_rows) { ext_step_body<FUSE_ADV>(P, store_f, rim_rows); }

// art = dx*dy; aru, arv = .25*(dx+dx')*(dy+dy') (initialize.f:361-367): true of the arrays the host handed over?  *flag stays
// non-zero if every cell agrees bit for bit.  The arrays are the caller's: the formulas are only USED after this check.
__global__ void k_check_areas(KP P, int *flag) {
  const int i = TID_I, j = TID_J;
  if (i > P.im || j > P.jm) return;
  bool ok = F2(art, i, j) == dx_(i, j) * dy_(i, j);
  if (i >= 2 && j >= 2) {
    ok = ok && F2(aru, i, j) == .25 * (dx_(i, j) + dx_(i - 1, j)) * (dy_(i, j) + dy_(i - 1, j));
    ok = ok && F2(arv, i, j) == .25 * (dx_(i, j) + dx_(i, j - 1)) * (dy_(i, j) + dy_(i, j - 1));
  }
  if (!ok) *flag = 0;
}
void launch_check_areas(pomgpu_ctx *c) {
  (void)hipMemsetAsync(c->d_areas, 1, sizeof(int), c->cur);   // 0x01010101: "canonical" until a cell disagrees
  LAUNCH(c, k_check_areas, grid2(c->P), blk2(), c->P, c->d_areas);
}
// ---- the same substep (advave fused), MARCHING DOWN THE ROWS: large tiles ----------------------------------------------
// k_ext_step<1> is one wavefront per row segment of 62 cells: 58 loads, then ~650 fp64 instructions that all wait for them,
// two wavefronts per SIMD (251 VGPRs).  The counters (2048x1536): a wavefront lives ~8.5 us and waits 57 % of it; the kernel
// is 51 such lifetimes per SIMD / 2 -- bound by that latency, and every stencil row is fetched 1.75 times.  Here a wavefront
// owns `rows` consecutive rows of its 62 columns and walks down them like the column kernels walk down the levels:
//  * the operands of rows j-1, j, j+1 stay in registers; a row costs ONE new row of the eight stencil arrays (requested an
//    iteration ahead) + the 18 pointwise operands of row j (requested at the top of the iteration, used in its second half),
//    26 + 6 loads instead of 58 + 6;
//  * what row j+1 recomputed from the same operands is carried instead: the new elevation of row j-1 (es), fluxva(j), and
//    advave's tps / fluxva (both halves) of the row above -- the SAME expressions on the SAME operands, so the bits do not
//    change;
//  * no load or store sits in a branch (vmcnt bookkeeping, as in the level loops): lanes / substeps with nothing to move aim
//    outside the buffer; lane 0's two true western operands are one-lane loads (every other lane aims outside).
// A segment starts one row early: a warm-up row, peeled at compile time (row_step<false>: continuity and three advave terms,
// no momentum, no stores), fills the carried values.  Workgroup order: XCD bands (below).  Cell areas: k_check_areas.
struct ExtRow { double d, ua, va, dx, dy, uab, vab, am, dW, dyW; };   // one row of the stencil operands; dW, dyW: column i-1
#ifndef POMGPU_EMU
// the western neighbour's value, true on lane 0 as well (w: the one-lane load).  The shift is made by ALL lanes before the
// select: inside the arm of a ?: it would run with lane 0 switched off, and lane 1 would read a disabled lane
__device__ __forceinline__ double west_true(double x, double w, int lane) {
  const double t = wave_up1(x);
  return lane == 0 ? w : t;
}
#endif
__global__ void __launch_bounds__(256) k_ext_march(KP P, int store_f, int rim_wgs, int rows, const int *areas, int use_areas) {
  const int gx = (int)(blockIdx.x * blockDim.x + threadIdx.x), lane = gx & 63, wg = gx >> 6;   // (the host emulation runs lanes as blocks of width 1)
  if (wg < rim_wgs) { ext_rim_cell(P, wg * 256 + (int)threadIdx.y * 64 + lane, store_f, 1); return; }
  // Workgroups are dealt to the XCDs round-robin (linear id & 7; rim_wgs is a multiple of 8): XCD x owns a band of segment
  // groups and walks it row by row, so that the wavefronts to the left and right of this one -- whose 64 columns overlap its
  // own by two, and whose 512-byte row pieces share a 128-byte line with it (a piece starts every 496 bytes: five lines per
  // load, not four) -- sit behind the same L2
  const int L = wg - rim_wgs, nbx = (P.iml + 61) / 62;
  const int nseg_ = (P.jmm1 - 3 + 1 + rows - 1) / rows, ngrp = (nseg_ + (int)blockDim.y - 1) / (int)blockDim.y, gpx = (ngrp + 7) / 8;
  const int m_ = L >> 3, grp = (L & 7) * gpx + m_ / nbx;
  if (grp >= ngrp) return;
  const int bx = m_ % nbx, seg = grp * (int)blockDim.y + (int)threadIdx.y;
  const int j0 = 3 + seg * rows;
  if (j0 > P.jmm1) return;                                 // a whole wavefront
  const int j1 = (j0 + rows - 1 < P.jmm1) ? j0 + rows - 1 : P.jmm1;
  const int i0 = bx * 62 + lane;
  const bool out = (lane >= 1 && lane <= 62 && i0 >= 3 && i0 <= P.imm1);
#ifdef POMGPU_EMU
  if (!out) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);    // halo / padding lanes shadow a valid column
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const unsigned vo = (unsigned)(i - 1) * 8u;
#ifdef POMGPU_EMU
  const unsigned voW = (unsigned)(iw - 1) * 8u;
#define WTRUE(x, w) (w)
#else
  const unsigned voW = (lane == 0) ? (unsigned)(iw - 1) * 8u : BOFF_NONE;
#define WTRUE(x, w) west_true((x), (w), lane)
#endif
#define RO(row) ((unsigned)WAVE_UNIFORM((row) - 1) * (unsigned)P.iml * 8u)
#define XD(g) buf2_of(P.x2[g], P.n2)
#define YD(g) buf2_of(P.y2[g], P.n2)
// every blk2d array through ONE descriptor (the block is contiguous; < 4 GiB checked by the launcher): the array's offset
// rides in the scalar offset -- a descriptor per array (4 SGPRs each, ~30 of them) spilled 220 SGPRs
#define FLD(name, voff, ro_) bld2(b2d, (voff), (unsigned)(P2_##name * P.n2 * 8u) + (ro_))
#define FST(name, voff, ro_, x) bst2(b2d, (voff), (unsigned)(P2_##name * P.n2 * 8u) + (ro_), (x))
#define WS(x, expr) halo_w(x, [&] { return expr; })
#define ES(x, expr) halo_e(x, [&] { return expr; })
  const BufA b2d = buf2_of(P.b2, (size_t)POM_NBLK2D * P.n2);
  auto load_row = [&](int row) {
    ExtRow r;
    const unsigned ro = RO(row);
    r.d = bld2(XD(X2_d), vo, ro); r.ua = bld2(XD(X2_ua), vo, ro); r.va = bld2(XD(X2_va), vo, ro);
    r.dx = FLD(dx, vo, ro); r.dy = FLD(dy, vo, ro);
    r.uab = bld2(XD(X2_uab), vo, ro); r.vab = bld2(XD(X2_vab), vo, ro); r.am = FLD(aam2d, vo, ro);
    r.dW = bld2(XD(X2_d), voW, ro); r.dyW = FLD(dy, voW, ro);
    return r;
  };
  // uniform conditions of the substep (advance.f:295-347) as store / load offsets
  // art, aru, arv from dx, dy where k_check_areas found the arrays to be exactly that (else: read them)
  const bool canon = use_areas && WAVE_UNIFORM(*areas) != 0;
  const unsigned o_area = canon ? BOFF_NONE : vo;
  const bool acc = (P.iext != P.isplit);
  const unsigned o_etf_ld = (P.iext >= P.isplit - 1) ? vo : BOFF_NONE, o_acc_ld = acc ? vo : BOFF_NONE;
  // rows jw-1, jw, jw+1 (jw = j0-1: the warm-up row)
  const int jw = j0 - 1;
  ExtRow rm = load_row(jw - 1), rc = load_row(jw), rp = load_row(jw + 1);
  rm.dW = WTRUE(rm.d, rm.dW); rm.dyW = WTRUE(rm.dy, rm.dyW);
  rc.dW = WTRUE(rc.d, rc.dW); rc.dyW = WTRUE(rc.dy, rc.dyW);
  rp.dW = WTRUE(rp.d, rp.dW); rp.dyW = WTRUE(rp.dy, rp.dyW);
  // pointwise operands of row jw-1 that the momentum equations read at j-1: filled by the warm-up row itself (rotation below)
  double el_m1 = 0., elb_m1 = 0., ea_m1 = 0., h_m1 = 0., cor_m1 = 0.;
  // carried: only fluxva(jw) must be right before the warm-up row; the others are produced by it
  double fva_0 = .25 * (rc.d + rm.d) * (rc.dx + rm.dx) * rc.va;
  double es = 0., tps0 = 0., fv0 = 0., gvM = 0.;
  const bool fu_on = (i0 >= 2 && i0 <= P.imm1), w_on = (i0 >= 2 && i0 <= P.im);
  // one row; LIVE = false: the warm-up row -- only what the carried values need (no momentum, no stores, 9 of the 22 pointwise loads)
  auto row_step = [&](const int j, auto live_tag) {
    constexpr bool LIVE = decltype(live_tag)::value;
    const unsigned ro = RO(j);
    // ---- requests: the pointwise operands of this row first (used below), then the stencil row of the next iteration
    const double el_0 = bld2(XD(X2_el), vo, ro), elb_0 = bld2(XD(X2_elb), vo, ro);
    const double art_l = FLD(art, o_area, ro), vfl_0 = FLD(vfluxf, vo, ro);
    const unsigned mk_0 = P.m8[(size_t)WAVE_UNIFORM(j - 1) * (size_t)P.iml + (size_t)(i - 1)];
    const double ea_0 = FLD(e_atmos, vo, ro), h_0 = FLD(h, vo, ro), cor_0 = FLD(cor, vo, ro);
    double adx2d = 0., aru = 0., drx2d = 0., wusurf = 0., wubot = 0., ady2d = 0., arv = 0., dry2d = 0., wvsurf = 0., wvbot = 0.;
    double etf_o = 0., egf_o = 0., utf_o = 0., vtf_o = 0.;
    if (LIVE) {
      adx2d = FLD(adx2d, vo, ro); aru = FLD(aru, o_area, ro); drx2d = FLD(drx2d, vo, ro);
      wusurf = FLD(wusurf, vo, ro); wubot = FLD(wubot, vo, ro);
      ady2d = FLD(ady2d, vo, ro); arv = FLD(arv, o_area, ro); dry2d = FLD(dry2d, vo, ro);
      wvsurf = FLD(wvsurf, vo, ro); wvbot = FLD(wvbot, vo, ro);
      etf_o = FLD(etf, o_etf_ld, ro); egf_o = FLD(egf, o_acc_ld, ro); utf_o = FLD(utf, o_acc_ld, ro); vtf_o = FLD(vtf, o_acc_ld, ro);
    }
    ExtRow rn = load_row(j + 2 <= P.jml ? j + 2 : P.jml);
    const double fsm_0 = (double)(mk_0 & 1u), dum_0 = (double)((mk_0 >> 1) & 1u), dvm_0 = (double)((mk_0 >> 2) & 1u);
    const unsigned o_st = out ? vo : BOFF_NONE;
    // ---- neighbour-lane operands (rows m1 = j-1: rm, 0 = j: rc, p1 = j+1: rp)
    const double dxW_0 = WS(rc.dx, dx_(iw, j)), dxW_p1 = WS(rp.dx, dx_(iw, j + 1));
    const double corW_0 = WS(cor_0, F2(cor, iw, j)), vaW_0 = WS(rc.va, va_(iw, j)), vaW_p1 = WS(rp.va, va_(iw, j + 1));
    const double elW_0 = WS(el_0, el_(iw, j)), elbW_0 = WS(elb_0, elb_(iw, j)), eaW_0 = WS(ea_0, F2(e_atmos, iw, j)), hW_0 = WS(h_0, F2(h, iw, j));
    const double uaE_0 = ES(rc.ua, ua_(ie, j)), uaE_m1 = ES(rm.ua, ua_(ie, j - 1));
    // ---- continuity (advance.f:211-231) at (i,j); (i,j-1) is the row above's, the west value the neighbour lane's
    const double dysx_0 = rc.dy + rc.dyW, dxsy_0 = rc.dx + rm.dx, dxsy_p1 = rp.dx + rc.dx;
    const double art_0 = canon ? rc.dx * rc.dy : art_l;                                      // initialize.f:361
    if (canon) { aru = .25 * (rc.dx + dxW_0) * dysx_0; arv = .25 * dxsy_0 * (rc.dy + rm.dy); }   // :366-367
    const double fua_0 = .25 * (rc.d + rc.dW) * dysx_0 * rc.ua;
    const double fuaE_0 = ES(fua_0, flux_ua(P, ie, j));
    const double fva_p1 = .25 * (rp.d + rc.d) * dxsy_p1 * rp.va;
    const double ec = (elb_0 + P.dte2 * (-(fuaE_0 - fua_0 + fva_p1 - fva_0) / art_0 - vfl_0)) * fsm_0;
    const double ew = WS(ec, elf_at(P, iw, j));
    // ---- advave (solver.f:16-121) on register operands, as advave_cell; tps0, fv0, gvM are the row above's tpsP, fvP, gv0
    double fu = 0., gu = 0., fvP = 0., gv0 = 0., tpsP = 0.;
    {
      const double dE_0 = ES(rc.d, d_(ie, j)), uabE_0 = ES(rc.uab, uab_(ie, j));
      const double vabW_p1 = WS(rp.vab, vab_(iw, j + 1));
      const double amW_0 = WS(rc.am, aam2d_(iw, j)), amW_p1 = WS(rp.am, aam2d_(iw, j + 1));
      const double DY4_0 = rc.dy + rc.dyW + rm.dy + rm.dyW, DY4_p1 = rp.dy + rp.dyW + rc.dy + rc.dyW;
      const double DX4_p1 = rp.dx + dxW_p1 + rc.dx + dxW_0;
      if (LIVE && fu_on) {
        double f = .125 * ((dE_0 + rc.d) * uaE_0 + (rc.d + rc.dW) * rc.ua) * (uaE_0 + rc.ua);
        f = f - rc.d * 2. * rc.am * (uabE_0 - rc.uab) / rc.dx;
        fu = f * rc.dy;
      }
      if (w_on) {
        tpsP = .25 * (rp.d + rp.dW + rc.d + rc.dW) * (rp.am + rc.am + amW_p1 + amW_0) *
               ((rp.uab - rc.uab) / DY4_p1 + (rp.vab - vabW_p1) / DX4_p1);
        if (LIVE) {
          const double g = .125 * ((rc.d + rc.dW) * rc.ua + (rm.d + rm.dW) * rm.ua) * (vaW_0 + rc.va);
          gu = (g - tps0) * .25 * DY4_0;
        }
        const double fP = .125 * ((rp.d + rc.d) * rp.va + (rp.dW + rc.dW) * vaW_p1) * (rp.ua + rc.ua);
        fvP = (fP - tpsP) * .25 * DX4_p1;
        double gv = .125 * ((rp.d + rc.d) * rp.va + (rc.d + rm.d) * rc.va) * (rp.va + rc.va);
        gv = gv - rc.d * 2. * rc.am * (rp.vab - rc.vab) / rc.dy;
        gv0 = gv * rc.dx;
      }
    }
    if (LIVE) {
      const double fu_w = WS(fu, advave_fu(P, i - 1, j));
      const double gu_e = ES(gu, (i + 1 <= P.im) ? advave_gu(P, i + 1, j, advave_tps(P, i + 1, j)) : 0.);
      const double advua = fu - fu_w + fvP - fv0;             // :65-66
      const double advva = gu_e - gu + gv0 - gvM;             // :116-117
      // ---- momentum (:237-290), as uaf_interior / vaf_interior
      double u = adx2d + advua - aru * .25 * (cor_0 * rc.d * (rp.va + rc.va) + corW_0 * rc.dW * (vaW_p1 + vaW_0)) +
                 .25 * P.grav * dysx_0 * (rc.d + rc.dW) *
                     ((1. - 2. * P.alpha) * (el_0 - elW_0) + P.alpha * (elb_0 - elbW_0 + ec - ew) + ea_0 - eaW_0) +
                 drx2d + aru * (wusurf - wubot);
      u = ((h_0 + elb_0 + hW_0 + elbW_0) * aru * rc.uab - 4. * P.dte * u) / ((h_0 + ec + hW_0 + ew) * aru);
      double v = ady2d + advva + arv * .25 * (cor_0 * rc.d * (uaE_0 + rc.ua) + cor_m1 * rm.d * (uaE_m1 + rm.ua)) +
                 .25 * P.grav * dxsy_0 * (rc.d + rm.d) *
                     ((1. - 2. * P.alpha) * (el_0 - el_m1) + P.alpha * (elb_0 - elb_m1 + ec - es) + ea_0 - ea_m1) +
                 dry2d + arv * (wvsurf - wvbot);
      v = ((h_0 + elb_0 + h_m1 + elb_m1) * arv * rc.vab - 4. * P.dte * v) / ((h_0 + ec + h_m1 + es) * arv);
      u = u * dum_0;
      v = v * dvm_0;
      // ---- stores: the new generation, etf weights, accumulation (:295-347), as ext_update_cell
      {
        const unsigned o_f = store_f ? o_st : BOFF_NONE;
        FST(elf, o_f, ro, ec); FST(uaf, o_f, ro, u); FST(vaf, o_f, ro, v);
        FST(advua, o_f, ro, advua); FST(advva, o_f, ro, advva);
        double etf_n = 0.;
        if (P.iext == P.isplit - 2) etf_n = .25 * P.smoth * ec;
        else if (P.iext == P.isplit - 1) etf_n = etf_o + .5 * (1. - .5 * P.smoth) * ec;
        else if (P.iext == P.isplit) etf_n = (etf_o + .5 * ec) * fsm_0;
        FST(etf, (P.iext >= P.isplit - 2) ? o_st : BOFF_NONE, ro, etf_n);
        bst2(YD(X2_uab), o_st, ro, rc.ua + .5 * P.smoth * (rc.uab - 2. * rc.ua + u));
        bst2(YD(X2_vab), o_st, ro, rc.va + .5 * P.smoth * (rc.vab - 2. * rc.va + v));
        bst2(YD(X2_elb), o_st, ro, el_0 + .5 * P.smoth * (elb_0 - 2. * el_0 + ec));
        bst2(YD(X2_el), o_st, ro, ec);
        const double dn = h_0 + ec;
        bst2(YD(X2_d), o_st, ro, dn);
        bst2(YD(X2_ua), o_st, ro, u);
        bst2(YD(X2_va), o_st, ro, v);
        const unsigned o_acc = acc ? o_st : BOFF_NONE;
        FST(egf, o_acc, ro, egf_o + ec * P.ispi);
        FST(utf, o_acc, ro, utf_o + u * (dn + (hW_0 + ew)) * P.isp2i);
        FST(vtf, o_acc, ro, vtf_o + v * (dn + (h_m1 + es)) * P.isp2i);
      }
    }
    // ---- one row down
    fva_0 = fva_p1; es = ec;
    tps0 = tpsP; fv0 = fvP; gvM = gv0;
    el_m1 = el_0; elb_m1 = elb_0; ea_m1 = ea_0; h_m1 = h_0; cor_m1 = cor_0;
    rm = rc; rc = rp;
    rn.dW = WTRUE(rn.d, rn.dW); rn.dyW = WTRUE(rn.dy, rn.dyW);
    rp = rn;
  };
  row_step(jw, std::false_type());
  for (int j = j0; j <= j1; j++) row_step(j, std::true_type());
#undef WTRUE
#undef RO
#undef XD
#undef YD
#undef FLD
#undef FST
#undef WS
#undef ES
}

// ---- TWO substeps per pass over memory (large tiles) --------------------------------------------------------------------
// k_ext_march moves 1.1-1.2 GB per substep at ~5.6 TB/s: it sits on the practical HBM ceiling, only fewer bytes help.  Of
// the ~36 array passes of a substep 16 are operands that do not change during the external loop (metrics, depth, Coriolis,
// the vertical integrals, surface / bottom stress ...), 7 are the generation it reads and 7 the generation it writes -- which
// the next substep reads again.  Here a wavefront that marches down its rows carries a SECOND generation one row behind the
// first: substep n is evaluated for row t from the operands in memory (generation X) exactly as k_ext_march does it, its
// results (generation Y: ua, va, d, el, elb, uab, vab of row t) stay in registers, and substep n+1 is evaluated for row t-1
// from the Y rows t-2, t-1, t and the same static operands; only its results (generation Z) and the accumulators with BOTH
// contributions are stored.  Per pair of substeps the statics and X are read once and Y never touches memory: ~38 passes
// instead of ~72.  The arithmetic of a row is ONE function (ext_row_math) used by both generations: the same expressions on
// the same operands in the same order as k_ext_step / k_ext_march, so the bits do not change.
//  * Columns: substep n+1 at column i reads Y at i-2 .. i+1.  Of a wavefront's 64 lanes the first substep is right on lanes
//    1..62 (d and the elevation on 0..62, thanks to lane 0's two true western operands), the second on lanes 2..61: a
//    wavefront owns 60 columns.
//  * Rows: a segment of `rows` rows of substep n+1 (j0..j1) needs Y rows j0-2 .. j1+1, i.e. substep n on rows+3 rows and a
//    warm-up row: taller segments than k_ext_march's (the redundant part is (rows + 4.5) / rows per pair).
//  * Ring: the lines 1..4 and the last two on every side are not marched (substep n+1 at line 3 would need Y of the rim lines
//    1, 2 with their boundary conditions).  ext_ring_cell evaluates BOTH substeps there by the general cell formulas: substep
//    n (X -> a third buffer set T) as the first workgroups of this kernel's own launch, substep n+1 (T -> Z) as a small launch
//    after it (k_ext_ring), for which this kernel stores the Y cells of the two lines next to the ring into T.  The marched
//    cells need nothing from the ring: Y of lines 3, 4 are plain interior cells of substep n, evaluated here from X.
//  * One wave per SIMD (~400 registers): 4 wavefronts per CU with ~50 loads in flight each keep the memory pipes busy; the
//    kernel is then bound about evenly by the issue of its ~1300 instructions per row pair and by its traffic.
#ifndef POMGPU_EMU
struct XRow { double d, ua, va, uab, vab, dW; };            // one row of the time-dependent stencil operands; dW: d of column i-1
struct SRow { double dx, dy, am, dyW; };                    // one row of the static stencil operands; dyW: dy of column i-1
struct PRow { double el, elb; };                            // the two time-dependent operands read at (i, i-1) x (j, j-1) only
struct SPt {                                                // static pointwise operands of one row
  double vfl, ea, h, cor, adx2d, aru, drx2d, wusurf, wubot, ady2d, arv, dry2d, wvsurf, wvbot, art;
  unsigned mk;
};
struct SPm { double ea, h, cor; };                          // ... what the row below still needs of them
struct ECarry { double fva_0, es, tps0, fv0, gvM, el_m1, elb_m1; };   // what row j hands to row j+1 (one set per generation)
struct EOut { double ec, ew, hW, esm, u, v, advua, advva, uab_n, vab_n, elb_n, dn; };
// One row of one substep on register operands: rows m = j-1 (rm, sm), 0 = j (rc, sc), p = j+1 (rp, sp); qm: e_atmos, h, cor
// of row j-1.  LIVE = false: the warm-up row of a segment -- continuity and the three advave terms row j+1 takes over, no
// momentum.  Every lane must call it (neighbour-lane shifts); as row_step of k_ext_march, operand for operand.
template <bool LIVE>
__device__ __forceinline__ void ext_row_math(const KP &P, const bool canon, const bool fu_on, const bool w_on, const XRow &rm, const XRow &rc,
                                             const XRow &rp, const SRow &sm, const SRow &sc, const SRow &sp, const PRow &pc, const SPt &q,
                                             const SPm &qm, ECarry &cy, EOut &o) {
  const double fsm_0 = (double)(q.mk & 1u), dum_0 = (double)((q.mk >> 1) & 1u), dvm_0 = (double)((q.mk >> 2) & 1u);
  const double el_0 = pc.el, elb_0 = pc.elb;
  // ---- neighbour-lane operands
  const double dxW_0 = wave_up1(sc.dx), dxW_p1 = wave_up1(sp.dx);
  const double corW_0 = wave_up1(q.cor), vaW_0 = wave_up1(rc.va), vaW_p1 = wave_up1(rp.va);
  const double elW_0 = wave_up1(el_0), elbW_0 = wave_up1(elb_0), eaW_0 = wave_up1(q.ea), hW_0 = wave_up1(q.h);
  const double uaE_0 = wave_dn1(rc.ua), uaE_m1 = wave_dn1(rm.ua);
  // ---- continuity (advance.f:211-231) at (i,j); (i,j-1) is the row above's, the west value the neighbour lane's
  const double dysx_0 = sc.dy + sc.dyW, dxsy_0 = sc.dx + sm.dx, dxsy_p1 = sp.dx + sc.dx;
  const double art_0 = canon ? sc.dx * sc.dy : q.art;                                        // initialize.f:361
  double aru = q.aru, arv = q.arv;
  if (canon) { aru = .25 * (sc.dx + dxW_0) * dysx_0; arv = .25 * dxsy_0 * (sc.dy + sm.dy); }   // :366-367
  const double fua_0 = .25 * (rc.d + rc.dW) * dysx_0 * rc.ua;
  const double fuaE_0 = wave_dn1(fua_0);
  const double fva_p1 = .25 * (rp.d + rc.d) * dxsy_p1 * rp.va;
  const double ec = (elb_0 + P.dte2 * (-(fuaE_0 - fua_0 + fva_p1 - cy.fva_0) / art_0 - q.vfl)) * fsm_0;
  const double ew = wave_up1(ec);
  // ---- advave (solver.f:16-121) on register operands; tps0, fv0, gvM are the row above's tpsP, fvP, gv0
  double fu = 0., gu = 0., fvP = 0., gv0 = 0., tpsP = 0.;
  {
    const double dE_0 = wave_dn1(rc.d), uabE_0 = wave_dn1(rc.uab);
    const double vabW_p1 = wave_up1(rp.vab);
    const double amW_0 = wave_up1(sc.am), amW_p1 = wave_up1(sp.am);
    const double DY4_0 = sc.dy + sc.dyW + sm.dy + sm.dyW, DY4_p1 = sp.dy + sp.dyW + sc.dy + sc.dyW;
    const double DX4_p1 = sp.dx + dxW_p1 + sc.dx + dxW_0;
    if (LIVE && fu_on) {
      double f = .125 * ((dE_0 + rc.d) * uaE_0 + (rc.d + rc.dW) * rc.ua) * (uaE_0 + rc.ua);
      f = f - rc.d * 2. * sc.am * (uabE_0 - rc.uab) / sc.dx;
      fu = f * sc.dy;
    }
    if (w_on) {
      tpsP = .25 * (rp.d + rp.dW + rc.d + rc.dW) * (sp.am + sc.am + amW_p1 + amW_0) *
             ((rp.uab - rc.uab) / DY4_p1 + (rp.vab - vabW_p1) / DX4_p1);
      if (LIVE) {
        const double g = .125 * ((rc.d + rc.dW) * rc.ua + (rm.d + rm.dW) * rm.ua) * (vaW_0 + rc.va);
        gu = (g - cy.tps0) * .25 * DY4_0;
      }
      const double fP = .125 * ((rp.d + rc.d) * rp.va + (rp.dW + rc.dW) * vaW_p1) * (rp.ua + rc.ua);
      fvP = (fP - tpsP) * .25 * DX4_p1;
      double gv = .125 * ((rp.d + rc.d) * rp.va + (rc.d + rm.d) * rc.va) * (rp.va + rc.va);
      gv = gv - rc.d * 2. * sc.am * (rp.vab - rc.vab) / sc.dy;
      gv0 = gv * sc.dx;
    }
  }
  o.ec = ec; o.ew = ew; o.hW = hW_0; o.esm = cy.es;
  if (LIVE) {
    const double fu_w = wave_up1(fu);
    const double gu_e = wave_dn1(gu);
    const double advua = fu - fu_w + fvP - cy.fv0;            // :65-66
    const double advva = gu_e - gu + gv0 - cy.gvM;            // :116-117
    // ---- momentum (advance.f:237-290)
    double u = q.adx2d + advua - aru * .25 * (q.cor * rc.d * (rp.va + rc.va) + corW_0 * rc.dW * (vaW_p1 + vaW_0)) +
               .25 * P.grav * dysx_0 * (rc.d + rc.dW) *
                   ((1. - 2. * P.alpha) * (el_0 - elW_0) + P.alpha * (elb_0 - elbW_0 + ec - ew) + q.ea - eaW_0) +
               q.drx2d + aru * (q.wusurf - q.wubot);
    u = ((q.h + elb_0 + hW_0 + elbW_0) * aru * rc.uab - 4. * P.dte * u) / ((q.h + ec + hW_0 + ew) * aru);
    double v = q.ady2d + advva + arv * .25 * (q.cor * rc.d * (uaE_0 + rc.ua) + qm.cor * rm.d * (uaE_m1 + rm.ua)) +
               .25 * P.grav * dxsy_0 * (rc.d + rm.d) *
                   ((1. - 2. * P.alpha) * (el_0 - cy.el_m1) + P.alpha * (elb_0 - cy.elb_m1 + ec - cy.es) + q.ea - qm.ea) +
               q.dry2d + arv * (q.wvsurf - q.wvbot);
    v = ((q.h + elb_0 + qm.h + cy.elb_m1) * arv * rc.vab - 4. * P.dte * v) / ((q.h + ec + qm.h + cy.es) * arv);
    u = u * dum_0;
    v = v * dvm_0;
    o.u = u; o.v = v; o.advua = advua; o.advva = advva;
    // ---- the next generation of this cell: Asselin filter and time rotation (:321-330)
    o.uab_n = rc.ua + .5 * P.smoth * (rc.uab - 2. * rc.ua + u);
    o.vab_n = rc.va + .5 * P.smoth * (rc.vab - 2. * rc.va + v);
    o.elb_n = el_0 + .5 * P.smoth * (elb_0 - 2. * el_0 + ec);
    o.dn = q.h + ec;
  }
  // ---- one row down
  cy.fva_0 = fva_p1; cy.es = ec;
  cy.tps0 = tpsP; cy.fv0 = fvP; cy.gvM = gv0;
  cy.el_m1 = el_0; cy.elb_m1 = elb_0;
}
struct Gen7 { double *p[POMGPU_NGEN]; };
// One generation of ua, va, d, el, elb, uab, vab behind ONE buffer descriptor: array g starts off[g] bytes into `base` (the blk2d block
// with the arrays' slot offsets, or a buffer set of its own).  A descriptor per array (4 SGPRs each, 21 of them for three generations)
// spilled ~180 SGPRs in k_ext_march2 -- ~200 v_readlane / v_writelane / s_mov per row.
struct GenD { const double *base; unsigned bytes; unsigned off[POMGPU_NGEN]; };
// etf of one substep (advance.f:295-318): e = the value before, ec the substep's new elevation
__device__ __forceinline__ double etf_rule(const KP &P, int iext, double e, double ec, double fsm) {
  if (iext == P.isplit - 2) return .25 * P.smoth * ec;
  if (iext == P.isplit - 1) return e + .5 * (1. - .5 * P.smoth) * ec;
  if (iext == P.isplit) return (e + .5 * ec) * fsm;
  return e;
}
struct YRow { XRow x; PRow p; };
struct EAcc { double egf, utf, vtf, etf; };
// The cells this kernel leaves to k_ext_ring (both substeps): lines 1..RING_LO and the last RING_HI ones on every side.  Substep
// n+1 at line 5 reads Y of lines 3, 4 -- plain interior cells of substep n, evaluated here from X (rows 1, 2 of X are in memory) --
// so this kernel needs nothing from the ring: its launch carries the ring's substep n as its first workgroups.
#define RING_LO 4
#define RING_HI 2
// P.iext = n (the first substep of the pair), P.x2 = generation X, P.y2 = generation Z; T: generation Y next to the ring, for
// the ring's second substep
__global__ void __launch_bounds__(256) k_ext_march2(KP P, Gen7 T, GenD GX, GenD GZ, GenD GT, int store_f2, int ring_wgs, int rows, const int *areas, int use_areas, int ring_first) {
  const int gx = (int)(blockIdx.x * blockDim.x + threadIdx.x), lane = gx & 63, wg = gx >> 6;
  // The ring's FIRST substep (X -> T) beside the marching workgroups -- as the LAST workgroups of the grid.  The marching grid is one
  // round of workgroups that each need a whole CU (one wave per SIMD, ~400 registers): dispatched first, the ring's small workgroups
  // landed on ~50 CUs whose marching workgroup then started a ring-workgroup's lifetime (~10 us) late, and the launch ended that much
  // later; dispatched last they share the few CUs the march leaves free (the launcher keeps some).
  const int nmw = (int)gridDim.x - ring_wgs;
  if (ring_first ? wg < ring_wgs : wg >= nmw) {
    KP R = P;
#pragma unroll
    for (int g = 0; g < POMGPU_NGEN; g++) R.y2[g] = T.p[g];
    ext_ring_cell(R, (ring_first ? wg : wg - nmw) * 256 + (int)threadIdx.y * 64 + lane, 0, RING_LO, RING_HI);
    return;
  }
  const int L = ring_first ? wg - ring_wgs : wg;
  // A workgroup = four ADJACENT wavefronts of one segment (not four segments of one column block as in k_ext_march): the segments
  // are tall here (one round of workgroups: ~55 rows at 2048x1536), and four wavefronts a megabyte apart in each of ~30 arrays
  // cycle through more 2-MiB pages than a CU's first-level TLB holds (k_profq's lesson, profiles/round2_tlb_profq.txt) -- side by
  // side they read the same rows of the same pages.  Workgroups in segment-major order: consecutive ones (dealt to the eight XCDs
  // in turn) are neighbours along the row.
  const int jlo = RING_LO + 1, jhi = P.jm - RING_HI, ilo = RING_LO + 1, ihi = P.im - RING_HI;      // the cells of substep n+1 marched here
  const int nbx = (ihi - 2) / 60 + 1, ncg = (nbx + (int)blockDim.y - 1) / (int)blockDim.y;
  const int nseg_ = (jhi - jlo + 1 + rows - 1) / rows;
  const int seg = L / ncg, bx = (L % ncg) * (int)blockDim.y + (int)threadIdx.y;
  if (seg >= nseg_ || bx >= nbx) return;                      // a whole wavefront
  const int j0 = jlo + seg * rows;
  const int j1 = (j0 + rows - 1 < jhi) ? j0 + rows - 1 : jhi;
  const int i0 = bx * 60 + lane;
  const bool own = (lane >= 2 && lane <= 61);               // the 60 columns of this wavefront
  const bool out2 = own && i0 >= ilo && i0 <= ihi;
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);    // halo / padding lanes shadow a valid column
  const int iw = i > 1 ? i - 1 : 1;
  const unsigned vo = (unsigned)(i - 1) * 8u;
  const unsigned voW = (lane == 0) ? (unsigned)(iw - 1) * 8u : BOFF_NONE;
  const unsigned o_st = out2 ? vo : BOFF_NONE;
  // generation Y goes to T where the ring's second substep will read it: two lines next to the ring
  const bool bandcol = (i0 > RING_LO && i0 <= RING_LO + 2) || (i0 >= P.im - RING_HI - 1 && i0 <= P.im - RING_HI);
#define RO(row) ((unsigned)WAVE_UNIFORM((row) - 1) * (unsigned)P.iml * 8u)
#define XLD(g, voff, ro_) bld2(bX, (voff), GX.off[g] + (ro_))
#define ZST(g, voff, ro_, x) bst2(bZ, (voff), GZ.off[g] + (ro_), (x))
#define TST(g, voff, ro_, x) bst2(bT, (voff), GT.off[g] + (ro_), (x))
#define FLD(name, voff, ro_) bld2(b2d, (voff), (unsigned)(P2_##name * P.n2 * 8u) + (ro_))
#define FST(name, voff, ro_, x) bst2(b2d, (voff), (unsigned)(P2_##name * P.n2 * 8u) + (ro_), (x))
  const BufA b2d = buf2_of(P.b2, (size_t)POM_NBLK2D * P.n2);
  BufA bX, bZ, bT;
  bX.r = __builtin_amdgcn_make_buffer_rsrc((void *)GX.base, 0, GX.bytes, 0x00020000);
  bZ.r = __builtin_amdgcn_make_buffer_rsrc((void *)GZ.base, 0, GZ.bytes, 0x00020000);
  bT.r = __builtin_amdgcn_make_buffer_rsrc((void *)GT.base, 0, GT.bytes, 0x00020000);
  const bool canon = use_areas && WAVE_UNIFORM(*areas) != 0;
  const unsigned o_area = canon ? BOFF_NONE : vo;
  const int n1 = P.iext, n2 = P.iext + 1;                   // the two substeps; the first one always accumulates (n1 < isplit)
  const bool acc2 = (n2 != P.isplit);
  const bool etf_ld = (n2 >= P.isplit - 1), etf_st = (n2 >= P.isplit - 2);
  const bool fu_on = (i0 >= 2 && i0 <= P.imm1), w_on = (i0 >= 2 && i0 <= P.im);
  const int jcl = P.jml;
  auto clampr = [&](int r) { return r < 1 ? 1 : (r > jcl ? jcl : r); };
  auto load_x = [&](int row, XRow &r, SRow &s_) {
    const unsigned ro = RO(clampr(row));
    r.d = XLD(X2_d, vo, ro); r.ua = XLD(X2_ua, vo, ro); r.va = XLD(X2_va, vo, ro);
    s_.dx = FLD(dx, vo, ro); s_.dy = FLD(dy, vo, ro);
    r.uab = XLD(X2_uab, vo, ro); r.vab = XLD(X2_vab, vo, ro); s_.am = FLD(aam2d, vo, ro);
    r.dW = XLD(X2_d, voW, ro); s_.dyW = FLD(dy, voW, ro);
  };
  auto fix_w = [&](XRow &r, SRow &s_) { r.dW = west_true(r.d, r.dW, lane); s_.dyW = west_true(s_.dy, s_.dyW, lane); };
  // the pointwise operands of one row of the first substep, and the accumulators of that row (accrow: a row this segment stores)
  auto load_pt = [&](int row, PRow &pr, SPt &q, EAcc &a, bool live, bool accrow) {
    const unsigned ro = RO(clampr(row));
    pr.el = XLD(X2_el, vo, ro); pr.elb = XLD(X2_elb, vo, ro);
    q.art = FLD(art, o_area, ro); q.vfl = FLD(vfluxf, vo, ro);
    q.mk = P.m8[(size_t)WAVE_UNIFORM(clampr(row) - 1) * (size_t)P.iml + (size_t)(i - 1)];
    q.ea = FLD(e_atmos, vo, ro); q.h = FLD(h, vo, ro); q.cor = FLD(cor, vo, ro);
    const unsigned vl = live ? vo : BOFF_NONE, vla = live ? o_area : BOFF_NONE;   // (wave-uniform) the warm-up row needs no momentum operands
    q.adx2d = FLD(adx2d, vl, ro); q.aru = FLD(aru, vla, ro); q.drx2d = FLD(drx2d, vl, ro);
    q.wusurf = FLD(wusurf, vl, ro); q.wubot = FLD(wubot, vl, ro);
    q.ady2d = FLD(ady2d, vl, ro); q.arv = FLD(arv, vla, ro); q.dry2d = FLD(dry2d, vl, ro);
    q.wvsurf = FLD(wvsurf, vl, ro); q.wvbot = FLD(wvbot, vl, ro);
    const unsigned va = accrow ? vo : BOFF_NONE;              // (wave-uniform)
    a.egf = FLD(egf, va, ro); a.utf = FLD(utf, va, ro); a.vtf = FLD(vtf, va, ro);
    a.etf = FLD(etf, (accrow && etf_ld) ? vo : BOFF_NONE, ro);
  };
  // generation Y of row `row` from the first substep's results; the lines next to the ring also go to T
  auto make_y = [&](int row, const EOut &o, YRow &y) {
    y.x.d = o.dn; y.x.ua = o.u; y.x.va = o.v; y.x.uab = o.uab_n; y.x.vab = o.vab_n;
    y.p.el = o.ec; y.p.elb = o.elb_n;
    y.x.dW = wave_up1(y.x.d);
    const bool bandrow = (row > RING_LO && row <= RING_LO + 2) || (row >= P.jm - RING_HI - 1 && row <= P.jm - RING_HI);   // wave-uniform
    const bool st = own && i0 > RING_LO && i0 <= P.im - RING_HI && row > RING_LO && row <= P.jm - RING_HI && (bandrow || bandcol);
    const unsigned vs = st ? vo : BOFF_NONE;
    const unsigned ro = RO(clampr(row));
    TST(X2_d, vs, ro, y.x.d); TST(X2_ua, vs, ro, y.x.ua); TST(X2_va, vs, ro, y.x.va);
    TST(X2_uab, vs, ro, y.x.uab); TST(X2_vab, vs, ro, y.x.vab);
    TST(X2_el, vs, ro, y.p.el); TST(X2_elb, vs, ro, y.p.elb);
  };
  // ---- first substep: rows tw-1, tw, tw+1 of X and the warm-up row tw = j0-3 (>= 2)
  const int tw = j0 - 3;
  XRow xm, xc, xp, xn;
  SRow sq, sm, sc, sp, sn;                                   // sq: the static row above the first substep's three (the second one's j-1)
  load_x(tw - 1, xm, sm); load_x(tw, xc, sc); load_x(tw + 1, xp, sp);
  fix_w(xm, sm); fix_w(xc, sc); fix_w(xp, sp);
  sq = sm;
  ECarry c1, c2;
  c1.fva_0 = .25 * (xc.d + xm.d) * (sc.dx + sm.dx) * xc.va;
  c1.es = c1.tps0 = c1.fv0 = c1.gvM = c1.el_m1 = c1.elb_m1 = 0.;
  c2 = c1;
  SPt q1, q1n, q1m;                                          // statics of the first substep's row, of the next one (in flight), of the row above (= the second substep's row)
  SPm qmm, qm2;                                              // e_atmos, h, cor of the row above the first substep's / above the second substep's
  PRow p1, p1n;
  EAcc a1, a1n, b1;                                          // accumulators of the first substep's row as loaded / of the next row (in flight); b1: of the second substep's row after the first substep
  EOut o1, o2;
  YRow ym, yc, yp;
  ym = YRow(); yc = YRow(); yp = YRow();
  b1.egf = b1.utf = b1.vtf = b1.etf = 0.;
  {
    load_pt(tw, p1, q1, a1, false, false);
    load_x(tw + 2, xn, sn);
    load_pt(tw + 1, p1n, q1n, a1n, true, false);              // the first live row's operands: one row ahead from here on
    qmm.ea = qmm.h = qmm.cor = 0.;
    ext_row_math<false>(P, canon, fu_on, w_on, xm, xc, xp, sm, sc, sp, p1, q1, qmm, c1, o1);
    q1m = q1;
    qmm.ea = q1.ea; qmm.h = q1.h; qmm.cor = q1.cor;
    qm2 = qmm;
    sq = sm; xm = xc; xc = xp; sm = sc; sc = sp; fix_w(xn, sn); xp = xn; sp = sn;
    p1 = p1n; q1 = q1n; a1 = a1n;
  }
  // One live row of the first substep (row t).  Before: xm/xc/xp = X rows t-1, t, t+1; sq/sm/sc/sp = static rows t-2 .. t+1; p1, q1, a1 =
  // the pointwise operands and accumulators of row t (requested one iteration ago); q1m = statics of row t-1, qmm = e_atmos, h, cor of row t-1.
  // After rotate(): everything one row further down; q1m = statics of row t, qm2 = e_atmos, h, cor of row t-1, b1 = row t's accumulators.
  EAcc an;
  auto first = [&](const int t, const bool next_accrow) {
    load_pt(t + 1, p1n, q1n, a1n, true, next_accrow);         // in flight during this iteration
    load_x(t + 2, xn, sn);
    ext_row_math<true>(P, canon, fu_on, w_on, xm, xc, xp, sm, sc, sp, p1, q1, qmm, c1, o1);
    an.egf = a1.egf + o1.ec * P.ispi;
    an.utf = a1.utf + o1.u * (o1.dn + (o1.hW + o1.ew)) * P.isp2i;
    an.vtf = a1.vtf + o1.v * (o1.dn + (qmm.h + o1.esm)) * P.isp2i;
    an.etf = etf_rule(P, n1, a1.etf, o1.ec, (double)(q1.mk & 1u));
    make_y(t, o1, yp);
  };
  auto rotate = [&]() {
    sq = sm; sm = sc; sc = sp; xm = xc; xc = xp;
    fix_w(xn, sn); xp = xn; sp = sn;
    qm2 = qmm;
    qmm.ea = q1.ea; qmm.h = q1.h; qmm.cor = q1.cor;
    q1m = q1;
    p1 = p1n; q1 = q1n; a1 = a1n;
    ym = yc; yc = yp;
    b1 = an;
  };
  // ---- Y rows j0-2, j0-1: two live rows of the first substep
  first(j0 - 2, false); rotate();
  first(j0 - 1, true); rotate();
  // ---- row j0 of the first substep, then the warm-up row j0-1 of the second: Y rows j0-2, j0-1, j0; statics sq, sm, sc = rows j0-2 .. j0
  {
    first(j0, j0 + 1 <= j1);
    c2.fva_0 = .25 * (yc.x.d + ym.x.d) * (sm.dx + sq.dx) * yc.x.va;
    SPm z3; z3.ea = z3.h = z3.cor = 0.;
    ext_row_math<false>(P, canon, fu_on, w_on, ym.x, yc.x, yp.x, sq, sm, sc, yc.p, q1m, z3, c2, o2);
    rotate();
  }
  // ---- the rows: first substep at t, second at t-1
  for (int t = j0 + 1; t <= j1 + 1; t++) {
    first(t, t + 1 <= j1);
    // second substep, row q = t-1: Y rows t-2, t-1, t; static rows sq, sm, sc; statics q1m (row q), qm2 (row q-1)
    ext_row_math<true>(P, canon, fu_on, w_on, ym.x, yc.x, yp.x, sq, sm, sc, yc.p, q1m, qm2, c2, o2);
    {
      const unsigned ro = RO(t - 1);
      const unsigned o_f = store_f2 ? o_st : BOFF_NONE;
      FST(elf, o_f, ro, o2.ec); FST(uaf, o_f, ro, o2.u); FST(vaf, o_f, ro, o2.v);
      FST(advua, o_f, ro, o2.advua); FST(advva, o_f, ro, o2.advva);
      FST(etf, etf_st ? o_st : BOFF_NONE, ro, etf_rule(P, n2, b1.etf, o2.ec, (double)(q1m.mk & 1u)));
      ZST(X2_uab, o_st, ro, o2.uab_n);
      ZST(X2_vab, o_st, ro, o2.vab_n);
      ZST(X2_elb, o_st, ro, o2.elb_n);
      ZST(X2_el, o_st, ro, o2.ec);
      ZST(X2_d, o_st, ro, o2.dn);
      ZST(X2_ua, o_st, ro, o2.u);
      ZST(X2_va, o_st, ro, o2.v);
      FST(egf, o_st, ro, acc2 ? b1.egf + o2.ec * P.ispi : b1.egf);
      FST(utf, o_st, ro, acc2 ? b1.utf + o2.u * (o2.dn + (o2.hW + o2.ew)) * P.isp2i : b1.utf);
      FST(vtf, o_st, ro, acc2 ? b1.vtf + o2.v * (o2.dn + (qm2.h + o2.esm)) * P.isp2i : b1.vtf);
    }
    rotate();
  }
#undef RO
#undef XLD
#undef ZST
#undef TST
#undef FLD
#undef FST
}
#endif
// Tried and dropped (profiles/round3_ext_pair.txt): the two substeps as a producer and a consumer WAVEFRONT of one workgroup (A evaluates
// substep n and mails each row -- generation Y, the static operands, the accumulators -- through LDS to B, which evaluates substep
// n+1 two rows behind; 256 registers each, two waves per SIMD, one workgroup barrier per row).  Bit-identical, the same 1.15 GB per
// pair, but 308-341 us per pair against 251: the waves wait at the per-row barrier (SQ_WAIT_INST_ANY 9.9e7 against 1.7e7 cycles),
// VALU busy 19 % per wave against 53 %.
// the ring of one substep (the cells k_ext_march2 leaves out), advave formed in place: P.x2 = the generation read, P.y2 = the generation written
__global__ void __launch_bounds__(64) k_ext_ring(KP P, int store_f, int nlo, int nhi) {
  ext_ring_cell(P, (int)(blockIdx.x * blockDim.x + threadIdx.x), store_f, nlo, nhi);
}

// ---- all the external substeps of an internal step in ONE launch (small tiles) ------------------------------------------
// On a small tile a substep's kernel is a few wavefronts per CU on a chain of dependent loads: ~20 us each, 30 of them per
// internal step, whatever the grid size (256x256x30: 0.68 of 1.34 ms per step) -- a launch boundary costs more than the
// work.  When every workgroup of the substep's grid is resident at once (k_ext_step<1> holds 245 VGPRs: two 256-thread
// workgroups per CU, 512 on the chip), k_ext_loop keeps them there and walks the substeps itself: the same body, the two
// buffer generations swapped in registers, and a grid-wide barrier in between (MI355X_MICROARCH.md, "barrier-counter":
// every storing wavefront drains its stores, the workgroup meets, one lane releases at agent scope -- the XCD's L2 writes
// its dirty lines back --, adds to a counter, polls it with relaxed loads, acquires -- the CU's L1 is invalidated -- and
// the workgroup goes on).  Every spin is bounded: a workgroup that waits too long raises the abort word, which every
// other poll also watches, sets the error flag and all of them leave; the host then reports the step as failed.
#ifndef POMGPU_EMU
__device__ __forceinline__ bool ext_grid_barrier(unsigned *bar, unsigned target) {   // bar[0] arrivals, bar[1] abort; false = give up
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wavefront's stores have left
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 1;
    for (unsigned spin = 0;; spin++) {
      if ((int)(__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) break;
      if (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spin > (1u << 22)) {   // ~1 s: something is not resident
        __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}
// Two views of the arguments, one per buffer generation, taken once: a KP whose pointers are swapped inside the loop lived
// in vector registers (260 VGPRs: one workgroup per CU, and a 256x256 tile needs two)
template <int FUSE_ADV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_ext_loop(KP P, int rim_rows, int first, int last, unsigned *bar, unsigned base, int *d_err) {
  const unsigned nblk = gridDim.x * gridDim.y;
  unsigned tgt = base;
  for (int n = first; n <= last; n += 2) {
    {
      KP Q = P;
      Q.iext = n;
      ext_step_body<FUSE_ADV>(Q, n == P.isplit, rim_rows);
    }
    if (n == last) break;
    tgt += nblk;
    if (!ext_grid_barrier(bar, tgt)) { if (threadIdx.x == 0 && threadIdx.y == 0) atomicOr(d_err, POMGPU_DERR_BARRIER); return; }
    {
      KP Q = P;
#pragma unroll
      for (int g = 0; g < POMGPU_NGEN; g++) { Q.x2[g] = P.y2[g]; Q.y2[g] = P.x2[g]; }
      Q.iext = n + 1;
      ext_step_body<FUSE_ADV>(Q, n + 1 == P.isplit, rim_rows);
    }
    if (n + 1 == last) break;
    tgt += nblk;
    if (!ext_grid_barrier(bar, tgt)) { if (threadIdx.x == 0 && threadIdx.y == 0) atomicOr(d_err, POMGPU_DERR_BARRIER); return; }
  }
}
#endif
__global__ void k_ext_step_rim(KP P, int store_f) {
  ext_rim_cell(P, (int)(blockIdx.x * blockDim.x + threadIdx.x), store_f, 0);
}

// mode_internal tail: rotate the 2-D time levels -- advance.f:525-531 (whole arrays) -- and, in the same pass, the derived
// coefficients of the new dt (k_coef_dt, k_tile.hip: the same expressions; a neighbour's dt is formed here as h + etf, the sum
// that cell's own thread stores -- the same bits)
__global__ void k_int_tail(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  F2(egb, i, j) = F2(egf, i, j);
  F2(etb, i, j) = F2(et, i, j);
  const double etf = F2(etf, i, j);
  F2(et, i, j) = etf;
  const double dt = F2(h, i, j) + etf;
  F2(dt, i, j) = dt;
  F2(utb, i, j) = F2(utf, i, j);
  F2(vtb, i, j) = F2(vtf, i, j);
  F2(vfluxb, i, j) = F2(vfluxf, i, j);
  const int iw = i > 1 ? i - 1 : 1, js = j > 1 ? j - 1 : 1;
  const double dtw = F2(h, iw, j) + F2(etf, iw, j), dts = F2(h, i, js) + F2(etf, i, js), dtws = F2(h, iw, js) + F2(etf, iw, js);
  K2(CMX, i, j) = 0.25 * (F2(dy, iw, j) + F2(dy, i, j)) * (dtw + dt);
  K2(CMY, i, j) = 0.25 * (F2(dx, i, js) + F2(dx, i, j)) * (dts + dt);
  K2(DTSX, i, j) = dt + dtw;
  K2(DTSY, i, j) = dt + dts;
  K2(DT4, i, j) = dt + dtw + dts + dtws;
}

// ---------------------------------------------------------------------------------------------
// surface forcing (bounds_forcing.f:871-983): x(1:im,1:jm) = record, and the linear interpolation in time
__global__ void k_frc_load(KP P, const double *ra, const double *rb, double *xf, double *yf) {   // records are (im,jm)
  const int i = TID_I, j = TID_J;
  if (i > P.im || j > P.jm) return;
  const size_t r = (size_t)(j - 1) * P.im + (size_t)(i - 1);
  G2(xf, i, j) = ra[r];
  if (yf) G2(yf, i, j) = rb[r];
}
__global__ void k_frc_interp(KP P, double fold, double fnew, double *x, const double *xb, const double *xf, double *y, const double *yb,
                             const double *yf) {
  const int i = TID_I, j = TID_J;
  if (i > P.im || j > P.jm) return;
  G2(x, i, j) = fold * G2(xb, i, j) + fnew * G2(xf, i, j);                                  // :909-910 / :952-955
  G2(y, i, j) = fold * G2(yb, i, j) + fnew * G2(yf, i, j);
}

// ---- launchers --------------------------------------------------------------------------------
void launch_frc_load(pomgpu_ctx *c, const double *ra, const double *rb, double *xf, double *yf) {
  LAUNCH(c, k_frc_load, grid2(c->P), blk2(), c->P, ra, rb, xf, yf);
}
void launch_frc_interp(pomgpu_ctx *c, double fold, double fnew, double *x, const double *xb, const double *xf, double *y, const double *yb,
                       const double *yf) {
  LAUNCH(c, k_frc_interp, grid2(c->P), blk2(), c->P, fold, fnew, x, xb, xf, y, yb, yf);
}
void launch_advave_a(pomgpu_ctx *c) { LAUNCH(c, k_advave_a, grid2(c->P), blk2(), c->P); }
void launch_advave_b(pomgpu_ctx *c) { LAUNCH(c, k_advave_b, grid2(c->P), blk2(), c->P); }
void launch_advave_c(pomgpu_ctx *c) { LAUNCH(c, k_advave_c, grid2(c->P), blk2(), c->P); }
void launch_advave_fused(pomgpu_ctx *c) {
  const KP &P = c->P;
  if (P.iml % 2 == 0 && !SW(c, NO_PAIR)) LAUNCHN(c, "k_advave_pair", k_advave_pair, dim3((P.iml / 2 + 61) / 62, (P.jml + 3) / 4, 1), blk2(), c->P);
  else LAUNCH(c, k_advave_fused, grid2_halo(c->P), blk2(), c->P);
}
void launch_advave_m2a(pomgpu_ctx *c) { LAUNCH(c, k_advave_m2a, grid2(c->P), blk2(), c->P); }
void launch_advave_m2b(pomgpu_ctx *c) { LAUNCH(c, k_advave_m2b, grid2(c->P), blk2(), c->P); }
void launch_vint(pomgpu_ctx *c, int only_aam, int ghost) { LAUNCH(c, k_vint, grid2(c->P), blk2(), c->P, only_aam, ghost); }
void launch_modeint_tail(pomgpu_ctx *c) { LAUNCH(c, k_modeint_tail, grid2(c->P), blk2(), c->P); }
void launch_ext_elf(pomgpu_ctx *c) { LAUNCH(c, k_ext_elf, grid2(c->P), blk2(), c->P); }
void launch_ext_uvaf(pomgpu_ctx *c, int interior) { LAUNCH(c, k_ext_uvaf, grid2(c->P), blk2(), c->P, interior); }
void launch_ext_update(pomgpu_ctx *c) { LAUNCH(c, k_ext_update, grid2(c->P), blk2(), c->P); }
// Q: c->P with y2 pointing at the next-generation buffers
void launch_ext_step(pomgpu_ctx *c, const KP &Q, int store_f, int fuse_adv) {
  const int n = 3 * Q.im + 3 * (Q.jm > 3 ? Q.jm - 3 : 0);
  dim3 g = grid2_halo(Q);
  if (SW(c, EXT_RIM_KERNEL) && !fuse_adv) {         // developer switch: the rim as a launch of its own
    LAUNCHN(c, "k_ext_step", k_ext_step<0>, g, blk2(), Q, store_f, 0);
    LAUNCH(c, k_ext_step_rim, dim3((n + 63) / 64, 1, 1), dim3(64, 1, 1), Q, store_f);
    return;
  }
  if (fuse_adv && !SW(c, EXT_NOMARCH)) {            // large tiles: a wavefront marches down `rows` rows (k_ext_march)
    // rows per wavefront: enough wavefronts first (>= ~7400, i.e. 3.6 per wave slot of the chip), then taller segments (fewer
    // halo rows): 7 at 2048x1536 (5-9 within 2 %), 2 at 1024x1024 (2.13 ms per step against 2.43 with 7 and 2.52 with the
    // one-row kernel); below ~16500 wavefront-rows (the extended tile of a 4- or 8-tile split of that grid) the one-row kernel
    // is faster (profiles/round2_ext_march.txt)
    const long wave_rows = (long)g.x * (Q.jmm1 - 2);
    int rows = (int)(wave_rows / 7400);
    rows = rows < 2 ? 2 : (rows > 7 ? 7 : rows);
    if (SW(c, EXT_ROWS)) rows = (int)SWV(c, EXT_ROWS);
    const int nseg = rows > 0 ? (Q.jmm1 - 3 + 1 + rows - 1) / rows : 0, nbx = (int)g.x;
    const bool fits = (size_t)POM_NBLK2D * Q.n2 * 8 < ((size_t)1 << 32);       // blk2d through one 32-bit buffer descriptor
    if (fits && rows >= 2 && rows <= 256 && Q.jmm1 >= 3 && (wave_rows >= 16500 || SW(c, EXT_MARCH))) {
      const int rim_wgs = ((n + 255) / 256 + 7) / 8 * 8, gpx = ((nseg + 3) / 4 + 7) / 8;
      LAUNCHN(c, "k_ext_step_adv", k_ext_march, dim3((unsigned)(rim_wgs + 8 * gpx * nbx), 1, 1), blk2(), Q, store_f, rim_wgs, rows, (const int *)c->d_areas, SW(c, EXT_AREAS_LOAD) ? 0 : 1);
      return;
    }
  }
  const int per_row = (int)g.x * 256, rim_rows = (n + per_row - 1) / per_row;
  g.y += rim_rows;
  if (fuse_adv) LAUNCHN(c, "k_ext_step_adv", k_ext_step<1>, g, blk2(), Q, store_f, rim_rows);
  else LAUNCHN(c, "k_ext_step", k_ext_step<0>, g, blk2(), Q, store_f, rim_rows);
}
// Two substeps (Q.iext, Q.iext + 1) in one pass over memory (k_ext_march2, the ring's first substep beside it) + the ring's second substep (k_ext_ring).  Q: c->P with x2 = the
// generation read, y2 = the generation written (the other buffer set); T: a third set for the rim's / the band's intermediate
// generation.  Returns 1 when launched, 0 when the tile is not one for this path (the caller takes the substeps one by one).
// is the tile one for that path?  (without launching: pomgpu_mode_external decides with it whether an odd substep may wait for its partner)
#ifndef POMGPU_EXT_PAIR_MIN_WAVE_ROWS
#define POMGPU_EXT_PAIR_MIN_WAVE_ROWS 6000
#endif
int launch_ext_pair_ok(const pomgpu_switches &sw, const KP &Q) {
#ifdef POMGPU_EMU
  (void)Q; (void)sw;
  return 0;                                                   // neighbour lanes are not emulated (tests/emu): the GPU tests cover this path
#else
  if (sw.on[SW_EXT_NOPAIR]) return 0;
  const bool fits = (size_t)POM_NBLK2D * Q.n2 * 8 < ((size_t)1 << 32);         // blk2d through one 32-bit buffer descriptor
  if (!fits || Q.im < 16 || Q.jm < 16) return 0;
  // small grids keep the one-substep kernels unless asked (tests): the ring's ~16 us of dependent latency per pair eats the gain
  // there.  The extended tile of a 4- or 8-tile split of 2048x1536 (33 x 452 / 33 x 260 wavefront-rows) gains since it shrinks as
  // it goes stale (round 4: 1.17 -> 1.07 / 1.99 -> 1.92 ms per step before the window reached this path).  Decided on the WHOLE
  // tile: the windows of later pairs are smaller.
  const long wave_rows = (long)grid2_halo(Q).x * (Q.jmm1 - 2);
  if (wave_rows < POMGPU_EXT_PAIR_MIN_WAVE_ROWS && !sw.on[SW_EXT_PAIR]) return 0;
  return 1;
#endif
}
int launch_ext_pair(pomgpu_ctx *c, const KP &Q, double *const *T, int store_f2) {
#ifdef POMGPU_EMU
  (void)c; (void)Q; (void)T; (void)store_f2;
  return 0;
#else
  if (Q.im < 16 || Q.jm < 16) return 0;                      // (the caller has asked launch_ext_pair_ok about the whole tile; Q may be a window of it)
  static int ncu = 0;
  if (!ncu) {
    hipDeviceProp_t pr;
    ncu = (hipGetDeviceProperties(&pr, c->device) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  // rows per segment: a workgroup (four segments, one wave per SIMD) fills a CU; the pass costs about
  // ceil(workgroups / CUs) x (rows + 4.5) row evaluations -- the shortest such product
  const int nbx = (Q.im - RING_HI - 2) / 60 + 1, nrow = Q.jm - RING_HI - RING_LO;
  int rows = 0;
  double best = 1e30;
  const int ncg = (nbx + 3) / 4;
  const int ncu_m = ncu > 16 ? ncu - 4 : ncu;                 // a few CUs stay free for the ring's workgroups (the last ones of the grid)
  for (int r = 6; r <= 96; r++) {
    const int nseg = (nrow + r - 1) / r;
    const long nwg = (long)ncg * nseg;
    const double cost = (double)((nwg + ncu_m - 1) / ncu_m) * (r + 4.5);
    if (cost < best) { best = cost; rows = r; }
  }
  if (SW(c, EXT_ROWS2)) rows = (int)SWV(c, EXT_ROWS2);
  if (rows < 2) return 0;
  const int nseg = (nrow + rows - 1) / rows;
  const int nl = RING_LO + RING_HI, n = nl * Q.im + nl * (Q.jm - nl);          // cells of the ring
  const int ring_wgs = ((n + 255) / 256 + 7) / 8 * 8;
  Gen7 Tg;
  for (int g = 0; g < POMGPU_NGEN; g++) Tg.p[g] = T[g];
  // each generation behind one descriptor: the blk2d block (arrays at their slots) or a buffer set of its own (one allocation)
  auto gen_of = [&](double *const *a, GenD &G) {
    const bool in_b2 = a[0] >= Q.b2 && a[0] < Q.b2 + (size_t)POM_NBLK2D * Q.n2;
    G.base = in_b2 ? Q.b2 : a[0];
    G.bytes = (unsigned)((in_b2 ? (size_t)POM_NBLK2D : (size_t)POMGPU_NGEN) * Q.n2 * 8);
    for (int g = 0; g < POMGPU_NGEN; g++) {
      const ptrdiff_t d = a[g] - G.base;
      if (d < 0 || (size_t)d * 8 + Q.n2 * 8 > (size_t)G.bytes) return false;            // not one block: not this path
      G.off[g] = (unsigned)((size_t)d * 8);
    }
    return true;
  };
  GenD GX, GZ, GT;
  if (!gen_of(Q.x2, GX) || !gen_of(Q.y2, GZ) || !gen_of(T, GT)) return 0;
  // launch 1: the ring's first substep (X -> T) beside the marching workgroups (X -> Z, band of Y -> T), as the grid's LAST workgroups
  // where the march is long, as its FIRST ones where it is short.  The ring's workgroups are workgroups of this kernel -- one wave
  // per SIMD like the marching ones -- so each takes a CU for its ~10 us.  Measured in one process each (round 4, ms per step for the
  // 15 pairs): 2048x1536 (57-row segments) first 4.04, last 3.52; extended tile of a 4-tile split (16 rows) 1.49 / 1.40; of an 8-tile
  // split (10 rows) 0.85 / 1.03 -- behind a short march the ring's dependent loads queue up behind the march's opening burst and
  // the launch ends with them.  POMGPU_EXT_RING_FIRST = 1 / 2 forces first / last.
  const int ring_first = SW(c, EXT_RING_FIRST) ? ((int)SWV(c, EXT_RING_FIRST) == 1) : (rows <= 12);
  LAUNCHN(c, "k_ext_pair", k_ext_march2, dim3((unsigned)(ring_wgs + nseg * ncg), 1, 1), blk2(), Q, Tg, GX, GZ, GT, store_f2, ring_wgs, rows, (const int *)c->d_areas,
          SW(c, EXT_AREAS_LOAD) ? 0 : 1, ring_first);
  // launch 2: the ring's second substep (T -> Z)
  KP R2 = Q;
  for (int g = 0; g < POMGPU_NGEN; g++) R2.x2[g] = T[g];
  R2.iext = Q.iext + 1;
  LAUNCHN(c, "k_ext_ring", k_ext_ring, dim3((n + 63) / 64, 1, 1), dim3(64, 1, 1), R2, store_f2, RING_LO, RING_HI);
  return 1;
#endif
}
// every substep first..last in one launch, or 0 when the tile is too large for all its workgroups to be resident (the
// caller then launches the substeps one by one).  Q: as for launch_ext_step (y2 = the other buffer set).
int launch_ext_loop(pomgpu_ctx *c, const KP &Q, int first, int last) {
#ifdef POMGPU_EMU
  (void)c; (void)Q; (void)first; (void)last;
  return 0;
#else
  // Opt-in (POMGPU_EXT_LOOP=1): measured on MI355X it does not pay.  256x256 (325 workgroups, two per CU): 1.40 ms for the
  // 30 substeps against 0.64 ms as 30 launches -- ~45 us per barrier once every workgroup has ~20 KB of freshly written
  // lines for the L2 write-back of the release and 325 lanes poll one counter; 65x49 (28 workgroups): 0.35 against 0.39 ms,
  // nothing on the step.  A launch boundary on this chip (~1.5-2 us + the ~20 us the substep's dependent loads take anyway)
  // is cheaper than this barrier; an XCD-hierarchical one (MI355X_MICROARCH.md: ~6 us + the publish) would be the next try.
  if (!SW(c, EXT_LOOP) || last <= first || c->ext_loop_off) return 0;
  static int occ = -1, ncu = 0;
  if (occ < 0) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, c->device) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_ext_loop<1>, 256, 0) != hipSuccess) { occ = 0; return 0; }
    ncu = pr.multiProcessorCount;
    if (occ > 2) occ = 2;                                     // never count on more than the register file certainly gives (245 VGPRs)
  }
  const int n = 3 * Q.im + 3 * (Q.jm > 3 ? Q.jm - 3 : 0);
  dim3 g = grid2_halo(Q);
  const int per_row = (int)g.x * 256, rim_rows = (n + per_row - 1) / per_row;
  g.y += rim_rows;
  const long blocks = (long)g.x * g.y;
  if (occ < 1 || blocks > (long)ncu * occ * 7 / 8) return 0;  // a margin: a workgroup that is not resident never reaches the barrier
  if (!c->ext_bar) {
    if (hipMalloc((void **)&c->ext_bar, 2 * sizeof(unsigned)) != hipSuccess) { c->ext_bar = NULL; return 0; }
    (void)hipMemsetAsync(c->ext_bar, 0, 2 * sizeof(unsigned), c->cur);
    c->ext_bar_base = 0;
  }
  // the error flag is the TILE's (the extended tile of the wide-halo mode reports through its parent: nobody pulls its own)
  LAUNCHN(c, "k_ext_loop", k_ext_loop<1>, g, blk2(), Q, rim_rows, first, last, c->ext_bar, c->ext_bar_base, (c->parent ? c->parent : c)->d_err);
  c->ext_bar_base += (unsigned)(blocks * (last - first));
  return 1;
#endif
}
void launch_copy2(pomgpu_ctx *c, double *dst, const double *src) { LAUNCH(c, k_copy2, grid2(c->P), blk2(), c->P, dst, src); }
void launch_int_tail(pomgpu_ctx *c) { LAUNCH(c, k_int_tail, grid2(c->P), blk2(), c->P); }
void launch_bcond1(pomgpu_ctx *c) {
  LAUNCH(c, k_bcond1, grid2(c->P), blk2(), c->P);
  LAUNCH(c, k_copy2, grid2(c->P), blk2(), c->P, c->P.b2 + (size_t)P2_elf * c->P.n2, (const double *)c->P.s2[3]);
}
