// k_ext.hip -- the external (2-D barotropic) mode: advave, mode_external, the 2-D parts of
// mode_interaction.  Every kernel is one thread per water column, threadIdx.x along i (coalesced
// 512-byte row segments per wavefront); each of them is a handful of 2-D array passes, i.e.
// HBM/L2-bound at ~10 flop per 8-byte word.
//
// Fusion rule used throughout: all statements of the reference between two halo-exchange points
// become ONE kernel; neighbour values of intermediates that the reference stored in scratch arrays
// (fluxua, fluxva) are recomputed in registers where that needs no data from beyond the tile.
#include "pomgpu_internal.hpp"

#define d_(i, j) F2(d, i, j)
#define dx_(i, j) F2(dx, i, j)
#define dy_(i, j) F2(dy, i, j)
#define ua_(i, j) F2(ua, i, j)
#define va_(i, j) F2(va, i, j)
#define uab_(i, j) F2(uab, i, j)
#define vab_(i, j) F2(vab, i, j)
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
__global__ void k_vint(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  double ax = 0., ay = 0., rx = 0., ry = 0., am = 0.;
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
  F2(egf, i, j) = F2(el, i, j) * P.ispi;
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
__global__ void k_ext_elf(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const int ii = (P.W && i == 1) ? 2 : ((P.E && i == P.im) ? P.imm1 : i);
  const int jj = (P.S && j == 1) ? 2 : ((P.N && j == P.jm) ? P.jmm1 : j);
  double v = F2(elf, i, j);
  if (ii >= 2 && ii <= P.imm1 && jj >= 2 && jj <= P.jmm1 && i <= P.im && j <= P.jm)
    v = F2(elb, ii, jj) +
        P.dte2 * (-(flux_ua(P, ii + 1, jj) - flux_ua(P, ii, jj) + flux_va(P, ii, jj + 1) - flux_va(P, ii, jj)) /
                      F2(art, ii, jj) -
                  F2(vfluxf, ii, jj));
  F2(elf, i, j) = v * F2(fsm, i, j);
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
__device__ __forceinline__ double uaf_interior(const KP &P, int i, int j) {
  double v = F2(adx2d, i, j) + F2(advua, i, j) -
             F2(aru, i, j) * .25 *
                 (F2(cor, i, j) * d_(i, j) * (va_(i, j + 1) + va_(i, j)) +
                  F2(cor, i - 1, j) * d_(i - 1, j) * (va_(i - 1, j + 1) + va_(i - 1, j))) +
             .25 * P.grav * (dy_(i, j) + dy_(i - 1, j)) * (d_(i, j) + d_(i - 1, j)) *
                 ((1. - 2. * P.alpha) * (F2(el, i, j) - F2(el, i - 1, j)) +
                  P.alpha * (F2(elb, i, j) - F2(elb, i - 1, j) + F2(elf, i, j) - F2(elf, i - 1, j)) +
                  F2(e_atmos, i, j) - F2(e_atmos, i - 1, j)) +
             F2(drx2d, i, j) + F2(aru, i, j) * (F2(wusurf, i, j) - F2(wubot, i, j));          // :239-250
  v = ((F2(h, i, j) + F2(elb, i, j) + F2(h, i - 1, j) + F2(elb, i - 1, j)) * F2(aru, i, j) * uab_(i, j) -
       4. * P.dte * v) /
      ((F2(h, i, j) + F2(elf, i, j) + F2(h, i - 1, j) + F2(elf, i - 1, j)) * F2(aru, i, j));  // :256-260
  return v;
}
__device__ __forceinline__ double vaf_interior(const KP &P, int i, int j) {
  double v = F2(ady2d, i, j) + F2(advva, i, j) +
             F2(arv, i, j) * .25 *
                 (F2(cor, i, j) * d_(i, j) * (ua_(i + 1, j) + ua_(i, j)) +
                  F2(cor, i, j - 1) * d_(i, j - 1) * (ua_(i + 1, j - 1) + ua_(i, j - 1))) +
             .25 * P.grav * (dx_(i, j) + dx_(i, j - 1)) * (d_(i, j) + d_(i, j - 1)) *
                 ((1. - 2. * P.alpha) * (F2(el, i, j) - F2(el, i, j - 1)) +
                  P.alpha * (F2(elb, i, j) - F2(elb, i, j - 1) + F2(elf, i, j) - F2(elf, i, j - 1)) +
                  F2(e_atmos, i, j) - F2(e_atmos, i, j - 1)) +
             F2(dry2d, i, j) + F2(arv, i, j) * (F2(wvsurf, i, j) - F2(wvbot, i, j));          // :266-276
  v = ((F2(h, i, j) + F2(elb, i, j) + F2(h, i, j - 1) + F2(elb, i, j - 1)) * F2(arv, i, j) * vab_(i, j) -
       4. * P.dte * v) /
      ((F2(h, i, j) + F2(elf, i, j) + F2(h, i, j - 1) + F2(elf, i, j - 1)) * F2(arv, i, j));  // :282-286
  return v;
}
// the open-boundary values of bcond(2); `interior` = 0 skips the advance.f formulas (bcond alone)
__device__ __forceinline__ void uvaf_cell(const KP &P, int i, int j, int interior, double &uo, double &vo) {
  double u = F2(uaf, i, j), v = F2(vaf, i, j);
  const bool jin = (j >= 2 && j <= P.jmm1), iin = (i >= 2 && i <= P.imm1);
  if (interior) {
    if (i >= 2 && i <= P.im && jin) u = uaf_interior(P, i, j);
    if (iin && j >= 2 && j <= P.jm) v = vaf_interior(P, i, j);
  }
  if (P.W && jin && (i == 1 || i == 2)) {                                                 // :47-53
    if (i == 1) v = BD1(vabw, j);
    u = BD1(uabw, j) - P.rfw * sqrt(P.grav / d_(2, j)) * (F2(el, 2, j) - BD1(elw, j));
    u = P.ramp * u;
  }
  if (P.E && jin && i == P.im) {                                                          // :56-61
    u = BD1(uabe, j) + P.rfe * sqrt(P.grav / d_(P.imm1, j)) * (F2(el, P.imm1, j) - BD1(ele, j));
    u = P.ramp * u;
    v = BD1(vabe, j);
  }
  if (P.S && iin && (j == 1 || j == 2)) {                                                 // :64-70
    if (j == 1) u = BD1(uabs, i);
    v = BD1(vabs, i) - P.rfs * sqrt(P.grav / d_(i, 2)) * (F2(el, i, 2) - BD1(els, i));
    v = P.ramp * v;
  }
  if (P.N && iin && j == P.jm) {                                                          // :73-78
    v = BD1(vabn, i) + P.rfn * sqrt(P.grav / d_(i, P.jmm1)) * (F2(el, i, P.jmm1) - BD1(eln, i));
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
  if (i <= P.im && j <= P.jm) uvaf_cell(P, i, j, interior, u, v);
  else { u = F2(uaf, i, j); v = F2(vaf, i, j); }
  F2(uaf, i, j) = u * F2(dum, i, j);                                                      // :80-81
  F2(vaf, i, j) = v * F2(dvm, i, j);
}

// mode_external, etf weights + Asselin filter + time rotation + accumulation -- advance.f:295-350
__global__ void k_ext_update(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const bool act = (i <= P.im && j <= P.jm);
  const double elf = F2(elf, i, j), uaf = F2(uaf, i, j), vaf = F2(vaf, i, j);
  if (act) {                                                                              // :295-318
    if (P.iext == P.isplit - 2) F2(etf, i, j) = .25 * P.smoth * elf;
    else if (P.iext == P.isplit - 1) F2(etf, i, j) = F2(etf, i, j) + .5 * (1. - .5 * P.smoth) * elf;
    else if (P.iext == P.isplit) F2(etf, i, j) = (F2(etf, i, j) + .5 * elf) * F2(fsm, i, j);
  }
  const double ua = ua_(i, j), va = va_(i, j), el = F2(el, i, j);
  F2(uab, i, j) = ua + .5 * P.smoth * (uab_(i, j) - 2. * ua + uaf);                       // :321,327
  F2(vab, i, j) = va + .5 * P.smoth * (vab_(i, j) - 2. * va + vaf);                       // :322,329
  F2(elb, i, j) = el + .5 * P.smoth * (F2(elb, i, j) - 2. * el + elf);                    // :323-324
  F2(el, i, j) = elf;                                                                     // :325
  const double dn = F2(h, i, j) + elf;
  F2(d, i, j) = dn;                                                                       // :326
  F2(ua, i, j) = uaf;                                                                     // :328
  F2(va, i, j) = vaf;                                                                     // :330
  if (P.iext != P.isplit && act) {                                                        // :332-347
    F2(egf, i, j) = F2(egf, i, j) + elf * P.ispi;
    if (i >= 2) F2(utf, i, j) = F2(utf, i, j) + uaf * (dn + (F2(h, i - 1, j) + F2(elf, i - 1, j))) * P.isp2i;
    if (j >= 2) F2(vtf, i, j) = F2(vtf, i, j) + vaf * (dn + (F2(h, i, j - 1) + F2(elf, i, j - 1))) * P.isp2i;
  }
}

// mode_internal tail: rotate the 2-D time levels -- advance.f:525-531 (whole arrays)
__global__ void k_int_tail(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  F2(egb, i, j) = F2(egf, i, j);
  F2(etb, i, j) = F2(et, i, j);
  const double etf = F2(etf, i, j);
  F2(et, i, j) = etf;
  F2(dt, i, j) = F2(h, i, j) + etf;
  F2(utb, i, j) = F2(utf, i, j);
  F2(vtb, i, j) = F2(vtf, i, j);
  F2(vfluxb, i, j) = F2(vfluxf, i, j);
}

// ---- launchers --------------------------------------------------------------------------------
void launch_advave_a(pomgpu_ctx *c) { LAUNCH(c, k_advave_a, grid2(c->P), blk2(), c->P); }
void launch_advave_b(pomgpu_ctx *c) { LAUNCH(c, k_advave_b, grid2(c->P), blk2(), c->P); }
void launch_advave_c(pomgpu_ctx *c) { LAUNCH(c, k_advave_c, grid2(c->P), blk2(), c->P); }
void launch_advave_fused(pomgpu_ctx *c) { LAUNCH(c, k_advave_fused, grid2_halo(c->P), blk2(), c->P); }
void launch_advave_m2a(pomgpu_ctx *c) { LAUNCH(c, k_advave_m2a, grid2(c->P), blk2(), c->P); }
void launch_advave_m2b(pomgpu_ctx *c) { LAUNCH(c, k_advave_m2b, grid2(c->P), blk2(), c->P); }
void launch_vint(pomgpu_ctx *c) { LAUNCH(c, k_vint, grid2(c->P), blk2(), c->P); }
void launch_modeint_tail(pomgpu_ctx *c) { LAUNCH(c, k_modeint_tail, grid2(c->P), blk2(), c->P); }
void launch_ext_elf(pomgpu_ctx *c) { LAUNCH(c, k_ext_elf, grid2(c->P), blk2(), c->P); }
void launch_ext_uvaf(pomgpu_ctx *c, int interior) { LAUNCH(c, k_ext_uvaf, grid2(c->P), blk2(), c->P, interior); }
void launch_ext_update(pomgpu_ctx *c) { LAUNCH(c, k_ext_update, grid2(c->P), blk2(), c->P); }
void launch_int_tail(pomgpu_ctx *c) { LAUNCH(c, k_int_tail, grid2(c->P), blk2(), c->P); }
void launch_bcond1(pomgpu_ctx *c) {
  LAUNCH(c, k_bcond1, grid2(c->P), blk2(), c->P);
  LAUNCH(c, k_copy2, grid2(c->P), blk2(), c->P, c->P.b2 + (size_t)P2_elf * c->P.n2, (const double *)c->P.s2[3]);
}
