// k_tile.hip -- wavefront-row stencil kernels.
//
// Profile of the first-cut cell kernels on MI355X (2048x1536x50): a cell of advt2 issued 56 global
// loads (each i+-1 / j+-1 neighbour a separate 512-byte wavefront request) and evaluated every face
// flux twice; the kernels ran at ~1.4-1.9 TB/s of algorithmic traffic, bound by the L1/TA request
// rate and fp64 divides, not by HBM.  Here a wavefront owns 64 consecutive i and MARCHES over a strip
// of rows at one level:
//   * i+-1 operands and the east face flux come from the neighbour lane (lane_w / lane_e), so a word
//     is requested once per wavefront, in aligned 512-byte rows;
//   * the strip keeps a two-row window in registers: the north face of row j is the south face of
//     row j+1, so every y-face flux (and its divide) is evaluated once;
//   * per-face metric sums/products come from the derived 2-D coefficient arrays (enum pomgpu_coef2).
// blockIdx.z = (row band, level) as for the cell kernels, so that a band's 2-D coefficients stay in
// L2 / Infinity Cache across the levels.
#include "pomgpu_internal.hpp"

#define ROWS_PER_STRIP 16

// ---- derived 2-D coefficients -------------------------------------------------------------------
__global__ void k_coef_static(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const int iw = i > 1 ? i - 1 : 1, js = j > 1 ? j - 1 : 1;
  K2(HSX, i, j) = F2(h, i, j) + F2(h, iw, j);
  K2(HSY, i, j) = F2(h, i, j) + F2(h, i, js);
  K2(DYSX, i, j) = F2(dy, i, j) + F2(dy, iw, j);
  K2(DXSX, i, j) = F2(dx, i, j) + F2(dx, iw, j);
  K2(DXSY, i, j) = F2(dx, i, j) + F2(dx, i, js);
  K2(DYSY, i, j) = F2(dy, i, j) + F2(dy, i, js);
  K2(DX4, i, j) = F2(dx, i, j) + F2(dx, iw, j) + F2(dx, i, js) + F2(dx, iw, js);
  K2(DY4, i, j) = F2(dy, i, j) + F2(dy, iw, j) + F2(dy, i, js) + F2(dy, iw, js);
  const int ie = i < P.iml ? i + 1 : P.iml, jn = j < P.jml ? j + 1 : P.jml;
  K2(CVA, i, j) = F2(dy, ie, j) - F2(dy, iw, j);
  K2(CVB, i, j) = F2(dx, i, jn) - F2(dx, i, js);
  K2(R2DXSX, i, j) = 2.0 / (F2(dx, i, j) + F2(dx, iw, j));
  K2(R2DYSY, i, j) = 2.0 / (F2(dy, i, j) + F2(dy, i, js));
}
__global__ void k_coef_dt(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const int iw = i > 1 ? i - 1 : 1, js = j > 1 ? j - 1 : 1;
  K2(CMX, i, j) = 0.25 * (F2(dy, iw, j) + F2(dy, i, j)) * (F2(dt, iw, j) + F2(dt, i, j));
  K2(CMY, i, j) = 0.25 * (F2(dx, i, js) + F2(dx, i, j)) * (F2(dt, i, js) + F2(dt, i, j));
  K2(DTSX, i, j) = F2(dt, i, j) + F2(dt, iw, j);
  K2(DTSY, i, j) = F2(dt, i, j) + F2(dt, i, js);
  K2(DT4, i, j) = F2(dt, i, j) + F2(dt, iw, j) + F2(dt, i, js) + F2(dt, iw, js);
}
__global__ void k_coef_eta(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  K2(HEA, i, j) = (F2(h, i, j) + F2(etb, i, j)) * F2(art, i, j);
  K2(HFA, i, j) = (F2(h, i, j) + F2(etf, i, j)) * F2(art, i, j);
}

// ---- advt2, nitera == 1 -- solver.f:577-731 with smol_adif's mask (:1898-1900) --------------------
__device__ __forceinline__ double upw_(double m, double lo, double hi) {     // solver.f:631-635
  return 0.5 * ((m + fabs(m)) * lo + (m - fabs(m)) * hi);
}
struct FaceT { double adv, dif; };
// west face of cell (i,j): operands of cell i ("c") and of cell i-1 ("w")
__device__ __forceinline__ FaceT advt2_xface(const KP &P, int i, int j, double uc, double fbc, double fbw, double fcc,
                                             double fcw, double amc, double amw) {
  FaceT f;
  f.adv = upw_(K2(CMX, i, j) * uc, fbw, fbc);                                  // :605-606, :631-635
  const double am = 0.5 * (amc + amw);                                         // :696
  f.dif = -am * K2(HSX, i, j) * P.tprni * ((fbc - fcc) - (fbw - fcw)) * F2(dum, i, j) * K2(DYSX, i, j) * 0.5 /
          K2(DXSX, i, j);                                                      // :705-707
  return f;
}
// south face of cell (i,j): operands of row j ("c") and of row j-1 ("s")
__device__ __forceinline__ FaceT advt2_yface(const KP &P, int i, int j, double vc, double fbc, double fbs, double fcc,
                                             double fcs, double amc, double ams) {
  FaceT f;
  f.adv = upw_(K2(CMY, i, j) * vc, fbs, fbc);                                  // :612-613, :637-641
  const double am = 0.5 * (amc + ams);                                         // :697
  f.dif = -am * K2(HSY, i, j) * P.tprni * ((fbc - fcc) - (fbs - fcs)) * F2(dvm, i, j) * K2(DXSY, i, j) * 0.5 /
          K2(DYSY, i, j);                                                      // :708-710
  return f;
}
__global__ void __launch_bounds__(256) k_advt2_rows(KP P, const double *fb, const double *f, const double *fcl, double *ff) {
  const int band = (int)blockIdx.z / P.kb;
  const int k = (int)blockIdx.z - band * P.kb + 1;
  const int strips_per_band = (int)(gridDim.y * blockDim.y);
  const int j0 = (band * strips_per_band + (int)(blockIdx.y * blockDim.y + threadIdx.y)) * ROWS_PER_STRIP + 1;
  if (j0 > P.jml) return;                                  // whole wavefront leaves together
  int j1 = j0 + ROWS_PER_STRIP - 1;
  if (j1 > P.jml) j1 = P.jml;
  const int i = TID_I;
  const bool icol = (i <= P.iml);
  const int ic = icol ? i : P.iml;                         // lanes past the array edge shadow the last column
  if (k > P.kbm1) {                                        // level kb: only smol_adif's mask
    if (icol)
      for (int j = j0; j <= j1; j++) G3(ff, ic, j, k) = G3(ff, ic, j, k) * F2(fsm, ic, j);
    return;
  }
  const bool iin = (i >= 2 && i <= P.imm1);
  const int iw = ic > 1 ? ic - 1 : 1, ie = ic < P.iml ? ic + 1 : P.iml;
  const double dzk = F1(dz, k);
  const bool top = (k == 1), bot = (k == P.kbm1);
  // two-row window of the operands that the y faces need
  int js = j0 > 1 ? j0 - 1 : 1;
  double fb_s = G3(fb, ic, js, k), fc_s = G3(fcl, ic, js, k), am_s = F3(aam, ic, js, k);
  double fb_c = G3(fb, ic, j0, k), fc_c = G3(fcl, ic, j0, k), am_c = F3(aam, ic, j0, k);
  FaceT ys = advt2_yface(P, ic, j0, F3(v, ic, j0, k), fb_c, fb_s, fc_c, fc_s, am_c, am_s);
  for (int j = j0; j <= j1; j++) {
    const int jn = j < P.jml ? j + 1 : P.jml;
    const double fb_n = G3(fb, ic, jn, k), fc_n = G3(fcl, ic, jn, k), am_n = F3(aam, ic, jn, k);
    const FaceT yn = advt2_yface(P, ic, jn, F3(v, ic, jn, k), fb_n, fb_c, fc_n, fc_c, am_n, am_c);
    // west face from the neighbour lane's operands; east face = the eastern lane's west face
    const double fb_w = lane_w(fb_c, [&] { return G3(fb, iw, j, k); });
    const double fc_w = lane_w(fc_c, [&] { return G3(fcl, iw, j, k); });
    const double am_w = lane_w(am_c, [&] { return F3(aam, iw, j, k); });
    const FaceT xw = advt2_xface(P, ic, j, F3(u, ic, j, k), fb_c, fb_w, fc_c, fc_w, am_c, am_w);
    FaceT xe;
    xe.adv = lane_e(xw.adv, [&] {
      return advt2_xface(P, ie, j, F3(u, ie, j, k), G3(fb, ie, j, k), fb_c, G3(fcl, ie, j, k), fc_c, F3(aam, ie, j, k), am_c).adv;
    });
    xe.dif = lane_e(xw.dif, [&] {
      return advt2_xface(P, ie, j, F3(u, ie, j, k), G3(fb, ie, j, k), fb_c, G3(fcl, ie, j, k), fc_c, F3(aam, ie, j, k), am_c).dif;
    });
    if (icol) {
      double r;
      if (iin && j >= 2 && j <= P.jmm1) {
        const double art = F2(art, ic, j);
        const double zu = top ? F3(w, ic, j, 1) * G3(f, ic, j, 1) * art
                              : upw_(F3(w, ic, j, k), fb_c, G3(fb, ic, j, k - 1)) * art;       // :646-662
        const double zl = bot ? 0. : upw_(F3(w, ic, j, k + 1), G3(fb, ic, j, k + 1), fb_c) * art;
        const double hfa = K2(HFA, ic, j);
        r = xe.adv - xw.adv + yn.adv - ys.adv + (zu - zl) / dzk;                              // :670-672
        r = (fb_c * K2(HEA, ic, j) - P.dti2 * r) / hfa;                                       // :673-674
        r = r * F2(fsm, ic, j);                                                               // :1899
        r = r - P.dti2 * (xe.dif - xw.dif + yn.dif - ys.dif) / hfa;                           // :721-723
      } else {
        r = G3(ff, ic, j, k) * F2(fsm, ic, j);                                                // :1899 (rim cells)
      }
      G3(ff, ic, j, k) = r;
    }
    fb_s = fb_c; fc_s = fc_c; am_s = am_c;
    fb_c = fb_n; fc_c = fc_n; am_c = am_n;
    ys = yn;
  }
}

// ---- advt2, nitera == 1, COLUMN-MARCHING version -----------------------------------------------------
// PMC profile of the row-marching kernel above (2048x1536x50): 31 GB of L2-miss traffic per launch
// against 7.5 GB algorithmic -- the 14 two-dimensional coefficient operands of a cell are re-fetched
// for every level -- and ~1e9 L1 line accesses.  Here a thread owns the water column (i,j):
//   * all 2-D coefficients of its three faces live in registers for the whole column;
//   * the level loop is software-pipelined: iteration L issues the loads of level L+1, evaluates
//     the faces of level L from registers filled one iteration earlier, and finishes level L-1
//     (whose bottom face is the top face of level L, evaluated once);
//   * west operands and the east face come from the neighbour lane (lane_w / lane_e).
// Per cell: 13 aligned 512-byte wavefront loads (was 56 in the cell kernel), 3 face evaluations
// in x/y (was 4) and 1 in z (was 2).
struct LevT {
  double fb_c, fb_s, fb_n, fc_c, fc_s, fc_n, am_c, am_s, am_n, u_c, v_c, v_n, w_c;
  double fb_w, fc_w, am_w;        // west operands: only lane 0 loads them (the others get them by shuffle)
  double fb_e, fc_e, am_e, u_e;   // east-face operands: only the last lane loads them
};
// every load of a level is issued here, in one batch, BEFORE the level that is being computed
// needs anything: vmcnt counts in order, so a load issued in the middle of the arithmetic would
// make the wave wait for the whole prefetch batch of the next level
__device__ __forceinline__ LevT advt2_load(const KP &P, const double *fb, const double *fcl, int iw, int i, int ie, int js, int j,
                                           int jn, int k) {
  LevT L;
  L.fb_c = G3(fb, i, j, k);  L.fb_s = G3(fb, i, js, k);  L.fb_n = G3(fb, i, jn, k);
  L.fc_c = G3(fcl, i, j, k); L.fc_s = G3(fcl, i, js, k); L.fc_n = G3(fcl, i, jn, k);
  L.am_c = F3(aam, i, j, k); L.am_s = F3(aam, i, js, k); L.am_n = F3(aam, i, jn, k);
  L.u_c = F3(u, i, j, k);    L.v_c = F3(v, i, j, k);     L.v_n = F3(v, i, jn, k);
  L.w_c = F3(w, i, j, k);
  L.fb_w = L.fc_w = L.am_w = L.fb_e = L.fc_e = L.am_e = L.u_e = 0.;
  if (threadIdx.x == 0) { L.fb_w = G3(fb, iw, j, k); L.fc_w = G3(fcl, iw, j, k); L.am_w = F3(aam, iw, j, k); }
  if (threadIdx.x == blockDim.x - 1) {
    L.fb_e = G3(fb, ie, j, k); L.fc_e = G3(fcl, ie, j, k); L.am_e = F3(aam, ie, j, k); L.u_e = F3(u, ie, j, k);
  }
  return L;
}
struct CoefT { double cm, hs, msk, ds_num, ds_den; };   // mass-flux coefficient, h sum, mask, metric sums of one face
__device__ __forceinline__ CoefT coef_x(const KP &P, int i, int j) {
  CoefT c; c.cm = K2(CMX, i, j); c.hs = K2(HSX, i, j); c.msk = F2(dum, i, j); c.ds_num = K2(DYSX, i, j); c.ds_den = K2(DXSX, i, j); return c;
}
__device__ __forceinline__ CoefT coef_y(const KP &P, int i, int j) {
  CoefT c; c.cm = K2(CMY, i, j); c.hs = K2(HSY, i, j); c.msk = F2(dvm, i, j); c.ds_num = K2(DXSY, i, j); c.ds_den = K2(DYSY, i, j); return c;
}
// face between a "lo" cell (west / south) and a "hi" cell; vel = u or v on that face
__device__ __forceinline__ FaceT advt2_face(const KP &P, const CoefT &c, double vel, double fb_hi, double fb_lo, double fc_hi,
                                            double fc_lo, double am_hi, double am_lo) {
  FaceT f;
  f.adv = upw_(c.cm * vel, fb_lo, fb_hi);
  const double am = 0.5 * (am_hi + am_lo);
  f.dif = -am * c.hs * P.tprni * ((fb_hi - fc_hi) - (fb_lo - fc_lo)) * c.msk * c.ds_num * 0.5 / c.ds_den;
  return f;
}
__global__ void __launch_bounds__(256) k_advt2_col(KP P, const double *fb, const double *f, const double *fcl, double *ff) {
  const int i0 = TID_I, j0 = TID_J;
  if (j0 > P.jml) return;                                   // whole wavefront (one row) leaves together
  const bool icol = (i0 <= P.iml);
  const int i = icol ? i0 : P.iml, j = j0;                  // lanes past the array edge shadow the last column
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const int js = j > 1 ? j - 1 : 1, jn = j < P.jml ? j + 1 : P.jml;
  const bool in = icol && (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1);
  const bool last = (threadIdx.x == blockDim.x - 1);
  const double fsm = F2(fsm, i, j);
  // column-resident coefficients of the west, south and north faces, of the east face on the last
  // lane, and of the cell
  const CoefT cw = coef_x(P, i, j), cs = coef_y(P, i, j), cn = coef_y(P, i, jn);
  CoefT ce = cw;
  if (last) ce = coef_x(P, ie, j);
  const double art = F2(art, i, j), hea = K2(HEA, i, j), hfa = K2(HFA, i, j);
  const double f1 = G3(f, i, j, 1);
  const int kbm1 = P.kbm1;
  LevT cur = advt2_load(P, fb, fcl, iw, i, ie, js, j, jn, 1), nxt = cur;
  // carried from level L-1 to its completion in iteration L
  double p_adv = 0., p_dif = 0., p_fb = 0., p_zu = 0.;
  for (int L = 1; L <= kbm1 + 1; L++) {
    if (L + 1 <= kbm1) nxt = advt2_load(P, fb, fcl, iw, i, ie, js, j, jn, L + 1);   // in flight during this iteration
    const double ffk = (L >= 2 && !in && icol) ? G3(ff, i, j, L - 1) : 0.;
    double zu = 0.;                                                                // top face of level L (0 below kbm1)
    double s_adv = 0., s_dif = 0.;
    if (L <= kbm1) {
      const double fb_w = lane_w(cur.fb_c, [&] { return cur.fb_w; });
      const double fc_w = lane_w(cur.fc_c, [&] { return cur.fc_w; });
      const double am_w = lane_w(cur.am_c, [&] { return cur.am_w; });
      const FaceT xw = advt2_face(P, cw, cur.u_c, cur.fb_c, fb_w, cur.fc_c, fc_w, cur.am_c, am_w);
      FaceT xl;
      xl.adv = xl.dif = 0.;
      if (last) xl = advt2_face(P, ce, cur.u_e, cur.fb_e, cur.fb_c, cur.fc_e, cur.fc_c, cur.am_e, cur.am_c);
      FaceT xe;
      xe.adv = lane_e(xw.adv, [&] { return xl.adv; });
      xe.dif = lane_e(xw.dif, [&] { return xl.dif; });
      const FaceT ys = advt2_face(P, cs, cur.v_c, cur.fb_c, cur.fb_s, cur.fc_c, cur.fc_s, cur.am_c, cur.am_s);
      const FaceT yn = advt2_face(P, cn, cur.v_n, cur.fb_n, cur.fb_c, cur.fc_n, cur.fc_c, cur.am_n, cur.am_c);
      s_adv = xe.adv - xw.adv + yn.adv - ys.adv;                                              // solver.f:670-671
      s_dif = xe.dif - xw.dif + yn.dif - ys.dif;                                              // :721-722
      zu = (L == 1) ? cur.w_c * f1 * art : upw_(cur.w_c, cur.fb_c, p_fb) * art;               // :646-662
    }
    if (L >= 2 && icol) {                                   // finish level L-1: its bottom face is this level's top face
      const int k = L - 1;
      double r;
      if (in) {
        r = p_adv + (p_zu - zu) / F1(dz, k);                                                  // :670-672
        r = (p_fb * hea - P.dti2 * r) / hfa;                                                  // :673-674
        r = r * fsm;                                                                          // :1899
        r = r - P.dti2 * p_dif / hfa;                                                         // :721-723
      } else {
        r = ffk * fsm;                                                                        // :1899 (rim cells)
      }
      G3(ff, i, j, k) = r;
    }
    p_adv = s_adv; p_dif = s_dif; p_fb = cur.fb_c; p_zu = zu;
    cur = nxt;
  }
  if (icol) G3(ff, i, j, P.kb) = G3(ff, i, j, P.kb) * fsm;                                    // :1899, level kb
}

// ---- advq, flux + step fused (single tile) -- solver.f:411-477 -----------------------------------------
// Same structure as k_advt2_col: column-resident face coefficients, software-pipelined level loop,
// neighbour-lane operands.  The reference exchanges xflux/yflux between the flux and the step
// loops (:458-459); with all neighbours -1 that exchange is a no-op and the two halves fuse: the
// fluxes never reach memory.  Used only when the context has no exchange hook (one tile).
struct LevQ {
  double q_c, q_s, q_n, qb_c, qb_s, qb_n, am_c, am_s, am_n, u_c, v_c, v_n, w_c;
  double q_w, qb_w, am_w;         // lane 0 only
  double q_e, qb_e, am_e, u_e;    // last lane only
};
__device__ __forceinline__ LevQ advq_load(const KP &P, const double *q, const double *qb, int iw, int i, int ie, int js, int j, int jn,
                                          int k) {
  LevQ L;
  L.q_c = G3(q, i, j, k);    L.q_s = G3(q, i, js, k);    L.q_n = G3(q, i, jn, k);
  L.qb_c = G3(qb, i, j, k);  L.qb_s = G3(qb, i, js, k);  L.qb_n = G3(qb, i, jn, k);
  L.am_c = F3(aam, i, j, k); L.am_s = F3(aam, i, js, k); L.am_n = F3(aam, i, jn, k);
  L.u_c = F3(u, i, j, k);    L.v_c = F3(v, i, j, k);     L.v_n = F3(v, i, jn, k);
  L.w_c = F3(w, i, j, k);
  L.q_w = L.qb_w = L.am_w = L.q_e = L.qb_e = L.am_e = L.u_e = 0.;
  if (threadIdx.x == 0) { L.q_w = G3(q, iw, j, k); L.qb_w = G3(qb, iw, j, k); L.am_w = F3(aam, iw, j, k); }
  if (threadIdx.x == blockDim.x - 1) {
    L.q_e = G3(q, ie, j, k); L.qb_e = G3(qb, ie, j, k); L.am_e = F3(aam, ie, j, k); L.u_e = F3(u, ie, j, k);
  }
  return L;
}
struct CoefQ { double dts, hs, msk, ds_den, ds_num; };
__device__ __forceinline__ CoefQ coefq_x(const KP &P, int i, int j) {
  CoefQ c; c.dts = K2(DTSX, i, j); c.hs = K2(HSX, i, j); c.msk = F2(dum, i, j); c.ds_den = K2(DXSX, i, j); c.ds_num = K2(DYSX, i, j); return c;
}
__device__ __forceinline__ CoefQ coefq_y(const KP &P, int i, int j) {
  CoefQ c; c.dts = K2(DTSY, i, j); c.hs = K2(HSY, i, j); c.msk = F2(dvm, i, j); c.ds_den = K2(DYSY, i, j); c.ds_num = K2(DXSY, i, j); return c;
}
// flux through the face between a "lo" (west/south) and a "hi" cell at w-level k (:428-453)
__device__ __forceinline__ double advq_face(const CoefQ &c, double q_hi, double q_lo, double vel_k, double vel_km1, double am_hi_k,
                                            double am_lo_k, double am_hi_m, double am_lo_m, double qb_hi, double qb_lo) {
  double x = .125 * (q_hi + q_lo) * c.dts * (vel_k + vel_km1);
  x = x - .25 * (am_hi_k + am_lo_k + am_hi_m + am_lo_m) * c.hs * (qb_hi - qb_lo) * c.msk / c.ds_den;
  return .5 * c.ds_num * x;
}
__global__ void __launch_bounds__(256) k_advq_col(KP P, const double *q, const double *qb, double *qf, int zero_else) {
  const int i0 = TID_I, j0 = TID_J;
  if (j0 > P.jml) return;
  const bool icol = (i0 <= P.iml);
  const int i = icol ? i0 : P.iml, j = j0;
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const int js = j > 1 ? j - 1 : 1, jn = j < P.jml ? j + 1 : P.jml;
  const bool in = icol && (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1);
  const bool last = (threadIdx.x == blockDim.x - 1);
  const CoefQ cw = coefq_x(P, i, j), cs = coefq_y(P, i, j), cn = coefq_y(P, i, jn);
  CoefQ ce = cw;
  if (last) ce = coefq_x(P, ie, j);
  const double art = F2(art, i, j), hea = K2(HEA, i, j), hfa = K2(HFA, i, j);
  const int kb = P.kb, kbm1 = P.kbm1;
  LevQ cur = advq_load(P, q, qb, iw, i, ie, js, j, jn, 1), nxt = cur, prv = cur;
  double am_w_prv = 0.;                       // aam(i-1,j,L-1) as seen by this lane
  double wq_pp = 0., wq_p = 0.;               // w*q of levels L-2 and L-1
  double xe_p = 0., xw_p = 0., yn_p = 0., ys_p = 0., qb_p = 0.;   // faces and qb of level L-1, waiting for w(L)*q(L)
  for (int L = 1; L <= kb; L++) {
    if (L + 1 <= kb) nxt = advq_load(P, q, qb, iw, i, ie, js, j, jn, L + 1);
    const double am_w = lane_w(cur.am_c, [&] { return cur.am_w; });
    double xe_c = 0., xw_c = 0., yn_c = 0., ys_c = 0.;
    if (L >= 2 && L <= kbm1) {
      const double q_w = lane_w(cur.q_c, [&] { return cur.q_w; });
      const double qb_w = lane_w(cur.qb_c, [&] { return cur.qb_w; });
      const double xw = advq_face(cw, cur.q_c, q_w, cur.u_c, prv.u_c, cur.am_c, am_w, prv.am_c, am_w_prv, cur.qb_c, qb_w);
      double xl = 0.;
      if (last) xl = advq_face(ce, cur.q_e, cur.q_c, cur.u_e, prv.u_e, cur.am_e, cur.am_c, prv.am_e, prv.am_c, cur.qb_e, cur.qb_c);
      xw_c = xw;
      xe_c = lane_e(xw, [&] { return xl; });
      ys_c = advq_face(cs, cur.q_c, cur.q_s, cur.v_c, prv.v_c, cur.am_c, cur.am_s, prv.am_c, prv.am_s, cur.qb_c, cur.qb_s);
      yn_c = advq_face(cn, cur.q_n, cur.q_c, cur.v_n, prv.v_n, cur.am_n, cur.am_c, prv.am_n, prv.am_c, cur.qb_n, cur.qb_c);
    }
    const double wq_c = cur.w_c * cur.q_c;
    if (icol) {
      const int k = L - 1;                                   // level completed in this iteration
      if (in && k >= 2 && k <= kbm1) {
        double r = (wq_pp - wq_c) * art / (F1(dz, k) + F1(dz, k - 1)) + xe_p - xw_p + yn_p - ys_p;   // :465-468
        r = (hea * qb_p - P.dti2 * r) / hfa;                                                    // :469-471
        G3(qf, i, j, k) = r;
      } else if (zero_else && k >= 1) {
        G3(qf, i, j, k) = 0.;
      }
    }
    wq_pp = wq_p; wq_p = wq_c; qb_p = cur.qb_c;
    xe_p = xe_c; xw_p = xw_c; yn_p = yn_c; ys_p = ys_c;
    am_w_prv = am_w;
    prv = cur; cur = nxt;
  }
  if (icol && zero_else) G3(qf, i, j, kb) = 0.;
}

// ---- launchers ------------------------------------------------------------------------------------
void launch_coef_static(pomgpu_ctx *c) { LAUNCH(c, k_coef_static, grid2(c->P), blk2(), c->P); }
void launch_coef_dt(pomgpu_ctx *c) { LAUNCH(c, k_coef_dt, grid2(c->P), blk2(), c->P); }
void launch_coef_eta(pomgpu_ctx *c) { LAUNCH(c, k_coef_eta, grid2(c->P), blk2(), c->P); }
// strips of ROWS_PER_STRIP rows; bands as gridm() but counted in strips
static dim3 grid_rows(const KP &P) {
  long rows = (4L << 20) / ((long)P.iml * 8);
  if (rows > P.jml) rows = P.jml;
  long strips = (rows + ROWS_PER_STRIP - 1) / ROWS_PER_STRIP;
  strips = ((strips + 3) / 4) * 4;                          // 4 wavefronts per workgroup
  const long band_rows = strips * ROWS_PER_STRIP;
  const int nbands = (int)((P.jml + band_rows - 1) / band_rows);
  return dim3((P.iml + 63) / 64, (unsigned)(strips / 4), (unsigned)(nbands * P.kb));
}
void launch_advq_col(pomgpu_ctx *c, const double *q, const double *qb, double *qf, int zero_else) {
  LAUNCH(c, k_advq_col, grid2(c->P), blk2(), c->P, q, qb, qf, zero_else);
}
void launch_advt2_rows(pomgpu_ctx *c, const double *fb, const double *f, const double *fc, double *ff) {
  if (getenv("POMGPU_ADVT2_ROWS")) LAUNCH(c, k_advt2_rows, grid_rows(c->P), dim3(64, 4, 1), c->P, fb, f, fc, ff);
  else LAUNCH(c, k_advt2_col, grid2(c->P), blk2(), c->P, fb, f, fc, ff);
}
