// k_bc.hip -- open-boundary conditions that run inside the internal mode: bcond(4), bcond(6)
// (bounds_forcing.f:151-324) and bcondorl(3) (:418-487).  Only the cells on a physical edge are
// touched, so these kernels are launched over the four edge lines (blockIdx.y selects the line),
// not over the field; the whole-field mask multiplies that end each branch are fused into the
// Asselin-filter passes that follow (k_adv.hip).
//
// Corner precedence: the reference sweeps east/west inside its j-loop and then south/north inside
// its i-loop, so at a corner cell the south/north value wins; the east/west lines therefore skip
// the corner cells that a physical south/north edge also owns.
#include "pomgpu_internal.hpp"

#define EDGE_CELL(ncorner_skip)                                                              \
  const int t = TID_I, line = (int)blockIdx.y, k = TID_K;                                    \
  int i, j;                                                                                  \
  if (line == 0) { if (!P.W || t > P.jm) return; i = 1; j = t; }                             \
  else if (line == 1) { if (!P.E || t > P.jm) return; i = P.im; j = t; }                     \
  else if (line == 2) { if (!P.S || t > P.im) return; i = t; j = 1; }                        \
  else { if (!P.N || t > P.im) return; i = t; j = P.jm; }                                    \
  if (ncorner_skip && line < 2 && ((P.S && j == 1) || (P.N && j == P.jm))) return;

// bcond(4): upstream / inflow T,S on the open edges (uf = T, vf = S) -- bounds_forcing.f:155-231
__global__ void k_bcond4_edges(KP P) {
  EDGE_CELL(1)
  if (k > P.kbm1) return;
  const bool vert = (k != 1 && k != P.kbm1);
  double u1, wm, tv, sv;
  if (line == 1) {          // east
    u1 = 2. * F3(u, P.im, j, k) * P.dti / (F2(dx, P.im, j) + F2(dx, P.imm1, j));
    if (u1 <= 0.) {
      tv = F3(t, P.im, j, k) - u1 * (BDJ(tbe, j, k) - F3(t, P.im, j, k));
      sv = F3(s, P.im, j, k) - u1 * (BDJ(sbe, j, k) - F3(s, P.im, j, k));
    } else {
      tv = F3(t, P.im, j, k) - u1 * (F3(t, P.im, j, k) - F3(t, P.imm1, j, k));
      sv = F3(s, P.im, j, k) - u1 * (F3(s, P.im, j, k) - F3(s, P.imm1, j, k));
      if (vert) {
        wm = .5 * (F3(w, P.imm1, j, k) + F3(w, P.imm1, j, k + 1)) * P.dti / ((F1(zz, k - 1) - F1(zz, k + 1)) * F2(dt, P.imm1, j));
        tv = tv - wm * (F3(t, P.imm1, j, k - 1) - F3(t, P.imm1, j, k + 1));
        sv = sv - wm * (F3(s, P.imm1, j, k - 1) - F3(s, P.imm1, j, k + 1));
      }
    }
  } else if (line == 0) {   // west
    u1 = 2. * F3(u, 2, j, k) * P.dti / (F2(dx, 1, j) + F2(dx, 2, j));
    if (u1 >= 0.) {
      tv = F3(t, 1, j, k) - u1 * (F3(t, 1, j, k) - BDJ(tbw, j, k));
      sv = F3(s, 1, j, k) - u1 * (F3(s, 1, j, k) - BDJ(sbw, j, k));
    } else {
      tv = F3(t, 1, j, k) - u1 * (F3(t, 2, j, k) - F3(t, 1, j, k));
      sv = F3(s, 1, j, k) - u1 * (F3(s, 2, j, k) - F3(s, 1, j, k));
      if (vert) {
        wm = .5 * (F3(w, 2, j, k) + F3(w, 2, j, k + 1)) * P.dti / ((F1(zz, k - 1) - F1(zz, k + 1)) * F2(dt, 2, j));
        tv = tv - wm * (F3(t, 2, j, k - 1) - F3(t, 2, j, k + 1));
        sv = sv - wm * (F3(s, 2, j, k - 1) - F3(s, 2, j, k + 1));
      }
    }
  } else if (line == 2) {   // south
    u1 = 2. * F3(v, i, 2, k) * P.dti / (F2(dy, i, 1) + F2(dy, i, 2));
    if (u1 >= 0.) {
      tv = F3(t, i, 1, k) - u1 * (F3(t, i, 1, k) - BDI(tbs, i, k));
      sv = F3(s, i, 1, k) - u1 * (F3(s, i, 1, k) - BDI(sbs, i, k));
    } else {
      tv = F3(t, i, 1, k) - u1 * (F3(t, i, 2, k) - F3(t, i, 1, k));
      sv = F3(s, i, 1, k) - u1 * (F3(s, i, 2, k) - F3(s, i, 1, k));
      if (vert) {
        wm = .5 * (F3(w, i, 2, k) + F3(w, i, 2, k + 1)) * P.dti / ((F1(zz, k - 1) - F1(zz, k + 1)) * F2(dt, i, 2));
        tv = tv - wm * (F3(t, i, 2, k - 1) - F3(t, i, 2, k + 1));
        sv = sv - wm * (F3(s, i, 2, k - 1) - F3(s, i, 2, k + 1));
      }
    }
  } else {                  // north
    u1 = 2. * F3(v, i, P.jm, k) * P.dti / (F2(dy, i, P.jm) + F2(dy, i, P.jmm1));
    if (u1 <= 0.) {
      tv = F3(t, i, P.jm, k) - u1 * (BDI(tbn, i, k) - F3(t, i, P.jm, k));
      sv = F3(s, i, P.jm, k) - u1 * (BDI(sbn, i, k) - F3(s, i, P.jm, k));
    } else {
      tv = F3(t, i, P.jm, k) - u1 * (F3(t, i, P.jm, k) - F3(t, i, P.jmm1, k));
      sv = F3(s, i, P.jm, k) - u1 * (F3(s, i, P.jm, k) - F3(s, i, P.jmm1, k));
      if (vert) {
        wm = .5 * (F3(w, i, P.jmm1, k) + F3(w, i, P.jmm1, k + 1)) * P.dti / ((F1(zz, k - 1) - F1(zz, k + 1)) * F2(dt, i, P.jmm1));
        tv = tv - wm * (F3(t, i, P.jmm1, k - 1) - F3(t, i, P.jmm1, k + 1));
        sv = sv - wm * (F3(s, i, P.jmm1, k - 1) - F3(s, i, P.jmm1, k + 1));
      }
    }
  }
  F3(uf, i, j, k) = tv;
  F3(vf, i, j, k) = sv;
}

// bcond(6): upstream q2, q2l on the open edges (uf = q2, vf = q2l), k = 1..kb -- :261-313
__global__ void k_bcond6_edges(KP P) {
  EDGE_CELL(1)
  if (k > P.kb) return;
  double u1, qv, lv;
  if (line == 0) {
    u1 = 2. * F3(u, 2, j, k) * P.dti / (F2(dx, 1, j) + F2(dx, 2, j));
    if (u1 >= 0.) {
      qv = F3(q2, 1, j, k) - u1 * (F3(q2, 1, j, k) - P.small_);
      lv = F3(q2l, 1, j, k) - u1 * (F3(q2l, 1, j, k) - P.small_);
    } else {
      qv = F3(q2, 1, j, k) - u1 * (F3(q2, 2, j, k) - F3(q2, 1, j, k));
      lv = F3(q2l, 1, j, k) - u1 * (F3(q2l, 2, j, k) - F3(q2l, 1, j, k));
    }
  } else if (line == 1) {
    u1 = 2. * F3(u, P.im, j, k) * P.dti / (F2(dx, P.im, j) + F2(dx, P.imm1, j));
    if (u1 <= 0.) {
      qv = F3(q2, P.im, j, k) - u1 * (P.small_ - F3(q2, P.im, j, k));
      lv = F3(q2l, P.im, j, k) - u1 * (P.small_ - F3(q2l, P.im, j, k));
    } else {
      qv = F3(q2, P.im, j, k) - u1 * (F3(q2, P.im, j, k) - F3(q2, P.imm1, j, k));
      lv = F3(q2l, P.im, j, k) - u1 * (F3(q2l, P.im, j, k) - F3(q2l, P.imm1, j, k));
    }
  } else if (line == 2) {
    u1 = 2. * F3(v, i, 2, k) * P.dti / (F2(dy, i, 1) + F2(dy, i, 2));
    if (u1 >= 0.) {
      qv = F3(q2, i, 1, k) - u1 * (F3(q2, i, 1, k) - P.small_);
      lv = F3(q2l, i, 1, k) - u1 * (F3(q2l, i, 1, k) - P.small_);
    } else {
      qv = F3(q2, i, 1, k) - u1 * (F3(q2, i, 2, k) - F3(q2, i, 1, k));
      lv = F3(q2l, i, 1, k) - u1 * (F3(q2l, i, 2, k) - F3(q2l, i, 1, k));
    }
  } else {
    u1 = 2. * F3(v, i, P.jm, k) * P.dti / (F2(dy, i, P.jm) + F2(dy, i, P.jmm1));
    if (u1 <= 0.) {
      qv = F3(q2, i, P.jm, k) - u1 * (P.small_ - F3(q2, i, P.jm, k));
      lv = F3(q2l, i, P.jm, k) - u1 * (P.small_ - F3(q2l, i, P.jm, k));
    } else {
      qv = F3(q2, i, P.jm, k) - u1 * (F3(q2, i, P.jm, k) - F3(q2, i, P.jmm1, k));
      lv = F3(q2l, i, P.jm, k) - u1 * (F3(q2l, i, P.jm, k) - F3(q2l, i, P.jmm1, k));
    }
  }
  F3(uf, i, j, k) = qv;
  F3(vf, i, j, k) = lv;
}

// bcondorl(3): Orlanski radiation for uf, vf -- bounds_forcing.f:422-476, then the dum/dvm mask
// (:478-485) on the rim cells only: every interior cell of uf/vf already carries its mask factor
// from profu/profv (masks are 0/1, so the second multiply of the reference changes nothing).
// Lines 0..5: i=1, i=2, i=im, j=1, j=2, j=jm.
__device__ __forceinline__ double orl(double fb_in, double ff_in, double f_in2, double fb_edge, double f_in) {
  double denom = (ff_in + fb_in - 2. * f_in2);
  if (denom == 0.) denom = 0.01;
  double cl = (fb_in - ff_in) / denom;
  if (cl > 1.) cl = 1.;
  if (cl < 0.) cl = 0.;
  return (fb_edge * (1. - cl) + 2. * cl * f_in) / (1. + cl);
}
__global__ void k_bcondorl3(KP P) {
  const int t = TID_I, line = (int)blockIdx.y, k = TID_K;
  if (k > P.kbm1) return;
  int i, j;
  if (line < 3) { if (t > P.jm) return; j = t; i = (line == 0) ? 1 : ((line == 1) ? 2 : P.im); }
  else { if (t > P.im) return; i = t; j = (line == 3) ? 1 : ((line == 4) ? 2 : P.jm); }
  const bool jin = (j >= 2 && j <= P.jmm1), iin = (i >= 2 && i <= P.imm1);
  double uf = F3(uf, i, j, k), vf = F3(vf, i, j, k);
  if (P.E && jin && i == P.im) {                                                            // :425-434
    uf = orl(F3(ub, P.im - 1, j, k), F3(uf, P.im - 1, j, k), F3(u, P.im - 2, j, k), F3(ub, P.im, j, k), F3(u, P.im - 1, j, k));
    vf = 0.;
  }
  if (P.W && jin && (i == 1 || i == 2)) {                                                   // :437-447
    uf = orl(F3(ub, 3, j, k), F3(uf, 3, j, k), F3(u, 4, j, k), F3(ub, 2, j, k), F3(u, 3, j, k));
    if (i == 1) vf = 0.;
  }
  if (P.S && iin && (j == 1 || j == 2)) {                                                   // :452-462
    vf = orl(F3(vb, i, 3, k), F3(vf, i, 3, k), F3(v, i, 4, k), F3(vb, i, 2, k), F3(v, i, 3, k));
    if (j == 1) uf = 0.;
  }
  if (P.N && iin && j == P.jm) {                                                            // :465-474
    vf = orl(F3(vb, i, P.jm - 1, k), F3(vf, i, P.jm - 1, k), F3(v, i, P.jm - 2, k), F3(vb, i, P.jm, k), F3(v, i, P.jm - 1, k));
    uf = 0.;
  }
  F3(uf, i, j, k) = uf * F2(dum, i, j);
  F3(vf, i, j, k) = vf * F2(dvm, i, j);
}

// ---------------------------------------------------------------------------------------------
// lateral_bc -- bounds_forcing.f:593-868.  One thread per edge point: t < jm_local serves the west and
// east arrays at j = t+1, the others the north and south arrays at i.  phase 0: a record (20 arrays
// concatenated, in the order of read_boundary_conditions_pnetcdf's arguments) lands in the "...f" members
// and el?, followed by the depth integrals uab?f, vab?f (:610-636, :755-771); phase 1: the "b" copies that
// the reference refreshes (:742-753 -- tb?b, sb?b, uabwb, uabeb, vabnb, vabsb only); phase 2: interpolation
// in time and the integrals uabe, uabw, vabn, vabs (:776-797).
__global__ void k_lateral(KP P, int phase, const double *rec, double fold, double fnew) {
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int kb = P.kb, jml = P.jml, iml = P.iml;
  if (t >= jml + iml) return;
  const bool west_east = (t < jml);
  const int a = west_east ? t + 1 : t - jml + 1;                      // j or i, 1-based
  const size_t njk = (size_t)jml * kb, nik = (size_t)iml * kb;
  if (phase == 0) {
    if (west_east) {
      const double *r = rec;                                         // tbwf sbwf ubwf vbwf tbef sbef ubef vbef: 8 x (jml,kb)
      double uw = 0., vw = 0., ue = 0., ve = 0.;
      for (int k = 1; k <= kb; k++) {
        const size_t o = (size_t)(k - 1) * jml + (a - 1);
        const double dzk = F1(dz, k);
        BDJ(tbwf, a, k) = r[o]; BDJ(sbwf, a, k) = r[njk + o];
        const double x = r[2 * njk + o], y = r[3 * njk + o];
        BDJ(ubwf, a, k) = x; BDJ(vbwf, a, k) = y;
        uw = uw + x * dzk; vw = vw + y * dzk;
        BDJ(tbef, a, k) = r[4 * njk + o]; BDJ(sbef, a, k) = r[5 * njk + o];
        const double p = r[6 * njk + o], q = r[7 * njk + o];
        BDJ(ubef, a, k) = p; BDJ(vbef, a, k) = q;
        ue = ue + p * dzk; ve = ve + q * dzk;
      }
      BD1(uabwf, a) = uw; BD1(vabwf, a) = vw; BD1(uabef, a) = ue; BD1(vabef, a) = ve;
      const double *e = rec + 8 * njk + 8 * nik;                     // elw ele (jml each), then eln els
      BD1(elw, a) = e[a - 1]; BD1(ele, a) = e[jml + a - 1];
    } else {
      const double *r = rec + 8 * njk;                               // tbnf sbnf vbnf ubnf tbsf sbsf vbsf ubsf: 8 x (iml,kb)
      double un = 0., vn = 0., us = 0., vs = 0.;
      for (int k = 1; k <= kb; k++) {
        const size_t o = (size_t)(k - 1) * iml + (a - 1);
        const double dzk = F1(dz, k);
        BDI(tbnf, a, k) = r[o]; BDI(sbnf, a, k) = r[nik + o];
        const double y = r[2 * nik + o], x = r[3 * nik + o];
        BDI(vbnf, a, k) = y; BDI(ubnf, a, k) = x;
        un = un + x * dzk; vn = vn + y * dzk;
        BDI(tbsf, a, k) = r[4 * nik + o]; BDI(sbsf, a, k) = r[5 * nik + o];
        const double q = r[6 * nik + o], p = r[7 * nik + o];
        BDI(vbsf, a, k) = q; BDI(ubsf, a, k) = p;
        us = us + p * dzk; vs = vs + q * dzk;
      }
      BD1(uabnf, a) = un; BD1(vabnf, a) = vn; BD1(uabsf, a) = us; BD1(vabsf, a) = vs;
      const double *e = rec + 8 * njk + 8 * nik + 2 * (size_t)jml;
      BD1(eln, a) = e[a - 1]; BD1(els, a) = e[iml + a - 1];
    }
  } else if (phase == 1) {
    if (west_east) {
      for (int k = 1; k <= kb; k++) {
        BDJ(tbwb, a, k) = BDJ(tbwf, a, k); BDJ(sbwb, a, k) = BDJ(sbwf, a, k);
        BDJ(tbeb, a, k) = BDJ(tbef, a, k); BDJ(sbeb, a, k) = BDJ(sbef, a, k);
      }
      BD1(uabwb, a) = BD1(uabwf, a); BD1(uabeb, a) = BD1(uabef, a);
    } else {
      for (int k = 1; k <= kb; k++) {
        BDI(tbnb, a, k) = BDI(tbnf, a, k); BDI(sbnb, a, k) = BDI(sbnf, a, k);
        BDI(tbsb, a, k) = BDI(tbsf, a, k); BDI(sbsb, a, k) = BDI(sbsf, a, k);
      }
      BD1(vabnb, a) = BD1(vabnf, a); BD1(vabsb, a) = BD1(vabsf, a);
    }
  } else {
    if (west_east) {
      double ue = 0., uw = 0.;
      if (a <= P.jm)
        for (int k = 1; k <= kb; k++) {
          const double dzk = F1(dz, k);
          BDJ(tbw, a, k) = fold * BDJ(tbwb, a, k) + fnew * BDJ(tbwf, a, k);
          BDJ(sbw, a, k) = fold * BDJ(sbwb, a, k) + fnew * BDJ(sbwf, a, k);
          const double x = fold * BDJ(ubwb, a, k) + fnew * BDJ(ubwf, a, k);
          BDJ(ubw, a, k) = x;
          BDJ(tbe, a, k) = fold * BDJ(tbeb, a, k) + fnew * BDJ(tbef, a, k);
          BDJ(sbe, a, k) = fold * BDJ(sbeb, a, k) + fnew * BDJ(sbef, a, k);
          const double p = fold * BDJ(ubeb, a, k) + fnew * BDJ(ubef, a, k);
          BDJ(ube, a, k) = p;
          ue = ue + p * dzk; uw = uw + x * dzk;
        }
      BD1(uabe, a) = ue; BD1(uabw, a) = uw;                          // zero beyond jm (:788-789)
    } else {
      double vn = 0., vs = 0.;
      if (a <= P.im)
        for (int k = 1; k <= kb; k++) {
          const double dzk = F1(dz, k);
          BDI(tbn, a, k) = fold * BDI(tbnb, a, k) + fnew * BDI(tbnf, a, k);
          BDI(sbn, a, k) = fold * BDI(sbnb, a, k) + fnew * BDI(sbnf, a, k);
          const double y = fold * BDI(vbnb, a, k) + fnew * BDI(vbnf, a, k);
          BDI(vbn, a, k) = y;
          BDI(tbs, a, k) = fold * BDI(tbsb, a, k) + fnew * BDI(tbsf, a, k);
          BDI(sbs, a, k) = fold * BDI(sbsb, a, k) + fnew * BDI(sbsf, a, k);
          const double q = fold * BDI(vbsb, a, k) + fnew * BDI(vbsf, a, k);
          BDI(vbs, a, k) = q;
          vn = vn + y * dzk; vs = vs + q * dzk;
        }
      BD1(vabn, a) = vn; BD1(vabs, a) = vs;
    }
  }
}

// ---- launchers --------------------------------------------------------------------------------
void launch_lat(pomgpu_ctx *c, int phase, const double *rec, double fold, double fnew) {
  const int n = c->P.jml + c->P.iml;
  LAUNCH(c, k_lateral, dim3((n + 63) / 64, 1, 1), dim3(64, 1, 1), c->P, phase, rec, fold, fnew);
}
void launch_bcond4_edges(pomgpu_ctx *c) {
  const KP &P = c->P;
  if (!(P.W || P.E || P.S || P.N)) return;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_bcond4_edges, dim3((len + 63) / 64, 4, P.kbm1), dim3(64, 1, 1), c->P);
}
void launch_bcond6_edges(pomgpu_ctx *c) {
  const KP &P = c->P;
  if (!(P.W || P.E || P.S || P.N)) return;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_bcond6_edges, dim3((len + 63) / 64, 4, P.kb), dim3(64, 1, 1), c->P);
}
void launch_bcondorl3(pomgpu_ctx *c) {
  const KP &P = c->P;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_bcondorl3, dim3((len + 63) / 64, 6, P.kbm1), dim3(64, 1, 1), c->P);
}

// ---- halo pack / unpack (parallel_mpi.f:154-351 pack loops) ----------------------------------------
// One launch moves the edge columns (dir 0) or rows (dir 1) of up to 8 arrays.  grid.x covers the
// edge length, grid.y the levels of the deepest array, grid.z the arrays.
// st[a]: array a lives in the storage type of the 3-D arrays (blk3d, the 3-D scratch arrays: fp32 in the fp32-storage
// variant), else it is a 2-D array of doubles.  The staging buffers always carry doubles: a stored fp32 value is widened
// on the way out and rounds back to itself on the way in, so one message may mix 2-D and 3-D arrays.
#define HALO_MAXARR 12                                      /* arrays one exchange point may carry (the reference's largest: six, advance.f:516-521; rim round Rq: nine) */
struct HaloArgs { double *ptr[HALO_MAXARR]; int nz[HALO_MAXARR]; size_t off[HALO_MAXARR]; int count; unsigned char st[HALO_MAXARR]; };
#ifdef POMGPU_STORE_F32
#define HGET(p, i, j, k) (A.st[a] ? (double)G3(p, i, j, k) : (p)[IX3(i, j, k)])
#define HPUT(p, i, j, k, v) do { if (A.st[a]) G3(p, i, j, k) = (v); else (p)[IX3(i, j, k)] = (v); } while (0)
#else
#define HGET(p, i, j, k) G3(p, i, j, k)
#define HPUT(p, i, j, k, v) G3(p, i, j, k) = (v)
#endif
__global__ void k_halo_pack(KP P, HaloArgs A, int dir, double *to_lo, double *to_hi) {
  const int t = TID_I, k = (int)blockIdx.y + 1, a = (int)blockIdx.z;
  const int len = dir == 0 ? P.jm : P.im;
  if (t > len || k > A.nz[a]) return;
  const double *p = A.ptr[a];
  const size_t o = A.off[a] * (size_t)len + (size_t)(k - 1) * len + (size_t)(t - 1);
  if (dir == 0) {
    if (to_lo) to_lo[o] = HGET(p, 2, t, k);
    if (to_hi) to_hi[o] = HGET(p, P.imm1, t, k);
  } else {
    if (to_lo) to_lo[o] = HGET(p, t, 2, k);
    if (to_hi) to_hi[o] = HGET(p, t, P.jmm1, k);
  }
}
__global__ void k_halo_unpack(KP P, HaloArgs A, int dir, const double *from_lo, const double *from_hi) {
  const int t = TID_I, k = (int)blockIdx.y + 1, a = (int)blockIdx.z;
  const int len = dir == 0 ? P.jm : P.im;
  if (t > len || k > A.nz[a]) return;
  double *p = A.ptr[a];
  const size_t o = A.off[a] * (size_t)len + (size_t)(k - 1) * len + (size_t)(t - 1);
  if (dir == 0) {
    if (from_lo) HPUT(p, 1, t, k, from_lo[o]);
    if (from_hi) HPUT(p, P.im, t, k, from_hi[o]);
  } else {
    if (from_lo) HPUT(p, t, 1, k, from_lo[o]);
    if (from_hi) HPUT(p, t, P.jm, k, from_hi[o]);
  }
}
// ---- single-phase exchange with up to eight neighbours -------------------------------------------------
// The reference's two phases (east/west, then north/south INCLUDING the fresh ghost columns) put into a
// corner ghost cell what the diagonal neighbour owns, or -- where one of the two adjacent sides is a
// physical edge -- the edge cell of the other adjacent neighbour.  The same final ghost values follow from
// ONE round: columns 2 / imm1 to W / E, rows 2 / jmm1 to S / N, the four cells (2,2), (imm1,2), (2,jmm1),
// (imm1,jmm1) to SW, SE, NW, NE; the receiver writes a column only over the rows that no N/S neighbour's row
// will own (1 or 2 .. jm or jmm1), a row over i = 1 or 2 .. im or imm1, corners from the diagonal tiles.
// Direction order of the buffer tables: W E S N SW SE NW NE.  Layout per buffer: arrays concatenated, each
// nz x len, level-major (len = jm, jm, im, im, 1, 1, 1, 1).
struct Halo8 { double *b[8]; };
__global__ void k_halo_pack8(KP P, HaloArgs A, Halo8 to) {
  const int t = TID_I, k = (int)blockIdx.y + 1, a = (int)blockIdx.z;
  if (k > A.nz[a]) return;
  const double *p = A.ptr[a];
  if (t <= P.jm) {
    const size_t o = A.off[a] * (size_t)P.jm + (size_t)(k - 1) * P.jm + (size_t)(t - 1);
    if (to.b[0]) to.b[0][o] = HGET(p, 2, t, k);
    if (to.b[1]) to.b[1][o] = HGET(p, P.imm1, t, k);
  }
  if (t <= P.im) {
    const size_t o = A.off[a] * (size_t)P.im + (size_t)(k - 1) * P.im + (size_t)(t - 1);
    if (to.b[2]) to.b[2][o] = HGET(p, t, 2, k);
    if (to.b[3]) to.b[3][o] = HGET(p, t, P.jmm1, k);
  }
  if (t == 1) {
    const size_t o = A.off[a] + (size_t)(k - 1);
    if (to.b[4]) to.b[4][o] = HGET(p, 2, 2, k);
    if (to.b[5]) to.b[5][o] = HGET(p, P.imm1, 2, k);
    if (to.b[6]) to.b[6][o] = HGET(p, 2, P.jmm1, k);
    if (to.b[7]) to.b[7][o] = HGET(p, P.imm1, P.jmm1, k);
  }
}
__global__ void k_halo_unpack8(KP P, HaloArgs A, Halo8 from) {
  const int t = TID_I, k = (int)blockIdx.y + 1, a = (int)blockIdx.z;
  if (k > A.nz[a]) return;
  double *p = A.ptr[a];
  const int jlo = P.S ? 1 : 2, jhi = P.N ? P.jm : P.jmm1, ilo = P.W ? 1 : 2, ihi = P.E ? P.im : P.imm1;
  if (t >= jlo && t <= jhi) {
    const size_t o = A.off[a] * (size_t)P.jm + (size_t)(k - 1) * P.jm + (size_t)(t - 1);
    if (from.b[0]) HPUT(p, 1, t, k, from.b[0][o]);
    if (from.b[1]) HPUT(p, P.im, t, k, from.b[1][o]);
  }
  if (t >= ilo && t <= ihi) {
    const size_t o = A.off[a] * (size_t)P.im + (size_t)(k - 1) * P.im + (size_t)(t - 1);
    if (from.b[2]) HPUT(p, t, 1, k, from.b[2][o]);
    if (from.b[3]) HPUT(p, t, P.jm, k, from.b[3][o]);
  }
  if (t == 1) {
    const size_t o = A.off[a] + (size_t)(k - 1);
    if (from.b[4]) HPUT(p, 1, 1, k, from.b[4][o]);
    if (from.b[5]) HPUT(p, P.im, 1, k, from.b[5][o]);
    if (from.b[6]) HPUT(p, 1, P.jm, k, from.b[6][o]);
    if (from.b[7]) HPUT(p, P.im, P.jm, k, from.b[7][o]);
  }
}
static int halo_args(pomgpu_ctx *c, double *const *dev, const int *nz, int count, HaloArgs &A, int &nzmax) {
  if (count < 1 || count > HALO_MAXARR) return -1;
  size_t off = 0;
  nzmax = 0;
  A.count = count;
  for (int n = 0; n < HALO_MAXARR; n++) { A.ptr[n] = NULL; A.nz[n] = 0; A.off[n] = 0; A.st[n] = 0; }
  for (int n = 0; n < count; n++) {
    if (!dev[n] || nz[n] < 1 || nz[n] > c->P.kb) return -1;
    A.ptr[n] = dev[n]; A.nz[n] = nz[n]; A.off[n] = off;
    {                                                         // inside blk3d or a 3-D scratch array?
      const KP &P = c->P;
      bool st = !(c->flags & POMGPU_CTX_2D) && P.b3 && dev[n] >= P.b3 && dev[n] < P.b3 + (size_t)POM_NBLK3D * P.a3;
      for (int q = 0; q < POMGPU_NSCR3 && !st; q++) st = P.s3[q] && dev[n] >= P.s3[q] && dev[n] < P.s3[q] + P.n3;
      A.st[n] = st ? 1 : 0;
    }
    off += (size_t)nz[n];
    if (nz[n] > nzmax) nzmax = nz[n];
  }
  return 0;
}
int launch_halo_pack(pomgpu_ctx *c, double *const *dev, const int *nz, int count, int dir, double *to_lo, double *to_hi) {
  HaloArgs A; int nzmax;
  if (halo_args(c, dev, nz, count, A, nzmax)) return -1;
  const int len = dir == 0 ? c->P.jm : c->P.im;
  LAUNCH(c, k_halo_pack, dim3((len + 63) / 64, nzmax, count), dim3(64, 1, 1), c->P, A, dir, to_lo, to_hi);
  return 0;
}
int launch_halo_unpack(pomgpu_ctx *c, double *const *dev, const int *nz, int count, int dir, const double *from_lo, const double *from_hi) {
  HaloArgs A; int nzmax;
  if (halo_args(c, dev, nz, count, A, nzmax)) return -1;
  const int len = dir == 0 ? c->P.jm : c->P.im;
  LAUNCH(c, k_halo_unpack, dim3((len + 63) / 64, nzmax, count), dim3(64, 1, 1), c->P, A, dir, from_lo, from_hi);
  return 0;
}
int launch_halo_pack8(pomgpu_ctx *c, double *const *dev, const int *nz, int count, double *const *to) {
  HaloArgs A; int nzmax;
  if (halo_args(c, dev, nz, count, A, nzmax)) return -1;
  Halo8 H;
  for (int d = 0; d < 8; d++) H.b[d] = to[d];
  const int len = c->P.im > c->P.jm ? c->P.im : c->P.jm;
  LAUNCH(c, k_halo_pack8, dim3((len + 63) / 64, nzmax, count), dim3(64, 1, 1), c->P, A, H);
  return 0;
}
int launch_halo_unpack8(pomgpu_ctx *c, double *const *dev, const int *nz, int count, const double *const *from) {
  HaloArgs A; int nzmax;
  if (halo_args(c, dev, nz, count, A, nzmax)) return -1;
  Halo8 H;
  for (int d = 0; d < 8; d++) H.b[d] = const_cast<double *>(from[d]);
  const int len = c->P.im > c->P.jm ? c->P.im : c->P.jm;
  LAUNCH(c, k_halo_unpack8, dim3((len + 63) / 64, nzmax, count), dim3(64, 1, 1), c->P, A, H);
  return 0;
}

// ---- rectangular block copies (wide-halo external mode: gather into / scatter from the extended tile) ------
// One launch serves a whole table of jobs (blockIdx.z = job); a job narrower / shorter than the launch's
// extent leaves its surplus threads idle.  i runs along threadIdx.x: rows are read and written contiguously.
__global__ void k_rect_copy(const RectJob *jobs, int njobs) {
  const int q = (int)blockIdx.z;
  if (q >= njobs) return;
  const RectJob J = jobs[q];
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x), j = (int)(blockIdx.y * blockDim.y + threadIdx.y);
  if (i >= J.ni || j >= J.nj) return;
  J.dst[(size_t)j * J.ld_d + i] = J.src[(size_t)j * J.ld_s + i];
}
void launch_rect_jobs(pomgpu_ctx *c, const RectJob *jobs_dev, int njobs, int max_ni, int max_nj) {
  if (njobs <= 0 || max_ni <= 0 || max_nj <= 0) return;
  LAUNCH(c, k_rect_copy, dim3((max_ni + 63) / 64, (max_nj + 3) / 4, njobs), dim3(64, 4, 1), jobs_dev, njobs);
}
