// k_adv.hip -- horizontal stencil kernels of the internal (3-D baroclinic) mode:
// advct, the Smagorinsky aam loop, advq, advt1, advt2 + smol_adif, dens, restore_interior,
// realvertvl and the Asselin-filter / time-rotation passes.
//
// Thread map: one thread per cell, threadIdx.x along i (the contiguous dimension of every array),
// blockIdx.z = sigma level.  Each wavefront therefore reads and writes whole 512-byte row
// segments; i+-1 neighbours come from the same cache lines, j+-1 neighbours from the rows the
// adjacent wavefronts of the block have just touched (L1/L2 resident).  All kernels are
// HBM-bound: a few flops per 8-byte word, divisions being the only expensive operation.
#include "pomgpu_internal.hpp"

#define dt_(i, j) F2(dt, i, j)
#define dx_(i, j) F2(dx, i, j)
#define dy_(i, j) F2(dy, i, j)
#define h_(i, j) F2(h, i, j)
#define u_(i, j, k) F3(u, i, j, k)
#define v_(i, j, k) F3(v, i, j, k)
#define ub_(i, j, k) F3(ub, i, j, k)
#define vb_(i, j, k) F3(vb, i, j, k)
#define w_(i, j, k) F3(w, i, j, k)
#define aam_(i, j, k) F3(aam, i, j, k)

// Launch geometry of the cell kernels (1-D grid, decoded here; see gridm() for the encoding).
// The plane is cut into BANDS of a few rows; bands are dealt round-robin to the 8 XCDs and every XCD
// walks ALL levels of one band before it takes its next band.  A band is sized so that its slice of
// every 2-D coefficient array plus the current levels of the 3-D operands fit the XCD's 4 MiB L2:
// the coefficients are then fetched from HBM once per band instead of once per level, and the j+-1
// rows of a stencil are L2 hits.  (A plain k-on-blockIdx.z launch re-streams every 2-D array and the
// k+-1 planes once per level -- measured 2-4x the algorithmic traffic at 2048x1536; marching k inside
// the thread serialises the loads of a column -- measured slower.)  Workgroups are observed to be
// dispatched round-robin over the XCDs in linear-id order (MI355X_MICROARCH.md); the mapping only
// relies on that for speed, never for correctness.
#define MARCH3(call)                                                                  \
  const int g_ = (int)(blockIdx.x * blockDim.x + threadIdx.x);   /* x-thread index; 64 per workgroup */ \
  const int L_ = g_ >> 6, xcd_ = L_ & 7, m_ = L_ >> 3;                                \
  const int bpl_ = P.g_bpl, nbx_ = P.g_nbx;                                           \
  const int p_ = m_ % bpl_, t_ = m_ / bpl_;                                           \
  const int k = t_ % P.kb + 1;                                                        \
  const int band_ = (t_ / P.kb) * 8 + xcd_;                                           \
  const int i = (p_ % nbx_) * 64 + (g_ & 63) + 1;                                     \
  const int j = band_ * P.g_rb + (p_ / nbx_) * 4 + (int)threadIdx.y + 1;              \
  if (i > P.iml || j > P.jml) return;                                                 \
  call;
// ---------------------------------------------------------------------------------------------
// advct phase a: curv, and the fluxes of the x-momentum equation -- solver.f:213-277
// scratch: s3[0]=curv  s3[1]=xflux  s3[2]=yflux
__device__ __forceinline__ void c_advct_a(const KP &P, const int i, const int j, const int k);
__global__ void k_advct_a(KP P) {
  MARCH3(c_advct_a(P, i, j, k))
}
__device__ __forceinline__ void c_advct_a(const KP &P, const int i, const int j, const int k) {
  if (i > P.im || j > P.jm) return;
  double cv = 0., xf = 0., yf = 0.;
  if (k <= P.kbm1) {
    const bool iin = (i >= 2 && i <= P.imm1);
    if (iin && j >= 2 && j <= P.jmm1)
      cv = .25 * ((v_(i, j + 1, k) + v_(i, j, k)) * K2(CVA, i, j) - (u_(i + 1, j, k) + u_(i, j, k)) * K2(CVB, i, j)) /
           F2(art, i, j);                                          // art = dx*dy
    if (iin)                                                       // DTSX(i+1,j) = dt(i+1,j)+dt(i,j)
      xf = .125 * (K2(DTSX, i + 1, j) * u_(i + 1, j, k) + K2(DTSX, i, j) * u_(i, j, k)) * (u_(i + 1, j, k) + u_(i, j, k));
    if (i >= 2 && j >= 2)
      yf = .125 * (K2(DTSY, i, j) * v_(i, j, k) + K2(DTSY, i - 1, j) * v_(i - 1, j, k)) * (u_(i, j, k) + u_(i, j - 1, k));
    if (iin && j >= 2) {
      xf = xf - dt_(i, j) * aam_(i, j, k) * 2. * (ub_(i + 1, j, k) - ub_(i, j, k)) / dx_(i, j);
      const double dtaam = .25 * K2(DT4, i, j) * (aam_(i, j, k) + aam_(i - 1, j, k) + aam_(i, j - 1, k) + aam_(i - 1, j - 1, k));
      const double dx4 = K2(DX4, i, j);
      yf = yf - dtaam * ((ub_(i, j, k) - ub_(i, j - 1, k)) / K2(DY4, i, j) + (vb_(i, j, k) - vb_(i - 1, j, k)) / dx4);
      xf = dy_(i, j) * xf;
      yf = .25 * dx4 * yf;
    }
  }
  G3(P.s3[0], i, j, k) = cv;
  G3(P.s3[1], i, j, k) = xf;
  G3(P.s3[2], i, j, k) = yf;
}

// advct phase b: advx, and the fluxes of the y-momentum equation -- solver.f:282-367
// scratch in: s3[0..2]; out: s3[3]=xflux' s3[4]=yflux'
__device__ __forceinline__ void c_advct_b(const KP &P, const int i, const int j, const int k);
__global__ void k_advct_b(KP P) {
  MARCH3(c_advct_b(P, i, j, k))
}
__device__ __forceinline__ void c_advct_b(const KP &P, const int i, const int j, const int k) {
  const double *cv = P.s3[0], *xf = P.s3[1], *yf = P.s3[2];
  double ax = 0., xg = 0., yg = 0.;
  if (k <= P.kbm1 && i <= P.im && j <= P.jm) {
    if (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1) {
      ax = G3(xf, i, j, k) - G3(xf, i - 1, j, k) + G3(yf, i, j + 1, k) - G3(yf, i, j, k);
      if (i >= (P.W ? 3 : 2))
        ax = ax - F2(aru, i, j) * .25 *
                      (G3(cv, i, j, k) * dt_(i, j) * (v_(i, j + 1, k) + v_(i, j, k)) +
                       G3(cv, i - 1, j, k) * dt_(i - 1, j) * (v_(i - 1, j + 1, k) + v_(i - 1, j, k)));
    }
    if (i >= 2 && j >= 2)
      xg = .125 * (K2(DTSX, i, j) * u_(i, j, k) + K2(DTSX, i, j - 1) * u_(i, j - 1, k)) * (v_(i, j, k) + v_(i - 1, j, k));
    if (j >= 2 && j <= P.jmm1) {                                   // DTSY(i,j+1) = dt(i,j+1)+dt(i,j)
      yg = .125 * (K2(DTSY, i, j + 1) * v_(i, j + 1, k) + K2(DTSY, i, j) * v_(i, j, k)) * (v_(i, j + 1, k) + v_(i, j, k));
      if (i >= 2) {
        const double dtaam = .25 * K2(DT4, i, j) * (aam_(i, j, k) + aam_(i - 1, j, k) + aam_(i, j - 1, k) + aam_(i - 1, j - 1, k));
        const double dy4 = K2(DY4, i, j);
        xg = xg - dtaam * ((ub_(i, j, k) - ub_(i, j - 1, k)) / dy4 + (vb_(i, j, k) - vb_(i - 1, j, k)) / K2(DX4, i, j));
        yg = yg - dt_(i, j) * aam_(i, j, k) * 2. * (vb_(i, j + 1, k) - vb_(i, j, k)) / dy_(i, j);
        xg = .25 * dy4 * xg;
        yg = dx_(i, j) * yg;
      }
    }
  }
  F3(advx, i, j, k) = ax;
  if (i <= P.im && j <= P.jm) {
    G3(P.s3[3], i, j, k) = xg;
    G3(P.s3[4], i, j, k) = yg;
  }
}

// advct phase c: advy -- solver.f:372-403
__device__ __forceinline__ void c_advct_c(const KP &P, const int i, const int j, const int k);
__global__ void k_advct_c(KP P) {
  MARCH3(c_advct_c(P, i, j, k))
}
__device__ __forceinline__ void c_advct_c(const KP &P, const int i, const int j, const int k) {
  const double *cv = P.s3[0], *xg = P.s3[3], *yg = P.s3[4];
  double ay = 0.;
  if (k <= P.kbm1 && i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1) {
    ay = G3(xg, i + 1, j, k) - G3(xg, i, j, k) + G3(yg, i, j, k) - G3(yg, i, j - 1, k);
    if (j >= (P.S ? 3 : 2))
      ay = ay + F2(arv, i, j) * .25 *
                    (G3(cv, i, j, k) * dt_(i, j) * (u_(i + 1, j, k) + u_(i, j, k)) +
                     G3(cv, i, j - 1, k) * dt_(i, j - 1) * (u_(i + 1, j - 1, k) + u_(i, j - 1, k)));
  }
  F3(advy, i, j, k) = ay;
}

// Smagorinsky lateral viscosity -- advance.f:122-136
__device__ __forceinline__ void c_aam(const KP &P, const int i, const int j, const int k);
__global__ void k_aam(KP P) {
  MARCH3(c_aam(P, i, j, k))
}
__device__ __forceinline__ void c_aam(const KP &P, const int i, const int j, const int k) {
  if (k > P.kbm1 || i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) return;
  F3(aam, i, j, k) =
      P.horcon * dx_(i, j) * dy_(i, j) *
      sqrt(sq((u_(i + 1, j, k) - u_(i, j, k)) / dx_(i, j)) + sq((v_(i, j + 1, k) - v_(i, j, k)) / dy_(i, j)) +
           .5 * sq(.25 * (u_(i, j + 1, k) + u_(i + 1, j + 1, k) - u_(i, j - 1, k) - u_(i + 1, j - 1, k)) / dy_(i, j) +
                   .25 * (v_(i + 1, j, k) + v_(i + 1, j + 1, k) - v_(i - 1, j, k) - v_(i - 1, j + 1, k)) / dx_(i, j)));
}

// rho = (rho-rmean)+rmean: the in-place round trip of baropg (solver.f:854,937) leaves rho changed
// by a rounding, which later reads of rho (profq) see
__device__ __forceinline__ void c_roundtrip(const KP &P, const int i, const int j, const int k, double *a, const double *b, int fix_kb);
__global__ void k_roundtrip(KP P, double *a, const double *b, int fix_kb) {
  MARCH3(c_roundtrip(P, i, j, k, a, b, fix_kb))
}
// one level only (the deferred round trip of rho, pomgpu_api.hip: its level kb, which dens never rewrites)
__global__ void k_roundtrip_level(KP P, double *a, const double *b, int k) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  double x = G3(a, i, j, k);
  const double y = G3(b, i, j, k);
  x = x - y;
  G3(a, i, j, k) = x + y;
}
__device__ __forceinline__ void c_roundtrip(const KP &P, const int i, const int j, const int k, double *a, const double *b, int fix_kb) {
  if (fix_kb == 2 && k == P.kb) return;                                // fix_kb = 2: levels 1..kbm1 only
  double x = G3(a, i, j, (fix_kb == 1 && k == P.kb) ? P.kbm1 : k);
  const double y = G3(b, i, j, k);
  x = x - y;
  G3(a, i, j, k) = x + y;
}

// ---------------------------------------------------------------------------------------------
// advq -- solver.f:411-477.  Phase 1: fluxes (exchanged), phase 2: step.
__device__ __forceinline__ void c_advq_flux(const KP &P, const int i, const int j, const int k, const double *q, const double *qb, double *xf, double *yf);
__global__ void k_advq_flux(KP P, const double *q, const double *qb, double *xf, double *yf) {
  MARCH3(c_advq_flux(P, i, j, k, q, qb, xf, yf))
}
__device__ __forceinline__ void c_advq_flux(const KP &P, const int i, const int j, const int k, const double *q, const double *qb, double *xf, double *yf) {
  if (i > P.im || j > P.jm) return;
  double x = 0., y = 0.;
  if (k >= 2 && k <= P.kbm1 && i >= 2 && j >= 2) {
    x = .125 * (G3(q, i, j, k) + G3(q, i - 1, j, k)) * (dt_(i, j) + dt_(i - 1, j)) * (u_(i, j, k) + u_(i, j, k - 1));
    y = .125 * (G3(q, i, j, k) + G3(q, i, j - 1, k)) * (dt_(i, j) + dt_(i, j - 1)) * (v_(i, j, k) + v_(i, j, k - 1));
    x = x - .25 * (aam_(i, j, k) + aam_(i - 1, j, k) + aam_(i, j, k - 1) + aam_(i - 1, j, k - 1)) * (h_(i, j) + h_(i - 1, j)) *
                (G3(qb, i, j, k) - G3(qb, i - 1, j, k)) * F2(dum, i, j) / (dx_(i, j) + dx_(i - 1, j));
    y = y - .25 * (aam_(i, j, k) + aam_(i, j - 1, k) + aam_(i, j, k - 1) + aam_(i, j - 1, k - 1)) * (h_(i, j) + h_(i, j - 1)) *
                (G3(qb, i, j, k) - G3(qb, i, j - 1, k)) * F2(dvm, i, j) / (dy_(i, j) + dy_(i, j - 1));
    x = .5 * (dy_(i, j) + dy_(i - 1, j)) * x;
    y = .5 * (dx_(i, j) + dx_(i, j - 1)) * y;
  }
  G3(xf, i, j, k) = x;
  G3(yf, i, j, k) = y;
}
// zero_else: the caller zero-filled qf beforehand in the reference (advance.f:403-404)
__device__ __forceinline__ void c_advq_step(const KP &P, const int i, const int j, const int k, const double *q, const double *qb, double *qf, const double *xf, const double *yf,
                            int zero_else);
__global__ void k_advq_step(KP P, const double *q, const double *qb, double *qf, const double *xf, const double *yf,
                            int zero_else) {
  MARCH3(c_advq_step(P, i, j, k, q, qb, qf, xf, yf, zero_else))
}
__device__ __forceinline__ void c_advq_step(const KP &P, const int i, const int j, const int k, const double *q, const double *qb, double *qf, const double *xf, const double *yf,
                            int zero_else) {
  if (k >= 2 && k <= P.kbm1 && i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1) {
    double r = (w_(i, j, k - 1) * G3(q, i, j, k - 1) - w_(i, j, k + 1) * G3(q, i, j, k + 1)) * F2(art, i, j) /
                   (F1(dz, k) + F1(dz, k - 1)) +
               G3(xf, i + 1, j, k) - G3(xf, i, j, k) + G3(yf, i, j + 1, k) - G3(yf, i, j, k);
    r = ((h_(i, j) + F2(etb, i, j)) * F2(art, i, j) * G3(qb, i, j, k) - P.dti2 * r) / ((h_(i, j) + F2(etf, i, j)) * F2(art, i, j));
    G3(qf, i, j, k) = r;
  } else if (zero_else) {
    G3(qf, i, j, k) = 0.;
  }
}

// bcond(6) mask + Asselin filter + rotation of q2/q2l -- bounds_forcing.f:315-322, advance.f:416-421
__device__ __forceinline__ void c_q_filter(const KP &P, const int i, const int j, const int k, int mask);
__global__ void k_q_filter(KP P, int mask) {
  MARCH3(c_q_filter(P, i, j, k, mask))
}
__device__ __forceinline__ void c_q_filter(const KP &P, const int i, const int j, const int k, int mask) {
  double uf = F3(uf, i, j, k), vf = F3(vf, i, j, k);
  if (mask && i <= P.im && j <= P.jm) {
    const double m = F2(fsm, i, j);
    uf = uf * m + 1.e-10;                                  // bcond(6)
    vf = vf * m + 1.e-10;
    // stored only where somebody still reads them: advt1/advt2 write the interior of levels 1..kbm1
    // and leave the rim columns and level kb of uf, vf as they are (solver.f:577-731) -- those
    // left-overs flow on into t, s.  Everywhere else the next writer replaces the value unread.
    if (k == P.kb || i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) {
      F3(uf, i, j, k) = uf;
      F3(vf, i, j, k) = vf;
    }
  }
  const double q = F3(q2, i, j, k), ql = F3(q2l, i, j, k);
  F3(q2b, i, j, k) = q + .5 * P.smoth * (uf + F3(q2b, i, j, k) - 2. * q);
  F3(q2, i, j, k) = uf;
  F3(q2lb, i, j, k) = ql + .5 * P.smoth * (vf + F3(q2lb, i, j, k) - 2. * ql);
  F3(q2l, i, j, k) = vf;
}
__global__ void k_q_filter_rim(KP P) {
  const int t = TID_I, line = (int)blockIdx.y, k = TID_K;
  if (k > P.kb) return;
  int i, j;
  if (line < 4) {                                          // columns 1, 2, imm1, im over all rows
    if (t > P.jm) return;
    i = line == 0 ? 1 : (line == 1 ? 2 : (line == 2 ? P.imm1 : P.im)); j = t;
    if (line == 2 && P.imm1 <= 2) return;                  // (tiny tiles: no line twice)
  } else {                                                 // rows 1, 2, jmm1, jm between them
    if (t < 3 || t > P.imm1 - 1) return;
    j = line == 4 ? 1 : (line == 5 ? 2 : (line == 6 ? P.jmm1 : P.jm)); i = t;
    if (line == 6 && P.jmm1 <= 2) return;
  }
  c_q_filter(P, i, j, k, 1);
}
__device__ __forceinline__ void c_mask_q(const KP &P, const int i, const int j, const int k);
__global__ void k_mask_q(KP P) {   // the mask of bcond(6) alone
  MARCH3(c_mask_q(P, i, j, k))
}
__device__ __forceinline__ void c_mask_q(const KP &P, const int i, const int j, const int k) {
  if (i > P.im || j > P.jm) return;
  const double m = F2(fsm, i, j);
  F3(uf, i, j, k) = F3(uf, i, j, k) * m + 1.e-10;
  F3(vf, i, j, k) = F3(vf, i, j, k) * m + 1.e-10;
}

// ---------------------------------------------------------------------------------------------
// advt1 -- solver.f:480-574.  Fluxes are recomputed per cell face (no intermediate arrays).
__device__ __forceinline__ double advt1_xflux(const KP &P, const double *f, const double *fb, const double *fc, int i, int j, int k) {
  double x = .25 * ((dt_(i, j) + dt_(i - 1, j)) * (G3(f, i, j, k) + G3(f, i - 1, j, k)) * u_(i, j, k));
  x = x - .5 * (aam_(i, j, k) + aam_(i - 1, j, k)) * (h_(i, j) + h_(i - 1, j)) * P.tprni *
              ((G3(fb, i, j, k) - G3(fc, i, j, k)) - (G3(fb, i - 1, j, k) - G3(fc, i - 1, j, k))) * F2(dum, i, j) /
              (dx_(i, j) + dx_(i - 1, j));
  return .5 * (dy_(i, j) + dy_(i - 1, j)) * x;
}
__device__ __forceinline__ double advt1_yflux(const KP &P, const double *f, const double *fb, const double *fc, int i, int j, int k) {
  double y = .25 * ((dt_(i, j) + dt_(i, j - 1)) * (G3(f, i, j, k) + G3(f, i, j - 1, k)) * v_(i, j, k));
  y = y - .5 * (aam_(i, j, k) + aam_(i, j - 1, k)) * (h_(i, j) + h_(i, j - 1)) * P.tprni *
              ((G3(fb, i, j, k) - G3(fc, i, j, k)) - (G3(fb, i, j - 1, k) - G3(fc, i, j - 1, k))) * F2(dvm, i, j) /
              (dy_(i, j) + dy_(i, j - 1));
  return .5 * (dx_(i, j) + dx_(i, j - 1)) * y;
}
__device__ __forceinline__ void c_advt1(const KP &P, const int i, const int j, const int k, const double *fb, const double *f, const double *fc, double *ff);
__global__ void k_advt1(KP P, const double *fb, const double *f, const double *fc, double *ff) {
  MARCH3(c_advt1(P, i, j, k, fb, f, fc, ff))
}
__device__ __forceinline__ void c_advt1(const KP &P, const int i, const int j, const int k, const double *fb, const double *f, const double *fc, double *ff) {
  if (k > P.kbm1 || i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) return;
  const double art = F2(art, i, j);
  const double zu = (k == 1) ? G3(f, i, j, 1) * w_(i, j, 1) * art
                             : .5 * (G3(f, i, j, k - 1) + G3(f, i, j, k)) * w_(i, j, k) * art;
  const double zl = (k == P.kbm1) ? 0. : .5 * (G3(f, i, j, k) + G3(f, i, j, k + 1)) * w_(i, j, k + 1) * art;
  double r = advt1_xflux(P, f, fb, fc, i + 1, j, k) - advt1_xflux(P, f, fb, fc, i, j, k) +
             advt1_yflux(P, f, fb, fc, i, j + 1, k) - advt1_yflux(P, f, fb, fc, i, j, k) + (zu - zl) / F1(dz, k);
  const double fbr = (G3(fb, i, j, k) - G3(fc, i, j, k)) + G3(fc, i, j, k);   // fb after :511,:532
  G3(ff, i, j, k) = (fbr * (h_(i, j) + F2(etb, i, j)) * art - P.dti2 * r) / ((h_(i, j) + F2(etf, i, j)) * art);
}
// f(:,:,kb) = f(:,:,kbm1)  (solver.f:495)
__global__ void k_copy_kb(KP P, double *f, double *g) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  G3(f, i, j, P.kb) = G3(f, i, j, P.kbm1);
  if (g) G3(g, i, j, P.kb) = G3(g, i, j, P.kbm1);
}

// ---------------------------------------------------------------------------------------------
// advt2 + smol_adif -- solver.f:577-731, :1880-1967  (general nitera path)
// scratch: s3[0]=xmassflux s3[1]=ymassflux s3[2]=zwflux s3[3]=fbmem(itera>=2)
__device__ __forceinline__ void c_advt2_mass(const KP &P, const int i, const int j, const int k);
__global__ void k_advt2_mass(KP P) {
  MARCH3(c_advt2_mass(P, i, j, k))
}
__device__ __forceinline__ void c_advt2_mass(const KP &P, const int i, const int j, const int k) {
  if (i > P.im || j > P.jm) return;
  double xm = 0., ym = 0.;
  if (k <= P.kbm1) {
    if (j >= 2 && j <= P.jmm1 && i >= 2) xm = 0.25 * (dy_(i - 1, j) + dy_(i, j)) * (dt_(i - 1, j) + dt_(i, j)) * u_(i, j, k);
    if (j >= 2 && i >= 2 && i <= P.imm1) ym = 0.25 * (dx_(i, j - 1) + dx_(i, j)) * (dt_(i, j - 1) + dt_(i, j)) * v_(i, j, k);
  }
  G3(P.s3[0], i, j, k) = xm;
  G3(P.s3[1], i, j, k) = ym;
  G3(P.s3[2], i, j, k) = w_(i, j, k);
}
__device__ __forceinline__ double upw(double m, double lo, double hi) {   // solver.f:631-635
  return 0.5 * ((m + fabs(m)) * lo + (m - fabs(m)) * hi);
}
__device__ __forceinline__ void c_advt2_step(const KP &P, const int i, const int j, const int k, const double *fbmem, const double *f, const double *eta, double *ff, int itera);
__global__ void k_advt2_step(KP P, const double *fbmem, const double *f, const double *eta, double *ff, int itera) {
  MARCH3(c_advt2_step(P, i, j, k, fbmem, f, eta, ff, itera))
}
__device__ __forceinline__ void c_advt2_step(const KP &P, const int i, const int j, const int k, const double *fbmem, const double *f, const double *eta, double *ff, int itera) {
  if (k > P.kbm1 || i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) return;
  const double *xm = P.s3[0], *ym = P.s3[1], *zw = P.s3[2];
  const double art = F2(art, i, j);
  const double fc = G3(fbmem, i, j, k);
  const double xe = upw(G3(xm, i + 1, j, k), fc, G3(fbmem, i + 1, j, k));
  const double xw = upw(G3(xm, i, j, k), G3(fbmem, i - 1, j, k), fc);
  const double yn = upw(G3(ym, i, j + 1, k), fc, G3(fbmem, i, j + 1, k));
  const double ys = upw(G3(ym, i, j, k), G3(fbmem, i, j - 1, k), fc);
  double zu, zl;
  if (k == 1) zu = (itera == 1) ? w_(i, j, 1) * G3(f, i, j, 1) * art : 0.;
  else zu = upw(G3(zw, i, j, k), fc, G3(fbmem, i, j, k - 1)) * art;
  if (k == P.kbm1) zl = 0.;
  else zl = upw(G3(zw, i, j, k + 1), G3(fbmem, i, j, k + 1), fc) * art;
  double r = xe - xw + yn - ys + (zu - zl) / F1(dz, k);
  G3(ff, i, j, k) = (fc * ((h_(i, j) + G2(eta, i, j)) * art) - P.dti2 * r) / ((h_(i, j) + F2(etf, i, j)) * art);
}
__device__ __forceinline__ void c_mask3(const KP &P, const int i, const int j, const int k, double *a, const double *m2);
__global__ void k_mask3(KP P, double *a, const double *m2) {   // a(:,:,k) = a(:,:,k)*m2 for k=1..kb
  MARCH3(c_mask3(P, i, j, k, a, m2))
}
__device__ __forceinline__ void c_mask3(const KP &P, const int i, const int j, const int k, double *a, const double *m2) {
  G3(a, i, j, k) = G3(a, i, j, k) * G2(m2, i, j);
}
__device__ __forceinline__ void c_smol(const KP &P, const int i, const int j, const int k, const double *ff);
__global__ void k_smol(KP P, const double *ff) {
  MARCH3(c_smol(P, i, j, k, ff))
}
__device__ __forceinline__ void c_smol(const KP &P, const int i, const int j, const int k, const double *ff) {
  if (i > P.im || j > P.jm) return;
  const double value_min = 1.e-9, epsilon = 1.0e-14;
  double *xm = P.s3[0], *ym = P.s3[1], *zw = P.s3[2];
  const double fc = G3(ff, i, j, k);
  if (k <= P.kbm1 && j >= 2 && j <= P.jmm1 && i >= 2) {
    const double fw = G3(ff, i - 1, j, k);
    double r = 0.;
    if (!(fc < value_min || fw < value_min)) {
      const double m = G3(xm, i, j, k);
      const double udx = fabs(m);
      const double u2dt = P.dti2 * m * m * 2. / (F2(aru, i, j) * (dt_(i - 1, j) + dt_(i, j)));
      const double mol = (fc - fw) / (fw + fc + epsilon);
      r = (udx - u2dt) * mol * P.sw;
      if (fabs(udx) < fabs(u2dt)) r = 0.;
    }
    G3(xm, i, j, k) = r;
  }
  if (k <= P.kbm1 && j >= 2 && i >= 2 && i <= P.imm1) {
    const double fs = G3(ff, i, j - 1, k);
    double r = 0.;
    if (!(fc < value_min || fs < value_min)) {
      const double m = G3(ym, i, j, k);
      const double vdy = fabs(m);
      const double v2dt = P.dti2 * m * m * 2. / (F2(arv, i, j) * (dt_(i, j - 1) + dt_(i, j)));
      const double mol = (fc - fs) / (fs + fc + epsilon);
      r = (vdy - v2dt) * mol * P.sw;
      if (fabs(vdy) < fabs(v2dt)) r = 0.;
    }
    G3(ym, i, j, k) = r;
  }
  if (k >= 2 && k <= P.kbm1 && i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1) {
    const double fu = G3(ff, i, j, k - 1);
    double r = 0.;
    if (!(fc < value_min || fu < value_min)) {
      const double m = G3(zw, i, j, k);
      const double wdz = fabs(m);
      const double w2dt = P.dti2 * m * m / (F1(dzz, k - 1) * dt_(i, j));
      const double mol = (fu - fc) / (fc + fu + epsilon);
      r = (wdz - w2dt) * mol * P.sw;
      if (fabs(wdz) < fabs(w2dt)) r = 0.;
    }
    G3(zw, i, j, k) = r;
  }
}
__device__ __forceinline__ void c_copy3(const KP &P, const int i, const int j, const int k, double *dst, const double *src);
__global__ void k_copy3(KP P, double *dst, const double *src) {
  MARCH3(c_copy3(P, i, j, k, dst, src))
}
__device__ __forceinline__ void c_copy3(const KP &P, const int i, const int j, const int k, double *dst, const double *src) {
  G3(dst, i, j, k) = G3(src, i, j, k);
}
__device__ __forceinline__ double advt2_xdiff(const KP &P, const double *fb, const double *fc, int i, int j, int k) {
  const double am = 0.5 * (aam_(i, j, k) + aam_(i - 1, j, k));
  return -am * (h_(i, j) + h_(i - 1, j)) * P.tprni *
         ((G3(fb, i, j, k) - G3(fc, i, j, k)) - (G3(fb, i - 1, j, k) - G3(fc, i - 1, j, k))) * F2(dum, i, j) *
         (dy_(i, j) + dy_(i - 1, j)) * 0.5 / (dx_(i, j) + dx_(i - 1, j));
}
__device__ __forceinline__ double advt2_ydiff(const KP &P, const double *fb, const double *fc, int i, int j, int k) {
  const double am = 0.5 * (aam_(i, j, k) + aam_(i, j - 1, k));
  return -am * (h_(i, j) + h_(i, j - 1)) * P.tprni *
         ((G3(fb, i, j, k) - G3(fc, i, j, k)) - (G3(fb, i, j - 1, k) - G3(fc, i, j - 1, k))) * F2(dvm, i, j) *
         (dx_(i, j) + dx_(i, j - 1)) * 0.5 / (dy_(i, j) + dy_(i, j - 1));
}
__device__ __forceinline__ void c_advt2_diff(const KP &P, const int i, const int j, const int k, const double *fb, const double *fc, double *ff);
__global__ void k_advt2_diff(KP P, const double *fb, const double *fc, double *ff) {
  MARCH3(c_advt2_diff(P, i, j, k, fb, fc, ff))
}
__device__ __forceinline__ void c_advt2_diff(const KP &P, const int i, const int j, const int k, const double *fb, const double *fc, double *ff) {
  if (k > P.kbm1 || i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) return;
  G3(ff, i, j, k) = G3(ff, i, j, k) -
                    P.dti2 * (advt2_xdiff(P, fb, fc, i + 1, j, k) - advt2_xdiff(P, fb, fc, i, j, k) +
                              advt2_ydiff(P, fb, fc, i, j + 1, k) - advt2_ydiff(P, fb, fc, i, j, k)) /
                        ((h_(i, j) + F2(etf, i, j)) * F2(art, i, j));
}
// (nitera == 1 takes k_advt2_col, k_tile.hip)

// ---------------------------------------------------------------------------------------------
// mode_internal, tracer tail in ONE pass (nadv=2 path): the in-place round trips of advt2 on tb/sb
// (solver.f:691,715), bcond(4)'s mask (bounds_forcing.f:233-240), the Asselin filter and rotation of
// t,s (advance.f:444-449), restore_interior's interpolation, relaxation and mask
// (bounds_forcing.f:1086-1120) and dens (solver.f:1162-1209): 14 reads + 5 writes per cell instead
// of the 38 array passes of the five separate kernels.
__device__ __forceinline__ double dens_point(const KP &P, double si, double ti, int i, int j, int k);
// TAU: 1 = taurstrb, taurstrf hold the values tau_b, tau_f everywhere (pomgpu_ctx::tau_known): not read
template <int TAU> __device__ __forceinline__ void c_ts_update(const KP &P, const int i, const int j, const int k, double fold, double fnew, int rt, int store_rst, double tau_b, double tau_f);
template <int TAU> __global__ void k_ts_update(KP P, double fold, double fnew, int rt, int store_rst, double tau_b, double tau_f) {
  MARCH3(c_ts_update<TAU>(P, i, j, k, fold, fnew, rt, store_rst, tau_b, tau_f))
}
template <int TAU> __device__ __forceinline__ void c_ts_update(const KP &P, const int i, const int j, const int k, double fold, double fnew, int rt, int store_rst, double tau_b, double tau_f) {
  const bool act = (i <= P.im && j <= P.jm), lev = (k <= P.kbm1);
  const double m = F2(fsm, i, j);
  double uf = F3(uf, i, j, k), vf = F3(vf, i, j, k);
  if (lev && act) {                                                    // bcond(4) mask
    uf = uf * m;                                                       // not stored: the only reader of the masked uf/vf is this
    vf = vf * m;                                                       // filter; advu/advv rewrite both arrays next (advance.f:459-460)
  }
  double tb = F3(tb, i, j, k), sb = F3(sb, i, j, k);
  if (rt) {                                                            // fb = fb-fclim ... fb = fb+fclim
    const double tc = F3(tclim, i, j, k), sc = F3(sclim, i, j, k);
    tb = (tb - tc) + tc;
    sb = (sb - sc) + sc;
  }
  const double t0 = F3(t, i, j, k), s0 = F3(s, i, j, k);
  double tbn = t0 + .5 * P.smoth * (uf + tb - 2. * t0);                // advance.f:444-449
  double sbn = s0 + .5 * P.smoth * (vf + sb - 2. * s0);
  double tn = uf, sn = vf;
  if (lev) {
    if (act) {                                                         // restore_interior
      const double tr = fold * F3(trstrb, i, j, k) + fnew * F3(trstrf, i, j, k);
      const double sr = fold * F3(srstrb, i, j, k) + fnew * F3(srstrf, i, j, k);
      const double ta = TAU ? fold * tau_b + fnew * tau_f : fold * F3(taurstrb, i, j, k) + fnew * F3(taurstrf, i, j, k);
      if (store_rst) {                                                 // else left to k_restore_fields (on demand)
        F3(trstr, i, j, k) = tr;
        F3(srstr, i, j, k) = sr;
        F3(taurstr, i, j, k) = ta;
      }
      const double c = 2. * P.dti / 86400.;
      tn = tn + c * ta * (tr - tn);
      tbn = tbn + c * ta * (tr - tbn);
      sn = sn + c * ta * (sr - sn);
      sbn = sbn + c * ta * (sr - sbn);
    }
    tn = tn * m; tbn = tbn * m; sn = sn * m; sbn = sbn * m;
  }
  F3(t, i, j, k) = tn;
  F3(tb, i, j, k) = tbn;
  F3(s, i, j, k) = sn;
  F3(sb, i, j, k) = sbn;
  if (lev && act) F3(rho, i, j, k) = dens_point(P, sn, tn, i, j, k);
}
// (Tried: 7 levels per wavefront with the next level's operands in flight, as k_aam_pair now does -- 4.08 against 3.90 ms: this kernel's
// 2.5 M one-level wavefronts at 8 waves per SIMD already keep its 17 passes moving at 5.5 TB/s; the chunked form holds 110 VGPRs, 4 waves.)
__device__ __forceinline__ void c_mask_ts(const KP &P, const int i, const int j, const int k);
__global__ void k_mask_ts(KP P) {   // the mask of bcond(4) alone
  MARCH3(c_mask_ts(P, i, j, k))
}
__device__ __forceinline__ void c_mask_ts(const KP &P, const int i, const int j, const int k) {
  if (k > P.kbm1 || i > P.im || j > P.jm) return;
  const double m = F2(fsm, i, j);
  F3(uf, i, j, k) = F3(uf, i, j, k) * m;
  F3(vf, i, j, k) = F3(vf, i, j, k) * m;
}
__device__ __forceinline__ void c_mask_uv(const KP &P, const int i, const int j, const int k);
__global__ void k_mask_uv(KP P) {   // the mask of bcondorl(3) alone
  MARCH3(c_mask_uv(P, i, j, k))
}
__device__ __forceinline__ void c_mask_uv(const KP &P, const int i, const int j, const int k) {
  if (k > P.kbm1 || i > P.im || j > P.jm) return;
  F3(uf, i, j, k) = F3(uf, i, j, k) * F2(dum, i, j);
  F3(vf, i, j, k) = F3(vf, i, j, k) * F2(dvm, i, j);
}
__device__ __forceinline__ void c_mask_w(const KP &P, const int i, const int j, const int k);
__global__ void k_mask_w(KP P) {    // bcond(5) / bcondorl(5)
  MARCH3(c_mask_w(P, i, j, k))
}
__device__ __forceinline__ void c_mask_w(const KP &P, const int i, const int j, const int k) {
  if (k > P.kbm1 || i > P.im || j > P.jm) return;
  F3(w, i, j, k) = F3(w, i, j, k) * F2(fsm, i, j);
}

// restore_interior, device part -- bounds_forcing.f:1086-1120 (interpolate, relax, mask)
__device__ __forceinline__ void c_restore(const KP &P, const int i, const int j, const int k, double fold, double fnew);
__global__ void k_restore(KP P, double fold, double fnew) {
  MARCH3(c_restore(P, i, j, k, fold, fnew))
}
__device__ __forceinline__ void c_restore(const KP &P, const int i, const int j, const int k, double fold, double fnew) {
  if (k > P.kbm1) return;
  double t = F3(t, i, j, k), tb = F3(tb, i, j, k), s = F3(s, i, j, k), sb = F3(sb, i, j, k);
  if (i <= P.im && j <= P.jm) {
    const double tr = fold * F3(trstrb, i, j, k) + fnew * F3(trstrf, i, j, k);
    const double sr = fold * F3(srstrb, i, j, k) + fnew * F3(srstrf, i, j, k);
    const double ta = fold * F3(taurstrb, i, j, k) + fnew * F3(taurstrf, i, j, k);
    F3(trstr, i, j, k) = tr;
    F3(srstr, i, j, k) = sr;
    F3(taurstr, i, j, k) = ta;
    const double c = 2. * P.dti / 86400.;
    t = t + c * ta * (tr - t);
    tb = tb + c * ta * (tr - tb);
    s = s + c * ta * (sr - s);
    sb = sb + c * ta * (sr - sb);
  }
  const double m = F2(fsm, i, j);
  F3(t, i, j, k) = t * m;
  F3(tb, i, j, k) = tb * m;
  F3(s, i, j, k) = s * m;
  F3(sb, i, j, k) = sb * m;
}
// the interpolated restoring fields alone (bounds_forcing.f:1090-1098).  Inside pomgpu_advance nothing
// reads trstr, srstr, taurstr, so k_ts_update does not write them; this kernel materialises them from
// the last step's weights before anybody can look (downloads, stand-alone entry points).
__device__ __forceinline__ void c_restore_fields(const KP &P, const int i, const int j, const int k, double fold, double fnew);
__global__ void k_restore_fields(KP P, double fold, double fnew) {
  MARCH3(c_restore_fields(P, i, j, k, fold, fnew))
}
__device__ __forceinline__ void c_restore_fields(const KP &P, const int i, const int j, const int k, double fold, double fnew) {
  if (k > P.kbm1 || i > P.im || j > P.jm) return;
  F3(trstr, i, j, k) = fold * F3(trstrb, i, j, k) + fnew * F3(trstrf, i, j, k);
  F3(srstr, i, j, k) = fold * F3(srstrb, i, j, k) + fnew * F3(srstrf, i, j, k);
  F3(taurstr, i, j, k) = fold * F3(taurstrb, i, j, k) + fnew * F3(taurstrf, i, j, k);
}
// trstrb = trstrf etc. for k <= kbm1 (bounds_forcing.f:1056-1064)
__device__ __forceinline__ void c_restore_shift(const KP &P, const int i, const int j, const int k);
__global__ void k_restore_shift(KP P) {
  MARCH3(c_restore_shift(P, i, j, k))
}
__device__ __forceinline__ void c_restore_shift(const KP &P, const int i, const int j, const int k) {
  if (k > P.kbm1 || i > P.im || j > P.jm) return;
  F3(trstrb, i, j, k) = F3(trstrf, i, j, k);
  F3(srstrb, i, j, k) = F3(srstrf, i, j, k);
  F3(taurstrb, i, j, k) = F3(taurstrf, i, j, k);
}
// trstrf(1:im,1:jm,:) = tr ; srstrf = sr ; taurstrf = 1./trst (whole array)
__device__ __forceinline__ void c_restore_load(const KP &P, const int i, const int j, const int k, const double *tr, const double *sr, double tau);
__global__ void k_restore_load(KP P, const double *tr, const double *sr, double tau) {
  MARCH3(c_restore_load(P, i, j, k, tr, sr, tau))
}
__device__ __forceinline__ void c_restore_load(const KP &P, const int i, const int j, const int k, const double *tr, const double *sr, double tau) {
  F3(taurstrf, i, j, k) = tau;
  if (i > P.im || j > P.jm) return;
  const size_t n = ((size_t)(k - 1) * P.jm + (size_t)(j - 1)) * P.im + (size_t)(i - 1);   // (im,jm,kb) record
  F3(trstrf, i, j, k) = tr[n];
  F3(srstrf, i, j, k) = sr[n];
}

// ---------------------------------------------------------------------------------------------
// dens -- solver.f:1162-1209.
// abs(sr)**1.5 is the one libm call on the hot path (the reference calls glibc's pow).  glibc's
// pow is accurate to ~0.52 ulp but NOT correctly rounded, and this flow amplifies a 1-ulp seed
// by 1e11 in 1000 steps, so gpow15() restates glibc 2.35's algorithm (FMA variant, the one x86-64
// hosts with FMA run) instruction for instruction: log via a 128-entry table + degree-7 tail
// polynomial in double-double, multiply by y, exp via a 128-entry 2^(k/128) table.  Verified
// bit-identical to pow() on 2e7 arguments by tools/check_glibc_pow_clone.c.
#include "glibc_pow_tables.h"
static __device__ const double GP_A[7] = GPOW_A;
static __device__ const double GP_LT[128][3] = GPOW_LOGTAB;
static __device__ const double GP_C[4] = GEXP_C;
static __device__ const unsigned long long GP_ET[256] = GEXP_TAB;
__device__ __forceinline__ double pow15_dd(double x) {   // correctly rounded x*sqrt(x); only for non-normal x
  if (x == 0.) return 0.;
  const double s = sqrt(x);
  const double e = __builtin_fma(-s, s, x) / (2. * s);
  const double p = x * s;
  const double pe = __builtin_fma(x, s, -p);
  return p + (pe + x * e);
}
__device__ __forceinline__ double gpow15(double x) {
  const double y = 1.5;
  const unsigned long long ix = __builtin_bit_cast(unsigned long long, x);
  const unsigned top = (unsigned)(ix >> 52);
  if (top - 1u >= 0x7feu) return pow15_dd(x);   // zero, subnormal, inf, nan: glibc's slow paths
  const unsigned long long tmp = ix - 0x3fe6955500000000ULL;
  const int i = (int)((tmp >> 45) & 127);
  const long long k = (long long)tmp >> 52;
  const double z = __builtin_bit_cast(double, ix - (tmp & 0xfff0000000000000ULL));
  const double kd = (double)k;
  const double invc = GP_LT[i][0], logc = GP_LT[i][1], logctail = GP_LT[i][2];
  const double t1 = __builtin_fma(kd, GPOW_LN2HI, logc);
  const double r = __builtin_fma(z, invc, -1.0);
  const double ar = r * GP_A[0];
  const double lo1 = __builtin_fma(kd, GPOW_LN2LO, logctail);
  const double q1 = __builtin_fma(r, GP_A[2], GP_A[1]);
  const double q2 = __builtin_fma(r, GP_A[4], GP_A[3]);
  const double t2 = r + t1;
  const double ar2 = r * ar;
  const double d1 = t1 - t2;
  const double ar3 = r * ar2;
  const double lo3 = __builtin_fma(ar, r, -ar2);
  const double lo2 = d1 + r;
  const double q3 = __builtin_fma(r, GP_A[6], GP_A[5]);
  const double hi = t2 + ar2;
  const double d2 = t2 - hi;
  const double q4 = __builtin_fma(q3, ar2, q2);
  const double lo4 = d2 + ar2;
  const double q5 = __builtin_fma(ar2, q4, q1);
  double sm = lo1 + lo2;
  sm = sm + lo3;
  sm = sm + lo4;
  const double lo = __builtin_fma(ar3, q5, sm);
  const double yl = hi + lo;
  const double tl = (hi - yl) + lo;
  const double ehi = y * yl;
  const double e2 = __builtin_fma(yl, y, -ehi);
  const double elo = __builtin_fma(y, tl, e2);
  double kd2 = __builtin_fma(ehi, GEXP_INVLN2N, GEXP_SHIFT);
  const unsigned long long ki = __builtin_bit_cast(unsigned long long, kd2);
  kd2 = kd2 - GEXP_SHIFT;
  double rr = __builtin_fma(kd2, GEXP_NEGLN2HIN, ehi);
  rr = __builtin_fma(kd2, GEXP_NEGLN2LON, rr);
  const unsigned idx = 2u * (unsigned)(ki & 127);
  const unsigned long long sbits = GP_ET[idx + 1] + (ki << 45);
  rr = elo + rr;
  const double p1 = __builtin_fma(rr, GP_C[1], GP_C[0]);
  const double s1 = rr + __builtin_bit_cast(double, GP_ET[idx]);
  const double r2 = rr * rr;
  const double p2 = __builtin_fma(rr, GP_C[3], GP_C[2]);
  const double s2 = __builtin_fma(p1, r2, s1);
  const double r4 = r2 * r2;
  const double tm = __builtin_fma(p2, r4, s2);
  const double sc = __builtin_bit_cast(double, sbits);
  return __builtin_fma(tm, sc, sc);
}
__device__ __forceinline__ void c_dens(const KP &P, const int i, const int j, const int k, const double *si, const double *ti, double *rhoo);
__global__ void k_dens(KP P, const double *si, const double *ti, double *rhoo) {
  MARCH3(c_dens(P, i, j, k, si, ti, rhoo))
}
__device__ __forceinline__ double dens_point(const KP &P, double si, double ti, int i, int j, int k) {
  const double tr = ti + P.tbias;
  const double sr = si + P.sbias;
  const double tr2 = tr * tr, tr3 = tr2 * tr, tr4 = tr3 * tr;
  const double p = P.grav * P.rhoref * (-F1(zz, k) * h_(i, j)) * 1.e-5;
  double rhor = -0.157406 + 6.793952e-2 * tr - 9.095290e-3 * tr2 + 1.001685e-4 * tr3 - 1.120083e-6 * tr4 + 6.536332e-9 * tr4 * tr;
  rhor = rhor + (0.824493 - 4.0899e-3 * tr + 7.6438e-5 * tr2 - 8.2467e-7 * tr3 + 5.3875e-9 * tr4) * sr +
         (-5.72466e-3 + 1.0227e-4 * tr - 1.6546e-6 * tr2) * gpow15(fabs(sr)) + 4.8314e-4 * sr * sr;
  const double cr = 1449.1 + .0821 * p + 4.55 * tr - .045 * tr2 + 1.34 * (sr - 35.);
  rhor = rhor + 1.e5 * p / (cr * cr) * (1. - 2. * p / (cr * cr));
  return rhor / P.rhoref * F2(fsm, i, j);
}
__device__ __forceinline__ void c_dens(const KP &P, const int i, const int j, const int k, const double *si, const double *ti, double *rhoo) {
  if (k > P.kbm1 || i > P.im || j > P.jm) return;
  G3(rhoo, i, j, k) = dens_point(P, G3(si, i, j, k), G3(ti, i, j, k), i, j, k);
}

// ---------------------------------------------------------------------------------------------
// realvertvl -- solver.f:2024-2067.  The zero-gradient edge copies (:2057-2060) become a clamped
// source index; the mask (:2062-2064) is applied on store.
__device__ __forceinline__ void c_realvertvl(const KP &P, const int i, const int j, const int k);
__global__ void k_realvertvl(KP P) {
  MARCH3(c_realvertvl(P, i, j, k))
}
__device__ __forceinline__ void c_realvertvl(const KP &P, const int i, const int j, const int k) {
  double v = 0.;
  if (k <= P.kbm1 && i <= P.im && j <= P.jm) {
    const int a = (P.W && i == 1) ? 2 : ((P.E && i == P.im) ? P.imm1 : i);
    const int b = (P.S && j == 1) ? 2 : ((P.N && j == P.jm) ? P.jmm1 : j);
    if (a >= 2 && a <= P.imm1 && b >= 2 && b <= P.jmm1) {
      const double zzk = F1(zz, k);
#define TPS(ii, jj) (zzk * dt_(ii, jj) + F2(et, ii, jj))
      const double tc = TPS(a, b);
      const double dxr = K2(R2DXSX, a + 1, b);    // 2.0/(dx(i+1,j)+dx(i,j))
      const double dxl = K2(R2DXSX, a, b);        // 2.0/(dx(i,j)+dx(i-1,j))
      const double dyt = K2(R2DYSY, a, b + 1);    // 2.0/(dy(i,j+1)+dy(i,j))
      const double dyb = K2(R2DYSY, a, b);        // 2.0/(dy(i,j)+dy(i,j-1))
      v = 0.5 * (w_(a, b, k) + w_(a, b, k + 1)) +
          0.5 * (u_(a + 1, b, k) * (TPS(a + 1, b) - tc) * dxr + u_(a, b, k) * (tc - TPS(a - 1, b)) * dxl +
                 v_(a, b + 1, k) * (TPS(a, b + 1) - tc) * dyt + v_(a, b, k) * (tc - TPS(a, b - 1)) * dyb) +
          (1.0 + zzk) * (F2(etf, a, b) - F2(etb, a, b)) / P.dti2;
#undef TPS
    }
    v = F2(fsm, i, j) * v;
  }
  F3(wr, i, j, k) = v;
}

// realvertvl, column-marching.  PMC of the cell kernel above: 23 loads per cell (17 of them 2-D
// operands re-fetched at every level), L1/TA 80 % busy, 2.2 ms for 4 algorithmic passes.  Here a
// thread owns the water column: the fifteen 2-D operands live in registers, a level costs five
// loads, the next level's are in flight while this one is evaluated.
struct LevR { double w_n, u_c, u_e, v_c, v_n; };
__device__ __forceinline__ LevR realvertvl_load(const KP &P, int a, int b, int k) {
  LevR L;
  L.w_n = w_(a, b, k + 1); L.u_c = u_(a, b, k); L.u_e = u_(a + 1, b, k); L.v_c = v_(a, b, k); L.v_n = v_(a, b + 1, k);
  return L;
}
__global__ void __launch_bounds__(256) k_realvertvl_col(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const bool act = (i <= P.im && j <= P.jm);
  const int a = (P.W && i == 1) ? 2 : ((P.E && i == P.im) ? P.imm1 : i);
  const int b = (P.S && j == 1) ? 2 : ((P.N && j == P.jm) ? P.jmm1 : j);
  const int kb = P.kb, kbm1 = P.kbm1;
  if (!(act && a >= 2 && a <= P.imm1 && b >= 2 && b <= P.jmm1)) {
    for (int k = 1; k <= kb; k++) F3(wr, i, j, k) = 0.;      // fsm * 0. (fsm is 0 or 1) and the untouched rim
    return;
  }
  const double dt_c = dt_(a, b), dt_e = dt_(a + 1, b), dt_w = dt_(a - 1, b), dt_n = dt_(a, b + 1), dt_s = dt_(a, b - 1);
  const double et_c = F2(et, a, b), et_e = F2(et, a + 1, b), et_w = F2(et, a - 1, b), et_n = F2(et, a, b + 1), et_s = F2(et, a, b - 1);
  const double dxr = K2(R2DXSX, a + 1, b), dxl = K2(R2DXSX, a, b), dyt = K2(R2DYSY, a, b + 1), dyb = K2(R2DYSY, a, b);
  const double detf = F2(etf, a, b) - F2(etb, a, b);
  const double m = F2(fsm, i, j);
  double w_k = w_(a, b, 1);
  LevR c = realvertvl_load(P, a, b, 1), nxt = c;
  for (int k = 1; k <= kbm1; k++) {
    PACE_BARRIER();                                          // the workgroup's wavefronts stay on one level (same pages, same DRAM rows)
    if (k + 1 <= kbm1) nxt = realvertvl_load(P, a, b, k + 1);
    const double zzk = F1(zz, k);
#define TPS(dtv, etv) (zzk * (dtv) + (etv))
    const double tc = TPS(dt_c, et_c);
    double v = 0.5 * (w_k + c.w_n) +
               0.5 * (c.u_e * (TPS(dt_e, et_e) - tc) * dxr + c.u_c * (tc - TPS(dt_w, et_w)) * dxl +
                      c.v_n * (TPS(dt_n, et_n) - tc) * dyt + c.v_c * (tc - TPS(dt_s, et_s)) * dyb) +
               (1.0 + zzk) * detf / P.dti2;
#undef TPS
    F3(wr, i, j, k) = m * v;
    w_k = c.w_n;
    c = nxt;
  }
  F3(wr, i, j, kb) = 0.;
}

// Smagorinsky viscosity, two columns per lane (see k_advave_pair, k_ext.hip): the cell kernel k_aam
// issues 14 loads per cell and runs at 78 % L1/TA busy; here seven aligned 16-byte loads serve two
// cells and every i+-1 operand is a neighbour-lane value.  Needs an even leading dimension.
#define LD2(ptr, ii, jj, kk) (*(const double2 *)&(ptr)[IX3(ii, jj, kk)])
#define A3(name) (P.b3 + (size_t)P3_##name * P.a3)
__device__ __forceinline__ double aam_point(const KP &P, double dx, double dy, double u_c, double u_e, double u_n, double u_ne, double u_s,
                                            double u_se, double v_c, double v_n, double v_e, double v_ne, double v_w, double v_nw) {
  return P.horcon * dx * dy *
         sqrt(sq((u_e - u_c) / dx) + sq((v_n - v_c) / dy) +
              .5 * sq(.25 * (u_n + u_ne - u_s - u_se) / dy + .25 * (v_e + v_ne - v_w - v_nw) / dx));
}
// the same with the four divisions by dx, dy through divi() (their reciprocals from k_coef_static: correctly rounded quotients, the same
// bits -- pomgpu_internal.hpp).  The counters put k_aam_pair on its arithmetic, not on memory (profiles/round3_pmc_kernels.txt: VALU
// 0.13 of every wave's cycles at 8 waves per SIMD): 8 division macros (~69 cycles each) + 2 square roots per lane; forming the
// reciprocals in the kernel (round 2) only moved the divisions.
__device__ __forceinline__ double aam_point_r(const KP &P, const InvD &dx, const InvD &dy, double u_c, double u_e, double u_n, double u_ne, double u_s,
                                              double u_se, double v_c, double v_n, double v_e, double v_ne, double v_w, double v_nw) {
  return P.horcon * dx.b * dy.b *
         sqrt(sq(divi(u_e - u_c, dx)) + sq(divi(v_n - v_c, dy)) +
              .5 * sq(divi(.25 * (u_n + u_ne - u_s - u_se), dy) + divi(.25 * (v_e + v_ne - v_w - v_nw), dx)));
}
// AAM_KCH levels per wavefront, the operands of level k+1 requested while level k is worked on: one level per wavefront was 1.3 M
// wavefronts of seven loads, ~600 cycles of dependent fp64 and two stores each -- 4.0 TB/s of real traffic, bound by that chain at
// 8 waves per SIMD (neither memory nor arithmetic: the divisions through reciprocals bought 1.3 %).
#ifndef AAM_KCH
#define AAM_KCH 7
#endif
struct AamLev { double2 u_s, u_0, u_n, v_0, v_n; };
__global__ void __launch_bounds__(256) k_aam_pair(KP P) {
  // banded XCD-aware decode as MARCH3, with 124 output columns per wavefront; the level index counts chunks of AAM_KCH levels
  const int g_ = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int L_ = g_ >> 6, xcd_ = L_ & 7, m_ = L_ >> 3;
  const int nbx_ = (P.iml / 2 + 61) / 62, bpl_ = nbx_ * (P.g_rb / 4);
  const int nch_ = (P.kbm1 + AAM_KCH - 1) / AAM_KCH;
  const int p_ = m_ % bpl_, t_ = m_ / bpl_;
  const int k0 = (t_ % nch_) * AAM_KCH + 1;
  const int band_ = (t_ / nch_) * 8 + xcd_;
  const int lane = g_ & 63;
  const int ia0 = 2 * ((p_ % nbx_) * 62 + lane - 1) + 1;
  const int j = band_ * P.g_rb + (p_ / nbx_) * 4 + (int)threadIdx.y + 1;
  if (k0 > P.kbm1 || j < 2 || j > P.jmm1) return;            // wave-uniform
  const int k1 = k0 + AAM_KCH - 1 < P.kbm1 ? k0 + AAM_KCH - 1 : P.kbm1;
  const bool out = (lane >= 1 && lane <= 62 && ia0 <= P.iml);
#ifdef POMGPU_EMU
  if (!out) return;
#endif
  const int ia = ia0 < 1 ? 1 : (ia0 > P.iml - 1 ? P.iml - 1 : ia0), ib = ia + 1;
  const int iw = ia > 1 ? ia - 1 : 1, ie = ib < P.iml ? ib + 1 : P.iml;
  const double2 dx = *(const double2 *)&F2(dx, ia, j), dy = *(const double2 *)&F2(dy, ia, j);
#ifndef AAM_PLAIN_DIV
  const double2 rdx = *(const double2 *)&K2(RDX, ia, j), rdy = *(const double2 *)&K2(RDY, ia, j);
  const InvD dxa = {dx.x, rdx.x}, dya = {dy.x, rdy.x}, dxb = {dx.y, rdx.y}, dyb = {dy.y, rdy.y};
#endif
  auto lev = [&](int k) {
    AamLev L;
    L.u_s = LD2(A3(u), ia, j - 1, k); L.u_0 = LD2(A3(u), ia, j, k); L.u_n = LD2(A3(u), ia, j + 1, k);
    L.v_0 = LD2(A3(v), ia, j, k); L.v_n = LD2(A3(v), ia, j + 1, k);
    return L;
  };
  AamLev cur = lev(k0);
  for (int k = k0; k <= k1; k++) {
    const AamLev nxt = lev(k < k1 ? k + 1 : k1);              // in flight during this level (the last one re-requests itself)
    const double2 u_s = cur.u_s, u_0 = cur.u_0, u_n = cur.u_n, v_0 = cur.v_0, v_n = cur.v_n;
    const double uE_s = halo_e(u_s.x, [&] { return u_(ie, j - 1, k); }), uE_0 = halo_e(u_0.x, [&] { return u_(ie, j, k); }),
                 uE_n = halo_e(u_n.x, [&] { return u_(ie, j + 1, k); });
    const double vE_0 = halo_e(v_0.x, [&] { return v_(ie, j, k); }), vE_n = halo_e(v_n.x, [&] { return v_(ie, j + 1, k); });
    const double vW_0 = halo_w(v_0.y, [&] { return v_(iw, j, k); }), vW_n = halo_w(v_n.y, [&] { return v_(iw, j + 1, k); });
#ifndef AAM_PLAIN_DIV
    const double aa = aam_point_r(P, dxa, dya, u_0.x, u_0.y, u_n.x, u_n.y, u_s.x, u_s.y, v_0.x, v_n.x, v_0.y, v_n.y, vW_0, vW_n);
    const double ab = aam_point_r(P, dxb, dyb, u_0.y, uE_0, u_n.y, uE_n, u_s.y, uE_s, v_0.y, v_n.y, vE_0, vE_n, v_0.x, v_n.x);
#else
    const double aa = aam_point(P, dx.x, dy.x, u_0.x, u_0.y, u_n.x, u_n.y, u_s.x, u_s.y, v_0.x, v_n.x, v_0.y, v_n.y, vW_0, vW_n);
    const double ab = aam_point(P, dx.y, dy.y, u_0.y, uE_0, u_n.y, uE_n, u_s.y, uE_s, v_0.y, v_n.y, vE_0, vE_n, v_0.x, v_n.x);
#endif
    if (out && ia0 >= 2 && ia0 <= P.imm1) F3(aam, ia0, j, k) = aa;
    if (out && ia0 + 1 <= P.imm1) F3(aam, ia0 + 1, j, k) = ab;
    cur = nxt;
  }
}
#undef LD2
#undef A3


// ---- launchers --------------------------------------------------------------------------------
void launch_advct_a(pomgpu_ctx *c) { LAUNCH(c, k_advct_a, gridm(c->P), blk2(), c->P); }
void launch_advct_b(pomgpu_ctx *c) { LAUNCH(c, k_advct_b, gridm(c->P), blk2(), c->P); }
void launch_advct_c(pomgpu_ctx *c) { LAUNCH(c, k_advct_c, gridm(c->P), blk2(), c->P); }
void launch_aam(pomgpu_ctx *c) {
  const KP &P = c->P;
#ifdef POMGPU_STORE_F32
  const bool pair_ok = false;                                 // k_aam_pair reads u, v as pairs of doubles
#else
  const bool pair_ok = true;
#endif
  if (pair_ok && P.iml % 2 == 0 && !SW(c, NO_PAIR)) {
    const long nbands = (P.jml + P.g_rb - 1) / P.g_rb, rounds = (nbands + 7) / 8;
    const long bpl = (long)((P.iml / 2 + 61) / 62) * (P.g_rb / 4);
    const long nch = (P.kbm1 + AAM_KCH - 1) / AAM_KCH;          // chunks of levels per wavefront
    LAUNCHN(c, "k_aam_pair", k_aam_pair, dim3((unsigned)(8 * rounds * nch * bpl), 1, 1), blk2(), c->P);
  } else {
    LAUNCH(c, k_aam, gridm(c->P), blk2(), c->P);
  }
}
void launch_roundtrip(pomgpu_ctx *c, double *a, const double *b, int fix_kb) { LAUNCH(c, k_roundtrip, gridm(c->P), blk2(), c->P, a, b, fix_kb); }
void launch_roundtrip_level(pomgpu_ctx *c, double *a, const double *b, int k) { LAUNCH(c, k_roundtrip_level, grid2(c->P), blk2(), c->P, a, b, k); }
void launch_advq_flux(pomgpu_ctx *c, const double *q, const double *qb, double *xf, double *yf) {
  LAUNCH(c, k_advq_flux, gridm(c->P), blk2(), c->P, q, qb, xf, yf);
}
void launch_advq_step(pomgpu_ctx *c, const double *q, const double *qb, double *qf, const double *xf, const double *yf, int zero_else) {
  LAUNCH(c, k_advq_step, gridm(c->P), blk2(), c->P, q, qb, qf, xf, yf, zero_else);
}
void launch_q_filter(pomgpu_ctx *c, int mask) { LAUNCH(c, k_q_filter, gridm(c->P), blk2(), c->P, mask); }
// the filter of the two outermost lines on every side alone (the interior rode on k_profq's back substitution)
void launch_q_filter_rim(pomgpu_ctx *c) {
  const KP &P = c->P;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_q_filter_rim, dim3((len + 63) / 64, 8, P.kb), dim3(64, 1, 1), c->P);
}
void launch_mask_q(pomgpu_ctx *c) { LAUNCH(c, k_mask_q, gridm(c->P), blk2(), c->P); }
void launch_advt1(pomgpu_ctx *c, double *fb, double *f, const double *fclim, double *ff) {
  LAUNCH(c, k_advt1, gridm(c->P), blk2(), c->P, (const double *)fb, (const double *)f, fclim, ff);
  LAUNCH(c, k_copy_kb, grid2(c->P), blk2(), c->P, f, (double *)NULL);
}
void launch_copy_kb(pomgpu_ctx *c, double *f, double *g) { LAUNCH(c, k_copy_kb, grid2(c->P), blk2(), c->P, f, g); }
void launch_advt2_mass(pomgpu_ctx *c) { LAUNCH(c, k_advt2_mass, gridm(c->P), blk2(), c->P); }
void launch_advt2_step(pomgpu_ctx *c, const double *fbmem, const double *f, const double *eta, double *ff, int itera) {
  LAUNCH(c, k_advt2_step, gridm(c->P), blk2(), c->P, fbmem, f, eta, ff, itera);
}
void launch_mask3(pomgpu_ctx *c, double *a, const double *m2) { LAUNCH(c, k_mask3, gridm(c->P), blk2(), c->P, a, m2); }
void launch_smol(pomgpu_ctx *c, const double *ff) { LAUNCH(c, k_smol, gridm(c->P), blk2(), c->P, ff); }
void launch_copy3(pomgpu_ctx *c, double *dst, const double *src) { LAUNCH(c, k_copy3, gridm(c->P), blk2(), c->P, dst, src); }
void launch_advt2_diff(pomgpu_ctx *c, const double *fb, const double *fc, double *ff) { LAUNCH(c, k_advt2_diff, gridm(c->P), blk2(), c->P, fb, fc, ff); }
void launch_ts_update(pomgpu_ctx *c, double fold, double fnew, int rt, int store_rst) {
  if (c->tau_known[0] && c->tau_known[1] && !SW(c, TAU_ARRAYS))
    LAUNCHN(c, "k_ts_update", (k_ts_update<1>), gridm(c->P), blk2(), c->P, fold, fnew, rt, store_rst, c->tau_val[0], c->tau_val[1]);
  else
    LAUNCHN(c, "k_ts_update", (k_ts_update<0>), gridm(c->P), blk2(), c->P, fold, fnew, rt, store_rst, 0., 0.);
}
void launch_restore_fields(pomgpu_ctx *c, double fold, double fnew) { LAUNCH(c, k_restore_fields, gridm(c->P), blk2(), c->P, fold, fnew); }
void launch_mask_ts(pomgpu_ctx *c) { LAUNCH(c, k_mask_ts, gridm(c->P), blk2(), c->P); }
void launch_mask_uv(pomgpu_ctx *c) { LAUNCH(c, k_mask_uv, gridm(c->P), blk2(), c->P); }
void launch_mask_w(pomgpu_ctx *c) { LAUNCH(c, k_mask_w, gridm(c->P), blk2(), c->P); }
void launch_restore(pomgpu_ctx *c, double fold, double fnew) { LAUNCH(c, k_restore, gridm(c->P), blk2(), c->P, fold, fnew); }
void launch_restore_shift(pomgpu_ctx *c) { LAUNCH(c, k_restore_shift, gridm(c->P), blk2(), c->P); }
void launch_restore_load(pomgpu_ctx *c, const double *tr, const double *sr, double tau) { LAUNCH(c, k_restore_load, gridm(c->P), blk2(), c->P, tr, sr, tau); }
void launch_dens(pomgpu_ctx *c, const double *si, const double *ti, double *rhoo) { LAUNCH(c, k_dens, gridm(c->P), blk2(), c->P, si, ti, rhoo); }
void launch_realvertvl(pomgpu_ctx *c) {
  if (SW(c, REALVERTVL_CELLS)) LAUNCH(c, k_realvertvl, gridm(c->P), blk2(), c->P);
  else LAUNCHN(c, "k_realvertvl_col", k_realvertvl_col, grid2(c->P), blk2(), c->P);
}
