// k_vert.hip -- the vertically coupled kernels of the internal mode: one thread per water column,
// threadIdx.x along i.  At a given k all 64 lanes of a wavefront read one contiguous 512-byte row
// segment, so the k-loops stream whole planes coalesced.  The Thomas recurrences (profq, proft,
// profu, profv) keep their ee/gg work vectors in per-thread private arrays instead of the five to
// eleven full 3-D temporaries the reference streams through memory (solver.f:1224-1230,1552-1554).
#include <type_traits>
#include "pomgpu_internal.hpp"
#include "dd_exp.h"

#define dt_(i, j) F2(dt, i, j)
#define dx_(i, j) F2(dx, i, j)
#define dy_(i, j) F2(dy, i, j)
#define h_(i, j) F2(h, i, j)
#define u_(i, j, k) F3(u, i, j, k)
#define v_(i, j, k) F3(v, i, j, k)
#define w_(i, j, k) F3(w, i, j, k)

// Register-resident column kernels: rows per workgroup, same-context A/B at 2048x1536x50 (profiles/round2_reg_kernel_rows.txt):
// profu / profv (one wave per SIMD) 1 row -7 / -3 % against 2, 4 rows +16 %; uv_filter 4 rows -4.5 %, 8 rows another -2 %; int_uvmean 4 rows -3 % (8: +1 %); proft 1 row -0.7 %
#define ROWS_PROFUV 1
#define ROWS_PROFT 1
#ifndef ROWS_UVF
#define ROWS_UVF 8
#endif
#ifndef ROWS_UVM
#define ROWS_UVM 4
#endif
#define COL2                               \
  const int i = TID_I, j = TID_J;          \
  if (i > P.iml || j > P.jml) return;

// ---------------------------------------------------------------------------------------------
// baropg -- solver.f:848-940.  The running vertical sum lives in a register; density anomalies
// rho-rmean are formed on the fly (the reference subtracts rmean in place first, :854).
// sum2d: also leave the vertical integrals drx2d, dry2d of advance.f:152-168 (k_vint)
__global__ void k_baropg(KP P, int sum2d) {
  COL2
  if (i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) {
    if (sum2d) {                                             // rim and padding columns: whatever drhox, drhoy hold there
      double rx = 0., ry = 0.;
      if (i <= P.im && j <= P.jm)
        for (int k = 1; k <= P.kbm1; k++) {
          const double dzk = F1(dz, k);
          rx = rx + F3(drhox, i, j, k) * dzk;
          ry = ry + F3(drhoy, i, j, k) * dzk;
        }
      F2(drx2d, i, j) = rx;
      F2(dry2d, i, j) = ry;
    }
    return;
  }
  const double dtc = dt_(i, j), dtw = dt_(i - 1, j), dts = dt_(i, j - 1);
  const double sx = .25 * (dtc + dtw), sy = .25 * (dtc + dts);
  const double mx = F2(dum, i, j), my = F2(dvm, i, j);
  const double ex = dy_(i, j) + dy_(i - 1, j), ey = dx_(i, j) + dx_(i, j - 1);
  double rc0 = F3(rho, i, j, 1) - F3(rmean, i, j, 1);
  double rw0 = F3(rho, i - 1, j, 1) - F3(rmean, i - 1, j, 1);
  double rs0 = F3(rho, i, j - 1, 1) - F3(rmean, i, j - 1, 1);
  double ax = .5 * P.grav * (-F1(zz, 1)) * (dtc + dtw) * (rc0 - rw0);                       // :859-860
  double ay = .5 * P.grav * (-F1(zz, 1)) * (dtc + dts) * (rc0 - rs0);                       // :895-896
  double ox = P.ramp * (sx * ax * mx * ex), oy = P.ramp * (sy * ay * my * ey);
  F3(drhox, i, j, 1) = ox;                                                                  // :883-885,931
  F3(drhoy, i, j, 1) = oy;
  double rx = 0., ry = 0.;
  rx = rx + ox * F1(dz, 1);
  ry = ry + oy * F1(dz, 1);
  for (int k = 2; k <= P.kbm1; k++) {
    PACE_BARRIER();                                          // the workgroup's wavefronts stay on one level (same pages, same DRAM rows)
    const double rc = F3(rho, i, j, k) - F3(rmean, i, j, k);
    const double rw = F3(rho, i - 1, j, k) - F3(rmean, i - 1, j, k);
    const double rs = F3(rho, i, j - 1, k) - F3(rmean, i, j - 1, k);
    const double zm = F1(zz, k - 1) - F1(zz, k), zp = F1(zz, k - 1) + F1(zz, k);
    ax = ax + P.grav * .25 * zm * (dtc + dtw) * (rc - rw + rc0 - rw0) +
         P.grav * .25 * zp * (dtc - dtw) * (rc + rw - rc0 - rw0);                           // :867-875
    ay = ay + P.grav * .25 * zm * (dtc + dts) * (rc - rs + rc0 - rs0) +
         P.grav * .25 * zp * (dtc - dts) * (rc + rs - rc0 - rs0);                           // :903-911
    ox = P.ramp * (sx * ax * mx * ex); oy = P.ramp * (sy * ay * my * ey);
    F3(drhox, i, j, k) = ox;
    F3(drhoy, i, j, k) = oy;
    rx = rx + ox * F1(dz, k);
    ry = ry + oy * F1(dz, k);
    rc0 = rc; rw0 = rw; rs0 = rs;
  }
  if (sum2d) { F2(drx2d, i, j) = rx; F2(dry2d, i, j) = ry; }
  F3(drhox, i, j, P.kb) = P.ramp * F3(drhox, i, j, P.kb);                                   // :928-935, k=kb
  F3(drhoy, i, j, P.kb) = P.ramp * F3(drhoy, i, j, P.kb);
}

// The same routine with halo-lane wavefronts and the rows of a workgroup shared through LDS (k_advct_col's layout): the
// anomaly rho-rmean of a cell is formed ONCE, by the lane that owns the cell; the western neighbour's comes from the neighbour
// lane, the southern one from the wavefront below through a slab (the first wavefront of a workgroup also forms the row under
// the workgroup).  Two loads per wavefront and level instead of six, the next level's in flight, one barrier per level.
// Expressions and their order are k_baropg's.
#ifndef BPG_ROWS
#define BPG_ROWS 16                                         /* kbench, one context: 2 rows 1.40, 4 1.30, 8 1.24, 16 1.15 ms; one thread per column (k_baropg) 1.32 */
#endif
static_assert(COL_WX == 1, "k_baropg_rs / k_vertvl_rs take threadIdx.y as the row inside the workgroup: one wavefront per workgroup row");
__global__ void __launch_bounds__(64 * BPG_ROWS) k_baropg_rs(KP P, int sum2d) {
  HALO_XCD_DECODE_R(BPG_ROWS)
  const int r = WAVE_UNIFORM((int)threadIdx.y);
  const bool jvalid = j <= P.jml;
  const int jc = jvalid ? j : P.jml;
  const bool own = jvalid && lane >= 1 && lane <= 62 && i0 >= 1 && i0 <= P.iml;
#ifdef POMGPU_EMU
  if (!own) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);
  const int iw = i > 1 ? i - 1 : 1, js = jc > 1 ? jc - 1 : 1;
  const bool in = own && i0 >= 2 && i0 <= P.imm1 && jc >= 2 && jc <= P.jmm1;
  const int kbm1 = P.kbm1;
  const double dtc = dt_(i, jc), dts = dt_(i, js);
  const double dtw = halo_w(dtc, [&] { return dt_(iw, jc); });
  const double sx = .25 * (dtc + dtw), sy = .25 * (dtc + dts);
  const double mx = F2(dum, i, jc), my = F2(dvm, i, jc);
  const double dyc = dy_(i, jc);
  const double ex = dyc + halo_w(dyc, [&] { return dy_(iw, jc); }), ey = dx_(i, jc) + dx_(i, js);
  const BufA brho = BUF3(A3(rho)), brm = BUF3(A3(rmean)), box = BUF3(A3(drhox)), boy = BUF3(A3(drhoy));
  const unsigned oc = BOFF2(i, jc), lvb = LVB;
  const unsigned osh = (r == 0) ? BOFF2(i, js) : BOFF_NONE;   // the row under the workgroup: its first wavefront's job
  const unsigned ost = in ? oc : BOFF_NONE;
#ifndef POMGPU_EMU
  __shared__ double slab[2][BPG_ROWS + 1][64];
#endif
  struct Lev { double rho, rm, rho_s, rm_s; };
  auto load = [&](Lev &L, int k) {
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    L.rho = bld(brho, oc, lv); L.rm = bld(brm, oc, lv);
    L.rho_s = bld(brho, osh, lv); L.rm_s = bld(brm, osh, lv);
  };
  double rc0 = 0., rw0 = 0., rs0 = 0., ax = 0., ay = 0., rx = 0., ry = 0.;
  auto step = [&](const int k, const Lev &cur, Lev &nxt) {
    const int par = k & 1;
    load(nxt, k + 2 <= kbm1 ? k + 2 : kbm1);                // two levels ahead: in flight during this iteration and the next
    const double rc = cur.rho - cur.rm;
#ifndef POMGPU_EMU
    slab[par][r + 1][lane] = rc;
    if (r == 0) slab[par][0][lane] = cur.rho_s - cur.rm_s;
    __syncthreads();
    const double rs = slab[par][r][lane];
#else
    (void)par;
    const double rs = F3(rho, i, js, k) - F3(rmean, i, js, k);
#endif
    const double rw = halo_w(rc, [&] { return F3(rho, iw, jc, k) - F3(rmean, iw, jc, k); });
    if (k == 1) {
      ax = .5 * P.grav * (-F1(zz, 1)) * (dtc + dtw) * (rc - rw);                              // :859-860
      ay = .5 * P.grav * (-F1(zz, 1)) * (dtc + dts) * (rc - rs);                              // :895-896
    } else {
      const double zm = F1(zz, k - 1) - F1(zz, k), zp = F1(zz, k - 1) + F1(zz, k);
      ax = ax + P.grav * .25 * zm * (dtc + dtw) * (rc - rw + rc0 - rw0) +
           P.grav * .25 * zp * (dtc - dtw) * (rc + rw - rc0 - rw0);                           // :867-875
      ay = ay + P.grav * .25 * zm * (dtc + dts) * (rc - rs + rc0 - rs0) +
           P.grav * .25 * zp * (dtc - dts) * (rc + rs - rc0 - rs0);                           // :903-911
    }
    const double ox = P.ramp * (sx * ax * mx * ex), oy = P.ramp * (sy * ay * my * ey);        // :883-885,931
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    bst(box, ost, lv, ox);
    bst(boy, ost, lv, oy);
    rx = rx + ox * F1(dz, k);
    ry = ry + oy * F1(dz, k);
    rc0 = rc; rw0 = rw; rs0 = rs;
  };
  Lev ra, rb, rc_;
  load(ra, 1);
  load(rb, 2 <= kbm1 ? 2 : kbm1);
  rc_ = ra;
  for (int k = 1; k <= kbm1; k += 3) {                        // the conditions are uniform for the whole workgroup (barrier inside)
    step(k, ra, rc_);
    if (k + 1 <= kbm1) step(k + 1, rb, ra);
    if (k + 2 <= kbm1) step(k + 2, rc_, rb);
  }
  if (!own) return;
  if (in) {
    if (sum2d) { F2(drx2d, i, jc) = rx; F2(dry2d, i, jc) = ry; }
    F3(drhox, i, jc, P.kb) = P.ramp * F3(drhox, i, jc, P.kb);                                 // :928-935, k=kb
    F3(drhoy, i, jc, P.kb) = P.ramp * F3(drhoy, i, jc, P.kb);
  } else if (sum2d) {                                         // rim and padding columns: whatever drhox, drhoy hold there
    double qx = 0., qy = 0.;
    if (i <= P.im && jc <= P.jm)
      for (int k = 1; k <= kbm1; k++) {
        const double dzk = F1(dz, k);
        qx = qx + F3(drhox, i, jc, k) * dzk;
        qy = qy + F3(drhoy, i, jc, k) * dzk;
      }
    F2(drx2d, i, jc) = qx;
    F2(dry2d, i, jc) = qy;
  }
}

// ---------------------------------------------------------------------------------------------
// baropg_mcc -- solver.f:943-1159 (npg = 2): McCalpin's 4th-order pressure gradient.  Same shape as
// k_baropg: the anomaly rho-rmean is formed on the fly, drho / rhou of the level above ride in
// registers, the reference's four automatic 3-D arrays are gone.  The one extra ghost column / row
// (rho4th(0,j,k), d4th(0,j), ...(i,0,...)) that order2d_mpi / order3d_mpi deliver on several tiles sits
// in the small buffers KP.g4[] (filled by k_order_pack + the order hook); on a physical edge the
// correction starts one cell further in, exactly as the reference's two branches (:980-1024).
// (1./24.), (1./24), (1./16.) are REAL(4) constants in the reference.
#define RA(ii, jj, kk) (F3(rho, ii, jj, kk) - F3(rmean, ii, jj, kk))
__global__ void k_order_pack(KP P, double *send_e, double *send_n) {     // what order2d/3d_mpi send: column iml-2, row jml-2
  const int t = TID_I, k = (int)blockIdx.y;                              // k = 0: d, k >= 1: rho-rmean at level k
  if (t <= P.jml && send_e) send_e[(size_t)k * P.jml + (t - 1)] = k == 0 ? F2(d, P.iml - 2, t) : RA(P.iml - 2, t, k);
  if (t <= P.iml && send_n) send_n[(size_t)k * P.iml + (t - 1)] = k == 0 ? F2(d, t, P.jml - 2) : RA(t, P.jml - 2, k);
}
__global__ void k_baropg_mcc(KP P, int sum2d) {
  COL2
  if (i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) {
    if (sum2d) {                                             // rim and padding columns: whatever drhox, drhoy hold there
      double rx = 0., ry = 0.;
      if (i <= P.im && j <= P.jm)
        for (int k = 1; k <= P.kbm1; k++) {
          const double dzk = F1(dz, k);
          rx = rx + F3(drhox, i, j, k) * dzk;
          ry = ry + F3(drhoy, i, j, k) * dzk;
        }
      F2(drx2d, i, j) = rx;
      F2(dry2d, i, j) = ry;
    }
    return;
  }
  const double c24 = (double)(1.f / 24.f), c16 = (double)(1.f / 16.f);
  const bool cx = (i >= (P.W ? 3 : 2)), cy = (j >= (P.S ? 3 : 2));      // 4th-order correction applies
  const bool gx = (i == 2), gy = (j == 2);                               // its outermost operand is the extra ghost
  const double *gw3 = P.g4[0], *gs3 = P.g4[1], *gw2 = P.g4[2], *gs2 = P.g4[3];
  const int iww = gx ? 1 : i - 2, jss = gy ? 1 : j - 2;                  // (in-range dummy when the ghost is used)
  const double mxc = F2(dum, i, j), mxe = F2(dum, i + 1, j), mxw = F2(dum, i - 1, j);
  const double myc = F2(dvm, i, j), myn = F2(dvm, i, j + 1), mys = F2(dvm, i, j - 1);
  const double dc = F2(d, i, j), de = F2(d, i + 1, j), dw = F2(d, i - 1, j), dn = F2(d, i, j + 1), ds = F2(d, i, j - 1);
  const double dww = gx ? gw2[j - 1] : F2(d, iww, j), dss = gy ? gs2[i - 1] : F2(d, i, jss);
  double ddxx = (dc - dw) * mxc, d4x = .5 * (dc + dw) * mxc;                                  // :976-977
  if (cx) {
    ddxx = ddxx - c24 * (mxe * (de - dc) - 2 * (dc - dw) + mxw * (dw - dww));                 // :993-996
    d4x = d4x + c16 * (mxe * (dc - de) + mxw * (dw - dww));                                   // :997-999
  }
  double ddxy = (dc - ds) * myc, d4y = .5 * (dc + ds) * myc;                                  // :1073-1074
  if (cy) {
    ddxy = ddxy - c24 * (myn * (dn - dc) - 2 * (dc - ds) + mys * (ds - dss));                 // :1090-1093
    d4y = d4y + c16 * (myn * (dc - dn) + mys * (ds - dss));                                   // :1094-1096
  }
  const double sx = .25 * (dt_(i, j) + dt_(i - 1, j)), ex = dy_(i, j) + dy_(i - 1, j);
  const double sy = .25 * (dt_(i, j) + dt_(i, j - 1)), ey = dx_(i, j) + dx_(i, j - 1);
  double ax = 0., ay = 0., drx_m = 0., rux_m = 0., dry_m = 0., ruy_m = 0., rx = 0., ry = 0.;
  for (int k = 1; k <= P.kbm1; k++) {
    const double rc = RA(i, j, k), re = RA(i + 1, j, k), rw = RA(i - 1, j, k), rn = RA(i, j + 1, k), rs = RA(i, j - 1, k);
    const double rww = gx ? gw3[(size_t)(k - 1) * P.jml + (j - 1)] : RA(iww, j, k);
    const double rss = gy ? gs3[(size_t)(k - 1) * P.iml + (i - 1)] : RA(i, jss, k);
    double drx = (rc - rw) * mxc, rux = 0.5 * (rc + rw) * mxc;                                // :971-974
    if (cx) {
      drx = drx - c24 * (mxe * (re - rc) - 2 * (rc - rw) + mxw * (rw - rww));                 // :984-987
      rux = rux + c16 * (mxe * (rc - re) + mxw * (rw - rww));                                 // :988-990
    }
    double dry = (rc - rs) * myc, ruy = .5 * (rc + rs) * myc;                                 // :1068-1071
    if (cy) {
      dry = dry - c24 * (myn * (rn - rc) - 2 * (rc - rs) + mys * (rs - rss));                 // :1081-1084
      ruy = ruy + c16 * (myn * (rc - rn) + mys * (rs - rss));                                 // :1085-1087
    }
    if (k == 1) {
      ax = P.grav * (-F1(zz, 1)) * d4x * drx;                                                 // :1029
      ay = P.grav * (-F1(zz, 1)) * d4y * dry;                                                 // :1126
    } else {
      ax = ax + P.grav * 0.5 * F1(dzz, k - 1) * d4x * (drx_m + drx) +
           P.grav * 0.5 * (F1(zz, k - 1) + F1(zz, k)) * ddxx * (rux - rux_m);                 // :1036-1041
      ay = ay + P.grav * 0.5 * F1(dzz, k - 1) * d4y * (dry_m + dry) +
           P.grav * 0.5 * (F1(zz, k - 1) + F1(zz, k)) * ddxy * (ruy - ruy_m);                 // :1133-1138
    }
    const double ox = P.ramp * (sx * ax * mxc * ex), oy = P.ramp * (sy * ay * myc * ey);      // :1049-1051, :1146-1148, :1157-1158
    F3(drhox, i, j, k) = ox;
    F3(drhoy, i, j, k) = oy;
    rx = rx + ox * F1(dz, k);
    ry = ry + oy * F1(dz, k);
    drx_m = drx; rux_m = rux; dry_m = dry; ruy_m = ruy;
  }
  F3(drhox, i, j, P.kb) = P.ramp * F3(drhox, i, j, P.kb);                                     // :1155-1160, k = kb
  F3(drhoy, i, j, P.kb) = P.ramp * F3(drhoy, i, j, P.kb);
  if (sum2d) { F2(drx2d, i, j) = rx; F2(dry2d, i, j) = ry; }
}
#undef RA

// ---------------------------------------------------------------------------------------------
// mode_internal: make the depth mean of (u,v) equal (ua,va) -- advance.f:365-393
__global__ void k_int_uvmean(KP P) {
  COL2
  double su = 0., sv = 0.;
  for (int k = 1; k <= P.kbm1; k++) {
    const double dzk = F1(dz, k);
    su = su + u_(i, j, k) * dzk;
    sv = sv + v_(i, j, k) * dzk;
  }
  if (j <= P.jm && i >= 2 && i <= P.im) {
    const double c = (F2(utb, i, j) + F2(utf, i, j)) / (dt_(i, j) + dt_(i - 1, j));
    for (int k = 1; k <= P.kbm1; k++) F3(u, i, j, k) = (u_(i, j, k) - su) + c;
  }
  if (i <= P.im && j >= 2 && j <= P.jm) {
    const double c = (F2(vtb, i, j) + F2(vtf, i, j)) / (dt_(i, j) + dt_(i, j - 1));
    for (int k = 1; k <= P.kbm1; k++) F3(v, i, j, k) = (v_(i, j, k) - sv) + c;
  }
  F2(tps, i, j) = sv;
}

// ---------------------------------------------------------------------------------------------
// vertvl (+ the fsm mask of bcondorl(5) when mask != 0) -- solver.f:1970-2021, bounds_forcing.f:550-561
__global__ void k_vertvl(KP P, int mask) {
  COL2
  if (i > P.im || j > P.jm) return;
  const double m = mask ? F2(fsm, i, j) : 1.;
  if (i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) {
    if (mask) for (int k = 1; k <= P.kbm1; k++) F3(w, i, j, k) = w_(i, j, k) * m;
    return;
  }
  const double cw = .25 * (dy_(i, j) + dy_(i - 1, j)) * (dt_(i, j) + dt_(i - 1, j));
  const double ce = .25 * (dy_(i + 1, j) + dy_(i, j)) * (dt_(i + 1, j) + dt_(i, j));
  const double cs = .25 * (dx_(i, j) + dx_(i, j - 1)) * (dt_(i, j) + dt_(i, j - 1));
  const double cn = .25 * (dx_(i, j + 1) + dx_(i, j)) * (dt_(i, j + 1) + dt_(i, j));
  const double area = dx_(i, j) * dy_(i, j);
  const double det = (F2(etf, i, j) - F2(etb, i, j)) / P.dti2;
  double wk = 0.5 * (F2(vfluxb, i, j) + F2(vfluxf, i, j));                                  // :2004
  F3(w, i, j, 1) = mask ? wk * m : wk;
  for (int k = 1; k <= P.kbm1; k++) {
    PACE_BARRIER();                                          // the workgroup's wavefronts stay on one level (same pages, same DRAM rows)
    wk = wk + F1(dz, k) * ((ce * u_(i + 1, j, k) - cw * u_(i, j, k) + cn * v_(i, j + 1, k) - cs * v_(i, j, k)) / area + det);
    F3(w, i, j, k + 1) = (mask && k + 1 <= P.kbm1) ? wk * m : wk;
  }
}

// The same routine with halo-lane wavefronts and tall workgroups (k_baropg_rs's layout): u(i+1) is the eastern neighbour lane's
// value, v(j+1) the value of the wavefront above, through an LDS slab (the last wavefront of a workgroup also loads the row above
// the workgroup): two loads per wavefront and level instead of four, two levels in flight; the division by the cell area
// through its reciprocal (divi: the same quotient, bit for bit).  Expressions and their order are k_vertvl's.
#ifndef VVL_ROWS
#define VVL_ROWS 8                                          /* kbench, one context: 8 rows 0.830, 16 rows 0.839 ms; one thread per column (k_vertvl) 0.870 */
#endif
__global__ void __launch_bounds__(64 * VVL_ROWS) k_vertvl_rs(KP P, int mask) {
  HALO_XCD_DECODE_R(VVL_ROWS)
  const int r = WAVE_UNIFORM((int)threadIdx.y), j0w = j - r;
  const bool jvalid = j <= P.jml;
  const int jc = jvalid ? j : P.jml;
  const bool own = jvalid && lane >= 1 && lane <= 62 && i0 >= 1 && i0 <= P.im && jc <= P.jm;
#ifdef POMGPU_EMU
  if (!own) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const int js = jc > 1 ? jc - 1 : 1, jn = jc < P.jml ? jc + 1 : P.jml;
  const bool in = own && i0 >= 2 && i0 <= P.imm1 && jc >= 2 && jc <= P.jmm1;
  const int kbm1 = P.kbm1;
  const double m = mask ? F2(fsm, i, jc) : 1.;
  const double dyc = dy_(i, jc), dtc = dt_(i, jc), dxc = dx_(i, jc);
  const double dyw = halo_w(dyc, [&] { return dy_(iw, jc); }), dtw = halo_w(dtc, [&] { return dt_(iw, jc); });
  const double dye = halo_e(dyc, [&] { return dy_(ie, jc); }), dte = halo_e(dtc, [&] { return dt_(ie, jc); });
  const double cw = .25 * (dyc + dyw) * (dtc + dtw);
  const double ce = .25 * (dye + dyc) * (dte + dtc);
  const double cs = .25 * (dxc + dx_(i, js)) * (dtc + dt_(i, js));
  const double cn = .25 * (dx_(i, jn) + dxc) * (dt_(i, jn) + dtc);
  const InvD area = inv_of(dxc * dyc);
  const double det = (F2(etf, i, jc) - F2(etb, i, jc)) / P.dti2;
  const BufA bu = BUF3(A3(u)), bv = BUF3(A3(v)), bw = BUF3(A3(w));
  const unsigned oc = BOFF2(i, jc), lvb = LVB;
  const int jh = j0w + VVL_ROWS <= P.jml ? j0w + VVL_ROWS : P.jml;     // the row above the workgroup: its last wavefront's job
  const unsigned onh = (r == VVL_ROWS - 1) ? BOFF2(i, jh) : BOFF_NONE;
  const unsigned ost = in ? oc : BOFF_NONE;
#ifndef POMGPU_EMU
  __shared__ double slab[2][VVL_ROWS + 1][64];
#endif
  struct Lev { double u, v, vh; };
  auto load = [&](Lev &L, int k) {
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    L.u = bld(bu, oc, lv); L.v = bld(bv, oc, lv); L.vh = bld(bv, onh, lv);
  };
  double wk = 0.5 * (F2(vfluxb, i, jc) + F2(vfluxf, i, jc));                                  // :2004
  bst(bw, ost, 0u, mask ? wk * m : wk);
  auto step = [&](const int k, const Lev &cur, Lev &nxt) {
    const int par = k & 1;
    load(nxt, k + 2 <= kbm1 ? k + 2 : kbm1);                // two levels ahead
#ifndef POMGPU_EMU
    slab[par][r][lane] = cur.v;
    if (r == VVL_ROWS - 1) slab[par][VVL_ROWS][lane] = cur.vh;
    __syncthreads();
    const double v_n = slab[par][r + 1][lane];
#else
    (void)par;
    const double v_n = F3(v, i, jn, k);
#endif
    const double u_e = halo_e(cur.u, [&] { return F3(u, ie, jc, k); });
    wk = wk + F1(dz, k) * (divi(ce * u_e - cw * cur.u + cn * v_n - cs * cur.v, area) + det);
    bst(bw, ost, (unsigned)WAVE_UNIFORM(k) * lvb, (mask && k + 1 <= kbm1) ? wk * m : wk);
  };
  Lev ra, rb, rc;
  load(ra, 1);
  load(rb, 2 <= kbm1 ? 2 : kbm1);
  rc = ra;
  for (int k = 1; k <= kbm1; k += 3) {                        // the conditions are uniform for the whole workgroup (barrier inside)
    step(k, ra, rc);
    if (k + 1 <= kbm1) step(k + 1, rb, ra);
    if (k + 2 <= kbm1) step(k + 2, rc, rb);
  }
  if (own && !in && mask)                                     // rim columns: the mask of bcondorl(5) alone
    for (int k = 1; k <= kbm1; k++) F3(w, i, jc, k) = w_(i, jc, k) * m;
}

// ---------------------------------------------------------------------------------------------
// profq -- solver.f:1212-1538
// (1) surface/bottom boundary values (2-D) -- :1281-1288   scratch: s2[4]=utau2
__global__ void k_profq_bc(KP P) {
  COL2
  if (i > P.im || j > P.jm) return;
  double ut = 0.;
  if (i <= P.imm1 && j <= P.jmm1) {
    ut = sqrt(sq(.5 * (F2(wusurf, i, j) + F2(wusurf, i + 1, j))) + sq(.5 * (F2(wvsurf, i, j) + F2(wvsurf, i, j + 1))));
    F3(uf, i, j, P.kb) =
        sqrt(sq(.5 * (F2(wubot, i, j) + F2(wubot, i + 1, j))) + sq(.5 * (F2(wvbot, i, j) + F2(wvbot, i, j + 1)))) * P.const1_profq;
  }
  G2(P.s2[4], i, j) = ut;
}
// speed of sound (squared root form) at one level -- :1308-1316
__device__ __forceinline__ double profq_cc(const KP &P, int i, int j, int k) {
  const double tp = F3(t, i, j, k) + P.tbias;
  const double sp = F3(s, i, j, k) + P.sbias;
  const double p = P.grav * P.rhoref * (-F1(zz, k) * h_(i, j)) * 1.e-4;
  double cc = 1449.1 + .00821 * p + 4.55 * tp - .045 * sq(tp) + 1.34 * (sp - 35.0);
  cc = cc / sqrt((1. - .01642 * p / cc) * (1. - 0.40 * p / sq(cc)));
  return cc;
}
__device__ __forceinline__ double profq_cc_v(const KP &P, double t, double s_, double hij, int k) {   // profq_cc on loaded operands
  const double tp = t + P.tbias;
  const double sp = s_ + P.sbias;
  const double p = P.grav * P.rhoref * (-F1(zz, k) * hij) * 1.e-4;
  double cc = 1449.1 + .00821 * p + 4.55 * tp - .045 * sq(tp) + 1.34 * (sp - 35.0);
  cc = cc / sqrt((1. - .01642 * p / cc) * (1. - 0.40 * p / sq(cc)));
  return cc;
}
// rho_rt: rho's round trip through rho - rmean has been deferred (pomgpu_ctx::rho_rt_pending): applied to what is loaded
__device__ __forceinline__ double profq_rho(const KP &P, int i, int j, int k, int rho_rt) {
  const double r = F3(rho, i, j, k);
  if (!rho_rt) return r;
  const double m = F3(rmean, i, j, k);
  return (r - m) + m;
}
__device__ __forceinline__ double profq_boygr(const KP &P, int i, int j, int k, double ccm, double cck, int rho_rt = 0) {   // :1327-1330
  return P.grav * (profq_rho(P, i, j, k - 1, rho_rt) - profq_rho(P, i, j, k, rho_rt)) / (F1(dzz, k - 1) * h_(i, j)) +
         sq(P.grav) * 2. / (sq(ccm) + sq(cck));
}
// (2) shear + buoyancy production (exchanged before the solves) -- :1359-1373   scratch: s3[0]=prod
__global__ void k_profq_prod(KP P, int rho_rt) {
  COL2
  if (i > P.im || j > P.jm) return;
  double *prod = P.s3[0];
  G3(prod, i, j, 1) = 0.;
  G3(prod, i, j, P.kb) = 0.;
  const bool in = (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1);
  if (!in) {
    for (int k = 2; k <= P.kbm1; k++) G3(prod, i, j, k) = 0.;
    return;
  }
  const double dh = h_(i, j) + F2(etf, i, j);
  const double sef = 1., shiw = 0.;
  double ccm = profq_cc(P, i, j, 1);
  for (int k = 2; k <= P.kbm1; k++) {
    const double cck = profq_cc(P, i, j, k);
    const double bg = profq_boygr(P, i, j, k, ccm, cck, rho_rt);
    const double km = F3(km, i, j, k);
    double p = km * .25 * sef *
                   (sq(u_(i, j, k) - u_(i, j, k - 1) + u_(i + 1, j, k) - u_(i + 1, j, k - 1)) +
                    sq(v_(i, j, k) - v_(i, j, k - 1) + v_(i, j + 1, k) - v_(i, j + 1, k - 1))) /
                   sq(F1(dzz, k - 1) * dh) -
               shiw * km * bg;
    p = p + F3(kh, i, j, k) * bg;
    G3(prod, i, j, k) = p;
    ccm = cck;
  }
}
// (2') tiles, the library's own exchange: the owned columns form their production term inside k_profq
// (fuse_prod = 2); the exchange at :1374 only needs the lines it sends (columns 2 / imm1, rows 2 / jmm1) and zeros
// on the rim where no neighbour will write.  One thread per (cell of a line, level): a column walk here would be
// eight lines of threads each waiting out 50 levels of dependent loads.
__global__ void k_profq_prod_lines(KP P, int rho_rt) {
  const int t = TID_I, line = (int)blockIdx.y, k = (int)blockIdx.z + 1;        // lines: i = 1, 2, imm1, im, then j = 1, 2, jmm1, jm
  int i, j;
  if (line < 4) { if (t > P.jm) return; j = t; i = line == 0 ? 1 : (line == 1 ? 2 : (line == 2 ? P.imm1 : P.im)); }
  else { if (t > P.im) return; i = t; j = line == 4 ? 1 : (line == 5 ? 2 : (line == 6 ? P.jmm1 : P.jm)); }
  double *prod = P.s3[0];
  double p = 0.;
  // a line of owned cells (i = 2, imm1; j = 2, jmm1) is only READ by the pack kernel of the exchange that follows: it is evaluated
  // where that side has a neighbour (whole-row tiles: two lines of the eight); the ghost lines hold 0 until the exchange fills them
  const bool wanted = line == 1 ? !P.W : (line == 2 ? !P.E : (line == 5 ? !P.S : (line == 6 ? !P.N : true)));
  if (!wanted) return;
  if (k >= 2 && k <= P.kbm1 && i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1) {
    const double dh = h_(i, j) + F2(etf, i, j);
    const double sef = 1., shiw = 0.;
    const double ccm = profq_cc(P, i, j, k - 1), cck = profq_cc(P, i, j, k);
    const double bg = profq_boygr(P, i, j, k, ccm, cck, rho_rt);
    const double km = F3(km, i, j, k);
    p = km * .25 * sef *
            (sq(u_(i, j, k) - u_(i, j, k - 1) + u_(i + 1, j, k) - u_(i + 1, j, k - 1)) +
             sq(v_(i, j, k) - v_(i, j, k - 1) + v_(i, j + 1, k) - v_(i, j + 1, k - 1))) /
            sq(F1(dzz, k - 1) * dh) -
        shiw * km * bg;
    p = p + F3(kh, i, j, k) * bg;
  }
  G3(prod, i, j, k) = p;
}
// (3) everything else: length scale, stability, two Thomas solves, new km/kh/kq.
//     ONE forward walk over the column does everything that is local to a level -- |q2b|, |q2lb|,
//     length scale, gh -> sm/sh, dtef, the new km/kh/kq -- and advances BOTH forward eliminations
//     (q2 and q2l share a, c and prod); one backward walk finishes uf and vf.  Every operand is read
//     once and every result written once; only the four elimination vectors ee1,gg1,ee2,gg2 are
//     private per-thread arrays.  (The first cut re-read l three times and dtef twice and kept gh
//     and two generations of ee/gg: 53 GB of HBM traffic per launch at 2048x1536x50 against 26 GB
//     algorithmic, rocprofv3 FETCH_SIZE/WRITE_SIZE.)
// fuse_filter (inside pomgpu_advance): the Asselin filter and time rotation of q2, q2l (advance.f:416-421,
// with bcond(6)'s mask) ride on the back substitution of the interior columns -- uf, vf never reach
// memory there (only their level-kb left-overs do, see k_q_filter); the two outermost lines of columns
// keep the plain path (bcond(6), which also reads their old q2/q2l, and on several tiles the exchange
// come between the solve and their filter).
// The elimination vectors do not fit registers or LDS at a useful occupancy (4 x kb doubles per column); they live in
// memory: gg1 in uf and gg2 in vf IN PLACE (the right-hand side of a level is consumed before its gg is written, and
// the back substitution overwrites them or -- with the fused filter -- nobody reads uf, vf of those columns again),
// ee1 / ee2 in the scratch arrays s3[4] / s3[5].  Same traffic as the private (scratch-memory) arrays of the first
// version, but ordinary buffer accesses that take part in the exact vmcnt bookkeeping below.
//
// What the ISA of the first version showed (k_profq spent 62-65 % of its wave cycles in s_waitcnt, SQ_WAIT_ANY): the
// ~15 operand loads of level k+1 were requested at the top of iteration k, but half of them sat inside branches
// (`if (mid)`, `if (pin)`, `if (!repl)` ...) -- where such branches join, the compiler cannot know how many memory
// operations are outstanding and waits for vmcnt(0), i.e. for the batch it has just issued: every level paid a full
// memory round trip.  Here NO memory instruction of the two walks sits inside a branch: every lane requests every
// operand of every level (a value that is not needed is not used), and a lane / level with nothing to store aims
// outside the buffer (BOFF_NONE: the hardware drops the store).  Buffer addressing throughout (no address arithmetic
// in vector registers), two register sets alternating as current / next level.
// FP: 0 = the production term comes from s3[0] (k_profq_prod), 1 = formed here for interior columns, 0 on the rim (one
// tile), 2 = formed here for interior columns, s3[0] on the rim (tiles, the library's exchange).  FF: fused filter.
struct LevQ { double t, s, rho, rm, q2b, q2lb, q2, km, kh, uf, vf, uc, ue, vc, vn, kq1, prod; };
struct LevB { double e1, g1, e2, g2, qb, qlb, q, ql; };
// rho_rt: rho is still to make its round trip through rho - rmean (pomgpu_ctx::rho_rt_pending): applied to the loaded value
// ROWS wavefronts (rows) per workgroup, kept on the same level by one s_barrier per two levels (PACE_BARRIER, no fence):
// the walk down touches 17 arrays x 2 levels, every (array, level) in a 2 MiB page of its own (level stride 25 MB at
// 2048x1536) -- wavefronts of a CU that drift apart by a few levels cycle through more pages than the CU's first-level
// TLB holds, and then EVERY access misses (TCP_UTCL1_TRANSLATION_MISS 4.8e7 per launch = 2 of 3 wave accesses, the
// translation FIFO stalled 27 % of the kernel; 1.7e6 with 8 paced rows -- profiles/round2_tlb_profq.txt).
#ifndef PROFQ_KL
#define PROFQ_KL 19                                         /* 20 = all 160 KB of a CU: see below */
#endif
#ifndef PROFQ_BIG
#define PROFQ_BIG 8                                         /* rows per workgroup on large grids */
#endif
template <int FP, int FF, int ROWS>
// jfirst, jlast: the rows this launch covers (1..jm: everything).  On tiles whose ghost lines are filled by a message round behind this
// kernel (pomgpu_api.hip, rim round Rq) the ghost rows are left out: a 2048x194 tile is 25 row blocks x 32 = 800 workgroups for the 768 that
// are resident at once -- a fourth, nearly empty round -- and 24 x 32 = 768 without its two ghost rows
__global__ void __launch_bounds__(64 * ROWS) k_profq(KP P, int rho_rt_pace, int jfirst, int jlast) {
#ifndef POMGPU_EMU
  // 8-row workgroups (one per CU): 19 levels = 152 KB.  NOT 20 = the whole 160 KB: a library that merely CONTAINS a kernel with
  // 163840 bytes of LDS made processes that share one GPU (tests, rehearsals: bench.py --gpus 4 on one card) run one after the
  // other instead of side by side -- 47.9 against 21.4 ms per step, whether that kernel was launched or not; 155648 bytes do
  // not (profiles/round2_profq_lds_vectors.txt).  2-row workgroups (small grids, tiles): 9 levels = 18 KB, four per CU.
  constexpr int KL = ROWS >= 8 ? PROFQ_KL : (PROFQ_KL < 9 ? PROFQ_KL : 9);
  // ee1, ee2 of the first KL levels wait for the walk up in LDS instead of the scratch arrays (a lane reads back what
  // it wrote itself: no barrier): 2 x 20 x 8 bytes per column is all that 160 KB per CU could hold at 512 columns per CU
  __shared__ double evec[2 * KL][ROWS][64];
#endif
  const int i = TID_I, j = TID_J + jfirst - 1;
  if (i > P.iml || j > P.jml || j > jlast) return;
  const int rho_rt = rho_rt_pace & 1, pace = rho_rt_pace & 2;
  const int ty = (int)threadIdx.y, tx = (int)threadIdx.x;
  if (i > P.im || j > P.jm) return;
  const double a1 = 0.92, b1 = 16.6, a2 = 0.74, b2 = 10.1, c1 = 0.08, e1 = 1.8, e2 = 1.33, surfl = 2.e5;
  const int kb = P.kb, kbm1 = P.kbm1;
  const double dh = h_(i, j) + F2(etf, i, j);
  const double utau2 = G2(P.s2[4], i, j);
  const double l0 = surfl * utau2 / P.grav;                                                 // :1299
  const double umol2 = 2. * P.umol;
  const double z1 = F1(z, 1), zkb = F1(z, kb);
  // stability-function constants (stf = 1) -- :1474-1483
  const double coef4 = 18. * a1 * a1 + 9. * a1 * a2, coef5 = 9. * a1 * a2;
  const double coef1 = a2 * (1. - 6. * a1 / b1 * 1.), coef2 = 3. * a2 * b2 / 1. + 18. * a1 * a2,
               coef3 = a1 * (1. - 3. * c1 - 6. * a1 / b1 * 1.);
  // which cells take this column's km/kh/kq: itself unless it is a physical-edge cell (those copy
  // their inward neighbour, north/south first, then east/west)
  const bool repl = (P.W && i == 1) || (P.E && i == P.im) || (P.S && j == 1) || (P.N && j == P.jm);
  const int ti = (P.W && i == 2) ? 1 : ((P.E && i == P.imm1) ? P.im : 0);
  const int tj = (P.S && j == 2) ? 1 : ((P.N && j == P.jmm1) ? P.jm : 0);
  const double fsm_c = F2(fsm, i, j);
  const double m_ti = F2(fsm, ti ? ti : i, j), m_tj = F2(fsm, i, tj ? tj : j), m_tij = F2(fsm, ti ? ti : i, tj ? tj : j);   // masks of the cells that copy this column
  // the shear + buoyancy production of k_profq_prod is formed here from the same sound speed / density / km / kh this
  // walk reads anyway (FP = 1, 2: the columns 2..imm1 x 2..jmm1; the rim columns -- ghost cells of a neighbour's owned
  // column, or a physical edge -- take the exchanged value from s3[0] (FP = 2) or zero)
  const bool pin = FP && (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1);
  // not the two outermost lines: bcond(6) reads the OLD q2, q2l of columns 2 / imm1 / jmm1 (bounds_forcing.f:262-318)
  const bool ffil = FF && (i >= 3 && i <= P.imm1 - 1 && j >= 3 && j <= P.jmm1 - 1);
  const int ie = i < P.iml ? i + 1 : i, jn = j < P.jml ? j + 1 : j;
  const double hij = h_(i, j);
  const InvD dhk = inv_of(dh * P.kappa);
  // buffers and per-lane offsets
  // descriptors are formed where they are used (uniform pointer -> scalar registers); kept as named values they were
  // carried through the loops in vector registers (too many for the scalar file) at the price of a waterfall loop per load
#define bt BUF3(A3(t))
#define bs_ BUF3(A3(s))
#define brho BUF3(A3(rho))
#define bq2b BUF3(A3(q2b))
#define bq2lb BUF3(A3(q2lb))
#define bq2 BUF3(A3(q2))
#define bq2l BUF3(A3(q2l))
#define bkm BUF3(A3(km))
#define bkh BUF3(A3(kh))
#define bkq BUF3(A3(kq))
#define buf BUF3(A3(uf))
#define bvf BUF3(A3(vf))
#define bu BUF3(A3(u))
#define bv BUF3(A3(v))
#define bl BUF3(A3(l))
#define bdt BUF3(A3(dtef))
#define bpr BUF3(P.s3[0])
#define be1 BUF3(P.s3[4])
#define be2 BUF3(P.s3[5])
  const unsigned oc = BOFF2(i, j), oe = BOFF2(ie, j), on = BOFF2(i, jn), lvb = LVB;
  const unsigned o_k = repl ? BOFF_NONE : oc;               // new km, kh, kq: not on physical-edge columns
  const unsigned o_abs = ffil ? BOFF_NONE : oc;             // |q2b|, |q2lb| on the way down: rewritten on the way up where the filter is fused
  const unsigned o_fil = ffil ? oc : BOFF_NONE, o_uv = ffil ? BOFF_NONE : oc;
  const unsigned o_pr = (FP == 0 || (FP == 2 && !pin)) ? oc : BOFF_NONE;
  const unsigned o_rm = rho_rt ? oc : BOFF_NONE;            // rmean is requested only where it is used (outside the buffer: no traffic)
  // boundary values of the two solves -- :1296-1297, :1417-1425
  const double vbot = P.kappa * (1 + F1(z, kbm1)) * dh * F3(q2, i, j, kbm1);
  const double ufbot = F3(uf, i, j, kb);
  double e1p = 0., g1p = P.cb_profq * utau2;    // ee1, gg1 of level k-1 (level 1: the surface boundary value)
  double e2p = 0., g2p = 0.;                    // ee2, gg2 of level k-1
  double ccm = 0., rhom = 0.;                   // sound speed and density of level k-1
  double kqm = 0., kqc = F3(kq, i, j, 1);       // OLD kq at k-1, k (k+1 arrives with the level: kq1)
  double ucm = 0., uem = 0., vcm = 0., vnm = 0.;   // u(i), u(i+1), v(j), v(j+1) of level k-1
#ifdef PROFQ_OLDWALK
  auto lev = [&](LevQ &L, int k) {              // k = 1..kb; every operand, every lane
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    L.t = bld(bt, oc, lv); L.s = bld(bs_, oc, lv); L.rho = bld(brho, oc, lv); L.rm = bld(BUF3(A3(rmean)), o_rm, lv);
    L.q2b = bld(bq2b, oc, lv); L.q2lb = bld(bq2lb, oc, lv); L.q2 = bld(bq2, oc, lv);
    L.km = bld(bkm, oc, lv); L.kh = bld(bkh, oc, lv);
    L.uf = bld(buf, oc, lv); L.vf = bld(bvf, oc, lv);
    if (FP) { L.uc = bld(bu, oc, lv); L.ue = bld(bu, oe, lv); L.vc = bld(bv, oc, lv); L.vn = bld(bv, on, lv); }
    else L.uc = L.ue = L.vc = L.vn = 0.;
    L.kq1 = bld(bkq, oc, (unsigned)WAVE_UNIFORM(k < kb ? k : kb - 1) * lvb);     // OLD kq of level k+1 (overwritten two iterations from now)
    L.prod = (FP != 1) ? bld(bpr, o_pr, lv) : 0.;
  };
  auto step = [&](const int k, const LevQ &cur, LevQ &nxt) {
    lev(nxt, k + 1 <= kb ? k + 1 : kb);         // in flight during this iteration (the last one re-requests level kb)
    const bool mid = (k >= 2 && k <= kbm1);
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    const double kqp = cur.kq1;
    // ---- level-local quantities
    double cck = 0., rhok = 0.;
    if (k <= kbm1) { cck = profq_cc_v(P, cur.t, cur.s, hij, k); rhok = rho_rt ? (cur.rho - cur.rm) + cur.rm : cur.rho; }
    double q2b = cur.q2b;
    double l, gh = 0.;
    double uck = 0., uek = 0., vck = 0., vnk = 0., bg = 0.;
    if (pin && k <= kbm1) { uck = cur.uc; uek = cur.ue; vck = cur.vc; vnk = cur.vn; }
    double q2lb = 0.;
    if (mid) {
      q2b = fabs(q2b);                                                                      // :1325-1326
      q2lb = fabs(cur.q2lb);
      bg = P.grav * (rhom - rhok) / (F1(dzz, k - 1) * hij) + sq(P.grav) * 2. / (sq(ccm) + sq(cck));   // :1327-1330
      l = fabs(q2lb / q2b);                                                                 // :1338-1344
      if (F1(z, k) > -0.5) l = fmax(l, P.kappa * l0);
      gh = fmin(sq(l) * bg / q2b, .028);
    } else {
      l = (k == 1) ? P.kappa * l0 : 0.;                                                     // :1351-1354
    }
    bst(bq2b, mid ? o_abs : BOFF_NONE, lv, q2b);
    bst(bq2lb, mid ? o_abs : BOFF_NONE, lv, q2lb);
    bst(bl, oc, lv, l);
    const double dtef1 = sqrt(fabs(q2b)) * 1. / (b1 * l + P.small_);                        // :1388-1389
    double dtef2 = dtef1;
    if (mid) dtef2 = dtef1 * (1. + e2 * sq(divi((1. / fabs(F1(z, k) - z1) + 1. / fabs(F1(z, k) - zkb)) * l, dhk)));   // :1429-1432
    bst(bdt, oc, lv, dtef2);
    // ---- both forward eliminations -- :1394-1404, :1436-1446
    if (mid) {
      const double a = -P.dti2 * (kqp + kqc + umol2) * .5 / (F1(dzz, k - 1) * F1(dz, k) * dh * dh);      // :1261-1264
      const double c = -P.dti2 * (kqm + kqc + umol2) * .5 / (F1(dzz, k - 1) * F1(dz, k - 1) * dh * dh);
      double pr;
      if (FP == 0 || (FP == 2 && !pin)) {
        pr = cur.prod;
      } else if (pin) {                                                                     // :1359-1373, as k_profq_prod
        const double sef = 1., shiw = 0.;
        const double km = cur.km;
        pr = km * .25 * sef * (sq(uck - ucm + uek - uem) + sq(vck - vcm + vnk - vnm)) / sq(F1(dzz, k - 1) * dh) - shiw * km * bg;
        pr = pr + cur.kh * bg;
      } else {
        pr = 0.;
      }
      const double g = 1. / (a + c * (1. - e1p) - (2. * P.dti2 * dtef1 + 1.));
      e1p = a * g;
      g1p = (-2. * P.dti2 * pr + c * g1p - cur.uf) * g;
      if (k == 2) {
        e2p = 0.;
        g2p = -P.kappa * F1(z, 2) * dh * cur.q2;
      } else {
        const double rhs = (k == kbm1) ? vbot : cur.vf;
        const double g2 = 1. / (a + c * (1. - e2p) - (P.dti2 * dtef2 + 1.));
        e2p = a * g2;
        g2p = (P.dti2 * (-pr * l * e1) + c * g2p - rhs) * g2;
      }
    }
    // the vectors' level k (level 1: the boundary values; level kb has none -- and uf, vf of level kb stay what they are)
    {
      const unsigned ov = (k <= kbm1) ? oc : BOFF_NONE;
#ifndef POMGPU_EMU
      const bool inl = k <= KL;                             // wave-uniform
      if (inl) { evec[k - 1][ty][tx] = e1p; evec[KL + k - 1][ty][tx] = e2p; }
      const unsigned oe_ = inl ? BOFF_NONE : ov;
#else
      const unsigned oe_ = ov;
#endif
      bst(be1, oe_, lv, e1p); bst(buf, ov, lv, g1p); bst(be2, oe_, lv, e2p); bst(bvf, ov, lv, g2p);
    }
    // ---- new mixing coefficients -- :1484-1503, cosmetics + mask :1510-1535
    double kq_n = 0., km_n = 0., kh_n = 0.;
    {
      const double sh = coef1 / (1. - coef2 * gh);
      double sm = coef3 + sh * coef4 * gh;
      sm = sm / (1. - coef5 * gh);
      const double pl = l * sqrt(fabs(cur.q2));
      kq_n = (pl * .41 * sh + kqc) * .5;
      km_n = (pl * sm + cur.km) * .5;
      kh_n = (pl * sh + cur.kh) * .5;
      // own cell: in place (old kq lives on in kqm / kqc / kq1)
      bst(bkq, o_k, lv, kq_n * fsm_c);
      bst(bkm, o_k, lv, km_n * fsm_c);
      bst(bkh, o_k, lv, kh_n * fsm_c);
    }
    // Physical-edge cells that copy this column are written to the staging arrays s3[1..3] (their own threads still
    // read the old kq) and moved by k_profq_rim.  Stores only, in the few columns next to such an edge.
    if ((ti | tj) && !repl) {
#define PUT(ii, jj, m)                               \
  {                                                  \
    G3(P.s3[1], ii, jj, k) = km_n * m;               \
    G3(P.s3[2], ii, jj, k) = kh_n * m;               \
    G3(P.s3[3], ii, jj, k) = kq_n * m;               \
  }
      if (ti) PUT(ti, j, m_ti)
      if (tj) PUT(i, tj, m_tj)
      if (ti && tj) PUT(ti, tj, m_tij)
#undef PUT
    }
    ccm = cck; rhom = rhok;
    ucm = uck; uem = uek; vcm = vck; vnm = vnk;
    kqm = kqc; kqc = kqp;
  };
  {
    LevQ ra, rb;
    lev(ra, 1);
    rb = ra;
    for (int k = 1; k <= kb; k += 2) {
      if (pace) PACE_BARRIER();
      step(k, ra, rb);
      if (k + 1 <= kb) step(k + 1, rb, ra);
    }
  }
#else
  // The walk down, PEELED BY PHASE.  The level loop above (kept under PROFQ_OLDWALK for A/B runs) asked "which level is this?" a
  // dozen times per level -- k == 1, k == 2, 2 <= k <= kbm1, k == kbm1, k <= KL, z(k) > -0.5 ... -- all wave-uniform, i.e. a
  // dozen scalar branches that cut every level into small basic blocks.  This kernel is bound by the latency of its dependent
  // fp64 chains (13 divisions and 3 square roots per level at two waves per SIMD: a wave issues an instruction every ~17
  // cycles, profiles/round3_profq_phases.txt), and the scheduler can only overlap chains that sit in ONE block.  Here the
  // phase is a compile-time tag: level 1, level 2, the levels whose elimination vectors go to LDS (3..KL), those that go to
  // memory (KL+1..kbm2), level kbm1, level kb -- inside a phase no question about the level is left, per-lane conditions are
  // selects, and the two loops' bodies are two levels in one block each.  Same operations on the same operands in the same order.
  enum { PH_1, PH_2, PH_LDS, PH_MEM, PH_KBM1, PH_KB };
  auto lev = [&](LevQ &L, int k) {              // k = 1..kb; every operand, every lane
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    L.t = bld(bt, oc, lv); L.s = bld(bs_, oc, lv); L.rho = bld(brho, oc, lv); L.rm = bld(BUF3(A3(rmean)), o_rm, lv);
    L.q2b = bld(bq2b, oc, lv); L.q2lb = bld(bq2lb, oc, lv); L.q2 = bld(bq2, oc, lv);
    L.km = bld(bkm, oc, lv); L.kh = bld(bkh, oc, lv);
    L.uf = bld(buf, oc, lv); L.vf = bld(bvf, oc, lv);
    if (FP) { L.uc = bld(bu, oc, lv); L.ue = bld(bu, oe, lv); L.vc = bld(bv, oc, lv); L.vn = bld(bv, on, lv); }
    else L.uc = L.ue = L.vc = L.vn = 0.;
    L.kq1 = bld(bkq, oc, (unsigned)WAVE_UNIFORM(k < kb ? k : kb - 1) * lvb);     // OLD kq of level k+1 (overwritten two iterations from now)
    L.prod = (FP != 1) ? bld(bpr, o_pr, lv) : 0.;
  };
  auto step = [&](auto ph, const int k, const LevQ &cur, LevQ &nxt) {
    constexpr int PH = decltype(ph)::value;
    constexpr bool mid = (PH != PH_1 && PH != PH_KB);       // 2 <= k <= kbm1
    if (PH != PH_KB) lev(nxt, k + 1);                       // in flight during this iteration
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    const double kqp = cur.kq1;
    // ---- level-local quantities
    double cck = 0., rhok = 0.;
    if (PH != PH_KB) { cck = profq_cc_v(P, cur.t, cur.s, hij, k); rhok = rho_rt ? (cur.rho - cur.rm) + cur.rm : cur.rho; }
    double q2b = cur.q2b;
    double l, gh = 0.;
    double uck = 0., uek = 0., vck = 0., vnk = 0., bg = 0.;
    if (PH != PH_KB) { uck = pin ? cur.uc : 0.; uek = pin ? cur.ue : 0.; vck = pin ? cur.vc : 0.; vnk = pin ? cur.vn : 0.; }
    double q2lb = 0.;
    if (mid) {
      q2b = fabs(q2b);                                                                      // :1325-1326
      q2lb = fabs(cur.q2lb);
      bg = P.grav * (rhom - rhok) / (F1(dzz, k - 1) * hij) + sq(P.grav) * 2. / (sq(ccm) + sq(cck));   // :1327-1330
      l = fabs(q2lb / q2b);                                                                 // :1338-1344
      const double lmx = fmax(l, P.kappa * l0);
      l = (F1(z, k) > -0.5) ? lmx : l;
      gh = fmin(sq(l) * bg / q2b, .028);
      bst(bq2b, o_abs, lv, q2b);
      bst(bq2lb, o_abs, lv, q2lb);
    } else {
      l = (PH == PH_1) ? P.kappa * l0 : 0.;                                                 // :1351-1354
    }
    bst(bl, oc, lv, l);
    const double dtef1 = sqrt(fabs(q2b)) * 1. / (b1 * l + P.small_);                        // :1388-1389
    double dtef2 = dtef1;
    if (mid) dtef2 = dtef1 * (1. + e2 * sq(divi((1. / fabs(F1(z, k) - z1) + 1. / fabs(F1(z, k) - zkb)) * l, dhk)));   // :1429-1432
    bst(bdt, oc, lv, dtef2);
    // ---- both forward eliminations -- :1394-1404, :1436-1446
    if (mid) {
      const double a = -P.dti2 * (kqp + kqc + umol2) * .5 / (F1(dzz, k - 1) * F1(dz, k) * dh * dh);      // :1261-1264
      const double c = -P.dti2 * (kqm + kqc + umol2) * .5 / (F1(dzz, k - 1) * F1(dz, k - 1) * dh * dh);
      double pr;
      if (FP == 0) {
        pr = cur.prod;
      } else {                                                                              // :1359-1373, as k_profq_prod; every lane
        const double sef = 1., shiw = 0.;                                                   // evaluates it, the lanes off the interior
        const double km = cur.km;                                                           // select their own value afterwards
        pr = km * .25 * sef * (sq(uck - ucm + uek - uem) + sq(vck - vcm + vnk - vnm)) / sq(F1(dzz, k - 1) * dh) - shiw * km * bg;
        pr = pr + cur.kh * bg;
        pr = pin ? pr : (FP == 2 ? cur.prod : 0.);
      }
      const double g = 1. / (a + c * (1. - e1p) - (2. * P.dti2 * dtef1 + 1.));
      e1p = a * g;
      g1p = (-2. * P.dti2 * pr + c * g1p - cur.uf) * g;
      if (PH == PH_2) {
        e2p = 0.;
        g2p = -P.kappa * F1(z, 2) * dh * cur.q2;
      } else {
        const double rhs = (PH == PH_KBM1) ? vbot : cur.vf;
        const double g2 = 1. / (a + c * (1. - e2p) - (P.dti2 * dtef2 + 1.));
        e2p = a * g2;
        g2p = (P.dti2 * (-pr * l * e1) + c * g2p - rhs) * g2;
      }
    }
    // the vectors' level k (level 1: the boundary values; level kb has none -- and uf, vf of level kb stay what they are)
    if (PH != PH_KB) {
#ifndef POMGPU_EMU
      if (PH == PH_1 || PH == PH_2 || PH == PH_LDS) {       // k <= KL (KL >= 9)
        evec[k - 1][ty][tx] = e1p; evec[KL + k - 1][ty][tx] = e2p;
      } else if (PH == PH_MEM) {
        bst(be1, oc, lv, e1p); bst(be2, oc, lv, e2p);
      } else {                                              // level kbm1: either side of KL
        const bool inl = k <= KL;                           // wave-uniform
        if (inl) { evec[k - 1][ty][tx] = e1p; evec[KL + k - 1][ty][tx] = e2p; }
        const unsigned oe_ = inl ? BOFF_NONE : oc;
        bst(be1, oe_, lv, e1p); bst(be2, oe_, lv, e2p);
      }
#else
      bst(be1, oc, lv, e1p); bst(be2, oc, lv, e2p);
#endif
      bst(buf, oc, lv, g1p); bst(bvf, oc, lv, g2p);
    }
    // ---- new mixing coefficients -- :1484-1503, cosmetics + mask :1510-1535
    double kq_n = 0., km_n = 0., kh_n = 0.;
    {
      const double sh = coef1 / (1. - coef2 * gh);
      double sm = coef3 + sh * coef4 * gh;
      sm = sm / (1. - coef5 * gh);
      const double pl = l * sqrt(fabs(cur.q2));
      kq_n = (pl * .41 * sh + kqc) * .5;
      km_n = (pl * sm + cur.km) * .5;
      kh_n = (pl * sh + cur.kh) * .5;
      // own cell: in place (old kq lives on in kqm / kqc / kq1)
      bst(bkq, o_k, lv, kq_n * fsm_c);
      bst(bkm, o_k, lv, km_n * fsm_c);
      bst(bkh, o_k, lv, kh_n * fsm_c);
    }
    // Physical-edge cells that copy this column are written to the staging arrays s3[1..3] (their own threads still
    // read the old kq) and moved by k_profq_rim.  Stores only, in the few columns next to such an edge.
    if ((ti | tj) && !repl) {
#define PUT(ii, jj, m)                               \
  {                                                  \
    G3(P.s3[1], ii, jj, k) = km_n * m;               \
    G3(P.s3[2], ii, jj, k) = kh_n * m;               \
    G3(P.s3[3], ii, jj, k) = kq_n * m;               \
  }
      if (ti) PUT(ti, j, m_ti)
      if (tj) PUT(i, tj, m_tj)
      if (ti && tj) PUT(ti, tj, m_tij)
#undef PUT
    }
    ccm = cck; rhom = rhok;
    ucm = uck; uem = uek; vcm = vck; vnm = vnk;
    kqm = kqc; kqc = kqp;
  };
  {
    using std::integral_constant;
    LevQ ra, rb;
    lev(ra, 1);
    if (pace) PACE_BARRIER();
    step(integral_constant<int, PH_1>(), 1, ra, rb);
    step(integral_constant<int, PH_2>(), 2, rb, ra);          // kb >= 4 (pomgpu_create): level 2 is never level kbm1
    int k = 3;
    const int kbm2 = kb - 2;
#ifndef POMGPU_EMU
    const int kA = KL < kbm2 ? KL : kbm2;                   // the levels whose ee1, ee2 wait in LDS
#else
    const int kA = 2;
#endif
    for (; k + 1 <= kA; k += 2) {
      if (pace) PACE_BARRIER();
      step(integral_constant<int, PH_LDS>(), k, ra, rb);
      step(integral_constant<int, PH_LDS>(), k + 1, rb, ra);
    }
    if (k <= kA) { step(integral_constant<int, PH_LDS>(), k, ra, rb); ra = rb; k++; }         // an odd level: back to the first register set
    for (; k + 1 <= kbm2; k += 2) {
      if (pace) PACE_BARRIER();
      step(integral_constant<int, PH_MEM>(), k, ra, rb);
      step(integral_constant<int, PH_MEM>(), k + 1, rb, ra);
    }
    if (k <= kbm2) { step(integral_constant<int, PH_MEM>(), k, ra, rb); ra = rb; k++; }
    if (pace) PACE_BARRIER();
    step(integral_constant<int, PH_KBM1>(), kbm1, ra, rb);
    step(integral_constant<int, PH_KB>(), kb, rb, ra);
  }
#endif
  // ---- back substitution -- :1406-1413, :1448-1455, abs :1467-1468; with the fused filter one level of k_q_filter rides
  // on every level: bcond(6)'s mask, Asselin filter, rotation; q2b, q2lb as the walk down leaves them (|.| at 2..kbm1)
  const double hs = .5 * P.smoth;
  auto levb = [&](LevB &L, int k) {             // k = 1..kb
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
#ifndef POMGPU_EMU
    const bool inl = k <= KL;
    const unsigned oe_ = inl ? BOFF_NONE : oc;              // outside the buffer: no traffic, the value comes from LDS
#else
    const unsigned oe_ = oc;
#endif
    L.e1 = bld(be1, oe_, lv); L.g1 = bld(buf, oc, lv); L.e2 = bld(be2, oe_, lv); L.g2 = bld(bvf, oc, lv);
#ifndef POMGPU_EMU
    if (inl) { L.e1 = evec[k - 1][ty][tx]; L.e2 = evec[KL + k - 1][ty][tx]; }
#endif
    if (FF) { L.qb = bld(bq2b, oc, lv); L.qlb = bld(bq2lb, oc, lv); L.q = bld(bq2, oc, lv); L.ql = bld(bq2l, oc, lv); }
    else L.qb = L.qlb = L.q = L.ql = 0.;
  };
  auto filt = [&](int k, const LevB &L, double ufv, double vfv) {
    const double ufm = ufv * fsm_c + 1.e-10, vfm = vfv * fsm_c + 1.e-10;
    double qb = L.qb, qlb = L.qlb;
    if (k >= 2 && k <= kbm1) { qb = fabs(qb); qlb = fabs(qlb); }
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    bst(bq2b, o_fil, lv, L.q + hs * (ufm + qb - 2. * L.q));
    bst(bq2, o_fil, lv, ufm);
    bst(bq2lb, o_fil, lv, L.ql + hs * (vfm + qlb - 2. * L.ql));
    bst(bq2l, o_fil, lv, vfm);
    return ufm;
  };
  double x = ufbot, y = 0.;
  auto stepb = [&](const int ki, const LevB &cur, LevB &nxt) {   // ki = kb (filter only) .. 1
    levb(nxt, ki > 1 ? ki - 1 : 1);
    const unsigned lv = (unsigned)WAVE_UNIFORM(ki - 1) * lvb;
    if (ki == kb) {
      if (FF) {
        const double ufm = ufbot * fsm_c + 1.e-10, vfm = 0. * fsm_c + 1.e-10;
        filt(kb, cur, ufbot, 0.);
        bst(buf, o_fil, lv, ufm);                           // advt leaves level kb of uf, vf as it finds it
        bst(bvf, o_fil, lv, vfm);
      }
      bst(bvf, o_uv, lv, 0.);                               // without the filter: vf(kb) = 0
      return;
    }
    x = cur.e1 * x + cur.g1;
    double yv = 0.;
    if (ki >= 2) {
      y = cur.e2 * y + cur.g2;
      yv = fabs(y);
    }
    const double xv = (ki >= 2) ? fabs(x) : x;
    if (FF) filt(ki, cur, xv, yv);
    bst(buf, o_uv, lv, xv);
    bst(bvf, o_uv, lv, yv);                                 // level 1: 0
  };
  {
    LevB ra, rb;
    levb(ra, kb);
    rb = ra;
    for (int ki = kb; ki >= 1; ki -= 2) {
      if (pace) PACE_BARRIER();
      stepb(ki, ra, rb);
      if (ki - 1 >= 1) stepb(ki - 1, rb, ra);
    }
  }
}

#undef bt
#undef bs_
#undef brho
#undef bq2b
#undef bq2lb
#undef bq2
#undef bq2l
#undef bkm
#undef bkh
#undef bkq
#undef buf
#undef bvf
#undef bu
#undef bv
#undef bl
#undef bdt
#undef bpr
#undef be1
#undef be2
// (4) move the staged km/kh/kq of the physical-edge cells into place (solver.f:1510-1529)
__global__ void k_profq_rim(KP P) {
  const int t = TID_I, line = (int)blockIdx.y, k = TID_K;
  if (k > P.kb) return;
  int i, j;
  if (line == 0) { if (!P.W || t > P.jm) return; i = 1; j = t; }
  else if (line == 1) { if (!P.E || t > P.jm) return; i = P.im; j = t; }
  else if (line == 2) { if (!P.S || t > P.im) return; i = t; j = 1; }
  else { if (!P.N || t > P.im) return; i = t; j = P.jm; }
  F3(km, i, j, k) = G3(P.s3[1], i, j, k);
  F3(kh, i, j, k) = G3(P.s3[2], i, j, k);
  F3(kq, i, j, k) = G3(P.s3[3], i, j, k);
}

// ---------------------------------------------------------------------------------------------
// proft -- solver.f:1541-1683
__global__ void k_proft(KP P, double *f, const double *wfsurf, const double *fsurf, int nbc) {
  COL2
  if (i > P.im || j > P.jm) return;
  const double r_[5] = {.58, .62, .67, .77, .78}, ad1_[5] = {.35, .60, 1.0, 1.5, 1.4}, ad2_[5] = {23., 20., 17., 14., 7.9};
  double ee[POMGPU_KBMAX], gg[POMGPU_KBMAX];
  const int kbm1 = P.kbm1, kbm2 = P.kbm2;
  const double dh = h_(i, j) + F2(etf, i, j);
  const bool sw_ = (nbc == 2 || nbc == 4);
  const double r = r_[P.ntp - 1], ad1 = ad1_[P.ntp - 1], ad2 = ad2_[P.ntp - 1];
  const double swr = sw_ ? F2(swrad, i, j) : 0.;
  // penetrative radiation at level k (0 at kb); the reference evaluates the expression in REAL(16) and rounds once (:1608-1611):
  // proft_rad_q (dd_exp.h) returns that double from double-double arithmetic
#define RAD(k) ((sw_ && (k) <= kbm1) ? proft_rad_q(swr, r, 1. - r, F1(z, k) * dh / ad1, F1(z, k) * dh / ad2) : 0.)
#define ACOEF(k) (((k) <= kbm2) ? -P.dti2 * (F3(kh, i, j, (k) + 1) + P.umol) / (F1(dz, k) * F1(dzz, k) * dh * dh) : 0.)
#define CCOEF(k) (-P.dti2 * (F3(kh, i, j, k) + P.umol) / (F1(dz, k) * F1(dzz, (k)-1) * dh * dh))
  const double a1 = ACOEF(1);
  double radk = RAD(1);
  if (nbc == 1) {
    ee[0] = a1 / (a1 - 1.);
    double g = P.dti2 * G2(wfsurf, i, j) / (F1(dz, 1) * dh) - G3(f, i, j, 1);
    gg[0] = g / (a1 - 1.);
  } else if (nbc == 2) {
    ee[0] = a1 / (a1 - 1.);
    double g = P.dti2 * (G2(wfsurf, i, j) + radk - RAD(2)) / (F1(dz, 1) * dh) - G3(f, i, j, 1);
    gg[0] = g / (a1 - 1.);
  } else {
    ee[0] = 0.;
    gg[0] = G2(fsurf, i, j);
  }
  radk = RAD(2);
  for (int k = 2; k <= kbm2; k++) {
    const double a = ACOEF(k), c = CCOEF(k);
    const double radn = RAD(k + 1);
    const double g = 1. / (a + c * (1. - ee[k - 2]) - 1.);
    ee[k - 1] = a * g;
    gg[k - 1] = (c * gg[k - 2] - G3(f, i, j, k) + P.dti2 * (radk - radn) / (dh * F1(dz, k))) * g;
    radk = radn;
  }
  {
    const double c = CCOEF(kbm1);
    // radk == rad(kbm1) here (also when the loop above did not run: kbm2 < 2 is not supported)
    double x = (c * gg[kbm2 - 1] - G3(f, i, j, kbm1) + P.dti2 * (radk - 0.) / (dh * F1(dz, kbm1))) /
               (c * (1. - ee[kbm2 - 1]) - 1.);
    G3(f, i, j, kbm1) = x;
    for (int ki = kbm2; ki >= 1; ki--) {
      x = (ee[ki - 1] * x + gg[ki - 1]);
      G3(f, i, j, ki) = x;
    }
  }
#undef RAD
#undef ACOEF
#undef CCOEF
}

// proft with the elimination vectors in REGISTERS.  The private ee/gg arrays of k_proft live in
// scratch memory: written on the way down, read on the way up, i.e. four extra array passes through
// L2/HBM per solve (PMC: 7.2 passes per launch for 3 algorithmic).  The VGPR file is the largest
// on-chip memory of a CU (512 KB): with the level loops fully unrolled (trip count = template KBT,
// the smallest instantiated bound >= kb) every ee[k]/gg[k] has a compile-time index and is a
// register.
//   * Phase A issues EVERY load of the column up front, into the registers that will later hold the
//     elimination vectors: f(k) sits in gg[k-1] until level k turns it into gg(k-1), kh(k) sits in
//     ee[k-1] until level k has used it as its c-coefficient and writes ee(k-1) there.  No extra
//     registers, ~2*kb loads in flight per lane, and a fence keeps the scheduler from sinking them
//     to their uses (it does: one load in flight, 560 ns per level, measured).
//   * Levels past kb are computed on clamped operands and discarded by selects -- no branch inside
//     the sweeps.
// SW = 1 compiles the short-wave penetration terms (nbc 2 or 4) in.  Arithmetic is the same
// expression sequence as k_proft.  When SW = 0 the term dti2*(rad(k)-rad(k+1))/(dh*dz(k)) is +0
// exactly (dti2 > 0, dh*dz > 0), written as "+ 0.".
template <int KBT, int SW>
static __device__ __forceinline__ void d_proft_reg(const KP &P, double *f, const double *wfsurf, const double *fsurf, int nbc) {
  COL2
  if (i > P.im || j > P.jm) return;
  const double r_[5] = {.58, .62, .67, .77, .78}, ad1_[5] = {.35, .60, 1.0, 1.5, 1.4}, ad2_[5] = {23., 20., 17., 14., 7.9};
  double ee[KBT], gg[KBT];
  const int kbm1 = P.kbm1, kbm2 = P.kbm2;
#define KC(k) ((k) < kbm1 ? (k) : kbm1)                       /* clamp a level index into the column */
  // ---- phase A: all loads
#pragma unroll
  for (int k = 2; k <= KBT - 1; k++) ee[k - 1] = F3(kh, i, j, KC(k));
#pragma unroll
  for (int k = 2; k <= KBT - 2; k++) gg[k - 1] = G3(f, i, j, KC(k));
  const double f_1 = G3(f, i, j, 1), f_kbm1 = G3(f, i, j, kbm1);
  const double dh = h_(i, j) + F2(etf, i, j);
  const double swr = SW ? F2(swrad, i, j) : 0.;
  const double wfs = G2(wfsurf, i, j), fs = G2(fsurf, i, j);
  SCHED_FENCE();
  // ---- phase B: forward elimination
  const double r = r_[P.ntp - 1], ad1 = ad1_[P.ntp - 1], ad2 = ad2_[P.ntp - 1];
#define RAD(k) ((SW && (k) <= kbm1) ? proft_rad_q(swr, r, 1. - r, F1(z, KC(k)) * dh / ad1, F1(z, KC(k)) * dh / ad2) : 0.)
#define ACOEF(k, khn) (-P.dti2 * ((khn) + P.umol) / (F1(dz, KC(k)) * F1(dzz, KC(k)) * dh * dh))       /* khn = kh(k+1) */
#define CCOEF(k, khk) (-P.dti2 * ((khk) + P.umol) / (F1(dz, KC(k)) * F1(dzz, KC((k)-1)) * dh * dh))   /* khk = kh(k)   */
  const double a1 = ACOEF(1, ee[1]);
  double radk = RAD(1);
  if (nbc == 1) {
    ee[0] = a1 / (a1 - 1.);
    double g = P.dti2 * wfs / (F1(dz, 1) * dh) - f_1;
    gg[0] = g / (a1 - 1.);
  } else if (nbc == 2) {
    ee[0] = a1 / (a1 - 1.);
    double g = P.dti2 * (wfs + radk - RAD(2)) / (F1(dz, 1) * dh) - f_1;
    gg[0] = g / (a1 - 1.);
  } else {
    ee[0] = 0.;
    gg[0] = fs;
  }
  radk = RAD(2);
  double e_last = ee[0], g_last = gg[0], rad_last = radk, kh_last = ee[1];   // ee, gg of level kbm2; rad, kh of level kbm1
#pragma unroll
  for (int k = 2; k <= KBT - 2; k++) {
    const double a = ACOEF(k, ee[k]), c = CCOEF(k, ee[k - 1]);
    const bool fin = (k == kbm2);
    kh_last = fin ? ee[k] : kh_last;
    const double radn = RAD(k + 1);
    const double g = 1. / (a + c * (1. - ee[k - 2]) - 1.);
    ee[k - 1] = a * g;
    if (SW) gg[k - 1] = (c * gg[k - 2] - gg[k - 1] + P.dti2 * (radk - radn) / (dh * F1(dz, KC(k)))) * g;
    else gg[k - 1] = (c * gg[k - 2] - gg[k - 1] + 0.) * g;
    radk = radn;
    e_last = fin ? ee[k - 1] : e_last;
    g_last = fin ? gg[k - 1] : g_last;
    rad_last = fin ? radk : rad_last;
  }
  // ---- phase C: bottom value and back substitution
  {
    const double c = CCOEF(kbm1, kh_last);
    double x;
    if (SW) x = (c * g_last - f_kbm1 + P.dti2 * (rad_last - 0.) / (dh * F1(dz, kbm1))) / (c * (1. - e_last) - 1.);
    else x = (c * g_last - f_kbm1 + 0.) / (c * (1. - e_last) - 1.);
    G3(f, i, j, kbm1) = x;
#pragma unroll
    for (int ki = KBT - 2; ki >= 1; ki--) {
      const double xn = (ee[ki - 1] * x + gg[ki - 1]);
      if (ki <= kbm2) { x = xn; G3(f, i, j, ki) = x; }
    }
  }
#undef KC
#undef RAD
#undef ACOEF
#undef CCOEF
}
template <int KBT, int SW>
__global__ void __launch_bounds__(64 * ROWS_PROFT) k_proft_reg(KP P, double *f, const double *wfsurf, const double *fsurf, int nbc) {
  d_proft_reg<KBT, SW>(P, f, wfsurf, fsurf, nbc);
}
// T and S (advance.f:439-440: two calls that share nothing but kh) as ONE grid, the tracer on blockIdx.z: on a tile of an 8-GPU split a
// launch is three to four rounds of workgroups whose last round is mostly empty -- one launch pays for that once, not twice
struct ProftArgs { double *f; const double *wfsurf, *fsurf; int nbc; };
template <int KBT, int SW>
__global__ void __launch_bounds__(64 * ROWS_PROFT) k_proft_reg2(KP P, ProftArgs a0, ProftArgs a1) {
  if (blockIdx.z == 0) d_proft_reg<KBT, SW>(P, a0.f, a0.wfsurf, a0.fsurf, a0.nbc);
  else d_proft_reg<KBT, SW>(P, a1.f, a1.wfsurf, a1.fsurf, a1.nbc);
}

// ---------------------------------------------------------------------------------------------
// advu + profu -- solver.f:734-788, :1686-1780.   do_adv / do_prof select the halves so that the
// stand-alone entry points exist; with both set the right-hand side never leaves the thread.
__global__ void k_advu_profu(KP P, int do_adv, int do_prof) {
  COL2
  if (i > P.im || j > P.jm) return;
  double ee[POMGPU_KBMAX], gg[POMGPU_KBMAX];
  const int kb = P.kb, kbm1 = P.kbm1, kbm2 = P.kbm2;
  const bool in = (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1);
  // ---- columns outside the interior: the reference leaves the vertical-flux values of advu there
  if (!in) {
    if (!do_adv) return;
    double fk = 0.;
    F3(uf, i, j, 1) = 0.;
    for (int k = 2; k <= kbm1; k++) {
      fk = (i >= 2) ? .25 * (w_(i, j, k) + w_(i - 1, j, k)) * (u_(i, j, k) + u_(i, j, k - 1)) : 0.;   // :744-751
      F3(uf, i, j, k) = fk;
    }
    F3(uf, i, j, kb) = 0.;
    return;
  }
  // ---- interior column
  const double ar = F2(aru, i, j);
  const double hc = P.grav * .125 * (dt_(i, j) + dt_(i - 1, j)) *
                    (F2(egf, i, j) - F2(egf, i - 1, j) + F2(egb, i, j) - F2(egb, i - 1, j) + (F2(e_atmos, i, j) - F2(e_atmos, i - 1, j)) * 2.) *
                    (dy_(i, j) + dy_(i - 1, j));
  const double step_a = (h_(i, j) + F2(etb, i, j) + h_(i - 1, j) + F2(etb, i - 1, j)) * ar;
  const double step_d = (h_(i, j) + F2(etf, i, j) + h_(i - 1, j) + F2(etf, i - 1, j)) * ar;
  // right-hand side of level k = the leapfrog step of advu (:758-782); fk/fn: vertical flux at levels k, k+1
#define RHS(k, fk, fn)                                                                                   \
  ((step_a * F3(ub, i, j, k) -                                                                            \
    2. * P.dti2 * (F3(advx, i, j, k) + ((fk) - (fn)) * ar / F1(dz, k) -                                   \
                   ar * .25 * (F2(cor, i, j) * dt_(i, j) * (v_(i, j + 1, k) + v_(i, j, k)) +                   \
                               F2(cor, i - 1, j) * dt_(i - 1, j) * (v_(i - 1, j + 1, k) + v_(i - 1, j, k))) + \
                   hc + F3(drhox, i, j, k))) /                                                            \
   step_d)
#define VFLUX(k) (.25 * (w_(i, j, k) + w_(i - 1, j, k)) * (u_(i, j, k) + u_(i, j, (k)-1)))
  if (do_adv && !do_prof) {                                // advu alone
    double fk = 0.;
    for (int k = 1; k <= kbm1; k++) {
      const double fn = (k + 1 <= kbm1) ? VFLUX(k + 1) : 0.;
      F3(uf, i, j, k) = RHS(k, fk, fn);
      fk = fn;
    }
    F3(uf, i, j, kb) = 0.;
    return;
  }
  // profu: forward sweep; with do_adv the right-hand side is formed on the fly and never stored
  const double dh = (h_(i, j) + F2(etf, i, j) + h_(i - 1, j) + F2(etf, i - 1, j)) * .5;
#define KMA(k) ((F3(km, i, j, k) + F3(km, i - 1, j, k)) * .5)
  double fk = 0.;
  double km_c = KMA(1), km_n = KMA(2);                      // averaged km at levels k and k+1
  double rhs_k = 0., g_prev = 0., e_prev = 0.;
  for (int k = 1; k <= kbm1; k++) {
    if (do_adv) {
      const double fn = (k + 1 <= kbm1) ? VFLUX(k + 1) : 0.;
      rhs_k = RHS(k, fk, fn);
      fk = fn;
    } else {
      rhs_k = F3(uf, i, j, k);
    }
    // a(k) = -dti2*(c(k+1)+umol)/(dz(k)*dzz(k)*dh*dh), c(k) = -dti2*(c(k)+umol)/(dz(k)*dzz(k-1)*dh*dh)   (:1712-1729)
    const double a = (k <= kbm2) ? -P.dti2 * (km_n + P.umol) / (F1(dz, k) * F1(dzz, k) * dh * dh) : 0.;
    if (k == 1) {
      e_prev = a / (a - 1.);
      g_prev = (-P.dti2 * F2(wusurf, i, j) / (-F1(dz, 1) * dh) - rhs_k) / (a - 1.);
      ee[0] = e_prev; gg[0] = g_prev;
    } else if (k <= kbm2) {
      const double c = -P.dti2 * (km_c + P.umol) / (F1(dz, k) * F1(dzz, k - 1) * dh * dh);
      const double g = 1. / (a + c * (1. - e_prev) - 1.);
      e_prev = a * g;
      g_prev = (c * g_prev - rhs_k) * g;
      ee[k - 1] = e_prev; gg[k - 1] = g_prev;
    }
    km_c = km_n;
    if (k + 2 <= kb) km_n = KMA(k + 2);
  }
  // here km_c = averaged km at kb, the loop's last c-level was kbm1: recompute c(kbm1) from KMA(kbm1)
  const double tps = 0.5 * (F2(cbc, i, j) + F2(cbc, i - 1, j)) *
                     sqrt(sq(F3(ub, i, j, kbm1)) +
                          sq(.25 * (F3(vb, i, j, kbm1) + F3(vb, i, j + 1, kbm1) + F3(vb, i - 1, j, kbm1) + F3(vb, i - 1, j + 1, kbm1))));
  const double c = -P.dti2 * (KMA(kbm1) + P.umol) / (F1(dz, kbm1) * F1(dzz, kbm2) * dh * dh);
  const double m = F2(dum, i, j);
  double x = (c * gg[kbm2 - 1] - rhs_k) / (tps * P.dti2 / (-F1(dz, kbm1) * dh) - 1. - (ee[kbm2 - 1] - 1.) * c);
  x = x * m;
  F3(uf, i, j, kbm1) = x;
  F2(wubot, i, j) = -tps * x;
  for (int ki = kbm2; ki >= 1; ki--) {
    x = (ee[ki - 1] * x + gg[ki - 1]) * m;
    F3(uf, i, j, ki) = x;
  }
  if (do_adv) F3(uf, i, j, kb) = 0.;
  F2(tps, i, j) = tps;
#undef RHS
#undef VFLUX
#undef KMA
}

// advv + profv -- solver.f:791-845, :1783-1877
__global__ void k_advv_profv(KP P, int do_adv, int do_prof) {
  COL2
  if (i > P.im || j > P.jm) return;
  double ee[POMGPU_KBMAX], gg[POMGPU_KBMAX];
  const int kb = P.kb, kbm1 = P.kbm1, kbm2 = P.kbm2;
  const bool in = (i >= 2 && i <= P.imm1 && j >= 2 && j <= P.jmm1);
  // ---- columns outside the interior: the reference leaves the vertical-flux values of advv there
  if (!in) {
    if (!do_adv) return;
    double fk = 0.;
    F3(vf, i, j, 1) = 0.;
    for (int k = 2; k <= kbm1; k++) {
      fk = (j >= 2) ? .25 * (w_(i, j, k) + w_(i, j - 1, k)) * (v_(i, j, k) + v_(i, j, k - 1)) : 0.;   // :801-808
      F3(vf, i, j, k) = fk;
    }
    F3(vf, i, j, kb) = 0.;
    return;
  }
  // ---- interior column
  const double ar = F2(arv, i, j);
  const double hc = P.grav * .125 * (dt_(i, j) + dt_(i, j - 1)) *
                    (F2(egf, i, j) - F2(egf, i, j - 1) + F2(egb, i, j) - F2(egb, i, j - 1) + (F2(e_atmos, i, j) - F2(e_atmos, i, j - 1)) * 2.) *
                    (dx_(i, j) + dx_(i, j - 1));
  const double step_a = (h_(i, j) + F2(etb, i, j) + h_(i, j - 1) + F2(etb, i, j - 1)) * ar;
  const double step_d = (h_(i, j) + F2(etf, i, j) + h_(i, j - 1) + F2(etf, i, j - 1)) * ar;
  // right-hand side of level k = the leapfrog step of advv (:815-839); fk/fn: vertical flux at levels k, k+1
#define RHS(k, fk, fn)                                                                                   \
  ((step_a * F3(vb, i, j, k) -                                                                            \
    2. * P.dti2 * (F3(advy, i, j, k) + ((fk) - (fn)) * ar / F1(dz, k) +                                   \
                   ar * .25 * (F2(cor, i, j) * dt_(i, j) * (u_(i + 1, j, k) + u_(i, j, k)) +                   \
                               F2(cor, i, j - 1) * dt_(i, j - 1) * (u_(i + 1, j - 1, k) + u_(i, j - 1, k))) + \
                   hc + F3(drhoy, i, j, k))) /                                                            \
   step_d)
#define VFLUX(k) (.25 * (w_(i, j, k) + w_(i, j - 1, k)) * (v_(i, j, k) + v_(i, j, (k)-1)))
  if (do_adv && !do_prof) {                                // advv alone
    double fk = 0.;
    for (int k = 1; k <= kbm1; k++) {
      const double fn = (k + 1 <= kbm1) ? VFLUX(k + 1) : 0.;
      F3(vf, i, j, k) = RHS(k, fk, fn);
      fk = fn;
    }
    F3(vf, i, j, kb) = 0.;
    return;
  }
  // profv: forward sweep; with do_adv the right-hand side is formed on the fly and never stored
  const double dh = .5 * (h_(i, j) + F2(etf, i, j) + h_(i, j - 1) + F2(etf, i, j - 1));
#define KMA(k) ((F3(km, i, j, k) + F3(km, i, j - 1, k)) * .5)
  double fk = 0.;
  double km_c = KMA(1), km_n = KMA(2);                      // averaged km at levels k and k+1
  double rhs_k = 0., g_prev = 0., e_prev = 0.;
  for (int k = 1; k <= kbm1; k++) {
    if (do_adv) {
      const double fn = (k + 1 <= kbm1) ? VFLUX(k + 1) : 0.;
      rhs_k = RHS(k, fk, fn);
      fk = fn;
    } else {
      rhs_k = F3(vf, i, j, k);
    }
    // a(k) = -dti2*(c(k+1)+umol)/(dz(k)*dzz(k)*dh*dh), c(k) = -dti2*(c(k)+umol)/(dz(k)*dzz(k-1)*dh*dh)   (:1810-1827)
    const double a = (k <= kbm2) ? -P.dti2 * (km_n + P.umol) / (F1(dz, k) * F1(dzz, k) * dh * dh) : 0.;
    if (k == 1) {
      e_prev = a / (a - 1.);
      g_prev = (-P.dti2 * F2(wvsurf, i, j) / (-F1(dz, 1) * dh) - rhs_k) / (a - 1.);
      ee[0] = e_prev; gg[0] = g_prev;
    } else if (k <= kbm2) {
      const double c = -P.dti2 * (km_c + P.umol) / (F1(dz, k) * F1(dzz, k - 1) * dh * dh);
      const double g = 1. / (a + c * (1. - e_prev) - 1.);
      e_prev = a * g;
      g_prev = (c * g_prev - rhs_k) * g;
      ee[k - 1] = e_prev; gg[k - 1] = g_prev;
    }
    km_c = km_n;
    if (k + 2 <= kb) km_n = KMA(k + 2);
  }
  // here km_c = averaged km at kb, the loop's last c-level was kbm1: recompute c(kbm1) from KMA(kbm1)
  const double tps = 0.5 * (F2(cbc, i, j) + F2(cbc, i, j - 1)) *
                     sqrt(sq(.25 * (F3(ub, i, j, kbm1) + F3(ub, i + 1, j, kbm1) + F3(ub, i, j - 1, kbm1) + F3(ub, i + 1, j - 1, kbm1))) +
                          sq(F3(vb, i, j, kbm1)));
  const double c = -P.dti2 * (KMA(kbm1) + P.umol) / (F1(dz, kbm1) * F1(dzz, kbm2) * dh * dh);
  const double m = F2(dvm, i, j);
  double x = (c * gg[kbm2 - 1] - rhs_k) / (tps * P.dti2 / (-F1(dz, kbm1) * dh) - 1. - (ee[kbm2 - 1] - 1.) * c);
  x = x * m;
  F3(vf, i, j, kbm1) = x;
  F2(wvbot, i, j) = -tps * x;
  for (int ki = kbm2; ki >= 1; ki--) {
    x = (ee[ki - 1] * x + gg[ki - 1]) * m;
    F3(vf, i, j, ki) = x;
  }
  if (do_adv) F3(vf, i, j, kb) = 0.;
  F2(tps, i, j) = tps;
#undef RHS
#undef VFLUX
#undef KMA
}

// ---------------------------------------------------------------------------------------------
// profu / profv with the elimination vectors in registers (see k_proft_reg): the right-hand side is
// uf / vf as left by k_advuv_col.  V = 0: profu, the km average runs over (i-1,j) and the west value
// is the neighbour lane's (halo-lane wavefronts);  V = 1: profv, (i,j-1) needs a second load per level.
// Phase A loads the whole column into the future ee/gg registers, phase B is the forward
// elimination (solver.f:1712-1745 / :1810-1843), phase C the bottom value and back substitution.
template <int KBT, int V>
static __device__ __forceinline__ void d_profuv_reg(const KP &P, const bool store_tps) {
  const int j = TID_J;
  if (j > P.jm) return;
  int i0, lane = 1;
  if (V) { i0 = TID_I; } else { lane = HALO_LANE; i0 = HALO_COL; }
  const bool out = (lane >= 1 && lane <= 62 && i0 >= 2 && i0 <= P.imm1 && j >= 2 && j <= P.jmm1);
#ifdef POMGPU_EMU
  if (!out) return;
#else
  if (V && !out) return;                                    // profv has no cross-lane traffic
  if (j < 2 || j > P.jmm1) return;                          // wave-uniform
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);
  const int iw = i > 1 ? i - 1 : 1, js = j > 1 ? j - 1 : 1;
  const int in_ = V ? i : iw, jn_ = V ? js : j;            // the second column of the averages
  double *f = P.b3 + (size_t)(V ? P3_vf : P3_uf) * P.a3;
  // per-level accesses as (uniform plane pointer)[32-bit column offset]: the plane pointer is scalar, so
  // the ~2*KBT loads of phase A share ONE offset register instead of holding a 64-bit address each
  // (32-bit BYTE offsets: the form the compiler maps onto the scalar-base + vector-offset addressing mode)
  const unsigned col = POMGPU_ST_BYTES * (unsigned)((j - 1) * P.iml + (i - 1)), col2 = POMGPU_ST_BYTES * (unsigned)((js - 1) * P.iml + (i - 1));
  const double *kmp = P.b3 + (size_t)P3_km * P.a3;
#define PLANE(ptr, k, off) REF3((pomgpu_st *)((char *)((pomgpu_st *)(ptr) + (size_t)((k)-1) * P.n2) + (off)), 0)
  const int kb = P.kb, kbm1 = P.kbm1, kbm2 = P.kbm2;
  double ee[KBT], gg[KBT];
#define KC(k) ((k) < kbm1 ? (k) : kbm1)
#define KCB(k) ((k) < kb ? (k) : kb)
  // ---- phase A: all loads.  km(k) -> ee[k-1]; its partner column -> the neighbour lane (V = 0) or,
  // for V = 1, gg[k-1] for a moment; then ee[k-1] = averaged km (:1705-1710 / :1803-1808) and
  // rhs(k) -> gg[k-1].  Two register arrays in total.
#pragma unroll
  for (int k = 1; k <= KBT; k++) ee[k - 1] = PLANE(kmp, KCB(k), col);
  if (V) {
#pragma unroll
    for (int k = 1; k <= KBT; k++) gg[k - 1] = PLANE(kmp, KCB(k), col2);
  } else {
#pragma unroll
    for (int k = 1; k <= KBT - 1; k++) gg[k - 1] = PLANE(f, KC(k), col);
  }
  const double rhs_b = G3(f, i, j, kbm1);
  const double dh = V ? .5 * (F2(h, i, j) + F2(etf, i, j) + F2(h, i, js) + F2(etf, i, js))
                      : (F2(h, i, j) + F2(etf, i, j) + F2(h, iw, j) + F2(etf, iw, j)) * .5;
  const double wsurf = V ? F2(wvsurf, i, j) : F2(wusurf, i, j);
  const double m = V ? F2(dvm, i, j) : F2(dum, i, j);
  const double cb2 = F2(cbc, i, j) + F2(cbc, in_, jn_);
  double tps;
  if (V) tps = 0.5 * cb2 * sqrt(sq(.25 * (F3(ub, i, j, kbm1) + F3(ub, i + 1, j, kbm1) + F3(ub, i, js, kbm1) + F3(ub, i + 1, js, kbm1))) +
                                sq(F3(vb, i, j, kbm1)));
  else tps = 0.5 * cb2 * sqrt(sq(F3(ub, i, j, kbm1)) +
                              sq(.25 * (F3(vb, i, j, kbm1) + F3(vb, i, j + 1, kbm1) + F3(vb, iw, j, kbm1) + F3(vb, iw, j + 1, kbm1))));
  SCHED_FENCE();
#pragma unroll
  for (int k = 1; k <= KBT; k++) {
    ee[k - 1] = (ee[k - 1] + (V ? gg[k - 1] : halo_w(ee[k - 1], [&] { return F3(km, iw, j, KCB(k)); }))) * .5;
    if (!V && (k & 3) == 0) SCHED_FENCE();                  // or the scheduler forms all lane shifts first: +KBT live doubles
  }
  if (V) {
    SCHED_FENCE();
#pragma unroll
    for (int k = 1; k <= KBT - 1; k++) gg[k - 1] = PLANE(f, KC(k), col);
  }
  SCHED_FENCE();
  // ---- phase B: forward elimination; ee[k-1] holds the averaged km of level k until level k overwrites it
#define KMA(k) ee[(k)-1]
  double km_c = KMA(1), km_n = KMA(2);
  double e_prev, g_prev;
  {
    const double a = -P.dti2 * (km_n + P.umol) / (F1(dz, 1) * F1(dzz, 1) * dh * dh);
    e_prev = a / (a - 1.);
    g_prev = (-P.dti2 * wsurf / (-F1(dz, 1) * dh) - gg[0]) / (a - 1.);
    ee[0] = e_prev; gg[0] = g_prev;
  }
  double e_last = e_prev, g_last = g_prev, km_last = km_n;   // ee, gg of level kbm2 and the averaged km of level kbm1
#pragma unroll
  for (int k = 2; k <= KBT - 2; k++) {
    km_c = km_n;
    km_n = KMA(k + 1);
    const double a = -P.dti2 * (km_n + P.umol) / (F1(dz, KC(k)) * F1(dzz, KC(k)) * dh * dh);
    const double c = -P.dti2 * (km_c + P.umol) / (F1(dz, KC(k)) * F1(dzz, KC(k - 1)) * dh * dh);
    const double g = 1. / (a + c * (1. - e_prev) - 1.);
    e_prev = a * g;
    g_prev = (c * g_prev - gg[k - 1]) * g;
    ee[k - 1] = e_prev; gg[k - 1] = g_prev;
    const bool fin = (k == kbm2);
    e_last = fin ? e_prev : e_last;
    g_last = fin ? g_prev : g_last;
    km_last = fin ? km_n : km_last;
  }
#undef KMA
  if (!out) return;
  // ---- phase C -- :1747-1777 / :1845-1874
  const double c = -P.dti2 * (km_last + P.umol) / (F1(dz, kbm1) * F1(dzz, kbm2) * dh * dh);
  double x = (c * g_last - rhs_b) / (tps * P.dti2 / (-F1(dz, kbm1) * dh) - 1. - (e_last - 1.) * c);
  x = x * m;
  G3(f, i, j, kbm1) = x;
  if (V) F2(wvbot, i, j) = -tps * x; else F2(wubot, i, j) = -tps * x;
#pragma unroll
  for (int ki = KBT - 2; ki >= 1; ki--) {
    const double xn = (ee[ki - 1] * x + gg[ki - 1]) * m;
    if (ki <= kbm2) { x = xn; PLANE(f, ki, col) = x; }
  }
  if (store_tps) F2(tps, i, j) = tps;
#undef KC
#undef KCB
#undef PLANE
}
template <int KBT, int V>
__global__ void __launch_bounds__(64 * ROWS_PROFUV) k_profuv_reg(KP P) { d_profuv_reg<KBT, V>(P, true); }
// profu and profv (advance.f:461-462) as ONE grid, the component on blockIdx.z (see k_proft_reg2); the grid is profu's (62 columns per
// wavefront), profv's workgroups beyond its 64-column count find nothing to do.  tps is the reference's shared scratch:
// only profv, whose value the reference leaves there, stores it.
template <int KBT>
__global__ void __launch_bounds__(64 * ROWS_PROFUV) k_profuv_reg2(KP P) {
  if (blockIdx.z == 0) d_profuv_reg<KBT, 0>(P, false);
  else d_profuv_reg<KBT, 1>(P, true);
}

// ---------------------------------------------------------------------------------------------
// column-mean-free Asselin filter of u,v and time rotation -- advance.f:469-514
// own: leave the ghost lines alone (lines a neighbour tile owns).  There the reference's filter ends in u = uf, v = vf and in ub, vb that the
// exchange of advance.f:516-521 replaces anyway: with the rounds of :466-467 and :516-521 on the side stream (pomgpu_api.hip, "rim rounds")
// the ghost lines of u, v are filled from the message that brings uf, vf, and this kernel need not wait for it.
#define UVF_GHOST(P, i, j) (((i) == 1 && !(P).W) || ((i) == (P).im && !(P).E) || ((j) == 1 && !(P).S) || ((j) == (P).jm && !(P).N))
__global__ void k_uv_filter(KP P, int own) {
  COL2
  if (own && UVF_GHOST(P, i, j)) return;
  const bool act = (i <= P.im && j <= P.jm);
  double su = 0., sv = 0.;
  if (act) {
    for (int k = 1; k <= P.kbm1; k++) {
      const double dzk = F1(dz, k);
      su = su + (F3(uf, i, j, k) + F3(ub, i, j, k) - 2. * F3(u, i, j, k)) * dzk;
      sv = sv + (F3(vf, i, j, k) + F3(vb, i, j, k) - 2. * F3(v, i, j, k)) * dzk;
    }
  }
  for (int k = 1; k <= P.kb; k++) {
    const double uf = F3(uf, i, j, k), vf = F3(vf, i, j, k);
    double u = F3(u, i, j, k), v = F3(v, i, j, k);
    if (act && k <= P.kbm1) {
      u = u + .5 * P.smoth * (uf + F3(ub, i, j, k) - 2. * u - su);
      v = v + .5 * P.smoth * (vf + F3(vb, i, j, k) - 2. * v - sv);
    }
    F3(ub, i, j, k) = u;
    F3(u, i, j, k) = uf;
    F3(vb, i, j, k) = v;
    F3(v, i, j, k) = vf;
  }
  F2(tps, i, j) = sv;
}

// int_uvmean with the column in registers (V = 0: u, V = 1: v): the scratch-free kernel above reads the
// column once for the depth mean and again to correct it (4 reads + 2 writes for both components);
// with the level loop unrolled for a template bound >= kb the column waits in registers: 1 read + 1 write.
template <int KBT, int V>
static __device__ __forceinline__ void d_int_uvmean_reg(const KP &P) {
  COL2
  double *c = P.b3 + (size_t)(V ? P3_v : P3_u) * P.a3;
  const int kbm1 = P.kbm1;
  double x[KBT - 1];
#define KC(k) ((k) < kbm1 ? (k) : kbm1)
#pragma unroll
  for (int k = 1; k <= KBT - 1; k++) x[k - 1] = G3(c, i, j, KC(k));
  const bool upd = V ? (i <= P.im && j >= 2 && j <= P.jm) : (j <= P.jm && i >= 2 && i <= P.im);
  double mean = 0.;
  if (upd) mean = V ? (F2(vtb, i, j) + F2(vtf, i, j)) / (dt_(i, j) + dt_(i, j - 1)) : (F2(utb, i, j) + F2(utf, i, j)) / (dt_(i, j) + dt_(i - 1, j));
  SCHED_FENCE();
  double s_ = 0.;
#pragma unroll
  for (int k = 1; k <= KBT - 1; k++)
    if (k <= kbm1) s_ = s_ + x[k - 1] * F1(dz, k);                                           // advance.f:365-372
  if (upd) {
#pragma unroll
    for (int k = 1; k <= KBT - 1; k++)
      if (k <= kbm1) G3(c, i, j, k) = (x[k - 1] - s_) + mean;                                // :375-392
  }
  if (V) F2(tps, i, j) = s_;
#undef KC
}
template <int KBT, int V>
__global__ void __launch_bounds__(64 * ROWS_UVM) k_int_uvmean_reg(KP P) { d_int_uvmean_reg<KBT, V>(P); }
template <int KBT>
__global__ void __launch_bounds__(64 * ROWS_UVM) k_int_uvmean_reg2(KP P) {   // u and v as ONE grid, the component on blockIdx.z (see k_proft_reg2)
  if (blockIdx.z == 0) d_int_uvmean_reg<KBT, 0>(P);
  else d_int_uvmean_reg<KBT, 1>(P);
}

// uv_filter with the column in registers (one component per launch: V = 0 u, V = 1 v).  The scratch-free
// kernel above sweeps the column twice and re-reads uf, ub, u for the second sweep (16 array passes for
// both components).  Here the first sweep leaves d(k) = uf+ub-2u and the old u(k) in registers
// (compile-time indices, loops unrolled for a template bound >= kb), stores u = uf at once, and the
// second sweep needs no memory: 3 reads + 2 writes per component.  u and ub of the whole column are
// requested up front into their final registers; uf streams through two small chunk buffers.
template <int KBT, int V>
static __device__ __forceinline__ void d_uv_filter_reg(const KP &P, const int own) {
  COL2
  if (i > P.im || j > P.jm) return;
  if (own && UVF_GHOST(P, i, j)) return;
  const double *f = P.b3 + (size_t)(V ? P3_vf : P3_uf) * P.a3;
  double *b = P.b3 + (size_t)(V ? P3_vb : P3_ub) * P.a3, *c = P.b3 + (size_t)(V ? P3_v : P3_u) * P.a3;
  const int kb = P.kb, kbm1 = P.kbm1;
  constexpr int CH = 8, NL = KBT - 1, NCH = (NL + CH - 1) / CH;
  double dk[NL], uo[NL], t0[CH], t1[CH];
#define KC(k) ((k) < kbm1 ? (k) : kbm1)
#pragma unroll
  for (int k = 1; k <= NL; k++) { uo[k - 1] = G3(c, i, j, KC(k)); dk[k - 1] = G3(b, i, j, KC(k)); }
#pragma unroll
  for (int q = 0; q < CH; q++) t0[q] = G3(f, i, j, KC(q + 1));
  const double c_kb = G3(c, i, j, kb), f_kb = G3(f, i, j, kb);
  double su = 0.;
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    if (ch + 1 < NCH) {
#pragma unroll
      for (int q = 0; q < CH; q++) {
        const double x = G3(f, i, j, KC((ch + 1) * CH + q + 1));
        if (ch & 1) t0[q] = x; else t1[q] = x;
      }
    }
    SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const int k = ch * CH + q + 1;
      if (k <= NL) {
        const double uf = (ch & 1) ? t1[q] : t0[q];
        const double d = uf + dk[k - 1] - 2. * uo[k - 1];                                   // advance.f:473-474 / :495-496
        dk[k - 1] = d;
        if (k <= kbm1) {
          su = su + d * F1(dz, k);
          G3(c, i, j, k) = uf;                                                              // :489 / :511
        }
      }
    }
    SCHED_FENCE();
  }
#pragma unroll
  for (int k = 1; k <= NL; k++)
    if (k <= kbm1) G3(b, i, j, k) = uo[k - 1] + .5 * P.smoth * (dk[k - 1] - su);            // :484-488 / :506-510
  G3(b, i, j, kb) = c_kb;
  G3(c, i, j, kb) = f_kb;
  if (V) F2(tps, i, j) = su;
#undef KC
}
template <int KBT, int V>
__global__ void __launch_bounds__(64 * ROWS_UVF) k_uv_filter_reg(KP P, int own) { d_uv_filter_reg<KBT, V>(P, own); }
template <int KBT>
__global__ void __launch_bounds__(64 * ROWS_UVF) k_uv_filter_reg2(KP P, int own) {   // u and v as ONE grid, the component on blockIdx.z (see k_proft_reg2)
  if (blockIdx.z == 0) d_uv_filter_reg<KBT, 0>(P, own);
  else d_uv_filter_reg<KBT, 1>(P, own);
}

// ---- launchers --------------------------------------------------------------------------------
static inline dim3 colblk() { return dim3(64, 2, 1); }
static inline dim3 rowblk(int rows) { return dim3(64, rows, 1); }
static inline dim3 rowgrid(const KP &P, int rows) { return dim3((P.iml + 63) / 64, (P.jml + rows - 1) / rows, 1); }
static inline dim3 colgrid(const KP &P) { return dim3((P.iml + 63) / 64, (P.jml + 1) / 2, 1); }
// baropg, vertvl: four paced rows per workgroup (same-context A/B against 2 unpaced rows: -2..-3 %; 8 rows: +3 % on baropg)
#define COLV_G(P) dim3(((P).iml + 63) / 64, ((P).jml + 3) / 4, 1)
#define COLV_B dim3(64, 4, 1)
void launch_baropg(pomgpu_ctx *c, int sum2d) {
  if (SW(c, BAROPG_CELLS)) { LAUNCH(c, k_baropg, COLV_G(c->P), COLV_B, c->P, sum2d); return; }   // developer switch: one thread per column, six loads per level
  LAUNCHN(c, "k_baropg", k_baropg_rs, grid1_halo_r(c->P, BPG_ROWS), blk_col_r(BPG_ROWS), c->P, sum2d);
}
void launch_baropg_mcc(pomgpu_ctx *c, int sum2d) { LAUNCH(c, k_baropg_mcc, colgrid(c->P), colblk(), c->P, sum2d); }
void launch_order_pack(pomgpu_ctx *c, double *send_e, double *send_n) {
  const KP &P = c->P;
  const int len = P.iml > P.jml ? P.iml : P.jml;
  LAUNCH(c, k_order_pack, dim3((len + 63) / 64, P.kb + 1, 1), dim3(64, 1, 1), c->P, send_e, send_n);
}
static inline dim3 twin(dim3 g) { g.z = 2; return g; }       // the two variants of a twin kernel on blockIdx.z
template <int KBT> static void launch_int_uvmean_reg_t(pomgpu_ctx *c) {
  if (!SW(c, NO_TWIN)) { LAUNCHN(c, "k_int_uvmean_reg2", (k_int_uvmean_reg2<KBT>), twin(rowgrid(c->P, ROWS_UVM)), rowblk(ROWS_UVM), c->P); return; }
  LAUNCHN(c, "k_int_uvmean_reg", (k_int_uvmean_reg<KBT, 0>), rowgrid(c->P, ROWS_UVM), rowblk(ROWS_UVM), c->P);
  LAUNCHN(c, "k_int_uvmean_reg", (k_int_uvmean_reg<KBT, 1>), rowgrid(c->P, ROWS_UVM), rowblk(ROWS_UVM), c->P);
}
void launch_int_uvmean(pomgpu_ctx *c) {
  const int kb = c->P.kb;
  if (SW(c, THOMAS_SCRATCH) || kb > 64 || kb < 6) LAUNCH(c, k_int_uvmean, colgrid(c->P), colblk(), c->P);
  else if (kb <= 24) launch_int_uvmean_reg_t<24>(c);
  else if (kb <= 32) launch_int_uvmean_reg_t<32>(c);
  else if (kb <= 40) launch_int_uvmean_reg_t<40>(c);
  else if (kb <= 44) launch_int_uvmean_reg_t<44>(c);
  else if (kb <= 50) launch_int_uvmean_reg_t<50>(c);
  else if (kb <= 56) launch_int_uvmean_reg_t<56>(c);
  else launch_int_uvmean_reg_t<64>(c);
}
void launch_vertvl(pomgpu_ctx *c, int mask) {
  if (SW(c, VERTVL_CELLS)) { LAUNCH(c, k_vertvl, COLV_G(c->P), COLV_B, c->P, mask); return; }   // developer switch: one thread per column, four loads per level
  LAUNCHN(c, "k_vertvl", k_vertvl_rs, grid1_halo_r(c->P, VVL_ROWS), blk_col_r(VVL_ROWS), c->P, mask);
}
void launch_profq_bc(pomgpu_ctx *c) { LAUNCH(c, k_profq_bc, colgrid(c->P), colblk(), c->P); }
void launch_profq_prod(pomgpu_ctx *c, int lines_only, int rho_rt) {
  if (!lines_only) { LAUNCH(c, k_profq_prod, colgrid(c->P), colblk(), c->P, rho_rt); return; }
  const KP &P = c->P;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_profq_prod_lines, dim3((len + 63) / 64, 8, P.kb), dim3(64, 1, 1), c->P, rho_rt);
}
void launch_profq(pomgpu_ctx *c, int fuse_prod, int fuse_filter, int rho_rt, int jfirst, int jlast) {
  if (jlast < jfirst) { jfirst = 1; jlast = c->P.jml; }      // (default arguments 0, -1: every row)
  const int nrows = jlast - jfirst + 1;
  // 8 paced rows per workgroup (one workgroup per CU at 2 waves per SIMD) where that still leaves every CU three workgroups
  // (a 514x769 tile of an 8-tile split: 3.4 per CU, k_profq 1.27 -> 1.22 ms, the tile's step 7.15 -> 6.9)
  // (developer switches: POMGPU_PROFQ_ROWS8 / _ROWS2 force a shape, POMGPU_PROFQ_NOPACE drops the barrier)
  const int rows8 = !SW(c, PROFQ_ROWS2) && (SW(c, PROFQ_ROWS8) || (long)((c->P.iml + 63) / 64) * ((c->P.jml + 7) / 8) >= 3 * 256);
  rho_rt = (rho_rt ? 1 : 0) | (SW(c, PROFQ_NOPACE) ? 0 : 2);
#define PQ(FP, FF)                                                                                                                       \
  do {                                                                                                                                   \
    if (rows8) LAUNCHN(c, "k_profq", (k_profq<FP, FF, PROFQ_BIG>), dim3((c->P.iml + 63) / 64, (nrows + PROFQ_BIG - 1) / PROFQ_BIG, 1), dim3(64, PROFQ_BIG, 1), c->P, rho_rt, jfirst, jlast); \
    else LAUNCHN(c, "k_profq", (k_profq<FP, FF, 2>), dim3((c->P.iml + 63) / 64, (nrows + 1) / 2, 1), dim3(64, 2, 1), c->P, rho_rt, jfirst, jlast);       \
  } while (0)
  if (fuse_filter) { if (fuse_prod == 0) PQ(0, 1); else if (fuse_prod == 1) PQ(1, 1); else PQ(2, 1); }
  else { if (fuse_prod == 0) PQ(0, 0); else if (fuse_prod == 1) PQ(1, 0); else PQ(2, 0); }
#undef PQ
  const KP &P = c->P;
  if (P.W || P.E || P.S || P.N) {
    const int len = P.im > P.jm ? P.im : P.jm;
    LAUNCH(c, k_profq_rim, dim3((len + 63) / 64, 4, P.kb), dim3(64, 1, 1), c->P);
  }
}
template <int KBT>
static void launch_proft_reg(pomgpu_ctx *c, double *f, const double *wfsurf, const double *fsurf, int nbc) {
  if (nbc == 2 || nbc == 4) LAUNCHN(c, "k_proft_reg", (k_proft_reg<KBT, 1>), rowgrid(c->P, ROWS_PROFT), rowblk(ROWS_PROFT), c->P, f, wfsurf, fsurf, nbc);
  else LAUNCHN(c, "k_proft_reg", (k_proft_reg<KBT, 0>), rowgrid(c->P, ROWS_PROFT), rowblk(ROWS_PROFT), c->P, f, wfsurf, fsurf, nbc);
}
void launch_proft(pomgpu_ctx *c, double *f, const double *wfsurf, const double *fsurf, int nbc) {
  const int kb = c->P.kb;
  if (SW(c, THOMAS_SCRATCH) || kb > 64 || kb < 6) LAUNCH(c, k_proft, colgrid(c->P), colblk(), c->P, f, wfsurf, fsurf, nbc);
  else if (kb <= 24) launch_proft_reg<24>(c, f, wfsurf, fsurf, nbc);
  else if (kb <= 32) launch_proft_reg<32>(c, f, wfsurf, fsurf, nbc);
  else if (kb <= 40) launch_proft_reg<40>(c, f, wfsurf, fsurf, nbc);
  else if (kb <= 44) launch_proft_reg<44>(c, f, wfsurf, fsurf, nbc);
  else if (kb <= 50) launch_proft_reg<50>(c, f, wfsurf, fsurf, nbc);
  else if (kb <= 56) launch_proft_reg<56>(c, f, wfsurf, fsurf, nbc);
  else launch_proft_reg<64>(c, f, wfsurf, fsurf, nbc);
}
// proft for T and S in one launch (advance.f:439-440); 0 = not applicable (no register kernel for this kb, or the two tracers' surface
// conditions differ in whether the short-wave term is compiled in): the caller launches them one by one
template <int KBT>
static void launch_proft2_reg(pomgpu_ctx *c, const ProftArgs &a0, const ProftArgs &a1, int sw) {
  if (sw) LAUNCHN(c, "k_proft_reg2", (k_proft_reg2<KBT, 1>), twin(rowgrid(c->P, ROWS_PROFT)), rowblk(ROWS_PROFT), c->P, a0, a1);
  else LAUNCHN(c, "k_proft_reg2", (k_proft_reg2<KBT, 0>), twin(rowgrid(c->P, ROWS_PROFT)), rowblk(ROWS_PROFT), c->P, a0, a1);
}
int launch_proft2(pomgpu_ctx *c, double *f0, const double *wfsurf0, const double *fsurf0, int nbc0, double *f1, const double *wfsurf1, const double *fsurf1, int nbc1) {
  const int kb = c->P.kb, sw0 = (nbc0 == 2 || nbc0 == 4), sw1 = (nbc1 == 2 || nbc1 == 4);
  if (SW(c, NO_TWIN) || SW(c, THOMAS_SCRATCH) || kb > 64 || kb < 6 || sw0 != sw1) return 0;
  const ProftArgs a0 = {f0, wfsurf0, fsurf0, nbc0}, a1 = {f1, wfsurf1, fsurf1, nbc1};
  if (kb <= 24) launch_proft2_reg<24>(c, a0, a1, sw0);
  else if (kb <= 32) launch_proft2_reg<32>(c, a0, a1, sw0);
  else if (kb <= 40) launch_proft2_reg<40>(c, a0, a1, sw0);
  else if (kb <= 44) launch_proft2_reg<44>(c, a0, a1, sw0);
  else if (kb <= 50) launch_proft2_reg<50>(c, a0, a1, sw0);
  else if (kb <= 56) launch_proft2_reg<56>(c, a0, a1, sw0);
  else launch_proft2_reg<64>(c, a0, a1, sw0);
  return 1;
}
void launch_advu_profu(pomgpu_ctx *c, int do_adv, int do_prof) { LAUNCH(c, k_advu_profu, colgrid(c->P), colblk(), c->P, do_adv, do_prof); }
void launch_advv_profv(pomgpu_ctx *c, int do_adv, int do_prof) { LAUNCH(c, k_advv_profv, colgrid(c->P), colblk(), c->P, do_adv, do_prof); }
template <int KBT> static void launch_profuv_reg_t(pomgpu_ctx *c) {
  const KP &P = c->P;
  if (!SW(c, NO_TWIN)) {
    LAUNCHN(c, "k_profuv_reg2", (k_profuv_reg2<KBT>), dim3((P.iml + 61) / 62, (P.jml + ROWS_PROFUV - 1) / ROWS_PROFUV, 2), rowblk(ROWS_PROFUV), c->P);
    return;
  }
  LAUNCHN(c, "k_profu_reg", (k_profuv_reg<KBT, 0>), dim3((P.iml + 61) / 62, (P.jml + ROWS_PROFUV - 1) / ROWS_PROFUV, 1), rowblk(ROWS_PROFUV), c->P);
  LAUNCHN(c, "k_profv_reg", (k_profuv_reg<KBT, 1>), rowgrid(P, ROWS_PROFUV), rowblk(ROWS_PROFUV), c->P);
}
int launch_profuv_reg(pomgpu_ctx *c) {
  const int kb = c->P.kb;
  if (SW(c, THOMAS_SCRATCH) || kb > 64 || kb < 6) return 0;
  if (kb <= 24) launch_profuv_reg_t<24>(c);
  else if (kb <= 32) launch_profuv_reg_t<32>(c);
  else if (kb <= 40) launch_profuv_reg_t<40>(c);
  else if (kb <= 44) launch_profuv_reg_t<44>(c);
  else if (kb <= 50) launch_profuv_reg_t<50>(c);
  else if (kb <= 56) launch_profuv_reg_t<56>(c);
  else launch_profuv_reg_t<64>(c);
  return 1;
}
template <int KBT> static void launch_uv_filter_reg_t(pomgpu_ctx *c, int own) {
  if (!SW(c, NO_TWIN)) { LAUNCHN(c, "k_uv_filter_reg2", (k_uv_filter_reg2<KBT>), twin(rowgrid(c->P, ROWS_UVF)), rowblk(ROWS_UVF), c->P, own); return; }
  LAUNCHN(c, "k_uv_filter_reg", (k_uv_filter_reg<KBT, 0>), rowgrid(c->P, ROWS_UVF), rowblk(ROWS_UVF), c->P, own);
  LAUNCHN(c, "k_uv_filter_reg", (k_uv_filter_reg<KBT, 1>), rowgrid(c->P, ROWS_UVF), rowblk(ROWS_UVF), c->P, own);
}
static int launch_uv_filter_reg(pomgpu_ctx *c, int own) {
  const int kb = c->P.kb;
  if (SW(c, THOMAS_SCRATCH) || kb > 64 || kb < 6) return 0;
  if (kb <= 24) launch_uv_filter_reg_t<24>(c, own);
  else if (kb <= 32) launch_uv_filter_reg_t<32>(c, own);
  else if (kb <= 40) launch_uv_filter_reg_t<40>(c, own);
  else if (kb <= 44) launch_uv_filter_reg_t<44>(c, own);
  else if (kb <= 50) launch_uv_filter_reg_t<50>(c, own);
  else if (kb <= 56) launch_uv_filter_reg_t<56>(c, own);
  else launch_uv_filter_reg_t<64>(c, own);
  return 1;
}
void launch_uv_filter(pomgpu_ctx *c, int own) {
  if (launch_uv_filter_reg(c, own)) return;
  LAUNCH(c, k_uv_filter, colgrid(c->P), colblk(), c->P, own);
}
