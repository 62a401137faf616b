// k_tile.hip -- column-marching stencil kernels and the derived 2-D coefficient arrays.
//
// Profile of the first-cut cell kernels on MI355X (2048x1536x50): a cell of advt2 issued 56 global
// loads (each i+-1 / j+-1 neighbour a separate 512-byte wavefront request) and evaluated every face
// flux twice; the kernels ran at ~1.4-1.9 TB/s of algorithmic traffic, bound by the L1/TA request
// rate and fp64 divides, not by HBM.  A row-marching version (a wavefront walks a strip of rows at one
// level) moved the bottleneck to L2-miss traffic: the 14 two-dimensional coefficient operands of a
// cell were re-fetched for every level (31 GB per launch against 7.5 GB algorithmic).  The kernels
// here let a thread own the water column (i,j) instead; see the comment above k_advt2_col.
#include "pomgpu_internal.hpp"


// ---- derived 2-D coefficients -------------------------------------------------------------------
__global__ void k_coef_static(KP P) {
  const int i = TID_I, j = TID_J;
  if (i > P.iml || j > P.jml) return;
  const int iw = i > 1 ? i - 1 : 1, js = j > 1 ? j - 1 : 1;
  P.m8[IX2(i, j)] = (unsigned char)((F2(fsm, i, j) != 0. ? 1 : 0) | (F2(dum, i, j) != 0. ? 2 : 0) | (F2(dvm, i, j) != 0. ? 4 : 0));
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
  K2(RDX, i, j) = 1.0 / F2(dx, i, j);
  K2(RDY, i, j) = 1.0 / F2(dy, i, j);
  // reciprocals of the vertical grid arrays (divi(), pomgpu_internal.hpp); 1/0 = inf where dz, dzz are 0 (level kb): never used
  if (j == 1)
    for (int n = i - 1; n < POM_NBLK1D * P.kb; n += P.iml) P.r1[n] = 1.0 / P.b1[n];
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
// ---- advt2, nitera == 1, COLUMN-MARCHING version -----------------------------------------------------
// PMC profile of the row-marching kernel above (2048x1536x50): 31 GB of L2-miss traffic per launch
// against 7.5 GB algorithmic -- the 14 two-dimensional coefficient operands of a cell are re-fetched
// for every level -- and ~1e9 L1 line accesses.  Here a thread owns the water column (i,j):
//   * all 2-D coefficients of its three faces live in registers for the whole column;
//   * the level loop is software-pipelined: iteration L issues the loads of level L+1, evaluates
//     the faces of level L from registers filled one iteration earlier, and finishes level L-1
//     (whose bottom face is the top face of level L, evaluated once);
//   * west operands and the east face come from the neighbour lane of a halo-lane wavefront
//     (pomgpu_internal.hpp): no lane ever loads in the middle of an iteration -- vmcnt counts in
//     order, so such a load would wait for the whole prefetch batch of the next level.
// Per cell: 13 512-byte wavefront loads (was 56 in the cell kernel), 3 face evaluations
// in x/y (was 4) and 1 in z (was 2).
// NF = 2 advances T and S in ONE pass (advance.f:431-432 calls advt2 twice): u, v, w, aam and the face
// coefficients are read once for both fields.
struct TFields { const double *fb[2], *f[2], *fcl[2]; double *ff[2]; };
// Operands of a level.  Shared with the rows above and below through the workgroup's LDS slab (RowShare,
// pomgpu_internal.hpp), in this order: fb[0..NF-1], fclim[0..NF-1], aam, v; this row only: u, w.
// c: the wavefront's own row, h: its share of the two rows outside the workgroup.
template <int NF> struct LevT { double c[2 * NF + 2], o[2], h[ROWSHARE_SLOTS(2 * NF + 2)]; };
template <int NS> struct NbrT { double s[NS], n[NS]; };
struct CoefT { double cm, hs, msk, ds_num; InvD den; };   // mass-flux coefficient, h sum, mask, metric sums of one face
__device__ __forceinline__ CoefT coef_x(const KP &P, int i, int j) {
  CoefT c; c.cm = K2(CMX, i, j); c.hs = K2(HSX, i, j); c.msk = F2(dum, i, j); c.ds_num = K2(DYSX, i, j); c.den = inv_of(K2(DXSX, i, j)); return c;
}
__device__ __forceinline__ CoefT coef_y(const KP &P, int i, int j) {
  CoefT c; c.cm = K2(CMY, i, j); c.hs = K2(HSY, i, j); c.msk = F2(dvm, i, j); c.ds_num = K2(DXSY, i, j); c.den = inv_of(K2(DYSY, i, j)); return c;
}
// face between a "lo" cell (west / south) and a "hi" cell; vel = u or v on that face
__device__ __forceinline__ FaceT advt2_face(const KP &P, const CoefT &c, double vel, double fb_hi, double fb_lo, double fc_hi,
                                            double fc_lo, double am_hi, double am_lo) {
  FaceT f;
  f.adv = upw_(c.cm * vel, fb_lo, fb_hi);
  const double am = 0.5 * (am_hi + am_lo);
  f.dif = divi(-am * c.hs * P.tprni * ((fb_hi - fc_hi) - (fb_lo - fc_lo)) * c.msk * c.ds_num * 0.5, c.den);
  return f;
}
#if COL_WX != 1
#error "the row-sharing column kernels keep one wavefront per workgroup row (COL_WX == 1)"
#endif
// What keeps the level loop a pipeline (each item was a stall in the ISA of the first version):
//   * every load of a level is issued in ONE batch a whole iteration before its first use, and NO memory instruction sits
//     inside a branch: where control flow joins, the compiler no longer knows how many operations are outstanding and
//     waits for vmcnt(0), i.e. for the batch it has just issued.  Lanes / iterations with nothing to store aim outside
//     the buffer (BOFF_NONE, dropped by the hardware); the last iterations re-request level kbm1;
//   * dz(k) comes through the constant address space (scalar load) -- as a vector load it queued behind the batch;
//   * buffer addressing: no vector instruction computes an address; two register sets alternate as current / next
//     level (no copies); divisors that are fixed along the column are inverted once (divi).
template <int NF>
__global__ void __launch_bounds__(64 * LDS_ROWS) k_advt2_col(KP P, TFields A) {
  constexpr int NS = 2 * NF + 2, NH = ROWSHARE_SLOTS(NS), AM = 2 * NF, VV = 2 * NF + 1;
  HALO_XCD_DECODE_R(LDS_ROWS)                                           // a workgroup outside the grid leaves as a whole
  const int r = WAVE_UNIFORM((int)threadIdx.y), j0w = j - r;
  const bool jrow = j <= P.jml;                             // rows beyond the tile shadow row jml and store nothing
  const int jc = jrow ? j : P.jml;
  const bool icol = jrow && (lane >= 1 && lane <= 62 && i0 <= P.iml);   // this lane owns an output column
#ifdef POMGPU_EMU
  if (!icol) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);          // halo / padding lanes shadow a valid column
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const int js = jc > 1 ? jc - 1 : 1, jn = jc < P.jml ? jc + 1 : P.jml;
  const bool in = icol && (i >= 2 && i <= P.imm1 && jc >= 2 && jc <= P.jmm1);
  const double fsm = F2(fsm, i, jc);
  // column-resident coefficients of the west, south and north faces and of the cell
  const CoefT cw = coef_x(P, i, jc), cs = coef_y(P, i, jc), cn = coef_y(P, i, jn);
  const double art = F2(art, i, jc), hea = K2(HEA, i, jc);
  const InvD hfa = inv_of(K2(HFA, i, jc));
  BufA bs[NS], bo[2], bff[NF], bh[NH];
  const double *ps[NS];
#pragma unroll
  for (int f = 0; f < NF; f++) { ps[f] = A.fb[f]; ps[NF + f] = A.fcl[f]; bff[f] = BUF3(A.ff[f]); }
  ps[AM] = A3(aam); ps[VV] = A3(v); bo[0] = BUF3(A3(u)); bo[1] = BUF3(A3(w));
#pragma unroll
  for (int x = 0; x < NS; x++) bs[x] = BUF3(ps[x]);
  const RowShare<NS> S = rowshare_setup<NS>(P, r, j, j0w, i);
#pragma unroll
  for (int q = 0; q < NH; q++) bh[q] = BUF3(rowshare_pick<NS>(ps, S.hop[q]));
  const unsigned oc = BOFF2(i, jc), lvb = LVB;
  const unsigned ost = in ? oc : BOFF_NONE;
#ifndef POMGPU_EMU
  __shared__ double slab[2][NS][ROWSHARE_ROWS][64];
#else
  const unsigned os = BOFF2(i, js), on = BOFF2(i, jn);
#endif
  auto load = [&](LevT<NF> &L, unsigned lv) {
#pragma unroll
    for (int x = 0; x < NS; x++) L.c[x] = bld(bs[x], oc, lv);
    L.o[0] = bld(bo[0], oc, lv); L.o[1] = bld(bo[1], oc, lv);
#pragma unroll
    for (int q = 0; q < NH; q++) L.h[q] = bld(bh[q], S.hoff[q], lv);
  };
  double f1[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) f1[f] = G3(A.f[f], i, jc, 1);
  const int kbm1 = P.kbm1;
  // carried from level L-1 to its completion in iteration L
  double p_adv[NF], p_dif[NF], p_fb[NF], p_zu[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) p_adv[f] = p_dif[f] = p_fb[f] = p_zu[f] = 0.;
  // one iteration: request level L+1 into `nxt`, park level L (`cur`) in the slab, evaluate its faces, finish level L-1.
  // (Tried: parking a level one iteration ahead, the level itself re-read from the slab, one register set less -- the
  // barrier then waits for nobody less than an iteration behind, but the LDS round trip of the own row sits in front of
  // every level's arithmetic: 3-5 % slower.)
  auto step = [&](const int L, const int par, const LevT<NF> &cur, LevT<NF> &nxt) {
    load(nxt, (unsigned)(L < kbm1 ? L : kbm1 - 1) * lvb);     // the last iterations re-request level kbm1
    NbrT<NS> nb;
#ifndef POMGPU_EMU
#pragma unroll
    for (int x = 0; x < NS; x++) slab[par][x][r + 1][lane] = cur.c[x];
#pragma unroll
    for (int q = 0; q < NH; q++) slab[par][S.hop[q]][S.hrow[q]][lane] = cur.h[q];
    __syncthreads();
#pragma unroll
    for (int x = 0; x < NS; x++) { nb.s[x] = slab[par][x][S.ss][lane]; nb.n[x] = slab[par][x][S.sn][lane]; }
#else
    const unsigned lvc = (unsigned)((L <= kbm1 ? L : kbm1) - 1) * lvb;
#pragma unroll
    for (int x = 0; x < NS; x++) { nb.s[x] = bld(bs[x], os, lvc); nb.n[x] = bld(bs[x], on, lvc); }
#endif
    const double am_c = cur.c[AM], u_c = cur.o[0], w_c = cur.o[1];
    const double am_w = halo_w(am_c, [&] { return F3(aam, iw, jc, L); });
    InvD dzk; dzk.b = dzk.y = 0.;
    if (L >= 2) { dzk.b = F1(dz, L - 1); dzk.y = R1(dz, L - 1); }
#pragma unroll
    for (int f = 0; f < NF; f++) {
      const double fb_c = cur.c[f], fc_c = cur.c[NF + f];
      double zu = 0.;                                                              // top face of level L (0 below kbm1)
      double s_adv = 0., s_dif = 0.;
      if (L <= kbm1) {
        const double fb_w = halo_w(fb_c, [&] { return G3(A.fb[f], iw, jc, L); });
        const double fc_w = halo_w(fc_c, [&] { return G3(A.fcl[f], iw, jc, L); });
        const FaceT xw = advt2_face(P, cw, u_c, fb_c, fb_w, fc_c, fc_w, am_c, am_w);
        auto east = [&] {                                   // emulation only: the east face from memory
          return advt2_face(P, coef_x(P, ie, jc), F3(u, ie, jc, L), G3(A.fb[f], ie, jc, L), fb_c, G3(A.fcl[f], ie, jc, L), fc_c,
                            F3(aam, ie, jc, L), am_c);
        };
        FaceT xe;
        xe.adv = halo_e(xw.adv, [&] { return east().adv; });
        xe.dif = halo_e(xw.dif, [&] { return east().dif; });
        const FaceT ys = advt2_face(P, cs, cur.c[VV], fb_c, nb.s[f], fc_c, nb.s[NF + f], am_c, nb.s[AM]);
        const FaceT yn = advt2_face(P, cn, nb.n[VV], nb.n[f], fb_c, nb.n[NF + f], fc_c, nb.n[AM], am_c);
        s_adv = xe.adv - xw.adv + yn.adv - ys.adv;                                            // solver.f:670-671
        s_dif = xe.dif - xw.dif + yn.dif - ys.dif;                                            // :721-722
        zu = (L == 1) ? w_c * f1[f] * art : upw_(w_c, fb_c, p_fb[f]) * art;                   // :646-662
      }
      {                                                     // finish level L-1: its bottom face is this level's top face
        double rr = p_adv[f] + divi(p_zu[f] - zu, dzk);                                       // :670-672
        rr = divi(p_fb[f] * hea - P.dti2 * rr, hfa);                                          // :673-674
        rr = rr * fsm;                                                                        // :1899
        rr = rr - divi(P.dti2 * p_dif[f], hfa);                                               // :721-723
        // every lane stores, every iteration: lanes without an interior column (and the first iteration, which has
        // no finished level yet) aim outside the buffer and the hardware drops the store
        bst(bff[f], (L >= 2 && L <= kbm1 + 1) ? ost : BOFF_NONE, (unsigned)(L >= 2 ? L - 2 : 0) * lvb, rr);
      }
      p_adv[f] = s_adv; p_dif[f] = s_dif; p_fb[f] = fb_c; p_zu[f] = zu;
    }
  };
  // two register sets take turns as "current" and "next" level: no copy between iterations
  LevT<NF> ra, rb;
  load(ra, 0u);
  rb = ra;
  for (int L = 1; L <= kbm1 + 1; L += 2) {
    step(L, 1, ra, rb);
    step(L + 1, 0, rb, ra);                                 // for even kbm1 one iteration too many: its stores aim outside the buffer
  }
  // :1899 on what the loop did not write: level kb everywhere, every level of the rim cells.  AFTER the loop: a load
  // inside it (even one only rim lanes execute) makes the compiler wait for vmcnt(0) where the paths join
  if (icol) {
    for (int k = in ? P.kb : 1; k <= P.kb; k++) {
#pragma unroll
      for (int f = 0; f < NF; f++) G3(A.ff[f], i, jc, k) = G3(A.ff[f], i, jc, k) * fsm;
    }
  }
}

// ---- advq, flux + step fused (single tile) -- solver.f:411-477 -----------------------------------------
// Same structure as k_advt2_col: column-resident face coefficients, software-pipelined level loop,
// neighbour-lane operands.  The reference exchanges xflux/yflux between the flux and the step
// loops (:458-459); with all neighbours -1 that exchange is a no-op and the two halves fuse: the
// fluxes never reach memory.  Used only when the context has no exchange hook (one tile).
// NF = 2 advances q2 and q2l in ONE pass (advance.f:407-408 calls advq twice): u, v, w, aam and every
// face coefficient are read once for both -- 12.5 array passes instead of 2 x 9.5.
struct QFields { const double *q[2], *qb[2]; double *qf[2]; };
// operands of a level, shared through the workgroup's LDS slab in this order: q[0..NF-1], qb[0..NF-1], aam, v; own row only: u, w
template <int NF> struct LevQa { double c[2 * NF + 2], o[2], h[ROWSHARE_SLOTS(2 * NF + 2)]; };
struct CoefQ { double dts, hs, msk, ds_num; InvD den; };
__device__ __forceinline__ CoefQ coefq_x(const KP &P, int i, int j) {
  CoefQ c; c.dts = K2(DTSX, i, j); c.hs = K2(HSX, i, j); c.msk = F2(dum, i, j); c.den = inv_of(K2(DXSX, i, j)); c.ds_num = K2(DYSX, i, j); return c;
}
__device__ __forceinline__ CoefQ coefq_y(const KP &P, int i, int j) {
  CoefQ c; c.dts = K2(DTSY, i, j); c.hs = K2(HSY, i, j); c.msk = F2(dvm, i, j); c.den = inv_of(K2(DYSY, i, j)); c.ds_num = K2(DXSY, i, j); return c;
}
// flux through the face between a "lo" (west/south) and a "hi" cell at w-level k (:428-453)
__device__ __forceinline__ double advq_face(const CoefQ &c, double q_hi, double q_lo, double vel_k, double vel_km1, double am_hi_k,
                                            double am_lo_k, double am_hi_m, double am_lo_m, double qb_hi, double qb_lo) {
  double x = .125 * (q_hi + q_lo) * c.dts * (vel_k + vel_km1);
  x = x - divi(.25 * (am_hi_k + am_lo_k + am_hi_m + am_lo_m) * c.hs * (qb_hi - qb_lo) * c.msk, c.den);
  return .5 * c.ds_num * x;
}
// loop discipline and row sharing as in k_advt2_col
template <int NF>
__global__ void __launch_bounds__(64 * LDS_ROWS) k_advq_col(KP P, QFields A, int zero_else) {
  constexpr int NS = 2 * NF + 2, NH = ROWSHARE_SLOTS(NS), AM = 2 * NF, VV = 2 * NF + 1;
  HALO_XCD_DECODE_R(LDS_ROWS)
  const int r = WAVE_UNIFORM((int)threadIdx.y), j0w = j - r;
  const bool jrow = j <= P.jml;
  const int jc = jrow ? j : P.jml;
  const bool icol = jrow && (lane >= 1 && lane <= 62 && i0 <= P.iml);
#ifdef POMGPU_EMU
  if (!icol) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const int js = jc > 1 ? jc - 1 : 1, jn = jc < P.jml ? jc + 1 : P.jml;
  const bool in = icol && (i >= 2 && i <= P.imm1 && jc >= 2 && jc <= P.jmm1);
  const CoefQ cw = coefq_x(P, i, jc), cs = coefq_y(P, i, jc), cn = coefq_y(P, i, jn);
  const double art = F2(art, i, jc), hea = K2(HEA, i, jc);
  const InvD hfa = inv_of(K2(HFA, i, jc));
  const int kb = P.kb, kbm1 = P.kbm1;
  BufA bs[NS], bo[2], bqf[NF], bh[NH];
  const double *ps[NS];
#pragma unroll
  for (int f = 0; f < NF; f++) { ps[f] = A.q[f]; ps[NF + f] = A.qb[f]; bqf[f] = BUF3(A.qf[f]); }
  ps[AM] = A3(aam); ps[VV] = A3(v); bo[0] = BUF3(A3(u)); bo[1] = BUF3(A3(w));
#pragma unroll
  for (int x = 0; x < NS; x++) bs[x] = BUF3(ps[x]);
  const RowShare<NS> S = rowshare_setup<NS>(P, r, j, j0w, i);
#pragma unroll
  for (int q = 0; q < NH; q++) bh[q] = BUF3(rowshare_pick<NS>(ps, S.hop[q]));
  const unsigned oc = BOFF2(i, jc), lvb = LVB;
  // interior columns get the new value, every other owned column a zero (zero_else) or nothing
  const unsigned ost = in ? oc : ((zero_else && icol) ? oc : BOFF_NONE);
#ifndef POMGPU_EMU
  __shared__ double slab[2][NS][ROWSHARE_ROWS][64];
#else
  const unsigned os = BOFF2(i, js), on = BOFF2(i, jn);
#endif
  auto load = [&](LevQa<NF> &L, int k) {
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
#pragma unroll
    for (int x = 0; x < NS; x++) L.c[x] = bld(bs[x], oc, lv);
    L.o[0] = bld(bo[0], oc, lv); L.o[1] = bld(bo[1], oc, lv);
#pragma unroll
    for (int q = 0; q < NH; q++) L.h[q] = bld(bh[q], S.hoff[q], lv);
  };
  double u_m = 0., v_m = 0., vn_m = 0., am_m = 0., ams_m = 0., amn_m = 0.;   // u, v, v(j+1), aam (c, s, n) of level L-1
  double am_w_prv = 0.;                       // aam(i-1,j,L-1) as seen by this lane
  double wq_pp[NF], wq_p[NF];                 // w*q of levels L-2 and L-1
  double xe_p[NF], xw_p[NF], yn_p[NF], ys_p[NF], qb_p[NF];   // faces and qb of level L-1, waiting for w(L)*q(L)
#pragma unroll
  for (int f = 0; f < NF; f++) wq_pp[f] = wq_p[f] = xe_p[f] = xw_p[f] = yn_p[f] = ys_p[f] = qb_p[f] = 0.;
  auto step = [&](const int L, const int par, const LevQa<NF> &cur, LevQa<NF> &nxt) {
    load(nxt, L + 1 <= kb ? L + 1 : kb);                    // the last iterations re-request level kb
    NbrT<NS> nb;
#ifndef POMGPU_EMU
#pragma unroll
    for (int x = 0; x < NS; x++) slab[par][x][r + 1][lane] = cur.c[x];
#pragma unroll
    for (int q = 0; q < NH; q++) slab[par][S.hop[q]][S.hrow[q]][lane] = cur.h[q];
    __syncthreads();
#pragma unroll
    for (int x = 0; x < NS; x++) { nb.s[x] = slab[par][x][S.ss][lane]; nb.n[x] = slab[par][x][S.sn][lane]; }
#else
    const unsigned lvc = (unsigned)((L <= kb ? L : kb) - 1) * lvb;
#pragma unroll
    for (int x = 0; x < NS; x++) { nb.s[x] = bld(bs[x], os, lvc); nb.n[x] = bld(bs[x], on, lvc); }
#endif
    const double am_c = cur.c[AM], v_c = cur.c[VV], u_c = cur.o[0], w_c = cur.o[1];
    const double am_s = nb.s[AM], am_n = nb.n[AM], v_n = nb.n[VV];
    const double am_w = halo_w(am_c, [&] { return F3(aam, iw, jc, L); });
    const int k = L - 1;                                    // level completed in this iteration
    InvD dz2; dz2.b = dz2.y = 0.;
    if (k >= 2 && k <= kbm1) { dz2.b = F1(dz, k) + F1(dz, k - 1); dz2.y = 1.0 / dz2.b; }
#pragma unroll
    for (int f = 0; f < NF; f++) {
      const double q_c = cur.c[f], qb_c = cur.c[NF + f];
      double xe_c = 0., xw_c = 0., yn_c = 0., ys_c = 0.;
      if (L >= 2 && L <= kbm1) {
        const double q_w = halo_w(q_c, [&] { return G3(A.q[f], iw, jc, L); });
        const double qb_w = halo_w(qb_c, [&] { return G3(A.qb[f], iw, jc, L); });
        const double xw = advq_face(cw, q_c, q_w, u_c, u_m, am_c, am_w, am_m, am_w_prv, qb_c, qb_w);
        xw_c = xw;
        xe_c = halo_e(xw, [&] {                             // emulation only: the east face from memory
          return advq_face(coefq_x(P, ie, jc), G3(A.q[f], ie, jc, L), q_c, F3(u, ie, jc, L), F3(u, ie, jc, L - 1), F3(aam, ie, jc, L), am_c,
                           F3(aam, ie, jc, L - 1), am_m, G3(A.qb[f], ie, jc, L), qb_c);
        });
        ys_c = advq_face(cs, q_c, nb.s[f], v_c, v_m, am_c, am_s, am_m, ams_m, qb_c, nb.s[NF + f]);
        yn_c = advq_face(cn, nb.n[f], q_c, v_n, vn_m, am_n, am_c, amn_m, am_m, nb.n[NF + f], qb_c);
      }
      const double wq_c = w_c * q_c;
      {
        double rr = 0.;
        if (in && k >= 2 && k <= kbm1) {
          rr = divi((wq_pp[f] - wq_c) * art, dz2) + xe_p[f] - xw_p[f] + yn_p[f] - ys_p[f];    // :465-468
          rr = divi(hea * qb_p[f] - P.dti2 * rr, hfa);                                        // :469-471
        }
        // levels 2..kbm1 of interior columns: the new value; with zero_else every other owned cell (levels 1..kb-1 here,
        // kb below): zero; everything else aims outside the buffer
        const bool lev_in = (k >= 2 && k <= kbm1);
        const unsigned o = (k >= 1) ? (lev_in ? ost : ((zero_else && icol) ? oc : BOFF_NONE)) : BOFF_NONE;
        bst(bqf[f], o, (unsigned)WAVE_UNIFORM(k >= 1 ? k - 1 : 0) * lvb, rr);
      }
      wq_pp[f] = wq_p[f]; wq_p[f] = wq_c; qb_p[f] = qb_c;
      xe_p[f] = xe_c; xw_p[f] = xw_c; yn_p[f] = yn_c; ys_p[f] = ys_c;
    }
    am_w_prv = am_w;
    u_m = u_c; v_m = v_c; vn_m = v_n; am_m = am_c; ams_m = am_s; amn_m = am_n;
  };
  LevQa<NF> ra, rb;
  load(ra, 1);
  rb = ra;
  for (int L = 1; L <= kb; L += 2) {
    step(L, 1, ra, rb);
    if (L + 1 <= kb) step(L + 1, 0, rb, ra);                // uniform for the whole workgroup (barrier inside)
  }
  if (icol && zero_else) {
#pragma unroll
    for (int f = 0; f < NF; f++) G3(A.qf[f], i, jc, kb) = 0.;
  }
}

// ---- advct, all three phases fused (single tile) -- solver.f:201-408 ------------------------------------
// The reference forms curv, xflux, yflux, exchanges them, forms advx and a second set of fluxes,
// exchanges, forms advy: 27 array passes through the three cell kernels of k_adv.hip (PMC: 33.7 GB per
// launch against 8.8 GB for "read u,v,ub,vb,aam, write advx,advy").  With all neighbours -1 nothing is
// exchanged and every flux is a function of the five operands around (i,j,k), so the fluxes stay in
// registers.  There is no vertical coupling; the level loop exists to keep the 28 two-dimensional
// coefficients of a column in registers and to have the next level's 15 row loads in flight.
//
// Lane map: a wavefront covers 64 consecutive columns but only lanes 1..62 own an output column;
// lanes 0 and 63 are HALO lanes that load their column and feed the shuffles (lane 0 supplies the
// west operands / west face flux of lane 1, lane 63 the east operands / corner flux of lane 62), so
// no lane ever needs a mid-iteration fallback load (vmcnt is in-order) and no flux is evaluated
// from memory.  Wave w covers columns 62w .. 62w+63.
// the same quantities evaluated from memory (emulation fallbacks only); i, j inside the tile
__device__ double advct_xf_mem(const KP &P, int i, int j, int k) {                 // x-eq. xflux, 2<=j
  if (i < 2 || i > P.imm1) return 0.;
  double xf = .125 * (K2(DTSX, i + 1, j) * F3(u, i + 1, j, k) + K2(DTSX, i, j) * F3(u, i, j, k)) * (F3(u, i + 1, j, k) + F3(u, i, j, k));
  xf = xf - F2(dt, i, j) * F3(aam, i, j, k) * 2. * (F3(ub, i + 1, j, k) - F3(ub, i, j, k)) / F2(dx, i, j);
  return F2(dy, i, j) * xf;
}
__device__ double advct_curv_mem(const KP &P, int i, int j, int k) {
  if (i < 2 || i > P.imm1 || j < 2 || j > P.jmm1) return 0.;
  return .25 * ((F3(v, i, j + 1, k) + F3(v, i, j, k)) * K2(CVA, i, j) - (F3(u, i + 1, j, k) + F3(u, i, j, k)) * K2(CVB, i, j)) / F2(art, i, j);
}
__device__ double advct_xg_mem(const KP &P, int i, int j, int k) {                 // y-eq. xflux at corner (i,j), 2<=i, 2<=j<=jmm1
  double xg = .125 * (K2(DTSX, i, j) * F3(u, i, j, k) + K2(DTSX, i, j - 1) * F3(u, i, j - 1, k)) * (F3(v, i, j, k) + F3(v, i - 1, j, k));
  const double dtaam = .25 * K2(DT4, i, j) * (F3(aam, i, j, k) + F3(aam, i - 1, j, k) + F3(aam, i, j - 1, k) + F3(aam, i - 1, j - 1, k));
  const double dy4 = K2(DY4, i, j);
  xg = xg - dtaam * ((F3(ub, i, j, k) - F3(ub, i, j - 1, k)) / dy4 + (F3(vb, i, j, k) - F3(vb, i - 1, j, k)) / K2(DX4, i, j));
  return .25 * dy4 * xg;
}
__device__ double advct_yf_mem(const KP &P, int i, int j, int k) {                 // x-eq. yflux at corner (i,j), 2<=i<=imm1, 2<=j
  double yf = .125 * (K2(DTSY, i, j) * F3(v, i, j, k) + K2(DTSY, i - 1, j) * F3(v, i - 1, j, k)) * (F3(u, i, j, k) + F3(u, i, j - 1, k));
  const double dtaam = .25 * K2(DT4, i, j) * (F3(aam, i, j, k) + F3(aam, i - 1, j, k) + F3(aam, i, j - 1, k) + F3(aam, i - 1, j - 1, k));
  const double dx4 = K2(DX4, i, j);
  yf = yf - dtaam * ((F3(ub, i, j, k) - F3(ub, i, j - 1, k)) / K2(DY4, i, j) + (F3(vb, i, j, k) - F3(vb, i - 1, j, k)) / dx4);
  return .25 * dx4 * yf;
}
__device__ double advct_yg_mem(const KP &P, int i, int j, int k) {                 // y-eq. yflux at the centre of (i,j), 2<=i, 2<=j<=jmm1
  double yg = .125 * (K2(DTSY, i, j + 1) * F3(v, i, j + 1, k) + K2(DTSY, i, j) * F3(v, i, j, k)) * (F3(v, i, j + 1, k) + F3(v, i, j, k));
  yg = yg - F2(dt, i, j) * F3(aam, i, j, k) * 2. * (F3(vb, i, j + 1, k) - F3(vb, i, j, k)) / F2(dy, i, j);
  return F2(dx, i, j) * yg;
}
// advct on tiles without exchanging whole intermediate arrays.  What a tile cannot form itself is little: advx(2,j)
// needs xflux(1,j) and curv(1,j) (solver.f:284-301), advy(i,2) needs the y-equation's yflux(i,1) and curv(i,1)
// (:374-391) -- in the reference they arrive with the exchanges of curv, xflux, yflux (:229, :279, :369).  Here the
// neighbour evaluates just those lines (k_advct_edge: its column imm1 / row jmm1, the same formulas on the same
// operands) and sends them east / north in one small message; k_advct_col then runs as on a single tile and
// k_advct_fix redoes advx(2,:) and advy(:,2) with the received values.  Message layout: curv, then the flux,
// each kbm1 x jm (east) or kbm1 x im (north), level-major.
__global__ void k_advct_edge(KP P, double *to_e, double *to_n) {
  const int t = TID_I, k = (int)blockIdx.y + 1;
  if (k > P.kbm1) return;
  if (to_e && t <= P.jm) {
    const size_t o = (size_t)(k - 1) * P.jm + (size_t)(t - 1);
    to_e[o] = advct_curv_mem(P, P.imm1, t, k);
    to_e[(size_t)P.kbm1 * P.jm + o] = t >= 2 ? advct_xf_mem(P, P.imm1, t, k) : 0.;
  }
  if (to_n && t <= P.im) {
    const size_t o = (size_t)(k - 1) * P.im + (size_t)(t - 1);
    to_n[o] = advct_curv_mem(P, t, P.jmm1, k);
    to_n[(size_t)P.kbm1 * P.im + o] = t >= 2 ? advct_yg_mem(P, t, P.jmm1, k) : 0.;
  }
}
__global__ void k_advct_fix(KP P, const double *from_w, const double *from_s) {
  const int t = TID_I, k = (int)blockIdx.y + 1;
  if (k > P.kbm1) return;
  if (from_w && t >= 2 && t <= P.jmm1) {                     // advx(2,j,k), j = t
    const int j = t;
    const size_t o = (size_t)(k - 1) * P.jm + (size_t)(j - 1);
    const double cv_w = from_w[o], xf_w = from_w[(size_t)P.kbm1 * P.jm + o];
    const double xf = advct_xf_mem(P, 2, j, k), yf_n = advct_yf_mem(P, 2, j + 1, k), yf_c = advct_yf_mem(P, 2, j, k);
    const double cv = advct_curv_mem(P, 2, j, k);
    const double ctx = cv * F2(dt, 2, j) * (F3(v, 2, j + 1, k) + F3(v, 2, j, k));
    const double ctx_w = cv_w * F2(dt, 1, j) * (F3(v, 1, j + 1, k) + F3(v, 1, j, k));
    double ax = xf - xf_w + yf_n - yf_c;                                                         // :284-288
    ax = ax - F2(aru, 2, j) * .25 * (ctx + ctx_w);                                               // :291-301
    F3(advx, 2, j, k) = ax;
  }
  if (from_s && t >= 2 && t <= P.imm1) {                     // advy(i,2,k), i = t
    const int i = t;
    const size_t o = (size_t)(k - 1) * P.im + (size_t)(i - 1);
    const double cv_s = from_s[o], yg_s = from_s[(size_t)P.kbm1 * P.im + o];
    const double xg_e = advct_xg_mem(P, i + 1, 2, k), xg = advct_xg_mem(P, i, 2, k), yg_c = advct_yg_mem(P, i, 2, k);
    const double cv = advct_curv_mem(P, i, 2, k);
    double ay = xg_e - xg + yg_c - yg_s;                                                         // :374-378
    ay = ay + F2(arv, i, 2) * .25 * (cv * F2(dt, i, 2) * (F3(u, i + 1, 2, k) + F3(u, i, 2, k)) +
                                     cv_s * F2(dt, i, 1) * (F3(u, i + 1, 1, k) + F3(u, i, 1, k)));     // :381-391
    F3(advy, i, 2, k) = ay;
  }
}
// sum2d: also leave the vertical integrals adx2d, ady2d of advance.f:152-168 (k_vint) -- the column is here anyway
// Operands of a level, all five shared through the workgroup's LDS slab (rows j-1, j, j+1): u, v, ub, vb, aam.
// Loop discipline and row sharing as in k_advt2_col.
// ROWS rows per workgroup: 8 (LDS_ROWS) on large grids; 4 on low tiles, where a launch is only three or four rounds of workgroups and two
// workgroups per compute unit leave the last round fuller (launch_advct_col)
template <int ROWS> struct LevCaT { double c[5], h[(2 * 5 + ROWS - 1) / ROWS]; };
template <int ROWS>
__global__ void __launch_bounds__(64 * ROWS) k_advct_col(KP P, int sum2d) {
  constexpr int NS = 5, NH = (2 * NS + ROWS - 1) / ROWS, U = 0, V = 1, UB = 2, VB = 3, AM = 4;
  typedef LevCaT<ROWS> LevCa;
  HALO_XCD_DECODE_R(ROWS)                                               // i0: 1-based column of this lane (0 for the very first halo lane)
#ifdef POMGPU_WGTIME                                        // developer build (tools/wg_times.py): when does each workgroup run, and where?
  const unsigned long long wgt0 = wall_clock64();
#endif
  const int r = WAVE_UNIFORM((int)threadIdx.y), j0w = j - r;
  const bool jvalid = j <= P.jml;                           // rows beyond the tile shadow row jml and store nothing
  const int jc = jvalid ? j : P.jml;
  const bool out = jvalid && (lane >= 1 && lane <= 62 && i0 <= P.iml);
#ifdef POMGPU_EMU
  if (!out) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const int js = jc > 1 ? jc - 1 : 1, jn = jc < P.jml ? jc + 1 : P.jml;
  const bool jrow = (jc >= 2 && jc <= P.jmm1);
  const bool iin = (i0 >= 2 && i0 <= P.imm1);
  const bool in = out && iin && jrow;
  const int kb = P.kb, kbm1 = P.kbm1;
  double ax2 = 0., ay2 = 0.;
  // column-resident coefficients
  const double dtsx_c = K2(DTSX, i, jc), dtsx_s = K2(DTSX, i, js);
  const double dtsx_e = halo_e(dtsx_c, [&] { return K2(DTSX, ie, jc); });
  const double dtsy_c = K2(DTSY, i, jc), dtsy_n = K2(DTSY, i, jn), dtsy_s = K2(DTSY, i, js);
  const double dtsy_w = halo_w(dtsy_c, [&] { return K2(DTSY, iw, jc); });
  const double dtsy_nw = halo_w(dtsy_n, [&] { return K2(DTSY, iw, jn); });
  const double dt4_c = K2(DT4, i, jc), dt4_n = K2(DT4, i, jn);
  const double dx4_c = K2(DX4, i, jc), dx4_n = K2(DX4, i, jn);
  const InvD dy4_c = inv_of(K2(DY4, i, jc)), dy4_n = inv_of(K2(DY4, i, jn)), idx4_c = inv_of(dx4_c), idx4_n = inv_of(dx4_n);
  const double dt_c = F2(dt, i, jc), dy_c = F2(dy, i, jc), dx_c = F2(dx, i, jc);
  const double dt_s = F2(dt, i, js), dx_s = F2(dx, i, js);
  const InvD idx_c = inv_of(dx_c), idy_c = inv_of(dy_c), idy_s = inv_of(F2(dy, i, js));
  const double cva_c = K2(CVA, i, jc), cvb_c = K2(CVB, i, jc);
  const double cva_s = K2(CVA, i, js), cvb_s = K2(CVB, i, js);
  const InvD art_c = inv_of(F2(art, i, jc)), art_s = inv_of(F2(art, i, js));
  const double aru = F2(aru, i, jc), arv = F2(arv, i, jc);
  const bool srow = (jc - 1 >= 2);                          // row j-1 carries y-eq. fluxes / curv
  const bool curvx = (i0 >= (P.W ? 3 : 2)), curvy = (jc >= (P.S ? 3 : 2));
  BufA bs[NS], bh[NH];
  const double *ps[NS] = {A3(u), A3(v), A3(ub), A3(vb), A3(aam)};
#pragma unroll
  for (int x = 0; x < NS; x++) bs[x] = BUF3(ps[x]);
  const BufA bax = BUF3(A3(advx)), bay = BUF3(A3(advy));
  const RowShare<NS, ROWS> S = rowshare_setup<NS, ROWS>(P, r, j, j0w, i);
#pragma unroll
  for (int q = 0; q < NH; q++) bh[q] = BUF3(rowshare_pick<NS>(ps, S.hop[q]));
  const unsigned oc = BOFF2(i, jc), lvb = LVB;
  const unsigned ost = (out && jrow) ? oc : BOFF_NONE;      // rim rows are zeroed after the loop
#ifndef POMGPU_EMU
  __shared__ double slab[2][NS][ROWS + 3][64];
#else
  const unsigned os = BOFF2(i, js), on = BOFF2(i, jn);
#endif
  auto load = [&](LevCa &L, int k) {
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
#pragma unroll
    for (int x = 0; x < NS; x++) L.c[x] = bld(bs[x], oc, lv);
#pragma unroll
    for (int q = 0; q < NH; q++) L.h[q] = bld(bh[q], S.hoff[q], lv);
  };
  auto step = [&](const int k, const int par, const LevCa &cur, LevCa &nxt) {
    load(nxt, k + 1 <= kbm1 ? k + 1 : kbm1);                // in flight during this iteration (the last one re-requests level kbm1)
    NbrT<NS> nb;
#ifndef POMGPU_EMU
#pragma unroll
    for (int x = 0; x < NS; x++) slab[par][x][r + 1][lane] = cur.c[x];
#pragma unroll
    for (int q = 0; q < NH; q++) slab[par][S.hop[q]][S.hrow[q]][lane] = cur.h[q];
    __syncthreads();
#pragma unroll
    for (int x = 0; x < NS; x++) { nb.s[x] = slab[par][x][S.ss][lane]; nb.n[x] = slab[par][x][S.sn][lane]; }
#else
    const unsigned lvc = (unsigned)(k - 1) * lvb;
#pragma unroll
    for (int x = 0; x < NS; x++) { nb.s[x] = bld(bs[x], os, lvc); nb.n[x] = bld(bs[x], on, lvc); }
#endif
    const double u_c = cur.c[U], v_c = cur.c[V], ub_c = cur.c[UB], vb_c = cur.c[VB], am_c = cur.c[AM];
    const double u_s = nb.s[U], v_s = nb.s[V], ub_s = nb.s[UB], vb_s = nb.s[VB], am_s = nb.s[AM];
    const double u_n = nb.n[U], v_n = nb.n[V], ub_n = nb.n[UB], vb_n = nb.n[VB], am_n = nb.n[AM];
    const double u_e = halo_e(u_c, [&] { return F3(u, ie, jc, k); });
    const double u_se = halo_e(u_s, [&] { return F3(u, ie, js, k); });
    const double ub_e = halo_e(ub_c, [&] { return F3(ub, ie, jc, k); });
    const double v_w = halo_w(v_c, [&] { return F3(v, iw, jc, k); });
    const double v_nw = halo_w(v_n, [&] { return F3(v, iw, jn, k); });
    const double vb_w = halo_w(vb_c, [&] { return F3(vb, iw, jc, k); });
    const double vb_nw = halo_w(vb_n, [&] { return F3(vb, iw, jn, k); });
    const double am_w = halo_w(am_c, [&] { return F3(aam, iw, jc, k); });
    const double am_sw = halo_w(am_s, [&] { return F3(aam, iw, js, k); });
    const double am_nw = halo_w(am_n, [&] { return F3(aam, iw, jn, k); });
    // x-equation xflux at the cell centre (:233-239, :257-262, :275); 0 outside 2..imm1
    double xf = 0., cv = 0.;
    if (iin) {
      xf = .125 * (dtsx_e * u_e + dtsx_c * u_c) * (u_e + u_c);
      xf = xf - divi(dt_c * am_c * 2. * (ub_e - ub_c), idx_c);
      xf = dy_c * xf;
      cv = divi(.25 * ((v_n + v_c) * cva_c - (u_e + u_c) * cvb_c), art_c);                     // :217-227
    }
    const double ctx = cv * dt_c * (v_n + v_c);                                                // :296-297
    const double xf_w = halo_w(xf, [&] { return advct_xf_mem(P, i - 1, jc, k); });
    const double ctx_w = halo_w(ctx, [&] {
      return advct_curv_mem(P, i - 1, jc, k) * F2(dt, i - 1, jc) * (F3(v, i - 1, jc + 1, k) + F3(v, i - 1, jc, k));
    });
    // corner (i,j): y-flux of the x-equation and x-flux of the y-equation share dtaam and the shear bracket
    const double dtaam = .25 * dt4_c * (am_c + am_w + am_s + am_sw);                           // :264-266
    const double br = divi(ub_c - ub_s, dy4_c) + divi(vb_c - vb_w, idx4_c);
    double yf_c = .125 * (dtsy_c * v_c + dtsy_w * v_w) * (u_c + u_s);                          // :244-250
    yf_c = yf_c - dtaam * br;                                                                  // :267-272
    yf_c = .25 * dx4_c * yf_c;                                                                 // :276-277
    double xg = .125 * (dtsx_c * u_c + dtsx_s * u_s) * (v_c + v_w);                            // :322-328
    xg = xg - dtaam * br;                                                                      // :348-353
    xg = .25 * dy4_c.b * xg;                                                                   // :363-364
    const double xg_e = halo_e(xg, [&] { return advct_xg_mem(P, i + 1, jc, k); });
    // corner (i,j+1): y-flux of the x-equation only
    const double dtaam_n = .25 * dt4_n * (am_n + am_nw + am_c + am_w);
    double yf_n = .125 * (dtsy_n * v_n + dtsy_nw * v_nw) * (u_n + u_c);
    yf_n = yf_n - dtaam_n * (divi(ub_n - ub_c, dy4_n) + divi(vb_n - vb_nw, idx4_n));
    yf_n = .25 * dx4_n * yf_n;
    // y-equation yflux at the centres of rows j and j-1 (:333-339, :355-358, :365)
    double yg_c = .125 * (dtsy_n * v_n + dtsy_c * v_c) * (v_n + v_c);
    yg_c = yg_c - divi(dt_c * am_c * 2. * (vb_n - vb_c), idy_c);
    yg_c = dx_c * yg_c;
    double yg_s = 0., cv_s = 0.;
    if (srow) {
      yg_s = .125 * (dtsy_c * v_c + dtsy_s * v_s) * (v_c + v_s);
      yg_s = yg_s - divi(dt_s * am_s * 2. * (vb_c - vb_s), idy_s);
      yg_s = dx_s * yg_s;
      if (iin) cv_s = divi(.25 * ((v_c + v_s) * cva_s - (u_se + u_s) * cvb_s), art_s);
    }
    double ax = 0., ay = 0.;
    if (in) {
      ax = xf - xf_w + yf_n - yf_c;                                                            // :284-288
      if (curvx) ax = ax - aru * .25 * (ctx + ctx_w);                                          // :291-301
      ay = xg_e - xg + yg_c - yg_s;                                                            // :374-378
      if (curvy) ay = ay + arv * .25 * (cv * dt_c * (u_e + u_c) + cv_s * dt_s * (u_se + u_s));   // :381-391
    }
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    bst(bax, ost, lv, ax);
    bst(bay, ost, lv, ay);
    const double dzk = F1(dz, k);
    ax2 = ax2 + ax * dzk;
    ay2 = ay2 + ay * dzk;
  };
  LevCa ra, rb;
#ifdef POMGPU_EMU
  if (jrow) {                                               // the emulation's neighbour fallbacks reach beyond a rim row
#endif
  load(ra, 1);
  rb = ra;
  for (int k = 1; k <= kbm1; k += 2) {
    step(k, 1, ra, rb);
    if (k + 1 <= kbm1) step(k + 1, 0, rb, ra);              // uniform for the whole workgroup (barrier inside)
  }
#ifdef POMGPU_EMU
  }
#endif
  if (out) {
    if (jrow) {
      F3(advx, i, jc, kb) = 0.; F3(advy, i, jc, kb) = 0.;
    } else {                                                // rim rows: advx = advy = 0 (solver.f:211,:317)
      for (int k = 1; k <= kb; k++) { F3(advx, i, jc, k) = 0.; F3(advy, i, jc, k) = 0.; }
    }
    if (sum2d) { F2(adx2d, i, jc) = jrow ? ax2 : 0.; F2(ady2d, i, jc) = jrow ? ay2 : 0.; }
  }
#ifdef POMGPU_WGTIME
  __syncthreads();
  if (threadIdx.y == 0 && lane == 0) {                      // four doubles per workgroup into wr (a diagnostic array nothing reads here)
    double *rec = (double *)(P.b3 + (size_t)P3_wr * P.a3) + 4 * (size_t)L__;
    rec[0] = (double)wgt0; rec[1] = (double)wall_clock64();
    rec[2] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID: wave, simd, pipe, cu, sh, se ...
    rec[3] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
  }
#endif
}

// ---- advu + advv in one pass -- solver.f:734-788, :791-845 --------------------------------------------
// The two leapfrog steps share w, u, v: read together they cost 9 array reads + 2 writes instead of
// 2 x (7 + 1) through the fused solve kernels of k_vert.hip, and the tridiagonal solves that follow
// (k_profuv_reg) then need only their right-hand side and km.  Column-marching, halo-lane wavefronts:
// the level loop carries the vertical fluxes, the (i-1) / (i+1) operands are neighbour-lane values.
// Columns outside the interior get the reference's left-over vertical flux (:744-751, :801-808).
// Operands of iteration k.  Shared with the rows next to this one through the workgroup's LDS slabs: w, u, v of level
// k+1 (c: own row, h: this wavefront's share of the two rows outside the workgroup); own row only, level k: ub, vb, advx,
// advy, drhox, drhoy.  The row south of u and the row north of v are needed at level k -- one level behind what is being
// parked -- so the slabs rotate through THREE buffers: level L lives in slab (L-1) % 3, iteration k fills the slab of
// level k+1 and reads that of level k, and the slab it will overwrite two iterations later is out of everybody's reach
// behind two barriers.  (PMC before: 14.8 array passes of HBM-side traffic for 11 algorithmic -- the three neighbour rows
// were never L2 hits.)
struct LevUV { double c[3], h[ROWSHARE_SLOTS(3)], ub, vb, advx, advy, drhox, drhoy; };
// Same loop discipline as k_advt2_col: one batch of loads per level, issued a whole iteration ahead and never inside a
// branch, stores of lanes without an output column aimed outside the buffer, buffer addressing, two register sets.
__global__ void __launch_bounds__(64 * LDS_ROWS) k_advuv_col(KP P) {
  constexpr int NS = 3, NH = ROWSHARE_SLOTS(NS), W = 0, U = 1, V = 2;
  HALO_XCD_DECODE_R(LDS_ROWS)
  const int r = WAVE_UNIFORM((int)threadIdx.y), j0w = j - r;
  const bool jvalid = j <= P.jm;                            // rows beyond the tile shadow row jm and store nothing
  const int jc = j <= P.jml ? j : P.jml;
  const bool out = jvalid && (lane >= 1 && lane <= 62 && i0 <= P.im);
#ifdef POMGPU_EMU
  if (!out) return;
#endif
  const int i = i0 < 1 ? 1 : (i0 > P.iml ? P.iml : i0);
  const int iw = i > 1 ? i - 1 : 1, ie = i < P.iml ? i + 1 : P.iml;
  const int js = jc > 1 ? jc - 1 : 1, jn = jc < P.jml ? jc + 1 : P.jml;
  const bool in = out && (i >= 2 && i <= P.imm1 && jc >= 2 && jc <= P.jmm1);
  const int kb = P.kb, kbm1 = P.kbm1;
  // column-resident coefficients
  const double aru = F2(aru, i, jc), arv = F2(arv, i, jc);
  const double dt_c = F2(dt, i, jc), dt_w = F2(dt, iw, jc), dt_s = F2(dt, i, js);
  const double cd_c = F2(cor, i, jc) * dt_c, cd_s = F2(cor, i, js) * dt_s;                  // cor*dt of this column and of (i,j-1)
  const double hcu = P.grav * .125 * (dt_c + dt_w) *
                     (F2(egf, i, jc) - F2(egf, iw, jc) + F2(egb, i, jc) - F2(egb, iw, jc) + (F2(e_atmos, i, jc) - F2(e_atmos, iw, jc)) * 2.) *
                     (F2(dy, i, jc) + F2(dy, iw, jc));                                          // :765-770
  const double hcv = P.grav * .125 * (dt_c + dt_s) *
                     (F2(egf, i, jc) - F2(egf, i, js) + F2(egb, i, jc) - F2(egb, i, js) + (F2(e_atmos, i, jc) - F2(e_atmos, i, js)) * 2.) *
                     (F2(dx, i, jc) + F2(dx, i, js));                                          // :822-827
  const double hb = F2(h, i, jc) + F2(etb, i, jc), hf = F2(h, i, jc) + F2(etf, i, jc);
  const double sau = (hb + F2(h, iw, jc) + F2(etb, iw, jc)) * aru;                            // :758
  const double sav = (hb + F2(h, i, js) + F2(etb, i, js)) * arv;                            // :815
  const InvD sdu = inv_of((hf + F2(h, iw, jc) + F2(etf, iw, jc)) * aru), sdv = inv_of((hf + F2(h, i, js) + F2(etf, i, js)) * arv);   // :781, :838
  BufA bs[NS], bh[NH];
  const double *ps[NS] = {A3(w), A3(u), A3(v)};
#pragma unroll
  for (int x = 0; x < NS; x++) bs[x] = BUF3(ps[x]);
  const BufA bub = BUF3(A3(ub)), bvb = BUF3(A3(vb)), bax = BUF3(A3(advx)), bay = BUF3(A3(advy)), bdx = BUF3(A3(drhox)), bdy = BUF3(A3(drhoy)),
             buf = BUF3(A3(uf)), bvf = BUF3(A3(vf));
  const RowShare<NS> S = rowshare_setup<NS>(P, r, j, j0w, i);
#pragma unroll
  for (int q = 0; q < NH; q++) bh[q] = BUF3(rowshare_pick<NS>(ps, S.hop[q]));
  const unsigned oc = BOFF2(i, jc), lvb = LVB;
  const unsigned ost = out ? oc : BOFF_NONE;
#ifndef POMGPU_EMU
  __shared__ double slab[3][NS][ROWSHARE_ROWS][64];
#else
  const unsigned os = BOFF2(i, js), on = BOFF2(i, jn);
#endif
  auto load = [&](LevUV &L, int k) {                        // shared operands of level k+1, own-row operands of level k
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb, lv1 = lv + lvb;
#pragma unroll
    for (int x = 0; x < NS; x++) L.c[x] = bld(bs[x], oc, lv1);
#pragma unroll
    for (int q = 0; q < NH; q++) L.h[q] = bld(bh[q], S.hoff[q], lv1);
    L.ub = bld(bub, oc, lv);  L.vb = bld(bvb, oc, lv);
    L.advx = bld(bax, oc, lv); L.advy = bld(bay, oc, lv); L.drhox = bld(bdx, oc, lv); L.drhoy = bld(bdy, oc, lv);
  };
  auto park = [&](const double (&cv)[3], const double (&hv)[NH], const int sl) {
#ifndef POMGPU_EMU
#pragma unroll
    for (int x = 0; x < NS; x++) slab[sl][x][r + 1][lane] = cv[x];
#pragma unroll
    for (int q = 0; q < NH; q++) slab[sl][S.hop[q]][S.hrow[q]][lane] = hv[q];
#endif
  };
  // level 1 of u, v (and w, unused) into slab 0: what the first iteration reads as its "level k" neighbours
  double u_k, v_k;                                          // u, v of this column at level k
  {
    double c1[3], h1[NH];
#pragma unroll
    for (int x = 0; x < NS; x++) c1[x] = bld(bs[x], oc, 0u);
#pragma unroll
    for (int q = 0; q < NH; q++) h1[q] = bld(bh[q], S.hoff[q], 0u);
    park(c1, h1, 0);
    u_k = c1[U]; v_k = c1[V];
  }
  double fu_k = 0., fv_k = 0.;                              // vertical fluxes at level k (0 at the surface)
  auto step = [&](const int k, const int sl, const LevUV &c, LevUV &nxt) {   // sl = k % 3: the slab of level k+1
    load(nxt, k + 1 <= kbm1 ? k + 1 : kbm1);                // in flight during this iteration (the last one re-requests level kbm1)
    park(c.c, c.h, sl);
    const int slm = sl == 0 ? 2 : sl - 1;                   // the slab of level k
#ifndef POMGPU_EMU
    __syncthreads();
    const double w_s = slab[sl][W][S.ss][lane], u_s = slab[slm][U][S.ss][lane], v_n = slab[slm][V][S.sn][lane];
#else
    const double w_s = bld(bs[W], os, (unsigned)k * lvb), u_s = bld(bs[U], os, (unsigned)(k - 1) * lvb), v_n = bld(bs[V], on, (unsigned)(k - 1) * lvb);
#endif
    const double w_c = c.c[W], u_c = c.c[U], v_c = c.c[V];  // level k+1
    const double w_w = halo_w(w_c, [&] { return F3(w, iw, jc, k + 1); });
    const double tc = cd_c * (v_n + v_k);                                                   // cor*dt*(v(i,j+1,k)+v(i,j,k))
    const double tw = halo_w(tc, [&] { return F2(cor, iw, jc) * F2(dt, iw, jc) * (F3(v, iw, jn, k) + F3(v, iw, jc, k)); });
    const double u_e = halo_e(u_k, [&] { return F3(u, ie, jc, k); });
    const double u_se = halo_e(u_s, [&] { return F3(u, ie, js, k); });
    // vertical fluxes at level k+1 (:744-751, :801-808); zero below kbm1, and where the column has no west / south neighbour
    double fu_n = 0., fv_n = 0.;
    if (k + 1 <= kbm1) {
      if (i >= 2) fu_n = .25 * (w_c + w_w) * (u_c + u_k);
      if (jc >= 2) fv_n = .25 * (w_c + w_s) * (v_c + v_k);
    }
    double ru = fu_k, rv = fv_k;                                                            // outside the interior: the flux itself
    if (in) {
      InvD dzk; dzk.b = F1(dz, k); dzk.y = R1(dz, k);
      ru = divi(sau * c.ub - 2. * P.dti2 * (c.advx + divi((fu_k - fu_n) * aru, dzk) - aru * .25 * (tc + tw) + hcu + c.drhox), sdu);   // :758-782
      rv = divi(sav * c.vb -
                    2. * P.dti2 * (c.advy + divi((fv_k - fv_n) * arv, dzk) + arv * .25 * (cd_c * (u_e + u_k) + cd_s * (u_se + u_s)) + hcv + c.drhoy),
                sdv);                                                                       // :815-839
    }
    const unsigned lv = (unsigned)WAVE_UNIFORM(k - 1) * lvb;
    bst(buf, ost, lv, ru);
    bst(bvf, ost, lv, rv);
    fu_k = fu_n; fv_k = fv_n;
    u_k = u_c; v_k = v_c;
  };
  LevUV ra, rb;
  load(ra, 1);
  rb = ra;
  for (int k = 1; k <= kbm1; k += 6) {                      // six iterations: two register sets x three slabs
    step(k, 1, ra, rb);
    if (k + 1 <= kbm1) step(k + 1, 2, rb, ra);
    if (k + 2 <= kbm1) step(k + 2, 0, ra, rb);
    if (k + 3 <= kbm1) step(k + 3, 1, rb, ra);
    if (k + 4 <= kbm1) step(k + 4, 2, ra, rb);
    if (k + 5 <= kbm1) step(k + 5, 0, rb, ra);
  }
  if (out) { F3(uf, i, jc, kb) = 0.; F3(vf, i, jc, kb) = 0.; }
}

// ---- launchers ------------------------------------------------------------------------------------
void launch_coef_static(pomgpu_ctx *c) { LAUNCH(c, k_coef_static, grid2(c->P), blk2(), c->P); }
void launch_coef_dt(pomgpu_ctx *c) { LAUNCH(c, k_coef_dt, grid2(c->P), blk2(), c->P); }
void launch_coef_eta(pomgpu_ctx *c) { LAUNCH(c, k_coef_eta, grid2(c->P), blk2(), c->P); }
void launch_advq_col(pomgpu_ctx *c, const double *q, const double *qb, double *qf, int zero_else) {
  QFields A; A.q[0] = A.q[1] = q; A.qb[0] = A.qb[1] = qb; A.qf[0] = A.qf[1] = qf;
  LAUNCHN(c, "k_advq_col", (k_advq_col<1>), grid1_halo_r(c->P, LDS_ROWS), blk_col_r(LDS_ROWS), c->P, A, zero_else);
}
void launch_advq2_col(pomgpu_ctx *c, const double *q, const double *qb, double *qf, const double *ql, const double *qlb, double *qlf, int zero_else) {
  QFields A; A.q[0] = q; A.qb[0] = qb; A.qf[0] = qf; A.q[1] = ql; A.qb[1] = qlb; A.qf[1] = qlf;
  LAUNCHN(c, "k_advq2_col", (k_advq_col<2>), grid1_halo_r(c->P, LDS_ROWS), blk_col_r(LDS_ROWS), c->P, A, zero_else);
}
void launch_advct_col(pomgpu_ctx *c, int sum2d) {
  // low tiles (a tile of an 8- or 4-tile split of 2048x1536): 4-row workgroups, two per compute unit
  const bool rows4 = SW(c, ADVCT_ROWS4) || (!SW(c, ADVCT_ROWS8) && c->P.jml <= 400 && c->exch);
  if (rows4) LAUNCHN(c, "k_advct_col", (k_advct_col<4>), grid1_halo_r(c->P, 4), blk_col_r(4), c->P, sum2d);
  else LAUNCHN(c, "k_advct_col", (k_advct_col<LDS_ROWS>), grid1_halo_r(c->P, LDS_ROWS), blk_col_r(LDS_ROWS), c->P, sum2d);
}
// with sum2d the column kernel has left adx2d, ady2d (advance.f:152-168) from its own advx(2,:), advy(:,2): redo the
// two lines from the corrected values, in the column kernel's order of summation
__global__ void k_advct_fix2d(KP P, int west, int south) {
  const int t = TID_I;
  if (west && t >= 2 && t <= P.jmm1) {
    double a = 0.;
#pragma unroll 8
    for (int k = 1; k <= P.kbm1; k++) a = a + F3(advx, 2, t, k) * F1(dz, k);
    F2(adx2d, 2, t) = a;
  }
  if (south && t >= 2 && t <= P.imm1) {
    double a = 0.;
#pragma unroll 8
    for (int k = 1; k <= P.kbm1; k++) a = a + F3(advy, t, 2, k) * F1(dz, k);
    F2(ady2d, t, 2) = a;
  }
}
void launch_advct_fix2d(pomgpu_ctx *c, int west, int south) {
  const KP &P = c->P;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_advct_fix2d, dim3((len + 63) / 64, 1, 1), dim3(64, 1, 1), c->P, west, south);
}
void launch_advct_edge(pomgpu_ctx *c, double *to_e, double *to_n) {
  const KP &P = c->P;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_advct_edge, dim3((len + 63) / 64, P.kbm1, 1), dim3(64, 1, 1), c->P, to_e, to_n);
}
void launch_advct_fix(pomgpu_ctx *c, const double *from_w, const double *from_s) {
  const KP &P = c->P;
  const int len = P.im > P.jm ? P.im : P.jm;
  LAUNCH(c, k_advct_fix, dim3((len + 63) / 64, P.kbm1, 1), dim3(64, 1, 1), c->P, from_w, from_s);
}
void launch_advuv_col(pomgpu_ctx *c) { LAUNCH(c, k_advuv_col, grid1_halo_r(c->P, LDS_ROWS), blk_col_r(LDS_ROWS), c->P); }
void launch_advt2_rows(pomgpu_ctx *c, const double *fb, const double *f, const double *fc, double *ff) {
  {
    TFields A; A.fb[0] = A.fb[1] = fb; A.f[0] = A.f[1] = f; A.fcl[0] = A.fcl[1] = fc; A.ff[0] = A.ff[1] = ff;
    LAUNCHN(c, "k_advt2_col", (k_advt2_col<1>), grid1_halo_r(c->P, LDS_ROWS), blk_col_r(LDS_ROWS), c->P, A);
  }
}
void launch_advt2x2_col(pomgpu_ctx *c, const double *tb, const double *t, const double *tc, double *tf, const double *sb, const double *s_,
                        const double *sc, double *sf) {
  TFields A; A.fb[0] = tb; A.f[0] = t; A.fcl[0] = tc; A.ff[0] = tf; A.fb[1] = sb; A.f[1] = s_; A.fcl[1] = sc; A.ff[1] = sf;
  LAUNCHN(c, "k_advt2x2_col", (k_advt2_col<2>), grid1_halo_r(c->P, LDS_ROWS), blk_col_r(LDS_ROWS), c->P, A);
}
