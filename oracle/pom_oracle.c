/* oracle/pom_oracle.c -- TEST INFRASTRUCTURE ONLY (see pom_oracle.h).
 *
 * Loop-by-loop C restatement of the reference's hot path.  Every function cites the reference
 * lines it follows; loop bounds, statement order, operation order and the (exact-in-binary)
 * single-precision literals are kept, because the pin is bit-identity with the flang-compiled
 * reference (x86-64, no FMA contraction, glibc libm on both sides).
 *
 * Conventions: `T` is the current tile; COMMON arrays are reached through the Fortran-style
 * accessors of pom_fields.h; the reference's automatic arrays (dimensioned (im,jm[,kb]), NOT
 * (im_local,jm_local)) come from T->scr[] through L3_/L2_ below.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pom_oracle.h"
#include "pom_fields.h"

/* scalars of blksiz / blkcon under their reference names */
#define im    (T->im_)
#define jm    (T->jm_)
#define kb    (T->kb_)
#define imm1  (T->im_ - 1)
#define imm2  (T->im_ - 2)
#define jmm1  (T->jm_ - 1)
#define jmm2  (T->jm_ - 2)
#define kbm1  (T->kb_ - 1)
#define kbm2  (T->kb_ - 2)
#define im_local (T->iml)
#define jm_local (T->jml)
#define n_west  (T->nw_)
#define n_east  (T->ne_)
#define n_south (T->ns_)
#define n_north (T->nn_)
#define CON_(x) (T->con->x)
#define alpha CON_(alpha)
#define dte CON_(dte)
#define dti CON_(dti)
#define dti2 CON_(dti2)
#define dte2 CON_(dte2)
#define grav CON_(grav)
#define kappa CON_(kappa)
#define ramp CON_(ramp)
#define rfe CON_(rfe)
#define rfn CON_(rfn)
#define rfs CON_(rfs)
#define rfw CON_(rfw)
#define rhoref CON_(rhoref)
#define sbias CON_(sbias)
#define small CON_(small)
#define tbias CON_(tbias)
#define tprni CON_(tprni)
#define umol CON_(umol)
#define vmaxl CON_(vmaxl)
#define horcon CON_(horcon)
#define ispi CON_(ispi)
#define isp2i CON_(isp2i)
#define smoth CON_(smoth)
#define sw CON_(sw)
#define time0 CON_(time0)
#define period CON_(period)
#define mode CON_(mode)
#define ntp CON_(ntp)
#define nadv CON_(nadv)
#define nbct CON_(nbct)
#define nbcs CON_(nbcs)
#define nitera CON_(nitera)
#define npg CON_(npg)
#define isplit CON_(isplit)
#define ispadv CON_(ispadv)
#define iext CON_(iext)
#define iint CON_(iint)
#define iend CON_(iend)
#define error_status CON_(error_status)

/* automatic arrays dimensioned (im,jm,kb) / (im,jm) */
#define L3_(p,i,j,k) (p)[((size_t)((k)-1)*(size_t)jm + (size_t)((j)-1))*(size_t)im + (size_t)((i)-1)]
#define L2_(p,i,j)   (p)[(size_t)((j)-1)*(size_t)im + (size_t)((i)-1)]
/* dummy-argument arrays dimensioned (im_local,jm_local,kb) */
#define G3_(p,i,j,k) (p)[IX3_(i,j,k)]
#define G2_(p,i,j)   (p)[IX2_(i,j)]

static inline double sq(double x) { return x * x; }

static double *zero3(pomo_tile *T, int n) {       /* "x = 0." for an (im,jm,kb) automatic */
  memset(T->scr[n], 0, sizeof(double) * (size_t)im * jm * kb);
  return T->scr[n];
}
static void X2(pomo_tile *T, double *a, int nx, int ny) { if (T->exch2d) T->exch2d(T->user, a, nx, ny); }
static void X3(pomo_tile *T, double *a, int nx, int ny, int nz) { if (T->exch3d) T->exch3d(T->user, a, nx, ny, nz); }

size_t pomo_tile_size(void) { return sizeof(pomo_tile); }

int pomo_bind(pomo_tile *T, int aim, int ajm, int akb, int iml_, int jml_, pom_blkcon *con,
              double *b1, double *b2, double *b3, double *bdry) {
  const int kb_ = akb;
  memset(T, 0, sizeof *T);
  im = aim; jm = ajm; kb = akb; T->iml = iml_; T->jml = jml_;
  T->n2 = (size_t)iml_ * jml_; T->n3 = T->n2 * kb_;
  n_west = n_east = n_south = n_north = -1;
  T->con = con; T->blk1d = b1; T->blk2d = b2; T->blk3d = b3; T->bdry = bdry;
  size_t off = 0; int s = 0;
#define BD_(name, shape) T->bd[s++] = bdry + off; off += BDN_##shape;
#define BDN_J  ((size_t)jml_)
#define BDN_I  ((size_t)iml_)
#define BDN_JK ((size_t)jml_ * kb_)
#define BDN_IK ((size_t)iml_ * kb_)
  POM_BDRY(BD_)
#undef BD_
  for (int n = 0; n < POMO_NSCR; n++) {
    T->scr[n] = (double *)calloc(T->n3 + 64, sizeof(double));
    if (!T->scr[n]) return -1;
  }
  return 0;
}
void pomo_set_order(pomo_tile *T, pomo_order_fn fn) { T->order = fn; }
void pomo_release(pomo_tile *T) { for (int n = 0; n < POMO_NSCR; n++) { free(T->scr[n]); T->scr[n] = NULL; } }

/* ===================================================================================== */
/* advave -- solver.f:6-198 */
void pomo_advave(pomo_tile *T) {
  int i, j;
  double *curv2d = T->scr[0];
  memset(A2_(advua), 0, sizeof(double) * T->n2);          /* :16-18 */
  memset(A2_(fluxua), 0, sizeof(double) * T->n2);
  memset(A2_(fluxva), 0, sizeof(double) * T->n2);
  for (j = 2; j <= jm; j++) for (i = 2; i <= imm1; i++)   /* :20-26 */
    fluxua(i,j) = .125*((d(i+1,j)+d(i,j))*ua(i+1,j) + (d(i,j)+d(i-1,j))*ua(i,j)) * (ua(i+1,j)+ua(i,j));
  for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)     /* :28-34 */
    fluxva(i,j) = .125*((d(i,j)+d(i,j-1))*va(i,j) + (d(i-1,j)+d(i-1,j-1))*va(i-1,j)) * (ua(i,j)+ua(i,j-1));
  for (j = 2; j <= jm; j++) for (i = 2; i <= imm1; i++)   /* :37-43 */
    fluxua(i,j) = fluxua(i,j) - d(i,j)*2.*aam2d(i,j)*(uab(i+1,j)-uab(i,j))/dx(i,j);
  for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {   /* :45-58 */
    tps(i,j) = .25*(d(i,j)+d(i-1,j)+d(i,j-1)+d(i-1,j-1))
               *(aam2d(i,j)+aam2d(i,j-1)+aam2d(i-1,j)+aam2d(i-1,j-1))
               *((uab(i,j)-uab(i,j-1))/(dy(i,j)+dy(i-1,j)+dy(i,j-1)+dy(i-1,j-1))
                +(vab(i,j)-vab(i-1,j))/(dx(i,j)+dx(i-1,j)+dx(i,j-1)+dx(i-1,j-1)));
    fluxua(i,j) = fluxua(i,j)*dy(i,j);
    fluxva(i,j) = (fluxva(i,j)-tps(i,j))*.25*(dx(i,j)+dx(i-1,j)+dx(i,j-1)+dx(i-1,j-1));
  }
  X2(T, A2_(fluxua), im_local, jm_local);                 /* :60-61 */
  X2(T, A2_(fluxva), im_local, jm_local);
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) /* :63-68 */
    advua(i,j) = fluxua(i,j)-fluxua(i-1,j)+fluxva(i,j+1)-fluxva(i,j);
  X2(T, A2_(advua), im_local, jm_local);                  /* :70 */

  memset(A2_(advva), 0, sizeof(double) * T->n2);          /* :73-75 */
  memset(A2_(fluxua), 0, sizeof(double) * T->n2);
  memset(A2_(fluxva), 0, sizeof(double) * T->n2);
  for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)     /* :78-84 */
    fluxua(i,j) = .125*((d(i,j)+d(i-1,j))*ua(i,j) + (d(i,j-1)+d(i-1,j-1))*ua(i,j-1)) * (va(i-1,j)+va(i,j));
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= im; i++)   /* :86-92 */
    fluxva(i,j) = .125*((d(i,j+1)+d(i,j))*va(i,j+1) + (d(i,j)+d(i,j-1))*va(i,j)) * (va(i,j+1)+va(i,j));
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= im; i++)   /* :95-101 */
    fluxva(i,j) = fluxva(i,j) - d(i,j)*2.*aam2d(i,j)*(vab(i,j+1)-vab(i,j))/dy(i,j);
  for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {   /* :103-109 */
    fluxva(i,j) = fluxva(i,j)*dx(i,j);
    fluxua(i,j) = (fluxua(i,j)-tps(i,j))*.25*(dy(i,j)+dy(i-1,j)+dy(i,j-1)+dy(i-1,j-1));
  }
  X2(T, A2_(fluxua), im_local, jm_local);                 /* :111-112 */
  X2(T, A2_(fluxva), im_local, jm_local);
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) /* :114-119 */
    advva(i,j) = fluxua(i+1,j)-fluxua(i,j)+fluxva(i,j)-fluxva(i,j-1);
  X2(T, A2_(advva), im_local, jm_local);                  /* :121 */

  if (mode == 2) {                                        /* :123-195 */
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)
      wubot(i,j) = -0.5*(cbc(i,j)+cbc(i-1,j))
                   *sqrt(sq(uab(i,j)) + sq(.25*(vab(i,j)+vab(i,j+1)+vab(i-1,j)+vab(i-1,j+1))))*uab(i,j);
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)
      wvbot(i,j) = -0.5*(cbc(i,j)+cbc(i,j-1))
                   *sqrt(sq(vab(i,j)) + sq(.25*(uab(i,j)+uab(i+1,j)+uab(i,j-1)+uab(i+1,j-1))))*vab(i,j);
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)
      L2_(curv2d,i,j) = .25*((va(i,j+1)+va(i,j))*(dy(i+1,j)-dy(i-1,j))
                            -(ua(i+1,j)+ua(i,j))*(dx(i,j+1)-dx(i,j-1)))/(dx(i,j)*dy(i,j));
    X2(T, curv2d, im, jm);
    for (j = 2; j <= jmm1; j++)
      for (i = (n_west == -1 ? 3 : 2); i <= imm1; i++)
        advua(i,j) = advua(i,j)-aru(i,j)*.25
                     *(L2_(curv2d,i,j)*d(i,j)*(va(i,j+1)+va(i,j))
                      +L2_(curv2d,i-1,j)*d(i-1,j)*(va(i-1,j+1)+va(i-1,j)));
    for (i = 2; i <= imm1; i++)
      for (j = (n_south == -1 ? 3 : 2); j <= jmm1; j++)
        advva(i,j) = advva(i,j)+arv(i,j)*.25
                     *(L2_(curv2d,i,j)*d(i,j)*(ua(i+1,j)+ua(i,j))
                      +L2_(curv2d,i,j-1)*d(i,j-1)*(ua(i+1,j-1)+ua(i,j-1)));
  }
}

/* ===================================================================================== */
/* advct -- solver.f:201-408 */
void pomo_advct(pomo_tile *T) {
  int i, j, k;
  double dtaam;
  double *curv = zero3(T, 0);                              /* :213-216 */
  memset(A3_(advx), 0, sizeof(double) * T->n3);
  double *xflux = zero3(T, 1), *yflux = zero3(T, 2);
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :218-228 */
    L3_(curv,i,j,k) = .25*((v(i,j+1,k)+v(i,j,k))*(dy(i+1,j)-dy(i-1,j))
                          -(u(i+1,j,k)+u(i,j,k))*(dx(i,j+1)-dx(i,j-1)))/(dx(i,j)*dy(i,j));
  X3(T, curv, im, jm, kbm1);                                /* :229 */
  for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 2; i <= imm1; i++)     /* :234-242 */
    L3_(xflux,i,j,k) = .125*((dt(i+1,j)+dt(i,j))*u(i+1,j,k)+(dt(i,j)+dt(i-1,j))*u(i,j,k))*(u(i+1,j,k)+u(i,j,k));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)       /* :244-252 */
    L3_(yflux,i,j,k) = .125*((dt(i,j)+dt(i,j-1))*v(i,j,k)+(dt(i-1,j)+dt(i-1,j-1))*v(i-1,j,k))*(u(i,j,k)+u(i,j-1,k));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= imm1; i++) {   /* :255-277 */
    L3_(xflux,i,j,k) = L3_(xflux,i,j,k) - dt(i,j)*aam(i,j,k)*2.*(ub(i+1,j,k)-ub(i,j,k))/dx(i,j);
    dtaam = .25*(dt(i,j)+dt(i-1,j)+dt(i,j-1)+dt(i-1,j-1))*(aam(i,j,k)+aam(i-1,j,k)+aam(i,j-1,k)+aam(i-1,j-1,k));
    L3_(yflux,i,j,k) = L3_(yflux,i,j,k)
        - dtaam*((ub(i,j,k)-ub(i,j-1,k))/(dy(i,j)+dy(i-1,j)+dy(i,j-1)+dy(i-1,j-1))
                +(vb(i,j,k)-vb(i-1,j,k))/(dx(i,j)+dx(i-1,j)+dx(i,j-1)+dx(i-1,j-1)));
    L3_(xflux,i,j,k) = dy(i,j)*L3_(xflux,i,j,k);
    L3_(yflux,i,j,k) = .25*(dx(i,j)+dx(i-1,j)+dx(i,j-1)+dx(i-1,j-1))*L3_(yflux,i,j,k);
  }
  X3(T, xflux, im, jm, kbm1);                               /* :279 */
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :282-289 */
    advx(i,j,k) = L3_(xflux,i,j,k)-L3_(xflux,i-1,j,k)+L3_(yflux,i,j+1,k)-L3_(yflux,i,j,k);
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++)                               /* :291-313 */
    for (i = (n_west == -1 ? 3 : 2); i <= imm1; i++)
      advx(i,j,k) = advx(i,j,k) - aru(i,j)*.25
                    *(L3_(curv,i,j,k)*dt(i,j)*(v(i,j+1,k)+v(i,j,k))
                     +L3_(curv,i-1,j,k)*dt(i-1,j)*(v(i-1,j+1,k)+v(i-1,j,k)));
  X3(T, A3_(advx), im_local, jm_local, kb);                 /* :315 */

  memset(A3_(advy), 0, sizeof(double) * T->n3);             /* :319-321 */
  xflux = zero3(T, 1); yflux = zero3(T, 2);
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)       /* :324-332 */
    L3_(xflux,i,j,k) = .125*((dt(i,j)+dt(i-1,j))*u(i,j,k)+(dt(i,j-1)+dt(i-1,j-1))*u(i,j-1,k))*(v(i,j,k)+v(i-1,j,k));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 1; i <= im; i++)     /* :334-342 */
    L3_(yflux,i,j,k) = .125*((dt(i,j+1)+dt(i,j))*v(i,j+1,k)+(dt(i,j)+dt(i,j-1))*v(i,j,k))*(v(i,j+1,k)+v(i,j,k));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= im; i++) {   /* :345-367 */
    dtaam = .25*(dt(i,j)+dt(i-1,j)+dt(i,j-1)+dt(i-1,j-1))*(aam(i,j,k)+aam(i-1,j,k)+aam(i,j-1,k)+aam(i-1,j-1,k));
    L3_(xflux,i,j,k) = L3_(xflux,i,j,k)
        - dtaam*((ub(i,j,k)-ub(i,j-1,k))/(dy(i,j)+dy(i-1,j)+dy(i,j-1)+dy(i-1,j-1))
                +(vb(i,j,k)-vb(i-1,j,k))/(dx(i,j)+dx(i-1,j)+dx(i,j-1)+dx(i-1,j-1)));
    L3_(yflux,i,j,k) = L3_(yflux,i,j,k) - dt(i,j)*aam(i,j,k)*2.*(vb(i,j+1,k)-vb(i,j,k))/dy(i,j);
    L3_(xflux,i,j,k) = .25*(dy(i,j)+dy(i-1,j)+dy(i,j-1)+dy(i-1,j-1))*L3_(xflux,i,j,k);
    L3_(yflux,i,j,k) = dx(i,j)*L3_(yflux,i,j,k);
  }
  X3(T, yflux, im, jm, kbm1);                               /* :369 */
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :372-379 */
    advy(i,j,k) = L3_(xflux,i+1,j,k)-L3_(xflux,i,j,k)+L3_(yflux,i,j,k)-L3_(yflux,i,j-1,k);
  for (k = 1; k <= kbm1; k++) for (i = 2; i <= imm1; i++)                               /* :381-403 */
    for (j = (n_south == -1 ? 3 : 2); j <= jmm1; j++)
      advy(i,j,k) = advy(i,j,k) + arv(i,j)*.25
                    *(L3_(curv,i,j,k)*dt(i,j)*(u(i+1,j,k)+u(i,j,k))
                     +L3_(curv,i,j-1,k)*dt(i,j-1)*(u(i+1,j-1,k)+u(i,j-1,k)));
  X3(T, A3_(advy), im_local, jm_local, kb);                 /* :405 */
}

/* ===================================================================================== */
/* advq(qb,q,qf) -- solver.f:411-477 */
void pomo_advq(pomo_tile *T, double *qb, double *q, double *qf) {
  int i, j, k;
  double *xflux = zero3(T, 0), *yflux = zero3(T, 1);        /* :421-422 */
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {     /* :425-434 */
    L3_(xflux,i,j,k) = .125*(G3_(q,i,j,k)+G3_(q,i-1,j,k))*(dt(i,j)+dt(i-1,j))*(u(i,j,k)+u(i,j,k-1));
    L3_(yflux,i,j,k) = .125*(G3_(q,i,j,k)+G3_(q,i,j-1,k))*(dt(i,j)+dt(i,j-1))*(v(i,j,k)+v(i,j,k-1));
  }
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {     /* :437-456 */
    L3_(xflux,i,j,k) = L3_(xflux,i,j,k)
        - .25*(aam(i,j,k)+aam(i-1,j,k)+aam(i,j,k-1)+aam(i-1,j,k-1))*(h(i,j)+h(i-1,j))
          *(G3_(qb,i,j,k)-G3_(qb,i-1,j,k))*dum(i,j)/(dx(i,j)+dx(i-1,j));
    L3_(yflux,i,j,k) = L3_(yflux,i,j,k)
        - .25*(aam(i,j,k)+aam(i,j-1,k)+aam(i,j,k-1)+aam(i,j-1,k-1))*(h(i,j)+h(i,j-1))
          *(G3_(qb,i,j,k)-G3_(qb,i,j-1,k))*dvm(i,j)/(dy(i,j)+dy(i,j-1));
    L3_(xflux,i,j,k) = .5*(dy(i,j)+dy(i-1,j))*L3_(xflux,i,j,k);
    L3_(yflux,i,j,k) = .5*(dx(i,j)+dx(i,j-1))*L3_(yflux,i,j,k);
  }
  X3(T, xflux, im, jm, kbm1);                               /* :458-459 */
  X3(T, yflux, im, jm, kbm1);
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) { /* :462-474 */
    G3_(qf,i,j,k) = (w(i,j,k-1)*G3_(q,i,j,k-1)-w(i,j,k+1)*G3_(q,i,j,k+1))*art(i,j)/(dz(k)+dz(k-1))
                    +L3_(xflux,i+1,j,k)-L3_(xflux,i,j,k)+L3_(yflux,i,j+1,k)-L3_(yflux,i,j,k);
    G3_(qf,i,j,k) = ((h(i,j)+etb(i,j))*art(i,j)*G3_(qb,i,j,k)-dti2*G3_(qf,i,j,k))/((h(i,j)+etf(i,j))*art(i,j));
  }
}

/* ===================================================================================== */
/* advt1(fb,f,fclim,ff) -- solver.f:480-574 */
void pomo_advt1(pomo_tile *T, double *fb, double *f, double *fclim, double *ff) {
  int i, j, k; size_t n;
  double *xflux = zero3(T, 0), *yflux = zero3(T, 1);        /* :492-493 */
  for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++) {                     /* :495-496 */
    G3_(f,i,j,kb) = G3_(f,i,j,kbm1);
    G3_(fb,i,j,kb) = G3_(fb,i,j,kbm1);
  }
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {     /* :499-508 */
    L3_(xflux,i,j,k) = .25*((dt(i,j)+dt(i-1,j))*(G3_(f,i,j,k)+G3_(f,i-1,j,k))*u(i,j,k));
    L3_(yflux,i,j,k) = .25*((dt(i,j)+dt(i,j-1))*(G3_(f,i,j,k)+G3_(f,i,j-1,k))*v(i,j,k));
  }
  for (n = 0; n < T->n3; n++) fb[n] = fb[n]-fclim[n];       /* :511 */
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {     /* :513-530 */
    L3_(xflux,i,j,k) = L3_(xflux,i,j,k)
        - .5*(aam(i,j,k)+aam(i-1,j,k))*(h(i,j)+h(i-1,j))*tprni
          *(G3_(fb,i,j,k)-G3_(fb,i-1,j,k))*dum(i,j)/(dx(i,j)+dx(i-1,j));
    L3_(yflux,i,j,k) = L3_(yflux,i,j,k)
        - .5*(aam(i,j,k)+aam(i,j-1,k))*(h(i,j)+h(i,j-1))*tprni
          *(G3_(fb,i,j,k)-G3_(fb,i,j-1,k))*dvm(i,j)/(dy(i,j)+dy(i,j-1));
    L3_(xflux,i,j,k) = .5*(dy(i,j)+dy(i-1,j))*L3_(xflux,i,j,k);
    L3_(yflux,i,j,k) = .5*(dx(i,j)+dx(i,j-1))*L3_(yflux,i,j,k);
  }
  for (n = 0; n < T->n3; n++) fb[n] = fb[n]+fclim[n];       /* :532 */
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {                             /* :535-540 */
    zflux(i,j,1) = G3_(f,i,j,1)*w(i,j,1)*art(i,j);
    zflux(i,j,kb) = 0.;
  }
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :542-548 */
    zflux(i,j,k) = .5*(G3_(f,i,j,k-1)+G3_(f,i,j,k))*w(i,j,k)*art(i,j);
  for (k = 1; k <= kbm1; k++) {                                                         /* :562-571 */
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)
      G3_(ff,i,j,k) = L3_(xflux,i+1,j,k)-L3_(xflux,i,j,k)+L3_(yflux,i,j+1,k)-L3_(yflux,i,j,k)
                      +(zflux(i,j,k)-zflux(i,j,k+1))/dz(k);
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)
      G3_(ff,i,j,k) = (G3_(fb,i,j,k)*(h(i,j)+etb(i,j))*art(i,j)-dti2*G3_(ff,i,j,k))/((h(i,j)+etf(i,j))*art(i,j));
  }
}

/* ===================================================================================== */
/* smol_adif(xmassflux,ymassflux,zwflux,ff) -- solver.f:1880-1967 */
static void smol_adif(pomo_tile *T, double *xmassflux, double *ymassflux, double *zwflux, double *ff) {
  int i, j, k;
  double mol, abs_1, abs_2, udx, u2dt, vdy, v2dt, wdz, w2dt;
  const double value_min = 1.e-9, epsilon = 1.0e-14;
  for (k = 1; k <= kb; k++) for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++)  /* :1898-1900 */
    G3_(ff,i,j,k) = G3_(ff,i,j,k)*fsm(i,j);
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= im; i++) {       /* :1903-1922 */
    if (G3_(ff,i,j,k) < value_min || G3_(ff,i-1,j,k) < value_min) {
      L3_(xmassflux,i,j,k) = 0.;
    } else {
      udx = fabs(L3_(xmassflux,i,j,k));
      u2dt = dti2*L3_(xmassflux,i,j,k)*L3_(xmassflux,i,j,k)*2./(aru(i,j)*(dt(i-1,j)+dt(i,j)));
      mol = (G3_(ff,i,j,k)-G3_(ff,i-1,j,k))/(G3_(ff,i-1,j,k)+G3_(ff,i,j,k)+epsilon);
      L3_(xmassflux,i,j,k) = (udx-u2dt)*mol*sw;
      abs_1 = fabs(udx); abs_2 = fabs(u2dt);
      if (abs_1 < abs_2) L3_(xmassflux,i,j,k) = 0.;
    }
  }
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= imm1; i++) {       /* :1924-1943 */
    if (G3_(ff,i,j,k) < value_min || G3_(ff,i,j-1,k) < value_min) {
      L3_(ymassflux,i,j,k) = 0.;
    } else {
      vdy = fabs(L3_(ymassflux,i,j,k));
      v2dt = dti2*L3_(ymassflux,i,j,k)*L3_(ymassflux,i,j,k)*2./(arv(i,j)*(dt(i,j-1)+dt(i,j)));
      mol = (G3_(ff,i,j,k)-G3_(ff,i,j-1,k))/(G3_(ff,i,j-1,k)+G3_(ff,i,j,k)+epsilon);
      L3_(ymassflux,i,j,k) = (vdy-v2dt)*mol*sw;
      abs_1 = fabs(vdy); abs_2 = fabs(v2dt);
      if (abs_1 < abs_2) L3_(ymassflux,i,j,k) = 0.;
    }
  }
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {     /* :1945-1964 */
    if (G3_(ff,i,j,k) < value_min || G3_(ff,i,j,k-1) < value_min) {
      L3_(zwflux,i,j,k) = 0.;
    } else {
      wdz = fabs(L3_(zwflux,i,j,k));
      w2dt = dti2*L3_(zwflux,i,j,k)*L3_(zwflux,i,j,k)/(dzz(k-1)*dt(i,j));
      mol = (G3_(ff,i,j,k-1)-G3_(ff,i,j,k))/(G3_(ff,i,j,k)+G3_(ff,i,j,k-1)+epsilon);
      L3_(zwflux,i,j,k) = (wdz-w2dt)*mol*sw;
      abs_1 = fabs(wdz); abs_2 = fabs(w2dt);
      if (abs_1 < abs_2) L3_(zwflux,i,j,k) = 0.;
    }
  }
}

/* ===================================================================================== */
/* advt2(fb,f,fclim,ff) -- solver.f:577-731 */
void pomo_advt2(pomo_tile *T, double *fb, double *f, double *fclim, double *ff) {
  int i, j, k, itera; size_t n;
  double *xflux = zero3(T, 0), *yflux = zero3(T, 1);        /* :597-600 */
  double *xmassflux = zero3(T, 2), *ymassflux = zero3(T, 3);
  double *zwflux = T->scr[4], *fbmem = T->scr[5], *eta = T->scr[6];
  for (k = 1; k <= kbm1; k++) {                                                         /* :602-616 */
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= im; i++)
      L3_(xmassflux,i,j,k) = 0.25*(dy(i-1,j)+dy(i,j))*(dt(i-1,j)+dt(i,j))*u(i,j,k);
    for (j = 2; j <= jm; j++) for (i = 2; i <= imm1; i++)
      L3_(ymassflux,i,j,k) = 0.25*(dx(i,j-1)+dx(i,j))*(dt(i,j-1)+dt(i,j))*v(i,j,k);
  }
  for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++) G3_(fb,i,j,kb) = G3_(fb,i,j,kbm1);   /* :618 */
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) L2_(eta,i,j) = etb(i,j);          /* :619 */
  for (k = 1; k <= kb; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {       /* :621-622 */
    L3_(zwflux,i,j,k) = w(i,j,k);
    L3_(fbmem,i,j,k) = G3_(fb,i,j,k);
  }
  for (itera = 1; itera <= nitera; itera++) {                                           /* :625-688 */
    for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {   /* :628-644 */
      L3_(xflux,i,j,k) = 0.5*((L3_(xmassflux,i,j,k)+fabs(L3_(xmassflux,i,j,k)))*L3_(fbmem,i-1,j,k)
                             +(L3_(xmassflux,i,j,k)-fabs(L3_(xmassflux,i,j,k)))*L3_(fbmem,i,j,k));
      L3_(yflux,i,j,k) = 0.5*((L3_(ymassflux,i,j,k)+fabs(L3_(ymassflux,i,j,k)))*L3_(fbmem,i,j-1,k)
                             +(L3_(ymassflux,i,j,k)-fabs(L3_(ymassflux,i,j,k)))*L3_(fbmem,i,j,k));
    }
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) zflux(i,j,1) = 0.;          /* :646 */
    if (itera == 1)                                                                     /* :647-650 */
      for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) zflux(i,j,1) = w(i,j,1)*G3_(f,i,j,1)*art(i,j);
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) zflux(i,j,kb) = 0.;         /* :651 */
    for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {   /* :653-664 */
      zflux(i,j,k) = 0.5*((L3_(zwflux,i,j,k)+fabs(L3_(zwflux,i,j,k)))*L3_(fbmem,i,j,k)
                         +(L3_(zwflux,i,j,k)-fabs(L3_(zwflux,i,j,k)))*L3_(fbmem,i,j,k-1));
      zflux(i,j,k) = zflux(i,j,k)*art(i,j);
    }
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) for (k = 1; k <= kbm1; k++) {   /* :667-677 */
      G3_(ff,i,j,k) = L3_(xflux,i+1,j,k)-L3_(xflux,i,j,k)+L3_(yflux,i,j+1,k)-L3_(yflux,i,j,k)
                      +(zflux(i,j,k)-zflux(i,j,k+1))/dz(k);
      G3_(ff,i,j,k) = (L3_(fbmem,i,j,k)*((h(i,j)+L2_(eta,i,j))*art(i,j))-dti2*G3_(ff,i,j,k))
                      /((h(i,j)+etf(i,j))*art(i,j));
    }
    X3(T, ff, im_local, jm_local, kbm1);                                                /* :679 */
    smol_adif(T, xmassflux, ymassflux, zwflux, ff);                                     /* :682 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) L2_(eta,i,j) = etf(i,j);        /* :684 */
    for (k = 1; k <= kb; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)       /* :685 */
      L3_(fbmem,i,j,k) = G3_(ff,i,j,k);
  }
  for (n = 0; n < T->n3; n++) fb[n] = fb[n]-fclim[n];                                   /* :691 */
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {     /* :693-700 */
    L3_(xmassflux,i,j,k) = 0.5*(aam(i,j,k)+aam(i-1,j,k));
    L3_(ymassflux,i,j,k) = 0.5*(aam(i,j,k)+aam(i,j-1,k));
  }
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {     /* :702-713 */
    L3_(xflux,i,j,k) = -L3_(xmassflux,i,j,k)*(h(i,j)+h(i-1,j))*tprni
                       *(G3_(fb,i,j,k)-G3_(fb,i-1,j,k))*dum(i,j)*(dy(i,j)+dy(i-1,j))*0.5/(dx(i,j)+dx(i-1,j));
    L3_(yflux,i,j,k) = -L3_(ymassflux,i,j,k)*(h(i,j)+h(i,j-1))*tprni
                       *(G3_(fb,i,j,k)-G3_(fb,i,j-1,k))*dvm(i,j)*(dx(i,j)+dx(i,j-1))*0.5/(dy(i,j)+dy(i,j-1));
  }
  for (n = 0; n < T->n3; n++) fb[n] = fb[n]+fclim[n];                                   /* :715 */
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) for (k = 1; k <= kbm1; k++)   /* :718-726 */
    G3_(ff,i,j,k) = G3_(ff,i,j,k)-dti2*(L3_(xflux,i+1,j,k)-L3_(xflux,i,j,k)+L3_(yflux,i,j+1,k)-L3_(yflux,i,j,k))
                                  /((h(i,j)+etf(i,j))*art(i,j));
  X3(T, ff, im_local, jm_local, kbm1);                                                  /* :728 */
}

/* ===================================================================================== */
/* advu -- solver.f:734-788 */
void pomo_advu(pomo_tile *T) {
  int i, j, k;
  memset(A3_(uf), 0, sizeof(double) * T->n3);                                           /* :742 */
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 2; i <= im; i++)       /* :744-751 */
    uf(i,j,k) = .25*(w(i,j,k)+w(i-1,j,k))*(u(i,j,k)+u(i,j,k-1));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :755-772 */
    uf(i,j,k) = advx(i,j,k)
                +(uf(i,j,k)-uf(i,j,k+1))*aru(i,j)/dz(k)
                -aru(i,j)*.25*(cor(i,j)*dt(i,j)*(v(i,j+1,k)+v(i,j,k))
                              +cor(i-1,j)*dt(i-1,j)*(v(i-1,j+1,k)+v(i-1,j,k)))
                +grav*.125*(dt(i,j)+dt(i-1,j))
                 *(egf(i,j)-egf(i-1,j)+egb(i,j)-egb(i-1,j)+(e_atmos(i,j)-e_atmos(i-1,j))*2.)
                 *(dy(i,j)+dy(i-1,j))
                +drhox(i,j,k);
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :775-785 */
    uf(i,j,k) = ((h(i,j)+etb(i,j)+h(i-1,j)+etb(i-1,j))*aru(i,j)*ub(i,j,k)-2.*dti2*uf(i,j,k))
                /((h(i,j)+etf(i,j)+h(i-1,j)+etf(i-1,j))*aru(i,j));
}

/* advv -- solver.f:791-845 */
void pomo_advv(pomo_tile *T) {
  int i, j, k;
  memset(A3_(vf), 0, sizeof(double) * T->n3);                                           /* :799 */
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 1; i <= im; i++)       /* :801-808 */
    vf(i,j,k) = .25*(w(i,j,k)+w(i,j-1,k))*(v(i,j,k)+v(i,j,k-1));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :812-829 */
    vf(i,j,k) = advy(i,j,k)
                +(vf(i,j,k)-vf(i,j,k+1))*arv(i,j)/dz(k)
                +arv(i,j)*.25*(cor(i,j)*dt(i,j)*(u(i+1,j,k)+u(i,j,k))
                              +cor(i,j-1)*dt(i,j-1)*(u(i+1,j-1,k)+u(i,j-1,k)))
                +grav*.125*(dt(i,j)+dt(i,j-1))
                 *(egf(i,j)-egf(i,j-1)+egb(i,j)-egb(i,j-1)+(e_atmos(i,j)-e_atmos(i,j-1))*2.)
                 *(dx(i,j)+dx(i,j-1))
                +drhoy(i,j,k);
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :832-842 */
    vf(i,j,k) = ((h(i,j)+etb(i,j)+h(i,j-1)+etb(i,j-1))*arv(i,j)*vb(i,j,k)-2.*dti2*vf(i,j,k))
                /((h(i,j)+etf(i,j)+h(i,j-1)+etf(i,j-1))*arv(i,j));
}

/* ===================================================================================== */
/* baropg -- solver.f:848-940 */
void pomo_baropg(pomo_tile *T) {
  int i, j, k; size_t n;
  double *prho = A3_(rho), *prm = A3_(rmean);
  for (n = 0; n < T->n3; n++) prho[n] = prho[n]-prm[n];                                 /* :854 */
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)                               /* :857-862 */
    drhox(i,j,1) = .5*grav*(-zz(1))*(dt(i,j)+dt(i-1,j))*(rho(i,j,1)-rho(i-1,j,1));
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :864-878 */
    drhox(i,j,k) = drhox(i,j,k-1)
                   +grav*.25*(zz(k-1)-zz(k))*(dt(i,j)+dt(i-1,j))
                    *(rho(i,j,k)-rho(i-1,j,k)+rho(i,j,k-1)-rho(i-1,j,k-1))
                   +grav*.25*(zz(k-1)+zz(k))*(dt(i,j)-dt(i-1,j))
                    *(rho(i,j,k)+rho(i-1,j,k)-rho(i,j,k-1)-rho(i-1,j,k-1));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :880-888 */
    drhox(i,j,k) = .25*(dt(i,j)+dt(i-1,j))*drhox(i,j,k)*dum(i,j)*(dy(i,j)+dy(i-1,j));
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)                               /* :893-898 */
    drhoy(i,j,1) = .5*grav*(-zz(1))*(dt(i,j)+dt(i,j-1))*(rho(i,j,1)-rho(i,j-1,1));
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :900-914 */
    drhoy(i,j,k) = drhoy(i,j,k-1)
                   +grav*.25*(zz(k-1)-zz(k))*(dt(i,j)+dt(i,j-1))
                    *(rho(i,j,k)-rho(i,j-1,k)+rho(i,j,k-1)-rho(i,j-1,k-1))
                   +grav*.25*(zz(k-1)+zz(k))*(dt(i,j)-dt(i,j-1))
                    *(rho(i,j,k)+rho(i,j-1,k)-rho(i,j,k-1)-rho(i,j-1,k-1));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :916-924 */
    drhoy(i,j,k) = .25*(dt(i,j)+dt(i,j-1))*drhoy(i,j,k)*dvm(i,j)*(dx(i,j)+dx(i,j-1));
  for (k = 1; k <= kb; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {   /* :928-935 */
    drhox(i,j,k) = ramp*drhox(i,j,k);
    drhoy(i,j,k) = ramp*drhoy(i,j,k);
  }
  for (n = 0; n < T->n3; n++) prho[n] = prho[n]+prm[n];                                 /* :937 */
}

/* ===================================================================================== */
/* baropg_mcc -- solver.f:943-1159: baroclinic pressure gradient with McCalpin's 4th-order
 * correction (npg = 2).  The (0:im_local,0:jm_local) work arrays rho4th, d4th differ from rho, d
 * only in their extra west column / south row, which order2d_mpi / order3d_mpi fill from the
 * neighbour (column nx-2 / row ny-2 of ITS array); they are read only where i-2 = 0 or j-2 = 0.
 * (1./24.), (1./24), (1./16.) are REAL(4) constants. */
void pomo_baropg_mcc(pomo_tile *T) {
  int i, j, k; size_t n;
  const double c24 = (double)(1.f/24.f), c16 = (double)(1.f/16.f);
  double *prho = A3_(rho), *prm = A3_(rmean);
  double *d4 = T->scr[0], *ddx = T->scr[1], *drho = T->scr[2], *rhou = T->scr[3];
  double *gw3 = T->scr[4], *gs3 = T->scr[5], *gw2 = T->scr[6], *gs2 = T->scr[7];   /* rho4th(0,j,k), rho4th(i,0,k), d4th(0,j), d4th(i,0) */
#define RHO4W(j,k) gw3[(size_t)((k)-1)*jm_local+(size_t)((j)-1)]
#define RHO4S(i,k) gs3[(size_t)((k)-1)*im_local+(size_t)((i)-1)]
  for (n = 0; n < T->n3; n++) prho[n] = prho[n]-prm[n];                                 /* :954 */
  memset(gw3, 0, sizeof(double)*(size_t)jm_local*kb); memset(gs3, 0, sizeof(double)*(size_t)im_local*kb);
  memset(gw2, 0, sizeof(double)*(size_t)jm_local);    memset(gs2, 0, sizeof(double)*(size_t)im_local);
  if (T->order) {                                                                       /* :958-959 */
    T->order(T->user, A2_(d), im_local, jm_local, 1, gw2, gs2);
    T->order(T->user, prho, im_local, jm_local, kb, gw3, gs3);
  }
  /* ---- x component */
  memset(ddx, 0, sizeof(double)*(size_t)im*jm); memset(d4, 0, sizeof(double)*(size_t)im*jm);   /* :962-965 */
  zero3(T, 2); zero3(T, 3);
  for (j = 1; j <= jm; j++) for (i = 2; i <= im; i++) {                                 /* :968-978 */
    for (k = 1; k <= kbm1; k++) {
      L3_(drho,i,j,k) = (rho(i,j,k)-rho(i-1,j,k))*dum(i,j);
      L3_(rhou,i,j,k) = 0.5*(rho(i,j,k)+rho(i-1,j,k))*dum(i,j);
    }
    L2_(ddx,i,j) = (d(i,j)-d(i-1,j))*dum(i,j);
    L2_(d4,i,j) = .5*(d(i,j)+d(i-1,j))*dum(i,j);
  }
  {
    const int i0 = (n_west == -1) ? 3 : 2;                                              /* :980-1024 */
    for (j = 1; j <= jm; j++) for (i = i0; i <= imm1; i++) {
#define RW2(k) ((i) >= 3 ? rho(i-2,j,k) : RHO4W(j,k))
      const double dw2 = (i >= 3) ? d(i-2,j) : gw2[j-1];
      for (k = 1; k <= kbm1; k++) {
        L3_(drho,i,j,k) = L3_(drho,i,j,k) - c24*
                          (dum(i+1,j)*(rho(i+1,j,k)-rho(i,j,k))-
                           2*(rho(i,j,k)-rho(i-1,j,k))+
                           dum(i-1,j)*(rho(i-1,j,k)-RW2(k)));
        L3_(rhou,i,j,k) = L3_(rhou,i,j,k) + c16*
                          (dum(i+1,j)*(rho(i,j,k)-rho(i+1,j,k))+
                           dum(i-1,j)*(rho(i-1,j,k)-RW2(k)));
      }
      L2_(ddx,i,j) = L2_(ddx,i,j)-c24*
                     (dum(i+1,j)*(d(i+1,j)-d(i,j))-
                      2*(d(i,j)-d(i-1,j))+
                      dum(i-1,j)*(d(i-1,j)-dw2));
      L2_(d4,i,j) = L2_(d4,i,j)+c16*
                    (dum(i+1,j)*(d(i,j)-d(i+1,j))+
                     dum(i-1,j)*(d(i-1,j)-dw2));
#undef RW2
    }
  }
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)                               /* :1027-1031 */
    drhox(i,j,1) = grav*(-zz(1))*L2_(d4,i,j)*L3_(drho,i,j,1);
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :1033-1044 */
    drhox(i,j,k) = drhox(i,j,k-1)
                   +grav*0.5*dzz(k-1)*L2_(d4,i,j)
                    *(L3_(drho,i,j,k-1)+L3_(drho,i,j,k))
                   +grav*0.5*(zz(k-1)+zz(k))*L2_(ddx,i,j)
                    *(L3_(rhou,i,j,k)-L3_(rhou,i,j,k-1));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :1046-1054 */
    drhox(i,j,k) = .25*(dt(i,j)+dt(i-1,j))
                      *drhox(i,j,k)*dum(i,j)
                      *(dy(i,j)+dy(i-1,j));
  /* ---- y component */
  memset(ddx, 0, sizeof(double)*(size_t)im*jm); memset(d4, 0, sizeof(double)*(size_t)im*jm);   /* :1059-1062 */
  zero3(T, 2); zero3(T, 3);
  for (j = 2; j <= jm; j++) for (i = 1; i <= im; i++) {                                 /* :1065-1075 */
    for (k = 1; k <= kbm1; k++) {
      L3_(drho,i,j,k) = (rho(i,j,k)-rho(i,j-1,k))*dvm(i,j);
      L3_(rhou,i,j,k) = .5*(rho(i,j,k)+rho(i,j-1,k))*dvm(i,j);
    }
    L2_(ddx,i,j) = (d(i,j)-d(i,j-1))*dvm(i,j);
    L2_(d4,i,j) = .5*(d(i,j)+d(i,j-1))*dvm(i,j);
  }
  {
    const int j0 = (n_south == -1) ? 3 : 2;                                             /* :1077-1121 */
    for (j = j0; j <= jmm1; j++) for (i = 1; i <= im; i++) {
#define RS2(k) ((j) >= 3 ? rho(i,j-2,k) : RHO4S(i,k))
      const double ds2 = (j >= 3) ? d(i,j-2) : gs2[i-1];
      for (k = 1; k <= kbm1; k++) {
        L3_(drho,i,j,k) = L3_(drho,i,j,k)-c24*
                          (dvm(i,j+1)*(rho(i,j+1,k)-rho(i,j,k))-
                           2*(rho(i,j,k)-rho(i,j-1,k))+
                           dvm(i,j-1)*(rho(i,j-1,k)-RS2(k)));
        L3_(rhou,i,j,k) = L3_(rhou,i,j,k)+c16*
                          (dvm(i,j+1)*(rho(i,j,k)-rho(i,j+1,k))+
                           dvm(i,j-1)*(rho(i,j-1,k)-RS2(k)));
      }
      L2_(ddx,i,j) = L2_(ddx,i,j)-c24*
                     (dvm(i,j+1)*(d(i,j+1)-d(i,j))-
                      2*(d(i,j)-d(i,j-1))+
                      dvm(i,j-1)*(d(i,j-1)-ds2));
      L2_(d4,i,j) = L2_(d4,i,j)+c16*
                    (dvm(i,j+1)*(d(i,j)-d(i,j+1))+
                     dvm(i,j-1)*(d(i,j-1)-ds2));
#undef RS2
    }
  }
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)                               /* :1124-1128 */
    drhoy(i,j,1) = grav*(-zz(1))*L2_(d4,i,j)*L3_(drho,i,j,1);
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :1130-1141 */
    drhoy(i,j,k) = drhoy(i,j,k-1)
                   +grav*0.5*dzz(k-1)*L2_(d4,i,j)
                    *(L3_(drho,i,j,k-1)+L3_(drho,i,j,k))
                   +grav*0.5*(zz(k-1)+zz(k))*L2_(ddx,i,j)
                    *(L3_(rhou,i,j,k)-L3_(rhou,i,j,k-1));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :1143-1151 */
    drhoy(i,j,k) = .25*(dt(i,j)+dt(i,j-1))
                      *drhoy(i,j,k)*dvm(i,j)
                      *(dx(i,j)+dx(i,j-1));
  for (k = 1; k <= kb; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {   /* :1155-1162 */
    drhox(i,j,k) = ramp*drhox(i,j,k);
    drhoy(i,j,k) = ramp*drhoy(i,j,k);
  }
  for (n = 0; n < T->n3; n++) prho[n] = prho[n]+prm[n];                                 /* :1164 */
#undef RHO4W
#undef RHO4S
}

/* ===================================================================================== */
/* dens(si,ti,rhoo) -- solver.f:1162-1209 */
void pomo_dens(pomo_tile *T, double *si, double *ti, double *rhoo) {
  int i, j, k;
  double cr, p, rhor, sr, tr, tr2, tr3, tr4;
  for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
    tr = G3_(ti,i,j,k)+tbias;
    sr = G3_(si,i,j,k)+sbias;
    tr2 = tr*tr; tr3 = tr2*tr; tr4 = tr3*tr;
    p = grav*rhoref*(-zz(k)*h(i,j))*1.e-5;                                              /* :1185 */
    rhor = -0.157406+6.793952e-2*tr-9.095290e-3*tr2+1.001685e-4*tr3-1.120083e-6*tr4+6.536332e-9*tr4*tr;
    rhor = rhor+(0.824493-4.0899e-3*tr+7.6438e-5*tr2-8.2467e-7*tr3+5.3875e-9*tr4)*sr
               +(-5.72466e-3+1.0227e-4*tr-1.6546e-6*tr2)*pow(fabs(sr), 1.5)
               +4.8314e-4*sr*sr;                                                        /* :1191-1196 */
    cr = 1449.1+.0821*p+4.55*tr-.045*tr2+1.34*(sr-35.);
    rhor = rhor+1.e5*p/(cr*cr)*(1.-2.*p/(cr*cr));                                       /* :1200 */
    G3_(rhoo,i,j,k) = rhor/rhoref*fsm(i,j);
  }
}

/* ===================================================================================== */
/* profq -- solver.f:1212-1538 */
void pomo_profq(pomo_tile *T) {
  int i, j, k, ki;
  const double a1 = 0.92, b1 = 16.6, a2 = 0.74, b2 = 10.1, c1 = 0.08;                   /* :1241-1244 */
  const double e1 = 1.8, e2 = 1.33, sef = 1., cbcnst = 100., surfl = 2.e5, shiw = 0.;
  double coef1, coef2, coef3, coef4, coef5, const1, ghc, p, sp, tp;
  double *a = zero3(T, 0), *c = zero3(T, 1), *ee = zero3(T, 2), *gg = zero3(T, 3);      /* :1252-1255 */
  double *sm = T->scr[4], *sh = T->scr[5], *cc = T->scr[6], *gh = T->scr[7];
  double *boygr = T->scr[8], *stf = T->scr[9], *prod = T->scr[10];
  double *dh = T->scr[11], *l0 = T->scr[12], *utau2 = T->scr[13];
  size_t n2a = (size_t)im * jm;
  (void)ghc; (void)e1;
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) L2_(dh,i,j) = h(i,j)+etf(i,j);    /* :1246-1250 */
  memset(utau2, 0, sizeof(double) * n2a);                                               /* :1256 */
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1258-1267 */
    L3_(a,i,j,k) = -dti2*(kq(i,j,k+1)+kq(i,j,k)+2.*umol)*.5/(dzz(k-1)*dz(k)*L2_(dh,i,j)*L2_(dh,i,j));
    L3_(c,i,j,k) = -dti2*(kq(i,j,k-1)+kq(i,j,k)+2.*umol)*.5/(dzz(k-1)*dz(k-1)*L2_(dh,i,j)*L2_(dh,i,j));
  }
  const1 = pow(16.6, 2./3.)*sef;                                                        /* :1273 */
  memset(l0, 0, sizeof(double) * n2a);                                                  /* :1277-1279 */
  boygr = zero3(T, 8); prod = zero3(T, 10);
  for (j = 1; j <= jmm1; j++) for (i = 1; i <= imm1; i++) {                             /* :1281-1288 */
    L2_(utau2,i,j) = sqrt(sq(.5*(wusurf(i,j)+wusurf(i+1,j)))+sq(.5*(wvsurf(i,j)+wvsurf(i,j+1))));
    uf(i,j,kb) = sqrt(sq(.5*(wubot(i,j)+wubot(i+1,j)))+sq(.5*(wvbot(i,j)+wvbot(i,j+1))))*const1;
  }
  X2(T, utau2, im, jm);                                                                 /* :1289-1290 */
  X2(T, &uf(1,1,kb), im_local, jm_local);
  {
    /* (15.8*cbcnst)**(2./3.) with REAL(4) literals 15.8 and 2./3. (:1297) */
    const double cb = pow((double)15.8f*cbcnst, (double)(2.f/3.f));
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {                               /* :1292-1301 */
      L3_(ee,i,j,1) = 0.;
      L3_(gg,i,j,1) = cb*L2_(utau2,i,j);
      L2_(l0,i,j) = surfl*L2_(utau2,i,j)/grav;
    }
  }
  cc = zero3(T, 6);                                                                     /* :1304 */
  for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1305-1319 */
    tp = t(i,j,k)+tbias;
    sp = s(i,j,k)+sbias;
    p = grav*rhoref*(-zz(k)*h(i,j))*1.e-4;
    L3_(cc,i,j,k) = 1449.1+.00821*p+4.55*tp-.045*sq(tp)+1.34*(sp-35.0);
    L3_(cc,i,j,k) = L3_(cc,i,j,k)/sqrt((1.-.01642*p/L3_(cc,i,j,k))*(1.-0.40*p/sq(L3_(cc,i,j,k))));
  }
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1322-1333 */
    q2b(i,j,k) = fabs(q2b(i,j,k));
    q2lb(i,j,k) = fabs(q2lb(i,j,k));
    L3_(boygr,i,j,k) = grav*(rho(i,j,k-1)-rho(i,j,k))/(dzz(k-1)*h(i,j))
                       +sq(grav)*2./(sq(L3_(cc,i,j,k-1))+sq(L3_(cc,i,j,k)));
  }
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1335-1347 */
    l(i,j,k) = fabs(q2lb(i,j,k)/q2b(i,j,k));
    if (z(k) > -0.5) l(i,j,k) = fmax(l(i,j,k), kappa*L2_(l0,i,j));
    L3_(gh,i,j,k) = sq(l(i,j,k))*L3_(boygr,i,j,k)/q2b(i,j,k);
    L3_(gh,i,j,k) = fmin(L3_(gh,i,j,k), .028);
  }
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {                                 /* :1349-1356 */
    l(i,j,1) = kappa*L2_(l0,i,j);
    l(i,j,kb) = 0.;
    L3_(gh,i,j,1) = 0.;
    L3_(gh,i,j,kb) = 0.;
  }
  for (k = 2; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) { /* :1359-1373 */
    L3_(prod,i,j,k) = km(i,j,k)*.25*sef
                      *(sq(u(i,j,k)-u(i,j,k-1)+u(i+1,j,k)-u(i+1,j,k-1))
                       +sq(v(i,j,k)-v(i,j,k-1)+v(i,j+1,k)-v(i,j+1,k-1)))
                      /sq(dzz(k-1)*L2_(dh,i,j))
                      -shiw*km(i,j,k)*L3_(boygr,i,j,k);
    L3_(prod,i,j,k) = L3_(prod,i,j,k)+kh(i,j,k)*L3_(boygr,i,j,k);
  }
  X3(T, &L3_(prod,1,1,2), im, jm, kbm2);                                                /* :1374 */
  ghc = -6.0;
  for (k = 1; k <= kb; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {       /* :1380-1392 */
    L3_(stf,i,j,k) = 1.;
    dtef(i,j,k) = sqrt(fabs(q2b(i,j,k)))*L3_(stf,i,j,k)/(b1*l(i,j,k)+small);
  }
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1394-1404 */
    L3_(gg,i,j,k) = 1./(L3_(a,i,j,k)+L3_(c,i,j,k)*(1.-L3_(ee,i,j,k-1))-(2.*dti2*dtef(i,j,k)+1.));
    L3_(ee,i,j,k) = L3_(a,i,j,k)*L3_(gg,i,j,k);
    L3_(gg,i,j,k) = (-2.*dti2*L3_(prod,i,j,k)+L3_(c,i,j,k)*L3_(gg,i,j,k-1)-uf(i,j,k))*L3_(gg,i,j,k);
  }
  for (k = 1; k <= kbm1; k++) {                                                         /* :1406-1413 */
    ki = kb-k;
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)
      uf(i,j,ki) = L3_(ee,i,j,ki)*uf(i,j,ki+1)+L3_(gg,i,j,ki);
  }
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {                                 /* :1417-1425 */
    vf(i,j,1) = 0.;
    vf(i,j,kb) = 0.;
    L3_(ee,i,j,2) = 0.;
    L3_(gg,i,j,2) = -kappa*z(2)*L2_(dh,i,j)*q2(i,j,2);
    vf(i,j,kb-1) = kappa*(1+z(kbm1))*L2_(dh,i,j)*q2(i,j,kbm1);
  }
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)       /* :1426-1435 */
    dtef(i,j,k) = dtef(i,j,k)
                  *(1.+e2*sq((1./fabs(z(k)-z(1))+1./fabs(z(k)-z(kb)))*l(i,j,k)/(L2_(dh,i,j)*kappa)));
  for (k = 3; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1436-1446 */
    L3_(gg,i,j,k) = 1./(L3_(a,i,j,k)+L3_(c,i,j,k)*(1.-L3_(ee,i,j,k-1))-(dti2*dtef(i,j,k)+1.));
    L3_(ee,i,j,k) = L3_(a,i,j,k)*L3_(gg,i,j,k);
    L3_(gg,i,j,k) = (dti2*(-L3_(prod,i,j,k)*l(i,j,k)*e1)+L3_(c,i,j,k)*L3_(gg,i,j,k-1)-vf(i,j,k))*L3_(gg,i,j,k);
  }
  for (k = 1; k <= kb-2; k++) {                                                         /* :1448-1455 */
    ki = kb-k;
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)
      vf(i,j,ki) = L3_(ee,i,j,ki)*vf(i,j,ki+1)+L3_(gg,i,j,ki);
  }
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1460-1471 */
    uf(i,j,k) = fabs(uf(i,j,k));
    vf(i,j,k) = fabs(vf(i,j,k));
  }
  coef4 = 18.*a1*a1+9.*a1*a2;                                                           /* :1474-1475 */
  coef5 = 9.*a1*a2;
  for (k = 1; k <= kb; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {       /* :1478-1489 */
    coef1 = a2*(1.-6.*a1/b1*L3_(stf,i,j,k));
    coef2 = 3.*a2*b2/L3_(stf,i,j,k)+18.*a1*a2;
    coef3 = a1*(1.-3.*c1-6.*a1/b1*L3_(stf,i,j,k));
    L3_(sh,i,j,k) = coef1/(1.-coef2*L3_(gh,i,j,k));
    L3_(sm,i,j,k) = coef3+L3_(sh,i,j,k)*coef4*L3_(gh,i,j,k);
    L3_(sm,i,j,k) = L3_(sm,i,j,k)/(1.-coef5*L3_(gh,i,j,k));
  }
  for (k = 1; k <= kb; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {       /* :1496-1506 */
    L3_(prod,i,j,k) = l(i,j,k)*sqrt(fabs(q2(i,j,k)));
    kq(i,j,k) = (L3_(prod,i,j,k)*.41*L3_(sh,i,j,k)+kq(i,j,k))*.5;
    km(i,j,k) = (L3_(prod,i,j,k)*L3_(sm,i,j,k)+km(i,j,k))*.5;
    kh(i,j,k) = (L3_(prod,i,j,k)*L3_(sh,i,j,k)+kh(i,j,k))*.5;
  }
  if (n_north == -1) for (k = 1; k <= kb; k++) for (i = 1; i <= im_local; i++) {        /* :1510-1514 */
    km(i,jm,k) = km(i,jmm1,k); kh(i,jm,k) = kh(i,jmm1,k); kq(i,jm,k) = kq(i,jmm1,k); }
  if (n_south == -1) for (k = 1; k <= kb; k++) for (i = 1; i <= im_local; i++) {        /* :1515-1519 */
    km(i,1,k) = km(i,2,k); kh(i,1,k) = kh(i,2,k); kq(i,1,k) = kq(i,2,k); }
  if (n_east == -1) for (k = 1; k <= kb; k++) for (j = 1; j <= jm_local; j++) {         /* :1520-1524 */
    km(im,j,k) = km(imm1,j,k); kh(im,j,k) = kh(imm1,j,k); kq(im,j,k) = kq(imm1,j,k); }
  if (n_west == -1) for (k = 1; k <= kb; k++) for (j = 1; j <= jm_local; j++) {         /* :1525-1529 */
    km(1,j,k) = km(2,j,k); kh(1,j,k) = kh(2,j,k); kq(1,j,k) = kq(2,j,k); }
  for (k = 1; k <= kb; k++) for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++) {   /* :1531-1535 */
    km(i,j,k) = km(i,j,k)*fsm(i,j);
    kh(i,j,k) = kh(i,j,k)*fsm(i,j);
    kq(i,j,k) = kq(i,j,k)*fsm(i,j);
  }
}

/* ===================================================================================== */
/* proft(f,wfsurf,fsurf,nbc) -- solver.f:1541-1683 */
void pomo_proft(pomo_tile *T, double *f, double *wfsurf, double *fsurf, int nbc) {
  int i, j, k, ki;
  static const double r[5]   = { .58, .62, .67, .77, .78 };                             /* :1561-1563 */
  static const double ad1[5] = { .35, .60, 1.0, 1.5, 1.4 };
  static const double ad2[5] = { 23., 20., 17., 14., 7.9 };
  double *a = zero3(T, 0), *c = zero3(T, 1), *ee = zero3(T, 2), *gg = zero3(T, 3);      /* :1584-1587 */
  double *dh = T->scr[11], *rad;
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) L2_(dh,i,j) = h(i,j)+etf(i,j);    /* :1578-1582 */
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1589-1598 */
    L3_(a,i,j,k-1) = -dti2*(kh(i,j,k)+umol)/(dz(k-1)*dzz(k-1)*L2_(dh,i,j)*L2_(dh,i,j));
    L3_(c,i,j,k) = -dti2*(kh(i,j,k)+umol)/(dz(k)*dzz(k-1)*L2_(dh,i,j)*L2_(dh,i,j));
  }
  rad = zero3(T, 4);                                                                    /* :1602 */
  if (nbc == 2 || nbc == 4) {                                                           /* :1604-1615 */
    /* the reference evaluates this expression in REAL(16) and rounds once to REAL(8) */
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
      __float128 e1q = (__float128)(z(k)*L2_(dh,i,j)/ad1[ntp-1]);
      __float128 e2q = (__float128)(z(k)*L2_(dh,i,j)/ad2[ntp-1]);
      extern __float128 expq(__float128);
      __float128 v = (__float128)swrad(i,j)*((__float128)r[ntp-1]*expq(e1q)+(__float128)(1.-r[ntp-1])*expq(e2q));
      L3_(rad,i,j,k) = (double)v;
    }
  }
  if (nbc == 1) {                                                                       /* :1617-1625 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
      L3_(ee,i,j,1) = L3_(a,i,j,1)/(L3_(a,i,j,1)-1.);
      L3_(gg,i,j,1) = dti2*G2_(wfsurf,i,j)/(dz(1)*L2_(dh,i,j))-G3_(f,i,j,1);
      L3_(gg,i,j,1) = L3_(gg,i,j,1)/(L3_(a,i,j,1)-1.);
    }
  } else if (nbc == 2) {                                                                /* :1627-1637 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
      L3_(ee,i,j,1) = L3_(a,i,j,1)/(L3_(a,i,j,1)-1.);
      L3_(gg,i,j,1) = dti2*(G2_(wfsurf,i,j)+L3_(rad,i,j,1)-L3_(rad,i,j,2))/(dz(1)*L2_(dh,i,j))-G3_(f,i,j,1);
      L3_(gg,i,j,1) = L3_(gg,i,j,1)/(L3_(a,i,j,1)-1.);
    }
  } else if (nbc == 3 || nbc == 4) {                                                    /* :1639-1646 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
      L3_(ee,i,j,1) = 0.;
      L3_(gg,i,j,1) = G2_(fsurf,i,j);
    }
  }
  for (k = 2; k <= kbm2; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1650-1661 */
    L3_(gg,i,j,k) = 1./(L3_(a,i,j,k)+L3_(c,i,j,k)*(1.-L3_(ee,i,j,k-1))-1.);
    L3_(ee,i,j,k) = L3_(a,i,j,k)*L3_(gg,i,j,k);
    L3_(gg,i,j,k) = (L3_(c,i,j,k)*L3_(gg,i,j,k-1)-G3_(f,i,j,k)
                     +dti2*(L3_(rad,i,j,k)-L3_(rad,i,j,k+1))/(L2_(dh,i,j)*dz(k)))*L3_(gg,i,j,k);
  }
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)                                   /* :1664-1671 */
    G3_(f,i,j,kbm1) = (L3_(c,i,j,kbm1)*L3_(gg,i,j,kbm2)-G3_(f,i,j,kbm1)
                       +dti2*(L3_(rad,i,j,kbm1)-L3_(rad,i,j,kb))/(L2_(dh,i,j)*dz(kbm1)))
                      /(L3_(c,i,j,kbm1)*(1.-L3_(ee,i,j,kbm2))-1.);
  for (k = 2; k <= kbm1; k++) {                                                         /* :1673-1680 */
    ki = kb-k;
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)
      G3_(f,i,j,ki) = (L3_(ee,i,j,ki)*G3_(f,i,j,ki+1)+L3_(gg,i,j,ki));
  }
}

/* ===================================================================================== */
/* profu -- solver.f:1686-1780 */
void pomo_profu(pomo_tile *T) {
  int i, j, k, ki;
  double *a = zero3(T, 0), *c = zero3(T, 1), *ee = zero3(T, 2), *gg = zero3(T, 3);      /* :1707-1710 */
  double *dh = T->scr[11];
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) L2_(dh,i,j) = 1.;                 /* :1699 */
  for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)                                   /* :1701-1705 */
    L2_(dh,i,j) = (h(i,j)+etf(i,j)+h(i-1,j)+etf(i-1,j))*.5;
  for (k = 1; k <= kb; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)         /* :1712-1718 */
    L3_(c,i,j,k) = (km(i,j,k)+km(i-1,j,k))*.5;
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1720-1729 */
    L3_(a,i,j,k-1) = -dti2*(L3_(c,i,j,k)+umol)/(dz(k-1)*dzz(k-1)*L2_(dh,i,j)*L2_(dh,i,j));
    L3_(c,i,j,k) = -dti2*(L3_(c,i,j,k)+umol)/(dz(k)*dzz(k-1)*L2_(dh,i,j)*L2_(dh,i,j));
  }
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {                                 /* :1731-1738 */
    L3_(ee,i,j,1) = L3_(a,i,j,1)/(L3_(a,i,j,1)-1.);
    L3_(gg,i,j,1) = (-dti2*wusurf(i,j)/(-dz(1)*L2_(dh,i,j))-uf(i,j,1))/(L3_(a,i,j,1)-1.);
  }
  for (k = 2; k <= kbm2; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1740-1748 */
    L3_(gg,i,j,k) = 1./(L3_(a,i,j,k)+L3_(c,i,j,k)*(1.-L3_(ee,i,j,k-1))-1.);
    L3_(ee,i,j,k) = L3_(a,i,j,k)*L3_(gg,i,j,k);
    L3_(gg,i,j,k) = (L3_(c,i,j,k)*L3_(gg,i,j,k-1)-uf(i,j,k))*L3_(gg,i,j,k);
  }
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {                             /* :1750-1761 */
    tps(i,j) = 0.5*(cbc(i,j)+cbc(i-1,j))
               *sqrt(sq(ub(i,j,kbm1))+sq(.25*(vb(i,j,kbm1)+vb(i,j+1,kbm1)+vb(i-1,j,kbm1)+vb(i-1,j+1,kbm1))));
    uf(i,j,kbm1) = (L3_(c,i,j,kbm1)*L3_(gg,i,j,kbm2)-uf(i,j,kbm1))
                   /(tps(i,j)*dti2/(-dz(kbm1)*L2_(dh,i,j))-1.-(L3_(ee,i,j,kbm2)-1.)*L3_(c,i,j,kbm1));
    uf(i,j,kbm1) = uf(i,j,kbm1)*dum(i,j);
  }
  for (k = 2; k <= kbm1; k++) {                                                         /* :1763-1770 */
    ki = kb-k;
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)
      uf(i,j,ki) = (L3_(ee,i,j,ki)*uf(i,j,ki+1)+L3_(gg,i,j,ki))*dum(i,j);
  }
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) wubot(i,j) = -tps(i,j)*uf(i,j,kbm1);   /* :1772-1776 */
  X2(T, A2_(wubot), im_local, jm_local);                                                /* :1777 */
}

/* profv -- solver.f:1783-1877 */
void pomo_profv(pomo_tile *T) {
  int i, j, k, ki;
  double *a = zero3(T, 0), *c = zero3(T, 1), *ee = zero3(T, 2), *gg = zero3(T, 3);      /* :1805-1808 */
  double *dh = T->scr[11];
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) L2_(dh,i,j) = 1.;                 /* :1797 */
  for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)                                   /* :1799-1803 */
    L2_(dh,i,j) = .5*(h(i,j)+etf(i,j)+h(i,j-1)+etf(i,j-1));
  for (k = 1; k <= kb; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)         /* :1810-1816 */
    L3_(c,i,j,k) = (km(i,j,k)+km(i,j-1,k))*.5;
  for (k = 2; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1818-1827 */
    L3_(a,i,j,k-1) = -dti2*(L3_(c,i,j,k)+umol)/(dz(k-1)*dzz(k-1)*L2_(dh,i,j)*L2_(dh,i,j));
    L3_(c,i,j,k) = -dti2*(L3_(c,i,j,k)+umol)/(dz(k)*dzz(k-1)*L2_(dh,i,j)*L2_(dh,i,j));
  }
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {                                 /* :1829-1835 */
    L3_(ee,i,j,1) = L3_(a,i,j,1)/(L3_(a,i,j,1)-1.);
    L3_(gg,i,j,1) = (-dti2*wvsurf(i,j)/(-dz(1)*L2_(dh,i,j))-vf(i,j,1))/(L3_(a,i,j,1)-1.);
  }
  for (k = 2; k <= kbm2; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {     /* :1837-1845 */
    L3_(gg,i,j,k) = 1./(L3_(a,i,j,k)+L3_(c,i,j,k)*(1.-L3_(ee,i,j,k-1))-1.);
    L3_(ee,i,j,k) = L3_(a,i,j,k)*L3_(gg,i,j,k);
    L3_(gg,i,j,k) = (L3_(c,i,j,k)*L3_(gg,i,j,k-1)-vf(i,j,k))*L3_(gg,i,j,k);
  }
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {                             /* :1847-1858 */
    tps(i,j) = 0.5*(cbc(i,j)+cbc(i,j-1))
               *sqrt(sq(.25*(ub(i,j,kbm1)+ub(i+1,j,kbm1)+ub(i,j-1,kbm1)+ub(i+1,j-1,kbm1)))+sq(vb(i,j,kbm1)));
    vf(i,j,kbm1) = (L3_(c,i,j,kbm1)*L3_(gg,i,j,kbm2)-vf(i,j,kbm1))
                   /(tps(i,j)*dti2/(-dz(kbm1)*L2_(dh,i,j))-1.-(L3_(ee,i,j,kbm2)-1.)*L3_(c,i,j,kbm1));
    vf(i,j,kbm1) = vf(i,j,kbm1)*dvm(i,j);
  }
  for (k = 2; k <= kbm1; k++) {                                                         /* :1860-1867 */
    ki = kb-k;
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)
      vf(i,j,ki) = (L3_(ee,i,j,ki)*vf(i,j,ki+1)+L3_(gg,i,j,ki))*dvm(i,j);
  }
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) wvbot(i,j) = -tps(i,j)*vf(i,j,kbm1);   /* :1869-1873 */
  X2(T, A2_(wvbot), im_local, jm_local);                                                /* :1874 */
}

/* ===================================================================================== */
/* vertvl -- solver.f:1970-2021 */
void pomo_vertvl(pomo_tile *T) {
  int i, j, k;
  double *xflux = zero3(T, 0), *yflux = zero3(T, 1);                                    /* :1977-1978 */
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)       /* :1981-1988 */
    L3_(xflux,i,j,k) = .25*(dy(i,j)+dy(i-1,j))*(dt(i,j)+dt(i-1,j))*u(i,j,k);
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++)       /* :1990-1997 */
    L3_(yflux,i,j,k) = .25*(dx(i,j)+dx(i,j-1))*(dt(i,j)+dt(i,j-1))*v(i,j,k);
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)                               /* :2002-2006 */
    w(i,j,1) = 0.5*(vfluxb(i,j)+vfluxf(i,j));
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)   /* :2008-2018 */
    w(i,j,k+1) = w(i,j,k)
                 +dz(k)*((L3_(xflux,i+1,j,k)-L3_(xflux,i,j,k)+L3_(yflux,i,j+1,k)-L3_(yflux,i,j,k))/(dx(i,j)*dy(i,j))
                        +(etf(i,j)-etb(i,j))/dti2);
}

/* realvertvl -- solver.f:2024-2067 */
void pomo_realvertvl(pomo_tile *T) {
  int i, j, k;
  double dxr, dxl, dyt, dyb;
  memset(A3_(wr), 0, sizeof(double) * T->n3);                                           /* :2031 */
  for (k = 1; k <= kbm1; k++) {
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) tps(i,j) = zz(k)*dt(i,j)+et(i,j);   /* :2034-2038 */
    for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {                           /* :2039-2052 */
      dxr = 2.0/(dx(i+1,j)+dx(i,j));
      dxl = 2.0/(dx(i,j)+dx(i-1,j));
      dyt = 2.0/(dy(i,j+1)+dy(i,j));
      dyb = 2.0/(dy(i,j)+dy(i,j-1));
      wr(i,j,k) = 0.5*(w(i,j,k)+w(i,j,k+1))
                  +0.5*(u(i+1,j,k)*(tps(i+1,j)-tps(i,j))*dxr
                       +u(i,j,k)*(tps(i,j)-tps(i-1,j))*dxl
                       +v(i,j+1,k)*(tps(i,j+1)-tps(i,j))*dyt
                       +v(i,j,k)*(tps(i,j)-tps(i,j-1))*dyb)
                  +(1.0+zz(k))*(etf(i,j)-etb(i,j))/dti2;
    }
  }
  X3(T, A3_(wr), im_local, jm_local, kbm1);                                             /* :2055 */
  if (n_south == -1) for (k = 1; k <= kb; k++) for (i = 1; i <= im_local; i++) wr(i,1,k) = wr(i,2,k);      /* :2057-2060 */
  if (n_north == -1) for (k = 1; k <= kb; k++) for (i = 1; i <= im_local; i++) wr(i,jm,k) = wr(i,jmm1,k);
  if (n_west == -1)  for (k = 1; k <= kb; k++) for (j = 1; j <= jm_local; j++) wr(1,j,k) = wr(2,j,k);
  if (n_east == -1)  for (k = 1; k <= kb; k++) for (j = 1; j <= jm_local; j++) wr(im,j,k) = wr(imm1,j,k);
  for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++)             /* :2062-2064 */
    wr(i,j,k) = fsm(i,j)*wr(i,j,k);
}

/* ===================================================================================== */
/* bcond(idx) -- bounds_forcing.f:6-328; branches used on the hot path: 1, 2, 4, 5, 6 */
void pomo_bcond(pomo_tile *T, int idx) {
  int i, j, k;
  double u1, wm;
  if (idx == 1) {                                                                       /* :18-41 */
    if (n_west == -1) for (j = 1; j <= jm_local; j++) elf(1,j) = elf(2,j);
    if (n_east == -1) for (j = 1; j <= jm_local; j++) elf(im,j) = elf(imm1,j);
    if (n_south == -1) for (i = 1; i <= im_local; i++) elf(i,1) = elf(i,2);
    if (n_north == -1) for (i = 1; i <= im_local; i++) elf(i,jm) = elf(i,jmm1);
    for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++) elf(i,j) = elf(i,j)*fsm(i,j);
  } else if (idx == 2) {                                                                /* :43-83 */
    if (n_west == -1) {
      for (j = 2; j <= jmm1; j++) uaf(2,j) = uabw(j)-rfw*sqrt(grav/d(2,j))*(el(2,j)-elw(j));
      for (j = 2; j <= jmm1; j++) uaf(2,j) = ramp*uaf(2,j);
      for (j = 2; j <= jmm1; j++) uaf(1,j) = uaf(2,j);
      for (j = 2; j <= jmm1; j++) vaf(1,j) = vabw(j);
    }
    if (n_east == -1) {
      for (j = 2; j <= jmm1; j++) uaf(im,j) = uabe(j)+rfe*sqrt(grav/d(imm1,j))*(el(imm1,j)-ele(j));
      for (j = 2; j <= jmm1; j++) uaf(im,j) = ramp*uaf(im,j);
      for (j = 2; j <= jmm1; j++) vaf(im,j) = vabe(j);
    }
    if (n_south == -1) {
      for (i = 2; i <= imm1; i++) vaf(i,2) = vabs(i)-rfs*sqrt(grav/d(i,2))*(el(i,2)-els(i));
      for (i = 2; i <= imm1; i++) vaf(i,2) = ramp*vaf(i,2);
      for (i = 2; i <= imm1; i++) vaf(i,1) = vaf(i,2);
      for (i = 2; i <= imm1; i++) uaf(i,1) = uabs(i);
    }
    if (n_north == -1) {
      for (i = 2; i <= imm1; i++) vaf(i,jm) = vabn(i)+rfn*sqrt(grav/d(i,jmm1))*(el(i,jmm1)-eln(i));
      for (i = 2; i <= imm1; i++) vaf(i,jm) = ramp*vaf(i,jm);
      for (i = 2; i <= imm1; i++) uaf(i,jm) = uabn(i);
    }
    for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++) {
      uaf(i,j) = uaf(i,j)*dum(i,j);
      vaf(i,j) = vaf(i,j)*dvm(i,j);
    }
  } else if (idx == 4) {                                                                /* :151-242 */
    for (k = 1; k <= kbm1; k++) {
      for (j = 1; j <= jm; j++) {
        if (n_east == -1) {
          u1 = 2.*u(im,j,k)*dti/(dx(im,j)+dx(imm1,j));
          if (u1 <= 0.) {
            uf(im,j,k) = t(im,j,k)-u1*(tbe(j,k)-t(im,j,k));
            vf(im,j,k) = s(im,j,k)-u1*(sbe(j,k)-s(im,j,k));
          } else {
            uf(im,j,k) = t(im,j,k)-u1*(t(im,j,k)-t(imm1,j,k));
            vf(im,j,k) = s(im,j,k)-u1*(s(im,j,k)-s(imm1,j,k));
            if (k != 1 && k != kbm1) {
              wm = .5*(w(imm1,j,k)+w(imm1,j,k+1))*dti/((zz(k-1)-zz(k+1))*dt(imm1,j));
              uf(im,j,k) = uf(im,j,k)-wm*(t(imm1,j,k-1)-t(imm1,j,k+1));
              vf(im,j,k) = vf(im,j,k)-wm*(s(imm1,j,k-1)-s(imm1,j,k+1));
            }
          }
        }
        if (n_west == -1) {
          u1 = 2.*u(2,j,k)*dti/(dx(1,j)+dx(2,j));
          if (u1 >= 0.) {
            uf(1,j,k) = t(1,j,k)-u1*(t(1,j,k)-tbw(j,k));
            vf(1,j,k) = s(1,j,k)-u1*(s(1,j,k)-sbw(j,k));
          } else {
            uf(1,j,k) = t(1,j,k)-u1*(t(2,j,k)-t(1,j,k));
            vf(1,j,k) = s(1,j,k)-u1*(s(2,j,k)-s(1,j,k));
            if (k != 1 && k != kbm1) {
              wm = .5*(w(2,j,k)+w(2,j,k+1))*dti/((zz(k-1)-zz(k+1))*dt(2,j));
              uf(1,j,k) = uf(1,j,k)-wm*(t(2,j,k-1)-t(2,j,k+1));
              vf(1,j,k) = vf(1,j,k)-wm*(s(2,j,k-1)-s(2,j,k+1));
            }
          }
        }
      }
      for (i = 1; i <= im; i++) {
        if (n_south == -1) {
          u1 = 2.*v(i,2,k)*dti/(dy(i,1)+dy(i,2));
          if (u1 >= 0.) {
            uf(i,1,k) = t(i,1,k)-u1*(t(i,1,k)-tbs(i,k));
            vf(i,1,k) = s(i,1,k)-u1*(s(i,1,k)-sbs(i,k));
          } else {
            uf(i,1,k) = t(i,1,k)-u1*(t(i,2,k)-t(i,1,k));
            vf(i,1,k) = s(i,1,k)-u1*(s(i,2,k)-s(i,1,k));
            if (k != 1 && k != kbm1) {
              wm = .5*(w(i,2,k)+w(i,2,k+1))*dti/((zz(k-1)-zz(k+1))*dt(i,2));
              uf(i,1,k) = uf(i,1,k)-wm*(t(i,2,k-1)-t(i,2,k+1));
              vf(i,1,k) = vf(i,1,k)-wm*(s(i,2,k-1)-s(i,2,k+1));
            }
          }
        }
        if (n_north == -1) {
          u1 = 2.*v(i,jm,k)*dti/(dy(i,jm)+dy(i,jmm1));
          if (u1 <= 0.) {
            uf(i,jm,k) = t(i,jm,k)-u1*(tbn(i,k)-t(i,jm,k));
            vf(i,jm,k) = s(i,jm,k)-u1*(sbn(i,k)-s(i,jm,k));
          } else {
            uf(i,jm,k) = t(i,jm,k)-u1*(t(i,jm,k)-t(i,jmm1,k));
            vf(i,jm,k) = s(i,jm,k)-u1*(s(i,jm,k)-s(i,jmm1,k));
            if (k != 1 && k != kbm1) {
              wm = .5*(w(i,jmm1,k)+w(i,jmm1,k+1))*dti/((zz(k-1)-zz(k+1))*dt(i,jmm1));
              uf(i,jm,k) = uf(i,jm,k)-wm*(t(i,jmm1,k-1)-t(i,jmm1,k+1));
              vf(i,jm,k) = vf(i,jm,k)-wm*(s(i,jmm1,k-1)-s(i,jmm1,k+1));
            }
          }
        }
      }
    }
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
      uf(i,j,k) = uf(i,j,k)*fsm(i,j);
      vf(i,j,k) = vf(i,j,k)*fsm(i,j);
    }
  } else if (idx == 5) {                                                                /* :244-255 */
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) w(i,j,k) = w(i,j,k)*fsm(i,j);
  } else if (idx == 6) {                                                                /* :257-324 */
    for (k = 1; k <= kb; k++) {
      for (j = 1; j <= jm; j++) {
        if (n_west == -1) {
          u1 = 2.*u(2,j,k)*dti/(dx(1,j)+dx(2,j));
          if (u1 >= 0.) {
            uf(1,j,k) = q2(1,j,k)-u1*(q2(1,j,k)-small);
            vf(1,j,k) = q2l(1,j,k)-u1*(q2l(1,j,k)-small);
          } else {
            uf(1,j,k) = q2(1,j,k)-u1*(q2(2,j,k)-q2(1,j,k));
            vf(1,j,k) = q2l(1,j,k)-u1*(q2l(2,j,k)-q2l(1,j,k));
          }
        }
        if (n_east == -1) {
          u1 = 2.*u(im,j,k)*dti/(dx(im,j)+dx(imm1,j));
          if (u1 <= 0.) {
            uf(im,j,k) = q2(im,j,k)-u1*(small-q2(im,j,k));
            vf(im,j,k) = q2l(im,j,k)-u1*(small-q2l(im,j,k));
          } else {
            uf(im,j,k) = q2(im,j,k)-u1*(q2(im,j,k)-q2(imm1,j,k));
            vf(im,j,k) = q2l(im,j,k)-u1*(q2l(im,j,k)-q2l(imm1,j,k));
          }
        }
      }
      for (i = 1; i <= im; i++) {
        if (n_south == -1) {
          u1 = 2.*v(i,2,k)*dti/(dy(i,1)+dy(i,2));
          if (u1 >= 0.) {
            uf(i,1,k) = q2(i,1,k)-u1*(q2(i,1,k)-small);
            vf(i,1,k) = q2l(i,1,k)-u1*(q2l(i,1,k)-small);
          } else {
            uf(i,1,k) = q2(i,1,k)-u1*(q2(i,2,k)-q2(i,1,k));
            vf(i,1,k) = q2l(i,1,k)-u1*(q2l(i,2,k)-q2l(i,1,k));
          }
        }
        if (n_north == -1) {
          u1 = 2.*v(i,jm,k)*dti/(dy(i,jm)+dy(i,jmm1));
          if (u1 <= 0.) {
            uf(i,jm,k) = q2(i,jm,k)-u1*(small-q2(i,jm,k));
            vf(i,jm,k) = q2l(i,jm,k)-u1*(small-q2l(i,jm,k));
          } else {
            uf(i,jm,k) = q2(i,jm,k)-u1*(q2(i,jm,k)-q2(i,jmm1,k));
            vf(i,jm,k) = q2l(i,jm,k)-u1*(q2l(i,jm,k)-q2l(i,jmm1,k));
          }
        }
      }
    }
    for (k = 1; k <= kb; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
      uf(i,j,k) = uf(i,j,k)*fsm(i,j)+1.e-10;
      vf(i,j,k) = vf(i,j,k)*fsm(i,j)+1.e-10;
    }
  } else {
    fprintf(stderr, "pom_oracle: bcond(%d) is not on the hot path\n", idx);
    abort();
  }
}

/* bcondorl(idx) -- bounds_forcing.f:331-590; branches used on the hot path: 3, 5 */
void pomo_bcondorl(pomo_tile *T, int idx) {
  int i, j, k;
  double cl, denom;
  if (idx == 3) {                                                                       /* :418-487 */
    for (k = 1; k <= kbm1; k++) {
      for (j = 2; j <= jmm1; j++) {
        if (n_east == -1) {
          denom = (uf(im-1,j,k)+ub(im-1,j,k)-2.*u(im-2,j,k));
          if (denom == 0.) denom = 0.01;
          cl = (ub(im-1,j,k)-uf(im-1,j,k))/denom;
          if (cl > 1.) cl = 1.;
          if (cl < 0.) cl = 0.;
          uf(im,j,k) = (ub(im,j,k)*(1.-cl)+2.*cl*u(im-1,j,k))/(1.+cl);
          vf(im,j,k) = 0.;
        }
        if (n_west == -1) {
          denom = (uf(3,j,k)+ub(3,j,k)-2.*u(4,j,k));
          if (denom == 0.) denom = 0.01;
          cl = (ub(3,j,k)-uf(3,j,k))/denom;
          if (cl > 1.) cl = 1.;
          if (cl < 0.) cl = 0.;
          uf(2,j,k) = (ub(2,j,k)*(1.-cl)+2.*cl*u(3,j,k))/(1.+cl);
          uf(1,j,k) = uf(2,j,k);
          vf(1,j,k) = 0.;
        }
      }
      for (i = 2; i <= imm1; i++) {
        if (n_south == -1) {
          denom = (vf(i,3,k)+vb(i,3,k)-2.*v(i,4,k));
          if (fabs(denom) == 0.0) denom = 0.01;
          cl = (vb(i,3,k)-vf(i,3,k))/denom;
          if (cl > 1.) cl = 1.;
          if (cl < 0.) cl = 0.;
          vf(i,2,k) = (vb(i,2,k)*(1.-cl)+2.*cl*v(i,3,k))/(1.+cl);
          vf(i,1,k) = vf(i,2,k);
          uf(i,1,k) = 0.;
        }
        if (n_north == -1) {
          denom = (vf(i,jm-1,k)+vb(i,jm-1,k)-2.*v(i,jm-2,k));
          if (fabs(denom) == 0.0) denom = 0.01;
          cl = (vb(i,jm-1,k)-vf(i,jm-1,k))/denom;
          if (cl > 1.) cl = 1.;
          if (cl < 0.) cl = 0.;
          vf(i,jm,k) = (vb(i,jm,k)*(1.-cl)+2.*cl*v(i,jm-1,k))/(1.+cl);
          uf(i,jm,k) = 0.;
        }
      }
    }
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
      uf(i,j,k) = uf(i,j,k)*dum(i,j);
      vf(i,j,k) = vf(i,j,k)*dvm(i,j);
    }
  } else if (idx == 5) {                                                                /* :550-561 */
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) w(i,j,k) = w(i,j,k)*fsm(i,j);
  } else {
    fprintf(stderr, "pom_oracle: bcondorl(%d) is not on the hot path\n", idx);
    abort();
  }
}

/* ===================================================================================== */
/* restore_interior -- bounds_forcing.f:1023-1121; the reader calls become look-ups of the
 * records registered in T->rec_t/rec_s (same records the reference build is served). */
static void load_record(pomo_tile *T, int n) {
  int i, j, k;
  if (n < 1 || n > POMO_MAXREC || !T->rec_t[n]) {
    fprintf(stderr, "pom_oracle: restore record %d was not registered\n", n);
    abort();
  }
  for (k = 1; k <= kb; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {       /* :1041-1042,1062-1063 */
    trstrf(i,j,k) = L3_(T->rec_t[n],i,j,k);
    srstrf(i,j,k) = L3_(T->rec_s[n],i,j,k);
  }
  { double *p = A3_(taurstrf); size_t n3; const double v = 1./30.;                       /* taurstrf = 1./trst (trst is REAL(8)) */
    for (n3 = 0; n3 < T->n3; n3++) p[n3] = v; }
}
void pomo_restore_interior(pomo_tile *T) {
  int i, j, k, ntime, irst;
  double trst, fold, fnew;
  trst = 30.;                                                                           /* :1034-1036 */
  irst = (int)(trst*86400./dti);
  ntime = (int)(CON_(time)/trst);
  if (iint == 2) load_record(T, (iint/irst)+1);                                         /* :1040-1053 */
  if (iint == 2 || (iint % irst) == 0) {                                                /* :1055-1083 */
    for (k = 1; k <= kbm1; k++) for (i = 1; i <= im; i++) for (j = 1; j <= jm; j++) {
      trstrb(i,j,k) = trstrf(i,j,k);
      srstrb(i,j,k) = srstrf(i,j,k);
      taurstrb(i,j,k) = taurstrf(i,j,k);
    }
    if (iint != iend) load_record(T, (iint+irst)/irst+1);
  }
  fnew = CON_(time)/trst-ntime;                                                         /* :1086-1087 */
  fold = 1.-fnew;
  for (k = 1; k <= kbm1; k++) for (i = 1; i <= im; i++) for (j = 1; j <= jm; j++) {     /* :1088-1096 */
    trstr(i,j,k) = fold*trstrb(i,j,k)+fnew*trstrf(i,j,k);
    srstr(i,j,k) = fold*srstrb(i,j,k)+fnew*srstrf(i,j,k);
    taurstr(i,j,k) = fold*taurstrb(i,j,k)+fnew*taurstrf(i,j,k);
  }
  for (k = 1; k <= kbm1; k++) for (i = 1; i <= im; i++) for (j = 1; j <= jm; j++) {     /* :1099-1112 */
    t(i,j,k) = t(i,j,k)+2.*dti/86400.*taurstr(i,j,k)*(trstr(i,j,k)-t(i,j,k));
    tb(i,j,k) = tb(i,j,k)+2.*dti/86400.*taurstr(i,j,k)*(trstr(i,j,k)-tb(i,j,k));
    s(i,j,k) = s(i,j,k)+2.*dti/86400.*taurstr(i,j,k)*(srstr(i,j,k)-s(i,j,k));
    sb(i,j,k) = sb(i,j,k)+2.*dti/86400.*taurstr(i,j,k)*(srstr(i,j,k)-sb(i,j,k));
  }
  for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm_local; j++) for (i = 1; i <= im_local; i++) {   /* :1115-1120 */
    t(i,j,k) = t(i,j,k)*fsm(i,j);
    tb(i,j,k) = tb(i,j,k)*fsm(i,j);
    s(i,j,k) = s(i,j,k)*fsm(i,j);
    sb(i,j,k) = sb(i,j,k)*fsm(i,j);
  }
}

/* ===================================================================================== */
/* get_time -- advance.f:62-75 */
void pomo_get_time(pomo_tile *T) {
  CON_(time) = dti*(double)(float)iint/86400.+time0;                                    /* :66 float(iint) */
  if (iint >= CON_(iswtch)) CON_(iprint) = (int)lround(CON_(prtd2)*24.*3600./dti);      /* :67 */
  if (T->lramp) {
    ramp = CON_(time)/period;
    if (ramp > 1.) ramp = 1.;
  } else {
    ramp = 1.;
  }
}

/* lateral_viscosity -- advance.f:96-141 */
void pomo_lateral_viscosity(pomo_tile *T) {
  int i, j, k;
  if (mode != 2) {
    pomo_advct(T);
    if (npg == 1) {
      pomo_baropg(T);
    } else if (npg == 2) {
      pomo_baropg_mcc(T);                                                               /* :115-116 */
    } else {
      error_status = 1;
      fprintf(stderr, "Error: invalid value for npg\n");
    }
    for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) /* :122-136 */
      aam(i,j,k) = horcon*dx(i,j)*dy(i,j)
                   *sqrt(sq((u(i+1,j,k)-u(i,j,k))/dx(i,j))
                        +sq((v(i,j+1,k)-v(i,j,k))/dy(i,j))
                        +.5*sq(.25*(u(i,j+1,k)+u(i+1,j+1,k)-u(i,j-1,k)-u(i+1,j-1,k))/dy(i,j)
                              +.25*(v(i+1,j,k)+v(i+1,j+1,k)-v(i-1,j,k)-v(i-1,j+1,k))/dx(i,j)));
    X3(T, A3_(aam), im_local, jm_local, kbm1);                                          /* :137 */
  }
}

/* mode_interaction -- advance.f:144-202 */
void pomo_mode_interaction(pomo_tile *T) {
  int i, j, k;
  if (mode != 2) {
    memset(A2_(adx2d), 0, sizeof(double) * T->n2);                                      /* :152-156 */
    memset(A2_(ady2d), 0, sizeof(double) * T->n2);
    memset(A2_(drx2d), 0, sizeof(double) * T->n2);
    memset(A2_(dry2d), 0, sizeof(double) * T->n2);
    memset(A2_(aam2d), 0, sizeof(double) * T->n2);
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {   /* :158-168 */
      adx2d(i,j) = adx2d(i,j)+advx(i,j,k)*dz(k);
      ady2d(i,j) = ady2d(i,j)+advy(i,j,k)*dz(k);
      drx2d(i,j) = drx2d(i,j)+drhox(i,j,k)*dz(k);
      dry2d(i,j) = dry2d(i,j)+drhoy(i,j,k)*dz(k);
      aam2d(i,j) = aam2d(i,j)+aam(i,j,k)*dz(k);
    }
    pomo_advave(T);                                                                     /* :170 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {                               /* :172-177 */
      adx2d(i,j) = adx2d(i,j)-advua(i,j);
      ady2d(i,j) = ady2d(i,j)-advva(i,j);
    }
  }
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) egf(i,j) = el(i,j)*ispi;          /* :181-185 */
  for (j = 1; j <= jm; j++) for (i = 2; i <= im; i++) utf(i,j) = ua(i,j)*(d(i,j)+d(i-1,j))*isp2i;   /* :187-191 */
  for (j = 2; j <= jm; j++) for (i = 1; i <= im; i++) vtf(i,j) = va(i,j)*(d(i,j)+d(i,j-1))*isp2i;   /* :192-196 */
  X2(T, A2_(utf), im_local, jm_local);                                                  /* :198-199 */
  X2(T, A2_(vtf), im_local, jm_local);
}

/* mode_external -- advance.f:205-353 */
void pomo_mode_external(pomo_tile *T) {
  int i, j; size_t n;
  for (j = 2; j <= jm; j++) for (i = 2; i <= im; i++) {                                 /* :211-218 */
    fluxua(i,j) = .25*(d(i,j)+d(i-1,j))*(dy(i,j)+dy(i-1,j))*ua(i,j);
    fluxva(i,j) = .25*(d(i,j)+d(i,j-1))*(dx(i,j)+dx(i,j-1))*va(i,j);
  }
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++)                               /* :222-229 */
    elf(i,j) = elb(i,j)+dte2*(-(fluxua(i+1,j)-fluxua(i,j)+fluxva(i,j+1)-fluxva(i,j))/art(i,j)-vfluxf(i,j));
  pomo_bcond(T, 1);                                                                     /* :231 */
  X2(T, A2_(elf), im_local, jm_local);                                                  /* :233 */
  if (iext % ispadv == 0) pomo_advave(T);                                               /* :235 */
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= im; i++)                                 /* :237-252 */
    uaf(i,j) = adx2d(i,j)+advua(i,j)
               -aru(i,j)*.25*(cor(i,j)*d(i,j)*(va(i,j+1)+va(i,j))+cor(i-1,j)*d(i-1,j)*(va(i-1,j+1)+va(i-1,j)))
               +.25*grav*(dy(i,j)+dy(i-1,j))*(d(i,j)+d(i-1,j))
                *((1.-2.*alpha)*(el(i,j)-el(i-1,j))
                  +alpha*(elb(i,j)-elb(i-1,j)+elf(i,j)-elf(i-1,j))
                  +e_atmos(i,j)-e_atmos(i-1,j))
               +drx2d(i,j)+aru(i,j)*(wusurf(i,j)-wubot(i,j));
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= im; i++)                                 /* :254-262 */
    uaf(i,j) = ((h(i,j)+elb(i,j)+h(i-1,j)+elb(i-1,j))*aru(i,j)*uab(i,j)-4.*dte*uaf(i,j))
               /((h(i,j)+elf(i,j)+h(i-1,j)+elf(i-1,j))*aru(i,j));
  for (j = 2; j <= jm; j++) for (i = 2; i <= imm1; i++)                                 /* :264-278 */
    vaf(i,j) = ady2d(i,j)+advva(i,j)
               +arv(i,j)*.25*(cor(i,j)*d(i,j)*(ua(i+1,j)+ua(i,j))+cor(i,j-1)*d(i,j-1)*(ua(i+1,j-1)+ua(i,j-1)))
               +.25*grav*(dx(i,j)+dx(i,j-1))*(d(i,j)+d(i,j-1))
                *((1.-2.*alpha)*(el(i,j)-el(i,j-1))
                  +alpha*(elb(i,j)-elb(i,j-1)+elf(i,j)-elf(i,j-1))
                  +e_atmos(i,j)-e_atmos(i,j-1))
               +dry2d(i,j)+arv(i,j)*(wvsurf(i,j)-wvbot(i,j));
  for (j = 2; j <= jm; j++) for (i = 2; i <= imm1; i++)                                 /* :280-288 */
    vaf(i,j) = ((h(i,j)+elb(i,j)+h(i,j-1)+elb(i,j-1))*arv(i,j)*vab(i,j)-4.*dte*vaf(i,j))
               /((h(i,j)+elf(i,j)+h(i,j-1)+elf(i,j-1))*arv(i,j));
  pomo_bcond(T, 2);                                                                     /* :290 */
  X2(T, A2_(uaf), im_local, jm_local);                                                  /* :292-293 */
  X2(T, A2_(vaf), im_local, jm_local);
  if (iext == (isplit-2)) {                                                             /* :295-318 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) etf(i,j) = .25*smoth*elf(i,j);
  } else if (iext == (isplit-1)) {
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) etf(i,j) = etf(i,j)+.5*(1.-.5*smoth)*elf(i,j);
  } else if (iext == isplit) {
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) etf(i,j) = (etf(i,j)+.5*elf(i,j))*fsm(i,j);
  }
  {                                                                                     /* :321-330, whole arrays */
    double *pua = A2_(ua), *puab = A2_(uab), *puaf = A2_(uaf), *pva = A2_(va), *pvab = A2_(vab), *pvaf = A2_(vaf);
    double *pel = A2_(el), *pelb = A2_(elb), *pelf = A2_(elf), *pd = A2_(d), *ph = A2_(h);
    for (n = 0; n < T->n2; n++) pua[n] = pua[n]+.5*smoth*(puab[n]-2.*pua[n]+puaf[n]);
    for (n = 0; n < T->n2; n++) pva[n] = pva[n]+.5*smoth*(pvab[n]-2.*pva[n]+pvaf[n]);
    for (n = 0; n < T->n2; n++) pel[n] = pel[n]+.5*smoth*(pelb[n]-2.*pel[n]+pelf[n]);
    for (n = 0; n < T->n2; n++) pelb[n] = pel[n];
    for (n = 0; n < T->n2; n++) pel[n] = pelf[n];
    for (n = 0; n < T->n2; n++) pd[n] = ph[n]+pel[n];
    for (n = 0; n < T->n2; n++) puab[n] = pua[n];
    for (n = 0; n < T->n2; n++) pua[n] = puaf[n];
    for (n = 0; n < T->n2; n++) pvab[n] = pva[n];
    for (n = 0; n < T->n2; n++) pva[n] = pvaf[n];
  }
  if (iext != isplit) {                                                                 /* :332-350 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) egf(i,j) = egf(i,j)+el(i,j)*ispi;
    for (j = 1; j <= jm; j++) for (i = 2; i <= im; i++) utf(i,j) = utf(i,j)+ua(i,j)*(d(i,j)+d(i-1,j))*isp2i;
    for (j = 2; j <= jm; j++) for (i = 1; i <= im; i++) vtf(i,j) = vtf(i,j)+va(i,j)*(d(i,j)+d(i,j-1))*isp2i;
    X2(T, A2_(utf), im_local, jm_local);
    X2(T, A2_(vtf), im_local, jm_local);
  }
}

/* mode_internal -- advance.f:356-537 */
void pomo_mode_internal(pomo_tile *T) {
  int i, j, k; size_t n;
  if ((iint != 1 || time0 != 0.) && mode != 2) {                                        /* :362 */
    double *ptps = A2_(tps);
    memset(ptps, 0, sizeof(double) * T->n2);                                            /* :365-369 */
    for (k = 1; k <= kbm1; k++) { double *pu = A3_(u)+(size_t)(k-1)*T->n2;
      for (n = 0; n < T->n2; n++) ptps[n] = ptps[n]+pu[n]*dz(k); }
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 2; i <= im; i++)     /* :371-378 */
      u(i,j,k) = (u(i,j,k)-tps(i,j))+(utb(i,j)+utf(i,j))/(dt(i,j)+dt(i-1,j));
    memset(ptps, 0, sizeof(double) * T->n2);                                            /* :380-384 */
    for (k = 1; k <= kbm1; k++) { double *pv = A3_(v)+(size_t)(k-1)*T->n2;
      for (n = 0; n < T->n2; n++) ptps[n] = ptps[n]+pv[n]*dz(k); }
    for (k = 1; k <= kbm1; k++) for (j = 2; j <= jm; j++) for (i = 1; i <= im; i++)     /* :386-393 */
      v(i,j,k) = (v(i,j,k)-tps(i,j))+(vtb(i,j)+vtf(i,j))/(dt(i,j)+dt(i,j-1));
    pomo_vertvl(T);                                                                     /* :396-400 */
    pomo_bcondorl(T, 5);
    X3(T, A3_(w), im_local, jm_local, kb);
    memset(A3_(uf), 0, sizeof(double) * T->n3);                                         /* :403-404 */
    memset(A3_(vf), 0, sizeof(double) * T->n3);
    pomo_advq(T, A3_(q2b), A3_(q2), A3_(uf));                                           /* :407-409 */
    pomo_advq(T, A3_(q2lb), A3_(q2l), A3_(vf));
    pomo_profq(T);
    X3(T, A3_(uf)+T->n2, im_local, jm_local, kbm2);                                     /* :411-412 */
    X3(T, A3_(vf)+T->n2, im_local, jm_local, kbm2);
    pomo_bcond(T, 6);                                                                   /* :414 */
    {                                                                                   /* :416-421 */
      double *pq2 = A3_(q2), *pq2b = A3_(q2b), *pq2l = A3_(q2l), *pq2lb = A3_(q2lb), *puf = A3_(uf), *pvf = A3_(vf);
      for (n = 0; n < T->n3; n++) pq2[n] = pq2[n]+.5*smoth*(puf[n]+pq2b[n]-2.*pq2[n]);
      for (n = 0; n < T->n3; n++) pq2l[n] = pq2l[n]+.5*smoth*(pvf[n]+pq2lb[n]-2.*pq2l[n]);
      memcpy(pq2b, pq2, sizeof(double) * T->n3);
      memcpy(pq2, puf, sizeof(double) * T->n3);
      memcpy(pq2lb, pq2l, sizeof(double) * T->n3);
      memcpy(pq2l, pvf, sizeof(double) * T->n3);
    }
    if (mode != 4) {                                                                    /* :424-456 */
      if (nadv == 1) {
        pomo_advt1(T, A3_(tb), A3_(t), A3_(tclim), A3_(uf));
        pomo_advt1(T, A3_(sb), A3_(s), A3_(sclim), A3_(vf));
      } else if (nadv == 2) {
        pomo_advt2(T, A3_(tb), A3_(t), A3_(tclim), A3_(uf));
        pomo_advt2(T, A3_(sb), A3_(s), A3_(sclim), A3_(vf));
      } else {
        error_status = 1;
        fprintf(stderr, "Error: invalid value for nadv\n");
      }
      X3(T, A3_(uf), im_local, jm_local, kbm1);                                         /* :436-437 */
      X3(T, A3_(vf), im_local, jm_local, kbm1);
      pomo_proft(T, A3_(uf), A2_(wtsurf), A2_(tsurf), nbct);                            /* :439-440 */
      pomo_proft(T, A3_(vf), A2_(wssurf), A2_(ssurf), nbcs);
      pomo_bcond(T, 4);                                                                 /* :442 */
      {                                                                                 /* :444-449 */
        double *pt = A3_(t), *ptb = A3_(tb), *ps = A3_(s), *psb = A3_(sb), *puf = A3_(uf), *pvf = A3_(vf);
        for (n = 0; n < T->n3; n++) pt[n] = pt[n]+.5*smoth*(puf[n]+ptb[n]-2.*pt[n]);
        for (n = 0; n < T->n3; n++) ps[n] = ps[n]+.5*smoth*(pvf[n]+psb[n]-2.*ps[n]);
        memcpy(ptb, pt, sizeof(double) * T->n3);
        memcpy(pt, puf, sizeof(double) * T->n3);
        memcpy(psb, ps, sizeof(double) * T->n3);
        memcpy(ps, pvf, sizeof(double) * T->n3);
      }
      pomo_restore_interior(T);                                                         /* :452 */
      pomo_dens(T, A3_(s), A3_(t), A3_(rho));                                           /* :454 */
    }
    pomo_advu(T);                                                                       /* :459-462 */
    pomo_advv(T);
    pomo_profu(T);
    pomo_profv(T);
    pomo_bcondorl(T, 3);                                                                /* :464 */
    X3(T, A3_(uf), im_local, jm_local, kbm1);                                           /* :466-467 */
    X3(T, A3_(vf), im_local, jm_local, kbm1);
    memset(ptps, 0, sizeof(double) * T->n2);                                            /* :469-478 */
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)
      tps(i,j) = tps(i,j)+(uf(i,j,k)+ub(i,j,k)-2.*u(i,j,k))*dz(k);
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)     /* :480-488 */
      u(i,j,k) = u(i,j,k)+.5*smoth*(uf(i,j,k)+ub(i,j,k)-2.*u(i,j,k)-tps(i,j));
    memset(ptps, 0, sizeof(double) * T->n2);                                            /* :490-499 */
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)
      tps(i,j) = tps(i,j)+(vf(i,j,k)+vb(i,j,k)-2.*v(i,j,k))*dz(k);
    for (k = 1; k <= kbm1; k++) for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)     /* :501-509 */
      v(i,j,k) = v(i,j,k)+.5*smoth*(vf(i,j,k)+vb(i,j,k)-2.*v(i,j,k)-tps(i,j));
    memcpy(A3_(ub), A3_(u), sizeof(double) * T->n3);                                    /* :511-514 */
    memcpy(A3_(u), A3_(uf), sizeof(double) * T->n3);
    memcpy(A3_(vb), A3_(v), sizeof(double) * T->n3);
    memcpy(A3_(v), A3_(vf), sizeof(double) * T->n3);
    X3(T, A3_(ub), im_local, jm_local, kb);                                             /* :516-521 */
    X3(T, A3_(u), im_local, jm_local, kb);
    X3(T, A3_(uf), im_local, jm_local, kb);
    X3(T, A3_(vb), im_local, jm_local, kb);
    X3(T, A3_(v), im_local, jm_local, kb);
    X3(T, A3_(vf), im_local, jm_local, kb);
  }
  memcpy(A2_(egb), A2_(egf), sizeof(double) * T->n2);                                   /* :525-531 */
  memcpy(A2_(etb), A2_(et), sizeof(double) * T->n2);
  memcpy(A2_(et), A2_(etf), sizeof(double) * T->n2);
  { double *pdt = A2_(dt), *ph = A2_(h), *pet = A2_(et); for (n = 0; n < T->n2; n++) pdt[n] = ph[n]+pet[n]; }
  memcpy(A2_(utb), A2_(utf), sizeof(double) * T->n2);
  memcpy(A2_(vtb), A2_(vtf), sizeof(double) * T->n2);
  memcpy(A2_(vfluxb), A2_(vfluxf), sizeof(double) * T->n2);
  pomo_realvertvl(T);                                                                   /* :534 */
}

/* check_velocity -- advance.f:611-641 */
/* ===================================================================================== */
/* surface_forcing -- advance.f:77-93 -> wind, heat, surface (bounds_forcing.f:871-983).  The readers
 * (read_wind_pnetcdf ...) are the caller's: records registered with pomo_set_forcing_record. */
void pomo_set_forcing_record(pomo_tile *T, int kind, int n, const double *a, const double *b) {
  if (kind < 0 || kind > 2 || n < 1 || n > POMO_MAXFREC) { fprintf(stderr, "pomo: bad forcing record %d/%d\n", kind, n); abort(); }
  T->frc_a[kind][n] = a; T->frc_b[kind][n] = b;
}
static void frc_read(pomo_tile *T, int kind, int n, double *fa, double *fb) {   /* x(1:im,1:jm) = record */
  int i, j;
  if (n < 1 || n > POMO_MAXFREC || !T->frc_a[kind][n]) { fprintf(stderr, "pomo: forcing record %d/%d was not supplied\n", kind, n); abort(); }
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
    G2_(fa,i,j) = L2_(T->frc_a[kind][n],i,j);
    if (fb) G2_(fb,i,j) = L2_(T->frc_b[kind][n],i,j);
  }
}
/* the shared shape of wind and heat: two fields, records every `tint` days, linear in time */
static void frc_interp(pomo_tile *T, int kind, double tint, double *x, double *xb, double *xf, double *y, double *yb, double *yf) {
  const double time = CON_(time);
  const int cont_bry = CON_(cont_bry);
  const int istep = (int)(tint*86400./dti);
  int i, j, ntime;
  double fold, fnew;
  if (iint == 1) frc_read(T, kind, (iint+cont_bry)/istep+1, xf, yf);                     /* :884-888 */
  if (iint == 1 || (iint+cont_bry) % istep == 0) {                                        /* :890-902 */
    for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) { G2_(xb,i,j) = G2_(xf,i,j); G2_(yb,i,j) = G2_(yf,i,j); }
    if (iint != iend) frc_read(T, kind, (iint+cont_bry+istep)/istep+1, xf, yf);
  }
  ntime = (int)(time/tint);                                                               /* :905-909 */
  fnew = time/tint-ntime;
  fold = 1.-fnew;
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++) {
    G2_(x,i,j) = fold*G2_(xb,i,j)+fnew*G2_(xf,i,j);
    G2_(y,i,j) = fold*G2_(yb,i,j)+fnew*G2_(yf,i,j);
  }
}
void pomo_wind(pomo_tile *T) {   /* twind = .125 */
  frc_interp(T, 0, .125, A2_(wusurf), A2_(wusurfb), A2_(wusurff), A2_(wvsurf), A2_(wvsurfb), A2_(wvsurff));
}
void pomo_heat(pomo_tile *T) {   /* theat = .125 */
  frc_interp(T, 1, .125, A2_(wtsurf), A2_(wtsurfb), A2_(wtsurff), A2_(swrad), A2_(swradb), A2_(swradf));
}
void pomo_surface(pomo_tile *T) {   /* tsrf = .125; SST only, no interpolation (:975-980) */
  const int cont_bry = CON_(cont_bry);
  const int isrf = (int)(.125*86400./dti);
  if (iint == 1 || (iint+cont_bry) % isrf == 0) frc_read(T, 2, (iint+cont_bry)/isrf+1, A2_(tsurf), NULL);
}
void pomo_surface_forcing(pomo_tile *T) { pomo_wind(T); pomo_heat(T); pomo_surface(T); }

/* ===================================================================================== */
/* lateral_bc -- bounds_forcing.f:593-868 (active lines; the z-to-sigma blocks are commented out in the
 * reference).  Records every tbc = 1./24. days (a REAL(4) constant).  Kept as written: only tb*, sb*, uabw/e
 * and vabn/s have their "b" copies refreshed at a record change (:742-753) -- ub?b, vb?b keep whatever
 * they held, although the interpolation reads them (:773-784). */
static const int LAT_SLOT[20] = {PB_tbwf, PB_sbwf, PB_ubwf, PB_vbwf, PB_tbef, PB_sbef, PB_ubef, PB_vbef, PB_tbnf, PB_sbnf,
                                 PB_vbnf, PB_ubnf, PB_tbsf, PB_sbsf, PB_vbsf, PB_ubsf, PB_elw, PB_ele, PB_eln, PB_els};
static size_t bd_len(pomo_tile *T, int slot) {          /* number of doubles of a bdry member */
  static const char shape[] = {
#define BDS_(name, shp) (#shp)[0] == 'J' ? ((#shp)[1] ? 'j' : 'J') : ((#shp)[1] ? 'i' : 'I'),
    POM_BDRY(BDS_)
#undef BDS_
  };
  switch (shape[slot]) { case 'J': return (size_t)jm_local; case 'I': return (size_t)im_local;
                         case 'j': return (size_t)jm_local*kb; default: return (size_t)im_local*kb; }
}
void pomo_set_lateral_record(pomo_tile *T, int n, const double *const *arrays20) {
  int a;
  if (n < 1 || n > POMO_MAXFREC) { fprintf(stderr, "pomo: bad lateral record %d\n", n); abort(); }
  for (a = 0; a < 20; a++) T->lat[n][a] = arrays20[a];
}
static void lat_read(pomo_tile *T, int n) {             /* read_boundary_conditions_pnetcdf + the depth integrals (:610-636) */
  int a, i, j, k;
  if (n < 1 || n > POMO_MAXFREC || !T->lat[n][0]) { fprintf(stderr, "pomo: lateral record %d was not supplied\n", n); abort(); }
  for (a = 0; a < 20; a++) memcpy(T->bd[LAT_SLOT[a]], T->lat[n][a], sizeof(double)*bd_len(T, LAT_SLOT[a]));
  for (j = 1; j <= jm_local; j++) { uabwf(j) = 0.; vabwf(j) = 0.; uabef(j) = 0.; vabef(j) = 0.; }
  for (i = 1; i <= im_local; i++) { uabnf(i) = 0.; vabnf(i) = 0.; uabsf(i) = 0.; vabsf(i) = 0.; }
  for (k = 1; k <= kb; k++) {
    for (j = 1; j <= jm_local; j++) {
      uabwf(j) = uabwf(j)+ubwf(j,k)*dz(k); vabwf(j) = vabwf(j)+vbwf(j,k)*dz(k);
      uabef(j) = uabef(j)+ubef(j,k)*dz(k); vabef(j) = vabef(j)+vbef(j,k)*dz(k);
    }
    for (i = 1; i <= im_local; i++) {
      uabnf(i) = uabnf(i)+ubnf(i,k)*dz(k); vabnf(i) = vabnf(i)+vbnf(i,k)*dz(k);
      uabsf(i) = uabsf(i)+ubsf(i,k)*dz(k); vabsf(i) = vabsf(i)+vbsf(i,k)*dz(k);
    }
  }
}
void pomo_lateral_bc(pomo_tile *T) {
  const double time = CON_(time);
  const int cont_bry = CON_(cont_bry);
  const double tbc = (double)(1.f/24.f);                                                  /* :602 */
  const int ibc = (int)(tbc*86400./dti);
  const int ntime = (int)(time/tbc);
  int i, j, k;
  double fold, fnew;
  if (iint == 1) lat_read(T, (iint+cont_bry)/ibc+1);                                      /* :607-636 */
  if (iint == 1 || (iint+cont_bry) % ibc == 0) {                                          /* :740-772 */
    for (k = 1; k <= kb; k++) {
      for (j = 1; j <= jm_local; j++) { tbwb(j,k) = tbwf(j,k); sbwb(j,k) = sbwf(j,k); tbeb(j,k) = tbef(j,k); sbeb(j,k) = sbef(j,k); }
      for (i = 1; i <= im_local; i++) { tbnb(i,k) = tbnf(i,k); sbnb(i,k) = sbnf(i,k); tbsb(i,k) = tbsf(i,k); sbsb(i,k) = sbsf(i,k); }
    }
    for (j = 1; j <= jm_local; j++) { uabwb(j) = uabwf(j); uabeb(j) = uabef(j); }
    for (i = 1; i <= im_local; i++) { vabnb(i) = vabnf(i); vabsb(i) = vabsf(i); }
    if (iint != iend) lat_read(T, (iint+cont_bry+ibc)/ibc+1);
  }
  fnew = time/tbc-(double)ntime;                                                          /* :774-775 */
  fold = 1.-fnew;
  for (k = 1; k <= kb; k++) {                                                             /* :776-787 */
    for (j = 1; j <= jm; j++) {
      tbw(j,k) = fold*tbwb(j,k)+fnew*tbwf(j,k); sbw(j,k) = fold*sbwb(j,k)+fnew*sbwf(j,k); ubw(j,k) = fold*ubwb(j,k)+fnew*ubwf(j,k);
      tbe(j,k) = fold*tbeb(j,k)+fnew*tbef(j,k); sbe(j,k) = fold*sbeb(j,k)+fnew*sbef(j,k); ube(j,k) = fold*ubeb(j,k)+fnew*ubef(j,k);
    }
    for (i = 1; i <= im; i++) {
      tbn(i,k) = fold*tbnb(i,k)+fnew*tbnf(i,k); sbn(i,k) = fold*sbnb(i,k)+fnew*sbnf(i,k); vbn(i,k) = fold*vbnb(i,k)+fnew*vbnf(i,k);
      tbs(i,k) = fold*tbsb(i,k)+fnew*tbsf(i,k); sbs(i,k) = fold*sbsb(i,k)+fnew*sbsf(i,k); vbs(i,k) = fold*vbsb(i,k)+fnew*vbsf(i,k);
    }
  }
  for (j = 1; j <= jm_local; j++) { uabe(j) = 0.; uabw(j) = 0.; }                         /* :788-791 */
  for (i = 1; i <= im_local; i++) { vabn(i) = 0.; vabs(i) = 0.; }
  for (k = 1; k <= kb; k++) {                                                             /* :792-797 */
    for (j = 1; j <= jm; j++) { uabe(j) = uabe(j)+ube(j,k)*dz(k); uabw(j) = uabw(j)+ubw(j,k)*dz(k); }
    for (i = 1; i <= im; i++) { vabn(i) = vabn(i)+vbn(i,k)*dz(k); vabs(i) = vabs(i)+vbs(i,k)*dz(k); }
  }
}

/* domain_stats -- advance.f:644-756.  dvol is assigned on 2:imm1 x 2:jmm1 only (:692-695) and is zero
 * elsewhere, so every 3-D sum -- including the "physical edge" additions of vtot, tavg, stot, ekin, which
 * read the zero edge of dvol / dmass -- is a sum over the interior; atot and eavg do include the edges
 * (without corners).  Plain left-to-right additions here (k outermost as in the array expressions). */
void pomo_domain_stats(pomo_tile *T, double *out, int sums_only) {
  int i, j, k;
  double vtot = 0., atot = 0., mtot = 0., stot = 0., tavg = 0., savg = 0., eavg = 0., ekin = 0.;
#define DAREA(i,j) (dx(i,j)*dy(i,j)*fsm(i,j))
  for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) { atot += DAREA(i,j); eavg += et(i,j)*DAREA(i,j); }   /* :666, :672 */
  if (n_west == -1)  for (j = 2; j <= jmm1; j++) { atot += DAREA(1,j);  eavg += et(1,j)*DAREA(1,j); }            /* :667, :673 */
  if (n_east == -1)  for (j = 2; j <= jmm1; j++) { atot += DAREA(im,j); eavg += et(im,j)*DAREA(im,j); }
  if (n_south == -1) for (i = 2; i <= imm1; i++) { atot += DAREA(i,1);  eavg += et(i,1)*DAREA(i,1); }
  if (n_north == -1) for (i = 2; i <= imm1; i++) { atot += DAREA(i,jm); eavg += et(i,jm)*DAREA(i,jm); }
  for (k = 1; k <= kbm1; k++) for (j = 2; j <= jmm1; j++) for (i = 2; i <= imm1; i++) {
    const double dvol = DAREA(i,j)*dt(i,j)*dz(k);                                                             /* :692-695 */
    const double dmass = dvol*(rho(i,j,k)*rhoref+1000.);                                                       /* :702-703 */
    vtot += dvol; mtot += dmass;
    tavg += tb(i,j,k)*dvol; stot += sb(i,j,k)*dvol;                                                            /* :707-708 */
    ekin += .5*(dmass*(u(i,j,k)*u(i,j,k)+v(i,j,k)*v(i,j,k)));                                                  /* :738-740 */
  }
#undef DAREA
  if (!sums_only) {                                                                                           /* :680-686, :728-736 */
    eavg = (atot != 0) ? eavg/atot : 0.;
    if (vtot != 0) { tavg = tavg/vtot; savg = stot/vtot; } else { tavg = 0.; savg = 0.; }
  }
  out[0] = vtot; out[1] = atot; out[2] = mtot; out[3] = stot; out[4] = tavg; out[5] = savg; out[6] = eavg; out[7] = ekin;
}

void pomo_check_velocity(pomo_tile *T) {
  int i, j;
  double vamax = 0.; int imax = 0, jmax = 0;
  for (j = 1; j <= jm; j++) for (i = 1; i <= im; i++)
    if (fabs(vaf(i,j)) >= vamax) { vamax = fabs(vaf(i,j)); imax = i; jmax = j; }
  T->vamax = vamax; T->imax = imax; T->jmax = jmax;
  if (vamax > vmaxl) {
    if (error_status == 0)
      fprintf(stderr, "Error: velocity condition violated\n iint =%8d vamax =%12.3e   imax,jmax =%5d%5d\n",
              iint, vamax, imax, jmax);
    error_status = 1;
  }
}

/* advance -- advance.f:6-59, without the file-driven forcing, print and output calls */
/* advance.f:6-59.  surface_forcing and lateral_bc (:14-18) read files in the reference; here they run when
 * the caller has registered the records those readers would deliver, and are skipped otherwise (the
 * constant-forcing test cases). */
void pomo_advance(pomo_tile *T) {
  pomo_get_time(T);
  if (T->frc_a[0][1] || T->frc_a[1][1] || T->frc_a[2][1]) pomo_surface_forcing(T);
  if (T->lat[1][0]) pomo_lateral_bc(T);
  pomo_lateral_viscosity(T);
  pomo_mode_interaction(T);
  for (iext = 1; iext <= isplit; iext++) pomo_mode_external(T);
  pomo_mode_internal(T);
  pomo_check_velocity(T);
}

void pomo_run(pomo_tile *T, int nsteps) {
  int n;
  for (n = 0; n < nsteps; n++) { iint = iint+1; pomo_advance(T); }
}
