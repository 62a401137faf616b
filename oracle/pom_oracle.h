/* oracle/pom_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's mode-split time step (advance.f + the solver.f
 * kernels + the bcond/bcondorl/restore_interior branches that run inside it).  It exists to CHECK
 * the HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (extpom_amd/, libpomgpu.so) never links, imports or calls anything in oracle/.
 *
 * Pinning: tests/test_oracle_vs_reference.py runs this restatement and the flang-compiled,
 * unmodified reference (oracle/_ref, built by oracle/build_ref.sh) on identical inputs and
 * requires BIT-IDENTICAL COMMON blocks; tests/golden/ holds digests of reference runs so the
 * pin also holds where /root/reference is absent.
 */
#ifndef POM_ORACLE_H
#define POM_ORACLE_H
#include <stddef.h>
#include "pom_layout.h"

#define POMO_MAXREC 8
#define POMO_NSCR 16

#define POMO_MAXFREC 16
typedef struct pomo_tile {
  int im_, jm_, kb_;         /* active extents im, jm, kb (blksiz) */
  int iml, jml;              /* leading dimensions im_local, jm_local */
  int nw_, ne_, ns_, nn_;    /* neighbour ranks n_west n_east n_south n_north, -1 = physical edge (blkpar) */
  int lramp;                 /* blklog */
  size_t n2, n3;             /* iml*jml, iml*jml*kb */
  pom_blkcon *con;           /* blkcon record (caller-owned) */
  double *blk1d, *blk2d, *blk3d, *bdry; /* COMMON-layout storage (caller-owned) */
  double *bd[80];            /* base pointer of each bdry member (filled by pomo_bind) */
  /* relaxation targets served to restore_interior, records 1..POMO_MAXREC, (im,jm,kb) each */
  const double *rec_t[POMO_MAXREC + 1], *rec_s[POMO_MAXREC + 1];
  /* halo-exchange hooks (NULL = single tile, i.e. all neighbours -1: parallel_mpi.f:171-236) */
  void (*exch2d)(void *user, double *a, int nx, int ny);
  void (*exch3d)(void *user, double *a, int nx, int ny, int nz);
  void *user;
  /* outputs of check_velocity */
  double vamax; int imax, jmax;
  /* scratch for the reference's automatic arrays */
  double *scr[POMO_NSCR];
  /* order2d_mpi / order3d_mpi (parallel_mpi.f:353-480): hand the caller column nx-2 and row ny-2 of a
   * (nx,ny[,nz]) array and take the west / south neighbour's into ghost_w (ny*nz values, j fastest) and
   * ghost_s (nx*nz values, i fastest); either ghost stays untouched on a physical edge.  NULL = one tile. */
  void (*order)(void *user, const double *a, int nx, int ny, int nz, double *ghost_w, double *ghost_s);
  /* records served to wind / heat / surface (what read_wind_pnetcdf etc. would return): kind 0 wind (wu,wv),
   * 1 heat (shf,swr), 2 surface (sst,sss); records 1..POMO_MAXFREC, (im,jm) each */
  const double *frc_a[3][POMO_MAXFREC + 1], *frc_b[3][POMO_MAXFREC + 1];
  /* records served to lateral_bc: the 20 arrays read_boundary_conditions_pnetcdf fills (bounds_forcing.f:610-613):
   * tbwf sbwf ubwf vbwf tbef sbef ubef vbef tbnf sbnf vbnf ubnf tbsf sbsf vbsf ubsf elw ele eln els, each in the
   * shape of the bdry member it lands in */
  const double *lat[POMO_MAXFREC + 1][20];
} pomo_tile;

/* bind storage; returns 0 or -1 on allocation failure */
int  pomo_bind(pomo_tile *t, int im, int jm, int kb, int iml, int jml, pom_blkcon *con,
               double *blk1d, double *blk2d, double *blk3d, double *bdry);
void pomo_release(pomo_tile *t);
size_t pomo_tile_size(void);

/* the reference's entry points, same names and argument meaning */
void pomo_advave(pomo_tile *t);
void pomo_advct(pomo_tile *t);
void pomo_advq(pomo_tile *t, double *qb, double *q, double *qf);
void pomo_advt1(pomo_tile *t, double *fb, double *f, double *fclim, double *ff);
void pomo_advt2(pomo_tile *t, double *fb, double *f, double *fclim, double *ff);
void pomo_advu(pomo_tile *t);
void pomo_advv(pomo_tile *t);
void pomo_baropg(pomo_tile *t);
void pomo_baropg_mcc(pomo_tile *t);   /* solver.f:943-1159 (npg = 2) */
typedef void (*pomo_order_fn)(void *user, const double *a, int nx, int ny, int nz, double *ghost_w, double *ghost_s);
void pomo_set_order(pomo_tile *t, pomo_order_fn fn);
void pomo_set_forcing_record(pomo_tile *t, int kind, int n, const double *a, const double *b);
void pomo_set_lateral_record(pomo_tile *t, int n, const double *const *arrays20);
void pomo_lateral_bc(pomo_tile *t);         /* bounds_forcing.f:593-868 */
void pomo_wind(pomo_tile *t);               /* bounds_forcing.f:871-912 */
void pomo_heat(pomo_tile *t);               /* bounds_forcing.f:915-960 */
void pomo_surface(pomo_tile *t);            /* bounds_forcing.f:963-983 */
void pomo_surface_forcing(pomo_tile *t);    /* advance.f:77-93 */
void pomo_dens(pomo_tile *t, double *si, double *ti, double *rhoo);
void pomo_profq(pomo_tile *t);
void pomo_proft(pomo_tile *t, double *f, double *wfsurf, double *fsurf, int nbc);
void pomo_profu(pomo_tile *t);
void pomo_profv(pomo_tile *t);
void pomo_vertvl(pomo_tile *t);
void pomo_realvertvl(pomo_tile *t);
void pomo_bcond(pomo_tile *t, int idx);
void pomo_bcondorl(pomo_tile *t, int idx);
void pomo_restore_interior(pomo_tile *t);
void pomo_get_time(pomo_tile *t);
void pomo_lateral_viscosity(pomo_tile *t);
void pomo_mode_interaction(pomo_tile *t);
void pomo_mode_external(pomo_tile *t);
void pomo_mode_internal(pomo_tile *t);
void pomo_check_velocity(pomo_tile *t);
/* domain_stats -- advance.f:644-756.  out = vtot, atot, mtot, stot, tavg, savg, eavg, ekin (the reference's
 * argument order).  sums_only != 0: this tile's partial sums before sum0d_mpi (tavg = sum tb*dvol, eavg =
 * sum et*darea, savg = 0); == 0: the single-task result (averages formed as on my_task 0).  The reference
 * uses the SUM intrinsic, whose order of additions is the compiler's: parity is to rounding, not bitwise. */
void pomo_domain_stats(pomo_tile *t, double *out, int sums_only);
/* hot-path sequence of advance (advance.f:6-59) for the step con->iint */
void pomo_advance(pomo_tile *t);
/* nsteps x { iint += 1; advance } */
void pomo_run(pomo_tile *t, int nsteps);
#endif
