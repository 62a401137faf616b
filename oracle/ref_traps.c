/* oracle/ref_traps.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The hot-path sources of the reference (solver.f advance.f bounds_forcing.f
 * initialize.f) contain call sites into the reference's PnetCDF I/O layer
 * (io_pnetcdf.F).  That layer needs libpnetcdf, which this image lacks, so it is
 * NOT built and NOT imitated here.  The hot path never calls any of these entry
 * points (they are reached only from file-driven forcing, restart and output
 * code that the oracle harness does not invoke).  They exist only so that the
 * shared object resolves under RTLD_NOW; reaching one is a harness bug and
 * aborts loudly.
 *
 * ONE exception, because the reference's mode_internal calls restore_interior
 * unconditionally (advance.f:452) and restore_interior takes its relaxation
 * targets from a reader (bounds_forcing.f:1039-1065): the reader's entry point
 * read_restore_ts_interior_pnetcdf(n,k,tr,sr) is served from records the test
 * harness registered beforehand with pomref_set_restore_record().  It is an
 * INPUT hook -- it hands the reference the same synthetic relaxation targets
 * the oracle and the HIP path are given -- and performs no arithmetic.  Asking
 * for a record that was not registered aborts.
 *
 * The same goes for the three readers behind surface_forcing (wind, heat, surface:
 * bounds_forcing.f:871-983): read_wind_pnetcdf(n,wu,wv), read_heat_pnetcdf(n,shf,swr),
 * read_surface_pnetcdf(n,sst,sss) hand over the pair of (im,jm) fields the harness
 * registered for record n with pomref_set_forcing_record(kind, n, a, b, count) --
 * the time interpolation that follows is the reference's own.  Likewise read_boundary_conditions_pnetcdf
 * behind lateral_bc (bounds_forcing.f:610-613,755-758): the 20 boundary arrays of record n as registered with
 * pomref_set_lateral_record(n, arrays, counts).
 */
#include <stdio.h>
#include <stdlib.h>

#define TRAP(name)                                                            \
  void name(void) {                                                           \
    fprintf(stderr, "oracle/_ref: reference I/O entry '%s' is not built here " \
                    "(PnetCDF absent) and must not be reached\n", #name);     \
    abort();                                                                  \
  }

TRAP(read_grid_pnetcdf_)
TRAP(read_initial_ts_pnetcdf_)
TRAP(read_clim_ts_pnetcdf_)
TRAP(read_water_pnetcdf_)
TRAP(read_restart_pnetcdf_)
TRAP(write_output_pnetcdf_)
TRAP(write_restart_pnetcdf_)

#include <string.h>
#define POMREF_MAXREC 8
static const double *rec_t[POMREF_MAXREC + 1], *rec_s[POMREF_MAXREC + 1];
static size_t rec_n[POMREF_MAXREC + 1];

/* harness side: register record n (1-based); the pointers must stay valid */
void pomref_set_restore_record(int n, const double *tr, const double *sr, size_t count) {
  if (n < 1 || n > POMREF_MAXREC) { fprintf(stderr, "pomref: bad restore record %d\n", n); abort(); }
  rec_t[n] = tr; rec_s[n] = sr; rec_n[n] = count;
}

/* reference side: signature of io_pnetcdf.F's reader as called at bounds_forcing.f:1040,1060
 * (tr, sr are (im,jm,kb) automatic arrays of the caller) */
void read_restore_ts_interior_pnetcdf_(int *n, int *k, double *tr, double *sr) {
  (void)k;
  if (*n < 1 || *n > POMREF_MAXREC || !rec_t[*n]) {
    fprintf(stderr, "oracle/_ref: restore record %d was not registered by the harness\n", *n);
    abort();
  }
  memcpy(tr, rec_t[*n], rec_n[*n] * sizeof(double));
  memcpy(sr, rec_s[*n], rec_n[*n] * sizeof(double));
}


/* ---- surface forcing records: kind 0 = wind (wu, wv), 1 = heat (shf, swr), 2 = surface (sst, sss) ---- */
#define POMREF_MAXFREC 16
static const double *frc_a[3][POMREF_MAXFREC + 1], *frc_b[3][POMREF_MAXFREC + 1];
static size_t frc_n[3][POMREF_MAXFREC + 1];
void pomref_set_forcing_record(int kind, int n, const double *a, const double *b, size_t count) {
  if (kind < 0 || kind > 2 || n < 1 || n > POMREF_MAXFREC) { fprintf(stderr, "pomref: bad forcing record %d/%d\n", kind, n); abort(); }
  frc_a[kind][n] = a; frc_b[kind][n] = b; frc_n[kind][n] = count;
}
static void serve(int kind, const char *what, int n, double *a, double *b) {
  if (n < 1 || n > POMREF_MAXFREC || !frc_a[kind][n]) {
    fprintf(stderr, "oracle/_ref: %s record %d was not registered by the harness\n", what, n);
    abort();
  }
  memcpy(a, frc_a[kind][n], frc_n[kind][n] * sizeof(double));
  memcpy(b, frc_b[kind][n], frc_n[kind][n] * sizeof(double));
}
void read_wind_pnetcdf_(int *n, double *wu, double *wv) { serve(0, "wind", *n, wu, wv); }          /* io_pnetcdf.F:2912 */
void read_heat_pnetcdf_(int *n, double *shf, double *swr) { serve(1, "heat", *n, shf, swr); }      /* io_pnetcdf.F:3110 */
void read_surface_pnetcdf_(int *n, double *sst, double *sss) { serve(2, "surface", *n, sst, sss); } /* io_pnetcdf.F:3170 */


/* ---- lateral boundary records: the 20 arrays of read_boundary_conditions_pnetcdf (io_pnetcdf.F:3393) ---- */
static const double *lat_a[POMREF_MAXFREC + 1][20];
static size_t lat_n[POMREF_MAXFREC + 1][20];
void pomref_set_lateral_record(int n, const double *const *arrays, const size_t *counts) {
  if (n < 1 || n > POMREF_MAXFREC) { fprintf(stderr, "pomref: bad lateral record %d\n", n); abort(); }
  for (int a = 0; a < 20; a++) { lat_a[n][a] = arrays[a]; lat_n[n][a] = counts[a]; }
}
void read_boundary_conditions_pnetcdf_(int *n, int *k, double *t_w, double *s_w, double *u_w, double *v_w, double *t_e, double *s_e,
                                       double *u_e, double *v_e, double *t_n, double *s_n, double *v_n, double *u_n, double *t_s,
                                       double *s_s, double *v_s, double *u_s, double *e_w, double *e_e, double *e_n, double *e_s) {
  double *dst[20] = {t_w, s_w, u_w, v_w, t_e, s_e, u_e, v_e, t_n, s_n, v_n, u_n, t_s, s_s, v_s, u_s, e_w, e_e, e_n, e_s};
  (void)k;
  if (*n < 1 || *n > POMREF_MAXFREC || !lat_a[*n][0]) {
    fprintf(stderr, "oracle/_ref: lateral boundary record %d was not registered by the harness\n", *n);
    abort();
  }
  for (int a = 0; a < 20; a++) memcpy(dst[a], lat_a[*n][a], lat_n[*n][a] * sizeof(double));
}
