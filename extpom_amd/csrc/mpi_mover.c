/* mpi_mover.c -- a mover for pomgpu_set_transport (include/pomgpu.h) on hosts whose ranks talk MPI: what the
 * reference does with MPI_SEND / MPI_RECV inside exchange2d_mpi / exchange3d_mpi / order2d_mpi / order3d_mpi
 * (parallel_mpi.f:154-480), for the library's packed staging buffers -- ONE round = the up to eight staging buffers
 * device -> host, MPI_Irecv / MPI_Isend with every neighbour, MPI_Waitall, host -> device.
 *
 * It is the alternative INTEGRATION.md names for hosts without RCCL between the ranks (several ranks on one GPU,
 * GPUs without peer access): every round synchronises the kernels' stream and crosses PCIe twice, so it is the
 * integration and test path -- the production mover is pomgpu_rccl_init (transport.hip).  Built into its own
 * libpomgpu_mpi.so (the product library has no MPI dependency); the Fortran host calls pomgpu_mpi_mover_install
 * with its communicator handle (pom_comm, parallel_mpi.f:131) and the eight neighbour ranks.
 *
 * Side-stream rounds (pomgpu_transport_side_agree): the ranks' own answers are reduced here with MPI_Allreduce(MIN),
 * so that all of them or none move rounds to the library's second stream. */
#include <limits.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "pomgpu.h"

static const int OPP[8] = {1, 0, 3, 2, 7, 6, 5, 4};   /* W E S N SW SE NW NE: what leaves towards d arrives from OPP[d] */

typedef struct mpi_mover {
  MPI_Comm comm;
  int nbr[8];
  double *hs[8], *hr[8];      /* pinned host staging, grown on demand */
  size_t cs[8], cr[8];
  pomgpu_ctx *ctx;            /* the context this mover serves: its key in the table below */
  long rounds;
} mpi_mover;

static int grow(double **p, size_t *cap, size_t need) {
  if (need <= *cap) return 0;
  if (*p) (void)hipHostFree(*p);
  *p = NULL; *cap = 0;
  if (hipHostMalloc((void **)p, need * sizeof(double), 0) != hipSuccess) return 1;
  *cap = need;
  return 0;
}

static void mover_fn(void *user, const double *const *send, const size_t *scount, double *const *recv, const size_t *rcount) {
  mpi_mover *m = (mpi_mover *)user;
  MPI_Request req[16];
  MPI_Status sta[16];
  int nreq = 0, bad = 0;
  /* ordered after everything the library has enqueued on the stream of THIS round (pomgpu_current_stream: the kernels' stream, or the
   * library's second stream, whose rounds arrive here already completed -- the kernels' stream keeps running beside them) */
  if (hipStreamSynchronize((hipStream_t)pomgpu_current_stream(m->ctx)) != hipSuccess) bad = 1;
  for (int d = 0; d < 8 && !bad; d++) {
    if (m->nbr[d] < 0) continue;
    if (rcount[d]) {
      if (rcount[d] > (size_t)INT_MAX) { bad = 1; break; }     /* MPI counts are ints */
      bad |= grow(&m->hr[d], &m->cr[d], rcount[d]);
      if (!bad) MPI_Irecv(m->hr[d], (int)rcount[d], MPI_DOUBLE, m->nbr[d], OPP[d], m->comm, &req[nreq++]);
    }
  }
  for (int d = 0; d < 8 && !bad; d++) {
    if (m->nbr[d] < 0 || !scount[d]) continue;
    if (scount[d] > (size_t)INT_MAX) { bad = 1; break; }
    bad |= grow(&m->hs[d], &m->cs[d], scount[d]);
    if (!bad && hipMemcpy(m->hs[d], send[d], scount[d] * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
    if (!bad) MPI_Isend(m->hs[d], (int)scount[d], MPI_DOUBLE, m->nbr[d], d, m->comm, &req[nreq++]);
  }
  if (nreq) MPI_Waitall(nreq, req, sta);
  for (int d = 0; d < 8 && !bad; d++)
    if (m->nbr[d] >= 0 && rcount[d] && hipMemcpy(recv[d], m->hr[d], rcount[d] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) bad = 1;
  m->rounds++;
  if (bad) {                  /* a mover cannot return a status: a rank that lost a round must not go on with stale ghost cells */
    fprintf(stderr, "pomgpu_mpi_mover: a message round failed (HIP copy, host staging buffer, or a message of more than INT_MAX doubles)\n");
    MPI_Abort(m->comm, 1);
  }
}

static mpi_mover *g_installed[64];   /* movers this library has installed, by slot: pomgpu_mpi_mover_remove frees them */
static void mover_free(mpi_mover *m) {
  if (!m) return;
  for (int d = 0; d < 8; d++) { if (m->hs[d]) (void)hipHostFree(m->hs[d]); if (m->hr[d]) (void)hipHostFree(m->hr[d]); }
  free(m);
}
/* fcomm: the Fortran handle of the communicator (MPI_Comm_f2c); neighbours8 in the order W E S N SW SE NW NE, -1 = none.
 * Collective over the communicator.  Returns a pomgpu_status. */
int pomgpu_mpi_mover_install(pomgpu_ctx *ctx, int fcomm, const int *neighbours8) {
  if (!ctx || !neighbours8) return POMGPU_EINVAL;
  mpi_mover *m = (mpi_mover *)calloc(1, sizeof *m);
  if (!m) return POMGPU_ENOMEM;
  m->comm = MPI_Comm_f2c((MPI_Fint)fcomm);
  for (int d = 0; d < 8; d++) m->nbr[d] = neighbours8[d];
  m->ctx = ctx;
  int rc = pomgpu_set_transport(ctx, neighbours8, mover_fn, m);
  /* every rank reaches the reduction, whatever happened to it above */
  int mine = rc == POMGPU_OK ? pomgpu_transport_side_capable(ctx) : 0, all = 0, worst = 0, myrc = rc;
  MPI_Allreduce(&mine, &all, 1, MPI_INT, MPI_MIN, m->comm);
  MPI_Allreduce(&myrc, &worst, 1, MPI_INT, MPI_MIN, m->comm);
  if (worst != POMGPU_OK) {                                   /* some rank could not: nobody keeps a mover */
    if (rc == POMGPU_OK) (void)pomgpu_set_transport(ctx, NULL, NULL, NULL);
    mover_free(m);
    return rc != POMGPU_OK ? rc : POMGPU_EINVAL;
  }
  int slot = -1;
  for (int k = 0; k < 64 && slot < 0; k++) if (!g_installed[k]) slot = k;
  if (slot < 0) {                                             /* nowhere to remember it: no mover rather than one nobody can free */
    (void)pomgpu_set_transport(ctx, NULL, NULL, NULL);
    mover_free(m);
    fprintf(stderr, "pomgpu_mpi_mover_install: 64 movers are installed already in this process\n");
    return POMGPU_ENOMEM;
  }
  g_installed[slot] = m;
  return pomgpu_transport_side_agree(ctx, all);
}
/* takes the mover off the context (back to a single tile's behaviour) and frees its pinned staging buffers; call it before
 * pomgpu_destroy, after the last step.  Not collective. */
int pomgpu_mpi_mover_remove(pomgpu_ctx *ctx) {
  if (!ctx) return POMGPU_EINVAL;
  const int rc = pomgpu_set_transport(ctx, NULL, NULL, NULL);  /* synchronises what the mover may still be part of */
  for (int k = 0; k < 64; k++)                                 /* the mover of THIS context (two contexts may share a caller-supplied stream) */
    if (g_installed[k] && g_installed[k]->ctx == ctx) { mover_free(g_installed[k]); g_installed[k] = NULL; }
  return rc;
}
/* Fortran-callable without an interface block (by-reference arguments, trailing underscore) is not offered: the host
 * binds it through ISO_C_BINDING like the rest of the C ABI (extpom_amd/fortran/pomgpu_iface.f90). */
