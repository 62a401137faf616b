/* include/pomgpu.h -- C ABI of the MI355X hot path of extPOM's mode-split time step.
 *
 * The reference has no plugin registry: its boundary is the Fortran external-procedure ABI plus
 * COMMON blocks (reference pom/pom.f:17-19 calls `advance`; every hot routine is an argument-less
 * subroutine working on the blocks of pom.h_dist:46-640).  This library keeps that shape:
 *
 *   - state crosses the boundary as whole COMMON blocks in the reference's own layout
 *     (include/pom_layout.h), handed over by address and mirrored in HBM;
 *   - every hot-path routine of the reference has an entry point of the same name
 *     (pomgpu_<name>) and the same argument meaning; routines whose Fortran actual arguments are
 *     COMMON arrays (advq, advt1, advt2, dens, proft: advance.f:407-408,426-430,439-440,454) take
 *     the HOST address of that array -- exactly what the Fortran caller passes by reference -- and
 *     the library translates it to the device mirror;
 *   - errors follow the reference's convention: a routine sets error_status=1 in blkcon and
 *     reports on stderr (advance.f:118-119,631-637); in addition every entry point returns 0 on
 *     success and a negative pomgpu_status otherwise (HIP failures map to POMGPU_EHIP and also
 *     set error_status=1).
 *
 * All entry points are plain C: pointers, ints and doubles only.  One context = one tile = one GPU.
 * The library never frees or reallocates host arrays; it owns only its device mirrors and stream.
 */
#ifndef POMGPU_H
#define POMGPU_H
#include <stddef.h>
#include "pom_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pomgpu_ctx pomgpu_ctx;

enum pomgpu_status {
  POMGPU_OK = 0,
  POMGPU_EINVAL = -1,   /* bad argument / unsupported option (message in pomgpu_last_error) */
  POMGPU_EHIP = -2,     /* a HIP runtime call failed */
  POMGPU_ENOMEM = -3,
  POMGPU_ENODEV = -4    /* no usable GPU: the hot path has no CPU fallback */
};

/* Tile description: blksiz + blkpar of the reference (pom.h_dist:46-54,58-78). */
typedef struct pomgpu_dims {
  int im, jm, kb;                         /* active extents */
  int im_local, jm_local;                 /* leading dimensions of every array */
  int n_west, n_east, n_south, n_north;   /* neighbour ranks, -1 = physical edge (parallel_mpi.f:111-119) */
} pomgpu_dims;

/* ---- life cycle ------------------------------------------------------------------------- */
/* `stream` is a hipStream_t (or NULL for the library's own stream) passed as void* so that this
 * header needs no HIP include. */
int  pomgpu_create(pomgpu_ctx **ctx, const pomgpu_dims *dims, int device, void *stream);
void pomgpu_destroy(pomgpu_ctx *ctx);
const char *pomgpu_last_error(const pomgpu_ctx *ctx);
int  pomgpu_sync(pomgpu_ctx *ctx);
void *pomgpu_stream(pomgpu_ctx *ctx);
/* The stream the library is enqueueing on at this moment.  Outside a callback this is pomgpu_stream(); inside a transport
 * callback it is the stream of the message round being served -- pomgpu_stream(), or the library's second stream for the rounds
 * that run beside the kernels (the early part of the wide exchange, the rim rounds, wr): a mover that enqueues its copies must
 * enqueue them THERE (or complete them before it returns), else the unpack kernel that follows on that stream does not wait
 * for them. */
void *pomgpu_current_stream(pomgpu_ctx *ctx);

/* ---- state transfer (pomgpu_upload_state / pomgpu_download_state of SURVEY 8b) ---------- */
/* Whole blocks; any pointer may be NULL to skip that block.  `bdry` is the bdry block
 * (pom.h_dist:532-608).  `lramp` is blklog. */
int pomgpu_upload(pomgpu_ctx *ctx, const double *blk1d, const double *blk2d, const double *blk3d,
                  const double *bdry, const pom_blkcon *con, int lramp);
int pomgpu_download(pomgpu_ctx *ctx, double *blk1d, double *blk2d, double *blk3d, double *bdry,
                    pom_blkcon *con);
/* Single arrays, addressed by their slot in the block (enum pom_slot2d / pom_slot3d). */
int pomgpu_upload_2d(pomgpu_ctx *ctx, int slot2d, const double *host);
int pomgpu_upload_3d(pomgpu_ctx *ctx, int slot3d, const double *host);
int pomgpu_download_2d(pomgpu_ctx *ctx, int slot2d, double *host);
int pomgpu_download_3d(pomgpu_ctx *ctx, int slot3d, double *host);
/* blkcon only (iint, time, ramp ... change every step on the host side of the reference). */
int pomgpu_set_con(pomgpu_ctx *ctx, const pom_blkcon *con, int lramp);
int pomgpu_get_con(pomgpu_ctx *ctx, pom_blkcon *con);
/* Register the host base addresses of blk2d/blk3d so that array arguments given as host
 * addresses (Fortran by-reference actuals) can be translated to device mirrors. */
int pomgpu_bind_host(pomgpu_ctx *ctx, const double *host_blk2d, const double *host_blk3d);
/* Relaxation targets that the reference's restore_interior reads through
 * read_restore_ts_interior_pnetcdf(n,kb,tr,sr) (bounds_forcing.f:1040,1060): record n (1-based),
 * arrays dimensioned (im,jm,kb).  Copied to the device. */
int pomgpu_set_restore_record(pomgpu_ctx *ctx, int n, const double *tr, const double *sr);
/* Device address of a mirror (for halo exchange by the caller, e.g. RCCL send/recv).  An address of a 3-D array stays valid until
 * pomgpu_destroy -- pomgpu_tune_placement, which moves those arrays, refuses once one has been handed out. */
double *pomgpu_device_2d(pomgpu_ctx *ctx, int slot2d);
double *pomgpu_device_3d(pomgpu_ctx *ctx, int slot3d);

/* ---- halo exchange hook (replaces exchange2d_mpi / exchange3d_mpi, parallel_mpi.f:154-351) */
/* Called on the library's stream order: `dev` is the DEVICE address of the first level to
 * exchange, leading dimensions (im_local, jm_local), `nz` levels (1 for 2-D).  `count` arrays are
 * exchanged at one program point (the caller may batch them into one message per neighbour).
 * NULL hook = single tile (all neighbours -1: every exchange is a no-op, parallel_mpi.f:171). */
typedef void (*pomgpu_exchange_fn)(void *user, double *const *dev, const int *nz, int count);
int pomgpu_set_exchange(pomgpu_ctx *ctx, pomgpu_exchange_fn fn, void *user);

/* order2d_mpi / order3d_mpi of baropg_mcc (parallel_mpi.f:353-480, called at solver.f:958-959): the
 * 4th-order pressure gradient needs ONE more ghost column to the west and ghost row to the south.  The
 * library packs what the reference sends -- column im_local-2 / row jm_local-2 of d and of rho-rmean --
 * into device buffers and asks the hook to move them: `send_east` (n_east doubles) goes to the eastern
 * neighbour, which receives it as `recv_west`; `send_north` (n_north) to the northern one, received as
 * `recv_south`.  A tile without the respective neighbour skips that transfer (its recv buffer is not
 * read).  NULL hook = single tile. */
typedef void (*pomgpu_order_fn)(void *user, const double *send_east, int n_east, const double *send_north, int n_north,
                                double *recv_west, double *recv_south);
int pomgpu_set_order_exchange(pomgpu_ctx *ctx, pomgpu_order_fn fn, void *user);

/* ---- surface forcing on the device (bounds_forcing.f:871-983, called from advance.f:77-93) ---------------
 * wind and heat keep two records (…b, …f members of blk2d) and interpolate linearly in time; surface sets the
 * SST without interpolation.  The readers stay the host's (read_wind_pnetcdf, read_heat_pnetcdf,
 * read_surface_pnetcdf: io_pnetcdf.F:2912,3110,3170): the host hands the library the pair of (im,jm) fields
 * of record n -- kind 0 = wind (wu, wv), 1 = heat (shf, swr), 2 = surface (sst, sss) -- before the step whose
 * wind / heat / surface call asks for it; the last four records per kind are kept.  pomgpu_get_time must have
 * set the step's time.  A record that was not supplied: error_status = 1, POMGPU_EINVAL. */
int pomgpu_set_forcing_record(pomgpu_ctx *ctx, int kind, int n, const double *a, const double *b);
int pomgpu_wind(pomgpu_ctx *ctx);              /* bounds_forcing.f:871-912 */
int pomgpu_heat(pomgpu_ctx *ctx);              /* bounds_forcing.f:915-960 */
int pomgpu_surface(pomgpu_ctx *ctx);           /* bounds_forcing.f:963-983 */
int pomgpu_surface_forcing(pomgpu_ctx *ctx);   /* advance.f:77-93: wind, heat, surface */

/* lateral_bc on the device (bounds_forcing.f:593-868): records every 1/24 day, the same shift / load /
 * interpolate pattern on the open-boundary arrays of `bdry`, plus the depth integrals uab?f, vab?f, uabe, uabw,
 * vabn, vabs.  The host's reader (read_boundary_conditions_pnetcdf, io_pnetcdf.F:3393) fills 20 arrays: tbwf sbwf
 * ubwf vbwf tbef sbef ubef vbef tbnf sbnf vbnf ubnf tbsf sbsf vbsf ubsf elw ele eln els -- hand them over in this
 * order, each in the shape of the bdry member it lands in ((jm_local,kb), (im_local,kb), (jm_local), (im_local)). */
int pomgpu_set_lateral_record(pomgpu_ctx *ctx, int n, const double *const *arrays20);
int pomgpu_lateral_bc(pomgpu_ctx *ctx);

/* Pack / unpack helpers for the hook (one kernel launch per direction instead of one copy per
 * array and edge).  dir 0 = east/west phase, 1 = north/south phase.  pack: the edge the western
 * (southern) neighbour needs -- column 2 (row 2) of every array -- goes to `to_lo`, column im-1
 * (row jm-1) to `to_hi`, arrays concatenated, each nz x jm (nz x im) doubles, level-major.  unpack:
 * `from_lo` (sent by the western / southern neighbour) lands in column 1 (row 1), `from_hi` in
 * column im (row jm).  A NULL buffer skips that side.  Buffers are device memory. */
int pomgpu_halo_pack(pomgpu_ctx *ctx, double *const *dev, const int *nz, int count, int dir, double *to_lo, double *to_hi);
int pomgpu_halo_unpack(pomgpu_ctx *ctx, double *const *dev, const int *nz, int count, int dir, const double *from_lo,
                       const double *from_hi);
/* The same exchange in ONE round with up to eight neighbours (half the message rounds per exchange point):
 * buffer tables in the order W E S N SW SE NW NE, NULL = no such neighbour.  pack8: column 2 / im-1 (jm values
 * per level) to W / E, row 2 / jm-1 (im values) to S / N, the cells (2,2), (im-1,2), (2,jm-1), (im-1,jm-1) to
 * SW, SE, NW, NE.  unpack8: what W / E sent lands in column 1 / im over the rows no N/S neighbour's row owns,
 * what S / N sent in row 1 / jm over the columns no W/E neighbour's column owns, the diagonal cells in the
 * four corners -- the ghost cells end up exactly as after the reference's two phases. */
int pomgpu_halo_pack8(pomgpu_ctx *ctx, double *const *dev, const int *nz, int count, double *const *to);
int pomgpu_halo_unpack8(pomgpu_ctx *ctx, double *const *dev, const int *nz, int count, const double *const *from);

/* ---- the library's own exchange: transport (replaces the MPI layer under exchange2d/3d_mpi, order2d/3d_mpi) --
 * With a transport the library serves every exchange point itself -- pack8, ONE message round with up to eight
 * neighbours, unpack8, all on its stream, no host code in between -- and the hooks above are not needed.
 * neighbours8: ranks in the order W E S N SW SE NW NE, -1 = none (W E S N must agree with pomgpu_dims).
 *
 * pomgpu_rccl_init: the production transport, grouped ncclSend / ncclRecv (RCCL, xGMI between the GPUs of a node)
 * enqueued on the library's stream.  The reference side is parallel_mpi.f:124-151 (initialize_mpi: communicator,
 * rank, size): one rank obtains `id128` (128 bytes) from pomgpu_rccl_unique_id and distributes it with whatever
 * the host has (MPI_Bcast in the Fortran driver), then every rank calls pomgpu_rccl_init.  librccl is opened at
 * run time (`librccl_path`, NULL = "librccl.so"), so the library has no link-time dependency on it.
 *
 * pomgpu_set_transport: a callback mover for hosts without RCCL between the ranks (tests: ranks sharing one GPU,
 * host threads).  It must deliver send[d][0..scount[d]) to neighbour d -- which receives it as recv[OPP(d)] --
 * and fill recv[d][0..rcount[d]) with what neighbour d sent towards OPP(d); buffers are device memory, the work
 * must be ordered after what is already enqueued on pomgpu_current_stream() (the stream of the round: pomgpu_stream(), or the
 * library's second stream, whose earlier work has completed when the callback is called) and before what follows there.  fn == NULL removes
 * the transport (and the wide-halo external mode with it). */
typedef void (*pomgpu_transport_fn)(void *user, const double *const *send, const size_t *scount, double *const *recv,
                                    const size_t *rcount);
int pomgpu_set_transport(pomgpu_ctx *ctx, const int *neighbours8, pomgpu_transport_fn fn, void *user);
int pomgpu_rccl_available(const char *librccl_path);   /* POMGPU_OK if librccl opens and has every entry point; no GPU call, not collective */
int pomgpu_rccl_unique_id(void *id128, const char *librccl_path);
int pomgpu_rccl_init(pomgpu_ctx *ctx, const void *id128, int rank, int nranks, const int *neighbours8,
                     const char *librccl_path);
/* Message rounds on the library's SECOND stream (the early part of the wide exchange; the rim rounds: advct's edge lines, advx + advy + aam, w,
 * the turbulence arrays and T / S / rho behind the kernels that produce them, the two velocity exchanges that end mode_internal; wr -- nine of
 * a full step's ten rounds) are a decision every rank of
 * the decomposition must take alike: a rank that kept them on the first stream while its neighbours moved them would post
 * its rounds in another order on another communicator, and the job would hang.  pomgpu_rccl_init agrees on it itself,
 * BEFORE anything collective depends on a rank's own answer: one ncclAllReduce(min) over the first communicator of every
 * rank's (a) willingness to split a second communicator off (ncclCommSplit present, second stream created, neither
 * POMGPU_NO_OVERLAP nor POMGPU_NO_SIDE_COMM in the switches its context was created under), (b) POMGPU_WR_MAIN unset, and
 * (c) a digest of the switches that choose which message rounds exist -- ranks started with different switch sets are all
 * refused with a message (POMGPU_EHIP) instead of hanging later; only if ALL ranks are willing does any of them enter
 * ncclCommSplit, and a second all-reduce settles whether every split succeeded.  With a callback mover the library cannot
 * reach the other ranks: side rounds stay OFF until the host has reduced pomgpu_transport_side_capable() (this rank: 0 =
 * cannot, 1 = can but wants wr on the main stream, 2 = can) to its minimum over ALL ranks and handed the result to
 * pomgpu_transport_side_agree() on every rank, after pomgpu_set_transport and before pomgpu_set_wide_external.  agreed = 0
 * is accepted with any transport.  pomgpu_switch_digest: the digest of (c), for such hosts to compare over their ranks. */
int pomgpu_transport_side_capable(pomgpu_ctx *ctx);
int pomgpu_transport_side_agree(pomgpu_ctx *ctx, int agreed);
/* A callback mover that ENQUEUES its work on pomgpu_current_stream() (device-to-device copies between contexts that share a GPU,
 * ordered by events) says so here (ordered = 1, after pomgpu_set_transport): the library then calls it for a round of the second
 * stream without completing that stream first -- such a mover runs beside the kernels like the RCCL transport does.  Default 0:
 * the callback finds the second stream's earlier work completed (movers that stage through the host). */
int pomgpu_transport_stream_ordered(pomgpu_ctx *ctx, int ordered);
unsigned pomgpu_switch_digest(pomgpu_ctx *ctx);
/* how many ranks RCCL itself reports for the communicator (ncclCommCount): 0 without the RCCL transport, -1 if unknown */
int pomgpu_rccl_nranks(pomgpu_ctx *ctx);
/* message rounds served by the transport since it was set (measurement) */
long pomgpu_exchange_rounds(pomgpu_ctx *ctx);
/* ... of them on the library's second stream (the early part of the wide exchange, the rim rounds, wr): beside kernels, not between them */
long pomgpu_exchange_rounds_side(pomgpu_ctx *ctx);

/* Wide-halo external mode (needs a transport).  It starts in pomgpu_mode_interaction and ends with the
 * pomgpu_mode_external call of the last substep (iext = isplit) -- the reference's own sequence in `advance`, and
 * pomgpu_advance / pomgpu_run, get it without change; in between the tile's 2-D arrays are not current (any other
 * entry point or download fetches them first).  The reference exchanges
 * 1-cell halos six times per external substep (advance.f:233,292-293,348-349; solver.f:60-61,70,111-112,121) --
 * ~180 of the ~200 message rounds of an internal step, each a few microseconds of data.  With on != 0 the 2-D part
 * of the step (advave and the tail of mode_interaction, all isplit substeps of mode_external) runs on a copy of
 * the tile extended by w = isplit + 4 cells towards every neighbour: ONE wide exchange per internal step brings
 * the neighbours' cells, the substeps run without any exchange (the band of stale cells at the rim of the extended
 * tile grows by one cell per substep, so after isplit substeps the tile and its ghost cells are still exact),
 * and the result is copied back -- the tile's arrays,
 * ghost cells included, hold bit for bit what the reference's exchanges would have left there.
 * Collective: every rank calls it with the same `on` and the smallest active extents (im, jm) over ALL tiles of
 * the decomposition; when a tile could be narrower than w + 3 cells every rank gets POMGPU_EINVAL and the
 * per-point exchanges stay in use. */
int pomgpu_set_wide_external(pomgpu_ctx *ctx, int on, int min_im, int min_jm);

/* ---- the hot path: orchestration (advance.f) -------------------------------------------- */
int pomgpu_get_time(pomgpu_ctx *ctx);            /* advance.f:62-75  */
/* With the wide-halo mode and side-stream rounds agreed, pomgpu_lateral_viscosity also starts the EARLY part of the wide
 * exchange on the second stream, and pomgpu_mode_interaction then sends only the late part: what the early part has moved (ua, va,
 * uab, vab, el, elb, d, the surface fluxes, wubot, wvbot, the open-boundary lines) is not moved again.  Between the two calls the host
 * therefore leaves the 2-D state alone: no forcing setters (pomgpu_set_forcing_record / _lateral_record, pomgpu_surface_forcing,
 * pomgpu_lateral_bc -- advance.f:14-18 runs them BEFORE lateral_viscosity), no writes through pomgpu_device_2d; pomgpu_upload and
 * pomgpu_upload_2d are allowed if EVERY rank makes them (they end the early part's validity, and pomgpu_mode_interaction then gathers
 * everything in one round -- on one rank only that would be unequal message counts).  The reference's own sequence (advance.f:14-21) and
 * pomgpu_advance satisfy this by construction.
 * pomgpu_lateral_viscosity and pomgpu_mode_internal also post the rim rounds (advct's edge lines; w, the turbulence arrays, T / S / rho and
 * the two velocity exchanges of mode_internal) on the second stream; whatever looks at the state afterwards waits for them by itself. */
int pomgpu_lateral_viscosity(pomgpu_ctx *ctx);   /* advance.f:96-141 */
int pomgpu_mode_interaction(pomgpu_ctx *ctx);    /* advance.f:144-202 */
/* advance.f:205-353; uses blkcon.iext.  One call per substep, as the reference makes them (advance.f:27-29).  PAIRING: on a
 * tile large enough for the two-substeps-per-pass kernel an ODD substep is held back (the call returns POMGPU_OK with nothing
 * launched) until the next call: if that is substep iext + 1 the two run as one pass and elf, uaf, vaf are stored for the
 * SECOND substep only (nobody reads the first's: the reference overwrites them in the next call); any other entry point, a
 * download or pomgpu_sync first runs the held substep alone (with its elf, uaf, vaf stored) under the scalars of the LAST
 * pomgpu_set_con -- do not change blkcon between the two calls of a pair.  If the held substep fails when it finally runs,
 * the context is marked failed: error_status = 1 and the next entry point / pomgpu_get_con reports it. */
int pomgpu_mode_external(pomgpu_ctx *ctx);
int pomgpu_mode_internal(pomgpu_ctx *ctx);       /* advance.f:356-537 */
/* advance.f:611-641; any of the out pointers may be NULL.  Synchronises the stream. */
int pomgpu_check_velocity(pomgpu_ctx *ctx, double *vamax, int *imax, int *jmax);
/* domain_stats (advance.f:644-756), the sums behind print_section, reduced on the device (deterministic
 * tree; no state download).  out[8] = vtot, atot, mtot, stot, tavg, savg, eavg, ekin in the reference's
 * argument order.  sums_only != 0: this tile's partial sums as they stand before sum0d_mpi (out[4] =
 * sum(tb*dvol), out[6] = sum(et*darea), out[5] = 0) -- the caller reduces over ranks and forms the averages
 * exactly as the reference does on my_task 0; sums_only == 0: the single-task result. */
int pomgpu_domain_stats(pomgpu_ctx *ctx, double *out, int sums_only);
/* ---- output and restart files without PnetCDF (io_pnetcdf.F:57-410, :1661-2083) -----------------------------
 * NetCDF "64-bit offset" (CDF-2) files with the reference's dimensions, variables, order, types and attribute
 * texts, written directly.  Every rank writes its (im,jm) patch at (i0,j0) = (i_global(1), j_global(1)) of the
 * global grid; the rank with create = 1 lays the file out first (the caller puts a barrier between it and the
 * others).  stats: the eight domain_stats values after the rank reduction (NULL: this tile's own).  The file
 * names are the caller's (the reference builds them from wrk_pth, netcdf_file / write_rst_file and the counter). */
typedef struct pomgpu_file_meta {
  const char *title;          /* blkchar title */
  const char *time_start;     /* blkchar time_start: "days since <time_start>" */
  int im_global, jm_global;
  int i0, j0;                 /* 1-based global indices of the tile's (1,1) */
  int create;                 /* 1: create the file and write header + scalars + own patch; 0: own patch only */
  const double *stats;        /* vtot atot mtot stot tavg savg eavg ekin, or NULL */
} pomgpu_file_meta;
/* The two writers return once the file is laid out and a snapshot of its arrays has been taken on the device; a host
 * thread brings the snapshot over on a copy stream and writes it while the model goes on.  pomgpu_io_wait joins it and
 * returns its status (pomgpu_sync, the next write and pomgpu_destroy call it too). */
int pomgpu_io_wait(pomgpu_ctx *ctx);
int pomgpu_write_output(pomgpu_ctx *ctx, const char *path, const pomgpu_file_meta *meta);    /* write_output_pnetcdf */
int pomgpu_write_restart(pomgpu_ctx *ctx, const char *path, const pomgpu_file_meta *meta);   /* write_restart_pnetcdf */

/* One internal step for the current blkcon.iint: get_time, [surface_forcing, lateral_bc -- once the host has
 * supplied forcing / lateral records, see below; skipped otherwise: constant forcing], lateral_viscosity,
 * mode_interaction, isplit x mode_external, mode_internal, check_velocity (advance.f:6-59 minus print and
 * output, which stay on the host).  Does not synchronise. */
int pomgpu_advance(pomgpu_ctx *ctx);
/* nsteps x { iint = iint+1; advance }  (pom.f:17-19).  Does not synchronise.  On several tiles with the second stream agreed
 * (pomgpu_rccl_init / pomgpu_transport_side_agree) every step but the last of the call leaves realvertvl (solver.f:2024-2067) and the
 * exchange of wr (:2055) to the step that follows: they run on the second stream beside that step's external substeps -- wr has
 * no reader on the hot path.  The last step of the call does both at once, so after the call the state is complete.  Like every
 * call that posts message rounds, every rank makes it with the same nsteps. */
int pomgpu_run(pomgpu_ctx *ctx, int nsteps);

/* Where the 3-D arrays live (no counterpart in the reference; optional).  The same kernels on the same bytes run up to 6 %
 * faster or slower with where in HBM their arrays lie; which places are good differs from process to process and cannot be seen from
 * user space, but is reproducible inside one process.  This call MEASURES it: blk3d and the 3-D scratch arrays move into one
 * allocation with room in front, up to max_try layouts (a start offset, and the arrays at their distance or 256 MiB further apart) are tried with one untimed and `steps` timed internal steps each (real
 * steps: the model advances by ntried x (steps + 1); results do not depend on where an array lives) and the fastest is kept.  ms_out / front_mib_out /
 * pad_mib_out (max_try entries each, may be NULL): what each trial took per step, how far into the allocation blk3d started and how much
 * further apart than their size its arrays were; *ntried = 0 when nothing was
 * tried (arrays below 64 MiB, no memory for the move).  On several tiles the trial steps post message rounds: every rank makes the
 * call alike (same steps, same max_try) and keeps its own best.  Needs the arrays' size plus the room again while it moves in.
 * The placement survives pomgpu_upload: a host that wants its run to start from an untouched state tunes on the initial state
 * and uploads it again (state and blkcon) before its first step.
 * The call MOVES the 3-D arrays (and frees the allocations they lived in): every address obtained from pomgpu_device_3d before it
 * would dangle, so the call refuses (POMGPU_EINVAL) once pomgpu_device_3d has been used on the context -- tune first, take addresses
 * afterwards.  If a move fails half way (a HIP error inside it) the mirrors no longer hold the state: error_status = 1 and every later
 * hot-path call on this context returns POMGPU_EHIP; destroy it and upload the state into a new one. */
int pomgpu_tune_placement(pomgpu_ctx *ctx, int steps, int max_try, double *ms_out, long *front_mib_out, long *pad_mib_out, int *ntried, int *kept);

/* ---- the hot path: kernels (solver.f, bounds_forcing.f), device-resident ---------------- */
int pomgpu_advave(pomgpu_ctx *ctx);              /* solver.f:6-198   */
int pomgpu_advct(pomgpu_ctx *ctx);               /* solver.f:201-408 */
int pomgpu_advq(pomgpu_ctx *ctx, const double *qb, const double *q, const double *qf);   /* :411-477 */
int pomgpu_advt1(pomgpu_ctx *ctx, const double *fb, const double *f, const double *fclim, const double *ff); /* :480-574 */
int pomgpu_advt2(pomgpu_ctx *ctx, const double *fb, const double *f, const double *fclim, const double *ff); /* :577-731 */
/* smol_adif(xmassflux,ymassflux,zwflux,ff) (solver.f:1880-1967) is the ONE routine of SURVEY 8(b)'s export list without an
 * entry point: its three flux arguments are automatic (stack) arrays of advt2 (solver.f:588-590), never COMMON arrays, so a
 * host has nothing to pass by reference; it runs inside pomgpu_advt2 (nitera > 1: k_advt2_smol; nitera = 1: fused away). */
int pomgpu_advu(pomgpu_ctx *ctx);                /* solver.f:734-788 */
int pomgpu_advv(pomgpu_ctx *ctx);                /* solver.f:791-845 */
int pomgpu_baropg(pomgpu_ctx *ctx);              /* solver.f:848-940 */
int pomgpu_baropg_mcc(pomgpu_ctx *ctx);          /* solver.f:943-1159 (npg = 2) */
int pomgpu_dens(pomgpu_ctx *ctx, const double *si, const double *ti, const double *rhoo); /* :1162-1209 */
int pomgpu_profq(pomgpu_ctx *ctx);               /* solver.f:1212-1538 */
int pomgpu_proft(pomgpu_ctx *ctx, const double *f, const double *wfsurf, const double *fsurf, int nbc); /* :1541-1683 */
int pomgpu_profu(pomgpu_ctx *ctx);               /* solver.f:1686-1780 */
int pomgpu_profv(pomgpu_ctx *ctx);               /* solver.f:1783-1877 */
int pomgpu_vertvl(pomgpu_ctx *ctx);              /* solver.f:1970-2021 */
int pomgpu_realvertvl(pomgpu_ctx *ctx);          /* solver.f:2024-2067 */
int pomgpu_bcond(pomgpu_ctx *ctx, int idx);      /* bounds_forcing.f:6-328   (idx 1,2,4,5,6) */
int pomgpu_bcondorl(pomgpu_ctx *ctx, int idx);   /* bounds_forcing.f:331-590 (idx 3,5) */
int pomgpu_restore_interior(pomgpu_ctx *ctx);    /* bounds_forcing.f:1023-1121 */

/* ---- measurement ------------------------------------------------------------------------ */
/* Per-kernel device time measured with HIP events on the library's stream.  Between
 * pomgpu_prof_begin and pomgpu_prof_end every launch is bracketed by events; pomgpu_prof_get
 * returns, for kernel number `k` (0 <= k < pomgpu_prof_count()), its name, launch count and
 * total milliseconds. */
int pomgpu_prof_begin(pomgpu_ctx *ctx);
/* Two phases are always bracketed as well, whatever the filter: "phase_step" (one pomgpu_advance, on the kernels' stream) and
 * "phase_external" (its isplit mode_external substeps); internal mode = the difference.
 * restrict the bracketing to one kernel (its name, e.g. "k_profq"); NULL or "" = every kernel */
int pomgpu_prof_filter(pomgpu_ctx *ctx, const char *kernel_name);
int pomgpu_prof_end(pomgpu_ctx *ctx);
int pomgpu_prof_count(pomgpu_ctx *ctx);
int pomgpu_prof_get(pomgpu_ctx *ctx, int k, const char **name, long *launches, double *total_ms);

/* developer switches are read from the environment ONCE, at pomgpu_create; this changes one of a live
 * context afterwards (name with or without "POMGPU_", value NULL = unset).  Tools and tests only. */
int pomgpu_debug_switch(pomgpu_ctx *ctx, const char *name, const char *value);
const char *pomgpu_version(void);
/* a digest of the sources this library was built from (measurement bookkeeping: profiles/traffic.json names the build its
 * counters were taken from) */
const char *pomgpu_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* POMGPU_H */
