! pom_gpu_host.f90 -- the hot-path subroutines of the reference under THEIR OWN NAMES, as thin
! Fortran wrappers over the C ABI.  Linking this file (plus pomgpu_iface.f90 and libpomgpu.so)
! instead of the reference's solver.f / advance.f hot routines makes `advance` run on the GPU;
! the reference's driver, initialisation, forcing and I/O keep calling the same names
! (reference pom/advance.f:6-59, pom/solver.f).
!
! State lives in HBM between calls.  pomgpu_upload_state / pomgpu_download_state move whole
! COMMON blocks (after initialisation / forcing updates, before print_section / output / restart:
! SURVEY 8b).  Scalars of blkcon that the host changes every step (iint, iext, time, ramp) are
! pushed before each call; error_status is pulled back after check_velocity, which is where the
! reference tests it (advance.f:631-637).

subroutine pomgpu_host_init(device)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer device
  type(pomgpu_dims) :: dm
  dm%im = im; dm%jm = jm; dm%kb = kb; dm%im_local = im_local; dm%jm_local = jm_local
  dm%n_west = n_west; dm%n_east = n_east; dm%n_south = n_south; dm%n_north = n_north
  if (pomgpu_create(pom_ctx, dm, int(device, c_int), c_null_ptr) /= 0) then
    error_status = 1
    write(6,'(/''Error: pomgpu_create failed (no GPU?)'')')
    stop 1
  end if
end subroutine

! Multi-GPU: what initialize_mpi / distribute_mpi set up for MPI (parallel_mpi.f:124-151, :34-122) is handed to the
! library once: the RCCL communicator (id128 = the 128 bytes rank 0 got from pomgpu_rccl_unique_id and the caller
! broadcast with MPI_Bcast over pom_comm) and the eight neighbour ranks.  From then on every exchange point of the hot
! path is served inside the library (pack, one grouped ncclSend/ncclRecv round, unpack, on its stream), and the 2-D
! external mode runs on a wide-halo copy of the tile (one exchange per internal step) when the tiles are wide enough.
! Call it after pomgpu_upload_state: the extended tile is sized by isplit (the library refuses with isplit = 0).
subroutine pomgpu_host_connect(id128)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  character(kind=c_char) :: id128(128)
  integer(c_int) :: nb(8), rc
  integer :: nranks, min_im, min_jm
  call pomgpu_host_neighbours(nb, nranks, min_im, min_jm)
  rc = pomgpu_rccl_init(pom_ctx, id128, int(my_task, c_int), int(nranks, c_int), nb, c_null_ptr)
  if (rc /= 0) then
    error_status = 1
    write(6,'(/''Error: pomgpu_rccl_init failed'')')
    return
  end if
  rc = pomgpu_set_wide_external(pom_ctx, 1_c_int, int(min_im, c_int), int(min_jm, c_int))   ! EINVAL = tiles too narrow: per-point exchanges stay
end subroutine

! The eight neighbour ranks of this tile in the C ABI's order W E S N SW SE NW NE, the number of tiles, and the extents of
! the smallest tile -- from what distribute_mpi left in blkpar / blksiz (parallel_mpi.f:54-119): the tiles are numbered row
! by row, nproc_x of them per row, the east- / north-most ones are the trimmed ones (:83-87, :98-102).
subroutine pomgpu_host_neighbours(nb, nranks, min_im, min_jm)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer(c_int), intent(out) :: nb(8)
  integer, intent(out) :: nranks, min_im, min_jm
  integer :: px, py, npx, npy
  npx = (im_global - 2 + im_local - 3) / (im_local - 2)        ! tiles in x, y (parallel_mpi.f:54-65)
  npy = (jm_global - 2 + jm_local - 3) / (jm_local - 2)
  px = mod(my_task, npx); py = my_task / npx
  nb(1) = n_west; nb(2) = n_east; nb(3) = n_south; nb(4) = n_north
  nb(5) = -1; nb(6) = -1; nb(7) = -1; nb(8) = -1                ! SW SE NW NE: the tiles across the corners
  if (px > 0 .and. py > 0) nb(5) = my_task - 1 - npx
  if (px < npx - 1 .and. py > 0) nb(6) = my_task + 1 - npx
  if (px > 0 .and. py < npy - 1) nb(7) = my_task - 1 + npx
  if (px < npx - 1 .and. py < npy - 1) nb(8) = my_task + 1 + npx
  nranks = npx * npy
  min_im = min(im_local, im_global - (npx - 1) * (im_local - 2))
  min_jm = min(jm_local, jm_global - (npy - 1) * (jm_local - 2))
end subroutine

! End of the run (the reference: finalize_mpi, pom.f:36): the output / restart file still being written behind the model's
! back is joined -- its status is the run's -- and the device state is released.
subroutine pomgpu_host_finalize
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_io_wait(pom_ctx) /= 0) then
    error_status = 1
    write(6,'(/''Error: an output / restart file could not be written'')')
  end if
  call pomgpu_destroy(pom_ctx)
  pom_ctx = c_null_ptr
end subroutine

subroutine pomgpu_upload_state
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer(c_int) :: lr, rc
  lr = 0
  if (lramp) lr = 1
  rc = pomgpu_upload(pom_ctx, c_loc(dz), c_loc(aam2d), c_loc(aam), c_loc(ele), c_loc(alpha), lr)
  rc = pomgpu_bind_host(pom_ctx, c_loc(aam2d), c_loc(aam))
  if (rc /= 0) error_status = 1
end subroutine

subroutine pomgpu_download_state
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer(c_int) :: rc
  rc = pomgpu_download(pom_ctx, c_loc(dz), c_loc(aam2d), c_loc(aam), c_loc(ele), c_loc(alpha))
  if (rc /= 0) error_status = 1
end subroutine

subroutine pomgpu_push_con
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer(c_int) :: lr, rc
  lr = 0
  if (lramp) lr = 1
  rc = pomgpu_set_con(pom_ctx, c_loc(alpha), lr)
end subroutine

! ---- orchestration (advance.f) ------------------------------------------------------------
subroutine lateral_viscosity
  use pomgpu_iface
  implicit none
  include 'pom.h'
  call pomgpu_push_con
  if (pomgpu_lateral_viscosity(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine mode_interaction
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_mode_interaction(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine mode_external
  use pomgpu_iface
  implicit none
  include 'pom.h'
  call pomgpu_push_con            ! iext is the host's DO variable (advance.f:27)
  if (pomgpu_mode_external(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine mode_internal
  use pomgpu_iface
  implicit none
  include 'pom.h'
  call pomgpu_push_con
  if (pomgpu_mode_internal(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine check_velocity
  use pomgpu_iface
  implicit none
  include 'pom.h'
  real(c_double) :: vamax
  integer(c_int) :: imax, jmax, rc
  rc = pomgpu_check_velocity(pom_ctx, vamax, imax, jmax)
  rc = pomgpu_get_con(pom_ctx, c_loc(alpha))     ! brings error_status back
end subroutine

! domain_stats (advance.f:644-756): the tile's sums come from the device; the reduction over ranks and
! the averages are the reference's own lines (sum0d_mpi / bcast0d_mpi stay the host's, parallel_mpi.f:125-151)
subroutine domain_stats(vtot, atot, mtot, stot, tavg, savg, eavg, ekin)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  double precision, intent(out) :: vtot, atot, mtot, stot, tavg, savg, eavg, ekin
  real(c_double) :: o(8)
  if (pomgpu_domain_stats(pom_ctx, o, 1_c_int) /= 0) error_status = 1
  vtot = o(1); atot = o(2); mtot = o(3); stot = o(4); tavg = o(5); savg = 0.d0; eavg = o(7); ekin = o(8)
  call sum0d_mpi(atot, 0)
  call sum0d_mpi(eavg, 0)
  call sum0d_mpi(vtot, 0)
  call sum0d_mpi(mtot, 0)
  call sum0d_mpi(stot, 0)
  call sum0d_mpi(tavg, 0)
  call sum0d_mpi(ekin, 0)
  if (my_task == 0) then
    if (atot /= 0) then
      eavg = eavg/atot
    else
      eavg = 0.
    end if
    if (vtot /= 0) then
      tavg = tavg/vtot
      savg = stot/vtot
    else
      tavg = 0.
      savg = 0.
    end if
  end if
  call bcast0d_mpi(atot, 0)
  call bcast0d_mpi(vtot, 0)
  call bcast0d_mpi(mtot, 0)
  call bcast0d_mpi(stot, 0)
  call bcast0d_mpi(tavg, 0)
  call bcast0d_mpi(savg, 0)
  call bcast0d_mpi(eavg, 0)
  call bcast0d_mpi(ekin, 0)
end subroutine

! ---- kernels (solver.f), reference signatures -----------------------------------------------
subroutine advave
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_advave(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine advct
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_advct(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine advq(qb, q, qf)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  double precision, target :: qb(im_local,jm_local,kb), q(im_local,jm_local,kb), qf(im_local,jm_local,kb)
  if (pomgpu_advq(pom_ctx, c_loc(qb), c_loc(q), c_loc(qf)) /= 0) error_status = 1
end subroutine
subroutine advt1(fb, f, fclim, ff)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  double precision, target :: fb(im_local,jm_local,kb), f(im_local,jm_local,kb)
  double precision, target :: fclim(im_local,jm_local,kb), ff(im_local,jm_local,kb)
  if (pomgpu_advt1(pom_ctx, c_loc(fb), c_loc(f), c_loc(fclim), c_loc(ff)) /= 0) error_status = 1
end subroutine
subroutine advt2(fb, f, fclim, ff)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  double precision, target :: fb(im_local,jm_local,kb), f(im_local,jm_local,kb)
  double precision, target :: fclim(im_local,jm_local,kb), ff(im_local,jm_local,kb)
  if (pomgpu_advt2(pom_ctx, c_loc(fb), c_loc(f), c_loc(fclim), c_loc(ff)) /= 0) error_status = 1
end subroutine
subroutine advu
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_advu(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine advv
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_advv(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine baropg
  use pomgpu_iface
  implicit none
  include 'pom.h'
  call pomgpu_push_con            ! ramp
  if (pomgpu_baropg(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine baropg_mcc
  use pomgpu_iface
  implicit none
  include 'pom.h'
  call pomgpu_push_con            ! ramp
  if (pomgpu_baropg_mcc(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine dens(si, ti, rhoo)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  double precision, target :: si(im_local,jm_local,kb), ti(im_local,jm_local,kb), rhoo(im_local,jm_local,kb)
  if (pomgpu_dens(pom_ctx, c_loc(si), c_loc(ti), c_loc(rhoo)) /= 0) error_status = 1
end subroutine
subroutine profq
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_profq(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine proft(f, wfsurf, fsurf, nbc)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  double precision, target :: f(im_local,jm_local,kb), wfsurf(im_local,jm_local), fsurf(im_local,jm_local)
  integer nbc
  if (pomgpu_proft(pom_ctx, c_loc(f), c_loc(wfsurf), c_loc(fsurf), int(nbc, c_int)) /= 0) error_status = 1
end subroutine
subroutine profu
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_profu(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine profv
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_profv(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine vertvl
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_vertvl(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine realvertvl
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_realvertvl(pom_ctx) /= 0) error_status = 1
end subroutine
subroutine bcond(idx)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer idx
  if (pomgpu_bcond(pom_ctx, int(idx, c_int)) /= 0) error_status = 1
end subroutine
subroutine bcondorl(idx)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer idx
  if (pomgpu_bcondorl(pom_ctx, int(idx, c_int)) /= 0) error_status = 1
end subroutine
subroutine restore_interior
  use pomgpu_iface
  implicit none
  include 'pom.h'
  call pomgpu_push_con
  if (pomgpu_restore_interior(pom_ctx) /= 0) error_status = 1
end subroutine
