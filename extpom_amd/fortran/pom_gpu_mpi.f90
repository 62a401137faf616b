! pom_gpu_mpi.f90 -- the MPI side of the Fortran host for hosts whose ranks talk MPI and not RCCL (several ranks on one GPU,
! GPUs without peer access): the communicator initialize_mpi made (parallel_mpi.f:124-151) and the neighbours distribute_mpi
! found (:34-122) are handed to the library's MPI mover (libpomgpu_mpi.so, extpom_amd/csrc/mpi_mover.c) -- from then on the
! library packs, moves and unpacks at every exchange point of the hot path by itself, and the 2-D external mode runs on a
! wide-halo copy of the tile.  The scalar reductions print_section / the output writers need (sum0d_mpi, bcast0d_mpi,
! parallel_mpi.f:482-513) and the barrier between the ranks' writes into one file are plain MPI calls on pom_comm; an
! integrator who keeps the reference's parallel_mpi.f links ITS sum0d_mpi / bcast0d_mpi instead of these two.
subroutine pomgpu_host_connect_mpi
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer(c_int) :: nb(8), rc
  integer :: nranks, min_im, min_jm
  call pomgpu_host_neighbours(nb, nranks, min_im, min_jm)
  rc = pomgpu_mpi_mover_install(pom_ctx, int(pom_comm, c_int), nb)
  if (rc /= 0) then
    error_status = 1
    write(6,'(/''Error: pomgpu_mpi_mover_install failed'')')
    return
  end if
  rc = pomgpu_set_wide_external(pom_ctx, 1_c_int, int(min_im, c_int), int(min_jm, c_int))   ! EINVAL (-1) = tiles too narrow: per-point exchanges stay
  if (rc /= 0 .and. rc /= -1) then                               ! anything else (device memory, HIP) is an error of the run
    error_status = 1
    write(6,'(/''Error: pomgpu_set_wide_external failed'')')
  end if
end subroutine

! end of a multi-rank run, before pomgpu_host_finalize: the mover leaves the context and gives its pinned buffers back
subroutine pomgpu_host_disconnect_mpi
  use pomgpu_iface
  implicit none
  include 'pom.h'
  if (pomgpu_mpi_mover_remove(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine sum0d_mpi(work, to)
  implicit none
  include 'mpif.h'
  include 'pom.h'
  double precision work, tmp
  integer to, ierr
  call mpi_reduce(work, tmp, 1, mpi_double_precision, mpi_sum, to, pom_comm, ierr)
  if (my_task == to) work = tmp
end subroutine

subroutine bcast0d_mpi(work, from)
  implicit none
  include 'mpif.h'
  include 'pom.h'
  double precision work
  integer from, ierr
  call mpi_bcast(work, 1, mpi_double_precision, from, pom_comm, ierr)
end subroutine

subroutine pomgpu_barrier_mpi
  implicit none
  include 'mpif.h'
  include 'pom.h'
  integer ierr
  call mpi_barrier(pom_comm, ierr)
end subroutine
