! pom_gpu_io.f90 -- write_output_pnetcdf and write_restart_pnetcdf (reference pom/io_pnetcdf.F:57-410, :1661-2083)
! without PnetCDF: the file names are built as the reference builds them, the files themselves (CDF-2, same
! dimensions / variables / attributes) are written by the library straight from the device state
! (pomgpu_write_output / pomgpu_write_restart); every rank writes its patch, rank 0 creates the file first.
! Link instead of the reference's two writers.  pomgpu_barrier_mpi is the integrator's one-liner
! (call mpi_barrier(pom_comm, ierr); nothing on a single rank): this file does not include mpif.h.
subroutine write_output_pnetcdf
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer :: nprint
  character(len=400) :: fname
  nprint = (iint+int(time0*86400./dti))/iprint
  write(fname, '(a,''out/'',a,''.'',i4.4,''.nc'')') trim(wrk_pth), trim(netcdf_file), nprint
  call pomgpu_write_file(fname, 0)
end subroutine

subroutine write_restart_pnetcdf
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer :: nprint
  character(len=400) :: fname
  nprint = (iint+int(time0*86400./dti))/irestart
  write(fname, '(a,''out/'',a,''.'',i4.4,''.nc'')') trim(wrk_pth), trim(write_rst_file), nprint
  call pomgpu_write_file(fname, 1)
end subroutine

subroutine pomgpu_write_file(fname, restart)
  use pomgpu_iface
  implicit none
  include 'pom.h'
  character(len=*), intent(in) :: fname
  integer, intent(in) :: restart
  type(pomgpu_file_meta) :: m
  character(kind=c_char, len=401), target :: cname
  character(kind=c_char, len=41), target :: ctitle
  character(kind=c_char, len=27), target :: cstart
  real(c_double), target :: st(8)
  integer(c_int) :: rc
  integer :: pass
  cname = trim(fname)//c_null_char
  ctitle = trim(title)//c_null_char
  cstart = trim(time_start)//c_null_char
  call pomgpu_push_con
  call domain_stats(st(1), st(2), st(3), st(4), st(5), st(6), st(7), st(8))   ! rank-reduced, as the reference's writer does
  if (my_task == 0) write(*,'(/''writing file '',a)') trim(fname)
  m%title = c_loc(ctitle); m%time_start = c_loc(cstart)
  m%im_global = im_global; m%jm_global = jm_global
  m%i0 = i_global(1); m%j0 = j_global(1)
  m%stats = c_loc(st)
  do pass = 1, 2                                  ! rank 0 lays the file out, then everybody else writes
    m%create = 0
    if (pass == 1 .and. my_task == 0) m%create = 1
    if ((pass == 1) .eqv. (my_task == 0)) then
      if (restart == 0) then
        rc = pomgpu_write_output(pom_ctx, c_loc(cname), m)
      else
        rc = pomgpu_write_restart(pom_ctx, c_loc(cname), m)
      end if
      if (rc /= 0) error_status = 1
    end if
    call pomgpu_barrier_mpi                        ! mpi_barrier(pom_comm) on several ranks
  end do
  ! The library returns once the file is laid out and a snapshot of its arrays is taken; a host thread writes it while the
  ! model goes on.  A RESTART file must be whole when this routine returns (the reference's writer is synchronous and
  ! collective, io_pnetcdf.F:1661-2083: a run that dies in the next step must find it), so it is joined here and its
  ! status becomes the run's; an output file is joined by the next write or by pomgpu_host_finalize.
  if (restart /= 0) then
    if (pomgpu_io_wait(pom_ctx) /= 0) error_status = 1
    call pomgpu_barrier_mpi
  end if
end subroutine
