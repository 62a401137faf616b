! pomgpu_iface.f90 -- ISO_C_BINDING view of the C ABI (include/pomgpu.h).
! One interface per entry point; array arguments are passed as the address of the COMMON array
! (c_loc), exactly what the reference's by-reference calls hand to its own kernels.
module pomgpu_iface
  use iso_c_binding
  implicit none
  type, bind(C) :: pomgpu_dims
    integer(c_int) :: im, jm, kb, im_local, jm_local, n_west, n_east, n_south, n_north
  end type
  type(c_ptr), save :: pom_ctx = c_null_ptr
  type, bind(C) :: pomgpu_file_meta               ! include/pomgpu.h
    type(c_ptr) :: title, time_start
    integer(c_int) :: im_global, jm_global, i0, j0, create
    type(c_ptr) :: stats
  end type
  interface
    integer(c_int) function pomgpu_create(ctx, dims, device, stream) bind(C, name='pomgpu_create')
      import; type(c_ptr) :: ctx; type(pomgpu_dims) :: dims; integer(c_int), value :: device; type(c_ptr), value :: stream
    end function
    subroutine pomgpu_destroy(ctx) bind(C, name='pomgpu_destroy')
      import; type(c_ptr), value :: ctx
    end subroutine
    integer(c_int) function pomgpu_upload(ctx, b1, b2, b3, bd, con, lramp) bind(C, name='pomgpu_upload')
      import; type(c_ptr), value :: ctx, b1, b2, b3, bd, con; integer(c_int), value :: lramp
    end function
    integer(c_int) function pomgpu_download(ctx, b1, b2, b3, bd, con) bind(C, name='pomgpu_download')
      import; type(c_ptr), value :: ctx, b1, b2, b3, bd, con
    end function
    integer(c_int) function pomgpu_set_con(ctx, con, lramp) bind(C, name='pomgpu_set_con')
      import; type(c_ptr), value :: ctx, con; integer(c_int), value :: lramp
    end function
    integer(c_int) function pomgpu_get_con(ctx, con) bind(C, name='pomgpu_get_con')
      import; type(c_ptr), value :: ctx, con
    end function
    integer(c_int) function pomgpu_bind_host(ctx, h2, h3) bind(C, name='pomgpu_bind_host')
      import; type(c_ptr), value :: ctx, h2, h3
    end function
    integer(c_int) function pomgpu_set_restore_record(ctx, n, tr, sr) bind(C, name='pomgpu_set_restore_record')
      import; type(c_ptr), value :: ctx, tr, sr; integer(c_int), value :: n
    end function
    ! the library's own exchange over RCCL (include/pomgpu.h "transport") and the wide-halo external mode
    integer(c_int) function pomgpu_rccl_unique_id(id128, librccl_path) bind(C, name='pomgpu_rccl_unique_id')
      import; character(kind=c_char) :: id128(128); type(c_ptr), value :: librccl_path
    end function
    integer(c_int) function pomgpu_rccl_init(ctx, id128, rank, nranks, neighbours8, librccl_path) bind(C, name='pomgpu_rccl_init')
      import; type(c_ptr), value :: ctx, librccl_path; character(kind=c_char) :: id128(128)
      integer(c_int), value :: rank, nranks; integer(c_int) :: neighbours8(8)
    end function
    integer(c_int) function pomgpu_set_wide_external(ctx, on, min_im, min_jm) bind(C, name='pomgpu_set_wide_external')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: on, min_im, min_jm
    end function
    ! hosts without RCCL between the ranks: the MPI mover of libpomgpu_mpi.so (extpom_amd/csrc/mpi_mover.c); fcomm = pom_comm
    integer(c_int) function pomgpu_mpi_mover_install(ctx, fcomm, neighbours8) bind(C, name='pomgpu_mpi_mover_install')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: fcomm; integer(c_int) :: neighbours8(8)
    end function
    integer(c_int) function pomgpu_mpi_mover_remove(ctx) bind(C, name='pomgpu_mpi_mover_remove')   ! before pomgpu_host_finalize: frees the pinned staging buffers
      import; type(c_ptr), value :: ctx
    end function
    integer(c_long) function pomgpu_exchange_rounds_side(ctx) bind(C, name='pomgpu_exchange_rounds_side')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_long) function pomgpu_exchange_rounds(ctx) bind(C, name='pomgpu_exchange_rounds')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function pomgpu_tune_placement(ctx, steps, max_try, ms_out, front_mib_out, pad_mib_out, ntried, kept) bind(C, name='pomgpu_tune_placement')
      import; type(c_ptr), value :: ctx, ms_out, front_mib_out, pad_mib_out, ntried, kept; integer(c_int), value :: steps, max_try   ! the five outputs may be c_null_ptr
    end function
    integer(c_int) function pomgpu_sync(ctx) bind(C, name='pomgpu_sync')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function pomgpu_check_velocity(ctx, vamax, imax, jmax) bind(C, name='pomgpu_check_velocity')
      import; type(c_ptr), value :: ctx; real(c_double) :: vamax; integer(c_int) :: imax, jmax
    end function
    integer(c_int) function pomgpu_set_forcing_record(ctx, kind, n, a, b) bind(C, name='pomgpu_set_forcing_record')
      import; type(c_ptr), value :: ctx, a, b; integer(c_int), value :: kind, n
    end function
    integer(c_int) function pomgpu_set_lateral_record(ctx, n, arrays) bind(C, name='pomgpu_set_lateral_record')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: n; type(c_ptr) :: arrays(20)
    end function
    integer(c_int) function pomgpu_io_wait(ctx) bind(C, name='pomgpu_io_wait')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function pomgpu_write_output(ctx, path, meta) bind(C, name='pomgpu_write_output')
      import; type(c_ptr), value :: ctx, path; type(pomgpu_file_meta) :: meta
    end function
    integer(c_int) function pomgpu_write_restart(ctx, path, meta) bind(C, name='pomgpu_write_restart')
      import; type(c_ptr), value :: ctx, path; type(pomgpu_file_meta) :: meta
    end function
    integer(c_int) function pomgpu_domain_stats(ctx, out, sums_only) bind(C, name='pomgpu_domain_stats')
      import; type(c_ptr), value :: ctx; real(c_double) :: out(8); integer(c_int), value :: sums_only
    end function
    integer(c_int) function pomgpu_advq(ctx, qb, q, qf) bind(C, name='pomgpu_advq')
      import; type(c_ptr), value :: ctx, qb, q, qf
    end function
    integer(c_int) function pomgpu_advt1(ctx, fb, f, fclim, ff) bind(C, name='pomgpu_advt1')
      import; type(c_ptr), value :: ctx, fb, f, fclim, ff
    end function
    integer(c_int) function pomgpu_advt2(ctx, fb, f, fclim, ff) bind(C, name='pomgpu_advt2')
      import; type(c_ptr), value :: ctx, fb, f, fclim, ff
    end function
    integer(c_int) function pomgpu_dens(ctx, si, ti, rhoo) bind(C, name='pomgpu_dens')
      import; type(c_ptr), value :: ctx, si, ti, rhoo
    end function
    integer(c_int) function pomgpu_proft(ctx, f, wfsurf, fsurf, nbc) bind(C, name='pomgpu_proft')
      import; type(c_ptr), value :: ctx, f, wfsurf, fsurf; integer(c_int), value :: nbc
    end function
    integer(c_int) function pomgpu_bcond(ctx, idx) bind(C, name='pomgpu_bcond')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: idx
    end function
    integer(c_int) function pomgpu_bcondorl(ctx, idx) bind(C, name='pomgpu_bcondorl')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: idx
    end function
  end interface
  ! the argument-less entry points share one abstract shape
  abstract interface
    integer(c_int) function pomgpu_noarg(ctx) bind(C)
      import; type(c_ptr), value :: ctx
    end function
  end interface
  procedure(pomgpu_noarg), bind(C, name='pomgpu_lateral_viscosity') :: pomgpu_lateral_viscosity
  procedure(pomgpu_noarg), bind(C, name='pomgpu_mode_interaction') :: pomgpu_mode_interaction
  procedure(pomgpu_noarg), bind(C, name='pomgpu_mode_external') :: pomgpu_mode_external
  procedure(pomgpu_noarg), bind(C, name='pomgpu_mode_internal') :: pomgpu_mode_internal
  procedure(pomgpu_noarg), bind(C, name='pomgpu_advave') :: pomgpu_advave
  procedure(pomgpu_noarg), bind(C, name='pomgpu_advct') :: pomgpu_advct
  procedure(pomgpu_noarg), bind(C, name='pomgpu_advu') :: pomgpu_advu
  procedure(pomgpu_noarg), bind(C, name='pomgpu_advv') :: pomgpu_advv
  procedure(pomgpu_noarg), bind(C, name='pomgpu_baropg') :: pomgpu_baropg
  procedure(pomgpu_noarg), bind(C, name='pomgpu_baropg_mcc') :: pomgpu_baropg_mcc
  procedure(pomgpu_noarg), bind(C, name='pomgpu_wind') :: pomgpu_wind
  procedure(pomgpu_noarg), bind(C, name='pomgpu_heat') :: pomgpu_heat
  procedure(pomgpu_noarg), bind(C, name='pomgpu_surface') :: pomgpu_surface
  procedure(pomgpu_noarg), bind(C, name='pomgpu_lateral_bc') :: pomgpu_lateral_bc
  procedure(pomgpu_noarg), bind(C, name='pomgpu_profq') :: pomgpu_profq
  procedure(pomgpu_noarg), bind(C, name='pomgpu_profu') :: pomgpu_profu
  procedure(pomgpu_noarg), bind(C, name='pomgpu_profv') :: pomgpu_profv
  procedure(pomgpu_noarg), bind(C, name='pomgpu_vertvl') :: pomgpu_vertvl
  procedure(pomgpu_noarg), bind(C, name='pomgpu_realvertvl') :: pomgpu_realvertvl
  procedure(pomgpu_noarg), bind(C, name='pomgpu_restore_interior') :: pomgpu_restore_interior
end module pomgpu_iface
