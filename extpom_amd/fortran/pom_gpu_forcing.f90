! pom_gpu_forcing.f90 -- surface_forcing's three routines (reference pom/bounds_forcing.f:871-983) with the
! fields on the device: the wrappers keep the reference's decisions about WHEN a record is read and call the
! reference's own readers (read_wind_pnetcdf, read_heat_pnetcdf, read_surface_pnetcdf: io_pnetcdf.F:2912,
! 3110,3170 -- host I/O stays the host's); the record goes to the library, which shifts, loads and
! interpolates in HBM (pomgpu_wind / pomgpu_heat / pomgpu_surface).  Link instead of the reference's wind,
! heat, surface.  (Not part of the stand-alone test driver, which has no file readers.)
subroutine wind
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer :: iwind, n
  integer(c_int) :: rc
  double precision :: twind
  double precision, dimension(im,jm), target :: wu, wv
  twind = .125
  iwind = int(twind*86400.d0/dti)
  if (iint.eq.1) then                                            ! bounds_forcing.f:884-888
    n = (iint+cont_bry)/iwind+1
    call read_wind_pnetcdf(n, wu, wv)
    rc = pomgpu_set_forcing_record(pom_ctx, 0_c_int, int(n, c_int), c_loc(wu), c_loc(wv))
  end if
  if (iint.eq.1 .or. mod(iint+cont_bry,iwind).eq.0) then         ! :890-902
    if (iint.ne.iend) then
      n = (iint+cont_bry+iwind)/iwind+1
      call read_wind_pnetcdf(n, wu, wv)
      rc = pomgpu_set_forcing_record(pom_ctx, 0_c_int, int(n, c_int), c_loc(wu), c_loc(wv))
    end if
  end if
  call pomgpu_push_con                                           ! iint, time
  if (pomgpu_wind(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine heat
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer :: iheat, n
  integer(c_int) :: rc
  double precision :: theat
  double precision, dimension(im,jm), target :: shf, swr
  theat = .125
  iheat = int(theat*86400.d0/dti)
  if (iint.eq.1) then                                            ! :928-932
    n = (iint+cont_bry)/iheat+1
    call read_heat_pnetcdf(n, shf, swr)
    rc = pomgpu_set_forcing_record(pom_ctx, 1_c_int, int(n, c_int), c_loc(shf), c_loc(swr))
  end if
  if (iint.eq.1 .or. mod(iint+cont_bry,iheat).eq.0) then         ! :934-946
    if (iint.ne.iend) then
      n = (iint+cont_bry+iheat)/iheat+1
      call read_heat_pnetcdf(n, shf, swr)
      rc = pomgpu_set_forcing_record(pom_ctx, 1_c_int, int(n, c_int), c_loc(shf), c_loc(swr))
    end if
  end if
  call pomgpu_push_con
  if (pomgpu_heat(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine surface
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer :: isrf, n
  integer(c_int) :: rc
  double precision :: tsrf
  double precision, dimension(im,jm), target :: sst, sss
  tsrf = .125
  isrf = int(tsrf*86400.d0/dti)
  if (iint.eq.1 .or. mod(iint+cont_bry,isrf).eq.0) then          ! :975-980
    n = (iint+cont_bry)/isrf+1
    call read_surface_pnetcdf(n, sst, sss)
    rc = pomgpu_set_forcing_record(pom_ctx, 2_c_int, int(n, c_int), c_loc(sst), c_loc(sss))
  end if
  call pomgpu_push_con
  if (pomgpu_surface(pom_ctx) /= 0) error_status = 1
end subroutine
