! pom_gpu_forcing.f90 -- surface_forcing's three routines (reference pom/bounds_forcing.f:871-983) with the
! fields on the device: the wrappers keep the reference's decisions about WHEN a record is read and call the
! reference's own readers (read_wind_pnetcdf, read_heat_pnetcdf, read_surface_pnetcdf: io_pnetcdf.F:2912,
! 3110,3170 -- host I/O stays the host's); the record goes to the library, which shifts, loads and
! interpolates in HBM (pomgpu_wind / pomgpu_heat / pomgpu_surface).  lateral_bc (bounds_forcing.f:593-868)
! likewise: read_boundary_conditions_pnetcdf fills the host's "...f" boundary arrays as before, their 20
! addresses go to pomgpu_set_lateral_record, the depth integrals / shift / interpolation run on the device.
! Link instead of the reference's wind, heat, surface, lateral_bc.  (Not part of the stand-alone test driver,
! which has no file readers.)
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
    if (rc /= 0) error_status = 1
  end if
  if (iint.eq.1 .or. mod(iint+cont_bry,iwind).eq.0) then         ! :890-902
    if (iint.ne.iend) then
      n = (iint+cont_bry+iwind)/iwind+1
      call read_wind_pnetcdf(n, wu, wv)
      rc = pomgpu_set_forcing_record(pom_ctx, 0_c_int, int(n, c_int), c_loc(wu), c_loc(wv))
      if (rc /= 0) error_status = 1
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
    if (rc /= 0) error_status = 1
  end if
  if (iint.eq.1 .or. mod(iint+cont_bry,iheat).eq.0) then         ! :934-946
    if (iint.ne.iend) then
      n = (iint+cont_bry+iheat)/iheat+1
      call read_heat_pnetcdf(n, shf, swr)
      rc = pomgpu_set_forcing_record(pom_ctx, 1_c_int, int(n, c_int), c_loc(shf), c_loc(swr))
      if (rc /= 0) error_status = 1
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
    if (rc /= 0) error_status = 1
  end if
  call pomgpu_push_con
  if (pomgpu_surface(pom_ctx) /= 0) error_status = 1
end subroutine

subroutine lateral_bc
  use pomgpu_iface
  implicit none
  include 'pom.h'
  integer :: ibc, n
  integer(c_int) :: rc
  double precision :: tbc
  tbc = 1./24.
  ibc = int(tbc*86400.d0/dti)
  if (iint.eq.1) then                                            ! bounds_forcing.f:607-613
    n = (iint+cont_bry)/ibc+1
    call read_and_hand_over(n)
  end if
  if (iint.eq.1 .or. mod(iint+cont_bry,ibc).eq.0) then           ! :740, :754-758
    if (iint.ne.iend) then
      n = (iint+cont_bry+ibc)/ibc+1
      call read_and_hand_over(n)
    end if
  end if
  call pomgpu_push_con
  if (pomgpu_lateral_bc(pom_ctx) /= 0) error_status = 1
contains
  subroutine read_and_hand_over(n)
    integer, intent(in) :: n
    type(c_ptr) :: a(20)
    call read_boundary_conditions_pnetcdf(n, kb, tbwf, sbwf, ubwf, vbwf, tbef, sbef, ubef, vbef, &
                                          tbnf, sbnf, vbnf, ubnf, tbsf, sbsf, vbsf, ubsf, elw, ele, eln, els)
    a = (/ c_loc(tbwf), c_loc(sbwf), c_loc(ubwf), c_loc(vbwf), c_loc(tbef), c_loc(sbef), c_loc(ubef), c_loc(vbef), &
           c_loc(tbnf), c_loc(sbnf), c_loc(vbnf), c_loc(ubnf), c_loc(tbsf), c_loc(sbsf), c_loc(vbsf), c_loc(ubsf), &
           c_loc(elw), c_loc(ele), c_loc(eln), c_loc(els) /)
    rc = pomgpu_set_lateral_record(pom_ctx, int(n, c_int), a)
    if (rc /= 0) error_status = 1
  end subroutine
end subroutine
