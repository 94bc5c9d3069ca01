!-----------------------------------------------------------------------
! Stand-alone Fortran host for the ocean hot path.
!
! Keeps the call sequence of the reference time loop (src/q-gcm.F:1220-1255
! and 1328-1366, ocean_only + no_oml_k247):
!
!     do nt = 1, nsteps                      (atmosphere steps)
!       if ( mod(nt,nstr).eq.1 ) then
!         call qgostep ; call ocinvq ; call ocqbdy (qo, po)
!       endif
!       if ( mod(nt-1,25*nstr).eq.0 ) <average time levels>
!     enddo
!
! through the drop-in modules of qgcm_hip_shim.F90.  Start-up data
! (parameters, modal matrices, initial po/pom/qo/qom, forcing) come from a
! case file written by tools/write_case.py; the homogeneous solutions are
! computed here with the shim's hsbxoc exactly as homsol does
! (src/conhoms.F:549-611).
!
!   usage: qgcm_ocean_host <case.bin> <out.bin> <n_ocean_steps>
!-----------------------------------------------------------------------
program qgcm_ocean_host
  use iso_c_binding
  use parameters
  use occonst
  use ochomog
  use ocstate
  use ocisubs_data
  use qgosubs
  use ocisubs, only : ocinvq, hsbxoc
  use vorsubs_hip
  use qgcm_hip_iface
  use qgcm_hip_state
  implicit none

  character(len=512) :: fin, fout, arg
  integer :: nsteps_oc, nstr, nt, ntmax, m, k, i, j, u
  integer(c_int) :: hdr(4)
  double precision, allocatable :: boc(:)
  double precision :: t0, t1

  if (command_argument_count() < 3) then
    print *, 'usage: qgcm_ocean_host <case.bin> <out.bin> <n_ocean_steps>'
    stop 2
  endif
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, arg)
  read(arg, *) nsteps_oc

  ! ---- case file (stream, little endian, see tools/write_case.py) --------
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) hdr
  nxpo = hdr(1); nypo = hdr(2); nlo = hdr(3); nstr = hdr(4)
  nxto = nxpo - 1; nyto = nypo - 1
  allocate(yporel(nypo), gpoc(nlo-1), hoc(nlo), ah2oc(nlo), ah4oc(nlo), amatoc(nlo,nlo), rdm2oc(nlo), &
           ctl2moc(nlo,nlo), ctm2loc(nlo,nlo), ddynoc(nxpo,nypo), bd2oc(nxto), boc(nxto))
  allocate(po(nxpo,nypo,nlo), pom(nxpo,nypo,nlo), qo(nxpo,nypo,nlo), qom(nxpo,nypo,nlo), &
           entoc(nxpo,nypo), wekpo(nxpo,nypo))
  allocate(ochom(nxpo,nypo,nlo-1), aipohs(nlo-1), cdiffo(nlo,nlo-1), cdhoc(nlo-1,nlo-1), &
           xon(nlo-1), dpioc(nlo-1), dpiocp(nlo-1))
  read(u) fnot, beta, dxo, dto, delek, bccooc, aoc
  read(u) ah2oc, ah4oc, hoc, gpoc, amatoc, rdm2oc, ctl2moc, ctm2loc
  read(u) yporel, bd2oc, ddynoc
  read(u) po, pom, qo, qom, wekpo, entoc, xon, dpioc, dpiocp
  close(u)
  dyo = dxo; dxom2 = 1.0d0/(dxo*dxo); tdto = 2.0d0*dto
  xlo = nxto*dxo; ylo = nyto*dyo

  ! ---- homogeneous solutions, src/conhoms.F:549-611 ------------------------
  do m = 1, nlo-1
    do i = 1, nxto
      boc(i) = bd2oc(i) - rdm2oc(m+1)
    enddo
    ochom(:,:,m) = 1.0d0
    call hsbxoc (ochom(1,1,m), boc)
    ochom(:,:,m) = 1.0d0 + rdm2oc(m+1)*ochom(:,:,m)
    aipohs(m) = xintp_host(ochom(:,:,m))*dxo*dyo
  enddo
  do k = 1, nlo-1
    do m = 1, nlo
      cdiffo(m,k) = ctm2loc(m,k+1) - ctm2loc(m,k)
    enddo
    do m = 1, nlo-1
      cdhoc(k,m) = ( ctm2loc(m+1,k+1) - ctm2loc(m+1,k) )*aipohs(m)
    enddo
  enddo

  call qgcm_hip_push

  ! ---- time loop -----------------------------------------------------------
  call cpu_time(t0)
  ntmax = 1 + (nsteps_oc-1)*nstr
  do nt = 1, ntmax
    if ( mod(nt,nstr).eq.1 ) then
      call qgostep
      call ocinvq
      call ocqbdy (qo, po)
    endif
    if ( mod(nt-1,25*nstr).eq.0 ) then
      call qgcm_hip_check(qgcm_hip_lf_average(qgcm_hip_handle), 'lf_average')
    endif
  enddo
  call qgcm_hip_pull
  call cpu_time(t1)
  print '(a,i8,a,f10.3,a)', '  ocean steps: ', nsteps_oc, '   host time: ', t1-t0, ' s'
  print '(a,1p,2d24.15)', '  dpioc  = ', dpioc

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) po, pom, qo, qom, dpioc, dpiocp
  close(u)
  call qgcm_hip_shutdown

contains

  ! trapezoid area sum of a p-grid field (same weights as xintp, src/intsubs.f:78-133)
  double precision function xintp_host(v)
    double precision, intent(in) :: v(:,:)
    integer :: ii, jj, n1, n2
    double precision :: s, sumi
    n1 = size(v,1); n2 = size(v,2)
    s = 0.0d0
    do jj = 1, n2
      sumi = 0.5d0*v(1,jj)
      do ii = 2, n1-1
        sumi = sumi + v(ii,jj)
      enddo
      sumi = sumi + 0.5d0*v(n1,jj)
      if (jj == 1 .or. jj == n2) sumi = 0.5d0*sumi
      s = s + sumi
    enddo
    xintp_host = s
  end function xintp_host

end program qgcm_ocean_host
