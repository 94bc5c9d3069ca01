!-----------------------------------------------------------------------
! TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
!
! C-callable harness around the *unmodified* reference ocean mixed layer
! routine oml / omladf (src/omlsubs.F:47-236, 244-763), SURVEY 8 row f1.
! Our own code; the reference modules are compiled where they lie by
! oracle/build_ref.sh.  The step sequence restated here is that of the
! reference main program: oml, qgostep, ocinvq, ocqbdy
! (src/q-gcm.F:1232-1249) and the leapfrog averaging incl. sst
! (src/q-gcm.F:1328-1366).
!-----------------------------------------------------------------------
module qgcm_ref_oml
  use iso_c_binding
  use parameters
  use occonst
  use ochomog
  use ocstate
  use intrfac
  use radiate, only : rrcpoc
  use monitor, only : cfraoc, centoc
  use omlsubs
  use qgosubs
  use ocisubs
  use vorsubs
  use valsubs
  implicit none

contains

  ! compile-time boundary options of this build of the reference
  subroutine ref_oml_flags(sb, nb) bind(C, name='ref_oml_flags')
    integer(c_int), intent(out) :: sb, nb
    sb = 0
    nb = 0
#ifdef sb_hflux
    sb = 1
#endif
#ifdef nb_hflux
    nb = 1
#endif
  end subroutine ref_oml_flags

  ! run-time parameters the mixed layer reads (input.params / q-gcm.F:438)
  subroutine ref_oml_init(hmoc_in, toc1, toc2, st2d_in, st4d_in, ycexp_in, rrcpoc_in, &
                          tsbdy_in, tnbdy_in) bind(C, name='ref_oml_init')
    real(c_double), value :: hmoc_in, toc1, toc2, st2d_in, st4d_in, ycexp_in, rrcpoc_in, tsbdy_in, tnbdy_in
    hmoc = hmoc_in
    toc(1) = toc1
    toc(2) = toc2
    st2d = st2d_in
    st4d = st4d_in
    ycexp = ycexp_in
    rrcpoc = rrcpoc_in
    tsbdy = tsbdy_in
    tnbdy = tnbdy_in
  end subroutine ref_oml_init

  subroutine ref_oml_set(sst_in, sstm_in, fnet_in, wekto_in, taux_in, tauy_in) bind(C, name='ref_oml_set')
    real(c_double), intent(in) :: sst_in(nxto,nyto), sstm_in(nxto,nyto), fnet_in(nxto,nyto), wekto_in(nxto,nyto)
    real(c_double), intent(in) :: taux_in(nxpo,nypo), tauy_in(nxpo,nypo)
    sst = sst_in
    sstm = sstm_in
    fnetoc = fnet_in
    wekto = wekto_in
    tauxo = taux_in
    tauyo = tauy_in
  end subroutine ref_oml_set

  ! out: sst, sstm, entoc; scal = (xon(1), cfraoc, centoc, enisoc(1), eninoc(1))
  subroutine ref_oml_get(sst_out, sstm_out, ent_out, scal) bind(C, name='ref_oml_get')
    real(c_double), intent(out) :: sst_out(nxto,nyto), sstm_out(nxto,nyto), ent_out(nxpo,nypo), scal(5)
    sst_out = sst
    sstm_out = sstm
    ent_out = entoc
    scal = 0.0d0
    scal(1) = xon(1)
    scal(2) = cfraoc
    scal(3) = centoc
#ifdef cyclic_ocean
    scal(4) = enisoc(1)
    scal(5) = eninoc(1)
#endif
  end subroutine ref_oml_get

  subroutine ref_oml() bind(C, name='ref_oml')
    call oml
  end subroutine ref_oml

  ! n ocean steps with the mixed layer switched on, from 1-based ocean step s0
  subroutine ref_steps_oml(s0, n) bind(C, name='ref_steps_oml')
    integer(c_int), value :: s0, n
    integer :: s, i, j, k
    do s = s0, s0 + n - 1
      call oml
      call qgostep
      call ocinvq
      call ocqbdy (qo, po)
      if ( mod(s-1, 25).eq.0 ) then
        ! src/q-gcm.F:1328-1366 (ocean part)
        do k = 1, nlo
          do j = 1, nypo
            do i = 1, nxpo
              qo(i,j,k) = 0.5d0*( qo(i,j,k) + qom(i,j,k) )
              po(i,j,k) = 0.5d0*( po(i,j,k) + pom(i,j,k) )
            enddo
          enddo
        enddo
        do j = 1, nyto
          do i = 1, nxto
            sst(i,j) = 0.5d0*( sst(i,j) + sstm(i,j) )
          enddo
        enddo
        do k = 1, nlo-1
          dpioc(k) = 0.5d0*( dpioc(k) + dpiocp(k) )
        enddo
#ifdef cyclic_ocean
        ocncs = 0.5d0*( ocncs + ocncsp )
        ocncn = 0.5d0*( ocncn + ocncnp )
#endif
      endif
    enddo
  end subroutine ref_steps_oml

  ! valids (src/valsubs.F:43-627): only the verdict is observable, its extremes are local variables
  subroutine ref_valids(ok, dtop_in) bind(C, name='ref_valids')
    integer(c_int), intent(out) :: ok
    real(c_double), intent(in) :: dtop_in(nxpo,nypo)
    logical :: solnok
    dtopoc = dtop_in
    solnok = .true.
    call valids (solnok)
    ok = 0
    if (solnok) ok = 1
  end subroutine ref_valids

  ! Restart dump in the wire format of SUBROUTINE resave (src/q-gcm.F:3053-3088; that routine lives in the main
  ! program file and cannot be linked here, so its WRITE sequence for an ocean_only build is restated): unformatted
  ! sequential records  tyrs | po,pom | sst,sstm | ast,astm | hmixa,hmixam.
  subroutine ref_write_restart(fname, nchar, tyrs_in) bind(C, name='ref_write_restart')
    integer(c_int), value :: nchar
    character(kind=c_char), intent(in) :: fname(nchar)
    real(c_double), value :: tyrs_in
    character(len=nchar) :: f
    integer :: i
    double precision :: tyrs
    do i = 1, nchar
      f(i:i) = fname(i)
    enddo
    tyrs = tyrs_in
    open (77, file=f, form='unformatted', status='replace')
    write (77) tyrs
    write (77) po,pom
    write (77) sst,sstm
    write (77) ast,astm
    write (77) hmixa,hmixam
    close (77)
  end subroutine ref_write_restart

  subroutine ref_atmos_dims(nxa, nya) bind(C, name='ref_atmos_dims')
    integer(c_int), intent(out) :: nxa, nya
    nxa = nxta
    nya = nyta
  end subroutine ref_atmos_dims

end module qgcm_ref_oml
