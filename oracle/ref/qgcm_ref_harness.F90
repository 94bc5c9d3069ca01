!-----------------------------------------------------------------------
! TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
!
! C-callable harness around the *unmodified* reference routines of
! jinkakei/q-gcm.  This file is our own code: it USEs the reference's
! modules (compiled where they lie under /root/reference/src by
! oracle/build_ref.sh, objects only into oracle/_ref/) and exposes them
! through bind(C) entry points so that python (ctypes) can
!   * generate the golden vectors under tests/golden/ and
!   * time the true reference Fortran+FFTPACK path as bench.py's
!     cpu_baseline (kind "reference").
!
! The reference main program (src/q-gcm.F) is not callable, so the few
! lines of grid set-up it performs inline are restated here, each with
! its citation:
!   grid / yporel              src/q-gcm.F:411-441
!   eigmod call                src/q-gcm.F:456-457
!   tridiagonal coefficients   src/q-gcm.F:932-954
!   homsol / constr / qcomp    src/q-gcm.F:976, 711, 719-731
!   time loop + LF averaging   src/q-gcm.F:1243-1249, 1328-1366
!   sponge-layer ramp r_spl    src/q-gcm.F:1154-1168   (builds with -Dsponge_layer_k247 only)
!-----------------------------------------------------------------------
module qgcm_ref_harness
  use iso_c_binding
  use parameters
  use occonst
  use ochomog
  use ocstate
  use ocisubs
  use qgosubs
  use vorsubs
  use intsubs
  use conhoms
  use eigmode
  implicit none

  double precision, parameter :: PI_ = 3.14159265358979324D0
  double precision, parameter :: TWOPI_ = 6.28318530717958648D0

contains

  subroutine ref_dims(nx, ny, nl, cyc) bind(C, name='ref_dims')
    integer(c_int), intent(out) :: nx, ny, nl, cyc
    nx = nxpo
    ny = nypo
    nl = nlo
#ifdef cyclic_ocean
    cyc = 1
#else
    cyc = 0
#endif
  end subroutine ref_dims

  subroutine ref_params(f0, bet) bind(C, name='ref_params')
    real(c_double), intent(out) :: f0, bet
    f0 = fnot
    bet = beta
  end subroutine ref_params

  ! Set run-time parameters, then perform the init sequence of the
  ! reference main program for the ocean (see header for citations).
  subroutine ref_init(dxo_in, dto_in, delek_in, bccooc_in, ah2_in, ah4_in, &
                      hoc_in, gpoc_in, ddyn_in) bind(C, name='ref_init')
    real(c_double), value :: dxo_in, dto_in, delek_in, bccooc_in
    real(c_double), intent(in) :: ah2_in(nlo), ah4_in(nlo), hoc_in(nlo), &
                                  gpoc_in(nlo-1), ddyn_in(nxpo,nypo)
    integer :: i, j, k, i1
    double precision :: dxa, dya, yla

    dxo = dxo_in
    dto = dto_in
    delek = delek_in
    bccooc = bccooc_in
    do k = 1, nlo
      ah2oc(k) = ah2_in(k)
      ah4oc(k) = ah4_in(k)
      hoc(k) = hoc_in(k)
    enddo
    do k = 1, nlo-1
      gpoc(k) = gpoc_in(k)
    enddo

    ! src/q-gcm.F:380-431
    dxa = ndxr*dxo
    dya = dxa
    yla = nyta*dya
    dyo = dxo
    hdxom1 = 0.5d0/dxo
    dxom2 = 1.0d0/(dxo*dxo)
    xlo = nxto*dxo
    ylo = nyto*dyo
    do i = 1, nxpo
      xpo(i) = (i-1)*dxo + (nx1-1)*dxa
    enddo
    do i = 1, nxto
      xto(i) = xpo(i) + 0.5d0*dxo
    enddo
    do j = 1, nypo
      ypo(j) = (ny1-1)*dya + (j-1)*dyo
      yporel(j) = ypo(j) - 0.5d0*yla
    enddo
    do j = 1, nyto
      yto(j) = ypo(j) + 0.5d0*dyo
      ytorel(j) = yto(j) - 0.5d0*yla
    enddo
    ! src/q-gcm.F:435-441
    rdxof0 = 1.0d0/(dxo*fnot)
    tdto = 2.0d0*dto
    hto = 0.0d0
    do k = 1, nlo
      hto = hto + hoc(k)
    enddo

    ! src/q-gcm.F:456-457
    call eigmod (nlo, gpoc, hoc, 'Ocean', amatoc, cphsoc, rdefoc, rdm2oc, &
                 ctl2moc, ctm2loc)

    ! dynamic topography is an input of the path (topset is out of scope)
    do j = 1, nypo
      do i = 1, nxpo
        ddynoc(i,j) = ddyn_in(i,j)
        dtopoc(i,j) = 0.0d0
      enddo
    enddo
    davgoc = 0.0d0

    ! forcing defaults, src/q-gcm.F:877-903
    do j = 1, nypo
      do i = 1, nxpo
        entoc(i,j) = 0.0d0
        wekpo(i,j) = 0.0d0
      enddo
    enddo
    do k = 1, nlo-1
      xon(k) = 0.0d0
    enddo
#ifdef cyclic_ocean
    do k = 1, nlo-1
      enisoc(k) = 0.0d0
      eninoc(k) = 0.0d0
    enddo
    do k = 1, nlo
      ajisoc(k) = 0.0d0
      ajinoc(k) = 0.0d0
    enddo
    txisoc = 0.0d0
    txinoc = 0.0d0
    bdrins = 0.0d0
    bdrinn = 0.0d0
#endif

    ! src/q-gcm.F:932-954
    aoc = 1.0d0/( dyo*dyo )
#ifdef cyclic_ocean
    do i = 2, nxto/2
      i1 = 2*i - 1
      bd2oc(i1-1) = -2.0d0*aoc &
                   + 2.0d0*dxom2*( cos( (i-1)*TWOPI_/nxto ) - 1.0d0 )
      bd2oc( i1 ) = bd2oc(i1-1)
    enddo
    bd2oc(  1 ) = -2.0d0*aoc
    bd2oc(nxto) = -2.0d0*aoc - 4.0d0*dxom2
    call drffti (nxto, oftwrk)
#else
    do i = 2, nxto
      bd2oc(i-1) = -2.0d0*aoc &
                  + 2.0d0*dxom2*( cos( (i-1)*PI_/nxto ) - 1.0d0 )
    enddo
    bd2oc(nxto) = 0.0d0
    call dsinti (nxto-1, oftwrk)
#endif

#ifdef sponge_layer_k247
    ! src/q-gcm.F:1154-1168: a Gaussian ramp in the distance to the nearer zonal boundary plus - unless the option
    ! nospl_in_ewbdy_k247 ("N-S boundary only", what a periodic channel needs) is defined - one in the distance to the
    ! nearer meridional boundary
    do j = 1, nypo
      do i = 1, nxpo
        r_spl(i,j) = 0.0d0 &
                   + exp( -2.0d0 * PI_ * ( ( 0.5d0 * dyo * dble(nypo) &
                     - abs( dyo * dble(j) - 0.5d0 * dyo * dble(nypo) ) ) / ( l_spl ) )**2.0d0 )
#ifndef nospl_in_ewbdy_k247
        r_spl(i,j) = r_spl(i,j) &
                   + exp( -2.0d0 * PI_ * ( ( 0.5d0 * dxo * dble(nxpo) &
                     - abs( dxo * dble(i) - 0.5d0 * dxo * dble(nxpo) ) ) / ( l_spl ) )**2.0d0 )
#endif
      enddo
    enddo
#endif

    ! src/q-gcm.F:976.  In a coupled build homsol also does the atmosphere, whose constants
    ! ref_atm_init sets: there the caller runs ref_homsol after both initialisations.
#ifdef ocean_only
    call homsol
#endif
  end subroutine ref_init

  ! the sponge-layer ramp and constants of a -Dsponge_layer_k247 build (on = 0 and zeros otherwise)
  subroutine ref_get_sponge(on, rspl_out, c1, lspl) bind(C, name='ref_get_sponge')
    integer(c_int), intent(out) :: on
    real(c_double), intent(out) :: rspl_out(nxpo,nypo), c1, lspl
#ifdef sponge_layer_k247
    on = 1
    rspl_out = r_spl
    c1 = c1_spl
    lspl = l_spl
#else
    on = 0
    rspl_out = 0.0d0
    c1 = 0.0d0
    lspl = 0.0d0
#endif
  end subroutine ref_get_sponge

  ! Load pressures, then derive q and the constraint scalars exactly as
  ! the reference main program does at start-up (src/q-gcm.F:711-731).
  subroutine ref_set_p(po_in, pom_in) bind(C, name='ref_set_p')
    real(c_double), intent(in) :: po_in(nxpo,nypo,nlo), pom_in(nxpo,nypo,nlo)
    po = po_in
    pom = pom_in
    call constr
    call qcomp (qo, po, amatoc, yporel, dxom2, nxpo, nypo, nlo, ddynoc, nlo)
    call qcomp (qom,pom,amatoc, yporel, dxom2, nxpo, nypo, nlo, ddynoc, nlo)
    call ocqbdy (qo, po )
    call ocqbdy (qom,pom)
#ifdef cyclic_ocean
    call merqcy (qo, po,  amatoc, yporel, dxom2, nxpo, nypo, nlo, ddynoc, nlo)
    call merqcy (qom,pom, amatoc, yporel, dxom2, nxpo, nypo, nlo, ddynoc, nlo)
#endif
  end subroutine ref_set_p

  ! Raw overwrite of the four prognostic fields (no derivation).
  subroutine ref_set_state(po_in, pom_in, qo_in, qom_in) bind(C, name='ref_set_state')
    real(c_double), intent(in) :: po_in(nxpo,nypo,nlo), pom_in(nxpo,nypo,nlo), &
                                  qo_in(nxpo,nypo,nlo), qom_in(nxpo,nypo,nlo)
    po = po_in
    pom = pom_in
    qo = qo_in
    qom = qom_in
  end subroutine ref_set_state

  subroutine ref_get_state(po_out, pom_out, qo_out, qom_out) bind(C, name='ref_get_state')
    real(c_double), intent(out) :: po_out(nxpo,nypo,nlo), pom_out(nxpo,nypo,nlo), &
                                   qo_out(nxpo,nypo,nlo), qom_out(nxpo,nypo,nlo)
    po_out = po
    pom_out = pom
    qo_out = qo
    qom_out = qom
  end subroutine ref_get_state

  subroutine ref_set_forcing(wekpo_in, entoc_in, xon_in) bind(C, name='ref_set_forcing')
    real(c_double), intent(in) :: wekpo_in(nxpo,nypo), entoc_in(nxpo,nypo), xon_in(nlo-1)
    wekpo = wekpo_in
    entoc = entoc_in
    xon = xon_in
  end subroutine ref_set_forcing

  ! cyclic-only forcing line integrals (txisoc/txinoc from xforc,
  ! enisoc/eninoc from oml); ignored in the box build.
  subroutine ref_set_cyc_forcing(txis, txin, enis, enin) bind(C, name='ref_set_cyc_forcing')
    real(c_double), value :: txis, txin
    real(c_double), intent(in) :: enis(nlo-1), enin(nlo-1)
#ifdef cyclic_ocean
    txisoc = txis
    txinoc = txin
    enisoc = enis
    eninoc = enin
#endif
  end subroutine ref_set_cyc_forcing

  ! scal layout: dpioc(nlo-1), dpiocp(nlo-1), then (cyclic only)
  ! ocncs, ocncn, ocncsp, ocncnp (nlo each); box: zeros.
  subroutine ref_get_scalars(scal) bind(C, name='ref_get_scalars')
    real(c_double), intent(out) :: scal(2*(nlo-1)+4*nlo)
    integer :: k, o
    scal = 0.0d0
    do k = 1, nlo-1
      scal(k) = dpioc(k)
      scal(nlo-1+k) = dpiocp(k)
    enddo
#ifdef cyclic_ocean
    o = 2*(nlo-1)
    do k = 1, nlo
      scal(o+k) = ocncs(k)
      scal(o+nlo+k) = ocncn(k)
      scal(o+2*nlo+k) = ocncsp(k)
      scal(o+3*nlo+k) = ocncnp(k)
    enddo
#endif
  end subroutine ref_get_scalars

#ifdef ocean_only
  ! Ekman pumping of an ocean-only run from the wind stress: `call xforc` as the main program does once at start-up
  ! (src/q-gcm.F:820-826; src/xfosubs.F:566-683): wekto on the T grid, wekpo on the p grid.
  subroutine ref_xforc(tx, ty, wekto_out, wekpo_out) bind(C, name='ref_xforc')
    use intrfac, only : tauxo, tauyo
    use ocstate, only : wekto, wekpo
    use xfosubs, only : xforc
    real(c_double), intent(in) :: tx(nxpo,nypo), ty(nxpo,nypo)
    real(c_double), intent(out) :: wekto_out(nxto,nyto), wekpo_out(nxpo,nypo)
    tauxo = tx
    tauyo = ty
    call xforc
    wekto_out = wekto
    wekpo_out = wekpo
  end subroutine ref_xforc
#endif

  ! layer averages of the progress print-out / monitoring (pavgoc, qavgoc: src/monitor_diag.F:729-739 integrates po, qo
  ! with boundary weights 1/2 by its PRIVATE genint - that file is the netCDF module and is not built here - and
  ! multiplies by ocnorm = 1/(nxto*nyto), src/parameters_data.F:88).  The same trapezoid integral is the reference's
  ! public xintp (src/intsubs.f:78-133), which constr uses on the same arrays.
  subroutine ref_layer_avgs(pavg, qavg) bind(C, name='ref_layer_avgs')
    use intsubs, only : xintp
    real(c_double), intent(out) :: pavg(nlo), qavg(nlo)
    double precision :: pint, qint
    integer :: k
    do k = 1, nlo
      call xintp (pint, po(1,1,k), nxpo, nypo)
      call xintp (qint, qo(1,1,k), nxpo, nypo)
      pavg(k) = pint*ocnorm
      qavg(k) = qint*ocnorm
    enddo
  end subroutine ref_layer_avgs

  ! continuity monitors of the cyclic ocinvq (MODULE monitor, src/ocisubs.F:268-283); zeros in a box build
  subroutine ref_get_monitors(erm, emf) bind(C, name='ref_get_monitors')
    use monitor, only : ermaso, emfroc
    real(c_double), intent(out) :: erm(nlo-1), emf(nlo-1)
    erm = 0.0d0
    emf = 0.0d0
#ifdef cyclic_ocean
    erm = ermaso
    emf = emfroc
#endif
  end subroutine ref_get_monitors

  subroutine ref_set_scalars(scal) bind(C, name='ref_set_scalars')
    real(c_double), intent(in) :: scal(2*(nlo-1)+4*nlo)
    integer :: k, o
    do k = 1, nlo-1
      dpioc(k) = scal(k)
      dpiocp(k) = scal(nlo-1+k)
    enddo
#ifdef cyclic_ocean
    o = 2*(nlo-1)
    do k = 1, nlo
      ocncs(k) = scal(o+k)
      ocncn(k) = scal(o+nlo+k)
      ocncsp(k) = scal(o+2*nlo+k)
      ocncnp(k) = scal(o+3*nlo+k)
    enddo
#endif
  end subroutine ref_set_scalars

  ! Modal / geometric constants produced by eigmod and the grid set-up.
  subroutine ref_get_consts(amat, cl2m, cm2l, rdm2, bd2, ypr, aoc_out) bind(C, name='ref_get_consts')
    real(c_double), intent(out) :: amat(nlo,nlo), cl2m(nlo,nlo), cm2l(nlo,nlo), &
                                   rdm2(nlo), bd2(nxto), ypr(nypo), aoc_out
    amat = amatoc
    cl2m = ctl2moc
    cm2l = ctm2loc
    rdm2 = rdm2oc
    bd2 = bd2oc
    ypr = yporel
    aoc_out = aoc
  end subroutine ref_get_consts

  ! Homogeneous-solution products of homsol.
  ! box:    hom = ochom(nxpo,nypo,nlo-1); aux = [aipohs(nlo-1), cdiffo(nlo,nlo-1), cdhoc(nlo-1,nlo-1)]
  ! cyclic: hom = [pch1oc(nypo,nlo-1), pch2oc(nypo,nlo-1), pbhoc(nypo)];
  !         aux = [aipcho(nlo-1), hc1soc, hc2soc, hc1noc, hc2noc (nlo-1 each), hbsioc, aipbho]
  subroutine ref_get_homog(hom, aux) bind(C, name='ref_get_homog')
    real(c_double), intent(out) :: hom(*), aux(*)
    integer :: i, j, m, k, n
#ifdef cyclic_ocean
    n = 0
    do m = 1, nlo-1
      do j = 1, nypo
        n = n + 1
        hom(n) = pch1oc(j,m)
      enddo
    enddo
    do m = 1, nlo-1
      do j = 1, nypo
        n = n + 1
        hom(n) = pch2oc(j,m)
      enddo
    enddo
    do j = 1, nypo
      n = n + 1
      hom(n) = pbhoc(j)
    enddo
    n = 0
    do m = 1, nlo-1
      aux(n+m) = aipcho(m)
      aux(n+(nlo-1)+m) = hc1soc(m)
      aux(n+2*(nlo-1)+m) = hc2soc(m)
      aux(n+3*(nlo-1)+m) = hc1noc(m)
      aux(n+4*(nlo-1)+m) = hc2noc(m)
    enddo
    aux(5*(nlo-1)+1) = hbsioc
    aux(5*(nlo-1)+2) = aipbho
#else
    n = 0
    do m = 1, nlo-1
      do j = 1, nypo
        do i = 1, nxpo
          n = n + 1
          hom(n) = ochom(i,j,m)
        enddo
      enddo
    enddo
    n = 0
    do m = 1, nlo-1
      n = n + 1
      aux(n) = aipohs(m)
    enddo
    do k = 1, nlo-1
      do m = 1, nlo
        n = n + 1
        aux(n) = cdiffo(m,k)
      enddo
    enddo
    do m = 1, nlo-1
      do k = 1, nlo-1
        n = n + 1
        aux(n) = cdhoc(k,m)
      enddo
    enddo
#endif
  end subroutine ref_get_homog

  subroutine ref_qgostep() bind(C, name='ref_qgostep')
    call qgostep
  end subroutine ref_qgostep

  subroutine ref_ocinvq() bind(C, name='ref_ocinvq')
    call ocinvq
  end subroutine ref_ocinvq

  subroutine ref_ocqbdy() bind(C, name='ref_ocqbdy')
    call ocqbdy (qo, po)
  end subroutine ref_ocqbdy

  ! Leapfrog time-level averaging, src/q-gcm.F:1328-1366 (ocean part,
  ! without sst which is not on this path).
  subroutine ref_lf_average() bind(C, name='ref_lf_average')
    integer :: i, j, k
    do k = 1, nlo
      do j = 1, nypo
        do i = 1, nxpo
          qo(i,j,k) = 0.5d0*( qo(i,j,k)+qom(i,j,k) )
          po(i,j,k) = 0.5d0*( po(i,j,k)+pom(i,j,k) )
        enddo
      enddo
    enddo
    do k = 1, nlo-1
      dpioc(k) = 0.5d0*( dpioc(k) + dpiocp(k) )
    enddo
#ifdef cyclic_ocean
    do k = 1, nlo
      ocncs(k) = 0.5d0*( ocncs(k) + ocncsp(k) )
      ocncn(k) = 0.5d0*( ocncn(k) + ocncnp(k) )
    enddo
#endif
  end subroutine ref_lf_average

  ! n ocean steps starting at 1-based ocean step index s0, with the
  ! averaging of src/q-gcm.F:1328 applied after steps s with
  ! mod(s-1,25)==0 (nt = 1+(s-1)*nstr  =>  mod(nt-1,25*nstr)==0).
  subroutine ref_steps(s0, n) bind(C, name='ref_steps')
    integer(c_int), value :: s0, n
    integer :: s
    do s = s0, s0+n-1
      call qgostep
      call ocinvq
      call ocqbdy (qo, po)
      if ( mod(s-1,25).eq.0 ) call ref_lf_average
    enddo
  end subroutine ref_steps

  ! One Helmholtz solve through the reference solver (hsbxoc / hscyoc).
  subroutine ref_helmholtz(wrk, boc) bind(C, name='ref_helmholtz')
    real(c_double), intent(inout) :: wrk(nxpo,nypo)
    real(c_double), intent(in) :: boc(nxto)
#ifdef cyclic_ocean
    call hscyoc (wrk, boc)
#else
    call hsbxoc (wrk, boc)
#endif
  end subroutine ref_helmholtz

  subroutine ref_xintp(val, res) bind(C, name='ref_xintp')
    real(c_double), intent(in) :: val(nxpo,nypo)
    real(c_double), intent(out) :: res
    call xintp (res, val, nxpo, nypo)
  end subroutine ref_xintp

  ! FFTPACK DST-I of arbitrary length n (x has n+1 elements, the last
  ! is workspace - see src/ocisubs.F:458-459).
  subroutine ref_dsint(n, x) bind(C, name='ref_dsint')
    integer(c_int), value :: n
    real(c_double), intent(inout) :: x(n+1)
    double precision, allocatable :: ws(:)
    allocate(ws(3*(n+1)+15))
    call dsinti (n, ws)
    call dsint (n, x, ws)
    deallocate(ws)
  end subroutine ref_dsint

  ! FFTPACK real forward / backward transforms of length n.
  subroutine ref_drfft(n, x, dirn) bind(C, name='ref_drfft')
    integer(c_int), value :: n, dirn
    real(c_double), intent(inout) :: x(n)
    double precision, allocatable :: ws(:)
    allocate(ws(2*n+15))
    call drffti (n, ws)
    if ( dirn.ge.0 ) then
      call drfftf (n, x, ws)
    else
      call drfftb (n, x, ws)
    endif
    deallocate(ws)
  end subroutine ref_drfft

  ! eigmod on arbitrary (h, g') for pinning the modal constants.
  subroutine ref_eigmod(nl, gpr, h, amat, rdm2, cl2m, cm2l) bind(C, name='ref_eigmod')
    integer(c_int), value :: nl
    real(c_double), intent(in) :: gpr(nl-1), h(nl)
    real(c_double), intent(out) :: amat(nl,nl), rdm2(nl), cl2m(nl,nl), cm2l(nl,nl)
    double precision :: cph(nl), rdf(nl)
    call eigmod (nl, gpr, h, 'Ocean', amat, cph, rdf, rdm2, cl2m, cm2l)
  end subroutine ref_eigmod

end module qgcm_ref_harness
