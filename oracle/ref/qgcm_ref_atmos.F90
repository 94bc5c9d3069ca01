!-----------------------------------------------------------------------
! TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
!
! C-callable harness around the *unmodified* reference atmosphere path
! (SURVEY 8 row f3, BASELINE configs[3] "double_gyre_coupled"):
!     call qgastep          src/qgasubs.F:45-166 (+ atadif 174-317)
!     call atinvq           src/atisubs.F:60-290  (+ hscyat 298-395)
!     call atqzbd (qa, pa)  src/vorsubs.F:396-480
! as the reference main program calls them once per atmospheric step
! (src/q-gcm.F:1259-1268), plus the atmospheric half of the leapfrog
! averaging block (src/q-gcm.F:1370-1404).  Our own code: it USEs the
! reference's modules, compiled where they lie by oracle/build_ref.sh in
! a *coupled* build (neither -Docean_only nor -Datmos_only), and only
! calls them.  The few lines of grid set-up that the main program does
! inline are restated with their citation:
!   atmospheric grid / yparel      src/q-gcm.F:390-406, 436-441
!   eigmod call                    src/q-gcm.F:518-519
!   tridiagonal coefficients       src/q-gcm.F:961-972
!   constr / qcomp / atqzbd / merqcy at start-up   src/q-gcm.F:711, 738-749
!-----------------------------------------------------------------------
#ifndef ocean_only
module qgcm_ref_atmos
  use iso_c_binding
  use parameters
  use atconst
  use athomog
  use atstate
  use atisubs
  use qgasubs
  use vorsubs
  use conhoms
  use eigmode
#ifndef atmos_only
  use qgosubs
  use ocisubs
  use ocstate, only : qo, po, qom, pom
  use ochomog, only : dpioc, dpiocp
#endif
  implicit none

  double precision, parameter, private :: TWOPI_ = 6.28318530717958648D0

contains

  subroutine ref_atm_dims(nx, ny, nl) bind(C, name='ref_atm_dims')
    integer(c_int), intent(out) :: nx, ny, nl
    nx = nxpa
    ny = nypa
    nl = nla
  end subroutine ref_atm_dims

  ! run-time parameters + the atmospheric part of the main program's init
  ! sequence, up to but excluding homsol (ref_homsol below: in a coupled
  ! build homsol does ocean and atmosphere in one call, so both halves
  ! must have been initialised first).
  subroutine ref_atm_init(dxa_in, dta_in, bccoat_in, ah4_in, hat_in, gpat_in, ddyn_in) bind(C, name='ref_atm_init')
    real(c_double), value :: dxa_in, dta_in, bccoat_in
    real(c_double), intent(in) :: ah4_in(nla), hat_in(nla), gpat_in(nla-1), ddyn_in(nxpa,nypa)
    integer :: i, j, k, i1

    dxa = dxa_in
    dta = dta_in
    bccoat = bccoat_in
    do k = 1, nla
      ah4at(k) = ah4_in(k)
      hat(k) = hat_in(k)
    enddo
    do k = 1, nla-1
      gpat(k) = gpat_in(k)
    enddo
    ! src/q-gcm.F:390-406
    dya = dxa
    hdxam1 = 0.5d0/dxa
    dxam2 = 1.0d0/(dxa*dxa)
    xla = nxta*dxa
    yla = nyta*dya
    do i = 1, nxpa
      xpa(i) = (i-1)*dxa
    enddo
    do i = 1, nxta
      xta(i) = xpa(i) + 0.5d0*dxa
    enddo
    do j = 1, nypa
      ypa(j) = (j-1)*dya
      yparel(j) = ypa(j) - 0.5d0*yla
    enddo
    do j = 1, nyta
      yta(j) = ypa(j) + 0.5d0*dya
      ytarel(j) = yta(j) - 0.5d0*yla
    enddo
    ! src/q-gcm.F:436-441
    rdxaf0 = 1.0d0/(dxa*fnot)
    tdta = 2.0d0*dta
    hta = 0.0d0
    do k = 1, nla
      hta = hta + hat(k)
    enddo
    ! src/q-gcm.F:518-519
    call eigmod (nla, gpat, hat, 'Atmosphere', amatat, cphsat, rdefat, rdm2at, ctl2mat, ctm2lat)
    ! dynamic topography is an input of the path (topset is out of scope)
    do j = 1, nypa
      do i = 1, nxpa
        ddynat(i,j) = ddyn_in(i,j)
        dtopat(i,j) = 0.0d0
        entat(i,j) = 0.0d0
        wekpa(i,j) = 0.0d0
      enddo
    enddo
    davgat = 0.0d0
    do k = 1, nla-1
      xan(k) = 0.0d0
      enisat(k) = 0.0d0
      eninat(k) = 0.0d0
    enddo
    do k = 1, nla
      ajisat(k) = 0.0d0
      ajinat(k) = 0.0d0
      ap5sat(k) = 0.0d0
      ap5nat(k) = 0.0d0
    enddo
    txisat = 0.0d0
    txinat = 0.0d0
    ! src/q-gcm.F:961-972
    aat = 1.0d0/( dya*dya )
    do i = 2, nxta/2
      i1 = 2*i - 1
      bd2at(i1-1) = -2.0d0*aat + 2.0d0*dxam2*( cos( (i-1)*TWOPI_/nxta ) - 1.0d0 )
      bd2at( i1 ) = bd2at(i1-1)
    enddo
    bd2at(  1 ) = -2.0d0*aat
    bd2at(nxta) = -2.0d0*aat - 4.0d0*dxam2
    call drffti (nxta, aftwrk)
  end subroutine ref_atm_init

  ! src/q-gcm.F:976 - after ref_init (ocean) and ref_atm_init
  subroutine ref_homsol() bind(C, name='ref_homsol')
    call homsol
  end subroutine ref_homsol

  ! Load pressures, then derive q and the constraint scalars exactly as the
  ! main program does at start-up (src/q-gcm.F:711, 738-749).  constr also
  ! recomputes the ocean's scalars from the ocean state as it stands; they are
  ! saved and restored so that the two halves of the harness stay independent.
  subroutine ref_atm_set_p(pa_in, pam_in) bind(C, name='ref_atm_set_p')
    real(c_double), intent(in) :: pa_in(nxpa,nypa,nla), pam_in(nxpa,nypa,nla)
#ifndef atmos_only
    double precision :: sv1(nlo-1), sv2(nlo-1)
    sv1 = dpioc
    sv2 = dpiocp
#endif
    pa = pa_in
    pam = pam_in
    call constr
#ifndef atmos_only
    dpioc = sv1
    dpiocp = sv2
#endif
    call qcomp (qa, pa, amatat, yparel, dxam2, nxpa, nypa, nla, ddynat, 1)
    call qcomp (qam,pam,amatat, yparel, dxam2, nxpa, nypa, nla, ddynat, 1)
    call atqzbd (qa, pa )
    call atqzbd (qam,pam)
    call merqcy (qa, pa,  amatat, yparel, dxam2, nxpa, nypa, nla, ddynat, 1)
    call merqcy (qam,pam, amatat, yparel, dxam2, nxpa, nypa, nla, ddynat, 1)
  end subroutine ref_atm_set_p

  subroutine ref_atm_set_state(pa_in, pam_in, qa_in, qam_in) bind(C, name='ref_atm_set_state')
    real(c_double), intent(in) :: pa_in(nxpa,nypa,nla), pam_in(nxpa,nypa,nla), &
                                  qa_in(nxpa,nypa,nla), qam_in(nxpa,nypa,nla)
    pa = pa_in
    pam = pam_in
    qa = qa_in
    qam = qam_in
  end subroutine ref_atm_set_state

  subroutine ref_atm_get_state(pa_out, pam_out, qa_out, qam_out) bind(C, name='ref_atm_get_state')
    real(c_double), intent(out) :: pa_out(nxpa,nypa,nla), pam_out(nxpa,nypa,nla), &
                                   qa_out(nxpa,nypa,nla), qam_out(nxpa,nypa,nla)
    pa_out = pa
    pam_out = pam
    qa_out = qa
    qam_out = qam
  end subroutine ref_atm_get_state

  ! what xforc / aml hand to the path: wekpa, entat (p grid), xan, and the
  ! boundary line integrals txisat/txinat, enisat/eninat
  subroutine ref_atm_set_forcing(wekpa_in, entat_in, xan_in, txis, txin, enis, enin) bind(C, name='ref_atm_set_forcing')
    real(c_double), intent(in) :: wekpa_in(nxpa,nypa), entat_in(nxpa,nypa), xan_in(nla-1)
    real(c_double), value :: txis, txin
    real(c_double), intent(in) :: enis(nla-1), enin(nla-1)
    wekpa = wekpa_in
    entat = entat_in
    xan = xan_in
    txisat = txis
    txinat = txin
    enisat = enis
    eninat = enin
  end subroutine ref_atm_set_forcing

  ! scal = dpiat(nla-1), dpiatp(nla-1), atmcs, atmcn, atmcsp, atmcnp (nla each)
  subroutine ref_atm_get_scalars(scal) bind(C, name='ref_atm_get_scalars')
    real(c_double), intent(out) :: scal(2*(nla-1)+4*nla)
    integer :: k, o
    do k = 1, nla-1
      scal(k) = dpiat(k)
      scal(nla-1+k) = dpiatp(k)
    enddo
    o = 2*(nla-1)
    do k = 1, nla
      scal(o+k) = atmcs(k)
      scal(o+nla+k) = atmcn(k)
      scal(o+2*nla+k) = atmcsp(k)
      scal(o+3*nla+k) = atmcnp(k)
    enddo
  end subroutine ref_atm_get_scalars

  subroutine ref_atm_set_scalars(scal) bind(C, name='ref_atm_set_scalars')
    real(c_double), intent(in) :: scal(2*(nla-1)+4*nla)
    integer :: k, o
    do k = 1, nla-1
      dpiat(k) = scal(k)
      dpiatp(k) = scal(nla-1+k)
    enddo
    o = 2*(nla-1)
    do k = 1, nla
      atmcs(k) = scal(o+k)
      atmcn(k) = scal(o+nla+k)
      atmcsp(k) = scal(o+2*nla+k)
      atmcnp(k) = scal(o+3*nla+k)
    enddo
  end subroutine ref_atm_set_scalars

  ! boundary sums of the last qgastep: ajisat, ajinat, ap5sat, ap5nat (nla each)
  subroutine ref_atm_get_bsums(b) bind(C, name='ref_atm_get_bsums')
    real(c_double), intent(out) :: b(4*nla)
    integer :: k
    do k = 1, nla
      b(k) = ajisat(k)
      b(nla+k) = ajinat(k)
      b(2*nla+k) = ap5sat(k)
      b(3*nla+k) = ap5nat(k)
    enddo
  end subroutine ref_atm_get_bsums

  ! continuity monitors of the last atinvq (MODULE monitor: ermasa, emfrat; src/atisubs.F:236-248)
  subroutine ref_atm_get_monitors(erm, emf) bind(C, name='ref_atm_get_monitors')
    use monitor, only : ermasa, emfrat
    real(c_double), intent(out) :: erm(nla-1), emf(nla-1)
    erm = ermasa
    emf = emfrat
  end subroutine ref_atm_get_monitors

  subroutine ref_atm_get_consts(amat, cl2m, cm2l, rdm2, bd2, ypr, aat_out) bind(C, name='ref_atm_get_consts')
    real(c_double), intent(out) :: amat(nla,nla), cl2m(nla,nla), cm2l(nla,nla), &
                                   rdm2(nla), bd2(nxta), ypr(nypa), aat_out
    amat = amatat
    cl2m = ctl2mat
    cm2l = ctm2lat
    rdm2 = rdm2at
    bd2 = bd2at
    ypr = yparel
    aat_out = aat
  end subroutine ref_atm_get_consts

  ! hom = [pch1at(nypa,nla-1), pch2at(nypa,nla-1), pbhat(nypa)];
  ! aux = [aipcha(nla-1), hc1sat, hc2sat, hc1nat, hc2nat (nla-1 each), hbsiat, aipbha]
  subroutine ref_atm_get_homog(hom, aux) bind(C, name='ref_atm_get_homog')
    real(c_double), intent(out) :: hom(*), aux(*)
    integer :: j, m, n
    n = 0
    do m = 1, nla-1
      do j = 1, nypa
        n = n + 1
        hom(n) = pch1at(j,m)
      enddo
    enddo
    do m = 1, nla-1
      do j = 1, nypa
        n = n + 1
        hom(n) = pch2at(j,m)
      enddo
    enddo
    do j = 1, nypa
      n = n + 1
      hom(n) = pbhat(j)
    enddo
    do m = 1, nla-1
      aux(m) = aipcha(m)
      aux((nla-1)+m) = hc1sat(m)
      aux(2*(nla-1)+m) = hc2sat(m)
      aux(3*(nla-1)+m) = hc1nat(m)
      aux(4*(nla-1)+m) = hc2nat(m)
    enddo
    aux(5*(nla-1)+1) = hbsiat
    aux(5*(nla-1)+2) = aipbha
  end subroutine ref_atm_get_homog

  subroutine ref_qgastep() bind(C, name='ref_qgastep')
    call qgastep
  end subroutine ref_qgastep

  subroutine ref_atinvq() bind(C, name='ref_atinvq')
    call atinvq
  end subroutine ref_atinvq

  subroutine ref_atqzbd() bind(C, name='ref_atqzbd')
    call atqzbd (qa, pa)
  end subroutine ref_atqzbd

  ! atmospheric half of the averaging block, src/q-gcm.F:1370-1404 (without
  ! ast / hmixa, which belong to the atmospheric mixed layer, not this path)
  subroutine ref_atm_lf_average() bind(C, name='ref_atm_lf_average')
    integer :: i, j, k
    do k = 1, nla
      do j = 1, nypa
        do i = 1, nxpa
          qa(i,j,k) = 0.5d0*( qa(i,j,k)+qam(i,j,k) )
          pa(i,j,k) = 0.5d0*( pa(i,j,k)+pam(i,j,k) )
        enddo
      enddo
    enddo
    do k = 1, nla-1
      dpiat(k) = 0.5d0*( dpiat(k) + dpiatp(k) )
    enddo
    do k = 1, nla
      atmcs(k) = 0.5d0*( atmcs(k) + atmcsp(k) )
      atmcn(k) = 0.5d0*( atmcn(k) + atmcnp(k) )
    enddo
  end subroutine ref_atm_lf_average

  ! n atmospheric steps nt = nt0 .. nt0+n-1 (1-based, the main program's nt):
  ! qgastep, atinvq, atqzbd, and the averaging when mod(nt-1,100).eq.0
  ! (src/q-gcm.F:1259-1268, 1370)
  subroutine ref_atm_steps(nt0, n) bind(C, name='ref_atm_steps')
    integer(c_int), value :: nt0, n
    integer :: nt
    do nt = nt0, nt0+n-1
      call qgastep
      call atinvq
      call atqzbd (qa, pa)
      if ( mod(nt-1,100).eq.0 ) call ref_atm_lf_average
    enddo
  end subroutine ref_atm_steps

#ifndef atmos_only
  ! The main loop of a coupled run with the forcing held fixed (xforc, oml,
  ! aml are not on this path): src/q-gcm.F:1220-1268 and the two averaging
  ! blocks :1328-1404.  nt = nt0 .. nt0+n-1, ocean stepped when mod(nt,nstr).eq.1.
  subroutine ref_coupled_steps(nt0, n, nstr) bind(C, name='ref_coupled_steps')
    integer(c_int), value :: nt0, n, nstr
    integer :: nt, i, j, k
    do nt = nt0, nt0+n-1
      if ( mod(nt,nstr).eq.1 ) then
        call qgostep
        call ocinvq
        call ocqbdy (qo, po)
      endif
      call qgastep
      call atinvq
      call atqzbd (qa, pa)
      if ( mod(nt-1,25*nstr).eq.0 ) then
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
      endif
      if ( mod(nt-1,100).eq.0 ) call ref_atm_lf_average
    enddo
  end subroutine ref_coupled_steps
#endif

  ! One Helmholtz solve through the reference solver hscyat.
  subroutine ref_atm_helmholtz(wrk, bat) bind(C, name='ref_atm_helmholtz')
    real(c_double), intent(inout) :: wrk(nxpa,nypa)
    real(c_double), intent(in) :: bat(nxta)
    call hscyat (wrk, bat)
  end subroutine ref_atm_helmholtz

end module qgcm_ref_atmos
#endif
