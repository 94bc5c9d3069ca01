!-----------------------------------------------------------------------
! Drop-in module procedures for the ocean hot path.
!
! The reference main program calls, once per ocean step and without
! arguments (src/q-gcm.F:1243-1249):
!
!       call qgostep          (MODULE qgosubs, src/qgosubs.F:45)
!       call ocinvq           (MODULE ocisubs, src/ocisubs.F:64)
!       call ocqbdy (qo, po)  (MODULE vorsubs, src/vorsubs.F:245)
!
! and homsol calls hsbxoc (wrk, boc) at start-up (src/conhoms.F:572).
! The modules below keep those names, argument lists and PUBLIC/PRIVATE
! split, so q-gcm.F compiles against them unchanged; the bodies forward to
! the C ABI of include/qgcm_hip.h.  Data still lives in the reference's own
! state modules (parameters, occonst, ochomog, ocstate), which are USEd,
! not replaced: the device copy is created from them on the first call
! and refreshed/pulled with qgcm_hip_push / qgcm_hip_pull.
!
! Build with the same cpp macros as the reference (-Docean_only ...).
!-----------------------------------------------------------------------
! -Dsponge_layer_k247 (the fork's sponge term of the leapfrog step, src/qgosubs.F:70-72,175-205): the ramp r_spl
! and c1_spl go to the device at the first "call qgostep" - the main program sets r_spl late in its set-up
! (src/q-gcm.F:1154-1168), after the point where the state is first pushed (after homsol, :976).
module qgcm_hip_state
  use iso_c_binding
  use qgcm_hip_iface
  implicit none
  private
  public :: qgcm_hip_handle, qgcm_hip_ensure, qgcm_hip_ensure_geometry, qgcm_hip_push, qgcm_hip_pull, &
            qgcm_hip_push_forcing, qgcm_hip_shutdown, qgcm_hip_homog_ready, qgcm_hip_device_owns
#ifdef sponge_layer_k247
  public :: push_sponge
#endif

  type(c_ptr), save :: qgcm_hip_handle = c_null_ptr
  logical, save :: homog_sent = .false., grid_sent = .false.
  ! .false. until the first qgcm_hip_push: the host module arrays are still the authoritative state
  ! (start-up: constr, qcomp, ocqbdy on po/pom act on host arrays, src/q-gcm.F:711-731)
  logical, save :: qgcm_hip_device_owns = .false.

contains

  ! Create the device context from MODULE parameters / occonst / ocisubs data and send the
  ! tridiagonal coefficients.  Safe to call repeatedly; homsol's first hsbxoc call triggers it
  ! (by then the main program has computed aoc, bd2oc: src/q-gcm.F:932-954 precede homsol, :976).
  subroutine qgcm_hip_ensure
    use occonst, only : yporel, ddynoc
    use ocisubs_data, only : bd2oc
    call qgcm_hip_ensure_geometry
    if (grid_sent) return
    call qgcm_hip_check(qgcm_hip_set_grid(qgcm_hip_handle, yporel, bd2oc, ddynoc), 'qgcm_hip_set_grid')
    grid_sent = .true.
  end subroutine qgcm_hip_ensure

  ! The part that is possible before bd2oc exists: the start-up "call ocqbdy (qo, po)" (src/q-gcm.F:724) comes after
  ! eigmod and topset but before the tridiagonal coefficients.  aoc = 1/dyo**2 as src/q-gcm.F:932 will set it.
  subroutine qgcm_hip_ensure_geometry
    use parameters
    use occonst
    type(qgcm_hip_params) :: p
    integer :: k, l
    if (c_associated(qgcm_hip_handle)) return
    call qgcm_hip_check_abi
    p%nxpo = nxpo; p%nypo = nypo; p%nlo = nlo
#ifdef cyclic_ocean
    p%cyclic = 1
#else
    p%cyclic = 0
#endif
    p%fnot = fnot; p%beta = beta; p%dxo = dxo; p%dyo = dyo
    p%tdto = tdto; p%delek = delek; p%bccooc = bccooc; p%aoc = 1.0d0/( dyo*dyo )
    p%slab_g0 = 0; p%slab_g1 = 0
    p%atmos = 0
    p%ah2oc = 0.0d0; p%ah4oc = 0.0d0; p%hoc = 0.0d0; p%gpoc = 0.0d0
    p%amatoc = 0.0d0; p%ctl2moc = 0.0d0; p%ctm2loc = 0.0d0; p%rdm2oc = 0.0d0
    do k = 1, nlo
      p%ah2oc(k) = ah2oc(k); p%ah4oc(k) = ah4oc(k); p%hoc(k) = hoc(k)
      p%rdm2oc(k) = rdm2oc(k)
      do l = 1, nlo
        p%amatoc(k + nlo*(l-1)) = amatoc(k,l)
        p%ctl2moc(k + nlo*(l-1)) = ctl2moc(k,l)
        p%ctm2loc(k + nlo*(l-1)) = ctm2loc(k,l)
      enddo
    enddo
    do k = 1, nlo-1
      p%gpoc(k) = gpoc(k)
    enddo
    call qgcm_hip_check(qgcm_hip_create(qgcm_hip_handle, p, -1_c_int), 'qgcm_hip_create')
    call qgcm_hip_check(qgcm_hip_set_geometry(qgcm_hip_handle, yporel, ddynoc), 'qgcm_hip_set_geometry')
  end subroutine qgcm_hip_ensure_geometry

  ! Send the homsol products once they exist (after "call homsol", src/q-gcm.F:976).
  subroutine qgcm_hip_homog_ready
    use ochomog
    if (homog_sent) return
    call qgcm_hip_ensure
#ifdef cyclic_ocean
    call qgcm_hip_check(qgcm_hip_set_homog_cyc(qgcm_hip_handle, pch1oc, pch2oc, pbhoc, aipcho, hc1soc, hc2soc, &
                                               hc1noc, hc2noc, hbsioc, aipbho), 'qgcm_hip_set_homog_cyc')
#else
    call qgcm_hip_check(qgcm_hip_set_homog_box(qgcm_hip_handle, ochom, cdiffo, cdhoc), 'qgcm_hip_set_homog_box')
#endif
    homog_sent = .true.
  end subroutine qgcm_hip_homog_ready

  ! host module arrays -> device (after initialisation / restart read / oml)
  subroutine qgcm_hip_push
    use parameters, only : nlo
    use ocstate
    use ochomog
    real(c_double) :: scal(2*(nlo-1) + 4*nlo)
    call qgcm_hip_homog_ready
    call qgcm_hip_check(qgcm_hip_set_state(qgcm_hip_handle, po, pom, qo, qom), 'qgcm_hip_set_state')
    scal = 0.0d0
    scal(1:nlo-1) = dpioc
    scal(nlo:2*(nlo-1)) = dpiocp
#ifdef cyclic_ocean
    scal(2*(nlo-1)+1:2*(nlo-1)+nlo) = ocncs
    scal(2*(nlo-1)+nlo+1:2*(nlo-1)+2*nlo) = ocncn
    scal(2*(nlo-1)+2*nlo+1:2*(nlo-1)+3*nlo) = ocncsp
    scal(2*(nlo-1)+3*nlo+1:2*(nlo-1)+4*nlo) = ocncnp
#endif
    call qgcm_hip_check(qgcm_hip_set_scalars(qgcm_hip_handle, scal), 'qgcm_hip_set_scalars')
    call qgcm_hip_push_forcing
    qgcm_hip_device_owns = .true.
  end subroutine qgcm_hip_push

  subroutine qgcm_hip_push_forcing
    use ocstate, only : wekpo, entoc
    use ochomog, only : xon
    call qgcm_hip_check(qgcm_hip_set_forcing(qgcm_hip_handle, wekpo, entoc, xon), 'qgcm_hip_set_forcing')
#ifdef cyclic_ocean
    call push_cyc
#endif
  end subroutine qgcm_hip_push_forcing

#ifdef sponge_layer_k247
  ! r_spl (MODULE occonst) is set late in the main program's set-up (src/q-gcm.F:1154-1168), before the time loop;
  ! sent once, by the first qgostep - it does not change afterwards
  subroutine push_sponge
    use parameters, only : c1_spl
    use occonst, only : r_spl
    logical, save :: sent = .false.
    if (sent) return
    call qgcm_hip_check(qgcm_hip_set_sponge(qgcm_hip_handle, r_spl, c1_spl), 'qgcm_hip_set_sponge')
    sent = .true.
  end subroutine push_sponge
#endif

#ifdef cyclic_ocean
  ! line integrals that xforc (txisoc/txinoc) and oml (enisoc/eninoc) maintain, src/ochomog_data.F
  subroutine push_cyc
    use ochomog, only : txisoc, txinoc, enisoc, eninoc
    call qgcm_hip_check(qgcm_hip_set_cyc_forcing(qgcm_hip_handle, txisoc, txinoc, enisoc, eninoc), &
                        'qgcm_hip_set_cyc_forcing')
  end subroutine push_cyc
#endif

  ! device -> host module arrays (before valids, prsamp, monnc_comp, resave, ocnc_out ...)
  subroutine qgcm_hip_pull
    use parameters, only : nlo
    use ocstate
    use ochomog
    real(c_double) :: scal(2*(nlo-1) + 4*nlo)
    call qgcm_hip_check(qgcm_hip_get_state(qgcm_hip_handle, po, pom, qo, qom), 'qgcm_hip_get_state')
    call qgcm_hip_check(qgcm_hip_get_scalars(qgcm_hip_handle, scal), 'qgcm_hip_get_scalars')
    dpioc = scal(1:nlo-1)
    dpiocp = scal(nlo:2*(nlo-1))
#ifdef cyclic_ocean
    ocncs = scal(2*(nlo-1)+1:2*(nlo-1)+nlo)
    ocncn = scal(2*(nlo-1)+nlo+1:2*(nlo-1)+2*nlo)
    ocncsp = scal(2*(nlo-1)+2*nlo+1:2*(nlo-1)+3*nlo)
    ocncnp = scal(2*(nlo-1)+3*nlo+1:2*(nlo-1)+4*nlo)
    ! continuity monitors of ocinvq (src/ocisubs.F:268-283) for monnc_comp / monit.nc
    call pull_monitors
#endif
  end subroutine qgcm_hip_pull

#ifdef cyclic_ocean
  subroutine pull_monitors
    use monitor, only : ermaso, emfroc
    call qgcm_hip_check(qgcm_hip_get_monitors(qgcm_hip_handle, ermaso, emfroc), 'qgcm_hip_get_monitors')
  end subroutine pull_monitors
#endif

  subroutine qgcm_hip_shutdown
    if (c_associated(qgcm_hip_handle)) then
      call qgcm_hip_check(qgcm_hip_destroy(qgcm_hip_handle), 'qgcm_hip_destroy')
      qgcm_hip_handle = c_null_ptr
    endif
  end subroutine qgcm_hip_shutdown

end module qgcm_hip_state

!-----------------------------------------------------------------------
module qgosubs
  ! same public surface as src/qgosubs.F:23-39 (qgostep PUBLIC; ocadif is gone:
  ! it is fused into the tendency kernel)
  implicit none
  private
  public :: qgostep
contains
  subroutine qgostep
    use qgcm_hip_iface
    use qgcm_hip_state
#ifdef sponge_layer_k247
    call push_sponge
#endif
    call qgcm_hip_check(qgcm_hip_qgostep(qgcm_hip_handle), 'qgostep')
  end subroutine qgostep
end module qgosubs

!-----------------------------------------------------------------------
module ocisubs
  ! same public surface as src/ocisubs.F:23-55: ocinvq, hsbxoc (box) and the
  ! module data lwftoc/oftwrk/aoc/bd2oc, re-exported from ocisubs_data so that
  ! the reference main program's "USE ocisubs" keeps working.
  use ocisubs_data
  implicit none
  private
  public :: ocinvq, lwftoc, oftwrk, aoc, bd2oc
#ifdef cyclic_ocean
  public :: hscyoc
#else
  public :: hsbxoc
#endif
contains
  subroutine ocinvq
    use qgcm_hip_iface
    use qgcm_hip_state
    call qgcm_hip_check(qgcm_hip_ocinvq(qgcm_hip_handle), 'ocinvq')
  end subroutine ocinvq

#ifdef cyclic_ocean
  subroutine hscyoc (wrk, boc)
    use parameters, only : nxpo, nypo, nxto
    use qgcm_hip_iface
    use qgcm_hip_state
    double precision, intent(inout) :: wrk(nxpo,nypo)
    double precision, intent(in) :: boc(nxto)
    call qgcm_hip_ensure
    call qgcm_hip_check(qgcm_hip_helmholtz(qgcm_hip_handle, wrk, boc), 'hscyoc')
  end subroutine hscyoc
#else
  subroutine hsbxoc (wrk, boc)
    use parameters, only : nxpo, nypo, nxto
    use qgcm_hip_iface
    use qgcm_hip_state
    double precision, intent(inout) :: wrk(nxpo,nypo)
    double precision, intent(in) :: boc(nxto)
    call qgcm_hip_ensure
    call qgcm_hip_check(qgcm_hip_helmholtz(qgcm_hip_handle, wrk, boc), 'hsbxoc')
  end subroutine hsbxoc
#endif
end module ocisubs

!-----------------------------------------------------------------------
module vorsubs_hip
  ! ocqbdy of src/vorsubs.F:245-388.  qcomp / merqcy (init only) stay in the
  ! reference's vorsubs ("USE vorsubs, ONLY : qcomp, merqcy"); this module's ocqbdy
  ! serves both kinds of call the main program makes:
  !   per step, "call ocqbdy (qo, po)" (src/q-gcm.F:1249): the arguments are the module
  !     arrays themselves and the device owns the data - the kernel updates the device copy;
  !   at start-up, "call ocqbdy (qo, po)" / "(qom, pom)" (src/q-gcm.F:724-725), before the
  !     first qgcm_hip_push: the host arrays are authoritative - p is staged to the device,
  !     the same kernel runs and the boundary ring of q comes back (qgcm_hip_ocqbdy_host).
  implicit none
  private
  public :: ocqbdy
contains
  subroutine ocqbdy (qo, po)
    use parameters, only : nxpo, nypo, nlo
    use qgcm_hip_iface
    use qgcm_hip_state
    double precision :: qo(nxpo,nypo,nlo), po(nxpo,nypo,nlo)
    if (qgcm_hip_device_owns) then
      call qgcm_hip_check(qgcm_hip_ocqbdy(qgcm_hip_handle), 'ocqbdy')
    else
      call qgcm_hip_ensure_geometry
      call qgcm_hip_check(qgcm_hip_ocqbdy_host(qgcm_hip_handle, qo, po), 'ocqbdy (host arrays)')
    endif
  end subroutine ocqbdy
end module vorsubs_hip


!-----------------------------------------------------------------------
! MODULE omlsubs: drop-in for `call oml` (src/q-gcm.F:1232, body src/omlsubs.F:47-236 + omladf 244-763;
! SURVEY 8 row f1).  The mixed layer runs on the device next to the PV path: entoc, xon(1) and (cyclic)
! enisoc(1)/eninoc(1) never leave the GPU, so the per-step PCIe traffic of a host-side oml
! (po(:,:,1) down, entoc up: 2 x 7.4 MB at 5 km) disappears.
!   call qgcm_hip_oml_push        once after the initial sst / forcing are set (and after a restart read;
!                                 again whenever xforc changes fnetoc / wekto / tauxo / tauyo)
!   call oml                      every ocean step, before qgostep (unchanged call site)
!   call qgcm_hip_oml_pull        before anything on the host reads sst / sstm / entoc / xon
! Build with the reference's cpp macros (sb_hflux / nb_hflux select the boundary variants there; here they
! become the run-time flags of qgcm_hip_oml_params).
!-----------------------------------------------------------------------
module omlsubs
  use iso_c_binding
  use qgcm_hip_iface
  use qgcm_hip_state
  implicit none
  private
  public :: oml, qgcm_hip_oml_push, qgcm_hip_oml_pull
  logical, save :: oml_ready = .false.
contains

  subroutine qgcm_hip_oml_push
    use occonst, only : toc, ycexp
    use ocstate, only : wekto
    use intrfac, only : sst, sstm, fnetoc, tauxo, tauyo, hmoc, st2d, st4d, tsbdy, tnbdy
    use radiate, only : rrcpoc
    type(qgcm_hip_oml_params) :: p
    call qgcm_hip_ensure
    if (.not. oml_ready) then
      p%hmoc = hmoc; p%toc1 = toc(1); p%toc2 = toc(2); p%st2d = st2d; p%st4d = st4d
      p%ycexp = ycexp; p%rrcpoc = rrcpoc; p%tsbdy = tsbdy; p%tnbdy = tnbdy
      p%sb_flag = 0; p%nb_flag = 0
#ifdef sb_hflux
      p%sb_flag = 1
#endif
#ifdef nb_hflux
      p%nb_flag = 1
#endif
      call qgcm_hip_check(qgcm_hip_oml_init(qgcm_hip_handle, p), 'qgcm_hip_oml_init')
      oml_ready = .true.
    endif
    call qgcm_hip_check(qgcm_hip_oml_set_state(qgcm_hip_handle, sst, sstm), 'qgcm_hip_oml_set_state')
    call qgcm_hip_check(qgcm_hip_oml_set_forcing(qgcm_hip_handle, fnetoc, wekto, tauxo, tauyo), 'qgcm_hip_oml_set_forcing')
  end subroutine qgcm_hip_oml_push

  subroutine oml
    if (.not. oml_ready) call qgcm_hip_oml_push
    call qgcm_hip_check(qgcm_hip_oml(qgcm_hip_handle), 'qgcm_hip_oml')
  end subroutine oml

  subroutine qgcm_hip_oml_pull
    use ocstate, only : entoc
    use ochomog
    use intrfac, only : sst, sstm
    real(c_double) :: diag(5)
    call qgcm_hip_check(qgcm_hip_oml_get_state(qgcm_hip_handle, sst, sstm), 'qgcm_hip_oml_get_state')
    call qgcm_hip_check(qgcm_hip_oml_get_diag(qgcm_hip_handle, entoc, diag), 'qgcm_hip_oml_get_diag')
    xon(1) = diag(1)
#ifdef cyclic_ocean
    enisoc(1) = diag(4)
    eninoc(1) = diag(5)
#endif
  end subroutine qgcm_hip_oml_pull

end module omlsubs


!-----------------------------------------------------------------------
! MODULE valsubs_hip: `valids_hip (solnok)` for the call site src/q-gcm.F:1278 (SURVEY 8 row f2).
! The scan of src/valsubs.F:272-527 runs on the device (20 doubles back instead of po, qo);
! only after a negative verdict is the state pulled so that the reference's own valids can
! print its neighbourhood diagnostics:
!       call valids_hip (solnok)
!       if (.not.solnok) then
!         call qgcm_hip_pull ; call valids (solnok)      ! reference routine, unchanged
!       endif
!-----------------------------------------------------------------------
module valsubs_hip
  use iso_c_binding
  use qgcm_hip_iface
  use qgcm_hip_state
  implicit none
  private
  public :: valids_hip
contains
  subroutine valids_hip (solnok, extremes)
    use parameters, only : nlo
    logical, intent(inout) :: solnok
    double precision, intent(out), optional :: extremes(14 + nlo)
    real(c_double) :: res(14 + nlo)
    integer(c_int) :: ok
    call qgcm_hip_check(qgcm_hip_valids(qgcm_hip_handle, res, ok), 'qgcm_hip_valids')
    if (ok == 0) solnok = .false.
    if (present(extremes)) extremes = res
  end subroutine valids_hip
end module valsubs_hip


#if defined(QGCM_DROPIN) && !defined(ocean_only)
!-----------------------------------------------------------------------
! Atmosphere (SURVEY 8 row f3; coupled and atmos_only builds).  The main program calls, every
! atmospheric step and without arguments (src/q-gcm.F:1262-1268):
!
!       call qgastep           (MODULE qgasubs, src/qgasubs.F:45)
!       call atinvq            (MODULE atisubs, src/atisubs.F:60)
!       call atqzbd (qa, pa)   (MODULE vorsubs, src/vorsubs.F:396)
!
! and homsol calls hscyat (wrk, bat) at start-up (src/conhoms.F:678-679).  Same scheme as the ocean: the
! reference's data modules (parameters, atconst, athomog, atstate) are USEd, a second device handle
! (qgcm_hip_params%atmos = 1) is created from them on first use.
!   call qgcm_hip_atm_push          after the initial constr / qcomp / atqzbd / merqcy and homsol (src/q-gcm.F:976)
!   call qgcm_hip_atm_push_forcing  after xforc / aml changed wekpa, entat, xan, txisat.., enisat..
!   call qgcm_hip_atm_pull          before anything on the host reads pa, pam, qa, qam, dpiat, atmcs..
!-----------------------------------------------------------------------
module qgcm_hip_atstate
  use iso_c_binding
  use qgcm_hip_iface
  implicit none
  private
  public :: qgcm_hip_atm_handle, qgcm_hip_atm_ensure, qgcm_hip_atm_ensure_geometry, qgcm_hip_atm_push, qgcm_hip_atm_pull, &
            qgcm_hip_atm_push_forcing, qgcm_hip_atm_shutdown, qgcm_hip_atm_device_owns

  type(c_ptr), save :: qgcm_hip_atm_handle = c_null_ptr
  logical, save :: homog_sent = .false., grid_sent = .false.
  logical, save :: qgcm_hip_atm_device_owns = .false.

contains

  subroutine qgcm_hip_atm_ensure
    use atconst, only : yparel, ddynat
    use atisubs_data, only : bd2at
    call qgcm_hip_atm_ensure_geometry
    if (grid_sent) return
    call qgcm_hip_check(qgcm_hip_set_grid(qgcm_hip_atm_handle, yparel, bd2at, ddynat), 'qgcm_hip_set_grid (atmosphere)')
    grid_sent = .true.
  end subroutine qgcm_hip_atm_ensure

  ! before bd2at exists (start-up atqzbd, src/q-gcm.F:743, precedes :961-972); aat = 1/dya**2 as :961 will set it
  subroutine qgcm_hip_atm_ensure_geometry
    use parameters
    use atconst
    type(qgcm_hip_params) :: p
    integer :: k, l
    if (c_associated(qgcm_hip_atm_handle)) return
    p%nxpo = nxpa; p%nypo = nypa; p%nlo = nla
    p%cyclic = 1
    p%atmos = 1
    p%fnot = fnot; p%beta = beta; p%dxo = dxa; p%dyo = dya
    p%tdto = tdta; p%delek = 0.0d0; p%bccooc = bccoat; p%aoc = 1.0d0/( dya*dya )
    p%slab_g0 = 0; p%slab_g1 = 0
    p%ah2oc = 0.0d0; p%ah4oc = 0.0d0; p%hoc = 0.0d0; p%gpoc = 0.0d0
    p%amatoc = 0.0d0; p%ctl2moc = 0.0d0; p%ctm2loc = 0.0d0; p%rdm2oc = 0.0d0
    do k = 1, nla
      p%ah4oc(k) = ah4at(k); p%hoc(k) = hat(k)
      p%rdm2oc(k) = rdm2at(k)
      do l = 1, nla
        p%amatoc(k + nla*(l-1)) = amatat(k,l)
        p%ctl2moc(k + nla*(l-1)) = ctl2mat(k,l)
        p%ctm2loc(k + nla*(l-1)) = ctm2lat(k,l)
      enddo
    enddo
    do k = 1, nla-1
      p%gpoc(k) = gpat(k)
    enddo
    call qgcm_hip_check(qgcm_hip_create(qgcm_hip_atm_handle, p, -1_c_int), 'qgcm_hip_create (atmosphere)')
    call qgcm_hip_check(qgcm_hip_set_geometry(qgcm_hip_atm_handle, yparel, ddynat), 'qgcm_hip_set_geometry (atmosphere)')
  end subroutine qgcm_hip_atm_ensure_geometry

  subroutine qgcm_hip_atm_push
    use parameters, only : nla
    use atstate
    use athomog
    real(c_double) :: scal(2*(nla-1) + 4*nla)
    call qgcm_hip_atm_ensure
    if (.not. homog_sent) then
      call qgcm_hip_check(qgcm_hip_set_homog_cyc(qgcm_hip_atm_handle, pch1at, pch2at, pbhat, aipcha, hc1sat, hc2sat, &
                                                 hc1nat, hc2nat, hbsiat, aipbha), 'qgcm_hip_set_homog_cyc (atmosphere)')
      homog_sent = .true.
    endif
    call qgcm_hip_check(qgcm_hip_set_state(qgcm_hip_atm_handle, pa, pam, qa, qam), 'qgcm_hip_set_state (atmosphere)')
    scal(1:nla-1) = dpiat
    scal(nla:2*(nla-1)) = dpiatp
    scal(2*(nla-1)+1:2*(nla-1)+nla) = atmcs
    scal(2*(nla-1)+nla+1:2*(nla-1)+2*nla) = atmcn
    scal(2*(nla-1)+2*nla+1:2*(nla-1)+3*nla) = atmcsp
    scal(2*(nla-1)+3*nla+1:2*(nla-1)+4*nla) = atmcnp
    call qgcm_hip_check(qgcm_hip_set_scalars(qgcm_hip_atm_handle, scal), 'qgcm_hip_set_scalars (atmosphere)')
    call qgcm_hip_atm_push_forcing
    qgcm_hip_atm_device_owns = .true.
  end subroutine qgcm_hip_atm_push

  ! what xforc / aml leave in MODULE atstate / athomog for the path
  subroutine qgcm_hip_atm_push_forcing
    use atstate, only : wekpa, entat
    use athomog, only : xan, txisat, txinat, enisat, eninat
    call qgcm_hip_check(qgcm_hip_set_forcing(qgcm_hip_atm_handle, wekpa, entat, xan), 'qgcm_hip_set_forcing (atmosphere)')
    call qgcm_hip_check(qgcm_hip_set_cyc_forcing(qgcm_hip_atm_handle, txisat, txinat, enisat, eninat), &
                        'qgcm_hip_set_cyc_forcing (atmosphere)')
  end subroutine qgcm_hip_atm_push_forcing

  subroutine qgcm_hip_atm_pull
    use parameters, only : nla
    use atstate
    use athomog
    real(c_double) :: scal(2*(nla-1) + 4*nla), b(4*nla)
    call qgcm_hip_check(qgcm_hip_get_state(qgcm_hip_atm_handle, pa, pam, qa, qam), 'qgcm_hip_get_state (atmosphere)')
    call qgcm_hip_check(qgcm_hip_get_scalars(qgcm_hip_atm_handle, scal), 'qgcm_hip_get_scalars (atmosphere)')
    dpiat = scal(1:nla-1)
    dpiatp = scal(nla:2*(nla-1))
    atmcs = scal(2*(nla-1)+1:2*(nla-1)+nla)
    atmcn = scal(2*(nla-1)+nla+1:2*(nla-1)+2*nla)
    atmcsp = scal(2*(nla-1)+2*nla+1:2*(nla-1)+3*nla)
    atmcnp = scal(2*(nla-1)+3*nla+1:2*(nla-1)+4*nla)
    call qgcm_hip_check(qgcm_hip_get_bsums(qgcm_hip_atm_handle, b), 'qgcm_hip_get_bsums (atmosphere)')
    ajisat = b(1:nla); ajinat = b(nla+1:2*nla); ap5sat = b(2*nla+1:3*nla); ap5nat = b(3*nla+1:4*nla)
    call atm_pull_monitors   ! ermasa, emfrat of atinvq (src/atisubs.F:236-248)
  end subroutine qgcm_hip_atm_pull

  subroutine atm_pull_monitors
    use monitor, only : ermasa, emfrat
    call qgcm_hip_check(qgcm_hip_get_monitors(qgcm_hip_atm_handle, ermasa, emfrat), 'qgcm_hip_get_monitors (atmosphere)')
  end subroutine atm_pull_monitors

  subroutine qgcm_hip_atm_shutdown
    if (c_associated(qgcm_hip_atm_handle)) then
      call qgcm_hip_check(qgcm_hip_destroy(qgcm_hip_atm_handle), 'qgcm_hip_destroy (atmosphere)')
      qgcm_hip_atm_handle = c_null_ptr
    endif
  end subroutine qgcm_hip_atm_shutdown

end module qgcm_hip_atstate

!-----------------------------------------------------------------------
module qgasubs
  ! same public surface as src/qgasubs.F:23-39 (qgastep PUBLIC; atadif is fused into the tendency kernel)
  implicit none
  private
  public :: qgastep
contains
  subroutine qgastep
    use qgcm_hip_iface
    use qgcm_hip_atstate
    call qgcm_hip_check(qgcm_hip_qgastep(qgcm_hip_atm_handle), 'qgastep')
  end subroutine qgastep
end module qgasubs

!-----------------------------------------------------------------------
module atisubs
  ! same public surface as src/atisubs.F:23-49: atinvq, hscyat and the module data lwftat / aftwrk / aat / bd2at
  use atisubs_data
  implicit none
  private
  public :: atinvq, hscyat, lwftat, aftwrk, aat, bd2at
contains
  subroutine atinvq
    use qgcm_hip_iface
    use qgcm_hip_atstate
    call qgcm_hip_check(qgcm_hip_atinvq(qgcm_hip_atm_handle), 'atinvq')
  end subroutine atinvq

  subroutine hscyat (wrk, bat)
    use parameters, only : nxpa, nypa, nxta
    use qgcm_hip_iface
    use qgcm_hip_atstate
    double precision, intent(inout) :: wrk(nxpa,nypa)
    double precision, intent(in) :: bat(nxta)
    call qgcm_hip_atm_ensure
    call qgcm_hip_check(qgcm_hip_helmholtz(qgcm_hip_atm_handle, wrk, bat), 'hscyat')
  end subroutine hscyat
end module atisubs

!-----------------------------------------------------------------------
module vorsubs_hip_at
  ! atqzbd of src/vorsubs.F:396-480, for the per-step call "call atqzbd (qa, pa)" (src/q-gcm.F:1268) and - before
  ! the first qgcm_hip_atm_push - for the start-up calls on (qa, pa), (qam, pam) (src/q-gcm.F:743-744), which are
  ! staged through the device (qgcm_hip_ocqbdy_host runs the same kernel on host arrays).
  implicit none
  private
  public :: atqzbd
contains
  subroutine atqzbd (qa, pa)
    use parameters, only : nxpa, nypa, nla
    use qgcm_hip_iface
    use qgcm_hip_atstate
    double precision :: qa(nxpa,nypa,nla), pa(nxpa,nypa,nla)
    if (qgcm_hip_atm_device_owns) then
      call qgcm_hip_check(qgcm_hip_atqzbd(qgcm_hip_atm_handle), 'atqzbd')
    else
      call qgcm_hip_atm_ensure_geometry
      call qgcm_hip_check(qgcm_hip_ocqbdy_host(qgcm_hip_atm_handle, qa, pa), 'atqzbd (host arrays)')
    endif
  end subroutine atqzbd
end module vorsubs_hip_at
#endif
