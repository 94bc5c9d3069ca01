!-----------------------------------------------------------------------
! qgcm_hip_iface - ISO_C_BINDING view of include/qgcm_hip.h
!
! One interface per C entry point; the derived type mirrors
! struct qgcm_hip_params field for field (QGCM_HIP_MAXL = 8).
!-----------------------------------------------------------------------
module qgcm_hip_iface
  use iso_c_binding
  implicit none
  public

  integer, parameter :: QGCM_HIP_MAXL = 8

  type, bind(C) :: qgcm_hip_params
    integer(c_int) :: nxpo, nypo, nlo, cyclic
    real(c_double) :: fnot, beta, dxo, dyo, tdto, delek, bccooc
    real(c_double) :: ah2oc(QGCM_HIP_MAXL), ah4oc(QGCM_HIP_MAXL)
    real(c_double) :: hoc(QGCM_HIP_MAXL), gpoc(QGCM_HIP_MAXL)
    real(c_double) :: amatoc(QGCM_HIP_MAXL*QGCM_HIP_MAXL)
    real(c_double) :: ctl2moc(QGCM_HIP_MAXL*QGCM_HIP_MAXL)
    real(c_double) :: ctm2loc(QGCM_HIP_MAXL*QGCM_HIP_MAXL)
    real(c_double) :: rdm2oc(QGCM_HIP_MAXL)
    real(c_double) :: aoc
    integer(c_int) :: slab_g0, slab_g1
    integer(c_int) :: atmos   ! 1: the handle is the atmospheric channel (qgastep / atinvq / atqzbd)
  end type qgcm_hip_params

  ! struct qgcm_hip_oml_params: run-time parameters of the ocean mixed layer
  type, bind(C) :: qgcm_hip_oml_params
    real(c_double) :: hmoc, toc1, toc2, st2d, st4d, ycexp, rrcpoc, tsbdy, tnbdy
    integer(c_int) :: sb_flag, nb_flag   ! sb_hflux, nb_hflux of the C struct (those names are cpp macros in reference builds)
  end type qgcm_hip_oml_params

  interface
    integer(c_int) function qgcm_hip_create(h, prm, device) bind(C, name='qgcm_hip_create')
      import :: c_ptr, c_int, qgcm_hip_params
      type(c_ptr), intent(out) :: h
      type(qgcm_hip_params), intent(in) :: prm
      integer(c_int), value :: device
    end function
    integer(c_int) function qgcm_hip_destroy(h) bind(C, name='qgcm_hip_destroy')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    type(c_ptr) function qgcm_hip_last_error() bind(C, name='qgcm_hip_last_error')
      import :: c_ptr
    end function
    integer(c_int) function qgcm_hip_set_grid(h, yporel, bd2oc, ddynoc) bind(C, name='qgcm_hip_set_grid')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: yporel(*), bd2oc(*), ddynoc(*)
    end function
    integer(c_int) function qgcm_hip_set_geometry(h, yporel, ddynoc) bind(C, name='qgcm_hip_set_geometry')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: yporel(*), ddynoc(*)
    end function
    integer(c_int) function qgcm_hip_set_homog_box(h, ochom, cdiffo, cdhoc) bind(C, name='qgcm_hip_set_homog_box')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: ochom(*), cdiffo(*), cdhoc(*)
    end function
    integer(c_int) function qgcm_hip_set_homog_cyc(h, pch1oc, pch2oc, pbhoc, aipcho, hc1soc, hc2soc, hc1noc, hc2noc, &
                                                   hbsioc, aipbho) bind(C, name='qgcm_hip_set_homog_cyc')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: pch1oc(*), pch2oc(*), pbhoc(*), aipcho(*), hc1soc(*), hc2soc(*), hc1noc(*), hc2noc(*)
      real(c_double), value :: hbsioc, aipbho
    end function
    integer(c_int) function qgcm_hip_abi_version() bind(C, name='qgcm_hip_abi_version')
      import :: c_int
    end function
    integer(c_int) function qgcm_hip_set_sponge(h, r_spl, c1_spl) bind(C, name='qgcm_hip_set_sponge')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: r_spl(*)
      real(c_double), value :: c1_spl
    end function
    integer(c_int) function qgcm_hip_set_cyc_forcing(h, txisoc, txinoc, enisoc, eninoc) bind(C, name='qgcm_hip_set_cyc_forcing')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), value :: txisoc, txinoc
      real(c_double), intent(in) :: enisoc(*), eninoc(*)
    end function
    integer(c_int) function qgcm_hip_set_state(h, po, pom, qo, qom) bind(C, name='qgcm_hip_set_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: po(*), pom(*), qo(*), qom(*)
    end function
    integer(c_int) function qgcm_hip_get_state(h, po, pom, qo, qom) bind(C, name='qgcm_hip_get_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: po(*), pom(*), qo(*), qom(*)
    end function
    integer(c_int) function qgcm_hip_set_forcing(h, wekpo, entoc, xon) bind(C, name='qgcm_hip_set_forcing')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: wekpo(*), entoc(*), xon(*)
    end function
    integer(c_int) function qgcm_hip_set_scalars(h, scal) bind(C, name='qgcm_hip_set_scalars')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: scal(*)
    end function
    integer(c_int) function qgcm_hip_get_scalars(h, scal) bind(C, name='qgcm_hip_get_scalars')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: scal(*)
    end function
    integer(c_int) function qgcm_hip_get_monitors(h, ermas, emfr) bind(C, name='qgcm_hip_get_monitors')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: ermas(*), emfr(*)
    end function
    integer(c_int) function qgcm_hip_qgostep(h) bind(C, name='qgcm_hip_qgostep')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_ocinvq(h) bind(C, name='qgcm_hip_ocinvq')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_ocqbdy(h) bind(C, name='qgcm_hip_ocqbdy')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    ! "call ocqbdy (q, p)" / "call atqzbd (q, p)" on host arrays (start-up calls, src/q-gcm.F:724-725, 743-744)
    integer(c_int) function qgcm_hip_ocqbdy_host(h, q, p) bind(C, name='qgcm_hip_ocqbdy_host')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(inout) :: q(*)
      real(c_double), intent(in) :: p(*)
    end function
    ! atmosphere path (handles created with prm%atmos = 1): src/q-gcm.F:1262-1268
    integer(c_int) function qgcm_hip_qgastep(h) bind(C, name='qgcm_hip_qgastep')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_atinvq(h) bind(C, name='qgcm_hip_atinvq')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_atqzbd(h) bind(C, name='qgcm_hip_atqzbd')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_get_bsums(h, b) bind(C, name='qgcm_hip_get_bsums')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: b(*)
    end function
    integer(c_int) function qgcm_hip_coupled_steps(oc, atm, nt0, n, nstr) bind(C, name='qgcm_hip_coupled_steps')
      import :: c_ptr, c_int
      type(c_ptr), value :: oc, atm
      integer(c_int), value :: nt0, n, nstr
    end function
    ! a handle's share of the GPU's compute units when two handles step side by side (qgcm_hip_coupled_steps)
    integer(c_int) function qgcm_hip_set_cu_range(h, first, count) bind(C, name='qgcm_hip_set_cu_range')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: first, count
    end function
    ! start-up / restart arithmetic and the progress sample on the device (src/q-gcm.F:711-731, 1933-2066;
    ! src/xfosubs.F:566-645)
    integer(c_int) function qgcm_hip_init_from_p(h) bind(C, name='qgcm_hip_init_from_p')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_wekpo_from_tau(h, tauxo, tauyo) bind(C, name='qgcm_hip_wekpo_from_tau')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: tauxo(*), tauyo(*)
    end function
    integer(c_int) function qgcm_hip_prsamp(h, out) bind(C, name='qgcm_hip_prsamp')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function qgcm_hip_lf_average(h) bind(C, name='qgcm_hip_lf_average')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_steps(h, s0, n) bind(C, name='qgcm_hip_steps')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: s0, n
    end function
    integer(c_int) function qgcm_hip_sync(h) bind(C, name='qgcm_hip_sync')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_helmholtz(h, wrk, boc) bind(C, name='qgcm_hip_helmholtz')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(inout) :: wrk(*)
      real(c_double), intent(in) :: boc(*)
    end function
    ! ocean mixed layer: "call oml" (src/q-gcm.F:1232) on the device
    integer(c_int) function qgcm_hip_oml_init(h, prm) bind(C, name='qgcm_hip_oml_init')
      import :: c_ptr, c_int, qgcm_hip_oml_params
      type(c_ptr), value :: h
      type(qgcm_hip_oml_params), intent(in) :: prm
    end function
    integer(c_int) function qgcm_hip_oml_set_state(h, sst, sstm) bind(C, name='qgcm_hip_oml_set_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: sst(*), sstm(*)
    end function
    integer(c_int) function qgcm_hip_oml_get_state(h, sst, sstm) bind(C, name='qgcm_hip_oml_get_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: sst(*), sstm(*)
    end function
    integer(c_int) function qgcm_hip_oml_set_forcing(h, fnetoc, wekto, tauxo, tauyo) bind(C, name='qgcm_hip_oml_set_forcing')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: fnetoc(*), wekto(*), tauxo(*), tauyo(*)
    end function
    integer(c_int) function qgcm_hip_oml(h) bind(C, name='qgcm_hip_oml')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_oml_get_diag(h, entoc, diag) bind(C, name='qgcm_hip_oml_get_diag')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: entoc(*), diag(5)
    end function
    ! validity scan: ocean part of "call valids (solnok)" (src/q-gcm.F:1278) on the device
    integer(c_int) function qgcm_hip_set_dtopoc(h, dtopoc) bind(C, name='qgcm_hip_set_dtopoc')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: dtopoc(*)
    end function
    integer(c_int) function qgcm_hip_valids(h, out, solnok) bind(C, name='qgcm_hip_valids')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: out(*)
      integer(c_int), intent(out) :: solnok
    end function
    ! y-slab runs, one process per GPU: rendezvous id (rank 0), communicator, whole distributed steps
    ! (the library issues the RCCL exchanges itself; include/qgcm_hip.h)
    integer(c_int) function qgcm_hip_comm_unique_id(id, nbytes) bind(C, name='qgcm_hip_comm_unique_id')
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: id(*)
      integer(c_int), value :: nbytes
    end function
    integer(c_int) function qgcm_hip_comm_init(h, id, nbytes, rank, nranks) bind(C, name='qgcm_hip_comm_init')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: id(*)
      integer(c_int), value :: nbytes, rank, nranks
    end function
    integer(c_int) function qgcm_hip_slab_steps(h, s0, n) bind(C, name='qgcm_hip_slab_steps')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: s0, n
    end function
    ! own exchanges: the right-hand-side independent part of the slab summaries, once after set_grid
    integer(c_int) function qgcm_hip_thomas_const_len(h) bind(C, name='qgcm_hip_thomas_const_len')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function qgcm_hip_thomas_consts(h, dst_dev) bind(C, name='qgcm_hip_thomas_consts')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, dst_dev
    end function
    integer(c_int) function qgcm_hip_set_thomas_consts(h, gath_dev, nranks) bind(C, name='qgcm_hip_set_thomas_consts')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, gath_dev
      integer(c_int), value :: nranks
    end function
    integer(c_int) function qgcm_hip_comm_probe(h, reps, us) bind(C, name='qgcm_hip_comm_probe')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: reps
      real(c_double), intent(out) :: us(3)
    end function
    integer(c_int) function qgcm_hip_comm_set_halo_p2p(h, on) bind(C, name='qgcm_hip_comm_set_halo_p2p')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function
    integer(c_int) function qgcm_hip_comm_set_overlap(h, on) bind(C, name='qgcm_hip_comm_set_overlap')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function
  end interface

contains

  ! The reference's error convention is print + stop (e.g. src/ocisubs.F:361-365).
  ! QGCM_HIP_ABI_VERSION of include/qgcm_hip.h this interface block was written against: an older libqgcm_hip.so
  ! (without qgcm_hip_get_monitors / qgcm_hip_set_sponge, or with the other halo default) must not be driven by it
  subroutine qgcm_hip_check_abi
    integer(c_int), parameter :: want = 3
    if (qgcm_hip_abi_version() /= want) then
      print *, ' qgcm_hip: libqgcm_hip.so has ABI version ', qgcm_hip_abi_version(), ', this host binds version ', want
      print *, ' program terminates'
      stop 1
    endif
  end subroutine qgcm_hip_check_abi

  subroutine qgcm_hip_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    integer :: i
    if (rc == 0) return
    call c_f_pointer(qgcm_hip_last_error(), msg, [512])
    print *, ' qgcm_hip error in ', where, ':'
    do i = 1, 512
      if (msg(i) == c_null_char) exit
      write(*, '(a)', advance='no') msg(i)
    enddo
    print *
    print *, ' program terminates'
    stop 1
  end subroutine qgcm_hip_check

end module qgcm_hip_iface
