!-----------------------------------------------------------------------
! Stand-alone versions of the state/constant modules the shim USEs.
!
! In a drop-in build these four modules are the reference's own files
! (src/parameters_data.F, src/occonst_data.F, src/ochomog_data.F,
! src/ocstate_data.F) and this file is NOT compiled.  For the stand-alone
! Fortran host of this repository the same module and variable NAMES are
! declared here with run-time (allocatable) dimensions, restricted to what
! the ocean hot path reads or writes.
!-----------------------------------------------------------------------
module parameters
  implicit none
  public
  save
  integer :: nxto, nyto, nxpo, nypo, nlo
  double precision :: fnot, beta
end module parameters

module occonst
  implicit none
  public
  save
  double precision :: dxo, dyo, dxom2, xlo, ylo, delek, dto, tdto, bccooc, ycexp
  double precision, allocatable :: toc(:)
  double precision, allocatable :: yporel(:), gpoc(:), hoc(:), ah2oc(:), ah4oc(:), &
                                   amatoc(:,:), rdm2oc(:), ctl2moc(:,:), ctm2loc(:,:), ddynoc(:,:)
end module occonst

module ochomog
  implicit none
  public
  save
  double precision, allocatable :: ochom(:,:,:), aipohs(:), cdiffo(:,:), cdhoc(:,:)
  double precision, allocatable :: xon(:), dpioc(:), dpiocp(:)
  ! cyclic ocean (same names as src/ochomog_data.F; used when the shim is built -Dcyclic_ocean)
  double precision, allocatable :: pch1oc(:,:), pch2oc(:,:), pbhoc(:), aipcho(:), hc1soc(:), hc2soc(:), &
                                   hc1noc(:), hc2noc(:), ocncs(:), ocncn(:), ocncsp(:), ocncnp(:), &
                                   enisoc(:), eninoc(:)
  double precision :: hbsioc, aipbho, txisoc, txinoc
end module ochomog

module ocstate
  implicit none
  public
  save
  double precision, allocatable :: po(:,:,:), pom(:,:,:), qo(:,:,:), qom(:,:,:), entoc(:,:), wekpo(:,:), wekto(:,:)
end module ocstate

! what the mixed-layer shim (MODULE omlsubs) reads of src/intrfac_data.F and src/radiate_data.F
module intrfac
  implicit none
  public
  save
  double precision, allocatable :: sst(:,:), sstm(:,:), fnetoc(:,:), tauxo(:,:), tauyo(:,:)
  double precision :: hmoc, st2d, st4d, tsbdy, tnbdy
end module intrfac

module radiate
  implicit none
  public
  save
  double precision :: rrcpoc
end module radiate
