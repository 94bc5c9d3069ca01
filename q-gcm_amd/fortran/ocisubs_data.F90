!-----------------------------------------------------------------------
! Module data of the reference's MODULE ocisubs (src/ocisubs.F:50-55) and
! MODULE atisubs (src/atisubs.F:44-49).  The FFTPACK work arrays are kept
! only because the main program still initialises them
! ("call dsinti (nxto-1, oftwrk)" / "call drffti (nxta, aftwrk)",
! src/q-gcm.F:945-972); the HIP path builds its own transform tables.
!
!  -DQGCM_DROPIN  drop-in build under the reference main program: sized
!                 from MODULE parameters exactly as the reference declares
!                 them, so that src/q-gcm.F:932-972 writes bd2oc / bd2at
!                 and oftwrk / aftwrk as before.
!  otherwise      the stand-alone host of this repository (run-time grid):
!                 allocatables, allocated by qgcm_ocean_host.F90.
!-----------------------------------------------------------------------
module ocisubs_data
#ifdef QGCM_DROPIN
  use parameters, only : nxto
#endif
  implicit none
  public
  save
#ifdef QGCM_DROPIN
  integer, parameter :: lwftoc = 3*nxto + 15   ! src/ocisubs.F:51
  double precision :: oftwrk(lwftoc)
  double precision :: aoc, bd2oc(nxto)
#else
  integer :: lwftoc = 0
  double precision, allocatable :: oftwrk(:), bd2oc(:)
  double precision :: aoc
#endif
end module ocisubs_data

#if defined(QGCM_DROPIN) && !defined(ocean_only)
module atisubs_data
  use parameters, only : nxta
  implicit none
  public
  save
  integer, parameter :: lwftat = 2*nxta + 15   ! src/atisubs.F:44
  double precision :: aftwrk(lwftat)
  double precision :: aat, bd2at(nxta)
end module atisubs_data
#endif
