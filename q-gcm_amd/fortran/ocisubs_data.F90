!-----------------------------------------------------------------------
! Module data of the reference's MODULE ocisubs (src/ocisubs.F:50-55): the
! FFTPACK work array is kept only so that the main program's
! "call dsinti (nxto-1, oftwrk)" (src/q-gcm.F:954) still links; the HIP
! path builds its own transform tables.
!-----------------------------------------------------------------------
module ocisubs_data
  implicit none
  public
  save
  integer :: lwftoc = 0
  double precision, allocatable :: oftwrk(:), bd2oc(:)
  double precision :: aoc
end module ocisubs_data
