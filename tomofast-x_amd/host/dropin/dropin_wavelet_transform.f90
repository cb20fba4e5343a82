!=========================================================================================================
! DROP-IN module `wavelet_transform` for the UNMODIFIED Tomofast-x sources (build recipe: INTEGRATION.md 0).
! Replaces src/utils/wavelet_transform.F90: same public names and argument lists (:23-30, :37-39, :56-58, :75-77, :158-160,
! :243-245, :374-376); every transform is tfx_wavelet of libtfx.so (HIP kernel k_wavelet_axis), in place, bit-identical to the
! reference's lifting scheme (tests/test_gpu_parity.py::test_wavelets_bit_exact_*).  The repository's own code.
!=========================================================================================================
module wavelet_transform
  use iso_c_binding
  use global_typedefs
  use tfx_binding
  use tfx_reference_api, only: tfx_api_context, api_check
  implicit none
  private

  public :: forward_wavelet
  public :: inverse_wavelet
  public :: Haar3D
  public :: iHaar3D
  public :: DaubD43D
  public :: iDaubD43D

contains

  subroutine on_device(s, n1, n2, n3, wavelet_type, direction, where)
    integer, intent(in) :: n1, n2, n3, wavelet_type, direction
    real(kind=CUSTOM_REAL), intent(inout) :: s(n1, n2, n3)
    character(len=*), intent(in) :: where
    call api_check(tfx_wavelet(tfx_api_context(0, 1), s, n1, n2, n3, 1_c_int64_t, int(wavelet_type, c_int), int(direction, c_int)), where, 0)
  end subroutine on_device

  subroutine forward_wavelet(s, n1, n2, n3, wavelet_type)
    integer, intent(in) :: n1, n2, n3, wavelet_type
    real(kind=CUSTOM_REAL), intent(inout) :: s(n1, n2, n3)
    if (wavelet_type /= 1 .and. wavelet_type /= 2) then
      print *, "Unknown wavelet type!"                                          ! :46-48
      stop
    endif
    call on_device(s, n1, n2, n3, wavelet_type, 1, 'forward_wavelet')
  end subroutine forward_wavelet

  subroutine inverse_wavelet(s, n1, n2, n3, wavelet_type)
    integer, intent(in) :: n1, n2, n3, wavelet_type
    real(kind=CUSTOM_REAL), intent(inout) :: s(n1, n2, n3)
    if (wavelet_type /= 1 .and. wavelet_type /= 2) then
      print *, "Unknown wavelet type!"                                          ! :65-67
      stop
    endif
    call on_device(s, n1, n2, n3, wavelet_type, 2, 'inverse_wavelet')
  end subroutine inverse_wavelet

  subroutine Haar3D(s, n1, n2, n3)
    integer, intent(in) :: n1, n2, n3
    real(kind=CUSTOM_REAL), intent(inout) :: s(n1, n2, n3)
    call on_device(s, n1, n2, n3, 1, 1, 'Haar3D')
  end subroutine Haar3D

  subroutine iHaar3D(s, n1, n2, n3)
    integer, intent(in) :: n1, n2, n3
    real(kind=CUSTOM_REAL), intent(inout) :: s(n1, n2, n3)
    call on_device(s, n1, n2, n3, 1, 2, 'iHaar3D')
  end subroutine iHaar3D

  subroutine DaubD43D(s, n1, n2, n3)
    integer, intent(in) :: n1, n2, n3
    real(kind=CUSTOM_REAL), intent(inout) :: s(n1, n2, n3)
    call on_device(s, n1, n2, n3, 2, 1, 'DaubD43D')
  end subroutine DaubD43D

  subroutine iDaubD43D(s, n1, n2, n3)
    integer, intent(in) :: n1, n2, n3
    real(kind=CUSTOM_REAL), intent(inout) :: s(n1, n2, n3)
    call on_device(s, n1, n2, n3, 2, 2, 'iDaubD43D')
  end subroutine iDaubD43D

end module wavelet_transform
