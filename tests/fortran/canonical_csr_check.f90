! Test driver (OUR code): tfx_reference_api's api_canonical_csr on rows as t_sparse_matrix%add builds them - any column order, repeated
! columns, empty rows (src/inversion/sparse_matrix.f90:213-229).  Prints the canonical rows; tests/test_host_builders.py checks them.
program canonical_csr_check
  use iso_c_binding
  use tfx_reference_api, only: api_canonical_csr
  implicit none
  integer, parameter :: nl = 5
  integer(c_int64_t) :: rowptr(nl + 1) = [0_c_int64_t, 4_c_int64_t, 4_c_int64_t, 7_c_int64_t, 8_c_int64_t, 13_c_int64_t]
  integer(c_int32_t) :: ija(13) = [7, 2, 7, 1,   3, 3, 3,   9,   5, 4, 5, 4, 1]
  real(c_float) :: sa(13) = [1.0, 2.0, 0.5, 4.0,   1.0, -1.0, 0.25,   8.0,   1.0, 2.0, 3.0, 4.0, 5.0]
  integer(c_int64_t), allocatable :: rp(:), where(:)
  integer(c_int32_t), allocatable :: cols(:)
  real(c_float), allocatable :: vals(:)
  integer :: r
  integer(c_int64_t) :: k
  call api_canonical_csr(nl, rowptr, ija, sa, rp, cols, vals, where)
  print '(a,6(1x,i0))', 'rp', rp
  do r = 1, nl
    do k = rp(r) + 1, rp(r + 1)
      print '(a,1x,i0,1x,i0,1x,es14.7)', 'entry', r, cols(k), vals(k)
    enddo
  enddo
  print '(a,13(1x,i0))', 'where', where(1:13)
end program canonical_csr_check
