! Golden-vector driver (OUR code): calls the reference's forward_wavelet / inverse_wavelet
! (src/utils/wavelet_transform.F90:37-70) on an array read from a big-endian stream file.
! stdin: n1 n2 n3 wavelet_type direction(1 fwd, 2 inv) / infile / outfile
program gold_wavelet
  use global_typedefs
  use wavelet_transform
  implicit none
  integer :: n1, n2, n3, wtype, dir
  character(len=512) :: fin, fout
  real(kind=CUSTOM_REAL), allocatable :: s(:, :, :)
  read(*, *) n1, n2, n3, wtype, dir
  read(*, '(a)') fin
  read(*, '(a)') fout
  allocate(s(n1, n2, n3))
  open(21, file=trim(fin), form='unformatted', access='stream', status='old', action='read')
  read(21) s
  close(21)
  if (dir == 1) then
    call forward_wavelet(s, n1, n2, n3, wtype)
  else
    call inverse_wavelet(s, n1, n2, n3, wtype)
  endif
  open(22, file=trim(fout), form='unformatted', access='stream', status='replace', action='write')
  write(22) s
  close(22)
end program gold_wavelet
