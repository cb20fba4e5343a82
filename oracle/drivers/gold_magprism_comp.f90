! Golden-vector driver (OUR code): the reference's t_magnetic_field%magprism
! (src/forward/gravmag/mag/magnetic_field.f90:118-297) with any supported component counts.
! stdin: nel ndata ncm ncd incl decl azim intensity / gridfile / obsfile / outfile
! outfile: per observation sensit_line(nel, ncm, ncd) in Fortran order.
program gold_magprism_comp
  use global_typedefs
  use grid
  use magnetic_field
  implicit none
  integer :: nel, ndata, ncm, ncd, i
  character(len=512) :: fgrid, fobs, fout
  type(t_grid) :: g
  type(t_magnetic_field) :: mf
  real(kind=CUSTOM_REAL) :: incl, decl, azim, intensity
  real(kind=CUSTOM_REAL), allocatable :: xd(:), yd(:), zd(:), line(:, :, :)
  read(*, *) nel, ndata, ncm, ncd, incl, decl, azim, intensity
  read(*, '(a)') fgrid
  read(*, '(a)') fobs
  read(*, '(a)') fout
  allocate(g%X1(nel), g%X2(nel), g%Y1(nel), g%Y2(nel), g%Z1(nel), g%Z2(nel))
  allocate(xd(ndata), yd(ndata), zd(ndata), line(nel, ncm, ncd))
  open(21, file=trim(fgrid), form='unformatted', access='stream', status='old', action='read')
  read(21) g%X1, g%X2, g%Y1, g%Y2, g%Z1, g%Z2
  close(21)
  open(21, file=trim(fobs), form='unformatted', access='stream', status='old', action='read')
  read(21) xd, yd, zd
  close(21)
  call mf%initialize(incl, decl, azim, intensity)
  open(22, file=trim(fout), form='unformatted', access='stream', status='replace', action='write')
  do i = 1, ndata
    call mf%magprism(nel, ncm, ncd, g, xd(i), yd(i), zd(i), line)
    write(22) line
  enddo
  close(22)
end program gold_magprism_comp
