! Golden-vector driver (OUR code): calls the reference's gradiprism_zz / gradiprism_full
! (src/forward/gravmag/grav/gravity_field.f90:315-362, :207-310) for a list of observation points.
! stdin: nel ndata / gridfile / obsfile / outfile
! outfile: per observation the zz line (nel), then the six full-tensor lines in the order the build stores them
! (src/forward/gravmag/sensitivity_gravmag.F90:210-212): XX, YY, ZZ, XY, YZ, ZX.
program gold_gradprism
  use global_typedefs
  use grid
  use gravity_field
  implicit none
  integer :: nel, ndata, i
  character(len=512) :: fgrid, fobs, fout
  type(t_grid) :: g
  real(kind=CUSTOM_REAL), allocatable :: xd(:), yd(:), zd(:), line(:), full(:, :)
  read(*, *) nel, ndata
  read(*, '(a)') fgrid
  read(*, '(a)') fobs
  read(*, '(a)') fout
  allocate(g%X1(nel), g%X2(nel), g%Y1(nel), g%Y2(nel), g%Z1(nel), g%Z2(nel))
  allocate(xd(ndata), yd(ndata), zd(ndata), line(nel), full(nel, 6))
  open(21, file=trim(fgrid), form='unformatted', access='stream', status='old', action='read')
  read(21) g%X1, g%X2, g%Y1, g%Y2, g%Z1, g%Z2
  close(21)
  open(21, file=trim(fobs), form='unformatted', access='stream', status='old', action='read')
  read(21) xd, yd, zd
  close(21)
  open(22, file=trim(fout), form='unformatted', access='stream', status='replace', action='write')
  do i = 1, ndata
    call gradiprism_zz(nel, g, xd(i), yd(i), zd(i), line)
    write(22) line
    call gradiprism_full(nel, g, xd(i), yd(i), zd(i), full(:, 1), full(:, 2), full(:, 3), full(:, 4), full(:, 5), &
                         full(:, 6), 0)
    write(22) full
  enddo
  close(22)
end program gold_gradprism
