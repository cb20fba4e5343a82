! Golden-vector driver (OUR code): calls the reference's graviprism_z
! (src/forward/gravmag/grav/gravity_field.f90:131-195) or, with full = 1, graviprism_full (:41-126)
! for a list of observation points.
! stdin: nel ndata [full] / gridfile (X1,X2,Y1,Y2,Z1,Z2 each nel fp64) / obsfile (X,Y,Z each ndata) /
!        outfile (full = 0: ndata rows of nel; full = 1: per observation LineX, LineY, LineZ of nel each)
program gold_prism
  use global_typedefs
  use grid
  use gravity_field
  implicit none
  integer :: nel, ndata, i, full, ios
  character(len=512) :: fgrid, fobs, fout, head
  type(t_grid) :: g
  real(kind=CUSTOM_REAL), allocatable :: xd(:), yd(:), zd(:), line(:), lx(:), ly(:)
  read(*, '(a)') head
  full = 0
  read(head, *, iostat=ios) nel, ndata, full
  if (ios /= 0) then
    full = 0
    read(head, *) nel, ndata
  endif
  read(*, '(a)') fgrid
  read(*, '(a)') fobs
  read(*, '(a)') fout
  allocate(g%X1(nel), g%X2(nel), g%Y1(nel), g%Y2(nel), g%Z1(nel), g%Z2(nel))
  allocate(xd(ndata), yd(ndata), zd(ndata), line(nel), lx(nel), ly(nel))
  open(21, file=trim(fgrid), form='unformatted', access='stream', status='old', action='read')
  read(21) g%X1, g%X2, g%Y1, g%Y2, g%Z1, g%Z2
  close(21)
  open(21, file=trim(fobs), form='unformatted', access='stream', status='old', action='read')
  read(21) xd, yd, zd
  close(21)
  open(22, file=trim(fout), form='unformatted', access='stream', status='replace', action='write')
  do i = 1, ndata
    if (full == 1) then
      call graviprism_full(nel, g, xd(i), yd(i), zd(i), lx, ly, line, 0)
      write(22) lx
      write(22) ly
      write(22) line
    else
      call graviprism_z(nel, g, xd(i), yd(i), zd(i), line, 0)
      write(22) line
    endif
  enddo
  close(22)
end program gold_prism
