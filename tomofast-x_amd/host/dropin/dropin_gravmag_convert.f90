!=========================================================================================================
! Helper of the drop-in modules (build recipe: INTEGRATION.md 0): the reference's own parameter / grid / data objects (its modules
! parameters_gravmag, parameters_grav, parameters_mag, grid, data_gravmag - compiled unmodified) -> the plain copies the C-ABI
! layer tfx_reference_api works on.  Field by field, so the compiler checks every name against the reference's definitions
! (parameters_gravmag.f90:30-108, parameters_mag.f90:30-48, grid.F90:30-50, data_gravmag.f90:30-52).  The repository's own code.
!=========================================================================================================
module dropin_gravmag_convert
  use global_typedefs
  use parameters_gravmag
  use parameters_grav
  use parameters_mag
  use grid
  use data_gravmag
  use tfx_reference_api, only: api_par_base => t_parameters_base, api_par_grav => t_parameters_grav, api_par_mag => t_parameters_mag, &
                               api_grid => t_grid, api_data => t_data
  implicit none
  private

  public :: to_api_parameters, to_api_grid, to_api_data, problem_type_of

contains

  integer function problem_type_of(par)
    class(t_parameters_base), intent(in) :: par
    problem_type_of = 1
    select type (par)
    class is (t_parameters_mag)
      problem_type_of = 2
    end select
  end function problem_type_of

  ! sensit_read: 0 calculate, 1 read kernel and depth weight from files, 2 calculate with the depth weight read from file
  ! (problem_joint_gravmag.F90:170-203) - the C-ABI layer only tells "read the files" (1) from "calculate" (0)
  subroutine to_api_parameters(par, out)
    class(t_parameters_base), intent(in) :: par
    class(api_par_base), allocatable, intent(out) :: out
    select type (par)
    class is (t_parameters_mag)
      allocate(api_par_mag :: out)
      select type (out)
      class is (api_par_mag)
        out%mi = par%mi
        out%md = par%md
        out%theta = par%theta
        out%intensity = par%intensity
      end select
    class default
      allocate(api_par_grav :: out)
    end select
    out%nx = par%nx; out%ny = par%ny; out%nz = par%nz
    out%nelements = par%nelements
    out%ndata = par%ndata
    out%ndata_components = par%ndata_components
    out%nmodel_components = par%nmodel_components
    out%data_type = par%data_type
    out%depth_weighting_type = par%depth_weighting_type
    out%depth_weighting_power = par%depth_weighting_power
    out%depth_weighting_beta = par%depth_weighting_beta
    out%Z0 = par%Z0
    out%compression_type = par%compression_type
    out%compression_rate = par%compression_rate
    out%sensit_read = merge(1, 0, par%sensit_read == 1)
    if (par%sensit_read == 1) then
      out%sensit_path = par%sensit_path                                   ! sensitivity_gravmag.F90:668-670
    else
      out%sensit_path = trim(path_output)//'/SENSIT/'                     ! :142-146
    endif
    out%sensit_write = 1                                                  ! the reference always writes the kernel files
  end subroutine to_api_parameters

  subroutine to_api_grid(g, out)
    type(t_grid), intent(in) :: g
    type(api_grid), intent(out) :: out
    out%nx = g%nx; out%ny = g%ny; out%nz = g%nz
    out%z_axis_dir = g%z_axis_dir
    allocate(out%X1, source=g%X1)
    allocate(out%X2, source=g%X2)
    allocate(out%Y1, source=g%Y1)
    allocate(out%Y2, source=g%Y2)
    allocate(out%Z1, source=g%Z1)
    allocate(out%Z2, source=g%Z2)
  end subroutine to_api_grid

  subroutine to_api_data(d, out)
    type(t_data), intent(in) :: d
    type(api_data), intent(out) :: out
    out%ndata = d%ndata
    out%ncomponents = d%ncomponents
    out%units_mult = d%units_mult
    out%z_axis_dir = d%z_axis_dir
    allocate(out%X, source=d%X)
    allocate(out%Y, source=d%Y)
    allocate(out%Z, source=d%Z)
    if (allocated(d%weight)) allocate(out%weight, source=d%weight)
  end subroutine to_api_data

end module dropin_gravmag_convert
