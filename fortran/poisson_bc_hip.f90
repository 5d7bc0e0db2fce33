!> Acceptance run of the FFT Poisson solver THROUGH THE SHIM, after the checks of the reference's
!> tests/verification/test_poisson_bc.f90 (which instantiates the OpenMP or CUDA backend itself and so cannot be
!> pointed at a third backend): f = product of cosines on the cell centres, solve, compare with the analytic
!> solution up to a constant; tolerance 1e-11 on norm2(err)/N.
!>   poisson_bc_hip <config>     config = 000 | 010 | 100 | 110
!>                               (sizes of that test: 128x64x32, 128x65x32, 129x64x32, 129x257x64)
!> n = 2 in every direction combination, n = 3 in the non-periodic direction (the test's n = 3 cases in periodic
!> directions are its XFAILs).  Exit code 1 on failure.
program poisson_bc_hip
  use mpi
  use m_allocator, only: allocator_t
  use m_base_backend, only: base_backend_t
  use m_common, only: dp, pi, DIR_C, CELL, VERT, get_argument
  use m_field, only: field_t
  use m_mesh, only: mesh_t
  use m_solver, only: allocate_tdsops
  use m_tdsops, only: dirps_t
  use m_hip_allocator, only: hip_allocator_t, hip_allocator_init
  use m_hip_backend, only: hip_backend_t, hip_backend_init
  use m_hip_common, only: SZ
  implicit none

  class(base_backend_t), pointer :: backend
  class(allocator_t), pointer :: allocator
  type(allocator_t), target :: host_alloc
  type(hip_backend_t), target :: hip_backend
  type(hip_allocator_t), target :: hip_allocator
  type(mesh_t), target :: mesh
  type(dirps_t), pointer :: xdirps, ydirps, zdirps
  character(len=20) :: BC_x(2), BC_y(2), BC_z(2)
  character(len=8) :: config
  integer :: dims_global(3), ierr, kind, n, nfail
  logical :: per(3)

  call MPI_Init(ierr)
  config = get_argument(1)
  BC_x = 'periodic'; BC_y = 'periodic'; BC_z = 'periodic'
  dims_global = [128, 64, 32]
  select case (trim(config))
  case ('000')
  case ('010')
    BC_y = 'dirichlet'; dims_global(2) = 65
  case ('100')
    BC_x = 'dirichlet'; dims_global(1) = 129
  case ('110')
    BC_x = 'dirichlet'; BC_y = 'dirichlet'; dims_global = [129, 257, 64]
  case default
    error stop 'poisson_bc_hip: config must be 000, 010, 100 or 110'
  end select
  per = [trim(BC_x(1)) == 'periodic', trim(BC_y(1)) == 'periodic', trim(BC_z(1)) == 'periodic']

  mesh = mesh_t(dims_global, [1, 1, 1], [1._dp, 1._dp, 1._dp], BC_x, BC_y, BC_z, use_2decomp=.false.)
  hip_allocator = hip_allocator_init(mesh%get_dims(VERT), SZ, 0)
  allocator => hip_allocator
  host_alloc = allocator_t(mesh%get_dims(VERT), SZ)
  hip_backend = hip_backend_init(mesh, allocator)
  backend => hip_backend

  allocate (xdirps, ydirps, zdirps)
  xdirps%dir = 1; ydirps%dir = 2; zdirps%dir = 3
  call allocate_tdsops(xdirps, backend, mesh, 'compact6', 'compact6', 'classic', 'compact6')
  call allocate_tdsops(ydirps, backend, mesh, 'compact6', 'compact6', 'classic', 'compact6')
  call allocate_tdsops(zdirps, backend, mesh, 'compact6', 'compact6', 'classic', 'compact6')
  call backend%init_poisson_fft(mesh, xdirps, ydirps, zdirps)

  nfail = 0
  do kind = 1, 4   ! cos x, cos y, cos x cos y, cos x cos y cos z
    call one(2, kind)
  end do
  do kind = 1, 2   ! n = 3 where the direction is not periodic
    if (.not. per(kind)) call one(3, kind)
  end do
  if (.not. per(1) .and. .not. per(2)) call one(3, 3)
  if (nfail > 0) then
    print *, 'poisson_bc_hip ', trim(config), ': FAILED cases: ', nfail
    call MPI_Finalize(ierr)
    stop 1
  end if
  print *, 'poisson_bc_hip ', trim(config), ': PASS'
  call MPI_Finalize(ierr)

contains

  real(dp) function cosines(c, k, kind) result(v)
    real(dp), intent(in) :: c(3), k
    integer, intent(in) :: kind
    select case (kind)
    case (1); v = cos(k*c(1))
    case (2); v = cos(k*c(2))
    case (3); v = cos(k*c(1))*cos(k*c(2))
    case default; v = cos(k*c(1))*cos(k*c(2))*cos(k*c(3))
    end select
  end function cosines

  subroutine one(n_wave, kind)
    integer, intent(in) :: n_wave, kind
    class(field_t), pointer :: f, temp, host
    integer :: i, j, k, d(3)
    real(dp) :: kk, den, err, e, shift
    d = mesh%get_dims(CELL)
    kk = n_wave*pi
    den = real(max(kind - 1, 1), dp)*kk*kk   ! 1, 1, 2, 3 cosine factors: -lap = (number of factors) k^2
    f => backend%allocator%get_block(DIR_C, CELL)
    temp => backend%allocator%get_block(DIR_C)
    host => host_alloc%get_block(DIR_C)
    host%data = 0._dp
    do k = 1, d(3); do j = 1, d(2); do i = 1, d(1)
      host%data(i, j, k) = cosines(mesh%get_coordinates(i, j, k, CELL), kk, kind)
    end do; end do; end do
    call backend%set_field_data(f, host%data, DIR_C)
    call f%set_data_loc(CELL)
    call backend%poisson_fft%solve_poisson(f, temp)
    call backend%get_field_data(host%data, f)
    shift = host%data(1, 1, 1) + cosines(mesh%get_coordinates(1, 1, 1, CELL), kk, kind)/den
    err = 0._dp
    do k = 1, d(3); do j = 1, d(2); do i = 1, d(1)
      e = host%data(i, j, k) - shift + cosines(mesh%get_coordinates(i, j, k, CELL), kk, kind)/den
      err = err + e*e
    end do; end do; end do
    err = sqrt(err)/product(d)
    print '(A,I2,A,I2,A,ES10.3)', '   n =', n_wave, '  kind', kind, '  norm2(err)/N =', err
    if (.not. (err <= 1e-11_dp)) nfail = nfail + 1
    call backend%allocator%release_block(f)
    call backend%allocator%release_block(temp)
    call host_alloc%release_block(host)
  end subroutine one

end program poisson_bc_hip
