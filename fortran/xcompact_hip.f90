!> Main program: src/xcompact.f90 with the third (HIP) backend branch.
!> Everything after backend construction is the reference's own code path.
program xcompact_hip
  use mpi
  use m_allocator
  use m_base_backend
  use m_base_case, only: base_case_t
  use m_common, only: pi, get_argument, VERT, dp
  use m_config, only: domain_config_t, solver_config_t
  use m_mesh
  use m_case_channel, only: case_channel_t
  use m_case_generic, only: case_generic_t
  use m_case_tgv, only: case_tgv_t
  use m_hip_allocator, only: hip_allocator_t, hip_allocator_init
  use m_hip_backend, only: hip_backend_t, hip_backend_init
  use m_hip_common, only: SZ
  implicit none

  class(base_backend_t), pointer :: backend
  class(allocator_t), pointer :: allocator
  type(allocator_t), pointer :: host_allocator
  type(mesh_t), target :: mesh
  class(base_case_t), allocatable :: flow_case
  type(hip_backend_t), target :: hip_backend
  type(hip_allocator_t), target :: hip_allocator
  type(allocator_t), target :: host_alloc
  type(domain_config_t) :: domain_cfg
  type(solver_config_t) :: solver_cfg
  integer :: dims(3), nrank, nproc, ierr

  call MPI_Init(ierr)
  call MPI_Comm_rank(MPI_COMM_WORLD, nrank, ierr)
  call MPI_Comm_size(MPI_COMM_WORLD, nproc, ierr)
  if (nrank == 0) print *, 'Parallel run with', nproc, 'ranks; backend: HIP (MI355X)'

  call domain_cfg%read(nml_file=get_argument(1))
  call solver_cfg%read(nml_file=get_argument(1))
  if (product(domain_cfg%nproc_dir) /= nproc) domain_cfg%nproc_dir = [1, 1, nproc]

  mesh = mesh_t(domain_cfg%dims_global, domain_cfg%nproc_dir, domain_cfg%L_global, &
                domain_cfg%BC_x, domain_cfg%BC_y, domain_cfg%BC_z, domain_cfg%stretching, &
                domain_cfg%beta, use_2decomp=.false.)
  dims = mesh%get_dims(VERT)

  hip_allocator = hip_allocator_init(dims, SZ, 0)
  allocator => hip_allocator
  host_alloc = allocator_t(dims, SZ)
  host_allocator => host_alloc
  hip_backend = hip_backend_init(mesh, allocator)
  backend => hip_backend
  if (nrank == 0) print *, 'HIP backend instantiated'

  select case (trim(domain_cfg%flow_case_name))
  case ('channel')
    allocate (case_channel_t :: flow_case)
    flow_case = case_channel_t(backend, mesh, host_allocator)
  case ('generic')
    allocate (case_generic_t :: flow_case)
    flow_case = case_generic_t(backend, mesh, host_allocator)
  case ('tgv')
    allocate (case_tgv_t :: flow_case)
    flow_case = case_tgv_t(backend, mesh, host_allocator)
  case default
    error stop 'Undefined flow_case.'
  end select
  if (nrank == 0) print *, 'solver instantiated'

  call flow_case%run()
  call MPI_Finalize(ierr)
end program xcompact_hip
