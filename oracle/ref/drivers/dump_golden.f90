! TEST INFRASTRUCTURE (oracle/ref): our own driver, linked against the
! reference's compiled modules, that runs the reference OpenMP backend on
! deterministic inputs and dumps inputs + outputs as golden vectors.
!
!   dump_golden <input.x3d> <out-prefix>
!
! Every MPI rank writes <out-prefix>.<rank>.bin : a stream of records
!   int32 namelen, name, int32 rank, int32 dims(rank), float64 data(product(dims))
! oracle/ref/gen_golden.py stitches rank files into tests/golden/*.npz.
!
! What is exercised (all through the reference's public interfaces):
!   m_mesh mesh_t(...), m_allocator allocator_t, m_omp_backend omp_backend_t,
!   m_solver solver_t (allocate_tdsops, transeq, divergence_v2p, gradient_p2v,
!   curl), base_backend_t%tds_solve/scalar_product/field_max_mean/vecadd,
!   m_time_integrator via solver%time_integrator%step,
!   m_poisson_fft base_init (wave numbers) through a hook-less extension type,
!   m_omp_spectral process_spectral_000.
module m_dump_io
  use m_common, only: dp
  implicit none
  integer :: dump_unit = 77
contains
  subroutine dump_open(fname)
    character(*), intent(in) :: fname
    open (unit=dump_unit, file=fname, access='stream', form='unformatted', &
          status='replace')
  end subroutine
  subroutine dump_close()
    close (dump_unit)
  end subroutine
  subroutine dump_hdr(name, rank, dims)
    character(*), intent(in) :: name
    integer, intent(in) :: rank, dims(:)
    write (dump_unit) int(len_trim(name), 4)
    write (dump_unit) trim(name)
    write (dump_unit) int(rank, 4)
    write (dump_unit) int(dims(1:rank), 4)
  end subroutine
  subroutine dump_r1(name, a)
    character(*), intent(in) :: name
    real(dp), intent(in) :: a(:)
    call dump_hdr(name, 1, shape(a))
    write (dump_unit) real(a, 8)
  end subroutine
  subroutine dump_r2(name, a)
    character(*), intent(in) :: name
    real(dp), intent(in) :: a(:, :)
    call dump_hdr(name, 2, shape(a))
    write (dump_unit) real(a, 8)
  end subroutine
  subroutine dump_r3(name, a)
    character(*), intent(in) :: name
    real(dp), intent(in) :: a(:, :, :)
    call dump_hdr(name, 3, shape(a))
    write (dump_unit) real(a, 8)
  end subroutine
  subroutine dump_i1(name, a)
    character(*), intent(in) :: name
    integer, intent(in) :: a(:)
    call dump_r1(name, real(a, dp))
  end subroutine
  subroutine dump_s(name, x)
    character(*), intent(in) :: name
    real(dp), intent(in) :: x
    call dump_r1(name, [x])
  end subroutine
end module m_dump_io

module m_dummy_poisson
  !! poisson_fft_t is abstract; this extension only supplies no-op hooks so
  !! that the reference's own base_init/waves_set can be run and dumped.
  use m_common, only: dp
  use m_field, only: field_t
  use m_poisson_fft, only: poisson_fft_t
  implicit none
  type, extends(poisson_fft_t) :: dummy_poisson_t
  contains
    procedure :: fft_forward_010 => fw
    procedure :: fft_forward_100 => fw
    procedure :: fft_forward_110 => fw
    procedure :: fft_forward => fw
    procedure :: fft_backward_010 => bw
    procedure :: fft_backward_100 => bw
    procedure :: fft_backward_110 => bw
    procedure :: fft_backward => bw
    procedure :: fft_postprocess_000 => pp
    procedure :: fft_postprocess_010 => pp
    procedure :: fft_postprocess_100 => pp
    procedure :: fft_postprocess_110 => pp
    procedure :: enforce_periodicity_x => fp
    procedure :: undo_periodicity_x => fp
    procedure :: enforce_periodicity_y => fp
    procedure :: undo_periodicity_y => fp
    procedure :: enforce_periodicity_xy => fp
    procedure :: undo_periodicity_xy => fp
  end type
contains
  subroutine fw(self, f_in)
    class(dummy_poisson_t) :: self
    class(field_t), intent(in) :: f_in
  end subroutine
  subroutine bw(self, f_out)
    class(dummy_poisson_t) :: self
    class(field_t), intent(inout) :: f_out
  end subroutine
  subroutine pp(self)
    class(dummy_poisson_t) :: self
  end subroutine
  subroutine fp(self, f_out, f_in)
    class(dummy_poisson_t) :: self
    class(field_t), intent(inout) :: f_out
    class(field_t), intent(in) :: f_in
  end subroutine
end module m_dummy_poisson

program dump_golden
  use mpi
  use m_allocator
  use m_base_backend
  use m_common
  use m_config, only: domain_config_t, solver_config_t
  use m_field, only: field_t, flist_t
  use m_mesh
  use m_omp_backend
  use m_omp_common, only: SZ
  use m_omp_spectral, only: process_spectral_000, process_spectral_010
  use m_solver, only: solver_t
  use m_tdsops, only: tdsops_t, dirps_t
  use m_dump_io
  use m_dummy_poisson
  implicit none

  class(base_backend_t), pointer :: backend
  class(allocator_t), pointer :: allocator
  type(allocator_t), pointer :: host_allocator
  type(mesh_t), target :: mesh
  type(omp_backend_t), target :: omp_backend
  type(allocator_t), target :: omp_allocator
  type(domain_config_t) :: domain_cfg
  type(solver_config_t) :: solver_cfg
  type(solver_t) :: solver
  type(dummy_poisson_t) :: pois
  integer :: dims(3), cdims(3), pdims(3), nrank, nproc, ierr, dir, i, j, k, it
  integer :: nspec(3)
  character(256) :: prefix, fname
  character(8) :: rtag
  real(dp), allocatable :: buf(:, :, :), u0(:, :, :), v0(:, :, :), w0(:, :, :)
  class(field_t), pointer :: du, dv, dw, f1, f2, div_u
  type(flist_t), allocatable :: curr(:), deriv(:)
  complex(dp), allocatable :: spec(:, :, :)
  real(dp), allocatable :: sre(:, :, :), sim(:, :, :)
  real(dp) :: s, mx, mn
  logical :: all_periodic, is_010
  integer :: dg
  character(1) :: dtag

  call MPI_Init(ierr)
  call MPI_Comm_rank(MPI_COMM_WORLD, nrank, ierr)
  call MPI_Comm_size(MPI_COMM_WORLD, nproc, ierr)

  call domain_cfg%read(nml_file=get_argument(1))
  call solver_cfg%read(nml_file=get_argument(1))
  prefix = get_argument(2)
  write (rtag, '(i0)') nrank
  fname = trim(prefix)//'.'//trim(rtag)//'.bin'
  call dump_open(trim(fname))

  mesh = mesh_t(domain_cfg%dims_global, domain_cfg%nproc_dir, &
                domain_cfg%L_global, domain_cfg%BC_x, domain_cfg%BC_y, &
                domain_cfg%BC_z, domain_cfg%stretching, domain_cfg%beta, &
                use_2decomp=.false.)
  dims = mesh%get_dims(VERT)
  cdims = mesh%get_dims(CELL)
  omp_allocator = allocator_t(dims, SZ)
  allocator => omp_allocator
  host_allocator => omp_allocator
  omp_backend = omp_backend_t(mesh, allocator)
  backend => omp_backend
  solver = solver_t(backend, mesh, host_allocator)
  pdims = allocator%get_padded_dims(DIR_C)

  ! ---- mesh / decomposition metadata
  call dump_i1('meta.nproc_dir', mesh%par%nproc_dir)
  call dump_i1('meta.nrank_dir', mesh%par%nrank_dir)
  call dump_i1('meta.n_offset', mesh%par%n_offset)
  call dump_i1('meta.vert_dims', dims)
  call dump_i1('meta.cell_dims', cdims)
  call dump_i1('meta.global_vert_dims', mesh%get_global_dims(VERT))
  call dump_i1('meta.global_cell_dims', mesh%get_global_dims(CELL))
  call dump_i1('meta.BCs_x', mesh%grid%BCs(1, :))
  call dump_i1('meta.BCs_y', mesh%grid%BCs(2, :))
  call dump_i1('meta.BCs_z', mesh%grid%BCs(3, :))
  call dump_r1('meta.L', mesh%geo%L)
  call dump_r1('meta.d', mesh%geo%d)
  call dump_r1('meta.beta', mesh%geo%beta)
  call dump_s('meta.nu', solver%nu)
  call dump_s('meta.dt', solver%dt)
  do dir = 1, 3
    call dump_r1('geo.vert_coords.'//dname(dir), mesh%geo%vert_coords(1:dims(dir), dir))
    call dump_r1('geo.vert_ds.'//dname(dir), mesh%geo%vert_ds(1:dims(dir), dir))
    call dump_r1('geo.vert_ds2.'//dname(dir), mesh%geo%vert_ds2(1:dims(dir), dir))
    call dump_r1('geo.vert_d2s.'//dname(dir), mesh%geo%vert_d2s(1:dims(dir), dir))
    call dump_r1('geo.midp_coords.'//dname(dir), mesh%geo%midp_coords(1:cdims(dir), dir))
    call dump_r1('geo.midp_ds.'//dname(dir), mesh%geo%midp_ds(1:cdims(dir), dir))
  end do

  ! ---- tdsops coefficient arrays as built by the reference's allocate_tdsops
  call dump_dirps('x', solver%xdirps)
  call dump_dirps('y', solver%ydirps)
  call dump_dirps('z', solver%zdirps)

  ! ---- deterministic pseudo-random inputs, a function of the GLOBAL index
  allocate (buf(pdims(1), pdims(2), pdims(3)))
  allocate (u0(pdims(1), pdims(2), pdims(3)))
  allocate (v0(pdims(1), pdims(2), pdims(3)))
  allocate (w0(pdims(1), pdims(2), pdims(3)))
  u0 = 0._dp; v0 = 0._dp; w0 = 0._dp
  do k = 1, dims(3)
    do j = 1, dims(2)
      do i = 1, dims(1)
        u0(i, j, k) = hashval(i + mesh%par%n_offset(1), j + mesh%par%n_offset(2), &
                              k + mesh%par%n_offset(3), 1)
        v0(i, j, k) = hashval(i + mesh%par%n_offset(1), j + mesh%par%n_offset(2), &
                              k + mesh%par%n_offset(3), 2)
        w0(i, j, k) = hashval(i + mesh%par%n_offset(1), j + mesh%par%n_offset(2), &
                              k + mesh%par%n_offset(3), 3)
      end do
    end do
  end do
  call dump_r3('in.u', u0(1:dims(1), 1:dims(2), 1:dims(3)))
  call dump_r3('in.v', v0(1:dims(1), 1:dims(2), 1:dims(3)))
  call dump_r3('in.w', w0(1:dims(1), 1:dims(2), 1:dims(3)))

  call solver%u%set_data_loc(VERT)
  call solver%v%set_data_loc(VERT)
  call solver%w%set_data_loc(VERT)
  call backend%set_field_data(solver%u, u0)
  call backend%set_field_data(solver%v, v0)
  call backend%set_field_data(solver%w, w0)

  ! ---- every tds operator in every direction, applied to u (VERT) or to a
  !      CELL field for the p2v operators (n_tds decides the pencil length)
  do dir = 1, 3
    call dump_tds_all(dir)
  end do

  ! ---- transeq (3 directions + sums), divergence, gradient, curl
  du => allocator%get_block(DIR_X)
  dv => allocator%get_block(DIR_X)
  dw => allocator%get_block(DIR_X)
  allocate (curr(3), deriv(3))
  curr(1)%ptr => solver%u; curr(2)%ptr => solver%v; curr(3)%ptr => solver%w
  deriv(1)%ptr => du; deriv(2)%ptr => dv; deriv(3)%ptr => dw
  call solver%transeq(deriv, curr)
  call dump_field('transeq.du', du, VERT)
  call dump_field('transeq.dv', dv, VERT)
  call dump_field('transeq.dw', dw, VERT)

  ! x-direction part alone (no reorders involved)
  call backend%transeq_x(du, dv, dw, solver%u, solver%v, solver%w, solver%nu, &
                         solver%xdirps)
  call dump_field('transeq_x.du', du, VERT)
  call dump_field('transeq_x.dv', dv, VERT)
  call dump_field('transeq_x.dw', dw, VERT)

  ! ---- transported scalar: the reference's solver%transeq_species (x, y, z passes with reorders and sums)
  !      on a deterministic field, diffusivity 0.37 nu
  block
    class(field_t), pointer :: spec, dspec
    type(flist_t), allocatable :: curr4(:), deriv1(:)
    real(dp), allocatable :: s0(:, :, :)
    allocate (s0(pdims(1), pdims(2), pdims(3)))
    s0 = 0._dp
    do k = 1, dims(3)
      do j = 1, dims(2)
        do i = 1, dims(1)
          s0(i, j, k) = hashval(i + mesh%par%n_offset(1), j + mesh%par%n_offset(2), &
                                k + mesh%par%n_offset(3), 6)
        end do
      end do
    end do
    call dump_r3('in.s', s0(1:dims(1), 1:dims(2), 1:dims(3)))
    spec => allocator%get_block(DIR_X)
    dspec => allocator%get_block(DIR_X)
    call spec%set_data_loc(VERT)
    call backend%set_field_data(spec, s0)
    if (.not. allocated(solver%nu_species)) allocate (solver%nu_species(1))
    solver%nu_species(1) = 0.37_dp*solver%nu
    allocate (curr4(4), deriv1(1))
    curr4(1)%ptr => solver%u; curr4(2)%ptr => solver%v; curr4(3)%ptr => solver%w
    curr4(4)%ptr => spec
    deriv1(1)%ptr => dspec
    call solver%transeq_species(deriv1, curr4)
    call dump_field('species.rhs', dspec, VERT)
    call allocator%release_block(spec)
    call allocator%release_block(dspec)
  end block

  div_u => allocator%get_block(DIR_Z)
  call solver%divergence_v2p(div_u, solver%u, solver%v, solver%w)
  call dump_field('div.div_u', div_u, CELL)
  call backend%field_max_mean(mx, mn, div_u)
  call dump_s('div.max', mx)
  call dump_s('div.mean', mn)

  call solver%gradient_p2v(du, dv, dw, div_u)
  call dump_field('grad.dpdx', du, VERT)
  call dump_field('grad.dpdy', dv, VERT)
  call dump_field('grad.dpdz', dw, VERT)
  call allocator%release_block(div_u)

  call du%set_data_loc(VERT); call dv%set_data_loc(VERT); call dw%set_data_loc(VERT)
  call solver%curl(du, dv, dw, solver%u, solver%v, solver%w)
  call dump_field('curl.i', du, VERT)
  call dump_field('curl.j', dv, VERT)
  call dump_field('curl.k', dw, VERT)
  s = 0.5_dp*(backend%scalar_product(du, du) + backend%scalar_product(dv, dv) &
              + backend%scalar_product(dw, dw))/solver%ngrid
  call dump_s('curl.enstrophy', s)

  ! ---- two full substeps of transeq + time integrator (no pressure solve):
  !      state after each substep
  do it = 1, 2*solver%time_integrator%nstage
    call solver%transeq(deriv, curr)
    call solver%time_integrator%step(curr, deriv, solver%dt)
    if (it == solver%time_integrator%nstage) then
      call dump_field('step1.u', solver%u, VERT)
      call dump_field('step1.v', solver%v, VERT)
      call dump_field('step1.w', solver%w, VERT)
    end if
  end do
  call dump_field('step2.u', solver%u, VERT)
  call dump_field('step2.v', solver%v, VERT)
  call dump_field('step2.w', solver%w, VERT)

  ! ---- spectral: wave numbers from the reference's base_init and the
  !      reference's process_spectral_000 on a deterministic complex array
  all_periodic = all(mesh%grid%periodic_BC)
  if (nproc == 1 .and. all_periodic) then
    nspec = [cdims(1)/2 + 1, cdims(2), cdims(3)]
    call pois%base_init(mesh, solver%xdirps, solver%ydirps, solver%zdirps, &
                        nspec, [0, 0, 0])
    call dump_r1('spec.ax', pois%ax); call dump_r1('spec.bx', pois%bx)
    call dump_r1('spec.ay', pois%ay); call dump_r1('spec.by', pois%by)
    call dump_r1('spec.az', pois%az); call dump_r1('spec.bz', pois%bz)
    call dump_r1('spec.k2x_re', real(pois%k2x, dp))
    call dump_r1('spec.k2y_re', real(pois%k2y, dp))
    call dump_r1('spec.k2z_re', real(pois%k2z, dp))
    allocate (sre(nspec(1), nspec(2), nspec(3)), sim(nspec(1), nspec(2), nspec(3)))
    sre = real(pois%waves, dp); sim = aimag(pois%waves)
    call dump_r3('spec.waves_re', sre)
    call dump_r3('spec.waves_im', sim)
    allocate (spec(nspec(1), nspec(2), nspec(3)))
    do k = 1, nspec(3)
      do j = 1, nspec(2)
        do i = 1, nspec(1)
          spec(i, j, k) = cmplx(hashval(i, j, k, 4), hashval(i, j, k, 5), kind=dp)
        end do
      end do
    end do
    sre = real(spec, dp); sim = aimag(spec)
    call dump_r3('spec.in_re', sre)
    call dump_r3('spec.in_im', sim)
    call process_spectral_000( &
      spec, pois%waves, nspec(1), nspec(2), nspec(3), 0, 0, 0, &
      cdims(1), cdims(2), cdims(3), &
      pois%ax, pois%bx, pois%ay, pois%by, pois%az, pois%bz)
    sre = real(spec, dp); sim = aimag(spec)
    call dump_r3('spec.out_re', sre)
    call dump_r3('spec.out_im', sim)
  end if

  ! ---- spectral, non-periodic y (010): base_init -> waves, transfer functions, the
  !      stretching matrices (src/poisson_fft.f90:275-652) and the reference's
  !      process_spectral_010 (OMP kernel; it ignores stretching) on a deterministic array
  is_010 = mesh%grid%periodic_BC(1) .and. (.not. mesh%grid%periodic_BC(2)) &
           .and. mesh%grid%periodic_BC(3)
  if (nproc == 1 .and. is_010) then
    nspec = [cdims(1)/2 + 1, cdims(2), cdims(3)]
    call pois%base_init(mesh, solver%xdirps, solver%ydirps, solver%zdirps, &
                        nspec, [0, 0, 0])
    call dump_r1('spec.ax', pois%ax); call dump_r1('spec.bx', pois%bx)
    call dump_r1('spec.ay', pois%ay); call dump_r1('spec.by', pois%by)
    call dump_r1('spec.az', pois%az); call dump_r1('spec.bz', pois%bz)
    call dump_r1('spec.k2x_re', real(pois%k2x, dp))
    call dump_r1('spec.k2y_re', real(pois%k2y, dp))
    call dump_r1('spec.k2z_re', real(pois%k2z, dp))
    call dump_r1('spec.kx_re', real(pois%kx, dp))
    call dump_r1('spec.ky_re', real(pois%ky, dp))
    call dump_r1('spec.kz_re', real(pois%kz, dp))
    allocate (sre(nspec(1), nspec(2), nspec(3)), sim(nspec(1), nspec(2), nspec(3)))
    sre = real(pois%waves, dp); sim = aimag(pois%waves)
    call dump_r3('spec.waves_re', sre)
    call dump_r3('spec.waves_im', sim)
    if (pois%stretched_y) then
      call dump_r1('spec.trans_x', pois%trans_x_re)
      call dump_r1('spec.trans_y', pois%trans_y_re)
      call dump_r1('spec.trans_z', pois%trans_z_re)
      call dump_s('spec.stretched_y_sym', merge(1._dp, 0._dp, pois%stretched_y_sym))
      do dg = 1, 5
        write (dtag, '(I1)') dg
        if (pois%stretched_y_sym) then
          call dump_r3('spec.a_odd_re.'//dtag, pois%a_odd_re(:, :, :, dg))
          call dump_r3('spec.a_odd_im.'//dtag, pois%a_odd_im(:, :, :, dg))
          call dump_r3('spec.a_even_re.'//dtag, pois%a_even_re(:, :, :, dg))
          call dump_r3('spec.a_even_im.'//dtag, pois%a_even_im(:, :, :, dg))
        else
          call dump_r3('spec.a_re.'//dtag, pois%a_re(:, :, :, dg))
          call dump_r3('spec.a_im.'//dtag, pois%a_im(:, :, :, dg))
        end if
      end do
    end if
    allocate (spec(nspec(1), nspec(2), nspec(3)))
    do k = 1, nspec(3)
      do j = 1, nspec(2)
        do i = 1, nspec(1)
          spec(i, j, k) = cmplx(hashval(i, j, k, 4), hashval(i, j, k, 5), kind=dp)
        end do
      end do
    end do
    sre = real(spec, dp); sim = aimag(spec)
    call dump_r3('spec.in_re', sre)
    call dump_r3('spec.in_im', sim)
    call process_spectral_010( &
      spec, pois%waves, nspec(1), nspec(2), nspec(3), 0, 0, 0, &
      cdims(1), cdims(2), cdims(3), &
      pois%ax, pois%bx, pois%ay, pois%by, pois%az, pois%bz)
    sre = real(spec, dp); sim = aimag(spec)
    call dump_r3('spec.out010_re', sre)
    call dump_r3('spec.out010_im', sim)
  end if

  call dump_close()
  call MPI_Finalize(ierr)

contains

  function dname(dir) result(c)
    integer, intent(in) :: dir
    character(1) :: c
    c = 'xyz' (dir:dir)
  end function

  pure function hashval(i, j, k, salt) result(r)
    !! cheap deterministic pseudo-random value in (-1, 1); only its dumped
    !! value matters (tests read it back from the fixture)
    integer, intent(in) :: i, j, k, salt
    real(dp) :: r, t
    t = sin(real(i, dp)*12.9898_dp + real(j, dp)*78.233_dp &
            + real(k, dp)*37.719_dp + real(salt, dp)*4.581_dp)*43758.5453_dp
    r = 2._dp*(t - floor(t)) - 1._dp
  end function

  subroutine dump_field(name, f, loc)
    character(*), intent(in) :: name
    class(field_t), intent(in) :: f
    integer, intent(in) :: loc
    integer :: d(3)
    d = mesh%get_dims(loc)
    call backend%get_field_data(buf, f)
    call dump_r3(name, buf(1:d(1), 1:d(2), 1:d(3)))
  end subroutine

  subroutine dump_tdsops(tag, t)
    character(*), intent(in) :: tag
    class(tdsops_t), intent(in) :: t
    real(dp) :: sc(8)
    sc = [real(t%n_tds, dp), real(t%n_rhs, dp), real(t%move, dp), &
          merge(1._dp, 0._dp, t%periodic), t%alpha, t%a, t%b, t%c]
    call dump_r1(tag//'.scalars', sc)
    call dump_s(tag//'.d', t%d)
    call dump_r1(tag//'.coeffs', t%coeffs)
    call dump_r2(tag//'.coeffs_s', t%coeffs_s)
    call dump_r2(tag//'.coeffs_e', t%coeffs_e)
    call dump_r1(tag//'.dist_fw', t%dist_fw)
    call dump_r1(tag//'.dist_bw', t%dist_bw)
    call dump_r1(tag//'.dist_sa', t%dist_sa)
    call dump_r1(tag//'.dist_sc', t%dist_sc)
    call dump_r1(tag//'.dist_af', t%dist_af)
    call dump_r1(tag//'.stretch', t%stretch)
    call dump_r1(tag//'.stretch_correct', t%stretch_correct)
  end subroutine

  subroutine dump_dirps(d, p)
    character(1), intent(in) :: d
    type(dirps_t), intent(in) :: p
    call dump_tdsops('tdsops.'//d//'.der1st', p%der1st)
    call dump_tdsops('tdsops.'//d//'.der1st_sym', p%der1st_sym)
    call dump_tdsops('tdsops.'//d//'.der2nd', p%der2nd)
    call dump_tdsops('tdsops.'//d//'.der2nd_sym', p%der2nd_sym)
    call dump_tdsops('tdsops.'//d//'.stagder_v2p', p%stagder_v2p)
    call dump_tdsops('tdsops.'//d//'.stagder_p2v', p%stagder_p2v)
    call dump_tdsops('tdsops.'//d//'.interpl_v2p', p%interpl_v2p)
    call dump_tdsops('tdsops.'//d//'.interpl_p2v', p%interpl_p2v)
  end subroutine

  subroutine dump_tds_one(tag, dir, t, src_x, loc_in)
    !! result = tds_solve(t) applied along `dir` to the DIR_X field src_x
    character(*), intent(in) :: tag
    integer, intent(in) :: dir, loc_in
    class(tdsops_t), intent(in) :: t
    class(field_t), intent(in) :: src_x
    class(field_t), pointer :: a, b
    integer :: rdr, loc_out, d(3)
    a => allocator%get_block(dir)
    b => allocator%get_block(dir)
    if (dir == DIR_X) then
      call backend%veccopy(a, src_x)
      call a%set_data_loc(src_x%data_loc)
    else
      rdr = merge(RDR_X2Y, RDR_X2Z, dir == DIR_Y)
      call backend%reorder(a, src_x, rdr)
    end if
    call backend%tds_solve(b, a, t)
    loc_out = b%data_loc
    d = mesh%get_dims(loc_out)
    call backend%get_field_data(buf, b)
    call dump_r3(tag, buf(1:d(1), 1:d(2), 1:d(3)))
    call allocator%release_block(a)
    call allocator%release_block(b)
  end subroutine

  subroutine dump_tds_all(dir)
    integer, intent(in) :: dir
    type(dirps_t), pointer :: p
    class(field_t), pointer :: c
    character(1) :: d
    integer :: loc_c
    d = dname(dir)
    select case (dir)
    case (1); p => solver%xdirps
    case (2); p => solver%ydirps
    case (3); p => solver%zdirps
    end select
    call dump_tds_one('tds.'//d//'.der1st', dir, p%der1st, solver%u, VERT)
    call dump_tds_one('tds.'//d//'.der1st_sym', dir, p%der1st_sym, solver%u, VERT)
    call dump_tds_one('tds.'//d//'.der2nd', dir, p%der2nd, solver%u, VERT)
    call dump_tds_one('tds.'//d//'.der2nd_sym', dir, p%der2nd_sym, solver%u, VERT)
    call dump_tds_one('tds.'//d//'.stagder_v2p', dir, p%stagder_v2p, solver%u, VERT)
    call dump_tds_one('tds.'//d//'.interpl_v2p', dir, p%interpl_v2p, solver%u, VERT)
    ! p2v operators act on data that is cell-centred along `dir` only:
    ! reuse u's values but tag the field as staggered in `dir`
    c => allocator%get_block(DIR_X)
    call backend%veccopy(c, solver%u)
    loc_c = move_data_loc(VERT, dir, 1)
    call c%set_data_loc(loc_c)
    call dump_tds_one('tds.'//d//'.stagder_p2v', dir, p%stagder_p2v, c, loc_c)
    call dump_tds_one('tds.'//d//'.interpl_p2v', dir, p%interpl_p2v, c, loc_c)
    call allocator%release_block(c)
  end subroutine

end program dump_golden
