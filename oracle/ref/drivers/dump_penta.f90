! TEST INFRASTRUCTURE (oracle/ref): our own driver, linked against the reference's compiled modules, that
! runs the reference's compact10_penta first derivative (tdsops_init + exec_dist_penta_compact /
! exec_dist_penta_periodic, src/tdsops.f90:235-251, 971-1103, src/backend/omp/exec_dist.f90:188-241) on
! deterministic inputs and dumps coefficients, inputs and outputs as golden vectors.
!
!   dump_penta <out.bin>
!
! Record format as dump_golden.f90: int32 namelen, name, int32 rank, int32 dims(rank), float64 data.
! Cases (set-up of tests/verification/test_omp_penta.f90, with lane-dependent data instead of one profile):
!   dd  BC_DIRICHLET both ends, halos zero
!   nt  BC_NEUMANN sym = .true.,  even mirror ghosts
!   nf  BC_NEUMANN sym = .false., odd mirror ghosts
!   pp  BC_PERIODIC, wrap-around halos
module m_dump_io2
  use m_common, only: dp
  implicit none
  integer :: dump_unit = 78
contains
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
end module m_dump_io2

program dump_penta
  use mpi
  use m_common, only: dp, pi, BC_PERIODIC, BC_DIRICHLET, BC_NEUMANN
  use m_omp_common, only: SZ
  use m_omp_exec_dist, only: exec_dist_penta_compact, exec_dist_penta_periodic
  use m_tdsops, only: tdsops_t, tdsops_init
  use m_dump_io2
  implicit none

  integer, parameter :: n = 40, n_block = 2, n_halo = 4
  character(len=512) :: fname
  integer :: ierr

  call MPI_Init(ierr)
  call get_command_argument(1, fname)
  open (unit=dump_unit, file=trim(fname), access='stream', form='unformatted', status='replace')
  call one_case('dd', BC_DIRICHLET, .false., 1._dp/real(n + 1, dp))
  call one_case('nt', BC_NEUMANN, .true., 1._dp/real(n - 1, dp))
  call one_case('nf', BC_NEUMANN, .false., 1._dp/real(n - 1, dp))
  call one_case('pp', BC_PERIODIC, .false., 1._dp/real(n, dp))
  close (dump_unit)
  call MPI_Finalize(ierr)

contains

  real(dp) function val(i, j, k, dx)
    !! smooth in j, different in every lane i and block k
    integer, intent(in) :: i, j, k
    real(dp), intent(in) :: dx
    real(dp) :: x
    x = real(j, dp)*dx
    val = sin(2._dp*pi*x*(1._dp + 0.25_dp*real(i - 1, dp))) &
          + 0.3_dp*cos(4._dp*pi*x + 0.1_dp*real(k, dp)) + 0.05_dp*real(i, dp)*x*x
  end function val

  subroutine one_case(tag, bc, sym, dx)
    character(*), intent(in) :: tag
    integer, intent(in) :: bc
    logical, intent(in) :: sym
    real(dp), intent(in) :: dx
    real(dp), allocatable, dimension(:, :, :) :: u, du, u_s, u_e
    type(tdsops_t) :: t
    integer :: i, j, k, m
    real(dp) :: sg

    allocate (u(SZ, n, n_block), du(SZ, n, n_block), u_s(SZ, n_halo, n_block), u_e(SZ, n_halo, n_block))
    do k = 1, n_block
      do j = 1, n
        do i = 1, SZ
          u(i, j, k) = val(i, j, k, dx)
        end do
      end do
    end do
    u_s = 0._dp; u_e = 0._dp
    if (bc == BC_NEUMANN) then
      sg = merge(1._dp, -1._dp, sym)
      do m = 1, n_halo   ! u_s(:, m) = row m - 4 (ghost of row 6 - m); u_e(:, m) = row n + m (ghost of row n - m)
        u_s(:, m, :) = sg*u(:, 6 - m, :)
        u_e(:, m, :) = sg*u(:, n - m, :)
      end do
    else if (bc == BC_PERIODIC) then
      do m = 1, n_halo
        u_s(:, m, :) = u(:, n - n_halo + m, :)
        u_e(:, m, :) = u(:, m, :)
      end do
    end if
    if (bc == BC_NEUMANN) then
      t = tdsops_init(n, dx, operation='first-deriv', scheme='compact10_penta', bc_start=bc, bc_end=bc, sym=sym)
    else
      t = tdsops_init(n, dx, operation='first-deriv', scheme='compact10_penta', bc_start=bc, bc_end=bc)
    end if
    if (bc == BC_PERIODIC) then
      call exec_dist_penta_periodic(du, u, u_s, u_e, t, n_block)
    else
      call exec_dist_penta_compact(du, u, u_s, u_e, t, n_block)
    end if
    call dump_r1('penta.'//tag//'.scalars', [real(t%n_tds, dp), real(t%n_rhs, dp), t%alpha, t%beta, &
                                            t%beta_lhs_s, t%a, t%b, t%c, dx])
    call dump_r1('penta.'//tag//'.coeffs', t%coeffs)
    call dump_r2('penta.'//tag//'.coeffs_s', t%coeffs_s)
    call dump_r2('penta.'//tag//'.coeffs_e', t%coeffs_e)
    call dump_r1('penta.'//tag//'.dist_fw', t%dist_fw)
    call dump_r1('penta.'//tag//'.dist_af', t%dist_af)
    call dump_r1('penta.'//tag//'.dist_sa', t%dist_sa)
    call dump_r1('penta.'//tag//'.dist_bw', t%dist_bw)
    call dump_r3('penta.'//tag//'.u', u)
    call dump_r3('penta.'//tag//'.u_s', u_s)
    call dump_r3('penta.'//tag//'.u_e', u_e)
    call dump_r3('penta.'//tag//'.du', du)
    deallocate (u, du, u_s, u_e)
  end subroutine one_case

end program dump_penta
