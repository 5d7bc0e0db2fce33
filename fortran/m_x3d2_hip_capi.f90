!> ISO_C_BINDING interface block for libx3d2_hip.so (include/x3d2_hip.h).
!> This is the reference-side binding a maintainer adds next to
!> src/backend/cuda/ : every procedure below is what the corresponding deferred
!> procedure of base_backend_t (src/backend/backend.f90:13-62) or hook of
!> poisson_fft_t (src/poisson_fft.f90:45-62) forwards to.  Compile-checked with
!> ROCm flang (see INTEGRATION.md); field arguments are device pointers
!> (type(c_ptr), value) held by the hip_field_t extension of field_t.
module m_x3d2_hip_capi
  use iso_c_binding
  implicit none

  !> the C type behind x3d_real (include/x3d2_hip.h): real(c_double) for libx3d2_hip.so, real(c_float) for libx3d2_hip_sp.so
  !> -- compile with -DSINGLE_PREC like the reference itself (src/common.f90:6-12) and link the FP32 flavour; the backend
  !> checks x3d_real_bytes() against this kind before its first call
#ifdef SINGLE_PREC
  integer, parameter :: x3d_creal = c_float
#else
  integer, parameter :: x3d_creal = c_double
#endif

  interface
    function x3d_last_error() bind(C, name='x3d_last_error') result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function
    integer(c_int) function x3d_real_bytes() bind(C, name='x3d_real_bytes')
      !! sizeof(x3d_real) of the library that was loaded: 8 (libx3d2_hip.so) or 4 (libx3d2_hip_sp.so)
      import :: c_int
    end function
    integer(c_int) function x3d_backend_create(handle, dims_vert, device, stream) &
      bind(C, name='x3d_backend_create')
      import :: c_ptr, c_int
      type(c_ptr), intent(out) :: handle
      integer(c_int), intent(in) :: dims_vert(3)
      integer(c_int), value :: device
      type(c_ptr), value :: stream
    end function
    integer(c_int) function x3d_backend_set_ring(b, dir, periodic_over_all_ranks) bind(C, name='x3d_backend_set_ring')
      !! direction dir is decomposed and periodic over all its ranks (mesh%grid%periodic_BC): the single-pass forms of the
      !! decomposed direction may take the open-ended circulant solve -- every rank says the same
      import :: c_ptr, c_int
      type(c_ptr), value :: b
      integer(c_int), value :: dir, periodic_over_all_ranks
    end function
    integer(c_int) function x3d_backend_destroy(b) bind(C, name='x3d_backend_destroy')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
    end function
    ! a second context on the same device and stream (the twin backends of Poisson 100 / 110)
    integer(c_int) function x3d_backend_create_like(handle, like, dims_vert) bind(C, name='x3d_backend_create_like')
      import :: c_ptr, c_int
      type(c_ptr), intent(out) :: handle
      type(c_ptr), value :: like
      integer(c_int), intent(in) :: dims_vert(3)
    end function
    ! deferred execution: the op-granular calls are recorded and rewritten onto the fused kernels (csrc/lazy.hip)
    integer(c_int) function x3d_lazy_set_dist_transeq(b, dir_mask, fn, user) bind(C, name='x3d_lazy_set_dist_transeq')
      !! the transeq of a decomposed direction recorded like a local one and run by fn when the queue executes it
      import :: c_ptr, c_int, c_funptr
      type(c_ptr), value :: b, user
      integer(c_int), value :: dir_mask
      type(c_funptr), value :: fn
    end function
    integer(c_int) function x3d_pfft_own_chunk(p, sendbuf, rank) bind(C, name='x3d_pfft_own_chunk')
      !! the next unpack_xy / unpack_yx reads this rank's own chunk out of the send buffer
      import :: c_ptr, c_int
      type(c_ptr), value :: p, sendbuf
      integer(c_int), value :: rank
    end function
    integer(c_int) function x3d_pfft_transpose_local(p, which) bind(C, name='x3d_pfft_transpose_local')
      !! a transposition of the pencil solver along a direction that is not divided: one kernel, no exchange
      import :: c_ptr, c_int
      type(c_ptr), value :: p
      integer(c_int), value :: which
    end function
    integer(c_int) function x3d_lazy_set_dist_tds(b, dir_mask, fn, user) bind(C, name='x3d_lazy_set_dist_tds')
      !! tds_solve along a decomposed direction recorded like a local one and run by fn when the queue executes it
      import :: c_ptr, c_int, c_funptr
      type(c_ptr), value :: b, user
      integer(c_int), value :: dir_mask
      type(c_funptr), value :: fn
    end function
    integer(c_int) function x3d_tdsops_dims(t, out) bind(C, name='x3d_tdsops_dims')
      import :: c_ptr, c_int
      type(c_ptr), value :: t
      integer(c_int), intent(out) :: out(2)
    end function
    integer(c_int) function x3d_lazy_enable(b, on) bind(C, name='x3d_lazy_enable')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
      integer(c_int), value :: on
    end function
    integer(c_int) function x3d_lazy_sync(b) bind(C, name='x3d_lazy_sync')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
    end function
    integer(c_int) function x3d_lazy_stats(b, out) bind(C, name='x3d_lazy_stats')
      import :: c_ptr, c_int, c_long
      type(c_ptr), value :: b
      integer(c_long), intent(out) :: out(24)
    end function
    ! allocator%release_block: the block's contents are dead until it is written again
    integer(c_int) function x3d_block_discard(b, f) bind(C, name='x3d_block_discard')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, f
    end function
    integer(c_size_t) function x3d_block_elems(b) bind(C, name='x3d_block_elems')
      import :: c_ptr, c_size_t
      type(c_ptr), value :: b
    end function
    integer(c_int) function x3d_block_alloc(b, blk) bind(C, name='x3d_block_alloc')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
      type(c_ptr), intent(out) :: blk
    end function
    integer(c_int) function x3d_block_fill(b, f, c) bind(C, name='x3d_block_fill')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      real(x3d_creal), value :: c
    end function
    ! alloc_tdsops: device copy of the arrays tdsops_init produced on the host
    integer(c_int) function x3d_tdsops_create(b, t, n_tds, n_rhs, move, periodic, coeffs, &
                                              coeffs_s, coeffs_e, dist_fw, dist_bw, dist_sa, &
                                              dist_sc, dist_af, stretch, stretch_correct) &
      bind(C, name='x3d_tdsops_create')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b
      type(c_ptr), intent(out) :: t
      integer(c_int), value :: n_tds, n_rhs, move, periodic
      real(x3d_creal), intent(in) :: coeffs(*), coeffs_s(*), coeffs_e(*), dist_fw(*), dist_bw(*), &
                                    dist_sa(*), dist_sc(*), dist_af(*), stretch(*), stretch_correct(*)
    end function
    ! tds_solve
    integer(c_int) function x3d_tds_solve(b, du, u, t, dir) bind(C, name='x3d_tds_solve')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, du, u, t
      integer(c_int), value :: dir
    end function
    ! transeq_x / transeq_y / transeq_z
    integer(c_int) function x3d_transeq(b, dir, du, dv, dw, u, v, w, nu, der1st, der1st_sym, &
                                        der2nd, der2nd_sym) bind(C, name='x3d_transeq')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, du, dv, dw, u, v, w, der1st, der1st_sym, der2nd, der2nd_sym
      integer(c_int), value :: dir
      real(x3d_creal), value :: nu
    end function
    ! distributed phases of exec_dist_tds_compact / exec_dist_transeq_compact
    integer(c_int) function x3d_npencils(b, dir) bind(C, name='x3d_npencils')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
      integer(c_int), value :: dir
    end function
    integer(c_int) function x3d_pack_halos(b, send_s, send_e, u, n, dir) bind(C, name='x3d_pack_halos')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, send_s, send_e, u
      integer(c_int), value :: n, dir
    end function
    integer(c_int) function x3d_tds_dist_fwd(b, du, du_send_s, du_send_e, u, u_recv_s, u_recv_e, t, dir) &
      bind(C, name='x3d_tds_dist_fwd')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, du, du_send_s, du_send_e, u, u_recv_s, u_recv_e, t
      integer(c_int), value :: dir
    end function
    integer(c_int) function x3d_tds_dist_bwd(b, du, du_send_s, du_recv_s, du_recv_e, t, dir) &
      bind(C, name='x3d_tds_dist_bwd')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, du, du_send_s, du_recv_s, du_recv_e, t
      integer(c_int), value :: dir
    end function
    integer(c_int) function x3d_transeq_dist_fwd(b, dir, rhs, send_s, send_e, u, u_recv_s, u_recv_e, conv, &
                                                 conv_recv_s, conv_recv_e, t_du, t_dud, t_d2u) &
      bind(C, name='x3d_transeq_dist_fwd')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, rhs, send_s, send_e, u, u_recv_s, u_recv_e, conv, conv_recv_s, conv_recv_e, &
                            t_du, t_dud, t_d2u
      integer(c_int), value :: dir
    end function
    integer(c_int) function x3d_transeq_dist_bwd(b, dir, rhs, send_s, recv_s, recv_e, conv, nu, t_du, t_dud, t_d2u) &
      bind(C, name='x3d_transeq_dist_bwd')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, rhs, send_s, recv_s, recv_e, conv, t_du, t_dud, t_d2u
      integer(c_int), value :: dir
      real(x3d_creal), value :: nu
    end function
    ! Poisson 100: copy between the block layouts of a backend and of its x <-> y transposed twin
    integer(c_int) function x3d_transpose_xy(b_src, b_dst, dst, src, nx, ny, nz) bind(C, name='x3d_transpose_xy')
      import :: c_ptr, c_int
      type(c_ptr), value :: b_src, b_dst, dst, src
      integer(c_int), value :: nx, ny, nz
    end function
    ! Poisson 110: z-first twin problem
    integer(c_int) function x3d_transpose_xyz_zxy(b_src, b_dst, dst, src, nx, ny, nz) &
      bind(C, name='x3d_transpose_xyz_zxy')
      import :: c_ptr, c_int
      type(c_ptr), value :: b_src, b_dst, dst, src
      integer(c_int), value :: nx, ny, nz
    end function
    integer(c_int) function x3d_transpose_zxy_xyz(b_src, b_dst, dst, src, nx, ny, nz) &
      bind(C, name='x3d_transpose_zxy_xyz')
      import :: c_ptr, c_int
      type(c_ptr), value :: b_src, b_dst, dst, src
      integer(c_int), value :: nx, ny, nz
    end function
    integer(c_int) function x3d_poisson_enforce_periodicity_z(p, f_out, f_in) &
      bind(C, name='x3d_poisson_enforce_periodicity_z')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_out, f_in
    end function
    integer(c_int) function x3d_poisson_undo_periodicity_z(p, f_out, f_in) &
      bind(C, name='x3d_poisson_undo_periodicity_z')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_out, f_in
    end function
    integer(c_int) function x3d_poisson_postprocess_011(p) bind(C, name='x3d_poisson_postprocess_011')
      import :: c_ptr, c_int
      type(c_ptr), value :: p
    end function
    ! exchange buffers and host staging (an MPI that is not GPU-aware)
    integer(c_int) function x3d_device_alloc(b, p, n) bind(C, name='x3d_device_alloc')
      import :: c_ptr, c_int, c_long
      type(c_ptr), value :: b
      type(c_ptr), intent(out) :: p
      integer(c_long), value :: n
    end function
    ! decomposed y / z directions in ONE pass (include/x3d2_hip.h, "decomposed (BC_HALO) y / z directions in ONE pass"):
    ! exec_dist_tds_compact / exec_dist_transeq_compact (src/backend/omp/exec_dist.f90:16-186) as pack -> exchange ->
    ! single-pass tile kernel (own boundary values out) -> exchange -> boundary-strip correction
    integer(c_long) function x3d_halo_row_size(b, dir) bind(C, name='x3d_halo_row_size')
      import :: c_ptr, c_int, c_long
      type(c_ptr), value :: b
      integer(c_int), value :: dir
    end function
    integer(c_int) function x3d_pack_halos_multi(b, send, fields, nf, n, dir) bind(C, name='x3d_pack_halos_multi')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, send
      type(c_ptr), intent(in) :: fields(*)
      integer(c_int), value :: nf, n, dir
    end function
    integer(c_int) function x3d_transeq_tile(b, dir, du, dv, dw, u, v, w, nu, der1st, der1st_sym, der2nd, der2nd_sym, &
                                             accumulate, halo_recv, bnd_send, other0, nother, done) &
      bind(C, name='x3d_transeq_tile')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, du, dv, dw, u, v, w, der1st, der1st_sym, der2nd, der2nd_sym, halo_recv, bnd_send
      integer(c_int), value :: dir, accumulate, other0, nother
      real(x3d_creal), value :: nu
      integer(c_int), intent(out) :: done
    end function
    integer(c_int) function x3d_transeq_halo_fix(b, dir, du, dv, dw, u, v, w, nu, der1st, der2nd, bnd_recv) &
      bind(C, name='x3d_transeq_halo_fix')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, du, dv, dw, u, v, w, der1st, der2nd, bnd_recv
      integer(c_int), value :: dir
      real(x3d_creal), value :: nu
    end function
    integer(c_int) function x3d_tds_pair_tile(b, dir, mode, out1, out2, in1, in2, ta, tb, halo_recv, bnd_send, other0, &
                                              nother, done) bind(C, name='x3d_tds_pair_tile')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, out1, out2, in1, in2, ta, tb, halo_recv, bnd_send
      integer(c_int), value :: dir, mode, other0, nother
      integer(c_int), intent(out) :: done
    end function
    integer(c_int) function x3d_tds_pair_halo_fix(b, dir, mode, out1, out2, ta, tb, bnd_recv) &
      bind(C, name='x3d_tds_pair_halo_fix')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, out1, out2, ta, tb, bnd_recv
      integer(c_int), value :: dir, mode
    end function
    ! device-to-device exchanges between ranks of a node (HIP inter-process handles; cf. src/backend/cuda/sendrecv.f90:13-42)
    integer(c_int) function x3d_device_count(n) bind(C, name='x3d_device_count')
      import :: c_int
      integer(c_int), intent(out) :: n
    end function
    integer(c_int) function x3d_ipc_export(b, dev, handle) bind(C, name='x3d_ipc_export')
      import :: c_ptr, c_int, c_signed_char
      type(c_ptr), value :: b, dev
      integer(c_signed_char), intent(out) :: handle(64)
    end function
    integer(c_int) function x3d_ipc_open(b, handle, dev) bind(C, name='x3d_ipc_open')
      import :: c_ptr, c_int, c_signed_char
      type(c_ptr), value :: b
      integer(c_signed_char), intent(in) :: handle(64)
      type(c_ptr), intent(out) :: dev
    end function
    integer(c_int) function x3d_ipc_close(b, dev) bind(C, name='x3d_ipc_close')
      !! unmap a peer's buffer (x3d_backend_destroy unmaps whatever is still mapped)
      import :: c_ptr, c_int
      type(c_ptr), value :: b, dev
    end function
    integer(c_int) function x3d_copy_device(b, dst, src, n) bind(C, name='x3d_copy_device')
      import :: c_ptr, c_int, c_long
      type(c_ptr), value :: b, dst, src
      integer(c_long), value :: n
    end function
    integer(c_int) function x3d_copy_to_host(b, host, dev, n) bind(C, name='x3d_copy_to_host')
      import :: c_ptr, c_int, c_long, x3d_creal
      type(c_ptr), value :: b, dev
      real(x3d_creal), intent(out) :: host(*)
      integer(c_long), value :: n
    end function
    integer(c_int) function x3d_copy_to_device(b, dev, host, n) bind(C, name='x3d_copy_to_device')
      import :: c_ptr, c_int, c_long, x3d_creal
      type(c_ptr), value :: b, dev
      real(x3d_creal), intent(in) :: host(*)
      integer(c_long), value :: n
    end function
    ! pencil-decomposed 000 Poisson solver: local stages (csrc/pfft.hip); the caller exchanges the packed buffers
    ! ---- 000 solve on y slabs [1, py, 1] of 512^3 cells per rank, z-first (csrc/sfftz.hip): the stand-alone hook forms
    integer(c_int) function x3d_sfftz_create(b, handle, nglob, py, ry, parts) bind(C, name='x3d_sfftz_create')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
      type(c_ptr), intent(out) :: handle
      integer(c_int), intent(in) :: nglob(3)
      integer(c_int), value :: py, ry, parts
    end function
    integer(c_int) function x3d_sfftz_destroy(p) bind(C, name='x3d_sfftz_destroy')
      import :: c_ptr, c_int
      type(c_ptr), value :: p
    end function
    integer(c_int) function x3d_sfftz_sizes(p, sz) bind(C, name='x3d_sfftz_sizes')
      import :: c_ptr, c_int, c_long
      type(c_ptr), value :: p
      integer(c_long), intent(out) :: sz(16)
    end function
    integer(c_int) function x3d_sfftz_set_waves(p, rw, ax, bx, ay, by, az, bz) bind(C, name='x3d_sfftz_set_waves')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: p
      real(x3d_creal), intent(in) :: rw(*), ax(*), bx(*), ay(*), by(*), az(*), bz(*)
    end function
    integer(c_int) function x3d_sfftz_spectrum(p, c, ny, px) bind(C, name='x3d_sfftz_spectrum')
      import :: c_ptr, c_int, c_long
      type(c_ptr), value :: p
      type(c_ptr), intent(out) :: c
      integer(c_int), intent(out) :: ny
      integer(c_long), intent(out) :: px
    end function
    integer(c_int) function x3d_poisson_create_proxy(b, handle, spectrum, ny, px, middle, user) &
      bind(C, name='x3d_poisson_create_proxy')
      !! a poisson object whose hooks are: z transform ; middle(user) ; inverse z transform (include/x3d2_hip.h)
      import :: c_ptr, c_int, c_long, c_funptr
      type(c_ptr), value :: b, spectrum, user
      type(c_ptr), intent(out) :: handle
      integer(c_int), value :: ny
      integer(c_long), value :: px
      type(c_funptr), value :: middle
    end function
    integer(c_int) function x3d_sfftz_z_field(p, f, inverse) bind(C, name='x3d_sfftz_z_field')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f
      integer(c_int), value :: inverse
    end function
    integer(c_int) function x3d_sfftz_x_forward(p, sendbuf, part) bind(C, name='x3d_sfftz_x_forward')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, sendbuf
      integer(c_int), value :: part
    end function
    integer(c_int) function x3d_sfftz_y_stage(p, recvbuf, part, what) bind(C, name='x3d_sfftz_y_stage')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, recvbuf
      integer(c_int), value :: part, what
    end function
    integer(c_int) function x3d_sfftz_x_backward(p, buf, part) bind(C, name='x3d_sfftz_x_backward')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
      integer(c_int), value :: part
    end function
    integer(c_int) function x3d_pfft_create(b, p, nglob_cell, py, pz, ry, rz) bind(C, name='x3d_pfft_create')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
      type(c_ptr), intent(out) :: p
      integer(c_int), intent(in) :: nglob_cell(3)
      integer(c_int), value :: py, pz, ry, rz
    end function
    integer(c_int) function x3d_pfft_sizes(p, sizes) bind(C, name='x3d_pfft_sizes')
      import :: c_ptr, c_int, c_long
      type(c_ptr), value :: p
      integer(c_long), intent(out) :: sizes(8)
    end function
    integer(c_int) function x3d_pfft_set_waves(p, waves_re, ax, bx, ay, by, az, bz) bind(C, name='x3d_pfft_set_waves')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: p
      real(x3d_creal), intent(in) :: waves_re(*), ax(*), bx(*), ay(*), by(*), az(*), bz(*)
    end function
    integer(c_int) function x3d_pfft_fwd_x(p, f_in) bind(C, name='x3d_pfft_fwd_x')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_in
    end function
    integer(c_int) function x3d_pfft_bwd_x(p, f_out) bind(C, name='x3d_pfft_bwd_x')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_out
    end function
    integer(c_int) function x3d_pfft_fft_y(p, inverse) bind(C, name='x3d_pfft_fft_y')
      import :: c_ptr, c_int
      type(c_ptr), value :: p
      integer(c_int), value :: inverse
    end function
    integer(c_int) function x3d_pfft_fft_z(p, inverse) bind(C, name='x3d_pfft_fft_z')
      import :: c_ptr, c_int
      type(c_ptr), value :: p
      integer(c_int), value :: inverse
    end function
    integer(c_int) function x3d_pfft_pack_xy(p, buf) bind(C, name='x3d_pfft_pack_xy')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_unpack_xy(p, buf) bind(C, name='x3d_pfft_unpack_xy')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_pack_yx(p, buf) bind(C, name='x3d_pfft_pack_yx')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_unpack_yx(p, buf) bind(C, name='x3d_pfft_unpack_yx')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_pack_yz(p, buf) bind(C, name='x3d_pfft_pack_yz')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_unpack_yz(p, buf) bind(C, name='x3d_pfft_unpack_yz')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_pack_zy(p, buf) bind(C, name='x3d_pfft_pack_zy')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_unpack_zy(p, buf) bind(C, name='x3d_pfft_unpack_zy')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, buf
    end function
    integer(c_int) function x3d_pfft_postprocess_000(p) bind(C, name='x3d_pfft_postprocess_000')
      import :: c_ptr, c_int
      type(c_ptr), value :: p
    end function
    ! transeq_species
    integer(c_int) function x3d_transeq_species(b, dir, dspec, uvw, spec, nu, der1st, der1st_sym, der2nd, &
                                                accumulate) bind(C, name='x3d_transeq_species')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, dspec, uvw, spec, der1st, der1st_sym, der2nd
      integer(c_int), value :: dir, accumulate
      real(x3d_creal), value :: nu
    end function
    ! compute_vorticity / compute_qcriterion: grads = 9 device blocks
    integer(c_int) function x3d_compute_vorticity(b, out, grads) bind(C, name='x3d_compute_vorticity')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, out
      type(c_ptr), intent(in) :: grads(9)
    end function
    integer(c_int) function x3d_compute_qcriterion(b, out, grads) bind(C, name='x3d_compute_qcriterion')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, out
      type(c_ptr), intent(in) :: grads(9)
    end function
    ! reorder / sum_yintox / sum_zintox
    integer(c_int) function x3d_reorder(b, u_, u, rdr) bind(C, name='x3d_reorder')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, u_, u
      integer(c_int), value :: rdr
    end function
    integer(c_int) function x3d_sum_intox(b, u, u_, dir_from) bind(C, name='x3d_sum_intox')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, u, u_
      integer(c_int), value :: dir_from
    end function
    ! veccopy / vecadd / vecmult / field_scale / field_shift
    integer(c_int) function x3d_veccopy(b, dst, src) bind(C, name='x3d_veccopy')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, dst, src
    end function
    integer(c_int) function x3d_vecadd(b, a, x, bb, y) bind(C, name='x3d_vecadd')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, x, y
      real(x3d_creal), value :: a, bb
    end function
    integer(c_int) function x3d_vecmult(b, y, x) bind(C, name='x3d_vecmult')
      import :: c_ptr, c_int
      type(c_ptr), value :: b, y, x
    end function
    integer(c_int) function x3d_field_scale(b, f, a) bind(C, name='x3d_field_scale')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      real(x3d_creal), value :: a
    end function
    integer(c_int) function x3d_field_shift(b, f, a) bind(C, name='x3d_field_shift')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      real(x3d_creal), value :: a
    end function
    ! reductions (rank-local; the shim adds MPI_Allreduce like the reference)
    integer(c_int) function x3d_scalar_product(b, x, y, dims, s) bind(C, name='x3d_scalar_product')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, x, y
      integer(c_int), intent(in) :: dims(3)
      real(x3d_creal), intent(out) :: s
    end function
    integer(c_int) function x3d_field_max_sum(b, f, dims, max_abs, sum_abs) bind(C, name='x3d_field_max_sum')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      integer(c_int), intent(in) :: dims(3)
      real(x3d_creal), intent(out) :: max_abs, sum_abs
    end function
    integer(c_int) function x3d_field_volume_integral(b, f, dims, s) bind(C, name='x3d_field_volume_integral')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      integer(c_int), intent(in) :: dims(3)
      real(x3d_creal), intent(out) :: s
    end function
    integer(c_int) function x3d_slice_max_sum(b, f, dims, dir, i_slice, max_val, sum_val) &
      bind(C, name='x3d_slice_max_sum')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      integer(c_int), intent(in) :: dims(3)
      integer(c_int), value :: dir, i_slice
      real(x3d_creal), intent(out) :: max_val, sum_val
    end function
    ! faces
    integer(c_int) function x3d_field_set_face(b, f, dims, c_start, c_end, face) bind(C, name='x3d_field_set_face')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      integer(c_int), intent(in) :: dims(3)
      real(x3d_creal), value :: c_start, c_end
      integer(c_int), value :: face
    end function
    integer(c_int) function x3d_field_set_face_from_field(b, f, f_start, dims, c_end, face, flow_rate_diff) &
      bind(C, name='x3d_field_set_face_from_field')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f, f_start
      integer(c_int), intent(in) :: dims(3)
      real(x3d_creal), value :: c_end, flow_rate_diff
      integer(c_int), value :: face
    end function
    ! copy_data_to_f / copy_f_to_data (Cartesian host arrays)
    integer(c_int) function x3d_set_field_data(b, f, host, dims) bind(C, name='x3d_set_field_data')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      real(x3d_creal), intent(in) :: host(*)
      integer(c_int), intent(in) :: dims(3)
    end function
    integer(c_int) function x3d_get_field_data(b, host, f, dims) bind(C, name='x3d_get_field_data')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      real(x3d_creal), intent(out) :: host(*)
      integer(c_int), intent(in) :: dims(3)
    end function
    integer(c_int) function x3d_set_field_data_pitched(b, f, host, hx, hy, dims) &
      bind(C, name='x3d_set_field_data_pitched')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      real(x3d_creal), intent(in) :: host(*)
      integer(c_int), value :: hx, hy
      integer(c_int), intent(in) :: dims(3)
    end function
    integer(c_int) function x3d_get_field_data_pitched(b, host, f, hx, hy, dims) &
      bind(C, name='x3d_get_field_data_pitched')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b, f
      real(x3d_creal), intent(out) :: host(*)
      integer(c_int), value :: hx, hy
      integer(c_int), intent(in) :: dims(3)
    end function
    integer(c_int) function x3d_padded_dims(b, dims) bind(C, name='x3d_padded_dims')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
      integer(c_int), intent(out) :: dims(3)
    end function
    integer(c_int) function x3d_device_sync(b) bind(C, name='x3d_device_sync')
      import :: c_ptr, c_int
      type(c_ptr), value :: b
    end function
    ! init_poisson_fft + poisson_fft_t hooks
    integer(c_int) function x3d_poisson_create(b, p, n, waves_re, ax, bx, ay, by, az, bz) &
      bind(C, name='x3d_poisson_create')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: b
      type(c_ptr), intent(out) :: p
      integer(c_int), intent(in) :: n(3)
      real(x3d_creal), intent(in) :: waves_re(*), ax(*), bx(*), ay(*), by(*), az(*), bz(*)
    end function
    integer(c_int) function x3d_poisson_fft_forward(p, f_in) bind(C, name='x3d_poisson_fft_forward')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_in
    end function
    integer(c_int) function x3d_poisson_postprocess_000(p) bind(C, name='x3d_poisson_postprocess_000')
      import :: c_ptr, c_int
      type(c_ptr), value :: p
    end function
    integer(c_int) function x3d_poisson_fft_backward(p, f_out) bind(C, name='x3d_poisson_fft_backward')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_out
    end function
    ! non-periodic y (010)
    integer(c_int) function x3d_poisson_enforce_periodicity_y(p, f_out, f_in) &
      bind(C, name='x3d_poisson_enforce_periodicity_y')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_out, f_in
    end function
    integer(c_int) function x3d_poisson_undo_periodicity_y(p, f_out, f_in) &
      bind(C, name='x3d_poisson_undo_periodicity_y')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f_out, f_in
    end function
    integer(c_int) function x3d_poisson_set_stretching(p, sym, a0, a1) &
      bind(C, name='x3d_poisson_set_stretching')
      import :: c_ptr, c_int, x3d_creal
      type(c_ptr), value :: p
      integer(c_int), value :: sym
      real(x3d_creal), intent(in) :: a0(*), a1(*)
    end function
    integer(c_int) function x3d_poisson_postprocess_010(p) bind(C, name='x3d_poisson_postprocess_010')
      import :: c_ptr, c_int
      type(c_ptr), value :: p
    end function
    !> fft_forward_010 ; fft_postprocess_010 ; fft_backward_010 in one call (rows of f in enforce_periodicity_y's order)
    integer(c_int) function x3d_poisson_solve_010_rows(p, f) bind(C, name='x3d_poisson_solve_010_rows')
      import :: c_ptr, c_int
      type(c_ptr), value :: p, f
    end function
  end interface

contains

  !> the reference has no status codes: turn a non-zero return into `error stop`
  subroutine x3d_check(rc)
    integer(c_int), intent(in) :: rc
    character(kind=c_char), pointer :: msg(:)
    character(len=512) :: text
    integer :: i
    if (rc == 0) return
    call c_f_pointer(x3d_last_error(), msg, [512])
    text = ''
    do i = 1, 512
      if (msg(i) == c_null_char) exit
      text(i:i) = msg(i)
    end do
    print *, trim(text)
    error stop 'libx3d2_hip'
  end subroutine x3d_check

end module m_x3d2_hip_capi
