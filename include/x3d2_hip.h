/*
 * x3d2_hip.h -- C ABI of the MI355X-native backend for x3d2's per-timestep
 * hot path (libx3d2_hip.so).
 *
 * This is the drop-in boundary: every entry point replaces one deferred
 * type-bound procedure of the reference's `base_backend_t`
 * (/root/reference/src/backend/backend.f90:13-62, interfaces :64-391) or one
 * hook of `poisson_fft_t` (/root/reference/src/poisson_fft.f90:45-62), or a
 * constructor the reference's concrete backends supply.  A Fortran
 * `bind(C)` interface block for these prototypes is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; every `x3d_real *` named f/u/du/... is a
 *    DEVICE pointer to one field block of x3d_block_elems() doubles
 *    (the reference's allocator block, src/allocator.f90:64-93).
 *  - return value: 0 = ok, non-zero = error (message: x3d_last_error()).
 *    The reference has no status codes -- it `error stop`s
 *    (e.g. src/backend/omp/backend.f90:349-351); a Fortran shim turns non-zero
 *    into `error stop trim(msg)`.
 *  - all calls are issued from one host thread per device and are ordered
 *    on the context's HIP stream (results of call k are visible to call k+1);
 *    only the functions that return host scalars synchronise.
 *  - DEVICE LAYOUT (backend-private, like the reference's per-backend layouts,
 *    src/ordering.f90:42-69 vs src/backend/cuda/kernels/reorder.f90:34,129):
 *    every block, whatever its DIR_X/Y/Z/C tag, is Cartesian x-fastest with a
 *    padded pitch: elem(i,j,k) = f[i + nxp*(j + nyp*k)], 0-based,
 *    nxp = round_up(nx_vert, 16), PLUS 16 where that is a multiple of 64 doubles (512^3: nxp = 528 -- keeps
 *    the rows of a y / z tile from aliasing in the memory channels), nyp = ny_vert, nzp = nz_vert.
 *    Never compute the pitch yourself: x3d_padded_dims() returns (nxp, nyp, nzp), x3d_block_elems() the
 *    block size, and x3d_get/set_field_data do the pitched copies.
 *    `dir` only selects the direction an operator works along.  A reorder is
 *    therefore a device copy and sum_{y,z}intox an axpy.
 */
#ifndef X3D2_HIP_H
#define X3D2_HIP_H

#include <stddef.h>

/* The library's real kind: the reference's `dp` (src/common.f90:6-12: double precision, or single with -DSINGLE_PREC).
 * libx3d2_hip.so is the FP64 build; libx3d2_hip_sp.so (make SP=1: every source compiled with -DX3D_SINGLE_PREC) is the
 * same code on 4-byte reals -- fields, tables, scalars and FFTs.  A caller compiles against this header with the same
 * macro as the library it links. */
#ifdef X3D_SINGLE_PREC
typedef float x3d_real;
#else
typedef double x3d_real;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* constants mirrored from src/common.f90:23-39 */
enum { X3D_DIR_X = 1, X3D_DIR_Y = 2, X3D_DIR_Z = 3, X3D_DIR_C = 4 };
enum { X3D_BC_PERIODIC = 0, X3D_BC_NEUMANN = 1, X3D_BC_DIRICHLET = 2, X3D_BC_HALO = -1 };
enum { X3D_X_FACE = 1100, X3D_Y_FACE = 1010, X3D_Z_FACE = 110 };
#define X3D_N_HALO 4 /* src/backend/backend.f90:28-29 */

typedef struct x3d_backend x3d_backend; /* omp_backend_t / cuda_backend_t analogue */
typedef struct x3d_tdsops x3d_tdsops;   /* device copy of one tdsops_t          */
typedef struct x3d_poisson x3d_poisson; /* poisson_fft_t extension              */

const char *x3d_last_error(void);
int x3d_abi_version(void);
/* sizeof(x3d_real) of THIS build: 8 (libx3d2_hip.so, the reference's default dp) or 4 (libx3d2_hip_sp.so, -DSINGLE_PREC,
 * src/common.f90:6-12) -- a binding written for one real kind checks it before the first call */
int x3d_real_bytes(void);

/* ---- construction: omp_backend_t(mesh, allocator), src/backend/omp/backend.f90:66-114 +
 *      allocator_t(dims, SZ), src/allocator.f90:64-93.
 * dims_vert: local vertex counts (mesh%get_dims(VERT)); stream: hipStream_t or NULL. */
int x3d_backend_create(x3d_backend **out, const int dims_vert[3], int device, void *stream);
int x3d_backend_destroy(x3d_backend *b);
/* a second context on the SAME device and stream as `like` (the twin backends of Poisson 100 / 110) */
int x3d_backend_create_like(x3d_backend **out, const x3d_backend *like, const int dims_vert[3]);
/* diagnostics: which = 0 -> launches of the three-components-in-one transeq kernels since creation,
 * 1 -> those of them that also applied a pending velocity correction (x3d_transeq_x_update),
 * 2 -> launches of the single-pass HALO forms of the tile kernels (decomposed directions) */
long x3d_backend_counter(const x3d_backend *b, int which);
int x3d_backend_set_stream(x3d_backend *b, void *stream);
/* round 5: CUs the persistent kernels (one workgroup per CU for a whole launch: the tile and scan kernels) leave FREE, so that
 * kernels of other streams -- RCCL's send / recv kernels of an exchange meant to run beside them -- find a CU to start on
 * before the launch ends.  0 (default): none; a multi-rank driver sets 8 (one per XCD).  The reference's GPU backend runs
 * everything on the default stream and synchronises before MPI (src/backend/cuda/sendrecv.f90:28): nothing to reserve for. */
int x3d_backend_set_comm_reserve(x3d_backend *b, int ncus);
/* round 6: direction `dir` is decomposed AND periodic over all its ranks (mesh%periodic_BC(dir), src/mesh.f90:44-53, with
 * nproc_dir(dir) > 1).  The single-pass forms of a decomposed direction (x3d_transeq_tile / x3d_tds_pair_tile with halos)
 * may then solve with the open-ended circulant recurrences instead of the reference's reduced 2 x 2 systems
 * (src/backend/omp/kernels/distributed.f90:186-206); the values exchanged per pencil and operator are then the first
 * row's solution and the forward end state instead of du_1 / X_n -- same count, same buffers.  Every rank of the direction
 * must be told the same (they are: one mesh).  Default 0. */
int x3d_backend_set_ring(x3d_backend *b, int dir, int periodic_over_all_ranks);
size_t x3d_block_elems(const x3d_backend *b);             /* allocator%ngrid */
int x3d_padded_dims(const x3d_backend *b, int dims_out[3]); /* get_padded_dims(DIR_C) */
int x3d_device_sync(x3d_backend *b);

/* block storage for callers without their own device allocator
 * (cuda_allocator_t%create_block, src/backend/cuda/allocator.f90:81-90) */
int x3d_block_alloc(x3d_backend *b, x3d_real **out);
int x3d_block_free(x3d_backend *b, x3d_real *p);
int x3d_block_fill(x3d_backend *b, x3d_real *f, x3d_real c); /* field_t%fill, src/field.f90:47-55 */
/* exchange buffers (n doubles, zeroed) and host staging for callers whose MPI is not GPU-aware (the Fortran shim on
 * several ranks: sendrecv_fields through host memory, cf. src/backend/cuda/sendrecv.f90:13-42).  The copies are
 * ordered behind the kernels queued so far and complete on return. */
int x3d_device_alloc(x3d_backend *b, x3d_real **out, long n);
int x3d_device_free(x3d_backend *b, x3d_real *p);
int x3d_copy_to_host(x3d_backend *b, x3d_real *host, const x3d_real *dev, long n);
int x3d_copy_to_device(x3d_backend *b, x3d_real *dev, const x3d_real *host, long n);
/* device-to-device exchanges between the ranks of one node without a GPU-aware MPI (what src/backend/cuda/sendrecv.f90:13-42
 * gets from one): x3d_ipc_export = hipIpcGetMemHandle of a buffer of x3d_device_alloc (64 bytes, sent to the neighbours
 * once), x3d_ipc_open maps a neighbour's buffer, x3d_copy_device copies n doubles between own and mapped memory on the
 * backend's stream (asynchronous, ordered like a kernel).  x3d_device_count: the devices this process sees -- the main
 * program picks mod(nrank, ndevs) as src/xcompact.f90:57-60 does. */
int x3d_device_count(int *n);
int x3d_ipc_export(x3d_backend *b, const x3d_real *dev, unsigned char handle[64]);
int x3d_ipc_open(x3d_backend *b, const unsigned char handle[64], x3d_real **dev);
int x3d_ipc_close(x3d_backend *b, x3d_real *dev);
int x3d_copy_device(x3d_backend *b, x3d_real *dst, const x3d_real *src, long n);

/* ---- alloc_tdsops (src/backend/backend.f90:352-372): device copy of the
 * arrays the host-side factory tdsops_init (src/tdsops.f90:63-203) produced.
 * coeffs[9]; coeffs_s/coeffs_e[4*9] with row r (0..3) = coeffs_s(:, r+1);
 * dist_* have n_rhs entries, stretch* n_tds entries. */
int x3d_tdsops_create(x3d_backend *b, x3d_tdsops **out, int n_tds, int n_rhs, int move,
                      int periodic, const x3d_real *coeffs, const x3d_real *coeffs_s,
                      const x3d_real *coeffs_e, const x3d_real *dist_fw, const x3d_real *dist_bw,
                      const x3d_real *dist_sa, const x3d_real *dist_sc, const x3d_real *dist_af,
                      const x3d_real *stretch, const x3d_real *stretch_correct);
int x3d_tdsops_destroy(x3d_tdsops *t);

/* ---- tds_solve (src/backend/backend.f90:131-150; omp: src/backend/omp/backend.f90:340-391
 * + src/backend/omp/exec_dist.f90:16-65).
 * Local form: the pencil direction is not decomposed (nproc_dir(dir)==1), the
 * periodic wrap / reduced 2x2 system is closed on the device. */
int x3d_tds_solve(x3d_backend *b, x3d_real *du, const x3d_real *u, const x3d_tdsops *t, int dir);
/* fusion extension (not in base_backend_t): accumulate != 0 gives
 * du += scale * tds_solve(u); folds sum_{y,z}intox / vecadd into the solve */
int x3d_tds_solve_acc(x3d_backend *b, x3d_real *du, const x3d_real *u, const x3d_tdsops *t, int dir,
                      int accumulate, x3d_real scale);

/* fusion extension for the operator pairs of divergence_v2c / gradient_c2v (src/vector_calculus.f90:142-332):
 *   mode 0: out1 = A(in1) + B(in2) (out2 unused)      mode 1: out1 = A(in1), out2 = B(in1) (in2 unused)
 * equal to x3d_tds_solve / x3d_tds_solve_acc issued one after the other; one kernel where the pencils allow. */
int x3d_tds_solve_pair(x3d_backend *b, int dir, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1,
                       const x3d_real *in2, const x3d_tdsops *ta, const x3d_tdsops *tb);
/* fusion extension for the 010 Poisson solve: the z pair next to the solver interleaves the y rows the way
 * enforce_periodicity_y / undo_periodicity_y do (src/backend/cuda/kernels/spectral_processing.f90:1062-1114, ny even):
 * mode 0 writes out1's y rows [0, ny) at their interleaved positions, mode 1 reads in1's rows from there.
 * *done = 0: not served for these pencils, nothing was done. */
int x3d_tds_solve_pair_yperm(x3d_backend *b, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1, const x3d_real *in2,
                             const x3d_tdsops *ta, const x3d_tdsops *tb, int ny, int *done);
/* ---- compact10_penta: 10th-order first derivative with a pentadiagonal left-hand side (src/tdsops.f90:235-251,
 * LU factors preprocess_penta_dist :971-1103; kernels der_penta_full / der_penta_periodic,
 * src/backend/omp/kernels/distributed.f90:339-691; drivers exec_dist_penta_compact / _periodic,
 * src/backend/omp/exec_dist.f90:188-241 -- local solves, no exchange).  The operator is created with
 * x3d_tdsops_create (dist_fw = 1/d, dist_af = l1, dist_sa = l2, dist_bw = u1, as the reference repurposes them) and
 * completed by x3d_tdsops_set_penta; halo_kind: 1 periodic (Sherman-Morrison-Woodbury correction), 2 BC_NEUMANN with
 * sym (even mirror ghosts), 3 BC_NEUMANN without (odd), 4 BC_DIRICHLET (one-sided closures, ghosts unused).
 * x3d_tds_penta_solve: u_s / u_e = ghost rows [4][npencil] or both NULL (formed in the kernel from halo_kind). */
int x3d_tdsops_set_penta(x3d_tdsops *t, x3d_real alpha, x3d_real beta, x3d_real beta_lhs_s, const x3d_real *dist_fw,
                         const x3d_real *dist_af, const x3d_real *dist_sa, const x3d_real *dist_bw, const x3d_real *coeffs_s,
                         const x3d_real *coeffs_e, int halo_kind);
int x3d_tds_penta_solve(x3d_backend *b, x3d_real *du, const x3d_real *u, const x3d_tdsops *t, int dir, const x3d_real *u_s,
                        const x3d_real *u_e);

/* ---- decomposed (BC_HALO) y / z directions in ONE pass + a boundary-strip correction, and plane ranges.
 * The reference's exec_dist_tds_compact / exec_dist_transeq_compact (src/backend/omp/exec_dist.f90:16-65, 67-186)
 * sweep, exchange one boundary value per pencil and operator with pprev / pnext, and sweep again.  Here the tile
 * kernels do the whole local solve in one pass with the neighbours' boundary values taken as zero and hand out
 * their own (bnd_send); after the exchange x3d_*_halo_fix adds what the received values contribute -- on the rows
 * where dist_sa / dist_sc are still above 2^-60 (the decay src/tdsops.f90:196-201 relies on for its 2 x 2
 * truncation).  Same linear system as the reference's.  Buffers (device):
 *   halo rows  [side 2][field nf][4][hr]: side 0 = rows 1..4 (sent to pprev) / rows -3..0 (received from pprev),
 *                                         side 1 = rows n-3..n (to pnext) / n+1..n+4 (from pnext)
 *              (copy_into_buffers + sendrecv_fields, src/backend/omp/backend.f90:714-737, sendrecv.f90:10-36);
 *              hr = x3d_halo_row_size(dir).  y: a row is packed [nz][nx].  z: a row is one xy plane in the BLOCK's
 *              own pitched layout, so rows 1..4 / n-3..n of a field are contiguous pieces of its block and can be
 *              sent straight out of it -- x3d_pack_halos_multi is then only needed for y
 *   boundary   [side 2][nb][np]: send side 0 = du_1 (to pprev), 1 = du_n (to pnext); recv side 0 = pprev's du_n,
 *              1 = pnext's du_1 (exec_dist.f90:52-54, 163-168)
 * x3d_transeq_tile / x3d_tds_pair_tile with halo_recv == bnd_send == NULL are the local (periodic) forms over a
 * range of planes [other0, other0 + nother) (nother < 0: all), for overlapping an exchange with the remaining
 * planes.  *done == 0: pencils not served by the tile kernels, nothing was written. */
int x3d_tdsops_halo_rows(const x3d_tdsops *t, int out[2]); /* rows 1..out[0], n-out[1]+1..n get a correction */
int x3d_tdsops_dims(const x3d_tdsops *t, int out[2]);      /* n_tds, n_rhs */
long x3d_halo_row_size(const x3d_backend *b, int dir);
int x3d_pack_halos_multi(x3d_backend *b, x3d_real *send, const x3d_real *const *fields, int nf, int n, int dir);
int x3d_transeq_tile(x3d_backend *b, int dir, x3d_real *du, x3d_real *dv, x3d_real *dw, const x3d_real *u, const x3d_real *v,
                     const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                     const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym, int accumulate, const x3d_real *halo_recv,
                     x3d_real *bnd_send, int other0, int nother, int *done);
int x3d_transeq_halo_fix(x3d_backend *b, int dir, x3d_real *du, x3d_real *dv, x3d_real *dw, const x3d_real *u, const x3d_real *v,
                         const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st, const x3d_tdsops *der2nd,
                         const x3d_real *bnd_recv);
/* modes 0, 1 of x3d_tds_solve_pair, mode 2: out1 = A(in1) */
int x3d_tds_pair_tile(x3d_backend *b, int dir, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1,
                      const x3d_real *in2, const x3d_tdsops *ta, const x3d_tdsops *tb, const x3d_real *halo_recv,
                      x3d_real *bnd_send, int other0, int nother, int *done);
int x3d_tds_pair_halo_fix(x3d_backend *b, int dir, int mode, x3d_real *out1, x3d_real *out2, const x3d_tdsops *ta,
                          const x3d_tdsops *tb, const x3d_real *bnd_recv);
/* x3d_tds_solve_pair_yperm for a decomposed z (the 010 solve on z slabs): the whole block through the halo form.
 * Mode 1's halo_recv planes are cut from the neighbours' interleaved fields, so they are read through the same
 * interleave; boundary values stay in pencil order; the strip correction of mode 0 lands on the interleaved rows. */
int x3d_tds_pair_tile_yperm(x3d_backend *b, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1, const x3d_real *in2,
                            const x3d_tdsops *ta, const x3d_tdsops *tb, const x3d_real *halo_recv, x3d_real *bnd_send,
                            int ny, int *done);
int x3d_tds_pair_halo_fix_yperm(x3d_backend *b, int mode, x3d_real *out1, x3d_real *out2, const x3d_tdsops *ta,
                                const x3d_tdsops *tb, const x3d_real *bnd_recv, int ny);
/* fusion extension: y = base + sum_i c[i]*x[i] (x3d_lincomb: the RK / AB stage) followed by du = tds_solve(y)
 * (the first x operators of divergence_v2c): one kernel for periodic 256 / 512-point x pencils, y is not read
 * back; otherwise the two calls one after the other.  y may be base. */
int x3d_tds_solve_lincomb(x3d_backend *b, int dir, x3d_real *du, const x3d_tdsops *t, x3d_real *y, const x3d_real *base,
                          int nterm, const x3d_real *c, const x3d_real *const *x);
/* fusion extension: the same with the y faces of y (vertex rows j = 0, ny-1) stamped from `wall` before the operator
 * acts = x3d_lincomb ; x3d_field_set_face_from_field(y, wall, Y_FACE) ; x3d_tds_solve (the RK stage, the channel
 * case's apply_BC, src/case/channel.f90:214-231, and the first x operator of divergence_v2c) */
int x3d_tds_solve_lincomb_wall(x3d_backend *b, int dir, x3d_real *du, const x3d_tdsops *t, x3d_real *y, const x3d_real *base,
                               int nterm, const x3d_real *c, const x3d_real *const *x, const x3d_real *wall);
/* fusion extension (round 6): x3d_tds_solve_lincomb[_wall] (wall may be NULL) followed by
 * x3d_field_mean_shift(y, dims, ncell, target, shift) -- the bulk-velocity integral the channel case's NEXT define_BC
 * asks for (src/case/channel.f90:66-72) taken while the rows of the new u are in the kernel's registers (1024-row x
 * pencils; elsewhere the two calls one after the other).  *shift: as x3d_field_mean_shift's (valid until the next
 * reduction of this backend); the partial sums are formed in another order than x3d_field_mean_shift's: round-off apart. */
int x3d_tds_solve_mean(x3d_backend *b, x3d_real *du, const x3d_real *u, const x3d_tdsops *t, int dir, const int dims[3],
                       x3d_real ncell, x3d_real target, const x3d_real **shift);  /* = x3d_tds_solve ; x3d_field_mean_shift(u) */
int x3d_tds_solve_lincomb_wall_mean(x3d_backend *b, int dir, x3d_real *du, const x3d_tdsops *t, x3d_real *y,
                                    const x3d_real *base, int nterm, const x3d_real *c, const x3d_real *const *x,
                                    const x3d_real *wall, const int dims[3], x3d_real ncell, x3d_real target,
                                    const x3d_real **shift);

/* Distributed form, one call per phase of exec_dist_tds_compact; halo and
 * boundary buffers are device arrays [rows][npencil] (npencil = x3d_npencils):
 *   x3d_pack_halos     = copy_into_buffers   (src/backend/omp/backend.f90:714-737)
 *   <caller exchanges u_send_* -> u_recv_* with pprev/pnext: sendrecv_fields>
 *   x3d_tds_dist_fwd   = der_univ_dist loop   (exec_dist.f90:36-47); fills du_send_s/e
 *   <caller exchanges du_send_* -> du_recv_*>
 *   x3d_tds_dist_bwd   = der_univ_subs loop   (exec_dist.f90:55-63) */
int x3d_npencils(const x3d_backend *b, int dir);
int x3d_pack_halos(x3d_backend *b, x3d_real *send_s, x3d_real *send_e, const x3d_real *u, int n, int dir);
int x3d_tds_dist_fwd(x3d_backend *b, x3d_real *du, x3d_real *du_send_s, x3d_real *du_send_e,
                     const x3d_real *u, const x3d_real *u_recv_s, const x3d_real *u_recv_e,
                     const x3d_tdsops *t, int dir);
int x3d_tds_dist_bwd(x3d_backend *b, x3d_real *du, const x3d_real *du_send_s, const x3d_real *du_recv_s,
                     const x3d_real *du_recv_e, const x3d_tdsops *t, int dir);
/* fusion extension: accumulate != 0 gives du += scale * result */
int x3d_tds_dist_bwd_acc(x3d_backend *b, x3d_real *du, const x3d_real *du_send_s, const x3d_real *du_recv_s,
                     const x3d_real *du_recv_e, const x3d_tdsops *t, int dir, int accumulate,
                         x3d_real scale);

/* ---- transeq_x / transeq_y / transeq_z (src/backend/backend.f90:64-92; omp:
 * src/backend/omp/backend.f90:145-184, 235-338; exec_dist.f90:67-186).
 * `dir` selects which one; u,v,w and du,dv,dw are passed exactly as the
 * caller passes them to transeq_<dir> (the permutation that makes the
 * advecting component first, :158-184, is applied inside).
 * Local form (direction not decomposed): */
int x3d_transeq(x3d_backend *b, int dir, x3d_real *du, x3d_real *dv, x3d_real *dw, const x3d_real *u,
                const x3d_real *v, const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st,
                const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                const x3d_tdsops *der2nd_sym);
/* fusion extension: accumulate != 0 gives d{u,v,w} += transeq_<dir>(u,v,w) */
int x3d_transeq_acc(x3d_backend *b, int dir, x3d_real *du, x3d_real *dv, x3d_real *dw, const x3d_real *u,
                    const x3d_real *v, const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st,
                    const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                    const x3d_tdsops *der2nd_sym, int accumulate);
/* fusion extension: transeq_x on a velocity whose pressure-gradient correction is still pending:
 * u += scale * tds_solve(gu, op_u), v += scale * tds_solve(gv, op_vw), w += scale * tds_solve(gw, op_vw) (the last x
 * operators of gradient_c2v + solver.f90:731-733) is applied per pencil inside the transeq kernel, then
 * du, dv, dw = transeq_x(u, v, w).  *done = 0: not applicable, nothing was done.  Bit-identical to
 * x3d_tds_solve_acc x 3 followed by x3d_transeq. */
int x3d_transeq_x_update(x3d_backend *b, x3d_real *du, x3d_real *dv, x3d_real *dw, x3d_real *u, x3d_real *v, x3d_real *w, x3d_real nu,
                         const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                         const x3d_tdsops *der2nd_sym, const x3d_real *gu, const x3d_real *gv, const x3d_real *gw,
                         const x3d_tdsops *op_u, const x3d_tdsops *op_vw, x3d_real scale, int *done);
/* fusion extension: transeq_x with the channel case's rotation forcing (src/case/channel.f90:191-207, there two
 * vecadd's after transeq: du = du - omega v, dv = dv + omega u) applied to the x contribution inside the kernel: the
 * y / z contributions are then accumulated onto the forced values (the same sum in another order).  *done = 0: not
 * served for these pencils, nothing was done.  u_shift != NULL: first u += *u_shift in place (the device scalar of
 * x3d_field_mean_shift: second half of the bulk-velocity correction, bit-identical to x3d_field_shift_by). */
int x3d_transeq_x_rot(x3d_backend *b, x3d_real *du, x3d_real *dv, x3d_real *dw, x3d_real *u, const x3d_real *v,
                      const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                      const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym, x3d_real omega, const x3d_real *u_shift,
                      int *done);
/* fusion extension (round 6): x3d_transeq_x_update and x3d_transeq_x_rot in ONE launch -- the pending correction
 * u, v, w += scale * tds_solve(g, op) applied per pencil, then u += *u_shift (NULL: no shift), then
 * du, dv, dw = transeq_x(u, v, w) with the rotation forcing on top (omega = 0: none).  The channel case's sub-step
 * (src/case/channel.f90:53-77, 191-207 + src/solver.f90:731-733) with the velocity correction deferred to the next
 * transeq_x: *u_shift is then x3d_field_mean_shift's scalar taken of the UNCORRECTED u (the correction is a periodic x
 * derivative: zero mean up to rounding).  Same arithmetic per point as x3d_tds_solve_acc x 3 ; x3d_field_shift_by ;
 * x3d_transeq_x_rot (u, v, w bit for bit; du, dv, dw to the last bit: another kernel body, the compiler's choice of fused
 * products).  *done = 0: not served (1024-row periodic x pencils on a uniform grid), nothing was done. */
int x3d_transeq_x_update_rot(x3d_backend *b, x3d_real *du, x3d_real *dv, x3d_real *dw, x3d_real *u, x3d_real *v, x3d_real *w,
                             x3d_real nu, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                             const x3d_tdsops *der2nd_sym, const x3d_real *gu, const x3d_real *gv, const x3d_real *gw,
                             const x3d_tdsops *op_u, const x3d_tdsops *op_vw, x3d_real scale, x3d_real omega,
                             const x3d_real *u_shift, int *done);
/* transeq_species (src/backend/backend.f90:37, omp :186-233): convection-diffusion of ONE transported
 * scalar along `dir`: dspec = [dspec +] -1/2 (uvw d(spec) + d(uvw spec)) + nu d2(spec), operators
 * (der1st, der1st_sym, der2nd); non-decomposed direction (decomposed: the dist_fwd / dist_bwd pair below
 * with u = spec, conv = uvw). */
int x3d_transeq_species(x3d_backend *b, int dir, x3d_real *dspec, const x3d_real *uvw, const x3d_real *spec, x3d_real nu,
                        const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                        int accumulate);

/* Distributed form for ONE component (transeq_dist_component, :299-338):
 * rhs = -1/2 (conv*du/dx + d(u*conv)/dx) + nu d2u/dx2.  send/recv are
 * [3][npencil] (du, dud, d2u boundary values), halos [4][npencil]. */
int x3d_transeq_dist_fwd(x3d_backend *b, int dir, x3d_real *rhs, x3d_real *send_s, x3d_real *send_e,
                         const x3d_real *u, const x3d_real *u_recv_s, const x3d_real *u_recv_e,
                         const x3d_real *conv, const x3d_real *conv_recv_s, const x3d_real *conv_recv_e,
                         const x3d_tdsops *t_du, const x3d_tdsops *t_dud, const x3d_tdsops *t_d2u);
int x3d_transeq_dist_bwd(x3d_backend *b, int dir, x3d_real *rhs, const x3d_real *send_s,
                         const x3d_real *recv_s, const x3d_real *recv_e, const x3d_real *conv, x3d_real nu,
                         const x3d_tdsops *t_du, const x3d_tdsops *t_dud, const x3d_tdsops *t_d2u);
/* fusion extension: accumulate != 0 gives rhs += result (folds the fused driver's vecadd) */
int x3d_transeq_dist_bwd_acc(x3d_backend *b, int dir, x3d_real *rhs, const x3d_real *send_s,
                         const x3d_real *recv_s, const x3d_real *recv_e, const x3d_real *conv, x3d_real nu,
                         const x3d_tdsops *t_du, const x3d_tdsops *t_dud, const x3d_tdsops *t_d2u,
                             int accumulate);

/* ---- reorder / sum_yintox / sum_zintox (src/backend/backend.f90:152-186) */
int x3d_reorder(x3d_backend *b, x3d_real *u_, const x3d_real *u, int rdr_code);
int x3d_sum_intox(x3d_backend *b, x3d_real *u, const x3d_real *u_, int dir_from);

/* ---- veccopy / vecadd / vecmult / field_scale / field_shift (:188-236, 273-291) */
int x3d_veccopy(x3d_backend *b, x3d_real *dst, const x3d_real *src);
int x3d_vecadd(x3d_backend *b, x3d_real a, const x3d_real *x, x3d_real bb, x3d_real *y);
int x3d_vecmult(x3d_backend *b, x3d_real *y, const x3d_real *x);
int x3d_field_scale(x3d_backend *b, x3d_real *f, x3d_real a);
int x3d_field_shift(x3d_backend *b, x3d_real *f, x3d_real a);
/* compute_vorticity / compute_qcriterion (src/backend/backend.f90:53-54, omp :616-649): pointwise,
 * grads = {dudx, dudy, dudz, dvdx, dvdy, dvdz, dwdx, dwdy, dwdz} device blocks, out distinct from them:
 * |curl u| and Q = -1/2 (dudx^2 + dvdy^2 + dwdz^2) - dudy dvdx - dudz dwdx - dvdz dwdy */
int x3d_compute_vorticity(x3d_backend *b, x3d_real *out, const x3d_real *const grads[9]);
int x3d_compute_qcriterion(x3d_backend *b, x3d_real *out, const x3d_real *const grads[9]);
/* fused time-integrator update (an extension, not in base_backend_t):
 * y = base + sum_i c[i]*x[i], nterm <= 5; base may be y itself.  Collapses the
 * veccopy/vecadd chains of src/time_integrator.f90:166-282 into one pass. */
int x3d_lincomb(x3d_backend *b, x3d_real *y, const x3d_real *base, int nterm, const x3d_real *c,
                const x3d_real *const *x);
/* fusion extension of the pair above for the last direction (y or z) of transeq_default when its pencils
 * run through the single-pass scan kernel (csrc/viax.hip): x3d_transeq_defer computes the three components
 * like x3d_transeq_acc(accumulate = 1) but leaves "d{u,v,w} += result" pending -- the results stay in
 * the blocks pu, pv, pw in pencil layout; *deferred = 0 when this path does not apply (nothing was done: call
 * x3d_transeq_acc).  x3d_lincomb_pending is x3d_lincomb with term x[ipend] completed on the fly,
 *   x[ipend] + pending -> d;  store != 0: x[ipend] = d;  y = base + sum_i c[i] * (i == ipend ? d : x[i]),
 * bit-identical to x3d_pending_flush (x[ipend] += pending) followed by x3d_lincomb. */
int x3d_transeq_defer(x3d_backend *b, int dir, x3d_real *pu, x3d_real *pv, x3d_real *pw, const x3d_real *u,
                      const x3d_real *v, const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st,
                      const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                      int *deferred);
int x3d_pending_flush(x3d_backend *b, int dir, x3d_real *r, const x3d_real *pend);
/* the same fusion when the last direction's pencils take the tile kernel (csrc/xscan.hip, k_ytile_transeq): the
 * component itself is computed inside the stage's linear combination.  x3d_transeq_stage_ok != 0: applicable.
 * kind 0: advecting component (der1st, der1st_sym, der2nd; conv == u), kind 1: the others (der1st_sym, der1st,
 * der2nd_sym).  Equal to x3d_transeq_species(dspec = x[ipend], accumulate = 1) followed by x3d_lincomb. */
int x3d_transeq_stage_ok(x3d_backend *b, int dir, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                         const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym);
int x3d_transeq_lincomb(x3d_backend *b, int dir, int kind, const x3d_real *u, const x3d_real *conv, x3d_real nu,
                        const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                        const x3d_tdsops *der2nd_sym, x3d_real *y, const x3d_real *base, int nterm, const x3d_real *c,
                        x3d_real *const *x, int ipend, int store);
/* round 5: transeq_<dir> (y or z, the last direction accumulated into du, dv, dw) + the stage's linear combinations of ALL
 * THREE variables in one launch of the three-components tile kernel: d_i = dvar_i + component_i; [store_i: dvar_i = d_i;]
 * y_i = base_i + sum_k c[5 i + k] (k == ipend_i ? d_i : x[5 i + k]), variable order u, v, w, x[5 i + ipend_i] == dvar_i.
 * Bit-identical to x3d_transeq_acc followed by x3d_lincomb per variable (src/solver.f90:291-389 then
 * src/time_integrator.f90:166-231).  *done = 0: not served for these pencils, nothing was done. */
int x3d_transeq_lincomb3(x3d_backend *b, int dir, x3d_real *du, x3d_real *dv, x3d_real *dw, const x3d_real *u,
                         const x3d_real *v, const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st,
                         const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                         x3d_real *const y[3], const x3d_real *const base[3], const int nterm[3], const x3d_real *c,
                         x3d_real *const *x, const int ipend[3], const int store[3], int *done);
int x3d_lincomb_pending(x3d_backend *b, int dir, x3d_real *y, const x3d_real *base, int nterm, const x3d_real *c,
                        x3d_real *const *x, int ipend, const x3d_real *pend, int store);

/* ---- reductions over the unpadded extent dims[3] of the field's data_loc.
 * Rank-local values; the caller does the cross-rank reduction (the reference
 * calls MPI_Allreduce inside: src/backend/omp/backend.f90:708, 805-808, 1063). */
int x3d_scalar_product(x3d_backend *b, const x3d_real *x, const x3d_real *y, const int dims[3],
                       x3d_real *out);
int x3d_field_max_sum(x3d_backend *b, const x3d_real *f, const int dims[3], x3d_real *max_abs,
                      x3d_real *sum_abs);
int x3d_field_volume_integral(x3d_backend *b, const x3d_real *f, const int dims[3], x3d_real *out);
/* slice_max_sum (:252-271): plane i_slice (1-based) normal to `dir` */
/* channel case without host round trips (define_BC_channel, src/case/channel.f90:59-130; one rank):
 *  x3d_field_shift_to_mean: f += target - volume_integral(f) / ncell, the sum finished on the device in the order of
 *    x3d_field_volume_integral (bit-identical to field_volume_integral + field_shift, :70-77);
 *  x3d_wall_noise: planes y = 1 and y = ny of f <- amp * (2 r - 1), r in [0, 1) from a counter-based generator
 *    (splitmix64 of seed + draw, then of that key + (face * nz + k) * nx + i; 53 bits) instead of the host's
 *    random_number planes and three full-block uploads per sub-step (:97-130) */
int x3d_field_shift_to_mean(x3d_backend *b, x3d_real *f, const int dims[3], x3d_real ncell, x3d_real target);
/* its two halves: *shift = device address of target - volume_integral(f) / ncell (valid until the backend's next
 * reduction) ; f += that device scalar (x3d_transeq_x_rot can do the second half inside its kernel) */
int x3d_field_mean_shift(x3d_backend *b, const x3d_real *f, const int dims[3], x3d_real ncell, x3d_real target,
                         const x3d_real **shift);
int x3d_field_shift_by(x3d_backend *b, x3d_real *f, const x3d_real *shift);
int x3d_wall_noise(x3d_backend *b, x3d_real *f, const int dims[3], x3d_real amp, unsigned long long seed,
                   unsigned long long draw);
int x3d_slice_max_sum(x3d_backend *b, const x3d_real *f, const int dims[3], int dir, int i_slice,
                      x3d_real *max_val, x3d_real *sum_val);

/* ---- field_set_face / field_set_face_from_field (:293-337; omp :903-1021), Y_FACE and X_FACE */
int x3d_field_set_face(x3d_backend *b, x3d_real *f, const int dims[3], x3d_real c_start, x3d_real c_end,
                       int face);
int x3d_field_set_face_from_field(x3d_backend *b, x3d_real *f, const x3d_real *f_start,
                                  const int dims[3], x3d_real c_end, int face,
                                  x3d_real flow_rate_diff);

/* ---- copy_data_to_f / copy_f_to_data via set/get_field_data
 * (src/backend/backend.f90:402-466): host Cartesian array [nz][ny][nx]
 * (x fastest, unpadded extents dims) <-> device block. */
int x3d_set_field_data(x3d_backend *b, x3d_real *f, const x3d_real *host, const int dims[3]);
int x3d_get_field_data(x3d_backend *b, x3d_real *host, const x3d_real *f, const int dims[3]);
/* same with a padded host array (leading dims hx, hy): what copy_data_to_f /
 * copy_f_to_data see, whole padded DIR_C arrays (src/backend/omp/backend.f90:1068-1082) */
int x3d_set_field_data_pitched(x3d_backend *b, x3d_real *f, const x3d_real *host, int hx, int hy,
                               const int dims[3]);
int x3d_get_field_data_pitched(x3d_backend *b, x3d_real *host, const x3d_real *f, int hx, int hy,
                               const int dims[3]);

/* ---- init_poisson_fft (src/backend/backend.f90:374-389) + poisson_fft_t hooks
 * (src/poisson_fft.f90:45-62).  Single-rank periodic (000) solver:
 * cell dims n[3]; waves_re[nz][ny][nx/2+1] (host, real part = imaginary part,
 * src/poisson_fft.f90:654-831); ax..bz host arrays of n[0], n[1], n[2]. */
int x3d_poisson_create(x3d_backend *b, x3d_poisson **out, const int n[3], const x3d_real *waves_re,
                       const x3d_real *ax, const x3d_real *bx, const x3d_real *ay, const x3d_real *by,
                       const x3d_real *az, const x3d_real *bz);
int x3d_poisson_destroy(x3d_poisson *p);
int x3d_poisson_fft_forward(x3d_poisson *p, const x3d_real *f_in);  /* fft_forward            */
int x3d_poisson_postprocess_000(x3d_poisson *p);                  /* fft_postprocess_000    */
int x3d_poisson_fft_backward(x3d_poisson *p, x3d_real *f_out);      /* fft_backward           */
int x3d_poisson_solve_000(x3d_poisson *p, x3d_real *f);             /* poisson_000, :216-226  */
/* ---- non-periodic y (010), single rank like the reference (src/poisson_fft.f90:177-180):
 * the same create call with the 010 waves and ay/by = sin/cos((i-1) pi / 2n);
 * poisson_010 = enforce_periodicity_y ; fft_forward ; fft_postprocess_010 ; fft_backward ;
 * undo_periodicity_y (src/poisson_fft.f90:228-242).  f_out != f_in (DIR_C blocks). */
int x3d_poisson_enforce_periodicity_y(x3d_poisson *p, x3d_real *f_out, const x3d_real *f_in);
int x3d_poisson_undo_periodicity_y(x3d_poisson *p, x3d_real *f_out, const x3d_real *f_in);
/* stretched y: the matrices of stretching_matrix (src/poisson_fft.f90:275-652), host arrays
 * [5][nz][n][nx/2+1] (diagonal -2..+2 slowest; real part = imaginary part).  sym != 0
 * ('centred' / 'top-bottom'): a0 = odd rows, a1 = even rows, n = ny/2; sym == 0 ('bottom'):
 * a0 = full system, n = ny, a1 ignored.  Factored once on the device (the reference
 * re-copies and re-eliminates them at every solve, src/backend/cuda/poisson_fft.f90:868-913). */
int x3d_poisson_set_stretching(x3d_poisson *p, int sym, const x3d_real *a0, const x3d_real *a1);
int x3d_poisson_postprocess_010(x3d_poisson *p);                  /* fft_postprocess_010    */
int x3d_poisson_solve_010(x3d_poisson *p, x3d_real *f, x3d_real *temp); /* poisson_010          */
/* poisson_010 (src/poisson_fft.f90:228-242) without enforce / undo_periodicity_y: the rows of f are already in
 * enforce_periodicity_y's order (the operator pair that produces the divergence writes them so) and the solution is
 * left in that order.  ny = 256 on a stretched grid: x and z transforms, then ONE pass over the spectrum for the y
 * transform, fft_postprocess_010 and the inverse y transform (csrc/y010.hip); X3D_NO_Y010=1: the 3-D transforms with
 * the post-processing kernels between them (what every other size runs). */
int x3d_poisson_solve_010_rows(x3d_poisson *p, x3d_real *f);
/* test hook: download / upload the spectral workspace [nz][ny][nx/2+1] complex */
/* Poisson 100 (x non-periodic, y and z periodic): the reference transposes x <-> y and runs the 010 solve on the
 * transposed problem (fft_forward_100 / fft_postprocess_100 / fft_backward_100,
 * src/backend/cuda/poisson_fft.f90:482-616, 781-820).  Here: a second backend of the transposed vertex dims on the
 * same stream, an x3d_poisson on it built from the swapped wave-number arrays, and this copy between the two
 * layouts: dst(y, x, z) = src(x, y, z) for x < nx, y < ny, z < nz (x2d2_amd/poisson_fft.py, HipPoissonFFT100). */
int x3d_transpose_xy(x3d_backend *b_src, x3d_backend *b_dst, x3d_real *dst, const x3d_real *src, int nx, int ny, int nz);
/* Poisson 110 (x and y non-periodic, z periodic): the reference moves z to the front (transposed copy to
 * (nz, nx, ny), R2C along z), applies enforce_periodicity_xy before and seven spectral kernels in between
 * (fft_forward_110 / fft_postprocess_110 / fft_backward_110, src/backend/cuda/poisson_fft.f90:401-480, 926-989).
 * Here: a twin backend of vertex dims (nz, nx, ny) and an x3d_poisson on it whose x is the reference's z, y its x,
 * z its y: transposed copies, the even / odd interleave along the twin's y (x3d_poisson_enforce_periodicity_y) and
 * z (…_z), and x3d_poisson_postprocess_011 = the seven kernels in the twin's layout (HipPoissonFFT110). */
int x3d_transpose_xyz_zxy(x3d_backend *b_src, x3d_backend *b_dst, x3d_real *dst, const x3d_real *src, int nx, int ny, int nz);
int x3d_transpose_zxy_xyz(x3d_backend *b_src, x3d_backend *b_dst, x3d_real *dst, const x3d_real *src, int nx, int ny, int nz);
int x3d_poisson_enforce_periodicity_z(x3d_poisson *p, x3d_real *f_out, const x3d_real *f_in);
int x3d_poisson_undo_periodicity_z(x3d_poisson *p, x3d_real *f_out, const x3d_real *f_in);
int x3d_poisson_postprocess_011(x3d_poisson *p);
int x3d_poisson_get_spectral(x3d_poisson *p, x3d_real *host_interleaved);
int x3d_poisson_set_spectral(x3d_poisson *p, const x3d_real *host_interleaved);

/* ---- distributed 000 solver for z-slab decompositions [1, 1, pz] with ny = 512 (the layout of
 * bench.py on N GPUs and of the reference's GPU backend, src/backend/cuda/poisson_fft.f90:219): one
 * all-to-all pair per solve, no pack / unpack passes (csrc/sfft.hip).  Buffers are device arrays of
 * 2 * pz * chunk doubles (x3d_sfft_sizes: chunk, zl, ys, nxs); peer r's chunk is contiguous.
 *   forward_local(f, S) | all-to-all S -> R | fft_z(R, 0) ; postprocess_000(R) ; fft_z(R, 1)
 *   | all-to-all R -> S | backward_local(S, f) */
typedef struct x3d_sfft x3d_sfft;
int x3d_sfft_create(x3d_backend *b, x3d_sfft **out, const int nglob[3], int pz, int rz);
int x3d_sfft_destroy(x3d_sfft *p);
int x3d_sfft_sizes(const x3d_sfft *p, long out[4]);
int x3d_sfft_set_waves(x3d_sfft *p, const x3d_real *waves, const x3d_real *ax, const x3d_real *bx, const x3d_real *ay,
                       const x3d_real *by, const x3d_real *az, const x3d_real *bz);
int x3d_sfft_forward_local(x3d_sfft *p, const x3d_real *f_in, x3d_real *sendbuf);
int x3d_sfft_fft_z(x3d_sfft *p, x3d_real *recvbuf, int dir);
int x3d_sfft_postprocess_000(x3d_sfft *p, x3d_real *recvbuf);
int x3d_sfft_backward_local(x3d_sfft *p, const x3d_real *recvbuf, x3d_real *f_out);
/* overlap of the all-to-all with the z stage (the reference issues cuFFTMp's slab transposes and its z
 * transforms one after the other, src/backend/cuda/poisson_fft.f90:519,568): a rank's share of ys y modes is cut
 * into `parts` pieces of ysc = ys / parts; S = [peer][part][zl][ysc][nxs], R = [part][peer][zl][ysc][nxs]; piece m
 * is sent / received as pz messages of zl * ysc * nxs complex numbers and transformed by the *_part calls while
 * the other pieces are in flight (x3d2_amd/poisson_fft.py, HipSlabPoissonFFT.poisson_000) */
int x3d_sfft_create_parts(x3d_backend *b, x3d_sfft **out, const int nglob[3], int pz, int rz, int parts);
int x3d_sfft_fft_z_part(x3d_sfft *p, x3d_real *recvbuf, int dir, int part);
int x3d_sfft_postprocess_000_part(x3d_sfft *p, x3d_real *recvbuf, int part);

/* ---- distributed 010 solver (non-periodic y: the channel case, BASELINE configs[4]) for z-slab decompositions
 * [1, 1, pz] (csrc/sfft010.hip).  The reference refuses this combination ("Multiple ranks are not yet supported for
 * non-periodic BCs!", src/poisson_fft.f90:177-180): its pencil layouts split y in spectral space, while
 * process_spectral_010 pairs the rows j and ny - j + 2 and the stretched operator is pentadiagonal along y
 * (src/backend/cuda/poisson_fft.f90:822-924).  Here the one transpose pair of a solve splits the x MODES
 * (xs = ceil((nx/2 + 1) / pz) columns per rank), y stays whole on every rank and the single-rank spectral kernels run
 * unchanged on the rank's modes.  Buffers: device arrays of 2 * pz * chunk doubles (x3d_sfft010_sizes: chunk, zl, xs,
 * i0 = first x mode of this rank, nx/2 + 1); peer r's chunk is contiguous.
 *   periodicity_y(t, f, 0) ; forward_local(t, S) | all-to-all S -> R | fft_z(R, 0) ; postprocess_010(R) ; fft_z(R, 1)
 *   | all-to-all R -> S | backward_local(S, t) ; periodicity_y(f, t, 1)
 * set_waves: this rank's block [nz][ny][xs] (pad columns one) + the global ax .. bz tables; set_stretching: this
 * rank's columns of the reference's stretching_matrix arrays, [5][nz][n][xs] (src/poisson_fft.f90:275-652). */
typedef struct x3d_sfft010 x3d_sfft010;
int x3d_sfft010_create(x3d_backend *b, x3d_sfft010 **out, const int nglob_cell[3], int pz, int rz);
int x3d_sfft010_destroy(x3d_sfft010 *p);
int x3d_sfft010_sizes(const x3d_sfft010 *p, long out[6]); /* chunk, zl, xs, i0, nx/2 + 1, parts */
/* overlap: the rank's columns travel and are solved in `parts` groups (<= 0: chosen by the library), S =
 * [peer][part][zl][ny][xsc], R = [part][peer][zl][ny][xsc]; a group holds all rows and all z of its columns, so the
 * *_part calls run on it while the next groups are in flight (x3d2_amd/poisson_fft.py, HipSlabPoissonFFT010) */
int x3d_sfft010_create_parts(x3d_backend *b, x3d_sfft010 **out, const int nglob_cell[3], int pz, int rz, int parts);
int x3d_sfft010_fft_z_part(x3d_sfft010 *p, x3d_real *recvbuf, int dir, int part);
int x3d_sfft010_postprocess_010_part(x3d_sfft010 *p, x3d_real *recvbuf, int part);
int x3d_sfft010_set_waves(x3d_sfft010 *p, const x3d_real *waves, const x3d_real *ax, const x3d_real *bx, const x3d_real *ay,
                          const x3d_real *by, const x3d_real *az, const x3d_real *bz);
int x3d_sfft010_set_stretching(x3d_sfft010 *p, int sym, const x3d_real *a0, const x3d_real *a1);
int x3d_sfft010_periodicity_y(x3d_sfft010 *p, x3d_real *f_out, const x3d_real *f_in, int undo);
int x3d_sfft010_forward_local(x3d_sfft010 *p, const x3d_real *f_in, x3d_real *sendbuf);
int x3d_sfft010_fft_z(x3d_sfft010 *p, x3d_real *recvbuf, int dir);
int x3d_sfft010_postprocess_010(x3d_sfft010 *p, x3d_real *recvbuf);
int x3d_sfft010_backward_local(x3d_sfft010 *p, const x3d_real *sendbuf, x3d_real *f_out);

/* ---- z-first form of the 000 solve (fusion extension, 512^3 cells on one rank; csrc/zfirst.hip): the transform along z
 * is done on the LDS tile of the z operator pairs that stand next to the solve in pressure_correction
 * (src/solver.f90:693-739: the last pair of divergence_v2c, the first of gradient_c2v), x and y follow on a spectrum
 * C[kz][y][x] whose half axis is z, the y transform, process_spectral_000 and the inverse y transform are one kernel:
 * 6.5 passes over the spectrum instead of 10.5; equal to x3d_poisson_solve_000 up to rounding.
 *   x3d_poisson_zfirst_ok        *ok = 1: on offer for this solver (X3D_NO_ZFIRST=1: never)
 *   x3d_tds_pair_zfirst mode 0   A(in1) + B(in2) -> spectrum (z transformed); out1, out2 unused
 *   x3d_poisson_zfirst_middle    x forward ; y forward + division + y inverse ; x inverse
 *   x3d_tds_pair_zfirst mode 1   spectrum -> out1 = A(p), out2 = B(p); in1, in2 unused.  *done = 0: nothing was done
 *   x3d_poisson_zfirst_forward / _backward: the z transform of a field in memory (stand-alone ends of the solve),
 *   x3d_poisson_solve_000_zfirst = forward ; middle ; backward, in place */
int x3d_poisson_zfirst_ok(x3d_poisson *p, int *ok);
/* ... and these two operators' z pair is one the z-transforming kernels take (probe, nothing is launched) */
int x3d_tds_pair_zfirst_ok(x3d_backend *b, const x3d_tdsops *ta, const x3d_tdsops *tb, int *ok);
int x3d_poisson_zfirst_middle(x3d_poisson *p);
int x3d_poisson_zfirst_forward(x3d_poisson *p, const x3d_real *f_in);
int x3d_poisson_zfirst_backward(x3d_poisson *p, x3d_real *f_out);
int x3d_poisson_solve_000_zfirst(x3d_poisson *p, x3d_real *f);
/* round 6: the same reordering for the channel's solve -- poisson_010 (src/poisson_fft.f90:228-242) on a stretched y with
 * 256 cell rows, nx = 1024, nz = 512: fft_postprocess_010's y stage (src/backend/cuda/poisson_fft.f90:822-924) works
 * column by column of (x mode, z mode), so the half axis may be z here too.  x3d_poisson_set_stretching_zfirst hands over
 * the pentadiagonal operators in that layout -- [5][nz/2+1][n][nx], every x mode (wave numbers mirrored above nx / 2 as
 * the reference mirrors them, :833-882), z modes 0 .. nz / 2 -- after x3d_poisson_set_stretching; x3d_poisson_zfirst_ok
 * then says 1 and x3d_tds_pair_zfirst / x3d_poisson_zfirst_middle serve the solve as above (the pairs also do the
 * solver's interleave of the y rows).  X3D_NO_ZFIRST010=1: never.  x3d_poisson_solve_010_rows_zfirst: the stand-alone
 * form on a field whose rows are interleaved already == x3d_poisson_solve_010_rows up to rounding. */
int x3d_poisson_set_stretching_zfirst(x3d_poisson *p, int sym, const x3d_real *a0, const x3d_real *a1);
/* round 6: a PROXY poisson object for a z-first solve whose middle -- everything between the two z transforms: x
 * transforms, all-to-alls between ranks, y transforms + process_spectral_000 -- the caller runs (the Fortran shim's y-slab
 * solve: fortran/m_hip_backend.f90 over csrc/sfftz.hip).  spectrum = C[257][ny][px] complex (the caller's).  The hooks
 * x3d_poisson_fft_forward / _postprocess_000 / _fft_backward and x3d_poisson_solve_000 on the proxy mean: z transform of
 * the field onto C ; middle(user) ; inverse z transform -- and under deferred execution the recorded hooks take the same
 * z-first rewrite as on one rank (the z transforms move onto the tiles of the neighbouring z operator pairs:
 * x3d_tds_pair_zfirst ; middle(user) ; x3d_tds_pair_zfirst).  middle returns 0 or an error code. */
int x3d_poisson_create_proxy(x3d_backend *b, x3d_poisson **out, x3d_real *spectrum, int ny, long px, int (*middle)(void *user),
                             void *user);
int x3d_poisson_solve_010_rows_zfirst(x3d_poisson *p, x3d_real *f);
int x3d_tds_pair_zfirst(x3d_backend *b, x3d_poisson *poisson, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1,
                        const x3d_real *in2, const x3d_tdsops *ta, const x3d_tdsops *tb, int *done);

/* ---- 000 solve on y slabs, z-first (csrc/sfftz.hip): nproc_dir = [1, py, 1], py = 1, 2, 4, 8, 512^3 cells per rank.
 * z is whole on every rank: the z operator pairs next to the solve transform along z on their tiles as on one rank
 * (x3d_sfftz_tds_pair = x3d_tds_pair_zfirst for this solver; x3d_sfftz_z: the same transform of a field in memory, for
 * the hooks).  Exchange buffers (caller's, x3d_sfftz_sizes out[3] complex elements each): [part][peer][512][kzc][xs];
 * part m = the kz planes out[4 + m] .. out[5 + m] - 1, its block starts at out[4 + m] * 512 * 512 complex elements, a
 * peer's chunk in it is 512 * kzc * xs long.  Per part: x_forward -> all-to-all -> y_stage (y forward +
 * process_spectral_000 + y inverse on the received block, in place) -> all-to-all back -> x_backward. */
typedef struct x3d_sfftz x3d_sfftz;
int x3d_sfftz_create(x3d_backend *b, x3d_sfftz **out, const int nglob_cell[3], int py, int ry, int parts);
int x3d_sfftz_destroy(x3d_sfftz *p);
int x3d_sfftz_sizes(const x3d_sfftz *p, long out[16]); /* parts, xs, xoff, buffer elements, kz0[0..parts] */
int x3d_sfftz_spectrum(const x3d_sfftz *p, x3d_real **c, int *ny, long *px); /* C[257][512][px], for x3d_poisson_create_proxy */
int x3d_sfftz_set_waves(x3d_sfftz *p, const x3d_real *rw, const x3d_real *ax, const x3d_real *bx, const x3d_real *ay,
                        const x3d_real *by, const x3d_real *az, const x3d_real *bz);
int x3d_sfftz_tds_pair(x3d_sfftz *p, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1, const x3d_real *in2,
                       const x3d_tdsops *ta, const x3d_tdsops *tb, int *done);
int x3d_sfftz_z(x3d_sfftz *p, x3d_real *f, int inverse);
/* the same as a hook of the reference under deferred execution: f is fft_forward's input (read through its handle) or
 * fft_backward's output (written whole); what the Fortran shim's y-slab Poisson solve calls (round 6) */
int x3d_sfftz_z_field(x3d_sfftz *p, x3d_real *f, int inverse);
int x3d_sfftz_x_forward(x3d_sfftz *p, x3d_real *sendbuf, int part);
int x3d_sfftz_y_stage(x3d_sfftz *p, x3d_real *recvbuf, int part, int what); /* what 0: all; 1 forward, 2 inverse, 3 division */
int x3d_sfftz_x_backward(x3d_sfftz *p, const x3d_real *buf, int part);
/* round 5: the same steps for the local y rows [y0, y0 + nyr) only -- a z pair works tile by tile and the rows' piece of
 * every (part, peer) chunk of the exchange layout is contiguous, so a group of rows can leave while the next group's z
 * pair runs, and come back while the previous group's z pair runs (poisson_fft.HipSlabPoissonFFTZ.zfirst_solve_pipelined;
 * the reference hands its transposes to cuFFTMp / 2decomp&FFT, which block: src/backend/cuda/poisson_fft.f90:218-219,
 * src/backend/omp/poisson_fft.f90:89-137) */
int x3d_sfftz_tds_pair_rows(x3d_sfftz *p, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1, const x3d_real *in2,
                            const x3d_tdsops *ta, const x3d_tdsops *tb, int y0, int nyr, int *done);
int x3d_sfftz_z_rows(x3d_sfftz *p, x3d_real *f, int inverse, int y0, int nyr);
int x3d_sfftz_x_forward_rows(x3d_sfftz *p, x3d_real *sendbuf, int part, int y0, int nyr);
int x3d_sfftz_x_backward_rows(x3d_sfftz *p, const x3d_real *buf, int part, int y0, int nyr);

/* ---- distributed form of the same solver: pencil FFT over a [1, py, pz]
 * decomposition (the 2decomp&FFT layout of the reference's CPU backend,
 * src/decomp/decomp_2decompfft.f90:42-48).  Only LOCAL stages live here; the
 * caller exchanges the packed buffers between them (peer r's chunk is
 * contiguous, sizes from x3d_pfft_sizes) -- see x3d2_amd/poisson_fft.py.
 *   fwd_x ; pack_xy | exchange(py group) | unpack_xy ; fft_y ;
 *   pack_yz | exchange(pz group) | unpack_yz ; fft_z ; postprocess_000 ;
 *   fft_z(inv) ; pack_zy | exchange | unpack_zy ; fft_y(inv) ;
 *   pack_yx | exchange | unpack_yx ; bwd_x                                   */
typedef struct x3d_pfft x3d_pfft;
int x3d_pfft_create(x3d_backend *b, x3d_pfft **out, const int nglob_cell[3], int py, int pz, int ry, int rz);
int x3d_pfft_destroy(x3d_pfft *p);
int x3d_pfft_sizes(const x3d_pfft *p, long out[8]); /* xs,xoff,ys,yoff,yl,zl,nxs,max complex count */
int x3d_pfft_set_waves(x3d_pfft *p, const x3d_real *waves_re, const x3d_real *ax, const x3d_real *bx,
                       const x3d_real *ay, const x3d_real *by, const x3d_real *az, const x3d_real *bz);
int x3d_pfft_fwd_x(x3d_pfft *p, const x3d_real *f_in);
int x3d_pfft_bwd_x(x3d_pfft *p, x3d_real *f_out);
int x3d_pfft_fft_y(x3d_pfft *p, int inverse);
int x3d_pfft_fft_z(x3d_pfft *p, int inverse);
int x3d_pfft_pack_xy(x3d_pfft *p, x3d_real *sendbuf);
int x3d_pfft_unpack_xy(x3d_pfft *p, const x3d_real *recvbuf);
int x3d_pfft_pack_yx(x3d_pfft *p, x3d_real *sendbuf);
int x3d_pfft_unpack_yx(x3d_pfft *p, const x3d_real *recvbuf);
int x3d_pfft_pack_yz(x3d_pfft *p, x3d_real *sendbuf);
int x3d_pfft_unpack_yz(x3d_pfft *p, const x3d_real *recvbuf);
int x3d_pfft_pack_zy(x3d_pfft *p, x3d_real *sendbuf);
int x3d_pfft_unpack_zy(x3d_pfft *p, const x3d_real *recvbuf);
/* a transposition along a direction that is not divided (py == 1 / pz == 1): one kernel, no exchange buffer.
 * which = 0: x-y forward, 1: y-z forward, 2: z-y backward, 3: y-x backward */
int x3d_pfft_transpose_local(x3d_pfft *p, int which);
/* the next x3d_pfft_unpack_xy / _unpack_yx takes this rank's own chunk (peer `rank` of the y group) out of `sendbuf`, the
 * buffer the matching pack filled: no copy of the own chunk into the receive buffer.  The chunk is looked up with the
 * WHOLE-solve layout (all zl planes): it serves x3d_pfft_pack_* / _unpack_* only -- the group entry points
 * (x3d_pfft_*_part) index a group's piece of the buffers and refuse a pending own chunk. */
int x3d_pfft_own_chunk(x3d_pfft *p, const x3d_real *sendbuf, int rank);
int x3d_pfft_postprocess_000(x3d_pfft *p);
/* the same solve in `parts` groups of zp = zl / parts local z planes (parts <= 0: the library's choice): everything
 * before the z transform is independent from plane to plane, so a group's x transform, xy exchange, y transform and
 * yz exchange run beside the other groups' transfers (the reference's 2decomp&FFT transposes block,
 * src/backend/omp/poisson_fft.f90:99-137).  x3d_pfft_part_layout: out = {parts, zp, complex elements of ONE group's
 * piece of the xy send / xy receive / yz send / yz receive buffer}; group m's piece starts at m times that; inside
 * it peer r's chunk starts at xoff_r*yl*zp / r*xs*yl*zp / yoff_r*xs*zp / r*ys*xs*zp.
 *   fwd_a(m): R2C x, pack -> send_xy | exchange | fwd_b(m): unpack, C2C y, pack -> send_yz | exchange |
 *   fwd_c(m): unpack ; all groups in: fft_z ; postprocess_000 ; fft_z(inv) ; bwd_c(m) | exchange | bwd_b(m) |
 *   exchange | bwd_a(m).  The whole-solve entry points above keep their buffer layout for any `parts`. */
int x3d_pfft_create_parts(x3d_backend *b, x3d_pfft **out, const int nglob_cell[3], int py, int pz, int ry, int rz,
                          int parts);
int x3d_pfft_part_layout(const x3d_pfft *p, long out[6]);
int x3d_pfft_fwd_a_part(x3d_pfft *p, const x3d_real *f_in, x3d_real *send_xy, int m);
int x3d_pfft_fwd_b_part(x3d_pfft *p, const x3d_real *recv_xy, x3d_real *send_yz, int m);
int x3d_pfft_fwd_c_part(x3d_pfft *p, const x3d_real *recv_yz, int m);
int x3d_pfft_bwd_c_part(x3d_pfft *p, x3d_real *send_zy, int m);
int x3d_pfft_bwd_b_part(x3d_pfft *p, const x3d_real *recv_zy, x3d_real *send_yx, int m);
int x3d_pfft_bwd_a_part(x3d_pfft *p, const x3d_real *recv_yx, x3d_real *f_out, int m);

/* ---- deferred execution of the reference's op-granular call sequence: fusion inside the library (csrc/lazy.hip).
 * The unchanged solver.f90 issues 16 reorder + 6 sum_*intox + ~20 veccopy / vecadd + 16 tds_solve + 3 transeq_* per
 * sub-step (src/solver.f90:291-389, 693-739; src/time_integrator.f90:166-282; src/vector_calculus.f90:142-332).  With
 * the mode on, x3d_transeq, x3d_transeq_species, x3d_tds_solve, x3d_reorder, x3d_sum_intox, x3d_veccopy, x3d_vecadd, x3d_vecmult,
 * x3d_field_scale / _shift, x3d_block_fill and the 000 hooks x3d_poisson_fft_forward / _postprocess_000 / _fft_backward
 * only RECORD their call; the queue is rewritten onto the fused kernels (x3d_transeq_acc, x3d_tds_solve_pair,
 * x3d_tds_solve_acc, x3d_lincomb, x3d_tds_solve_lincomb, x3d_poisson_solve_000: the same arithmetic in the same order)
 * and run when a result must be visible: reductions, get / set_field_data, x3d_device_sync, or any entry point that is
 * not recorded.  reorder / veccopy become aliases (copy on write): while the mode is on a block address is a HANDLE
 * whose data may live in another block's memory -- only pass handles back to this library; x3d_lazy_sync restores
 * "every block holds its own data".  x3d_block_discard = allocator%release_block (src/allocator.f90:160-168): the
 * contents are dead until the block is written again (lets the queue drop temporaries and reuse their memory).
 * Blocks of x3d_block_alloc are handles by themselves; memory allocated elsewhere: x3d_lazy_register_block.
 * Several ranks (round 4): the two-phase distributed entry points (x3d_pack_halos, x3d_tds_dist_fwd / _bwd,
 * x3d_transeq_dist_fwd / _bwd, x3d_pfft_fwd_x / _bwd_x) run at once on the buffers that hold their handles' data
 * (flush + translate, no copies), so the local directions of a decomposed run keep their rewrites.
 * x3d_lazy_stats: [0] calls recorded, [1] launches issued, [2] aliases, [3] transeq_acc, [4] pairs, [5] tds_solve_acc,
 * [6] lincombs, [7] tds_solve_lincomb, [8] solve_000, [9] updates run out of place (buffer swaps), [10] copies made for
 * an in-place update of a shared buffer, [11] copies made by x3d_lazy_sync, [12] flushes, [14] transeq_x launches that carry the velocity
 * correction of the pressure step (x3d_transeq_x_update), [13] calls dropped by the
 * rewrite, [15] extra buffers held, [16] pressure corrections through the z-first solve, [17] DECLINED: operations that ran
 * unfused although the reference's fixed sequences always offer a rewrite (a sum_{y,z}intox or a transeq_y / z launched by
 * itself) -- together with [10] the witness of a call sequence the rewrites do not recognise; a process that ends with
 * [17] + [10] > 0 says so once on stderr (results are the call-by-call ones, only slower); [18] sub-steps whose RK stage of
 * u, v, w ran inside the last transeq launch (x3d_transeq_lincomb3); [19..23] reserved (0).
 * Rewrites can be switched off one by one for A/B runs: X3D_LAZY_RULES = bit mask (bit 9 = 512: the RK stage above). */
int x3d_lazy_enable(x3d_backend *b, int on);
int x3d_lazy_flush(x3d_backend *b);
int x3d_lazy_sync(x3d_backend *b);
int x3d_lazy_unregister_block(x3d_backend *b, x3d_real *f); /* sync, then forget a block of x3d_lazy_register_block */
int x3d_lazy_register_block(x3d_backend *b, x3d_real *f);
int x3d_block_discard(x3d_backend *b, x3d_real *f);
int x3d_lazy_stats(x3d_backend *b, long out[24]);
/* round 5: a transeq of a DECOMPOSED direction recorded like a local one (so that the three sum_<d>intox behind it fold
 * into the accumulating form) and executed by the host when the queue runs: fn gets the buffers that hold the handles'
 * data and does the exchanges + x3d_transeq_tile + x3d_transeq_halo_fix itself (fortran/m_hip_backend.f90); every rank
 * runs the same queue at the same call of the program.  dir_mask: bit d = direction d (y = 2, z = 3); fn = NULL: off. */
typedef int (*x3d_dist_transeq_fn)(void *user, int dir, x3d_real *du, x3d_real *dv, x3d_real *dw, const x3d_real *u,
                                   const x3d_real *v, const x3d_real *w, x3d_real nu, const x3d_tdsops *der1st,
                                   const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                                   int accumulate);
int x3d_lazy_set_dist_transeq(x3d_backend *b, unsigned dir_mask, x3d_dist_transeq_fn fn, void *user);
/* ... and tds_solve along a decomposed direction: recorded, paired by the rewrites like local solves, executed by fn
 * (mode 2: out1 = ta(in1); 0: out1 = ta(in1) + tb(in2); 1: out1 = ta(in1), out2 = tb(in1); unused pointers NULL) */
typedef int (*x3d_dist_tds_fn)(void *user, int dir, int mode, x3d_real *out1, x3d_real *out2, const x3d_real *in1,
                               const x3d_real *in2, const x3d_tdsops *ta, const x3d_tdsops *tb);
int x3d_lazy_set_dist_tds(x3d_backend *b, unsigned dir_mask, x3d_dist_tds_fn fn, void *user);

/* ---- measurement support: HIP-event timing on the backend's stream */
int x3d_timer_start(x3d_backend *b);
int x3d_timer_stop_ms(x3d_backend *b, float *ms);
/* per-kernel-class timers (HIP events around every launch while enabled);
 * dir = 1..3 selects a pencil direction, 0 sums over all */
enum { X3D_K_TRANSEQ_FWD = 0, X3D_K_TRANSEQ_BWD = 1, X3D_K_TDS_FWD = 2, X3D_K_TDS_BWD = 3,
       X3D_K_BLAS1 = 4, X3D_K_COPY = 5, X3D_K_REDUCE = 6, X3D_K_FFT = 7, X3D_K_SPECTRAL = 8,
       X3D_K_PACK = 9 };
int x3d_prof_enable(x3d_backend *b, int on);
int x3d_prof_select(x3d_backend *b, unsigned mask); /* bit k = class X3D_K_* timed while the timers are on (default all) */
int x3d_prof_reset(x3d_backend *b);
int x3d_prof_get(x3d_backend *b, int kind, int dir, long *count, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* X3D2_HIP_H */
