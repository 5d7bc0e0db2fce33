// Distributed spectral Poisson solver for NON-PERIODIC y (010: the channel case) on z-slab decompositions [1, 1, pz]
// -- BASELINE configs[4] on several GPUs.  The reference has no such solver: poisson_fft_t%base_init stops with
// "Multiple ranks are not yet supported for non-periodic BCs!" (/root/reference/src/poisson_fft.f90:177-180), because
// its 2decomp / cuFFTMp pencil layouts split y in spectral space while process_spectral_010 couples the rows j and
// ny - j + 2 and the stretched-mesh operator is a pentadiagonal system along y
// (src/backend/cuda/poisson_fft.f90:822-924, kernels/spectral_processing.f90:385-702).
//
// Here y never leaves the rank: the ONE transpose pair of a solve splits the x MODES.
//   f[zl][ny][nx]  (rows already in enforce_periodicity_y's order)
//   --rocFFT 2-D, R2C along x, C2C along y, batched over the zl local planes-->  C0[zl][ny][nxs]   nxs = pz * xs
//   --pack-->  S[peer][zl][ny][xs]          (peer r gets the x modes r xs .. (r + 1) xs - 1 of every local row)
//   --all-to-all among the pz ranks-->  R[peer][zl][ny][xs] = W[nz][ny][xs]   (chunks arrive in z order)
//   --32 x 32 LDS-tiled transpose-->  T[ny][xs][nz] (z contiguous)  --rocFFT C2C along z-->  this rank's xs modes x ALL
//     ny rows x ALL nz modes   (rocFFT's strided plan on W itself takes 2.0 - 2.1 ms per direction for 1.08 GB,
//     transpose + contiguous plan 0.42 + 0.41, scratch/zfft_bench.hip -- and the spectral stage runs on T as it is)
//   --fft_postprocess_010 on T: the single-rank kernels (spectral010.h) in their z-fastest form, mode offset
//     i0 = rz xs: uniform y one kernel, stretched y fw ; pentadiagonal solves (factored once at set-up) ; bw
//   --inverse z transform, transpose back, all-to-all back, unpack, inverse 2-D transform-->  f
// xs = parts * xsc >= ceil((nx/2 + 1) / pz): every rank owns the same number of mode columns, the columns beyond
// nx/2 + 1 on the last rank are padding (zeros in, wave numbers one, matrices zero: the kernels' guarded divisions
// leave zeros).
// Overlap: a rank's columns travel in `parts` groups of xsc columns, S = [peer][part][zl][ny][xsc] and
// R = [part][peer][zl][ny][xsc] = [part][nz][ny][xsc]: every group holds ALL rows and ALL z of its columns, so its z
// transforms, the paired split and the pentadiagonal solves run as soon as it has arrived, beside the transfer of the
// next groups (the *_part entry points; x3d2_amd/poisson_fft.py, HipSlabPoissonFFT010.poisson_010).  T, waves and the
// factors are stored per group: T[part][ny][xsc][nz], lu[part][5][n][xsc][nz].
#include <hipfft/hipfft.h>

#include "spectral010.h"

#define X3D_FFT(expr)                                                                          \
    do {                                                                                       \
        hipfftResult r_ = (expr);                                                              \
        if (r_ != HIPFFT_SUCCESS) {                                                            \
            x3d_set_error("%s failed: hipfft error %d (%s:%d)", #expr, (int)r_, __FILE__,      \
                          __LINE__);                                                           \
            return 3;                                                                          \
        }                                                                                      \
    } while (0)

struct x3d_sfft010 {
    x3d_backend *b;
    int nx, ny, nz;      // global cell dims
    int nxm;             // nx/2 + 1 modes along x
    int pz, rz, zl;      // ranks along z, this rank, local planes
    int xs, nxs;         // mode columns per rank; nxs = pz * xs = row pitch of the local 2-D spectrum
    int parts, xsc;      // the columns travel and are solved in `parts` groups of xsc (xs = parts * xsc)
    hipfftHandle plan_xy_fw, plan_xy_bw, plan_z;
    int split_xy;        // rocFFT refused the 2-D real plan for these lengths: 1-D x plan (batched over all local rows)
                         // + 1-D strided y plan run once per local plane
    hipfftHandle plan_x_fw, plan_x_bw, plan_y;
    real2_t *c0;         // [zl][ny][nxs]
    real2_t *t;          // [part][ny][xsc][nz]: z-contiguous copy of the received array; the spectral stage works here
    real_t *waves;       // [part][ny][xsc][nz] (pads: one)
    real_t *ab;          // ax bx (padded to max(nx, nxs)) ay by az bz
    int nab_x;           // length of the ax / bx tables on the device
    int stretched, sym;
    real_t *lu[2];       // factored pentadiagonal operators [part][5][n][xsc][nz]
    void *work;
};

// C0[zl][ny][nxs] -> S[peer][part][zl][ny][xsc] (UNPACK: the other way); one thread per complex number of C0
template <bool UNPACK>
__global__ void __launch_bounds__(256)
    k_sfft010_pack(real2_t *__restrict__ s, real2_t *__restrict__ c0, long rows, int xsc, int parts, int pz)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int xs = xsc * parts, nxs = xs * pz;
    const long row = t / nxs;
    if (row >= rows) return;
    const int col = (int)(t % nxs), peer = col / xs, part = (col % xs) / xsc, i = col % xsc;
    const long si = (((long)peer * parts + part) * rows + row) * xsc + i;
    if (UNPACK) c0[t] = s[si];
    else s[si] = c0[t];
}

// set-up: src [nz][rows][xs] (x fastest, as the host builds it) -> dst [part][rows][xsc][nz] with a stride between parts
__global__ void __launch_bounds__(256)
    k_sfft010_gather(real_t *__restrict__ dst, const real_t *__restrict__ src, int nz, int rows, int xsc, int parts,
                     long part_stride)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int xs = xsc * parts;
    const long n = (long)nz * rows * xs;
    if (t >= n) return;
    const int k = (int)(t % nz);
    const long c = t / nz;                 // (part, row, i) flattened, k fastest on the destination side
    const int i = (int)(c % xsc), j = (int)((c / xsc) % rows), m = (int)(c / ((long)xsc * rows));
    dst[m * part_stride + ((long)j * xsc + i) * nz + k] = src[((long)k * rows + j) * xs + m * xsc + i];
}

// 32 x 32 tiles through LDS: src [nB][nA] (A contiguous) -> dst [nA][nB] (B contiguous)
template <class E>
__global__ void __launch_bounds__(256) k_sfft010_transpose(E *__restrict__ dst, const E *__restrict__ src, long nA, long nB)
{
    __shared__ E tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long a0 = (long)blockIdx.x * 32, b0 = (long)blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const long bb = b0 + ty + 8 * r, aa = a0 + tx;
        if (aa < nA && bb < nB) tile[ty + 8 * r][tx] = src[bb * nA + aa];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const long aa = a0 + ty + 8 * r, bb = b0 + tx;
        if (aa < nA && bb < nB) dst[aa * nB + bb] = tile[tx][ty + 8 * r];
    }
}
template <class E>
static int transpose_launch(hipStream_t st, E *dst, const E *src, long nA, long nB)
{
    hipLaunchKernelGGL(k_sfft010_transpose<E>, dim3((unsigned)((nA + 31) / 32), (unsigned)((nB + 31) / 32)), dim3(256), 0, st,
                       dst, src, nA, nB);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_sfft010_create_parts(x3d_backend *b, x3d_sfft010 **out, const int nglob[3], int pz, int rz, int parts);
extern "C" int x3d_sfft010_create(x3d_backend *b, x3d_sfft010 **out, const int nglob[3], int pz, int rz)
{
    X3D_RANGE(__func__);
    return x3d_sfft010_create_parts(b, out, nglob, pz, rz, 1);
}

// parts <= 0: chosen here -- the count among 4, 5, 3 that pads the rank's columns least (8 ranks, 513 modes: 65 = 5 x 13)
extern "C" int x3d_sfft010_create_parts(x3d_backend *b, x3d_sfft010 **out, const int nglob[3], int pz, int rz, int parts)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && nglob, "x3d_sfft010_create: null argument");
    X3D_REQUIRE(pz >= 1 && rz >= 0 && rz < pz, "x3d_sfft010_create: bad rank grid");
    X3D_REQUIRE(nglob[2] % pz == 0, "x3d_sfft010_create: nz = %d does not divide by pz = %d", nglob[2], pz);
    x3d_sfft010 *p = new x3d_sfft010();
    memset(p, 0, sizeof *p);
    p->b = b;
    p->nx = nglob[0]; p->ny = nglob[1]; p->nz = nglob[2];
    p->nxm = p->nx / 2 + 1;
    p->pz = pz; p->rz = rz; p->zl = p->nz / pz;
    {
        const int base = (p->nxm + pz - 1) / pz;
        if (parts <= 0) {
            const int cand[3] = {4, 5, 3};
            int best = 1, pad = 1 << 30;
            for (int c : cand) {
                if (c > base) continue;
                const int w = (base + c - 1) / c * c - base;
                if (w < pad) { pad = w; best = c; }
            }
            parts = best;
        }
        X3D_REQUIRE(parts >= 1 && parts <= base, "x3d_sfft010_create: %d column groups for %d columns", parts, base);
        p->parts = parts;
        p->xsc = (base + parts - 1) / parts;
        p->xs = p->xsc * parts;
    }
    p->nxs = p->xs * pz;
    X3D_REQUIRE(p->nx <= b->nxp && p->ny <= b->nyp && p->zl <= b->nzp, "x3d_sfft010_create: local block mismatch");
    const size_t n0 = (size_t)p->zl * p->ny * p->nxs, nw = (size_t)p->nz * p->ny * p->xs;
    X3D_HIP(hipMalloc(&p->c0, sizeof(real2_t) * n0));
    X3D_HIP(hipMemset(p->c0, 0, sizeof(real2_t) * n0));  // (pad columns stay zero: the transforms never write them)
    X3D_HIP(hipMalloc(&p->waves, sizeof(real_t) * nw));
    X3D_HIP(hipMalloc(&p->t, sizeof(real2_t) * nw));
    p->nab_x = p->nx > p->nxs ? p->nx : p->nxs;
    X3D_HIP(hipMalloc(&p->ab, sizeof(real_t) * 2 * ((size_t)p->nab_x + p->ny + p->nz)));
    X3D_HIP(hipMemset(p->ab, 0, sizeof(real_t) * 2 * ((size_t)p->nab_x + p->ny + p->nz)));
    hipfftHandle *pl[6] = {&p->plan_xy_fw, &p->plan_xy_bw, &p->plan_z, &p->plan_x_fw, &p->plan_x_bw, &p->plan_y};
    size_t ws[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        X3D_FFT(hipfftCreate(pl[i]));
        X3D_FFT(hipfftSetAutoAllocation(*pl[i], 0));
    }
    // real side: the pitched block's planes; spectral side: dense rows of nxs (the first nxm are written)
    int nn[2] = {p->ny, p->nx}, re[2] = {b->nyp, b->nxp}, ce[2] = {p->ny, p->nxs};
    const hipfftResult r_fw = hipfftMakePlanMany(p->plan_xy_fw, 2, nn, re, 1, b->nxp * b->nyp, ce, 1, p->ny * p->nxs,
                                                 X3D_FFT_R2C, p->zl, &ws[0]);
    const hipfftResult r_bw = hipfftMakePlanMany(p->plan_xy_bw, 2, nn, ce, 1, p->ny * p->nxs, re, 1, b->nxp * b->nyp,
                                                 X3D_FFT_C2R, p->zl, &ws[1]);
    {
        const char *e = getenv("X3D_SFFT010_SPLIT_XY");
        p->split_xy = r_fw != HIPFFT_SUCCESS || r_bw != HIPFFT_SUCCESS || (e && e[0] == '1');
    }
    if (p->split_xy) {
        ws[0] = ws[1] = 0;
        int nx1[1] = {p->nx}, rx[1] = {b->nxp}, cx[1] = {p->nxs}, ny1[1] = {p->ny}, ey[1] = {p->ny};
        // x: every local row (the block's rows are nxp apart whatever the plane: nyp rows per plane, of which ny are used --
        // one batch per plane keeps to the used rows)
        X3D_FFT(hipfftMakePlanMany(p->plan_x_fw, 1, nx1, rx, 1, b->nxp, cx, 1, p->nxs, X3D_FFT_R2C, p->ny, &ws[3]));
        X3D_FFT(hipfftMakePlanMany(p->plan_x_bw, 1, nx1, cx, 1, p->nxs, rx, 1, b->nxp, X3D_FFT_C2R, p->ny, &ws[4]));
        X3D_FFT(hipfftMakePlanMany(p->plan_y, 1, ny1, ey, p->nxs, 1, ey, p->nxs, 1, X3D_FFT_C2C, p->nxs, &ws[5]));
    }
    // z transform on the z-contiguous copy T[ny * xs][nz]
    int nzv[1] = {p->nz}, ze[1] = {p->nz};
    X3D_FFT(hipfftMakePlanMany(p->plan_z, 1, nzv, ze, 1, p->nz, ze, 1, p->nz, X3D_FFT_C2C, p->ny * p->xsc, &ws[2]));
    size_t wmax = 0;
    for (int i = 0; i < 6; i++) wmax = ws[i] > wmax ? ws[i] : wmax;
    if (wmax) X3D_HIP(hipMalloc(&p->work, wmax));
    for (int i = 0; i < 6; i++) {
        const bool used = i == 2 || (p->split_xy ? i >= 3 : i < 2);
        if (used) X3D_FFT(hipfftSetWorkArea(*pl[i], p->work));
    }
    *out = p;
    return 0;
}

extern "C" int x3d_sfft010_destroy(x3d_sfft010 *p)
{
    X3D_RANGE(__func__);
    if (!p) return 0;
    hipfftDestroy(p->plan_xy_fw); hipfftDestroy(p->plan_xy_bw); hipfftDestroy(p->plan_z);
    hipfftDestroy(p->plan_x_fw); hipfftDestroy(p->plan_x_bw); hipfftDestroy(p->plan_y);
    hipFree(p->c0); hipFree(p->t); hipFree(p->waves); hipFree(p->ab); hipFree(p->work); hipFree(p->lu[0]); hipFree(p->lu[1]);
    delete p;
    return 0;
}

// out = {chunk (complex numbers per peer), zl, xs, i0 (first x mode of this rank), nxm, parts}
extern "C" int x3d_sfft010_sizes(const x3d_sfft010 *p, long out[6])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && out, "null argument");
    out[0] = (long)p->zl * p->ny * p->xs; out[1] = p->zl; out[2] = p->xs; out[3] = (long)p->rz * p->xs; out[4] = p->nxm;
    out[5] = p->parts;
    return 0;
}

// waves: this rank's block [nz][ny][xs] (x fastest; pad columns: one); ax .. bz: the global tables
extern "C" int x3d_sfft010_set_waves(x3d_sfft010 *p, const real_t *waves, const real_t *ax, const real_t *bx,
                                     const real_t *ay, const real_t *by, const real_t *az, const real_t *bz)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && waves && ax && bx && ay && by && az && bz, "null argument");
    {   // [nz][ny][xs] on the host -> [part][ny][xsc][nz] on the device (T serves as the landing zone)
        const size_t nw = (size_t)p->nz * p->ny * p->xs;
        X3D_HIP(hipMemcpy(p->t, waves, sizeof(real_t) * nw, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sfft010_gather, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, p->b->stream, p->waves,
                           (const real_t *)p->t, p->nz, p->ny, p->xsc, p->parts, (long)p->ny * p->xsc * p->nz);
        X3D_HIP(hipGetLastError());
        X3D_HIP(hipStreamSynchronize(p->b->stream));
    }
    const real_t *src[6] = {ax, bx, ay, by, az, bz};
    const int len[6] = {p->nx, p->nx, p->ny, p->ny, p->nz, p->nz};
    const int slot[6] = {p->nab_x, p->nab_x, p->ny, p->ny, p->nz, p->nz};
    real_t *d = p->ab;
    for (int i = 0; i < 6; i++) {
        X3D_HIP(hipMemcpy(d, src[i], sizeof(real_t) * len[i], hipMemcpyHostToDevice));
        d += slot[i];
    }
    return 0;
}

// a0, a1: this rank's columns of the pentadiagonal operators, [5][nz][n][xs] (pad columns zero); sym: odd / even rows
extern "C" int x3d_sfft010_set_stretching(x3d_sfft010 *p, int sym, const real_t *a0, const real_t *a1)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && a0 && (!sym || a1), "x3d_sfft010_set_stretching: null argument");
    X3D_REQUIRE(!sym || p->ny % 2 == 0, "x3d_sfft010_set_stretching: odd/even split needs an even ny");
    const int n = sym ? p->ny / 2 : p->ny;
    X3D_REQUIRE(n >= 3, "x3d_sfft010_set_stretching: too few rows");
    const size_t nd = (size_t)p->nz * n * p->xs, bytes = sizeof(real_t) * 5 * nd;
    const real_t *src[2] = {a0, a1};
    real_t *tmp = nullptr;  // one diagonal [nz][n][xs] as uploaded, before it goes to its place in [part][5][n][xsc][nz]
    X3D_HIP(hipMalloc(&tmp, sizeof(real_t) * nd));
    const long ndp = (long)n * p->xsc * p->nz;  // one diagonal of one group
    for (int s = 0; s < (sym ? 2 : 1); s++) {
        if (!p->lu[s]) X3D_HIP(hipMalloc(&p->lu[s], bytes));
        for (int d = 0; d < 5; d++) {
            X3D_HIP(hipMemcpy(tmp, src[s] + d * nd, sizeof(real_t) * nd, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_sfft010_gather, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, p->b->stream,
                               p->lu[s] + d * ndp, (const real_t *)tmp, p->nz, n, p->xsc, p->parts, 5 * ndp);
            X3D_HIP(hipGetLastError());
            X3D_HIP(hipStreamSynchronize(p->b->stream));
        }
        for (int m = 0; m < p->parts; m++)
            hipLaunchKernelGGL(k_penta_factor<true>, penta_grid(p->xsc, p->nz), dim3(64), 0, p->b->stream,
                               p->lu[s] + (long)m * 5 * ndp, p->xsc, n, p->nz);
        X3D_HIP(hipGetLastError());
    }
    X3D_HIP(hipStreamSynchronize(p->b->stream));
    X3D_HIP(hipFree(tmp));
    p->stretched = 1;
    p->sym = sym;
    return 0;
}

// enforce / undo_periodicity_y on the rank's zl planes (y is whole on every rank)
extern "C" int x3d_sfft010_periodicity_y(x3d_sfft010 *p, real_t *f_out, const real_t *f_in, int undo)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out && f_in && f_out != f_in, "x3d_sfft010_periodicity_y: bad argument");
    X3D_LAZY_SYNC(p->b);  // (field blocks are read / written in place)
    x3d_backend *b = p->b;
    dim3 grid((p->nx + 255) / 256, p->ny, p->zl);
    ProfScope ps(b, X3D_K_COPY);
    if (undo)
        hipLaunchKernelGGL(k_periodicity_y<true>, grid, dim3(256), 0, b->stream, f_out, f_in, p->nx, p->ny, p->zl,
                           (long)b->nxp, (long)b->nxp * b->nyp);
    else
        hipLaunchKernelGGL(k_periodicity_y<false>, grid, dim3(256), 0, b->stream, f_out, f_in, p->nx, p->ny, p->zl,
                           (long)b->nxp, (long)b->nxp * b->nyp);
    X3D_HIP(hipGetLastError());
    return 0;
}

// 2-D transform of the local planes; the result lands in sendbuf as [peer][part][zl][ny][xsc]
extern "C" int x3d_sfft010_forward_local(x3d_sfft010 *p, const real_t *f_in, real_t *sendbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_in && sendbuf, "null argument");
    X3D_LAZY_SYNC(p->b);  // (field blocks are read / written in place)
    if (!p->split_xy) {
        ProfScope ps(p->b, X3D_K_FFT, 1);
        X3D_FFT(hipfftSetStream(p->plan_xy_fw, p->b->stream));
        X3D_FFT(x3d_fftExecR2C(p->plan_xy_fw, (x3d_fft_real *)f_in, (x3d_fft_cplx *)p->c0));
    } else {
        ProfScope ps(p->b, X3D_K_FFT, 1);
        X3D_FFT(hipfftSetStream(p->plan_x_fw, p->b->stream));
        X3D_FFT(hipfftSetStream(p->plan_y, p->b->stream));
        for (int k = 0; k < p->zl; k++) {
            x3d_fft_cplx *c = (x3d_fft_cplx *)(p->c0 + (size_t)k * p->ny * p->nxs);
            X3D_FFT(x3d_fftExecR2C(p->plan_x_fw, (x3d_fft_real *)f_in + (size_t)k * p->b->nxp * p->b->nyp, c));
            X3D_FFT(x3d_fftExecC2C(p->plan_y, c, c, HIPFFT_FORWARD));
        }
    }
    ProfScope ps(p->b, X3D_K_PACK);
    const long rows = (long)p->zl * p->ny, n = rows * p->nxs;
    hipLaunchKernelGGL(k_sfft010_pack<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->b->stream,
                       (real2_t *)sendbuf, p->c0, rows, p->xsc, p->parts, p->pz);
    X3D_HIP(hipGetLastError());
    return 0;
}

// dir 0: group `part` of the received array, W_m[nz][ny][xsc] -> T_m[ny][xsc][nz], forward z transform (the spectrum
// stays in T); dir 1: backward z transform of T_m, then back to W_m
extern "C" int x3d_sfft010_fft_z_part(x3d_sfft010 *p, real_t *recvbuf, int dir, int part)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf && part >= 0 && part < p->parts, "x3d_sfft010_fft_z_part: bad argument");
    const long cols = (long)p->ny * p->xsc;
    real2_t *W = (real2_t *)recvbuf + (size_t)part * p->nz * cols, *T = p->t + (size_t)part * p->nz * cols;
    if (dir == 0) {
        ProfScope ps(p->b, X3D_K_PACK);
        if (int rc = transpose_launch<real2_t>(p->b->stream, T, (const real2_t *)W, cols, p->nz)) return rc;
    }
    {
        ProfScope ps(p->b, X3D_K_FFT, 3);
        X3D_FFT(hipfftSetStream(p->plan_z, p->b->stream));
        X3D_FFT(x3d_fftExecC2C(p->plan_z, (x3d_fft_cplx *)T, (x3d_fft_cplx *)T, dir ? HIPFFT_BACKWARD : HIPFFT_FORWARD));
    }
    if (dir == 1) {
        ProfScope ps(p->b, X3D_K_PACK);
        if (int rc = transpose_launch<real2_t>(p->b->stream, W, (const real2_t *)T, p->nz, cols)) return rc;
    }
    return 0;
}

extern "C" int x3d_sfft010_fft_z(x3d_sfft010 *p, real_t *recvbuf, int dir)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    for (int m = 0; m < p->parts; m++)
        if (int rc = x3d_sfft010_fft_z_part(p, recvbuf, dir, m)) return rc;
    return 0;
}

// fft_postprocess_010 on one group of this rank's x modes (all rows, all z modes), in the z-contiguous copy
extern "C" int x3d_sfft010_postprocess_010_part(x3d_sfft010 *p, real_t *recvbuf, int part)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf && part >= 0 && part < p->parts, "x3d_sfft010_postprocess_010_part: bad argument");
    ProfScope ps(p->b, X3D_K_SPECTRAL);
    const real_t *ax = p->ab, *bx = ax + p->nab_x, *ay = bx + p->nab_x, *by = ay + p->ny, *az = by + p->ny,
                 *bz = az + p->nz;
    const size_t off = (size_t)part * p->nz * p->ny * p->xsc;
    const int n = p->sym ? p->ny / 2 : p->ny;
    real_t *lu[2] = {p->lu[0] ? p->lu[0] + (size_t)part * 5 * n * p->xsc * p->nz : nullptr,
                     p->lu[1] ? p->lu[1] + (size_t)part * 5 * n * p->xsc * p->nz : nullptr};
    return spectral_010_launch_t<true>(p->b->stream, p->t + off, p->waves + off, p->xsc, p->nx, p->ny, p->nz,
                                       p->rz * p->xs + part * p->xsc, ax, bx, ay, by, az, bz, p->stretched, p->sym, lu);
}

extern "C" int x3d_sfft010_postprocess_010(x3d_sfft010 *p, real_t *recvbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    for (int m = 0; m < p->parts; m++)
        if (int rc = x3d_sfft010_postprocess_010_part(p, recvbuf, m)) return rc;
    return 0;
}

// unpack the returned array S[peer][part][zl][ny][xsc] and transform back to the real planes
extern "C" int x3d_sfft010_backward_local(x3d_sfft010 *p, const real_t *sendbuf, real_t *f_out)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && sendbuf && f_out, "null argument");
    X3D_LAZY_SYNC(p->b);  // (field blocks are read / written in place)
    {
        ProfScope ps(p->b, X3D_K_PACK);
        const long rows = (long)p->zl * p->ny, n = rows * p->nxs;
        hipLaunchKernelGGL(k_sfft010_pack<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->b->stream,
                           (real2_t *)sendbuf, p->c0, rows, p->xsc, p->parts, p->pz);
        X3D_HIP(hipGetLastError());
    }
    ProfScope ps(p->b, X3D_K_FFT, 2);
    if (!p->split_xy) {
        X3D_FFT(hipfftSetStream(p->plan_xy_bw, p->b->stream));
        X3D_FFT(x3d_fftExecC2R(p->plan_xy_bw, (x3d_fft_cplx *)p->c0, (x3d_fft_real *)f_out));
        return 0;
    }
    X3D_FFT(hipfftSetStream(p->plan_x_bw, p->b->stream));
    X3D_FFT(hipfftSetStream(p->plan_y, p->b->stream));
    for (int k = 0; k < p->zl; k++) {
        x3d_fft_cplx *c = (x3d_fft_cplx *)(p->c0 + (size_t)k * p->ny * p->nxs);
        X3D_FFT(x3d_fftExecC2C(p->plan_y, c, c, HIPFFT_BACKWARD));
        X3D_FFT(x3d_fftExecC2R(p->plan_x_bw, c, (x3d_fft_real *)f_out + (size_t)k * p->b->nxp * p->b->nyp));
    }
    return 0;
}
