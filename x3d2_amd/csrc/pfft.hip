// Distributed spectral Poisson solver (all-periodic, 000): pencil FFT over a
// [1, py, pz] decomposition -- x whole, y split over py ranks, z over pz ranks
// -- the same 2-D decomposition 2decomp&FFT gives the reference's CPU backend
// (/root/reference/src/decomp/decomp_2decompfft.f90:42-48,
// src/backend/omp/poisson_fft.f90:72-97).  This file holds the LOCAL stages
// (rocFFT batched 1-D transforms, pack/unpack, spectral post-processing); the
// two transposes per direction are exchanges between the stages, done by the
// caller over RCCL (x3d2_amd/poisson_fft.py).
//
//   X pencils  f[zl][yl][nx]   --R2C x-->  C0[zl][yl][nxs]            (x fastest)
//   --xy exchange in the py group-->       C1[zl][xs][ny]             (y fastest)
//   --C2C y-->  --yz exchange in the pz group--> C2[xs][ys][nz]       (z fastest)
//   --C2C z--> process_spectral_000 --> inverse path.
// xs / ys are this rank's shares of the nx/2+1 x-modes / ny y-modes
// (first `rem` ranks get one extra), all pencil axes are contiguous for rocFFT.
//
// Everything up to the z transform is independent from one local z plane to the next, so the solve can run in
// `parts` groups of zp = zl / parts planes (x3d_pfft_create_parts, the *_part entry points): the x transform, the
// xy exchange, the y transform and the yz exchange of a group run beside the transfers of the others; the
// reference's 2decomp&FFT transposes are blocking (src/backend/omp/poisson_fft.f90:99-137).  Exchange buffers of a
// group: [part][peer][zp][..][..], a peer's chunk contiguous (x3d_pfft_part_layout).
#include <hipfft/hipfft.h>

#include "common.h"

#define X3D_FFT(expr)                                                                          \
    do {                                                                                       \
        hipfftResult r_ = (expr);                                                              \
        if (r_ != HIPFFT_SUCCESS) {                                                            \
            x3d_set_error("%s failed: hipfft error %d (%s:%d)", #expr, (int)r_, __FILE__,      \
                          __LINE__);                                                           \
            return 3;                                                                          \
        }                                                                                      \
    } while (0)

static inline int share(int n, int p, int r) { return n / p + (r < n % p ? 1 : 0); }
static inline int share_off(int n, int p, int r) { return r * (n / p) + (r < n % p ? r : n % p); }

struct x3d_pfft {
    x3d_backend *b;
    int nx, ny, nz, nxs;          // global cell dims
    int py, pz, ry, rz;
    int yl, zl;                   // local physical extents (ny/py, nz/pz)
    int xs, xoff, ys, yoff;       // spectral shares of this rank
    int parts, zp;                // groups of local z planes (zp = zl / parts); the x and y plans are per group
    hipfftHandle plan_r2c, plan_c2r, plan_y, plan_z;
    real2_t *c0, *c1, *c2;        // stage buffers
    real_t *waves, *ab;
    void *work;
    // x3d_pfft_own_chunk: the next unpack takes the chunk of peer own_rank (this rank's own) out of the SEND buffer own
    const real2_t *own;
    int own_rank;
};

// dst[i*d0 + j*d1 + k*d2] = src[i*s0 + j*s1 + k*s2]; i is the fastest loop index
__global__ void __launch_bounds__(256) k_permute(real2_t *__restrict__ dst, const real2_t *__restrict__ src, int n0,
                                                 int n1, int n2, long d0, long d1, long d2, long s0, long s1, long s2)
{
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long tot = (long)n0 * n1 * n2;
    if (q >= tot) return;
    const int i = (int)(q % n0);
    const int j = (int)((q / n0) % n1);
    const int k = (int)(q / ((long)n0 * n1));
    dst[i * d0 + j * d1 + k * d2] = src[i * s0 + j * s1 + k * s2];
}

// ... when BOTH are contiguous along the first index (a strided copy of rows: the exchange chunks cut along x or y): no
// index arithmetic per element -- blockIdx.y / .z are the two outer indices (k_permute's 64-bit divisions cost it half its rate)
__global__ void __launch_bounds__(256) k_copy_rows(real2_t *__restrict__ dst, const real2_t *__restrict__ src, int n0,
                                                   long d1, long d2, long s1, long s2)
{
    const long so = (long)blockIdx.y * s1 + (long)blockIdx.z * s2, dof = (long)blockIdx.y * d1 + (long)blockIdx.z * d2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += gridDim.x * blockDim.x) dst[dof + i] = src[so + i];
}

// the same when src is contiguous along dimension A and dst along a different dimension B (a genuine
// transposition): 32 x 32 tiles through LDS so that both the loads (along A) and the stores (along B) are
// 512-byte contiguous; C is the remaining dimension.  k_permute alone runs at 2.5 TB/s on these.
__global__ void __launch_bounds__(256)
    k_transpose_tiled(real2_t *__restrict__ dst, const real2_t *__restrict__ src, int nA, int nB, int nC, long dA,
                      long dC, long sB, long sC)
{
    __shared__ real2_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32, c = blockIdx.z;
    const real2_t *__restrict__ sp = src + (long)c * sC;
    real2_t *__restrict__ dp = dst + (long)c * dC;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int bb = b0 + ty + 8 * r, aa = a0 + tx;
        if (aa < nA && bb < nB) tile[ty + 8 * r][tx] = sp[aa + (long)bb * sB];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int aa = a0 + ty + 8 * r, bb = b0 + tx;
        if (aa < nA && bb < nB) dp[(long)aa * dA + bb] = tile[tx][ty + 8 * r];
    }
}

static int permute(x3d_backend *b, real2_t *dst, const real2_t *src, int n0, int n1, int n2, long d0, long d1,
                   long d2, long s0, long s1, long s2)
{
    const long tot = (long)n0 * n1 * n2;
    if (tot == 0) return 0;
    ProfScope ps(b, X3D_K_PACK);
    const int n[3] = {n0, n1, n2};
    const long d[3] = {d0, d1, d2}, s_[3] = {s0, s1, s2};
    int A = -1, B = -1;
    for (int i = 0; i < 3; i++) {
        if (s_[i] == 1 && A < 0) A = i;
        if (d[i] == 1 && B < 0) B = i;
    }
    if (A >= 0 && B >= 0 && A != B && n[3 - A - B] <= 65535) {
        const int C = 3 - A - B;
        dim3 grid((n[A] + 31) / 32, (n[B] + 31) / 32, n[C]);
        if (grid.y <= 65535) {
            hipLaunchKernelGGL(k_transpose_tiled, grid, dim3(256), 0, b->stream, dst, src, n[A], n[B], n[C], d[A], d[C],
                               s_[B], s_[C]);
            X3D_HIP(hipGetLastError());
            return 0;
        }
    }
    if (s0 == 1 && d0 == 1 && n1 <= 65535 && n2 <= 65535) {
        hipLaunchKernelGGL(k_copy_rows, dim3((n0 + 255) / 256, n1, n2), dim3(n0 >= 256 ? 256 : (n0 > 64 ? 128 : 64)), 0, b->stream,
                           dst, src, n0, d1, d2, s1, s2);
        X3D_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_permute, dim3((tot + 255) / 256), dim3(256), 0, b->stream, dst, src, n0, n1, n2, d0, d1, d2,
                       s0, s1, s2);
    X3D_HIP(hipGetLastError());
    return 0;
}

// process_spectral_000 on the z-fastest spectral pencils C2[xs][ys][nz]
// (src/backend/omp/kernels/spectral_processing.f90:7-106, offsets = sp_st)
__global__ void __launch_bounds__(256)
    k_process_spectral_000_z(real2_t *__restrict__ c, const real_t *__restrict__ waves, int xs, int ys, int nz,
                             int xoff, int yoff, int nx, int ny, const real_t *__restrict__ ax,
                             const real_t *__restrict__ bx, const real_t *__restrict__ ay,
                             const real_t *__restrict__ by, const real_t *__restrict__ az,
                             const real_t *__restrict__ bz)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;  // iz - 1
    const int jl = blockIdx.y, il = blockIdx.z;
    if (k >= nz) return;
    const int i = il + xoff, j = jl + yoff;
    const size_t idx = ((size_t)il * ys + jl) * nz + k;
    real2_t v = c[idx];
    real_t div_r = v.x / nx / ny / nz, div_c = v.y / nx / ny / nz;
    const real_t azk = az[k], bzk = bz[k], ayj = ay[j], byj = by[j], axi = ax[i], bxi = bx[i];
    const bool fz = (k + 1) > nz / 2 + 1, fy = (j + 1) > ny / 2 + 1;
    real_t tr, tc;
    tr = div_r; tc = div_c;
    div_r = tr * bzk + tc * azk; div_c = tc * bzk - tr * azk;
    if (fz) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;
    div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
    if (fy) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;
    div_r = tr * bxi + tc * axi; div_c = tc * bxi - tr * axi;
    const real_t wv = waves[idx];
    if (wv < 1.e-16) { div_r = 0.0; div_c = 0.0; }
    else { div_r = -div_r / wv; div_c = -div_c / wv; }
    tr = div_r; tc = div_c;
    div_r = tr * bzk - tc * azk; div_c = -tc * bzk - tr * azk;
    if (fz) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;
    div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
    if (fy) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;
    div_r = tr * bxi + tc * axi; div_c = -tc * bxi + tr * axi;
    c[idx] = make_real2(div_r, div_c);
}

extern "C" int x3d_pfft_create_parts(x3d_backend *b, x3d_pfft **out, const int nglob[3], int py, int pz, int ry,
                                     int rz, int parts);
extern "C" int x3d_pfft_create(x3d_backend *b, x3d_pfft **out, const int nglob[3], int py, int pz, int ry, int rz)
{
    X3D_RANGE(__func__);
    return x3d_pfft_create_parts(b, out, nglob, py, pz, ry, rz, 1);
}

// parts <= 0: the library's choice (4, or the largest divisor of zl below it)
extern "C" int x3d_pfft_create_parts(x3d_backend *b, x3d_pfft **out, const int nglob[3], int py, int pz, int ry,
                                     int rz, int parts)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && nglob, "x3d_pfft_create: null argument");
    X3D_REQUIRE(py >= 1 && pz >= 1 && ry >= 0 && ry < py && rz >= 0 && rz < pz, "x3d_pfft_create: bad rank grid");
    X3D_REQUIRE(nglob[1] % py == 0 && nglob[2] % pz == 0, "x3d_pfft_create: ny, nz must divide by py, pz");
    x3d_pfft *p = new x3d_pfft();
    memset(p, 0, sizeof *p);
    p->b = b;
    p->nx = nglob[0]; p->ny = nglob[1]; p->nz = nglob[2]; p->nxs = nglob[0] / 2 + 1;
    p->py = py; p->pz = pz; p->ry = ry; p->rz = rz;
    p->yl = p->ny / py; p->zl = p->nz / pz;
    X3D_REQUIRE(p->nx <= b->nxp && p->yl == b->nyp && p->zl <= b->nzp,
                "x3d_pfft_create: the block (%d x %d x %d) does not hold %d x %d x %d cells per rank", b->nxp, b->nyp,
                b->nzp, p->nx, p->yl, p->zl);
    if (parts <= 0) for (parts = 4; parts > 1 && p->zl % parts; parts--) {}
    X3D_REQUIRE(parts >= 1 && p->zl % parts == 0, "x3d_pfft_create: %d parts do not divide %d local z planes", parts,
                p->zl);
    p->parts = parts; p->zp = p->zl / parts;
    p->xs = share(p->nxs, py, ry); p->xoff = share_off(p->nxs, py, ry);
    p->ys = share(p->ny, pz, rz); p->yoff = share_off(p->ny, pz, rz);
    const size_t n0 = (size_t)p->zl * p->yl * p->nxs, n1 = (size_t)p->zl * p->xs * p->ny,
                 n2 = (size_t)p->xs * p->ys * p->nz;
    X3D_HIP(hipMalloc(&p->c0, sizeof(real2_t) * n0));
    X3D_HIP(hipMalloc(&p->c1, sizeof(real2_t) * n1));
    X3D_HIP(hipMalloc(&p->c2, sizeof(real2_t) * n2));
    X3D_HIP(hipMalloc(&p->waves, sizeof(real_t) * n2));
    X3D_HIP(hipMalloc(&p->ab, sizeof(real_t) * 2 * ((size_t)p->nx + p->ny + p->nz)));
    int nxv[1] = {p->nx}, nyv[1] = {p->ny}, nzv[1] = {p->nz};
    size_t ws[4] = {0, 0, 0, 0};
    hipfftHandle *pl[4] = {&p->plan_r2c, &p->plan_c2r, &p->plan_y, &p->plan_z};
    for (int i = 0; i < 4; i++) {
        X3D_FFT(hipfftCreate(pl[i]));
        X3D_FFT(hipfftSetAutoAllocation(*pl[i], 0));
    }
    int rembed[1] = {b->nxp}, cembed[1] = {p->nxs};
    X3D_FFT(hipfftMakePlanMany(p->plan_r2c, 1, nxv, rembed, 1, b->nxp, cembed, 1, p->nxs, X3D_FFT_R2C,
                               p->yl * p->zp, &ws[0]));
    X3D_FFT(hipfftMakePlanMany(p->plan_c2r, 1, nxv, cembed, 1, p->nxs, rembed, 1, b->nxp, X3D_FFT_C2R,
                               p->yl * p->zp, &ws[1]));
    X3D_FFT(hipfftMakePlanMany(p->plan_y, 1, nyv, nyv, 1, p->ny, nyv, 1, p->ny, X3D_FFT_C2C, p->zp * p->xs, &ws[2]));
    X3D_FFT(hipfftMakePlanMany(p->plan_z, 1, nzv, nzv, 1, p->nz, nzv, 1, p->nz, X3D_FFT_C2C, p->xs * p->ys, &ws[3]));
    size_t wmax = 0;
    for (int i = 0; i < 4; i++) wmax = ws[i] > wmax ? ws[i] : wmax;
    if (wmax) X3D_HIP(hipMalloc(&p->work, wmax));
    for (int i = 0; i < 4; i++) X3D_FFT(hipfftSetWorkArea(*pl[i], p->work));
    *out = p;
    return 0;
}

extern "C" int x3d_pfft_destroy(x3d_pfft *p)
{
    X3D_RANGE(__func__);
    if (!p) return 0;
    hipfftDestroy(p->plan_r2c); hipfftDestroy(p->plan_c2r); hipfftDestroy(p->plan_y); hipfftDestroy(p->plan_z);
    hipFree(p->c0); hipFree(p->c1); hipFree(p->c2); hipFree(p->waves); hipFree(p->ab); hipFree(p->work);
    delete p;
    return 0;
}

// out = {xs, xoff, ys, yoff, yl, zl, nxs, n_exchange_max (complex elements)}
extern "C" int x3d_pfft_sizes(const x3d_pfft *p, long out[8])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && out, "null argument");
    const long n0 = (long)p->zl * p->yl * p->nxs, n1 = (long)p->zl * p->xs * p->ny, n2 = (long)p->xs * p->ys * p->nz;
    long m = n0 > n1 ? n0 : n1;
    m = m > n2 ? m : n2;
    out[0] = p->xs; out[1] = p->xoff; out[2] = p->ys; out[3] = p->yoff; out[4] = p->yl; out[5] = p->zl;
    out[6] = p->nxs; out[7] = m;
    return 0;
}

// waves_re: this rank's block [xs][ys][nz] (z fastest); ax..bz: global-length arrays
extern "C" int x3d_pfft_set_waves(x3d_pfft *p, const real_t *waves_re, const real_t *ax, const real_t *bx,
                                  const real_t *ay, const real_t *by, const real_t *az, const real_t *bz)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && waves_re && ax && bx && ay && by && az && bz, "null argument");
    const size_t n2 = (size_t)p->xs * p->ys * p->nz;
    X3D_HIP(hipMemcpy(p->waves, waves_re, sizeof(real_t) * n2, hipMemcpyHostToDevice));
    real_t *d = p->ab;
    const real_t *src[6] = {ax, bx, ay, by, az, bz};
    const int len[6] = {p->nx, p->nx, p->ny, p->ny, p->nz, p->nz};
    for (int i = 0; i < 6; i++) {
        X3D_HIP(hipMemcpy(d, src[i], sizeof(real_t) * len[i], hipMemcpyHostToDevice));
        d += len[i];
    }
    return 0;
}

// ---- local stages on the planes [z0, z0 + nzp_) of this rank (a whole number of groups)
static int fwd_x(x3d_pfft *p, const real_t *f_in, int m0, int m1)
{
    ProfScope ps(p->b, X3D_K_FFT, 1);
    X3D_FFT(hipfftSetStream(p->plan_r2c, p->b->stream));
    for (int m = m0; m < m1; m++)
        X3D_FFT(x3d_fftExecR2C(p->plan_r2c, (x3d_fft_real *)f_in + (size_t)m * p->zp * p->yl * p->b->nxp,
                              (x3d_fft_cplx *)p->c0 + (size_t)m * p->zp * p->yl * p->nxs));
    return 0;
}
static int bwd_x(x3d_pfft *p, real_t *f_out, int m0, int m1)
{
    ProfScope ps(p->b, X3D_K_FFT, 2);
    X3D_FFT(hipfftSetStream(p->plan_c2r, p->b->stream));
    for (int m = m0; m < m1; m++)
        X3D_FFT(x3d_fftExecC2R(p->plan_c2r, (x3d_fft_cplx *)p->c0 + (size_t)m * p->zp * p->yl * p->nxs,
                              (x3d_fft_real *)f_out + (size_t)m * p->zp * p->yl * p->b->nxp));
    return 0;
}
static int fft_y(x3d_pfft *p, int inverse, int m0, int m1)
{
    if (p->xs == 0) return 0;
    ProfScope ps(p->b, X3D_K_FFT, 3);
    X3D_FFT(hipfftSetStream(p->plan_y, p->b->stream));
    for (int m = m0; m < m1; m++) {
        x3d_fft_cplx *c = (x3d_fft_cplx *)p->c1 + (size_t)m * p->zp * p->xs * p->ny;
        X3D_FFT(x3d_fftExecC2C(p->plan_y, c, c, inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD));
    }
    return 0;
}

extern "C" int x3d_pfft_fwd_x(x3d_pfft *p, const real_t *f_in)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_in, "null argument");
    X3D_LAZY_IN(p->b, f_in);
    return fwd_x(p, f_in, 0, p->parts);
}

extern "C" int x3d_pfft_bwd_x(x3d_pfft *p, real_t *f_out)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out, "null argument");
    // the whole real extent is written and nothing reads a block's padding: a block that still shares its buffer -- the
    // reference's p_temp is a reordered alias of div_u, released only behind the solve (src/solver.f90:653-678) -- takes a
    // free buffer instead of a copy of the old contents
    X3D_LAZY_OUT(p->b, f_out, true);
    return bwd_x(p, f_out, 0, p->parts);
}

extern "C" int x3d_pfft_fft_y(x3d_pfft *p, int inverse)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p, "null argument");
    return fft_y(p, inverse, 0, p->parts);
}

extern "C" int x3d_pfft_fft_z(x3d_pfft *p, int inverse)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p, "null argument");
    if (p->xs == 0 || p->ys == 0) return 0;
    ProfScope ps(p->b, X3D_K_FFT, 3);
    X3D_FFT(hipfftSetStream(p->plan_z, p->b->stream));
    X3D_FFT(x3d_fftExecC2C(p->plan_z, (x3d_fft_cplx *)p->c2, (x3d_fft_cplx *)p->c2,
                          inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD));
    return 0;
}

// ---- exchange buffers.  For the planes [z0, z0 + nzl) peer r's chunk is contiguous:
// xy: send chunk r = C0[z0.., :, xoff_r : xoff_r+xs_r] packed [nzl][yl][xs_r] at s + xoff_r * yl * nzl;
//     recv chunk r = [nzl][yl][xs] from the rank owning y-slab r  ->  C1[z][x][r*yl + y]
// yz: send chunk r = C1[z0.., :, yoff_r : yoff_r+ys_r] packed [nzl][xs][ys_r] at s + yoff_r * xs * nzl;
//     recv chunk r = [nzl][xs][ys] from the rank owning z-slab r  ->  C2[x][y][r*zl + z]
// (whole solve: z0 = 0, nzl = zl; a group: z0 = m * zp, nzl = zp, s = the group's piece of the buffer)
static int xy_c0(x3d_pfft *p, real2_t *s, int z0, int nzl, bool pack)
{
    real2_t *c0 = p->c0 + (size_t)z0 * p->yl * p->nxs;
    for (int r = 0; r < p->py; r++) {
        const int xr = share(p->nxs, p->py, r), xo = share_off(p->nxs, p->py, r);
        real2_t *ch = s + (long)xo * p->yl * nzl;
        const real2_t *src = (!pack && p->own && r == p->own_rank) ? p->own + (long)r * p->xs * p->yl * nzl : ch;
        const int rc = pack ? permute(p->b, ch, c0 + xo, xr, p->yl, nzl, 1, xr, (long)xr * p->yl, 1, p->nxs,
                                      (long)p->nxs * p->yl)
                            : permute(p->b, c0 + xo, src, xr, p->yl, nzl, 1, p->nxs, (long)p->nxs * p->yl, 1, xr,
                                      (long)xr * p->yl);
        if (rc) return rc;
    }
    if (!pack) p->own = nullptr;
    return 0;
}
static int xy_c1(x3d_pfft *p, real2_t *s, int z0, int nzl, bool pack)
{
    real2_t *c1 = p->c1 + (size_t)z0 * p->xs * p->ny;
    const long chunk = (long)p->xs * p->yl * nzl;
    for (int r = 0; r < p->py; r++) {
        // (own chunk straight out of the send buffer, where xy_c0 packed it: x3d_pfft_own_chunk)
        const real2_t *src = (!pack && p->own && r == p->own_rank) ? p->own + (long)share_off(p->nxs, p->py, r) * p->yl * nzl
                                                                    : s + r * chunk;
        const int rc = pack ? permute(p->b, s + r * chunk, c1 + (long)r * p->yl, p->xs, p->yl, nzl, 1, p->xs,
                                      (long)p->xs * p->yl, p->ny, 1, (long)p->xs * p->ny)
                            : permute(p->b, c1 + (long)r * p->yl, src, p->xs, p->yl, nzl, p->ny, 1,
                                      (long)p->xs * p->ny, 1, p->xs, (long)p->xs * p->yl);
        if (rc) return rc;
    }
    if (!pack) p->own = nullptr;
    return 0;
}
static int yz_c1(x3d_pfft *p, real2_t *s, int z0, int nzl, bool pack)
{
    real2_t *c1 = p->c1 + (size_t)z0 * p->xs * p->ny;
    for (int r = 0; r < p->pz; r++) {
        const int yr = share(p->ny, p->pz, r), yo = share_off(p->ny, p->pz, r);
        real2_t *ch = s + (long)yo * p->xs * nzl;
        const int rc = pack ? permute(p->b, ch, c1 + yo, yr, p->xs, nzl, 1, yr, (long)yr * p->xs, 1, p->ny,
                                      (long)p->ny * p->xs)
                            : permute(p->b, c1 + yo, ch, yr, p->xs, nzl, 1, p->ny, (long)p->ny * p->xs, 1, yr,
                                      (long)yr * p->xs);
        if (rc) return rc;
    }
    return 0;
}
static int yz_c2(x3d_pfft *p, real2_t *s, int z0, int nzl, bool pack)
{
    const long chunk = (long)p->ys * p->xs * nzl;
    for (int r = 0; r < p->pz; r++) {
        real2_t *c2 = p->c2 + (long)r * p->zl + z0;
        const int rc = pack ? permute(p->b, s + r * chunk, c2, p->ys, p->xs, nzl, 1, p->ys, (long)p->ys * p->xs,
                                      p->nz, (long)p->ys * p->nz, 1)
                            : permute(p->b, c2, s + r * chunk, p->ys, p->xs, nzl, p->nz, (long)p->ys * p->nz, 1, 1,
                                      p->ys, (long)p->ys * p->xs);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int x3d_pfft_pack_xy(x3d_pfft *p, real_t *sendbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && sendbuf, "null argument");
    return xy_c0(p, (real2_t *)sendbuf, 0, p->zl, true);
}
extern "C" int x3d_pfft_unpack_xy(x3d_pfft *p, const real_t *recvbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    return xy_c1(p, (real2_t *)recvbuf, 0, p->zl, false);
}
// inverse of the pair above: C1 -> chunks [zl][yl][xs] per y-slab owner -> C0 columns
extern "C" int x3d_pfft_pack_yx(x3d_pfft *p, real_t *sendbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && sendbuf, "null argument");
    return xy_c1(p, (real2_t *)sendbuf, 0, p->zl, true);
}
extern "C" int x3d_pfft_unpack_yx(x3d_pfft *p, const real_t *recvbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    return xy_c0(p, (real2_t *)recvbuf, 0, p->zl, false);
}
extern "C" int x3d_pfft_pack_yz(x3d_pfft *p, real_t *sendbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && sendbuf, "null argument");
    return yz_c1(p, (real2_t *)sendbuf, 0, p->zl, true);
}
extern "C" int x3d_pfft_unpack_yz(x3d_pfft *p, const real_t *recvbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    return yz_c2(p, (real2_t *)recvbuf, 0, p->zl, false);
}
extern "C" int x3d_pfft_pack_zy(x3d_pfft *p, real_t *sendbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && sendbuf, "null argument");
    return yz_c2(p, (real2_t *)sendbuf, 0, p->zl, true);
}
extern "C" int x3d_pfft_unpack_zy(x3d_pfft *p, const real_t *recvbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    return yz_c1(p, (real2_t *)recvbuf, 0, p->zl, false);
}

// A direction that is NOT divided (py == 1 or pz == 1): the "exchange" of that transposition is with oneself, the packed
// chunk has exactly the layout of the stage buffer it was cut from -- so the stage buffer itself is the exchange buffer and
// the transposition is one kernel.  which = 0: x-y forward (C0 -> C1), 1: y-z forward (C1 -> C2), 2: z-y backward, 3: y-x
// the next x3d_pfft_unpack_xy / _unpack_yx reads the chunk of peer `rank` (this rank itself) out of `sendbuf`, the buffer the
// matching pack filled, instead of the receive buffer: the host need not copy a rank's own chunk from one to the other
extern "C" int x3d_pfft_own_chunk(x3d_pfft *p, const real_t *sendbuf, int rank)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && sendbuf && rank >= 0 && rank < p->py, "x3d_pfft_own_chunk: bad argument");
    p->own = (const real2_t *)sendbuf;
    p->own_rank = rank;
    return 0;
}

extern "C" int x3d_pfft_transpose_local(x3d_pfft *p, int which)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && which >= 0 && which <= 3, "x3d_pfft_transpose_local: bad argument");
    X3D_REQUIRE((which == 0 || which == 3) ? p->py == 1 : p->pz == 1, "x3d_pfft_transpose_local: the direction is divided");
    switch (which) {
    case 0: return xy_c1(p, p->c0, 0, p->zl, false);
    case 1: return yz_c2(p, p->c1, 0, p->zl, false);
    case 2: return yz_c2(p, p->c1, 0, p->zl, true);
    default: return xy_c1(p, p->c0, 0, p->zl, true);
    }
}

// ---- the solve in groups of planes.  out = {parts, zp, then in complex elements: a group's piece of the xy send
// buffer (peer r's chunk at xoff_r * yl * zp, xs_r * yl * zp long), of the xy receive buffer (peer r's chunk at
// r * xs * yl * zp), of the yz send buffer (peer r's chunk at yoff_r * xs * zp, ys_r * xs * zp long), of the yz
// receive buffer (peer r's chunk at r * ys * xs * zp)}; group m's piece starts at m times that
extern "C" int x3d_pfft_part_layout(const x3d_pfft *p, long out[6])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && out, "null argument");
    out[0] = p->parts; out[1] = p->zp;
    out[2] = (long)p->nxs * p->yl * p->zp;
    out[3] = (long)p->py * p->xs * p->yl * p->zp;
    out[4] = (long)p->ny * p->xs * p->zp;
    out[5] = (long)p->pz * p->ys * p->xs * p->zp;
    return 0;
}
// (x3d_pfft_own_chunk points at a whole-solve send buffer: a group's unpack would read another group's planes -- ADVICE round 5)
#define PFFT_PART(p, m, name)                                                                               \
    X3D_REQUIRE((p) && (m) >= 0 && (m) < (p)->parts, name ": group %d of %d", (m), (p) ? (p)->parts : 0);   \
    X3D_REQUIRE(!(p)->own, name ": x3d_pfft_own_chunk is pending (whole-solve unpacks only)")
// forward: R2C x of the group, its xy chunks into send_xy
extern "C" int x3d_pfft_fwd_a_part(x3d_pfft *p, const real_t *f_in, real_t *send_xy, int m)
{
    X3D_RANGE(__func__);
    PFFT_PART(p, m, "x3d_pfft_fwd_a_part");
    X3D_REQUIRE(f_in && send_xy, "null argument");
    X3D_LAZY_IN(p->b, f_in);  // (deferred execution: flush, then the buffer that holds the field)
    if (int rc = fwd_x(p, f_in, m, m + 1)) return rc;
    return xy_c0(p, (real2_t *)send_xy + (long)m * p->nxs * p->yl * p->zp, m * p->zp, p->zp, true);
}
// received xy chunks -> y pencils, C2C y, yz chunks into send_yz
extern "C" int x3d_pfft_fwd_b_part(x3d_pfft *p, const real_t *recv_xy, real_t *send_yz, int m)
{
    X3D_RANGE(__func__);
    PFFT_PART(p, m, "x3d_pfft_fwd_b_part");
    X3D_REQUIRE(recv_xy && send_yz, "null argument");
    if (int rc = xy_c1(p, (real2_t *)recv_xy + (long)m * p->py * p->xs * p->yl * p->zp, m * p->zp, p->zp, false))
        return rc;
    if (int rc = fft_y(p, 0, m, m + 1)) return rc;
    return yz_c1(p, (real2_t *)send_yz + (long)m * p->ny * p->xs * p->zp, m * p->zp, p->zp, true);
}
// received yz chunks -> the group's planes of the z pencils (x3d_pfft_fft_z once every group is in)
extern "C" int x3d_pfft_fwd_c_part(x3d_pfft *p, const real_t *recv_yz, int m)
{
    X3D_RANGE(__func__);
    PFFT_PART(p, m, "x3d_pfft_fwd_c_part");
    X3D_REQUIRE(recv_yz, "null argument");
    return yz_c2(p, (real2_t *)recv_yz + (long)m * p->pz * p->ys * p->xs * p->zp, m * p->zp, p->zp, false);
}
// backward, the same three in reverse (buffers: what was received forward is sent now)
extern "C" int x3d_pfft_bwd_c_part(x3d_pfft *p, real_t *send_zy, int m)
{
    X3D_RANGE(__func__);
    PFFT_PART(p, m, "x3d_pfft_bwd_c_part");
    X3D_REQUIRE(send_zy, "null argument");
    return yz_c2(p, (real2_t *)send_zy + (long)m * p->pz * p->ys * p->xs * p->zp, m * p->zp, p->zp, true);
}
extern "C" int x3d_pfft_bwd_b_part(x3d_pfft *p, const real_t *recv_zy, real_t *send_yx, int m)
{
    X3D_RANGE(__func__);
    PFFT_PART(p, m, "x3d_pfft_bwd_b_part");
    X3D_REQUIRE(recv_zy && send_yx, "null argument");
    if (int rc = yz_c1(p, (real2_t *)recv_zy + (long)m * p->ny * p->xs * p->zp, m * p->zp, p->zp, false)) return rc;
    if (int rc = fft_y(p, 1, m, m + 1)) return rc;
    return xy_c1(p, (real2_t *)send_yx + (long)m * p->py * p->xs * p->yl * p->zp, m * p->zp, p->zp, true);
}
extern "C" int x3d_pfft_bwd_a_part(x3d_pfft *p, const real_t *recv_yx, real_t *f_out, int m)
{
    X3D_RANGE(__func__);
    PFFT_PART(p, m, "x3d_pfft_bwd_a_part");
    X3D_REQUIRE(recv_yx && f_out, "null argument");
    // (a group's planes of the real extent are written; the first group of a solve takes the block over -- see
    //  x3d_pfft_bwd_x -- and finds it its own from then on)
    // (m > 0: a block that is shared AGAIN by now -- an alias recorded between two groups -- keeps the planes the earlier
    //  groups wrote; a solely owned block gets its own buffer back at no cost)
    X3D_LAZY_OUT(p->b, f_out, m == 0);
    if (int rc = xy_c0(p, (real2_t *)recv_yx + (long)m * p->nxs * p->yl * p->zp, m * p->zp, p->zp, false)) return rc;
    return bwd_x(p, f_out, m, m + 1);
}

extern "C" int x3d_pfft_postprocess_000(x3d_pfft *p)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p, "null argument");
    if (p->xs == 0 || p->ys == 0) return 0;
    const real_t *ax = p->ab, *bx = ax + p->nx, *ay = bx + p->nx, *by = ay + p->ny, *az = by + p->ny,
                 *bz = az + p->nz;
    dim3 grid((p->nz + 255) / 256, p->ys, p->xs);
    ProfScope ps(p->b, X3D_K_SPECTRAL);
    hipLaunchKernelGGL(k_process_spectral_000_z, grid, dim3(256), 0, p->b->stream, p->c2, p->waves, p->xs, p->ys,
                       p->nz, p->xoff, p->yoff, p->nx, p->ny, ax, bx, ay, by, az, bz);
    X3D_HIP(hipGetLastError());
    return 0;
}
