// Distributed spectral Poisson solver (000) for z-slab decompositions [1, 1, pz] with ny = 512:
// the layout bench.py uses on N GPUs, and the one the reference's own GPU backend requires
// (/root/reference/src/backend/cuda/poisson_fft.py:219 "1,1,N"; hooks src/poisson_fft.f90:45-62,
// CPU analogue through 2decomp&FFT: src/backend/omp/poisson_fft.f90:72-137).
//
// y is local, so there is ONE transpose pair per solve and no separate pack / unpack pass at all:
//   f[zl][ny][nx]  --rocFFT R2C x-->  C0[zl][ny][nxs]
//   --k_fft512 (y forward), storing straight into the exchange layout-->  S[r][zl][ys][nxs]
//        (chunk r = the y-range that rank r will own, contiguous -> sent as is)
//   --all-to-all among the pz ranks-->  R[r][zl][ys][nxs] = R[nz][ys][nxs]  (chunks arrive in z order)
//   --512 planes per rank: k_fft512_peers / k_radix_peers (fft512.hip) on R itself: DFTs across the pz chunks,
//     512-point transforms, process_spectral_000 and the inverses in ONE kernel (pz = 8: three), 2 passes
//   --other plane counts: 32x32 LDS-tiled transpose to T[ys*nxs][nz], rocFFT C2C z (contiguous),
//     process_spectral_000, inverse z, transpose back (rocFFT's strided 1-D plan on R itself takes 2.05 ms per
//     direction at 512^3, the transpose + contiguous transform 0.42 + 0.45)
//   --all-to-all back (chunk r of R = rank r's z-range, contiguous)-->  S
// Overlap: a rank's share of the y modes is cut into `parts` pieces, S = [peer][part][zl][ysc][nxs] and
// R = [part][peer][zl][ysc][nxs] = [part][nz][ysc][nxs]: the pieces travel one after the other on the communication
// stream while the z stage of the pieces that have arrived runs (x3d_sfft_fft_z / _postprocess_000 take a part).
//   --k_fft512 (y backward), loading from the exchange layout-->  C0  --rocFFT C2R x-->  f
// ~19 field passes per solve against ~30 for the generic pencil solver (pfft.hip: contiguous-axis rocFFT
// stages with pack / transpose passes around every exchange).
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

int x3d_fft512_init();
int x3d_fft512_run_x(x3d_backend *b, real2_t *c, int nxs, int ny, int nz, int axis, int mode, const real_t *waves,
                     const real_t *ab, int nx, real2_t *xbuf, int ys, int ysc);

int x3d_fft512_r2c(x3d_backend *b, real2_t *c, const real_t *f, long nrows, long frow, long crow);  // fft512.hip
int x3d_fft512_peers(x3d_backend *b, real2_t *R, long W, int npeers, const real_t *waves, const real_t *ab, int nx, int ny,
                     int nz, int nxs, int yoff, bool *done);  // fft512.hip

struct x3d_sfft {
    x3d_backend *b;
    int nx, ny, nz, nxs;  // global cell dims (ny = 512)
    int pz, rz, zl, ys;   // ranks along z, this rank, local z extent, this rank's share of the y modes
    int parts, ysc;       // the share is exchanged and z-transformed in `parts` pieces of ysc y modes (overlap)
    hipfftHandle plan_x_fw, plan_x_bw, plan_z;
    real2_t *c0;          // [zl][ny][nxs]
    real2_t *t;           // [ys*nxs][nz]: z-contiguous copy of the received array
    int fused_z;          // 512 local planes, pz in {1, 2, 4, 8}: the whole z stage is ONE kernel on the received
                          // array itself (fft512.hip, k_fft512_peers): no transposed copy, no rocFFT z plan calls
    real_t *waves, *ab;   // -1 / waves [ys][nxs][nz] (0 where waves < 1e-16); ax bx ay by az bz
    void *work;
};

// 32 x 32 tiles through LDS: src [nB][nA] (A contiguous) -> dst [nA][nB] (B contiguous), 512-byte rows both ways
__global__ void __launch_bounds__(256)
    k_sfft_transpose(real2_t *__restrict__ dst, const real2_t *__restrict__ src, int nA, int nB)
{
    __shared__ real2_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int bb = b0 + ty + 8 * r, aa = a0 + tx;
        if (aa < nA && bb < nB) tile[ty + 8 * r][tx] = src[(long)bb * nA + aa];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int aa = a0 + ty + 8 * r, bb = b0 + tx;
        if (aa < nA && bb < nB) dst[(long)aa * nB + bb] = tile[tx][ty + 8 * r];
    }
}

// process_spectral_000 (src/backend/omp/kernels/spectral_processing.f90:7-106) on T[ys][nxs][nz] (z fastest):
// y index offset = rz * ys (sp_st(2)), one thread per mode
__global__ void __launch_bounds__(256)
    k_process_spectral_000_slab(real2_t *__restrict__ c, const real_t *__restrict__ waves, int nxs, int ys, int nz,
                                int yoff, int nx, int ny, const real_t *__restrict__ ax, const real_t *__restrict__ bx,
                                const real_t *__restrict__ ay, const real_t *__restrict__ by,
                                const real_t *__restrict__ az, const real_t *__restrict__ bz)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, jl = blockIdx.z;
    if (k >= nz) return;
    const int j = jl + yoff;
    const size_t idx = ((size_t)jl * nxs + i) * nz + k;
    real2_t v = c[idx];
    const real_t rn = 1.0 / nx / ny / nz;
    real_t div_r = v.x * rn, div_c = v.y * rn;
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
    const real_t rw = waves[idx];  // (-1 / waves, x3d_sfft_set_waves)
    div_r = div_r * rw; div_c = div_c * rw;
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

extern "C" int x3d_sfft_create_parts(x3d_backend *b, x3d_sfft **out, const int nglob[3], int pz, int rz, int parts);
extern "C" int x3d_sfft_create(x3d_backend *b, x3d_sfft **out, const int nglob[3], int pz, int rz)
{
    X3D_RANGE(__func__);
    return x3d_sfft_create_parts(b, out, nglob, pz, rz, 1);
}

extern "C" int x3d_sfft_create_parts(x3d_backend *b, x3d_sfft **out, const int nglob[3], int pz, int rz, int parts)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && nglob, "x3d_sfft_create: null argument");
    X3D_REQUIRE(pz >= 1 && rz >= 0 && rz < pz, "x3d_sfft_create: bad rank grid");
    X3D_REQUIRE(nglob[1] == 512, "x3d_sfft_create: the slab solver needs ny = 512 (got %d)", nglob[1]);
    X3D_REQUIRE(nglob[2] % pz == 0 && 512 % pz == 0, "x3d_sfft_create: nz and ny must divide by pz");
    x3d_sfft *p = new x3d_sfft();
    memset(p, 0, sizeof *p);
    p->b = b;
    p->nx = nglob[0]; p->ny = nglob[1]; p->nz = nglob[2];
    {
        // spectral rows padded to 8 complex numbers (128 B) like the single-rank solver's: the 128 / 256-byte row
        // segments of the strided passes are line-aligned (pad columns hold zeros, their wave numbers ones)
        const char *e = getenv("X3D_NO_SPECTRAL_PAD");
        const int nxm = nglob[0] / 2 + 1;
        p->nxs = (e && e[0] == '1') ? nxm : (nxm + 7) / 8 * 8;
    }
    p->pz = pz; p->rz = rz; p->zl = p->nz / pz; p->ys = p->ny / pz;
    X3D_REQUIRE(parts >= 1 && p->ys % parts == 0, "x3d_sfft_create: %d y modes per rank do not split into %d parts",
                p->ys, parts);
    p->parts = parts; p->ysc = p->ys / parts;
    {
        const char *e = getenv("X3D_NO_SLAB_FUSED_Z");
        p->fused_z = p->zl == 512 && (pz == 1 || pz == 2 || pz == 4 || pz == 8) && !(e && e[0] == '1');
    }
    X3D_REQUIRE(p->nx <= b->nxp && p->ny == b->nyp && p->zl <= b->nzp, "x3d_sfft_create: local block mismatch");
    const size_t n0 = (size_t)p->zl * p->ny * p->nxs;
    X3D_HIP(hipMalloc(&p->c0, sizeof(real2_t) * n0));
    X3D_HIP(hipMemset(p->c0, 0, sizeof(real2_t) * n0));  // (the pad columns stay zero: the x transforms never write them)
    X3D_HIP(hipMalloc(&p->t, sizeof(real2_t) * (size_t)p->nz * p->ys * p->nxs));
    X3D_HIP(hipMalloc(&p->waves, sizeof(real_t) * (size_t)p->nz * p->ys * p->nxs));
    X3D_HIP(hipMalloc(&p->ab, sizeof(real_t) * 2 * ((size_t)p->nx + p->ny + p->nz)));
    int nn[1] = {p->nx}, re[1] = {b->nxp}, ce[1] = {p->nxs}, nzv[1] = {p->nz};
    const int batch = p->ny * p->zl, zstride = p->ysc * p->nxs;  // (z plan: one part at a time)
    hipfftHandle *pl[3] = {&p->plan_x_fw, &p->plan_x_bw, &p->plan_z};
    size_t ws[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) {
        X3D_FFT(hipfftCreate(pl[i]));
        X3D_FFT(hipfftSetAutoAllocation(*pl[i], 0));
    }
    X3D_FFT(hipfftMakePlanMany(p->plan_x_fw, 1, nn, re, 1, b->nxp, ce, 1, p->nxs, X3D_FFT_R2C, batch, &ws[0]));
    X3D_FFT(hipfftMakePlanMany(p->plan_x_bw, 1, nn, ce, 1, p->nxs, re, 1, b->nxp, X3D_FFT_C2R, batch, &ws[1]));
    // z transform on the z-contiguous copy T[ys*nxs][nz]
    X3D_FFT(hipfftMakePlanMany(p->plan_z, 1, nzv, nzv, 1, p->nz, nzv, 1, p->nz, X3D_FFT_C2C, zstride, &ws[2]));
    size_t wmax = 0;
    for (int i = 0; i < 3; i++) wmax = ws[i] > wmax ? ws[i] : wmax;
    if (wmax) X3D_HIP(hipMalloc(&p->work, wmax));
    for (int i = 0; i < 3; i++) X3D_FFT(hipfftSetWorkArea(*pl[i], p->work));
    if (int rc = x3d_fft512_init()) return rc;
    *out = p;
    return 0;
}

extern "C" int x3d_sfft_destroy(x3d_sfft *p)
{
    X3D_RANGE(__func__);
    if (!p) return 0;
    hipfftDestroy(p->plan_x_fw); hipfftDestroy(p->plan_x_bw); hipfftDestroy(p->plan_z);
    hipFree(p->c0); hipFree(p->t); hipFree(p->waves); hipFree(p->ab); hipFree(p->work);
    delete p;
    return 0;
}

// out = {chunk (complex elements per peer), zl, ys, nxs}
extern "C" int x3d_sfft_sizes(const x3d_sfft *p, long out[4])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && out, "null argument");
    out[0] = (long)p->zl * p->ys * p->nxs; out[1] = p->zl; out[2] = p->ys; out[3] = p->nxs;
    return 0;
}


// waves: this rank's spectral block [ys][nxs][nz], z fastest (real part = imaginary part); ax..bz: full arrays
extern "C" int x3d_sfft_set_waves(x3d_sfft *p, const real_t *waves, const real_t *ax, const real_t *bx,
                                  const real_t *ay, const real_t *by, const real_t *az, const real_t *bz)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && waves && ax && bx && ay && by && az && bz, "null argument");
    {
        // stored as -1 / waves (0 where waves < 1e-16): the kernels multiply (the reference divides per element; one
        // reciprocal here and one of nx ny nz there: <= 2 ulp apart, as in the single-rank solver)
        const size_t n = (size_t)p->nz * p->ys * p->nxs;
        std::vector<real_t> rw(n);
        for (size_t i = 0; i < n; i++) rw[i] = waves[i] < 1.e-16 ? 0.0 : -1.0 / waves[i];
        X3D_HIP(hipMemcpy(p->waves, rw.data(), sizeof(real_t) * n, hipMemcpyHostToDevice));
    }
    const real_t *src[6] = {ax, bx, ay, by, az, bz};
    const int len[6] = {p->nx, p->nx, p->ny, p->ny, p->nz, p->nz};
    real_t *d = p->ab;
    for (int i = 0; i < 6; i++) {
        X3D_HIP(hipMemcpy(d, src[i], sizeof(real_t) * len[i], hipMemcpyHostToDevice));
        d += len[i];
    }
    return 0;
}

// x R2C, y forward; the result lands in sendbuf as [peer][zl][ys][nxs]
extern "C" int x3d_sfft_forward_local(x3d_sfft *p, const real_t *f_in, real_t *sendbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_in && sendbuf, "null argument");
    X3D_LAZY_IN(p->b, f_in);  // (deferred execution: flush, then the buffer that holds the field)
    {
        ProfScope ps(p->b, X3D_K_FFT, 1);
        static int own = -1;  // single-kernel real-to-complex x pass (fft512.hip), as in the single-rank solver
        if (own < 0) { const char *e = getenv("X3D_NO_R2C512"); own = (e && e[0] == '1') ? 0 : 1; }
        const long rows = (long)p->zl * p->ny;
        if (own && p->nx == 512 && rows % 2 == 0) {
            if (int rc = x3d_fft512_r2c(p->b, p->c0, f_in, rows, p->b->nxp, p->nxs)) return rc;
        } else {
            X3D_FFT(hipfftSetStream(p->plan_x_fw, p->b->stream));
            X3D_FFT(x3d_fftExecR2C(p->plan_x_fw, (x3d_fft_real *)f_in, (x3d_fft_cplx *)p->c0));
        }
    }
    return x3d_fft512_run_x(p->b, p->c0, p->nxs, p->ny, p->zl, 1, 0, nullptr, nullptr, p->nx, (real2_t *)sendbuf,
                            p->ys, p->ysc);
}

// dir 0: part `part` of the received array, R_m[nz][ysc][nxs] -> T_m[ysc*nxs][nz], forward z transform (the
// spectrum stays in T); dir 1: backward z transform of T_m, then back to R_m
// fused_ok: the caller runs forward ; division ; backward of this part back to back (the *_part entry points of the
// pipelined solve): with fused_z the three are ONE kernel, launched by the division call.  The plain hooks
// (x3d_sfft_fft_z / x3d_sfft_postprocess_000 = fft_forward / fft_postprocess_000 / fft_backward of the reference,
// src/poisson_fft.f90:45-62) keep their own meaning at every grid size: after fft_forward the spectrum IS transformed
static int sfft_fft_z_part(x3d_sfft *p, real_t *recvbuf, int dir, int part, bool fused_ok)
{
    X3D_REQUIRE(p && recvbuf && part >= 0 && part < p->parts, "x3d_sfft_fft_z_part: bad argument");
    if (p->fused_z && fused_ok) return 0;  // (forward, division and backward are one kernel: sfft_postprocess_part)
    const int W = p->ysc * p->nxs;
    real2_t *R = (real2_t *)recvbuf + (size_t)part * p->nz * W, *T = p->t + (size_t)part * p->nz * W;
    if (dir == 0) {
        ProfScope ps(p->b, X3D_K_PACK);
        hipLaunchKernelGGL(k_sfft_transpose, dim3((W + 31) / 32, (p->nz + 31) / 32), dim3(256), 0, p->b->stream, T,
                           (const real2_t *)R, W, p->nz);
        X3D_HIP(hipGetLastError());
    }
    {
        ProfScope ps(p->b, X3D_K_FFT, 3);
        X3D_FFT(hipfftSetStream(p->plan_z, p->b->stream));
        X3D_FFT(x3d_fftExecC2C(p->plan_z, (x3d_fft_cplx *)T, (x3d_fft_cplx *)T,
                              dir ? HIPFFT_BACKWARD : HIPFFT_FORWARD));
    }
    if (dir == 1) {
        ProfScope ps(p->b, X3D_K_PACK);
        hipLaunchKernelGGL(k_sfft_transpose, dim3((p->nz + 31) / 32, (W + 31) / 32), dim3(256), 0, p->b->stream, R,
                           (const real2_t *)T, p->nz, W);
        X3D_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" int x3d_sfft_fft_z_part(x3d_sfft *p, real_t *recvbuf, int dir, int part)
{
    X3D_RANGE(__func__);
    return sfft_fft_z_part(p, recvbuf, dir, part, true);
}

extern "C" int x3d_sfft_fft_z(x3d_sfft *p, real_t *recvbuf, int dir)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    for (int m = 0; m < p->parts; m++)
        if (int rc = sfft_fft_z_part(p, recvbuf, dir, m, false)) return rc;
    return 0;
}

static int sfft_postprocess_part(x3d_sfft *p, real_t *recvbuf, int part, bool fused_ok)
{
    X3D_REQUIRE(p && recvbuf && part >= 0 && part < p->parts, "x3d_sfft_postprocess_000_part: bad argument");
    const real_t *ax = p->ab, *bx = ax + p->nx, *ay = bx + p->nx, *by = ay + p->ny, *az = by + p->ny,
                 *bz = az + p->nz;
    const size_t off = (size_t)part * p->nz * p->ysc * p->nxs;  // waves[ys][nxs][nz] and T share the part offset
    if (p->fused_z && fused_ok) {
        bool ok = false;
        ProfScope ps(p->b, X3D_K_FFT, 3);
        if (int rc = x3d_fft512_peers(p->b, (real2_t *)recvbuf + off, (long)p->ysc * p->nxs, p->pz, p->waves + off, p->ab,
                                      p->nx, p->ny, p->nz, p->nxs, p->rz * p->ys + part * p->ysc, &ok))
            return rc;
        X3D_REQUIRE(ok, "x3d_sfft_postprocess_000_part: fused z stage refused");
        return 0;
    }
    dim3 grid((p->nz + 255) / 256, p->nxs, p->ysc);
    ProfScope ps(p->b, X3D_K_SPECTRAL);
    hipLaunchKernelGGL(k_process_spectral_000_slab, grid, dim3(256), 0, p->b->stream, p->t + off, p->waves + off, p->nxs,
                       p->ysc, p->nz, p->rz * p->ys + part * p->ysc, p->nx, p->ny, ax, bx, ay, by, az, bz);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_sfft_postprocess_000_part(x3d_sfft *p, real_t *recvbuf, int part)
{
    X3D_RANGE(__func__);
    return sfft_postprocess_part(p, recvbuf, part, true);
}

extern "C" int x3d_sfft_postprocess_000(x3d_sfft *p, real_t *recvbuf)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf, "null argument");
    for (int m = 0; m < p->parts; m++)
        if (int rc = sfft_postprocess_part(p, recvbuf, m, false)) return rc;
    return 0;
}

// y backward from the exchange layout, x C2R
extern "C" int x3d_sfft_backward_local(x3d_sfft *p, const real_t *recvbuf, real_t *f_out)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && recvbuf && f_out, "null argument");
    // the whole real extent is written and nothing reads a block's padding: a block that still shares its buffer -- the
    // reference's p_temp is a reordered alias of div_u, released only behind the solve (src/solver.f90:653-678) -- takes a
    // free buffer instead of a copy of the old contents
    X3D_LAZY_OUT(p->b, f_out, true);
    if (int rc = x3d_fft512_run_x(p->b, p->c0, p->nxs, p->ny, p->zl, 1, 1, nullptr, nullptr, p->nx,
                                  (real2_t *)recvbuf, p->ys, p->ysc))
        return rc;
    ProfScope ps(p->b, X3D_K_FFT, 2);
    X3D_FFT(hipfftSetStream(p->plan_x_bw, p->b->stream));
    X3D_FFT(x3d_fftExecC2R(p->plan_x_bw, (x3d_fft_cplx *)p->c0, (x3d_fft_real *)f_out));
    return 0;
}
