// Spectral Poisson solver, all-periodic (000) case, one GPU:
// rocFFT (through the hipFFT API) 3-D D2Z / Z2D straight on the pitched
// Cartesian block + the spectral post-processing kernel.
//
// Reference behaviour mirrored (paths under /root/reference):
//   src/poisson_fft.f90:216-226                      poisson_000 = fwd ; postprocess ; bwd
//   src/backend/omp/poisson_fft.f90:89-97, 129-137   fft_forward/backward: unnormalised 3-D
//       r2c / c2r, forward e^{-i}, spectral array (nx/2+1, ny, nz)
//   src/backend/omp/kernels/spectral_processing.f90:7-106 process_spectral_000
//   (CUDA analogue: src/backend/cuda/poisson_fft.f90:618-777, kernels/spectral_processing.f90:127-224)
#include <hipfft/hipfft.h>

#include "common.h"

struct x3d_poisson {
    x3d_backend *b;
    int nx, ny, nz, nxs;  // cell dims, nxs = nx/2+1
    hipfftHandle plan_fw, plan_bw;
    double2 *c;           // spectral workspace [nz][ny][nxs]
    double *waves;        // [nz][ny][nxs]
    double *ab;           // ax bx ay by az bz
    void *work;
    size_t work_size;
};

#define X3D_FFT(expr)                                                                          \
    do {                                                                                       \
        hipfftResult r_ = (expr);                                                              \
        if (r_ != HIPFFT_SUCCESS) {                                                            \
            x3d_set_error("%s failed: hipfft error %d (%s:%d)", #expr, (int)r_, __FILE__,      \
                          __LINE__);                                                           \
            return 3;                                                                          \
        }                                                                                      \
    } while (0)

// One thread per spectral entry, x fastest -> coalesced 16 B per lane.
// 1R(c) + 1R(waves) + 1W(c): 40 B per complex entry.
__global__ void __launch_bounds__(256)
    k_process_spectral_000(double2 *__restrict__ c, const double *__restrict__ waves, int nxs, int ny, int nz,
                           int nx, const double *__restrict__ ax, const double *__restrict__ bx,
                           const double *__restrict__ ay, const double *__restrict__ by,
                           const double *__restrict__ az, const double *__restrict__ bz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // 0-based ix-1
    const int j = blockIdx.y, k = blockIdx.z;
    if (i >= nxs) return;
    const size_t idx = ((size_t)k * ny + j) * nxs + i;
    double2 v = c[idx];
    // normalisation (:36-37): the three divisions of the reference, same order
    double div_r = v.x / nx / ny / nz, div_c = v.y / nx / ny / nz;
    const double azk = az[k], bzk = bz[k], ayj = ay[j], byj = by[j], axi = ax[i], bxi = bx[i];
    const bool fz = (k + 1) > nz / 2 + 1, fy = (j + 1) > ny / 2 + 1;
    double tr, tc;
    tr = div_r; tc = div_c;                       // z forward (:46-51)
    div_r = tr * bzk + tc * azk;
    div_c = tc * bzk - tr * azk;
    if (fz) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;                       // y forward (:54-59)
    div_r = tr * byj + tc * ayj;
    div_c = tc * byj - tr * ayj;
    if (fy) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;                       // x forward (:62-65)
    div_r = tr * bxi + tc * axi;
    div_c = tc * bxi - tr * axi;
    const double wv = waves[idx];                 // real part == imaginary part (:68-76)
    if (wv < 1.e-16) { div_r = 0.0; div_c = 0.0; }
    else { div_r = -div_r / wv; div_c = -div_c / wv; }
    tr = div_r; tc = div_c;                       // z backward (:80-85)
    div_r = tr * bzk - tc * azk;
    div_c = -tc * bzk - tr * azk;
    if (fz) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;                       // y backward (:88-93)
    div_r = tr * byj + tc * ayj;
    div_c = tc * byj - tr * ayj;
    if (fy) { div_r = -div_r; div_c = -div_c; }
    tr = div_r; tc = div_c;                       // x backward (:96-99)
    div_r = tr * bxi + tc * axi;
    div_c = -tc * bxi + tr * axi;
    c[idx] = make_double2(div_r, div_c);
}

extern "C" int x3d_poisson_create(x3d_backend *b, x3d_poisson **out, const int n[3], const double *waves_re,
                                  const double *ax, const double *bx, const double *ay, const double *by,
                                  const double *az, const double *bz)
{
    X3D_REQUIRE(b && out && n && waves_re && ax && bx && ay && by && az && bz,
                "x3d_poisson_create: null argument");
    X3D_REQUIRE(n[0] <= b->nxp && n[1] <= b->nyp && n[2] <= b->nzp, "x3d_poisson_create: dims exceed block");
    x3d_poisson *p = new x3d_poisson();
    memset(p, 0, sizeof *p);
    p->b = b;
    p->nx = n[0]; p->ny = n[1]; p->nz = n[2]; p->nxs = n[0] / 2 + 1;
    const size_t ns = (size_t)p->nz * p->ny * p->nxs;
    X3D_HIP(hipMalloc(&p->c, sizeof(double2) * ns));
    X3D_HIP(hipMalloc(&p->waves, sizeof(double) * ns));
    X3D_HIP(hipMemcpy(p->waves, waves_re, sizeof(double) * ns, hipMemcpyHostToDevice));
    const size_t nab = 2 * ((size_t)n[0] + n[1] + n[2]);
    X3D_HIP(hipMalloc(&p->ab, sizeof(double) * nab));
    double *d = p->ab;
    const double *src[6] = {ax, bx, ay, by, az, bz};
    const int len[6] = {n[0], n[0], n[1], n[1], n[2], n[2]};
    for (int i = 0; i < 6; i++) {
        X3D_HIP(hipMemcpy(d, src[i], sizeof(double) * len[i], hipMemcpyHostToDevice));
        d += len[i];
    }
    // real side lives in the pitched block: embed = {nzp, nyp, nxp}; spectral side dense
    int dims[3] = {p->nz, p->ny, p->nx};
    int rembed[3] = {b->nzp, b->nyp, b->nxp};
    int cembed[3] = {p->nz, p->ny, p->nxs};
    X3D_FFT(hipfftCreate(&p->plan_fw));
    X3D_FFT(hipfftCreate(&p->plan_bw));
    X3D_FFT(hipfftSetAutoAllocation(p->plan_fw, 0));
    X3D_FFT(hipfftSetAutoAllocation(p->plan_bw, 0));
    size_t ws_fw = 0, ws_bw = 0;
    X3D_FFT(hipfftMakePlanMany(p->plan_fw, 3, dims, rembed, 1, (int)b->nblock, cembed, 1, (int)ns, HIPFFT_D2Z, 1,
                               &ws_fw));
    X3D_FFT(hipfftMakePlanMany(p->plan_bw, 3, dims, cembed, 1, (int)ns, rembed, 1, (int)b->nblock, HIPFFT_Z2D, 1,
                               &ws_bw));
    p->work_size = ws_fw > ws_bw ? ws_fw : ws_bw;
    if (p->work_size) X3D_HIP(hipMalloc(&p->work, p->work_size));
    X3D_FFT(hipfftSetWorkArea(p->plan_fw, p->work));
    X3D_FFT(hipfftSetWorkArea(p->plan_bw, p->work));
    *out = p;
    return 0;
}

extern "C" int x3d_poisson_destroy(x3d_poisson *p)
{
    if (!p) return 0;
    hipfftDestroy(p->plan_fw);
    hipfftDestroy(p->plan_bw);
    hipFree(p->c); hipFree(p->waves); hipFree(p->ab); hipFree(p->work);
    delete p;
    return 0;
}

extern "C" int x3d_poisson_fft_forward(x3d_poisson *p, const double *f_in)
{
    X3D_REQUIRE(p && f_in, "x3d_poisson_fft_forward: null argument");
    ProfScope ps(p->b, X3D_K_FFT, 1);
    X3D_FFT(hipfftSetStream(p->plan_fw, p->b->stream));
    X3D_FFT(hipfftExecD2Z(p->plan_fw, (hipfftDoubleReal *)f_in, (hipfftDoubleComplex *)p->c));
    return 0;
}

extern "C" int x3d_poisson_postprocess_000(x3d_poisson *p)
{
    X3D_REQUIRE(p, "x3d_poisson_postprocess_000: null argument");
    const double *ax = p->ab, *bx = ax + p->nx, *ay = bx + p->nx, *by = ay + p->ny, *az = by + p->ny,
                 *bz = az + p->nz;
    dim3 grid((p->nxs + 255) / 256, p->ny, p->nz);
    ProfScope ps(p->b, X3D_K_SPECTRAL);
    hipLaunchKernelGGL(k_process_spectral_000, grid, dim3(256), 0, p->b->stream, p->c, p->waves, p->nxs, p->ny,
                       p->nz, p->nx, ax, bx, ay, by, az, bz);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_poisson_fft_backward(x3d_poisson *p, double *f_out)
{
    X3D_REQUIRE(p && f_out, "x3d_poisson_fft_backward: null argument");
    ProfScope ps(p->b, X3D_K_FFT, 2);
    X3D_FFT(hipfftSetStream(p->plan_bw, p->b->stream));
    X3D_FFT(hipfftExecZ2D(p->plan_bw, (hipfftDoubleComplex *)p->c, (hipfftDoubleReal *)f_out));
    return 0;
}

extern "C" int x3d_poisson_solve_000(x3d_poisson *p, double *f)
{
    if (int rc = x3d_poisson_fft_forward(p, f)) return rc;
    if (int rc = x3d_poisson_postprocess_000(p)) return rc;
    return x3d_poisson_fft_backward(p, f);
}

extern "C" int x3d_poisson_get_spectral(x3d_poisson *p, double *host)
{
    X3D_REQUIRE(p && host, "null argument");
    const size_t ns = (size_t)p->nz * p->ny * p->nxs;
    X3D_HIP(hipStreamSynchronize(p->b->stream));
    X3D_HIP(hipMemcpy(host, p->c, sizeof(double2) * ns, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int x3d_poisson_set_spectral(x3d_poisson *p, const double *host)
{
    X3D_REQUIRE(p && host, "null argument");
    const size_t ns = (size_t)p->nz * p->ny * p->nxs;
    X3D_HIP(hipStreamSynchronize(p->b->stream));
    X3D_HIP(hipMemcpy(p->c, host, sizeof(double2) * ns, hipMemcpyHostToDevice));
    return 0;
}
