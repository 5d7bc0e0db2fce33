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

#include "poisson_priv.h"
int x3d_proxy_hook(x3d_poisson *p, int which, real_t *f);  // zfirst.hip

int x3d_fft512_init();
int x3d_fft512_run(x3d_backend *b, real2_t *c, int nxs, int ny, int nz, int axis, int mode, const real_t *waves,
                   const real_t *ab, int nx);

#define X3D_FFT(expr)                                                                          \
    do {                                                                                       \
        hipfftResult r_ = (expr);                                                              \
        if (r_ != HIPFFT_SUCCESS) {                                                            \
            x3d_set_error("%s failed: hipfft error %d (%s:%d)", #expr, (int)r_, __FILE__,      \
                          __LINE__);                                                           \
            return 3;                                                                          \
        }                                                                                      \
    } while (0)

// pad columns [nxm, nxs) of a pitched array of doubles <- v
__global__ void k_fill_pad(real_t *__restrict__ a, size_t rows, int nxm, int nxs, real_t v)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (int i = nxm; i < nxs; i++) a[r * nxs + i] = v;
}

// dense host rows [rows][nxm] -> pitched device rows [rows][nxs] (elem: bytes per element), pads = 0
static int upload_pitched(void *dst, const void *src, size_t rows, int nxm, int nxs, size_t elem)
{
    X3D_HIP(hipMemset(dst, 0, rows * nxs * elem));
    X3D_HIP(hipMemcpy2D(dst, nxs * elem, src, nxm * elem, nxm * elem, rows, hipMemcpyHostToDevice));
    return 0;
}

// One thread per spectral entry, x fastest -> coalesced 16 B per lane.
// 1R(c) + 1R(waves) + 1W(c): 40 B per complex entry.
__global__ void __launch_bounds__(256)
    k_process_spectral_000(real2_t *__restrict__ c, const real_t *__restrict__ waves, int nxs, int ny, int nz,
                           int nx, const real_t *__restrict__ ax, const real_t *__restrict__ bx,
                           const real_t *__restrict__ ay, const real_t *__restrict__ by,
                           const real_t *__restrict__ az, const real_t *__restrict__ bz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // 0-based ix-1
    const int j = blockIdx.y, k = blockIdx.z;
    if (i >= nxs) return;
    const size_t idx = ((size_t)k * ny + j) * nxs + i;
    real2_t v = c[idx];
    // normalisation (:36-37): the three divisions of the reference, same order
    real_t div_r = v.x / nx / ny / nz, div_c = v.y / nx / ny / nz;
    const real_t azk = az[k], bzk = bz[k], ayj = ay[j], byj = by[j], axi = ax[i], bxi = bx[i];
    const bool fz = (k + 1) > nz / 2 + 1, fy = (j + 1) > ny / 2 + 1;
    real_t tr, tc;
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
    const real_t wv = waves[idx];                 // real part == imaginary part (:68-76)
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
    c[idx] = make_real2(div_r, div_c);
}

extern "C" int x3d_poisson_create(x3d_backend *b, x3d_poisson **out, const int n[3], const real_t *waves_re,
                                  const real_t *ax, const real_t *bx, const real_t *ay, const real_t *by,
                                  const real_t *az, const real_t *bz)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && n && waves_re && ax && bx && ay && by && az && bz,
                "x3d_poisson_create: null argument");
    X3D_REQUIRE(n[0] <= b->nxp && n[1] <= b->nyp && n[2] <= b->nzp, "x3d_poisson_create: dims exceed block");
    x3d_poisson *p = new x3d_poisson();
    memset(p, 0, sizeof *p);
    p->b = b;
    p->nx = n[0]; p->ny = n[1]; p->nz = n[2]; p->nxm = n[0] / 2 + 1;
    {
        const char *e = getenv("X3D_NO_SPECTRAL_PAD");  // (A/B: dense rows as in round 1)
        p->nxs = (e && e[0] == '1') ? p->nxm : (p->nxm + 7) / 8 * 8;
    }
    const size_t rows = (size_t)p->nz * p->ny, ns = rows * p->nxs;
    {   // (room for the z-first layout of the same spectrum too: [nz / 2 + 1][ny][nx + 16], csrc/zfirst.hip)
        const size_t nzf = (size_t)(p->nz / 2 + 1) * p->ny * (p->nx + 16);
        p->c_elems = ns > nzf ? ns : nzf;
    }
    X3D_HIP(hipMalloc(&p->c, sizeof(real2_t) * p->c_elems));
    X3D_HIP(hipMemset(p->c, 0, sizeof(real2_t) * p->c_elems));
    X3D_HIP(hipMalloc(&p->waves, sizeof(real_t) * ns));
    if (int rc = upload_pitched(p->waves, waves_re, rows, p->nxm, p->nxs, sizeof(real_t))) return rc;
    if (p->nxs > p->nxm) {
        hipLaunchKernelGGL(k_fill_pad, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, p->waves, rows, p->nxm,
                           p->nxs, 1.0);
        X3D_HIP(hipDeviceSynchronize());
    }
    const size_t nab = 2 * ((size_t)n[0] + n[1] + n[2]);
    X3D_HIP(hipMalloc(&p->ab, sizeof(real_t) * nab));
    real_t *d = p->ab;
    const real_t *src[6] = {ax, bx, ay, by, az, bz};
    const int len[6] = {n[0], n[0], n[1], n[1], n[2], n[2]};
    for (int i = 0; i < 6; i++) {
        X3D_HIP(hipMemcpy(d, src[i], sizeof(real_t) * len[i], hipMemcpyHostToDevice));
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
    X3D_FFT(hipfftMakePlanMany(p->plan_fw, 3, dims, rembed, 1, (int)b->nblock, cembed, 1, (int)ns, X3D_FFT_R2C, 1,
                               &ws_fw));
    X3D_FFT(hipfftMakePlanMany(p->plan_bw, 3, dims, cembed, 1, (int)ns, rembed, 1, (int)b->nblock, X3D_FFT_C2R, 1,
                               &ws_bw));
    p->work_size = ws_fw > ws_bw ? ws_fw : ws_bw;
    if (p->work_size) X3D_HIP(hipMalloc(&p->work, p->work_size));
    X3D_FFT(hipfftSetWorkArea(p->plan_fw, p->work));
    X3D_FFT(hipfftSetWorkArea(p->plan_bw, p->work));
    const char *no512 = getenv("X3D_NO_FFT512");
    if (p->ny == 512 && p->nz == 512 && !(no512 && no512[0] == '1')) {
        // batched 1-D r2c / c2r along x straight on the pitched block (rows nxp apart -> nxs complex)
        int nn[1] = {p->nx}, re[1] = {b->nxp}, ce[1] = {p->nxs};
        const int batch = p->ny * p->nz;
        X3D_REQUIRE(b->nyp == p->ny, "x3d_poisson_create: fast path needs nyp == ny");
        X3D_FFT(hipfftPlanMany(&p->plan_x_fw, 1, nn, re, 1, b->nxp, ce, 1, p->nxs, X3D_FFT_R2C, batch));
        X3D_FFT(hipfftPlanMany(&p->plan_x_bw, 1, nn, ce, 1, p->nxs, re, 1, b->nxp, X3D_FFT_C2R, batch));
        if (int rc = x3d_fft512_init()) return rc;
        p->fast512 = 1;
        {   // z-fastest reciprocal wave numbers for the fused z pass (spectral division in the transform's layout)
            const char *no_rwt = getenv("X3D_NO_RWT");
            if (!(no_rwt && no_rwt[0] == '1')) {
                const size_t nsr = (size_t)p->nz * p->ny * p->nxs;
                std::vector<real_t> h(nsr, 0.0);  // (pad columns: 0)
                for (int k = 0; k < p->nz; k++)
                    for (int j = 0; j < p->ny; j++)
                        for (int i = 0; i < p->nxm; i++) {
                            const real_t wv = waves_re[((size_t)k * p->ny + j) * p->nxm + i];
                            h[((size_t)j * p->nxs + i) * p->nz + k] = wv < 1.e-16 ? 0.0 : -1.0 / wv;
                        }
                X3D_HIP(hipMalloc(&p->rwT, sizeof(real_t) * nsr));
                X3D_HIP(hipMemcpy(p->rwT, h.data(), sizeof(real_t) * nsr, hipMemcpyHostToDevice));
            }
        }
        const char *no_r2c = getenv("X3D_NO_R2C512");
        p->r2c512 = p->nx == 512 && (batch % 2) == 0 && !(no_r2c && no_r2c[0] == '1');
    }
    *out = p;
    return 0;
}

extern "C" int x3d_poisson_destroy(x3d_poisson *p)
{
    X3D_RANGE(__func__);
    if (!p) return 0;
    if (p->ext_middle) { delete p; return 0; }  // (a proxy owns nothing)
    hipfftDestroy(p->plan_fw);
    hipfftDestroy(p->plan_bw);
    if (p->fast512) { hipfftDestroy(p->plan_x_fw); hipfftDestroy(p->plan_x_bw); }
    if (p->y010 == 1) { hipfftDestroy(p->plan_x010_fw); hipfftDestroy(p->plan_x010_bw); }
    hipFree(p->c); hipFree(p->waves); hipFree(p->ab); hipFree(p->work); hipFree(p->rwT); hipFree(p->rwZ);
    hipFree(p->lu[0]); hipFree(p->lu[1]); hipFree(p->luz[0]); hipFree(p->luz[1]);
    delete p;
    return 0;
}

int x3d_fft512_r2c(x3d_backend *b, real2_t *c, const real_t *f, long nrows, long frow, long crow);
void x3d_fft512_set_rwT(const real_t *rwT);

// x pass of the fast path: real rows (nxp apart) -> nxs complex modes
static int x_forward_512(x3d_poisson *p, const real_t *f)
{
    ProfScope ps(p->b, X3D_K_FFT, 1);
    if (p->r2c512)
        return x3d_fft512_r2c(p->b, (real2_t *)p->c, f, (long)p->ny * p->nz, p->b->nxp, p->nxs);
    X3D_FFT(hipfftSetStream(p->plan_x_fw, p->b->stream));
    X3D_FFT(x3d_fftExecR2C(p->plan_x_fw, (x3d_fft_real *)f, (x3d_fft_cplx *)p->c));
    return 0;
}

extern "C" int x3d_poisson_fft_forward(x3d_poisson *p, const real_t *f_in)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_in, "x3d_poisson_fft_forward: null argument");
    if (x3d_lazy_active(p->b)) return x3d_lazy_fft(p->b, 0, p, const_cast<real_t *>(f_in));
    if (p->ext_middle) return x3d_proxy_hook(p, 0, const_cast<real_t *>(f_in));
    if (p->fast512) {
        if (int rc = x_forward_512(p, f_in)) return rc;
        if (int rc = x3d_fft512_run(p->b, p->c, p->nxs, p->ny, p->nz, 1, 0, nullptr, nullptr, p->nx)) return rc;
        return x3d_fft512_run(p->b, p->c, p->nxs, p->ny, p->nz, 2, 0, nullptr, nullptr, p->nx);
    }
    ProfScope ps(p->b, X3D_K_FFT, 1);
    X3D_FFT(hipfftSetStream(p->plan_fw, p->b->stream));
    X3D_FFT(x3d_fftExecR2C(p->plan_fw, (x3d_fft_real *)f_in, (x3d_fft_cplx *)p->c));
    return 0;
}

extern "C" int x3d_poisson_postprocess_000(x3d_poisson *p)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p, "x3d_poisson_postprocess_000: null argument");
    if (x3d_lazy_active(p->b)) return x3d_lazy_fft(p->b, 1, p, nullptr);
    if (p->ext_middle) return x3d_proxy_hook(p, 1, nullptr);
    const real_t *ax = p->ab, *bx = ax + p->nx, *ay = bx + p->nx, *by = ay + p->ny, *az = by + p->ny,
                 *bz = az + p->nz;
    dim3 grid((p->nxs + 255) / 256, p->ny, p->nz);
    ProfScope ps(p->b, X3D_K_SPECTRAL);
    hipLaunchKernelGGL(k_process_spectral_000, grid, dim3(256), 0, p->b->stream, p->c, p->waves, p->nxs, p->ny,
                       p->nz, p->nx, ax, bx, ay, by, az, bz);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_poisson_fft_backward(x3d_poisson *p, real_t *f_out)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out, "x3d_poisson_fft_backward: null argument");
    if (x3d_lazy_active(p->b)) return x3d_lazy_fft(p->b, 2, p, f_out);
    if (p->ext_middle) return x3d_proxy_hook(p, 2, f_out);
    if (p->fast512) {
        if (int rc = x3d_fft512_run(p->b, p->c, p->nxs, p->ny, p->nz, 2, 1, nullptr, nullptr, p->nx)) return rc;
        if (int rc = x3d_fft512_run(p->b, p->c, p->nxs, p->ny, p->nz, 1, 1, nullptr, nullptr, p->nx)) return rc;
        ProfScope ps(p->b, X3D_K_FFT, 2);
        X3D_FFT(hipfftSetStream(p->plan_x_bw, p->b->stream));
        X3D_FFT(x3d_fftExecC2R(p->plan_x_bw, (x3d_fft_cplx *)p->c, (x3d_fft_real *)f_out));
        return 0;
    }
    ProfScope ps(p->b, X3D_K_FFT, 2);
    X3D_FFT(hipfftSetStream(p->plan_bw, p->b->stream));
    X3D_FFT(x3d_fftExecC2R(p->plan_bw, (x3d_fft_cplx *)p->c, (x3d_fft_real *)f_out));
    return 0;
}

extern "C" int x3d_poisson_solve_000(x3d_poisson *p, real_t *f)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f, "x3d_poisson_solve_000: null argument");
    X3D_LAZY_OUT(p->b, f, false);
    X3D_LAZY_EAGER(p->b);
    if (p->ext_middle) {  // a proxy: z ; the caller's middle ; z
        if (int rc = x3d_proxy_hook(p, 0, f)) return rc;
        if (int rc = x3d_proxy_hook(p, 1, nullptr)) return rc;
        return x3d_proxy_hook(p, 2, f);
    }
    if (p->fast512) {  // x r2c ; y ; z forward + process_spectral_000 + z backward in one pass ; y ; x c2r
        if (int rc = x_forward_512(p, f)) return rc;
        if (int rc = x3d_fft512_run(p->b, p->c, p->nxs, p->ny, p->nz, 1, 0, nullptr, nullptr, p->nx)) return rc;
        x3d_fft512_set_rwT(p->rwT);
        if (int rc = x3d_fft512_run(p->b, p->c, p->nxs, p->ny, p->nz, 2, 2, p->waves, p->ab, p->nx)) return rc;
        x3d_fft512_set_rwT(nullptr);
        if (int rc = x3d_fft512_run(p->b, p->c, p->nxs, p->ny, p->nz, 1, 1, nullptr, nullptr, p->nx)) return rc;
        ProfScope ps(p->b, X3D_K_FFT, 2);
        X3D_FFT(hipfftSetStream(p->plan_x_bw, p->b->stream));
        X3D_FFT(x3d_fftExecC2R(p->plan_x_bw, (x3d_fft_cplx *)p->c, (x3d_fft_real *)f));
        return 0;
    }
    if (int rc = x3d_poisson_fft_forward(p, f)) return rc;
    if (int rc = x3d_poisson_postprocess_000(p)) return rc;
    return x3d_poisson_fft_backward(p, f);
}

// ---------------------------------------------------------------------------
// Non-periodic y (010).  Reference behaviour mirrored:
//   src/poisson_fft.f90:228-242                         poisson_010
//   src/backend/cuda/kernels/spectral_processing.f90:1062-1114 enforce / undo_periodicity_y
//   src/backend/omp/kernels/spectral_processing.f90:108-283    process_spectral_010
//   src/backend/cuda/kernels/spectral_processing.f90:385-702   _010_fw / _010_poisson / _010_bw
//   src/backend/cuda/poisson_fft.f90:822-924                   fft_postprocess_010
// ---------------------------------------------------------------------------

#include "spectral010.h"

// Poisson 110, the middle of fft_postprocess_110 (src/backend/cuda/poisson_fft.f90:926-989) in the layout of the
// z-first transposed problem c[z'][y'][x'] (x' = z: R2C; y' = x, z' = y: both non-periodic): paired split along
// z' (process_spectral_110_y_pair_fw), division by the wave numbers with the Nyquist line zeroed (_poisson),
// recombination (_y_pair_bw) -- src/backend/cuda/kernels/spectral_processing.f90:803-931.  All three only couple
// planes k and nz-k+2: one thread per (x' mode, y' index, plane pair).
__global__ void __launch_bounds__(256)
    k_spectral_pair_z(real2_t *__restrict__ c, const real_t *__restrict__ waves, int nxs, int ny, int nz, int nx,
                      const real_t *__restrict__ az, const real_t *__restrict__ bz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, k = blockIdx.z + 1;  // k = 1 .. nz/2+1
    if (i >= nxs) return;
    const int kr = nz - k + 2;
    const bool paired = k >= 2, self = paired && kr == k;
    const size_t il = ((size_t)(k - 1) * ny + j) * nxs + i;
    const size_t ir = paired ? ((size_t)(kr - 1) * ny + j) * nxs + i : il;
    const real2_t L = c[il], R = paired && !self ? c[ir] : L;
    real_t l_r = L.x, l_c = L.y, r_r = R.x, r_c = R.y;
    const real_t a = az[k - 1], b = bz[k - 1], a2 = paired ? az[kr - 1] : 0.0, b2 = paired ? bz[kr - 1] : 0.0;
    if (paired) {
        const real_t n_lr = 0.5 * (l_r * b + l_c * a + r_r * b - r_c * a);
        const real_t n_lc = 0.5 * (-l_r * a + l_c * b + r_r * a + r_c * b);
        const real_t n_rr = 0.5 * (r_r * b2 + r_c * a2 + l_r * b2 - l_c * a2);
        const real_t n_rc = 0.5 * (-r_r * a2 + r_c * b2 + l_r * a2 + l_c * b2);
        l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
        if (self) { l_r = r_r; l_c = r_c; }  // the second store wins on the self-paired plane
    }
    {
        // (the reference zeroes the mode that is Nyquist in its x and z: here y' and x')
        const bool zero_line = (j + 1) == ny / 2 + 1 && (i + 1) == nx / 2 + 1;
        const real_t wl = waves[il];
        l_r = fabs(wl) < 1.e-16 ? 0.0 : -l_r / wl;
        l_c = fabs(wl) < 1.e-16 ? 0.0 : -l_c / wl;
        if (zero_line) { l_r = 0.0; l_c = 0.0; }
        if (paired) {
            const real_t wr = waves[ir];
            r_r = fabs(wr) < 1.e-16 ? 0.0 : -r_r / wr;
            r_c = fabs(wr) < 1.e-16 ? 0.0 : -r_c / wr;
            if (zero_line) { r_r = 0.0; r_c = 0.0; }
        }
    }
    if (paired) {
        if (self) { r_r = l_r; r_c = l_c; }
        const real_t n_lr = l_r * b - l_c * a + r_r * a + r_c * b;
        const real_t n_lc = l_r * a + l_c * b - r_r * b + r_c * a;
        const real_t n_rr = r_r * b2 - r_c * a2 + l_r * a2 + l_c * b2;
        const real_t n_rc = r_r * a2 + r_c * b2 - l_r * b2 + l_c * a2;
        l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
        if (self) { l_r = r_r; l_c = r_c; }
    }
    c[il] = make_real2(l_r, l_c);
    if (paired && !self) c[ir] = make_real2(r_r, r_c);
}

// even/odd interleave along z on the pitched Cartesian block (the y part of enforce_periodicity_xy / undo_..., :1116-1196,
// in the z-first transposed problem): out(i, j, k) = in(i, j, src(k))
template <bool UNDO>
__global__ void __launch_bounds__(256)
    k_periodicity_z(real_t *__restrict__ out, const real_t *__restrict__ in, int nx, int ny, int nz, long nxp, long plane)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, k = blockIdx.z + 1;
    if (i >= nx) return;
    const int n2 = nz / 2;
    int ks, kd;
    if (!UNDO) {
        kd = k;
        if (k <= n2) ks = 2 * k - 1;
        else if ((nz & 1) && k == n2 + 1) ks = nz;
        else ks = 2 * nz - 2 * k + 2;
    } else {
        ks = k;
        if (k <= n2) kd = 2 * k - 1;
        else if ((nz & 1) && k == n2 + 1) kd = nz;
        else kd = 2 * (nz - k + 1);
    }
    out[(long)(kd - 1) * plane + (long)j * nxp + i] = in[(long)(ks - 1) * plane + (long)j * nxp + i];
}

extern "C" int x3d_poisson_enforce_periodicity_y(x3d_poisson *p, real_t *f_out, const real_t *f_in)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out && f_in && f_out != f_in, "x3d_poisson_enforce_periodicity_y: bad argument");
    X3D_LAZY_IN(p->b, f_in);
    X3D_LAZY_OUT(p->b, f_out, true);
    X3D_LAZY_EAGER(p->b);
    x3d_backend *b = p->b;
    dim3 grid((p->nx + 255) / 256, p->ny, p->nz);
    ProfScope ps(b, X3D_K_COPY);
    hipLaunchKernelGGL(k_periodicity_y<false>, grid, dim3(256), 0, b->stream, f_out, f_in, p->nx, p->ny, p->nz,
                       (long)b->nxp, (long)b->nxp * b->nyp);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_poisson_undo_periodicity_y(x3d_poisson *p, real_t *f_out, const real_t *f_in)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out && f_in && f_out != f_in, "x3d_poisson_undo_periodicity_y: bad argument");
    X3D_LAZY_IN(p->b, f_in);
    X3D_LAZY_OUT(p->b, f_out, true);
    X3D_LAZY_EAGER(p->b);
    x3d_backend *b = p->b;
    dim3 grid((p->nx + 255) / 256, p->ny, p->nz);
    ProfScope ps(b, X3D_K_COPY);
    hipLaunchKernelGGL(k_periodicity_y<true>, grid, dim3(256), 0, b->stream, f_out, f_in, p->nx, p->ny, p->nz,
                       (long)b->nxp, (long)b->nxp * b->nyp);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_poisson_enforce_periodicity_z(x3d_poisson *p, real_t *f_out, const real_t *f_in)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out && f_in && f_out != f_in, "x3d_poisson_enforce_periodicity_z: bad argument");
    X3D_LAZY_SYNC(p->b);
    X3D_LAZY_EAGER(p->b);
    x3d_backend *b = p->b;
    dim3 grid((p->nx + 255) / 256, p->ny, p->nz);
    ProfScope ps(b, X3D_K_COPY);
    hipLaunchKernelGGL(k_periodicity_z<false>, grid, dim3(256), 0, b->stream, f_out, f_in, p->nx, p->ny, p->nz,
                       (long)b->nxp, (long)b->nxp * b->nyp);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_poisson_undo_periodicity_z(x3d_poisson *p, real_t *f_out, const real_t *f_in)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out && f_in && f_out != f_in, "x3d_poisson_undo_periodicity_z: bad argument");
    X3D_LAZY_SYNC(p->b);
    X3D_LAZY_EAGER(p->b);
    x3d_backend *b = p->b;
    dim3 grid((p->nx + 255) / 256, p->ny, p->nz);
    ProfScope ps(b, X3D_K_COPY);
    hipLaunchKernelGGL(k_periodicity_z<true>, grid, dim3(256), 0, b->stream, f_out, f_in, p->nx, p->ny, p->nz,
                       (long)b->nxp, (long)b->nxp * b->nyp);
    X3D_HIP(hipGetLastError());
    return 0;
}

// fft_postprocess_110 (src/backend/cuda/poisson_fft.f90:926-989) on the z-first transposed problem: this plan's x is
// the reference's z (periodic, R2C), its y the reference's x, its z the reference's y.  The reference's seven
// launches as three: normalisation + rotation along x' + paired split along y' ; paired split along z' +
// division + recombination along z' ; recombination along y' + inverse rotation along x'.
extern "C" int x3d_poisson_postprocess_011(x3d_poisson *p)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p, "x3d_poisson_postprocess_011: null argument");
    X3D_LAZY_FLUSH(p->b);
    X3D_LAZY_EAGER(p->b);
    X3D_REQUIRE(!p->stretched, "x3d_poisson_postprocess_011: uniform grids only");
    const real_t *ax = p->ab, *bx = ax + p->nx, *ay = bx + p->nx, *by = ay + p->ny, *az = by + p->ny,
                 *bz = az + p->nz;
    ProfScope ps(p->b, X3D_K_SPECTRAL);
    hipStream_t st = p->b->stream;
    dim3 gy = spectral_010_grid(p->nxs, p->ny, p->nz), gz((p->nxs + 255) / 256, p->ny, p->nz / 2 + 1);
    hipLaunchKernelGGL((k_spectral_010<0, false>), gy, dim3(256), 0, st, p->c, p->waves, p->nxs, p->ny, p->nz, p->nx, 0, ax,
                       bx, ay, by, az, bz);
    hipLaunchKernelGGL(k_spectral_pair_z, gz, dim3(256), 0, st, p->c, p->waves, p->nxs, p->ny, p->nz, p->nx, az, bz);
    hipLaunchKernelGGL((k_spectral_010<1, false>), gy, dim3(256), 0, st, p->c, p->waves, p->nxs, p->ny, p->nz, p->nx, 0, ax,
                       bx, ay, by, az, bz);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_poisson_set_stretching(x3d_poisson *p, int sym, const real_t *a0, const real_t *a1)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && a0 && (!sym || a1), "x3d_poisson_set_stretching: null argument");
    X3D_REQUIRE(!sym || p->ny % 2 == 0, "x3d_poisson_set_stretching: odd/even split needs an even ny");
    const int n = sym ? p->ny / 2 : p->ny;
    X3D_REQUIRE(n >= 3, "x3d_poisson_set_stretching: too few rows");
    const size_t rows = 5 * (size_t)p->nz * n, bytes = sizeof(real_t) * rows * p->nxs;
    const real_t *src[2] = {a0, a1};
    for (int s = 0; s < (sym ? 2 : 1); s++) {
        if (!p->lu[s]) X3D_HIP(hipMalloc(&p->lu[s], bytes));
        // (pad columns all zero: k_penta_factor / k_penta_solve guard their divisions by |a3| > eps)
        if (int rc = upload_pitched(p->lu[s], src[s], rows, p->nxm, p->nxs, sizeof(real_t))) return rc;
        hipLaunchKernelGGL(k_penta_factor<false>, penta_grid(p->nxs, p->nz), dim3(64), 0, p->b->stream, p->lu[s], p->nxs, n,
                           p->nz);
        X3D_HIP(hipGetLastError());
    }
    X3D_HIP(hipStreamSynchronize(p->b->stream));
    p->stretched = 1;
    p->sym = sym;
    return 0;
}

// the same operators for the z-first form of the solve (csrc/zfirst.hip): a0 / a1 = [5][nz / 2 + 1][n][nx] dense -- every
// x mode (the wave numbers above nx / 2 are the mirrored ones, as the reference's are along z, src/poisson_fft.f90:833-882),
// the z modes 0 .. nz / 2.  Factored once like x3d_poisson_set_stretching's, stored at row pitch nx + 16.
extern "C" int x3d_poisson_set_stretching_zfirst(x3d_poisson *p, int sym, const real_t *a0, const real_t *a1)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && a0 && (!sym || a1), "x3d_poisson_set_stretching_zfirst: null argument");
    X3D_REQUIRE(p->stretched && p->sym == sym, "x3d_poisson_set_stretching_zfirst: call x3d_poisson_set_stretching first (same sym)");
    X3D_REQUIRE(p->nz % 2 == 0 && (!sym || p->ny % 2 == 0), "x3d_poisson_set_stretching_zfirst: even nz (and ny with sym)");
    const int n = sym ? p->ny / 2 : p->ny, nzh = p->nz / 2 + 1, px = p->nx + 16;
    const size_t rows = 5 * (size_t)nzh * n, bytes = sizeof(real_t) * rows * px;
    const real_t *src[2] = {a0, a1};
    for (int s = 0; s < (sym ? 2 : 1); s++) {
        if (!p->luz[s]) X3D_HIP(hipMalloc(&p->luz[s], bytes));
        if (int rc = upload_pitched(p->luz[s], src[s], rows, p->nx, px, sizeof(real_t))) return rc;
        hipLaunchKernelGGL(k_penta_factor<false>, penta_grid(px, nzh), dim3(64), 0, p->b->stream, p->luz[s], px, n, nzh);
        X3D_HIP(hipGetLastError());
    }
    X3D_HIP(hipStreamSynchronize(p->b->stream));
    return 0;
}

extern "C" int x3d_poisson_postprocess_010(x3d_poisson *p)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p, "x3d_poisson_postprocess_010: null argument");
    if (x3d_lazy_active(p->b)) return x3d_lazy_fft(p->b, 3, p, nullptr);
    ProfScope ps(p->b, X3D_K_SPECTRAL);
    return spectral_010_launch(p->b->stream, p->c, p->waves, p->nxs, p->nx, p->ny, p->nz, 0, p->ab, p->stretched, p->sym,
                               p->lu);
}

// poisson_010 (src/poisson_fft.f90:228-242): f holds the rhs on entry and the solution on exit
extern "C" int x3d_poisson_solve_010(x3d_poisson *p, real_t *f, real_t *temp)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f && temp, "x3d_poisson_solve_010: null argument");
    X3D_LAZY_OUT(p->b, f, false);
    X3D_LAZY_OUT(p->b, temp, true);
    X3D_LAZY_EAGER(p->b);
    if (int rc = x3d_poisson_enforce_periodicity_y(p, temp, f)) return rc;
    if (int rc = x3d_poisson_solve_010_rows(p, temp)) return rc;
    return x3d_poisson_undo_periodicity_y(p, f, temp);
}

int x3d_y010_run(x3d_backend *b, real2_t *c, int nxs, int nx, int ny, int nz, int mode, const real_t *tables, int sym,
                 real_t *const lu[2], bool *done, int nzl = 0);
#define Y010_DEFAULT_FORM 1  // staged (y010.hip has the measurements)

// plans of the y-last form: 2-D transforms over (z, x) -- r2c along x on the pitched block, c2c along z -- batched over
// the y rows (the middle dimension of both layouts is the batch: plane pitch as the inner embedding)
static int y010_setup(x3d_poisson *p)
{
    if (p->y010) return 0;
    p->y010 = -1;
    const char *e = getenv("X3D_NO_Y010");
    x3d_backend *b = p->b;
    // (only the stretched solve takes the y-last path: no plans, no larger work area -- and no implicit device sync inside
    //  a step -- for anything else)
    if ((e && e[0] == '1') || !p->stretched || p->ny != 256 || p->nxs % 8 != 0) return 0;
    int nn[2] = {p->nz, p->nx};
    int re[2] = {b->nzp, b->nyp * b->nxp}, ce[2] = {p->nz, p->ny * p->nxs};
    // a plan that cannot be made is not an error of the solve: the partial plans go, y010 stays -1, the 3-D path runs
    bool have_fw = false, have_bw = false;
    auto give_up = [&]() {
        if (have_fw) hipfftDestroy(p->plan_x010_fw);
        if (have_bw) hipfftDestroy(p->plan_x010_bw);
        return 0;
    };
    if (hipfftCreate(&p->plan_x010_fw) != HIPFFT_SUCCESS) return give_up();
    have_fw = true;
    if (hipfftCreate(&p->plan_x010_bw) != HIPFFT_SUCCESS) return give_up();
    have_bw = true;
    if (hipfftSetAutoAllocation(p->plan_x010_fw, 0) != HIPFFT_SUCCESS || hipfftSetAutoAllocation(p->plan_x010_bw, 0) != HIPFFT_SUCCESS)
        return give_up();
    size_t ws_fw = 0, ws_bw = 0;
    if (hipfftMakePlanMany(p->plan_x010_fw, 2, nn, re, 1, b->nxp, ce, 1, p->nxs, X3D_FFT_R2C, p->ny, &ws_fw) != HIPFFT_SUCCESS ||
        hipfftMakePlanMany(p->plan_x010_bw, 2, nn, ce, 1, p->nxs, re, 1, b->nxp, X3D_FFT_C2R, p->ny, &ws_bw) != HIPFFT_SUCCESS)
        return give_up();
    const size_t ws = ws_fw > ws_bw ? ws_fw : ws_bw;
    if (ws > p->work_size) {  // (the 3-D plans keep working in the larger area)
        X3D_HIP(hipFree(p->work));
        p->work = nullptr;
        X3D_HIP(hipMalloc(&p->work, ws));
        p->work_size = ws;
        X3D_FFT(hipfftSetWorkArea(p->plan_fw, p->work));
        X3D_FFT(hipfftSetWorkArea(p->plan_bw, p->work));
    }
    X3D_FFT(hipfftSetWorkArea(p->plan_x010_fw, p->work));
    X3D_FFT(hipfftSetWorkArea(p->plan_x010_bw, p->work));
    p->y010 = 1;
    return 0;
}

// poisson_010 (src/poisson_fft.f90:228-242) without its two row permutations: f's y rows are already in
// enforce_periodicity_y's order and the solution is left in that order.  ny = 256: x and z (one 2-D plan) ; the fused y pass (y010.hip:
// y transform, fft_postprocess_010, inverse y transform in one pass over the spectrum) ; z ; x.  Otherwise the 3-D
// transforms with the post-processing kernels between them.
extern "C" int x3d_poisson_solve_010_rows(x3d_poisson *p, real_t *f)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f, "x3d_poisson_solve_010_rows: null argument");
    X3D_LAZY_OUT(p->b, f, false);
    X3D_LAZY_EAGER(p->b);
    if (int rc = y010_setup(p)) return rc;
    if (p->y010 == 1 && p->stretched) {  // (uniform y: one post-processing kernel between the 3-D transforms)
        x3d_backend *b = p->b;
        {
            ProfScope ps(b, X3D_K_FFT, 1);
            X3D_FFT(hipfftSetStream(p->plan_x010_fw, b->stream));
            X3D_FFT(x3d_fftExecR2C(p->plan_x010_fw, (x3d_fft_real *)f, (x3d_fft_cplx *)p->c));
        }
        bool done = false;
        // X3D_Y010_FORM: "split" = k_y010<0> ; k_penta_solve x 2 ; k_y010<1>.  "staged" = the forward sweeps on the
        // tile of the first kernel, the backward sweeps on the tile of the second, the factored operator staged in LDS
        // (k_y010<3> ; k_y010<4>).  "fused" = everything in k_y010<2>, the operator streamed by the chains (slow).
        const char *ef = getenv("X3D_Y010_FORM");
        const int form = !ef || !ef[0] ? Y010_DEFAULT_FORM : (ef[0] == 'f' ? 2 : (ef[0] == 's' && ef[1] == 't' ? 1 : 0));
        if (form == 2) {
            ProfScope ps(b, X3D_K_SPECTRAL);
            if (int rc = x3d_y010_run(b, p->c, p->nxs, p->nx, p->ny, p->nz, 2, p->ab, p->sym, p->lu, &done)) return rc;
        } else if (form == 1) {
            {
                ProfScope ps(b, X3D_K_SPECTRAL);
                if (int rc = x3d_y010_run(b, p->c, p->nxs, p->nx, p->ny, p->nz, 3, p->ab, p->sym, p->lu, &done)) return rc;
            }
            if (done) {
                ProfScope ps(b, X3D_K_SPECTRAL);
                if (int rc = x3d_y010_run(b, p->c, p->nxs, p->nx, p->ny, p->nz, 4, p->ab, p->sym, p->lu, &done)) return rc;
                X3D_REQUIRE(done, "x3d_poisson_solve_010_rows: y pass refused");
            }
        }
        if (!done) {
            {
                ProfScope ps(b, X3D_K_FFT, 1);
                if (int rc = x3d_y010_run(b, p->c, p->nxs, p->nx, p->ny, p->nz, 0, p->ab, p->sym, p->lu, &done)) return rc;
                X3D_REQUIRE(done, "x3d_poisson_solve_010_rows: y pass refused");
            }
            {
                ProfScope ps(b, X3D_K_SPECTRAL);
                const dim3 g2 = penta_grid(p->nxs, p->nz);
                if (p->sym) {
                    hipLaunchKernelGGL(k_penta_solve<false>, g2, dim3(64), 0, b->stream, p->c, p->lu[0], 0, 2, p->nxs, p->ny, p->nz,
                                       p->ny / 2, p->nx, 0);
                    hipLaunchKernelGGL(k_penta_solve<false>, g2, dim3(64), 0, b->stream, p->c, p->lu[1], 1, 2, p->nxs, p->ny, p->nz,
                                       p->ny / 2, p->nx, 0);
                } else {
                    hipLaunchKernelGGL(k_penta_solve<false>, g2, dim3(64), 0, b->stream, p->c, p->lu[0], 0, 1, p->nxs, p->ny, p->nz,
                                       p->ny, p->nx, 0);
                }
                X3D_HIP(hipGetLastError());
            }
            ProfScope ps(b, X3D_K_FFT, 2);
            if (int rc = x3d_y010_run(b, p->c, p->nxs, p->nx, p->ny, p->nz, 1, p->ab, p->sym, p->lu, &done)) return rc;
        }
        ProfScope ps(b, X3D_K_FFT, 2);
        X3D_FFT(hipfftSetStream(p->plan_x010_bw, b->stream));
        X3D_FFT(x3d_fftExecC2R(p->plan_x010_bw, (x3d_fft_cplx *)p->c, (x3d_fft_real *)f));
        return 0;
    }
    if (int rc = x3d_poisson_fft_forward(p, f)) return rc;
    if (int rc = x3d_poisson_postprocess_010(p)) return rc;
    return x3d_poisson_fft_backward(p, f);
}

extern "C" int x3d_poisson_get_spectral(x3d_poisson *p, real_t *host)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && host, "null argument");
    X3D_LAZY_FLUSH(p->b);
    X3D_LAZY_EAGER(p->b);
    X3D_HIP(hipStreamSynchronize(p->b->stream));
    X3D_HIP(hipMemcpy2D(host, p->nxm * sizeof(real2_t), p->c, p->nxs * sizeof(real2_t), p->nxm * sizeof(real2_t),
                        (size_t)p->nz * p->ny, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int x3d_poisson_set_spectral(x3d_poisson *p, const real_t *host)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && host, "null argument");
    X3D_LAZY_FLUSH(p->b);
    X3D_LAZY_EAGER(p->b);
    X3D_HIP(hipStreamSynchronize(p->b->stream));
    return upload_pitched(p->c, host, (size_t)p->nz * p->ny, p->nxm, p->nxs, sizeof(real2_t));
}
