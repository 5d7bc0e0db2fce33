// z-first form of the all-periodic spectral Poisson solve at 512^3 on one rank (poisson_000,
// /root/reference/src/poisson_fft.f90:216-226: fft_forward ; fft_postprocess_000 ; fft_backward).
//
// The reference transforms x, y, z one after the other (src/backend/omp/poisson_fft.f90:89-97, 129-137), divides in
// a pass of its own (src/backend/omp/kernels/spectral_processing.f90:7-106) and comes back: with every 1-D stage a
// read and a write of the spectrum that is 10.5 passes even with the middle transform, the division and its inverse in
// one kernel (csrc/fft512.hip).  The 3-D DFT does not care about the order.  Here
//   z   real -> complex on the LDS tile of the kernel that produced the field (k_ytile_tds_pair<.., ZF>: the last z
//       operator pair of divergence_v2c, src/vector_calculus.f90:142-246), or k_ztile_fft<true> on a field in memory
//   x   complex, contiguous rows (k_c2c512_x)                                        2 passes
//   y   forward + process_spectral_000 + inverse in one kernel (k_fft512<2, ., ZH>)  2.5 passes (reciprocal wave numbers)
//   x   inverse                                                                      2 passes
//   z   complex -> real on the tile of the kernel that consumes the pressure (the first z pair of gradient_c2v,
//       :248-332), or k_ztile_fft<false>
// = 6.5 passes.  Spectrum C[kz][y][x]: the HALF axis is z (kz = 0 .. 256), x and y are full; rows of ZH_PX complex
// numbers; it lives in the solver's spectral workspace.  The spectral operation is the reference's sequence of
// rotations, division and inverse rotations with the x index mirrored above nx / 2 the way y and z are there (the
// rotations cancel pairwise: |b - i a| = 1, so any consistent choice gives the result up to rounding).
#include <vector>

#include "poisson_priv.h"
#include "zfft_tile.h"

#define ZH_PX 520  // row pitch of C in complex numbers (512 + 8: consecutive y rows do not share an HBM channel pattern)

int x3d_fft512_init();
const real2_t *x3d_fft512_twiddles();
int x3d_ztile_fft_run(x3d_backend *b, real_t *f, const ZfArg &zf, bool fwd, int y0, int nyr);
int x3d_fft512_run_zh(x3d_backend *b, real2_t *c, long px, int kz0, int nkz, const real_t *rwZ, const real_t *ab, int nx,
                      int ny, int nz);

// rwZ[kz][x][y] = -1 / waves(min(x, nx - x), y, kz)  (0 where waves < 1e-16); waves = [nz][ny][nxs] (x: nx/2+1 modes)
__global__ void __launch_bounds__(256)
    k_zh_rw(real_t *__restrict__ rwZ, const real_t *__restrict__ waves, int nx, int ny, int nxs)
{
    __shared__ real_t t[32][33];
    const int kz = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + tx, y = y0 + r, xm = x <= nx / 2 ? x : nx - x;
        const real_t wv = waves[((size_t)kz * ny + y) * nxs + xm];
        t[r][tx] = wv < 1.e-16 ? 0.0 : -1.0 / wv;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) rwZ[((size_t)kz * nx + x0 + r) * ny + y0 + tx] = t[tx][r];
}

// complex transform of contiguous rows of 512 (the x axis of C), one row per wave, in place
template <int S>
__global__ void __launch_bounds__(512)
    k_c2c512_x(real2_t *__restrict__ c, const real2_t *__restrict__ twg, long nrows, long pitch)
{
    extern __shared__ real2_t zx[];  // [8][FP] + 256 twiddles
    real2_t *__restrict__ tws = zx + 8 * FP;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    __syncthreads();
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    real2_t *__restrict__ pen = zx + w * FP;
    long row = (long)blockIdx.x * 8 + w;
    const long step = (long)gridDim.x * 8;
    real2_t a[8], nx8[8];
    if (row < nrows) {
#pragma unroll
        for (int k = 0; k < 8; k++) nx8[k] = c[row * pitch + l + 64 * k];
    }
    for (; row < nrows; row += step) {
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = nx8[k];
        if (row + step < nrows) {
#pragma unroll
            for (int k = 0; k < 8; k++) nx8[k] = c[(row + step) * pitch + l + 64 * k];
        }
        fft512_wave<S>(a, pen, tws, l);
#pragma unroll
        for (int k = 0; k < 8; k++) c[row * pitch + l + 64 * k] = a[k];
    }
}

// complex transform of contiguous rows of 1024 (the x axis of the channel's z-first spectrum, round 6), one row per wave, in
// place.  Decimation in frequency over the two halves of the row: lane l holds x[l + 64 m], m = 0 .. 15, so x[j] and
// x[j + 512] sit in the same lane:  X[2 k] = FFT512(x[j] + x[j + 512]),  X[2 k + 1] = FFT512((x[j] - x[j + 512]) W1024^j)
// -- two fft512_wave calls through the wave's own LDS region; their outputs a[k] = X[2 (l + 64 k)], o[k] = X[2 (l + 64 k) + 1]
// are neighbours in memory: 32 contiguous bytes per lane and k.  tw1024 = W1024^j, j = 0 .. 511 (host-generated).
template <int S>
__global__ void __launch_bounds__(512)
    k_c2c1024_x(real2_t *__restrict__ c, const real2_t *__restrict__ twg, const real2_t *__restrict__ tw1024, long nrows,
                long pitch)
{
    extern __shared__ real2_t zx[];  // [8][FP] + 256 twiddles
    real2_t *__restrict__ tws = zx + 8 * FP;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    __syncthreads();
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    real2_t *__restrict__ pen = zx + w * FP;
    real2_t wt[8];  // W1024^(S-signed (l + 64 m)): constant over the rows
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const real2_t t = tw1024[l + 64 * m];
        wt[m] = make_real2(t.x, S > 0 ? -t.y : t.y);
    }
    const long step = (long)gridDim.x * 8;
    for (long row = (long)blockIdx.x * 8 + w; row < nrows; row += step) {
        real2_t *__restrict__ r = c + row * pitch;
        real2_t a[8], o[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const real2_t lo = r[l + 64 * m], hi = r[l + 64 * (m + 8)];
            a[m] = cadd(lo, hi);
            o[m] = cmul(csub(lo, hi), wt[m]);
        }
        fft512_wave<S>(a, pen, tws, l);
        fft512_wave<S>(o, pen, tws, l);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            r[2 * (l + 64 * k)] = a[k];
            r[2 * (l + 64 * k) + 1] = o[k];
        }
    }
}

// the z transform of a field in memory, tile by tile (FWD: f -> C, else C -> f); the same tile mechanics as the fused
// forms in k_ytile_tds_pair
template <bool FWD>
__global__ void __launch_bounds__(1024)
    k_ztile_fft(real_t *__restrict__ f, ZfArg zf, int ntx, int tile0, int ntiles, long prow, long pplane)
{
    ntiles += tile0;  // tiles [tile0, tile0 + ntiles): a range of y rows (csrc/sfftz.hip)
    extern __shared__ real_t zarea[];  // ZF_AREA_DOUBLES + 256 twiddles
    constexpr int TP = 516;
    real2_t *__restrict__ tws = reinterpret_cast<real2_t *>(zarea + ZF_AREA_DOUBLES);
    if (threadIdx.x < 256) tws[threadIdx.x] = zf.tw[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cy = threadIdx.x >> 3, cc = threadIdx.x & 7;
    const long kzs = (long)zf.ny * zf.px;
    __syncthreads();
    ZfRows v{};
    if (!FWD && tile0 + (int)blockIdx.x < ntiles)
        v = zf_inverse_load(zf.c + (long)((tile0 + blockIdx.x) / ntx) * zf.px + ((tile0 + blockIdx.x) % ntx) * 16, kzs);
    for (int tl = tile0 + blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = (long)(tl / ntx) * pplane + (long)(tl % ntx) * 16;
        real2_t *__restrict__ crow = zf.c + (long)(tl / ntx) * zf.px + (tl % ntx) * 16;
        if (FWD) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const real2_t g = *reinterpret_cast<const real2_t *>(f + off + (long)(cy + 128 * i) * prow + 2 * cc);
                zarea[(2 * cc) * TP + cy + 128 * i] = g.x;
                zarea[(2 * cc + 1) * TP + cy + 128 * i] = g.y;
            }
            __syncthreads();
            zf_forward<TP>(zarea, tws, crow, kzs, wave, lane);
        } else {
            zf_inverse<TP>(zarea, tws, v, wave, lane);
            const int tn = tl + gridDim.x;
            if (tn < ntiles) v = zf_inverse_load(zf.c + (long)(tn / ntx) * zf.px + (tn % ntx) * 16, kzs);
#pragma unroll
            for (int i = 0; i < 4; i++)
                *reinterpret_cast<real2_t *>(f + off + (long)(cy + 128 * i) * prow + 2 * cc) =
                    make_real2(zarea[(2 * cc) * TP + cy + 128 * i], zarea[(2 * cc + 1) * TP + cy + 128 * i]);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- host side
static bool zfirst_sizes(const x3d_poisson *p)
{
    const x3d_backend *b = p->b;
    return p->fast512 && p->nx == 512 && p->ny == 512 && p->nz == 512 && b->nx == 512 && b->ny == 512 && b->nz == 512 &&
           (size_t)257 * 512 * ZH_PX <= (size_t)p->nz * p->ny * p->nxs && !p->stretched;
}

// ---- the channel's solve (010: non-periodic, stretched y, 256 cell rows; nx = 1024, nz = 512), round 6.  The y stage of that
// solve (csrc/y010.hip) works column by column of (x mode, z mode): which axis is the half one does not matter to it, so the
// transform along z moves onto the tiles of the z operator pairs here too, the x transform becomes a complex one over
// contiguous rows (k_c2c1024_x) and the two rocFFT kernels per direction of x3d_poisson_solve_010_rows go away:
//   z pairs (+ z transform)  ->  x forward  ->  k_y010<3> ; k_y010<4>  ->  x inverse  ->  z pairs
// 13 passes over the spectrum and the operators instead of 17.  Spectrum C[kz][y'][x]: kz = 0 .. 256, y' = the solver's
// interleaved row order (the pairs write / read it there: ZfArg::permn), x = 0 .. 1023 at row pitch ZH010_PX.  The
// operators come factored in the same layout (x3d_poisson_set_stretching_zfirst); the rotations by the staggered-grid
// phases take their mirrored form above nx / 2 -- what the reference does along z (spectral010.h, Rot::flip).
#define ZH010_PX 1040
int x3d_y010_run(x3d_backend *b, real2_t *c, int nxs, int nx, int ny, int nz, int mode, const real_t *tables, int sym,
                 real_t *const lu[2], bool *done, int nzl);
static real2_t *g_tw1024 = nullptr;  // W1024^j = exp(-2 pi i j / 1024), j = 0 .. 511
static int fft1024_init()
{
    if (g_tw1024) return 0;
    std::vector<real2_t> h(512);
    for (int j = 0; j < 512; j++) {
        const long double a = -2.0L * 3.141592653589793238462643383279502884L * j / 1024.0L;
        h[j] = make_real2((real_t)cosl(a), (real_t)sinl(a));
    }
    X3D_HIP(hipMalloc(&g_tw1024, sizeof(real2_t) * 512));
    X3D_HIP(hipMemcpy(g_tw1024, h.data(), sizeof(real2_t) * 512, hipMemcpyHostToDevice));
    return 0;
}
static bool zfirst010_off()
{
    static int off = -1;
    if (off < 0) { const char *e = getenv("X3D_NO_ZFIRST010"); off = (e && e[0] == '1') ? 1 : 0; }
    return off != 0;
}
static bool zfirst010_sizes(const x3d_poisson *p)
{
    const x3d_backend *b = p->b;
    return p->stretched && p->luz[0] && (!p->sym || p->luz[1]) && p->nx == 1024 && p->ny == 256 && p->nz == 512 &&
           b->nx == 1024 && b->nz == 512 && b->ny >= 256 && p->c_elems >= (size_t)257 * 256 * ZH010_PX;
}

// X3D_NO_ZFIRST=1, read ONCE per process (both functions below ask here)
static bool zfirst_off()
{
    static int off = -1;
    if (off < 0) { const char *e = getenv("X3D_NO_ZFIRST"); off = (e && e[0] == '1') ? 1 : 0; }
    return off != 0;
}

// the z-first solve is on offer for this solver (X3D_NO_ZFIRST=1: never); builds the reciprocal wave numbers on first use
int x3d_zfirst_arg(x3d_poisson *p, ZfArg *out, bool *ok)
{
    *ok = false;
    if (p->ext_middle) {  // a proxy (the shim's y-slab solve): the spectrum is the slab solver's
        if (int rc = x3d_fft512_init()) return rc;
        if (out) *out = ZfArg{p->ext_c, x3d_fft512_twiddles(), p->ext_ny, p->ext_px, 0};
        *ok = !zfirst_off();
        return 0;
    }
    if (zfirst_off()) return 0;
    if (!zfirst010_off() && zfirst010_sizes(p)) {  // the channel's 010 solve
        if (int rc = x3d_fft512_init()) return rc;
        if (int rc = fft1024_init()) return rc;
        if (out) *out = ZfArg{p->c, x3d_fft512_twiddles(), p->ny, (long)ZH010_PX, p->ny};
        *ok = true;
        return 0;
    }
    if (!zfirst_sizes(p)) return 0;
    if (!p->rwZ) {
        X3D_HIP(hipMalloc(&p->rwZ, sizeof(real_t) * 257 * 512 * 512));
        hipLaunchKernelGGL(k_zh_rw, dim3(16, 16, 257), dim3(256), 0, p->b->stream, p->rwZ, p->waves, p->nx, p->ny, p->nxs);
        X3D_HIP(hipGetLastError());
    }
    if (out) *out = ZfArg{p->c, x3d_fft512_twiddles(), p->ny, (long)ZH_PX};
    *ok = true;
    return 0;
}

// (for the deferred-execution layer's rewrite: no side effects)
bool x3d_zfirst_on_offer(x3d_poisson *p)
{
    return p && !zfirst_off() && (p->ext_middle != nullptr || zfirst_sizes(p));
}

extern "C" int x3d_poisson_zfirst_ok(x3d_poisson *p, int *ok)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && ok, "x3d_poisson_zfirst_ok: null argument");
    bool o = false;
    if (int rc = x3d_zfirst_arg(p, nullptr, &o)) return rc;
    *ok = o ? 1 : 0;
    return 0;
}

template <int S>
static int c2c_x(x3d_poisson *p, int kz0, int nkz)
{
    const int lds = sizeof(real2_t) * (8 * FP + 256);
    X3D_LDS_OPTIN(p->b, (k_c2c512_x<S>));
    const long nrows = (long)nkz * p->ny;
    long blocks = (nrows + 7) / 8;
    if (blocks > 2048) blocks = 2048;
    ProfScope ps(p->b, X3D_K_FFT, S < 0 ? 1 : 2);
    hipLaunchKernelGGL((k_c2c512_x<S>), dim3((unsigned)blocks), dim3(512), lds, p->b->stream,
                       p->c + (long)kz0 * p->ny * ZH_PX, x3d_fft512_twiddles(), nrows, (long)ZH_PX);
    X3D_HIP(hipGetLastError());
    return 0;
}

// C (z already transformed) -> x forward ; y forward + process_spectral_000 + y inverse ; x inverse -> C
extern "C" int x3d_poisson_zfirst_middle(x3d_poisson *p)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p, "x3d_poisson_zfirst_middle: null argument");
    bool ok = false;
    if (int rc = x3d_zfirst_arg(p, nullptr, &ok)) return rc;
    X3D_REQUIRE(ok || p->ext_middle, "x3d_poisson_zfirst_middle: not a 512^3 all-periodic solver (x3d_poisson_zfirst_ok)");
    X3D_LAZY_EAGER(p->b);
    if (p->ext_middle) return p->ext_middle(p->ext_user);
    if (p->stretched) {  // 010: x forward ; the y pass with the pentadiagonal sweeps on its tiles ; x inverse
        x3d_backend *b = p->b;
        const int lds = sizeof(real2_t) * (8 * FP + 256);
        const long nrows = 257L * p->ny;
        const unsigned blocks = (unsigned)(nrows / 8 < 2048 ? (nrows + 7) / 8 : 2048);
        X3D_LDS_OPTIN(b, (k_c2c1024_x<-1>));
        X3D_LDS_OPTIN(b, (k_c2c1024_x<1>));
        {
            ProfScope ps(b, X3D_K_FFT, 1);
            hipLaunchKernelGGL((k_c2c1024_x<-1>), dim3(blocks), dim3(512), lds, b->stream, p->c, x3d_fft512_twiddles(), g_tw1024,
                               nrows, (long)ZH010_PX);
        }
        bool done = false;
        {
            ProfScope ps(b, X3D_K_SPECTRAL);
            if (int rc = x3d_y010_run(b, p->c, ZH010_PX, p->nx, p->ny, p->nz, 3, p->ab, p->sym, p->luz, &done, 257)) return rc;
            X3D_REQUIRE(done, "x3d_poisson_zfirst_middle: y pass refused");
        }
        {
            ProfScope ps(b, X3D_K_SPECTRAL);
            if (int rc = x3d_y010_run(b, p->c, ZH010_PX, p->nx, p->ny, p->nz, 4, p->ab, p->sym, p->luz, &done, 257)) return rc;
            X3D_REQUIRE(done, "x3d_poisson_zfirst_middle: y pass refused");
        }
        ProfScope ps(b, X3D_K_FFT, 2);
        hipLaunchKernelGGL((k_c2c1024_x<1>), dim3(blocks), dim3(512), lds, b->stream, p->c, x3d_fft512_twiddles(), g_tw1024,
                           nrows, (long)ZH010_PX);
        X3D_HIP(hipGetLastError());
        return 0;
    }
    // (the three kernels on groups of 8 .. 64 kz planes, so that a group stays in the memory-side cache between them:
    // measured, no gain -- profiles/README.md)
    if (int rc = c2c_x<-1>(p, 0, 257)) return rc;
    if (int rc = x3d_fft512_run_zh(p->b, p->c, ZH_PX, 0, 257, p->rwZ, p->ab, p->nx, p->ny, p->nz)) return rc;
    return c2c_x<1>(p, 0, 257);
}

// ---- proxy of a z-first solve whose middle the CALLER runs (include/x3d2_hip.h, x3d_poisson_create_proxy)
extern "C" int x3d_poisson_create_proxy(x3d_backend *b, x3d_poisson **out, x3d_real *spectrum, int ny, long px,
                                        int (*middle)(void *), void *user)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && spectrum && middle && ny > 0 && px >= b->nx, "x3d_poisson_create_proxy: bad argument");
    X3D_REQUIRE(b->nz == 512 && b->nx % 16 == 0 && ny <= b->ny, "x3d_poisson_create_proxy: 512-row z pencils");
    x3d_poisson *p = new x3d_poisson();
    memset(p, 0, sizeof *p);
    p->b = b;
    p->nx = b->nx; p->ny = ny; p->nz = 512;
    p->ext_middle = middle; p->ext_user = user;
    p->ext_c = reinterpret_cast<real2_t *>(spectrum); p->ext_ny = ny; p->ext_px = px;
    if (int rc = x3d_fft512_init()) return rc;
    *out = p;
    return 0;
}
// the proxy's hooks run eagerly (the deferred layer off): forward = z transform of the field, postprocess = the caller's
// middle, backward = inverse z transform
int x3d_proxy_hook(x3d_poisson *p, int which, real_t *f)
{
    if (which == 1) return p->ext_middle(p->ext_user);
    ZfArg zf{p->ext_c, x3d_fft512_twiddles(), p->ext_ny, p->ext_px, 0};
    return x3d_ztile_fft_run(p->b, f, zf, which == 0, 0, -1);
}

// the z transform of a block's field, tile by tile (also for csrc/sfftz.hip)
int x3d_ztile_fft_run(x3d_backend *b, real_t *f, const ZfArg &zf, bool fwd, int y0, int nyr)
{
    X3D_REQUIRE(b->nz == 512 && b->nx % 16 == 0 && zf.ny <= b->ny && zf.permn == 0, "x3d_ztile_fft_run: 512-row z pencils");
    if (nyr < 0) { y0 = 0; nyr = zf.ny; }
    X3D_REQUIRE(y0 >= 0 && y0 + nyr <= zf.ny, "x3d_ztile_fft_run: rows [%d, %d) of %d", y0, y0 + nyr, zf.ny);
    if (nyr == 0) return 0;
    const int ntx = b->nx / 16, ntiles = ntx * nyr, tile0 = ntx * y0;
    const size_t lds = sizeof(real_t) * (ZF_AREA_DOUBLES + 512);
    const long pxy = (long)b->nxp * b->nyp;
    ProfScope ps(b, X3D_K_FFT, 3);
    if (fwd) {
        X3D_LDS_OPTIN(b, (k_ztile_fft<true>));
        hipLaunchKernelGGL((k_ztile_fft<true>), dim3(ntiles < 512 ? ntiles : 512), dim3(1024), lds, b->stream, f, zf, ntx, tile0, ntiles, pxy, (long)b->nxp);
    } else {
        X3D_LDS_OPTIN(b, (k_ztile_fft<false>));
        hipLaunchKernelGGL((k_ztile_fft<false>), dim3(ntiles < 512 ? ntiles : 512), dim3(1024), lds, b->stream, f, zf, ntx, tile0, ntiles, pxy, (long)b->nxp);
    }
    X3D_HIP(hipGetLastError());
    return 0;
}

static int ztile(x3d_poisson *p, real_t *f, bool fwd)
{
    ZfArg zf;
    bool ok = false;
    if (int rc = x3d_zfirst_arg(p, &zf, &ok)) return rc;
    X3D_REQUIRE(ok && !p->stretched, "x3d_poisson_zfirst: not a 512^3 all-periodic solver (x3d_poisson_zfirst_ok)");
    return x3d_ztile_fft_run(p->b, f, zf, fwd, 0, -1);
}

// stand-alone ends of the z-first solve: f (cell data of a block) -> C, and back
extern "C" int x3d_poisson_zfirst_forward(x3d_poisson *p, const real_t *f_in)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_in, "x3d_poisson_zfirst_forward: null argument");
    X3D_LAZY_SYNC(p->b);
    return ztile(p, const_cast<real_t *>(f_in), true);
}
extern "C" int x3d_poisson_zfirst_backward(x3d_poisson *p, real_t *f_out)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f_out, "x3d_poisson_zfirst_backward: null argument");
    X3D_LAZY_SYNC(p->b);
    return ztile(p, f_out, false);
}
// x3d_poisson_solve_010_rows through the z-first stages with the z transforms as kernels of their own, in place: f's first
// 256 y rows are in enforce_periodicity_y's order and stay there (== x3d_poisson_solve_010_rows up to rounding; the fused
// driver has the z transforms inside its z operator pairs instead, x3d_tds_pair_zfirst)
extern "C" int x3d_poisson_solve_010_rows_zfirst(x3d_poisson *p, real_t *f)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f, "x3d_poisson_solve_010_rows_zfirst: null argument");
    X3D_LAZY_SYNC(p->b);
    ZfArg zf;
    bool ok = false;
    if (int rc = x3d_zfirst_arg(p, &zf, &ok)) return rc;
    X3D_REQUIRE(ok && p->stretched, "x3d_poisson_solve_010_rows_zfirst: not on offer for this solver (x3d_poisson_zfirst_ok)");
    zf.permn = 0;  // (the rows of f are interleaved already)
    if (int rc = x3d_ztile_fft_run(p->b, f, zf, true, 0, -1)) return rc;
    if (int rc = x3d_poisson_zfirst_middle(p)) return rc;
    return x3d_ztile_fft_run(p->b, f, zf, false, 0, -1);
}

// poisson_000 through the z-first stages, in place (== x3d_poisson_solve_000 up to rounding)
extern "C" int x3d_poisson_solve_000_zfirst(x3d_poisson *p, real_t *f)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f, "x3d_poisson_solve_000_zfirst: null argument");
    if (int rc = x3d_poisson_zfirst_forward(p, f)) return rc;
    if (int rc = x3d_poisson_zfirst_middle(p)) return rc;
    return x3d_poisson_zfirst_backward(p, f);
}
