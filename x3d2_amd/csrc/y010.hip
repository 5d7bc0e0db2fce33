// K6y: the y pass of the non-periodic-y (010) Poisson solve for ny = 256 cells (the channel case, BASELINE configs[4]):
//   forward  y transform + process_spectral_010's forward part     in ONE pass over the spectrum   (MODE 0)
//   its backward part + inverse y transform                        in ONE pass                     (MODE 1)
//   both, with the pentadiagonal solves of the stretched-mesh operator between them, in ONE pass   (MODE 2)
// instead of rocFFT's strided y stage, k_spectral_010<0>, k_penta_solve x 2, k_spectral_010<1>, rocFFT's inverse y
// stage (six passes; spectral010.h has those kernels and the reference lines they mirror:
// src/backend/cuda/kernels/spectral_processing.f90:385-702, src/backend/cuda/poisson_fft.f90:822-924).
// The 3-D DFT is separable: the reference runs x, y, z through cuFFT / 2decomp&FFT and then post-processes; here x and z
// are transformed first (rocFFT, 1-D batched plans, poisson.hip) and y LAST, so that everything that couples the rows
// of one (x mode, z mode) column -- the y transform, the paired split of rows j and ny - j + 2, the pentadiagonal
// systems along y -- happens while the column is on chip.
//
// Work decomposition: a workgroup of 8 waves owns the 8 x-adjacent columns of one z mode: c[kz][0..255][x0..x0+7]
// (128-byte row segments).  The tile sits in LDS as 8 pencils of Y010_P double2; wave w transforms pencil w with
// fft256_wave (fft512_core.h, 4 points per lane, exchanges through the pencil's own LDS region) and does the paired
// split on it; nothing but the load, the pentadiagonal phase and the store needs a block barrier.
// Pentadiagonal phase (MODE 2, X3D_Y010_FUSED=1 -- MEASURED SLOWER, off by default): wave 0, lane = (x 0..7, re / im,
// odd / even system): 32 independent serial chains over the 128 rows of a system, right-hand sides in LDS, the factored
// operator (k_penta_factor's storage, 5 doubles per entry) streamed from memory in chunks of 8 rows requested one chunk
// ahead; the eight x of a row are one 64-byte segment.  Arithmetic = k_penta_solve's, operation by operation.
// Channel bench, 1024 x 257 x 512 (profiles/r04_channel_round4.txt): 3-D transforms + four post-processing kernels 14.9 ms per step
// in the fft + spectral classes; MODE 0 / 1 around the two k_penta_solve launches 12.7 ms (step 60.7 -> 58.1 ms);
// MODE 2 23.1 ms (5.85 ms per launch): 32 lanes per workgroup wait a full memory latency per 8-row chunk of the
// operator, 32 times per tile, and 147 VGPRs leave 3 workgroups per CU to hide it.  Staging the operator in LDS by all
// threads would need 32 + 48 KB per tile beside the 39 KB tile (one workgroup per CU).  The k_penta_solve launches
// stay: they stream 104 B per entry at 4.6 TB/s.
#include "fft512_core.h"
#include "spectral010.h"

#define Y010_P 274  // >= 271 (fft256_wave's layouts) and = 2 mod 16: the 8 pencils' rows j sit in different LDS banks

struct Y010Arg {
    const double *ax, *bx, *ay, *by, *az, *bz;  // global tables (spectral010.h)
    const double *lu0, *lu1;                    // factored pentadiagonal operators [5][nz][n][nxs]
    int nxs, nz, nx, sym;
};

__device__ __forceinline__ void y010_pair_fw(double2 *__restrict__ pen, int j0, const Rot &rz, const Rot &rx,
                                             const double *__restrict__ ay, const double *__restrict__ by, int nx, int nz)
{
    constexpr int ny = 256;
    const int jr0 = ny - j0;  // (0-based partner row; j0 = 0 has none)
    const bool paired = j0 >= 1, self = paired && jr0 == j0;
    const double2 L = pen[j0], R = paired && !self ? pen[jr0] : L;
    double l_r = L.x, l_c = L.y, r_r = R.x, r_c = R.y;
    l_r = l_r / nx / ny / nz; l_c = l_c / nx / ny / nz;
    rot_fw(l_r, l_c, rz);
    rot_fw(l_r, l_c, rx);
    if (self) { r_r = l_r; r_c = l_c; }
    else if (paired) {
        r_r = r_r / nx / ny / nz; r_c = r_c / nx / ny / nz;
        rot_fw(r_r, r_c, rz);
        rot_fw(r_r, r_c, rx);
    }
    if (paired) {
        const double a = ay[j0], b = by[j0], a2 = ay[jr0], b2 = by[jr0];
        const double n_lr = 0.5 * (l_r * b + l_c * a + r_r * b - r_c * a);
        const double n_lc = 0.5 * (-l_r * a + l_c * b + r_r * a + r_c * b);
        const double n_rr = 0.5 * (r_r * b2 + r_c * a2 + l_r * b2 - l_c * a2);
        const double n_rc = 0.5 * (-r_r * a2 + r_c * b2 + l_r * a2 + l_c * b2);
        l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
        if (self) { l_r = r_r; l_c = r_c; }  // the second store wins on the self-paired row
    }
    pen[j0] = make_double2(l_r, l_c);
    if (paired && !self) pen[jr0] = make_double2(r_r, r_c);
}

__device__ __forceinline__ void y010_pair_bw(double2 *__restrict__ pen, int j0, const Rot &rz, const Rot &rx,
                                             const double *__restrict__ ay, const double *__restrict__ by)
{
    constexpr int ny = 256;
    const int jr0 = ny - j0;
    const bool paired = j0 >= 1, self = paired && jr0 == j0;
    const double2 L = pen[j0], R = paired && !self ? pen[jr0] : L;
    double l_r = L.x, l_c = L.y, r_r = R.x, r_c = R.y;
    if (paired) {
        if (self) { r_r = l_r; r_c = l_c; }
        const double a = ay[j0], b = by[j0], a2 = ay[jr0], b2 = by[jr0];
        const double n_lr = l_r * b - l_c * a + r_r * a + r_c * b;
        const double n_lc = l_r * a + l_c * b - r_r * b + r_c * a;
        const double n_rr = r_r * b2 - r_c * a2 + l_r * a2 + l_c * b2;
        const double n_rc = r_r * a2 + r_c * b2 - l_r * b2 + l_c * a2;
        l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
        if (self) { l_r = r_r; l_c = r_c; }
    }
    rot_bw(l_r, l_c, rz);
    rot_bw(l_r, l_c, rx);
    if (paired && !self) {
        rot_bw(r_r, r_c, rz);
        rot_bw(r_r, r_c, rx);
    }
    pen[j0] = make_double2(l_r, l_c);
    if (paired && !self) pen[jr0] = make_double2(r_r, r_c);
}

// k_penta_solve (spectral010.h) on the tile in LDS: this lane's chain = component `ri` of column x, system s
// (sym: rows 2 j + s - 2, 0-based, j = 1 .. n = 128; else all 256 rows)
__device__ __forceinline__ void y010_penta(double2 *__restrict__ sm, const Y010Arg &g, int kz, int x0, int lane)
{
    constexpr int ny = 256, U = 8;
    const int x = lane & 7, ri = (lane >> 3) & 1, s = lane >> 4;
    if (s >= (g.sym ? 2 : 1)) return;
    const int inc = g.sym ? 2 : 1, n = ny / inc;
    const double *__restrict__ lu = s ? g.lu1 : g.lu0;
    const size_t ds = (size_t)g.nz * n * g.nxs;
    const double *__restrict__ lub = lu + (size_t)kz * n * g.nxs + x0 + x;
#define LU(j, d) lub[(size_t)((d) - 1) * ds + (size_t)((j) - 1) * g.nxs]
    double *__restrict__ pd = reinterpret_cast<double *>(sm + x * Y010_P) + ri;
#define C(j) pd[2 * (inc * (j) + s - inc)]  // (inc j + off - h - 1 with off = s, h = inc / 2: 2 j + s - 2 or j - 1)
    const double eps = 1.e-16;
    // forward: rows j+1, j+2 -= m * row j; two rows are carried in registers
    double r0 = C(1), r1 = C(2);
    double m1c[U], m2c[U], m1n[U], m2n[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int j = 1 + u;
        m1c[u] = j <= n - 2 ? LU(j, 2) : 0.0;
        m2c[u] = j <= n - 2 ? LU(j, 1) : 0.0;
    }
    for (int jb = 1; jb <= n - 2; jb += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = jb + U + u;
            m1n[u] = j <= n - 2 ? LU(j, 2) : 0.0;
            m2n[u] = j <= n - 2 ? LU(j, 1) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = jb + u;
            if (j <= n - 2) {
                double r2 = C(j + 2);
                r1 = r1 - m1c[u] * r0;
                r2 = r2 - m2c[u] * r0;
                C(j) = r0;
                r0 = r1; r1 = r2;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { m1c[u] = m1n[u]; m2c[u] = m2n[u]; }
    }
    // last two rows: r0 = row n-1, r1 = row n
    const double tmp = LU(n - 1, 2), dd = LU(n, 3), inv = LU(n - 1, 3), a4n = LU(n - 1, 4);
    // (the first backward chunk, requested before the divisions)
    double ivc[U], a4c[U], a5c[U], ivn[U], a4x[U], a5n[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int j = n - 2 - u;
        ivc[u] = j >= 1 ? LU(j, 3) : 0.0;
        a4c[u] = j >= 1 ? LU(j, 4) : 0.0;
        a5c[u] = j >= 1 ? LU(j, 5) : 0.0;
    }
    double xn, xn1;
    if (fabs(dd) > eps) {
        const double tt = tmp / dd;
        xn = r1 / dd - tt * r0;
    } else {
        xn = 0.0;
    }
    const double q = a4n * inv;
    xn1 = r0 * inv - xn * q;
    const bool zero_line = (x0 + x + 1) == g.nx / 2 + 1 && (kz + 1) == g.nz / 2 + 1;
    if (zero_line) { xn = 0.0; xn1 = 0.0; }
    C(n) = xn;
    C(n - 1) = xn1;
    // backward
    double x1 = xn1, x2 = xn;
    for (int jb = n - 2; jb >= 1; jb -= U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = jb - U - u;
            ivn[u] = j >= 1 ? LU(j, 3) : 0.0;
            a4x[u] = j >= 1 ? LU(j, 4) : 0.0;
            a5n[u] = j >= 1 ? LU(j, 5) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = jb - u;
            if (j >= 1) {
                const double r = C(j);
                double xv = ivc[u] * (r - a4c[u] * x1 - a5c[u] * x2);
                if (zero_line) xv = 0.0;
                C(j) = xv;
                x2 = x1; x1 = xv;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { ivc[u] = ivn[u]; a4c[u] = a4x[u]; a5c[u] = a5n[u]; }
    }
#undef LU
#undef C
}

template <int MODE>
__global__ void __launch_bounds__(512) k_y010(double2 *__restrict__ c, const double2 *__restrict__ twg, Y010Arg g)
{
    constexpr int ny = 256;
    extern __shared__ double2 sm[];  // [8][Y010_P] + 256 twiddles
    double2 *__restrict__ tws = sm + 8 * Y010_P;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntx = g.nxs / 8;
    const int kz = blockIdx.x / ntx, x0 = (blockIdx.x % ntx) * 8;
    double2 *__restrict__ base = c + (size_t)kz * ny * g.nxs + x0;
    const int tx = threadIdx.x & 7, tr = threadIdx.x >> 3;  // 64 rows per pass
    {
        double2 v[4];
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = base[(size_t)(tr + 64 * p) * g.nxs + tx];
#pragma unroll
        for (int p = 0; p < 4; p++) sm[tx * Y010_P + tr + 64 * p] = v[p];
    }
    __syncthreads();
    double2 *__restrict__ pen = sm + wave * Y010_P;
    const int ig = x0 + wave;
    const Rot rz{g.az[kz], g.bz[kz], (kz + 1) > g.nz / 2 + 1}, rx{g.ax[ig], g.bx[ig], (ig + 1) > g.nx / 2 + 1};
    double2 a[4];
    if (MODE != 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = pen[lane + 64 * k];
        fft256_wave<-1>(a, pen, tws, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) pen[lane + 64 * k] = a[k];
        wave_lds_fence();
        // rows 0 ; 1..63 with 255..193 ; 64..127 with 192..129 ; 128 (pairs with itself)
        y010_pair_fw(pen, lane, rz, rx, g.ay, g.by, g.nx, g.nz);
        y010_pair_fw(pen, lane + 64, rz, rx, g.ay, g.by, g.nx, g.nz);
        if (lane == 0) y010_pair_fw(pen, 128, rz, rx, g.ay, g.by, g.nx, g.nz);
        wave_lds_fence();
    }
    if (MODE == 2) {
        __syncthreads();
        if (wave == 0 && lane < 32) y010_penta(sm, g, kz, x0, lane);
        __syncthreads();
    }
    if (MODE != 0) {
        y010_pair_bw(pen, lane, rz, rx, g.ay, g.by);
        y010_pair_bw(pen, lane + 64, rz, rx, g.ay, g.by);
        if (lane == 0) y010_pair_bw(pen, 128, rz, rx, g.ay, g.by);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = pen[lane + 64 * k];
        fft256_wave<1>(a, pen, tws, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) pen[lane + 64 * k] = a[k];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; p++) base[(size_t)(tr + 64 * p) * g.nxs + tx] = sm[tx * Y010_P + tr + 64 * p];
}

const double2 *x3d_fft512_twiddles();
int x3d_fft512_init();

// c[nz][256][nxs], x and z already transformed (mode 0, 2) / still transformed (mode 1, 2).  tables = ax bx ay by az bz
// back to back (global lengths nx nx ny ny nz nz).  *done = false: not served (other ny, odd row pitch)
int x3d_y010_run(x3d_backend *b, double2 *c, int nxs, int nx, int ny, int nz, int mode, const double *tables, int sym,
                 double *const lu[2], bool *done)
{
    *done = false;
    if (ny != 256 || nxs % 8 != 0 || nx % 2 != 0) return 0;
    if (mode == 2 && !(lu && lu[0] && (!sym || lu[1]))) return 0;
    if (int rc = x3d_fft512_init()) return rc;
    Y010Arg g;
    g.ax = tables; g.bx = g.ax + nx; g.ay = g.bx + nx; g.by = g.ay + ny; g.az = g.by + ny; g.bz = g.az + nz;
    g.lu0 = lu ? lu[0] : nullptr; g.lu1 = lu ? lu[1] : nullptr;
    g.nxs = nxs; g.nz = nz; g.nx = nx; g.sym = sym;
    const size_t lds = sizeof(double2) * (8 * Y010_P + 256);
    const dim3 grid((unsigned)((size_t)nz * (nxs / 8)));
#define GO(M_) hipLaunchKernelGGL((k_y010<M_>), grid, dim3(512), lds, b->stream, c, x3d_fft512_twiddles(), g)
    if (mode == 0) GO(0); else if (mode == 1) GO(1); else GO(2);
#undef GO
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}
