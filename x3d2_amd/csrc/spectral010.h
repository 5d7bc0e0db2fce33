// Kernels of the non-periodic-y (010) spectral Poisson solve, shared by the single-rank solver (poisson.hip) and the
// z-slab solver that splits the x modes over the ranks (sfft010.hip).  Reference behaviour mirrored (paths under
// /root/reference):
//   src/backend/cuda/kernels/spectral_processing.f90:1062-1114 enforce / undo_periodicity_y
//   src/backend/omp/kernels/spectral_processing.f90:108-283    process_spectral_010
//   src/backend/cuda/kernels/spectral_processing.f90:385-702   _010_fw / _010_poisson / _010_bw
//   src/backend/cuda/poisson_fft.f90:822-924                   fft_postprocess_010
// Every spectral-side array is [nz][ny][nxs], x fastest, rows dense at pitch nxs; a rank of the slab solver holds the
// x modes i0 .. i0 + nxs - 1 of the global problem (single rank: i0 = 0).  Thread indices are flattened over whole
// rows, so a short row (65 modes per rank on 8 ranks) still fills its wavefronts.
#pragma once
#include "common.h"

// even/odd interleave along y on the pitched Cartesian block: out(i, j, k) = in(i, src(j), k)
template <bool UNDO>
static __global__ void __launch_bounds__(256)
    k_periodicity_y(real_t *__restrict__ out, const real_t *__restrict__ in, int nx, int ny, int nz, long nxp,
                    long plane)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y + 1, k = blockIdx.z;  // j 1-based as in the reference
    if (i >= nx) return;
    const int n2 = ny / 2;
    int js, jd;  // source / destination rows, 1-based
    if (!UNDO) {  // :1062-1089
        jd = j;
        if (j <= n2) js = 2 * j - 1;
        else if ((ny & 1) && j == n2 + 1) js = ny;
        else js = 2 * ny - 2 * j + 2;
    } else {      // :1091-1114: out(2j-1) = in(j), out(2j) = in(ny-j+1), odd centre out(ny) = in(n2+1)
        js = j;
        if (j <= n2) jd = 2 * j - 1;
        else if ((ny & 1) && j == n2 + 1) jd = ny;
        else jd = 2 * (ny - j + 1);
    }
    out[(long)k * plane + (long)(jd - 1) * nxp + i] = in[(long)k * plane + (long)(js - 1) * nxp + i];
}

struct Rot { real_t a, b; bool flip; };

__device__ __forceinline__ void rot_fw(real_t &r, real_t &c, const Rot &t)
{
    const real_t tr = r, tc = c;
    r = tr * t.b + tc * t.a;
    c = tc * t.b - tr * t.a;
    if (t.flip) { r = -r; c = -c; }
}
__device__ __forceinline__ void rot_bw(real_t &r, real_t &c, const Rot &t)
{
    const real_t tr = r, tc = c;
    r = tr * t.b - tc * t.a;
    c = tc * t.b + tr * t.a;
    if (t.flip) { r = -r; c = -c; }
}

// One thread per (i, row pair, k): rows j and jr = ny-j+2 (j = 1 has no partner; for even ny the
// middle row pairs with itself and keeps the second store, like the reference).
// MODE 0: fw only (normalise, z/x rotations, y split)            -> stretched path, step 1
// MODE 1: bw only (y recombination, z/x inverse rotations)       -> stretched path, step 3
// MODE 2: fw, -1/waves, bw fused (uniform y): 1R + 1W of c, 1R of waves
// ZROT = false (Poisson 110 on the z-first transposed problem): the third direction is not periodic either -- no
// rotation along it here, it gets its own paired split (k_spectral_pair_z)
// ax, bx: the GLOBAL tables (indexed i0 + i); nx, ny, nz: global cell counts
// ZF: the arrays are [ny][nxs][nz], z fastest (the slab solver's z-contiguous copy, sfft010.hip) instead of [nz][ny][nxs]
template <int MODE, bool ZROT = true, bool ZF = false>
static __global__ void __launch_bounds__(256)
    k_spectral_010(real2_t *__restrict__ c, const real_t *__restrict__ waves, int nxs, int ny, int nz, int nx, int i0,
                   const real_t *__restrict__ ax, const real_t *__restrict__ bx, const real_t *__restrict__ ay,
                   const real_t *__restrict__ by, const real_t *__restrict__ az, const real_t *__restrict__ bz)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int i, j, k;
    if (ZF) {
        k = (int)(t % nz); i = (int)(t / nz); j = blockIdx.y + 1;
        if (i >= nxs) return;
    } else {
        i = (int)(t % nxs); j = (int)(t / nxs) + 1; k = blockIdx.y;  // j = 1 .. ny/2+1
        if (j > ny / 2 + 1) return;
    }
    const int ig = i0 + i;
    const int jr = ny - j + 2;
    const bool paired = j >= 2, self = paired && jr == j;
    const size_t il = ZF ? ((size_t)(j - 1) * nxs + i) * nz + k : ((size_t)k * ny + (j - 1)) * nxs + i;
    const size_t ir = !paired ? il : (ZF ? ((size_t)(jr - 1) * nxs + i) * nz + k : ((size_t)k * ny + (jr - 1)) * nxs + i);
    const Rot rz{az[k], bz[k], (k + 1) > nz / 2 + 1}, rx{ax[ig], bx[ig], (ig + 1) > nx / 2 + 1};
    real2_t L = c[il], R = paired && !self ? c[ir] : L;
    real_t l_r = L.x, l_c = L.y, r_r = R.x, r_c = R.y;
    if (MODE != 1) {
        l_r = l_r / nx / ny / nz; l_c = l_c / nx / ny / nz;
        if (ZROT) rot_fw(l_r, l_c, rz);
        rot_fw(l_r, l_c, rx);
        if (self) { r_r = l_r; r_c = l_c; }
        else if (paired) {
            r_r = r_r / nx / ny / nz; r_c = r_c / nx / ny / nz;
            if (ZROT) rot_fw(r_r, r_c, rz);
            rot_fw(r_r, r_c, rx);
        }
        if (paired) {
            const real_t a = ay[j - 1], b = by[j - 1], a2 = ay[jr - 1], b2 = by[jr - 1];
            const real_t n_lr = 0.5 * (l_r * b + l_c * a + r_r * b - r_c * a);
            const real_t n_lc = 0.5 * (-l_r * a + l_c * b + r_r * a + r_c * b);
            const real_t n_rr = 0.5 * (r_r * b2 + r_c * a2 + l_r * b2 - l_c * a2);
            const real_t n_rc = 0.5 * (-r_r * a2 + r_c * b2 + l_r * a2 + l_c * b2);
            l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
            if (self) { l_r = r_r; l_c = r_c; }  // the second store wins on the self-paired row
        }
    }
    if (MODE == 2) {
        const bool zero_line = (ig + 1) == nx / 2 + 1 && (k + 1) == nz / 2 + 1;
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
    if (MODE != 0) {
        if (paired) {
            if (self) { r_r = l_r; r_c = l_c; }
            const real_t a = ay[j - 1], b = by[j - 1], a2 = ay[jr - 1], b2 = by[jr - 1];
            const real_t n_lr = l_r * b - l_c * a + r_r * a + r_c * b;
            const real_t n_lc = l_r * a + l_c * b - r_r * b + r_c * a;
            const real_t n_rr = r_r * b2 - r_c * a2 + l_r * a2 + l_c * b2;
            const real_t n_rc = r_r * a2 + r_c * b2 - l_r * b2 + l_c * a2;
            l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
            if (self) { l_r = r_r; l_c = r_c; }
        }
        if (ZROT) rot_bw(l_r, l_c, rz);
        rot_bw(l_r, l_c, rx);
        if (paired && !self) {
            if (ZROT) rot_bw(r_r, r_c, rz);
            rot_bw(r_r, r_c, rx);
        }
    }
    c[il] = make_real2(l_r, l_c);
    if (paired && !self) c[ir] = make_real2(r_r, r_c);
}

static inline dim3 spectral_010_grid(int nxs, int ny, int nz, bool zf = false)
{
    if (zf) return dim3((unsigned)(((long)nxs * nz + 255) / 256), (unsigned)(ny / 2 + 1));
    return dim3((unsigned)(((long)nxs * (ny / 2 + 1) + 255) / 256), (unsigned)nz);
}

// Pentadiagonal operators: the reference eliminates the matrix in place at EVERY solve
// (and re-copies it from a store first, src/backend/cuda/poisson_fft.f90:868-913).  The
// elimination does not depend on the right-hand side, so it is done once here with the
// reference's operation order and only what the right-hand side needs is kept, in the
// matrix's own storage ([d][k][j][i], d = diagonal slot):
//   slot 0 row j : m2_j  multiplier of row j into row j+2          (j <= n-2)
//   slot 1 row j : m1_j  multiplier of row j into row j+1          (j <= n-1; j = n-1: last-row tmp)
//   slot 2 row j : 1/a3_j (0 where |a3_j| <= eps)   (j <= n-1);  row n: the last pivot dd
//   slot 3 row j : a4_j eliminated                                  (j <= n-1)
//   slot 4 row j : a5_j                                             (j <= n-2)
// so that the solve applies bit-for-bit the same operations to the right-hand side.
template <bool ZF = false>  // ZF: [5][n][nxs][nz] instead of [5][nz][n][nxs]
static __global__ void __launch_bounds__(64) k_penta_factor(real_t *__restrict__ a, int nxs, int n, int nz)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = ZF ? (int)(t / nz) : (int)(t % nxs), k = ZF ? (int)(t % nz) : (int)(t / nxs);
    if (ZF ? i >= nxs : k >= nz) return;
    const real_t eps = 1.e-16;
    const size_t ds = (size_t)nz * n * nxs;
#define A(j, d)                                                                                              \
    a[(size_t)((d) - 1) * ds + (ZF ? ((size_t)((j) - 1) * nxs + i) * nz + k : ((size_t)k * n + ((j) - 1)) * nxs + i)]
    for (int j = 1; j <= n - 2; j++) {
        const real_t a3 = A(j, 3), a4 = A(j, 4), a5 = A(j, 5);
        const real_t m1 = fabs(a3) > eps ? A(j + 1, 2) / a3 : 0.0;
        A(j + 1, 3) = A(j + 1, 3) - m1 * a4;
        A(j + 1, 4) = A(j + 1, 4) - m1 * a5;
        const real_t m2 = fabs(a3) > eps ? A(j + 2, 1) / a3 : 0.0;
        A(j + 2, 2) = A(j + 2, 2) - m2 * a4;
        A(j + 2, 3) = A(j + 2, 3) - m2 * a5;
        A(j, 1) = m2;
        A(j, 2) = m1;
        A(j, 3) = fabs(a3) > eps ? 1.0 / a3 : 0.0;
    }
    const real_t a3 = A(n - 1, 3);
    const real_t tmp = fabs(a3) > eps ? A(n, 2) / a3 : 0.0;
    const real_t dd = A(n, 3) - tmp * A(n - 1, 4);
    A(n - 1, 2) = tmp;
    A(n - 1, 3) = fabs(a3) > eps ? 1.0 / a3 : 0.0;
    A(n, 3) = dd;
#undef A
}

// One thread per (i, k): coalesced along i.  Per spectral entry: forward r/w of the rhs + m1, m2;
// backward r/w + 1/a3, a4, a5 = 104 B.
template <bool ZF = false>
static __global__ void __launch_bounds__(64)
    k_penta_solve(real2_t *__restrict__ c, const real_t *__restrict__ lu, int off, int inc, int nxs, int ny,
                  int nz, int n, int nx, int i0)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = ZF ? (int)(t / nz) : (int)(t % nxs), k = ZF ? (int)(t % nz) : (int)(t / nxs);
    if (ZF ? i >= nxs : k >= nz) return;
    const real_t eps = 1.e-16;
    const size_t ds = (size_t)nz * n * nxs;
#define LU(j, d)                                                                                             \
    lu[(size_t)((d) - 1) * ds + (ZF ? ((size_t)((j) - 1) * nxs + i) * nz + k : ((size_t)k * n + ((j) - 1)) * nxs + i)]
#define C(jm) c[ZF ? ((size_t)((jm) - 1) * nxs + i) * nz + k : ((size_t)k * ny + ((jm) - 1)) * nxs + i]
    const int h = inc / 2;
    // forward: rows j+1, j+2 -= m * row j; two rows are carried in registers
    real2_t r0 = C(inc * 1 + off - h), r1 = C(inc * 2 + off - h);
    for (int j = 1; j <= n - 2; j++) {
        const int jm = inc * j + off - h;
        real2_t r2 = C(jm + 2 * inc);
        const real_t m1 = LU(j, 2), m2 = LU(j, 1);
        r1.x = r1.x - m1 * r0.x; r1.y = r1.y - m1 * r0.y;
        r2.x = r2.x - m2 * r0.x; r2.y = r2.y - m2 * r0.y;
        C(jm) = r0;
        r0 = r1; r1 = r2;
    }
    // last two rows: r0 = row n-1, r1 = row n
    const int nm = inc * n + off - h;
    const real_t tmp = LU(n - 1, 2), dd = LU(n, 3), inv = LU(n - 1, 3), a4n = LU(n - 1, 4);
    real2_t xn, xn1;
    if (fabs(dd) > eps) {
        const real_t tt = tmp / dd;
        xn.x = r1.x / dd - tt * r0.x;
        xn.y = r1.y / dd - tt * r0.y;
    } else {
        xn.x = 0.0; xn.y = 0.0;
    }
    const real_t q = a4n * inv;
    xn1.x = r0.x * inv - xn.x * q;
    xn1.y = r0.y * inv - xn.y * q;
    const bool zero_line = (i0 + i + 1) == nx / 2 + 1 && (k + 1) == nz / 2 + 1;
    if (zero_line) { xn = make_real2(0.0, 0.0); xn1 = make_real2(0.0, 0.0); }
    C(nm) = xn;
    C(nm - inc) = xn1;
    // backward
    real2_t x1 = xn1, x2 = xn;
    for (int j = n - 2; j >= 1; j--) {
        const int jm = inc * j + off - h;
        const real2_t r = C(jm);
        const real_t iv = LU(j, 3), a4 = LU(j, 4), a5 = LU(j, 5);
        real2_t x;
        x.x = iv * (r.x - a4 * x1.x - a5 * x2.x);
        x.y = iv * (r.y - a4 * x1.y - a5 * x2.y);
        if (zero_line) x = make_real2(0.0, 0.0);
        C(jm) = x;
        x2 = x1; x1 = x;
    }
#undef LU
#undef C
}

static inline dim3 penta_grid(int nxs, int nz) { return dim3((unsigned)(((long)nxs * nz + 63) / 64)); }

// fft_postprocess_010 (src/backend/cuda/poisson_fft.f90:822-924) on c[nz][ny][nxs] (zf: c[ny][nxs][nz]) = the x modes
// i0.. of the global problem: uniform y: the whole of process_spectral_010 in one kernel; stretched y: fw ;
// pentadiagonal solves on the odd rows and the even rows (sym) or on all rows ; bw.  ax .. bz: the global tables
template <bool ZF>
static inline int spectral_010_launch_t(hipStream_t st, real2_t *c, const real_t *waves, int nxs, int nx, int ny, int nz,
                                        int i0, const real_t *ax, const real_t *bx, const real_t *ay, const real_t *by,
                                        const real_t *az, const real_t *bz, int stretched, int sym, real_t *const lu[2])
{
    const dim3 grid = spectral_010_grid(nxs, ny, nz, ZF);
#define SPEC(M_)                                                                                                    \
    hipLaunchKernelGGL((k_spectral_010<M_, true, ZF>), grid, dim3(256), 0, st, c, waves, nxs, ny, nz, nx, i0, ax, bx, ay, by, \
                       az, bz)
    if (!stretched) {
        SPEC(2);
    } else {
        SPEC(0);
        const dim3 g2 = penta_grid(nxs, nz);
        if (sym) {  // odd rows, then even rows (src/backend/cuda/poisson_fft.f90:880-895)
            hipLaunchKernelGGL(k_penta_solve<ZF>, g2, dim3(64), 0, st, c, lu[0], 0, 2, nxs, ny, nz, ny / 2, nx, i0);
            hipLaunchKernelGGL(k_penta_solve<ZF>, g2, dim3(64), 0, st, c, lu[1], 1, 2, nxs, ny, nz, ny / 2, nx, i0);
        } else {
            hipLaunchKernelGGL(k_penta_solve<ZF>, g2, dim3(64), 0, st, c, lu[0], 0, 1, nxs, ny, nz, ny, nx, i0);
        }
        SPEC(1);
    }
#undef SPEC
    X3D_HIP(hipGetLastError());
    return 0;
}

// tables = ax bx ay by az bz back to back (global lengths nx nx ny ny nz nz), arrays [nz][ny][nxs]
static inline int spectral_010_launch(hipStream_t st, real2_t *c, const real_t *waves, int nxs, int nx, int ny, int nz, int i0,
                                      const real_t *tables, int stretched, int sym, real_t *const lu[2])
{
    const real_t *ax = tables, *bx = ax + nx, *ay = bx + nx, *by = ay + ny, *az = by + ny, *bz = az + nz;
    return spectral_010_launch_t<false>(st, c, waves, nxs, nx, ny, nz, i0, ax, bx, ay, by, az, bz, stretched, sym, lu);
}
