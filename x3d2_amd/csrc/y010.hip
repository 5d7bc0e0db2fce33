// K6y: the y pass of the non-periodic-y (010) Poisson solve for ny = 256 cells on a stretched grid (the channel case,
// BASELINE configs[4]).  The reference's fft_postprocess_010 (src/backend/cuda/poisson_fft.f90:822-924) after a 3-D
// transform is: process_spectral_010's forward part ; pentadiagonal solves along y (odd rows, even rows) ; backward part
// (src/backend/cuda/kernels/spectral_processing.f90:385-702; spectral010.h has them as k_spectral_010<0>, k_penta_solve,
// k_spectral_010<1>) -- with the y stages of the transforms, six passes over the 1.07 GB spectrum, 232 B per entry.
// The 3-D DFT is separable: here x and z are transformed first (one 2-D rocFFT plan batched over the y rows, poisson.hip)
// and y LAST, so that everything that couples the rows of one (x mode, z mode) column -- the y transform, the paired split
// of rows j and ny - j + 2, the pentadiagonal systems -- happens while the column is on chip.
//
// Work decomposition: a workgroup of 8 waves owns the 8 x-adjacent columns of one z mode: c[kz][0..255][x0..x0+7]
// (128-byte row segments).  The tile sits in LDS as 8 pencils of Y010_P real2_t; wave w transforms pencil w with
// fft256_wave (fft512_core.h, 4 points per lane, exchanges through the pencil's own LDS region) and does the paired
// split on it; only the load, the pentadiagonal phase and the store need block barriers.
//
// Forms (x3d_poisson_solve_010_rows, X3D_Y010_FORM; channel bench 1024 x 257 x 512, profiles/r04_channel_round4.txt):
//   "3d"     X3D_NO_Y010=1: 3-D transforms + the four kernels                    fft + spectral classes 14.9 ms per step
//   "split"  k_y010<0> (y transform + forward part) ; k_penta_solve x 2 ; k_y010<1> (backward part + inverse y transform)
//            168 B per entry                                                                               12.8 ms
//   "staged" k_y010<3> = <0> + the FORWARD sweeps of both systems on the tile ; k_y010<4> = the BACKWARD sweeps + <1>:
//            104 B per entry (the spectrum twice, the factored operator once).  The operator's rows of the tile are
//            requested by all 512 threads at kernel start and staged in LDS (<3>: 2 diagonals, 33 KB; <4>: 3 diagonals in
//            two halves of 25 KB); wave 0 runs the 32 chains (x 0..7, re / im, odd / even system), k_penta_solve's
//            arithmetic operation by operation.  DEFAULT                                                    11.4 ms
//            (without the chains -- wrong results, timing only -- 9.9: they cost 0.5 ms per solve; 2 workgroups per CU)
//   "fused"  k_y010<2>: everything in one kernel, 72 B per entry, the chains stream the operator from memory themselves:
//            23.1 ms -- 32 lanes wait a full memory latency per 8-row chunk, 32 times per tile.  Kept as the measured
//            negative; parity-green like the others.
#include <type_traits>

#include "fft512_core.h"
#include "spectral010.h"

#define Y010_P 274  // >= 271 (fft256_wave's layouts) and = 2 mod 16: the 8 pencils' rows j sit in different LDS banks

struct Y010Arg {
    const real_t *ax, *bx, *ay, *by, *az, *bz;  // global tables (spectral010.h)
    const real_t *lu0, *lu1;                    // factored pentadiagonal operators [5][nz][n][nxs]
    int nxs, nz, nx, sym;
    int db;   // 1: the double-buffered sweeps (round 6), 0: the chunk-by-chunk ones
    int nzl;  // z planes held by c and by the factored operators: nz (x-first layout, x modes 0 .. nx / 2) or nz / 2 + 1 (the
              // z-first layout, csrc/zfirst.hip: ALL x modes -- the rotation's flip above nx / 2 then applies along x)
};

__device__ __forceinline__ void y010_pair_fw(real2_t *__restrict__ pen, int j0, const Rot &rz, const Rot &rx,
                                             const real_t *__restrict__ ay, const real_t *__restrict__ by, int nx, int nz)
{
    constexpr int ny = 256;
    const int jr0 = ny - j0;  // (0-based partner row; j0 = 0 has none)
    const bool paired = j0 >= 1, self = paired && jr0 == j0;
    const real2_t L = pen[j0], R = paired && !self ? pen[jr0] : L;
    real_t l_r = L.x, l_c = L.y, r_r = R.x, r_c = R.y;
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
        const real_t a = ay[j0], b = by[j0], a2 = ay[jr0], b2 = by[jr0];
        const real_t n_lr = 0.5 * (l_r * b + l_c * a + r_r * b - r_c * a);
        const real_t n_lc = 0.5 * (-l_r * a + l_c * b + r_r * a + r_c * b);
        const real_t n_rr = 0.5 * (r_r * b2 + r_c * a2 + l_r * b2 - l_c * a2);
        const real_t n_rc = 0.5 * (-r_r * a2 + r_c * b2 + l_r * a2 + l_c * b2);
        l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
        if (self) { l_r = r_r; l_c = r_c; }  // the second store wins on the self-paired row
    }
    pen[j0] = make_real2(l_r, l_c);
    if (paired && !self) pen[jr0] = make_real2(r_r, r_c);
}

__device__ __forceinline__ void y010_pair_bw(real2_t *__restrict__ pen, int j0, const Rot &rz, const Rot &rx,
                                             const real_t *__restrict__ ay, const real_t *__restrict__ by)
{
    constexpr int ny = 256;
    const int jr0 = ny - j0;
    const bool paired = j0 >= 1, self = paired && jr0 == j0;
    const real2_t L = pen[j0], R = paired && !self ? pen[jr0] : L;
    real_t l_r = L.x, l_c = L.y, r_r = R.x, r_c = R.y;
    if (paired) {
        if (self) { r_r = l_r; r_c = l_c; }
        const real_t a = ay[j0], b = by[j0], a2 = ay[jr0], b2 = by[jr0];
        const real_t n_lr = l_r * b - l_c * a + r_r * a + r_c * b;
        const real_t n_lc = l_r * a + l_c * b - r_r * b + r_c * a;
        const real_t n_rr = r_r * b2 - r_c * a2 + l_r * a2 + l_c * b2;
        const real_t n_rc = r_r * a2 + r_c * b2 - l_r * b2 + l_c * a2;
        l_r = n_lr; l_c = n_lc; r_r = n_rr; r_c = n_rc;
        if (self) { l_r = r_r; l_c = r_c; }
    }
    rot_bw(l_r, l_c, rz);
    rot_bw(l_r, l_c, rx);
    if (paired && !self) {
        rot_bw(r_r, r_c, rz);
        rot_bw(r_r, r_c, rx);
    }
    pen[j0] = make_real2(l_r, l_c);
    if (paired && !self) pen[jr0] = make_real2(r_r, r_c);
}

// k_penta_solve (spectral010.h) on the tile in LDS: this lane's chain = component `ri` of column x, system s
// (sym: rows 2 j + s - 2, 0-based, j = 1 .. n = 128; else all 256 rows)
__device__ __forceinline__ void y010_penta(real2_t *__restrict__ sm, const Y010Arg &g, int kz, int x0, int lane)
{
    constexpr int ny = 256, U = 8;
    const int x = lane & 7, ri = (lane >> 3) & 1, s = lane >> 4;
    if (s >= (g.sym ? 2 : 1)) return;
    const int inc = g.sym ? 2 : 1, n = ny / inc;
    const real_t *__restrict__ lu = s ? g.lu1 : g.lu0;
    const size_t ds = (size_t)g.nzl * n * g.nxs;
    const real_t *__restrict__ lub = lu + (size_t)kz * n * g.nxs + x0 + x;
#define LU(j, d) lub[(size_t)((d) - 1) * ds + (size_t)((j) - 1) * g.nxs]
    real_t *__restrict__ pd = reinterpret_cast<real_t *>(sm + x * Y010_P) + ri;
#define C(j) pd[2 * (inc * (j) + s - inc)]  // (inc j + off - h - 1 with off = s, h = inc / 2: 2 j + s - 2 or j - 1)
    const real_t eps = 1.e-16;
    // forward: rows j+1, j+2 -= m * row j; two rows are carried in registers
    real_t r0 = C(1), r1 = C(2);
    real_t m1c[U], m2c[U], m1n[U], m2n[U];
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
                real_t r2 = C(j + 2);
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
    const real_t tmp = LU(n - 1, 2), dd = LU(n, 3), inv = LU(n - 1, 3), a4n = LU(n - 1, 4);
    // (the first backward chunk, requested before the divisions)
    real_t ivc[U], a4c[U], a5c[U], ivn[U], a4x[U], a5n[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int j = n - 2 - u;
        ivc[u] = j >= 1 ? LU(j, 3) : 0.0;
        a4c[u] = j >= 1 ? LU(j, 4) : 0.0;
        a5c[u] = j >= 1 ? LU(j, 5) : 0.0;
    }
    real_t xn, xn1;
    if (fabs(dd) > eps) {
        const real_t tt = tmp / dd;
        xn = r1 / dd - tt * r0;
    } else {
        xn = 0.0;
    }
    const real_t q = a4n * inv;
    xn1 = r0 * inv - xn * q;
    const bool zero_line = (x0 + x + 1) == g.nx / 2 + 1 && (kz + 1) == g.nz / 2 + 1;
    if (zero_line) { xn = 0.0; xn1 = 0.0; }
    C(n) = xn;
    C(n - 1) = xn1;
    // backward
    real_t x1 = xn1, x2 = xn;
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
                const real_t r = C(j);
                real_t xv = ivc[u] * (r - a4c[u] * x1 - a5c[u] * x2);
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

// ---- MODES 3 / 4: the sweeps of k_penta_solve on the tile with the operator STAGED in LDS by all 512 threads.
// Row slots: R = s n + (j - 1) in 0 .. 255 (sym: two systems of n = 128 rows; else one of 256); a thread stages
// (x = t & 7, R = (t >> 3) + 64 h, h = 0 .. 3) of every diagonal: one 64-byte segment per row.
// A slot = 8 doubles, 8 more between the two systems (their chains read rows R and R + n together).
// One wave runs the 32 chains and nothing else runs beside it in the workgroup: the loops are written for a short
// instruction stream (constant strides, a chunk's operands requested together, no register real_t buffer).
#define Y010_LD (256 * 8 + 16)  // doubles per staged diagonal (MODE 3: all rows of 2 diagonals)
#define Y010_LH (128 * 8 + 16)  // MODE 4: half the rows of 3 diagonals at a time

struct Y010Chain {
    int x, s;
    real_t *pd;  // row 1 of this chain's component of column x in the tile (rows 16 inc bytes apart)
    bool on, zero_line;
};
template <bool SYM>
__device__ __forceinline__ Y010Chain y010_chain(real2_t *__restrict__ sm, const Y010Arg &g, int kz, int x0, int lane, int wave)
{
    Y010Chain c;
    c.x = lane & 7;
    c.s = (lane >> 4) & 1;
#ifdef Y010_NO_SWEEP  // (timing experiment: what the two kernels take without their chains)
    c.on = false;
#else
    c.on = wave == 0 && lane < 32 && c.s < (SYM ? 2 : 1);
#endif
    c.pd = reinterpret_cast<real_t *>(sm + c.x * Y010_P) + ((lane >> 3) & 1) + 2 * c.s;
    c.zero_line = (x0 + c.x + 1) == g.nx / 2 + 1 && (kz + 1) == g.nz / 2 + 1;
    return c;
}

// forward sweep + the last two rows; lf = [2][Y010_LD]: diagonal 1 (m2), diagonal 2 (m1); tl = LU(n-1,2), LU(n,3),
// LU(n-1,3), LU(n-1,4) of this chain (loaded by the caller at kernel start).
// Round 6: chunks of 4 rows, DOUBLE BUFFERED -- the operands of the next chunk (its multipliers and the right-hand side
// rows two ahead, which no earlier row has written) are requested before this chunk's dependent updates run, so that an
// LDS round trip no longer stands between the chunks of the one wave everything else waits for.  Same operations in the
// same order on every row (k_penta_solve's, spectral010.h).
template <bool SYM>
__device__ __forceinline__ void y010_sweep_fw(const Y010Chain &c, const real_t *__restrict__ lf, const real_t (&tl)[4])
{
    constexpr int U = 4, n = SYM ? 128 : 256, RS = SYM ? 4 : 2;  // RS: doubles between a chain's rows in the tile
    const real_t *__restrict__ m2p = lf + (c.s * n) * 8 + c.s * 8 + c.x;  // row j at m2p[(j - 1) * 8]
    const real_t *__restrict__ m1p = m2p + Y010_LD;
    real_t *__restrict__ cp = c.pd;  // row j at cp[(j - 1) * RS]
    real_t r0 = cp[0], r1 = cp[RS];
    real_t m1a[U], m2a[U], r2a[U], m1b[U], m2b[U], r2b[U];
    auto load = [&](real_t (&m1c)[U], real_t (&m2c)[U], real_t (&r2c)[U], int jb) {  // rows jb .. jb + U - 1
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb + u <= n - 2) {
                m2c[u] = m2p[(jb - 1 + u) * 8];
                m1c[u] = m1p[(jb - 1 + u) * 8];
                r2c[u] = cp[(jb - 1 + u + 2) * RS];
            }
        }
    };
    auto run = [&](const real_t (&m1c)[U], const real_t (&m2c)[U], const real_t (&r2c)[U], int jb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb + u <= n - 2) {
                real_t r2 = r2c[u];
                r1 = r1 - m1c[u] * r0;
                r2 = r2 - m2c[u] * r0;
                cp[(jb - 1 + u) * RS] = r0;
                r0 = r1; r1 = r2;
            }
        }
    };
    load(m1a, m2a, r2a, 1);
    for (int jb = 1; jb <= n - 2; jb += 2 * U) {
        load(m1b, m2b, r2b, jb + U);
        run(m1a, m2a, r2a, jb);
        load(m1a, m2a, r2a, jb + 2 * U);
        run(m1b, m2b, r2b, jb + U);
    }
    const real_t eps = 1.e-16;
    const real_t tmp = tl[0], dd = tl[1], inv = tl[2], a4n = tl[3];
    real_t xn, xn1;
    if (fabs(dd) > eps) {
        const real_t tt = tmp / dd;
        xn = r1 / dd - tt * r0;
    } else {
        xn = 0.0;
    }
    const real_t q = a4n * inv;
    xn1 = r0 * inv - xn * q;
    if (c.zero_line) { xn = 0.0; xn1 = 0.0; }
    cp[(n - 1) * RS] = xn;
    cp[(n - 2) * RS] = xn1;
}

// rows jhi .. jlo (descending) of the backward sweep; lb = [3][Y010_LH]: diagonals 3 (1/a3), 4, 5 of the rows
// jbase + 1 .. jbase + n/2 (slot = s n/2 + j - jbase - 1); x1, x2 carried by the caller.  Chunks of 4 rows, double buffered
// like the forward sweep (a row's right-hand side is read before any later row of the sweep writes: rows below are untouched)
template <bool SYM>
__device__ __forceinline__ void y010_sweep_bw(const Y010Chain &c, const real_t *__restrict__ lb, int jhi, int jlo, int jbase,
                                              real_t &x1, real_t &x2)
{
    constexpr int U = 4, n = SYM ? 128 : 256, RS = SYM ? 4 : 2;
    const real_t *__restrict__ ivp = lb + (c.s * (n / 2) - jbase - 1) * 8 + c.s * 8 + c.x;  // row j at ivp[j * 8]
    const real_t *__restrict__ a4p = ivp + Y010_LH, *__restrict__ a5p = ivp + 2 * Y010_LH;
    real_t *__restrict__ cp = c.pd - RS;  // row j at cp[j * RS]
    real_t iva[U], a4a[U], a5a[U], ra[U], ivb[U], a4b[U], a5b[U], rb[U];
    auto load = [&](real_t (&iv)[U], real_t (&a4)[U], real_t (&a5)[U], real_t (&r)[U], int jb) {  // rows jb, jb - 1, ..
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb - u >= jlo) {
                iv[u] = ivp[(jb - u) * 8];
                a4[u] = a4p[(jb - u) * 8];
                a5[u] = a5p[(jb - u) * 8];
                r[u] = cp[(jb - u) * RS];
            }
        }
    };
    auto run = [&](const real_t (&iv)[U], const real_t (&a4)[U], const real_t (&a5)[U], const real_t (&r)[U], int jb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb - u >= jlo) {
                real_t xv = iv[u] * (r[u] - a4[u] * x1 - a5[u] * x2);
                if (c.zero_line) xv = 0.0;
                cp[(jb - u) * RS] = xv;
                x2 = x1; x1 = xv;
            }
        }
    };
    load(iva, a4a, a5a, ra, jhi);
    for (int jb = jhi; jb >= jlo; jb -= 2 * U) {
        load(ivb, a4b, a5b, rb, jb - U);
        run(iva, a4a, a5a, ra, jb);
        load(iva, a4a, a5a, ra, jb - 2 * U);
        run(ivb, a4b, a5b, rb, jb - U);
    }
}

// ---- the sweeps of rounds 4-5 (8-row chunks, operands requested chunk by chunk): kept for the A/B (X3D_Y010_NO_DB=1)
// forward sweep + the last two rows; lf = [2][Y010_LD]: diagonal 1 (m2), diagonal 2 (m1); tl = LU(n-1,2), LU(n,3),
// LU(n-1,3), LU(n-1,4) of this chain (loaded by the caller at kernel start)
template <bool SYM>
__device__ __forceinline__ void y010_sweep_fw_plain(const Y010Chain &c, const real_t *__restrict__ lf, const real_t (&tl)[4])
{
    constexpr int U = 8, n = SYM ? 128 : 256, RS = SYM ? 4 : 2;  // RS: doubles between a chain's rows in the tile
    const real_t *__restrict__ m2p = lf + (c.s * n) * 8 + c.s * 8 + c.x;  // row j at m2p[(j - 1) * 8]
    const real_t *__restrict__ m1p = m2p + Y010_LD;
    real_t *__restrict__ cp = c.pd;  // row j at cp[(j - 1) * RS]
    real_t r0 = cp[0], r1 = cp[RS];
    for (int jb = 1; jb <= n - 2; jb += U) {
        real_t m1c[U], m2c[U], r2c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb + u <= n - 2) {
                m2c[u] = m2p[u * 8];
                m1c[u] = m1p[u * 8];
                r2c[u] = cp[(u + 2) * RS];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb + u <= n - 2) {
                real_t r2 = r2c[u];
                r1 = r1 - m1c[u] * r0;
                r2 = r2 - m2c[u] * r0;
                cp[u * RS] = r0;
                r0 = r1; r1 = r2;
            }
        }
        m2p += U * 8; m1p += U * 8; cp += U * RS;
    }
    cp = c.pd;
    const real_t eps = 1.e-16;
    const real_t tmp = tl[0], dd = tl[1], inv = tl[2], a4n = tl[3];
    real_t xn, xn1;
    if (fabs(dd) > eps) {
        const real_t tt = tmp / dd;
        xn = r1 / dd - tt * r0;
    } else {
        xn = 0.0;
    }
    const real_t q = a4n * inv;
    xn1 = r0 * inv - xn * q;
    if (c.zero_line) { xn = 0.0; xn1 = 0.0; }
    cp[(n - 1) * RS] = xn;
    cp[(n - 2) * RS] = xn1;
}

// rows jhi .. jlo (descending) of the backward sweep; lb = [3][Y010_LH]: diagonals 3 (1/a3), 4, 5 of the rows
// jbase + 1 .. jbase + n/2 (slot = s n/2 + j - jbase - 1); x1, x2 carried by the caller
template <bool SYM>
__device__ __forceinline__ void y010_sweep_bw_plain(const Y010Chain &c, const real_t *__restrict__ lb, int jhi, int jlo, int jbase,
                                              real_t &x1, real_t &x2)
{
    constexpr int U = 8, n = SYM ? 128 : 256, RS = SYM ? 4 : 2;
    const real_t *__restrict__ ivp = lb + (c.s * (n / 2) + jhi - jbase - 1) * 8 + c.s * 8 + c.x;  // row jhi - u at ivp[-u * 8]
    const real_t *__restrict__ a4p = ivp + Y010_LH, *__restrict__ a5p = ivp + 2 * Y010_LH;
    real_t *__restrict__ cp = c.pd + (jhi - 1) * RS;
    for (int jb = jhi; jb >= jlo; jb -= U) {
        real_t ivc[U], a4c[U], a5c[U], rc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb - u >= jlo) {
                ivc[u] = ivp[-u * 8];
                a4c[u] = a4p[-u * 8];
                a5c[u] = a5p[-u * 8];
                rc[u] = cp[-u * RS];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb - u >= jlo) {
                real_t xv = ivc[u] * (rc[u] - a4c[u] * x1 - a5c[u] * x2);
                if (c.zero_line) xv = 0.0;
                cp[-u * RS] = xv;
                x2 = x1; x1 = xv;
            }
        }
        ivp -= U * 8; a4p -= U * 8; a5p -= U * 8; cp -= U * RS;
    }
}

template <int MODE>
__global__ void __launch_bounds__(512) k_y010(real2_t *__restrict__ c, const real2_t *__restrict__ twg, Y010Arg g)
{
    constexpr int ny = 256;
    extern __shared__ real2_t sm[];  // [8][Y010_P] + 256 twiddles
    real2_t *__restrict__ tws = sm + 8 * Y010_P;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntx = g.nxs / 8;
    const int kz = blockIdx.x / ntx, x0 = (blockIdx.x % ntx) * 8;
    real2_t *__restrict__ base = c + (size_t)kz * ny * g.nxs + x0;
    const int tx = threadIdx.x & 7, tr = threadIdx.x >> 3;  // 64 rows per pass
    // MODES 3 / 4: this thread's share of the factored operator, requested first (in flight during the transforms)
    real_t *__restrict__ lst = reinterpret_cast<real_t *>(tws + 256);
    constexpr int ND = MODE == 3 ? 2 : 3;
    real_t lreg[MODE >= 3 ? ND * 4 : 1];
    real_t tl[4] = {0.0, 0.0, 0.0, 0.0};
    const int inc_ = g.sym ? 2 : 1, n_ = ny / inc_;
    const size_t ds_ = (size_t)g.nzl * n_ * g.nxs;
    if constexpr (MODE >= 3) {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int R = tr + 64 * h, s_ = R / n_, j = R % n_ + 1;
            const real_t *__restrict__ lu = s_ ? g.lu1 : g.lu0;
#pragma unroll
            for (int d = 0; d < ND; d++) {
                const int diag = MODE == 3 ? d : d + 2;  // (0-based: 0 m2, 1 m1, 2 1/a3, 3 a4, 4 a5)
                lreg[d * 4 + h] = j <= n_ - 2 ? lu[(size_t)diag * ds_ + ((size_t)kz * n_ + (j - 1)) * g.nxs + x0 + tx] : 0.0;
            }
        }
        if (MODE == 3 && wave == 0 && lane < 32 && ((lane >> 4) & 1) < (g.sym ? 2 : 1)) {
            const real_t *__restrict__ lu = ((lane >> 4) & 1) ? g.lu1 : g.lu0;
            const real_t *__restrict__ lb_ = lu + (size_t)kz * n_ * g.nxs + x0 + (lane & 7);
            tl[0] = lb_[(size_t)1 * ds_ + (size_t)(n_ - 2) * g.nxs];  // LU(n-1, 2)
            tl[1] = lb_[(size_t)2 * ds_ + (size_t)(n_ - 1) * g.nxs];  // LU(n, 3)
            tl[2] = lb_[(size_t)2 * ds_ + (size_t)(n_ - 2) * g.nxs];  // LU(n-1, 3)
            tl[3] = lb_[(size_t)3 * ds_ + (size_t)(n_ - 2) * g.nxs];  // LU(n-1, 4)
        }
    }
    {
        real2_t v[4];
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = base[(size_t)(tr + 64 * p) * g.nxs + tx];
#pragma unroll
        for (int p = 0; p < 4; p++) sm[tx * Y010_P + tr + 64 * p] = v[p];
    }
    __syncthreads();
    real2_t *__restrict__ pen = sm + wave * Y010_P;
    const int ig = x0 + wave, it = ig < g.nx ? ig : 0;  // (pad columns: any table entry, their values are never used)
    const Rot rz{g.az[kz], g.bz[kz], (kz + 1) > g.nz / 2 + 1}, rx{g.ax[it], g.bx[it], (ig + 1) > g.nx / 2 + 1};
    real2_t a[4];
    if (MODE != 1 && MODE != 4) {
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
    if constexpr (MODE == 3) {  // forward sweeps on the tile, both diagonals staged whole
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int R = tr + 64 * h, so = (R / n_) * 8;
            lst[R * 8 + so + tx] = lreg[h];
            lst[Y010_LD + R * 8 + so + tx] = lreg[4 + h];
        }
        __syncthreads();
        if (g.sym) {
            const Y010Chain ch = y010_chain<true>(sm, g, kz, x0, lane, wave);
            if (ch.on) { if (g.db) y010_sweep_fw<true>(ch, lst, tl); else y010_sweep_fw_plain<true>(ch, lst, tl); }
        } else {
            const Y010Chain ch = y010_chain<false>(sm, g, kz, x0, lane, wave);
            if (ch.on) { if (g.db) y010_sweep_fw<false>(ch, lst, tl); else y010_sweep_fw_plain<false>(ch, lst, tl); }
        }
    }
    if constexpr (MODE == 4) {  // backward sweeps: rows n-2 .. n/2+1 staged first, rows n/2 .. 1 behind them
        const int nh = n_ / 2;
        auto stage = [&](bool upper) {
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const int R = tr + 64 * h, s_ = R / n_, j = R % n_ + 1;
                if ((j > nh) == upper) {
                    const int slot = s_ * nh + (upper ? j - nh - 1 : j - 1);
#pragma unroll
                    for (int d = 0; d < 3; d++) lst[d * Y010_LH + slot * 8 + s_ * 8 + tx] = lreg[d * 4 + h];
                }
            }
        };
        stage(true);
        __syncthreads();
        real_t x1 = 0.0, x2 = 0.0;
        auto sweep = [&](auto symt, bool upper) {
            constexpr bool SYM = decltype(symt)::value;
            constexpr int n = SYM ? 128 : 256, RS = SYM ? 4 : 2;
            const Y010Chain ch = y010_chain<SYM>(sm, g, kz, x0, lane, wave);
            if (!ch.on) return;
            if (upper) {
                x1 = ch.pd[(n - 2) * RS]; x2 = ch.pd[(n - 1) * RS];
                if (g.db) y010_sweep_bw<SYM>(ch, lst, n - 2, n / 2 + 1, n / 2, x1, x2);
                else y010_sweep_bw_plain<SYM>(ch, lst, n - 2, n / 2 + 1, n / 2, x1, x2);
            } else {
                if (g.db) y010_sweep_bw<SYM>(ch, lst, n / 2, 1, 0, x1, x2);
                else y010_sweep_bw_plain<SYM>(ch, lst, n / 2, 1, 0, x1, x2);
            }
        };
        if (g.sym) sweep(std::true_type{}, true); else sweep(std::false_type{}, true);
        __syncthreads();
        stage(false);
        __syncthreads();
        if (g.sym) sweep(std::true_type{}, false); else sweep(std::false_type{}, false);
        __syncthreads();
    }
    if (MODE != 0 && MODE != 3) {
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

const real2_t *x3d_fft512_twiddles();
int x3d_fft512_init();

// c[nz][256][nxs], x and z already transformed (mode 0, 2) / still transformed (mode 1, 2).  tables = ax bx ay by az bz
// back to back (global lengths nx nx ny ny nz nz).  *done = false: not served (other ny, odd row pitch)
// nzl (0: nz): the z planes c and lu hold -- nz / 2 + 1 with every x mode for the z-first layout
int x3d_y010_run(x3d_backend *b, real2_t *c, int nxs, int nx, int ny, int nz, int mode, const real_t *tables, int sym,
                 real_t *const lu[2], bool *done, int nzl)
{
    *done = false;
    if (ny != 256 || nxs % 8 != 0 || nx % 2 != 0) return 0;
    if (mode >= 2 && !(lu && lu[0] && (!sym || lu[1]))) return 0;
    if (int rc = x3d_fft512_init()) return rc;
    Y010Arg g;
    g.ax = tables; g.bx = g.ax + nx; g.ay = g.bx + nx; g.by = g.ay + ny; g.az = g.by + ny; g.bz = g.az + nz;
    g.lu0 = lu ? lu[0] : nullptr; g.lu1 = lu ? lu[1] : nullptr;
    g.nxs = nxs; g.nz = nz; g.nx = nx; g.sym = sym;
    g.nzl = nzl > 0 ? nzl : nz;
    static int nodb = -1;
    if (nodb < 0) { const char *e = getenv("X3D_Y010_NO_DB"); nodb = (e && e[0] == '1') ? 1 : 0; }
    g.db = nodb ? 0 : 1;
    const size_t lds = sizeof(real2_t) * (8 * Y010_P + 256) + sizeof(real_t) * (mode == 3 ? 2 * Y010_LD : (mode == 4 ? 3 * Y010_LH : 0));
    const dim3 grid((unsigned)((size_t)g.nzl * (nxs / 8)));
#define GO(M_)                                                                                          \
    do {                                                                                               \
        X3D_LDS_OPTIN(b, (k_y010<M_>));                                                                \
        hipLaunchKernelGGL((k_y010<M_>), grid, dim3(512), lds, b->stream, c, x3d_fft512_twiddles(), g); \
    } while (0)
    if (mode == 0) GO(0); else if (mode == 1) GO(1); else if (mode == 2) GO(2); else if (mode == 3) GO(3); else GO(4);
#undef GO
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}
