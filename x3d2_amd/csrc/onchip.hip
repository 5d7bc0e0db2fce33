// Single-pass tds_solve for a non-decomposed direction (kernel family K1c):
// the whole pencil stays on chip.  A workgroup owns 64 pencils (lanes across x,
// 512-B rows); wave c owns rows [c*M+1, (c+1)*M] of them (M = 32 for pencils up
// to 256 rows, 64 up to 512) and keeps its forward-eliminated values in registers.  HBM traffic:
// read u once, write du once (plus 8 halo rows per wave, served by L2).
//
// The serial recurrences of the reference kernels
//   src/backend/omp/kernels/distributed.f90:34-166 (forward / backward)
//   src/backend/omp/kernels/distributed.f90:186-228 (2x2 systems, substitution)
// are evaluated chunk-parallel with an exact carry correction: with e the
// forward-eliminated values and X the back-substituted ones,
//   e_j = ehat_j + PF_j * e_{s-1},   X_j = Xhat_j + QB_j * X_{t+1}      (chunk [s, t])
// where ehat / Xhat are the chunk-local sweeps started from zero and
// PF_j = prod_{l=s..j} (-F_l A_l), QB_j = prod_{l=j..t} (-bw_l) are tables built
// with the operator (tds.hip).  The chunk ends are chained through LDS.  Same
// linear system, same coefficients; results differ from the serial order by
// re-association only (~1e-16 relative).
#include "common.h"

#define MAXC 16  // waves per workgroup (1024 threads -> 128 VGPRs per lane; the 32-row chunk needs ~95)

__device__ __forceinline__ double ext_row_c(const double *__restrict__ u, long base, long rs, int jj, int nr,
                                            int n_wrap)
{
    // periodic image of a non-decomposed direction (src/backend/omp/sendrecv.f90:20-22)
    if (jj < 1) return u[base + (long)(n_wrap + jj - 1) * rs];
    if (jj > nr) return jj <= nr + 4 ? u[base + (long)(jj - nr - 1) * rs] : 0.0;
    return u[base + (long)(jj - 1) * rs];
}

template <int M, bool ACC>
__global__ void __launch_bounds__(64 * MAXC)
    k_tds_onchip(double *__restrict__ du, const double *__restrict__ u, TdsTab t, PencilGeom g, int n_wrap,
                 double scale)
{
    __shared__ double ends[MAXC][64], starts[MAXC][64], misc[2][64];
    // the wave index is uniform: tell the compiler, so that row indices, table loads and the
    // boundary-row branches stay scalar
    const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), C = blockDim.x >> 6;
    int p = blockIdx.x * 64 + lane;
    const bool active = p < g.np;
    p = active ? p : g.np - 1;  // inactive lanes shadow a valid pencil and never store
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int n = t.n_tds, nr = t.n_rhs;
    const int s = c * M + 1;

    // ---- P1: load the chunk (one burst), chunk-local forward elimination in place
    double x[M], w[9];
#pragma unroll
    for (int q = 0; q < M; q++) x[q] = ext_row_c(u, base, rs, s + q, nr, n_wrap);
#pragma unroll
    for (int m = 0; m < 4; m++) w[m] = ext_row_c(u, base, rs, s - 4 + m, nr, n_wrap);
    double hr[4];
#pragma unroll
    for (int m = 0; m < 4; m++) hr[m] = ext_row_c(u, base, rs, s + M + m, nr, n_wrap);
#pragma unroll
    for (int m = 0; m < 5; m++) w[4 + m] = x[m];
    double cb[9];
#pragma unroll
    for (int m = 0; m < 9; m++) cb[m] = t.Cs[72 + m];
    double prev = 0.0;
#pragma unroll
    for (int q = 0; q < M; q++) {
        const int j = s + q;
        double e = 0.0;
        if (j <= nr) {
            const double *cs = (j <= 4) ? t.Cs + (j - 1) * 9 : t.Cs + 36 + (j - (nr - 4) - 1) * 9;
            const bool bulk = j > 4 && j <= nr - 4;
            double acc;
            if (bulk)
                acc = cb[0] * w[0] + cb[1] * w[1] + cb[2] * w[2] + cb[3] * w[3] + cb[4] * w[4] + cb[5] * w[5] +
                      cb[6] * w[6] + cb[7] * w[7] + cb[8] * w[8];
            else
                acc = cs[0] * w[0] + cs[1] * w[1] + cs[2] * w[2] + cs[3] * w[3] + cs[4] * w[4] + cs[5] * w[5] +
                      cs[6] * w[6] + cs[7] * w[7] + cs[8] * w[8];
            e = T_F(t, j) * (acc - T_A(t, j) * prev);
            prev = e;
        }
        const double feed = (q + 5 < M) ? x[(q + 5) % M] : hr[(q + 5 - M) & 3];
        x[q] = e;
#pragma unroll
        for (int m = 0; m < 8; m++) w[m] = w[m + 1];
        w[8] = feed;
        if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // bound the live range of the row tables
    }
    ends[c][lane] = prev;
    __syncthreads();

    // ---- P2: forward carry, then chunk-local back-substitution
    {
        double carry = 0.0;
        for (int cc = 0; cc < c; cc++) {
            const int tt = (cc + 1) * M < nr ? (cc + 1) * M : nr;
            carry = ends[cc][lane] + T_PF(t, tt) * carry;
        }
        double nxt = 0.0;
#pragma unroll
        for (int q = M - 1; q >= 0; q--) {
            const int j = s + q;
            if (j <= n) {
                const double e = x[q] + T_PF(t, j) * carry;
                const double hj = (j >= 2 && j <= n - 2) ? -T_BW(t, j) : 0.0;  // rows 1, n-1, n: no update
                x[q] = e + hj * nxt;
                nxt = x[q];
            }
            if ((q & 7) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        starts[c][lane] = x[0];
    }
    __syncthreads();

    // ---- P3: backward carry; publish du_1 (needs X_1, X_2) and X_n
    {
        double carry = 0.0;
        for (int cc = C - 1; cc > c; cc--) {
            const int ss = cc * M + 1;
            if (ss <= n) carry = starts[cc][lane] + T_QB(t, ss) * carry;
        }
#pragma unroll
        for (int q = 0; q < M; q++) {
            const int j = s + q;
            if (j <= n) x[q] = x[q] + T_QB(t, j) * carry;
            if (j == n) misc[1][lane] = x[q];
            if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        if (c == 0) misc[0][lane] = t.last_r * (x[0] - t.bw1 * x[1]);  // distributed.f90:161-166
    }
    __syncthreads();

    // ---- P4: reduced 2x2 systems with the periodic self-exchange, substitution, store
    const double du1 = misc[0][lane], xn = misc[1][lane];
    const double du_s = t.rs_s * (du1 - t.sa1 * xn);  // recv_s = own send_e = X_n
    const double du_e = t.rs_e * (xn - t.scn * du1);  // recv_e = own send_s = du_1
#pragma unroll
    for (int q0 = 0; q0 < M; q0 += 8) {
        double old[8];
        if (ACC) {
#pragma unroll
            for (int k = 0; k < 8; k++) old[k] = (s + q0 + k <= n) ? du[base + (long)(s + q0 + k - 1) * rs] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int q = q0 + k, j = s + q;
            if (j <= n) {
                double r = (x[q] - T_SA(t, j) * du_s - T_SC(t, j) * du_e) * T_ST(t, j);  // :215-222
                r = (j == 1) ? du_s * T_ST(t, j) : r;                               // :209-213
                r = (j == n) ? du_e * T_ST(t, j) : r;                               // :224-228
                if (active) du[base + (long)(j - 1) * rs] = ACC ? old[k] + scale * r : r;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

int x3d_onchip_tds(x3d_backend *b, double *du, const double *u, const x3d_tdsops *t, int dir, int acc, double scale)
{
    PencilGeom g = x3d_geom(b, dir);
    const int M = t->tab.chunk;
    const int C = (t->n_rhs + M - 1) / M;
    X3D_REQUIRE(C >= 1 && C <= MAXC, "on-chip tds_solve: pencil of %d rows does not fit %d waves", t->n_rhs, MAXC);
    ProfScope ps(b, X3D_K_TDS_FWD, dir);
    dim3 grid((g.np + 63) / 64), block(64 * C);
#define LAUNCH(M_, A_, SC_)                                                                                    \
    hipLaunchKernelGGL((k_tds_onchip<M_, A_>), grid, block, 0, b->stream, du, u, t->tab, g, t->n_tds, SC_)
    if (M == 32) { if (acc) LAUNCH(32, true, scale); else LAUNCH(32, false, 1.0); }
    else { if (acc) LAUNCH(64, true, scale); else LAUNCH(64, false, 1.0); }
#undef LAUNCH
    X3D_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K1e: the same chunk-parallel single-pass solve for 512-row pencils with TWO workgroups per CU.
// K1c needs 16 waves x 128 VGPRs = a whole CU for 64 pencils, so its load, compute and store phases
// cannot overlap with anything (0.77 ms vs 0.85 ms for the two-sweep pair; at 256 rows, where two
// 8-wave workgroups fit, it is 36 % faster).  Here a workgroup is 8 waves and owns 32 pencils: the
// two halves of a wave are two different 32-row chunks of the same 32 pencils (lane = x + 32 h,
// chunk = 2 wave + h), every half-wave row access is 256 contiguous bytes.  The row index now differs
// between the halves, so the row tables cannot come from scalar loads: they are staged in LDS once per
// workgroup ([table][row], 2 distinct addresses per wave access = broadcast reads).
// ---------------------------------------------------------------------------------------------
#define K1E_M 32
#define K1E_TAB 8  // F A PF HB QB SA SC ST

struct Coef9 { double c[9]; };

// (wave-uniform pointer) + (32-bit per-lane BYTE offset): the form the backend turns into
// global_load/store v, v_off, s[base:base+1].  With an element offset it cannot prove that 8*off fits 32
// bits and builds a 64-bit address in two VGPRs per access (77 v_lshl_add_u64 in this kernel).
// (plain accesses: nontemporal ones measured 4 % slower here, A/B on the same box)
__device__ __forceinline__ double ldg(const double *base, unsigned boff)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + boff);
}
__device__ __forceinline__ void stg(double *base, unsigned boff, double v)
{
    *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + boff) = v;
}

// M = 32: 512-row pencils, M = 16: 256-row pencils (16 chunks either way)
template <bool ACC, int M, bool NARROW>
__global__ void __launch_bounds__(512, 4)  // 4 waves per SIMD = two workgroups per CU, <= 128 VGPRs
    k_tds_onchip2(double *__restrict__ du, const double *__restrict__ u, TdsTab t, PencilGeom g, double scale,
                  Coef9 cf)
{
    // periodic-type operator on pencils of exactly 16 * M rows (n_tds = n_rhs, bulk stencil everywhere):
    // no row guards, no boundary stencils, wrap-around halos by index arithmetic
    extern __shared__ double lds[];  // K1E_TAB tables of LR rows, then ends[16][32], starts[16][32], misc[2][32]
    constexpr int n = 16 * M, LR = n + 8;
    double *tF = lds, *tA = tF + LR, *tPF = tA + LR, *tHB = tPF + LR, *tQB = tHB + LR, *tSA = tQB + LR,
           *tSC = tSA + LR, *tST = tSC + LR;
    double *ends = tST + LR, *starts = ends + 16 * 32, *misc = starts + 16 * 32;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xl = lane & 31, c = 2 * wv + (lane >> 5);
    const int p = blockIdx.x * 32 + xl;  // np is a multiple of 32 (launcher)
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int s = c * M + 1;
    // addresses as (wave-uniform row pointer) + (one 32-bit per-lane element offset): with 40 per-lane
    // 64-bit addresses in flight the kernel spills (a block has < 2^31 elements)
    const unsigned off = (unsigned)(base + (long)(s - 1) * rs) * 8u;  // bytes; a block is < 4 GiB

    // ---- P1: load the chunk + 4 + 4 halo rows (periodic image), chunk-local forward elimination in place.
    // x[q] is overwritten by the eliminated value, so the 4 original rows behind the current one are kept
    // in p0..p3; the rows ahead are still original in x[] (the last 4 come from the right halo, loaded late)
    double x[M], hr[4];
#pragma unroll
    for (int q = 0; q < M; q++) x[q] = ldg(u + (long)q * rs, off);
    double p0 = ldg(u, (unsigned)(base + (long)((s - 5 + n) & (n - 1)) * rs) * 8u),
           p1 = ldg(u, (unsigned)(base + (long)((s - 4 + n) & (n - 1)) * rs) * 8u),
           p2 = ldg(u, (unsigned)(base + (long)((s - 3 + n) & (n - 1)) * rs) * 8u),
           p3 = ldg(u, (unsigned)(base + (long)((s - 2 + n) & (n - 1)) * rs) * 8u);
    // the row tables are staged while the loads above are in flight
    for (int j = threadIdx.x; j < LR; j += blockDim.x) {
        const bool in = j >= 1 && j <= n;
        tF[j] = in ? T_F(t, j) : 0.0;
        tA[j] = in ? T_A(t, j) : 0.0;
        tPF[j] = in ? (M == 32 ? T_PF(t, j) : T_PF16(t, j)) : 0.0;  // chunk-local products for M-row chunks
        tHB[j] = (in && j >= 2 && j <= n - 2) ? -T_BW(t, j) : 0.0;  // rows 1, n-1, n: no backward update
        tQB[j] = in ? (M == 32 ? T_QB(t, j) : T_QB16(t, j)) : 0.0;
        tSA[j] = in ? T_SA(t, j) : 0.0;
        tSC[j] = in ? T_SC(t, j) : 0.0;
        tST[j] = in ? T_ST(t, j) : 0.0;
    }
    __syncthreads();
    const double c0 = cf.c[0], c1 = cf.c[1], c2 = cf.c[2], c3 = cf.c[3], c4 = cf.c[4], c5 = cf.c[5], c6 = cf.c[6],
                 c7 = cf.c[7], c8 = cf.c[8];  // kernel arguments: SGPRs
    double prev = 0.0;
#pragma unroll
    for (int q = 0; q < M; q++) {
        const int j = s + q;
        if (q == M - 12) {
#pragma unroll
            for (int m = 0; m < 4; m++) hr[m] = ldg(u, (unsigned)(base + (long)((s + M - 1 + m) & (n - 1)) * rs) * 8u);
        }
#define AHEAD(d) ((q + (d) < M) ? x[(q + (d)) % M] : hr[(q + (d) - M) & 3])
        const double cur = x[q];
        // NARROW: compact6 / classic stencils reach 2 rows only: skip the zero taps (adding 0 * x is exact)
        const double acc = NARROW ? c2 * p2 + c3 * p3 + c4 * cur + c5 * AHEAD(1) + c6 * AHEAD(2)
                                  : c0 * p0 + c1 * p1 + c2 * p2 + c3 * p3 + c4 * cur + c5 * AHEAD(1) + c6 * AHEAD(2) +
                                        c7 * AHEAD(3) + c8 * AHEAD(4);
#undef AHEAD
        const double e = tF[j] * (acc - tA[j] * prev);
        prev = e;
        x[q] = e;
        p0 = p1; p1 = p2; p2 = p3; p3 = cur;
        if (q & 1) __builtin_amdgcn_sched_barrier(0);  // bound the live range of the LDS table reads
    }
    ends[c * 32 + xl] = prev;
    __syncthreads();

    // ---- P2: forward carry (serial over the chunks before this one), chunk-local back-substitution
    {
        double carry = 0.0;
        for (int cc = 0; cc < c; cc++) carry = ends[cc * 32 + xl] + tPF[(cc + 1) * M] * carry;
        double nxt = 0.0;
#pragma unroll
        for (int q = M - 1; q >= 0; q--) {
            const int j = s + q;
            const double e = x[q] + tPF[j] * carry;
            x[q] = e + tHB[j] * nxt;
            nxt = x[q];
            if ((q & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        starts[c * 32 + xl] = x[0];
    }
    __syncthreads();

    // ---- P3: backward carry (applied on the fly in P4); publish du_1 and X_n
    double carry = 0.0;
    for (int cc = 15; cc > c; cc--) carry = starts[cc * 32 + xl] + tQB[cc * M + 1] * carry;
    if (c == 15) misc[32 + xl] = x[M - 1];  // carry = 0 there
    if (c == 0) misc[xl] = t.last_r * ((x[0] + tQB[1] * carry) - t.bw1 * (x[1] + tQB[2] * carry));  // :161-166
    __syncthreads();

    // ---- P4: reduced 2x2 systems with the periodic self-exchange, substitution, store
    const double du1 = misc[xl], xn = misc[32 + xl];
    const double du_s = t.rs_s * (du1 - t.sa1 * xn);
    const double du_e = t.rs_e * (xn - t.scn * du1);
#pragma unroll
    for (int q0 = 0; q0 < M; q0 += 2) {
        double old[2];
        if (ACC) {
#pragma unroll
            for (int k = 0; k < 2; k++) old[k] = ldg(du + (long)(q0 + k) * rs, off);
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int q = q0 + k, j = s + q;
            const double X = x[q] + tQB[j] * carry;
            double r = (X - tSA[j] * du_s - tSC[j] * du_e) * tST[j];  // :215-222
            if (q == 0) r = (c == 0) ? du_s * tST[j] : r;        // row 1, :209-213
            if (q == M - 1) r = (c == 15) ? du_e * tST[j] : r;   // row n, :224-228
            stg(du + (long)q * rs, off, ACC ? old[k] + scale * r : r);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int M>
static int launch_onchip2(x3d_backend *b, double *du, const double *u, const x3d_tdsops *t, const PencilGeom &g,
                          int dir, int acc, double scale)
{
    const size_t lds = sizeof(double) * ((size_t)K1E_TAB * (16 * M + 8) + 2 * 16 * 32 + 64);
    static bool attr = false;
    if (!attr) {
        const void *ks[4] = {(const void *)k_tds_onchip2<false, M, false>, (const void *)k_tds_onchip2<true, M, false>,
                             (const void *)k_tds_onchip2<false, M, true>, (const void *)k_tds_onchip2<true, M, true>};
        for (const void *k : ks) X3D_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    ProfScope ps(b, X3D_K_TDS_FWD, dir);
    dim3 grid(g.np / 32), block(512);
    Coef9 cf;
    for (int m = 0; m < 9; m++) cf.c[m] = t->coeffs[m];
    const bool narrow = cf.c[0] == 0.0 && cf.c[1] == 0.0 && cf.c[7] == 0.0 && cf.c[8] == 0.0;
#define GO(A_, N_, S_) hipLaunchKernelGGL((k_tds_onchip2<A_, M, N_>), grid, block, lds, b->stream, du, u, t->tab, g, S_, cf)
    if (acc) { if (narrow) GO(true, true, scale); else GO(true, false, scale); }
    else { if (narrow) GO(false, true, 1.0); else GO(false, false, 1.0); }
#undef GO
    X3D_HIP(hipGetLastError());
    return 0;
}

int x3d_onchip2_tds(x3d_backend *b, double *du, const double *u, const x3d_tdsops *t, int dir, int acc, double scale,
                    bool *done)
{
    *done = false;
    PencilGeom g = x3d_geom(b, dir);
    const int n = t->n_tds;
    if (!(t->tab.bulk_only && (n == 512 || n == 256) && t->n_rhs == n && g.dim0 % 32 == 0 && g.np % 32 == 0)) return 0;
    const int rc = n == 512 ? launch_onchip2<32>(b, du, u, t, g, dir, acc, scale)
                            : launch_onchip2<16>(b, du, u, t, g, dir, acc, scale);
    if (rc) return rc;
    *done = true;
    return 0;
}
