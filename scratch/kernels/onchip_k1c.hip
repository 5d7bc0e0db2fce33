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

