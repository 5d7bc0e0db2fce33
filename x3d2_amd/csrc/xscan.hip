// x-direction operators, one pencil per WAVE (kernel family K3s): the x pencil
// is contiguous, so lane l owns rows l*Q+1 .. (l+1)*Q (Q = 4 or 8 -> pencils of
// up to 512 points), loads them as one 32/64-byte vector and keeps everything
// in registers: read the inputs once, write the result once -- 2 field passes
// for tds_solve, 3 for a transport-equation component -- fully coalesced, no
// transposition, no scratch.
//
// The serial recurrences of the reference kernels
//   src/backend/omp/kernels/distributed.f90:34-166 (forward / backward)
//   src/backend/omp/kernels/distributed.f90:186-337 (2x2 systems, substitution, fused_subs)
// run lane-locally from zero and are closed across lanes by a log-step scan:
//   e_end(l) = ehat_end(l) + G_l e_end(l-1)   with G_l = prod over lane l of (-F_j A_j)
// The multipliers of every scan step are data independent and precomputed per
// lane (tds.hip: lane tables).  Same linear system and coefficients as the
// serial sweeps; results differ by re-association only.
// Row tables live in LDS as [entry][lane] (conflict-free ds_read_b64).
#include "common.h"

int npmax_of(const x3d_backend *b);

// lane-table entry indices (per operator): 8*Q row entries then the scan multipliers
#define LT_F(q) (0 * Q + (q))
#define LT_A(q) (1 * Q + (q))
#define LT_PF(q) (2 * Q + (q))
#define LT_H(q) (3 * Q + (q))
#define LT_QB(q) (4 * Q + (q))
#define LT_SA(q) (5 * Q + (q))
#define LT_SC(q) (6 * Q + (q))
#define LT_ST(q) (7 * Q + (q))
#define LT_STC(q) (8 * Q + (q))
#define LT_MF(k) (9 * Q + (k))
#define LT_MB(k) (9 * Q + 6 + (k))
#define LT_N(Q_) (9 * (Q_) + 12)

__device__ __forceinline__ double shfl_up_d(double v, int d, int lane)
{
    const double r = __shfl_up(v, d, 64);
    return lane >= d ? r : 0.0;
}
__device__ __forceinline__ double shfl_down_d(double v, int d, int lane)
{
    const double r = __shfl_down(v, d, 64);
    return lane + d < 64 ? r : 0.0;
}

// extended pencil row jj in [-3, nr+4] (non-decomposed direction: periodic image,
// src/backend/omp/sendrecv.f90:20-22); rows beyond nr+4 read as zero
__device__ __forceinline__ double ext_x(const double *__restrict__ row, int jj, int nr, int n_wrap)
{
    if (jj < 1) return row[n_wrap + jj - 1];
    if (jj > nr) return jj <= nr + 4 ? row[jj - nr - 1] : 0.0;
    return row[jj - 1];
}

// one operator, lane-local + scan: in: w[Q+8] = rows first-4 .. last+4; out: X[Q] back-substituted
// values (before the reduced-system substitution), du1 and xn broadcast to all lanes
template <int Q>
__device__ __forceinline__ void scan_solve(const double (&w)[Q + 8], double (&X)[Q], double &du1, double &xn,
                                           const double *__restrict__ lt, const TdsTab &t, int lane, int first)
{
    const int nr = t.n_rhs, n = t.n_tds;
    const double *__restrict__ cb = t.Cs + 72;
    const double c0 = cb[0], c1 = cb[1], c2 = cb[2], c3 = cb[3], c4 = cb[4], c5 = cb[5], c6 = cb[6], c7 = cb[7],
                 c8 = cb[8];
    double acc[Q];
#pragma unroll
    for (int q = 0; q < Q; q++)
        acc[q] = c0 * w[q] + c1 * w[q + 1] + c2 * w[q + 2] + c3 * w[q + 3] + c4 * w[q + 4] + c5 * w[q + 5] +
                 c6 * w[q + 6] + c7 * w[q + 7] + c8 * w[q + 8];
    // boundary rows (1..4 and n_rhs-3..n_rhs) use their own stencils: only the two end lanes get here
    if (first <= 4 || first + Q - 1 > nr - 4) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = first + q;
            if (j <= nr && (j <= 4 || j > nr - 4)) {
                const double *__restrict__ cs = j <= 4 ? t.Cs + (j - 1) * 9 : t.Cs + 36 + (j - (nr - 4) - 1) * 9;
                acc[q] = cs[0] * w[q] + cs[1] * w[q + 1] + cs[2] * w[q + 2] + cs[3] * w[q + 3] + cs[4] * w[q + 4] +
                         cs[5] * w[q + 5] + cs[6] * w[q + 6] + cs[7] * w[q + 7] + cs[8] * w[q + 8];
            }
        }
    }
    // ---- lane-local forward elimination from zero
    double prev = 0.0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        X[q] = lt[LT_F(q) * 64 + lane] * (acc[q] - lt[LT_A(q) * 64 + lane] * prev);
        prev = X[q];
    }
    // ---- scan of the lane-end values, then carry-in = true e at the end of lane l-1
    double v = prev;
#pragma unroll
    for (int k = 0; k < 6; k++) v += lt[LT_MF(k) * 64 + lane] * shfl_up_d(v, 1 << k, lane);
    double carry = shfl_up_d(v, 1, lane);
    // ---- apply, lane-local back-substitution from zero
    double nxt = 0.0;
#pragma unroll
    for (int q = Q - 1; q >= 0; q--) {
        X[q] = (X[q] + lt[LT_PF(q) * 64 + lane] * carry) + lt[LT_H(q) * 64 + lane] * nxt;
        nxt = X[q];
    }
    if (n == nr) {}  // (row n_rhs = n+1 of a v2p operator carries F = H = 0 in the tables)
    v = nxt;
#pragma unroll
    for (int k = 0; k < 6; k++) v += lt[LT_MB(k) * 64 + lane] * shfl_down_d(v, 1 << k, lane);
    carry = shfl_down_d(v, 1, lane);
#pragma unroll
    for (int q = 0; q < Q; q++) X[q] = X[q] + lt[LT_QB(q) * 64 + lane] * carry;
    // du_1 = last_r * X_1 (X_1 = e_1 - bw_1 X_2, distributed.f90:161-166); X_n = e_n
    du1 = t.last_r * __shfl(X[0], 0, 64);
    const int ln = (n - 1) / Q, qn = (n - 1) % Q;
    double xsel = 0.0;
#pragma unroll
    for (int q = 0; q < Q; q++) xsel = (q == qn) ? X[q] : xsel;
    xn = __shfl(xsel, ln, 64);
}

template <int Q>
__device__ __forceinline__ void load_window(double (&w)[Q + 8], const double *__restrict__ row, int first, int nr,
                                            int n_wrap, bool interior)
{
    if (interior) {  // rows first-4 .. first+Q+3 all inside [1, nr]: 16-byte aligned vector loads
        const double2 *__restrict__ v2 = reinterpret_cast<const double2 *>(row + first - 5);
#pragma unroll
        for (int m = 0; m < (Q + 8) / 2; m++) {
            const double2 t2 = v2[m];
            w[2 * m] = t2.x;
            w[2 * m + 1] = t2.y;
        }
    } else {
#pragma unroll
        for (int m = 0; m < Q + 8; m++) w[m] = ext_x(row, first - 4 + m, nr, n_wrap);
    }
}

// ---------------------------------------------------------------- tds_solve
template <int Q, bool ACC>
__global__ void __launch_bounds__(256) k_xscan_tds(double *__restrict__ du, const double *__restrict__ u, TdsTab t,
                                                   int np, long pitch, int n_wrap, double scale)
{
    extern __shared__ double lt[];  // [LT_N(Q)][64]
    for (int i = threadIdx.x; i < LT_N(Q) * 64; i += blockDim.x) lt[i] = t.TL[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int n = t.n_tds, nr = t.n_rhs;
    const int first = lane * Q + 1;
    const bool interior = first - 4 >= 1 && first + Q + 3 <= nr;
    for (int p = blockIdx.x * (blockDim.x >> 6) + wave; p < np; p += nwaves) {
        const double *__restrict__ row = u + (long)p * pitch;
        double w[Q + 8], X[Q], du1, xn;
        load_window<Q>(w, row, first, nr, n_wrap, interior);
        scan_solve<Q>(w, X, du1, xn, lt, t, lane, first);
        const double du_s = t.rs_s * (du1 - t.sa1 * xn);  // periodic self-exchange: recv_s = X_n
        const double du_e = t.rs_e * (xn - t.scn * du1);  //                          recv_e = du_1
        double *__restrict__ orow = du + (long)p * pitch;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = first + q;
            if (j <= n) {
                double r = (X[q] - lt[LT_SA(q) * 64 + lane] * du_s - lt[LT_SC(q) * 64 + lane] * du_e) *
                           lt[LT_ST(q) * 64 + lane];
                r = (j == 1) ? du_s * lt[LT_ST(q) * 64 + lane] : r;
                r = (j == n) ? du_e * lt[LT_ST(q) * 64 + lane] : r;
                orow[j - 1] = ACC ? orow[j - 1] + scale * r : r;
            }
        }
    }
}

// ---------------------------------------------------------------- transeq component
template <int Q, bool SAME, bool ACC>
__global__ void __launch_bounds__(512)
    k_xscan_transeq(double *__restrict__ rhs, const double *__restrict__ u, const double *__restrict__ cv, TdsTab t1,
                    TdsTab t2, TdsTab t3, int np, long pitch, double nu)
{
    extern __shared__ double lt[];  // three operators: [3][LT_N(Q)][64]
    constexpr int LN = LT_N(Q) * 64;
    for (int i = threadIdx.x; i < LN; i += blockDim.x) {
        lt[i] = t1.TL[i];
        lt[LN + i] = t2.TL[i];
        lt[2 * LN + i] = t3.TL[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int n = t1.n_tds;
    const int first = lane * Q + 1;
    const bool interior = first - 4 >= 1 && first + Q + 3 <= n;
    const double *__restrict__ l1 = lt, *__restrict__ l2 = lt + LN, *__restrict__ l3 = lt + 2 * LN;
    for (int p = blockIdx.x * (blockDim.x >> 6) + wave; p < np; p += nwaves) {
        const double *__restrict__ ru = u + (long)p * pitch;
        const double *__restrict__ rc = cv + (long)p * pitch;
        double wu[Q + 8], wp[Q + 8], vq[Q];
        load_window<Q>(wu, ru, first, n, n, interior);
        if (SAME) {
#pragma unroll
            for (int m = 0; m < Q + 8; m++) wp[m] = wu[m] * wu[m];
#pragma unroll
            for (int q = 0; q < Q; q++) vq[q] = wu[q + 4];
        } else {
            load_window<Q>(wp, rc, first, n, n, interior);
#pragma unroll
            for (int q = 0; q < Q; q++) vq[q] = wp[q + 4];
#pragma unroll
            for (int m = 0; m < Q + 8; m++) wp[m] = wu[m] * wp[m];  // ud = u*conv incl. halo products
        }
        double X1[Q], X2[Q], X3[Q], a1, b1, a2, b2, a3, b3;
        scan_solve<Q>(wu, X1, a1, b1, l1, t1, lane, first);
        scan_solve<Q>(wp, X2, a2, b2, l2, t2, lane, first);
        scan_solve<Q>(wu, X3, a3, b3, l3, t3, lane, first);
        const double du_s = t1.rs_s * (a1 - t1.sa1 * b1), du_e = t1.rs_e * (b1 - t1.scn * a1);
        const double dud_s = t2.rs_s * (a2 - t2.sa1 * b2), dud_e = t2.rs_e * (b2 - t2.scn * a2);
        const double d2u_s = t3.rs_s * (a3 - t3.sa1 * b3), d2u_e = t3.rs_e * (b3 - t3.scn * a3);
        double *__restrict__ orow = rhs + (long)p * pitch;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = first + q;
            if (j <= n) {
                const double st1 = l1[LT_ST(q) * 64 + lane], st2 = l2[LT_ST(q) * 64 + lane],
                             st3 = l3[LT_ST(q) * 64 + lane], stc = l3[LT_STC(q) * 64 + lane];
                const double v = vq[q];
                const double temp_du = st1 * (X1[q] - l1[LT_SA(q) * 64 + lane] * du_s - l1[LT_SC(q) * 64 + lane] * du_e);
                const double temp_dud = st2 * (X2[q] - l2[LT_SA(q) * 64 + lane] * dud_s - l2[LT_SC(q) * 64 + lane] * dud_e);
                const double temp_d2u =
                    st3 * (X3[q] - l3[LT_SA(q) * 64 + lane] * d2u_s - l3[LT_SC(q) * 64 + lane] * d2u_e) + temp_du * stc;
                double r = -0.5 * (v * temp_du + temp_dud) + nu * temp_d2u;  // distributed.f90:315-324
                if (j == 1)
                    r = -0.5 * (v * du_s * st1 + dud_s * st2) + nu * (d2u_s * st3 + du_s * st1 * stc);  // :304-311
                if (j == n)
                    r = -0.5 * (v * du_e * st1 + dud_e * st2) + nu * (d2u_e * st3 + du_e * st1 * stc);  // :328-335
                orow[j - 1] = ACC ? orow[j - 1] + r : r;
            }
        }
    }
}

// ---------------------------------------------------------------- launchers
static bool xscan_ok(const x3d_tdsops *t) { return t->tab.TL != nullptr && (t->tab.Q == 4 || t->tab.Q == 8); }

int x3d_xscan_tds(x3d_backend *b, double *du, const double *u, const x3d_tdsops *t, int acc, double scale, bool *done)
{
    *done = false;
    if (!xscan_ok(t)) return 0;
    const int Q = t->tab.Q, np = b->ny * b->nz;
    const size_t lds = sizeof(double) * LT_N(Q) * 64;
    int blocks = (np + 3) / 4;
    blocks = blocks > 2048 ? 2048 : blocks;
    ProfScope ps(b, X3D_K_TDS_FWD, X3D_DIR_X);
#define LAUNCH(Q_, A_, SC_)                                                                                    \
    hipLaunchKernelGGL((k_xscan_tds<Q_, A_>), dim3(blocks), dim3(256), lds, b->stream, du, u, t->tab, np,     \
                       (long)b->nxp, t->n_tds, SC_)
    if (Q == 8) { if (acc) LAUNCH(8, true, scale); else LAUNCH(8, false, 1.0); }
    else { if (acc) LAUNCH(4, true, scale); else LAUNCH(4, false, 1.0); }
#undef LAUNCH
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}

template <int Q, bool SAME, bool ACC>
static int launch_transeq(x3d_backend *b, double *rhs, const double *u, const double *conv, double nu,
                          const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int np, int blocks,
                          size_t lds)
{
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KB of dynamic LDS needs the opt-in
        X3D_HIP(hipFuncSetAttribute((const void *)k_xscan_transeq<Q, SAME, ACC>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_xscan_transeq<Q, SAME, ACC>), dim3(blocks), dim3(512), lds, b->stream, rhs, u, conv, t1->tab,
                       t2->tab, t3->tab, np, (long)b->nxp, nu);
    X3D_HIP(hipGetLastError());
    return 0;
}

int x3d_xscan_transeq(x3d_backend *b, double *rhs, const double *u, const double *conv, double nu,
                      const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, bool *done)
{
    *done = false;
    if (!xscan_ok(t1) || !xscan_ok(t2) || !xscan_ok(t3) || t1->tab.Q != t2->tab.Q || t1->tab.Q != t3->tab.Q) return 0;
    const int Q = t1->tab.Q, np = b->ny * b->nz;
    const size_t lds = sizeof(double) * 3 * LT_N(Q) * 64;
    if (lds > 160 * 1024) return 0;
    int blocks = (np + 7) / 8;
    blocks = blocks > 256 ? 256 : blocks;  // one 8-wave workgroup per CU (129 KB of lane tables in LDS)
    const bool same = u == conv;
    ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X);
    int rc;
#define GO(Q_)                                                                                                 \
    (same ? (acc ? launch_transeq<Q_, true, true>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds)           \
                 : launch_transeq<Q_, true, false>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds))         \
          : (acc ? launch_transeq<Q_, false, true>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds)          \
                 : launch_transeq<Q_, false, false>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds)))
    rc = Q == 8 ? GO(8) : GO(4);
#undef GO
    if (rc) return rc;
    *done = true;
    return 0;
}
