// Single-launch DistD2 solve for a direction that is NOT decomposed across
// ranks (kernel family K2): forward sweep, reduced 2x2 systems, backward sweep
// and substitution of one pencil all happen in the lane that owns it.
//
// The forward-eliminated values d_j are NOT streamed through HBM (the
// two-sweep kernels of tds.hip write and re-read 3 arrays for a transport
// equation component).  Instead the forward sweep keeps one checkpoint of d
// every CK rows; the backward sweep walks the pencil block by block in reverse,
// re-reads the CK+8 input rows of the block (recently touched -> L2 /
// Infinity-Cache hits), recomputes the block's d_j in registers from the
// checkpoint and back-substitutes through it.  Arithmetic and operation order
// are those of the reference kernels:
//   src/backend/omp/kernels/distributed.f90:11-168  (der_univ_dist)
//   src/backend/omp/kernels/distributed.f90:170-337 (der_univ_subs / _fused_subs)
// HBM traffic per transport-equation component: read u, conv (+ partial
// re-read), write rhs: 3-5 passes instead of 10.
#include "common.h"

#define CK 16

int npmax_of(const x3d_backend *b);

template <bool HB_UNUSED = false>
__device__ __forceinline__ double ext_row_l(const double *__restrict__ u, long base, long rs, int jj, int nr,
                                            int n_wrap)
{
    // periodic image of a non-decomposed direction (sendrecv_fields nproc==1,
    // src/backend/omp/sendrecv.f90:20-22): u_s(r) = u(n_wrap-4+r), u_e(r) = u(r)
    if (jj < 1) return u[base + (long)(n_wrap + jj - 1) * rs];
    if (jj > nr) return u[base + (long)(jj - nr - 1) * rs];
    return u[base + (long)(jj - 1) * rs];
}

__device__ __forceinline__ double dot9f(const double *__restrict__ c, const double (&w)[9])
{
    return c[0] * w[0] + c[1] * w[1] + c[2] * w[2] + c[3] * w[3] + c[4] * w[4] + c[5] * w[5] + c[6] * w[6] +
           c[7] * w[7] + c[8] * w[8];
}

__device__ __forceinline__ const double *stencil_row_f(const double *__restrict__ Cs, int j, int nr)
{
    if (j <= 4) return Cs + (j - 1) * 9;
    if (j > nr - 4) return Cs + 36 + (j - (nr - 4) - 1) * 9;
    return Cs + 72;
}

struct Tabs3 { TdsTab t[3]; };

// NOPS = 1: tds_solve (out = [out + scale *] T(u))
// NOPS = 3: transport-equation component with operators t[0]=du, t[1]=dud, t[2]=d2u
template <int NOPS, bool SAME, bool ACC>
__global__ void __launch_bounds__(64)
    k_fused(double *__restrict__ out, const double *__restrict__ u, const double *__restrict__ cv,
            double *__restrict__ ckpt, Tabs3 T, PencilGeom g, int n_wrap, double nu, double scale)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int n = T.t[0].n_tds, nr = T.t[0].n_rhs;
    const long np = g.np;

    double wu[9], wp[9];
    double prev[NOPS], S[NOPS], first[NOPS], last[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; o++) { prev[o] = 0.0; S[o] = 0.0; first[o] = 0.0; last[o] = 0.0; }

    auto load_window = [&](int jc) {  // window centred on row jc
#pragma unroll
        for (int m = 0; m < 9; m++) {
            wu[m] = ext_row_l(u, base, rs, jc - 4 + m, nr, n_wrap);
            if (NOPS == 3) wp[m] = wu[m] * (SAME ? wu[m] : ext_row_l(cv, base, rs, jc - 4 + m, nr, n_wrap));
        }
    };
    auto shift_in = [&](int jj) {  // advance the window by one row, feeding row jj
        double a = 0.0, b2 = 0.0;
        if (jj <= nr + 4) {
            a = ext_row_l(u, base, rs, jj, nr, n_wrap);
            if (NOPS == 3) b2 = a * (SAME ? a : ext_row_l(cv, base, rs, jj, nr, n_wrap));
        }
#pragma unroll
        for (int m = 0; m < 8; m++) { wu[m] = wu[m + 1]; if (NOPS == 3) wp[m] = wp[m + 1]; }
        wu[8] = a;
        if (NOPS == 3) wp[8] = b2;
    };
    auto eliminate = [&](int j, double (&e)[NOPS]) {  // forward-elimination step of row j
#pragma unroll
        for (int o = 0; o < NOPS; o++) {
            const TdsTab &t = T.t[o];
            const double acc = dot9f(stencil_row_f(t.Cs, j, nr), (NOPS == 3 && o == 1) ? wp : wu);
            e[o] = t.F[j] * (acc - t.A[j] * prev[o]);
            prev[o] = e[o];
        }
    };

    // ---------------- forward sweep: checkpoints, S, first and last rows
    load_window(1);
    for (int j = 1; j <= nr; j++) {
        double e[NOPS];
        eliminate(j, e);
        if (j <= n) {
#pragma unroll
            for (int o = 0; o < NOPS; o++) {
                S[o] += T.t[o].W[j] * e[o];
                if (j == 1) first[o] = e[o];
                if (j == n) last[o] = e[o];
            }
            if (j % CK == 0) {
#pragma unroll
                for (int o = 0; o < NOPS; o++) ckpt[((long)(j / CK) * NOPS + o) * np + p] = e[o];
            }
        }
        shift_in(j + 5);
    }

    // ---------------- reduced 2x2 systems with the periodic self-exchange
    // (recv_s = own d_n, recv_e = own du_1), distributed.f90:186-206
    double xs[NOPS], xe[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; o++) {
        const TdsTab &t = T.t[o];
        const double du1 = t.last_r * (first[o] - t.bw1 * S[o]);
        xs[o] = t.rs_s * (du1 - t.sa1 * last[o]);
        xe[o] = t.rs_e * (last[o] - t.scn * du1);
    }

    auto emit = [&](int j, const double (&c)[NOPS], bool is_first, bool is_last) {
        const long o_ = base + (long)(j - 1) * rs;
        double r;
        if (NOPS == 1) {
            const TdsTab &t = T.t[0];
            if (is_last) r = xe[0] * t.St[j];
            else if (is_first) r = xs[0] * t.St[j];
            else r = (c[0] - t.Sa[j] * xs[0] - t.Sc[j] * xe[0]) * t.St[j];
            r = ACC ? out[o_] + scale * r : r;
        } else {
            const TdsTab &a = T.t[0], &b2 = T.t[NOPS > 1 ? 1 : 0], &d2 = T.t[NOPS > 2 ? 2 : 0];
            const double v = SAME ? u[o_] : cv[o_];
            if (is_last || is_first) {  // distributed.f90:304-311, 328-335
                const double s1 = is_last ? xe[0] : xs[0], s2 = is_last ? xe[NOPS > 1 ? 1 : 0] : xs[NOPS > 1 ? 1 : 0],
                             s3 = is_last ? xe[NOPS > 2 ? 2 : 0] : xs[NOPS > 2 ? 2 : 0];
                r = -0.5 * (v * s1 * a.St[j] + s2 * b2.St[j]) + nu * (s3 * d2.St[j] + s1 * a.St[j] * d2.Stc[j]);
            } else {  // :315-324
                const double temp_du = a.St[j] * (c[0] - a.Sa[j] * xs[0] - a.Sc[j] * xe[0]);
                const double temp_dud =
                    b2.St[j] * (c[NOPS > 1 ? 1 : 0] - b2.Sa[j] * xs[NOPS > 1 ? 1 : 0] - b2.Sc[j] * xe[NOPS > 1 ? 1 : 0]);
                const double temp_d2u =
                    d2.St[j] * (c[NOPS > 2 ? 2 : 0] - d2.Sa[j] * xs[NOPS > 2 ? 2 : 0] - d2.Sc[j] * xe[NOPS > 2 ? 2 : 0]) +
                    temp_du * d2.Stc[j];
                r = -0.5 * (v * temp_du + temp_dud) + nu * temp_d2u;
            }
            r = ACC ? out[o_] + r : r;
        }
        out[o_] = r;
    };

    // ---------------- backward sweep, block by block from the end
    double nxt[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; o++) nxt[o] = 0.0;
    const int nb = (n + CK - 1) / CK;
    for (int b = nb - 1; b >= 0; b--) {
        const int j0 = b * CK;
#pragma unroll
        for (int o = 0; o < NOPS; o++) prev[o] = b == 0 ? 0.0 : ckpt[((long)b * NOPS + o) * np + p];
        double dl[CK][NOPS];
        load_window(j0 + 1);
#pragma unroll
        for (int q = 0; q < CK; q++) {
            const int j = j0 + 1 + q;
            if (j <= n) {
                double e[NOPS];
                eliminate(j, e);
#pragma unroll
                for (int o = 0; o < NOPS; o++) dl[q][o] = e[o];
                if (q + 1 < CK) shift_in(j + 5);
            }
        }
#pragma unroll
        for (int q = CK - 1; q >= 0; q--) {
            const int j = j0 + 1 + q;
            if (j <= n) {
                double c[NOPS];
#pragma unroll
                for (int o = 0; o < NOPS; o++) {
                    // rows n and n-1 keep their forward values (distributed.f90:154: j = n-2..2)
                    c[o] = (j >= n - 1) ? dl[q][o] : dl[q][o] - T.t[o].Bw[j] * nxt[o];
                }
                emit(j, c, j == 1, j == n);
                if (j > 1) {
#pragma unroll
                    for (int o = 0; o < NOPS; o++) nxt[o] = c[o];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ launchers
static double *ckpt_buf(x3d_backend *b) { return b->scratch[2]; }

int x3d_fused_tds_local(x3d_backend *b, double *du, const double *u, const x3d_tdsops *t, int dir, int acc,
                        double scale)
{
    PencilGeom g = x3d_geom(b, dir);
    Tabs3 T;
    T.t[0] = T.t[1] = T.t[2] = t->tab;
    ProfScope ps(b, X3D_K_TDS_FWD, dir);
    dim3 grid((g.np + 63) / 64);
    if (acc)
        hipLaunchKernelGGL((k_fused<1, true, true>), grid, dim3(64), 0, b->stream, du, u, u, ckpt_buf(b), T, g,
                           t->n_tds, 0.0, scale);
    else
        hipLaunchKernelGGL((k_fused<1, true, false>), grid, dim3(64), 0, b->stream, du, u, u, ckpt_buf(b), T, g,
                           t->n_tds, 0.0, 1.0);
    X3D_HIP(hipGetLastError());
    return 0;
}

int x3d_fused_transeq_local(x3d_backend *b, int dir, double *rhs, const double *u, const double *conv, double nu,
                            const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc)
{
    PencilGeom g = x3d_geom(b, dir);
    Tabs3 T;
    T.t[0] = t1->tab; T.t[1] = t2->tab; T.t[2] = t3->tab;
    ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
    dim3 grid((g.np + 63) / 64);
    const bool same = (u == conv);
#define LAUNCH(S_, A_)                                                                                         \
    hipLaunchKernelGGL((k_fused<3, S_, A_>), grid, dim3(64), 0, b->stream, rhs, u, conv, ckpt_buf(b), T, g,     \
                       t1->n_tds, nu, 1.0)
    if (same && acc) LAUNCH(true, true);
    else if (same) LAUNCH(true, false);
    else if (acc) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    X3D_HIP(hipGetLastError());
    return 0;
}
