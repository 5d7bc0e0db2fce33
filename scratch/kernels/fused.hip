// Block-recompute backward sweep of the DistD2 solve (kernel family K2).
//
// The two-sweep kernels of tds.hip stream the forward-eliminated values d_j
// through HBM (3 arrays written and re-read per transport-equation component).
// Here the forward kernel keeps only one checkpoint of d every X3D_CK rows; this
// kernel walks each pencil block by block from its end, re-reads the CK+8 input
// rows of the block, recomputes the block's d_j in registers starting from the
// checkpoint and back-substitutes through it, fused with the reduced-system
// substitution, the stretching factors and (transeq) the skew-symmetric
// combination.  Per component: read u, conv twice (1x forward, ~1.5x here),
// write rhs once: ~7 field passes instead of 10-11.
//
// Arithmetic and operation order are the reference's:
//   src/backend/omp/kernels/distributed.f90:11-168  (der_univ_dist)
//   src/backend/omp/kernels/distributed.f90:170-337 (der_univ_subs / _fused_subs)
#include "common.h"

#define CK X3D_CK

template <bool HB>
__device__ __forceinline__ double ext_row_b(const double *__restrict__ u, long base, long rs, int jj, int nr,
                                            int n_wrap, const double *__restrict__ hs,
                                            const double *__restrict__ he, int np, int p)
{
    if (jj < 1) {
        if (HB) return hs[(long)(jj + 3) * np + p];
        return u[base + (long)(n_wrap + jj - 1) * rs];
    }
    if (jj > nr) {
        const int r = jj - nr - 1;
        if (HB) return he[(long)r * np + p];
        return u[base + (long)r * rs];
    }
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
struct Halos { const double *us, *ue, *cs, *ce; };

// NOPS = 1: tds_solve (out = [out + scale *] T(u))
// NOPS = 3: transport-equation component, operators t[0]=du, t[1]=dud, t[2]=d2u
template <int NOPS, bool SAME, bool ACC, bool HB>
__global__ void __launch_bounds__(64)
    k_ck_bwd(double *out, const double *__restrict__ u, const double *__restrict__ cv,
             const double *__restrict__ ckpt, const double *__restrict__ own_s, const double *__restrict__ recv_s,
             const double *__restrict__ recv_e, int bstride, Halos h, Tabs3 T, PencilGeom g, int n_wrap, double nu,
             double scale)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int n = T.t[0].n_tds, nr = T.t[0].n_rhs;
    const long np = g.np;

    double wu[9], wp[9], prev[NOPS];

    auto in_u = [&](int jj) { return ext_row_b<HB>(u, base, rs, jj, nr, n_wrap, h.us, h.ue, g.np, p); };
    auto in_c = [&](int jj) { return ext_row_b<HB>(cv, base, rs, jj, nr, n_wrap, h.cs, h.ce, g.np, p); };
    auto eliminate = [&](int j, double (&e)[NOPS]) {
#pragma unroll
        for (int o = 0; o < NOPS; o++) {
            const TdsTab &t = T.t[o];
            const double acc = dot9f(stencil_row_f(t.Cs, j, nr), (NOPS == 3 && o == 1) ? wp : wu);
            e[o] = T_F(t, j) * (acc - T_A(t, j) * prev[o]);
            prev[o] = e[o];
        }
    };

    // reduced 2x2 systems (distributed.f90:186-206); d_n of this pencil is recomputed below,
    // so it is taken from the exchange buffers: own_e == what this rank sent as send_e
    double xs[NOPS], xe[NOPS];

    auto emit = [&](int j, const double (&c)[NOPS], bool is_first, bool is_last) {
        const long o_ = base + (long)(j - 1) * rs;
        double r;
        if (NOPS == 1) {
            const TdsTab &t = T.t[0];
            if (is_last) r = xe[0] * T_ST(t, j);
            else if (is_first) r = xs[0] * T_ST(t, j);
            else r = (c[0] - T_SA(t, j) * xs[0] - T_SC(t, j) * xe[0]) * T_ST(t, j);
            r = ACC ? out[o_] + scale * r : r;
        } else {
            const TdsTab &a = T.t[0], &b2 = T.t[NOPS > 1 ? 1 : 0], &d2 = T.t[NOPS > 2 ? 2 : 0];
            const int i1 = NOPS > 1 ? 1 : 0, i2 = NOPS > 2 ? 2 : 0;
            const double v = SAME ? u[o_] : cv[o_];
            if (is_last || is_first) {  // distributed.f90:304-311, 328-335
                const double s1 = is_last ? xe[0] : xs[0], s2 = is_last ? xe[i1] : xs[i1],
                             s3 = is_last ? xe[i2] : xs[i2];
                r = -0.5 * (v * s1 * T_ST(a, j) + s2 * T_ST(b2, j)) + nu * (s3 * T_ST(d2, j) + s1 * T_ST(a, j) * T_STC(d2, j));
            } else {  // :315-324
                const double temp_du = T_ST(a, j) * (c[0] - T_SA(a, j) * xs[0] - T_SC(a, j) * xe[0]);
                const double temp_dud = T_ST(b2, j) * (c[i1] - T_SA(b2, j) * xs[i1] - T_SC(b2, j) * xe[i1]);
                const double temp_d2u =
                    T_ST(d2, j) * (c[i2] - T_SA(d2, j) * xs[i2] - T_SC(d2, j) * xe[i2]) + temp_du * T_STC(d2, j);
                r = -0.5 * (v * temp_du + temp_dud) + nu * temp_d2u;
            }
            r = ACC ? out[o_] + r : r;
        }
        out[o_] = r;
    };

    double nxt[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; o++) nxt[o] = 0.0;
    const int nb = (n + CK - 1) / CK;
    for (int b = nb - 1; b >= 0; b--) {
        const int j0 = b * CK;
#pragma unroll
        for (int o = 0; o < NOPS; o++) prev[o] = b == 0 ? 0.0 : ckpt[((long)b * NOPS + o) * np + p];
        double dl[CK][NOPS];
#pragma unroll
        for (int m = 0; m < 9; m++) {
            wu[m] = in_u(j0 - 3 + m);
            if (NOPS == 3) wp[m] = wu[m] * (SAME ? wu[m] : in_c(j0 - 3 + m));
        }
#pragma unroll
        for (int q = 0; q < CK; q++) {
            const int j = j0 + 1 + q;
            if (j <= n) {
                double e[NOPS];
                eliminate(j, e);
#pragma unroll
                for (int o = 0; o < NOPS; o++) dl[q][o] = e[o];
                if (q + 1 < CK && j + 1 <= n) {
                    const double a = in_u(j + 5);
                    double b2 = 0.0;
                    if (NOPS == 3) b2 = a * (SAME ? a : in_c(j + 5));
#pragma unroll
                    for (int m = 0; m < 8; m++) { wu[m] = wu[m + 1]; if (NOPS == 3) wp[m] = wp[m + 1]; }
                    wu[8] = a;
                    if (NOPS == 3) wp[8] = b2;
                }
            }
        }
        if (b == nb - 1) {
            // first block processed holds row n: d_n is now known -> close the 2x2 systems
            const int qn = n - 1 - j0;
#pragma unroll
            for (int o = 0; o < NOPS; o++) {
                double dn = 0.0;
#pragma unroll
                for (int q = 0; q < CK; q++) if (q == qn) dn = dl[q][o];
                const TdsTab &t = T.t[o];
                xs[o] = t.rs_s * (own_s[(long)o * bstride + p] - t.sa1 * recv_s[(long)o * bstride + p]);
                xe[o] = t.rs_e * (dn - t.scn * recv_e[(long)o * bstride + p]);
            }
        }
#pragma unroll
        for (int q = CK - 1; q >= 0; q--) {
            const int j = j0 + 1 + q;
            if (j <= n) {
                double c[NOPS];
#pragma unroll
                for (int o = 0; o < NOPS; o++)  // rows n, n-1 keep their forward values (:154: j = n-2..2)
                    c[o] = (j >= n - 1) ? dl[q][o] : dl[q][o] - T_BW(T.t[o], j) * nxt[o];
                emit(j, c, j == 1, j == n);
#pragma unroll
                for (int o = 0; o < NOPS; o++) nxt[o] = c[o];
            }
        }
    }
}

// ------------------------------------------------------------------ launchers
int x3d_ck_bwd_tds(x3d_backend *b, double *du, const double *u, const double *hs, const double *he,
                   const double *ckpt, const double *own_s, const double *recv_s, const double *recv_e,
                   const x3d_tdsops *t, int dir, int acc, double scale)
{
    PencilGeom g = x3d_geom(b, dir);
    Tabs3 T;
    T.t[0] = T.t[1] = T.t[2] = t->tab;
    Halos h{hs, he, hs, he};
    ProfScope ps(b, X3D_K_TDS_BWD, dir);
    dim3 grid((g.np + 63) / 64);
    const bool hb = hs != nullptr;
#define LAUNCH(A_, H_)                                                                                         \
    hipLaunchKernelGGL((k_ck_bwd<1, true, A_, H_>), grid, dim3(64), 0, b->stream, du, u, u, ckpt, own_s,      \
                       recv_s, recv_e, g.np, h, T, g, t->n_tds, 0.0, scale)
    if (acc && hb) LAUNCH(true, true);
    else if (acc) LAUNCH(true, false);
    else if (hb) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    X3D_HIP(hipGetLastError());
    return 0;
}

int x3d_ck_bwd_transeq(x3d_backend *b, int dir, double *rhs, const double *u, const double *us, const double *ue,
                       const double *conv, const double *cs, const double *ce, const double *ckpt,
                       const double *own_s, const double *recv_s, const double *recv_e, int bstride, double nu,
                       const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc)
{
    PencilGeom g = x3d_geom(b, dir);
    Tabs3 T;
    T.t[0] = t1->tab; T.t[1] = t2->tab; T.t[2] = t3->tab;
    Halos h{us, ue, cs, ce};
    ProfScope ps(b, X3D_K_TRANSEQ_BWD, dir);
    dim3 grid((g.np + 63) / 64);
    const bool same = (u == conv), hb = us != nullptr;
#define LAUNCH(S_, A_, H_)                                                                                     \
    hipLaunchKernelGGL((k_ck_bwd<3, S_, A_, H_>), grid, dim3(64), 0, b->stream, rhs, u, conv, ckpt, own_s,    \
                       recv_s, recv_e, bstride, h, T, g, t1->n_tds, nu, 1.0)
    if (hb) {
        if (acc) LAUNCH(false, true, true);
        else LAUNCH(false, false, true);
    } else if (same) {
        if (acc) LAUNCH(true, true, false);
        else LAUNCH(true, false, false);
    } else {
        if (acc) LAUNCH(false, true, false);
        else LAUNCH(false, false, false);
    }
#undef LAUNCH
    X3D_HIP(hipGetLastError());
    return 0;
}
