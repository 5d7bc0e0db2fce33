// Single-pass transport-equation component for periodic 512-row y / z pencils (kernel family K1t).
//
// The two-sweep kernels (tds.hip) move 11 field passes per component: u, conv in; three
// forward-eliminated arrays out and back in; conv again; rhs read-modify-write.  Here the pencil
// stays on chip: a workgroup (8 waves) owns 16 pencils (128 contiguous bytes per row), the four
// quarters of a wave are four different 16-row chunks (lane = x + 16 h, chunk = 4 wave + h), and
// the three operators are solved one after the other with the chunk-parallel scheme of
// onchip.hip (chunk-local sweeps from zero + exact carry correction through LDS), each one
// substituted at once and accumulated into the result held in registers:
//   r = -1/2 (v du + dud) + nu (d2u + du stc)          src/backend/omp/kernels/distributed.f90:231-337
// HBM traffic: u, conv in, rhs out (+ rhs in when accumulating); the second reads of u / conv
// come from L2 / Infinity Cache.  Row tables are staged in LDS per operator ([table][row]: the row
// index differs between the quarters of a wave, so scalar loads cannot serve it).
// STATUS: opt-in (X3D_ONCHIP_TRANSEQ=1), parity-tested, not faster yet: 2.5 ms (conv != u) / 2.8 ms per
// component against 2.3 / 2.1 ms for the two-sweep pair, although it moves 4 HBM passes instead of 11.
// History: written as three consecutive operator blocks LLVM interleaved them (350 spilled dwords, 6.4 ms);
// as a real loop over the operators with per-trip opaque offsets (otherwise the loop-invariant loads of u
// are hoisted and kept for all trips) it is down to 40-100 spilled dwords.  What remains is latency: a
// workgroup runs 3 operators x 4 barrier-separated phases with two serial 31-step carry chains each, and only
// two workgroups fit a CU.  Next: log-step carry scan, all three table sets resident, 16 rows x 32 pencils.
//   src/backend/omp/kernels/distributed.f90:11-168   der_univ_dist
//   src/backend/omp/exec_dist.f90:67-186             exec_dist_transeq_compact
#include "common.h"

#define TQ_M 16   // rows per chunk
#define TQ_C 32   // chunks per pencil
#define TQ_LR 520 // padded rows per LDS table


// tables in LDS: FA[j][2], PH[j][2] (PF, HB), SS[j][2] (SA, SC), QB[j], ST[j], STC3[j]
// keeps the LDS table reads of a row next to their use: without it LLVM reads the tables of all 16 rows
// (and of the loops that follow) up front and spills them straight to scratch
#define ROW_FENCE()                                \
    do {                                           \
        __asm__ volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

struct TqLds {
    double2 *FA, *PH, *SS;
    double *QB, *ST, *STC3, *ends, *starts, *misc;
};

// what the kernel needs of one operator (kernel arguments: SGPRs)
struct TqOp {
    const double *RF, *RB;
    double last_r, bw1, rs_s, rs_e, sa1, scn;
    double c[9];
};

__device__ __forceinline__ void tq_stage_tables(const TqLds &L, const double *RF, const double *RB,
                                                const double *RB3)
{
    constexpr int n = 512;
    // laundered: otherwise the staging loads of all three operators are hoisted to the top of the kernel
    // (read-only memory, nothing orders them) and ~100 VGPRs of table values are spilled until their turn
    for (int j = threadIdx.x; j < TQ_LR; j += blockDim.x) {
        const bool in = j >= 1 && j <= n;
        int jj = in ? j : 1;
        asm volatile("" : "+v"(jj));  // (laundering the pointers instead turns the loads into flat loads)
        const double f = RF[4 * jj], a = RF[4 * jj + 1];
        const double bw = RB[8 * jj], sa = RB[8 * jj + 1], sc = RB[8 * jj + 2], st = RB[8 * jj + 3],
                     pf16 = RB[8 * jj + 6], qb16 = RB[8 * jj + 7], stc3 = RB3[8 * jj + 4];
        const double hb = (j >= 2 && j <= n - 2) ? -bw : 0.0;  // rows 1, n-1, n: no backward update
        L.FA[j] = in ? make_double2(f, a) : make_double2(0.0, 0.0);
        L.PH[j] = in ? make_double2(pf16, hb) : make_double2(0.0, 0.0);
        L.SS[j] = in ? make_double2(sa, sc) : make_double2(0.0, 0.0);
        L.QB[j] = in ? qb16 : 0.0;
        L.ST[j] = in ? st : 0.0;
        L.STC3[j] = in ? stc3 : 0.0;
    }
}

// one operator on this lane's chunk: x[] holds the 16 input rows, hl the 4 rows before; the 4 rows after
// are fetched late through load_hr(m).  On return x[] holds the back-substituted chunk-local values; the
// backward carry and the reduced-system values s_, e_ are returned and applied by TQ_SUBS.
template <class LoadHr>
__device__ __forceinline__ void tq_solve(double (&x)[TQ_M], const double (&hl)[4], LoadHr &&load_hr, const TqLds &L,
                                         const TqOp &t, int s, int c, int xl, double &s_out, double &e_out,
                                         double &carry_out)
{
    constexpr int M = TQ_M;
    const double c0 = t.c[0], c1 = t.c[1], c2 = t.c[2], c3 = t.c[3], c4 = t.c[4], c5 = t.c[5], c6 = t.c[6],
                 c7 = t.c[7], c8 = t.c[8];
    double p0 = hl[0], p1 = hl[1], p2 = hl[2], p3 = hl[3], hr[4];
    double prev = 0.0;
#pragma unroll
    for (int q = 0; q < M; q++) {
        if (q == 4) {
#pragma unroll
            for (int m = 0; m < 4; m++) hr[m] = load_hr(m);
        }
#define AHEAD(d) ((q + (d) < M) ? x[(q + (d)) % M] : hr[(q + (d) - M) & 3])
        const double2 fa = L.FA[s + q];
        const double cur = x[q];
        const double acc = c0 * p0 + c1 * p1 + c2 * p2 + c3 * p3 + c4 * cur + c5 * AHEAD(1) + c6 * AHEAD(2) +
                           c7 * AHEAD(3) + c8 * AHEAD(4);
#undef AHEAD
        const double e = fa.x * (acc - fa.y * prev);
        prev = e;
        x[q] = e;
        p0 = p1; p1 = p2; p2 = p3; p3 = cur;
        if (q & 1) ROW_FENCE();
    }
    L.ends[c * 16 + xl] = prev;
    __syncthreads();
    {   // forward carry, chunk-local back-substitution
        double carry = 0.0;
        for (int cc = 0; cc < c; cc++) carry = L.ends[cc * 16 + xl] + L.PH[(cc + 1) * M].x * carry;
        double nxt = 0.0;
#pragma unroll
        for (int q = M - 1; q >= 0; q--) {
            const double2 ph = L.PH[s + q];
            x[q] = (x[q] + ph.x * carry) + ph.y * nxt;
            nxt = x[q];
            if ((q & 1) == 0) ROW_FENCE();
        }
        L.starts[c * 16 + xl] = x[0];
    }
    __syncthreads();
    // backward carry (applied on the fly by TQ_SUBS); publish du_1 and X_n
    double carry = 0.0;
    for (int cc = TQ_C - 1; cc > c; cc--) carry = L.starts[cc * 16 + xl] + L.QB[cc * M + 1] * carry;
    if (c == TQ_C - 1) L.misc[16 + xl] = x[M - 1];  // carry = 0 there
    if (c == 0)
        L.misc[xl] = t.last_r * ((x[0] + L.QB[1] * carry) - t.bw1 * (x[1] + L.QB[2] * carry));  // :161-166
    __syncthreads();
    const double du1 = L.misc[xl], xn = L.misc[16 + xl];
    s_out = t.rs_s * (du1 - t.sa1 * xn);  // periodic self-exchange (sendrecv.f90:20-22)
    e_out = t.rs_e * (xn - t.scn * du1);
    carry_out = carry;
}

// substitution of row q (distributed.f90:304-335 written per operator: rows 1 and n take du_s * st / du_e * st)
#define TQ_SUBS(q)                                                                   \
    ({                                                                               \
        const int j_ = s + (q);                                                      \
        const double2 ss_ = L.SS[j_];                                                \
        const double st_ = L.ST[j_];                                                 \
        const double X_ = x[q] + L.QB[j_] * cy;                                      \
        double v_ = st_ * (X_ - ss_.x * s_ - ss_.y * e_);                            \
        v_ = (j_ == 1) ? s_ * st_ : v_;                                              \
        v_ = (j_ == 512) ? e_ * st_ : v_;                                            \
        v_;                                                                          \
    })

struct TqOps { TqOp o[3]; };  // order of evaluation: d(u conv)/dx, du/dx, d2u/dx2

template <bool SAME, bool ACC>
__global__ void __launch_bounds__(512, 4)  // two workgroups per CU
    k_transeq_onchip(double *__restrict__ rhs, const double *__restrict__ u, const double *__restrict__ cv, TqOps P,
                     PencilGeom g, double nu)
{
    extern __shared__ double lds[];
    constexpr int M = TQ_M, n = 512;
    TqLds L;
    L.FA = reinterpret_cast<double2 *>(lds);
    L.PH = L.FA + TQ_LR;
    L.SS = L.PH + TQ_LR;
    L.QB = reinterpret_cast<double *>(L.SS + TQ_LR);
    L.ST = L.QB + TQ_LR;
    L.STC3 = L.ST + TQ_LR;
    L.ends = L.STC3 + TQ_LR;
    L.starts = L.ends + TQ_C * 16;
    L.misc = L.starts + TQ_C * 16;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xl = lane & 15, c = 4 * wv + (lane >> 4);
    const int p = blockIdx.x * 16 + xl;
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int s = c * M + 1;
    const unsigned off = (unsigned)(base + (long)(s - 1) * rs);
#define offl(m) ((unsigned)(base + (long)((s - 5 + (m) + n) & (n - 1)) * rs))
#define offr(m) ((unsigned)(base + (long)((s + M - 1 + (m)) & (n - 1)) * rs))
    double r[M];
#pragma unroll
    for (int q = 0; q < M; q++) r[q] = 0.0;
    // One operator per trip of a REAL loop: written as three consecutive blocks, LLVM interleaves the
    // operators (table reads and address arithmetic of the next one hoisted over the current one) and spills
    // ~350 dwords; a loop body is scheduled on its own and fits (one operator alone needs 76 VGPRs).
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
        const TqOp &op = P.o[k];
        tq_stage_tables(L, op.RF, op.RB, P.o[2].RB);
        // opaque per-trip offsets: u and conv are loop-invariant, and LLVM would otherwise hoist their loads
        // out of the operator loop and keep 24-48 rows in registers for all three trips
        unsigned offk = off;
        long basek = base;
        asm volatile("" : "+v"(offk), "+v"(basek));
#define offlk(m) ((unsigned)(basek + (long)((s - 5 + (m) + n) & (n - 1)) * rs))
#define offrk(m) ((unsigned)(basek + (long)((s + M - 1 + (m)) & (n - 1)) * rs))
        double x[M], hl[4];
#pragma unroll
        for (int q = 0; q < M; q++) x[q] = (u + (long)q * rs)[offk];
#pragma unroll
        for (int m = 0; m < 4; m++) hl[m] = u[offlk(m)];
        const bool prod = k == 0;  // the first operator acts on u * conv
        if (prod) {
            if (SAME) {
#pragma unroll
                for (int q = 0; q < M; q++) x[q] = x[q] * x[q];
#pragma unroll
                for (int m = 0; m < 4; m++) hl[m] = hl[m] * hl[m];
            } else {
#pragma unroll
                for (int q0 = 0; q0 < M; q0 += 4) {
                    double t4[4];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) t4[kk] = (cv + (long)(q0 + kk) * rs)[offk];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) x[q0 + kk] *= t4[kk];
                }
#pragma unroll
                for (int m = 0; m < 4; m++) hl[m] *= cv[offlk(m)];
            }
        }
        __syncthreads();  // tables staged
        double s_, e_, cy;
        tq_solve(x, hl,
                 [&](int m) {
                     const double a = u[offrk(m)];
                     return prod ? (SAME ? a * a : a * cv[offrk(m)]) : a;
                 },
                 L, op, s, c, xl, s_, e_, cy);
        // the substituted values are consumed in the loop that produces them
        if (k == 0) {
#pragma unroll
            for (int q = 0; q < M; q++) {
                r[q] = -0.5 * TQ_SUBS(q);
                if (q & 1) ROW_FENCE();
            }
        } else if (k == 1) {
            const double *__restrict__ vp = SAME ? u : cv;
#pragma unroll
            for (int q0 = 0; q0 < M; q0 += 4) {
                double v4[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) v4[kk] = (vp + (long)(q0 + kk) * rs)[offk];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int q = q0 + kk;
                    const double d = TQ_SUBS(q);
                    r[q] = r[q] - 0.5 * (v4[kk] * d) + nu * (d * L.STC3[s + q]);
                }
                ROW_FENCE();
            }
        } else {
#pragma unroll
            for (int q0 = 0; q0 < M; q0 += 4) {
                double o4[4];
                if (ACC) {
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) o4[kk] = (rhs + (long)(q0 + kk) * rs)[off];
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int q = q0 + kk;
                    const double val = r[q] + nu * TQ_SUBS(q);
                    (rhs + (long)q * rs)[off] = ACC ? o4[kk] + val : val;
                }
                ROW_FENCE();
            }
        }
        __syncthreads();  // tables and carries are reused by the next operator
    }
}

int x3d_onchip_transeq(x3d_backend *b, int dir, double *rhs, const double *u, const double *conv, double nu,
                       const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, bool *done)
{
    *done = false;
    PencilGeom g = x3d_geom(b, dir);
    const x3d_tdsops *ts[3] = {t1, t2, t3};
    for (int o = 0; o < 3; o++)
        if (!(ts[o]->tab.bulk_only && ts[o]->n_tds == 512 && ts[o]->n_rhs == 512)) return 0;
    if (g.dim0 % 16 != 0 || g.np % 16 != 0) return 0;
    TqOps P;
    const int order[3] = {1, 0, 2};  // evaluation order: dud (t2), du (t1), d2u (t3)
    for (int o = 0; o < 3; o++) {
        const x3d_tdsops *tt = ts[order[o]];
        const TdsTab &tb = tt->tab;
        P.o[o] = TqOp{tb.RF, tb.RB, tb.last_r, tb.bw1, tb.rs_s, tb.rs_e, tb.sa1, tb.scn, {0}};
        for (int m = 0; m < 9; m++) P.o[o].c[m] = tt->coeffs[m];
    }
    const size_t lds = sizeof(double) * ((size_t)9 * TQ_LR + 2 * TQ_C * 16 + 32);
    const bool same = u == conv;
    ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
    dim3 grid(g.np / 16), block(512);
#define GO(S_, A_)                                                                                          \
    do {                                                                                                    \
        static bool attr = false;                                                                           \
        if (!attr) {                                                                                        \
            X3D_HIP(hipFuncSetAttribute((const void *)k_transeq_onchip<S_, A_>,                             \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
            attr = true;                                                                                    \
        }                                                                                                   \
        hipLaunchKernelGGL((k_transeq_onchip<S_, A_>), grid, block, lds, b->stream, rhs, u, conv, P, g, nu); \
    } while (0)
    if (same) { if (acc) GO(true, true); else GO(true, false); }
    else { if (acc) GO(false, true); else GO(false, false); }
#undef GO
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}
