// K3g: the LDS-tile kernels of xscan.hip (K3y) for y / z pencils of ANY length up to 512 rows and any boundary
// closure -- non-periodic ends with their own boundary stencils, v2p operators with n_rhs = n_tds + 1, stretched
// grids (channel case: 257 wall-normal vertices, Dirichlet, BASELINE configs[4]).  Same data movement: a workgroup
// of 16 waves owns the 16 x-adjacent pencils of one plane, reads the [rows][16 x] tile in 128-byte row segments,
// every wave solves its pencil out of LDS with the scan solver, the workgroup writes the tile back: every field
// is touched once (transeq component: 3 passes, an operator pair: 3 instead of 4 + 4 of the two-sweep kernels).
//
// What differs from the periodic fast forms:
//  * the tile keeps 4 zero rows before row 1 and zeros after the field's last row; a lane reads its whole window
//    (rows first-4 .. first+Q+3) as one contiguous piece of LDS, no DPP halo moves, no periodic wrap: the boundary
//    stencils have zero weight outside the pencil (src/tdsops.f90:357-533), the reference multiplies whatever its
//    halo buffers hold by those zeros;
//  * scan_solve's general form (boundary rows with their own stencils, staged in LDS; lane tables with zero
//    multipliers beyond row n; X_n picked from the lane that holds it);
//  * loads cover the block's rows (ny / nz), stores the operator's n_tds rows.
// Arithmetic per operator = k_xscan_tds / k_xscan_transeq (general form): the reference's distributed.f90:34-166
// sweeps and :186-337 substitution, re-associated by the scans.
#include "xscan_core.h"

template <int Q> struct GenGeom {
    static constexpr int NMAX = 64 * Q;       // rows the 64 lanes cover
    static constexpr int TP = NMAX + 10;      // doubles per pencil in the tile: 4 zero rows, NMAX rows, 4 + 2 pad
    static constexpr int NI = (NMAX + 127) / 128;
};

// ---------------------------------------------------------------- DIRECT form (round 6)
// A non-periodic operator on one rank is a plain tridiagonal system; tds.hip recovers it from the preprocessed arrays,
// factors it (F_j = 1 / (b_j - a_j g_{j-1}), g_j = c_j F_j) and checks the factors against the reference's sweeps.  The
// solve is then the Thomas recurrences e_j = F_j r_j + FA_j e_{j-1}, x_j = e_j + H_j x_{j+1} (FA = -F a, H = -g), made
// parallel over the lanes exactly like scan_solve (lane-local sweeps from zero, Kogge-Stone scans of the lane-end values
// with precomputed multipliers, carries applied) -- and that is all: no reduced 2 x 2 system, no SA / SC spike reads, no
// du_1 / X_n broadcasts, no row selects (scan_solve + gen_solve: ~154 vector instructions per right-hand side at Q = 5,
// this: ~100), one dependent FMA per row in the sweeps instead of two.  Same linear system as the reference's
// distributed.f90:34-229 with both neighbours absent; results differ by round-off (1e-15 relative, tds.hip's check).
// LP: lanes per pencil.  32 = TWO pencils per wave (lanes 0..31, 32..63): the scans' cross-row steps shrink to one and
// their cost is shared.  XR (LP * Q + 1 rows, the channel's 257 = 32 * 8 + 1): the pencil's last lane also carries row
// LP Q + 1 -- its right-hand side from the lane's own window (rows first+Q-4 .. first+Q), its forward step after the carry
// is applied, and its value starts that lane's backward sweep; xr returns the row's result.
#define DT_F(q) (0 * Q + (q))
#define DT_FA(q) (1 * Q + (q))
#define DT_PF(q) (2 * Q + (q))
#define DT_H(q) (3 * Q + (q))
#define DT_QB(q) (4 * Q + (q))
#define DT_ST(q) (5 * Q + (q))
#define DT_MF(k) (6 * Q + (k))
#define DT_MB(k) (6 * Q + 6 + (k))
#define DT_X(k) (6 * Q + 12 + (k))     // XR: F, FA, ST, STC of row LP Q + 1 (zeros otherwise)
#define DT_STC(q) (6 * Q + 16 + (q))
#define DT_N(Q_) (7 * (Q_) + 16)
#define DT_NC(Q_) (6 * (Q_) + 16)      // entries without the STC block
// stencil slots (stage_cs): lane 0, lanes ls and ls + 1 of the pencil, bulk
template <int Q, int LP>
__device__ __forceinline__ int cs_slot(int lane, int n_rhs)
{
    const int lr = lane & (LP - 1), ls = (n_rhs - 4) / Q;
    return (lr == 0 ? 0 : (lr == ls ? 1 : (lr == ls + 1 ? 2 : 3))) * (Q * 10);
}
// lane: the lane's index in its pencil (LP = 32: lane & 31; the tables are staged [entry][LP]), hi: upper half of the wave
// mid(x): called once the window is dead (the right-hand sides are formed) with a value the caller can tie its loads to --
// the place to issue global loads whose destination registers would not fit beside the window
struct NoMid { __device__ __forceinline__ void operator()(real_t) const {} };
template <int Q, int LP, bool NARROW, bool XR = false, class T = real_t, bool WD2 = true, class MID = NoMid>
__device__ __forceinline__ void thomas_solve(const T (&w)[Q + 8], T (&X)[Q], T &xr, const real_t *__restrict__ lt,
                                             const real_t *__restrict__ cs, int co, int &lane, bool hi, MID mid = MID())
{
#define PHASE(x) asm volatile("" : "+v"(lane) : "v"(first_of(x)))
#define DTR(e) lt_read(lt, (e) * LP + lane)
    T acc[Q], accx = zero_of<T>();
    {
        constexpr bool D2 = WD2 && (NARROW || sizeof(T) == sizeof(real_t));
        int co1 = co;
        if constexpr (D2) asm volatile("" : "+v"(co1));
        if constexpr (XR) {  // row LP Q + 1 = window entry Q + 4; its (end) stencil reaches back only: entries Q .. Q + 4
            const real2_t *__restrict__ c2 = reinterpret_cast<const real2_t *>(cs + CS_N(Q));
            const real2_t ca = c2[0], cb = c2[1], cc = c2[2];
            accx = ca.x * w[Q] + ca.y * w[Q + 1] + cb.x * w[Q + 2] + cb.y * w[Q + 3] + cc.x * w[Q + 4];
            asm volatile("" : "+v"(co) : "v"(first_of(accx)));
        }
#pragma unroll
        for (int q = Q - 1; q >= 0; q--) {  // (from the far end: the window's entries die as the sums are formed)
            int &coq = (D2 && (q & 1)) ? co1 : co;
            const real2_t *__restrict__ c2 = reinterpret_cast<const real2_t *>(cs + coq + q * 10);
            if (NARROW) {
                const real2_t cb = c2[1], cc = c2[2], cd = c2[3];
                acc[q] = cb.x * w[q + 2] + cb.y * w[q + 3] + cc.x * w[q + 4] + cc.y * w[q + 5] + cd.x * w[q + 6];
            } else {
                const real2_t ca = c2[0], cb = c2[1], cc = c2[2], cd = c2[3], ce = c2[4];
                acc[q] = ca.x * w[q] + ca.y * w[q + 1] + cb.x * w[q + 2] + cb.y * w[q + 3] + cc.x * w[q + 4] +
                         cc.y * w[q + 5] + cd.x * w[q + 6] + cd.y * w[q + 7] + ce.x * w[q + 8];
            }
            asm volatile("" : "+v"(coq) : "v"(first_of(acc[q])));
        }
        mid(first_of(acc[0]));
    }
    // ---- lane-local forward sweep from zero
    T prev = zero_of<T>();
#pragma unroll
    for (int q = 0; q < Q; q++) {
        X[q] = DTR(DT_F(q)) * acc[q] + DTR(DT_FA(q)) * prev;
        prev = X[q];
    }
    T v = prev;
    PHASE(X[Q / 2]);
    v += DTR(DT_MF(0)) * dpp0<0x111>(v);
    v += DTR(DT_MF(1)) * dpp0<0x112>(v);
    v += DTR(DT_MF(2)) * dpp0<0x114>(v);
    v += DTR(DT_MF(3)) * dpp0<0x118>(v);
    v += DTR(DT_MF(4)) * dpp0<0x142>(v);
    if constexpr (LP == 64) v += DTR(DT_MF(5)) * dpp0<0x143>(v);
    T carry = dpp0<0x138>(v);  // wave_shr:1 (lane 32 of the two-pencil form: PF = 0, its pencil starts there)
    T nxt = zero_of<T>();
    if constexpr (XR) {
        // v = e at this lane's last row; in the pencil's last lane that is row LP Q: the extra row's forward step and,
        // being the system's last row, its solution
        const T ex = DTR(DT_X(0)) * accx + DTR(DT_X(1)) * v;
        nxt = sel_of(lane == LP - 1, ex, nxt);
        xr = ex * DTR(DT_X(2));
    }
    PHASE(carry);
#pragma unroll
    for (int q = Q - 1; q >= 0; q--) {
        X[q] = (X[q] + DTR(DT_PF(q)) * carry) + DTR(DT_H(q)) * nxt;
        nxt = X[q];
    }
    v = nxt;
    PHASE(X[Q / 2]);
    v += DTR(DT_MB(0)) * dpp0<0x101>(v);
    v += DTR(DT_MB(1)) * dpp0<0x102>(v);
    v += DTR(DT_MB(2)) * dpp0<0x104>(v);
    v += DTR(DT_MB(3)) * dpp0<0x108>(v);
    {
        const T s16 = readlane_d(v, 16), s48 = readlane_d(v, 48);
        v += DTR(DT_MB(4)) * sel_of(hi, s48, s16);
        if constexpr (LP == 64) v += DTR(DT_MB(5)) * readlane_d(v, 32);
    }
    carry = dpp0<0x130>(v);  // wave_shl:1 (a pencil's last lane: QB = 0)
#pragma unroll
    for (int q = 0; q < Q; q++) X[q] = (X[q] + DTR(DT_QB(q)) * carry) * DTR(DT_ST(q));
    PHASE(X[0]);
#undef DTR
#undef PHASE
}

// one operator on the window w, general form: r = its tds_solve rows (rows beyond n_tds come out as 0)
// (T = V2: two right-hand sides of the SAME operator in one solve -- every table value read from LDS serves both)
// (DIRECT: co = the lane's stencil slot, cs_slot, formed once per kernel and operator)
template <int Q, bool NARROW, class T = real_t, bool DIRECT = false>
__device__ __forceinline__ void gen_solve(const T (&w)[Q + 8], T (&r)[Q], const real_t *__restrict__ lt,
                                          const real_t *__restrict__ cs, const XOp &t, int &lane, int co = 0)
{
    if constexpr (DIRECT) {
        T xr;
        thomas_solve<Q, 64, NARROW, false, T>(w, r, xr, lt, cs, co, lane, lane >= 32);
        return;
    }
    const int first = lane * Q + 1, n = t.n_tds;
    T X[Q], du1, xn;
    scan_solve<Q, false, NARROW, T>(w, X, du1, xn, lt, t, lane, first, 0, cs);
    const T du_s = t.rs_s * (du1 - t.sa1 * xn), du_e = t.rs_e * (xn - t.scn * du1);
#ifdef XSCAN_CS_DEPTH2
    int la[2] = {lane, lane};  // two rows' table reads in flight: row q + 2's wait for row q, row q + 1's do not
    asm volatile("" : "+v"(la[1]));
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int j = first + q;
        int &lq = la[q & 1];
        const real_t st = lt_read(lt, LT_ST(q) * 64 + lq);
        T x = (X[q] - lt_read(lt, LT_SA(q) * 64 + lq) * du_s - lt_read(lt, LT_SC(q) * 64 + lq) * du_e) * st;
        if (q == 0) x = sel_of(lane == 0, du_s * st, x);  // (row 1)
        x = sel_of(j == n, du_e * st, x);
        r[q] = x;
        asm volatile("" : "+v"(lq) : "v"(first_of(x)));
    }
    asm volatile("" : "+v"(lane) : "v"(first_of(r[Q - 1])));
#else
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int j = first + q;
        const real_t st = LTR(lt, LT_ST(q));
        T x = (X[q] - LTR(lt, LT_SA(q)) * du_s - LTR(lt, LT_SC(q)) * du_e) * st;
        if (q == 0) x = sel_of(lane == 0, du_s * st, x);  // (row 1)
        x = sel_of(j == n, du_e * st, x);
        r[q] = x;
        asm volatile("" : "+v"(lane) : "v"(first_of(x)));  // (one row's table reads at a time: front-loaded, they were spilled)
    }
#endif
}

// the pieces every kernel below shares (tile <-> memory, tile <-> registers)
template <int Q> struct GenTile {
    using G = GenGeom<Q>;
    real_t *tile;
    int wave, lane, cy, cc, nrow;
    long prow;
    // rows cy + 128 i of the 16-wide segment pair cc; rows >= nrow do not exist
    __device__ __forceinline__ void gload(real_t (&v)[2 * G::NI], const real_t *__restrict__ src) const
    {
#pragma unroll
        for (int i = 0; i < G::NI; i++) {
            real2_t t = make_real2(0.0, 0.0);
            if (cy + 128 * i < nrow) t = *reinterpret_cast<const real2_t *>(src + (long)(cy + 128 * i) * prow + 2 * cc);
            v[2 * i] = t.x;
            v[2 * i + 1] = t.y;
        }
    }
    __device__ __forceinline__ void to_tile(const real_t (&v)[2 * G::NI]) const
    {
#pragma unroll
        for (int i = 0; i < G::NI; i++) {
            if (cy + 128 * i < nrow) {
                tile[(2 * cc) * G::TP + 4 + cy + 128 * i] = v[2 * i];
                tile[(2 * cc + 1) * G::TP + 4 + cy + 128 * i] = v[2 * i + 1];
            }
        }
    }
    // this lane's window: rows first-4 .. first+Q+3 = tile entries lane Q .. lane Q + Q + 7 of the wave's pencil
    __device__ __forceinline__ void window(real_t (&w)[Q + 8]) const
    {
        if constexpr (Q % 2 == 0) {
            const real2_t *__restrict__ s2 = reinterpret_cast<const real2_t *>(tile + wave * G::TP + lane * Q);
#pragma unroll
            for (int m = 0; m < (Q + 8) / 2; m++) {
                const real2_t t = s2[m];
                w[2 * m] = t.x;
                w[2 * m + 1] = t.y;
            }
        } else {  // odd Q: the window starts on an 8-byte boundary only (lane stride 40 B at Q = 5: conflict-free)
            const real_t *__restrict__ s1 = tile + wave * G::TP + lane * Q;
#pragma unroll
            for (int m = 0; m < Q + 8; m++) w[m] = lt_read(s1, m);
        }
    }
    __device__ __forceinline__ void put(const real_t (&r)[Q]) const
    {
        if constexpr (Q % 2 == 0) {
            real2_t *__restrict__ d2 = reinterpret_cast<real2_t *>(tile + wave * G::TP + 4 + lane * Q);
#pragma unroll
            for (int m = 0; m < Q / 2; m++) d2[m] = make_real2(r[2 * m], r[2 * m + 1]);
        } else {
            real_t *__restrict__ d1 = tile + wave * G::TP + 4 + lane * Q;
#pragma unroll
            for (int m = 0; m < Q; m++) d1[m] = r[m];
        }
    }
    // rows < nout of the tile -> memory; ACC: out = old + r with the old rows already in registers (gload)
    template <bool ACC>
    __device__ __forceinline__ void from_tile(real_t *__restrict__ o, int nout, const real_t (&old)[2 * G::NI]) const
    {
#pragma unroll
        for (int i = 0; i < G::NI; i++) {
            if (cy + 128 * i < nout) {
                real2_t v = make_real2(tile[(2 * cc) * G::TP + 4 + cy + 128 * i], tile[(2 * cc + 1) * G::TP + 4 + cy + 128 * i]);
                if (ACC) { v.x += old[2 * i]; v.y += old[2 * i + 1]; }
                *reinterpret_cast<real2_t *>(o + (long)(cy + 128 * i) * prow + 2 * cc) = v;
            }
        }
    }
};

// ---------------------------------------------------------------- operator pairs / single operators
//   MODE 0: out1 = A(in1) + B(in2)     MODE 1: out1 = A(in1), out2 = B(in1)     MODE 2: out1 = A(in1)
// (the pairs of divergence_v2c / gradient_c2v, src/vector_calculus.f90:142-332, as in k_ytile_tds_pair)
template <int Q, int MODE, bool NARROW, bool DIRECT = false>
__global__ void __launch_bounds__(1024)
    k_ygen_pair(real_t *out1, real_t *out2, const real_t *__restrict__ in1, const real_t *__restrict__ in2, XOp ta, XOp tb,
                int ntx, int ntiles, long prow, long pplane, int nrow, int permn)
{
    using G = GenGeom<Q>;
    extern __shared__ real_t lt[];
    constexpr int LN = (DIRECT ? DT_NC(Q) : LT_N(Q)) * 64;
    for (int i = threadIdx.x; i < LN; i += blockDim.x) {
        lt[i] = ta.TL[i];
        if (MODE != 2) lt[LN + i] = tb.TL[i];
    }
    real_t *tile = lt + (MODE == 2 ? 1 : 2) * LN;
    real_t *cs = tile + 16 * G::TP;
    for (int i = threadIdx.x; i < 16 * G::TP; i += blockDim.x) tile[i] = 0.0;
    stage_cs<Q>(cs, ta);
    if (MODE != 2) stage_cs<Q>(cs + CS_N(Q), tb);
    int lane = threadIdx.x & 63;
    const int coa = cs_slot<Q, 64>(lane, ta.n_rhs), cob = cs_slot<Q, 64>(lane, tb.n_rhs);
    GenTile<Q> T{tile, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane, (int)(threadIdx.x >> 3), (int)(threadIdx.x & 7),
                 nrow, prow};
    auto tile_off = [&](int tl) { return (long)(tl / ntx) * pplane + (long)(tl % ntx) * 16; };
    // permn > 0: y rows interleaved for the 010 Poisson solver (see k_ytile_tds_pair): MODE 0 writes, MODE 1 reads there
    auto tile_off_p = [&](int tl) {
        int r = tl / ntx;
        if (permn > 0 && r < permn) r = (r & 1) ? permn - ((r + 1) >> 1) : (r >> 1);
        return (long)r * pplane + (long)(tl % ntx) * 16;
    };
    auto in1_off = [&](int tl) { return MODE == 1 ? tile_off_p(tl) : tile_off(tl); };
    __syncthreads();
    real_t nxt[2 * G::NI];  // next tile's in1 rows, in flight during the solves
    if ((int)blockIdx.x < ntiles) T.gload(nxt, in1 + in1_off(blockIdx.x));
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = tile_off(tl);
        asm volatile("" : "+v"(lane));
        real_t w[Q + 8], ra[Q], rb[Q], g2[2 * G::NI];
        const real_t none[2 * G::NI] = {};
        if (MODE == 0) T.gload(g2, in2 + off);
        T.to_tile(nxt);
        __syncthreads();
        T.window(w);
        if (MODE == 0) {
            // (the window must BE in registers before the barrier: its loads go through __restrict__ pointers, which the
            //  compiler may otherwise sink below it -- see k_ytile_tds_pair)
#pragma unroll
            for (int m = 0; m < Q + 8; m++) asm volatile("" : "+v"(w[m]));
            __syncthreads();  // all windows read: the second input may overwrite the tile
        }
        {
            const int tn = tl + gridDim.x;
            if (tn < ntiles) T.gload(nxt, in1 + in1_off(tn));
        }
        gen_solve<Q, NARROW, real_t, DIRECT>(w, ra, lt, cs, ta, lane, coa);
        if (MODE == 0) {
            T.to_tile(g2);
            __syncthreads();
            T.window(w);
        }
        if (MODE != 2) {
            gen_solve<Q, NARROW, real_t, DIRECT>(w, rb, lt + LN, cs + CS_N(Q), tb, lane, cob);
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < Q; q++) ra[q] = ra[q] + 1.0 * rb[q];  // (the accumulating form: old + 1.0 * r)
            }
        }
        T.put(ra);  // (a wave only rewrites its own pencil's rows, which only it reads)
        __syncthreads();
        T.template from_tile<false>(out1 + (MODE == 0 ? tile_off_p(tl) : off), ta.n_tds, none);
        if (MODE == 1) {
            __syncthreads();
            T.put(rb);
            __syncthreads();
            T.template from_tile<false>(out2 + off, tb.n_tds, none);
        }
        __syncthreads();  // the tile is free again
    }
}

// ---------------------------------------------------------------- transeq: the three components of a direction
// component 0 = (u0, conv = u0), 1, 2 = (u1, u0), (u2, u0) (src/backend/omp/backend.f90:145-184); der1st and
// der1st_sym, der2nd and der2nd_sym must be equal as lane tables (Dirichlet ends: they are) -- two table sets.
// NARROW1 / NARROW: der1st's / both operators' stencils reach 2 rows at most (the Dirichlet closure of der2nd's first
// and last row reaches 3: the channel case runs <.., false, true>)
// NP: pencils (= waves) per workgroup.  8 (round 5, 257..320-row pencils): two workgroups per CU -- the lock-step phases of
// one (load, solve, store, barriers) run beside the other's; their 64-byte row segments pair up into 128-byte lines, so the
// two tiles of a pair go to workgroups of the same XCD (one L2) that run at the same time
template <int Q, bool ACC, bool NARROW, bool NARROW1 = NARROW, int NP = 16, bool DIRECT = false>
__global__ void __launch_bounds__(64 * NP, NP == 8 ? 4 : 1)
    k_ygen_transeq3(real_t *rhs0, real_t *rhs1, real_t *rhs2, const real_t *__restrict__ u0, const real_t *__restrict__ u1,
                    const real_t *__restrict__ u2, XOp tD1, XOp tD2, int ntx, int ntiles, long prow, long pplane, int nrow,
                    real_t nu)
{
    using G = GenGeom<Q>;
    // the first two solves of a component as one solve over the pair type where that fits the 128 registers
    // (Q = 8 and the wide-stencil Q = 6 form spill with it: 68 - 220 bytes of scratch)
#ifndef YGEN_NO_P12
    constexpr bool P12 = Q <= 5 || (NARROW1 && Q <= 6);
#else
    constexpr bool P12 = false;
#endif
    constexpr bool LATE = P12 && Q <= 5;  // where the rows the result is added to are requested (register budget)
    extern __shared__ real_t lt[];
    constexpr int LN = (DIRECT ? DT_N(Q) : LT_N(Q)) * 64, L1N = (DIRECT ? DT_NC(Q) : LT_NC(Q)) * 64;  // (the first operator's STC block is never read)
    constexpr int E_STC = DIRECT ? DT_STC(0) : LT_STC(0);
    for (int i = threadIdx.x; i < L1N; i += blockDim.x) lt[i] = tD1.TL[i];
    for (int i = threadIdx.x; i < LN; i += blockDim.x) lt[L1N + i] = tD2.TL[i];
    const real_t *__restrict__ l1 = lt, *__restrict__ l3 = lt + L1N;
    real_t *tile = lt + L1N + LN;
    real_t *cs = tile + NP * G::TP;
    for (int i = threadIdx.x; i < NP * G::TP; i += blockDim.x) tile[i] = 0.0;
    stage_cs<Q>(cs, tD1);
    stage_cs<Q>(cs + CS_N(Q), tD2);
    int lane = threadIdx.x & 63;
    const int co1 = cs_slot<Q, 64>(lane, tD1.n_rhs), co3 = cs_slot<Q, 64>(lane, tD2.n_rhs);
    GenTile<Q> T{tile, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane, (int)(threadIdx.x / (NP / 2)),
                 (int)(threadIdx.x % (NP / 2)), nrow, prow};
    auto tile_off = [&](int tl) { return (long)(tl / ntx) * pplane + (long)(tl % ntx) * NP; };
    // NP = 8: blocks b and b + 8 (same XCD, dispatched back to back) take the tiles 2 j and 2 j + 1
    int bid = blockIdx.x;
    if constexpr (NP == 8) {
        const int g16 = bid / 16, r16 = bid % 16;
        bid = g16 * 16 + 2 * (r16 % 8) + r16 / 8;
    }
    __syncthreads();
    real_t nxt[2 * G::NI];  // the rows needed next (next component's field, or the next tile's u0)
    if (bid < ntiles) T.gload(nxt, u0 + tile_off(bid));
    for (int tl = bid; tl < ntiles; tl += gridDim.x) {
        const long off = tile_off(tl);
        real_t cb[Q];  // this pencil's rows of the advecting velocity
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            asm volatile("" : "+v"(lane));
            real_t r[Q], X[Q];
            T.to_tile(nxt);
            __syncthreads();
            {
                const int tn = tl + gridDim.x;
                const real_t *nsrc = c == 0 ? u1 + off : (c == 1 ? u2 + off : u0 + tile_off(tn < ntiles ? tn : tl));
                if (c < 2 || tn < ntiles) T.gload(nxt, nsrc);
            }
            // the rows this component is added to: requested now, used after the three solves (the memory system
            // works during the arithmetic instead of after it)
            real_t *o = (c == 0 ? rhs0 : (c == 1 ? rhs1 : rhs2)) + off;
            real_t old[2 * G::NI] = {};
            if (ACC && !LATE) T.gload(old, o);  // (LATE: after the pair solve, whose two windows need the registers)
            if constexpr (P12) {
                // d(u conv) and du: the same operator on two right-hand sides -- ONE solve over the pair type (round 4:
                // the kernel is bound by the rate of its LDS read instructions; 245 -> 178 per lane and component)
                V2 w2[Q + 8], X2[Q];
                {
                    real_t wu[Q + 8], wc[Q + 8];
                    T.window(wu);
                    if (c == 0) {
#pragma unroll
                        for (int q = 0; q < Q; q++) cb[q] = wu[4 + q];
                    }
                    window_from_body_zero<Q>(wc, cb);
#pragma unroll
                    for (int m = 0; m < Q + 8; m++) w2[m] = V2{wu[m] * wc[m], wu[m]};
                }
                gen_solve<Q, NARROW1, V2, DIRECT>(w2, X2, l1, cs, tD1, lane, co1);
#pragma unroll
                for (int q = 0; q < Q; q++) r[q] = -0.5 * (cb[q] * X2[q].b + X2[q].a) + nu * (X2[q].b * LTR(l3, E_STC + q));
            } else {
            {
                // d(u conv): the product window; the field's own window is read again from the tile afterwards
                // (it stays there until the result is put back) instead of living through this solve
                real_t wp[Q + 8], wc[Q + 8];
                T.window(wp);
                if (c == 0) {
#pragma unroll
                    for (int q = 0; q < Q; q++) cb[q] = wp[4 + q];
                }
                window_from_body_zero<Q>(wc, cb);
#pragma unroll
                for (int m = 0; m < Q + 8; m++) wp[m] = wp[m] * wc[m];
                gen_solve<Q, NARROW1, real_t, DIRECT>(wp, X, l1, cs, tD1, lane, co1);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] = X[q];
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            {
                real_t wu[Q + 8];
                T.window(wu);
                gen_solve<Q, NARROW1, real_t, DIRECT>(wu, X, l1, cs, tD1, lane, co1);  // du
            }
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] = -0.5 * (cb[q] * X[q] + r[q]) + nu * (X[q] * LTR(l3, E_STC + q));
            }
            if (ACC && LATE) T.gload(old, o);
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            {
                real_t wu[Q + 8];  // (read again: a window is dead once its stencil sums are formed)
                T.window(wu);
                gen_solve<Q, NARROW, real_t, DIRECT>(wu, X, l3, cs + CS_N(Q), tD2, lane, co3);  // d2u
            }
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] += nu * X[q];
            T.put(r);  // (a wave only rewrites its own pencil's rows, which only it reads)
            __syncthreads();
            T.template from_tile<ACC>(o, tD1.n_tds, old);
            __syncthreads();  // the tile is free again
        }
    }
}

// ---------------------------------------------------------------- K3h: transeq on 257-row pencils, TWO pencils per wave
// The channel's wall-normal direction (BASELINE configs[4]: 257 vertices, Dirichlet, stretched) in the DIRECT form with a
// pencil per HALF-wave: lanes 0..31 / 32..63 own the pencils 2 w / 2 w + 1 of the tile, 8 rows per lane + row 257 on the
// pencil's last lane (thomas_solve<8, 32, .., XR>).  Against k_ygen_transeq3<5, .., DIRECT> (one pencil per wave, 5 rows per
// lane = 320 row slots for 257 rows): no idle row slots, and every wave-wide cost -- the scans' DPP steps, every lane-table
// and stencil read -- serves two pencils.  A workgroup = 8 waves = the 16 x-adjacent pencils of one plane (128-byte row
// segments), two workgroups per CU (76 KB of LDS each: the lane tables staged [entry][32]).
//  * tile: [16 pencils][YH_TP]; row j of a pencil at entry yh_sw(j + 3) -- 4 zero rows before row 1, zeros after row 257,
//    and TWO pad entries after every 32: a lane's window (entries 8 l .. 8 l + 15) is read as 8 ds_read_b128 whose 16-lane
//    groups would otherwise hit 4 banks-quads out of 16 (lane stride 64 B); with the pads they cover all 16
//  * the window of the advecting velocity (components 1, 2: the tile holds u1 / u2) is rebuilt from the lane's own rows
//    by DPP; at the seam between the two pencils a lane picks up the OTHER pencil's rows -- where the stencils of its rows
//    have zero weight (rows before row 1, rows after row 257), as with the zero rows of the tile -- except row 257 itself
//    in the right halo of the pencil's last lane, which that lane keeps in a register
constexpr int YH_Q = 8, YH_NP = 16, YH_TP = 282, YH_NI = 5, YH_CSN = CS_N(YH_Q) + 10;
__device__ __forceinline__ int yh_sw(int t) { return t + 2 * (t >> 5); }
template <bool ACC, bool NARROW, bool NARROW1>
__global__ void __launch_bounds__(512, 4)
    k_yhalf_transeq3(real_t *rhs0, real_t *rhs1, real_t *rhs2, const real_t *__restrict__ u0, const real_t *__restrict__ u1,
                     const real_t *__restrict__ u2, XOp tD1, XOp tD2, int ntx, int ntiles, long prow, long pplane, real_t nu)
{
    constexpr int Q = YH_Q, NROW = 257;
    extern __shared__ real_t lt[];
    constexpr int L1N = DT_NC(Q) * 32, L3N = DT_N(Q) * 32;
    for (int i = threadIdx.x; i < L1N; i += blockDim.x) lt[i] = tD1.TL[(i >> 5) * 64 + (i & 31)];
    for (int i = threadIdx.x; i < L3N; i += blockDim.x) lt[L1N + i] = tD2.TL[(i >> 5) * 64 + (i & 31)];
    const real_t *__restrict__ l1 = lt, *__restrict__ l3 = lt + L1N;
    real_t *tile = lt + L1N + L3N;
    real_t *cs = tile + YH_NP * YH_TP;
    for (int i = threadIdx.x; i < YH_NP * YH_TP; i += blockDim.x) tile[i] = 0.0;
    stage_cs<Q>(cs, tD1);
    stage_cs<Q>(cs + YH_CSN, tD2);
    if (threadIdx.x < 20) {  // row 257's stencil (the last end row), per operator: [9] + a pad
        const int o = threadIdx.x / 10, m = threadIdx.x % 10;
        cs[o * YH_CSN + CS_N(Q) + m] = m < 9 ? (o ? tD2 : tD1).Cs[63 + m] : 0.0;
    }
    const int lane = threadIdx.x & 63;
    int ll = lane & 31;
    const bool hi = lane >= 32;
    const int pen = 2 * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + (hi ? 1 : 0);
    // the lane's window = entries 8 l .. 8 l + 7 (piece A) and 8 l + 8 .. 8 l + 15 (piece B); its rows = entries 8 l + 4 .. + 11
    real_t *pA = tile + pen * YH_TP + yh_sw(8 * ll), *pB = tile + pen * YH_TP + yh_sw(8 * ll + 8);
    const int co1 = cs_slot<Q, 32>(lane, NROW), co3 = co1;
    const int cy = threadIdx.x >> 3, cc = threadIdx.x & 7;
    const unsigned voff = (unsigned)(((long)cy * prow + 2 * cc) * X3D_RB);
    real_t *tA = tile + (2 * cc) * YH_TP, *tB = tA + YH_TP;
    auto row_ptr = [&](const real_t *base, int i) {
        return reinterpret_cast<const real2_t *>(reinterpret_cast<const char *>(base + (long)(64 * i) * prow) + voff);
    };
    auto gload = [&](real2_t (&v)[YH_NI], const real_t *__restrict__ src) {
#pragma unroll
        for (int i = 0; i < YH_NI; i++) {
            v[i] = make_real2(0.0, 0.0);
            if (i < 4 || cy == 0) v[i] = *row_ptr(src, i);  // rows cy + 64 i < 257
        }
    };
    auto to_tile = [&](const real2_t (&v)[YH_NI]) {
#pragma unroll
        for (int i = 0; i < YH_NI; i++) {
            if (i < 4 || cy == 0) {
                const int e = yh_sw(4 + cy + 64 * i);
                tA[e] = v[i].x;
                tB[e] = v[i].y;
            }
        }
    };
    auto window = [&](real_t (&w)[Q + 8]) {
        const real2_t *__restrict__ a = reinterpret_cast<const real2_t *>(pA), *__restrict__ b_ = reinterpret_cast<const real2_t *>(pB);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const real2_t x = a[m], y = b_[m];
            w[2 * m] = x.x; w[2 * m + 1] = x.y;
            w[8 + 2 * m] = y.x; w[9 + 2 * m] = y.y;
        }
    };
    auto tile_off = [&](int tl) { return (long)(tl / ntx) * pplane + (long)(tl % ntx) * YH_NP; };
    __syncthreads();
    real2_t nxt[YH_NI];
    if ((int)blockIdx.x < ntiles) gload(nxt, u0 + tile_off(blockIdx.x));
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = tile_off(tl);
        real_t cb[Q], cbx = 0.0;  // this pencil's rows of the advecting velocity (cbx: row 257, the pencil's last lane)
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            asm volatile("" : "+v"(ll));
            to_tile(nxt);
            __syncthreads();
            {
                const int tn = tl + gridDim.x;
                const real_t *nsrc = c == 0 ? u1 + off : (c == 1 ? u2 + off : u0 + tile_off(tn < ntiles ? tn : tl));
                if (c < 2 || tn < ntiles) gload(nxt, nsrc);
            }
            real_t r[Q], rx, X[Q], xr;
            {
                real_t wp[Q + 8];
                window(wp);
                if (c == 0) {
#pragma unroll
                    for (int q = 0; q < Q; q++) cb[q] = wp[4 + q];
                    cbx = wp[Q + 4];
#pragma unroll
                    for (int m = 0; m < Q + 8; m++) wp[m] = wp[m] * wp[m];
                } else {
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        wp[m] = wp[m] * dpp0<0x138>(cb[Q - 4 + m]);  // wave_shr:1
                        real_t rh = dpp0<0x130>(cb[m]);               // wave_shl:1
                        if (m == 0) rh = ll == 31 ? cbx : rh;
                        wp[Q + 4 + m] = wp[Q + 4 + m] * rh;
                    }
#pragma unroll
                    for (int q = 0; q < Q; q++) wp[4 + q] = wp[4 + q] * cb[q];
                }
                thomas_solve<Q, 32, NARROW1, true, real_t, NARROW1>(wp, X, xr, l1, cs, co1, ll, hi);  // d(u conv)
            }
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] = X[q];
            rx = xr;
            asm volatile("" : "+v"(ll) : "v"(r[0]));
            {
                real_t wu[Q + 8];
                window(wu);
                thomas_solve<Q, 32, NARROW1, true, real_t, NARROW1>(wu, X, xr, l1, cs, co1, ll, hi);  // du
            }
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] = -0.5 * (cb[q] * X[q] + r[q]) + nu * (X[q] * lt_read(l3, DT_STC(q) * 32 + ll));
            rx = -0.5 * (cbx * xr + rx) + nu * (xr * lt_read(l3, DT_X(3) * 32 + ll));
            real_t *o = (c == 0 ? rhs0 : (c == 1 ? rhs1 : rhs2)) + off;
            real2_t old[YH_NI];
            asm volatile("" : "+v"(ll) : "v"(r[0]));
            {
                real_t wu[Q + 8];
                window(wu);
                thomas_solve<Q, 32, NARROW, true, real_t, NARROW>(wu, X, xr, l3, cs + YH_CSN, co3, ll, hi);  // d2u
            }
            {
                real2_t *__restrict__ a = reinterpret_cast<real2_t *>(pA), *__restrict__ b_ = reinterpret_cast<real2_t *>(pB);
                a[2] = make_real2(r[0] + nu * X[0], r[1] + nu * X[1]);
                a[3] = make_real2(r[2] + nu * X[2], r[3] + nu * X[3]);
                b_[0] = make_real2(r[4] + nu * X[4], r[5] + nu * X[5]);
                b_[1] = make_real2(r[6] + nu * X[6], r[7] + nu * X[7]);
                if (ll == 31) pB[4] = rx + nu * xr;  // row 257
            }
            if constexpr (ACC) {
                // the rows the result is added to: requested when the solves' registers are free (any earlier and the 20
                // registers they land in spill: 93 VGPRs to scratch); the other workgroup of the CU covers the wait
                const real_t *oo = o;
                asm volatile("" : "+s"(oo) : "v"(xr));
                gload(old, oo);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < YH_NI; i++) {
                if (i < 4 || cy == 0) {
                    const int e = yh_sw(4 + cy + 64 * i);
                    real2_t v = make_real2(tA[e], tB[e]);
                    if (ACC) { v.x += old[i].x; v.y += old[i].y; }
                    *const_cast<real2_t *>(row_ptr(o, i)) = v;
                }
            }
            __syncthreads();  // the tile is free again
        }
    }
}

// ---------------------------------------------------------------- launchers
static bool gen_env_on()
{
    static int on = -1;
    if (on < 0) {
        const char *names[4] = {"X3D_NO_YGEN", "X3D_NO_YTILE", "X3D_NO_XSCAN", "X3D_NO_VIA_X"};
        on = 1;
        for (const char *nm : names) { const char *e = getenv(nm); if (e && e[0] == '1') on = 0; }
    }
    return on == 1;
}
static bool gen_ok(const x3d_backend *b, int dir, const x3d_tdsops *t, int Q)
{
    const int n = dir == X3D_DIR_Y ? b->ny : b->nz;
    const bool tables = Q == 5 ? t->tl5 != nullptr : (t->tab.TL != nullptr && t->tab.Q == Q && (Q == 4 || Q == 6 || Q == 8));
    // (periodic-type operators need the periodic image / the neighbours' rows in the halo rows: K3y's business)
    return tables && n <= 64 * Q && t->tab.n_rhs <= n && b->nx % 16 == 0 && !t->periodic && !t->tab.bulk_only;
}
// rows per lane for a launch on these operators: 5 where every one of them carries the 5-row tables (257..320-row
// pencils: 52 of 64 lanes busy instead of 43), else the operators' own Q.  X3D_NO_Q5=1: always the latter (A/B)
static int gen_q(const x3d_backend *b, int dir, const x3d_tdsops *const *ops, int nops)
{
    static int q5 = -1;
    if (q5 < 0) { const char *e = getenv("X3D_NO_Q5"); q5 = (e && e[0] == '1') ? 0 : 1; }
    const int n = dir == X3D_DIR_Y ? b->ny : b->nz;
    bool all5 = q5 && n <= 320;
    for (int k = 0; k < nops; k++) all5 = all5 && ops[k]->tl5 != nullptr;
    return all5 ? 5 : ops[0]->tab.Q;
}
static XOp gen_xop(const x3d_tdsops *t, int Q, bool direct = false)
{
    XOp o = xop_of(t);
    if (Q == 5) o.TL = direct ? t->td5 : t->tl5;
    return o;
}
// the DIRECT form (thomas_solve) where every operator of the launch offers its tables.  X3D_NO_DIRECT=1: never (A/B)
static bool gen_direct(const x3d_tdsops *const *ops, int nops, int Q)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_DIRECT"); on = (e && e[0] == '1') ? 0 : 1; }
    bool d = on && Q == 5;
    for (int k = 0; k < nops; k++) d = d && ops[k]->direct && ops[k]->td5 != nullptr;
    return d;
}
struct GenLaunch {
    int ntx, ntiles, blocks, nrow;
    long rstride, ostride;
};
static GenLaunch gen_launch(const x3d_backend *b, int dir)
{
    const long pxy = (long)b->nxp * b->nyp;
    GenLaunch g;
    g.ntx = b->nx / 16;
    g.ntiles = g.ntx * (dir == X3D_DIR_Y ? b->nz : b->ny);
    g.blocks = x3d_persistent_blocks(b, g.ntiles);
    g.nrow = dir == X3D_DIR_Y ? b->ny : b->nz;
    g.rstride = dir == X3D_DIR_Y ? (long)b->nxp : pxy;
    g.ostride = dir == X3D_DIR_Y ? pxy : (long)b->nxp;
    return g;
}

// operator pair (mode 0 / 1) or single operator (mode 2) along y or z; *done = false: not served here
int x3d_ygen_pair(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                  const x3d_tdsops *ta, const x3d_tdsops *tb, bool *done)
{
    *done = false;
    if (!gen_env_on() || dir == X3D_DIR_X) return 0;
    const x3d_tdsops *const ops[2] = {ta, mode == 2 ? ta : tb};
    const int Q = gen_q(b, dir, ops, 2);
    if (!gen_ok(b, dir, ta, Q) || (mode != 2 && !gen_ok(b, dir, tb, Q))) return 0;
    const bool direct = gen_direct(ops, 2, Q);
    const size_t lds = sizeof(real_t) * ((size_t)(mode == 2 ? 1 : 2) * (direct ? DT_NC(Q) : LT_N(Q)) * 64 + 16 * (64 * Q + 10) + 2 * CS_N(Q));
    if (lds > 160 * 1024) return 0;
    const GenLaunch g = gen_launch(b, dir);
    const x3d_tdsops *tb_ = mode == 2 ? ta : tb;
    const bool narrow = ta->narrow_all && tb_->narrow_all;
    const int permn = b->pair_yperm;
    if (permn > 0 && (dir != X3D_DIR_Z || mode == 2)) return 0;
    {
    ProfScope ps(b, X3D_K_TDS_FWD, dir);
#define GO(Q_, M_, N_, D_)                                                                                      \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_ygen_pair<Q_, M_, N_, D_>));                                                        \
        hipLaunchKernelGGL((k_ygen_pair<Q_, M_, N_, D_>), dim3(g.blocks), dim3(1024), lds, b->stream, out1, out2, in1, in2, \
                           gen_xop(ta, Q_, D_), gen_xop(tb_, Q_, D_), g.ntx, g.ntiles, g.rstride, g.ostride, g.nrow, permn); \
    } while (0)
#define GON(Q_, M_, D_) do { if (narrow) GO(Q_, M_, true, D_); else GO(Q_, M_, false, D_); } while (0)
#define GOM(Q_, D_) do { if (mode == 0) GON(Q_, 0, D_); else if (mode == 1) GON(Q_, 1, D_); else GON(Q_, 2, D_); } while (0)
    if (Q == 8) GOM(8, false); else if (Q == 6) GOM(6, false); else if (Q == 5) { if (direct) GOM(5, true); else GOM(5, false); } else GOM(4, false);
#undef GOM
#undef GON
#undef GO
    }
    X3D_HIP(hipGetLastError());
    if (b->prof && mode != 2) { ProfScope ps2(b, X3D_K_TDS_FWD, dir); }  // two operators
    *done = true;
    return 0;
}

// transeq of direction y or z in one launch; f[0] is the advecting component
int x3d_ygen_transeq3(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                      const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                      const x3d_tdsops *der2nd_sym, int acc, bool *done)
{
    *done = false;
    if (!gen_env_on() || dir == X3D_DIR_X) return 0;
    const x3d_tdsops *const ops[4] = {der1st, der1st_sym, der2nd, der2nd_sym};
    const int Q = gen_q(b, dir, ops, 4);
    if (!gen_ok(b, dir, der1st, Q) || !gen_ok(b, dir, der1st_sym, Q) || !gen_ok(b, dir, der2nd, Q) ||
        !gen_ok(b, dir, der2nd_sym, Q))
        return 0;
    if (der1st->tl_hash != der1st_sym->tl_hash || der2nd->tl_hash != der2nd_sym->tl_hash) return 0;
    if (der1st->n_tds != der2nd->n_tds) return 0;
    // (tl_hash covers the lane tables and the boundary / bulk stencils: tds.hip)
    // 257 rows, DIRECT, every operator with the half-wave tables: two pencils per wave (K3h).  Measured SLOWER than the
    // one-pencil-per-wave DIRECT form (2.60 against 2.44 ms per launch at 1024 x 257 x 512: 30 % fewer vector instructions,
    // but three serial 8-row solves per component where that form runs a pair solve of two interleaved 5-row chains and
    // one more -- the kernel is bound by its dependency chains at 4 waves per SIMD): only with X3D_YHALF=1
    {
        static int yh = -1;
        if (yh < 0) { const char *e = getenv("X3D_YHALF"); yh = (e && e[0] == '1') ? 1 : 0; }
        const int n = dir == X3D_DIR_Y ? b->ny : b->nz;
        bool ok = yh && n == 257 && der1st->n_tds == 257 && der1st->tab.n_rhs == 257 && der2nd->tab.n_rhs == 257 && gen_direct(ops, 4, 5);
        for (int k = 0; k < 4; k++) ok = ok && ops[k]->td8h != nullptr;
        if (ok) {
            const size_t lds = sizeof(real_t) * ((size_t)(DT_NC(YH_Q) + DT_N(YH_Q)) * 32 + YH_NP * YH_TP + 2 * YH_CSN);
            GenLaunch g = gen_launch(b, dir);
            const long cap = 2 * (long)x3d_persistent_blocks(b, X3D_NCU);
            g.blocks = (int)(g.ntiles > cap ? cap : g.ntiles);
            XOp o1 = xop_of(der1st), o3 = xop_of(der2nd);
            o1.TL = der1st->td8h;
            o3.TL = der2nd->td8h;
            const bool narrow1 = der1st->narrow_all, narrow = narrow1 && der2nd->narrow_all;
            {
                ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
#define GOH(A_, N_, N1_)                                                                                                    \
    do {                                                                                                                    \
        X3D_LDS_OPTIN(b, (k_yhalf_transeq3<A_, N_, N1_>));                                                                  \
        hipLaunchKernelGGL((k_yhalf_transeq3<A_, N_, N1_>), dim3(g.blocks), dim3(512), lds, b->stream, r[0], r[1], r[2], f[0], f[1], \
                           f[2], o1, o3, g.ntx, g.ntiles, g.rstride, g.ostride, nu);                                        \
    } while (0)
#define GOHN(A_) do { if (narrow) GOH(A_, true, true); else if (narrow1) GOH(A_, false, true); else GOH(A_, false, false); } while (0)
                if (acc) GOHN(true); else GOHN(false);
#undef GOHN
#undef GOH
            }
            X3D_HIP(hipGetLastError());
            b->n_tq3++;
            if (b->prof) {
                for (int k = 0; k < 2; k++) { ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir); }
            }
            *done = true;
            return 0;
        }
    }
    // 8 pencils per workgroup, two workgroups per CU: 257..320-row pencils (Q = 5), nx a multiple of 32 (tile pairs share
    // 128-byte lines and must not straddle rows of tiles).  X3D_YGEN_NP16=1: the 16-pencil form (A/B)
    static int np16 = -1;
    if (np16 < 0) { const char *e = getenv("X3D_YGEN_NP16"); np16 = (e && e[0] == '1') ? 1 : 0; }
    const int NP = (Q == 5 && !np16 && b->nx % 32 == 0) ? 8 : 16;
    const bool direct = gen_direct(ops, 4, Q);
    const size_t lds = sizeof(real_t) * ((size_t)(direct ? DT_NC(Q) + DT_N(Q) : LT_NC(Q) + LT_N(Q)) * 64 + NP * (64 * Q + 10) + 2 * CS_N(Q));
    if (lds > 160 * 1024) return 0;
    GenLaunch g = gen_launch(b, dir);
    if (NP == 8) {
        g.ntx = b->nx / 8;
        g.ntiles = g.ntx * (dir == X3D_DIR_Y ? b->nz : b->ny);
        const long cap = 2 * (long)x3d_persistent_blocks(b, X3D_NCU);
        g.blocks = (int)(g.ntiles > cap ? cap : g.ntiles);
        g.blocks -= g.blocks % 16;  // (the pairing of blocks b and b + 8)
        if (g.blocks < 16) return 0;
    }
    const bool narrow1 = der1st->narrow_all, narrow = narrow1 && der2nd->narrow_all;
    {
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
#define GO(Q_, A_, N_, N1_, P_, D_)                                                                             \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_ygen_transeq3<Q_, A_, N_, N1_, P_, D_>));                                           \
        hipLaunchKernelGGL((k_ygen_transeq3<Q_, A_, N_, N1_, P_, D_>), dim3(g.blocks), dim3(64 * P_), lds, b->stream, r[0], r[1], r[2], f[0], \
                           f[1], f[2], gen_xop(der1st, Q_, D_), gen_xop(der2nd, Q_, D_), g.ntx, g.ntiles, g.rstride, g.ostride, g.nrow, nu); \
    } while (0)
#define GON(Q_, A_, P_, D_) do { if (narrow) GO(Q_, A_, true, true, P_, D_); else if (narrow1) GO(Q_, A_, false, true, P_, D_); else GO(Q_, A_, false, false, P_, D_); } while (0)
#define GOA(Q_, P_, D_) do { if (acc) GON(Q_, true, P_, D_); else GON(Q_, false, P_, D_); } while (0)
        if (Q == 8) GOA(8, 16, false); else if (Q == 6) GOA(6, 16, false);
        else if (Q == 5) { if (NP == 8) { if (direct) GOA(5, 8, true); else GOA(5, 8, false); } else { if (direct) GOA(5, 16, true); else GOA(5, 16, false); } }
        else GOA(4, 16, false);
#undef GOA
#undef GON
#undef GO
    }
    X3D_HIP(hipGetLastError());
    b->n_tq3++;
    if (b->prof) {
        for (int k = 0; k < 2; k++) { ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir); }
    }
    *done = true;
    return 0;
}
