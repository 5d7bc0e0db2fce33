// Shared device code of the wave-per-pencil scan kernels (xscan.hip: 256 / 512-row pencils, K3s / K3y;
// xwide.hip: 1024-row pencils): lane-table layout, the DPP scans and scan_solve.
#pragma once
#include "common.h"

#ifndef XSCAN_EXP
#define XSCAN_EXP 0  // timing experiments only (1: no stores, 2: no loads, 3: no scans)
#endif
// general form: TWO rows' stencil weights / table entries in flight instead of one (row q + 2's LDS reads wait for
// row q's result, row q + 1's do not): +11 VGPRs, K3g's transeq_y 1.42 -> 1.37 ms per component at 257 rows.
// -DXSCAN_CS_DEPTH1: one row at a time (A/B)
#if !defined(XSCAN_CS_DEPTH1) && !defined(XSCAN_CS_DEPTH2)
#define XSCAN_CS_DEPTH2
#endif

// lane-table entry indices (per operator): 8*Q row entries then the scan multipliers
#define LT_F(q) (0 * Q + (q))
#define LT_A(q) (1 * Q + (q))
#define LT_PF(q) (2 * Q + (q))
#define LT_H(q) (3 * Q + (q))
#define LT_QB(q) (4 * Q + (q))
#define LT_SA(q) (5 * Q + (q))
#define LT_SC(q) (6 * Q + (q))
#define LT_ST(q) (7 * Q + (q))
#define LT_MF(k) (8 * Q + (k))
#define LT_MB(k) (8 * Q + 6 + (k))
#define LT_STC(q) (8 * Q + 12 + (q))  // last: only the d2u operator of a transeq component reads it
#define LT_N(Q_) (9 * (Q_) + 12)
#define LT_NC(Q_) (8 * (Q_) + 12)     // entries without the STC block

// One lane-table value.  Two of these reads off one base register are merged by the compiler into
// ds_read2st64_b64, which the LDS serves at HALF the rate of two ds_read_b64 (measured, scratch/ldsbench.hip:
// 4.3 against 2.6 LDS clocks per 512-byte row; MI355X_MICROARCH.md, LDS table) -- and these kernels are
// LDS-bound on exactly these reads (LdsUtil 83 % in k_ytile_transeq3).  The empty asm is a barrier for the
// load / store combiner only (it ends a merge window); it emits nothing.
#if XSCAN_EXP == 4  // timing experiment: no lane-table reads at all (results are garbage): what the LDS reads cost
__device__ __forceinline__ real_t lt_read(const real_t *__restrict__ l, int idx) { return 0.37 + 1e-3 * (idx & 7); }
#elif !defined(XS_READ2)
__device__ __forceinline__ real_t lt_read(const real_t *__restrict__ l, int idx)
{
    const real_t v = l[idx];
    asm volatile("" ::: "memory");
    return v;
}
#else
__device__ __forceinline__ real_t lt_read(const real_t *__restrict__ l, int idx) { return l[idx]; }
#endif
#define LTR(l, e) lt_read((l), (e) * 64 + lane)

// Compressed lane tables (LS = LTC_LS instead of 64; xwide.hip, 1024-row pencils: Q = 16 would need 80 KB per
// operator in the full form).  The row entries of a periodic-type operator on a uniform grid are bitwise constant
// along the pencil except near its two ends (the elimination factors converge geometrically, the same decay the
// reference's truncation to a 2 x 2 system relies on, src/tdsops.f90:196-201), so an entry keeps lanes 0..7, one
// value for lanes 8..55 and lanes 56..63 (tds.hip checks that they are constant, bit by bit); the 12 scan
// multipliers depend on the lane's position in its row of 16 and stay full.  Layout: [9 Q row entries][LTC_LS]
// (F A PF H QB SA SC ST STC), then [12][64].  Lanes 8..55 read one address: an LDS broadcast.
#define LTC_LS 17
#define LTC_STC(q) (8 * Q + (q))
#define LTC_M0(Q_) (9 * (Q_) * LTC_LS)
#define LTC_N(Q_) (LTC_M0(Q_) + 12 * 64)  // doubles
__device__ __forceinline__ int ltc_lane(int lane) { return lane < 8 ? lane : (lane > 55 ? lane - 47 : 8); }
// inside templates with LS and ll in scope: one row entry / one scan multiplier of either layout
#define LTX(l, e) lt_read((l), (e) * LS + (LS == 64 ? lane : ll))
// (MOFF >= 0: the compressed layout WITHOUT its ST / STC blocks -- the multipliers start at MOFF, xwide.hip's uniform-grid form)
#define LTM(l, k) lt_read((l), (LS == 64 ? (8 * Q + (k)) * 64 : (MOFF >= 0 ? MOFF : LTC_M0(Q)) + (k) * 64) + lane)

// what the kernels need of one operator: 17 SGPRs instead of the whole TdsTab
struct XOp {
    const real_t *TL, *Cs;
    real_t last_r, rs_s, rs_e, sa1, scn;
    real_t c[9];  // bulk stencil by value: kernel arguments live in SGPRs (a load through Cs would be a
                  // VMEM load per pencil and operator, the stores may alias it)
    int n_tds, n_rhs, bulk_only;
};
static inline XOp xop_of(const x3d_tdsops *t)
{
    const TdsTab &b = t->tab;
    XOp o{b.TL, b.Cs, b.last_r, b.rs_s, b.rs_e, b.sa1, b.scn, {0}, b.n_tds, b.n_rhs, b.bulk_only};
    for (int m = 0; m < 9; m++) o.c[m] = t->coeffs[m];
    return o;
}

// scan_solve's general form: the stencil weights of every row position of lane 0, lanes ls and ls + 1 (which hold
// the boundary rows 1..4 and n_rhs-3..n_rhs; Cs = [4][9] start rows, [4][9] end rows, [9] bulk) and the bulk
// stencil, as [4 slots][Q][10] doubles in LDS (16-byte aligned)
#define CS_N(Q_) (4 * (Q_) * 10)
template <int Q>
__device__ __forceinline__ void stage_cs(real_t *dst, const XOp &t)
{
    const int nr = t.n_rhs, ls = (nr - 4) / Q;
    for (int i = threadIdx.x; i < CS_N(Q); i += blockDim.x) {
        const int s_ = i / (Q * 10), q = (i / 10) % Q, m = i % 10;
        const int L = s_ == 0 ? 0 : (s_ == 1 ? ls : ls + 1), j = L * Q + q + 1;
        real_t v = 0.0;
        if (m < 9) {
            if (s_ < 3 && j <= 4) v = t.Cs[(j - 1) * 9 + m];
            else if (s_ < 3 && j > nr - 4 && j <= nr) v = t.Cs[36 + (j - (nr - 4) - 1) * 9 + m];
            else v = t.Cs[72 + m];
        }
        dst[i] = v;
    }
}

// DPP move of a real_t; lanes whose source lane does not exist (or whose row is masked out) read 0
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ real_t dpp0(real_t v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (all rows enabled: bound_ctrl:1 makes the lanes without a source read 0 by itself -- no zero-initialised
    // destination, two v_mov_b32 less per move; a partial row mask needs the zeros for the disabled rows)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, ROWMASK == 0xf);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, ROWMASK == 0xf);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ real_t readlane_d(real_t v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
// ---- two pencils per wave: every lane-table value read from LDS serves both (the kernels are LDS-pipe
// bound on the table reads), and the two independent dependency chains interleave.  The solver below is
// written once over T = real_t or V2.
#ifdef X3D_SINGLE_PREC
// FP32 (round 6): the pair IS a packed register pair -- gfx950 runs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at the rate
// of their scalar forms, so a pair solve costs the vector instructions of ONE solve (the FP32 flavour moves half the bytes
// with the FP64 flavour's instruction count otherwise: it is bound by instruction issue, not by HBM).  Same operations per
// component, same rounding (a packed FMA is two FMAs).  X3D_SP_NO_PK: the two-scalars form (A/B)
#ifndef X3D_SP_NO_PK
#define X3D_V2_PACKED 1
#endif
#endif
#ifdef X3D_V2_PACKED
typedef float x3d_f2 __attribute__((ext_vector_type(2)));
union V2 {
    struct { real_t a, b; };
    x3d_f2 v;
};
__device__ __forceinline__ V2 v2_of(x3d_f2 v) { V2 r; r.v = v; return r; }
__device__ __forceinline__ V2 operator+(V2 x, V2 y) { return v2_of(x.v + y.v); }
__device__ __forceinline__ V2 operator-(V2 x, V2 y) { return v2_of(x.v - y.v); }
__device__ __forceinline__ V2 operator*(V2 x, V2 y) { return v2_of(x.v * y.v); }
__device__ __forceinline__ V2 operator*(real_t c, V2 x) { return v2_of(c * x.v); }
__device__ __forceinline__ V2 operator*(V2 x, real_t c) { return v2_of(x.v * c); }
__device__ __forceinline__ V2 &operator+=(V2 &x, V2 y) { x.v += y.v; return x; }
#else
struct V2 { real_t a, b; };
__device__ __forceinline__ V2 operator+(V2 x, V2 y) { return V2{x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ V2 operator-(V2 x, V2 y) { return V2{x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ V2 operator*(V2 x, V2 y) { return V2{x.a * y.a, x.b * y.b}; }
__device__ __forceinline__ V2 operator*(real_t c, V2 x) { return V2{c * x.a, c * x.b}; }
__device__ __forceinline__ V2 operator*(V2 x, real_t c) { return V2{x.a * c, x.b * c}; }
__device__ __forceinline__ V2 &operator+=(V2 &x, V2 y) { x.a += y.a; x.b += y.b; return x; }
#endif
// c * x + y with ONE rounding, spelled out: with -ffp-contract=fast the compiler otherwise picks the fused products per
// kernel body, and two instantiations of the same solve may differ in the last bit (circ_solve: the UPD form of transeq_x
// against k_xscan_tds<ACC> + the plain form must agree bit for bit)
__device__ __forceinline__ real_t fma_of(real_t c, real_t x, real_t y) { return fma_r(c, x, y); }
#ifdef X3D_V2_PACKED
__device__ __forceinline__ V2 fma_of(real_t c, V2 x, V2 y) { const x3d_f2 cc = {c, c}; return v2_of(__builtin_elementwise_fma(cc, x.v, y.v)); }
#else
__device__ __forceinline__ V2 fma_of(real_t c, V2 x, V2 y) { return V2{fma_r(c, x.a, y.a), fma_r(c, x.b, y.b)}; }
#endif
#ifdef X3D_V2_PACKED
__device__ __forceinline__ V2 fma_of(V2 c, V2 x, V2 y) { return v2_of(__builtin_elementwise_fma(c.v, x.v, y.v)); }
#else
__device__ __forceinline__ V2 fma_of(V2 c, V2 x, V2 y) { return V2{fma_r(c.a, x.a, y.a), fma_r(c.b, x.b, y.b)}; }
#endif
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ V2 dpp0(V2 v) { return V2{dpp0<CTRL, ROWMASK>(v.a), dpp0<CTRL, ROWMASK>(v.b)}; }
__device__ __forceinline__ V2 readlane_d(V2 v, int l) { return V2{readlane_d(v.a, l), readlane_d(v.b, l)}; }
template <class T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ real_t zero_of<real_t>() { return 0.0; }
template <> __device__ __forceinline__ V2 zero_of<V2>() { return V2{0.0, 0.0}; }
__device__ __forceinline__ real_t first_of(real_t x) { return x; }
__device__ __forceinline__ real_t first_of(V2 x) { return x.a; }
// c ? x : y, component by component (a ternary on the struct itself goes through a stack slot)
__device__ __forceinline__ real_t sel_of(bool c, real_t x, real_t y) { return c ? x : y; }
__device__ __forceinline__ V2 sel_of(bool c, V2 x, V2 y) { return V2{c ? x.a : y.a, c ? x.b : y.b}; }

__device__ __forceinline__ real_t shfl_up_d(real_t v, int d, int lane)
{
    const real_t r = __shfl_up(v, d, 64);
    return lane >= d ? r : 0.0;
}
__device__ __forceinline__ real_t shfl_down_d(real_t v, int d, int lane)
{
    const real_t r = __shfl_down(v, d, 64);
    return lane + d < 64 ? r : 0.0;
}

// extended pencil row jj in [-3, nr+4] (non-decomposed direction: periodic image,
// src/backend/omp/sendrecv.f90:20-22); rows beyond nr+4 read as zero
__device__ __forceinline__ real_t ext_x(const real_t *__restrict__ row, int jj, int nr, int n_wrap)
{
    if (jj < 1) return row[n_wrap + jj - 1];
    if (jj > nr) return jj <= nr + 4 ? row[jj - nr - 1] : 0.0;
    return row[jj - 1];
}

// one operator, lane-local + scan: in: w[Q+8] = rows first-4 .. last+4; out: X[Q] back-substituted
// values (before the reduced-system substitution), du1 and xn broadcast to all lanes.
// General form (FAST = false): cs = the operator's stencil table in LDS (stage_cs), n_rhs >= 8.
template <int Q, bool FAST, bool NARROW = false, class T = real_t, int LS = 64, int MOFF = -1>
__device__ __forceinline__ void scan_solve(const T (&w)[Q + 8], T (&X)[Q], T &du1, T &xn,
                                           const real_t *__restrict__ lt, const XOp &t, int &lane, int first, int ll = 0,
                                           const real_t *__restrict__ cs = nullptr)
{
    // PHASE(x): the lane-table reads of the next phase may not be issued before x is known; without
    // it the scheduler front-loads all ~76 reads of an operator (152 VGPRs) and spills
#define PHASE(x) asm volatile("" : "+v"(lane) : "v"(first_of(x)))
    const int nr = t.n_rhs, n = t.n_tds;
    const real_t c0 = t.c[0], c1 = t.c[1], c2 = t.c[2], c3 = t.c[3], c4 = t.c[4], c5 = t.c[5], c6 = t.c[6],
                 c7 = t.c[7], c8 = t.c[8];
    T acc[Q];
    // NARROW: the compact6 / classic stencils only reach 2 rows: skip the zero taps (adding 0 * w is exact,
    // so both forms give the same bits); chosen by the launcher from the operators' coefficients
    if constexpr (!FAST) {
    } else if (NARROW) {
#pragma unroll
        for (int q = 0; q < Q; q++)
            acc[q] = c2 * w[q + 2] + c3 * w[q + 3] + c4 * w[q + 4] + c5 * w[q + 5] + c6 * w[q + 6];
    } else {
#pragma unroll
        for (int q = 0; q < Q; q++)
            acc[q] = c0 * w[q] + c1 * w[q + 1] + c2 * w[q + 2] + c3 * w[q + 3] + c4 * w[q + 4] + c5 * w[q + 5] +
                     c6 * w[q + 6] + c7 * w[q + 7] + c8 * w[q + 8];
    }
    // General form: rows 1..4 and n_rhs-3..n_rhs use their own stencils.  They sit in lane 0 and in lanes ls, ls + 1
    // (ls = (n_rhs - 4) / Q); stage_cs builds a table [4 slots][Q][10] in LDS -- the nine weights of every row
    // position of those three lanes, and the bulk stencil as slot 3 -- and EVERY lane forms its sums with weights
    // read from its slot (lanes of slot 3 read one address: a broadcast).  No branches, static register indices,
    // same summation order as the bulk form.  (Per-lane pointers into the stencils under a divergent branch cost
    // the tile kernels 130+ VGPRs; separate passes over the boundary rows 72 more LDS reads and FMAs per operator.)
    if constexpr (!FAST) {
        const int ls = (nr - 4) / Q;
        int co = (lane == 0 ? 0 : (lane == ls ? 1 : (lane == ls + 1 ? 2 : 3))) * (Q * 10);
#ifdef XSCAN_CS_DEPTH2
        // (the pair type with the wide stencils: one row at a time -- two rows' ten weights each do not fit beside two windows)
        constexpr bool D2 = NARROW || sizeof(T) == sizeof(real_t);
#else
        constexpr bool D2 = false;
#endif
        int co1 = co;  // two rows' weights in flight: row q + 2's reads wait for acc[q], row q + 1's do not
        if constexpr (D2) asm volatile("" : "+v"(co1));
#pragma unroll
        for (int q = 0; q < Q; q++) {
            int &coq = (D2 && (q & 1)) ? co1 : co;
            const real2_t *__restrict__ c2 = reinterpret_cast<const real2_t *>(cs + coq + q * 10);
            if (NARROW) {  // no stencil of the operator reaches beyond 2 rows (x3d_tdsops::narrow_all): taps 2..6 only
                const real2_t cb = c2[1], cc = c2[2], cd = c2[3];
                acc[q] = cb.x * w[q + 2] + cb.y * w[q + 3] + cc.x * w[q + 4] + cc.y * w[q + 5] + cd.x * w[q + 6];
            } else {
                const real2_t ca = c2[0], cb = c2[1], cc = c2[2], cd = c2[3], ce = c2[4];
                acc[q] = ca.x * w[q] + ca.y * w[q + 1] + cb.x * w[q + 2] + cb.y * w[q + 3] + cc.x * w[q + 4] +
                         cc.y * w[q + 5] + cd.x * w[q + 6] + cd.y * w[q + 7] + ce.x * w[q + 8];
            }
            asm volatile("" : "+v"(coq) : "v"(first_of(acc[q])));  // (one / two rows' weights at a time: 10 VGPRs each, not 10 Q)
        }
    }
    // ---- lane-local forward elimination from zero
    T prev = zero_of<T>();
#pragma unroll
    for (int q = 0; q < Q; q++) {
        X[q] = LTX(lt, LT_F(q)) * (acc[q] - LTX(lt, LT_A(q)) * prev);
        prev = X[q];
    }
    // ---- scan of the lane-end values, then carry-in = true e at the end of lane l-1
    T v = prev;
    PHASE(X[Q / 2]);
#if XSCAN_EXP == 3
    T carry = v;
#else
    // prefix scan without LDS traffic: in-row Kogge-Stone by DPP row shifts, then lane 15 / 47 into rows
    // 1 / 3 and lane 31 into rows 2, 3 (row_bcast); lanes without a source read 0.  (The row_bcast moves run with
    // all rows enabled: the rows that are not meant to receive carry a multiplier of exactly 0 in the tables,
    // tds.hip, so whatever finite value they pick up adds 0 -- and no zeroed destination is needed.)
    v += LTM(lt, 0) * dpp0<0x111>(v);
    v += LTM(lt, 1) * dpp0<0x112>(v);
    v += LTM(lt, 2) * dpp0<0x114>(v);
    v += LTM(lt, 3) * dpp0<0x118>(v);
    v += LTM(lt, 4) * dpp0<0x142>(v);
    v += LTM(lt, 5) * dpp0<0x143>(v);
    T carry = dpp0<0x138>(v);  // wave_shr:1
#endif
    // ---- apply, lane-local back-substitution from zero
    T nxt = zero_of<T>();
    PHASE(carry);
#pragma unroll
    for (int q = Q - 1; q >= 0; q--) {
        X[q] = (X[q] + LTX(lt, LT_PF(q)) * carry) + LTX(lt, LT_H(q)) * nxt;
        nxt = X[q];
    }
    if (n == nr) {}  // (row n_rhs = n+1 of a v2p operator carries F = H = 0 in the tables)
    v = nxt;
    PHASE(X[Q / 2]);
#if XSCAN_EXP == 3
    carry = v;
#else
    // suffix scan: row shifts the other way, then lane 16 / 48 into rows 0 / 2 and lane 32 into rows 0, 1
    v += LTM(lt, 6 + 0) * dpp0<0x101>(v);
    v += LTM(lt, 6 + 1) * dpp0<0x102>(v);
    v += LTM(lt, 6 + 2) * dpp0<0x104>(v);
    v += LTM(lt, 6 + 3) * dpp0<0x108>(v);
    {
        const T s16 = readlane_d(v, 16), s48 = readlane_d(v, 48);
        v += LTM(lt, 6 + 4) * sel_of(lane < 32, s16, s48);
        v += LTM(lt, 6 + 5) * readlane_d(v, 32);
    }
    carry = dpp0<0x130>(v);  // wave_shl:1
#endif
#pragma unroll
    for (int q = 0; q < Q; q++) X[q] = X[q] + LTX(lt, LT_QB(q)) * carry;
    // du_1 = last_r * X_1 (X_1 = e_1 - bw_1 X_2, distributed.f90:161-166); X_n = e_n
    du1 = t.last_r * readlane_d(X[0], 0);
    if constexpr (FAST) {
        xn = readlane_d(X[Q - 1], 63);
    } else {
        // row n sits in lane ln at position qn, both wave-uniform: v_readlane instead of __shfl = ds_bpermute (an LDS
        // round trip on the critical path to du_e); a scalar branch over q instead of the select chain spills in
        // k_ygen_transeq3<6, true, false>
        const int ln = __builtin_amdgcn_readfirstlane((n - 1) / Q), qn = __builtin_amdgcn_readfirstlane((n - 1) % Q);
        T xsel = zero_of<T>();
#pragma unroll
        for (int q = 0; q < Q; q++) xsel = sel_of(q == qn, X[q], xsel);
        xn = readlane_d(xsel, ln);
    }
    PHASE(xn);
#undef PHASE
}

// CIRCULANT form (round 6): one periodic operator on a uniform grid, lane-local + scans, WITHOUT lane tables and without the
// reduced system.  (alpha, 1, alpha) = (alpha / rho) (1 + rho z^-1) (1 + rho z): e_j = g r_j - rho e_{j-1} around the ring,
// x_j = e_j - rho x_{j+1} around the ring (g = rho / alpha, folded into the stencil).  Lane l runs both recurrences over its
// Q rows from zero; the value carried in from its neighbour is E_l = V_l + mu E_{l-1} (mu = (-rho)^Q), a CONSTANT-multiplier
// recurrence over the lanes: Kogge-Stone inside each row of 16 lanes by DPP row shifts with mu, mu^2, mu^4 (, mu^8) from
// SGPRs -- mu^8 (Q = 8) / mu^16 (Q = 4) are below 2^-60 for every scheme of the reference (rho <= 0.382: compact6's first
// derivative), the truncation its own 2 x 2 closure makes (src/tdsops.f90:196-201) -- then the neighbouring row's total
// enters through its first (last) lane by a wave rotate, which also closes the ring (lane 0 <- lane 63: the periodic wrap
// needs no closure at all), and spreads along the row by the same shifts.  Per right-hand side at Q = 8, 5-tap stencil:
// ~118 vector instructions and NO LDS reads (scan_solve's periodic form + substitution: ~150 and 68).
#ifdef CIRC_SCHED  // (experiment: scheduling barriers between the phases -- the EPI form of k_ytile_transeq3 then fits its
                   //  128 VGPRs (14 spilled without), and runs SLOWER: 3.33 against 3.11 ms; the y launch 1.86 against 1.72)
#define CIRC_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define CIRC_SB()
#endif
// CircOp2: TWO operators side by side -- the pair type's two right-hand sides each take their own constants (the operator
// pairs of the pressure correction: different operators on the same or on two inputs, ONE pass through the phases, two
// independent dependency chains in flight)
struct CircOp2 {
    V2 c[9], nr, pf[8], mu[4];
};
__device__ __forceinline__ CircOp2 circ_pair(const CircOp &a, const CircOp &b)
{
    CircOp2 r;
#pragma unroll
    for (int m = 0; m < 9; m++) r.c[m] = V2{a.c[m], b.c[m]};
    r.nr = V2{a.nr, b.nr};
#pragma unroll
    for (int m = 0; m < 8; m++) r.pf[m] = V2{a.pf[m], b.pf[m]};
#pragma unroll
    for (int m = 0; m < 4; m++) r.mu[m] = V2{a.mu[m], b.mu[m]};
    return r;
}
// OPEN (the HALO form: a decomposed direction, T = real_t): the pencil's two ends are open instead of joined -- nothing is
// carried into lane 0, the backward sweep's lane 63 takes phi0 * E, what this rank's own forward end state E makes of the
// next rank's first row; *bnd = {x_1 (for the previous rank), E (for the next)}.  What the neighbours' values add comes
// afterwards, on the boundary strips (tds.hip, tabc).
template <int Q, bool NARROW, class T = real_t, class OP = CircOp, bool OPEN = false>
__device__ __forceinline__ void circ_solve(const T (&w)[Q + 8], T (&X)[Q], const OP &t, int lane, T *bnd = nullptr)
{
    constexpr bool S8 = Q < 8;  // a fourth shift step (distance 8) where mu^8 is not yet negligible
    const auto c0 = t.c[0], c1 = t.c[1], c2 = t.c[2], c3 = t.c[3], c4 = t.c[4], c5 = t.c[5], c6 = t.c[6], c7 = t.c[7],
               c8 = t.c[8];
    const auto nr = t.nr, m1 = t.mu[0], m2 = t.mu[1], m4 = t.mu[2], m8 = t.mu[3];
    T acc[Q];
    if (NARROW) {
#pragma unroll
        for (int q = 0; q < Q; q++)
            acc[q] = fma_of(c6, w[q + 6], fma_of(c5, w[q + 5], fma_of(c4, w[q + 4], fma_of(c3, w[q + 3], c2 * w[q + 2]))));
    } else {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            T a = fma_of(c3, w[q + 3], fma_of(c2, w[q + 2], fma_of(c1, w[q + 1], c0 * w[q])));
            a = fma_of(c6, w[q + 6], fma_of(c5, w[q + 5], fma_of(c4, w[q + 4], a)));
            acc[q] = fma_of(c8, w[q + 8], fma_of(c7, w[q + 7], a));
        }
    }
    const bool row_first = (lane & 15) == 0, row_last = (lane & 15) == 15;
    CIRC_SB();
    T prev = zero_of<T>();
#pragma unroll
    for (int q = 0; q < Q; q++) {
        X[q] = fma_of(nr, prev, acc[q]);
        prev = X[q];
    }
    T v = prev;
    CIRC_SB();
    v = fma_of(m1, dpp0<0x111>(v), v);  // row_shr:1, 2, 4 (, 8): lanes without a source read 0
    v = fma_of(m2, dpp0<0x112>(v), v);
    v = fma_of(m4, dpp0<0x114>(v), v);
    if (S8) v = fma_of(m8, dpp0<0x118>(v), v);
    {
        // wave_ror:1: the previous row's total, lane 0 <- lane 63 (OPEN: wave_shr:1, lane 0 <- 0)
        T z = sel_of(row_first, OPEN ? dpp0<0x138>(v) : dpp0<0x13C>(v), zero_of<T>());
        z = fma_of(m1, dpp0<0x111>(z), z);
        z = fma_of(m2, dpp0<0x112>(z), z);
        z = fma_of(m4, dpp0<0x114>(z), z);
        if (S8) z = fma_of(m8, dpp0<0x118>(z), z);
        v = fma_of(m1, z, v);
    }
    T carry = OPEN ? dpp0<0x138>(v) : dpp0<0x13C>(v);
    T nxt = zero_of<T>();
    T endc = zero_of<T>();  // OPEN: what enters the backward sweep at the pencil's end
    if constexpr (OPEN) {
        const T E = readlane_d(v, 63);
        bnd[1] = E;
        endc = t.phi0 * E;
    }
    CIRC_SB();
    // (16 rows per lane, the 1024-row x pencils: (-rho)^(q + 1) = (-rho)^(q - 7) (-rho)^8 for the upper eight rows)
    T carry8 = zero_of<T>();
    if constexpr (Q > 8) carry8 = t.pf[7] * carry;
#pragma unroll
    for (int q = Q - 1; q >= 0; q--) {
        X[q] = fma_of(nr, nxt, q < 8 ? fma_of(t.pf[q], carry, X[q]) : fma_of(t.pf[q - 8], carry8, X[q]));
        nxt = X[q];
    }
    v = nxt;
    CIRC_SB();
    v = fma_of(m1, dpp0<0x101>(v), v);  // row_shl:1, 2, 4 (, 8)
    v = fma_of(m2, dpp0<0x102>(v), v);
    v = fma_of(m4, dpp0<0x104>(v), v);
    if (S8) v = fma_of(m8, dpp0<0x108>(v), v);
    {
        // wave_rol:1: the next row's total, lane 63 <- lane 0 (OPEN: wave_shl:1, lane 63 <- endc)
        T z = sel_of(row_last, OPEN ? sel_of(lane == 63, endc, dpp0<0x130>(v)) : dpp0<0x134>(v), zero_of<T>());
        z = fma_of(m1, dpp0<0x101>(z), z);
        z = fma_of(m2, dpp0<0x102>(z), z);
        z = fma_of(m4, dpp0<0x104>(z), z);
        if (S8) z = fma_of(m8, dpp0<0x108>(z), z);
        v = fma_of(m1, z, v);
    }
    carry = OPEN ? sel_of(lane == 63, endc, dpp0<0x130>(v)) : dpp0<0x134>(v);
    CIRC_SB();
    if constexpr (Q > 8) carry8 = t.pf[7] * carry;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int k = Q - 1 - q;
        X[q] = k < 8 ? fma_of(t.pf[k], carry, X[q]) : fma_of(t.pf[k - 8], carry8, X[q]);
    }
    if constexpr (OPEN) bnd[0] = readlane_d(X[0], 0);
}

// Two DIFFERENT operators (lane tables la / lb, descriptors ta / tb) on two right-hand sides (w[.].a, w[.].b) as ONE
// interleaved solve -- the FAST (periodic-type) form of scan_solve with every table value read per operator: the same
// number of LDS reads as two solves one after the other, but two independent dependency chains in flight and one pass
// through the scan's phases (round 5: the z-transforming operator pairs are bound by their serial on-chip chain,
// profiles/r05_zf_pair_phases.txt).  Arithmetic per right-hand side = scan_solve<Q, true, NARROW>, expression by expression.
template <int Q, bool NARROW>
__device__ __forceinline__ void scan_solve_dual(const V2 (&w)[Q + 8], V2 (&X)[Q], V2 &du1, V2 &xn,
                                                const real_t *__restrict__ la, const real_t *__restrict__ lb, const XOp &ta,
                                                const XOp &tb, int &lane)
{
#define PHASE2(x) asm volatile("" : "+v"(lane) : "v"((x).a))
#define L2(e) V2{lt_read(la, (e) * 64 + lane), lt_read(lb, (e) * 64 + lane)}
    V2 acc[Q];
    if (NARROW) {
        const V2 c2{ta.c[2], tb.c[2]}, c3{ta.c[3], tb.c[3]}, c4{ta.c[4], tb.c[4]}, c5{ta.c[5], tb.c[5]}, c6{ta.c[6], tb.c[6]};
#pragma unroll
        for (int q = 0; q < Q; q++)
            acc[q] = c2 * w[q + 2] + c3 * w[q + 3] + c4 * w[q + 4] + c5 * w[q + 5] + c6 * w[q + 6];
    } else {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            V2 a = V2{ta.c[0], tb.c[0]} * w[q];
#pragma unroll
            for (int m = 1; m < 9; m++) a = a + V2{ta.c[m], tb.c[m]} * w[q + m];
            acc[q] = a;
        }
    }
    V2 prev = zero_of<V2>();
#pragma unroll
    for (int q = 0; q < Q; q++) {
        X[q] = L2(LT_F(q)) * (acc[q] - L2(LT_A(q)) * prev);
        prev = X[q];
    }
    V2 v = prev;
    PHASE2(X[Q / 2]);
    v += L2(LT_MF(0)) * dpp0<0x111>(v);
    v += L2(LT_MF(1)) * dpp0<0x112>(v);
    v += L2(LT_MF(2)) * dpp0<0x114>(v);
    v += L2(LT_MF(3)) * dpp0<0x118>(v);
    v += L2(LT_MF(4)) * dpp0<0x142>(v);
    v += L2(LT_MF(5)) * dpp0<0x143>(v);
    V2 carry = dpp0<0x138>(v);  // wave_shr:1
    V2 nxt = zero_of<V2>();
    PHASE2(carry);
#pragma unroll
    for (int q = Q - 1; q >= 0; q--) {
        X[q] = (X[q] + L2(LT_PF(q)) * carry) + L2(LT_H(q)) * nxt;
        nxt = X[q];
    }
    v = nxt;
    PHASE2(X[Q / 2]);
    v += L2(LT_MB(0)) * dpp0<0x101>(v);
    v += L2(LT_MB(1)) * dpp0<0x102>(v);
    v += L2(LT_MB(2)) * dpp0<0x104>(v);
    v += L2(LT_MB(3)) * dpp0<0x108>(v);
    {
        const V2 s16 = readlane_d(v, 16), s48 = readlane_d(v, 48);
        v += L2(LT_MB(4)) * sel_of(lane < 32, s16, s48);
        v += L2(LT_MB(5)) * readlane_d(v, 32);
    }
    carry = dpp0<0x130>(v);  // wave_shl:1
#pragma unroll
    for (int q = 0; q < Q; q++) X[q] = X[q] + L2(LT_QB(q)) * carry;
    du1 = V2{ta.last_r, tb.last_r} * readlane_d(X[0], 0);
    xn = readlane_d(X[Q - 1], 63);
    PHASE2(xn);
#undef PHASE2
#undef L2
}

// nr == 64*Q and n_wrap == nr: every lane's body is a full aligned vector and the halos are
// the neighbours' rows or the periodic image -> no per-lane branches at all
template <int Q>
__device__ __forceinline__ void load_window_exact(real_t (&w)[Q + 8], const real_t *__restrict__ row, int lane,
                                                  int nr)
{
    const real2_t *__restrict__ body = reinterpret_cast<const real2_t *>(row + lane * Q);
#pragma unroll
    for (int m = 0; m < Q / 2; m++) {
        const real2_t t2 = body[m];
        w[4 + 2 * m] = t2.x;
        w[5 + 2 * m] = t2.y;
    }
    const int il = lane == 0 ? nr - 4 : lane * Q - 4;       // rows first-4..first-1 (periodic image for lane 0)
    const int ir = lane == 63 ? 0 : lane * Q + Q;           // rows last+1..last+4
    const real2_t *__restrict__ hl = reinterpret_cast<const real2_t *>(row + il);
    const real2_t *__restrict__ hr = reinterpret_cast<const real2_t *>(row + ir);
    const real2_t a0 = hl[0], a1 = hl[1], b0 = hr[0], b1 = hr[1];
    w[0] = a0.x; w[1] = a0.y; w[2] = a1.x; w[3] = a1.y;
    w[Q + 4] = b0.x; w[Q + 5] = b0.y; w[Q + 6] = b1.x; w[Q + 7] = b1.y;
}

// ---- 4 x 4 transpose of 16-byte pairs inside each quad of lanes (stores of xscan.hip; loads below)
__device__ __forceinline__ real_t dpp_quad(real_t v, int k)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (k == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x93, 0xf, 0xf, false);
                  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x93, 0xf, 0xf, false); }
    if (k == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false);
                  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }
    if (k == 3) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x39, 0xf, 0xf, false);
                  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x39, 0xf, 0xf, false); }
    return __hiloint2double(hi, lo);
}
// (scalars, not arrays: LLVM turns `c ? a[i] : a[k]` into a dynamically indexed stack array)
__device__ __forceinline__ void rotl_pairs(real_t &a0, real_t &a1, real_t &b0, real_t &b1, real_t &c0, real_t &c1,
                                           real_t &d0, real_t &d1, int j)
{
    const bool r1 = j & 1, r2 = j & 2;
    real_t t0 = a0, t1 = a1;  // rotate left by one pair where r1
    a0 = r1 ? b0 : a0; a1 = r1 ? b1 : a1;
    b0 = r1 ? c0 : b0; b1 = r1 ? c1 : b1;
    c0 = r1 ? d0 : c0; c1 = r1 ? d1 : c1;
    d0 = r1 ? t0 : d0; d1 = r1 ? t1 : d1;
    t0 = a0; t1 = a1;         // by two pairs where r2
    a0 = r2 ? c0 : a0; a1 = r2 ? c1 : a1;
    c0 = r2 ? t0 : c0; c1 = r2 ? t1 : c1;
    t0 = b0; t1 = b1;
    b0 = r2 ? d0 : b0; b1 = r2 ? d1 : b1;
    d0 = r2 ? t0 : d0; d1 = r2 ? t1 : d1;
}
// XS_TLOAD (experiment, round 4): the lane's 64 bytes fetched as whole 64-byte sectors per instruction (instruction m
// takes piece 4 m + j of the quad's 256 bytes) and transposed back by the quad; 2: with the nontemporal hint
#ifndef XS_TLOAD
#define XS_TLOAD 0
#endif
__device__ __forceinline__ void load_body_q8t(real_t (&b)[8], const real_t *__restrict__ row, int lane)
{
    const int j = lane & 3;
    const real_t *__restrict__ q = row + (lane & ~3) * 8 + 2 * j;
#if XS_TLOAD == 2
    const real2_t *__restrict__ qs = reinterpret_cast<const real2_t *>(q);
    const real2_t v0 = ldg_stream(qs), v1 = ldg_stream(qs + 4), v2 = ldg_stream(qs + 8), v3 = ldg_stream(qs + 12);
#else
    const real2_t *__restrict__ q2 = reinterpret_cast<const real2_t *>(q);
    const real2_t v0 = q2[0], v1 = q2[4], v2 = q2[8], v3 = q2[12];
#endif
    real_t a0 = v0.x, a1 = v0.y, b0 = v1.x, b1 = v1.y, c0 = v2.x, c1 = v2.y, d0 = v3.x, d1 = v3.y;
    rotl_pairs(a0, a1, b0, b1, c0, c1, d0, d1, j);
    b0 = dpp_quad(b0, 1); b1 = dpp_quad(b1, 1);
    c0 = dpp_quad(c0, 2); c1 = dpp_quad(c1, 2);
    d0 = dpp_quad(d0, 3); d1 = dpp_quad(d1, 3);
    rotl_pairs(a0, a1, b0, b1, c0, c1, d0, d1, j);
    b[0] = a0; b[1] = a1; b[2] = d0; b[3] = d1; b[4] = c0; b[5] = c1; b[6] = b0; b[7] = b1;
}

// FAST path: only the lane's own Q rows come from memory (4 aligned 16-byte loads, issued one
// pencil ahead); the 4+4 halo rows are the neighbour lanes' rows (periodic wrap across the wave)
template <int Q>
__device__ __forceinline__ void load_body(real_t (&b)[Q], const real_t *__restrict__ row, int lane)
{
#if XS_TLOAD
    if constexpr (Q == 8) { load_body_q8t(b, row, lane); return; }
#endif
    const real2_t *__restrict__ body = reinterpret_cast<const real2_t *>(row + lane * Q);
#pragma unroll
    for (int m = 0; m < Q / 2; m++) {
        // (plain, not ldg_stream: a lane's 64 bytes are four instructions that each touch a QUARTER of every 64-byte
        //  sector -- the lines must stay cached between them; with the nontemporal hint the x kernels ran 25 % slower,
        //  profiles/README.md round 4)
        const real2_t t2 = body[m];
        b[2 * m] = t2.x;
        b[2 * m + 1] = t2.y;
    }
}
template <int Q, class T = real_t>
__device__ __forceinline__ void window_from_body(T (&w)[Q + 8], const T (&b)[Q], int lane)
{
    (void)lane;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        w[m] = dpp0<0x13C>(b[Q - 4 + m]);      // wave_ror:1: from lane - 1, periodic wrap
        w[Q + 4 + m] = dpp0<0x134>(b[m]);      // wave_rol:1: from lane + 1
    }
#pragma unroll
    for (int q = 0; q < Q; q++) w[4 + q] = b[q];
}

// non-periodic pencil: rows before row 1 and after row 64 Q read as zero (their stencil weights are zero)
template <int Q>
__device__ __forceinline__ void window_from_body_zero(real_t (&w)[Q + 8], const real_t (&b)[Q])
{
#pragma unroll
    for (int m = 0; m < 4; m++) {
        w[m] = dpp0<0x138>(b[Q - 4 + m]);      // wave_shr:1: from lane - 1, lane 0 reads 0
        w[Q + 4 + m] = dpp0<0x130>(b[m]);      // wave_shl:1: from lane + 1, lane 63 reads 0
    }
#pragma unroll
    for (int q = 0; q < Q; q++) w[4 + q] = b[q];
}

// decomposed direction (BC_HALO ends): rows -3..0 and n+1..n+4 of the pencil are the neighbour ranks' rows, handed
// over in hl[0..3] / hl[4..7] (this wave's 8 halo values, staged in LDS) instead of the periodic image
template <int Q, class T = real_t>
__device__ __forceinline__ void window_from_body_halo(T (&w)[Q + 8], const T (&b)[Q], int lane,
                                                      const real_t *__restrict__ hl)
{
    window_from_body<Q, T>(w, b, lane);
    // (two one-lane branches: selects on all lanes keep 8 more values live where the kernels have no room)
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < 4; m++) w[m] = hl[m];
    }
    if (lane == 63) {
#pragma unroll
        for (int m = 0; m < 4; m++) w[Q + 4 + m] = hl[4 + m];
    }
}

// y = base + sum_k c[k] x[k] as the prologue of an x operator (k_xscan_tds_lin, k_xwide_tds_lin); wall != null: the
// pencils of the two y faces take `wall`'s rows instead (ny = rows per plane)
struct LinRows {
    real_t *y;
    const real_t *base;
    const real_t *x[5];
    real_t c[5];
    int n;
    const real_t *wall;
    int ny;
};
