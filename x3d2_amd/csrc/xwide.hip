// x-direction operators on 1024-row pencils, one pencil per wave (kernel family K3w): the single-pass scan
// kernels of xscan.hip with Q = 16 rows per lane (channel case: nx = 1024, BASELINE configs[4]).
//
// Two things differ from the 256 / 512-row kernels:
//  * lane tables: 156 entries x 64 lanes = 80 KB per operator would not fit twice; periodic-type operators on a
//    uniform grid have row entries that are bitwise constant away from the pencil's ends, so the tables are staged
//    in the compressed form of xscan_core.h (LTC_*: 26 KB per operator);
//  * memory access: a lane's 16 rows are 128 bytes -- loading them lane by lane would touch 16 bytes of 64
//    different lines per instruction.  Each wave moves its pencil with fully coalesced 16-byte accesses (1 KB
//    per instruction) and redistributes through a wave-private LDS strip (lane l's rows at l * 18 doubles: the
//    two padding doubles make both the coalesced and the lane-owned 16-byte accesses conflict-free); no barriers,
//    a wave's LDS operations execute in order (wave_lds_fence).
// Arithmetic: scan_solve, the same re-association of src/backend/omp/kernels/distributed.f90:34-166 (sweeps) and
// :186-337 (2 x 2 systems, substitution) as the other scan kernels; periodic self-exchange (sendrecv.f90:20-22).
#include "xscan_core.h"

constexpr int WQ = 16;             // rows per lane
constexpr int WSTRIP = 64 * 18;    // doubles of LDS per wave

// the wave's pencil <-> registers, coalesced: instruction m moves the 16-byte pieces lane + 64 m
__device__ __forceinline__ void wide_gload(real_t (&v)[16], const real_t *__restrict__ row, int lane)
{
    const real2_t *__restrict__ r2 = reinterpret_cast<const real2_t *>(row) + lane;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const real2_t t = r2[64 * m];
        v[2 * m] = t.x;
        v[2 * m + 1] = t.y;
    }
}
// piece g = lane + 64 m belongs to lane g / 8 (its pair g % 8)
__device__ __forceinline__ int wide_coal_off(int lane, int m) { return ((lane >> 3) + 8 * m) * 18 + (lane & 7) * 2; }
__device__ __forceinline__ void wide_to_strip(real_t *strip, const real_t (&v)[16], int lane)
{
#pragma unroll
    for (int m = 0; m < 8; m++)
        *reinterpret_cast<real2_t *>(strip + wide_coal_off(lane, m)) = make_real2(v[2 * m], v[2 * m + 1]);
}
__device__ __forceinline__ void wide_own_rows(real_t (&b)[WQ], const real_t *strip, int lane)
{
    const real2_t *s2 = reinterpret_cast<const real2_t *>(strip + lane * 18);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const real2_t t = s2[k];
        b[2 * k] = t.x;
        b[2 * k + 1] = t.y;
    }
}
__device__ __forceinline__ void wide_put_rows(real_t *strip, const real_t (&r)[WQ], int lane)
{
    real2_t *s2 = reinterpret_cast<real2_t *>(strip + lane * 18);
#pragma unroll
    for (int k = 0; k < 8; k++) s2[k] = make_real2(r[2 * k], r[2 * k + 1]);
}
// strip -> memory, coalesced; ACC: out = old + scale * r (the arithmetic of k_xscan_tds<ACC>)
template <bool ACC>
__device__ __forceinline__ void wide_store(real_t *__restrict__ orow, const real_t *strip, int lane, real_t scale)
{
    real2_t *__restrict__ o2 = reinterpret_cast<real2_t *>(orow) + lane;
    real2_t old[8];
    if (ACC) {
#pragma unroll
        for (int m = 0; m < 8; m++) old[m] = o2[64 * m];
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
        real2_t v = *reinterpret_cast<const real2_t *>(strip + wide_coal_off(lane, m));
        if (ACC) { v.x = fma_r(scale, v.x, old[m].x); v.y = fma_r(scale, v.y, old[m].y); }
        o2[64 * m] = v;
    }
}

// strip -> memory with c * (another field's rows, as wide_gload delivered them) added
__device__ __forceinline__ void wide_store_add(real_t *__restrict__ orow, const real_t *strip, int lane, real_t c,
                                               const real_t (&add)[16])
{
    real2_t *__restrict__ o2 = reinterpret_cast<real2_t *>(orow) + lane;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        real2_t v = *reinterpret_cast<const real2_t *>(strip + wide_coal_off(lane, m));
        v.x = fma_r(c, add[2 * m], v.x);
        v.y = fma_r(c, add[2 * m + 1], v.y);
        o2[64 * m] = v;
    }
}

// one operator on the window w: r = its tds_solve rows (closed with the periodic self-exchange)
template <bool NARROW>
__device__ __forceinline__ void wide_solve(const real_t (&w)[WQ + 8], real_t (&r)[WQ], const real_t *__restrict__ lt,
                                           const XOp &t, int &lane, int ll)
{
    constexpr int Q = WQ, LS = LTC_LS;
    real_t X[WQ], du1, xn;
    scan_solve<WQ, true, NARROW, real_t, LTC_LS>(w, X, du1, xn, lt, t, lane, lane * WQ + 1, ll);
    const real_t du_s = t.rs_s * (du1 - t.sa1 * xn), du_e = t.rs_e * (xn - t.scn * du1);
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        const real_t st = LTX(lt, LT_ST(q));
        real_t x = (X[q] - LTX(lt, LT_SA(q)) * du_s - LTX(lt, LT_SC(q)) * du_e) * st;
        if (q == 0) x = (lane == 0) ? du_s * st : x;
        if (q == WQ - 1) x = (lane == 63) ? du_e * st : x;
        r[q] = x;
    }
}

// Uniform grid (x3d_tdsops::uniform: ST == 1 and STC == 0 on every row): the lane tables are staged WITHOUT those two
// blocks (21.4 instead of 25.7 KB per operator) and the solve skips the multiplications by 1: the transeq kernels below.
constexpr int LTU_ROWS = 7 * WQ * LTC_LS;  // F A PF H QB SA SC of the compressed layout
constexpr int LTU_N = LTU_ROWS + 12 * 64;  // doubles per operator

template <bool NARROW>
__device__ __forceinline__ void wide_solve_u(const real_t (&w)[WQ + 8], real_t (&r)[WQ], const real_t *__restrict__ lt,
                                             const XOp &t, int &lane, int ll)
{
    constexpr int Q = WQ, LS = LTC_LS;
    real_t X[WQ], du1, xn;
    scan_solve<WQ, true, NARROW, real_t, LTC_LS, LTU_ROWS>(w, X, du1, xn, lt, t, lane, lane * WQ + 1, ll);
    const real_t du_s = t.rs_s * (du1 - t.sa1 * xn), du_e = t.rs_e * (xn - t.scn * du1);
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        real_t x = X[q] - LTX(lt, LT_SA(q)) * du_s - LTX(lt, LT_SC(q)) * du_e;
        if (q == 0) x = (lane == 0) ? du_s : x;
        if (q == WQ - 1) x = (lane == 63) ? du_e : x;
        r[q] = x;
    }
}

// ---------------------------------------------------------------- tds_solve
// psum != null (ACC = false only): the sum of u over the pencils of the y rows j < ny_sum rides along, one partial per wave
// (see k_xwide_tds_lin below: the channel case's bulk-velocity integral)
// CIRC (round 6): the operator in the circulant form (xscan_core.h circ_solve with 16 rows per lane; co) -- no lane tables,
// the workgroup's LDS is its eight strips
template <bool ACC, bool NARROW, bool CIRC = false>
__global__ void __launch_bounds__(512) k_xwide_tds(real_t *__restrict__ du, const real_t *__restrict__ u, XOp t, int np,
                                                   long pitch, real_t scale, real_t *__restrict__ psum, int ny_sum, int ny,
                                                   CircOp co)
{
    extern __shared__ real_t lt[];  // [LTC_N(16)] tables, then one strip per wave
    if constexpr (!CIRC) {
        for (int i = threadIdx.x; i < LTC_N(WQ); i += blockDim.x) lt[i] = t.TL[i];
        __syncthreads();
    }
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    real_t *strip = lt + (CIRC ? 0 : LTC_N(WQ)) + wave * WSTRIP;
    const int ll = ltc_lane(lane);
    const int p0 = blockIdx.x * (blockDim.x >> 6) + wave;
    real_t nxt[16];  // next pencil's pieces, in flight while this one is solved
    if (p0 < np) wide_gload(nxt, u + (long)p0 * pitch, lane);
    real_t wsum = 0.0;
    for (int p = p0; p < np; p += nwaves) {
        asm volatile("" : "+v"(lane));  // keep the lane-table reads inside the loop
        real_t w[WQ + 8], r[WQ];
        if (!ACC && psum && p % ny < ny_sum) {  // (wave-uniform)
            real_t s_ = 0.0;
#pragma unroll
            for (int m = 0; m < 16; m++) s_ += nxt[m];
            wsum += s_;
        }
        {
            real_t b[WQ];
            wide_to_strip(strip, nxt, lane);
            wave_lds_fence();
            wide_own_rows(b, strip, lane);
            window_from_body<WQ>(w, b, lane);
        }
        if (p + nwaves < np) wide_gload(nxt, u + (long)(p + nwaves) * pitch, lane);
        if constexpr (CIRC) circ_solve<WQ, NARROW>(w, r, co, lane);
        else wide_solve<NARROW>(w, r, lt, t, lane, ll);
        wave_lds_fence();  // (every lane has read its rows: the strip may be rewritten)
        wide_put_rows(strip, r, lane);
        wave_lds_fence();
        wide_store<ACC>(du + (long)p * pitch, strip, lane, scale);
        wave_lds_fence();
    }
    if (!ACC && psum) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
        if (lane == 0) psum[blockIdx.x * (blockDim.x >> 6) + wave] = wsum;
    }
}

// ---------------------------------------------------------------- tds_solve of a field that is still to be formed
// k_xscan_tds_lin for 1024-row pencils: y = base + sum c_k x_k (the RK stage, summation order of k_lincomb) formed on
// the coalesced pieces, the y faces stamped from `wall` if given, y stored, du = tds_solve(y) -- y is not read back.
// psum != null (round 6): the sum over the pencils of the y rows j < ny_sum of the y that was just formed -- the volume integral
// the channel case's next define_BC asks for (src/case/channel.f90:66-72), taken while the rows are in registers
// instead of by a reduction pass of its own: one partial per wave in a fixed order (pencils in the wave's loop order,
// the 16 pieces of a lane, then the lanes by butterfly), psum[global wave index]; x3d_xwide_tds_lincomb finishes them.
template <bool NARROW, bool CIRC = false>
__global__ void __launch_bounds__(512) k_xwide_tds_lin(real_t *__restrict__ du, LinRows lr, XOp t, int np, long pitch,
                                                       real_t *__restrict__ psum, int ny_sum, CircOp co)
{
    extern __shared__ real_t lt[];
    if constexpr (!CIRC) {
        for (int i = threadIdx.x; i < LTC_N(WQ); i += blockDim.x) lt[i] = t.TL[i];
        __syncthreads();
    }
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    real_t *strip = lt + (CIRC ? 0 : LTC_N(WQ)) + wave * WSTRIP;
    const int ll = ltc_lane(lane);
    real_t wsum = 0.0;
    for (int p = blockIdx.x * (blockDim.x >> 6) + wave; p < np; p += nwaves) {
        const long ro = (long)p * pitch;
        asm volatile("" : "+v"(lane));
        real_t v[16];
        wide_gload(v, lr.base + ro, lane);
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (k < lr.n) {
                real_t xk[16];
                wide_gload(xk, lr.x[k] + ro, lane);
#pragma unroll
                for (int m = 0; m < 16; m++) v[m] = lr.c[k] * xk[m] + v[m];
            }
        if (lr.wall) {
            const int j = p % lr.ny;
            if (j == 0 || j == lr.ny - 1) wide_gload(v, lr.wall + ro, lane);
        }
        {
            real2_t *__restrict__ o2 = reinterpret_cast<real2_t *>(lr.y + ro) + lane;
#pragma unroll
            for (int m = 0; m < 8; m++) o2[64 * m] = make_real2(v[2 * m], v[2 * m + 1]);
        }
        if (psum && p % lr.ny < ny_sum) {  // (wave-uniform)
            real_t s = 0.0;
#pragma unroll
            for (int m = 0; m < 16; m++) s += v[m];
            wsum += s;
        }
        real_t w[WQ + 8], r[WQ];
        {
            real_t b[WQ];
            wide_to_strip(strip, v, lane);
            wave_lds_fence();
            wide_own_rows(b, strip, lane);
            window_from_body<WQ>(w, b, lane);
        }
        if constexpr (CIRC) circ_solve<WQ, NARROW>(w, r, co, lane);
        else wide_solve<NARROW>(w, r, lt, t, lane, ll);
        wave_lds_fence();
        wide_put_rows(strip, r, lane);
        wave_lds_fence();
        wide_store<false>(du + ro, strip, lane, 1.0);
        wave_lds_fence();
    }
    if (psum) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
        if (lane == 0) psum[blockIdx.x * (blockDim.x >> 6) + wave] = wsum;
    }
}

// ---------------------------------------------------------------- transeq_x: the three components at once
// (u0, conv = u0), (u1, u0), (u2, u0) as in k_xscan_transeq2x3 (src/backend/omp/backend.f90:145-184,
// exec_dist.f90:85-169 per component): the pencil's rows of the advecting velocity stay in registers, 6 field
// passes for the direction.  der1st == der1st_sym and der2nd == der2nd_sym as lane tables (periodic operators).
// ROT: the channel case's rotation forcing on top (src/case/channel.f90:191-207: du -= omega v, dv += omega u,
// there two vecadd's after the three directions = 6 field passes): v's rows are in flight when du is stored, u's
// rows are in registers when dv is formed -- no extra traffic.
// UNI: all operators on a uniform grid -- compact tables, wide_solve_u, no stretch-correction term (x * 1 and + nu T 0
// dropped: the same values; the compiler's choice of which products it contracts may move a last bit against the general form)
template <bool ACC, bool NARROW, bool ROT = false, bool UNI = false, bool CIRC = false>
__global__ void __launch_bounds__(512)
    k_xwide_transeq3(real_t *__restrict__ rhs0, real_t *__restrict__ rhs1, real_t *__restrict__ rhs2,
                     const real_t *u0, const real_t *__restrict__ u1, const real_t *__restrict__ u2, XOp tD1,
                     XOp tD2, int np, long pitch, real_t nu, real_t omega, const real_t *__restrict__ ushift, Circ4 cc)
{
    extern __shared__ real_t lt[];
    constexpr int LN = CIRC ? 0 : (UNI ? LTU_N : LTC_N(WQ)), Q = WQ, LS = LTC_LS;
    static_assert(!CIRC || UNI, "CIRC: uniform grids");
    if constexpr (CIRC) {
    } else if constexpr (UNI) {
        for (int i = threadIdx.x; i < LTU_ROWS; i += blockDim.x) {
            lt[i] = tD1.TL[i];
            lt[LN + i] = tD2.TL[i];
        }
        for (int i = threadIdx.x; i < 12 * 64; i += blockDim.x) {
            lt[LTU_ROWS + i] = tD1.TL[LTC_M0(WQ) + i];
            lt[LN + LTU_ROWS + i] = tD2.TL[LTC_M0(WQ) + i];
        }
    } else {
        for (int i = threadIdx.x; i < LN; i += blockDim.x) {
            lt[i] = tD1.TL[i];
            lt[LN + i] = tD2.TL[i];
        }
    }
    __syncthreads();
    const real_t *__restrict__ l1 = lt, *__restrict__ l3 = lt + (CIRC ? 1 : LN);  // (CIRC: only told apart, never read)
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    real_t *strip = lt + 2 * LN + wave * WSTRIP;
    const int ll = ltc_lane(lane);
    const int p0 = blockIdx.x * (blockDim.x >> 6) + wave;
    real_t nxt[16];  // the rows needed next (next component's field, or the next pencil's u0)
    if (p0 < np) wide_gload(nxt, u0 + (long)p0 * pitch, lane);
    // ushift != null: u0 += *ushift first, in place (the channel case's bulk-velocity shift, x3d_field_mean_shift:
    // the pencil is in registers anyway -- one write pass instead of a read + write pass of its own)
    const real_t ush = ushift ? *ushift : 0.0;
    for (int p = p0; p < np; p += nwaves) {
        const long ro = (long)p * pitch;
        real_t cb[WQ];
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            asm volatile("" : "+v"(lane));
            real_t wu[WQ + 8], wp[WQ + 8];
            {
                real_t b[WQ];
                if (c == 0 && ushift) {
                    real2_t *o2 = reinterpret_cast<real2_t *>(const_cast<real_t *>(u0) + ro) + lane;
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        nxt[2 * m] += ush;
                        nxt[2 * m + 1] += ush;
                        o2[64 * m] = make_real2(nxt[2 * m], nxt[2 * m + 1]);
                    }
                }
                wide_to_strip(strip, nxt, lane);
                wave_lds_fence();
                wide_own_rows(b, strip, lane);
                if (c == 0) {
#pragma unroll
                    for (int q = 0; q < WQ; q++) cb[q] = b[q];
                }
                window_from_body<WQ>(wu, b, lane);
                window_from_body<WQ>(wp, cb, lane);
#pragma unroll
                for (int m = 0; m < WQ + 8; m++) wp[m] = wu[m] * wp[m];
            }
            {
                const int pn = p + nwaves;
                const real_t *nsrc = c == 0 ? u1 + ro : (c == 1 ? u2 + ro : u0 + (long)(pn < np ? pn : p) * pitch);
                if (c < 2 || pn < np) wide_gload(nxt, nsrc, lane);
            }
            real_t r[WQ], T[WQ];
            auto solve = [&](const real_t (&w)[WQ + 8], const real_t *__restrict__ l, const XOp &t) {
                if constexpr (CIRC) { if (l == l1) circ_solve<WQ, NARROW>(w, T, cc.o[0], lane); else circ_solve<WQ, NARROW>(w, T, cc.o[1], lane); }
                else if constexpr (UNI) wide_solve_u<NARROW>(w, T, l, t, lane, ll);
                else wide_solve<NARROW>(w, T, l, t, lane, ll);
            };
            solve(wp, l1, tD1);  // d(u conv)/dx first: wp is dead afterwards
#pragma unroll
            for (int q = 0; q < WQ; q++) r[q] = T[q];
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            solve(wu, l1, tD1);  // du/dx
#pragma unroll
            for (int q = 0; q < WQ; q++) {
                if constexpr (UNI) r[q] = -0.5 * fma_r(cb[q], T[q], r[q]);
                else r[q] = -0.5 * (cb[q] * T[q] + r[q]) + nu * (T[q] * LTX(l3, LTC_STC(q)));
            }
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            solve(wu, l3, tD2);  // d2u/dx2
#pragma unroll
            for (int q = 0; q < WQ; q++) {
                if constexpr (UNI) r[q] = fma_r(nu, T[q], r[q]);  // (r is itself a product: see fma_r)
                else r[q] += nu * T[q];
            }
            if (ROT && c == 1) {
#pragma unroll
                for (int q = 0; q < WQ; q++) r[q] = fma_r(omega, cb[q], r[q]);
            }
            wave_lds_fence();
            wide_put_rows(strip, r, lane);
            wave_lds_fence();
            if (ROT && c == 0) wide_store_add(rhs0 + ro, strip, lane, -omega, nxt);  // (nxt = this pencil's u1 rows)
            else wide_store<ACC>((c == 0 ? rhs0 : (c == 1 ? rhs1 : rhs2)) + ro, strip, lane, 1.0);
            wave_lds_fence();
        }
    }
}

// ---------------------------------------------------------------- transeq_x + the pending velocity correction
// k_xscan_transeq2x3<UPD> for 1024-row pencils (the channel case, round 6): the velocity still waits for the previous
// sub-step's pressure-gradient correction u_c += scale * tds_solve(g_c) (the last x operators of gradient_c2v,
// src/vector_calculus.f90:318-330 + src/solver.f90:731-733; op tS for c = 0, tI for c = 1, 2).  Per pencil and component:
// g_c's rows -> solve -> the correction is added to u_c's rows (arithmetic of k_xwide_tds<ACC>: old + scale * r), u_c is
// written once (with the bulk-velocity shift of c = 0 on top, as k_xwide_transeq3 does it) and used at once: 12 field
// passes instead of 7 + 9.  Four operators' lane tables do not fit beside the eight strips in the compressed form
// (4 x 25.7 KB + 73.7 KB); on a uniform grid ST == 1 and STC == 0 on every row (x3d_tdsops::uniform), so the tables are
// staged WITHOUT those two blocks (LTU_N: 21.4 KB per operator, 159.2 KB in all) and the solves skip the
// multiplications by 1 and the additions of 0 (as k_xwide_transeq3<UNI>; against it u, v, w come out bit for bit, the
// derivatives to the last bit -- tests/test_hip_poisson_010.py).
// ROT: the rotation forcing needs the CORRECTED u1 for rhs0: component 0's rows wait in registers until component 1 has
// formed them (rhs0 = -omega u1 + 1.0 r0, rhs1 = omega u0 + 1.0 r1: the expressions of k_xwide_transeq3<ROT>).
struct WideUpd {
    const real_t *g[3];
    real_t scale, omega;
    const real_t *ushift;
};

template <bool NARROW, bool ROT, bool CIRC = false>
__global__ void __launch_bounds__(512)
    k_xwide_transeq3_upd(real_t *__restrict__ rhs0, real_t *__restrict__ rhs1, real_t *__restrict__ rhs2, real_t *u0,
                         real_t *u1, real_t *u2, XOp tD1, XOp tD2, XOp tS, XOp tI, WideUpd upd, int np, long pitch, real_t nu,
                         Circ4 cc)
{
    extern __shared__ real_t lt[];
    if constexpr (!CIRC) {
        const real_t *src[4] = {tD1.TL, tD2.TL, tS.TL, tI.TL};
#pragma unroll
        for (int o = 0; o < 4; o++) {
            for (int i = threadIdx.x; i < LTU_ROWS; i += blockDim.x) lt[o * LTU_N + i] = src[o][i];
            for (int i = threadIdx.x; i < 12 * 64; i += blockDim.x) lt[o * LTU_N + LTU_ROWS + i] = src[o][LTC_M0(WQ) + i];
        }
        __syncthreads();
    }
    const real_t *__restrict__ l1 = lt, *__restrict__ l3 = lt + LTU_N;
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    real_t *strip = lt + (CIRC ? 0 : 4 * LTU_N) + wave * WSTRIP;
    const int ll = ltc_lane(lane);
    const int p0 = blockIdx.x * (blockDim.x >> 6) + wave;
    real_t unx[16], gnx[16];  // the rows needed next: u_c's and g_c's pieces
    if (p0 < np) {
        wide_gload(unx, u0 + (long)p0 * pitch, lane);
        wide_gload(gnx, upd.g[0] + (long)p0 * pitch, lane);
    }
    const real_t ush = upd.ushift ? *upd.ushift : 0.0;
    for (int p = p0; p < np; p += nwaves) {
        const long ro = (long)p * pitch;
        const int pn = p + nwaves;
        const long rn = (long)(pn < np ? pn : p) * pitch;
        real_t cb[WQ], r0[WQ];
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            asm volatile("" : "+v"(lane));
            real_t b[WQ];  // u_c's rows of this lane: raw, then corrected (and shifted)
            wide_to_strip(strip, unx, lane);
            wave_lds_fence();
            wide_own_rows(b, strip, lane);
            wave_lds_fence();
            {
                const real_t *nsrc = c == 0 ? u1 + ro : (c == 1 ? u2 + ro : u0 + rn);
                if (c < 2 || pn < np) wide_gload(unx, nsrc, lane);
            }
            {
                real_t wg[WQ + 8], gr[WQ];
                {
                    real_t t[WQ];
                    wide_to_strip(strip, gnx, lane);
                    wave_lds_fence();
                    wide_own_rows(t, strip, lane);
                    window_from_body<WQ>(wg, t, lane);
                }
                {
                    const real_t *nsrc = c == 0 ? upd.g[1] + ro : (c == 1 ? upd.g[2] + ro : upd.g[0] + rn);
                    if (c < 2 || pn < np) wide_gload(gnx, nsrc, lane);
                }
                const real_t *__restrict__ lg = lt + (c == 0 ? 2 : 3) * LTU_N;
                if constexpr (CIRC) { if (c == 0) circ_solve<WQ, NARROW>(wg, gr, cc.o[2], lane); else circ_solve<WQ, NARROW>(wg, gr, cc.o[3], lane); }
                else wide_solve_u<NARROW>(wg, gr, lg, c == 0 ? tS : tI, lane, ll);
#pragma unroll
                for (int q = 0; q < WQ; q++) b[q] = fma_r(upd.scale, gr[q], b[q]);
            }
            if (c == 0 && upd.ushift) {  // (wave-uniform)
#pragma unroll
                for (int q = 0; q < WQ; q++) b[q] += ush;
            }
            wave_lds_fence();
            wide_put_rows(strip, b, lane);
            wave_lds_fence();
            wide_store<false>((c == 0 ? u0 : (c == 1 ? u1 : u2)) + ro, strip, lane, 1.0);
            wave_lds_fence();
            if (c == 0) {
#pragma unroll
                for (int q = 0; q < WQ; q++) cb[q] = b[q];
            }
            real_t wu[WQ + 8], wp[WQ + 8];
            window_from_body<WQ>(wu, b, lane);
            window_from_body<WQ>(wp, cb, lane);
#pragma unroll
            for (int m = 0; m < WQ + 8; m++) wp[m] = wu[m] * wp[m];
            real_t r[WQ], T[WQ];
            if constexpr (CIRC) circ_solve<WQ, NARROW>(wp, T, cc.o[0], lane);
            else wide_solve_u<NARROW>(wp, T, l1, tD1, lane, ll);  // d(u conv)/dx
#pragma unroll
            for (int q = 0; q < WQ; q++) r[q] = T[q];
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            if constexpr (CIRC) circ_solve<WQ, NARROW>(wu, T, cc.o[0], lane);
            else wide_solve_u<NARROW>(wu, T, l1, tD1, lane, ll);  // du/dx
#pragma unroll
            for (int q = 0; q < WQ; q++) r[q] = -0.5 * fma_r(cb[q], T[q], r[q]);
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            if constexpr (CIRC) circ_solve<WQ, NARROW>(wu, T, cc.o[1], lane);
            else wide_solve_u<NARROW>(wu, T, l3, tD2, lane, ll);  // d2u/dx2
#pragma unroll
            for (int q = 0; q < WQ; q++) r[q] = fma_r(nu, T[q], r[q]);  // (as k_xwide_transeq3<UNI>: the same bits)
            if (ROT && c == 0) {
#pragma unroll
                for (int q = 0; q < WQ; q++) r0[q] = r[q];
                continue;  // (stored with component 1, when the corrected u1 is known)
            }
            if (ROT && c == 1) {
#pragma unroll
                for (int q = 0; q < WQ; q++) {
                    r0[q] = fma_r(-upd.omega, b[q], r0[q]);
                    r[q] = fma_r(upd.omega, cb[q], r[q]);
                }
                wave_lds_fence();
                wide_put_rows(strip, r0, lane);
                wave_lds_fence();
                wide_store<false>(rhs0 + ro, strip, lane, 1.0);
            }
            wave_lds_fence();
            wide_put_rows(strip, r, lane);
            wave_lds_fence();
            wide_store<false>((c == 0 ? rhs0 : (c == 1 ? rhs1 : rhs2)) + ro, strip, lane, 1.0);
            wave_lds_fence();
        }
    }
}

// ---------------------------------------------------------------- launchers
static bool wide_env_on()
{
    static int on = -1;
    if (on < 0) {
        const char *names[3] = {"X3D_NO_XWIDE", "X3D_NO_XSCAN", "X3D_XDIR_GENERIC"};
        on = 1;
        for (const char *nm : names) { const char *e = getenv(nm); if (e && e[0] == '1') on = 0; }
    }
    return on == 1;
}
static bool wide_ok(const x3d_backend *b, const x3d_tdsops *t)
{
    return t->tlc != nullptr && t->tab.Q == WQ && t->tab.bulk_only && t->n_tds == 64 * WQ && t->tab.n_rhs == t->n_tds &&
           b->nx == 64 * WQ;
}
static bool wide_narrow(const x3d_tdsops *t)
{
    return t->coeffs[0] == 0.0 && t->coeffs[1] == 0.0 && t->coeffs[7] == 0.0 && t->coeffs[8] == 0.0;
}
// the circulant form of the 1024-row kernels (circ_solve with 16 rows per lane).  X3D_NO_CIRC=1 / X3D_NO_XCIRC=1: not (A/B)
static bool wide_circ(const x3d_tdsops *t)
{
    static int on = -1;
    if (on < 0) {
        on = 1;
        for (const char *nm : {"X3D_NO_CIRC", "X3D_NO_XCIRC"}) { const char *e = getenv(nm); if (e && e[0] == '1') on = 0; }
    }
    return on && t->circ_ok && t->uniform && t->tab.Q == 16;
}
static XOp wide_xop(const x3d_tdsops *t)
{
    XOp o = xop_of(t);
    o.TL = t->tlc;
    return o;
}

// tds_solve along x on 1024-row pencils; *done = false: not served here
// psum / ny_sum / nsum: as x3d_xwide_tds_lincomb's (the sum of u, acc == 0 only)
int x3d_xwide_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int acc, real_t scale, bool *done,
                  real_t *psum, int ny_sum, int *nsum)
{
    *done = false;
    if (nsum) *nsum = 0;
    if (acc) psum = nullptr;
    if (!wide_env_on() || !wide_ok(b, t)) return 0;
    const int np = b->ny * b->nz;
    const bool narrow = wide_narrow(t);
    const bool circ = narrow && wide_circ(t);
    const size_t lds = sizeof(real_t) * ((circ ? 0 : LTC_N(WQ)) + 8 * WSTRIP);
    int blocks = (np + 7) / 8;
    // 98 KB of LDS: one 8-wave workgroup per CU; circulant form: 74 KB of strips, two (not with the partial sums: their
    // finishing kernel takes up to 2048)
    const int cap = (circ && !psum) ? 512 : 256;
    blocks = blocks > cap ? cap : blocks;
    ProfScope ps(b, X3D_K_TDS_FWD, X3D_DIR_X);
#define GO(A_, N_, C_)                                                                                          \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_xwide_tds<A_, N_, C_>));                                                            \
        hipLaunchKernelGGL((k_xwide_tds<A_, N_, C_>), dim3(blocks), dim3(512), lds, b->stream, du, u, wide_xop(t), np, \
                           (long)b->nxp, A_ ? scale : 1.0, psum, ny_sum, b->ny, t->circ);                       \
    } while (0)
    if (acc) { if (circ) GO(true, true, true); else if (narrow) GO(true, true, false); else GO(true, false, false); }
    else { if (circ) GO(false, true, true); else if (narrow) GO(false, true, false); else GO(false, false, false); }
#undef GO
    X3D_HIP(hipGetLastError());
    if (nsum && psum) *nsum = blocks * 8;
    *done = true;
    return 0;
}

// y = base + sum c_k x_k ; y faces of y <- wall (if given) ; du = tds_solve(y) along x; *done = false: not served here
// psum / ny_sum: see k_xwide_tds_lin; *nsum = the number of partials written (the launch's waves)
int x3d_xwide_tds_lincomb(x3d_backend *b, real_t *du, const x3d_tdsops *t, real_t *y, const real_t *base, int nterm,
                          const real_t *c, const real_t *const *x, const real_t *wall, bool *done, real_t *psum, int ny_sum,
                          int *nsum)
{
    *done = false;
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_TDS_LINCOMB"); on = (e && e[0] == '1') ? 0 : 1; }
    if (!on || !wide_env_on() || !wide_ok(b, t)) return 0;
    const int np = b->ny * b->nz;
    const bool circ = wide_narrow(t) && wide_circ(t);  // (the same choice as x3d_xwide_tds: the same bits)
    const size_t lds = sizeof(real_t) * ((circ ? 0 : LTC_N(WQ)) + 8 * WSTRIP);
    int blocks = (np + 7) / 8;
    const int cap = (circ && !psum) ? 512 : 256;
    blocks = blocks > cap ? cap : blocks;
    LinRows lr;
    lr.y = y; lr.base = base; lr.n = nterm; lr.wall = wall; lr.ny = b->ny;
    for (int k = 0; k < 5; k++) { lr.x[k] = k < nterm ? x[k] : x[0]; lr.c[k] = k < nterm ? c[k] : 0.0; }
    ProfScope ps(b, X3D_K_TDS_FWD, X3D_DIR_X);
    if (circ) {
        X3D_LDS_OPTIN(b, (k_xwide_tds_lin<true, true>));
        hipLaunchKernelGGL((k_xwide_tds_lin<true, true>), dim3(blocks), dim3(512), lds, b->stream, du, lr, wide_xop(t), np,
                           (long)b->nxp, psum, ny_sum, t->circ);
    } else if (wide_narrow(t)) {
        X3D_LDS_OPTIN(b, (k_xwide_tds_lin<true>));
        hipLaunchKernelGGL((k_xwide_tds_lin<true>), dim3(blocks), dim3(512), lds, b->stream, du, lr, wide_xop(t), np,
                           (long)b->nxp, psum, ny_sum, t->circ);
    } else {
        X3D_LDS_OPTIN(b, (k_xwide_tds_lin<false>));
        hipLaunchKernelGGL((k_xwide_tds_lin<false>), dim3(blocks), dim3(512), lds, b->stream, du, lr, wide_xop(t), np,
                           (long)b->nxp, psum, ny_sum, t->circ);
    }
    X3D_HIP(hipGetLastError());
    if (nsum) *nsum = blocks * 8;
    *done = true;
    return 0;
}

// transeq_x in one launch; f[0] is the advecting component
int x3d_xwide_transeq3(x3d_backend *b, real_t *const r[3], const real_t *const f[3], real_t nu, const x3d_tdsops *der1st,
                       const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym, int acc,
                       real_t omega, const real_t *ushift, bool *done)
{
    *done = false;
    if (omega != 0.0 && acc) return 0;
    if (!wide_env_on() || !wide_ok(b, der1st) || !wide_ok(b, der1st_sym) || !wide_ok(b, der2nd) || !wide_ok(b, der2nd_sym))
        return 0;
    if (der1st->tl_hash != der1st_sym->tl_hash || der2nd->tl_hash != der2nd_sym->tl_hash) return 0;
    const int np = b->ny * b->nz;
    static int uni_on = -1;  // X3D_NO_UNIFORM=1: the general tables on uniform grids too (A/B, as for the tile kernels)
    if (uni_on < 0) { const char *e = getenv("X3D_NO_UNIFORM"); uni_on = (e && e[0] == '1') ? 0 : 1; }
    const bool uni = uni_on && der1st->uniform && der1st_sym->uniform && der2nd->uniform && der2nd_sym->uniform;
    const bool narrow = wide_narrow(der1st) && wide_narrow(der2nd);
    const bool circ = uni && narrow && wide_circ(der1st) && wide_circ(der1st_sym) && wide_circ(der2nd) && wide_circ(der2nd_sym);
    const size_t lds = sizeof(real_t) * ((circ ? 0 : 2 * (uni ? LTU_N : LTC_N(WQ))) + 8 * WSTRIP);
    const int blocks = x3d_persistent_blocks(b, (np + 7) / 8);
    const Circ4 c4{{der1st->circ, der2nd->circ, der1st->circ, der1st->circ}};
    {
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X);
#define GOC(A_, N_, R_, U_, C_)                                                                                 \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_xwide_transeq3<A_, N_, R_, U_, C_>));                                               \
        hipLaunchKernelGGL((k_xwide_transeq3<A_, N_, R_, U_, C_>), dim3(blocks), dim3(512), lds, b->stream, r[0], r[1], r[2], \
                           f[0], f[1], f[2], wide_xop(der1st), wide_xop(der2nd), np, (long)b->nxp, nu, omega, ushift, c4); \
    } while (0)
#define GO(A_, N_, R_, U_) GOC(A_, N_, R_, U_, false)
#define GOU(A_, N_, R_) do { if (circ && (N_)) GOC(A_, true, R_, true, true); else if (uni) GO(A_, N_, R_, true); else GO(A_, N_, R_, false); } while (0)
        if (omega != 0.0) { if (narrow) GOU(false, true, true); else GOU(false, false, true); }
        else if (acc) { if (narrow) GOU(true, true, false); else GOU(true, false, false); }
        else { if (narrow) GOU(false, true, false); else GOU(false, false, false); }
#undef GOU
#undef GO
#undef GOC
    }
    X3D_HIP(hipGetLastError());
    b->n_tq3++;
    if (b->prof) {  // three components (bench.py divides the direction's time by the count)
        for (int k = 0; k < 2; k++) { ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X); }
    }
    *done = true;
    return 0;
}

// transeq_x with the pending velocity correction f[c] += scale * tds_solve(g[c]) (op_s for c = 0, op_i for c = 1, 2)
// applied first, inside the kernel; omega != 0: the rotation forcing on top; ushift: f[0] += *ushift as well
int x3d_xwide_transeq3_upd(x3d_backend *b, real_t *const r[3], real_t *const f[3], real_t nu, const x3d_tdsops *der1st,
                           const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                           const real_t *const g[3], const x3d_tdsops *op_s, const x3d_tdsops *op_i, real_t scale,
                           real_t omega, const real_t *ushift, bool *done)
{
    *done = false;
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_XWIDE_UPD"); on = (e && e[0] == '1') ? 0 : 1; }
    const x3d_tdsops *ops[6] = {der1st, der1st_sym, der2nd, der2nd_sym, op_s, op_i};
    if (!on || !wide_env_on()) return 0;
    for (const x3d_tdsops *t : ops)
        if (!wide_ok(b, t) || !t->uniform) return 0;
    if (der1st->tl_hash != der1st_sym->tl_hash || der2nd->tl_hash != der2nd_sym->tl_hash) return 0;
    const int np = b->ny * b->nz;
    const bool narrow = wide_narrow(der1st) && wide_narrow(der2nd) && wide_narrow(op_s) && wide_narrow(op_i);
    bool circ = narrow;
    for (const x3d_tdsops *t : ops) circ = circ && wide_circ(t);
    const size_t lds = sizeof(real_t) * ((circ ? 0 : 4 * LTU_N) + 8 * WSTRIP);
    if (lds > 160 * 1024) return 0;
    const int blocks = x3d_persistent_blocks(b, (np + 7) / 8);
    WideUpd wu{{g[0], g[1], g[2]}, scale, omega, ushift};
    const Circ4 c4{{der1st->circ, der2nd->circ, op_s->circ, op_i->circ}};
    {
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X);
#define GO(N_, R_, C_)                                                                                          \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_xwide_transeq3_upd<N_, R_, C_>));                                                   \
        hipLaunchKernelGGL((k_xwide_transeq3_upd<N_, R_, C_>), dim3(blocks), dim3(512), lds, b->stream, r[0], r[1], r[2], \
                           f[0], f[1], f[2], wide_xop(der1st), wide_xop(der2nd), wide_xop(op_s), wide_xop(op_i), wu, np, \
                           (long)b->nxp, nu, c4);                                                               \
    } while (0)
        if (omega != 0.0) { if (circ) GO(true, true, true); else if (narrow) GO(true, true, false); else GO(false, true, false); }
        else { if (circ) GO(true, false, true); else if (narrow) GO(true, false, false); else GO(false, false, false); }
#undef GO
    }
    X3D_HIP(hipGetLastError());
    b->n_tq3++;
    b->n_upd++;
    if (b->prof) {
        for (int k = 0; k < 2; k++) { ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X); }
    }
    *done = true;
    return 0;
}
