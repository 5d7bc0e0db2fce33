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

// FAST transeq kernel: one workgroup per CU (129 KB of lane tables).  Measured: 12 waves with next-pencil
// prefetch (163 VGPRs) 0.96 ms per component, 16 waves without it (123 VGPRs) 0.90 ms.
#ifndef YT_NOPREF
#define YT_PREF 1  // K3y: next tile's u rows prefetched into registers (same-box A/B: 1.05 -> 0.92 ms per component)
#endif
#ifndef XS_TQ_THREADS
#define XS_TQ_THREADS 1024
#ifndef XS_PREF
#define XS_NOPREF 1
#endif
#endif

// nontemporal loads / stores of the rhs rows in the three-in-one tile kernel (-DYT_NO_NT: plain): same-box A/B
// k_ytile_transeq3 z 2.21 -> 2.16 ms, y 2.24 -> 2.23; full step at 512^3 within noise, at 256^3 5.52 -> 5.44 ms
#ifndef YT_NO_NT
#define YT_NT 1
#endif

#ifndef XSCAN_EXP
#define XSCAN_EXP 0  // timing experiments only (1: no stores, 2: no loads, 3: no scans)
#endif

int npmax_of(const x3d_backend *b);

#include "xscan_core.h"
#include "zfft_tile.h"

// HALO forms of the tile kernels (K3y): the reduced 2 x 2 systems of a decomposed direction couple to the
// NEIGHBOUR ranks' boundary values (src/backend/omp/kernels/distributed.f90:186-206), which do not exist yet
// when this rank's sweep runs.  The kernel closes every operator with recv_s = recv_e = 0, stores its own
// boundary values du_1 / X_n for the exchange ([side][op][pencil]), and k_*_halo_fix adds the terms linear in
// the received values afterwards -- on the boundary strips only: dist_sa / dist_sc decay by 0.15 - 0.38 per row
// (src/tdsops.f90:196-201 relies on the same decay), so beyond ~45 rows they are below 2^-60.

// ---- stores: a lane owns Q = 8 consecutive rows (64 B).  Storing them as four 16-byte pieces makes
// every store instruction touch 16 B of each 64-byte sector (measured: the kernel then runs at
// 3.2 TB/s); a 4x4 transpose of the 16-byte pairs inside each quad of lanes lets every instruction
// write whole 64-byte sectors instead (5.0 TB/s).  T_j[m] = P_m[j]: rotate left by j, quad_perm
// DPP by the register index, rotate left by j again, read out reversed.
// out row = orow[0 .. 64*8): lane l holds r[0..8) = rows 8l .. 8l+7
template <bool ACC>
__device__ __forceinline__ void store_rows_q8(real_t *__restrict__ orow, int lane, const real_t (&r)[8],
                                              real_t scale)
{
    const int j = lane & 3;
    real_t a0 = r[0], a1 = r[1], b0 = r[2], b1 = r[3], c0 = r[4], c1 = r[5], d0 = r[6], d1 = r[7];
    rotl_pairs(a0, a1, b0, b1, c0, c1, d0, d1, j);
    b0 = dpp_quad(b0, 1); b1 = dpp_quad(b1, 1);
    c0 = dpp_quad(c0, 2); c1 = dpp_quad(c1, 2);
    d0 = dpp_quad(d0, 3); d1 = dpp_quad(d1, 3);
    rotl_pairs(a0, a1, b0, b1, c0, c1, d0, d1, j);
    real2_t *__restrict__ o2 = reinterpret_cast<real2_t *>(orow + (lane & ~3) * 8) + j;
    // instruction m stores T[m] = D[(-m) & 3]: D[0] = a, D[3] = d, D[2] = c, D[1] = b
    real2_t v0, v1, v2, v3;
    if (ACC) {
        v0 = o2[0]; v1 = o2[4]; v2 = o2[8]; v3 = o2[12];
        v0.x += scale * a0; v0.y += scale * a1;
        v1.x += scale * d0; v1.y += scale * d1;
        v2.x += scale * c0; v2.y += scale * c1;
        v3.x += scale * b0; v3.y += scale * b1;
    } else {
        v0.x = a0; v0.y = a1; v1.x = d0; v1.y = d1; v2.x = c0; v2.y = c1; v3.x = b0; v3.y = b1;
    }
    // (plain stores: an instruction writes whole 64-byte sectors but only half of each 128-byte line)
#ifdef XS_NT_STORES  // experiment: nontemporal stores only (the loads stay cached)
    stg_stream(o2, v0); stg_stream(o2 + 4, v1); stg_stream(o2 + 8, v2); stg_stream(o2 + 12, v3);
#else
    o2[0] = v0; o2[4] = v1; o2[8] = v2; o2[12] = v3;
#endif
}

// Q = 4: a lane owns 32 B, a pair of lanes one 64-byte sector: 2 x 2 transpose of the 16-byte pairs
template <bool ACC>
__device__ __forceinline__ void store_rows_q4(real_t *__restrict__ orow, int lane, const real_t (&r)[4],
                                              real_t scale)
{
    const bool odd = lane & 1;
    const real_t s0 = odd ? r[0] : r[2], s1 = odd ? r[1] : r[3];  // what the partner needs
    int lo, hi;
    lo = __double2loint(s0); hi = __double2hiint(s0);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false);
    const real_t g0 = __hiloint2double(hi, lo);
    lo = __double2loint(s1); hi = __double2hiint(s1);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false);
    const real_t g1 = __hiloint2double(hi, lo);
    // instruction m writes bytes [32 m + 16 p, +16) of the pair's sector: pair p of lane m of the pair
    const real_t a0 = odd ? g0 : r[0], a1 = odd ? g1 : r[1];  // m = 0: P_0[p]
    const real_t b0 = odd ? r[2] : g0, b1 = odd ? r[3] : g1;  // m = 1: P_1[p]
    real2_t *__restrict__ o2 = reinterpret_cast<real2_t *>(orow + (lane & ~1) * 4) + (lane & 1);
    real2_t v0, v1;
    if (ACC) {
        v0 = o2[0]; v1 = o2[2];
        v0.x += scale * a0; v0.y += scale * a1; v1.x += scale * b0; v1.y += scale * b1;
    } else {
        v0.x = a0; v0.y = a1; v1.x = b0; v1.y = b1;
    }
    o2[0] = v0; o2[2] = v1;
}

template <int Q>
__device__ __forceinline__ void load_window(real_t (&w)[Q + 8], const real_t *__restrict__ row, int first, int nr,
                                            int n_wrap, bool interior)
{
    if (interior) {  // rows first-4 .. first+Q+3 all inside [1, nr]: 16-byte aligned vector loads
        const real2_t *__restrict__ v2 = reinterpret_cast<const real2_t *>(row + first - 5);
#pragma unroll
        for (int m = 0; m < (Q + 8) / 2; m++) {
            const real2_t t2 = v2[m];
            w[2 * m] = t2.x;
            w[2 * m + 1] = t2.y;
        }
    } else {
#pragma unroll
        for (int m = 0; m < Q + 8; m++) w[m] = ext_x(row, first - 4 + m, nr, n_wrap);
    }
}

// ---------------------------------------------------------------- tds_solve
// FAST: 0 general, 1 branch-free periodic form, 2 the same with the 5-tap stencil, 3 (round 6): 2 on a uniform grid in the
// circulant form (circ_solve, co: no lane tables, no LDS at all)
template <int Q, bool ACC, int FAST>
__global__ void __launch_bounds__(512) k_xscan_tds(real_t *__restrict__ du, const real_t *__restrict__ u, XOp t,
                                                   int np, long pitch, int n_wrap, real_t scale, CircOp co)
{
    extern __shared__ real_t lt[];  // [LT_N(Q)][64], general form: + the stencil table [CS_N(Q)]
    constexpr bool CIRC = FAST == 3;
    const real_t *cs = lt + LT_N(Q) * 64;
    if constexpr (!CIRC) {
        for (int i = threadIdx.x; i < LT_N(Q) * 64; i += blockDim.x) lt[i] = t.TL[i];
        if (!FAST) stage_cs<Q>(lt + LT_N(Q) * 64, t);
        __syncthreads();
    }
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int n = t.n_tds, nr = t.n_rhs;
    const int first = lane * Q + 1;
    const bool interior = first - 4 >= 1 && first + Q + 3 <= nr;
    const bool exact = FAST || (nr == 64 * Q && n_wrap == nr);
    const int p0 = blockIdx.x * (blockDim.x >> 6) + wave;
    real_t nb[Q];  // FAST: next pencil's rows, in flight while this one is solved
    if (FAST && p0 < np) load_body<Q>(nb, u + (long)p0 * pitch, lane);
    for (int p = p0; p < np; p += nwaves) {
        const real_t *__restrict__ row = u + (long)p * pitch;
        asm volatile("" : "+v"(lane));  // keep the lane-table reads inside the loop (no 150-VGPR hoist)
        real_t w[Q + 8], X[Q], du1, xn;
        if (FAST) {
            window_from_body<Q>(w, nb, lane);
#if XSCAN_EXP == 2
            for (int q = 0; q < Q; q++) nb[q] = nb[q] * 1.0000001 + lane;
#else
            if (p + nwaves < np) load_body<Q>(nb, u + (long)(p + nwaves) * pitch, lane);
#endif
        } else if (exact) load_window_exact<Q>(w, row, lane, nr);
        else load_window<Q>(w, row, first, nr, n_wrap, interior);
        real_t *__restrict__ orow = du + (long)p * pitch;
        real_t r[Q];
        if constexpr (CIRC) {
            circ_solve<Q, true>(w, r, co, lane);
        } else {
        scan_solve<Q, (FAST != 0), (FAST == 2)>(w, X, du1, xn, lt, t, lane, first, 0, cs);
        const real_t du_s = t.rs_s * (du1 - t.sa1 * xn);  // periodic self-exchange: recv_s = X_n
        const real_t du_e = t.rs_e * (xn - t.scn * du1);  //                          recv_e = du_1
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = first + q;
            const real_t st = LTR(lt, LT_ST(q));
            r[q] = (X[q] - LTR(lt, LT_SA(q)) * du_s - LTR(lt, LT_SC(q)) * du_e) * st;
            if (FAST) {  // n = 64 Q: row 1 is (lane 0, q = 0), row n is (lane 63, q = Q - 1)
                if (q == 0) r[q] = (lane == 0) ? du_s * st : r[q];
                if (q == Q - 1) r[q] = (lane == 63) ? du_e * st : r[q];
            } else {
                r[q] = (j == 1) ? du_s * st : r[q];
                r[q] = (j == n) ? du_e * st : r[q];
            }
        }
        }  // (!CIRC)
#if XSCAN_EXP == 1
        if (r[0] != 12345.678) continue;
#endif
        if constexpr (FAST != 0) {
            if constexpr (Q == 8) store_rows_q8<ACC>(orow, lane, r, scale);
            else store_rows_q4<ACC>(orow, lane, r, scale);
        } else if (exact && n == nr) {
            real2_t *__restrict__ o2 = reinterpret_cast<real2_t *>(orow + lane * Q);
#pragma unroll
            for (int m = 0; m < Q / 2; m++) {
                real2_t v2;
                if (ACC) { v2 = o2[m]; v2.x += scale * r[2 * m]; v2.y += scale * r[2 * m + 1]; }
                else { v2.x = r[2 * m]; v2.y = r[2 * m + 1]; }
                o2[m] = v2;
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = first + q;
                if (j <= n) orow[j - 1] = ACC ? orow[j - 1] + scale * r[q] : r[q];
            }
        }
    }
}

// ---------------------------------------------------------------- tds_solve of a field that is still to be formed
// The first x operators of the pressure correction act on the velocity the RK / AB stage has just produced
// (y = base + sum c_k x_k, src/time_integrator.f90:166-282 -> divergence_v2c, src/vector_calculus.f90:160-175).
// Here the stage's linear combination is the kernel's prologue: y is formed in registers (summation order of
// k_lincomb, bit-identical), stored, and solved at once -- y is not read back (one field pass less per variable).
// lr.wall != null: the pencils of the two y faces (j = 0, ny - 1) take the rows of `wall` instead
// (field_set_face_from_field(Y_FACE) between the stage and the divergence: the channel case's apply_BC,
// src/case/channel.f90:214-231)
// CIRC: the operator in the circulant form (as k_xscan_tds<.., 3>: the two kernels must give the same bits)
template <int Q, bool NARROW, bool CIRC = false>
__global__ void __launch_bounds__(512) k_xscan_tds_lin(real_t *__restrict__ du, LinRows lr, XOp t, int np, long pitch,
                                                       CircOp co)
{
    extern __shared__ real_t lt[];  // [LT_N(Q)][64]
    if constexpr (!CIRC) {
        for (int i = threadIdx.x; i < LT_N(Q) * 64; i += blockDim.x) lt[i] = t.TL[i];
        __syncthreads();
    }
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int first = lane * Q + 1;
    for (int p = blockIdx.x * (blockDim.x >> 6) + wave; p < np; p += nwaves) {
        const long ro = (long)p * pitch;
        asm volatile("" : "+v"(lane));  // keep the lane-table reads inside the loop
        real_t b[Q];
        load_body<Q>(b, lr.base + ro, lane);
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (k < lr.n) {
                real_t xk[Q];
                load_body<Q>(xk, lr.x[k] + ro, lane);
#pragma unroll
                for (int q = 0; q < Q; q++) b[q] = lr.c[k] * xk[q] + b[q];
            }
        if (lr.wall) {
            const int j = p % lr.ny;
            if (j == 0 || j == lr.ny - 1) load_body<Q>(b, lr.wall + ro, lane);
        }
        if constexpr (Q == 8) store_rows_q8<false>(lr.y + ro, lane, b, 1.0);
        else store_rows_q4<false>(lr.y + ro, lane, b, 1.0);
        real_t w[Q + 8], X[Q], du1, xn;
        window_from_body<Q>(w, b, lane);
        real_t r[Q];
        if constexpr (CIRC) {
            circ_solve<Q, NARROW>(w, r, co, lane);
        } else {
        scan_solve<Q, true, NARROW>(w, X, du1, xn, lt, t, lane, first);
        const real_t du_s = t.rs_s * (du1 - t.sa1 * xn), du_e = t.rs_e * (xn - t.scn * du1);
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const real_t st = LTR(lt, LT_ST(q));
            r[q] = (X[q] - LTR(lt, LT_SA(q)) * du_s - LTR(lt, LT_SC(q)) * du_e) * st;
            if (q == 0) r[q] = (lane == 0) ? du_s * st : r[q];
            if (q == Q - 1) r[q] = (lane == 63) ? du_e * st : r[q];
        }
        }
        if constexpr (Q == 8) store_rows_q8<false>(du + ro, lane, r, 1.0);
        else store_rows_q4<false>(du + ro, lane, r, 1.0);
    }
}

// ---------------------------------------------------------------- transeq component
template <int Q, bool SAME, bool ACC, int FAST>
__global__ void __launch_bounds__(FAST ? XS_TQ_THREADS : 512)
    k_xscan_transeq(real_t *__restrict__ rhs, const real_t *__restrict__ u, const real_t *__restrict__ cv, XOp t1,
                    XOp t2, XOp t3, int np, long pitch, real_t nu)
{
    extern __shared__ real_t lt[];  // three operators: [3][LT_N(Q)][64], general form: + their stencil tables [3][CS_N(Q)]
    constexpr int LN = LT_N(Q) * 64;
    for (int i = threadIdx.x; i < LN; i += blockDim.x) {
        lt[i] = t1.TL[i];
        lt[LN + i] = t2.TL[i];
        lt[2 * LN + i] = t3.TL[i];
    }
    const real_t *cs0 = lt + 3 * LN;
    if (!FAST) {
        stage_cs<Q>(lt + 3 * LN, t1);
        stage_cs<Q>(lt + 3 * LN + CS_N(Q), t2);
        stage_cs<Q>(lt + 3 * LN + 2 * CS_N(Q), t3);
    }
    __syncthreads();
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int n = t1.n_tds;
    const int first = lane * Q + 1;
    const bool interior = first - 4 >= 1 && first + Q + 3 <= n;
    const real_t *__restrict__ l1 = lt, *__restrict__ l2 = lt + LN, *__restrict__ l3 = lt + 2 * LN;
    const int p0 = blockIdx.x * (blockDim.x >> 6) + wave;
    real_t nbu[Q], nbc[Q];  // FAST: next pencil's rows of u and conv, in flight during the solve
#ifndef XS_NOPREF
    if (FAST && p0 < np) {
        load_body<Q>(nbu, u + (long)p0 * pitch, lane);
        if (!SAME) load_body<Q>(nbc, cv + (long)p0 * pitch, lane);
    }
#endif
    for (int p = p0; p < np; p += nwaves) {
        const real_t *__restrict__ ru = u + (long)p * pitch;
        const real_t *__restrict__ rc = cv + (long)p * pitch;
        asm volatile("" : "+v"(lane));  // keep the lane-table reads inside the loop
        real_t wu[Q + 8], wp[Q + 8], vq[Q];
        const bool exact = FAST || n == 64 * Q;
        if (FAST) {
#ifdef XS_NOPREF
            load_body<Q>(nbu, u + (long)p * pitch, lane);
            if (!SAME) load_body<Q>(nbc, cv + (long)p * pitch, lane);
#endif
            window_from_body<Q>(wu, nbu, lane);
            if (!SAME) {
                window_from_body<Q>(wp, nbc, lane);
#pragma unroll
                for (int q = 0; q < Q; q++) vq[q] = nbc[q];
            } else {
#pragma unroll
                for (int q = 0; q < Q; q++) vq[q] = nbu[q];
            }
#ifndef XS_NOPREF
            if (p + nwaves < np) {
                load_body<Q>(nbu, u + (long)(p + nwaves) * pitch, lane);
                if (!SAME) load_body<Q>(nbc, cv + (long)(p + nwaves) * pitch, lane);
            }
#endif
#pragma unroll
            for (int m = 0; m < Q + 8; m++) wp[m] = SAME ? wu[m] * wu[m] : wu[m] * wp[m];
        } else {
        if (exact) load_window_exact<Q>(wu, ru, lane, n);
        else load_window<Q>(wu, ru, first, n, n, interior);
        if (SAME) {
#pragma unroll
            for (int m = 0; m < Q + 8; m++) wp[m] = wu[m] * wu[m];
#pragma unroll
            for (int q = 0; q < Q; q++) vq[q] = wu[q + 4];
        } else {
            if (exact) load_window_exact<Q>(wp, rc, lane, n);
            else load_window<Q>(wp, rc, first, n, n, interior);
#pragma unroll
            for (int q = 0; q < Q; q++) vq[q] = wp[q + 4];
#pragma unroll
            for (int m = 0; m < Q + 8; m++) wp[m] = wu[m] * wp[m];  // ud = u*conv incl. halo products
        }
        }
        // one operator at a time, substituted at once (distributed.f90:304-335 written per operator:
        // rows 1 and n take du_s*st / du_e*st, which is what the general formula gives with these temps)
        auto solve_subs = [&](const real_t (&w)[Q + 8], real_t (&T)[Q], const real_t *__restrict__ l, const XOp &t) {
            real_t a, b;
            scan_solve<Q, (FAST != 0), (FAST == 2)>(w, T, a, b, l, t, lane, first, 0, cs0 + (l - lt) / LN * CS_N(Q));
            const real_t s_ = t.rs_s * (a - t.sa1 * b), e_ = t.rs_e * (b - t.scn * a);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = first + q;
                const real_t st = LTR(l, LT_ST(q));
                real_t x = st * (T[q] - LTR(l, LT_SA(q)) * s_ - LTR(l, LT_SC(q)) * e_);
                if (FAST) {  // n = 64 Q: row 1 is (lane 0, q = 0), row n is (lane 63, q = Q - 1)
                    if (q == 0) x = (lane == 0) ? s_ * st : x;
                    if (q == Q - 1) x = (lane == 63) ? e_ * st : x;
                } else {
                    x = (j == 1) ? s_ * st : x;
                    x = (j == n) ? e_ * st : x;
                }
                T[q] = x;
            }
        };
        real_t r[Q], T[Q];
        solve_subs(wp, T, l2, t2);  // d(u*conv)/dx first: wp is dead afterwards
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] = T[q];
        asm volatile("" : "+v"(lane) : "v"(r[0]));  // order the operators: bounds the live lane-table reads
        solve_subs(wu, T, l1, t1);  // du/dx
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] = -0.5 * (vq[q] * T[q] + r[q]) + nu * (T[q] * LTR(l3, LT_STC(q)));
        asm volatile("" : "+v"(lane) : "v"(r[0]));
        solve_subs(wu, T, l3, t3);  // d2u/dx2
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] += nu * T[q];
        real_t *__restrict__ orow = rhs + (long)p * pitch;
        if constexpr (FAST != 0) {
            if constexpr (Q == 8) store_rows_q8<ACC>(orow, lane, r, 1.0);
            else store_rows_q4<ACC>(orow, lane, r, 1.0);
        } else if (exact) {
            real2_t *__restrict__ o2 = reinterpret_cast<real2_t *>(orow + lane * Q);
#pragma unroll
            for (int m = 0; m < Q / 2; m++) {
                real2_t v2;
                if (ACC) { v2 = o2[m]; v2.x += r[2 * m]; v2.y += r[2 * m + 1]; }
                else { v2.x = r[2 * m]; v2.y = r[2 * m + 1]; }
                o2[m] = v2;
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = first + q;
                if (j <= n) orow[j - 1] = ACC ? orow[j - 1] + r[q] : r[q];
            }
        }
    }
}

// ---------------------------------------------------------------- transeq component, two pencils per wave
// FAST form only (periodic-type operators, n = 64 Q).  8 waves x 2 pencils per workgroup: the same 16 pencils
// in flight per CU as k_xscan_transeq, half the LDS table traffic.
template <int Q, bool SAME, bool ACC, bool NARROW>
__global__ void __launch_bounds__(512)
    k_xscan_transeq2(real_t *__restrict__ rhs, const real_t *__restrict__ u, const real_t *__restrict__ cv, XOp t1,
                     XOp t2, XOp t3, int np, long pitch, real_t nu)
{
    extern __shared__ real_t lt[];  // three operators: [3][LT_N(Q)][64]
    constexpr int LN = LT_N(Q) * 64;
    for (int i = threadIdx.x; i < LN; i += blockDim.x) {
        lt[i] = t1.TL[i];
        lt[LN + i] = t2.TL[i];
        lt[2 * LN + i] = t3.TL[i];
    }
    __syncthreads();
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int first = lane * Q + 1;
    const real_t *__restrict__ l1 = lt, *__restrict__ l2 = lt + LN, *__restrict__ l3 = lt + 2 * LN;
    // next pair's rows of u and conv, in flight during the solve (256-VGPR budget: 8 waves per CU)
    real_t nua[Q], nub[Q], nca[Q], ncb[Q];
    const int pstart = 2 * (blockIdx.x * (blockDim.x >> 6) + wave);
    if (pstart < np) {
        load_body<Q>(nua, u + (long)pstart * pitch, lane);
        load_body<Q>(nub, u + (long)(pstart + 1) * pitch, lane);
        if (!SAME) {
            load_body<Q>(nca, cv + (long)pstart * pitch, lane);
            load_body<Q>(ncb, cv + (long)(pstart + 1) * pitch, lane);
        }
    }
    for (int p = pstart; p < np; p += 2 * nwaves) {
        asm volatile("" : "+v"(lane));  // keep the lane-table reads inside the loop
        V2 wu[Q + 8], wp[Q + 8], vq[Q];
        {
            V2 b2[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) b2[q] = V2{nua[q], nub[q]};
            window_from_body<Q, V2>(wu, b2, lane);
            if (!SAME) {
#pragma unroll
                for (int q = 0; q < Q; q++) b2[q] = V2{nca[q], ncb[q]};
                window_from_body<Q, V2>(wp, b2, lane);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) vq[q] = b2[q];
            const int pn = p + 2 * nwaves;
            if (pn < np) {
                load_body<Q>(nua, u + (long)pn * pitch, lane);
                load_body<Q>(nub, u + (long)(pn + 1) * pitch, lane);
                if (!SAME) {
                    load_body<Q>(nca, cv + (long)pn * pitch, lane);
                    load_body<Q>(ncb, cv + (long)(pn + 1) * pitch, lane);
                }
            }
#pragma unroll
            for (int m = 0; m < Q + 8; m++) wp[m] = SAME ? wu[m] * wu[m] : wu[m] * wp[m];
        }
        auto solve_subs = [&](const V2 (&w)[Q + 8], V2 (&T)[Q], const real_t *__restrict__ l, const XOp &t) {
            V2 a, b;
            scan_solve<Q, true, NARROW, V2>(w, T, a, b, l, t, lane, first);
            const V2 s_ = t.rs_s * (a - t.sa1 * b), e_ = t.rs_e * (b - t.scn * a);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const real_t st = LTR(l, LT_ST(q));
                V2 x = st * (T[q] - LTR(l, LT_SA(q)) * s_ - LTR(l, LT_SC(q)) * e_);
                if (q == 0) x = (lane == 0) ? s_ * st : x;
                if (q == Q - 1) x = (lane == 63) ? e_ * st : x;
                T[q] = x;
            }
        };
        V2 r[Q], T[Q];
        solve_subs(wp, T, l2, t2);
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] = T[q];
        asm volatile("" : "+v"(lane) : "v"(r[0].a));
        solve_subs(wu, T, l1, t1);
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] = -0.5 * (vq[q] * T[q] + r[q]) + nu * (T[q] * LTR(l3, LT_STC(q)));
        asm volatile("" : "+v"(lane) : "v"(r[0].a));
        solve_subs(wu, T, l3, t3);
        real_t ra[Q], rb[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const V2 v = r[q] + nu * T[q];
            ra[q] = v.a;
            rb[q] = v.b;
        }
        real_t *__restrict__ oa = rhs + (long)p * pitch, *__restrict__ ob = oa + pitch;
        if constexpr (Q == 8) { store_rows_q8<ACC>(oa, lane, ra, 1.0); store_rows_q8<ACC>(ob, lane, rb, 1.0); }
        else { store_rows_q4<ACC>(oa, lane, ra, 1.0); store_rows_q4<ACC>(ob, lane, rb, 1.0); }
    }
}

// ---------------------------------------------------------------- transeq_x: the three components at once
// Two pencils per wave as above; the pair's rows of the advecting velocity u0 stay in registers for the three
// components (u0, u0), (u1, u0), (u2, u0): 6 field passes instead of 8.  der1st == der1st_sym and
// der2nd == der2nd_sym as lane tables (periodic operators): two table sets in LDS.
//
// UPD: the velocity is still waiting for the previous sub-step's pressure-gradient correction
// u_c += scale * tds_solve(g_c)  (gradient_c2v's last x operators, src/vector_calculus.f90:318-330 +
// src/solver.f90:731-733): done here, per pencil, before the component is used -- u_c is written, not read back
// (12 field passes instead of 9 + 6).  Four table sets, three of them without the STC block: 156 KB of LDS.
struct XUpd {
    const real_t *g[3];  // gradient inputs of u0, u1, u2
    real_t scale;
    // the channel case's extras (xwide.hip, k_xwide_transeq3<ROT>, has the 1024-row form): rotation forcing
    // rhs0 -= omega u1, rhs1 += omega u0 on top of the result; u0 += *ushift in place first
    real_t omega;
    const real_t *ushift;
};

// CIRC (round 6; uniform grid, NARROW): every operator in the circulant form (circ_solve; cc[0..3] = der1st, der2nd, op_s,
// op_i) -- no lane tables, no LDS
template <int Q, bool ACC, bool NARROW, bool UPD, bool CHN = false, bool CIRC = false>  // CHN: the channel case's extras (XUpd)
__global__ void __launch_bounds__(512)
    k_xscan_transeq2x3(real_t *__restrict__ rhs0, real_t *__restrict__ rhs1, real_t *__restrict__ rhs2, real_t *u0,
                       real_t *u1, real_t *u2, XOp tD1, XOp tD2, int np, long pitch, real_t nu, XUpd upd, XOp tS,
                       XOp tI, Circ4 cc)
{
    extern __shared__ real_t lt[];
    constexpr int LNF = LT_N(Q) * 64, LNC = LT_NC(Q) * 64, L1N = UPD ? LNC : LNF;
    static_assert(!CIRC || NARROW, "CIRC: 5-tap stencils");
    if constexpr (!CIRC) {
        for (int i = threadIdx.x; i < L1N; i += blockDim.x) lt[i] = tD1.TL[i];
        for (int i = threadIdx.x; i < LNF; i += blockDim.x) lt[L1N + i] = tD2.TL[i];
        if (UPD) {
            for (int i = threadIdx.x; i < LNC; i += blockDim.x) {
                lt[L1N + LNF + i] = tS.TL[i];
                lt[L1N + LNF + LNC + i] = tI.TL[i];
            }
        }
        __syncthreads();
    }
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int first = lane * Q + 1;
    const real_t *__restrict__ l1 = lt, *__restrict__ l3 = lt + L1N;
    real_t na[Q], nb[Q];  // the rows needed next (next component's field, or the next pair's u0)
    const int pstart = 2 * (blockIdx.x * (blockDim.x >> 6) + wave);
    if (pstart < np) {
        load_body<Q>(na, u0 + (long)pstart * pitch, lane);
        load_body<Q>(nb, u0 + (long)(pstart + 1) * pitch, lane);
    }
    for (int p = pstart; p < np; p += 2 * nwaves) {
        V2 cb[Q];
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            asm volatile("" : "+v"(lane));  // keep the lane-table reads inside the loops
            V2 wu[Q + 8], wp[Q + 8];
            {
                V2 b2[Q];
#pragma unroll
                for (int q = 0; q < Q; q++) b2[q] = V2{na[q], nb[q]};
                const int pn = p + 2 * nwaves;
                {
                    const real_t *nsrc = c == 0 ? u1 + (long)p * pitch : (c == 1 ? u2 + (long)p * pitch
                                                                                 : u0 + (long)(pn < np ? pn : p) * pitch);
                    if (c < 2 || pn < np) {
                        load_body<Q>(na, nsrc, lane);
                        load_body<Q>(nb, nsrc + pitch, lane);
                    }
                }
                if constexpr (UPD) {
                    // u_c += scale * tds_solve(g_c), arithmetic of k_xscan_tds<ACC>: old + scale * r
                    const real_t *gsrc = (c == 0 ? upd.g[0] : (c == 1 ? upd.g[1] : upd.g[2])) + (long)p * pitch;
                    real_t ga[Q], gb[Q];
                    load_body<Q>(ga, gsrc, lane);
                    load_body<Q>(gb, gsrc + pitch, lane);
                    V2 g2[Q], wg[Q + 8], X[Q], du1, xn;
#pragma unroll
                    for (int q = 0; q < Q; q++) g2[q] = V2{ga[q], gb[q]};
                    window_from_body<Q, V2>(wg, g2, lane);
                    if constexpr (CIRC) {
                        if (c == 0) circ_solve<Q, NARROW, V2>(wg, X, cc.o[2], lane);
                        else circ_solve<Q, NARROW, V2>(wg, X, cc.o[3], lane);
#pragma unroll
                        for (int q = 0; q < Q; q++) b2[q] = b2[q] + upd.scale * X[q];
                    } else {
                    const real_t *__restrict__ lg = lt + L1N + LNF + (c == 0 ? 0 : LNC);
                    const XOp &tg = c == 0 ? tS : tI;
                    scan_solve<Q, true, NARROW, V2>(wg, X, du1, xn, lg, tg, lane, first);
                    const V2 du_s = tg.rs_s * (du1 - tg.sa1 * xn), du_e = tg.rs_e * (xn - tg.scn * du1);
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const real_t st = LTR(lg, LT_ST(q));
                        V2 r = (X[q] - LTR(lg, LT_SA(q)) * du_s - LTR(lg, LT_SC(q)) * du_e) * st;
                        if (q == 0) r = (lane == 0) ? du_s * st : r;
                        if (q == Q - 1) r = (lane == 63) ? du_e * st : r;
                        b2[q] = b2[q] + upd.scale * r;
                    }
                    }
                    real_t ua[Q], ub[Q];
#pragma unroll
                    for (int q = 0; q < Q; q++) { ua[q] = b2[q].a; ub[q] = b2[q].b; }
                    real_t *uw = (c == 0 ? u0 : (c == 1 ? u1 : u2)) + (long)p * pitch;
                    if constexpr (Q == 8) { store_rows_q8<false>(uw, lane, ua, 1.0); store_rows_q8<false>(uw + pitch, lane, ub, 1.0); }
                    else { store_rows_q4<false>(uw, lane, ua, 1.0); store_rows_q4<false>(uw + pitch, lane, ub, 1.0); }
                    asm volatile("" : "+v"(lane) : "v"(b2[0].a));
                }
                if (CHN && c == 0 && upd.ushift) {  // (wave-uniform; never together with UPD)
                    const real_t ush = *upd.ushift;
                    real_t ua[Q], ub[Q];
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        b2[q].a += ush;
                        b2[q].b += ush;
                        ua[q] = b2[q].a;
                        ub[q] = b2[q].b;
                    }
                    real_t *uw = u0 + (long)p * pitch;
                    if constexpr (Q == 8) { store_rows_q8<false>(uw, lane, ua, 1.0); store_rows_q8<false>(uw + pitch, lane, ub, 1.0); }
                    else { store_rows_q4<false>(uw, lane, ua, 1.0); store_rows_q4<false>(uw + pitch, lane, ub, 1.0); }
                }
                if (c == 0) {
#pragma unroll
                    for (int q = 0; q < Q; q++) cb[q] = b2[q];
                }
                window_from_body<Q, V2>(wu, b2, lane);
                window_from_body<Q, V2>(wp, cb, lane);
#pragma unroll
                for (int m = 0; m < Q + 8; m++) wp[m] = wu[m] * wp[m];
            }
            auto solve_subs = [&](const V2 (&w)[Q + 8], V2 (&T)[Q], const real_t *__restrict__ l, const XOp &t) {
                if constexpr (CIRC) {
                    if (l == l1) circ_solve<Q, NARROW, V2>(w, T, cc.o[0], lane);
                    else circ_solve<Q, NARROW, V2>(w, T, cc.o[1], lane);
                    return;
                }
                V2 a, b;
                scan_solve<Q, true, NARROW, V2>(w, T, a, b, l, t, lane, first);
                const V2 s_ = t.rs_s * (a - t.sa1 * b), e_ = t.rs_e * (b - t.scn * a);
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const real_t st = LTR(l, LT_ST(q));
                    V2 x = st * (T[q] - LTR(l, LT_SA(q)) * s_ - LTR(l, LT_SC(q)) * e_);
                    if (q == 0) x = (lane == 0) ? s_ * st : x;
                    if (q == Q - 1) x = (lane == 63) ? e_ * st : x;
                    T[q] = x;
                }
            };
            V2 r[Q], T[Q];
            solve_subs(wp, T, l1, tD1);
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] = T[q];
            asm volatile("" : "+v"(lane) : "v"(r[0].a));
            solve_subs(wu, T, l1, tD1);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                if constexpr (CIRC) r[q] = -0.5 * (cb[q] * T[q] + r[q]);  // (uniform grid: + nu * (T * 0.0) adds nothing)
                else r[q] = -0.5 * (cb[q] * T[q] + r[q]) + nu * (T[q] * LTR(l3, LT_STC(q)));
            }
            asm volatile("" : "+v"(lane) : "v"(r[0].a));
            solve_subs(wu, T, l3, tD2);
            real_t ra[Q], rb[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const V2 v = r[q] + nu * T[q];
                ra[q] = v.a;
                rb[q] = v.b;
            }
            if (CHN && upd.omega != 0.0) {  // (na, nb = this pair's u1 rows while c == 0; cb = its u0 rows)
                if (c == 0) {
#pragma unroll
                    for (int q = 0; q < Q; q++) { ra[q] = -upd.omega * na[q] + 1.0 * ra[q]; rb[q] = -upd.omega * nb[q] + 1.0 * rb[q]; }
                } else if (c == 1) {
#pragma unroll
                    for (int q = 0; q < Q; q++) { ra[q] = upd.omega * cb[q].a + 1.0 * ra[q]; rb[q] = upd.omega * cb[q].b + 1.0 * rb[q]; }
                }
            }
            real_t *oa = (c == 0 ? rhs0 : (c == 1 ? rhs1 : rhs2)) + (long)p * pitch, *ob = oa + pitch;
            if constexpr (Q == 8) { store_rows_q8<ACC>(oa, lane, ra, 1.0); store_rows_q8<ACC>(ob, lane, rb, 1.0); }
            else { store_rows_q4<ACC>(oa, lane, ra, 1.0); store_rows_q4<ACC>(ob, lane, rb, 1.0); }
        }
    }
}

// ---------------------------------------------------------------- K3y: y pencils without transposed copies
// One workgroup = 16 waves = the 16 x-adjacent y pencils of one z plane.  The tile [16 x][n y] goes through
// LDS: the workgroup reads it from the Cartesian block in 128-byte row segments (16 doubles of x per y),
// every wave picks its pencil out of LDS (lane l: rows l Q .. l Q + Q - 1, contiguous), solves the three
// operators exactly like k_xscan_transeq, writes its result back into the tile, and the workgroup adds the
// tile to rhs in 128-byte segments again: u, conv and rhs are touched once each (3-4 field passes instead of
// the 8 of the transposed-copy route in viax.hip), all inside one 2 MB page per tile.
// LDS: lane tables + 16 (n + 4) doubles of tile; for n = 512 this fits only when der1st and der1st_sym have
// identical lane tables (periodic operators: they do), which the launcher checks.
// EPI: the component is the LAST contribution to rhs and the RK / AB stage follows at once: the store phase also
// does the stage's linear combination (time_integrator.py, runge_kutta_fused), per point
//     d = rhs + result;  [rhs = d;]  y = base + sum_k c[k] * (k == ipend ? d : x[k])
// in the summation order of k_lincomb (backend.hip): bit-identical to the accumulating form followed by
// x3d_lincomb, without re-reading d (and without writing it after the last stage).
// (struct TileEpi: common.h)

template <int Q, bool SAME, bool ACC, int FAST, bool EPI = false>
__global__ void __launch_bounds__(1024)
    k_ytile_transeq(real_t *rhs, const real_t *__restrict__ u, const real_t *__restrict__ cv, XOp t1, XOp t2, XOp t3,
                    int share12, int ntx, int ntiles, long prow, long pplane, real_t nu, const TileEpi *epp)
{
    extern __shared__ real_t lt[];
    constexpr int LN = LT_N(Q) * 64, n = 64 * Q, TP = n + 4, NI = n / 128;
    for (int i = threadIdx.x; i < LN; i += blockDim.x) {
        lt[i] = t1.TL[i];
        if (!share12) lt[LN + i] = t2.TL[i];
        lt[(share12 ? 1 : 2) * LN + i] = t3.TL[i];
    }
    const real_t *__restrict__ l1 = lt, *__restrict__ l2 = share12 ? lt : lt + LN,
                 *__restrict__ l3 = lt + (share12 ? 1 : 2) * LN;
    real_t *tile = lt + (share12 ? 2 : 3) * LN;
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int first = lane * Q + 1;
    // cooperative mapping: item i of thread t is the real2_t (row y, columns 2c, 2c + 1)
    const int cy = threadIdx.x >> 3, cc = threadIdx.x & 7;
    auto gload = [&](real2_t (&v)[NI], const real_t *__restrict__ src) {
#pragma unroll
        for (int i = 0; i < NI; i++) v[i] = *reinterpret_cast<const real2_t *>(src + (long)(cy + 128 * i) * prow + 2 * cc);
    };
    auto to_tile = [&](const real2_t (&v)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            tile[(2 * cc) * TP + cy + 128 * i] = v[i].x;
            tile[(2 * cc + 1) * TP + cy + 128 * i] = v[i].y;
        }
    };
    auto pick = [&](real_t (&b)[Q]) {
        const real2_t *__restrict__ src = reinterpret_cast<const real2_t *>(tile + wave * TP + lane * Q);
#pragma unroll
        for (int m = 0; m < Q / 2; m++) {
            const real2_t t2_ = src[m];
            b[2 * m] = t2_.x;
            b[2 * m + 1] = t2_.y;
        }
    };
    __syncthreads();
#ifdef YT_PREF
    real2_t nxt[NI];  // next tile's u rows, in flight during the solve
    if ((int)blockIdx.x < ntiles) gload(nxt, u + (long)(blockIdx.x / ntx) * pplane + (long)(blockIdx.x % ntx) * 16);
#endif
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = (long)(tl / ntx) * pplane + (long)(tl % ntx) * 16;
        asm volatile("" : "+v"(lane));  // keep the lane-table reads inside the loop
        real_t wu[Q + 8], wp[Q + 8], vq[Q];
        {
            real_t b[Q];
            real2_t gu[NI], gc[NI];
#ifdef YT_PREF
#pragma unroll
            for (int i = 0; i < NI; i++) gu[i] = nxt[i];
#else
            gload(gu, u + off);  // both fields in flight at once: one exposed memory latency per tile, not two
#endif
            if (!SAME) gload(gc, cv + off);
            to_tile(gu);
            __syncthreads();
            pick(b);
            window_from_body<Q>(wu, b, lane);
            if (!SAME) {
                __syncthreads();
                to_tile(gc);
                __syncthreads();
                pick(b);
                window_from_body<Q>(wp, b, lane);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) vq[q] = b[q];
#pragma unroll
            for (int m = 0; m < Q + 8; m++) wp[m] = SAME ? wu[m] * wu[m] : wu[m] * wp[m];
        }
        // (no barrier here: until the store phase a wave only rewrites its own pencil's region of the tile)
#ifdef YT_PREF
        {
            const int tn = tl + gridDim.x;
            if (tn < ntiles) gload(nxt, u + (long)(tn / ntx) * pplane + (long)(tn % ntx) * 16);
        }
#endif
        auto solve_subs = [&](const real_t (&w)[Q + 8], real_t (&T)[Q], const real_t *__restrict__ l, const XOp &t) {
            real_t a, b;
            scan_solve<Q, true, (FAST == 2)>(w, T, a, b, l, t, lane, first);
            const real_t s_ = t.rs_s * (a - t.sa1 * b), e_ = t.rs_e * (b - t.scn * a);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const real_t st = LTR(l, LT_ST(q));
                real_t x = st * (T[q] - LTR(l, LT_SA(q)) * s_ - LTR(l, LT_SC(q)) * e_);
                if (q == 0) x = (lane == 0) ? s_ * st : x;
                if (q == Q - 1) x = (lane == 63) ? e_ * st : x;
                T[q] = x;
            }
        };
        real_t r[Q], T[Q];
        solve_subs(wp, T, l2, t2);
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] = T[q];
        asm volatile("" : "+v"(lane) : "v"(r[0]));
        solve_subs(wu, T, l1, t1);
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] = -0.5 * (vq[q] * T[q] + r[q]) + nu * (T[q] * LTR(l3, LT_STC(q)));
        asm volatile("" : "+v"(lane) : "v"(r[0]));
        solve_subs(wu, T, l3, t3);
        if constexpr (!EPI) {
            real_t *__restrict__ o = rhs + off;
            real2_t old[NI];
            if (ACC) gload(old, o);  // in flight while the results go through the tile
            real2_t *__restrict__ dst = reinterpret_cast<real2_t *>(tile + wave * TP + lane * Q);
#pragma unroll
            for (int m = 0; m < Q / 2; m++) dst[m] = make_real2(r[2 * m] + nu * T[2 * m], r[2 * m + 1] + nu * T[2 * m + 1]);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NI; i++) {
                real2_t v = make_real2(tile[(2 * cc) * TP + cy + 128 * i], tile[(2 * cc + 1) * TP + cy + 128 * i]);
                if (ACC) { v.x += old[i].x; v.y += old[i].y; }
                *reinterpret_cast<real2_t *>(o + (long)(cy + 128 * i) * prow + 2 * cc) = v;
            }
        } else {
            // the first two terms other than the pending one travel with `old` and `base` (all an RK3 stage has);
            // further ones (RK4) are loaded in the loop
            // the stage's description is read here, per tile, through a laundered pointer: as kernel arguments (or
            // hoisted out of the tile loop) its 27 scalars stay live across the solves, and with the three
            // operator descriptors the SGPR file overflows into VGPRs (330 spilled)
            const TileEpi *pe = epp;
            asm volatile("" : "+s"(pe) : "v"(T[0]));
            const TileEpi epi = *pe;
            real2_t old[NI], bs[NI];
            // (the addresses are tied to the last solve: issued any earlier the loads stay live across the solves)
            int cce = cc;
            asm volatile("" : "+v"(cce) : "v"(T[0]));
            auto eload = [&](real2_t (&v)[NI], const real_t *src) {
#pragma unroll
                for (int i = 0; i < NI; i++)
                    v[i] = *reinterpret_cast<const real2_t *>(src + (long)(cy + 128 * i) * prow + 2 * cce);
            };
            eload(old, rhs + off);
            eload(bs, epi.base + off);
            real2_t *__restrict__ dst = reinterpret_cast<real2_t *>(tile + wave * TP + lane * Q);
#pragma unroll
            for (int m = 0; m < Q / 2; m++) dst[m] = make_real2(r[2 * m] + nu * T[2 * m], r[2 * m + 1] + nu * T[2 * m + 1]);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const long at = off + (long)(cy + 128 * i) * prow + 2 * cce;
                real2_t d = make_real2(tile[(2 * cc) * TP + cy + 128 * i], tile[(2 * cc + 1) * TP + cy + 128 * i]);
                d.x += old[i].x;
                d.y += old[i].y;
                if (epi.store) *reinterpret_cast<real2_t *>(rhs + at) = d;
                real2_t v = bs[i];
#pragma unroll
                for (int k = 0; k < 5; k++)
                    if (k < epi.n) {
                        real2_t t = d;
                        if (k != epi.ipend) t = *reinterpret_cast<const real2_t *>(epi.x[k] + at);
                        v.x = epi.c[k] * t.x + v.x;
                        v.y = epi.c[k] * t.y + v.y;
                    }
                *reinterpret_cast<real2_t *>(epi.y + at) = v;
            }
        }
        __syncthreads();  // the tile is free again
    }
}

// Row segment (cy + 128 i) of a tile = uniform base (an SGPR pair, bumped per i) + ONE 32-bit lane offset shared
// by all loads and stores of all fields (global_load_dwordx4 v, v_off, s[base:base+1]) instead of a 64-bit
// address pair per row: 7 VGPRs less in kernels that sit on the 128-VGPR limit.  Needs 128 * prow * 8 < 4 GiB
// (checked by the launchers).
template <int RP = 128>
__device__ __forceinline__ const real2_t *tile_row(const real_t *base, long prow, int i, unsigned voff)
{
    return reinterpret_cast<const real2_t *>(reinterpret_cast<const char *>(base + (long)(RP * i) * prow) + voff);
}

// ---------------------------------------------------------------- K3y, the three components of a direction at once
// transeq_<dir> = three components that share the advecting velocity (src/backend/omp/backend.f90:145-184):
// component 0 is (u0, conv = u0), components 1, 2 are (u1, u0), (u2, u0).  One workgroup does all three for
// its tile, keeping its pencil's rows of u0 in registers: u0 is read once instead of three times (9 field
// passes instead of 11).  Needs der1st == der1st_sym and der2nd == der2nd_sym as lane tables (periodic
// operators), so that all components use the same two table sets (tD1 for du and d(u conv), tD2 for d2u).
#ifdef YT_TIMING
// phase timing of k_ytile_transeq3 (scratch builds only): shader-clock ticks summed by wave 0 of every workgroup
__device__ unsigned long long g_yt[8 + 32];  // [8..23]: per wave, ticks from component start to the end of its solves;
                                             // [24..39]: per wave, ticks spent in the solves themselves
extern "C" int x3d_debug_yt(unsigned long long *out, int reset)
{
    X3D_RANGE(__func__);
    if (reset) { unsigned long long z[40] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_yt), z, sizeof z) != hipSuccess; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_yt), sizeof(unsigned long long) * 40) != hipSuccess;
}
#define YT_T(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_yt[k], t_ - yt_last); yt_last = t_; } } while (0)
#else
#define YT_T(k)
#endif

// UNI: both operators live on a uniform grid (stretch == 1, stretch_correct == 0 on every row): the ST / STC lane-table
// reads and their multiplications are skipped (x * 1.0 and + nu * (x * 0.0): the same values).  The solves of this
// kernel are bound by the RATE of LDS read instructions -- 2.6 clocks per ds_read_b64 and CU whatever the lanes read
// (scratch/ldsbench.hip, round 4: broadcast, compressed and exec-masked reads cost the same), 236 table reads per lane
// and component; without any table read the kernel runs at its memory time (2.30 -> 1.88 ms, -DXSCAN_EXP=4).
// P12 (with UNI): the component's first two solves -- d(u conv) and du, the SAME operator on two right-hand sides -- run
// as ONE solve over the pair type V2: every table value read from LDS serves both (204 -> 136 reads per lane and
// component) and the two dependency chains interleave.  Same arithmetic per right-hand side, bit for bit.
// EPI (round 5; local accumulating form): the launch is the LAST contribution to rhs and the RK / AB stage follows at once
// (time_integrator.py, runge_kutta_fused): component c's store phase also does ITS variable's linear combination,
//     d = rhs_c + result;  [rhs_c = d;]  y_c = base_c + sum_k c_k (k == ipend ? d : x_k)
// in the summation order of k_lincomb / k_xscan_tds_lin: bit-identical to this kernel followed by the stage, without d being
// read back by the stage's kernel (and without writing it after the last stage).  epp: three TileEpi in device memory,
// component order (advecting component first).  Per step the EPI launches save 6 of ~ 200 field passes (stages 1 and 3
// of RK3; stage 2 gains nothing: DESIGN.md 3.2) and trade the stage kernel's streaming rate for the tile pattern's.
// NPW (round 6): pencils per wave.  2 = a tile of 32 x-adjacent pencils, wave w solves pencils w and w + 16 one after the
// other.  FP32: a row of the tile is then a 128-byte segment again (16 pencils of 4-byte reals are 64 bytes) and the
// workgroup holds as many bytes in flight as a 512-row FP64 tile, at the same register counts (NI and the advecting rows
// double, every value is half as wide).  256-row FP64 pencils: opt-in (ytile_npw)
// CIRC (round 6, with UNI, P12, local form): the circulant form of both operators (xscan_core.h, circ_solve; cD1 / cD2) -- no
// lane tables are staged (the workgroup's LDS is its tile), no closure; results equal the table form's to round-off
template <int Q, bool ACC, bool NARROW, bool HALO, bool UNI = false, bool P12 = false, bool EPI = false, int NPW = 1,
          bool CIRC = false>
__global__ void __launch_bounds__(1024)
    k_ytile_transeq3(real_t *rhs0, real_t *rhs1, real_t *rhs2, const real_t *__restrict__ u0,
                     const real_t *__restrict__ u1, const real_t *__restrict__ u2, XOp tD1, XOp tD2, int ntx,
                     int tile0, int ntiles, long prow, long pplane, real_t nu, TileHalo th, const TileEpi *epp,
                     CircOp cD1, CircOp cD2)
{
    static_assert(!CIRC || (UNI && (P12 != HALO)), "CIRC: uniform grids; the local form with the pair solve, the HALO form without");
    static_assert(!EPI || (ACC && !HALO), "EPI: the local accumulating form");
    static_assert(NPW == 1 || !HALO, "two pencils per wave: the local form");
    extern __shared__ real_t lt[];
    constexpr int LN = LT_N(Q) * 64, n = 64 * Q, TP = n + 4, NPT = 16 * NPW, RP = 128 / NPW, NI = n / RP;
    if constexpr (!CIRC) {
        for (int i = threadIdx.x; i < LN; i += blockDim.x) {
            lt[i] = tD1.TL[i];
            lt[LN + i] = tD2.TL[i];
        }
    }
    const real_t *__restrict__ l1 = lt, *__restrict__ l3 = lt + LN;
    real_t *tile = lt + (CIRC ? 0 : 2 * LN);
    // HALO: [16 pencils][8] halo values of the current field, then the same for the advecting velocity u0
    real_t *hal = tile + NPT * TP, *hal0 = hal + 128, *bnd = hal0 + 128;  // bnd: [16 pencils][9 ops][du_1, X_n]
    ntiles += tile0;  // tiles [tile0, tile0 + ntiles) (a range of planes: overlap of the neighbour exchange)
#ifdef YT_TIMING
    unsigned long long yt_last = __builtin_readcyclecounter();
#endif
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int first = lane * Q + 1;
    const int cy = threadIdx.x / (8 * NPW), cc = threadIdx.x % (8 * NPW);
    const unsigned voff = (unsigned)(((long)cy * prow + 2 * cc) * X3D_RB);
    auto gload = [&](real2_t (&v)[NI], const real_t *__restrict__ src) {
#pragma unroll
        for (int i = 0; i < NI; i++) v[i] = *tile_row<RP>(src, prow, i, voff);  // (ldg_stream: same time, round 4 A/B)
    };
    auto to_tile = [&](const real2_t (&v)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            tile[(2 * cc) * TP + cy + RP * i] = v[i].x;
            tile[(2 * cc + 1) * TP + cy + RP * i] = v[i].y;
        }
    };
    auto pick = [&](real_t (&b)[Q], int pw) {
        const real2_t *__restrict__ src = reinterpret_cast<const real2_t *>(tile + (wave + 16 * pw) * TP + lane * Q);
#pragma unroll
        for (int m = 0; m < Q / 2; m++) {
            const real2_t t2_ = src[m];
            b[2 * m] = t2_.x;
            b[2 * m + 1] = t2_.y;
        }
    };
    auto tile_off = [&](int tl) { return (long)(tl / ntx) * pplane + (long)(tl % ntx) * NPT; };
    // HALO: thread t < 128 carries halo value (pencil t >> 3, slot t & 7: 0..3 start side, 4..7 end side) of field f
    auto hload = [&](int tl, int f) {
        const int hw = threadIdx.x >> 3, hk = threadIdx.x & 7;
        const long pp = (long)(tl / ntx) * th.hp + (long)(tl % ntx) * 16 + hw;  // pencil (x, other) of a halo plane
        return th.recv[((long)((hk >> 2) * th.nf + f) * 4 + (hk & 3)) * th.hnp + pp];
    };
    __syncthreads();
    real2_t nxt[NI];  // the rows needed next (next component's field, or the next tile's u0), in flight during the solves
    real_t hnx = 0.0;
    if (tile0 + (int)blockIdx.x < ntiles) {
        gload(nxt, u0 + tile_off(tile0 + blockIdx.x));
        if (HALO && threadIdx.x < 128) hnx = hload(tile0 + blockIdx.x, 0);
    }
    // behind its prefetch a component issues NI loads of the old rhs rows (ACC) and NI stores (common.h)
    vmcnt_pad_stores<(ACC ? 2 : 1) * NI>();
    for (int tl = tile0 + blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = tile_off(tl);
        real_t cb[NPW][Q];  // this wave's pencils' rows of the advecting velocity
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            asm volatile("" : "+v"(lane));
            YT_T(5);
#ifdef YT_TIMING
            const unsigned long long yt_c0 = __builtin_readcyclecounter();
#endif
            to_tile(nxt);
            if (HALO && threadIdx.x < 128) {
                hal[threadIdx.x] = hnx;
                if (c == 0) hal0[threadIdx.x] = hnx;
            }
            YT_T(0);
            __syncthreads();
            YT_T(1);
            real2_t old[NI];
            [[maybe_unused]] real2_t bs[NI];
            [[maybe_unused]] TileEpi epi;
            real_t *o = (c == 0 ? rhs0 : (c == 1 ? rhs1 : rhs2)) + off;
#pragma unroll
            for (int pw = 0; pw < NPW; pw++) {
            real_t wu[Q + 8], wp[Q + 8];
            {
                real_t b[Q];
                pick(b, pw);
                if (c == 0) {
#pragma unroll
                    for (int q = 0; q < Q; q++) cb[pw][q] = b[q];
                }
                if constexpr (HALO) {
                    window_from_body_halo<Q>(wu, b, lane, hal + wave * 8);
                    window_from_body_halo<Q>(wp, cb[pw], lane, hal0 + wave * 8);
                } else {
                    window_from_body<Q>(wu, b, lane);
                    window_from_body<Q>(wp, cb[pw], lane);
                }
#pragma unroll
                for (int m = 0; m < Q + 8; m++) wp[m] = wu[m] * wp[m];
            }
            // (no barrier here: until the store phase a wave only rewrites its own pencils' regions of the tile)
            if (pw == 0) {
                const int tn = tl + gridDim.x;
                const real_t *nsrc = c == 0 ? u1 + off : (c == 1 ? u2 + off : u0 + tile_off(tn < ntiles ? tn : tl));
                if (c < 2 || tn < ntiles) {
                    gload(nxt, nsrc);
                    if (HALO && threadIdx.x < 128) hnx = hload(c < 2 ? tl : tn, c < 2 ? c + 1 : 0);
                }
            }

            auto solve_subs = [&](const real_t (&w)[Q + 8], real_t (&T)[Q], const real_t *__restrict__ l, const XOp &t,
                                  int op) {
                if constexpr (CIRC && HALO) {  // the open-ended circulant solve; its two boundary values go out as du_1 / X_n do
                    real_t ab[2];
                    if (op < 2) circ_solve<Q, NARROW, real_t, CircOp, true>(w, T, cD1, lane, ab);
                    else circ_solve<Q, NARROW, real_t, CircOp, true>(w, T, cD2, lane, ab);
                    bnd[(wave * 9 + c * 3 + op) * 2] = ab[0];
                    bnd[(wave * 9 + c * 3 + op) * 2 + 1] = ab[1];
                    return;
                }
                real_t a, b;
                scan_solve<Q, true, NARROW>(w, T, a, b, l, t, lane, first);
                real_t s_, e_;
                if constexpr (HALO) {
                    // recv_s = recv_e = 0 for now; own boundary values (wave-uniform) parked in LDS, written out
                    // after the third component for the exchange (k_transeq_halo_fix)
                    s_ = t.rs_s * a; e_ = t.rs_e * b;
                    bnd[(wave * 9 + c * 3 + op) * 2] = a;
                    bnd[(wave * 9 + c * 3 + op) * 2 + 1] = b;
                } else {
                    s_ = t.rs_s * (a - t.sa1 * b); e_ = t.rs_e * (b - t.scn * a);
                }
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    if constexpr (UNI) {
                        real_t x = T[q] - LTR(l, LT_SA(q)) * s_ - LTR(l, LT_SC(q)) * e_;
                        if (q == 0) x = (lane == 0) ? s_ : x;
                        if (q == Q - 1) x = (lane == 63) ? e_ : x;
                        T[q] = x;
                    } else {
                        const real_t st = LTR(l, LT_ST(q));
                        real_t x = st * (T[q] - LTR(l, LT_SA(q)) * s_ - LTR(l, LT_SC(q)) * e_);
                        if (q == 0) x = (lane == 0) ? s_ * st : x;
                        if (q == Q - 1) x = (lane == 63) ? e_ * st : x;
                        T[q] = x;
                    }
                }
            };
            real_t r[Q], T[Q];
            YT_T(2);
#ifdef YT_TIMING
            const unsigned long long yt_s0 = __builtin_readcyclecounter();
#endif
            if constexpr (CIRC && !HALO) {
                V2 w2[Q + 8], T2[Q];
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m] = V2{wp[m], wu[m]};
                circ_solve<Q, NARROW, V2>(w2, T2, cD1, lane);
#pragma unroll
                for (int q = 0; q < Q; q++) r[q] = -0.5 * (cb[pw][q] * T2[q].b + T2[q].a);
            } else if constexpr (P12 && UNI) {
                V2 w2[Q + 8], T2[Q], a2, b2;
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m] = V2{wp[m], wu[m]};
                scan_solve<Q, true, NARROW, V2>(w2, T2, a2, b2, l1, tD1, lane, first);
                V2 s_, e_;
                if constexpr (HALO) {
                    s_ = tD1.rs_s * a2; e_ = tD1.rs_e * b2;
                    bnd[(wave * 9 + c * 3 + 0) * 2] = a2.a; bnd[(wave * 9 + c * 3 + 0) * 2 + 1] = b2.a;
                    bnd[(wave * 9 + c * 3 + 1) * 2] = a2.b; bnd[(wave * 9 + c * 3 + 1) * 2 + 1] = b2.b;
                } else {
                    s_ = tD1.rs_s * (a2 - tD1.sa1 * b2); e_ = tD1.rs_e * (b2 - tD1.scn * a2);
                }
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const real_t sa = LTR(l1, LT_SA(q)), sc = LTR(l1, LT_SC(q));
                    real_t xa = T2[q].a - sa * s_.a - sc * e_.a, xb = T2[q].b - sa * s_.b - sc * e_.b;
                    if (q == 0) { xa = (lane == 0) ? s_.a : xa; xb = (lane == 0) ? s_.b : xb; }
                    if (q == Q - 1) { xa = (lane == 63) ? e_.a : xa; xb = (lane == 63) ? e_.b : xb; }
                    r[q] = -0.5 * (cb[pw][q] * xb + xa);
                }
            } else {
            solve_subs(wp, T, l1, tD1, 0);
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] = T[q];
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            solve_subs(wu, T, l1, tD1, 1);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                if constexpr (UNI) r[q] = -0.5 * (cb[pw][q] * T[q] + r[q]);
                else r[q] = -0.5 * (cb[pw][q] * T[q] + r[q]) + nu * (T[q] * LTR(l3, LT_STC(q)));
            }
            }
            asm volatile("" : "+v"(lane) : "v"(r[0]));
            if constexpr (P12 && UNI) {
                // the field's window again, from the tile (it still holds the component's rows): keeping wu alive
                // across the pair solve does not fit the 128 registers (121 + 56 bytes of scratch)
                real_t b[Q];
                pick(b, pw);
                if constexpr (HALO) window_from_body_halo<Q>(wu, b, lane, hal + wave * 8);
                else window_from_body<Q>(wu, b, lane);
            }
            if constexpr (CIRC && !HALO) circ_solve<Q, NARROW>(wu, T, cD2, lane);
            else solve_subs(wu, T, l3, tD2, 2);
            // (the loads below are issued here, before the LAST pencil's rows go back into the tile, not before the solves:
            //  16 more live VGPRs there spill -- 0.81 -> 1.28 ms per component; with the partial result parked in the tile
            //  during the last solve to make room for them: no spills, but 2.37 instead of 2.24 ms per launch; issued after
            //  the FIRST solve, where the product window's registers are free (123 VGPRs, no spills): 2.18 ms both ways --
            //  the load's latency is not what limits, the memory system is busy throughout with this pattern's segments)
            if (pw == NPW - 1) {
                if constexpr (EPI) {
                    // the stage's description, per component, through a laundered pointer (as kernel arguments its scalars
                    // would stay live across the solves and overflow the SGPR file -- see k_ytile_transeq<EPI>)
                    const TileEpi *pe = epp + c;
                    asm volatile("" : "+s"(pe) : "v"(T[0]));
                    epi = *pe;
                    gload(old, o);
                    gload(bs, epi.base + off);
                } else {
                    YT_T(3);
#ifdef YT_TIMING
                    if (lane == 0) {
                        const unsigned long long t_ = __builtin_readcyclecounter();
                        atomicAdd(&g_yt[8 + wave], t_ - yt_c0);
                        atomicAdd(&g_yt[24 + wave], t_ - yt_s0);
                    }
#endif
#ifdef YT_NT
                    if (ACC) {
#pragma unroll
                        for (int i = 0; i < NI; i++) {
                            const real_t *q_ = reinterpret_cast<const real_t *>(tile_row<RP>(o, prow, i, voff));
                            old[i] = make_real2(__builtin_nontemporal_load(q_), __builtin_nontemporal_load(q_ + 1));
                        }
                    }
#else
                    if (ACC) gload(old, o);
#endif
                }
            }
            {
                real2_t *__restrict__ dst = reinterpret_cast<real2_t *>(tile + (wave + 16 * pw) * TP + lane * Q);
#pragma unroll
                for (int m = 0; m < Q / 2; m++)
                    dst[m] = make_real2(r[2 * m] + nu * T[2 * m], r[2 * m + 1] + nu * T[2 * m + 1]);
            }
            if (NPW > 1) asm volatile("" : "+v"(lane) : "v"(r[0]));  // (one pencil at a time: the next one's reads wait)
            }  // (pw)
            {
                if constexpr (EPI) {
                    __syncthreads();
                    real2_t d[NI];
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        d[i] = make_real2(tile[(2 * cc) * TP + cy + RP * i], tile[(2 * cc + 1) * TP + cy + RP * i]);
                        d[i].x += old[i].x;
                        d[i].y += old[i].y;
                        if (epi.store) *const_cast<real2_t *>(tile_row<RP>(o, prow, i, voff)) = d[i];
                    }
#pragma unroll  // (static indices into epi: a run-time index would put the struct on the stack)
                    for (int k = 0; k < 5; k++) {
                        if (k >= epi.n) continue;
                        if (k == epi.ipend) {
#pragma unroll
                            for (int i = 0; i < NI; i++) { bs[i].x = epi.c[k] * d[i].x + bs[i].x; bs[i].y = epi.c[k] * d[i].y + bs[i].y; }
                        } else {
                            real2_t xk[NI];
                            gload(xk, epi.x[k] + off);
#pragma unroll
                            for (int i = 0; i < NI; i++) { bs[i].x = epi.c[k] * xk[i].x + bs[i].x; bs[i].y = epi.c[k] * xk[i].y + bs[i].y; }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NI; i++) *const_cast<real2_t *>(tile_row<RP>(epi.y + off, prow, i, voff)) = bs[i];
                } else {
                __syncthreads();
                YT_T(4);
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    real2_t v = make_real2(tile[(2 * cc) * TP + cy + RP * i], tile[(2 * cc + 1) * TP + cy + RP * i]);
                    if (ACC) { v.x += old[i].x; v.y += old[i].y; }
#ifdef YT_NT
                    {
                        real_t *q_ = reinterpret_cast<real_t *>(const_cast<real2_t *>(tile_row<RP>(o, prow, i, voff)));
                        __builtin_nontemporal_store(v.x, q_);
                        __builtin_nontemporal_store(v.y, q_ + 1);
                    }
#else
                    *const_cast<real2_t *>(tile_row<RP>(o, prow, i, voff)) = v;
#endif
                }
                }  // (!EPI)
            }
            YT_T(6);
            __syncthreads();  // the tile is free again
        }
        if (HALO && threadIdx.x < 288) {  // 16 pencils x 9 operators x {du_1, X_n} (the barrier above ordered them)
            const int hw = threadIdx.x / 18, k = threadIdx.x % 18;
            const long pp = (long)(tl / ntx) * (ntx * 16) + (long)(tl % ntx) * 16 + hw;
            th.bsend[((long)(k & 1) * th.nb + (k >> 1)) * th.np + pp] = bnd[threadIdx.x];
        }
    }
}

// ---------------------------------------------------------------- K3y for tds_solve pairs (pressure correction)
// The y operators of divergence_v2c / gradient_c2v come in pairs that share an output or an input
// (src/vector_calculus.f90:142-332 as sequenced by pressure_correction_fused):
//   MODE 0:  out  = A(in1) + B(in2)      (interpl_y(du_x) + stagder_y(dv_x):   3 field passes instead of 2 + 3)
//   MODE 1:  out1 = A(in1), out2 = B(in1) (interpl_y(p), stagder_y(p):         3 instead of 2 + 2)
// Same tile mechanics as k_ytile_transeq, same arithmetic as k_xscan_tds (MODE 0 adds the two results exactly
// like the accumulating form: old + 1.0 * r).
// ZF (z pencils of 512 rows, MODE 0 / 1, local form): the z-first Poisson solve's transform along z on the tile
// (csrc/zfft_tile.h) -- MODE 0 stores the 257 x 16 modes of its result into the spectrum instead of out1, MODE 1 builds
// its input tile from the spectrum instead of in1 (neither array is touched).  The tile area grows to the transforms'
// 72 KB; the table sets are staged without their STC block, which tds_solve does not read.
// UNI: both operators on a uniform grid -- no ST reads / multiplications (see k_ytile_transeq3)
// CIRC (round 6, with UNI, local form): both operators in the circulant form (circ_solve; ca / cb), no lane tables staged
// CDUAL (with CIRC, MODE 0 / 1): the pair's two solves as ONE solve over the pair type with each operator's constants
// (circ_pair): one pass through the phases, two dependency chains in flight.  Measured even (z-transforming mode 1 0.745 ->
// 0.716 ms, mode 0 0.782 -> 0.794, plain pairs unchanged): only with X3D_CIRC_DUAL=1
template <int Q, int MODE, bool NARROW, bool HALO, bool ZF = false, bool UNI = false, bool CIRC = false, bool CDUAL = false>
__global__ void __launch_bounds__(1024)
    k_ytile_tds_pair(real_t *out1, real_t *out2, const real_t *__restrict__ in1, const real_t *__restrict__ in2,
                     XOp ta, XOp tb, int ntx, int tile0, int ntiles, long prow, long pplane, TileHalo th, int permn,
                     ZfArg zf, CircOp ca, CircOp cb)
{
    extern __shared__ real_t lt[];
    constexpr int LN = (ZF ? LT_NC(Q) : LT_N(Q)) * 64, n = 64 * Q, TP = n + 4, NI = n / 128;
    static_assert(!ZF || (Q == 8 && MODE != 2 && !HALO), "the z-transforming forms: local pairs on 512-row pencils");
    static_assert(!CIRC || UNI, "CIRC: uniform grids");
    if constexpr (!CIRC) {
        for (int i = threadIdx.x; i < LN; i += blockDim.x) {
            lt[i] = ta.TL[i];
            if (MODE != 2) lt[LN + i] = tb.TL[i];
        }
    }
    const real_t *__restrict__ la = lt, *__restrict__ lb = lt + LN;
    real_t *tile = lt + (CIRC ? 0 : (MODE == 2 ? 1 : 2) * LN);  // (a single operator stages one table set)
    real2_t *tws = reinterpret_cast<real2_t *>(tile + ZF_AREA_DOUBLES);  // ZF: W512^k behind the 72 KB tile area
    if (ZF && threadIdx.x < 256) tws[threadIdx.x] = zf.tw[threadIdx.x];
    const long kzs = (long)zf.ny * zf.px;
    auto zf_row = [&](int tl) {
        int r = tl / ntx;  // (zf.permn: the 010 solver's row order, as tile_off_p below)
        if (zf.permn > 0 && r < zf.permn) r = (r & 1) ? zf.permn - ((r + 1) >> 1) : (r >> 1);
        return zf.c + (long)r * zf.px + (long)(tl % ntx) * 16;
    };
    // HALO, MODE 0: no room for the next tile's rows next to the second input's (with them in flight across the
    // solves the kernel spills inside the tile loop, and every reload is an exposed memory latency: 1.40 ms
    // against 0.95 for the local form): this tile's rows are requested at its top instead
    constexpr bool NOPREF = HALO && MODE == 0;
    real_t *hal = tile + 16 * TP, *bnd = hal + 128;  // HALO: [16 pencils][8] halo values of the input in the tile;
                                                     // [16 pencils][2 ops][du_1, X_n]
    ntiles += tile0;                                 // tiles [tile0, tile0 + ntiles)
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int first = lane * Q + 1;
    const int cy = threadIdx.x >> 3, cc = threadIdx.x & 7;
    const unsigned voff = (unsigned)(((long)cy * prow + 2 * cc) * X3D_RB);
    auto gload = [&](real2_t (&v)[NI], const real_t *__restrict__ src) {
#pragma unroll
        for (int i = 0; i < NI; i++) v[i] = *tile_row(src, prow, i, voff);  // (ldg_stream: same time, round 4 A/B)
    };
    auto to_tile = [&](const real2_t (&v)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            tile[(2 * cc) * TP + cy + 128 * i] = v[i].x;
            tile[(2 * cc + 1) * TP + cy + 128 * i] = v[i].y;
        }
    };
    auto pick = [&](real_t (&b)[Q]) {
        const real2_t *__restrict__ src = reinterpret_cast<const real2_t *>(tile + wave * TP + lane * Q);
#pragma unroll
        for (int m = 0; m < Q / 2; m++) {
            const real2_t t2_ = src[m];
            b[2 * m] = t2_.x;
            b[2 * m + 1] = t2_.y;
        }
    };
    auto put = [&](const real_t (&r)[Q]) {
        real2_t *__restrict__ dst = reinterpret_cast<real2_t *>(tile + wave * TP + lane * Q);
#pragma unroll
        for (int m = 0; m < Q / 2; m++) dst[m] = make_real2(r[2 * m], r[2 * m + 1]);
    };
    auto from_tile = [&](real_t *o) {
#pragma unroll
        for (int i = 0; i < NI; i++)
            *const_cast<real2_t *>(tile_row(o, prow, i, voff)) =
                make_real2(tile[(2 * cc) * TP + cy + 128 * i], tile[(2 * cc + 1) * TP + cy + 128 * i]);
    };
    // one operator on the window w: r = its tds_solve rows (der_univ_subs with the periodic self-exchange; HALO:
    // with recv_s = recv_e = 0, the own boundary values stored for the exchange -- see TileHalo)
    auto solve = [&](const real_t (&w)[Q + 8], real_t (&r)[Q], const real_t *__restrict__ l, const XOp &t, int op) {
#if ZF_EXP & 2  // (timing experiment: no solve)
        if constexpr (ZF) {
#pragma unroll
            for (int q = 0; q < Q; q++) r[q] = w[q + 4];
            return;
        }
#endif
        if constexpr (CIRC && HALO) {
            real_t ab[2];
            if (op == 0) circ_solve<Q, NARROW, real_t, CircOp, true>(w, r, ca, lane, ab);
            else circ_solve<Q, NARROW, real_t, CircOp, true>(w, r, cb, lane, ab);
            bnd[(wave * 2 + op) * 2] = ab[0];
            bnd[(wave * 2 + op) * 2 + 1] = ab[1];
            return;
        } else if constexpr (CIRC) {
            if (op == 0) circ_solve<Q, NARROW>(w, r, ca, lane);
            else circ_solve<Q, NARROW>(w, r, cb, lane);
            return;
        }
        real_t X[Q], du1, xn;
        scan_solve<Q, true, NARROW>(w, X, du1, xn, l, t, lane, first);
        real_t du_s, du_e;
        if constexpr (HALO) {
            du_s = t.rs_s * du1; du_e = t.rs_e * xn;
            bnd[(wave * 2 + op) * 2] = du1;  // (wave-uniform values, parked in LDS until the tile is done)
            bnd[(wave * 2 + op) * 2 + 1] = xn;
        } else {
            du_s = t.rs_s * (du1 - t.sa1 * xn); du_e = t.rs_e * (xn - t.scn * du1);
        }
#pragma unroll
        for (int q = 0; q < Q; q++) {
            if constexpr (UNI) {
                r[q] = X[q] - LTR(l, LT_SA(q)) * du_s - LTR(l, LT_SC(q)) * du_e;
                if (q == 0) r[q] = (lane == 0) ? du_s : r[q];
                if (q == Q - 1) r[q] = (lane == 63) ? du_e : r[q];
            } else {
                const real_t st = LTR(l, LT_ST(q));
                r[q] = (X[q] - LTR(l, LT_SA(q)) * du_s - LTR(l, LT_SC(q)) * du_e) * st;
                if (q == 0) r[q] = (lane == 0) ? du_s * st : r[q];
                if (q == Q - 1) r[q] = (lane == 63) ? du_e * st : r[q];
            }
        }
    };
    auto tile_off = [&](int tl) { return (long)(tl / ntx) * pplane + (long)(tl % ntx) * 16; };
    // permn > 0 (z pencils, tl / ntx = the y row): the y rows of the 010 Poisson solver's input / output are
    // interleaved (enforce / undo_periodicity_y, src/backend/cuda/kernels/spectral_processing.f90:1062-1114: row
    // 2j-1 <-> position j, row 2j <-> position ny-j+1, 1-based): MODE 0 writes its result at the row's position,
    // MODE 1 reads its input there -- the two copy kernels of the solver are not launched
    auto tile_off_p = [&](int tl) {
        int r = tl / ntx;
        if (permn > 0 && r < permn) r = (r & 1) ? permn - ((r + 1) >> 1) : (r >> 1);
        return (long)r * pplane + (long)(tl % ntx) * 16;
    };
    auto in1_off = [&](int tl) { return MODE == 1 ? tile_off_p(tl) : tile_off(tl); };
    auto hload = [&](int tl, int f) {  // thread t < 128: halo value (pencil t >> 3, slot t & 7) of input f
        const int hw = threadIdx.x >> 3, hk = threadIdx.x & 7;
        int r = tl / ntx;  // (MODE 1's input comes interleaved from the neighbours as well)
        if (MODE == 1 && permn > 0 && r < permn) r = (r & 1) ? permn - ((r + 1) >> 1) : (r >> 1);
        const long pp = (long)r * th.hp + (long)(tl % ntx) * 16 + hw;
        return th.recv[((long)((hk >> 2) * th.nf + f) * 4 + (hk & 3)) * th.hnp + pp];
    };
    __syncthreads();
    real2_t nxt[NI];  // next tile's in1 rows, in flight during the solves
    ZfRows spn{};  // ZF, MODE 1: next tile's modes instead
    real_t hnx = 0.0;
    if (!NOPREF && tile0 + (int)blockIdx.x < ntiles) {
        if constexpr (ZF && MODE == 1) spn = zf_inverse_load(zf_row(tile0 + blockIdx.x), kzs);
        else gload(nxt, in1 + in1_off(tile0 + blockIdx.x));
        if (HALO && threadIdx.x < 128) hnx = hload(tile0 + blockIdx.x, 0);
    }
    for (int tl = tile0 + blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = tile_off(tl);
        asm volatile("" : "+v"(lane));
        real_t w[Q + 8], b[Q], ra[Q], rb[Q];
        real2_t g2[NI];
        real_t h2 = 0.0;
        if (MODE == 0) {
            gload(g2, in2 + off);
            if (HALO && threadIdx.x < 128) h2 = hload(tl, 1);
        }
        if (NOPREF) {
            gload(nxt, in1 + in1_off(tl));
            if (threadIdx.x < 128) hnx = hload(tl, 0);
        }
        if constexpr (ZF && MODE == 1) {
            zf_inverse<TP>(tile, tws, spn, wave, lane);  // (ends behind a barrier)
        } else {
            to_tile(nxt);
            if (HALO && threadIdx.x < 128) hal[threadIdx.x] = hnx;
            __syncthreads();
        }
        pick(b);
        if constexpr (HALO) window_from_body_halo<Q>(w, b, lane, hal + wave * 8);
        else window_from_body<Q>(w, b, lane);
        // (barriers only before COOPERATIVE accesses to the tile: a wave's own results go to its own pencil's region)
        if (MODE == 0) {
            // the rows must BE in registers before the barrier: pick reads through a __restrict__ pointer, which lets the
            // compiler sink its loads below the barrier (round 6: seen in the ISA of the CIRC form, whose first solve no
            // longer touches LDS -- two of the four ds_read_b128 landed behind the other waves' to_tile(g2))
#pragma unroll
            for (int q = 0; q < Q; q++) asm volatile("" : "+v"(b[q]));
            __syncthreads();  // all rows picked: the second input may overwrite the tile
        }
        if (!NOPREF) {
            const int tn = tl + gridDim.x;
            if (tn < ntiles) {
                if constexpr (ZF && MODE == 1) spn = zf_inverse_load(zf_row(tn), kzs);
                else gload(nxt, in1 + in1_off(tn));
                if (HALO && threadIdx.x < 128) hnx = hload(tn, 0);
            }
        }
#ifdef XS_DUAL
        // DUAL (round 5 experiment, OFF: measured no better -- ZF mode 0 0.874 -> 0.905 ms, mode 1 0.773 -> 0.767, step
        // 43.07 -> 43.2 ms; profiles/r05_zf_pair_phases.txt): the pair's two solves -- different operators -- as ONE
        // interleaved solve over both table sets (scan_solve_dual).  The solves are bound by the rate of LDS read
        // instructions, which interleaving does not lower, and mode 0 loses the overlap of its first solve with the
        // second input's trip through the tile.  Parity-green (30 pair / z-first / full-step tests).
        constexpr bool DUAL = UNI && !HALO && MODE != 2;
#else
        constexpr bool DUAL = false;
#endif
        if constexpr (CIRC && CDUAL && MODE != 2) {
            V2 w2[Q + 8], X2[Q];
            if (MODE == 0) {
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m].a = w[m];
                to_tile(g2);
                __syncthreads();
                pick(b);
                window_from_body<Q>(w, b, lane);
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m].b = w[m];
            } else {
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m] = V2{w[m], w[m]};
            }
            circ_solve<Q, NARROW, V2, CircOp2>(w2, X2, circ_pair(ca, cb), lane);
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < Q; q++) ra[q] = X2[q].a + 1.0 * X2[q].b;
                put(ra);
                __syncthreads();
                if constexpr (ZF) zf_forward<TP>(tile, tws, zf_row(tl), kzs, wave, lane);
                else from_tile(out1 + tile_off_p(tl));
            } else {
#pragma unroll
                for (int q = 0; q < Q; q++) { ra[q] = X2[q].a; rb[q] = X2[q].b; }
                put(ra);
                __syncthreads();
                from_tile(out1 + off);
                __syncthreads();  // out1's tile has been read
                put(rb);
                __syncthreads();
                from_tile(out2 + off);
            }
        } else if constexpr (DUAL) {
            V2 w2[Q + 8];
            if (MODE == 0) {
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m].a = w[m];
                to_tile(g2);
                __syncthreads();
                pick(b);
                window_from_body<Q>(w, b, lane);
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m].b = w[m];
            } else {
#pragma unroll
                for (int m = 0; m < Q + 8; m++) w2[m] = V2{w[m], w[m]};
            }
            V2 X2[Q], d1, xn2;
            scan_solve_dual<Q, NARROW>(w2, X2, d1, xn2, la, lb, ta, tb, lane);
            const real_t sa_ = ta.rs_s * (d1.a - ta.sa1 * xn2.a), ea_ = ta.rs_e * (xn2.a - ta.scn * d1.a);
            const real_t sb_ = tb.rs_s * (d1.b - tb.sa1 * xn2.b), eb_ = tb.rs_e * (xn2.b - tb.scn * d1.b);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                ra[q] = X2[q].a - LTR(la, LT_SA(q)) * sa_ - LTR(la, LT_SC(q)) * ea_;
                rb[q] = X2[q].b - LTR(lb, LT_SA(q)) * sb_ - LTR(lb, LT_SC(q)) * eb_;
                if (q == 0) { ra[q] = (lane == 0) ? sa_ : ra[q]; rb[q] = (lane == 0) ? sb_ : rb[q]; }
                if (q == Q - 1) { ra[q] = (lane == 63) ? ea_ : ra[q]; rb[q] = (lane == 63) ? eb_ : rb[q]; }
            }
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < Q; q++) ra[q] = ra[q] + 1.0 * rb[q];
                put(ra);
                __syncthreads();
                if constexpr (ZF) zf_forward<TP>(tile, tws, zf_row(tl), kzs, wave, lane);
                else from_tile(out1 + tile_off_p(tl));
            } else {
                put(ra);
                __syncthreads();
                from_tile(out1 + off);
                __syncthreads();  // out1's tile has been read
                put(rb);
                __syncthreads();
                from_tile(out2 + off);
            }
        } else {
        solve(w, ra, la, ta, 0);
        if (MODE == 0) {
            to_tile(g2);
            if (HALO && threadIdx.x < 128) hal[threadIdx.x] = h2;
            __syncthreads();
            pick(b);
            if constexpr (HALO) window_from_body_halo<Q>(w, b, lane, hal + wave * 8);
            else window_from_body<Q>(w, b, lane);
            asm volatile("" : "+v"(lane) : "v"(ra[0]));
            solve(w, rb, lb, tb, 1);
#pragma unroll
            for (int q = 0; q < Q; q++) ra[q] = ra[q] + 1.0 * rb[q];
            put(ra);
            __syncthreads();
            if constexpr (ZF) zf_forward<TP>(tile, tws, zf_row(tl), kzs, wave, lane);
            else from_tile(out1 + tile_off_p(tl));
        } else if (MODE == 1) {
            put(ra);
            __syncthreads();
            from_tile(out1 + off);
            asm volatile("" : "+v"(lane) : "v"(ra[0]));
            solve(w, rb, lb, tb, 1);
            __syncthreads();  // out1's tile has been read
            put(rb);
            __syncthreads();
            from_tile(out2 + off);
        } else {  // MODE 2: a single operator (a tds_solve of a decomposed direction through the tile mechanics)
            put(ra);
            __syncthreads();
            from_tile(out1 + off);
        }
        }  // (!DUAL)
        if (HALO && threadIdx.x < 64) {  // 16 pencils x 2 operators x {du_1, X_n} (ordered by the barriers above)
            const int hw = threadIdx.x >> 2, k = threadIdx.x & 3;
            const long pp = (long)(tl / ntx) * (ntx * 16) + (long)(tl % ntx) * 16 + hw;
            if ((k >> 1) < th.nb) th.bsend[((long)(k & 1) * th.nb + (k >> 1)) * th.np + pp] = bnd[threadIdx.x];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- HALO: the terms linear in the received values
// One pencil per lane, lanes across x (every access a contiguous row segment), rows wave-uniform -> the row tables
// come from scalar loads.  With ds = -rs_s sa_1 recv_s and de = -rs_e sc_n recv_e (what the neighbours' values
// add to du_s, du_e of src/backend/omp/kernels/distributed.f90:196-206):
//     x_j += -st_j (sa_j ds + sc_j de)   (2 <= j <= n - 1),   x_1 += st_1 ds,   x_n += st_n de
// on rows 1..ws and n-we+1..n (the launcher cuts where |sa_j|, |sc_j| < 2^-60; ws + we >= n: every row).
__device__ __forceinline__ real_t halo_fix_row(const TdsTab &t, int j, int n, real_t ds, real_t de)
{
    const real_t st = T_ST(t, j);
    if (j == 1) return st * ds;
    if (j == n) return st * de;
    return -st * (T_SA(t, j) * ds + T_SC(t, j) * de);
}

template <int MODE>
__global__ void __launch_bounds__(256)
    k_tds_halo_fix(real_t *__restrict__ out1, real_t *__restrict__ out2, const real_t *__restrict__ brecv, TdsTab ta,
                   TdsTab tb, PencilGeom g, int ws, int we, int permn)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const int nb = MODE == 2 ? 1 : 2, n = ta.n_tds, seg = blockIdx.y;  // seg 0: start strip, 1: end strip
    int r1 = p / g.dim0;  // permn > 0 (z pencils, MODE 0): out1's y rows are interleaved (k_ytile_tds_pair)
    if (MODE == 0 && permn > 0 && r1 < permn) r1 = (r1 & 1) ? permn - ((r1 + 1) >> 1) : (r1 >> 1);
    const long base = (long)(p % g.dim0) * g.s0 + (long)r1 * g.s1;
    const real_t dsa = -ta.rs_s * ta.sa1 * brecv[p], dea = -ta.rs_e * ta.scn * brecv[(long)nb * g.np + p];
    real_t dsb = 0.0, deb = 0.0;
    if (MODE != 2) {
        dsb = -tb.rs_s * tb.sa1 * brecv[(long)g.np + p];
        deb = -tb.rs_e * tb.scn * brecv[(long)(nb + 1) * g.np + p];
    }
    const int j0 = seg == 0 ? 1 : (n - we + 1 > ws ? n - we + 1 : ws + 1), j1 = seg == 0 ? (ws < n ? ws : n) : n;
    for (int j = j0; j <= j1; j++) {
        const long o = base + (long)(j - 1) * g.rs;
        const real_t fa = halo_fix_row(ta, j, n, dsa, dea);
        if (MODE == 0) out1[o] += fa + halo_fix_row(tb, j, n, dsb, deb);
        else out1[o] += fa;
        if (MODE == 1) out2[o] += halo_fix_row(tb, j, n, dsb, deb);
    }
}

// transeq_<dir>, three components: rhs_c += -1/2 (v ddu + ddud) + nu (dd2u + ddu stc)
// (src/backend/omp/kernels/distributed.f90:304-335 is linear in du, dud, d2u); brecv = [side][c * 3 + op][np],
// op 0 = d(u conv) (tD1), 1 = du (tD1), 2 = d2u (tD2); v = the advecting velocity
__global__ void __launch_bounds__(256)
    k_transeq_halo_fix(real_t *__restrict__ rhs0, real_t *__restrict__ rhs1, real_t *__restrict__ rhs2,
                       const real_t *__restrict__ v, const real_t *__restrict__ brecv, TdsTab t1, TdsTab t2,
                       PencilGeom g, real_t nu, int ws, int we)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const int n = t1.n_tds, seg = blockIdx.y;
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, np = g.np;
    real_t ds[9], de[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const TdsTab &t = (k % 3 == 2) ? t2 : t1;
        ds[k] = -t.rs_s * t.sa1 * brecv[k * np + p];
        de[k] = -t.rs_e * t.scn * brecv[(9 + k) * np + p];
    }
    real_t *__restrict__ rhs[3] = {rhs0, rhs1, rhs2};
    const int j0 = seg == 0 ? 1 : (n - we + 1 > ws ? n - we + 1 : ws + 1), j1 = seg == 0 ? (ws < n ? ws : n) : n;
    for (int j = j0; j <= j1; j++) {
        const long o = base + (long)(j - 1) * g.rs;
        const real_t vj = v[o], stc = T_STC(t2, j);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const real_t ddud = halo_fix_row(t1, j, n, ds[3 * c], de[3 * c]);
            const real_t ddu = halo_fix_row(t1, j, n, ds[3 * c + 1], de[3 * c + 1]);
            const real_t dd2u = halo_fix_row(t2, j, n, ds[3 * c + 2], de[3 * c + 2]);
            rhs[c][o] += -0.5 * (vj * ddu + ddud) + nu * (dd2u + ddu * stc);
        }
    }
}

// ---------------------------------------------------------------- launchers
// bulk stencil within +-2 rows (compact6 / classic schemes): the kernels skip the four zero taps
static bool stencil_narrow(const x3d_tdsops *t)
{
    return t->coeffs[0] == 0.0 && t->coeffs[1] == 0.0 && t->coeffs[7] == 0.0 && t->coeffs[8] == 0.0;
}
// the circulant form (circ_solve) where every operator of the launch offers it.  X3D_NO_CIRC=1: never (A/B)
static bool circ_env_on()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_CIRC"); on = (e && e[0] == '1') ? 0 : 1; }
    return on == 1;
}
// the pair kernels' two circulant solves as one over the pair type (k_ytile_tds_pair<.., CDUAL>): X3D_CIRC_DUAL=1 everywhere,
// default: the z-transforming mode 1 only
static bool circ_dual_on()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_CIRC_DUAL"); on = e ? (e[0] == '1') : 0; }
    return on == 1;
}
// the HALO (decomposed-direction) forms in the circulant form: the same predicate picks the main kernel's solve AND the strip
// kernel's tables, on every rank (the operators are the same everywhere).  X3D_NO_HALO_CIRC=1: the table form (A/B)
static bool halo_circ(const x3d_tdsops *t)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_HALO_CIRC"); on = (e && e[0] == '1') ? 0 : 1; }
    return on && circ_env_on() && t->circ_open_ok && t->uniform && stencil_narrow(t);
}
static bool use_uniform_forms()
{
    static int uni_on = -1;
    if (uni_on < 0) { const char *e = getenv("X3D_NO_UNIFORM"); uni_on = (e && e[0] == '1') ? 0 : 1; }
    return uni_on == 1;
}
static bool pair_halo_circ(const x3d_backend *b, int dir, const x3d_tdsops *ta, const x3d_tdsops *tb)
{
    return b->ring[dir] && use_uniform_forms() && halo_circ(ta) && halo_circ(tb);
}
static bool transeq_halo_circ(const x3d_backend *b, int dir, const x3d_tdsops *der1st, const x3d_tdsops *der2nd)
{
    return b->ring[dir] && use_uniform_forms() && halo_circ(der1st) && halo_circ(der2nd);
}
static bool circ_dual_off()  // X3D_CIRC_DUAL=0: nowhere (A/B)
{
    static int off = -1;
    if (off < 0) { const char *e = getenv("X3D_CIRC_DUAL"); off = (e && e[0] == '0') ? 1 : 0; }
    return off == 1;
}

static bool xscan_ok(const x3d_tdsops *t) { return t->tab.TL != nullptr && (t->tab.Q == 4 || t->tab.Q == 8); }
// the general (FAST = 0) forms of the x kernels also exist for 6 rows per lane (257 .. 384-row pencils)
static bool xscan_ok_gen(const x3d_tdsops *t) { return xscan_ok(t) || (t->tab.TL != nullptr && t->tab.Q == 6); }

int x3d_xscan_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int acc, real_t scale, bool *done)
{
    *done = false;
    if (!xscan_ok_gen(t)) return 0;
    const int Q = t->tab.Q, np = b->ny * b->nz;
    // FAST: periodic-type stencils on a pencil the 64 lanes tile exactly, p2p or v2v (n_rhs == n_tds)
    const bool fast = t->tab.bulk_only && t->n_tds == 64 * Q && t->tab.n_rhs == t->n_tds && Q != 6;
    const bool narrow = stencil_narrow(t);
    const bool circ = fast && narrow && t->uniform && t->circ_ok && circ_env_on();
    const size_t lds = circ ? 0 : sizeof(real_t) * (LT_N(Q) * 64 + CS_N(Q));
    int blocks = (np + 7) / 8;
    const int cap = circ ? 1024 : 768;  // 43 KB of lane tables per 8-wave workgroup: 3 per CU; circulant form: no LDS, 4
    blocks = blocks > cap ? cap : blocks;
    ProfScope ps(b, X3D_K_TDS_FWD, X3D_DIR_X);
#define LAUNCH(Q_, A_, F_, SC_)                                                                                \
    hipLaunchKernelGGL((k_xscan_tds<Q_, A_, F_>), dim3(blocks), dim3(512), lds, b->stream, du, u, xop_of(t), np, \
                       (long)b->nxp, t->n_tds, SC_, t->circ)
#define PICK(Q_)                                                                                               \
    do {                                                                                                       \
        if (circ) { if (acc) LAUNCH(Q_, true, 3, scale); else LAUNCH(Q_, false, 3, 1.0); }                     \
        else if (fast && narrow) { if (acc) LAUNCH(Q_, true, 2, scale); else LAUNCH(Q_, false, 2, 1.0); }      \
        else if (fast) { if (acc) LAUNCH(Q_, true, 1, scale); else LAUNCH(Q_, false, 1, 1.0); }                \
        else { if (acc) LAUNCH(Q_, true, 0, scale); else LAUNCH(Q_, false, 0, 1.0); }                          \
    } while (0)
    if (Q == 8) PICK(8);
    else if (Q == 6) { if (acc) LAUNCH(6, true, 0, scale); else LAUNCH(6, false, 0, 1.0); }
    else PICK(4);
#undef PICK
#undef LAUNCH
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}

template <int Q, bool SAME, bool ACC, int FAST>
static int launch_transeq(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                          const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int np, int blocks,
                          size_t lds, long pitch)
{
    X3D_LDS_OPTIN(b, (k_xscan_transeq<Q, SAME, ACC, FAST>));  // > 64 KB of dynamic LDS needs the opt-in
    hipLaunchKernelGGL((k_xscan_transeq<Q, SAME, ACC, FAST>), dim3(blocks), dim3(FAST ? XS_TQ_THREADS : 512), lds, b->stream, rhs, u, conv, xop_of(t1),
                       xop_of(t2), xop_of(t3), np, pitch, nu);
    X3D_HIP(hipGetLastError());
    return 0;
}

// np pencils of contiguous rows, `pitch` doubles apart: the x pencils of the Cartesian block, or the pencils
// of a transposed copy (tds.hip, transeq_via_x)
int x3d_xscan_transeq_np(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                         const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, int np, long pitch,
                         int dirtag, bool *done)
{
    *done = false;
    if (!xscan_ok_gen(t1) || !xscan_ok_gen(t2) || !xscan_ok_gen(t3) || t1->tab.Q != t2->tab.Q || t1->tab.Q != t3->tab.Q) return 0;
    const int Q = t1->tab.Q;
    const size_t lds = sizeof(real_t) * (3 * LT_N(Q) * 64 + 3 * CS_N(Q));
    if (lds > 160 * 1024) return 0;
    int blocks = (np + 7) / 8;
    blocks = blocks > 256 ? 256 : blocks;  // one 8/12-wave workgroup per CU (129 KB of lane tables in LDS)
    const bool same = u == conv;
    ProfScope ps(b, X3D_K_TRANSEQ_FWD, dirtag, dirtag >= 0);  // dirtag < 0: timed by the caller
    int rc;
    const bool fast = t1->tab.bulk_only && t2->tab.bulk_only && t3->tab.bulk_only && t1->n_tds == 64 * Q && Q != 6;
    const bool narrow = stencil_narrow(t1) && stencil_narrow(t2) && stencil_narrow(t3);
#define GO2(Q_, F_)                                                                                            \
    (same ? (acc ? launch_transeq<Q_, true, true, F_>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds, pitch)       \
                 : launch_transeq<Q_, true, false, F_>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds, pitch))     \
          : (acc ? launch_transeq<Q_, false, true, F_>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds, pitch)      \
                 : launch_transeq<Q_, false, false, F_>(b, rhs, u, conv, nu, t1, t2, t3, np, blocks, lds, pitch)))
#define GO(Q_) (fast ? (narrow ? GO2(Q_, 2) : GO2(Q_, 1)) : GO2(Q_, 0))
    static int p2 = -1;
    if (p2 < 0) { const char *e = getenv("X3D_XSCAN_P1"); p2 = (e && e[0] == '1') ? 0 : 1; }  // same-box A/B: 0.61 -> 0.59 ms
    if (p2 && fast && np % 2 == 0) {
        const int blocks2 = x3d_persistent_blocks(b, (np / 2 + 7) / 8);
#define LP2(Q_, S_, A_, N_)                                                                                    \
        do {                                                                                                   \
            X3D_LDS_OPTIN(b, (k_xscan_transeq2<Q_, S_, A_, N_>));                                              \
            hipLaunchKernelGGL((k_xscan_transeq2<Q_, S_, A_, N_>), dim3(blocks2), dim3(512), lds, b->stream, rhs, u, \
                               conv, xop_of(t1), xop_of(t2), xop_of(t3), np, pitch, nu);                       \
        } while (0)
#define LP2N(Q_, S_, A_) do { if (narrow) LP2(Q_, S_, A_, true); else LP2(Q_, S_, A_, false); } while (0)
#define LP2A(Q_, S_) do { if (acc) LP2N(Q_, S_, true); else LP2N(Q_, S_, false); } while (0)
#define LP2S(Q_) do { if (same) LP2A(Q_, true); else LP2A(Q_, false); } while (0)
        if (Q == 8) LP2S(8); else LP2S(4);
#undef LP2S
#undef LP2A
#undef LP2N
#undef LP2
        X3D_HIP(hipGetLastError());
        *done = true;
        return 0;
    }
    rc = Q == 8 ? GO(8) : (Q == 6 ? GO2(6, 0) : GO(4));
#undef GO
#undef GO2
    if (rc) return rc;
    *done = true;
    return 0;
}

int x3d_xscan_transeq(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                      const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, bool *done)
{
    return x3d_xscan_transeq_np(b, rhs, u, conv, nu, t1, t2, t3, acc, b->ny * b->nz, (long)b->nxp, X3D_DIR_X, done);
}

// true when the branch-free kernels apply to these operators (periodic / halo ends, n = 64 Q rows)
bool x3d_xscan_fast_ok(const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3)
{
    if (!xscan_ok(t1) || !xscan_ok(t2) || !xscan_ok(t3) || t1->tab.Q != t2->tab.Q || t1->tab.Q != t3->tab.Q) return false;
    return t1->tab.bulk_only && t2->tab.bulk_only && t3->tab.bulk_only && t1->n_tds == 64 * t1->tab.Q &&
           t2->n_tds == t1->n_tds && t3->n_tds == t1->n_tds;
}

// K3y launcher: one transeq component along y, straight from / to the Cartesian block
static bool use_ytile()
{
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("X3D_NO_YTILE"), *f = getenv("X3D_NO_VIA_X"), *g = getenv("X3D_NO_XSCAN");
        mode = ((e && e[0] == '1') || (f && f[0] == '1') || (g && g[0] == '1')) ? 0 : 1;  // (those two: two-sweep K1)
    }
    return mode == 1;
}

template <int Q, bool SAME, bool ACC, int FAST>
static int launch_ytile(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                        const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int share12, size_t lds,
                        int dir, const TileEpi *epi)
{
    X3D_LDS_OPTIN(b, (k_ytile_transeq<Q, SAME, ACC, FAST>));
    // y: rows nxp apart, one tile row per z plane; z: rows nxp * nyp apart, one tile row per y
    const long pxy = (long)b->nxp * b->nyp;
    const int ntx = b->nx / 16, ntiles = ntx * (dir == X3D_DIR_Y ? b->nz : b->ny);
    const int blocks = x3d_persistent_blocks(b, ntiles);
    if (epi) {
        // one device slot is enough: copy and kernel are ordered on the backend's stream
        X3D_HIP(hipMemcpyAsync(b->epi_dev, epi, sizeof(TileEpi), hipMemcpyHostToDevice, b->stream));
        X3D_LDS_OPTIN(b, (k_ytile_transeq<Q, SAME, true, FAST, true>));
        hipLaunchKernelGGL((k_ytile_transeq<Q, SAME, true, FAST, true>), dim3(blocks), dim3(1024), lds, b->stream, rhs, u,
                           conv, xop_of(t1), xop_of(t2), xop_of(t3), share12, ntx, ntiles,
                           dir == X3D_DIR_Y ? (long)b->nxp : pxy, dir == X3D_DIR_Y ? pxy : (long)b->nxp, nu,
                           (const TileEpi *)b->epi_dev);
        X3D_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL((k_ytile_transeq<Q, SAME, ACC, FAST>), dim3(blocks), dim3(1024), lds, b->stream, rhs, u, conv,
                       xop_of(t1), xop_of(t2), xop_of(t3), share12, ntx, ntiles,
                       dir == X3D_DIR_Y ? (long)b->nxp : pxy, dir == X3D_DIR_Y ? pxy : (long)b->nxp, nu,
                       (const TileEpi *)nullptr);
    X3D_HIP(hipGetLastError());
    return 0;
}

// would x3d_ytile_transeq take this component?  (used by x3d_transeq_defer: the tile kernel beats the deferred
// transposed-copy route)
bool x3d_ytile_applicable(x3d_backend *b, int dir, const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3)
{
    if (!use_ytile() || !x3d_xscan_fast_ok(t1, t2, t3)) return false;
    const int Q = t1->tab.Q;
    if ((dir == X3D_DIR_Y ? b->ny : b->nz) != 64 * Q || b->nx % 16 != 0) return false;
    if (dir == X3D_DIR_Z) { const char *e = getenv("X3D_NO_ZTILE"); if (e && e[0] == '1') return false; }
    const int share12 = t1->tl_hash == t2->tl_hash;
    return sizeof(real_t) * ((size_t)(share12 ? 2 : 3) * LT_N(Q) * 64 + 16 * (64 * Q + 4)) <= 160 * 1024;
}

static int ytile_transeq_impl(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                              const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc,
                              const TileEpi *epi, bool *done);

int x3d_ytile_transeq(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                      const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, bool *done)
{
    return ytile_transeq_impl(b, dir, rhs, u, conv, nu, t1, t2, t3, acc, nullptr, done);
}

// the component + the stage's linear combination (TileEpi); rhs is x[ipend]
int x3d_ytile_transeq_lincomb(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                              const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, real_t *y,
                              const real_t *base, int nterm, const real_t *c, real_t *const *x, int ipend, int store,
                              bool *done)
{
    TileEpi e;
    e.y = y; e.base = base; e.n = nterm; e.ipend = ipend; e.store = store;
    for (int k = 0; k < 5; k++) { e.x[k] = k < nterm ? x[k] : x[0]; e.c[k] = k < nterm ? c[k] : 0.0; }
    return ytile_transeq_impl(b, dir, rhs, u, conv, nu, t1, t2, t3, 1, &e, done);
}

static int ytile_transeq_impl(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                              const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc,
                              const TileEpi *epi, bool *done)
{
    *done = false;
    if (!use_ytile() || !x3d_xscan_fast_ok(t1, t2, t3)) return 0;
    const int Q = t1->tab.Q;
    if ((dir == X3D_DIR_Y ? b->ny : b->nz) != 64 * Q || b->nx % 16 != 0) return 0;
    if (dir == X3D_DIR_Z) {
        static int zt = -1;  // z tiles: rows 2 MB apart; 1.03 ms per component against 1.6 through transposed copies
        if (zt < 0) { const char *e = getenv("X3D_NO_ZTILE"); zt = (e && e[0] == '1') ? 0 : 1; }
        if (!zt) return 0;
    }
    const int share12 = t1->tl_hash == t2->tl_hash;
    const size_t lds = sizeof(real_t) * ((size_t)(share12 ? 2 : 3) * LT_N(Q) * 64 + 16 * (64 * Q + 4));
    if (lds > 160 * 1024) return 0;
    const bool same = u == conv;
    const bool narrow = stencil_narrow(t1) && stencil_narrow(t2) && stencil_narrow(t3);
    ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
    int rc;
#define GO2(Q_, F_)                                                                                     \
    (same ? (acc ? launch_ytile<Q_, true, true, F_>(b, rhs, u, conv, nu, t1, t2, t3, share12, lds, dir, epi)       \
                 : launch_ytile<Q_, true, false, F_>(b, rhs, u, conv, nu, t1, t2, t3, share12, lds, dir, epi))     \
          : (acc ? launch_ytile<Q_, false, true, F_>(b, rhs, u, conv, nu, t1, t2, t3, share12, lds, dir, epi)      \
                 : launch_ytile<Q_, false, false, F_>(b, rhs, u, conv, nu, t1, t2, t3, share12, lds, dir, epi)))
#define GO(Q_) (narrow ? GO2(Q_, 2) : GO2(Q_, 1))
    rc = Q == 8 ? GO(8) : GO(4);
#undef GO
#undef GO2
    if (rc) return rc;
    *done = true;
    return 0;
}

// tile range of a launch: planes [other0, other0 + nother) of the coordinate the tile rows run over (z for y
// pencils, y for z pencils); nother < 0: all of them
static void tile_range(const x3d_backend *b, int dir, int other0, int nother, int ntx, int *tile0, int *ntiles)
{
    const int nall = dir == X3D_DIR_Y ? b->nz : b->ny;
    if (nother < 0) { other0 = 0; nother = nall; }
    *tile0 = other0 * ntx;
    *ntiles = nother * ntx;
}


// K3y pair launcher: see k_ytile_tds_pair; y and z (rows nxp or nxp * nyp apart, as for k_ytile_transeq).
// mode 2: out1 = A(in1) only.  halo != null: decomposed direction (TileHalo: nf = 1 or 2 inputs, nb operators)
int x3d_ytile_tds_pair(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                       const x3d_tdsops *ta, const x3d_tdsops *tb, const TileHalo *halo, int other0, int nother,
                       bool *done)
{
    *done = false;
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_TDS_PAIR"); on = (e && e[0] == '1') ? 0 : 1; }
    if (!on || !use_ytile() || !xscan_ok(ta) || !xscan_ok(tb) || ta->tab.Q != tb->tab.Q) return 0;
    const int Q = ta->tab.Q;
    auto fast = [&](const x3d_tdsops *t) {
        return t->tab.bulk_only && t->n_tds == 64 * Q && t->tab.n_rhs == t->n_tds;
    };
    if (!fast(ta) || !fast(tb) || (dir == X3D_DIR_Y ? b->ny : b->nz) != 64 * Q || b->nx % 16 != 0) return 0;
    if (dir == X3D_DIR_Z) { const char *e = getenv("X3D_NO_ZTILE"); if (e && e[0] == '1') return 0; }
    const bool narrow = stencil_narrow(ta) && stencil_narrow(tb);
    static int uni_on = -1;
    if (uni_on < 0) { const char *e = getenv("X3D_NO_UNIFORM"); uni_on = (e && e[0] == '1') ? 0 : 1; }
    const bool uni = uni_on && ta->uniform && tb->uniform;
    const bool circ = narrow && uni && circ_env_on() && (halo ? pair_halo_circ(b, dir, ta, tb) : (ta->circ_ok && tb->circ_ok));
    const size_t lds = sizeof(real_t) * ((size_t)(circ ? 0 : (mode == 2 ? 1 : 2) * LT_N(Q) * 64) + 16 * (64 * Q + 4) + (halo ? 128 + 64 : 0));
    if (lds > 160 * 1024) return 0;
    const long pxy = (long)b->nxp * b->nyp;
    const int ntx = b->nx / 16;
    int tile0, ntiles;
    tile_range(b, dir, other0, nother, ntx, &tile0, &ntiles);
    const int permn = b->pair_yperm;  // (whole blocks of z pencils only: the interleave pairs rows across the block)
    if (permn > 0 && (dir != X3D_DIR_Z || mode == 2 || tile0 != 0 || (ntiles > 0 && ntiles != ntx * b->ny))) return 0;
    if (ntiles <= 0) { *done = true; return 0; }
    const long rstride = dir == X3D_DIR_Y ? (long)b->nxp : pxy, ostride = dir == X3D_DIR_Y ? pxy : (long)b->nxp;
    if (128 * rstride * X3D_RB >= (1L << 32)) return 0;  // tile_row's 32-bit lane offset
    static int cap = -1;
    if (cap < 0) { const char *e = getenv("X3D_TILE_BLOCKS"); cap = e ? atoi(e) : 256; }
    const int blocks = x3d_persistent_blocks(b, ntiles > cap ? cap : ntiles);
    const TileHalo th = halo ? *halo : TileHalo{nullptr, nullptr, 0, 0, 0, 0, 0};
    ProfScope ps(b, X3D_K_TDS_FWD, dir);
#define GOD(Q_, M_, N_, H_, U_, C_, D_)                                                                         \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_ytile_tds_pair<Q_, M_, N_, H_, false, U_, C_, D_>));                                \
        hipLaunchKernelGGL((k_ytile_tds_pair<Q_, M_, N_, H_, false, U_, C_, D_>), dim3(blocks), dim3(1024), lds, b->stream, out1, out2, \
                           in1, in2, xop_of(ta), xop_of(tb), ntx, tile0, ntiles, rstride, ostride, th, permn,   \
                           ZfArg{}, ta->circ, tb->circ);                                                        \
    } while (0)
#define GOC(Q_, M_, N_, H_, U_, C_) do { if ((C_) && (M_) != 2 && circ_dual_on()) GOD(Q_, M_, N_, H_, U_, C_, C_); else GOD(Q_, M_, N_, H_, U_, C_, false); } while (0)
#define GO(Q_, M_, N_, H_, U_) GOC(Q_, M_, N_, H_, U_, false)
#define GOH(Q_, M_, N_, U_) do { if (halo) GO(Q_, M_, N_, true, U_); else GO(Q_, M_, N_, false, U_); } while (0)
#define GON(Q_, M_) do { if (circ && halo) GOD(Q_, M_, true, true, true, true, false); else if (circ) GOC(Q_, M_, true, false, true, true); else if (narrow && uni) GOH(Q_, M_, true, true); else if (narrow) GOH(Q_, M_, true, false); else GOH(Q_, M_, false, false); } while (0)
#define GOM(Q_) do { if (mode == 0) GON(Q_, 0); else if (mode == 1) GON(Q_, 1); else GON(Q_, 2); } while (0)
    if (Q == 8) GOM(8); else GOM(4);
#undef GOM
#undef GON
#undef GOC
#undef GOD
#undef GOH
#undef GO
    if (halo) b->n_halo++;
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}

// these two operators' z pair is served by the z-transforming form of the tile kernel on this backend's blocks
bool x3d_zfirst_pairs_ok(const x3d_backend *b, const x3d_tdsops *ta, const x3d_tdsops *tb)
{
    if (!use_ytile() || !xscan_ok(ta) || !xscan_ok(tb) || ta->tab.Q != 8 || tb->tab.Q != 8) return false;
    auto fast = [&](const x3d_tdsops *t) { return t->tab.bulk_only && t->n_tds == 512 && t->tab.n_rhs == t->n_tds; };
    if (!fast(ta) || !fast(tb) || b->nz != 512 || b->nx % 16 != 0) return false;
    if (sizeof(real_t) * ((size_t)2 * LT_NC(8) * 64 + ZF_AREA_DOUBLES + 512) > 160 * 1024) return false;
    return 128 * (long)b->nxp * b->nyp * X3D_RB < (1L << 32);
}

int x3d_zfpair8(x3d_backend *b, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                const x3d_tdsops *ta, const x3d_tdsops *tb, const ZfArg &zf, bool *done);  // zfpair8.hip
// the z pairs next to the z-first Poisson solve (k_ytile_tds_pair<.., ZF>): mode 0: A(in1) + B(in2) -> spectrum,
// mode 1: spectrum -> out1 = A(p), out2 = B(p); whole blocks of 512^3
// y0, nyr: the tiles of the y rows [y0, y0 + nyr) only (nyr < 0: all) -- csrc/sfftz.hip cuts a solve into groups of y rows
// so that a group's exchange runs beside the next group's pair
int x3d_ytile_tds_pair_zf(x3d_backend *b, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                          const x3d_tdsops *ta, const x3d_tdsops *tb, const ZfArg &zf, bool *done, int y0, int nyr)
{
    *done = false;
    // (the spectrum may hold fewer rows than the block: the channel's 256 cell rows of 257 -- the pair then leaves the last
    //  row alone, as the solver's own transforms do)
    if (mode < 0 || mode > 1 || zf.ny > b->ny || (zf.permn == 0 && zf.ny != b->ny) || !x3d_zfirst_pairs_ok(b, ta, tb)) return 0;
    if (nyr < 0) { y0 = 0; nyr = zf.ny; }
    X3D_REQUIRE(y0 >= 0 && nyr >= 0 && y0 + nyr <= zf.ny, "tds_pair (z-first): rows [%d, %d) of %d", y0, y0 + nyr, zf.ny);
    X3D_REQUIRE(zf.permn == 0 || (zf.permn == zf.ny && y0 == 0 && nyr == zf.ny), "tds_pair (z-first): interleaved rows, whole blocks only");
    if (nyr == 0) { *done = true; return 0; }
    if (y0 == 0 && nyr == zf.ny) {  // whole blocks: the 8-pencil form, two workgroups per CU (zfpair8.hip)
        if (int rc = x3d_zfpair8(b, mode, out1, out2, in1, in2, ta, tb, zf, done)) return rc;
        if (*done) return 0;
    }
    const bool narrow = stencil_narrow(ta) && stencil_narrow(tb);
    static int uni_on = -1;
    if (uni_on < 0) { const char *e = getenv("X3D_NO_UNIFORM"); uni_on = (e && e[0] == '1') ? 0 : 1; }
    const bool uni = uni_on && ta->uniform && tb->uniform;
    const bool circ = narrow && uni && circ_env_on() && ta->circ_ok && tb->circ_ok;
    const size_t lds = sizeof(real_t) * ((size_t)(circ ? 0 : 2 * LT_NC(8) * 64) + ZF_AREA_DOUBLES + 512);
    const long pxy = (long)b->nxp * b->nyp;
    const int ntx = b->nx / 16, ntiles = ntx * nyr, tile0 = ntx * y0;
    static int cap = -1;
    if (cap < 0) { const char *e = getenv("X3D_TILE_BLOCKS"); cap = e ? atoi(e) : 256; }
    const int blocks = x3d_persistent_blocks(b, ntiles > cap ? cap : ntiles);
    const TileHalo th{nullptr, nullptr, 0, 0, 0, 0, 0};
    ProfScope ps(b, X3D_K_TDS_FWD, X3D_DIR_Z);
#define GOD(M_, N_, U_, C_, D_)                                                                                 \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_ytile_tds_pair<8, M_, N_, false, true, U_, C_, D_>));                               \
        hipLaunchKernelGGL((k_ytile_tds_pair<8, M_, N_, false, true, U_, C_, D_>), dim3(blocks), dim3(1024), lds, b->stream, out1, \
                           out2, in1, in2, xop_of(ta), xop_of(tb), ntx, tile0, ntiles, pxy, (long)b->nxp, th, 0, zf, ta->circ, tb->circ); \
    } while (0)
// (the dual solve where it measured faster: the z-transforming mode 1, 0.745 -> 0.716 ms; mode 0 0.782 -> 0.794: not)
#define GO(M_, N_, U_, C_) do { if ((C_) && ((M_) == 1 || circ_dual_on()) && !circ_dual_off()) GOD(M_, N_, U_, C_, C_); else GOD(M_, N_, U_, C_, false); } while (0)
#define GOU(M_) do { if (circ) GO(M_, true, true, true); else if (narrow && uni) GO(M_, true, true, false); else if (narrow) GO(M_, true, false, false); else GO(M_, false, false, false); } while (0)
    if (mode == 0) GOU(0); else GOU(1);
#undef GOU
#undef GO
#undef GOD
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}

// rows 1..ws and n-we+1..n carry more than 2^-60 of the reduced system's coupling (tds.hip, x3d_tdsops_create)
int x3d_tds_halo_fix(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *brecv,
                     const x3d_tdsops *ta, const x3d_tdsops *tb)
{
    const PencilGeom g = x3d_geom(b, dir);
    // (the circulant HALO form: the same launch on the second set of row records -- the choice x3d_ytile_tds_pair made)
    const bool hc = pair_halo_circ(b, dir, ta, mode == 2 ? ta : tb);
    const TdsTab &tta = hc ? ta->tabc : ta->tab, &ttb = hc ? tb->tabc : tb->tab;
    const int wsa = hc ? ta->halo_ws_c : ta->halo_ws, wsb = hc ? tb->halo_ws_c : tb->halo_ws;
    const int wea = hc ? ta->halo_we_c : ta->halo_we, web = hc ? tb->halo_we_c : tb->halo_we;
    const int ws = mode == 2 ? wsa : (wsa > wsb ? wsa : wsb);
    const int we = mode == 2 ? wea : (wea > web ? wea : web);
    dim3 grid((g.np + 255) / 256, ws + we >= ta->n_tds ? 1 : 2);
    const int ws1 = grid.y == 1 ? ta->n_tds : ws;
    const int permn = b->pair_yperm;
    X3D_REQUIRE(permn == 0 || (dir == X3D_DIR_Z && mode != 2), "tds_halo_fix: interleaved rows are for z pairs");
    ProfScope ps(b, X3D_K_TDS_BWD, dir);
    if (mode == 0) hipLaunchKernelGGL(k_tds_halo_fix<0>, grid, dim3(256), 0, b->stream, out1, out2, brecv, tta, ttb, g, ws1, we, permn);
    else if (mode == 1) hipLaunchKernelGGL(k_tds_halo_fix<1>, grid, dim3(256), 0, b->stream, out1, out2, brecv, tta, ttb, g, ws1, we, permn);
    else hipLaunchKernelGGL(k_tds_halo_fix<2>, grid, dim3(256), 0, b->stream, out1, out2, brecv, tta, tta, g, ws1, we, 0);
    X3D_HIP(hipGetLastError());
    return 0;
}

int x3d_transeq_halo_fix_launch(x3d_backend *b, int dir, real_t *const r[3], const real_t *conv, real_t nu,
                                const real_t *brecv, const x3d_tdsops *der1st, const x3d_tdsops *der2nd)
{
    const PencilGeom g = x3d_geom(b, dir);
    const bool hc = transeq_halo_circ(b, dir, der1st, der2nd);  // (the choice x3d_ytile_transeq3 made)
    const TdsTab &tt1 = hc ? der1st->tabc : der1st->tab, &tt2 = hc ? der2nd->tabc : der2nd->tab;
    const int ws1_ = hc ? der1st->halo_ws_c : der1st->halo_ws, ws2_ = hc ? der2nd->halo_ws_c : der2nd->halo_ws;
    const int we1_ = hc ? der1st->halo_we_c : der1st->halo_we, we2_ = hc ? der2nd->halo_we_c : der2nd->halo_we;
    const int ws = ws1_ > ws2_ ? ws1_ : ws2_;
    const int we = we1_ > we2_ ? we1_ : we2_;
    dim3 grid((g.np + 255) / 256, ws + we >= der1st->n_tds ? 1 : 2);
    const int ws1 = grid.y == 1 ? der1st->n_tds : ws;
    ProfScope ps(b, X3D_K_TRANSEQ_BWD, dir);
    hipLaunchKernelGGL(k_transeq_halo_fix, grid, dim3(256), 0, b->stream, r[0], r[1], r[2], conv, brecv, tt1,
                       tt2, g, nu, ws1, we);
    X3D_HIP(hipGetLastError());
    return 0;
}

// pencils per wave of the tile kernels: 2 where a 16-pencil tile would move 64-byte row segments (FP32), local form, nx a
// multiple of 32 (X3D_NO_NPW2=1: always 1).  256-row FP64 pencils (BASELINE configs[1]) with X3D_NPW2_256=1 only: measured
// SLOWER there (5.24 against 5.04 ms per step at 256^3, k_ytile_transeq3<4> at 0.46 against 0.54 of the peak) -- a 256-row
// solve costs what a 512-row one does less the per-row work (the scans, the reduced system), so two of them per wave double
// the solve phase for the bytes of one 512-row tile: the 256-row tile is short of on-chip time, not of bytes in flight
static int ytile_npw(const x3d_backend *b, int Q, bool halo)
{
    static int on = -1, on256 = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_NPW2"); on = (e && e[0] == '1') ? 0 : 1; }
    if (on256 < 0) { const char *e = getenv("X3D_NPW2_256"); on256 = (e && e[0] == '1') ? 1 : 0; }
    return (on && !halo && b->nx % 32 == 0 && (X3D_RB == 4 || (Q == 4 && on256))) ? 2 : 1;
}

static bool ytile_circ(const x3d_tdsops *a, const x3d_tdsops *a2, const x3d_tdsops *c, const x3d_tdsops *c2)
{
    return circ_env_on() && a->circ_ok && a2->circ_ok && c->circ_ok && c2->circ_ok;
}

// K3y, three components in one launch (k_ytile_transeq3); f[0] is the advecting component.
// halo != null: decomposed direction (TileHalo: nf = 3 fields in the order f[0..2], nb = 9 boundary values)
int x3d_ytile_transeq3(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                       const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                       const x3d_tdsops *der2nd_sym, int acc, const TileHalo *halo, int other0, int nother, bool *done)
{
    *done = false;
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_TILE3"); on = (e && e[0] == '1') ? 0 : 1; }
    if (!on || !x3d_ytile_applicable(b, dir, der1st, der1st_sym, der2nd)) return 0;
    if (der1st->tl_hash != der1st_sym->tl_hash || der2nd->tl_hash != der2nd_sym->tl_hash) return 0;
    const int Q = der1st->tab.Q;
    const bool narrow = stencil_narrow(der1st) && stencil_narrow(der2nd);
    static int uni_on = -1;
    if (uni_on < 0) { const char *e = getenv("X3D_NO_UNIFORM"); uni_on = (e && e[0] == '1') ? 0 : 1; }
    const bool uni = uni_on && der1st->uniform && der1st_sym->uniform && der2nd->uniform && der2nd_sym->uniform;
    const int npw = (narrow && uni) ? ytile_npw(b, Q, halo != nullptr) : 1;  // (two pencils per wave: the uniform-grid forms)
    const bool circ = narrow && uni && (halo ? transeq_halo_circ(b, dir, der1st, der2nd) : ytile_circ(der1st, der1st_sym, der2nd, der2nd_sym));
    const size_t lds = sizeof(real_t) * ((size_t)(circ ? 0 : 2 * LT_N(Q) * 64) + 16 * npw * (64 * Q + 4) + (halo ? 256 + 288 : 0));
    if (lds > 160 * 1024) return 0;
    const long pxy = (long)b->nxp * b->nyp;
    const int ntx = b->nx / (16 * npw);
    int tile0, ntiles;
    tile_range(b, dir, other0, nother, ntx, &tile0, &ntiles);
    if (ntiles <= 0) { *done = true; return 0; }
    const int blocks = x3d_persistent_blocks(b, ntiles);
    const long rstride = dir == X3D_DIR_Y ? (long)b->nxp : pxy, ostride = dir == X3D_DIR_Y ? pxy : (long)b->nxp;
    if (128 * rstride * X3D_RB >= (1L << 32)) return 0;  // tile_row's 32-bit lane offset
    const TileHalo th = halo ? *halo : TileHalo{nullptr, nullptr, 0, 0, 0, 0, 0};
    // (profiler: three components = three "forward" launches of this direction, in one kernel)
#define GOW(Q_, A_, N_, H_, U_, W_, C_)                                                                         \
    do {                                                                                                        \
        /* (the pair solve P12 only in the local form: with the HALO extras it spills, 128 VGPRs + 124 bytes) */ \
        X3D_LDS_OPTIN(b, (k_ytile_transeq3<Q_, A_, N_, H_, U_, (U_ && !(H_)), false, W_, C_>));                 \
        hipLaunchKernelGGL((k_ytile_transeq3<Q_, A_, N_, H_, U_, (U_ && !(H_)), false, W_, C_>), dim3(blocks), dim3(1024), lds, b->stream, r[0], r[1], r[2], \
                           f[0], f[1], f[2], xop_of(der1st), xop_of(der2nd), ntx, tile0, ntiles, rstride, ostride, nu, th, \
                           (const TileEpi *)nullptr, der1st->circ, der2nd->circ);                               \
    } while (0)
#define GOC(Q_, A_, W_) do { if (circ) GOW(Q_, A_, true, false, true, W_, true); else GOW(Q_, A_, true, false, true, W_, false); } while (0)
#define GO(Q_, A_, N_, H_, U_) do { if ((N_) && (U_) && !(H_)) { if (npw == 2) GOC(Q_, A_, 2); else GOC(Q_, A_, 1); } else if ((N_) && (U_) && (H_) && circ) GOW(Q_, A_, true, true, true, 1, true); else GOW(Q_, A_, N_, H_, U_, 1, false); } while (0)
#define GOH(Q_, A_, N_, U_) do { if (halo) GO(Q_, A_, N_, true, U_); else GO(Q_, A_, N_, false, U_); } while (0)
#define GON(Q_, A_) do { if (narrow && uni) GOH(Q_, A_, true, true); else if (narrow) GOH(Q_, A_, true, false); else GOH(Q_, A_, false, false); } while (0)
#define GOA(Q_) do { if (acc) GON(Q_, true); else GON(Q_, false); } while (0)
    {
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
        if (Q == 8) GOA(8); else GOA(4);
    }
#undef GOA
#undef GON
#undef GOH
#undef GO
#undef GOC
#undef GOW
    X3D_HIP(hipGetLastError());
    b->n_tq3++;
    if (halo) b->n_halo++;
    if (b->prof) {  // count the launch as three components (bench.py divides the direction's time by the count)
        for (int k = 0; k < 2; k++) { ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir); }
    }
    *done = true;
    return 0;
}

// ... with the RK / AB stage of the three variables in the store phases (k_ytile_transeq3<.., EPI>): epi[c] describes
// component c's combination (c = 0: the advecting component).  Local periodic uniform-grid pencils only (the bench's form);
// *done = false: not served, nothing was launched
int x3d_ytile_transeq3_epi(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                           const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                           const x3d_tdsops *der2nd_sym, const TileEpi epi[3], bool *done)
{
    *done = false;
    static int on = -1;
    if (on < 0) { const char *e = getenv("X3D_NO_TILE3"); on = (e && e[0] == '1') ? 0 : 1; }
    if (!on || !x3d_ytile_applicable(b, dir, der1st, der1st_sym, der2nd)) return 0;
    if (der1st->tl_hash != der1st_sym->tl_hash || der2nd->tl_hash != der2nd_sym->tl_hash) return 0;
    const int Q = der1st->tab.Q;
    const int npw = ytile_npw(b, Q, false);
    // (FP64, 512 rows: the circulant form of THIS instantiation spills 14 VGPRs and runs at the table form's 3.1 ms; it is
    //  taken all the same -- the launch must give the bits of the plain z launch followed by the stage)
    const bool circ = ytile_circ(der1st, der1st_sym, der2nd, der2nd_sym);
    const size_t lds = sizeof(real_t) * ((size_t)(circ ? 0 : 2 * LT_N(Q) * 64) + 16 * npw * (64 * Q + 4));
    const bool narrow = stencil_narrow(der1st) && stencil_narrow(der2nd);
    const bool uni = der1st->uniform && der1st_sym->uniform && der2nd->uniform && der2nd_sym->uniform;
    if (lds > 160 * 1024 || !narrow || !uni) return 0;
    const long pxy = (long)b->nxp * b->nyp;
    const int ntx = b->nx / (16 * npw), ntiles = ntx * (dir == X3D_DIR_Y ? b->nz : b->ny);
    const int blocks = x3d_persistent_blocks(b, ntiles);
    const long rstride = dir == X3D_DIR_Y ? (long)b->nxp : pxy, ostride = dir == X3D_DIR_Y ? pxy : (long)b->nxp;
    if (128 * rstride * X3D_RB >= (1L << 32)) return 0;
    static_assert(3 * sizeof(TileEpi) <= 512, "epi_dev slot");
    X3D_HIP(hipMemcpyAsync(b->epi_dev, epi, 3 * sizeof(TileEpi), hipMemcpyHostToDevice, b->stream));
    const TileHalo th{nullptr, nullptr, 0, 0, 0, 0, 0};
    {
        // (timed under direction slot 0: x3d_prof_get(kind, 3) stays the plain z launches, the sum over slots has these)
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, 0);
#define GOE(Q_, W_, C_)                                                                                         \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_ytile_transeq3<Q_, true, true, false, true, true, true, W_, C_>));                  \
        hipLaunchKernelGGL((k_ytile_transeq3<Q_, true, true, false, true, true, true, W_, C_>), dim3(blocks), dim3(1024), lds, b->stream, \
                           r[0], r[1], r[2], f[0], f[1], f[2], xop_of(der1st), xop_of(der2nd), ntx, 0, ntiles, rstride, \
                           ostride, nu, th, (const TileEpi *)b->epi_dev, der1st->circ, der2nd->circ);           \
    } while (0)
#define GOEC(Q_, W_) do { if (circ) GOE(Q_, W_, true); else GOE(Q_, W_, false); } while (0)
        if (Q == 8) { if (npw == 2) GOEC(8, 2); else GOEC(8, 1); }
        else { if (npw == 2) GOEC(4, 2); else GOEC(4, 1); }
#undef GOEC
#undef GOE
    }
    X3D_HIP(hipGetLastError());
    b->n_tq3++;
    if (b->prof) {
        for (int k = 0; k < 2; k++) { ProfScope ps(b, X3D_K_TRANSEQ_FWD, 0); }
    }
    *done = true;
    return 0;
}

// transeq_x in one launch (k_xscan_transeq2x3); f[0] is the advecting component.  upd_g != null: the pending
// correction f[c] += scale * tds_solve(upd_g[c]) with op_s (c = 0) / op_i (c = 1, 2) is applied first (UPD form)
int x3d_xscan_transeq3(x3d_backend *b, real_t *const r[3], const real_t *const f[3], real_t nu,
                       const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                       const x3d_tdsops *der2nd_sym, int acc, const real_t *const *upd_g, const x3d_tdsops *op_s,
                       const x3d_tdsops *op_i, real_t scale, real_t omega, const real_t *ushift, bool *done)
{
    *done = false;
    if ((omega != 0.0 || ushift) && (acc || upd_g)) return 0;
    static int on = -1;
    if (on < 0) {
        const char *names[4] = {"X3D_NO_TILE3", "X3D_XSCAN_P1", "X3D_NO_XSCAN", "X3D_XDIR_GENERIC"};
        on = 1;
        for (const char *nm : names) { const char *e = getenv(nm); if (e && e[0] == '1') on = 0; }
    }
    if (!on || !x3d_xscan_fast_ok(der1st, der1st_sym, der2nd) || !x3d_xscan_fast_ok(der1st_sym, der1st, der2nd_sym)) return 0;
    if (der1st->tl_hash != der1st_sym->tl_hash || der2nd->tl_hash != der2nd_sym->tl_hash) return 0;
    const int Q = der1st->tab.Q, np = b->ny * b->nz;
    if (b->nx != 64 * Q || np % 2) return 0;
    const bool upd = upd_g != nullptr;
    if (upd) {
        auto ok = [&](const x3d_tdsops *t) {
            return xscan_ok(t) && t->tab.Q == Q && t->tab.bulk_only && t->n_tds == 64 * Q && t->tab.n_rhs == t->n_tds;
        };
        if (!ok(op_s) || !ok(op_i)) return 0;
    }
    const bool narrow = stencil_narrow(der1st) && stencil_narrow(der2nd) && (!upd || (stencil_narrow(op_s) && stencil_narrow(op_i)));
    static int xcirc = -1;
    if (xcirc < 0) { const char *e = getenv("X3D_NO_XCIRC"); xcirc = (e && e[0] == '1') ? 0 : 1; }
    const bool circ = xcirc && circ_env_on() && narrow && der1st->circ_ok && der1st_sym->circ_ok && der2nd->circ_ok && der2nd_sym->circ_ok &&
                      (!upd || (op_s->circ_ok && op_i->circ_ok));
    const size_t lds = circ ? 0 : sizeof(real_t) * 64 * (upd ? 3 * LT_NC(Q) + LT_N(Q) : 2 * LT_N(Q));
    if (lds > 160 * 1024) return 0;
    const int blocks = x3d_persistent_blocks(b, (np / 2 + 7) / 8);
    XUpd xu{};
    if (upd) { xu.g[0] = upd_g[0]; xu.g[1] = upd_g[1]; xu.g[2] = upd_g[2]; xu.scale = scale; }
    xu.omega = omega; xu.ushift = ushift;
    const x3d_tdsops *ts = upd ? op_s : der1st, *ti = upd ? op_i : der1st;
    const Circ4 c4{{der1st->circ, der2nd->circ, ts->circ, ti->circ}};
#define GOX(Q_, A_, N_, U_, C_, X_)                                                                             \
    do {                                                                                                        \
        X3D_LDS_OPTIN(b, (k_xscan_transeq2x3<Q_, A_, N_, U_, C_, X_>));                                         \
        hipLaunchKernelGGL((k_xscan_transeq2x3<Q_, A_, N_, U_, C_, X_>), dim3(blocks), dim3(512), lds, b->stream, r[0], r[1], \
                           r[2], (real_t *)f[0], (real_t *)f[1], (real_t *)f[2], xop_of(der1st), xop_of(der2nd), np,  \
                           (long)b->nxp, nu, xu, xop_of(ts), xop_of(ti), c4);                                   \
    } while (0)
#define GO(Q_, A_, N_, U_, C_) do { if ((N_) && circ) GOX(Q_, A_, true, U_, C_, true); else GOX(Q_, A_, N_, U_, C_, false); } while (0)
#define GOU(Q_, A_, N_) do { if (upd) GO(Q_, A_, N_, true, false); else GO(Q_, A_, N_, false, false); } while (0)
#define GON(Q_, A_) do { if (narrow) GOU(Q_, A_, true); else GOU(Q_, A_, false); } while (0)
#define GOA(Q_)                                                                                                 \
    do {                                                                                                        \
        if (omega != 0.0 || ushift) { if (narrow) GO(Q_, false, true, false, true); else GO(Q_, false, false, false, true); } \
        else if (acc) GON(Q_, true);                                                                            \
        else GON(Q_, false);                                                                                    \
    } while (0)
    {
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X);
        if (Q == 8) GOA(8); else GOA(4);
    }
#undef GOA
#undef GON
#undef GOU
#undef GO
#undef GOX
    X3D_HIP(hipGetLastError());
    b->n_tq3++;
    if (upd) b->n_upd++;
    if (b->prof) {  // three components (bench.py divides the direction's time by the count)
        for (int k = 0; k < 2; k++) { ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X); }
    }
    *done = true;
    return 0;
}

// du = tdsops(y) with y = base + sum c_k x_k formed (and stored) by the same kernel; x direction
int x3d_xscan_tds_lincomb(x3d_backend *b, real_t *du, const x3d_tdsops *t, real_t *y, const real_t *base, int nterm,
                          const real_t *c, const real_t *const *x, const real_t *wall, bool *done)
{
    *done = false;
    static int on = -1;
    if (on < 0) {
        const char *names[3] = {"X3D_NO_TDS_LINCOMB", "X3D_NO_XSCAN", "X3D_XDIR_GENERIC"};
        on = 1;
        for (const char *nm : names) { const char *e = getenv(nm); if (e && e[0] == '1') on = 0; }
    }
    if (!on || !xscan_ok(t)) return 0;
    const int Q = t->tab.Q;
    if (!(t->tab.bulk_only && t->n_tds == 64 * Q && t->tab.n_rhs == t->n_tds && b->nx == 64 * Q)) return 0;
    const int np = b->ny * b->nz;
    const bool narrow = stencil_narrow(t);
    const bool circ = narrow && t->uniform && t->circ_ok && circ_env_on();  // (the same choice as x3d_xscan_tds: same bits)
    const size_t lds = circ ? 0 : sizeof(real_t) * LT_N(Q) * 64;
    int blocks = (np + 7) / 8;
    const int cap = circ ? 1024 : 768;
    blocks = blocks > cap ? cap : blocks;
    LinRows lr;
    lr.y = y; lr.base = base; lr.n = nterm; lr.wall = wall; lr.ny = b->ny;
    for (int k = 0; k < 5; k++) { lr.x[k] = k < nterm ? x[k] : x[0]; lr.c[k] = k < nterm ? c[k] : 0.0; }
    ProfScope ps(b, X3D_K_TDS_FWD, X3D_DIR_X);
#define GO(Q_, N_, C_) hipLaunchKernelGGL((k_xscan_tds_lin<Q_, N_, C_>), dim3(blocks), dim3(512), lds, b->stream, du, lr, xop_of(t), np, (long)b->nxp, t->circ)
    if (Q == 8) { if (circ) GO(8, true, true); else if (narrow) GO(8, true, false); else GO(8, false, false); }
    else { if (circ) GO(4, true, true); else if (narrow) GO(4, true, false); else GO(4, false, false); }
#undef GO
    X3D_HIP(hipGetLastError());
    *done = true;
    return 0;
}
