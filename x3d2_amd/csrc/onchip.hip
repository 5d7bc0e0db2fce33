// Single-pass tds_solve for a non-decomposed direction, the whole pencil on chip (kernel family K1e).
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
// (The first version, one 16-wave workgroup per CU with scalar row tables, is kept in
// scratch/kernels/onchip_k1c.hip: it only matched the two-sweep pair.)
#include "common.h"

// ---------------------------------------------------------------------------------------------
// K1e: a workgroup is 8 waves and owns 32 pencils, TWO workgroups per CU: the
// two halves of a wave are two different 32-row chunks of the same 32 pencils (lane = x + 32 h,
// chunk = 2 wave + h), every half-wave row access is 256 contiguous bytes.  The row index now differs
// between the halves, so the row tables cannot come from scalar loads: they are staged in LDS once per
// workgroup ([table][row], 2 distinct addresses per wave access = broadcast reads).
// ---------------------------------------------------------------------------------------------
#define K1E_M 32
#define K1E_TAB 8  // F A PF HB QB SA SC ST

struct Coef9 { real_t c[9]; };

// (wave-uniform pointer) + (32-bit per-lane BYTE offset): the form the backend turns into
// global_load/store v, v_off, s[base:base+1].  With an element offset it cannot prove that 8*off fits 32
// bits and builds a 64-bit address in two VGPRs per access (77 v_lshl_add_u64 in this kernel).
// (plain accesses: nontemporal ones measured 4 % slower here, A/B on the same box)
__device__ __forceinline__ real_t ldg(const real_t *base, unsigned boff)
{
    return *reinterpret_cast<const real_t *>(reinterpret_cast<const char *>(base) + boff);
}
__device__ __forceinline__ void stg(real_t *base, unsigned boff, real_t v)
{
    *reinterpret_cast<real_t *>(reinterpret_cast<char *>(base) + boff) = v;
}

// M = 32: 512-row pencils, M = 16: 256-row pencils (16 chunks either way)
template <bool ACC, int M, bool NARROW>
__global__ void __launch_bounds__(512, 4)  // 4 waves per SIMD = two workgroups per CU, <= 128 VGPRs
    k_tds_onchip2(real_t *__restrict__ du, const real_t *__restrict__ u, TdsTab t, PencilGeom g, real_t scale,
                  Coef9 cf)
{
    // periodic-type operator on pencils of exactly 16 * M rows (n_tds = n_rhs, bulk stencil everywhere):
    // no row guards, no boundary stencils, wrap-around halos by index arithmetic
    extern __shared__ real_t lds[];  // K1E_TAB tables of LR rows, then ends[16][32], starts[16][32], misc[2][32]
    constexpr int n = 16 * M, LR = n + 8;
    real_t *tF = lds, *tA = tF + LR, *tPF = tA + LR, *tHB = tPF + LR, *tQB = tHB + LR, *tSA = tQB + LR,
           *tSC = tSA + LR, *tST = tSC + LR;
    real_t *ends = tST + LR, *starts = ends + 16 * 32, *misc = starts + 16 * 32;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xl = lane & 31, c = 2 * wv + (lane >> 5);
    const int p = blockIdx.x * 32 + xl;  // np is a multiple of 32 (launcher)
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int s = c * M + 1;
    // addresses as (wave-uniform row pointer) + (one 32-bit per-lane element offset): with 40 per-lane
    // 64-bit addresses in flight the kernel spills (a block has < 2^31 elements)
    const unsigned off = (unsigned)(base + (long)(s - 1) * rs) * (unsigned)X3D_RB;  // bytes; a block is < 4 GiB

    // ---- P1: load the chunk + 4 + 4 halo rows (periodic image), chunk-local forward elimination in place.
    // x[q] is overwritten by the eliminated value, so the 4 original rows behind the current one are kept
    // in p0..p3; the rows ahead are still original in x[] (the last 4 come from the right halo, loaded late)
    real_t x[M], hr[4];
#pragma unroll
    for (int q = 0; q < M; q++) x[q] = ldg(u + (long)q * rs, off);
    real_t p0 = ldg(u, (unsigned)(base + (long)((s - 5 + n) & (n - 1)) * rs) * (unsigned)X3D_RB),
           p1 = ldg(u, (unsigned)(base + (long)((s - 4 + n) & (n - 1)) * rs) * (unsigned)X3D_RB),
           p2 = ldg(u, (unsigned)(base + (long)((s - 3 + n) & (n - 1)) * rs) * (unsigned)X3D_RB),
           p3 = ldg(u, (unsigned)(base + (long)((s - 2 + n) & (n - 1)) * rs) * (unsigned)X3D_RB);
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
    const real_t c0 = cf.c[0], c1 = cf.c[1], c2 = cf.c[2], c3 = cf.c[3], c4 = cf.c[4], c5 = cf.c[5], c6 = cf.c[6],
                 c7 = cf.c[7], c8 = cf.c[8];  // kernel arguments: SGPRs
    real_t prev = 0.0;
#pragma unroll
    for (int q = 0; q < M; q++) {
        const int j = s + q;
        if (q == M - 12) {
#pragma unroll
            for (int m = 0; m < 4; m++) hr[m] = ldg(u, (unsigned)(base + (long)((s + M - 1 + m) & (n - 1)) * rs) * (unsigned)X3D_RB);
        }
#define AHEAD(d) ((q + (d) < M) ? x[(q + (d)) % M] : hr[(q + (d) - M) & 3])
        const real_t cur = x[q];
        // NARROW: compact6 / classic stencils reach 2 rows only: skip the zero taps (adding 0 * x is exact)
        const real_t acc = NARROW ? c2 * p2 + c3 * p3 + c4 * cur + c5 * AHEAD(1) + c6 * AHEAD(2)
                                  : c0 * p0 + c1 * p1 + c2 * p2 + c3 * p3 + c4 * cur + c5 * AHEAD(1) + c6 * AHEAD(2) +
                                        c7 * AHEAD(3) + c8 * AHEAD(4);
#undef AHEAD
        const real_t e = tF[j] * (acc - tA[j] * prev);
        prev = e;
        x[q] = e;
        p0 = p1; p1 = p2; p2 = p3; p3 = cur;
        if (q & 1) __builtin_amdgcn_sched_barrier(0);  // bound the live range of the LDS table reads
    }
    ends[c * 32 + xl] = prev;
    __syncthreads();

    // ---- P2: forward carry (serial over the chunks before this one), chunk-local back-substitution
    {
        real_t carry = 0.0;
        for (int cc = 0; cc < c; cc++) carry = ends[cc * 32 + xl] + tPF[(cc + 1) * M] * carry;
        real_t nxt = 0.0;
#pragma unroll
        for (int q = M - 1; q >= 0; q--) {
            const int j = s + q;
            const real_t e = x[q] + tPF[j] * carry;
            x[q] = e + tHB[j] * nxt;
            nxt = x[q];
            if ((q & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        starts[c * 32 + xl] = x[0];
    }
    __syncthreads();

    // ---- P3: backward carry (applied on the fly in P4); publish du_1 and X_n
    real_t carry = 0.0;
    for (int cc = 15; cc > c; cc--) carry = starts[cc * 32 + xl] + tQB[cc * M + 1] * carry;
    if (c == 15) misc[32 + xl] = x[M - 1];  // carry = 0 there
    if (c == 0) misc[xl] = t.last_r * ((x[0] + tQB[1] * carry) - t.bw1 * (x[1] + tQB[2] * carry));  // :161-166
    __syncthreads();

    // ---- P4: reduced 2x2 systems with the periodic self-exchange, substitution, store
    const real_t du1 = misc[xl], xn = misc[32 + xl];
    const real_t du_s = t.rs_s * (du1 - t.sa1 * xn);
    const real_t du_e = t.rs_e * (xn - t.scn * du1);
#pragma unroll
    for (int q0 = 0; q0 < M; q0 += 2) {
        real_t old[2];
        if (ACC) {
#pragma unroll
            for (int k = 0; k < 2; k++) old[k] = ldg(du + (long)(q0 + k) * rs, off);
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int q = q0 + k, j = s + q;
            const real_t X = x[q] + tQB[j] * carry;
            real_t r = (X - tSA[j] * du_s - tSC[j] * du_e) * tST[j];  // :215-222
            if (q == 0) r = (c == 0) ? du_s * tST[j] : r;        // row 1, :209-213
            if (q == M - 1) r = (c == 15) ? du_e * tST[j] : r;   // row n, :224-228
            stg(du + (long)q * rs, off, ACC ? old[k] + scale * r : r);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int M>
static int launch_onchip2(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, const PencilGeom &g,
                          int dir, int acc, real_t scale)
{
    const size_t lds = sizeof(real_t) * ((size_t)K1E_TAB * (16 * M + 8) + 2 * 16 * 32 + 64);
    {
        const void *ks[4] = {(const void *)k_tds_onchip2<false, M, false>, (const void *)k_tds_onchip2<true, M, false>,
                             (const void *)k_tds_onchip2<false, M, true>, (const void *)k_tds_onchip2<true, M, true>};
        for (const void *k : ks) X3D_LDS_OPTIN(b, k);
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

int x3d_onchip2_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir, int acc, real_t scale,
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
