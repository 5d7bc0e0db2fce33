// x-direction operators for the Cartesian-pitched block (kernel family K3).
//
// An x pencil is contiguous in memory, so "one pencil per lane" would put the
// 64 lanes of a wave 4 KB apart.  Instead a wave owns 64 consecutive rows
// (= 64 pencils = one contiguous [64][nxp] slab) and streams it in tiles of
// 16 columns: each tile is fetched with 16-byte loads, 128 B contiguous per 8
// lanes, transposed through a wave-private LDS tile (pitch 65 doubles ->
// conflict-free both ways), and consumed by the lane that owns the row.  The
// arithmetic is the same two-sweep DistD2 formulation as tds.hip; the
// forward-eliminated intermediates live in wave-transposed scratch
// ([wave][row j][lane], coalesced as is), the final result goes back through
// an LDS tile so that global stores are coalesced too.
//
// Reference arithmetic: src/backend/omp/kernels/distributed.f90:11-337 (see tds.hip).
#include "common.h"

int x3d_generic_tds_local(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir, int acc,
                          real_t scale);
int x3d_generic_transeq_local(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv,
                              real_t nu, const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3,
                              int acc);
int npmax_of(const x3d_backend *b);

#ifndef X3D_XTW
#define X3D_XTW 16   // tile width (columns): 128 B contiguous per row per fetch (32 costs too many VGPRs)
#endif
#define TW X3D_XTW
#define TP 65        // LDS pitch in doubles
#define TILE (TW * TP)
#define TLD (TW / 2) // 16-byte loads per lane per tile: 64 rows * TW cols / (64 lanes * 2)
#define LPR (TW / 2) // lanes per row in a fetch
#define RPF (64 / LPR) // rows per fetch instruction

__device__ __forceinline__ real_t dot9x(const real_t *__restrict__ c, const real_t (&w)[9])
{
    return c[0] * w[0] + c[1] * w[1] + c[2] * w[2] + c[3] * w[3] + c[4] * w[4] + c[5] * w[5] + c[6] * w[6] +
           c[7] * w[7] + c[8] * w[8];
}

__device__ __forceinline__ const real_t *stencil_row_x(const real_t *__restrict__ Cs, int j, int nr)
{
    if (j <= 4) return Cs + (j - 1) * 9;
    if (j > nr - 4) return Cs + 36 + (j - (nr - 4) - 1) * 9;
    return Cs + 72;
}

// cooperative tile fetch: TLD x (16 B per lane); LPR lanes cover one row
// segment of TW columns (TW*8 B contiguous), RPF rows per instruction
struct TileRegs { real2_t v[TLD]; };

__device__ __forceinline__ void tile_load(TileRegs &r, const real_t *__restrict__ slab, long pitch, int col0,
                                          int rows_valid, int lane)
{
    const int cp = (lane % LPR) * 2;
#pragma unroll
    for (int i = 0; i < TLD; i++) {
        int row = (lane / LPR) + RPF * i;
        row = row < rows_valid ? row : rows_valid - 1;  // partial wave: replicate a valid row
        r.v[i] = *reinterpret_cast<const real2_t *>(slab + (long)row * pitch + col0 + cp);
    }
}

__device__ __forceinline__ void tile_to_lds(const TileRegs &r, real_t *__restrict__ lds, int lane)
{
    const int cp = (lane % LPR) * 2;
#pragma unroll
    for (int i = 0; i < TLD; i++) {
        const int row = (lane / LPR) + RPF * i;
        lds[cp * TP + row] = r.v[i].x;
        lds[(cp + 1) * TP + row] = r.v[i].y;
    }
}

template <bool ACC>
__device__ __forceinline__ void tile_store(const real_t *__restrict__ lds, real_t *__restrict__ slab, long pitch,
                                           int col0, int rows_valid, int lane, real_t scale)
{
    const int cp = (lane % LPR) * 2;
#pragma unroll
    for (int i = 0; i < TLD; i++) {
        const int row = (lane / LPR) + RPF * i;
        if (row < rows_valid) {
            real2_t v;
            v.x = lds[cp * TP + row];
            v.y = lds[(cp + 1) * TP + row];
            real2_t *dst = reinterpret_cast<real2_t *>(slab + (long)row * pitch + col0 + cp);
            if (ACC) {
                const real2_t o = *dst;
                v.x = o.x + scale * v.x;
                v.y = o.y + scale * v.y;
            }
            *dst = v;
        }
    }
}

// ---------------------------------------------------------------- tds, forward
// d (wave-transposed scratch): d[(wave*n + (j-1))*64 + lane]
__global__ void __launch_bounds__(64) k_xtds_fwd(real_t *__restrict__ d, real_t *__restrict__ send_s,
                                                 real_t *__restrict__ send_e, const real_t *__restrict__ u,
                                                 TdsTab t, int np, long pitch, int n_wrap)
{
    __shared__ real_t lds[TILE];
    const int lane = threadIdx.x, wave = blockIdx.x;
    const int p = wave * 64 + lane;
    int rows_valid = np - wave * 64;
    rows_valid = rows_valid > 64 ? 64 : rows_valid;
    const real_t *__restrict__ slab = u + (long)wave * 64 * pitch;
    const int n = t.n_tds, nr = t.n_rhs;
    const int my = lane < rows_valid ? lane : rows_valid - 1;
    const real_t *__restrict__ mine = slab + (long)my * pitch;
    // periodic images (sendrecv_fields nproc==1): u_s(r) = u(n_wrap-4+r), u_e(r) = u(r)
    real_t w[9];
#pragma unroll
    for (int m = 0; m < 5; m++) w[m] = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m++) w[5 + m] = mine[n_wrap - 4 + m];
    real_t he[4];
#pragma unroll
    for (int m = 0; m < 4; m++) he[m] = mine[m];
    real_t dprev = 0.0, S = 0.0, d1 = 0.0, dn = 0.0;
    real_t *__restrict__ dw = d + (long)wave * n * 64 + lane;

    real_t cb[9];
#pragma unroll
    for (int m = 0; m < 9; m++) cb[m] = t.Cs[72 + m];
    const int ntile = (nr + TW - 1) / TW;
    TileRegs regs;
    tile_load(regs, slab, pitch, 0, rows_valid, lane);
    for (int tI = 0; tI <= ntile; tI++) {
        __syncthreads();
        if (tI < ntile) tile_to_lds(regs, lds, lane);
        __syncthreads();
        if (tI + 1 < ntile) tile_load(regs, slab, pitch, (tI + 1) * TW, rows_valid, lane);
#pragma unroll 4
        for (int c = 0; c < TW; c++) {
            const int e = tI * TW + c + 1;  // element fed into the window
            if (e > nr + 4) break;
            const real_t ve = e <= nr ? lds[c * TP + lane] : he[e - nr - 1];
#pragma unroll
            for (int m = 0; m < 8; m++) w[m] = w[m + 1];
            w[8] = ve;
            const int j = e - 4;
            if (j >= 1 && j <= nr) {
                const real_t acc = (j > 4 && j <= nr - 4) ? dot9x(cb, w) : dot9x(stencil_row_x(t.Cs, j, nr), w);
                const real_t dj = T_F(t, j) * (acc - T_A(t, j) * dprev);
                if (j <= n) {
                    dw[(long)(j - 1) * 64] = dj;
                    S += T_W(t, j) * dj;
                    if (j == 1) d1 = dj;
                    if (j == n) dn = dj;
                }
                dprev = dj;
            }
        }
    }
    if (lane < rows_valid) {
        send_e[p] = dn;
        send_s[p] = t.last_r * (d1 - t.bw1 * S);
    }
}

// ---------------------------------------------------------------- tds, backward + subs
template <bool ACC>
__global__ void __launch_bounds__(64) k_xtds_bwd(real_t *__restrict__ du, const real_t *__restrict__ d,
                                                 const real_t *__restrict__ own_s,
                                                 const real_t *__restrict__ recv_s,
                                                 const real_t *__restrict__ recv_e, TdsTab t, int np, long pitch,
                                                 real_t scale)
{
    __shared__ real_t lds[TILE];
    const int lane = threadIdx.x, wave = blockIdx.x;
    int rows_valid = np - wave * 64;
    rows_valid = rows_valid > 64 ? 64 : rows_valid;
    const int p = wave * 64 + (lane < rows_valid ? lane : rows_valid - 1);
    real_t *__restrict__ slab = du + (long)wave * 64 * pitch;
    const int n = t.n_tds;
    const real_t *__restrict__ dw = d + (long)wave * n * 64 + lane;
    const real_t dn = dw[(long)(n - 1) * 64];
    const real_t du_s = t.rs_s * (own_s[p] - t.sa1 * recv_s[p]);
    const real_t du_e = t.rs_e * (dn - t.scn * recv_e[p]);
    real_t nxt = 0.0;
    const int ntile = (n + TW - 1) / TW;
    for (int tI = ntile - 1; tI >= 0; tI--) {
        real_t dv[TW];
#pragma unroll
        for (int c = 0; c < TW; c++) {  // unconditional, clamped: lets the loads issue back to back
            int j = tI * TW + c + 1;
            j = j <= n ? j : n;
            dv[c] = dw[(long)(j - 1) * 64];
        }
        __syncthreads();
#pragma unroll
        for (int c = TW - 1; c >= 0; c--) {
            const int j = tI * TW + c + 1;
            real_t out = 0.0;
            if (j <= n) {
                const real_t cur = (j >= n - 1) ? dv[c] : dv[c] - T_BW(t, j) * nxt;
                out = (cur - T_SA(t, j) * du_s - T_SC(t, j) * du_e) * T_ST(t, j);
                out = (j == n) ? du_e * T_ST(t, j) : out;
                out = (j == 1) ? du_s * T_ST(t, j) : out;
                nxt = cur;
            }
            lds[c * TP + lane] = out;
        }
        __syncthreads();
        tile_store<ACC>(lds, slab, pitch, tI * TW, rows_valid, lane, scale);
    }
}

// ---------------------------------------------------------------- transeq, forward
template <bool SAME>
__global__ void __launch_bounds__(64)
    k_xtranseq_fwd(real_t *__restrict__ d1a, real_t *__restrict__ d2a, real_t *__restrict__ d3a,
                   real_t *__restrict__ send_s, real_t *__restrict__ send_e, const real_t *__restrict__ u,
                   const real_t *__restrict__ cv, TdsTab t1, TdsTab t2, TdsTab t3, int np, long pitch,
                   int npmax)
{
    __shared__ real_t lu[TILE], lc[SAME ? 1 : TILE];
    const int lane = threadIdx.x, wave = blockIdx.x;
    const int p = wave * 64 + lane;
    int rows_valid = np - wave * 64;
    rows_valid = rows_valid > 64 ? 64 : rows_valid;
    const real_t *__restrict__ su = u + (long)wave * 64 * pitch;
    const real_t *__restrict__ sc = cv + (long)wave * 64 * pitch;
    const int n = t1.n_tds;
    const int my = lane < rows_valid ? lane : rows_valid - 1;
    real_t wu[9], wp[9], heu[4], hep[4];
#pragma unroll
    for (int m = 0; m < 5; m++) { wu[m] = 0.0; wp[m] = 0.0; }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const real_t a = su[(long)my * pitch + n - 4 + m];
        const real_t c = SAME ? a : sc[(long)my * pitch + n - 4 + m];
        wu[5 + m] = a; wp[5 + m] = a * c;
        const real_t a2 = su[(long)my * pitch + m];
        const real_t c2 = SAME ? a2 : sc[(long)my * pitch + m];
        heu[m] = a2; hep[m] = a2 * c2;
    }
    real_t p1 = 0, p2 = 0, p3 = 0, S1 = 0, S2 = 0, S3 = 0, f1 = 0, f2 = 0, f3 = 0, l1 = 0, l2 = 0, l3 = 0;
    const long wo = (long)wave * n * 64 + lane;
    real_t b1[9], b2[9], b3[9];
#pragma unroll
    for (int m = 0; m < 9; m++) { b1[m] = t1.Cs[72 + m]; b2[m] = t2.Cs[72 + m]; b3[m] = t3.Cs[72 + m]; }
    const int ntile = (n + TW - 1) / TW;
    TileRegs ru, rc;
    tile_load(ru, su, pitch, 0, rows_valid, lane);
    if (!SAME) tile_load(rc, sc, pitch, 0, rows_valid, lane);
    for (int tI = 0; tI <= ntile; tI++) {
        __syncthreads();
        if (tI < ntile) {
            tile_to_lds(ru, lu, lane);
            if (!SAME) tile_to_lds(rc, lc, lane);
        }
        __syncthreads();
        if (tI + 1 < ntile) {
            tile_load(ru, su, pitch, (tI + 1) * TW, rows_valid, lane);
            if (!SAME) tile_load(rc, sc, pitch, (tI + 1) * TW, rows_valid, lane);
        }
#pragma unroll 2
        for (int c = 0; c < TW; c++) {
            const int e = tI * TW + c + 1;
            if (e > n + 4) break;
            real_t ve, vp;
            if (e <= n) {
                ve = lu[c * TP + lane];
                vp = ve * (SAME ? ve : lc[c * TP + lane]);
            } else {
                ve = heu[e - n - 1];
                vp = hep[e - n - 1];
            }
#pragma unroll
            for (int m = 0; m < 8; m++) { wu[m] = wu[m + 1]; wp[m] = wp[m + 1]; }
            wu[8] = ve; wp[8] = vp;
            const int j = e - 4;
            if (j >= 1) {
                const bool bulk = j > 4 && j <= n - 4;
                const real_t a1 = bulk ? dot9x(b1, wu) : dot9x(stencil_row_x(t1.Cs, j, n), wu);
                const real_t a3 = bulk ? dot9x(b3, wu) : dot9x(stencil_row_x(t3.Cs, j, n), wu);
                const real_t a2 = bulk ? dot9x(b2, wp) : dot9x(stencil_row_x(t2.Cs, j, n), wp);
                const real_t e1 = T_F(t1, j) * (a1 - T_A(t1, j) * p1);
                const real_t e2 = T_F(t2, j) * (a2 - T_A(t2, j) * p2);
                const real_t e3 = T_F(t3, j) * (a3 - T_A(t3, j) * p3);
                const long o = wo + (long)(j - 1) * 64;
                d1a[o] = e1; d2a[o] = e2; d3a[o] = e3;
                S1 += T_W(t1, j) * e1; S2 += T_W(t2, j) * e2; S3 += T_W(t3, j) * e3;
                if (j == 1) { f1 = e1; f2 = e2; f3 = e3; }
                if (j == n) { l1 = e1; l2 = e2; l3 = e3; }
                p1 = e1; p2 = e2; p3 = e3;
            }
        }
    }
    if (lane < rows_valid) {
        send_e[p] = l1; send_e[npmax + p] = l2; send_e[2 * npmax + p] = l3;
        send_s[p] = t1.last_r * (f1 - t1.bw1 * S1);
        send_s[npmax + p] = t2.last_r * (f2 - t2.bw1 * S2);
        send_s[2 * npmax + p] = t3.last_r * (f3 - t3.bw1 * S3);
    }
}

// ---------------------------------------------------------------- transeq, backward + fused subs
template <bool ACC>
__global__ void __launch_bounds__(64)
    k_xtranseq_bwd(real_t *__restrict__ rhs, const real_t *__restrict__ d1a, const real_t *__restrict__ d2a,
                   const real_t *__restrict__ d3a, const real_t *__restrict__ cv,
                   const real_t *__restrict__ own_s, const real_t *__restrict__ recv_s,
                   const real_t *__restrict__ recv_e, real_t nu, TdsTab t1, TdsTab t2, TdsTab t3, int np,
                   long pitch, int npmax)
{
    __shared__ real_t lo[TILE], lc[TILE];
    const int lane = threadIdx.x, wave = blockIdx.x;
    int rows_valid = np - wave * 64;
    rows_valid = rows_valid > 64 ? 64 : rows_valid;
    const int p = wave * 64 + (lane < rows_valid ? lane : rows_valid - 1);
    real_t *__restrict__ so = rhs + (long)wave * 64 * pitch;
    const real_t *__restrict__ sc = cv + (long)wave * 64 * pitch;
    const int n = t1.n_tds;
    const long wo = (long)wave * n * 64 + lane;
    const long on = wo + (long)(n - 1) * 64;
    const real_t du_s = t1.rs_s * (own_s[p] - t1.sa1 * recv_s[p]);
    const real_t dud_s = t2.rs_s * (own_s[npmax + p] - t2.sa1 * recv_s[npmax + p]);
    const real_t d2u_s = t3.rs_s * (own_s[2 * npmax + p] - t3.sa1 * recv_s[2 * npmax + p]);
    real_t n1 = d1a[on], n2 = d2a[on], n3 = d3a[on];
    const real_t du_e = t1.rs_e * (n1 - t1.scn * recv_e[p]);
    const real_t dud_e = t2.rs_e * (n2 - t2.scn * recv_e[npmax + p]);
    const real_t d2u_e = t3.rs_e * (n3 - t3.scn * recv_e[2 * npmax + p]);
    const int ntile = (n + TW - 1) / TW;
    TileRegs rc;
    tile_load(rc, sc, pitch, (ntile - 1) * TW, rows_valid, lane);
    for (int tI = ntile - 1; tI >= 0; tI--) {
        __syncthreads();
        tile_to_lds(rc, lc, lane);
        __syncthreads();
        if (tI > 0) tile_load(rc, sc, pitch, (tI - 1) * TW, rows_valid, lane);
#pragma unroll
        for (int h = 1; h >= 0; h--) {  // two half-tiles: their d loads are issued as one batch
            constexpr int HB_ = TW / 2;
            real_t a1[HB_], a2[HB_], a3[HB_];
#pragma unroll
            for (int k = 0; k < HB_; k++) {
                int j = tI * TW + h * HB_ + k + 1;
                j = j <= n ? j : n;
                const long o = wo + (long)(j - 1) * 64;
                a1[k] = d1a[o]; a2[k] = d2a[o]; a3[k] = d3a[o];
            }
#pragma unroll
            for (int k = HB_ - 1; k >= 0; k--) {
                const int c = h * HB_ + k;
                const int j = tI * TW + c + 1;
                real_t out = 0.0;
                if (j <= n) {
                    const real_t v = lc[c * TP + lane];
                    const bool keep = j >= n - 1;  // rows n, n-1: forward values (distributed.f90:154)
                    const real_t c1 = keep ? a1[k] : a1[k] - T_BW(t1, j) * n1;
                    const real_t c2 = keep ? a2[k] : a2[k] - T_BW(t2, j) * n2;
                    const real_t c3 = keep ? a3[k] : a3[k] - T_BW(t3, j) * n3;
                    const real_t temp_du = T_ST(t1, j) * (c1 - T_SA(t1, j) * du_s - T_SC(t1, j) * du_e);
                    const real_t temp_dud = T_ST(t2, j) * (c2 - T_SA(t2, j) * dud_s - T_SC(t2, j) * dud_e);
                    const real_t temp_d2u =
                        T_ST(t3, j) * (c3 - T_SA(t3, j) * d2u_s - T_SC(t3, j) * d2u_e) + temp_du * T_STC(t3, j);
                    out = -0.5 * (v * temp_du + temp_dud) + nu * temp_d2u;
                    if (j == n)
                        out = -0.5 * (v * du_e * T_ST(t1, n) + dud_e * T_ST(t2, n)) +
                              nu * (d2u_e * T_ST(t3, n) + du_e * T_ST(t1, n) * T_STC(t3, n));
                    if (j == 1)
                        out = -0.5 * (v * du_s * T_ST(t1, 1) + dud_s * T_ST(t2, 1)) +
                              nu * (d2u_s * T_ST(t3, 1) + du_s * T_ST(t1, 1) * T_STC(t3, 1));
                    n1 = c1; n2 = c2; n3 = c3;
                }
                lo[c * TP + lane] = out;
            }
        }
        __syncthreads();
        tile_store<ACC>(lo, so, pitch, tI * TW, rows_valid, lane, 1.0);
    }
}

// ---------------------------------------------------------------- launchers
static bool xdir_tiled()
{
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("X3D_XDIR_GENERIC");
        mode = (e && e[0] == '1') ? 0 : 1;
    }
    return mode == 1;
}

int x3d_xscan_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int acc, real_t scale, bool *done);
int x3d_xwide_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int acc, real_t scale, bool *done,
                  real_t *psum = nullptr, int ny_sum = 0, int *nsum = nullptr);  // xwide.hip
int x3d_xscan_transeq(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                      const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, bool *done);
static bool use_xscan()
{
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("X3D_NO_XSCAN");
        mode = (e && e[0] == '1') ? 0 : 1;
    }
    return mode == 1;
}

int x3d_xdir_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int acc, real_t scale)
{
    if (use_xscan() && xdir_tiled()) {
        bool done = false;
        if (int rc = x3d_xscan_tds(b, du, u, t, acc, scale, &done)) return rc;
        if (done) return 0;
        if (int rc = x3d_xwide_tds(b, du, u, t, acc, scale, &done)) return rc;  // 1024-row pencils (xwide.hip)
        if (done) return 0;
    }
    if (!xdir_tiled()) return x3d_generic_tds_local(b, du, u, t, X3D_DIR_X, acc, scale);
    const int np = b->ny * b->nz, nw = (np + 63) / 64;
    {
        ProfScope ps(b, X3D_K_TDS_FWD, X3D_DIR_X);
        hipLaunchKernelGGL(k_xtds_fwd, dim3(nw), dim3(64), 0, b->stream, b->scratch[2], b->send_s, b->send_e, u,
                           t->tab, np, (long)b->nxp, t->n_tds);
    }
    {
        ProfScope ps(b, X3D_K_TDS_BWD, X3D_DIR_X);
        if (acc)
            hipLaunchKernelGGL(k_xtds_bwd<true>, dim3(nw), dim3(64), 0, b->stream, du, b->scratch[2], b->send_s,
                               b->send_e, b->send_s, t->tab, np, (long)b->nxp, scale);
        else
            hipLaunchKernelGGL(k_xtds_bwd<false>, dim3(nw), dim3(64), 0, b->stream, du, b->scratch[2], b->send_s,
                               b->send_e, b->send_s, t->tab, np, (long)b->nxp, 1.0);
    }
    X3D_HIP(hipGetLastError());
    return 0;
}

int x3d_xdir_transeq(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                     const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc)
{
    if (use_xscan() && xdir_tiled()) {
        bool done = false;
        if (int rc = x3d_xscan_transeq(b, rhs, u, conv, nu, t1, t2, t3, acc, &done)) return rc;
        if (done) return 0;
    }
    if (!xdir_tiled()) return x3d_generic_transeq_local(b, X3D_DIR_X, rhs, u, conv, nu, t1, t2, t3, acc);
    const int np = b->ny * b->nz, nw = (np + 63) / 64, npm = npmax_of(b);
    {
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, X3D_DIR_X);
        if (u == conv)
            hipLaunchKernelGGL(k_xtranseq_fwd<true>, dim3(nw), dim3(64), 0, b->stream, b->scratch[2],
                               b->scratch[0], b->scratch[1], b->send_s, b->send_e, u, conv, t1->tab, t2->tab,
                               t3->tab, np, (long)b->nxp, npm);
        else
            hipLaunchKernelGGL(k_xtranseq_fwd<false>, dim3(nw), dim3(64), 0, b->stream, b->scratch[2],
                               b->scratch[0], b->scratch[1], b->send_s, b->send_e, u, conv, t1->tab, t2->tab,
                               t3->tab, np, (long)b->nxp, npm);
    }
    {
        ProfScope ps(b, X3D_K_TRANSEQ_BWD, X3D_DIR_X);
        if (acc)
            hipLaunchKernelGGL(k_xtranseq_bwd<true>, dim3(nw), dim3(64), 0, b->stream, rhs, b->scratch[2],
                               b->scratch[0], b->scratch[1], conv, b->send_s, b->send_e, b->send_s, nu, t1->tab,
                               t2->tab, t3->tab, np, (long)b->nxp, npm);
        else
            hipLaunchKernelGGL(k_xtranseq_bwd<false>, dim3(nw), dim3(64), 0, b->stream, rhs, b->scratch[2],
                               b->scratch[0], b->scratch[1], conv, b->send_s, b->send_e, b->send_s, nu, t1->tab,
                               t2->tab, t3->tab, np, (long)b->nxp, npm);
    }
    X3D_HIP(hipGetLastError());
    return 0;
}
