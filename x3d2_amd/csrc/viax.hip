// K3t: y/z transport-equation components through the wave-per-pencil scan kernel (xscan.hip, K3s) on
// transposed copies -- the route for periodic 256 / 512-row pencils that the tile kernel K3y (xscan.hip,
// k_ytile_transeq: no copies at all) does not take, e.g. nx not a multiple of 16.
//
// The lane-per-pencil two-sweep pair of tds.hip moves 11 field passes per component (3 intermediate arrays
// written by the forward sweep and read back by the backward sweep).  The scan kernel does a component in one
// pass, but wants its pencils contiguous.  For periodic pencils of 64 Q rows this file therefore
//     T   = transpose(field)          [z][y][x] -> [z][x][y]   (dir y)   or   [y][x][z]   (dir z)
//     tmp = k_xscan_transeq(T, T0)    one pass, pencils 8 Q*64 bytes long
//     rhs (+)= transpose^-1(tmp)
// = 2 + 3 (2) + 3 (2) passes at the streaming rate instead of 11, all three intermediates in the backend's
// scratch blocks that the two-sweep path would have used for its own intermediates.
// Same arithmetic as x3d_xdir_transeq, i.e. exec_dist_transeq_compact (src/backend/omp/exec_dist.f90:79-190)
// + the 2x2 reduced systems of der_univ_subs, and the component permutation of
// src/backend/omp/backend.f90:145-184.
#include "common.h"

int x3d_xscan_transeq_np(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                         const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, int np, long pitch,
                         int dirtag, bool *done);
bool x3d_xscan_fast_ok(const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3);
bool x3d_ytile_applicable(x3d_backend *b, int dir, const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3);
int x3d_ytile_transeq3(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                       const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                       const x3d_tdsops *der2nd_sym, int acc, const TileHalo *halo, int other0, int nother,
                       bool *done);
int x3d_ytile_transeq_lincomb(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                              const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, real_t *y,
                              const real_t *base, int nterm, const real_t *c, real_t *const *x, int ipend, int store,
                              bool *done);
int x3d_ytile_transeq(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                      const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, bool *done);

// dst[c*dC + a*dA + b] (+)= src[c*sC + b*sB + a]: 64 x 64 tiles through LDS, 512-byte rows on both sides
template <bool ACC, bool NT>
__global__ void __launch_bounds__(256)
    k_transpose64(real_t *__restrict__ dst, const real_t *__restrict__ src, int nA, int nB, long dA, long dC, long sB,
                  long sC)
{
    __shared__ real_t tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const int a0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const real_t *__restrict__ sp = src + (long)blockIdx.z * sC;
    real_t *__restrict__ dp = dst + (long)blockIdx.z * dC;
    const bool full = a0 + 64 <= nA && b0 + 64 <= nB;
    if (full) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            {
            const real_t *q = &sp[a0 + tx + (long)(b0 + ty + 4 * r) * sB];
            tile[ty + 4 * r][tx] = NT ? __builtin_nontemporal_load(q) : *q;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            real_t *q = &dp[(long)(a0 + ty + 4 * r) * dA + b0 + tx];
            const real_t v = tile[tx][ty + 4 * r];
            if (ACC) *q += v;
            else *q = v;
        }
    } else {
        for (int r = 0; r < 16; r++) {
            const int bb = b0 + ty + 4 * r, aa = a0 + tx;
            if (aa < nA && bb < nB) tile[ty + 4 * r][tx] = sp[aa + (long)bb * sB];
        }
        __syncthreads();
        for (int r = 0; r < 16; r++) {
            const int aa = a0 + ty + 4 * r, bb = b0 + tx;
            if (aa < nA && bb < nB) {
                real_t *q = &dp[(long)aa * dA + bb];
                const real_t v = tile[tx][ty + 4 * r];
                if (ACC) *q += v;
                else *q = v;
            }
        }
    }
}

static bool use_via_x()
{
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("X3D_NO_VIA_X"), *f = getenv("X3D_NO_XSCAN");
        mode = ((e && e[0] == '1') || (f && f[0] == '1')) ? 0 : 1;
    }
    return mode == 1;
}

static int via_slabs(int nC)
{
    static int n = -1;
    if (n < 0) {
        const char *e = getenv("X3D_VIA_SLABS");
        n = e ? atoi(e) : 1;
        if (n < 1) n = 1;
    }
    return n > nC ? nC : n;
}

struct ViaGeom {
    int nA, nB, nC;        // forward transpose: A = x (contiguous in the block), B = the pencil direction, C = the other
    long f_dA, f_dC, f_sB, f_sC;
    int np;
    long pitch;
};

static ViaGeom via_geom(const x3d_backend *b, int dir)
{
    ViaGeom g;
    const long px = b->nxp, pxy = (long)b->nxp * b->nyp;
    if (dir == X3D_DIR_Y) {  // T[z][x][y]
        g.nA = b->nx; g.nB = b->ny; g.nC = b->nz;
        g.f_sB = px; g.f_sC = pxy;
        g.f_dA = b->ny; g.f_dC = (long)b->nx * b->ny;
        g.pitch = b->ny; g.np = b->nx * b->nz;
    } else {  // T[y][x][z]
        g.nA = b->nx; g.nB = b->nz; g.nC = b->ny;
        g.f_sB = pxy; g.f_sC = px;
        g.f_dA = b->nz; g.f_dC = (long)b->nx * b->nz;
        g.pitch = b->nz; g.np = b->nx * b->ny;
    }
    return g;
}

// (c0, nc): the slab of C planes to move
static int to_pencils(x3d_backend *b, const ViaGeom &g, real_t *T, const real_t *f, int c0, int nc)
{
    dim3 grid((g.nA + 63) / 64, (g.nB + 63) / 64, nc);
    hipLaunchKernelGGL((k_transpose64<false, false>), grid, dim3(256), 0, b->stream, T + c0 * g.f_dC, f + c0 * g.f_sC,
                       g.nA, g.nB, g.f_dA, g.f_dC, g.f_sB, g.f_sC);
    X3D_HIP(hipGetLastError());
    return 0;
}

static int from_pencils(x3d_backend *b, int dir, const ViaGeom &g, real_t *r, const real_t *T, int acc, int c0,
                        int nc)
{
    ProfScope ps(b, X3D_K_TRANSEQ_BWD, dir);
    // inverse: A' = the pencil direction (contiguous in T), B' = x
    dim3 grid((g.nB + 63) / 64, (g.nA + 63) / 64, nc);
    r += c0 * g.f_sC;
    T += c0 * g.f_dC;
    if (acc)
        hipLaunchKernelGGL((k_transpose64<true, true>), grid, dim3(256), 0, b->stream, r, T, g.nB, g.nA, g.f_sB, g.f_sC, g.f_dA,
                           g.f_dC);
    else
        hipLaunchKernelGGL((k_transpose64<false, true>), grid, dim3(256), 0, b->stream, r, T, g.nB, g.nA, g.f_sB, g.f_sC,
                           g.f_dA, g.f_dC);
    X3D_HIP(hipGetLastError());
    return 0;
}

// r[c] (+)= transeq component c of direction dir (y or z); f[0] is the advecting component
int x3d_transeq_via_x(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                      const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                      const x3d_tdsops *der2nd_sym, int acc, bool *done)
{
    *done = false;
    if (!use_via_x() || dir == X3D_DIR_X) return 0;
    if (!x3d_xscan_fast_ok(der1st, der1st_sym, der2nd) || !x3d_xscan_fast_ok(der1st_sym, der1st, der2nd_sym)) return 0;
    const int n = dir == X3D_DIR_Y ? b->ny : b->nz;
    if (der1st->n_tds != n) return 0;
    const ViaGeom g = via_geom(b, dir);
    if (g.nC > 65535 || (size_t)b->nx * b->ny * b->nz > b->nblock) return 0;
    {
        // K3y (xscan.hip): the same kernel fed through an LDS tile, no transposed copies; all three components in
        // one launch when the operators allow
        bool ok = false;
        if (int rc = x3d_ytile_transeq3(b, dir, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, acc, nullptr, 0, -1, &ok))
            return rc;
        if (ok) { *done = true; return 0; }
        if (int rc = x3d_ytile_transeq(b, dir, r[0], f[0], f[0], nu, der1st, der1st_sym, der2nd, acc, &ok)) return rc;
        if (ok) {
            for (int c = 1; c < 3; c++) {
                if (int rc = x3d_ytile_transeq(b, dir, r[c], f[c], f[0], nu, der1st_sym, der1st, der2nd_sym, acc, &ok)) return rc;
                X3D_REQUIRE(ok, "x3d_transeq_via_x: tile kernel refused component %d", c);
            }
            *done = true;
            return 0;
        }
    }
    real_t *T0 = b->scratch[0], *T1 = b->scratch[1], *tmp = b->scratch[2];
    // slabs of C planes: the transposed copies and the scan kernel's output of one slab are produced and
    // consumed back to back, so that most of their re-reads hit the 256 MB Infinity Cache
    const int nslab = via_slabs(g.nC);
    for (int sl = 0; sl < nslab; sl++) {
        const int c0 = (int)((long)g.nC * sl / nslab), nc = (int)((long)g.nC * (sl + 1) / nslab) - c0;
        const long off = c0 * g.f_dC;  // pencils c0 * nA ... of the transposed copies
        const int np = nc * g.nA;
        for (int c = 0; c < 3; c++) {
            bool ok = false;
            real_t *T = c == 0 ? T0 : T1;
            // profiler: transpose + scan kernel = the component's "forward" launch, the accumulating
            // inverse transpose its "backward" launch (bench.py prices a component as fwd + bwd)
            {
                ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
                if (int rc = to_pencils(b, g, T, f[c], c0, nc)) return rc;
                if (int rc = x3d_xscan_transeq_np(b, tmp + off, T + off, T0 + off, nu, c == 0 ? der1st : der1st_sym,
                                                  c == 0 ? der1st_sym : der1st, c == 0 ? der2nd : der2nd_sym, 0, np,
                                                  g.pitch, -1, &ok))
                    return rc;
            }
            if (!ok) {
                X3D_REQUIRE(sl == 0 && c == 0, "x3d_transeq_via_x: scan kernel refused component %d", c);
                return 0;  // nothing written to r yet
            }
            if (int rc = from_pencils(b, dir, g, r[c], tmp, acc, c0, nc)) return rc;
        }
    }
    *done = true;
    return 0;
}

// ---------------------------------------------------------------- deferred accumulation (fused RK stage)
// The last direction's accumulating transposes can be folded into the time integrator's linear
// combination (time_integrator.py, runge_kutta_fused): x3d_transeq_defer leaves the scan kernel's outputs in
// pend[c] (pencil layout), x3d_lincomb_pending then computes, per point,
//     d = x[ipend] + transpose^-1(pend);  [x[ipend] = d;]  y = base + sum_k c[k] * (k == ipend ? d : x[k])
// in the summation order of k_lincomb (backend.hip): bit-identical to from_pencils followed by x3d_lincomb, with
// 4 field passes fewer per variable and RK3 step.
struct LinPend {
    const real_t *x[5];
    real_t c[5];
    int n, ipend;
};

__device__ __forceinline__ real2_t ldnt2(const real_t *p)
{
    return make_real2(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1));
}

template <bool STORE>
__global__ void __launch_bounds__(256)
    k_transpose_lincomb(real_t *y, const real_t *base, real_t *xp, const real_t *__restrict__ src, int nA, int nB,
                        long dA, long dC, long sB, long sC, LinPend a)
{
    __shared__ real_t tile[64][65];
    const int a0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const real_t *__restrict__ sp = src + (long)blockIdx.z * sC;
    const long cof = (long)blockIdx.z * dC;
    const bool full = a0 + 64 <= nA && b0 + 64 <= nB && (dA & 1) == 0 && (dC & 1) == 0;
    if (full) {
        {
            const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
            for (int r = 0; r < 16; r++)
                tile[ty + 4 * r][tx] = __builtin_nontemporal_load(&sp[a0 + tx + (long)(b0 + ty + 4 * r) * sB]);
        }
        __syncthreads();
        // 16-byte accesses along the contiguous index of the block, every term streamed once
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll 2
        for (int r = 0; r < 8; r++) {
            const int al = ty + 8 * r;
            const long i = cof + (long)(a0 + al) * dA + b0 + 2 * tx;
            real2_t d = ldnt2(xp + i);
            d.x += tile[2 * tx][al];
            d.y += tile[2 * tx + 1][al];
            if (STORE) { xp[i] = d.x; xp[i + 1] = d.y; }
            real2_t v = ldnt2(base + i);
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (k < a.n) {
                    real2_t t = d;
                    if (k != a.ipend) t = ldnt2(a.x[k] + i);
                    v.x = a.c[k] * t.x + v.x;
                    v.y = a.c[k] * t.y + v.y;
                }
            __builtin_nontemporal_store(v.x, y + i);
            __builtin_nontemporal_store(v.y, y + i + 1);
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = 0; r < 16; r++) {
        const int bb = b0 + ty + 4 * r, aa = a0 + tx;
        if (aa < nA && bb < nB) tile[ty + 4 * r][tx] = sp[aa + (long)bb * sB];
    }
    __syncthreads();
    for (int r = 0; r < 16; r++) {
        const int aa = a0 + ty + 4 * r, bb = b0 + tx;
        if (aa < nA && bb < nB) {
            const long i = cof + (long)aa * dA + bb;
            const real_t d = xp[i] + tile[tx][ty + 4 * r];
            if (STORE) xp[i] = d;
            real_t v = base[i];
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (k < a.n) v = a.c[k] * (k == a.ipend ? d : a.x[k][i]) + v;
            y[i] = v;
        }
    }
}

static int defer_perm(int dir, int c)  // component c of direction dir -> index into (du, dv, dw)
{
    static const int py[3] = {1, 0, 2}, pz[3] = {2, 0, 1};
    return dir == X3D_DIR_Y ? py[c] : pz[c];
}

extern "C" int x3d_transeq_defer(x3d_backend *b, int dir, real_t *pu, real_t *pv, real_t *pw, const real_t *u,
                                 const real_t *v, const real_t *w, real_t nu, const x3d_tdsops *der1st,
                                 const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                                 const x3d_tdsops *der2nd_sym, int *deferred)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && pu && pv && pw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym && deferred,
                "x3d_transeq_defer: null argument");
    *deferred = 0;
    if (!use_via_x() || (dir != X3D_DIR_Y && dir != X3D_DIR_Z)) return 0;
    if (x3d_ytile_applicable(b, dir, der1st, der1st_sym, der2nd) &&
        x3d_ytile_applicable(b, dir, der1st_sym, der1st, der2nd_sym))
        return 0;  // the tile kernel + plain lincomb is faster than transposed copies + the fused RK stage
    if (!x3d_xscan_fast_ok(der1st, der1st_sym, der2nd) || !x3d_xscan_fast_ok(der1st_sym, der1st, der2nd_sym)) return 0;
    const int n = dir == X3D_DIR_Y ? b->ny : b->nz;
    if (der1st->n_tds != n) return 0;
    const ViaGeom g = via_geom(b, dir);
    if (g.nC > 65535 || (size_t)b->nx * b->ny * b->nz > b->nblock) return 0;
    real_t *pend[3] = {pu, pv, pw};
    const real_t *fld[3] = {u, v, w};
    real_t *T0 = b->scratch[0], *T1 = b->scratch[1];
    for (int c = 0; c < 3; c++) {
        const int m = defer_perm(dir, c);
        real_t *T = c == 0 ? T0 : T1;
        bool ok = false;
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
        if (int rc = to_pencils(b, g, T, fld[m], 0, g.nC)) return rc;
        if (int rc = x3d_xscan_transeq_np(b, pend[m], T, T0, nu, c == 0 ? der1st : der1st_sym,
                                          c == 0 ? der1st_sym : der1st, c == 0 ? der2nd : der2nd_sym, 0, g.np, g.pitch,
                                          -1, &ok))
            return rc;
        if (!ok) {
            X3D_REQUIRE(c == 0, "x3d_transeq_defer: scan kernel refused component %d", c);
            return 0;
        }
    }
    *deferred = 1;
    return 0;
}

// r += transpose^-1(pend): the plain completion of a deferred component
extern "C" int x3d_pending_flush(x3d_backend *b, int dir, real_t *r, const real_t *pend)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && r && pend, "x3d_pending_flush: null argument");
    X3D_REQUIRE(dir == X3D_DIR_Y || dir == X3D_DIR_Z, "x3d_pending_flush: dir must be Y or Z");
    return from_pencils(b, dir, via_geom(b, dir), r, pend, 1, 0, via_geom(b, dir).nC);
}

extern "C" int x3d_lincomb_pending(x3d_backend *b, int dir, real_t *y, const real_t *base, int nterm,
                                   const real_t *c, real_t *const *x, int ipend, const real_t *pend, int store)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && y && base && c && x && pend, "x3d_lincomb_pending: null argument");
    X3D_REQUIRE(nterm >= 1 && nterm <= 5 && ipend >= 0 && ipend < nterm, "x3d_lincomb_pending: bad term count / index");
    X3D_REQUIRE(dir == X3D_DIR_Y || dir == X3D_DIR_Z, "x3d_lincomb_pending: dir must be Y or Z");
    X3D_REQUIRE(y != x[ipend] && base != x[ipend], "x3d_lincomb_pending: the pending term aliases y or base");
    const ViaGeom g = via_geom(b, dir);
    LinPend a;
    a.n = nterm;
    a.ipend = ipend;
    for (int k = 0; k < 5; k++) {
        a.x[k] = k < nterm ? x[k] : x[0];
        a.c[k] = k < nterm ? c[k] : 0.0;
    }
    // booked as a "backward" launch of direction 0 (n/a): bench.py adds the RK stage's own algorithmic bytes
    // for these launches (HipBackend.rk_fused_passes)
    ProfScope ps(b, X3D_K_TRANSEQ_BWD, 0);
    dim3 grid((g.nB + 63) / 64, (g.nA + 63) / 64, g.nC);
    if (store)
        hipLaunchKernelGGL(k_transpose_lincomb<true>, grid, dim3(256), 0, b->stream, y, base, x[ipend], pend, g.nB, g.nA,
                           g.f_sB, g.f_sC, g.f_dA, g.f_dC, a);
    else
        hipLaunchKernelGGL(k_transpose_lincomb<false>, grid, dim3(256), 0, b->stream, y, base, x[ipend], pend, g.nB,
                           g.nA, g.f_sB, g.f_sC, g.f_dA, g.f_dC, a);
    X3D_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- RK stage inside the tile kernel
// When the last direction's pencils take the tile kernel (xscan.hip, k_ytile_transeq), the whole component
// is computed inside the stage's linear combination: x[ipend] (= rhs, holding the other directions'
// contributions) + this component -> d; store != 0: x[ipend] = d; y = base + sum_i c[i] * (i == ipend ? d : x[i]).
// kind 0: the advecting component (operators der1st, der1st_sym, der2nd; conv == u), kind 1: the other two
// (der1st_sym, der1st, der2nd_sym), src/backend/omp/backend.f90:246-260.
// Bit-identical to x3d_transeq_species(accumulate = 1) followed by x3d_lincomb.
extern "C" int x3d_transeq_stage_ok(x3d_backend *b, int dir, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                                    const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym)
{
    X3D_RANGE(__func__);
    if (!b || !der1st || !der1st_sym || !der2nd || !der2nd_sym || (dir != X3D_DIR_Y && dir != X3D_DIR_Z)) return 0;
    return use_via_x() && x3d_ytile_applicable(b, dir, der1st, der1st_sym, der2nd) &&
           x3d_ytile_applicable(b, dir, der1st_sym, der1st, der2nd_sym);
}

extern "C" int x3d_transeq_lincomb(x3d_backend *b, int dir, int kind, const real_t *u, const real_t *conv, real_t nu,
                                   const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                                   const x3d_tdsops *der2nd_sym, real_t *y, const real_t *base, int nterm,
                                   const real_t *c, real_t *const *x, int ipend, int store)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && u && conv && der1st && der1st_sym && der2nd && der2nd_sym && y && base && c && x,
                "x3d_transeq_lincomb: null argument");
    X3D_REQUIRE(nterm >= 1 && nterm <= 5 && ipend >= 0 && ipend < nterm, "x3d_transeq_lincomb: bad term count / index");
    X3D_REQUIRE(kind == 0 || kind == 1, "x3d_transeq_lincomb: kind must be 0 or 1");
    X3D_REQUIRE(x[ipend] != u && x[ipend] != conv && y != x[ipend] && base != x[ipend],
                "x3d_transeq_lincomb: the pending term aliases an input");
    X3D_REQUIRE(x3d_transeq_stage_ok(b, dir, der1st, der1st_sym, der2nd, der2nd_sym),
                "x3d_transeq_lincomb: the tile kernel does not take these pencils (see x3d_transeq_stage_ok)");
    bool done = false;
    const x3d_tdsops *t1 = kind == 0 ? der1st : der1st_sym, *t2 = kind == 0 ? der1st_sym : der1st,
                     *t3 = kind == 0 ? der2nd : der2nd_sym;
    if (int rc = x3d_ytile_transeq_lincomb(b, dir, x[ipend], u, conv, nu, t1, t2, t3, y, base, nterm, c, x, ipend, store,
                                           &done))
        return rc;
    X3D_REQUIRE(done, "x3d_transeq_lincomb: tile kernel refused");
    return 0;
}

// ---------------------------------------------------------------- the RK stage of u, v, w inside ONE three-component launch
// transeq_<dir> (dir = y or z, the LAST direction accumulated into du, dv, dw) + the stage's linear combinations of the
// three variables (src/time_integrator.f90:166-231 after src/solver.f90:291-389):
//     d_i = dvar_i + component_i;  [store_i: dvar_i = d_i;]  y_i = base_i + sum_k c_i[k] (k == ipend_i ? d_i : x_i[k])
// variable order u, v, w; c, x: [3][5] row by row; x_i[ipend_i] must be dvar_i.  Bit-identical to x3d_transeq_acc followed by
// x3d_lincomb per variable.  *done = 0: these pencils are not served by the three-in-one tile kernel (nothing was done).
int x3d_ytile_transeq3_epi(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                           const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                           const x3d_tdsops *der2nd_sym, const TileEpi epi[3], bool *done);
extern "C" int x3d_transeq_lincomb3(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u,
                                    const real_t *v, const real_t *w, real_t nu, const x3d_tdsops *der1st,
                                    const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                                    real_t *const y[3], const real_t *const base[3], const int nterm[3], const real_t *c,
                                    real_t *const *x, const int ipend[3], const int store[3], int *done)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && dv && dw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym && y && base && nterm &&
                c && x && ipend && store && done, "x3d_transeq_lincomb3: null argument");
    X3D_REQUIRE(dir == X3D_DIR_Y || dir == X3D_DIR_Z, "x3d_transeq_lincomb3: dir must be y or z");
    *done = 0;
    real_t *dv_[3] = {du, dv, dw};
    const real_t *fv[3] = {u, v, w};
    TileEpi e[3];
    // component order of the direction: the advecting variable first (x3d_transeq_acc)
    const int var_of[3] = {dir == X3D_DIR_Y ? 1 : 2, 0, dir == X3D_DIR_Y ? 2 : 1};
    real_t *r[3];
    const real_t *f[3];
    for (int cpt = 0; cpt < 3; cpt++) {
        const int i = var_of[cpt];
        X3D_REQUIRE(nterm[i] >= 1 && nterm[i] <= 5 && ipend[i] >= 0 && ipend[i] < nterm[i],
                    "x3d_transeq_lincomb3: bad term count / index");
        X3D_REQUIRE(y[i] && base[i] && x[5 * i + ipend[i]] == dv_[i], "x3d_transeq_lincomb3: x[ipend] must be the variable's derivative block");
        // (y_i may BE the variable it updates: a workgroup owns its tile's pencils, reads field i's rows of the tile before
        //  component i's store phase writes them, and keeps the advecting velocity's rows in registers; nothing else may alias)
        for (int k = 0; k < 3; k++)
            X3D_REQUIRE((y[i] != fv[k] || k == i) && y[i] != dv_[k], "x3d_transeq_lincomb3: a result aliases an input of the launch");
        r[cpt] = dv_[i];
        f[cpt] = fv[i];
        e[cpt].y = y[i]; e[cpt].base = base[i]; e[cpt].n = nterm[i]; e[cpt].ipend = ipend[i]; e[cpt].store = store[i];
        for (int k = 0; k < 5; k++) {
            e[cpt].x[k] = k < nterm[i] ? x[5 * i + k] : x[5 * i];
            e[cpt].c[k] = k < nterm[i] ? c[5 * i + k] : 0.0;
        }
    }
    bool ok = false;
    if (int rc = x3d_ytile_transeq3_epi(b, dir, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, e, &ok)) return rc;
    *done = ok ? 1 : 0;
    return 0;
}
