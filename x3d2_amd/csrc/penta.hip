// compact10_penta: the 10th-order first derivative with a PENTADIAGONAL left-hand side.
//
// Reference arithmetic restated for CDNA4 (paths under /root/reference):
//   src/backend/omp/kernels/distributed.f90:339-463  der_penta_full      (RHS, L-solve, U-solve)
//   src/backend/omp/kernels/distributed.f90:465-691  der_penta_periodic  (+ Sherman-Morrison-Woodbury, rank 4)
//   src/backend/omp/exec_dist.f90:188-241            exec_dist_penta_compact / _periodic (no exchange: local)
//   src/tdsops.f90:971-1103                          preprocess_penta_dist (the LU factors, built on the host)
// The reference's backends never dispatch tds_solve to this path (only its verification tests call it); here
// tds_solve takes it whenever the operator was created with scheme = 'compact10_penta'.
//
// One pencil per lane, lanes across x for y / z pencils (contiguous 512-byte rows), rows wave-uniform -> the four
// LU factors of a row come from one scalar load.  Two sweeps: forward = RHS stencil + L-solve (1R + 1W),
// backward = U-solve in place (1R + 1W); periodic: the backward sweep also leaves the four Woodbury coefficients
// of its pencil, and a third pass subtracts Z c4 (the reference rebuilds Z = A^-1 [e_1 e_2 e_{n-1} e_n] and the
// 4 x 4 factors for every group of every call; they only depend on the operator and are built once here).
#include "common.h"

struct PentaTab {
    const real_t *lu;   // [n + 1][4]: 1/d, l1, l2, u1 of row j (1-based)
    const real_t *z;    // periodic: [n + 1][4] = Z(j, 1..4)
    real_t m[16];       // periodic: LU of M = I + W^T Z (Doolittle, row-major)
    real_t c[9];        // bulk stencil
    const real_t *cs;   // [4][9] start stencils, [4][9] end stencils
    real_t alpha, beta, beta_s;
    int n, bulk_only;
};

struct x3d_penta {
    real_t *dev;
    PentaTab tab;
    int periodic, halo_mode;  // halo_mode: 1 periodic image, 2 even mirror, 3 odd mirror, 4 zeros
};

// extended pencil row jj in [-3, n + 4]
template <int HM>
__device__ __forceinline__ real_t penta_row(const real_t *__restrict__ u, long base, long rs, int jj, int n,
                                            const real_t *__restrict__ hs, const real_t *__restrict__ he, int np, int p)
{
    if (jj >= 1 && jj <= n) return u[base + (long)(jj - 1) * rs];
    if (HM == 0) return jj < 1 ? hs[(long)(jj + 3) * np + p] : he[(long)(jj - n - 1) * np + p];
    if (HM == 1) return u[base + (long)((jj < 1 ? jj + n : jj - n) - 1) * rs];
    if (HM == 4) return 0.0;
    // mirror ghosts about row 1 / row n: row 1 - k <-> row 1 + k, row n + k <-> row n - k
    const real_t v = u[base + (long)((jj < 1 ? 2 - jj : 2 * n - jj) - 1) * rs];
    return HM == 2 ? v : -v;
}

template <int HM>
__global__ void __launch_bounds__(64) k_penta_fwd(real_t *__restrict__ du, const real_t *__restrict__ u,
                                                  const real_t *__restrict__ hs, const real_t *__restrict__ he,
                                                  PentaTab t, PencilGeom g)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int n = t.n;
    real_t w[9];
#pragma unroll
    for (int m = 0; m < 9; m++) w[m] = penta_row<HM>(u, base, rs, m - 3, n, hs, he, g.np, p);
    real_t y1 = 0.0, y2 = 0.0;  // y_{j-1}, y_{j-2}
    for (int j = 1; j <= n; j++) {
        const real_t *__restrict__ c = (t.bulk_only || (j > 4 && j <= n - 4)) ? nullptr
                                       : (j <= 4 ? t.cs + (j - 1) * 9 : t.cs + 36 + (j - (n - 4) - 1) * 9);
        real_t r;
        if (c) r = c[0] * w[0] + c[1] * w[1] + c[2] * w[2] + c[3] * w[3] + c[4] * w[4] + c[5] * w[5] + c[6] * w[6] +
                   c[7] * w[7] + c[8] * w[8];
        else r = t.c[1] * w[1] + t.c[2] * w[2] + t.c[3] * w[3] + t.c[4] * w[4] + t.c[5] * w[5] + t.c[6] * w[6] +
                 t.c[7] * w[7];  // (distributed.f90:399-404: the bulk rows take the 7 inner taps)
        const real_t l1 = t.lu[4 * j + 1], l2 = t.lu[4 * j + 2];
        const real_t y = j == 1 ? r : (j == 2 ? r - l1 * y1 : r - l1 * y1 - l2 * y2);  // :432-445
        du[base + (long)(j - 1) * rs] = y;
        y2 = y1; y1 = y;
#pragma unroll
        for (int m = 0; m < 8; m++) w[m] = w[m + 1];
        w[8] = penta_row<HM>(u, base, rs, j + 5, n, hs, he, g.np, p);
    }
}

template <bool PER>
__global__ void __launch_bounds__(64) k_penta_bwd(real_t *__restrict__ du, real_t *__restrict__ c4out, PentaTab t,
                                                  PencilGeom g)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const int n = t.n;
    // :448-462
    real_t xn = du[base + (long)(n - 1) * rs] * t.lu[4 * n];
    real_t xm = (du[base + (long)(n - 2) * rs] - t.lu[4 * (n - 1) + 3] * xn) * t.lu[4 * (n - 1)];
    du[base + (long)(n - 1) * rs] = xn;
    du[base + (long)(n - 2) * rs] = xm;
    const real_t x_n = xn, x_nm1 = xm;
    real_t a1 = xm, a2 = xn;  // x_{j+1}, x_{j+2}
    for (int j = n - 2; j >= 2; j--) {
        const real_t x = (du[base + (long)(j - 1) * rs] - t.lu[4 * j + 3] * a1 - t.beta * a2) * t.lu[4 * j];
        du[base + (long)(j - 1) * rs] = x;
        a2 = a1; a1 = x;
    }
    const real_t x1 = (du[base] - t.lu[4 + 3] * a1 - t.beta_s * a2) * t.lu[4];
    du[base] = x1;
    if (PER) {  // c4 = M^-1 (W^T y), :643-668
        const real_t alp = t.alpha, bet = t.beta;
        real_t c1 = bet * x_nm1 + alp * x_n, c2 = bet * x_n, c3 = bet * x1, c4 = alp * x1 + bet * a1;
        const real_t *M = t.m;
        c2 = c2 - M[4] * c1;
        c3 = c3 - M[8] * c1 - M[9] * c2;
        c4 = c4 - M[12] * c1 - M[13] * c2 - M[14] * c3;
        c4 = c4 / M[15];
        c3 = (c3 - M[11] * c4) / M[10];
        c2 = (c2 - M[6] * c3 - M[7] * c4) / M[5];
        c1 = (c1 - M[1] * c2 - M[2] * c3 - M[3] * c4) / M[0];
        c4out[p] = c1; c4out[(long)g.np + p] = c2; c4out[2L * g.np + p] = c3; c4out[3L * g.np + p] = c4;
    }
}

__global__ void __launch_bounds__(64) k_penta_corr(real_t *__restrict__ du, const real_t *__restrict__ c4, PentaTab t,
                                                   PencilGeom g)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1, rs = g.rs;
    const real_t c1 = c4[p], c2 = c4[(long)g.np + p], c3 = c4[2L * g.np + p], c4v = c4[3L * g.np + p];
    for (int j = 1; j <= t.n; j++) {  // :671-688
        const long o = base + (long)(j - 1) * rs;
        du[o] = du[o] - t.z[4 * j] * c1 - t.z[4 * j + 1] * c2 - t.z[4 * j + 2] * c3 - t.z[4 * j + 3] * c4v;
    }
}

// halo_kind: 1 periodic, 2 even mirror ghosts (BC_NEUMANN, sym), 3 odd mirror ghosts (BC_NEUMANN, not sym),
// 4 zeros (BC_DIRICHLET: the closures do not read the ghosts)
extern "C" int x3d_tdsops_set_penta(x3d_tdsops *t, real_t alpha, real_t beta, real_t beta_lhs_s, const real_t *dist_fw,
                                    const real_t *dist_af, const real_t *dist_sa, const real_t *dist_bw,
                                    const real_t *coeffs_s, const real_t *coeffs_e, int halo_kind)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(t && dist_fw && dist_af && dist_sa && dist_bw && coeffs_s && coeffs_e, "x3d_tdsops_set_penta: null argument");
    X3D_REQUIRE(halo_kind >= 1 && halo_kind <= 4, "x3d_tdsops_set_penta: halo_kind must be 1..4");
    X3D_REQUIRE(t->n_rhs == t->n_tds, "x3d_tdsops_set_penta: n_rhs must equal n_tds");
    const int n = t->n_tds;
    std::vector<real_t> h((size_t)8 * (n + 1) + 72, 0.0);
    real_t *lu = h.data(), *z = lu + 4 * (n + 1), *cs = z + 4 * (n + 1);
    for (int j = 1; j <= n; j++) {
        lu[4 * j] = dist_fw[j - 1]; lu[4 * j + 1] = dist_af[j - 1]; lu[4 * j + 2] = dist_sa[j - 1];
        lu[4 * j + 3] = dist_bw[j - 1];
    }
    memcpy(cs, coeffs_s, sizeof(real_t) * 36);
    memcpy(cs + 36, coeffs_e, sizeof(real_t) * 36);
    x3d_penta *p = new x3d_penta();
    memset(p, 0, sizeof *p);
    p->periodic = halo_kind == 1;
    p->halo_mode = halo_kind;
    PentaTab &tb = p->tab;
    if (p->periodic) {
        // Z(:, k) = A_np^-1 e_pk, p = [1, 2, n-1, n] (distributed.f90:592-610), then M = I + W^T Z and its LU (:612-641)
        for (int k = 0; k < 4; k++) {
            std::vector<real_t> v(n + 1, 0.0);
            const int pk = k == 0 ? 1 : (k == 1 ? 2 : (k == 2 ? n - 1 : n));
            v[pk] = 1.0;
            v[2] = v[2] - dist_af[1] * v[1];
            for (int j = 3; j <= n; j++) v[j] = v[j] - dist_af[j - 1] * v[j - 1] - dist_sa[j - 1] * v[j - 2];
            v[n] = v[n] * dist_fw[n - 1];
            v[n - 1] = (v[n - 1] - dist_bw[n - 2] * v[n]) * dist_fw[n - 2];
            for (int j = n - 2; j >= 2; j--) v[j] = (v[j] - dist_bw[j - 1] * v[j + 1] - beta * v[j + 2]) * dist_fw[j - 1];
            v[1] = (v[1] - dist_bw[0] * v[2] - beta_lhs_s * v[3]) * dist_fw[0];
            for (int j = 1; j <= n; j++) z[4 * j + k] = v[j];
        }
        real_t M[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
        for (int k = 0; k < 4; k++) {
            M[0][k] += beta * z[4 * (n - 1) + k] + alpha * z[4 * n + k];
            M[1][k] += beta * z[4 * n + k];
            M[2][k] += beta * z[4 + k];
            M[3][k] += alpha * z[4 + k] + beta * z[8 + k];
        }
        for (int c = 0; c < 3; c++) {
            for (int r = c + 1; r < 4; r++) M[r][c] = M[r][c] / M[c][c];
            for (int cc = c + 1; cc < 4; cc++)
                for (int r = c + 1; r < 4; r++) M[r][cc] = M[r][cc] - M[r][c] * M[c][cc];
        }
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) tb.m[4 * r + c] = M[r][c];
    }
    X3D_HIP(hipMalloc(&p->dev, sizeof(real_t) * h.size()));
    X3D_HIP(hipMemcpy(p->dev, h.data(), sizeof(real_t) * h.size(), hipMemcpyHostToDevice));
    tb.lu = p->dev; tb.z = p->dev + 4 * (n + 1); tb.cs = p->dev + 8 * (n + 1);
    for (int m = 0; m < 9; m++) tb.c[m] = t->coeffs[m];
    tb.alpha = alpha; tb.beta = beta; tb.beta_s = beta_lhs_s; tb.n = n;
    tb.bulk_only = 1;
    for (int i = 0; i < 72; i++)
        if (cs[i] != t->coeffs[i % 9]) tb.bulk_only = 0;
    if (t->penta) { hipFree(t->penta->dev); delete t->penta; }
    t->penta = p;
    return 0;
}

void x3d_penta_free(x3d_tdsops *t)
{
    if (t && t->penta) { hipFree(t->penta->dev); delete t->penta; t->penta = nullptr; }
}

// du = compact10_penta first derivative of u along dir (exec_dist_penta_compact / _periodic).  u_s / u_e: ghost
// rows [4][np] (rows -3..0 and n+1..n+4) or both NULL: the ghosts implied by the operator's boundary conditions
// (periodic image, even / odd mirror, zeros) are formed in the kernel.
extern "C" int x3d_tds_penta_solve(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir,
                                   const real_t *u_s, const real_t *u_e)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && u && t, "x3d_tds_penta_solve: null argument");
    X3D_REQUIRE(t->penta, "x3d_tds_penta_solve: not a pentadiagonal operator (x3d_tdsops_set_penta)");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_tds_penta_solve: bad dir %d", dir);
    X3D_REQUIRE(du != u, "x3d_tds_penta_solve: du and u must be distinct blocks");
    X3D_REQUIRE((u_s == nullptr) == (u_e == nullptr), "x3d_tds_penta_solve: u_s and u_e go together");
    const int ext = dir == X3D_DIR_X ? b->nx : (dir == X3D_DIR_Y ? b->ny : b->nz);
    X3D_REQUIRE(t->n_tds <= ext && t->n_tds >= 9, "x3d_tds_penta_solve: %d rows do not fit the block / too few", t->n_tds);
    const PencilGeom g = x3d_geom(b, dir);
    const x3d_penta *p = t->penta;
    const dim3 grid((g.np + 63) / 64), blk(64);
    {
        ProfScope ps(b, X3D_K_TDS_FWD, dir);
        const int hm = u_s ? 0 : p->halo_mode;
#define FWD(H_) hipLaunchKernelGGL(k_penta_fwd<H_>, grid, blk, 0, b->stream, du, u, u_s, u_e, p->tab, g)
        if (hm == 0) FWD(0); else if (hm == 1) FWD(1); else if (hm == 2) FWD(2); else if (hm == 3) FWD(3); else FWD(4);
#undef FWD
    }
    {
        ProfScope ps(b, X3D_K_TDS_BWD, dir);
        if (p->periodic) {
            hipLaunchKernelGGL(k_penta_bwd<true>, grid, blk, 0, b->stream, du, b->scratch[2], p->tab, g);
            hipLaunchKernelGGL(k_penta_corr, grid, blk, 0, b->stream, du, (const real_t *)b->scratch[2], p->tab, g);
        } else {
            hipLaunchKernelGGL(k_penta_bwd<false>, grid, blk, 0, b->stream, du, (real_t *)nullptr, p->tab, g);
        }
    }
    X3D_HIP(hipGetLastError());
    return 0;
}
