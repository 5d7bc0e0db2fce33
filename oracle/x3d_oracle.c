/*
 * oracle/x3d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the arithmetic on x3d2's per-timestep hot path,
 * used only as the checker for the HIP backend (tests/, __graft_entry__.smoke,
 * bench.py's cpu_baseline leg).  Nothing under x3d2_amd/ may link, import or
 * call this file.
 *
 * Parity status: PINNED.  Every routine here is checked in
 * tests/test_oracle_vs_reference.py against vectors produced by the REAL
 * reference (its OpenMP backend compiled from /root/reference with ROCm flang,
 * recipe in oracle/ref/) -- see tests/golden/ref_*.npz -- to <= 1e-13 relative.
 *
 * Layout follows the reference's OpenMP backend: a "group" is SZ pencils
 * side by side, data(lane, j, group) with lane fastest
 * (/root/reference/src/backend/omp/common.f90:4, src/ordering.f90:42-69).
 * Each function cites the reference lines it restates.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SZ 16
#define NH 4 /* n_halo, hard-wired in the reference: src/backend/backend.f90:28-29 */

enum { BC_PERIODIC = 0, BC_NEUMANN = 1, BC_DIRICHLET = 2, BC_HALO = -1 };
enum { OP_FIRST_DERIV = 0, OP_SECOND_DERIV = 1, OP_INTERPOLATE = 2, OP_STAG_DERIV = 3 };
enum { SCH_COMPACT6 = 0, SCH_COMPACT6_HYPERVISCOUS = 1, SCH_CLASSIC = 2, SCH_OPTIMISED = 3,
       SCH_AGGRESSIVE = 4 };
enum { FT_NONE = 0, FT_V2P = 1, FT_P2V = 2 };
enum { DIR_X = 1, DIR_Y = 2, DIR_Z = 3, DIR_C = 4 };

/* threads the OpenMP loops below actually run on (bench.py's cpu_baseline reports this, not what it asked for) */
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* tdsops factory: src/tdsops.f90                                             */
/* ------------------------------------------------------------------------- */

typedef struct {
    int n_tds, n_rhs, move, periodic;
    double alpha, a, b, c, d;
    double coeffs[9];
    double coeffs_s[NH][9]; /* coeffs_s(:, i) of the reference = coeffs_s[i-1][:] */
    double coeffs_e[NH][9];
    double *dist_fw, *dist_bw, *dist_sa, *dist_sc, *dist_af; /* length n_rhs */
    double *stretch, *stretch_correct;                       /* length n_tds */
} orc_tdsops;

static void set9(double *dst, double c0, double c1, double c2, double c3, double c4, double c5,
                 double c6, double c7, double c8)
{
    dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = c3; dst[4] = c4;
    dst[5] = c5; dst[6] = c6; dst[7] = c7; dst[8] = c8;
}

static void scale9(double *dst, double s)
{
    for (int i = 0; i < 9; i++) dst[i] = dst[i] * s;
}

static void bulk_rows(orc_tdsops *t)
{
    for (int i = 0; i < NH; i++) {
        memcpy(t->coeffs_s[i], t->coeffs, sizeof t->coeffs);
        memcpy(t->coeffs_e[i], t->coeffs, sizeof t->coeffs);
    }
}

/* src/tdsops.f90:874-931 preprocess_dist (Algorithm 3 of doi 10.1109/MCSE.2021.3130544).
 * 1-based indices of the reference are kept through the macro below. */
static void preprocess_dist(orc_tdsops *t, const double *dist_b_1)
{
#define FW(i) t->dist_fw[(i) - 1]
#define BW(i) t->dist_bw[(i) - 1]
#define SA(i) t->dist_sa[(i) - 1]
#define SC(i) t->dist_sc[(i) - 1]
#define AF(i) t->dist_af[(i) - 1]
#define B(i) dist_b_1[(i) - 1]
    int n = t->n_tds;
    for (int i = 1; i <= 2; i++) { /* :886-891 */
        SA(i) = SA(i) / B(i);
        SC(i) = SC(i) / B(i);
        BW(i) = SC(i);
        AF(i) = 1.0 / B(i);
    }
    for (int i = 3; i <= n; i++) { /* :894-908 */
        FW(i) = 1.0 / (B(i) - SA(i) * SC(i - 1));
        AF(i) = SA(i);
        SA(i) = -FW(i) * SA(i) * SA(i - 1);
        SC(i) = FW(i) * SC(i);
    }
    for (int i = n - 2; i >= 2; i--) { /* :911-916 */
        SA(i) = SA(i) - SC(i) * SA(i + 1);
        BW(i) = SC(i);
        SC(i) = -SC(i) * SC(i + 1);
    }
    FW(1) = 1.0 / (1.0 - SC(1) * SA(2)); /* :925 */
    SA(1) = FW(1) * SA(1);               /* :928 */
    SC(1) = -FW(1) * SC(1) * SC(2);      /* :929 */
#undef FW
#undef BW
#undef SA
#undef SC
#undef AF
#undef B
}

static void fill(double *a, int n, double v)
{
    for (int i = 0; i < n; i++) a[i] = v;
}

/* src/tdsops.f90:205-405 deriv_1st (tridiagonal compact6 only; the
 * pentadiagonal compact10 scheme is outside the hot-path scope) */
static int deriv_1st(orc_tdsops *t, double delta, int scheme, int bc_start, int bc_end, int sym,
                     double *dist_b)
{
    if (scheme != SCH_COMPACT6) return 1;
    double alpha = 1.0 / 3.0, afi = 7.0 / 9.0 / delta, bfi = 1.0 / 36.0 / delta, cfi = 0.0;
    int n = t->n_tds;
    t->alpha = alpha; t->a = afi; t->b = bfi; t->c = cfi;
    set9(t->coeffs, 0.0, -cfi, -bfi, -afi, 0.0, afi, bfi, cfi, 0.0);
    bulk_rows(t);
    fill(t->dist_sa, t->n_rhs, alpha);
    fill(t->dist_sc, t->n_rhs, alpha);
    fill(dist_b, t->n_rhs, 1.0);

    if (bc_start == BC_NEUMANN) { /* :275-303 */
        if (sym) {
            t->dist_sa[0] = 0.0; t->dist_sc[0] = 0.0;
            set9(t->coeffs_s[0], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_s[1], 0, 0, 0, -afi, -bfi, afi, bfi, 0, 0);
        } else {
            t->dist_sa[0] = 0.0; t->dist_sc[0] = 2 * alpha;
            set9(t->coeffs_s[0], 0, 0, 0, 0, 0, 2 * afi, 2 * bfi, 0, 0);
            set9(t->coeffs_s[1], 0, 0, 0, -afi, bfi, afi, bfi, 0, 0);
        }
    } else if (bc_start == BC_DIRICHLET) { /* :304-320 */
        t->dist_sa[0] = 0.0; t->dist_sc[0] = 2.0;
        set9(t->coeffs_s[0], 0, 0, 0, 0, -2.5, 2.0, 0.5, 0, 0);
        for (int i = 0; i < 9; i++) t->coeffs_s[0][i] = t->coeffs_s[0][i] / delta;
        t->dist_sa[1] = 0.25; t->dist_sc[1] = 0.25;
        set9(t->coeffs_s[1], 0, 0, 0, -0.75, 0, 0.75, 0, 0, 0);
        for (int i = 0; i < 9; i++) t->coeffs_s[1][i] = t->coeffs_s[1][i] / delta;
    }
    if (bc_end == BC_NEUMANN) { /* :339-367 */
        if (sym) {
            t->dist_sa[n - 1] = 0.0; t->dist_sc[n - 1] = 0.0;
            set9(t->coeffs_e[3], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_e[2], 0, 0, -bfi, -afi, bfi, afi, 0, 0, 0);
        } else {
            t->dist_sa[n - 1] = 2 * alpha; t->dist_sc[n - 1] = 0.0;
            set9(t->coeffs_e[3], 0, 0, -2 * bfi, -2 * afi, 0, 0, 0, 0, 0);
            set9(t->coeffs_e[2], 0, 0, -bfi, -afi, -bfi, afi, 0, 0, 0);
        }
    } else if (bc_end == BC_DIRICHLET) { /* :368-384 */
        t->dist_sa[n - 1] = 2.0; t->dist_sc[n - 1] = 0.0;
        set9(t->coeffs_e[3], 0, 0, -0.5, -2.0, 2.5, 0, 0, 0, 0);
        for (int i = 0; i < 9; i++) t->coeffs_e[3][i] = t->coeffs_e[3][i] / delta;
        t->dist_sa[n - 2] = 0.25; t->dist_sc[n - 2] = 0.25;
        set9(t->coeffs_e[2], 0, 0, 0, -0.75, 0, 0.75, 0, 0, 0);
        for (int i = 0; i < 9; i++) t->coeffs_e[2][i] = t->coeffs_e[2][i] / delta;
    }
    return 0;
}

/* src/tdsops.f90:407-618 deriv_2nd */
static int deriv_2nd(orc_tdsops *t, double delta, int scheme, int bc_start, int bc_end, int sym,
                     double c_nu, double nu0_nu, double *dist_b)
{
    double d2 = delta * delta;
    double alpha, asi, bsi, csi, dsi;
    int n = t->n_tds;
    if (scheme == SCH_COMPACT6) { /* :437-442 */
        alpha = 2.0 / 11.0;
        asi = 12.0 / 11.0 / d2;
        bsi = 3.0 / 44.0 / d2;
        csi = 0.0;
        dsi = 0.0;
    } else if (scheme == SCH_COMPACT6_HYPERVISCOUS) { /* :443-460 */
        double pi = 4 * atan(1.0);
        double dpis3 = 2.0 * pi / 3.0;
        double xnpi2 = pi * pi * (1.0 + nu0_nu);
        double xmpi2 = dpis3 * dpis3 * (1.0 + c_nu * nu0_nu);
        double den = 405.0 * xnpi2 - 640.0 * xmpi2 + 144.0;
        alpha = 0.5 - (320.0 * xmpi2 - 1296.0) / den;
        asi = -(4329.0 * xnpi2 / 8.0 - 32.0 * xmpi2 - 140.0 * xnpi2 * xmpi2 + 286.0) / den / d2;
        bsi = (2115.0 * xnpi2 - 1792.0 * xmpi2 - 280.0 * xnpi2 * xmpi2 + 1328.0) / den / (4.0 * d2);
        csi = -(7695.0 * xnpi2 / 8.0 + 288.0 * xmpi2 - 180.0 * xnpi2 * xmpi2 - 2574.0) / den /
              (9.0 * d2);
        dsi = (198.0 * xnpi2 + 128.0 * xmpi2 - 40.0 * xnpi2 * xmpi2 - 736.0) / den / (16.0 * d2);
    } else {
        return 1;
    }
    t->alpha = alpha; t->a = asi; t->b = bsi; t->c = csi; t->d = dsi;
    set9(t->coeffs, dsi, csi, bsi, asi, -2.0 * (asi + bsi + csi + dsi), asi, bsi, csi, dsi);
    bulk_rows(t);
    fill(t->dist_sa, t->n_rhs, alpha);
    fill(t->dist_sc, t->n_rhs, alpha);
    fill(dist_b, t->n_rhs, 1.0);

    if (bc_start == BC_NEUMANN) { /* :479-518 */
        if (sym) {
            t->dist_sa[0] = 0.0; t->dist_sc[0] = 2 * alpha;
            set9(t->coeffs_s[0], 0, 0, 0, 0, -2 * asi - 2 * bsi - 2 * csi - 2 * dsi, 2 * asi,
                 2 * bsi, 2 * csi, 2 * dsi);
            set9(t->coeffs_s[1], 0, 0, 0, asi, -2 * asi - bsi - 2 * csi - 2 * dsi, asi + csi,
                 bsi + dsi, csi, dsi);
            set9(t->coeffs_s[2], 0, 0, bsi, asi + csi, -2 * asi - 2 * bsi - 2 * csi - dsi, asi, bsi,
                 csi, dsi);
            set9(t->coeffs_s[3], 0, csi, bsi + dsi, asi, -2 * asi - 2 * bsi - 2 * csi - 2 * dsi, asi,
                 bsi, csi, dsi);
        } else {
            t->dist_sa[0] = 0.0; t->dist_sc[0] = 0.0;
            set9(t->coeffs_s[0], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_s[1], 0, 0, 0, asi, -2 * asi - 3 * bsi - 2 * csi - 2 * dsi, asi - csi,
                 bsi - dsi, csi, dsi);
            set9(t->coeffs_s[2], 0, 0, bsi, asi - csi, -2 * asi - 2 * bsi - 2 * csi - 3 * dsi, asi,
                 bsi, csi, dsi);
            set9(t->coeffs_s[3], 0, -csi, bsi - dsi, asi, -2 * asi - 2 * bsi - 2 * csi - 2 * dsi,
                 asi, bsi, csi, dsi);
        }
    } else if (bc_start == BC_DIRICHLET) { /* :519-545 */
        t->dist_sa[0] = 0.0; t->dist_sc[0] = 11.0;
        set9(t->coeffs_s[0], 0, 0, 0, 0, 13.0 / d2, -27.0 / d2, 15.0 / d2, -1.0 / d2, 0);
        t->dist_sa[1] = 0.1; t->dist_sc[1] = 0.1;
        set9(t->coeffs_s[1], 0, 0, 0, 1.2 / d2, -2.4 / d2, 1.2 / d2, 0, 0, 0);
        t->dist_sa[2] = 2.0 / 11.0; t->dist_sc[2] = 2.0 / 11.0;
        double temp1 = 3.0 / 44.0 / d2, temp2 = 12.0 / 11.0 / d2;
        set9(t->coeffs_s[2], 0, 0, temp1, temp2, -2.0 * (temp1 + temp2), temp2, temp1, 0, 0);
        t->dist_sa[3] = 2.0 / 11.0; t->dist_sc[3] = 2.0 / 11.0;
        memcpy(t->coeffs_s[3], t->coeffs_s[2], sizeof t->coeffs_s[2]);
    }
    if (bc_end == BC_NEUMANN) { /* :548-587 */
        if (sym) {
            t->dist_sa[n - 1] = 2 * alpha; t->dist_sc[n - 1] = 0.0;
            set9(t->coeffs_e[3], 2 * dsi, 2 * csi, 2 * bsi, 2 * asi,
                 -2 * asi - 2 * bsi - 2 * csi - 2 * dsi, 0, 0, 0, 0);
            set9(t->coeffs_e[2], dsi, csi, bsi + dsi, asi + csi, -2 * asi - bsi - 2 * csi - 2 * dsi,
                 asi, 0, 0, 0);
            set9(t->coeffs_e[1], dsi, csi, bsi, asi, -2 * asi - 2 * bsi - 2 * csi - dsi, asi + csi,
                 bsi, 0, 0);
            set9(t->coeffs_e[0], dsi, csi, bsi, asi, -2 * asi - 2 * bsi - 2 * csi - 2 * dsi, asi,
                 bsi + dsi, csi, 0);
        } else {
            t->dist_sa[n - 1] = 0.0; t->dist_sc[n - 1] = 0.0;
            set9(t->coeffs_e[3], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_e[2], dsi, csi, bsi - dsi, asi - csi,
                 -2 * asi - 3 * bsi - 2 * csi - 2 * dsi, asi, 0, 0, 0);
            set9(t->coeffs_e[1], dsi, csi, bsi, asi, -2 * asi - 2 * bsi - 2 * csi - 3 * dsi,
                 asi - csi, bsi, 0, 0);
            set9(t->coeffs_e[0], dsi, csi, bsi, asi, -2 * asi - 2 * bsi - 2 * csi - 2 * dsi, asi,
                 bsi - dsi, -csi, 0);
        }
    } else if (bc_end == BC_DIRICHLET) { /* :588-613 */
        t->dist_sa[n - 1] = 11.0; t->dist_sc[n - 1] = 0.0;
        set9(t->coeffs_e[3], 0, -1.0 / d2, 15.0 / d2, -27.0 / d2, 13.0 / d2, 0, 0, 0, 0);
        t->dist_sa[n - 2] = 0.1; t->dist_sc[n - 2] = 0.1;
        set9(t->coeffs_e[2], 0, 0, 0, 1.2 / d2, -2.4 / d2, 1.2 / d2, 0, 0, 0);
        t->dist_sa[n - 3] = 2.0 / 11.0; t->dist_sc[n - 3] = 2.0 / 11.0;
        double temp1 = 3.0 / 44.0 / d2, temp2 = 12.0 / 11.0 / d2;
        set9(t->coeffs_e[1], 0, 0, temp1, temp2, -2.0 * (temp1 + temp2), temp2, temp1, 0, 0);
        t->dist_sa[n - 4] = 2.0 / 11.0; t->dist_sc[n - 4] = 2.0 / 11.0;
        memcpy(t->coeffs_e[0], t->coeffs_e[1], sizeof t->coeffs_e[1]);
    }
    return 0;
}

/* src/tdsops.f90:620-764 interpl_mid */
static int interpl_mid(orc_tdsops *t, int scheme, int from_to, int bc_start, int bc_end,
                       double *dist_b)
{
    double alpha, aici, bici, cici, dici;
    int n = t->n_tds;
    if (scheme == SCH_CLASSIC) { /* :637-642 */
        alpha = 0.3; aici = 0.75; bici = 0.05; cici = 0.0; dici = 0.0;
    } else if (scheme == SCH_OPTIMISED) { /* :643-648 */
        alpha = 0.461658;
        dici = 0.00146508;
        aici = (75.0 + 70.0 * alpha - 640.0 * dici) / 128.0;
        bici = (-25.0 + 126.0 * alpha + 2304.0 * dici) / 256.0;
        cici = (3.0 - 10.0 * alpha - 1280.0 * dici) / 256.0;
    } else if (scheme == SCH_AGGRESSIVE) { /* :649-654 */
        alpha = 0.49;
        aici = (75.0 + 70.0 * alpha) / 128.0;
        bici = (-25.0 + 126.0 * alpha) / 256.0;
        cici = (3.0 - 10.0 * alpha) / 256.0;
        dici = 0.0;
    } else {
        return 1;
    }
    t->alpha = alpha; t->a = aici; t->b = bici; t->c = cici; t->d = dici;
    if (from_to == FT_V2P)
        set9(t->coeffs, 0.0, dici, cici, bici, aici, aici, bici, cici, dici);
    else if (from_to == FT_P2V)
        set9(t->coeffs, dici, cici, bici, aici, aici, bici, cici, dici, 0.0);
    else
        return 1;
    bulk_rows(t);
    fill(t->dist_sa, t->n_rhs, alpha);
    fill(t->dist_sc, t->n_rhs, alpha);
    fill(dist_b, t->n_rhs, 1.0);

    if (bc_start == BC_NEUMANN) { /* :686-720 */
        t->dist_sa[0] = 0.0;
        if (from_to == FT_V2P) {
            dist_b[0] = 1.0 + alpha;
            set9(t->coeffs_s[0], 0, 0, 0, 0, aici, aici + bici, bici + cici, cici + dici, dici);
            set9(t->coeffs_s[1], 0, 0, 0, bici, aici + cici, aici + dici, bici, cici, dici);
            set9(t->coeffs_s[2], 0, 0, cici, bici + dici, aici, aici, bici, cici, dici);
        } else {
            t->dist_sc[0] = 2 * alpha;
            set9(t->coeffs_s[0], 0, 0, 0, 0, 2 * aici, 2 * bici, 2 * cici, 2 * dici, 0);
            set9(t->coeffs_s[1], 0, 0, 0, aici + bici, aici + cici, bici + dici, cici, dici, 0);
            set9(t->coeffs_s[2], 0, 0, bici + cici, aici + dici, aici, bici, cici, dici, 0);
            set9(t->coeffs_s[3], 0, cici + dici, bici, aici, aici, bici, cici, dici, 0);
        }
    } else if (bc_start == BC_DIRICHLET) {
        return 2; /* reference: error stop, :722 */
    }
    if (bc_end == BC_NEUMANN) { /* :726-758 */
        t->dist_sc[n - 1] = 0.0;
        if (from_to == FT_V2P) {
            dist_b[n - 1] = 1.0 + alpha;
            set9(t->coeffs_e[3], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_e[2], 0, dici, cici + dici, bici + cici, aici + bici, aici, 0, 0, 0);
            set9(t->coeffs_e[1], 0, dici, cici, bici, aici + dici, aici + cici, bici, 0, 0);
            set9(t->coeffs_e[0], 0, dici, cici, bici, aici, aici, bici + dici, cici, 0);
        } else {
            t->dist_sa[n - 1] = 2 * alpha;
            set9(t->coeffs_e[3], 2 * dici, 2 * cici, 2 * bici, 2 * aici, 0, 0, 0, 0, 0);
            set9(t->coeffs_e[2], dici, cici, bici + dici, aici + cici, aici + bici, 0, 0, 0, 0);
            set9(t->coeffs_e[1], dici, cici, bici, aici, aici + dici, bici + cici, 0, 0, 0);
            set9(t->coeffs_e[0], dici, cici, bici, aici, aici, bici, cici + dici, 0, 0);
        }
    } else if (bc_end == BC_DIRICHLET) {
        return 2;
    }
    return 0;
}

/* src/tdsops.f90:766-872 stagder_1st */
static int stagder_1st(orc_tdsops *t, double delta, int scheme, int from_to, int bc_start,
                       int bc_end, double *dist_b)
{
    if (scheme != SCH_COMPACT6) return 1;
    double alpha = 9.0 / 62.0, aci = 63.0 / 62.0 / delta, bci = 17.0 / 62.0 / 3.0 / delta;
    int n = t->n_tds;
    t->alpha = alpha; t->a = aci; t->b = bci;
    if (from_to == FT_V2P)
        set9(t->coeffs, 0, 0, 0, -bci, -aci, aci, bci, 0, 0);
    else if (from_to == FT_P2V)
        set9(t->coeffs, 0, 0, -bci, -aci, aci, bci, 0, 0, 0);
    else
        return 1;
    bulk_rows(t);
    fill(t->dist_sa, t->n_rhs, alpha);
    fill(t->dist_sc, t->n_rhs, alpha);
    fill(dist_b, t->n_rhs, 1.0);

    if (bc_start == BC_NEUMANN) { /* :817-840 */
        t->dist_sa[0] = 0.0;
        if (from_to == FT_V2P) {
            dist_b[0] = 1.0 + alpha;
            set9(t->coeffs_s[0], 0, 0, 0, 0, -aci - 2 * bci, aci + bci, bci, 0, 0);
            set9(t->coeffs_s[1], 0, 0, 0, -bci, -aci, aci, bci, 0, 0);
        } else {
            t->dist_sc[0] = 0.0;
            set9(t->coeffs_s[0], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_s[1], 0, 0, 0, -aci - bci, aci, bci, 0, 0, 0);
        }
    } else if (bc_start == BC_DIRICHLET) {
        return 2;
    }
    if (bc_end == BC_NEUMANN) { /* :845-865 */
        t->dist_sc[n - 1] = 0.0;
        if (from_to == FT_V2P) {
            dist_b[n - 1] = 1.0 + alpha;
            set9(t->coeffs_e[3], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_e[2], 0, 0, 0, -bci, -aci - bci, aci + 2 * bci, 0, 0, 0);
        } else {
            t->dist_sa[n - 1] = 0.0;
            set9(t->coeffs_e[3], 0, 0, 0, 0, 0, 0, 0, 0, 0);
            set9(t->coeffs_e[2], 0, 0, -bci, -aci, aci + bci, 0, 0, 0, 0);
        }
    } else if (bc_end == BC_DIRICHLET) {
        return 2;
    }
    return 0;
}

/* src/tdsops.f90:63-203 tdsops_init.  The caller owns all arrays:
 * dist_* have n_rhs entries (n_rhs = n_tds+1 only for v2p with a non-periodic
 * end, :114-123), stretch* have n_tds entries.  scalars_out =
 * {n_tds, n_rhs, move, periodic, alpha, a, b, c, d}.  Entries the reference
 * leaves unassigned (dist_fw(2); fw/bw/af(n_tds+1)) are 0 here. */
int orc_tdsops_init(int n_tds, double delta, int operation, int scheme, int bc_start, int bc_end,
                    const double *stretch, const double *stretch_correct, int from_to, int sym,
                    double c_nu, double nu0_nu, double *scalars_out, double *coeffs,
                    double *coeffs_s, double *coeffs_e, double *dist_fw, double *dist_bw,
                    double *dist_sa, double *dist_sc, double *dist_af, double *stretch_out,
                    double *stretch_correct_out)
{
    orc_tdsops t;
    memset(&t, 0, sizeof t);
    t.n_tds = n_tds;
    if (from_to == FT_V2P && (bc_end == BC_NEUMANN || bc_end == BC_DIRICHLET))
        t.n_rhs = n_tds + 1;
    else
        t.n_rhs = n_tds;
    t.dist_fw = dist_fw; t.dist_bw = dist_bw; t.dist_sa = dist_sa; t.dist_sc = dist_sc;
    t.dist_af = dist_af; t.stretch = stretch_out; t.stretch_correct = stretch_correct_out;
    fill(dist_fw, t.n_rhs, 0.0); fill(dist_bw, t.n_rhs, 0.0); fill(dist_af, t.n_rhs, 0.0);
    for (int i = 0; i < n_tds; i++) {
        stretch_out[i] = stretch ? stretch[i] : 1.0;
        stretch_correct_out[i] = stretch_correct ? stretch_correct[i] : 0.0;
    }
    t.periodic = (bc_start == BC_PERIODIC && bc_end == BC_PERIODIC);
    double *dist_b = (double *)malloc(sizeof(double) * (size_t)t.n_rhs);
    int rc;
    switch (operation) {
    case OP_FIRST_DERIV: rc = deriv_1st(&t, delta, scheme, bc_start, bc_end, sym, dist_b); break;
    case OP_SECOND_DERIV:
        rc = deriv_2nd(&t, delta, scheme, bc_start, bc_end, sym, c_nu, nu0_nu, dist_b);
        break;
    case OP_INTERPOLATE: rc = interpl_mid(&t, scheme, from_to, bc_start, bc_end, dist_b); break;
    case OP_STAG_DERIV: rc = stagder_1st(&t, delta, scheme, from_to, bc_start, bc_end, dist_b); break;
    default: rc = 3;
    }
    if (rc == 0) preprocess_dist(&t, dist_b);
    free(dist_b);
    if (rc) return rc;
    t.move = from_to == FT_V2P ? 1 : (from_to == FT_P2V ? -1 : 0);
    scalars_out[0] = t.n_tds; scalars_out[1] = t.n_rhs; scalars_out[2] = t.move;
    scalars_out[3] = t.periodic; scalars_out[4] = t.alpha; scalars_out[5] = t.a;
    scalars_out[6] = t.b; scalars_out[7] = t.c; scalars_out[8] = t.d;
    memcpy(coeffs, t.coeffs, sizeof t.coeffs);
    memcpy(coeffs_s, t.coeffs_s, sizeof t.coeffs_s);
    memcpy(coeffs_e, t.coeffs_e, sizeof t.coeffs_e);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* DistD2 kernels: src/backend/omp/kernels/distributed.f90                    */
/* One group = data[(j-1)*SZ + lane], j = 1..n                                */
/* ------------------------------------------------------------------------- */

#define U(j) (u + ((j) - 1) * SZ)
#define DU(j) (du + ((j) - 1) * SZ)

/* value of the (1-based) extended pencil index jj in [-3, n+4]: halo below 1
 * comes from u_s(:,1:4) (= rows -3..0), above n_ext from u_e(:,1:4) */
static inline const double *ext_row(const double *u, const double *u_s, const double *u_e, int n_in,
                                    int jj)
{
    if (jj < 1) return u_s + (jj + NH - 1) * SZ;
    if (jj > n_in) return u_e + (jj - n_in - 1) * SZ;
    return U(jj);
}

/* src/backend/omp/kernels/distributed.f90:11-168 der_univ_dist.
 * n_in = number of input rows held in u (= n_rhs): rows beyond it come from
 * the end halo exactly as the reference's explicit u_e(i,1..4) terms do. */
static void der_univ_dist(double *du, double *send_u_s, double *send_u_e, const double *u,
                          const double *u_s, const double *u_e, int n_tds, int n_rhs,
                          const double *coeffs_s, const double *coeffs_e, const double *coeffs,
                          const double *ffr, const double *fbc, const double *faf)
{
    double last_r = ffr[0];
    /* rows 1..4 with the start stencils (:34-79) */
    for (int r = 1; r <= 4; r++) {
        const double *c = coeffs_s + (r - 1) * 9;
        const double *p[9];
        for (int m = 0; m < 9; m++) p[m] = ext_row(u, u_s, u_e, n_rhs, r - 4 + m);
#pragma omp simd
        for (int i = 0; i < SZ; i++) {
            double acc = c[0] * p[0][i] + c[1] * p[1][i] + c[2] * p[2][i] + c[3] * p[3][i] +
                         c[4] * p[4][i] + c[5] * p[5][i] + c[6] * p[6][i] + c[7] * p[7][i] +
                         c[8] * p[8][i];
            if (r <= 2)
                DU(r)[i] = acc * faf[r - 1];
            else
                DU(r)[i] = ffr[r - 1] * (acc - faf[r - 1] * DU(r - 1)[i]);
        }
    }
    /* bulk (:82-96): one alpha for all interior rows */
    double alpha = faf[4];
    for (int j = 5; j <= n_rhs - 4; j++) {
#pragma omp simd
        for (int i = 0; i < SZ; i++) {
            double acc = coeffs[0] * U(j - 4)[i] + coeffs[1] * U(j - 3)[i] + coeffs[2] * U(j - 2)[i] +
                         coeffs[3] * U(j - 1)[i] + coeffs[4] * U(j)[i] + coeffs[5] * U(j + 1)[i] +
                         coeffs[6] * U(j + 2)[i] + coeffs[7] * U(j + 3)[i] + coeffs[8] * U(j + 4)[i];
            DU(j)[i] = ffr[j - 1] * (acc - alpha * DU(j - 1)[i]);
        }
    }
    /* rows n_rhs-3..n_rhs with the end stencils (:98-145) */
    for (int r = 1; r <= 4; r++) {
        int j = n_rhs - 4 + r;
        const double *c = coeffs_e + (r - 1) * 9;
        const double *p[9];
        for (int m = 0; m < 9; m++) p[m] = ext_row(u, u_s, u_e, n_rhs, j - 4 + m);
#pragma omp simd
        for (int i = 0; i < SZ; i++) {
            double acc = c[0] * p[0][i] + c[1] * p[1][i] + c[2] * p[2][i] + c[3] * p[3][i] +
                         c[4] * p[4][i] + c[5] * p[5][i] + c[6] * p[6][i] + c[7] * p[7][i] +
                         c[8] * p[8][i];
            DU(j)[i] = ffr[j - 1] * (acc - faf[j - 1] * DU(j - 1)[i]);
        }
    }
    for (int i = 0; i < SZ; i++) send_u_e[i] = DU(n_tds)[i]; /* :147-151 */
    for (int j = n_tds - 2; j >= 2; j--) {                    /* :154-160 */
#pragma omp simd
        for (int i = 0; i < SZ; i++) DU(j)[i] = DU(j)[i] - fbc[j - 1] * DU(j + 1)[i];
    }
    for (int i = 0; i < SZ; i++) { /* :161-166 */
        DU(1)[i] = last_r * (DU(1)[i] - fbc[0] * DU(2)[i]);
        send_u_s[i] = DU(1)[i];
    }
}

/* :170-229 der_univ_subs */
static void der_univ_subs(double *du, const double *recv_u_s, const double *recv_u_e, int n,
                          const double *dist_sa, const double *dist_sc, const double *strch)
{
    double du_s[SZ], du_e[SZ];
    for (int i = 0; i < SZ; i++) {
        double bl = dist_sa[0], ur = dist_sa[0];
        double recp = 1.0 / (1.0 - ur * bl);
        du_s[i] = recp * (DU(1)[i] - bl * recv_u_s[i]);
        bl = dist_sc[n - 1];
        ur = dist_sc[n - 1];
        recp = 1.0 / (1.0 - ur * bl);
        du_e[i] = recp * (DU(n)[i] - ur * recv_u_e[i]);
    }
    for (int i = 0; i < SZ; i++) DU(1)[i] = du_s[i] * strch[0];
    for (int j = 2; j <= n - 1; j++) {
#pragma omp simd
        for (int i = 0; i < SZ; i++)
            DU(j)[i] = (DU(j)[i] - dist_sa[j - 1] * du_s[i] - dist_sc[j - 1] * du_e[i]) * strch[j - 1];
    }
    for (int i = 0; i < SZ; i++) DU(n)[i] = du_e[i] * strch[n - 1];
}

/* :231-337 der_univ_fused_subs */
static void der_univ_fused_subs(double *rhs_du, const double *dud, const double *d2u,
                                const double *v, const double *du_recv_s, const double *du_recv_e,
                                const double *dud_recv_s, const double *dud_recv_e,
                                const double *d2u_recv_s, const double *d2u_recv_e, double nu, int n,
                                const double *du_sa, const double *du_sc, const double *du_strch,
                                const double *dud_sa, const double *dud_sc, const double *dud_strch,
                                const double *d2u_sa, const double *d2u_sc, const double *d2u_strch,
                                const double *d2u_strch_cor)
{
#define R(a, j) ((a) + ((j) - 1) * SZ)
    double du_s[SZ], du_e[SZ], dud_s[SZ], dud_e[SZ], d2u_s[SZ], d2u_e[SZ];
    for (int i = 0; i < SZ; i++) {
        double bl, ur, recp;
        bl = du_sa[0]; ur = du_sa[0]; recp = 1.0 / (1.0 - ur * bl);
        du_s[i] = recp * (R(rhs_du, 1)[i] - bl * du_recv_s[i]);
        bl = dud_sa[0]; ur = dud_sa[0]; recp = 1.0 / (1.0 - ur * bl);
        dud_s[i] = recp * (R(dud, 1)[i] - bl * dud_recv_s[i]);
        bl = d2u_sa[0]; ur = d2u_sa[0]; recp = 1.0 / (1.0 - ur * bl);
        d2u_s[i] = recp * (R(d2u, 1)[i] - bl * d2u_recv_s[i]);
        bl = du_sc[n - 1]; ur = du_sc[n - 1]; recp = 1.0 / (1.0 - ur * bl);
        du_e[i] = recp * (R(rhs_du, n)[i] - ur * du_recv_e[i]);
        bl = dud_sc[n - 1]; ur = dud_sc[n - 1]; recp = 1.0 / (1.0 - ur * bl);
        dud_e[i] = recp * (R(dud, n)[i] - ur * dud_recv_e[i]);
        bl = d2u_sc[n - 1]; ur = d2u_sc[n - 1]; recp = 1.0 / (1.0 - ur * bl);
        d2u_e[i] = recp * (R(d2u, n)[i] - ur * d2u_recv_e[i]);
    }
    for (int i = 0; i < SZ; i++)
        R(rhs_du, 1)[i] = -0.5 * (R(v, 1)[i] * du_s[i] * du_strch[0] + dud_s[i] * dud_strch[0]) +
                          nu * (d2u_s[i] * d2u_strch[0] + du_s[i] * du_strch[0] * d2u_strch_cor[0]);
    for (int j = 2; j <= n - 1; j++) {
#pragma omp simd
        for (int i = 0; i < SZ; i++) {
            double temp_du =
                du_strch[j - 1] * (R(rhs_du, j)[i] - du_sa[j - 1] * du_s[i] - du_sc[j - 1] * du_e[i]);
            double temp_dud =
                dud_strch[j - 1] * (R(dud, j)[i] - dud_sa[j - 1] * dud_s[i] - dud_sc[j - 1] * dud_e[i]);
            double temp_d2u =
                d2u_strch[j - 1] * (R(d2u, j)[i] - d2u_sa[j - 1] * d2u_s[i] - d2u_sc[j - 1] * d2u_e[i]) +
                temp_du * d2u_strch_cor[j - 1];
            R(rhs_du, j)[i] = -0.5 * (R(v, j)[i] * temp_du + temp_dud) + nu * temp_d2u;
        }
    }
    for (int i = 0; i < SZ; i++)
        R(rhs_du, n)[i] =
            -0.5 * (R(v, n)[i] * du_e[i] * du_strch[n - 1] + dud_e[i] * dud_strch[n - 1]) +
            nu * (d2u_e[i] * d2u_strch[n - 1] + du_e[i] * du_strch[n - 1] * d2u_strch_cor[n - 1]);
#undef R
}

/* Operator handed over from Python as flat arrays. */
typedef struct {
    int n_tds, n_rhs;
    const double *coeffs, *coeffs_s, *coeffs_e;
    const double *dist_fw, *dist_bw, *dist_sa, *dist_sc, *dist_af, *stretch, *stretch_correct;
} orc_op;

/* src/backend/omp/backend.f90:714-737 copy_into_buffers.
 * field(lane, j, k) with leading dims (SZ, n_pad); buffers (SZ, 4, k). */
void orc_copy_into_buffers(double *send_s, double *send_e, const double *u, int n, int n_pad,
                           int n_groups)
{
#pragma omp parallel for
    for (int k = 0; k < n_groups; k++)
        for (int j = 0; j < NH; j++)
            for (int i = 0; i < SZ; i++) {
                send_s[((size_t)k * NH + j) * SZ + i] = u[((size_t)k * n_pad + j) * SZ + i];
                send_e[((size_t)k * NH + j) * SZ + i] = u[((size_t)k * n_pad + (n - NH + j)) * SZ + i];
            }
}

/* src/backend/omp/exec_dist.f90:36-47, first parallel loop of
 * exec_dist_tds_compact */
void orc_tds_phase1(double *du, double *du_send_s, double *du_send_e, const double *u,
                    const double *u_recv_s, const double *u_recv_e, const orc_op *op, int n_pad,
                    int n_groups)
{
#pragma omp parallel for
    for (int k = 0; k < n_groups; k++)
        der_univ_dist(du + (size_t)k * n_pad * SZ, du_send_s + (size_t)k * SZ,
                      du_send_e + (size_t)k * SZ, u + (size_t)k * n_pad * SZ,
                      u_recv_s + (size_t)k * NH * SZ, u_recv_e + (size_t)k * NH * SZ, op->n_tds,
                      op->n_rhs, op->coeffs_s, op->coeffs_e, op->coeffs, op->dist_fw, op->dist_bw,
                      op->dist_af);
}

/* src/backend/omp/exec_dist.f90:55-63, second parallel loop */
void orc_tds_phase2(double *du, const double *du_recv_s, const double *du_recv_e, const orc_op *op,
                    int n_pad, int n_groups)
{
#pragma omp parallel for
    for (int k = 0; k < n_groups; k++)
        der_univ_subs(du + (size_t)k * n_pad * SZ, du_recv_s + (size_t)k * SZ,
                      du_recv_e + (size_t)k * SZ, op->n_tds, op->dist_sa, op->dist_sc, op->stretch);
}

/* src/backend/omp/exec_dist.f90:114-160, first loop of exec_dist_transeq_compact */
void orc_transeq_phase1(double *rhs_du, double *dud, double *d2u, double *du_send_s,
                        double *du_send_e, double *dud_send_s, double *dud_send_e,
                        double *d2u_send_s, double *d2u_send_e, const double *u,
                        const double *u_recv_s, const double *u_recv_e, const double *v,
                        const double *v_recv_s, const double *v_recv_e, const orc_op *op_du,
                        const orc_op *op_dud, const orc_op *op_d2u, int n_pad, int n_groups)
{
#pragma omp parallel
    {
        int n = op_dud->n_tds;
        double *ud = (double *)malloc(sizeof(double) * (size_t)n * SZ);
        double ud_s[NH * SZ], ud_e[NH * SZ];
#pragma omp for
        for (int k = 0; k < n_groups; k++) {
            size_t o = (size_t)k * n_pad * SZ, ob = (size_t)k * SZ, oh = (size_t)k * NH * SZ;
            der_univ_dist(rhs_du + o, du_send_s + ob, du_send_e + ob, u + o, u_recv_s + oh,
                          u_recv_e + oh, op_du->n_tds, op_du->n_rhs, op_du->coeffs_s,
                          op_du->coeffs_e, op_du->coeffs, op_du->dist_fw, op_du->dist_bw,
                          op_du->dist_af);
            der_univ_dist(d2u + o, d2u_send_s + ob, d2u_send_e + ob, u + o, u_recv_s + oh,
                          u_recv_e + oh, op_d2u->n_tds, op_d2u->n_rhs, op_d2u->coeffs_s,
                          op_d2u->coeffs_e, op_d2u->coeffs, op_d2u->dist_fw, op_d2u->dist_bw,
                          op_d2u->dist_af);
            for (int j = 0; j < n; j++)
#pragma omp simd
                for (int i = 0; i < SZ; i++) ud[j * SZ + i] = u[o + j * SZ + i] * v[o + j * SZ + i];
            for (int j = 0; j < NH; j++)
                for (int i = 0; i < SZ; i++) {
                    ud_s[j * SZ + i] = u_recv_s[oh + j * SZ + i] * v_recv_s[oh + j * SZ + i];
                    ud_e[j * SZ + i] = u_recv_e[oh + j * SZ + i] * v_recv_e[oh + j * SZ + i];
                }
            der_univ_dist(dud + o, dud_send_s + ob, dud_send_e + ob, ud, ud_s, ud_e, op_dud->n_tds,
                          op_dud->n_rhs, op_dud->coeffs_s, op_dud->coeffs_e, op_dud->coeffs,
                          op_dud->dist_fw, op_dud->dist_bw, op_dud->dist_af);
        }
        free(ud);
    }
}

/* src/backend/omp/exec_dist.f90:170-184, second loop */
void orc_transeq_phase2(double *rhs_du, const double *dud, const double *d2u, const double *v,
                        const double *du_recv_s, const double *du_recv_e, const double *dud_recv_s,
                        const double *dud_recv_e, const double *d2u_recv_s,
                        const double *d2u_recv_e, double nu, const orc_op *op_du,
                        const orc_op *op_dud, const orc_op *op_d2u, int n_pad, int n_groups)
{
#pragma omp parallel for
    for (int k = 0; k < n_groups; k++) {
        size_t o = (size_t)k * n_pad * SZ, ob = (size_t)k * SZ;
        der_univ_fused_subs(rhs_du + o, dud + o, d2u + o, v + o, du_recv_s + ob, du_recv_e + ob,
                            dud_recv_s + ob, dud_recv_e + ob, d2u_recv_s + ob, d2u_recv_e + ob, nu,
                            op_du->n_tds, op_du->dist_sa, op_du->dist_sc, op_du->stretch,
                            op_dud->dist_sa, op_dud->dist_sc, op_dud->stretch, op_d2u->dist_sa,
                            op_d2u->dist_sc, op_d2u->stretch, op_d2u->stretch_correct);
    }
}

/* ------------------------------------------------------------------------- */
/* index maps and reorders: src/ordering.f90, src/backend/omp/backend.f90     */
/* ------------------------------------------------------------------------- */

/* src/ordering.f90:11-40 get_index_ijk (all indices 1-based) */
static void get_index_ijk(int *i, int *j, int *k, int di, int dj, int dk, int dir, int nxp, int nyp)
{
    switch (dir) {
    case DIR_X: *i = dj; *j = ((dk - 1) % (nyp / SZ)) * SZ + di; *k = 1 + (dk - 1) / (nyp / SZ); break;
    case DIR_Y: *i = ((dk - 1) % (nxp / SZ)) * SZ + di; *j = dj; *k = 1 + (dk - 1) / (nxp / SZ); break;
    case DIR_Z: *i = ((dk - 1) % (nxp / SZ)) * SZ + di; *j = 1 + (dk - 1) / (nxp / SZ); *k = dj; break;
    default: *i = di; *j = dj; *k = dk;
    }
}

/* src/ordering.f90:42-69 get_index_dir */
static void get_index_dir(int *di, int *dj, int *dk, int i, int j, int k, int dir, int nxp, int nyp)
{
    switch (dir) {
    case DIR_X: *di = (j - 1) % SZ + 1; *dj = i; *dk = (nyp / SZ) * (k - 1) + 1 + (j - 1) / SZ; break;
    case DIR_Y: *di = (i - 1) % SZ + 1; *dj = j; *dk = (nxp / SZ) * (k - 1) + 1 + (i - 1) / SZ; break;
    case DIR_Z: *di = (i - 1) % SZ + 1; *dj = k; *dk = (nxp / SZ) * (j - 1) + 1 + (i - 1) / SZ; break;
    default: *di = i; *dj = j; *dk = k;
    }
}

static void dims_of(int dir, int nxp, int nyp, int nzp, int *d)
{
    /* src/allocator.f90:82-90 dims_padded_dir */
    switch (dir) {
    case DIR_X: d[0] = SZ; d[1] = nxp; d[2] = nyp * nzp / SZ; break;
    case DIR_Y: d[0] = SZ; d[1] = nyp; d[2] = nxp * nzp / SZ; break;
    case DIR_Z: d[0] = SZ; d[1] = nzp; d[2] = nxp * nyp / SZ; break;
    default: d[0] = nxp; d[1] = nyp; d[2] = nzp;
    }
}

/* src/backend/omp/backend.f90:393-452 reorder_omp: u_(map(i,j,k)) = u(i,j,k);
 * accumulate != 0 gives sum_intox_omp (:470-527): u(i,j,k) += u_(map(i,j,k))
 * with `from` = the field being looped over. */
void orc_reorder(double *dst, const double *src, int dir_from, int dir_to, int nxp, int nyp, int nzp)
{
    int d[3], e[3];
    dims_of(dir_from, nxp, nyp, nzp, d);
    dims_of(dir_to, nxp, nyp, nzp, e);
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= d[2]; k++)
        for (int j = 1; j <= d[1]; j++)
            for (int i = 1; i <= d[0]; i++) {
                int ci, cj, ck, oi, oj, ok;
                get_index_ijk(&ci, &cj, &ck, i, j, k, dir_from, nxp, nyp);
                get_index_dir(&oi, &oj, &ok, ci, cj, ck, dir_to, nxp, nyp);
                dst[((size_t)(ok - 1) * e[1] + (oj - 1)) * e[0] + (oi - 1)] =
                    src[((size_t)(k - 1) * d[1] + (j - 1)) * d[0] + (i - 1)];
            }
}

void orc_sum_intox(double *u_x, const double *u_dir, int dir_to, int nxp, int nyp, int nzp)
{
    int d[3], e[3];
    dims_of(DIR_X, nxp, nyp, nzp, d);
    dims_of(dir_to, nxp, nyp, nzp, e);
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= d[2]; k++)
        for (int j = 1; j <= d[1]; j++)
            for (int i = 1; i <= d[0]; i++) {
                int ci, cj, ck, oi, oj, ok;
                get_index_ijk(&ci, &cj, &ck, i, j, k, DIR_X, nxp, nyp);
                get_index_dir(&oi, &oj, &ok, ci, cj, ck, dir_to, nxp, nyp);
                size_t a = ((size_t)(k - 1) * d[1] + (j - 1)) * d[0] + (i - 1);
                u_x[a] = u_x[a] + u_dir[((size_t)(ok - 1) * e[1] + (oj - 1)) * e[0] + (oi - 1)];
            }
}

/* src/backend/omp/backend.f90:529-614 veccopy/vecadd/vecmult over whole padded blocks */
void orc_vecadd(double a, const double *x, double b, double *y, size_t n)
{
#pragma omp parallel for simd
    for (size_t i = 0; i < n; i++) y[i] = a * x[i] + b * y[i];
}

void orc_veccopy(double *dst, const double *src, size_t n)
{
#pragma omp parallel for simd
    for (size_t i = 0; i < n; i++) dst[i] = src[i];
}

/* ------------------------------------------------------------------------- */
/* spectral Poisson: src/poisson_fft.f90, src/backend/omp/kernels/spectral_processing.f90 */
/* ------------------------------------------------------------------------- */

/* src/poisson_fft.f90:833-882 wave_numbers.  k2/e/k hold equal re and im
 * parts in the reference, so only the real number is kept. */
void orc_wave_numbers(double *a, double *b, double *k, double *e, double *k2, int n, double L,
                      double d, int periodic, double c_a, double c_b, double c_alpha)
{
    double pi = 4 * atan(1.0);
    for (int i = 1; i <= n; i++) {
        if (periodic) {
            a[i - 1] = sin((i - 1) * pi / n);
            b[i - 1] = cos((i - 1) * pi / n);
        } else {
            a[i - 1] = sin((i - 1) * pi / 2 / n);
            b[i - 1] = cos((i - 1) * pi / 2 / n);
        }
    }
    if (periodic) {
        for (int i = 1; i <= n / 2 + 1; i++) {
            double w = 2 * pi * (i - 1) / n;
            double wp = c_a * 2 * d * sin(0.5 * w) + c_b * 2 * d * sin(1.5 * w);
            wp = wp / (1.0 + 2 * c_alpha * cos(w));
            k[i - 1] = n * wp / L;
            e[i - 1] = n * w / L;
            k2[i - 1] = (n * wp / L) * (n * wp / L);
        }
        for (int i = n / 2 + 2; i <= n; i++) {
            k[i - 1] = k[n - i + 1];
            e[i - 1] = e[n - i + 1];
            k2[i - 1] = k2[n - i + 1];
        }
    } else {
        for (int i = 1; i <= n; i++) {
            double w = pi * (i - 1) / n;
            double wp = c_a * 2 * d * sin(0.5 * w) + c_b * 2 * d * sin(1.5 * w);
            wp = wp / (1.0 + 2 * c_alpha * cos(w));
            k[i - 1] = n * wp / L;
            e[i - 1] = n * w / L;
            k2[i - 1] = (n * wp / L) * (n * wp / L);
        }
    }
}

/* src/poisson_fft.f90:781-818 waves_set, branch for the 000 / 010 cases.
 * itp = {a, b, c, d, alpha} of interpl_v2p per direction (x,y,z);
 * waves(i,j,k) real part only (= imaginary part in the reference). */
void orc_waves_set(double *waves, int nx_spec, int ny_spec, int nz_spec, const int *sp_st,
                   const double *exs, const double *eys, const double *ezs, const double *k2x,
                   const double *k2y, const double *k2z, const double *dxyz, const double *itp_x,
                   const double *itp_y, const double *itp_z)
{
    for (int k = 1; k <= nz_spec; k++)
        for (int j = 1; j <= ny_spec; j++)
            for (int i = 1; i <= nx_spec; i++) {
                int ix = i + sp_st[0], iy = j + sp_st[1], iz = k + sp_st[2];
                double rlexs = exs[ix - 1] * dxyz[0];
                double rleys = eys[iy - 1] * dxyz[1];
                double rlezs = ezs[iz - 1] * dxyz[2];
                double xtt = 2 * (itp_x[0] * cos(rlexs * 0.5) + itp_x[1] * cos(rlexs * 1.5) +
                                  itp_x[2] * cos(rlexs * 2.5) + itp_x[3] * cos(rlexs * 3.5));
                double ytt = 2 * (itp_y[0] * cos(rleys * 0.5) + itp_y[1] * cos(rleys * 1.5) +
                                  itp_y[2] * cos(rleys * 2.5) + itp_y[3] * cos(rleys * 3.5));
                double ztt = 2 * (itp_z[0] * cos(rlezs * 0.5) + itp_z[1] * cos(rlezs * 1.5) +
                                  itp_z[2] * cos(rlezs * 2.5) + itp_z[3] * cos(rlezs * 3.5));
                double xt1 = 1.0 + 2 * itp_x[4] * cos(rlexs);
                double yt1 = 1.0 + 2 * itp_y[4] * cos(rleys);
                double zt1 = 1.0 + 2 * itp_z[4] * cos(rlezs);
                double fx = (ytt / yt1) * (ztt / zt1);
                double fy = (xtt / xt1) * (ztt / zt1);
                double fz = (xtt / xt1) * (ytt / yt1);
                double xt2 = k2x[ix - 1] * (fx * fx);
                double yt2 = k2y[iy - 1] * (fy * fy);
                double zt2 = k2z[iz - 1] * (fz * fz);
                waves[((size_t)(k - 1) * ny_spec + (j - 1)) * nx_spec + (i - 1)] = xt2 + yt2 + zt2;
            }
}

/* src/backend/omp/kernels/spectral_processing.f90:7-106 process_spectral_000.
 * div is interleaved complex (re, im), index (i, j, k) i fastest;
 * waves_re / waves_im separately. */
void orc_process_spectral_000(double *div, const double *waves_re, const double *waves_im,
                              int nx_spec, int ny_spec, int nz_spec, int x_sp_st, int y_sp_st,
                              int z_sp_st, int nx, int ny, int nz, const double *ax,
                              const double *bx, const double *ay, const double *by,
                              const double *az, const double *bz)
{
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= nz_spec; k++)
        for (int j = 1; j <= ny_spec; j++)
            for (int i = 1; i <= nx_spec; i++) {
                size_t idx = ((size_t)(k - 1) * ny_spec + (j - 1)) * nx_spec + (i - 1);
                double div_r = div[2 * idx] / nx / ny / nz;
                double div_c = div[2 * idx + 1] / nx / ny / nz;
                int ix = i + x_sp_st, iy = j + y_sp_st, iz = k + z_sp_st;
                double tmp_r, tmp_c;
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * bz[iz - 1] + tmp_c * az[iz - 1];
                div_c = tmp_c * bz[iz - 1] - tmp_r * az[iz - 1];
                if (iz > nz / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * by[iy - 1] + tmp_c * ay[iy - 1];
                div_c = tmp_c * by[iy - 1] - tmp_r * ay[iy - 1];
                if (iy > ny / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * bx[ix - 1] + tmp_c * ax[ix - 1];
                div_c = tmp_c * bx[ix - 1] - tmp_r * ax[ix - 1];
                tmp_r = waves_re[idx]; tmp_c = waves_im[idx];
                if (tmp_r < 1.e-16 || tmp_c < 1.e-16) {
                    div_r = 0.0; div_c = 0.0;
                } else {
                    div_r = -div_r / tmp_r;
                    div_c = -div_c / tmp_c;
                }
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * bz[iz - 1] - tmp_c * az[iz - 1];
                div_c = -tmp_c * bz[iz - 1] - tmp_r * az[iz - 1];
                if (iz > nz / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * by[iy - 1] + tmp_c * ay[iy - 1];
                div_c = tmp_c * by[iy - 1] - tmp_r * ay[iy - 1];
                if (iy > ny / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * bx[ix - 1] + tmp_c * ax[ix - 1];
                div_c = -tmp_c * bx[ix - 1] + tmp_r * ax[ix - 1];
                div[2 * idx] = div_r;
                div[2 * idx + 1] = div_c;
            }
}

int orc_sz(void) { return SZ; }

/* ---------------------------------------------------------------------------
 * Non-periodic y (010) spectral post-processing.
 *   src/backend/omp/kernels/spectral_processing.f90:108-283  process_spectral_010
 *   src/backend/cuda/kernels/spectral_processing.f90:385-702 process_spectral_010_fw /
 *       _poisson (pentadiagonal, stretched y) / _bw   [CUDA Fortran: cannot be built here]
 * The OMP kernel is fw -> divide by waves -> bw; the CUDA fw / bw kernels carry the
 * same arithmetic as the OMP kernel's first two / last two loops, so pinning
 * orc_process_spectral_010 against the reference's OMP kernel (tests/golden/ref_c010*.npz)
 * pins fw and bw too.  The pentadiagonal solve is pinned by the stretching matrices
 * (dumped from the reference's base_init) and by div(grad(p)) = f, the acceptance
 * check of tests/verification/test_poisson_bc.f90.
 * div: interleaved complex, (i, j, k) with i fastest.
 * ------------------------------------------------------------------------- */
#define C_IDX(i, j, k) (((size_t)((k) - 1) * ny_spec + ((j) - 1)) * nx_spec + ((i) - 1))

void orc_spectral_010_fw(double *div, int nx_spec, int ny_spec, int nz_spec, int x_sp_st, int y_sp_st,
                         int z_sp_st, int nx, int ny, int nz, const double *ax, const double *bx,
                         const double *ay, const double *by, const double *az, const double *bz)
{
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= nz_spec; k++)
        for (int j = 1; j <= ny_spec; j++)
            for (int i = 1; i <= nx_spec; i++) {
                size_t id = C_IDX(i, j, k);
                int ix = i + x_sp_st, iz = k + z_sp_st;
                double div_r = div[2 * id] / nx / ny / nz, div_c = div[2 * id + 1] / nx / ny / nz;
                double tmp_r = div_r, tmp_c = div_c;
                div_r = tmp_r * bz[iz - 1] + tmp_c * az[iz - 1];
                div_c = tmp_c * bz[iz - 1] - tmp_r * az[iz - 1];
                if (iz > nz / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * bx[ix - 1] + tmp_c * ax[ix - 1];
                div_c = tmp_c * bx[ix - 1] - tmp_r * ax[ix - 1];
                if (ix > nx / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                div[2 * id] = div_r; div[2 * id + 1] = div_c;
            }
    /* paired split j <-> ny_spec - j + 2 (for even ny_spec the middle row pairs with itself:
     * both stores hit it, the second one stays, as in the reference) */
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= nz_spec; k++)
        for (int j = 2; j <= ny_spec / 2 + 1; j++)
            for (int i = 1; i <= nx_spec; i++) {
                int jr = ny_spec - j + 2, iy = j + y_sp_st, iy_r = jr + y_sp_st;
                size_t il = C_IDX(i, j, k), ir = C_IDX(i, jr, k);
                double l_r = div[2 * il], l_c = div[2 * il + 1], r_r = div[2 * ir], r_c = div[2 * ir + 1];
                double a = ay[iy - 1], b = by[iy - 1], a2 = ay[iy_r - 1], b2 = by[iy_r - 1];
                div[2 * il] = 0.5 * (l_r * b + l_c * a + r_r * b - r_c * a);
                div[2 * il + 1] = 0.5 * (-l_r * a + l_c * b + r_r * a + r_c * b);
                div[2 * ir] = 0.5 * (r_r * b2 + r_c * a2 + l_r * b2 - l_c * a2);
                div[2 * ir + 1] = 0.5 * (-r_r * a2 + r_c * b2 + l_r * a2 + l_c * b2);
            }
}

void orc_spectral_010_bw(double *div, int nx_spec, int ny_spec, int nz_spec, int x_sp_st, int y_sp_st,
                         int z_sp_st, int nx, int ny, int nz, const double *ax, const double *bx,
                         const double *ay, const double *by, const double *az, const double *bz)
{
    (void)ny;
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= nz_spec; k++)
        for (int j = 2; j <= ny_spec / 2 + 1; j++)
            for (int i = 1; i <= nx_spec; i++) {
                int jr = ny_spec - j + 2, iy = j + y_sp_st, iy_r = jr + y_sp_st;
                size_t il = C_IDX(i, j, k), ir = C_IDX(i, jr, k);
                double l_r = div[2 * il], l_c = div[2 * il + 1], r_r = div[2 * ir], r_c = div[2 * ir + 1];
                double a = ay[iy - 1], b = by[iy - 1], a2 = ay[iy_r - 1], b2 = by[iy_r - 1];
                div[2 * il] = l_r * b - l_c * a + r_r * a + r_c * b;
                div[2 * il + 1] = l_r * a + l_c * b - r_r * b + r_c * a;
                div[2 * ir] = r_r * b2 - r_c * a2 + l_r * a2 + l_c * b2;
                div[2 * ir + 1] = r_r * a2 + r_c * b2 - l_r * b2 + l_c * a2;
            }
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= nz_spec; k++)
        for (int j = 1; j <= ny_spec; j++)
            for (int i = 1; i <= nx_spec; i++) {
                size_t id = C_IDX(i, j, k);
                int ix = i + x_sp_st, iz = k + z_sp_st;
                double div_r = div[2 * id], div_c = div[2 * id + 1];
                double tmp_r = div_r, tmp_c = div_c;
                div_r = tmp_r * bz[iz - 1] - tmp_c * az[iz - 1];
                div_c = tmp_c * bz[iz - 1] + tmp_r * az[iz - 1];
                if (iz > nz / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                tmp_r = div_r; tmp_c = div_c;
                div_r = tmp_r * bx[ix - 1] - tmp_c * ax[ix - 1];
                div_c = tmp_c * bx[ix - 1] + tmp_r * ax[ix - 1];
                if (ix > nx / 2 + 1) { div_r = -div_r; div_c = -div_c; }
                div[2 * id] = div_r; div[2 * id + 1] = div_c;
            }
}

/* uniform y: -div / waves per part, zero where |waves| < 1e-16, and the (nx/2+1, *, nz/2+1) line */
void orc_spectral_010_divide(double *div, const double *waves_re, const double *waves_im, int nx_spec,
                             int ny_spec, int nz_spec, int nx, int nz)
{
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= nz_spec; k++)
        for (int j = 1; j <= ny_spec; j++)
            for (int i = 1; i <= nx_spec; i++) {
                size_t id = C_IDX(i, j, k);
                double div_r = div[2 * id], div_c = div[2 * id + 1];
                double tmp_r = waves_re[id], tmp_c = waves_im[id];
                div_r = fabs(tmp_r) < 1.e-16 ? 0.0 : -div_r / tmp_r;
                div_c = fabs(tmp_c) < 1.e-16 ? 0.0 : -div_c / tmp_c;
                if (i == nx / 2 + 1 && k == nz / 2 + 1) { div_r = 0.0; div_c = 0.0; }
                div[2 * id] = div_r; div[2 * id + 1] = div_c;
            }
}

void orc_process_spectral_010(double *div, const double *waves_re, const double *waves_im, int nx_spec,
                              int ny_spec, int nz_spec, int x_sp_st, int y_sp_st, int z_sp_st, int nx,
                              int ny, int nz, const double *ax, const double *bx, const double *ay,
                              const double *by, const double *az, const double *bz)
{
    orc_spectral_010_fw(div, nx_spec, ny_spec, nz_spec, x_sp_st, y_sp_st, z_sp_st, nx, ny, nz, ax, bx, ay,
                        by, az, bz);
    orc_spectral_010_divide(div, waves_re, waves_im, nx_spec, ny_spec, nz_spec, nx, nz);
    orc_spectral_010_bw(div, nx_spec, ny_spec, nz_spec, x_sp_st, y_sp_st, z_sp_st, nx, ny, nz, ax, bx, ay,
                        by, az, bz);
}

/* process_spectral_010_poisson (CUDA kernel :462-617): pentadiagonal solve along y per (i, k),
 * rows j = 1..n mapped to spectral rows jm = inc*j + off - inc/2 (odd rows: inc 2, off 0; even
 * rows: inc 2, off 1; all rows: inc 1, off 0).  a_re / a_im: [5][nz_spec][n][nx_spec] (diagonal
 * index slowest), MODIFIED in place like the reference's device copies. */
void orc_spectral_010_penta(double *div, double *a_re, double *a_im, int off, int inc, int nx_spec,
                            int ny_spec, int nz_spec, int n, int nx, int nz)
{
    const double eps = 1.e-16;
    const size_t dstride = (size_t)nz_spec * n * nx_spec;
#define A_(M, i, j, k, d) M[((d) - 1) * dstride + ((size_t)((k) - 1) * n + ((j) - 1)) * nx_spec + ((i) - 1)]
#define RE(i, j, k) div[2 * C_IDX(i, j, k)]
#define IM(i, j, k) div[2 * C_IDX(i, j, k) + 1]
#pragma omp parallel for collapse(2)
    for (int k = 1; k <= nz_spec; k++)
        for (int i = 1; i <= nx_spec; i++) {
            double tmp_r, tmp_c, div_r, div_c;
            for (int j = 1; j <= n - 2; j++) {
                int jm = inc * j + off - inc / 2;
                tmp_r = fabs(A_(a_re, i, j, k, 3)) > eps ? A_(a_re, i, j + 1, k, 2) / A_(a_re, i, j, k, 3) : 0.0;
                tmp_c = fabs(A_(a_im, i, j, k, 3)) > eps ? A_(a_im, i, j + 1, k, 2) / A_(a_im, i, j, k, 3) : 0.0;
                RE(i, jm + inc, k) = RE(i, jm + inc, k) - tmp_r * RE(i, jm, k);
                IM(i, jm + inc, k) = IM(i, jm + inc, k) - tmp_c * IM(i, jm, k);
                A_(a_re, i, j + 1, k, 3) -= tmp_r * A_(a_re, i, j, k, 4);
                A_(a_im, i, j + 1, k, 3) -= tmp_c * A_(a_im, i, j, k, 4);
                A_(a_re, i, j + 1, k, 4) -= tmp_r * A_(a_re, i, j, k, 5);
                A_(a_im, i, j + 1, k, 4) -= tmp_c * A_(a_im, i, j, k, 5);
                tmp_r = fabs(A_(a_re, i, j, k, 3)) > eps ? A_(a_re, i, j + 2, k, 1) / A_(a_re, i, j, k, 3) : 0.0;
                tmp_c = fabs(A_(a_im, i, j, k, 3)) > eps ? A_(a_im, i, j + 2, k, 1) / A_(a_im, i, j, k, 3) : 0.0;
                RE(i, jm + 2 * inc, k) = RE(i, jm + 2 * inc, k) - tmp_r * RE(i, jm, k);
                IM(i, jm + 2 * inc, k) = IM(i, jm + 2 * inc, k) - tmp_c * IM(i, jm, k);
                A_(a_re, i, j + 2, k, 2) -= tmp_r * A_(a_re, i, j, k, 4);
                A_(a_im, i, j + 2, k, 2) -= tmp_c * A_(a_im, i, j, k, 4);
                A_(a_re, i, j + 2, k, 3) -= tmp_r * A_(a_re, i, j, k, 5);
                A_(a_im, i, j + 2, k, 3) -= tmp_c * A_(a_im, i, j, k, 5);
            }
            /* last two rows */
            tmp_r = fabs(A_(a_re, i, n - 1, k, 3)) > eps ? A_(a_re, i, n, k, 2) / A_(a_re, i, n - 1, k, 3) : 0.0;
            tmp_c = fabs(A_(a_im, i, n - 1, k, 3)) > eps ? A_(a_im, i, n, k, 2) / A_(a_im, i, n - 1, k, 3) : 0.0;
            div_r = A_(a_re, i, n, k, 3) - tmp_r * A_(a_re, i, n - 1, k, 4);
            div_c = A_(a_im, i, n, k, 3) - tmp_c * A_(a_im, i, n - 1, k, 4);
            int nm = inc * n + off - inc / 2;
            if (fabs(div_r) > eps) {
                tmp_r = tmp_r / div_r;
                div_r = RE(i, nm, k) / div_r - tmp_r * RE(i, nm - inc, k);
            } else { tmp_r = 0.0; div_r = 0.0; }
            if (fabs(div_c) > eps) {
                tmp_c = tmp_c / div_c;
                div_c = IM(i, nm, k) / div_c - tmp_c * IM(i, nm - inc, k);
            } else { tmp_c = 0.0; div_c = 0.0; }
            RE(i, nm, k) = div_r; IM(i, nm, k) = div_c;
            tmp_r = fabs(A_(a_re, i, n - 1, k, 3)) > eps ? 1.0 / A_(a_re, i, n - 1, k, 3) : 0.0;
            tmp_c = fabs(A_(a_im, i, n - 1, k, 3)) > eps ? 1.0 / A_(a_im, i, n - 1, k, 3) : 0.0;
            div_r = A_(a_re, i, n - 1, k, 4) * tmp_r;
            div_c = A_(a_im, i, n - 1, k, 4) * tmp_c;
            RE(i, nm - inc, k) = RE(i, nm - inc, k) * tmp_r - RE(i, nm, k) * div_r;
            IM(i, nm - inc, k) = IM(i, nm - inc, k) * tmp_c - IM(i, nm, k) * div_c;
            const int zero_mode = (i == nx / 2 + 1 && k == nz / 2 + 1);
            if (zero_mode) {
                RE(i, nm, k) = 0.0; IM(i, nm, k) = 0.0;
                RE(i, nm - inc, k) = 0.0; IM(i, nm - inc, k) = 0.0;
            }
            for (int j = n - 2; j >= 1; j--) {
                int jm = inc * j + off - inc / 2;
                tmp_r = fabs(A_(a_re, i, j, k, 3)) > eps ? 1.0 / A_(a_re, i, j, k, 3) : 0.0;
                tmp_c = fabs(A_(a_im, i, j, k, 3)) > eps ? 1.0 / A_(a_im, i, j, k, 3) : 0.0;
                RE(i, jm, k) = tmp_r * (RE(i, jm, k) - A_(a_re, i, j, k, 4) * RE(i, jm + inc, k) -
                                        A_(a_re, i, j, k, 5) * RE(i, jm + 2 * inc, k));
                IM(i, jm, k) = tmp_c * (IM(i, jm, k) - A_(a_im, i, j, k, 4) * IM(i, jm + inc, k) -
                                        A_(a_im, i, j, k, 5) * IM(i, jm + 2 * inc, k));
                if (zero_mode) { RE(i, jm, k) = 0.0; IM(i, jm, k) = 0.0; }
            }
        }
#undef A_
#undef RE
#undef IM
}
#undef C_IDX
