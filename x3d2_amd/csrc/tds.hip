// DistD2 compact-scheme kernels: tds_solve and the fused transport-equation
// component, one pencil per lane, two sweeps.
//
// Reference arithmetic restated for CDNA4 (paths under /root/reference):
//   src/backend/omp/kernels/distributed.f90:11-168  der_univ_dist
//   src/backend/omp/kernels/distributed.f90:170-229 der_univ_subs
//   src/backend/omp/kernels/distributed.f90:231-337 der_univ_fused_subs
//   src/backend/omp/exec_dist.f90:16-65, 67-186     the two-phase drivers
//
// Structure (differs from the reference's three sweeps on purpose):
//   forward sweep  : RHS stencil + forward elimination d_j, written once;
//                    accumulates S = sum_k W_k d_k so that the first unknown of
//                    the backward chain, du_1, is known when the sweep ends
//                    (W_k = prod_{l<k} (-dist_bw_l)): du_1 and d_n are what the
//                    2x2 reduced systems need, so they can be exchanged NOW.
//   backward sweep : back-substitution, reduced-system substitution, stretching
//                    and (transeq) the skew-symmetric combination fused in one
//                    reverse pass that writes the final result.
// Every table is indexed by the wave-uniform row j -> scalar loads; lanes only
// differ in their pencil base address.  Loads are coalesced whenever lanes run
// over x (dir Y/Z); dir X goes through LDS tile transposes (xdir.hip).
#include "common.h"
#include "zfft_tile.h"

// ------------------------------------------------------------------ tables
extern "C" int x3d_tdsops_create(x3d_backend *b, x3d_tdsops **out, int n_tds, int n_rhs, int move,
                                 int periodic, const real_t *coeffs, const real_t *coeffs_s,
                                 const real_t *coeffs_e, const real_t *dist_fw, const real_t *dist_bw,
                                 const real_t *dist_sa, const real_t *dist_sc, const real_t *dist_af,
                                 const real_t *stretch, const real_t *stretch_correct)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && coeffs && coeffs_s && coeffs_e && dist_fw && dist_bw && dist_sa && dist_sc &&
                    dist_af && stretch && stretch_correct,
                "x3d_tdsops_create: null argument");
    X3D_REQUIRE(n_rhs == n_tds || n_rhs == n_tds + 1, "x3d_tdsops_create: n_rhs must be n_tds or n_tds+1");
    X3D_REQUIRE(n_tds >= 8, "x3d_tdsops_create: n_tds=%d too small (the 4+4 boundary rows need n>=8)", n_tds);
    const int n = n_tds, nr = n_rhs, L = nr + 2;  // tables are 1-based, one spare entry
    // layout: F A W Bw Sa Sc St Stc PF QB (10 tables of L) + 81 stencil coefficients
    std::vector<real_t> h((size_t)10 * L + 81, 0.0);
    real_t *F = &h[0], *A = &h[L], *W = &h[2 * L], *Bw = &h[3 * L], *Sa = &h[4 * L], *Sc = &h[5 * L],
           *St = &h[6 * L], *Stc = &h[7 * L], *PF = &h[8 * L], *QB = &h[9 * L], *Cs = &h[10 * L];
    for (int j = 1; j <= nr; j++) {
        const bool real_row = j <= n;  // row n_tds+1 of a v2p operator is junk in the reference too
        if (j <= 2) {                  // distributed.f90:45,56: du = rhs*faf(j)
            F[j] = dist_af[j - 1];
            A[j] = 0.0;
        } else {
            F[j] = real_row ? dist_fw[j - 1] : 0.0;
            // distributed.f90:83: one alpha = faf(5) for the bulk rows 5..n_rhs-4
            A[j] = (j >= 5 && j <= nr - 4) ? dist_af[4] : (real_row ? dist_af[j - 1] : 0.0);
        }
        if (real_row) {
            Sa[j] = dist_sa[j - 1];
            Sc[j] = dist_sc[j - 1];
            St[j] = stretch[j - 1];
            Stc[j] = stretch_correct[j - 1];
        }
        // dist_bw is read for rows 1..n-2 only (distributed.f90:154-163)
        Bw[j] = (j <= n - 2) ? dist_bw[j - 1] : 0.0;
    }
    // du_2 = sum_{k=2}^{n-1} W_k d_k from du_j = d_j - bw_j du_{j+1}, j = n-2..2,
    // with du_{n-1} = d_{n-1} untouched by the backward pass
    W[2] = 1.0;
    for (int k = 2; k <= n - 2; k++) W[k + 1] = W[k] * (-dist_bw[k - 1]);
    // chunk-parallel form (onchip.hip): e_j = ehat_j + PF_j * e_{s-1}, X_j = Xhat_j + QB_j * X_{t+1}
    // for a chunk [s, t] of `chunk` rows, with g_l = -F_l A_l and h_l = -bw_l (2 <= l <= n-2, else 0)
    const int chunk = nr <= 512 ? 32 : 64;
    for (int s0 = 1; s0 <= nr; s0 += chunk) {
        const int t0 = s0 + chunk - 1 < nr ? s0 + chunk - 1 : nr;
        real_t pr = 1.0;
        for (int j = s0; j <= t0; j++) { pr *= -F[j] * A[j]; PF[j] = pr; }
        pr = 1.0;
        for (int j = t0; j >= s0; j--) {
            const real_t hj = (j >= 2 && j <= n - 2) ? -dist_bw[j - 1] : 0.0;
            pr *= hj;
            QB[j] = pr;
        }
    }
    // the same for 16-row chunks (onchip.hip, 256-row pencils)
    std::vector<real_t> PF16(L, 0.0), QB16(L, 0.0);
    for (int s0 = 1; s0 <= nr; s0 += 16) {
        const int t0 = s0 + 15 < nr ? s0 + 15 : nr;
        real_t pr = 1.0;
        for (int j = s0; j <= t0; j++) { pr *= -F[j] * A[j]; PF16[j] = pr; }
        pr = 1.0;
        for (int j = t0; j >= s0; j--) {
            pr *= (j >= 2 && j <= n - 2) ? -dist_bw[j - 1] : 0.0;
            QB16[j] = pr;
        }
    }
    memcpy(Cs, coeffs_s, sizeof(real_t) * 36);
    memcpy(Cs + 36, coeffs_e, sizeof(real_t) * 36);
    memcpy(Cs + 72, coeffs, sizeof(real_t) * 9);

    x3d_tdsops *t = new x3d_tdsops();
    for (int m = 0; m < 9; m++) t->coeffs[m] = coeffs[m];
    t->b = b; t->n_tds = n; t->n_rhs = nr; t->move = move; t->periodic = periodic;
    // device image: interleaved row records (common.h) so that one wide scalar
    // load serves a row: RF[4j..] = F A W PF ; RB[8j..] = Bw Sa Sc St Stc QB PF16 QB16 ; then Cs
    std::vector<real_t> img((size_t)12 * L + 81, 0.0);
    for (int j = 0; j < L; j++) {
        real_t *rf = &img[(size_t)4 * j], *rb = &img[(size_t)4 * L + (size_t)8 * j];
        rf[0] = F[j]; rf[1] = A[j]; rf[2] = W[j]; rf[3] = PF[j];
        rb[0] = Bw[j]; rb[1] = Sa[j]; rb[2] = Sc[j]; rb[3] = St[j]; rb[4] = Stc[j]; rb[5] = QB[j];
        rb[6] = PF16[j]; rb[7] = QB16[j];
    }
    memcpy(&img[(size_t)12 * L], Cs, sizeof(real_t) * 81);
    // lane tables for the wave-per-pencil x kernels (xscan.hip): lane l owns rows l*Q+1..(l+1)*Q
    // (6: 257 .. 384 rows, e.g. the channel case's 257 wall-normal vertices -- 43 of the 64 lanes busy instead of 33)
    const int Q = nr <= 256 ? 4 : (nr <= 384 ? 6 : (nr <= 512 ? 8 : (nr <= 1024 ? 16 : 0)));
    const size_t tl_off = img.size();
    auto build_tl = [&](const int Q, std::vector<real_t> &out) {  // the [9 Q + 12][64] table for Q rows per lane
        const int NE = 9 * Q + 12;
        out.assign((size_t)NE * 64, 0.0);
        real_t *tl = out.data();
        auto E = [&](int e, int l) -> real_t & { return tl[(size_t)e * 64 + l]; };
        real_t G[64], HL[64];
        for (int l = 0; l < 64; l++) {
            real_t pg = 1.0;
            for (int q = 0; q < Q; q++) {
                const int j = l * Q + q + 1;
                const bool in = j <= nr, real = j <= n;
                const real_t f = in ? F[j] : 0.0, a = in ? A[j] : 0.0;
                E(0 * Q + q, l) = f;
                E(1 * Q + q, l) = a;
                pg *= -f * a;
                E(2 * Q + q, l) = pg;
                E(3 * Q + q, l) = (j >= 1 && j <= n - 2) ? -dist_bw[j - 1] : 0.0;  // incl. row 1: X_1 = e_1 - bw_1 X_2
                E(5 * Q + q, l) = real ? Sa[j] : 0.0;
                E(6 * Q + q, l) = real ? Sc[j] : 0.0;
                E(7 * Q + q, l) = real ? St[j] : 0.0;
                E(8 * Q + 12 + q, l) = real ? Stc[j] : 0.0;  // (last block: kernels that do not need it stage 8 Q + 12 entries)
            }
            G[l] = pg;
            real_t ph = 1.0;
            for (int q = Q - 1; q >= 0; q--) { ph *= E(3 * Q + q, l); E(4 * Q + q, l) = ph; }
            HL[l] = ph;
        }
        real_t mf[64], mb[64];
        for (int l = 0; l < 64; l++) { mf[l] = G[l]; mb[l] = HL[l]; }
        // data-independent multipliers of the wave scans (xscan.hip, scan_solve): Kogge-Stone steps
        // 1, 2, 4, 8 inside each row of 16 lanes (DPP row shifts), then two steps across the rows
        for (int k = 0; k < 4; k++) {
            const int d = 1 << k;
            real_t nf[64], nb[64];
            for (int l = 0; l < 64; l++) {
                E(8 * Q + k, l) = mf[l];
                E(8 * Q + 6 + k, l) = mb[l];
                nf[l] = l >= d ? mf[l] * mf[l - d] : mf[l];
                nb[l] = l + d < 64 ? mb[l] * mb[l + d] : mb[l];
            }
            for (int l = 0; l < 64; l++) { mf[l] = nf[l]; mb[l] = nb[l]; }
        }
        for (int l = 0; l < 64; l++) {
            real_t c15 = 1.0, c31 = 1.0, d16 = 1.0, d32 = 1.0;
            for (int i = l & ~15; i <= l; i++) c15 *= G[i];   // row start .. l: carries lane 15 / 47 into rows 1 / 3
            for (int i = 32; i <= l; i++) c31 *= G[i];        // 32 .. l: carries lane 31 into rows 2, 3
            for (int i = l; i <= (l | 15); i++) d16 *= HL[i]; // l .. row end: carries lane 16 / 48 into rows 0 / 2
            for (int i = l; i <= 31; i++) d32 *= HL[i];       // l .. 31: carries lane 32 into rows 0, 1
            const int row = l >> 4;
            E(8 * Q + 4, l) = (row == 1 || row == 3) ? c15 : 0.0;
            E(8 * Q + 5, l) = row >= 2 ? c31 : 0.0;
            E(8 * Q + 6 + 4, l) = (row == 0 || row == 2) ? d16 : 0.0;
            E(8 * Q + 6 + 5, l) = row < 2 ? d32 : 0.0;
        }
    };
    if (Q) {
        std::vector<real_t> tlv;
        build_tl(Q, tlv);
        img.insert(img.end(), tlv.begin(), tlv.end());
    }
    // compressed form of the row entries (xscan_core.h, LTC_*): lanes 0..7 | one value for lanes 8..55 | lanes
    // 56..63 -- only where those middle lanes really are bitwise equal (periodic-type operators on a uniform grid)
    size_t tlc_off = 0;
    if (Q == 16 || Q == 8) {  // (Q = 8, round 6: the 8-pencil z-transforming pairs, zfpair8.hip)
        const real_t *tl = &img[tl_off];
        bool ok = true;
        auto row_entry = [&](int k) { return k < 8 * Q ? k : 8 * Q + 12 + (k - 8 * Q); };  // k = 0 .. 9Q-1
        // dist_sa / dist_sc never become constant, they decay (by 0.15 - 0.38 per row): in the middle lanes (rows
        // 129 .. 896) they are below 2^-60 and stored as 0, the truncation the HALO strip corrections use too
        auto decaying = [&](int k) { return k >= 5 * Q && k < 7 * Q; };
        for (int k = 0; k < 9 * Q && ok; k++)
            for (int l = 8; l <= 55 && ok; l++) {
                const real_t *v = &tl[(size_t)row_entry(k) * 64];
                ok = decaying(k) ? fabs(v[l]) < 8.673617379884035e-19 : memcmp(&v[l], &v[8], sizeof(real_t)) == 0;
            }
        if (ok) {
            std::vector<real_t> c((size_t)9 * Q * 17 + 12 * 64);
            for (int k = 0; k < 9 * Q; k++)
                for (int m = 0; m < 17; m++)
                    c[(size_t)k * 17 + m] = (m == 8 && decaying(k)) ? 0.0 : tl[(size_t)row_entry(k) * 64 + (m <= 8 ? m : m + 47)];
            for (int k = 0; k < 12; k++)
                for (int l = 0; l < 64; l++) c[(size_t)9 * Q * 17 + k * 64 + l] = tl[(size_t)(8 * Q + k) * 64 + l];
            tlc_off = img.size();
            img.insert(img.end(), c.begin(), c.end());
        }
    }
    // 257 .. 320 rows: a second table with 5 rows per lane for the tile kernels of ygen.hip (52 of 64 lanes busy
    // instead of 43 with 6; the x kernels keep the even Q: they move their rows as aligned 16-byte pairs)
    size_t tl5_off = 0;
    if (nr > 256 && nr <= 320) {
        std::vector<real_t> tlv;
        build_tl(5, tlv);
        tl5_off = img.size();
        img.insert(img.end(), tlv.begin(), tlv.end());
    }
    // DIRECT tables (round 6, ygen.hip thomas_solve): a non-periodic operator on ONE rank is a plain tridiagonal system
    // a_j x_{j-1} + b_j x_j + c_j x_{j+1} = r_j with a_1 = c_n = 0 -- the reference's distributed form (two interface
    // rows, spike vectors dist_sa / dist_sc, 2 x 2 closure; distributed.f90:170-229) solves exactly that system when
    // both neighbours are absent (rs_s = rs_e = 1, sa(1) = sc(n) = 0).  The tile kernels then run the Thomas
    // recurrences themselves, parallel over lanes by the same scans, WITHOUT the closure phase (no SA / SC reads, no
    // du_1 / X_n broadcasts, no row selects): the matrix is recovered from the preprocessed arrays (the inverse of
    // preprocess_dist, src/tdsops.f90:874-931), factored here, and the result is checked on the host against the
    // reference's own sweeps before the tables are offered.
    size_t td5_off = 0, td8h_off = 0;
    {
        bool direct = !periodic && dist_sa[0] == 0.0 && dist_sc[n - 1] == 0.0 && n >= 8;
        std::vector<real_t> ma(n + 2, 0.0), mb(n + 2, 0.0), mc(n + 2, 0.0), TF(L, 0.0), TFA(L, 0.0), TH(L, 0.0);
        if (direct) {
            auto scp = [&](int i) { return i <= n - 3 ? dist_bw[i] : dist_sc[i]; };  // (0-based) c_i after its forward step
            for (int i = 0; i < 2; i++) {
                mb[i + 1] = 1.0 / dist_af[i];
                mc[i + 1] = scp(i) * mb[i + 1];
            }
            ma[2] = (dist_sa[1] + scp(1) * dist_sa[2]) * mb[2];  // (the backward step of preprocess_dist undone for row 2)
            for (int i = 2; i < n; i++) {
                ma[i + 1] = dist_af[i];
                mb[i + 1] = 1.0 / dist_fw[i] + ma[i + 1] * scp(i - 1);
                mc[i + 1] = scp(i) / dist_fw[i];
            }
            mc[n] = 0.0;
            real_t g = 0.0;
            for (int j = 1; j <= n; j++) {
                const real_t f = 1.0 / (mb[j] - ma[j] * g);
                TF[j] = f;
                TFA[j] = -f * ma[j];
                g = mc[j] * f;
                TH[j] = j < n ? -g : 0.0;
                if (!std::isfinite(f)) direct = false;
            }
        }
        if (direct) {  // host check: the reference's sweeps against the plain Thomas sweeps, two right-hand sides
            std::vector<real_t> r(n + 2), d(n + 2), x(n + 2), e(n + 2), y(n + 2);
            unsigned long long s = 0x9E3779B97F4A7C15ull;
            for (int rep = 0; rep < 2 && direct; rep++) {
                for (int j = 1; j <= n; j++) {
                    s = s * 6364136223846793005ull + 1442695040888963407ull;
                    r[j] = (real_t)((double)(s >> 11) / 9007199254740992.0 - 0.5);
                }
                d[1] = r[1] * F[1];
                d[2] = r[2] * F[2];
                for (int j = 3; j <= n; j++) d[j] = F[j] * (r[j] - A[j] * d[j - 1]);
                x = d;
                for (int j = n - 2; j >= 2; j--) x[j] = d[j] - Bw[j] * x[j + 1];
                x[1] = dist_fw[0] * (d[1] - Bw[1] * x[2]);
                const real_t du_s = x[1], du_e = x[n];
                real_t xmax = 0.0, emax = 0.0;
                e[0] = 0.0;
                for (int j = 1; j <= n; j++) e[j] = TF[j] * r[j] + TFA[j] * e[j - 1];
                y[n + 1] = 0.0;
                for (int j = n; j >= 1; j--) y[j] = e[j] + TH[j] * y[j + 1];
                for (int j = 1; j <= n; j++) {
                    const real_t ref = j == 1 ? du_s : (j == n ? du_e : x[j] - Sa[j] * du_s - Sc[j] * du_e);
                    xmax = fmax(xmax, fabs(ref));
                    emax = fmax(emax, fabs(ref - y[j]));
                }
                if (!(emax <= (sizeof(real_t) == 8 ? 1e-13 : 1e-5) * xmax)) direct = false;
            }
        }
        t->direct = direct ? 1 : 0;
        // [7 Q + 16][64]: F FA PF H QB ST | 6 + 6 scan multipliers | XR (4) | STC; LP lanes per pencil (32: two pencils per
        // wave, the tables of lanes 32..63 repeat lanes 0..31); XR: row LP Q + 1 exists and is carried by the pencil's last
        // lane (257 rows = 32 x 8 + 1): its F, FA, ST, STC, every lane the same
        auto build_direct = [&](const int Q, const int LP, std::vector<real_t> &out) {
            const bool XR = n == LP * Q + 1;
            const int NE = 7 * Q + 16;
            out.assign((size_t)NE * 64, 0.0);
            real_t *tl = out.data();
            auto E = [&](int e, int l) -> real_t & { return tl[(size_t)e * 64 + l]; };
            real_t G[64], HL[64];
            for (int l = 0; l < LP; l++) {
                real_t pg = 1.0;
                for (int q = 0; q < Q; q++) {
                    const int j = l * Q + q + 1;
                    const bool real = j <= n;
                    E(0 * Q + q, l) = real ? TF[j] : 0.0;
                    E(1 * Q + q, l) = real ? TFA[j] : 0.0;
                    pg *= real ? TFA[j] : 0.0;
                    E(2 * Q + q, l) = pg;
                    E(3 * Q + q, l) = real ? TH[j] : 0.0;
                    E(5 * Q + q, l) = real ? St[j] : 0.0;
                    E(6 * Q + 16 + q, l) = real ? Stc[j] : 0.0;
                }
                G[l] = pg;
                real_t ph = 1.0;
                // (XR: the last lane starts its backward sweep from the extra row's value, not from a carry)
                for (int q = Q - 1; q >= 0; q--) { ph *= E(3 * Q + q, l); E(4 * Q + q, l) = (XR && l == LP - 1) ? 0.0 : ph; }
                HL[l] = (XR && l == LP - 1) ? 0.0 : ph;
            }
            real_t mf[64], mbk[64];
            for (int l = 0; l < LP; l++) { mf[l] = G[l]; mbk[l] = HL[l]; }
            for (int k = 0; k < 4; k++) {
                const int d = 1 << k;
                real_t nf[64], nb[64];
                for (int l = 0; l < LP; l++) {
                    E(6 * Q + k, l) = mf[l];
                    E(6 * Q + 6 + k, l) = mbk[l];
                    // (DPP row shifts stay inside a row of 16 lanes)
                    nf[l] = (l & 15) >= d ? mf[l] * mf[l - d] : mf[l];
                    nb[l] = (l & 15) + d < 16 ? mbk[l] * mbk[l + d] : mbk[l];
                }
                for (int l = 0; l < LP; l++) { mf[l] = nf[l]; mbk[l] = nb[l]; }
            }
            for (int l = 0; l < LP; l++) {
                real_t c15 = 1.0, c31 = 1.0, d16 = 1.0, d32 = 1.0;
                for (int i = l & ~15; i <= l; i++) c15 *= G[i];
                for (int i = 32; i <= l; i++) c31 *= G[i];
                for (int i = l; i <= (l | 15); i++) d16 *= HL[i];
                for (int i = l; i <= 31; i++) d32 *= HL[i];
                const int row = l >> 4;
                E(6 * Q + 4, l) = (row == 1 || row == 3) ? c15 : 0.0;
                E(6 * Q + 5, l) = (LP == 64 && row >= 2) ? c31 : 0.0;
                E(6 * Q + 6 + 4, l) = (row == 0 || row == 2) ? d16 : 0.0;
                E(6 * Q + 6 + 5, l) = (LP == 64 && row < 2) ? d32 : 0.0;
            }
            if (XR) {
                for (int l = 0; l < LP; l++) {
                    E(6 * Q + 12 + 0, l) = TF[n];
                    E(6 * Q + 12 + 1, l) = TFA[n];
                    E(6 * Q + 12 + 2, l) = St[n];
                    E(6 * Q + 12 + 3, l) = Stc[n];
                }
            }
            if (LP == 32)
                for (int e = 0; e < NE; e++)
                    for (int l = 0; l < 32; l++) E(e, 32 + l) = E(e, l);
        };
        if (direct && nr > 256 && nr <= 320) {
            std::vector<real_t> tlv;
            build_direct(5, 64, tlv);
            td5_off = img.size();
            img.insert(img.end(), tlv.begin(), tlv.end());
        }
        if (direct && nr == 257) {  // (n = 257: with the extra row; n = 256: a v2p operator, 32 x 8 rows)
            std::vector<real_t> tlv;
            build_direct(8, 32, tlv);
            td8h_off = img.size();
            img.insert(img.end(), tlv.begin(), tlv.end());
        }
    }
    // CIRCULANT form (round 6, xscan_core.h circ_solve): a periodic operator on a uniform grid whose rows all carry the
    // bulk stencil is the circulant tridiagonal system (alpha, 1, alpha) -- two constant-coefficient recurrences, no lane
    // tables at all.  alpha = dist_af(5) (the reference's one bulk alpha, distributed.f90:83); checked here against the
    // reference's sweeps + 2 x 2 closure (periodic self-exchange) before it is offered.
    t->circ_ok = 0;
    t->circ_open_ok = 0;
    memset(&t->circ, 0, sizeof(t->circ));
    size_t circ_rb_off = 0;
    int circ_ws = 1, circ_we = 1;
    real_t circ_sa1 = 0.0, circ_scn = 0.0;
    {
        bool bulk = true, unif = true;
        for (int r = 0; r < 4; r++)
            for (int m = 0; m < 9; m++)
                if (coeffs_s[r * 9 + m] != coeffs[m] || coeffs_e[r * 9 + m] != coeffs[m]) bulk = false;
        for (int j = 1; j <= n; j++)
            if (St[j] != 1.0 || Stc[j] != 0.0) unif = false;
        const real_t alpha = n >= 8 ? dist_af[4] : 0.0;
        // (bulk: periodic, or both ends open to neighbour ranks -- BC_HALO: the same rows; the closed form is offered to
        //  periodic operators only, the open one to both)
        if (bulk && unif && nr == n && (Q == 4 || Q == 8 || Q == 16) && n == 64 * Q && alpha != 0.0 && fabs(alpha) < 0.5) {
            const double a = (double)alpha, rho = (1.0 - sqrt(1.0 - 4.0 * a * a)) / (2.0 * a);
            CircOp &co = t->circ;
            for (int m = 0; m < 9; m++) co.c[m] = (real_t)(coeffs[m] * (rho / a));
            co.nr = (real_t)(-rho);
            double pw = 1.0;
            for (int q = 0; q < 8; q++) { pw *= -rho; co.pf[q] = (real_t)pw; }
            double mu = 1.0;
            for (int q = 0; q < Q; q++) mu *= -rho;
            co.mu[0] = (real_t)mu; co.mu[1] = (real_t)(mu * mu); co.mu[2] = (real_t)(mu * mu * mu * mu);
            co.mu[3] = (real_t)pow(mu, 8.0);
            co.phi0 = (real_t)(-rho / (1.0 - rho * rho));
            // (the scans drop mu^8 at 8 rows per lane, mu^16 at 4: below 2^-60 or the form is not offered)
            bool ok = pow(fabs(mu), Q >= 8 ? 8.0 : 16.0) < 8.673617379884035e-19;
            // host check: the reference's sweeps with the periodic closure against the two recurrences run twice around
            // the ring (rho^n underflows: once around reaches the fixed point to the last bit)
            std::vector<real_t> r(n + 2), d(n + 2), x(n + 2), e(n + 2), y(n + 2);
            unsigned long long s = 0x9E3779B97F4A7C15ull;
            for (int rep = 0; rep < 2 && ok; rep++) {
                for (int j = 1; j <= n; j++) {
                    s = s * 6364136223846793005ull + 1442695040888963407ull;
                    r[j] = (real_t)((double)(s >> 11) / 9007199254740992.0 - 0.5);
                }
                d[1] = r[1] * F[1];
                d[2] = r[2] * F[2];
                for (int j = 3; j <= n; j++) d[j] = F[j] * (r[j] - A[j] * d[j - 1]);
                x = d;
                for (int j = n - 2; j >= 2; j--) x[j] = d[j] - Bw[j] * x[j + 1];
                x[1] = dist_fw[0] * (d[1] - Bw[1] * x[2]);
                const real_t sa1 = dist_sa[0], scn = dist_sc[n - 1];
                const real_t du_s = (x[1] - sa1 * x[n]) / (1.0 - sa1 * sa1), du_e = (x[n] - scn * x[1]) / (1.0 - scn * scn);
                real_t ee = 0.0;
                for (int pass = 0; pass < 2; pass++)
                    for (int j = 1; j <= n; j++) { ee = (real_t)(rho / a) * r[j] + co.nr * ee; e[j] = ee; }
                real_t yy = 0.0;
                for (int pass = 0; pass < 2; pass++)
                    for (int j = n; j >= 1; j--) { yy = e[j] + co.nr * yy; y[j] = yy; }
                real_t xmax = 0.0, emax = 0.0;
                for (int j = 1; j <= n; j++) {
                    const real_t ref = j == 1 ? du_s : (j == n ? du_e : x[j] - Sa[j] * du_s - Sc[j] * du_e);
                    xmax = fmax(xmax, fabs(ref));
                    emax = fmax(emax, fabs(ref - y[j]));
                }
                if (!(emax <= (sizeof(real_t) == 8 ? 1e-13 : 2e-5) * xmax)) ok = false;
            }
            t->circ_ok = (ok && periodic) ? 1 : 0;
            t->circ_open_ok = ok ? 1 : 0;
            if (ok) {
                // HALO form (a decomposed direction): every rank runs the two recurrences over ITS rows with open ends, hands
                // its forward end state E to the next rank and its first row x_1 to the previous one; what the received
                // values add is geometric -- E_prev (-rho)^j / (1 - rho^2) from the start, x_1,next (-rho)^(n-j+1) from the
                // end -- and goes through the strip kernels of the table form (xscan.hip, halo_fix_row) on a second set of
                // row records: Sa'_j = -(-rho)^(j-1), Sc'_j = -(-rho)^(n-j), St = 1; ds = -(rho / (1 - rho^2)) E_prev,
                // de = -rho x_1,next
                circ_rb_off = img.size();
                img.resize(img.size() + (size_t)8 * L, 0.0);
                real_t *rb = &img[circ_rb_off];
                const double tiny = 8.673617379884035e-19;
                circ_ws = 1; circ_we = 1;
                for (int j = 1; j <= n; j++) {
                    const double sa = -pow(-rho, (double)(j - 1)), sc = -pow(-rho, (double)(n - j));
                    rb[(size_t)8 * j + 1] = fabs(sa) >= tiny ? (real_t)sa : 0.0;
                    rb[(size_t)8 * j + 2] = fabs(sc) >= tiny ? (real_t)sc : 0.0;
                    rb[(size_t)8 * j + 3] = 1.0;
                    if (fabs(sa) >= tiny) circ_ws = j;
                    if (fabs(sc) >= tiny && n - j + 1 > circ_we) circ_we = n - j + 1;
                }
                circ_sa1 = (real_t)(rho / (1.0 - rho * rho));
                circ_scn = (real_t)rho;
            }
        }
    }
    X3D_HIP(hipMalloc(&t->dev, sizeof(real_t) * img.size()));
    X3D_HIP(hipMemcpy(t->dev, img.data(), sizeof(real_t) * img.size(), hipMemcpyHostToDevice));
    t->tl5 = tl5_off ? t->dev + tl5_off : nullptr;
    t->td5 = td5_off ? t->dev + td5_off : nullptr;
    t->td8h = td8h_off ? t->dev + td8h_off : nullptr;
    TdsTab &tb = t->tab;
    tb.n_tds = n; tb.n_rhs = nr; tb.chunk = chunk;
    tb.RF = t->dev; tb.RB = t->dev + (size_t)4 * L;
    tb.Cs = t->dev + (size_t)12 * L;
    tb.Q = Q;
    tb.bulk_only = 1;
    for (int r = 0; r < 4; r++)
        for (int m = 0; m < 9; m++)
            if (coeffs_s[r * 9 + m] != coeffs[m] || coeffs_e[r * 9 + m] != coeffs[m]) tb.bulk_only = 0;
    t->uniform = 1;
    for (int j = 1; j <= n; j++)
        if (St[j] != 1.0 || Stc[j] != 0.0) t->uniform = 0;
    t->narrow_all = 1;
    for (int m = 0; m < 9; m++) {
        if (m >= 2 && m <= 6) continue;
        if (coeffs[m] != 0.0) t->narrow_all = 0;
        for (int r = 0; r < 4; r++)
            if (coeffs_s[r * 9 + m] != 0.0 || coeffs_e[r * 9 + m] != 0.0) t->narrow_all = 0;
    }
    tb.TL = Q ? t->dev + tl_off : nullptr;
    t->tlc = tlc_off ? t->dev + tlc_off : nullptr;
    t->tl_hash = 1469598103934665603ull;
    if (Q) {
        const unsigned char *bytes = reinterpret_cast<const unsigned char *>(&img[tl_off]);
        for (size_t i = 0; i < (img.size() - tl_off) * sizeof(real_t); i++)
            t->tl_hash = (t->tl_hash ^ bytes[i]) * 1099511628211ull;
        for (int m = 0; m < 81; m++) {  // + the boundary and bulk stencils (Cs: [4][9] start, [4][9] end, [9] bulk)
            const unsigned char *cb = reinterpret_cast<const unsigned char *>(&Cs[m]);
            for (size_t i = 0; i < sizeof(real_t); i++) t->tl_hash = (t->tl_hash ^ cb[i]) * 1099511628211ull;
        }
    }
    tb.last_r = dist_fw[0];
    tb.bw1 = dist_bw[0];
    tb.sa1 = dist_sa[0];
    tb.scn = dist_sc[n - 1];
    tb.rs_s = 1.0 / (1.0 - dist_sa[0] * dist_sa[0]);          // distributed.f90:196-198
    tb.rs_e = 1.0 / (1.0 - dist_sc[n - 1] * dist_sc[n - 1]);  // distributed.f90:203-205
    t->tabc = tb;
    t->halo_ws_c = circ_ws; t->halo_we_c = circ_we;
    if (t->circ_open_ok) {
        t->tabc.RB = t->dev + circ_rb_off;
        t->tabc.rs_s = 1.0; t->tabc.sa1 = circ_sa1;
        t->tabc.rs_e = 1.0; t->tabc.scn = circ_scn;
    }
    // rows that still feel the reduced system's unknowns: |dist_sa(j)| decays from row 1, |dist_sc(j)| from row n
    // (the same decay lets the reference truncate the coupling to a 2 x 2 system, src/tdsops.f90:196-201)
    const real_t tiny = 8.673617379884035e-19;  // 2^-60
    t->halo_ws = 1;
    t->halo_we = 1;
    for (int j = 1; j <= n; j++) {
        if (fabs(dist_sa[j - 1]) >= tiny) t->halo_ws = j;
        if (fabs(dist_sc[j - 1]) >= tiny && n - j + 1 > t->halo_we) t->halo_we = n - j + 1;
    }
    *out = t;
    return 0;
}

extern "C" int x3d_tdsops_destroy(x3d_tdsops *t)
{
    X3D_RANGE(__func__);
    if (!t) return 0;
    x3d_penta_free(t);
    hipFree(t->dev);
    delete t;
    return 0;
}

// rows 1..out[0] and n-out[1]+1..n: where the coupling to the reduced system's unknowns is still above 2^-60,
// i.e. the rows x3d_*_halo_fix touch
extern "C" int x3d_tdsops_dims(const x3d_tdsops *t, int out[2])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(t && out, "x3d_tdsops_dims: null argument");
    out[0] = t->n_tds;
    out[1] = t->tab.n_rhs;
    return 0;
}

extern "C" int x3d_tdsops_halo_rows(const x3d_tdsops *t, int out[2])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(t && out, "x3d_tdsops_halo_rows: null argument");
    out[0] = t->halo_ws; out[1] = t->halo_we;
    return 0;
}

extern "C" int x3d_npencils(const x3d_backend *b, int dir)
{
    X3D_RANGE(__func__);
    if (!b || !x3d_dir_ok(dir)) return -1;
    return x3d_geom(b, dir).np;
}

// ------------------------------------------------------------------ kernels
// Extended pencil row jj in [-3, n_rhs+4] (1-based).  HB: halos come from
// exchange buffers [4][np]; otherwise the direction is not decomposed and the
// halo is the periodic image, exactly what sendrecv_fields' nproc==1 branch
// produces (src/backend/omp/sendrecv.f90:20-22): u_s(r) = u(n_wrap-4+r),
// u_e(r) = u(r).
template <bool HB>
__device__ __forceinline__ real_t ext_row(const real_t *__restrict__ u, long base, long rs, int jj, int nr,
                                          int n_wrap, const real_t *__restrict__ hs,
                                          const real_t *__restrict__ he, int np, int p)
{
    if (jj < 1) {
        if (HB) return hs[(long)(jj + 3) * np + p];
        return u[base + (long)(n_wrap + jj - 1) * rs];
    }
    if (jj > nr) {
        const int r = jj - nr - 1;
        if (HB) return he[(long)r * np + p];
        return u[base + (long)r * rs];
    }
    return u[base + (long)(jj - 1) * rs];  // (cached on purpose: a nontemporal load here costs 12 %, the fields are
                                           //  re-read by the next component's kernel)
}

__device__ __forceinline__ real_t dot9(const real_t *__restrict__ c, const real_t (&w)[9])
{
    // same left-to-right order as distributed.f90:89-93
    return c[0] * w[0] + c[1] * w[1] + c[2] * w[2] + c[3] * w[3] + c[4] * w[4] + c[5] * w[5] + c[6] * w[6] +
           c[7] * w[7] + c[8] * w[8];
}

__device__ __forceinline__ const real_t *stencil_row(const real_t *__restrict__ Cs, int j, int nr)
{
    if (j <= 4) return Cs + (j - 1) * 9;
    if (j > nr - 4) return Cs + 36 + (j - (nr - 4) - 1) * 9;
    return Cs + 72;
}

__device__ __forceinline__ long pencil_base(const PencilGeom &g, int p)
{
    return (long)(p % g.dim0) * g.s0 + (long)(p / g.dim0) * g.s1;
}

// forward sweep, one operator (der_univ_dist without its backward loop)
template <bool HB>
__global__ void __launch_bounds__(64) k_tds_fwd(real_t *__restrict__ d, real_t *__restrict__ send_s,
                                                real_t *__restrict__ send_e, const real_t *__restrict__ u,
                                                const real_t *__restrict__ hs, const real_t *__restrict__ he,
                                                TdsTab t, PencilGeom g, int n_wrap)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = pencil_base(g, p), rs = g.rs;
    const int n = t.n_tds, nr = t.n_rhs;
    real_t w[9];
#pragma unroll
    for (int m = 0; m < 9; m++) w[m] = ext_row<HB>(u, base, rs, m - 3, nr, n_wrap, hs, he, g.np, p);
    real_t dprev = 0.0, S = 0.0, d1 = 0.0, dn = 0.0;

    auto row = [&](int j, const real_t *__restrict__ c, real_t wnext) {
        const real_t acc = dot9(c, w);
        const real_t dj = T_F(t, j) * (acc - T_A(t, j) * dprev);
        if (j <= n) {
            __builtin_nontemporal_store(dj, &d[base + (long)(j - 1) * rs]);
            S += T_W(t, j) * dj;
            if (j == 1) d1 = dj;
            if (j == n) dn = dj;
        }
        dprev = dj;
#pragma unroll
        for (int m = 0; m < 8; m++) w[m] = w[m + 1];
        w[8] = wnext;
    };

    for (int j = 1; j <= 4; j++)
        row(j, stencil_row(t.Cs, j, nr), ext_row<HB>(u, base, rs, j + 5, nr, n_wrap, hs, he, g.np, p));
    {
        const real_t *__restrict__ c = t.Cs + 72;
        const real_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], c6 = c[6], c7 = c[7],
                     c8 = c[8];
        const real_t cb[9] = {c0, c1, c2, c3, c4, c5, c6, c7, c8};
#pragma unroll 4
        for (int j = 5; j <= nr - 4; j++) {
            // rows j+5 <= n_rhs+1: interior except for the very last bulk row
            const real_t wn = (j + 5 <= nr) ? u[base + (long)(j + 4) * rs]
                                            : ext_row<HB>(u, base, rs, j + 5, nr, n_wrap, hs, he, g.np, p);
            row(j, cb, wn);
        }
    }
    for (int j = (nr - 3 > 5 ? nr - 3 : 5); j <= nr; j++) {
        const real_t wn = (j < nr) ? ext_row<HB>(u, base, rs, j + 5, nr, n_wrap, hs, he, g.np, p) : 0.0;
        row(j, stencil_row(t.Cs, j, nr), wn);
    }
    send_e[p] = dn;                                 // distributed.f90:147-151
    send_s[p] = t.last_r * (d1 - t.bw1 * S);         // distributed.f90:161-166 with du_2 = S
}

// backward sweep fused with der_univ_subs.  d holds the forward-eliminated
// values and never aliases du (the launchers keep it in backend scratch), so
// both are __restrict__ and the loads of a batch of UB rows are issued before
// the serial back-substitution touches them.  ACC: du += scale * result, which
// is how the fused driver folds sum_yintox/sum_zintox and the velocity
// correction vecadd(-1, dpdx, 1, u) into this pass.
#define UB 8
template <bool ACC>
__global__ void __launch_bounds__(64) k_tds_bwd(real_t *__restrict__ du, const real_t *__restrict__ d,
                                                const real_t *__restrict__ own_s,
                                                const real_t *__restrict__ recv_s,
                                                const real_t *__restrict__ recv_e, TdsTab t, PencilGeom g,
                                                real_t scale)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = pencil_base(g, p), rs = g.rs;
    const int n = t.n_tds;
    auto put = [&](long o, real_t v, real_t old) {
        __builtin_nontemporal_store(ACC ? old + scale * v : v, &du[o]);
    };
    const real_t dn = d[base + (long)(n - 1) * rs];
    const real_t du_s = t.rs_s * (own_s[p] - t.sa1 * recv_s[p]);  // distributed.f90:196-199
    const real_t du_e = t.rs_e * (dn - t.scn * recv_e[p]);        // distributed.f90:203-206
    {
        const long o = base + (long)(n - 1) * rs;
        put(o, du_e * T_ST(t, n), ACC ? du[o] : 0.0);  // :224-228
    }
    real_t nxt = d[base + (long)(n - 2) * rs];  // row n-1: no backward update
    {
        const long o = base + (long)(n - 2) * rs;
        put(o, (nxt - T_SA(t, n - 1) * du_s - T_SC(t, n - 1) * du_e) * T_ST(t, n - 1), ACC ? du[o] : 0.0);
    }
    int j = n - 2;
    for (; j - UB + 1 >= 2; j -= UB) {
        real_t dv[UB], ov[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const long o = base + (long)(j - k - 1) * rs;
            dv[k] = __builtin_nontemporal_load(&d[o]);
            ov[k] = ACC ? __builtin_nontemporal_load(&du[o]) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const int jj = j - k;
            const real_t cur = dv[k] - T_BW(t, jj) * nxt;                                      // :154-160
            put(base + (long)(jj - 1) * rs, (cur - T_SA(t, jj) * du_s - T_SC(t, jj) * du_e) * T_ST(t, jj), ov[k]);  // :215-222
            nxt = cur;
        }
    }
    for (; j >= 2; j--) {
        const long o = base + (long)(j - 1) * rs;
        const real_t cur = d[o] - T_BW(t, j) * nxt;
        put(o, (cur - T_SA(t, j) * du_s - T_SC(t, j) * du_e) * T_ST(t, j), ACC ? du[o] : 0.0);
        nxt = cur;
    }
    put(base, du_s * T_ST(t, 1), ACC ? du[base] : 0.0);  // :209-213
}

// forward sweep of one transport-equation component: three operators share
// the loads of u and conv (exec_dist.f90:114-160)
template <bool HB, bool SAME>
__global__ void __launch_bounds__(64)
    k_transeq_fwd(real_t *__restrict__ d_du, real_t *__restrict__ d_dud, real_t *__restrict__ d_d2u,
                  real_t *__restrict__ send_s, real_t *__restrict__ send_e, const real_t *__restrict__ u,
                  const real_t *__restrict__ us, const real_t *__restrict__ ue, const real_t *__restrict__ cv,
                  const real_t *__restrict__ cs, const real_t *__restrict__ ce, TdsTab t1, TdsTab t2, TdsTab t3,
                  PencilGeom g, int n_wrap, int npmax)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = pencil_base(g, p), rs = g.rs;
    const int n = t1.n_tds;  // n_rhs == n_tds for first/second derivatives (src/tdsops.f90:114-123)
    real_t wu[9], wp[9];
#pragma unroll
    for (int m = 0; m < 9; m++) {
        wu[m] = ext_row<HB>(u, base, rs, m - 3, n, n_wrap, us, ue, g.np, p);
        const real_t c = SAME ? wu[m] : ext_row<HB>(cv, base, rs, m - 3, n, n_wrap, cs, ce, g.np, p);
        wp[m] = wu[m] * c;  // ud = u*v incl. the halo products, exec_dist.f90:133-149
    }
    real_t p1 = 0, p2 = 0, p3 = 0, S1 = 0, S2 = 0, S3 = 0, f1 = 0, f2 = 0, f3 = 0, l1 = 0, l2 = 0, l3 = 0;

    auto row = [&](int j, const real_t *__restrict__ c1, const real_t *__restrict__ c2,
                   const real_t *__restrict__ c3, real_t un, real_t pn) {
        const real_t a1 = dot9(c1, wu), a3 = dot9(c3, wu), a2 = dot9(c2, wp);
        const real_t e1 = T_F(t1, j) * (a1 - T_A(t1, j) * p1);
        const real_t e2 = T_F(t2, j) * (a2 - T_A(t2, j) * p2);
        const real_t e3 = T_F(t3, j) * (a3 - T_A(t3, j) * p3);
        {
            const long o = base + (long)(j - 1) * rs;
            __builtin_nontemporal_store(e1, &d_du[o]); __builtin_nontemporal_store(e2, &d_dud[o]);
            __builtin_nontemporal_store(e3, &d_d2u[o]);
        }
        S1 += T_W(t1, j) * e1; S2 += T_W(t2, j) * e2; S3 += T_W(t3, j) * e3;
        if (j == 1) { f1 = e1; f2 = e2; f3 = e3; }
        if (j == n) { l1 = e1; l2 = e2; l3 = e3; }
        p1 = e1; p2 = e2; p3 = e3;
#pragma unroll
        for (int m = 0; m < 8; m++) { wu[m] = wu[m + 1]; wp[m] = wp[m + 1]; }
        wu[8] = un; wp[8] = pn;
    };
    auto nextrow = [&](int jj, real_t &un, real_t &pn) {
        un = ext_row<HB>(u, base, rs, jj, n, n_wrap, us, ue, g.np, p);
        const real_t c = SAME ? un : ext_row<HB>(cv, base, rs, jj, n, n_wrap, cs, ce, g.np, p);
        pn = un * c;
    };

    for (int j = 1; j <= 4; j++) {
        real_t un, pn;
        nextrow(j + 5, un, pn);
        row(j, stencil_row(t1.Cs, j, n), stencil_row(t2.Cs, j, n), stencil_row(t3.Cs, j, n), un, pn);
    }
    {
        real_t b1[9], b2[9], b3[9];
#pragma unroll
        for (int m = 0; m < 9; m++) { b1[m] = t1.Cs[72 + m]; b2[m] = t2.Cs[72 + m]; b3[m] = t3.Cs[72 + m]; }
#pragma unroll 2
        for (int j = 5; j <= n - 4; j++) {
            real_t un, pn;
            if (j + 5 <= n) {
                un = u[base + (long)(j + 4) * rs];
                pn = un * (SAME ? un : cv[base + (long)(j + 4) * rs]);
            } else {
                nextrow(j + 5, un, pn);
            }
            row(j, b1, b2, b3, un, pn);
        }
    }
    for (int j = (n - 3 > 5 ? n - 3 : 5); j <= n; j++) {
        real_t un = 0.0, pn = 0.0;
        if (j < n) nextrow(j + 5, un, pn);
        row(j, stencil_row(t1.Cs, j, n), stencil_row(t2.Cs, j, n), stencil_row(t3.Cs, j, n), un, pn);
    }
    send_e[p] = l1; send_e[npmax + p] = l2; send_e[2 * npmax + p] = l3;
    send_s[p] = t1.last_r * (f1 - t1.bw1 * S1);
    send_s[npmax + p] = t2.last_r * (f2 - t2.bw1 * S2);
    send_s[2 * npmax + p] = t3.last_r * (f3 - t3.bw1 * S3);
}

// backward sweep fused with der_univ_fused_subs (distributed.f90:231-337).
// The three forward-eliminated arrays live in backend scratch and never alias
// rhs; ACC: rhs += result.  Loads of UT rows are batched ahead of the serial
// back-substitution.
#define UT 4
template <bool ACC>
__global__ void __launch_bounds__(64)
    k_transeq_bwd(real_t *__restrict__ rhs, const real_t *__restrict__ d_du, const real_t *__restrict__ d_dud,
                  const real_t *__restrict__ d_d2u, const real_t *__restrict__ cv,
                  const real_t *__restrict__ own_s, const real_t *__restrict__ recv_s,
                  const real_t *__restrict__ recv_e, real_t nu, TdsTab t1, TdsTab t2, TdsTab t3, PencilGeom g,
                  int npmax)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = pencil_base(g, p), rs = g.rs;
    const int n = t1.n_tds;
    const long on = base + (long)(n - 1) * rs;
    const real_t du_s = t1.rs_s * (own_s[p] - t1.sa1 * recv_s[p]);
    const real_t dud_s = t2.rs_s * (own_s[npmax + p] - t2.sa1 * recv_s[npmax + p]);
    const real_t d2u_s = t3.rs_s * (own_s[2 * npmax + p] - t3.sa1 * recv_s[2 * npmax + p]);
    // streaming data (read once / written once per launch): nontemporal accesses, measured -10 % on the
    // forward and -4 % on the backward kernel
    auto put = [&](long o, real_t v, real_t old) {
        __builtin_nontemporal_store(ACC ? old + v : v, &rhs[o]);
    };
    real_t n1 = d_du[on], n2 = d_dud[on], n3 = d_d2u[on];
    const real_t du_e = t1.rs_e * (n1 - t1.scn * recv_e[p]);
    const real_t dud_e = t2.rs_e * (n2 - t2.scn * recv_e[npmax + p]);
    const real_t d2u_e = t3.rs_e * (n3 - t3.scn * recv_e[2 * npmax + p]);
    // row n (:328-335)
    put(on, -0.5 * (cv[on] * du_e * T_ST(t1, n) + dud_e * T_ST(t2, n)) +
                nu * (d2u_e * T_ST(t3, n) + du_e * T_ST(t1, n) * T_STC(t3, n)), ACC ? rhs[on] : 0.0);

    auto emit = [&](int j, real_t c1, real_t c2, real_t c3, real_t v, real_t old) {
        const real_t temp_du = T_ST(t1, j) * (c1 - T_SA(t1, j) * du_s - T_SC(t1, j) * du_e);
        const real_t temp_dud = T_ST(t2, j) * (c2 - T_SA(t2, j) * dud_s - T_SC(t2, j) * dud_e);
        const real_t temp_d2u = T_ST(t3, j) * (c3 - T_SA(t3, j) * d2u_s - T_SC(t3, j) * d2u_e) + temp_du * T_STC(t3, j);
        put(base + (long)(j - 1) * rs, -0.5 * (v * temp_du + temp_dud) + nu * temp_d2u, old);  // :315-324
    };
    {
        const long o = base + (long)(n - 2) * rs;  // row n-1: forward values, no backward update
        n1 = d_du[o]; n2 = d_dud[o]; n3 = d_d2u[o];
        emit(n - 1, n1, n2, n3, cv[o], ACC ? rhs[o] : 0.0);
    }
    int j = n - 2;
    for (; j - UT + 1 >= 2; j -= UT) {
        real_t a1[UT], a2[UT], a3[UT], vv[UT], ov[UT];
#pragma unroll
        for (int k = 0; k < UT; k++) {
            const long o = base + (long)(j - k - 1) * rs;
            a1[k] = __builtin_nontemporal_load(&d_du[o]); a2[k] = __builtin_nontemporal_load(&d_dud[o]);
            a3[k] = __builtin_nontemporal_load(&d_d2u[o]); vv[k] = __builtin_nontemporal_load(&cv[o]);
            ov[k] = ACC ? __builtin_nontemporal_load(&rhs[o]) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < UT; k++) {
            const int jj = j - k;
            const real_t c1 = a1[k] - T_BW(t1, jj) * n1;
            const real_t c2 = a2[k] - T_BW(t2, jj) * n2;
            const real_t c3 = a3[k] - T_BW(t3, jj) * n3;
            emit(jj, c1, c2, c3, vv[k], ov[k]);
            n1 = c1; n2 = c2; n3 = c3;
        }
    }
    for (; j >= 2; j--) {
        const long o = base + (long)(j - 1) * rs;
        const real_t c1 = d_du[o] - T_BW(t1, j) * n1;
        const real_t c2 = d_dud[o] - T_BW(t2, j) * n2;
        const real_t c3 = d_d2u[o] - T_BW(t3, j) * n3;
        emit(j, c1, c2, c3, cv[o], ACC ? rhs[o] : 0.0);
        n1 = c1; n2 = c2; n3 = c3;
    }
    // row 1 (:304-311)
    put(base, -0.5 * (cv[base] * du_s * T_ST(t1, 1) + dud_s * T_ST(t2, 1)) +
                  nu * (d2u_s * T_ST(t3, 1) + du_s * T_ST(t1, 1) * T_STC(t3, 1)), ACC ? rhs[base] : 0.0);
}

// copy_into_buffers (src/backend/omp/backend.f90:714-737): rows 1..4 and n-3..n
__global__ void k_pack_halos(real_t *__restrict__ send_s, real_t *__restrict__ send_e,
                             const real_t *__restrict__ u, int n, PencilGeom g)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = pencil_base(g, p);
#pragma unroll
    for (int r = 0; r < X3D_NH; r++) {
        send_s[(long)r * g.np + p] = u[base + (long)r * g.rs];
        send_e[(long)r * g.np + p] = u[base + (long)(n - X3D_NH + r) * g.rs];
    }
}

// the same for nf fields at once, in the layout the HALO tile kernels read and ONE message per neighbour carries:
// send[side][field][4][np], side 0 = rows 1..4 (to prev), side 1 = rows n-3..n (to next)
struct PackFields { const real_t *f[3]; };
// (hp, hnp: TileHalo's plane layout -- pencil p = (x, o) goes to o * hp + x of a plane of hnp entries)
__global__ void k_pack_halos_multi(real_t *__restrict__ send, PackFields pf, int nf, int n, PencilGeom g, int hp, long hnp)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.np) return;
    const long base = pencil_base(g, p);
    const long hq = (long)(p / g.dim0) * hp + p % g.dim0;
    const int k = blockIdx.y;  // field
    const real_t *__restrict__ u = pf.f[k];
#pragma unroll
    for (int r = 0; r < X3D_NH; r++) {
        send[((long)k * X3D_NH + r) * hnp + hq] = u[base + (long)r * g.rs];
        send[((long)(nf + k) * X3D_NH + r) * hnp + hq] = u[base + (long)(n - X3D_NH + r) * g.rs];
    }
}

void x3d_halo_layout(const x3d_backend *b, int dir, int *hp, long *hnp)
{
    if (dir == X3D_DIR_Z) { *hp = b->nxp; *hnp = (long)b->nxp * b->nyp; }
    else { *hp = b->nx; *hnp = (long)x3d_geom(b, dir).np; }
}
// entries of one halo row of direction dir (y or z) in the tile kernels' halo buffers [side 2][nf][4][this]
extern "C" long x3d_halo_row_size(const x3d_backend *b, int dir)
{
    X3D_RANGE(__func__);
    if (!b || !x3d_dir_ok(dir)) return 0;
    int hp;
    long hnp;
    x3d_halo_layout(b, dir, &hp, &hnp);
    return hnp;
}

// ------------------------------------------------------------------ launchers
int x3d_onchip2_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir, int acc, real_t scale,
                    bool *done);  // onchip.hip (K1e)
static bool use_onchip2()
{
    static int mode = -1;
    if (mode < 0) {
        // default on: single-pass solve for periodic 512-row y/z pencils, 0.50 ms vs 0.85 ms for the
        // two-sweep pair (profiles/README.md); X3D_NO_ONCHIP2=1 falls back
        const char *e = getenv("X3D_NO_ONCHIP2");
        mode = (e && e[0] == '1') ? 0 : 1;
    }
    return mode == 1;
}
int x3d_transeq_via_x(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                      const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                      const x3d_tdsops *der2nd_sym, int acc, bool *done);  // viax.hip
int x3d_ytile_transeq(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                      const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc, bool *done);  // xscan.hip
int x3d_xscan_transeq3(x3d_backend *b, real_t *const r[3], const real_t *const f[3], real_t nu,
                       const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                       const x3d_tdsops *der2nd_sym, int acc, const real_t *const *upd_g, const x3d_tdsops *op_s,
                       const x3d_tdsops *op_i, real_t scale, real_t omega, const real_t *ushift, bool *done);  // xscan.hip
// xdir.hip
int x3d_xwide_transeq3_upd(x3d_backend *b, real_t *const r[3], real_t *const f[3], real_t nu, const x3d_tdsops *der1st,
                           const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                           const real_t *const g[3], const x3d_tdsops *op_s, const x3d_tdsops *op_i, real_t scale,
                           real_t omega, const real_t *ushift, bool *done);
int x3d_xwide_transeq3(x3d_backend *b, real_t *const r[3], const real_t *const f[3], real_t nu, const x3d_tdsops *der1st,
                       const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym, int acc,
                       real_t omega, const real_t *ushift, bool *done);  // xwide.hip
int x3d_ygen_pair(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                  const x3d_tdsops *ta, const x3d_tdsops *tb, bool *done);  // ygen.hip
int x3d_ygen_transeq3(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                      const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                      const x3d_tdsops *der2nd_sym, int acc, bool *done);  // ygen.hip
int x3d_xdir_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int acc, real_t scale);
int x3d_xdir_transeq(x3d_backend *b, real_t *rhs, const real_t *u, const real_t *conv, real_t nu,
                     const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3, int acc);
int x3d_generic_tds_local(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir, int acc,
                          real_t scale);
int x3d_generic_transeq_local(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv,
                              real_t nu, const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3,
                              int acc);

static inline dim3 grid_for(const PencilGeom &g) { return dim3((g.np + 63) / 64); }

static int check_len(const x3d_backend *b, const x3d_tdsops *t, int dir, const char *who)
{
    const int ext = dir == X3D_DIR_X ? b->nx : (dir == X3D_DIR_Y ? b->ny : b->nz);
    X3D_REQUIRE(t->n_rhs <= ext, "%s: operator needs %d rows but the block has %d along dir %d", who, t->n_rhs,
                ext, dir);
    return 0;
}

extern "C" int x3d_pack_halos(x3d_backend *b, real_t *send_s, real_t *send_e, const real_t *u, int n, int dir)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && send_s && send_e && u, "x3d_pack_halos: null argument");
    // deferred execution on several ranks: what has been recorded runs first, then this entry point works on the
    // buffer that holds u's data (the exchange buffers are not handles)
    X3D_LAZY_IN(b, u);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_pack_halos: bad dir %d", dir);
    PencilGeom g = x3d_geom(b, dir);
    ProfScope ps(b, X3D_K_PACK, dir);
    hipLaunchKernelGGL(k_pack_halos, dim3((g.np + 255) / 256), dim3(256), 0, b->stream, send_s, send_e, u, n, g);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_tds_dist_fwd(x3d_backend *b, real_t *du, real_t *du_send_s, real_t *du_send_e,
                                const real_t *u, const real_t *u_recv_s, const real_t *u_recv_e,
                                const x3d_tdsops *t, int dir)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && du && du_send_s && du_send_e && u && u_recv_s && u_recv_e && t,
                "x3d_tds_dist_fwd: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_tds_dist_fwd: bad dir %d", dir);
    X3D_REQUIRE(du != u, "x3d_tds_dist_fwd: du and u must be distinct blocks");
    X3D_LAZY_IN(b, u);
    X3D_LAZY_EAGER(b);
    if (int rc = check_len(b, t, dir, "tds_solve")) return rc;
    PencilGeom g = x3d_geom(b, dir);
    ProfScope ps(b, X3D_K_TDS_FWD, dir);
    (void)du;  // the eliminated values stay in backend scratch until x3d_tds_dist_bwd
    hipLaunchKernelGGL((k_tds_fwd<true>), grid_for(g), dim3(64), 0, b->stream, b->scratch[2], du_send_s,
                       du_send_e, u, u_recv_s, u_recv_e, t->tab, g, t->n_tds);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_tds_dist_bwd_acc(x3d_backend *b, real_t *du, const real_t *du_send_s, const real_t *du_recv_s,
                                    const real_t *du_recv_e, const x3d_tdsops *t, int dir, int accumulate,
                                    real_t scale)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && du && du_send_s && du_recv_s && du_recv_e && t, "x3d_tds_dist_bwd: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_tds_dist_bwd: bad dir %d", dir);
    X3D_LAZY_OUT(b, du, !accumulate);
    X3D_LAZY_EAGER(b);
    PencilGeom g = x3d_geom(b, dir);
    ProfScope ps(b, X3D_K_TDS_BWD, dir);
    if (accumulate)  // fusion extension: du += scale * result (folds the vecadd of the fused driver)
        hipLaunchKernelGGL(k_tds_bwd<true>, grid_for(g), dim3(64), 0, b->stream, du, (const real_t *)b->scratch[2],
                           du_send_s, du_recv_s, du_recv_e, t->tab, g, scale);
    else
        hipLaunchKernelGGL(k_tds_bwd<false>, grid_for(g), dim3(64), 0, b->stream, du, (const real_t *)b->scratch[2],
                           du_send_s, du_recv_s, du_recv_e, t->tab, g, 1.0);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_tds_dist_bwd(x3d_backend *b, real_t *du, const real_t *du_send_s, const real_t *du_recv_s,
                                const real_t *du_recv_e, const x3d_tdsops *t, int dir)
{
    X3D_RANGE(__func__);
    return x3d_tds_dist_bwd_acc(b, du, du_send_s, du_recv_s, du_recv_e, t, dir, 0, 1.0);
}

// Local form: sendrecv_fields with nproc==1 hands every rank its own buffers
// back swapped (src/backend/omp/sendrecv.f90:20-22): recv_s = send_e, recv_e = send_s.
extern "C" int x3d_tds_solve_acc(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir,
                                 int accumulate, real_t scale)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && u && t, "x3d_tds_solve: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_tds_solve: bad dir %d", dir);
    X3D_REQUIRE(du != u, "x3d_tds_solve: du and u must be distinct blocks");
    if (int rc = check_len(b, t, dir, "tds_solve")) return rc;
    if (dir == X3D_DIR_X) return x3d_xdir_tds(b, du, u, t, accumulate, scale);
    if (use_onchip2()) {
        bool done = false;
        if (int rc = x3d_onchip2_tds(b, du, u, t, dir, accumulate, scale, &done)) return rc;
        if (done) return 0;
    }
    if (!accumulate) {  // K3g (ygen.hip): non-periodic / odd-length pencils through the LDS tile, single pass
        bool done = false;
        if (int rc = x3d_ygen_pair(b, dir, 2, du, nullptr, u, nullptr, t, t, &done)) return rc;
        if (done) return 0;
    }
    return x3d_generic_tds_local(b, du, u, t, dir, accumulate, scale);
}

int x3d_ytile_tds_pair(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                       const x3d_tdsops *ta, const x3d_tdsops *tb, const TileHalo *halo, int other0, int nother,
                       bool *done);  // xscan.hip
int x3d_tds_halo_fix(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *brecv,
                     const x3d_tdsops *ta, const x3d_tdsops *tb);  // xscan.hip
int x3d_transeq_halo_fix_launch(x3d_backend *b, int dir, real_t *const r[3], const real_t *conv, real_t nu,
                                const real_t *brecv, const x3d_tdsops *der1st, const x3d_tdsops *der2nd);  // xscan.hip
int x3d_ytile_transeq3(x3d_backend *b, int dir, real_t *const r[3], const real_t *const f[3], real_t nu,
                       const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                       const x3d_tdsops *der2nd_sym, int acc, const TileHalo *halo, int other0, int nother,
                       bool *done);  // xscan.hip

// fusion extension for the operator pairs of divergence_v2c / gradient_c2v (src/vector_calculus.f90:142-332):
//   mode 0: out1 = A(in1) + B(in2)          mode 1: out1 = A(in1), out2 = B(in1)
// one kernel for periodic 256 / 512-row y pencils (xscan.hip, k_ytile_tds_pair), else the two tds_solve's
extern "C" int x3d_tds_solve_pair(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *in1,
                                  const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && out1 && in1 && ta && tb && (mode == 0 ? in2 != nullptr : out2 != nullptr),
                "x3d_tds_solve_pair: null argument");
    X3D_REQUIRE(mode == 0 || mode == 1, "x3d_tds_solve_pair: mode must be 0 or 1");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_tds_solve_pair: bad dir %d", dir);
    X3D_REQUIRE(out1 != in1 && out1 != in2 && out2 != in1 && (mode == 0 || out1 != out2),
                "x3d_tds_solve_pair: outputs alias inputs");
    if (int rc = check_len(b, ta, dir, "tds_solve_pair")) return rc;
    if (int rc = check_len(b, tb, dir, "tds_solve_pair")) return rc;
    if (dir != X3D_DIR_X) {
        bool done = false;
        if (int rc = x3d_ytile_tds_pair(b, dir, mode, out1, out2, in1, in2, ta, tb, nullptr, 0, -1, &done)) return rc;
        if (done) return 0;
        if (int rc = x3d_ygen_pair(b, dir, mode, out1, out2, in1, in2, ta, tb, &done)) return rc;  // K3g
        if (done) return 0;
    }
    if (int rc = x3d_tds_solve_acc(b, out1, in1, ta, dir, 0, 1.0)) return rc;
    return mode == 0 ? x3d_tds_solve_acc(b, out1, in2, tb, dir, 1, 1.0) : x3d_tds_solve_acc(b, out2, in1, tb, dir, 0, 1.0);
}

// fusion extension for the 010 Poisson solve (non-periodic y): the z pair next to the solver does the solver's
// interleave of the y rows on the way (enforce_periodicity_y / undo_periodicity_y,
// src/backend/cuda/kernels/spectral_processing.f90:1062-1114, ny even: row 2j-1 <-> position j, row 2j <->
// position ny-j+1, 1-based).  mode 0 (divergence's last pair): out1's y rows [0, ny) are written at their
// positions = what enforce_periodicity_y would have made of the result; mode 1 (gradient's first pair): in1's y rows
// are read from their positions = in1 is what the solver's backward transform left, undo_periodicity_y not run.
// *done = 0: these pencils are not served by a tile kernel, nothing was done (run the copies + x3d_tds_solve_pair).
extern "C" int x3d_tds_solve_pair_yperm(x3d_backend *b, int mode, real_t *out1, real_t *out2, const real_t *in1,
                                        const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb, int ny, int *done)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && out1 && in1 && ta && tb && done && (mode == 0 ? in2 != nullptr : out2 != nullptr),
                "x3d_tds_solve_pair_yperm: null argument");
    X3D_REQUIRE(mode == 0 || mode == 1, "x3d_tds_solve_pair_yperm: mode must be 0 or 1");
    X3D_REQUIRE(out1 != in1 && out1 != in2 && out2 != in1 && (mode == 0 || out1 != out2),
                "x3d_tds_solve_pair_yperm: outputs alias inputs");
    X3D_REQUIRE(ny > 0 && ny <= b->ny, "x3d_tds_solve_pair_yperm: ny = %d outside the block (%d rows)", ny, b->ny);
    *done = 0;
    if (int rc = check_len(b, ta, X3D_DIR_Z, "tds_solve_pair_yperm")) return rc;
    if (int rc = check_len(b, tb, X3D_DIR_Z, "tds_solve_pair_yperm")) return rc;
    if (ny & 1) return 0;  // (the odd-size interleave keeps a centre row: the copy kernels handle it)
    bool ok = false;
    b->pair_yperm = ny;
    int rc = x3d_ytile_tds_pair(b, X3D_DIR_Z, mode, out1, out2, in1, in2, ta, tb, nullptr, 0, -1, &ok);
    if (!rc && !ok) rc = x3d_ygen_pair(b, X3D_DIR_Z, mode, out1, out2, in1, in2, ta, tb, &ok);
    b->pair_yperm = 0;
    if (rc) return rc;
    *done = ok ? 1 : 0;
    return 0;
}

int x3d_xscan_tds_lincomb(x3d_backend *b, real_t *du, const x3d_tdsops *t, real_t *y, const real_t *base, int nterm,
                          const real_t *c, const real_t *const *x, const real_t *wall, bool *done);  // xscan.hip
int x3d_xwide_tds_lincomb(x3d_backend *b, real_t *du, const x3d_tdsops *t, real_t *y, const real_t *base, int nterm,
                          const real_t *c, const real_t *const *x, const real_t *wall, bool *done, real_t *psum = nullptr,
                          int ny_sum = 0, int *nsum = nullptr);  // xwide.hip
extern "C" int x3d_field_set_face_from_field(x3d_backend *b, real_t *f, const real_t *f_start, const int dims[3],
                                             real_t c_end, int face, real_t flow_rate_diff);
extern "C" int x3d_lincomb(x3d_backend *b, real_t *y, const real_t *base, int nterm, const real_t *c,
                           const real_t *const *x);

// fusion extension: y = base + sum_i c[i] x[i] (x3d_lincomb) followed by du = tds_solve(y), in one kernel for
// periodic 256 / 512-point x pencils (y is not read back); otherwise the two calls one after the other
// nsum != null: the caller wants the sum of y over dims_sum's rows as well (x3d_tds_solve_lincomb_wall_mean): where a kernel
// takes it along, its partials are in b->red_buf and *nsum says how many; else *nsum = 0
static int tds_solve_lincomb_wall(x3d_backend *b, int dir, real_t *du, const x3d_tdsops *t, real_t *y, const real_t *base,
                                  int nterm, const real_t *c, const real_t *const *x, const real_t *wall,
                                  const int *dims_sum = nullptr, int *nsum = nullptr)
{
    if (nsum) *nsum = 0;
    X3D_REQUIRE(b && du && t && y && base && c && x, "x3d_tds_solve_lincomb: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_tds_solve_lincomb: bad dir %d", dir);
    X3D_REQUIRE(nterm >= 1 && nterm <= 5, "x3d_tds_solve_lincomb: nterm must be 1..5");
    X3D_REQUIRE(du != y && du != base && du != wall && y != wall, "x3d_tds_solve_lincomb: du aliases an input");
    for (int k = 0; k < nterm; k++) X3D_REQUIRE(du != x[k], "x3d_tds_solve_lincomb: du aliases an input");
    if (int rc = check_len(b, t, dir, "tds_solve_lincomb")) return rc;
    if (dir == X3D_DIR_X) {
        bool done = false;
        if (int rc = x3d_xscan_tds_lincomb(b, du, t, y, base, nterm, c, x, wall, &done)) return rc;
        if (done) return 0;
        // (n = 1024; the sum rides along where it covers whole x pencils and every z plane)
        const bool sum = nsum && dims_sum && dims_sum[0] == b->nx && dims_sum[2] == b->nz && dims_sum[1] <= b->ny;
        if (int rc = x3d_xwide_tds_lincomb(b, du, t, y, base, nterm, c, x, wall, &done, sum ? b->red_buf : nullptr,
                                           sum ? dims_sum[1] : 0, sum ? nsum : nullptr))
            return rc;
        if (done) return 0;
    }
    if (int rc = x3d_lincomb(b, y, base, nterm, c, x)) return rc;
    if (wall) {
        const int dims[3] = {b->nx, b->ny, b->nz};
        if (int rc = x3d_field_set_face_from_field(b, y, wall, dims, 0.0, X3D_Y_FACE, 0.0)) return rc;
    }
    return x3d_tds_solve_acc(b, du, y, t, dir, 0, 1.0);
}

extern "C" int x3d_tds_solve_lincomb(x3d_backend *b, int dir, real_t *du, const x3d_tdsops *t, real_t *y,
                                     const real_t *base, int nterm, const real_t *c, const real_t *const *x)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    return tds_solve_lincomb_wall(b, dir, du, t, y, base, nterm, c, x, nullptr);
}

// the same with the y faces of y (vertex rows j = 0 and ny - 1) stamped from `wall` before the operator acts:
// lincomb ; field_set_face_from_field(y, wall, Y_FACE) ; tds_solve -- RK stage, the channel case's apply_BC
// (src/case/channel.f90:214-231) and the first x operator of divergence_v2c in one kernel where the pencils allow
extern "C" int x3d_tds_solve_lincomb_wall(x3d_backend *b, int dir, real_t *du, const x3d_tdsops *t, real_t *y,
                                          const real_t *base, int nterm, const real_t *c, const real_t *const *x,
                                          const real_t *wall)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(wall, "x3d_tds_solve_lincomb_wall: null argument");
    return tds_solve_lincomb_wall(b, dir, du, t, y, base, nterm, c, x, wall);
}

// x3d_tds_solve(du, u, t, dir) followed by x3d_field_mean_shift(u, dims, ncell, target, shift) -- the first x operator of
// the divergence on the new u and the bulk-velocity integral the channel case's next define_BC needs: the 1024-row kernel
// sums u's rows while it holds them (k_xwide_tds); elsewhere the two calls one after the other
int x3d_xwide_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int acc, real_t scale, bool *done,
                  real_t *psum, int ny_sum, int *nsum);
int x3d_finish_mean_shift(x3d_backend *b, int nparts, real_t ncell, real_t target, const real_t **shift);
extern "C" int x3d_tds_solve_mean(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir, const int dims[3],
                                  real_t ncell, real_t target, const real_t **shift)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && u && t && dims && shift && ncell > 0.0, "x3d_tds_solve_mean: bad argument");
    X3D_REQUIRE(x3d_dir_ok(dir) && du != u, "x3d_tds_solve_mean: bad dir / du aliases u");
    if (int rc = check_len(b, t, dir, "tds_solve_mean")) return rc;
    if (dir == X3D_DIR_X && dims[0] == b->nx && dims[2] == b->nz && dims[1] <= b->ny && dims[1] > 0) {
        bool done = false;
        int nsum = 0;
        if (int rc = x3d_xwide_tds(b, du, u, t, 0, 1.0, &done, b->red_buf, dims[1], &nsum)) return rc;
        if (done && nsum > 0) return x3d_finish_mean_shift(b, nsum, ncell, target, shift);
        if (done) return x3d_field_mean_shift(b, u, dims, ncell, target, shift);
    }
    if (int rc = x3d_tds_solve_acc(b, du, u, t, dir, 0, 1.0)) return rc;
    return x3d_field_mean_shift(b, u, dims, ncell, target, shift);
}

// x3d_tds_solve_lincomb[_wall] (wall may be NULL) followed by x3d_field_mean_shift(y, dims, ncell, target, shift): the
// channel case's RK stage + apply_BC + first x operator of the divergence, and the bulk-velocity integral its NEXT
// define_BC needs (src/case/channel.f90:66-72), which the 1024-row kernel forms while y's rows are in its registers
// (k_xwide_tds_lin) -- one reduction pass over u less per sub-step.  Elsewhere the two calls one after the other.
extern "C" int x3d_tds_solve_lincomb_wall_mean(x3d_backend *b, int dir, real_t *du, const x3d_tdsops *t, real_t *y,
                                               const real_t *base, int nterm, const real_t *c, const real_t *const *x,
                                               const real_t *wall, const int dims[3], real_t ncell, real_t target,
                                               const real_t **shift)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(dims && shift && ncell > 0.0, "x3d_tds_solve_lincomb_wall_mean: bad argument");
    int nsum = 0;
    if (int rc = tds_solve_lincomb_wall(b, dir, du, t, y, base, nterm, c, x, wall, dims, &nsum)) return rc;
    if (nsum > 0) return x3d_finish_mean_shift(b, nsum, ncell, target, shift);
    return x3d_field_mean_shift(b, y, dims, ncell, target, shift);
}

extern "C" int x3d_tds_solve(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir)
{
    X3D_RANGE(__func__);
    if (b && x3d_lazy_active(b)) {  // recorded -- after the checks the eager path makes at the call site
        X3D_REQUIRE(du && u && t, "x3d_tds_solve: null argument");
        X3D_REQUIRE(x3d_dir_ok(dir), "x3d_tds_solve: bad dir %d", dir);
        X3D_REQUIRE(du != u, "x3d_tds_solve: du and u must be distinct blocks");
        if (int rc = check_len(b, t, dir, "tds_solve")) return rc;
        return x3d_lazy_tds(b, du, u, t, dir);
    }
    return x3d_tds_solve_acc(b, du, u, t, dir, 0, 1.0);
}

int x3d_generic_tds_local(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir, int acc,
                          real_t scale)
{
    PencilGeom g = x3d_geom(b, dir);
    real_t *d = b->scratch[2];
    {
        ProfScope ps(b, X3D_K_TDS_FWD, dir);
        hipLaunchKernelGGL((k_tds_fwd<false>), grid_for(g), dim3(64), 0, b->stream, d, b->send_s, b->send_e, u,
                           (const real_t *)nullptr, (const real_t *)nullptr, t->tab, g, t->n_tds);
    }
    {
        ProfScope ps(b, X3D_K_TDS_BWD, dir);
        if (acc)
            hipLaunchKernelGGL(k_tds_bwd<true>, grid_for(g), dim3(64), 0, b->stream, du, (const real_t *)d,
                               b->send_s, b->send_e, b->send_s, t->tab, g, scale);
        else
            hipLaunchKernelGGL(k_tds_bwd<false>, grid_for(g), dim3(64), 0, b->stream, du, (const real_t *)d,
                               b->send_s, b->send_e, b->send_s, t->tab, g, 1.0);
    }
    X3D_HIP(hipGetLastError());
    return 0;
}

int npmax_of(const x3d_backend *b)
{
    int m = b->ny * b->nz;
    if (b->nx * b->nz > m) m = b->nx * b->nz;
    if (b->nx * b->ny > m) m = b->nx * b->ny;
    return m;
}

static int transeq_check(const x3d_backend *b, int dir, const x3d_tdsops *t1, const x3d_tdsops *t2,
                         const x3d_tdsops *t3)
{
    X3D_REQUIRE(t1->n_tds == t2->n_tds && t1->n_tds == t3->n_tds && t1->n_rhs == t1->n_tds &&
                    t2->n_rhs == t2->n_tds && t3->n_rhs == t3->n_tds,
                "transeq: the three operators must share n_tds == n_rhs");
    return check_len(b, t1, dir, "transeq");
}

extern "C" int x3d_transeq_dist_fwd(x3d_backend *b, int dir, real_t *rhs, real_t *send_s, real_t *send_e,
                                    const real_t *u, const real_t *u_recv_s, const real_t *u_recv_e,
                                    const real_t *conv, const real_t *conv_recv_s, const real_t *conv_recv_e,
                                    const x3d_tdsops *t_du, const x3d_tdsops *t_dud, const x3d_tdsops *t_d2u)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && rhs && send_s && send_e && u && u_recv_s && u_recv_e && conv && conv_recv_s &&
                    conv_recv_e && t_du && t_dud && t_d2u,
                "x3d_transeq_dist_fwd: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_transeq_dist_fwd: bad dir %d", dir);
    X3D_LAZY_IN(b, u);
    X3D_LAZY_IN(b, conv);
    X3D_LAZY_EAGER(b);
    if (int rc = transeq_check(b, dir, t_du, t_dud, t_d2u)) return rc;
    PencilGeom g = x3d_geom(b, dir);
    // [3][npencil] boundary buffers are contiguous with stride np
    ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
    (void)rhs;
    hipLaunchKernelGGL((k_transeq_fwd<true, false>), grid_for(g), dim3(64), 0, b->stream, b->scratch[2],
                       b->scratch[0], b->scratch[1], send_s, send_e, u, u_recv_s, u_recv_e, conv, conv_recv_s, conv_recv_e,
                       t_du->tab, t_dud->tab, t_d2u->tab, g, t_du->n_tds, g.np);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_transeq_dist_bwd_acc(x3d_backend *b, int dir, real_t *rhs, const real_t *send_s,
                                        const real_t *recv_s, const real_t *recv_e, const real_t *conv, real_t nu,
                                        const x3d_tdsops *t_du, const x3d_tdsops *t_dud, const x3d_tdsops *t_d2u,
                                        int accumulate)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && rhs && send_s && recv_s && recv_e && conv && t_du && t_dud && t_d2u,
                "x3d_transeq_dist_bwd: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_transeq_dist_bwd: bad dir %d", dir);
    X3D_LAZY_IN(b, conv);
    X3D_LAZY_OUT(b, rhs, !accumulate);
    X3D_LAZY_EAGER(b);
    PencilGeom g = x3d_geom(b, dir);
    ProfScope ps(b, X3D_K_TRANSEQ_BWD, dir);
    if (accumulate)  // fusion extension: rhs += result
        hipLaunchKernelGGL(k_transeq_bwd<true>, grid_for(g), dim3(64), 0, b->stream, rhs,
                           (const real_t *)b->scratch[2], b->scratch[0], b->scratch[1], conv, send_s, recv_s, recv_e, nu,
                           t_du->tab, t_dud->tab, t_d2u->tab, g, g.np);
    else
        hipLaunchKernelGGL(k_transeq_bwd<false>, grid_for(g), dim3(64), 0, b->stream, rhs,
                           (const real_t *)b->scratch[2], b->scratch[0], b->scratch[1], conv, send_s, recv_s, recv_e, nu,
                           t_du->tab, t_dud->tab, t_d2u->tab, g, g.np);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_transeq_dist_bwd(x3d_backend *b, int dir, real_t *rhs, const real_t *send_s,
                                    const real_t *recv_s, const real_t *recv_e, const real_t *conv, real_t nu,
                                    const x3d_tdsops *t_du, const x3d_tdsops *t_dud, const x3d_tdsops *t_d2u)
{
    X3D_RANGE(__func__);
    return x3d_transeq_dist_bwd_acc(b, dir, rhs, send_s, recv_s, recv_e, conv, nu, t_du, t_dud, t_d2u, 0);
}

static int transeq_component_local(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv,
                                   real_t nu, const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3,
                                   int acc)
{
    if (dir == X3D_DIR_X) return x3d_xdir_transeq(b, rhs, u, conv, nu, t1, t2, t3, acc);
    {
        // K3y (xscan.hip): periodic 256 / 512-row pencils straight from the Cartesian block
        bool done = false;
        if (int rc = x3d_ytile_transeq(b, dir, rhs, u, conv, nu, t1, t2, t3, acc, &done)) return rc;
        if (done) return 0;
    }
    return x3d_generic_transeq_local(b, dir, rhs, u, conv, nu, t1, t2, t3, acc);
}

int x3d_generic_transeq_local(x3d_backend *b, int dir, real_t *rhs, const real_t *u, const real_t *conv,
                              real_t nu, const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3,
                              int acc)
{
    PencilGeom g = x3d_geom(b, dir);
    real_t *d1 = b->scratch[2];
    const int npm = npmax_of(b);
    const real_t *z = nullptr;
    {
        ProfScope ps(b, X3D_K_TRANSEQ_FWD, dir);
        if (u == conv)
            hipLaunchKernelGGL((k_transeq_fwd<false, true>), grid_for(g), dim3(64), 0, b->stream, d1,
                               b->scratch[0], b->scratch[1], b->send_s, b->send_e, u, z, z, conv, z, z, t1->tab,
                               t2->tab, t3->tab, g, t1->n_tds, npm);
        else
            hipLaunchKernelGGL((k_transeq_fwd<false, false>), grid_for(g), dim3(64), 0, b->stream, d1,
                               b->scratch[0], b->scratch[1], b->send_s, b->send_e, u, z, z, conv, z, z, t1->tab,
                               t2->tab, t3->tab, g, t1->n_tds, npm);
    }
    {
        ProfScope ps(b, X3D_K_TRANSEQ_BWD, dir);
        if (acc)
            hipLaunchKernelGGL(k_transeq_bwd<true>, grid_for(g), dim3(64), 0, b->stream, rhs, (const real_t *)d1,
                               b->scratch[0], b->scratch[1], conv, b->send_s, b->send_e, b->send_s, nu, t1->tab,
                               t2->tab, t3->tab, g, npm);
        else
            hipLaunchKernelGGL(k_transeq_bwd<false>, grid_for(g), dim3(64), 0, b->stream, rhs, (const real_t *)d1,
                               b->scratch[0], b->scratch[1], conv, b->send_s, b->send_e, b->send_s, nu, t1->tab,
                               t2->tab, t3->tab, g, npm);
    }
    X3D_HIP(hipGetLastError());
    return 0;
}

// transeq_x/y/z -> transeq_omp_dist with the component permutation of
// src/backend/omp/backend.f90:145-184 and the operator pairing of :246-260
extern "C" int x3d_transeq(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u,
                           const real_t *v, const real_t *w, real_t nu, const x3d_tdsops *der1st,
                           const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                           const x3d_tdsops *der2nd_sym)
{
    X3D_RANGE(__func__);
    if (b && x3d_lazy_active(b)) {  // recorded -- after the checks the eager path makes at the call site
        X3D_REQUIRE(du && dv && dw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym, "x3d_transeq: null argument");
        X3D_REQUIRE(x3d_dir_ok(dir), "x3d_transeq: bad dir %d", dir);
        if (int rc = transeq_check(b, dir, der1st, der1st_sym, der2nd)) return rc;
        if (int rc = transeq_check(b, dir, der1st_sym, der1st, der2nd_sym)) return rc;
        X3D_REQUIRE(du != u && du != v && du != w && dv != u && dv != v && dv != w && dw != u && dw != v && dw != w,
                    "x3d_transeq: outputs alias inputs");
        return x3d_lazy_transeq(b, dir, du, dv, dw, u, v, w, nu, der1st, der1st_sym, der2nd, der2nd_sym);
    }
    return x3d_transeq_acc(b, dir, du, dv, dw, u, v, w, nu, der1st, der1st_sym, der2nd, der2nd_sym, 0);
}

extern "C" int x3d_transeq_acc(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u,
                               const real_t *v, const real_t *w, real_t nu, const x3d_tdsops *der1st,
                               const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd,
                               const x3d_tdsops *der2nd_sym, int accumulate)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && dv && dw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym,
                "x3d_transeq: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_transeq: bad dir %d", dir);
    if (int rc = transeq_check(b, dir, der1st, der1st_sym, der2nd)) return rc;
    if (int rc = transeq_check(b, dir, der1st_sym, der1st, der2nd_sym)) return rc;
    real_t *r[3];
    const real_t *f[3];
    if (dir == X3D_DIR_X) { r[0] = du; r[1] = dv; r[2] = dw; f[0] = u; f[1] = v; f[2] = w; }
    else if (dir == X3D_DIR_Y) { r[0] = dv; r[1] = du; r[2] = dw; f[0] = v; f[1] = u; f[2] = w; }
    else { r[0] = dw; r[1] = du; r[2] = dv; f[0] = w; f[1] = u; f[2] = v; }
    for (int c = 0; c < 3; c++) {
        X3D_REQUIRE(r[c] != f[0] && r[c] != f[1] && r[c] != f[2], "x3d_transeq: outputs alias inputs");
    }
    const int a = accumulate;
    if (dir == X3D_DIR_X) {
        // the three components in one launch of the scan kernel (xscan.hip), the advecting velocity read once
        bool done = false;
        if (int rc = x3d_xscan_transeq3(b, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, a, nullptr, nullptr, nullptr,
                                        0.0, 0.0, nullptr, &done))
            return rc;
        if (done) return 0;
        if (int rc = x3d_xwide_transeq3(b, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, a, 0.0, nullptr, &done)) return rc;  // n = 1024
        if (done) return 0;
    }
    if (dir != X3D_DIR_X) {
        // K3t (viax.hip): periodic pencils of 256 / 512 rows go through the single-pass scan kernel
        bool done = false;
        if (int rc = x3d_transeq_via_x(b, dir, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, a, &done)) return rc;
        if (done) return 0;
    }
    if (dir != X3D_DIR_X) {
        // K3g (ygen.hip): non-periodic / odd-length pencils, the three components in one launch
        bool done = false;
        if (int rc = x3d_ygen_transeq3(b, dir, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, a, &done)) return rc;
        if (done) return 0;
    }
    if (int rc = transeq_component_local(b, dir, r[0], f[0], f[0], nu, der1st, der1st_sym, der2nd, a)) return rc;
    if (int rc = transeq_component_local(b, dir, r[1], f[1], f[0], nu, der1st_sym, der1st, der2nd_sym, a)) return rc;
    if (int rc = transeq_component_local(b, dir, r[2], f[2], f[0], nu, der1st_sym, der1st, der2nd_sym, a)) return rc;
    return 0;
}

// fusion extension: transeq_x (all three components, du dv dw written) on a velocity whose pressure-gradient
// correction is still pending: first u += scale * tds_solve(gu, op_u), v += scale * tds_solve(gv, op_vw),
// w += scale * tds_solve(gw, op_vw) (the last x operators of gradient_c2v + the velocity update of
// src/solver.f90:731-733), per pencil, inside the transeq kernel.  *done = 0: not applicable here, nothing was
// done (issue x3d_tds_solve_acc x 3 and x3d_transeq).  Bit-identical to that sequence.
extern "C" int x3d_transeq_x_update(x3d_backend *b, real_t *du, real_t *dv, real_t *dw, real_t *u, real_t *v, real_t *w,
                                    real_t nu, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                                    const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym, const real_t *gu,
                                    const real_t *gv, const real_t *gw, const x3d_tdsops *op_u,
                                    const x3d_tdsops *op_vw, real_t scale, int *done)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && dv && dw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym && gu && gv && gw &&
                    op_u && op_vw && done,
                "x3d_transeq_x_update: null argument");
    *done = 0;
    if (int rc = transeq_check(b, X3D_DIR_X, der1st, der1st_sym, der2nd)) return rc;
    if (int rc = check_len(b, op_u, X3D_DIR_X, "transeq_x_update")) return rc;
    if (int rc = check_len(b, op_vw, X3D_DIR_X, "transeq_x_update")) return rc;
    real_t *r[3] = {du, dv, dw};
    const real_t *f[3] = {u, v, w}, *g[3] = {gu, gv, gw};
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++)
            X3D_REQUIRE(r[c] != f[k] && r[c] != g[k] && f[c] != g[k], "x3d_transeq_x_update: arguments alias");
    bool ok = false;
    if (int rc = x3d_xscan_transeq3(b, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, 0, g, op_u, op_vw, scale, 0.0,
                                    nullptr, &ok))
        return rc;
    if (!ok) {  // 1024-row pencils (K3w)
        real_t *fw[3] = {u, v, w};
        if (int rc = x3d_xwide_transeq3_upd(b, r, fw, nu, der1st, der1st_sym, der2nd, der2nd_sym, g, op_u, op_vw, scale, 0.0,
                                            nullptr, &ok))
            return rc;
    }
    *done = ok ? 1 : 0;
    return 0;
}

// fusion extension (round 6): the two above in one launch -- pending correction, bulk-velocity shift, transeq_x, rotation
// forcing (include/x3d2_hip.h).  Served for 1024-row pencils (K3w); omega == 0 and no shift: x3d_transeq_x_update.
extern "C" int x3d_transeq_x_update_rot(x3d_backend *b, real_t *du, real_t *dv, real_t *dw, real_t *u, real_t *v, real_t *w,
                                        real_t nu, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                                        const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym, const real_t *gu,
                                        const real_t *gv, const real_t *gw, const x3d_tdsops *op_u,
                                        const x3d_tdsops *op_vw, real_t scale, real_t omega, const real_t *u_shift, int *done)
{
    X3D_RANGE(__func__);
    if (omega == 0.0 && !u_shift)
        return x3d_transeq_x_update(b, du, dv, dw, u, v, w, nu, der1st, der1st_sym, der2nd, der2nd_sym, gu, gv, gw, op_u, op_vw,
                                    scale, done);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && dv && dw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym && gu && gv && gw &&
                    op_u && op_vw && done,
                "x3d_transeq_x_update_rot: null argument");
    *done = 0;
    if (int rc = transeq_check(b, X3D_DIR_X, der1st, der1st_sym, der2nd)) return rc;
    if (int rc = transeq_check(b, X3D_DIR_X, der1st_sym, der1st, der2nd_sym)) return rc;
    if (int rc = check_len(b, op_u, X3D_DIR_X, "transeq_x_update_rot")) return rc;
    if (int rc = check_len(b, op_vw, X3D_DIR_X, "transeq_x_update_rot")) return rc;
    real_t *r[3] = {du, dv, dw}, *f[3] = {u, v, w};
    const real_t *g[3] = {gu, gv, gw};
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++)
            X3D_REQUIRE(r[c] != f[k] && r[c] != g[k] && f[c] != g[k], "x3d_transeq_x_update_rot: arguments alias");
    bool ok = false;
    if (int rc = x3d_xwide_transeq3_upd(b, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, g, op_u, op_vw, scale, omega,
                                        u_shift, &ok))
        return rc;
    *done = ok ? 1 : 0;
    return 0;
}

// fusion extension: transeq_x (du, dv, dw written) with the channel case's rotation forcing on top
// (src/case/channel.f90:191-207: du = du - omega v, dv = dv + omega u -- there two vecadd's after transeq; here
// applied to the x contribution, so the y / z contributions are added to the forced values: the same sum in
// another order, round-off level).  u_shift != null: first u += *u_shift in place (device scalar of
// x3d_field_mean_shift: the second half of the bulk-velocity correction, :70-77, bit-identical to x3d_field_shift_by).
// *done = 0: not served for these pencils, nothing was done (issue x3d_field_shift_by, x3d_transeq and, after the
// other directions, x3d_vecadd x 2).  Served: periodic x pencils of 256 / 512 rows (K3s) and of 1024 rows (K3w).
extern "C" int x3d_transeq_x_rot(x3d_backend *b, real_t *du, real_t *dv, real_t *dw, real_t *u, const real_t *v,
                                 const real_t *w, real_t nu, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                                 const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym, real_t omega,
                                 const real_t *u_shift, int *done)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && du && dv && dw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym && done,
                "x3d_transeq_x_rot: null argument");
    *done = 0;
    if (int rc = transeq_check(b, X3D_DIR_X, der1st, der1st_sym, der2nd)) return rc;
    if (int rc = transeq_check(b, X3D_DIR_X, der1st_sym, der1st, der2nd_sym)) return rc;
    real_t *r[3] = {du, dv, dw};
    const real_t *f[3] = {u, v, w};
    for (int c = 0; c < 3; c++)
        X3D_REQUIRE(r[c] != f[0] && r[c] != f[1] && r[c] != f[2], "x3d_transeq_x_rot: outputs alias inputs");
    if (omega == 0.0 && !u_shift) return 0;
    bool ok = false;
    if (int rc = x3d_xscan_transeq3(b, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, 0, nullptr, nullptr, nullptr, 0.0,
                                    omega, u_shift, &ok))
        return rc;  // 256 / 512-row pencils
    if (!ok)
        if (int rc = x3d_xwide_transeq3(b, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, 0, omega, u_shift, &ok)) return rc;
    *done = ok ? 1 : 0;
    return 0;
}

// transeq_species (src/backend/omp/backend.f90:186-233): one convection-diffusion component of a transported
// scalar: field = spec, advecting velocity = uvw, operators (der1st, der1st_sym, der2nd), local direction
extern "C" int x3d_transeq_species(x3d_backend *b, int dir, real_t *dspec, const real_t *uvw, const real_t *spec,
                                   real_t nu, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                                   const x3d_tdsops *der2nd, int accumulate)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && dspec && uvw && spec && der1st && der1st_sym && der2nd, "x3d_transeq_species: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_transeq_species: bad dir %d", dir);
    X3D_REQUIRE(dspec != uvw && dspec != spec, "x3d_transeq_species: output aliases an input");
    if (int rc = transeq_check(b, dir, der1st, der1st_sym, der2nd)) return rc;
    if (x3d_lazy_active(b)) return x3d_lazy_species(b, dir, dspec, uvw, spec, nu, der1st, der1st_sym, der2nd, accumulate);
    return transeq_component_local(b, dir, dspec, spec, uvw, nu, der1st, der1st_sym, der2nd, accumulate);
}


// ------------------------------------------------------------------ decomposed directions, single pass
// exec_dist_tds_compact / exec_dist_transeq_compact (src/backend/omp/exec_dist.f90:16-186) do: sweep -> exchange
// the boundary values -> substitution sweep.  Here the tile kernels (xscan.hip, HALO forms) do the whole local
// solve in one pass with the neighbours' boundary values taken as zero and hand out their own; after the
// exchange x3d_*_halo_fix adds what the received values contribute, on the boundary strips only.  Same linear
// system; the result differs from the three-sweep order by re-association and by coupling terms < 2^-60.

// send[side 2][field nf][4][np]: rows 1..4 (side 0, for prev) and n-3..n (side 1, for next) of nf <= 3 fields
extern "C" int x3d_pack_halos_multi(x3d_backend *b, real_t *send, const real_t *const *fields, int nf, int n, int dir)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && send && fields, "x3d_pack_halos_multi: null argument");
    X3D_REQUIRE(nf >= 1 && nf <= 3, "x3d_pack_halos_multi: 1..3 fields");
    X3D_REQUIRE(x3d_dir_ok(dir), "x3d_pack_halos_multi: bad dir %d", dir);
    PencilGeom g = x3d_geom(b, dir);
    PackFields pf{};
    for (int k = 0; k < nf; k++) {
        X3D_REQUIRE(fields[k], "x3d_pack_halos_multi: null field");
        const real_t *fk = fields[k];
        X3D_LAZY_IN(b, fk);  // (deferred execution: flush, then the buffer that holds the field)
        pf.f[k] = fk;
    }
    X3D_LAZY_EAGER(b);
    ProfScope ps(b, X3D_K_PACK, dir);
    int hp;
    long hnp;
    x3d_halo_layout(b, dir, &hp, &hnp);
    hipLaunchKernelGGL(k_pack_halos_multi, dim3((g.np + 255) / 256, nf), dim3(256), 0, b->stream, send, pf, nf, n, g, hp, hnp);
    X3D_HIP(hipGetLastError());
    return 0;
}

// transeq_<dir> (all three components) of a y or z direction through the tile kernel, over the planes
// [other0, other0 + nother) of the coordinate the tiles are stacked along (z for y pencils, y for z pencils;
// nother < 0: all) -- a plane range lets the caller overlap a neighbour exchange with the rest.
// halo_recv == NULL: local (periodic) direction.  Otherwise the direction is decomposed: halo_recv[2][3][4][np]
// holds the neighbours' rows of the fields in COMPONENT order (dir y: v, u, w; dir z: w, u, v: the advecting one
// first, src/backend/omp/backend.f90:145-184) and bnd_send[2][9][np] receives this rank's boundary values
// ([component * 3 + {d(u conv), du, d2u}]).  *done = 0: these pencils are not served here (use x3d_transeq_acc /
// the two-sweep x3d_transeq_dist_* calls).
extern "C" int x3d_transeq_tile(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u,
                                const real_t *v, const real_t *w, real_t nu, const x3d_tdsops *der1st,
                                const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                                int accumulate, const real_t *halo_recv, real_t *bnd_send, int other0, int nother,
                                int *done)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && du && dv && dw && u && v && w && der1st && der1st_sym && der2nd && der2nd_sym && done,
                "x3d_transeq_tile: null argument");
    X3D_REQUIRE((halo_recv == nullptr) == (bnd_send == nullptr), "x3d_transeq_tile: halo_recv and bnd_send go together");
    *done = 0;
    X3D_REQUIRE(du != u && du != v && du != w && dv != u && dv != v && dv != w && dw != u && dw != v && dw != w,
                "x3d_transeq_tile: outputs alias inputs");
    if (nother != 0) {  // (a launch over zero planes is a probe: nothing is read or written)
        X3D_LAZY_IN(b, u); X3D_LAZY_IN(b, v); X3D_LAZY_IN(b, w);
        const bool full = !accumulate && nother < 0;  // (a plane range leaves the other planes as they are)
        X3D_LAZY_OUT(b, du, full); X3D_LAZY_OUT(b, dv, full); X3D_LAZY_OUT(b, dw, full);
    }
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(dir == X3D_DIR_Y || dir == X3D_DIR_Z, "x3d_transeq_tile: dir must be y or z");
    if (int rc = transeq_check(b, dir, der1st, der1st_sym, der2nd)) return rc;
    if (int rc = transeq_check(b, dir, der1st_sym, der1st, der2nd_sym)) return rc;
    real_t *r[3];
    const real_t *f[3];
    if (dir == X3D_DIR_Y) { r[0] = dv; r[1] = du; r[2] = dw; f[0] = v; f[1] = u; f[2] = w; }
    else { r[0] = dw; r[1] = du; r[2] = dv; f[0] = w; f[1] = u; f[2] = v; }
    for (int c = 0; c < 3; c++) X3D_REQUIRE(r[c] != f[0] && r[c] != f[1] && r[c] != f[2], "x3d_transeq_tile: outputs alias inputs");
    TileHalo th{halo_recv, bnd_send, x3d_geom(b, dir).np, 3, 9, 0, 0};
    x3d_halo_layout(b, dir, &th.hp, &th.hnp);
    bool ok = false;
    if (int rc = x3d_ytile_transeq3(b, dir, r, f, nu, der1st, der1st_sym, der2nd, der2nd_sym, accumulate,
                                    halo_recv ? &th : nullptr, other0, nother, &ok))
        return rc;
    *done = ok ? 1 : 0;
    return 0;
}

// ... and the contribution of the received boundary values bnd_recv[2][9][np] (side 0: from prev = its X_n,
// side 1: from next = its du_1), added to du, dv, dw
extern "C" int x3d_transeq_halo_fix(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u,
                                    const real_t *v, const real_t *w, real_t nu, const x3d_tdsops *der1st,
                                    const x3d_tdsops *der2nd, const real_t *bnd_recv)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && du && dv && dw && u && v && w && der1st && der2nd && bnd_recv, "x3d_transeq_halo_fix: null argument");
    X3D_REQUIRE(dir == X3D_DIR_Y || dir == X3D_DIR_Z, "x3d_transeq_halo_fix: dir must be y or z");
    X3D_LAZY_IN(b, u); X3D_LAZY_IN(b, v); X3D_LAZY_IN(b, w);
    X3D_LAZY_OUT(b, du, false); X3D_LAZY_OUT(b, dv, false); X3D_LAZY_OUT(b, dw, false);
    X3D_LAZY_EAGER(b);
    real_t *r[3];
    if (dir == X3D_DIR_Y) { r[0] = dv; r[1] = du; r[2] = dw; }
    else { r[0] = dw; r[1] = du; r[2] = dv; }
    return x3d_transeq_halo_fix_launch(b, dir, r, dir == X3D_DIR_Y ? v : w, nu, bnd_recv, der1st, der2nd);
}

// tds_solve pairs / single operators of a y or z direction through the tile kernel (modes of x3d_tds_solve_pair,
// + mode 2: out1 = A(in1)), over a plane range as above.  halo_recv == NULL: local direction.  Otherwise
// halo_recv[2][nf][4][np] with nf = 2 (mode 0: in1, in2) or 1, and bnd_send[2][nb][np], nb = 2 (modes 0, 1: A, B) or 1
static int pair_tile(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *in1,
                     const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb, const real_t *halo_recv,
                     real_t *bnd_send, int other0, int nother, int *done)
{
    X3D_REQUIRE(b && out1 && in1 && ta && done, "x3d_tds_pair_tile: null argument");
    X3D_REQUIRE((halo_recv == nullptr) == (bnd_send == nullptr), "x3d_tds_pair_tile: halo_recv and bnd_send go together");
    *done = 0;
    X3D_REQUIRE(mode >= 0 && mode <= 2, "x3d_tds_pair_tile: mode must be 0, 1 or 2");
    X3D_REQUIRE((mode == 0 ? in2 != nullptr : true) && (mode == 1 ? out2 != nullptr : true) && (mode == 2 || tb),
                "x3d_tds_pair_tile: null argument");
    X3D_REQUIRE(dir == X3D_DIR_Y || dir == X3D_DIR_Z, "x3d_tds_pair_tile: dir must be y or z");
    X3D_REQUIRE(out1 != in1 && out1 != in2 && out2 != in1 && (mode != 1 || out1 != out2),
                "x3d_tds_pair_tile: outputs alias inputs");
    if (nother != 0) {  // (zero planes: a probe)
        X3D_LAZY_IN(b, in1);
        if (mode == 0) X3D_LAZY_IN(b, in2);
        X3D_LAZY_OUT(b, out1, nother < 0);
        if (mode == 1) X3D_LAZY_OUT(b, out2, nother < 0);
    }
    X3D_LAZY_EAGER(b);
    if (mode == 2) tb = ta;
    if (int rc = check_len(b, ta, dir, "tds_pair_tile")) return rc;
    if (int rc = check_len(b, tb, dir, "tds_pair_tile")) return rc;
    TileHalo th{halo_recv, bnd_send, x3d_geom(b, dir).np, mode == 0 ? 2 : 1, mode == 2 ? 1 : 2, 0, 0};
    x3d_halo_layout(b, dir, &th.hp, &th.hnp);
    bool ok = false;
    if (int rc = x3d_ytile_tds_pair(b, dir, mode, out1, out2, in1, in2, ta, tb, halo_recv ? &th : nullptr, other0,
                                    nother, &ok))
        return rc;
    if (!ok && !halo_recv && other0 == 0 && (nother < 0 || nother == (dir == X3D_DIR_Y ? b->nz : b->ny))) {
        if (int rc = x3d_ygen_pair(b, dir, mode, out1, out2, in1, in2, ta, tb, &ok)) return rc;  // K3g, whole block only
    }
    *done = ok ? 1 : 0;
    return 0;
}

extern "C" int x3d_tds_pair_tile(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2, const real_t *in1,
                                 const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb, const real_t *halo_recv,
                                 real_t *bnd_send, int other0, int nother, int *done)
{
    X3D_RANGE(__func__);
    return pair_tile(b, dir, mode, out1, out2, in1, in2, ta, tb, halo_recv, bnd_send, other0, nother, done);
}
// the decomposed-z form of x3d_tds_solve_pair_yperm: the whole block, halo_recv's planes in the row order of in1
// (mode 1: interleaved like the field they were cut from), bnd_send in pencil order
extern "C" int x3d_tds_pair_tile_yperm(x3d_backend *b, int mode, real_t *out1, real_t *out2, const real_t *in1,
                                       const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb,
                                       const real_t *halo_recv, real_t *bnd_send, int ny, int *done)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && done, "x3d_tds_pair_tile_yperm: null argument");
    X3D_REQUIRE(mode == 0 || mode == 1, "x3d_tds_pair_tile_yperm: mode must be 0 or 1");
    X3D_REQUIRE(ny > 0 && ny <= b->ny, "x3d_tds_pair_tile_yperm: ny = %d outside the block (%d rows)", ny, b->ny);
    X3D_REQUIRE(halo_recv && bnd_send, "x3d_tds_pair_tile_yperm: halo_recv and bnd_send are required");
    b->pair_yperm = ny;
    const int rc = pair_tile(b, X3D_DIR_Z, mode, out1, out2, in1, in2, ta, tb, halo_recv, bnd_send, 0, -1, done);
    b->pair_yperm = 0;
    return rc;
}
extern "C" int x3d_tds_pair_halo_fix_yperm(x3d_backend *b, int mode, real_t *out1, real_t *out2, const x3d_tdsops *ta,
                                           const x3d_tdsops *tb, const real_t *bnd_recv, int ny)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "x3d_tds_pair_halo_fix_yperm: null argument");
    X3D_REQUIRE(mode == 0 || mode == 1, "x3d_tds_pair_halo_fix_yperm: mode must be 0 or 1");
    X3D_REQUIRE(ny > 0 && ny <= b->ny, "x3d_tds_pair_halo_fix_yperm: ny = %d outside the block (%d rows)", ny, b->ny);
    b->pair_yperm = ny;
    const int rc = x3d_tds_pair_halo_fix(b, X3D_DIR_Z, mode, out1, out2, ta, tb, bnd_recv);
    b->pair_yperm = 0;
    return rc;
}
// the z pairs on either side of the z-first Poisson solve (csrc/zfirst.hip): mode 0 = the last pair of
// divergence_v2c, its result leaves as the spectrum's z-transformed planes (out1 unused); mode 1 = the first pair of
// gradient_c2v, its input arrives that way (in1 unused).  *done = 0: not on offer here, nothing was done
struct x3d_poisson;
int x3d_zfirst_arg(x3d_poisson *p, ZfArg *out, bool *ok);
int x3d_ytile_tds_pair_zf(x3d_backend *b, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                          const x3d_tdsops *ta, const x3d_tdsops *tb, const ZfArg &zf, bool *done, int y0, int nyr);
extern "C" int x3d_tds_pair_zfirst(x3d_backend *b, x3d_poisson *poisson, int mode, real_t *out1, real_t *out2,
                                   const real_t *in1, const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb,
                                   int *done)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && poisson && ta && tb && done, "x3d_tds_pair_zfirst: null argument");
    X3D_REQUIRE(mode == 0 || mode == 1, "x3d_tds_pair_zfirst: mode must be 0 or 1");
    X3D_REQUIRE(mode == 0 ? (in1 && in2) : (out1 && out2 && out1 != out2), "x3d_tds_pair_zfirst: null argument");
    *done = 0;
    X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    if (int rc = check_len(b, ta, X3D_DIR_Z, "tds_pair_zfirst")) return rc;
    if (int rc = check_len(b, tb, X3D_DIR_Z, "tds_pair_zfirst")) return rc;
    ZfArg zf{};
    bool ok = false;
    if (int rc = x3d_zfirst_arg(poisson, &zf, &ok)) return rc;
    if (!ok) return 0;
    if (int rc = x3d_ytile_tds_pair_zf(b, mode, out1, out2, in1, in2, ta, tb, zf, &ok, 0, -1)) return rc;
    *done = ok ? 1 : 0;
    return 0;
}
bool x3d_zfirst_pairs_ok(const x3d_backend *b, const x3d_tdsops *ta, const x3d_tdsops *tb);  // xscan.hip
// probe: would the z-transforming pair kernels (x3d_tds_pair_zfirst, x3d_sfftz_tds_pair) take this operator pair
extern "C" int x3d_tds_pair_zfirst_ok(x3d_backend *b, const x3d_tdsops *ta, const x3d_tdsops *tb, int *ok)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && ta && tb && ok, "x3d_tds_pair_zfirst_ok: null argument");
    *ok = x3d_zfirst_pairs_ok(b, ta, tb) ? 1 : 0;
    return 0;
}

extern "C" int x3d_tds_pair_halo_fix(x3d_backend *b, int dir, int mode, real_t *out1, real_t *out2,
                                     const x3d_tdsops *ta, const x3d_tdsops *tb, const real_t *bnd_recv)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out1 && ta && bnd_recv && (mode == 2 || tb) && (mode != 1 || out2), "x3d_tds_pair_halo_fix: null argument");
    X3D_REQUIRE(mode >= 0 && mode <= 2, "x3d_tds_pair_halo_fix: mode must be 0, 1 or 2");
    X3D_REQUIRE(dir == X3D_DIR_Y || dir == X3D_DIR_Z, "x3d_tds_pair_halo_fix: dir must be y or z");
    X3D_LAZY_OUT(b, out1, false);
    if (mode == 1) X3D_LAZY_OUT(b, out2, false);
    X3D_LAZY_EAGER(b);
    return x3d_tds_halo_fix(b, dir, mode, out1, out2, bnd_recv, ta, mode == 2 ? ta : tb);
}
