"""Host-side compact-scheme operator factory: mirror of the reference's
backend-agnostic tdsops_t / tdsops_init / dirps_t
(/root/reference/src/tdsops.f90:9-59, 63-203; schemes :205-872;
DistD2 factorisation preprocess_dist :874-931).

O(n) set-up work, done once per operator on the host in numpy; the arrays are
then handed to the device through x3d_tdsops_create, exactly as the
reference's GPU backend copies the host tdsops_t into its device extension
(src/backend/cuda/tdsops.f90:9-90)."""
import math

import numpy as np

from .common import BC_DIRICHLET, BC_NEUMANN, BC_PERIODIC, X3dError

N_HALO = 4


def _row(*c):
    return np.array(c, dtype=np.float64)


class Tdsops:
    """tdsops_t.  coeffs_s[i-1] / coeffs_e[i-1] hold the reference's
    coeffs_s(:, i) / coeffs_e(:, i)."""

    def __init__(self, n_tds, delta, operation, scheme, bc_start, bc_end, stretch=None,
                 stretch_correct=None, n_halo=None, from_to=None, sym=None, c_nu=None, nu0_nu=None):
        self.n_tds = int(n_tds)
        if from_to is not None and from_to == "v2p" and bc_end in (BC_NEUMANN, BC_DIRICHLET):
            self.n_rhs = self.n_tds + 1  # :114-123
        else:
            self.n_rhs = self.n_tds
        self.n_halo = 4 if n_halo is None else int(n_halo)
        n = self.n_rhs
        self.dist_fw, self.dist_bw = np.zeros(n), np.zeros(n)
        self.dist_sa, self.dist_sc, self.dist_af = np.zeros(n), np.zeros(n), np.zeros(n)
        self.coeffs = np.zeros(9)
        self.coeffs_s, self.coeffs_e = np.zeros((4, 9)), np.zeros((4, 9))
        self.stretch = np.ones(self.n_tds) if stretch is None else \
            np.array(stretch[:self.n_tds], dtype=np.float64)
        self.stretch_correct = np.zeros(self.n_tds) if stretch_correct is None else \
            np.array(stretch_correct[:self.n_tds], dtype=np.float64)
        self.periodic = bc_start == BC_PERIODIC and bc_end == BC_PERIODIC
        self.alpha = self.a = self.b = 0.0
        self.c = self.d = 0.0
        # pentadiagonal LHS schemes (compact10_penta): src/tdsops.f90:34-37
        self.beta, self.beta_lhs_s, self.pentadiag = 0.0, 0.0, False
        self.operation, self.scheme = operation, scheme
        symmetry = bool(sym) if sym is not None else False
        self.bc_start, self.bc_end, self.sym = int(bc_start), int(bc_end), symmetry
        if operation == "first-deriv":
            dist_b = self._deriv_1st(delta, scheme, bc_start, bc_end, symmetry)
        elif operation == "second-deriv":
            dist_b = self._deriv_2nd(delta, scheme, bc_start, bc_end, symmetry, c_nu, nu0_nu)
        elif operation == "interpolate":
            dist_b = self._interpl_mid(scheme, from_to, bc_start, bc_end)
        elif operation == "stag-deriv":
            dist_b = self._stagder_1st(delta, scheme, from_to, bc_start, bc_end)
        else:
            raise X3dError("operation is not defined")
        if self.pentadiag:
            self._preprocess_penta_dist(bc_start, bc_end, symmetry)  # :398-399
        else:
            self._preprocess_dist(dist_b)
        self.move = {"v2p": 1, "p2v": -1}.get(from_to, 0)
        self.handle = None  # device copy, set by the backend's alloc_tdsops
        tol = 1e-16
        self.truncation = float(self.dist_sa[self.n_tds - 1])  # health check of :196-201

    # -- helpers
    def _bulk(self, alpha):
        self.coeffs_s[:] = self.coeffs
        self.coeffs_e[:] = self.coeffs
        self.dist_sa[:] = alpha
        self.dist_sc[:] = alpha
        return np.ones(self.n_rhs)

    # -- src/tdsops.f90:205-405
    def _deriv_1st(self, delta, scheme, bc_start, bc_end, sym):
        if scheme == "compact6":
            alpha, afi, bfi, cfi = 1.0 / 3.0, 7.0 / 9.0 / delta, 1.0 / 36.0 / delta, 0.0
        elif scheme == "compact10_penta":
            # Lele (1992) Table 1, 10th-order pentadiagonal first derivative (:235-251):
            # beta f'_{i-2} + alpha f'_{i-1} + f'_i + alpha f'_{i+1} + beta f'_{i+2} = a.. b.. c..
            self.pentadiag = True
            alpha, self.beta = 0.5, 1.0 / 20.0
            afi, bfi, cfi = 17.0 / 24.0 / delta, 101.0 / 600.0 / delta, 1.0 / 600.0 / delta
        else:
            raise X3dError("scheme is not defined")
        self.alpha, self.a, self.b, self.c = alpha, afi, bfi, cfi
        self.coeffs[:] = _row(0.0, -cfi, -bfi, -afi, 0.0, afi, bfi, cfi, 0.0)
        dist_b = self._bulk(alpha)
        n, cs, ce, sa, sc = self.n_tds, self.coeffs_s, self.coeffs_e, self.dist_sa, self.dist_sc
        if self.pentadiag:
            # the start / end stencils stay the interior one except for the compact one-sided Dirichlet closures
            # (4th order, same alpha, beta; :322-334, 383-395); Neumann: mirror ghosts come with the halos
            sa[:] = 0.0
            sc[:] = 0.0
            if bc_start == BC_DIRICHLET:
                cs[0] = _row(0, 0, 0, 0, -529.0 / 240.0, 71.0 / 20.0, -9.0 / 4.0, 67.0 / 60.0, -17.0 / 80.0) / delta
                cs[1] = _row(0, 0, 0, -301.0 / 240.0, 103.0 / 120.0, -3.0 / 40.0, 13.0 / 24.0, -17.0 / 240.0, 0) / delta
            if bc_end == BC_DIRICHLET:
                ce[3] = _row(17.0 / 80.0, -67.0 / 60.0, 9.0 / 4.0, -71.0 / 20.0, 529.0 / 240.0, 0, 0, 0, 0) / delta
                ce[2] = _row(0, 17.0 / 240.0, -13.0 / 24.0, 3.0 / 40.0, -103.0 / 120.0, 301.0 / 240.0, 0, 0, 0) / delta
            return dist_b
        if bc_start == BC_NEUMANN:
            if sym:
                sa[0], sc[0] = 0.0, 0.0
                cs[0] = 0.0
                cs[1] = _row(0, 0, 0, -afi, -bfi, afi, bfi, 0, 0)
            else:
                sa[0], sc[0] = 0.0, 2 * alpha
                cs[0] = _row(0, 0, 0, 0, 0, 2 * afi, 2 * bfi, 0, 0)
                cs[1] = _row(0, 0, 0, -afi, bfi, afi, bfi, 0, 0)
        elif bc_start == BC_DIRICHLET:
            sa[0], sc[0] = 0.0, 2.0
            cs[0] = _row(0, 0, 0, 0, -2.5, 2.0, 0.5, 0, 0) / delta
            sa[1], sc[1] = 0.25, 0.25
            cs[1] = _row(0, 0, 0, -0.75, 0, 0.75, 0, 0, 0) / delta
        if bc_end == BC_NEUMANN:
            if sym:
                sa[n - 1], sc[n - 1] = 0.0, 0.0
                ce[3] = 0.0
                ce[2] = _row(0, 0, -bfi, -afi, bfi, afi, 0, 0, 0)
            else:
                sa[n - 1], sc[n - 1] = 2 * alpha, 0.0
                ce[3] = _row(0, 0, -2 * bfi, -2 * afi, 0, 0, 0, 0, 0)
                ce[2] = _row(0, 0, -bfi, -afi, -bfi, afi, 0, 0, 0)
        elif bc_end == BC_DIRICHLET:
            sa[n - 1], sc[n - 1] = 2.0, 0.0
            ce[3] = _row(0, 0, -0.5, -2.0, 2.5, 0, 0, 0, 0) / delta
            sa[n - 2], sc[n - 2] = 0.25, 0.25
            ce[2] = _row(0, 0, 0, -0.75, 0, 0.75, 0, 0, 0) / delta
        return dist_b

    # -- src/tdsops.f90:407-618
    def _deriv_2nd(self, delta, scheme, bc_start, bc_end, sym, c_nu, nu0_nu):
        d2 = delta * delta
        if scheme == "compact6":
            alpha, asi, bsi, csi, dsi = 2.0 / 11.0, 12.0 / 11.0 / d2, 3.0 / 44.0 / d2, 0.0, 0.0
        elif scheme == "compact6-hyperviscous":
            if c_nu is None or nu0_nu is None:
                raise X3dError("compact6-hyperviscous requires c_nu and nu0_nu")
            pi = 4.0 * math.atan(1.0)
            dpis3 = 2.0 * pi / 3.0
            xnpi2 = pi * pi * (1.0 + nu0_nu)
            xmpi2 = dpis3 * dpis3 * (1.0 + c_nu * nu0_nu)
            den = 405.0 * xnpi2 - 640.0 * xmpi2 + 144.0
            alpha = 0.5 - (320.0 * xmpi2 - 1296.0) / den
            asi = -(4329.0 * xnpi2 / 8.0 - 32.0 * xmpi2 - 140.0 * xnpi2 * xmpi2 + 286.0) / den / d2
            bsi = (2115.0 * xnpi2 - 1792.0 * xmpi2 - 280.0 * xnpi2 * xmpi2 + 1328.0) / den / (4.0 * d2)
            csi = -(7695.0 * xnpi2 / 8.0 + 288.0 * xmpi2 - 180.0 * xnpi2 * xmpi2 - 2574.0) / den / (9.0 * d2)
            dsi = (198.0 * xnpi2 + 128.0 * xmpi2 - 40.0 * xnpi2 * xmpi2 - 736.0) / den / (16.0 * d2)
        else:
            raise X3dError("scheme is not defined")
        self.alpha, self.a, self.b, self.c, self.d = alpha, asi, bsi, csi, dsi
        self.coeffs[:] = _row(dsi, csi, bsi, asi, -2.0 * (asi + bsi + csi + dsi), asi, bsi, csi, dsi)
        dist_b = self._bulk(alpha)
        n, cs, ce, sa, sc = self.n_tds, self.coeffs_s, self.coeffs_e, self.dist_sa, self.dist_sc
        A, B, C, Dd = asi, bsi, csi, dsi
        if bc_start == BC_NEUMANN:
            if sym:
                sa[0], sc[0] = 0.0, 2 * alpha
                cs[0] = _row(0, 0, 0, 0, -2 * A - 2 * B - 2 * C - 2 * Dd, 2 * A, 2 * B, 2 * C, 2 * Dd)
                cs[1] = _row(0, 0, 0, A, -2 * A - B - 2 * C - 2 * Dd, A + C, B + Dd, C, Dd)
                cs[2] = _row(0, 0, B, A + C, -2 * A - 2 * B - 2 * C - Dd, A, B, C, Dd)
                cs[3] = _row(0, C, B + Dd, A, -2 * A - 2 * B - 2 * C - 2 * Dd, A, B, C, Dd)
            else:
                sa[0], sc[0] = 0.0, 0.0
                cs[0] = 0.0
                cs[1] = _row(0, 0, 0, A, -2 * A - 3 * B - 2 * C - 2 * Dd, A - C, B - Dd, C, Dd)
                cs[2] = _row(0, 0, B, A - C, -2 * A - 2 * B - 2 * C - 3 * Dd, A, B, C, Dd)
                cs[3] = _row(0, -C, B - Dd, A, -2 * A - 2 * B - 2 * C - 2 * Dd, A, B, C, Dd)
        elif bc_start == BC_DIRICHLET:
            sa[0], sc[0] = 0.0, 11.0
            cs[0] = _row(0, 0, 0, 0, 13.0 / d2, -27.0 / d2, 15.0 / d2, -1.0 / d2, 0)
            sa[1], sc[1] = 0.1, 0.1
            cs[1] = _row(0, 0, 0, 1.2 / d2, -2.4 / d2, 1.2 / d2, 0, 0, 0)
            sa[2], sc[2] = 2.0 / 11.0, 2.0 / 11.0
            t1, t2 = 3.0 / 44.0 / d2, 12.0 / 11.0 / d2
            cs[2] = _row(0, 0, t1, t2, -2.0 * (t1 + t2), t2, t1, 0, 0)
            sa[3], sc[3] = 2.0 / 11.0, 2.0 / 11.0
            cs[3] = cs[2]
        if bc_end == BC_NEUMANN:
            if sym:
                sa[n - 1], sc[n - 1] = 2 * alpha, 0.0
                ce[3] = _row(2 * Dd, 2 * C, 2 * B, 2 * A, -2 * A - 2 * B - 2 * C - 2 * Dd, 0, 0, 0, 0)
                ce[2] = _row(Dd, C, B + Dd, A + C, -2 * A - B - 2 * C - 2 * Dd, A, 0, 0, 0)
                ce[1] = _row(Dd, C, B, A, -2 * A - 2 * B - 2 * C - Dd, A + C, B, 0, 0)
                ce[0] = _row(Dd, C, B, A, -2 * A - 2 * B - 2 * C - 2 * Dd, A, B + Dd, C, 0)
            else:
                sa[n - 1], sc[n - 1] = 0.0, 0.0
                ce[3] = 0.0
                ce[2] = _row(Dd, C, B - Dd, A - C, -2 * A - 3 * B - 2 * C - 2 * Dd, A, 0, 0, 0)
                ce[1] = _row(Dd, C, B, A, -2 * A - 2 * B - 2 * C - 3 * Dd, A - C, B, 0, 0)
                ce[0] = _row(Dd, C, B, A, -2 * A - 2 * B - 2 * C - 2 * Dd, A, B - Dd, -C, 0)
        elif bc_end == BC_DIRICHLET:
            sa[n - 1], sc[n - 1] = 11.0, 0.0
            ce[3] = _row(0, -1.0 / d2, 15.0 / d2, -27.0 / d2, 13.0 / d2, 0, 0, 0, 0)
            sa[n - 2], sc[n - 2] = 0.1, 0.1
            ce[2] = _row(0, 0, 0, 1.2 / d2, -2.4 / d2, 1.2 / d2, 0, 0, 0)
            sa[n - 3], sc[n - 3] = 2.0 / 11.0, 2.0 / 11.0
            t1, t2 = 3.0 / 44.0 / d2, 12.0 / 11.0 / d2
            ce[1] = _row(0, 0, t1, t2, -2.0 * (t1 + t2), t2, t1, 0, 0)
            sa[n - 4], sc[n - 4] = 2.0 / 11.0, 2.0 / 11.0
            ce[0] = ce[1]
        return dist_b

    # -- src/tdsops.f90:620-764
    def _interpl_mid(self, scheme, from_to, bc_start, bc_end):
        if scheme == "classic":
            alpha, a, b, c, d = 0.3, 0.75, 0.05, 0.0, 0.0
        elif scheme == "optimised":
            alpha, d = 0.461658, 0.00146508
            a = (75.0 + 70.0 * alpha - 640.0 * d) / 128.0
            b = (-25.0 + 126.0 * alpha + 2304.0 * d) / 256.0
            c = (3.0 - 10.0 * alpha - 1280.0 * d) / 256.0
        elif scheme == "aggressive":
            alpha = 0.49
            a = (75.0 + 70.0 * alpha) / 128.0
            b = (-25.0 + 126.0 * alpha) / 256.0
            c = (3.0 - 10.0 * alpha) / 256.0
            d = 0.0
        else:
            raise X3dError("scheme is not defined")
        self.alpha, self.a, self.b, self.c, self.d = alpha, a, b, c, d
        if from_to == "v2p":
            self.coeffs[:] = _row(0.0, d, c, b, a, a, b, c, d)
        elif from_to == "p2v":
            self.coeffs[:] = _row(d, c, b, a, a, b, c, d, 0.0)
        else:
            raise X3dError("interpolation needs from_to = 'v2p' or 'p2v'")
        dist_b = self._bulk(alpha)
        n, cs, ce, sa, sc = self.n_tds, self.coeffs_s, self.coeffs_e, self.dist_sa, self.dist_sc
        if bc_start == BC_NEUMANN:
            sa[0] = 0.0
            if from_to == "v2p":
                dist_b[0] = 1.0 + alpha
                cs[0] = _row(0, 0, 0, 0, a, a + b, b + c, c + d, d)
                cs[1] = _row(0, 0, 0, b, a + c, a + d, b, c, d)
                cs[2] = _row(0, 0, c, b + d, a, a, b, c, d)
            else:
                sc[0] = 2 * alpha
                cs[0] = _row(0, 0, 0, 0, 2 * a, 2 * b, 2 * c, 2 * d, 0)
                cs[1] = _row(0, 0, 0, a + b, a + c, b + d, c, d, 0)
                cs[2] = _row(0, 0, b + c, a + d, a, b, c, d, 0)
                cs[3] = _row(0, c + d, b, a, a, b, c, d, 0)
        elif bc_start == BC_DIRICHLET:
            raise X3dError("Dirichlet BC is not supported for midpoint interpolations!")
        if bc_end == BC_NEUMANN:
            sc[n - 1] = 0.0
            if from_to == "v2p":
                dist_b[n - 1] = 1.0 + alpha
                ce[3] = 0.0
                ce[2] = _row(0, d, c + d, b + c, a + b, a, 0, 0, 0)
                ce[1] = _row(0, d, c, b, a + d, a + c, b, 0, 0)
                ce[0] = _row(0, d, c, b, a, a, b + d, c, 0)
            else:
                sa[n - 1] = 2 * alpha
                ce[3] = _row(2 * d, 2 * c, 2 * b, 2 * a, 0, 0, 0, 0, 0)
                ce[2] = _row(d, c, b + d, a + c, a + b, 0, 0, 0, 0)
                ce[1] = _row(d, c, b, a, a + d, b + c, 0, 0, 0)
                ce[0] = _row(d, c, b, a, a, b, c + d, 0, 0)
        elif bc_end == BC_DIRICHLET:
            raise X3dError("Dirichlet BC is not supported for midpoint interpolations!")
        return dist_b

    # -- src/tdsops.f90:766-872
    def _stagder_1st(self, delta, scheme, from_to, bc_start, bc_end):
        if scheme != "compact6":
            raise X3dError("scheme is not defined")
        alpha, aci, bci = 9.0 / 62.0, 63.0 / 62.0 / delta, 17.0 / 62.0 / 3.0 / delta
        self.alpha, self.a, self.b = alpha, aci, bci
        if from_to == "v2p":
            self.coeffs[:] = _row(0, 0, 0, -bci, -aci, aci, bci, 0, 0)
        elif from_to == "p2v":
            self.coeffs[:] = _row(0, 0, -bci, -aci, aci, bci, 0, 0, 0)
        else:
            raise X3dError("staggered derivative needs from_to = 'v2p' or 'p2v'")
        dist_b = self._bulk(alpha)
        n, cs, ce, sa, sc = self.n_tds, self.coeffs_s, self.coeffs_e, self.dist_sa, self.dist_sc
        if bc_start == BC_NEUMANN:
            sa[0] = 0.0
            if from_to == "v2p":
                dist_b[0] = 1.0 + alpha
                cs[0] = _row(0, 0, 0, 0, -aci - 2 * bci, aci + bci, bci, 0, 0)
                cs[1] = _row(0, 0, 0, -bci, -aci, aci, bci, 0, 0)
            else:
                sc[0] = 0.0
                cs[0] = 0.0
                cs[1] = _row(0, 0, 0, -aci - bci, aci, bci, 0, 0, 0)
        elif bc_start == BC_DIRICHLET:
            raise X3dError("Dirichlet BC is not supported for midpoint derivatives!")
        if bc_end == BC_NEUMANN:
            sc[n - 1] = 0.0
            if from_to == "v2p":
                dist_b[n - 1] = 1.0 + alpha
                ce[3] = 0.0
                ce[2] = _row(0, 0, 0, -bci, -aci - bci, aci + 2 * bci, 0, 0, 0)
            else:
                sa[n - 1] = 0.0
                ce[3] = 0.0
                ce[2] = _row(0, 0, -bci, -aci, aci + bci, 0, 0, 0, 0)
        elif bc_end == BC_DIRICHLET:
            raise X3dError("Dirichlet BC is not supported for midpoint derivatives!")
        return dist_b

    # -- src/tdsops.f90:874-931 (sequential recurrences, same order)
    # -- src/tdsops.f90:971-1103
    def _preprocess_penta_dist(self, bc_start, bc_end, sym):
        """LU of the pentadiagonal LHS for the non-periodic Thomas solve (der_penta_full); repurposed arrays:
        dist_fw = 1 / d_i, dist_af = l1_i, dist_sa = l2_i, dist_bw = u1_i, upper-2 = beta (beta_lhs_s in row 1).
        A periodic operator gets the Dirichlet-path (non-cyclic interior) factors, which der_penta_periodic
        corrects by Sherman-Morrison-Woodbury."""
        alp, bet, n = self.alpha, self.beta, self.n_tds
        fw, af, sa, bw = self.dist_fw, self.dist_af, self.dist_sa, self.dist_bw
        if bc_start == BC_NEUMANN:
            u1_1, self.beta_lhs_s = (0.0, 0.0) if sym else (2.0 * alp, 2.0 * bet)
        else:
            u1_1, self.beta_lhs_s = alp, bet
        sa[0], af[0], fw[0], bw[0] = 0.0, 0.0, 1.0, u1_1
        sa[1], af[1] = 0.0, alp
        if bc_start == BC_NEUMANN:
            d_i = ((1.0 - bet) if sym else (1.0 + bet)) - alp * u1_1
        else:
            d_i = 1.0 - alp * u1_1
        fw[1] = 1.0 / d_i
        bw[1] = alp - alp * self.beta_lhs_s
        l2 = bet * fw[0]
        l1 = (alp - l2 * bw[0]) * fw[1]
        d_i = (1.0 - l2 * self.beta_lhs_s) - l1 * bw[1]
        sa[2], af[2], fw[2], bw[2] = l2, l1, 1.0 / d_i, alp - l1 * bet
        for i in range(3, n):
            l2 = bet * fw[i - 2]
            l1 = (alp - l2 * bw[i - 2]) * fw[i - 1]
            d_i = (1.0 - l2 * bet) - l1 * bw[i - 1]
            sa[i], af[i], fw[i], bw[i] = l2, l1, 1.0 / d_i, alp - l1 * bet
        if bc_end == BC_NEUMANN:
            i = n - 2
            l2 = bet * fw[i - 2]
            l1 = (alp - l2 * bw[i - 2]) * fw[i - 1]
            d_i = ((1.0 - bet - l2 * bet) if sym else (1.0 + bet - l2 * bet)) - l1 * bw[i - 1]
            sa[i], af[i], fw[i], bw[i] = l2, l1, 1.0 / d_i, alp - l1 * bet
            if sym:
                sa[n - 1], af[n - 1], fw[n - 1], bw[n - 1] = 0.0, 0.0, 1.0, 0.0
            else:
                l2 = 2.0 * bet * fw[n - 3]
                l1 = (2.0 * alp - l2 * bw[n - 3]) * fw[n - 2]
                d_i = 1.0 - l2 * bet - l1 * bw[n - 2]
                sa[n - 1], af[n - 1], fw[n - 1], bw[n - 1] = l2, l1, 1.0 / d_i, alp - l1 * bet
        self.dist_sc[:] = 0.0

    def _preprocess_dist(self, b):
        n = self.n_tds
        fw, bw, sa, sc, af = self.dist_fw, self.dist_bw, self.dist_sa, self.dist_sc, self.dist_af
        for i in (0, 1):
            sa[i] = sa[i] / b[i]
            sc[i] = sc[i] / b[i]
            bw[i] = sc[i]
            af[i] = 1.0 / b[i]
        for i in range(2, n):
            fw[i] = 1.0 / (b[i] - sa[i] * sc[i - 1])
            af[i] = sa[i]
            sa[i] = -fw[i] * sa[i] * sa[i - 1]
            sc[i] = fw[i] * sc[i]
        for i in range(n - 3, 0, -1):
            sa[i] = sa[i] - sc[i] * sa[i + 1]
            bw[i] = sc[i]
            sc[i] = -sc[i] * sc[i + 1]
        fw[0] = 1.0 / (1.0 - sc[0] * sa[1])
        sa[0] = fw[0] * sa[0]
        sc[0] = -fw[0] * sc[0] * sc[1]


class Dirps:
    """dirps_t, src/tdsops.f90:51-59"""
    NAMES = ("der1st", "der1st_sym", "der2nd", "der2nd_sym", "stagder_v2p", "stagder_p2v",
             "interpl_v2p", "interpl_p2v")

    def __init__(self, direction):
        self.dir = direction
        for k in self.NAMES:
            setattr(self, k, None)
