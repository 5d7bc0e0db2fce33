"""Pins the oracle's non-periodic-y (010) Poisson path: wave numbers, spectral
equivalence constants, stretching matrices and process_spectral_010 against
vectors dumped from the REAL reference (tests/golden/ref_c010*.npz), and the
assembled solve against the acceptance checks of the reference's
tests/verification/test_poisson_bc.f90 (analytic cosines to 1e-11; div(grad(p)) = f).
CPU only."""
import numpy as np
import pytest

from oracle import x3d_oracle as orc
from test_oracle_vs_reference import load, make_solver, relerr, TOL

CASES = ["c010u_rk3", "c010_rk3", "c010b_rk3", "c010c_rk3"]  # uniform, top-bottom, bottom, centred


def poisson_of(g):
    s = make_solver(g)
    return s, orc.PoissonFFT(s.mesh, s.xdirps, s.ydirps, s.zdirps)


@pytest.mark.parametrize("name", CASES)
def test_wave_numbers_and_waves_010(name):
    g = load(name)
    s, p = poisson_of(g)
    assert p.case == "010"
    for k in ("ax", "bx", "ay", "by", "az", "bz"):
        assert np.allclose(getattr(p, k), g["spec." + k], rtol=1e-14, atol=1e-16), k
    for k in ("kx", "ky", "kz", "k2x", "k2y", "k2z"):
        assert np.allclose(getattr(p, k), g[f"spec.{k}_re"], rtol=1e-13, atol=1e-14), k
    assert np.allclose(p.waves, g["spec.waves_re"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(g["spec.waves_re"], g["spec.waves_im"])


@pytest.mark.parametrize("name", CASES)
def test_process_spectral_010_matches_reference_kernel(name):
    """the reference's OpenMP kernel (it ignores stretching): fw, -1/waves, bw"""
    g = load(name)
    s, p = poisson_of(g)
    c = g["spec.in_re"] + 1j * g["spec.in_im"]
    buf = np.ascontiguousarray(c).view(np.float64)
    orc.lib().orc_process_spectral_010(orc._p(buf), orc._p(p.waves), orc._p(p.waves), *p._args())
    ref = g["spec.out010_re"] + 1j * g["spec.out010_im"]
    assert relerr(buf.view(np.complex128), ref) < TOL


@pytest.mark.parametrize("name", ["c010_rk3", "c010c_rk3", "c010b_rk3"])
def test_stretching_matrices_match_reference(name):
    g = load(name)
    s, p = poisson_of(g)
    assert p.stretched_y
    for k in ("x", "y", "z"):
        assert np.allclose(getattr(p, "trans_" + k), g["spec.trans_" + k], rtol=1e-13, atol=1e-15), k
    assert bool(g["spec.stretched_y_sym"][0]) == p.stretched_y_sym
    if p.stretched_y_sym:
        sets = (("a_odd", p.a_odd), ("a_even", p.a_even))
    else:
        sets = (("a", p.a_full),)
    for tag, mine in sets:
        n = mine.shape[2]
        for d in range(1, 6):
            ref = g[f"spec.{tag}_re.{d}"]
            sl = slice(0, n)
            # entries the reference never sets / reads past ky(ny) for, and the solve never uses
            if d == 4:
                sl = slice(0, n - 1)
            elif d == 5:
                sl = slice(0, n - 2)
            elif d == 2:
                sl = slice(1, n)
            elif d == 1:
                sl = slice(2, n)
            assert np.array_equal(ref[:, sl], g[f"spec.{tag}_im.{d}"][:, sl])  # one copy is enough
            a, b = mine[d - 1][:, sl], ref[:, sl]
            scale = max(np.max(np.abs(b)), 1e-300)
            assert np.max(np.abs(a - b)) / scale < 1e-12, (tag, d)


def cosine_fields(mesh, n_wave, kind):
    x = mesh.midp_coords[0][None, None, :]
    y = mesh.midp_coords[1][None, :, None]
    z = mesh.midp_coords[2][:, None, None]
    k = n_wave * np.pi
    one = np.ones((len(mesh.midp_coords[2]), len(mesh.midp_coords[1]), len(mesh.midp_coords[0])))
    if kind == "COS_Y":
        f = np.cos(k * y) * one
        return f, -f / k ** 2
    if kind == "COS_X":
        f = np.cos(k * x) * one
        return f, -f / k ** 2
    if kind == "COS_XY":
        f = np.cos(k * x) * np.cos(k * y) * one
        return f, -f / (2 * k ** 2)
    f = np.cos(k * x) * np.cos(k * y) * np.cos(k * z) * one
    return f, -f / (3 * k ** 2)


def div_grad(s, sol):
    """div(grad(p)) through the oracle's vector calculus, as run_single_test does"""
    b = s.backend
    c = b.get_block(orc.DIR_C, orc.CELL)
    nz, ny, nx = sol.shape
    c.data[...] = 0.0
    c.data[:nz, :ny, :nx] = sol
    p = b.get_block(orc.DIR_Z, orc.CELL)
    b.reorder(p, c, 43)
    p.data_loc = orc.CELL
    dpdx, dpdy, dpdz = (b.get_block(orc.DIR_X) for _ in range(3))
    s.gradient_p2v(dpdx, dpdy, dpdz, p)
    res = b.get_block(orc.DIR_Z)
    s.divergence_v2p(res, dpdx, dpdy, dpdz)
    return b.get_field_data(res, orc.CELL)


@pytest.mark.parametrize("n_wave,kind", [(2, "COS_X"), (2, "COS_Y"), (2, "COS_XY"), (2, "COS_XYZ"),
                                         (3, "COS_Y")])
def test_poisson_010_uniform_analytic_and_divgrad(n_wave, kind):
    """tests/verification/test_poisson_bc.f90 config 010 (128 x 65 x 32, L = 1): both checks, 1e-11"""
    mesh = orc.Mesh([128, 65, 32], [1, 1, 1], [1.0, 1.0, 1.0], ["periodic"] * 2, ["dirichlet"] * 2,
                    ["periodic"] * 2)
    s = orc.Solver(mesh, poisson="FFT")
    f, exact = cosine_fields(mesh, n_wave, kind)
    sol = s.poisson_fft.solve(f)
    err = (sol - sol[0, 0, 0]) - (exact - exact[0, 0, 0])
    assert np.linalg.norm(err.ravel()) / err.size <= 1e-11
    res = div_grad(s, sol) - f
    assert np.linalg.norm(res.ravel()) / res.size <= 1e-11


@pytest.mark.parametrize("n_wave,kind", [(2, "COS_X"), (2, "COS_Y"), (2, "COS_XY"), (2, "COS_XYZ"), (3, "COS_X")])
def test_poisson_100_analytic_and_divgrad(n_wave, kind):
    """tests/verification/test_poisson_bc.f90 config 100 (x-dirichlet, 129 x 64 x 32, L = 1): the solver the reference
    only has on its CUDA backend (poisson_100 = the 010 machinery on the x <-> y transposed problem), so there are
    no reference vectors to pin it to -- the pin is the reference's own acceptance test: every n = 2 case and the
    n = 3 case in the non-periodic direction pass both checks at 1e-11 (the other n = 3 cases are its XFAILs)"""
    mesh = orc.Mesh([129, 64, 32], [1, 1, 1], [1.0, 1.0, 1.0], ["dirichlet"] * 2, ["periodic"] * 2,
                    ["periodic"] * 2)
    s = orc.Solver(mesh, poisson="FFT")
    assert s.poisson_fft.case == "100"
    f, exact = cosine_fields(mesh, n_wave, kind)
    sol = s.poisson_fft.solve(f)
    err = (sol - sol[0, 0, 0]) - (exact - exact[0, 0, 0])
    assert np.linalg.norm(err.ravel()) / err.size <= 1e-11
    res = div_grad(s, sol) - f
    assert np.linalg.norm(res.ravel()) / res.size <= 1e-11


@pytest.mark.parametrize("n_wave,kind", [(2, "COS_X"), (2, "COS_Y"), (2, "COS_XY"), (2, "COS_XYZ"), (3, "COS_X"),
                                         (3, "COS_Y"), (3, "COS_XY")])
def test_poisson_110_analytic_and_divgrad(n_wave, kind):
    """tests/verification/test_poisson_bc.f90 config 110 (x,y-dirichlet; its 129 x 257 x 64 halved in y and z to keep
    the CPU suite short: 129 x 129 x 32, L = 1): every case that test expects to pass -- n = 2 all four, n = 3 where
    no periodic direction is involved -- passes both checks at 1e-11.  CUDA-only in the reference: this acceptance
    matrix is the pin."""
    mesh = orc.Mesh([129, 129, 32], [1, 1, 1], [1.0, 1.0, 1.0], ["dirichlet"] * 2, ["dirichlet"] * 2,
                    ["periodic"] * 2)
    s = orc.Solver(mesh, poisson="FFT")
    assert s.poisson_fft.case == "110"
    f, exact = cosine_fields(mesh, n_wave, kind)
    sol = s.poisson_fft.solve(f)
    err = (sol - sol[0, 0, 0]) - (exact - exact[0, 0, 0])
    assert np.linalg.norm(err.ravel()) / err.size <= 1e-11
    res = div_grad(s, sol) - f
    assert np.linalg.norm(res.ravel()) / res.size <= 1e-11


def test_poisson_100_is_the_transposed_010_solve():
    """the same right-hand side with x and y exchanged through the 010 solver gives the transposed answer"""
    rng = np.random.default_rng(5)
    m100 = orc.Mesh([33, 16, 8], [1, 1, 1], [1.0, 2.0, 1.5], ["dirichlet"] * 2, ["periodic"] * 2, ["periodic"] * 2)
    m010 = orc.Mesh([16, 33, 8], [1, 1, 1], [2.0, 1.0, 1.5], ["periodic"] * 2, ["dirichlet"] * 2, ["periodic"] * 2)
    s100, s010 = orc.Solver(m100, poisson="FFT"), orc.Solver(m010, poisson="FFT")
    f = rng.standard_normal((8, 16, 32))  # [nz][ny][nx cells]
    f -= f.mean()
    a = s100.poisson_fft.solve(f)
    b = s010.poisson_fft.solve(np.ascontiguousarray(np.swapaxes(f, 1, 2)))
    assert relerr(a, np.swapaxes(b, 1, 2)) < 1e-13


@pytest.mark.parametrize("stretching,beta,tol", [("top-bottom", 0.259065151, 1e-6), ("centred", 1.3, 1e-6),
                                                 ("bottom", 0.5, 1e-2)])
def test_poisson_010_stretched_inverts_div_grad(stretching, beta, tol):
    """stretched y: the pentadiagonal spectral solve inverts the discrete div(grad) of the stretched
    operators up to the accuracy of the reference's own matrices (which the oracle reproduces to
    1e-12, test above): ~1e-7 for the symmetric stretchings, ~2e-3 for 'bottom' at this size.
    (The reference has no test on a stretched mesh; the exactness of the solve itself is checked
    against a dense solve below.)"""
    mesh = orc.Mesh([32, 33, 16], [1, 1, 1], [4.0, 2.0, 2.0], ["periodic"] * 2, ["dirichlet"] * 2,
                    ["periodic"] * 2, stretching=("uniform", stretching, "uniform"), beta=(1.0, beta, 1.0))
    s = orc.Solver(mesh, poisson="FFT")
    rng = np.random.default_rng(7)
    nx, ny, nz = (int(v) for v in mesh.global_cell_dims)
    f = rng.standard_normal((nz, ny, nx))
    # compatibility: the rhs of a Neumann problem must be in the range of div(grad)
    f = div_grad(s, s.poisson_fft.solve(f))
    sol = s.poisson_fft.solve(f)
    res = div_grad(s, sol) - f
    assert np.max(np.abs(res)) / np.max(np.abs(f)) < tol


@pytest.mark.parametrize("stretching,beta", [("top-bottom", 0.259065151), ("bottom", 0.5)])
def test_pentadiagonal_solve_against_dense_solve(stretching, beta):
    """process_spectral_010_poisson (CUDA-only in the reference, restated in the oracle): for the
    reference's matrices it must return the solution of the pentadiagonal system, rows taken
    odd / even / all as off, inc say"""
    mesh = orc.Mesh([16, 25, 8], [1, 1, 1], [4.0, 2.0, 2.0], ["periodic"] * 2, ["dirichlet"] * 2,
                    ["periodic"] * 2, stretching=("uniform", stretching, "uniform"), beta=(1.0, beta, 1.0))
    s = orc.Solver(mesh, poisson="FFT")
    p = s.poisson_fft
    nxs, nys, nzs = p.n_spec
    rng = np.random.default_rng(3)
    c = rng.standard_normal((nzs, nys, nxs)) + 1j * rng.standard_normal((nzs, nys, nxs))
    if p.stretched_y_sym:
        runs = [(p.a_odd, 0, 2, slice(0, None, 2)), (p.a_even, 1, 2, slice(1, None, 2))]
    else:
        runs = [(p.a_full, 0, 1, slice(None))]
    for a, off, inc, rows in runs:
        n = a.shape[2]
        buf = np.ascontiguousarray(c).view(np.float64).copy()
        ar, ai = a.copy(), a.copy()
        orc.lib().orc_spectral_010_penta(orc._p(buf), orc._p(ar), orc._p(ai), off, inc, nxs, nys, nzs, n,
                                         p.nx, p.nz)
        out = buf.view(np.complex128)
        for k in range(nzs):
            for i in range(nxs):
                if i == p.nx // 2 and k == p.nz // 2:
                    assert np.all(out[k, rows, i] == 0)  # the reference zeroes this line
                    continue
                A = np.zeros((n, n))
                for d in range(5):
                    for j in range(n):
                        if 0 <= j + d - 2 < n:
                            A[j, j + d - 2] = a[d, k, j, i]
                x = np.linalg.solve(A, c[k, rows, i])
                assert np.max(np.abs(x - out[k, rows, i])) <= 1e-10 * np.max(np.abs(x)) * np.linalg.cond(A) ** 0.5


def test_enforce_undo_periodicity_roundtrip():
    rng = np.random.default_rng(1)
    for ny in (8, 9):
        f = rng.standard_normal((3, ny, 5))
        g = orc.PoissonFFT.enforce_periodicity_y(f)
        assert np.array_equal(orc.PoissonFFT.undo_periodicity_y(g), f)
        n2 = ny // 2
        assert np.array_equal(g[:, :n2], f[:, 0:2 * n2:2])
        assert np.array_equal(g[:, -1], f[:, 1])
