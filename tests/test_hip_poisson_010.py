"""GPU parity for the non-periodic-y (010) Poisson path and the channel case, through the C ABI:
 (a) the reference's process_spectral_010 vectors (tests/golden/ref_c010*.npz),
 (b) the oracle on seeded inputs (uniform / top-bottom / centred / bottom stretching),
 (c) the acceptance checks of the reference's tests/verification/test_poisson_bc.f90 config 010
     (analytic cosines and div(grad(p)) = f, 1e-11) at its own size,
 (d) the reference's channel trace (xcompact, Poisson off) and oracle channel steps with Poisson.
FP64; tolerances stated per assertion."""
import numpy as np
import pytest

from util import assert_signature, load_big_steps, load_golden, namelist, product_mesh, read_csv, relerr, signature_of

pytestmark = pytest.mark.gpu
TOL = 1e-12


def product_solver(dims, stretching="uniform", beta=1.0, L=(4.0, 2.0, 2.0), poisson="FFT", fused=False,
                   Re=4200.0, dt=0.005):
    from x3d2_amd import Mesh
    from x3d2_amd.backend import HipBackend
    from x3d2_amd.solver import Solver, SolverConfig
    mesh = Mesh(dims, (1, 1, 1), L, ("periodic",) * 2, ("dirichlet",) * 2, ("periodic",) * 2,
                ("uniform", stretching, "uniform"), (1.0, beta, 1.0))
    return Solver(HipBackend(mesh), mesh, SolverConfig(Re=Re, dt=dt, poisson_solver_type=poisson, fused=fused))


def oracle_solver(dims, stretching="uniform", beta=1.0, L=(4.0, 2.0, 2.0), poisson="FFT", Re=4200.0, dt=0.005):
    from oracle import x3d_oracle as orc
    mesh = orc.Mesh(list(dims), [1, 1, 1], list(L), ["periodic"] * 2, ["dirichlet"] * 2, ["periodic"] * 2,
                    stretching=("uniform", stretching, "uniform"), beta=(1.0, beta, 1.0))
    return orc.Solver(mesh, Re=Re, dt=dt, time_intg="RK3", poisson=poisson)


def hip_poisson_solve(s, f):
    """solver%poisson_fft on a cell-centred Cartesian rhs [nz, ny, nx]"""
    from x3d2_amd.common import CELL, DIR_C
    b, al = s.backend, s.backend.allocator
    p, t = al.get_block(DIR_C, CELL), al.get_block(DIR_C)
    p.fill(0.0)
    b.set_field_data(p, f, CELL)
    b.poisson_fft.solve_poisson(p, t)
    out = b.get_field_data(p, CELL)
    al.release_block(p); al.release_block(t)
    return out


@pytest.mark.parametrize("name", ["c010u_rk3", "c010_rk3", "c010b_rk3", "c010c_rk3"])
def test_process_spectral_010_vs_reference_kernel(name):
    """uniform-y kernel (fw, -1/waves, bw fused) on the reference's own input/output pair; the
    reference's OMP kernel ignores stretching, so the product is built on the unstretched mesh"""
    from x3d2_amd.backend import HipBackend
    from x3d2_amd.solver import Solver, SolverConfig
    g = load_golden(name)
    c = namelist(g)
    c["stretching"] = ["uniform"] * 3
    mesh = product_mesh(c)
    s = Solver(HipBackend(mesh), mesh, SolverConfig(Re=c["Re"], dt=c["dt"]))
    pf = s.backend.poisson_fft
    assert pf.case == "010" and not pf.stretched_y
    # waves do not depend on the stretching
    assert np.allclose(pf.waves, g["spec.waves_re"], rtol=1e-13, atol=1e-13)
    pf.set_spectral(g["spec.in_re"] + 1j * g["spec.in_im"])
    pf.fft_postprocess_010()
    ref = g["spec.out010_re"] + 1j * g["spec.out010_im"]
    assert relerr(pf.get_spectral(), ref) < TOL


@pytest.mark.parametrize("ny", [16, 17])
def test_enforce_undo_periodicity_y(ny):
    from oracle import x3d_oracle as orc
    from x3d2_amd.common import CELL, DIR_C
    s = product_solver((12, ny + 1, 8))
    b, al = s.backend, s.backend.allocator
    rng = np.random.default_rng(ny)
    nx, nyc, nz = s.mesh.get_dims(CELL)
    assert nyc == ny
    f = rng.standard_normal((nz, ny, nx))
    a, o = al.get_block(DIR_C, CELL), al.get_block(DIR_C, CELL)
    b.set_field_data(a, f, CELL)
    b.poisson_fft.enforce_periodicity_y(o, a)
    got = b.get_field_data(o, CELL)
    assert np.array_equal(got, orc.PoissonFFT.enforce_periodicity_y(f))
    b.poisson_fft.undo_periodicity_y(a, o)
    assert np.array_equal(b.get_field_data(a, CELL), f)


@pytest.mark.parametrize("stretching,beta", [("uniform", 1.0), ("top-bottom", 0.259065151), ("centred", 1.3),
                                             ("bottom", 0.5)])
def test_poisson_010_solve_vs_oracle(stretching, beta):
    """poisson_010 end to end (enforce, rocFFT, spectral kernels / factored pentadiagonal solve,
    undo) against the oracle (numpy FFT + restated kernels, matrices eliminated per solve)"""
    dims = (24, 33, 16)
    s = product_solver(dims, stretching, beta)
    o = oracle_solver(dims, stretching, beta)
    pf = s.backend.poisson_fft
    assert pf.stretched_y == (stretching != "uniform")
    rng = np.random.default_rng(11)
    nx, ny, nz = (int(v) for v in o.mesh.global_cell_dims)
    f = rng.standard_normal((nz, ny, nx))
    ref = o.poisson_fft.solve(f)
    got = hip_poisson_solve(s, f)
    assert relerr(got, ref) < 1e-10, relerr(got, ref)


@pytest.mark.parametrize("form", ["", "split", "staged", "fused"])
@pytest.mark.parametrize("stretching,beta", [("top-bottom", 0.259065151), ("centred", 1.3), ("bottom", 0.5)])
def test_poisson_010_y_last_form_at_256_cells(stretching, beta, form, monkeypatch):
    """256 cells along a stretched y (the channel case's wall-normal direction): x and z are transformed first and
    ONE pass over the spectrum does the y transform, fft_postprocess_010's paired split and the inverse y transform
    (csrc/y010.hip; "staged" / "fused": the pentadiagonal sweeps on the tile too) -- against the oracle and against the 3-D-transform form
    (X3D_NO_Y010=1) on the same right-hand side"""
    dims = (32, 257, 16)
    monkeypatch.setenv("X3D_Y010_FORM", form)  # ("": the default form)
    s = product_solver(dims, stretching, beta)
    o = oracle_solver(dims, stretching, beta)
    rng = np.random.default_rng(12)
    nx, ny, nz = (int(v) for v in o.mesh.global_cell_dims)
    assert ny == 256
    f = rng.standard_normal((nz, ny, nx))
    got = hip_poisson_solve(s, f)
    ref = o.poisson_fft.solve(f)
    assert relerr(got, ref) < 1e-10, relerr(got, ref)
    monkeypatch.setenv("X3D_NO_Y010", "1")
    s3 = product_solver(dims, stretching, beta)
    got3 = hip_poisson_solve(s3, f)
    assert relerr(got3, ref) < 1e-10
    assert relerr(got, got3) < 1e-12, relerr(got, got3)
    assert np.max(np.abs(got - got3)) > 0.0  # (the two forms are different code: identical bits would mean one ran twice)


@pytest.mark.parametrize("n_wave,kind", [(2, "COS_X"), (2, "COS_Y"), (2, "COS_XY"), (2, "COS_XYZ"), (3, "COS_Y")])
def test_poisson_bc_010_acceptance(n_wave, kind):
    """tests/verification/test_poisson_bc.f90, config 010: 128 x 65 x 32, L = 1, tolerance 1e-11 on
    norm2(err)/N for (1) the analytic solution and (2) div(grad(p)) - f"""
    from x3d2_amd.common import CELL, DIR_C, DIR_X, DIR_Z, RDR_C2Z
    s = product_solver((128, 65, 32), L=(1.0, 1.0, 1.0))
    m, b, al = s.mesh, s.backend, s.backend.allocator
    x = m.midp_coords[0][None, None, :]
    y = m.midp_coords[1][None, :, None]
    z = m.midp_coords[2][:, None, None]
    k = n_wave * np.pi
    one = np.ones((len(m.midp_coords[2]), len(m.midp_coords[1]), len(m.midp_coords[0])))
    f, den = {"COS_X": (np.cos(k * x) * one, 1.0), "COS_Y": (np.cos(k * y) * one, 1.0),
              "COS_XY": (np.cos(k * x) * np.cos(k * y) * one, 2.0),
              "COS_XYZ": (np.cos(k * x) * np.cos(k * y) * np.cos(k * z) * one, 3.0)}[kind]
    exact = -f / (den * k * k)
    sol = hip_poisson_solve(s, f)
    err = (sol - sol[0, 0, 0]) - (exact - exact[0, 0, 0])
    assert np.linalg.norm(err.ravel()) / err.size <= 1e-11
    # check 2: gradient_c2v then divergence_v2c
    c = al.get_block(DIR_C, CELL)
    b.set_field_data(c, sol, CELL)
    p = al.get_block(DIR_Z, CELL)
    b.reorder(p, c, RDR_C2Z)
    dpdx, dpdy, dpdz = (al.get_block(DIR_X) for _ in range(3))
    s.gradient_p2v(dpdx, dpdy, dpdz, p)
    res = al.get_block(DIR_Z)
    s.divergence_v2p(res, dpdx, dpdy, dpdz)
    r = b.get_field_data(res, CELL) - f
    assert np.linalg.norm(r.ravel()) / r.size <= 1e-11


def _solvers_100(dims, L):
    from oracle import x3d_oracle as orc
    from x3d2_amd import Mesh
    from x3d2_amd.backend import HipBackend
    from x3d2_amd.solver import Solver, SolverConfig
    mesh = Mesh(dims, (1, 1, 1), L, ("dirichlet",) * 2, ("periodic",) * 2, ("periodic",) * 2)
    s = Solver(HipBackend(mesh), mesh, SolverConfig(poisson_solver_type="FFT"))
    om = orc.Mesh(list(dims), [1, 1, 1], list(L), ["dirichlet"] * 2, ["periodic"] * 2, ["periodic"] * 2)
    return s, orc.Solver(om, poisson="FFT")


@pytest.mark.parametrize("dims", [(33, 16, 8), (129, 64, 32), (66, 40, 12)])
def test_poisson_100_solve_vs_oracle(dims):
    """x non-periodic (poisson_100, src/poisson_fft.f90:244-256; CUDA-only in the reference): the transposed 010
    solve of HipPoissonFFT100 against the oracle on a seeded right-hand side (even and odd cell counts in x)"""
    s, o = _solvers_100(dims, (1.0, 2.0, 1.5))
    assert s.backend.poisson_fft.case == "100"
    rng = np.random.default_rng(11)
    f = rng.standard_normal(tuple(int(n) for n in o.mesh.global_cell_dims)[::-1])
    f -= f.mean()
    assert relerr(hip_poisson_solve(s, f), o.poisson_fft.solve(f)) < 1e-10


@pytest.mark.parametrize("n_wave,kind", [(2, "COS_X"), (2, "COS_Y"), (2, "COS_XY"), (2, "COS_XYZ"), (3, "COS_X")])
def test_poisson_bc_100_acceptance(n_wave, kind):
    """tests/verification/test_poisson_bc.f90, config 100: 129 x 64 x 32, L = 1, 1e-11 on the analytic solution
    (its n = 3 cases in periodic directions are the test's own XFAILs)"""
    s, _ = _solvers_100((129, 64, 32), (1.0, 1.0, 1.0))
    m = s.mesh
    x = m.midp_coords[0][None, None, :]
    y = m.midp_coords[1][None, :, None]
    z = m.midp_coords[2][:, None, None]
    k = n_wave * np.pi
    one = np.ones((len(m.midp_coords[2]), len(m.midp_coords[1]), len(m.midp_coords[0])))
    f, den = {"COS_X": (np.cos(k * x) * one, 1.0), "COS_Y": (np.cos(k * y) * one, 1.0),
              "COS_XY": (np.cos(k * x) * np.cos(k * y) * one, 2.0),
              "COS_XYZ": (np.cos(k * x) * np.cos(k * y) * np.cos(k * z) * one, 3.0)}[kind]
    exact = -f / (den * k * k)
    sol = hip_poisson_solve(s, f)
    err = (sol - sol[0, 0, 0]) - (exact - exact[0, 0, 0])
    assert np.linalg.norm(err.ravel()) / err.size <= 1e-11


def _solvers_110(dims, L):
    from oracle import x3d_oracle as orc
    from x3d2_amd import Mesh
    from x3d2_amd.backend import HipBackend
    from x3d2_amd.solver import Solver, SolverConfig
    mesh = Mesh(dims, (1, 1, 1), L, ("dirichlet",) * 2, ("dirichlet",) * 2, ("periodic",) * 2)
    s = Solver(HipBackend(mesh), mesh, SolverConfig(poisson_solver_type="FFT"))
    om = orc.Mesh(list(dims), [1, 1, 1], list(L), ["dirichlet"] * 2, ["dirichlet"] * 2, ["periodic"] * 2)
    return s, orc.Solver(om, poisson="FFT")


@pytest.mark.parametrize("dims", [(33, 17, 8), (65, 129, 16), (34, 21, 12)])
def test_poisson_110_solve_vs_oracle(dims):
    """x and y non-periodic (poisson_110, src/poisson_fft.f90:258-273; CUDA-only in the reference): HipPoissonFFT110
    (z-first twin problem, x3d_poisson_postprocess_011) against the oracle on a seeded right-hand side; even and odd
    cell counts"""
    s, o = _solvers_110(dims, (1.0, 2.0, 1.5))
    assert s.backend.poisson_fft.case == "110"
    rng = np.random.default_rng(13)
    f = rng.standard_normal(tuple(int(n) for n in o.mesh.global_cell_dims)[::-1])
    f -= f.mean()
    assert relerr(hip_poisson_solve(s, f), o.poisson_fft.solve(f)) < 1e-10


@pytest.mark.parametrize("n_wave,kind", [(2, "COS_X"), (2, "COS_Y"), (2, "COS_XY"), (2, "COS_XYZ"), (3, "COS_X"),
                                         (3, "COS_Y"), (3, "COS_XY")])
def test_poisson_bc_110_acceptance(n_wave, kind):
    """tests/verification/test_poisson_bc.f90, config 110 at its own size 129 x 257 x 64, L = 1: the analytic
    solution to 1e-11 for every case that test expects to pass"""
    s, _ = _solvers_110((129, 257, 64), (1.0, 1.0, 1.0))
    m = s.mesh
    x = m.midp_coords[0][None, None, :]
    y = m.midp_coords[1][None, :, None]
    z = m.midp_coords[2][:, None, None]
    k = n_wave * np.pi
    one = np.ones((len(m.midp_coords[2]), len(m.midp_coords[1]), len(m.midp_coords[0])))
    f, den = {"COS_X": (np.cos(k * x) * one, 1.0), "COS_Y": (np.cos(k * y) * one, 1.0),
              "COS_XY": (np.cos(k * x) * np.cos(k * y) * one, 2.0),
              "COS_XYZ": (np.cos(k * x) * np.cos(k * y) * np.cos(k * z) * one, 3.0)}[kind]
    exact = -f / (den * k * k)
    sol = hip_poisson_solve(s, f)
    err = (sol - sol[0, 0, 0]) - (exact - exact[0, 0, 0])
    assert np.linalg.norm(err.ravel()) / err.size <= 1e-11


@pytest.mark.parametrize("config", ["000", "010", "100", "110"])
def test_poisson_bc_acceptance_through_the_fortran_shim(config, tmp_path):
    """fortran/_build/poisson_bc_hip: the reference's mesh / tdsops / poisson_fft_t%base_init (waves_set incl. its
    100 branch) and solve_poisson drivers on hip_backend_t, with the cosine checks of its test_poisson_bc.f90
    (which cannot be pointed at a third backend itself): analytic solution to 1e-11"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fortran", "_build", "poisson_bc_hip")
    if not os.path.exists(exe):
        pytest.skip("shim binary not built (needs the reference tree at build time)")
    r = subprocess.run([exe, config], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_channel_trace_no_poisson_vs_reference():
    """the reference's xcompact channel run (16 x 17 x 12, top-bottom stretching, rotation until
    iteration 3, no noise, Poisson off): monitoring.csv digit for digit"""
    from x3d2_amd import make_channel
    ref = read_csv("channel17_rk3_nopoisson")
    case = make_channel((16, 17, 12), poisson="CG", rotation=True, omega_rot=0.12, n_rotate=3)
    case.solver.n_output = 2
    rows = np.array(case.run(n_iters=6))
    assert np.allclose(rows[:, 1], ref[:, 1], rtol=5e-12, atol=0)
    assert np.allclose(rows[1:, 2], ref[1:, 2], rtol=1e-9)
    assert np.allclose(rows[1:, 3], ref[1:, 3], rtol=1e-9)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("stretching,beta", [("top-bottom", 0.259065151), ("uniform", 1.0)])
def test_channel_steps_with_poisson_vs_oracle(stretching, beta, fused):
    """two full channel steps (6 sub-steps: define_BC, transeq, forcings, RK3, apply_BC, pressure
    correction with the 010 Poisson solve) against the oracle; div u after projection"""
    _channel_steps((24, 33, 16), stretching, beta, fused, 2)


def test_channel_step_at_the_bench_pencil_lengths_vs_oracle():
    """1024-point x pencils and 257 stretched wall-normal vertices (BASELINE configs[4]'s pencils, 16 planes of
    them): the kernels bench.py --case channel runs -- K3w (csrc/xwide.hip) along x, K3g (csrc/ygen.hip) along y,
    the 010 Poisson solve -- in one full step against the oracle"""
    from x3d2_amd import _lib
    lib = _lib.load()
    case = _channel_steps((1024, 257, 16), "top-bottom", 0.259065151, True, 1)
    assert int(lib.x3d_backend_counter(case.solver.backend.h, 0)) >= 6  # x and y: three-in-one launches, 3 sub-steps
    assert case.solver.n_rot_fused == 3  # the rotation forcing rode on K3w's transeq_x (k_xwide_transeq3<ROT>)
    assert case.solver.n_interleaved == 0  # (16-row z pencils: not the tile kernel's; the bench's 512 are)


def test_channel_step_at_the_bench_size_two_poisson_forms(monkeypatch):
    """BASELINE configs[4] at its full size, 1024 x 257 x 512: one step of the fused driver with the 010 solve in its
    default form (x, z ; y last inside the post-processing kernels, pentadiagonal sweeps on their tiles -- csrc/y010.hip)
    against the same step with the 3-D transforms and the stand-alone post-processing kernels (X3D_NO_Y010=1): different
    transforms, different pentadiagonal code, the same velocity to 1e-11.  The ORACLE at this size is paid once, in
    tests/test_hip_channel_multirank.py::test_channel_two_slabs_at_the_bench_pencil_lengths, which holds the DEFAULT form's
    step against it (1e-10); through the 1e-11 here the X3D_NO_Y010 form is held to the oracle as well (its pieces are
    pinned to the oracle directly at the small sizes above).  The projection leaves the divergence it leaves at small sizes."""
    import gc
    import torch
    from x3d2_amd import make_channel
    dims = (1024, 257, 512)
    out = []
    # round 6: the default is the z-first form (csrc/zfirst.hip: z transforms on the tiles of the z operator pairs, complex
    # 1024-point x transforms, the y pass on [kz][y][x] with every x mode); X3D_NO_ZFIRST010=1: round 5's x-first form of the
    # same y pass; X3D_NO_Y010=1: the 3-D transforms + stand-alone kernels
    for envs, zf in (((), 3), (("X3D_NO_ZFIRST010",), 0), (("X3D_NO_ZFIRST010", "X3D_NO_Y010"), 0)):
        for k in ("X3D_NO_ZFIRST010", "X3D_NO_Y010"):
            monkeypatch.delenv(k, raising=False)
        for k in envs:
            monkeypatch.setenv(k, "1")
        case = make_channel(dims, fused=True, rotation=True, omega_rot=0.12, n_rotate=2)
        case.step(1)
        s = case.solver
        assert s.n_zfirst == zf
        _, ens, dmax, _ = case.postprocess(1, 0.005)
        out.append(([s.backend.get_field_data(f) for f in (s.u, s.v, s.w)], ens, dmax))
        del case, s
        gc.collect()
        torch.cuda.empty_cache()
    (fz, ez, dz), (fa, ea, da), (fb, eb, db) = out
    for ref, other in ((fb, fa), (fb, fz)):
        for x, y, nm in zip(other, ref, "uvw"):
            assert np.max(np.abs(x - y)) < 1e-11 * max(np.max(np.abs(y)), 1.0), nm
            assert np.max(np.abs(x - y)) > 0.0, nm  # (different routes)
    assert abs(ea - eb) < 1e-11 * abs(eb) and abs(ez - eb) < 1e-11 * abs(eb)
    assert da < 1e-9 and db < 1e-9 and dz < 1e-9, (da, db, dz)


@pytest.mark.parametrize("stretching,beta", [("top-bottom", 0.259065151), ("bottom", 0.3)])
def test_zfirst_form_of_the_010_solve_vs_the_x_first_form(stretching, beta):
    """x3d_poisson_solve_010_rows_zfirst (round 6: z transform ; complex x transform over all 1024 modes ; y pass on the
    half-z spectrum with the operators of x3d_poisson_set_stretching_zfirst ; back) against x3d_poisson_solve_010_rows
    (rocFFT x and z, y pass on the half-x spectrum) on one random right-hand side at the channel's size: the same solve,
    every transform and the factored operators in another layout -- 1e-11; sym (odd / even systems) and the full system"""
    import torch
    from x3d2_amd.common import CELL, DIR_C
    s = product_solver((1024, 257, 512), stretching, beta)
    b, al, pf = s.backend, s.backend.allocator, s.backend.poisson_fft
    assert pf.zfirst_ok() and pf.stretched_y_sym == (stretching == "top-bottom")
    f1, f2 = al.get_block(DIR_C, CELL), al.get_block(DIR_C, CELL)
    g = torch.Generator(device="cpu").manual_seed(7)
    rhs = torch.randn(tuple(f1.data.shape), generator=g, dtype=torch.float64)
    f1.data.copy_(rhs.to(f1.data.device))
    f2.data.copy_(f1.data)
    pf.solve_interleaved(f1)
    pf.solve_interleaved_zfirst(f2)
    x, y = b.get_field_data(f2, CELL), b.get_field_data(f1, CELL)
    assert np.all(np.isfinite(x))
    assert np.max(np.abs(x - y)) < 1e-11 * np.max(np.abs(y)) and np.max(np.abs(x - y)) > 0.0
    al.release_block(f1); al.release_block(f2)


@pytest.mark.parametrize("dims", [(1024, 33, 16), (256, 33, 16)])
def test_rotation_forcing_inside_transeq_x_vs_the_two_vecadds(dims, monkeypatch):
    """k_xwide_transeq3<ROT> / k_xscan_transeq2x3<CHN>: du = transeq_x(u) - omega v, dv = transeq_x(v) + omega u
    formed in the x kernel against the reference's order (two vecadd's after the three directions,
    src/case/channel.f90:191-207): the same sum in another order -- equal to round-off"""
    fusedrot = _channel_steps(dims, "top-bottom", 0.259065151, True, 1)
    assert fusedrot.solver.n_rot_fused == 3
    monkeypatch.setenv("X3D_NO_ROT_FUSED", "1")
    plain = _channel_steps(dims, "top-bottom", 0.259065151, True, 1)
    assert plain.solver.n_rot_fused == 0
    b1, b2 = fusedrot.solver.backend, plain.solver.backend
    for a, c in zip((fusedrot.solver.u, fusedrot.solver.v, fusedrot.solver.w), (plain.solver.u, plain.solver.v, plain.solver.w)):
        x, y = b1.get_field_data(a), b2.get_field_data(c)
        assert np.max(np.abs(x - y)) < 1e-13 * max(np.max(np.abs(y)), 1.0)


@pytest.mark.parametrize("dims", [(32, 17, 256), (16, 9, 512)])
def test_z_pairs_interleave_the_y_rows_like_the_solvers_copy_kernels(dims):
    """x3d_tds_solve_pair_yperm: divergence's last z pair writes the y rows where enforce_periodicity_y would put
    them, gradient's first z pair reads them where the backward transform leaves them -- bit for bit the same as
    the pair kernel followed / preceded by the copy kernel (tile kernel K3y: periodic z pencils of 256 / 512 rows;
    other pencils are not served: the caller keeps the copies, second test below)"""
    from x3d2_amd.common import DIR_C, DIR_Z
    s = product_solver(dims)
    b, al, z = s.backend, s.backend.allocator, s.zdirps
    pf = b.poisson_fft
    nyc = pf.interleaved_rows()
    assert nyc == dims[1] - 1
    rng = np.random.default_rng(5)
    blk = [al.get_block(DIR_C) for _ in range(7)]
    i1, i2, o1, o2, r1, r2, tmp = blk
    for f in (i1, i2):
        f.data.copy_(__import__("torch").from_numpy(rng.standard_normal(tuple(f.data.shape))).to(f.data.device))
    for f in (o1, o2, r1, r2, tmp):
        f.fill(0.0)
    # mode 0: out = A(in1) + B(in2), rows interleaved on the way out
    assert b.tds_pair_yperm(0, o1, None, i1, i2, z.interpl_v2p, z.stagder_v2p, nyc)
    b.tds_pair(0, tmp, None, i1, i2, z.interpl_v2p, z.stagder_v2p, DIR_Z)
    pf.enforce_periodicity_y(r1, tmp)
    nx, ny, nz = dims
    got, ref = b.get_field_data(o1), b.get_field_data(r1)
    assert np.array_equal(got[:, :nyc, :], ref[:, :nyc, :])
    # mode 1: out1 = A(in1), out2 = B(in1), in1's rows read through the interleave
    assert b.tds_pair_yperm(1, o1, o2, i1, None, z.interpl_p2v, z.stagder_p2v, nyc)
    pf.undo_periodicity_y(tmp, i1)
    b.tds_pair(1, r1, r2, tmp, None, z.interpl_p2v, z.stagder_p2v, DIR_Z)
    for g, r in ((o1, r1), (o2, r2)):
        got, ref = b.get_field_data(g), b.get_field_data(r)
        assert np.array_equal(got[:, :nyc, :], ref[:, :nyc, :])
    for f in blk:
        al.release_block(f)


@pytest.mark.parametrize("dims,taken", [((32, 17, 256), True), ((48, 33, 24), False)])
def test_fused_channel_step_with_and_without_the_interleaving_pairs(dims, taken, monkeypatch):
    """the fused pressure correction with the solver's two copy kernels folded into the z pairs == the same step
    with the copies (X3D_NO_YPERM=1), bit for bit; and against the oracle.  24-row z pencils are not served by the
    tile kernel: the driver keeps the copies by itself."""
    case = _channel_steps(dims, "top-bottom", 0.259065151, True, 2)
    assert case.solver.n_interleaved == (6 if taken else 0)
    monkeypatch.setenv("X3D_NO_YPERM", "1")
    plain = _channel_steps(dims, "top-bottom", 0.259065151, True, 2)
    assert plain.solver.n_interleaved == 0
    for a, c in zip((case.solver.u, case.solver.v, case.solver.w), (plain.solver.u, plain.solver.v, plain.solver.w)):
        assert np.array_equal(case.solver.backend.get_field_data(a), plain.solver.backend.get_field_data(c))


def test_channel_fusion_entry_points_decline_or_fail_loudly():
    """the fusion extensions of the channel path say so when they do not apply (the caller then issues the plain
    sequence) and reject bad arguments like the other entry points"""
    from x3d2_amd.common import DIR_X, VERT, X3dError
    s = product_solver((32, 18, 256))  # 17 cell rows: odd
    b, al, z = s.backend, s.backend.allocator, s.zdirps
    f = [al.get_block(DIR_X, VERT) for _ in range(6)]
    for x in f:
        x.fill(0.0)
    assert b.poisson_fft.interleaved_rows() == 0  # odd row count: the solver keeps its copy kernels
    assert not b.tds_pair_yperm(0, f[0], None, f[1], f[2], z.interpl_v2p, z.stagder_v2p, 17)
    with pytest.raises(X3dError):
        b.tds_pair_yperm(0, f[0], None, f[1], f[2], z.interpl_v2p, z.stagder_v2p, 400)   # more rows than the block has
    with pytest.raises(X3dError):
        b.tds_pair_yperm(0, f[0], None, f[0], f[2], z.interpl_v2p, z.stagder_v2p, 16)    # output aliases an input
    # nothing to fuse: declined, fields untouched
    assert not b.transeq_x_rot(f[0], f[1], f[2], f[3], f[4], f[5], s.nu, s.xdirps, 0.0)
    # 32-point x pencils: no single-pass kernel takes the forcing
    assert not b.transeq_x_rot(f[0], f[1], f[2], f[3], f[4], f[5], s.nu, s.xdirps, 0.12)
    with pytest.raises(X3dError):
        b.transeq_x_rot(f[0], f[1], f[2], f[0], f[4], f[5], s.nu, s.xdirps, 0.12)        # du aliases u
    with pytest.raises(X3dError):
        b.tds_lincomb(f[0], s.xdirps.stagder_v2p, DIR_X, f[1], f[2], [1.0], [f[3]], wall=f[0])  # du aliases the wall field
    for x in f:
        al.release_block(x)


@pytest.mark.parametrize("nx", [1024, 512, 256])
@pytest.mark.parametrize("omega", [0.12, 0.0])
def test_bulk_velocity_shift_inside_transeq_x(omega, nx):
    """x3d_transeq_x_rot(u_shift): u += the device scalar of x3d_field_mean_shift inside the transeq_x kernel (K3w
    at 1024-point pencils, K3s at 256 / 512) ==
    x3d_field_shift_by followed by the same kernel without it, bit for bit (u and du, dv, dw); together the two
    halves == x3d_field_shift_to_mean"""
    import torch
    from x3d2_amd.common import DIR_X, VERT
    s = product_solver((nx, 9, 8))
    b, al = s.backend, s.backend.allocator
    rng = np.random.default_rng(3)
    blk = [al.get_block(DIR_X, VERT) for _ in range(13)]
    u, v, w, u2, u3 = blk[:5]
    d1, d2 = blk[5:8], blk[8:11]
    for f in (u, v, w):
        f.data.copy_(torch.from_numpy(rng.standard_normal(tuple(f.data.shape))).to(f.data.device))
    u2.data.copy_(u.data)
    u3.data.copy_(u.data)
    sh = b.field_mean_shift(u, 2.0 / 3.0)
    assert b.transeq_x_rot(*d1, u, v, w, s.nu, s.xdirps, omega, sh)
    b.field_shift_by(u2, sh)  # (the scalar is still in place: no reduction since)
    if omega != 0.0:
        assert b.transeq_x_rot(*d2, u2, v, w, s.nu, s.xdirps, omega)
    else:
        b.transeq_dir(DIR_X, *d2, u2, v, w, s.nu, s.xdirps)
    b.field_shift_to_mean(u3, 2.0 / 3.0)
    assert np.array_equal(b.get_field_data(u, VERT), b.get_field_data(u2, VERT))
    assert np.array_equal(b.get_field_data(u, VERT), b.get_field_data(u3, VERT))
    for x, y in zip(d1, d2):
        assert np.array_equal(b.get_field_data(x, VERT), b.get_field_data(y, VERT))
    for f in blk:
        al.release_block(f)


@pytest.mark.parametrize("omega,shift", [(0.12, True), (0.0, True), (0.12, False), (0.0, False)])
def test_velocity_correction_shift_and_rotation_inside_transeq_x_at_1024_rows(omega, shift):
    """x3d_transeq_x_update_rot (round 6, csrc/xwide.hip k_xwide_transeq3_upd): the pending velocity correction
    u, v, w -= tds_solve(g), the bulk-velocity shift and the rotation forcing inside the transeq_x kernel of the
    1024-row pencils == x3d_tds_solve_acc x 3 ; x3d_field_shift_by ; x3d_transeq_x_rot: the corrected u, v, w bit for
    bit, du, dv, dw to the last bit or two.  Lane tables staged without the ST / STC blocks (uniform x)."""
    import torch
    from x3d2_amd import _lib
    from x3d2_amd.common import DIR_X, VERT
    s = product_solver((1024, 9, 8))
    b, al, x = s.backend, s.backend.allocator, s.xdirps
    rng = np.random.default_rng(11)
    blk = [al.get_block(DIR_X, VERT) for _ in range(15)]
    u, v, w, gu, gv, gw, u2, v2, w2 = blk[:9]
    d1, d2 = blk[9:12], blk[12:15]
    for f in (u, v, w, gu, gv, gw):
        f.data.copy_(torch.from_numpy(rng.standard_normal(tuple(f.data.shape))).to(f.data.device))
    for a, c in ((u2, u), (v2, v), (w2, w)):
        a.data.copy_(c.data)
    n0 = int(_lib.load().x3d_backend_counter(b.h, 1))
    sh = b.field_mean_shift(u, 2.0 / 3.0) if shift else None
    assert b.transeq_x_update_rot(*d1, u, v, w, s.nu, x, (gu, gv, gw), x.stagder_p2v, x.interpl_p2v, -1.0, omega, sh)
    assert int(_lib.load().x3d_backend_counter(b.h, 1)) == n0 + 1
    b.tds_apply(u2, gu, x.stagder_p2v, DIR_X, accumulate=True, scale=-1.0)
    b.tds_apply(v2, gv, x.interpl_p2v, DIR_X, accumulate=True, scale=-1.0)
    b.tds_apply(w2, gw, x.interpl_p2v, DIR_X, accumulate=True, scale=-1.0)
    if omega != 0.0 or shift:
        assert b.transeq_x_rot(*d2, u2, v2, w2, s.nu, x, omega, sh)  # (the scalar is still in place: no reduction since)
    else:
        b.transeq_dir(DIR_X, *d2, u2, v2, w2, s.nu, x)
    for a, c, nm in ((u, u2, "u"), (v, v2, "v"), (w, w2, "w")):
        x_, y_ = b.get_field_data(a, VERT), b.get_field_data(c, VERT)
        assert np.array_equal(x_, y_), (nm, float(np.max(np.abs(x_ - y_))))
    # du, dv, dw: the same expressions in another kernel body -- which product of a sum of two the compiler fuses is its
    # choice per kernel (-ffp-contract=fast): a last bit may differ (observed: 1 ulp of the largest value, 2.3e-16)
    for a, c, nm in zip(d1, d2, ("du", "dv", "dw")):
        x_, y_ = b.get_field_data(a, VERT), b.get_field_data(c, VERT)
        assert np.max(np.abs(x_ - y_)) <= 1e-15 * np.max(np.abs(y_)), (nm, float(np.max(np.abs(x_ - y_))), float(np.max(np.abs(y_))))
    for f in blk:
        al.release_block(f)


def test_channel_steps_with_the_velocity_correction_deferred_to_transeq_x(monkeypatch):
    """ChannelCase.correction_deferrable (round 6): the fused channel step whose pressure-gradient correction waits for
    the next sub-step's transeq_x kernel -- the bulk-velocity shift then comes from the mean of the UNCORRECTED u (the
    correction is a periodic x derivative: every x pencil of it sums to zero up to rounding) -- against the same steps
    with the correction applied by the pressure step (X3D_NO_CHANNEL_DEFER_GRAD=1): round-off apart; and both against
    the oracle (_channel_steps)."""
    from x3d2_amd import _lib
    dims = (1024, 33, 8)
    case = _channel_steps(dims, "top-bottom", 0.259065151, True, 2, div_bound=None)
    assert int(_lib.load().x3d_backend_counter(case.solver.backend.h, 1)) == 4  # sub-steps 2, 3 of both steps
    assert case.solver.n_rot_fused == 3 and case.solver.pending_grad is None
    # ... and the volume integral define_BC asks for was taken by the kernel that formed the new u (k_xwide_tds_lin),
    # x3d_tds_solve_lincomb_wall_mean: another summation order than x3d_field_mean_shift's, round-off apart
    assert case.solver.n_mean_taken == 4
    monkeypatch.setenv("X3D_NO_MEAN_IN_LINCOMB", "1")
    nomean = _channel_steps(dims, "top-bottom", 0.259065151, True, 2, div_bound=None)
    assert nomean.solver.n_mean_taken == 0
    for a, c in zip((case.solver.u, case.solver.v, case.solver.w), (nomean.solver.u, nomean.solver.v, nomean.solver.w)):
        x, y = case.solver.backend.get_field_data(a), nomean.solver.backend.get_field_data(c)
        assert np.max(np.abs(x - y)) < 1e-13 * max(np.max(np.abs(y)), 1.0)
    monkeypatch.setenv("X3D_NO_CHANNEL_DEFER_GRAD", "1")
    plain = _channel_steps(dims, "top-bottom", 0.259065151, True, 2, div_bound=None)
    assert int(_lib.load().x3d_backend_counter(plain.solver.backend.h, 1)) == 0
    for a, c in zip((case.solver.u, case.solver.v, case.solver.w), (plain.solver.u, plain.solver.v, plain.solver.w)):
        x, y = case.solver.backend.get_field_data(a), plain.solver.backend.get_field_data(c)
        assert np.max(np.abs(x - y)) < 1e-13 * max(np.max(np.abs(y)), 1.0)


@pytest.mark.parametrize("nx,loc", [(1024, "VERT"), (1024, "CELL"), (256, "VERT")])
def test_volume_integral_taken_by_the_kernel_that_forms_the_field(nx, loc):
    """x3d_tds_solve_lincomb_wall_mean == x3d_tds_solve_lincomb_wall ; x3d_field_mean_shift: y and du bit for bit, the shift
    scalar to round-off (the 1024-row kernel sums per wave, x3d_field_mean_shift per row); 256-row pencils take the
    two calls one after the other: the same bits"""
    import torch
    from x3d2_amd import _lib
    from x3d2_amd.common import CELL, DIR_X, VERT
    s = product_solver((nx, 9, 8))
    b, al, x = s.backend, s.backend.allocator, s.xdirps
    dl = VERT if loc == "VERT" else CELL
    rng = np.random.default_rng(nx)
    blk = [al.get_block(DIR_X, dl) for _ in range(8)]
    base, d1, d2, wall, y1, y2, o1, o2 = blk
    for f in (base, d1, d2, wall):
        f.data.copy_(torch.from_numpy(rng.standard_normal(tuple(f.data.shape))).to(f.data.device))
    for f in (y1, y2, o1, o2):
        f.fill(0.0)

    def scalar(ptr):
        t = torch.empty(1, dtype=torch.float64, device=b.device)
        _lib.check(b.lib.x3d_copy_device(b.h, t.data_ptr(), ptr, 1))
        b.sync()
        return float(t.cpu()[0])

    sh1 = b.tds_lincomb(o1, x.stagder_v2p, DIR_X, y1, base, [0.3, -0.7], [d1, d2], wall=wall, mean_target=2.0 / 3.0)
    v1 = scalar(sh1)
    b.tds_lincomb(o2, x.stagder_v2p, DIR_X, y2, base, [0.3, -0.7], [d1, d2], wall=wall)
    v2 = scalar(b.field_mean_shift(y2, 2.0 / 3.0))
    assert np.array_equal(b.get_field_data(y1, dl), b.get_field_data(y2, dl))
    assert np.array_equal(b.get_field_data(o1, dl), b.get_field_data(o2, dl))
    assert abs(v1 - v2) < 1e-14 * max(abs(v2), 1.0) and (nx == 1024 or v1 == v2)
    # ... and by the plain operator's kernel (x3d_tds_solve_mean: the form the bench's sub-step takes -- its RK stage is done by
    # transeq's z launch, so the divergence's first x operator is the first kernel to read the new u)
    o1.fill(0.0)
    v3 = scalar(b.tds_apply_mean(o1, y2, x.stagder_v2p, DIR_X, 2.0 / 3.0))
    assert np.array_equal(b.get_field_data(o1, dl), b.get_field_data(o2, dl))
    assert abs(v3 - v2) < 1e-14 * max(abs(v2), 1.0) and (nx == 1024 or v3 == v2)
    for f in blk:
        al.release_block(f)


@pytest.mark.parametrize("nx", [1024, 256, 48])
def test_rk_stage_wall_values_and_first_x_operator_in_one_kernel(nx):
    """x3d_tds_solve_lincomb_wall (K3w at 1024-point pencils, K3s at 256, the three calls one after the other
    otherwise) == x3d_lincomb ; x3d_field_set_face_from_field(Y_FACE) ; x3d_tds_solve, bit for bit"""
    import torch
    from x3d2_amd.common import DIR_X, VERT, Y_FACE
    s = product_solver((nx, 9, 8))
    b, al, x = s.backend, s.backend.allocator, s.xdirps
    rng = np.random.default_rng(nx)
    blk = [al.get_block(DIR_X, VERT) for _ in range(8)]
    base, d1, d2, wall, y1, y2, o1, o2 = blk
    for f in (base, d1, d2, wall):
        f.data.copy_(torch.from_numpy(rng.standard_normal(tuple(f.data.shape))).to(f.data.device))
    for op in (x.stagder_v2p, x.interpl_v2p):
        for f in (y1, y2, o1, o2):
            f.fill(0.0)
        b.tds_lincomb(o1, op, DIR_X, y1, base, [0.3, -0.7], [d1, d2], wall=wall)
        b.lincomb(y2, base, [0.3, -0.7], [d1, d2])
        b.field_set_face_from_field(y2, wall, 0.0, Y_FACE)
        b.tds_apply(o2, y2, op, DIR_X)
        assert np.array_equal(b.get_field_data(y1, VERT), b.get_field_data(y2, VERT))
        assert np.array_equal(b.get_field_data(y1, VERT)[:, 0, :], b.get_field_data(wall, VERT)[:, 0, :])
        assert np.array_equal(b.get_field_data(y1, VERT)[:, -1, :], b.get_field_data(wall, VERT)[:, -1, :])
        assert np.array_equal(b.get_field_data(o1, VERT), b.get_field_data(o2, VERT))
    for f in blk:
        al.release_block(f)


@pytest.mark.parametrize("dims", [(1024, 33, 8), (256, 33, 16), (48, 33, 8)])
def test_channel_step_with_the_wall_values_stamped_inside_the_divergence_kernels(dims, monkeypatch):
    """fused channel step with apply_BC folded into the kernels that form the new velocity == the step with the
    stage, the stamping and the operators as separate launches (X3D_NO_DEFER_WALLS=1), bit for bit"""
    # (div_bound: the residual of these very coarse grids is the algorithm's, equal to the oracle's; not bounded here)
    # (round 6: the bulk-velocity integral taken by whichever kernel forms / first reads the new u sums in that kernel's order:
    #  both runs take it by x3d_field_mean_shift here, so that the wall stamping is ALL that differs between them)
    monkeypatch.setenv("X3D_NO_MEAN_IN_LINCOMB", "1")
    case = _channel_steps(dims, "top-bottom", 0.259065151, True, 1, div_bound=None)
    monkeypatch.setenv("X3D_NO_DEFER_WALLS", "1")
    plain = _channel_steps(dims, "top-bottom", 0.259065151, True, 1, div_bound=None)
    for a, c in zip((case.solver.u, case.solver.v, case.solver.w), (plain.solver.u, plain.solver.v, plain.solver.w)):
        assert np.array_equal(case.solver.backend.get_field_data(a), plain.solver.backend.get_field_data(c))


def _channel_steps(dims, stretching, beta, fused, nsteps, div_bound=1e-6):
    from x3d2_amd import make_channel
    from x3d2_amd.common import VERT
    case = make_channel(dims, stretching=stretching, beta=beta, fused=fused, rotation=True, omega_rot=0.12,
                        n_rotate=2)
    key = "channel%dx%dx%d" % tuple(dims)
    fix = load_big_steps()
    if (fix is not None and key + ".enstrophy" in fix and nsteps == 1 and stretching == "top-bottom" and beta == 0.259065151
            and __import__("os").environ.get("X3D_TEST_RUN_ORACLE") != "1"):
        # round 6: the bench-size step against the oracle's STORED signatures (oracle/gen_step_fixtures.py restates the
        # oracle side below; ~60 s of host time per call otherwise)
        s, m = case.solver, case.solver.mesh
        X = 2 * np.pi * m.vert_coords[0][None, None, :] / m.L[0]
        Y = np.pi * m.vert_coords[1][None, :, None] / m.L[1]
        Z = 2 * np.pi * m.vert_coords[2][:, None, None] / m.L[2]
        pert = (0.05 * np.sin(X) * np.sin(Y) ** 2 * np.cos(Z), 0.04 * np.cos(X) * np.sin(Y) ** 2 * np.sin(Z),
                0.03 * np.sin(2 * X) * np.sin(Y) ** 2 * np.cos(Z))
        for fp, d in zip((s.u, s.v, s.w), pert):
            s.backend.set_field_data(fp, s.backend.get_field_data(fp) + d)
        case.step(1)
        for fp, nm in ((s.u, "u"), (s.v, "v"), (s.w, "w")):
            sg = signature_of(fix, key + "." + nm)
            assert_signature(s.backend.get_field_data(fp), sg, 1e-10, nm, scale=max(float(sg["absmax"]), 1.0))
        _, ens, dmax, dmean = case.postprocess(1, 0.01)
        eo = (float(fix[key + ".enstrophy"]), float(fix[key + ".div_max"]))
        assert abs(ens - eo[0]) < 1e-10 * abs(eo[0])
        assert abs(dmax - eo[1]) < 1e-6 * eo[1] + 1e-12 and (div_bound is None or dmax < div_bound)
        return case
    o = oracle_solver(dims, stretching, beta)
    o.init_channel(rotation=True, omega_rot=0.12, n_rotate=2)
    # perturb both identically so that all three components are active (smooth: the Nyquist
    # modes of the periodic directions are not invertible, cf. the XFAIL cases of test_poisson_bc)
    s = case.solver
    m = o.mesh
    X = 2 * np.pi * m.vert_coords[0][None, None, :] / m.L[0]
    Y = np.pi * m.vert_coords[1][None, :, None] / m.L[1]
    Z = 2 * np.pi * m.vert_coords[2][:, None, None] / m.L[2]
    pert = (0.05 * np.sin(X) * np.sin(Y) ** 2 * np.cos(Z), 0.04 * np.cos(X) * np.sin(Y) ** 2 * np.sin(Z),
            0.03 * np.sin(2 * X) * np.sin(Y) ** 2 * np.cos(Z))
    for (fo, fp), d in zip(((o.u, s.u), (o.v, s.v), (o.w, s.w)), pert):
        a = o.backend.get_field_data(fo) + d
        o.backend.set_field_data(fo, a)
        s.backend.set_field_data(fp, a)
    for it in range(1, nsteps + 1):
        o.step_channel(it)
        case.step(it)
    for fo, fp, nm in ((o.u, s.u, "u"), (o.v, s.v, "v"), (o.w, s.w, "w")):
        ref = o.backend.get_field_data(fo)
        got = s.backend.get_field_data(fp)
        assert np.max(np.abs(got - ref)) < 1e-10 * max(np.max(np.abs(ref)), 1.0), nm
    _, ens, dmax, dmean = case.postprocess(nsteps, 0.01)
    eo = o.monitor()
    assert abs(ens - eo[0]) < 1e-10 * abs(eo[0])
    # div u after the projection: the residual the reference algorithm itself leaves
    # (absolute part: round-off of a max over the grid, ~1e-13 at 4M points)
    assert abs(dmax - eo[1]) < 1e-6 * eo[1] + 1e-12 and (div_bound is None or dmax < div_bound)
    return case


def test_unchanged_reference_channel_case_through_fortran_shim(tmp_path):
    """fortran/_build/xcompact_hip: the reference's own channel case (case/channel.f90, solver.f90,
    poisson_fft.f90 incl. stretching_matrix) on the HIP backend with the 010 FFT Poisson solve --
    a configuration the reference itself only supports on its CUDA backend.  Its monitoring.csv
    must agree with the Python host driver of this package on the same library."""
    import os
    import subprocess
    from x3d2_amd import make_channel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fortran", "_build", "xcompact_hip")
    if not os.path.exists(exe):
        pytest.skip("shim binary not built (needs the reference tree at build time)")
    r = subprocess.run([exe, os.path.join(root, "fortran", "channel33.x3d")], cwd=tmp_path, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = np.loadtxt(tmp_path / "monitoring.csv", delimiter=",", comments="#")
    case = make_channel((32, 33, 16), rotation=True, omega_rot=0.12, n_rotate=3)
    case.solver.n_output = 2
    mine = np.array(case.run(n_iters=6))
    assert rows.shape[0] == mine.shape[0] == 4
    assert np.allclose(rows[:, 1], mine[:, 1], rtol=1e-11, atol=0), (rows[:, 1], mine[:, 1])
    assert np.all(rows[1:, 2] < 1e-5) and np.allclose(rows[1:, 2], mine[1:, 2], rtol=1e-3)


def test_bulk_velocity_shift_on_the_device_is_bit_identical_to_the_host_path():
    """define_BC_channel's correction (src/case/channel.f90:70-77): field_volume_integral -> host -> field_shift
    against x3d_field_shift_to_mean (partial sums finished on the device in the same order)"""
    from x3d2_amd import make_channel
    from x3d2_amd.common import CELL
    case = make_channel((48, 33, 20), poisson="CG")
    s = case.solver
    b = s.backend
    rng = np.random.default_rng(5)
    a = rng.standard_normal((20, 33, 48))
    b.set_field_data(s.u, a)
    ub = b.field_volume_integral(s.u) / float(np.prod(s.mesh.get_global_dims(CELL)))
    b.field_shift(s.u, 2.0 / 3.0 - ub)
    host = b.get_field_data(s.u)
    b.set_field_data(s.u, a)
    b.field_shift_to_mean(s.u, 2.0 / 3.0)
    assert np.array_equal(b.get_field_data(s.u), host)


def test_wall_noise_generated_on_the_device():
    """x3d_wall_noise against a numpy restatement of its counter-based generator (splitmix64): the two wall
    planes bit for bit, every other plane untouched, repeatable for a seed, a new draw differs; and the channel
    case with inlet noise runs without host uploads and keeps the bulk velocity at 2/3"""
    from x3d2_amd import make_channel
    from x3d2_amd.common import CELL, DIR_X, VERT

    def mix64(z):
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    nx, ny, nz = 40, 17, 12
    case = make_channel((nx, ny, nz), poisson="CG", inlet_noise=(0.125, 0.25, 0.5), seed=1234)
    b = case.solver.backend
    f = b.allocator.get_block(DIR_X, VERT)
    f.fill(7.0)
    seed, draw, amp = 1234, 5, 0.25
    b.wall_noise(f, amp, seed, draw)
    got = b.get_field_data(f, VERT)
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(seed + draw))
        q = np.arange(2 * nx * nz, dtype=np.uint64)
        r = (mix64(key + q) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    want = (amp * (2.0 * r - 1.0)).reshape(2, nz, nx)
    assert np.array_equal(got[:, 0, :], want[0]) and np.array_equal(got[:, -1, :], want[1])
    assert np.all(got[:, 1:-1, :] == 7.0)
    assert abs(want.mean()) < 0.02 and abs(want.std() - amp / np.sqrt(3.0)) < 0.02 and np.abs(want).max() <= amp
    b.wall_noise(f, amp, seed, draw + 1)
    assert not np.array_equal(b.get_field_data(f, VERT)[:, 0, :], want[0])
    # the case: two steps with wall noise; the walls carry the noise, the bulk velocity stays at 2/3
    case.step(1)
    case.step(2)
    s = case.solver
    case.define_BC()
    ub = b.field_volume_integral(s.u) / float(np.prod(s.mesh.get_global_dims(CELL)))
    # (the reference divides the sum over VERTICES by the number of CELLS, src/case/channel.f90:68-72, so the
    #  corrected mean sits at 2/3 only up to (ny_vert - ny_cell) / ny_cell of the shift)
    assert abs(ub - 2.0 / 3.0) < 1e-3
    wall = b.get_field_data(case.bc_start_y[2], VERT)
    assert 0.0 < np.abs(wall[:, 0, :]).max() <= 0.5 and np.all(wall[:, 1:-1, :] == 0.0)
    assert np.all(np.isfinite(b.get_field_data(s.u)))
