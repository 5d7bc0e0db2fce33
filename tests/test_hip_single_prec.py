"""The FP32 flavour of the library -- libx3d2_hip_sp.so, every kernel, table, scalar and transform on 4-byte reals
(make SP=1 = -DX3D_SINGLE_PREC; the reference's -DSINGLE_PREC, /root/reference/src/common.f90:6-12, whose CUDA backend
plans single-precision cuFFT transforms then, src/backend/cuda/poisson_fft.f90:427-458) -- against the reference's FP64
vectors and the FP64 library.  Tolerances are FP32's: 1e-5 relative for operators on O(1) fields (second derivatives
amplify the inputs' rounding by 1 / dx^2: 2e-4 there), 1e-5 on the enstrophy trace.  Each case runs tests/sp_worker.py in a
process of its own (the real kind is chosen when x3d2_amd is imported)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(*args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(HERE, "sp_worker.py")] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=timeout, env=dict(os.environ, X3D_SINGLE_PREC="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("SPRESULT ")][-1][9:])


@pytest.mark.parametrize("fixture", ["p000_rk3", "c010_rk3", "n111_rk2"])
def test_single_precision_operators_against_reference_vectors(fixture):
    """all 24 tds_solve operators (8 per direction: every closure, n_rhs = n_tds + 1, stretched y), transeq, divergence,
    gradient, curl and the enstrophy reduction of the periodic / Dirichlet + stretched / Neumann fixtures"""
    res = _worker("operators", fixture)[fixture]
    loose = {k: v for k, v in res["all"].items() if "der2nd" in k or k.startswith("transeq")}
    tight = {k: v for k, v in res["all"].items() if k not in loose}
    assert max(tight.values()) < 2e-5, (max(tight, key=tight.get), max(tight.values()))
    assert max(loose.values()) < 5e-4, (max(loose, key=loose.get), max(loose.values()))


def test_single_precision_tgv_trace():
    """TGV 64^3, RK3, FFT Poisson solve, 20 steps, fused driver and the reference's call sequence through the deferred
    layer: the enstrophy of the FP64 trace fixture to 1e-5, the projected field's divergence at FP32 round-off"""
    res = _worker("trace")
    for driver in ("fused", "lazy"):
        assert max(res[driver]["enstrophy_rel"]) < 1e-5, (driver, res[driver])
        assert max(res[driver]["div_max"][1:]) < 5e-5, (driver, res[driver])


def test_single_precision_step_at_the_bench_size(tmp_path):
    """one fused step at 512^3 -- the size-specialised kernels of the bench (three-in-one scan and tile kernels, on-chip solves,
    the z-first Poisson solve with the transforms on the z pairs' tiles) on 4-byte reals -- against the FP64 library"""
    from x3d2_amd import make_tgv
    ref = make_tgv(512, fused=True)
    ref.step(1)
    s = ref.solver
    want = [s.backend.get_field_data(f) for f in (s.u, s.v, s.w)]
    ens = ref.postprocess(1, 1e-3)[1]
    del ref, s
    out = tmp_path / "sp512.npz"
    res = _worker("step512", out, timeout=1200)
    assert res["dtype"] == "float32" and res["n_zfirst"] == 3
    got = np.load(out)
    for w, k in zip(want, "uvw"):
        assert np.max(np.abs(got[k].astype(np.float64) - w)) < 2e-5 * max(np.max(np.abs(w)), 1.0), k
    assert abs(res["enstrophy"] - ens) < 1e-5 * ens
    assert res["div_max"] < 1e-3  # (max |div u| of an FP32 projection at dx = 2 pi / 512: round-off / dx)


def test_fp32_through_the_boundary_the_reference_built_with_single_prec_on_the_fp32_library(tmp_path):
    """round 6: the Fortran side of the boundary in single precision.  fortran/_build/sp/xcompact_hip = the reference's own
    solver.f90 / cases / monitoring compiled with -DSINGLE_PREC (src/common.f90:6-12: dp = kind(0.0e0), MPI_REAL) + this
    repo's shim compiled with the same flag (m_x3d2_hip_capi.f90: x3d_creal = c_float) + libx3d2_hip_sp.so.  TGV 64^3,
    RK3, FFT Poisson, 20 steps: the enstrophy series of the FP64 fixture to 1e-5 relative (the tolerance of the FP32
    flavour's own trace test), the projection's divergence at FP32 round-off.  (Either shim checks x3d_real_bytes() against
    the kind it was compiled for before its first call and stops on the other flavour of the library.)"""
    import os
    import subprocess
    from util import read_trace_fixture
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fortran", "_build", "sp", "xcompact_hip")
    if not os.path.exists(exe) or not os.path.exists(os.path.join(root, "x3d2_amd", "libx3d2_hip_sp.so")):
        pytest.skip("FP32 shim binary not built (needs the reference tree at build time)")
    r = subprocess.run([exe, os.path.join(root, "fortran", "tgv64.x3d")], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = np.loadtxt(tmp_path / "monitoring.csv", delimiter=",", comments="#")
    fx = read_trace_fixture()
    assert rows.shape[0] >= 3
    assert np.all(np.abs(rows[:3, 1] - fx[:, 1]) < 1e-5 * fx[:, 1]), (rows[:3, 1], fx[:, 1])
    assert rows[:, 2].max() < 1e-4  # max |div u| after the projection, FP32
