"""FP32 flavour of the library (libx3d2_hip_sp.so = every source compiled with -DX3D_SINGLE_PREC, the reference's
-DSINGLE_PREC of src/common.f90:6-12), run in a process of its own (X3D_SINGLE_PREC is read when x3d2_amd is imported):
  (1) every operator of the reference's periodic fixture (ref_p000_rk3.npz: der1st / der2nd / interpolations / staggered
      derivatives in x, y, z, transeq) against the reference's FP64 vectors, tolerance 1e-5 relative (max norm);
  (2) the Dirichlet + stretched fixture (ref_c010_rk3.npz) likewise (general kernels, stretching tables);
  (3) TGV 64^3, RK3, FFT Poisson, 20 steps, fused driver and the deferred layer: enstrophy against the FP64 trace fixture
      (1e-5 relative), max |div u| at FP32 round-off;
  (4) a 512^3 fused step (the bench's kernels: tile kernels, z-first Poisson solve) against the same step at 1e-4 of the
      field maximum after ONE step from the analytic initial field (FP64 result computed by the FP64 library in the parent)."""
import json
import os
import sys

import numpy as np

os.environ["X3D_SINGLE_PREC"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import util  # noqa: E402
from x3d2_amd import _lib, make_tgv  # noqa: E402

assert _lib.SINGLE and _lib.LIB_PATH.endswith("_sp.so")
out = {}
what = sys.argv[1]
if what == "operators":
    from test_hip_parity import make_solver, set_inputs
    from util import OPNAMES, load_golden, relerr
    from x3d2_amd.common import DIR_X, DIR_Z, VERT, move_data_loc
    for fx in sys.argv[2].split(","):
        g = load_golden(fx)
        s = make_solver(g)
        set_inputs(s, g)
        b, al = s.backend, s.backend.allocator
        worst = {}
        for d, (dn, dp) in enumerate(zip("xyz", (s.xdirps, s.ydirps, s.zdirps)), 1):
            for op in OPNAMES:
                src = al.get_block(DIR_X, VERT)
                b.veccopy(src, s.u)
                if op.endswith("p2v"):
                    src.set_data_loc(move_data_loc(VERT, d, 1))
                a, o = al.get_block(d), al.get_block(d)
                if d == 1:
                    b.veccopy(a, src)
                    a.set_data_loc(src.data_loc)
                else:
                    b.reorder(a, src, 10 + d)
                b.tds_solve(o, a, getattr(dp, op))
                worst["tds.%s.%s" % (dn, op)] = float(relerr(b.get_field_data(o), g["tds.%s.%s" % (dn, op)]))
                for f in (src, a, o):
                    al.release_block(f)
        rhs = [al.get_block(DIR_X) for _ in range(3)]
        s.transeq(rhs, [s.u, s.v, s.w])
        for f, k in zip(rhs, ("du", "dv", "dw")):
            worst["transeq." + k] = float(relerr(b.get_field_data(f), g["transeq." + k]))
        div_u = al.get_block(DIR_Z)
        s.divergence_v2p(div_u, s.u, s.v, s.w)
        worst["div"] = float(relerr(b.get_field_data(div_u), g["div.div_u"]))
        s.gradient_p2v(*rhs, div_u)
        for f, k in zip(rhs, ("dpdx", "dpdy", "dpdz")):
            worst["grad." + k] = float(relerr(b.get_field_data(f), g["grad." + k]))
        for f in rhs:
            f.set_data_loc(VERT)
        s.curl(*rhs, s.u, s.v, s.w)
        for f, k in zip(rhs, "ijk"):
            worst["curl." + k] = float(relerr(b.get_field_data(f), g["curl." + k]))
        ens = 0.5 * sum(b.scalar_product(f, f) for f in rhs) / s.ngrid
        worst["enstrophy"] = float(abs(ens - g["curl.enstrophy"][0]) / abs(g["curl.enstrophy"][0]))
        out[fx] = {"max": max(worst.values()), "argmax": max(worst, key=worst.get), "all": worst}
elif what == "trace":
    fx = util.read_trace_fixture()
    for driver in ("fused", "lazy"):
        case = make_tgv(64, fused=(driver == "fused"), lazy=(driver == "lazy"))
        case.solver.n_output = 10
        rows = case.run(n_iters=20)
        out[driver] = {"enstrophy_rel": [abs(r[1] - f[1]) / f[1] for r, f in zip(rows, fx)],
                       "div_max": [float(r[2]) for r in rows]}
elif what == "step512":
    case = make_tgv(512, fused=True)
    case.step(1)
    s = case.solver
    fields = [s.backend.get_field_data(f) for f in (s.u, s.v, s.w)]
    np.savez(sys.argv[2], u=fields[0], v=fields[1], w=fields[2])
    row = case.postprocess(1, 1e-3)
    out = {"enstrophy": float(row[1]), "div_max": float(row[2]), "dtype": str(fields[0].dtype), "n_zfirst": int(s.n_zfirst)}
print("SPRESULT " + json.dumps(out))
