#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: write tests/golden/oracle_big_steps.npz -- signatures (tests/util.py: a 24^3 lattice of samples, sum,
sum of squares, hash-weighted sum, max) of the ORACLE's fields for the full-size cases of the GPU suite, so that the GPU box
pays only the HIP side of those tests (task 8: the suite ran 752 s of a 1200 s limit, a third of it oracle
time on the host):

  tgv256x512x512, tgv512      one full TGV step (RK3, FFT Poisson), tests/test_hip_parity.py::
                              test_fused_full_step_against_the_oracle_at_fast_path_sizes
  pc512                       pressure_correction of a rough field (Taylor-Green + 10 % hash noise), tests/pc512_worker.py
  channel1024x257x512         one channel step (top-bottom stretching, rotation forcing, 010 Poisson), tests/
                              test_hip_channel_multirank.py::test_channel_two_slabs_at_the_bench_pencil_lengths and
                              test_hip_poisson_010.py

The oracle that writes them is the pinned one: the script first repeats oracle/gen_trace_fixture.py's check -- TGV 64^3
must reproduce the three enstrophy values the survey recorded from the reference's xcompact to 2e-13 -- and REFUSES to
write otherwise; the small-size forms of every case stay compared with the oracle element by element on the GPU box.

    python oracle/gen_step_fixtures.py [case ...]        (~30 GB of memory, minutes of host time per case)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import x3d_oracle as orc  # noqa: E402
from util import field_signature, noisy_tgv  # noqa: E402

SURVEY = [3.749999996799e-01, 3.749898433321e-01, 3.749874980813e-01]  # SURVEY.md 8c, t = 0, 0.01, 0.02
TWOPI = 6.283185307179586
PATH = os.path.join(ROOT, "tests", "golden", "oracle_big_steps.npz")


def pinned():
    mesh = orc.Mesh([64] * 3, [1, 1, 1], [TWOPI] * 3, ["periodic"] * 2, ["periodic"] * 2, ["periodic"] * 2)
    s = orc.Solver(mesh, Re=1600.0, dt=1e-3, time_intg="RK3", poisson="FFT")
    s.init_tgv()
    ens = [s.monitor()[0]]
    for it in range(1, 21):
        s.step()
        if it % 10 == 0:
            ens.append(s.monitor()[0])
    for e, ref in zip(ens, SURVEY):
        if abs(e - ref) > 2e-13:
            raise SystemExit(f"oracle enstrophy {e!r} does not reproduce the survey's {ref!r}: nothing written")


def put(out, name, fields, **scalars):
    for nm, a in zip("uvw", fields):
        for k, v in field_signature(a).items():
            out[f"{name}.{nm}.{k}"] = v
    for k, v in scalars.items():
        out[f"{name}.{k}"] = np.float64(v)


def tgv_step(out, name, dims):
    om = orc.Mesh(list(dims), [1, 1, 1], [TWOPI] * 3, ["periodic"] * 2, ["periodic"] * 2, ["periodic"] * 2)
    o = orc.Solver(om, Re=1600.0, dt=1e-3, time_intg="RK3", poisson="FFT")
    o.init_tgv()
    o.step()
    ens, dmax, _ = o.monitor()
    put(out, name, [o.backend.get_field_data(f) for f in (o.u, o.v, o.w)], enstrophy=ens, div_max=dmax)


def pc512(out):
    n = 512
    data = [np.ascontiguousarray(a) for a in noisy_tgv((n, n, n), (0, 0, 0), (n, n, n), amp=0.1)]
    om = orc.Mesh([n] * 3, [1, 1, 1], [TWOPI] * 3, ["periodic"] * 2, ["periodic"] * 2, ["periodic"] * 2)
    o = orc.Solver(om, poisson="FFT")
    for of, d in zip((o.u, o.v, o.w), data):
        of.data_loc = orc.VERT
        o.backend.set_field_data(of, d)
    o.pressure_correction(o.u, o.v, o.w)
    _, omx, _ = o.monitor()
    put(out, "pc512", [o.backend.get_field_data(f) for f in (o.u, o.v, o.w)], div_max=omx)


def channel(out, dims=(1024, 257, 512), nsteps=1):
    """tests/test_hip_poisson_010.py::_channel_steps's oracle side, restated"""
    mesh = orc.Mesh(list(dims), [1, 1, 1], [4.0, 2.0, 2.0], ["periodic"] * 2, ["dirichlet"] * 2, ["periodic"] * 2,
                    stretching=("uniform", "top-bottom", "uniform"), beta=(1.0, 0.259065151, 1.0))
    o = orc.Solver(mesh, Re=4200.0, dt=0.005, time_intg="RK3", poisson="FFT")
    o.init_channel(rotation=True, omega_rot=0.12, n_rotate=2)
    m = o.mesh
    X = 2 * np.pi * m.vert_coords[0][None, None, :] / m.L[0]
    Y = np.pi * m.vert_coords[1][None, :, None] / m.L[1]
    Z = 2 * np.pi * m.vert_coords[2][:, None, None] / m.L[2]
    pert = (0.05 * np.sin(X) * np.sin(Y) ** 2 * np.cos(Z), 0.04 * np.cos(X) * np.sin(Y) ** 2 * np.sin(Z),
            0.03 * np.sin(2 * X) * np.sin(Y) ** 2 * np.cos(Z))
    for fo, d in zip((o.u, o.v, o.w), pert):
        o.backend.set_field_data(fo, o.backend.get_field_data(fo) + d)
    for it in range(1, nsteps + 1):
        o.step_channel(it)
    eo = o.monitor()
    put(out, "channel%dx%dx%d" % tuple(dims), [o.backend.get_field_data(f) for f in (o.u, o.v, o.w)], enstrophy=eo[0],
        div_max=eo[1])


CASES = {"tgv256x512x512": lambda out: tgv_step(out, "tgv256x512x512", (256, 512, 512)),
         "tgv512": lambda out: tgv_step(out, "tgv512", (512, 512, 512)), "pc512": pc512,
         "channel1024x257x512": channel}


def main():
    want = sys.argv[1:] or list(CASES)
    pinned()
    out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
    for name in want:
        t0 = time.perf_counter()
        CASES[name](out)
        print("%-22s %.1f s" % (name, time.perf_counter() - t0), flush=True)
        np.savez_compressed(PATH, **out)
    print("wrote", PATH, "(%d entries)" % len(out))


if __name__ == "__main__":
    main()
