#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: monitoring traces of north-star length from the oracle (the CPU restatement of the reference's
OpenMP path, pinned to the flang-built reference by tests/test_oracle_vs_reference.py and to the survey's TGV 64^3
enstrophy values by oracle/gen_trace_fixture.py), written as small CSV fixtures the GPU tests read:

    tests/golden/oracle_trace_tgv64_rk3_1000.csv      TGV 64^3,  RK3, dt 1e-3, FFT Poisson, 1000 steps (t = 1), every 50
    tests/golden/oracle_trace_tgv128_rk3_200.csv      TGV 128^3, RK3, dt 1e-3, FFT Poisson,  200 steps, every 25
    tests/golden/oracle_trace_channel_64x65x32_200.csv channel 64 x 65 x 32, y Dirichlet + top-bottom stretching (beta
                                                      0.259065151), Re 4200, dt 5e-3, RK3, rotation forcing, 010 Poisson
                                                      solve with the pentadiagonal spectral operator, 200 steps, every 25

Columns: time, enstrophy, kinetic energy, max |div u|, mean |div u| (+ bulk velocity for the channel) -- what
monitoring_t writes per output step (/root/reference/src/postprocess/monitoring.f90:46-90: enstrophy and the
divergence norms) plus the scalar_product(u, u)-based kinetic energy the north star asks for.

The channel's initial condition: the reference's laminar profile 1 - y^2 (src/case/channel.f90:139-189 with
init_noise = 0; its noise is an UNSEEDED random_number) plus the deterministic three-dimensional perturbation of
`channel_perturbation` below, so that both sides start from the same field and the flow is not one-dimensional.

    OMP_NUM_THREADS=2 python oracle/gen_long_traces.py tgv64 | tgv128 | channel | all
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    "tgv64": dict(n=64, steps=1000, every=50, file="oracle_trace_tgv64_rk3_1000.csv"),
    "tgv128": dict(n=128, steps=200, every=25, file="oracle_trace_tgv128_rk3_200.csv"),
    "channel": dict(dims=(64, 65, 32), L=(4.0, 2.0, 2.0), beta=0.259065151, Re=4200.0, dt=5e-3, omega_rot=0.12,
                    steps=200, every=25, file="oracle_trace_channel_64x65x32_200.csv"),
}


def channel_perturbation(xv, yv, zv, L):
    """(du, dv, dw) added to the laminar channel profile: smooth, zero on both walls, three-dimensional.
    xv, yv, zv: vertex coordinates of the three directions (1-D arrays); fields are [z][y][x]"""
    x = xv[None, None, :] * (2.0 * np.pi / L[0])
    z = zv[:, None, None] * (2.0 * np.pi / L[2])
    eta = yv[None, :, None] - L[1] / 2.0        # wall-normal coordinate in [-1, 1] for L_y = 2
    wall = (1.0 - eta * eta) ** 2               # vanishes at both walls with zero slope
    du = 0.05 * wall * np.sin(x) * np.cos(2.0 * z)
    dv = 0.03 * wall * np.cos(x) * np.sin(z) * eta
    dw = 0.04 * wall * np.sin(2.0 * x) * np.sin(z)
    return du, dv, dw


def kinetic_energy(s):
    b = s.backend
    return 0.5 * (b.scalar_product(s.u, s.u) + b.scalar_product(s.v, s.v) + b.scalar_product(s.w, s.w)) / s.ngrid


def run_tgv(c):
    from oracle import x3d_oracle as orc
    twopi = 6.283185307179586
    n = c["n"]
    mesh = orc.Mesh([n] * 3, [1, 1, 1], [twopi] * 3, ["periodic"] * 2, ["periodic"] * 2, ["periodic"] * 2)
    s = orc.Solver(mesh, Re=1600.0, dt=1e-3, time_intg="RK3", poisson="FFT")
    s.init_tgv()
    rows = [(0.0,) + _mon(s)]
    t0 = time.time()
    for it in range(1, c["steps"] + 1):
        s.step()
        if it % c["every"] == 0:
            rows.append((it * 1e-3,) + _mon(s))
            print(c["file"], it, rows[-1], "%.0f s" % (time.time() - t0), flush=True)
    return rows, ("# TGV %d^3, RK3, dt=1e-3, Re=1600, compact6/classic, FFT Poisson; rows every %d steps\n"
                  "# time, enstrophy, kinetic_energy, div_u_max, div_u_mean\n" % (n, c["every"]))


def _mon(s):
    ens, mx, mean = s.monitor()
    return (ens, kinetic_energy(s), mx, mean)


def run_channel(c):
    from oracle import x3d_oracle as orc
    from oracle.x3d_oracle import CELL
    mesh = orc.Mesh(list(c["dims"]), [1, 1, 1], list(c["L"]), ["periodic"] * 2, ["dirichlet"] * 2, ["periodic"] * 2,
                    stretching=["uniform", "top-bottom", "uniform"], beta=[1.0, c["beta"], 1.0])
    s = orc.Solver(mesh, Re=c["Re"], dt=c["dt"], time_intg="RK3", poisson="FFT")
    s.init_channel(rotation=True, omega_rot=c["omega_rot"], n_rotate=10 ** 9)
    b = s.backend
    pert = channel_perturbation(mesh.vert_coords[0], mesh.vert_coords[1], mesh.vert_coords[2], c["L"])
    for f, d in zip((s.u, s.v, s.w), pert):
        b.set_field_data(f, b.get_field_data(f) + d)
    ncell = float(np.prod(mesh.get_dims(CELL, glob=True)))

    def mon():
        return _mon(s) + (b.field_volume_integral(s.u) / ncell,)
    rows = [(0.0,) + mon()]
    t0 = time.time()
    for it in range(1, c["steps"] + 1):
        s.step_channel(it)
        if it % c["every"] == 0:
            rows.append((it * c["dt"],) + mon())
            print(c["file"], it, rows[-1], "%.0f s" % (time.time() - t0), flush=True)
    return rows, ("# channel %d x %d x %d verts, L = %s, y Dirichlet + top-bottom stretching beta %.9f, Re=%g, dt=%g, RK3,\n"
                  "# rotation forcing omega = %g every step, 010 Poisson (stretched: pentadiagonal spectral solve); initial field =\n"
                  "# laminar 1 - y^2 + oracle/gen_long_traces.py::channel_perturbation; rows every %d steps\n"
                  "# time, enstrophy, kinetic_energy, div_u_max, div_u_mean, bulk_velocity\n"
                  % (c["dims"] + (c["L"], c["beta"], c["Re"], c["dt"], c["omega_rot"], c["every"])))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for name, c in CASES.items():
        if which not in ("all", name):
            continue
        rows, hdr = (run_channel if name == "channel" else run_tgv)(c)
        path = os.path.join(ROOT, "tests", "golden", c["file"])
        with open(path, "w") as f:
            f.write("# generated by oracle/gen_long_traces.py %s (oracle restatement; numpy DFT in place of 2decomp&FFT)\n" % name)
            f.write(hdr)
            for r in rows:
                f.write(", ".join("%.16e" % v for v in r) + "\n")
        print("wrote", path)


if __name__ == "__main__":
    main()
