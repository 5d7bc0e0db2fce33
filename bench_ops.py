#!/usr/bin/env python3
"""Isolated-operator benchmark, the counterpart of the reference's tests/performance/perf_cuda_tridiag.f90 and
perf_cuda_transeq.f90: tds_solve with the periodic compact6 second derivative and the
fused transport-equation component on 512^2 pencils of n points, input sin(j dx) as there, n_warmup untimed +
n_iters timed launches.  Prints one JSON line per case: achieved GB/s on ALGORITHMIC bytes
(tds_solve 16 B/DoF, transeq component 24 B/DoF, 16 when conv = u) and the reference's own convention
(the "consumed bandwidth" its perf tests assume: 6 passes = 48 B for tds_solve, 16 passes = 128 B for transeq).

    python bench_ops.py [--n 256,512,1024] [--iters 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", default="256,512,1024")  # the sizes of perf_cuda_tridiag
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    args = ap.parse_args()
    import torch
    from x3d2_amd import Mesh
    from x3d2_amd.backend import HipBackend
    from x3d2_amd.common import DIR_X, VERT
    from x3d2_amd.solver import Solver, SolverConfig
    twopi = 6.283185307179586
    per = ("periodic",) * 2
    for n in (int(v) for v in args.n.split(",")):
        for d, dname in ((1, "x"), (2, "y"), (3, "z")):
            dims = [512, 512, 512]
            dims[d - 1] = n
            mesh = Mesh(tuple(dims), (1, 1, 1), (twopi,) * 3, per, per, per)
            s = Solver(HipBackend(mesh), mesh, SolverConfig(poisson_solver_type="CG", fused=True))
            b, al = s.backend, s.backend.allocator
            dirps = (s.xdirps, s.ydirps, s.zdirps)[d - 1]
            nx, ny, nz = dims
            idx = [np.arange(m) for m in (nz, ny, nx)]
            grid = np.meshgrid(*idx, indexing="ij")[3 - d]
            dx = twopi / n
            for f, fn in ((s.u, np.sin), (s.v, np.cos)):
                f.set_data_loc(VERT)
                b.set_field_data(f, fn(grid * dx))
            s.w.set_data_loc(VERT)
            b.set_field_data(s.w, np.cos(grid * dx))
            out = [al.get_block(DIR_X, VERT) for _ in range(3)]
            dof = nx * ny * nz

            def timed(fn):
                for _ in range(args.warmup):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / args.iters

            t = timed(lambda: b.tds_apply(out[0], s.u, dirps.der2nd, d))
            print(json.dumps({"op": "tds_solve second-deriv compact6 periodic", "dir": dname, "n": n,
                              "pencils": dof // n, "ms": t * 1e3, "GBs_algorithmic_16B": 16 * dof / t / 1e9,
                              "GBs_reference_convention_48B": 48 * dof / t / 1e9}))
            # the three components of one direction (advecting velocity = component d)
            t = timed(lambda: b.transeq_dir(d, out[0], out[1], out[2], s.u, s.v, s.w, 1.0, dirps, accumulate=False))
            print(json.dumps({"op": "transeq (3 components, fused subs)", "dir": dname, "n": n,
                              "pencils": dof // n, "ms": t * 1e3, "ms_per_component": t * 1e3 / 3,
                              "GBs_algorithmic_64B": 64 * dof / t / 1e9,
                              "GBs_reference_convention_384B": 384 * dof / t / 1e9}))
            del s, b, al, out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
