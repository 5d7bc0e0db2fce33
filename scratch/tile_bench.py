#!/usr/bin/env python3
"""Micro-harness for the tile kernels at 512^3 (one process, one GPU): times, per launch,
  transeq3 y / z (local), transeq3 z HALO, tds pair modes 0 / 1 / 2 y / z (local and HALO), fix kernels.
X3D_LIB=<path to an alternative libx3d2_hip.so> selects the library (A/B inside one gpurun session).
    python scratch/tile_bench.py [--iters 20] [--only transeq|pair]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--dims", default="", help="nx,ny,nz (y-direction kernels only make sense with --only pairy)")
    args = ap.parse_args()
    alt = os.environ.get("X3D_LIB")
    from x3d2_amd import _lib
    if alt:
        _lib.LIB_PATH = alt
    os.environ["X3D_EMULATE_DECOMP"] = "z"
    import numpy as np
    import torch
    from x3d2_amd import make_tgv
    from x3d2_amd.common import DIR_X, DIR_Y, DIR_Z
    dims = tuple(int(v) for v in args.dims.split(",")) if args.dims else args.n
    case = make_tgv(dims, poisson="CG", fused=True)
    s = case.solver
    b, al = s.backend, s.backend.allocator
    s.w.fill(0.3)
    o = [al.get_block(DIR_X) for _ in range(4)]
    for f in o:
        f.fill(0.0)
    n = args.n
    npts = (dims[0] * dims[1] * dims[2]) if args.dims else n ** 3

    def timed(name, fn, passes):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.iters * 1e3
        print("%-44s %7.3f ms  %5.2f TB/s (%d passes)" % (name, ms, passes * 8.0 * npts / ms / 1e9, passes), flush=True)

    y, z = s.ydirps, s.zdirps
    nu = s.nu
    if args.only in ("", "transeq"):
        b._emulate = ""
        timed("transeq3 y local (acc)", lambda: b.transeq_planes(DIR_Y, o[0], o[1], o[2], s.u, s.v, s.w, nu, y, True, 0, n), 9)
        timed("transeq3 z local (acc)", lambda: b.transeq_planes(DIR_Z, o[0], o[1], o[2], s.u, s.v, s.w, nu, z, True, 0, n), 9)
        timed("transeq3 y local, two halves", lambda: (b.transeq_planes(DIR_Y, o[0], o[1], o[2], s.u, s.v, s.w, nu, y, True, 0, n // 2),
                                                       b.transeq_planes(DIR_Y, o[0], o[1], o[2], s.u, s.v, s.w, nu, y, True, n // 2, n // 2)), 9)
        b._emulate = "z"
        h = b.transeq_halo_begin(DIR_Z, s.u, s.v, s.w)
        timed("transeq3 z HALO main (acc)", lambda: b.transeq_halo_main(DIR_Z, o[0], o[1], o[2], s.u, s.v, s.w, nu, z, True, h), 9)
        hb = b.transeq_halo_main(DIR_Z, o[0], o[1], o[2], s.u, s.v, s.w, nu, z, True, h)
        timed("transeq z halo fix", lambda: b.transeq_halo_finish(DIR_Z, o[0], o[1], o[2], s.u, s.v, s.w, nu, z, hb), 1)
        timed("pack halos x3 + self exchange", lambda: b.transeq_halo_begin(DIR_Z, s.u, s.v, s.w), 1)
    if args.only == "x":
        x = s.xdirps
        b._emulate = ""
        timed("tds_lin x, 1 term (4 passes)", lambda: b.tds_lincomb(o[0], x.stagder_v2p, DIR_X, o[1], s.u, [0.5], [s.v]), 4)
        timed("tds_lin x, 3 terms (6 passes)", lambda: b.tds_lincomb(o[0], x.stagder_v2p, DIR_X, o[1], s.u, [0.5, 0.25, 0.1], [s.v, s.w, o[2]]), 6)
        timed("tds x (2 passes)", lambda: b.tds_apply(o[0], s.u, x.stagder_v2p, DIR_X), 2)
        timed("tds x acc (3 passes)", lambda: b.tds_apply(o[0], s.u, x.stagder_v2p, DIR_X, accumulate=True, scale=-1.0), 3)
        timed("transeq x 3-in-1 (6 passes)", lambda: b.transeq_dir(DIR_X, o[0], o[1], o[2], s.u, s.v, s.w, nu, x, accumulate=False), 6)
        timed("lincomb 3 terms (5 passes)", lambda: b.lincomb(o[1], s.u, [0.5, 0.25, 0.1], [s.v, s.w, o[2]]), 5)
        return
    if args.only == "pairy":
        d, dp, nm, nz_ = DIR_Y, y, "y", dims[2]
        b._emulate = ""
        j0 = (0, o[0], None, s.u, s.v, dp.interpl_v2p, dp.stagder_v2p)
        j1 = (1, o[0], o[1], s.u, None, dp.interpl_p2v, dp.stagder_p2v)
        j2 = (2, o[0], None, s.u, None, dp.interpl_v2p, None)
        timed(f"pair mode 0 {nm} local", lambda: b.tds_tile_planes(d, j0, 0, nz_), 3)
        timed(f"pair mode 1 {nm} local", lambda: b.tds_tile_planes(d, j1, 0, nz_), 3)
        timed(f"single (mode 2) {nm} local tile", lambda: b.tds_tile_planes(d, j2, 0, nz_), 2)
        timed(f"single {nm} K1e", lambda: b.tds_apply(o[0], s.u, dp.interpl_v2p, d), 2)
        timed("transeq3 y local (acc)", lambda: b.transeq_planes(DIR_Y, o[0], o[1], o[2], s.u, s.v, s.w, nu, y, True, 0, nz_), 9)
        return
    if args.only in ("", "pair"):
        for d, dp, nm in ((DIR_Y, y, "y"), (DIR_Z, z, "z")):
            b._emulate = ""
            j0 = (0, o[0], None, s.u, s.v, dp.interpl_v2p, dp.stagder_v2p)
            j1 = (1, o[0], o[1], s.u, None, dp.interpl_p2v, dp.stagder_p2v)
            j2 = (2, o[0], None, s.u, None, dp.interpl_v2p, None)
            timed(f"pair mode 0 {nm} local", lambda: b.tds_tile_planes(d, j0, 0, n), 3)
            timed(f"pair mode 1 {nm} local", lambda: b.tds_tile_planes(d, j1, 0, n), 3)
            timed(f"single (mode 2) {nm} local tile", lambda: b.tds_tile_planes(d, j2, 0, n), 2)
            timed(f"single {nm} K1e", lambda: b.tds_apply(o[0], s.u, dp.interpl_v2p, d), 2)
        b._emulate = "z"
        d, dp = DIR_Z, z
        for k, j in enumerate(((0, o[0], None, s.u, s.v, dp.interpl_v2p, dp.stagder_v2p),
                               (1, o[0], o[1], s.u, None, dp.interpl_p2v, dp.stagder_p2v),
                               (2, o[0], None, s.u, None, dp.interpl_v2p, None))):
            h = b.tds_halo_begin(d, j, k)
            timed(f"pair mode {j[0]} z HALO main", lambda: b.tds_halo_main(d, j, k, h), 3 if j[0] < 2 else 2)
            hb = b.tds_halo_main(d, j, k, h)
            timed(f"pair mode {j[0]} z halo fix", lambda: b.tds_halo_finish(d, j, k, hb), 1)


if __name__ == "__main__":
    main()
