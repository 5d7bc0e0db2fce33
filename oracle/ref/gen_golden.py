#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: generate tests/golden/* from the REAL reference.

Runs the binaries built by oracle/ref/Makefile (the reference's own Fortran
sources compiled in place from /root/reference, plus our dump driver) and
stores inputs + outputs as small fixtures.  Only runs where /root/reference is
mounted; the fixtures it writes are committed so that the GPU box never needs
the reference.

    make -C oracle/ref && python oracle/ref/gen_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_OUT = os.path.normpath(os.path.join(HERE, "..", "_ref"))
GOLDEN = os.path.normpath(os.path.join(HERE, "..", "..", "tests", "golden"))
MPIEXEC = "/opt/conda/bin/mpiexec"

NML = """&domain_settings
flow_case_name = '{case}'
L_global = {L}
dims_global = {dims}
nproc_dir = {nproc}
BC_x = {bcx}
BC_y = {bcy}
BC_z = {bcz}
stretching = {stretching}
beta = {beta}
/End
&solver_params
Re = {Re}
time_intg = '{time_intg}'
dt = {dt}
n_iters = {n_iters}
n_output = {n_output}
poisson_solver_type = 'CG'
der1st_scheme = 'compact6'
der2nd_scheme = '{der2nd}'
interpl_scheme = '{interpl}'
stagder_scheme = 'compact6'
/End
&channel_nml
init_noise = 0. 0. 0.
inlet_noise = 0. 0. 0.
rotation = {rotation}
omega_rot = {omega_rot}
n_rotate = {n_rotate}
/End
"""

TWO_PI = "6.283185307179586d0"


def cfg(**kw):
    d = dict(case="tgv", L=f"{TWO_PI}, {TWO_PI}, {TWO_PI}", dims="16, 24, 20",
             nproc="1, 1, 1", bcx="'periodic', 'periodic'",
             bcy="'periodic', 'periodic'", bcz="'periodic', 'periodic'",
             stretching="'uniform', 'uniform', 'uniform'", beta="1d0, 1d0, 1d0",
             Re="1600d0", time_intg="RK3", dt="0.001d0", n_iters=4, n_output=2,
             der2nd="compact6", interpl="classic", rotation="F", omega_rot="0d0", n_rotate=0)
    d.update(kw)
    return d


# dump_golden configurations (name -> namelist values)
DUMPS = {
    # all-periodic, sizes not multiples of the reference's SZ=16 (padding path)
    "p000_rk3": cfg(dims="12, 20, 16"),
    "p000_ab3": cfg(dims="12, 20, 16", time_intg="AB3"),
    # same global problem on 2 ranks, split in z and in y (DistD2 across ranks)
    "p000_rk3_z2": cfg(dims="8, 12, 32", nproc="1, 1, 2"),
    "p000_rk3_z1": cfg(dims="8, 12, 32", nproc="1, 1, 1"),
    "p000_rk3_y2": cfg(dims="8, 32, 12", nproc="1, 2, 1"),
    "p000_rk3_y1": cfg(dims="8, 32, 12", nproc="1, 1, 1"),
    # channel-like: Dirichlet walls in y on a stretched mesh
    "c010_rk3": cfg(dims="12, 25, 12", L="4d0, 2d0, 2d0",
                    bcy="'dirichlet', 'dirichlet'",
                    stretching="'uniform', 'top-bottom', 'uniform'",
                    beta="1d0, 0.259065151d0, 1d0", Re="4200d0", dt="0.005d0"),
    # 010 Poisson inputs: uniform y, 'bottom' (full pentadiagonal) and 'centred' stretching
    "c010u_rk3": cfg(dims="12, 17, 8", L="4d0, 2d0, 2d0", bcy="'dirichlet', 'dirichlet'",
                     Re="4200d0", dt="0.005d0"),
    "c010b_rk3": cfg(dims="12, 17, 8", L="4d0, 2d0, 2d0", bcy="'dirichlet', 'dirichlet'",
                     stretching="'uniform', 'bottom', 'uniform'", beta="1d0, 0.5d0, 1d0",
                     Re="4200d0", dt="0.005d0"),
    "c010c_rk3": cfg(dims="12, 17, 8", L="4d0, 2d0, 2d0", bcy="'dirichlet', 'dirichlet'",
                     stretching="'uniform', 'centred', 'uniform'", beta="1d0, 1.3d0, 1d0",
                     Re="4200d0", dt="0.005d0"),
    # every non-periodic closure: Neumann x, Dirichlet y, Neumann z,
    # 'optimised' interpolation (hyperviscous der2nd cannot be reached through
    # the reference's allocate_tdsops: it never passes c_nu/nu0_nu)
    "n111_rk2": cfg(dims="13, 21, 11", L="3d0, 2d0, 2.5d0",
                    bcx="'neumann', 'neumann'", bcy="'dirichlet', 'dirichlet'",
                    bcz="'neumann', 'neumann'", time_intg="RK2",
                    interpl="optimised",
                    stretching="'uniform', 'centred', 'uniform'",
                    beta="1d0, 1.3d0, 1d0"),
}

# full xcompact runs (TGV, Poisson off = the reference's 'CG' placeholder)
TRACES = {
    "tgv32_rk3_nopoisson": cfg(dims="32, 32, 32", n_iters=6, n_output=2),
    "tgv32_ab3_nopoisson": cfg(dims="32, 32, 32", time_intg="AB3", n_iters=6, n_output=2),
    "tgv64_rk3_nopoisson": cfg(dims="64, 64, 64", n_iters=4, n_output=2),
    "tgv32_rk3_nopoisson_z2": cfg(dims="32, 32, 32", nproc="1, 1, 2", n_iters=6, n_output=2),
    # channel case hooks (bulk-velocity shift, rotation forcing switched off at iter 3, wall stamping),
    # deterministic: no noise; stretched top-bottom mesh
    "channel17_rk3_nopoisson": cfg(case="channel", dims="16, 17, 12", L="4d0, 2d0, 2d0",
                                   bcy="'dirichlet', 'dirichlet'",
                                   stretching="'uniform', 'top-bottom', 'uniform'",
                                   beta="1d0, 0.259065151d0, 1d0", Re="4200d0", dt="0.005d0",
                                   rotation="T", omega_rot="0.12d0", n_rotate=3, n_iters=6, n_output=2),
}


def read_bin(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    p = 0
    while p < len(data):
        (nl,) = struct.unpack_from("<i", data, p); p += 4
        name = data[p:p + nl].decode(); p += nl
        (rk,) = struct.unpack_from("<i", data, p); p += 4
        dims = struct.unpack_from("<%di" % rk, data, p); p += 4 * rk
        n = int(np.prod(dims))
        arr = np.frombuffer(data, dtype="<f8", count=n, offset=p).copy(); p += 8 * n
        # Fortran order (i fastest) -> numpy array indexed [k, j, i]
        out[name] = arr.reshape(dims[::-1])
    return out


def nranks(c):
    return int(np.prod([int(x) for x in c["nproc"].split(",")]))


def run(exe, c, workdir, extra=()):
    nml = os.path.join(workdir, "input.x3d")
    with open(nml, "w") as f:
        f.write(NML.format(**c))
    n = nranks(c)
    cmd = [os.path.join(REF_OUT, exe), nml, *extra]
    if n > 1:
        cmd = [MPIEXEC, "-n", str(n)] + cmd
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, cwd=workdir, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:] + r.stderr[-3000:])
        raise SystemExit(f"{exe} failed for {c}")
    return r.stdout


FIELD_PREFIXES = ("in.", "tds.", "transeq", "div.div_u", "grad.", "curl.i",
                  "curl.j", "curl.k", "step1.", "step2.")


def stitch(per_rank):
    """merge rank-local dumps: 3-D fields are placed by n_offset, everything
    else is taken from rank 0 (rank-local 1-D arrays keep a .r<rank> suffix)."""
    r0 = per_rank[0]
    if len(per_rank) == 1:
        return dict(r0)
    out = {}
    nproc_dir = r0["meta.nproc_dir"].astype(int)
    for name, a0 in r0.items():
        if a0.ndim == 3 and name.startswith(FIELD_PREFIXES):
            # global extent = sum of local extents along decomposed dirs
            shp = list(a0.shape)  # [k, j, i]
            ext = {0: {}, 1: {}, 2: {}}
            for pr in per_rank:
                rd = pr["meta.nrank_dir"].astype(int)  # x,y,z
                a = pr[name]
                for ax, d in ((0, 2), (1, 1), (2, 0)):
                    ext[ax][rd[d]] = a.shape[ax]
            gshape = [sum(ext[ax].values()) for ax in range(3)]
            g = np.zeros(gshape)
            for pr in per_rank:
                rd = pr["meta.nrank_dir"].astype(int)
                a = pr[name]
                off = []
                for ax, d in ((0, 2), (1, 1), (2, 0)):
                    off.append(sum(ext[ax][r] for r in range(rd[d])))
                g[off[0]:off[0] + a.shape[0], off[1]:off[1] + a.shape[1],
                  off[2]:off[2] + a.shape[2]] = a
            out[name] = g
        else:
            out[name] = a0
            for r, pr in enumerate(per_rank[1:], 1):
                if name in pr and (pr[name].shape != a0.shape or not np.array_equal(pr[name], a0)):
                    out[f"{name}.r{r}"] = pr[name]
    return out


# keep fixtures small: which records each dump keeps (None = all)
KEEP = {
    "p000_ab3": ("meta.", "in.", "step1.", "step2."),
    "p000_rk3_z2": ("meta.", "in.", "tds.z.", "transeq.", "div.", "grad.", "curl.", "step2.", "species."),
    "p000_rk3_z1": ("meta.", "in.", "tds.z.", "transeq.", "div.", "grad.", "curl.", "step2.", "species."),
    "p000_rk3_y2": ("meta.", "in.", "tds.y.", "transeq.", "div.", "grad.", "curl.", "step2.", "species."),
    "p000_rk3_y1": ("meta.", "in.", "tds.y.", "transeq.", "div.", "grad.", "curl.", "step2.", "species."),
    "c010u_rk3": ("meta.", "in.", "spec.", "step2."),
    "c010b_rk3": ("meta.", "in.", "spec.", "step2."),
    "c010c_rk3": ("meta.", "in.", "spec.", "step2."),
}


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    for name, c in DUMPS.items():
        with tempfile.TemporaryDirectory() as wd:
            run("dump_golden", c, wd, extra=[os.path.join(wd, "dump")])
            per_rank = [read_bin(os.path.join(wd, f"dump.{r}.bin")) for r in range(nranks(c))]
        merged = stitch(per_rank)
        if KEEP.get(name):
            merged = {k: v for k, v in merged.items() if k.startswith(KEEP[name])}
        merged["cfg.namelist"] = np.frombuffer(NML.format(**c).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(GOLDEN, f"ref_{name}.npz"), **merged)
        print(f"ref_{name}.npz: {len(merged)} arrays")
    # compact10_penta first derivative (the reference's pentadiagonal scheme: tdsops_init + exec_dist_penta_*),
    # set-ups of tests/verification/test_omp_penta.f90 -- oracle/ref/drivers/dump_penta.f90
    with tempfile.TemporaryDirectory() as wd:
        out = os.path.join(wd, "penta.bin")
        r = subprocess.run([os.path.join(REF_OUT, "dump_penta"), out], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
            raise SystemExit("dump_penta failed")
        penta = read_bin(out)
    np.savez_compressed(os.path.join(GOLDEN, "ref_penta.npz"), **penta)
    print(f"ref_penta.npz: {len(penta)} arrays")
    for name, c in TRACES.items():
        with tempfile.TemporaryDirectory() as wd:
            run("xcompact", c, wd)
            with open(os.path.join(wd, "monitoring.csv")) as f:
                txt = f.read()
        with open(os.path.join(GOLDEN, f"ref_{name}.csv"), "w") as f:
            f.write("# generated by oracle/ref/gen_golden.py from the reference's xcompact "
                    "(OMP backend, poisson_solver_type='CG' = no pressure solve)\n")
            f.write("# " + " | ".join(f"{k}={v}" for k, v in c.items()) + "\n")
            f.write(txt)
        print(f"ref_{name}.csv")


if __name__ == "__main__":
    main()
