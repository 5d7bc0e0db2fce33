"""helpers shared by the test modules"""
import io
import os
import re

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OPNAMES = ["der1st", "der1st_sym", "der2nd", "der2nd_sym", "stagder_v2p", "stagder_p2v", "interpl_v2p",
           "interpl_p2v"]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"ref_{name}.npz")))


def read_csv(name):
    with open(os.path.join(GOLDEN, f"ref_{name}.csv")) as f:
        rows = [l for l in f if not l.startswith("#")]
    return np.loadtxt(io.StringIO("".join(rows)), delimiter=",")


def namelist(g):
    """parse the namelist text stored in a fixture"""
    txt = bytes(g["cfg.namelist"]).decode()

    def get(key):
        return re.search(rf"^{key}\s*=\s*(.*)$", txt, re.M).group(1).strip()

    def nums(key, f=float):
        return [f(x.replace("d", "e")) for x in get(key).split(",")]

    def strs(key):
        return [x.strip().strip("'") for x in get(key).split(",")]

    return dict(dims=nums("dims_global", int), nproc=nums("nproc_dir", int), L=nums("L_global"),
                bcx=strs("BC_x"), bcy=strs("BC_y"), bcz=strs("BC_z"), stretching=strs("stretching"),
                beta=nums("beta"), Re=nums("Re")[0], dt=nums("dt")[0],
                time_intg=get("time_intg").strip("'"), interpl=get("interpl_scheme").strip("'"),
                der2nd=get("der2nd_scheme").strip("'"))


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def product_mesh(c, rank=0):
    from x3d2_amd.mesh import Mesh
    return Mesh(c["dims"], c["nproc"], c["L"], c["bcx"], c["bcy"], c["bcz"], c["stretching"], c["beta"],
                nrank=rank)
