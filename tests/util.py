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


def read_trace_fixture(name="oracle_tgv64_rk3_fft"):
    """rows (time, enstrophy, div_u_max, div_u_mean, survey_enstrophy, survey_div_u_max) of a fixture written
    by oracle/gen_trace_fixture.py"""
    with open(os.path.join(GOLDEN, name + ".csv")) as f:
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


class ThreadComm:
    """lock-step exchange between oracle solvers that run as threads of this process (the oracle's comm
    interface: sendrecv / allreduce).  send_s goes to prev (arrives as its recv_e), send_e to next."""

    def __init__(self, rank, size, shared):
        self.rank, self.size, self.sh = rank, size, shared

    @staticmethod
    def shared(size):
        import threading
        return {"barrier": threading.Barrier(size), "box": {}}

    def sendrecv(self, send_s, send_e, prev, nxt):
        box, bar = self.sh["box"], self.sh["barrier"]
        box[(self.rank, "s")], box[(self.rank, "e")] = send_s, send_e
        bar.wait()
        recv_s, recv_e = box[(int(prev), "e")].copy(), box[(int(nxt), "s")].copy()
        bar.wait()
        return recv_s, recv_e

    def allreduce(self, x, op="sum"):
        box, bar = self.sh["box"], self.sh["barrier"]
        box[(self.rank, "r")] = x
        bar.wait()
        vals = [box[(r, "r")] for r in range(self.size)]  # fixed order: every rank gets the same bits
        bar.wait()
        return max(vals) if op == "max" else sum(vals)


def run_rank_threads(size, body):
    """body(rank, comm) -> result, one thread per rank; returns the list of results (exceptions re-raised)"""
    import threading
    sh = ThreadComm.shared(size)
    out, err = [None] * size, [None] * size

    def run(r):
        try:
            out[r] = body(r, ThreadComm(r, size, sh))
        except BaseException as e:  # noqa: BLE001 -- re-raised in the parent
            err[r] = e
            sh["barrier"].abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(size)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, __import__("threading").BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out


def stitch_ranks(parts, offsets, name):
    """assemble rank-local [k, j, i] arrays into the global one by their n_offset (x, y, z)"""
    ext = [max(int(o[d]) + p[name].shape[2 - d] for p, o in zip(parts, offsets)) for d in range(3)]
    full = np.zeros((ext[2], ext[1], ext[0]))
    for p, o in zip(parts, offsets):
        a = p[name]
        ox, oy, oz = (int(v) for v in o)
        full[oz:oz + a.shape[0], oy:oy + a.shape[1], ox:ox + a.shape[2]] = a
    return full
