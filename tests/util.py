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


def BATTERY_FIELDS(dn):
    """dn: the decomposed direction(s), "y", "z" or "yz" (a 2-D pencil split runs the operators of both)"""
    return [f"tds.{d}.{op}" for d in dn for op in OPNAMES] + ["transeq.du", "transeq.dv", "transeq.dw", "div.div_u", "grad.dpdx",
            "grad.dpdy", "grad.dpdz", "curl.i", "curl.j", "curl.k", "step2.u", "step2.v", "step2.w"]


def oracle_battery(g, nranks):
    """the operator sequence oracle/ref/drivers/dump_golden.f90 drives the reference with, on the ORACLE with
    nranks ranks (threads exchanging in lock step) decomposed as the namelist in g says; inputs g["in.*"] are
    global arrays.  Returns (dict of stitched global fields, list of per-rank dicts)."""
    from oracle import x3d_oracle as orc
    c = namelist(g)
    ds = [k + 1 for k, p in enumerate(c["nproc"]) if int(p) > 1]  # every decomposed direction ([1, py, pz]: y and z)
    dn = "".join("xyz"[d - 1] for d in ds)

    def local(key, mesh):
        ox, oy, oz = (int(v) for v in mesh.n_offset)
        nx, ny, nz = (int(v) for v in mesh.vert_dims)
        return g[key][oz:oz + nz, oy:oy + ny, ox:ox + nx]

    def body(rank, comm):
        mesh = orc.Mesh(c["dims"], c["nproc"], c["L"], c["bcx"], c["bcy"], c["bcz"], c["stretching"], c["beta"],
                        rank=rank)
        s = orc.Solver(mesh, Re=c["Re"], dt=c["dt"], time_intg=c["time_intg"], poisson="CG", comm=comm,
                       interpl=c["interpl"], der2nd=c["der2nd"])
        b = s.backend
        out = {}
        for f, k in ((s.u, "in.u"), (s.v, "in.v"), (s.w, "in.w")):
            f.data_loc = orc.VERT
            b.set_field_data(f, local(k, s.mesh))
        for d in ds:
            dp = (s.xdirps, s.ydirps, s.zdirps)[d - 1]
            for op in OPNAMES:
                src = b.get_block(orc.DIR_X, orc.VERT)
                b.veccopy(src, s.u)
                if op.endswith("p2v"):
                    src.data_loc = orc.move_data_loc(orc.VERT, d, 1)
                a, o = b.get_block(d), b.get_block(d)
                b.reorder(a, src, orc.RDR[(1, d)])
                b.tds_solve(o, a, getattr(dp, op))
                out[f"tds.{'xyz'[d - 1]}.{op}"] = b.get_field_data(o)
        rhs = [b.get_block(orc.DIR_X) for _ in range(3)]
        s.transeq(rhs, [s.u, s.v, s.w])
        for f, k in zip(rhs, ("du", "dv", "dw")):
            out["transeq." + k] = b.get_field_data(f)
        spec = b.get_block(orc.DIR_X, orc.VERT)
        b.set_field_data(spec, local("in.s", s.mesh))
        srhs = b.get_block(orc.DIR_X)
        s.transeq_species([srhs], [s.u, s.v, s.w, spec], [0.37 * s.nu])
        out["species.rhs"] = b.get_field_data(srhs, orc.VERT)
        div_u = b.get_block(orc.DIR_Z)
        s.divergence_v2p(div_u, s.u, s.v, s.w)
        out["div.div_u"] = b.get_field_data(div_u)
        out["div.maxmean"] = np.array(b.field_max_mean(div_u))
        s.gradient_p2v(*rhs, div_u)
        for f, k in zip(rhs, ("dpdx", "dpdy", "dpdz")):
            out["grad." + k] = b.get_field_data(f)
        for f in rhs:
            f.data_loc = orc.VERT
        s.curl(*rhs, s.u, s.v, s.w)
        for f, k in zip(rhs, "ijk"):
            out["curl." + k] = b.get_field_data(f)
        out["curl.enstrophy"] = np.array([0.5 * sum(b.scalar_product(f, f) for f in rhs) / s.ngrid])
        for it in range(2 * s.time_integrator.nstage):
            r3 = [b.get_block(orc.DIR_X) for _ in range(3)]
            s.transeq(r3, [s.u, s.v, s.w])
            s.time_integrator.step([s.u, s.v, s.w], r3, s.dt)
        for f, k in ((s.u, "u"), (s.v, "v"), (s.w, "w")):
            out["step2." + k] = b.get_field_data(f)
        return out, s.mesh.n_offset.copy()

    res = run_rank_threads(nranks, body)
    parts, offs = [r[0] for r in res], [r[1] for r in res]
    full = {k: stitch_ranks(parts, offs, k) for k in BATTERY_FIELDS(dn) + ["species.rhs"]}
    return full, parts


NML = """&domain_settings
flow_case_name = 'tgv'
L_global = 6.283185307179586d0, 6.283185307179586d0, 6.283185307179586d0
dims_global = {dims}
nproc_dir = {nproc}
BC_x = 'periodic', 'periodic'
BC_y = 'periodic', 'periodic'
BC_z = 'periodic', 'periodic'
stretching = 'uniform', 'uniform', 'uniform'
beta = 1d0, 1d0, 1d0
/End
&solver_params
Re = 1600d0
time_intg = 'RK3'
dt = 0.001d0
interpl_scheme = 'classic'
der2nd_scheme = 'compact6'
/End
"""


def synthetic_case(dims, nproc, seed=7):
    """a fixture-shaped dict (namelist + global inputs) for the operator battery: smooth fields + 10 % noise"""
    rng = np.random.default_rng(seed)
    nx, ny, nz = dims
    x = np.arange(nx)[None, None, :] * (2 * np.pi / nx)
    y = np.arange(ny)[None, :, None] * (2 * np.pi / ny)
    z = np.arange(nz)[:, None, None] * (2 * np.pi / nz)
    g = {"cfg.namelist": np.frombuffer(NML.format(dims=", ".join(map(str, dims)),
                                                  nproc=", ".join(map(str, nproc))).encode(), dtype=np.uint8)}
    for k, (a, b_, c) in zip(("in.u", "in.v", "in.w", "in.s"), ((1, 2, 1), (2, 1, 3), (1, 1, 2), (3, 2, 1))):
        g[k] = np.sin(a * x + 0.3) * np.cos(b_ * y) * np.cos(c * z + 0.1) + 0.1 * rng.standard_normal((nz, ny, nx))
    return g


def hash_noise(gi, gj, gk, salt):
    """deterministic pseudo-noise in [-0.5, 0.5) from GLOBAL grid indices (broadcastable integer arrays): every rank
    of a decomposed run and the single-rank oracle evaluate the same expression on the same host, so they get the same
    bits without anybody holding the global array"""
    t = np.sin(gi * 12.9898 + gj * 78.233 + gk * 37.719 + salt * 0.618) * 43758.5453
    return t - np.floor(t) - 0.5


def noisy_tgv(dims, offset, local_dims, amp=0.1):
    """Taylor-Green velocity + amp * hash noise on the box [offset, offset + local_dims) of a 2 pi periodic grid of
    `dims` vertices; arrays [k, j, i]"""
    ox, oy, oz = (int(v) for v in offset)
    nx, ny, nz = (int(v) for v in local_dims)
    gi = np.arange(ox, ox + nx)[None, None, :]
    gj = np.arange(oy, oy + ny)[None, :, None]
    gk = np.arange(oz, oz + nz)[:, None, None]
    twopi = 6.283185307179586
    x, y, z = gi * (twopi / dims[0]), gj * (twopi / dims[1]), gk * (twopi / dims[2])
    u = np.sin(x) * np.cos(y) * np.cos(z) + amp * hash_noise(gi, gj, gk, 1)
    v = -np.cos(x) * np.sin(y) * np.cos(z) + amp * hash_noise(gi, gj, gk, 2)
    w = amp * hash_noise(gi, gj, gk, 3) + 0.0 * x * y
    return u, v, w


# ---- signatures of large fields (round 6): the 512^3 / 1024 x 257 x 512 oracle runs of the GPU suite (25 s ... 60 s of host
# time each on the GPU box) are paid ONCE, by oracle/gen_step_fixtures.py, which stores these signatures of the oracle's
# fields under tests/golden/oracle_big_steps.npz; the GPU tests then compare the HIP fields' signatures with them
def _sig_axes(shape):
    return [np.unique(np.linspace(0, n - 1, min(n, 24)).astype(np.int64)) for n in shape]


def _sig_weights(shape):
    """integer-hash weights in [-1, 1): a permutation of rows or planes changes the weighted sum"""
    nz, ny, nx = shape
    gk = np.arange(nz, dtype=np.int64)[:, None, None]
    gj = np.arange(ny, dtype=np.int64)[None, :, None]
    gi = np.arange(nx, dtype=np.int64)[None, None, :]
    return (((gi * 7 + gj * 13 + gk * 29 + (gi * gj + gk) * 3) % 64) - 31.5) / 32.0


def field_signature(a):
    """a: [nz, ny, nx] -> {sample (a 24^3 lattice incl. the corners), sum, sumabs, sumsq, wsum, absmax}"""
    a = np.asarray(a, dtype=np.float64)
    iz, iy, ix = _sig_axes(a.shape)
    return {"sample": a[np.ix_(iz, iy, ix)].copy(), "sum": np.float64(a.sum()), "sumabs": np.float64(np.abs(a).sum()),
            "sumsq": np.float64(np.vdot(a, a)), "wsum": np.float64((a * _sig_weights(a.shape)).sum()),
            "absmax": np.float64(np.abs(a).max())}


def assert_signature(a, sig, rel, what="", scale=None):
    """the field a against a stored signature: every sampled point within rel * absmax; the plain and the hash-weighted
    sums within 50 sqrt(n) rel absmax (round-off differences of up to rel * absmax per point add up like a random walk; a
    coherent deviation of a tenth of that per point is caught); the sum of squares within 2 rel absmax sum|a|"""
    got = field_signature(a)
    scale = max(float(sig["absmax"]), 1e-300) if scale is None else float(scale)  # (scale given: an absolute tolerance)
    assert got["sample"].shape == sig["sample"].shape, what
    assert np.max(np.abs(got["sample"] - sig["sample"])) <= rel * scale, (what, "sample", float(np.max(np.abs(got["sample"] - sig["sample"]))))
    n = a.size
    for k in ("sum", "wsum"):
        assert abs(got[k] - float(sig[k])) <= 50.0 * np.sqrt(n) * rel * scale, (what, k, float(got[k]), float(sig[k]))
    assert abs(got["sumsq"] - float(sig["sumsq"])) <= 2.0 * rel * scale * float(sig["sumabs"]) + 1e-300, (what, "sumsq")
    assert abs(got["absmax"] - float(sig["absmax"])) <= rel * scale, (what, "absmax")


def load_big_steps():
    p = os.path.join(GOLDEN, "oracle_big_steps.npz")
    return dict(np.load(p)) if os.path.exists(p) else None


def signature_of(fix, prefix):
    return {k: fix[prefix + "." + k] for k in ("sample", "sum", "sumabs", "sumsq", "wsum", "absmax")}
