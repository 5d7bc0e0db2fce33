"""worker for the two-rank reference-vector tests: 2 processes share cuda:0 and exchange through gloo
(host-staged); each rank drives the HIP backend through the same operator sequence oracle/ref/drivers/
dump_golden.f90 drove the reference with on two MPI ranks and saves its local results.

    torchrun --nproc-per-node 2 mp_fixture_worker.py <fixture> <op|fused> <out prefix>
    torchrun --nproc-per-node 2 mp_fixture_worker.py trace <nproc_dir> <out prefix>
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def local(g, key, mesh):
    ox, oy, oz = (int(v) for v in mesh.n_offset)
    nx, ny, nz = (int(v) for v in mesh.vert_dims)
    return np.ascontiguousarray(g[key][oz:oz + nz, oy:oy + ny, ox:ox + nx])


def battery(name, fused, rank, comm):
    from util import OPNAMES, load_golden, namelist, product_mesh
    from x3d2_amd.backend import HipBackend
    from x3d2_amd.common import DIR_X, DIR_Z, VERT, move_data_loc
    from x3d2_amd.solver import Solver, SolverConfig
    g = dict(np.load(name)) if name.endswith(".npz") else load_golden(name)
    c = namelist(g)
    mesh = product_mesh(c, rank)
    ds = [k + 1 for k, p in enumerate(c["nproc"]) if int(p) > 1]  # ([1, py, pz]: the operators of y and of z)
    b = HipBackend(mesh, comm=comm)
    s = Solver(b, mesh, SolverConfig(Re=c["Re"], dt=c["dt"], time_intg=c["time_intg"], poisson_solver_type="CG",
                                     interpl_scheme=c["interpl"], der2nd_scheme=c["der2nd"], fused=fused,
                                     n_species=1, pr_species=[1.0 / 0.37]))
    al = b.allocator
    out = {"offset": np.array(mesh.n_offset)}
    for f, k in ((s.u, "in.u"), (s.v, "in.v"), (s.w, "in.w"), (s.species[0], "in.s")):
        f.set_data_loc(VERT)
        b.set_field_data(f, local(g, k, mesh))
    for d in ds:
        dp = (s.xdirps, s.ydirps, s.zdirps)[d - 1]
        for op in OPNAMES:
            src = al.get_block(DIR_X, VERT)
            b.veccopy(src, s.u)
            if op.endswith("p2v"):
                src.set_data_loc(move_data_loc(VERT, d, 1))
            a, o = al.get_block(d), al.get_block(d)
            b.reorder(a, src, 10 + d)
            b.tds_solve(o, a, getattr(dp, op))
            out[f"tds.{'xyz'[d - 1]}.{op}"] = b.get_field_data(o)
            for f in (src, a, o):
                al.release_block(f)
    curr = [s.u, s.v, s.w, s.species[0]]
    rhs = [al.get_block(DIR_X) for _ in range(4)]
    s.transeq(rhs, curr)  # (momentum + the transported scalar, nu_species = 0.37 nu as in the dump driver)
    for f, k in zip(rhs, ("transeq.du", "transeq.dv", "transeq.dw", "species.rhs")):
        out[k] = b.get_field_data(f, VERT)
    div_u = al.get_block(DIR_Z)
    s.divergence_v2p(div_u, s.u, s.v, s.w)
    out["div.div_u"] = b.get_field_data(div_u)
    out["div.maxmean"] = np.array(b.field_max_mean(div_u))
    s.gradient_p2v(*rhs[:3], div_u)
    for f, k in zip(rhs, ("dpdx", "dpdy", "dpdz")):
        out["grad." + k] = b.get_field_data(f)
    for f in rhs[:3]:
        f.set_data_loc(VERT)
    s.curl(*rhs[:3], s.u, s.v, s.w)
    for f, k in zip(rhs, "ijk"):
        out["curl." + k] = b.get_field_data(f)
    out["curl.enstrophy"] = np.array([0.5 * sum(b.scalar_product(f, f) for f in rhs[:3]) / s.ngrid])
    for it in range(2 * s.time_integrator.nstage):
        s.transeq(rhs, curr)
        s.time_integrator.step(curr, rhs, s.dt)
    for f, k in ((s.u, "u"), (s.v, "v"), (s.w, "w")):
        out["step2." + k] = b.get_field_data(f)
    out["halo_launches"] = np.array([b.halo_launches])
    return out


def trace(nproc_dir, rank, comm):
    from x3d2_amd import make_tgv
    case = make_tgv(32, nproc_dir=nproc_dir, rank=rank, poisson="CG", comm=comm, fused=True)
    case.solver.n_output = 2
    return {"rows": np.array(case.run(n_iters=6))}


def main():
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    from x3d2_amd.parallel import Comm
    comm = Comm()
    if sys.argv[1] == "trace":
        out = trace(tuple(int(x) for x in sys.argv[2].split(",")), rank, comm)
    else:
        out = battery(sys.argv[1], sys.argv[2] == "fused", rank, comm)
    np.savez(sys.argv[3] + f".{rank}.npz", **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
