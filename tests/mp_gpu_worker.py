"""worker for the multi-rank GPU parity test: N processes share cuda:0, exchange
through gloo (host-staged); each rank runs the decomposed TGV and rank 0 compares
the monitoring series with the values passed in (single-rank run)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def set_species(case):
    """transported scalars start as smooth functions of the GLOBAL coordinates"""
    from x3d2_amd.common import VERT
    s = case.solver
    m = s.mesh
    x = m.vert_coords[0][None, None, :]
    y = m.vert_coords[1][None, :, None]
    z = m.vert_coords[2][:, None, None]
    for i, f in enumerate(s.species):
        f.set_data_loc(VERT)
        s.backend.set_field_data(f, np.cos((i + 1) * x) * np.sin(y) * np.cos(2 * z) + 0.3)


def main():
    nproc_dir = tuple(int(x) for x in sys.argv[1].split(","))
    dims = tuple(int(x) for x in sys.argv[2].split(","))
    n_iters = int(sys.argv[3])
    fused = sys.argv[4] == "fused"
    poisson = sys.argv[5]
    out = sys.argv[6]
    if os.environ.get("X3D_TEST_NCCL") == "1":  # one device per rank, RCCL (tests/test_hip_parity.py, nccl test)
        lr = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
    rank = dist.get_rank()
    from x3d2_amd import make_tgv
    from x3d2_amd.parallel import Comm
    nsp = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    case = make_tgv(dims, nproc_dir=nproc_dir, rank=rank, poisson=poisson, comm=Comm(), fused=fused, n_species=nsp,
                    pr_species=[0.7] * nsp, lazy=(sys.argv[4] == "lazy"))
    set_species(case)
    if len(sys.argv) > 8 and float(sys.argv[8]) > 0.0:
        # initial velocity = Taylor-Green + hash noise of the GLOBAL indices (tests/util.py: the single-rank oracle
        # evaluates the same expression on the whole grid)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from util import noisy_tgv
        from x3d2_amd.common import VERT
        m = case.solver.mesh
        for f, a in zip((case.solver.u, case.solver.v, case.solver.w),
                        noisy_tgv(dims, m.n_offset, m.vert_dims, amp=float(sys.argv[8]))):
            f.set_data_loc(VERT)
            case.solver.backend.set_field_data(f, np.ascontiguousarray(a))
    case.solver.n_output = n_iters
    rows = case.run(n_iters=n_iters)
    s = case.solver
    local = [s.backend.get_field_data(f) for f in (s.u, s.v, s.w)]
    extra = {"s%d" % i: s.backend.get_field_data(f) for i, f in enumerate(s.species)}
    np.savez(out + f".{rank}.npz", u=local[0], v=local[1], w=local[2], offset=np.array(s.mesh.n_offset),
             rows=np.array(rows), halo_launches=np.array([s.backend.halo_launches]),
             n_zfirst=np.array([s.n_zfirst]),
             lazy=np.array([s.backend.lazy_stats().get(k, 0) for k in ("transeq_acc", "pairs", "lincombs", "tds_lincomb",
                                                                        "sync_copies")] if s.backend.lazy else [0] * 5),
             **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
