"""The y-slab solver's all-to-all for py = 2, 4, 8 ranks THROUGH RCCL on a one-GPU box (world size 1): rank 0 of a
[1, py, 1] job whose every peer is the rank itself (X3D_COMM_FAKE_PEERS=1) -- per solve 2 x 4 groups of kz planes, each
group py send / recv pairs to self in one RCCL group on the communication stream (X3D_COMM_SELF_VIA_NCCL=1), the y stage
of a group running beside the transfers of the others, exactly the call pattern of bench.py --gpus py.  The spectrum that
comes back is not a Poisson solution (every slot holds this rank's own chunk): the test is that RCCL delivers, bit for bit,
what device copies deliver, overlapped and ordered, and what the exchanges cost.

    python rccl_self_py8_worker.py <py> [out file]
"""
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
py = int(sys.argv[1])
os.environ["X3D_COMM_FAKE_PEERS"] = "1"
from x3d2_amd import Mesh  # noqa: E402
from x3d2_amd.backend import HipBackend  # noqa: E402
from x3d2_amd.common import CELL, DIR_C  # noqa: E402
from x3d2_amd.parallel import Comm  # noqa: E402
from x3d2_amd.poisson_fft import HipSlabPoissonFFTZ  # noqa: E402
from x3d2_amd.solver import Solver, SolverConfig  # noqa: E402

twopi = 6.283185307179586
per = ("periodic",) * 2
n = 512
rng = np.random.default_rng(5)
f = rng.standard_normal((n, n, n))
f -= f.mean()
lines = []


def run(comm, reps=1):
    mesh = Mesh((n, n * py, n), (1, py, 1), (twopi, twopi * py, twopi), per, per, per, nrank=0)
    s = Solver(HipBackend(mesh, comm=comm), mesh, SolverConfig(fused=True))
    pf = s.backend.poisson_fft
    assert type(pf) is HipSlabPoissonFFTZ and pf.py == py and pf.parts == 4
    q = s.backend.allocator.get_block(DIR_C, CELL)
    s.backend.set_field_data(q, f, CELL)
    pf.poisson_000(q)
    out = s.backend.get_field_data(q, CELL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pf.poisson_000(q)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    comm.timed = True
    pf.poisson_000(q)
    rep = comm.timing_report()
    comm.timed = False
    return out, ms, rep


ref, ms_copy, _ = run(Comm(), 5)
with socket.socket() as so:
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
os.environ["X3D_COMM_SELF_VIA_NCCL"] = "1"
comm = Comm()
got, ms_rccl, rep = run(comm, 5)
lines.append("py = %d: poisson_000 on y slabs, every all-to-all group %d send / recv pairs to self" % (py, py))
lines.append("  device copies %.2f ms per solve; RCCL to self, overlapped %.2f ms per solve; self-check %s; RCCL %s"
             % (ms_copy, ms_rccl, comm.self_check_result, ".".join(str(v) for v in torch.cuda.nccl.version())))
lines.append("  exchanges of one solve (events on the communication stream): %s" % rep)
os.environ["X3D_NO_OVERLAP"] = "1"
comm2 = Comm()
got2, ms_ord, rep2 = run(comm2, 5)
lines.append("  RCCL to self, ordered on the compute stream %.2f ms per solve; exchanges: %s" % (ms_ord, rep2))
ok = comm.self_via_nccl and comm.fake_peers and comm.self_check_result is True and np.array_equal(got, ref) and \
    np.array_equal(got2, ref) and np.all(np.isfinite(ref))
lines.append("  bit for bit the device-copy exchange: overlapped %s, ordered %s" % (np.array_equal(got, ref), np.array_equal(got2, ref)))
print("\n".join(lines))
if len(sys.argv) > 2:
    with open(sys.argv[2], "a") as fh:
        fh.write("\n".join(lines) + "\n")
print("RCCL-TO-SELF-PY", "OK" if ok else "FAILED")
dist.destroy_process_group()
