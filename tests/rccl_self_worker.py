"""RCCL on a one-GPU box: world size 1, every exchange of the emulated N > 1 path (X3D_EMULATE_DECOMP=z: halo rows,
boundary values, the slab solver's all-to-all parts) is a RCCL send / recv of this rank to ITSELF, started on the
communication stream like a real neighbour exchange (X3D_COMM_SELF_VIA_NCCL=1).  Fields after two TGV steps against the
same emulation with device copies: must be bit-identical; the overlapped path's self-check must pass."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512  # 512: the full 512^3 bench size, else (64, 512, n)
# second argument "y": the y-slab path (HALO y kernels, the z-first solver csrc/sfftz.hip with 4 groups of planes; 512^3)
yslabs = len(sys.argv) > 2 and sys.argv[2] == "y"
os.environ["X3D_EMULATE_DECOMP"] = "y" if yslabs else "z"
os.environ["X3D_FORCE_PENCIL_FFT"] = "yslab" if yslabs else "slab"
if yslabs:
    os.environ["X3D_SLAB_PARTS"] = "4"
    os.environ.setdefault("X3D_SLAB_YPARTS", "4")  # round 5: blocks of rows x kz planes, as on several ranks
from x3d2_amd import make_tgv  # noqa: E402
from x3d2_amd.parallel import Comm  # noqa: E402


def run(comm):
    c = make_tgv(512 if n == 512 else (64, 512, 512 if n == 511 else n), fused=True, comm=comm)
    c.step(1)
    c.step(2)
    s = c.solver
    return [s.backend.get_field_data(f) for f in (s.u, s.v, s.w)], s.backend.halo_launches


ref, _ = run(Comm())
with socket.socket() as so:
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
os.environ["X3D_COMM_SELF_VIA_NCCL"] = "1"
comm = Comm()
got, halo = run(comm)
print("self via RCCL:", comm.self_via_nccl, "overlap:", comm.overlap, "self-check:", comm.self_check_result, "halo launches:", halo)
for a, b, nm in zip(got, ref, "uvw"):
    print(nm, "max |difference| to the device-copy emulation:", float(np.max(np.abs(a - b))))
os.environ["X3D_NO_OVERLAP"] = "1"
comm2 = Comm()
got2, _ = run(comm2)
for a, b, nm in zip(got2, ref, "uvw"):
    print(nm, "ordered path (X3D_NO_OVERLAP=1):", float(np.max(np.abs(a - b))))
ok = comm.self_via_nccl and comm.self_check_result is True and all(np.array_equal(a, b) for a, b in zip(got, ref)) and \
    all(np.array_equal(a, b) for a, b in zip(got2, ref))
print("RCCL-TO-SELF", "OK" if ok else "FAILED", "halo_launches=%d" % halo)
dist.destroy_process_group()
