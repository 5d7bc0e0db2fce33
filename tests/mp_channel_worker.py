"""worker for the multi-rank channel tests (BASELINE configs[4] on several ranks): N processes share cuda:0 and
exchange through gloo (host-staged), or own one device each over RCCL (X3D_TEST_NCCL=1); each rank runs the z-slab
decomposed channel case (perturbed like tests/test_hip_poisson_010.py::_channel_steps) and stores its fields."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def perturb(case):
    """smooth perturbation of all three components, a function of the GLOBAL coordinates"""
    s = case.solver
    m = s.mesh
    X = 2 * np.pi * m.vert_coords[0][None, None, :] / m.L[0]
    Y = np.pi * m.vert_coords[1][None, :, None] / m.L[1]
    Z = 2 * np.pi * m.vert_coords[2][:, None, None] / m.L[2]
    pert = (0.05 * np.sin(X) * np.sin(Y) ** 2 * np.cos(Z), 0.04 * np.cos(X) * np.sin(Y) ** 2 * np.sin(Z),
            0.03 * np.sin(2 * X) * np.sin(Y) ** 2 * np.cos(Z))
    for f, d in zip((s.u, s.v, s.w), pert):
        s.backend.set_field_data(f, s.backend.get_field_data(f) + d)


def main():
    nproc_dir = tuple(int(x) for x in sys.argv[1].split(","))
    dims = tuple(int(x) for x in sys.argv[2].split(","))
    nsteps = int(sys.argv[3])
    stretching, beta = sys.argv[4], float(sys.argv[5])
    fused = sys.argv[6] == "fused"
    out = sys.argv[7]
    if os.environ.get("X3D_TEST_NCCL") == "1":
        lr = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
    rank = dist.get_rank()
    from x3d2_amd import make_channel
    from x3d2_amd.parallel import Comm
    case = make_channel(dims, stretching=stretching, beta=beta, fused=fused, rotation=True, omega_rot=0.12, n_rotate=2,
                        nproc_dir=nproc_dir, rank=rank, comm=Comm())
    perturb(case)
    for it in range(1, nsteps + 1):
        case.step(it)
    s = case.solver
    row = case.postprocess(nsteps, 0.01)
    local = [s.backend.get_field_data(f) for f in (s.u, s.v, s.w)]
    np.savez(out + f".{rank}.npz", u=local[0], v=local[1], w=local[2], offset=np.array(s.mesh.n_offset),
             row=np.array(row), halo_launches=np.array([s.backend.halo_launches]),
             n_interleaved=np.array([s.n_interleaved]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
