"""What ONE RCCL point-to-point group costs on this pool (the per-group constant of X3D_COMM_EMULATE_LATENCY_US): world size 1,
groups of 8 send / recv pairs to self with 8-byte and 8 MB messages, posted on a side stream like parallel.Comm does."""
import socket
import time

import torch
import torch.distributed as dist

with socket.socket() as so:
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
for nbytes in (8, 1 << 20, 8 << 20):
    n = nbytes // 8
    s = [torch.ones(n, dtype=torch.float64, device="cuda") for _ in range(8)]
    r = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(8)]

    def group():
        ops = [dist.P2POp(dist.isend, t, 0) for t in s] + [dist.P2POp(dist.irecv, t, 0) for t in r]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for _ in range(5):
        group()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(50):
        group()
    e1.record()
    torch.cuda.synchronize()
    print("group of 8 pairs to self, %8d B each: %.1f us per group on the stream, %.1f us of host time per group"
          % (nbytes, e0.elapsed_time(e1) / 50 * 1e3, (time.perf_counter() - t0) / 50 * 1e6))
dist.destroy_process_group()
