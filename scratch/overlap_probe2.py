"""round 5: under bench.py --virtual-ranks no hold on the communication stream is hidden.  Which streams run beside the
library's kernels (launched on the stream HipBackend was created on) once RCCL is up?"""
import os, socket, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
if mode == "nccl":
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(8, device="cuda"); dist.all_reduce(t)
from x3d2_amd import make_tgv
case = make_tgv(512, fused=True)
case.step(1); torch.cuda.synchronize()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(50_000_000); e1.record(); torch.cuda.synchronize()
cps = 50_000_000 / (e0.elapsed_time(e1) * 1e-3)
print(mode, "GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), "| one fused step: %.2f ms" % timed(lambda: case.step(2, more=True)))
streams = [torch.cuda.Stream() for _ in range(6)] + [torch.cuda.Stream(priority=-1)]
for i, s in enumerate(streams):
    def f():
        # the bench's pattern: side waits for the compute stream, holds 20 ms, compute continues, then waits for side
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            torch.cuda._sleep(int(20e-3 * cps))
            ev = torch.cuda.Event(); ev.record()
        case.step(2, more=True)
        torch.cuda.current_stream().wait_event(ev)
    print("  stream %d (%s): step + 20 ms hold posted mid-stream: %.2f ms" % (i, "high priority" if i == 6 else "pool", timed(f)))
if mode == "nccl":
    dist.destroy_process_group()
