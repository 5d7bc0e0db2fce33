"""does a torch.cuda._sleep on a side stream run BESIDE kernels of the current stream? (round 5: the link-time emulation of
parallel.Comm holds the communication stream with it)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
a = torch.ones(1 << 28, dtype=torch.float64, device=dev)  # 2 GiB
b = torch.empty_like(a)
side = torch.cuda.Stream()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
# calibrate
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(50_000_000); e1.record(); torch.cuda.synchronize()
cps = 50_000_000 / (e0.elapsed_time(e1) * 1e-3)
print("cycles per second of torch.cuda._sleep: %.3e" % cps)
two_ms = int(2e-3 * cps)
def copies(): 
    for _ in range(4): b.copy_(a)
print("4 copies of 2 GiB on the current stream: %.2f ms" % timed(copies))
def sleep_side():
    with torch.cuda.stream(side): torch.cuda._sleep(2 * two_ms)
print("4 ms sleep on the side stream: %.2f ms" % timed(sleep_side))
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): torch.cuda._sleep(2 * two_ms)
    copies()
    torch.cuda.current_stream().wait_stream(side)
print("sleep (side) posted FIRST, then the copies (current): %.2f ms  (sum = serial, max = beside each other)" % timed(both))
def both2():
    copies()
    with torch.cuda.stream(side): torch.cuda._sleep(2 * two_ms)
    torch.cuda.current_stream().wait_stream(side)
print("copies (current) posted first, then sleep (side, no dependency): %.2f ms" % timed(both2))
# with the library's kernels: a TGV 512^3 step beside a sleeping side stream
from x3d2_amd import make_tgv
case = make_tgv(512, fused=True)
case.step(1)
def step(): case.step(2, more=True)
print("one fused TGV 512^3 step: %.2f ms" % timed(step, 3))
def step_sleep():
    with torch.cuda.stream(side): torch.cuda._sleep(10 * two_ms)
    case.step(2, more=True)
    torch.cuda.current_stream().wait_stream(side)
print("the step with a 20 ms sleep on the side stream posted first: %.2f ms" % timed(step_sleep, 3))
