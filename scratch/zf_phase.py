"""round 5: what the z-transforming operator pairs (k_ytile_tds_pair<.., ZF>) spend their time on.  X3D_LIB=<variant library>
python scratch/zf_phase.py -- 30 launches of each pair form at 512^3, HIP events."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x3d2_amd import _lib
if os.environ.get("X3D_LIB"):
    _lib.LIB_PATH = os.environ["X3D_LIB"]
from x3d2_amd import make_tgv
from x3d2_amd.common import DIR_X, DIR_Z
s = make_tgv(512, fused=True).solver
b, al, z = s.backend, s.backend.allocator, s.zdirps
a1, a2, t2, t3, o1 = (al.get_block(DIR_X) for _ in range(5))
for f in (a1, a2):
    f.data.normal_()
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
r = {}
r["zf mode 0"] = timed(lambda: b.tds_pair_zfirst(0, None, None, a1, a2, z.interpl_v2p, z.stagder_v2p))
r["zf mode 1"] = timed(lambda: b.tds_pair_zfirst(1, t2, t3, None, None, z.interpl_p2v, z.stagder_p2v))
r["plain z pair mode 0"] = timed(lambda: b.tds_pair(0, o1, None, a1, a2, z.interpl_v2p, z.stagder_v2p, DIR_Z))
r["plain z pair mode 1"] = timed(lambda: b.tds_pair(1, t2, t3, a1, None, z.interpl_p2v, z.stagder_p2v, DIR_Z))
print("%-26s " % os.path.basename(os.environ.get("X3D_LIB", "libx3d2_hip.so")) + "  ".join("%s %.3f ms" % kv for kv in r.items()))
