"""phase breakdown of k_ytile_transeq3 (library built with -DYT_TIMING, scratch/exp/lib_timing.so)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x3d2_amd import _lib
_lib.LIB_PATH = os.environ.get("X3D_LIB", os.path.join(ROOT, "scratch", "exp", "lib_timing.so"))
import torch
from x3d2_amd import make_tgv
from x3d2_amd.common import DIR_X, DIR_Y, DIR_Z
case = make_tgv(512, poisson="CG", fused=True)
s = case.solver; b, al = s.backend, s.backend.allocator
s.w.fill(0.3)
o = [al.get_block(DIR_X) for _ in range(3)]
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 40)()
names = ["to_tile (LDS writes; waits for the prefetched rows)", "barrier after to_tile", "pick + window + issue prefetch", "three solves",
         "issue old-rhs load + write result to tile + barrier", "(loop overhead / next component start)", "coop read + add old + global store"]
for d, dp, nm in ((DIR_Y, s.ydirps, "y"), (DIR_Z, s.zdirps, "z")):
    for _ in range(2):
        b.transeq_planes(d, o[0], o[1], o[2], s.u, s.v, s.w, s.nu, dp, True, 0, 512)
    torch.cuda.synchronize()
    lib.x3d_debug_yt(out, 1)
    n = 5
    for _ in range(n):
        b.transeq_planes(d, o[0], o[1], o[2], s.u, s.v, s.w, s.nu, dp, True, 0, 512)
    torch.cuda.synchronize()
    lib.x3d_debug_yt(out, 0)
    tot = sum(out[:7])
    print(f"transeq3 {nm}: ticks per launch per workgroup {tot / n / 256:.0f}")
    for k in (0, 1, 2, 3, 4, 6, 5):
        print(f"   {names[k]:58s} {100.0 * out[k] / tot:5.1f} %")
    print("   per wave: component start -> end of its solves / of which in the solves (% of wave 0's total)")
    print("   " + " ".join(f"{100.0 * out[8 + w] / tot:4.0f}/{100.0 * out[24 + w] / tot:2.0f}" for w in range(16)))
