import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["X3D_LAZY_DUMP"] = "1"
from x3d2_amd import make_tgv
c = make_tgv(512, fused=False, lazy=True)
c.step(1)
print(c.solver.backend.lazy_stats())
