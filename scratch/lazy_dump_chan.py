import os, sys
sys.path.insert(0, "/root/repo")
os.environ["X3D_LAZY_DUMP"] = "1"
from x3d2_amd import make_channel
c = make_channel((32, 33, 24), fused=False, lazy=True, rotation=True, omega_rot=0.12, n_rotate=5)
c.step(1)
print(c.solver.backend.lazy_stats())
