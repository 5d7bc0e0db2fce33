import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d2_amd import make_tgv
c = make_tgv(512, fused=True)
c.solver.n_output = 10
rows = c.run(n_iters=20)
for r in rows:
    print("t=%.3f enstrophy=%.13f div_max=%.3e div_mean=%.3e" % tuple(r[:4]))
