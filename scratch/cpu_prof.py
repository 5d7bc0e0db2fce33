import sys, time, cProfile, pstats, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", sys.argv[2] if len(sys.argv) > 2 else "128")
os.environ.setdefault("OMP_PROC_BIND", "close"); os.environ.setdefault("OMP_PLACES", "cores")
from oracle import x3d_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
twopi = 6.283185307179586
mesh = orc.Mesh([n] * 3, [1, 1, 1], [twopi] * 3, ["periodic"] * 2, ["periodic"] * 2, ["periodic"] * 2)
s = orc.Solver(mesh, Re=1600.0, dt=1e-3, time_intg="RK3", poisson="FFT")
s.init_tgv(); s.step()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); s.step(); dt = time.perf_counter() - t0
pr.disable()
print(n, "threads", os.environ["OMP_NUM_THREADS"], "step %.3f s  %.3g DoF*steps/s" % (dt, n ** 3 / dt))
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
