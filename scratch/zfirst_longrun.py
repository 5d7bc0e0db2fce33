"""z-first against x-first Poisson solve over 100 TGV steps at 512^3 (fused driver): enstrophy and max |div u| traces"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, json; sys.path.insert(0, %r); from x3d2_amd import make_tgv; c = make_tgv(512, fused=True); "
        "c.solver.n_output = 20; rows = c.run(n_iters=100); print('ROWS' + json.dumps([list(map(float, r)) for r in rows]), c.solver.n_zfirst)" % root)
out = {}
for v in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, X3D_NO_ZFIRST=v), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("ROWS")][0]
    out[v] = json.loads(line[4:line.rindex("]") + 1])
    print("X3D_NO_ZFIRST=" + v, "z-first solves:", line.split()[-1])
for a, b in zip(out["0"], out["1"]):
    print("t %.3f  enstrophy %.15e | %.15e  rel diff %.1e   max|div u| %.2e | %.2e" % (a[0], a[1], b[1], abs(a[1] - b[1]) / b[1], a[2], b[2]))
