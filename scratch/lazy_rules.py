"""which rewrite of csrc/lazy.hip changes bits: deferred op-granular TGV steps against call-by-call execution, one
rule at a time (X3D_LAZY_RULES mask) -- max |difference| of u, v, w and the launch counters"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
only = sys.argv[3] if len(sys.argv) > 3 else None
from x3d2_amd import make_channel, make_tgv  # noqa: E402
chan = os.environ.get("CHANNEL")  # e.g. CHANNEL=32,33,24: the channel case instead of the TGV


def run(lazy, mask=None, keep0=False):
    if mask is not None:
        os.environ["X3D_LAZY_RULES"] = str(mask)
    os.environ["X3D_LAZY_KEEP_ZERO_TERMS"] = "1" if keep0 else "0"
    if chan:
        c = make_channel(tuple(int(v) for v in chan.split(",")), fused=False, lazy=lazy, rotation=True, omega_rot=0.12, n_rotate=2)
    else:
        c = make_tgv(n, fused=False, lazy=lazy)
    for it in range(1, steps + 1):
        c.step(it)
    s = c.solver
    f = [s.backend.get_field_data(x) for x in (s.u, s.v, s.w)]
    return f, (s.backend.lazy_stats() if lazy else None)


ref, _ = run(False)
for name, mask, keep0 in (("none", 0, True), ("none, zero terms dropped", 0, False), ("transeq_acc", 1, True), ("pair0", 2, True),
                          ("pair1", 4, True), ("tds_acc", 8, True), ("lincomb merge", 16, True),
                          ("lincomb merge + tds_lin", 48, True), ("solve000", 64, True), ("all but transeq_upd", 127, False), ("tds_acc + transeq_upd", 136, True),
                          ("all", 255, False)):
    if only is not None and name != only:
        continue
    f, st = run(True, mask, keep0)
    d = max(float(np.max(np.abs(a - b))) for a, b in zip(f, ref))
    print("%-28s max|diff| %.3e  launched %d of %d recorded, materialised %d oop %d aliases %d" %
          (name, d, st["launched"], st["recorded"], st["materialised"], st["out_of_place"], st["aliases"]), flush=True)
