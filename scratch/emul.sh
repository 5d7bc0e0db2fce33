#!/bin/bash
# same-box comparison: N = 1 path vs the N > 1 code path emulated in one process (local kernel cost)
cd $GRAFT_REPO_ROOT
python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/emul_n1.json 2>gpurun_out/emul_n1.err
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/emul_z.json 2>gpurun_out/emul_z.err
python - <<'PY'
import json
for tag in ("n1", "z"):
    try:
        d = json.loads(open(f"gpurun_out/emul_{tag}.json").read().strip().split("\n")[-1])
    except Exception as e:
        print(tag, "failed", e); print(open(f"gpurun_out/emul_{tag}.err").read()[-1500:]); continue
    print(tag, "ms/step %.2f" % d["ms_per_step"])
    for k, v in d["kernel_ms"].items():
        if v["launches"]:
            print("   %-12s launches %4d  ms/step %7.3f" % (k, v["launches"], v["ms"] / d["steps"]))
PY
