#!/bin/bash
# z slabs, 010 solve: the solver's row interleave carried by the halo forms of the z pairs (default) against the
# solver's own two copy kernels (X3D_NO_YPERM=1) -- parity tests, then the emulated channel bench both ways
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_hip_channel_multirank.py tests/test_hip_poisson_010.py -x -q -m gpu > gpurun_out/yperm_tests.log 2>&1
tail -5 gpurun_out/yperm_tests.log
for v in 0 1; do
  X3D_NO_YPERM=$v X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --case channel --steps 5 --warmup 2 > gpurun_out/yperm_emul_$v.json 2> gpurun_out/yperm_emul_$v.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/yperm_emul_$v.json"))
    print("X3D_NO_YPERM=$v ms_per_step", round(d["ms_per_step"], 2), "parity", d.get("parity"), d["config"].get("overlap_self_check"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/yperm_emul_$v.err").read()[-1500:])
PY
done
