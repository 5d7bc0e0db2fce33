#!/bin/bash
# pencil Poisson solver ([1, py, pz] path) emulated in one process at 512^3: stages one after the other
# (X3D_PENCIL_PARTS=1) against 4 groups of planes, exchanges as device copies and through RCCL to self
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for nccl in 0 1; do for parts in 1 4 8; do
  X3D_COMM_SELF_VIA_NCCL=$nccl X3D_PENCIL_PARTS=$parts X3D_EMULATE_DECOMP=yz X3D_FORCE_PENCIL_FFT=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pencil_${nccl}_${parts}.json 2> gpurun_out/pencil_${nccl}_${parts}.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/pencil_${nccl}_${parts}.json"))
    print("rccl_self=$nccl parts=$parts ms_per_step", round(d["ms_per_step"], 2), d["config"].get("overlap_self_check"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/pencil_${nccl}_${parts}.err").read()[-1500:])
PY
done; done
