#!/bin/bash
# how many pieces should the slab solvers' spectrum travel in?  One GPU, every exchange a RCCL send / recv to self on the
# communication stream (X3D_COMM_SELF_VIA_NCCL=1): the z stage of piece m really runs beside the transfer of piece m + 1
cd "$(dirname "$0")/.."
export X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab X3D_COMM_SELF_VIA_NCCL=1
for case in tgv channel; do
  for parts in 1 2 4 8; do
    X3D_SLAB_PARTS=$parts python bench.py --case $case --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/parts_${case}_$parts.json
    python -c "
import json
d = json.load(open('gpurun_out/parts_${case}_$parts.json')); print('$case', 'parts', $parts, round(d['ms_per_step'], 2))"
  done
done
X3D_SLAB_PARTS=4 X3D_NO_OVERLAP=1 python bench.py --case channel --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d = json.loads(sys.stdin.read()); print('channel parts 4 ordered', round(d['ms_per_step'], 2))"
