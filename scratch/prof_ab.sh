#!/bin/bash
# overhead of bench.py's per-launch HIP-event timers: X3D_BENCH_PROF=1 / 0, same box
for p in 1 0 1 0; do
  P=$p X3D_BENCH_PROF=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('prof', os.environ.get('P'), d['ms_per_step'])"
done
