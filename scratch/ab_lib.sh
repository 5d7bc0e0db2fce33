#!/bin/bash
# same-box A/B of two library builds: scratch/ab_lib.sh <libA.so> <libB.so> [bench.py args]
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
for i in 1 2 3; do for L in $A $B; do
  X3D_LIB=$PWD/$L python scratch/chan_ab.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); km = d['kernel_ms']
print('$L', round(d['ms_per_step'], 2), 'tds_fwd', round(km['tds_fwd']['ms'], 2))"
done; done
