#!/bin/bash
# the UNCHANGED reference solver.f90 / cases through the Fortran shim (fortran/_build/xcompact_hip, built where
# /root/reference is mounted, shipped as a built artefact) on the GPU box: its own "Averaged time per step" with the
# library's deferred execution (default) and call by call (X3D_NO_LAZY=1), and the monitoring traces of both.
#   scratch/shim_run.sh <input.x3d> <tag>
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
IN=$ROOT/$1
TAG=$2
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
for mode in lazy eager; do
  wd=$(mktemp -d)
  cd "$wd"
  if [ $mode = eager ]; then export X3D_NO_LAZY=1; else unset X3D_NO_LAZY; fi
  timeout 900 "$ROOT/fortran/_build/xcompact_hip" "$IN" > "$OUT/shim_${TAG}_${mode}.log" 2>&1
  echo "rc $?" >> "$OUT/shim_${TAG}_${mode}.log"
  cp monitoring.csv "$OUT/shim_${TAG}_${mode}_monitoring.csv" 2>/dev/null
  cd "$ROOT"; rm -rf "$wd"
done
grep -H "Averaged time per step\|rc " "$OUT"/shim_${TAG}_*.log
cmp "$OUT/shim_${TAG}_lazy_monitoring.csv" "$OUT/shim_${TAG}_eager_monitoring.csv" && echo "monitoring.csv: deferred == call by call, byte for byte"
