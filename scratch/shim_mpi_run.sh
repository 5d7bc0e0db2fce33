#!/bin/bash
# the unchanged reference solver through the Fortran shim on SEVERAL ranks sharing the box's one GPU (round 4):
# device-to-device exchanges + deferred execution (default) against rounds 2-3's path (host-staged MPI exchanges, call by
# call) and the two mixed forms; its own "Averaged time per step" (two ranks on one GPU: each rank's kernels run beside
# the other's, the figure is the pair's step).   scratch/shim_mpi_run.sh <input.x3d> <ranks> <tag>
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
IN=$ROOT/$1; NR=$2; TAG=$3
OUT=$ROOT/gpurun_out/r04
mkdir -p "$OUT"
MPIRUN=$(command -v mpirun || echo /opt/conda/bin/mpirun)
for mode in d2d_lazy d2d_eager host_lazy host_eager; do
  wd=$(mktemp -d); cd "$wd"
  unset X3D_NO_LAZY X3D_SHIM_HOST_STAGED
  case $mode in *eager) export X3D_NO_LAZY=1;; esac
  case $mode in host*) export X3D_SHIM_HOST_STAGED=1;; esac
  X3D_LAZY_REPORT=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1200 "$MPIRUN" -n "$NR" "$ROOT/fortran/_build/xcompact_hip" "$IN" > "$OUT/shim_${TAG}_${mode}.log" 2>&1
  echo "rc $?" >> "$OUT/shim_${TAG}_${mode}.log"
  cp monitoring.csv "$OUT/shim_${TAG}_${mode}_monitoring.csv" 2>/dev/null
  cd "$ROOT"; rm -rf "$wd"
done
grep -H "Averaged time per step\|rc \|x3d_lazy_report" "$OUT"/shim_${TAG}_*.log | cut -c1-330
for m in d2d_eager host_lazy host_eager; do cmp "$OUT/shim_${TAG}_d2d_lazy_monitoring.csv" "$OUT/shim_${TAG}_${m}_monitoring.csv" && echo "monitoring.csv: d2d_lazy == $m"; done
cat "$OUT/shim_${TAG}_d2d_lazy_monitoring.csv"
