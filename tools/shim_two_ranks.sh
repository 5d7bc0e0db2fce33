#!/bin/bash
MPIRUN=$(which mpirun || echo /opt/conda/bin/mpirun)
for sw in NONE NONE X3D_NO_CIRC; do
  wd=$(mktemp -d); cd $wd
  env $sw=1 HSA_ENABLE_IPC_MODE_LEGACY=0 $MPIRUN -n 2 $GRAFT_REPO_ROOT/fortran/_build/xcompact_hip $GRAFT_REPO_ROOT/fortran/tgv512_y2.x3d > log.txt 2>&1
  echo "$sw: $(grep -i 'per step' log.txt | head -1) | $(tail -1 monitoring.csv 2>/dev/null)"; tail -2 log.txt | head -1
  cd $GRAFT_REPO_ROOT
done
