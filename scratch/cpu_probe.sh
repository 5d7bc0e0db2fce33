#!/bin/bash
# what the GPU box's host gives the CPU baselines (round 4): cores visible, cgroup quota, affinity, and the reference's
# xcompact under mpirun at 64^3 with 1 / 4 / 16 ranks of one thread
cd $GRAFT_REPO_ROOT
echo "nproc $(nproc)  physical $(python -c 'import psutil; print(psutil.cpu_count(logical=False))')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | cut -c1-80
taskset -p $$
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import bench
for nd, t in (((1,1,1),1), ((1,1,1),8), ((1,2,2),1), ((1,4,4),1), ((1,4,4),2)):
    t0 = time.time()
    r = bench.cpu_reference(64, 4, t, nd)
    print(nd, t, r and (round(r['value']), r['seconds_per_step']), round(time.time() - t0, 1), flush=True)
PY
MPIRUN=$(command -v mpirun || echo /opt/conda/bin/mpirun); $MPIRUN --version 2>&1 | head -3
$MPIRUN -n 4 bash -c 'taskset -p $$' 2>&1 | head -4
