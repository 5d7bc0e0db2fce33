#!/bin/bash
# PMC counters of the tile kernels on the micro-harness (one counter group per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "LdsUtil VALUBusy" "LdsBankConflict MemUnitBusy" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" "MemUnitStalled WriteUnitStalled"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_tile_$tag -- python3 scratch/tile_bench.py --only ${1:-transeq} --iters 2 > gpurun_out/pmc_tile_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_tile_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-48:]
        if "ytile" in k or "xscan" in k or "halo_fix" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 3) for c, v in sorted(d.items())})
PY
