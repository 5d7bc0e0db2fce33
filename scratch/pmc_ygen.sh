#!/bin/bash
# PMC counters of the K3g kernels (and their periodic siblings) inside the channel bench, one counter group per pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "LdsUtil VALUBusy" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf gpurun_out/pmc_ygen_$tag
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_ygen_$tag -- python3 bench.py --case channel --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_ygen_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ygen_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-44:]
        if "ygen" in k or "ytile" in k or "xwide" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/pmc_ygen_summary.txt", "w") as out:
    for k, d in sorted(acc.items()):
        line = "%-46s %s" % (k, {c: round(sum(v) / len(v), 3) for c, v in sorted(d.items())})
        print(line); out.write(line + "\n")
PY
