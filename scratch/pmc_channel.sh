#!/bin/bash
# utilisation counters of the channel bench's kernels (separate passes; counters only with --kernel-trace): scratch/pmc_channel.sh <tag>
cd "$(dirname "$0")/.."; tag=${1:-chan}; mkdir -p gpurun_out/r04
for grp in "LdsUtil VALUBusy" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"; do
  t=$(echo $grp | tr ' ' '_')
  rm -rf gpurun_out/pmc_${tag}_$t
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --case channel > gpurun_out/pmc_${tag}_$t.log 2>&1
done
python3 - <<PY > gpurun_out/r04/pmc_utilisation_${tag}.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        if any(s in k for s in ("ygen", "xwide", "penta", "ytile", "spectral")):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 3) for c, v in sorted(d.items())})
PY
cat gpurun_out/r04/pmc_utilisation_${tag}.txt
