#!/bin/bash
# LDS / VALU utilisation counters of the z-first solve's kernels in the bench (one counter group per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "LdsUtil VALUBusy" "LdsBankConflict MemUnitBusy" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf gpurun_out/pmc_zf_$tag
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_zf_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_zf_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_zf_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-52:]
        if "tds_pair" in k or "c2c512" in k or "k_fft512" in k or "transeq3" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 2) for c, v in sorted(d.items())})
PY
