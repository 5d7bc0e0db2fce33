#!/bin/bash
# utilisation counters of the bench's kernels on the final round-5 tree (separate passes; counters only with --kernel-trace)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
for case in tgv channel; do
  for grp in "LdsUtil VALUBusy" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"; do
    t=${case}_$(echo $grp | tr ' ' '_')
    rm -rf gpurun_out/pmcu_$t
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcu_$t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-live-traffic --case $case > gpurun_out/pmcu_$t.log 2>&1
  done
done
python3 - <<'PY' > gpurun_out/r05/pmc_utilisation.txt
import csv, glob, collections
for case in ("tgv", "channel"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/pmcu_%s_*/*/*counter_collection.csv" % case):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][-64:]
            if any(s in k for s in ("ygen", "xwide", "ytile", "xscan", "fft512", "c2c512", "onchip", "y010")):
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("# %s (python bench.py --steps 1 --warmup 1 --case %s; averages per launch)" % (case, case))
    for k, d in sorted(acc.items()):
        print(k, {c: round(sum(v) / len(v), 3) for c, v in sorted(d.items())})
PY
cat gpurun_out/r05/pmc_utilisation.txt
