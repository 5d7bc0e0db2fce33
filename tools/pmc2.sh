#!/bin/bash
# SQ-level counters for selected kernels: tools/pmc2.sh <tag> <kernel-substring>
tag=$1; pat=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/sq_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/sq_$tag.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM --kernel-trace --output-format csv -d gpurun_out/sq2_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/sq2_$tag.log 2>&1
python - $tag "$pat" <<'PY'
import csv, glob, sys, collections
tag, pat = sys.argv[1], sys.argv[2]
for d in ("sq_", "sq2_"):
    f = glob.glob(f"gpurun_out/{d}{tag}/*/*counter_collection.csv")
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].split("(")[0]
        if any(p in k for p in pat.split(",")):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k[-60:])
        for c, v in cs.items():
            print("   %-24s %14.0f" % (c, sum(v) / len(v)))
PY
