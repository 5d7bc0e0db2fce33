#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag>  -- separate PMC passes as the microarch guide prescribes
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-live-traffic > gpurun_out/pmc_${tag}_$c.log 2>&1
done
python - $tag <<'PY'
import csv, glob, sys, collections, json
tag = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{tag}_{c}/*/*counter_collection.csv")
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = sum(v) / len(v)
        res[k]["n"] = len(v)
out = {k: v for k, v in res.items() if ("k_" in k or "fft" in k)}
json.dump(out, open(f"gpurun_out/pmc_{tag}_summary.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    print("%-70s n=%3d FETCH %10.0f KB  WRITE %10.0f KB" % (k[-70:], v.get("n", 0), v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
PY
