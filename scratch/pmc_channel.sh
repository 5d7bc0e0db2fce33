#!/bin/bash
# HBM bytes per kernel of the channel bench (separate PMC passes, as tools_pmc.sh does for the TGV bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_chan_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_chan_$c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --case channel > gpurun_out/pmc_chan_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"gpurun_out/pmc_chan_{c}/*/*counter_collection.csv"))[-1]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == c:
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = sum(v) / len(v); res[k]["n"] = len(v)
GiB = 1024.0 ** 3
with open("gpurun_out/pmc_chan_traffic.csv", "w") as out:
    out.write("kernel,launches,fetch_bytes_corrected,write_bytes,total_GiB\n")
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) * 2 + kv[1].get("WRITE_SIZE", 0))):
        fe, wr = v.get("FETCH_SIZE", 0.0) * 1024 * 2, v.get("WRITE_SIZE", 0.0) * 1024
        out.write(f"\"{k}\",{v.get('n', 0)},{fe:.0f},{wr:.0f},{(fe + wr) / GiB:.3f}\n")
        if "k_" in k: print("%-60s n=%3d  fetch %6.2f GiB  write %6.2f GiB" % (k[-60:], v.get("n", 0), fe / GiB, wr / GiB))
PY
