#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag>  -> gpurun_out/prof_<tag>/ + kernel stats on stdout
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag  # a repeated tag must not pick up the previous run's stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms: %.1f" % (tot/1e6))
for r in rows[:24]:
    name = r["Name"].split("(")[0][-60:]
    print("%-62s calls %4s  avg %8.3f ms  %5.1f%%" % (name, r["Calls"], float(r["AverageNs"])/1e6, float(r["Percentage"])))
PY
tail -c 400 gpurun_out/prof_$tag.log | head -c 300
