#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <tag> [ENV=1 ...] -- <bench.py args>   one bench line under the given switches
tag=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
cd $GRAFT_REPO_ROOT
env "${envs[@]}" python bench.py --no-cpu-baseline --no-other-configs --no-live-traffic "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
python - gpurun_out/ab_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("%-28s ms_per_step %.3f  frac %.3f" % (sys.argv[2], d["ms_per_step"], r.get("frac", 0.0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
