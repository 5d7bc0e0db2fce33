#!/bin/bash
# usage (GPU box): tools/ab_channel.sh <tag> [ENV=1 ...]  -- one channel bench line (1024 x 257 x 512) under the given switches
tag=$1; shift
cd $GRAFT_REPO_ROOT
env "$@" python bench.py --case channel --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-live-traffic > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
python - gpurun_out/ab_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    km = d.get("kernel_ms", {})
    print("%-28s ms_per_step %.3f" % (sys.argv[2], d["ms_per_step"]), {k: round(v, 2) for k, v in km.items() if isinstance(v, (int, float))})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
