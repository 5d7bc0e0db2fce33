#!/bin/bash
# usage (GPU box): tools/ab_virtual.sh <tag> [ENV=1 ...]  -- bench.py --virtual-ranks 8 (the y-slab timeline of an 8-rank job on one GPU)
tag=$1; shift
cd $GRAFT_REPO_ROOT
env "$@" python bench.py --virtual-ranks 8 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-live-traffic > gpurun_out/abv_$tag.json 2> gpurun_out/abv_$tag.err
python - gpurun_out/abv_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pd = d["roofline"].get("per_direction", {})
    print("%-24s ms_per_step %.3f" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms_per_component"], 3) for k, v in pd.items()})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
