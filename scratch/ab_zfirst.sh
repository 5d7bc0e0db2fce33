#!/bin/bash
# z-first 000 Poisson solve (default where it applies) against the x-first one (X3D_NO_ZFIRST=1): bench lines on one
# box, then kernel-trace stats of the z-first run
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in 0 1 0 1; do
  X3D_NO_ZFIRST=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/zfirst_$v.json 2> gpurun_out/zfirst_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/zfirst_$v.json").read().strip().splitlines()[-1])
    km = d["kernel_ms"]
    print("X3D_NO_ZFIRST=$v ms_per_step", round(d["ms_per_step"], 2), {k: (v["launches"], round(v["ms"], 2)) for k, v in km.items() if isinstance(v, dict) and v["launches"]})
except Exception as e:
    print("failed", e); print(open("gpurun_out/zfirst_$v.err").read()[-1500:])
PY
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_zfirst -o zfirst -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_zfirst.log 2>&1
python - <<'PY'
import glob, csv
for f in glob.glob("gpurun_out/prof_zfirst/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:20]:
        print("%6.2f%% %9.1f us x %4s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
