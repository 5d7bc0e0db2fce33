#!/bin/bash
# kernel-trace stats of the emulated y-slab path (the N > 1 TGV default) + its bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r03_emuly.json 2>/dev/null
rm -rf gpurun_out/prof_emuly
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emuly -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_emuly.log 2>&1
python - <<'PY'
import glob, csv, json
d = json.loads(open("gpurun_out/bench_r03_emuly.json").read().strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"])
for f in glob.glob("gpurun_out/prof_emuly/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:26]:
        print("%6.2f%% %9.1f us x %4s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
PY
