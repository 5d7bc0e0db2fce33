#!/bin/bash
# channel case (BASELINE configs[4]) on one GPU: the plain single-rank path, and the code path of an N > 1 run on z
# slabs in one process (X3D_EMULATE_DECOMP=z: z "decomposed", the neighbour is the rank itself; slab 010 Poisson solver
# with pz = 1) -- bench lines + kernel-trace stats of the emulated run
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --case channel --steps 5 --warmup 2 > gpurun_out/bench_channel.json 2> gpurun_out/bench_channel.err
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --case channel --steps 5 --warmup 2 > gpurun_out/bench_channel_emul.json 2> gpurun_out/bench_channel_emul.err
export TMPDIR=/tmp
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab rocprofv3 --kernel-trace --stats -d gpurun_out/prof_chan_emul -o chan_emul -- python bench.py --case channel --steps 3 --warmup 1 > gpurun_out/prof_chan_emul.log 2>&1
python - <<'PY'
import json, glob, csv
for f in ("gpurun_out/bench_channel.json", "gpurun_out/bench_channel_emul.json"):
    try:
        d = json.load(open(f))
        print(f, "ms_per_step", round(d["ms_per_step"], 2), {k: (v["launches"], round(v["ms"], 2)) for k, v in d["kernel_ms"].items() if isinstance(v, dict)})
    except Exception as e:
        print(f, "failed", e)
for f in glob.glob("gpurun_out/prof_chan_emul/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:24]:
        print("%6.2f%% %9.1f us x %4s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
