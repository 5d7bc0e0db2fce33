#!/bin/bash
# Regenerates everything under profiles/ for one round from the CURRENT tree (run through gpurun, then
# `python tools/summarize.py <tag> <tag> <round>` here, where git knows the commit):
#   bash scratch/round_artifacts.sh r02
rnd=${1:-r02}
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/${rnd}_tests.txt; cat gpurun_out/${rnd}_tests.txt
bash tools/prof.sh ${rnd} | grep -E "calls|total" | head -24
bash tools/pmc.sh ${rnd} > gpurun_out/pmc_${rnd}.txt
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_${rnd}.json 2> gpurun_out/bench_${rnd}.err; tail -c 300 gpurun_out/bench_${rnd}.json
python bench.py --steps 5 --warmup 2 --op-granular --no-cpu-baseline > gpurun_out/bench_${rnd}_opg.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --n 256 --no-poisson --no-cpu-baseline > gpurun_out/bench_${rnd}_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${rnd}_emulz.json 2>/dev/null
python bench_ops.py > gpurun_out/bench_${rnd}_ops.jsonl 2>/dev/null
for f in opg 256 channel emulz; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_${rnd}_$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'])"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# kernel-trace stats of the channel bench and of the emulated N > 1 path, PMC traffic of the channel bench
bash scratch/extra_profiles.sh > gpurun_out/${rnd}_extra_profiles.txt 2>&1
bash scratch/pmc_channel.sh > gpurun_out/${rnd}_pmc_channel.txt 2>&1; tail -3 gpurun_out/${rnd}_pmc_channel.txt
