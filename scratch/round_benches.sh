#!/bin/bash
# the bench lines of scratch/round_artifacts.sh only (no tests, no rocprof passes): bash scratch/round_benches.sh r02
rnd=${1:-r02}
cd $GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_${rnd}.json 2> gpurun_out/bench_${rnd}.err; tail -c 200 gpurun_out/bench_${rnd}.json
python bench.py --steps 5 --warmup 2 --op-granular --no-cpu-baseline > gpurun_out/bench_${rnd}_opg.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --n 256 --no-poisson --no-cpu-baseline > gpurun_out/bench_${rnd}_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${rnd}_emulz.json 2>/dev/null
for f in "" _opg _256 _channel _emulz; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_${rnd}$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
