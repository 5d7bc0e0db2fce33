#!/bin/bash
# Everything under profiles/r03_* from the CURRENT tree, in one gpurun session:
#   gpurun -- 'bash scratch/round3_artifacts.sh'     then here: python tools_summarize.py r03 r03 r03 && python scratch/collect_r03.py
rnd=r03
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( time timeout 1500 python -m pytest tests -q -m gpu ) 2>&1 | tail -6 > gpurun_out/${rnd}_tests.txt; cat gpurun_out/${rnd}_tests.txt
fi
# kernel-trace stats + PMC traffic of the bench command (TGV 512^3, fused driver)
bash tools_prof.sh ${rnd} | grep -E "calls|total" | head -24
bash tools_pmc.sh ${rnd} > gpurun_out/pmc_${rnd}.txt
# bench lines
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_${rnd}.json 2> gpurun_out/bench_${rnd}.err; tail -c 300 gpurun_out/bench_${rnd}.json
python bench.py --steps 10 --warmup 2 --lazy --no-cpu-baseline > gpurun_out/bench_${rnd}_lazy.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --op-granular --no-cpu-baseline > gpurun_out/bench_${rnd}_opg.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --n 256 --no-poisson --no-cpu-baseline > gpurun_out/bench_${rnd}_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${rnd}_emulz.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel_emulz.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab X3D_EMULATE_ALIAS=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${rnd}_emulz_alias.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab X3D_EMULATE_ALIAS=1 python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel_emulz_alias.json 2>/dev/null
python bench_ops.py > gpurun_out/bench_${rnd}_ops.jsonl 2>/dev/null
for f in lazy opg 256 channel emulz channel_emulz emulz_alias channel_emulz_alias; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_${rnd}_$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'])"; done
# the unchanged reference solver through the Fortran shim: deferred execution vs call by call
bash scratch/shim_run.sh fortran/tgv512.x3d tgv512
# kernel-trace stats of the lazy (op sequence through the queue) run, the channel bench, the emulated N > 1 paths
for tag in lazy chan emulz chan_emulz; do rm -rf gpurun_out/prof_$tag; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lazy -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --lazy > gpurun_out/prof_lazy.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chan -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --case channel > gpurun_out/prof_chan.log 2>&1
export X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emulz -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_emulz.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chan_emulz -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --case channel > gpurun_out/prof_chan_emulz.log 2>&1
unset X3D_EMULATE_DECOMP X3D_FORCE_PENCIL_FFT
bash scratch/pmc_channel.sh > gpurun_out/${rnd}_pmc_channel.txt 2>&1; tail -3 gpurun_out/${rnd}_pmc_channel.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
