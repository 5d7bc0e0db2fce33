#!/bin/bash
# Everything under profiles/r05_* that comes from the CURRENT tree, in one gpurun session:
#   gpurun -- 'bash scratch/round5_artifacts.sh'   then here: python tools_summarize.py r05 r05 r05 && python scratch/collect_r05.py
rnd=r05
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# kernel-trace stats + PMC traffic of the bench command (TGV 512^3, fused driver)
bash tools_prof.sh ${rnd} --no-other-configs --no-live-traffic | grep -E "calls|total" | head -24
bash tools_pmc.sh ${rnd} > gpurun_out/pmc_${rnd}.txt
# the default bench line: headline + roofline with live traffic + cpu_baseline + other_configs (256^3, channel, the shim)
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${rnd}.json 2> gpurun_out/bench_${rnd}.err; tail -c 300 gpurun_out/bench_${rnd}.json
python bench.py --steps 10 --warmup 2 --lazy --no-cpu-baseline --no-other-configs --no-live-traffic > gpurun_out/bench_${rnd}_lazy.json 2>/dev/null
X3D_SINGLE_PREC=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/bench_${rnd}_fp32.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel.json 2>/dev/null
python bench_ops.py > gpurun_out/bench_${rnd}_ops.jsonl 2>/dev/null
for f in lazy fp32 channel; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_${rnd}_$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))"; done
# dry run of the N > 1 bench line on one GPU (gloo, host staged): both layouts in one run
X3D_BENCH_SHARE_GPU=1 python bench.py --gpus 4 --n 128 --steps 2 --warmup 1 > gpurun_out/bench_${rnd}_share4_dryrun.json 2>/dev/null
# the emulated scaling curve (one process = rank 0 of V, links at 61.4 GB/s): y slabs, V = 2, 4, 8
for v in 2 4 8; do python bench.py --virtual-ranks $v --steps 8 --warmup 3 > gpurun_out/bench_${rnd}_virtual$v.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_${rnd}_virtual$v.json').read().strip().split('\n')[-1]); print('virtual ranks $v', d['ms_per_step'], d['config']['exchanges_one_step'])"; done
# kernel-trace stats of the emulated 8-rank step (which kernels the N > 1 code path adds)
rm -rf gpurun_out/prof_v8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v8 -- python bench.py --virtual-ranks 8 --steps 2 --warmup 1 > gpurun_out/prof_v8.log 2>&1
# kernel-trace stats of the channel bench
rm -rf gpurun_out/prof_chan
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chan -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --case channel --no-live-traffic > gpurun_out/prof_chan.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
