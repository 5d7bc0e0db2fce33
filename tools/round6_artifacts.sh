#!/bin/bash
# Everything under profiles/r06_* that comes from the CURRENT tree, in one gpurun session:
#   gpurun -- 'bash tools/round6_artifacts.sh'   then here: python tools/summarize.py r06 r06 r06 && python tools/collect_r06.py
rnd=r06
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# kernel-trace stats + PMC traffic of the bench command (TGV 512^3, fused driver)
bash tools/prof.sh ${rnd} --no-other-configs --no-live-traffic | grep -E "calls|total" | head -24
bash tools/pmc.sh ${rnd} > gpurun_out/pmc_${rnd}.txt
# the default bench line: headline + roofline with live traffic + cpu_baseline + other_configs (256^3, channel, AB3, the shim)
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${rnd}.json 2> gpurun_out/bench_${rnd}.err; tail -c 300 gpurun_out/bench_${rnd}.json
python bench.py --steps 10 --warmup 2 --lazy --no-cpu-baseline --no-other-configs --no-live-traffic > gpurun_out/bench_${rnd}_lazy.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel.json 2>/dev/null
for f in lazy channel; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_${rnd}_$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))"; done
# the channel's same-box A/B, switch by switch (round 6's changes and the sweeps)
for sw in NONE X3D_NO_CHANNEL_DEFER_GRAD X3D_NO_ZFIRST010 X3D_NO_MEAN_IN_LINCOMB X3D_Y010_NO_DB X3D_NO_DIRECT X3D_NO_XCIRC X3D_NO_CIRC X3D_YHALF; do bash tools/ab_channel.sh ${rnd}_$sw $sw=1; done | tee gpurun_out/${rnd}_channel_ab.txt
X3D_NO_CHANNEL_DEFER_GRAD=1 X3D_NO_ZFIRST010=1 X3D_NO_MEAN_IN_LINCOMB=1 X3D_Y010_NO_DB=1 X3D_NO_DIRECT=1 X3D_NO_CIRC=1 bash tools/ab_channel.sh ${rnd}_round5_form X3D_NONE=1 | tee -a gpurun_out/${rnd}_channel_ab.txt
# TGV 512^3, configs[1] (256^3) and the FP32 flavour: the circulant form on / off, same box
{
for sw in NONE X3D_NO_XCIRC X3D_NO_CIRC; do bash tools/ab_bench.sh ${rnd}_tgv_$sw $sw=1 -- --steps 10 --warmup 3; done
for sw in NONE X3D_NO_CIRC X3D_NPW2_256; do bash tools/ab_bench.sh ${rnd}_c256_$sw $sw=1 -- --n 256 --no-poisson --steps 10 --warmup 3; done
for sw in NONE X3D_NO_CIRC X3D_NO_NPW2; do bash tools/ab_bench.sh ${rnd}_sp_$sw X3D_SINGLE_PREC=1 $sw=1 -- --steps 10 --warmup 3; done
} | tee gpurun_out/${rnd}_tgv_ab.txt
# dry run of the N > 1 bench line on one GPU (gloo, host staged), the driver's own command line at 512^3 per rank
X3D_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_${rnd}_share2_dryrun.json 2> gpurun_out/bench_${rnd}_share2_dryrun.err
tail -c 200 gpurun_out/bench_${rnd}_share2_dryrun.json
# kernel-trace stats of the channel bench
rm -rf gpurun_out/prof_chan
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chan -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --case channel --no-live-traffic --no-other-configs > gpurun_out/prof_chan.log 2>&1
# roctx: marker + kernel trace of two channel steps and two TGV steps through the deferred layer (ranges per C-ABI entry point / per rewritten operation)
rm -rf gpurun_out/mark_chan gpurun_out/mark_lazy
X3D_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d gpurun_out/mark_chan -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --case channel --no-live-traffic --no-other-configs > gpurun_out/mark_chan.log 2>&1
X3D_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d gpurun_out/mark_lazy -- python bench.py --steps 1 --warmup 1 --lazy --no-cpu-baseline --no-live-traffic --no-other-configs > gpurun_out/mark_lazy.log 2>&1
ls gpurun_out/mark_chan/*/ gpurun_out/mark_lazy/*/ 2>/dev/null | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
