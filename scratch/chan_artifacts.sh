#!/bin/bash
# channel bench artefacts of the current tree (gpurun): bench line, kernel-trace stats, PMC traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --case channel --steps 5 --warmup 2 > gpurun_out/bench_channel.json 2> gpurun_out/bench_channel.err
tail -c 600 gpurun_out/bench_channel.json
rm -rf gpurun_out/prof_chan
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chan -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --case channel > gpurun_out/prof_chan.log 2>&1
bash scratch/pmc_channel.sh
