#!/bin/bash
# kernel-trace stats of the channel bench and of the emulated N > 1 path (gpurun; then copy the csv into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in chan emulz; do rm -rf gpurun_out/prof_$tag; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chan -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --case channel > gpurun_out/prof_chan.log 2>&1
export X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emulz -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_emulz.log 2>&1
ls gpurun_out/prof_chan/*/*kernel_stats.csv gpurun_out/prof_emulz/*/*kernel_stats.csv
