#!/bin/bash
# same-box A/B of library builds (round 4): scratch/ab4.sh "<libA.so> <libB.so> ..." [reps] [bench.py args]
# prints ms per step, the dominant kernel's launch time, the transeq / tds / fft / spectral class times of the extra step
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r04
LIBS=$1; REPS=${2:-2}; shift 2
for i in $(seq $REPS); do for L in $LIBS; do
  X3D_LIB=$PWD/$L python scratch/chan_ab.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); km = d['kernel_ms']; r = d['roofline']
dk = r.get('dominant_kernel') or {}
pd = r.get('per_direction', {})
print('%-28s step %6.2f  dom %.3f  x %.3f y %.3f z %.3f | transeq %.2f tds %.2f fft %.2f spec %.2f' % ('$L'.split('/')[-1], d['ms_per_step'], dk.get('avg_launch_ms', 0),
  *[pd.get(k, {}).get('ms_per_component', 0) for k in 'xyz'], km['transeq_fwd']['ms'] + km['transeq_bwd']['ms'], km['tds_fwd']['ms'] + km['tds_bwd']['ms'], km['fft']['ms'], km['spectral']['ms']))"
done; done
