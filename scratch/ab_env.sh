#!/bin/bash
# same-box A/B of environment switches: scratch/ab_env.sh "<VAR=1|-> <VAR2=1> ..." reps [bench.py args]   ("-" = no switch)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r04
SW=$1; REPS=${2:-2}; shift 2
for i in $(seq $REPS); do for S in $SW; do
  ( [ "$S" != "-" ] && export $S; python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json; d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); km = d['kernel_ms']
print('%-22s step %6.2f | transeq %.2f tds %.2f fft %.2f spec %.2f' % ('$S', d['ms_per_step'], km['transeq_fwd']['ms'] + km['transeq_bwd']['ms'], km['tds_fwd']['ms'] + km['tds_bwd']['ms'], km['fft']['ms'], km['spectral']['ms']))" )
done; done
