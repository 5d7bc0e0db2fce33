#!/bin/bash
# pencil Poisson solver in groups of planes: the one-process test and the multi-rank runs that use it
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pencil_solver_in_groups or multirank_full_step or fallback_kernel or multirank_over_rccl or emulated_multirank" > gpurun_out/pencil_tests.log 2>&1
tail -8 gpurun_out/pencil_tests.log
