#!/bin/bash
# usage: scratch/isa.sh <file.hip> <mangled-name regex> [extra hipcc flags]  -- device ISA of one kernel (gfx950):
# global loads / stores, barriers, vmcnt waits, loop labels and the register count
f=$1; pat=$2; shift 2
cd "$(dirname "$0")/../x3d2_amd/csrc" || exit 1
out=/tmp/isa_$(basename $f .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value "$@" \
  --cuda-device-only -S $f -o $out 2>&1 | grep -v "hip-link"
L=$(grep -n "^$pat.*:" $out | head -1 | cut -d: -f1)
[ -z "$L" ] && { echo "no kernel matches $pat"; exit 1; }
E=$(awk -v l=$L 'NR>l && /^\.Lfunc_end/ {print NR; exit}' $out)
echo "== $(sed -n ${L}p $out | cut -c1-120)  (lines $L..$E of $out)"
awk -v l=$L -v e=$E 'NR>=l && NR<=e+80' $out | grep -n "global_load\|global_store\|buffer_load\|buffer_store\|s_barrier\|vmcnt\|^\.LBB.*:\|NumVgprs\|ScratchSize\|Occupancy" | ${ISA_FILTER:-cat}
