#!/bin/bash
# usage: scratch/regs.sh <file.hip> <kernel-name regex> [extra flags] -- VGPRs and scratch bytes of every instance (gfx950)
f=$1; pat=$2; shift 2
out=/tmp/regs_$(basename $f .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -ffp-contract=fast "$@" \
  --cuda-device-only -S /root/repo/x3d2_amd/csrc/$f -o $out 2>&1 | grep -v "hip-link"
awk -v pat="$pat" '/\.amdhsa_kernel/ {n = ($2 ~ pat) ? $2 : ""} /\.amdhsa_next_free_vgpr|\.amdhsa_private_segment_fixed_size/ {if (n != "") printf "%s %s %s\n", substr(n, 1, 60), $1, $2} /\.end_amdhsa_kernel/ {n = ""}' $out | paste - - | awk '{print $1, "scratch", $3, "vgpr", $6}'
