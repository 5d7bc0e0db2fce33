#!/bin/bash
# usage: scratch/regs.sh <file.hip> [name filter]  -- VGPRs / scratch bytes / SGPR spills per kernel (gfx950)
f=$1; pat=${2:-.}
cd "$(dirname "$0")/../x3d2_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value $EXTRA \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/regs_chk.o 2>&1 | python3 -c "
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)', line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        cur = {'name': v}; rows.append(cur)
    else: cur[k] = v
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip().split('(')[0]
    if re.search(r'''$pat''', name):
        print('%-70s vgpr %4s scratch %5s sgpr-spill %3s occ %s' % (name[-70:], r.get('VGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('SGPRs Spill'), r.get('Occupancy [waves/SIMD]')))
"
