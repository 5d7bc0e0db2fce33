#!/usr/bin/env python3
"""Register / scratch use per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c f.hip -o /tmp/f.o 2> /tmp/f.rem ; tools/regs.py /tmp/f.rem [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        pass
    if flt and flt not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print(f"{name[:90]:90s} VGPR {g(' VGPRs'):>4} spill {g('VGPRs Spill'):>3} scratch {g('lane.')} SGPR {g('TotalSGPRs')} sspill {g('SGPRs Spill')} occ {g('SIMD.')}")
