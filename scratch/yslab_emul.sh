#!/bin/bash
# the N > 1 code path in one process at 512^3, z slabs (HALO z kernels + csrc/sfft.hip) against y slabs (HALO y kernels +
# the z-first solve of csrc/sfftz.hip); stand-in copies for the all-to-all, the same with the buffers aliased, and every
# exchange through RCCL to self
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ys_$name.json 2> gpurun_out/ys_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ys_$name.json").read().strip().splitlines()[-1])
    km = d["kernel_ms"]
    print("%-28s ms_per_step %6.2f  zfirst %s  %s" % ("$name", d["ms_per_step"], d["config"].get("poisson_z_first"), {k: (v["launches"], round(v["ms"], 2)) for k, v in km.items() if isinstance(v, dict) and v["launches"]}))
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/ys_$name.err").read()[-800:])
PY
}
run n1
run zslab X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab
run yslab X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab
run zslab_alias X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab X3D_EMULATE_ALIAS=1
run yslab_alias X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab X3D_EMULATE_ALIAS=1
run zslab_rccl X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab X3D_COMM_SELF_VIA_NCCL=1
run yslab_rccl X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab X3D_COMM_SELF_VIA_NCCL=1
run yslab_rccl_p4 X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab X3D_COMM_SELF_VIA_NCCL=1 X3D_SLAB_PARTS=4
