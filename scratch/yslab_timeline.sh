#!/bin/bash
# round 5: the y-slab Poisson solve's exchange schedule on ONE GPU with emulated link times (bench.py --virtual-ranks 8)
out=gpurun_out/r05/yslab_timeline.txt
mkdir -p gpurun_out/r05
: > $out
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 python scratch/rccl_group_latency.py >> $out 2>/dev/null
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --virtual-ranks 8 --steps 8 --warmup 3 > gpurun_out/r05/v8_$name.json 2> gpurun_out/r05/v8_$name.err
  python - "$name" >> $out <<PY
import json, sys
try:
    o = json.loads([l for l in open("gpurun_out/r05/v8_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    x = o["config"]["exchanges_one_step"]
    print("%-34s %7.2f ms per step | all-to-all: %3d groups %6.1f MB %6.2f ms on the comm stream | neighbours: %3d groups %6.1f MB %5.2f ms | yparts %s parts %s"
          % (sys.argv[1], o["ms_per_step"], x["alltoall"]["exchanges"], x["alltoall"]["MB_sent"], x["alltoall"]["ms"],
             x["sendrecv"]["exchanges"], x["sendrecv"]["MB_sent"], x["sendrecv"]["ms"], o["emulation"]["slab_yparts"], o["emulation"]["slab_parts"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r05/v8_n1.json 2>/dev/null
python -c "
import json
o=json.loads([l for l in open('gpurun_out/r05/v8_n1.json') if l.startswith('{')][-1])
print('%-34s %7.2f ms per step' % ('N = 1 (same box)', o['ms_per_step']))" >> $out
run kz_groups_only_yparts1 X3D_SLAB_YPARTS=1
run blocks_yparts4_tail1 X3D_SLAB_YPARTS=4 X3D_SLAB_TAIL=1
run blocks_yparts4_default X3D_SLAB_YPARTS=4
run blocks_yparts8_default X3D_SLAB_YPARTS=8
run blocks_yparts4_parts2 X3D_SLAB_YPARTS=4 X3D_SLAB_PARTS=2
run blocks_yparts4_parts8 X3D_SLAB_YPARTS=4 X3D_SLAB_PARTS=8
run blocks_rows_96_160_160_96 X3D_SLAB_ROWS=96,160,160,96
run blocks_yparts4_link_peak X3D_SLAB_YPARTS=4 X3D_COMM_EMULATE_LINKS=76.8
run blocks_yparts4_free_links X3D_SLAB_YPARTS=4 X3D_COMM_EMULATE_LINKS=100000 X3D_COMM_EMULATE_LATENCY_US=0
run ordered_no_overlap_yparts1 X3D_SLAB_YPARTS=1 X3D_NO_OVERLAP=1
run no_stream_probe_yparts4 X3D_SLAB_YPARTS=4 X3D_COMM_NO_STREAM_PROBE=1
cat $out
python - <<PY
import json
for n in ("blocks_yparts4_default", "no_stream_probe_yparts4"):
    o = json.loads([l for l in open("gpurun_out/r05/v8_%s.json" % n) if l.startswith("{")][-1])
    print(n, "comm stream probe (pair ms, serial ms):", o["config"]["comm_stream_probe_ms"])
PY
