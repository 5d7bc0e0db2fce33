cd /root/repo
cp x3d2_amd/libx3d2_hip.so /tmp/lib_final.so
for v in v0 v1 v2 v3 v0 v3; do
  cp scratch/exp/lib_$v.so x3d2_amd/libx3d2_hip.so
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$v', round(d['ms_per_step'],2), {a: round(b['ms']/3,2) for a,b in k.items() if b['ms']>0})"
done
cp /tmp/lib_final.so x3d2_amd/libx3d2_hip.so
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
