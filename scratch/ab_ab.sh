cd /root/repo
cp x3d2_amd/libx3d2_hip.so /tmp/lib_final.so
for v in a b a b; do
  cp scratch/exp/lib_$v.so x3d2_amd/libx3d2_hip.so
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); pd=d['roofline']['per_direction']; print('$v', round(d['ms_per_step'],2), {k: round(v['ms_per_component'],3) for k,v in pd.items()})"
done
cp /tmp/lib_final.so x3d2_amd/libx3d2_hip.so
