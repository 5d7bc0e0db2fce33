cd /root/repo
cp x3d2_amd/libx3d2_hip.so /tmp/lib_final.so
for v in yA yB yA yB; do
  cp scratch/exp/lib_$v.so x3d2_amd/libx3d2_hip.so
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), round(d['roofline']['per_direction']['y']['ms_per_component'],3))"
done
cp scratch/exp/lib_yB.so x3d2_amd/libx3d2_hip.so
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast or deferred" 2>&1 | tail -2
cp /tmp/lib_final.so x3d2_amd/libx3d2_hip.so
