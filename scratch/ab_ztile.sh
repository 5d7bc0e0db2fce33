cd /root/repo
for i in 1 2; do
  X3D_NO_DEFER=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nodefer', round(d['ms_per_step'],2), d['roofline']['per_direction'])"
  X3D_NO_DEFER=1 X3D_ZTILE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nodefer+ztile', round(d['ms_per_step'],2), d['roofline']['per_direction'])"
done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],2), d['roofline']['per_direction'])"
X3D_ZTILE=1 X3D_NO_DEFER=1 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators" 2>&1 | tail -2
