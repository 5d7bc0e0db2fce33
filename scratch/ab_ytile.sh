cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast or deferred or multirank_full" 2>&1 | tail -4
for i in 1 2; do
  X3D_NO_YTILE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('no-ytile', round(d['ms_per_step'],2), d['roofline']['per_direction']['y'])"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ytile', round(d['ms_per_step'],2), d['roofline']['per_direction']['y'])"
done
