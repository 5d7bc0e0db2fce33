cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast or deferred or species" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); pd=d['roofline']['per_direction']; print('nobar', round(d['ms_per_step'],2), {k: round(v['ms_per_component'],3) for k,v in pd.items()})"; done
