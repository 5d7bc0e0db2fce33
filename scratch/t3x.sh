cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_size_pencils or tgv512_fast or fused_full_step or transeq_div or fused_transeq" 2>&1 | tail -3
for i in 1 2; do
X3D_NO_TILE3=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); pd=d['roofline']['per_direction']; print('per-comp', round(d['ms_per_step'],2), {k: round(v['ms_per_component'],3) for k,v in pd.items()}, round(d['roofline']['frac'],3))"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); pd=d['roofline']['per_direction']; print('3-comp xyz', round(d['ms_per_step'],2), {k: round(v['ms_per_component'],3) for k,v in pd.items()}, round(d['roofline']['frac'],3))"; done
