cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deferred or tgv512_fast or fused_full_step or fused_tgv or full_size" 2>&1 | tail -5
for i in 1 2; do
X3D_NO_DEFER=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('no-defer', round(d['ms_per_step'],2))"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); pd=d['roofline']['per_direction']; k=d['kernel_ms']; print('default', round(d['ms_per_step'],2), {a: round(v['ms_per_component'],3) for a,v in pd.items()}, {a: round(v['ms']/3,2) for a,v in k.items() if v['ms']>0})"; done
