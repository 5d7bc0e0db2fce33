cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_size_pencils or tgv512_fast or fused_full_step or deferred or fused_tgv or multirank_full" 2>&1 | tail -4
for i in 1 2; do
X3D_NO_DEFER=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('separate', round(d['ms_per_step'],2), {a: round(v['ms']/3,2) for a,v in k.items() if v['ms']>0})"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('tds+lincomb', round(d['ms_per_step'],2), {a: round(v['ms']/3,2) for a,v in k.items() if v['ms']>0})"; done
