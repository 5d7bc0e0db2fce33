cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deferred or tgv512_fast or fused_full_step or fused_tgv or multirank_full" 2>&1 | tail -3
for i in 1 2 3; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],2), round(d['roofline']['frac'],3))"; done
python scratch/tgv20.py 2>&1 | tail -3
