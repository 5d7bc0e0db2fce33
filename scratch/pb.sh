cd /root/repo
for i in 1 2; do timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast or fused_full_step" 2>&1 | tail -1; done
for i in 1 2 3; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('step', round(d['ms_per_step'],2), 'tds', round(k['tds_fwd']['ms']/3,2))"; done
