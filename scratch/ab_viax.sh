cd /root/repo
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast or fused_full_step" 2>&1 | tail -5
for i in 1 2; do
  X3D_NO_VIA_X=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-sweep', d['ms_per_step'])"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('via_x', d['ms_per_step'])"
done
