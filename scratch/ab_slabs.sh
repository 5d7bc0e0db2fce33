cd /root/repo
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast" 2>&1 | tail -2
for s in 1 4 8 16 32 1; do
  X3D_VIA_SLABS=$s timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slabs $s', d['ms_per_step'])"
done
X3D_VIA_SLABS=7 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast" 2>&1 | tail -2
