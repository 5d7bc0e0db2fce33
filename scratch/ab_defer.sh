cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deferred or tgv512_fast or fused_full_step or fused_tgv or fused_transeq" 2>&1 | tail -8
for i in 1 2; do
  X3D_NO_DEFER=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-defer', d['ms_per_step'])"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer', d['ms_per_step'])"
done
