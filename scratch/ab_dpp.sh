cd /root/repo
./scratch/dpptest
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "yz_operators or tgv512_fast or fused_full_step or x_direction_scan or tds_solve_all or transeq_div" 2>&1 | tail -5
for i in 1 2; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dpp', d['ms_per_step'])"
done
