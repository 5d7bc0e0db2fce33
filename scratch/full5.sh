cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools_prof.sh r01q | grep -E "calls|total" | head -22
bash tools_pmc.sh r01q > gpurun_out/pmc_r01q.txt
python tools_summarize.py r01q r01q r01
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r01q.json 2> gpurun_out/bench_r01q.err; tail -c 200 gpurun_out/bench_r01q.json
python bench_ops.py > gpurun_out/bench_ops_r01q.jsonl 2>/dev/null
cp profiles/traffic.json profiles/r01_pmc_traffic.csv profiles/r01_kernel_stats.csv gpurun_out/
