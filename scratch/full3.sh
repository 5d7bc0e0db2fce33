cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools_prof.sh r01o | grep -E "calls|total" | head -20
bash tools_pmc.sh r01o | head -24
python tools_summarize.py r01o r01o r01
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r01o.json 2> gpurun_out/bench_r01o.err; tail -c 300 gpurun_out/bench_r01o.json
cp profiles/traffic.json profiles/r01_pmc_traffic.csv profiles/r01_kernel_stats.csv gpurun_out/
