cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools_prof.sh r01l | grep -E "calls|total" | head -20
bash tools_pmc.sh r01l | head -30
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r01l.json 2> gpurun_out/bench_r01l.err; tail -c 600 gpurun_out/bench_r01l.json
