cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools_prof.sh r01w | grep -E "calls|total" | head -24
bash tools_pmc.sh r01w > gpurun_out/pmc_r01w.txt
python tools_summarize.py r01w r01w r01
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r01w.json 2> gpurun_out/bench_r01w.err; tail -c 200 gpurun_out/bench_r01w.json
python bench.py --steps 5 --warmup 2 --op-granular --no-cpu-baseline > gpurun_out/bench_r01w_opg.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --n 256 --no-poisson --no-cpu-baseline > gpurun_out/bench_r01w_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_r01w_channel.json 2>/dev/null
for f in opg 256 channel; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_r01w_$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'])"; done
cp profiles/traffic.json profiles/r01_pmc_traffic.csv profiles/r01_kernel_stats.csv gpurun_out/
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
