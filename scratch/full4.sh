cd /root/repo
bash tools_prof.sh r01p | grep -E "calls|total" | head -22
bash tools_pmc.sh r01p > gpurun_out/pmc_r01p.txt
python tools_summarize.py r01p r01p r01
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r01p.json 2> gpurun_out/bench_r01p.err; tail -c 200 gpurun_out/bench_r01p.json
python bench.py --steps 5 --warmup 2 --op-granular --no-cpu-baseline > gpurun_out/bench_r01p_opg.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --n 256 --no-poisson --no-cpu-baseline > gpurun_out/bench_r01p_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_r01p_channel.json 2>/dev/null
python bench_ops.py > gpurun_out/bench_ops_r01p.jsonl 2>/dev/null
for f in opg 256 channel; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_r01p_$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'])"; done
tail -5 gpurun_out/bench_ops_r01p.jsonl | cut -c1-300
cp profiles/traffic.json profiles/r01_pmc_traffic.csv profiles/r01_kernel_stats.csv gpurun_out/
