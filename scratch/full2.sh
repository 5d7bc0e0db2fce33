cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
X3D_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-400
bash tools_prof.sh r01n | grep -E "calls|total" | head -20
bash tools_pmc.sh r01n | head -24
python tools_summarize.py r01n r01n r01
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r01n.json 2> gpurun_out/bench_r01n.err; tail -c 300 gpurun_out/bench_r01n.json
cp profiles/traffic.json profiles/r01_pmc_traffic.csv profiles/r01_kernel_stats.csv gpurun_out/
