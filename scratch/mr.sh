cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "multirank" 2>&1 | tail -3
X3D_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), {k: (v['launches'], round(v['ms']/2,2)) for k,v in d['kernel_ms'].items() if v['ms']>0}); print(d['roofline']['per_direction'])"
