cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fft512 or tgv512_fast or r2c512 or poisson" 2>&1 | tail -2
for i in 1 2; do
X3D_NO_RWT=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('tile spectral', round(d['ms_per_step'],2), 'spectral', round(k['spectral']['ms']/3,2))"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('register spectral', round(d['ms_per_step'],2), 'spectral', round(k['spectral']['ms']/3,2))"
done
