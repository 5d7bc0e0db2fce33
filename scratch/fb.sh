cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fft512 or tgv512_fast or r2c512 or poisson or slab or multirank" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('step', round(d['ms_per_step'],2), 'fft', round(k['fft']['ms']/3,2), 'spectral', round(k['spectral']['ms']/3,2))"; done
