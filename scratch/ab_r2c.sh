cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fft512 or tgv512_fast or slab_poisson or poisson" 2>&1 | tail -3
for i in 1 2; do
  X3D_NO_R2C512=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rocfft-x', round(d['ms_per_step'],2), round(d['kernel_ms']['fft']['ms']/3,2))"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('r2c512', round(d['ms_per_step'],2), round(d['kernel_ms']['fft']['ms']/3,2))"
done
