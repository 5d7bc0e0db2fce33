cd /root/repo
for args in "--steps 1 --warmup 0" "--steps 2 --warmup 1 --time-intg AB3" "--steps 2 --warmup 1 --time-intg RK4" "--steps 3 --warmup 1 --n 256" "--steps 3 --warmup 1 --n 384" "--steps 3 --warmup 1 --n 128"; do
  timeout 300 python bench.py $args --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$args', 'ms/step', round(d['ms_per_step'],2), 'value %.3e' % d['value'], 'frac', round(d['roofline']['frac'],3))
except Exception as e: print('$args', 'FAILED', e)"
done
