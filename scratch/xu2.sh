cd /root/repo
for i in 1 2 3; do
X3D_NO_DEFER_GRAD=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('separate grad ops', round(d['ms_per_step'],2))"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('grad in transeq_x ', round(d['ms_per_step'],2))"; done
