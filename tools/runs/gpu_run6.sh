export TMPDIR=/tmp
for rep in 1 2 3; do
for ks in 0 1; do
timeout 600 python bench.py --steps 100 --warmup 10 --gate-ksplit $ks --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ksplit $ks', d['value'], d['ms_per_step'], [ (k['kernel'],k.get('avg_us')) for k in d['kernels']][:4])"
done; done
