export TMPDIR=/tmp
cd _ab/abl
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']
    print('$1', 'ms/step %.4f kernel %.4f' % (d['ms_per_step'], r['kernel_avg_ms']))
"; }
for rep in 1 2; do
for ab in ${ABLATES:-0 65536 32768 98304 262144 294912}; do
  DUALIP_HIP_ABLATE=$ab timeout 600 python bench.py --proj simplex --steps 30 --warmup 5 --no-verify --no-late --no-cpu-baseline 2>/dev/null | line "100M simplex ablate=$ab"
done
DUALIP_HIP_ABLATE=0 timeout 600 python bench.py --proj box --steps 30 --warmup 5 --no-verify --no-late --no-cpu-baseline 2>/dev/null | line "100M box"
done
