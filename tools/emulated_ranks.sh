export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; c = (d['aux'].get('collective') or {})
    print('$1', 'ms/step %.4f  fused kernel %.4f ms  frac %.3f  exchange bracket %s us  backend %s' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], c.get('us_per_exchange'), c.get('backend')))
"; }
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "1 GPU, 100M entities          "
for w in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2971$w bench.py --gpus 1 --force-sharded --emulate-world $w --steps 60 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "rank of $w (emulated, 100M/$w)   "
done
