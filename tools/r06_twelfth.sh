export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']
    print('$1', 'ms/step %.4f kernel %.4f' % (d['ms_per_step'], r['kernel_avg_ms']))
"; }
for rep in 1 2 3; do
for n in 8 5 3; do
  DUALIP_DEV_LIBRARY=1 DUALIP_HIP_BALANCE_LAUNCHES=$n python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 6-25 first=$n rep$rep"
  DUALIP_DEV_LIBRARY=1 DUALIP_HIP_BALANCE_LAUNCHES=$n python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 60 --warmup 30 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 31-90 first=$n rep$rep"
  DUALIP_DEV_LIBRARY=1 DUALIP_HIP_BALANCE_LAUNCHES=$n python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M steps 6-25 first=$n rep$rep"
done; done
