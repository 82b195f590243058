export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']
    print('$1', 'ms/step %.4f  fused kernel %.4f ms' % (d['ms_per_step'], r['kernel_avg_ms']))
"; }
for rep in 1 2 3; do
  for v in 1 0; do
    DUALIP_HIP_FUSE_APPLY=$v python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "12.5M rank/8  FUSE_APPLY=$v rep$rep"
    DUALIP_HIP_FUSE_APPLY=$v python bench.py --entities 10000000 --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "10M mixed     FUSE_APPLY=$v rep$rep"
  done
done
