run() { tag=$1; shift; env "$@" timeout 300 python bench.py --entities 10000000 --steps 20 --no-cpu-baseline --proj simplex 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],3))"; }
for A in 0 16 8 4 1 2 3; do run "simplex ablate=$A" DUALIP_HIP_ABLATE=$A; done
