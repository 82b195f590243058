#!/bin/bash
# FIRST CONTACT with a multi-GPU node (none was available to any build session: DESIGN.md section 5).  Run from the repository root on a box with
# >= 2 MI355X; every step is capped, the whole script at ~5 minutes of work per stage, and it stops at the first failure -- nothing is timed before
# the exchange has been shown correct across DIFFERENT devices.
#   1. the one test no 1-GPU box can run: two kernel handles on two devices of ONE process (dist_utils.split_tensors_to_devices idiom)
#   2. dl_comm_selftest at W = 2 on two devices (P2P mailboxes over hipIpc between different GPUs, peer access, both orderings; then RCCL)
#   3. the reference's 2-rank golden trace through the sharded C loop, ranks bit-identical
#   4. bench.py --gpus 2 / 4 / 8 --no-late: each line proves by itself which GPUs took part (aux.collective.ranks: UUID / PCI per rank,
#      distinct_gpus == N or bench.py refuses), which exchange ran (state, degrade_happened) and the ranks' skew (per_rank)
set -u
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
OUT=${1:-gpurun_out/first_contact}; mkdir -p "$OUT"
N=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $N" | tee "$OUT/summary.txt"
if [ "$N" -lt 2 ]; then echo "first_contact needs >= 2 GPUs" | tee -a "$OUT/summary.txt"; exit 3; fi
step() { # name, timeout seconds, command...
  local name=$1 cap=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  if timeout "$cap" "$@" > "$OUT/$name.log" 2>&1; then echo "   ok" | tee -a "$OUT/summary.txt"; else echo "   FAILED (rc $?) -- see $OUT/$name.log; stopping" | tee -a "$OUT/summary.txt"; tail -20 "$OUT/$name.log"; exit 1; fi
}
step 1_two_devices_one_process 300 python -m pytest tests/test_gpu_edge_cases.py -q -x -k test_handles_on_two_devices_in_one_process --timeout 280 -rs
step 2_exchange_two_devices 300 python -m pytest tests/test_gpu_multi_device.py -q -x -k "allreduce_across_two_devices" --timeout 280 -rs
step 3_golden_trace_two_devices 300 python -m pytest tests/test_gpu_multi_device.py -q -x -k "golden_across_two_devices" --timeout 280 -rs
for W in 2 4 8; do
  [ "$W" -le "$N" ] || continue
  step 4_bench_gpus_$W 300 python bench.py --gpus $W --steps 30 --warmup 5 --no-late
  tail -1 "$OUT/4_bench_gpus_$W.log" | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c = d['aux']['collective']
print('   N=%d  %.1f it/s  %.4f ms/step  exchange %s  degraded %s  distinct GPUs %d  kernel skew %.3f  us/exchange %s  verified %s' % (
    d['n_gpus'], d['value'], d['ms_per_step'], c['state'], c['degrade_happened'], c['distinct_gpus'], c['per_rank']['kernel_skew'], c.get('us_per_exchange'), (d['aux'].get('verified') or {}).get('ok_all_ranks')))
for r in c['ranks']: print('     rank', r['rank'], r['uuid'], r['pci'], 'ordinal', r['device_ordinal'])" | tee -a "$OUT/summary.txt"
done
echo "first contact complete" | tee -a "$OUT/summary.txt"
