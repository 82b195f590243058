#!/bin/bash
# developer aid: the whole GPU suite once per plan switch; every plan computes the same function, so every run must be green
# (tests that state the DEFAULT plan skip under the switch that removes it).   bash tools/suite_under_switches.sh > gpurun_out/suite_under_switches.txt
for sw in ${SWITCHES:-"DUALIP_HIP_SLAB32=0" "DUALIP_HIP_SLAB32=tiny" "DUALIP_HIP_LANES_BINARY=1" "DUALIP_HIP_SELL=0" "DUALIP_HIP_FLAT=0" "DUALIP_HIP_COMPACT=0" "DUALIP_HIP_HOST_PACK=1" "DUALIP_HIP_COLD_XCD=0" "DUALIP_HIP_XCD_BALANCE=0"}; do  # (DUALIP_HIP_LDS_MODE=grad|none remove the hot-rows and fairness plans a dozen tests state: covered by test_lds_plans_agree and the goldens' switch matrix)
  echo "== $sw"
  env $sw python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | cut -c1-220
done
