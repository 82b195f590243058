#!/bin/bash
# developer aid: compile the 256-wide kernel to ISA and list the loads / waits / spills of the f32,u16,LDS instantiation
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -S -o k4.s /root/repo/dualip_amd/csrc/matching_kernels4.hip 2>&1 | grep -v "hip-link" | head
python3 - <<'PY'
import re
s=open('/tmp/k4.s').read()
name='_ZN2dl22matching_fused_kernel4IftLb1ELb1ELb0ELb0EEEvNS_9FusedArgsIT_EE'
i=s.index('\n'+name+':'); j=s.index('.Lfunc_end', i)
body=s[i:j]; L=body.split('\n')
print('lines', len(L), 'writelane', body.count('v_writelane'), 'readlane', body.count('v_readlane'), 'scratch', body.count('scratch_'))
k=s[s.index('.amdhsa_kernel '+name):]
print(re.findall(r'amdhsa_next_free_[vs]gpr \d+', k[:4000]))
for n,l in enumerate(L):
    t=l.strip()
    if t.startswith('s_waitcnt vmcnt') or t.startswith('global_load_dwordx') or (t.startswith('global_load') and 's[' in t) or 'Loop Header: Depth=1' in t or t.startswith('s_branch .LBB0_') :
        print(n, t[:80])
PY
