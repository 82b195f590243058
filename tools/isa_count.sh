#!/bin/bash
# developer aid: static instruction mix of the f32/u16/LDS instantiation of the 256-wide kernel (whole kernel and by opcode)
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -S -o k4.s /root/repo/dualip_amd/csrc/matching_kernels4.hip 2>&1 | grep -v "hip-link" | head -3
python3 - <<'PY'
import collections
s=open('/tmp/k4.s').read()
name='_ZN2dl22matching_fused_kernel4IftLb1ELb1ELb0ELb0EEEvNS_9FusedArgsIT_EE'
i=s.index('\n'+name+':'); j=s.index('.Lfunc_end', i)
L=[l.strip() for l in s[i:j].split('\n')]
c=collections.Counter(l.split()[0] for l in L if l and not l.startswith((';','.')))
print('total lines', len(L), 'VALU', sum(v for k,v in c.items() if k.startswith('v_')), 'SALU', sum(v for k,v in c.items() if k.startswith('s_')), 'pk', sum(v for k,v in c.items() if k.startswith('v_pk')))
print(sorted(((v,k) for k,v in c.items() if k.startswith('v_')), reverse=True)[:14])
PY
