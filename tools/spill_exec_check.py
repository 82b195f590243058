#!/usr/bin/env python3
"""Screen device assembly for the code-generation defect described in dualip_amd/_build.py (_spill_defects) and DESIGN.md
section 8: a VGPR spill placed ahead of the `s_or_b64 exec, exec, ...` of a control-flow join.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -S -o k.s file.hip
    python tools/spill_exec_check.py k.s [...]          # exit status 1 if any occurrence is found

The in-tree build runs the same screen on every translation unit (the objects are assembled from the screened text)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualip_amd._build import _spill_defects  # noqa: E402

found = []
for path in sys.argv[1:]:
    found += [f"{path}: {d}" for d in _spill_defects(path)]
print("\n".join(found) if found else "clean")
print(f"{len(found)} VGPR spill(s) ahead of an exec restore")
sys.exit(1 if found else 0)
