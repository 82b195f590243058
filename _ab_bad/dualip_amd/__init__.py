"""dualip_amd -- MI355X-native dual-decomposition LP solver for the matching objective.

Keeps the operator API of linkedin/DuaLip (ObjectiveFunction / ProjectionMap / Maximizer and run_solver()),
with the accelerated-gradient inner loop running as hand-written HIP kernels for gfx950 behind the C ABI of
``include/dualip_hip.h``.  Module paths mirror the reference package (``dualip.objectives.matching`` ->
``dualip_amd.objectives.matching`` ...), so a caller switches by changing the import root.
"""
__version__ = "0.1.0"
