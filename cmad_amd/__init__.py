"""cmad_amd -- MI355X-native batched constitutive-model evaluator behind CMAD's operator API.

Only the hot path of sandialabs/cmad lives here: the per-Gauss-point
return-mapping Newton, its sensitivities and the calibration objective built on them, as hand-written
HIP kernels (cmad_amd/csrc) reached through a C-ABI (include/cmad_hip.h).  The Python modules mirror
the reference's `cmad.parameters`, `cmad.models`, `cmad.qois` and `cmad.objectives` interfaces.
"""
__version__ = "0.1.0"
