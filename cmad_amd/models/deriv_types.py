"""Derivative seeds of the stateful `Model.evaluate()` surface; names and values as the reference's `DerivType`
(/root/reference/cmad/models/deriv_types.py:4-10), which callers pass by name and the C-ABI's `which` argument by value."""
from enum import IntEnum

DerivType = IntEnum("DerivType", ["DXI", "DXI_PREV", "DPARAMS", "DU", "DU_PREV", "DNONE"], start=0)
DerivType.__doc__ = "DXI = 0, DXI_PREV = 1, DPARAMS = 2, DU = 3, DU_PREV = 4, DNONE = 5 (CM_W_* of cmad_hip.h)"
