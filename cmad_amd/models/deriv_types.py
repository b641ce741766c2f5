"""Derivative seeds (mirror of /root/reference/cmad/models/deriv_types.py:4-10)."""
from enum import IntEnum


class DerivType(IntEnum):
    DXI = 0
    DXI_PREV = 1
    DPARAMS = 2
    DU = 3
    DU_PREV = 4
    DNONE = 5
