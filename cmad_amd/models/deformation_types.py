"""Deformation types (mirror of /root/reference/cmad/models/deformation_types.py:4-20)."""
from enum import IntEnum


class DefType(IntEnum):
    FULL_3D = 0
    PLANE_STRAIN = 1
    PLANE_STRESS = 2
    UNIAXIAL_STRESS = 3
    PURE_SHEAR = 4


def def_type_ndims(def_type: int) -> int:
    if def_type == DefType.FULL_3D:
        return 3
    if def_type in (DefType.PLANE_STRAIN, DefType.PLANE_STRESS):
        return 2
    if def_type in (DefType.UNIAXIAL_STRESS, DefType.PURE_SHEAR):
        return 1
    raise NotImplementedError
