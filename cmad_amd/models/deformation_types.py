"""Deformation types; names and values as the reference's `DefType`
(/root/reference/cmad/models/deformation_types.py:4-20) = `cm_def_type` of include/cmad_hip.h plus PURE_SHEAR."""
from enum import IntEnum

DefType = IntEnum("DefType", ["FULL_3D", "PLANE_STRAIN", "PLANE_STRESS", "UNIAXIAL_STRESS", "PURE_SHEAR"], start=0)

# spatial dimension of grad u for each type (the kernels' n_gradu is its square)
_NDIMS = {DefType.FULL_3D: 3, DefType.PLANE_STRAIN: 2, DefType.PLANE_STRESS: 2, DefType.UNIAXIAL_STRESS: 1, DefType.PURE_SHEAR: 1}


def def_type_ndims(def_type: int) -> int:
    try:
        return _NDIMS[DefType(def_type)]
    except (KeyError, ValueError):
        raise NotImplementedError(f"unknown deformation type {def_type!r}") from None
