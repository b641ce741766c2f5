"""State-variable types and the symmetric-tensor <-> 6-vector convention
(mirror of /root/reference/cmad/models/var_types.py:11-84; numpy instead of jax.numpy)."""
from enum import IntEnum

import numpy as np


class VarType(IntEnum):
    SCALAR = 0
    VECTOR = 1
    SYM_TENSOR = 2
    TENSOR = 3


def get_num_eqs(var_type: int, ndims: int) -> int:
    if var_type == VarType.SCALAR:
        return 1
    if var_type == VarType.VECTOR:
        return ndims
    if var_type == VarType.SYM_TENSOR:
        return (ndims + 1) * ndims // 2
    if var_type == VarType.TENSOR:
        return ndims ** 2
    raise ValueError(f"Unknown var_type: {var_type}")


def get_sym_tensor_from_vector(vec, ndims: int):
    if ndims == 3:
        return np.array([[vec[0], vec[1], vec[2]], [vec[1], vec[3], vec[4]], [vec[2], vec[4], vec[5]]])
    if ndims == 2:
        return np.array([[vec[0], vec[1]], [vec[1], vec[2]]])
    if ndims == 1:
        return np.array([[vec[0]]])
    raise ValueError("Dimension != 1, 2, or 3")


def get_vector_from_sym_tensor(tensor, ndims: int):
    if ndims == 3:
        return np.array([tensor[0, 0], tensor[0, 1], tensor[0, 2], tensor[1, 1], tensor[1, 2], tensor[2, 2]])
    if ndims == 2:
        return np.array([tensor[0, 0], tensor[0, 1], tensor[1, 1]])
    if ndims == 1:
        return np.array([tensor[0, 0]])
    raise ValueError("Dimension != 1, 2, or 3")
