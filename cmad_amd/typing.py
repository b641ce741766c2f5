"""Result containers shared by the objectives (mirror of /root/reference/cmad/typing.py:227-237)."""
from typing import NamedTuple

import numpy as np


class GradientResult(NamedTuple):
    J: float
    grad: np.ndarray


class HessianResult(NamedTuple):
    J: float
    grad: np.ndarray
    hessian: np.ndarray
