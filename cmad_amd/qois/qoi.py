"""Material-point quantity-of-interest base class: host mirror of /root/reference/cmad/qois/qoi.py:17-110
(`evaluate(step)` dispatching on `model.deriv_mode()`, `J()`, `dJ()`, `model()`, `data()`, `weight()`).
The stress and its derivative blocks come from the model's HIP evaluation; the scalar chain rule on a
single 3x3 tensor is host glue."""
from __future__ import annotations

from abc import ABC

import numpy as np

from ..models.deriv_types import DerivType


class QoI(ABC):
    problem_type = "material_point"

    def evaluate(self, step) -> None:
        raise NotImplementedError

    def evaluate_hessians(self, step) -> None:
        raise NotImplementedError("QoI Hessians are a SURVEY section 8(f) 'next' row")

    def J(self):
        return self._J

    def dJ(self):
        assert self._dJ is not None, "dJ() requires a non-DNONE deriv mode (seed_xi/xi_prev/params)"
        return self._dJ

    def model(self):
        return self._model

    def data(self):
        return self._data

    def weight(self):
        return self._weight

    def data_at_step(self, step):
        raise NotImplementedError

    def weight_at_step(self, step):
        raise NotImplementedError
