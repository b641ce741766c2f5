"""`Calibration` QoI: J = 1/2 sum (w * (sigma - data))^2 with a constant 3x3 weight mask.
Host mirror of /root/reference/cmad/qois/calibration.py:21-66."""
from __future__ import annotations

import numpy as np

from ..models.deriv_types import DerivType
from .qoi import QoI


class Calibration(QoI):
    registry_name = "calibration"

    def __init__(self, model, data, weight) -> None:
        self._model = model
        self._data = data
        assert weight.shape == (3, 3)          # weight constant and same shape as cauchy stress (:29-31)
        self._weight = weight
        self._J = None
        self._dJ = None

    @classmethod
    def from_deck(cls, qoi_section, model, data, weight) -> "Calibration":
        return cls(model, data, weight)

    def data_at_step(self, step):
        return self._data[..., step]

    def weight_at_step(self, step):
        return self._weight

    def evaluate(self, step) -> None:
        """reference qoi.py:80-110 with `_qoi` of calibration.py:56-66."""
        model = self._model
        mode = model.deriv_mode()
        d, w = self.data_at_step(step), self.weight_at_step(step)
        saved = (model._deriv_mode,)
        model.seed_none()
        model.evaluate_cauchy()
        sigma = model.Sigma()
        mismatch = w * (sigma - d)
        if mode == DerivType.DNONE:
            self._J = np.asarray(0.5 * np.sum(mismatch * mismatch), dtype=model.dtype)
            self._dJ = None
        else:
            model._deriv_mode = mode
            if mode in (DerivType.DU_PREV,):
                ds = np.zeros((3, 3, model.ndims ** 2))
            else:
                model.evaluate_cauchy()
                ds = model.dSigma()
            if mode == DerivType.DPARAMS:
                ds = ds.reshape(3, 3, -1)                      # (9, P) -> (3, 3, P)
            self._dJ = np.atleast_2d(np.einsum("ij,ijk->k", w * mismatch, ds))
        model._deriv_mode = saved[0]
