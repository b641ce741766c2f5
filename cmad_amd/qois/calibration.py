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

    def _folded(self):
        """Squared weights of the 6 stored entries, the equivalent symmetric data and the constant the folding leaves
        (models/device.py `fold_weight_and_data`)."""
        from ..models.device import fold_weight_and_data
        wsq6, data6, const = fold_weight_and_data(self._weight, np.asarray(self._data))       # (6,), (6, K+1), (K+1,)
        return wsq6, data6, const

    def fused_calibration(self):
        wsq6, data6, const = self._folded()
        return wsq6, np.ascontiguousarray(data6.T[:, :, None]), float(np.sum(const[1:]))

    def stress_curvature(self):
        return self._folded()[0]

    def history_cotangents(self, sigma_hist, xi_hist):
        wsq6, data6, const = self._folded()
        mism = sigma_hist - data6.T[:, :, None]                       # (K+1, 6, B)
        sbar = wsq6[None, :, None] * mism
        sbar[0] = 0.0
        J = 0.5 * float(np.sum(sbar[1:] * mism[1:])) + float(np.sum(const[1:])) * sigma_hist.shape[2]
        return J, sbar, None

    def evaluate_hessians(self, step) -> None:
        """d2J_dxi2 (n_xi, n_xi), d2J_dxi_dparams (n_xi, P), d2J_dparams2 (P, P) of qoi.py:160-188.
        The mixed block is the true d2J/dxi dparams; the reference builds it from jacfwd(..., DXI_PREV)
        (qoi.py:51-53), which is identically zero -- the two agree whenever the stress does not depend on the
        active parameters (flow-stress calibration, every reference test)."""
        from ..models.device import fold_weight_and_data
        model = self._model
        d2C, d2S, dC, dS, info, nx = model._second_derivative_pass()
        T1, T2 = model._param_chain(info)
        wsq6, data6, _ = fold_weight_and_data(self.weight_at_step(step), self.data_at_step(step)[:, :, None])
        model.seed_none(); model.evaluate_cauchy()
        S = model.Sigma()
        s6 = np.array([S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]])
        r = wsq6 * (s6 - data6[:, 0])
        Hq = np.einsum("r,ra,rb->ab", wsq6, dS, dS) + np.einsum("r,rab->ab", r, d2S)
        gq = r @ dS
        a, c = slice(0, nx), slice(2 * nx, None)
        self.d2J_dxi2 = Hq[a, a]
        self.d2J_dxi_dparams = Hq[a, c] @ T1
        self.d2J_dparams2 = T1.T @ Hq[c, c] @ T1 + np.einsum("p,pij->ij", gq[c], T2)
