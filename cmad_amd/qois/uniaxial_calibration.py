"""`UniaxialCalibration` QoI: compares [sigma_aa, stretch_1 - 1, stretch_2 - 1] with data, per-step weights.
Host mirror of /root/reference/cmad/qois/uniaxial_calibration.py:21-85."""
from __future__ import annotations

import numpy as np

from ..models.deriv_types import DerivType
from .qoi import QoI


class UniaxialCalibration(QoI):
    registry_name = "uniaxial_calibration"

    def __init__(self, model, data, weight, uniaxial_stress_idx: int, stretch_var_idx: int) -> None:
        self._model = model
        assert data.shape == weight.shape
        self._data, self._weight = data, weight
        self._idx, self._svar = int(uniaxial_stress_idx), int(stretch_var_idx)
        self._J = None
        self._dJ = None

    @classmethod
    def from_deck(cls, qoi_section, model, data, weight) -> "UniaxialCalibration":
        return cls(model=model, data=data, weight=weight, uniaxial_stress_idx=qoi_section["uniaxial_stress_idx"],
                   stretch_var_idx=qoi_section["stretch_var_idx"])

    def update_data(self, data) -> None:
        assert data.shape == self._data.shape
        self._data = data

    def data_at_step(self, step):
        return self._data[..., step]

    def weight_at_step(self, step):
        return self._weight[:, step]

    def history_cotangents(self, sigma_hist, xi_hist):
        """J = 1/2 sum_k || w_k o ([sigma_aa, stretch_1 - 1, stretch_2 - 1]_k - data_k) ||^2 (uniaxial_calibration.py:70-85):
        the stress term contributes dJ/dsigma_aa, the stretches are state variables and contribute dJ/dxi directly."""
        a = self._idx
        r_aa = (0, 3, 5)[a]                                            # slot of (a, a) in [xx,xy,xz,yy,yz,zz]
        off = self._model.delta_xi_offset(self._svar, 0)
        K1, _, B = sigma_hist.shape
        d, w = np.asarray(self._data), np.asarray(self._weight)       # (3, K+1)
        pred = np.stack([sigma_hist[:, r_aa, :], xi_hist[:, off, :] - 1.0, xi_hist[:, off + 1, :] - 1.0], axis=1)   # (K+1, 3, B)
        wm = (pred - d.T[:, :, None]) * w.T[:, :, None]
        wm[0] = 0.0
        J = 0.5 * float(np.sum(wm * wm))
        cot = wm * w.T[:, :, None]
        sbar = np.zeros_like(sigma_hist)
        sbar[:, r_aa, :] = cot[:, 0, :]
        xibar = np.zeros_like(xi_hist)
        xibar[:, off, :] = cot[:, 1, :]
        xibar[:, off + 1, :] = cot[:, 2, :]
        return J, sbar, xibar

    def stress_curvature(self):
        """d2J_k/dsigma_aa^2 = w_0k^2 per step (the weights change from step to step): (K+1, 6)."""
        w = np.asarray(self._weight, dtype=np.float64)
        h = np.zeros((w.shape[1], 6))
        h[:, (0, 3, 5)[self._idx]] = w[0] ** 2
        h[0] = 0.0
        return h

    def state_curvature(self):
        """d2J_k/d(stretch_i)^2 = w_ik^2 at the two lateral-stretch entries of the state: (K+1, n_xi)."""
        w = np.asarray(self._weight, dtype=np.float64)
        off = self._model.delta_xi_offset(self._svar, 0)
        h = np.zeros((w.shape[1], self._model.num_dofs))
        h[:, off] = w[1] ** 2
        h[:, off + 1] = w[2] ** 2
        h[0] = 0.0
        return h

    def evaluate_hessians(self, step) -> None:
        """d2J_dxi2 (n_xi, n_xi), d2J_dxi_dparams (n_xi, P), d2J_dparams2 (P, P) at the model's current state (reference
        qoi.py:160-188 with `_qoi` of uniaxial_calibration.py:70-85): the axial-stress term through the model's first and second
        stress derivatives (cm_hessians), the two stretch terms -- linear in the state -- as their constant diagonal."""
        model = self._model
        d2C, d2S, dC, dS, info, nx = model._second_derivative_pass()
        T1, T2 = model._param_chain(info)
        d, w = self.data_at_step(step), self.weight_at_step(step)
        r_aa = (0, 3, 5)[self._idx]
        model.seed_none(); model.evaluate_cauchy()
        r = w[0] ** 2 * (model.Sigma()[self._idx, self._idx] - d[0])
        Hq = w[0] ** 2 * np.outer(dS[r_aa], dS[r_aa]) + r * d2S[r_aa]
        gq = r * dS[r_aa]
        off = model.delta_xi_offset(self._svar, 0)
        Hq[off, off] += w[1] ** 2
        Hq[off + 1, off + 1] += w[2] ** 2
        a, c = slice(0, nx), slice(2 * nx, None)
        self.d2J_dxi2 = Hq[a, a]
        self.d2J_dxi_dparams = Hq[a, c] @ T1
        self.d2J_dparams2 = T1.T @ Hq[c, c] @ T1 + np.einsum("p,pij->ij", gq[c], T2)

    def evaluate(self, step) -> None:
        """reference qoi.py:80-110 with `_qoi` of uniaxial_calibration.py:70-85."""
        model = self._model
        mode = model.deriv_mode()
        d, w = self.data_at_step(step), self.weight_at_step(step)
        a = self._idx
        model.seed_none()
        model.evaluate_cauchy()
        stretches = np.asarray(model.xi()[self._svar], dtype=float)
        pred = np.r_[model.Sigma()[a, a], stretches[0] - 1., stretches[1] - 1.]
        mismatch = (pred - d) * w
        model._deriv_mode = mode
        if mode == DerivType.DNONE:
            self._J = np.asarray(0.5 * np.sum(mismatch * mismatch), dtype=model.dtype)
            self._dJ = None
            return
        if mode == DerivType.DU_PREV:
            self._dJ = np.zeros((1, model.ndims ** 2))
            return
        model.evaluate_cauchy()
        ds = model.dSigma()
        if mode == DerivType.DPARAMS:
            ds = ds.reshape(3, 3, -1)
        dpred = np.zeros((3, ds.shape[-1]))
        dpred[0] = ds[a, a]
        if mode == DerivType.DXI:                               # the stretches are state variables
            off = model.delta_xi_offset(self._svar, 0)
            dpred[1, off] = 1.0
            dpred[2, off + 1] = 1.0
        self._dJ = np.atleast_2d((mismatch * w) @ dpred)
