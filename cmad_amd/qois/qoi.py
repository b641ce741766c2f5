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
        raise NotImplementedError("abstract: Calibration.evaluate_hessians is the per-step form; the objectives take the "
                                  "whole-history route (stress_curvature / state_curvature with cm_hessian_history)")

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

    # ---- whole-history protocol used by cmad_amd.objectives (the kernels differentiate the model, the QoI its own formula)
    def history_cotangents(self, sigma_hist, xi_hist):
        """Evaluate the QoI on a whole history: sigma_hist (K+1, 6, B) stored global stress entries [xx,xy,xz,yy,yz,zz],
        xi_hist (K+1, n_xi, B).  Returns (J, sigma_bar_hist (K+1, 6, B) = dJ/dsigma, xi_bar_hist (K+1, n_xi, B) = explicit
        dJ/dxi or None); slot 0 is the initial configuration and carries no contribution."""
        raise NotImplementedError

    def stress_curvature(self):
        """Diagonal d2J_k/dsigma_r^2 over the 6 stored entries, (6,) constant in time or (K+1, 6) per step, for the second-order pass."""
        raise NotImplementedError("this QoI has no second-order pass")

    def state_curvature(self):
        """Diagonal d2J_k/dxi_i^2 per step, (K+1, n_xi), of a QoI with an explicit dJ/dxi; None for QoIs of the stress only."""
        return None

    def fused_calibration(self):
        """(wsq6, data6_hist (K+1, 6, 1), constant) when the QoI is the weighted stress mismatch the kernels fuse
        (cm_objective_grad_history), else None."""
        return None

    def data_at_step(self, step):
        raise NotImplementedError

    def weight_at_step(self, step):
        raise NotImplementedError
