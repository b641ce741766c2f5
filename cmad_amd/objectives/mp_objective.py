"""Sensitivity-providing objectives for material-point calibration -- host mirror of
/root/reference/cmad/objectives/mp_objective.py:23-215 (`MPObjective`, `MPAdjointObjective`,
`MPDirectObjective`; same constructor, `evaluate(flat_active_values) -> GradientResult`, same loops).
Every `model.evaluate*()` / `newton_solve` inside is a launch of the HIP library with B = 1; this is the
reference's one-point-per-call shape kept for drop-in use.  For many points use
`cmad_amd.objectives.batched.BatchedCalibrationObjective`, which runs the same mathematics as batched
kernels (one launch per load step for the whole batch)."""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from ..models.global_fields import mp_U_from_F
from ..models.nonlinear_solver import newton_solve
from ..typing import GradientResult


class MPObjective(ABC):
    def __init__(self, qoi, global_state) -> None:
        self._qoi = qoi
        self._model = qoi.model()
        self._parameters = qoi.model().parameters
        self._global_state = global_state
        self._num_steps = qoi.data().shape[-1] - 1
        self._xi_at_step = [[None] * self._model.num_residuals for _ in range(self._num_steps + 1)]
        self._model.store_xi(self._xi_at_step, self._model.xi(), 0)

    def evaluate(self, flat_active_values):
        self._parameters.set_active_values_from_flat(flat_active_values)
        return self._evaluate()

    @abstractmethod
    def _evaluate(self): ...

    def _forward_pass_with_storage(self) -> float:            # reference :62-89
        qoi, model, F = self._qoi, self._model, self._global_state
        model.set_xi_to_init_vals()
        # the reference stores step 0 once, at construction (:51); re-storing it here keeps the adjoint
        # right when the model was left in another state between construction and evaluation
        model.store_xi(self._xi_at_step, model.xi(), 0)
        J = 0.
        for step in range(1, self._num_steps + 1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            newton_solve(model)
            model.store_xi(self._xi_at_step, model.xi(), step)
            model.seed_none()
            qoi.evaluate(step)
            J += qoi.J()
            model.advance_xi()
        return float(J)


class MPAdjointObjective(MPObjective):
    """Gradient via reverse-time adjoint pass after a forward pass (reference :92-147)."""

    def _evaluate(self) -> GradientResult:
        qoi, model, F = self._qoi, self._model, self._global_state
        xi_at_step, num_steps = self._xi_at_step, self._num_steps
        J = self._forward_pass_with_storage()
        grad = np.zeros((1, model.parameters.num_active_params))
        history_vec = np.zeros((model.num_dofs, 1))
        for step in range(num_steps, 0, -1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            model.gather_xi(xi_at_step[step], xi_at_step[step - 1])
            model.seed_xi()
            model.evaluate()
            dC_dxi = model.Jac()
            qoi.evaluate(step)
            dJ_dxi = qoi.dJ()
            phi = np.linalg.solve(dC_dxi.T, -dJ_dxi.T + history_vec)
            model.seed_xi_prev()
            model.evaluate()
            history_vec = -model.Jac().T @ phi
            model.seed_params()
            model.evaluate()
            dC_dp = model.Jac()
            qoi.evaluate(step)
            grad += phi.T @ dC_dp + qoi.dJ()
        grad = grad.squeeze()
        model.parameters.transform_grad(grad)
        return GradientResult(J=J, grad=grad)


class MPDirectObjective(MPObjective):
    """Gradient via forward sensitivity (tangent) pass (reference :150-215)."""

    def _evaluate(self) -> GradientResult:
        qoi, model, F = self._qoi, self._model, self._global_state
        model.set_xi_to_init_vals()
        nap = model.parameters.num_active_params
        J = 0.
        grad = np.zeros((1, nap))
        dxi_dp = np.zeros((model.num_dofs, nap))
        for step in range(1, self._num_steps + 1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            newton_solve(model)
            model.seed_none()
            qoi.evaluate(step)
            J += qoi.J()
            model.seed_xi()
            model.evaluate()
            dC_dxi = model.Jac()
            qoi.evaluate(step)
            dJ_dxi = qoi.dJ()
            model.seed_xi_prev()
            model.evaluate()
            dC_dxi_prev = model.Jac()
            model.seed_params()
            model.evaluate()
            dC_dp = model.Jac()
            qoi.evaluate(step)
            dJ_dp = qoi.dJ()
            dxi_dp = np.linalg.solve(dC_dxi, -dC_dp - dC_dxi_prev @ dxi_dp)
            grad += dJ_dxi @ dxi_dp + dJ_dp
            model.advance_xi()
        grad = grad.squeeze()
        model.parameters.transform_grad(grad)
        return GradientResult(J=float(J), grad=grad)


class MPDirectAdjointObjective(MPObjective):
    """Gradient + Hessian (reference :218-345): needs second derivatives of the residual -- a SURVEY
    section 8(f) 'next' row."""

    def _evaluate(self):
        raise NotImplementedError("Hessian objective not built yet (SURVEY section 8(f))")
