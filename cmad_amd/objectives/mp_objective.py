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
    """Gradient + Hessian via the direct-adjoint method (reference :218-345, arXiv:2501.04584): adjoint pass
    storing phi per step, then a forward sensitivity pass contracting the second derivatives of the residual
    (`model.evaluate_hessians()`, cm_hessians) and of the QoI with dxi/dp -- the reference's 13 einsum terms."""

    def _evaluate(self):
        from ..typing import HessianResult
        qoi, model, F = self._qoi, self._model, self._global_state
        xi_at_step, num_steps = self._xi_at_step, self._num_steps
        J = self._forward_pass_with_storage()
        nap = model.parameters.num_active_params
        grad = np.zeros((1, nap))
        num_dofs = model.num_dofs
        history_vec = np.zeros((num_dofs, 1))
        phi_at_step = [np.zeros(num_dofs)] * (num_steps + 1)
        for step in range(num_steps, 0, -1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            model.gather_xi(xi_at_step[step], xi_at_step[step - 1])
            model.seed_xi(); model.evaluate()
            dC_dxi = model.Jac()
            qoi.evaluate(step)
            phi = np.linalg.solve(dC_dxi.T, -qoi.dJ().T + history_vec)
            phi_at_step[step] = phi.squeeze()
            model.seed_xi_prev(); model.evaluate()
            history_vec = -model.Jac().T @ phi
            model.seed_params(); model.evaluate()
            dC_dp = model.Jac()
            qoi.evaluate(step)
            grad += phi.T @ dC_dp + qoi.dJ()
        grad = grad.squeeze()
        untransformed_grad = grad.copy()
        model.parameters.transform_grad(grad)

        hessian = np.zeros((nap, nap))
        dxi_dp_prev = np.zeros((num_dofs, nap))
        for step in range(1, num_steps + 1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            model.gather_xi(xi_at_step[step], xi_at_step[step - 1])
            model.seed_xi(); model.evaluate(); dC_dxi = model.Jac()
            model.seed_xi_prev(); model.evaluate(); dC_dxi_prev = model.Jac()
            model.seed_params(); model.evaluate(); dC_dp = model.Jac()
            dxi_dp = np.linalg.solve(dC_dxi, -dC_dp - dC_dxi_prev @ dxi_dp_prev)
            model.evaluate_hessians()
            d2C_dxi2, d2C_dxi_dxi_prev, d2C_dxi_prev2 = model.d2C_dxi2, model.d2C_dxi_dxi_prev, model.d2C_dxi_prev2
            d2C_dp2 = model.d2C_dparams2
            d2C_dp_dxi = model.d2C_dxi_dparams.transpose((0, 2, 1))
            d2C_dp_dxi_prev = model.d2C_dxi_prev_dparams.transpose((0, 2, 1))
            qoi.evaluate_hessians(step)
            d2J_dxi2, d2J_dp2, d2J_dp_dxi = qoi.d2J_dxi2, qoi.d2J_dparams2, qoi.d2J_dxi_dparams.T
            phi = phi_at_step[step]
            hessian += d2J_dp2 \
                + np.einsum("q,qij->ij", phi, d2C_dp2) \
                + np.einsum("ik,kj->ij", d2J_dp_dxi, dxi_dp) \
                + np.einsum("q,qik,kj->ij", phi, d2C_dp_dxi, dxi_dp) \
                + np.einsum("jk,ki->ij", d2J_dp_dxi, dxi_dp) \
                + np.einsum("q,qjk,ki->ij", phi, d2C_dp_dxi, dxi_dp) \
                + np.einsum("km,ki,mj->ij", d2J_dxi2, dxi_dp, dxi_dp) \
                + np.einsum("q,qkm,ki,mj->ij", phi, d2C_dxi2, dxi_dp, dxi_dp) \
                + np.einsum("q,qik,kj->ij", phi, d2C_dp_dxi_prev, dxi_dp_prev) \
                + np.einsum("q,qkm,ki,mj->ij", phi, d2C_dxi_dxi_prev, dxi_dp, dxi_dp_prev) \
                + np.einsum("q,qmk,ki,mj->ij", phi, d2C_dxi_dxi_prev, dxi_dp_prev, dxi_dp) \
                + np.einsum("q,qkm,ki,mj->ij", phi, d2C_dxi_prev2, dxi_dp_prev, dxi_dp_prev) \
                + np.einsum("q,qjk,ki->ij", phi, d2C_dp_dxi_prev, dxi_dp_prev)
            dxi_dp_prev = dxi_dp
        model.parameters.transform_hessian(hessian, untransformed_grad)
        return HessianResult(J=J, grad=grad, hessian=hessian)
