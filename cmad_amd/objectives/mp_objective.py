"""Material-point calibration objectives with first- and second-order sensitivities, built on whole-history kernels.

Public surface as the reference's (/root/reference/cmad/objectives/mp_objective.py:23-345): `MPObjective(qoi, global_state)`
with `evaluate(flat_active_values)` (canonical values in) returning `GradientResult(J, grad)` -- `HessianResult(J, grad,
hessian)` for `MPDirectAdjointObjective` -- in canonical parameters.  The construction is different: where the
reference steps through the load history in Python and assembles every step from `model.evaluate()` blocks, each
objective here is a fixed, small number of launches over the WHOLE history (`Model.history_engine()`):

    MPAdjointObjective        Calibration QoI : 1 launch   cm_objective_grad_history (forward + adjoint, QoI fused)
                              any other QoI   : 2 launches cm_update_history -> qoi.history_cotangents -> cm_adjoint_history
    MPDirectObjective         2 launches      cm_update_history -> qoi.history_cotangents -> cm_direct_history
    MPDirectAdjointObjective  4 launches      cm_update_history, cm_adjoint_history (keeps lam_k), cm_direct_history
                                              (keeps dxi_k/dp), cm_hessian_history (sum_k D_k^T W_k D_k); with active leaves
                                              outside the 12 native parameters: + cm_param_adjoint_history, cm_direct_history_ep,
                                              and cm_hessian_history_ep in place of cm_hessian_history

The QoI differentiates only its own formula (dJ/dsigma, explicit dJ/dxi, diagonal curvature in sigma); the model
derivatives, the recursions over the steps and the reductions happen in the kernels.  Sensitivities leave the kernels
w.r.t. the 12 native kernel parameters (KP order, include/cmad_hip.h) and are chained to the active parameters of
`cmad_amd.parameters.Parameters` by the model (`active_grad_from_kp`, `active_hessian_from_kp`), then to canonical ones.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from ..typing import GradientResult, HessianResult


class MPObjective(ABC):
    """A QoI over the load history `global_state` = F (ndims, ndims, num_steps + 1) of one material point."""

    def __init__(self, qoi, global_state) -> None:
        self._qoi = qoi
        self._model = qoi.model()
        self._parameters = self._model.parameters
        F = np.asarray(global_state, dtype=np.float64)
        nd, nsnap = F.shape[0], F.shape[2]
        if qoi.data().shape[-1] != nsnap:
            raise ValueError("the QoI data and the deformation history must have the same number of steps")
        # grad u = F - I per snapshot as one SoA history of a single point: (K+1, ndims^2, 1)
        self._gradu_hist = np.ascontiguousarray((np.moveaxis(F, 2, 0) - np.eye(nd)).reshape(nsnap, nd * nd, 1))
        self._num_steps = nsnap - 1

    def evaluate(self, flat_active_values):
        self._parameters.set_active_values_from_flat(flat_active_values)
        return self._evaluate(self._model.history_engine())

    @abstractmethod
    def _evaluate(self, engine): ...

    # shared pieces -------------------------------------------------------------------------------------------------
    def _primal_and_cotangents(self, engine):
        """Forward pass with storage, then the QoI on the stored history: (xi0, xi_hist, J, sigma_bar_hist, xi_bar_hist)."""
        xi0 = self._model.init_state(1)
        xi_hist, sigma_hist = engine.primal(self._gradu_hist, xi0)
        J, sbar, xibar = self._qoi.history_cotangents(sigma_hist, xi_hist)
        return xi0, xi_hist, J, sbar, xibar

    def _canonical_gradient(self, g_kp, info, g_ext=None):
        grad = self._model.active_grad_from_kp(g_kp, info, g_ext)
        native = grad.copy()
        self._parameters.transform_grad(grad)
        return grad, native

    def _extended_gradient(self, engine, xi0, xi_hist, sbar, xibar, lam_hist=None):
        """{active position: dJ/dp} of the active leaves that are differentiated by forward-mode evaluation of the model
        (rotation matrix, Hosford exponent, Hill coefficients of the network surfaces, network weights; `cm_param_blocks`
        arithmetic contracted with the adjoint vectors of the history: one more launch, `cm_param_adjoint_history`)."""
        ext = self._model.extended_active(engine.info)
        if not ext:
            return None
        if lam_hist is None:
            _, lam_hist = engine.adjoint(self._gradu_hist, sbar, xi0, xibar, want_lam=True)
        g = engine.extended([e for _, e in ext], self._gradu_hist, xi_hist, lam_hist, sbar)
        return {pos: float(g[i]) for i, (pos, _) in enumerate(ext)}


class MPAdjointObjective(MPObjective):
    """Gradient by the adjoint recursion: one solve with A_k^T per step, backwards in time."""

    def _evaluate(self, engine) -> GradientResult:
        fused = self._qoi.fused_calibration()
        g_ext = None
        if fused is not None and not self._model.extended_active(engine.info):
            wsq6, data6_hist, const = fused
            J, g_kp = engine.calibration(self._gradu_hist, data6_hist, wsq6, self._model.init_state(1))
            J += const
        else:
            xi0, xi_hist, J, sbar, xibar = self._primal_and_cotangents(engine)
            g_kp, lam_hist = engine.adjoint(self._gradu_hist, sbar, xi0, xibar, want_lam=True)
            g_ext = self._extended_gradient(engine, xi0, xi_hist, sbar, xibar, lam_hist)
        grad, _ = self._canonical_gradient(g_kp, engine.info, g_ext)
        return GradientResult(J=float(J), grad=grad)


class MPDirectObjective(MPObjective):
    """Gradient by forward sensitivities: dxi_k/dp carried through the history, contracted with the QoI cotangents."""

    def _evaluate(self, engine) -> GradientResult:
        xi0, xi_hist, J, sbar, xibar = self._primal_and_cotangents(engine)
        g_kp, _ = engine.direct(self._gradu_hist, xi_hist, sbar, xibar)
        # leaves outside the 12 native parameters have no carried sensitivity block: their share comes from the adjoint form
        g_ext = self._extended_gradient(engine, xi0, xi_hist, sbar, xibar)
        grad, _ = self._canonical_gradient(g_kp, engine.info, g_ext)
        return GradientResult(J=float(J), grad=grad)


class MPDirectAdjointObjective(MPObjective):
    """Gradient and Hessian by the direct-adjoint method (arXiv:2501.04584): with q_k = [xi_k, xi_{k-1}, p],
    D_k = dq_k/dp from the forward sensitivities and lam_k from the adjoint pass,

        d2J/dp2 = sum_k D_k^T ( d2J_k/dq2 - sum_r lam_k[r] d2C_k[r]/dq2 ) D_k

    evaluated per step on the device (`cm_hessian_history`).  The QoI supplies its own first derivatives (dJ/dsigma, explicit
    dJ/dxi) and its diagonal curvatures in the stress and in the state (`stress_curvature`, `state_curvature`), so QoIs with
    an explicit state term (UniaxialCalibration) take the second-order pass too (reference cmad/qois/qoi.py:47-57)."""

    def _evaluate(self, engine) -> HessianResult:
        qoi, model = self._qoi, self._model
        hss, hxx = qoi.stress_curvature(), qoi.state_curvature()
        xi0, xi_hist, J, sbar, xibar = self._primal_and_cotangents(engine)
        ext = model.extended_active(engine.info)      # leaves outside the 12 native parameters (rotation matrix, Hosford exponent, ...)
        g_kp, lam_hist = engine.adjoint(self._gradu_hist, sbar, xi0, xibar, want_lam=True)
        _, dxi_dp_hist = engine.direct(self._gradu_hist, xi_hist, sbar, xibar, want_blocks=True)
        g_ext = None
        if ext:
            ep = [e for _, e in ext]
            g_ext = self._extended_gradient(engine, xi0, xi_hist, sbar, xibar, lam_hist)
            dxe_hist = engine.direct_ep(ep, self._gradu_hist, xi_hist)
            H_kp = engine.hessian_ep(ep, self._gradu_hist, xi_hist, lam_hist, dxi_dp_hist, dxe_hist, sbar, hss, hxx)
        else:
            H_kp = engine.hessian(self._gradu_hist, xi_hist, lam_hist, dxi_dp_hist, sbar, hss, hxx)
        grad, native_grad = self._canonical_gradient(g_kp, engine.info, g_ext)
        hessian = model.active_hessian_from_kp(H_kp, g_kp, engine.info, ext)
        hessian = 0.5 * (hessian + hessian.T)
        self._parameters.transform_hessian(hessian, native_grad)
        return HessianResult(J=float(J), grad=grad, hessian=hessian)
