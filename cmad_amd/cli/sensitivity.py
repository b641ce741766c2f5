"""`sensitivity.type` -> an object with `evaluate_grad(x)` / `evaluate_hess(x)`.
Counterpart of /root/reference/cmad/cli/sensitivity.py:120-180 (same per-subcommand restrictions)."""
from __future__ import annotations

import sys
from typing import Any

import numpy as np

from ..models.nonlinear_solver import make_newton_solve
from ..objectives import MPAdjointObjective, MPDirectAdjointObjective, MPDirectObjective, MPJVPObjective
from ..typing import GradientResult, HessianResult


class _StoredStateDriver:
    """adjoint / direct / direct_adjoint: the imperative objectives of objectives/mp_objective.py."""

    def __init__(self, objective) -> None:
        self._objective = objective

    def evaluate_grad(self, x) -> GradientResult:
        r = self._objective.evaluate(x)
        return GradientResult(J=r.J, grad=r.grad)

    def evaluate_hess(self, x) -> HessianResult:
        r = self._objective.evaluate(x)
        if not isinstance(r, HessianResult):
            raise TypeError(f"{type(self._objective).__name__} does not produce a Hessian")
        return r


class _JVPDriver:
    """jvp: the solver-differentiating objective; its Newton uses the deck's tolerances
    (reference sensitivity.py:84-100 -- `max_ls_evals` is not forwarded to the traced solver)."""

    def __init__(self, qoi, F, newton_kwargs: dict[str, Any]) -> None:
        update = make_newton_solve(qoi.model()._residual, max_iters=newton_kwargs["max_iters"],
                                   abs_tol=newton_kwargs["abs_tol"], rel_tol=newton_kwargs["rel_tol"])
        self._objective = MPJVPObjective(qoi, F, update)

    def evaluate_grad(self, x) -> GradientResult:
        J, g = self._objective.evaluate_objective_and_grad(x)
        return GradientResult(J=float(J), grad=np.asarray(g, dtype=np.float64))

    def evaluate_hess(self, x) -> HessianResult:
        J, g = self._objective.evaluate_objective_and_grad(x)
        H = self._objective.evaluate_hessian(x)
        return HessianResult(J=float(J), grad=np.asarray(g, dtype=np.float64), hessian=np.asarray(H, dtype=np.float64))


def build_sensitivity_driver(sensitivity_section, qoi, F, newton_kwargs, subcommand: str):
    kind = sensitivity_section["type"]
    if subcommand == "hessian" and kind in ("adjoint", "direct"):
        raise ValueError(f"sensitivity.type: 'cmad hessian' requires 'direct_adjoint' or 'jvp'; got {kind!r}")
    if subcommand == "calibrate" and kind == "direct_adjoint":
        raise ValueError("sensitivity.type: 'cmad calibrate' accepts 'adjoint', 'direct', or 'jvp' "
                         f"(first-order only); got {kind!r}")
    if subcommand == "gradient" and kind == "direct_adjoint":
        print("warning: sensitivity.type=direct_adjoint computes a Hessian as a side effect; for gradient-only "
              "work prefer 'adjoint', 'direct', or 'jvp'", file=sys.stderr)
    if kind == "jvp":
        return _JVPDriver(qoi, F, newton_kwargs)
    table = {"adjoint": MPAdjointObjective, "direct": MPDirectObjective, "direct_adjoint": MPDirectAdjointObjective}
    if kind not in table:
        raise ValueError(f"sensitivity.type: unknown value {kind!r}")
    return _StoredStateDriver(table[kind](qoi, F))
