"""Local Newton drivers with the reference's names (/root/reference/cmad/models/nonlinear_solver.py).

* `newton_solve(model, ...)` (:14-85): the imperative solve used by the material-point objectives.  The whole iteration
  is ONE `cm_update` launch on the gathered state, with plain Newton steps (`max_ls_evals == 0`, the default everywhere in
  the reference) or with the legacy backtracking of :55-81 (`max_ls_evals > 0`: the kernels' CM_LS_LEGACY line search,
  include/cmad_hip.h).
* `make_newton_solve(residual, ...)` (:88-174): returns `solve(xi_prev, params, U, U_prev) -> xi` backed by
  `cm_update` with the quadratic Armijo line search; `residual` must be a bound `Model._residual`.
"""
from __future__ import annotations

import numpy as np

from .device import DEFAULT_LINE_SEARCH_SETTINGS, NewtonSettings


def newton_solve(model, max_iters: int = 10, abs_tol: float = 1e-14, rel_tol: float = 1e-14,
                 max_ls_evals: int = 0):
    """Returns (iterations, norm of the residual at the returned state) like the reference; the state is left in the model."""
    search = {"max evals": int(max_ls_evals), "kind": "legacy"} if max_ls_evals > 0 else None
    iters, _ = model.device_newton(max_iters, abs_tol, rel_tol, line_search=search)
    model.seed_none()
    model.evaluate()
    return iters, float(np.linalg.norm(model.C()))


def make_newton_solve(residual, max_iters: int = 10, abs_tol: float = 1e-14, rel_tol: float = 1e-14,
                      print_local_convergence: bool = False, line_search_settings=None):
    model = getattr(residual, "__self__", None)
    if model is None or not hasattr(model, "device_evaluator"):
        raise NotImplementedError("make_newton_solve needs a bound Model._residual (arbitrary residual callables "
                                  "cannot be traced into a HIP kernel)")
    settings = NewtonSettings(max_iters, abs_tol, rel_tol, {**DEFAULT_LINE_SEARCH_SETTINGS, **(line_search_settings or {})})

    def solve(xi_prev, params, U, U_prev=None):
        import torch
        from .device import DeviceEvaluator, build_desc
        desc, info = build_desc(params, def_type=model._def_type, model_kind=model._model_kind,
                                yield_tol=model._yield_tol, uniaxial_stress_idx=model._uniaxial_stress_idx,
                                newton=settings)
        ev = DeviceEvaluator(desc, info)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 1)).cuda()
        xi, _, _ = ev.update(t(np.asarray(U.grad_fields["u"])), t(model._flat(xi_prev)), want_sigma=False, want_status=False)
        return model._split(xi.cpu().numpy()[:, 0])

    solve.model, solve.settings = model, settings          # read by MPJVPObjective
    return solve
