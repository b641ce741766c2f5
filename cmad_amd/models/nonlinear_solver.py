"""Local Newton drivers with the reference's names (/root/reference/cmad/models/nonlinear_solver.py).

* `newton_solve(model, ...)` (:14-85): the imperative solve used by the material-point objectives.  With
  `max_ls_evals == 0` (the default everywhere in the reference) the whole iteration runs in ONE `cm_update`
  launch; with the legacy backtracking (`max_ls_evals > 0`) the reference's Python loop is kept, each
  `model.evaluate()` being a `cm_evaluate` launch.
* `make_newton_solve(residual, ...)` (:88-174): returns `solve(xi_prev, params, U, U_prev) -> xi` backed by
  `cm_update` with the quadratic Armijo line search; `residual` must be a bound `Model._residual`.
"""
from __future__ import annotations

import numpy as np

from .device import DEFAULT_LINE_SEARCH_SETTINGS, NewtonSettings


def newton_solve(model, max_iters: int = 10, abs_tol: float = 1e-14, rel_tol: float = 1e-14,
                 max_ls_evals: int = 0):
    if max_ls_evals == 0 and hasattr(model, "device_newton") and getattr(model, "has_device_newton", True):
        iters, _ = model.device_newton(max_iters, abs_tol, rel_tol)
        model.seed_none()
        model.evaluate()
        return iters, float(np.linalg.norm(model.C()))

    converged = False
    ii = 0
    C_norm_0 = 1.
    C_norm = 0.
    beta, eta = 1e-4, 0.5
    while ii < max_iters and not converged:
        model.seed_none()
        model.evaluate()
        Cv = model.C()
        C_norm = np.linalg.norm(Cv)
        C_norm_rel = 1. if ii == 0 else C_norm / C_norm_0
        if ii == 0:
            C_norm_0 = C_norm
        if C_norm_rel < rel_tol or C_norm < abs_tol:
            converged = True
            break
        model.seed_xi()
        model.evaluate()
        delta_xi = np.linalg.solve(model.Jac(), -Cv)
        model.add_to_xi(delta_xi)
        if max_ls_evals > 0:
            model.seed_none()
            model.evaluate()
            psi_0 = 0.5 * C_norm ** 2
            psi_0_deriv = -2. * psi_0
            jj = 1
            alpha_j = 1.
            psi_j = 0.5 * np.linalg.norm(model.C()) ** 2
            while psi_j >= ((1. - 2. * beta * alpha_j) * psi_0):
                alpha_prev = alpha_j
                alpha_j = max(eta * alpha_j, -(alpha_j ** 2 * psi_0_deriv) / (2. * (psi_j - psi_0 - alpha_j * psi_0_deriv)))
                if jj == max_ls_evals:
                    print("reached max ls evals")
                    break
                jj += 1
                model.add_to_xi((alpha_j - alpha_prev) * delta_xi)
                model.evaluate()
                psi_j = 0.5 * np.linalg.norm(model.C()) ** 2
        ii += 1
    return ii, float(C_norm)


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
