"""`cmad` entry point: `python -m cmad_amd.cli.main {primal,objective,gradient,hessian,calibrate} deck.yaml`.
Same subcommands, exit codes and output files as /root/reference/cmad/cli/main.py:21-64 for
`problem.type: material_point` decks."""
from __future__ import annotations

import argparse
import sys
from pathlib import Path
from typing import Any

import numpy as np

from ..io.writers import (write_grad, write_hessian, write_J, write_opt_history, write_opt_params,
                          write_opt_status, write_resolved_deck)
from .common import active_param_paths, build_mp_problem, resolve_output
from .primal import run_primal, run_primal_pass, write_primal_outputs
from .sensitivity import build_sensitivity_driver


def run_objective(deck_path: Path) -> int:
    """Forward pass with the QoI accumulated; primal outputs + J.json (reference cli/objective.py:52-70)."""
    problem = build_mp_problem(deck_path, "objective")
    cauchy, trajectory, log, J = run_primal_pass(problem.model, problem.F, problem.F.shape[2] - 1,
                                                 problem.resolved["solver"]["newton"], qoi=problem.qoi)
    out_dir, prefix, _ = write_primal_outputs(problem.resolved, cauchy, trajectory, log)
    write_J(out_dir, prefix, J)
    return 0


def _driver(problem, subcommand):
    return build_sensitivity_driver(problem.resolved["sensitivity"], problem.qoi, problem.F,
                                    problem.resolved["solver"]["newton"], subcommand=subcommand)


def run_gradient(deck_path: Path) -> int:
    """J.json + grad.{npy,csv} in canonical coordinates (reference cli/gradient.py:53-71)."""
    problem = build_mp_problem(deck_path, "gradient")
    r = _driver(problem, "gradient").evaluate_grad(problem.parameters.flat_active_values(return_canonical=True))
    out_dir, prefix, fmt = resolve_output(problem.resolved)
    write_resolved_deck(out_dir, prefix, problem.resolved)
    write_J(out_dir, prefix, float(r.J))
    write_grad(out_dir, prefix, r.grad, fmt)
    return 0


def run_hessian(deck_path: Path) -> int:
    """J.json + grad + hess (reference cli/hessian.py:58-77)."""
    problem = build_mp_problem(deck_path, "hessian")
    r = _driver(problem, "hessian").evaluate_hess(problem.parameters.flat_active_values(return_canonical=True))
    out_dir, prefix, fmt = resolve_output(problem.resolved)
    write_resolved_deck(out_dir, prefix, problem.resolved)
    write_J(out_dir, prefix, float(r.J))
    write_grad(out_dir, prefix, r.grad, fmt)
    write_hessian(out_dir, prefix, r.hessian, fmt)
    return 0


def run_calibrate(deck_path: Path) -> int:
    """scipy.optimize.minimize(jac=True) over the canonical active parameters; writes opt_history.json,
    opt_params.yaml, opt_status.json (reference cli/calibrate.py:72-125, 210-236)."""
    from scipy.optimize import minimize

    problem = build_mp_problem(deck_path, "calibrate")
    parameters = problem.parameters
    driver = _driver(problem, "calibrate")
    opt = problem.resolved["optimizer"]
    x0 = parameters.flat_active_values(return_canonical=True)
    if opt["initial_guess"] != "from_deck":
        x0 = np.asarray(opt["initial_guess"], dtype=np.float64)
    log_params = bool(opt["log_params"])
    trace: list[dict[str, Any]] = []

    def fun(x):
        r = driver.evaluate_grad(x)
        entry: dict[str, Any] = {"J": float(r.J), "grad_norm": float(np.linalg.norm(r.grad))}
        if log_params:
            entry["params"] = parameters.flat_active_values(return_canonical=False).tolist()
        trace.append(entry)
        return float(r.J), r.grad

    result = minimize(fun, x0, jac=True, method=opt["algorithm"], bounds=parameters.opt_bounds,
                      options=opt["options"])
    parameters.set_active_values_from_flat(result.x, are_canonical=True)

    status: dict[str, Any] = {"success": bool(result.success), "status": int(result.status),
                              "message": str(result.message), "fun": float(result.fun)}
    for counter in ("nfev", "njev", "nhev", "nit"):
        if getattr(result, counter, None) is not None:
            status[counter] = int(getattr(result, counter))

    out_dir, prefix, _ = resolve_output(problem.resolved)
    write_resolved_deck(out_dir, prefix, problem.resolved)
    write_opt_history(out_dir, prefix, trace, active_param_paths(parameters) if log_params else None)
    write_opt_params(out_dir, prefix, problem.resolved["parameters"], parameters.values)
    write_opt_status(out_dir, prefix, status)
    return 0


_SUBCOMMANDS = {
    "primal": (run_primal, "Run a forward (primal) solve."),
    "objective": (run_objective, "Run a forward solve and accumulate the QoI J."),
    "gradient": (run_gradient, "Compute (J, grad) via the chosen sensitivity strategy."),
    "hessian": (run_hessian, "Compute (J, grad, hess) via direct_adjoint or jvp."),
    "calibrate": (run_calibrate, "Optimize active parameters against the QoI via scipy."),
}


def main(argv: list[str] | None = None) -> int:
    parser = argparse.ArgumentParser(prog="cmad")
    sub = parser.add_subparsers(dest="subcommand", required=True)
    for name, (_, text) in _SUBCOMMANDS.items():
        sub.add_parser(name, help=text).add_argument("deck", type=Path, help="Path to the YAML deck.")
    args = parser.parse_args(argv)
    return _SUBCOMMANDS[args.subcommand][0](args.deck)


if __name__ == "__main__":
    sys.exit(main())
