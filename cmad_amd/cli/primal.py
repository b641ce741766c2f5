"""`cmad primal`: forward solve of one material point over the deck's F history.
Counterpart of /root/reference/cmad/cli/primal.py:63-80, 129-176."""
from __future__ import annotations

from pathlib import Path
from typing import Any

import numpy as np

from ..io.writers import write_cauchy, write_resolved_deck, write_solver_log, write_xi
from ..models.global_fields import mp_U_from_F
from ..models.nonlinear_solver import newton_solve
from .common import build_mp_problem, resolve_output


def run_primal_pass(model, F, num_steps: int, newton_kwargs: dict[str, Any], qoi=None):
    """-> (cauchy (3,3,N+1), xi_trajectory[step][block], solver_log, J).  J stays 0.0 without a QoI."""
    cauchy = np.zeros((3, 3, num_steps + 1))
    model.set_xi_to_init_vals()
    trajectory = [[np.array(b, copy=True) for b in model.xi()]]
    log: list[dict[str, Any]] = []
    J = 0.0
    for step in range(1, num_steps + 1):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        iters, final_res = newton_solve(model, **newton_kwargs)
        model.advance_xi()
        model.evaluate_cauchy()
        cauchy[:, :, step] = model.Sigma()
        trajectory.append([np.array(b, copy=True) for b in model.xi()])
        log.append({"iters": int(iters), "final_residual": float(final_res)})
        if qoi is not None:
            model.seed_none()
            qoi.evaluate(step)
            J += float(np.asarray(qoi.J()))
    return cauchy, trajectory, log, J


def write_primal_outputs(resolved, cauchy, trajectory, log) -> tuple[Path, str, str]:
    out_dir, prefix, fmt = resolve_output(resolved)
    write_cauchy(out_dir, prefix, cauchy, fmt)
    write_xi(out_dir, prefix, trajectory, fmt)
    write_solver_log(out_dir, prefix, log)
    write_resolved_deck(out_dir, prefix, resolved)
    return out_dir, prefix, fmt


def run_primal(deck_path: Path) -> int:
    problem = build_mp_problem(deck_path, "primal")
    cauchy, trajectory, log, _ = run_primal_pass(problem.model, problem.F, problem.F.shape[2] - 1,
                                                 problem.resolved["solver"]["newton"])
    if "output" in problem.resolved:
        write_primal_outputs(problem.resolved, cauchy, trajectory, log)
    return 0
