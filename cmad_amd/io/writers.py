"""Result writers for the material-point subcommands.  File names, shapes and JSON layouts are those of
/root/reference/cmad/io/writers.py:63-186, 397-520:

  <prefix>cauchy.npy (3, 3, N+1) | cauchy.csv (N+1 rows x 9, header "S11 ... S33")
  <prefix>xi_block_<kk>.npy | .csv   (N+1, block size), one per residual block
  <prefix>solver.json  [{iters, final_residual}, ...]      <prefix>J.json  {"J": value}
  <prefix>grad.npy | .csv   <prefix>hess.npy | .csv        <prefix>deck.resolved.yaml
  <prefix>opt_history.json  <prefix>opt_params.yaml  <prefix>opt_status.json   (calibrate)"""
from __future__ import annotations

import copy
import json
from pathlib import Path
from typing import Any

import numpy as np
import yaml

CAUCHY_COLUMNS = "S11 S12 S13 S21 S22 S23 S31 S32 S33"


def _array(out_dir: Path, stem: str, arr, fmt: str, **savetxt_kw) -> None:
    if fmt == "npy":
        np.save(out_dir / f"{stem}.npy", arr)
    elif fmt == "text":
        np.savetxt(out_dir / f"{stem}.csv", arr, **savetxt_kw)
    else:
        raise ValueError(f"output.format: expected 'npy' or 'text', got {fmt!r}")


def _json(out_dir: Path, name: str, payload) -> None:
    with (out_dir / name).open("w") as fh:
        json.dump(payload, fh, indent=2)


def write_cauchy(out_dir: Path, prefix: str, cauchy, fmt: str) -> None:
    if fmt == "text":
        _array(out_dir, f"{prefix}cauchy", np.moveaxis(cauchy, 2, 0).reshape(-1, 9), fmt, header=CAUCHY_COLUMNS)
    else:
        _array(out_dir, f"{prefix}cauchy", cauchy, fmt)


def write_xi(out_dir: Path, prefix: str, xi_trajectory, fmt: str) -> None:
    """`xi_trajectory[step][block]` is a 1-D array."""
    if fmt not in ("npy", "text"):
        raise ValueError(f"output.format: expected 'npy' or 'text', got {fmt!r}")
    if not xi_trajectory:
        return
    for k in range(len(xi_trajectory[0])):
        _array(out_dir, f"{prefix}xi_block_{k:02d}", np.stack([step[k] for step in xi_trajectory]), fmt)


def write_solver_log(out_dir: Path, prefix: str, solver_log) -> None:
    _json(out_dir, f"{prefix}solver.json", solver_log)


def write_J(out_dir: Path, prefix: str, J: float) -> None:
    _json(out_dir, f"{prefix}J.json", {"J": J})


def write_grad(out_dir: Path, prefix: str, grad, fmt: str) -> None:
    _array(out_dir, f"{prefix}grad", grad, fmt)


def write_hessian(out_dir: Path, prefix: str, hessian, fmt: str) -> None:
    _array(out_dir, f"{prefix}hess", hessian, fmt)


def write_resolved_deck(out_dir: Path, prefix: str, resolved_deck: dict[str, Any]) -> None:
    with (out_dir / f"{prefix}deck.resolved.yaml").open("w") as fh:
        yaml.safe_dump(resolved_deck, fh, default_flow_style=False, sort_keys=False)


def write_opt_history(out_dir: Path, prefix: str, history, active_param_paths=None) -> None:
    payload: dict[str, Any] = {"history": history}
    if active_param_paths is not None:
        payload["active_param_paths"] = active_param_paths
    _json(out_dir, f"{prefix}opt_history.json", payload)


def _plain(x: Any) -> Any:
    return x.tolist() if hasattr(x, "tolist") else x


def _overlay(deck_node: Any, values_node: Any) -> Any:
    if isinstance(deck_node, dict):
        if "value" in deck_node:
            return {**deck_node, "value": _plain(values_node)}
        return {k: _overlay(v, values_node[k]) for k, v in deck_node.items()}
    return _plain(values_node)


def write_opt_params(out_dir: Path, prefix: str, deck_parameters: dict[str, Any], current_values: Any) -> None:
    """The deck's `parameters:` subtree with every leaf value replaced by the current raw value (envelope
    metadata kept), wrapped as `{parameters: ...}` so it can be pasted into a follow-up deck."""
    tree = _overlay(copy.deepcopy(deck_parameters), current_values)
    with (out_dir / f"{prefix}opt_params.yaml").open("w") as fh:
        yaml.safe_dump({"parameters": tree}, fh, default_flow_style=False, sort_keys=False)


def write_opt_status(out_dir: Path, prefix: str, status: dict[str, Any]) -> None:
    _json(out_dir, f"{prefix}opt_status.json", status)
