"""Deck -> (parameters, model, F history, QoI) for the material-point subcommands, and the output location.
Counterpart of /root/reference/cmad/cli/common.py:59-152 (`MPProblem`, `build_mp_problem`, `resolve_output`)."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any

import numpy as np

from ..io.deck import apply_deck_defaults, load_deck, validate_deck
from ..io.deformation import load_history
from ..io.params_builder import build_parameters
from ..io.qoi_data import load_qoi_data
from ..io.registry import resolve_model, resolve_qoi
from ..models.deformation_types import DefType
from ..models.model import Model
from ..parameters import Parameters
from ..qois.qoi import QoI


@dataclass(frozen=True)
class MPProblem:
    resolved: dict[str, Any]
    parameters: Parameters
    model: Model
    F: np.ndarray
    qoi: QoI | None


def build_mp_problem(deck_path: Path, subcommand: str) -> MPProblem:
    resolved = apply_deck_defaults(load_deck(Path(deck_path)))
    validate_deck(resolved, subcommand)

    model_cls = resolve_model(resolved["model"]["name"])
    material = dict(resolved["parameters"])
    for key, default in model_cls.material_defaults().items():       # e.g. identity `rotation matrix`
        material.setdefault(key, default)
    parameters = build_parameters(material)
    def_type = DefType[resolved["model"]["def_type"].upper()]
    model = model_cls.from_deck(resolved["model"], parameters, def_type)
    F = load_history(resolved["deformation"], expected_ndims=model.ndims)

    qoi = None
    if subcommand != "primal":
        qoi_cls = resolve_qoi(resolved["qoi"]["name"])
        if qoi_cls.problem_type != "material_point":
            raise ValueError(f"qoi.name '{resolved['qoi']['name']}' is registered for problem_type="
                             f"'{qoi_cls.problem_type}', but the deck has problem.type='material_point'")
        data, weight = load_qoi_data(resolved["qoi"])
        qoi = qoi_cls.from_deck(resolved["qoi"], model, data, weight)
    return MPProblem(resolved=resolved, parameters=parameters, model=model, F=F, qoi=qoi)


def resolve_output(resolved: dict[str, Any]) -> tuple[Path, str, str]:
    """(directory [created], prefix, format); an absent `output:` block means cwd / "" / npy."""
    block = resolved.get("output", {})
    out_dir = Path(block.get("path", "."))
    out_dir.mkdir(parents=True, exist_ok=True)
    return out_dir, block.get("prefix", ""), block.get("format", "npy")


def active_param_paths(parameters: Parameters) -> list[str]:
    """Dotted labels of the active parameters in gradient order; blanks inside keys become underscores
    (/root/reference/cmad/cli/calibrate.py:255-278)."""
    return [".".join(str(seg).replace(" ", "_") for seg in path) for path in parameters.active_paths()]
