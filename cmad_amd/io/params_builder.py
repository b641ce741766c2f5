"""Deck `parameters:` tree -> `Parameters` (/root/reference/cmad/io/params_builder.py:27-74).

A leaf is a bare number / list (inactive, untransformed) or an envelope `{value, active?, transform?}` with
`transform` one of `{bounds: [lo, hi]}` / `{log: ref}`.  Lists become float64 arrays, integers become floats."""
from __future__ import annotations

from typing import Any

import numpy as np

from ..parameters import Parameters


def _value(v: Any) -> Any:
    if isinstance(v, list):
        return np.asarray(v, dtype=np.float64)
    if isinstance(v, int) and not isinstance(v, bool):
        return float(v)
    return v


def _transform(spec: Any):
    if spec is None:
        return None
    if isinstance(spec, dict) and "bounds" in spec:
        return np.asarray(spec["bounds"], dtype=np.float64)
    if isinstance(spec, dict) and "log" in spec:
        return np.asarray([spec["log"]], dtype=np.float64)
    raise ValueError(f"unknown transform spec: {spec!r}")


def split_parameter_tree(node: Any) -> tuple[Any, Any, Any]:
    """-> (values, active_flags, transforms), three trees of identical structure."""
    if isinstance(node, dict):
        if "value" in node:
            return _value(node["value"]), bool(node.get("active", False)), _transform(node.get("transform"))
        triples = {k: split_parameter_tree(v) for k, v in node.items()}
        return tuple({k: t[i] for k, t in triples.items()} for i in range(3))
    return _value(node), False, None


def build_parameters(parameters_section: dict[str, Any]) -> Parameters:
    values, active, transforms = split_parameter_tree(parameters_section)
    return Parameters(values=values, active_flags=active, transforms=transforms)
