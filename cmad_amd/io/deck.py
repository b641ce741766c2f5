"""YAML deck: load, normalise, fill defaults, validate -- the `problem.type: material_point` subset.

Behaviour follows /root/reference/cmad/io/deck.py:96-204 (load, single-key wrapper unwrap, Calibr8-only section
strip, defaults for `solver.newton`, `output`, `optimizer`) and the per-subcommand required sections of
/root/reference/cmad/io/schema.py + cmad/io/schemas/*.yaml.  The reference validates with jsonschema; that
package is not assumed here, so the same constraints are written out as plain checks that raise ValueError
with the offending dotted path in the message.  `problem.type: fe` decks are rejected: the FE driver is outside
this package's scope."""
from __future__ import annotations

import copy
import warnings
from pathlib import Path
from typing import Any

import yaml

NEWTON_DEFAULTS = {"max_iters": 10, "abs_tol": 1e-14, "rel_tol": 1e-14, "max_ls_evals": 0}      # deck.py:46-53
OPTIMIZER_DEFAULTS = {"initial_guess": "from_deck", "options": {}, "log_params": True}         # deck.py:54-58
IGNORED_SECTIONS = ("linear algebra", "regression")                                            # deck.py:94

DEF_TYPES = ("full_3d", "plane_stress", "uniaxial_stress")
SENSITIVITY_TYPES = ("adjoint", "direct", "direct_adjoint", "jvp")
OUTPUT_FORMATS = ("npy", "text")

# sections each subcommand needs beyond problem/model/parameters/deformation (schema.py)
_NEEDS = {
    "primal": (),
    "objective": ("qoi",),
    "gradient": ("qoi", "sensitivity"),
    "hessian": ("qoi", "sensitivity"),
    "calibrate": ("qoi", "sensitivity", "optimizer"),
}
_KNOWN_TOP = {"problem", "model", "parameters", "deformation", "solver", "qoi", "sensitivity", "optimizer", "output"}


def load_deck(path: Path) -> dict[str, Any]:
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"deck not found: {path}")
    with path.open("r") as fh:
        tree = yaml.safe_load(fh)
    if tree is None:
        raise ValueError(f"deck is empty: {path}")
    if not isinstance(tree, dict):
        raise ValueError(f"deck top-level must be a mapping; got {type(tree).__name__} at {path}")
    return tree


def unwrap_top_level(deck: dict[str, Any]) -> dict[str, Any]:
    """`{name: {problem: ..., ...}}` -> the inner mapping (Calibr8 style); anything else passes through."""
    if len(deck) == 1:
        (inner,) = deck.values()
        if isinstance(inner, dict) and "problem" in inner:
            return inner
    return deck


def strip_calibr8_only(deck: dict[str, Any]) -> dict[str, Any]:
    present = [s for s in IGNORED_SECTIONS if s in deck]
    if not present:
        return deck
    kept = {k: v for k, v in deck.items() if k not in present}
    for s in present:
        warnings.warn(f"deck section '{s}' is recognized but unused by cmad (Calibr8-only); ignored",
                      UserWarning, stacklevel=3)
    return kept


def apply_deck_defaults(deck: dict[str, Any]) -> dict[str, Any]:
    out = strip_calibr8_only(unwrap_top_level(copy.deepcopy(deck)))
    is_mp = out.get("problem", {}).get("type") == "material_point"
    if is_mp:
        newton = out.setdefault("solver", {}).setdefault("newton", {})
        for k, v in NEWTON_DEFAULTS.items():
            newton.setdefault(k, v)
    if "output" in out:                       # an absent output block means "write nothing" for primal
        out["output"].setdefault("prefix", "")
        if is_mp:
            out["output"].setdefault("format", "npy")
    if "optimizer" in out:
        for k, v in OPTIMIZER_DEFAULTS.items():
            out["optimizer"].setdefault(k, copy.deepcopy(v))
    return out


def _fail(where: str, what: str):
    raise ValueError(f"deck validation failed at '{where}': {what}")


def _mapping(node, where):
    if not isinstance(node, dict):
        _fail(where, f"expected a mapping, got {type(node).__name__}")
    return node


def _only(node, where, allowed):
    extra = sorted(set(node) - set(allowed))
    if extra:
        _fail(where, f"unknown key(s) {extra}; allowed: {sorted(allowed)}")


def _number(x, where):
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        _fail(where, f"expected a number, got {x!r}")


def validate_deck(deck: dict[str, Any], subcommand: str) -> None:
    """Raise ValueError unless `deck` (already defaulted) is a valid material-point deck for `subcommand`."""
    if subcommand not in _NEEDS:
        raise ValueError(f"unknown subcommand {subcommand!r}")
    deck = strip_calibr8_only(unwrap_top_level(deck))
    _mapping(deck, "<root>")
    ptype = _mapping(deck.get("problem", None), "problem").get("type")
    if ptype == "fe":
        raise NotImplementedError("problem.type 'fe': the finite-element driver is outside cmad_amd's scope "
                                  "(only the material-point path runs on the device)")
    if ptype != "material_point":
        _fail("problem.type", f"expected 'material_point', got {ptype!r}")
    _only(deck["problem"], "problem", {"type", "name"})
    for sec in ("model", "parameters", "deformation") + _NEEDS[subcommand]:
        if sec not in deck:
            _fail(sec, f"section required by 'cmad {subcommand}' is missing")
    _only(deck, "<root>", _KNOWN_TOP)

    from .registry import model_names, qoi_names
    model = _mapping(deck["model"], "model")
    if model.get("name") not in model_names():
        _fail("model.name", f"{model.get('name')!r} is not a registered model; known: {model_names()}")
    if model.get("def_type") not in DEF_TYPES:
        _fail("model.def_type", f"expected one of {DEF_TYPES}, got {model.get('def_type')!r}")
    _only(model, "model", {"name", "def_type", "effective_stress", "elastic_stress", "uniaxial_stress_idx"})
    if "effective_stress" in model and model["effective_stress"] not in ("J2", "hill", "barlat", "hosford"):
        _fail("model.effective_stress", f"unknown yield function {model['effective_stress']!r}")
    idx = model.get("uniaxial_stress_idx", 0)
    if isinstance(idx, bool) or not isinstance(idx, int) or idx < 0:
        _fail("model.uniaxial_stress_idx", f"expected a non-negative integer, got {idx!r}")

    _validate_parameters(_mapping(deck["parameters"], "parameters"), "parameters")

    deform = _mapping(deck["deformation"], "deformation")
    _only(deform, "deformation", {"history_file", "inline"})
    if ("history_file" in deform) == ("inline" in deform):
        _fail("deformation", "exactly one of 'history_file' / 'inline' is required")

    newton = _mapping(_mapping(deck.get("solver", {}), "solver").get("newton", {}), "solver.newton")
    _only(deck.get("solver", {}), "solver", {"newton"})
    _only(newton, "solver.newton", set(NEWTON_DEFAULTS))
    for k in ("abs_tol", "rel_tol"):                       # schemas/solver.yaml: exclusiveMinimum 0
        if k in newton:
            _number(newton[k], f"solver.newton.{k}")
            if not newton[k] > 0:
                _fail(f"solver.newton.{k}", f"expected a positive number, got {newton[k]!r}")
    for k, lowest in (("max_iters", 1), ("max_ls_evals", 0)):
        if k in newton and (isinstance(newton[k], bool) or not isinstance(newton[k], int) or newton[k] < lowest):
            _fail(f"solver.newton.{k}", f"expected an integer >= {lowest}, got {newton[k]!r}")

    if "qoi" in deck:
        qoi = _mapping(deck["qoi"], "qoi")
        if qoi.get("name") not in qoi_names():
            _fail("qoi.name", f"{qoi.get('name')!r} is not a registered QoI; known: {qoi_names()}")
        _only(qoi, "qoi", {"name", "data_file", "weight", "weight_file"})
        if "data_file" not in qoi:
            _fail("qoi.data_file", "required")
        if ("weight" in qoi) == ("weight_file" in qoi):
            _fail("qoi", "exactly one of 'weight' / 'weight_file' is required")
    if "sensitivity" in deck:
        sens = _mapping(deck["sensitivity"], "sensitivity")
        _only(sens, "sensitivity", {"type"})
        if sens.get("type") not in SENSITIVITY_TYPES:
            _fail("sensitivity.type", f"expected one of {SENSITIVITY_TYPES}, got {sens.get('type')!r}")
    if "optimizer" in deck:
        opt = _mapping(deck["optimizer"], "optimizer")
        _only(opt, "optimizer", {"algorithm", "initial_guess", "options", "log_params"})
        if not isinstance(opt.get("algorithm"), str):
            _fail("optimizer.algorithm", "required (a scipy.optimize.minimize method name)")
        guess = opt.get("initial_guess", "from_deck")
        if not (guess == "from_deck" or (isinstance(guess, list) and all(isinstance(g, (int, float)) for g in guess))):
            _fail("optimizer.initial_guess", "expected 'from_deck' or a list of numbers")
    if "output" in deck:
        outp = _mapping(deck["output"], "output")
        _only(outp, "output", {"path", "prefix", "format"})
        if not isinstance(outp.get("path"), str):
            _fail("output.path", "required")
        if outp.get("format", "npy") not in OUTPUT_FORMATS:
            _fail("output.format", f"expected one of {OUTPUT_FORMATS}, got {outp.get('format')!r}")


def _validate_parameters(node, where):
    """Leaves are numbers / (nested) lists of numbers, or `{value, active?, transform?}` envelopes."""
    for key, child in node.items():
        here = f"{where}.{key}"
        if isinstance(child, dict) and "value" in child:
            _only(child, here, {"value", "active", "transform"})
            if "active" in child and not isinstance(child["active"], bool):
                _fail(f"{here}.active", "expected a boolean")
            tr = child.get("transform")
            if tr is not None:
                ok = isinstance(tr, dict) and len(tr) == 1 and next(iter(tr)) in ("bounds", "log")
                if not ok:
                    _fail(f"{here}.transform", "expected {bounds: [lo, hi]} or {log: ref}")
        elif isinstance(child, dict):
            _validate_parameters(child, here)
        elif not isinstance(child, (int, float, list)):
            _fail(here, f"unsupported leaf {child!r}")
