"""Name -> class tables for the deck's `model.name` and `qoi.name`.

Same role as /root/reference/cmad/io/registry.py:1-224 (the reference discovers names from its schema directory
and imports lazily; here the table is explicit because every registered class lives in this package).
`register_model` / `register_qoi` let a caller add or override an entry."""
from __future__ import annotations

_MODELS: dict[str, type] = {}
_QOIS: dict[str, type] = {}


def _populate() -> None:
    if _MODELS:
        return
    from ..models import SmallElasticPlastic, SmallRateElasticPlastic
    from ..qois import Calibration, UniaxialCalibration
    for cls in (SmallElasticPlastic, SmallRateElasticPlastic):
        _MODELS.setdefault(cls.registry_name, cls)
    for cls in (Calibration, UniaxialCalibration):
        _QOIS.setdefault(cls.registry_name, cls)


def register_model(name: str, cls: type) -> None:
    _populate()
    _MODELS[name] = cls


def register_qoi(name: str, cls: type) -> None:
    _populate()
    _QOIS[name] = cls


def model_names() -> list[str]:
    _populate()
    return sorted(_MODELS)


def qoi_names() -> list[str]:
    _populate()
    return sorted(_QOIS)


def resolve_model(name: str) -> type:
    _populate()
    try:
        return _MODELS[name]
    except KeyError:
        raise ValueError(f"model.name: '{name}' is not registered; known: {model_names()}") from None


def resolve_qoi(name: str) -> type:
    _populate()
    try:
        return _QOIS[name]
    except KeyError:
        raise ValueError(f"qoi.name: '{name}' is not registered; known: {qoi_names()}") from None
