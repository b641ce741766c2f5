"""Deformation-gradient history reader -> float64 array of shape (n, n, N+1).

Formats are the reference's (/root/reference/cmad/io/deformation.py:45-120): `.npy` holding (n, n, N) or
(N, n, n) (the first wins when N == n); `.csv` (comma) / `.txt` (whitespace) with one row-major flattened
matrix per line; or an inline step-first list in the deck.  n must equal the model's `ndims`."""
from __future__ import annotations

import math
from pathlib import Path
from typing import Any

import numpy as np


def load_history(deformation_section: dict[str, Any], expected_ndims: int) -> np.ndarray:
    if "history_file" in deformation_section:
        hist = _read_file(Path(deformation_section["history_file"]))
    elif "inline" in deformation_section:
        steps = np.asarray(deformation_section["inline"], dtype=np.float64)
        if steps.ndim != 3 or steps.shape[1] != steps.shape[2]:
            raise ValueError("deformation.inline: expected a list of n-by-n matrices yielding shape (N, n, n); "
                             f"got {steps.shape}")
        hist = np.ascontiguousarray(np.moveaxis(steps, 0, 2))
    else:
        raise ValueError("deformation: must contain either 'history_file' or 'inline'")
    if hist.shape[0] != expected_ndims:
        raise ValueError(f"deformation: shape (n, n, N) with n={hist.shape[0]} does not match the model's expected "
                         f"ndims={expected_ndims} (full_3d→3, plane_stress/plane_strain→2, "
                         "uniaxial_stress/uniaxial_strain→1)")
    return hist


def _read_file(path: Path) -> np.ndarray:
    if not path.exists():
        raise FileNotFoundError(f"deformation.history_file: file not found at {path}")
    kind = path.suffix.lower()
    if kind == ".npy":
        raw = np.load(path).astype(np.float64)
    elif kind in (".csv", ".txt"):
        rows = np.loadtxt(path, delimiter="," if kind == ".csv" else None, ndmin=2).astype(np.float64)
        n = math.isqrt(rows.shape[1])
        if n * n != rows.shape[1]:
            raise ValueError("deformation.history_file: expected n*n columns per row (flattened n-by-n matrix); "
                             f"got {rows.shape[1]} columns in {path}")
        raw = rows.reshape(-1, n, n)
    else:
        raise ValueError(f"deformation.history_file: unsupported extension '{kind}' (path: {path}); "
                         "supported: .npy, .csv, .txt")
    if raw.ndim == 3 and raw.shape[0] == raw.shape[1]:
        return raw
    if raw.ndim == 3 and raw.shape[1] == raw.shape[2]:
        return np.ascontiguousarray(np.moveaxis(raw, 0, 2))
    raise ValueError(f"deformation: expected shape (n, n, N) or (N, n, n); got {raw.shape}")
