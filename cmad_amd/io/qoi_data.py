"""QoI data / weight readers for material-point decks (`.npy` only; /root/reference/cmad/io/qoi_data.py:31-37,
105-129).  Shapes are checked by the QoI constructors, not here."""
from __future__ import annotations

from pathlib import Path
from typing import Any

import numpy as np


def _npy(field: str, where: str) -> np.ndarray:
    path = Path(where)
    if not path.exists():
        raise FileNotFoundError(f"{field}: file not found at {path}")
    if path.suffix.lower() != ".npy":
        raise ValueError(f"{field}: unsupported extension '{path.suffix.lower()}' (path: {path}); supported: .npy")
    return np.load(path).astype(np.float64)


def load_qoi_data(qoi_section: dict[str, Any]) -> tuple[np.ndarray, np.ndarray]:
    data = _npy("qoi.data_file", qoi_section["data_file"])
    if "weight" in qoi_section:
        weight = np.asarray(qoi_section["weight"], dtype=np.float64)
    else:
        weight = _npy("qoi.weight_file", qoi_section["weight_file"])
    return data, weight
