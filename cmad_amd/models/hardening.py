"""Isotropic hardening laws and the lookup table models take as `hardening_funs`
(/root/reference/cmad/models/hardening.py:9-34).  The Voce and linear laws are built into the HIP kernels
(`cm_model_desc.has_voce / has_linear`, `cm::hardening` in csrc/cm_device.hpp); these host functions are the table's
entries -- a model recognises them by identity -- and the numpy statement of the same formulas for host-side use."""
from __future__ import annotations

import numpy as np


def voce_hardening(alpha, voce_params):
    return voce_params["S"] * (1. - np.exp(-voce_params["D"] * alpha))


def linear_hardening(alpha, linear_params):
    return linear_params["K"] * alpha


def get_hardening_funs():
    return {"voce": voce_hardening, "linear": linear_hardening}


def combined_hardening_fun(alpha, params, hardening_funs):
    return np.sum(np.array([hardening_funs[htype](alpha, hparams) for htype, hparams in params.items()]), axis=0)
