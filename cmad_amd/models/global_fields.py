"""Global fields at an evaluation point (mirror of /root/reference/cmad/models/global_fields.py:12-41)."""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class GlobalFieldsAtPoint:
    """Interpolated global fields + their gradients at an evaluation point.
    ``grad_fields["u"][k, j] = d u_k / d x_j``."""
    fields: dict
    grad_fields: dict


def mp_U_from_F(F) -> GlobalFieldsAtPoint:
    """Material-point U from a prescribed F: grad_fields['u'] = F - I."""
    F = np.asarray(F)
    n = F.shape[0]
    return GlobalFieldsAtPoint(fields={"u": np.zeros(n)}, grad_fields={"u": F - np.eye(n)})
