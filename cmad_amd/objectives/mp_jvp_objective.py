"""`MPJVPObjective`: objective, gradient and Hessian of a material-point calibration by differentiating the whole
time loop through the local solver's implicit-function rule.  Host mirror of
/root/reference/cmad/objectives/mp_jvp_objective.py:14-80 (constructor `(qoi, global_state, update_fun)`,
`evaluate_objective(x)`, `evaluate_objective_and_grad(x)`, `evaluate_hessian(x)`, canonical active values in,
derivatives w.r.t. the canonical values out).

The reference lets JAX trace `fori_loop(update_fun)` and transpose the solver's `custom_jvp`
(`value_and_grad`, :31-33) into one program.  Here the whole history is one `cm_objective_grad_history` launch with
the settings of `update_fun = make_newton_solve(model._residual, ...)`: the forward loop with the states stored and
the reverse sweep of the same implicit-function transposition, written as a kernel (and usable with any batch size
through `BatchedCalibrationObjective`, of which this is the one-point case).  The Hessian reuses the direct-adjoint
contraction of `MPDirectAdjointObjective`, which is the same quantity `jax.hessian` returns.
"""
from __future__ import annotations

import numpy as np

from .batched import BatchedCalibrationObjective
from .mp_objective import MPDirectAdjointObjective

_V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]


class MPJVPObjective:
    def __init__(self, qoi, global_state, update_fun) -> None:
        import torch
        from ..models.device import fold_weight_and_data
        self._qoi, self._F = qoi, np.asarray(global_state)
        model = qoi.model()
        if getattr(update_fun, "model", None) is not model:
            raise NotImplementedError("update_fun must come from make_newton_solve(model._residual) of the QoI's model")
        self._model, self._settings = model, update_fun.settings
        nd = self._F.shape[0]
        K = self._F.shape[2] - 1
        gh = np.stack([(self._F[:, :, k] - np.eye(nd)).reshape(nd * nd, 1) for k in range(K + 1)])
        weight = np.asarray(qoi.weight())
        _, data6, const = fold_weight_and_data(weight, np.asarray(qoi.data()))      # (6, K+1), (K+1,)
        self._const = float(np.sum(const[1:]))
        dh = np.ascontiguousarray(data6.T[:, :, None])                             # (K+1, 6, 1)
        dev = torch.device("cuda")
        self._batched = BatchedCalibrationObjective(model, torch.from_numpy(gh).to(dev).contiguous(),
                                                    torch.from_numpy(dh).to(dev).contiguous(), weight,
                                                    newton=self._settings)

    def evaluate_objective_and_grad(self, flat_active_values):
        r = self._batched.evaluate(np.asarray(flat_active_values, dtype=np.float64))
        return r.J + self._const, r.grad

    def evaluate_objective(self, flat_active_values):
        return self.evaluate_objective_and_grad(flat_active_values)[0]

    def evaluate_hessian(self, flat_active_values):
        return MPDirectAdjointObjective(self._qoi, self._F).evaluate(np.asarray(flat_active_values, dtype=np.float64)).hessian
