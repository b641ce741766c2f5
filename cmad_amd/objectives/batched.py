"""Batched calibration objective: the mathematics of `MPAdjointObjective`
(/root/reference/cmad/objectives/mp_objective.py:95-147) + `Calibration._qoi` (qois/calibration.py:56-66)
for B independent Gauss points at once, sharded over ranks.

    J(p)    = sum_b sum_k 1/2 || w o (sigma_b,k(p) - data_b,k) ||^2
    grad(p) = dJ/dp in Parameters' flat active order, canonical (transform_grad applied) like the reference

A K-step history is one `cm_objective_grad_history` launch (state in registers from step to step); with
`fused_history=False` -- the default for the configurations whose `cm_update` runs on the work pool (network surfaces, Hosford
under the line search) -- one `cm_update` launch per step forward and one `cm_adjoint_step` launch per step in reverse.
A single-step history uses the fused `cm_objective_grad`.  Multi-GPU: each rank owns a contiguous shard of
the points; the only exchange is one all-reduce of (1 + 12) doubles per evaluation (RCCL on GPUs).
"""
from __future__ import annotations

import numpy as np

from ..models.device import fold_weight_and_data
from ..typing import GradientResult


def shard_bounds(B: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of B points for `rank` (sizes differ by at most one)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


COLLECTIVE_CALLS = 0        # how many result vectors went through dist.all_reduce in this process (tests assert on it)


def allreduce_sum_(t, group=None):
    """In-place SUM all-reduce of the (1 + CM_NUM_PARAMS) result vector whenever torch.distributed is initialised -- also for
    a one-rank group, so that a single-GPU rehearsal sends the vector through the same ncclAllReduce the N-GPU job uses
    (backend nccl == RCCL on GPUs; gloo in the CPU tests).  Without a process group there is nothing to exchange."""
    global COLLECTIVE_CALLS
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        COLLECTIVE_CALLS += 1
    return t


class BatchedCalibrationObjective:
    """`evaluate(flat_active_values) -> GradientResult` like `MPObjective.evaluate` (canonical values in,
    canonical gradient out), over this rank's shard of the batch.

    gradu_hist: (K+1, n_gradu, B_local) CUDA float64 (step 0 = reference configuration)
    data_hist : (K+1, 6, B_local) CUDA float64 measured stresses (6 stored entries), step 0 unused
    weight    : (3, 3) constant mask (Calibration's convention)
    xi0       : (n_xi, B_local) initial state (default: the model's init values)
    """

    def __init__(self, model, gradu_hist, data_hist, weight, xi0=None, newton=None, group=None, fused_history=None):
        import torch
        self._model = model
        self._parameters = model.parameters
        self._g, self._d = gradu_hist, data_hist
        self._K = gradu_hist.shape[0] - 1
        self._B = gradu_hist.shape[2]
        self._wsq6 = fold_weight_and_data(weight)
        self._newton = newton
        self._group = group
        dev = gradu_hist.device
        if xi0 is None:
            init = np.concatenate([np.atleast_1d(b) for b in model._init_xi]).astype(np.float64)
            xi0 = torch.from_numpy(init).to(dev)[:, None].repeat(1, self._B).contiguous()
        self._xi0 = xi0
        self._out = torch.zeros(13, dtype=torch.float64, device=dev)
        # K > 1: one `cm_objective_grad_history` launch per evaluation (False: one launch per step and direction).  None picks:
        # the single launch, except for the iteration-bound configurations whose `cm_update` runs on the work pool
        # (`DeviceEvaluator.pool_route`) -- for them per-step launches (work-pool update forward, adjoint step backward) are faster
        # than the lockstep history kernel
        self._fused_history = fused_history
        self._xi_hist = None

    def evaluate(self, flat_active_values) -> GradientResult:
        self._parameters.set_active_values_from_flat(flat_active_values)
        return self._evaluate()

    def evaluate_native(self) -> GradientResult:
        """Objective and gradient at the CURRENT parameter values, gradient w.r.t. native parameters."""
        return self._evaluate(transform=False)

    def _evaluate(self, transform=True) -> GradientResult:
        import torch
        model = self._model
        ev = model.device_evaluator(self._newton)
        g, d, K = self._g, self._d, self._K
        out = self._out
        rate = getattr(model, "_model_kind", 0) == 1          # the rate form also takes the previous step's grad u
        prev = (lambda k: {"gradu_prev": g[k - 1]}) if rate else (lambda k: {})
        pooled = (not rate) and (ev.pool_route(self._B) or ev.screened(self._B))      # update kernel(s) + reverse kernel per step
        fused = (not pooled) if self._fused_history is None else bool(self._fused_history)
        if K == 1:
            # with a state buffer the entry point routes the iteration-bound configurations through the work pool
            ev.objective_grad(g[1], self._xi0, d[1], self._wsq6, out=out, want_xi=pooled, **prev(1))
        elif fused:
            _, self._xi_hist = ev.objective_grad_history(g, d, self._wsq6, self._xi0, xi_hist=self._xi_hist, out=out)
        else:
            xs = [self._xi0]
            for k in range(1, K + 1):
                if rate:
                    x, _, _ = ev.update_rate(g[k], g[k - 1], xs[-1], want_sigma=False, want_status=False)
                else:
                    x, _, _ = ev.update(g[k], xs[-1], want_sigma=False, want_status=False)
                xs.append(x)
            out.zero_()
            hist = torch.zeros_like(self._xi0)
            for k in range(K, 0, -1):
                ev.adjoint_step(g[k], xs[k - 1], xs[k], d[k], self._wsq6, hist, hist, out, accumulate=True, **prev(k))
        allreduce_sum_(out, self._group)
        res = out.cpu().numpy()
        grad = model.active_grad_from_kp(res[1:], ev.info)
        if transform:
            self._parameters.transform_grad(grad)
        return GradientResult(J=float(res[0]), grad=grad)
