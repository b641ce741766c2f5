"""Layout bridge for the FE COUPLED path.

In the reference each integration point runs `local_newton(xi_prev, params, U_ip, U_ip_prev)` and
`jacfwd` differentiates through its custom_jvp rule
(/root/reference/cmad/global_residuals/global_residual.py:341-400, driven per element by
cmad/fem/assembly.py:416-535 with state `xi (n_elems, n_ips, n_xi)`, :444-449).  This module does the
per-IP part for ALL (element, ip) pairs in one launch of `cm_update_tangent`:

    xi, sigma, dsigma_dgradu, status = local_update_with_tangent(model, grad_u, xi_prev)

    grad_u  (n_elems, n_ips, nd, nd)   interpolated displacement gradient, grad_u[..., k, j] = d u_k / d x_j
    xi_prev (n_elems, n_ips, n_xi)     the FE state layout (array-of-structures)
    ->  xi (n_elems, n_ips, n_xi), sigma (n_elems, n_ips, 3, 3),
        dsigma_dgradu (n_elems, n_ips, 3, 3, nd, nd), status (n_elems, n_ips)

The weak-form contraction `R = grad N . sigma w dv`, `dR/dU = grad N^T (dsigma/dgrad u) grad N` stays with
the caller (cmad.fem).  Transposes between the FE's AoS layout and the kernels' SoA layout are torch ops on
the device (plumbing); wiring this under `jax.pure_callback` needs a box with JAX and is not done here.
"""
from __future__ import annotations

import numpy as np

from ..models.device import NewtonSettings

_V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]


def local_update_with_tangent(model, grad_u, xi_prev, newton: NewtonSettings | None = None, grad_u_prev=None):
    """`newton` defaults to the FE binding's local settings (global_residual.py:292-297 / io/deck.py:71-82):
    20 iterations, 1e-12 tolerances, line search with 4 evaluations.
    `grad_u_prev` (same layout as grad_u) is required for `SmallRateElasticPlastic` blocks, whose local residual
    sees grad u - grad u_prev (`cm_update_rate_tangent`); d sigma / d grad u_prev is minus the returned tangent."""
    import torch
    newton = newton or NewtonSettings.traced(max_iters=20, abs_tol=1e-12, rel_tol=1e-12)
    ne, nip, nd, _ = grad_u.shape
    B = ne * nip
    nx = xi_prev.shape[-1]
    g = grad_u.reshape(B, nd * nd).t().contiguous()                 # (nd*nd, B) SoA
    xp = xi_prev.reshape(B, nx).t().contiguous()
    if getattr(model, "_model_kind", 0) == 1:
        if grad_u_prev is None:
            raise ValueError("the rate-form model needs grad_u_prev")
        gp = grad_u_prev.reshape(B, nd * nd).t().contiguous()
        xi, sig6, status, ds = model.device_evaluator(newton).update_rate(g, gp, xp, tangent=True)
        ds = ds.reshape(6, nd * nd, B)
    else:
        xi, sig6, status, ds = model.device_evaluator(newton).update(g, xp, tangent=True)
    xi_aos = xi.t().reshape(ne, nip, nx)
    sigma = torch.empty((B, 3, 3), dtype=torch.float64, device=g.device)
    dsig = torch.empty((B, 3, 3, nd * nd), dtype=torch.float64, device=g.device)
    for r, (i, j) in enumerate(_V6):
        sigma[:, i, j] = sig6[r]; sigma[:, j, i] = sig6[r]
        dsig[:, i, j, :] = ds[r].t(); dsig[:, j, i, :] = ds[r].t()
    return (xi_aos, sigma.reshape(ne, nip, 3, 3), dsig.reshape(ne, nip, 3, 3, nd, nd),
            status.reshape(ne, nip))
