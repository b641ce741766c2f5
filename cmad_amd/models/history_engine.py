"""Whole-history launches for one model: what the material-point objectives are built from.

Every method is one entry point of include/cmad_hip.h on the current torch CUDA stream (numpy in, numpy out; the point
axis is last, B = 1 for the reference's material-point drivers, any B for batches):

    primal       cm_update_history            K stress updates per point, states and stresses of every step
    calibration  cm_objective_grad_history    forward pass + adjoint recursion with the Calibration QoI fused
    adjoint      cm_adjoint_history           the adjoint recursion for any QoI (the caller supplies dJ/dsigma, dJ/dxi)
    direct       cm_direct_history            the forward-sensitivity recursion + gradient contraction
    hessian      cm_hessian_history           the second-order (direct-adjoint) quadratic form
    direct_ep, hessian_ep  cm_direct_history_ep, cm_hessian_history_ep  the same with leaves outside the 12 native parameters
    extended     cm_param_adjoint_history     gradient share of the leaves differentiated by forward-mode evaluation

There is no CPU implementation behind it.  tests/host_facade.py substitutes a host build of the same kernel arithmetic
for CPU CI of the Python logic above it.
"""
from __future__ import annotations

import numpy as np


class HistoryEngine:
    def __init__(self, evaluator):
        """`evaluator`: a `cmad_amd.models.device.DeviceEvaluator` (model description at the current parameter values)."""
        self._ev = evaluator
        self.info = self._ev.info
        self.nx, self.nu = self._ev.nx, self._ev.nu
        self._cache = {}

    def _dev(self, a, key=None):
        """numpy -> contiguous float64 device tensor; arrays passed with a key are uploaded once per engine."""
        import torch
        if key is not None and key in self._cache and self._cache[key][0] is a:
            return self._cache[key][1]
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
        if key is not None:
            self._cache[key] = (a, t)
        return t

    def primal(self, gradu_hist, xi0):
        xi_hist, sig_hist, _ = self._ev.update_history(self._dev(gradu_hist, "g"), self._dev(xi0), want_status=False)
        return xi_hist.cpu().numpy(), sig_hist.cpu().numpy()

    def calibration(self, gradu_hist, data6_hist, wsq6, xi0):
        K = gradu_hist.shape[0] - 1
        g, d, x0 = self._dev(gradu_hist, "g"), self._dev(data6_hist, "d"), self._dev(xi0)
        if K == 1:                                   # single step: the fused objective kernel
            kw = {"gradu_prev": g[0]} if self._ev.desc.model_kind == 1 else {}
            out, _ = self._ev.objective_grad(g[1], x0, d[1], wsq6, **kw)
        else:
            out, _ = self._ev.objective_grad_history(g, d, wsq6, x0)
        out = out.cpu().numpy()
        return float(out[0]), out[1:]

    def adjoint(self, gradu_hist, sbar_hist, xi0, xibar_hist=None, want_lam=False):
        g, _, lam = self._ev.adjoint_history(self._dev(gradu_hist, "g"), self._dev(sbar_hist), self._dev(xi0),
                                             None if xibar_hist is None else self._dev(xibar_hist), want_lam=want_lam)
        return g.cpu().numpy(), (lam.cpu().numpy() if want_lam else None)

    def direct(self, gradu_hist, xi_hist, sbar_hist, xibar_hist=None, want_blocks=False):
        g, dx, _ = self._ev.direct_history(self._dev(gradu_hist, "g"), self._dev(xi_hist), self._dev(sbar_hist),
                                           None if xibar_hist is None else self._dev(xibar_hist), want_blocks=want_blocks)
        return g.cpu().numpy(), (dx.cpu().numpy() if want_blocks else None)

    def hessian(self, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, sbar_hist, hss, hxx=None):
        H = self._ev.hessian_history(self._dev(gradu_hist, "g"), self._dev(xi_hist), self._dev(lam_hist), self._dev(dxi_dp_hist),
                                     self._dev(sbar_hist), hss, hxx)
        return H.cpu().numpy()

    def direct_ep(self, ep_index, gradu_hist, xi_hist):
        """cm_direct_history_ep: (K+1, n_xi, n_ep, B) forward sensitivities of the extended parameters."""
        return self._ev.direct_history_ep(ep_index, self._dev(gradu_hist, "g"), self._dev(xi_hist)).cpu().numpy()

    def hessian_ep(self, ep_index, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, dxi_dpe_hist, sbar_hist, hss, hxx=None):
        """cm_hessian_history_ep: (12 + n_ep, 12 + n_ep) Hessian, native parameters first."""
        H = self._ev.hessian_history_ep(ep_index, self._dev(gradu_hist, "g"), self._dev(xi_hist), self._dev(lam_hist),
                                        self._dev(dxi_dp_hist), self._dev(dxi_dpe_hist), self._dev(sbar_hist), hss, hxx)
        return H.cpu().numpy()

    def extended(self, ep_index, gradu_hist, xi_hist, lam_hist, sbar_hist):
        g = self._ev.param_adjoint_history(ep_index, self._dev(gradu_hist, "g"), self._dev(xi_hist), self._dev(lam_hist),
                                           self._dev(sbar_hist))
        return g.cpu().numpy()
