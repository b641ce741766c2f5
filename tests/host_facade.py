"""TEST INFRASTRUCTURE: the cmad_amd facade classes with their three device entry points re-routed to the host
build of the same kernel arithmetic (tests/native).  Lets CPU CI exercise the Python-side logic of
Model / QoI / objectives (block slicing, parameter chain rules, the 13-term Hessian contraction) without a GPU.
Never imported by the product."""
import numpy as np

import host_harness_lib as hh
from cmad_amd import _lib
from cmad_amd.models import SmallElasticPlastic
from cmad_amd.models.deriv_types import DerivType
from cmad_amd.models.device import NewtonSettings


class HostHistoryEngine:
    """`cmad_amd.models.history_engine.HistoryEngine` with every launch replaced by the host build of the same per-point
    code (cm::primal_history_point, history_point, direct_history_point, hessian_weight)."""

    def __init__(self, model=None, newton=None, desc=None, info=None):
        self._desc, self.info = model._desc(newton=newton) if model is not None else (desc, info)

    def primal(self, gradu_hist, xi0):
        xi_hist, sig_hist, _ = hh.primal_history(self._desc, gradu_hist, xi0)
        return xi_hist, sig_hist

    def calibration(self, gradu_hist, data6_hist, wsq6, xi0):
        out, _ = hh.history(self._desc, gradu_hist, data6_hist, wsq6, xi0)
        return float(out[0]), out[1:]

    def adjoint(self, gradu_hist, sbar_hist, xi0, xibar_hist=None, want_lam=False):
        g, _, lam = hh.adjoint_history(self._desc, gradu_hist, sbar_hist, xi0, xibar_hist, want_lam=want_lam)
        return g, lam

    def direct(self, gradu_hist, xi_hist, sbar_hist, xibar_hist=None, want_blocks=False):
        g, dx, _ = hh.direct_history(self._desc, gradu_hist, xi_hist, sbar_hist, xibar_hist, want_blocks=want_blocks)
        return g, dx

    def hessian(self, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, sbar_hist, hss, hxx=None):
        return hh.hessian_history(self._desc, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, sbar_hist, hss, hxx)

    def direct_ep(self, ep_index, gradu_hist, xi_hist):
        return hh.direct_history_ep(self._desc, ep_index, gradu_hist, xi_hist)

    def hessian_ep(self, ep_index, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, dxi_dpe_hist, sbar_hist, hss, hxx=None):
        return hh.hessian_history_ep(self._desc, ep_index, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, dxi_dpe_hist, sbar_hist, hss, hxx)

    def extended(self, ep_index, gradu_hist, xi_hist, lam_hist, sbar_hist):
        K, nx, B = xi_hist.shape[0] - 1, xi_hist.shape[1], xi_hist.shape[2]
        g = np.zeros(len(ep_index))
        rate = self._desc.model_kind == 1
        for k in range(1, K + 1):
            dC, dS = hh.param_blocks(self._desc, ep_index, gradu_hist[k], xi_hist[k - 1], xi_hist[k], nx,
                                     gradu_prev=gradu_hist[k - 1] if rate else None)
            g += np.einsum("erb,rb->e", dS, sbar_hist[k]) - np.einsum("exb,xb->e", dC, lam_hist[k])
        return g


def _host_complex_solve(model, settings):
    """Model._complex_solve on the host build of cm::newton_cx."""
    p_im, ext_im, G, Gp, xp, x0 = model._complex_arrays()
    desc, _ = model._desc(newton=settings)
    xi, res, sig, st = hh.update_complex(desc, p_im, G, xp, x0, gradu_prev=Gp, ext_imag=ext_im)
    c = lambda a: a[0, :, 0] + 1j * a[1, :, 0]
    return c(xi), c(res), c(sig), int(st[0])


class HostSmallElasticPlastic(SmallElasticPlastic):
    def history_engine(self, newton=None):
        return HostHistoryEngine(self, newton)

    def _desc(self, params=None, newton=None):
        desc, info = super()._desc(params, newton)
        if "nn_packed" in info:                          # the host build reads the network weights from host memory (info keeps them alive)
            desc.nn_weights = info["nn_packed"].ctypes.data
        return desc, info

    def _point_evaluate(self, which, xi, xi_prev, params, U, want_jac=True, U_prev=None):
        desc, info = self._desc(params)
        nx = self.num_dofs
        G = np.asarray(U.grad_fields["u"], dtype=np.float64).reshape(-1, 1)
        C, J, s, S = hh.evaluate(desc, int(which), G, self._flat(xi_prev).reshape(-1, 1), self._flat(xi).reshape(-1, 1), nx)
        if which == DerivType.DNONE or not want_jac:
            return C[:, 0], None, s[:, 0], None, info
        return C[:, 0], J[:, :, 0], s[:, 0], S[:, :, 0], info

    def device_newton(self, max_iters=10, abs_tol=1e-14, rel_tol=1e-14, line_search=None):
        st = NewtonSettings(max_iters, abs_tol, rel_tol, line_search or {"max evals": 0}, warm_start=False)    # as Model.device_newton
        if self._is_complex:
            return self._complex_newton(st)
        desc, info = self._desc(newton=st)
        G = np.asarray(self._U.grad_fields["u"], dtype=np.float64).reshape(-1, 1)
        xi, sig, status = hh.update(desc, G, self._flat(self._xi_prev).reshape(-1, 1), self.num_dofs)
        self._xi = [b.astype(self.dtype) for b in self._split(xi[:, 0])]
        s = int(status[0])
        return s & _lib.STATUS_ITERS_MASK, bool(s & _lib.STATUS_CONVERGED)

    def _extended_blocks(self, ep_list, xi, xi_prev, params, U, U_prev):
        desc, info = self._desc(params)
        G = np.asarray(U.grad_fields["u"], dtype=np.float64).reshape(-1, 1)
        dC, dS = hh.param_blocks(desc, ep_list, G, self._flat(xi_prev).reshape(-1, 1), self._flat(xi).reshape(-1, 1), self.num_dofs)
        return dC[:, :, 0], dS[:, :, 0]

    def _complex_solve(self, settings):
        return _host_complex_solve(self, settings)

    def _second_derivative_pass(self):
        xi, xi_prev, params, U, U_prev = self.variables()
        desc, info = self._desc(params)
        nx = self.num_dofs
        G = np.asarray(U.grad_fields["u"], dtype=np.float64).reshape(-1, 1)
        d2C, d2S, dC, dS = hh.hessians(desc, G, self._flat(xi_prev).reshape(-1, 1), self._flat(xi).reshape(-1, 1), nx)
        return d2C[0], d2S[0], dC[0], dS[0], info, nx
