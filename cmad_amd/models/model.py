"""`Model`: material-point constitutive model contract -- host mirror of the reference's
``cmad.models.model.Model`` (/root/reference/cmad/models/model.py:25-563), same method names, argument
meaning and assertion behaviour.

Where the reference jit-compiles a residual and lets JAX produce Jacobians, every evaluation here is a
launch of the HIP library (include/cmad_hip.h): `evaluate()` / `evaluate_cauchy()` call `cm_evaluate` with
B = 1; the batched additions (`update_batch`, `update_tangent_batch`, `update_vjp_batch`,
`objective_grad_batch`) expose the data-parallel kernels directly.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
from abc import ABC
from typing import Any, ClassVar, Sequence

import numpy as np

from .. import _lib
from ..parameters.parameters import Parameters
from .deformation_types import DefType
from .deriv_types import DerivType
from .device import DeviceEvaluator, NewtonSettings, _ptr, build_desc, kp_to_leaf_grad, leaf_ep_index
from .global_fields import GlobalFieldsAtPoint
from .var_types import VarType

_V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]


def _sym3(v6):
    """6 stored entries (leading axis) -> symmetric 3x3 (two leading axes)."""
    v6 = np.asarray(v6)
    out = np.zeros((3, 3) + v6.shape[1:], dtype=v6.dtype)
    for r, (i, j) in enumerate(_V6):
        out[i, j] = v6[r]
        out[j, i] = v6[r]
    return out


class Model(ABC):
    """Material-point constitutive model (reference :25-88)."""

    supports_closed_form_cauchy: ClassVar[bool] = False
    supports_mixed: ClassVar[bool] = False

    # set by subclasses before Model.__init__ (reference :42-49)
    parameters: Parameters
    dtype: type
    _is_complex: bool
    _ndims: int
    _def_type: int
    _model_kind: int = 0
    _yield_tol: float = 1e-14
    _uniaxial_stress_idx: int = 0
    _hybrid = None                       # HybridHillEffectiveStress or None
    _hardening_nn = None                 # (input_scale, output_scale) of the network hardening law, or None

    @classmethod
    def from_deck(cls, model_section: dict, parameters: Parameters, def_type: int) -> "Model":
        raise NotImplementedError

    @classmethod
    def material_defaults(cls) -> dict:
        return {}

    def __init__(self) -> None:
        self._deriv_mode = DerivType.DNONE
        self.newton_settings = NewtonSettings()
        self.parameters.compute_mixed_block_shapes(self._num_eqs)
        self._Jac = None
        self._dSigma = None

    # ------------------------------------------------------------------ device plumbing
    def _desc(self, params=None, newton: NewtonSettings | None = None):
        if getattr(self, "_is_complex", False):             # complex-step instance: the description carries the real parts
            from .device import real_tree
            params = real_tree(self.parameters.values if params is None else params)
        return build_desc(self.parameters.values if params is None else params, def_type=self._def_type,
                          model_kind=self._model_kind, yield_tol=self._yield_tol,
                          uniaxial_stress_idx=self._uniaxial_stress_idx, newton=newton or self.newton_settings,
                          hybrid=self._hybrid, hardening_nn=self._hardening_nn)

    def device_evaluator(self, newton: NewtonSettings | None = None) -> DeviceEvaluator:
        """A `DeviceEvaluator` for the CURRENT parameter values (rebuilt per call: parameters change between
        objective evaluations, and the description is a few hundred bytes)."""
        return DeviceEvaluator(*self._desc(newton=newton))

    def history_engine(self, newton: NewtonSettings | None = None):
        """Whole-history launches (cm_update_history, cm_objective_grad_history, cm_adjoint_history, cm_direct_history,
        cm_hessian_history) for the CURRENT parameter values: what `cmad_amd.objectives` is built from."""
        from .history_engine import HistoryEngine
        return HistoryEngine(self.device_evaluator(newton))

    def init_state(self, B: int = 1) -> np.ndarray:
        """(n_xi, B) initial local state (reference `set_xi_to_init_vals`, :296-299)."""
        x0 = np.concatenate([np.atleast_1d(np.asarray(b, dtype=np.float64)) for b in self._init_xi])
        return np.repeat(x0[:, None], B, axis=1)

    def active_hessian_from_kp(self, H_kp, g_kp, info=None, ext=None):
        """Kernel-order Hessian + gradient (12,) -> Hessian w.r.t. the NATIVE active parameters: the chain rule through the
        elastic-constant map (lambda, mu)(E, nu, ...) adds sum_kp g_kp d2 kp / dp_i dp_j.  With `ext` = `extended_active()`
        the Hessian is (12 + n_ext)^2 (cm_hessian_history_ep: native parameters first, then the extended leaves in `ext`
        order); each extended leaf IS an active parameter (identity column)."""
        info = info or self._desc()[1]
        ext = ext or []
        T1, T2 = self._param_chain(info, skip={pos for pos, _ in ext})
        if ext:
            E = np.zeros((len(ext), T1.shape[1]))
            for i, (pos, _) in enumerate(ext):
                E[i, pos] = 1.0
            T1 = np.vstack([T1, E])
        return T1.T @ np.asarray(H_kp) @ T1 + np.einsum("k,kij->ij", np.asarray(g_kp), T2)

    @staticmethod
    def _flat(blocks) -> np.ndarray:
        return np.concatenate([np.atleast_1d(np.asarray(b, dtype=np.float64)).ravel() for b in blocks])

    def _point_evaluate(self, which, xi, xi_prev, params, U, want_jac=True, U_prev=None):
        """cm_evaluate (cm_evaluate_rate for the rate form) for one point -> (C, J, sigma6, S) as numpy."""
        import torch
        desc, info = self._desc(params)
        L = _lib.lib()
        nx, nu = L.cm_num_xi(C.byref(desc)), L.cm_num_gradu(C.byref(desc))
        if nx < 0:
            raise NotImplementedError("def_type not available in the HIP library")
        dev = torch.device("cuda")
        if "nn_packed" in info:
            nn_dev = torch.from_numpy(info["nn_packed"]).to(dev)
            desc.nn_weights = nn_dev.data_ptr()
        G = np.asarray(U.grad_fields["u"], dtype=np.float64).reshape(nu, 1)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 1)).to(dev)
        g, x1, x0 = t(G), t(self._flat(xi)), t(self._flat(xi_prev))
        ncols = {0: nx, 1: nx, 2: _lib.CM_NUM_PARAMS, 3: nu, 4: nu, 5: 1}[int(which)]
        Cd = torch.empty((nx, 1), dtype=torch.float64, device=dev)
        s = torch.empty((6, 1), dtype=torch.float64, device=dev)
        J = torch.empty((nx * ncols, 1), dtype=torch.float64, device=dev) if want_jac else None
        S = torch.empty((6 * ncols, 1), dtype=torch.float64, device=dev) if want_jac else None
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self._model_kind == 1:
            Gp = np.zeros_like(G) if U_prev is None else np.asarray(U_prev.grad_fields["u"], dtype=np.float64).reshape(nu, 1)
            gp = t(Gp)                                    # named: must outlive the launch
            rc = L.cm_evaluate_rate(C.byref(desc), 1, int(which), _ptr(g), _ptr(gp), _ptr(x0), _ptr(x1), _ptr(Cd),
                                    _ptr(J), _ptr(s), _ptr(S), stream)
            _lib.check(rc, "cm_evaluate_rate")
        else:
            rc = L.cm_evaluate(C.byref(desc), 1, int(which), _ptr(g), _ptr(x0), _ptr(x1), _ptr(Cd), _ptr(J), _ptr(s),
                               _ptr(S), stream)
            _lib.check(rc, "cm_evaluate")
        out_J = J.cpu().numpy().reshape(nx, ncols) if want_jac and which != DerivType.DNONE else None
        out_S = S.cpu().numpy().reshape(6, ncols) if want_jac and which != DerivType.DNONE else None
        return Cd.cpu().numpy()[:, 0], out_J, s.cpu().numpy()[:, 0], out_S, info

    def extended_active(self, info=None):
        """[(position among the active parameters, EP index)] of the active leaves that are differentiated by forward-mode
        evaluation of the whole model (`leaf_ep_index`): rotation matrix, Hosford exponent, Hill coefficients of the network
        surfaces, network weights."""
        info = info or self._desc()[1]
        out = []
        for pos, path in enumerate(self.parameters.active_paths()):
            e = leaf_ep_index(path, info)
            if e is not None:
                out.append((pos, e))
        return out

    def _extended_blocks(self, ep_list, xi, xi_prev, params, U, U_prev):
        """cm_param_blocks at the gathered state (one point): dC (n_ep, n_xi), dsigma6 (n_ep, 6)."""
        import torch
        ev = DeviceEvaluator(*self._desc(params))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 1)).cuda()
        G = np.asarray(U.grad_fields["u"], dtype=np.float64)
        gp = None
        if self._model_kind == 1:
            gp = t(np.zeros_like(G) if U_prev is None else np.asarray(U_prev.grad_fields["u"], dtype=np.float64))
        dC, dS = ev.param_blocks(ep_list, t(G), t(self._flat(xi_prev)), t(self._flat(xi)), gradu_prev=gp)
        return dC.cpu().numpy()[:, :, 0], dS.cpu().numpy()[:, :, 0]

    def _active_columns(self, M_kp, info, ext=None):
        """(rows, KP) kernel-order block -> (rows, num_active_params) in Parameters' flat active order
        (reference `_active_params_jacobian`, parameters.py:368-377).  `ext`: {active position: column} for the leaves
        served by `cm_param_blocks`."""
        ext = ext or {}
        cols = [ext[pos] if pos in ext else
                kp_to_leaf_grad(path[:-1] if isinstance(path[-1], (int, np.integer)) else path, M_kp.T, info)
                for pos, path in enumerate(self.parameters.active_paths())]
        return np.stack(cols, axis=1) if cols else np.zeros((M_kp.shape[0], 0))

    # ------------------------------------------------------------------ complex-step instances (reference: is_complex=True)
    def _complex_arrays(self):
        """(p_imag (12,), ext_imag or None, gradu (n, 1), gradu_prev or None, xi_prev (2, n_xi, 1), xi (2, n_xi, 1)) of the
        gathered state."""
        from .device import complex_parameter_parts
        _, info = self._desc()
        p_im, ext_im = complex_parameter_parts(self.parameters.values, self.parameters.flat_paths(), info)
        split = lambda blocks: np.stack([f(np.concatenate([np.atleast_1d(np.asarray(b, dtype=complex)).ravel() for b in blocks]))
                                         for f in (np.real, np.imag)])[:, :, None]
        G = np.asarray(self._U.grad_fields["u"], dtype=np.float64).reshape(-1, 1)
        Gp = np.asarray(self._U_prev.grad_fields["u"], dtype=np.float64).reshape(-1, 1) if self._model_kind == 1 else None
        return p_im, ext_im, G, Gp, np.ascontiguousarray(split(self._xi_prev)), np.ascontiguousarray(split(self._xi))

    def _complex_solve(self, settings: NewtonSettings):
        """One `cm_update_complex` launch at the gathered state, started at the current xi: (xi, C, sigma6, status word), complex."""
        import torch
        p_im, ext_im, G, Gp, xp, x0 = self._complex_arrays()
        ev = self.device_evaluator(settings)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
        xi, res, sig, status = ev.update_complex(p_im, t(G), t(xp), t(x0), gradu_prev=t(Gp), ext_imag=ext_im)
        c = lambda a: a.cpu().numpy()[0, :, 0] + 1j * a.cpu().numpy()[1, :, 0]
        return c(xi), c(res), c(sig), int(status.cpu().numpy().astype(np.uint32)[0])

    def _complex_values(self):
        """Residual and stress of a complex-step instance at its current state (values only: such an instance exists to be
        differentiated by the complex step itself)."""
        if self._deriv_mode != DerivType.DNONE:
            raise NotImplementedError("complex-step instances evaluate values (DerivType.DNONE); derivatives come from the real model")
        _, Cv, s6, _ = self._complex_solve(NewtonSettings(0, self.newton_settings.abs_tol, self.newton_settings.rel_tol, {"max evals": 0}))
        return Cv, s6

    # ------------------------------------------------------------------ reference :168-190
    def evaluate(self) -> None:
        """Evaluate the residual (C) or its jacobian (Jac)."""
        if getattr(self, "_is_complex", False):
            self._C, self._Jac = self._complex_values()[0], None
            return
        xi, xi_prev, params, U, U_prev = self.variables()
        mode = self._deriv_mode
        if mode == DerivType.DNONE:
            Cv, _, _, _, _ = self._point_evaluate(DerivType.DNONE, xi, xi_prev, params, U, want_jac=False, U_prev=U_prev)
            self._C = np.asarray(Cv, dtype=self.dtype)
            self._Jac = None
        elif mode == DerivType.DPARAMS:
            _, J, _, _, info = self._point_evaluate(DerivType.DPARAMS, xi, xi_prev, params, U, U_prev=U_prev)
            ext = self.extended_active(info)
            cols = {}
            if ext:
                dC, _ = self._extended_blocks([e for _, e in ext], xi, xi_prev, params, U, U_prev)
                cols = {pos: dC[i] for i, (pos, _) in enumerate(ext)}
            self._Jac = np.asarray(self._active_columns(J, info, cols), dtype=np.float64)
        elif mode == DerivType.DU_PREV and self._model_kind == 0:
            nu = self._ndims ** 2
            self._Jac = np.zeros((self.num_dofs, nu))          # the total-form residual ignores U_prev
        else:
            _, J, _, _, _ = self._point_evaluate(mode, xi, xi_prev, params, U, U_prev=U_prev)
            self._Jac = J

    # ------------------------------------------------------------------ reference :245-270
    def _param_chain(self, info, skip=()):
        """First- and second-order maps from the kernel's KP parameters to the active parameters:
        T1[kp, i] = d kp / d p_i ; T2[kp, i, j] = d2 kp / d p_i d p_j (non-zero only for lambda, mu)."""
        from .elastic_constants import lame_second_derivs
        from .device import HILL_NAMES
        paths = self.parameters.active_paths()
        P = len(paths)
        T1 = np.zeros((_lib.CM_NUM_PARAMS, P)); T2 = np.zeros((_lib.CM_NUM_PARAMS, P, P))
        names, H = lame_second_derivs(self.parameters.values["elastic"])
        el_pos = {}
        for i, path in enumerate(paths):
            if i in skip:                                    # an extended leaf: no native kernel parameter depends on it
                continue
            unit = np.zeros(_lib.CM_NUM_PARAMS)
            for kp in range(_lib.CM_NUM_PARAMS):
                unit[:] = 0.0; unit[kp] = 1.0
                T1[kp, i] = kp_to_leaf_grad(path[:-1] if isinstance(path[-1], int) else path, unit, info)
            if path[0] == "elastic":
                el_pos[i] = info["elastic_names"].index(path[-1])
        for i, ji in el_pos.items():
            for k, jk in el_pos.items():
                T2[_lib.P_LAMBDA, i, k] = H[0, ji, jk]
                T2[_lib.P_MU, i, k] = H[1, ji, jk]
        return T1, T2

    def _second_derivative_pass(self):
        """cm_hessians for the gathered state -> raw arrays w.r.t. q = [xi, xi_prev, p(KP)]."""
        import torch
        xi, xi_prev, params, U, U_prev = self.variables()
        desc, info = self._desc(params)
        if "nn_packed" in info:                                           # network surfaces: weights on the device
            nn_dev = torch.from_numpy(info["nn_packed"]).cuda()
            desc.nn_weights = nn_dev.data_ptr()
        L = _lib.lib()
        nx, nu = L.cm_num_xi(C.byref(desc)), L.cm_num_gradu(C.byref(desc))
        nq = 2 * nx + _lib.CM_NUM_PARAMS
        dev = torch.device("cuda")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 1)).to(dev)
        G = np.asarray(U.grad_fields["u"], dtype=np.float64).reshape(nu, 1)
        e = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        d2C, d2S, dC, dS = e(nx, nq, nq), e(6, nq, nq), e(nx, nq), e(6, nq)
        g, x0, x1 = t(G), t(self._flat(xi_prev)), t(self._flat(xi))       # keep the inputs alive across the launch
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self._model_kind == 1:                                         # rate form: the residual also takes grad u_prev
            gp = t(np.asarray(U_prev.grad_fields["u"], dtype=np.float64).reshape(nu, 1))
            rc = L.cm_hessians_rate(C.byref(desc), 1, _ptr(g), _ptr(gp), _ptr(x0), _ptr(x1),
                                    _ptr(d2C), _ptr(d2S), _ptr(dC), _ptr(dS), None, None, stream)
            _lib.check(rc, "cm_hessians_rate")
        else:
            rc = L.cm_hessians(C.byref(desc), 1, _ptr(g), _ptr(x0), _ptr(x1),
                               _ptr(d2C), _ptr(d2S), _ptr(dC), _ptr(dS), stream)
            _lib.check(rc, "cm_hessians")
        return d2C.cpu().numpy(), d2S.cpu().numpy(), dC.cpu().numpy(), dS.cpu().numpy(), info, nx

    def evaluate_hessians(self) -> None:
        """Evaluate the Hessians of the residual (reference :245-270): d2C_dxi2, d2C_dxi_dxi_prev, d2C_dxi_prev2
        (n, n_xi, n_xi), d2C_dxi_dparams, d2C_dxi_prev_dparams (n, n_xi, P), d2C_dparams2 (n, P, P)."""
        d2C, d2S, dC, dS, info, nx = self._second_derivative_pass()
        T1, T2 = self._param_chain(info)
        a, b, c = slice(0, nx), slice(nx, 2 * nx), slice(2 * nx, None)
        self.d2C_dxi2 = d2C[:, a, a]
        self.d2C_dxi_dxi_prev = d2C[:, a, b]
        self.d2C_dxi_prev2 = d2C[:, b, b]
        self.d2C_dxi_dparams = np.einsum("qkp,pi->qki", d2C[:, a, c], T1)
        self.d2C_dxi_prev_dparams = np.einsum("qkp,pi->qki", d2C[:, b, c], T1)
        self.d2C_dparams2 = np.einsum("qpr,pi,rj->qij", d2C[:, c, c], T1, T1) + np.einsum("qp,pij->qij", dC[:, c], T2)
        self._hessian_cache = (d2S, dS, T1, T2, nx)

    # ------------------------------------------------------------------ reference :273-293
    def evaluate_cauchy(self) -> None:
        """Evaluate the cauchy stress (Sigma) or its derivatives (dSigma)."""
        if getattr(self, "_is_complex", False):
            self._Sigma, self._dSigma = _sym3(self._complex_values()[1]), None
            return
        xi, xi_prev, params, U, U_prev = self.variables()
        mode = self._deriv_mode
        if mode == DerivType.DNONE:
            _, _, s6, _, _ = self._point_evaluate(DerivType.DNONE, xi, xi_prev, params, U, want_jac=False, U_prev=U_prev)
            self._Sigma = np.asarray(_sym3(s6), dtype=np.float64)
            self._dSigma = None
        elif mode == DerivType.DU_PREV:
            self._dSigma = np.zeros((3, 3, self._ndims ** 2))
        elif mode == DerivType.DPARAMS:
            _, _, _, S, info = self._point_evaluate(DerivType.DPARAMS, xi, xi_prev, params, U, U_prev=U_prev)
            S9 = _sym3(S).reshape(9, -1)                                   # (9, KP)
            ext = self.extended_active(info)
            cols = {}
            if ext:
                _, dS = self._extended_blocks([e for _, e in ext], xi, xi_prev, params, U, U_prev)
                cols = {pos: _sym3(dS[i]).reshape(9) for i, (pos, _) in enumerate(ext)}
            self._dSigma = np.asarray(self._active_columns(S9, info, cols), dtype=np.float64)
        else:
            _, _, _, S, _ = self._point_evaluate(mode, xi, xi_prev, params, U, U_prev=U_prev)
            self._dSigma = _sym3(S)                                         # (3, 3, n_xi) == np.dstack of blocks

    def set_xi_to_init_vals(self) -> None:
        for ii in range(self.num_residuals):
            self._xi[ii] = self._init_xi[ii].copy().astype(self.dtype)
            self._xi_prev[ii] = self._init_xi[ii].copy().astype(self.dtype)

    def C(self):
        return self._C

    def Jac(self):
        assert self._Jac is not None, "Jac() requires a non-DNONE deriv mode (seed_xi/xi_prev/params)"
        return self._Jac

    def Sigma(self):
        return self._Sigma

    def dSigma(self):
        assert self._dSigma is not None, "dSigma() requires a non-DNONE deriv mode (seed_xi/xi_prev/params)"
        return self._dSigma

    # ------------------------------------------------------------------ pure-function surface (:125-153, :316-374)
    def _split(self, flat):
        out, pos = [], 0
        for n in self._num_eqs:
            out.append(np.array(flat[pos:pos + n]))
            pos += n
        return out

    def _residual(self, xi, xi_prev, params, U, U_prev):
        return self._point_evaluate(DerivType.DNONE, xi, xi_prev, params, U, want_jac=False, U_prev=U_prev)[0]

    def cauchy(self, xi, xi_prev, params, U, U_prev):
        return _sym3(self._point_evaluate(DerivType.DNONE, xi, xi_prev, params, U, want_jac=False, U_prev=U_prev)[2])

    def _block_list(self, J):
        """(n_xi, n_xi) -> list over variable blocks of (n_xi, n_block) arrays, the pytree jacfwd returns."""
        out, pos = [], 0
        for n in self._num_eqs:
            out.append(J[:, pos:pos + n])
            pos += n
        return out

    def dC_dxi(self, xi, xi_prev, params, U, U_prev):
        return self._block_list(self._point_evaluate(DerivType.DXI, xi, xi_prev, params, U, U_prev=U_prev)[1])

    def dC_dxi_prev(self, xi, xi_prev, params, U, U_prev):
        return self._block_list(self._point_evaluate(DerivType.DXI_PREV, xi, xi_prev, params, U, U_prev=U_prev)[1])

    def dC_dp(self, xi, xi_prev, params, U, U_prev):
        """d C / d params as a tree parallel to `params` (reference model.py:316-374: jacrev w.r.t. the params
        pytree): leaf shape (n_xi,) + leaf.shape.  The 12 native kernel parameters come from `cm_evaluate`'s hand-derived
        block; rotation-matrix entries, the Hosford exponent, Barlat coefficients, Hill coefficients of the network surfaces
        and network weights from one `cm_param_blocks` launch (forward-mode evaluation of the model).  A leaf neither covers
        (none in the reference's models) comes back filled with NaN rather than with a silent zero."""
        _, J, _, _, info = self._point_evaluate(DerivType.DPARAMS, xi, xi_prev, params, U, U_prev=U_prev)
        n = self.num_dofs
        pending = []                                              # (output array, flat element, EP index)

        def walk(node, path):
            if isinstance(node, dict):
                return {k: walk(v, path + (k,)) for k, v in node.items()}
            if isinstance(node, (list, tuple)):
                return type(node)(walk(v, path + (i,)) for i, v in enumerate(node))
            shape = np.shape(node)
            try:
                col = np.asarray(kp_to_leaf_grad(path, J.T, info), dtype=np.float64)
                return np.broadcast_to(col.reshape((n,) + (1,) * len(shape)), (n,) + shape).copy()
            except (NotImplementedError, KeyError):
                pass
            out = np.full((n,) + shape, np.nan)
            size = int(np.prod(shape)) if shape else 1
            try:
                eps = [leaf_ep_index(path + ((e,) if shape else ()), info) for e in range(size)]
            except (NotImplementedError, KeyError, ValueError, IndexError):
                return out
            if all(e is not None for e in eps):
                pending.extend((out, e, ep) for e, ep in enumerate(eps))
            return out
        tree = walk(params, ())
        if pending:
            dC, _ = self._extended_blocks([ep for _, _, ep in pending], xi, xi_prev, params, U, U_prev)
            for i, (out, e, _) in enumerate(pending):
                out.reshape(n, -1)[:, e] = dC[i]
        return tree

    def dC_dU(self, xi, xi_prev, params, U, U_prev):
        J = self._point_evaluate(DerivType.DU, xi, xi_prev, params, U, U_prev=U_prev)[1]
        n = self._ndims
        return GlobalFieldsAtPoint(fields={"u": np.zeros((self.num_dofs, n))},
                                   grad_fields={"u": J.reshape(self.num_dofs, n, n)})

    def dC_dU_prev(self, xi, xi_prev, params, U, U_prev):
        n = self._ndims
        if self._model_kind == 1:
            J = self._point_evaluate(DerivType.DU_PREV, xi, xi_prev, params, U, U_prev=U_prev)[1]
            return GlobalFieldsAtPoint(fields={"u": np.zeros((self.num_dofs, n))},
                                       grad_fields={"u": J.reshape(self.num_dofs, n, n)})
        return GlobalFieldsAtPoint(fields={"u": np.zeros((self.num_dofs, n))},
                                   grad_fields={"u": np.zeros((self.num_dofs, n, n))})

    def variables(self):
        return (self._xi, self._xi_prev, self.parameters.values, self._U, self._U_prev)

    # ------------------------------------------------------------------ reference :383-408
    def _init_residuals(self, num_residuals: int) -> None:
        self.num_residuals = num_residuals
        self._num_eqs = np.zeros(num_residuals, dtype=int)
        self._var_types = np.zeros(num_residuals, dtype=int)
        self.resid_names = [None] * num_residuals
        self.var_names = [None] * num_residuals

    def _init_state_variables(self) -> None:
        n = self.num_residuals
        self._xi = [None] * n
        self._xi_prev = [None] * n
        self.num_dofs = 0
        self._delta_xi_offsets = np.zeros(n, dtype=int)
        for ii in range(n):
            self._delta_xi_offsets[ii] += self.num_dofs
            self.num_dofs += self._num_eqs[ii]

    def delta_xi_offset(self, res_idx: int, eq_idx: int) -> int:
        return self._delta_xi_offsets[res_idx] + eq_idx

    def var_type(self, residual: int) -> int:
        return self._var_types[residual]

    def resid_name(self, residual: int):
        return self.resid_names[residual]

    def state_output_fields(self):
        return [(self.var_names[r], VarType(int(self._var_types[r]))) for r in range(self.num_residuals)]

    def derived_output_field_names(self):
        return []

    @property
    def ndims(self) -> int:
        return self._ndims

    # ------------------------------------------------------------------ reference :451-525
    def gather_global(self, U: GlobalFieldsAtPoint, U_prev: GlobalFieldsAtPoint) -> None:
        self._U = U
        self._U_prev = U_prev

    def gather_xi(self, xi: Sequence, xi_prev: Sequence) -> None:
        self._xi = list(xi)
        self._xi_prev = list(xi_prev)

    def seed_xi(self) -> None:
        self._deriv_mode = DerivType.DXI

    def seed_xi_prev(self) -> None:
        self._deriv_mode = DerivType.DXI_PREV

    def seed_params(self) -> None:
        self._deriv_mode = DerivType.DPARAMS

    def seed_none(self) -> None:
        self._deriv_mode = DerivType.DNONE

    def deriv_mode(self) -> int:
        return self._deriv_mode

    def xi(self):
        return self._xi

    def xi_prev(self):
        return self._xi_prev

    def advance_xi(self) -> None:
        for ii in range(self.num_residuals):
            self._xi_prev[ii] = self._xi[ii].copy()

    def set_scalar_xi(self, idx: int, xi) -> None:
        self._xi[idx] = xi.copy()

    def set_vector_xi(self, idx: int, xi) -> None:
        self._xi[idx] = xi.copy()

    def set_sym_tensor_xi(self, idx: int, xi) -> None:
        self._set_sym(self._xi, idx, xi)

    def _set_sym(self, target, idx, t) -> None:
        n = self._num_eqs[idx]
        if n == 6:
            target[idx][:] = [t[0, 0], t[0, 1], t[0, 2], t[1, 1], t[1, 2], t[2, 2]]
        elif n == 3:
            target[idx][:] = [t[0, 0], t[0, 1], t[1, 1]]
        elif n == 1:
            target[idx][0] = t[0, 0]

    _NDIM_BY_NUM_EQS: ClassVar[dict] = {9: 3, 4: 2, 1: 1}

    @staticmethod
    def get_tensor_ndim(num_eqs: int) -> int:
        try:
            return Model._NDIM_BY_NUM_EQS[num_eqs]
        except KeyError as e:
            raise ValueError(f"Unknown num_eqs for tensor variable: {num_eqs}") from e

    def set_tensor_xi(self, idx: int, xi) -> None:
        n = Model.get_tensor_ndim(self._num_eqs[idx])
        self._xi[idx][:] = np.asarray(xi)[:n, :n].reshape(-1)

    def add_to_xi(self, delta_xi) -> None:
        for idx in range(self.num_residuals):
            if self._var_types[idx] != VarType.SCALAR:
                for eq in range(self._num_eqs[idx]):
                    self._xi[idx][eq] += delta_xi[self.delta_xi_offset(idx, eq)]
            else:
                self._xi[idx] = self._xi[idx] + delta_xi[self.delta_xi_offset(idx, 0)]

    def set_scalar_xi_prev(self, idx: int, xi_prev) -> None:
        self._xi_prev[idx] = xi_prev.copy()

    def set_vector_xi_prev(self, idx: int, xi_prev) -> None:
        self._xi_prev[idx] = xi_prev.copy()

    def set_sym_tensor_xi_prev(self, idx: int, xi_prev) -> None:
        self._set_sym(self._xi_prev, idx, xi_prev)

    def set_tensor_xi_prev(self, idx: int, xi_prev) -> None:
        n = Model.get_tensor_ndim(self._num_eqs[idx])
        self._xi_prev[idx][:] = np.asarray(xi_prev)[:n, :n].reshape(-1)

    @staticmethod
    def store_xi(xi_list, xi_val, step: int) -> None:
        for idx in range(len(xi_list[step])):
            xi_list[step][idx] = xi_val[idx].copy()

    # ------------------------------------------------------------------ local Newton on the device (B = 1)
    def device_newton(self, max_iters=10, abs_tol=1e-14, rel_tol=1e-14, line_search=None):
        """Solve the local problem for the gathered state with ONE `cm_update` launch; sets xi, returns
        (iters, converged).  Same algorithm as newton_solve / make_newton_solve (nonlinear_solver.py)."""
        import torch
        st = NewtonSettings(max_iters, abs_tol, rel_tol, line_search or {"max evals": 0}, warm_start=False)     # returns the reference's count
        if getattr(self, "_is_complex", False):
            return self._complex_newton(st)
        ev = self.device_evaluator(st)
        dev = torch.device("cuda")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 1)).to(dev)
        G = np.asarray(self._U.grad_fields["u"], dtype=np.float64)
        xi, _, status = ev.update(t(G), t(self._flat(self._xi_prev)), want_sigma=False)
        self._xi = [b.astype(self.dtype) for b in self._split(xi.cpu().numpy()[:, 0])]
        s = int(status.cpu().numpy().astype(np.uint32)[0])
        return s & _lib.STATUS_ITERS_MASK, bool(s & _lib.STATUS_CONVERGED)

    def _complex_newton(self, st: NewtonSettings):
        if st.line_search.get("max evals", 0) > 0:
            raise NotImplementedError("complex-step instances: plain Newton steps (the reference's tests use newton_solve defaults)")
        xi, _, _, s = self._complex_solve(st)
        self._xi = [np.asarray(b, dtype=complex) for b in self._split(xi)]
        return s & _lib.STATUS_ITERS_MASK, bool(s & _lib.STATUS_CONVERGED)

    # ------------------------------------------------------------------ batched API (new capability)
    def update_batch(self, gradu, xi_prev, newton: NewtonSettings | None = None, **kw):
        """(n_gradu, B), (n_xi, B) float64 CUDA tensors -> xi, sigma6, status."""
        return self.device_evaluator(newton).update(gradu, xi_prev, **kw)

    def update_tangent_batch(self, gradu, xi_prev, newton: NewtonSettings | None = None):
        return self.device_evaluator(newton).update(gradu, xi_prev, tangent=True)

    def active_grad_from_kp(self, g_kp, info=None, g_ext=None):
        """Kernel-order gradient (12,) -> Parameters' flat active order, NATIVE parameters (apply
        `parameters.transform_grad` afterwards for canonical ones, as the objectives do).  `g_ext`: {active position:
        value} for the leaves of `extended_active()` (from `cm_param_adjoint_history`)."""
        info = info or self._desc()[1]
        g = np.asarray(g_kp, dtype=np.float64)
        g_ext = g_ext or {}
        return np.array([g_ext[pos] if pos in g_ext else
                         kp_to_leaf_grad(path[:-1] if isinstance(path[-1], (int, np.integer)) else path, g, info)
                         for pos, path in enumerate(self.parameters.active_paths())])

    def update_vjp_batch(self, gradu, xi_prev, xi, sigma_bar, newton: NewtonSettings | None = None, **kw):
        ev = self.device_evaluator(newton)
        g, xb, ub = ev.update_vjp(gradu, xi_prev, xi, sigma_bar, **kw)
        return self.active_grad_from_kp(g.cpu().numpy(), ev.info), xb, ub
