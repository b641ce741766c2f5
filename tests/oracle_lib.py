"""ctypes front-end of the CPU oracle (oracle/libcmad_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py.  Never imported by anything under cmad_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libcmad_oracle.so")

FULL_3D, PLANE_STRAIN, PLANE_STRESS, UNIAXIAL_STRESS = 0, 1, 2, 3
SMALL_EP, SMALL_RATE_EP = 0, 1
Y_J2, Y_HILL, Y_HOSFORD, Y_HYBRID, Y_SCALED_HYBRID, Y_BARLAT = 0, 1, 2, 3, 4, 5
BARLAT_NAMES = tuple(f"{pre}_{ij}" for pre in ("sp", "dp") for ij in ("12", "13", "21", "23", "31", "32", "44", "55", "66")) + ("a",)
LS_NONE, LS_TRACED, LS_LEGACY = 0, 1, 2
W_XI, W_XI_PREV, W_PARAMS, W_U, W_U_PREV = 0, 1, 2, 3, 4

# elastic pair codes; keys are the two names in the order (E, nu, mu, kappa, lambda)
ELASTIC_PAIRS = {
    ("E", "nu"): 0, ("mu", "lambda"): 1, ("mu", "kappa"): 2, ("E", "mu"): 3, ("E", "kappa"): 4,
    ("nu", "mu"): 5, ("nu", "kappa"): 6, ("nu", "lambda"): 7, ("kappa", "lambda"): 8, ("E", "lambda"): 9,
}
_CONST_ORDER = ("E", "nu", "mu", "kappa", "lambda")

P_Q, P_EL0, P_EL1, P_Y, P_VOCE_S, P_VOCE_D, P_LIN_K, P_YC, NP = 0, 9, 10, 11, 12, 13, 14, 15, 21
HILL_NAMES = ("F", "G", "H", "L", "M", "N")


class Desc(C.Structure):
    _fields_ = [("model_kind", C.c_int), ("def_type", C.c_int), ("yield_kind", C.c_int),
                ("elastic_pair", C.c_int), ("has_voce", C.c_int), ("has_linear", C.c_int),
                ("uniaxial_idx", C.c_int), ("hardening_order", C.c_int), ("yield_tol", C.c_double),
                ("nn_nlayers", C.c_int), ("nn_widths", C.c_int * 8), ("nn_w", C.POINTER(C.c_double)),
                ("beta_equivalent_stress", C.c_double), ("beta_max_iters", C.c_int),
                ("beta_abs_tol", C.c_double), ("beta_rel_tol", C.c_double), ("barlat", C.c_double * 19),
                ("hnn_width", C.c_int), ("hnn_w", C.POINTER(C.c_double)),
                ("hnn_nhidden", C.c_int), ("hnn_widths", C.c_int * 4)]


class Newton(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("abs_tol", C.c_double), ("rel_tol", C.c_double),
                ("ls_kind", C.c_int), ("ls_max_evals", C.c_int), ("ls_c1", C.c_double),
                ("ls_lo", C.c_double), ("ls_hi", C.c_double)]


def newton_settings(max_iters=10, abs_tol=1e-14, rel_tol=1e-14, ls_kind=LS_NONE, ls_max_evals=0,
                    c1=1e-4, lo=0.5, hi=0.9):
    return Newton(max_iters, abs_tol, rel_tol, ls_kind, ls_max_evals, c1, lo, hi)


_PORT_SO = os.path.join(_ORACLE_DIR, "libcmad_port.so")


def build(force=False):
    """Compile the oracle (and the CPU port of the kernel arithmetic that bench.py times beside it) if needed
    (g++, a few seconds)."""
    src = [os.path.join(_ORACLE_DIR, f) for f in ("cmad_oracle.cpp", "dual.hpp")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    csrc = os.path.join(_ROOT, "cmad_amd", "csrc")
    psrc = [os.path.join(_ORACLE_DIR, "cmad_port.cpp"), os.path.join(_ROOT, "include", "cmad_hip.h")] + \
           [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hpp")]
    pstale = (not os.path.exists(_PORT_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_PORT_SO) for s in psrc)
    if force:
        subprocess.run(["make", "-C", _ORACLE_DIR, "clean"], check=True, capture_output=True)
    if force or stale or pstale:
        subprocess.run(["make", "-C", _ORACLE_DIR], check=True, capture_output=True)
    return _SO


_port = None


def port_update_and_vjp(desc, gradu, xi_prev, sbar6, general=False, nthreads=1):
    """oracle/cmad_port.cpp: the kernels' own per-point arithmetic (host build of cmad_amd/csrc) in an OpenMP loop -- update +
    vjp of `cm_update_and_vjp` for FULL_3D, Q = I, plain Newton.  `desc` is a cmad_amd._lib.ModelDesc.
    Returns (xi (7, B), sigma (6, B), grad KP (12,))."""
    global _port
    if _port is None:
        build()
        _port = C.CDLL(_PORT_SO)
    from cmad_amd import _lib as cl
    assert _port.port_sizeof_desc() == C.sizeof(cl.ModelDesc)
    gradu, xi_prev, sbar6 = f64(gradu), f64(xi_prev), f64(sbar6)
    B = gradu.shape[1]
    xi = np.empty((7, B)); sig = np.empty((6, B)); g = np.zeros(12)
    rc = _port.port_update_and_vjp(C.byref(desc), C.c_int64(B), _p(gradu), _p(xi_prev), _p(sbar6), _p(xi), _p(sig), _p(g),
                                   C.c_int(int(general)), C.c_int(int(nthreads)))
    if rc != 0:
        raise NotImplementedError("cmad_port: FULL_3D, Q = I, no line search, J2 / Hill / Hosford only")
    return xi, sig, g


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.orc_nxi.argtypes = [C.POINTER(Desc)]; L.orc_nxi.restype = C.c_int
        L.orc_nu.argtypes = [C.POINTER(Desc)]; L.orc_nu.restype = C.c_int
        L.orc_residual.argtypes = [C.POINTER(Desc)] + [dp] * 6
        L.orc_jacobian.argtypes = [C.POINTER(Desc), C.c_int] + [dp] * 6
        L.orc_cauchy.argtypes = [C.POINTER(Desc)] + [dp] * 4
        L.orc_dcauchy.argtypes = [C.POINTER(Desc), C.c_int] + [dp] * 6
        L.orc_quad_min.argtypes = [C.c_double] * 4
        L.orc_quad_min.restype = C.c_double
        L.orc_line_search_linear.argtypes = [C.c_double] * 5 + [C.c_int] + [C.c_double] * 4 + [dp]
        L.orc_line_search_linear.restype = C.c_double
        L.orc_yield.argtypes = [C.POINTER(Desc)] + [dp] * 6
        L.orc_newton.argtypes = [C.POINTER(Desc), C.POINTER(Newton)] + [dp] * 6 + [C.POINTER(C.c_int)]
        L.orc_newton.restype = C.c_int
        L.orc_solve.argtypes = [C.c_int, dp, dp, C.c_int]
        L.orc_second_derivs.argtypes = [C.POINTER(Desc), dp, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp]
        L.orc_update_batch.argtypes = [C.POINTER(Desc), C.POINTER(Newton), dp, C.c_int64,
                                       dp, dp, dp, dp, dp, ip, ip, C.c_int]
        L.orc_tangent_batch.argtypes = [C.POINTER(Desc), dp, C.c_int64, dp, dp, dp, dp, dp, dp, C.c_int]
        L.orc_objective_grad_batch.argtypes = [C.POINTER(Desc), C.POINTER(Newton), dp, C.c_int64, C.c_int,
                                               dp, dp, dp, dp, dp, dp, dp, dp, C.c_int]
        L.orc_update_vjp_batch.argtypes = [C.POINTER(Desc), dp, C.c_int64, dp, dp, dp, dp, dp, dp, dp, dp, C.c_int]
        _lib = L
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def quad_min(phi0, dphi0, a, phi):
    return lib().orc_quad_min(phi0, dphi0, a, phi)


def line_search_linear(r0, r1, phi0, dphi0, init_aux=1.0, nan_above=np.inf, max_evals=4, c1=1e-4, lo=0.5, hi=0.9):
    """The oracle's line search (the one its Newton uses) on the merit 1/2 (r0 + r1 alpha)^2 -> (alpha, aux)."""
    aux = C.c_double()
    alpha = lib().orc_line_search_linear(r0, r1, nan_above, phi0, dphi0, max_evals, c1, lo, hi, init_aux, C.byref(aux))
    return alpha, aux.value


class Material:
    """Oracle-side material description: Desc + flat parameter vector p[NP].

    ``values`` is the CMAD nested parameter dict (the same tree cmad.parameters
    holds), so tests can build it exactly like tests/support/test_problems.py.
    """

    def __init__(self, values, def_type=FULL_3D, model_kind=SMALL_EP, yield_tol=1e-14, uniaxial_idx=0,
                 nn=None, scaled=None, hardening_nn=None):
        """nn = (layer_widths, packed weights) selects the hybrid Hill + ICNN surface; scaled = (equivalent_stress,
        max_iters, abs_tol, rel_tol) wraps it in `scaled_effective_stress` (effective_stress.py:97-146);
        hardening_nn = (H, packed [W1[H], b1[H], W2[H], b2, in_scale, out_scale]) adds the network hardening law
        (simple_neural_network.py:13-46 as hardening_funs["neural network"]); ([H1, ..., Hn], packed [W_l, b_l for every layer,
        in_scale, out_scale]) the same with several hidden layers."""
        self.values = values
        p = np.zeros(NP)
        p[P_Q:P_Q + 9] = np.asarray(values.get("rotation matrix", np.eye(3)), dtype=float).reshape(9)
        el = values["elastic"]
        given = tuple(n for n in _CONST_ORDER if n in el)
        if len(given) != 2:
            raise ValueError(f"need exactly two elastic constants, got {given}")
        p[P_EL0], p[P_EL1] = float(el[given[0]]), float(el[given[1]])
        pl = values["plastic"]
        ykey = next(iter(pl["effective stress"]))
        if nn is not None:
            yk = Y_SCALED_HYBRID if scaled is not None else Y_HYBRID
        else:
            yk = {"J2": Y_J2, "hill": Y_HILL, "hosford": Y_HOSFORD, "barlat": Y_BARLAT}[ykey]
        if yk in (Y_HILL, Y_HYBRID, Y_SCALED_HYBRID):
            h = pl["effective stress"]["hill"]
            p[P_YC:P_YC + 6] = [float(h[k]) for k in HILL_NAMES]
        elif yk == Y_HOSFORD:
            p[P_YC] = float(pl["effective stress"]["hosford"]["a"])
        fs = pl["flow stress"]
        p[P_Y] = float(fs["initial yield"]["Y"])
        hard = fs.get("hardening", {})
        keys = list(hard.keys())
        has_voce, has_lin = int("voce" in hard), int("linear" in hard)
        if has_voce:
            p[P_VOCE_S], p[P_VOCE_D] = float(hard["voce"]["S"]), float(hard["voce"]["D"])
        if has_lin:
            p[P_LIN_K] = float(hard["linear"]["K"])
        order = 1 if (has_voce and has_lin and keys.index("linear") < keys.index("voce")) else 0
        self.p = p
        self.elastic_names = given
        self.desc = Desc(model_kind, def_type, yk, ELASTIC_PAIRS[given], has_voce, has_lin, uniaxial_idx, order,
                         yield_tol, 0, (C.c_int * 8)(), None)
        self._hnn_keep = None
        if hardening_nn is not None:
            Hn, hp = hardening_nn                       # H (widths [1, H, 1]) or the list of hidden widths [H1, ..., Hn], n >= 2
            self._hnn_keep = f64(hp)
            if isinstance(Hn, (list, tuple)):
                assert 2 <= len(Hn) <= 4
                self.desc.hnn_width, self.desc.hnn_nhidden = int(Hn[0]), len(Hn)
                for i, wdt in enumerate(Hn):
                    self.desc.hnn_widths[i] = int(wdt)
            else:
                self.desc.hnn_width = int(Hn)
            self.desc.hnn_w = _p(self._hnn_keep)
        self._nn_keep = None
        if nn is not None:
            widths, packed = nn
            packed = f64(packed)
            self._nn_keep = packed
            self.desc.nn_nlayers = len(widths)
            for i, w in enumerate(widths):
                self.desc.nn_widths[i] = int(w)
            self.desc.nn_w = _p(packed)
        if yk == Y_BARLAT:
            bc = pl["effective stress"]["barlat"]
            for i, name in enumerate(BARLAT_NAMES):
                self.desc.barlat[i] = float(bc[name])
        if scaled is not None:
            self.desc.beta_equivalent_stress = float(scaled[0])
            self.desc.beta_max_iters = int(scaled[1])
            self.desc.beta_abs_tol, self.desc.beta_rel_tol = float(scaled[2]), float(scaled[3])
        self.nx = lib().orc_nxi(C.byref(self.desc))
        self.nu = lib().orc_nu(C.byref(self.desc))

    # ---- leaf name -> oracle flat index (for mapping gradients) ----
    def param_index(self, path):
        """path: tuple of dict keys, e.g. ("plastic","flow stress","initial yield","Y")."""
        leaf = path[-1]
        if path[0] == "elastic":
            return P_EL0 + self.elastic_names.index(leaf)
        if leaf == "Y":
            return P_Y
        if path[-2] == "voce":
            return {"S": P_VOCE_S, "D": P_VOCE_D}[leaf]
        if path[-2] == "linear":
            return P_LIN_K
        if path[-2] == "hill":
            return P_YC + HILL_NAMES.index(leaf)
        if path[-2] == "hosford":
            return P_YC
        raise KeyError(path)

    # ---- single-point calls ----
    def _uu(self, U, Up):
        U = f64(U).reshape(-1)
        Up = np.zeros_like(U) if Up is None else f64(Up).reshape(-1)
        assert U.size == self.nu
        return U, Up

    def residual(self, xi, xi_prev, U, Up=None):
        U, Up = self._uu(U, Up)
        out = np.zeros(self.nx)
        lib().orc_residual(C.byref(self.desc), _p(f64(xi)), _p(f64(xi_prev)), _p(self.p), _p(U), _p(Up), _p(out))
        return out

    def ncols(self, which):
        return self.nx if which in (W_XI, W_XI_PREV) else (NP if which == W_PARAMS else self.nu)

    def jacobian(self, which, xi, xi_prev, U, Up=None):
        U, Up = self._uu(U, Up)
        out = np.zeros((self.nx, self.ncols(which)))
        lib().orc_jacobian(C.byref(self.desc), which, _p(f64(xi)), _p(f64(xi_prev)), _p(self.p), _p(U), _p(Up), _p(out))
        return out

    def cauchy(self, xi, U):
        U, _ = self._uu(U, None)
        out = np.zeros(9)
        lib().orc_cauchy(C.byref(self.desc), _p(f64(xi)), _p(self.p), _p(U), _p(out))
        return out.reshape(3, 3)

    def dcauchy(self, which, xi, xi_prev, U, Up=None):
        U, Up = self._uu(U, Up)
        out = np.zeros((9, self.ncols(which)))
        lib().orc_dcauchy(C.byref(self.desc), which, _p(f64(xi)), _p(f64(xi_prev)), _p(self.p), _p(U), _p(Up), _p(out))
        return out

    def second_derivs(self, xi, xi_prev, U, Up=None):
        """d2C (nx, nq, nq) and d2S (9, nq, nq) w.r.t. q = [xi, xi_prev, p (oracle order, NP)]."""
        U, Up = self._uu(U, Up)
        nq = 2 * self.nx + NP
        d2C = np.zeros((self.nx, nq, nq)); d2S = np.zeros((9, nq, nq))
        c, s9 = np.zeros(self.nx), np.zeros(9)
        xi, xi_prev = f64(xi), f64(xi_prev)
        for a in range(nq):
            for b in range(a, nq):
                lib().orc_second_derivs(C.byref(self.desc), _p(xi), _p(xi_prev), _p(self.p), _p(U), _p(Up), a, b, _p(c), _p(s9))
                d2C[:, a, b] = c; d2C[:, b, a] = c
                d2S[:, a, b] = s9; d2S[:, b, a] = s9
        return d2C, d2S

    def yield_state(self, xi, U):
        U, _ = self._uu(U, None)
        phi, f, n = C.c_double(), C.c_double(), np.zeros(9)
        lib().orc_yield(C.byref(self.desc), _p(f64(xi)), _p(self.p), _p(U), C.byref(phi), C.byref(f), _p(n))
        return phi.value, f.value, n.reshape(3, 3)

    def newton(self, settings, xi_prev, U, Up=None):
        U, Up = self._uu(U, Up)
        x = np.zeros(self.nx)
        cn, cv = C.c_double(), C.c_int()
        it = lib().orc_newton(C.byref(self.desc), C.byref(settings), _p(f64(xi_prev)), _p(self.p), _p(U), _p(Up),
                              _p(x), C.byref(cn), C.byref(cv))
        return x, it, cn.value, bool(cv.value)

    def init_xi(self):
        x = np.zeros(self.nx)
        if self.desc.def_type == PLANE_STRESS:
            x[7] = 1.0
        elif self.desc.def_type == UNIAXIAL_STRESS:
            x[7] = x[8] = 1.0
        return x

    # ---- batched SoA calls ----
    def update_batch(self, settings, gradu, xi_prev, gradu_prev=None, nthreads=0, want_sigma=True):
        gradu, xi_prev = f64(gradu), f64(xi_prev)
        B = gradu.shape[1]
        assert gradu.shape == (self.nu, B) and xi_prev.shape == (self.nx, B)
        xi = np.zeros((self.nx, B)); sig = np.zeros((6, B)) if want_sigma else None
        iters = np.zeros(B, dtype=np.int32); conv = np.zeros(B, dtype=np.int32)
        gp = None if gradu_prev is None else f64(gradu_prev)
        lib().orc_update_batch(C.byref(self.desc), C.byref(settings), _p(self.p), B, _p(gradu), _p(gp), _p(xi_prev),
                               _p(xi), _p(sig), _pi(iters), _pi(conv), nthreads)
        return xi, sig, iters, conv

    def tangent_batch(self, gradu, xi_prev, xi, gradu_prev=None, nthreads=0):
        gradu, xi_prev, xi = f64(gradu), f64(xi_prev), f64(xi)
        B = gradu.shape[1]
        ds = np.zeros((6 * self.nu, B)); dx = np.zeros((self.nx * self.nu, B))
        gp = None if gradu_prev is None else f64(gradu_prev)
        lib().orc_tangent_batch(C.byref(self.desc), _p(self.p), B, _p(gradu), _p(gp), _p(xi_prev), _p(xi),
                                _p(ds), _p(dx), nthreads)
        return ds.reshape(6, self.nu, B), dx.reshape(self.nx, self.nu, B)

    def objective_grad_batch(self, settings, gradu_hist, data_hist, w, xi0, nthreads=0):
        """gradu_hist (K+1, nu, B), data_hist (K+1, 9, B), w (3,3), xi0 (nx, B) -> J, grad[NP], Jb[B], xiK."""
        gradu_hist, data_hist, xi0 = f64(gradu_hist), f64(data_hist), f64(xi0)
        K = gradu_hist.shape[0] - 1
        B = gradu_hist.shape[2]
        w = f64(w).reshape(9)
        J = C.c_double(); g = np.zeros(NP); Jb = np.zeros(B); xk = np.zeros((self.nx, B))
        lib().orc_objective_grad_batch(C.byref(self.desc), C.byref(settings), _p(self.p), B, K, _p(gradu_hist),
                                       _p(data_hist), _p(w), _p(xi0), C.cast(C.byref(J), C.POINTER(C.c_double)),
                                       _p(g), _p(Jb), _p(xk), nthreads)
        return J.value, g, Jb, xk

    def update_vjp_batch(self, gradu, xi_prev, xi, sbar6, gradu_prev=None, nthreads=0, want_bars=True):
        gradu, xi_prev, xi, sbar6 = f64(gradu), f64(xi_prev), f64(xi), f64(sbar6)
        B = gradu.shape[1]
        g = np.zeros(NP)
        xb = np.zeros((self.nx, B)) if want_bars else None
        ub = np.zeros((self.nu, B)) if want_bars else None
        gp = None if gradu_prev is None else f64(gradu_prev)
        lib().orc_update_vjp_batch(C.byref(self.desc), _p(self.p), B, _p(gradu), _p(gp), _p(xi_prev), _p(xi),
                                   _p(sbar6), _p(g), _p(xb), _p(ub), nthreads)
        return g, xb, ub


def j2_voce_values(E=200e3, nu=0.3, Y=200., S=200., D=20., Q=None, yield_kind="J2", hill=None, a=None, barlat=None):
    """Parameter tree of tests/support/test_problems.py:9-40 (J2 / J2-equivalent Hill / Hosford)."""
    if yield_kind == "J2":
        eff = {"J2": 0.}
    elif yield_kind == "hill":
        h = 0.5 * np.ones(6) if hill is None else hill
        eff = {"hill": dict(zip(HILL_NAMES, [float(x) for x in h]))}
    elif yield_kind == "hosford":
        eff = {"hosford": {"a": 4. if a is None else float(a)}}
    elif yield_kind == "barlat":
        coeffs = np.r_[np.ones(18), 4. if a is None else float(a)] if barlat is None else np.asarray(barlat, dtype=float)
        eff = {"barlat": dict(zip(BARLAT_NAMES, [float(x) for x in coeffs]))}
    else:
        raise ValueError(yield_kind)
    return {
        "rotation matrix": np.eye(3) if Q is None else np.asarray(Q, dtype=float),
        "elastic": {"E": E, "nu": nu},
        "plastic": {"effective stress": eff,
                    "flow stress": {"initial yield": {"Y": Y}, "hardening": {"voce": {"S": S, "D": D}}}}}
