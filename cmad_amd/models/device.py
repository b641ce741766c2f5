"""Host side of the HIP path: builds `cm_model_desc` from a CMAD parameter tree and launches the
C-ABI entry points (include/cmad_hip.h) on torch device tensors.  torch is only the carrier of device
memory and streams; all arithmetic happens in cmad_amd/csrc.

There is deliberately no CPU implementation behind these calls: without the HIP library (or without a
GPU) they raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from .. import _lib
from .deformation_types import DefType
from .elastic_constants import lame_jacobian

YIELD_KINDS = {"J2": 0, "hill": 1, "hosford": 2, "barlat": 5}
HILL_NAMES = ("F", "G", "H", "L", "M", "N")
# flatten_barlat_params order, cmad/models/effective_stress.py:55-78
BARLAT_NAMES = tuple(f"{pre}_{ij}" for pre in ("sp", "dp")
                     for ij in ("12", "13", "21", "23", "31", "32", "44", "55", "66")) + ("a",)

# cmad/util/line_search.py:40-46
DEFAULT_LINE_SEARCH_SETTINGS = {
    "max evals": 4,
    "sufficient decrease": 1.0e-4,
    "min backtrack factor": 0.5,
    "max backtrack factor": 0.9,
}


@dataclass
class NewtonSettings:
    """Local Newton controls.  Defaults = imperative `newton_solve` (cmad/models/nonlinear_solver.py:14-20):
    10 iterations, 1e-14 tolerances, no line search.  `traced()` gives `make_newton_solve`'s defaults
    (:88-100) with the quadratic Armijo search of cmad/util/line_search.py."""
    max_iters: int = 10
    abs_tol: float = 1e-14
    rel_tol: float = 1e-14
    line_search: dict = field(default_factory=lambda: {**DEFAULT_LINE_SEARCH_SETTINGS, "max evals": 0})
    # J2: the kernels run the Newton iteration restricted to the subspace it never leaves -- the radial line under FULL_3D,
    # the plane span{dev(eps - eps_p_prev), dev z} x (alpha, F33) under PLANE_STRESS (same iterates and
    # counts).  False forces the general 7 / 8-dof iteration (include/cmad_hip.h CM_SOLVER_GENERAL_NEWTON).  Ignored for
    # every other configuration.
    j2_radial_line: bool = True
    # Warm starts (include/cmad_hip.h CM_SOLVER_REFERENCE_ITERATES): where the backward-Euler equations reduce to a scalar return
    # map or a small benign system -- Hill / FULL_3D, J2 / PLANE_STRESS, Hosford a >= 20 / FULL_3D -- the kernels solve that first
    # and start the reference's Newton at its result (same root; the reference's convergence test on the reference's residual
    # still decides; `status` counts iterations from the warm start).  False: the iteration starts at x_prev and reproduces the
    # reference's iterates and counts -- what the imperative `newton_solve` facade asks for, since it RETURNS the count.
    warm_start: bool = True
    # cm_update runs the iteration-bound configurations on a work pool (a lane that has finished its point takes the next
    # one, include/cmad_hip.h CM_SOLVER_LOCKSTEP); True keeps one point per lane for the whole kernel (A/B measurements).
    lockstep: bool = False

    @classmethod
    def traced(cls, max_iters=10, abs_tol=1e-14, rel_tol=1e-14, line_search_settings=None):
        return cls(max_iters, abs_tol, rel_tol, {**DEFAULT_LINE_SEARCH_SETTINGS, **(line_search_settings or {})})


def _first_key(d):
    return next(iter(d))


def unit_records(nn_params):
    """Per unit of the first hidden layer of an ICNN [6, H, ..., 1]: W0[0..5][o], b0[o], Wz[o], exp(b0[o]), exp(b0[o]) Wz[o]
    (Wz = first z-layer's weights when the next layer is the scalar output, else zeros: deeper networks do not use it)."""
    W0 = np.asarray(nn_params["x params"][0]["weights"], dtype=np.float64)
    b0 = np.asarray(nn_params["x params"][0]["biases"], dtype=np.float64).ravel()
    Wz = np.asarray(nn_params["z params"][0]["weights"], dtype=np.float64)
    wz = Wz.ravel() if Wz.shape[1] == 1 else np.zeros_like(b0)
    with np.errstate(over="ignore"):
        q = np.exp(b0)                              # entries with |b0| >= 150 are never read (two-sided form in the kernel)
        return np.column_stack([W0.T, b0, wz, q, q * wz]).ravel()


class HybridHillEffectiveStress:
    """Selector for the hybrid yield surface phi = phi_hill(sigma) + ICNN(dev sigma)
    (`hybrid_hill_effective_stress`, cmad/models/effective_stress.py:149-163).  Pass an instance as
    `SmallElasticPlastic(..., effective_stress_fun=HybridHillEffectiveStress(icnn))` where the reference passes
    `partial(hybrid_hill_effective_stress, nn_fun=icnn.evaluate)`; the network weights are read from
    params["plastic"]["effective stress"]["neural network"] like the reference does.
    One hidden layer ([6, H, 1]) runs on the kernels' fast evaluation; more hidden layers ([6, H1, ..., Hn, 1], n <= 4,
    at most 64 hidden units) on the general one.  `ScaledHybridHillEffectiveStress` is the beta-rescaled variant
    (`scaled_effective_stress`, :130-146), with the same networks."""
    yield_kind = 3

    def __init__(self, icnn):
        w = list(icnn.layer_widths)
        if len(w) < 3 or w[0] != 6 or w[-1] != 1:
            raise NotImplementedError("the HIP kernels evaluate ICNNs with layer widths [6, H1, ..., Hn, 1]")
        if len(w) > 3 and (len(w) > 6 or sum(w[1:-1]) > 64):
            raise NotImplementedError("networks with several hidden layers: at most 4 hidden layers and 64 hidden units in all "
                                      "(general evaluation of the library's EXT build)")
        self.icnn = icnn

    def packed(self, values):
        """Device layout (include/cmad_hip.h, `nn_weights`): the ICNN packing, then f(0) of the scaled network, then one
        record of 10 doubles per unit of the first hidden layer -- W0[0..5][o], b0[o], Wz[o], exp(b0[o]), exp(b0[o]) Wz[o] --
        which is what the kernels' loop over the hidden units reads (contiguous: a few wide scalar loads per unit; the
        exponential of the bias lets both signs of a unit come from ONE exponential of the lane's own argument)."""
        from ..neural_networks.input_convex_neural_network import forward
        nn_params = values["plastic"]["effective stress"].get("neural network", self.icnn.params)
        widths, w = self.icnn.pack_for_device(nn_params)
        f0 = float(np.asarray(forward(np.zeros(widths[0]), nn_params)).ravel()[0])
        rec = unit_records(nn_params) if len(widths) == 3 else np.zeros(0)       # the fast evaluation's per-unit records
        return widths, np.concatenate([w, [f0], rec])


class ScaledHybridHillEffectiveStress(HybridHillEffectiveStress):
    """`scaled_effective_stress(cauchy, params, effective_stress_fun=hybrid, update_fun=beta_make_newton_solve(hybrid,
    equivalent_stress, max_iters, abs_tol, rel_tol))` (cmad/models/effective_stress.py:97-108, 130-146):
    phi(sigma) = phi_h(beta sigma) / beta with beta such that phi_h(beta sigma) = equivalent_stress, so the network
    is always queried on the level set it was fitted on.  Same argument defaults as `beta_make_newton_solve`."""
    yield_kind = 4

    def __init__(self, icnn, equivalent_stress, max_iters=10, abs_tol=1e-14, rel_tol=1e-14):
        super().__init__(icnn)
        if not equivalent_stress > 0.0:
            raise ValueError("equivalent_stress must be positive")
        self.equivalent_stress = float(equivalent_stress)
        self.max_iters, self.abs_tol, self.rel_tol = int(max_iters), float(abs_tol), float(rel_tol)


def real_tree(values):
    """The real parts of a (possibly complex) parameter tree: what `build_desc` describes for a complex-step model instance."""
    from ..parameters.parameters import tree_map
    return tree_map(lambda v: np.real(v) if np.iscomplexobj(v) else v, values)


def complex_native_parameters(values, yield_type):
    """The 12 native kernel parameters (KP order, include/cmad_hip.h) of a complex parameter tree, as complex numbers: the Lame
    pair through the same conversion table as the real path (analytic in its arguments), the flow-stress and yield coefficients
    as they are.  `cm_update_complex` takes the imaginary parts; the real parts are the model description's."""
    from .elastic_constants import ElasticConstants
    ec = ElasticConstants.from_params({k: complex(v) for k, v in values["elastic"].items()})
    kp = np.zeros(_lib.CM_NUM_PARAMS, dtype=complex)
    kp[_lib.P_LAMBDA], kp[_lib.P_MU] = ec.lmbda, ec.mu
    fs = values["plastic"]["flow stress"]
    kp[_lib.P_Y] = complex(fs["initial yield"]["Y"])
    hard = fs.get("hardening", {}) or {}
    if "voce" in hard:
        kp[_lib.P_VOCE_S], kp[_lib.P_VOCE_D] = complex(hard["voce"]["S"]), complex(hard["voce"]["D"])
    if "linear" in hard:
        kp[_lib.P_LIN_K] = complex(hard["linear"]["K"])
    eff = values["plastic"]["effective stress"]
    if yield_type == "hill":
        for i, n in enumerate(HILL_NAMES):
            kp[_lib.P_YC0 + i] = complex(eff["hill"][n])
    elif yield_type == "hosford":
        kp[_lib.P_YC0] = complex(eff["hosford"]["a"])
    return kp


def complex_parameter_parts(values, flat_paths, info):
    """(p_imag (12,), ext_imag or None) of a complex parameter tree: the imaginary parts `cm_update_complex` takes.  The reference's
    complex instance carries a perturbation of ANY leaf through the model (small_elastic_plastic.py:118-127); here the 12 native
    parameters go in `p_imag` and every other leaf the kernels read -- Barlat coefficients, rotation matrix, network weights of the
    yield surface and of the hardening law -- in `ext_imag`, indexed like the extended parameter index (`leaf_ep_index`) from 12
    on: 13 + 9 + one entry per packed network double.  `flat_paths`: `Parameters.flat_paths()` of the tree."""
    from ..parameters.parameters import ravel_pytree
    p_im = complex_native_parameters(values, info["yield_type"]).imag.copy()
    flat = np.asarray(ravel_pytree(values)[0])
    n_nn = int(np.asarray(info["nn_packed"]).size) if "nn_packed" in info else 0
    ext = None
    if np.iscomplexobj(flat):
        for path, v in zip(flat_paths, flat):
            im = float(np.imag(v))
            if im == 0.0:
                continue
            e = leaf_ep_index(path, info)
            if e is None:                             # a native parameter: complex_native_parameters carries it
                continue
            if e < _lib.CM_NUM_PARAMS:
                p_im[e] = im
            else:
                if ext is None:
                    ext = np.zeros(EP_NN0 - EP_YC6 + n_nn)
                ext[e - EP_YC6] = im
    return p_im, ext


def build_desc(values, def_type=DefType.FULL_3D, model_kind=0, yield_tol=1e-14, uniaxial_stress_idx=0,
               newton: NewtonSettings | None = None, effective_stress_type: str | None = None, hybrid=None,
               hardening_nn=None):
    """Flatten a CMAD parameter tree into a `cm_model_desc`.

    hardening_nn: (input_scale, output_scale) of the `SimpleNeuralNetwork` whose parameters sit under
    params["plastic"]["flow stress"]["hardening"]["neural network"] (the model extracts them from hardening_funs).

    Returns (desc, info) where info carries what the sensitivity mapping needs:
    elastic names and d(lambda, mu)/d(elastic pair)."""
    newton = newton or NewtonSettings()
    d = _lib.ModelDesc()
    d.model_kind = int(model_kind)
    d.def_type = int(def_type)
    plastic = values["plastic"]
    ytype = effective_stress_type or _first_key(plastic["effective stress"])     # small_elastic_plastic.py:190-195
    if hybrid is not None:
        ytype = "hill"
    if ytype not in YIELD_KINDS:
        raise NotImplementedError(f"effective stress '{ytype}' has no HIP kernel")
    d.yield_kind = hybrid.yield_kind if hybrid is not None else YIELD_KINDS[ytype]
    if hybrid is not None and hybrid.yield_kind == 4:
        d.beta_equivalent_stress = hybrid.equivalent_stress
        d.beta_max_iters = hybrid.max_iters
        d.beta_abs_tol, d.beta_rel_tol = hybrid.abs_tol, hybrid.rel_tol
    Q = np.asarray(values.get("rotation matrix", np.eye(3)), dtype=np.float64).reshape(3, 3)
    for i in range(9):
        d.Q[i] = float(Q.reshape(9)[i])
    d.rotation_is_identity = int(np.array_equal(Q, np.eye(3)))
    names, lm, mu, J = lame_jacobian(values["elastic"])
    d.lmbda, d.mu = float(lm), float(mu)
    fs = plastic["flow stress"]
    d.Y = float(fs["initial yield"]["Y"])
    hard = fs.get("hardening", {}) or {}
    for k in hard:
        if k not in ("voce", "linear", "neural network"):
            raise NotImplementedError(f"hardening '{k}'")
    if "neural network" in hard and hardening_nn is None:
        raise KeyError("hardening 'neural network' needs hardening_funs={'neural network': SimpleNeuralNetwork(...).evaluate}")
    d.has_voce = int("voce" in hard)
    d.has_linear = int("linear" in hard)
    if d.has_voce:
        d.voce_S, d.voce_D = float(hard["voce"]["S"]), float(hard["voce"]["D"])
    if d.has_linear:
        d.lin_K = float(hard["linear"]["K"])
    if ytype == "hill":
        h = plastic["effective stress"]["hill"]
        for i, n in enumerate(HILL_NAMES):
            d.yc[i] = float(h[n])
    elif ytype == "hosford":
        d.yc[0] = float(plastic["effective stress"]["hosford"]["a"])
    elif ytype == "barlat":
        coeffs = plastic["effective stress"]["barlat"]
        for i, n in enumerate(BARLAT_NAMES):
            d.yc[i] = float(coeffs[n])
    d.uniaxial_idx = int(uniaxial_stress_idx)
    d.yield_tol = float(yield_tol)
    d.max_iters = int(newton.max_iters)
    d.abs_tol, d.rel_tol = float(newton.abs_tol), float(newton.rel_tol)
    ls = newton.line_search
    d.ls_max_evals = int(ls.get("max evals", 0))
    d.ls_c1 = float(ls.get("sufficient decrease", 1e-4))
    d.ls_lo = float(ls.get("min backtrack factor", 0.5))
    d.ls_hi = float(ls.get("max backtrack factor", 0.9))
    # "kind": "legacy" = the backtracking of the imperative newton_solve(max_ls_evals > 0) (nonlinear_solver.py:55-81), whose
    # beta = 1e-4 and eta = 0.5 travel in ls_c1 / ls_lo; default: the Armijo search of cmad/util/line_search.py
    kind = ls.get("kind", "armijo")
    if kind not in ("armijo", "legacy"):
        raise ValueError(f"line search kind '{kind}'")
    d.ls_kind = _lib.LS_LEGACY if kind == "legacy" else _lib.LS_ARMIJO
    if kind == "legacy":
        d.ls_c1, d.ls_lo = 1e-4, 0.5
    d.solver_flags = (0 if getattr(newton, "j2_radial_line", True) else _lib.SOLVER_GENERAL_NEWTON) | \
                     (_lib.SOLVER_LOCKSTEP if getattr(newton, "lockstep", False) else 0) | \
                     (0 if getattr(newton, "warm_start", True) else _lib.SOLVER_REFERENCE_ITERATES)
    info = {"elastic_names": names, "lame_jac": J, "yield_type": ytype}
    if hybrid is not None:
        widths, packed = hybrid.packed(values)
        d.nn_nlayers = len(widths)
        for i, w in enumerate(widths):
            d.nn_widths[i] = int(w)
        info["yield_type"] = "hybrid"
        info["nn_widths"] = [int(w) for w in widths]
        info["nn_packed"] = np.ascontiguousarray(packed, dtype=np.float64)   # the caller places it and sets d.nn_weights
    if "neural network" in hard:
        # the hardening network travels in the same device buffer, after the yield network's pack (if any)
        from ..neural_networks.simple_neural_network import pack_hardening_network, pack_hardening_network_deep
        base = info.get("nn_packed", np.zeros(0))
        if len(hard["neural network"]) == 2:                    # widths [1, H, 1]: the one-exponential-per-unit kernel loop
            Hn, hpack = pack_hardening_network(hard["neural network"], *hardening_nn)
            info["hnn"] = (int(Hn), int(base.size))
        else:                                                   # [1, H1, ..., Hn, 1]: the general forward pass of the EXT build
            hidden, hpack = pack_hardening_network_deep(hard["neural network"], *hardening_nn)
            Hn = hidden[0]
            d.hnn_nhidden = len(hidden)
            for i, wdt in enumerate(hidden):
                d.hnn_widths[i] = int(wdt)
            info["hnn"] = (int(Hn), int(base.size))
            info["hnn_hidden"] = [int(wdt) for wdt in hidden]
        d.hnn_width, d.hnn_offset = int(Hn), int(base.size)
        info["nn_packed"] = np.ascontiguousarray(np.concatenate([base, hpack]), dtype=np.float64)
    return d, info


class _ExtendedLeaf(NotImplementedError):
    """The leaf is not among the 12 native kernel parameters: its sensitivities come from the forward-mode entry points
    (`leaf_ep_index` -> cm_param_blocks / cm_param_adjoint_history / cm_*_history_ep).  Callers catch this and take that route."""


def kp_to_leaf_grad(path, g_kp, info):
    """Sensitivity w.r.t. the parameter-tree leaf at `path` from the kernel's KP-order vector
    (include/cmad_hip.h `cm_param_index`).  `g_kp` may have trailing batch/row dims on axis 0 = KP.
    Raises `_ExtendedLeaf` (a NotImplementedError) for the leaves served by `leaf_ep_index` instead."""
    leaf = path[-1]
    if path[0] == "elastic":
        j = info["elastic_names"].index(leaf)
        J = info["lame_jac"]
        return g_kp[_lib.P_LAMBDA] * J[0, j] + g_kp[_lib.P_MU] * J[1, j]
    if path[0] == "rotation matrix":
        raise _ExtendedLeaf("rotation matrix: extended parameter (leaf_ep_index)")
    if leaf == "Y":
        return g_kp[_lib.P_Y]
    parent = path[-2] if len(path) >= 2 else None
    if parent == "voce":
        return g_kp[{"S": _lib.P_VOCE_S, "D": _lib.P_VOCE_D}[leaf]]
    if parent == "linear":
        return g_kp[_lib.P_LIN_K]
    if parent == "hill":
        if info.get("yield_type") == "hybrid":
            raise _ExtendedLeaf("Hill coefficients of the hybrid surface: extended parameters (leaf_ep_index)")
        return g_kp[_lib.P_YC0 + HILL_NAMES.index(leaf)]
    if "neural network" in path:
        raise _ExtendedLeaf("network weights: extended parameters (leaf_ep_index)")
    if parent == "effective stress" and leaf == "J2":
        return g_kp[0] * 0.0
    if parent == "hosford":
        raise _ExtendedLeaf("Hosford exponent: extended parameter (leaf_ep_index)")
    if parent == "barlat":
        raise _ExtendedLeaf("Barlat coefficients: extended parameters (leaf_ep_index)")
    raise KeyError(path)


# extended parameter ("EP") indexing of cm_param_blocks / cm_param_adjoint_history (include/cmad_hip.h)
EP_YC6, EP_Q0, EP_NN0 = 12, 25, 34


def leaf_ep_index(path, info):
    """EP index of the parameter-tree entry at `path` (as `Parameters.active_paths()` gives it: array leaves end with the
    flat element index) when the entry is differentiated by forward-mode evaluation of the whole model (rotation matrix,
    Hosford exponent, Hill coefficients of the network surfaces, network weights); None when the 12 hand-derived native
    sensitivities (`kp_to_leaf_grad`) cover it."""
    if path[0] == "rotation matrix":
        idx = [k for k in path[1:] if isinstance(k, (int, np.integer))]
        flat = idx[0] if len(idx) == 1 else 3 * idx[0] + idx[1]
        return EP_Q0 + int(flat)
    names = [k for k in path if isinstance(k, str)]
    if "hardening" in names and "neural network" in names:
        # [1, H, 1] hardening network in the nn buffer at info["hnn"] = (H, offset): W1[H], b1[H], W2[H], b2
        Hn, off = info["hnn"]
        ints = [k for k in path if isinstance(k, (int, np.integer))]
        layer, elem = int(ints[0]), (int(ints[-1]) if len(ints) > 1 else 0)
        what = names[-1]
        if "hnn_hidden" in info:
            # general layout: for every layer W[n_in][n_out] then b[n_out] (array leaves carry the flat element index)
            sizes = [1] + info["hnn_hidden"] + [1]
            loff = sum(sizes[l] * sizes[l + 1] + sizes[l + 1] for l in range(layer))
            return EP_NN0 + off + loff + (elem if what == "weights" else sizes[layer] * sizes[layer + 1] + elem)
        if layer == 0:
            return EP_NN0 + off + (elem if what == "weights" else Hn + elem)
        return EP_NN0 + off + (2 * Hn + elem if what == "weights" else 3 * Hn)
    if "neural network" in names:
        widths = info.get("nn_widths")
        if widths is None:
            raise NotImplementedError("network-weight sensitivities need the hybrid surface")
        # packed layout (pack_for_device): x-layer k: W[6][H_k], b[H_k] for k = 1 .. n+1, then z-layer k: W[H_k][H_{k+1}]
        ints = [k for k in path if isinstance(k, (int, np.integer))]
        layer, elem = int(ints[0]), (int(ints[-1]) if len(ints) > 1 else 0)
        outs = list(widths[1:])
        if "x params" in names:
            off = sum(7 * h for h in outs[:layer])
            return EP_NN0 + off + (elem if names[-1] == "weights" else 6 * outs[layer] + elem)
        if "z params" in names:
            off = sum(7 * h for h in outs) + sum(a * b for a, b in zip(outs[:layer], outs[1:layer + 1]))
            return EP_NN0 + off + elem
        raise KeyError(path)
    parent = names[-2] if len(names) >= 2 else None
    if parent == "hosford":
        return _lib.P_YC0
    if parent == "hill" and info.get("yield_type") == "hybrid":
        return _lib.P_YC0 + HILL_NAMES.index(names[-1])
    if parent == "barlat":
        i = BARLAT_NAMES.index(names[-1])                          # cm_model_desc.yc slot: 0..5 native slots, 6..18 extended
        return _lib.P_YC0 + i if i < 6 else EP_YC6 + (i - 6)
    return None


def _torch():
    import torch
    return torch


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check_soa(t, rows, B, name):
    torch = _torch()
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
            and tuple(t.shape) == (rows, B)):
        raise ValueError(f"{name}: expected a contiguous float64 CUDA tensor of shape ({rows}, {B}), "
                         f"got {type(t).__name__} {getattr(t, 'shape', None)} {getattr(t, 'dtype', None)} "
                         f"{getattr(t, 'device', None)}")


def fold_weight_and_data(weight, data9=None):
    """Fold a 3x3 weight mask (qois/calibration.py:23-33) into squared weights of the 6 unique stress
    entries; optionally fold (possibly non-symmetric) 3x3 data into the equivalent symmetric 6-vector
    data plus a constant:  1/2 w_ij^2 (s - d_ij)^2 + 1/2 w_ji^2 (s - d_ji)^2
                         = 1/2 W (s - D)^2 + const,  W = w_ij^2 + w_ji^2,  D = (w_ij^2 d_ij + w_ji^2 d_ji)/W."""
    w = np.asarray(weight, dtype=np.float64).reshape(3, 3)
    pairs = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    wsq6 = np.array([w[i, j] ** 2 if i == j else w[i, j] ** 2 + w[j, i] ** 2 for i, j in pairs])
    if data9 is None:
        return wsq6
    d = np.asarray(data9, dtype=np.float64)            # (3, 3, ...)
    data6 = np.zeros((6,) + d.shape[2:])
    const = np.zeros(d.shape[2:])
    for r, (i, j) in enumerate(pairs):
        if i == j:
            data6[r] = d[i, j]
        else:
            a, b = w[i, j] ** 2, w[j, i] ** 2
            W = a + b
            D = (a * d[i, j] + b * d[j, i]) / W if W != 0.0 else 0.5 * (d[i, j] + d[j, i])
            data6[r] = D
            const = const + 0.5 * (a * d[i, j] ** 2 + b * d[j, i] ** 2 - W * D ** 2)
    return wsq6, data6, const


class DeviceEvaluator:
    """Launches the batched kernels for one model description on the current torch CUDA stream."""

    def __init__(self, desc, info):
        self.desc, self.info = desc, info
        self.L = _lib.lib()
        self.nx = self.L.cm_num_xi(C.byref(desc))
        self.nu = self.L.cm_num_gradu(C.byref(desc))
        if self.nx < 0 or self.nu < 0:
            raise NotImplementedError("def_type not available in the HIP library")
        self._ws, self._captured_ws = {}, []
        self._nn_dev = None
        if "nn_packed" in info:                    # network weights live in device memory for the kernels
            torch = _torch()
            self._nn_dev = torch.from_numpy(info["nn_packed"]).cuda()
            desc.nn_weights = self._nn_dev.data_ptr()

    # -- helpers
    def _stream(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _workspace(self, B, device):
        """Device scratch of `cm_workspace_bytes(B)` bytes (block partials of the reductions + the screened update's list,
        include/cmad_hip.h).  One per STREAM -- launches on different streams may run at the same time -- and a private one for
        every launch captured into a HIP graph, which keeps its workspace for all its replays."""
        torch = _torch()
        need = self.L.cm_workspace_bytes(B)
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=device)      # from the graph's private pool
            self._captured_ws.append(ws)
            return ws, need
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * 8 < need:
            ws = self._ws[key] = torch.empty((need + 7) // 8, dtype=torch.float64, device=device)
        return ws, need

    def screened(self, B):
        """True when `cm_update_ws` runs this configuration screened (k_screen + k_update_listed; `screen_route` in
        cmad_hip.hip): FULL_3D, total form, the network surfaces and Barlat, 4096 <= B < 2^29."""
        d = self.desc
        if d.model_kind != 0 or d.def_type != 0 or (d.solver_flags & _lib.SOLVER_LOCKSTEP) or not (4096 <= B < 2 ** 29):
            return False
        if os.environ.get("CM_DEBUG_NO_SCREEN", "0") not in ("", "0"):
            return False
        return d.yield_kind in (3, 4, 5)

    def _update_workspace(self, B, device):
        """(workspace, bytes) for cm_update_ws: the evaluator's per-stream workspace when the screened route applies, else none
        (the call is then plain cm_update; CM_DEBUG_NO_SCREEN=1 keeps the work pool for A/B measurements)."""
        if not self.screened(B):
            return None, 0
        ws, need = self._workspace(B, device)
        return ws, need

    # -- entry points
    def pool_route(self, B):
        """True when `cm_update` runs this configuration on the work pool (network surfaces, Hosford under the line search;
        B >= 256, total form, no CM_SOLVER_LOCKSTEP -- `pool_route` in cmad_hip.hip).  For those the multi-kernel routes
        (work-pool update + reverse kernel per step) beat the single fused lockstep kernels, whose wavefronts wait for their
        slowest point; callers use this to pick the route (`cm_update_and_vjp` and `cm_objective_grad` with a state buffer
        do it themselves)."""
        d = self.desc
        if d.model_kind != 0 or (d.solver_flags & _lib.SOLVER_LOCKSTEP) or B < 256:
            return False
        if d.yield_kind == YIELD_KINDS["hosford"]:
            # FULL_3D with a >= 20 starts the Newton at the analytic warm start (cm::hosford_warm_start): lockstep kernels
            warm = (d.def_type == 0 and d.yc[0] >= 20.0 and not (d.solver_flags & (_lib.SOLVER_GENERAL_NEWTON | _lib.SOLVER_REFERENCE_ITERATES))
                    and not (d.ls_max_evals > 0 and d.ls_kind == 1) and d.hnn_width == 0)
            return d.ls_max_evals > 0 and not warm
        return d.yield_kind in (3, 4)

    def update(self, gradu, xi_prev, want_sigma=True, want_status=True, tangent=False, out=None):
        torch = _torch()
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(xi_prev, self.nx, B, "xi_prev")
        dev = gradu.device
        o = out or {}
        xi = o.get("xi") if o.get("xi") is not None else torch.empty((self.nx, B), dtype=torch.float64, device=dev)
        sigma = (o.get("sigma") if o.get("sigma") is not None else
                 torch.empty((6, B), dtype=torch.float64, device=dev)) if want_sigma else None
        status = (o.get("status") if o.get("status") is not None else
                  torch.empty((B,), dtype=torch.int32, device=dev)) if want_status else None
        if tangent:
            ds = o.get("dsigma") if o.get("dsigma") is not None else torch.empty((6 * self.nu, B), dtype=torch.float64, device=dev)
            _check_soa(ds, 6 * self.nu, B, "dsigma")
            ws, need = self._update_workspace(B, dev)
            rc = self.L.cm_update_tangent_ws(C.byref(self.desc), B, _ptr(gradu), _ptr(xi_prev), _ptr(xi), _ptr(sigma),
                                             _ptr(ds), _ptr(status), _ptr(ws), need, self._stream())
            _lib.check(rc, "cm_update_tangent")
            return xi, sigma, status, ds.view(6, self.nu, B)
        ws, need = self._update_workspace(B, dev)
        rc = self.L.cm_update_ws(C.byref(self.desc), B, _ptr(gradu), _ptr(xi_prev), _ptr(xi), _ptr(sigma),
                                 _ptr(status), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_update")
        return xi, sigma, status

    def update_rate(self, gradu, gradu_prev, xi_prev, want_sigma=True, want_status=True, tangent=False):
        """Rate-form model (`small_rate_elastic_plastic`): xi = [sigma(6), alpha (, F33)].
        tangent=True also returns d sigma / d gradu (6 * n_gradu, B) (`cm_update_rate_tangent`); the derivative
        w.r.t. gradu_prev is its negative."""
        torch = _torch()
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(gradu_prev, self.nu, B, "gradu_prev")
        _check_soa(xi_prev, self.nx, B, "xi_prev")
        dev = gradu.device
        xi = torch.empty((self.nx, B), dtype=torch.float64, device=dev)
        sigma = torch.empty((6, B), dtype=torch.float64, device=dev) if want_sigma else None
        status = torch.empty((B,), dtype=torch.int32, device=dev) if want_status else None
        if tangent:
            dsig = torch.empty((6 * self.nu, B), dtype=torch.float64, device=dev)
            rc = self.L.cm_update_rate_tangent(C.byref(self.desc), B, _ptr(gradu), _ptr(gradu_prev), _ptr(xi_prev), _ptr(xi),
                                               _ptr(sigma), _ptr(dsig), _ptr(status), self._stream())
            _lib.check(rc, "cm_update_rate_tangent")
            return xi, sigma, status, dsig
        rc = self.L.cm_update_rate(C.byref(self.desc), B, _ptr(gradu), _ptr(gradu_prev), _ptr(xi_prev), _ptr(xi),
                                   _ptr(sigma), _ptr(status), self._stream())
        _lib.check(rc, "cm_update_rate")
        return xi, sigma, status

    def update_complex(self, p_imag, gradu, xi_prev, xi_start, gradu_prev=None, ext_imag=None):
        """`cm_update_complex`: the local Newton solve of a complex-step model instance (reference `Model(..., is_complex=True)`
        under `newton_solve`).  p_imag: (12,) imaginary parts of the native parameters (host); ext_imag: imaginary parts of the
        extended parameters (`complex_parameter_parts`; host array or None); gradu (n_gradu, B) real; xi_prev, xi_start
        (2, n_xi, B): real rows then imaginary rows.  Returns (xi (2, n_xi, B), residual (2, n_xi, B), sigma (2, 6, B), status);
        `max_iters = 0` in the description evaluates residual and stress at xi_start."""
        torch = _torch()
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu")
        gp = self._rate_prev(gradu_prev, B)
        for t, name in ((xi_prev, "xi_prev"), (xi_start, "xi_start")):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
                    and tuple(t.shape) == (2, self.nx, B)):
                raise ValueError(f"{name}: expected a contiguous float64 CUDA tensor of shape (2, {self.nx}, {B})")
        dev = gradu.device
        xi = xi_start.clone()
        res = torch.empty((2, self.nx, B), dtype=torch.float64, device=dev)
        sigma = torch.empty((2, 6, B), dtype=torch.float64, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        pim = (C.c_double * _lib.CM_NUM_PARAMS)(*[float(v) for v in p_imag])
        ext = None
        if ext_imag is not None:
            n_nn = int(np.asarray(self.info["nn_packed"]).size) if "nn_packed" in self.info else 0
            if np.asarray(ext_imag).shape != (EP_NN0 - EP_YC6 + n_nn,):
                raise ValueError(f"ext_imag: expected {EP_NN0 - EP_YC6 + n_nn} entries (13 + 9 + the packed network doubles)")
            ext = torch.from_numpy(np.ascontiguousarray(ext_imag, dtype=np.float64)).to(dev)
        rc = self.L.cm_update_complex(C.byref(self.desc), B, pim, _ptr(ext), _ptr(gradu), _ptr(gp), _ptr(xi_prev), _ptr(xi), _ptr(res),
                                      _ptr(sigma), _ptr(status), self._stream())
        _lib.check(rc, "cm_update_complex")
        return xi, res, sigma, status

    def _rate_prev(self, gradu_prev, B):
        """The rate-form model (desc.model_kind = 1) takes the previous grad u in every entry point."""
        if self.desc.model_kind != 1:
            if gradu_prev is not None:
                raise ValueError("gradu_prev only applies to the rate-form model")
            return None
        if gradu_prev is None:
            raise ValueError("the rate-form model needs gradu_prev")
        _check_soa(gradu_prev, self.nu, B, "gradu_prev")
        return gradu_prev

    def update_vjp(self, gradu, xi_prev, xi, sigma_bar, want_xi_prev_bar=False, want_gradu_bar=False, gradu_prev=None):
        """gradu_prev: rate-form model only (the returned gradu cotangent's negative is the one of gradu_prev)."""
        torch = _torch()
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(xi_prev, self.nx, B, "xi_prev")
        _check_soa(xi, self.nx, B, "xi"); _check_soa(sigma_bar, 6, B, "sigma_bar")
        dev = gradu.device
        g = torch.empty(_lib.CM_NUM_PARAMS, dtype=torch.float64, device=dev)
        xb = torch.empty((self.nx, B), dtype=torch.float64, device=dev) if want_xi_prev_bar else None
        ub = torch.empty((self.nu, B), dtype=torch.float64, device=dev) if want_gradu_bar else None
        ws, need = self._workspace(B, dev)
        gp = self._rate_prev(gradu_prev, B)
        if gp is not None:
            rc = self.L.cm_update_rate_vjp(C.byref(self.desc), B, _ptr(gradu), _ptr(gp), _ptr(xi_prev), _ptr(xi),
                                           _ptr(sigma_bar), _ptr(g), _ptr(xb), _ptr(ub), _ptr(ws), need, self._stream())
            _lib.check(rc, "cm_update_rate_vjp")
            return g, xb, ub
        rc = self.L.cm_update_vjp(C.byref(self.desc), B, _ptr(gradu), _ptr(xi_prev), _ptr(xi), _ptr(sigma_bar),
                                  _ptr(g), _ptr(xb), _ptr(ub), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_update_vjp")
        return g, xb, ub

    def update_and_vjp(self, gradu, xi_prev, sigma_bar, want_sigma=True, out=None, gradu_prev=None):
        """Fused cm_update + cm_update_vjp; returns xi, sigma, grad_kp (device, KP order)."""
        torch = _torch()
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(xi_prev, self.nx, B, "xi_prev")
        _check_soa(sigma_bar, 6, B, "sigma_bar")
        dev = gradu.device
        o = out or {}
        xi = o.get("xi") if o.get("xi") is not None else torch.empty((self.nx, B), dtype=torch.float64, device=dev)
        sigma = (o.get("sigma") if o.get("sigma") is not None else
                 torch.empty((6, B), dtype=torch.float64, device=dev)) if want_sigma else None
        g = o.get("grad") if o.get("grad") is not None else torch.empty(_lib.CM_NUM_PARAMS, dtype=torch.float64, device=dev)
        ws, need = self._workspace(B, dev)
        gp = self._rate_prev(gradu_prev, B)
        if gp is not None:
            rc = self.L.cm_update_rate_and_vjp(C.byref(self.desc), B, _ptr(gradu), _ptr(gp), _ptr(xi_prev), _ptr(sigma_bar),
                                               _ptr(xi), _ptr(sigma), _ptr(g), _ptr(ws), need, self._stream())
            _lib.check(rc, "cm_update_rate_and_vjp")
            return xi, sigma, g
        rc = self.L.cm_update_and_vjp(C.byref(self.desc), B, _ptr(gradu), _ptr(xi_prev), _ptr(sigma_bar), _ptr(xi),
                                      _ptr(sigma), _ptr(g), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_update_and_vjp")
        return xi, sigma, g

    def objective_grad(self, gradu, xi_prev, data6, wsq6, want_xi=False, out=None, gradu_prev=None):
        torch = _torch()
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(xi_prev, self.nx, B, "xi_prev")
        _check_soa(data6, 6, B, "data")
        dev = gradu.device
        res = out if out is not None else torch.empty(1 + _lib.CM_NUM_PARAMS, dtype=torch.float64, device=dev)
        xi = torch.empty((self.nx, B), dtype=torch.float64, device=dev) if want_xi else None
        w = (C.c_double * 6)(*[float(v) for v in wsq6])
        ws, need = self._workspace(B, dev)
        gp = self._rate_prev(gradu_prev, B)
        if gp is not None:
            rc = self.L.cm_objective_grad_rate(C.byref(self.desc), B, _ptr(gradu), _ptr(gp), _ptr(xi_prev), _ptr(data6), w,
                                               _ptr(res), _ptr(xi), _ptr(ws), need, self._stream())
            _lib.check(rc, "cm_objective_grad_rate")
            return res, xi
        rc = self.L.cm_objective_grad(C.byref(self.desc), B, _ptr(gradu), _ptr(xi_prev), _ptr(data6), w,
                                      _ptr(res), _ptr(xi), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_objective_grad")
        return res, xi

    def direct_step(self, gradu, xi_prev, xi, dxi_prev_dp=None, want_dsigma=True, gradu_prev=None):
        """`cm_direct_step`: forward parameter sensitivities of one converged step.  Returns (dxi_dp (n_xi, 12, B),
        dsigma_dp (6, 12, B) or None), native parameters in KP order; chain the columns to the caller's parameters with
        `kp_to_leaf_grad` / `Model.active_grad_from_kp`."""
        torch = _torch()
        B = gradu.shape[1]
        NP = _lib.CM_NUM_PARAMS
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(xi_prev, self.nx, B, "xi_prev"); _check_soa(xi, self.nx, B, "xi")
        gp = self._rate_prev(gradu_prev, B)
        if dxi_prev_dp is not None:
            if not (dxi_prev_dp.is_cuda and dxi_prev_dp.dtype == torch.float64 and dxi_prev_dp.is_contiguous()
                    and tuple(dxi_prev_dp.shape) == (self.nx, NP, B)):
                raise ValueError(f"dxi_prev_dp: expected a contiguous float64 CUDA tensor of shape ({self.nx}, {NP}, {B})")
        dev = gradu.device
        dx = torch.empty((self.nx, NP, B), dtype=torch.float64, device=dev)
        ds = torch.empty((6, NP, B), dtype=torch.float64, device=dev) if want_dsigma else None
        rc = self.L.cm_direct_step(C.byref(self.desc), B, _ptr(gradu), _ptr(gp), _ptr(xi_prev), _ptr(xi), _ptr(dxi_prev_dp),
                                   _ptr(dx), _ptr(ds), self._stream())
        _lib.check(rc, "cm_direct_step")
        return dx, ds

    def update_history(self, gradu_hist, xi0, want_xi=True, want_sigma=True, want_status=True):
        """`cm_update_history`: K updates per point in one launch.  Returns (xi_hist (K+1, n_xi, B),
        sigma_hist (K+1, 6, B), status_hist (K+1, B) int32), None for the ones not requested."""
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        if K < 1:
            raise ValueError("a load history needs at least one step after the initial configuration")
        if not (want_xi or want_sigma):
            raise ValueError("request the state history, the stress history or both")
        if not (gradu_hist.dim() == 3 and gradu_hist.is_contiguous()):
            raise ValueError("gradu_hist (K+1, n_gradu, B) must be contiguous")
        _check_soa(gradu_hist[0], self.nu, B, "gradu_hist")
        _check_soa(xi0, self.nx, B, "xi0")
        dev = gradu_hist.device
        xi_hist = torch.empty((K + 1, self.nx, B), dtype=torch.float64, device=dev) if want_xi else None
        sig_hist = torch.empty((K + 1, 6, B), dtype=torch.float64, device=dev) if want_sigma else None
        st_hist = torch.empty((K + 1, B), dtype=torch.int32, device=dev) if want_status else None
        rc = self.L.cm_update_history(C.byref(self.desc), B, K, _ptr(gradu_hist), _ptr(xi0), _ptr(xi_hist), _ptr(sig_hist),
                                      _ptr(st_hist), self._stream())
        _lib.check(rc, "cm_update_history")
        return xi_hist, sig_hist, st_hist

    def objective_grad_history(self, gradu_hist, data_hist, wsq6, xi0, xi_hist=None, out=None):
        """`cm_objective_grad_history`: objective + gradient over a whole (K+1, n, B) load history in one launch.
        Returns (out[1 + CM_NUM_PARAMS] on the device, xi_hist (K+1, n_xi, B))."""
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        if K < 1:
            raise ValueError("a load history needs at least one step after the initial configuration")
        if not (gradu_hist.dim() == 3 and data_hist.dim() == 3 and gradu_hist.is_contiguous() and data_hist.is_contiguous()
                and data_hist.shape[0] == K + 1):
            raise ValueError("gradu_hist (K+1, n_gradu, B) and data_hist (K+1, 6, B) must be contiguous with the same K")
        _check_soa(gradu_hist[0], self.nu, B, "gradu_hist"); _check_soa(data_hist[0], 6, B, "data_hist")
        _check_soa(xi0, self.nx, B, "xi0")
        dev = gradu_hist.device
        if xi_hist is None:
            xi_hist = torch.empty((K + 1, self.nx, B), dtype=torch.float64, device=dev)
        elif tuple(xi_hist.shape) != (K + 1, self.nx, B) or not xi_hist.is_contiguous() or xi_hist.dtype != torch.float64:
            raise ValueError("xi_hist must be a contiguous float64 (K+1, n_xi, B) tensor")
        res = out if out is not None else torch.empty(1 + _lib.CM_NUM_PARAMS, dtype=torch.float64, device=dev)
        w = (C.c_double * 6)(*[float(v) for v in wsq6])
        ws, need = self._workspace(B, dev)
        rc = self.L.cm_objective_grad_history(C.byref(self.desc), B, K, _ptr(gradu_hist), _ptr(data_hist), w, _ptr(xi0),
                                              _ptr(xi_hist), _ptr(res), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_objective_grad_history")
        return res, xi_hist

    def _check_hist(self, t, rows, K, B, name):
        torch = _torch()
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
                and tuple(t.shape) == (K + 1, rows, B)):
            raise ValueError(f"{name}: expected a contiguous float64 CUDA tensor of shape ({K + 1}, {rows}, {B})")

    def adjoint_history(self, gradu_hist, sigma_bar_hist, xi0, xi_bar_hist=None, want_lam=False, xi_hist=None):
        """`cm_adjoint_history`: the adjoint recursion of a whole history for caller-supplied QoI cotangents.
        Returns (grad_kp (12,) device, xi_hist (K+1, n_xi, B), lam_hist (K+1, n_xi, B) or None)."""
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        if K < 1:
            raise ValueError("a load history needs at least one step after the initial configuration")
        self._check_hist(gradu_hist, self.nu, K, B, "gradu_hist")
        self._check_hist(sigma_bar_hist, 6, K, B, "sigma_bar_hist")
        if xi_bar_hist is not None:
            self._check_hist(xi_bar_hist, self.nx, K, B, "xi_bar_hist")
        _check_soa(xi0, self.nx, B, "xi0")
        dev = gradu_hist.device
        if xi_hist is None:
            xi_hist = torch.empty((K + 1, self.nx, B), dtype=torch.float64, device=dev)
        lam = torch.zeros((K + 1, self.nx, B), dtype=torch.float64, device=dev) if want_lam else None
        g = torch.empty(_lib.CM_NUM_PARAMS, dtype=torch.float64, device=dev)
        ws, need = self._workspace(B, dev)
        rc = self.L.cm_adjoint_history(C.byref(self.desc), B, K, _ptr(gradu_hist), _ptr(sigma_bar_hist), _ptr(xi_bar_hist), _ptr(xi0),
                                       _ptr(xi_hist), _ptr(lam), _ptr(g), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_adjoint_history")
        return g, xi_hist, lam

    def direct_history(self, gradu_hist, xi_hist, sigma_bar_hist=None, xi_bar_hist=None, want_blocks=False):
        """`cm_direct_history`: forward sensitivities over a stored history in one launch.  Returns (grad_kp (12,) device
        or None, dxi_dp_hist (K+1, n_xi, 12, B) or None, dsigma_dp_hist (K+1, 6, 12, B) or None)."""
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        NP = _lib.CM_NUM_PARAMS
        self._check_hist(gradu_hist, self.nu, K, B, "gradu_hist")
        self._check_hist(xi_hist, self.nx, K, B, "xi_hist")
        if sigma_bar_hist is None and not want_blocks:
            raise ValueError("request the gradient (pass the QoI cotangents), the sensitivity blocks, or both")
        if sigma_bar_hist is not None:
            self._check_hist(sigma_bar_hist, 6, K, B, "sigma_bar_hist")
        if xi_bar_hist is not None:
            self._check_hist(xi_bar_hist, self.nx, K, B, "xi_bar_hist")
        dev = gradu_hist.device
        dx = torch.empty((K + 1, self.nx, NP, B), dtype=torch.float64, device=dev) if want_blocks else None
        ds = torch.empty((K + 1, 6, NP, B), dtype=torch.float64, device=dev) if want_blocks else None
        g = ws = None
        need = 0
        if sigma_bar_hist is not None:
            g = torch.empty(NP, dtype=torch.float64, device=dev)
            need = self.L.cm_direct_workspace_bytes(B)
            ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=dev)
        rc = self.L.cm_direct_history(C.byref(self.desc), B, K, _ptr(gradu_hist), _ptr(xi_hist), _ptr(sigma_bar_hist),
                                      _ptr(xi_bar_hist), _ptr(dx), _ptr(ds), _ptr(g), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_direct_history")
        return g, dx, ds

    def hessian_history(self, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, sigma_bar_hist, hss, hxx=None):
        """`cm_hessian_history`: d2J/dp2 (12, 12) device tensor, KP order, summed over the batch and the steps.
        `hss`: the QoI's diagonal stress curvature, (6,) constant in time or (K+1, 6) per step; `hxx`: (K+1, n_xi) diagonal
        curvature in the state entries of a QoI with an explicit dJ/dxi, or None."""
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        NP = _lib.CM_NUM_PARAMS
        self._check_hist(gradu_hist, self.nu, K, B, "gradu_hist")
        self._check_hist(xi_hist, self.nx, K, B, "xi_hist")
        self._check_hist(lam_hist, self.nx, K, B, "lam_hist")
        self._check_hist(sigma_bar_hist, 6, K, B, "sigma_bar_hist")
        if not (dxi_dp_hist.is_cuda and dxi_dp_hist.is_contiguous() and tuple(dxi_dp_hist.shape) == (K + 1, self.nx, NP, B)):
            raise ValueError(f"dxi_dp_hist: expected a contiguous float64 CUDA tensor of shape ({K + 1}, {self.nx}, {NP}, {B})")
        dev = gradu_hist.device
        H = torch.empty((NP, NP), dtype=torch.float64, device=dev)
        need = self.L.cm_hessian_workspace_bytes(C.byref(self.desc), B, K)
        if need < 0:
            _lib.check(int(need), "cm_hessian_workspace_bytes")
        ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=dev)
        hss = np.asarray(hss, dtype=np.float64)
        h6, hs_dev, hx_dev = None, None, None
        if hss.ndim == 1:
            h6 = (C.c_double * 6)(*[float(v) for v in hss])
        else:
            if hss.shape != (K + 1, 6):
                raise ValueError(f"hss: expected (6,) or ({K + 1}, 6)")
            hs_dev = torch.from_numpy(np.ascontiguousarray(hss)).to(dev)
        if hxx is not None:
            hxx = np.asarray(hxx, dtype=np.float64)
            if hxx.shape != (K + 1, self.nx):
                raise ValueError(f"hxx: expected ({K + 1}, {self.nx})")
            hx_dev = torch.from_numpy(np.ascontiguousarray(hxx)).to(dev)
        rc = self.L.cm_hessian_history(C.byref(self.desc), B, K, _ptr(gradu_hist), _ptr(xi_hist), _ptr(lam_hist), _ptr(dxi_dp_hist),
                                       _ptr(sigma_bar_hist), h6, _ptr(hs_dev), _ptr(hx_dev), _ptr(H), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_hessian_history")
        return H

    MAX_EXTENDED = 64          # kMaxEp in cmad_hip.hip: extended directions per second-order pass (the quadratic form's LDS tile)

    def _check_extended(self, ep_index, what):
        if len(ep_index) > self.MAX_EXTENDED:
            raise NotImplementedError(f"{what}: {len(ep_index)} active leaves outside the 12 native kernel parameters; the second-order "
                                      f"pass carries at most {self.MAX_EXTENDED} per evaluation (e.g. every weight of an ICNN [6, 16, 1] "
                                      "is 135): activate a subset, or take gradients only (MPAdjointObjective has no such limit)")

    def direct_history_ep(self, ep_index, gradu_hist, xi_hist):
        """`cm_direct_history_ep`: forward sensitivities dxi_k/dp_e of the extended parameters over a stored history,
        (K+1, n_xi, n_ep, B) device tensor."""
        self._check_extended(ep_index, "cm_direct_history_ep")
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        self._check_hist(gradu_hist, self.nu, K, B, "gradu_hist")
        self._check_hist(xi_hist, self.nx, K, B, "xi_hist")
        dev = gradu_hist.device
        ep = torch.tensor([int(e) for e in ep_index], dtype=torch.int32, device=dev)
        dxe = torch.empty((K + 1, self.nx, len(ep_index), B), dtype=torch.float64, device=dev)
        rc = self.L.cm_direct_history_ep(C.byref(self.desc), B, K, len(ep_index), _ptr(ep), _ptr(gradu_hist), _ptr(xi_hist), _ptr(dxe),
                                         self._stream())
        _lib.check(rc, "cm_direct_history_ep")
        return dxe

    def hessian_history_ep(self, ep_index, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, dxi_dpe_hist, sigma_bar_hist, hss, hxx=None):
        """`cm_hessian_history_ep`: d2J/d[p, pe]2, (12 + n_ep, 12 + n_ep) device tensor (native parameters first, KP order)."""
        self._check_extended(ep_index, "cm_hessian_history_ep")
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        ne = len(ep_index)
        npt = _lib.CM_NUM_PARAMS + ne
        dev = gradu_hist.device
        ep = torch.tensor([int(e) for e in ep_index], dtype=torch.int32, device=dev)
        H = torch.empty((npt, npt), dtype=torch.float64, device=dev)
        need = self.L.cm_hessian_ep_workspace_bytes(C.byref(self.desc), B, K, ne)
        if need < 0:
            _lib.check(int(need), "cm_hessian_ep_workspace_bytes")
        ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=dev)
        hss = np.asarray(hss, dtype=np.float64)
        h6 = (C.c_double * 6)(*[float(v) for v in hss]) if hss.ndim == 1 else None
        hs_dev = None if hss.ndim == 1 else torch.from_numpy(np.ascontiguousarray(hss)).to(dev)
        hx_dev = None if hxx is None else torch.from_numpy(np.ascontiguousarray(hxx, dtype=np.float64)).to(dev)
        rc = self.L.cm_hessian_history_ep(C.byref(self.desc), B, K, ne, _ptr(ep), _ptr(gradu_hist), _ptr(xi_hist), _ptr(lam_hist),
                                          _ptr(dxi_dp_hist), _ptr(dxi_dpe_hist), _ptr(sigma_bar_hist), h6, _ptr(hs_dev), _ptr(hx_dev),
                                          _ptr(H), _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_hessian_history_ep")
        return H

    def param_blocks(self, ep_index, gradu, xi_prev, xi, gradu_prev=None):
        """`cm_param_blocks`: (dC_dp (n_ep, n_xi, B), dsigma_dp (n_ep, 6, B)) for the extended parameter indices."""
        torch = _torch()
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(xi_prev, self.nx, B, "xi_prev"); _check_soa(xi, self.nx, B, "xi")
        gp = self._rate_prev(gradu_prev, B)
        dev = gradu.device
        ep = torch.tensor([int(e) for e in ep_index], dtype=torch.int32, device=dev)
        dC = torch.empty((len(ep_index), self.nx, B), dtype=torch.float64, device=dev)
        dS = torch.empty((len(ep_index), 6, B), dtype=torch.float64, device=dev)
        rc = self.L.cm_param_blocks(C.byref(self.desc), B, len(ep_index), _ptr(ep), _ptr(gradu), _ptr(gp), _ptr(xi_prev), _ptr(xi),
                                    _ptr(dC), _ptr(dS), self._stream())
        _lib.check(rc, "cm_param_blocks")
        return dC, dS

    def param_adjoint_history(self, ep_index, gradu_hist, xi_hist, lam_hist, sigma_bar_hist):
        """`cm_param_adjoint_history`: the extended parameters' share of the objective gradient, (n_ep,) device tensor."""
        torch = _torch()
        K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
        self._check_hist(gradu_hist, self.nu, K, B, "gradu_hist"); self._check_hist(xi_hist, self.nx, K, B, "xi_hist")
        self._check_hist(lam_hist, self.nx, K, B, "lam_hist"); self._check_hist(sigma_bar_hist, 6, K, B, "sigma_bar_hist")
        dev = gradu_hist.device
        n = len(ep_index)
        ep = torch.tensor([int(e) for e in ep_index], dtype=torch.int32, device=dev)
        g = torch.empty(n, dtype=torch.float64, device=dev)
        ws = torch.empty(max(1, B * n), dtype=torch.float64, device=dev)
        rc = self.L.cm_param_adjoint_history(C.byref(self.desc), B, K, n, _ptr(ep), _ptr(gradu_hist), _ptr(xi_hist), _ptr(lam_hist),
                                             _ptr(sigma_bar_hist), _ptr(g), _ptr(ws), ws.numel() * 8, self._stream())
        _lib.check(rc, "cm_param_adjoint_history")
        return g

    def adjoint_step(self, gradu, xi_prev, xi, data6, wsq6, hist_in, hist_out, out, accumulate=True, gradu_prev=None):
        B = gradu.shape[1]
        _check_soa(gradu, self.nu, B, "gradu"); _check_soa(xi_prev, self.nx, B, "xi_prev")
        _check_soa(xi, self.nx, B, "xi"); _check_soa(data6, 6, B, "data"); _check_soa(hist_out, self.nx, B, "hist_out")
        w = (C.c_double * 6)(*[float(v) for v in wsq6])
        ws, need = self._workspace(B, gradu.device)
        gp = self._rate_prev(gradu_prev, B)
        if gp is not None:
            rc = self.L.cm_adjoint_step_rate(C.byref(self.desc), B, _ptr(gradu), _ptr(gp), _ptr(xi_prev), _ptr(xi), _ptr(data6), w,
                                             _ptr(hist_in), _ptr(hist_out), _ptr(out), int(bool(accumulate)),
                                             _ptr(ws), need, self._stream())
            _lib.check(rc, "cm_adjoint_step_rate")
            return hist_out, out
        rc = self.L.cm_adjoint_step(C.byref(self.desc), B, _ptr(gradu), _ptr(xi_prev), _ptr(xi), _ptr(data6), w,
                                    _ptr(hist_in), _ptr(hist_out), _ptr(out), int(bool(accumulate)),
                                    _ptr(ws), need, self._stream())
        _lib.check(rc, "cm_adjoint_step")
        return hist_out, out
