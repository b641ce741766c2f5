"""Test problems shared by the facade tests: the reference's J2AnalyticalProblem parameter trees
(/root/reference/tests/support/test_problems.py:9-162) built with cmad_amd.parameters.Parameters."""
import copy

import numpy as np

from cmad_amd.parameters import Parameters
from cmad_amd.parameters.parameters import tree_map


def params_J2_voce(flat_param_values=(200e3, 0.3, 200., 200., 20.), scale_params=True, yield_kind="J2"):
    E, nu, Y, S, D = flat_param_values
    if yield_kind == "J2":
        eff = {"J2": 0.}
    elif yield_kind == "hill":
        eff = {"hill": dict(zip("FGHLMN", [0.5] * 6))}
    else:
        eff = {"hosford": {"a": 4.}}
    values = {
        "rotation matrix": np.eye(3),
        "elastic": {"E": E, "nu": nu},
        "plastic": {"effective stress": eff,
                    "flow stress": {"initial yield": {"Y": Y}, "hardening": {"voce": {"S": S, "D": D}}}}}
    flags = tree_map(lambda a: False, copy.deepcopy(values))
    flags["plastic"]["flow stress"] = tree_map(lambda x: True, flags["plastic"]["flow stress"])
    transforms = tree_map(lambda a: None, copy.deepcopy(values))
    if scale_params:
        fs = transforms["plastic"]["flow stress"]
        fs["initial yield"]["Y"] = np.array([200.])
        fs["hardening"]["voce"]["S"] = np.array([100., 300.])
        fs["hardening"]["voce"]["D"] = np.array([10., 30.])
    return Parameters(values, flags, transforms)


def plane_stress_F(strain_increment=0.02, num_pts_per_increment=50):
    """tests/objectives/test_J2_fd_checks.py:266-289 (0.02) / test_calibrations.py:60-80 (0.01)."""
    init = strain_increment / num_pts_per_increment
    eps_xx = np.r_[np.zeros(1), np.linspace(init, strain_increment, num_pts_per_increment),
                   np.ones(num_pts_per_increment) * strain_increment]
    eps_yy = np.r_[np.zeros(1), np.zeros(num_pts_per_increment), np.linspace(init, strain_increment, num_pts_per_increment)]
    n = 2 * num_pts_per_increment
    F = np.repeat(np.eye(2)[:, :, None], n + 1, axis=2)
    F[0, 0, :] += eps_xx[:n + 1]
    F[1, 1, :] += eps_yy[:n + 1]
    return F


def extended_leaf_problem(model_cls, yield_kind, active_rotation, is_complex=False):
    """PLANE_STRESS calibration history whose active leaves lie outside the 12 native kernel parameters: the Hosford exponent
    and/or the rotation matrix (differentiated by forward-mode evaluation of the whole model, cm_param_blocks)."""
    from cmad_amd.models import DefType
    from cmad_amd.qois import Calibration
    network = yield_kind.startswith("network")
    kw = {}
    if network:
        # hybrid Hill + ICNN surface, one ("network") or two ("network deep") hidden layers; active: Y, one Hill coefficient,
        # the input weights of the last layer and (deep) the weights between the hidden layers
        from cmad_amd.models import HybridHillEffectiveStress
        from cmad_amd.models.device import ScaledHybridHillEffectiveStress
        from cmad_amd.synthetic import al7079_hybrid_setup
        # "network wide": 42 + 7 + 6 network entries + the rotation matrix = 64 extended directions, the second-order pass's limit
        widths = (6, 4, 3, 1) if yield_kind.endswith("deep") else ((6, 7, 1) if yield_kind.endswith("wide") else (6, 5, 1))
        icnn, values = al7079_hybrid_setup(widths)
        values = copy.deepcopy(values)
        values["plastic"]["flow stress"]["initial yield"]["Y"] = 200.0
        kw = {"effective_stress_fun": ScaledHybridHillEffectiveStress(icnn, 525.0) if "scaled" in yield_kind else HybridHillEffectiveStress(icnn)}
    else:
        base = params_J2_voce(yield_kind=yield_kind, scale_params=False)
        values = copy.deepcopy(base.values)
    th = 0.35
    values["rotation matrix"] = np.array([[np.cos(th), -np.sin(th), 0.], [np.sin(th), np.cos(th), 0.], [0., 0., 1.]])
    if yield_kind == "hill":
        values["plastic"]["effective stress"]["hill"] = dict(zip("FGHLMN", [0.35, 0.55, 0.6, 1.4, 1.6, 1.7]))
    flags = tree_map(lambda a: False, copy.deepcopy(values))
    flags["plastic"]["flow stress"]["initial yield"]["Y"] = True
    if yield_kind == "hosford":
        flags["plastic"]["effective stress"]["hosford"]["a"] = True
    if active_rotation:
        flags["rotation matrix"] = True                        # an array leaf is active as a whole: nine entries
    if network:
        nn = flags["plastic"]["effective stress"]["neural network"]
        flags["plastic"]["effective stress"]["hill"]["G"] = not yield_kind.endswith("wide")     # (an extended leaf on this surface)
        nn["x params"][-1]["weights"] = True
        nn["x params"][0]["biases"] = True
        if nn["z params"][:-1]:
            nn["z params"][0]["weights"] = True
        if yield_kind.endswith("wide"):
            nn["x params"][0]["weights"] = True
    params = Parameters(values, flags, tree_map(lambda a: None, copy.deepcopy(values)))
    F = plane_stress_F(0.02, 3)
    if is_complex:
        kw["is_complex"] = True
    model = model_cls(params, DefType.PLANE_STRESS, **kw)
    n = F.shape[2]
    t = np.linspace(0.0, 1.0, n)
    data = np.zeros((3, 3, n))
    data[0, 0] = 260.0 * np.tanh(4 * t); data[1, 1] = 180.0 * t; data[0, 1] = data[1, 0] = 15.0 * np.sin(3 * t)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.; weight[0, 1] = weight[1, 0] = 0.7
    return model, Calibration(model, data.astype(complex) if is_complex else data, weight), F


def complex_step_problem(model_cls, kind, is_complex):
    """(model, qoi, F) of the complex-step checks beyond the reference's J2 case: "barlat" (active: Y, two coefficients in the
    native yc slots' range and beyond it, the exponent), "hardening network" / "hardening network deep" (every weight and bias
    of the network hardening law), or one of `extended_leaf_problem`'s network surfaces with the rotation matrix active."""
    from cmad_amd.models import DefType
    from cmad_amd.qois import Calibration
    if not (kind == "barlat" or kind.startswith("hardening network")):
        return extended_leaf_problem(model_cls, kind, True, is_complex=is_complex)
    import parity_cases as pc
    kw = {"is_complex": True} if is_complex else {}
    if kind == "barlat":
        from cmad_amd.models.device import BARLAT_NAMES
        import oracle_lib as ol
        values = ol.j2_voce_values()
        values["plastic"]["effective stress"] = {"barlat": dict(zip(BARLAT_NAMES, pc.AL7079_BARLAT[:18] + [8.0]))}
        th = 0.35
        values["rotation matrix"] = np.array([[np.cos(th), -np.sin(th), 0.], [np.sin(th), np.cos(th), 0.], [0., 0., 1.]])
        flags = tree_map(lambda a: False, copy.deepcopy(values))
        flags["plastic"]["flow stress"]["initial yield"]["Y"] = True
        for name in ("sp_13", "dp_23", "dp_66", "a"):
            flags["plastic"]["effective stress"]["barlat"][name] = True
    else:
        values, net, _ = pc.nn_hardening_values(hidden=[4, 3] if kind.endswith("deep") else None)
        kw["hardening_funs"] = {"neural network": net.evaluate}
        flags = tree_map(lambda a: False, copy.deepcopy(values))
        flags["plastic"]["flow stress"] = tree_map(lambda a: True, flags["plastic"]["flow stress"])
    params = Parameters(values, flags, tree_map(lambda a: None, copy.deepcopy(values)))
    F = plane_stress_F(0.02, 3)
    model = model_cls(params, DefType.PLANE_STRESS, **kw)
    n = F.shape[2]
    t = np.linspace(0.0, 1.0, n)
    data = np.zeros((3, 3, n))
    data[0, 0] = 260.0 * np.tanh(4 * t); data[1, 1] = 180.0 * t; data[0, 1] = data[1, 0] = 15.0 * np.sin(3 * t)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.; weight[0, 1] = weight[1, 0] = 0.7
    return model, Calibration(model, data.astype(complex) if is_complex else data, weight), F


def check_complex_step_extended(model_cls, kind):
    """`Model(..., is_complex=True)` on the configurations the reference's own complex-step test does not reach (its
    `is_complex` is a constructor flag of every configuration, cmad/models/small_elastic_plastic.py:90,118-127): Barlat, the
    network surfaces, the network hardening law, with perturbed leaves outside the 12 native kernel parameters (Barlat
    coefficients, rotation matrix, network weights).  Im J(p + i h d) / h at h = 1e-20 -- no derivative code at all -- equals
    d . grad J of the REAL model's adjoint objective to round-off, and Re J is the real objective."""
    from cmad_amd.models import mp_U_from_F, newton_solve
    from cmad_amd.objectives import MPAdjointObjective
    model, qoi, F = complex_step_problem(model_cls, kind, False)
    model_c, qoi_c, _ = complex_step_problem(model_cls, kind, True)
    assert model_c.dtype is complex
    x = model.parameters.flat_active_values(True)
    ra = MPAdjointObjective(qoi, F).evaluate(x)

    def J_complex(flat_complex):
        model_c.parameters.set_active_values_from_flat(flat_complex, is_complex=True)
        model_c.set_xi_to_init_vals()
        acc = 0.0 + 0.0j
        for step in range(1, F.shape[2]):
            model_c.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            newton_solve(model_c, max_iters=30)
            qoi_c.evaluate(step)
            acc = acc + qoi_c.J()
            model_c.advance_xi()
        return complex(acc)

    J0 = J_complex(x.astype(complex))
    assert J0.imag == 0.0 and abs(J0.real - ra.J) <= 1e-9 * abs(ra.J)
    rng = np.random.default_rng(22)
    h = 1e-20
    for _ in range(2):
        d = rng.uniform(-1.0, 1.0, size=x.size)
        Jc = J_complex(x.astype(complex) + 1j * h * d)
        assert abs(Jc.real - ra.J) <= 1e-9 * abs(ra.J)
        np.testing.assert_allclose(Jc.imag / h, d @ ra.grad, rtol=1e-7, atol=1e-9 * np.abs(ra.grad).max())
    # one leaf at a time: every active entry's perturbation reaches the kernel (none is dropped on the way)
    for k in range(x.size):
        e = np.zeros(x.size); e[k] = 1.0
        np.testing.assert_allclose(J_complex(x.astype(complex) + 1j * h * e).imag / h, ra.grad[k], rtol=1e-6,
                                   atol=1e-9 * np.abs(ra.grad).max(), err_msg=f"active parameter {k}")


def _same_structure(a, b):
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a) == list(b) and all(_same_structure(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and all(_same_structure(x, y) for x, y in zip(a, b))
    return not isinstance(b, (dict, list, tuple))


def check_named_derivatives(ModelClass):
    """The reference's tests/models/test_abc_contract.py:31-65: dC_dxi / dC_dxi_prev / dC_dp / dC_dU / dC_dU_prev
    return trees parallel to xi / xi_prev / params / U / U_prev -- plus the values against finite differences."""
    from cmad_amd.models import DefType, mp_U_from_F
    model = ModelClass(params_J2_voce(), DefType.FULL_3D)
    model.set_xi_to_init_vals()
    F = np.eye(3) + 0.004 * np.diag([1.0, -0.4, -0.3])                     # plastic step
    model.gather_global(mp_U_from_F(F), mp_U_from_F(np.eye(3)))
    xi, xi_prev, params, U, U_prev = model.variables()
    xi = [xi[0] + 1e-4 * np.array([1.0, 0.2, 0.0, -0.5, 0.1, -0.5]), xi[1] + 1e-3]
    assert _same_structure(model.dC_dxi(xi, xi_prev, params, U, U_prev), xi)
    assert _same_structure(model.dC_dxi_prev(xi, xi_prev, params, U, U_prev), xi_prev)
    dU = model.dC_dU(xi, xi_prev, params, U, U_prev)
    assert set(dU.fields) == set(U.fields) and set(dU.grad_fields) == set(U.grad_fields)
    dUp = model.dC_dU_prev(xi, xi_prev, params, U, U_prev)
    assert dUp.grad_fields["u"].shape == (7, 3, 3) and not dUp.grad_fields["u"].any()
    dp = model.dC_dp(xi, xi_prev, params, U, U_prev)
    assert _same_structure(dp, params)
    assert dp["rotation matrix"].shape == (7, 3, 3) and np.isfinite(dp["rotation matrix"]).all()
    assert dp["plastic"]["effective stress"]["J2"].shape == (7,) and not dp["plastic"]["effective stress"]["J2"].any()
    # d C / d Y and d C / d E against central differences of the residual
    import copy
    for path, h in ((("plastic", "flow stress", "initial yield", "Y"), 1e-4), (("elastic", "E"), 1e-1),
                    (("rotation matrix", (0, 1)), 1e-6), (("rotation matrix", (2, 2)), 1e-6)):
        elem = path[-1] if isinstance(path[-1], tuple) else None       # entry of an array leaf (rotation matrix: through cm_param_blocks)
        if elem is not None:
            path = path[:-1]

        def C_at(delta):
            p2 = copy.deepcopy(params)
            node = p2
            for k in path[:-1]:
                node = node[k]
            if elem is not None:
                node[path[-1]] = np.array(node[path[-1]], dtype=np.float64)
                node[path[-1]][elem] += delta
            else:
                node[path[-1]] = node[path[-1]] + delta
            r = model._residual(xi, xi_prev, p2, U, U_prev)
            return np.concatenate([np.atleast_1d(b) for b in r]) if isinstance(r, (list, tuple)) else np.asarray(r)
        fd = (C_at(h) - C_at(-h)) / (2 * h)
        leaf = dp
        for k in path:
            leaf = leaf[k]
        if elem is not None:
            leaf = leaf[(slice(None),) + elem]
        np.testing.assert_allclose(leaf, fd, rtol=1e-6, atol=1e-12)


def check_complex_step(ModelCls, scale_params, num_pts_per_increment=50):
    """The complex-step half of the reference's tests/objectives/test_J2_fd_checks.py:301-386 (plane_stress_fd_checks): a
    second model instance built with is_complex=True evaluates the objective at p + i h d (gradient: :163-197) and at
    p + h d e^{+-i pi/3} (Hessian: :200-235) by stepping through the history with newton_solve, and the AD-free directional
    derivatives must approach those of MPDirectObjective / MPAdjointObjective / MPDirectAdjointObjective of the REAL model
    with an error that drops by more than five decades over h = 1 ... 1e-9."""
    from cmad_amd.models import DefType, mp_U_from_F, newton_solve
    from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective, MPDirectObjective
    from cmad_amd.qois import Calibration

    def run_history(model, per_step):
        model.set_xi_to_init_vals()
        for step in range(1, F.shape[2]):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            model.seed_xi()
            newton_solve(model)
            model.seed_none()
            per_step(step)
            model.advance_xi()

    F = plane_stress_F(0.02, num_pts_per_increment)
    model = ModelCls(params_J2_voce(scale_params=scale_params), DefType.PLANE_STRESS, is_complex=False)
    cauchy = np.zeros((3, 3, F.shape[2]))

    def store(step):
        model.evaluate_cauchy()
        cauchy[:, :, step] = model.Sigma()
    run_history(model, store)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    qoi = Calibration(model, cauchy, weight)
    model_c = ModelCls(params_J2_voce(scale_params=scale_params), DefType.PLANE_STRESS, is_complex=True)
    qoi_c = Calibration(model_c, cauchy.astype(complex), weight)
    assert model_c.dtype is complex

    def J_complex(flat_complex):
        model_c.parameters.set_active_values_from_flat(flat_complex, is_complex=True)
        acc = [0.0 + 0.0j]

        def add(step):
            qoi_c.evaluate(step)
            acc[0] = acc[0] + qoi_c.J()
        run_history(model_c, add)
        return complex(acc[0])

    # evaluate away from the values the data came from, like the reference
    offset = 1.1 * model.parameters.flat_active_values(False)
    model.parameters.set_active_values_from_flat(offset, False)
    model_c.parameters.set_active_values_from_flat(offset.astype(complex), False, is_complex=True)
    x = model.parameters.flat_active_values(True)
    x_c = model_c.parameters.flat_active_values(True).astype(complex)
    np.testing.assert_allclose(x_c.real, x, rtol=1e-14)
    _, g_direct = MPDirectObjective(qoi, F).evaluate(x)
    _, g_adjoint = MPAdjointObjective(qoi, F).evaluate(x)
    J_ref, _, H = MPDirectAdjointObjective(qoi, F).evaluate(x)
    np.random.seed(22)
    d = np.random.uniform(low=-1.0, size=x.size)
    hs = np.logspace(0, -9, 10)
    # with no perturbation the complex instance reproduces the real objective
    J0 = J_complex(x_c)
    assert abs(J0.imag) == 0.0 and abs(J0.real - J_ref) <= 1e-9 * max(1.0, abs(J_ref))
    err_d, err_a, err_h = [], [], []
    for h in hs:
        dd = J_complex(x_c + 1j * h * d).imag / h
        err_d.append(abs(dd - d @ g_direct)); err_a.append(abs(dd - d @ g_adjoint))
        J1 = J_complex(x_c + complex(0.5, np.sqrt(3.) / 2.) * h * d)
        J2 = J_complex(x_c + complex(-0.5, -np.sqrt(3.) / 2.) * h * d)
        err_h.append(abs((J1 + J2).imag / (np.sqrt(3.) / 2. * h ** 2) - d @ H @ d))
    err_d, err_a, err_h = np.array(err_d), np.array(err_a), np.array(err_h)
    assert np.isfinite(err_d).all() and np.isfinite(err_h).all()
    assert np.allclose(err_d, err_a)
    tiny = 1e-300
    assert np.log10(err_d.max() / max(err_d.min(), tiny)) > 5.0
    assert np.log10(err_h.max() / max(err_h.min(), tiny)) > 5.0
    # the complex step has no subtractive cancellation: at h = 1e-9 the directional derivative is exact to round-off
    assert err_d[-1] <= 1e-9 * max(1.0, abs(d @ g_direct))
    model_c.parameters.set_active_values_from_flat(x_c, is_complex=True)
