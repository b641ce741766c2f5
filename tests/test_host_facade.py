"""CPU coverage of the facade's Python logic (Model block slicing, parameter chain rules, QoI derivatives, the
adjoint / direct / direct-adjoint objectives) with the device entry points re-routed to the host build of the
kernel arithmetic (tests/host_facade.py).  The same scenarios run on the GPU in tests/test_gpu_facade.py."""
import copy

import numpy as np
import pytest

from host_facade import HostSmallElasticPlastic
from problems import params_J2_voce, plane_stress_F


def _cauchy_history(model, F):
    from cmad_amd.models import mp_U_from_F, newton_solve
    n = F.shape[2] - 1
    cauchy = np.zeros((3, 3, n + 1))
    model.set_xi_to_init_vals()
    for step in range(1, n + 1):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        newton_solve(model)
        model.evaluate_cauchy()
        cauchy[:, :, step] = model.Sigma().copy()
        model.advance_xi()
    return cauchy


def _problem(active_elastic, K=8):
    from cmad_amd.models import DefType
    from cmad_amd.parameters import Parameters
    from cmad_amd.parameters.parameters import tree_map
    from cmad_amd.qois import Calibration
    params = params_J2_voce()
    if active_elastic:
        values = params.values
        flags = tree_map(lambda a: False, copy.deepcopy(values))
        flags["elastic"] = {"E": True, "nu": True}
        flags["plastic"]["flow stress"] = tree_map(lambda x: True, flags["plastic"]["flow stress"])
        tr = tree_map(lambda a: None, copy.deepcopy(values))
        tr["elastic"]["E"] = np.array([200e3])
        tr["plastic"]["flow stress"]["initial yield"]["Y"] = np.array([200.])
        tr["plastic"]["flow stress"]["hardening"]["voce"]["S"] = np.array([100., 300.])
        tr["plastic"]["flow stress"]["hardening"]["voce"]["D"] = np.array([10., 30.])
        params = Parameters(values, flags, tr)
    F = plane_stress_F(0.02, K // 2)
    model = HostSmallElasticPlastic(params, DefType.PLANE_STRESS)
    cauchy = _cauchy_history(model, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    data = cauchy + np.random.default_rng(22).normal(0., 2., cauchy.shape)
    qoi = Calibration(model, data, weight)
    model.parameters.set_active_values_from_flat(1.1 * model.parameters.flat_active_values(False), False)
    return model, qoi, F


@pytest.mark.parametrize("active_elastic", [False, True])
def test_gradients_and_hessian(active_elastic):
    from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective, MPDirectObjective
    model, qoi, F = _problem(active_elastic)
    x = model.parameters.flat_active_values(True)
    J, grad, H = MPDirectAdjointObjective(qoi, F).evaluate(x)
    Ja, ga = MPAdjointObjective(qoi, F).evaluate(x)
    Jd, gd = MPDirectObjective(qoi, F).evaluate(x)
    assert abs(J - Ja) <= 1e-12 * abs(J) and abs(J - Jd) <= 1e-12 * abs(J)
    np.testing.assert_allclose(ga, gd, rtol=1e-9, atol=1e-11 * np.abs(ga).max())
    np.testing.assert_allclose(grad, ga, rtol=1e-10, atol=1e-12 * np.abs(ga).max())
    np.testing.assert_allclose(H, H.T, rtol=1e-9, atol=1e-9 * np.abs(H).max())
    n, h = x.size, 1e-5
    H_fd = np.zeros((n, n))
    for k in range(n):
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        H_fd[:, k] = (MPAdjointObjective(qoi, F).evaluate(xp_).grad - MPAdjointObjective(qoi, F).evaluate(xm_).grad) / (2 * h)
    np.testing.assert_allclose(H, H_fd, rtol=5e-5, atol=5e-6 * np.abs(H).max())
    g_fd = np.zeros(n)
    for k in range(n):
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        g_fd[k] = (MPAdjointObjective(qoi, F).evaluate(xp_).J - MPAdjointObjective(qoi, F).evaluate(xm_).J) / (2 * h)
    np.testing.assert_allclose(ga, g_fd, rtol=1e-5, atol=1e-7 * np.abs(ga).max())


def test_named_derivatives_parallel_their_inputs():
    from problems import check_named_derivatives
    check_named_derivatives(HostSmallElasticPlastic)


@pytest.mark.parametrize("yield_kind,active_rotation", [("hosford", False), ("hill", True), ("hosford", True),
                                                        ("network", False), ("network deep", False)])
def test_gradient_of_extended_leaves(yield_kind, active_rotation):
    """Objective gradients w.r.t. the Hosford exponent and entries of the rotation matrix (reference: jacrev over the params
    pytree, cmad/models/model.py:125-153): adjoint == direct, both == central differences of the objective; and the DPARAMS
    Jacobian block of Model.evaluate() carries the same leaves."""
    from cmad_amd.objectives import MPAdjointObjective, MPDirectObjective
    from problems import extended_leaf_problem
    model, qoi, F = extended_leaf_problem(HostSmallElasticPlastic, yield_kind, active_rotation)
    x = model.parameters.flat_active_values(True)
    n_ext = {"hosford": 1, "hill": 0, "network": 1 + 6 + 5, "network deep": 1 + 6 + 4 + 12}[yield_kind] + (9 if active_rotation else 0)
    assert len(model.extended_active()) == n_ext
    Ja, ga = MPAdjointObjective(qoi, F).evaluate(x)
    Jd, gd = MPDirectObjective(qoi, F).evaluate(x)
    assert abs(Ja - Jd) <= 1e-12 * abs(Ja)
    np.testing.assert_allclose(ga, gd, rtol=1e-8, atol=1e-10 * np.abs(ga).max())
    g_fd = np.zeros_like(x)
    for k in range(x.size):
        h = 1e-6 * max(1.0, abs(x[k]))
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        g_fd[k] = (MPAdjointObjective(qoi, F).evaluate(xp_).J - MPAdjointObjective(qoi, F).evaluate(xm_).J) / (2 * h)
    np.testing.assert_allclose(ga, g_fd, rtol=2e-5, atol=1e-7 * np.abs(ga).max())
    # Model.evaluate() with seed_params: residual Jacobian columns of the same leaves vs central differences
    from cmad_amd.models import mp_U_from_F
    model.parameters.set_active_values_from_flat(x)
    model.set_xi_to_init_vals()
    model.gather_global(mp_U_from_F(F[:, :, 5]), mp_U_from_F(F[:, :, 4]))
    model._xi = [np.array([3e-4, 1e-4, 0., -2e-4, 0., -1e-4]), np.array([4e-4]), np.array([0.998])]
    model.seed_params(); model.evaluate()
    Jac = model.Jac().copy()
    for k in range(x.size):
        h = 1e-6 * max(1.0, abs(x[k]))
        cols = []
        for sgn in (1.0, -1.0):
            xk = x.copy(); xk[k] += sgn * h
            model.parameters.set_active_values_from_flat(xk)
            model.seed_none(); model.evaluate()
            cols.append(np.asarray(model.C()).copy())
        np.testing.assert_allclose(Jac[:, k], (cols[0] - cols[1]) / (2 * h), rtol=2e-5, atol=1e-9)
    model.parameters.set_active_values_from_flat(x)


def test_newton_solve_with_the_legacy_backtracking_is_one_device_solve():
    """`newton_solve(model, max_ls_evals=n)` (cmad/models/nonlinear_solver.py:55-81) runs as ONE `device_newton` call with the
    kernels' legacy line search, and converges to the states of the plain solve (the backtracking may change the iteration
    count, not the solution; iterate-by-iterate parity with the oracle's LS_LEGACY is in test_host_math.py)."""
    from cmad_amd.models import DefType, mp_U_from_F, newton_solve
    F = plane_stress_F(0.02, 4)
    out = []
    for n in (0, 5):
        model = HostSmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
        calls = []
        orig = model.device_newton
        model.device_newton = lambda *a, _o=orig, _c=calls, **k: (_c.append(k.get("line_search")), _o(*a, **k))[1]
        model.set_xi_to_init_vals()
        hist = []
        for step in range(1, F.shape[2]):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            it, res = newton_solve(model, max_ls_evals=n)
            hist.append((it, np.concatenate([np.atleast_1d(b) for b in model.xi()])))
            assert res < 1e-12
            model.advance_xi()
        assert len(calls) == F.shape[2] - 1                                     # one device solve per step, no host loop
        assert all(c == ({"max evals": 5, "kind": "legacy"} if n else None) for c in calls)
        out.append(hist)
    for (it0, x0), (it1, x1) in zip(*out):
        assert abs(it0 - it1) <= 2
        np.testing.assert_allclose(x1, x0, rtol=1e-9, atol=1e-13)


def test_direct_adjoint_hessian_of_the_uniaxial_calibration_qoi():
    """Second-order pass for a QoI with an explicit state term: UniaxialCalibration (axial stress + the two lateral stretches,
    per-step weights; cmad/qois/uniaxial_calibration.py:70-85, differentiated twice in the reference by hessian(qoi_fun),
    cmad/qois/qoi.py:47-57) on a UNIAXIAL_STRESS Hill model.  MPDirectAdjointObjective's gradient equals the adjoint one and
    its Hessian is symmetric and matches central differences of the adjoint gradient."""
    from cmad_amd.models import DefType, mp_U_from_F, newton_solve
    from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective
    from cmad_amd.qois import UniaxialCalibration
    K = 10
    F = np.repeat(np.eye(1)[:, :, None], K + 1, axis=2)
    F[0, 0, :] += np.linspace(0., 0.006, K + 1)
    model = HostSmallElasticPlastic(params_J2_voce(yield_kind="hill"), DefType.UNIAXIAL_STRESS, uniaxial_stress_idx=1)
    data = np.zeros((3, K + 1))
    model.set_xi_to_init_vals()
    for step in range(1, K + 1):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        newton_solve(model)
        model.evaluate_cauchy()
        data[:, step] = [model.Sigma()[1, 1], model.xi()[2][0] - 1., model.xi()[2][1] - 1.]
        model.advance_xi()
    weight = np.ones((3, K + 1)); weight[1:, :] = 1e4
    weight[:, 1::2] *= 0.5                                         # weights that change from step to step
    qoi = UniaxialCalibration(model, data, weight, uniaxial_stress_idx=1, stretch_var_idx=2)
    model.parameters.set_active_values_from_flat(1.1 * model.parameters.flat_active_values(False), False)
    x = model.parameters.flat_active_values(True)
    J, grad, H = MPDirectAdjointObjective(qoi, F).evaluate(x)
    Ja, ga = MPAdjointObjective(qoi, F).evaluate(x)
    assert abs(J - Ja) <= 1e-12 * abs(J) and J > 0
    np.testing.assert_allclose(grad, ga, rtol=1e-9, atol=1e-11 * np.abs(ga).max())
    np.testing.assert_allclose(H, H.T, rtol=1e-9, atol=1e-9 * np.abs(H).max())
    n, h = x.size, 1e-5
    H_fd = np.zeros((n, n))
    for k in range(n):
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        H_fd[:, k] = (MPAdjointObjective(qoi, F).evaluate(xp_).grad - MPAdjointObjective(qoi, F).evaluate(xm_).grad) / (2 * h)
    np.testing.assert_allclose(H, H_fd, rtol=5e-5, atol=5e-6 * np.abs(H).max())
    # the state term matters: without it the Hessian differs by more than the finite-difference check tolerates
    class StressOnly(UniaxialCalibration):
        def state_curvature(self):
            return None
    q2 = StressOnly(model, data, weight, uniaxial_stress_idx=1, stretch_var_idx=2)
    H2 = MPDirectAdjointObjective(q2, F).evaluate(x).hessian
    assert np.abs(H2 - H).max() > 1e-4 * np.abs(H).max()
    assert not np.allclose(H2, H_fd, rtol=5e-5, atol=5e-6 * np.abs(H).max())
    # the per-step form (reference QoI.evaluate_hessians, qoi.py:160-188) at the state of the last step, against central
    # differences of the QoI's own first derivatives
    model.parameters.set_active_values_from_flat(x)
    model.set_xi_to_init_vals()
    for step in range(1, K + 1):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        newton_solve(model)
        if step < K:
            model.advance_xi()
    qoi.evaluate_hessians(K)
    xi0 = [b.copy() for b in model.xi()]

    def dJ_dxi():
        model.seed_xi(); qoi.evaluate(K); model.seed_none()
        return np.asarray(qoi.dJ()).ravel().copy()
    n_xi = model.num_dofs
    fd = np.zeros((n_xi, n_xi))
    col = 0
    for blk in range(len(xi0)):
        for e in range(xi0[blk].size):
            hh_ = 1e-7 * max(1.0, abs(xi0[blk].ravel()[e]))
            g = []
            for sgn in (1.0, -1.0):
                model._xi = [b.copy() for b in xi0]
                model._xi[blk].ravel()[e] += sgn * hh_
                g.append(dJ_dxi())
            fd[:, col] = (g[0] - g[1]) / (2 * hh_)
            col += 1
    model._xi = [b.copy() for b in xi0]
    np.testing.assert_allclose(qoi.d2J_dxi2, fd, rtol=1e-5, atol=1e-6 * np.abs(fd).max())
    assert qoi.d2J_dxi_dparams.shape == (n_xi, x.size) and qoi.d2J_dparams2.shape == (x.size, x.size)
    assert np.isfinite(qoi.d2J_dxi_dparams).all() and np.isfinite(qoi.d2J_dparams2).all()


@pytest.mark.parametrize("hidden", [None, [4, 3]])
def test_network_hardening_law_through_the_model_api(hidden):
    """`HostSmallElasticPlastic(parameters, def_type, hardening_funs={"neural network": SimpleNeuralNetwork([1, 5, 1], ...).evaluate})`
    as examples/noisy_calibration.py:245-252 builds it: the adjoint and the direct objective agree, and the gradient w.r.t.
    the initial yield and EVERY network weight and bias (active leaves of the params pytree) matches central differences."""
    import copy
    from cmad_amd.models import DefType
    from cmad_amd.objectives import MPAdjointObjective, MPDirectObjective
    from cmad_amd.parameters import Parameters
    from cmad_amd.parameters.parameters import tree_map
    from cmad_amd.qois import Calibration
    import parity_cases as pc
    values, net, _ = pc.nn_hardening_values(hidden=hidden)              # widths [1, 5, 1] or [1, 4, 3, 1]
    flags = tree_map(lambda a: False, copy.deepcopy(values))
    flags["plastic"]["flow stress"] = tree_map(lambda a: True, flags["plastic"]["flow stress"])
    params = Parameters(values, flags, tree_map(lambda a: None, copy.deepcopy(values)))
    F = plane_stress_F(0.02, 3)
    model = HostSmallElasticPlastic(params, DefType.PLANE_STRESS, hardening_funs={"neural network": net.evaluate})
    cauchy = _cauchy_history(model, F)
    assert np.abs(cauchy).max() > 100.0
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    qoi = Calibration(model, cauchy + np.random.default_rng(3).normal(0., 2., cauchy.shape), weight)
    x = 1.05 * model.parameters.flat_active_values(True)
    n = x.size
    assert n == (1 + 3 * 5 + 1 if hidden is None else 1 + (4 + 4) + (12 + 3) + (3 + 1))    # Y, then every weight and bias
    ra = MPAdjointObjective(qoi, F).evaluate(x)
    rd = MPDirectObjective(qoi, F).evaluate(x)
    assert abs(ra.J - rd.J) <= 1e-12 * abs(ra.J)
    np.testing.assert_allclose(rd.grad, ra.grad, rtol=1e-8, atol=1e-10 * np.abs(ra.grad).max())
    for k in range(n):
        h = (1e-5 if hidden else 1e-6) * max(1.0, abs(x[k]))        # (small sensitivities through several sigmoid layers: wider step)
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        fd = (MPAdjointObjective(qoi, F).evaluate(xp_).J - MPAdjointObjective(qoi, F).evaluate(xm_).J) / (2 * h)
        np.testing.assert_allclose(ra.grad[k], fd, rtol=2e-4 if hidden else 5e-5, atol=(1e-6 if hidden else 1e-7) * np.abs(ra.grad).max(),
                                   err_msg=f"active parameter {k}")
    assert np.count_nonzero(np.abs(ra.grad) > 1e-9 * np.abs(ra.grad).max()) >= n - 1      # all but the output bias matter


@pytest.mark.parametrize("yield_kind,active_rotation", [("hosford", False), ("hill", True), ("network deep", False),
                                                        ("network wide", True)])      # (64 extended directions: the kernels' limit)
def test_direct_adjoint_hessian_with_extended_leaves(yield_kind, active_rotation):
    """Second-order sensitivities w.r.t. leaves outside the 12 native kernel parameters -- the Hosford exponent, the nine
    entries of the rotation matrix -- together with a native one (Y): the reference takes Hessians over the whole params pytree
    (cmad/models/model.py:133-147 under cmad/objectives/mp_objective.py:218-345).  MPDirectAdjointObjective's gradient equals the
    adjoint one; its Hessian is symmetric and matches central differences of the adjoint gradient."""
    from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective
    from problems import extended_leaf_problem
    model, qoi, F = extended_leaf_problem(HostSmallElasticPlastic, yield_kind, active_rotation)
    x = model.parameters.flat_active_values(True)
    J, grad, H = MPDirectAdjointObjective(qoi, F).evaluate(x)
    ra = MPAdjointObjective(qoi, F).evaluate(x)
    assert abs(J - ra.J) <= 1e-12 * abs(J)
    np.testing.assert_allclose(grad, ra.grad, rtol=1e-9, atol=1e-11 * np.abs(ra.grad).max())
    np.testing.assert_allclose(H, H.T, rtol=1e-8, atol=1e-9 * np.abs(H).max())
    n = x.size
    H_fd = np.zeros((n, n))
    for k in range(n):
        h = 1e-5 * max(1.0, abs(x[k]))
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        H_fd[:, k] = (MPAdjointObjective(qoi, F).evaluate(xp_).grad - MPAdjointObjective(qoi, F).evaluate(xm_).grad) / (2 * h)
    np.testing.assert_allclose(H, H_fd, rtol=2e-4, atol=2e-5 * np.abs(H).max())


@pytest.mark.parametrize("scale_params", [False, True])
def test_complex_step_model_instances(scale_params):
    """`Model(..., is_complex=True)`: tests/objectives/test_J2_fd_checks.py:301-386 (complex-step gradient and Hessian checks of
    the three material-point objectives) on the host build of cm::newton_cx; the GPU twin runs both model forms through
    cm_update_complex."""
    from problems import check_complex_step
    check_complex_step(HostSmallElasticPlastic, scale_params, num_pts_per_increment=12)


@pytest.mark.parametrize("def_type_name,yield_kind", [("UNIAXIAL_STRESS", "hill"), ("FULL_3D", "hosford"), ("PLANE_STRESS", "hill")])
def test_complex_step_instances_on_other_configurations(def_type_name, yield_kind):
    """Complex-step instances beyond the reference's J2 / PLANE_STRESS check: the stress at the end of a short history from a
    model built with is_complex=True equals the real model's, and Im sigma(p + i h d) / h (h = 1e-20) equals the central
    difference of the real model's stress along d."""
    from cmad_amd.models import DefType, mp_U_from_F, newton_solve
    dt = getattr(DefType, def_type_name)
    kw = {"uniaxial_stress_idx": 1} if dt == DefType.UNIAXIAL_STRESS else {}
    nd = {DefType.UNIAXIAL_STRESS: 1, DefType.FULL_3D: 3, DefType.PLANE_STRESS: 2}[dt]
    a = 1 if dt == DefType.UNIAXIAL_STRESS else 0
    K = 6
    F = np.repeat(np.eye(nd)[:, :, None], K + 1, axis=2)
    F[0, 0, :] += np.linspace(0., 0.006, K + 1)

    def final_stress(model):
        model.set_xi_to_init_vals()
        for step in range(1, K + 1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            newton_solve(model)
            model.evaluate_cauchy()
            model.advance_xi()
        return model.Sigma()[a, a]

    real = HostSmallElasticPlastic(params_J2_voce(yield_kind=yield_kind), dt, **kw)
    cplx = HostSmallElasticPlastic(params_J2_voce(yield_kind=yield_kind), dt, is_complex=True, **kw)
    x = real.parameters.flat_active_values(True)
    d, h, eps = np.array([0.3, -0.7, 0.5])[:x.size], 1e-20, 1e-6
    cplx.parameters.set_active_values_from_flat(x.astype(complex) + 1j * h * d, is_complex=True)
    s_real, s_cplx = final_stress(real), final_stress(cplx)
    assert abs(s_cplx.real - s_real) <= 1e-12 * abs(s_real)
    fd = []
    for sgn in (1.0, -1.0):
        real.parameters.set_active_values_from_flat(x + sgn * eps * d)
        fd.append(final_stress(real))
    np.testing.assert_allclose(s_cplx.imag / h, (fd[0] - fd[1]) / (2 * eps), rtol=1e-6)


def test_hardening_funs_is_the_reference_lookup_table():
    """`hardening_funs` is the table `combined_hardening_fun` indexes (cmad/models/hardening.py:22-34): passing
    `get_hardening_funs()` (the Voce / linear laws the kernels have built in), with or without a "neural network" entry, is
    valid; only a custom callable has no kernel."""
    import copy
    from cmad_amd.models import DefType
    from cmad_amd.models.hardening import combined_hardening_fun, get_hardening_funs
    from cmad_amd.neural_networks.simple_neural_network import SimpleNeuralNetwork, hardening_network_scales
    from cmad_amd.parameters import Parameters
    from cmad_amd.parameters.parameters import tree_map
    from cmad_amd.synthetic import j2_voce_values
    table = get_hardening_funs()
    assert set(table) == {"voce", "linear"}
    assert abs(combined_hardening_fun(0.01, {"voce": {"S": 200., "D": 20.}, "linear": {"K": 50.}}, table)
               - (200. * (1. - np.exp(-0.2)) + 0.5)) < 1e-13
    assert hardening_network_scales(table) is None
    net = SimpleNeuralNetwork([1, 4, 1], input_scale=3., output_scale=7.)
    assert hardening_network_scales(dict(table, **{"neural network": net.evaluate})) == (3., 7.)
    with pytest.raises(NotImplementedError):
        hardening_network_scales({"voce": lambda alpha, p: 0. * alpha})
    values = j2_voce_values()
    params = Parameters(values, tree_map(lambda a: False, copy.deepcopy(values)), tree_map(lambda a: None, copy.deepcopy(values)))
    model = HostSmallElasticPlastic(params, DefType.FULL_3D, hardening_funs=table)          # the built-in laws: no network entry needed
    assert model._hardening_nn is None


def test_complex_perturbations_of_every_leaf_are_carried():
    """`cm_update_complex` takes the imaginary parts of the 12 native kernel parameters by value and those of every other leaf
    (rotation matrix, Barlat coefficients, network weights) in the extended array, indexed like `leaf_ep_index`: nothing is
    dropped on the way (a dropped perturbation would read as a zero directional derivative)."""
    import copy
    from cmad_amd.models.device import EP_Q0, EP_YC6, build_desc, complex_parameter_parts, real_tree
    from cmad_amd.parameters import Parameters
    from cmad_amd.parameters.parameters import tree_map
    from cmad_amd.synthetic import j2_voce_values
    values = copy.deepcopy(j2_voce_values())
    values["plastic"]["flow stress"]["initial yield"]["Y"] = 200. + 1e-20j
    values["rotation matrix"] = np.eye(3) + 1e-20j * np.arange(9.).reshape(3, 3)
    real = real_tree(values)                                # (the tree's key paths do not depend on the dtype of its leaves)
    params = Parameters(real, tree_map(lambda a: False, copy.deepcopy(real)), tree_map(lambda a: None, copy.deepcopy(real)))
    _, info = build_desc(real)
    p_im, ext = complex_parameter_parts(values, params.flat_paths(), info)
    assert p_im[2] == 1e-20 and p_im[0] == 0.0
    assert ext.shape == (22,) and not ext[:EP_Q0 - EP_YC6].any()
    np.testing.assert_array_equal(ext[EP_Q0 - EP_YC6:], 1e-20 * np.arange(9.))
    values["rotation matrix"] = np.eye(3)
    assert complex_parameter_parts(values, params.flat_paths(), info)[1] is None           # nothing beyond the native 12: no array


@pytest.mark.parametrize("kind", ["barlat", "network", "network scaled deep", "hardening network deep"])
def test_complex_step_instances_on_dense_surfaces_and_extended_leaves(kind):
    from problems import check_complex_step_extended
    check_complex_step_extended(HostSmallElasticPlastic, kind)
