"""The reference's own tests, re-run through the cmad_amd facade (`-m gpu`; every model/qoi evaluation is a
HIP launch):
  * tests/models/test_elastic_plastic_models.py:15-125   analytical J2+Voce fields, J2/Hill/Hosford models
  * tests/objectives/test_J2_fd_checks.py:303-386        direct == adjoint, FD error drops > 5 decades
  * tests/objectives/test_calibrations.py:57-109         L-BFGS-B recovers [D, S, Y] to 1e-7
  * tests/global_residuals/test_for_model_coupled.py:231-295  local equilibrium + IFT tangent vs FD
plus the batched objective against the one-point-per-call objectives and the oracle."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from problems import params_J2_voce, plane_stress_F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _models():
    from cmad_amd.models import DefType, SmallElasticPlastic
    return DefType, SmallElasticPlastic


@pytest.mark.parametrize("def_name", ["FULL_3D", "PLANE_STRESS"])
@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
def test_models_reproduce_analytical_fields(golden_dir, def_name, yield_kind):
    from cmad_amd.models import mp_U_from_F, newton_solve
    from cmad_amd.qois import Calibration
    DefType, SmallElasticPlastic = _models()
    def_type = getattr(DefType, def_name)
    nd = 3 if def_type == DefType.FULL_3D else 2
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    for name in ("uniaxial", "biaxial"):
        stress, strain, alpha, mask = g[f"{name}_stress"], g[f"{name}_strain"], g[f"{name}_alpha"], g[f"{name}_mask"]
        num_steps = 100
        F = np.repeat(np.eye(nd)[:, :, None], num_steps + 1, axis=2)
        F[:, :, 1:] += strain[:nd, :nd, :]
        model = SmallElasticPlastic(params_J2_voce(yield_kind=yield_kind), def_type)
        weight = np.abs(mask)
        cauchy = np.zeros((3, 3, num_steps + 1))
        qoi = Calibration(model, cauchy.copy(), weight)
        J = 0.
        alphas = []
        model.set_xi_to_init_vals()
        for step in range(1, num_steps + 1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            newton_solve(model)
            alphas.append(model.xi()[1][0])
            model.seed_none()
            qoi.evaluate(step)
            J += qoi.J()
            model.evaluate_cauchy()
            cauchy[:, :, step] = model.Sigma().copy()
            model.advance_xi()
        tol = 1e-6
        assert np.linalg.norm(np.array(alphas) - alpha) < tol
        assert np.linalg.norm(cauchy[:, :, 1:] - stress) < tol
        assert abs(J - 0.5 * np.linalg.norm(weight[:, :, None] * cauchy) ** 2) < tol


def _compute_cauchy(model, F):
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


def _J_only(qoi, F):
    from cmad_amd.models import mp_U_from_F, newton_solve
    model = qoi.model()
    model.set_xi_to_init_vals()
    J = 0.
    for step in range(1, F.shape[2]):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        newton_solve(model)
        model.seed_none()
        qoi.evaluate(step)
        J += qoi.J()
        model.advance_xi()
    return float(J)


@pytest.mark.parametrize("rate,scale_params", [(False, False), (True, True)])
def test_direct_equals_adjoint_and_fd_error_drops(rate, scale_params):
    """tests/objectives/test_J2_fd_checks.py:303-352, 390-396: Models = [SmallElasticPlastic,
    SmallRateElasticPlastic] with scale_params = [False, True]."""
    from cmad_amd.models import SmallRateElasticPlastic
    from cmad_amd.objectives import MPAdjointObjective, MPDirectObjective
    from cmad_amd.qois import Calibration
    DefType, SmallElasticPlastic = _models()
    F = plane_stress_F(0.02, 10)                       # 20 steps (reference uses 100; same two-leg path)
    model = (SmallRateElasticPlastic if rate else SmallElasticPlastic)(params_J2_voce(scale_params=scale_params),
                                                                      DefType.PLANE_STRESS)
    cauchy = _compute_cauchy(model, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    qoi = Calibration(model, cauchy, weight)
    true_vals = model.parameters.flat_active_values(False)
    model.parameters.set_active_values_from_flat(1.1 * true_vals, False)
    x = model.parameters.flat_active_values(True)
    Jd, gd = MPDirectObjective(qoi, F).evaluate(x)
    Ja, ga = MPAdjointObjective(qoi, F).evaluate(x)
    assert abs(Jd - Ja) <= 1e-12 * abs(Jd)
    np.testing.assert_allclose(gd, ga, rtol=1e-9, atol=1e-10 * np.abs(ga).max())
    rng = np.random.default_rng(22)
    d = rng.uniform(-1., 1., size=x.size)
    errs = []
    for h in np.logspace(0, -9, 10):                   # the reference's perturbation ladder (:335)
        model.parameters.set_active_values_from_flat(x + h * d)
        Jp = _J_only(qoi, F)
        model.parameters.set_active_values_from_flat(x - h * d)
        Jm = _J_only(qoi, F)
        errs.append(abs((Jp - Jm) / (2 * h) - ga @ d))
    assert np.log10(max(errs) / min(errs)) > 5.0       # reference: error_drop_tol = 5 decades


def test_batched_objective_matches_pointwise_and_oracle():
    """B copies of one history with per-point data: the batched kernels == sum of one-point objectives."""
    import torch
    from cmad_amd.objectives import BatchedCalibrationObjective, MPAdjointObjective
    from cmad_amd.qois import Calibration
    DefType, SmallElasticPlastic = _models()
    F = plane_stress_F(0.02, 4)                        # 8 steps
    K = F.shape[2] - 1
    params = params_J2_voce()
    model = SmallElasticPlastic(params, DefType.PLANE_STRESS)
    cauchy = _compute_cauchy(model, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    rng = np.random.default_rng(22)
    B = 3
    datas = [cauchy + rng.normal(0., 5., cauchy.shape) for _ in range(B)]
    datas = [0.5 * (d + d.transpose(1, 0, 2)) for d in datas]
    model.parameters.set_active_values_from_flat(1.05 * model.parameters.flat_active_values(False), False)
    x = model.parameters.flat_active_values(True)
    J_ref, g_ref = 0., 0.
    for d in datas:
        r = MPAdjointObjective(Calibration(model, d, weight), F).evaluate(x)
        J_ref += r.J; g_ref = g_ref + r.grad
    gh = torch.from_numpy(np.stack([np.tile((F[:, :, k] - np.eye(2)).reshape(4, 1), (1, B)) for k in range(K + 1)])).cuda()
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    dh = torch.from_numpy(np.stack([np.stack([[d[i, j, k] for d in datas] for i, j in V6]) for k in range(K + 1)])).cuda()
    obj = BatchedCalibrationObjective(model, gh.contiguous(), dh.contiguous(), weight)
    r = obj.evaluate(x)
    np.testing.assert_allclose(r.J, J_ref, rtol=1e-11)
    np.testing.assert_allclose(r.grad, g_ref, rtol=1e-8, atol=1e-10 * np.abs(g_ref).max())
    # one launch per step and direction instead of cm_objective_grad_history: same numbers
    r2 = BatchedCalibrationObjective(model, gh.contiguous(), dh.contiguous(), weight, fused_history=False).evaluate(x)
    np.testing.assert_allclose(r2.J, r.J, rtol=1e-13)
    np.testing.assert_allclose(r2.grad, r.grad, rtol=1e-10, atol=1e-12 * np.abs(r.grad).max())


def test_calibration_recovers_truth():
    """L-BFGS-B on the batched objective from canonical 0.1 recovers [D, S, Y] = [20, 200, 200]."""
    import torch
    from scipy.optimize import fmin_l_bfgs_b
    from cmad_amd.objectives import BatchedCalibrationObjective
    DefType, SmallElasticPlastic = _models()
    F = plane_stress_F(0.01, 25)                       # 50 steps
    K = F.shape[2] - 1
    model = SmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
    true_params = model.parameters.flat_active_values()
    cauchy = _compute_cauchy(model, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    gh = torch.from_numpy(np.stack([(F[:, :, k] - np.eye(2)).reshape(4, 1) for k in range(K + 1)])).cuda().contiguous()
    dh = torch.from_numpy(np.stack([np.array([[cauchy[i, j, k]] for i, j in V6]) for k in range(K + 1)])).cuda().contiguous()
    obj = BatchedCalibrationObjective(model, gh, dh, weight)
    fun = lambda x: tuple(obj.evaluate(x))
    opt, fval, info = fmin_l_bfgs_b(fun, 0.1 * np.ones(3), bounds=model.parameters.opt_bounds, factr=10)
    model.parameters.set_active_values_from_flat(opt)
    assert np.linalg.norm(model.parameters.flat_active_values() - true_params) < 1e-7


def test_local_equilibrium_and_ift_tangent_vs_fd():
    """tests/global_residuals/test_for_model_coupled.py:65-82,231-295: fixed plastic point
    U[1,0]=0.005, U[2,1]=0.003, U[3,2]=0.002 -> ||C(xi*)|| < 1e-10 and IFT tangent == central FD."""
    import torch
    from cmad_amd.models import GlobalFieldsAtPoint, NewtonSettings, make_newton_solve
    DefType, SmallElasticPlastic = _models()
    model = SmallElasticPlastic(params_J2_voce(scale_params=False), DefType.FULL_3D)
    G = np.zeros((3, 3)); G[0, 0] = 0.005; G[1, 1] = 0.003; G[2, 2] = 0.002
    U = GlobalFieldsAtPoint(fields={"u": np.zeros(3)}, grad_fields={"u": G})
    solve = make_newton_solve(model._residual, max_iters=20, abs_tol=1e-12, rel_tol=1e-12)
    xi0 = [b.copy() for b in model._init_xi]
    xi = solve(xi0, model.parameters.values, U, U)
    assert xi[1][0] > 0.0
    C = model._residual(xi, xi0, model.parameters.values, U, U)
    assert np.linalg.norm(C) < 1e-10
    st = NewtonSettings.traced(max_iters=20, abs_tol=1e-12, rel_tol=1e-12)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 1)).cuda()
    xp = t(np.zeros(7))
    _, sig, _, ds = model.update_tangent_batch(t(G), xp, st)
    ds = ds.cpu().numpy()[:, :, 0]
    h = 1e-7
    fd = np.zeros((6, 9))
    for c in range(9):
        Gp, Gm = G.reshape(9).copy(), G.reshape(9).copy()
        Gp[c] += h; Gm[c] -= h
        sp = model.update_batch(t(Gp), xp, st)[1].cpu().numpy()[:, 0]
        sm = model.update_batch(t(Gm), xp, st)[1].cpu().numpy()[:, 0]
        fd[:, c] = (sp - sm) / (2 * h)
    np.testing.assert_allclose(ds, fd, rtol=1e-5, atol=1e-7 * np.abs(fd).max())


def test_named_derivatives_parallel_their_inputs():
    """The reference's tests/models/test_abc_contract.py:31-65 through the HIP library: dC_dxi / dC_dxi_prev / dC_dp / dC_dU /
    dC_dU_prev are trees parallel to their inputs; dC_dp covers the native leaves (cm_evaluate) and the rotation matrix
    (cm_param_blocks), both against central differences of the residual."""
    from cmad_amd.models import SmallElasticPlastic
    from problems import check_named_derivatives
    check_named_derivatives(SmallElasticPlastic)


def test_facade_error_behaviour():
    from cmad_amd.models import DefType, SmallElasticPlastic
    with pytest.raises(NotImplementedError):
        SmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRAIN)
    m = SmallElasticPlastic(params_J2_voce(), DefType.FULL_3D)
    with pytest.raises(AssertionError):
        m.Jac()                                         # DNONE mode, reference model.py:303-314
    with pytest.raises(AssertionError):
        m.dSigma()


@pytest.mark.parametrize("def_name", ["FULL_3D", "PLANE_STRESS", "UNIAXIAL_STRESS"])
def test_rate_model_reproduces_analytical_fields(golden_dir, def_name):
    """`small rate` leg of tests/models/test_elastic_plastic_models.py:15-125, 138-146 through the facade
    (UNIAXIAL_STRESS: 12 local dofs, dual-number blocks + host Newton loop)."""
    from cmad_amd.models import DefType, SmallRateElasticPlastic, mp_U_from_F, newton_solve
    def_type = getattr(DefType, def_name)
    nd = {"FULL_3D": 3, "PLANE_STRESS": 2, "UNIAXIAL_STRESS": 1}[def_name]
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    for name in (("uniaxial",) if nd == 1 else ("uniaxial", "biaxial")):
        stress, strain, alpha = g[f"{name}_stress"], g[f"{name}_strain"], g[f"{name}_alpha"]
        F = np.repeat(np.eye(nd)[:, :, None], 101, axis=2)
        F[:, :, 1:] += strain[:nd, :nd, :]
        model = SmallRateElasticPlastic(params_J2_voce(), def_type)
        cauchy = np.zeros((3, 3, 101)); alphas = []
        model.set_xi_to_init_vals()
        for step in range(1, 101):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            newton_solve(model)
            alphas.append(model.xi()[1][0])
            model.evaluate_cauchy()
            cauchy[:, :, step] = model.Sigma().copy()
            model.advance_xi()
        assert np.linalg.norm(np.array(alphas) - alpha) < 1e-6
        assert np.linalg.norm(cauchy[:, :, 1:] - stress) < 1e-6


def test_fe_coupled_bridge_layouts():
    """(n_elems, n_ips, ...) AoS in/out around cm_update_tangent vs the oracle's IFT tangent."""
    import torch
    from cmad_amd.global_residuals import local_update_with_tangent
    from cmad_amd.models import NewtonSettings
    from cmad_amd.synthetic import gauss_point_batch
    DefType, SmallElasticPlastic = _models()
    ne, nip = 37, 8
    B = ne * nip
    model = SmallElasticPlastic(params_J2_voce(scale_params=False), DefType.FULL_3D)
    g = gauss_point_batch(B, seed=5, skew=True)                       # (9, B)
    mat = ol.Material(ol.j2_voce_values())
    st_o = ol.newton_settings(max_iters=20, abs_tol=1e-12, rel_tol=1e-12, ls_kind=ol.LS_TRACED, ls_max_evals=4)
    xp, _, _, _ = mat.update_batch(st_o, 0.7 * g, np.zeros((7, B)))
    xi_o, sig_o, _, cv = mat.update_batch(st_o, g, xp)
    ds_o, _ = mat.tangent_batch(g, xp, xi_o)
    grad_u = torch.from_numpy(g.T.reshape(ne, nip, 3, 3).copy()).cuda()
    xi_prev = torch.from_numpy(xp.T.reshape(ne, nip, 7).copy()).cuda()
    xi, sigma, dsig, status = local_update_with_tangent(model, grad_u, xi_prev)
    np.testing.assert_allclose(xi.cpu().numpy().reshape(B, 7).T, xi_o, rtol=1e-10, atol=1e-11)
    s = sigma.cpu().numpy().reshape(B, 3, 3)
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    for r, (i, j) in enumerate(V6):
        np.testing.assert_allclose(s[:, i, j], sig_o[r], rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(s[:, j, i], sig_o[r], rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(dsig.cpu().numpy().reshape(B, 3, 3, 9)[:, i, j, :].T, ds_o[r], rtol=1e-8,
                                   atol=1e-9 * np.abs(ds_o).max())
    assert ((status.cpu().numpy().astype(np.uint32) >> 16) & 1).all()


def test_fe_coupled_bridge_rate_model():
    """The same bridge for a SmallRateElasticPlastic block (cm_update_rate_tangent; the reference's FE tests run this
    model, tests/fem/test_mixed_up_plastic.py:140-147) vs the oracle's IFT tangent with the previous grad u."""
    import torch
    from cmad_amd.global_residuals import local_update_with_tangent
    from cmad_amd.models import DefType, SmallRateElasticPlastic
    ne, nip = 19, 8
    B = ne * nip
    from cmad_amd.synthetic import gauss_point_batch
    model = SmallRateElasticPlastic(params_J2_voce(scale_params=False), DefType.FULL_3D)
    g = gauss_point_batch(B, seed=6, skew=True)
    mat = ol.Material(ol.j2_voce_values(), model_kind=ol.SMALL_RATE_EP)
    st_o = ol.newton_settings(max_iters=20, abs_tol=1e-12, rel_tol=1e-12, ls_kind=ol.LS_TRACED, ls_max_evals=4)
    xp, _, _, _ = mat.update_batch(st_o, 0.7 * g, np.zeros((7, B)), gradu_prev=np.zeros_like(g))
    xi_o, sig_o, _, cv = mat.update_batch(st_o, g, xp, gradu_prev=0.7 * g)
    ds_o, _ = mat.tangent_batch(g, xp, xi_o, gradu_prev=0.7 * g)
    aos = lambda a, n: torch.from_numpy(a.T.reshape(ne, nip, *n).copy()).cuda()
    xi, sigma, dsig, status = local_update_with_tangent(model, aos(g, (3, 3)), aos(xp, (7,)), grad_u_prev=aos(0.7 * g, (3, 3)))
    np.testing.assert_allclose(xi.cpu().numpy().reshape(B, 7).T[:6], xi_o[:6], rtol=1e-10, atol=1e-7)
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    d = dsig.cpu().numpy().reshape(B, 3, 3, 9)
    for r, (i, j) in enumerate(V6):
        np.testing.assert_allclose(sigma.cpu().numpy().reshape(B, 3, 3)[:, i, j], sig_o[r], rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(d[:, i, j, :].T, ds_o[r], rtol=1e-8, atol=1e-9 * np.abs(ds_o).max())
        np.testing.assert_allclose(d[:, j, i, :], d[:, i, j, :])
    with pytest.raises(ValueError):
        local_update_with_tangent(model, aos(g, (3, 3)), aos(xp, (7,)))



@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
def test_uniaxial_stress_model_reproduces_analytical_fields(golden_dir, yield_kind):
    """test_small_uniaxial_stress of tests/models/test_elastic_plastic_models.py (n_xi = 9, 1x1 grad u)."""
    from cmad_amd.models import DefType, SmallElasticPlastic, mp_U_from_F, newton_solve
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    stress, strain, alpha = g["uniaxial_stress"], g["uniaxial_strain"], g["uniaxial_alpha"]
    F = np.repeat(np.eye(1)[:, :, None], 101, axis=2)
    F[:, :, 1:] += strain[:1, :1, :]
    model = SmallElasticPlastic(params_J2_voce(yield_kind=yield_kind), DefType.UNIAXIAL_STRESS)
    cauchy = np.zeros((3, 3, 101)); alphas = []
    model.set_xi_to_init_vals()
    for step in range(1, 101):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        newton_solve(model)
        alphas.append(model.xi()[1][0])
        model.evaluate_cauchy()
        cauchy[:, :, step] = model.Sigma().copy()
        model.advance_xi()
    assert np.linalg.norm(np.array(alphas) - alpha) < 1e-6
    assert np.linalg.norm(cauchy[:, :, 1:] - stress) < 1e-6


def test_uniaxial_stress_batched_objective_matches_pointwise():
    """BatchedCalibrationObjective on a UNIAXIAL_STRESS Hill model with active Hill coefficients (the parameter set
    of cmad/calibrations/al7079/multi_experiment_hill_calibration.py): cm_update + cm_adjoint_step over the history
    == the sum of the pointwise adjoint objectives on the axial stress."""
    import copy
    import torch
    from cmad_amd.models import DefType, SmallElasticPlastic
    from cmad_amd.objectives import BatchedCalibrationObjective, MPAdjointObjective
    from cmad_amd.parameters import Parameters
    from cmad_amd.parameters.parameters import tree_map
    from cmad_amd.qois import Calibration
    K, B, idx = 12, 3, 1
    F = np.repeat(np.eye(1)[:, :, None], K + 1, axis=2)
    F[0, 0, :] += np.linspace(0., 0.006, K + 1)
    base = params_J2_voce(yield_kind="hill")
    values = copy.deepcopy(base.values)
    hill = values["plastic"]["effective stress"]["hill"]
    for key, v in zip(sorted(hill), [0.45, 0.55, 0.5, 1.4, 1.6, 1.5]):
        hill[key] = v
    flags = tree_map(lambda a: False, copy.deepcopy(values))
    flags["plastic"]["effective stress"]["hill"] = tree_map(lambda a: True, flags["plastic"]["effective stress"]["hill"])
    flags["plastic"]["flow stress"] = tree_map(lambda a: True, flags["plastic"]["flow stress"])
    transforms = tree_map(lambda a: None, copy.deepcopy(values))
    rng = np.random.default_rng(4)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    values["rotation matrix"] = Q * np.sign(np.linalg.det(Q))
    model = SmallElasticPlastic(Parameters(values, flags, transforms), DefType.UNIAXIAL_STRESS, uniaxial_stress_idx=idx)
    cauchy = _compute_cauchy(model, F)
    assert abs(cauchy[idx, idx, -1]) > 150.
    weight = np.zeros((3, 3)); weight[idx, idx] = 1.
    datas = [cauchy + rng.normal(0., 5., cauchy.shape) for _ in range(B)]
    datas = [0.5 * (d + d.transpose(1, 0, 2)) for d in datas]
    model.parameters.set_active_values_from_flat(1.05 * model.parameters.flat_active_values(False), False)
    x = model.parameters.flat_active_values(True)
    J_ref, g_ref = 0., 0.
    for d in datas:
        r = MPAdjointObjective(Calibration(model, d, weight), F).evaluate(x)
        J_ref += r.J; g_ref = g_ref + r.grad
    gh = torch.from_numpy(np.stack([np.full((1, B), F[0, 0, k] - 1.) for k in range(K + 1)])).cuda()
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    dh = torch.from_numpy(np.stack([np.stack([[d[i, j, k] for d in datas] for i, j in V6]) for k in range(K + 1)])).cuda()
    r = BatchedCalibrationObjective(model, gh.contiguous(), dh.contiguous(), weight).evaluate(x)
    assert len(r.grad) == 9
    np.testing.assert_allclose(r.J, J_ref, rtol=1e-11)
    np.testing.assert_allclose(r.grad, g_ref, rtol=1e-8, atol=1e-10 * np.abs(g_ref).max())


def test_uniaxial_calibration_direct_equals_adjoint_and_fd():
    """UniaxialCalibration QoI (stress + lateral strains) on a UNIAXIAL_STRESS Hill model: the two sensitivity
    strategies agree and match central finite differences."""
    from cmad_amd.models import DefType, SmallElasticPlastic, mp_U_from_F, newton_solve
    from cmad_amd.objectives import MPAdjointObjective, MPDirectObjective
    from cmad_amd.qois import UniaxialCalibration
    K = 16
    F = np.repeat(np.eye(1)[:, :, None], K + 1, axis=2)
    F[0, 0, :] += np.linspace(0., 0.006, K + 1)
    model = SmallElasticPlastic(params_J2_voce(yield_kind="hill"), DefType.UNIAXIAL_STRESS, uniaxial_stress_idx=1)
    data = np.zeros((3, K + 1))
    model.set_xi_to_init_vals()
    for step in range(1, K + 1):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        newton_solve(model)
        model.evaluate_cauchy()
        data[:, step] = [model.Sigma()[1, 1], model.xi()[2][0] - 1., model.xi()[2][1] - 1.]
        model.advance_xi()
    weight = np.ones((3, K + 1)); weight[1:, :] = 1e4
    qoi = UniaxialCalibration(model, data, weight, uniaxial_stress_idx=1, stretch_var_idx=2)
    model.parameters.set_active_values_from_flat(1.1 * model.parameters.flat_active_values(False), False)
    x = model.parameters.flat_active_values(True)
    Jd, gd = MPDirectObjective(qoi, F).evaluate(x)
    Ja, ga = MPAdjointObjective(qoi, F).evaluate(x)
    assert abs(Jd - Ja) <= 1e-12 * abs(Jd) and Jd > 0
    np.testing.assert_allclose(gd, ga, rtol=1e-8, atol=1e-10 * np.abs(ga).max())
    h = 1e-6
    for k in range(3):
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        Jp = MPDirectObjective(qoi, F).evaluate(xp_).J
        Jm = MPDirectObjective(qoi, F).evaluate(xm_).J
        np.testing.assert_allclose((Jp - Jm) / (2 * h), ga[k], rtol=2e-5, atol=1e-7 * np.abs(ga).max())


def test_rate_model_calibration_recovers_truth():
    """tests/objectives/test_calibrations.py:57-109 as written in the reference: SmallRateElasticPlastic,
    PLANE_STRESS, Calibration QoI, L-BFGS-B from canonical 0.1 with the adjoint and the direct objectives."""
    from scipy.optimize import fmin_l_bfgs_b
    from cmad_amd.models import DefType, SmallRateElasticPlastic
    from cmad_amd.objectives import MPAdjointObjective, MPDirectObjective
    from cmad_amd.qois import Calibration
    F = plane_stress_F(0.01, 8)                        # 16 steps (the reference uses 100 on the same path)
    model = SmallRateElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
    true_params = model.parameters.flat_active_values()
    cauchy = _compute_cauchy(model, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    qoi = Calibration(model, cauchy, weight)
    x0 = 0.1 * np.ones(3)
    ra = MPAdjointObjective(qoi, F).evaluate(x0)
    rd = MPDirectObjective(qoi, F).evaluate(x0)
    assert abs(ra.J - rd.J) <= 1e-12 * abs(ra.J)
    np.testing.assert_allclose(ra.grad, rd.grad, rtol=1e-8, atol=1e-10 * np.abs(ra.grad).max())
    obj = MPAdjointObjective(qoi, F)
    opt, fval, info = fmin_l_bfgs_b(lambda x: tuple(obj.evaluate(x)), x0, bounds=model.parameters.opt_bounds, factr=10)
    model.parameters.set_active_values_from_flat(opt)
    assert np.linalg.norm(model.parameters.flat_active_values() - true_params) < 1e-6


@pytest.mark.parametrize("def_type_name", ["FULL_3D", "PLANE_STRESS"])
def test_rate_model_batched_objective_matches_pointwise(def_type_name):
    """BatchedCalibrationObjective on SmallRateElasticPlastic (cm_update_rate + cm_adjoint_step_rate) == the sum of
    the reference-style pointwise adjoint objectives, per-point data, canonical gradient."""
    import torch
    from cmad_amd.models import DefType, SmallRateElasticPlastic
    from cmad_amd.objectives import BatchedCalibrationObjective, MPAdjointObjective
    from cmad_amd.qois import Calibration
    def_type = getattr(DefType, def_type_name)
    if def_type == DefType.PLANE_STRESS:
        F = plane_stress_F(0.02, 4)
    else:
        Fp = plane_stress_F(0.02, 4)
        F = np.tile(np.eye(3)[:, :, None], (1, 1, Fp.shape[2]))
        F[:2, :2, :] = Fp
        F[2, 2, :] = 1. - 0.4 * (Fp[0, 0, :] - 1.)
    nd = F.shape[0]
    K = F.shape[2] - 1
    model = SmallRateElasticPlastic(params_J2_voce(), def_type)
    cauchy = _compute_cauchy(model, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.; weight[0, 1] = weight[1, 0] = 0.5
    rng = np.random.default_rng(23)
    B = 3
    datas = [cauchy + rng.normal(0., 5., cauchy.shape) for _ in range(B)]
    datas = [0.5 * (d + d.transpose(1, 0, 2)) for d in datas]
    model.parameters.set_active_values_from_flat(1.05 * model.parameters.flat_active_values(False), False)
    x = model.parameters.flat_active_values(True)
    J_ref, g_ref = 0., 0.
    for d in datas:
        r = MPAdjointObjective(Calibration(model, d, weight), F).evaluate(x)
        J_ref += r.J; g_ref = g_ref + r.grad
    gh = torch.from_numpy(np.stack([np.tile((F[:, :, k] - np.eye(nd)).reshape(nd * nd, 1), (1, B)) for k in range(K + 1)])).cuda()
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    dh = torch.from_numpy(np.stack([np.stack([[d[i, j, k] for d in datas] for i, j in V6]) for k in range(K + 1)])).cuda()
    r = BatchedCalibrationObjective(model, gh.contiguous(), dh.contiguous(), weight).evaluate(x)
    np.testing.assert_allclose(r.J, J_ref, rtol=1e-11)
    np.testing.assert_allclose(r.grad, g_ref, rtol=1e-8, atol=1e-10 * np.abs(g_ref).max())
    r2 = BatchedCalibrationObjective(model, gh.contiguous(), dh.contiguous(), weight, fused_history=False).evaluate(x)
    np.testing.assert_allclose(r2.J, r.J, rtol=1e-13)
    np.testing.assert_allclose(r2.grad, r.grad, rtol=1e-10, atol=1e-12 * np.abs(r.grad).max())


def _hessian_problem(active_elastic=False, K=10, rate=False):
    from cmad_amd.models import DefType, SmallElasticPlastic, SmallRateElasticPlastic
    from cmad_amd.qois import Calibration
    params = params_J2_voce()
    if active_elastic:
        import copy
        from cmad_amd.parameters import Parameters
        from cmad_amd.parameters.parameters import tree_map
        values = params.values
        flags = tree_map(lambda a: False, copy.deepcopy(values))
        flags["elastic"] = {"E": True, "nu": True}
        flags["plastic"]["flow stress"] = tree_map(lambda x: True, flags["plastic"]["flow stress"])
        tr = tree_map(lambda a: None, copy.deepcopy(values))
        tr["elastic"]["E"] = np.array([200e3])                 # log transform
        tr["plastic"]["flow stress"]["initial yield"]["Y"] = np.array([200.])
        tr["plastic"]["flow stress"]["hardening"]["voce"]["S"] = np.array([100., 300.])
        tr["plastic"]["flow stress"]["hardening"]["voce"]["D"] = np.array([10., 30.])
        params = Parameters(values, flags, tr)
    F = plane_stress_F(0.02, K // 2)
    model = (SmallRateElasticPlastic if rate else SmallElasticPlastic)(params, DefType.PLANE_STRESS)
    cauchy = _compute_cauchy(model, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    rng = np.random.default_rng(22)
    data = cauchy + rng.normal(0., 2., cauchy.shape)
    qoi = Calibration(model, data, weight)
    model.parameters.set_active_values_from_flat(1.1 * model.parameters.flat_active_values(False), False)
    return model, qoi, F


@pytest.mark.parametrize("active_elastic,rate", [(False, False), (True, False), (False, True), (True, True)])
def test_direct_adjoint_hessian(active_elastic, rate):
    """tests/objectives/test_J2_fd_checks.py:66-98, 366-372, 390-396 (both SmallElasticPlastic and
    SmallRateElasticPlastic): the Hessian of MPDirectAdjointObjective is symmetric, matches central differences of
    the adjoint gradient, and its directional second derivative's FD error drops by decades.  With E, nu active it
    also exercises the second-order elastic-constant chain."""
    from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective
    model, qoi, F = _hessian_problem(active_elastic, rate=rate)
    x = model.parameters.flat_active_values(True)
    J, grad, H = MPDirectAdjointObjective(qoi, F).evaluate(x)
    Ja, ga = MPAdjointObjective(qoi, F).evaluate(x)
    assert abs(J - Ja) <= 1e-12 * abs(J)
    np.testing.assert_allclose(grad, ga, rtol=1e-10, atol=1e-12 * np.abs(ga).max())
    np.testing.assert_allclose(H, H.T, rtol=1e-9, atol=1e-9 * np.abs(H).max())
    n = x.size
    H_fd = np.zeros((n, n))
    h = 1e-5
    for k in range(n):
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        H_fd[:, k] = (MPAdjointObjective(qoi, F).evaluate(xp_).grad - MPAdjointObjective(qoi, F).evaluate(xm_).grad) / (2 * h)
    np.testing.assert_allclose(H, H_fd, rtol=5e-5, atol=5e-6 * np.abs(H).max())
    rng = np.random.default_rng(22)
    d = rng.uniform(-1., 1., size=n)
    ref = d @ H @ d
    errs = []
    for hh_ in np.logspace(-1, -4, 4):
        Jp = MPAdjointObjective(qoi, F).evaluate(x + hh_ * d).J
        Jm = MPAdjointObjective(qoi, F).evaluate(x - hh_ * d).J
        errs.append(abs((Jp + Jm - 2. * J) / hh_ ** 2 - ref))
    assert np.log10(max(errs) / min(errs)) > 3.0


def test_evaluate_hessians_shapes_and_errors():
    from cmad_amd.models import DefType, SmallElasticPlastic, SmallRateElasticPlastic, mp_U_from_F, newton_solve
    m = SmallElasticPlastic(params_J2_voce(), DefType.FULL_3D)
    G = np.diag([0.004, -0.001, -0.001])
    m.gather_global(mp_U_from_F(np.eye(3) + G), mp_U_from_F(np.eye(3)))
    newton_solve(m)
    m.evaluate_hessians()
    assert m.d2C_dxi2.shape == (7, 7, 7) and m.d2C_dxi_dxi_prev.shape == (7, 7, 7) and m.d2C_dxi_prev2.shape == (7, 7, 7)
    assert m.d2C_dparams2.shape == (7, 3, 3) and m.d2C_dxi_dparams.shape == (7, 7, 3) and m.d2C_dxi_prev_dparams.shape == (7, 7, 3)
    assert np.abs(m.d2C_dxi2).max() > 0
    r = SmallRateElasticPlastic(params_J2_voce(), DefType.FULL_3D)              # cm_hessians_rate
    r.gather_global(mp_U_from_F(np.eye(3) + G), mp_U_from_F(np.eye(3)))
    newton_solve(r)
    r.evaluate_hessians()
    assert r.d2C_dxi2.shape == (7, 7, 7) and r.d2C_dparams2.shape == (7, 3, 3) and np.abs(r.d2C_dxi2).max() > 0
    from cmad_amd.models import HybridHillEffectiveStress
    from cmad_amd.parameters import Parameters
    from cmad_amd.synthetic import al7079_hybrid_setup
    icnn, values = al7079_hybrid_setup()
    hmodel = SmallElasticPlastic(Parameters(values), DefType.FULL_3D, effective_stress_fun=HybridHillEffectiveStress(icnn))
    hmodel.gather_global(mp_U_from_F(np.eye(3) + 3.0 * G), mp_U_from_F(np.eye(3)))
    hmodel.set_scalar_xi(1, np.array([2e-4]))                                    # trial state past the yield surface: plastic branch
    hmodel.evaluate_hessians()                                                  # network surfaces: arithmetic-T model (cm_hessians)
    assert hmodel.d2C_dxi2.shape == (7, 7, 7) and np.isfinite(hmodel.d2C_dxi2).all()     # values: test_second_derivatives_network_surfaces
    from cmad_amd.models.device import BARLAT_NAMES
    bvals = params_J2_voce().values
    bvals["plastic"]["effective stress"] = {"barlat": dict(zip(BARLAT_NAMES, [1.0] * 18 + [8.0]))}
    bmodel = SmallElasticPlastic(Parameters(bvals), DefType.FULL_3D)
    bmodel.gather_global(mp_U_from_F(np.eye(3) + G), mp_U_from_F(np.eye(3)))
    bmodel.set_scalar_xi(1, np.array([2e-4]))
    bmodel.evaluate_hessians()                                                  # Barlat: Jacobi eigen-decomposition in arithmetic T
    assert bmodel.d2C_dxi2.shape == (7, 7, 7) and np.isfinite(bmodel.d2C_dxi2).all()     # values: test_barlat_second_derivatives_...


def test_jvp_objective_agrees_with_direct_adjoint():
    """tests/objectives/test_jvp_vs_original.py:31-97: J rtol 1e-12, grad 1e-9, Hessian 1e-8 between
    MPDirectAdjointObjective (per-point evaluate blocks) and MPJVPObjective (batched update / adjoint kernels)."""
    from cmad_amd.models import DefType, SmallElasticPlastic, make_newton_solve
    from cmad_amd.objectives import MPDirectAdjointObjective, MPJVPObjective
    from cmad_amd.qois import Calibration
    F = plane_stress_F(0.02, 6)
    truth = SmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
    cauchy = _compute_cauchy(truth, F)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.
    x0 = 0.1 * np.ones(3)
    m1 = SmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
    J1, g1, H1 = MPDirectAdjointObjective(Calibration(m1, cauchy, weight), F).evaluate(x0)
    m2 = SmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
    obj = MPJVPObjective(Calibration(m2, cauchy, weight), F, make_newton_solve(m2._residual))
    J2, g2 = obj.evaluate_objective_and_grad(x0)
    H2 = obj.evaluate_hessian(x0)
    np.testing.assert_allclose(J1, J2, rtol=1e-10)
    np.testing.assert_allclose(g1, g2, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(H1, H2, rtol=1e-8, atol=1e-8)


# ---- `cmad` command line on the device (reference tests/cli/test_*_roundtrip.py) -------------------------------

@pytest.mark.gpu
def test_cli_primal_and_objective(tmp_path):
    import cli_cases
    from cmad_amd.cli.main import main
    (tmp_path / "p").mkdir(); (tmp_path / "o").mkdir()
    cli_cases.check_primal(main, tmp_path / "p")
    cli_cases.check_objective(main, tmp_path / "o")


@pytest.mark.gpu
def test_cli_gradient_and_hessian_strategies(tmp_path):
    import cli_cases
    from cmad_amd.cli.main import main
    (tmp_path / "g").mkdir(); (tmp_path / "h").mkdir()
    cli_cases.check_gradient(main, tmp_path / "g", ["adjoint", "direct", "direct_adjoint", "jvp"])
    cli_cases.check_hessian(main, tmp_path / "h", ["direct_adjoint", "jvp"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adjoint", "jvp"])
def test_cli_calibrate(tmp_path, kind):
    import cli_cases
    from cmad_amd.cli.main import main
    cli_cases.check_calibrate(main, tmp_path, num_pts=50 if kind == "jvp" else 20, kind=kind)


@pytest.mark.gpu
def test_scaled_hybrid_uniaxial_stress_forward():
    """Mirror of the reference's calibrations/al7079/nn_hill_uniaxial_stress_forward.py:84-151: UNIAXIAL_STRESS along
    axis 1 with the beta-rescaled hybrid Hill + ICNN surface, rotated material frames, imperative newton_solve with
    the legacy back-tracking (max_ls_evals=5); the script's own check is that the stress stays uniaxial."""
    from cmad_amd.models import (DefType, ScaledHybridHillEffectiveStress, SmallElasticPlastic, mp_U_from_F,
                                 newton_solve)
    from cmad_amd.parameters import Parameters
    from cmad_amd.synthetic import al7079_hybrid_setup
    icnn, values = al7079_hybrid_setup()
    params = Parameters(values)
    model = SmallElasticPlastic(params, DefType.UNIAXIAL_STRESS, uniaxial_stress_idx=1,
                                effective_stress_fun=ScaledHybridHillEffectiveStress(icnn, 525.0))
    num_steps = 40
    F = np.repeat(np.eye(1)[:, :, None], num_steps + 1, axis=2)
    F[0, 0, :] += np.linspace(0.0, 0.03, num_steps + 1)
    rng = np.random.default_rng(3)
    peak = []
    for _ in range(2):
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        params.set_rotation_matrix(q * np.sign(np.linalg.det(q)))
        model.set_xi_to_init_vals()
        cauchy = np.zeros((3, 3, num_steps + 1))
        for step in range(1, num_steps + 1):
            model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
            iters, res = newton_solve(model, abs_tol=1e-13, rel_tol=1e-13, max_ls_evals=5)
            assert res < 1e-11, (step, iters, res)
            model.seed_none()
            model.evaluate_cauchy()
            cauchy[:, :, step] = model.Sigma().copy()
            model.advance_xi()
        assert abs(np.linalg.norm(cauchy) - np.linalg.norm(cauchy[1, 1, :])) < 1e-11
        assert model.xi()[1][0] > 1e-3                          # the point did yield
        peak.append(cauchy[1, 1, -1])
    assert 400.0 < min(peak) and max(peak) < 2000.0 and abs(peak[0] - peak[1]) > 1e-3   # anisotropic response


@pytest.mark.gpu
@pytest.mark.parametrize("history", ["uniaxial", "biaxial"])
def test_isotropic_barlat_reproduces_the_j2_analytical_fields(history):
    """Yld2004-18p with unit coefficients and a = 4 is von Mises: the facade model with a `barlat` parameter block
    must reproduce the reference's analytical J2 + Voce fields (tests/models/test_elastic_plastic_models.py:15-125,
    tolerance 1e-6) -- including the uniaxial history, whose repeated eigenvalue is where differentiating through
    eigh breaks down and the kernel's spectral form takes the limit instead."""
    from cmad_amd.models import DefType, SmallElasticPlastic, mp_U_from_F, newton_solve
    from cmad_amd.parameters import Parameters
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "j2_voce_analytical.npz"))
    stress, strain, alpha = g[f"{history}_stress"], g[f"{history}_strain"], g[f"{history}_alpha"]
    values = ol.j2_voce_values(yield_kind="barlat", a=4.0)
    model = SmallElasticPlastic(Parameters(values), DefType.FULL_3D)
    n = strain.shape[2]
    F = np.repeat(np.eye(3)[:, :, None], n + 1, axis=2)
    F[:, :, 1:] += strain
    cauchy, alphas = np.zeros((3, 3, n + 1)), np.zeros(n)
    model.set_xi_to_init_vals()
    for step in range(1, n + 1):
        model.gather_global(mp_U_from_F(F[:, :, step]), mp_U_from_F(F[:, :, step - 1]))
        newton_solve(model)
        model.advance_xi()
        model.evaluate_cauchy()
        cauchy[:, :, step] = model.Sigma()
        alphas[step - 1] = model.xi()[1][0]
    assert np.linalg.norm(alphas - alpha) < 1e-6
    assert np.linalg.norm(cauchy[:, :, 1:] - stress) < 1e-6


@pytest.mark.gpu
def test_config0_uniaxial_ramp_on_the_fe_layout(golden_dir):
    """BASELINE.json configs[0] (examples/elastic_plastic_uniaxial.yaml: J2 + Voce cube under a uniaxial-stress ramp,
    1331 hexes x 8 integration points = 10 648 Gauss points at --n 11) without the FE solver, which is out of scope:
    every integration point follows the homogeneous solution's strain path through the FE COUPLED bridge with the FE
    binding's local solver settings, and must reproduce the reference's closed form (compute_plastic_fields; the
    reference's own FE round trip checks it at rtol 1e-3, tests/cli/test_primal_fe_roundtrip.py:183-234)."""
    import torch
    from cmad_amd.global_residuals import local_update_with_tangent
    DefType, SmallElasticPlastic = _models()
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    strain, stress, alpha = g["uniaxial30_strain"], g["uniaxial30_stress"], g["uniaxial30_alpha"]
    ne, nip = 1331, 8
    model = SmallElasticPlastic(params_J2_voce(scale_params=False), DefType.FULL_3D)
    xi = torch.zeros((ne, nip, 7), dtype=torch.float64, device="cuda")
    for step in range(strain.shape[2]):
        grad_u = torch.from_numpy(strain[:, :, step]).cuda().expand(ne, nip, 3, 3).contiguous()
        xi, sigma, dsig, status = local_update_with_tangent(model, grad_u, xi)
        assert bool(((status.to(torch.int64) >> 16) & 1).all())
        s = sigma.reshape(-1, 3, 3)
        assert float((s - s[0]).abs().max()) == 0.0                      # every point identical, bit for bit
        np.testing.assert_allclose(s[0].cpu().numpy(), stress[:, :, step], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(float(xi[0, 0, 6]), alpha[step], rtol=1e-6, atol=1e-9)
    # consistent tangent at the final state: d sigma_xx / d eps_xx under the constrained (uniaxial-strain) perturbation
    # is finite, symmetric in the minor indices and softer than the elastic modulus
    d = dsig.reshape(-1, 3, 3, 3, 3)[0].cpu().numpy()
    np.testing.assert_allclose(d, d.transpose(1, 0, 2, 3), rtol=0, atol=1e-9)
    lam, mu = 200e3 * 0.3 / (1.3 * 0.4), 200e3 / 2.6
    assert 0.0 < d[0, 0, 0, 0] < lam + 2 * mu


_RCCL_CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
use_group = sys.argv[2] == "1"
import torch
import torch.distributed as dist
if use_group:                                  # the process group comes up before anything else touches the GPU
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cmad_amd.models import DefType, SmallElasticPlastic
from cmad_amd.objectives import BatchedCalibrationObjective
from cmad_amd.objectives.batched import shard_bounds
from cmad_amd.synthetic import gauss_point_batch
from problems import params_J2_voce
B, K = 20000, 3
lo, hi = shard_bounds(B, 0, 1)
g1 = torch.from_numpy(gauss_point_batch(B, seed=5, ndims=2)[:, lo:hi]).cuda()
ramp = torch.linspace(0.0, 1.5, K + 1, dtype=torch.float64, device="cuda")
gh = (ramp[:, None, None] * g1[None]).contiguous()
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
dh = 50.0 * torch.randn((K + 1, 6, hi - lo), dtype=torch.float64, device="cuda", generator=gen)
model = SmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
w = np.zeros((3, 3)); w[0, 0] = w[1, 1] = 1.0
obj = BatchedCalibrationObjective(model, gh, dh, w)
x = model.parameters.flat_active_values(True)
import cmad_amd.objectives.batched as batched
calls = []
orig = dist.all_reduce
def spy(t, *a, **k):                           # what the product hands to the collective, and on which backend
    calls.append((tuple(t.shape), str(t.dtype), t.is_cuda, dist.get_backend(k.get("group"))))
    return orig(t, *a, **k)
dist.all_reduce = spy
r = obj.evaluate(x)
dist.all_reduce = orig
if use_group:
    backend = dist.get_backend()
    dist.destroy_process_group()
else:
    backend = "none"
print("RESULT " + json.dumps({"J": r.J, "grad": list(map(float, r.grad)), "backend": backend,
                              "collective_calls": batched.COLLECTIVE_CALLS, "seen": calls}))
'''


def test_sharded_objective_through_rccl_process_group(tmp_path):
    """The multi-GPU path's collective on hardware: a fresh child process brings up a 1-rank `nccl` (= RCCL) process group
    before touching the GPU and evaluates `BatchedCalibrationObjective` on its shard.  `allreduce_sum_` sends the product's own
    (1 + 12)-vector through `dist.all_reduce` whenever a group is initialised (one rank included), so the vector really
    traverses ncclAllReduce: the child records every tensor handed to the collective (a CUDA float64 vector of 13 on the nccl
    backend, exactly once per evaluation) and the counter the collective path increments; objective and gradient must equal
    those of a child that runs without a group (whose counter stays 0).
    (The 8-GPU scaling run is the driver's; world size 2 over gloo runs in tests/test_host_logic.py.)"""
    import json
    import subprocess
    import sys
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD)
    port = str(29600 + (os.getpid() % 1000))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = {}
    for flag in ("1", "0"):
        res = subprocess.run([sys.executable, str(script), ROOT, flag, port], capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        out[flag] = json.loads(line[len("RESULT "):])
    assert out["1"]["backend"] == "nccl"
    assert out["1"]["collective_calls"] == 1 and out["0"]["collective_calls"] == 0
    assert out["1"]["seen"] == [[[13], "torch.float64", True, "nccl"]] and out["0"]["seen"] == []
    assert np.isfinite(out["1"]["J"]) and out["1"]["J"] > 0.0
    np.testing.assert_allclose(out["1"]["J"], out["0"]["J"], rtol=1e-14)
    np.testing.assert_allclose(out["1"]["grad"], out["0"]["grad"], rtol=1e-13, atol=0.0)


@pytest.mark.parametrize("yield_kind,active_rotation", [("hosford", False), ("hill", True), ("hosford", True),
                                                        ("network", False), ("network deep", False)])
def test_gradient_of_extended_leaves(yield_kind, active_rotation):
    """Objective gradients w.r.t. the Hosford exponent and the entries of the rotation matrix -- leaves the reference reaches
    by jacrev over the params pytree (cmad/models/model.py:125-153) and the kernels by forward-mode evaluation of the whole
    model (cm_param_blocks, cm_param_adjoint_history): adjoint == direct == central differences of the objective."""
    from cmad_amd.objectives import MPAdjointObjective, MPDirectObjective
    from problems import extended_leaf_problem
    DefType, SmallElasticPlastic = _models()
    model, qoi, F = extended_leaf_problem(SmallElasticPlastic, yield_kind, active_rotation)
    x = model.parameters.flat_active_values(True)
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
    model = _models()[1](params_J2_voce(yield_kind="hill"), DefType.UNIAXIAL_STRESS, uniaxial_stress_idx=1)
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
    """`SmallElasticPlastic(parameters, def_type, hardening_funs={"neural network": SimpleNeuralNetwork([1, 5, 1], ...).evaluate})`
    (and widths [1, 4, 3, 1]: the reference's forward loops over any depth, simple_neural_network.py:19-23)
    as examples/noisy_calibration.py:245-252 builds it: the adjoint and the direct objective agree, and the gradient w.r.t.
    the initial yield and EVERY network weight and bias (active leaves of the params pytree) matches central differences."""
    import copy
    from cmad_amd.models import DefType, SmallElasticPlastic, SmallRateElasticPlastic
    from cmad_amd.objectives import MPAdjointObjective, MPDirectObjective
    from cmad_amd.parameters import Parameters
    from cmad_amd.parameters.parameters import tree_map
    from cmad_amd.qois import Calibration
    import parity_cases as pc
    values, net, _ = pc.nn_hardening_values(hidden=hidden)
    flags = tree_map(lambda a: False, copy.deepcopy(values))
    flags["plastic"]["flow stress"] = tree_map(lambda a: True, flags["plastic"]["flow stress"])
    params = Parameters(values, flags, tree_map(lambda a: None, copy.deepcopy(values)))
    F = plane_stress_F(0.02, 3)
    model = SmallElasticPlastic(params, DefType.PLANE_STRESS, hardening_funs={"neural network": net.evaluate})
    cauchy = _compute_cauchy(model, F)
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
    # the rate-form model, as the reference's example uses the law
    rmodel = SmallRateElasticPlastic(params, DefType.PLANE_STRESS, hardening_funs={"neural network": net.evaluate})
    rq = Calibration(rmodel, qoi.data(), weight)
    rra = MPAdjointObjective(rq, F).evaluate(x)
    rrd = MPDirectObjective(rq, F).evaluate(x)
    np.testing.assert_allclose(rrd.grad, rra.grad, rtol=1e-8, atol=1e-10 * np.abs(rra.grad).max())
    np.testing.assert_allclose(rra.J, ra.J, rtol=1e-6)                                    # same material, two formulations


@pytest.mark.parametrize("yield_kind,active_rotation", [("hosford", False), ("hill", True), ("network deep", False)])
def test_direct_adjoint_hessian_with_extended_leaves(yield_kind, active_rotation):
    """Second-order sensitivities w.r.t. leaves outside the 12 native kernel parameters -- the Hosford exponent, the nine
    entries of the rotation matrix -- together with a native one (Y): the reference takes Hessians over the whole params pytree
    (cmad/models/model.py:133-147 under cmad/objectives/mp_objective.py:218-345).  MPDirectAdjointObjective's gradient equals the
    adjoint one; its Hessian is symmetric and matches central differences of the adjoint gradient."""
    from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective
    from problems import extended_leaf_problem
    model, qoi, F = extended_leaf_problem(_models()[1], yield_kind, active_rotation)
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


@pytest.mark.gpu
def test_second_order_pass_at_its_extended_parameter_limit():
    """64 active leaves outside the 12 native parameters (every input weight and bias of an ICNN [6, 7, 1], its pass-through
    weights and the rotation matrix) -- the most `cm_hessian_history_ep` carries per evaluation: its quadratic form's tile is
    117 KB of dynamic LDS, above what a kernel gets without asking (hipFuncSetAttribute).  Gradient = the adjoint gradient, the
    Hessian is symmetric and three of its columns match central differences of the adjoint gradient; one leaf more raises a clear
    NotImplementedError instead of a bare CM_ERR_BAD_ARG."""
    from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective
    from problems import extended_leaf_problem
    model, qoi, F = extended_leaf_problem(_models()[1], "network wide", True)
    x = model.parameters.flat_active_values(True)
    assert x.size == 1 + 64                                     # Y + 42 + 7 + 6 network entries + 9 rotation entries
    J, grad, H = MPDirectAdjointObjective(qoi, F).evaluate(x)
    ra = MPAdjointObjective(qoi, F).evaluate(x)
    assert abs(J - ra.J) <= 1e-12 * abs(J)
    np.testing.assert_allclose(grad, ra.grad, rtol=1e-9, atol=1e-11 * np.abs(ra.grad).max())
    np.testing.assert_allclose(H, H.T, rtol=1e-8, atol=1e-9 * np.abs(H).max())
    for k in (0, 7, x.size - 3):
        h = 1e-5 * max(1.0, abs(x[k]))
        xp_, xm_ = x.copy(), x.copy()
        xp_[k] += h; xm_[k] -= h
        col = (MPAdjointObjective(qoi, F).evaluate(xp_).grad - MPAdjointObjective(qoi, F).evaluate(xm_).grad) / (2 * h)
        np.testing.assert_allclose(H[:, k], col, rtol=2e-4, atol=2e-5 * np.abs(H).max())
    # one more array leaf (the z-layer's 7 weights): beyond the limit
    ev = model.device_evaluator()
    with pytest.raises(NotImplementedError):
        ev._check_extended(list(range(65)), "cm_hessian_history_ep")


@pytest.mark.gpu
@pytest.mark.parametrize("rate,scale_params", [(False, False), (True, True)])
def test_complex_step_model_instances(rate, scale_params):
    """The reference's complex-step checks (tests/objectives/test_J2_fd_checks.py:301-386: SmallElasticPlastic unscaled,
    SmallRateElasticPlastic with scaled parameters) through `Model(..., is_complex=True)` on cm_update_complex."""
    from cmad_amd.models import SmallElasticPlastic, SmallRateElasticPlastic
    from problems import check_complex_step
    check_complex_step(SmallRateElasticPlastic if rate else SmallElasticPlastic, scale_params, num_pts_per_increment=25)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["barlat", "network", "network scaled deep", "hardening network deep"])
def test_complex_step_instances_on_dense_surfaces_and_extended_leaves(kind):
    """`is_complex=True` is a constructor flag of every configuration in the reference (small_elastic_plastic.py:90,118-127):
    Barlat, the network surfaces (plain; beta-rescaled around two hidden layers) and the deep network hardening law, with
    perturbed Barlat coefficients / rotation matrix / network weights, on cm_update_complex."""
    from cmad_amd.models import SmallElasticPlastic
    from problems import check_complex_step_extended
    check_complex_step_extended(SmallElasticPlastic, kind)
