"""CPU coverage of the hand-derived kernel arithmetic: cmad_amd/csrc/cm_device.hpp compiled for the
host (tests/native/host_harness.cpp) against the dual-number oracle.  The same scenarios run on the
GPU through the C-ABI in tests/test_gpu_update.py / test_gpu_sensitivities.py."""
import pytest

import oracle_lib as ol
import parity_cases as pc

BACKEND = pc.HostBackend()


@pytest.fixture(params=["structured", "dense", "passes"], autouse=True)
def solver_variant(request):
    """FULL_3D and PLANE_STRESS run three times: the structured (bordered) block solve the kernels use, the dense LU of the
    same system, and the resumable one-evaluation-per-pass form of the iteration (cm_pool.hpp) the work-pool kernels run."""
    import host_harness_lib as hh
    hh.set_dense(request.param == "dense")
    hh.set_passes(request.param == "passes")
    yield request.param
    hh.set_dense(False)
    hh.set_passes(False)


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_update(def_type, yield_kind, kw, rot, ls):
    pc.check_update(BACKEND, pc.Scenario(def_type, yield_kind, kw, rot, ls, B=768))


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.SENS_YIELDS)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_tangent(def_type, yield_kind, kw, rot):
    pc.check_tangent(BACKEND, pc.Scenario(def_type, yield_kind, kw, rot, False, B=256))


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.SENS_YIELDS)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_vjp(def_type, yield_kind, kw, rot):
    pc.check_vjp(BACKEND, pc.Scenario(def_type, yield_kind, kw, rot, False, B=256))


@pytest.mark.parametrize("uidx", [0, 1, 2])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
def test_uniaxial_stress_tangent_and_vjp(yield_kind, kw, rot, uidx, solver_variant):
    """UNIAXIAL_STRESS (9 local dofs, one grad-u entry): IFT tangent and reverse sweep of the dense path vs the
    oracle, the way cmad/calibrations/al7079/multi_experiment_hill_calibration.py differentiates it."""
    if solver_variant == "dense":
        pytest.skip("always dense")
    sc = pc.Scenario(ol.UNIAXIAL_STRESS, yield_kind, kw, rot, False, B=192, uniaxial_idx=uidx)
    pc.check_update(BACKEND, sc)
    pc.check_tangent(BACKEND, sc)
    pc.check_vjp(BACKEND, sc, grad_atol=1e-9)


def _host_history(desc, info, gh, d6, wsq6, xi0):
    import host_harness_lib as hh
    if "nn_packed" in info:
        desc.nn_weights = info["nn_packed"].ctypes.data
    return hh.history(desc, gh, d6, wsq6, xi0)


def _host_primal(desc, info, gh, xi0):
    import host_harness_lib as hh
    if "nn_packed" in info:
        desc.nn_weights = info["nn_packed"].ctypes.data
    return hh.primal_history(desc, gh, xi0)


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_history_objective_grad(def_type, yield_kind, kw, rot, rate, solver_variant):
    """cm::history_point (the body of cm_objective_grad_history): K updates forward with the state carried in
    registers, K adjoint steps backward, both model kinds."""
    if solver_variant != "structured":
        pytest.skip("one variant: the history loop is the same code on both solver paths")
    pc.check_history(_host_history, def_type, yield_kind, kw, rot, rate=rate, B=96, uniaxial_idx=1, primal=_host_primal)


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_direct_sensitivities(def_type, yield_kind, kw, rot, rate, solver_variant):
    """cm::direct_point (the body of cm_direct_step): forward parameter sensitivities propagated over a history."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("one variant: built on the explicit blocks")
    pc.check_direct(lambda desc, info, g, gp, xp, x, dxp: hh.direct_step(desc, g, xp, x, dxp, gradu_prev=gp),
                    def_type, yield_kind, kw, rot, rate=rate, B=64, uniaxial_idx=1)


def test_history_objective_grad_with_line_search():
    pc.check_history(_host_history, ol.FULL_3D, "J2", {}, False, ls=True, B=96, primal=_host_primal)
    pc.check_history(_host_history, ol.FULL_3D, "J2", {}, True, ls=True, B=96, primal=_host_primal, solver_flags=2)
    pc.check_history(_host_history, ol.FULL_3D, "J2", {}, False, B=96, primal=_host_primal, solver_flags=2)
    pc.check_history(_host_history, ol.PLANE_STRESS, "hill", pc.YIELDS[1][1], True, ls=True, B=96, primal=_host_primal)


@pytest.mark.parametrize("plastic", [True, False])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.SENS_YIELDS)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_explicit_blocks_at_arbitrary_states(def_type, yield_kind, kw, rot, plastic):
    """cm_evaluate's blocks (C, dC/dxi, dC/dxi_prev, dC/dparams, dC/dgradu, sigma and its derivatives) at
    non-converged states on both branches vs the oracle's dual-number Jacobians."""
    import numpy as np
    import host_harness_lib as hh
    from cmad_amd.models.device import build_desc, kp_to_leaf_grad
    from test_oracle_vs_torch_ad import _state
    rng = np.random.default_rng(3)
    values = ol.j2_voce_values(yield_kind=yield_kind, Q=pc.rand_rot(rng) if rot else None, **kw)
    mat = ol.Material(values, def_type=def_type)
    desc, info = build_desc(values, def_type=def_type)
    V6 = [0, 1, 2, 4, 5, 8]
    for trial in range(4):
        for _ in range(50):                      # draw until the requested branch is hit
            xi, xp, U = _state(rng, mat, plastic)
            if (mat.yield_state(xi, U)[1] > 0) == plastic:
                break
        else:
            raise AssertionError("could not draw a state on the requested branch")
        g, x1, x0 = U.reshape(-1, 1), xi.reshape(-1, 1), xp.reshape(-1, 1)
        for which_o, which_d in ((ol.W_XI, 0), (ol.W_XI_PREV, 1), (ol.W_U, 3)):
            C, J, s, S = hh.evaluate(desc, which_d, g, x0, x1, mat.nx)
            Jo = mat.jacobian(which_o, xi, xp, U)
            So = mat.dcauchy(which_o, xi, xp, U)[V6, :]
            np.testing.assert_allclose(C[:, 0], mat.residual(xi, xp, U), rtol=1e-11, atol=1e-16)
            np.testing.assert_allclose(s[:, 0], mat.cauchy(xi, U).reshape(9)[V6], rtol=1e-11, atol=1e-10)
            np.testing.assert_allclose(J[:, :, 0], Jo, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(Jo).max()))
            np.testing.assert_allclose(S[:, :, 0], So, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(So).max()))
        C, J, s, S = hh.evaluate(desc, 2, g, x0, x1, mat.nx)
        Jo = mat.jacobian(ol.W_PARAMS, xi, xp, U)
        So = mat.dcauchy(ol.W_PARAMS, xi, xp, U)[V6, :]
        for path in pc.param_paths(yield_kind):
            got = kp_to_leaf_grad(path, np.moveaxis(J[:, :, 0], 1, 0), info)
            ref = Jo[:, mat.param_index(path)]
            np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * max(1e-6, np.abs(Jo).max()), err_msg=str(path))
            got = kp_to_leaf_grad(path, np.moveaxis(S[:, :, 0], 1, 0), info)
            ref = So[:, mat.param_index(path)]
            np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * max(1e-6, np.abs(So).max()), err_msg=str(path))


@pytest.mark.parametrize("reference_iteration", [False, True])
def test_hosford_a100_notch_material(reference_iteration, solver_variant):
    if solver_variant != "structured" and not reference_iteration:
        pytest.skip("the warm start is a variant of the structured solver (cm::newton_fast)")
    pc.check_hosford_a100(BACKEND, B=1024, reference_iteration=reference_iteration)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_hybrid_hill_icnn(def_type, rot):
    pc.check_hybrid_nn(BACKEND, def_type, B=192, rot=rot)


@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "J2", {}), (ol.FULL_3D, "hill", {"hill": pc.HILL}),
                                                    (ol.PLANE_STRESS, "J2", {}), (ol.FULL_3D, "hosford", {"a": 8.5})])
def test_line_search_rejections(def_type, yield_kind, kw):
    pc.check_line_search_rejections(BACKEND, def_type, yield_kind, kw, B=384)


@pytest.mark.parametrize("max_evals", [2, 6])
@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "hosford", {"a": 20.0}), (ol.PLANE_STRESS, "hosford", {"a": 20.0}),
                                                    (ol.UNIAXIAL_STRESS, "hosford", {"a": 20.0}), (ol.FULL_3D, "J2", {}),
                                                    (ol.FULL_3D, "hill", {"hill": pc.HILL})])
def test_legacy_line_search(def_type, yield_kind, kw, max_evals):
    if yield_kind != "hosford":
        pytest.skip("covered by the GPU suite; on the host the Hosford cases exercise every solver form") if max_evals == 2 else None
    try:
        pc.check_legacy_line_search(BACKEND, def_type, yield_kind, kw, B=320, max_evals=max_evals)
    except AssertionError as e:
        if "never engaged" in str(e) and yield_kind != "hosford":
            pytest.skip("full steps always pass the backtracking test for this surface")
        raise


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_barlat_calibrated_coefficients(def_type):
    pc.check_barlat_calibrated(BACKEND, def_type, B=256)


@pytest.mark.parametrize("def_type,rot", [(ol.FULL_3D, True), (ol.PLANE_STRESS, False)])
def test_scaled_hybrid_hill_icnn(def_type, rot):
    """`scaled_effective_stress`: inner scalar Newton for beta, closed-form normal and Hessian through the
    implicit-function rule, against the oracle's dual-number differentiation of the same construction."""
    pc.check_hybrid_nn(BACKEND, def_type, B=96, rot=rot, scaled=True)


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_model_update(def_type, yield_kind, kw, rot, ls, solver_variant):
    import host_harness_lib as hh
    pc.check_rate_model(lambda desc, info, g, gp, xp: hh.update_rate(desc, g, gp, xp, {0: 7, 2: 8, 3: 12}[desc.def_type]),
                        def_type, yield_kind, kw, rot, ls, B=256)


@pytest.mark.parametrize("uidx", [0, 1, 2])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
def test_uniaxial_stress_update_and_blocks(yield_kind, kw, rot, uidx, solver_variant):
    """UNIAXIAL_STRESS (n_xi = 9): 12-step uniaxial ramp through the device math vs the oracle, plus every
    explicit derivative block at the visited states."""
    import numpy as np
    import host_harness_lib as hh
    from cmad_amd.models.device import build_desc, kp_to_leaf_grad
    if solver_variant == "dense":
        pytest.skip("always dense")
    rng = np.random.default_rng(8)
    values = ol.j2_voce_values(yield_kind=yield_kind, Q=pc.rand_rot(rng) if rot else None, **kw)
    mat = ol.Material(values, def_type=ol.UNIAXIAL_STRESS, uniaxial_idx=uidx)
    desc, info = build_desc(values, def_type=ol.UNIAXIAL_STRESS, uniaxial_stress_idx=uidx)
    B = 16
    amp = rng.uniform(0.5, 1.5, B) * rng.choice([-1.0, 1.0], B)
    xp = np.tile(mat.init_xi()[:, None], (1, B))
    V6 = [0, 1, 2, 4, 5, 8]
    for step in range(1, 13):
        g = (amp * 4e-4 * step)[None, :]
        xi_o, sig_o, it_o, cv = mat.update_batch(ol.newton_settings(), g, xp)
        xi_h, sig_h, st = hh.update(desc, g, xp, 9)
        assert cv.all() and ((st >> 16) & 1).all()
        np.testing.assert_allclose(xi_h, xi_o, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(sig_h, sig_o, rtol=1e-10, atol=1e-8)
        if step in (2, 6, 12):
            for b in (0, 5):
                for which_o, which_d in ((ol.W_XI, 0), (ol.W_XI_PREV, 1), (ol.W_U, 3)):
                    C, J, s, S = hh.evaluate(desc, which_d, g[:, b:b + 1], xp[:, b:b + 1], xi_o[:, b:b + 1], 9)
                    Jo = mat.jacobian(which_o, xi_o[:, b], xp[:, b], g[:, b])
                    So = mat.dcauchy(which_o, xi_o[:, b], xp[:, b], g[:, b])[V6, :]
                    np.testing.assert_allclose(J[:, :, 0], Jo, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(Jo).max()))
                    np.testing.assert_allclose(S[:, :, 0], So, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(So).max()))
                C, J, s, S = hh.evaluate(desc, 2, g[:, b:b + 1], xp[:, b:b + 1], xi_o[:, b:b + 1], 9)
                Jo = mat.jacobian(ol.W_PARAMS, xi_o[:, b], xp[:, b], g[:, b])
                So = mat.dcauchy(ol.W_PARAMS, xi_o[:, b], xp[:, b], g[:, b])[V6, :]
                for path in pc.param_paths(yield_kind):
                    np.testing.assert_allclose(kp_to_leaf_grad(path, np.moveaxis(J[:, :, 0], 1, 0), info), Jo[:, mat.param_index(path)],
                                               rtol=1e-9, atol=1e-12 * max(1e-6, np.abs(Jo).max()), err_msg=str(path))
                    np.testing.assert_allclose(kp_to_leaf_grad(path, np.moveaxis(S[:, :, 0], 1, 0), info), So[:, mat.param_index(path)],
                                               rtol=1e-9, atol=1e-12 * max(1e-6, np.abs(So).max()), err_msg=str(path))
        xp = xi_o
    assert (it_o > 0).any()


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_rate_model_explicit_blocks(def_type, yield_kind, kw, rot, solver_variant):
    """cm_evaluate_rate's blocks vs the oracle's dual-number Jacobians of the rate-form residual, at the states
    visited by a three-step history (elastic and plastic points)."""
    import numpy as np
    import host_harness_lib as hh
    from cmad_amd.models.device import build_desc, kp_to_leaf_grad
    from cmad_amd.synthetic import gauss_point_batch
    if solver_variant != "structured":
        pytest.skip("always dense")
    rng = np.random.default_rng(3)
    values = ol.j2_voce_values(yield_kind=yield_kind, Q=pc.rand_rot(rng) if rot else None, **kw)
    mat = ol.Material(values, def_type=def_type, model_kind=ol.SMALL_RATE_EP)
    desc, info = build_desc(values, def_type=def_type, model_kind=1)
    nd = 3 if def_type == ol.FULL_3D else 2
    B = 12
    g0 = gauss_point_batch(B, seed=4, skew=True, ndims=nd)
    gp = np.zeros_like(g0); xp = np.tile(mat.init_xi()[:, None], (1, B))
    V6 = [0, 1, 2, 4, 5, 8]
    nplastic = 0
    for scale in (0.5, 1.0, 1.4):
        g = scale * g0
        xi, sig, it, cv = mat.update_batch(ol.newton_settings(), g, xp, gradu_prev=gp)
        nplastic += int((it > 0).sum())
        for which_o, which_d in ((ol.W_XI, 0), (ol.W_XI_PREV, 1), (ol.W_U, 3), (ol.W_U_PREV, 4)):
            C, J, s, S = hh.evaluate_rate(desc, which_d, g, gp, xp, xi, mat.nx)
            for b in range(B):
                U, Up = g[:, b].reshape(nd, nd), gp[:, b].reshape(nd, nd)
                Jo = mat.jacobian(which_o, xi[:, b], xp[:, b], U, Up)
                So = mat.dcauchy(which_o, xi[:, b], xp[:, b], U, Up)[V6, :]
                np.testing.assert_allclose(C[:, b], mat.residual(xi[:, b], xp[:, b], U, Up), rtol=1e-9, atol=1e-15)
                np.testing.assert_allclose(J[:, :, b], Jo, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(Jo).max()))
                np.testing.assert_allclose(S[:, :, b], So, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(So).max()))
        C, J, s, S = hh.evaluate_rate(desc, 2, g, gp, xp, xi, mat.nx)
        for b in range(B):
            U, Up = g[:, b].reshape(nd, nd), gp[:, b].reshape(nd, nd)
            Jo = mat.jacobian(ol.W_PARAMS, xi[:, b], xp[:, b], U, Up)
            for path in pc.param_paths(yield_kind):
                np.testing.assert_allclose(kp_to_leaf_grad(path, np.moveaxis(J[:, :, b], 1, 0), info), Jo[:, mat.param_index(path)],
                                           rtol=1e-9, atol=1e-12 * max(1e-6, np.abs(Jo).max()), err_msg=str(path))
        xp, gp = xi, g
    assert nplastic > 0


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_random_materials_rate_form(def_type, yield_kind, ls, solver_variant):
    import host_harness_lib as hh
    if solver_variant == "passes":
        pytest.skip("the pass-based solver is the work pool's (total form)")
    nx = {ol.FULL_3D: 7, ol.PLANE_STRESS: 8}[def_type]
    pc.check_random_materials_rate(lambda desc, info, g, gp, xp: hh.update_rate(desc, g, gp, xp, nx),
                                   lambda desc, info, g, gp, xp, x: hh.tangent_rate(desc, g, gp, xp, x),
                                   lambda desc, info, g, gp, xp, x, sb: hh.vjp_rate(desc, g, gp, xp, x, sb),
                                   def_type, yield_kind, ls, seeds=range(3), B=96)


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_random_materials(def_type, yield_kind, ls, solver_variant):
    if solver_variant == "passes" and not ls:
        pytest.skip("one pass-based variant is enough")
    pc.check_random_materials(BACKEND, def_type, yield_kind, ls, seeds=range(3), B=96)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_j2_radial_line_newton_matches_general_path(def_type, rot, solver_variant):
    if solver_variant != "structured":
        pytest.skip("specialisation of the structured path")
    import host_harness_lib as hh
    hh.subspace_fallbacks()
    pc.check_j2_radial_line(BACKEND, B=2048, rot=rot, def_type=def_type)
    # the restricted iteration carried (nearly) every point: a handful per 10^4 may leave for the general path when the
    # reduced and the full residual norm straddle the tolerance
    assert hh.subspace_fallbacks() <= 30          # (six runs of two load steps over 2048 points)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_j2_closed_form_parameter_gradient(def_type, rot, solver_variant):
    """cm::reverse_j2_radial / cm::reverse_j2_plane (the parameter gradient of the fused J2 kernels at converged states: the
    return map differentiated in its own coordinates instead of the 7 / 8-dof transposed solve) against the oracle's IFT
    gradient, two load steps from a hardened state."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("specialisation of the structured path")
    hh.set_radial_vjp(True)
    try:
        for ls in (False, True):
            pc.check_vjp(BACKEND, pc.Scenario(def_type, "J2", {}, rot, ls, B=1024))
    finally:
        hh.set_radial_vjp(False)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_edge_cases(def_type):
    pc.check_edge_cases(BACKEND, def_type)


@pytest.mark.parametrize("plastic", [True, False])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_second_derivatives_vs_oracle(def_type, yield_kind, kw, plastic, solver_variant):
    """cm_hessians (hyper-dual evaluation of the residual in the product code, host build) vs the oracle's nested duals."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_second_derivs(hh.hessians, hh.evaluate, def_type, yield_kind, kw, plastic)


@pytest.mark.parametrize("plastic", [True, False])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_rate_form_second_derivatives_vs_oracle(def_type, yield_kind, kw, plastic, solver_variant):
    """cm_hessians_rate (host build) vs the oracle's nested duals, both branches."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_rate_second_derivs(hh.hessians, hh.evaluate_rate, def_type, yield_kind, kw, plastic)


@pytest.mark.parametrize("plastic", [True, False])
@pytest.mark.parametrize("idx", [0, 1, 2])
@pytest.mark.parametrize("yield_kind,kw", [pc.YIELDS[0], pc.YIELDS[1]])
def test_rate_form_uniaxial_by_dual_numbers(yield_kind, kw, idx, plastic, solver_variant):
    """small_rate_elastic_plastic under UNIAXIAL_STRESS (12 local dofs) by dual-number evaluation (host build)."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_rate_uniaxial_dual(hh.hessians, yield_kind, kw, idx, plastic)



@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_form_tangent(def_type, yield_kind, kw, rot, solver_variant):
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("rate form always uses the dense path")
    pc.check_rate_tangent(lambda desc, info, g, gp, xp, x: hh.tangent_rate(desc, g, gp, xp, x), def_type, yield_kind, kw, rot, B=192)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_form_vjp(def_type, yield_kind, kw, rot, solver_variant):
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("rate form always uses the dense path")
    pc.check_rate_vjp(lambda desc, info, g, gp, xp, x, sb: hh.vjp_rate(desc, g, gp, xp, x, sb), def_type, yield_kind, kw, rot, B=192)


def test_exp_s():
    """cm::exp_s (the exponential of the hardening laws, scalar-register coefficients on the device) against libm:
    below 1 ulp on the working range, monotone limits outside."""
    import ctypes as C
    import numpy as np
    import host_harness_lib as hh
    L = hh.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40., 40., 200000), rng.uniform(-1., 1., 200000), -np.logspace(-18, 2.8, 2000),
                        np.array([0., -0., 1e-300, -1e-300, 709.7, -745.1, -800., 800., 1e6, -1e6, np.inf, -np.inf])])
    y = np.zeros_like(x)
    L.hh_exp_s(C.c_int64(x.size), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    with np.errstate(over="ignore"):
        ref = np.exp(np.clip(x, -1100., 1100.))
    fin = np.isfinite(ref) & (ref > 1e-300)
    ulp = np.abs(y[fin] - ref[fin]) / np.spacing(ref[fin])
    assert ulp.max() <= 1.0, ulp.max()
    assert np.mean(ulp == 0) > 0.7
    assert (y[x > 710.] == np.inf).all() and (y[x < -746.] == 0.).all()


def test_softplus_pieces():
    """cm::log1p_01 (log(1 + t), t in [0, 1]) and cm::soft_unit (softplus and sigmoid of the network surfaces) against
    libm / the stable closed forms."""
    import ctypes as C
    import numpy as np
    import host_harness_lib as hh
    L = hh.lib()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.uniform(0., 1., 300000), np.logspace(-320, 0, 3000), np.array([0., 1., 0.41421356237309503, 0.4142135623730951])])
    y = np.zeros_like(t)
    L.hh_log1p_01(C.c_int64(t.size), vp(t), vp(y))
    ref = np.log1p(t)
    nz = ref > 0
    assert (y[~nz] == 0.).all()
    ulp = np.abs(y[nz] - ref[nz]) / np.spacing(ref[nz])
    assert ulp.max() <= 2.0, ulp.max()
    a = np.concatenate([rng.uniform(-40., 40., 200000), rng.uniform(-2., 2., 100000), np.array([0., -0., 745., -745., 1e4, -1e4])])
    sp, sg = np.zeros_like(a), np.zeros_like(a)
    L.hh_soft_unit(C.c_int64(a.size), vp(a), vp(sp), vp(sg))
    with np.errstate(over="ignore"):
        sp_ref = np.logaddexp(a, 0.)
        sg_ref = np.where(a >= 0, 1. / (1. + np.exp(-np.abs(a))), np.exp(-np.abs(a)) / (1. + np.exp(-np.abs(a))))
    np.testing.assert_allclose(sp, sp_ref, rtol=1e-15, atol=1e-320)
    np.testing.assert_allclose(sg, sg_ref, rtol=1e-15, atol=1e-320)
    # cm::log_pos: log of any positive normal double
    x = np.concatenate([np.exp(rng.uniform(-700., 700., 300000)), rng.uniform(0.5, 2.0, 100000),
                        np.array([1.0, 0.7071067811865476, 0.7071067811865475, 1.4142135623730951, 2.0, 1e-300, 1e300])])
    y = np.zeros_like(x)
    L.hh_log_pos(C.c_int64(x.size), vp(x), vp(y))
    ref = np.log(x)
    assert y[x == 1.0].any() == False
    nz = ref != 0
    ulp = np.abs(y[nz] - ref[nz]) / np.spacing(np.abs(ref[nz]))
    assert ulp.max() <= 2.0, ulp.max()


def test_icnn_one_exponential_per_unit():
    """cm::icnn_symmetric (both signs of a hidden unit from one exponential, one logarithm, one reciprocal; table exp(b0)
    appended to the weight pack) against the plain formulas f(x) + f(-x), its gradient and Hessian in numpy -- moderate inputs,
    saturated units beyond the clamp, and a unit with |b0| above the fast form's range (two-sided fallback)."""
    import ctypes as C
    import numpy as np
    import host_harness_lib as hh
    L = hh.lib()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(5)
    H = 9
    W0 = rng.normal(size=(6, H)); b0 = rng.normal(size=H) * 2.0; b0[3] = 160.0; b0[5] = -155.0; b0[7] = 40.0
    Wx1 = rng.normal(size=6); b1 = 0.3; Wz = np.abs(rng.normal(size=H))
    with np.errstate(over="ignore"):
        rec = np.column_stack([W0.T, b0, Wz, np.exp(b0), np.exp(b0) * Wz]).ravel()      # cmad_amd.models.device.unit_records
        pack = np.concatenate([W0.ravel(), b0, Wx1, [b1], Wz, np.ones(6), np.zeros(6), [1.0, 0.0, 0.0], rec])
    n = 4000
    xs = rng.normal(size=(n, 6)) * rng.choice([0.0, 1e-9, 0.3, 3.0, 60.0, 400.0], size=(n, 1))
    F, G, Hx = np.zeros(n), np.zeros((n, 6)), np.zeros((n, 21))
    L.hh_icnn_symmetric(vp(pack), C.c_int(H), C.c_int64(n), vp(np.ascontiguousarray(xs)), vp(F), vp(G), vp(Hx))
    T = np.zeros((n, H))                                                          # the kernel's summation order (the large
    for i in range(6):                                                            # inputs cancel to O(1) arguments)
        T = T + xs[:, i:i + 1] * W0[i][None, :]
    sp = lambda a: np.logaddexp(a, 0.0)
    with np.errstate(over="ignore"):
        sg = lambda a: np.where(a >= 0, 1. / (1. + np.exp(-np.abs(a))), np.exp(-np.abs(a)) / (1. + np.exp(-np.abs(a))))
        Fr = 2 * b1 + ((sp(b0 + T) + sp(b0 - T)) * Wz).sum(1)
        c1 = (sg(b0 + T) - sg(b0 - T)) * Wz
        c2 = (sg(b0 + T) * (1 - sg(b0 + T)) + sg(b0 - T) * (1 - sg(b0 - T))) * Wz
    Gr = c1 @ W0.T
    Hr = np.einsum("no,io,jo->nij", c2, W0, W0)
    scale = (np.abs(sp(b0 + T)) + np.abs(sp(b0 - T))) @ Wz + 1.0                   # every term to ~1e-16 of the sum of magnitudes
    assert (np.abs(F - Fr) <= 4e-15 * scale).all(), (np.abs(F - Fr) / scale).max()
    gs = (np.abs(W0)[None] * Wz[None, None]).sum(2).max()
    assert np.abs(G - Gr).max() <= 2e-15 * gs
    iu = np.triu_indices(6)
    assert np.abs(Hx - Hr[:, iu[0], iu[1]]).max() <= 2e-15 * max(1.0, np.abs(Hr).max())


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_history_second_order_vs_oracle(def_type, yield_kind, kw, rate, solver_variant):
    """cm_adjoint_history / cm_direct_history / cm_hessian_history (host build of their per-point code) against the
    oracle-assembled gradient, adjoint vectors, forward sensitivities and Hessian."""
    from cmad_amd.models.device import build_desc
    from host_facade import HostHistoryEngine
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_history_second_order(lambda desc, info: HostHistoryEngine(desc=desc, info=info),
                                  lambda values, dt, mk: build_desc(values, def_type=dt, model_kind=mk),
                                  def_type, yield_kind, kw, rate=rate)


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_extended_parameter_blocks(def_type, yield_kind, kw, rate, solver_variant):
    """cm_param_blocks (host build of cm::param_direction): rotation-matrix entries, Hosford exponent, native parameters by
    forward-mode evaluation of the whole model against the oracle's AD; the rate form under UNIAXIAL_STRESS included."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_param_blocks(hh.param_blocks, def_type, yield_kind, kw, rate=rate)


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_extended_parameter_blocks_network(def_type, scaled, solver_variant):
    """Hill coefficients and network weights of the hybrid surfaces (plain and beta-rescaled)."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_param_blocks_network(hh.param_blocks, def_type, scaled=scaled)


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_second_derivatives_network_surfaces(def_type, scaled, solver_variant):
    """cm_hessians for the hybrid Hill + network yield surfaces (host build) against the oracle's nested duals."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_second_derivs_network(hh.hessians, def_type, scaled=scaled)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_barlat_second_derivatives_and_coefficient_sensitivities(def_type, solver_variant):
    """Barlat Yld2004-18p in the arithmetic-T model (host build): Hessians vs the oracle, coefficient sensitivities vs FD."""
    import host_harness_lib as hh
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    pc.check_barlat_generic(hh.hessians, hh.param_blocks, def_type)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_form_with_a_dense_yield_surface(def_type, solver_variant):
    """small_rate_elastic_plastic takes any effective stress (cmad/models/small_rate_elastic_plastic.py:116-126): the rate form
    with Barlat Yld2004-18p through every per-point routine the kernels are made of -- update (three load steps), tangent,
    reverse sweep, second derivatives, whole-history gradient / adjoint vectors / forward sensitivities / Hessian -- against
    the oracle (FULL_3D, PLANE_STRESS; UNIAXIAL_STRESS, the 12-dof form: update, tangent, reverse sweep)."""
    import host_harness_lib as hh
    from cmad_amd.models.device import build_desc
    from host_facade import HostHistoryEngine
    if solver_variant != "structured":
        pytest.skip("not solver dependent")
    yk, kw = pc.BARLAT
    nx = {0: 7, 2: 8, 3: 12}
    pc.check_rate_model(lambda desc, info, g, gp, xp: hh.update_rate(desc, g, gp, xp, nx[desc.def_type]), def_type, yk, kw, True, False, B=96)
    pc.check_rate_tangent(lambda desc, info, g, gp, xp, x: hh.tangent_rate(desc, g, gp, xp, x), def_type, yk, kw, True, B=64)
    pc.check_rate_vjp(lambda desc, info, g, gp, xp, x, sb: hh.vjp_rate(desc, g, gp, xp, x, sb), def_type, yk, kw, True, B=64)
    if def_type != ol.UNIAXIAL_STRESS:
        pc.check_rate_second_derivs(hh.hessians, hh.evaluate_rate, def_type, yk, kw, True)
        pc.check_history_second_order(lambda desc, info: HostHistoryEngine(desc=desc, info=info),
                                      lambda values, dt, mk: build_desc(values, def_type=dt, model_kind=mk),
                                      def_type, yk, kw, rate=True)


@pytest.mark.parametrize("hidden", [None, [4, 3], [3, 2, 4]])
@pytest.mark.parametrize("with_voce", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_network_hardening_law(def_type, with_voce, hidden, solver_variant):
    import host_harness_lib as hh
    if def_type == ol.UNIAXIAL_STRESS and solver_variant != "structured":
        pytest.skip("one solver form under UNIAXIAL_STRESS")
    if def_type == ol.UNIAXIAL_STRESS:
        pytest.skip("covered through the facade")
    pc.check_nn_hardening(BACKEND, hh.param_blocks, def_type, with_voce=with_voce, B=192, hidden=hidden)


DEEP = (6, 7, 5, 1)            # two hidden layers (the reference's forward loops over any number, input_convex_neural_network.py:58-69)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_hybrid_surface_with_a_multi_layer_network(def_type, solver_variant):
    """Hybrid Hill + ICNN with TWO hidden layers: the general network evaluation (cm::icnn_symmetric_deep: forward pass with
    per-unit pre-activation gradients, backward pass for the Hessian weights) in the update and the reverse sweep against
    the oracle, the weight sensitivities (arithmetic-T model, backward-pass gradient) against central differences, and the
    second derivatives against the oracle's nested duals."""
    import host_harness_lib as hh
    pc.check_hybrid_nn(BACKEND, def_type, B=96, rot=(def_type == ol.PLANE_STRESS), widths=DEEP)
    pc.check_hybrid_nn(BACKEND, def_type, B=64, rot=(def_type == ol.PLANE_STRESS), widths=DEEP, scaled=True)     # beta-rescaled around it
    if solver_variant == "structured":
        for scaled in (False, True):
            pc.check_param_blocks_network(hh.param_blocks, def_type, scaled=scaled, layer_widths=DEEP)
            pc.check_second_derivs_network(hh.hessians, def_type, scaled=scaled, layer_widths=DEEP)


@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_plain_newton_through_the_line_search_kernels(def_type, yield_kind, kw, solver_variant):
    """The library builds Hosford, the dense surfaces, the rate form and the HNN build with the line-search Newton loops only;
    with ls_max_evals == 0 those loops take the full step without a merit test (uniform branch).  Same states, stresses and
    status words, bit for bit, as the LS = false instantiations -- structured, dense and by passes, total and rate form
    (UNIAXIAL_STRESS rate form: cm::ru_newton)."""
    import numpy as np
    import host_harness_lib as hh
    sc = pc.Scenario(def_type, yield_kind, kw, True, False, B=384)
    runs = [lambda: BACKEND.update(sc, sc.gradu, sc.xi1)]
    if yield_kind in ("J2", "hill", "hosford"):
        for dt in ((def_type, ol.UNIAXIAL_STRESS) if def_type == ol.FULL_3D else (def_type,)):
            def rate_run(dt=dt):
                outs = []
                def upd(desc, info, g, gp, xp):
                    r = hh.update_rate(desc, g, gp, xp, {0: 7, 2: 8, 3: 12}[desc.def_type])
                    outs.extend(r)
                    return r
                pc.check_rate_model(upd, dt, yield_kind, kw, False, False, B=128)       # three load steps, checked against the oracle
                return outs
            runs.append(rate_run)
    ref = [r() for r in runs]
    hh.set_force_ls(True)
    try:
        got = [r() for r in runs]
    finally:
        hh.set_force_ls(False)
    for a, b in zip(ref, got):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "hill", {"hill": pc.HILL}), (ol.PLANE_STRESS, "J2", {}), (ol.PLANE_STRESS, "hill", {"hill": pc.HILL}),
                                                    (ol.FULL_3D, "hosford", {"a": 20.0}), (ol.FULL_3D, "hosford", {"a": 64.0}),
                                                    (ol.UNIAXIAL_STRESS, "J2", {}), (ol.UNIAXIAL_STRESS, "hill", {"hill": pc.HILL})])
def test_warm_started_newton_against_the_oracle(def_type, yield_kind, kw, rot, ls, solver_variant):
    """Scalar return maps / analytic warm starts (the default of the batched entry points) against the oracle's general Newton."""
    if yield_kind == "hosford" and not ls:
        pytest.skip("plain Newton from x_prev does not converge for large Hosford exponents: nothing to compare with")
    if solver_variant != "structured":
        pytest.skip("the warm starts are variants of the structured solver")
    pc.check_warm_start(BACKEND, def_type, yield_kind, kw, rot, ls, B=512, uniaxial_idx=2 if rot else 1)


@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "hill", {"hill": pc.HILL}), (ol.PLANE_STRESS, "J2", {}), (ol.PLANE_STRESS, "hill", {"hill": pc.HILL}),
                                                    (ol.FULL_3D, "hosford", {"a": 100.0}), (ol.FULL_3D, "hosford", {"a": 20.0}),
                                                    (ol.UNIAXIAL_STRESS, "J2", {}), (ol.UNIAXIAL_STRESS, "hill", {"hill": pc.HILL})])
def test_warm_started_newton_edge_cases(def_type, yield_kind, kw, solver_variant):
    """Zero / volumetric strains, 20- and 200-yield-strain increments and a step from a heavily hardened state on the default route."""
    if solver_variant != "structured":
        pytest.skip("the warm starts are variants of the structured solver")
    pc.check_warm_start_edge_cases(BACKEND, def_type, yield_kind, kw)
