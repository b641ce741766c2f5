"""GPU parity of the sensitivity kernels (tangent, vjp, fused update+vjp, fused objective+grad, adjoint
step) against the CPU oracle's dual-number AD, through the C-ABI (`-m gpu`)."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_cases as pc

pytestmark = pytest.mark.gpu
CASES = pc.SENS_YIELDS
IDX9 = [0, 1, 2, 1, 3, 4, 2, 4, 5]


@pytest.fixture(scope="module")
def backend():
    return pc.GpuBackend()


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", CASES)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_tangent(backend, def_type, yield_kind, kw, rot):
    pc.check_tangent(backend, pc.Scenario(def_type, yield_kind, kw, rot, False, B=512))


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", CASES)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_vjp_and_fused(backend, def_type, yield_kind, kw, rot):
    sc = pc.Scenario(def_type, yield_kind, kw, rot, False, B=1000)
    sbar, ref = pc.check_vjp(backend, sc)
    # fused update + vjp gives the same numbers and the same state
    t = backend.t
    xi_f, sig_f, g_f = backend.ev(sc).update_and_vjp(t(sc.gradu), t(sc.xi1), t(sbar))
    np.testing.assert_allclose(xi_f.cpu().numpy(), sc.xi2, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(sig_f.cpu().numpy(), sc.sig2, rtol=1e-10, atol=1e-8)
    got_f = np.array([__import__("cmad_amd.models.device", fromlist=["x"]).kp_to_leaf_grad(p, g_f.cpu().numpy(), sc.info)
                      for p in pc.param_paths(yield_kind)])
    np.testing.assert_allclose(got_f, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", CASES)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_objective_grad_single_step(backend, def_type, yield_kind, kw, rot):
    from cmad_amd.models.device import fold_weight_and_data
    sc = pc.Scenario(def_type, yield_kind, kw, rot, False, B=1000)
    mat, B, t = sc.mat, sc.B, backend.t
    rng = np.random.default_rng(7)
    data6 = sc.sig2 + rng.normal(0., 5., size=sc.sig2.shape)
    w = np.zeros((3, 3)); w[0, 0] = 1.; w[1, 1] = 1.; w[0, 1] = 0.5; w[1, 0] = 0.5
    gh = np.stack([np.zeros((mat.nu, B)), sc.gradu])           # oracle: K = 1 history
    dh = np.stack([np.zeros((9, B)), data6[IDX9, :]])
    J_o, g_o, Jb, xk = mat.objective_grad_batch(sc.st_o, gh, dh, w, sc.xi1)
    res, xi_d = backend.ev(sc).objective_grad(t(sc.gradu), t(sc.xi1), t(data6), fold_weight_and_data(w), want_xi=True)
    res = res.cpu().numpy()
    np.testing.assert_allclose(xi_d.cpu().numpy(), xk, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(res[0], J_o, rtol=1e-10)
    got, ref = pc.leaf_grads(res[1:], sc.info, mat, yield_kind, g_o)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("yield_kind,kw", CASES[:2])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_adjoint_history(backend, def_type, yield_kind, kw):
    """K-step history: forward cm_update per step, reverse cm_adjoint_step per step
    (cmad/objectives/mp_objective.py:95-147) vs the oracle's adjoint."""
    import torch
    from cmad_amd.models.device import fold_weight_and_data
    from cmad_amd.synthetic import gauss_point_batch
    K = 4
    sc = pc.Scenario(def_type, yield_kind, kw, True, False, B=600)
    mat, B, t, ev = sc.mat, sc.B, backend.t, backend.ev(sc)
    base = gauss_point_batch(B, seed=3, skew=False, ndims=sc.nd)
    gh = np.stack([k * 0.6 * base for k in range(K + 1)])
    rng = np.random.default_rng(11)
    xs, sigs = [sc.xi0], [np.zeros((6, B))]
    for k in range(1, K + 1):
        x, s, _, cv = mat.update_batch(sc.st_o, gh[k], xs[-1])
        assert cv.all()
        xs.append(x); sigs.append(s)
    data6 = [s + rng.normal(0., 5., size=s.shape) for s in sigs]
    dh = np.stack([d[IDX9, :] for d in data6])
    w = np.eye(3)
    J_o, g_o, _, _ = mat.objective_grad_batch(sc.st_o, gh, dh, w, sc.xi0)
    wsq6 = fold_weight_and_data(w)
    xd = [t(sc.xi0)]
    for k in range(1, K + 1):
        x, _, _ = ev.update(t(gh[k]), xd[-1], want_sigma=False, want_status=False)
        xd.append(x)
    out = torch.zeros(13, dtype=torch.float64, device="cuda")
    hist = torch.zeros((mat.nx, B), dtype=torch.float64, device="cuda")
    for k in range(K, 0, -1):
        ev.adjoint_step(t(gh[k]), xd[k - 1], xd[k], t(data6[k]), wsq6, hist, hist, out, accumulate=True)
    res = out.cpu().numpy()
    np.testing.assert_allclose(res[0], J_o, rtol=1e-10)
    got, ref = pc.leaf_grads(res[1:], sc.info, mat, yield_kind, g_o)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-11 * np.abs(ref).max())


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_form_tangent(def_type, yield_kind, kw, rot):
    """cm_update_rate_tangent: the update reproduces the oracle's state and the tangent its IFT Jacobian."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator

    def run(desc, info, g, gp, xp, x_expected):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        xi, sig, st, ds = DeviceEvaluator(desc, info).update_rate(t(g), t(gp), t(xp), tangent=True)
        np.testing.assert_allclose(xi.cpu().numpy()[:6], x_expected[:6], rtol=1e-10, atol=1e-7)
        return ds.cpu().numpy()
    pc.check_rate_tangent(run, def_type, yield_kind, kw, rot, B=1024)


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_random_materials_rate_form(def_type, yield_kind, ls):
    """The rate-form model's kernels (structured solver through the change of variables of newton_s_rate) for randomly drawn
    materials: update over three load steps, tangent, reverse sweep against the oracle."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def update(desc, info, g, gp, xp):
        xi, sig, st = DeviceEvaluator(desc, info).update_rate(t(g), t(gp), t(xp))
        return xi.cpu().numpy(), sig.cpu().numpy(), st.cpu().numpy().astype(np.uint32)

    def tangent(desc, info, g, gp, xp, x_expected):
        return DeviceEvaluator(desc, info).update_rate(t(g), t(gp), t(xp), tangent=True)[3].cpu().numpy()

    def vjp(desc, info, g, gp, xp, x, sbar):
        gk, xb, ub = DeviceEvaluator(desc, info).update_vjp(t(g), t(xp), t(x), t(sbar), want_xi_prev_bar=True, want_gradu_bar=True,
                                                            gradu_prev=t(gp))
        return gk.cpu().numpy(), xb.cpu().numpy(), ub.cpu().numpy()
    pc.check_random_materials_rate(update, tangent, vjp, def_type, yield_kind, ls, seeds=range(4), B=320)


def _rate_case(def_type, yield_kind, kw, rot, B, seed=22):
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    from cmad_amd.synthetic import gauss_point_batch
    rng = np.random.default_rng(seed)
    values = ol.j2_voce_values(yield_kind=yield_kind, Q=pc.rand_rot(rng) if rot else None, **kw)
    st_o, st_d = pc.settings_pair(False)
    mat = ol.Material(values, def_type=def_type, model_kind=ol.SMALL_RATE_EP)
    desc, info = build_desc(values, def_type=def_type, model_kind=1, newton=st_d)
    g0 = gauss_point_batch(B, seed=seed, skew=True, ndims=3 if def_type == ol.FULL_3D else 2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return mat, st_o, DeviceEvaluator(desc, info), info, g0, t


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_form_vjp_and_fused(def_type, yield_kind, kw, rot):
    """cm_update_rate_vjp vs the oracle's reverse sweep; cm_update_rate_and_vjp gives the same numbers from
    xi_prev alone (cmad/models/small_rate_elastic_plastic.py under cmad/objectives/mp_objective.py:95-147)."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator
    seen = {}

    def run(desc, info, g, gp, xp, x, sbar):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        ev = DeviceEvaluator(desc, info)
        gk, xb, ub = ev.update_vjp(t(g), t(xp), t(x), t(sbar), want_xi_prev_bar=True, want_gradu_bar=True, gradu_prev=t(gp))
        xi_f, sig_f, g_f = ev.update_and_vjp(t(g), t(xp), t(sbar), gradu_prev=t(gp))
        np.testing.assert_allclose(xi_f.cpu().numpy()[:6], x[:6], rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(xi_f.cpu().numpy()[6:], x[6:], rtol=1e-10, atol=1e-12)
        _, sig_u, _ = ev.update_rate(t(g), t(gp), t(xp))
        np.testing.assert_allclose(sig_f.cpu().numpy(), sig_u.cpu().numpy(), rtol=1e-12, atol=1e-9)
        seen["fused"], seen["split"] = g_f.cpu().numpy(), gk.cpu().numpy()
        return gk.cpu().numpy(), xb.cpu().numpy(), ub.cpu().numpy()
    pc.check_rate_vjp(run, def_type, yield_kind, kw, rot, B=1000)
    np.testing.assert_allclose(seen["fused"], seen["split"], rtol=1e-8, atol=1e-11 * np.abs(seen["split"]).max())


@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_rate_form_adjoint_history(def_type, yield_kind, kw):
    """K-step calibration history of the rate form: cm_update_rate forward + cm_adjoint_step_rate backward, and
    the single-step fused cm_objective_grad_rate, vs the oracle's adjoint (previous grad u = the history's
    previous step)."""
    import torch
    from cmad_amd.models.device import fold_weight_and_data
    K, B = 4, 600
    mat, st_o, ev, info, g0, t = _rate_case(def_type, yield_kind, kw, True, B)
    gh = np.stack([k * 0.5 * g0 for k in range(K + 1)])
    xi0 = np.tile(mat.init_xi()[:, None], (1, B))
    xs, plastic = [xi0], 0.0
    for k in range(1, K + 1):
        x, s, it, cv = mat.update_batch(st_o, gh[k], xs[-1], gradu_prev=gh[k - 1])
        assert cv.all()
        plastic = max(plastic, (it > 0).mean())
        xs.append(x)
    assert plastic > 0.2
    rng = np.random.default_rng(11)
    data6 = [x[:6] + rng.normal(0., 5., size=(6, B)) for x in xs]
    dh = np.stack([d[IDX9, :] for d in data6])
    w = np.zeros((3, 3)); w[0, 0] = 1.; w[1, 1] = 1.; w[0, 1] = 0.5; w[1, 0] = 0.5; w[2, 2] = 0.25
    if def_type == ol.PLANE_STRESS:
        w[2, 2] = 0.
    J_o, g_o, _, xk = mat.objective_grad_batch(st_o, gh, dh, w, xi0)
    wsq6 = fold_weight_and_data(w)
    xd = [t(xi0)]
    for k in range(1, K + 1):
        x, _, _ = ev.update_rate(t(gh[k]), t(gh[k - 1]), xd[-1], want_sigma=False, want_status=False)
        xd.append(x)
    np.testing.assert_allclose(xd[-1].cpu().numpy()[:6], xk[:6], rtol=1e-10, atol=1e-7)
    out = torch.zeros(13, dtype=torch.float64, device="cuda")
    hist = torch.zeros((mat.nx, B), dtype=torch.float64, device="cuda")
    for k in range(K, 0, -1):
        ev.adjoint_step(t(gh[k]), xd[k - 1], xd[k], t(data6[k]), wsq6, hist, hist, out, accumulate=True,
                        gradu_prev=t(gh[k - 1]))
    res = out.cpu().numpy()
    np.testing.assert_allclose(res[0], J_o, rtol=1e-10)
    got, ref = pc.leaf_grads(res[1:], info, mat, yield_kind, g_o)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-11 * np.abs(ref).max())
    # single step, fused (state enters only through xi_prev and the two grad u)
    J1, g1, _, x1 = mat.objective_grad_batch(st_o, gh[2:4], dh[2:4], w, xs[2])
    res1, xi1 = ev.objective_grad(t(gh[3]), xd[2], t(data6[3]), wsq6, want_xi=True, gradu_prev=t(gh[2]))
    res1 = res1.cpu().numpy()
    np.testing.assert_allclose(xi1.cpu().numpy()[:6], x1[:6], rtol=1e-10, atol=1e-7)
    np.testing.assert_allclose(res1[0], J1, rtol=1e-9)
    got, ref = pc.leaf_grads(res1[1:], info, mat, yield_kind, g1)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-11 * np.abs(ref).max())


@pytest.mark.parametrize("uidx", [0, 1, 2])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
def test_uniaxial_stress_sensitivities(backend, yield_kind, kw, rot, uidx):
    """UNIAXIAL_STRESS (9 local dofs, one grad-u entry): update, IFT tangent, reverse sweep, fused update + vjp and
    the single-step fused objective vs the oracle.  (d/d nu vanishes analytically here -- the axial stress of a
    uniaxial-stress state does not see nu -- and is left with the round-off of O(1e6)-scaled cancelling lambda / mu
    terms, hence the absolute tolerance relative to the largest gradient entry.)"""
    from cmad_amd.models.device import fold_weight_and_data, kp_to_leaf_grad
    sc = pc.Scenario(ol.UNIAXIAL_STRESS, yield_kind, kw, rot, False, B=1000, uniaxial_idx=uidx)
    pc.check_update(backend, sc)
    pc.check_tangent(backend, sc)
    sbar, ref = pc.check_vjp(backend, sc, grad_atol=1e-9)
    t = backend.t
    xi_f, sig_f, g_f = backend.ev(sc).update_and_vjp(t(sc.gradu), t(sc.xi1), t(sbar))
    np.testing.assert_allclose(xi_f.cpu().numpy(), sc.xi2, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(sig_f.cpu().numpy(), sc.sig2, rtol=1e-10, atol=1e-8)
    got_f = np.array([kp_to_leaf_grad(p, g_f.cpu().numpy(), sc.info) for p in pc.param_paths(yield_kind)])
    np.testing.assert_allclose(got_f, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    # objective: the axial stress entry only (UniaxialCalibration's stress term, qois/uniaxial_calibration.py)
    mat, B = sc.mat, sc.B
    data6 = sc.sig2 + np.random.default_rng(7).normal(0., 5., size=sc.sig2.shape)
    w = np.zeros((3, 3)); w[uidx, uidx] = 1.
    gh = np.stack([np.zeros((1, B)), sc.gradu])
    dh = np.stack([np.zeros((9, B)), data6[IDX9, :]])
    J_o, g_o, _, xk = mat.objective_grad_batch(sc.st_o, gh, dh, w, sc.xi1)
    res, xi_d = backend.ev(sc).objective_grad(t(sc.gradu), t(sc.xi1), t(data6), fold_weight_and_data(w), want_xi=True)
    res = res.cpu().numpy()
    np.testing.assert_allclose(xi_d.cpu().numpy(), xk, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(res[0], J_o, rtol=1e-10)
    got, ref = pc.leaf_grads(res[1:], sc.info, mat, yield_kind, g_o)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())


def _gpu_history(desc, info, gh, d6, wsq6, xi0):
    import torch
    from cmad_amd.models.device import DeviceEvaluator
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out, xi_hist = DeviceEvaluator(desc, info).objective_grad_history(t(gh), t(d6), wsq6, t(xi0))
    return out.cpu().numpy(), xi_hist.cpu().numpy()


def _gpu_primal(desc, info, gh, xi0):
    import torch
    from cmad_amd.models.device import DeviceEvaluator
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return tuple(o.cpu().numpy() for o in DeviceEvaluator(desc, info).update_history(t(gh), t(xi0)))


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_history_objective_grad(def_type, yield_kind, kw, rot, rate):
    """cm_objective_grad_history: the whole K-step history per point in one launch vs the oracle's adjoint
    (cmad/objectives/mp_objective.py:95-147), and cm_update_history (the forward pass alone: states, stresses, iteration
    counts per step); block tail (B not a multiple of the block) included."""
    # (the 7-step path ends with a large reversal that the rate form's plain Newton does not survive at every point)
    pc.check_history(_gpu_history, def_type, yield_kind, kw, rot, rate=rate, B=1000, K=5 if rate else 7, uniaxial_idx=2,
                     primal=_gpu_primal)


def test_history_objective_grad_variants():
    """Line search variants, the hybrid network and Barlat surfaces, and the fused history against the per-step
    launches of BatchedCalibrationObjective."""
    pc.check_history(_gpu_history, ol.FULL_3D, "J2", {}, False, ls=True, B=700, primal=_gpu_primal)
    pc.check_history(_gpu_history, ol.FULL_3D, "J2", {}, True, ls=True, B=700, primal=_gpu_primal, solver_flags=2)
    pc.check_history(_gpu_history, ol.FULL_3D, "J2", {}, False, B=700, primal=_gpu_primal, solver_flags=2)
    pc.check_history(_gpu_history, ol.PLANE_STRESS, "hill", pc.YIELDS[1][1], True, ls=True, B=700, primal=_gpu_primal)
    pc.check_history(_gpu_history, ol.PLANE_STRESS, "hosford", pc.YIELDS[2][1], True, rate=True, ls=True, B=700, primal=_gpu_primal)


def test_history_entries_edge_cases():
    """Empty batch (J = 0, grad = 0, nothing read), operand checks before any launch, and an iteration cap that leaves
    steps unconverged: like the reference (Newton failure is not an error) the history carries the last iterate on,
    exactly as the same number of single-step calls does, and reports it in the per-step status."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc, fold_weight_and_data
    from cmad_amd.synthetic import gauss_point_batch
    values = ol.j2_voce_values()
    desc, info = build_desc(values)
    ev = DeviceEvaluator(desc, info)
    f64 = dict(dtype=torch.float64, device="cuda")
    wsq6 = fold_weight_and_data(np.eye(3))
    out, xh = ev.objective_grad_history(torch.empty((4, 9, 0), **f64), torch.empty((4, 6, 0), **f64), wsq6, torch.empty((7, 0), **f64))
    assert xh.shape == (4, 7, 0) and not out.cpu().numpy().any()
    xh, sh, st = ev.update_history(torch.empty((4, 9, 0), **f64), torch.empty((7, 0), **f64))
    assert xh.shape == (4, 7, 0) and sh.shape == (4, 6, 0) and st.shape == (4, 0)
    g = torch.zeros((3, 9, 8), **f64)
    with pytest.raises(ValueError):
        ev.update_history(g[:1], torch.zeros((7, 8), **f64))                       # no step after the initial one
    with pytest.raises(ValueError):
        ev.update_history(g, torch.zeros((8, 8), **f64))                           # wrong n_xi
    with pytest.raises(ValueError):
        ev.objective_grad_history(g, torch.zeros((2, 6, 8), **f64), wsq6, torch.zeros((7, 8), **f64))   # K mismatch
    with pytest.raises(ValueError):
        ev.objective_grad_history(g.transpose(1, 2), torch.zeros((3, 6, 8), **f64), wsq6, torch.zeros((7, 8), **f64))
    # combinations without a kernel are refused (CM_ERR_UNSUPPORTED -> NotImplementedError) before any launch, never a silent
    # no-op: a network surface with a width the kernels do not implement, on entry points of both model kinds
    z7, z6 = torch.zeros((7, 8), **f64), torch.zeros((3, 6, 8), **f64)
    for mk in (0, 1):
        desc_b, info_b = build_desc(ol.j2_voce_values(), model_kind=mk)
        desc_b.yield_kind = 3                                       # hybrid Hill + network ...
        desc_b.nn_nlayers, desc_b.nn_weights = 0, None              # ... without a network
        ev_b = DeviceEvaluator(desc_b, info_b)
        kw_prev = {"gradu_prev": g[0]} if mk else {}
        with pytest.raises(NotImplementedError):
            ev_b.objective_grad_history(g, z6, wsq6, z7)
        with pytest.raises(NotImplementedError):
            ev_b.update_history(g, z7)
        with pytest.raises(NotImplementedError):
            ev_b.direct_step(g[1], z7, z7, **kw_prev)
        with pytest.raises(NotImplementedError):
            ev_b.update_vjp(g[1], z7, z7, torch.zeros((6, 8), **f64), **kw_prev)
    # iteration cap 1: unconverged iterates are carried from step to step, identically to per-step calls
    B, K = 300, 3
    desc1, info1 = build_desc(values, newton=NewtonSettings(max_iters=1))
    ev1 = DeviceEvaluator(desc1, info1)
    g0 = gauss_point_batch(B, seed=9, dev_scale=8.0)
    gh = torch.from_numpy(np.stack([c * g0 for c in (0., 0.6, 1.0, 1.3)])).cuda()
    xh, sh, st = ev1.update_history(gh, torch.zeros((7, B), **f64))
    x = torch.zeros((7, B), **f64)
    for k in range(1, K + 1):
        x, s, stk = ev1.update(gh[k], x)
        np.testing.assert_allclose(xh[k].cpu().numpy(), x.cpu().numpy(), rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(sh[k].cpu().numpy(), s.cpu().numpy(), rtol=1e-10, atol=1e-9)
        mask = 0xFFFF | (1 << 16) | (1 << 18)
        assert torch.equal(st[k] & mask, stk & mask)
    assert (((st[1:].cpu().numpy().astype(np.uint32) >> 16) & 1) == 0).any()


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_direct_sensitivities(def_type, yield_kind, kw, rot, rate):
    """cm_direct_step: forward parameter sensitivities dxi/dp, dsigma/dp propagated over a history
    (cmad/objectives/mp_objective.py:158-215) vs the oracle's Jacobians, and the objective gradient they give vs the
    oracle's adjoint."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def direct(desc, info, g, gp, xp, x, dxp):
        dx, ds = DeviceEvaluator(desc, info).direct_step(t(g), t(xp), t(x), dxi_prev_dp=t(dxp), gradu_prev=t(gp))
        return dx.cpu().numpy(), ds.cpu().numpy()
    pc.check_direct(direct, def_type, yield_kind, kw, rot, rate=rate, B=300, uniaxial_idx=0)


# ---- second derivatives: cm_hessians / cm_hessians_rate on the GPU against the oracle's nested duals -------------------
# (reference cmad/models/model.py:133-147, 245-270; pinned there by tests/objectives/test_jvp_vs_original.py:97)
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("plastic", [True, False])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_second_derivatives_vs_oracle(def_type, yield_kind, kw, plastic, rot):
    """cm_hessians through the C-ABI: d2C, d2 sigma (every block w.r.t. xi, xi_prev, params) and the first derivatives of
    the same pass (dC, d sigma) against `orc_second_derivs` / `orc_jacobian` and against the hand-derived cm_evaluate blocks."""
    import gpu_api
    pc.check_second_derivs(gpu_api.hessians, gpu_api.evaluate, def_type, yield_kind, kw, plastic, rot=rot)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("plastic", [True, False])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_rate_form_second_derivatives_vs_oracle(def_type, yield_kind, kw, plastic, rot):
    """cm_hessians_rate through the C-ABI against the oracle, both branches."""
    import gpu_api
    pc.check_rate_second_derivs(gpu_api.hessians, gpu_api.evaluate_rate, def_type, yield_kind, kw, plastic, rot=rot)


@pytest.mark.parametrize("plastic", [True, False])
@pytest.mark.parametrize("idx", [0, 1, 2])
@pytest.mark.parametrize("yield_kind,kw", [pc.YIELDS[0], pc.YIELDS[1]])
def test_rate_form_uniaxial_by_dual_numbers(yield_kind, kw, idx, plastic):
    """The 12-dof rate form under UNIAXIAL_STRESS (cm_hessians_rate: residual, stress, first and second derivatives)."""
    import gpu_api
    pc.check_rate_uniaxial_dual(gpu_api.hessians, yield_kind, kw, idx, plastic)


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_history_second_order_vs_oracle(def_type, yield_kind, kw, rate):
    """cm_adjoint_history, cm_direct_history and cm_hessian_history through the C-ABI: gradient (both ways), the adjoint
    vector and the forward sensitivities of every step and the Hessian d2J/dp2, against the same quantities assembled from
    the oracle's per-step AD blocks (reference cmad/objectives/mp_objective.py:95-345)."""
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    from cmad_amd.models.history_engine import HistoryEngine
    pc.check_history_second_order(lambda desc, info: HistoryEngine(DeviceEvaluator(desc, info)),
                                  lambda values, dt, mk: build_desc(values, def_type=dt, model_kind=mk),
                                  def_type, yield_kind, kw, rate=rate)


@pytest.mark.parametrize("rate", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_extended_parameter_blocks(def_type, yield_kind, kw, rate):
    """cm_param_blocks through the C-ABI: sensitivities w.r.t. the rotation matrix, the Hosford exponent and the native
    parameters by forward-mode evaluation of the whole model, against the oracle's AD (reference model.py:125-153)."""
    import gpu_api
    pc.check_param_blocks(gpu_api.param_blocks, def_type, yield_kind, kw, rate=rate)


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_extended_parameter_blocks_network(def_type, scaled):
    """Hill coefficients (oracle AD) and network weights (finite differences of the oracle) of the hybrid surfaces."""
    import gpu_api
    pc.check_param_blocks_network(gpu_api.param_blocks, def_type, scaled=scaled)


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_second_derivatives_network_surfaces(def_type, scaled):
    """cm_hessians for the hybrid Hill + network yield surfaces (plain and beta-rescaled) against the oracle's nested duals
    (reference model.py:133-147 has no restriction on the yield surface)."""
    import gpu_api
    pc.check_second_derivs_network(gpu_api.hessians, def_type, scaled=scaled)


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_multi_layer_network_sensitivities(def_type, scaled):
    """Two hidden layers (plain and beta-rescaled surface): weight sensitivities (cm_param_blocks; extended-parameter index =
    position in the weight blob) against central differences of the oracle, second derivatives against its nested duals."""
    import gpu_api
    pc.check_param_blocks_network(gpu_api.param_blocks, def_type, scaled=scaled, layer_widths=(6, 7, 5, 1))
    pc.check_second_derivs_network(gpu_api.hessians, def_type, scaled=scaled, layer_widths=(6, 7, 5, 1))


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_barlat_second_derivatives_and_coefficient_sensitivities(def_type):
    """Barlat Yld2004-18p through the arithmetic-T model on the GPU: `cm_hessians` against the oracle's nested duals,
    `cm_param_blocks` w.r.t. its 19 coefficients (central differences of the oracle) and the rotation matrix (oracle AD)."""
    import gpu_api
    pc.check_barlat_generic(gpu_api.hessians, gpu_api.param_blocks, def_type)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_form_with_a_dense_yield_surface(def_type):
    """small_rate_elastic_plastic takes any effective stress (cmad/models/small_rate_elastic_plastic.py:116-126): the rate form
    with Barlat Yld2004-18p on every batched entry point that refused it until round 3 -- cm_update_rate_vjp /
    cm_update_rate_and_vjp, cm_objective_grad_history + cm_update_history, cm_direct_step, cm_hessians_rate, cm_adjoint_history /
    cm_direct_history / cm_hessian_history, cm_param_blocks -- against the oracle (the 12-dof UNIAXIAL_STRESS form: update,
    tangent, reverse sweep, history, forward sensitivities)."""
    import gpu_api
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    from cmad_amd.models.history_engine import HistoryEngine
    yk, kw = pc.BARLAT
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def update(desc, info, g, gp, xp):
        xi, sig, st = DeviceEvaluator(desc, info).update_rate(t(g), t(gp), t(xp))
        return xi.cpu().numpy(), sig.cpu().numpy(), st.cpu().numpy().astype(np.uint32)

    def tangent(desc, info, g, gp, xp, x_expected):
        return DeviceEvaluator(desc, info).update_rate(t(g), t(gp), t(xp), tangent=True)[3].cpu().numpy()

    def vjp(desc, info, g, gp, xp, x, sbar):
        ev = DeviceEvaluator(desc, info)
        gk, xb, ub = ev.update_vjp(t(g), t(xp), t(x), t(sbar), want_xi_prev_bar=True, want_gradu_bar=True, gradu_prev=t(gp))
        xi_f, sig_f, g_f = ev.update_and_vjp(t(g), t(xp), t(sbar), gradu_prev=t(gp))
        np.testing.assert_allclose(xi_f.cpu().numpy()[:6], x[:6], rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(g_f.cpu().numpy(), gk.cpu().numpy(), rtol=1e-8, atol=1e-11 * float(gk.abs().max()))
        return gk.cpu().numpy(), xb.cpu().numpy(), ub.cpu().numpy()

    def direct(desc, info, g, gp, xp, x, dxp):
        dx, ds = DeviceEvaluator(desc, info).direct_step(t(g), t(xp), t(x), dxi_prev_dp=t(dxp), gradu_prev=t(gp))
        return dx.cpu().numpy(), ds.cpu().numpy()
    pc.check_rate_model(update, def_type, yk, kw, True, False, B=600)
    pc.check_rate_tangent(tangent, def_type, yk, kw, True, B=400)
    pc.check_rate_vjp(vjp, def_type, yk, kw, True, B=400)
    pc.check_history(_gpu_history, def_type, yk, kw, True, rate=True, B=300, K=4, uniaxial_idx=2, primal=_gpu_primal)
    pc.check_direct(direct, def_type, yk, kw, True, rate=True, B=200, uniaxial_idx=0)
    if def_type != ol.UNIAXIAL_STRESS:
        pc.check_rate_second_derivs(gpu_api.hessians, gpu_api.evaluate_rate, def_type, yk, kw, True, rot=True)
        pc.check_history_second_order(lambda desc, info: HistoryEngine(DeviceEvaluator(desc, info)),
                                      lambda values, dt, mk: build_desc(values, def_type=dt, model_kind=mk),
                                      def_type, yk, kw, rate=True)
    pc.check_param_blocks(gpu_api.param_blocks, def_type, yk, kw, rate=True)
