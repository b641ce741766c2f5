"""GPU parity of the HIP stress update against the CPU oracle, through the C-ABI (`-m gpu`).
Scenarios and tolerances: tests/parity_cases.py."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    return pc.GpuBackend()


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_update(backend, def_type, yield_kind, kw, rot, ls):
    pc.check_update(backend, pc.Scenario(def_type, yield_kind, kw, rot, ls, B=4096))


def test_ragged_and_tiny_batches():
    """B not a multiple of the block / wave size, B = 1, B = 0."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    from cmad_amd.synthetic import gauss_point_batch
    values = ol.j2_voce_values()
    mat = ol.Material(values)
    desc, info = build_desc(values)
    ev = DeviceEvaluator(desc, info)
    for B in (1, 63, 65, 257, 1000):
        gradu = gauss_point_batch(B, seed=B, dev_scale=6.0)
        xi_prev = np.zeros((7, B))
        xi_o, sig_o, it_o, _ = mat.update_batch(ol.newton_settings(), gradu, xi_prev)
        xi_d, sig_d, st = ev.update(torch.from_numpy(gradu).cuda(), torch.from_numpy(xi_prev).cuda())
        np.testing.assert_allclose(xi_d.cpu().numpy(), xi_o, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(sig_d.cpu().numpy(), sig_o, rtol=1e-10, atol=1e-8)
    g0 = torch.empty((9, 0), dtype=torch.float64, device="cuda")
    x0 = torch.empty((7, 0), dtype=torch.float64, device="cuda")
    xi, sig, st = ev.update(g0, x0)
    assert xi.shape == (7, 0)


def test_elastic_points_take_zero_iterations():
    """FULL_3D elastic steps satisfy C_e(x_prev) = 0 exactly -> 0 iterations."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    from cmad_amd.synthetic import gauss_point_batch
    values = ol.j2_voce_values()
    desc, info = build_desc(values)
    ev = DeviceEvaluator(desc, info)
    gradu = gauss_point_batch(2048, dev_scale=0.5)          # all below yield
    xi, sig, st = ev.update(torch.from_numpy(gradu).cuda(), torch.zeros((7, 2048), dtype=torch.float64, device="cuda"))
    st = st.cpu().numpy().astype(np.uint32)
    assert ((st & 0xFFFF) == 0).all() and ((st >> 16) & 1).all() and not ((st >> 17) & 1).any()
    assert torch.count_nonzero(xi).item() == 0


def test_bad_arguments_raise():
    """Host-side operand checks run before any launch: wrong shape / dtype / device are ValueErrors."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    desc, info = build_desc(ol.j2_voce_values())
    ev = DeviceEvaluator(desc, info)
    g = torch.zeros((9, 8), dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        ev.update(g, torch.zeros((7, 9), dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        ev.update(g.float(), torch.zeros((7, 8), dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        ev.update(g.cpu(), torch.zeros((7, 8), dtype=torch.float64))


@pytest.mark.parametrize("reference_iteration", [False, True])
def test_hosford_a100_notch_material(backend, reference_iteration):
    """configs[2]'s material on both solver routes: the analytic warm start (default, lockstep kernels) and the reference's
    iteration from x_prev (CM_SOLVER_GENERAL_NEWTON; B = 4096 >= 256: the work-pool kernel)."""
    pc.check_hosford_a100(backend, B=4096, reference_iteration=reference_iteration)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_hybrid_hill_icnn(backend, def_type):
    """BASELINE.json configs[3]: neural-network yield surface plugged into the return mapping."""
    pc.check_hybrid_nn(backend, def_type, B=2048, rot=(def_type == ol.FULL_3D))


@pytest.mark.parametrize("widths", [(6, 7, 5, 1), (6, 12, 8, 6, 1)])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_hybrid_hill_multi_layer_icnn(backend, def_type, widths):
    """Networks with two and three hidden layers (reference input_convex_neural_network.py:58-69 loops over any depth):
    the general evaluation of the library's second build, update + reverse sweep against the oracle."""
    pc.check_hybrid_nn(backend, def_type, B=1024, rot=(def_type == ol.FULL_3D), widths=widths)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_scaled_hybrid_with_a_multi_layer_icnn(backend, def_type):
    """`scaled_effective_stress` around a network with two hidden layers: the reference composes any ICNN with the beta-rescaling
    (input_convex_neural_network.py:58-69 under effective_stress.py:130-146); EXT build of the library."""
    pc.check_hybrid_nn(backend, def_type, B=768, rot=(def_type == ol.FULL_3D), scaled=True, widths=(6, 7, 5, 1))


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_random_materials(backend, def_type, yield_kind, ls):
    """Randomly drawn materials (elastic constants, yield stress, Voce and / or linear hardening, orientation, Hill coefficients /
    Hosford exponent; uniaxial axis): update over two load steps, consistent tangent and VJP against the oracle."""
    pc.check_random_materials(backend, def_type, yield_kind, ls, seeds=range(6), B=320)


@pytest.mark.parametrize("surface", ["hybrid", "hosford100"])
def test_work_pool_route_equals_lockstep_kernels(surface):
    """The iteration-bound configurations run cm_update on the work pool, and the fused entry points (cm_update_and_vjp,
    cm_objective_grad with a state buffer) go work-pool update -> reverse kernel for them; CM_SOLVER_LOCKSTEP keeps the single
    lockstep kernels.  Same per-point iteration either way: states, stresses, iteration counts, objective and gradient agree."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, HybridHillEffectiveStress, NewtonSettings, build_desc
    from cmad_amd.synthetic import al7079_hybrid_setup, gauss_point_batch, hosford_values
    B = 3001
    if surface == "hybrid":
        icnn, values = al7079_hybrid_setup()
        mk = lambda lock: NewtonSettings(50, 1e-12, 1e-12, {"max evals": 10, "sufficient decrease": 1e-4, "min backtrack factor": 0.5,
                                                           "max backtrack factor": 0.9}, lockstep=lock)
        evs = [DeviceEvaluator(*build_desc(values, newton=mk(lock), hybrid=HybridHillEffectiveStress(icnn))) for lock in (False, True)]
        eps_y = 525.0 / 70.2e3
    else:
        values = hosford_values()
        # (the reference's iteration from x_prev: the default -- Newton from the analytic warm start -- needs no pool)
        mk = lambda lock: NewtonSettings(500, 1e-12, 1e-12, {"max evals": 100, "sufficient decrease": 1e-4, "min backtrack factor": 0.5,
                                                            "max backtrack factor": 0.9}, j2_radial_line=False, lockstep=lock)
        evs = [DeviceEvaluator(*build_desc(values, newton=mk(lock))) for lock in (False, True)]
        eps_y = 2e-3
    g = torch.from_numpy(gauss_point_batch(B, seed=31, eps_y=eps_y)).cuda()
    xp = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    sb = torch.randn((6, B), dtype=torch.float64, device="cuda", generator=gen)
    data = 100.0 * torch.randn((6, B), dtype=torch.float64, device="cuda", generator=gen)
    wsq = [1.0, 0.5, 0.0, 2.0, 0.0, 1.0]
    up = [ev.update(g, xp) for ev in evs]
    assert torch.equal(up[0][2], up[1][2])                                   # status words: iterations, converged, plastic
    assert float(((up[0][2].to(torch.int64) >> 16) & 1).double().mean()) > 0.999
    np.testing.assert_allclose(up[0][0].cpu().numpy(), up[1][0].cpu().numpy(), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(up[0][1].cpu().numpy(), up[1][1].cpu().numpy(), rtol=1e-12, atol=1e-10)
    fused = [ev.update_and_vjp(g, xp, sb) for ev in evs]
    for a, b in zip(fused[0], fused[1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-11, atol=1e-11 * float(b.abs().max()))
    np.testing.assert_allclose(fused[0][0].cpu().numpy(), up[0][0].cpu().numpy(), rtol=0, atol=0)      # the routed path IS cm_update
    tang = [ev.update(g, xp, tangent=True) for ev in evs]                  # cm_update_tangent: work pool + tangent kernel vs lockstep
    assert torch.equal(tang[0][2], tang[1][2])
    np.testing.assert_allclose(tang[0][0].cpu().numpy(), up[0][0].cpu().numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(tang[0][3].cpu().numpy(), tang[1][3].cpu().numpy(), rtol=1e-10,
                               atol=1e-10 * float(tang[1][3].abs().max()))
    obj = [evs[0].objective_grad(g, xp, data, wsq, want_xi=True)[0], evs[0].objective_grad(g, xp, data, wsq)[0],
           evs[1].objective_grad(g, xp, data, wsq, want_xi=True)[0]]
    for o in obj[1:]:
        np.testing.assert_allclose(o.cpu().numpy(), obj[0].cpu().numpy(), rtol=1e-11, atol=1e-11 * float(obj[0].abs().max()))


@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "J2", {}), (ol.FULL_3D, "hill", {"hill": pc.HILL}),
                                                    (ol.PLANE_STRESS, "J2", {}), (ol.FULL_3D, "hosford", {"a": 8.5})])
def test_line_search_rejections(backend, def_type, yield_kind, kw):
    """Every Armijo test fails (c1 = 0.6): the LDS-parked retry path of the line search and, for J2 / FULL_3D, the
    fallback from the radial line to the general search."""
    pc.check_line_search_rejections(backend, def_type, yield_kind, kw, B=4096)


@pytest.mark.parametrize("max_evals", [2, 6])
@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "hosford", {"a": 20.0}), (ol.PLANE_STRESS, "hosford", {"a": 20.0}),
                                                    (ol.UNIAXIAL_STRESS, "hosford", {"a": 20.0})])
def test_legacy_line_search(backend, def_type, yield_kind, kw, max_evals):
    """CM_LS_LEGACY (the backtracking of newton_solve(max_ls_evals > 0), cmad/models/nonlinear_solver.py:55-81) in the lockstep
    kernels (B = 200: structured solver, the 4 x 4 UNIAXIAL step) and in the work-pool kernel (B = 2048, Hosford with a line
    search) against the oracle's LS_LEGACY: states, stresses, iteration counts."""
    pc.check_legacy_line_search(backend, def_type, yield_kind, kw, B=200, max_evals=max_evals)
    pc.check_legacy_line_search(backend, def_type, yield_kind, kw, B=2048, max_evals=max_evals)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_barlat_calibrated_coefficients(backend, def_type):
    """Yld2004-18p, Al7079 coefficients, a = 18.2."""
    pc.check_barlat_calibrated(backend, def_type, B=2048)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_scaled_hybrid_hill_icnn(backend, def_type):
    """beta-rescaled network surface (`scaled_effective_stress`, effective_stress.py:130-146)."""
    pc.check_hybrid_nn(backend, def_type, B=1024, rot=(def_type == ol.FULL_3D), scaled=True)


def test_hybrid_through_the_facade():
    import torch
    from cmad_amd.models import DefType, HybridHillEffectiveStress, SmallElasticPlastic
    from cmad_amd.parameters import Parameters
    from cmad_amd.synthetic import gauss_point_batch
    icnn, values = pc.al7079_hybrid_setup()
    model = SmallElasticPlastic(Parameters(values), DefType.FULL_3D, effective_stress_fun=HybridHillEffectiveStress(icnn))
    B = 256
    g = gauss_point_batch(B, eps_y=525.0 / 70.2e3, dev_scale=5.0)
    xi, sig, st = model.update_batch(torch.from_numpy(g).cuda(), torch.zeros((7, B), dtype=torch.float64, device="cuda"))
    mat = ol.Material(values, nn=icnn.pack_for_device())
    xi_o, sig_o, it_o, cv = mat.update_batch(ol.newton_settings(), g, np.zeros((7, B)))
    ok = cv.astype(bool)
    np.testing.assert_allclose(xi.cpu().numpy()[:, ok], xi_o[:, ok], rtol=1e-9, atol=1e-12)
    with pytest.raises(NotImplementedError):
        SmallElasticPlastic(Parameters(values), DefType.FULL_3D, effective_stress_fun=lambda c, p: 0.0)


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
def test_rate_model_update(def_type, yield_kind, kw, ls):
    """small_rate_elastic_plastic (cm_update_rate) vs the oracle, three load steps."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator

    def run(desc, info, g, gp, xp):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        out = DeviceEvaluator(desc, info).update_rate(t(g), t(gp), t(xp))
        return out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy().astype(np.uint32)
    pc.check_rate_model(run, def_type, yield_kind, kw, True, ls, B=2048)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_j2_radial_line_newton_matches_general_path(backend, def_type, rot):
    pc.check_j2_radial_line(backend, B=8192, rot=rot, def_type=def_type)
    # the fused kernel gives the same state and gradient with the restriction (default) and without it
    import torch
    from cmad_amd.models.deformation_types import DefType
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
    from cmad_amd.synthetic import gauss_point_batch, j2_voce_values
    B = 4099
    ps = def_type == ol.PLANE_STRESS
    dt = DefType.PLANE_STRESS if ps else DefType.FULL_3D
    g = torch.from_numpy(gauss_point_batch(B, ndims=2 if ps else 3)).cuda()
    xp = torch.zeros((8 if ps else 7, B), dtype=torch.float64, device="cuda")
    if ps:
        xp[7] = 1.0
    sb = torch.randn((6, B), dtype=torch.float64, device="cuda")
    a = DeviceEvaluator(*build_desc(j2_voce_values(), def_type=dt, newton=NewtonSettings(j2_radial_line=False))).update_and_vjp(g, xp, sb)
    b = DeviceEvaluator(*build_desc(j2_voce_values(), def_type=dt, newton=NewtonSettings())).update_and_vjp(g, xp, sb)
    np.testing.assert_allclose(b[0].cpu().numpy(), a[0].cpu().numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(b[1].cpu().numpy(), a[1].cpu().numpy(), rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(b[2].cpu().numpy(), a[2].cpu().numpy(), rtol=1e-9, atol=1e-9 * a[2].abs().max().item())
    # ... and over a load history (state carried from step to step): objective + gradient of the whole history
    if ps:
        K = 6
        gh = torch.from_numpy(np.stack([gauss_point_batch(B, ndims=2, seed=40) * (k / K) for k in range(K + 1)])).cuda()
        dh = 50.0 * torch.randn((K + 1, 6, B), dtype=torch.float64, device="cuda")
        outs = []
        for radial in (False, True):
            ev = DeviceEvaluator(*build_desc(j2_voce_values(), def_type=dt, newton=NewtonSettings(j2_radial_line=radial)))
            outs.append(ev.objective_grad_history(gh, dh, [1., 1., 0., 1., 0., 0.], xp)[0].cpu().numpy())
        np.testing.assert_allclose(outs[1], outs[0], rtol=1e-9, atol=1e-9 * np.abs(outs[0]).max())


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_edge_cases(backend, def_type):
    pc.check_edge_cases(backend, def_type)


@pytest.mark.parametrize("line_search", [False, True])
def test_full_size_properties(line_search):
    """BASELINE.json configs[1] at its full size (10^7 points, J2 + Voce, FULL_3D): the oracle cannot finish that in
    seconds, so parity is carried by (i) a random sample of the full launch against the oracle, (ii) independence
    from the launch size (a slice launched alone is bit-identical), (iii) the yield condition and consistency of
    every returned state, (iv) idempotence (re-applying the same strain is an elastic step that returns the state
    unchanged), (v) additivity of the reduced gradient over shards (a checksum of checksums)."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
    from cmad_amd.synthetic import gauss_point_batch
    B = 10_000_000
    values = ol.j2_voce_values()
    newton = NewtonSettings.traced() if line_search else NewtonSettings()
    desc, info = build_desc(values, newton=newton)
    ev = DeviceEvaluator(desc, info)
    g_host = gauss_point_batch(B, seed=22)
    gradu = torch.from_numpy(g_host).cuda()
    xi_prev = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    sbar = torch.randn((6, B), dtype=torch.float64, device="cuda", generator=gen)

    xi, sig, grad = ev.update_and_vjp(gradu, xi_prev, sbar)
    xi_u, sig_u, st = ev.update(gradu, xi_prev)
    assert torch.equal(xi, xi_u) and torch.equal(sig, sig_u)            # fused and plain kernels agree bit for bit
    st = st.to(torch.int64)
    assert bool(((st >> 16) & 1).all())                                  # every point converged
    iters = st & 0xFFFF
    plastic = ((st >> 17) & 1).bool()
    assert 0.6 < plastic.double().mean().item() < 0.85 and int(iters.max()) <= 6
    assert bool((iters[~plastic] == 0).all())

    # (i) random sample against the oracle
    idx = np.sort(np.random.default_rng(5).choice(B, 4096, replace=False))
    tidx = torch.from_numpy(idx).cuda()
    mat = ol.Material(values)
    st_o = (ol.newton_settings(ls_kind=ol.LS_TRACED, ls_max_evals=4) if line_search else ol.newton_settings())
    xi_o, sig_o, it_o, cv_o = mat.update_batch(st_o, g_host[:, idx], np.zeros((7, idx.size)))
    np.testing.assert_allclose(xi[:, tidx].cpu().numpy(), xi_o, rtol=1e-10, atol=pc.XI_ATOL)
    np.testing.assert_allclose(sig[:, tidx].cpu().numpy(), sig_o, rtol=1e-10, atol=1e-8)
    np.testing.assert_array_equal(iters[tidx].cpu().numpy(), it_o)
    # ... and the headline's GRADIENT on the same sample: the fused kernel over the 4096 sampled points of the full launch (their
    # own cotangents) against the oracle's reverse sweep at its converged states -- every active leaf of the parameter tree
    g_o, _, _ = mat.update_vjp_batch(g_host[:, idx], np.zeros((7, idx.size)), xi_o, sbar[:, tidx].cpu().numpy())
    xi_s4, sig_s4, g_s4 = ev.update_and_vjp(gradu[:, tidx].contiguous(), xi_prev[:, tidx].contiguous(), sbar[:, tidx].contiguous())
    assert torch.equal(xi_s4, xi[:, tidx]) and torch.equal(sig_s4, sig[:, tidx])      # the sample's states ARE the full launch's
    got, ref = pc.leaf_grads(g_s4.cpu().numpy(), info, mat, "J2", g_o)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())

    # (ii) a slice launched on its own gives the same bits (no dependence on grid size or neighbours)
    lo, n = 3_333_333, 100_001
    xi_s, sig_s, _ = ev.update(gradu[:, lo:lo + n].contiguous(), xi_prev[:, lo:lo + n].contiguous())
    assert torch.equal(xi_s, xi[:, lo:lo + n]) and torch.equal(sig_s, sig[:, lo:lo + n])

    # (iii) yield condition f = (phi - Y - H(alpha)) / 2mu: ~0 on plastic points, < 0 on elastic ones;
    #       plastic strain is deviatoric and along the stress deviator
    E, nu, Y, S, D = 200e3, 0.3, 200.0, 200.0, 20.0
    two_mu = E / (1.0 + nu)
    w = torch.tensor([1.0, 2.0, 2.0, 1.0, 2.0, 1.0], dtype=torch.float64, device="cuda")[:, None]
    p = (sig[0] + sig[3] + sig[5]) / 3.0
    dev = sig.clone(); dev[0] -= p; dev[3] -= p; dev[5] -= p
    vm = torch.sqrt(1.5 * (w * dev * dev).sum(0))
    alpha = xi[6]
    f = (vm - Y - S * (1.0 - torch.exp(-D * alpha))) / two_mu
    assert float(f[plastic].abs().max()) < 1e-13 and float(f[~plastic].max()) < 1e-14
    assert float((xi[0] + xi[3] + xi[5]).abs().max()) < 1e-15
    ep = xi[:6]
    cosang = (w * ep * dev).sum(0) / (torch.sqrt((w * ep * ep).sum(0)) * torch.sqrt((w * dev * dev).sum(0)))
    assert float((cosang[plastic] - 1.0).abs().max()) < 1e-12
    assert bool((alpha[~plastic] == 0).all()) and float(alpha[plastic].min()) > 0.0

    # (iv) idempotence
    xi2, sig2, st2 = ev.update(gradu, xi)
    assert bool(((st2.to(torch.int64) & 0xFFFF) == 0).all())
    assert torch.equal(xi2, xi)
    assert float((sig2 - sig).abs().max()) < 1e-9

    # (v) the reduced gradient is additive over shards
    acc = torch.zeros_like(grad)
    bounds = [0, 2_500_000, 5_000_001, 7_499_999, B]
    for a, b in zip(bounds[:-1], bounds[1:]):
        _, _, gpart = ev.update_and_vjp(gradu[:, a:b].contiguous(), xi_prev[:, a:b].contiguous(), sbar[:, a:b].contiguous())
        acc += gpart
    np.testing.assert_allclose(acc.cpu().numpy(), grad.cpu().numpy(), rtol=1e-11, atol=1e-11 * float(grad.abs().max()))
    # and repeatable bit for bit (fixed reduction order, no atomics)
    _, _, grad2 = ev.update_and_vjp(gradu, xi_prev, sbar)
    assert torch.equal(grad, grad2)


@pytest.mark.parametrize("line_search", [False, True])
def test_full_size_properties_plane_stress(line_search):
    """10^7 J2 points under PLANE_STRESS (the deformation type of the reference's material-point tests), solved in the
    coordinates of the J2 plane (newton_j2_plane): every point converges, a random sample agrees with the oracle's 8-dof Newton
    in state, stress and iteration count, a slice launched alone is bit-identical, sigma_33 vanishes and the yield condition
    holds on every returned state, re-applying the same strain is a 0-iteration step on EVERY point, and the fused kernel's
    gradient equals the general path's."""
    import torch
    from cmad_amd.models.deformation_types import DefType
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
    from cmad_amd.synthetic import gauss_point_batch
    B = 10_000_000
    values = ol.j2_voce_values()
    newton = NewtonSettings.traced() if line_search else NewtonSettings()
    ev = DeviceEvaluator(*build_desc(values, def_type=DefType.PLANE_STRESS, newton=newton))
    g_host = gauss_point_batch(B, seed=23, ndims=2)
    gradu = torch.from_numpy(g_host).cuda()
    xi_prev = torch.zeros((8, B), dtype=torch.float64, device="cuda"); xi_prev[7] = 1.0
    gen = torch.Generator(device="cuda"); gen.manual_seed(8)
    sbar = torch.randn((6, B), dtype=torch.float64, device="cuda", generator=gen)

    xi, sig, grad = ev.update_and_vjp(gradu, xi_prev, sbar)
    xi_u, sig_u, st = ev.update(gradu, xi_prev)
    assert torch.equal(xi, xi_u) and torch.equal(sig, sig_u)
    st = st.to(torch.int64)
    assert bool(((st >> 16) & 1).all())
    iters = st & 0xFFFF
    # the default route: the plane's scalar return map first -- (nearly) every point passes the reference's test on arrival
    assert int(iters.max()) <= 8 and float((iters == 0).double().mean()) > 0.999

    idx = np.sort(np.random.default_rng(6).choice(B, 4096, replace=False))
    tidx = torch.from_numpy(idx).cuda()
    mat = ol.Material(values, def_type=ol.PLANE_STRESS)
    st_o = (ol.newton_settings(ls_kind=ol.LS_TRACED, ls_max_evals=4) if line_search else ol.newton_settings())
    xp_o = np.zeros((8, idx.size)); xp_o[7] = 1.0
    xi_o, sig_o, it_o, cv_o = mat.update_batch(st_o, g_host[:, idx], xp_o)
    np.testing.assert_allclose(xi[:, tidx].cpu().numpy(), xi_o, rtol=1e-10, atol=pc.XI_ATOL)
    np.testing.assert_allclose(sig[:, tidx].cpu().numpy(), sig_o, rtol=1e-10, atol=1e-8)
    # CM_SOLVER_REFERENCE_ITERATES on the sample: the plane iteration from x_prev -- the oracle's iteration counts, the same states
    ref = DeviceEvaluator(*build_desc(values, def_type=DefType.PLANE_STRESS,
                                      newton=NewtonSettings(line_search=newton.line_search, warm_start=False)))
    xi_r, sig_r, st_r = ref.update(gradu[:, tidx].contiguous(), xi_prev[:, tidx].contiguous())
    it_r = (st_r.to(torch.int64) & 0xFFFF).cpu().numpy()
    assert float(np.mean(it_r == it_o)) > 0.995 and 0.3 < float(np.mean(it_r > 1)) < 0.8
    np.testing.assert_allclose(xi_r.cpu().numpy(), xi[:, tidx].cpu().numpy(), rtol=1e-10, atol=pc.XI_ATOL)

    lo, n = 4_444_444, 100_003
    xi_s, sig_s, _ = ev.update(gradu[:, lo:lo + n].contiguous(), xi_prev[:, lo:lo + n].contiguous())
    assert torch.equal(xi_s, xi[:, lo:lo + n]) and torch.equal(sig_s, sig[:, lo:lo + n])

    E, nu, Y, S, D = 200e3, 0.3, 200.0, 200.0, 20.0
    two_mu = E / (1.0 + nu)
    assert float(sig[5].abs().max()) < 1e-8 * Y                         # plane stress: sigma_33 = 0 (Q = I)
    w = torch.tensor([1.0, 2.0, 2.0, 1.0, 2.0, 1.0], dtype=torch.float64, device="cuda")[:, None]
    p = (sig[0] + sig[3] + sig[5]) / 3.0
    dev = sig.clone(); dev[0] -= p; dev[3] -= p; dev[5] -= p
    vm = torch.sqrt(1.5 * (w * dev * dev).sum(0))
    alpha = xi[6]
    f = (vm - Y - S * (1.0 - torch.exp(-D * alpha))) / two_mu
    plastic = alpha > 0
    assert float(f[plastic].abs().max()) < 1e-13 and float(f[~plastic].max()) < 1e-14
    assert float((xi[0] + xi[3] + xi[5]).abs().max()) < 1e-15

    xi2, sig2, st2 = ev.update(gradu, xi)                                # idempotence, every point
    assert bool(((st2.to(torch.int64) & 0xFFFF) == 0).all())
    assert torch.equal(xi2, xi)

    general = DeviceEvaluator(*build_desc(values, def_type=DefType.PLANE_STRESS,
                                          newton=NewtonSettings(line_search=newton.line_search, j2_radial_line=False)))
    sl = slice(0, 2_000_000)
    xg, sg, gg = general.update_and_vjp(gradu[:, sl].contiguous(), xi_prev[:, sl].contiguous(), sbar[:, sl].contiguous())
    xf, sf, gf = ev.update_and_vjp(gradu[:, sl].contiguous(), xi_prev[:, sl].contiguous(), sbar[:, sl].contiguous())
    np.testing.assert_allclose(xf.cpu().numpy(), xg.cpu().numpy(), rtol=1e-10, atol=pc.XI_ATOL)
    np.testing.assert_allclose(gf.cpu().numpy(), gg.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(gg.abs().max()))


def test_full_size_properties_plane_stress_hill():
    """10^7 Hill points under PLANE_STRESS on the default route (the scalar return map with the stretch eliminated, then the
    reference's 8-dof Newton from there): every point converges, nearly all on arrival; a random sample agrees with the oracle's
    Newton from x_prev in state and stress, and with the reference-iterates route in iteration counts; a slice launched alone is
    bit-identical; sigma_33 vanishes; re-applying the same strain is a 0-iteration step that returns every state bit for bit;
    the gradient of the fused kernel equals the one of the reference-iterates route."""
    import torch
    from cmad_amd.models.deformation_types import DefType
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
    from cmad_amd.synthetic import gauss_point_batch
    B = 10_000_000
    values = ol.j2_voce_values(yield_kind="hill", hill=pc.HILL)
    ev = DeviceEvaluator(*build_desc(values, def_type=DefType.PLANE_STRESS, newton=NewtonSettings()))
    g_host = gauss_point_batch(B, seed=29, ndims=2)
    gradu = torch.from_numpy(g_host).cuda()
    xi_prev = torch.zeros((8, B), dtype=torch.float64, device="cuda"); xi_prev[7] = 1.0
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    sbar = torch.randn((6, B), dtype=torch.float64, device="cuda", generator=gen)
    xi, sig, grad = ev.update_and_vjp(gradu, xi_prev, sbar)
    xi_u, sig_u, st = ev.update(gradu, xi_prev)
    # (two kernels, two inlined copies of the same map: the compiler's fused-multiply-add choices may differ in the last bits)
    assert float((xi - xi_u).abs().max()) < 1e-14 and float((sig - sig_u).abs().max()) < 1e-9
    st = st.to(torch.int64)
    assert bool(((st >> 16) & 1).all())
    iters = st & 0xFFFF
    assert int(iters.max()) <= 8 and float((iters == 0).double().mean()) > 0.99
    xi, sig = xi_u, sig_u

    idx = np.sort(np.random.default_rng(7).choice(B, 4096, replace=False))
    tidx = torch.from_numpy(idx).cuda()
    mat = ol.Material(values, def_type=ol.PLANE_STRESS)
    xp_o = np.zeros((8, idx.size)); xp_o[7] = 1.0
    xi_o, sig_o, it_o, cv_o = mat.update_batch(ol.newton_settings(), g_host[:, idx], xp_o)
    assert cv_o.all()
    np.testing.assert_allclose(xi[:, tidx].cpu().numpy(), xi_o, rtol=1e-10, atol=pc.XI_ATOL)
    np.testing.assert_allclose(sig[:, tidx].cpu().numpy(), sig_o, rtol=1e-10, atol=1e-8)
    ref = DeviceEvaluator(*build_desc(values, def_type=DefType.PLANE_STRESS, newton=NewtonSettings(warm_start=False)))
    xi_r, sig_r, st_r = ref.update(gradu[:, tidx].contiguous(), xi_prev[:, tidx].contiguous())
    it_r = (st_r.to(torch.int64) & 0xFFFF).cpu().numpy()
    assert float(np.mean(it_r == it_o)) > 0.99 and 0.3 < float(np.mean(it_r > 1)) < 0.9
    np.testing.assert_allclose(xi_r.cpu().numpy(), xi[:, tidx].cpu().numpy(), rtol=1e-10, atol=pc.XI_ATOL)

    lo, n = 3_333_333, 100_003
    xi_s, sig_s, _ = ev.update(gradu[:, lo:lo + n].contiguous(), xi_prev[:, lo:lo + n].contiguous())
    assert torch.equal(xi_s, xi[:, lo:lo + n]) and torch.equal(sig_s, sig[:, lo:lo + n])
    assert float(sig[5].abs().max()) < 1e-8 * 200.0                      # plane stress: sigma_33 = 0 (Q = I)
    assert float((xi[0] + xi[3] + xi[5]).abs().max()) < 1e-15            # the flow is trace-free

    xi2, sig2, st2 = ev.update(gradu, xi)                                # idempotence, every point
    assert bool(((st2.to(torch.int64) & 0xFFFF) == 0).all())
    assert torch.equal(xi2, xi)

    sl = slice(0, 2_000_000)
    xg, sg, gg = ref.update_and_vjp(gradu[:, sl].contiguous(), xi_prev[:, sl].contiguous(), sbar[:, sl].contiguous())
    xf, sf, gf = ev.update_and_vjp(gradu[:, sl].contiguous(), xi_prev[:, sl].contiguous(), sbar[:, sl].contiguous())
    np.testing.assert_allclose(xf.cpu().numpy(), xg.cpu().numpy(), rtol=1e-10, atol=pc.XI_ATOL)
    np.testing.assert_allclose(gf.cpu().numpy(), gg.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(gg.abs().max()))


def test_full_size_objective_consistency():
    """configs[4] per-GPU size (10^7 points): the fused objective kernel against the update and vjp kernels run
    separately -- J = 1/2 sum w^2 (sigma - data)^2 from the stored stresses, gradient = vjp with
    sigma_bar = w^2 (sigma - data)."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    from cmad_amd.synthetic import gauss_point_batch
    B = 10_000_000
    desc, info = build_desc(ol.j2_voce_values())
    ev = DeviceEvaluator(desc, info)
    gradu = torch.from_numpy(gauss_point_batch(B, seed=23)).cuda()
    xi_prev = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    data = 100.0 * torch.randn((6, B), dtype=torch.float64, device="cuda", generator=gen)
    wsq = [1.0, 0.0, 0.25, 1.0, 0.0, 4.0]
    res, _ = ev.objective_grad(gradu, xi_prev, data, wsq)
    xi, sig, _ = ev.update(gradu, xi_prev, want_status=False)
    wt = torch.tensor(wsq, dtype=torch.float64, device="cuda")[:, None]
    mis = sig - data
    J = 0.5 * float((wt * mis * mis).sum())
    np.testing.assert_allclose(float(res[0]), J, rtol=1e-12)
    g = ev.update_vjp(gradu, xi_prev, xi, (wt * mis).contiguous())
    g = g[0] if isinstance(g, tuple) else g
    np.testing.assert_allclose(res[1:].cpu().numpy(), g.cpu().numpy(), rtol=1e-11, atol=1e-11 * float(g.abs().max()))


def test_full_size_hosford_a100():
    """configs[2] at 10^7 points: everything converges under the notch deck's solver settings, a random sample agrees
    with the oracle, the returned states satisfy the Hosford yield condition, slices are launch-size independent."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
    from cmad_amd.synthetic import gauss_point_batch, hosford_values
    B = 10_000_000
    values = hosford_values()
    newton = NewtonSettings.traced(max_iters=500, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 100})
    desc, info = build_desc(values, newton=newton)
    ev = DeviceEvaluator(desc, info)
    g_host = gauss_point_batch(B, seed=24, eps_y=2e-3)
    gradu = torch.from_numpy(g_host).cuda()
    xi_prev = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    xi, sig, st = ev.update(gradu, xi_prev)
    st = st.to(torch.int64)
    assert float(((st >> 16) & 1).double().mean()) > 0.9999
    ok = ((st >> 16) & 1).bool()
    plastic = ((st >> 17) & 1).bool() & ok
    assert 0.5 < plastic.double().mean().item() < 0.9
    # Hosford: phi = (1/2 sum |d_i|^a)^(1/a) over the normal-stress differences
    a, E, nu, Y, S, D = 100.0, 1000.0, 0.25, 2.0, 10.0, 2.0
    d = torch.stack([sig[0] - sig[3], sig[3] - sig[5], sig[5] - sig[0]]).abs()
    mx = d.max(0).values
    phi = mx * (0.5 * ((d / mx) ** a).sum(0)) ** (1.0 / a)
    f = (phi - Y - S * (1.0 - torch.exp(-D * xi[6]))) / (E / (1.0 + nu))
    assert float(f[plastic].abs().max()) < 1e-10 and float(f[ok & ~plastic].max()) < 1e-12
    idx = np.sort(np.random.default_rng(6).choice(B, 1024, replace=False))
    tidx = torch.from_numpy(idx).cuda()
    mat = ol.Material(values)
    # The default route starts the Newton at the analytic warm start and returns states that satisfy the reference's residual
    # to round-off, so the sample is compared with the oracle iterated to 1e-14: rtol 1e-10 (north star), atol = 10 x that
    # tolerance in strain units, E x that for the stress.  (The reference's iteration at the deck's 1e-12 stops anywhere inside its
    # tolerance ball -- states 1e-12 apart, 1e-9 relative on |xi| ~ 1e-3: that route's full-size check is tests/test_gpu_pool.py.)
    st_o = ol.newton_settings(max_iters=500, abs_tol=1e-14, rel_tol=1e-14, ls_kind=ol.LS_TRACED, ls_max_evals=100)
    xi_o, sig_o, it_o, cv_o = mat.update_batch(st_o, g_host[:, idx], np.zeros((7, idx.size)))
    both = cv_o.astype(bool) & ok[tidx].cpu().numpy()
    assert both.mean() > 0.999
    np.testing.assert_allclose(xi[:, tidx].cpu().numpy()[:, both], xi_o[:, both], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(sig[:, tidx].cpu().numpy()[:, both], sig_o[:, both], rtol=1e-10, atol=1e-10)
    assert float(((st & 0xFFFF) == 0).double().mean()) > 0.999              # converged at the warm start: one residual evaluation
    lo, n = 7_000_001, 65_537
    xi_s, sig_s, _ = ev.update(gradu[:, lo:lo + n].contiguous(), xi_prev[:, lo:lo + n].contiguous())
    assert torch.equal(xi_s, xi[:, lo:lo + n]) and torch.equal(sig_s, sig[:, lo:lo + n])


def test_full_size_hybrid_network_surface():
    """configs[3] at its full size (5 x 10^6 points, hybrid Hill + ICNN [6,16,1], traced Newton with line search):
    convergence, a random sample against the oracle, launch-size independence, idempotence."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, HybridHillEffectiveStress, NewtonSettings, build_desc
    from cmad_amd.synthetic import al7079_hybrid_setup, gauss_point_batch
    B = 5_000_000
    icnn, values = al7079_hybrid_setup()
    newton = NewtonSettings.traced(max_iters=50, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 10})
    desc, info = build_desc(values, newton=newton, hybrid=HybridHillEffectiveStress(icnn))
    ev = DeviceEvaluator(desc, info)
    g_host = gauss_point_batch(B, seed=25, eps_y=525.0 / 70.2e3)
    gradu = torch.from_numpy(g_host).cuda()
    xi_prev = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    xi, sig, st = ev.update(gradu, xi_prev)
    st = st.to(torch.int64)
    ok = ((st >> 16) & 1).bool()
    assert float(ok.double().mean()) > 0.9999
    plastic = ((st >> 17) & 1).bool() & ok
    assert 0.3 < plastic.double().mean().item() < 0.9
    assert float((xi[0] + xi[3] + xi[5]).abs().max()) < 1e-14          # the surface is pressure independent
    idx = np.sort(np.random.default_rng(8).choice(B, 512, replace=False))
    tidx = torch.from_numpy(idx).cuda()
    mat = ol.Material(values, nn=icnn.pack_for_device())
    st_o = ol.newton_settings(max_iters=50, abs_tol=1e-12, rel_tol=1e-12, ls_kind=ol.LS_TRACED, ls_max_evals=10)
    xi_o, sig_o, it_o, cv_o = mat.update_batch(st_o, g_host[:, idx], np.zeros((7, idx.size)))
    both = cv_o.astype(bool) & ok[tidx].cpu().numpy()
    assert both.mean() > 0.99
    # rtol 1e-10 (north star) with the atol the 1e-12 Newton tolerance sets: both iterations stop anywhere inside the tolerance
    # ball ||C|| < 1e-12 (strain units), so states may differ by ~1e-11 and stresses by 2 mu x that = 5e4 x 1e-11
    np.testing.assert_allclose(xi[:, tidx].cpu().numpy()[:, both], xi_o[:, both], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(sig[:, tidx].cpu().numpy()[:, both], sig_o[:, both], rtol=1e-10, atol=1e-6)
    lo, n = 1_234_567, 50_001
    xi_s, sig_s, _ = ev.update(gradu[:, lo:lo + n].contiguous(), xi_prev[:, lo:lo + n].contiguous())
    assert torch.equal(xi_s, xi[:, lo:lo + n]) and torch.equal(sig_s, sig[:, lo:lo + n])
    sel = slice(0, 200_000)
    xi2, sig2, st2 = ev.update(gradu[:, sel].contiguous(), xi[:, sel].contiguous())
    it2 = st2.to(torch.int64) & 0xFFFF
    assert float((it2[ok[sel]] == 0).double().mean()) > 0.999              # re-applying the strain: elastic, no iterations


def test_full_size_history_consistency():
    """configs[4] with a K-step history at 4 x 10^6 points per GPU: the one-launch history entries against one launch per
    step -- every stored state and stress (cm_update_history vs K x cm_update), the objective and gradient
    (cm_objective_grad_history vs K x cm_update + K x cm_adjoint_step), a random sample against the oracle's adjoint, and
    additivity over shards."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc, fold_weight_and_data
    from cmad_amd.synthetic import gauss_point_batch
    B, K = 4_000_000, 4
    values = ol.j2_voce_values()
    desc, info = build_desc(values)
    ev = DeviceEvaluator(desc, info)
    g0 = gauss_point_batch(B, seed=31)
    path = [0.0, 0.5, 0.9, 1.3, 1.1]
    f64 = dict(dtype=torch.float64, device="cuda")
    gh = torch.stack([c * torch.from_numpy(g0).cuda() for c in path]).contiguous()
    xi0 = torch.zeros((7, B), **f64)
    xh, sh, st = ev.update_history(gh, xi0)
    assert bool((((st.to(torch.int64) >> 16) & 1) == 1).all())
    gen = torch.Generator(device="cuda"); gen.manual_seed(12)
    dh = (sh + 20.0 * torch.randn(sh.shape, generator=gen, **f64)).contiguous()
    w = np.zeros((3, 3)); w[0, 0] = 1.; w[1, 1] = 1.; w[2, 2] = 0.5; w[0, 1] = w[1, 0] = 0.5
    wsq6 = fold_weight_and_data(w)
    out, xh2 = ev.objective_grad_history(gh, dh, wsq6, xi0)
    assert torch.equal(xh2, xh)
    x = xi0
    acc = torch.zeros(13, **f64)
    hist = torch.zeros((7, B), **f64)
    xs = [xi0]
    for k in range(1, K + 1):
        x, s, _ = ev.update(gh[k], x)
        np.testing.assert_allclose(xh[k].cpu().numpy(), x.cpu().numpy(), rtol=1e-10, atol=1e-14)
        assert float((sh[k] - s).abs().max()) < 1e-8
        xs.append(x)
    for k in range(K, 0, -1):
        ev.adjoint_step(gh[k], xs[k - 1], xs[k], dh[k], wsq6, hist, hist, acc, accumulate=True)
    np.testing.assert_allclose(float(out[0]), float(acc[0]), rtol=1e-12)
    np.testing.assert_allclose(out[1:].cpu().numpy(), acc[1:].cpu().numpy(), rtol=1e-9, atol=1e-11 * float(acc[1:].abs().max()))
    # additivity over uneven shards, and bit-for-bit repeatability
    tot = torch.zeros(13, **f64)
    bounds = [0, 1_000_001, 2_999_999, B]
    for a, b in zip(bounds[:-1], bounds[1:]):
        o, _ = ev.objective_grad_history(gh[:, :, a:b].contiguous(), dh[:, :, a:b].contiguous(), wsq6, xi0[:, a:b].contiguous())
        tot += o
    np.testing.assert_allclose(tot.cpu().numpy(), out.cpu().numpy(), rtol=1e-11, atol=1e-11 * float(out.abs().max()))
    out2, _ = ev.objective_grad_history(gh, dh, wsq6, xi0)
    assert torch.equal(out, out2)
    # a random sample against the oracle's adjoint
    idx = np.sort(np.random.default_rng(8).choice(B, 2048, replace=False))
    tidx = torch.from_numpy(idx).cuda()
    mat = ol.Material(values)
    gh_s = gh[:, :, tidx].cpu().numpy()
    dh_s = dh[:, :, tidx].cpu().numpy()
    idx9 = [0, 1, 2, 1, 3, 4, 2, 4, 5]
    J_o, g_o, _, xk = mat.objective_grad_batch(ol.newton_settings(), gh_s, dh_s[:, idx9, :], w, np.zeros((7, idx.size)))
    o_s, xh_s = ev.objective_grad_history(gh[:, :, tidx].contiguous(), dh[:, :, tidx].contiguous(), wsq6, xi0[:, tidx].contiguous())
    np.testing.assert_allclose(xh_s[K].cpu().numpy(), xk, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(float(o_s[0]), J_o, rtol=1e-10)
    got, ref = pc.leaf_grads(o_s[1:].cpu().numpy(), info, mat, "J2", g_o)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-11 * np.abs(ref).max())


def test_entry_points_are_graph_capture_safe():
    """include/cmad_hip.h promises that nothing is allocated or synchronised inside the entry points: a per-step
    calibration sequence (cm_update, cm_adjoint_step, cm_update_and_vjp with the two reduction launches each) is captured
    into a HIP graph on a side stream and replayed on new input values; the results equal the eager calls."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc, fold_weight_and_data
    from cmad_amd.synthetic import gauss_point_batch
    B = 5000
    desc, info = build_desc(ol.j2_voce_values())
    ev = DeviceEvaluator(desc, info)
    f64 = dict(dtype=torch.float64, device="cuda")
    g_a = torch.from_numpy(gauss_point_batch(B, seed=41)).cuda()
    g_b = torch.from_numpy(gauss_point_batch(B, seed=42, dev_scale=5.0)).cuda()
    gradu = g_a.clone()
    xi0 = torch.zeros((7, B), **f64)
    data = torch.zeros((6, B), **f64)
    sbar = torch.ones((6, B), **f64)
    wsq6 = fold_weight_and_data(np.eye(3))
    out13 = torch.zeros(13, **f64)
    hist = torch.zeros((7, B), **f64)
    res = {"xi": torch.empty((7, B), **f64), "sigma": torch.empty((6, B), **f64), "grad": torch.empty(12, **f64)}
    bufs = {"xi": torch.empty((7, B), **f64), "sigma": torch.empty((6, B), **f64)}

    def sequence():
        ev.update(gradu, xi0, want_status=False, out=bufs)
        ev.adjoint_step(gradu, xi0, bufs["xi"], data, wsq6, None, hist, out13, accumulate=False)
        ev.update_and_vjp(gradu, xi0, sbar, out=res)

    ev._workspace(B, gradu.device)                      # allocated before the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sequence()                                      # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        sequence()
    for g_new in (g_b, g_a):
        gradu.copy_(g_new)
        graph.replay()
        torch.cuda.synchronize()
        got = (bufs["xi"].clone(), out13.clone(), hist.clone(), res["grad"].clone(), res["sigma"].clone())
        sequence()
        torch.cuda.synchronize()
        for a, b in zip(got, (bufs["xi"], out13, hist, res["grad"], res["sigma"])):
            assert torch.equal(a, b)
    assert float(out13[0]) > 0.0


@pytest.mark.parametrize("model_kind", [0, 1])
def test_hill_material_rotations_golden(golden_dir, model_kind):
    """The reference's own known-answer test for anisotropy (tests/models/test_hill_material_rotations.py:40-158: Al7079
    Hill coefficients, 12 material orientations, 200 load steps, tolerance 1e-8 on the yy stress history) run directly on
    the GPU path: each history is one cm_update_history launch, once with the rotation applied to the deformation and the
    stress outside the model (Q = I) and once with params["rotation matrix"] = R."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    g = np.load(os.path.join(golden_dir, "hill_rotations.npz"))
    hill, el, Y, voce = g["hill"], g["elastic"], float(g["Y"]), g["voce"]
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]

    def history(values, F):
        desc, info = build_desc(values, model_kind=model_kind)
        ev = DeviceEvaluator(desc, info)
        K = F.shape[2] - 1
        gh = torch.from_numpy(np.stack([(F[:, :, k] - np.eye(3)).reshape(9, 1) for k in range(K + 1)])).cuda().contiguous()
        xi0 = torch.zeros((7, 1), dtype=torch.float64, device="cuda")
        _, sh, st = ev.update_history(gh, xi0)
        assert bool((((st.to(torch.int64) >> 16) & 1) == 1).all())
        s6 = sh.cpu().numpy()[:, :, 0]
        S = np.zeros((3, 3, K + 1))
        for r, (i, j) in enumerate(V6):
            S[i, j] = S[j, i] = s6[:, r]
        return S

    for R, stress, strain in zip(g["R"], g["stress"], g["strain"]):
        num_steps = 200
        F = np.repeat(np.eye(3)[:, :, None], num_steps + 1, axis=2)
        F[:, :, 1:] += strain
        ref_yy = stress[1, 1, :]
        vals = ol.j2_voce_values(E=el[0], nu=el[1], Y=Y, S=voce[0], D=voce[1], yield_kind="hill", hill=hill)
        Fr = np.stack([R.T @ F[:, :, k] @ R for k in range(num_steps + 1)], axis=2)
        cauchy = history(vals, Fr)
        num_yy = np.array([(R @ cauchy[:, :, k] @ R.T)[1, 1] for k in range(1, num_steps + 1)])
        assert np.linalg.norm(ref_yy - num_yy) < 1e-8
        vals = ol.j2_voce_values(E=el[0], nu=el[1], Y=Y, S=voce[0], D=voce[1], yield_kind="hill", hill=hill, Q=R)
        cauchy = history(vals, F)
        assert np.linalg.norm(ref_yy - cauchy[1, 1, 1:]) < 1e-8


@pytest.mark.parametrize("model_kind", [0, 1])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
def test_j2_voce_analytical_golden(golden_dir, model_kind, def_type, yield_kind):
    """The reference's analytical J2 + Voce known-answer test (tests/models/test_elastic_plastic_models.py:15-125 with
    cmad/verification/solutions.py:30-58: uniaxial and biaxial stress histories, 100 steps, tolerance 1e-6 in norm on the
    stress and the hardening variable) run directly on the GPU path, one cm_update_history launch per history; the
    J2-equivalent Hill and Hosford (a = 2 equivalent settings of tests/support/test_problems.py) reproduce the same fields."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    nd = {ol.FULL_3D: 3, ol.PLANE_STRESS: 2, ol.UNIAXIAL_STRESS: 1}[def_type]
    V6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    desc, info = build_desc(ol.j2_voce_values(yield_kind=yield_kind), def_type=def_type, model_kind=model_kind)
    ev = DeviceEvaluator(desc, info)
    mat = ol.Material(ol.j2_voce_values(yield_kind=yield_kind), def_type=def_type, model_kind=model_kind)
    for name in (["uniaxial", "biaxial"] if def_type != ol.UNIAXIAL_STRESS else ["uniaxial"]):
        stress, strain, alpha = g[f"{name}_stress"], g[f"{name}_strain"], g[f"{name}_alpha"]
        K = 100
        gh = np.zeros((K + 1, nd * nd, 1))
        gh[1:, :, 0] = strain[:nd, :nd, :].reshape(nd * nd, K).T
        xi0 = torch.from_numpy(mat.init_xi()[:, None].copy()).cuda()
        xh, sh, st = ev.update_history(torch.from_numpy(gh).cuda().contiguous(), xi0)
        st = st.to(torch.int64)
        assert bool((((st >> 16) & 1) == 1).all()) and int((st & 0xFFFF).max()) <= 10
        s6 = sh.cpu().numpy()[:, :, 0]
        cauchy = np.zeros((3, 3, K + 1))
        for r, (i, j) in enumerate(V6):
            cauchy[i, j] = cauchy[j, i] = s6[:, r]
        assert np.linalg.norm(xh.cpu().numpy()[1:, 6, 0] - alpha) < 1e-6
        assert np.linalg.norm(cauchy[:, :, 1:] - stress) < 1e-6


@pytest.mark.parametrize("hidden", [None, [4, 3], [3, 2, 4]])       # widths [1, 5, 1] / [1, 4, 3, 1] / [1, 3, 2, 4, 1]
@pytest.mark.parametrize("with_voce", [False, True])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_network_hardening_law(backend, def_type, with_voce, hidden):
    """hardening_funs = {"neural network": SimpleNeuralNetwork([1, H, 1]).evaluate} (cmad/models/small_elastic_plastic.py:115,
    cmad/neural_networks/simple_neural_network.py:13-46): update, reverse sweep and weight sensitivities through the C-ABI
    against the oracle; also on the J2 subspace kernels (the default for J2) and the general path."""
    import gpu_api
    pc.check_nn_hardening(backend, gpu_api.param_blocks, def_type, with_voce=with_voce, B=1500, hidden=hidden)


@pytest.mark.parametrize("surface", ["hybrid", "scaled hybrid", "barlat", "hosford reference iteration"])
def test_screened_route_against_the_oracle(backend, surface):
    """FULL_3D batches of >= 4096 points of the expensive surfaces run screened when the caller provides the workspace
    (`cm_update_ws`; DeviceEvaluator.update always does): k_screen finishes the elastic points and lists the plastic ones,
    k_update_listed iterates over the list.  Same oracle comparison as the lockstep / work-pool routes, at B = 4608 (rotated
    frame where the check supports it), then the same call with screening switched off must agree bit for bit on the states."""
    B = 4608
    if surface == "hybrid":
        pc.check_hybrid_nn(backend, ol.FULL_3D, B=B, rot=True)
    elif surface == "scaled hybrid":
        pc.check_hybrid_nn(backend, ol.FULL_3D, B=B, rot=True, scaled=True)
    elif surface == "barlat":
        pc.check_barlat_calibrated(backend, ol.FULL_3D, B=B)
    else:
        pc.check_hosford_a100(backend, B=B, reference_iteration=True)


def test_screened_route_is_taken_and_equals_the_lockstep_kernels():
    """The screened route's Newton is the lockstep kernel's (newton_any over a list instead of a block of consecutive points):
    with CM_SOLVER_LOCKSTEP the same batch runs in k_update, and states, stresses and status words agree bit for bit; the
    evaluator reports the route it takes."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, HybridHillEffectiveStress, NewtonSettings, build_desc
    from cmad_amd.synthetic import al7079_hybrid_setup, gauss_point_batch
    B = 20_001
    icnn, values = al7079_hybrid_setup()
    mk = lambda lock: NewtonSettings(50, 1e-12, 1e-12, {"max evals": 10, "sufficient decrease": 1e-4, "min backtrack factor": 0.5,
                                                       "max backtrack factor": 0.9}, lockstep=lock)
    evs = [DeviceEvaluator(*build_desc(values, newton=mk(lock), hybrid=HybridHillEffectiveStress(icnn))) for lock in (False, True)]
    assert evs[0].screened(B) and not evs[1].screened(B) and not evs[0].screened(4095)
    g = torch.from_numpy(gauss_point_batch(B, seed=33, eps_y=525.0 / 70.2e3)).cuda()
    xp = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    out = [{"xi": torch.full((7, B), float("nan"), dtype=torch.float64, device="cuda"),
            "sigma": torch.full((6, B), float("nan"), dtype=torch.float64, device="cuda"),
            "status": torch.full((B,), -1, dtype=torch.int32, device="cuda")} for _ in evs]
    for ev, o in zip(evs, out):
        ev.update(g, xp, out=o)
    torch.cuda.synchronize()
    for k in ("xi", "sigma", "status"):
        assert not bool((out[0][k] != out[0][k]).any()) if k != "status" else not bool((out[0][k] == -1).any())
        assert torch.equal(out[0][k], out[1][k]), k
    it = out[0]["status"].to(torch.int64) & 0xFFFF
    assert 0.3 < float((it > 0).double().mean()) < 0.8


def test_screened_route_classifies_points_at_the_yield_surface_like_the_lockstep_kernels():
    """k_screen evaluates the network term in float and settles the points within its error band in double: trial states ON the
    yield surface and within 1e-14 ... 1e-3 (relative) of it, on both sides, must be split into elastic and plastic exactly as
    the double-precision lockstep kernel splits them -- states, stresses and status words bit for bit."""
    import torch
    from cmad_amd.models.device import DeviceEvaluator, HybridHillEffectiveStress, NewtonSettings, build_desc
    from cmad_amd.synthetic import al7079_hybrid_setup, gauss_point_batch
    n = 4096
    icnn, values = al7079_hybrid_setup()
    mk = lambda lock: NewtonSettings(50, 1e-12, 1e-12, {"max evals": 10, "sufficient decrease": 1e-4, "min backtrack factor": 0.5,
                                                       "max backtrack factor": 0.9}, lockstep=lock)
    ev_s, ev_l = [DeviceEvaluator(*build_desc(values, newton=mk(lock), hybrid=HybridHillEffectiveStress(icnn))) for lock in (False, True)]
    d = torch.from_numpy(gauss_point_batch(n, seed=41, eps_y=525.0 / 70.2e3)).cuda()          # directions in strain space
    xp = torch.zeros((7, n), dtype=torch.float64, device="cuda")
    # the scale at which each direction reaches the surface: bisection on the lockstep kernel's own elastic / plastic decision
    lo = torch.zeros(n, dtype=torch.float64, device="cuda")
    hi = torch.full((n,), 8.0, dtype=torch.float64, device="cuda")
    plastic = lambda t: (ev_l.update(d * t[None, :], xp)[2].to(torch.int64) & 0xFFFF) > 0
    keep = plastic(hi) & ~plastic(lo + 1e-3)            # (nearly volumetric directions never yield: left out)
    assert float(keep.double().mean()) > 0.9
    n = int(keep.sum())
    d, xp, lo, hi = d[:, keep].contiguous(), xp[:, keep].contiguous(), lo[keep], hi[keep]
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        pl = plastic(mid)
        hi = torch.where(pl, mid, hi); lo = torch.where(pl, lo, mid)
    rel = torch.tensor([0.0, 1e-14, -1e-14, 1e-11, -1e-11, 1e-8, -1e-8, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3, 0.5],
                       dtype=torch.float64, device="cuda")
    t = (hi[None, :] * (1.0 + rel[:, None])).reshape(-1)                                  # 16 x 4096 points around the surface
    g = (d.repeat(1, rel.numel()) * t[None, :]).contiguous()
    B = g.shape[1]
    x0 = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    assert ev_s.screened(B) and not ev_l.screened(B)
    a, b = ev_s.update(g, x0), ev_l.update(g, x0)
    torch.cuda.synchronize()
    for u, v, what in zip(a, b, ("xi", "sigma", "status")):
        assert torch.equal(u, v), what
    its = (a[2].to(torch.int64) & 0xFFFF).reshape(rel.numel(), n)
    assert bool((its[rel > 1e-9] > 0).all()) and bool((its[rel < -1e-9] == 0).all())      # (sanity: the bracket is where the surface is)


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "hill", {"hill": pc.HILL}), (ol.PLANE_STRESS, "J2", {}), (ol.PLANE_STRESS, "hill", {"hill": pc.HILL}),
                                                    (ol.FULL_3D, "hosford", {"a": 20.0}), (ol.FULL_3D, "hosford", {"a": 64.0}),
                                                    (ol.UNIAXIAL_STRESS, "J2", {}), (ol.UNIAXIAL_STRESS, "hill", {"hill": pc.HILL})])
def test_warm_started_newton_against_the_oracle(backend, def_type, yield_kind, kw, rot, ls):
    """Scalar return maps / analytic warm starts (the default of the batched entry points) against the oracle's general Newton."""
    if yield_kind == "hosford" and not ls:
        pytest.skip("plain Newton from x_prev does not converge for large Hosford exponents: nothing to compare with")
    pc.check_warm_start(backend, def_type, yield_kind, kw, rot, ls, B=4096, uniaxial_idx=2 if rot else 1)


@pytest.mark.parametrize("def_type,yield_kind,kw", [(ol.FULL_3D, "hill", {"hill": pc.HILL}), (ol.PLANE_STRESS, "J2", {}), (ol.PLANE_STRESS, "hill", {"hill": pc.HILL}),
                                                    (ol.FULL_3D, "hosford", {"a": 100.0}), (ol.FULL_3D, "hosford", {"a": 20.0}),
                                                    (ol.UNIAXIAL_STRESS, "J2", {}), (ol.UNIAXIAL_STRESS, "hill", {"hill": pc.HILL})])
def test_warm_started_newton_edge_cases(backend, def_type, yield_kind, kw):
    """Zero / volumetric strains, 20- and 200-yield-strain increments and a step from a heavily hardened state on the default route."""
    pc.check_warm_start_edge_cases(backend, def_type, yield_kind, kw)
