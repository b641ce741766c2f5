"""GPU parity of the HIP stress update against the CPU oracle, through the C-ABI (`-m gpu`).
Scenarios and tolerances: tests/parity_cases.py."""
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


def test_hosford_a100_notch_material(backend):
    pc.check_hosford_a100(backend, B=4096)


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_hybrid_hill_icnn(backend, def_type):
    """BASELINE.json configs[3]: neural-network yield surface plugged into the return mapping."""
    pc.check_hybrid_nn(backend, def_type, B=2048, rot=(def_type == ol.FULL_3D))


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
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
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
def test_j2_radial_line_newton_matches_general_path(backend, rot):
    pc.check_j2_radial_line(backend, B=8192, rot=rot)
    # the fused kernel with the flag set gives the same state and gradient as without it
    import torch
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
    from cmad_amd.synthetic import gauss_point_batch, j2_voce_values
    B = 4096
    g = torch.from_numpy(gauss_point_batch(B)).cuda(); xp = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    sb = torch.randn((6, B), dtype=torch.float64, device="cuda")
    a = DeviceEvaluator(*build_desc(j2_voce_values(), newton=NewtonSettings())).update_and_vjp(g, xp, sb)
    b = DeviceEvaluator(*build_desc(j2_voce_values(), newton=NewtonSettings(j2_radial_line=True))).update_and_vjp(g, xp, sb)
    np.testing.assert_allclose(b[0].cpu().numpy(), a[0].cpu().numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(b[1].cpu().numpy(), a[1].cpu().numpy(), rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(b[2].cpu().numpy(), a[2].cpu().numpy(), rtol=1e-9, atol=1e-9 * a[2].abs().max().item())


def test_edge_cases(backend):
    pc.check_edge_cases(backend)
