"""Pin the CPU oracle against the reference's own known-answer vectors.

Mirrors /root/reference/tests/models/test_elastic_plastic_models.py:15-125 (J2 + Voce analytical
fields; J2, J2-equivalent Hill and J2-equivalent Hosford(a=4) models; `small` and `small rate`;
FULL_3D / PLANE_STRESS / UNIAXIAL_STRESS) and tests/models/test_hill_material_rotations.py:40-158,
with the same tolerances.  Expected values come from tests/golden/*.npz, which were produced by
executing the reference's cmad/verification/solutions.py (tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest

import oracle_lib as ol


def _masks(def_type, g):
    if def_type in (ol.FULL_3D, ol.PLANE_STRESS):
        return ["uniaxial", "biaxial"]
    return ["uniaxial"]


def _run_history(mat, F, settings):
    """The forward loop of run_model_and_compare (reference :79-125) on the oracle."""
    nd = F.shape[0]
    num_steps = F.shape[2] - 1
    xi_prev = mat.init_xi()
    cauchy = np.zeros((3, 3, num_steps + 1))
    alphas = np.zeros(num_steps)
    iters = np.zeros(num_steps, dtype=int)
    for step in range(1, num_steps + 1):
        U = F[:, :, step] - np.eye(nd)
        Up = F[:, :, step - 1] - np.eye(nd)
        xi, it, cn, cv = mat.newton(settings, xi_prev, U, Up)
        cauchy[:, :, step] = mat.cauchy(xi, U)
        alphas[step - 1] = xi[6]
        iters[step - 1] = it
        xi_prev = xi
    return cauchy, alphas, iters


@pytest.mark.parametrize("model_kind", [ol.SMALL_EP, ol.SMALL_RATE_EP])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
@pytest.mark.parametrize("yield_kind", ["J2", "hill", "hosford"])
def test_j2_voce_analytical(golden_dir, model_kind, def_type, yield_kind):
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    nd = {ol.FULL_3D: 3, ol.PLANE_STRESS: 2, ol.UNIAXIAL_STRESS: 1}[def_type]
    settings = ol.newton_settings()           # newton_solve defaults: 10 iters, 1e-14, no line search
    for name in _masks(def_type, g):
        stress, strain, alpha = g[f"{name}_stress"], g[f"{name}_strain"], g[f"{name}_alpha"]
        mask = g[f"{name}_mask"]
        num_steps = 100
        F = np.repeat(np.eye(nd)[:, :, None], num_steps + 1, axis=2)     # get_F, reference :37-42
        F[:, :, 1:] += strain[:nd, :nd, :]
        mat = ol.Material(ol.j2_voce_values(yield_kind=yield_kind), def_type=def_type, model_kind=model_kind)
        cauchy, model_alpha, iters = _run_history(mat, F, settings)
        tol = 1e-6
        assert np.linalg.norm(model_alpha - alpha) < tol
        assert np.linalg.norm(cauchy[:, :, 1:] - stress) < tol
        w = np.abs(mask)
        J = sum(0.5 * np.sum((w * cauchy[:, :, k]) ** 2) for k in range(1, num_steps + 1))
        assert abs(J - 0.5 * np.linalg.norm(w[:, :, None] * cauchy) ** 2) < tol
        assert iters.max() <= 10


def test_five_step_known_values(golden_dir):
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    np.testing.assert_allclose(g["uniaxial_5step_sigma_xx"][:2], [200.0, 383.58300028], rtol=1e-9)


@pytest.mark.parametrize("model_kind", [ol.SMALL_EP, ol.SMALL_RATE_EP])
def test_hill_material_rotations(golden_dir, model_kind):
    g = np.load(os.path.join(golden_dir, "hill_rotations.npz"))
    hill, el, Y, voce = g["hill"], g["elastic"], float(g["Y"]), g["voce"]
    settings = ol.newton_settings()
    for R, stress, strain in zip(g["R"], g["stress"], g["strain"]):
        num_steps = 200
        F = np.repeat(np.eye(3)[:, :, None], num_steps + 1, axis=2)
        F[:, :, 1:] += strain
        ref_yy = stress[1, 1, :]
        # method 1: rotate outside the model (Q = I)
        vals = ol.j2_voce_values(E=el[0], nu=el[1], Y=Y, S=voce[0], D=voce[1], yield_kind="hill", hill=hill)
        mat = ol.Material(vals, model_kind=model_kind)
        Fr = np.stack([R.T @ F[:, :, k] @ R for k in range(num_steps + 1)], axis=2)
        cauchy, _, _ = _run_history(mat, Fr, settings)
        num_yy = np.array([(R @ cauchy[:, :, k] @ R.T)[1, 1] for k in range(1, num_steps + 1)])
        assert np.linalg.norm(ref_yy - num_yy) < 1e-8
        # method 2: rotation matrix inside the model
        vals = ol.j2_voce_values(E=el[0], nu=el[1], Y=Y, S=voce[0], D=voce[1], yield_kind="hill", hill=hill, Q=R)
        mat = ol.Material(vals, model_kind=model_kind)
        cauchy, _, _ = _run_history(mat, F, settings)
        assert np.linalg.norm(ref_yy - cauchy[1, 1, 1:]) < 1e-8


def test_isotropic_barlat_is_j2(golden_dir):
    """Yld2004-18p with all 18 coefficients = 1 and a = 4 (or 2) is von Mises for ANY stress state (both
    transformed stresses are the deviator, (1/4 sum_ij |s_i - s_j|^a)^(1/a) = (1/2 sum_{i<j} |s_i - s_j|^a)^(1/a)):
    value, normal and full residual Jacobian equal J2's, and the reference's analytical J2 + Voce biaxial history
    (tests/models/test_elastic_plastic_models.py:15-125, distinct principal stresses) is reproduced at its 1e-6
    tolerance.  (The uniaxial history has a repeated eigenvalue, where the eigh derivative rule the reference's
    AD uses -- and therefore this oracle -- divides by zero.)"""
    rng = np.random.default_rng(9)
    mj = ol.Material(ol.j2_voce_values())
    for a in (2.0, 4.0):
        mb = ol.Material(ol.j2_voce_values(yield_kind="barlat", a=a))
        for trial in range(5):
            G = rng.normal(size=(3, 3)) * 3e-3
            x = np.r_[rng.normal(size=6) * 1e-4, abs(rng.normal()) * 1e-3]
            x[5] = -(x[0] + x[3])
            pb, fb, nb = mb.yield_state(x, G.ravel())
            pj, fj, nj = mj.yield_state(x, G.ravel())
            np.testing.assert_allclose(pb, pj, rtol=1e-13)
            np.testing.assert_allclose(nb, nj, rtol=0, atol=1e-13)
            np.testing.assert_allclose(mb.jacobian(ol.W_XI, x, np.zeros(7), G.ravel()),
                                       mj.jacobian(ol.W_XI, x, np.zeros(7), G.ravel()), rtol=0, atol=1e-11)
    g = np.load(os.path.join(golden_dir, "j2_voce_analytical.npz"))
    stress, strain, alpha = g["biaxial_stress"], g["biaxial_strain"], g["biaxial_alpha"]
    F = np.repeat(np.eye(3)[:, :, None], 101, axis=2)
    F[:, :, 1:] += strain
    mb = ol.Material(ol.j2_voce_values(yield_kind="barlat", a=4.0))
    cauchy, model_alpha, iters = _run_history(mb, F, ol.newton_settings())
    assert np.linalg.norm(model_alpha - alpha) < 1e-6
    assert np.linalg.norm(cauchy[:, :, 1:] - stress) < 1e-6


# ---- the reference's own line-search unit tests (tests/util/test_line_search.py), on the oracle's search ----------

def test_line_search_quad_min_recovers_quadratic_minimizer():
    """reference :39-55: q(t) = (t - 0.4)^2 from its value and slope at 0 and its value at a = 1."""
    q = lambda t: (t - 0.4) ** 2
    assert ol.quad_min(q(0.0), 2.0 * (0.0 - 0.4), 1.0, q(1.0)) == pytest.approx(0.4, abs=1e-12)


def test_line_search_full_step_accepted():
    """reference :84-93: r(a) = 1 - a; the full step is accepted and its residual comes back."""
    alpha, aux = ol.line_search_linear(1.0, -1.0, 0.5, -1.0, init_aux=1.0)
    assert alpha == pytest.approx(1.0) and aux == pytest.approx(0.0)


def test_line_search_backtracks_on_overshoot():
    """reference :57-81, 96-118: r(a) = 1 - 3a, the full step triples the residual; the quadratic model damps it to
    0 < alpha < 1 with sufficient decrease and returns the residual at the accepted step."""
    phi = lambda a: 0.5 * (1.0 - 3.0 * a) ** 2
    alpha, aux = ol.line_search_linear(1.0, -3.0, phi(0.0), -3.0, init_aux=1.0)
    assert 0.0 < alpha < 1.0
    assert phi(alpha) < phi(0.0) and phi(alpha) <= phi(0.0) + 1e-4 * alpha * (-3.0)
    assert aux == pytest.approx(1.0 - 3.0 * alpha)


def test_line_search_disabled_returns_full_step():
    """reference :137-147: max evals = 0 takes the full step and returns init_aux unprobed."""
    alpha, aux = ol.line_search_linear(1.0, -3.0, 0.5, -3.0, init_aux=7.0, max_evals=0)
    assert alpha == 1.0 and aux == 7.0


def test_line_search_nonfinite_probe_contracts():
    """reference :173-190: the merit is NaN for alpha > 0.75; the step is halved into the finite region."""
    alpha, aux = ol.line_search_linear(1.0, -1.0, 0.5, -1.0, init_aux=1.0, nan_above=0.75)
    assert 0.0 < alpha <= 0.75
    assert aux == pytest.approx(1.0 - alpha)
