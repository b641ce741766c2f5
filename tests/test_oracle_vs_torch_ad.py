"""Oracle Jacobians (nested dual numbers, C++) vs an independent torch.func fp64 AD restatement.

This is link (iii) of the transitive parity chain in SURVEY.md section 8(c): JAX cannot run here, so the
oracle's AD is cross-checked against a second AD engine on the same formulas.
"""
import numpy as np
import pytest

import oracle_lib as ol
import torch_ref


def _rand_rot(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    return q


def _state(rng, mat, plastic):
    nd = {ol.FULL_3D: 3, ol.PLANE_STRESS: 2, ol.UNIAXIAL_STRESS: 1}[mat.desc.def_type]
    scale = 4e-3 if plastic else 5e-5
    k = 1.0 if plastic else 0.2
    U = rng.normal(size=(nd, nd)) * scale
    xp = mat.init_xi()
    xp[:6] = rng.normal(size=6) * 2e-4 * k
    xp[3] = -xp[0] - xp[5] + 1e-5
    xp[6] = abs(rng.normal()) * 1e-3
    if mat.nx > 7:
        xp[7:] = 1.0 + rng.normal(size=mat.nx - 7) * 1e-4 * k
    xi = xp.copy()
    xi[:6] += rng.normal(size=6) * 1e-4 * k
    xi[6] += abs(rng.normal()) * 5e-4
    if mat.nx > 7:
        xi[7:] += rng.normal(size=mat.nx - 7) * 1e-4 * k
    return xi, xp, U


@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS, ol.UNIAXIAL_STRESS])
@pytest.mark.parametrize("yield_kind,kw", [("J2", {}), ("hill", {"hill": [0.1477, 0.6805, 0.5345, 1.7977, 1.7148, 2.1675]}),
                                           ("hosford", {"a": 4.}), ("hosford", {"a": 8.5})])
@pytest.mark.parametrize("plastic", [True, False])
def test_residual_and_jacobians(def_type, yield_kind, kw, plastic):
    rng = np.random.default_rng(22)
    for trial in range(3):
        Q = _rand_rot(rng) if trial else np.eye(3)
        mat = ol.Material(ol.j2_voce_values(yield_kind=yield_kind, Q=Q, **kw), def_type=def_type,
                          uniaxial_idx=trial % 3)
        xi, xp, U = _state(rng, mat, plastic)
        _, f, _ = mat.yield_state(xi, U)
        assert (f > 0) == plastic
        ref = torch_ref.jacobians(xi, xp, mat.p, U, mat.desc)
        np.testing.assert_allclose(mat.residual(xi, xp, U), ref["C"], rtol=1e-12, atol=1e-15)
        for which in (ol.W_XI, ol.W_XI_PREV, ol.W_PARAMS, ol.W_U):
            got = mat.jacobian(which, xi, xp, U)
            scale = max(1.0, np.abs(ref[which]).max())
            np.testing.assert_allclose(got, ref[which], rtol=1e-10, atol=1e-12 * scale, err_msg=f"dC which={which}")
        np.testing.assert_allclose(mat.cauchy(xi, U).reshape(9), ref["S"], rtol=1e-12, atol=1e-10)
        for which in (ol.W_XI, ol.W_PARAMS, ol.W_U):
            got = mat.dcauchy(which, xi, xp, U)
            scale = max(1.0, np.abs(ref[("S", which)]).max())
            np.testing.assert_allclose(got, ref[("S", which)], rtol=1e-10, atol=1e-12 * scale, err_msg=f"dS which={which}")
