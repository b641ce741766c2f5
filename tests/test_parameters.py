"""`Parameters` semantics: mirror of /root/reference/tests/parameters/test_parameters.py:22-120 plus the
flat-ordering contract (sorted dict keys = JAX pytree order)."""
import numpy as np

from cmad_amd.parameters.parameters import Parameters, ravel_pytree, transform_from_canonical
from problems import params_J2_voce


def test_flat_order_is_sorted_keys():
    p = params_J2_voce()
    assert p._names[:6] == ["E", "nu", "J2", "D", "S", "Y"]
    assert p._names[6] == "rotation matrix" and p.num_params == 6 + 9
    assert list(p.active_idx) == [3, 4, 5]                       # [D, S, Y]
    assert [path[-1] for path in p.active_paths()] == ["D", "S", "Y"]
    np.testing.assert_allclose(p.flat_active_values(), [20., 200., 200.])
    flat, unravel = ravel_pytree(p.values)
    back = unravel(flat)
    assert back["elastic"]["E"] == 200e3 and back["rotation matrix"].shape == (3, 3)


def test_opt_bounds():
    p = params_J2_voce()
    assert p.opt_bounds.tolist() == [[-1., 1.], [-1., 1.], [None, None]]       # D, S bounds ; Y log


def test_round_trip_canonical_and_raw():
    p = params_J2_voce()
    c = p.flat_active_values(return_canonical=True)
    np.testing.assert_allclose(c, [0., 0., 0.], atol=1e-15)
    p.set_active_values_from_flat(c, are_canonical=True)
    np.testing.assert_allclose(p.flat_active_values(return_canonical=True), c, rtol=1e-12, atol=1e-15)
    raw = p.flat_active_values(False)
    p.set_active_values_from_flat(1.1 * raw, are_canonical=False)
    np.testing.assert_allclose(p.flat_active_values(False), 1.1 * raw, rtol=1e-12)
    assert p.values["elastic"]["E"] == 200e3                      # inactive leaves untouched


def test_canonical_maps():
    p = params_J2_voce()
    p.set_active_values_from_flat(np.array([0.5, -1.0, np.log(1.1)]))
    np.testing.assert_allclose(p.flat_active_values(False), [25., 100., 220.], rtol=1e-13)


def test_transform_grad_and_hessian_match_fd():
    p = params_J2_voce()
    p.set_active_values_from_flat(np.array([0.3, -0.2, 0.1]))
    c = p.flat_active_values(True)
    tr = p._flat_active_transforms
    n = p.num_active_params
    g = np.ones(n)
    p.transform_grad(g)
    h = 1e-5
    fd = np.array([(transform_from_canonical(c[i] + h, True, tr[i]) - transform_from_canonical(c[i] - h, True, tr[i])) / (2 * h)
                   for i in range(n)])
    np.testing.assert_allclose(g, fd, rtol=1e-6, atol=1e-8)
    H = np.eye(n)
    p.transform_hessian(H, np.ones(n))
    assert (H[~np.eye(n, dtype=bool)] == 0).all()
    h = 1e-4
    fd2 = []
    for i in range(n):
        vp, v0, vm = (transform_from_canonical(c[i] + s * h, True, tr[i]) for s in (1, 0, -1))
        fd2.append(((vp - vm) / (2 * h)) ** 2 + (vp - 2 * v0 + vm) / h ** 2)
    np.testing.assert_allclose(np.diag(H), fd2, rtol=1e-5, atol=1e-7)


def test_active_params_jacobian_slicing():
    p = params_J2_voce()
    jac = {"rotation matrix": np.zeros((7, 3, 3)), "elastic": {"E": np.full(7, 1.), "nu": np.full(7, 2.)},
           "plastic": {"effective stress": {"J2": np.zeros(7)},
                       "flow stress": {"initial yield": {"Y": np.full(7, 5.)},
                                       "hardening": {"voce": {"S": np.full(7, 4.), "D": np.full(7, 3.)}}}}}
    A = p.model_active_params_jacobian(jac, 7)
    assert A.shape == (7, 3)
    np.testing.assert_array_equal(A[0], [3., 4., 5.])            # D, S, Y


def test_get_params_pytree_from_flat_canonical_active():
    p = params_J2_voce()
    tree = p.get_params_pytree_from_flat_canonical_active(np.array([1.0, 1.0, 0.0]))
    assert tree["plastic"]["flow stress"]["hardening"]["voce"]["D"] == 30.
    assert tree["plastic"]["flow stress"]["hardening"]["voce"]["S"] == 300.
    assert tree["plastic"]["flow stress"]["initial yield"]["Y"] == 200.
