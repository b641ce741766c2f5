"""CPU-side checks of the host layer: elastic constants, model description, weight folding, the C-ABI
library (loads, exports every symbol of include/cmad_hip.h, struct layout), sharding + all-reduce (gloo)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_elastic_constants_all_pairs():
    """Mirror of /root/reference/tests/models/test_elastic_constants.py:13-37: every pair round-trips."""
    from cmad_amd.models.elastic_constants import ElasticConstants, lame_jacobian
    E, nu = 200e3, 0.3
    ref = ElasticConstants.from_params({"E": E, "nu": nu})
    full = {"E": E, "nu": nu, "mu": ref.mu, "kappa": ref.kappa, "lambda": ref.lmbda}
    names = list(full)
    for i in range(5):
        for j in range(i + 1, 5):
            pair = {names[i]: full[names[i]], names[j]: full[names[j]]}
            ec = ElasticConstants.from_params(pair)
            np.testing.assert_allclose([ec.lmbda, ec.mu, ec.E, ec.nu, ec.kappa], [ref.lmbda, ref.mu, E, nu, ref.kappa], rtol=1e-12)
            got_names, lm, mu, J = lame_jacobian(pair)
            for c, n in enumerate(got_names):                 # d(lambda, mu)/d(pair) vs central FD
                h = 1e-6 * abs(pair[n])
                up, dn = dict(pair), dict(pair)
                up[n] += h; dn[n] -= h
                a, b = ElasticConstants.from_params(up), ElasticConstants.from_params(dn)
                np.testing.assert_allclose(J[:, c], [(a.lmbda - b.lmbda) / (2 * h), (a.mu - b.mu) / (2 * h)], rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError):
        ElasticConstants.from_params({"E": E})
    with pytest.raises(ValueError):
        ElasticConstants.from_params({"E": E, "nu": nu, "mu": 1.0})


def test_build_desc_and_errors():
    from cmad_amd.models.device import NewtonSettings
    from cmad_amd.models.device import NewtonSettings, build_desc
    from cmad_amd.synthetic import hosford_values, j2_voce_values
    d, info = build_desc(j2_voce_values(), newton=NewtonSettings.traced(max_iters=20, abs_tol=1e-12, rel_tol=1e-12))
    assert (d.yield_kind, d.has_voce, d.has_linear, d.rotation_is_identity) == (0, 1, 0, 1)
    np.testing.assert_allclose([d.lmbda, d.mu], [115384.61538461539, 76923.07692307692])
    assert (d.max_iters, d.ls_max_evals, d.ls_c1, d.ls_lo, d.ls_hi) == (20, 4, 1e-4, 0.5, 0.9)
    assert info["elastic_names"] == ("E", "nu")
    d, _ = build_desc(hosford_values())
    assert d.yield_kind == 2 and d.yc[0] == 100.
    from cmad_amd.models.device import BARLAT_NAMES
    bar = j2_voce_values(); bar["plastic"]["effective stress"] = {"barlat": {n: 1.0 + 0.01 * i for i, n in enumerate(BARLAT_NAMES)}}
    d, _ = build_desc(bar)
    assert d.yield_kind == 5 and d.yc[0] == 1.0 and d.yc[18] == pytest.approx(1.18)
    bad = j2_voce_values(); bad["plastic"]["effective stress"] = {"tresca": {}}
    with pytest.raises(NotImplementedError):
        build_desc(bad)
    assert build_desc(j2_voce_values())[0].solver_flags == 0                              # radial-line restriction on
    assert build_desc(j2_voce_values(), newton=NewtonSettings(j2_radial_line=False))[0].solver_flags == 2
    assert build_desc(j2_voce_values(), newton=NewtonSettings(warm_start=False))[0].solver_flags == 8    # CM_SOLVER_REFERENCE_ITERATES


def test_fold_weight_and_data_is_exact():
    from cmad_amd.models.device import fold_weight_and_data
    rng = np.random.default_rng(0)
    w = rng.uniform(0, 1, (3, 3)); d = rng.normal(size=(3, 3, 5)); s6 = rng.normal(size=(6, 5))
    idx = [[0, 1, 2], [1, 3, 4], [2, 4, 5]]
    s = np.array([[s6[idx[i][j]] for j in range(3)] for i in range(3)])
    J_ref = 0.5 * np.sum((w[:, :, None] * (s - d)) ** 2, axis=(0, 1))
    wsq6, data6, const = fold_weight_and_data(w, d)
    J = 0.5 * np.sum(wsq6[:, None] * (s6 - data6) ** 2, axis=0) + const
    np.testing.assert_allclose(J, J_ref, rtol=1e-13)


def test_library_exports_every_declared_symbol():
    from cmad_amd import _lib
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "cmad_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(cm_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/cmad_hip.h but not exported"
    assert set(_lib.EXPORTS) <= declared
    assert L.cm_sizeof_model_desc() == ctypes.sizeof(_lib.ModelDesc)
    from cmad_amd.models.device import build_desc
    from cmad_amd.synthetic import j2_voce_values
    d, _ = build_desc(j2_voce_values())
    assert (L.cm_num_xi(ctypes.byref(d)), L.cm_num_gradu(ctypes.byref(d))) == (7, 9)
    d.def_type = 2
    assert (L.cm_num_xi(ctypes.byref(d)), L.cm_num_gradu(ctypes.byref(d))) == (8, 4)
    assert L.cm_workspace_bytes(1000) >= 13 * 8


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the batched entry points must refuse, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cmad_amd.models.device import DeviceEvaluator, build_desc
    from cmad_amd.synthetic import j2_voce_values
    ev = DeviceEvaluator(*build_desc(j2_voce_values()))
    with pytest.raises(ValueError):
        ev.update(torch.zeros((9, 4), dtype=torch.float64), torch.zeros((7, 4), dtype=torch.float64))


def test_product_never_imports_test_infrastructure():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cmad_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "libcmad_oracle" not in txt and "host_harness" not in txt, f


def test_shard_bounds_cover_batch():
    from cmad_amd.objectives.batched import shard_bounds
    for B in (0, 1, 7, 8, 1000, 80_000_001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(B, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == B
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, B, q):
    import torch
    import torch.distributed as dist
    import host_harness_lib as hh
    import oracle_lib as ol  # noqa: F401
    from cmad_amd.models.device import build_desc, fold_weight_and_data  # noqa: F401
    from cmad_amd.objectives.batched import allreduce_sum_, shard_bounds
    from cmad_amd.synthetic import gauss_point_batch, j2_voce_values
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc, info = build_desc(j2_voce_values())
    gradu = gauss_point_batch(B); xi_prev = np.zeros((7, B))
    sbar = np.random.default_rng(1).normal(size=(6, B))
    lo, hi = shard_bounds(B, rank, world)
    xi, sig, st = hh.update(desc, gradu[:, lo:hi], xi_prev[:, lo:hi], 7)          # this rank's shard (host math)
    g, _, _ = hh.vjp(desc, gradu[:, lo:hi], xi_prev[:, lo:hi], xi, sbar[:, lo:hi])
    out = torch.from_numpy(np.r_[float(hi - lo), g])
    allreduce_sum_(out)
    if rank == 0:
        q.put(out.numpy())
    dist.destroy_process_group()


def test_sharded_gradient_allreduce_gloo_world2():
    """N > 1 path on CPU: two ranks own disjoint shards, exchange only the (1+12)-vector (gloo), and the
    result equals the single-rank sum.  The per-shard numbers come from the host build of the kernel math
    (test infrastructure) since the HIP kernels need a GPU."""
    import torch.multiprocessing as mp
    import host_harness_lib as hh
    from cmad_amd.models.device import build_desc
    from cmad_amd.synthetic import gauss_point_batch, j2_voce_values
    B = 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, _ = build_desc(j2_voce_values())
    gradu = gauss_point_batch(B); xi_prev = np.zeros((7, B)); sbar = np.random.default_rng(1).normal(size=(6, B))
    xi, _, _ = hh.update(desc, gradu, xi_prev, 7)
    g, _, _ = hh.vjp(desc, gradu, xi_prev, xi, sbar)
    assert res[0] == B
    np.testing.assert_allclose(res[1:], g, rtol=1e-12, atol=1e-12 * np.abs(g).max())


def test_affine_scaler_known_values():
    """The reference's tests/neural_networks/test_affine_scaler.py:13-40 on cmad_amd's AffineScaler (the input /
    output scalers of the ICNN yield term)."""
    from cmad_amd.neural_networks import AffineScaler
    samples = np.array([[0.0, 10.0], [2.0, 20.0], [4.0, 30.0]])           # column 0 spans [0, 4], column 1 [10, 30]
    sc = AffineScaler().fit(samples)
    scaled = sc.scale_ * samples + sc.min_
    np.testing.assert_allclose(scaled.min(axis=0), [-1.0, -1.0])
    np.testing.assert_allclose(scaled.max(axis=0), [1.0, 1.0])
    sc = AffineScaler(feature_range=(0.0, 1.0)).fit(samples)
    np.testing.assert_allclose(sc.scale_, [0.25, 0.05])
    np.testing.assert_allclose(sc.min_, [0.0, -0.5])
    const = np.array([[5.0], [5.0], [5.0]])
    sc = AffineScaler(feature_range=(0.0, 1.0)).fit(const)
    np.testing.assert_allclose(sc.scale_ * const + sc.min_, 0.0)


def test_device_quad_min_known_value():
    """The kernels' quad_min (cm_device.hpp, host build) on the reference's known answer
    (tests/util/test_line_search.py:39-55) and its degenerate-denominator branch (line_search.py:74-85)."""
    import ctypes as C
    import host_harness_lib as hh
    f = hh.lib().hh_quad_min
    f.argtypes = [C.c_double] * 4
    f.restype = C.c_double
    q = lambda t: (t - 0.4) ** 2
    assert f(q(0.0), 2.0 * (0.0 - 0.4), 1.0, q(1.0)) == pytest.approx(0.4, abs=1e-12)
    assert f(1.0, -1.0, 0.5, 0.5) == 0.25                     # phi - phi0 - dphi0 a = 0 -> a / 2


def test_cpu_port_of_the_kernel_arithmetic_matches_the_oracle():
    """oracle/cmad_port.cpp (what bench.py times as `cpu_baseline` kind "port": the kernels' own arithmetic in an OpenMP loop)
    against the nested-dual oracle on the bench batch: states, stresses and the parameter gradient of update + vjp, on the J2
    radial line (what the GPU runs by default), on the general structured Newton, and for the Hill surface; one and two threads."""
    import numpy as np
    import oracle_lib as ol
    from cmad_amd.models.device import build_desc
    from cmad_amd.synthetic import AL7079_HILL, gauss_point_batch, j2_voce_values
    B = 6000
    g = gauss_point_batch(B); xp = np.zeros((7, B)); sb = np.random.default_rng(0).normal(size=(6, B))
    for hill in (False, True):
        values = j2_voce_values()
        if hill:
            values["plastic"]["effective stress"] = {"hill": dict(zip("FGHLMN", AL7079_HILL))}
        desc, _ = build_desc(values)
        mat = ol.Material(values)
        xo, so, it, cv = mat.update_batch(ol.newton_settings(), g, xp)
        go, _, _ = mat.update_vjp_batch(g, xp, xo, sb)
        assert cv.all() and (it > 0).any()
        for general, nthreads in ((False, 1), (True, 1), (False, 2)):
            x, s, gk = ol.port_update_and_vjp(desc, g, xp, sb, general=general, nthreads=nthreads)
            np.testing.assert_allclose(x, xo, rtol=1e-10, atol=1e-13)
            np.testing.assert_allclose(s, so, rtol=1e-10, atol=1e-8)
            np.testing.assert_allclose([gk[2], gk[3], gk[4]], [go[ol.P_Y], go[ol.P_VOCE_S], go[ol.P_VOCE_D]], rtol=1e-9)
