"""Deterministic synthetic Gauss-point batches (BASELINE.md section 3 / SURVEY.md section 8d).

Material = the reference's J2AnalyticalProblem values (tests/support/test_problems.py:151 in the
reference): E=2e5, nu=0.3, Y=200, Voce S=200, D=20, Q=I.  Per point a random symmetric strain with
deviatoric magnitude U(0, 4 eps_y) and volumetric part U(-eps_y, eps_y), eps_y = Y/E; about half the
points yield.  grad u = strain (+ optional skew part, which the small-strain model ignores).
"""
import numpy as np

SEED = 22


def j2_voce_values(E=200e3, nu=0.3, Y=200., S=200., D=20.):
    return {
        "rotation matrix": np.eye(3),
        "elastic": {"E": E, "nu": nu},
        "plastic": {"effective stress": {"J2": 0.},
                    "flow stress": {"initial yield": {"Y": Y}, "hardening": {"voce": {"S": S, "D": D}}}}}


def hosford_values(a=100., E=1000., nu=0.25, Y=2., S=10., D=2.):
    """examples/notch_hosford.yaml:36-42 material."""
    v = j2_voce_values(E, nu, Y, S, D)
    v["plastic"]["effective stress"] = {"hosford": {"a": a}}
    return v


AL7079_HILL = (0.1477, 0.6805, 0.5345, 1.7977, 1.7148, 2.1675)    # F, G, H, L, M, N (calibrations/al7079/support.py:76-78)


def al7079_hybrid_setup(layer_widths=(6, 16, 1)):
    """BASELINE.json configs[3]: hybrid Hill + symmetric ICNN [6, 16, 1] yield surface with the Al7079 elastic
    constants and Hill coefficients (cmad/calibrations/al7079/support.py:76-78,
    nn_hill_uniaxial_stress_forward.py:84), weights from the seeded initialiser (seed 22), input scaler on
    (0, 1) with zero offset and output scaler on the +/- sigma_c range as in
    fit_hybrid_icnn_effective_stress.py:46-63,258-263.  The trained pickle is not in the reference repo.
    Returns (icnn, parameter values)."""
    from .neural_networks import AffineScaler, InputConvexNeuralNetwork
    sig_c = np.array([525., 512., 515., 505., 493., 511., 530., 510., 544., 523., 486., 485.])
    feats = np.abs(np.random.default_rng(22).normal(size=(24, 6))) * 300.0 + 50.0      # stand-in deviator samples
    in_sc = AffineScaler(feature_range=(0.0, 1.0)).fit(np.vstack([feats, np.zeros((1, 6))]))
    in_sc.min_ = in_sc.min_ * 0.0
    out_sc = AffineScaler(feature_range=(0.0, 1.0)).fit(np.r_[-sig_c, sig_c].reshape(-1, 1))
    icnn = InputConvexNeuralNetwork(list(layer_widths), in_sc, out_sc, seed=22)
    values = j2_voce_values(E=70.22857142857143e3, nu=0.33396551724137924, Y=525.0, S=200., D=20.)
    values["plastic"]["effective stress"] = {"hill": dict(zip("FGHLMN", AL7079_HILL)),
                                             "neural network": icnn.params}
    return icnn, values


def gauss_point_batch(B, eps_y=1e-3, seed=SEED, skew=False, dev_scale=4.0, chunk=1 << 20, ndims=3):
    """Returns gradu (ndims^2, B) float64 SoA (numpy).  Chunked so 1e7 points need < 1 GB transient."""
    rng = np.random.default_rng(seed)
    nu = ndims * ndims
    out = np.empty((nu, B), dtype=np.float64)
    for s in range(0, B, chunk):
        n = min(chunk, B - s)
        A = rng.standard_normal((n, 3, 3))
        A = 0.5 * (A + A.transpose(0, 2, 1))
        A -= np.trace(A, axis1=1, axis2=2)[:, None, None] / 3.0 * np.eye(3)
        A /= np.linalg.norm(A, axis=(1, 2))[:, None, None]
        m = rng.uniform(0.0, dev_scale * eps_y, n) + 1e-12          # never exactly zero strain
        v = rng.uniform(-eps_y, eps_y, n)
        E = m[:, None, None] * A + v[:, None, None] / 3.0 * np.eye(3)
        if skew:
            W = rng.standard_normal((n, 3, 3)) * eps_y
            E = E + 0.5 * (W - W.transpose(0, 2, 1))
        out[:, s:s + n] = E[:, :ndims, :ndims].reshape(n, nu).T
    return out
