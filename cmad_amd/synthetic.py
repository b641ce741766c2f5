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
