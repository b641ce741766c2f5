"""Generate the golden fixtures under tests/golden/ from the reference itself.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_golden.py

What is executed from the reference: the JAX-free file
``cmad/verification/solutions.py`` (numpy only), loaded *by path* so that
``cmad/__init__.py`` (which imports jax, not installed here) is never touched.
Its ``compute_plastic_fields`` / ``compute_elastic_fields`` produce the
analytical known-answer histories that the reference's own tests compare the
models against:

* tests/models/test_elastic_plastic_models.py:15-125 (J2 + Voce, uniaxial and
  +/- biaxial stress masks, 100 steps to alpha = 0.5, tolerance 1e-6),
* tests/models/test_hill_material_rotations.py:40-158 (Hill + Voce, twelve
  Al7079 slab orientations, 100 elastic + 100 plastic steps, tolerance 1e-8).

The yield functions handed to ``compute_plastic_fields`` live in
``cmad/verification/functions.py``, which imports jax at module level; the
closed-form numpy expressions (von Mises and Hill-48 effective stress and their
normals) are therefore restated below -- a few lines of textbook math each.

Only DATA is written (inputs + expected outputs), never reference source.
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_solutions():
    path = os.path.join(REF, "cmad", "verification", "solutions.py")
    spec = importlib.util.spec_from_file_location("_ref_solutions", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- closed-form yield functions (inputs to the reference's solution code) ----
def vm_phi(c):
    s = c - np.trace(c) / 3.0 * np.eye(3)
    return np.sqrt(1.5) * np.linalg.norm(s)


def vm_normal(c):
    s = c - np.trace(c) / 3.0 * np.eye(3)
    return np.sqrt(1.5) * s / np.linalg.norm(s)


def hill_phi(c, h):
    F, G, H, L, M, N = h
    return np.sqrt(F * (c[1, 1] - c[2, 2]) ** 2 + G * (c[2, 2] - c[0, 0]) ** 2
                   + H * (c[0, 0] - c[1, 1]) ** 2
                   + 2.0 * (L * c[1, 2] ** 2 + M * c[0, 2] ** 2 + N * c[0, 1] ** 2))


def hill_normal(c, h):
    F, G, H, L, M, N = h
    n = np.array([
        [(G + H) * c[0, 0] - H * c[1, 1] - G * c[2, 2], N * c[0, 1], M * c[0, 2]],
        [N * c[0, 1], (F + H) * c[1, 1] - H * c[0, 0] - F * c[2, 2], L * c[1, 2]],
        [M * c[0, 2], L * c[1, 2], (G + F) * c[2, 2] - G * c[0, 0] - F * c[1, 1]]])
    return n / hill_phi(c, h)


# ---- Al7079 slab orientations: DATA of cmad/calibrations/al7079/support.py:12-69 ----
def _R_from_basis(basis):
    return np.array([[np.eye(3)[i] @ basis[j] for j in range(3)] for i in range(3)])


def slab_rotations():
    d2r = np.pi / 180.0
    Rs = []
    for a in np.array([0., 15., 30., 45., 60., 75., 90.]) * d2r:
        Rs.append(_R_from_basis(np.array([[-1., 0., 0.], [0., np.sin(a), np.cos(a)], [0., np.cos(a), -np.sin(a)]])))
    for b in np.array([45., 60., 90.]) * d2r:
        Rs.append(_R_from_basis(np.array([[0., np.sin(b), np.cos(b)], [1., 0., 0.], [0., np.cos(b), -np.sin(b)]])))
    for g in np.array([45., 60.]) * d2r:
        Rs.append(_R_from_basis(np.array([[np.cos(g), np.sin(g), 0.], [-np.sin(g), np.cos(g), 0.], [0., 0., 1.]])))
    return Rs


def main():
    sol = _load_solutions()

    # (1) J2 + Voce analytical fields, tests/support/test_problems.py:142-162
    iso = np.array([200e3, 0.3, 200., 200., 20.])   # E, nu, Y, S, D
    masks = {"uniaxial": np.diag([1., 0., 0.]), "biaxial": np.diag([1., -1., 0.])}
    out = {"isotropic_params": iso, "max_alpha": 0.5, "num_steps": 100}
    for name, mask in masks.items():
        stress, strain, alpha = sol.compute_plastic_fields(mask, vm_phi, vm_normal, iso, 0.5, 100)
        out[f"{name}_mask"] = mask
        out[f"{name}_stress"] = stress
        out[f"{name}_strain"] = strain
        out[f"{name}_alpha"] = alpha
    # the 5-step value quoted in SURVEY.md section 8(c)
    s5, _, _ = sol.compute_plastic_fields(np.diag([1., 0., 0.]), vm_phi, vm_normal, iso, 0.5, 5)
    out["uniaxial_5step_sigma_xx"] = s5[0, 0, :]
    # 30-step FULL_3D uniaxial-stress history of tests/cli/test_gradient_roundtrip.py:97-148
    s30, e30, a30 = sol.compute_plastic_fields(np.diag([1., 0., 0.]), vm_phi, vm_normal, iso, 0.5, 30)
    out["uniaxial30_stress"] = s30; out["uniaxial30_strain"] = e30; out["uniaxial30_alpha"] = a30
    np.savez_compressed(os.path.join(HERE, "j2_voce_analytical.npz"), **out)

    # (2) Hill rotations, tests/models/test_hill_material_rotations.py:40-158
    hill = np.array([0.1477, 0.6805, 0.5345, 1.7977, 1.7148, 2.1675])   # support.py:76-78
    p_el = np.array([70.22857142857143e3, 0.33396551724137924])
    Y = 525.0                                                           # alpha_sigma_c_values[0]
    iso_h = np.array([p_el[0], p_el[1], Y, 200., 20.])
    mask = np.diag([0., 1., 0.])
    Rs = slab_rotations()
    stresses, strains = [], []
    for R in Rs:
        mmask = R.T @ mask @ R
        ps, pe, _ = sol.compute_plastic_fields(mmask, lambda c: hill_phi(c, hill), lambda c: hill_normal(c, hill),
                                               iso_h, 0.1, 100)
        es, ee = sol.compute_elastic_fields(ps[:, :, 0], 0.1, 0.99, p_el, 100)
        ms = np.dstack([es, ps]); me = np.dstack([ee, pe])
        stresses.append(np.dstack([R @ ms[:, :, i] @ R.T for i in range(200)]))
        strains.append(np.dstack([R @ me[:, :, i] @ R.T for i in range(200)]))
    np.savez_compressed(os.path.join(HERE, "hill_rotations.npz"),
                        hill=hill, elastic=p_el, Y=Y, voce=np.array([200., 20.]),
                        R=np.array(Rs), stress=np.array(stresses), strain=np.array(strains))
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are committed, nothing to do")
    main()
