"""Round-trip scenarios for the `cmad` command line, shared by the CPU run (model classes re-routed to the host
build of the kernel math, tests/host_facade.py) and the GPU run.  Each mirrors one of the reference's
tests/cli/test_{primal,objective,gradient,hessian,calibrate}_roundtrip.py: write a deck + .npy inputs into a
temp dir, call `main([...])`, read the files back."""
import json
from pathlib import Path

import numpy as np
import yaml

GOLDEN = Path(__file__).parent / "golden"


def j2_parameters(E=200_000.0, nu=0.3, Y=200.0, S=200.0, D=20.0, active=False, factor=1.0):
    def leaf(v, ref):
        return {"value": v * factor, "active": True, "transform": {"log": ref}} if active else v
    return {
        "rotation matrix": [[1, 0, 0], [0, 1, 0], [0, 0, 1]],
        "elastic": {"E": E, "nu": nu},
        "plastic": {
            "effective stress": {"J2": 0.0},
            "flow stress": {"initial yield": {"Y": leaf(Y, Y)},
                            "hardening": {"voce": {"S": leaf(S, S), "D": leaf(D, D)}}},
        },
    }


def base_deck(tmp: Path, def_type: str, parameters: dict, **sections) -> dict:
    deck = {"problem": {"type": "material_point"},
            "model": {"name": "small_elastic_plastic", "def_type": def_type, "effective_stress": "J2"},
            "parameters": parameters,
            "deformation": {"history_file": str(tmp / "F.npy")}}
    deck.update(sections)
    deck["output"] = {"path": str(tmp / "out")}
    return deck


def write_deck(tmp: Path, deck: dict, name="deck.yaml") -> Path:
    path = tmp / name
    path.write_text(yaml.safe_dump(deck, sort_keys=False))
    return path


def analytical_history():
    """30-step uniaxial-stress J2+Voce history from the reference's closed-form solution (golden fixture)."""
    g = np.load(GOLDEN / "j2_voce_analytical.npz")
    strain, stress = g["uniaxial30_strain"], g["uniaxial30_stress"]
    n = strain.shape[2]
    F = np.repeat(np.eye(3)[:, :, None], n + 1, axis=2)
    F[:, :, 1:] += strain
    return F, stress


def biaxial_F(num_pts=50, inc=0.02):
    """Plane-stress history of the reference's calibrate round trip: ramp xx, then hold xx and ramp yy."""
    first = inc / num_pts
    exx = np.r_[0.0, np.linspace(first, inc, num_pts), np.full(num_pts, inc)]
    eyy = np.r_[0.0, np.zeros(num_pts), np.linspace(first, inc, num_pts)]
    F = np.repeat(np.eye(2)[:, :, None], 2 * num_pts + 1, axis=2)
    F[0, 0, :] += exx
    F[1, 1, :] += eyy
    return F


def check_primal(main, tmp: Path):
    """reference tests/cli/test_primal_roundtrip.py:17-66."""
    F, stress_ref = analytical_history()
    np.save(tmp / "F.npy", F)
    deck = base_deck(tmp, "full_3d", j2_parameters())
    assert main(["primal", str(write_deck(tmp, deck))]) == 0
    out = tmp / "out"
    cauchy = np.load(out / "cauchy.npy")
    assert cauchy.shape == (3, 3, F.shape[2])
    np.testing.assert_allclose(cauchy[:, :, 1:], stress_ref, rtol=1e-6, atol=1e-8)
    xi0, xi1 = np.load(out / "xi_block_00.npy"), np.load(out / "xi_block_01.npy")
    assert xi0.shape == (F.shape[2], 6) and xi1.shape == (F.shape[2], 1)
    assert np.all(np.diff(xi1[:, 0]) >= -1e-15)                     # alpha never decreases
    log = json.loads((out / "solver.json").read_text())
    assert len(log) == F.shape[2] - 1 and set(log[0]) == {"iters", "final_residual"}
    assert max(e["final_residual"] for e in log) < 1e-10
    resolved = yaml.safe_load((out / "deck.resolved.yaml").read_text())
    assert resolved["solver"]["newton"] == {"max_iters": 10, "abs_tol": 1e-14, "rel_tol": 1e-14, "max_ls_evals": 0}
    assert resolved["output"]["format"] == "npy" and resolved["output"]["prefix"] == ""

    # text format + prefix
    deck["output"].update({"format": "text", "prefix": "run1_", "path": str(tmp / "out_txt")})
    assert main(["primal", str(write_deck(tmp, deck, "deck_txt.yaml"))]) == 0
    rows = np.loadtxt(tmp / "out_txt" / "run1_cauchy.csv")
    assert rows.shape == (F.shape[2], 9)
    np.testing.assert_allclose(rows.reshape(-1, 3, 3).transpose(1, 2, 0), cauchy, rtol=1e-15, atol=1e-12)
    assert (tmp / "out_txt" / "run1_cauchy.csv").read_text().startswith("# S11 S12 S13 S21")
    assert np.loadtxt(tmp / "out_txt" / "run1_xi_block_00.csv").shape == (F.shape[2], 6)

    # no output block: runs, writes nothing
    del deck["output"]
    before = sorted(p.name for p in tmp.iterdir())
    assert main(["primal", str(write_deck(tmp, deck, "deck_quiet.yaml"))]) == 0
    assert sorted(p.name for p in tmp.iterdir()) == sorted(before + ["deck_quiet.yaml"])
    return cauchy


def check_objective(main, tmp: Path):
    """J = 0 at truth (reference tests/cli/test_objective_roundtrip.py:47), J > 0 off truth."""
    F, _ = analytical_history()
    np.save(tmp / "F.npy", F)
    assert main(["primal", str(write_deck(tmp, base_deck(tmp, "full_3d", j2_parameters())))]) == 0
    np.save(tmp / "data.npy", np.load(tmp / "out" / "cauchy.npy"))
    qoi = {"name": "calibration", "data_file": str(tmp / "data.npy"), "weight": [[1, 0, 0], [0, 0, 0], [0, 0, 0]]}
    deck = base_deck(tmp, "full_3d", j2_parameters(), qoi=qoi)
    deck["output"]["path"] = str(tmp / "obj")
    assert main(["objective", str(write_deck(tmp, deck, "obj.yaml"))]) == 0
    assert json.loads((tmp / "obj" / "J.json").read_text())["J"] < 1e-16
    for name in ("cauchy.npy", "xi_block_00.npy", "solver.json", "deck.resolved.yaml"):
        assert (tmp / "obj" / name).exists()
    deck["parameters"] = j2_parameters(Y=220.0)
    assert main(["objective", str(write_deck(tmp, deck, "obj2.yaml"))]) == 0
    assert json.loads((tmp / "obj" / "J.json").read_text())["J"] > 1.0


def _plane_stress_problem(main, tmp: Path, num_pts):
    F = biaxial_F(num_pts)
    np.save(tmp / "F.npy", F)
    truth = base_deck(tmp, "plane_stress", j2_parameters(E=70_000.0))
    truth["output"]["path"] = str(tmp / "truth")
    assert main(["primal", str(write_deck(tmp, truth, "truth.yaml"))]) == 0
    np.save(tmp / "cauchy_data.npy", np.load(tmp / "truth" / "cauchy.npy"))
    return {"name": "calibration", "data_file": str(tmp / "cauchy_data.npy"),
            "weight": [[1, 0, 0], [0, 1, 0], [0, 0, 0]]}


def check_gradient(main, tmp: Path, strategies):
    """All sensitivity strategies write the same (J, grad) (reference tests/cli/test_gradient_roundtrip.py:98)."""
    qoi = _plane_stress_problem(main, tmp, 6)
    results = {}
    for kind in strategies:
        deck = base_deck(tmp, "plane_stress", j2_parameters(E=70_000.0, active=True, factor=1.1),
                         qoi=qoi, sensitivity={"type": kind})
        deck["output"]["path"] = str(tmp / f"grad_{kind}")
        assert main(["gradient", str(write_deck(tmp, deck, f"grad_{kind}.yaml"))]) == 0
        J = json.loads((tmp / f"grad_{kind}" / "J.json").read_text())["J"]
        g = np.load(tmp / f"grad_{kind}" / "grad.npy")
        assert g.shape == (3,) and J > 0
        results[kind] = (J, g)
    ref = results[strategies[0]]
    for kind in strategies[1:]:
        np.testing.assert_allclose(results[kind][0], ref[0], rtol=1e-10)
        np.testing.assert_allclose(results[kind][1], ref[1], rtol=1e-8, atol=1e-10 * np.abs(ref[1]).max())
    return ref


def check_hessian(main, tmp: Path, strategies):
    """direct_adjoint (and jvp) Hessians agree, are symmetric, and `hessian` refuses first-order strategies
    (reference tests/cli/test_hessian_roundtrip.py:95)."""
    import pytest
    qoi = _plane_stress_problem(main, tmp, 4)
    got = {}
    for kind in strategies:
        deck = base_deck(tmp, "plane_stress", j2_parameters(E=70_000.0, active=True, factor=1.1),
                         qoi=qoi, sensitivity={"type": kind})
        deck["output"]["path"] = str(tmp / f"hess_{kind}")
        assert main(["hessian", str(write_deck(tmp, deck, f"hess_{kind}.yaml"))]) == 0
        H = np.load(tmp / f"hess_{kind}" / "hess.npy")
        g = np.load(tmp / f"hess_{kind}" / "grad.npy")
        assert H.shape == (3, 3) and g.shape == (3,)
        np.testing.assert_allclose(H, H.T, rtol=1e-8, atol=1e-8 * np.abs(H).max())
        got[kind] = (g, H)
    first = got[strategies[0]]
    for kind in strategies[1:]:
        np.testing.assert_allclose(got[kind][0], first[0], rtol=1e-8, atol=1e-10 * np.abs(first[0]).max())
        np.testing.assert_allclose(got[kind][1], first[1], rtol=1e-7, atol=1e-9 * np.abs(first[1]).max())
    deck["sensitivity"] = {"type": "adjoint"}
    with pytest.raises(ValueError, match="requires 'direct_adjoint' or 'jvp'"):
        main(["hessian", str(write_deck(tmp, deck, "hess_bad.yaml"))])


def check_calibrate(main, tmp: Path, num_pts=50, kind="adjoint"):
    """Start 10% off, recover [Y, S, D] = [200, 200, 20] (reference tests/cli/test_calibrate_roundtrip.py:146-199)."""
    qoi = _plane_stress_problem(main, tmp, num_pts)
    deck = base_deck(tmp, "plane_stress", j2_parameters(E=70_000.0, active=True, factor=1.1), qoi=qoi,
                     sensitivity={"type": kind},
                     optimizer={"algorithm": "L-BFGS-B", "options": {"ftol": 1e-14, "gtol": 1e-10, "maxiter": 200}})
    assert main(["calibrate", str(write_deck(tmp, deck, "cal.yaml"))]) == 0
    out = tmp / "out"
    status = json.loads((out / "opt_status.json").read_text())
    assert status["success"] and status["fun"] < 1e-10
    assert {"nfev", "njev", "nit", "message", "status"} <= set(status) and "x" not in status
    flow = yaml.safe_load((out / "opt_params.yaml").read_text())["parameters"]["plastic"]["flow stress"]
    np.testing.assert_allclose(flow["initial yield"]["Y"]["value"], 200.0, rtol=1e-5)
    np.testing.assert_allclose(flow["hardening"]["voce"]["S"]["value"], 200.0, rtol=1e-5)
    np.testing.assert_allclose(flow["hardening"]["voce"]["D"]["value"], 20.0, rtol=1e-5)
    assert flow["initial yield"]["Y"]["active"] is True and flow["initial yield"]["Y"]["transform"] == {"log": 200.0}
    hist = json.loads((out / "opt_history.json").read_text())
    assert hist["active_param_paths"] == ["plastic.flow_stress.hardening.voce.D", "plastic.flow_stress.hardening.voce.S",
                                          "plastic.flow_stress.initial_yield.Y"]
    assert len(hist["history"]) == status["nfev"] and len(hist["history"][0]["params"]) == 3
    np.testing.assert_allclose(hist["history"][0]["params"], [22.0, 220.0, 220.0], rtol=1e-12)
    with_bad = dict(deck, sensitivity={"type": "direct_adjoint"})
    import pytest
    with pytest.raises(ValueError, match="first-order only"):
        main(["calibrate", str(write_deck(tmp, with_bad, "cal_bad.yaml"))])
