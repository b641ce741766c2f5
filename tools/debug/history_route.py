"""K-step calibration objective + gradient for the iteration-bound surfaces: the single lockstep history kernel against per-step
launches (work-pool update forward, adjoint step backward).  Prints ms per evaluation."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from cmad_amd.models import DefType, SmallElasticPlastic
from cmad_amd.models.device import HybridHillEffectiveStress, NewtonSettings
from cmad_amd.objectives import BatchedCalibrationObjective
from cmad_amd.parameters import Parameters
from cmad_amd.parameters.parameters import tree_map
from cmad_amd.synthetic import al7079_hybrid_setup, gauss_point_batch, hosford_values

B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 5
for name in ("hybrid", "hosford100"):
    if name == "hybrid":
        icnn, values = al7079_hybrid_setup()
        eff, eps_y = HybridHillEffectiveStress(icnn), 525.0 / 70.2e3
        newton = NewtonSettings.traced(max_iters=50, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 10})
    else:
        values, eff, eps_y = hosford_values(), None, 2e-3
        newton = NewtonSettings.traced(max_iters=500, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 100})
    flags = tree_map(lambda leaf: False, values)
    flags["plastic"]["flow stress"] = tree_map(lambda leaf: True, flags["plastic"]["flow stress"])
    model = SmallElasticPlastic(Parameters(values, flags, tree_map(lambda leaf: None, values)), DefType.FULL_3D,
                                **({"effective_stress_fun": eff} if eff is not None else {}))
    g1 = torch.from_numpy(gauss_point_batch(B, seed=5, eps_y=eps_y)).cuda()
    ramp = torch.linspace(0.0, 1.5, K + 1, dtype=torch.float64, device="cuda")
    gh = (ramp[:, None, None] * g1[None]).contiguous()
    dh = 50.0 * torch.randn((K + 1, 6, B), dtype=torch.float64, device="cuda")
    w = np.zeros((3, 3)); w[0, 0] = w[1, 1] = 1.0
    res = {}
    for fused in (True, False, None):
        obj = BatchedCalibrationObjective(model, gh, dh, w, newton=newton, fused_history=fused)
        r = obj.evaluate_native(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = obj.evaluate_native()
        torch.cuda.synchronize()
        res[fused] = ((time.perf_counter() - t0) / 3 * 1e3, r.J, r.grad)
        print(name, "fused_history =", fused, "| %.2f ms per evaluation | J = %.10e" % (res[fused][0], r.J), flush=True)
    np.testing.assert_allclose(res[False][2], res[True][2], rtol=1e-9, atol=1e-9 * np.abs(res[True][2]).max())
