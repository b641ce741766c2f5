"""Latency of the material-point objectives of the facade (cmad_amd.objectives.MPAdjointObjective / MPDirectObjective /
MPDirectAdjointObjective) on the reference's own regime -- ONE material point, a 100-step plane-stress history
(tests/objectives/test_J2_fd_checks.py:266-289): each evaluation is 1 / 2 / 4 whole-history launches.
Prints one JSON line per objective; --trace-launches counts kernel launches with the torch profiler."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from cmad_amd.models import DefType, SmallElasticPlastic                                   # noqa: E402
from cmad_amd.objectives import MPAdjointObjective, MPDirectAdjointObjective, MPDirectObjective   # noqa: E402
from cmad_amd.qois import Calibration, UniaxialCalibration                                # noqa: E402
from problems import params_J2_voce, plane_stress_F                                       # noqa: E402


def main():
    F = plane_stress_F(0.02, 50)                                  # 100 steps
    n = F.shape[2]
    model = SmallElasticPlastic(params_J2_voce(), DefType.PLANE_STRESS)
    t = np.linspace(0.0, 1.0, n)
    data = np.zeros((3, 3, n)); data[0, 0] = 260.0 * np.tanh(4 * t); data[1, 1] = 180.0 * t
    w = np.zeros((3, 3)); w[0, 0] = w[1, 1] = 1.0
    qoi = Calibration(model, data, w)
    x = model.parameters.flat_active_values(True)
    for name, cls in (("MPAdjointObjective", MPAdjointObjective), ("MPDirectObjective", MPDirectObjective),
                      ("MPDirectAdjointObjective", MPDirectAdjointObjective)):
        obj = cls(qoi, F)
        for _ in range(3):
            r = obj.evaluate(x)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            r = obj.evaluate(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        launches = None
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                obj.evaluate(x)
                torch.cuda.synchronize()
            launches = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and
                           ("k_" in e.name or "cm" in e.name))
        except Exception as e:                                     # noqa: BLE001
            launches = f"n/a ({type(e).__name__})"
        print(json.dumps({"objective": name, "history_steps": n - 1, "points": 1, "ms_per_evaluation": ms,
                          "library_kernel_launches": launches, "J": r.J}), flush=True)


if __name__ == "__main__":
    main()
