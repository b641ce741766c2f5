"""Latency of one calibration objective + gradient evaluation over a K-step history for small batches (the
reference's own regime: a handful of material points, 100+ load steps): cm_objective_grad_history (one launch)
against one launch per step and direction.  Prints one JSON line per case."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cmad_amd.models import DefType, SmallElasticPlastic          # noqa: E402
from cmad_amd.objectives import BatchedCalibrationObjective        # noqa: E402
from cmad_amd.parameters import Parameters                         # noqa: E402
from cmad_amd.parameters.parameters import tree_map                # noqa: E402
from cmad_amd.synthetic import gauss_point_batch, j2_voce_values   # noqa: E402


def case(B, K, fused, reps=20):
    dev = torch.device("cuda:0")
    values = j2_voce_values()
    flags = tree_map(lambda leaf: False, values)
    flags["plastic"]["flow stress"] = tree_map(lambda leaf: True, flags["plastic"]["flow stress"])
    model = SmallElasticPlastic(Parameters(values, flags, tree_map(lambda leaf: None, values)), DefType.PLANE_STRESS)
    g1 = torch.from_numpy(gauss_point_batch(B, seed=22, ndims=2)).to(dev)
    ramp = torch.linspace(0.0, 1.5, K + 1, dtype=torch.float64, device=dev)
    gh = (ramp[:, None, None] * g1[None]).contiguous()
    gen = torch.Generator(device=dev); gen.manual_seed(99)
    dh = 50.0 * torch.randn((K + 1, 6, B), dtype=torch.float64, device=dev, generator=gen)
    w = np.zeros((3, 3)); w[0, 0] = w[1, 1] = 1.0
    obj = BatchedCalibrationObjective(model, gh, dh, w, fused_history=fused)
    for _ in range(3):
        r = obj.evaluate_native()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = obj.evaluate_native()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps({"points": B, "history_steps": K, "fused_history": fused, "ms_per_evaluation": ms,
                      "point_steps_per_s": B * K / ms * 1e3, "J": r.J}), flush=True)


if __name__ == "__main__":
    for B, K in ((1, 100), (10_648, 100), (1_000_000, 20)):
        for fused in (True, False):
            case(B, K, fused)
