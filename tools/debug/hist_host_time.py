"""Where does BatchedCalibrationObjective.evaluate_native spend host time?  (diagnostic; run on the GPU box)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from cmad_amd.models import DefType, SmallElasticPlastic
from cmad_amd.objectives import BatchedCalibrationObjective
from cmad_amd.parameters import Parameters
from cmad_amd.parameters.parameters import tree_map
from cmad_amd.synthetic import gauss_point_batch, j2_voce_values

dev = torch.device("cuda:0")
for ps in (False, True):
    K, B = 10, 2_000_000
    values = j2_voce_values()
    flags = tree_map(lambda leaf: False, values)
    flags["plastic"]["flow stress"] = tree_map(lambda leaf: True, flags["plastic"]["flow stress"])
    model = SmallElasticPlastic(Parameters(values, flags, tree_map(lambda leaf: None, values)), DefType.PLANE_STRESS if ps else DefType.FULL_3D)
    g1 = torch.from_numpy(gauss_point_batch(B, seed=22, ndims=2 if ps else 3)).to(dev)
    ramp = torch.linspace(0.0, 1.5, K + 1, dtype=torch.float64, device=dev)
    gh = (ramp[:, None, None] * g1[None]).contiguous()
    dh = 50.0 * torch.randn((K + 1, 6, B), dtype=torch.float64, device=dev)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.0
    obj = BatchedCalibrationObjective(model, gh, dh, weight, fused_history=True)
    for _ in range(3):
        obj.evaluate_native()
    torch.cuda.synchronize()
    T = {}
    def tick(name, t0):
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    for _ in range(5):
        t0 = time.perf_counter(); ev = model.device_evaluator(obj._newton); tick("device_evaluator", t0)
        t0 = time.perf_counter(); ev.objective_grad_history(obj._g, obj._d, obj._wsq6, obj._xi0, xi_hist=obj._xi_hist, out=obj._out); tick("objective_grad_history", t0)
        t0 = time.perf_counter(); res = obj._out.cpu().numpy(); tick("out.cpu", t0)
        t0 = time.perf_counter(); model.active_grad_from_kp(res[1:], ev.info); tick("active_grad", t0)
        t0 = time.perf_counter(); obj.evaluate_native(); tick("evaluate_native", t0)
    print("PLANE_STRESS" if ps else "FULL_3D", {k: round(v / 5 * 1e3, 3) for k, v in T.items()})
