import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mode = sys.argv[1]
from cmad_amd.models.device import DeviceEvaluator, build_desc
from cmad_amd.synthetic import gauss_point_batch, j2_voce_values
B = 2048
if mode == "cuda_first":
    torch.zeros(1, device="cuda")
desc, info = build_desc(j2_voce_values())
ev = DeviceEvaluator(desc, info)
t = lambda a: torch.from_numpy(a).to("cuda:0")
g, xp, sb = t(gauss_point_batch(B)), t(np.zeros((7, B))), t(np.random.default_rng(0).normal(size=(6, B)))
try:
    if mode == "update_first":
        ev.update(g, xp); torch.cuda.synchronize(); print(mode, "update ok")
    if mode == "vjp_first":
        xi, _, _ = ev.update(g, xp)
    r = ev.update_and_vjp(g, xp, sb); torch.cuda.synchronize(); print(mode, "update_and_vjp ok", r[2][:5].cpu().numpy())
except Exception as e:
    print(mode, "FAILED", e)
