"""Print cm_objective_grad's 13 numbers for a seeded batch (A/B check of library variants: values must agree to rounding)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sys
import numpy as np
import torch
from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
from cmad_amd.synthetic import gauss_point_batch, j2_voce_values

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_017
dev = torch.device("cuda:0")
desc, info = build_desc(j2_voce_values(), newton=NewtonSettings())
ev = DeviceEvaluator(desc, info)
g = torch.from_numpy(gauss_point_batch(B, seed=5, eps_y=1e-3)).to(dev)
xp = torch.zeros((7, B), dtype=torch.float64, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(7)
data = 100.0 * torch.randn((6, B), dtype=torch.float64, device=dev, generator=gen)
res = torch.empty(13, dtype=torch.float64, device=dev)
ev.objective_grad(g, xp, data, [1., 2., 1., .5, 1., 3.], out=res)
torch.cuda.synchronize()
print(" ".join("%.13e" % v for v in res.cpu().numpy()))
