import numpy as np, torch, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib as ol
from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
from cmad_amd.synthetic import gauss_point_batch
B = 2_000_000
values = ol.j2_voce_values()
for rl in (True, False):
    desc, info = build_desc(values, newton=NewtonSettings(j2_radial_line=rl))
    ev = DeviceEvaluator(desc, info)
    gradu = torch.from_numpy(gauss_point_batch(B, seed=22)).cuda()
    xi_prev = torch.zeros((7, B), dtype=torch.float64, device="cuda")
    xi, sig, st = ev.update(gradu, xi_prev)
    xi2, sig2, st2 = ev.update(gradu, xi)
    it2 = (st2.to(torch.int64) & 0xFFFF)
    bad = (it2 > 0).nonzero().flatten()
    print("radial" if rl else "general", "lanes iterating on re-application:", bad.numel(), "max it", int(it2.max()))
    if bad.numel():
        b = bad[:5]
        print(" st1", (st[b].to(torch.int64) & 0xFFFF).tolist(), "alpha", xi[6, b].tolist())
        print(" dxi", (xi2[:, b] - xi[:, b]).abs().max(0).values.tolist())
        # true f at xi
        E, nu, Y, S, D = 200e3, 0.3, 200.0, 200.0, 20.0
        w = torch.tensor([1., 2., 2., 1., 2., 1.], dtype=torch.float64, device="cuda")[:, None]
        p = (sig[0] + sig[3] + sig[5]) / 3
        dev = sig.clone(); dev[0] -= p; dev[3] -= p; dev[5] -= p
        vm = torch.sqrt(1.5 * (w * dev * dev).sum(0))
        f = (vm - Y - S * (1 - torch.exp(-D * xi[6]))) / (E / (1 + nu))
        print(" f at xi", f[b].tolist())
