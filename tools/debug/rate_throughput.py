"""Throughput of the rate-form model's kernels (cm_update_rate, cm_update_rate_and_vjp) against the total form's on the same batch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from cmad_amd.models.deformation_types import DefType
from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
from cmad_amd.synthetic import gauss_point_batch, j2_voce_values

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for dt, nd, nx in ((DefType.FULL_3D, 3, 7), (DefType.PLANE_STRESS, 2, 8)):
    g = torch.from_numpy(gauss_point_batch(B, seed=22, ndims=nd)).cuda()
    gp = torch.zeros_like(g)
    sb = torch.randn((6, B), dtype=torch.float64, device="cuda")
    for mk, name in ((0, "total form"), (1, "rate form")):
        ev = DeviceEvaluator(*build_desc(j2_voce_values(), def_type=dt, model_kind=mk, newton=NewtonSettings()))
        xp = torch.zeros((nx, B), dtype=torch.float64, device="cuda")
        if dt == DefType.PLANE_STRESS:
            xp[7] = 1.0
        def run(fused):
            if mk == 0:
                return ev.update_and_vjp(g, xp, sb) if fused else ev.update(g, xp, want_status=False)
            return ev.update_and_vjp(g, xp, sb, gradu_prev=gp) if fused else ev.update_rate(g, gp, xp, want_status=False)
        for fused in (False, True):
            for _ in range(2):
                run(fused)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                run(fused)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            print(f"{dt.name:12s} {name:10s} {'update+vjp' if fused else 'update    '} {ms:7.3f} ms  {B / ms * 1e3:.3g} points/s", flush=True)
