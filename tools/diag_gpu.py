"""GPU-side diagnostics: which HIP runtime is mapped, does a trivial launch through the C-ABI work."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

print("torch", torch.__version__, "hip", torch.version.hip, "avail", torch.cuda.is_available())
x = torch.zeros(4, device="cuda")
print("device:", torch.cuda.get_device_name(0))
from cmad_amd import _lib
L = _lib.lib()
maps = open('/proc/self/maps').read()
print("\n".join(sorted(set(l.split()[-1] for l in maps.splitlines() if 'amdhip64' in l or 'hsa-runtime' in l or 'cmad' in l))))
from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
from cmad_amd.synthetic import gauss_point_batch, j2_voce_values
desc, info = build_desc(j2_voce_values())
ev = DeviceEvaluator(desc, info)
B = 256
g = torch.from_numpy(gauss_point_batch(B)).cuda()
xp = torch.zeros((7, B), dtype=torch.float64, device="cuda")
try:
    xi, sig, st = ev.update(g, xp)
    torch.cuda.synchronize()
    print("update ok", xi[:, :2].cpu().numpy(), st[:8].cpu().numpy())
except Exception as e:
    print("update failed:", e)
hip = C.CDLL("libamdhip64.so.7")
n = C.c_int(-1)
print("hipGetDeviceCount rc", hip.hipGetDeviceCount(C.byref(n)), n.value)
print("hipGetLastError", hip.hipGetLastError())

# ---- compare with the oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as ol
values = ol.j2_voce_values()
mat = ol.Material(values)
B = 512
gradu = gauss_point_batch(B, seed=22, skew=True)
xi_prev = np.zeros((7, B))
xi_o, sig_o, it_o, cv_o = mat.update_batch(ol.newton_settings(), gradu, xi_prev)
xi_d, sig_d, st = ev.update(torch.from_numpy(gradu).cuda(), torch.from_numpy(xi_prev).cuda())
xi_d = xi_d.cpu().numpy(); st = st.cpu().numpy().astype(np.uint32)
err = np.abs(xi_d - xi_o).max(axis=0)
w = int(np.argmax(err))
print("max err", err.max(), "at", w, "iters gpu", st[w] & 0xFFFF, "oracle", it_o[w], "conv", (st[w] >> 16) & 1, cv_o[w])
print("gpu   ", xi_d[:, w]); print("oracle", xi_o[:, w]); print("gradu ", gradu[:, w])
print("iters hist gpu", np.bincount(st & 0xFFFF), "oracle", np.bincount(it_o))
print("n bad", (err > 1e-12).sum(), "of", B)
