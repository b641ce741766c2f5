import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as ol, parity_cases as pc, host_harness_lib as hh
from cmad_amd import _lib
from cmad_amd.models.device import HybridHillEffectiveStress, build_desc, DeviceEvaluator, _ptr
from cmad_amd.synthetic import gauss_point_batch
icnn, values = pc.al7079_hybrid_setup()
desc, info = build_desc(values, hybrid=HybridHillEffectiveStress(icnn))
B = 4
g = gauss_point_batch(B, eps_y=525.0/70.2e3, dev_scale=5.0)
xp = np.zeros((7, B)); x = xp.copy(); x[6] = 1e-4
desc.nn_weights = info["nn_packed"].ctypes.data
Ch, Jh, sh, Sh = hh.evaluate(desc, 0, g, xp, x, 7)
ev = DeviceEvaluator(desc, info)
print("nn ptr", hex(ev.desc.nn_weights), "widths", list(ev.desc.nn_widths), ev.desc.nn_nlayers, ev.desc.yield_kind)
L = _lib.lib()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Cd = torch.zeros((7, B), dtype=torch.float64, device="cuda"); Jd = torch.zeros((49, B), dtype=torch.float64, device="cuda")
sd = torch.zeros((6, B), dtype=torch.float64, device="cuda"); Sd = torch.zeros((42, B), dtype=torch.float64, device="cuda")
rc = L.cm_evaluate(C.byref(ev.desc), B, 0, _ptr(t(g)), _ptr(t(xp)), _ptr(t(x)), _ptr(Cd), _ptr(Jd), _ptr(sd), _ptr(Sd), None)
torch.cuda.synchronize()
print("rc", rc)
print("host C[:,0]", Ch[:, 0]); print("gpu  C[:,0]", Cd.cpu().numpy()[:, 0])
print("max diff J", np.abs(Jd.cpu().numpy().reshape(7, 7, B) - Jh).max())
