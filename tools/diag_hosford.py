import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as ol, parity_cases as pc
sc = pc.Scenario(ol.FULL_3D, "hosford", {"a": 8.}, False, False, B=4096)
gb = pc.GpuBackend()
for gradu, xp, xi_o, it_o in ((sc.gradu0, sc.xi0, sc.xi1, sc.it1), (sc.gradu, sc.xi1, sc.xi2, sc.it2)):
    xi, sig, st = gb.update(sc, gradu, xp)
    it = st & 0xffff; cv = (st >> 16) & 1
    print("gpu iters", np.bincount(it), "oracle", np.bincount(it_o), "nonconv", (cv == 0).sum(), "maxerr", np.abs(xi - xi_o).max())
    bad = np.where(cv == 0)[0]
    for w in bad[:3]:
        print(" point", w, "status", hex(st[w]), "oracle it", it_o[w])
        print("  gpu   ", xi[:, w]); print("  oracle", xi_o[:, w]); print("  gradu ", gradu[:, w]); print("  xprev ", xp[:, w])
        U = gradu[:, w].reshape(3, 3)
        print("  oracle residual at gpu xi", sc.mat.residual(xi[:, w], xp[:, w], U))
