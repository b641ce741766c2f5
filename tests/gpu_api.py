"""numpy-in / numpy-out wrappers of the C-ABI entry points of libcmad_hip.so on cuda:0, with the call signatures of
tests/host_harness_lib.py, so that one checker (tests/parity_cases.py) runs on the host build of the kernel arithmetic
(CPU CI) and on the product path (`-m gpu`).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _place_network(desc, info, keep):
    if "nn_packed" in info:
        keep.append(_t(info["nn_packed"]))
        desc.nn_weights = keep[-1].data_ptr()


def hessians(desc, gradu, xi_prev, xi, nx, gradu_prev=None, values=False, info=None):
    """cm_hessians / cm_hessians_rate: d2C (B,nx,nq,nq), d2S (B,6,nq,nq), dC (B,nx,nq), dS (B,6,nq) [, C0 (B,nx), S0 (B,6)]."""
    import torch
    from cmad_amd import _lib
    L = _lib.lib()
    keep = []
    _place_network(desc, info or {}, keep)
    B = gradu.shape[1]
    nq = 2 * nx + _lib.CM_NUM_PARAMS
    z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device="cuda")
    d2C, d2S, dC, dS = z(B, nx, nq, nq), z(B, 6, nq, nq), z(B, nx, nq), z(B, 6, nq)
    g, xp, x = _t(gradu), _t(xi_prev), _t(xi)
    if gradu_prev is not None:
        gp, C0, S0 = _t(gradu_prev), z(B, nx), z(B, 6)
        rc = L.cm_hessians_rate(C.byref(desc), B, _ptr(g), _ptr(gp), _ptr(xp), _ptr(x), _ptr(d2C), _ptr(d2S), _ptr(dC), _ptr(dS),
                                _ptr(C0), _ptr(S0), None)
        _lib.check(rc, "cm_hessians_rate")
        torch.cuda.synchronize()
        out = (d2C, d2S, dC, dS, C0, S0) if values else (d2C, d2S, dC, dS)
    else:
        rc = L.cm_hessians(C.byref(desc), B, _ptr(g), _ptr(xp), _ptr(x), _ptr(d2C), _ptr(d2S), _ptr(dC), _ptr(dS), None)
        _lib.check(rc, "cm_hessians")
        torch.cuda.synchronize()
        out = (d2C, d2S, dC, dS)
    return tuple(o.cpu().numpy() for o in out)


def _evaluate(desc, which, gradu, gradu_prev, xi_prev, xi, nx, info=None):
    import torch
    from cmad_amd import _lib
    L = _lib.lib()
    keep = []
    _place_network(desc, info or {}, keep)
    B, nu = gradu.shape[1], gradu.shape[0]
    ncols = {0: nx, 1: nx, 2: _lib.CM_NUM_PARAMS, 3: nu, 4: nu, 5: 1}[which]
    z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device="cuda")
    Cc, J, s, S = z(nx, B), z(nx * ncols, B), z(6, B), z(6 * ncols, B)
    g, xp, x = _t(gradu), _t(xi_prev), _t(xi)
    if gradu_prev is not None:
        gp = _t(gradu_prev)
        rc = L.cm_evaluate_rate(C.byref(desc), B, int(which), _ptr(g), _ptr(gp), _ptr(xp), _ptr(x), _ptr(Cc), _ptr(J), _ptr(s),
                                _ptr(S), None)
        _lib.check(rc, "cm_evaluate_rate")
    else:
        rc = L.cm_evaluate(C.byref(desc), B, int(which), _ptr(g), _ptr(xp), _ptr(x), _ptr(Cc), _ptr(J), _ptr(s), _ptr(S), None)
        _lib.check(rc, "cm_evaluate")
    torch.cuda.synchronize()
    return Cc.cpu().numpy(), J.cpu().numpy().reshape(nx, ncols, B), s.cpu().numpy(), S.cpu().numpy().reshape(6, ncols, B)


def evaluate(desc, which, gradu, xi_prev, xi, nx, info=None):
    """cm_evaluate: C (nx,B), J (nx,ncols,B), sigma6 (6,B), S (6,ncols,B)."""
    return _evaluate(desc, which, gradu, None, xi_prev, xi, nx, info)


def evaluate_rate(desc, which, gradu, gradu_prev, xi_prev, xi, nx, info=None):
    return _evaluate(desc, which, gradu, gradu_prev, xi_prev, xi, nx, info)


def param_blocks(desc, ep_index, gradu, xi_prev, xi, nx, gradu_prev=None, info=None):
    """cm_param_blocks: dC_dp (n_ep, nx, B), dsigma_dp (n_ep, 6, B) for the extended parameter indices."""
    import torch
    from cmad_amd import _lib
    L = _lib.lib()
    keep = []
    _place_network(desc, info or {}, keep)
    ep = torch.tensor(list(ep_index), dtype=torch.int32, device="cuda")
    B = gradu.shape[1]
    dC = torch.zeros((len(ep_index), nx, B), dtype=torch.float64, device="cuda")
    dS = torch.zeros((len(ep_index), 6, B), dtype=torch.float64, device="cuda")
    g, xp, x = _t(gradu), _t(xi_prev), _t(xi)
    gp = None if gradu_prev is None else _t(gradu_prev)
    rc = L.cm_param_blocks(C.byref(desc), B, len(ep_index), _ptr(ep), _ptr(g), _ptr(gp), _ptr(xp), _ptr(x), _ptr(dC), _ptr(dS), None)
    _lib.check(rc, "cm_param_blocks")
    torch.cuda.synchronize()
    return dC.cpu().numpy(), dS.cpu().numpy()
