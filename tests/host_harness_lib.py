"""Loader of tests/native/libhost_harness.so: the per-point device math of cmad_amd/csrc/cm_device.hpp
compiled for the host.  TEST INFRASTRUCTURE ONLY (lets CPU CI and sanitizers exercise the hand-derived
formulas the GPU kernels use); the product never loads it."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")
_SO = os.path.join(_DIR, "libhost_harness.so")
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


NPARTS = 7          # host_harness.cpp compiles in independent pieces selected by -DHH_PART=k (see its header comment)


def build(force=False, sanitize=False):
    src = os.path.join(_DIR, "host_harness.cpp")
    csrc = os.path.join(_ROOT, "cmad_amd", "csrc")
    deps = [src, os.path.join(_ROOT, "include", "cmad_hip.h")] + [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hpp")]
    out = _SO if not sanitize else os.path.join(_DIR, "libhost_harness_asan.so")
    is_stale = lambda: (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)
    if not (force or is_stale()):
        return out
    import fcntl
    with open(out + ".lock", "w") as lock:                      # pytest-xdist workers: one builds, the others wait and reuse
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or is_stale():
            flags = ["-O1", "-g", "-std=c++20", "-fPIC", "-ffp-contract=off"]
            if sanitize:
                flags += ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
            tag = ("asan" if sanitize else "host") + f"_{os.getpid()}"
            objs = [os.path.join(_DIR, f"hh_{tag}_part{k}.o") for k in range(NPARTS)]
            procs = [subprocess.Popen(["g++"] + flags + [f"-DHH_PART={k}", "-c", src, "-o", objs[k]],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for k in range(NPARTS)]
            logs = [p.communicate()[0] for p in procs]
            if any(p.returncode != 0 for p in procs):
                raise RuntimeError("g++ failed on tests/native/host_harness.cpp:\n" + "\n".join(logs))
            subprocess.run(["g++"] + flags + ["-shared", "-o", out + ".tmp"] + objs, check=True, capture_output=True)
            os.replace(out + ".tmp", out)
            for o in objs:
                os.remove(o)
    return out


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def set_dense(flag):
    """Force the dense 7x7 path also for FULL_3D (the kernels use the structured solve there)."""
    lib().hh_set_dense(int(bool(flag)))


def subspace_fallbacks(reset=True):
    """Points that left the J2 line / plane Newton iterations for the general path since the last reset."""
    L = lib()
    L.hh_subspace_fallbacks.restype = C.c_longlong
    return int(L.hh_subspace_fallbacks(int(bool(reset))))


def set_radial_vjp(flag):
    """J2 / FULL_3D: compute the parameter gradient of `vjp` by cm::reverse_j2_radial, the closed form of the fused J2 kernels,
    instead of the transposed structured solve."""
    lib().hh_set_radial_vjp(int(bool(flag)))


def set_passes(flag):
    """Solve by cm::newton_pass (cm_pool.hpp: the resumable, one-evaluation-per-pass form of the same iteration that the
    work-pool kernels run) instead of cm::newton / cm::newton_s."""
    lib().hh_set_passes(int(bool(flag)))


def set_force_ls(flag):
    """Run the line-search (LS = true) instantiations of the Newton loops also when ls_max_evals == 0: what the library does
    for the configurations it builds once (Hosford and the dense surfaces, the rate form, the HNN build; cmad_hip.hip dispatch)."""
    lib().hh_set_force_ls(int(bool(flag)))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def update(desc, gradu, xi_prev, nx, tangent=False):
    L = lib()
    from cmad_amd import _lib as cl
    assert L.hh_sizeof_desc() == C.sizeof(cl.ModelDesc)
    gradu = np.ascontiguousarray(gradu, dtype=np.float64); xi_prev = np.ascontiguousarray(xi_prev, dtype=np.float64)
    B = gradu.shape[1]
    xi = np.zeros((nx, B)); sig = np.zeros((6, B)); st = np.zeros(B, dtype=np.uint32)
    ds = np.zeros((6 * gradu.shape[0], B)) if tangent else None
    rc = L.hh_update(C.byref(desc), C.c_int64(B), _p(gradu), _p(xi_prev), _p(xi), _p(sig), _p(st), _p(ds))
    assert rc == 0
    return (xi, sig, st, ds.reshape(6, gradu.shape[0], B)) if tangent else (xi, sig, st)


def vjp(desc, gradu, xi_prev, xi, sbar, xin=None):
    L = lib()
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    gradu, xi_prev, xi, sbar, xin = c(gradu), c(xi_prev), c(xi), c(sbar), c(xin)
    B = gradu.shape[1]
    g = np.zeros(12); xb = np.zeros_like(xi); ub = np.zeros_like(gradu)
    rc = L.hh_vjp(C.byref(desc), C.c_int64(B), _p(gradu), _p(xi_prev), _p(xi), _p(sbar), _p(xin), _p(g), _p(xb), _p(ub))
    assert rc == 0
    return g, xb, ub


def vjp_rate(desc, gradu, gradu_prev, xi_prev, xi, sbar, xin=None):
    L = lib()
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    gradu, gradu_prev, xi_prev, xi, sbar, xin = c(gradu), c(gradu_prev), c(xi_prev), c(xi), c(sbar), c(xin)
    B = gradu.shape[1]
    g = np.zeros(12); xb = np.zeros_like(xi); ub = np.zeros_like(gradu)
    rc = L.hh_vjp_rate(C.byref(desc), C.c_int64(B), _p(gradu), _p(gradu_prev), _p(xi_prev), _p(xi), _p(sbar), _p(xin),
                       _p(g), _p(xb), _p(ub))
    assert rc == 0
    return g, xb, ub


def history(desc, gradu_hist, data_hist, wsq6, xi0):
    """cm::history_point over the batch: (out[13] = {J, grad KP}, xi_hist (K+1, nx, B))."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu_hist, data_hist, wsq6, xi0 = c(gradu_hist), c(data_hist), c(wsq6), c(xi0)
    K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
    xi_hist = np.zeros((K + 1,) + xi0.shape); out = np.zeros(13)
    rc = L.hh_history(C.byref(desc), C.c_int64(B), C.c_int(K), _p(gradu_hist), _p(data_hist), _p(wsq6), _p(xi0), _p(xi_hist), _p(out))
    assert rc == 0
    return out, xi_hist


def adjoint_history(desc, gradu_hist, sbar_hist, xi0, xibar_hist=None, want_lam=False):
    """cm_adjoint_history on the host build: (grad KP (12,), xi_hist (K+1, nx, B), lam_hist (K+1, nx, B) or None)."""
    L = lib()
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    gradu_hist, sbar_hist, xibar_hist, xi0 = c(gradu_hist), c(sbar_hist), c(xibar_hist), c(xi0)
    K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
    xi_hist = np.zeros((K + 1,) + xi0.shape); grad = np.zeros(12)
    lam = np.zeros((K + 1,) + xi0.shape) if want_lam else None
    rc = L.hh_adjoint_history(C.byref(desc), C.c_int64(B), C.c_int(K), _p(gradu_hist), _p(sbar_hist), _p(xibar_hist), _p(xi0),
                              _p(xi_hist), _p(lam), _p(grad))
    assert rc == 0
    return grad, xi_hist, lam


def direct_history(desc, gradu_hist, xi_hist, sbar_hist=None, xibar_hist=None, want_blocks=True):
    """cm_direct_history on the host build: (grad KP (12,), dxi_dp_hist (K+1, nx, 12, B), dsigma_dp_hist (K+1, 6, 12, B))."""
    L = lib()
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    gradu_hist, xi_hist, sbar_hist, xibar_hist = c(gradu_hist), c(xi_hist), c(sbar_hist), c(xibar_hist)
    K, nx, B = xi_hist.shape[0] - 1, xi_hist.shape[1], xi_hist.shape[2]
    dx = np.zeros((K + 1, nx, 12, B)) if want_blocks else None
    ds = np.zeros((K + 1, 6, 12, B)) if want_blocks else None
    grad = np.zeros(12)
    rc = L.hh_direct_history(C.byref(desc), C.c_int64(B), C.c_int(K), _p(gradu_hist), _p(xi_hist), _p(sbar_hist), _p(xibar_hist),
                             _p(dx), _p(ds), _p(grad))
    assert rc == 0
    return grad, dx, ds


def hessian_history(desc, gradu_hist, xi_hist, lam_hist, dx_dp_hist, sbar_hist, hss, hxx=None):
    """cm_hessian_history on the host build: stage 1 (W per point and step) by cm::hessian_weight, the quadratic form
    sum D^T W D in numpy.  Returns hess (12, 12), KP order."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu_hist, xi_hist, lam_hist, dx_dp_hist, sbar_hist, hss = map(c, (gradu_hist, xi_hist, lam_hist, dx_dp_hist, sbar_hist, hss))
    hxx = None if hxx is None else c(hxx)
    K, nx, B = xi_hist.shape[0] - 1, xi_hist.shape[1], xi_hist.shape[2]
    nq = 2 * nx + 12
    W = np.zeros((K, B, nq, nq))
    hss6, hss_hist = (hss, None) if hss.ndim == 1 else (None, hss)            # (6,) constant or (K+1, 6) per step
    rc = L.hh_hessian_weights(C.byref(desc), C.c_int64(B), C.c_int(K), _p(gradu_hist), _p(xi_hist), _p(lam_hist), _p(sbar_hist),
                              _p(hss6), _p(hss_hist), _p(hxx), _p(W))
    assert rc == 0
    H = np.zeros((12, 12))
    for k in range(1, K + 1):
        for b in range(B):
            D = np.vstack([dx_dp_hist[k, :, :, b], dx_dp_hist[k - 1, :, :, b], np.eye(12)])
            H += D.T @ W[k - 1, b] @ D
    return H


def direct_history_ep(desc, ep_index, gradu_hist, xi_hist):
    """cm_direct_history_ep on the host build: dxi_dpe_hist (K+1, nx, n_ep, B)."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu_hist, xi_hist = c(gradu_hist), c(xi_hist)
    ep = np.ascontiguousarray(ep_index, dtype=np.int32)
    K, nx, B = xi_hist.shape[0] - 1, xi_hist.shape[1], xi_hist.shape[2]
    dxe = np.zeros((K + 1, nx, len(ep), B))
    rc = L.hh_direct_history_ep(C.byref(desc), C.c_int64(B), C.c_int(K), C.c_int(len(ep)), ep.ctypes.data_as(C.c_void_p),
                                _p(gradu_hist), _p(xi_hist), _p(dxe))
    assert rc == 0
    return dxe


def hessian_history_ep(desc, ep_index, gradu_hist, xi_hist, lam_hist, dx_dp_hist, dxe_hist, sbar_hist, hss, hxx=None):
    """cm_hessian_history_ep on the host build: stage 1 (W per point and step, NQ = 2 nx + 12 + n_ep) by cm::hessian_weight,
    the quadratic form sum D^T W D in numpy.  Returns hess (12 + n_ep, 12 + n_ep), native parameters first."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu_hist, xi_hist, lam_hist, dx_dp_hist, dxe_hist, sbar_hist, hss = map(c, (gradu_hist, xi_hist, lam_hist, dx_dp_hist, dxe_hist, sbar_hist, hss))
    hxx = None if hxx is None else c(hxx)
    ep = np.ascontiguousarray(ep_index, dtype=np.int32)
    K, nx, B = xi_hist.shape[0] - 1, xi_hist.shape[1], xi_hist.shape[2]
    npt = 12 + len(ep)
    nq = 2 * nx + npt
    W = np.zeros((K, B, nq, nq))
    hss6, hss_hist = (hss, None) if hss.ndim == 1 else (None, hss)
    rc = L.hh_hessian_weights_ep(C.byref(desc), C.c_int64(B), C.c_int(K), C.c_int(len(ep)), ep.ctypes.data_as(C.c_void_p), _p(gradu_hist),
                                 _p(xi_hist), _p(lam_hist), _p(sbar_hist), _p(hss6), _p(hss_hist), _p(hxx), _p(W))
    assert rc == 0
    H = np.zeros((npt, npt))
    for k in range(1, K + 1):
        for b in range(B):
            D = np.vstack([np.hstack([dx_dp_hist[k, :, :, b], dxe_hist[k, :, :, b]]),
                           np.hstack([dx_dp_hist[k - 1, :, :, b], dxe_hist[k - 1, :, :, b]]), np.eye(npt)])
            H += D.T @ W[k - 1, b] @ D
    return H


def param_blocks(desc, ep_index, gradu, xi_prev, xi, nx, gradu_prev=None, info=None):
    """cm_param_blocks on the host build: dC_dp (n_ep, nx, B), dsigma_dp (n_ep, 6, B) for the extended parameter indices."""
    L = lib()
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    gradu, xi_prev, xi, gradu_prev = c(gradu), c(xi_prev), c(xi), c(gradu_prev)
    ep = np.ascontiguousarray(ep_index, dtype=np.int32)
    B = gradu.shape[1]
    dC = np.zeros((len(ep), nx, B)); dS = np.zeros((len(ep), 6, B))
    rc = L.hh_param_blocks(C.byref(desc), C.c_int64(B), C.c_int(len(ep)), ep.ctypes.data_as(C.c_void_p), _p(gradu), _p(gradu_prev),
                           _p(xi_prev), _p(xi), _p(dC), _p(dS))
    assert rc == 0
    return dC, dS


def update_complex(desc, p_imag, gradu, xi_prev, xi_start, gradu_prev=None, ext_imag=None):
    """cm_update_complex on the host build (cm::newton_cx): complex arrays as (2, rows, B).  Returns (xi, residual, sigma, status)."""
    L = lib()
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    gradu, xi_prev, gradu_prev = c(gradu), c(xi_prev), c(gradu_prev)
    xi = np.array(xi_start, dtype=np.float64, order="C")
    B, nx = gradu.shape[1], xi.shape[1]
    res = np.zeros((2, nx, B)); sig = np.zeros((2, 6, B)); st = np.zeros(B, dtype=np.uint32)
    pim = np.ascontiguousarray(p_imag, dtype=np.float64)
    ext = None if ext_imag is None else np.ascontiguousarray(ext_imag, dtype=np.float64)
    rc = L.hh_update_complex(C.byref(desc), C.c_int64(B), _p(pim), _p(ext), _p(gradu), _p(gradu_prev), _p(xi_prev), _p(xi), _p(res),
                             _p(sig), st.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return xi, res, sig, st


def direct_step(desc, gradu, xi_prev, xi, dxp_dp=None, gradu_prev=None):
    """cm::direct_point over the batch: dxi_dp (nx, 12, B), dsigma_dp (6, 12, B)."""
    L = lib()
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    gradu, xi_prev, xi, dxp_dp, gradu_prev = c(gradu), c(xi_prev), c(xi), c(dxp_dp), c(gradu_prev)
    B, nx = gradu.shape[1], xi.shape[0]
    dx = np.zeros((nx, 12, B)); ds = np.zeros((6, 12, B))
    rc = L.hh_direct_step(C.byref(desc), C.c_int64(B), _p(gradu), _p(gradu_prev), _p(xi_prev), _p(xi), _p(dxp_dp), _p(dx), _p(ds))
    assert rc == 0
    return dx, ds


def primal_history(desc, gradu_hist, xi0):
    """cm::primal_history_point over the batch: xi_hist (K+1, nx, B), sigma_hist (K+1, 6, B), status_hist (K+1, B)."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu_hist, xi0 = c(gradu_hist), c(xi0)
    K, B = gradu_hist.shape[0] - 1, gradu_hist.shape[2]
    xi_hist = np.zeros((K + 1,) + xi0.shape); sig = np.zeros((K + 1, 6, B)); st = np.zeros((K + 1, B), dtype=np.uint32)
    rc = L.hh_primal_history(C.byref(desc), C.c_int64(B), C.c_int(K), _p(gradu_hist), _p(xi0), _p(xi_hist), _p(sig), _p(st))
    assert rc == 0
    return xi_hist, sig, st


def evaluate(desc, which, gradu, xi_prev, xi, nx):
    """Explicit blocks at given states: C (nx,B), J (nx,ncols,B), sigma6 (6,B), S (6,ncols,B)."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu, xi_prev, xi = c(gradu), c(xi_prev), c(xi)
    B, nu = gradu.shape[1], gradu.shape[0]
    ncols = {0: nx, 1: nx, 2: 12, 3: nu, 5: 1}[which]
    Cc = np.zeros((nx, B)); J = np.zeros((nx * ncols, B)); s = np.zeros((6, B)); S = np.zeros((6 * ncols, B))
    rc = L.hh_evaluate(C.byref(desc), C.c_int64(B), C.c_int(which), _p(gradu), _p(xi_prev), _p(xi), _p(Cc), _p(J), _p(s), _p(S))
    assert rc == 0
    return Cc, J.reshape(nx, ncols, B), s, S.reshape(6, ncols, B)


def update_rate(desc, gradu, gradu_prev, xi_prev, nx):
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu, gradu_prev, xi_prev = c(gradu), c(gradu_prev), c(xi_prev)
    B = gradu.shape[1]
    xi = np.zeros((nx, B)); sig = np.zeros((6, B)); st = np.zeros(B, dtype=np.uint32)
    rc = L.hh_update_rate(C.byref(desc), C.c_int64(B), _p(gradu), _p(gradu_prev), _p(xi_prev), _p(xi), _p(sig), _p(st))
    assert rc == 0
    return xi, sig, st


def tangent_rate(desc, gradu, gradu_prev, xi_prev, xi):
    """d sigma / d gradu (6 * n_gradu, B) of the rate form at converged states."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu, gradu_prev, xi_prev, xi = c(gradu), c(gradu_prev), c(xi_prev), c(xi)
    B, nu = gradu.shape[1], gradu.shape[0]
    ds = np.zeros((6 * nu, B))
    rc = L.hh_tangent_rate(C.byref(desc), C.c_int64(B), _p(gradu), _p(gradu_prev), _p(xi_prev), _p(xi), _p(ds))
    assert rc == 0
    return ds


def evaluate_rate(desc, which, gradu, gradu_prev, xi_prev, xi, nx):
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu, gradu_prev, xi_prev, xi = c(gradu), c(gradu_prev), c(xi_prev), c(xi)
    B, nu = gradu.shape[1], gradu.shape[0]
    ncols = {0: nx, 1: nx, 2: 12, 3: nu, 4: nu, 5: 1}[which]
    Cc = np.zeros((nx, B)); J = np.zeros((nx * ncols, B)); s = np.zeros((6, B)); S = np.zeros((6 * ncols, B))
    rc = L.hh_evaluate_rate(C.byref(desc), C.c_int64(B), C.c_int(which), _p(gradu), _p(gradu_prev), _p(xi_prev), _p(xi),
                            _p(Cc), _p(J), _p(s), _p(S))
    assert rc == 0
    return Cc, J.reshape(nx, ncols, B), s, S.reshape(6, ncols, B)


def hessians(desc, gradu, xi_prev, xi, nx, gradu_prev=None, values=False, info=None):
    """d2C (B,nx,nq,nq), d2S (B,6,nq,nq), dC (B,nx,nq), dS (B,6,nq); q = [xi, xi_prev, p(KP)].
    gradu_prev: rate-form model (desc.model_kind = 1)."""
    L = lib()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    gradu, xi_prev, xi = c(gradu), c(xi_prev), c(xi)
    B = gradu.shape[1]
    nq = 2 * nx + 12
    d2C = np.zeros((B, nx, nq, nq)); d2S = np.zeros((B, 6, nq, nq)); dC = np.zeros((B, nx, nq)); dS = np.zeros((B, 6, nq))
    if gradu_prev is not None:
        gradu_prev = c(gradu_prev)
        C0 = np.zeros((B, nx)); S0 = np.zeros((B, 6))
        rc = L.hh_hessians_rate(C.byref(desc), C.c_int64(B), _p(gradu), _p(gradu_prev), _p(xi_prev), _p(xi),
                                _p(d2C), _p(d2S), _p(dC), _p(dS), _p(C0), _p(S0))
        assert rc == 0
        if values:
            return d2C, d2S, dC, dS, C0, S0
    else:
        rc = L.hh_hessians(C.byref(desc), C.c_int64(B), _p(gradu), _p(xi_prev), _p(xi), _p(d2C), _p(d2S), _p(dC), _p(dS))
    assert rc == 0
    return d2C, d2S, dC, dS
