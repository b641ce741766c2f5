"""Independent fp64 torch.func restatement of the reference residual, used ONLY to cross-check the
oracle's dual-number derivatives with a second AD engine (tests, CPU).

Follows /root/reference/cmad/models/small_elastic_plastic.py:33-92,237-321, effective_stress.py:30-52,
167-177, hardening.py:9-13, elastic_stress.py:14-21, var_types.py:43-78, kinematics.py:10-53, paths.py:26-27.
Parameters are passed as a flat tensor in the oracle layout (tests/oracle_lib.py P_* indices, E/nu pair).
"""
import torch
from torch.func import grad, jacfwd

import oracle_lib as ol

torch.set_default_dtype(torch.float64)


def sym_from_vec(v):
    return torch.stack([torch.stack([v[0], v[1], v[2]]), torch.stack([v[1], v[3], v[4]]),
                        torch.stack([v[2], v[4], v[5]])])


def vec_from_sym(A):
    return torch.stack([A[0, 0], A[0, 1], A[0, 2], A[1, 1], A[1, 2], A[2, 2]])


def J2(c, p):
    s = c - torch.trace(c) / 3. * torch.eye(3)
    return (3. / 2.) ** 0.5 * torch.sqrt(torch.sum(s * s))


def hill(c, p):
    F, G, H, L, M, N = [p[ol.P_YC + k] for k in range(6)]
    return torch.sqrt(F * (c[1, 1] - c[2, 2]) ** 2 + G * (c[2, 2] - c[0, 0]) ** 2 + H * (c[0, 0] - c[1, 1]) ** 2
                      + L * (c[2, 1] ** 2 + c[1, 2] ** 2) + M * (c[2, 0] ** 2 + c[0, 2] ** 2)
                      + N * (c[1, 0] ** 2 + c[0, 1] ** 2))


def hosford(c, p):
    vm = J2(c, p)
    a = p[ol.P_YC]
    sc = c / vm
    d01 = torch.abs(sc[0, 0] - sc[1, 1]) ** a
    d12 = torch.abs(sc[1, 1] - sc[2, 2]) ** a
    d20 = torch.abs(sc[2, 2] - sc[0, 0]) ** a
    return vm * (0.5 * (d01 + d12 + d20)) ** (a ** -1)


EFF = {ol.Y_J2: J2, ol.Y_HILL: hill, ol.Y_HOSFORD: hosford}


def gather_F(xi, U, def_type, uidx):
    if def_type == ol.FULL_3D:
        return torch.eye(3) + U.reshape(3, 3)
    if def_type == ol.PLANE_STRESS:
        F = torch.zeros(3, 3)
        F2 = torch.eye(2) + U.reshape(2, 2)
        rows = [torch.stack([F2[0, 0], F2[0, 1], torch.zeros(())]),
                torch.stack([F2[1, 0], F2[1, 1], torch.zeros(())]),
                torch.stack([torch.zeros(()), torch.zeros(()), xi[7]])]
        return torch.stack(rows)
    Fu = 1. + U.reshape(-1)[0]
    d = {0: [Fu, xi[7], xi[8]], 1: [xi[7], Fu, xi[8]], 2: [xi[7], xi[8], Fu]}[uidx]
    return torch.diag(torch.stack(d))


def elastic_strain(xi, p, U, def_type, uidx):
    F = gather_F(xi, U, def_type, uidx)
    P = sym_from_vec(xi[:6])
    gu = F - torch.eye(3)
    eps = 0.5 * (gu + gu.T)
    Q = p[ol.P_Q:ol.P_Q + 9].reshape(3, 3)
    if def_type == ol.UNIAXIAL_STRESS:
        off = Q @ P @ Q.T
        mask = torch.eye(3)
        con = mask * eps + (1 - mask) * off
        m = Q.T @ con @ Q
    else:
        m = Q.T @ eps @ Q
    return m - P


def lame(p):
    E, nu = p[ol.P_EL0], p[ol.P_EL1]
    return E * nu / ((1. + nu) * (1. - 2. * nu)), E / (2. * (1. + nu))


def residual(xi, xp, p, U, desc):
    def_type, yk, uidx = desc.def_type, desc.yield_kind, desc.uniaxial_idx
    lm, mu = lame(p)
    ee = elastic_strain(xi, p, U, def_type, uidx)
    cm = lm * torch.trace(ee) * torch.eye(3) + 2. * mu * ee
    eff = EFF[yk]
    phi = eff(cm, p)
    n = grad(eff)(cm, p)
    alpha, alpha_prev = xi[6], xp[6]
    H = p[ol.P_VOCE_S] * (1. - torch.exp(-p[ol.P_VOCE_D] * alpha)) * desc.has_voce + p[ol.P_LIN_K] * alpha * desc.has_linear
    f = (phi - p[ol.P_Y] - H) / (2. * mu)
    dg = alpha - alpha_prev
    Ce_t = sym_from_vec(xi[:6]) - sym_from_vec(xp[:6])
    Ce = [vec_from_sym(Ce_t), dg.reshape(1)]
    Cp = [vec_from_sym(Ce_t - dg * n), f.reshape(1)]
    if def_type in (ol.PLANE_STRESS, ol.UNIAXIAL_STRESS):
        Q = p[ol.P_Q:ol.P_Q + 9].reshape(3, 3)
        g = Q @ cm @ Q.T
        if def_type == ol.PLANE_STRESS:
            st = (g[2, 2] / (2. * mu)).reshape(1)
        else:
            a, b = [(1, 2), (0, 2), (0, 1)][uidx]
            st = torch.stack([g[a, a], g[b, b]]) / (2. * mu)
        Ce.append(st); Cp.append(st)
    Ce, Cp = torch.cat(Ce), torch.cat(Cp)
    plastic = torch.logical_or(f > desc.yield_tol, torch.abs(f) < desc.yield_tol)
    return torch.where(plastic, Cp, Ce)


def cauchy(xi, p, U, desc):
    lm, mu = lame(p)
    ee = elastic_strain(xi, p, U, desc.def_type, desc.uniaxial_idx)
    cm = lm * torch.trace(ee) * torch.eye(3) + 2. * mu * ee
    Q = p[ol.P_Q:ol.P_Q + 9].reshape(3, 3)
    return (Q @ cm @ Q.T).reshape(9)


def jacobians(xi, xp, p, U, desc):
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    xi, xp, p, U = t(xi), t(xp), t(p), t(U).reshape(-1)
    f = lambda a, b, c, d: residual(a, b, c, d, desc)
    g = lambda a, c, d: cauchy(a, c, d, desc)
    out = {
        "C": f(xi, xp, p, U),
        ol.W_XI: jacfwd(f, 0)(xi, xp, p, U), ol.W_XI_PREV: jacfwd(f, 1)(xi, xp, p, U),
        ol.W_PARAMS: jacfwd(f, 2)(xi, xp, p, U), ol.W_U: jacfwd(f, 3)(xi, xp, p, U),
        "S": g(xi, p, U), ("S", ol.W_XI): jacfwd(g, 0)(xi, p, U), ("S", ol.W_PARAMS): jacfwd(g, 1)(xi, p, U),
        ("S", ol.W_U): jacfwd(g, 2)(xi, p, U),
    }
    return {k: v.detach().numpy() for k, v in out.items()}
