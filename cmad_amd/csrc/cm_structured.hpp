// Structured linear algebra for the FULL_3D / PLANE_STRESS stress update with a pressure-independent yield surface
// (J2, Hill, Hosford); PLANE_STRESS borders the same 7x7 block with the F33 column and the sigma_33 row (solve_s).
//
// Same Newton iteration, same residual, same Jacobian as cm_device.hpp -- only the 7x7 solve is done by
// block elimination instead of a dense LU, using what the Jacobian looks like for J2 / Hill / Hosford:
//
//   Ht = blk(S, t) - rho gt gt^T      S: 3x3 on the normal slots {xx,yy,zz}, t: diagonal on {xy,xz,yz}
//        J2 / Hill: phi = sqrt(s^T A s): S = A3/phi, t = (a11,a22,a44)/phi, rho = 1/phi ; Hosford: t = rho = 0
//   Ht d = 0, gt . d = 0              (pressure independence; d = diagonal indicator) -> every lambda term of
//                                     Cel = 2 mu I + lambda d d^T drops out of the Jacobian
//   A = dC/dx = [ B - eta n gt^T    -n  ]     B   = I + beta W^-1 blk(S, t), beta = 2 mu dgam  (block diagonal,
//               [     -gt^T        j66 ]     eta = beta rho, n = W^-1 gt, j66 = -H'(alpha)/2mu   3x3 + 3 scalars)
//
//   A x = (bv, ba):  p = B^-1 bv, q = B^-1 n, k = 1 + j66 eta,
//                    tau = (ba + k gt.p) / (j66 - k gt.q), s = gt.p + gt.q tau, xa = tau - eta s, xv = p + q tau
//   A^T x = (bv, ba): the same with the roles of n and gt exchanged (B is symmetric).
//
// ~1/3 of the flops of the dense path (no 6x6 Hessian, no 7x7 LU); iterates agree with it to round-off, which
// tests/test_host_math.py checks by running both against the oracle.
//
// The surfaces with a dense 6x6 Hessian (hybrid Hill + network, its beta-rescaled form, Barlat Yld2004) are pressure
// independent as well (they see the stress through its deviator / through L' s, L'' s with zero row sums), so the same
// elimination holds with  B = M = I + beta W^-1 Ht  (dense 6x6, rho = 0): one 6x6 LU and a scalar Schur complement instead of
// assembling and factoring the 7x7 (8x8) system.  M^T = W M W^-1 (W diagonal, Ht symmetric), so transposed solves reuse the
// same factors.
#pragma once
#include "cm_device.hpp"
#include "cm_rate_uniaxial.hpp"

namespace cm {

// For J2 / Hill the blocks are (constant coefficients) x rho, so they are rebuilt from rho where needed instead
// of being carried across the Newton loop (9 doubles = 18 VGPRs less live state); Hosford stores its 3x3 block.
template <int YK>
struct YieldS {
    static constexpr bool QUAD = (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL);
    static constexpr bool DENSE = is_dense_yield(YK);
    double phi, rho;
    double gt[6];
    double Sst[QUAD ? 1 : (DENSE ? 21 : 6)];   // Hosford: S00,S03,S05,S33,S35,S55; dense surfaces: upper triangle of Ht, row-major
};


// S (3x3 on slots 0,3,5: S00,S03,S05,S33,S35,S55) and t (shear slots 1,2,4)
template <int YK>
CM_D void yield_blocks(const cm_model_desc& m, const YieldS<YK>& y, double S[6], double t[3]) {
    if constexpr (YieldS<YK>::QUAD) {
        const QuadForm q = quad_form<YK>(m);
        S[0] = q.a00 * y.rho; S[1] = q.a03 * y.rho; S[2] = q.a05 * y.rho;
        S[3] = q.a33 * y.rho; S[4] = q.a35 * y.rho; S[5] = q.a55 * y.rho;
        t[0] = q.a11 * y.rho; t[1] = q.a22 * y.rho; t[2] = q.a44 * y.rho;
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) S[k] = y.Sst[k];
        t[0] = t[1] = t[2] = 0.0;
    }
}

template <int YK>
CM_D void yield_eval_s(const cm_model_desc& m, const double s[6], YieldS<YK>& y) {
    if constexpr (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL) {
        const QuadForm q = quad_form<YK>(m);
        double As[6];
        As[0] = q.a00 * s[0] + q.a03 * s[3] + q.a05 * s[5];
        As[3] = q.a03 * s[0] + q.a33 * s[3] + q.a35 * s[5];
        As[5] = q.a05 * s[0] + q.a35 * s[3] + q.a55 * s[5];
        As[1] = q.a11 * s[1]; As[2] = q.a22 * s[2]; As[4] = q.a44 * s[4];
        double qq = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) qq += s[k] * As[k];
        y.phi = sqrt(qq);
        // at zero stress the reference's normal is NaN and masked by the branch select (phi = 0 is always
        // elastic); here the normal is defined as 0 there so that arithmetic masking can replace the selects
        const double ip = (qq > 0.0) ? rcp(y.phi) : 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) y.gt[k] = As[k] * ip;
        y.rho = ip;
    } else if constexpr (YieldS<YK>::DENSE) {
        yield_eval_p<YK, true>(m, s, y.phi, y.gt, y.Sst);      // built packed, kept packed
        y.rho = 0.0;
    } else {
        double Hp[21];
        yield_eval_p<YK, true>(m, s, y.phi, y.gt, Hp);         // Hosford: the Hessian lives on the normal block only
        y.rho = 0.0;
        y.Sst[0] = Hp[sym6(0, 0)]; y.Sst[1] = Hp[sym6(0, 3)]; y.Sst[2] = Hp[sym6(0, 5)];
        y.Sst[3] = Hp[sym6(3, 3)]; y.Sst[4] = Hp[sym6(3, 5)]; y.Sst[5] = Hp[sym6(5, 5)];
    }
}

// blk(S, t) u
template <int YK>
CM_D void blk_apply(const cm_model_desc& m, const YieldS<YK>& y, const double u[6], double out[6]) {
    if constexpr (YieldS<YK>::DENSE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < 6; ++l) a += y.Sst[sym6(k, l)] * u[l];
            out[k] = a;
        }
        return;
    }
    double S[6], t[3];
    yield_blocks<YK>(m, y, S, t);
    out[0] = S[0] * u[0] + S[1] * u[3] + S[2] * u[5];
    out[3] = S[1] * u[0] + S[3] * u[3] + S[4] * u[5];
    out[5] = S[2] * u[0] + S[4] * u[3] + S[5] * u[5];
    out[1] = t[0] * u[1]; out[2] = t[1] * u[2]; out[4] = t[2] * u[4];
}
// Ht u = blk u - rho gt (gt . u)
template <int YK>
CM_D void hess_apply(const cm_model_desc& m, const YieldS<YK>& y, const double u[6], double out[6]) {
    blk_apply<YK>(m, y, u, out);
    const double c = y.rho * dot<6>(y.gt, u);
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] -= c * y.gt[k];
}

// state evaluation, FULL_3D
template <int YK>
struct EvalS {
    double e[6], s[6], tr, f, dgam;
    bool plastic;
    Hard hd;
    YieldS<YK> y;
};

// DEF = PLANE_STRESS: x[7] = F33 enters the strain through z = V(q3 q3^T) and adds the row C[7] = sigma_33 / 2mu
// (cm::strain_stress, cm::residual); `z` is not read for FULL_3D.
// KNOWN_HD: ev.hd already holds hardening(x[6]) (the caller evaluated it at this very alpha) -- skips the exp.
template <int YK, int DEF = CM_FULL_3D, bool KNOWN_HD = false>
CM_D void residual_s(const cm_model_desc& m, const double eg[6], const double* z, const double* x, const double* xp,
                     EvalS<YK>& ev, double* C) {
    static_assert(DEF == CM_FULL_3D || DEF == CM_PLANE_STRESS, "structured path: FULL_3D and PLANE_STRESS");
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        ev.e[k] = eg[k] - x[k];
        if constexpr (DEF == CM_PLANE_STRESS) ev.e[k] += (x[7] - 1.0) * z[k];
    }
    ev.tr = ev.e[0] + ev.e[3] + ev.e[5];
    const double twomu = 2.0 * m.mu, lt = m.lambda * ev.tr;
#pragma unroll
    for (int k = 0; k < 6; ++k) ev.s[k] = twomu * ev.e[k] + (kDiag[k] ? lt : 0.0);
    yield_eval_s<YK>(m, ev.s, ev.y);
    if constexpr (!KNOWN_HD) ev.hd = hardening(m, x[6]);
    const double i2mu = half_over_mu(m);
    ev.f = (ev.y.phi - (m.Y + ev.hd.H)) * i2mu;
    ev.dgam = x[6] - xp[6];
    ev.plastic = (ev.f > m.yield_tol) || (fabs(ev.f) < m.yield_tol);
    if constexpr (YieldS<YK>::QUAD) {
        const double dgp = ev.plastic ? ev.dgam : 0.0;           // one select instead of six (gt is finite, see above)
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = (x[k] - xp[k]) - dgp * kIW[k] * ev.y.gt[k];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double ce = x[k] - xp[k];
            C[k] = ev.plastic ? (ce - ev.dgam * ev.y.gt[k] * kIW[k]) : ce;
        }
    }
    C[6] = ev.plastic ? ev.f : ev.dgam;
    if constexpr (DEF == CM_PLANE_STRESS) {
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) r += kW[k] * z[k] * ev.s[k];
        C[7] = r * i2mu;
    }
}
template <int YK>
CM_D void residual_s(const cm_model_desc& m, const double eg[6], const double* x, const double* xp, EvalS<YK>& ev, double* C) {
    residual_s<YK, CM_FULL_3D>(m, eg, nullptr, x, xp, ev, C);
}

// A = dC/dx in structured form at an evaluated state
template <bool DENSE>
struct PlasticOpT {
    double inv3[6];   // (I + beta S)^-1, symmetric: 00,03,05,33,35,55
    double ib[3];     // 1 / (1 + beta t_j / 2)
    double eta, j66, k, beta;
    bool plastic, ok;
    double M[DENSE ? 6 : 1][DENSE ? 6 : 1];   // dense surfaces: LU factors of I + beta W^-1 Ht (cm::lu_factor layout)
};
using PlasticOp = PlasticOpT<false>;
template <int YK> using PlasticOpFor = PlasticOpT<is_dense_yield(YK)>;

// J2: A3 = 3/2 I - 1/2 1 1^T and a11 = a22 = a44 = 3, so B = a I - c (1 1^T on the normal block) with
// a = 1 + 3/2 beta rho, c = 1/2 beta rho, and B^-1 v = (v + c (v0+v3+v5) d) / a   (Sherman-Morrison, a - 3c = 1).
// Stored as inv3[0] = 1/a, inv3[1] = c.
template <int YK>
CM_D void op_build(const cm_model_desc& m, const EvalS<YK>& ev, PlasticOpFor<YK>& op) {
    const YieldS<YK>& y = ev.y;
    op.plastic = ev.plastic;
    op.beta = 2.0 * m.mu * ev.dgam;
    const double b = op.beta;
    if constexpr (YieldS<YK>::DENSE) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int l = 0; l < 6; ++l) op.M[k][l] = ((k == l) ? 1.0 : 0.0) + b * kIW[k] * y.Sst[sym6(k, l)];
        op.ok = lu_factor<6>(op.M);
        op.eta = 0.0;
        op.j66 = -ev.hd.dH * half_over_mu(m);
        op.k = 1.0;
        return;
    }
    if constexpr (YK == CM_YIELD_J2) {
        const double br = b * y.rho;
        const double a = 1.0 + 1.5 * br;
        op.ok = fabs(a) > 1e-300;
        op.inv3[0] = rcp(a); op.inv3[1] = 0.5 * br;
        op.eta = br;
        op.j66 = -ev.hd.dH * half_over_mu(m);
        op.k = 1.0 + op.j66 * op.eta;
        return;
    }
    double S[6], t[3];
    yield_blocks<YK>(m, y, S, t);
    const double B00 = 1.0 + b * S[0], B03 = b * S[1], B05 = b * S[2];
    const double B33 = 1.0 + b * S[3], B35 = b * S[4], B55 = 1.0 + b * S[5];
    const double c00 = B33 * B55 - B35 * B35, c03 = B05 * B35 - B03 * B55, c05 = B03 * B35 - B05 * B33;
    const double c33 = B00 * B55 - B05 * B05, c35 = B03 * B05 - B00 * B35, c55 = B00 * B33 - B03 * B03;
    const double det = B00 * c00 + B03 * c03 + B05 * c05;
    const double b1 = 1.0 + 0.5 * b * t[0], b2 = 1.0 + 0.5 * b * t[1], b4 = 1.0 + 0.5 * b * t[2];
    op.ok = (fabs(det) > 1e-300) && (fabs(b1) > 1e-300) && (fabs(b2) > 1e-300) && (fabs(b4) > 1e-300);
    const double id = rcp(det);
    op.inv3[0] = c00 * id; op.inv3[1] = c03 * id; op.inv3[2] = c05 * id;
    op.inv3[3] = c33 * id; op.inv3[4] = c35 * id; op.inv3[5] = c55 * id;
    op.ib[0] = rcp(b1); op.ib[1] = rcp(b2); op.ib[2] = rcp(b4);
    op.eta = b * y.rho;
    op.j66 = -ev.hd.dH * half_over_mu(m);
    op.k = 1.0 + op.j66 * op.eta;
}

// out = B^-1 v (TRANSPOSED: B^-T v; B is symmetric except for the dense surfaces, where B^T = W B W^-1)
template <int YK, bool TRANSPOSED = false>
CM_D void binv(const PlasticOpFor<YK>& op, const double v[6], double out[6]) {
    if constexpr (YieldS<YK>::DENSE) {
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = TRANSPOSED ? v[k] * kIW[k] : v[k];
        lu_subst<6>(op.M, t);
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = TRANSPOSED ? t[k] * kW[k] : t[k];
        return;
    }
    if constexpr (YK == CM_YIELD_J2) {
        const double ia = op.inv3[0], cs = op.inv3[1] * (v[0] + v[3] + v[5]);
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = ia * (v[k] + (kDiag[k] ? cs : 0.0));
        return;
    }
    out[0] = op.inv3[0] * v[0] + op.inv3[1] * v[3] + op.inv3[2] * v[5];
    out[3] = op.inv3[1] * v[0] + op.inv3[3] * v[3] + op.inv3[4] * v[5];
    out[5] = op.inv3[2] * v[0] + op.inv3[4] * v[3] + op.inv3[5] * v[5];
    out[1] = op.ib[0] * v[1]; out[2] = op.ib[1] * v[2]; out[4] = op.ib[2] * v[4];
}

// x = A^-1 b (TRANSPOSED: A^-T b); b and x may alias
template <bool TRANSPOSED, int YK>
CM_D void op_solve(const PlasticOpFor<YK>& op, const YieldS<YK>& y, const double* b, double* x) {
    if (!op.plastic) {
#pragma unroll
        for (int k = 0; k < 7; ++k) x[k] = b[k];
        return;
    }
    if constexpr (YK == CM_YIELD_J2) {
        // B^-1 v = (v + c tr(v) d) / a (op_build), so the two block solves collapse to four scalars:
        //   tb = tr(b_v), tg = tr(gt), gb = row . b_v, gn = row . col = sum gt_k^2 / w_k   (either orientation)
        //   rp = (gb + c tb tg) / a ,  rq = (gn + c tg tg) / a
        //   x_v = (b_v + col tau + c d (tb + tg tau)) / a
        const double ia = op.inv3[0], c = op.inv3[1];
        const double tb = b[0] + b[3] + b[5], tg = y.gt[0] + y.gt[3] + y.gt[5];
        double gb = 0.0, gn = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            gb += (TRANSPOSED ? kIW[k] : 1.0) * y.gt[k] * b[k];
            gn += kIW[k] * y.gt[k] * y.gt[k];
        }
        const double rp = (gb + c * tb * tg) * ia, rq = (gn + c * tg * tg) * ia;
        const double tau = (b[6] + op.k * rp) * rcp(op.j66 - op.k * rq);
        const double cd = c * (tb + tg * tau);
#pragma unroll
        for (int k = 0; k < 6; ++k)
            x[k] = ia * (b[k] + (TRANSPOSED ? 1.0 : kIW[k]) * y.gt[k] * tau + (kDiag[k] ? cd : 0.0));
        x[6] = tau - op.eta * (rp + rq * tau);
        return;
    }
    double n[6], p[6], q[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) n[k] = y.gt[k] * kIW[k];
    const double* col = TRANSPOSED ? y.gt : n;      // the vector multiplying tau
    const double* row = TRANSPOSED ? n : y.gt;      // the vector contracted with x_v
    binv<YK, TRANSPOSED>(op, b, p);
    binv<YK, TRANSPOSED>(op, col, q);
    const double rp = dot<6>(row, p), rq = dot<6>(row, q);
    const double tau = (b[6] + op.k * rp) * rcp(op.j66 - op.k * rq);
    const double s = rp + rq * tau;
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = p[k] + q[k] * tau;
    x[6] = tau - op.eta * s;
}

// ---- PLANE_STRESS: the 8x8 system is the structured 7x7 block bordered by the F33 column and the sigma_33 row ---
//   A8 = [ A7   u ]    u = dC[0:7]/dF33 = [ -beta W^-1 Ht z ; gt . z ]   (plastic; 0 elastic -- Ht d = gt . d = 0
//        [ r^T  d ]                                                        removes the lambda terms of Cel z)
//                      r = dC7/dx[0:7] = -[ w o z + (lambda / 2mu) tr(z) d ; 0 ],   d = (w o z) . z + (lambda / 2mu) tr(z)^2
//   A8 y = b:    p = A7^-1 b[0:7], q = A7^-1 u, tau = (b7 - r.p) / (d - r.q), y = [p - tau q ; tau]
//   A8^T y = b:  p = A7^-T b[0:7], q = A7^-T r, tau = (b7 - u.p) / (d - u.q), y = [p - tau q ; tau]
struct Border {
    double u[7], r[6], d;
};
template <int YK>
CM_D void border_build(const cm_model_desc& m, const PlasticOpFor<YK>& op, const EvalS<YK>& ev, const double z[6], Border& bd) {
    const double l2m = m.lambda * half_over_mu(m), zt = z[0] + z[3] + z[5];
    double dd = l2m * zt * zt;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        bd.r[k] = -(kW[k] * z[k] + (kDiag[k] ? l2m * zt : 0.0));
        dd += kW[k] * z[k] * z[k];
    }
    bd.d = dd;
    double hz[6];
    hess_apply<YK>(m, ev.y, z, hz);
#pragma unroll
    for (int k = 0; k < 6; ++k) bd.u[k] = ev.plastic ? -op.beta * kIW[k] * hz[k] : 0.0;
    bd.u[6] = ev.plastic ? dot<6>(ev.y.gt, z) : 0.0;
}

// y = A^-1 b (TRANSPOSED: A^-T b) for DEF's full system; b and y may alias.  Returns false on a zero pivot.
template <int DEF, bool TRANSPOSED, int YK>
CM_D bool solve_s(const cm_model_desc& m, const PlasticOpFor<YK>& op, const EvalS<YK>& ev, const double* z, const double* b, double* y) {
    if constexpr (DEF == CM_FULL_3D) {
        op_solve<TRANSPOSED>(op, ev.y, b, y);
        return true;
    } else {
        Border bd;
        border_build<YK>(m, op, ev, z, bd);
        double p[7], q[7];
        const double b7 = b[7];
        op_solve<TRANSPOSED>(op, ev.y, b, p);
        double rp, rq;
        if constexpr (!TRANSPOSED) {
            op_solve<false>(op, ev.y, bd.u, q);
            rp = dot<6>(bd.r, p); rq = dot<6>(bd.r, q);
        } else {
            double r7[7];
#pragma unroll
            for (int k = 0; k < 6; ++k) r7[k] = bd.r[k];
            r7[6] = 0.0;
            op_solve<true>(op, ev.y, r7, q);
            rp = dot<7>(bd.u, p); rq = dot<7>(bd.u, q);
        }
        const double den = bd.d - rq;
        const double tau = (b7 - rp) * rcp(den);
#pragma unroll
        for (int k = 0; k < 7; ++k) y[k] = p[k] - tau * q[k];
        y[7] = tau;
        return fabs(den) > 1e-300;
    }
}

// ---- local Newton, structured (same control flow as cm::newton) -----------------------------------------
// `ev` is left holding the evaluation at the returned x (the one that passed the convergence test), so the
// reverse sweep of the fused kernels does not have to redo it.
// LS: the line search is a compile-time variant.  While trial points are evaluated the base iterate and the
// Newton direction are parked in the lane's LDS column (`stage`, 2*NX doubles); they are read back only when a
// trial is rejected.
// On the device the column lives in LDS and is addressed as such (address space 3: ds_read / ds_write with immediate offsets;
// through a generic pointer the same accesses are flat_load / flat_store, which travel the vector-memory path).
#if defined(CM_HOST_BUILD)
typedef double cm_stage_double;
#else
typedef __attribute__((address_space(3))) double cm_stage_double;
#endif
struct LaneStage {
    cm_stage_double* p;   // this lane's first slot
    int stride;           // distance between consecutive slots of one lane (block size on the device, 1 on the host)
    // volatile: without it the compiler forwards the stored values to the loads and keeps them in registers
    CM_D volatile cm_stage_double& at(int k) const { return const_cast<volatile cm_stage_double*>(p)[k * stride]; }
};
// the lane's column of a __shared__ staging array (device) / of a plain array (host)
CM_D LaneStage lane_stage(double* base, int lane_offset, int stride) {
    return LaneStage{(cm_stage_double*)base + lane_offset, stride};
}

// ||C||^2 as the convergence test and the line search's merit see it: the plain sum of squares, or (NORM = RateNorm below) the
// norm of a constant linear image of C -- Newton's iterates do not depend on such a map, its stopping test does.
struct PlainNorm {
    template <int NX> CM_D double sq(const double* C) const { return dot<NX>(C, C); }
};
// x0 (optional): start of the iteration when it is not x_prev (newton_s_rate)
template <int YK, bool LS, int DEF = CM_FULL_3D, class NORM = PlainNorm>
CM_D uint32_t newton_s(const cm_model_desc& m, const double eg[6], const double* xp, double* x, bool lane_valid,
                       EvalS<YK>& ev, LaneStage stage = LaneStage{nullptr, 0}, const double* z = nullptr,
                       const NORM norm = NORM{}, const double* x0 = nullptr, double n0sq_known = -1.0) {
    constexpr int NX = Dims<DEF>::NX;
    double C[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) x[k] = x0 ? x0[k] : xp[k];
    residual_s<YK, DEF>(m, eg, z, x, xp, ev, C);
    // squared-norm form of nonlinear_solver.py:140-150, see cm::newton.  n0sq_known >= 0: ||C(x_prev)||^2 when the iteration starts
    // somewhere else (hosford_warm_start) -- the relative tolerance stays the reference's, measured against the residual at x_prev
    const double n0sq = (n0sq_known >= 0.0) ? n0sq_known : norm.template sq<NX>(C);      // (by value: a pointer would pin it to scratch)
    const double rel2 = m.rel_tol * m.rel_tol * n0sq, abs2 = m.abs_tol * m.abs_tol;
    int it = 0;
    bool running = lane_valid;
    uint32_t flags = 0;
    if constexpr (!LS) {
        for (;;) {
            const double nsq = norm.template sq<NX>(C);
            const bool conv = (nsq < rel2) || (nsq < abs2);
            if (running && conv) { running = false; flags |= CM_STATUS_CONVERGED; }
            if (running && it >= m.max_iters) running = false;
            if (!__any(running)) break;
            if (running) {
                double delta[NX];
                PlasticOpFor<YK> op;
                op_build<YK>(m, ev, op);               // ev is the evaluation at the current x (carried)
                if (!op.ok) flags |= CM_STATUS_SINGULAR;
                if (!solve_s<DEF, false>(m, op, ev, z, C, delta)) flags |= CM_STATUS_SINGULAR;
#pragma unroll
                for (int k = 0; k < NX; ++k) x[k] -= delta[k];
                ++it;
            }
            // evaluated by every lane: a lane that has stopped re-evaluates its unchanged x (same ev, C), which
            // costs nothing in lockstep and spares the register copies a predicated redefinition would need
            residual_s<YK, DEF>(m, eg, z, x, xp, ev, C);
        }
        return flags | (uint32_t)it;
    } else {
        // Newton iterations and line-search trials share ONE residual evaluation per pass of the loop (at the
        // bottom); everything above it is bookkeeping on the evaluation that just arrived.  Trials are evaluated
        // straight into (x, ev, C), so an accepted trial already is the carried state.
        // phase: 0 at an accepted iterate | 1 a trial arrived | 2 the lowest-merit step re-evaluated (commit)
        //        3, 4: every trial was non-finite -> full step with the base residual carried (line_search.py:181-183)
        int phase = 0, n = 0;
        double alpha = 1.0, best_alpha = 1.0, best_phi = INFINITY, cc = 0.0;
        for (;;) {
            if (running && phase != 0) {
                bool commit = (phase == 2);
                if (phase == 1 && m.ls_kind == CM_LS_LEGACY) {          // uniform: newton_solve's backtracking (ls_trial_legacy)
                    const double phi = 0.5 * norm.template sq<NX>(C);
                    const double step = ls_trial_legacy(m, phi, cc, alpha, n);
                    if (step == 0.0) commit = true;
                    else {
#pragma unroll
                        for (int k = 0; k < NX; ++k) x[k] -= step * stage.at(NX + k);
                    }
                } else if (phase == 1) {
                    const double phi = 0.5 * norm.template sq<NX>(C);            // merit; phi(0) = cc / 2, phi'(0) = -cc
                    const bool finite = isfinite(phi);
                    if (finite && phi < best_phi) { best_alpha = alpha; best_phi = phi; }
                    // ls_max_evals == 0 (uniform): plain Newton through this kernel -- the full step is the next iterate whatever
                    // its merit, exactly the LS = false loop above (cmad_hip.hip always_searches<>; measured against an explicit
                    // plain branch on the pool kernels: profiles/r03_ls_plain_fallback_ab.txt)
                    const bool accepted = (m.ls_max_evals <= 0) || (finite && (phi <= 0.5 * cc + alpha * (m.ls_c1 * -cc)));
                    ++n;
                    if (accepted) commit = true;
                    else if (n < m.ls_max_evals) {
                        const double am = quad_min(0.5 * cc, -cc, alpha, phi);
                        alpha = finite ? fmin(fmax(am, m.ls_lo * alpha), m.ls_hi * alpha) : 0.5 * alpha;
#pragma unroll
                        for (int k = 0; k < NX; ++k) x[k] = stage.at(k) - alpha * stage.at(NX + k);
                    } else if (best_phi < INFINITY) {                  // no trial accepted: lowest-merit step tried
#pragma unroll
                        for (int k = 0; k < NX; ++k) x[k] = stage.at(k) - best_alpha * stage.at(NX + k);
                        phase = 2;
                    } else {                                           // base point again, to recover its residual
#pragma unroll
                        for (int k = 0; k < NX; ++k) x[k] = stage.at(k);
                        phase = 3;
                    }
                } else if (phase == 3) {                               // C is the base residual: park it, full step
#pragma unroll
                    for (int k = 0; k < NX; ++k) { x[k] -= stage.at(NX + k); stage.at(k) = C[k]; }
                    phase = 4;
                } else if (phase == 4) {                               // ev is at the full step; carry the base residual
#pragma unroll
                    for (int k = 0; k < NX; ++k) C[k] = stage.at(k);
                    commit = true;
                }
                if (commit) { phase = 0; ++it; }
            }
            if (running && phase == 0) {
                const double nsq = norm.template sq<NX>(C);
                const bool conv = (nsq < rel2) || (nsq < abs2);
                if (conv) { running = false; flags |= CM_STATUS_CONVERGED; }
                else if (it >= m.max_iters) running = false;
                else {
                    double delta[NX];
                    PlasticOpFor<YK> op;
                    op_build<YK>(m, ev, op);
                    if (!op.ok) flags |= CM_STATUS_SINGULAR;
                    if (!solve_s<DEF, false>(m, op, ev, z, C, delta)) flags |= CM_STATUS_SINGULAR;
#pragma unroll
                    for (int k = 0; k < NX; ++k) { stage.at(k) = x[k]; stage.at(NX + k) = delta[k]; x[k] -= delta[k]; }
                    cc = nsq; alpha = 1.0; best_alpha = 1.0; best_phi = INFINITY; n = 0; phase = 1;
                }
            }
            if (!__any(running)) break;
            residual_s<YK, DEF>(m, eg, z, x, xp, ev, C);       // every lane (see above)
        }
        return flags | (uint32_t)it;
    }
}

// ---- Hosford, FULL_3D: analytic warm start of the local Newton ---------------------------------------------------------------
// With a large exponent (the notch deck's a = 100) the surface is a Tresca hexagon with rounded corners, and the reference's
// iteration -- Newton on the 7-dof residual started at x_prev, Armijo search on ||C||^2 / 2 -- spends 10-18 residual evaluations
// on every point whose return ends in a corner zone (a quarter of the points of BASELINE configs[2]; 4.1 evaluations per point
// on average).  The same equations are benign in other variables.  Only the normal entries matter (effective_stress.py:167:
// the yield function sees the diagonal), the flow direction is n = D^T p with p_i = d phi / d d_i = 1/2 sgn(d_i) q_i,
// q_i = (|d_i| / phi)^(a-1), d = (s00 - s11, s11 - s22, s22 - s00), and with m / j the largest / second largest |d_i| of the
// trial state (opposite signs, since sum d = 0; the third obeys |d_3| <= |d_m| / 2, so q_3 <= 2^-(a-1) is dropped) the
// backward-Euler equations read, in w_i = log q_i and dgam,
//     R_m = phi r_m - |d_m|_trial + mu dgam (2 q_m + q_j) = 0        r_i = exp(w_i / (a-1)) = |d_i| / phi ,  q_i = exp(w_i)
//     R_j = phi r_j - |d_j|_trial + mu dgam (2 q_j + q_m) = 0        phi = Y + H(alpha_prev + dgam)   (f = 0 by construction)
//     R_n = q_m r_m + q_j r_j - 2 = 0                                 (the definition of phi: 1/2 sum (|d_i| / phi)^a = 1)
// -- the stiff map d -> q = (d / phi)^(a-1) is only ever evaluated in its benign direction q -> d.  Newton on these three,
// started at the return onto the Tresca hexagon (face or corner, each one scalar equation in dgam), converges in 2-6 steps of
// ~five exponentials.  The result is NOT taken on trust: it is handed to newton_s as its starting point, so what is returned
// passed the reference's own convergence test on the reference's residual (relative tolerance against ||C(x_prev)||, as in
// nonlinear_solver.py:140-150) -- a converged warm start costs one residual evaluation, anything else is finished by the
// reference's Newton steps / line search from there.  Same root, to the Newton tolerance; iteration counts are counted from
// the warm start.  CM_SOLVER_GENERAL_NEWTON runs the reference's iteration from x_prev instead.
constexpr double kHosfordWarmMinA = 20.0;          // below: q_3 is not negligible and the reference iteration is cheap anyway
constexpr int kHosfordWarmMaxIt = 14;
CM_D bool hosford_warm_start(const cm_model_desc& m, const double eg[6], const double* xp, double* x0, double& n0sq, bool lane_valid) {
#pragma unroll
    for (int k = 0; k < 7; ++k) x0[k] = xp[k];
    n0sq = 0.0;
    const double a = m.yc[0];
    if (!(a >= kHosfordWarmMinA)) return false;                                // uniform
    const double twomu = 2.0 * m.mu, i2mu = half_over_mu(m);
    const double e0 = eg[0] - xp[0], e3 = eg[3] - xp[3], e5 = eg[5] - xp[5], lt = m.lambda * (e0 + e3 + e5);
    const double s[6] = {twomu * e0 + lt, 0.0, 0.0, twomu * e3 + lt, 0.0, twomu * e5 + lt};
    double phi_tr, gtr[6], Hdummy[1];
    yield_eval_p<CM_YIELD_HOSFORD, false>(m, s, phi_tr, gtr, Hdummy);          // the value residual_s sees at x_prev
    const double alpha_p = xp[6];
    const Hard hp = hardening(m, alpha_p);
    const double f0 = (phi_tr - (m.Y + hp.H)) * i2mu;
    const bool plastic0 = (f0 > m.yield_tol) || (fabs(f0) < m.yield_tol);
    n0sq = plastic0 ? f0 * f0 : 0.0;                                           // C(x_prev) = [0, f0] resp. 0 (elastic branch)
    // (a state that already passes the reference's test at x_prev is returned as it is, 0 iterations: re-applying a strain to
    // its own result changes nothing, bit for bit)
    bool active = lane_valid && plastic0 && (f0 > 0.0) && !(n0sq < m.abs_tol * m.abs_tol);
    if (!__any(active)) return false;
    const double d0 = s[0] - s[3], d1 = s[3] - s[5], d2 = s[5] - s[0];
    const double t0 = fabs(d0), t1 = fabs(d1), t2 = fabs(d2);
    const bool m0 = (t0 >= t1) && (t0 >= t2), m1 = !m0 && (t1 >= t2), m2 = !m0 && !m1;
    const bool j0 = (m1 && t0 >= t2) || (m2 && t0 >= t1), j1 = (m0 && t1 >= t2) || (m2 && !(t0 >= t1)), j2 = !(m2 || j0 || j1);
    const double dm = m0 ? t0 : (m1 ? t1 : t2), dj = j0 ? t0 : (j1 ? t1 : t2);
    const double ia1 = rcp(a - 1.0), ap = a * ia1, mu = m.mu;
    double gam, wj, wm;
    if (hardening_seed_ok(m)) {
        // The start in float (cm_device.hpp, "single-precision seeds"; exponentials and logarithms are one instruction each
        // there): the return onto the hexagon -- face (one scalar equation: two Newton steps), then, when the second
        // difference overtakes the first on the way, the corner (both faces active: dgam from the sum of the two equations, the
        // split from their difference; flow stress linearised) -- followed by Newton steps on the three equations themselves
        // until they hold to 1e-4; the double-precision loop below then needs one or two steps.
        const float af = (float)a, ia1f = (float)ia1, apf = (float)ap, muf = (float)mu, twomuf = 2.0f * muf;
        const float dmf = (float)dm, djf = (float)dj, Yf = (float)m.Y, alf = (float)alpha_p;
        const float kf = exp_f(-0.6931471805599453f * rcp_f(af)), c = 2.0f * kf;   // 2^(-1/a): phi = kf |d_m| on a face; q_m = c there
        const HardF hq = hardening_f(m, alf);
        float g = -(kf * dmf - (Yf + hq.H)) * rcp_f(-twomuf * c * kf - hq.dH);
        const HardF h1 = hardening_f(m, alf + g);
        const float g1 = g;
        g = fmaxf(g - (kf * (dmf - twomuf * c * g) - (Yf + h1.H)) * rcp_f(-twomuf * c * kf - h1.dH), 0.0f);
        const float phi_f = Yf + h1.H + h1.dH * (g - g1);                       // flow stress at the face solution (linearised)
        const bool corner = muf * c * g > dmf - djf;
        const float gc = fmaxf(g - (dmf + djf - 6.0f * muf * g - 2.0f * phi_f) * rcp_f(-6.0f * muf - 2.0f * h1.dH), 1e-30f);
        const float half_split = 0.5f * (dmf - djf) * rcp_f(muf * gc);          // (q_m - q_j) / 2 at the corner, q_m + q_j = 2
        float gf = corner ? gc : g;
        // q_j: the corner's split, or (face) what the second difference reaches at the face solution, r_j^(a-1).  Between the two
        // regimes -- the second difference ends within a few per cent of phi -- the face value overshoots (it ignores the flow
        // q_j itself causes) and the corner's split is about zero or below: start at 0.05 there (true values 0.01 .. 0.3)
        const float qj_c = fminf(fmaxf(1.0f - half_split, 0.05f), 1.0f);
        const float rj_f = fminf(fmaxf((djf - muf * c * g) * rcp_f(phi_f), 1e-30f), 1.0f);
        const float wj_face = fmaxf(log_f(rj_f) * (af - 1.0f), -80.0f), wj_cap = log_f(qj_c);
        const bool capped = !corner && (wj_face > wj_cap);
        float wjf = (corner || capped) ? wj_cap : wj_face;
        float wmf = log_f(corner ? 2.0f - qj_c : (capped ? c - 0.9f * qj_c : c));
        bool sdone = !(active && gf > 0.0f);
        for (int it = 0; it < 8; ++it) {
            const float rm = exp_f(wmf * ia1f), qm = exp_f(wmf), qj = exp_f(wjf), rj = exp_f(wjf * ia1f);
            const HardF h = hardening_f(m, alf + gf);
            const float phi = Yf + h.H, mg = muf * gf;
            const float Rm = phi * rm - dmf + mg * (2.0f * qm + qj), Rj = phi * rj - djf + mg * (2.0f * qj + qm);
            const float Rn = qm * rm + qj * rj - 2.0f;
            const float res = fmaxf(fmaxf(fabsf(Rm), fabsf(Rj)) * rcp_f(dmf), fabsf(Rn));
            if (!(res >= 1e-4f)) sdone = true;                                  // (also: not finite)
            if (!sdone) {
                const float J00 = phi * rm * ia1f + 2.0f * mg * qm, J01 = mg * qj, J02 = h.dH * rm + muf * (2.0f * qm + qj);
                const float J10 = mg * qm, J11 = phi * rj * ia1f + 2.0f * mg * qj, J12 = h.dH * rj + muf * (2.0f * qj + qm);
                const float J20 = apf * qm * rm, J21 = apf * qj * rj;
                const float c00 = -J12 * J21, c01 = J12 * J20, c02 = J10 * J21 - J11 * J20;
                const float idet = rcp_f(J00 * c00 + J01 * c01 + J02 * c02);
                const float dwm = (Rm * c00 + J01 * (J12 * Rn) + J02 * (Rj * J21 - J11 * Rn)) * idet;
                const float dwj = (J00 * (-J12 * Rn) + Rm * c01 + J02 * (J10 * Rn - Rj * J20)) * idet;
                const float dgm = (J00 * (J11 * Rn - Rj * J21) + J01 * (Rj * J20 - J10 * Rn) + Rm * c02) * idet;
                wmf = fminf(fmaxf(wmf - dwm, -3.0f), 0.75f);
                gf = fmaxf(gf - dgm, 1e-3f * gf);
                // a large step that LOWERS q_j is taken in the variable q_j (see the loop below)
                wjf = fmaxf(wjf + ((dwj > 0.25f) ? log_f(fmaxf(1.0f - dwj, 0.05f)) : fminf(-dwj, 2.0f)), -80.0f);
            }
            if (!__any(!sdone)) break;
        }
        const bool fin = (gf > 0.0f) && (gf < 1e30f) && (wmf > -4.0f) && (wjf > -90.0f);   // finite (a NaN fails every comparison)
        gam = fin ? (double)gf : 0.0;                                           // (0: the lane leaves the warm start below)
        wj = fin ? (double)wjf : 0.0;
        wm = fin ? (double)wmf : 0.0;
    } else {
        const double ia = rcp(a);
        const double kf = exp_s<false>(-0.6931471805599453 * ia), c = 2.0 * kf;    // 2^(-1/a): phi = kf |d_m| on a face; q_m = c there
        // return onto the hexagon in double (the network hardening law has no float twin): see above
        double g = -(kf * dm - (m.Y + hp.H)) * rcp(-twomu * c * kf - hp.dH);
        const Hard h1 = hardening(m, alpha_p + g);
        const double g1 = g;
        g = fmax(g - (kf * (dm - twomu * c * g) - (m.Y + h1.H)) * rcp(-twomu * c * kf - h1.dH), 0.0);
        const double phi_f = m.Y + h1.H + h1.dH * (g - g1);
        const bool corner = mu * c * g > dm - dj;
        const double gc = fmax(g - (dm + dj - 6.0 * mu * g - 2.0 * phi_f) * rcp(-6.0 * mu - 2.0 * h1.dH), 1e-300);
        const double half_split = 0.5 * (dm - dj) * rcp(mu * gc);
        gam = corner ? gc : g;
        const double qj_c = fmin(fmax(1.0 - half_split, 0.05), 1.0);
        const double rj_f = fmin(fmax((dj - mu * c * g) * rcp(phi_f), 1e-300), 1.0);
        const double wj_face = fmax(log_pos(rj_f) * (a - 1.0), -700.0), wj_cap = log_pos(qj_c);
        const bool capped = !corner && (wj_face > wj_cap);
        wj = (corner || capped) ? wj_cap : wj_face;
        wm = log_pos(corner ? 2.0 - qj_c : (capped ? c - 0.9 * qj_c : c));
    }
    active = active && (gam > 0.0);
    bool done = !active, ok = false;
    for (int it = 0; it < kHosfordWarmMaxIt; ++it) {
        // q = e^w; r = e^(w / (a-1)): for the major difference |w_m / (a-1)| < 0.04 and its exponential is a short series
        const double um = wm * ia1;
        double rm = CM_SCALAR(1.0 / 720.0);
        rm = __builtin_fma(um, rm, CM_SCALAR(1.0 / 120.0));
        rm = __builtin_fma(um, rm, CM_SCALAR(1.0 / 24.0));
        rm = __builtin_fma(um, rm, CM_SCALAR(1.0 / 6.0));
        rm = __builtin_fma(um, rm, 0.5);
        rm = __builtin_fma(um, rm, 1.0);
        rm = __builtin_fma(um, rm, 1.0);
        const double qm = exp_s<false>(wm), qj = exp_s<false>(wj), rj = exp_s<false>(wj * ia1);
        const Hard h = hardening(m, alpha_p + gam);
        const double phi = m.Y + h.H, mg = mu * gam;
        const double Rm = phi * rm - dm + mg * (2.0 * qm + qj), Rj = phi * rj - dj + mg * (2.0 * qj + qm);
        const double Rn = qm * rm + qj * rj - 2.0;
        const double res = fmax(fmax(fabs(Rm), fabs(Rj)) * rcp(dm), fabs(Rn));
        if (!done && !(res < 1e300)) done = true;                              // not finite: give up, the reference iteration takes over
        if (!done && res < 1e-13) { done = true; ok = true; }                  // converged as evaluated
        // quadratic convergence: from 1e-7 the next iterate is converged to round-off, and newton_s checks it anyway
        const bool last = res < 1e-7;
        double dwj = 0.0;
        if (!done) {
            const double J00 = phi * rm * ia1 + 2.0 * mg * qm, J01 = mg * qj, J02 = h.dH * rm + mu * (2.0 * qm + qj);
            const double J10 = mg * qm, J11 = phi * rj * ia1 + 2.0 * mg * qj, J12 = h.dH * rj + mu * (2.0 * qj + qm);
            const double J20 = ap * qm * rm, J21 = ap * qj * rj;               // J22 = 0
            // Cramer on the 3 x 3 system J delta = R
            const double c00 = -J12 * J21, c01 = J12 * J20, c02 = J10 * J21 - J11 * J20;
            const double det = J00 * c00 + J01 * c01 + J02 * c02;
            const double idet = rcp(det);
            const double dwm = (Rm * c00 + J01 * (J12 * Rn) + J02 * (Rj * J21 - J11 * Rn)) * idet;
            dwj = (J00 * (-J12 * Rn) + Rm * c01 + J02 * (J10 * Rn - Rj * J20)) * idet;
            const double dgm = (J00 * (J11 * Rn - Rj * J21) + J01 * (Rj * J20 - J10 * Rn) + Rm * c02) * idet;
            wm = fmin(fmax(wm - dwm, -3.0), 0.75);                             // q_m stays in [0.05, 2.1]
            gam = fmax(gam - dgm, 1e-3 * gam);
            if (last) { done = true; ok = true; }
        }
        // A large step that LOWERS q_j = e^(w_j) is taken in the variable q_j (q <- q (1 - dw): the equations are nearly linear in
        // it, and the exponential would creep down by one unit of w per iteration); everything else in w (at most e^2 up per step).
        // The logarithm this needs is only evaluated while some lane of the wavefront takes such a step (the first one or two).
        const bool qstep = (dwj > 0.25);
        double lg = 0.0;
        if (__any(qstep)) lg = log_pos(fmax(1.0 - dwj, 0.05));
        wj = fmax(wj + (qstep ? lg : fmin(-dwj, 2.0)), -700.0);
        if (!__any(!done)) break;
    }
    ok = ok && (gam > 0.0) && (wm < 1.0) && (wj < 1.0);
    if (ok) {
        const double qm = exp_s(wm), qj = exp_s(wj);
        const double p0 = 0.5 * ((d0 >= 0.0) ? 1.0 : -1.0) * (m0 ? qm : (j0 ? qj : 0.0));
        const double p1 = 0.5 * ((d1 >= 0.0) ? 1.0 : -1.0) * (m1 ? qm : (j1 ? qj : 0.0));
        const double p2 = 0.5 * ((d2 >= 0.0) ? 1.0 : -1.0) * (m2 ? qm : (j2 ? qj : 0.0));
        x0[0] = xp[0] + gam * (p0 - p2);
        x0[3] = xp[3] + gam * (p1 - p0);
        x0[5] = xp[5] + gam * (p2 - p1);
        x0[6] = alpha_p + gam;
    }
    return true;
}

// ---- Hill, FULL_3D: the classical scalar return map as warm start of the local Newton -----------------------------------------
// For a quadratic surface phi = sqrt(s^T A s) the backward-Euler equations  s = s_trial - 2 mu dgam W^-1 A s / phi,
// phi = Y + H(alpha_prev + dgam)  reduce to ONE scalar equation in kappa = 2 mu dgam / phi:
//     (I + kappa W^-1 A) s = s_trial     3 x 3 on the normal entries (A3 has the null vector 1: in the orthonormal deviatoric
//                                        coordinates y = E^T s_n, E = [(1,-1,0)/sqrt2, (1,1,-2)/sqrt6], a 2 x 2 system with
//                                        A2 = E^T A3 E; the mean stress does not move) and three shear scalars
//     F(kappa) = phi(kappa) - Y - H(alpha_prev + kappa phi(kappa) / 2mu) = 0 ,    phi^2 = y^T A2 y + sum_shear a_kk s_k^2
//     phi phi' = -u^T (I + kappa A2)^-1 u - 1/2 sum_shear (a_kk s_k)^2 / (1 + kappa a_kk / 2) ,   u = A2 y
// F is decreasing and convex on kappa >= 0, so Newton from kappa = 0 converges monotonically (~40 flops and one hardening
// evaluation per step instead of the 7-dof step's ~240 instructions).  As with hosford_warm_start the result is only the START
// of the reference's Newton: newton_s evaluates the reference residual there and applies the reference's convergence test
// (relative to ||C(x_prev)||), so a converged map costs one residual evaluation and anything else is finished by the general
// iteration.  CM_SOLVER_GENERAL_NEWTON / CM_SOLVER_REFERENCE_ITERATES keep the iteration from x_prev.
constexpr int kHillWarmMaxIt = 12;
// (Tried for the hybrid Hill + network surface with the network term frozen at its trial value: on BASELINE configs[3] the
// reference's Newton needs MORE iterations from that seed than from x_prev, 4.84 against 4.46 -- the network term bends the flow
// direction too much for the Hill normal to be a useful start.  Not built.)
CM_D bool hill_warm_start(const cm_model_desc& m, const double eg[6], const double* xp, double* x0, double& n0sq, bool lane_valid) {
#pragma unroll
    for (int k = 0; k < 7; ++k) x0[k] = xp[k];
    const QuadForm q = quad_form<CM_YIELD_HILL>(m);
    const double twomu = 2.0 * m.mu, i2mu = half_over_mu(m);
    double e[6], s[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) e[k] = eg[k] - xp[k];
    const double lt = m.lambda * (e[0] + e[3] + e[5]);
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = twomu * e[k] + (kDiag[k] ? lt : 0.0);
    // deviatoric coordinates of the normal entries and the 2 x 2 form of A3 in them
    constexpr double r2 = 0.7071067811865476, r6 = 0.4082482904638631, r12 = 0.2886751345948129;
    const double A11 = 0.5 * (q.a00 - 2.0 * q.a03 + q.a33);
    const double A12 = (q.a00 - q.a33 - 2.0 * q.a05 + 2.0 * q.a35) * r12;
    const double A22 = (q.a00 + q.a33 + 4.0 * q.a55 + 2.0 * q.a03 - 4.0 * q.a05 - 4.0 * q.a35) * (1.0 / 6.0);
    const double yt1 = (s[0] - s[3]) * r2, yt2 = (s[0] + s[3] - 2.0 * s[5]) * r6;
    const double h1 = 0.5 * q.a11, h2 = 0.5 * q.a22, h4 = 0.5 * q.a44;      // W^-1 A on the shear entries
    const double ph2_tr = yt1 * (A11 * yt1 + A12 * yt2) + yt2 * (A12 * yt1 + A22 * yt2)
                        + q.a11 * s[1] * s[1] + q.a22 * s[2] * s[2] + q.a44 * s[4] * s[4];
    const double phi_tr = sqrt(fmax(ph2_tr, 0.0));
    const double alpha_p = xp[6];
    const Hard hp = hardening(m, alpha_p);
    const double f0 = (phi_tr - (m.Y + hp.H)) * i2mu;
    const bool plastic0 = (f0 > m.yield_tol) || (fabs(f0) < m.yield_tol);
    n0sq = plastic0 ? f0 * f0 : 0.0;                                           // C(x_prev) = [0, f0] resp. 0 (elastic branch)
    const bool active = lane_valid && plastic0 && (f0 > 0.0) && !(n0sq < m.abs_tol * m.abs_tol);     // (see hosford_warm_start)
    if (!__any(active)) return true;
    double kap = 0.0, phi = phi_tr, y1 = yt1, y2 = yt2, u1 = 0.0, u2 = 0.0, s1 = s[1], s2 = s[2], s4 = s[4];
    bool done = !active, ok = false;
    // Newton in c = lb kappa / (1 + lb kappa), lb = 3/2 (J2's eigenvalue of W^-1 A; Hill's lie around it): phi ~ phi_trial (1 - c)
    // and dgam ~ c phi_trial / (2 mu lb) are nearly LINEAR in c (exactly so for J2 with linear hardening), where in kappa
    // phi ~ 1 / (1 + lb kappa) is strongly convex and Newton from kappa = 0 creeps (6-7 steps against 3-5)
    constexpr double lb = 1.5;
    {   // kappa = 0: the trial state itself (B = I, the hardening at alpha_prev is already there)
        u1 = A11 * yt1 + A12 * yt2; u2 = A12 * yt1 + A22 * yt2;
        const double t1 = q.a11 * s[1], t2 = q.a22 * s[2], t4 = q.a44 * s[4];
        const double iphi = (ph2_tr > 0.0) ? rcp(phi_tr) : 0.0;
        const double dphi = -(u1 * u1 + u2 * u2 + 0.5 * (t1 * t1 + t2 * t2 + t4 * t4)) * iphi;
        const double F = phi_tr - (m.Y + hp.H), dF = dphi - hp.dH * phi_tr * i2mu;
        const double cn = fmin(fmax(-F * lb * rcp(dF), 0.0), 0.999999);
        if (!done) kap = cn * rcp(lb * (1.0 - cn));
    }
    if (hardening_seed_ok(m)) {
        // two Newton steps in float from there (cm_device.hpp, "single-precision seeds")
        const float A11f = (float)A11, A12f = (float)A12, A22f = (float)A22, yt1f = (float)yt1, yt2f = (float)yt2;
        const float h1f = (float)h1, h2f = (float)h2, h4f = (float)h4, s1t = (float)s[1], s2t = (float)s[2], s4t = (float)s[4];
        const float a11f = (float)q.a11, a22f = (float)q.a22, a44f = (float)q.a44;
        const float Yf = (float)m.Y, apf = (float)alpha_p, i2muf = (float)i2mu, lbf = (float)lb;
        float kf = (float)kap;
#pragma unroll
        for (int sit = 0; sit < 2; ++sit) {
            const float b11 = 1.0f + kf * A11f, b22 = 1.0f + kf * A22f, b12 = kf * A12f;
            const float idet = rcp_f(b11 * b22 - b12 * b12);
            const float z1 = (b22 * yt1f - b12 * yt2f) * idet, z2 = (b11 * yt2f - b12 * yt1f) * idet;
            const float i1 = rcp_f(1.0f + kf * h1f), i2 = rcp_f(1.0f + kf * h2f), i4 = rcp_f(1.0f + kf * h4f);
            const float r1 = s1t * i1, r2f = s2t * i2, r4 = s4t * i4;
            const float w1 = A11f * z1 + A12f * z2, w2 = A12f * z1 + A22f * z2;
            const float t1 = a11f * r1, t2 = a22f * r2f, t4 = a44f * r4;
            const float ph2 = z1 * w1 + z2 * w2 + t1 * r1 + t2 * r2f + t4 * r4;
            const float rph = rsq_f(ph2), ph = ph2 * rph;
            const float v1 = (b22 * w1 - b12 * w2) * idet, v2 = (b11 * w2 - b12 * w1) * idet;
            const float dph = -(w1 * v1 + w2 * v2 + 0.5f * (t1 * t1 * i1 + t2 * t2 * i2 + t4 * t4 * i4)) * rph;
            const HardF h = hardening_f(m, apf + kf * ph * i2muf);
            const float F = ph - (Yf + h.H), dF = dph - h.dH * (ph + kf * dph) * i2muf;
            const float w = 1.0f + lbf * kf;
            const float cn = lbf * kf * rcp_f(w) - F * lbf * rcp_f(dF * w * w);
            if (cn >= 0.0f && cn < 0.999999f) kf = cn * rcp_f(lbf * (1.0f - cn));   // (a step out of the interval, or not finite, is dropped)
        }
        if (!done) kap = (double)kf;
    }
    for (int it = 0; it < kHillWarmMaxIt; ++it) {
        const double b11 = 1.0 + kap * A11, b22 = 1.0 + kap * A22, b12 = kap * A12;
        const double idet = rcp(b11 * b22 - b12 * b12);
        y1 = (b22 * yt1 - b12 * yt2) * idet; y2 = (b11 * yt2 - b12 * yt1) * idet;
        const double i1 = rcp(1.0 + kap * h1), i2 = rcp(1.0 + kap * h2), i4 = rcp(1.0 + kap * h4);
        s1 = s[1] * i1; s2 = s[2] * i2; s4 = s[4] * i4;
        u1 = A11 * y1 + A12 * y2; u2 = A12 * y1 + A22 * y2;
        const double t1 = q.a11 * s1, t2 = q.a22 * s2, t4 = q.a44 * s4;
        const double ph2 = y1 * u1 + y2 * u2 + t1 * s1 + t2 * s2 + t4 * s4;
        const double iphi = (ph2 > 0.0) ? rsqrt_pos(ph2) : 0.0;
        phi = ph2 * iphi;
        // phi phi' = -u^T (I + kappa A2)^-1 u - 1/2 sum (a_kk s_k)^2 / (1 + kappa a_kk / 2)
        const double v1 = (b22 * u1 - b12 * u2) * idet, v2 = (b11 * u2 - b12 * u1) * idet;
        const double dphi = -(u1 * v1 + u2 * v2 + 0.5 * (t1 * t1 * i1 + t2 * t2 * i2 + t4 * t4 * i4)) * iphi;
        const Hard h = hardening(m, alpha_p + kap * phi * i2mu);
        const double F = phi - (m.Y + h.H), dF = dphi - h.dH * (phi + kap * dphi) * i2mu;
        const double res = fabs(F);
        if (!done && !(res < 1e300)) done = true;
        if (!done && res < 1e-14 * phi_tr) { done = true; ok = true; }
        if (!done) {
            const double w = 1.0 + lb * kap;
            const double cn = fmin(fmax(lb * kap * rcp(w) - F * lb * rcp(dF * w * w), 0.0), 0.999999);
            kap = cn * rcp(lb * (1.0 - cn));
            if (res < 1e-8 * phi_tr) { done = true; ok = true; }   // quadratic convergence: this iterate is converged to round-off
        }
        if (!__any(!done)) break;
    }
    if (ok) {
        // the state at kappa (one more evaluation of the linear maps at the final kappa), then x = x_prev + dgam n, n_k = (A s)_k / (phi w_k)
        const double b11 = 1.0 + kap * A11, b22 = 1.0 + kap * A22, b12 = kap * A12;
        const double idet = rcp(b11 * b22 - b12 * b12);
        y1 = (b22 * yt1 - b12 * yt2) * idet; y2 = (b11 * yt2 - b12 * yt1) * idet;
        s1 = s[1] * rcp(1.0 + kap * h1); s2 = s[2] * rcp(1.0 + kap * h2); s4 = s[4] * rcp(1.0 + kap * h4);
        u1 = A11 * y1 + A12 * y2; u2 = A12 * y1 + A22 * y2;
        const double ph2 = y1 * u1 + y2 * u2 + q.a11 * s1 * s1 + q.a22 * s2 * s2 + q.a44 * s4 * s4;
        phi = sqrt(fmax(ph2, 0.0));
        const double dg_over_phi = kap * i2mu;                     // dgam / phi
        x0[0] = xp[0] + dg_over_phi * (u1 * r2 + u2 * r6);
        x0[3] = xp[3] + dg_over_phi * (-u1 * r2 + u2 * r6);
        x0[5] = xp[5] + dg_over_phi * (-2.0 * u2 * r6);
        x0[1] = xp[1] + dg_over_phi * (0.5 * q.a11 * s1);
        x0[2] = xp[2] + dg_over_phi * (0.5 * q.a22 * s2);
        x0[4] = xp[4] + dg_over_phi * (0.5 * q.a44 * s4);
        x0[6] = alpha_p + dg_over_phi * phi;
    }
    return true;
}

// ---- Hill, PLANE_STRESS: the same scalar map with the stretch eliminated ---------------------------------------------------------
// The trial stress is linear in the out-of-plane stretch t = F33 - 1:  s_trial(t) = s0 + t sz,  s0 = Cel (eg - v_prev),  sz = Cel z,
// so for a given kappa = 2 mu dgam / phi the stress  s = (I + kappa W^-1 A)^-1 (s0 + t sz)  is linear in t too, and the plane-stress
// row (w o z) . s = 0 gives t in closed form:
//     t(kappa) = -N / D ,   N = zeta . u0 + tau p0 + 2 sum_shear z_k s0_k i_k ,   D = the same with (uz, pz, sz_k) ,
// with u0 = B2^-1 y0, uz = B2^-1 yz the deviatoric coordinates of the normal entries (B2 = I + kappa A2, see hill_warm_start; the
// mean stresses p0, pz pass through: A has the null vector 1), i_k = 1 / (1 + kappa a_kk / 2), zeta = E^T z_n, tau = tr z.
// What is left is  F(kappa) = phi(kappa) - Y - H(alpha_prev + kappa phi / 2mu) = 0  with  phi^2 = y^T A2 y + sum a_kk s_k^2,
// y = u0 + t uz, s_k = (s0_k + t sz_k) i_k, and dF/dkappa from  d(B2^-1 v) = -B2^-1 A2 B2^-1 v,  d i_k = -h_k i_k^2,
// t' = -(N' D - N D') / D^2.  Newton in c = lb kappa / (1 + lb kappa), a closed-form first step, two steps in float, then double
// (cm_device.hpp "single-precision seeds").  As under FULL_3D the result only STARTS the reference's Newton (newton_s on the 8-dof
// residual, its convergence test against ||C(x_prev)||); the elastic step (kappa = 0, stretch t(0)) is taken only when the state
// at x_prev, with the old stretch, is elastic too (two roots otherwise: see newton_j2_plane).
template <class T> struct HillPsC {
    T A11, A12, A22, h1, h2, h4, a11, a22, a44;        // the quadratic form
    T y01, y02, yz1, yz2, p0, pz, s01, s02, s04, sz1, sz2, sz4;   // s0 and sz: deviatoric coordinates, mean, shear entries
    T ze1, ze2, tau, z1, z2, z4;                        // the plane-stress row
    T Y, alpha_p, i2mu;
};
// F, dF / dkappa, and the state (t, y, s_k, phi) at kappa
template <class T, class HF>
CM_D void hill_ps_eval(const HillPsC<T>& c, T kap, HF&& hard, T& F, T& dF, T& t, T& y1, T& y2, T& s1, T& s2, T& s4, T& phi) {
    const T one = (T)1, two = (T)2;
    const T b11 = one + kap * c.A11, b22 = one + kap * c.A22, b12 = kap * c.A12;
    T idet, i1, i2, i4;
    if constexpr (std::is_same<T, float>::value) {
        idet = rcp_f(b11 * b22 - b12 * b12); i1 = rcp_f(one + kap * c.h1); i2 = rcp_f(one + kap * c.h2); i4 = rcp_f(one + kap * c.h4);
    } else {
        idet = rcp(b11 * b22 - b12 * b12); i1 = rcp(one + kap * c.h1); i2 = rcp(one + kap * c.h2); i4 = rcp(one + kap * c.h4);
    }
    auto inv1 = [&](T v1, T v2) { return (b22 * v1 - b12 * v2) * idet; };
    auto inv2 = [&](T v1, T v2) { return (b11 * v2 - b12 * v1) * idet; };
    const T u01 = inv1(c.y01, c.y02), u02 = inv2(c.y01, c.y02), uz1 = inv1(c.yz1, c.yz2), uz2 = inv2(c.yz1, c.yz2);
    const T N = c.ze1 * u01 + c.ze2 * u02 + c.tau * c.p0 + two * (c.z1 * c.s01 * i1 + c.z2 * c.s02 * i2 + c.z4 * c.s04 * i4);
    const T D = c.ze1 * uz1 + c.ze2 * uz2 + c.tau * c.pz + two * (c.z1 * c.sz1 * i1 + c.z2 * c.sz2 * i2 + c.z4 * c.sz4 * i4);
    T iD;
    if constexpr (std::is_same<T, float>::value) iD = rcp_f(D); else iD = rcp(D);
    t = -N * iD;
    y1 = u01 + t * uz1; y2 = u02 + t * uz2;
    s1 = (c.s01 + t * c.sz1) * i1; s2 = (c.s02 + t * c.sz2) * i2; s4 = (c.s04 + t * c.sz4) * i4;
    const T Ay1 = c.A11 * y1 + c.A12 * y2, Ay2 = c.A12 * y1 + c.A22 * y2;
    const T t1 = c.a11 * s1, t2 = c.a22 * s2, t4 = c.a44 * s4;
    const T ph2 = y1 * Ay1 + y2 * Ay2 + t1 * s1 + t2 * s2 + t4 * s4;
    T iphi;
    if constexpr (std::is_same<T, float>::value) iphi = rsq_f(ph2); else iphi = (ph2 > 0.0) ? rsqrt_pos(ph2) : 0.0;
    phi = ph2 * iphi;
    // derivatives w.r.t. kappa
    const T a01 = c.A11 * u01 + c.A12 * u02, a02 = c.A12 * u01 + c.A22 * u02, az1 = c.A11 * uz1 + c.A12 * uz2, az2 = c.A12 * uz1 + c.A22 * uz2;
    const T du01 = -inv1(a01, a02), du02 = -inv2(a01, a02), duz1 = -inv1(az1, az2), duz2 = -inv2(az1, az2);
    const T g1 = c.h1 * i1 * i1, g2 = c.h2 * i2 * i2, g4 = c.h4 * i4 * i4;
    const T dN = c.ze1 * du01 + c.ze2 * du02 - two * (c.z1 * c.s01 * g1 + c.z2 * c.s02 * g2 + c.z4 * c.s04 * g4);
    const T dD = c.ze1 * duz1 + c.ze2 * duz2 - two * (c.z1 * c.sz1 * g1 + c.z2 * c.sz2 * g2 + c.z4 * c.sz4 * g4);
    const T dt = -(dN * D - N * dD) * (iD * iD);
    const T dy1 = du01 + dt * uz1 + t * duz1, dy2 = du02 + dt * uz2 + t * duz2;
    const T ds1 = dt * c.sz1 * i1 - s1 * c.h1 * i1, ds2 = dt * c.sz2 * i2 - s2 * c.h2 * i2, ds4 = dt * c.sz4 * i4 - s4 * c.h4 * i4;
    const T dphi = (Ay1 * dy1 + Ay2 * dy2 + t1 * ds1 + t2 * ds2 + t4 * ds4) * iphi;
    T H, dH;
    hard(c.alpha_p + kap * phi * c.i2mu, H, dH);
    F = phi - (c.Y + H);
    dF = dphi - dH * (phi + kap * dphi) * c.i2mu;
}
CM_D bool hill_ps_warm_start(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, double* x0, double& n0sq,
                             bool lane_valid) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x0[k] = xp[k];
    const QuadForm q = quad_form<CM_YIELD_HILL>(m);
    const double twomu = 2.0 * m.mu, i2mu = half_over_mu(m), Kb = m.lambda + (2.0 / 3.0) * m.mu;
    constexpr double r2 = 0.7071067811865476, r6 = 0.4082482904638631, r12 = 0.2886751345948129;
    HillPsC<double> c;
    c.A11 = 0.5 * (q.a00 - 2.0 * q.a03 + q.a33);
    c.A12 = (q.a00 - q.a33 - 2.0 * q.a05 + 2.0 * q.a35) * r12;
    c.A22 = (q.a00 + q.a33 + 4.0 * q.a55 + 2.0 * q.a03 - 4.0 * q.a05 - 4.0 * q.a35) * (1.0 / 6.0);
    c.h1 = 0.5 * q.a11; c.h2 = 0.5 * q.a22; c.h4 = 0.5 * q.a44; c.a11 = q.a11; c.a22 = q.a22; c.a44 = q.a44;
    double e[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) e[k] = eg[k] - xp[k];
    const double tre = e[0] + e[3] + e[5], trz = z[0] + z[3] + z[5];
    // deviatoric coordinates / mean / shear entries of s0 = Cel e and sz = Cel z (the lambda tr terms only enter the means)
    c.y01 = twomu * (e[0] - e[3]) * r2; c.y02 = twomu * (e[0] + e[3] - 2.0 * e[5]) * r6; c.p0 = Kb * tre;
    c.yz1 = twomu * (z[0] - z[3]) * r2; c.yz2 = twomu * (z[0] + z[3] - 2.0 * z[5]) * r6; c.pz = Kb * trz;
    c.s01 = twomu * e[1]; c.s02 = twomu * e[2]; c.s04 = twomu * e[4];
    c.sz1 = twomu * z[1]; c.sz2 = twomu * z[2]; c.sz4 = twomu * z[4];
    c.ze1 = (z[0] - z[3]) * r2; c.ze2 = (z[0] + z[3] - 2.0 * z[5]) * r6; c.tau = trz; c.z1 = z[1]; c.z2 = z[2]; c.z4 = z[4];
    c.Y = m.Y; c.alpha_p = xp[6]; c.i2mu = i2mu;
    const Hard hp = hardening(m, c.alpha_p);
    bool plastic_prev;
    {   // ||C(x_prev)||^2 = C6^2 + C7^2 with the OLD stretch (the strain rows vanish at x_prev)
        const double tp = xp[7] - 1.0;
        const double y1 = c.y01 + tp * c.yz1, y2 = c.y02 + tp * c.yz2, s1 = c.s01 + tp * c.sz1, s2 = c.s02 + tp * c.sz2, s4 = c.s04 + tp * c.sz4;
        const double ph2 = y1 * (c.A11 * y1 + c.A12 * y2) + y2 * (c.A12 * y1 + c.A22 * y2) + q.a11 * s1 * s1 + q.a22 * s2 * s2 + q.a44 * s4 * s4;
        const double phi0 = (ph2 > 0.0) ? ph2 * rsqrt_pos(ph2) : 0.0;
        const double f0 = (phi0 - (m.Y + hp.H)) * i2mu;
        plastic_prev = (f0 > m.yield_tol) || (fabs(f0) < m.yield_tol);
        const double c6 = plastic_prev ? f0 : 0.0;
        const double c7 = (c.ze1 * y1 + c.ze2 * y2 + c.tau * (c.p0 + tp * c.pz) + 2.0 * (c.z1 * s1 + c.z2 * s2 + c.z4 * s4)) * i2mu;
        n0sq = c6 * c6 + c7 * c7;
    }
    // (a state that already passes the reference's test at x_prev is returned as it is, 0 iterations)
    bool done = !lane_valid || (n0sq < m.abs_tol * m.abs_tol), ok = false;
    if (!__any(!done)) return true;
    constexpr double lb = 1.5;
    auto hard_d = [&](double alpha, double& H, double& dH) { const Hard h = hardening(m, alpha); H = h.H; dH = h.dH; };
    double kap = 0.0, t = 0.0, y1, y2, s1, s2, s4, phi, F, dF;
    {   // kappa = 0: the elastic step with the relaxed stretch t(0); the hardening there is the one at alpha_prev
        auto hard_0 = [&](double, double& H, double& dH) { H = hp.H; dH = hp.dH; };
        hill_ps_eval<double>(c, 0.0, hard_0, F, dF, t, y1, y2, s1, s2, s4, phi);
        if (!done && !(F > 0.0)) {
            done = true;
            if (!plastic_prev) x0[7] = 1.0 + t;                  // elastic at x_prev and after the relaxation: the elastic root
        }
        if (!done) { const double cn = fmin(fmax(-F * lb * rcp(dF), 0.0), 0.999999); kap = cn * rcp(lb * (1.0 - cn)); }
    }
    if (__any(!done) && hardening_seed_ok(m)) {
        HillPsC<float> cf;
        cf.A11 = (float)c.A11; cf.A12 = (float)c.A12; cf.A22 = (float)c.A22; cf.h1 = (float)c.h1; cf.h2 = (float)c.h2; cf.h4 = (float)c.h4;
        cf.a11 = (float)c.a11; cf.a22 = (float)c.a22; cf.a44 = (float)c.a44;
        cf.y01 = (float)c.y01; cf.y02 = (float)c.y02; cf.yz1 = (float)c.yz1; cf.yz2 = (float)c.yz2; cf.p0 = (float)c.p0; cf.pz = (float)c.pz;
        cf.s01 = (float)c.s01; cf.s02 = (float)c.s02; cf.s04 = (float)c.s04; cf.sz1 = (float)c.sz1; cf.sz2 = (float)c.sz2; cf.sz4 = (float)c.sz4;
        cf.ze1 = (float)c.ze1; cf.ze2 = (float)c.ze2; cf.tau = (float)c.tau; cf.z1 = (float)c.z1; cf.z2 = (float)c.z2; cf.z4 = (float)c.z4;
        cf.Y = (float)c.Y; cf.alpha_p = (float)c.alpha_p; cf.i2mu = (float)c.i2mu;
        auto hard_f = [&](float alpha, float& H, float& dH) { const HardF h = hardening_f(m, alpha); H = h.H; dH = h.dH; };
        float kf = (float)kap;
#pragma unroll
        for (int sit = 0; sit < 2; ++sit) {
            float Ff, dFf, tf, a1, a2, b1, b2, b4, pf;
            hill_ps_eval<float>(cf, kf, hard_f, Ff, dFf, tf, a1, a2, b1, b2, b4, pf);
            const float w = 1.0f + (float)lb * kf;
            const float cn = (float)lb * kf * rcp_f(w) - Ff * (float)lb * rcp_f(dFf * w * w);
            if (cn >= 0.0f && cn < 0.999999f) kf = cn * rcp_f((float)lb * (1.0f - cn));
        }
        if (!done) kap = (double)kf;
    }
    const double phi_ref = m.Y + hp.H;
    for (int it = 0; it < kHillWarmMaxIt; ++it) {
        double tn, a1, a2, b1, b2, b4, pn;
        hill_ps_eval<double>(c, kap, hard_d, F, dF, tn, a1, a2, b1, b2, b4, pn);
        const double res = fabs(F);
        if (!done && !(res < 1e300)) done = true;
        if (!done && res < 1e-14 * phi_ref) { done = true; ok = true; }
        if (!done) {
            const double w = 1.0 + lb * kap;
            const double cn = fmin(fmax(lb * kap * rcp(w) - F * lb * rcp(dF * w * w), 0.0), 0.999999);
            kap = cn * rcp(lb * (1.0 - cn));
            if (res < 1e-8 * phi_ref) { done = true; ok = true; }  // quadratic convergence: this iterate is converged to round-off
        }
        if (!__any(!done)) break;
    }
    if (ok) {
        auto hard_n = [&](double, double& H, double& dH) { H = 0.0; dH = 0.0; };      // (state only)
        hill_ps_eval<double>(c, kap, hard_n, F, dF, t, y1, y2, s1, s2, s4, phi);
        const double u1 = c.A11 * y1 + c.A12 * y2, u2 = c.A12 * y1 + c.A22 * y2, dgp = kap * i2mu;     // dgam / phi
        x0[0] = xp[0] + dgp * (u1 * r2 + u2 * r6);
        x0[3] = xp[3] + dgp * (-u1 * r2 + u2 * r6);
        x0[5] = xp[5] + dgp * (-2.0 * u2 * r6);
        x0[1] = xp[1] + dgp * (0.5 * q.a11 * s1);
        x0[2] = xp[2] + dgp * (0.5 * q.a22 * s2);
        x0[4] = xp[4] + dgp * (0.5 * q.a44 * s4);
        x0[6] = c.alpha_p + dgp * phi;
        x0[7] = 1.0 + t;
    }
    return true;
}

#if defined(CM_HOST_BUILD)
// host build only (tests): how many points left newton_j2_line / newton_j2_plane for the general path
inline long long& subspace_fallbacks() { static long long n = 0; return n; }
#define CM_COUNT_FALLBACK() (++subspace_fallbacks())
#else
#define CM_COUNT_FALLBACK() ((void)0)
#endif

// ---- J2, FULL_3D, plain Newton: the same iteration restricted to its invariant subspace ---------------------
// Default for J2 / FULL_3D (cm_model_desc.solver_flags & CM_SOLVER_GENERAL_NEWTON turns it off).  For J2 the 7-dof Newton iterates started at
// x_prev never leave the radial line v = v_prev + dgam n_trial: C[0:6] vanishes on it, dC/dx maps it to itself,
// and the step reduces *exactly* to the scalar Newton step on f(alpha):
//     phi(dgam) = phi_trial - 3 mu dgam ,  f = (phi - Y - H(alpha)) / 2mu ,  d alpha = f / ( -(3 mu + H') / 2mu )
// (derivation in DESIGN.md section 3).  Same iterates, same iteration counts, same convergence test (||C|| = |f|)
// as newton_s; ~50 instead of ~275 instructions per iteration.  An iterate that falls on the elastic side of the
// branch select (cannot happen for concave hardening) sends the lane to the general path.
// LS: with the line search on, a step on the line is the first trial (alpha = 1) of the reference's search; its
// merit is f^2 / 2 (C[0:6] = 0 on the line), so the Armijo test is f_new^2 <= (1 - 2 c1) f_old^2.  A lane whose
// full step fails it (or is not finite) leaves for the general line-search path; every other lane produces the
// iterates the general path would.
template <bool LS = false>
CM_D uint32_t newton_j2_line(const cm_model_desc& m, const double eg[6], const double* xp, double* x, bool lane_valid,
                             EvalS<CM_YIELD_J2>& ev, LaneStage stage = LaneStage{nullptr, 0}) {
    double C[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = xp[k];
    residual_s<CM_YIELD_J2>(m, eg, x, xp, ev, C);              // trial state: phi_trial, normal, f0
    const double n0sq = dot<7>(C, C);
    const double rel2 = m.rel_tol * m.rel_tol * n0sq, abs2 = m.abs_tol * m.abs_tol;
    const double phi_tr = ev.y.phi, i2mu = half_over_mu(m), three_mu = 3.0 * m.mu, alpha_p = xp[6];
    double alpha = alpha_p, f = C[6], dH = ev.hd.dH;
    int it = 0;
    bool running = lane_valid, fallback = false;
    uint32_t flags = 0;
    for (;;) {
        const double nsq = ev.plastic ? f * f : n0sq;            // elastic at the trial state: C_e(x_prev) = 0
        const bool conv = (nsq < rel2) || (nsq < abs2);
        if (running && conv) { running = false; flags |= CM_STATUS_CONVERGED; }
        if (running && it >= m.max_iters) running = false;
        if (!__any(running)) break;
        if (running) {
            const double f_old = f;
            alpha -= f * rcp(-(three_mu + dH) * i2mu);
            const Hard hd = hardening(m, alpha);
            ev.hd = hd;                                              // hardening at the current alpha (reused below)
            f = (phi_tr - three_mu * (alpha - alpha_p) - (m.Y + hd.H)) * i2mu;
            dH = hd.dH;
            if (!((f > m.yield_tol) || (fabs(f) < m.yield_tol))) { fallback = true; running = false; }
            if constexpr (LS) {
                // phi(1) <= phi(0) + c1 phi'(0) with phi = f^2 / 2, phi'(0) = -f_old^2
                const double ph = 0.5 * f * f, ph0 = 0.5 * f_old * f_old;
                if (!(isfinite(ph) && ph <= ph0 + m.ls_c1 * (-(f_old * f_old)))) { fallback = true; running = false; }
            }
            ++it;
        }
    }
    const double dgam = alpha - alpha_p;
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = xp[k] + dgam * ev.y.gt[k] * kIW[k];
    x[6] = alpha;
    uint32_t st = flags | (uint32_t)it;
    // The evaluation at the returned state, which the reverse sweep needs, follows from the trial evaluation: the normal is
    // the trial normal, e and s move along it, phi = phi_trial - 3 mu dgam; ev.hd is hardening(x[6]) already (from the last
    // step of the loop, or from the trial evaluation when no step was taken).  The scalar f of the line and the 7-dof residual
    // at the rounded state differ by round-off (~1e-18 against tolerances of 1e-14), and a lane that stopped within that
    // distance of the tolerance must behave exactly like the general path (e.g. re-applying the same strain to the returned
    // state is a 0-iteration step): a lane within 1 % of the tolerance (a few in 10^4) takes the general path.
    if (ev.plastic) {
        const double twomu_dg = 2.0 * m.mu * dgam;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double nk = ev.y.gt[k] * kIW[k];
            ev.e[k] -= dgam * nk;
            ev.s[k] -= twomu_dg * nk;
        }
        ev.y.phi = phi_tr - three_mu * dgam;
        ev.y.rho = rcp(ev.y.phi);
        ev.f = f;
        ev.dgam = dgam;
        if ((flags & CM_STATUS_CONVERGED) && !((f * f < 0.99 * rel2) || (f * f < 0.99 * abs2))) fallback = true;
    }
    if (__any(fallback)) {
        if (fallback) { CM_COUNT_FALLBACK(); st = newton_s<CM_YIELD_J2, LS>(m, eg, xp, x, lane_valid, ev, stage); }
    }
    return st;
}

// ---- J2, PLANE_STRESS: the same Newton iteration in the coordinates of the plane it never leaves ---------------------
// Default for J2 / PLANE_STRESS (CM_SOLVER_GENERAL_NEWTON turns it off).  With a = dev(eg - v_prev),
// b = dev z (z = V(q3 q3^T), the direction F33 acts in) and t = F33 - 1 the deviatoric elastic strain is
// a + t b - (v - v_prev), the J2 normal is parallel to it, and dC/dx maps span{a, b} x (alpha, F33) to itself: the 8-dof Newton
// iterates started at x_prev are  v = v_prev + c_a a + c_b b  for all iterations, and the step is *exactly* the Newton step
// of the four residuals in u = (c_a, c_b, alpha, t).  With p = 1 - c_a, q = t - c_b, P = |p a + q b|^2, phi = mu sqrt(6 P),
// g = dgam 3 mu / phi (0 on the elastic side of the branch select):
//     r_a = c_a - g p ,  r_b = c_b - g q ,  C6 = f or dgam ,  C7 = z : sigma / 2mu = p (z:a) + q (z:b) + K tr(z) (tr0 + t tr z) / 2mu
//     ||C||^2 = r_a^2 (a.a) + 2 r_a r_b (a.b) + r_b^2 (b.b) + C6^2 + C7^2      ((.) = sums over the 6 stored entries)
// The 2x2 block d r_c / d c = (1 + g) I - (g / P) w v^T  (w = (p, q), v = (p a:a + q a:b, p a:b + q b:b), v.w = P) has the
// closed-form inverse (I + (g / P) w v^T) / (1 + g); (alpha, t) follow from its 2x2 Schur complement.  Same iterates, iteration
// counts and convergence test as newton_s<J2, false, PLANE_STRESS>; ~150 instead of ~320 instructions per iteration.  A lane that
// stops within round-off of the tolerance takes the general path (see the end of the function).
// LS: the full step is the first trial of the reference's search and is accepted when it passes the Armijo test on ||C||^2 / 2;
// a lane whose full step fails it leaves for the general line-search path (as in newton_j2_line<true>).
template <bool LS = false>
CM_D uint32_t newton_j2_plane(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, double* x,
                              bool lane_valid, EvalS<CM_YIELD_J2>& ev, LaneStage stage = LaneStage{nullptr, 0}) {
    double a[6], b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = eg[k] - xp[k];
    const double tr0 = a[0] + a[3] + a[5], trz = z[0] + z[3] + z[5];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (kDiag[k]) a[k] -= tr0 * (1.0 / 3.0);
        b[k] = z[k] - (kDiag[k] ? trz * (1.0 / 3.0) : 0.0);
    }
    double aa = 0.0, ab = 0.0, bb = 0.0, A2 = 0.0, AB2 = 0.0, B2 = 0.0, za = 0.0, zb = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        aa += kW[k] * a[k] * a[k]; ab += kW[k] * a[k] * b[k]; bb += kW[k] * b[k] * b[k];
        A2 += a[k] * a[k]; AB2 += a[k] * b[k]; B2 += b[k] * b[k];
        za += kW[k] * z[k] * a[k]; zb += kW[k] * z[k] * b[k];
    }
    const double i2mu = half_over_mu(m), sqrt6mu = 2.449489742783178 * m.mu;
    const double Kz = (m.lambda + (2.0 / 3.0) * m.mu) * i2mu * trz, d77 = zb + Kz * trz;
    const double alpha_p = xp[6], abs2 = m.abs_tol * m.abs_tol;
    double ca = 0.0, cb = 0.0, alpha = alpha_p, t = xp[7] - 1.0, rel2 = 0.0, nsq_from = 0.0;
    int it = 0;
    bool running = lane_valid, fallback = false, first = true;
    uint32_t flags = 0;
    Hard hd;
    if (!(m.solver_flags & CM_SOLVER_REFERENCE_ITERATES)) {                    // uniform
        // Warm start: the classical plane-stress return map in the plane's coordinates (a legal optimisation,
        // the iteration below stays the definition of the result).  At the solution r_a = r_b = 0 give p = 1 / (1 + g),
        // q = t / (1 + g), the sigma_33 row is LINEAR in t for a given g,
        //     t(g) = -(z:a / (1 + g) + Kz tr0) / (z:b / (1 + g) + Kz tr z) ,
        // and what is left is one scalar equation in g = dgam 3 mu / phi (g = 0: the elastic step, whose stretch is t(0)):
        //     F(g) = sqrt6 mu sqrt(S) / (1 + g) - Y - H(alpha_prev + g sqrt(S) / (k (1 + g))) ,  S = |a + t b|^2 ,  k = 3 / sqrt6 .
        // ~80 instead of ~150 instructions per step (two of them in float), and an elastic point needs none.  The loop below then starts at the mapped
        // state: it evaluates the plane's residual there and applies the reference's convergence test (relative tolerance against
        // ||C(x_prev)||, computed here from the state at x_prev), so a converged map costs one evaluation; otherwise the plane's
        // Newton continues from it (the Jacobian maps the plane to itself wherever the iterate sits in it).
        const double tp = t;
        bool plastic_prev, conv_prev;
        const Hard h0 = hardening(m, alpha_p);
        {   // ||C(x_prev)||^2 = C6^2 + C7^2 (r_a = r_b = 0 at c = 0, dgam = 0)
            const double P0 = aa + 2.0 * tp * ab + tp * tp * bb;
            const double phi0 = sqrt6mu * ((P0 > 0.0) ? P0 * rsqrt_pos(P0) : 0.0);
            const double f0 = (phi0 - (m.Y + h0.H)) * i2mu;
            plastic_prev = (f0 > m.yield_tol) || (fabs(f0) < m.yield_tol);
            const double c6 = plastic_prev ? f0 : 0.0;
            const double c7 = za + tp * zb + Kz * (tr0 + tp * trz);
            rel2 = m.rel_tol * m.rel_tol * (c6 * c6 + c7 * c7);
            first = false;
            conv_prev = (c6 * c6 + c7 * c7) < abs2;
        }
        // The map in c = g / (1 + g) (= c_a of the plane): 1 / (1 + g) = 1 - c, so with w = 1 - c
        //     t(c) = -(z:a w + Kz tr0) / (z:b w + Kz tr z) ,  phi = sqrt6 mu R w ,  dgam = c R / (3 / sqrt6) ,  R = sqrt(S(t(c)))
        // -- phi ~ phi(0) (1 - c) and dgam ~ c phi(0) / 3 mu are nearly linear in c (in g, phi ~ 1 / (1 + g) is strongly convex and
        // Newton from g = 0 creeps: 6-7 steps against 3-4), and no division by 1 + g is left.  ~80 instructions per step.
        constexpr double ic32 = 0.816496580927726;                  // sqrt6 / 3
        const double n0 = Kz * tr0, d0 = Kz * trz;
        double c = 0.0;
        // (a state that already passes the reference's test at x_prev is returned as it is, 0 iterations: re-applying a strain
        // to its own result changes nothing, bit for bit)
        bool done = !lane_valid || conv_prev, ok = false;
        {   // c = 0: the elastic step (stretch t(0)); the hardening there is the one at alpha_prev
            const double iD = rcp(zb + d0), N = za + n0;
            const double tn = -N * iD, dt = (za * (zb + d0) - N * zb) * (iD * iD);
            const double S = aa + 2.0 * tn * ab + tn * tn * bb, dS = 2.0 * (ab + tn * bb) * dt;
            const double rS = (S > 0.0) ? rsqrt_pos(S) : 0.0, R = S * rS, dR = 0.5 * dS * rS;
            const double F = sqrt6mu * R - (m.Y + h0.H), dF = sqrt6mu * (dR - R) - h0.dH * (R * ic32);
            // Elastic step: taken only when the state at x_prev -- with the OLD stretch -- is on the elastic branch too: a point
            // that is plastic there but elastic once the stretch relaxes has two roots of the reference's residual (the elastic
            // one and one on the plastic branch with dgam < 0), and which of them the reference's iteration from x_prev ends in
            // is decided by its iterates -- such a lane starts at x_prev and retraces them.
            if (!done && !(F > 0.0)) { done = true; ok = !plastic_prev; }
            if (!done) c = fmin(fmax(-F * rcp(dF), 0.0), 0.999999);
        }
        if (__any(!done) && hardening_seed_ok(m)) {
            // two Newton steps in float from there (see cm_device.hpp, "single-precision seeds")
            const float zaf = (float)za, zbf = (float)zb, n0f = (float)n0, d0f = (float)d0, aaf = (float)aa, abf = (float)ab, bbf = (float)bb;
            const float s6f = (float)sqrt6mu, Yf = (float)m.Y, apf = (float)alpha_p;
            float cf = (float)c;
#pragma unroll
            for (int sit = 0; sit < 2; ++sit) {
                const float w = 1.0f - cf, N = zaf * w + n0f, D = zbf * w + d0f, iD = rcp_f(D);
                const float tn = -N * iD, dt = (zaf * D - N * zbf) * (iD * iD);
                const float S = aaf + 2.0f * tn * abf + tn * tn * bbf, dS = 2.0f * (abf + tn * bbf) * dt;
                const float rS = rsq_f(S), R = S * rS, dR = 0.5f * dS * rS;
                const HardF h = hardening_f(m, apf + cf * R * (float)ic32);
                const float F = s6f * R * w - (Yf + h.H), dF = s6f * (dR * w - R) - h.dH * ((R + cf * dR) * (float)ic32);
                const float cn = cf - F * rcp_f(dF);
                if (cn >= 0.0f && cn < 0.999999f) cf = cn;          // (a step that leaves the interval, or is not finite, is dropped)
            }
            if (!done) c = (double)cf;
        }
        for (int wit = 0; wit < 12; ++wit) {
            const double w = 1.0 - c, N = za * w + n0, D = zb * w + d0, iD = rcp(D);
            const double tn = -N * iD, dt = (za * D - N * zb) * (iD * iD);
            const double S = aa + 2.0 * tn * ab + tn * tn * bb, dS = 2.0 * (ab + tn * bb) * dt;
            const double rS = (S > 0.0) ? rsqrt_pos(S) : 0.0, R = S * rS, dR = 0.5 * dS * rS;
            const Hard h = hardening(m, alpha_p + c * R * ic32);
            const double YH = m.Y + h.H;
            const double F = sqrt6mu * R * w - YH, dF = sqrt6mu * (dR * w - R) - h.dH * ((R + c * dR) * ic32);
            const double res = fabs(F);
            if (!done && !(res < 1e300)) done = true;
            if (!done && res < 1e-14 * YH) { done = true; ok = true; }
            // a lane that has finished keeps what it finished with: its result must not depend on how long the other lanes of
            // its wavefront keep iterating (a batch gives the same bits however it is sliced into launches)
            if (!done) {
                c = fmin(fmax(c - F * rcp(dF), 0.0), 0.999999);
                if (res < 1e-8 * YH) { done = true; ok = true; }    // quadratic convergence: this iterate is converged to round-off
            }
            if (!__any(!done)) break;
        }
        if (ok) {
            const double w = 1.0 - c, tg = -(za * w + n0) * rcp(zb * w + d0);
            const double S = aa + 2.0 * tg * ab + tg * tg * bb, Rg = (S > 0.0) ? S * rsqrt_pos(S) : 0.0;
            ca = c; cb = tg * c; t = tg;
            alpha = alpha_p + c * Rg * ic32;
        }
    }
    double p, q, rP, phi, f, nsq;          // the evaluation at the current u: read after the loop (a stopped lane re-evaluates its unchanged u)
    bool plastic;
    for (;;) {
        // evaluation at u (every lane, as in newton_s)
        p = 1.0 - ca; q = t - cb;
        const double vp = p * aa + q * ab, vq = p * ab + q * bb, P = p * vp + q * vq;
        rP = (P > 0.0) ? rsqrt_pos(P) : 0.0;                            // 1 / sqrt(P): phi, 3 mu / phi and 1 / P from one rsq
        phi = sqrt6mu * (P * rP);
        hd = hardening(m, alpha);
        f = (phi - (m.Y + hd.H)) * i2mu;
        const double dgam = alpha - alpha_p;
        plastic = (f > m.yield_tol) || (fabs(f) < m.yield_tol);
        const double h = plastic ? 1.224744871391589 * rP : 0.0, g = dgam * h;        // 3 mu / phi = (3 / sqrt 6) / sqrt P
        const double ra = ca - g * p, rb = cb - g * q, c6 = plastic ? f : dgam;
        const double c7 = p * za + q * zb + Kz * (tr0 + t * trz);
        nsq = ra * ra * A2 + 2.0 * ra * rb * AB2 + rb * rb * B2 + c6 * c6 + c7 * c7;
        if (first) { rel2 = m.rel_tol * m.rel_tol * nsq; first = false; }      // ||C(x_prev)||^2: the relative tolerance's reference
        if (running && !isfinite(nsq)) { running = false; fallback = true; }
        if constexpr (LS) {
            // the full step is the search's first trial (alpha = 1): phi(1) <= phi(0) + c1 phi'(0), phi = ||C||^2 / 2, phi'(0) = -||C||^2
            if (running && it > 0 && !(0.5 * nsq <= 0.5 * nsq_from + m.ls_c1 * -nsq_from)) { running = false; fallback = true; }
        }
        const bool conv = (nsq < rel2) || (nsq < abs2);
        if (running && conv) { running = false; flags |= CM_STATUS_CONVERGED; }
        if (running && it >= m.max_iters) running = false;
        if (!__any(running)) break;
        if (running) {
            // B = d r_c / d c = (1 + g) I - (g / P) w v^T has B w = w, and v . (d r_c / d t) = 0:
            //   B^-1 (d r_c / d alpha) = -h w ,  B^-1 (d r_c / d t) = ((g / P) v_q w - g e_b) / (1 + g) ,  v . B^-1 r = v . r ,
            // so the Schur complement on (alpha, t) is  [ -(3 mu + H') / 2mu , h v_q ; -h (z:a p + z:b q) , d77 + z . ut ]
            // (first row [1, 0] on the elastic side) -- its (0,0) entry is the radial-return denominator.
            const double gP = g * rP * rP, i1g = rcp(1.0 + g);
            const double vr = vp * ra + vq * rb, sg = gP * vr;
            const double br0 = (ra + p * sg) * i1g, br1 = (rb + q * sg) * i1g;                  // B^-1 r_c
            const double ut0 = gP * vq * p * i1g, ut1 = (gP * vq * q - g) * i1g;               // B^-1 d r_c / d t
            const double s00 = plastic ? -(hd.dH * i2mu + h * h * P) : 1.0, s01 = h * vq;
            const double s10 = -h * (za * p + zb * q), s11 = d77 + (za * ut0 + zb * ut1);
            const double r0 = c6 + h * vr, r1 = c7 + (za * br0 + zb * br1);
            const double det = s00 * s11 - s01 * s10;
            if (!(fabs(det) > 1e-300)) flags |= CM_STATUS_SINGULAR;
            const double idet = rcp(det);
            const double dal = (r0 * s11 - s01 * r1) * idet, dt = (s00 * r1 - s10 * r0) * idet;
            ca -= br0 + h * p * dal - ut0 * dt;
            cb -= br1 + h * q * dal - ut1 * dt;
            alpha -= dal;
            t -= dt;
            nsq_from = nsq;
            ++it;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = xp[k] + ca * a[k] + cb * b[k];
    x[6] = alpha;
    x[7] = 1.0 + t;
    uint32_t st = flags | (uint32_t)it;
    // The evaluation at the returned state, which the reverse sweep needs, in plane coordinates (the values of the last loop
    // evaluation): e_dev = p a + q b, s = 2 mu e_dev + K tr(e) I, gt_k = 3 mu w_k e_dev_k / phi.  The plane's ||C|| and the
    // 8-dof residual norm at the rounded state differ by round-off (the latter subtracts x - x_prev in the full strain's
    // precision: up to ~1e-4 relative at norms near 1e-14), and a lane that stopped within that distance of the tolerance must
    // behave exactly like the general path (e.g. re-applying the same strain to the returned state is a 0-iteration step):
    // a lane within 1 % of the tolerance (a few in 10^4) takes the general path.
    const double tre = tr0 + t * trz, twomu = 2.0 * m.mu, Ktre = (m.lambda + (2.0 / 3.0) * m.mu) * tre;
    const double rho = rP * rcp(sqrt6mu);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double ed = p * a[k] + q * b[k];
        ev.e[k] = ed + (kDiag[k] ? tre * (1.0 / 3.0) : 0.0);
        ev.s[k] = twomu * ed + (kDiag[k] ? Ktre : 0.0);
        ev.y.gt[k] = (3.0 * m.mu * rho) * (kW[k] * ed);
    }
    ev.tr = tre; ev.f = f; ev.dgam = alpha - alpha_p; ev.plastic = plastic; ev.hd = hd;
    ev.y.phi = phi; ev.y.rho = rho;
    if ((flags & CM_STATUS_CONVERGED) && !((nsq < 0.99 * rel2) || (nsq < 0.99 * abs2))) fallback = true;
    if (__any(fallback)) {
        if (fallback) { CM_COUNT_FALLBACK(); st = newton_s<CM_YIELD_J2, LS, CM_PLANE_STRESS>(m, eg, xp, x, lane_valid, ev, stage, z); }
    }
    return st;
}

// ---- J2 / FULL_3D, parameter gradient of a converged step in closed form ----------------------------------------------
// pbar of reverse_point_s for xin = NULL (no incoming state cotangent) at a CONVERGED J2 state, without the transposed solve:
// the return map is  e = (eg - v_prev) - dgam N,  N the trial normal (independent of the parameters),  dgam the root of
// f = phi_trial - 3 mu dgam - Y - H(alpha_prev + dgam); so with J = sum_k sbm_k s_k, s = lambda tr(e) d + 2 mu e:
//     dJ/dlambda = tr(e) (sbm . d)
//     dJ/dmu     = 2 sbm . e - 2 mu (sbm . N) d dgam / d mu ,  d dgam / d mu = (phi_trial / mu - 3 dgam) / (3 mu + H') = (phi / mu) / (3 mu + H')
//     dJ/dY      = L ,  dJ/dS = L (1 - exp(-D alpha)) ,  dJ/dD = L S alpha exp(-D alpha) ,  dJ/dK = L alpha ,  L = 2 mu (sbm . N) / (3 mu + H')
// This is the same derivative the IFT rule gives (A^-T applied to the stress cotangent, then the dC/dp contraction) evaluated
// on the solution manifold f = 0: the fused kernels use it when every point of the wavefront converged and fall back to
// reverse_point_s otherwise (an unconverged state is differentiated as the reference does it, through A(x) at that state).
// ~45 instead of ~250 instructions.
CM_D void reverse_j2_radial(const cm_model_desc& m, const double eg[6], const double* x, const double sbm[6],
                            const EvalS<CM_YIELD_J2>& ev, double* pbar) {
    double sbe = 0.0, sbn = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        sbe += sbm[k] * (eg[k] - x[k]);
        sbn += sbm[k] * (ev.y.gt[k] * kIW[k]);
    }
    const double etr = (eg[0] - x[0]) + (eg[3] - x[3]) + (eg[5] - x[5]);
    const double L = ev.plastic ? 2.0 * m.mu * sbn * rcp(3.0 * m.mu + ev.hd.dH) : 0.0;
    pbar[CM_P_LAMBDA] = (sbm[0] + sbm[3] + sbm[5]) * etr;
    pbar[CM_P_MU] = 2.0 * sbe - L * ev.y.phi * (2.0 * half_over_mu(m));
    pbar[CM_P_Y] = L;
    pbar[CM_P_VOCE_S] = m.has_voce ? L * (1.0 - ev.hd.expo) : 0.0;
    pbar[CM_P_VOCE_D] = m.has_voce ? L * m.voce_S * x[6] * ev.hd.expo : 0.0;
    pbar[CM_P_LIN_K] = m.has_linear ? L * x[6] : 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) pbar[CM_P_YC0 + j] = 0.0;
}

// ---- J2 / PLANE_STRESS, parameter gradient of a converged step from the 4x4 system of the plane ---------------------------
// The same quantity as reverse_point_s<J2, ., PLANE_STRESS> returns in pbar for xin = NULL at a CONVERGED state, by the IFT rule
// applied to the four residuals r(u, p) of newton_j2_plane instead of the 8 of C:  dJ/dp = dJ/dp|_u - lam . dr/dp,
// (dr/du)^T lam = dJ/du.  With e_dev = p a + q b (read off the state), J = sum_k sbm_k s_k = 2 mu (sbm . e_dev) + Kb tr(e) (sbm . d):
//   * r_a, r_b do not depend on the parameters (g = dgam 3 mu / phi = dgam (sqrt 6 / 2) / |e_dev|), so only lam_6, lam_7 are
//     needed, and they follow from the transposed 2x2 Schur complement of the plane's Newton step (same entries s00 .. s11):
//         S^T (lam_6, lam_7) = ( -2 mu h (sbm . e_dev) ,  2 mu (sbm . b) + Kb tr z (sbm . d) + 2 mu ((g/P) v_q (sbm . e_dev) - g (sbm . b)) / (1 + g) )
//   * dr_6/dp = (phi / 2mu^2, -1/2mu, -(1 - e^{-D alpha})/2mu, -S alpha e^{-D alpha}/2mu, -alpha/2mu) for (mu, Y, S, D, K) at f = 0,
//     dr_7/dlambda = tr z tr(e) / 2mu,  dr_7/dmu = -lambda tr z tr(e) / 2mu^2.
// The fused kernels use it when every point of the wavefront converged (reverse_point_s otherwise).  ~110 instead of ~330
// instructions.
CM_D void reverse_j2_plane(const cm_model_desc& m, const double eg[6], const double* z, const double* x, const double sbm[6],
                           const EvalS<CM_YIELD_J2>& ev, double* pbar) {
    double ed[6], b[6];
    const double t = x[7] - 1.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) ed[k] = (eg[k] - x[k]) + t * z[k];
    const double tre = ed[0] + ed[3] + ed[5], trz = z[0] + z[3] + z[5];
    double P = 0.0, vq = 0.0, zb = 0.0, ze = 0.0, se = 0.0, sb = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (kDiag[k]) ed[k] -= tre * (1.0 / 3.0);
        b[k] = z[k] - (kDiag[k] ? trz * (1.0 / 3.0) : 0.0);
        P += kW[k] * ed[k] * ed[k]; vq += kW[k] * ed[k] * b[k]; zb += kW[k] * z[k] * b[k]; ze += kW[k] * z[k] * ed[k];
        se += sbm[k] * ed[k]; sb += sbm[k] * b[k];
    }
    const double sbd = sbm[0] + sbm[3] + sbm[5];
    const double i2mu = half_over_mu(m), twomu = 2.0 * m.mu, Kb = m.lambda + (2.0 / 3.0) * m.mu;
    const double rP = (P > 0.0) ? rsqrt_pos(P) : 0.0;
    const double h = ev.plastic ? 1.224744871391589 * rP : 0.0, g = ev.dgam * h, gP = g * rP * rP, i1g = rcp(1.0 + g);
    const double d77 = zb + Kb * i2mu * trz * trz;
    const double s00 = ev.plastic ? -(ev.hd.dH * i2mu + h * h * P) : 1.0, s01 = h * vq;
    const double s10 = -h * ze, s11 = d77 + (gP * vq * ze - g * zb) * i1g;
    const double r6 = -twomu * h * se, r7 = twomu * sb + Kb * trz * sbd + twomu * (gP * vq * se - g * sb) * i1g;
    const double idet = rcp(s00 * s11 - s01 * s10);
    const double l6 = (r6 * s11 - s10 * r7) * idet, l7 = (s00 * r7 - s01 * r6) * idet;
    const double L = l6 * i2mu, tz = trz * tre * i2mu;
    pbar[CM_P_LAMBDA] = tre * sbd - l7 * tz;
    pbar[CM_P_MU] = 2.0 * se + (2.0 / 3.0) * tre * sbd - L * ev.y.phi * (2.0 * i2mu) + l7 * tz * m.lambda * (2.0 * i2mu);
    pbar[CM_P_Y] = L;
    pbar[CM_P_VOCE_S] = m.has_voce ? L * (1.0 - ev.hd.expo) : 0.0;
    pbar[CM_P_VOCE_D] = m.has_voce ? L * m.voce_S * x[6] * ev.hd.expo : 0.0;
    pbar[CM_P_LIN_K] = m.has_linear ? L * x[6] : 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) pbar[CM_P_YC0 + j] = 0.0;
}

#if defined(CM_HNN_VARIANT) && CM_HNN_VARIANT
#define CM_HNN_BUILD_HAS_SUBSPACE false
#else
#define CM_HNN_BUILD_HAS_SUBSPACE true
#endif
// which (def_type, yield, line search) combinations have a Newton iteration restricted to its invariant subspace
// (newton_j2_line, newton_j2_plane): the launchers pick the RL = true kernel variants for them unless CM_SOLVER_GENERAL_NEWTON
template <int DEF, int YK, bool LS>
constexpr bool has_j2_subspace() {
    // (the HNN build of the library -- network hardening law, a rarely used configuration -- keeps only the general kernels)
    return CM_HNN_BUILD_HAS_SUBSPACE && YK == CM_YIELD_J2 && (DEF == CM_FULL_3D || DEF == CM_PLANE_STRESS);
}
// what the launchers test to pick the RL = true kernel variants (UNIAXIAL_STRESS needs no variant: its kernels always take the
// 9 x 9 Newton step through the 4 x 4 form, uniaxial_solve in cm_device.hpp, which is the same step)
template <int DEF, int YK, bool LS>
constexpr bool has_fast_newton() { return has_j2_subspace<DEF, YK, LS>(); }
// Hill / Hosford under FULL_3D: the reference's Newton started at a scalar return map (hill_warm_start) / the analytic warm start
// (hosford_warm_start) instead of x_prev.  No kernel variant of their own: the same kernels read the description's switch
// (warm_start_on, wave-uniform), so CM_SOLVER_REFERENCE_ITERATES costs a branch, not a second set of kernels.
template <int DEF, int YK>
constexpr bool has_warm_start() {
    return CM_HNN_BUILD_HAS_SUBSPACE && ((DEF == CM_FULL_3D && (YK == CM_YIELD_HILL || YK == CM_YIELD_HOSFORD)) ||
                                         (DEF == CM_PLANE_STRESS && YK == CM_YIELD_HILL));
}
CM_D bool warm_start_on(const cm_model_desc& m) {
    return !(m.solver_flags & (CM_SOLVER_GENERAL_NEWTON | CM_SOLVER_REFERENCE_ITERATES)) && !(m.ls_max_evals > 0 && m.ls_kind == CM_LS_LEGACY);
}
template <int DEF, bool LS>
CM_D uint32_t newton_j2_sub(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, double* x,
                            bool lane_valid, EvalS<CM_YIELD_J2>& ev, LaneStage stage = LaneStage{nullptr, 0}) {
    static_assert(has_j2_subspace<DEF, CM_YIELD_J2, LS>(), "J2 subspace Newton: FULL_3D and PLANE_STRESS");
    if constexpr (DEF == CM_FULL_3D) return newton_j2_line<LS>(m, eg, xp, x, lane_valid, ev, stage);
    else return newton_j2_plane<LS>(m, eg, z, xp, x, lane_valid, ev, stage);
}
// HOST side: does this description run the RL = true variants (what has_fast_newton<> offers), resp. the warm starts?  The J2
// subspace iterations treat a full step as the Armijo search's first trial; under the legacy backtracking (CM_LS_LEGACY) the
// acceptance test is another one, so those configurations run the general path.  For J2 CM_SOLVER_REFERENCE_ITERATES is read
// inside newton_j2_plane (the subspace form stays); for Hill / Hosford it switches the warm start off (warm_start_on).
inline bool use_fast_newton(const cm_model_desc* m) {
    if ((m->solver_flags & CM_SOLVER_GENERAL_NEWTON) || (m->ls_max_evals > 0 && m->ls_kind == CM_LS_LEGACY)) return false;
    return m->yield_kind == CM_YIELD_J2 || !(m->solver_flags & CM_SOLVER_REFERENCE_ITERATES);
}
// the RL = true kernel variants' solver: what has_fast_newton<> promises for (DEF, YK); `ev` holds the evaluation at the returned x
template <int DEF, int YK, bool LS>
CM_D uint32_t newton_fast(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, double* x,
                          bool lane_valid, EvalS<YK>& ev, LaneStage stage = LaneStage{nullptr, 0}) {
    static_assert(has_fast_newton<DEF, YK, LS>(), "no subspace iteration for this configuration");
    return newton_j2_sub<DEF, LS>(m, eg, z, xp, x, lane_valid, ev, stage);
}
// the structured solve of every other kernel: newton_s, started at the warm start where the configuration has one and the
// description leaves it on
template <int DEF, int YK, bool LS>
CM_D uint32_t newton_s_warm(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, double* x,
                            bool lane_valid, EvalS<YK>& ev, LaneStage stage = LaneStage{nullptr, 0}) {
    if constexpr (has_warm_start<DEF, YK>()) {
        constexpr int NX = Dims<DEF>::NX;
        double x0[NX], n0sq = 0.0;
        bool warm = false;
        if (warm_start_on(m)) {                                     // uniform
            if constexpr (DEF == CM_PLANE_STRESS) warm = hill_ps_warm_start(m, eg, z, xp, x0, n0sq, lane_valid);
            else if constexpr (YK == CM_YIELD_HILL) warm = hill_warm_start(m, eg, xp, x0, n0sq, lane_valid);
            else warm = hosford_warm_start(m, eg, xp, x0, n0sq, lane_valid);
        } else {
#pragma unroll
            for (int k = 0; k < NX; ++k) x0[k] = xp[k];
        }
        return newton_s<YK, LS, DEF>(m, eg, xp, x, lane_valid, ev, stage, z, PlainNorm{}, x0, warm ? n0sq : -1.0);
    }
    else return newton_s<YK, LS, DEF>(m, eg, xp, x, lane_valid, ev, stage, z);
}

// ---- reverse sweep, structured (same contract as cm::reverse_point; FULL_3D and PLANE_STRESS) ---------------
template <int YK, bool HAVE_EV = false, int DEF = CM_FULL_3D>
CM_D bool reverse_point_s(const cm_model_desc& m, const double eg[6], const double* x, const double* xp,
                          const double sbm[6], const double* xin, double* pbar, double* xpbar, double* egbar,
                          EvalS<YK>* evp = nullptr, const double* z = nullptr, double* lam_out = nullptr) {
    constexpr int NX = Dims<DEF>::NX;
    constexpr bool PS = (DEF == CM_PLANE_STRESS);
    EvalS<YK> evl;
    double C[NX], lam[NX];
    if constexpr (!HAVE_EV) residual_s<YK, DEF>(m, eg, z, x, xp, evl, C);
    const EvalS<YK>& ev = HAVE_EV ? *evp : evl;
    // strain and stress are rebuilt from (eg, x) here rather than read from `ev`: 6 subtractions instead of
    // 12-24 registers carried across the Newton loop of the fused kernels
    double ee[6], ss[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        ee[k] = eg[k] - x[k];
        if constexpr (PS) ee[k] += (x[7] - 1.0) * z[k];
    }
    const double etr = ee[0] + ee[3] + ee[5];
#pragma unroll
    for (int k = 0; k < 6; ++k) ss[k] = 2.0 * m.mu * ee[k] + (kDiag[k] ? m.lambda * etr : 0.0);
    PlasticOpFor<YK> op;
    op_build<YK>(m, ev, op);
    double csb[6];
    apply_cel(m, sbm, csb);
#pragma unroll
    for (int k = 0; k < 6; ++k) lam[k] = -csb[k];
    lam[6] = 0.0;
    if constexpr (PS) lam[7] = dot<6>(z, csb);                   // d s / d F33 = Cel z
    if (xin) {
#pragma unroll
        for (int k = 0; k < NX; ++k) lam[k] += xin[k];
    }
    const bool ok = solve_s<DEF, true>(m, op, ev, z, lam, lam);
    if (lam_out) {
#pragma unroll
        for (int k = 0; k < NX; ++k) lam_out[k] = lam[k];
    }
    const double i2mu = half_over_mu(m);
    const YieldS<YK>& y = ev.y;
    double u[6], hu[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = ev.plastic ? (-ev.dgam * lam[k] * kIW[k]) : 0.0;
    hess_apply<YK>(m, y, u, hu);
    const double lam6 = ev.plastic ? lam[6] : 0.0;
    if (pbar) {
        const double ge = dot<6>(y.gt, ee), hue = dot<6>(hu, ee);
        const double sbd = sbm[0] + sbm[3] + sbm[5], sbe = dot<6>(sbm, ee);
        // lam[0:7] . dC/dlambda = 0 for a pressure-independent surface (Ht d = 0, gt . d = 0)
        double cl = 0.0, cm_ = 2.0 * hue + lam6 * (2.0 * ge * i2mu - ev.f * 2.0 * i2mu);
        if constexpr (PS) {
            const double zt = z[0] + z[3] + z[5];
            double zwe = 0.0, zws = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { zwe += kW[k] * z[k] * ee[k]; zws += kW[k] * z[k] * ss[k]; }
            cl = lam[7] * zt * etr * i2mu;
            cm_ += lam[7] * (2.0 * zwe * i2mu - zws * i2mu * 2.0 * i2mu);      // C7 = (w o z) . s / 2mu
        }
        pbar[CM_P_LAMBDA] = sbd * etr - cl;
        pbar[CM_P_MU] = 2.0 * sbe - cm_;
        pbar[CM_P_Y] = lam6 * i2mu;
        pbar[CM_P_VOCE_S] = m.has_voce ? lam6 * (1.0 - ev.hd.expo) * i2mu : 0.0;
        pbar[CM_P_VOCE_D] = m.has_voce ? lam6 * m.voce_S * x[6] * ev.hd.expo * i2mu : 0.0;
        pbar[CM_P_LIN_K] = m.has_linear ? lam6 * x[6] * i2mu : 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) pbar[CM_P_YC0 + j] = 0.0;
        if constexpr (YK == CM_YIELD_HILL) {
            if (ev.plastic) {
                const double* s = ss;
                const double ip = y.rho;
                const double d12 = s[3] - s[5], d20 = s[5] - s[0], d01 = s[0] - s[3];
                const double qj[6] = {d12 * d12, d20 * d20, d01 * d01, 2.0 * s[4] * s[4], 2.0 * s[2] * s[2], 2.0 * s[1] * s[1]};
                const double uAs[6] = {(u[3] - u[5]) * d12, (u[5] - u[0]) * d20, (u[0] - u[3]) * d01,
                                       2.0 * u[4] * s[4], 2.0 * u[2] * s[2], 2.0 * u[1] * s[1]};
                const double ug = dot<6>(u, y.gt);
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    pbar[CM_P_YC0 + j] = -(uAs[j] * ip - ug * qj[j] * 0.5 * ip * ip + lam6 * qj[j] * 0.5 * ip * i2mu);
            }
        }
    }
    if (xpbar) {
#pragma unroll
        for (int k = 0; k < 6; ++k) xpbar[k] = lam[k];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += y.gt[k] * kIW[k] * lam[k];
        xpbar[6] = ev.plastic ? -s : lam[6];
        if constexpr (PS) xpbar[7] = 0.0;
    }
    if (egbar) {
        double t[6], ct[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            t[k] = hu[k] + lam6 * y.gt[k] * i2mu;
            if constexpr (PS) t[k] += lam[7] * kW[k] * z[k] * i2mu;
        }
        apply_cel(m, t, ct);
#pragma unroll
        for (int k = 0; k < 6; ++k) egbar[k] = csb[k] - ct[k];
    }
    return op.ok && ok;
}

// ---- forward tangent, structured (same contract as cm::tangent_point; FULL_3D and PLANE_STRESS) -------------
template <int YK, int DEF = CM_FULL_3D>
CM_D bool tangent_point_s(const cm_model_desc& m, const double eg[6], const double* x, const double* xp, double (&T)[6][6],
                          const double* z = nullptr) {
    constexpr int NX = Dims<DEF>::NX;
    constexpr bool PS = (DEF == CM_PLANE_STRESS);
    EvalS<YK> ev;
    double C[NX];
    residual_s<YK, DEF>(m, eg, z, x, xp, ev, C);
    PlasticOpFor<YK> op;
    op_build<YK>(m, ev, op);
    const YieldS<YK>& y = ev.y;
    bool ok = op.ok;
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        double b[NX], unit[6], hcol[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) unit[k] = (k == l) ? 1.0 : 0.0;
        hess_apply<YK>(m, y, unit, hcol);                             // column l of Ht
        // b = -dC/deg_l = [ beta/w_k Ht_kl ; -gt_l ]  (plastic), 0 (elastic)
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = ev.plastic ? op.beta * kIW[k] * hcol[k] : 0.0;
        b[6] = ev.plastic ? -y.gt[l] : 0.0;
        if constexpr (PS) {                                           // -dC7/deg_l = -(Cel (w o z))_l / 2mu
            const double zt = z[0] + z[3] + z[5];
            b[7] = -(kW[l] * z[l] + (kDiag[l] ? m.lambda * half_over_mu(m) * zt : 0.0));
        }
        ok = solve_s<DEF, false>(m, op, ev, z, b, b) && ok;
        double de[6], ds[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            de[k] = ((k == l) ? 1.0 : 0.0) - b[k];
            if constexpr (PS) de[k] += z[k] * b[7];
        }
        apply_cel(m, de, ds);
#pragma unroll
        for (int r = 0; r < 6; ++r) T[r][l] = ds[r];
    }
    return ok;
}

// ---- front doors: structured for FULL_3D / PLANE_STRESS (every yield surface is pressure independent); the dense
// unrolled LU for UNIAXIAL_STRESS and the rate form (STRUCT = false forces the dense path) -------------------------------------------------------------------
template <int DEF, int YK>
constexpr bool has_structured() {
    return DEF == CM_FULL_3D || DEF == CM_PLANE_STRESS;
}
// RL: the J2 radial-line iteration (default; cm_model_desc.solver_flags & CM_SOLVER_GENERAL_NEWTON disables it); a compile-time
// variant chosen by the launcher so that the default kernels do not carry its code and registers.
template <int DEF, int YK, bool LS, bool STRUCT = true, bool RL = false>
CM_D uint32_t newton_any(const cm_model_desc& m, const double eg[6], const double z[6], const double* xp, double* x, bool valid,
                         LaneStage stage = LaneStage{nullptr, 0}) {
    if constexpr (STRUCT && has_structured<DEF, YK>()) {
        EvalS<YK> ev;
        if constexpr (RL) return newton_fast<DEF, YK, LS>(m, eg, z, xp, x, valid, ev, stage);
        else return newton_s_warm<DEF, YK, LS>(m, eg, z, xp, x, valid, ev, stage);
    }
    else return newton<DEF, YK, CM_SMALL_ELASTIC_PLASTIC, LS, STRUCT>(m, eg, z, xp, x, valid);   // STRUCT = false: the dense reference path
}
// ---- rate-form model on the structured solver -----------------------------------------------------------------------------------
// cmad/models/small_rate_elastic_plastic.py:249-346: unknown x = [sigma(6), alpha (, F33)], residual
//   C_sigma = (sigma - sigma_prev - Cel (deg + (F33 - F33_prev) z) + dgam Cel n) / 2mu ,  C_6 = f or dgam ,  C_7 = (w o z) . (Cel de - dgam Cel n) / 2mu.
// With v defined by  sigma = Cel (E0 + (F33 - F33_prev) z - v),  E0 = Cel^-1 sigma_prev + deg,  this is
//   C_sigma = -Cel (v - dgam n) / 2mu ,  C_7 = (w o z) . sigma / 2mu - (w o z) . sigma_prev / 2mu - (w o z) . C_sigma :
// an affine change of variables and a CONSTANT nonsingular linear map of the total-form residual [v - dgam n, C_6, (w o z) . sigma / 2mu]
// of a point with strain E0, previous plastic strain 0 and previous stretch 1.  Newton's iterates are invariant under both, so the
// rate-form iteration started at sigma_prev is the total-form iteration in v started at v = deg -- same Jacobian, same structured
// solve (3x3 + shear, dense-surface 6x6), every yield surface -- with the convergence test and the line search's merit taken on the
// mapped residual (RateNorm).  Same iterates and iteration counts as cm::newton<DEF, YK, RATE> (host tests run both against the
// oracle); ~2.5x (FULL_3D) to 4x (PLANE_STRESS) fewer instructions than the dense 7x7 / 8x8 path.
template <int DEF>
struct RateNorm {
    double l2;             // lambda / 2mu
    const double* z;       // PLANE_STRESS: V(q3 q3^T)
    double c7_shift;       // (w o z) . sigma_prev / 2mu
    template <int NX> CM_D double sq(const double* C) const {
        const double t = l2 * (C[0] + C[3] + C[5]);
        double n = C[6] * C[6], zc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double cs = C[k] + (kDiag[k] ? t : 0.0);              // -(C_sigma)_k = (Cel C_v)_k / 2mu
            n += cs * cs;
            if constexpr (DEF == CM_PLANE_STRESS) zc += kW[k] * z[k] * cs;
        }
        if constexpr (DEF == CM_PLANE_STRESS) { const double c7 = C[7] - c7_shift + zc; n += c7 * c7; }
        return n;
    }
};
template <int YK, bool LS, int DEF>
CM_D uint32_t newton_s_rate(const cm_model_desc& m, const double deg[6], const double* z, const double* xp, double* x,
                            bool lane_valid, LaneStage stage = LaneStage{nullptr, 0}) {
    constexpr int NX = Dims<DEF>::NX;
    constexpr bool PS = (DEF == CM_PLANE_STRESS);
    const double i2mu = half_over_mu(m), trs = xp[0] + xp[3] + xp[5];
    const double cc = m.lambda * rcp(3.0 * m.lambda + 2.0 * m.mu);      // Cel^-1 s = (s - cc tr(s) d) / 2mu
    double E0[6], v0[NX], vp[NX], v[NX];
    double shift = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        E0[k] = (xp[k] - (kDiag[k] ? cc * trs : 0.0)) * i2mu + deg[k];
        vp[k] = 0.0; v0[k] = deg[k];
        if constexpr (PS) shift += kW[k] * z[k] * xp[k];
    }
    vp[6] = v0[6] = xp[6];
    if constexpr (PS) { vp[7] = 1.0; v0[7] = 1.0; }
    const RateNorm<DEF> norm{m.lambda * i2mu, z, shift * i2mu};
    EvalS<YK> ev;
    const uint32_t st = newton_s<YK, LS, DEF, RateNorm<DEF>>(m, E0, vp, v, lane_valid, ev, stage, z, norm, v0);
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = ev.s[k];
    x[6] = v[6];
    if constexpr (PS) x[7] = xp[7] + (v[7] - 1.0);
    return st;
}
// A_rate^-T through the structured solver: A_rate = T A_tot P^-1 with the constant maps of the comment above
// (P = d(sigma, alpha, F33) / d(v, alpha, F33) = [[-Cel, 0, Cel z], [0, 1, 0], [0, 0, 1]],  T = [[-Cel / 2mu, 0, 0], [0, 1, 0],
// [(w o z)^T Cel / 2mu, 0, 1]]), so  lam = T^-T A_tot^-T P^T b :  b_v = -Cel b_sigma, b_7 += (Cel z) . b_sigma ;  mu = A_tot^-T b ;
// lam_sigma = -2mu Cel^-1 mu_v + (w o z) mu_7.
template <int YKS>
struct StructRateSolveT {
    template <int DEF, int YK>
    CM_D bool apply(const cm_model_desc& m, const double z[6], const double* x, const double* xp, const Eval<DEF>&, const double (*)[6],
                    double* lam) const {
        static_assert(YK == YKS, "one yield surface per instantiation");
        constexpr int NX = Dims<DEF>::NX;
        constexpr bool PS = (DEF == CM_PLANE_STRESS);
        const double i2mu = half_over_mu(m), cc = m.lambda * rcp(3.0 * m.lambda + 2.0 * m.mu);
        // the structured evaluation at this state: a total-form point with elastic strain Cel^-1 sigma
        const double trs = x[0] + x[3] + x[5];
        double ee[6], xi[NX], xip[NX], Cd[NX], b[NX], mu[NX];
#pragma unroll
        for (int k = 0; k < 6; ++k) { ee[k] = (x[k] - (kDiag[k] ? cc * trs : 0.0)) * i2mu; xi[k] = 0.0; xip[k] = 0.0; }
        xi[6] = x[6]; xip[6] = xp[6];
        if constexpr (PS) { xi[7] = 1.0; xip[7] = 1.0; }
        EvalS<YK> evs;
        residual_s<YK, DEF>(m, ee, z, xi, xip, evs, Cd);
        PlasticOpFor<YK> op;
        op_build<YK>(m, evs, op);
        double cb[6];
        apply_cel(m, lam, cb);                                   // Cel b_sigma
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = -cb[k];
        b[6] = lam[6];
        if constexpr (PS) b[7] = lam[7] + dot<6>(z, cb);         // (Cel z) . b_sigma = z . Cel b_sigma
        const bool ok = solve_s<DEF, true>(m, op, evs, z, b, mu);
        const double trm = mu[0] + mu[3] + mu[5];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            lam[k] = -(mu[k] - (kDiag[k] ? cc * trm : 0.0));     // -2mu Cel^-1 mu_v
            if constexpr (PS) lam[k] += kW[k] * z[k] * mu[7];
        }
        lam[6] = mu[6];
        if constexpr (PS) lam[7] = mu[7];
        return ok && op.ok;
    }
};
// A_rate^-1 = P A_tot^-1 T^-1 for several right-hand sides (the tangent):  b_v = -2mu Cel^-1 b_sigma, b_7 += (w o z) . b_sigma ;
// y = A_tot^-1 b ;  x_sigma = -Cel y_v + Cel z y_7.
template <int YKS, int DEFS>
struct StructRateSolve {
    EvalS<YKS> evs;
    PlasticOpFor<YKS> op;
    template <int DEF, int YK>
    CM_D bool setup(const cm_model_desc& m, const double*, const double z[6], const double* x, const double* xp) {
        static_assert(YK == YKS && DEF == DEFS, "one configuration per instantiation");
        constexpr int NX = Dims<DEF>::NX;
        const double i2mu = half_over_mu(m), cc = m.lambda * rcp(3.0 * m.lambda + 2.0 * m.mu), trs = x[0] + x[3] + x[5];
        double ee[6], xi[NX], xip[NX], Cd[NX];
#pragma unroll
        for (int k = 0; k < 6; ++k) { ee[k] = (x[k] - (kDiag[k] ? cc * trs : 0.0)) * i2mu; xi[k] = 0.0; xip[k] = 0.0; }
        xi[6] = x[6]; xip[6] = xp[6];
        if constexpr (DEF == CM_PLANE_STRESS) { xi[7] = 1.0; xip[7] = 1.0; }
        residual_s<YK, DEF>(m, ee, z, xi, xip, evs, Cd);
        op_build<YK>(m, evs, op);
        return op.ok;
    }
    template <int DEF, int YK>
    CM_D void solve(const cm_model_desc& m, const double* z, double (&b)[Dims<DEF>::NX]) const {
        constexpr int NX = Dims<DEF>::NX;
        constexpr bool PS = (DEF == CM_PLANE_STRESS);
        const double cc = m.lambda * rcp(3.0 * m.lambda + 2.0 * m.mu);
        const double trb = b[0] + b[3] + b[5];
        double t[NX], y[NX];
        double zb = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            t[k] = -(b[k] - (kDiag[k] ? cc * trb : 0.0));                // -2mu Cel^-1 b_sigma
            if constexpr (PS) zb += kW[k] * z[k] * b[k];
        }
        t[6] = b[6];
        if constexpr (PS) t[7] = b[7] + zb;
        solve_s<DEF, false>(m, op, evs, z, t, y);
        double u[6], cu[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) u[k] = -y[k] + (PS ? z[k] * y[NX - 1] : 0.0);
        apply_cel(m, u, cu);
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = cu[k];
        b[6] = y[6];
        if constexpr (PS) b[7] = y[7];
    }
};
template <int DEF, int YK, bool STRUCT = true>
CM_D bool tangent_rate_any(const cm_model_desc& m, const double deg[6], const double z[6], const double* x, const double* xp,
                           double (&T)[6][6]) {
    if constexpr (STRUCT && has_structured<DEF, YK>()) return tangent_point_rate<DEF, YK, StructRateSolve<YK, DEF>>(m, deg, z, x, xp, T);
    else return tangent_point_rate<DEF, YK>(m, deg, z, x, xp, T);
}

template <int DEF, int YK, bool STRUCT = true>
CM_D bool reverse_rate_any(const cm_model_desc& m, const double deg[6], const double z[6], const double* x, const double* xp,
                           const double sbm[6], const double* xin, double* pbar, double* xpbar, double* degbar, double* lam_out = nullptr) {
    if constexpr (STRUCT && has_structured<DEF, YK>())
        return reverse_point_rate<DEF, YK, StructRateSolveT<YK>>(m, deg, z, x, xp, sbm, xin, pbar, xpbar, degbar, lam_out);
    else return reverse_point_rate<DEF, YK>(m, deg, z, x, xp, sbm, xin, pbar, xpbar, degbar, lam_out);
}

// the rate-form model's Newton: the structured solver where the total form has one, cm::newton otherwise (STRUCT = false: always)
template <int DEF, int YK, bool LS, bool STRUCT = true>
CM_D uint32_t newton_rate_any(const cm_model_desc& m, const double deg[6], const double* z, const double* xp, double* x, bool valid,
                              LaneStage stage = LaneStage{nullptr, 0}) {
    if constexpr (STRUCT && has_structured<DEF, YK>()) return newton_s_rate<YK, LS, DEF>(m, deg, z, xp, x, valid, stage);
    else return newton<DEF, YK, CM_SMALL_RATE_ELASTIC_PLASTIC, LS>(m, deg, z, xp, x, valid);
}

template <int DEF, int YK, bool STRUCT = true>
CM_D bool reverse_any(const cm_model_desc& m, const double eg[6], const double z[6], const double* x, const double* xp,
                      const double sbm[6], const double* xin, double* pbar, double* xpbar, double* egbar,
                      double* lam_out = nullptr) {
    if constexpr (STRUCT && has_structured<DEF, YK>())
        return reverse_point_s<YK, false, DEF>(m, eg, x, xp, sbm, xin, pbar, xpbar, egbar, nullptr, z, lam_out);
    else return reverse_point<DEF, YK, STRUCT>(m, eg, z, x, xp, sbm, xin, pbar, xpbar, egbar, lam_out);   // STRUCT = false: the dense reference path
}
template <int DEF, int YK, bool STRUCT = true>
CM_D bool tangent_any(const cm_model_desc& m, const double eg[6], const double z[6], const double* x, const double* xp,
                      double (&T)[6][6]) {
    if constexpr (STRUCT && has_structured<DEF, YK>()) return tangent_point_s<YK, DEF>(m, eg, x, xp, T, z);
    else return tangent_point<DEF, YK, STRUCT>(m, eg, z, x, xp, T);
}


// ---- forward pass over a whole load history of one point (state carried in registers) ------------------------------
// cmad/objectives/mp_objective.py:62-89 (forward pass with storage) / cmad/cli/primal.py:129-176 without the QoI:
// xi_hist[(K+1)][NX] rows (slot 0 = initial state), sigma_hist[(K+1)][6] rows (global Cauchy stress; slot 0 = stress of
// the initial state under gradu_hist[0]) and status_hist[(K+1)] rows are optional (null = not stored).
template <int DEF, int YK, bool ROT, bool LS, int MK, bool RL = false, class IO>
CM_D void primal_history_point(const cm_model_desc& m, int K, const double* gradu_hist, const double* xi0, double* xi_hist,
                               double* sigma_hist, uint32_t* status_hist, bool valid, LaneStage stage, const IO& io) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU;
    constexpr bool RU = (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS);
    double x[NX], xp[NX], z[Dims<DEF>::NZ];
    strain_z<DEF, ROT>(m, z);
    io.template load<NX>(xi0, 0, x);
    for (int k = 0; k <= K; ++k) {
        double G[NU], eg[6];
        io.template load<NU>(gradu_hist, (int64_t)k * NU, G);
        uint32_t st = CM_STATUS_CONVERGED;
        if (k > 0) {
            if constexpr (RU) {
                double Gp[NU];
                io.template load<NU>(gradu_hist, (int64_t)(k - 1) * NU, Gp);
#pragma unroll
                for (int i = 0; i < NX; ++i) xp[i] = x[i];
                st = ru_newton<YK, LS>(m, G[0] - Gp[0], xp, x, valid);
            } else if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) {
                double Gp[NU], dG[NU], deg[6];
                io.template load<NU>(gradu_hist, (int64_t)(k - 1) * NU, Gp);
#pragma unroll
                for (int i = 0; i < NU; ++i) dG[i] = G[i] - Gp[i];
                strain_from_gradu<DEF, ROT>(m, dG, deg);
#pragma unroll
                for (int i = 0; i < NX; ++i) xp[i] = x[i];
                st = newton_rate_any<DEF, YK, LS>(m, deg, z, xp, x, valid, stage);
            } else {
                strain_from_gradu<DEF, ROT>(m, G, eg);
#pragma unroll
                for (int i = 0; i < NX; ++i) xp[i] = x[i];
                st = newton_any<DEF, YK, LS, true, RL>(m, eg, z, xp, x, valid, stage);
            }
        }
        if (!valid) continue;
        if (xi_hist) io.template store<NX>(xi_hist, (int64_t)k * NX, x);
        if (sigma_hist) {
            double sg[6];
            if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) to_global<ROT>(m, x, sg);
            else {
                Eval<DEF> ev;
                strain_from_gradu<DEF, ROT>(m, G, eg);
                strain_stress<DEF>(m, eg, z, x, ev);
                to_global<ROT>(m, ev.s, sg);
            }
            io.template store<6>(sigma_hist, (int64_t)k * 6, sg);
        }
        if (status_hist) io.store_status(status_hist, k, st);
    }
}

// ---- a whole load history of one point: K updates forward, K adjoint steps backward ---------------------------------
// cmad/objectives/mp_objective.py:62-89 (forward pass with storage) and :112-142 (adjoint recursion) with the QoI of
// cmad/qois/calibration.py:56-66, for one Gauss point.  The state stays in registers from step to step: per step the
// forward pass reads grad u and writes xi, the backward pass reads grad u, the previous xi and the data.
//   gradu_hist[(K+1)][NU] rows, data_hist[(K+1)][6] rows, xi_hist[(K+1)][NX] rows (slot 0 = initial state, filled here)
// IO supplies the row access (the kernels: SGPR-based SoA rows; the host build: plain indexing):
//   io.load<N>(base, step_row0, out) / io.store<N>(base, step_row0, v) / io.phase_barrier()
// red[0] += J, red[1 + j] += dJ/dp_j (KP order).  MK selects the total-form or the rate-form model (the latter takes
// grad u of the previous step as well).
// HistoryCotangents (cm_adjoint_history): instead of the calibration QoI the caller may hand in the QoI's cotangents per
// step -- sbar_hist[(K+1)][6] = dJ/d sigma (the 6 stored global entries) replaces wsq o (sigma - data) (red[0] stays 0),
// xibar_hist[(K+1)][NX] = explicit dJ/d xi is added to the incoming state cotangent -- and ask for the adjoint vector of
// every step, lam_hist[(K+1)][NX] (slot 0 unused; phi = -lam in cmad/objectives/mp_objective.py:112-142).
struct HistoryCotangents {
    const double* sbar_hist;
    const double* xibar_hist;
    double* lam_hist;
};

template <int DEF, int YK, bool ROT, bool LS, int MK, bool RL = false, class IO>
CM_D void history_point(const cm_model_desc& m, int K, const double* gradu_hist, const double* data_hist, const double wsq[6],
                        const double* xi0, double* xi_hist, bool valid, LaneStage stage, const IO& io, double* red,
                        HistoryCotangents hc = HistoryCotangents{nullptr, nullptr, nullptr}) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU;
    constexpr bool RU = (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS);
    double x[NX], xp[NX], z[Dims<DEF>::NZ];
    strain_z<DEF, ROT>(m, z);
    io.template load<NX>(xi0, 0, x);
    if (valid) io.template store<NX>(xi_hist, 0, x);
    // eg: the material strain (increment); under RU only eg[0] is used and holds the scalar increment of grad u
    auto strain_at = [&](int k, double eg[6]) {
        double G[NU];
        io.template load<NU>(gradu_hist, (int64_t)k * NU, G);
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) {
            double Gp[NU];
            io.template load<NU>(gradu_hist, (int64_t)(k - 1) * NU, Gp);
#pragma unroll
            for (int i = 0; i < NU; ++i) G[i] -= Gp[i];
        }
        if constexpr (RU) eg[0] = G[0];
        else strain_from_gradu<DEF, ROT>(m, G, eg);
    };
    for (int k = 1; k <= K; ++k) {
        double eg[6];
        strain_at(k, eg);
#pragma unroll
        for (int i = 0; i < NX; ++i) xp[i] = x[i];
        if constexpr (RU) ru_newton<YK, LS>(m, eg[0], xp, x, valid);
        else if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) newton_rate_any<DEF, YK, LS>(m, eg, z, xp, x, valid, stage);
        else newton_any<DEF, YK, LS, true, RL>(m, eg, z, xp, x, valid, stage);
        if (valid) io.template store<NX>(xi_hist, (int64_t)k * NX, x);
    }
    io.phase_barrier();          // tail lanes shadow another lane's point: its stored states must be visible to them
    double xin[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xin[i] = 0.0;
    for (int k = K; k >= 1; --k) {
        double eg[6], sd[6], sg[6], sb[6], sbm[6], pbar[CM_NUM_PARAMS], xpbar[NX], lam[NX];
        strain_at(k, eg);
        io.template load<NX>(xi_hist, (int64_t)(k - 1) * NX, xp);
        double J = 0.0;
        if (hc.sbar_hist) {                                     // the caller's QoI: cotangents given, objective on its side
            io.template load<6>(hc.sbar_hist, (int64_t)k * 6, sb);
        } else {
            io.template load<6>(data_hist, (int64_t)k * 6, sd);
            if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) to_global<ROT>(m, x, sg);
            else {
                Eval<DEF> ev;
                strain_stress<DEF>(m, eg, z, x, ev);
                to_global<ROT>(m, ev.s, sg);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double mm = sg[i] - sd[i];
                sb[i] = wsq[i] * mm;
                J += 0.5 * sb[i] * mm;
            }
        }
        if (hc.xibar_hist) {
            double xb[NX];
            io.template load<NX>(hc.xibar_hist, (int64_t)k * NX, xb);
#pragma unroll
            for (int i = 0; i < NX; ++i) xin[i] += xb[i];
        }
        cotangent_to_material<ROT>(m, sb, sbm);
        if constexpr (RU) ru_reverse<YK>(m, eg[0], x, xp, sb, xin, pbar, xpbar, nullptr, lam);
        else if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) reverse_rate_any<DEF, YK>(m, eg, z, x, xp, sbm, xin, pbar, xpbar, nullptr, lam);
        else reverse_any<DEF, YK>(m, eg, z, x, xp, sbm, xin, pbar, xpbar, nullptr, lam);
        if (hc.lam_hist && valid) io.template store<NX>(hc.lam_hist, (int64_t)k * NX, lam);
        red[0] += J;
#pragma unroll
        for (int j = 0; j < CM_NUM_PARAMS; ++j) red[1 + j] += pbar[j];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xin[i] = xpbar[i]; x[i] = xp[i]; }
    }
}

// ---- forward (direct) sensitivities over a stored history of one point (cm_direct_history) ---------------------------------
// cmad/objectives/mp_objective.py:158-215: dxi_k/dp = -A_k^-1 (dC_k/dp + dC_k/dxi_prev dxi_{k-1}/dp), dsigma_k/dp by the chain
// rule (cm::direct_point per step, the sensitivity block carried from step to step), and the gradient contraction
//   g_j = sum_k sbar_k . dsigma_k/dp_j + xibar_k . dxi_k/dp_j     (sbar_hist / xibar_hist may be null: no contraction).
// IO: io.get(base, row) / io.put(base, row, v) address row `row` of this point.  Row layouts as include/cmad_hip.h.
template <int DEF, int YK, bool ROT, int MK, class IO>
CM_D void direct_history_point(const cm_model_desc& m, int K, const double* gradu_hist, const double* xi_hist,
                               const double* sbar_hist, const double* xibar_hist, double* dx_dp_hist, double* ds_dp_hist,
                               const IO& io, double* g) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU, NP_ = CM_NUM_PARAMS;
    constexpr bool RU = (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS);
    double G[NU], Gp[NU], xp[NX], x[NX], din[NX * NP_], dout[NX * NP_], dsig[6 * NP_];
    for (int j = 0; j < NP_; ++j) g[j] = 0.0;
    for (int i = 0; i < NX * NP_; ++i) din[i] = 0.0;
    if (dx_dp_hist) for (int i = 0; i < NX * NP_; ++i) io.put(dx_dp_hist, i, 0.0);                   // slot 0
    if (ds_dp_hist) for (int i = 0; i < 6 * NP_; ++i) io.put(ds_dp_hist, i, 0.0);
    for (int step = 1; step <= K; ++step) {
        // both configurations and both states are read per step (nothing but the sensitivity block is carried)
        for (int k = 0; k < NU; ++k) { G[k] = io.get(gradu_hist, (int64_t)step * NU + k); Gp[k] = io.get(gradu_hist, (int64_t)(step - 1) * NU + k); }
        for (int k = 0; k < NX; ++k) { x[k] = io.get(xi_hist, (int64_t)step * NX + k); xp[k] = io.get(xi_hist, (int64_t)(step - 1) * NX + k); }
        if constexpr (RU) ru_direct<YK>(m, G[0] - Gp[0], x, xp, step > 1 ? din : nullptr, dout, dsig);
        else direct_point<MK, DEF, YK, ROT>(m, G, Gp, x, xp, step > 1 ? din : nullptr, dout, dsig);
        if (dx_dp_hist) for (int i = 0; i < NX * NP_; ++i) io.put(dx_dp_hist, (int64_t)step * NX * NP_ + i, dout[i]);
        if (ds_dp_hist) for (int i = 0; i < 6 * NP_; ++i) io.put(ds_dp_hist, (int64_t)step * 6 * NP_ + i, dsig[i]);
        if (sbar_hist) {
            for (int r = 0; r < 6; ++r) {
                const double sb = io.get(sbar_hist, (int64_t)step * 6 + r);
                for (int j = 0; j < NP_; ++j) g[j] += sb * dsig[r * NP_ + j];
            }
        }
        if (xibar_hist) {
            for (int i = 0; i < NX; ++i) {
                const double xb = io.get(xibar_hist, (int64_t)step * NX + i);
                for (int j = 0; j < NP_; ++j) g[j] += xb * dout[i * NP_ + j];
            }
        }
        for (int i = 0; i < NX * NP_; ++i) din[i] = dout[i];
    }
}

}  // namespace cm
