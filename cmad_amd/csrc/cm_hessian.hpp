// Second derivatives of the local residual and of the Cauchy stress (the reference's Model.evaluate_hessians,
// cmad/models/model.py:133-147,245-270, and the QoI Hessians, cmad/qois/qoi.py:41-58,160-188).
//
// One thread computes ONE mixed second derivative d^2 / (d q_a d q_b) of the whole residual vector and of the six
// stress entries at one point, by evaluating the residual in hyper-dual arithmetic (value, d/da, d/db, d2/dadb).
// q = [xi (n_xi), xi_prev (n_xi), p (CM_NUM_PARAMS, KP order)].  This is the one place of the library that
// differentiates by operator overloading instead of a hand-derived formula: the objects are third derivatives of
// the yield function, needed only by the Newton-type calibration driver (MPDirectAdjointObjective), B is small,
// and each thread needs just four doubles per scalar.  The first-derivative parts (d/da, d/db) are checked in
// tests against the hand-derived blocks of cm_evaluate, the second-order parts against the oracle.
#pragma once
#include "cm_device.hpp"

namespace cm {

struct HD {            // f, f_a, f_b, f_ab
    double v, a, b, ab;
};
CM_D HD hd(double c) { return HD{c, 0.0, 0.0, 0.0}; }
CM_D HD operator+(const HD& x, const HD& y) { return HD{x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; }
CM_D HD operator-(const HD& x, const HD& y) { return HD{x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; }
CM_D HD operator-(const HD& x) { return HD{-x.v, -x.a, -x.b, -x.ab}; }
CM_D HD operator*(const HD& x, const HD& y) {
    return HD{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
}
CM_D HD operator*(double c, const HD& x) { return HD{c * x.v, c * x.a, c * x.b, c * x.ab}; }
CM_D HD operator*(const HD& x, double c) { return c * x; }
CM_D HD operator+(const HD& x, double c) { return HD{x.v + c, x.a, x.b, x.ab}; }
CM_D HD operator+(double c, const HD& x) { return x + c; }
CM_D HD operator-(const HD& x, double c) { return HD{x.v - c, x.a, x.b, x.ab}; }
CM_D HD operator-(double c, const HD& x) { return HD{c - x.v, -x.a, -x.b, -x.ab}; }
// g(x) with derivatives g1 = g'(x.v), g2 = g''(x.v)
CM_D HD hd_chain(const HD& x, double g0, double g1, double g2) {
    return HD{g0, g1 * x.a, g1 * x.b, g1 * x.ab + g2 * x.a * x.b};
}
CM_D HD hd_inv(const HD& x) { const double i = 1.0 / x.v; return hd_chain(x, i, -i * i, 2.0 * i * i * i); }
CM_D HD operator/(const HD& x, const HD& y) { return x * hd_inv(y); }
CM_D HD operator/(const HD& x, double c) { return (1.0 / c) * x; }
CM_D HD operator/(double c, const HD& y) { return c * hd_inv(y); }
CM_D HD hd_sqrt(const HD& x) { const double r = sqrt(x.v); return hd_chain(x, r, 0.5 / r, -0.25 / (r * x.v)); }
CM_D HD hd_exp(const HD& x) { const double e = exp(x.v); return hd_chain(x, e, e, e); }
CM_D HD hd_log(const HD& x) { return hd_chain(x, log(x.v), 1.0 / x.v, -1.0 / (x.v * x.v)); }
CM_D HD hd_abs(const HD& x) { const double s = (x.v > 0.0) ? 1.0 : ((x.v < 0.0) ? -1.0 : 0.0); return HD{fabs(x.v), s * x.a, s * x.b, s * x.ab}; }

// scalar-type dispatch so the same templates run on double (host tests) and HD
CM_D double t_sqrt(double x) { return sqrt(x); }
CM_D double t_exp(double x) { return exp(x); }
CM_D double t_log(double x) { return log(x); }
CM_D double t_abs(double x) { return fabs(x); }
CM_D double t_val(double x) { return x; }
CM_D HD t_sqrt(const HD& x) { return hd_sqrt(x); }
CM_D HD t_exp(const HD& x) { return hd_exp(x); }
CM_D HD t_log(const HD& x) { return hd_log(x); }
CM_D HD t_abs(const HD& x) { return hd_abs(x); }
CM_D double t_val(const HD& x) { return x.v; }

// parameters as scalars of type T, KP order (include/cmad_hip.h cm_param_index)
template <class T>
struct MatT { T lambda, mu, Y, S, D, K, yc[6]; };

// effective stress value and 6-vector gradient gt in arithmetic T (the closed forms of yield_eval)
template <int YK, class T>
CM_D void yield_T(const MatT<T>& p, const T s[6], T& phi, T gt[6]) {
    if constexpr (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL) {
        T a00, a33, a55, a03, a05, a35, a11, a22, a44;
        if constexpr (YK == CM_YIELD_J2) {
            a00 = a33 = a55 = T{1.0}; a03 = a05 = a35 = T{-0.5}; a11 = a22 = a44 = T{3.0};
        } else {
            a00 = p.yc[1] + p.yc[2]; a33 = p.yc[0] + p.yc[2]; a55 = p.yc[0] + p.yc[1];
            a03 = -p.yc[2]; a05 = -p.yc[1]; a35 = -p.yc[0];
            a11 = 2.0 * p.yc[5]; a22 = 2.0 * p.yc[4]; a44 = 2.0 * p.yc[3];
        }
        T As[6];
        As[0] = a00 * s[0] + a03 * s[3] + a05 * s[5];
        As[3] = a03 * s[0] + a33 * s[3] + a35 * s[5];
        As[5] = a05 * s[0] + a35 * s[3] + a55 * s[5];
        As[1] = a11 * s[1]; As[2] = a22 * s[2]; As[4] = a44 * s[4];
        T qq = s[0] * As[0];
        for (int k = 1; k < 6; ++k) qq = qq + s[k] * As[k];
        phi = t_sqrt(qq);
        for (int k = 0; k < 6; ++k) gt[k] = As[k] / phi;
    } else {   // Hosford, cmad/models/effective_stress.py:167-177 in the reduced form of yield_eval
        const T& a = p.yc[0];
        const T d[3] = {s[0] - s[3], s[3] - s[5], s[5] - s[0]};
        T t[3], ta[3];
        for (int i = 0; i < 3; ++i) { t[i] = t_abs(d[i]); ta[i] = (t_val(t[i]) > 0.0) ? t_exp(a * t_log(t[i])) : T{0.0}; }
        const T S = 0.5 * (ta[0] + ta[1] + ta[2]);
        phi = t_exp(t_log(S) / a);
        T pd[3];
        for (int i = 0; i < 3; ++i) {
            // d phi / d d_i = 1/2 (t_i / phi)^(a-1) sign(d_i)
            const double sg = (t_val(d[i]) > 0.0) ? 1.0 : ((t_val(d[i]) < 0.0) ? -1.0 : 0.0);
            pd[i] = (t_val(t[i]) > 0.0) ? (0.5 * sg) * t_exp((a - 1.0) * (t_log(t[i]) - t_log(phi))) : T{0.0};
        }
        for (int k = 0; k < 6; ++k) gt[k] = T{0.0};
        gt[0] = pd[0] - pd[2]; gt[3] = pd[1] - pd[0]; gt[5] = pd[2] - pd[1];
    }
}

// residual C(x, xp, p) and material stress s in arithmetic T for the total-form model; `eg`, `z` are plain
// doubles (the second derivatives taken here are w.r.t. xi, xi_prev and the parameters only).
// `plastic` is decided on the primal value (jnp.where semantics, cmad/models/paths.py:26-27).
template <int DEF, int YK, class T>
CM_D void residual_T(const cm_model_desc& m, const MatT<T>& p, const double eg[6], const double* z,
                     const T* x, const T* xp, T* C, T s[6]) {
    T e[6];
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        T t[3] = {T{0.0}, x[7] - 1.0, x[8] - 1.0};
        for (int i = 0; i < 3; ++i) {
            T pv = T{0.0};
            for (int k = 0; k < 6; ++k) pv = pv + (kW[k] * z[6 * i + k]) * x[k];
            t[i] = t[i] - pv;
        }
        for (int k = 0; k < 6; ++k) e[k] = eg[k] + z[k] * t[0] + z[6 + k] * t[1] + z[12 + k] * t[2];
    } else {
        for (int k = 0; k < 6; ++k) {
            e[k] = eg[k] - x[k];
            if constexpr (DEF == CM_PLANE_STRESS) e[k] = e[k] + z[k] * (x[7] - 1.0);
        }
    }
    const T tr = e[0] + e[3] + e[5];
    const T twomu = 2.0 * p.mu;
    for (int k = 0; k < 6; ++k) { s[k] = twomu * e[k]; if (kDiag[k]) s[k] = s[k] + p.lambda * tr; }
    T phi, gt[6];
    yield_T<YK, T>(p, s, phi, gt);
    T H = T{0.0};
    if (m.has_voce) H = H + p.S * (1.0 - t_exp(-(p.D * x[6])));
    if (m.has_linear) H = H + p.K * x[6];
    const T f = (phi - (p.Y + H)) / twomu;
    const T dg = x[6] - xp[6];
    const double fv = t_val(f);
    const bool plastic = (fv > m.yield_tol) || (fabs(fv) < m.yield_tol);
    for (int k = 0; k < 6; ++k) {
        C[k] = x[k] - xp[k];
        if (plastic) C[k] = C[k] - (kIW[k] * dg) * gt[k];
    }
    C[6] = plastic ? f : dg;
    if constexpr (DEF == CM_PLANE_STRESS) {
        T r = T{0.0};
        for (int k = 0; k < 6; ++k) r = r + (kW[k] * z[k]) * s[k];
        C[7] = r / twomu;
    }
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        T ra = T{0.0}, rb = T{0.0};
        for (int k = 0; k < 6; ++k) { ra = ra + (kW[k] * z[6 + k]) * s[k]; rb = rb + (kW[k] * z[12 + k]) * s[k]; }
        C[7] = ra / twomu; C[8] = rb / twomu;
    }
}

// the rate-form residual (cm::residual_rate, cmad/models/small_rate_elastic_plastic.py:249-346) in arithmetic T:
// the unknown x[0:6] is the material stress itself, `deg` the material strain increment (plain doubles).
template <int DEF, int YK, class T>
CM_D void residual_rate_T(const cm_model_desc& m, const MatT<T>& p, const double deg[6], const double* z,
                          const T* x, const T* xp, T* C, T s[6]) {
    static_assert(DEF != CM_UNIAXIAL_STRESS, "rate form: FULL_3D and PLANE_STRESS");
    T e[6];
    for (int k = 0; k < 6; ++k) {
        s[k] = x[k];
        e[k] = T{deg[k]};
        if constexpr (DEF == CM_PLANE_STRESS) e[k] = e[k] + z[k] * (x[7] - xp[7]);
    }
    const T tr = e[0] + e[3] + e[5];
    const T twomu = 2.0 * p.mu;
    T phi, gt[6];
    yield_T<YK, T>(p, s, phi, gt);
    T H = T{0.0};
    if (m.has_voce) H = H + p.S * (1.0 - t_exp(-(p.D * x[6])));
    if (m.has_linear) H = H + p.K * x[6];
    const T f = (phi - (p.Y + H)) / twomu;
    const T dg = x[6] - xp[6];
    const double fv = t_val(f);
    const bool plastic = (fv > m.yield_tol) || (fabs(fv) < m.yield_tol);
    const T gd = gt[0] + gt[3] + gt[5];
    T r7 = T{0.0};
    for (int k = 0; k < 6; ++k) {
        T dc = twomu * e[k];
        if (kDiag[k]) dc = dc + p.lambda * tr;
        if (plastic) {
            T cn = (twomu * kIW[k]) * gt[k];
            if (kDiag[k]) cn = cn + p.lambda * gd;
            dc = dc - dg * cn;
        }
        C[k] = (x[k] - xp[k] - dc) / twomu;
        if constexpr (DEF == CM_PLANE_STRESS) r7 = r7 + (kW[k] * z[k]) * dc;
    }
    C[6] = plastic ? f : dg;
    if constexpr (DEF == CM_PLANE_STRESS) C[7] = r7 / twomu;
}

// local dofs of the AD-evaluated models: the rate form under UNIAXIAL_STRESS carries 12 (stress 6, alpha, two
// off-axis stretches, three off-axis strain increments; small_rate_elastic_plastic.py:171-196), which the
// hand-derived kernels (Dims<DEF>) do not implement
template <int DEF, int MK>
constexpr int nx_of() { return (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS) ? 12 : Dims<DEF>::NX; }

// rate form, UNIAXIAL_STRESS (small_rate_elastic_plastic.py:34-75, 249-346): the global strain increment has the
// on-axis entry dU from the caller, the off-axis normal entries from the stretch unknowns x[7:9] - xp[7:9] and the
// shear entries x[9:12] themselves; every off-axis entry of the global stress increment must vanish.
template <int YK, class T>
CM_D void residual_rate_uniaxial_T(const cm_model_desc& m, const MatT<T>& p, double dU,
                                   const T* x, const T* xp, T* C, T s[6]) {
    const int on = m.uniaxial_idx, ia = (on == 0) ? 1 : 0, ib = (on == 2) ? 1 : 2;
    T eg[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) eg[i][j] = T{0.0};
    eg[on][on] = T{dU};
    eg[ia][ia] = x[7] - xp[7];
    eg[ib][ib] = x[8] - xp[8];
    eg[0][1] = eg[1][0] = x[9]; eg[0][2] = eg[2][0] = x[10]; eg[1][2] = eg[2][1] = x[11];
    // material frame: e = Q^T eg Q  (Q = m.Q row-major, Q_ij = e_i(global) . e_j(material))
    T em[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        T acc = T{0.0};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) acc = acc + (m.Q[3 * a + i] * m.Q[3 * b + j]) * eg[a][b];
        em[i][j] = acc;
    }
    const T e[6] = {em[0][0], em[0][1], em[0][2], em[1][1], em[1][2], em[2][2]};
    for (int k = 0; k < 6; ++k) s[k] = x[k];
    const T tr = e[0] + e[3] + e[5];
    const T twomu = 2.0 * p.mu;
    T phi, gt[6];
    yield_T<YK, T>(p, s, phi, gt);
    T H = T{0.0};
    if (m.has_voce) H = H + p.S * (1.0 - t_exp(-(p.D * x[6])));
    if (m.has_linear) H = H + p.K * x[6];
    const T f = (phi - (p.Y + H)) / twomu;
    const T dg = x[6] - xp[6];
    const double fv = t_val(f);
    const bool plastic = (fv > m.yield_tol) || (fabs(fv) < m.yield_tol);
    const T gd = gt[0] + gt[3] + gt[5];
    T dc[6];
    for (int k = 0; k < 6; ++k) {
        dc[k] = twomu * e[k];
        if (kDiag[k]) dc[k] = dc[k] + p.lambda * tr;
        if (plastic) {
            T cn = (twomu * kIW[k]) * gt[k];
            if (kDiag[k]) cn = cn + p.lambda * gd;
            dc[k] = dc[k] - dg * cn;
        }
        C[k] = (x[k] - xp[k] - dc[k]) / twomu;
    }
    C[6] = plastic ? f : dg;
    // global stress increment Q dc Q^T
    const T dm[3][3] = {{dc[0], dc[1], dc[2]}, {dc[1], dc[3], dc[4]}, {dc[2], dc[4], dc[5]}};
    T dgl[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        T acc = T{0.0};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) acc = acc + (m.Q[3 * i + a] * m.Q[3 * j + b]) * dm[a][b];
        dgl[i][j] = acc;
    }
    C[7] = dgl[ia][ia] / twomu; C[8] = dgl[ib][ib] / twomu;
    C[9] = dgl[0][1] / twomu; C[10] = dgl[0][2] / twomu; C[11] = dgl[1][2] / twomu;
}

// one (a, b) pair: out_C[NX] = d2 C / dq_a dq_b, out_S[6] = d2 sigma_global / dq_a dq_b,
// and the first derivatives wrt q_a (for cross-checks): out_Ca[NX], out_Sa[6]
// MK = CM_SMALL_RATE_ELASTIC_PLASTIC: G must already hold grad u - grad u_prev (the strain is linear in it).
template <int DEF, int YK, bool ROT, int MK = CM_SMALL_ELASTIC_PLASTIC>
CM_D void hessian_pair(const cm_model_desc& m, const double* G, const double* xv, const double* xpv, int a, int b,
                       double* out_C, double* out_S, double* out_Ca, double* out_Sa,
                       double* out_C0 = nullptr, double* out_S0 = nullptr, double* out_Sb = nullptr) {
    constexpr int NX = nx_of<DEF, MK>();
    constexpr bool RATE_UNI = (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS);
    double eg[6], z[Dims<DEF>::NZ];
    if constexpr (!RATE_UNI) {
        strain_from_gradu<DEF, ROT>(m, G, eg);
        strain_z<DEF, ROT>(m, z);
    }
    HD x[NX], xp[NX], C[NX], s[6];
    MatT<HD> p;
    double pv[CM_NUM_PARAMS] = {m.lambda, m.mu, m.Y, m.voce_S, m.voce_D, m.lin_K,
                                m.yc[0], m.yc[1], m.yc[2], m.yc[3], m.yc[4], m.yc[5]};
    HD q[2 * NX + CM_NUM_PARAMS];
    for (int k = 0; k < NX; ++k) { q[k] = hd(xv[k]); q[NX + k] = hd(xpv[k]); }
    for (int k = 0; k < CM_NUM_PARAMS; ++k) q[2 * NX + k] = hd(pv[k]);
    q[a].a = 1.0; q[b].b = 1.0;
    for (int k = 0; k < NX; ++k) { x[k] = q[k]; xp[k] = q[NX + k]; }
    const HD* pp = q + 2 * NX;
    p.lambda = pp[0]; p.mu = pp[1]; p.Y = pp[2]; p.S = pp[3]; p.D = pp[4]; p.K = pp[5];
    for (int k = 0; k < 6; ++k) p.yc[k] = pp[6 + k];
    if constexpr (RATE_UNI) residual_rate_uniaxial_T<YK, HD>(m, p, G[0], x, xp, C, s);
    else if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) residual_rate_T<DEF, YK, HD>(m, p, eg, z, x, xp, C, s);
    else residual_T<DEF, YK, HD>(m, p, eg, z, x, xp, C, s);
    for (int k = 0; k < NX; ++k) { out_C[k] = C[k].ab; out_Ca[k] = C[k].a; }
    double s2[6], s1[6], s0[6], g2[6], g1[6], g0[6];
    for (int k = 0; k < 6; ++k) { s2[k] = s[k].ab; s1[k] = s[k].a; s0[k] = s[k].v; }
    to_global<ROT>(m, s2, g2);
    to_global<ROT>(m, s1, g1);
    for (int k = 0; k < 6; ++k) { out_S[k] = g2[k]; out_Sa[k] = g1[k]; }
    if (out_Sb) {                                         // first derivative of the global stress w.r.t. q_b
        double sb1[6], gb1[6];
        for (int k = 0; k < 6; ++k) sb1[k] = s[k].b;
        to_global<ROT>(m, sb1, gb1);
        for (int k = 0; k < 6; ++k) out_Sb[k] = gb1[k];
    }
    if (out_C0) for (int k = 0; k < NX; ++k) out_C0[k] = C[k].v;
    if (out_S0) { to_global<ROT>(m, s0, g0); for (int k = 0; k < 6; ++k) out_S0[k] = g0[k]; }
}

// ---- one entry of the second-order weight matrix of a history step (cm_hessian_history) ---------------------------------
// W[a][b] = sbar . d2 sigma / dq_a dq_b + sum_r hss_r d sigma_r / dq_a  d sigma_r / dq_b - lam . d2 C / dq_a dq_b,
// q = [xi, xi_prev, p]: the Hessian in q of the step's Lagrangian J_k(sigma) - lam . C_k for a QoI whose curvature in the six
// stored stress entries is diagonal (cmad/qois/calibration.py:56-66: hss = folded squared weights), with lam = -phi of
// cmad/objectives/mp_objective.py:255-281.  G: grad u (rate form: grad u - grad u_prev).
template <int DEF, int YK, bool ROT, int MK>
CM_D double hessian_weight(const cm_model_desc& m, const double* G, const double* x, const double* xp, const double* lam,
                           const double sbar[6], const double hss[6], int a, int b) {
    constexpr int NX = nx_of<DEF, MK>();
    double oC[NX], oS[6], oCa[NX], oSa[6], oSb[6];
    hessian_pair<DEF, YK, ROT, MK>(m, G, x, xp, a, b, oC, oS, oCa, oSa, nullptr, nullptr, oSb);
    double w = 0.0;
    for (int k = 0; k < NX; ++k) w -= lam[k] * oC[k];
    for (int r = 0; r < 6; ++r) w += sbar[r] * oS[r] + hss[r] * oSa[r] * oSb[r];
    return w;
}

}  // namespace cm
