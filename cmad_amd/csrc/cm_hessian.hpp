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

// first-order dual number (value, one directional derivative): the arithmetic of the extended parameter sensitivities
struct D1 {
    double v, d;
};
CM_D D1 d1(double c) { return D1{c, 0.0}; }
CM_D D1 operator+(const D1& x, const D1& y) { return D1{x.v + y.v, x.d + y.d}; }
CM_D D1 operator-(const D1& x, const D1& y) { return D1{x.v - y.v, x.d - y.d}; }
CM_D D1 operator-(const D1& x) { return D1{-x.v, -x.d}; }
CM_D D1 operator*(const D1& x, const D1& y) { return D1{x.v * y.v, x.d * y.v + x.v * y.d}; }
CM_D D1 operator*(double c, const D1& x) { return D1{c * x.v, c * x.d}; }
CM_D D1 operator*(const D1& x, double c) { return c * x; }
CM_D D1 operator+(const D1& x, double c) { return D1{x.v + c, x.d}; }
CM_D D1 operator+(double c, const D1& x) { return x + c; }
CM_D D1 operator-(const D1& x, double c) { return D1{x.v - c, x.d}; }
CM_D D1 operator-(double c, const D1& x) { return D1{c - x.v, -x.d}; }
CM_D D1 d1_chain(const D1& x, double g0, double g1) { return D1{g0, g1 * x.d}; }
CM_D D1 operator/(const D1& x, const D1& y) { const double i = 1.0 / y.v; return D1{x.v * i, (x.d - x.v * i * y.d) * i}; }
CM_D D1 operator/(const D1& x, double c) { return (1.0 / c) * x; }
CM_D D1 operator/(double c, const D1& y) { const double i = 1.0 / y.v; return D1{c * i, -c * i * i * y.d}; }

// complex scalars: the arithmetic of the reference's complex-step model instances (is_complex = True, e.g.
// cmad/models/small_elastic_plastic.py:118-127: the whole model traced with complex dtype so that Im J(p + i h d) / h checks the
// gradients, tests/objectives/test_J2_fd_checks.py:163-235).  Holomorphic continuations of the real functions; |z| and every
// comparison follow the complex-step convention (decided on the real part), so the model stays analytic along the real axis.
struct CX {
    double re, im;
};
CM_D CX operator+(const CX& x, const CX& y) { return CX{x.re + y.re, x.im + y.im}; }
CM_D CX operator-(const CX& x, const CX& y) { return CX{x.re - y.re, x.im - y.im}; }
CM_D CX operator-(const CX& x) { return CX{-x.re, -x.im}; }
CM_D CX operator*(const CX& x, const CX& y) { return CX{x.re * y.re - x.im * y.im, x.re * y.im + x.im * y.re}; }
CM_D CX operator*(double c, const CX& x) { return CX{c * x.re, c * x.im}; }
CM_D CX operator*(const CX& x, double c) { return c * x; }
CM_D CX operator+(const CX& x, double c) { return CX{x.re + c, x.im}; }
CM_D CX operator+(double c, const CX& x) { return x + c; }
CM_D CX operator-(const CX& x, double c) { return CX{x.re - c, x.im}; }
CM_D CX operator-(double c, const CX& x) { return CX{c - x.re, -x.im}; }
CM_D CX cx_inv(const CX& y) { const double n = 1.0 / (y.re * y.re + y.im * y.im); return CX{y.re * n, -y.im * n}; }
CM_D CX operator/(const CX& x, const CX& y) { return x * cx_inv(y); }
CM_D CX operator/(const CX& x, double c) { return (1.0 / c) * x; }
CM_D CX operator/(double c, const CX& y) { return c * cx_inv(y); }
CM_D CX cx_sqrt(const CX& z) {                                  // principal branch, stable for |im| << |re|
    const double r = sqrt(z.re * z.re + z.im * z.im);
    if (r == 0.0) return CX{0.0, 0.0};
    if (z.re >= 0.0) { const double a = sqrt(0.5 * (r + z.re)); return CX{a, 0.5 * z.im / a}; }
    const double b = sqrt(0.5 * (r - z.re));
    return CX{0.5 * fabs(z.im) / b, (z.im >= 0.0) ? b : -b};
}
CM_D CX cx_exp(const CX& z) { const double e = exp(z.re); return CX{e * cos(z.im), e * sin(z.im)}; }
CM_D CX cx_log(const CX& z) { return CX{0.5 * log(z.re * z.re + z.im * z.im), atan2(z.im, z.re)}; }
CM_D CX cx_abs(const CX& z) { return (z.re > 0.0) ? z : ((z.re < 0.0) ? -z : CX{0.0, 0.0}); }

// dual numbers over the complex scalars: the Jacobian d C / d x of the complex residual, one column per evaluation
struct DC {
    CX v, d;
};
CM_D DC operator+(const DC& x, const DC& y) { return DC{x.v + y.v, x.d + y.d}; }
CM_D DC operator-(const DC& x, const DC& y) { return DC{x.v - y.v, x.d - y.d}; }
CM_D DC operator-(const DC& x) { return DC{-x.v, -x.d}; }
CM_D DC operator*(const DC& x, const DC& y) { return DC{x.v * y.v, x.d * y.v + x.v * y.d}; }
CM_D DC operator*(double c, const DC& x) { return DC{c * x.v, c * x.d}; }
CM_D DC operator*(const DC& x, double c) { return c * x; }
CM_D DC operator+(const DC& x, double c) { return DC{x.v + c, x.d}; }
CM_D DC operator+(double c, const DC& x) { return x + c; }
CM_D DC operator-(const DC& x, double c) { return DC{x.v - c, x.d}; }
CM_D DC operator-(double c, const DC& x) { return DC{c - x.v, -x.d}; }
CM_D DC operator/(const DC& x, const DC& y) { const CX i = cx_inv(y.v); return DC{x.v * i, (x.d - x.v * i * y.d) * i}; }
CM_D DC operator/(const DC& x, double c) { return (1.0 / c) * x; }
CM_D DC operator/(double c, const DC& y) { const CX i = cx_inv(y.v); return DC{c * i, -(c * i * i) * y.d}; }

// scalar-type dispatch so the same templates run on double (host tests), D1 and HD
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
CM_D D1 t_sqrt(const D1& x) { const double r = sqrt(x.v); return d1_chain(x, r, 0.5 / r); }
CM_D D1 t_exp(const D1& x) { const double e = exp(x.v); return d1_chain(x, e, e); }
CM_D D1 t_log(const D1& x) { return d1_chain(x, log(x.v), 1.0 / x.v); }
CM_D D1 t_abs(const D1& x) { const double sg = (x.v > 0.0) ? 1.0 : ((x.v < 0.0) ? -1.0 : 0.0); return D1{fabs(x.v), sg * x.d}; }
CM_D double t_val(const D1& x) { return x.v; }
CM_D CX t_sqrt(const CX& x) { return cx_sqrt(x); }
CM_D CX t_exp(const CX& x) { return cx_exp(x); }
CM_D CX t_log(const CX& x) { return cx_log(x); }
CM_D CX t_abs(const CX& x) { return cx_abs(x); }
CM_D double t_val(const CX& x) { return x.re; }
CM_D DC t_sqrt(const DC& x) { const CX r = cx_sqrt(x.v); return DC{r, (0.5 * cx_inv(r)) * x.d}; }
CM_D DC t_exp(const DC& x) { const CX e = cx_exp(x.v); return DC{e, e * x.d}; }
CM_D DC t_log(const DC& x) { return DC{cx_log(x.v), cx_inv(x.v) * x.d}; }
CM_D DC t_abs(const DC& x) { return (x.v.re > 0.0) ? x : ((x.v.re < 0.0) ? -x : DC{CX{0.0, 0.0}, CX{0.0, 0.0}}); }
CM_D double t_val(const DC& x) { return x.v.re; }
// constants of type T, and the seed of a differentiation variable (D1: its one direction)
template <class T> CM_D T t_const(double c);
template <> CM_D double t_const<double>(double c) { return c; }
template <> CM_D D1 t_const<D1>(double c) { return D1{c, 0.0}; }
template <> CM_D HD t_const<HD>(double c) { return HD{c, 0.0, 0.0, 0.0}; }
template <> CM_D CX t_const<CX>(double c) { return CX{c, 0.0}; }
template <> CM_D DC t_const<DC>(double c) { return DC{CX{c, 0.0}, CX{0.0, 0.0}}; }
CM_D void t_seed(CX&) {}
CM_D void t_seed(DC& x) { x.d = CX{1.0, 0.0}; }
CM_D void t_seed2(CX&) {}
CM_D void t_seed2(DC&) {}
// the imaginary part of a complex-step parameter (real arithmetic types have none)
template <class T> CM_D void t_add_imag(T&, double) {}
CM_D void t_add_imag(CX& x, double im) { x.im += im; }
CM_D void t_add_imag(DC& x, double im) { x.v.im += im; }
CM_D void t_seed(double&) {}
CM_D void t_seed(D1& x) { x.d = 1.0; }
CM_D void t_seed(HD& x) { x.a = 1.0; }         // first differentiation direction of a hyper-dual ...
CM_D void t_seed2(double&) {}
CM_D void t_seed2(D1&) {}
CM_D void t_seed2(HD& x) { x.b = 1.0; }        // ... and the second one

// parameters as scalars of type T, KP order (include/cmad_hip.h cm_param_index)
// yc: the yield-surface coefficients as cm_model_desc.yc (Hill F..N | Hosford a | Barlat 18 + a), Q: the rotation matrix,
// nn: the packed network weights (plain doubles; weight nn_seed, if any, carries the derivative direction).
template <class T>
struct MatT {
    T lambda, mu, Y, S, D, K, yc[19], Q[9];
    const double* nn;
    const double* nn_im;             // complex-step instances: imaginary parts of the packed weights (same indexing), or null
    int nn_seed, nn_seed2;           // packed-weight index carrying the (first / second) derivative direction, or -1
    CM_D T nn_at(int i) const {
        T w = t_const<T>(nn[i]);
        if (nn_im) t_add_imag(w, nn_im[i]);
        if (i == nn_seed) t_seed(w);
        if (i == nn_seed2) t_seed2(w);
        return w;
    }
};

// every parameter at its value in the model description (no derivative parts)
template <class T>
CM_D void mat_from_desc(const cm_model_desc& m, MatT<T>& p) {
    p.lambda = t_const<T>(m.lambda); p.mu = t_const<T>(m.mu); p.Y = t_const<T>(m.Y);
    p.S = t_const<T>(m.voce_S); p.D = t_const<T>(m.voce_D); p.K = t_const<T>(m.lin_K);
    for (int k = 0; k < 19; ++k) p.yc[k] = t_const<T>(m.yc[k]);
    for (int k = 0; k < 9; ++k) p.Q[k] = t_const<T>(m.Q[k]);
    p.nn = m.nn_weights; p.nn_im = nullptr; p.nn_seed = -1; p.nn_seed2 = -1;
}

// Extended parameter ("EP") index of the sensitivities beyond the 12 of cm_param_index: 0..11 = KP order,
// 12..24 = yc[6..18] (Barlat), 25..33 = Q[0..8] row-major, 34 + i = packed network weight i.
enum { CM_EP_YC6 = 12, CM_EP_Q0 = 25, CM_EP_NN0 = 34 };
template <class T>
CM_D void mat_seed(MatT<T>& p, int e) {
    if (e == CM_P_LAMBDA) t_seed(p.lambda);
    else if (e == CM_P_MU) t_seed(p.mu);
    else if (e == CM_P_Y) t_seed(p.Y);
    else if (e == CM_P_VOCE_S) t_seed(p.S);
    else if (e == CM_P_VOCE_D) t_seed(p.D);
    else if (e == CM_P_LIN_K) t_seed(p.K);
    else if (e < CM_EP_Q0) t_seed(p.yc[e - CM_P_YC0]);
    else if (e < CM_EP_NN0) t_seed(p.Q[e - CM_EP_Q0]);
    else p.nn_seed = e - CM_EP_NN0;
}

// ---- the network term of the hybrid surface in arithmetic T (cm::icnn_yield_term: value and d/ds6) ---------------------
// jax.nn.softplus = logaddexp(a, 0) and its derivative, the logistic function, in the overflow-safe forms
template <class T>
CM_D void softplus_T(const T& a, T& sp, T& sg) {
    const bool pos = t_val(a) > 0.0;
    const T e = t_exp(pos ? -a : a);                           // exp(-|a|)
    const T inv = 1.0 / (1.0 + e);
    const T l = t_log(1.0 + e);
    sp = pos ? a + l : l;
    sg = pos ? inv : e * inv;
}
// the isotropic hardening laws in arithmetic T (cm::hardening; cmad/models/hardening.py:9-34 and the network law of
// cmad/neural_networks/simple_neural_network.py:13-46, widths [1, H, 1], weights from the nn blob: every weight can carry the
// derivative direction)
template <class T>
CM_D T hardening_T(const cm_model_desc& m, const MatT<T>& p, const T& alpha) {
    T H = t_const<T>(0.0);
    if (m.has_voce) H = H + p.S * (1.0 - t_exp(-(p.D * alpha)));
    if (m.has_linear) H = H + p.K * alpha;
    if (CM_HNN != 0 && m.hnn_width > 0 && m.hnn_nhidden >= 2) {   // widths [1, H1, ..., Hn, 1]: cm::hardening_network_deep in arithmetic T
        const int nh = m.hnn_nhidden, o = m.hnn_offset, tail = hnn_general_size(m);
        const double si = p.nn[o + tail], so = p.nn[o + tail + 1];
        auto sigmoid = [](const T& a) {
            const bool pos = t_val(a) > 0.0;
            const T e = t_exp(pos ? -a : a);
            const T inv = 1.0 / (1.0 + e);
            return pos ? inv : e * inv;
        };
        auto forward = [&](const T& x) {                       // every weight can carry the derivative direction (p.nn_at)
            T a[2][kHnnMaxUnits];
            a[0][0] = x;
            int off = 0, nin = 1, cur = 0;
            for (int l = 0; l < nh; ++l) {
                const int nout = m.hnn_widths[l];
                for (int v = 0; v < nout; ++v) {
                    T z = p.nn_at(o + off + nin * nout + v);
                    for (int i = 0; i < nin; ++i) z = z + p.nn_at(o + off + i * nout + v) * a[cur][i];
                    a[cur ^ 1][v] = sigmoid(z);
                }
                off += nin * nout + nout; nin = nout; cur ^= 1;
            }
            T y = p.nn_at(o + off + nin);
            for (int i = 0; i < nin; ++i) y = y + p.nn_at(o + off + i) * a[cur][i];
            return y;
        };
        H = H + so * (forward(si * alpha) - forward(t_const<T>(0.0)));
    } else if (CM_HNN != 0 && m.hnn_width > 0) {               // (only the HNN build of the library carries the law, cm_device.hpp)
        const int Hn = m.hnn_width, o = m.hnn_offset;
        const double si = p.nn[o + 3 * Hn + 1], so = p.nn[o + 3 * Hn + 2];
        auto sigmoid = [](const T& a) {
            const bool pos = t_val(a) > 0.0;
            const T e = t_exp(pos ? -a : a);
            const T inv = 1.0 / (1.0 + e);
            return pos ? inv : e * inv;
        };
        T acc = t_const<T>(0.0);
        for (int u = 0; u < Hn; ++u) {
            const T w1 = p.nn_at(o + u), b1 = p.nn_at(o + Hn + u), w2 = p.nn_at(o + 2 * Hn + u);
            acc = acc + w2 * (sigmoid(w1 * (si * alpha) + b1) - sigmoid(b1));
        }
        H = H + so * acc;
    }
    return H;
}

// A network with any number of hidden layers in arithmetic T (cm::icnn_symmetric_deep): f(x) and grad f(x), the gradient by the
// backward pass mu_n = wz_n o sigmoid(a_n), mu_k = (Wz_k mu_{k+1}) o sigmoid(a_k), grad f = wx_last + sum_k Wx_k mu_k.
template <class T>
CM_D void icnn_deep_value_grad_T(const cm_model_desc& m, const MatT<T>& p, const T x[6], T& f, T g[6]) {
    const int nl = m.nn_nlayers, nh = nl - 2;
    int xoff[kIcnnMaxLayers], zoff[kIcnnMaxLayers], uoff[kIcnnMaxLayers];
    {
        int off = 0, u = 0;
        for (int k = 1; k <= nh + 1; ++k) { xoff[k] = off; off += 7 * m.nn_widths[k]; }
        for (int k = 1; k <= nh; ++k) { zoff[k] = off; off += m.nn_widths[k] * m.nn_widths[k + 1]; }
        for (int k = 1; k <= nh; ++k) { uoff[k] = u; u += m.nn_widths[k]; }
    }
    T spv[kIcnnMaxUnits], sgv[kIcnnMaxUnits], mu[kIcnnMaxUnits];
    for (int k = 1; k <= nh; ++k) {
        const int Hk = m.nn_widths[k], Hp = (k > 1) ? m.nn_widths[k - 1] : 0;
        for (int v = 0; v < Hk; ++v) {
            T a = p.nn_at(xoff[k] + 6 * Hk + v);
            for (int i = 0; i < 6; ++i) a = a + x[i] * p.nn_at(xoff[k] + i * Hk + v);
            for (int u = 0; u < Hp; ++u) a = a + p.nn_at(zoff[k - 1] + u * Hk + v) * spv[uoff[k - 1] + u];
            softplus_T(a, spv[uoff[k] + v], sgv[uoff[k] + v]);
        }
    }
    const int Hn = m.nn_widths[nh];
    f = p.nn_at(xoff[nh + 1] + 6);
    for (int i = 0; i < 6; ++i) { g[i] = p.nn_at(xoff[nh + 1] + i); f = f + x[i] * g[i]; }
    for (int u = 0; u < Hn; ++u) {
        const T wz = p.nn_at(zoff[nh] + u);
        f = f + wz * spv[uoff[nh] + u];
        mu[uoff[nh] + u] = wz * sgv[uoff[nh] + u];
    }
    for (int k = nh - 1; k >= 1; --k) {
        const int Hk = m.nn_widths[k], Hq = m.nn_widths[k + 1];
        for (int u = 0; u < Hk; ++u) {
            T t = t_const<T>(0.0);
            for (int v = 0; v < Hq; ++v) t = t + p.nn_at(zoff[k] + u * Hq + v) * mu[uoff[k + 1] + v];
            mu[uoff[k] + u] = t * sgv[uoff[k] + u];
        }
    }
    for (int k = 1; k <= nh; ++k) {
        const int Hk = m.nn_widths[k];
        for (int v = 0; v < Hk; ++v)
            for (int i = 0; i < 6; ++i) g[i] = g[i] + p.nn_at(xoff[k] + i * Hk + v) * mu[uoff[k] + v];
    }
}

template <class T>
CM_D void icnn_yield_T(const cm_model_desc& m, const MatT<T>& p, const T s[6], T& val, T g6[6]) {
    if (CM_HNN != 0 && m.nn_nlayers > 3) {                      // more than one hidden layer (general evaluation; EXT / host builds)
        const double* sc = p.nn + icnn_deep_scalers_offset(m);
        const T h = (s[0] + s[3] + s[5]) * (1.0 / 3.0);
        const T x[6] = {s[0] - h, s[3] - h, s[5] - h, s[1], s[2], s[4]};
        T xs[6], xn[6], x0[6], fp, fn, f0, gp[6], gn[6], g0[6];
        for (int i = 0; i < 6; ++i) { xs[i] = sc[i] * x[i] + sc[6 + i]; xn[i] = -xs[i]; x0[i] = t_const<T>(0.0); }
        icnn_deep_value_grad_T<T>(m, p, xs, fp, gp);
        icnn_deep_value_grad_T<T>(m, p, xn, fn, gn);
        icnn_deep_value_grad_T<T>(m, p, x0, f0, g0);
        const double ios = 1.0 / sc[12];
        val = (0.5 * (fp + fn) - f0 - sc[13]) * ios;
        T gx[6];
        for (int i = 0; i < 6; ++i) gx[i] = (0.5 * sc[i] * ios) * (gp[i] - gn[i]);
        constexpr int XI[6] = {0, 3, 4, 1, 5, 2};
        const T gm = (gx[0] + gx[1] + gx[2]) * (1.0 / 3.0);
        for (int k = 0; k < 6; ++k) { g6[k] = gx[XI[k]]; if (kDiag[k]) g6[k] = g6[k] - gm; }
        return;
    }
    const int H = m.nn_widths[1];
    const int oW0 = 0, ob0 = 6 * H, ob1 = 7 * H + 6, oWz = 7 * H + 7;
    const double* sc = p.nn + 8 * H + 7;                         // in_scale[6], in_min[6], out_scale, out_min (constants)
    const T h = (s[0] + s[3] + s[5]) * (1.0 / 3.0);
    const T x[6] = {s[0] - h, s[3] - h, s[5] - h, s[1], s[2], s[4]};
    T xs[6];
    for (int i = 0; i < 6; ++i) xs[i] = sc[i] * x[i] + sc[6 + i];
    T F = 2.0 * p.nn_at(ob1), f0 = p.nn_at(ob1), G[6];
    for (int i = 0; i < 6; ++i) G[i] = t_const<T>(0.0);
    for (int o = 0; o < H; ++o) {
        T wc[6], t = t_const<T>(0.0);
        for (int i = 0; i < 6; ++i) { wc[i] = p.nn_at(oW0 + i * H + o); t = t + xs[i] * wc[i]; }
        const T b = p.nn_at(ob0 + o), wz = p.nn_at(oWz + o);
        T spp, sgp, spn, sgn, sp0, sg0;
        softplus_T(b + t, spp, sgp);
        softplus_T(b - t, spn, sgn);
        softplus_T(b, sp0, sg0);                               // the network at the origin: forward(0) = softplus(b0) . Wz + b1
        F = F + (spp + spn) * wz;
        f0 = f0 + sp0 * wz;
        const T c1 = (sgp - sgn) * wz;
        for (int i = 0; i < 6; ++i) G[i] = G[i] + c1 * wc[i];
    }
    const double ios = 1.0 / sc[12];
    val = (0.5 * F - f0 - sc[13]) * ios;                        // (1/2 (f(x) + f(-x)) - f(0) - out_min) / out_scale
    T gx[6];
    for (int i = 0; i < 6; ++i) gx[i] = (0.5 * sc[i] * ios) * G[i];
    constexpr int XI[6] = {0, 3, 4, 1, 5, 2};
    const T gm = (gx[0] + gx[1] + gx[2]) * (1.0 / 3.0);
    for (int k = 0; k < 6; ++k) { g6[k] = gx[XI[k]]; if (kDiag[k]) g6[k] = g6[k] - gm; }
}

// ---- Barlat Yld2004-18p in arithmetic T (cm::barlat_eval: value and d/ds6) ---------------------------------------------
// cyclic Jacobi on a symmetric 3x3 of type T (cm::eig_sym3); rotations are decided on the values.  Where an off-diagonal
// entry is already negligible against the eigenvalue gap the rotation angle is taken from first-order perturbation theory
// (t = a_pq / (a_qq - a_pp)), which stays finite when the VALUE of a_pq is exactly zero but its derivative part is not
// (stresses aligned with the material axes).  Repeated eigenvalues have no differentiable eigenvectors: derivative parts are
// then not meaningful (the reference's eigh rule divides by zero there as well).
template <class T>
CM_D void eig_sym3_T(const T s[6], T lam[3], T V[3][3]) {
    T a00 = s[0], a01 = s[1], a02 = s[2], a11 = s[3], a12 = s[4], a22 = s[5];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = t_const<T>((i == j) ? 1.0 : 0.0);
    auto rotate = [&](T& app, T& aqq, T& apq, T& apr, T& aqr, int P, int Q) {
        const double gap = t_val(aqq) - t_val(app), off = t_val(apq);
        T t;
        if (fabs(off) < 1e-9 * fabs(gap)) t = apq / (aqq - app);
        else if (off == 0.0) return;                              // nothing to rotate (and no gap to define a direction)
        else {
            const T theta = (aqq - app) / (2.0 * apq);
            const double sg = (t_val(theta) >= 0.0) ? 1.0 : -1.0;
            t = sg / (sg * theta + t_sqrt(theta * theta + 1.0));
        }
        const T c = 1.0 / t_sqrt(t * t + 1.0), sn = t * c;
        app = app - t * apq; aqq = aqq + t * apq; apq = t_const<T>(0.0);
        const T xr = apr, yr = aqr;
        apr = c * xr - sn * yr; aqr = sn * xr + c * yr;
        for (int k = 0; k < 3; ++k) {
            const T vp = V[k][P], vq = V[k][Q];
            V[k][P] = c * vp - sn * vq; V[k][Q] = sn * vp + c * vq;
        }
    };
    for (int sweep = 0; sweep < 6; ++sweep) {
        rotate(a00, a11, a01, a02, a12, 0, 1);
        rotate(a00, a22, a02, a01, a12, 0, 2);
        rotate(a11, a22, a12, a01, a02, 1, 2);
    }
    lam[0] = a00; lam[1] = a11; lam[2] = a22;
}

template <class T>
CM_D void barlat_T(const cm_model_desc& m, const MatT<T>& p, const T s[6], T& phi, T gt[6]) {
    const T& a = p.yc[18];
    T lam[2][3], V[2][3][3], UL[2][3][3], csh[2][3];
    for (int set = 0; set < 2; ++set) {
        const T* q = p.yc + 9 * set;                             // c12, c13, c21, c23, c31, c32, c44, c55, c66
        const double t3 = 1.0 / 3.0;
        UL[set][0][0] = t3 * (q[0] + q[1]); UL[set][0][1] = t3 * (q[1] - 2.0 * q[0]); UL[set][0][2] = t3 * (q[0] - 2.0 * q[1]);
        UL[set][1][0] = t3 * (q[3] - 2.0 * q[2]); UL[set][1][1] = t3 * (q[2] + q[3]); UL[set][1][2] = t3 * (q[2] - 2.0 * q[3]);
        UL[set][2][0] = t3 * (q[5] - 2.0 * q[4]); UL[set][2][1] = t3 * (q[4] - 2.0 * q[5]); UL[set][2][2] = t3 * (q[4] + q[5]);
        csh[set][0] = q[6]; csh[set][1] = q[7]; csh[set][2] = q[8];
        T S6[6];
        S6[0] = UL[set][0][0] * s[0] + UL[set][0][1] * s[3] + UL[set][0][2] * s[5];
        S6[3] = UL[set][1][0] * s[0] + UL[set][1][1] * s[3] + UL[set][1][2] * s[5];
        S6[5] = UL[set][2][0] * s[0] + UL[set][2][1] * s[3] + UL[set][2][2] * s[5];
        S6[1] = q[6] * s[1]; S6[4] = q[7] * s[4]; S6[2] = q[8] * s[2];     // xy: c44, yz: c55, zx: c66
        eig_sym3_T<T>(S6, lam[set], V[set]);
    }
    T Dm[3][3], u[3][3], ua[3][3], Ssum = t_const<T>(0.0);
    double mx = 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Dm[i][j] = lam[0][i] - lam[1][j]; mx = fmax(mx, fabs(t_val(Dm[i][j]))); }
    const double imx = (mx > 0.0) ? 1.0 / mx : 0.0;             // a constant scale (no derivative parts): cancels analytically
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        u[i][j] = imx * t_abs(Dm[i][j]);
        ua[i][j] = (t_val(u[i][j]) > 0.0) ? t_exp(a * t_log(u[i][j])) : t_const<T>(0.0);
        Ssum = Ssum + ua[i][j];
    }
    Ssum = 0.25 * Ssum;
    const T Sr = t_exp(t_log(Ssum) / a);
    phi = mx * Sr;
    // d phi / d lambda'_i = 1/4 sum_j sign r_ij^(a-1), r = u / Sr ; d phi / d lambda''_j = -(same, summed over i)
    T f1[3] = {t_const<T>(0.0), t_const<T>(0.0), t_const<T>(0.0)}, f2[3] = {t_const<T>(0.0), t_const<T>(0.0), t_const<T>(0.0)};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        if (!(t_val(u[i][j]) > 0.0)) continue;
        const double sg = (t_val(Dm[i][j]) > 0.0) ? 1.0 : -1.0;
        const T qij = (0.25 * sg) * t_exp((a - 1.0) * (t_log(u[i][j]) - t_log(Sr)));
        f1[i] = f1[i] + qij; f2[j] = f2[j] - qij;
    }
    for (int k = 0; k < 6; ++k) gt[k] = t_const<T>(0.0);
    for (int set = 0; set < 2; ++set)
        for (int i = 0; i < 3; ++i) {
            const T* fx = set ? f2 : f1;
            const T v0 = V[set][0][i], v1 = V[set][1][i], v2 = V[set][2][i];
            const T mxx = v0 * v0, myy = v1 * v1, mzz = v2 * v2;     // gradient of lambda_i: L^T (w o V(v v^T)) (cm::barlat_pull)
            gt[0] = gt[0] + fx[i] * (UL[set][0][0] * mxx + UL[set][1][0] * myy + UL[set][2][0] * mzz);
            gt[3] = gt[3] + fx[i] * (UL[set][0][1] * mxx + UL[set][1][1] * myy + UL[set][2][1] * mzz);
            gt[5] = gt[5] + fx[i] * (UL[set][0][2] * mxx + UL[set][1][2] * myy + UL[set][2][2] * mzz);
            gt[1] = gt[1] + fx[i] * (csh[set][0] * (2.0 * (v0 * v1)));
            gt[2] = gt[2] + fx[i] * (csh[set][2] * (2.0 * (v0 * v2)));
            gt[4] = gt[4] + fx[i] * (csh[set][1] * (2.0 * (v1 * v2)));
        }
}

// effective stress value and 6-vector gradient gt in arithmetic T (the closed forms of yield_eval)
// YK = CM_YIELD_ANY: the surface is chosen at run time from m.yield_kind (a wave-uniform switch).  The kernels built on this
// model (second derivatives, extended parameter sensitivities: cold, one thread per derivative direction) are instantiated
// once per deformation type and model kind instead of once per yield surface as well.
constexpr int CM_YIELD_ANY = -1;
template <int YK, class T>
CM_D void yield_T(const cm_model_desc& m, const MatT<T>& p, const T s[6], T& phi, T gt[6]) {
    if constexpr (YK == CM_YIELD_ANY) {
        switch (m.yield_kind) {
            case CM_YIELD_J2: yield_T<CM_YIELD_J2, T>(m, p, s, phi, gt); break;
            case CM_YIELD_HILL: yield_T<CM_YIELD_HILL, T>(m, p, s, phi, gt); break;
#if defined(CM_HNN_VARIANT) && CM_HNN_VARIANT             // the EXT build of the library: the network surfaces are its only dense ones
            case CM_YIELD_HYBRID_HILL_NN: yield_T<CM_YIELD_HYBRID_HILL_NN, T>(m, p, s, phi, gt); break;
            case CM_YIELD_SCALED_HYBRID_HILL_NN: yield_T<CM_YIELD_SCALED_HYBRID_HILL_NN, T>(m, p, s, phi, gt); break;
            default: yield_T<CM_YIELD_HOSFORD, T>(m, p, s, phi, gt); break;
#else
            case CM_YIELD_HOSFORD: yield_T<CM_YIELD_HOSFORD, T>(m, p, s, phi, gt); break;
            case CM_YIELD_HYBRID_HILL_NN: yield_T<CM_YIELD_HYBRID_HILL_NN, T>(m, p, s, phi, gt); break;
            case CM_YIELD_SCALED_HYBRID_HILL_NN: yield_T<CM_YIELD_SCALED_HYBRID_HILL_NN, T>(m, p, s, phi, gt); break;
            default: yield_T<CM_YIELD_BARLAT, T>(m, p, s, phi, gt); break;
#endif
        }
    } else if constexpr (YK == CM_YIELD_BARLAT) {
        barlat_T<T>(m, p, s, phi, gt);
    } else if constexpr (YK == CM_YIELD_HYBRID_HILL_NN) {
        yield_T<CM_YIELD_HILL, T>(m, p, s, phi, gt);
        T v, g6[6];
        icnn_yield_T<T>(m, p, s, v, g6);
        phi = phi + v;
        for (int k = 0; k < 6; ++k) gt[k] = gt[k] + g6[k];
    } else if constexpr (YK == CM_YIELD_SCALED_HYBRID_HILL_NN) {
        // cm::scaled_hybrid_eval: phi(s) = phi_h(beta s) / beta, beta from the scalar Newton on phi_h(beta s) / Yeq - 1 (same
        // iteration and line search, decided on the values); two more plain Newton steps in arithmetic T at the converged
        // value carry the first and second derivative parts of beta to the implicit-function values
        T pj, gj[6];
        yield_T<CM_YIELD_J2, T>(m, p, s, pj, gj);
        if (!(fabs(t_val(pj)) > 1e-14)) {
            phi = pj;
            for (int k = 0; k < 6; ++k) gt[k] = t_const<T>(0.0);
            return;
        }
        const double iy = 1.0 / m.beta_equivalent_stress;
        auto r_dr = [&](const T& b, T& r, T& dr) {
            T t[6], ph, g[6];
            for (int k = 0; k < 6; ++k) t[k] = b * s[k];
            yield_T<CM_YIELD_HYBRID_HILL_NN, T>(m, p, t, ph, g);
            T gs = g[0] * s[0];
            for (int k = 1; k < 6; ++k) gs = gs + g[k] * s[k];
            r = ph * iy - 1.0; dr = gs * iy;
        };
        T beta = p.Y / pj;
        {
            constexpr int kMaxEvals = 4;
            T C, dC;
            r_dr(beta, C, dC);
            const double n0 = fabs(t_val(C));
            for (int it = 0; it < m.beta_max_iters; ++it) {
                const double nrm = fabs(t_val(C));
                if (nrm / n0 < m.beta_rel_tol || nrm < m.beta_abs_tol) break;
                const T delta = C / dC;
                const double cv = t_val(C), phi0 = 0.5 * cv * cv, dphi0 = -cv * cv, armijo = m.ls_c1 * dphi0;
                int n = 0;
                double alpha = 1.0, best_alpha = 1.0, best_phi = INFINITY;
                T best_C = C, best_dC = dC, Ct = C, dCt = dC;
                bool accepted = false, have_best = false;
                while (n < kMaxEvals && !accepted) {
                    r_dr(beta - alpha * delta, Ct, dCt);
                    const double ph = 0.5 * t_val(Ct) * t_val(Ct);
                    const bool finite = isfinite(ph);
                    if (finite && ph < best_phi) { best_alpha = alpha; best_phi = ph; best_C = Ct; best_dC = dCt; have_best = true; }
                    accepted = finite && (ph <= phi0 + alpha * armijo);
                    const double am = quad_min(phi0, dphi0, alpha, ph);
                    const double ac = fmin(fmax(am, m.ls_lo * alpha), m.ls_hi * alpha);
                    if (!accepted) alpha = finite ? ac : 0.5 * alpha;
                    ++n;
                }
                if (accepted) { beta = beta - alpha * delta; C = Ct; dC = dCt; }
                else if (have_best) { beta = beta - best_alpha * delta; C = best_C; dC = best_dC; }
                else { beta = beta - delta; T Cn; r_dr(beta, Cn, dC); }
            }
            for (int polish = 0; polish < 2; ++polish) {
                r_dr(beta, C, dC);
                beta = beta - C / dC;
            }
        }
        T tau[6], ph, g[6];
        for (int k = 0; k < 6; ++k) tau[k] = beta * s[k];
        yield_T<CM_YIELD_HYBRID_HILL_NN, T>(m, p, tau, ph, g);
        T c = g[0] * tau[0];
        for (int k = 1; k < 6; ++k) c = c + g[k] * tau[k];
        phi = ph / beta;
        for (int k = 0; k < 6; ++k) gt[k] = ph * g[k] / c;
    } else if constexpr (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL) {
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

// out = V(M^T T(a) M) (TRANSPOSE_FIRST) or V(M T(a) M^T), M row-major 3x3 of type T (cm::congruence)
template <bool TRANSPOSE_FIRST, class T>
CM_D void congruence_T(const T* M, const T a[6], T out[6]) {
    const T A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
    T Tm[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T acc = A[i][0] * (TRANSPOSE_FIRST ? M[j] : M[3 * j]);
            for (int k = 1; k < 3; ++k) acc = acc + A[i][k] * (TRANSPOSE_FIRST ? M[3 * k + j] : M[3 * j + k]);
            Tm[i][j] = acc;
        }
    constexpr int I6[6] = {0, 0, 0, 1, 1, 2}, J6[6] = {0, 1, 2, 1, 2, 2};
    for (int r = 0; r < 6; ++r) {
        T acc = (TRANSPOSE_FIRST ? M[I6[r]] : M[3 * I6[r]]) * Tm[0][J6[r]];
        for (int k = 1; k < 3; ++k) acc = acc + (TRANSPOSE_FIRST ? M[3 * k + I6[r]] : M[3 * I6[r] + k]) * Tm[k][J6[r]];
        out[r] = acc;
    }
}

// residual C(x, xp, p) and material stress s in arithmetic T for the total-form model; the material-frame strain `eg`
// and the frame vectors `z` are of type T as well (they depend on the rotation matrix, kinematics_T).
// `plastic` is decided on the primal value (jnp.where semantics, cmad/models/paths.py:26-27).
template <int DEF, int YK, class T>
CM_D void residual_T(const cm_model_desc& m, const MatT<T>& p, const T eg[6], const T* z,
                     const T* x, const T* xp, T* C, T s[6], double dU = 0.0) {
    T e[6];
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        // small_elastic_plastic.py:38-62 as written (no use of Q^T Q = I, so that derivatives w.r.t. the entries of Q are the
        // reference's): global total strain with the off-axis shears of the global plastic strain Q eps_p Q^T, the on-axis
        // entry dU and the off-axis stretches on the diagonal, rotated back to the material frame.  `dU` = grad u.
        const int on = m.uniaxial_idx, ia = (on == 0) ? 1 : 0, ib = (on == 2) ? 1 : 2;
        constexpr int DIAG6[3] = {0, 3, 5};
        T v6[6], gp6[6], em6[6];
        for (int k = 0; k < 6; ++k) v6[k] = x[k];
        congruence_T<false, T>(p.Q, v6, gp6);                  // Q eps_p Q^T
        gp6[DIAG6[on]] = t_const<T>(dU);
        gp6[DIAG6[ia]] = x[7] - 1.0;
        gp6[DIAG6[ib]] = x[8] - 1.0;
        congruence_T<true, T>(p.Q, gp6, em6);                  // Q^T (constrained global total strain) Q
        for (int k = 0; k < 6; ++k) e[k] = em6[k] - x[k];
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
    yield_T<YK, T>(m, p, s, phi, gt);
    const T H = hardening_T<T>(m, p, x[6]);
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
        for (int k = 0; k < 6; ++k) r = r + kW[k] * (z[k] * s[k]);
        C[7] = r / twomu;
    }
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        T ra = T{0.0}, rb = T{0.0};
        for (int k = 0; k < 6; ++k) { ra = ra + kW[k] * (z[6 + k] * s[k]); rb = rb + kW[k] * (z[12 + k] * s[k]); }
        C[7] = ra / twomu; C[8] = rb / twomu;
    }
}

// the rate-form residual (cm::residual_rate, cmad/models/small_rate_elastic_plastic.py:249-346) in arithmetic T:
// the unknown x[0:6] is the material stress itself, `deg` the material strain increment (plain doubles).
template <int DEF, int YK, class T>
CM_D void residual_rate_T(const cm_model_desc& m, const MatT<T>& p, const T deg[6], const T* z,
                          const T* x, const T* xp, T* C, T s[6]) {
    static_assert(DEF != CM_UNIAXIAL_STRESS, "rate form: FULL_3D and PLANE_STRESS");
    T e[6];
    for (int k = 0; k < 6; ++k) {
        s[k] = x[k];
        e[k] = deg[k];
        if constexpr (DEF == CM_PLANE_STRESS) e[k] = e[k] + z[k] * (x[7] - xp[7]);
    }
    const T tr = e[0] + e[3] + e[5];
    const T twomu = 2.0 * p.mu;
    T phi, gt[6];
    yield_T<YK, T>(m, p, s, phi, gt);
    const T H = hardening_T<T>(m, p, x[6]);
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
        if constexpr (DEF == CM_PLANE_STRESS) r7 = r7 + kW[k] * (z[k] * dc);
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
CM_D void residual_rate_uniaxial_T(const cm_model_desc& m, const MatT<T>& p, const T& dU,
                                   const T* x, const T* xp, T* C, T s[6]) {
    const int on = m.uniaxial_idx, ia = (on == 0) ? 1 : 0, ib = (on == 2) ? 1 : 2;
    T eg[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) eg[i][j] = T{0.0};
    eg[on][on] = dU;
    eg[ia][ia] = x[7] - xp[7];
    eg[ib][ib] = x[8] - xp[8];
    eg[0][1] = eg[1][0] = x[9]; eg[0][2] = eg[2][0] = x[10]; eg[1][2] = eg[2][1] = x[11];
    // material frame: e = Q^T eg Q  (Q = m.Q row-major, Q_ij = e_i(global) . e_j(material))
    T em[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        T acc = T{0.0};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) acc = acc + (p.Q[3 * a + i] * p.Q[3 * b + j]) * eg[a][b];
        em[i][j] = acc;
    }
    const T e[6] = {em[0][0], em[0][1], em[0][2], em[1][1], em[1][2], em[2][2]};
    for (int k = 0; k < 6; ++k) s[k] = x[k];
    const T tr = e[0] + e[3] + e[5];
    const T twomu = 2.0 * p.mu;
    T phi, gt[6];
    yield_T<YK, T>(m, p, s, phi, gt);
    const T H = hardening_T<T>(m, p, x[6]);
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
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) acc = acc + (p.Q[3 * i + a] * p.Q[3 * j + b]) * dm[a][b];
        dgl[i][j] = acc;
    }
    C[7] = dgl[ia][ia] / twomu; C[8] = dgl[ib][ib] / twomu;
    C[9] = dgl[0][1] / twomu; C[10] = dgl[0][2] / twomu; C[11] = dgl[1][2] / twomu;
}

// ---- the whole model in arithmetic T: kinematics with the rotation matrix, residual, global Cauchy stress -----------------
// material-frame total strain and frame vectors (cm::strain_from_gradu, cm::strain_z, cm::uniaxial_frame) for a rotation
// matrix of type T; G: grad u (rate form: grad u - grad u_prev), plain doubles
template <int DEF, class T>
CM_D void kinematics_T(const cm_model_desc& m, const MatT<T>& p, const double* G, T eg[6], T* z /* Dims<DEF>::NZ */) {
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        const int on = m.uniaxial_idx, ia = (on == 0) ? 1 : 0, ib = (on == 2) ? 1 : 2;
        const int rows[3] = {on, ia, ib};
        for (int r = 0; r < 3; ++r) {
            const T* q = p.Q + 3 * rows[r];
            T* Z = z + 6 * r;
            Z[0] = q[0] * q[0]; Z[1] = q[0] * q[1]; Z[2] = q[0] * q[2]; Z[3] = q[1] * q[1]; Z[4] = q[1] * q[2]; Z[5] = q[2] * q[2];
        }
        for (int k = 0; k < 6; ++k) eg[k] = G[0] * z[k];
    } else {
        T E[6];
        if constexpr (DEF == CM_FULL_3D) {
            E[0] = t_const<T>(G[0]); E[1] = t_const<T>(0.5 * (G[1] + G[3])); E[2] = t_const<T>(0.5 * (G[2] + G[6]));
            E[3] = t_const<T>(G[4]); E[4] = t_const<T>(0.5 * (G[5] + G[7])); E[5] = t_const<T>(G[8]);
        } else {
            E[0] = t_const<T>(G[0]); E[1] = t_const<T>(0.5 * (G[1] + G[2])); E[2] = t_const<T>(0.0);
            E[3] = t_const<T>(G[3]); E[4] = t_const<T>(0.0); E[5] = t_const<T>(0.0);
        }
        congruence_T<true, T>(p.Q, E, eg);
        const T* q = p.Q + 6;                                   // third row: d(material strain)/dF33 = V(q3 q3^T)
        z[0] = q[0] * q[0]; z[1] = q[0] * q[1]; z[2] = q[0] * q[2]; z[3] = q[1] * q[1]; z[4] = q[1] * q[2]; z[5] = q[2] * q[2];
    }
}

// C(x, xp, p) and the GLOBAL Cauchy stress sg(x, p) of either model kind in arithmetic T, everything (rotation matrix,
// yield-surface coefficients, network weights included) taken from p.  G as kinematics_T.
template <int DEF, int YK, int MK, class T>
CM_D void model_eval_T(const cm_model_desc& m, const MatT<T>& p, const double* G, const T* x, const T* xp, T* C, T sg[6]) {
    T s[6];
    if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS) {
        residual_rate_uniaxial_T<YK, T>(m, p, t_const<T>(G[0]), x, xp, C, s);
    } else {
        T eg[6], z[Dims<DEF>::NZ];
        kinematics_T<DEF, T>(m, p, G, eg, z);
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) residual_rate_T<DEF, YK, T>(m, p, eg, z, x, xp, C, s);
        else residual_T<DEF, YK, T>(m, p, eg, z, x, xp, C, s, G[0]);
    }
    congruence_T<false, T>(p.Q, s, sg);
}

// one (a, b) pair: out_C[NX] = d2 C / dq_a dq_b, out_S[6] = d2 sigma_global / dq_a dq_b,
// and the first derivatives wrt q_a (for cross-checks): out_Ca[NX], out_Sa[6]
// MK = CM_SMALL_RATE_ELASTIC_PLASTIC: G must already hold grad u - grad u_prev (the strain is linear in it).
// (ROT is kept in the signature for the callers; the rotation products always run here: Q = I reproduces the plain result.)
template <int DEF, int YK, bool ROT, int MK = CM_SMALL_ELASTIC_PLASTIC>
CM_D void hessian_pair(const cm_model_desc& m, const double* G, const double* xv, const double* xpv, int a, int b,
                       double* out_C, double* out_S, double* out_Ca, double* out_Sa,
                       double* out_C0 = nullptr, double* out_S0 = nullptr, double* out_Sb = nullptr,
                       const int32_t* ep_index = nullptr) {
    constexpr int NX = nx_of<DEF, MK>();
    HD x[NX], xp[NX], C[NX], sg[6];
    MatT<HD> p;
    mat_from_desc<HD>(m, p);
    for (int k = 0; k < NX; ++k) { x[k] = hd(xv[k]); xp[k] = hd(xpv[k]); }
    // q = [xi, xi_prev, p (KP), extended parameters ep_index[0 .. n_ep)]: direction a in the first derivative slot, b in the second
    auto seed = [&](int i, bool second) {
        HD* t = nullptr;
        if (i < NX) t = &x[i];
        else if (i < 2 * NX) t = &xp[i - NX];
        else {
            int e = i - 2 * NX;
            if (e >= CM_NUM_PARAMS) e = ep_index[e - CM_NUM_PARAMS];          // EP index (mat_seed): yc[6..18], Q, network weights
            if (e == CM_P_LAMBDA) t = &p.lambda;
            else if (e == CM_P_MU) t = &p.mu;
            else if (e == CM_P_Y) t = &p.Y;
            else if (e == CM_P_VOCE_S) t = &p.S;
            else if (e == CM_P_VOCE_D) t = &p.D;
            else if (e == CM_P_LIN_K) t = &p.K;
            else if (e < CM_EP_Q0) t = &p.yc[e - CM_P_YC0];
            else if (e < CM_EP_NN0) t = &p.Q[e - CM_EP_Q0];
            else { if (second) p.nn_seed2 = e - CM_EP_NN0; else p.nn_seed = e - CM_EP_NN0; }
        }
        if (t) { if (second) t->b = 1.0; else t->a = 1.0; }
    };
    seed(a, false);
    seed(b, true);
    model_eval_T<DEF, YK, MK, HD>(m, p, G, x, xp, C, sg);
    for (int k = 0; k < NX; ++k) { out_C[k] = C[k].ab; out_Ca[k] = C[k].a; }
    for (int k = 0; k < 6; ++k) { out_S[k] = sg[k].ab; out_Sa[k] = sg[k].a; }
    if (out_Sb) for (int k = 0; k < 6; ++k) out_Sb[k] = sg[k].b;
    if (out_C0) for (int k = 0; k < NX; ++k) out_C0[k] = C[k].v;
    if (out_S0) for (int k = 0; k < 6; ++k) out_S0[k] = sg[k].v;
}

// ---- first-order sensitivities w.r.t. ONE extended parameter (EP index e, see mat_seed) at a given state ---------------
// dC[NX] = dC/dp_e, dS[6] = d sigma_global/dp_e by forward-mode evaluation of the whole model: the way every parameter the
// hand-derived kernels have no closed form for (rotation matrix, Hosford exponent, Hill coefficients of the network
// surfaces, network weights) is differentiated -- the reference's jacrev over the params pytree
// (cmad/models/model.py:125-153, cmad/parameters/parameters.py:368-377).
template <int DEF, int YK, int MK>
CM_D void param_direction(const cm_model_desc& m, const double* G, const double* xv, const double* xpv, int e,
                          double* dC, double* dS) {
    constexpr int NX = nx_of<DEF, MK>();
    D1 x[NX], xp[NX], C[NX], sg[6];
    MatT<D1> p;
    mat_from_desc<D1>(m, p);
    mat_seed<D1>(p, e);
    for (int k = 0; k < NX; ++k) { x[k] = d1(xv[k]); xp[k] = d1(xpv[k]); }
    model_eval_T<DEF, YK, MK, D1>(m, p, G, x, xp, C, sg);
    for (int k = 0; k < NX; ++k) dC[k] = C[k].d;
    for (int k = 0; k < 6; ++k) dS[k] = sg[k].d;
}

// ---- one entry of the second-order weight matrix of a history step (cm_hessian_history) ---------------------------------
// W[a][b] = sbar . d2 sigma / dq_a dq_b + sum_r hss_r d sigma_r / dq_a  d sigma_r / dq_b - lam . d2 C / dq_a dq_b,
// q = [xi, xi_prev, p]: the Hessian in q of the step's Lagrangian J_k(sigma) - lam . C_k for a QoI whose curvature in the six
// stored stress entries is diagonal (cmad/qois/calibration.py:56-66: hss = folded squared weights), with lam = -phi of
// cmad/objectives/mp_objective.py:255-281.  G: grad u (rate form: grad u - grad u_prev).
template <int DEF, int YK, bool ROT, int MK>
CM_D double hessian_weight(const cm_model_desc& m, const double* G, const double* x, const double* xp, const double* lam,
                           const double sbar[6], const double hss[6], int a, int b, const int32_t* ep_index = nullptr) {
    constexpr int NX = nx_of<DEF, MK>();
    double oC[NX], oS[6], oCa[NX], oSa[6], oSb[6];
    hessian_pair<DEF, YK, ROT, MK>(m, G, x, xp, a, b, oC, oS, oCa, oSa, nullptr, nullptr, oSb, ep_index);
    double w = 0.0;
    for (int k = 0; k < NX; ++k) w -= lam[k] * oC[k];
    for (int r = 0; r < 6; ++r) w += sbar[r] * oS[r] + hss[r] * oSa[r] * oSb[r];
    return w;
}

// ---- forward sensitivity column of one EXTENDED parameter (EP index e) at a converged step ---------------------------------
// d = dxi/dp_e = -A^-1 (dC/dp_e + dC/dxi_prev d_prev): cm::direct_column with the parameter column taken from the forward-mode
// evaluation of the whole model (param_direction) instead of the closed-form block -- what the second-order pass needs for the
// leaves outside the 12 native parameters (D_k = dq_k/dp in cm_hessian_history_ep).  FULL_3D / PLANE_STRESS / UNIAXIAL_STRESS
// (total form), FULL_3D / PLANE_STRESS (rate form).
template <int MK, int DEF, int YK, bool ROT>
CM_D bool direct_column_ep(const cm_model_desc& m, const double* G, const double* Gp, const double* x, const double* xp, int e,
                           const double* d_prev, double* d) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    static_assert(!(MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS), "12-dof rate form: not built");
    double C[NX], sg[6], rhs[NX], dS[6], Geff[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) Geff[k] = (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) ? G[k] - Gp[k] : G[k];
    param_direction<DEF, YK, MK>(m, Geff, x, xp, e, rhs, dS);
#pragma unroll
    for (int i = 0; i < NX; ++i) rhs[i] = -rhs[i];
    if (d_prev) {
        double Axp[NX * NX];
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) evaluate_blocks_rate<DEF, YK, ROT>(m, G, Gp, x, xp, CM_W_XI_PREV, C, Axp, sg, nullptr);
        else evaluate_blocks<DEF, YK, ROT>(m, G, x, xp, CM_W_XI_PREV, C, Axp, sg, nullptr);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) t += Axp[i * NX + k] * d_prev[k];
            rhs[i] -= t;
        }
    }
    double Ax[NX * NX], A[NX][NX];
    if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) evaluate_blocks_rate<DEF, YK, ROT>(m, G, Gp, x, xp, CM_W_XI, C, Ax, sg, nullptr);
    else evaluate_blocks<DEF, YK, ROT>(m, G, x, xp, CM_W_XI, C, Ax, sg, nullptr);
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int k = 0; k < NX; ++k) A[i][k] = Ax[i * NX + k];
    const bool ok = lu_factor<NX>(A);
    lu_subst<NX>(A, rhs);
#pragma unroll
    for (int i = 0; i < NX; ++i) d[i] = rhs[i];
    return ok;
}

// ---- complex-step model instances: the imperative Newton on the complex residual ------------------------------------------
// The reference builds `Model(parameters, def_type, is_complex=True)` (small_elastic_plastic.py:118-127,
// small_rate_elastic_plastic.py:125-134) so that its tests can take Im J(p + i h d) / h as an AD-free directional derivative
// (tests/objectives/test_J2_fd_checks.py:163-235, 355-386): newton_solve (nonlinear_solver.py:14-85) then iterates on complex
// states with the holomorphic Jacobian.  Here: the arithmetic-T model with T = CX for the residual and T = DC (duals over CX,
// one column per evaluation) for d C / d x, an unpivoted complex LU, the reference's stopping rule on ||C||_2.
// p_im[CM_NUM_PARAMS]: imaginary parts of the native parameters (KP order); the real parts are the model description's.
// ext_im (may be null): imaginary parts of the extended parameters in EP order from CM_EP_YC6 on -- yc[6..18] (13), Q row-major
// (9), then every packed network weight (indexed like cm_model_desc.nn_weights).
constexpr int kCxExtNN0 = CM_EP_NN0 - CM_EP_YC6;                 // 22: where the network weights start in ext_im
template <class T>
CM_D void mat_add_imag(MatT<T>& p, const double* p_im, const double* ext_im) {
    t_add_imag(p.lambda, p_im[CM_P_LAMBDA]); t_add_imag(p.mu, p_im[CM_P_MU]); t_add_imag(p.Y, p_im[CM_P_Y]);
    t_add_imag(p.S, p_im[CM_P_VOCE_S]); t_add_imag(p.D, p_im[CM_P_VOCE_D]); t_add_imag(p.K, p_im[CM_P_LIN_K]);
    for (int k = 0; k < CM_NUM_PARAMS - CM_P_YC0; ++k) t_add_imag(p.yc[k], p_im[CM_P_YC0 + k]);
    if (ext_im) {
        for (int k = 0; k < 13; ++k) t_add_imag(p.yc[6 + k], ext_im[k]);
        for (int k = 0; k < 9; ++k) t_add_imag(p.Q[k], ext_im[13 + k]);
        p.nn_im = ext_im + kCxExtNN0;
    }
}

// x: in = the starting iterate, out = the returned state; C, sg: residual and global Cauchy stress at the returned state.
// G: grad u (rate form: grad u - grad u_prev).  max_iters = 0 evaluates C and sg at x.  Returns the status word.
template <int DEF, int YK, int MK>
CM_D uint32_t newton_cx(const cm_model_desc& m, const double* p_im, const double* ext_im, const double* G, const CX* xp, CX* x, CX* C,
                        CX sg[6]) {
    constexpr int NX = nx_of<DEF, MK>();
    MatT<CX> p;
    mat_from_desc<CX>(m, p);
    mat_add_imag<CX>(p, p_im, ext_im);
    MatT<DC> pd;
    mat_from_desc<DC>(m, pd);
    mat_add_imag<DC>(pd, p_im, ext_im);
    auto norm2 = [&]() { double n = 0.0; for (int k = 0; k < NX; ++k) n += C[k].re * C[k].re + C[k].im * C[k].im; return n; };
    model_eval_T<DEF, YK, MK, CX>(m, p, G, x, xp, C, sg);
    const double n0 = norm2(), rel2 = m.rel_tol * m.rel_tol * n0, abs2 = m.abs_tol * m.abs_tol;
    uint32_t flags = 0;
    int it = 0;
    for (;;) {
        const double nsq = norm2();
        if ((nsq < rel2) || (nsq < abs2)) { flags |= CM_STATUS_CONVERGED; break; }
        if (it >= m.max_iters) break;
        CX A[NX][NX], delta[NX];
        {
            DC xd[NX], xpd[NX], Cd[NX], sgd[6];
            for (int k = 0; k < NX; ++k) { xd[k] = DC{x[k], CX{0.0, 0.0}}; xpd[k] = DC{xp[k], CX{0.0, 0.0}}; }
            for (int j = 0; j < NX; ++j) {
                xd[j].d = CX{1.0, 0.0};
                model_eval_T<DEF, YK, MK, DC>(m, pd, G, xd, xpd, Cd, sgd);
                xd[j].d = CX{0.0, 0.0};
                for (int k = 0; k < NX; ++k) A[k][j] = Cd[k].d;
            }
        }
        for (int k = 0; k < NX; ++k) {                              // LU without pivoting (cm::lu_factor), complex
            const CX piv = A[k][k];
            if (!(piv.re * piv.re + piv.im * piv.im > 1e-300)) flags |= CM_STATUS_SINGULAR;
            const CX ip = cx_inv(piv);
            A[k][k] = ip;
            for (int r = k + 1; r < NX; ++r) {
                const CX l = A[r][k] * ip;
                A[r][k] = l;
                for (int c = k + 1; c < NX; ++c) A[r][c] = A[r][c] - l * A[k][c];
            }
        }
        for (int k = 0; k < NX; ++k) delta[k] = C[k];
        for (int k = 0; k < NX; ++k)
            for (int r = k + 1; r < NX; ++r) delta[r] = delta[r] - A[r][k] * delta[k];
        for (int k = NX - 1; k >= 0; --k) {
            CX t = delta[k];
            for (int c = k + 1; c < NX; ++c) t = t - A[k][c] * delta[c];
            delta[k] = t * A[k][k];
        }
        for (int k = 0; k < NX; ++k) x[k] = x[k] - delta[k];
        ++it;
        model_eval_T<DEF, YK, MK, CX>(m, p, G, x, xp, C, sg);
    }
    return flags | (uint32_t)it;
}

}  // namespace cm
