// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see cmad_oracle.cpp).
//
// Forward-mode dual numbers, nestable (Dual<Dual<double,M>,N>), so the oracle
// can differentiate the reference's residual the way the reference does:
// `jax.grad(effective_stress)` inside the residual
// (cmad/models/small_elastic_plastic.py:90) wrapped by `jax.jacfwd(residual)`
// (cmad/models/model.py:125-131, cmad/models/nonlinear_solver.py:122).
// No hand-derived derivative appears anywhere in the oracle.
#pragma once
#include <cmath>

template <class S, int N>
struct Dual {
    S v;
    S d[N];
    Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
    Dual(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
    Dual(const S& c, int) : v(c) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
};

// ---- value extraction (strip all derivative levels) ----
inline double val(double x) { return x; }
template <class S, int N> inline double val(const Dual<S, N>& x) { return val(x.v); }

// ---- lift a lower-level scalar into a higher-level one ----
template <class T> struct Lift;
template <> struct Lift<double> { static double from(double c) { return c; } };
template <class S, int N> struct Lift<Dual<S, N>> {
    static Dual<S, N> from(const S& c) { return Dual<S, N>(c, 0); }
};

// ---- arithmetic ----
#define DUAL_T template <class S, int N>
#define DU Dual<S, N>

DUAL_T inline DU operator+(const DU& a, const DU& b) {
    DU r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
DUAL_T inline DU operator-(const DU& a, const DU& b) {
    DU r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
DUAL_T inline DU operator-(const DU& a) {
    DU r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
DUAL_T inline DU operator*(const DU& a, const DU& b) {
    DU r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
DUAL_T inline DU operator/(const DU& a, const DU& b) {
    DU r; r.v = a.v / b.v;
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r; }

DUAL_T inline DU operator+(const DU& a, double b) { DU r = a; r.v = a.v + b; return r; }
DUAL_T inline DU operator+(double b, const DU& a) { return a + b; }
DUAL_T inline DU operator-(const DU& a, double b) { DU r = a; r.v = a.v - b; return r; }
DUAL_T inline DU operator-(double b, const DU& a) { return (-a) + b; }
DUAL_T inline DU operator*(const DU& a, double b) {
    DU r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
DUAL_T inline DU operator*(double b, const DU& a) { return a * b; }
DUAL_T inline DU operator/(const DU& a, double b) {
    DU r; r.v = a.v / b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / b; return r; }
DUAL_T inline DU operator/(double b, const DU& a) { return DU(b) / a; }
DUAL_T inline DU& operator+=(DU& a, const DU& b) { a = a + b; return a; }
DUAL_T inline DU& operator-=(DU& a, const DU& b) { a = a - b; return a; }

// comparisons act on the primal value (jnp.where / lax.cond semantics)
DUAL_T inline bool operator>(const DU& a, double b) { return val(a) > b; }
DUAL_T inline bool operator<(const DU& a, double b) { return val(a) < b; }
DUAL_T inline bool operator==(const DU& a, double b) { return val(a) == b; }

// ---- elementary functions ----
inline double dsqrt(double x) { return std::sqrt(x); }
inline double dexp(double x) { return std::exp(x); }
inline double dlog(double x) { return std::log(x); }
inline double dabs(double x) { return std::fabs(x); }
inline double dpow(double x, double y) { return std::pow(x, y); }
// jax.nn.softplus = logaddexp(x, 0)
inline double dsoftplus(double x) { return std::fmax(x, 0.0) + std::log1p(std::exp(-std::fabs(x))); }
inline double dsigmoid(double x) { return 1.0 / (1.0 + std::exp(-x)); }

DUAL_T inline DU dsqrt(const DU& a) {
    DU r; r.v = dsqrt(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / (r.v * 2.0);
    return r; }
DUAL_T inline DU dexp(const DU& a) {
    DU r; r.v = dexp(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v;
    return r; }
DUAL_T inline DU dlog(const DU& a) {
    DU r; r.v = dlog(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / a.v;
    return r; }
// jnp.abs: derivative sign(x), 0 at x == 0
DUAL_T inline DU dabs(const DU& a) {
    double x = val(a);
    double s = (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0);
    DU r; r.v = dabs(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s;
    return r; }
DUAL_T inline DU dsigmoid(const DU& a) {
    DU r; r.v = dsigmoid(a.v);
    S g = r.v * (S(1.0) - r.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g;
    return r; }
DUAL_T inline DU dsoftplus(const DU& a) {
    DU r; r.v = dsoftplus(a.v);
    S g = dsigmoid(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g;
    return r; }
// jnp.power(x, y), both differentiable. JAX's rules (jax/_src/lax/lax.py
// _pow_jvp_lhs/_pow_jvp_rhs, third-party, unpinned): d/dx = y * x**(y-1)
// (exponent replaced by 1 when y == 0), d/dy = log(x == 0 ? 1 : x) * x**y.
DUAL_T inline DU dpow(const DU& x, const DU& y) {
    DU r; r.v = dpow(x.v, y.v);
    S ym1 = (val(y) == 0.0) ? S(1.0) : (y.v - 1.0);
    S gx = y.v * dpow(x.v, ym1);
    S xs = (val(x) == 0.0) ? S(1.0) : x.v;
    S gy = dlog(xs) * r.v;
    for (int i = 0; i < N; ++i) r.d[i] = x.d[i] * gx + y.d[i] * gy;
    return r; }

#undef DUAL_T
#undef DU
