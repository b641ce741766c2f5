// Device-side math of the per-Gauss-point stress update (gfx950, fp64, one point per lane).
//
// Everything here is hand-derived closed form -- no AD on the device.  The formulas are checked
// against the dual-number oracle (oracle/) and torch.func AD in tests/.  Notation:
//   6-vectors are CMAD's un-weighted [xx,xy,xz,yy,yz,zz] (cmad/models/var_types.py:43-84);
//   w  = [1,2,2,1,2,1]  multiplicity of a slot in a full 3x3 contraction;
//   d  = [1,0,0,1,0,1]  diagonal indicator;
//   gt = d phi~/d s_k   gradient of the effective stress w.r.t. the 6-vector (= w_k * n_k, n the
//        reference's 3x3 yield normal jax.grad(effective_stress), small_elastic_plastic.py:90);
//   Ht = d^2 phi~/d s_k d s_l  (6x6 symmetric).
#pragma once
#include <stdint.h>
#include "../../include/cmad_hip.h"

#if defined(CM_HOST_BUILD)
// Host compilation of the same per-point math, used ONLY by tests/native (CPU sanitizer / debug builds
// of the kernel arithmetic).  A "wave" is a single lane there.
#include <cmath>
#define CM_D inline
static inline bool __any(bool p) { return p; }
using std::fabs; using std::fmax; using std::fmin; using std::sqrt; using std::exp; using std::pow; using std::isfinite; using std::log1p; using std::log;
#else
#include <hip/hip_runtime.h>
#define CM_D __device__ __forceinline__
#endif

// CM_HNN = 1: this build carries the rarely used network features -- the network hardening law and input-convex networks with
// more than one hidden layer (scratch-resident general evaluation).  The library's BASE build sets 0 (cmad_hip.hip: neither may
// cost the common configurations an instruction or a register); its EXT build and the host build of the tests set 1.
#ifndef CM_HNN
#define CM_HNN 1
#endif

namespace cm {

// Read-only, wave-uniform device data (the network weights): through the CONSTANT address space the loads are s_load_* into
// scalar registers.  Through a generic pointer the compiler cannot prove that the kernel's own stores do not alias the data
// and emits a vector load per lane of the same address plus a vmcnt wait inside the loop over the hidden units.
#if defined(CM_HOST_BUILD)
typedef const double* cm_uniform_ptr;
#else
typedef const __attribute__((address_space(4))) double* cm_uniform_ptr;
#endif
CM_D cm_uniform_ptr uniform_ptr(const double* p) { return (cm_uniform_ptr)p; }

// 1/a without the IEEE division sequence: v_rcp_f64 (~2^-26 relative) + two Newton steps -> ~1 ulp.
// Used where a is a well-scaled, non-zero quantity (phi, pivots, determinants); 5 instructions instead of ~12.
CM_D double rcp(double a) {
#if defined(CM_HOST_BUILD)
    return 1.0 / a;
#else
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    return r;
#endif
}

// 1/sqrt(a) for a well-scaled a > 0: v_rsq_f64 + two Newton steps (~1 ulp); one result gives sqrt(a) = a r, 1/a = r^2
CM_D double rsqrt_pos(double a) {
#if defined(CM_HOST_BUILD)
    return 1.0 / std::sqrt(a);
#else
    double r = __builtin_amdgcn_rsq(a);
    r = __builtin_fma(0.5 * r, __builtin_fma(-a * r, r, 1.0), r);
    r = __builtin_fma(0.5 * r, __builtin_fma(-a * r, r, 1.0), r);
    return r;
#endif
}

// A literal that stays in a scalar register pair: the fma that consumes it reads it as its one scalar operand, instead
// of the v_mov_b32 pair + v_fmac_f64 the compiler emits for a literal addend (two VALU instructions per coefficient).
#if defined(CM_HOST_BUILD)
#define CM_SCALAR(c) (c)
#else
#define CM_SCALAR(c) ([] { double c_ = (c); asm("" : "+s"(c_)); return c_; }())
#endif

// exp(x) in ~22 VALU instructions instead of the math library's 44: x = k ln2 + r with |r| <= ln2 / 2 (Cody-Waite, two
// constants), e^r by its Taylor polynomial of degree 13 (truncation 4e-18 relative), scaled by 2^k with one ldexp (which
// also delivers overflow to inf and gradual underflow to 0).  Below 1 ulp over the range the hardening laws use; the
// argument is clamped to +-1100 so that k fits an int.  tests/test_host_math.py::test_exp_s checks it against libm.
// EXP_CLAMP = false: the caller guarantees |x| <= 700 (no clamp instructions)
template <bool EXP_CLAMP = true>
CM_D double exp_s(double x) {
    if constexpr (EXP_CLAMP) x = fmax(fmin(x, 1100.0), -1100.0);
    const double k = __builtin_rint(x * CM_SCALAR(0x1.71547652b82fep+0));
    double r = __builtin_fma(k, CM_SCALAR(-0x1.62e42fefa39efp-1), x);
    r = __builtin_fma(k, CM_SCALAR(-0x1.abc9e3b39803fp-56), r);
    double p = CM_SCALAR(1.0 / 6227020800.0);
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 479001600.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 39916800.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 3628800.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 362880.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 40320.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 5040.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 720.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 120.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 24.0));
    p = __builtin_fma(r, p, CM_SCALAR(1.0 / 6.0));
    p = __builtin_fma(r, p, 0.5);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// 1 / (2 mu), the residual's scale factor (elastic_stress.py:71-72); every other division by mu is a product with it
CM_D double half_over_mu(const cm_model_desc& m) { return 0.5 / m.mu; }   // one expression everywhere: a single division per kernel

constexpr double kIW[6] = {1.0, 0.5, 0.5, 1.0, 0.5, 1.0};   // 1 / w_k
constexpr double kW[6] = {1.0, 2.0, 2.0, 1.0, 2.0, 1.0};
constexpr bool kDiag[6] = {true, false, false, true, false, true};
// Symmetric 6x6 matrices (the yield-surface Hessians) are built as their packed upper triangle, row-major:
// entry (k, l), any order, sits at sym6(k, l); 21 doubles instead of 36.
constexpr int sym6(int k, int l) { return (k <= l) ? (k * (11 - k)) / 2 + l : (l * (11 - l)) / 2 + k; }
CM_D void sym6_expand(const double Hp[21], double Ht[6][6]) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int l = 0; l < 6; ++l) Ht[k][l] = Hp[sym6(k, l)];
}

template <int DEF> struct Dims;
// NZ: length of the per-kernel frame array `z` (see strain_z)
template <> struct Dims<CM_FULL_3D> { static constexpr int NX = 7, NU = 9, NE = 0, NZ = 6; };
template <> struct Dims<CM_PLANE_STRESS> { static constexpr int NX = 8, NU = 4, NE = 1, NZ = 6; };
template <> struct Dims<CM_UNIAXIAL_STRESS> { static constexpr int NX = 9, NU = 1, NE = 2, NZ = 18; };

// ---- symmetric 3x3 congruences on 6-vectors --------------------------------------------------
// out = V(M^T T(a) M) (TRANSPOSE_FIRST) or V(M T(a) M^T)
template <bool TRANSPOSE_FIRST>
CM_D void congruence(const double* M, const double a[6], double out[6]) {
    const double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
    double T[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += A[i][k] * (TRANSPOSE_FIRST ? M[3 * k + j] : M[3 * j + k]);
            T[i][j] = s;                       // A M   or   A M^T
        }
    constexpr int I6[6] = {0, 0, 0, 1, 1, 2}, J6[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += (TRANSPOSE_FIRST ? M[3 * k + I6[r]] : M[3 * I6[r] + k]) * T[k][J6[r]];
        out[r] = s;                            // M^T A M   or   M A M^T
    }
}

// total strain 6-vector in the material frame from grad u
// (cmad/models/small_elastic_plastic.py:38-62, kinematics.py:10-26; the out-of-plane stretch of
//  PLANE_STRESS enters separately through z, see strain_z)
// UNIAXIAL_STRESS frame vectors: Z^i = V(q_i q_i^T), q_i = row i of Q, ordered [on-axis, first off-axis,
// second off-axis] (cmad/models/kinematics.py:35-50,62-65).  With them the constrained kinematics of
// small_elastic_plastic.py:47-60 (off-axis shear of the global total strain := that of the global plastic strain)
// collapse to  e = sum_i Z^i (eps_i - (w o Z^i) . v): the global elastic strain is diagonal.
template <bool ROT>
CM_D void uniaxial_frame(const cm_model_desc& m, double z[18]) {
    const int on = m.uniaxial_idx, a = (on == 0) ? 1 : 0, b = (on == 2) ? 1 : 2;
    const int rows[3] = {on, a, b};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int i = rows[r];
        double q0, q1, q2;
        if constexpr (ROT) {
            q0 = (i == 0) ? m.Q[0] : ((i == 1) ? m.Q[3] : m.Q[6]);
            q1 = (i == 0) ? m.Q[1] : ((i == 1) ? m.Q[4] : m.Q[7]);
            q2 = (i == 0) ? m.Q[2] : ((i == 1) ? m.Q[5] : m.Q[8]);
        } else {
            q0 = (i == 0) ? 1.0 : 0.0; q1 = (i == 1) ? 1.0 : 0.0; q2 = (i == 2) ? 1.0 : 0.0;
        }
        double* Z = z + 6 * r;
        Z[0] = q0 * q0; Z[1] = q0 * q1; Z[2] = q0 * q2; Z[3] = q1 * q1; Z[4] = q1 * q2; Z[5] = q2 * q2;
    }
}

template <int DEF, bool ROT>
CM_D void strain_from_gradu(const cm_model_desc& m, const double* G, double eg[6]) {
    double E[6];
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        double z[18];
        uniaxial_frame<ROT>(m, z);
#pragma unroll
        for (int k = 0; k < 6; ++k) eg[k] = G[0] * z[k];       // on-axis strain along Z^on
        return;
    } else if constexpr (DEF == CM_FULL_3D) {
        E[0] = G[0]; E[1] = 0.5 * (G[1] + G[3]); E[2] = 0.5 * (G[2] + G[6]);
        E[3] = G[4]; E[4] = 0.5 * (G[5] + G[7]); E[5] = G[8];
    } else {
        E[0] = G[0]; E[1] = 0.5 * (G[1] + G[2]); E[2] = 0.0; E[3] = G[3]; E[4] = 0.0; E[5] = 0.0;
    }
    if constexpr (ROT) congruence<true>(m.Q, E, eg);
    else {
#pragma unroll
        for (int k = 0; k < 6; ++k) eg[k] = E[k];
    }
}

// PLANE_STRESS: d(material strain)/d F33 = V(Q^T e3 e3^T Q) = V(q3 q3^T), q3 = third row of Q
template <int DEF, bool ROT>
CM_D void strain_z(const cm_model_desc& m, double* z /* Dims<DEF>::NZ */) {
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        uniaxial_frame<ROT>(m, z);
    } else if constexpr (ROT) {
        const double a = m.Q[6], b = m.Q[7], c = m.Q[8];
        z[0] = a * a; z[1] = a * b; z[2] = a * c; z[3] = b * b; z[4] = b * c; z[5] = c * c;
    } else {
        z[0] = z[1] = z[2] = z[3] = z[4] = 0.0; z[5] = 1.0;
    }
}

// ---- effective stress: value, gradient gt[6], Hessian Ht[6][6] --------------------------------
// J2   cmad/models/effective_stress.py:30-37   phi = sqrt(s^T A s), A = 3/2 (W - d d^T / 3)
// Hill cmad/models/effective_stress.py:40-52   phi = sqrt(s^T A s), A from F,G,H,L,M,N
struct QuadForm { double a00, a33, a55, a03, a05, a35, a11, a22, a44; };

template <int YK>
CM_D QuadForm quad_form(const cm_model_desc& m) {
    QuadForm q;
    if constexpr (YK == CM_YIELD_J2) {
        q.a00 = q.a33 = q.a55 = 1.0; q.a03 = q.a05 = q.a35 = -0.5; q.a11 = q.a22 = q.a44 = 3.0;
    } else {
        const double F = m.yc[0], G = m.yc[1], H = m.yc[2], L = m.yc[3], M = m.yc[4], N = m.yc[5];
        q.a00 = G + H; q.a33 = F + H; q.a55 = F + G; q.a03 = -H; q.a05 = -G; q.a35 = -F;
        q.a11 = 2.0 * N; q.a22 = 2.0 * M; q.a44 = 2.0 * L;
    }
    return q;
}

// ---- symmetric input-convex network [6, H, 1] (one hidden layer) --------------------------------------
// cmad/neural_networks/input_convex_neural_network.py:36-69.  Packed weights (device memory, uniform
// addresses -> scalar loads): W0[6][H], b0[H], Wx1[6], b1, Wz[H], in_scale[6], in_min[6], out_scale, out_min, f(0).
// f(x) = softplus(x W0 + b0) . Wz + x . Wx1 + b1.  The yield term only needs the symmetrised network
// 1/2 (f(x) + f(-x)), so both signs share one pass over the hidden units (one dot product, one rank-1 Hessian
// update per unit; the pass-through x . Wx1 cancels):
//   F = f(x) + f(-x),  G = grad f(x) - grad f(-x) = d F / d x,  Hx = d2 F / d x2      (w.r.t. the SCALED input xs)
// log(1 + t) for t in [0, 1] in ~28 VALU instructions (the math library's log1p takes ~65): with v = 1 + t, halved when
// above sqrt(2), log v = 2 atanh(s), s = (v - 1) / (v + 1) within +-0.1716, by the odd series up to s^21 / 21 (truncation
// 6e-19 relative); numerator and denominator are formed from t directly, so small t keeps full relative accuracy.
// tests/test_host_math.py::test_softplus_pieces checks it against libm (<= 2 ulp).
CM_D double log1p_01(double t) {
    const bool hi = t > 0.41421356237309503;
    const double num = hi ? t - 1.0 : t;
    const double den = hi ? t + 3.0 : t + 2.0;
    const double s = num * rcp(den);
    const double s2 = s * s;
    double p = CM_SCALAR(1.0 / 21.0);
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 19.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 17.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 15.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 13.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 11.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 9.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 7.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 5.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 3.0));
    const double two_s = s + s;
    const double r = __builtin_fma(two_s * s2, p, two_s);
    return hi ? r + 0.6931471805599453 : r;
}

// log(S) for S in [1/2, 2] with an ABSOLUTE error of ~1e-16 (enough where the result is scaled down and exponentiated,
// as in S^(1/a)): S or 2 S is brought into [1, 2] and handed to log1p_01.
CM_D double log_near_one(double S) {
    const bool lo = S < 1.0;
    const double t = lo ? __builtin_fma(2.0, S, -1.0) : S - 1.0;
    const double r = log1p_01(t);
    return lo ? r - 0.6931471805599453 : r;
}

// log(x) for any positive normal x, ~1 ulp, in ~30 VALU instructions: x = m 2^e with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh((m - 1) / (m + 1)) by the same odd series as log1p_01 (|s| <= 0.1716, truncation 6e-19 relative),
// log x = e ln2_hi + (e ln2_lo + log m) with the usual split of ln 2 (e ln2_hi is exact for |e| < 2^11).
// tests/test_host_math.py::test_softplus_pieces checks it against libm.
CM_D double log_pos(double x) {
#if defined(CM_HOST_BUILD)
    int e;
    double mant = std::frexp(x, &e);                            // [1/2, 1)
#else
    double mant = __builtin_amdgcn_frexp_mant(x);
    int e = __builtin_amdgcn_frexp_exp(x);
#endif
    const bool lo = mant < 0.7071067811865476;
    mant = lo ? mant + mant : mant;
    e = lo ? e - 1 : e;
    const double s = (mant - 1.0) * rcp(mant + 1.0);
    const double s2 = s * s;
    double p = CM_SCALAR(1.0 / 21.0);
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 19.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 17.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 15.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 13.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 11.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 9.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 7.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 5.0));
    p = __builtin_fma(s2, p, CM_SCALAR(1.0 / 3.0));
    const double two_s = s + s;
    const double r = __builtin_fma(two_s * s2, p, two_s);
    const double ed = (double)e;
    return __builtin_fma(ed, CM_SCALAR(0x1.62e42fee00000p-1), __builtin_fma(ed, CM_SCALAR(0x1.a39ef35793c76p-33), r));
}

struct SoftUnit { double sp, sg; };
CM_D SoftUnit soft_unit(double a) {
    const double e = exp_s(-fabs(a));                           // in (0, 1]
    const double inv = rcp(1.0 + e);
    SoftUnit u;
    u.sp = fmax(a, 0.0) + log1p_01(e);                          // jax.nn.softplus = logaddexp(a, 0)
    u.sg = (a >= 0.0) ? inv : e * inv;                          // sigmoid(a)
    return u;
}
// Both signs of one hidden unit from ONE exponential.  With t = xs . W0[:, o], b = b0[o], p = e^t, q = e^b (uniform: read from
// the table the host appends to the packed weights, include/cmad_hip.h), A = 1 + q p, Bq = p + q, Pi = A Bq, R = 1 / Pi:
//     softplus(b + t) + softplus(b - t) = log((1 + q p)(1 + q / p)) = log(Pi) - t
//     sigmoid(b + t) - sigmoid(b - t)   = q (p^2 - 1) R
//     sum of sigmoid (1 - sigmoid)      = q p (Bq^2 + A^2) R^2
// -- one exp, one log, one reciprocal per unit instead of two of each (the two-sided form of soft_unit: ~112 instead of ~189
// VALU instructions per unit with the Hessian).  Range: t is clamped to +-kIcnnTClamp for the exponential; beyond the clamp the
// sigmoids are saturated to below 1e-17 and the value grows by |t| - kIcnnTClamp exactly, which is added back.  Units with
// |b| >= kIcnnBMax (where the clamp would not saturate, or Pi could overflow: 2 (T + |b|) < 709) take the two-sided form; b is
// uniform, so that branch is too.  Absolute accuracy of every term is ~1e-16 of the unit's own magnitude (p^2 - 1 cancels
// near t = 0, where the term itself vanishes).
constexpr double kIcnnTClamp = 190.0, kIcnnBMax = 150.0;
constexpr int kIcnnRec = 10;             // doubles per hidden-unit record of the device pack (include/cmad_hip.h, nn_weights)
// offsets into the device pack for widths [6, H, 1]
CM_D int icnn_off_b1(int H) { return 7 * H + 6; }
CM_D int icnn_off_scalers(int H) { return 8 * H + 7; }          // in_scale[6], in_min[6], out_scale, out_min, f(0)
CM_D int icnn_off_records(int H) { return 8 * H + 7 + 15; }     // per unit: W0[0..5][o], b0[o], Wz[o], e^b0[o], e^b0[o] Wz[o]
template <bool HESS>
CM_D void icnn_symmetric(const double* __restrict__ w, int H, const double xs[6], double& F, double G[6], double Hx[21]) {
    const cm_uniform_ptr u = uniform_ptr(w);
    const cm_uniform_ptr rec = u + icnn_off_records(H);
    F = 2.0 * u[icnn_off_b1(H)];
#pragma unroll
    for (int i = 0; i < 6; ++i) G[i] = 0.0;
    if constexpr (HESS) {
#pragma unroll
        for (int i = 0; i < 21; ++i) Hx[i] = 0.0;
    }
    // one contiguous record per unit (two wide scalar loads); the next unit's record is requested before this unit's
    // arithmetic, so its latency is covered
    double rn[kIcnnRec];
#pragma unroll
    for (int i = 0; i < kIcnnRec; ++i) rn[i] = rec[i];
#ifndef CM_ICNN_UNROLL
#define CM_ICNN_UNROLL 1            // experiment knob: units per loop iteration; 2 and 4 measured equal (profiles/r03_icnn_unroll_ab.txt)
#endif
#pragma unroll CM_ICNN_UNROLL
    for (int o = 0; o < H; ++o) {
        double r[kIcnnRec];
#pragma unroll
        for (int i = 0; i < kIcnnRec; ++i) r[i] = rn[i];
        const cm_uniform_ptr rnext = rec + kIcnnRec * ((o + 1 < H) ? o + 1 : o);
#pragma unroll
        for (int i = 0; i < kIcnnRec; ++i) rn[i] = rnext[i];
        double t = 0.0;
        double wc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { wc[i] = r[i]; t += xs[i] * wc[i]; }
        const double b = r[6], wz = r[7];
        double c1, c2 = 0.0;
        if (fabs(b) < kIcnnBMax) {                                  // uniform
            const double q = r[8], qwz = r[9];
            const double tc = fmax(fmin(t, kIcnnTClamp), -kIcnnTClamp);
            const double p = exp_s<false>(tc);
            const double A = __builtin_fma(q, p, 1.0), Bq = p + q, Pi = A * Bq;
            const double R = rcp(Pi);
            const double v = (log_pos(Pi) - tc) + fmax(fabs(t) - kIcnnTClamp, 0.0);
            F = __builtin_fma(v, wz, F);
            c1 = qwz * __builtin_fma(p, p, -1.0) * R;
            if constexpr (HESS) c2 = (qwz * p) * (R * R) * __builtin_fma(A, A, Bq * Bq);
        } else {
            const SoftUnit up = soft_unit(b + t), un = soft_unit(b - t);
            F += (up.sp + un.sp) * wz;
            c1 = (up.sg - un.sg) * wz;
            if constexpr (HESS) c2 = (up.sg * (1.0 - up.sg) + un.sg * (1.0 - un.sg)) * wz;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) G[i] += c1 * wc[i];
        if constexpr (HESS) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double ci = c2 * wc[i];
#pragma unroll
                for (int j = i; j < 6; ++j) Hx[sym6(i, j)] += ci * wc[j];
            }
        }
    }
}

// ---- input-convex network with any number of hidden layers [6, H1, ..., Hn, 1] -----------------------------------------------
// cmad/neural_networks/input_convex_neural_network.py:58-69:  z_1 = softplus(x W0 + b_1),
// z_{k+1} = softplus(z_k Wz_k + x Wx_{k+1} + b_{k+1}),  f = z_n wz_n + x wx_last + b_last.  Same contract as icnn_symmetric
// (F = f(x) + f(-x) and its first two derivatives w.r.t. the scaled input), by one forward pass that keeps every unit's
// softplus / sigmoid and the gradient of its pre-activation, one backward pass for nu_u = df/dz_u, and
//     grad f = wx_last + sum_{u in layer n} wz_n[u] sigmoid(a_u) grad a_u ,
//     hess f = sum_{all units u} nu_u sigmoid'(a_u) (grad a_u)(grad a_u)^T .
// The per-unit arrays are indexed at run time (private memory): this is the general path, compiled only where CM_HNN = 1;
// one hidden layer always takes icnn_symmetric.  Packed layout (the oracle's): x-layer k = 1 .. n+1: W[6][H_k], b[H_k]
// (H_{n+1} = 1); z-layer k = 1 .. n: W[H_k][H_{k+1}]; scalers; f(0).
constexpr int kIcnnMaxUnits = 64;        // hidden units of a multi-layer network, all layers together
constexpr int kIcnnMaxLayers = 6;        // entries of nn_widths in use: 6, H1 .. Hn, 1 with n <= 4
CM_D int icnn_deep_scalers_offset(const cm_model_desc& m) {
    int off = 0;
    for (int k = 1; k < m.nn_nlayers; ++k) off += 6 * m.nn_widths[k] + m.nn_widths[k];
    for (int k = 1; k + 1 < m.nn_nlayers; ++k) off += m.nn_widths[k] * m.nn_widths[k + 1];
    return off;
}
template <bool HESS>
CM_D void icnn_symmetric_deep(const cm_model_desc& m, const double xs[6], double& F, double G[6], double Hx[21]) {
    const cm_uniform_ptr w = uniform_ptr(m.nn_weights);
    const int nl = m.nn_nlayers, nh = nl - 2;                  // hidden layers 1 .. nh
    int xoff[kIcnnMaxLayers], zoff[kIcnnMaxLayers], uoff[kIcnnMaxLayers];   // x-layer k, z-layer k, first unit of layer k
    {
        int off = 0, u = 0;
        for (int k = 1; k <= nh + 1; ++k) { xoff[k] = off; off += 7 * m.nn_widths[k]; }
        for (int k = 1; k <= nh; ++k) { zoff[k] = off; off += m.nn_widths[k] * m.nn_widths[k + 1]; }
        for (int k = 1; k <= nh; ++k) { uoff[k] = u; u += m.nn_widths[k]; }
    }
    F = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) G[i] = 0.0;
    if constexpr (HESS) {
#pragma unroll
        for (int i = 0; i < 21; ++i) Hx[i] = 0.0;
    }
    for (int sgn = 0; sgn < 2; ++sgn) {
        const double sx = sgn ? -1.0 : 1.0;
        double x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = sx * xs[i];
        double spv[kIcnnMaxUnits], sgv[kIcnnMaxUnits], nu[kIcnnMaxUnits], Ja[kIcnnMaxUnits][6];
        for (int k = 1; k <= nh; ++k) {                          // forward
            const int Hk = m.nn_widths[k], Hp = (k > 1) ? m.nn_widths[k - 1] : 0;
            for (int v = 0; v < Hk; ++v) {
                double a = w[xoff[k] + 6 * Hk + v], J[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) { J[i] = w[xoff[k] + i * Hk + v]; a += x[i] * J[i]; }
                for (int u = 0; u < Hp; ++u) {
                    const double wz = w[zoff[k - 1] + u * Hk + v], c = wz * sgv[uoff[k - 1] + u];
                    a += wz * spv[uoff[k - 1] + u];
#pragma unroll
                    for (int i = 0; i < 6; ++i) J[i] += c * Ja[uoff[k - 1] + u][i];
                }
                const SoftUnit su = soft_unit(a);
                spv[uoff[k] + v] = su.sp; sgv[uoff[k] + v] = su.sg;
#pragma unroll
                for (int i = 0; i < 6; ++i) Ja[uoff[k] + v][i] = J[i];
            }
        }
        // output layer: value and gradient (the x . wx_last pass-through cancels between the two signs; kept for clarity)
        const int Hn = m.nn_widths[nh];
        double f = w[xoff[nh + 1] + 6], g[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { g[i] = w[xoff[nh + 1] + i]; f += x[i] * g[i]; }
        for (int u = 0; u < Hn; ++u) {
            const double wz = w[zoff[nh] + u];
            f += wz * spv[uoff[nh] + u];
            const double c = wz * sgv[uoff[nh] + u];
#pragma unroll
            for (int i = 0; i < 6; ++i) g[i] += c * Ja[uoff[nh] + u][i];
            nu[uoff[nh] + u] = wz;
        }
        F += f;
#pragma unroll
        for (int i = 0; i < 6; ++i) G[i] += sx * g[i];
        if constexpr (HESS) {
            for (int k = nh - 1; k >= 1; --k) {                  // backward: nu_u = df/dz_u
                const int Hk = m.nn_widths[k], Hq = m.nn_widths[k + 1];
                for (int u = 0; u < Hk; ++u) {
                    double t = 0.0;
                    for (int v = 0; v < Hq; ++v) t += w[zoff[k] + u * Hq + v] * nu[uoff[k + 1] + v] * sgv[uoff[k + 1] + v];
                    nu[uoff[k] + u] = t;
                }
            }
            const int ntot = uoff[nh] + Hn;
            for (int u = 0; u < ntot; ++u) {
                const double c2 = nu[u] * sgv[u] * (1.0 - sgv[u]);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double ci = c2 * Ja[u][i];
#pragma unroll
                    for (int j = i; j < 6; ++j) Hx[sym6(i, j)] += ci * Ja[u][j];
                }
            }
        }
    }
}

// NN(flat dev s) of hybrid_hill_effective_stress (cmad/models/effective_stress.py:149-163) in the 6-vector
// basis: value, d/ds6, d2/ds6 ds6.  NN input order is [xx,yy,zz,xy,xz,yz] of the deviator.
// value, d/ds6 and (HESS) the second derivative written to the packed Hessian Hp
template <bool HESS>
CM_D void icnn_yield_term(const cm_model_desc& m, const double s[6], double& val, double g6[6], double Hp[21]) {
    const double* __restrict__ w = m.nn_weights;
    const int H = m.nn_widths[1];
    const bool deep = (CM_HNN != 0) && m.nn_nlayers > 3;        // uniform; more than one hidden layer: the general evaluation
    const cm_uniform_ptr sc = uniform_ptr(w) + (deep ? icnn_deep_scalers_offset(m) : icnn_off_scalers(H));   // in_scale[6], in_min[6], out_scale, out_min, f0
    const double h = (s[0] + s[3] + s[5]) * (1.0 / 3.0);
    const double x[6] = {s[0] - h, s[3] - h, s[5] - h, s[1], s[2], s[4]};
    // scaled input; the reference evaluates g(xs) and g(-xs) (input_convex_neural_network.py:59-69)
    double xs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xs[i] = sc[i] * x[i] + sc[6 + i];
    double F, G[6], Hx[HESS ? 21 : 1];
    if constexpr (CM_HNN != 0) {
        if (deep) icnn_symmetric_deep<HESS>(m, xs, F, G, Hx);
        else icnn_symmetric<HESS>(w, H, xs, F, G, Hx);
    } else icnn_symmetric<HESS>(w, H, xs, F, G, Hx);
    const double ios = 1.0 / sc[12];
    val = (0.5 * F - sc[14] - sc[13]) * ios;                   // (1/2 (f(x)+f(-x)) - f(0) - out_min) / out_scale
    double gx[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) gx[i] = 0.5 * G[i] * sc[i] * ios;
    // chain through x(s6): normal slots k -> x index (0,1,2) minus the mean; shear slots 1,2,4 -> x index 3,4,5
    constexpr int XI[6] = {0, 3, 4, 1, 5, 2};
    const double gm = (gx[0] + gx[1] + gx[2]) * (1.0 / 3.0);
#pragma unroll
    for (int k = 0; k < 6; ++k) g6[k] = gx[XI[k]] - (kDiag[k] ? gm : 0.0);
    if constexpr (HESS) {
        double rm[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) Hx[sym6(i, j)] *= 0.5 * sc[i] * sc[j] * ios;
        // J = d x / d s6: H6 = J^T Hx J with J_ik = delta(i, XI[k]) - (i < 3 && diag k) / 3
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) rm[i] = (Hx[sym6(i, 0)] + Hx[sym6(i, 1)] + Hx[sym6(i, 2)]) * (1.0 / 3.0);   // row means over normal cols
#pragma unroll
        for (int i = 0; i < 3; ++i) tot += rm[i];
        tot *= (1.0 / 3.0);
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int l = k; l < 6; ++l) {
                double v = Hx[sym6(XI[k], XI[l])];
                if (kDiag[l]) v -= rm[XI[k]];
                if (kDiag[k]) v -= rm[XI[l]];
                if (kDiag[k] && kDiag[l]) v += tot;
                Hp[sym6(k, l)] = v;
            }
    }
}

CM_D double quad_min(double phi0, double dphi0, double a, double phi);

// the network-backed surfaces (dense Hessian, no structured solve, no second-derivative kernel)
constexpr bool is_nn_yield(int yk) { return yk == CM_YIELD_HYBRID_HILL_NN || yk == CM_YIELD_SCALED_HYBRID_HILL_NN; }
// surfaces whose Hessian is a dense 6x6 (dense LU path only)
constexpr bool is_dense_yield(int yk) { return is_nn_yield(yk) || yk == CM_YIELD_BARLAT; }

// ---- Barlat Yld2004-18p (cmad/verification/functions.py:71-154, cmad/models/effective_stress.py:55-84) ----------
// S' = L' s, S'' = L'' s (two linear maps of the stress), phi = (1/4 sum_ij |l'_i - l''_j|^a)^(1/a) over the
// eigenvalues of S' and S''.  The reference differentiates through jnp.linalg.eigh; here the value, gradient and
// Hessian come from the spectral representation of an isotropic function of two symmetric tensors:
//   d l_i = v_i^T dS v_i ,   d2 l_i = 2 sum_{j != i} (v_i^T dS v_j)^2 / (l_i - l_j)
//   with r_ij = |l'_i - l''_j| / phi, q_ij = sign r_ij^(a-1), D_ij = l'_i - l''_j:
//   d phi = 1/4 sum q_ij dD_ij ,   d2 phi = (a-1)/phi [ 1/4 sum r_ij^(a-2) dD_ij^2 - d phi^2 ] + sum_x phi_x d2 l_x
// Equal eigenvalues: (phi_i - phi_j)/(l_i - l_j) -> phi_ii - phi_ij (phi is symmetric in each triple), so nothing
// is singular there -- unlike the eigh derivative rule, which divides by zero.
// cyclic Jacobi, symmetric 3x3 given as [xx,xy,xz,yy,yz,zz]; V columns = eigenvectors
CM_D void eig_sym3(const double s[6], double lam[3], double V[3][3]) {
    double a00 = s[0], a01 = s[1], a02 = s[2], a11 = s[3], a12 = s[4], a22 = s[5];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    // one rotation in the (p, q) plane; app, aqq, apq the 2x2 block, apr / aqr the couplings to the third index
    // reciprocals and square roots by rcp / rsqrt_pos (1 ulp, ~5 instructions) instead of the IEEE division and sqrt sequences: a
    // rotation drops from ~90 to ~55 instructions, and the eigen-decompositions are a third of a Barlat evaluation.  theta is
    // clamped so that theta^2 stays finite (beyond 1e150 the rotation angle is below 1e-150 anyway; a NaN from a denormal
    // off-diagonal entry clamps to the same no-op rotation).
#define CM_JACOBI(app, aqq, apq, apr, aqr, P, Q)                                                   \
    if (apq != 0.0) {                                                                              \
        const double theta = fmin(fmax((aqq - app) * (0.5 * rcp(apq)), -1e150), 1e150);            \
        const double h2 = theta * theta + 1.0;                                                     \
        const double t = ((theta >= 0.0) ? 1.0 : -1.0) * rcp(fabs(theta) + h2 * rsqrt_pos(h2));    \
        const double c = rsqrt_pos(t * t + 1.0), sn = t * c;                                       \
        app -= t * apq; aqq += t * apq; apq = 0.0;                                                 \
        const double xr = apr, yr = aqr;                                                           \
        apr = c * xr - sn * yr; aqr = sn * xr + c * yr;                                            \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                            \
            const double vp = V[k][P], vq = V[k][Q];                                               \
            V[k][P] = c * vp - sn * vq; V[k][Q] = sn * vp + c * vq;                                \
        }                                                                                          \
    }
    for (int sweep = 0; sweep < 6; ++sweep) {              // quadratic convergence: 4-5 sweeps reach round-off
        CM_JACOBI(a00, a11, a01, a02, a12, 0, 1)
        CM_JACOBI(a00, a22, a02, a01, a12, 0, 2)
        CM_JACOBI(a11, a22, a12, a01, a02, 1, 2)
        // done when what is left off the diagonal cannot move an eigenvalue or a vector by 1e-20 relative (wave-uniform exit)
        const double off = fabs(a01) + fabs(a02) + fabs(a12), dia = fabs(a00) + fabs(a11) + fabs(a22);
        if (!__any(off > 1e-20 * dia)) break;
    }
#undef CM_JACOBI
    lam[0] = a00; lam[1] = a11; lam[2] = a22;
}

// a_x = L^T (w o V(v v^T)) resp. L^T (w o V(sym(v_i v_j^T))): gradient of v_i^T S v_j w.r.t. the stress 6-vector
CM_D void barlat_pull(const double UL[3][3], double c44, double c55, double c66,
                      const double vi[3], const double vj[3], double out[6]) {
    const double mxx = vi[0] * vj[0], myy = vi[1] * vj[1], mzz = vi[2] * vj[2];
    out[0] = UL[0][0] * mxx + UL[1][0] * myy + UL[2][0] * mzz;
    out[3] = UL[0][1] * mxx + UL[1][1] * myy + UL[2][1] * mzz;
    out[5] = UL[0][2] * mxx + UL[1][2] * myy + UL[2][2] * mzz;
    out[1] = c44 * (vi[0] * vj[1] + vi[1] * vj[0]);
    out[2] = c66 * (vi[0] * vj[2] + vi[2] * vj[0]);
    out[4] = c55 * (vi[1] * vj[2] + vi[2] * vj[1]);
}

template <bool HESS>
CM_D void barlat_eval(const cm_model_desc& m, const double s[6], double& phi, double gt[6], double Hp[21]) {
    const double a = m.yc[18];
    double lam[2][3], V[2][3][3], UL[2][3][3], csh[2][3];
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        const double* q = m.yc + 9 * set;                      // c12, c13, c21, c23, c31, c32, c44, c55, c66
        const double t3 = 1.0 / 3.0;
        UL[set][0][0] = (q[0] + q[1]) * t3; UL[set][0][1] = (-2.0 * q[0] + q[1]) * t3; UL[set][0][2] = (q[0] - 2.0 * q[1]) * t3;
        UL[set][1][0] = (-2.0 * q[2] + q[3]) * t3; UL[set][1][1] = (q[2] + q[3]) * t3; UL[set][1][2] = (q[2] - 2.0 * q[3]) * t3;
        UL[set][2][0] = (-2.0 * q[4] + q[5]) * t3; UL[set][2][1] = (q[4] - 2.0 * q[5]) * t3; UL[set][2][2] = (q[4] + q[5]) * t3;
        csh[set][0] = q[6]; csh[set][1] = q[7]; csh[set][2] = q[8];
        double S6[6];
        S6[0] = UL[set][0][0] * s[0] + UL[set][0][1] * s[3] + UL[set][0][2] * s[5];
        S6[3] = UL[set][1][0] * s[0] + UL[set][1][1] * s[3] + UL[set][1][2] * s[5];
        S6[5] = UL[set][2][0] * s[0] + UL[set][2][1] * s[3] + UL[set][2][2] * s[5];
        S6[1] = q[6] * s[1]; S6[4] = q[7] * s[4]; S6[2] = q[8] * s[2];     // xy: c44, yz: c55, zx: c66
        eig_sym3(S6, lam[set], V[set]);
    }
    double Dm[3][3], mx = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { Dm[i][j] = lam[0][i] - lam[1][j]; mx = fmax(mx, fabs(Dm[i][j])); }
    const double imx = (mx > 0.0) ? rcp(mx) : 0.0;
    double u[3][3], ua[3][3], uam2[3][3], Ssum = 0.0;
    const int ai = (int)a;
    const bool int_a = (a == (double)ai) && ai >= 2 && ai <= 65536;
    // Integer exponents (the usual case): u^(a-2) by repeated squaring, u^a = u^(a-2) u^2 -- the (a-2)-th power is what the
    // gradient and the Hessian need, so no division by u^2 afterwards; reciprocals by rcp, S^(1/a) through log_pos / exp_s
    // (max u = 1 puts S in [1/4, 9/4]) instead of the IEEE division and libm sequences (~390 instructions of an evaluation).
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            u[i][j] = fabs(Dm[i][j]) * imx;
            double p, p2;
            if (int_a) {
                p2 = 1.0;
                double base = u[i][j];
                for (int e = ai - 2; e != 0; e >>= 1) { if (e & 1) p2 *= base; base *= base; }
                p = p2 * (u[i][j] * u[i][j]);
            } else {
                p = (u[i][j] > 0.0) ? exp(a * log(u[i][j])) : 0.0;
                p2 = (u[i][j] > 0.0) ? p / (u[i][j] * u[i][j]) : ((a == 2.0) ? 1.0 : 0.0);
            }
            ua[i][j] = p; uam2[i][j] = p2; Ssum += p;
        }
    Ssum *= 0.25;
    double Sr, iSr, c2;
    if (int_a) {
        Sr = (Ssum > 0.0) ? exp_s(log_pos(Ssum) * rcp(a)) : 0.0;
        iSr = (Ssum > 0.0) ? rcp(Sr) : 0.0;
        c2 = (Ssum > 0.0) ? Sr * Sr * rcp(Ssum) : 0.0;
    } else {
        Sr = (Ssum > 0.0) ? exp(log(Ssum) / a) : 0.0;
        iSr = (Ssum > 0.0) ? 1.0 / Sr : 0.0;
        c2 = (Ssum > 0.0) ? Sr * Sr / Ssum : 0.0;
    }
    phi = mx * Sr;
    // q_ij = sign r^(a-1), e_ij = r^(a-2) with r = |D| / phi = u / Sr
    double qm[3][3], em[3][3], f1[3] = {0.0, 0.0, 0.0}, f2[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double uu = u[i][j];
            em[i][j] = uam2[i][j] * c2;                        // (u = 0: 0^(a-2) from the power loop / the a = 2 case above)
            const double r = uu * iSr;
            qm[i][j] = ((Dm[i][j] > 0.0) ? 1.0 : ((Dm[i][j] < 0.0) ? -1.0 : 0.0)) * em[i][j] * r;
            f1[i] += 0.25 * qm[i][j]; f2[j] -= 0.25 * qm[i][j];       // d phi / d l'_i , d phi / d l''_j
        }
    double ax[2][3][6];                                        // gradients of the eigenvalues w.r.t. the stress
#pragma unroll
    for (int set = 0; set < 2; ++set)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double vi[3] = {V[set][0][i], V[set][1][i], V[set][2][i]};
            barlat_pull(UL[set], csh[set][0], csh[set][1], csh[set][2], vi, vi, ax[set][i]);
        }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double g = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) g += f1[i] * ax[0][i][k] + f2[i] * ax[1][i][k];
        gt[k] = g;
    }
    if constexpr (HESS) {
        const double ip = imx * iSr, am1 = (a - 1.0) * ip;          // 1 / phi (both factors vanish with phi)
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int l = k; l < 6; ++l) Hp[sym6(k, l)] = -am1 * gt[k] * gt[l];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double cf = 0.25 * am1 * em[i][j];
                double dv[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) dv[k] = ax[0][i][k] - ax[1][j][k];
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int l = k; l < 6; ++l) Hp[sym6(k, l)] += cf * dv[k] * dv[l];
            }
        // eigenvector rotation terms: 2 theta_ij b_ij b_ij^T, theta_ij = (phi_i - phi_j) / (l_i - l_j)
#pragma unroll
        for (int set = 0; set < 2; ++set) {
            const double* fx = set ? f2 : f1;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i + 1; j < 3; ++j) {
                    const double dl = lam[set][i] - lam[set][j];
                    const double scale = fabs(lam[set][i]) + fabs(lam[set][j]) + mx;
                    double theta;
                    if (fabs(dl) > 1e-7 * scale) theta = (fx[i] - fx[j]) * rcp(dl);
                    else {                                     // limit: phi_ii - phi_ij in eigenvalue space
                        double hii = 0.0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) hii += 0.25 * am1 * (set ? em[k][i] : em[i][k]);
                        theta = hii - am1 * fx[i] * fx[i] + am1 * fx[i] * fx[j];
                    }
                    const double vi[3] = {V[set][0][i], V[set][1][i], V[set][2][i]};
                    const double vj[3] = {V[set][0][j], V[set][1][j], V[set][2][j]};
                    double b[6];
                    barlat_pull(UL[set], csh[set][0], csh[set][1], csh[set][2], vi, vj, b);
#pragma unroll
                    for (int k = 0; k < 6; ++k)
#pragma unroll
                        for (int l = k; l < 6; ++l) Hp[sym6(k, l)] += 2.0 * theta * b[k] * b[l];
                }
        }
    }
}

template <int YK, bool HESS>
CM_D void yield_eval_p(const cm_model_desc& m, const double s[6], double& phi, double gt[6], double Hp[21]);

// scaled_effective_stress around the hybrid surface (cmad/models/effective_stress.py:97-108, 130-146):
//   phi(s) = phi_h(beta s) / beta,  beta: phi_h(beta s) = Yeq  (scalar make_newton_solve started at Y / phi_J2(s),
//   default line search), phi = phi_J2 when phi_J2 is within 1e-14 of zero.
// With tau = beta s, g = grad phi_h(tau), H = hess phi_h(tau), c = g . tau, and the implicit-function derivative
// d beta = -beta (g . ds) / (g . s) of nonlinear_solver.py:158-171:
//   grad phi  = phi_h g / c
//   hess phi  = beta M (I - tau g^T / c),   M = g g^T / c + phi_h H / c - phi_h g (H tau + g)^T / c^2
template <bool HESS>
CM_D void scaled_hybrid_eval(const cm_model_desc& m, const double s[6], double& phi, double gt[6], double Hp[21]) {
    double pj;
    {
        double gj[6];
        yield_eval_p<CM_YIELD_J2, false>(m, s, pj, gj, Hp);
    }
    if (!(fabs(pj) > 1e-14)) {                                  // jnp.isclose(phi_J2, 0., tol, tol): J2 value, normal := 0
        phi = pj;
#pragma unroll
        for (int k = 0; k < 6; ++k) gt[k] = 0.0;
        if constexpr (HESS) {
#pragma unroll
            for (int k = 0; k < 21; ++k) Hp[k] = 0.0;
        }
        return;
    }
    const double iy = 1.0 / m.beta_equivalent_stress;
    // r(beta) = phi_h(beta s) / Yeq - 1 and r'(beta) = grad phi_h(beta s) . s / Yeq
    auto r_dr = [&](double b, double& r, double& dr) {
        double t[6], ph, g[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = b * s[k];
        yield_eval_p<CM_YIELD_HYBRID_HILL_NN, false>(m, t, ph, g, Hp);
        double gs = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) gs += g[k] * s[k];
        r = ph * iy - 1.0; dr = gs * iy;
    };
    double beta = m.Y / pj;
    {
        constexpr int kMaxEvals = 4;                            // DEFAULT_LINE_SEARCH_SETTINGS, line_search.py:40-46
        double C, dC;
        r_dr(beta, C, dC);
        const double n0 = fabs(C);
        for (int it = 0; it < m.beta_max_iters; ++it) {
            const double nrm = fabs(C);
            if (nrm / n0 < m.beta_rel_tol || nrm < m.beta_abs_tol) break;
            const double delta = C / dC;
            const double phi0 = 0.5 * C * C, dphi0 = -C * C, armijo = m.ls_c1 * dphi0;
            int n = 0;
            double alpha = 1.0, best_alpha = 1.0, best_phi = INFINITY, best_C = C, best_dC = dC, Ct = C, dCt = dC;
            bool accepted = false, have_best = false;
            while (n < kMaxEvals && !accepted) {
                r_dr(beta - alpha * delta, Ct, dCt);
                const double ph = 0.5 * Ct * Ct;
                const bool finite = isfinite(ph);
                if (finite && ph < best_phi) { best_alpha = alpha; best_phi = ph; best_C = Ct; best_dC = dCt; have_best = true; }
                accepted = finite && (ph <= phi0 + alpha * armijo);
                const double am = quad_min(phi0, dphi0, alpha, ph);
                const double ac = fmin(fmax(am, m.ls_lo * alpha), m.ls_hi * alpha);
                if (!accepted) alpha = finite ? ac : 0.5 * alpha;
                ++n;
            }
            if (accepted) { beta -= alpha * delta; C = Ct; dC = dCt; }
            else if (have_best) { beta -= best_alpha * delta; C = best_C; dC = best_dC; }
            else {                                              // every trial non-finite: full step, base residual carried
                beta -= delta;
                double Cn;
                r_dr(beta, Cn, dC);
            }
        }
    }
    double tau[6], ph, g[6], H[HESS ? 21 : 1];
#pragma unroll
    for (int k = 0; k < 6; ++k) tau[k] = beta * s[k];
    yield_eval_p<CM_YIELD_HYBRID_HILL_NN, HESS>(m, tau, ph, g, H);
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) c += g[k] * tau[k];
    const double ic = 1.0 / c;
    phi = ph / beta;
#pragma unroll
    for (int k = 0; k < 6; ++k) gt[k] = ph * g[k] * ic;
    if constexpr (HESS) {
        double Ht_au[6], tHt = 0.0;                              // H tau, tau . H tau
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < 6; ++l) a += H[sym6(k, l)] * tau[l];
            Ht_au[k] = a; tHt += tau[k] * a;
        }
        // M tau = g + phi_h H tau / c - phi_h g (tau . H tau + c) / c^2
        double Mt[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) Mt[k] = g[k] + ph * ic * Ht_au[k] - ph * g[k] * (tHt + c) * ic * ic;
        // the Hessian is symmetric: the upper triangle of beta (M - M tau g^T / c) is all of it
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int l = k; l < 6; ++l) {
                const double Mkl = g[k] * g[l] * ic + ph * ic * H[sym6(k, l)] - ph * g[k] * (Ht_au[l] + g[l]) * ic * ic;
                Hp[sym6(k, l)] = beta * (Mkl - Mt[k] * g[l] * ic);
            }
    }
}

template <int YK, bool HESS>
CM_D void yield_eval_p(const cm_model_desc& m, const double s[6], double& phi, double gt[6], double Hp[21]) {
    if constexpr (YK == CM_YIELD_BARLAT) {
        barlat_eval<HESS>(m, s, phi, gt, Hp);
    } else if constexpr (YK == CM_YIELD_SCALED_HYBRID_HILL_NN) {
        scaled_hybrid_eval<HESS>(m, s, phi, gt, Hp);
    } else if constexpr (YK == CM_YIELD_HYBRID_HILL_NN) {
        // the network term first (its loop over the hidden units carries 28 accumulators), the Hill quadratic form added after it,
        // so that the Hill Hessian is not live across that loop
        double v;
        icnn_yield_term<HESS>(m, s, v, gt, Hp);
        const QuadForm q = quad_form<CM_YIELD_HILL>(m);
        double As[6];
        As[0] = q.a00 * s[0] + q.a03 * s[3] + q.a05 * s[5];
        As[3] = q.a03 * s[0] + q.a33 * s[3] + q.a35 * s[5];
        As[5] = q.a05 * s[0] + q.a35 * s[3] + q.a55 * s[5];
        As[1] = q.a11 * s[1]; As[2] = q.a22 * s[2]; As[4] = q.a44 * s[4];
        double qq = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) qq += s[k] * As[k];
        const double ph = sqrt(qq), ip = (qq > 0.0) ? rcp(ph) : 0.0;
        phi = ph + v;
#pragma unroll
        for (int k = 0; k < 6; ++k) As[k] *= ip;               // Hill gradient
        if constexpr (HESS) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int l = k; l < 6; ++l) Hp[sym6(k, l)] -= As[k] * As[l] * ip;
            Hp[sym6(0, 0)] += q.a00 * ip; Hp[sym6(3, 3)] += q.a33 * ip; Hp[sym6(5, 5)] += q.a55 * ip;
            Hp[sym6(0, 3)] += q.a03 * ip; Hp[sym6(0, 5)] += q.a05 * ip; Hp[sym6(3, 5)] += q.a35 * ip;
            Hp[sym6(1, 1)] += q.a11 * ip; Hp[sym6(2, 2)] += q.a22 * ip; Hp[sym6(4, 4)] += q.a44 * ip;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) gt[k] += As[k];
    } else if constexpr (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL) {
        const QuadForm q = quad_form<YK>(m);
        double As[6];
        As[0] = q.a00 * s[0] + q.a03 * s[3] + q.a05 * s[5];
        As[3] = q.a03 * s[0] + q.a33 * s[3] + q.a35 * s[5];
        As[5] = q.a05 * s[0] + q.a35 * s[3] + q.a55 * s[5];
        As[1] = q.a11 * s[1]; As[2] = q.a22 * s[2]; As[4] = q.a44 * s[4];
        double qq = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) qq += s[k] * As[k];
        phi = sqrt(qq);
        // zero stress: the reference's normal is NaN there and hidden by the branch select (phi = 0 is always
        // elastic); it is defined as 0 here so that nothing downstream has to be guarded
        const double ip = (qq > 0.0) ? rcp(phi) : 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) gt[k] = As[k] * ip;
        if constexpr (HESS) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int l = k; l < 6; ++l) Hp[sym6(k, l)] = -gt[k] * gt[l] * ip;
            Hp[sym6(0, 0)] += q.a00 * ip; Hp[sym6(3, 3)] += q.a33 * ip; Hp[sym6(5, 5)] += q.a55 * ip;
            Hp[sym6(0, 3)] += q.a03 * ip; Hp[sym6(0, 5)] += q.a05 * ip; Hp[sym6(3, 5)] += q.a35 * ip;
            Hp[sym6(1, 1)] += q.a11 * ip; Hp[sym6(2, 2)] += q.a22 * ip; Hp[sym6(4, 4)] += q.a44 * ip;
        }
    } else if constexpr (YK == CM_YIELD_HOSFORD) {
        // cmad/models/effective_stress.py:167-177: phi = vm (1/2 sum |d_i/vm|^a)^(1/a) with
        // d = (s00-s11, s11-s22, s22-s00); the vm scaling cancels analytically, so
        // phi = (1/2 sum |d_i|^a)^(1/a).  Scaled here by max|d_i| instead (same purpose: no overflow).
        const double a = m.yc[0];
        const double dd[3] = {s[0] - s[3], s[3] - s[5], s[5] - s[0]};
        const double t0 = fabs(dd[0]), t1 = fabs(dd[1]), t2 = fabs(dd[2]);
        const double mx = fmax(t0, fmax(t1, t2));
        const double imx = (mx > 0.0) ? rcp(mx) : 0.0;              // equal normal stresses: phi = 0, normal := 0
        const double u[3] = {t0 * imx, t1 * imx, t2 * imx};
        // u_i^a (u_i in [0,1]); (|d_i|/phi)^(a-2) = u_i^a Sr^2 / (u_i^2 S) -- no further pow.
        // Integer exponents (the usual case: 6, 8, 100) by repeated squaring, ~log2(a) multiplications per term
        // and a few ulp; anything else as exp(a log u).  `a` is a kernel argument, so the branch is uniform.
        double ua[3], uam2[3];                                      // u_i^a and u_i^(a-2)
        const int ai = (int)a;
        const bool int_pow = (a == (double)ai && ai >= 2 && ai <= 65536);
        if (int_pow) {
            // u^(a-2) by repeated squaring, then u^a = u^(a-2) u^2: the (a-2)-th power is what the gradient and the Hessian
            // need (no reciprocal of u_i afterwards)
            double base[3] = {u[0], u[1], u[2]};
            uam2[0] = uam2[1] = uam2[2] = 1.0;
            for (int e = ai - 2; e != 0; e >>= 1) {
                if (e & 1) { uam2[0] *= base[0]; uam2[1] *= base[1]; uam2[2] *= base[2]; }
                base[0] *= base[0]; base[1] *= base[1]; base[2] *= base[2];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) ua[i] = uam2[i] * (u[i] * u[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ua[i] = (u[i] > 0.0) ? exp(a * log(u[i])) : 0.0;
                uam2[i] = (u[i] > 0.0) ? ua[i] / (u[i] * u[i]) : ((a == 2.0) ? 1.0 : 0.0);
            }
        }
        const double S = 0.5 * (ua[0] + ua[1] + ua[2]);
        // S^(1/a): max u_i = 1 puts S in [1/2, 3/2] whenever it is non-zero, so the short logarithm applies; reciprocals by
        // rcp (1 ulp) instead of the IEEE division sequence on the integer-exponent path
        double Sr, iSr, iS;
        if (int_pow) {
            const double xl = log_near_one(S) * rcp(a);             // |xl| <= 0.7 / a
            if (a >= 64.0) {
                // |xl| < 2^-6.5: e^x by its Taylor polynomial of degree 7 (truncation below 1e-19 relative) -- no range reduction
                double pe = CM_SCALAR(1.0 / 5040.0);
                pe = __builtin_fma(xl, pe, CM_SCALAR(1.0 / 720.0));
                pe = __builtin_fma(xl, pe, CM_SCALAR(1.0 / 120.0));
                pe = __builtin_fma(xl, pe, CM_SCALAR(1.0 / 24.0));
                pe = __builtin_fma(xl, pe, CM_SCALAR(1.0 / 6.0));
                pe = __builtin_fma(xl, pe, 0.5);
                pe = __builtin_fma(xl, pe, 1.0);
                Sr = (S > 0.0) ? __builtin_fma(xl, pe, 1.0) : 0.0;
            } else {
                Sr = (S > 0.0) ? exp_s(xl) : 0.0;
            }
            iSr = (S > 0.0) ? rcp(Sr) : 0.0;
            iS = (S > 0.0) ? rcp(S) : 0.0;
        } else {
            Sr = (S > 0.0) ? exp(log(S) / a) : 0.0;
            iSr = (S > 0.0) ? 1.0 / Sr : 0.0;
            iS = (S > 0.0) ? 1.0 / S : 0.0;
        }
        phi = mx * Sr;
        double p[3], r[3], sg[3], ram2[3];
        const double c2 = Sr * Sr * iS;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            r[i] = u[i] * iSr;                                  // |d_i| / phi
            sg[i] = (dd[i] > 0.0) ? 1.0 : ((dd[i] < 0.0) ? -1.0 : 0.0);
            ram2[i] = uam2[i] * c2;                             // (|d_i| / phi)^(a-2) = u_i^(a-2) Sr^2 / S   (a = 2: u^0 = 1)
            p[i] = 0.5 * ram2[i] * r[i] * sg[i];               // d phi / d d_i
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) gt[k] = 0.0;
        gt[0] = p[0] - p[2]; gt[3] = p[1] - p[0]; gt[5] = p[2] - p[1];
        if constexpr (HESS) {
#pragma unroll
            for (int k = 0; k < 21; ++k) Hp[k] = 0.0;
            double Hd[3][3];
            const double ip = imx * iSr;                            // 1 / phi (both factors vanish with phi)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Hd[i][j] = 0.5 * (a - 1.0) * ram2[i] * ((i == j ? 1.0 : 0.0) - sg[i] * r[i] * p[j]) * ip;
            // D = d d / d (s0,s3,s5): rows d_i, cols (0,3,5)
            constexpr double D[3][3] = {{1.0, -1.0, 0.0}, {0.0, 1.0, -1.0}, {-1.0, 0.0, 1.0}};
            constexpr int IDX[3] = {0, 3, 5};
#pragma unroll
            for (int A = 0; A < 3; ++A)
#pragma unroll
                for (int Bc = A; Bc < 3; ++Bc) {
                    double sH = 0.0;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) sH += D[i][A] * Hd[i][j] * D[j][Bc];
                    Hp[sym6(IDX[A], IDX[Bc])] = sH;
                }
        }
    }
}

// the same with the Hessian as a full 6x6 array (dense Jacobian assembly: UNIAXIAL_STRESS, the rate form)
template <int YK, bool HESS>
CM_D void yield_eval(const cm_model_desc& m, const double s[6], double& phi, double gt[6], double Ht[6][6]) {
    double Hp[HESS ? 21 : 1];
    yield_eval_p<YK, HESS>(m, s, phi, gt, Hp);
    if constexpr (HESS) sym6_expand(Hp, Ht);
}

// ---- hardening  cmad/models/hardening.py:9-34 --------------------------------------------------
struct Hard { double H, dH, expo; };
// The network hardening law (widths [1, H, 1], include/cmad_hip.h hnn_width): value and derivative of
//   out_scale * sum_u W2[u] (sigmoid(W1[u] in_scale alpha + b1[u]) - sigmoid(b1[u])).
// Compiled only into the HNN build of the library (CM_HNN = 1, cmad_hip.hip): every kernel evaluates the hardening laws inside
// its Newton loop, and this loop there costs the Voce / linear configurations scalar registers and spill traffic whether it
// is inlined or called (profiles/r03_hnn_ab.txt).  The host build (tests) always has it.
struct HnnTerm { double H, dH; };
CM_D HnnTerm hardening_network(const double* wv, int Hn, double alpha) {
    const cm_uniform_ptr w = uniform_ptr(wv);
    const double si = w[3 * Hn + 1], so = w[3 * Hn + 2];
    double acc = 0.0, dacc = 0.0;
    for (int u = 0; u < Hn; ++u) {
        const double w1 = w[u] * si, b1 = w[Hn + u], w2 = w[2 * Hn + u], sg0 = w[3 * Hn + 3 + u];
        const double a = __builtin_fma(w1, alpha, b1);
        const double e = exp_s(-fabs(a)), inv = rcp(1.0 + e);
        const double sg = (a >= 0.0) ? inv : e * inv;           // sigmoid(a)
        acc = __builtin_fma(w2, sg - sg0, acc);
        dacc = __builtin_fma(w2 * w1, sg * (1.0 - sg), dacc);
    }
    return HnnTerm{so * acc, so * dacc};
}
// Widths [1, H1, ..., Hn, 1] with n >= 2 (cm_model_desc.hnn_nhidden, general weight layout: include/cmad_hip.h): the reference's
// forward loops over any depth (simple_neural_network.py:19-23).  Forward pass carrying d/d alpha through the layers
//   z = W^T a + b ,  a' = sigmoid(z) ,  da'/dalpha = sigmoid'(z) (W^T da/dalpha)
// in two ping-pong arrays of kHnnMaxUnits entries (scratch-resident: correct, not fast -- an EXT-build feature like the deep
// ICNN); forward(0) is the constant the host put behind the scales.
constexpr int kHnnMaxHidden = 4, kHnnMaxUnits = 64;
CM_D int hnn_general_size(const cm_model_desc& m) {           // doubles of the layers' weights and biases (scales follow)
    int n = 0, nin = 1;
    for (int l = 0; l < m.hnn_nhidden; ++l) { n += nin * m.hnn_widths[l] + m.hnn_widths[l]; nin = m.hnn_widths[l]; }
    return n + nin + 1;
}
CM_D HnnTerm hardening_network_deep(const cm_model_desc& m, double alpha) {
    const cm_uniform_ptr w = uniform_ptr(m.nn_weights + m.hnn_offset);
    const int nh = m.hnn_nhidden, tail = hnn_general_size(m);
    const double si = w[tail], so = w[tail + 1], f0 = w[tail + 2];
    double a[2][kHnnMaxUnits], da[2][kHnnMaxUnits];
    a[0][0] = si * alpha; da[0][0] = si;
    int off = 0, nin = 1, cur = 0;
    for (int l = 0; l < nh; ++l) {
        const int nout = m.hnn_widths[l];
        for (int o = 0; o < nout; ++o) {
            double z = w[off + nin * nout + o], dz = 0.0;
            for (int i = 0; i < nin; ++i) {
                const double wio = w[off + i * nout + o];
                z = __builtin_fma(wio, a[cur][i], z);
                dz = __builtin_fma(wio, da[cur][i], dz);
            }
            const double e = exp_s(-fabs(z)), inv = rcp(1.0 + e);
            const double sg = (z >= 0.0) ? inv : e * inv;
            a[cur ^ 1][o] = sg;
            da[cur ^ 1][o] = sg * (1.0 - sg) * dz;
        }
        off += nin * nout + nout;
        nin = nout;
        cur ^= 1;
    }
    double y = w[off + nin], dy = 0.0;                         // output layer: W[nin][1], b[1]
    for (int i = 0; i < nin; ++i) {
        y = __builtin_fma(w[off + i], a[cur][i], y);
        dy = __builtin_fma(w[off + i], da[cur][i], dy);
    }
    return HnnTerm{so * (y - f0), so * dy};
}
CM_D Hard hardening(const cm_model_desc& m, double alpha) {
    Hard h; h.H = 0.0; h.dH = 0.0; h.expo = 0.0;
    if (m.has_voce) {
        h.expo = exp_s(-m.voce_D * alpha);
        h.H += m.voce_S * (1.0 - h.expo);
        h.dH += m.voce_S * m.voce_D * h.expo;
    }
    if (m.has_linear) { h.H += m.lin_K * alpha; h.dH += m.lin_K; }
    if constexpr (CM_HNN != 0) {
        if (m.hnn_width > 0) {                                  // uniform: the network hardening law, widths [1, H, 1] (cmad_hip.h)
            const HnnTerm t = (m.hnn_nhidden >= 2) ? hardening_network_deep(m, alpha)
                                                   : hardening_network(m.nn_weights + m.hnn_offset, m.hnn_width, alpha);
            h.H += t.H;
            h.dH += t.dH;
        }
    }
    return h;
}

// ---- single-precision seeds ---------------------------------------------------------------------------------------------------
// The scalar return maps that warm-start the local Newton (cm_structured.hpp, newton<> below) first run a few steps in float:
// 1/x, 1/sqrt(x), e^x and log(x) are ONE instruction each there (v_rcp_f32, v_rsq_f32, v_exp_f32, v_log_f32 against 5 / 7 / 22 /
// ~30 in double), and a seed good to 1e-6 leaves the double-precision loop one or two quadratically converging steps.  Nothing of
// it reaches the result except as a starting point: the map is polished in double and then checked by the reference's Newton.
#if defined(CM_HOST_BUILD)
CM_D float rcp_f(float a) { return 1.0f / a; }
CM_D float rsq_f(float a) { return 1.0f / std::sqrt(a); }
CM_D float exp_f(float a) { return std::exp(a); }
CM_D float log_f(float a) { return std::log(a); }
#else
CM_D float rcp_f(float a) { return __builtin_amdgcn_rcpf(a); }
CM_D float rsq_f(float a) { return __builtin_amdgcn_rsqf(a); }
CM_D float exp_f(float a) { return __builtin_amdgcn_exp2f(a * 1.4426950408889634f); }
CM_D float log_f(float a) { return __builtin_amdgcn_logf(a) * 0.6931471805599453f; }
#endif
// the Voce / linear hardening laws in float (the network law has no seed: hardening_seed_ok)
struct HardF { float H, dH; };
CM_D bool hardening_seed_ok(const cm_model_desc& m) { return CM_HNN == 0 || m.hnn_width <= 0; }     // uniform
CM_D HardF hardening_f(const cm_model_desc& m, float alpha) {
    HardF h; h.H = 0.0f; h.dH = 0.0f;
    if (m.has_voce) {
        const float S = (float)m.voce_S, D = (float)m.voce_D, e = exp_f(-D * alpha);
        h.H += S * (1.0f - e);
        h.dH += S * D * e;
    }
    if (m.has_linear) { h.H += (float)m.lin_K * alpha; h.dH += (float)m.lin_K; }
    return h;
}

// NN(s) of icnn_yield_term for a network [6, H, 1], VALUE only, hidden units in float (~34 instead of ~66 instructions per unit; a
// float copy of the weights in the pack was measured and bought nothing -- the kernel is no longer bound by VALU issue):
// what k_screen classifies with; a point whose yield function lands within the float evaluation's error band of zero is
// re-evaluated in double there.  Constants (biases of the output, f(0), scalers) stay double.
CM_D double icnn_value_f(const cm_model_desc& m, const double s[6]) {
    const int H = m.nn_widths[1];
    const cm_uniform_ptr u = uniform_ptr(m.nn_weights);
    const cm_uniform_ptr sc = u + icnn_off_scalers(H);
    const cm_uniform_ptr rec = u + icnn_off_records(H);
    const double h = (s[0] + s[3] + s[5]) * (1.0 / 3.0);
    const double x[6] = {s[0] - h, s[3] - h, s[5] - h, s[1], s[2], s[4]};
    float xs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xs[i] = (float)(sc[i] * x[i] + sc[6 + i]);
    float F = 0.0f;
    for (int o = 0; o < H; ++o) {
        const cm_uniform_ptr r = rec + kIcnnRec * o;
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) t += xs[i] * (float)r[i];
        const float b = (float)r[6], wz = (float)r[7];
        const float ap = b + t, an = b - t;
        const float spp = fmaxf(ap, 0.0f) + log_f(1.0f + exp_f(-fabsf(ap)));
        const float spn = fmaxf(an, 0.0f) + log_f(1.0f + exp_f(-fabsf(an)));
        F += (spp + spn) * wz;
    }
    return (0.5 * (double)F + u[icnn_off_b1(H)] - sc[14] - sc[13]) / sc[12];
}

// Cel a = 2 mu a + lambda (a0+a3+a5) d
CM_D void apply_cel(const cm_model_desc& m, const double a[6], double out[6]) {
    const double t = m.lambda * (a[0] + a[3] + a[5]), twomu = 2.0 * m.mu;
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = twomu * a[k] + (kDiag[k] ? t : 0.0);
}

// ---- state evaluation ----------------------------------------------------------------------
// Everything the residual and its derivatives need at one iterate.
template <int DEF>
struct Eval {
    double e[6];      // elastic strain (material frame)
    double s[6];      // material Cauchy stress
    double tr;        // tr(e)
    double phi, f, dgam;
    double gt[6];
    bool plastic;
    Hard hd;
};

// e = eg (+ (F33-1) z) - v ; s = lambda tr(e) d + 2 mu e   (elastic_stress.py:14-21)
template <int DEF>
CM_D void strain_stress(const cm_model_desc& m, const double eg[6], const double z[6], const double* x, Eval<DEF>& ev) {
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        // e = eg + (x7 - 1) Z^a + (x8 - 1) Z^b - Pi v,  Pi v = sum_i Z^i ((w o Z^i) . v)
        double t[3] = {0.0, x[7] - 1.0, x[8] - 1.0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double pv = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) pv += kW[k] * z[6 * i + k] * x[k];
            t[i] -= pv;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) ev.e[k] = eg[k] + t[0] * z[k] + t[1] * z[6 + k] + t[2] * z[12 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double e = eg[k] - x[k];
            if constexpr (DEF == CM_PLANE_STRESS) e += (x[7] - 1.0) * z[k];
            ev.e[k] = e;
        }
    }
    ev.tr = ev.e[0] + ev.e[3] + ev.e[5];
    const double twomu = 2.0 * m.mu;
#pragma unroll
    for (int k = 0; k < 6; ++k) ev.s[k] = twomu * ev.e[k] + (kDiag[k] ? m.lambda * ev.tr : 0.0);
}

// residual C(x) of cmad/models/small_elastic_plastic.py:237-302 with the branch select of
// cmad/models/paths.py:26-27 evaluated at the current iterate.  Ht filled when HESS.
template <int DEF, int YK, bool HESS>
CM_D void residual(const cm_model_desc& m, const double eg[6], const double z[6],
                   const double* x, const double* xp, Eval<DEF>& ev, double* C, double Ht[6][6]) {
    constexpr int NX = Dims<DEF>::NX;
    strain_stress<DEF>(m, eg, z, x, ev);
    yield_eval<YK, HESS>(m, ev.s, ev.phi, ev.gt, Ht);
    ev.hd = hardening(m, x[6]);
    const double i2mu = half_over_mu(m);
    ev.f = (ev.phi - (m.Y + ev.hd.H)) * i2mu;
    ev.dgam = x[6] - xp[6];
    ev.plastic = (ev.f > m.yield_tol) || (fabs(ev.f) < m.yield_tol);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double ce = x[k] - xp[k];
        C[k] = ev.plastic ? (ce - ev.dgam * ev.gt[k] * kIW[k]) : ce;
    }
    C[6] = ev.plastic ? ev.f : ev.dgam;
    if constexpr (DEF == CM_PLANE_STRESS) {
        // (Q s Q^T)[2][2] / 2mu = sum_k w_k z_k s_k / 2mu  (small_elastic_plastic.py:288-300)
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) r += kW[k] * z[k] * ev.s[k];
        C[NX - 1] = r * i2mu;
    }
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        // off-axis normal stresses (Q s Q^T)[a][a], [b][b] / 2mu  (small_elastic_plastic.py:291-297)
        double ra = 0.0, rb = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { ra += kW[k] * z[6 + k] * ev.s[k]; rb += kW[k] * z[12 + k] * ev.s[k]; }
        C[7] = ra * i2mu; C[8] = rb * i2mu;
    }
}

// A = dC/dx at the evaluated state (needs Ht).  TRANSPOSED stores A^T (for adjoint solves).
template <int DEF, bool TRANSPOSED>
CM_D void jacobian_x(const cm_model_desc& m, const double z[6], const Eval<DEF>& ev, const double Ht[6][6],
                     double (&A)[Dims<DEF>::NX][Dims<DEF>::NX]) {
    constexpr int NX = Dims<DEF>::NX;
    const double twomu = 2.0 * m.mu, i2mu = half_over_mu(m), lam = m.lambda;
#define CM_A(r, c) (TRANSPOSED ? A[c][r] : A[r][c])
#pragma unroll
    for (int r = 0; r < NX; ++r)
#pragma unroll
        for (int c = 0; c < NX; ++c) CM_A(r, c) = (r == c) ? 1.0 : 0.0;
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        // d s / d v = -Cel Pi, d s / d x7 = Cel Z^a, d s / d x8 = Cel Z^b ; constraint rows (w o Z^c) . s / 2mu
        double cz[2][6];
        apply_cel(m, z + 6, cz[0]);
        apply_cel(m, z + 12, cz[1]);
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            double pl[6], cp[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) pl[k] = kW[l] * (z[k] * z[l] + z[6 + k] * z[6 + l] + z[12 + k] * z[12 + l]);   // Pi[:, l]
            apply_cel(m, pl, cp);
            double ra = 0.0, rb = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { ra += kW[k] * z[6 + k] * cp[k]; rb += kW[k] * z[12 + k] * cp[k]; }
            CM_A(7, l) = -ra * i2mu; CM_A(8, l) = -rb * i2mu;
            if (ev.plastic) {
                double gcp = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    double h = 0.0;
#pragma unroll
                    for (int q = 0; q < 6; ++q) h += Ht[k][q] * cp[q];
                    CM_A(k, l) += ev.dgam * kIW[k] * h;
                    gcp += ev.gt[k] * cp[k];
                }
                CM_A(6, l) = -gcp * i2mu;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double ra = 0.0, rb = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { ra += kW[k] * z[6 + k] * cz[j][k]; rb += kW[k] * z[12 + k] * cz[j][k]; }
            CM_A(7, 7 + j) = ra * i2mu; CM_A(8, 7 + j) = rb * i2mu;
            CM_A(7, 6) = 0.0; CM_A(8, 6) = 0.0;
            if (ev.plastic) {
                double gcz = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    double h = 0.0;
#pragma unroll
                    for (int q = 0; q < 6; ++q) h += Ht[k][q] * cz[j][q];
                    CM_A(k, 7 + j) = -ev.dgam * kIW[k] * h;
                    gcz += ev.gt[k] * cz[j][k];
                }
                CM_A(6, 7 + j) = gcz * i2mu;
            } else {
#pragma unroll
                for (int k = 0; k < 7; ++k) CM_A(k, 7 + j) = 0.0;
            }
        }
        if (ev.plastic) {
#pragma unroll
            for (int k = 0; k < 6; ++k) CM_A(k, 6) = -ev.gt[k] * kIW[k];
            CM_A(6, 6) = -ev.hd.dH * i2mu;
        }
        return;
    }
    if (ev.plastic) {
        double gd = ev.gt[0] + ev.gt[3] + ev.gt[5];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double hd = Ht[k][0] + Ht[k][3] + Ht[k][5];
            const double sc = ev.dgam * kIW[k];
#pragma unroll
            for (int l = 0; l < 6; ++l)
                CM_A(k, l) += sc * (twomu * Ht[k][l] + (kDiag[l] ? lam * hd : 0.0));
            CM_A(k, 6) = -ev.gt[k] * kIW[k];
            CM_A(6, k) = -(ev.gt[k] + (kDiag[k] ? lam * gd * i2mu : 0.0));
        }
        CM_A(6, 6) = -ev.hd.dH * i2mu;
    }
    if constexpr (DEF == CM_PLANE_STRESS) {
        // Cel z and Cel (w o z);  z0 + z3 + z5 = |q3|^2
        const double zt = z[0] + z[3] + z[5];
        double cz[6], czw[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            cz[k] = twomu * z[k] + (kDiag[k] ? lam * zt : 0.0);
            czw[k] = twomu * kW[k] * z[k] + (kDiag[k] ? lam * zt : 0.0);
        }
        double zcz = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) zcz += kW[k] * z[k] * cz[k];
#pragma unroll
        for (int l = 0; l < 6; ++l) CM_A(7, l) = -czw[l] * i2mu;
        CM_A(7, 6) = 0.0;
        CM_A(7, 7) = zcz * i2mu;
        if (ev.plastic) {
            double gcz = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double hcz = 0.0;
#pragma unroll
                for (int l = 0; l < 6; ++l) hcz += Ht[k][l] * cz[l];
                CM_A(k, 7) = -ev.dgam * kIW[k] * hcz;
                gcz += ev.gt[k] * cz[k];
            }
            CM_A(6, 7) = gcz * i2mu;
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) CM_A(k, 7) = 0.0;
        }
    }
#undef CM_A
}

// ---- rate-form model: unknown is the material Cauchy stress ------------------------------------------------
// cmad/models/small_rate_elastic_plastic.py:249-346.  x = [sigma(6), alpha (, F33)]; `deg` is the material-frame
// total-strain increment V(Q^T (eps - eps_prev) Q) (:34-76; PLANE_STRESS adds (F33 - F33_prev) z).
//   C_e = [ (sigma - sigma_prev - Cel de) / 2mu , dgam ]
//   C_p = [ (sigma - sigma_prev - Cel de + dgam Cel n) / 2mu , f(sigma, alpha) ],  n = W^-1 gt(sigma)
//   PLANE_STRESS row: (w o z) . (Cel de [- dgam Cel n]) / 2mu
template <int DEF, int YK, bool HESS>
CM_D void residual_rate(const cm_model_desc& m, const double deg[6], const double z[6],
                        const double* x, const double* xp, Eval<DEF>& ev, double* C, double Ht[6][6]) {
    constexpr int NX = Dims<DEF>::NX;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        ev.s[k] = x[k];
        double e = deg[k];
        if constexpr (DEF == CM_PLANE_STRESS) e += (x[7] - xp[7]) * z[k];
        ev.e[k] = e;                                             // strain increment
    }
    ev.tr = ev.e[0] + ev.e[3] + ev.e[5];
    yield_eval<YK, HESS>(m, ev.s, ev.phi, ev.gt, Ht);
    ev.hd = hardening(m, x[6]);
    const double i2mu = half_over_mu(m), twomu = 2.0 * m.mu;
    ev.f = (ev.phi - (m.Y + ev.hd.H)) * i2mu;
    ev.dgam = x[6] - xp[6];
    ev.plastic = (ev.f > m.yield_tol) || (fabs(ev.f) < m.yield_tol);
    const double gd = ev.gt[0] + ev.gt[3] + ev.gt[5];
    double r7 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double tdc = twomu * ev.e[k] + (kDiag[k] ? m.lambda * ev.tr : 0.0);
        const double cn = twomu * ev.gt[k] * kIW[k] + (kDiag[k] ? m.lambda * gd : 0.0);
        const double dc = ev.plastic ? (tdc - ev.dgam * cn) : tdc;   // select, not multiply: the normal is NaN at sigma = 0
        C[k] = (x[k] - xp[k] - dc) * i2mu;
        if constexpr (DEF == CM_PLANE_STRESS) r7 += kW[k] * z[k] * dc;
    }
    C[6] = ev.plastic ? ev.f : ev.dgam;
    if constexpr (DEF == CM_PLANE_STRESS) C[NX - 1] = r7 * i2mu;
}

template <int DEF, bool TRANSPOSED>
CM_D void jacobian_rate(const cm_model_desc& m, const double z[6], const Eval<DEF>& ev, const double Ht[6][6],
                        double (&A)[Dims<DEF>::NX][Dims<DEF>::NX]) {
    constexpr int NX = Dims<DEF>::NX;
    const double twomu = 2.0 * m.mu, i2mu = half_over_mu(m), lam = m.lambda;
#define CM_A(r, c) (TRANSPOSED ? A[c][r] : A[r][c])
#pragma unroll
    for (int r = 0; r < NX; ++r)
#pragma unroll
        for (int c = 0; c < NX; ++c) CM_A(r, c) = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) CM_A(k, k) = i2mu;
    CM_A(6, 6) = 1.0;
    double M[6][6];                                  // dgam Cel W^-1 Ht (plastic) -- also feeds the PS row
    const double gd = ev.gt[0] + ev.gt[3] + ev.gt[5];
    if (ev.plastic) {
        double hd[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) hd[l] = Ht[0][l] + Ht[3][l] + Ht[5][l];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int l = 0; l < 6; ++l) {
                M[k][l] = ev.dgam * (twomu * Ht[k][l] * kIW[k] + (kDiag[k] ? lam * hd[l] : 0.0));
                CM_A(k, l) += M[k][l] * i2mu;
            }
            CM_A(k, 6) = (twomu * ev.gt[k] * kIW[k] + (kDiag[k] ? lam * gd : 0.0)) * i2mu;
            CM_A(6, k) = ev.gt[k] * i2mu;
        }
        CM_A(6, 6) = -ev.hd.dH * i2mu;
    }
    if constexpr (DEF == CM_PLANE_STRESS) {
        const double zt = z[0] + z[3] + z[5];
        double zcz = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double cz = twomu * z[k] + (kDiag[k] ? lam * zt : 0.0);
            CM_A(k, 7) = -cz * i2mu;
            zcz += kW[k] * z[k] * cz;
        }
        CM_A(7, 7) = zcz * i2mu;
        if (ev.plastic) {
            double cnz = 0.0;
#pragma unroll
            for (int l = 0; l < 6; ++l) {
                double sM = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) sM += kW[k] * z[k] * M[k][l];
                CM_A(7, l) = -sM * i2mu;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) cnz += kW[k] * z[k] * (twomu * ev.gt[k] * kIW[k] + (kDiag[k] ? lam * gd : 0.0));
            CM_A(7, 6) = -cnz * i2mu;
        }
    }
#undef CM_A
}

// model-kind front doors used by the generic (dense) Newton
template <int MK, int DEF, int YK, bool HESS>
CM_D void residual_mk(const cm_model_desc& m, const double eg[6], const double z[6], const double* x, const double* xp,
                      Eval<DEF>& ev, double* C, double Ht[6][6]) {
    if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) residual_rate<DEF, YK, HESS>(m, eg, z, x, xp, ev, C, Ht);
    else residual<DEF, YK, HESS>(m, eg, z, x, xp, ev, C, Ht);
}
template <int MK, int DEF>
CM_D void jacobian_mk(const cm_model_desc& m, const double z[6], const Eval<DEF>& ev, const double Ht[6][6],
                      double (&A)[Dims<DEF>::NX][Dims<DEF>::NX]) {
    if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) jacobian_rate<DEF, false>(m, z, ev, Ht, A);
    else jacobian_x<DEF, false>(m, z, ev, Ht, A);
}

// ---- dense N x N solves, fully unrolled, no pivoting ---------------------------------------------
// (the leading 6x6 block is I + dgam * (PSD-like); a vanishing pivot is reported)
template <int N>
CM_D bool lu_factor(double (&A)[N][N]) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double piv = A[k][k];
        ok = ok && (fabs(piv) > 1e-300);
        const double ip = rcp(piv);
        A[k][k] = ip;                       // store reciprocal pivot
#pragma unroll
        for (int r = k + 1; r < N; ++r) {
            const double l = A[r][k] * ip;
            A[r][k] = l;
#pragma unroll
            for (int c = k + 1; c < N; ++c) A[r][c] -= l * A[k][c];
        }
    }
    return ok;
}
template <int N>
CM_D void lu_subst(const double (&A)[N][N], double (&b)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int r = k + 1; r < N; ++r) b[r] -= A[r][k] * b[k];
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        double s = b[k];
#pragma unroll
        for (int c = k + 1; c < N; ++c) s -= A[k][c] * b[c];
        b[k] = s * A[k][k];
    }
}

// ---- UNIAXIAL_STRESS (total form): the Newton step A delta = C through a 4 x 4 system -------------------------------------
// A = jacobian_x<UNIAXIAL_STRESS> depends on delta v only through the three projections d tau_i = (w o Z^i) . delta v, because the
// elastic strain is e = eg + sum_i Z^i theta_i with theta = (-tau_0, x7 - 1 - tau_1, x8 - 1 - tau_2).  Writing the rows of A in
// d theta and d alpha (hs_i = Ht Cel Z^i, G_ji = Z^j . hs_i, n_j = Z^j . gt, gc_i = gt . Cel Z^i, K_ji = (w o Z^j) . Cel Z^i):
//     delta v = C_v + dgam w^-1 o (sum_i hs_i d theta_i) + w^-1 o gt d alpha                      (rows 0..5, plastic)
//     (1 + dgam G_00) d theta_0 + dgam (G_01 d theta_1 + G_02 d theta_2) + n_0 d alpha = -(w o Z^0) . C_v   (their projection on Z^0)
//     sum_i K_1i d theta_i = 2mu C_7 ,  sum_i K_2i d theta_i = 2mu C_8                             (rows 7, 8)
//     sum_i gc_i d theta_i - H' d alpha = 2mu C_6                                                 (row 6; elastic: d alpha = C_6)
// then d tau_1,2 from the projections on Z^1, Z^2 and delta x7 = d theta_1 + d tau_1, delta x8 = d theta_2 + d tau_2.  The same
// delta as the 9 x 9 LU (exact elimination, not an approximation): ~300 instead of ~900 operations per iteration and no 81-entry
// matrix in registers.  Works for every yield surface (the Hessian enters as the 6 x 6 Ht).
// Factored form: everything that does not depend on the right-hand side, for the step (apply), for solves with A^T (apply_T:
// adjoint) and for several right-hand sides (tangent).
struct UniaxialOp {
    double hs[3][6];          // Ht Cel Z^i (zero on the elastic branch)
    double G[3][3], n[3], gc[3];
    double M[4][4];           // LU factors of the 4 x 4 system in (d theta_0..2, d alpha)
    double T[3][3];           // LU factors of the 3 x 3 system of the transposed solve in (mu_0, lam_7, lam_8)
    double gtw[6];            // w^-1 o gt (zero on the elastic branch)
    double pg, a66, d6;
    bool ok;
};
// the quadratic surfaces (J2, Hill: Ht = (A - gt gt^T) / phi) apply the Hessian from that form: no 6 x 6 array (72 registers) is
// built, and the callers evaluate the residual without its Hessian (uniaxial_needs_hessian<YK>() == false)
template <int YK>
constexpr bool uniaxial_needs_hessian() { return !(YK == CM_YIELD_J2 || YK == CM_YIELD_HILL); }
template <int YK>
CM_D void uniaxial_setup(const cm_model_desc& m, const double z[18], const Eval<CM_UNIAXIAL_STRESS>& ev, const double Ht[6][6],
                         UniaxialOp& op, bool want_transposed) {
    const double i2mu = half_over_mu(m);
    double cz[3][6], K[2][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) apply_cel(m, z + 6 * i, cz[i]);
    op.pg = ev.plastic ? ev.dgam : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double gc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) gc += ev.gt[k] * cz[i][k];
        if constexpr (uniaxial_needs_hessian<YK>()) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double h = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) h += Ht[k][q] * cz[i][q];
                op.hs[i][k] = ev.plastic ? h : 0.0;
            }
        } else {
            const QuadForm qf = quad_form<YK>(m);
            const double ip = (ev.plastic && ev.phi > 0.0) ? rcp(ev.phi) : 0.0;
            const double* c = cz[i];
            double Ac[6];
            Ac[0] = qf.a00 * c[0] + qf.a03 * c[3] + qf.a05 * c[5];
            Ac[3] = qf.a03 * c[0] + qf.a33 * c[3] + qf.a35 * c[5];
            Ac[5] = qf.a05 * c[0] + qf.a35 * c[3] + qf.a55 * c[5];
            Ac[1] = qf.a11 * c[1]; Ac[2] = qf.a22 * c[2]; Ac[4] = qf.a44 * c[4];
#pragma unroll
            for (int k = 0; k < 6; ++k) op.hs[i][k] = ip * (Ac[k] - ev.gt[k] * gc);
        }
        op.gc[i] = ev.plastic ? gc * i2mu : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) op.gtw[k] = ev.plastic ? ev.gt[k] * kIW[k] : 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double nj = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) nj += z[6 * j + k] * ev.gt[k];
        op.n[j] = ev.plastic ? nj : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double g = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) g += z[6 * j + k] * op.hs[i][k];
            op.G[j][i] = g;
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double kk = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) kk += kW[k] * z[6 * (c + 1) + k] * cz[i][k];
            K[c][i] = kk * i2mu;
        }
    op.a66 = ev.plastic ? -ev.hd.dH * i2mu : 1.0;
    op.M[0][0] = 1.0 + op.pg * op.G[0][0]; op.M[0][1] = op.pg * op.G[0][1]; op.M[0][2] = op.pg * op.G[0][2]; op.M[0][3] = op.n[0];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < 3; ++i) op.M[1 + c][i] = K[c][i];
        op.M[1 + c][3] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) op.M[3][i] = op.gc[i];
    op.M[3][3] = op.a66;
    op.ok = lu_factor<4>(op.M);
    op.d6 = op.n[0] * rcp(op.a66);
    if (want_transposed) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            op.T[i][0] = ((i == 0) ? 1.0 : 0.0) + op.pg * op.G[0][i] - op.gc[i] * op.d6;
            op.T[i][1] = -K[0][i];
            op.T[i][2] = -K[1][i];
        }
        op.ok = lu_factor<3>(op.T) && op.ok;
    }
}
// delta = A^-1 C
CM_D void uniaxial_apply(const UniaxialOp& op, const double z[18], const double* C, double* delta) {
    double ct[3], r[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double cj = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) cj += kW[k] * z[6 * j + k] * C[k];
        ct[j] = cj;
    }
    r[0] = -ct[0]; r[1] = C[7]; r[2] = C[8]; r[3] = C[6];
    lu_subst<4>(op.M, r);                                         // r = (d theta_0, d theta_1, d theta_2, d alpha)
#pragma unroll
    for (int k = 0; k < 6; ++k)
        delta[k] = C[k] + kIW[k] * op.pg * (op.hs[0][k] * r[0] + op.hs[1][k] * r[1] + op.hs[2][k] * r[2]) + op.gtw[k] * r[3];
    delta[6] = r[3];
#pragma unroll
    for (int j = 1; j < 3; ++j)
        delta[6 + j] = r[j] + ct[j] + op.pg * (op.G[j][0] * r[0] + op.G[j][1] * r[1] + op.G[j][2] * r[2]) + op.n[j] * r[3];
}
// lam = A^-T b (in place allowed).  With mu = L^T lam (the three multipliers of d theta): lam_v = b_v + V mu, mu_1 = b_7, mu_2 = b_8,
// lam_6 = (b_6 + n~ . lam_v) / a66, and (mu_0, lam_7, lam_8) from the 3 x 3 system T.
CM_D void uniaxial_apply_T(const UniaxialOp& op, const double z[18], const double* b, double* lam) {
    double cv[6], q[3];
    const double b6 = b[6], b7 = b[7], b8 = b[8];
    double ncv = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        cv[k] = b[k] + kW[k] * (z[6 + k] * b7 + z[12 + k] * b8);
        ncv += op.gtw[k] * cv[k];
    }
    const double c6 = (b6 + ncv) * rcp(op.a66);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double uc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) uc += kIW[k] * op.hs[i][k] * cv[k];
        q[i] = -op.pg * uc + op.gc[i] * c6 - ((i == 1) ? b7 : 0.0) - ((i == 2) ? b8 : 0.0);
    }
    lu_subst<3>(op.T, q);                                         // q = (mu_0, lam_7, lam_8)
#pragma unroll
    for (int k = 0; k < 6; ++k) lam[k] = cv[k] + kW[k] * z[k] * q[0];
    lam[6] = c6 + op.d6 * q[0];
    lam[7] = q[1];
    lam[8] = q[2];
}
template <int YK>
CM_D bool uniaxial_solve(const cm_model_desc& m, const double z[18], const Eval<CM_UNIAXIAL_STRESS>& ev, const double Ht[6][6],
                         const double* C, double* delta) {
    UniaxialOp op;
    uniaxial_setup<YK>(m, z, ev, Ht, op, false);
    uniaxial_apply(op, z, C, delta);
    return op.ok;
}

template <int N>
CM_D double norm2(const double* v) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) s += v[k] * v[k];
    return sqrt(s);
}
template <int N>
CM_D double dot(const double* a, const double* b) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) s += a[k] * b[k];
    return s;
}

// cmad/util/line_search.py:74-85
CM_D double quad_min(double phi0, double dphi0, double a, double phi) {
    const double denom = 2.0 * (phi - phi0 - dphi0 * a);
    const double safe = (denom == 0.0) ? 1.0 : denom;
    return (denom == 0.0) ? 0.5 * a : (-dphi0 * a * a / safe);
}

// One trial of the LEGACY backtracking of newton_solve (cmad/models/nonlinear_solver.py:55-81), shared by every solver form.
// `phi` = psi_j = ||C||^2 / 2 at the trial just evaluated (step length `alpha`, the n-th evaluation so far counting from 0),
// `cc` = ||C||^2 at the base iterate (psi_0 = cc / 2, psi_0' = -cc).  Returns 0 when the trial is kept -- accepted,
// psi_j < (1 - 2 beta alpha) psi_0 (a NaN merit compares false in the reference's `>=` and is kept as well), or the evaluations
// have run out ("reached max ls evals": the state stays at the last EVALUATED alpha) -- else the increment (alpha_new - alpha)
// by which the trial moves along the Newton direction (x <- x - increment * delta, as the reference's add_to_xi does), with
// alpha_new = max(eta alpha, -alpha^2 psi_0' / (2 (psi_j - psi_0 - alpha psi_0'))).  beta = m.ls_c1, eta = m.ls_lo.
CM_D double ls_trial_legacy(const cm_model_desc& m, double phi, double cc, double& alpha, int& n) {
    const double psi0 = 0.5 * cc;
    ++n;
    if (!(phi >= (1.0 - 2.0 * m.ls_c1 * alpha) * psi0)) return 0.0;
    const double a_new = fmax(m.ls_lo * alpha, (alpha * alpha * cc) / (2.0 * (phi - psi0 + alpha * cc)));
    if (n >= m.ls_max_evals) return 0.0;
    const double inc = a_new - alpha;
    alpha = a_new;
    return inc;
}

// ---- local Newton ------------------------------------------------------------------------------
// make_newton_solve (cmad/models/nonlinear_solver.py:102-155) with the quadratic Armijo line search
// of cmad/util/line_search.py:95-189 (ls_max_evals > 0) or the plain Newton of newton_solve
// (:14-85, ls_max_evals == 0).  One point per lane; wave-level ballots (__any) drive the loops so an
// all-elastic / all-converged wavefront leaves at once and the rest iterate under the exec mask.
// Returns the status word.
// FAST: UNIAXIAL_STRESS, total form: the step through uniaxial_solve (4 x 4) instead of the 9 x 9 LU (same delta).  The kernels
// always use it; FAST = false is the dense reference path of the host tests.  No effect on the other configurations.
// UNIAXIAL_STRESS with a quadratic surface (J2, Hill): the 1-d return map as warm start (include/cmad_hip.h "Warm starts";
// CM_SOLVER_REFERENCE_ITERATES / CM_SOLVER_GENERAL_NEWTON opt out).  The elastic strain is e = sum_i c_i Z^i in the frame dyads
// (strain_stress), so with isotropic elasticity and the two lateral constraint rows the stress at the solution is sigma_0 Z^0:
//     c_1 = c_2 = -nu c_0 ,  sigma_0 = E c_0 ,  phi = h0 |sigma_0| ,  h0^2 = Z^0 . A Z^0 ,  n = sgn(sigma_0) N ,  N_k = (A Z^0)_k / (h0 w_k)
// a CONSTANT flow direction, whose projections on the dyads are nu_i = Z^i . A Z^0 / h0 (nu_0 = h0).  One scalar equation
//     F(dgam) = h0 E (|c_0,trial| - h0 dgam) - Y - H(alpha_prev + dgam) = 0        (linear in dgam but for the hardening law)
// and the state follows: v = v_prev + dgam sgn N, alpha, x_7 = 1 + tau_1 + c_1, x_8 = 1 + tau_2 + c_2 (tau_i = (w o Z^i) . v).
// Returns true when x0 holds a warm start.  `plastic_prev`: the branch the reference's first iterate takes at x_prev (with the OLD
// stretches): a point plastic there but elastic once the stretches relax has two roots of the residual (cf. newton_j2_plane) and
// is left to the reference's iteration.
template <int YK>
CM_D bool uniaxial_warm_start(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, bool plastic_prev,
                              bool active, double* x0) {
    static_assert(YK == CM_YIELD_J2 || YK == CM_YIELD_HILL, "quadratic surfaces");
    const QuadForm q = quad_form<YK>(m);
    const double* Z0 = z;
    double AZ[6];
    AZ[0] = q.a00 * Z0[0] + q.a03 * Z0[3] + q.a05 * Z0[5];
    AZ[3] = q.a03 * Z0[0] + q.a33 * Z0[3] + q.a35 * Z0[5];
    AZ[5] = q.a05 * Z0[0] + q.a35 * Z0[3] + q.a55 * Z0[5];
    AZ[1] = q.a11 * Z0[1]; AZ[2] = q.a22 * Z0[2]; AZ[4] = q.a44 * Z0[4];
    double h2 = 0.0, n1 = 0.0, n2 = 0.0, c0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        h2 += Z0[k] * AZ[k]; n1 += z[6 + k] * AZ[k]; n2 += z[12 + k] * AZ[k];
        c0 += kW[k] * Z0[k] * (eg[k] - xp[k]);                   // c_0 at x_prev: axial strain minus tau_0
        t1 += kW[k] * z[6 + k] * xp[k]; t2 += kW[k] * z[12 + k] * xp[k];
    }
    if (!(h2 > 0.0)) return false;
    const double h0 = sqrt(h2), ih0 = rcp(h0);
    n1 *= ih0; n2 *= ih0;
    const double lm = m.lambda + m.mu, Em = m.mu * (3.0 * m.lambda + 2.0 * m.mu) * rcp(lm), nup = 0.5 * m.lambda * rcp(lm);
    const double sgn = (c0 >= 0.0) ? 1.0 : -1.0, a0 = fabs(c0), alpha_p = xp[6];
    double dg = 0.0;
    bool done = !active, ok = false;
    for (int it = 0; it < 12; ++it) {
        const Hard h = hardening(m, alpha_p + dg);
        const double F = h0 * Em * (a0 - h0 * dg) - (m.Y + h.H), dF = -h2 * Em - h.dH;
        if (!done && it == 0 && !(F > 0.0)) { done = true; ok = !plastic_prev; }      // elastic step: dgam = 0, relaxed stretches
        const double res = fabs(F) * rcp(m.Y + h.H);
        if (!done && !(res < 1e300)) done = true;
        if (!done && res < 1e-14) { done = true; ok = true; }
        const bool last = res < 1e-8;                              // quadratic convergence: the next iterate is converged to round-off
        if (!done) {
            dg = fmax(dg - F * rcp(dF), 0.0);
            if (last) { done = true; ok = true; }
        }
        if (!__any(!done)) break;
    }
    if (!ok) return false;
    const double c0n = sgn * (a0 - h0 * dg), lat = -nup * c0n, sdg = sgn * dg;
#pragma unroll
    for (int k = 0; k < 6; ++k) x0[k] = xp[k] + sdg * AZ[k] * ih0 * kIW[k];
    x0[6] = alpha_p + dg;
    x0[7] = 1.0 + t1 + sdg * n1 + lat;
    x0[8] = 1.0 + t2 + sdg * n2 + lat;
    return true;
}

template <int DEF, int YK, int MK, bool LS, bool FAST = false>
CM_D uint32_t newton(const cm_model_desc& m, const double eg[6], const double z[6], const double* xp, double* x,
                     bool lane_valid) {
    constexpr int NX = Dims<DEF>::NX;
    Eval<DEF> ev;
    double C[NX], Ht[6][6];
#pragma unroll
    for (int k = 0; k < NX; ++k) x[k] = xp[k];
    residual_mk<MK, DEF, YK, false>(m, eg, z, x, xp, ev, C, Ht);
    // ||C||/||C0|| < rel_tol or ||C|| < abs_tol (nonlinear_solver.py:140-150), tested on squared norms
    // (no sqrt, no division; 0/0 -> NaN -> false in the reference == 0 < 0 -> false here)
    const double n0sq = dot<NX>(C, C);
    const double rel2 = m.rel_tol * m.rel_tol * n0sq, abs2 = m.abs_tol * m.abs_tol;
    int it = 0;
    bool running = lane_valid;
    uint32_t flags = 0;
    if constexpr (FAST && DEF == CM_UNIAXIAL_STRESS && MK == CM_SMALL_ELASTIC_PLASTIC && (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL)) {
        // warm start (uniform switch): the 1-d return map, then the same loop from its result -- the test below, on the reference's
        // residual and against ||C(x_prev)||, decides.  A state that already passes the test at x_prev stays as it is.
        const bool warm_on = !(m.solver_flags & (CM_SOLVER_REFERENCE_ITERATES | CM_SOLVER_GENERAL_NEWTON)) &&
                             !(m.ls_max_evals > 0 && m.ls_kind == CM_LS_LEGACY);
        if (warm_on) {
            double x0[NX];
            const bool act = lane_valid && !((n0sq < rel2) || (n0sq < abs2));
            if (__any(act)) {
                const bool warm = uniaxial_warm_start<YK>(m, eg, z, xp, ev.plastic, act, x0);
                if (warm && act) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) x[k] = x0[k];
                }
                residual_mk<MK, DEF, YK, false>(m, eg, z, x, xp, ev, C, Ht);
            }
        }
    }
    for (;;) {
        const double nsq = dot<NX>(C, C);
        const bool conv = (nsq < rel2) || (nsq < abs2);
        if (running && conv) { running = false; flags |= CM_STATUS_CONVERGED; }
        if (running && it >= m.max_iters) running = false;
        if (!__any(running)) break;
        if (running) {
            double delta[NX];
            constexpr bool UXS = FAST && DEF == CM_UNIAXIAL_STRESS && MK == CM_SMALL_ELASTIC_PLASTIC;
            residual_mk<MK, DEF, YK, !UXS || uniaxial_needs_hessian<YK>()>(m, eg, z, x, xp, ev, C, Ht);
            if constexpr (UXS) {
                if (!uniaxial_solve<YK>(m, z, ev, Ht, C, delta)) flags |= CM_STATUS_SINGULAR;
            } else {
                double A[NX][NX];
                jacobian_mk<MK, DEF>(m, z, ev, Ht, A);
                if (!lu_factor<NX>(A)) flags |= CM_STATUS_SINGULAR;
#pragma unroll
                for (int k = 0; k < NX; ++k) delta[k] = C[k];
                lu_subst<NX>(A, delta);
            }
            // LS kernels serve plain Newton too (ls_max_evals == 0, uniform): the cold configurations are built once
            bool plain = !LS;
            if constexpr (LS) plain = (m.ls_max_evals <= 0);
            if (plain) {
#pragma unroll
                for (int k = 0; k < NX; ++k) x[k] -= delta[k];
                residual_mk<MK, DEF, YK, false>(m, eg, z, x, xp, ev, C, Ht);
            } else if constexpr (LS) {
                const double cc = dot<NX>(C, C);
                const double phi0 = 0.5 * cc, dphi0 = -cc, armijo = m.ls_c1 * dphi0;
                int n = 0;
                double alpha = 1.0, best_alpha = 1.0, best_phi = INFINITY;
                bool accepted = false;
                double Cbest[NX], Ct[NX], xt[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) { Cbest[k] = C[k]; Ct[k] = C[k]; }
                bool ls = true;
                const bool legacy = (m.ls_kind == CM_LS_LEGACY);      // uniform
                while (__any(ls)) {
                    if (ls) {
#pragma unroll
                        for (int k = 0; k < NX; ++k) xt[k] = x[k] - alpha * delta[k];
                        Eval<DEF> et;
                        residual_mk<MK, DEF, YK, false>(m, eg, z, xt, xp, et, Ct, Ht);
                        const double phi = 0.5 * dot<NX>(Ct, Ct);
                        if (legacy) {                                  // newton_solve's backtracking: the last evaluated trial is kept
                            const double a_eval = alpha;
                            if (ls_trial_legacy(m, phi, cc, alpha, n) == 0.0) { alpha = a_eval; accepted = true; ls = false; }
                            continue;
                        }
                        const bool finite = isfinite(phi);
                        if (finite && phi < best_phi) {
                            best_alpha = alpha; best_phi = phi;
#pragma unroll
                            for (int k = 0; k < NX; ++k) Cbest[k] = Ct[k];
                        }
                        accepted = finite && (phi <= phi0 + alpha * armijo);
                        const double am = quad_min(phi0, dphi0, alpha, phi);
                        const double ac = fmin(fmax(am, m.ls_lo * alpha), m.ls_hi * alpha);
                        if (!accepted) alpha = finite ? ac : 0.5 * alpha;
                        ++n;
                        ls = (n < m.ls_max_evals) && !accepted;
                    }
                }
                const double ra = accepted ? alpha : best_alpha;
#pragma unroll
                for (int k = 0; k < NX; ++k) { x[k] -= ra * delta[k]; C[k] = accepted ? Ct[k] : Cbest[k]; }
            }
            ++it;
        }
    }
    return flags | (uint32_t)it;
}

// global Cauchy stress 6-vector from the material one (small_elastic_plastic.py:318-319)
template <bool ROT>
CM_D void to_global(const cm_model_desc& m, const double s[6], double out[6]) {
    if constexpr (ROT) congruence<false>(m.Q, s, out);
    else {
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = s[k];
    }
}
// pull a cotangent of the 6 stored global entries back to the material 6-vector:
// sbar_m[k] = sum_r sbar[r] d sig_g[r] / d s[k] = w_k V(Q^T Sbar Q)_k with Sbar_ij = sbar_ij / w_ij
template <bool ROT>
CM_D void cotangent_to_material(const cm_model_desc& m, const double sb[6], double out[6]) {
    if constexpr (ROT) {
        double t[6], r[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = sb[k] * kIW[k];
        congruence<true>(m.Q, t, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = kW[k] * r[k];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = sb[k];
    }
}

// ---- reverse sweep at a converged state -------------------------------------------------------------
// Given the cotangent sbar_m of the MATERIAL stress 6-vector and an incoming cotangent xin of xi
// (NULL = 0), solve  lam = A^-T ( (d s/d x)^T sbar_m + xin )  and return
//   pbar[j]  = sbar_m . d s/d p_j - lam . dC/dp_j            (KP order, CM_NUM_PARAMS entries)
//   xpbar    = -(dC/dx_prev)^T lam
//   egbar    = Cel sbar_m - (dC/d eg)^T lam                   (cotangent of the material total strain)
// This is the transpose of the IFT rule cmad/models/nonlinear_solver.py:158-171 and one step of
// cmad/objectives/mp_objective.py:112-142 (with phi = -lam, history = -xin).
// FAST (UNIAXIAL_STRESS only): A^-T through the factored 4 x 4 / 3 x 3 form (uniaxial_apply_T) instead of the 9 x 9 LU.
template <int DEF, int YK, bool FAST = true>
CM_D bool reverse_point(const cm_model_desc& m, const double eg[6], const double z[6],
                        const double* x, const double* xp, const double sbm[6], const double* xin,
                        double* pbar, double* xpbar, double* egbar, double* lam_out = nullptr) {
    constexpr int NX = Dims<DEF>::NX;
    constexpr bool UX = FAST && DEF == CM_UNIAXIAL_STRESS;
    Eval<DEF> ev;
    double C[NX], Ht[6][6], At[UX ? 1 : NX][UX ? 1 : NX], lam[NX];
    residual<DEF, YK, true>(m, eg, z, x, xp, ev, C, Ht);          // Ht is also applied to u below
    UniaxialOp uop;
    bool ok;
    if constexpr (UX) { uniaxial_setup<YK>(m, z, ev, Ht, uop, true); ok = uop.ok; }
    else { jacobian_x<DEF, true>(m, z, ev, Ht, At); ok = lu_factor<NX>(At); }
    double csb[6];
    apply_cel(m, sbm, csb);                                  // Cel sbar_m (Cel symmetric)
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        // d s / d v = -Cel Pi (Pi[:, l] = w_l sum_i Z^i Z^i_l), d s / d x7 = Cel Z^a, d s / d x8 = Cel Z^b
        const double zc[3] = {dot<6>(z, csb), dot<6>(z + 6, csb), dot<6>(z + 12, csb)};
#pragma unroll
        for (int l = 0; l < 6; ++l) lam[l] = -kW[l] * (z[l] * zc[0] + z[6 + l] * zc[1] + z[12 + l] * zc[2]);
        lam[7] = zc[1]; lam[8] = zc[2];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) lam[k] = -csb[k];        // d s / d v = -Cel
    }
    lam[6] = 0.0;
    if constexpr (DEF == CM_PLANE_STRESS) lam[7] = dot<6>(z, csb);   // d s / d F33 = Cel z
    if (xin) {
#pragma unroll
        for (int k = 0; k < NX; ++k) lam[k] += xin[k];
    }
    if constexpr (UX) uniaxial_apply_T(uop, z, lam, lam);
    else lu_subst<NX>(At, lam);
    if (lam_out) {                                           // the adjoint vector of this step (phi = -lam in the reference)
#pragma unroll
        for (int k = 0; k < NX; ++k) lam_out[k] = lam[k];
    }
    const double i2mu = half_over_mu(m);
    // ---- u_k = -dgam lam_k / w_k (plastic rows), hu = Ht u
    double u[6], hu[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = ev.plastic ? (-ev.dgam * lam[k] * kIW[k]) : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l) s += Ht[k][l] * u[l];
        hu[k] = ev.plastic ? s : 0.0;
    }
    const double lam6 = ev.plastic ? lam[6] : 0.0;           // elastic: dC_6/d(anything but x) = 0
    if (pbar) {
        const double gd = ev.gt[0] + ev.gt[3] + ev.gt[5], ge = dot<6>(ev.gt, ev.e);
        const double hud = hu[0] + hu[3] + hu[5], hue = dot<6>(hu, ev.e);
        const double sbd = sbm[0] + sbm[3] + sbm[5], sbe = dot<6>(sbm, ev.e);
        // lam . dC/dp
        double cl = hud * ev.tr + lam6 * gd * ev.tr * i2mu;                        // lambda
        double cm_ = 2.0 * hue + lam6 * (2.0 * ge * i2mu - ev.f * 2.0 * i2mu);           // mu
        if constexpr (DEF == CM_PLANE_STRESS) {
            const double zt = z[0] + z[3] + z[5];
            double zwe = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) zwe += kW[k] * z[k] * ev.e[k];
            cl += lam[7] * zt * ev.tr * i2mu;
            cm_ += lam[7] * (2.0 * zwe * i2mu - C[7] * 2.0 * i2mu);
        }
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {                    // the two off-axis normal-stress rows
                const double* Z = z + 6 * (1 + j);
                const double zt = Z[0] + Z[3] + Z[5];
                double zwe = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) zwe += kW[k] * Z[k] * ev.e[k];
                cl += lam[7 + j] * zt * ev.tr * i2mu;
                cm_ += lam[7 + j] * (2.0 * zwe * i2mu - C[7 + j] * 2.0 * i2mu);
            }
        }
        pbar[CM_P_LAMBDA] = sbd * ev.tr - cl;
        pbar[CM_P_MU] = 2.0 * sbe - cm_;
        pbar[CM_P_Y] = lam6 * i2mu;
        pbar[CM_P_VOCE_S] = m.has_voce ? lam6 * (1.0 - ev.hd.expo) * i2mu : 0.0;
        pbar[CM_P_VOCE_D] = m.has_voce ? lam6 * m.voce_S * x[6] * ev.hd.expo * i2mu : 0.0;
        pbar[CM_P_LIN_K] = m.has_linear ? lam6 * x[6] * i2mu : 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) pbar[CM_P_YC0 + j] = 0.0;
        if constexpr (YK == CM_YIELD_HILL) {
            // phi = sqrt(sum_j c_j q_j(s));  d phi/d c_j = q_j / (2 phi);
            // d gt/d c_j = (dA/dc_j) s / phi - gt q_j / (2 phi^2)
            if (ev.plastic) {
                const double* s = ev.s;
                const double ip = 1.0 / ev.phi;
                const double d12 = s[3] - s[5], d20 = s[5] - s[0], d01 = s[0] - s[3];
                const double qj[6] = {d12 * d12, d20 * d20, d01 * d01, 2.0 * s[4] * s[4], 2.0 * s[2] * s[2], 2.0 * s[1] * s[1]};
                // u . (dA/dc_j) s
                const double uAs[6] = {(u[3] - u[5]) * d12, (u[5] - u[0]) * d20, (u[0] - u[3]) * d01,
                                       2.0 * u[4] * s[4], 2.0 * u[2] * s[2], 2.0 * u[1] * s[1]};
                const double ug = dot<6>(u, ev.gt);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double ldc = uAs[j] * ip - ug * qj[j] * 0.5 * ip * ip + lam6 * qj[j] * 0.5 * ip * i2mu;
                    pbar[CM_P_YC0 + j] = -ldc;
                }
            }
        }
    }
    if (xpbar) {
        // plastic: dC_k/dv_prev = -delta, dC_k/dalpha_prev = +n_k ; elastic: -I on the first 7
#pragma unroll
        for (int k = 0; k < 6; ++k) xpbar[k] = lam[k];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += ev.gt[k] * kIW[k] * lam[k];
        xpbar[6] = ev.plastic ? -s : lam[6];
        if constexpr (DEF == CM_PLANE_STRESS) xpbar[7] = 0.0;
        if constexpr (DEF == CM_UNIAXIAL_STRESS) { xpbar[7] = 0.0; xpbar[8] = 0.0; }
    }
    if (egbar) {
        // dC_k/deg_l = -dgam/w_k (Ht Cel)_kl ; dC_6/deg_l = (gt Cel)_l / 2mu ; PS row: (Cel (w o z))_l / 2mu
        double t[6], ct[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = hu[k] + lam6 * ev.gt[k] * i2mu;
        if constexpr (DEF == CM_PLANE_STRESS) {
#pragma unroll
            for (int k = 0; k < 6; ++k) t[k] += lam[7] * kW[k] * z[k] * i2mu;
        }
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
#pragma unroll
            for (int k = 0; k < 6; ++k) t[k] += kW[k] * (lam[7] * z[6 + k] + lam[8] * z[12 + k]) * i2mu;
        }
        apply_cel(m, t, ct);
#pragma unroll
        for (int k = 0; k < 6; ++k) egbar[k] = csb[k] - ct[k];
    }
    return ok;
}

// ---- forward tangent at a converged state -----------------------------------------------------------
// T[r][l] = d s_r / d eg_l  (material stress w.r.t. material total strain), IFT rule
// cmad/models/nonlinear_solver.py:158-171:  dx/deg = -A^-1 dC/deg ; ds/deg = Cel (I - dv/deg + z dF33/deg)
template <int DEF, int YK, bool FAST = true>
CM_D bool tangent_point(const cm_model_desc& m, const double eg[6], const double z[6],
                        const double* x, const double* xp, double (&T)[6][6]) {
    constexpr int NX = Dims<DEF>::NX;
    constexpr bool UX = FAST && DEF == CM_UNIAXIAL_STRESS;
    Eval<DEF> ev;
    double C[NX], Ht[6][6], A[UX ? 1 : NX][UX ? 1 : NX];
    residual<DEF, YK, true>(m, eg, z, x, xp, ev, C, Ht);          // the tangent's right-hand sides use Ht
    UniaxialOp uop;
    bool ok;
    if constexpr (UX) { uniaxial_setup<YK>(m, z, ev, Ht, uop, false); ok = uop.ok; }
    else { jacobian_x<DEF, false>(m, z, ev, Ht, A); ok = lu_factor<NX>(A); }
    const double twomu = 2.0 * m.mu, i2mu = half_over_mu(m), lam = m.lambda;
    const double gd = ev.gt[0] + ev.gt[3] + ev.gt[5];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        double b[NX];
        // b = -dC/deg_l
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double hd = Ht[k][0] + Ht[k][3] + Ht[k][5];
            b[k] = ev.plastic ? ev.dgam * kIW[k] * (twomu * Ht[k][l] + (kDiag[l] ? lam * hd : 0.0)) : 0.0;
        }
        b[6] = ev.plastic ? -(ev.gt[l] + (kDiag[l] ? lam * gd * i2mu : 0.0)) : 0.0;
        if constexpr (DEF == CM_PLANE_STRESS) {
            const double zt = z[0] + z[3] + z[5];
            b[7] = -(twomu * kW[l] * z[l] + (kDiag[l] ? lam * zt : 0.0)) * i2mu;
        }
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double* Z = z + 6 * (1 + j);
                b[7 + j] = -(twomu * kW[l] * Z[l] + (kDiag[l] ? lam * (Z[0] + Z[3] + Z[5]) : 0.0)) * i2mu;
            }
        }
        if constexpr (UX) { double bb[NX]; uniaxial_apply(uop, z, b, bb);
#pragma unroll
            for (int k = 0; k < NX; ++k) b[k] = bb[k]; }
        else lu_subst<NX>(A, b);                             // b = dx/deg_l
        double de[6];
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
            // e = eg + (x7 - 1) Z^a + (x8 - 1) Z^b - Pi v
            double t3[3] = {0.0, b[7], b[8]};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double pv = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) pv += kW[k] * z[6 * i + k] * b[k];
                t3[i] -= pv;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) de[k] = ((k == l) ? 1.0 : 0.0) + t3[0] * z[k] + t3[1] * z[6 + k] + t3[2] * z[12 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                de[k] = ((k == l) ? 1.0 : 0.0) - b[k];
                if constexpr (DEF == CM_PLANE_STRESS) de[k] += z[k] * b[7];
            }
        }
        double ds[6];
        apply_cel(m, de, ds);
#pragma unroll
        for (int r = 0; r < 6; ++r) T[r][l] = ds[r];
    }
    return ok;
}


// ---- reverse sweep of the rate-form model at a converged state --------------------------------------------------
// Same contract as reverse_point for small_rate_elastic_plastic.py:249-359.  The unknown x[0:6] is the material stress,
// so (d s / d x)^T sbar_m = [sbar_m ; 0] and the stress has no direct dependence on parameters or strain:
//   lam = A^-T ([sbar_m ; 0] + xin) ,  pbar = -(dC/dp)^T lam ,  xpbar = -(dC/dx_prev)^T lam ,
//   degbar = -(dC/d deg)^T lam   (cotangent of the material strain INCREMENT; grad u gets +, grad u_prev gets -).
// Block formulas: evaluate_blocks_rate below (checked there against the oracle's dual-number Jacobians).
// SOLVE_T: how A^-T is applied -- the dense 7x7 / 8x8 LU here, the structured solver of the total form through the
// change of variables of cm::newton_s_rate (StructRateSolveT, cm_structured.hpp)
struct DenseRateSolveT {
    template <int DEF, int YK>
    CM_D bool apply(const cm_model_desc& m, const double z[6], const double*, const double*, const Eval<DEF>& ev, const double Ht[6][6],
                    double* lam) const {
        constexpr int NX = Dims<DEF>::NX;
        double At[NX][NX], b[NX];
        jacobian_rate<DEF, true>(m, z, ev, Ht, At);
        const bool ok = lu_factor<NX>(At);
#pragma unroll
        for (int k = 0; k < NX; ++k) b[k] = lam[k];
        lu_subst<NX>(At, b);
#pragma unroll
        for (int k = 0; k < NX; ++k) lam[k] = b[k];
        return ok;
    }
};
template <int DEF, int YK, class SOLVE_T = DenseRateSolveT>
CM_D bool reverse_point_rate(const cm_model_desc& m, const double deg[6], const double z[6],
                             const double* x, const double* xp, const double sbm[6], const double* xin,
                             double* pbar, double* xpbar, double* degbar, double* lam_out = nullptr, const SOLVE_T solve_t = SOLVE_T{}) {
    static_assert(DEF != CM_UNIAXIAL_STRESS, "batched rate-form reverse sweep: FULL_3D and PLANE_STRESS");
    constexpr int NX = Dims<DEF>::NX;
    constexpr bool PS = (DEF == CM_PLANE_STRESS);
    Eval<DEF> ev;
    double C[NX], Ht[6][6], lam[NX];
    residual_rate<DEF, YK, true>(m, deg, z, x, xp, ev, C, Ht);
#pragma unroll
    for (int k = 0; k < 6; ++k) lam[k] = sbm[k];
#pragma unroll
    for (int k = 6; k < NX; ++k) lam[k] = 0.0;
    if (xin) {
#pragma unroll
        for (int k = 0; k < NX; ++k) lam[k] += xin[k];
    }
    const bool ok = solve_t.template apply<DEF, YK>(m, z, x, xp, ev, Ht, lam);      // lam <- A^-T lam
    if (lam_out) {
#pragma unroll
        for (int k = 0; k < NX; ++k) lam_out[k] = lam[k];
    }
    const double i2mu = half_over_mu(m);
    const double gd = ev.gt[0] + ev.gt[3] + ev.gt[5];
    const double dgp = ev.plastic ? ev.dgam : 0.0;
    double n[6], cn[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) n[k] = ev.plastic ? ev.gt[k] * kIW[k] : 0.0;
    apply_cel(m, n, cn);
    const double lam6 = ev.plastic ? lam[6] : 0.0;
    const double ld = lam[0] + lam[3] + lam[5];
    double zw[6], czw[6], zt = 0.0, lam7 = 0.0;
    if constexpr (PS) {
        zt = z[0] + z[3] + z[5];
        lam7 = lam[7];
#pragma unroll
        for (int k = 0; k < 6; ++k) zw[k] = kW[k] * z[k];
        apply_cel(m, zw, czw);
    }
    if (pbar) {
        double le = 0.0, ln = 0.0, lc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { le += lam[k] * ev.e[k]; ln += lam[k] * n[k]; lc += lam[k] * C[k]; }
        // lam . dC/dlambda , lam . dC/dmu
        double cl = ld * (-ev.tr + dgp * gd) * i2mu;
        double cm_ = (-2.0 * le + dgp * 2.0 * ln) * i2mu - lc * 2.0 * i2mu + lam6 * (-ev.f * 2.0 * i2mu);
        if constexpr (PS) {
            double zwe = 0.0, zwn = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { zwe += zw[k] * ev.e[k]; zwn += zw[k] * n[k]; }
            cl += lam7 * zt * (ev.tr - dgp * gd) * i2mu;
            cm_ += lam7 * ((2.0 * zwe - dgp * 2.0 * zwn) * i2mu - C[7] * 2.0 * i2mu);
        }
        pbar[CM_P_LAMBDA] = -cl;
        pbar[CM_P_MU] = -cm_;
        pbar[CM_P_Y] = lam6 * i2mu;
        pbar[CM_P_VOCE_S] = m.has_voce ? lam6 * (1.0 - ev.hd.expo) * i2mu : 0.0;
        pbar[CM_P_VOCE_D] = m.has_voce ? lam6 * m.voce_S * x[6] * ev.hd.expo * i2mu : 0.0;
        pbar[CM_P_LIN_K] = m.has_linear ? lam6 * x[6] * i2mu : 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) pbar[CM_P_YC0 + j] = 0.0;
        if constexpr (YK == CM_YIELD_HILL) {
            if (ev.plastic) {
                const double* s = ev.s;
                const double ip = 1.0 / ev.phi;
                const double d12 = s[3] - s[5], d20 = s[5] - s[0], d01 = s[0] - s[3];
                const double qj[6] = {d12 * d12, d20 * d20, d01 * d01, 2.0 * s[4] * s[4], 2.0 * s[2] * s[2], 2.0 * s[1] * s[1]};
                // u = Cel (lam_v - lam7 (w o z)) : lam . dC/dc_j = dgam i2mu u . dn/dc_j + lam6 q_j / (2 phi) i2mu
                double lv[6], u[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) lv[k] = lam[k] - (PS ? lam7 * zw[k] : 0.0);
                apply_cel(m, lv, u);
                // dn_k/dc_j = (dAs_jk / phi - gt_k q_j / (2 phi^2)) / w_k ; u . (dAs_j o 1/w) in closed form:
                const double uw[6] = {u[0], u[1] * 0.5, u[2] * 0.5, u[3], u[4] * 0.5, u[5]};
                const double uAs[6] = {(uw[3] - uw[5]) * d12, (uw[5] - uw[0]) * d20, (uw[0] - uw[3]) * d01,
                                       2.0 * uw[4] * s[4], 2.0 * uw[2] * s[2], 2.0 * uw[1] * s[1]};
                double ugw = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) ugw += uw[k] * ev.gt[k];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double udn = uAs[j] * ip - ugw * qj[j] * 0.5 * ip * ip;
                    pbar[CM_P_YC0 + j] = -(ev.dgam * udn * i2mu + lam6 * qj[j] * 0.5 * ip * i2mu);
                }
            }
        }
    }
    if (xpbar) {
        double lcn = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { xpbar[k] = lam[k] * i2mu; lcn += lam[k] * cn[k]; }
        double zcn = 0.0;
        if constexpr (PS) {
#pragma unroll
            for (int k = 0; k < 6; ++k) zcn += zw[k] * cn[k];
        }
        xpbar[6] = ev.plastic ? (lcn * i2mu - (PS ? lam7 * zcn * i2mu : 0.0)) : lam[6];
        if constexpr (PS) {
            double cz[6], lcz = 0.0, zcz = 0.0;
            apply_cel(m, z, cz);
#pragma unroll
            for (int k = 0; k < 6; ++k) { lcz += lam[k] * cz[k]; zcz += zw[k] * cz[k]; }
            xpbar[7] = -(lcz * i2mu - lam7 * zcz * i2mu);
        }
    }
    if (degbar) {
        double lv[6], cl[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) lv[k] = lam[k] - (PS ? lam7 * zw[k] : 0.0);
        apply_cel(m, lv, cl);
#pragma unroll
        for (int k = 0; k < 6; ++k) degbar[k] = cl[k] * i2mu;
    }
    return ok;
}

// ---- forward tangent of the rate-form model at a converged state ------------------------------------------------
// T[r][l] = d s_r / d (d eg)_l : the unknown x[0:6] IS the material stress, so T is the first six rows of
// dx / d deg = -A^-1 dC/d deg with dC_k/d deg_l = -Cel_kl / 2mu (both branches) and, under PLANE_STRESS,
// dC_7/d deg_l = (Cel (w o z))_l / 2mu  (small_rate_elastic_plastic.py:249-346; IFT rule nonlinear_solver.py:158-171).
// SOLVE: factor A once, apply A^-1 to several right-hand sides -- the dense LU here, the structured solver of the total form
// through the change of variables of cm::newton_s_rate (StructRateSolve, cm_structured.hpp)
template <int NXMAX>
struct DenseRateSolve {
    double A[NXMAX][NXMAX];
    template <int DEF, int YK>
    CM_D bool setup(const cm_model_desc& m, const double deg[6], const double z[6], const double* x, const double* xp) {
        static_assert(Dims<DEF>::NX == NXMAX, "one size per instantiation");
        Eval<DEF> ev;
        double C[NXMAX], Ht[6][6];
        residual_rate<DEF, YK, true>(m, deg, z, x, xp, ev, C, Ht);
        jacobian_rate<DEF, false>(m, z, ev, Ht, A);
        return lu_factor<NXMAX>(A);
    }
    template <int DEF, int YK>
    CM_D void solve(const cm_model_desc&, const double*, double (&b)[NXMAX]) const { lu_subst<NXMAX>(A, b); }
};
template <int DEF, int YK, class SOLVE = DenseRateSolve<Dims<DEF>::NX>>
CM_D bool tangent_point_rate(const cm_model_desc& m, const double deg[6], const double z[6],
                             const double* x, const double* xp, double (&T)[6][6]) {
    static_assert(DEF != CM_UNIAXIAL_STRESS, "batched rate-form tangent: FULL_3D and PLANE_STRESS");
    constexpr int NX = Dims<DEF>::NX;
    SOLVE solver;
    const bool ok = solver.template setup<DEF, YK>(m, deg, z, x, xp);
    const double i2mu = half_over_mu(m);
    double czw[6];
    if constexpr (DEF == CM_PLANE_STRESS) {
        double zw[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) zw[k] = kW[k] * z[k];
        apply_cel(m, zw, czw);
    }
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        double unit[6], col[6], b[NX];
#pragma unroll
        for (int k = 0; k < 6; ++k) unit[k] = (k == l) ? 1.0 : 0.0;
        apply_cel(m, unit, col);                                 // Cel e_l
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = col[k] * i2mu;        // -dC_k / d deg_l
        b[6] = 0.0;
        if constexpr (DEF == CM_PLANE_STRESS) b[7] = -czw[l] * i2mu;
        solver.template solve<DEF, YK>(m, z, b);                 // b <- A^-1 b
#pragma unroll
        for (int r = 0; r < 6; ++r) T[r][l] = b[r];
    }
    return ok;
}

// ---- explicit derivative blocks at an arbitrary state (the reference's stateful evaluate() surface) ----
// cmad/models/model.py:168-190 (Jac for DXI / DXI_PREV / DPARAMS / DU) and :273-293 (dSigma).
// Column counts: DXI, DXI_PREV -> NX ; DPARAMS -> CM_NUM_PARAMS (KP order) ; DU -> NU.
// J is row-major [NX][ncols], S is [6][ncols] (global stress 6-vector rows).
enum { CM_W_XI = 0, CM_W_XI_PREV = 1, CM_W_PARAMS = 2, CM_W_U = 3, CM_W_NONE = 5 };

template <int DEF, int YK, bool ROT>
CM_D void evaluate_blocks(const cm_model_desc& m, const double* G, const double* x, const double* xp, int which,
                          double* C, double* J /* NX x ncols or null */, double* sg /* 6 */, double* S /* 6 x ncols or null */) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    double eg[6], z[Dims<DEF>::NZ], Ht[6][6];
    Eval<DEF> ev;
    strain_from_gradu<DEF, ROT>(m, G, eg);
    strain_z<DEF, ROT>(m, z);
    residual<DEF, YK, true>(m, eg, z, x, xp, ev, C, Ht);
    to_global<ROT>(m, ev.s, sg);
    const double i2mu = half_over_mu(m);
    const int ncols = (which == CM_W_XI || which == CM_W_XI_PREV) ? NX : (which == CM_W_PARAMS ? CM_NUM_PARAMS : NU);
    if (which == CM_W_NONE) return;
    // material-frame stress derivative columns ds[6] per column, then rotated
    auto put_S = [&](int c, const double ds[6]) {
        if (!S) return;
        double dg[6];
        to_global<ROT>(m, ds, dg);
        for (int r = 0; r < 6; ++r) S[r * ncols + c] = dg[r];
    };
    const double zero6[6] = {0, 0, 0, 0, 0, 0};
    if (which == CM_W_XI) {
        double A[NX][NX];
        jacobian_x<DEF, false>(m, z, ev, Ht, A);
        if (J) for (int r = 0; r < NX; ++r) for (int c = 0; c < NX; ++c) J[r * NX + c] = A[r][c];
        for (int c = 0; c < 6; ++c) {
            double de[6] = {0, 0, 0, 0, 0, 0}, ds[6];
            if constexpr (DEF == CM_UNIAXIAL_STRESS) {
                for (int k = 0; k < 6; ++k) de[k] = -kW[c] * (z[k] * z[c] + z[6 + k] * z[6 + c] + z[12 + k] * z[12 + c]);   // -Pi[:, c]
            } else de[c] = -1.0;
            apply_cel(m, de, ds);
            put_S(c, ds);
        }
        put_S(6, zero6);
        if constexpr (DEF == CM_PLANE_STRESS) { double ds[6]; apply_cel(m, z, ds); put_S(7, ds); }
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
            double ds[6];
            apply_cel(m, z + 6, ds); put_S(7, ds);
            apply_cel(m, z + 12, ds); put_S(8, ds);
        }
    } else if (which == CM_W_XI_PREV) {
        if (J) {
            for (int r = 0; r < NX; ++r) for (int c = 0; c < NX; ++c) J[r * NX + c] = 0.0;
            for (int k = 0; k < 6; ++k) J[k * NX + k] = -1.0;
            if (ev.plastic) { for (int k = 0; k < 6; ++k) J[k * NX + 6] = ev.gt[k] * kIW[k]; }
            else J[6 * NX + 6] = -1.0;
        }
        for (int c = 0; c < NX; ++c) put_S(c, zero6);
    } else if (which == CM_W_PARAMS) {
        constexpr int NP_ = CM_NUM_PARAMS;
        if (J) for (int i = 0; i < NX * NP_; ++i) J[i] = 0.0;
        double dvec[6], e2[6], Hd[6], He[6];
        for (int k = 0; k < 6; ++k) { dvec[k] = kDiag[k] ? 1.0 : 0.0; e2[k] = 2.0 * ev.e[k]; }
        for (int k = 0; k < 6; ++k) {
            double a = 0, b = 0;
            for (int l = 0; l < 6; ++l) { a += Ht[k][l] * dvec[l]; b += Ht[k][l] * e2[l]; }
            Hd[k] = a; He[k] = b;
        }
        if (J && ev.plastic) {
            const double gd = ev.gt[0] + ev.gt[3] + ev.gt[5], ge2 = dot<6>(ev.gt, e2);
            for (int k = 0; k < 6; ++k) {
                J[k * NP_ + CM_P_LAMBDA] = -ev.dgam * kIW[k] * Hd[k] * ev.tr;
                J[k * NP_ + CM_P_MU] = -ev.dgam * kIW[k] * He[k];
            }
            J[6 * NP_ + CM_P_LAMBDA] = gd * ev.tr * i2mu;
            J[6 * NP_ + CM_P_MU] = ge2 * i2mu - ev.f * 2.0 * i2mu;
            J[6 * NP_ + CM_P_Y] = -i2mu;
            if (m.has_voce) {
                J[6 * NP_ + CM_P_VOCE_S] = -(1.0 - ev.hd.expo) * i2mu;
                J[6 * NP_ + CM_P_VOCE_D] = -m.voce_S * x[6] * ev.hd.expo * i2mu;
            }
            if (m.has_linear) J[6 * NP_ + CM_P_LIN_K] = -x[6] * i2mu;
            if constexpr (YK == CM_YIELD_HILL) {
                const double* s = ev.s;
                const double ip = 1.0 / ev.phi;
                const double d12 = s[3] - s[5], d20 = s[5] - s[0], d01 = s[0] - s[3];
                const double qj[6] = {d12 * d12, d20 * d20, d01 * d01, 2.0 * s[4] * s[4], 2.0 * s[2] * s[2], 2.0 * s[1] * s[1]};
                double dAs[6][6] = {{0}};     // [j][k] = (dA/dc_j s)_k
                dAs[0][3] = d12; dAs[0][5] = -d12;
                dAs[1][5] = d20; dAs[1][0] = -d20;
                dAs[2][0] = d01; dAs[2][3] = -d01;
                dAs[3][4] = 2.0 * s[4]; dAs[4][2] = 2.0 * s[2]; dAs[5][1] = 2.0 * s[1];
                for (int j = 0; j < 6; ++j) {
                    for (int k = 0; k < 6; ++k) {
                        const double dg = dAs[j][k] * ip - ev.gt[k] * qj[j] * 0.5 * ip * ip;
                        J[k * NP_ + CM_P_YC0 + j] = -ev.dgam * kIW[k] * dg;
                    }
                    J[6 * NP_ + CM_P_YC0 + j] = qj[j] * 0.5 * ip * i2mu;
                }
            }
        }
        if constexpr (DEF == CM_PLANE_STRESS) {
            if (J) {
                const double zt = z[0] + z[3] + z[5];
                double zwe = 0.0;
                for (int k = 0; k < 6; ++k) zwe += kW[k] * z[k] * ev.e[k];
                J[7 * NP_ + CM_P_LAMBDA] = zt * ev.tr * i2mu;
                J[7 * NP_ + CM_P_MU] = 2.0 * zwe * i2mu - C[7] * 2.0 * i2mu;
            }
        }
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
            if (J) {
                for (int j = 0; j < 2; ++j) {
                    const double* Z = z + 6 * (1 + j);
                    const double zt = Z[0] + Z[3] + Z[5];
                    double zwe = 0.0;
                    for (int k = 0; k < 6; ++k) zwe += kW[k] * Z[k] * ev.e[k];
                    J[(7 + j) * NP_ + CM_P_LAMBDA] = zt * ev.tr * i2mu;
                    J[(7 + j) * NP_ + CM_P_MU] = 2.0 * zwe * i2mu - C[7 + j] * 2.0 * i2mu;
                }
            }
        }
        for (int c = 0; c < NP_; ++c) {
            double ds[6] = {0, 0, 0, 0, 0, 0};
            if (c == CM_P_LAMBDA) for (int k = 0; k < 6; ++k) ds[k] = dvec[k] * ev.tr;
            if (c == CM_P_MU) for (int k = 0; k < 6; ++k) ds[k] = e2[k];
            put_S(c, ds);
        }
    } else if (which == CM_W_U) {
        for (int c = 0; c < NU; ++c) {
            double Gd[NU], dm[6], cd[6];
            for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
            strain_from_gradu<DEF, ROT>(m, Gd, dm);        // d eg / d G_c
            apply_cel(m, dm, cd);                          // Cel d eg
            put_S(c, cd);
            if (J) {
                for (int k = 0; k < 6; ++k) {
                    double hc = 0.0;
                    for (int l = 0; l < 6; ++l) hc += Ht[k][l] * cd[l];
                    J[k * NU + c] = ev.plastic ? -ev.dgam * kIW[k] * hc : 0.0;
                }
                J[6 * NU + c] = ev.plastic ? dot<6>(ev.gt, cd) * i2mu : 0.0;
                if constexpr (DEF == CM_PLANE_STRESS) {
                    double r = 0.0;
                    for (int k = 0; k < 6; ++k) r += kW[k] * z[k] * cd[k];
                    J[7 * NU + c] = r * i2mu;
                }
                if constexpr (DEF == CM_UNIAXIAL_STRESS) {
                    double ra = 0.0, rb = 0.0;
                    for (int k = 0; k < 6; ++k) { ra += kW[k] * z[6 + k] * cd[k]; rb += kW[k] * z[12 + k] * cd[k]; }
                    J[7 * NU + c] = ra * i2mu; J[8 * NU + c] = rb * i2mu;
                }
            }
        }
    }
}


// ---- explicit derivative blocks of the rate-form model at an arbitrary state ----------------------------
// Same contract as evaluate_blocks (model.py:168-190, :273-293) for small_rate_elastic_plastic.py:249-359;
// `which` may also be CM_W_U_PREV = 4 (d/d grad u_prev = - d/d grad u).  sigma_global = Q x[0:6] Q^T.
enum { CM_W_U_PREV = 4 };

template <int DEF, int YK, bool ROT>
CM_D void evaluate_blocks_rate(const cm_model_desc& m, const double* G, const double* Gp, const double* x, const double* xp,
                               int which, double* C, double* J, double* sg, double* S) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU, NP_ = CM_NUM_PARAMS;
    double deg[6], z[Dims<DEF>::NZ], Ht[6][6], Gd[NU];
    Eval<DEF> ev;
    for (int k = 0; k < NU; ++k) Gd[k] = G[k] - Gp[k];
    strain_from_gradu<DEF, ROT>(m, Gd, deg);
    strain_z<DEF, ROT>(m, z);
    residual_rate<DEF, YK, true>(m, deg, z, x, xp, ev, C, Ht);
    to_global<ROT>(m, x, sg);
    if (which == CM_W_NONE) return;
    const double i2mu = half_over_mu(m);
    const int ncols = (which == CM_W_XI || which == CM_W_XI_PREV) ? NX : (which == CM_W_PARAMS ? NP_ : NU);
    if (S) for (int i = 0; i < 6 * ncols; ++i) S[i] = 0.0;
    if (J) for (int i = 0; i < NX * ncols; ++i) J[i] = 0.0;
    const double gd = ev.gt[0] + ev.gt[3] + ev.gt[5];
    double n[6], cn[6];
    for (int k = 0; k < 6; ++k) n[k] = ev.gt[k] * kIW[k];
    apply_cel(m, n, cn);
    if (which == CM_W_XI) {
        double A[NX][NX];
        jacobian_rate<DEF, false>(m, z, ev, Ht, A);
        if (J) for (int r = 0; r < NX; ++r) for (int c = 0; c < NX; ++c) J[r * NX + c] = A[r][c];
        if (S) for (int c = 0; c < 6; ++c) {
            double un[6] = {0, 0, 0, 0, 0, 0}, dg[6];
            un[c] = 1.0;
            to_global<ROT>(m, un, dg);
            for (int r = 0; r < 6; ++r) S[r * ncols + c] = dg[r];
        }
    } else if (which == CM_W_XI_PREV) {
        if (J) {
            for (int k = 0; k < 6; ++k) {
                J[k * NX + k] = -i2mu;
                if (ev.plastic) J[k * NX + 6] = -cn[k] * i2mu;
            }
            if (!ev.plastic) J[6 * NX + 6] = -1.0;
            if constexpr (DEF == CM_PLANE_STRESS) {
                double cz[6], zcz = 0.0, zcn = 0.0;
                apply_cel(m, z, cz);
                for (int k = 0; k < 6; ++k) { J[k * NX + 7] = cz[k] * i2mu; zcz += kW[k] * z[k] * cz[k]; zcn += kW[k] * z[k] * cn[k]; }
                J[7 * NX + 7] = -zcz * i2mu;
                if (ev.plastic) J[7 * NX + 6] = zcn * i2mu;
            }
        }
    } else if (which == CM_W_PARAMS) {
        if (J) {
            const double dgp = ev.plastic ? ev.dgam : 0.0;
            const double trn = ev.plastic ? gd : 0.0;
            for (int k = 0; k < 6; ++k) {
                const double nk = ev.plastic ? n[k] : 0.0;
                J[k * NP_ + CM_P_LAMBDA] = (kDiag[k] ? (-ev.tr + dgp * trn) : 0.0) * i2mu;
                J[k * NP_ + CM_P_MU] = (-2.0 * ev.e[k] + dgp * 2.0 * nk) * i2mu - C[k] * 2.0 * i2mu;
            }
            if (ev.plastic) {
                J[6 * NP_ + CM_P_MU] = -ev.f * 2.0 * i2mu;
                J[6 * NP_ + CM_P_Y] = -i2mu;
                if (m.has_voce) {
                    J[6 * NP_ + CM_P_VOCE_S] = -(1.0 - ev.hd.expo) * i2mu;
                    J[6 * NP_ + CM_P_VOCE_D] = -m.voce_S * x[6] * ev.hd.expo * i2mu;
                }
                if (m.has_linear) J[6 * NP_ + CM_P_LIN_K] = -x[6] * i2mu;
            }
            double dndc[6][6] = {{0}};        // [j][k] = d n_k / d c_j (Hill)
            if constexpr (YK == CM_YIELD_HILL) {
                if (ev.plastic) {
                    const double* s = ev.s;
                    const double ip = 1.0 / ev.phi;
                    const double d12 = s[3] - s[5], d20 = s[5] - s[0], d01 = s[0] - s[3];
                    const double qj[6] = {d12 * d12, d20 * d20, d01 * d01, 2.0 * s[4] * s[4], 2.0 * s[2] * s[2], 2.0 * s[1] * s[1]};
                    double dAs[6][6] = {{0}};
                    dAs[0][3] = d12; dAs[0][5] = -d12; dAs[1][5] = d20; dAs[1][0] = -d20; dAs[2][0] = d01; dAs[2][3] = -d01;
                    dAs[3][4] = 2.0 * s[4]; dAs[4][2] = 2.0 * s[2]; dAs[5][1] = 2.0 * s[1];
                    for (int j = 0; j < 6; ++j) {
                        double cdn[6];
                        for (int k = 0; k < 6; ++k) dndc[j][k] = (dAs[j][k] * ip - ev.gt[k] * qj[j] * 0.5 * ip * ip) * kIW[k];
                        apply_cel(m, dndc[j], cdn);
                        for (int k = 0; k < 6; ++k) J[k * NP_ + CM_P_YC0 + j] = ev.dgam * cdn[k] * i2mu;
                        J[6 * NP_ + CM_P_YC0 + j] = qj[j] * 0.5 * ip * i2mu;
                    }
                }
            }
            if constexpr (DEF == CM_PLANE_STRESS) {
                const double zt = z[0] + z[3] + z[5];
                double zwe = 0.0, zwn = 0.0;
                for (int k = 0; k < 6; ++k) { zwe += kW[k] * z[k] * ev.e[k]; zwn += kW[k] * z[k] * n[k]; }
                J[7 * NP_ + CM_P_LAMBDA] = zt * (ev.tr - dgp * trn) * i2mu;
                J[7 * NP_ + CM_P_MU] = (2.0 * zwe - dgp * 2.0 * zwn) * i2mu - C[7] * 2.0 * i2mu;
                if constexpr (YK == CM_YIELD_HILL) {
                    if (ev.plastic) for (int j = 0; j < 6; ++j) {
                        double cdn[6], r = 0.0;
                        apply_cel(m, dndc[j], cdn);
                        for (int k = 0; k < 6; ++k) r += kW[k] * z[k] * cdn[k];
                        J[7 * NP_ + CM_P_YC0 + j] = -ev.dgam * r * i2mu;
                    }
                }
            }
        }
    } else if (which == CM_W_U || which == CM_W_U_PREV) {
        const double sgn = (which == CM_W_U) ? 1.0 : -1.0;
        if (J) for (int c = 0; c < NU; ++c) {
            double Gu[NU], dm[6], cd[6];
            for (int k = 0; k < NU; ++k) Gu[k] = (k == c) ? 1.0 : 0.0;
            strain_from_gradu<DEF, ROT>(m, Gu, dm);
            apply_cel(m, dm, cd);
            for (int k = 0; k < 6; ++k) J[k * NU + c] = -sgn * cd[k] * i2mu;
            if constexpr (DEF == CM_PLANE_STRESS) {
                double r = 0.0;
                for (int k = 0; k < 6; ++k) r += kW[k] * z[k] * cd[k];
                J[7 * NU + c] = sgn * r * i2mu;
            }
        }
    }
}

// ---- forward (direct) parameter sensitivities of one converged step -------------------------------------------------
// cmad/objectives/mp_objective.py:158-215 (MPDirectObjective): with A = dC/dxi at the converged state,
//   dxi/dp = -A^-1 (dC/dp + dC/dxi_prev dxi_prev/dp) ,  dsigma/dp = dsigma/dp|_xi + dsigma/dxi dxi/dp
// for the CM_NUM_PARAMS native parameters (KP order).  Built on the explicit blocks of evaluate_blocks[_rate] (the same
// ones the stateful evaluate() surface returns), one LU factorisation and CM_NUM_PARAMS substitutions per point.
//   dxp_dp [NX][NP] row-major or null (= 0: first step of a history) -> dx_dp [NX][NP], ds_dp [6][NP] (global stress; may be null)
template <int MK, int DEF, int YK, bool ROT>
CM_D bool direct_point(const cm_model_desc& m, const double* G, const double* Gp, const double* x, const double* xp,
                       const double* dxp_dp, double* dx_dp, double* ds_dp) {
    constexpr int NX = Dims<DEF>::NX, NP_ = CM_NUM_PARAMS;
    double C[NX], sg[6], Ax[NX * NX], Sx[6 * NX], Cp[NX * NP_], Sp[6 * NP_], Axp[NX * NX];
    auto blocks = [&](int which, double* J, double* S) {
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) evaluate_blocks_rate<DEF, YK, ROT>(m, G, Gp, x, xp, which, C, J, sg, S);
        else evaluate_blocks<DEF, YK, ROT>(m, G, x, xp, which, C, J, sg, S);
    };
    blocks(CM_W_XI, Ax, Sx);
    blocks(CM_W_PARAMS, Cp, Sp);
    double A[NX][NX];
    for (int i = 0; i < NX; ++i) for (int j = 0; j < NX; ++j) A[i][j] = Ax[i * NX + j];
    const bool ok = lu_factor<NX>(A);
    if (dxp_dp) blocks(CM_W_XI_PREV, Axp, nullptr);
    for (int j = 0; j < NP_; ++j) {
        double rhs[NX];
        for (int i = 0; i < NX; ++i) {
            double t = Cp[i * NP_ + j];
            if (dxp_dp) for (int k = 0; k < NX; ++k) t += Axp[i * NX + k] * dxp_dp[k * NP_ + j];
            rhs[i] = -t;
        }
        lu_subst<NX>(A, rhs);
        for (int i = 0; i < NX; ++i) dx_dp[i * NP_ + j] = rhs[i];
        if (ds_dp) for (int r = 0; r < 6; ++r) {
            double t = Sp[r * NP_ + j];
            for (int k = 0; k < NX; ++k) t += Sx[r * NX + k] * rhs[k];
            ds_dp[r * NP_ + j] = t;
        }
    }
    return ok;
}

// ---- one COLUMN of the forward sensitivities of a converged step ------------------------------------------------------
// The same recursion as direct_point for ONE native parameter j (KP order):  d = dxi/dp_j = -A^-1 (dC/dp_j + dC/dxi_prev d_prev),
// ds = dsigma/dp_j|_xi + dsigma/dxi d.  One thread per (point, parameter): twelve lanes share a point instead of one lane
// carrying a 7..9 x 12 block through runtime-indexed (scratch) arrays -- every array here is indexed by compile-time
// constants after unrolling (the column of the parameter block is picked by selects), and the twelve columns of a point
// advance in parallel, which is what the one-point-at-a-time material-point objectives need (cmad/objectives/
// mp_objective.py:158-215 with B = 1).  d_prev: the carried column (null = 0 on the first step).
template <int MK, int DEF, int YK, bool ROT>
CM_D bool direct_column(const cm_model_desc& m, const double* G, const double* Gp, const double* x, const double* xp, int j,
                        const double* d_prev, double* d, double* ds) {
    constexpr int NX = Dims<DEF>::NX, NP_ = CM_NUM_PARAMS;
    double C[NX], sg[6], rhs[NX], sp[6];
    {   // dC/dp_j and dsigma/dp_j: the parameter block, column j picked by selects (no runtime-indexed array)
        double Cp[NX * NP_], Sp[6 * NP_];
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) evaluate_blocks_rate<DEF, YK, ROT>(m, G, Gp, x, xp, CM_W_PARAMS, C, Cp, sg, Sp);
        else evaluate_blocks<DEF, YK, ROT>(m, G, x, xp, CM_W_PARAMS, C, Cp, sg, Sp);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double v = 0.0;
#pragma unroll
            for (int jj = 0; jj < NP_; ++jj) v = (jj == j) ? Cp[i * NP_ + jj] : v;
            rhs[i] = -v;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double v = 0.0;
#pragma unroll
            for (int jj = 0; jj < NP_; ++jj) v = (jj == j) ? Sp[r * NP_ + jj] : v;
            sp[r] = v;
        }
    }
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
    double Ax[NX * NX], Sx[6 * NX], A[NX][NX];
    if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) evaluate_blocks_rate<DEF, YK, ROT>(m, G, Gp, x, xp, CM_W_XI, C, Ax, sg, Sx);
    else evaluate_blocks<DEF, YK, ROT>(m, G, x, xp, CM_W_XI, C, Ax, sg, Sx);
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int k = 0; k < NX; ++k) A[i][k] = Ax[i * NX + k];
    const bool ok = lu_factor<NX>(A);
    lu_subst<NX>(A, rhs);
#pragma unroll
    for (int i = 0; i < NX; ++i) d[i] = rhs[i];
    if (ds) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double t = sp[r];
#pragma unroll
            for (int k = 0; k < NX; ++k) t += Sx[r * NX + k] * rhs[k];
            ds[r] = t;
        }
    }
    return ok;
}

}  // namespace cm
