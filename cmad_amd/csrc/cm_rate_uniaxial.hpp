// The rate-form model under UNIAXIAL_STRESS (cmad/models/small_rate_elastic_plastic.py:34-75, 171-196, 249-346): 12 local
// dofs -- material Cauchy stress (6), alpha, the two off-axis stretches, the three off-axis shear-strain increments -- and one
// grad-u entry.  The reference uses it for one material point at a time (uniaxial calibration data); nothing here is hot, so
// instead of hand-deriving a 12 x 12 Jacobian every derivative block comes from forward-mode evaluation of the arithmetic-T
// residual (cm_hessian.hpp: residual_rate_uniaxial_T with the first-order dual D1), one direction per evaluation -- exactly
// what jacfwd does in the reference (cmad/models/model.py:125-131).  On top of the blocks: the local Newton (same control flow,
// tolerances and line search as cm::newton), the implicit-function tangent, the reverse sweep and the forward sensitivities,
// with the contracts of their FULL_3D / PLANE_STRESS counterparts in cm_device.hpp, so that every batched entry point of the
// rate form serves this deformation type as well.
#pragma once
#include "cm_hessian.hpp"

namespace cm {

constexpr int kRuNX = 12;

// values: residual C[12] and global Cauchy stress sg[6] at (x, xp); dU = grad u - grad u_prev (one entry)
template <int YK>
CM_D void ru_eval(const cm_model_desc& m, double dU, const double* x, const double* xp, double* C, double* sg) {
    MatT<double> p;
    mat_from_desc<double>(m, p);
    double s[6];
    residual_rate_uniaxial_T<YK, double>(m, p, dU, x, xp, C, s);
    congruence_T<false, double>(p.Q, s, sg);
}

// one derivative block by forward-mode evaluation: J[12][ncols], S[6][ncols] (row-major, either may be null)
//   CM_W_XI / CM_W_XI_PREV: 12 columns; CM_W_PARAMS: CM_NUM_PARAMS columns (KP order); CM_W_U: 1 column (CM_W_U_PREV = its negative)
template <int YK>
CM_D void ru_block(const cm_model_desc& m, double dU, const double* xv, const double* xpv, int which, double* J, double* S) {
    const int ncols = (which == CM_W_XI || which == CM_W_XI_PREV) ? kRuNX : (which == CM_W_PARAMS ? CM_NUM_PARAMS : 1);
    const double sgn = (which == CM_W_U_PREV) ? -1.0 : 1.0;
    for (int c = 0; c < ncols; ++c) {
        D1 x[kRuNX], xp[kRuNX], C[kRuNX], s[6], sg[6];
        MatT<D1> p;
        mat_from_desc<D1>(m, p);
        for (int k = 0; k < kRuNX; ++k) { x[k] = d1(xv[k]); xp[k] = d1(xpv[k]); }
        D1 du = d1(dU);
        if (which == CM_W_XI) x[c].d = 1.0;
        else if (which == CM_W_XI_PREV) xp[c].d = 1.0;
        else if (which == CM_W_PARAMS) mat_seed<D1>(p, c);
        else du.d = 1.0;
        residual_rate_uniaxial_T<YK, D1>(m, p, du, x, xp, C, s);
        congruence_T<false, D1>(p.Q, s, sg);
        if (J) for (int r = 0; r < kRuNX; ++r) J[r * ncols + c] = sgn * C[r].d;
        if (S) for (int r = 0; r < 6; ++r) S[r * ncols + c] = sgn * sg[r].d;
    }
}

// dense solve with partial pivoting (the 12 x 12 system mixes stress-, strain- and stretch-like unknowns): A (n x n,
// row-major, destroyed), B (n x nrhs, row-major) -> solution in B.  Returns false on a vanishing pivot.
CM_D bool ru_solve(int n, double* A, double* B, int nrhs) {
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(A[k * n + k]);
        for (int r = k + 1; r < n; ++r) if (fabs(A[r * n + k]) > best) { best = fabs(A[r * n + k]); piv = r; }
        if (!(best > 1e-300)) { ok = false; continue; }
        if (piv != k) {
            for (int c = 0; c < n; ++c) { const double t = A[k * n + c]; A[k * n + c] = A[piv * n + c]; A[piv * n + c] = t; }
            for (int c = 0; c < nrhs; ++c) { const double t = B[k * nrhs + c]; B[k * nrhs + c] = B[piv * nrhs + c]; B[piv * nrhs + c] = t; }
        }
        const double ip = 1.0 / A[k * n + k];
        for (int r = k + 1; r < n; ++r) {
            const double l = A[r * n + k] * ip;
            if (l != 0.0) {
                for (int c = k + 1; c < n; ++c) A[r * n + c] -= l * A[k * n + c];
                for (int c = 0; c < nrhs; ++c) B[r * nrhs + c] -= l * B[k * nrhs + c];
            }
        }
    }
    for (int k = n - 1; k >= 0; --k)
        for (int c = 0; c < nrhs; ++c) {
            double sacc = B[k * nrhs + c];
            for (int j = k + 1; j < n; ++j) sacc -= A[k * n + j] * B[j * nrhs + c];
            B[k * nrhs + c] = sacc / A[k * n + k];
        }
    return ok;
}

// local Newton: cm::newton's control flow (make_newton_solve with the quadratic Armijo search, or plain Newton) on the 12 dofs
template <int YK, bool LS>
CM_D uint32_t ru_newton(const cm_model_desc& m, double dU, const double* xp, double* x, bool lane_valid) {
    constexpr int NX = kRuNX;
    double C[NX], sg[6];
    for (int k = 0; k < NX; ++k) x[k] = xp[k];
    ru_eval<YK>(m, dU, x, xp, C, sg);
    const double n0sq = dot<NX>(C, C);
    const double rel2 = m.rel_tol * m.rel_tol * n0sq, abs2 = m.abs_tol * m.abs_tol;
    int it = 0;
    bool running = lane_valid;
    uint32_t flags = 0;
    for (;;) {
        const double nsq = dot<NX>(C, C);
        const bool conv = (nsq < rel2) || (nsq < abs2);
        if (running && conv) { running = false; flags |= CM_STATUS_CONVERGED; }
        if (running && it >= m.max_iters) running = false;
        if (!__any(running)) break;
        if (running) {
            double A[NX * NX], delta[NX];
            ru_block<YK>(m, dU, x, xp, CM_W_XI, A, nullptr);
            for (int k = 0; k < NX; ++k) delta[k] = C[k];
            if (!ru_solve(NX, A, delta, 1)) flags |= CM_STATUS_SINGULAR;
            bool plain = !LS;                                          // LS kernels serve plain Newton too (uniform)
            if constexpr (LS) plain = (m.ls_max_evals <= 0);
            if (plain) {
                for (int k = 0; k < NX; ++k) x[k] -= delta[k];
                ru_eval<YK>(m, dU, x, xp, C, sg);
            } else if constexpr (LS) {
                const double cc = dot<NX>(C, C);
                const double phi0 = 0.5 * cc, dphi0 = -cc, armijo = m.ls_c1 * dphi0;
                int n = 0;
                double alpha = 1.0, best_alpha = 1.0, best_phi = INFINITY;
                bool accepted = false;
                double Cbest[NX], Ct[NX], xt[NX];
                for (int k = 0; k < NX; ++k) { Cbest[k] = C[k]; Ct[k] = C[k]; }
                while (n < m.ls_max_evals && !accepted) {
                    for (int k = 0; k < NX; ++k) xt[k] = x[k] - alpha * delta[k];
                    ru_eval<YK>(m, dU, xt, xp, Ct, sg);
                    const double phi = 0.5 * dot<NX>(Ct, Ct);
                    if (m.ls_kind == CM_LS_LEGACY) {                    // newton_solve's backtracking: the last evaluated trial is kept
                        const double a_eval = alpha;
                        if (ls_trial_legacy(m, phi, cc, alpha, n) == 0.0) { alpha = a_eval; accepted = true; }
                        continue;
                    }
                    const bool finite = isfinite(phi);
                    if (finite && phi < best_phi) { best_alpha = alpha; best_phi = phi; for (int k = 0; k < NX; ++k) Cbest[k] = Ct[k]; }
                    accepted = finite && (phi <= phi0 + alpha * armijo);
                    const double am = quad_min(phi0, dphi0, alpha, phi);
                    const double ac = fmin(fmax(am, m.ls_lo * alpha), m.ls_hi * alpha);
                    if (!accepted) alpha = finite ? ac : 0.5 * alpha;
                    ++n;
                }
                const double ra = accepted ? alpha : best_alpha;
                for (int k = 0; k < NX; ++k) { x[k] -= ra * delta[k]; C[k] = accepted ? Ct[k] : Cbest[k]; }
            }
            ++it;
        }
    }
    return flags | (uint32_t)it;
}

// d sigma_global / d (dU) at a converged state (6 entries): IFT rule nonlinear_solver.py:158-171
template <int YK>
CM_D bool ru_tangent(const cm_model_desc& m, double dU, const double* x, const double* xp, double* dsig /* 6 */) {
    constexpr int NX = kRuNX;
    double A[NX * NX], Sx[6 * NX], Cu[NX], Su[6];
    ru_block<YK>(m, dU, x, xp, CM_W_XI, A, Sx);
    ru_block<YK>(m, dU, x, xp, CM_W_U, Cu, Su);
    for (int k = 0; k < NX; ++k) Cu[k] = -Cu[k];
    const bool ok = ru_solve(NX, A, Cu, 1);                        // dx / d dU
    for (int r = 0; r < 6; ++r) {
        double t = Su[r];
        for (int k = 0; k < NX; ++k) t += Sx[r * NX + k] * Cu[k];
        dsig[r] = t;
    }
    return ok;
}

// reverse sweep (contract of cm::reverse_point_rate, with the cotangent sb of the GLOBAL stress entries):
//   lam = A^-T (Sx^T sb + xin), pbar = Sp^T sb - Cp^T lam, xpbar = -Cxp^T lam, ubar = Su . sb - Cu . lam (cotangent of dU)
template <int YK>
CM_D bool ru_reverse(const cm_model_desc& m, double dU, const double* x, const double* xp, const double sb[6], const double* xin,
                     double* pbar, double* xpbar, double* ubar, double* lam_out = nullptr) {
    constexpr int NX = kRuNX, NP_ = CM_NUM_PARAMS;
    double A[NX * NX], At[NX * NX], Sx[6 * NX], lam[NX];
    ru_block<YK>(m, dU, x, xp, CM_W_XI, A, Sx);
    for (int r = 0; r < NX; ++r) for (int c = 0; c < NX; ++c) At[r * NX + c] = A[c * NX + r];
    for (int k = 0; k < NX; ++k) {
        double t = xin ? xin[k] : 0.0;
        for (int r = 0; r < 6; ++r) t += Sx[r * NX + k] * sb[r];
        lam[k] = t;
    }
    const bool ok = ru_solve(NX, At, lam, 1);
    if (lam_out) for (int k = 0; k < NX; ++k) lam_out[k] = lam[k];
    if (pbar) {
        double Cp[NX * NP_], Sp[6 * NP_];
        ru_block<YK>(m, dU, x, xp, CM_W_PARAMS, Cp, Sp);
        for (int j = 0; j < NP_; ++j) {
            double t = 0.0;
            for (int r = 0; r < 6; ++r) t += Sp[r * NP_ + j] * sb[r];
            for (int k = 0; k < NX; ++k) t -= Cp[k * NP_ + j] * lam[k];
            pbar[j] = t;
        }
    }
    if (xpbar) {
        double Cxp[NX * NX];
        ru_block<YK>(m, dU, x, xp, CM_W_XI_PREV, Cxp, nullptr);
        for (int c = 0; c < NX; ++c) {
            double t = 0.0;
            for (int k = 0; k < NX; ++k) t -= Cxp[k * NX + c] * lam[k];
            xpbar[c] = t;
        }
    }
    if (ubar) {
        double Cu[NX], Su[6], t = 0.0;
        ru_block<YK>(m, dU, x, xp, CM_W_U, Cu, Su);
        for (int r = 0; r < 6; ++r) t += Su[r] * sb[r];
        for (int k = 0; k < NX; ++k) t -= Cu[k] * lam[k];
        *ubar = t;
    }
    return ok;
}

// forward parameter sensitivities of one converged step (contract of cm::direct_point)
template <int YK>
CM_D bool ru_direct(const cm_model_desc& m, double dU, const double* x, const double* xp, const double* dxp_dp,
                    double* dx_dp /* 12 x NP */, double* ds_dp /* 6 x NP or null */) {
    constexpr int NX = kRuNX, NP_ = CM_NUM_PARAMS;
    double A[NX * NX], Sx[6 * NX], Cp[NX * NP_], Sp[6 * NP_];
    ru_block<YK>(m, dU, x, xp, CM_W_XI, A, Sx);
    ru_block<YK>(m, dU, x, xp, CM_W_PARAMS, Cp, Sp);
    if (dxp_dp) {
        double Cxp[NX * NX];
        ru_block<YK>(m, dU, x, xp, CM_W_XI_PREV, Cxp, nullptr);
        for (int i = 0; i < NX; ++i) for (int j = 0; j < NP_; ++j) {
            double t = 0.0;
            for (int k = 0; k < NX; ++k) t += Cxp[i * NX + k] * dxp_dp[k * NP_ + j];
            Cp[i * NP_ + j] += t;
        }
    }
    for (int i = 0; i < NX * NP_; ++i) dx_dp[i] = -Cp[i];
    const bool ok = ru_solve(NX, A, dx_dp, NP_);
    if (ds_dp) for (int r = 0; r < 6; ++r) for (int j = 0; j < NP_; ++j) {
        double t = Sp[r * NP_ + j];
        for (int k = 0; k < NX; ++k) t += Sx[r * NX + k] * dx_dp[k * NP_ + j];
        ds_dp[r * NP_ + j] = t;
    }
    return ok;
}

}  // namespace cm
