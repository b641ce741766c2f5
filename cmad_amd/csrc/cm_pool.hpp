// Resumable local Newton: the same iteration as cm::newton / cm::newton_s (make_newton_solve,
// cmad/models/nonlinear_solver.py:102-155, with the quadratic Armijo search of cmad/util/line_search.py:95-189, or the
// plain Newton of newton_solve, :14-85) cut into PASSES with exactly one residual evaluation each, all solver state in a
// small per-lane record.  A lane can therefore finish one Gauss point and start the next in the middle of its wavefront's
// loop: the work-pool kernels (k_update_pool, cmad_hip.hip) keep every lane iterating instead of letting the lanes of a
// wavefront wait for its slowest point -- on the iteration-bound workloads a wavefront otherwise runs max(iterations)
// over its 64 points (Hosford a = 100: 9.6 against a mean of 2.4; hybrid Hill + network: 5.0 against 2.4).
//
// One pass = evaluate the residual (and what the Jacobian needs) at x, then act on what arrived:
//   INIT     first evaluation at x_prev: fixes the relative tolerance, then as ITERATE
//   ITERATE  x is an accepted iterate: convergence test (nonlinear_solver.py:140-150), iteration cap, else Newton step;
//            with the line search the base point and the direction are parked in the lane's LDS column and x becomes the
//            first trial (alpha = 1)
//   TRIAL    merit of a trial arrived: Armijo test -> accepted (the trial IS the next iterate: continue as ITERATE in the
//            same pass, its residual reused like the reference's `aux`), else next trial / lowest-merit step / base point
//   BEST     the lowest-merit step re-evaluated -> next iterate
//   BASE, FULL  every trial was non-finite: full step with the base residual carried (line_search.py:181-183)
// Iterates and iteration counts are those of cm::newton_s / cm::newton (tests run both against the oracle).
#pragma once
#include "cm_structured.hpp"

namespace cm {

enum { CM_PH_INIT = 0, CM_PH_ITERATE = 1, CM_PH_TRIAL = 2, CM_PH_BEST = 3, CM_PH_BASE = 4, CM_PH_FULL = 5 };

struct PassState {
    int phase, n, it;
    uint32_t flags;
    double alpha, best_alpha, best_phi, cc, rel2;
};

CM_D void pass_reset(PassState& s) {
    s.phase = CM_PH_INIT; s.n = 0; s.it = 0; s.flags = 0;
    s.alpha = 1.0; s.best_alpha = 1.0; s.best_phi = INFINITY; s.cc = 0.0; s.rel2 = 0.0;
}

// `running` lanes advance by one pass; the others only take part in the (lockstep) evaluation.  `stage`: 2 * NX doubles
// of lane-private storage (LS only).  On return with running == false, x is the solver's result and s.flags | s.it its
// status word.
template <int DEF, int YK, int MK, bool LS>
CM_D void newton_pass(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, double* x,
                      PassState& s, bool& running, LaneStage stage) {
    constexpr int NX = Dims<DEF>::NX;
    constexpr bool STRUCT = (MK == CM_SMALL_ELASTIC_PLASTIC) && has_structured<DEF, YK>();
    double C[NX];
    EvalS<STRUCT ? YK : CM_YIELD_J2> evs;
    Eval<DEF> ev;
    double Ht[STRUCT ? 1 : 6][6];
    if constexpr (STRUCT) residual_s<YK, DEF>(m, eg, z, x, xp, evs, C);
    else residual_mk<MK, DEF, YK, true>(m, eg, z, x, xp, ev, C, Ht);
    if (running) {
        bool at_iterate = false;
        if (s.phase == CM_PH_INIT) {
            s.rel2 = m.rel_tol * m.rel_tol * dot<NX>(C, C);
            at_iterate = true;
        } else if (s.phase == CM_PH_ITERATE) {
            at_iterate = true;
        } else if constexpr (LS) {
            if (s.phase == CM_PH_TRIAL && m.ls_kind == CM_LS_LEGACY) {  // uniform: newton_solve's backtracking (ls_trial_legacy)
                const double step = ls_trial_legacy(m, 0.5 * dot<NX>(C, C), s.cc, s.alpha, s.n);
                if (step == 0.0) { ++s.it; at_iterate = true; }
                else {
#pragma unroll
                    for (int k = 0; k < NX; ++k) x[k] -= step * stage.at(NX + k);
                }
            } else if (s.phase == CM_PH_TRIAL) {
                const double phi = 0.5 * dot<NX>(C, C);                // merit; phi(0) = cc / 2, phi'(0) = -cc
                const bool finite = isfinite(phi);
                if (finite && phi < s.best_phi) { s.best_alpha = s.alpha; s.best_phi = phi; }
                // ls_max_evals == 0 (uniform): plain Newton through this kernel -- the full step is the next iterate whatever its
                // merit, exactly the LS = false code (the cold configurations are built once, cmad_hip.hip always_searches<>)
                const bool accepted = (m.ls_max_evals <= 0) || (finite && (phi <= 0.5 * s.cc + s.alpha * (m.ls_c1 * -s.cc)));
                ++s.n;
                if (accepted) { ++s.it; at_iterate = true; }
                else if (s.n < m.ls_max_evals) {
                    const double am = quad_min(0.5 * s.cc, -s.cc, s.alpha, phi);
                    s.alpha = finite ? fmin(fmax(am, m.ls_lo * s.alpha), m.ls_hi * s.alpha) : 0.5 * s.alpha;
#pragma unroll
                    for (int k = 0; k < NX; ++k) x[k] = stage.at(k) - s.alpha * stage.at(NX + k);
                } else if (s.best_phi < INFINITY) {                     // no trial accepted: the lowest-merit step
#pragma unroll
                    for (int k = 0; k < NX; ++k) x[k] = stage.at(k) - s.best_alpha * stage.at(NX + k);
                    s.phase = CM_PH_BEST;
                } else {                                                // base point again, to recover its residual
#pragma unroll
                    for (int k = 0; k < NX; ++k) x[k] = stage.at(k);
                    s.phase = CM_PH_BASE;
                }
            } else if (s.phase == CM_PH_BEST) {
                ++s.it; at_iterate = true;
            } else if (s.phase == CM_PH_BASE) {                         // C is the base residual: park it, take the full step
#pragma unroll
                for (int k = 0; k < NX; ++k) { x[k] -= stage.at(NX + k); stage.at(k) = C[k]; }
                s.phase = CM_PH_FULL;
            } else {                                                    // CM_PH_FULL: evaluated at the full step, base residual carried
#pragma unroll
                for (int k = 0; k < NX; ++k) C[k] = stage.at(k);
                ++s.it; at_iterate = true;
            }
        }
        if (at_iterate) {
            const double nsq = dot<NX>(C, C), abs2 = m.abs_tol * m.abs_tol;
            if ((nsq < s.rel2) || (nsq < abs2)) { running = false; s.flags |= CM_STATUS_CONVERGED; }
            else if (s.it >= m.max_iters) running = false;
            else {
                double delta[NX];
                if constexpr (STRUCT) {
                    PlasticOpFor<YK> op;
                    op_build<YK>(m, evs, op);
                    if (!op.ok) s.flags |= CM_STATUS_SINGULAR;
                    if (!solve_s<DEF, false>(m, op, evs, z, C, delta)) s.flags |= CM_STATUS_SINGULAR;
                } else {
                    if constexpr (DEF == CM_UNIAXIAL_STRESS && MK == CM_SMALL_ELASTIC_PLASTIC) {
                        if (!uniaxial_solve<YK>(m, z, ev, Ht, C, delta)) s.flags |= CM_STATUS_SINGULAR;    // the 9 x 9 step through 4 x 4
                    } else {
                        double A[NX][NX];
                        jacobian_mk<MK, DEF>(m, z, ev, Ht, A);
                        if (!lu_factor<NX>(A)) s.flags |= CM_STATUS_SINGULAR;
#pragma unroll
                        for (int k = 0; k < NX; ++k) delta[k] = C[k];
                        lu_subst<NX>(A, delta);
                    }
                }
                if constexpr (LS) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) { stage.at(k) = x[k]; stage.at(NX + k) = delta[k]; x[k] -= delta[k]; }
                    s.cc = nsq; s.alpha = 1.0; s.best_alpha = 1.0; s.best_phi = INFINITY; s.n = 0;
                    s.phase = CM_PH_TRIAL;
                } else {
#pragma unroll
                    for (int k = 0; k < NX; ++k) x[k] -= delta[k];
                    ++s.it;
                    s.phase = CM_PH_ITERATE;
                }
            }
        }
    }
}

// the whole solve of one point by passes (host build and single-point use): same contract as cm::newton_any
template <int DEF, int YK, int MK, bool LS>
CM_D uint32_t newton_by_passes(const cm_model_desc& m, const double eg[6], const double* z, const double* xp, double* x,
                               LaneStage stage) {
    constexpr int NX = Dims<DEF>::NX;
    PassState s;
    pass_reset(s);
#pragma unroll
    for (int k = 0; k < NX; ++k) x[k] = xp[k];
    bool running = true;
    while (running) newton_pass<DEF, YK, MK, LS>(m, eg, z, xp, x, s, running, stage);
    return s.flags | (uint32_t)s.it;
}

}  // namespace cm
