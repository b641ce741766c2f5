// CPU port of the hot path for bench.py's `cpu_baseline` leg: the SAME per-point arithmetic the HIP kernels run
// (cmad_amd/csrc/cm_structured.hpp compiled for the host with -DCM_HOST_BUILD: hand-derived residual / Jacobian blocks, the
// structured solve, the J2 radial-line Newton of cm::newton_j2_line, the reverse sweep of cm::reverse_point_s / the closed
// form of cm::reverse_j2_radial) in an OpenMP loop over the points, g++ -O3.  SURVEY.md 8(d) asks for a "same algorithm"
// C++/OpenMP restatement timed on the GPU box's host cores; the nested-dual oracle (cmad_oracle.cpp) differentiates by
// forward-mode AD like the reference and is ~15x slower per point than this, so both are reported.
//
// BENCH / TEST INFRASTRUCTURE ONLY (like everything under oracle/): nothing under cmad_amd/ loads it, and it is checked
// against the oracle in tests/test_host_logic.py before its timing means anything.
#define CM_HOST_BUILD 1
#include "../cmad_amd/csrc/cm_structured.hpp"
#include <omp.h>
using namespace cm;

// J2 / Hill / Hosford, FULL_3D, Q = I: update + vjp w.r.t. the 12 native parameters given sigma_bar (cm_update_and_vjp).
// general != 0: the general 7-dof structured Newton (newton_s) and the transposed solve instead of the J2 radial line.
template <int YK>
static void run(const cm_model_desc& m, int64_t B, const double* gradu, const double* xi_prev, const double* sbar,
                double* xi, double* sigma, double* grad, int general, int nthreads) {
    double g[CM_NUM_PARAMS];
    for (int j = 0; j < CM_NUM_PARAMS; ++j) g[j] = 0.0;
#pragma omp parallel num_threads(nthreads)
    {
        double gl[CM_NUM_PARAMS];
        for (int j = 0; j < CM_NUM_PARAMS; ++j) gl[j] = 0.0;
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            double G[9], xp[7], x[7], eg[6], sb[6], pb[CM_NUM_PARAMS];
            for (int k = 0; k < 9; ++k) G[k] = gradu[k * B + b];
            for (int k = 0; k < 7; ++k) xp[k] = xi_prev[k * B + b];
            for (int k = 0; k < 6; ++k) sb[k] = sbar[k * B + b];
            strain_from_gradu<CM_FULL_3D, false>(m, G, eg);
            EvalS<YK> ev;
            bool radial = false;
            uint32_t st;
            if constexpr (YK == CM_YIELD_J2) {
                if (!general) { st = newton_j2_line<false>(m, eg, xp, x, true, ev); radial = true; }
            }
            if (!radial) st = newton_s<YK, false>(m, eg, xp, x, true, ev);
            if constexpr (YK == CM_YIELD_J2) {
                if (radial && (st & CM_STATUS_CONVERGED)) reverse_j2_radial(m, eg, x, sb, ev, pb);
                else reverse_point_s<YK, true>(m, eg, x, xp, sb, nullptr, pb, nullptr, nullptr, &ev);
            } else reverse_point_s<YK, true>(m, eg, x, xp, sb, nullptr, pb, nullptr, nullptr, &ev);
            Eval<CM_FULL_3D> e2;
            strain_stress<CM_FULL_3D>(m, eg, nullptr, x, e2);
            for (int k = 0; k < 7; ++k) xi[k * B + b] = x[k];
            for (int k = 0; k < 6; ++k) sigma[k * B + b] = e2.s[k];
            for (int j = 0; j < CM_NUM_PARAMS; ++j) gl[j] += pb[j];
        }
#pragma omp critical
        for (int j = 0; j < CM_NUM_PARAMS; ++j) g[j] += gl[j];
    }
    for (int j = 0; j < CM_NUM_PARAMS; ++j) grad[j] = g[j];
}

extern "C" int port_update_and_vjp(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
                                   const double* sbar, double* xi, double* sigma, double* grad, int general, int nthreads) {
    if (!m || m->def_type != CM_FULL_3D || !m->rotation_is_identity || m->ls_max_evals != 0) return -1;
    if (nthreads < 1) nthreads = 1;
    switch (m->yield_kind) {
        case CM_YIELD_J2: run<CM_YIELD_J2>(*m, B, gradu, xi_prev, sbar, xi, sigma, grad, general, nthreads); return 0;
        case CM_YIELD_HILL: run<CM_YIELD_HILL>(*m, B, gradu, xi_prev, sbar, xi, sigma, grad, 1, nthreads); return 0;
        case CM_YIELD_HOSFORD: run<CM_YIELD_HOSFORD>(*m, B, gradu, xi_prev, sbar, xi, sigma, grad, 1, nthreads); return 0;
    }
    return -1;
}
extern "C" int port_sizeof_desc(void) { return (int)sizeof(cm_model_desc); }
