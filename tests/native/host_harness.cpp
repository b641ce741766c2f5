// TEST INFRASTRUCTURE ONLY: compiles the per-point device math of cmad_amd/csrc/cm_device.hpp for the
// host (g++, -DCM_HOST_BUILD) so the hand-derived formulas can be checked against the oracle -- and
// run under AddressSanitizer/UBSan -- on machines without a GPU.  Not a product path: nothing under
// cmad_amd/ loads this.
#include <cstdint>
#include <cstring>
#define CM_HOST_BUILD 1
#include "../../cmad_amd/csrc/cm_pool.hpp"
#include "../../cmad_amd/csrc/cm_hessian.hpp"

using namespace cm;

// The file compiles in independent pieces (-DHH_PART=0..6, tests/host_harness_lib.py) so the template instantiations
// build in parallel; without HH_PART everything is one translation unit.
#ifndef HH_PART
#define HH_PART (-1)
#endif
#define HH_HAS(k) (HH_PART == -1 || HH_PART == (k))
#define g_dense hh_g_dense
#define HH_LS(m) ((m).ls_max_evals > 0 || hh_g_force_ls)
#if HH_HAS(0)
int hh_g_dense = 0;       // 1: force the dense 7x7 path also for FULL_3D
int hh_g_passes = 0;      // 1: solve by cm::newton_pass (the resumable form the work-pool kernels run)
int hh_g_force_ls = 0;    // 1: the LS = true instantiations also for ls_max_evals == 0 (what the library's cold configurations run)
int hh_g_radial_vjp = 0;  // 1: J2 / FULL_3D parameter gradient by cm::reverse_j2_radial (what the fused J2 kernels use)
#else
extern int hh_g_dense;
extern int hh_g_passes;
extern int hh_g_force_ls;
extern int hh_g_radial_vjp;
#endif

template <int DEF, int YK, bool ROT>
static void run_update(const cm_model_desc& m, int64_t B, const double* gradu, const double* xi_prev,
                       double* xi, double* sigma, uint32_t* status, double* dsig) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    for (int64_t b = 0; b < B; ++b) {
        double G[NU], xp[NX], x[NX], eg[6], z[Dims<DEF>::NZ];
        for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + b];
        for (int k = 0; k < NX; ++k) xp[k] = xi_prev[k * B + b];
        strain_from_gradu<DEF, ROT>(m, G, eg);
        strain_z<DEF, ROT>(m, z);
        const bool ls = HH_LS(m);
        double parked[2 * 9];                      // the kernels keep this in the lane's LDS column
        const LaneStage stage{parked, 1};
        uint32_t st;
        bool done = false;
        if (hh_g_passes) {
            st = ls ? newton_by_passes<DEF, YK, CM_SMALL_ELASTIC_PLASTIC, true>(m, eg, z, xp, x, stage)
                    : newton_by_passes<DEF, YK, CM_SMALL_ELASTIC_PLASTIC, false>(m, eg, z, xp, x, stage);
            done = true;
        }
        if constexpr (has_fast_newton<DEF, YK, false>()) {          // same choice as launch_update (cmad_hip.hip)
            if (!done && !g_dense && use_fast_newton(&m)) {
                st = ls ? newton_any<DEF, YK, true, true, true>(m, eg, z, xp, x, true, stage)
                        : newton_any<DEF, YK, false, true, true>(m, eg, z, xp, x, true, stage);
                done = true;
            }
        }
        if (!done)
            st = g_dense ? (ls ? newton_any<DEF, YK, true, false>(m, eg, z, xp, x, true, stage) : newton_any<DEF, YK, false, false>(m, eg, z, xp, x, true, stage))
                         : (ls ? newton_any<DEF, YK, true, true>(m, eg, z, xp, x, true, stage) : newton_any<DEF, YK, false, true>(m, eg, z, xp, x, true, stage));
        Eval<DEF> ev;
        strain_stress<DEF>(m, eg, z, x, ev);
        double sg[6];
        to_global<ROT>(m, ev.s, sg);
        for (int k = 0; k < NX; ++k) xi[k * B + b] = x[k];
        if (sigma) for (int k = 0; k < 6; ++k) sigma[k * B + b] = sg[k];
        if (status) status[b] = st;
        if (dsig) {
            double T[6][6];
            if (g_dense) tangent_any<DEF, YK, false>(m, eg, z, x, xp, T); else tangent_any<DEF, YK, true>(m, eg, z, x, xp, T);
            for (int c = 0; c < NU; ++c) {
                double Gd[NU], dm[6], t[6], tg[6];
                for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
                strain_from_gradu<DEF, ROT>(m, Gd, dm);
                for (int r = 0; r < 6; ++r) { double s = 0; for (int l = 0; l < 6; ++l) s += T[r][l] * dm[l]; t[r] = s; }
                to_global<ROT>(m, t, tg);
                for (int r = 0; r < 6; ++r) dsig[(int64_t)(r * NU + c) * B + b] = tg[r];
            }
        }
    }
}

// reverse sweep for a given sigma cotangent at given converged xi: pbar summed over points (KP order)
template <int DEF, int YK, bool ROT>
static void run_vjp(const cm_model_desc& m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
                    const double* sbar, const double* xin, double* grad, double* xpbar, double* gbar) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    for (int k = 0; k < CM_NUM_PARAMS; ++k) grad[k] = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        double G[NU], xp[NX], x[NX], eg[6], z[Dims<DEF>::NZ], sb[6], sbm[6], pb[CM_NUM_PARAMS], xb[NX], eb[6], xi_in[NX];
        for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + b];
        for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + b]; x[k] = xi[k * B + b]; if (xin) xi_in[k] = xin[k * B + b]; }
        for (int k = 0; k < 6; ++k) sb[k] = sbar[k * B + b];
        strain_from_gradu<DEF, ROT>(m, G, eg);
        strain_z<DEF, ROT>(m, z);
        cotangent_to_material<ROT>(m, sb, sbm);
        if (g_dense) reverse_any<DEF, YK, false>(m, eg, z, x, xp, sbm, xin ? xi_in : nullptr, pb, xb, eb);
        else reverse_any<DEF, YK, true>(m, eg, z, x, xp, sbm, xin ? xi_in : nullptr, pb, xb, eb);
        if constexpr (DEF == CM_FULL_3D && YK == CM_YIELD_J2) {
            if (hh_g_radial_vjp && !xin) {      // the fused J2 kernels' closed-form parameter gradient (cm::reverse_j2_radial)
                EvalS<CM_YIELD_J2> evs;
                double C[7];
                residual_s<CM_YIELD_J2>(m, eg, x, xp, evs, C);
                reverse_j2_radial(m, eg, x, sbm, evs, pb);
            }
        }
        if constexpr (DEF == CM_PLANE_STRESS && YK == CM_YIELD_J2) {
            if (hh_g_radial_vjp && !xin) {      // ... and the plane's (cm::reverse_j2_plane)
                EvalS<CM_YIELD_J2> evs;
                double C[8];
                residual_s<CM_YIELD_J2, CM_PLANE_STRESS>(m, eg, z, x, xp, evs, C);
                reverse_j2_plane(m, eg, z, x, sbm, evs, pb);
            }
        }
        for (int k = 0; k < CM_NUM_PARAMS; ++k) grad[k] += pb[k];
        if (xpbar) for (int k = 0; k < NX; ++k) xpbar[k * B + b] = xb[k];
        if (gbar) for (int c = 0; c < NU; ++c) {
            double Gd[NU], dm[6];
            for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
            strain_from_gradu<DEF, ROT>(m, Gd, dm);
            gbar[c * B + b] = dot<6>(eb, dm);
        }
    }
}

template <int DEF, int YK, bool ROT>
static void run_vjp_rate(const cm_model_desc& m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev,
                         const double* xi, const double* sbar, const double* xin, double* grad, double* xpbar, double* gbar) {
    constexpr int NX = nx_of<DEF, CM_SMALL_RATE_ELASTIC_PLASTIC>(), NU = Dims<DEF>::NU;
    for (int k = 0; k < CM_NUM_PARAMS; ++k) grad[k] = 0.0;
    if constexpr (true) {
        for (int64_t b = 0; b < B; ++b) {
            double G[NU], xp[NX], x[NX], deg[6], z[Dims<DEF>::NZ], sb[6], sbm[6], pb[CM_NUM_PARAMS], xb[NX], eb[6], xi_in[NX];
            for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + b] - gradu_prev[k * B + b];
            for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + b]; x[k] = xi[k * B + b]; if (xin) xi_in[k] = xin[k * B + b]; }
            for (int k = 0; k < 6; ++k) sb[k] = sbar[k * B + b];
            if constexpr (DEF == CM_UNIAXIAL_STRESS) {           // 12 dofs: blocks by forward-mode evaluation (cm_rate_uniaxial.hpp)
                double ubar = 0.0;
                ru_reverse<YK>(m, G[0], x, xp, sb, xin ? xi_in : nullptr, pb, xb, &ubar);
                for (int k = 0; k < CM_NUM_PARAMS; ++k) grad[k] += pb[k];
                if (xpbar) for (int k = 0; k < NX; ++k) xpbar[k * B + b] = xb[k];
                if (gbar) gbar[b] = ubar;
                continue;
            }
            strain_from_gradu<DEF, ROT>(m, G, deg);
            strain_z<DEF, ROT>(m, z);
            cotangent_to_material<ROT>(m, sb, sbm);
            if (g_dense) reverse_rate_any<(DEF == CM_UNIAXIAL_STRESS) ? CM_FULL_3D : DEF, YK, false>(m, deg, z, x, xp, sbm, xin ? xi_in : nullptr, pb, xb, eb);
            else reverse_rate_any<(DEF == CM_UNIAXIAL_STRESS) ? CM_FULL_3D : DEF, YK, true>(m, deg, z, x, xp, sbm, xin ? xi_in : nullptr, pb, xb, eb);
            for (int k = 0; k < CM_NUM_PARAMS; ++k) grad[k] += pb[k];
            if (xpbar) for (int k = 0; k < NX; ++k) xpbar[k * B + b] = xb[k];
            if (gbar) for (int c = 0; c < NU; ++c) {
                double Gd[NU], dm[6];
                for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
                strain_from_gradu<DEF, ROT>(m, Gd, dm);
                gbar[c * B + b] = dot<6>(eb, dm);
            }
        }
    }
}

template <int DEF, int YK, bool ROT>
static void run_evaluate(const cm_model_desc& m, int64_t B, int which, const double* gradu, const double* xi_prev,
                         const double* xi, double* C_out, double* J_out, double* s_out, double* S_out) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    const int ncols = (which == CM_W_XI || which == CM_W_XI_PREV) ? NX : (which == CM_W_PARAMS ? CM_NUM_PARAMS : NU);
    for (int64_t b = 0; b < B; ++b) {
        double G[NU], xp[NX], x[NX], C[NX], sg[6], J[NX * CM_NUM_PARAMS], S[6 * CM_NUM_PARAMS];
        for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + b];
        for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + b]; x[k] = xi[k * B + b]; }
        evaluate_blocks<DEF, YK, ROT>(m, G, x, xp, which, C, J, sg, S);
        for (int k = 0; k < NX; ++k) C_out[k * B + b] = C[k];
        for (int k = 0; k < 6; ++k) s_out[k * B + b] = sg[k];
        if (which != CM_W_NONE) {
            for (int i = 0; i < NX * ncols; ++i) J_out[(int64_t)i * B + b] = J[i];
            for (int i = 0; i < 6 * ncols; ++i) S_out[(int64_t)i * B + b] = S[i];
        }
    }
}

template <int DEF, int YK, bool ROT>
static void run_update_rate(const cm_model_desc& m, int64_t B, const double* gradu, const double* gradu_prev,
                            const double* xi_prev, double* xi, double* sigma, uint32_t* status) {
    constexpr int NX = nx_of<DEF, CM_SMALL_RATE_ELASTIC_PLASTIC>(), NU = Dims<DEF>::NU;
    for (int64_t b = 0; b < B; ++b) {
        double G[NU], xp[NX], x[NX], deg[6], z[Dims<DEF>::NZ], sg[6];
        for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + b] - gradu_prev[k * B + b];
        for (int k = 0; k < NX; ++k) xp[k] = xi_prev[k * B + b];
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
            const uint32_t stu = HH_LS(m) ? ru_newton<YK, true>(m, G[0], xp, x, true) : ru_newton<YK, false>(m, G[0], xp, x, true);
            to_global<ROT>(m, x, sg);
            for (int k = 0; k < NX; ++k) xi[k * B + b] = x[k];
            for (int k = 0; k < 6; ++k) sigma[k * B + b] = sg[k];
            status[b] = stu;
            continue;
        }
        strain_from_gradu<DEF, ROT>(m, G, deg);
        strain_z<DEF, ROT>(m, z);
        double parked[2 * 9];
        const LaneStage stage{parked, 1};
        uint32_t st;
        if (hh_g_passes) st = HH_LS(m) ? newton_by_passes<(DEF == CM_UNIAXIAL_STRESS) ? CM_FULL_3D : DEF, YK, CM_SMALL_RATE_ELASTIC_PLASTIC, true>(m, deg, z, xp, x, stage)
                                                   : newton_by_passes<(DEF == CM_UNIAXIAL_STRESS) ? CM_FULL_3D : DEF, YK, CM_SMALL_RATE_ELASTIC_PLASTIC, false>(m, deg, z, xp, x, stage);
        else {      // same choice as k_update_rate: the structured solver where there is one; the "dense" variant keeps cm::newton
            constexpr int D2 = (DEF == CM_UNIAXIAL_STRESS) ? CM_FULL_3D : DEF;
            if (g_dense) st = HH_LS(m) ? newton_rate_any<D2, YK, true, false>(m, deg, z, xp, x, true, stage)
                                                   : newton_rate_any<D2, YK, false, false>(m, deg, z, xp, x, true, stage);
            else st = HH_LS(m) ? newton_rate_any<D2, YK, true, true>(m, deg, z, xp, x, true, stage)
                                           : newton_rate_any<D2, YK, false, true>(m, deg, z, xp, x, true, stage);
        }
        to_global<ROT>(m, x, sg);
        for (int k = 0; k < NX; ++k) xi[k * B + b] = x[k];
        for (int k = 0; k < 6; ++k) sigma[k * B + b] = sg[k];
        status[b] = st;
    }
}

// d sigma_global / d grad u of the rate form at converged states (the epilogue of k_update_rate<TANGENT>)
template <int DEF, int YK, bool ROT>
static void run_tangent_rate(const cm_model_desc& m, int64_t B, const double* gradu, const double* gradu_prev,
                             const double* xi_prev, const double* xi, double* dsig) {
    constexpr int NX = nx_of<DEF, CM_SMALL_RATE_ELASTIC_PLASTIC>(), NU = Dims<DEF>::NU;
    for (int64_t b = 0; b < B; ++b) {
        double G[NU], xp[NX], x[NX], deg[6], z[Dims<DEF>::NZ], T[6][6];
        for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + b] - gradu_prev[k * B + b];
        for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + b]; x[k] = xi[k * B + b]; }
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
            double ds[6];
            ru_tangent<YK>(m, G[0], x, xp, ds);
            for (int r = 0; r < 6; ++r) dsig[(int64_t)r * B + b] = ds[r];
            continue;
        }
        strain_from_gradu<DEF, ROT>(m, G, deg);
        strain_z<DEF, ROT>(m, z);
        if (g_dense) tangent_rate_any<(DEF == CM_UNIAXIAL_STRESS) ? CM_FULL_3D : DEF, YK, false>(m, deg, z, x, xp, T);
        else tangent_rate_any<(DEF == CM_UNIAXIAL_STRESS) ? CM_FULL_3D : DEF, YK, true>(m, deg, z, x, xp, T);
        for (int c = 0; c < NU; ++c) {
            double Gd[NU], dm[6], t[6], tg[6];
            for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
            strain_from_gradu<DEF, ROT>(m, Gd, dm);
            for (int r = 0; r < 6; ++r) { double s = 0.0; for (int l = 0; l < 6; ++l) s += T[r][l] * dm[l]; t[r] = s; }
            to_global<ROT>(m, t, tg);
            for (int r = 0; r < 6; ++r) dsig[(int64_t)(r * NU + c) * B + b] = tg[r];
        }
    }
}

template <int DEF, int YK, bool ROT>
static void run_evaluate_rate(const cm_model_desc& m, int64_t B, int which, const double* gradu, const double* gradu_prev,
                              const double* xi_prev, const double* xi, double* C_out, double* J_out, double* s_out, double* S_out) {
    constexpr int NX = nx_of<DEF, CM_SMALL_RATE_ELASTIC_PLASTIC>(), NU = Dims<DEF>::NU;
    const int ncols = (which == CM_W_XI || which == CM_W_XI_PREV) ? NX : (which == CM_W_PARAMS ? CM_NUM_PARAMS : NU);
    for (int64_t b = 0; b < B; ++b) {
        double G[NU], Gp[NU], xp[NX], x[NX], C[NX], sg[6], J[NX * CM_NUM_PARAMS], S[6 * CM_NUM_PARAMS];
        for (int k = 0; k < NU; ++k) { G[k] = gradu[k * B + b]; Gp[k] = gradu_prev[k * B + b]; }
        for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + b]; x[k] = xi[k * B + b]; }
        if constexpr (DEF == CM_UNIAXIAL_STRESS) {
            ru_eval<YK>(m, G[0] - Gp[0], x, xp, C, sg);
            if (which != CM_W_NONE) ru_block<YK>(m, G[0] - Gp[0], x, xp, which, J, S);
        } else
        evaluate_blocks_rate<DEF, YK, ROT>(m, G, Gp, x, xp, which, C, J, sg, S);
        for (int k = 0; k < NX; ++k) C_out[k * B + b] = C[k];
        for (int k = 0; k < 6; ++k) s_out[k * B + b] = sg[k];
        if (which != CM_W_NONE) {
            for (int i = 0; i < NX * ncols; ++i) J_out[(int64_t)i * B + b] = J[i];
            for (int i = 0; i < 6 * ncols; ++i) S_out[(int64_t)i * B + b] = S[i];
        }
    }
}

template <int DEF, int YK, bool ROT, int MK = CM_SMALL_ELASTIC_PLASTIC>
static void run_hessians(const cm_model_desc& m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                         double* d2C, double* d2S, double* dC, double* dS, double* C0 = nullptr, double* S0 = nullptr) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU, NQ = 2 * NX + CM_NUM_PARAMS;
    if constexpr (true) {
        for (int64_t pt = 0; pt < B; ++pt) {
            double G[NU], xp[NX], x[NX], oC[NX], oS[6], oCa[NX], oSa[6], oC0[NX], oS0[6];
            for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + pt] - (gradu_prev ? gradu_prev[k * B + pt] : 0.0);
            for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + pt]; x[k] = xi[k * B + pt]; }
            for (int a = 0; a < NQ; ++a) for (int b = a; b < NQ; ++b) {
                const bool first = (a == 0 && b == 0);
                hessian_pair<DEF, CM_YIELD_ANY, ROT, MK>(m, G, x, xp, a, b, oC, oS, oCa, oSa, first ? oC0 : nullptr, first ? oS0 : nullptr);
                for (int k = 0; k < NX; ++k) { d2C[((pt * NX + k) * NQ + a) * NQ + b] = oC[k]; d2C[((pt * NX + k) * NQ + b) * NQ + a] = oC[k]; }
                for (int k = 0; k < 6; ++k) { d2S[((pt * 6 + k) * NQ + a) * NQ + b] = oS[k]; d2S[((pt * 6 + k) * NQ + b) * NQ + a] = oS[k]; }
                if (a == b) {
                    for (int k = 0; k < NX; ++k) dC[(pt * NX + k) * NQ + a] = oCa[k];
                    for (int k = 0; k < 6; ++k) dS[(pt * 6 + k) * NQ + a] = oSa[k];
                }
                if (first) {
                    if (C0) for (int k = 0; k < NX; ++k) C0[pt * NX + k] = oC0[k];
                    if (S0) for (int k = 0; k < 6; ++k) S0[pt * 6 + k] = oS0[k];
                }
            }
        }
    }
}

// whole-history objective + gradient (cm::history_point, the body of k_history) with plain row indexing
struct HostRowsIO {
    int64_t B, b;
    template <int N> void load(const double* base, int64_t row0, double* out) const { for (int k = 0; k < N; ++k) out[k] = base[(row0 + k) * B + b]; }
    template <int N> void store(double* base, int64_t row0, const double* v) const { for (int k = 0; k < N; ++k) base[(row0 + k) * B + b] = v[k]; }
    void phase_barrier() const {}
    void store_status(uint32_t* base, int64_t row, uint32_t v) const { base[row * B + b] = v; }
    double get(const double* base, int64_t row) const { return base[row * B + b]; }
    void put(double* base, int64_t row, double v) const { base[row * B + b] = v; }
};

template <int DEF, int YK, bool ROT, int MK>
static void run_history(const cm_model_desc& m, int64_t B, int K, const double* gradu_hist, const double* data_hist,
                        const double* wsq6, const double* xi0, double* xi_hist, double* out,
                        HistoryCotangents hc = HistoryCotangents{nullptr, nullptr, nullptr}) {
    for (int k = 0; k < 1 + CM_NUM_PARAMS; ++k) out[k] = 0.0;
    const bool ls = HH_LS(m);
    for (int64_t b = 0; b < B; ++b) {
        double red[1 + CM_NUM_PARAMS] = {0.0};
        double parked[2 * 9];
        const LaneStage stage{parked, 1};
        const HostRowsIO io{B, b};
        bool done = false;
        if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC && has_fast_newton<DEF, YK, false>()) {   // same choice as launch_history
            if (use_fast_newton(&m)) {
                if (ls) history_point<DEF, YK, ROT, true, MK, true>(m, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, true, stage, io, red, hc);
                else history_point<DEF, YK, ROT, false, MK, true>(m, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, true, stage, io, red, hc);
                done = true;
            }
        }
        if (done) {}
        else if (ls) history_point<DEF, YK, ROT, true, MK>(m, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, true, stage, io, red, hc);
        else history_point<DEF, YK, ROT, false, MK>(m, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, true, stage, io, red, hc);
        for (int k = 0; k < 1 + CM_NUM_PARAMS; ++k) out[k] += red[k];
    }
}

template <int DEF, int YK, bool ROT, int MK>
static void run_primal_history(const cm_model_desc& m, int64_t B, int K, const double* gradu_hist, const double* xi0,
                               double* xi_hist, double* sigma_hist, uint32_t* status_hist) {
    const bool ls = HH_LS(m);
    for (int64_t b = 0; b < B; ++b) {
        double parked[2 * 9];
        const LaneStage stage{parked, 1};
        const HostRowsIO io{B, b};
        bool done = false;
        if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC && has_fast_newton<DEF, YK, false>()) {
            if (use_fast_newton(&m)) {
                if (ls) primal_history_point<DEF, YK, ROT, true, MK, true>(m, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist, true, stage, io);
                else primal_history_point<DEF, YK, ROT, false, MK, true>(m, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist, true, stage, io);
                done = true;
            }
        }
        if (done) {}
        else if (ls) primal_history_point<DEF, YK, ROT, true, MK>(m, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist, true, stage, io);
        else primal_history_point<DEF, YK, ROT, false, MK>(m, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist, true, stage, io);
    }
}

template <bool UNI = false, class F>
static int dispatch(const cm_model_desc* m, F&& f) {
    const bool rot = !m->rotation_is_identity;
#define CM_CASE(D, Y) \
    if (m->def_type == D && m->yield_kind == Y) { if (rot) f.template operator()<D, Y, true>(); else f.template operator()<D, Y, false>(); return 0; }
    CM_CASE(CM_FULL_3D, CM_YIELD_J2)
    CM_CASE(CM_FULL_3D, CM_YIELD_HILL)
    CM_CASE(CM_FULL_3D, CM_YIELD_HOSFORD)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_J2)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_HILL)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_HOSFORD)
    CM_CASE(CM_FULL_3D, CM_YIELD_HYBRID_HILL_NN)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_HYBRID_HILL_NN)
    CM_CASE(CM_FULL_3D, CM_YIELD_SCALED_HYBRID_HILL_NN)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_SCALED_HYBRID_HILL_NN)
    CM_CASE(CM_FULL_3D, CM_YIELD_BARLAT)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_BARLAT)
    if constexpr (UNI) {
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_BARLAT)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_J2)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_HILL)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_HOSFORD)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_HYBRID_HILL_NN)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_SCALED_HYBRID_HILL_NN)
    }
#undef CM_CASE
    return -2;
}

extern "C" {
#if HH_HAS(0)
double hh_quad_min(double phi0, double dphi0, double a, double phi) { return quad_min(phi0, dphi0, a, phi); }
#endif
#if HH_HAS(0)
int hh_update(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
              double* xi, double* sigma, uint32_t* status, double* dsig) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_update<D, Y, R>(*m, B, gradu, xi_prev, xi, sigma, status, dsig); });
}
#endif
#if HH_HAS(1)
int hh_vjp(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
           const double* sbar, const double* xin, double* grad, double* xpbar, double* gbar) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_vjp<D, Y, R>(*m, B, gradu, xi_prev, xi, sbar, xin, grad, xpbar, gbar); });
}
#endif
#if HH_HAS(2)
int hh_update_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev,
                   double* xi, double* sigma, uint32_t* status) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_update_rate<D, Y, R>(*m, B, gradu, gradu_prev, xi_prev, xi, sigma, status); });
}
#endif
#if HH_HAS(2)
int hh_vjp_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev,
                const double* xi, const double* sbar, const double* xin, double* grad, double* xpbar, double* gbar) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_vjp_rate<D, Y, R>(*m, B, gradu, gradu_prev, xi_prev, xi, sbar, xin, grad, xpbar, gbar); });
}
#endif
#if HH_HAS(2)
int hh_tangent_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                    const double* xi_prev, const double* xi, double* dsig) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_tangent_rate<D, Y, R>(*m, B, gradu, gradu_prev, xi_prev, xi, dsig); });
}
#endif
#if HH_HAS(2)
int hh_evaluate_rate(const cm_model_desc* m, int64_t B, int which, const double* gradu, const double* gradu_prev,
                     const double* xi_prev, const double* xi, double* C, double* J, double* s, double* S) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_evaluate_rate<D, Y, R>(*m, B, which, gradu, gradu_prev, xi_prev, xi, C, J, s, S); });
}
#endif
#if HH_HAS(4)
int hh_hessians(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
                double* d2C, double* d2S, double* dC, double* dS) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_hessians<D, Y, R>(*m, B, gradu, nullptr, xi_prev, xi, d2C, d2S, dC, dS); });
}
#endif
#if HH_HAS(5)
int hh_hessians_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev,
                     const double* xi, double* d2C, double* d2S, double* dC, double* dS, double* C0, double* S0) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() {
        run_hessians<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(*m, B, gradu, gradu_prev, xi_prev, xi, d2C, d2S, dC, dS, C0, S0); });
}
#endif
#if HH_HAS(3)
int hh_evaluate(const cm_model_desc* m, int64_t B, int which, const double* gradu, const double* xi_prev,
                const double* xi, double* C, double* J, double* s, double* S) {
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { run_evaluate<D, Y, R>(*m, B, which, gradu, xi_prev, xi, C, J, s, S); });
}
#endif
#if HH_HAS(6)
int hh_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* data_hist,
               const double* wsq6, const double* xi0, double* xi_hist, double* out) {
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() {
            run_history<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(*m, B, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, out); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() {
        run_history<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(*m, B, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, out); });
}
#endif
#if HH_HAS(6)
int hh_primal_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* xi0,
                      double* xi_hist, double* sigma_hist, uint32_t* status_hist) {
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() {
            run_primal_history<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(*m, B, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() {
        run_primal_history<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(*m, B, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist); });
}
#endif
#if HH_HAS(3)
int hh_direct_step(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev,
                   const double* xi, const double* dxp_dp, double* dx_dp, double* ds_dp) {
    auto body = [&]<int D, int Y, bool R, int MK>() {
        constexpr int NX = nx_of<D, MK>(), NU = Dims<D>::NU, NP_ = CM_NUM_PARAMS;
        for (int64_t b = 0; b < B; ++b) {
            double G[NU], Gp[NU], xp[NX], x[NX], din[NX * NP_], dout[NX * NP_], dsig[6 * NP_];
            for (int k = 0; k < NU; ++k) { G[k] = gradu[k * B + b]; Gp[k] = gradu_prev ? gradu_prev[k * B + b] : 0.0; }
            for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + b]; x[k] = xi[k * B + b]; }
            if (dxp_dp) for (int i = 0; i < NX * NP_; ++i) din[i] = dxp_dp[(int64_t)i * B + b];
            if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && D == CM_UNIAXIAL_STRESS) ru_direct<Y>(*m, G[0] - Gp[0], x, xp, dxp_dp ? din : nullptr, dout, dsig);
            else {
                // column by column, as k_direct_step_cols / k_direct_history_cols do (cm::direct_column; hh_direct_history below
                // runs the block form cm::direct_point: both are checked against the oracle)
                if constexpr (NX == Dims<D>::NX) {
                    for (int j = 0; j < NP_; ++j) {
                        double dprev[NX], dcol[NX], dscol[6];
                        for (int k = 0; k < NX; ++k) dprev[k] = dxp_dp ? din[k * NP_ + j] : 0.0;
                        direct_column<MK, D, Y, R>(*m, G, Gp, x, xp, j, dxp_dp ? dprev : nullptr, dcol, dscol);
                        for (int k = 0; k < NX; ++k) dout[k * NP_ + j] = dcol[k];
                        for (int r = 0; r < 6; ++r) dsig[r * NP_ + j] = dscol[r];
                    }
                }
            }
            for (int i = 0; i < NX * NP_; ++i) dx_dp[(int64_t)i * B + b] = dout[i];
            for (int i = 0; i < 6 * NP_; ++i) ds_dp[(int64_t)i * B + b] = dsig[i];
        }
    };
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(); });
}
#endif
#if HH_HAS(6)
// cm_adjoint_history: the history adjoint for caller-supplied QoI cotangents (grad[12] = KP gradient)
int hh_adjoint_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* sbar_hist,
                       const double* xibar_hist, const double* xi0, double* xi_hist, double* lam_hist, double* grad) {
    double out[1 + CM_NUM_PARAMS];
    const double wsq0[6] = {0, 0, 0, 0, 0, 0};
    const HistoryCotangents hc{sbar_hist, xibar_hist, lam_hist};
    int rc;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        rc = dispatch<true>(m, [&]<int D, int Y, bool R>() {
            run_history<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(*m, B, K, gradu_hist, nullptr, wsq0, xi0, xi_hist, out, hc); });
    else
        rc = dispatch<true>(m, [&]<int D, int Y, bool R>() {
            run_history<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(*m, B, K, gradu_hist, nullptr, wsq0, xi0, xi_hist, out, hc); });
    for (int j = 0; j < CM_NUM_PARAMS; ++j) grad[j] = out[1 + j];
    return rc;
}
#endif
#if HH_HAS(3)
// cm_direct_history (cm::direct_history_point over the batch)
int hh_direct_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* xi_hist,
                      const double* sbar_hist, const double* xibar_hist, double* dx_dp_hist, double* ds_dp_hist, double* grad) {
    for (int j = 0; j < CM_NUM_PARAMS; ++j) grad[j] = 0.0;
    auto body = [&]<int D, int Y, bool R, int MK>() {
        for (int64_t b = 0; b < B; ++b) {
            double g[CM_NUM_PARAMS];
            direct_history_point<D, Y, true, MK>(*m, K, gradu_hist, xi_hist, sbar_hist, xibar_hist, dx_dp_hist, ds_dp_hist, HostRowsIO{B, b}, g);
            for (int j = 0; j < CM_NUM_PARAMS; ++j) grad[j] += g[j];
        }
    };
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(); });
}
#endif
#if HH_HAS(4)
// stage 1 of cm_hessian_history: W[(step-1)*B + pt][NQ][NQ] (cm::hessian_weight per pair)
int hh_hessian_weights(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* xi_hist,
                       const double* lam_hist, const double* sbar_hist, const double* hss6, const double* hss_hist,
                       const double* hxx_hist, double* W) {
    auto body = [&]<int D, int Y, bool R, int MK>() {
        constexpr int NX = nx_of<D, MK>(), NU = Dims<D>::NU, NQ = 2 * NX + CM_NUM_PARAMS;
        for (int step = 1; step <= K; ++step) for (int64_t pt = 0; pt < B; ++pt) {
            double G[NU], xp[NX], x[NX], lam[NX], sbar[6];
            for (int k = 0; k < NU; ++k) {
                G[k] = gradu_hist[((int64_t)step * NU + k) * B + pt];
                if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) G[k] -= gradu_hist[((int64_t)(step - 1) * NU + k) * B + pt];
            }
            for (int k = 0; k < NX; ++k) {
                xp[k] = xi_hist[((int64_t)(step - 1) * NX + k) * B + pt];
                x[k] = xi_hist[((int64_t)step * NX + k) * B + pt];
                lam[k] = lam_hist[((int64_t)step * NX + k) * B + pt];
            }
            for (int r = 0; r < 6; ++r) sbar[r] = sbar_hist[((int64_t)step * 6 + r) * B + pt];
            const int64_t ps = (int64_t)(step - 1) * B + pt;
            double hs[6];                       // as k_hessian_weights: per-step stress curvature, state curvature on the diagonal
            for (int r = 0; r < 6; ++r) hs[r] = hss_hist ? hss_hist[step * 6 + r] : hss6[r];
            for (int a = 0; a < NQ; ++a) for (int b = a; b < NQ; ++b) {
                double w = hessian_weight<D, CM_YIELD_ANY, true, MK>(*m, G, x, xp, lam, sbar, hs, a, b);
                if (hxx_hist && a == b && a < NX) w += hxx_hist[step * NX + a];
                W[(ps * NQ + a) * NQ + b] = w; W[(ps * NQ + b) * NQ + a] = w;
            }
        }
    };
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(); });
}
#endif
#if HH_HAS(4)
// cm_direct_history_ep (cm::direct_column_ep over the steps): dxe_hist[(K+1)][NX*n_ep][B]
int hh_direct_history_ep(const cm_model_desc* m, int64_t B, int K, int n_ep, const int32_t* ep_index, const double* gradu_hist,
                         const double* xi_hist, double* dxe_hist) {
    auto body = [&]<int D, int Y, bool R, int MK>() {
        if constexpr (!(MK == CM_SMALL_RATE_ELASTIC_PLASTIC && D == CM_UNIAXIAL_STRESS)) {
            constexpr int NX = Dims<D>::NX, NU = Dims<D>::NU;
            for (int64_t b = 0; b < B; ++b) for (int j = 0; j < n_ep; ++j) {
                double d[NX], dn[NX];
                for (int k = 0; k < NX; ++k) { d[k] = 0.0; dxe_hist[(int64_t)(k * n_ep + j) * B + b] = 0.0; }
                for (int step = 1; step <= K; ++step) {
                    double G[NU], Gp[NU], xp[NX], x[NX];
                    for (int k = 0; k < NU; ++k) { G[k] = gradu_hist[((int64_t)step * NU + k) * B + b]; Gp[k] = gradu_hist[((int64_t)(step - 1) * NU + k) * B + b]; }
                    for (int k = 0; k < NX; ++k) { x[k] = xi_hist[((int64_t)step * NX + k) * B + b]; xp[k] = xi_hist[((int64_t)(step - 1) * NX + k) * B + b]; }
                    direct_column_ep<MK, D, Y, true>(*m, G, Gp, x, xp, ep_index[j], step > 1 ? d : nullptr, dn);
                    for (int k = 0; k < NX; ++k) { dxe_hist[((int64_t)step * NX * n_ep + k * n_ep + j) * B + b] = dn[k]; d[k] = dn[k]; }
                }
            }
        }
    };
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(); });
}
// stage 1 of cm_hessian_history_ep: W[(step-1)*B + pt][NQ][NQ], NQ = 2 NX + 12 + n_ep
int hh_hessian_weights_ep(const cm_model_desc* m, int64_t B, int K, int n_ep, const int32_t* ep_index, const double* gradu_hist,
                          const double* xi_hist, const double* lam_hist, const double* sbar_hist, const double* hss6,
                          const double* hss_hist, const double* hxx_hist, double* W) {
    auto body = [&]<int D, int Y, bool R, int MK>() {
        constexpr int NX = nx_of<D, MK>(), NU = Dims<D>::NU;
        const int NQ = 2 * NX + CM_NUM_PARAMS + n_ep;
        for (int step = 1; step <= K; ++step) for (int64_t pt = 0; pt < B; ++pt) {
            double G[NU], xp[NX], x[NX], lam[NX], sbar[6], hs[6];
            for (int k = 0; k < NU; ++k) {
                G[k] = gradu_hist[((int64_t)step * NU + k) * B + pt];
                if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) G[k] -= gradu_hist[((int64_t)(step - 1) * NU + k) * B + pt];
            }
            for (int k = 0; k < NX; ++k) {
                xp[k] = xi_hist[((int64_t)(step - 1) * NX + k) * B + pt];
                x[k] = xi_hist[((int64_t)step * NX + k) * B + pt];
                lam[k] = lam_hist[((int64_t)step * NX + k) * B + pt];
            }
            for (int r = 0; r < 6; ++r) { sbar[r] = sbar_hist[((int64_t)step * 6 + r) * B + pt]; hs[r] = hss_hist ? hss_hist[step * 6 + r] : hss6[r]; }
            const int64_t ps = (int64_t)(step - 1) * B + pt;
            for (int a = 0; a < NQ; ++a) for (int b = a; b < NQ; ++b) {
                double w = hessian_weight<D, CM_YIELD_ANY, true, MK>(*m, G, x, xp, lam, sbar, hs, a, b, ep_index);
                if (hxx_hist && a == b && a < NX) w += hxx_hist[step * NX + a];
                W[(ps * NQ + a) * NQ + b] = w; W[(ps * NQ + b) * NQ + a] = w;
            }
        }
    };
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_RATE_ELASTIC_PLASTIC>(); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, R, CM_SMALL_ELASTIC_PLASTIC>(); });
}
#endif
#if HH_HAS(5)
// cm_param_blocks (cm::param_direction per point and requested extended parameter)
int hh_param_blocks(const cm_model_desc* m, int64_t B, int n_ep, const int32_t* ep_index, const double* gradu, const double* gradu_prev,
                    const double* xi_prev, const double* xi, double* dC, double* dS) {
    auto body = [&]<int D, int Y, int MK>() {
        constexpr int NX = nx_of<D, MK>(), NU = Dims<D>::NU;
        for (int64_t pt = 0; pt < B; ++pt) for (int j = 0; j < n_ep; ++j) {
            double G[NU], xp[NX], x[NX], oC[NX], oS[6];
            for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + pt] - (gradu_prev ? gradu_prev[k * B + pt] : 0.0);
            for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + pt]; x[k] = xi[k * B + pt]; }
            param_direction<D, CM_YIELD_ANY, MK>(*m, G, x, xp, ep_index[j], oC, oS);
            for (int k = 0; k < NX; ++k) dC[((int64_t)j * NX + k) * B + pt] = oC[k];
            for (int k = 0; k < 6; ++k) dS[((int64_t)j * 6 + k) * B + pt] = oS[k];
        }
    };
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, CM_SMALL_RATE_ELASTIC_PLASTIC>(); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, CM_SMALL_ELASTIC_PLASTIC>(); });
}
// cm_update_complex's per-point body (cm::newton_cx): complex arrays as (2, rows, B)
int hh_update_complex(const cm_model_desc* m, int64_t B, const double* p_im, const double* ext_im, const double* gradu,
                      const double* gradu_prev, const double* xi_prev, double* xi, double* residual, double* sigma, uint32_t* status) {
    auto body = [&]<int D, int Y, int MK>() {
        {
            constexpr int NX = nx_of<D, MK>(), NU = Dims<D>::NU;
            for (int64_t pt = 0; pt < B; ++pt) {
                double G[NU];
                CX xp[NX], x[NX], C[NX], sg[6];
                for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + pt] - (gradu_prev ? gradu_prev[k * B + pt] : 0.0);
                for (int k = 0; k < NX; ++k) {
                    xp[k] = CX{xi_prev[k * B + pt], xi_prev[(NX + k) * B + pt]};
                    x[k] = CX{xi[k * B + pt], xi[(NX + k) * B + pt]};
                }
                const uint32_t st = newton_cx<D, CM_YIELD_ANY, MK>(*m, p_im, ext_im, G, xp, x, C, sg);
                for (int k = 0; k < NX; ++k) { xi[k * B + pt] = x[k].re; xi[(NX + k) * B + pt] = x[k].im; }
                if (residual) for (int k = 0; k < NX; ++k) { residual[k * B + pt] = C[k].re; residual[(NX + k) * B + pt] = C[k].im; }
                if (sigma) for (int k = 0; k < 6; ++k) { sigma[k * B + pt] = sg[k].re; sigma[(6 + k) * B + pt] = sg[k].im; }
                if (status) status[pt] = st;
            }
        }
    };
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, CM_SMALL_RATE_ELASTIC_PLASTIC>(); });
    return dispatch<true>(m, [&]<int D, int Y, bool R>() { body.template operator()<D, Y, CM_SMALL_ELASTIC_PLASTIC>(); });
}
#endif
#if HH_HAS(0)
void hh_log1p_01(int64_t n, const double* x, double* y) { for (int64_t i = 0; i < n; ++i) y[i] = log1p_01(x[i]); }
void hh_log_pos(int64_t n, const double* x, double* y) { for (int64_t i = 0; i < n; ++i) y[i] = log_pos(x[i]); }
// cm::icnn_symmetric on n scaled inputs: F[n], G[n][6], Hx[n][21] (packed upper triangle); w = the device weight pack
void hh_icnn_symmetric(const double* w, int H, int64_t n, const double* xs, double* F, double* G, double* Hx) {
    for (int64_t i = 0; i < n; ++i) icnn_symmetric<true>(w, H, xs + 6 * i, F[i], G + 6 * i, Hx + 21 * i);
}
void hh_soft_unit(int64_t n, const double* x, double* sp, double* sg) {
    for (int64_t i = 0; i < n; ++i) { const SoftUnit u = soft_unit(x[i]); sp[i] = u.sp; sg[i] = u.sg; }
}
#endif
#if HH_HAS(0)
void hh_exp_s(int64_t n, const double* x, double* y) { for (int64_t i = 0; i < n; ++i) y[i] = exp_s(x[i]); }
#endif
#if HH_HAS(0)
void hh_set_dense(int d) { g_dense = d; }
void hh_set_passes(int d) { hh_g_passes = d; }
void hh_set_force_ls(int d) { hh_g_force_ls = d; }
void hh_set_radial_vjp(int d) { hh_g_radial_vjp = d; }
// points that left the J2 subspace iterations for the general path since the last reset (cm::subspace_fallbacks)
long long hh_subspace_fallbacks(int reset) { const long long n = cm::subspace_fallbacks(); if (reset) cm::subspace_fallbacks() = 0; return n; }
#endif
#if HH_HAS(0)
int hh_sizeof_desc(void) { return (int)sizeof(cm_model_desc); }
#endif
}
