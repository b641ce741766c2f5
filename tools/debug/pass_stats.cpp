// Diagnostic (not part of the product or the tests): per-point pass / iteration counts of the resumable Newton
// (cm_pool.hpp) on a batch, host build.  g++ -O2 -std=c++20 -DCM_HOST_BUILD -shared -fPIC tools/debug/pass_stats.cpp -o /tmp/libpass_stats.so
#define CM_HOST_BUILD 1
#include "../../cmad_amd/csrc/cm_pool.hpp"
using namespace cm;

template <int DEF, int YK>
static void run(const cm_model_desc& m, int64_t B, const double* gradu, const double* xi_prev, int* passes, int* iters, int* trials) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    for (int64_t b = 0; b < B; ++b) {
        double G[NU], xp[NX], x[NX], eg[6], z[Dims<DEF>::NZ];
        for (int k = 0; k < NU; ++k) G[k] = gradu[k * B + b];
        for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[k * B + b]; x[k] = xp[k]; }
        strain_from_gradu<DEF, false>(m, G, eg);
        strain_z<DEF, false>(m, z);
        double parked[2 * 9];
        const LaneStage stage{parked, 1};
        PassState s; pass_reset(s);
        bool running = true;
        int np = 0, nt = 0;
        while (running) {
            if (s.phase == CM_PH_TRIAL) ++nt;
            newton_pass<DEF, YK, CM_SMALL_ELASTIC_PLASTIC, true>(m, eg, z, xp, x, s, running, stage);
            ++np;
        }
        passes[b] = np; iters[b] = s.it; trials[b] = nt;
    }
}
extern "C" int ps_stats(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, int* passes, int* iters, int* trials) {
    if (m->yield_kind == CM_YIELD_HOSFORD) run<CM_FULL_3D, CM_YIELD_HOSFORD>(*m, B, gradu, xi_prev, passes, iters, trials);
    else if (m->yield_kind == CM_YIELD_HYBRID_HILL_NN) run<CM_FULL_3D, CM_YIELD_HYBRID_HILL_NN>(*m, B, gradu, xi_prev, passes, iters, trials);
    else if (m->yield_kind == CM_YIELD_HILL) run<CM_FULL_3D, CM_YIELD_HILL>(*m, B, gradu, xi_prev, passes, iters, trials);
    else return -1;
    return 0;
}
