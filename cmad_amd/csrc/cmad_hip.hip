// Kernels and C-ABI of the batched constitutive-model evaluator (see include/cmad_hip.h).
// gfx950 only.  One Gauss point per lane; SoA arrays so that lane b of a wavefront reads
// element [k*B + b] -> every global access is a 512-byte contiguous row per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
// Two builds of this file make the library (cmad_amd/build.py): the BASE build (CM_HNN_VARIANT = 0) without the network hardening
// law -- so that law costs the Voce / linear configurations nothing, not an instruction and not a register (inlined into every
// Newton loop it cost the fused J2 objective and PLANE_STRESS kernels 4 %, as a call 15-20 % more VALU instructions per
// wavefront: profiles/r03_hnn_ab.txt) -- and the HNN build (CM_HNN_VARIANT = 1) with it, for J2 / Hill / Hosford.  Every launching
// entry point is defined under a suffixed name (cm_update_base / cm_update_hnn: the renames of cm_entries.inc); the public symbols
// of include/cmad_hip.h are wrappers that pick the build from cm_model_desc.hnn_width (bottom of this file, compiled once).
#ifndef CM_HNN_VARIANT
#define CM_HNN_VARIANT 0
#endif
#define CM_HNN CM_HNN_VARIANT
#define CM_ENTRIES_RENAME
#include "cm_entries.inc"
#undef CM_ENTRIES_RENAME
#include "cm_pool.hpp"
#include "cm_hessian.hpp"

// The library can be built from this one file in twelve independent pieces (hipcc -DCM_PART=0..11, see
// cmad_amd/build.py) so the template instantiations compile in parallel; without CM_PART everything is one TU.
//   9: cm_param_blocks, cm_param_adjoint_history (extended parameter sensitivities)
//   0: cm_update            2: cm_update_vjp, cm_adjoint_step   4: cm_update_tangent          6: cm_objective_grad,
//   1: cm_update_rate, info 3: cm_update_and_vjp                5: cm_evaluate(_rate)            cm_hessians
#ifndef CM_PART
#define CM_PART (-1)
#endif
#define CM_HAS_PART(k) (CM_PART == -1 || CM_PART == (k))

extern int g_cm_last_hip_error;
#if CM_HAS_PART(1) && !CM_HNN_VARIANT
int g_cm_last_hip_error = 0;
#endif

// the deterministic reductions (k_reduce_stage1/2, k_sum_rows) exist once in the library (part 1 of the base build); every
// part launches them through these two helpers
__attribute__((visibility("hidden"))) void cm_detail_reduce(hipStream_t s, const double* partials, int64_t nrows, double* stage,
                                                            double* out, int first, int accumulate);
__attribute__((visibility("hidden"))) void cm_detail_sum_rows(hipStream_t s, const double* part, int64_t nrows, int ncols, double* out);

namespace {

// 128 lanes (two wavefronts) per workgroup: measured against 64 and 256 on the same box (tools/ab_multi.sh): -3 % on the
// fused update + vjp kernel against 256 (finer-grained release of wave slots and LDS), within +-1.5 % elsewhere; 64 is slower.
#ifndef CM_BLOCK
#define CM_BLOCK 128
#endif
constexpr int kBlock = CM_BLOCK;      // threads per workgroup of the per-point kernels
constexpr int kRBlock = 256;          // threads per workgroup of the reduction kernels
constexpr int kRed = 1 + CM_NUM_PARAMS;

using namespace cm;

// SoA row access.  `p` is the block's base (array + blockIdx.x * kBlock, wave-uniform -> SGPR pair) and `t`
// the lane's offset inside the block, so each access is `global_load/store v, v_off, s[base]` with the row
// stride added on the scalar unit -- no per-lane 64-bit address arithmetic.
// Row k of a block's slice is the wave-uniform base p + k * B plus a 32-bit lane offset.  Each base is pinned to
// an SGPR pair (empty asm with an "s" constraint, in the global address space) so that every access is the
// `global_load/store v, v_lane_offset, s[base:base+1]` form: one shared VGPR offset, scalar-unit address arithmetic.
// Left to itself the compiler re-associates the address into (p + lane) + k * B and spends a 64-bit VALU add and a
// VGPR pair per row.
// Non-temporal loads and stores for the single-pass kernels: every input is read once and every output written once, and
// streaming hints move the 35-row pattern of update + vjp closer to the hardware's streaming rate (tools/soa_stream_bench.hip:
// 6.35 against 5.93 TB/s; +1-3 % on the kernels).  The history kernels re-read the states they wrote and stay temporal.
#ifndef CM_NONTEMPORAL
#define CM_NONTEMPORAL 1
#endif
typedef const __attribute__((address_space(1))) char* cm_gcptr;
typedef __attribute__((address_space(1))) char* cm_gptr;
template <int N, bool NT = (CM_NONTEMPORAL != 0)>
__device__ __forceinline__ void load_soa(const double* __restrict__ p, int64_t B, unsigned t, double* out) {
    const uint32_t off = t * 8u;                  // t < kBlock
#pragma unroll
    for (int k = 0; k < N; ++k) {
        cm_gcptr row = (cm_gcptr)(p + (int64_t)k * B);
        asm volatile("" : "+s"(row));
        if constexpr (NT) out[k] = __builtin_nontemporal_load((const __attribute__((address_space(1))) double*)(row + off));
        else out[k] = *(const __attribute__((address_space(1))) double*)(row + off);
    }
}
template <int N, bool NT = (CM_NONTEMPORAL != 0)>
__device__ __forceinline__ void store_soa(double* __restrict__ p, int64_t B, unsigned t, const double* v) {
    const uint32_t off = t * 8u;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        cm_gptr row = (cm_gptr)(p + (int64_t)k * B);
        asm volatile("" : "+s"(row));
        if constexpr (NT) __builtin_nontemporal_store(v[k], (__attribute__((address_space(1))) double*)(row + off));
        else *(__attribute__((address_space(1))) double*)(row + off) = v[k];
    }
}

// minimum waves per SIMD requested from the register allocator (512 VGPRs per lane-slot / waves, in steps of 8:
// 4 waves <= 128, 3 waves <= 168, 2 waves <= 256).  Experiment knobs: -DCM_OCC_<name>=<waves>.
#ifndef CM_OCC_UPD_J2
#define CM_OCC_UPD_J2 1
#endif
#ifndef CM_OCC_UPD_J2_LS
#define CM_OCC_UPD_J2_LS 1
#endif
#ifndef CM_OCC_UPD_HOSFORD
#define CM_OCC_UPD_HOSFORD 1
#endif
#ifndef CM_OCC_UPD_HOSFORD_LS
#define CM_OCC_UPD_HOSFORD_LS 1
#endif
#ifndef CM_OCC_UPD_HYBRID
#define CM_OCC_UPD_HYBRID 1         // lockstep fallback (B < 256 or CM_SOLVER_LOCKSTEP): registers instead of scratch; the pool kernel is the fast path
#endif
#ifndef CM_OCC_UPD_BARLAT
#define CM_OCC_UPD_BARLAT 1
#endif
#ifndef CM_OCC_REV_HILL
#define CM_OCC_REV_HILL 1
#endif
#ifndef CM_OCC_UNIAXIAL
#define CM_OCC_UNIAXIAL 2          // UNIAXIAL_STRESS update / objective kernels on the quadratic surfaces: 302 -> 256 VGPRs + 136 B scratch in the 9x9 Newton step no lane takes since the 1-d return map; 0.150 -> 0.122 ms per 2e6 points (profiles/r04_occupancy_ab.txt)
#endif
#ifndef CM_OCC_PS_J2_PLANE
#define CM_OCC_PS_J2_PLANE 1       // J2 / PLANE_STRESS kernels on the plane iteration (newton_j2_plane), plain Newton
#endif
#ifndef CM_OCC_PS_J2_PLANE_LS
#define CM_OCC_PS_J2_PLANE_LS 3    // ... with the line search: 182 -> 168 VGPRs (40 B scratch), 0.76 instead of 0.79 ms per 1e7 points (update + vjp)
#endif
#ifndef CM_OCC_REV_J2
#define CM_OCC_REV_J2 4             // fused J2 / FULL_3D plain-Newton kernels (headline, objective)
#endif
#ifndef CM_OCC_REV_J2_LS
#define CM_OCC_REV_J2_LS 3          // 172 -> 168 VGPRs (16 B scratch): 0.85 ms instead of 0.90 ms per 1e7 points
#endif
constexpr int kLsSlots = 2 * 8;         // line search: parked iterate and direction per lane, 2 * max NX (cm_structured.hpp)

template <int DEF, int YK, bool LS, bool TANGENT, bool RL = false>
constexpr int min_waves_update() {
    if (DEF == CM_PLANE_STRESS && YK == CM_YIELD_J2 && RL && !TANGENT) return LS ? CM_OCC_PS_J2_PLANE_LS : CM_OCC_PS_J2_PLANE;
    if (DEF == CM_UNIAXIAL_STRESS && (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL) && !TANGENT) return CM_OCC_UNIAXIAL;
    if (DEF != CM_FULL_3D) return 1;
    if (YK == CM_YIELD_J2) return LS ? CM_OCC_UPD_J2_LS : CM_OCC_UPD_J2;
    if (YK == CM_YIELD_HOSFORD) return LS ? CM_OCC_UPD_HOSFORD_LS : CM_OCC_UPD_HOSFORD;
    if (is_nn_yield(YK)) return CM_OCC_UPD_HYBRID;
    if (YK == CM_YIELD_BARLAT) return CM_OCC_UPD_BARLAT;
    return 1;
}

// ---- cm_update / cm_update_tangent ----------------------------------------------------------------
template <int DEF, int YK, bool ROT, bool LS, bool TANGENT, bool RL = false>
__global__ __launch_bounds__(kBlock, (min_waves_update<DEF, YK, LS, TANGENT, RL>())) void k_update(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ xi_prev,
        double* __restrict__ xi, double* __restrict__ sigma, double* __restrict__ dsig, uint32_t* __restrict__ status) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    const int64_t blk0 = (int64_t)blockIdx.x * kBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);   // tail lanes shadow the last point, never store
    gradu += blk0; xi_prev += blk0; xi += blk0;
    if (sigma) sigma += blk0;
    if (status) status += blk0;
    if (dsig) dsig += blk0;
    double G[NU], xp[NX], x[NX], eg[6], z[Dims<DEF>::NZ];
    load_soa<NU>(gradu, B, b, G);
    load_soa<NX>(xi_prev, B, b, xp);
    strain_from_gradu<DEF, ROT>(m, G, eg);
    strain_z<DEF, ROT>(m, z);
    __shared__ double ls_stage[(LS && has_structured<DEF, YK>()) ? kLsSlots * kBlock : 1];   // parked iterate + direction
    uint32_t st = newton_any<DEF, YK, LS, true, RL>(m, eg, z, xp, x, valid, lane_stage(ls_stage, LS ? threadIdx.x : 0, kBlock));
    Eval<DEF> ev;
    strain_stress<DEF>(m, eg, z, x, ev);
    if (status) {
        // branch at the returned state (informational)
        double phi, gt[6], Ht[6][6];
        yield_eval<YK, false>(m, ev.s, phi, gt, Ht);
        const double f = (phi - (m.Y + hardening(m, x[6]).H)) * 0.5 / m.mu;
        if ((f > m.yield_tol) || (fabs(f) < m.yield_tol)) st |= CM_STATUS_PLASTIC;
    }
    double sg[6];
    to_global<ROT>(m, ev.s, sg);
    if (valid) {
        store_soa<NX>(xi, B, b, x);
        if (sigma) store_soa<6>(sigma, B, b, sg);
        if (status) status[b] = st;
    }
    if constexpr (TANGENT) {
        double T[6][6];
        const bool ok = tangent_any<DEF, YK>(m, eg, z, x, xp, T);
        if (!ok && valid && status) status[b] = st | CM_STATUS_SINGULAR;
        // d sig_g / d G_c = Rg T Rm dE/dG_c : push each unit grad-u direction through
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            double Gd[NU], dm[6], t[6], tg[6];
#pragma unroll
            for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
            strain_from_gradu<DEF, ROT>(m, Gd, dm);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < 6; ++l) s += T[r][l] * dm[l];
                t[r] = s;
            }
            to_global<ROT>(m, t, tg);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 6; ++r) (dsig + (int64_t)(r * NU + c) * B)[b] = tg[r];
            }
        }
    }
}

// ---- cm_update on a work pool: every lane iterates, no lane waits for the slowest point of its wavefront ---------------
// One wavefront per workgroup, persistent: wavefront w owns the chunks w, w + W, w + 2W, ... of kPoolChunk consecutive
// points (W = wavefronts in the grid; static assignment -- no atomics, no workspace, results independent of scheduling).
// Each lane runs the resumable Newton of cm_pool.hpp on its current point; when kPoolRefill lanes of the wavefront have
// finished theirs (or nothing is left to hand out), those lanes store their results and take the next consecutive points
// of the wavefront's stream.  Used for the iteration-bound configurations (pool_pays<> / pool_route() below: the network
// surfaces, and Hosford under the line search): a lockstep wavefront runs max(iterations) over its 64 points there.
//
// Input staging (round 3).  With ~4 passes per point a quarter of the lanes finish in EVERY pass, so a refill that loads its
// 16 input rows from global memory puts an HBM round trip (~2 us) on the critical path of every pass (round-2 counters:
// 43 % of the wave-cycles waiting, 28 % issuing VALU).  The wavefront's point stream is therefore staged through LDS in halves
// of 32 consecutive points: an LDS-DMA copy (global_load_lds: coalesced 16-byte pieces straight into LDS, no registers) of the
// half after next is issued as soon as a half has been handed out, and lands while the lanes iterate; a refill is 16 ds_reads.
// Ring: 2 halves x (n_gradu + n_xi) rows x 32 points (8 KB under FULL_3D).  `wide` = 16-byte pieces (B even and both arrays
// 16-byte aligned), else 4-byte pieces (any alignment a double array has).
#ifndef CM_POOL_WAVES_HOSFORD
#define CM_POOL_WAVES_HOSFORD 1
#endif
#ifndef CM_POOL_WAVES_NN
#define CM_POOL_WAVES_NN 2
#endif
// idle lanes that trigger a retire / refill step; measured (profiles/r03_pool_refill_ab.txt): 16 for Hosford (8 and 4 equal within
// noise, 32: -8 %), 8 for the network surfaces (+3 % over 16; 32: -15 %)
#ifndef CM_POOL_REFILL
#define CM_POOL_REFILL 16
#endif
#ifndef CM_POOL_REFILL_NN
#define CM_POOL_REFILL_NN 8
#endif
constexpr int kPoolHalf = 32;            // points per staged half (one LDS-DMA instruction moves 4 rows of it in 16-byte pieces)

// Counters of the dynamic chunk assignment.  Every launch that can be in flight at the same time as another one must draw from
// its own counter, and a launch captured into a HIP graph keeps the counter it was captured with for every replay:
//   * eager launches: one counter per STREAM (launches on one stream are ordered, so they can share; a host-side table maps the
//     stream handle to its slot -- kPoolStreamSlots streams, after which a new stream's launches use the static assignment);
//   * captured launches (hipStreamIsCapturing): a private counter per captured launch, never handed out again
//     (kPoolCaptureSlots of them per build of this file, after which captures use the static assignment) -- a replayed graph can
//     therefore overlap eager launches on any stream and other graphs.  (One graph executable cannot run concurrently with
//     itself; instantiating the same captured graph twice and running both at once is the one combination left out.)
// The counter is zeroed by a one-thread kernel on the launch's stream right before the pool kernel (a kernel node under graph
// capture; no host-side address, no runtime call that a capturing stream could object to).
#ifndef CM_POOL_DYNAMIC
#define CM_POOL_DYNAMIC 1
#endif
#ifndef CM_POOL_DYNAMIC_MIN
#define CM_POOL_DYNAMIC_MIN 2048       // points per resident wavefront from which the chunks are drawn dynamically
#endif
constexpr int kPoolStreamSlots = 64, kPoolCaptureSlots = 4032;
constexpr int kPoolTicketSlots = kPoolStreamSlots + kPoolCaptureSlots;
__device__ unsigned long long g_pool_ticket[kPoolTicketSlots];
__global__ void k_pool_ticket_zero(int slot);
#if CM_HAS_PART(0)                       // launched by cm_update only: one copy per build
__global__ void k_pool_ticket_zero(int slot) { g_pool_ticket[slot] = 0ull; }
#endif
// the launch's counter, or -1: static assignment (always correct, a few per cent slower on the largest batches)
inline int pool_ticket_slot(hipStream_t s) {
    static std::mutex mu;
    static std::unordered_map<uintptr_t, int> by_stream;        // (device, stream handle) -> slot
    static int next_capture = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s != nullptr && hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return -1; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    // CM_DEBUG_POOL_SLOTS=<n>: shrink both tables (tests force the out-of-slots fallback with it)
    static const int limit = [] { const char* e = getenv("CM_DEBUG_POOL_SLOTS"); return e ? atoi(e) : -1; }();
    const int stream_slots = (limit >= 0 && limit < kPoolStreamSlots) ? limit : kPoolStreamSlots;
    const int capture_slots = (limit >= 0 && limit < kPoolCaptureSlots) ? limit : kPoolCaptureSlots;
    std::lock_guard<std::mutex> lock(mu);
    if (cs != hipStreamCaptureStatusNone) return (next_capture < capture_slots) ? kPoolStreamSlots + next_capture++ : -1;
    const uintptr_t key = (uintptr_t)s ^ ((uintptr_t)(unsigned)dev << 56);
    const auto it = by_stream.find(key);
    if (it != by_stream.end()) return it->second;
    if ((int)by_stream.size() >= stream_slots) return -1;
    const int slot = (int)by_stream.size();
    by_stream.emplace(key, slot);
    return slot;
}

template <int DEF, int YK>
constexpr int min_waves_pool() { return is_dense_yield(YK) ? CM_POOL_WAVES_NN : CM_POOL_WAVES_HOSFORD; }
// Where the pool pays (measured, profiles/r02_pool_ab.txt): a pass must cost much more than the retire / refill bookkeeping and
// the per-lane addressing -- the network surfaces (1.3x on top of the structured 6x6 solve) and Hosford under the line search
// (a = 100 with the notch deck's settings: 2.2x).  J2 / Hill lose 5-30 %, Barlat (spills in the pool kernel) 10 %: lockstep.
#ifndef CM_POOL_HILL
#define CM_POOL_HILL 0              // experiment knob: Hill (plain Newton and line search) on the work pool as well
#endif
template <int YK, bool LS>
constexpr bool pool_pays() { return is_nn_yield(YK) || (YK == CM_YIELD_HOSFORD && LS) || (CM_POOL_HILL && YK == CM_YIELD_HILL); }

typedef __attribute__((address_space(3))) double cm_lds_double;
typedef __attribute__((address_space(3))) void* cm_lds_vptr;
typedef const __attribute__((address_space(1))) void* cm_gvptr;

template <int DEF, int YK, bool ROT, bool LS>
__global__ __launch_bounds__(64, (min_waves_pool<DEF, YK>())) void k_update_pool(cm_model_desc m, int64_t B, int chunk_shift, int wide, int ticket_slot,
        const double* __restrict__ gradu, const double* __restrict__ xi_prev,
        double* __restrict__ xi, double* __restrict__ sigma, uint32_t* __restrict__ status) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU, NIN = NU + NX;
    constexpr int kRefill = is_dense_yield(YK) ? CM_POOL_REFILL_NN : CM_POOL_REFILL;
    __shared__ double ring[2 * NIN * kPoolHalf];                 // [half & 1][row][32 points]
    __shared__ double ls_stage[LS ? 2 * NX * 64 : 1];
    const LaneStage stage = lane_stage(ls_stage, LS ? threadIdx.x : 0, 64);
    volatile cm_lds_double* const ringl = (volatile cm_lds_double*)ring;
    const unsigned lane = threadIdx.x;
    const int64_t nwaves = gridDim.x;
    // The wavefront's stream in halves of 32 points.  Static assignment (ticket == nullptr; small batches): half h lies in chunk
    // (blockIdx.x + (h >> hshift) * nwaves) at offset (h & hmask) * 32; chunks hold 2^chunk_shift points (a multiple of 32).
    // Dynamic assignment (large batches): every half is the next 32 points of the batch, taken from a device counter when its
    // copy is issued -- wavefronts that run faster (or drew cheaper points) simply take more halves, so the grid drains together
    // instead of waiting for its slowest static share (round-3 counters: the mean wavefront lived 77-82 % of the kernel).  A
    // point's result does not depend on which lane computes it, so the output is the same either way.
    const int hshift = chunk_shift - 5, hmask = (1 << hshift) - 1;
    unsigned long long* const ticket = (ticket_slot >= 0) ? &g_pool_ticket[ticket_slot] : nullptr;     // wave-uniform
    volatile __shared__ long long hbase[4];                      // first point of the halves in flight (index: half & 3)
#define CM_HALF_BASE_STATIC(h) ((((int64_t)blockIdx.x + (int64_t)((h) >> hshift) * nwaves) << chunk_shift) + (int64_t)(((h) & hmask) * kPoolHalf))
    int next_issue = 0, ready = 0, cons_half = 0, cons_off = 0;  // wave-uniform cursors: issued / arrived / handed out
    bool stream_end = false;
    // issue the LDS-DMA copies of every half whose ring slot is free (at most two halves ahead of the consumer)
#define CM_TRY_ISSUE() \
    while (!stream_end && next_issue < cons_half + 2) { \
        int64_t g0_; \
        if (ticket) { \
            /* the next chunk of 2^chunk_shift points of the batch, drawn when its first half is issued.  A SCALAR atomic: one */ \
            /* request per wavefront, the result in SGPRs, and its wait (lgkmcnt) does not drain the vector memory queue.  One */ \
            /* ticket per CHUNK, not per half: all wavefronts hit one address, and the device serves ~80 M same-address atomics */ \
            /* per second -- 312 500 tickets of 32 points took 3.9 ms by themselves (profiles/r03_pool_dynamic_ab.txt) */ \
            if ((next_issue & hmask) == 0) { \
                unsigned long long r_, inc_ = 1ull << chunk_shift; \
                asm volatile("s_atomic_add_x2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) : "s"(ticket), "0"(inc_) : "memory"); \
                dyn_base = (int64_t)r_; \
            } \
            g0_ = dyn_base + (int64_t)((next_issue & hmask) * kPoolHalf); \
        } else g0_ = CM_HALF_BASE_STATIC(next_issue); \
        if (g0_ >= B) { stream_end = true; break; } \
        if (lane == 0) hbase[next_issue & 3] = g0_; \
        const unsigned dst_ = (unsigned)__builtin_amdgcn_readfirstlane((next_issue & 1) * NIN * kPoolHalf); \
        if (wide) { \
            int64_t pt_ = g0_ + 2 * (lane & 15); \
            if (pt_ > B - 2) pt_ = B - 2;                       /* pairs past the end re-read the last pair; never handed out */ \
            _Pragma("unroll") for (int i_ = 0; i_ < (NIN + 3) / 4; ++i_) { \
                const int r_ = 4 * i_ + (int)(lane >> 4); \
                if (r_ < NIN) { \
                    const double* src_ = (r_ < NU) ? gradu + (int64_t)r_ * B + pt_ : xi_prev + (int64_t)(r_ - NU) * B + pt_; \
                    __builtin_amdgcn_global_load_lds((cm_gvptr)src_, (cm_lds_vptr)((cm_lds_double*)ring + dst_ + 4 * i_ * kPoolHalf), 16, 0, 0); \
                } \
            } \
        } else { \
            int64_t pt_ = g0_ + (lane >> 1); \
            if (pt_ > B - 1) pt_ = B - 1; \
            _Pragma("unroll") for (int r_ = 0; r_ < NIN; ++r_) { \
                const double* src_ = (r_ < NU) ? gradu + (int64_t)r_ * B + pt_ : xi_prev + (int64_t)(r_ - NU) * B + pt_; \
                __builtin_amdgcn_global_load_lds((cm_gvptr)((const char*)src_ + 4 * (lane & 1)), \
                                                 (cm_lds_vptr)((cm_lds_double*)ring + dst_ + r_ * kPoolHalf), 4, 0, 0); \
            } \
        } \
        ++next_issue; \
    }
    double x[NX], xp[NX], eg[6], z[Dims<DEF>::NZ];
    strain_z<DEF, ROT>(m, z);
#pragma unroll
    for (int k = 0; k < NX; ++k) { x[k] = 0.0; xp[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 6; ++k) eg[k] = 0.0;
    int64_t pt = -1;
    bool running = false;
    PassState st;
    pass_reset(st);
    int64_t dyn_base = 0;                                        // wave-uniform: first point of the chunk being issued (dynamic assignment)
    CM_TRY_ISSUE()
    for (;;) {
        const uint64_t idle_mask = __ballot(!running);
        const int nidle = __popcll(idle_mask);
        const bool more = cons_half < next_issue;                // uniform: issued points not yet handed out
        if (nidle == 64 || (more && nidle >= kRefill)) {
            // Order matters for what the wave waits on.  (1) Confirm the copies issued at EARLIER steps (a pass or more ago: they
            // have landed, and so have the stores of the previous retire -- the wait is free), (2) retire, (3) hand out points of
            // confirmed halves only, (4) issue the copies for the ring slots this step freed.  Nothing waits for a memory
            // operation issued in this same step (round 3's first version waited vmcnt(0) behind the retire's stores).
#ifndef CM_POOL_LAZY_CONFIRM
#define CM_POOL_LAZY_CONFIRM 0          // experiment knob: confirm only when the confirmed halves cannot serve every idle lane
#endif
            const int conf_avail = (ready - cons_half) * kPoolHalf - cons_off;      // points of confirmed halves not handed out yet
            if (ready < next_issue && (!CM_POOL_LAZY_CONFIRM || conf_avail < nidle)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ready = next_issue;
            }
            if (!running && pt >= 0) {                           // retire: state, stress and status of the finished point
                Eval<DEF> ev;
                strain_stress<DEF>(m, eg, z, x, ev);
                double sg[6];
                to_global<ROT>(m, ev.s, sg);
#pragma unroll
                for (int k = 0; k < NX; ++k) (xi + (int64_t)k * B)[pt] = x[k];
                if (sigma) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) (sigma + (int64_t)k * B)[pt] = sg[k];
                }
                if (status) {
                    uint32_t sw = st.flags | (uint32_t)st.it;
                    double phi, gt[6], Ht[6][6];
                    yield_eval<YK, false>(m, ev.s, phi, gt, Ht);
                    const double f = (phi - (m.Y + hardening(m, x[6]).H)) * 0.5 / m.mu;
                    if ((f > m.yield_tol) || (fabs(f) < m.yield_tol)) sw |= CM_STATUS_PLASTIC;
                    status[pt] = sw;
                }
                pt = -1;
            }
            if (more) {                                          // refill: consecutive points to the idle lanes, in lane order
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                int need = nidle, first = 0, slot = -1;
                while (need > 0 && cons_half < ready) {          // uniform; confirmed halves only (at most two per refill)
                    const long long hb = hbase[cons_half & 3];     // written when the half was issued (same wavefront: in order)
                    const int64_t g0 = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)hb >> 32)) << 32) |
                                                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)hb & 0xffffffffull)));
                    const int vc = (B - g0 < kPoolHalf) ? (int)(B - g0) : kPoolHalf;     // >= 1: only halves that start below B are issued
                    const int avail = vc - cons_off;
                    const int take = need < avail ? need : avail;
                    if (!running && slot < 0 && rank >= first && rank < first + take) {
                        const int sl = cons_off + (rank - first);
                        pt = g0 + sl;
                        slot = (cons_half & 1) * NIN * kPoolHalf + sl;
                    }
                    cons_off += take; first += take; need -= take;
                    if (cons_off == vc) { ++cons_half; cons_off = 0; }
                }
                if (!running && slot >= 0) {
                    double G[NU];
#pragma unroll
                    for (int k = 0; k < NU; ++k) G[k] = ringl[slot + k * kPoolHalf];
#pragma unroll
                    for (int k = 0; k < NX; ++k) { xp[k] = ringl[slot + (NU + k) * kPoolHalf]; x[k] = xp[k]; }
                    strain_from_gradu<DEF, ROT>(m, G, eg);
                    pass_reset(st);
                    running = true;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the ring reads are complete before a slot is refilled
                CM_TRY_ISSUE()
            }
        }
        if (!__any(running)) break;          // nothing running => nothing left: a step with 64 idle lanes confirms and hands out >= 1 point
        newton_pass<DEF, YK, CM_SMALL_ELASTIC_PLASTIC, LS>(m, eg, z, xp, x, st, running, stage);
    }
#undef CM_TRY_ISSUE
#undef CM_HALF_BASE_STATIC
}

// ---- cm_update, screened: elastic points finish in a streaming kernel, the plastic ones are listed and solved densely -----------
// FULL_3D only (an elastic trial state IS the answer there: C_e(x_prev) = 0 exactly, 0 iterations).  For the surfaces whose
// residual evaluation costs thousands of instructions -- the network surfaces, Barlat, Hosford on the reference's iteration --
// a lockstep wavefront drags its elastic lanes (about half of a typical batch) through every evaluation of its plastic ones,
// and the work pool pays for its refill machinery on every pass.  With a caller-provided workspace of 4 B per point:
//   k_screen        one point per lane, coalesced: trial stress, effective-stress VALUE (no normal, no Hessian), f0.  Elastic:
//                   state, stress and status are final and stored here.  Plastic: the point's index is appended to a list
//                   (one atomic per wavefront; ballot / mbcnt ranks inside it).
//   k_update_listed lockstep Newton over the list: every lane of every wavefront holds a plastic point.
// A point's result does not depend on its position in the list, so the output is independent of the order the wavefronts of
// k_screen append in.  Costs: the inputs of the plastic points are read twice and their outputs are written by a second kernel
// (partial rows): about 2x the algorithmic HBM bytes on kernels that run at 5-20 % of the HBM roof.
#ifndef CM_SCREEN
#define CM_SCREEN 1
#endif
template <int YK>
constexpr bool screen_pays() { return CM_SCREEN != 0 && is_dense_yield(YK); }

// 1024 lanes per workgroup: the append is ONE atomic per workgroup (wave counts summed through LDS).  All wavefronts of the grid
// add to the same address, and the device serves about 80 M same-address atomics per second: one per wavefront -- 78 000 for
// 5 x 10^6 points -- took 0.96 ms by itself (profiles/r04_screen_atomics.txt); one per 1024 points takes 0.06 ms.
#ifndef CM_SCREEN_BLOCK
#define CM_SCREEN_BLOCK 1024
#endif
#ifndef CM_SCREEN_NT
#define CM_SCREEN_NT 1              // non-temporal input loads in k_screen: the batch's rows (640 MB at 5e6 points) do not survive in any
                                    // cache until k_update_listed reads them again; 1.7 % on hybrid_update (profiles/r04_occupancy_ab.txt)
#endif
constexpr int kScreenBlock = CM_SCREEN_BLOCK;
template <int YK, bool ROT>
__global__ __launch_bounds__(kScreenBlock) void k_screen(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ xi_prev,
        double* __restrict__ xi, double* __restrict__ sigma, uint32_t* __restrict__ status,
        uint32_t* __restrict__ list, unsigned long long* __restrict__ count) {
    const int64_t blk0 = (int64_t)blockIdx.x * kScreenBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);
    gradu += blk0; xi_prev += blk0; xi += blk0;
    if (sigma) sigma += blk0;
    if (status) status += blk0;
    double G[9], xp[7], eg[6], z[6];
    load_soa<9, CM_SCREEN_NT != 0>(gradu, B, b, G);
    load_soa<7, CM_SCREEN_NT != 0>(xi_prev, B, b, xp);
    strain_from_gradu<CM_FULL_3D, ROT>(m, G, eg);
    strain_z<CM_FULL_3D, ROT>(m, z);
    Eval<CM_FULL_3D> ev;
    strain_stress<CM_FULL_3D>(m, eg, z, xp, ev);                 // trial stress: what k_update stores for an elastic point
    const double* s = ev.s;
    double phi, gt[6], Hd[1];
    const double YH = m.Y + hardening(m, xp[6]).H;
    bool exact = true;
    if constexpr (YK == CM_YIELD_HYBRID_HILL_NN) {
        // One hidden layer: the network term in float first (icnn_value_f: a third of this kernel's instructions less).  Its
        // error is ~1e-6 of the terms it sums; a point within 2e-4 of (|phi_hill| + |N| + Y + H) of the surface -- one in a few
        // thousand -- sends its wavefront through the double evaluation, which then decides for every lane.
        if (m.nn_nlayers == 3) {                                  // uniform
            double ph;
            yield_eval_p<CM_YIELD_HILL, false>(m, s, ph, gt, Hd);
            const double nf = icnn_value_f(m, s);
            phi = ph + nf;
            const bool sure = fabs(phi - YH) > 2e-4 * (fabs(ph) + fabs(nf) + YH);      // (false for a NaN)
            exact = __any(valid && !sure);
        }
    }
    if (exact) yield_eval_p<YK, false>(m, s, phi, gt, Hd);       // the value residual_s / residual see at x_prev
    const double f0 = (phi - YH) * half_over_mu(m);
    const bool plastic0 = (f0 > m.yield_tol) || (fabs(f0) < m.yield_tol);
    if (valid && !plastic0) {                                    // cond_residual's elastic branch at x_prev: C = 0, nothing to iterate
        double sg[6];
        to_global<ROT>(m, s, sg);
        store_soa<7>(xi, B, b, xp);
        if (sigma) store_soa<6>(sigma, B, b, sg);
        if (status) status[b] = CM_STATUS_CONVERGED;
    }
    // append the workgroup's plastic points: ranks from the wave ballots, wave offsets through LDS, one atomic for the workgroup
    __shared__ unsigned wave_n[kScreenBlock / 64];
    __shared__ unsigned long long block_base;
    const bool take = valid && plastic0;
    const uint64_t mask = __ballot(take);
    const unsigned w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) wave_n[w] = (unsigned)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
#pragma unroll
        for (int k = 0; k < kScreenBlock / 64; ++k) total += wave_n[k];
        block_base = total ? atomicAdd(count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    if (take) {
        unsigned before = 0;
        for (unsigned k = 0; k < w; ++k) before += wave_n[k];     // uniform per wavefront
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        const unsigned long long at = block_base + before + (unsigned long long)rank;
        if (at < (unsigned long long)B) list[at] = (uint32_t)(blk0 + threadIdx.x);        // (at < B always, for a counter that started at 0)
    }
}
// the list's length starts at zero: a one-thread kernel on the launch's stream (a kernel node under graph capture, like the work
// pool's ticket -- no runtime call that a capturing stream could treat differently)
__global__ void k_screen_reset(unsigned long long* count);
#if CM_HAS_PART(0)
__global__ void k_screen_reset(unsigned long long* count) { *count = 0ull; }
#endif

template <int YK, bool ROT, bool LS>
constexpr int min_waves_listed() { return min_waves_update<CM_FULL_3D, YK, LS, false, false>(); }

// the lockstep update of k_update over a list of points (FULL_3D); rows are addressed by point index (B * 8 < 2^32)
template <int YK, bool ROT, bool LS>
__global__ __launch_bounds__(kBlock, (min_waves_listed<YK, ROT, LS>())) void k_update_listed(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ xi_prev,
        double* __restrict__ xi, double* __restrict__ sigma, uint32_t* __restrict__ status,
        const uint32_t* __restrict__ list, const unsigned long long* __restrict__ count) {
    constexpr int NX = 7, NU = 9;
    const unsigned long long nraw = *count;                      // uniform
    const unsigned long long n = nraw < (unsigned long long)B ? nraw : (unsigned long long)B;
    const unsigned long long t0 = (unsigned long long)blockIdx.x * kBlock;
    if (t0 >= n) return;
    const bool valid = t0 + threadIdx.x < n;
    const uint32_t pt = list[valid ? t0 + threadIdx.x : n - 1];  // tail lanes shadow the last listed point, never store
    const uint32_t off = pt * 8u;
#ifndef CM_LISTED_NT
#define CM_LISTED_NT 0              // experiment knob: non-temporal row accesses of the listed points
#endif
    auto row_load = [&](const double* base, int k) {
        cm_gcptr row = (cm_gcptr)(base + (int64_t)k * B);
        asm volatile("" : "+s"(row));
        if constexpr (CM_LISTED_NT != 0) return __builtin_nontemporal_load((const __attribute__((address_space(1))) double*)(row + off));
        else return *(const __attribute__((address_space(1))) double*)(row + off);
    };
    auto row_store = [&](double* base, int k, double v) {
        cm_gptr row = (cm_gptr)(base + (int64_t)k * B);
        asm volatile("" : "+s"(row));
        if constexpr (CM_LISTED_NT != 0) __builtin_nontemporal_store(v, (__attribute__((address_space(1))) double*)(row + off));
        else *(__attribute__((address_space(1))) double*)(row + off) = v;
    };
    double G[NU], xp[NX], x[NX], eg[6], z[6];
#pragma unroll
    for (int k = 0; k < NU; ++k) G[k] = row_load(gradu, k);
#pragma unroll
    for (int k = 0; k < NX; ++k) xp[k] = row_load(xi_prev, k);
    strain_from_gradu<CM_FULL_3D, ROT>(m, G, eg);
    strain_z<CM_FULL_3D, ROT>(m, z);
    __shared__ double ls_stage[LS ? kLsSlots * kBlock : 1];
    uint32_t st = newton_any<CM_FULL_3D, YK, LS, true, false>(m, eg, z, xp, x, valid, lane_stage(ls_stage, LS ? threadIdx.x : 0, kBlock));
    Eval<CM_FULL_3D> ev;
    strain_stress<CM_FULL_3D>(m, eg, z, x, ev);
    if (status) {
        double phi, gt[6], Ht[6][6];
        yield_eval<YK, false>(m, ev.s, phi, gt, Ht);
        const double f = (phi - (m.Y + hardening(m, x[6]).H)) * 0.5 / m.mu;
        if ((f > m.yield_tol) || (fabs(f) < m.yield_tol)) st |= CM_STATUS_PLASTIC;
    }
    double sg[6];
    to_global<ROT>(m, ev.s, sg);
    if (valid) {
#pragma unroll
        for (int k = 0; k < NX; ++k) row_store(xi, k, x[k]);
        if (sigma) {
#pragma unroll
            for (int k = 0; k < 6; ++k) row_store(sigma, k, sg[k]);
        }
        if (status) status[pt] = st;
    }
}

// ---- cm_update_rate: rate-form model (small_rate_elastic_plastic) ------------------------------------------
template <int DEF, int YK, bool ROT, bool LS, bool TANGENT = false>
__global__ __launch_bounds__(kBlock) void k_update_rate(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev, const double* __restrict__ xi_prev,
        double* __restrict__ xi, double* __restrict__ sigma, double* __restrict__ dsig, uint32_t* __restrict__ status) {
    constexpr int NX = nx_of<DEF, CM_SMALL_RATE_ELASTIC_PLASTIC>(), NU = Dims<DEF>::NU;
    constexpr bool RU = (DEF == CM_UNIAXIAL_STRESS);             // 12 local dofs, derivative blocks by forward-mode evaluation
    const int64_t blk0 = (int64_t)blockIdx.x * kBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);
    gradu += blk0; gradu_prev += blk0; xi_prev += blk0; xi += blk0;
    if (sigma) sigma += blk0;
    if (status) status += blk0;
    if (dsig) dsig += blk0;
    double G[NU], Gp[NU], xp[NX], x[NX], deg[6], z[Dims<DEF>::NZ];
    load_soa<NU>(gradu, B, b, G);
    load_soa<NU>(gradu_prev, B, b, Gp);
    load_soa<NX>(xi_prev, B, b, xp);
#pragma unroll
    for (int k = 0; k < NU; ++k) G[k] -= Gp[k];                  // eps - eps_prev is linear in grad u
    uint32_t st;
    // the line search of the structured solver parks its iterate and direction in the lane's LDS column
    constexpr bool STAGED = LS && !RU && has_structured<DEF, YK>();
    __shared__ double ls_stage[STAGED ? kLsSlots * kBlock : 1];
    if constexpr (RU) st = ru_newton<YK, LS>(m, G[0], xp, x, valid);
    else {
        strain_from_gradu<DEF, ROT>(m, G, deg);
        strain_z<DEF, ROT>(m, z);
        st = newton_rate_any<DEF, YK, LS>(m, deg, z, xp, x, valid, lane_stage(ls_stage, STAGED ? threadIdx.x : 0, kBlock));
    }
    double sg[6];
    to_global<ROT>(m, x, sg);                                    // small_rate_elastic_plastic.py:351-359
    if (status) {
        double phi, gt[6], Ht[6][6];
        yield_eval<YK, false>(m, x, phi, gt, Ht);
        const double f = (phi - (m.Y + hardening(m, x[6]).H)) * 0.5 / m.mu;
        if ((f > m.yield_tol) || (fabs(f) < m.yield_tol)) st |= CM_STATUS_PLASTIC;
    }
    if (valid) {
        store_soa<NX>(xi, B, b, x);
        if (sigma) store_soa<6>(sigma, B, b, sg);
        if (status) status[b] = st;
    }
    if constexpr (TANGENT) {
        // d sig_g / d G_c = Rg T Rm dE/dG_c (and minus that w.r.t. grad u_prev): the same chain as k_update
        if constexpr (RU) {
            double ds[6];
            const bool okr = ru_tangent<YK>(m, G[0], x, xp, ds);
            if (!okr && valid && status) status[b] = st | CM_STATUS_SINGULAR;
            if (valid) {
#pragma unroll
                for (int r = 0; r < 6; ++r) (dsig + (int64_t)r * B)[b] = ds[r];
            }
            return;
        }
        double T[6][6];
        const bool ok = tangent_rate_any<RU ? CM_FULL_3D : DEF, YK>(m, deg, z, x, xp, T);
        if (!ok && valid && status) status[b] = st | CM_STATUS_SINGULAR;
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            double Gd[NU], dm[6], t[6], tg[6];
#pragma unroll
            for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
            strain_from_gradu<DEF, ROT>(m, Gd, dm);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < 6; ++l) s += T[r][l] * dm[l];
                t[r] = s;
            }
            to_global<ROT>(m, t, tg);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 6; ++r) (dsig + (int64_t)(r * NU + c) * B)[b] = tg[r];
            }
        }
    }
}

// ---- block reduction of NV doubles per lane into partials[blockIdx][NV] ----------------------------
// Through LDS instead of 6 x NV cross-lane shuffles: every lane parks its NV values (row k at sh[k * kRedStride],
// conflict-free), then thread (k, j) = (t / 16, t % 16) adds the 16 entries i * 16 + j of row k and the 16 partial
// sums of a row are combined with four 16-lane shuffles.  ~60 instead of ~230 instructions per lane; fixed
// summation order, so the result is reproducible bit for bit.  `sh` may alias storage the Newton loop used
// (hence the leading barrier); it needs NV * kRedStride doubles.
constexpr int kRedStride = kBlock + 16;      // rows start 32 banks apart: the four 16-lane groups of a wave do not collide
// NACT: only the first NACT values can be non-zero (J2: objective + six parameters) -- the others are written as zeros
// without passing through LDS, and with NACT <= 64 / kGroup the block's other wavefronts skip the summation altogether.
template <int NV, int NACT = NV>
__device__ __forceinline__ void block_reduce_store(const double* v, double* __restrict__ partials, double* sh) {
    // kGroup lanes share a row: each adds kBlock / kGroup = 16 entries, then log2(kGroup) shuffle steps (256 lanes: 16 x 16
    // and four steps; 64 lanes: 4 x 16 and two)
    constexpr int kGroup = kBlock / 16;
    static_assert((kBlock == 256 || kBlock == 128 || kBlock == 64) && NV <= 16 && NACT <= NV, "thread (k, j) layout: 16 rows of kBlock / 16 lanes");
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NACT; ++k) sh[k * kRedStride + threadIdx.x] = v[k];
    __syncthreads();
    const int k = threadIdx.x / kGroup, j = threadIdx.x % kGroup;
    if (NACT * kGroup <= 64 && threadIdx.x >= 64) {                 // wave-uniform: this wavefront's rows are all zero
        if (k < NV && j == 0) partials[(int64_t)blockIdx.x * NV + k] = 0.0;
        return;
    }
    double a = 0.0;
    if (k < NACT) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a += sh[k * kRedStride + i * kGroup + j];
    }
#pragma unroll
    for (int off = kGroup / 2; off > 0; off >>= 1) a += __shfl_xor(a, off, kGroup);
    if (k < NV && j == 0) partials[(int64_t)blockIdx.x * NV + k] = a;
}

// deterministic two-stage reduction of the block partials (fixed order, no atomics):
// stage 1: kRedBlocks blocks each sum a contiguous slice of rows -> stage[kRedBlocks][NV]
// stage 2: one block sums the kRedBlocks rows -> out[k] (+)= ...
constexpr int kRedBlocks = 128;

#if CM_HAS_PART(1) && !CM_HNN_VARIANT      // one copy in the library: every part launches them through cm_detail_reduce()
template <int NV>
__device__ __forceinline__ void reduce_rows(const double* __restrict__ rows, int64_t begin, int64_t end, double* res) {
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    for (int64_t i = begin + threadIdx.x; i < end; i += kRBlock) {
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] += rows[i * NV + k];
    }
    __shared__ double sh[kRBlock / 64][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double a = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        acc[k] = a;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) sh[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kRBlock / 64; ++w) s += sh[w][threadIdx.x];
        res[threadIdx.x] = s;
    }
}

template <int NV>
__global__ __launch_bounds__(kRBlock) void k_reduce_stage1(const double* __restrict__ partials, int64_t nrows,
                                                          double* __restrict__ stage) {
    const int64_t per = (nrows + kRedBlocks - 1) / kRedBlocks;
    const int64_t begin = (int64_t)blockIdx.x * per;
    const int64_t end = begin + per < nrows ? begin + per : nrows;
    __shared__ double res[NV];
    reduce_rows<NV>(partials, begin < nrows ? begin : nrows, end, res);
    __syncthreads();
    if (threadIdx.x < NV) stage[(int64_t)blockIdx.x * NV + threadIdx.x] = res[threadIdx.x];
}

// out[k - first] (+)= total[k] for k >= first (first = 1 drops the objective slot: the vjp entry points
// return only the 12 gradient entries, written straight to the caller's array -- no device-to-device copy)
template <int NV>
__global__ __launch_bounds__(kRBlock) void k_reduce_stage2(const double* __restrict__ stage, double* __restrict__ out,
                                                          int first, int accumulate) {
    __shared__ double res[NV];
    reduce_rows<NV>(stage, 0, kRedBlocks, res);
    __syncthreads();
    if ((int)threadIdx.x < NV && (int)threadIdx.x >= first) {
        const int o = (int)threadIdx.x - first;
        out[o] = accumulate ? out[o] + res[threadIdx.x] : res[threadIdx.x];
    }
}

#endif
// ---- cm_update_vjp / cm_objective_grad / cm_adjoint_step ----------------------------------------------
// MODE 0: vjp for a given sigma_bar (xi given, converged)
// MODE 1: fused update + calibration QoI + gradient (xi computed here)
// MODE 2: adjoint step (xi given; QoI cotangent + incoming history)
// MODE 3: fused update + vjp for a given sigma_bar (xi and sigma computed and stored here)
struct Wsq { double w[6]; };

// minimum waves per SIMD requested from the register allocator: the J2 FULL_3D plain-Newton variants sit at
// ~130 VGPRs, two above the 4-wave limit (128); every other variant is left unconstrained.
template <int DEF, int YK, bool LS, int MODE, bool RL = false>
constexpr int min_waves() {
    if (DEF == CM_PLANE_STRESS && YK == CM_YIELD_J2 && RL) return LS ? CM_OCC_PS_J2_PLANE_LS : CM_OCC_PS_J2_PLANE;
    if (DEF == CM_UNIAXIAL_STRESS && (YK == CM_YIELD_J2 || YK == CM_YIELD_HILL) && MODE == 1) return CM_OCC_UNIAXIAL;   // (MODE 3, update + vjp: 4 % slower at 2 waves)
    if (DEF == CM_FULL_3D && YK == CM_YIELD_J2 && LS) return CM_OCC_REV_J2_LS;
    if (DEF == CM_FULL_3D && YK == CM_YIELD_HILL && !LS) return CM_OCC_REV_HILL;
    return (DEF == CM_FULL_3D && YK == CM_YIELD_J2 && !LS && (MODE == 1 || MODE == 3)) ? CM_OCC_REV_J2 : 1;
}

template <int DEF, int YK, bool ROT, bool LS, int MODE, bool RL = false>
__global__ __launch_bounds__(kBlock, (min_waves<DEF, YK, LS, MODE, RL>())) void k_reverse(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ xi_prev, const double* __restrict__ xi_in,
        const double* __restrict__ sbar_or_data, Wsq wsq, const double* hist_in,
        double* __restrict__ xi_out, double* __restrict__ sigma_out, double* xpbar_out,
        double* __restrict__ gbar_out, double* __restrict__ partials) {
    // hist_in and xpbar_out carry no __restrict__: cm_adjoint_step documents that hist_out may alias hist_in (the history
    // vector is updated in place; each lane reads its own column before it writes it)
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    const int64_t blk0 = (int64_t)blockIdx.x * kBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);
    gradu += blk0; xi_prev += blk0; sbar_or_data += blk0;
    if (xi_in) xi_in += blk0;
    if (hist_in) hist_in += blk0;
    if (xi_out) xi_out += blk0;
    if (sigma_out) sigma_out += blk0;
    if (xpbar_out) xpbar_out += blk0;
    if (gbar_out) gbar_out += blk0;
    double G[NU], xp[NX], x[NX], eg[6], z[Dims<DEF>::NZ], sd[6];
    load_soa<NU>(gradu, B, b, G);
    load_soa<NX>(xi_prev, B, b, xp);
#ifndef CM_EARLY_SBAR
#define CM_EARLY_SBAR 1
#endif
    // the fused J2 update + vjp kernel has registers to spare since round 2 (108 of 128): all 22 row loads in flight before the
    // solve instead of 16 + 6 (-3 % and steadier, profiles/r02_early_sigma_bar_ab.txt); the objective kernel (125) would spill
    constexpr bool EARLY = (CM_EARLY_SBAR != 0) && RL && DEF == CM_FULL_3D && !LS && MODE == 3;
    if constexpr (MODE == 0 || MODE == 2 || EARLY) load_soa<6>(sbar_or_data, B, b, sd);
    strain_from_gradu<DEF, ROT>(m, G, eg);
    strain_z<DEF, ROT>(m, z);
    // fused modes on the structured path keep the converged-state evaluation for the reverse sweep
    constexpr bool SFAST = (has_structured<DEF, YK>() && (MODE == 1 || MODE == 3));
    EvalS<SFAST ? YK : CM_YIELD_J2> evs;
    // one LDS buffer: the line search's parked iterates during the Newton loop, the gradient reduction afterwards
    constexpr int kLdsDoubles = ((SFAST && LS) ? kLsSlots * kBlock : 0) > kRed * kRedStride ? kLsSlots * kBlock : kRed * kRedStride;
    __shared__ double lds_buf[kLdsDoubles];
    double* const ls_stage = lds_buf;
    uint32_t st = CM_STATUS_CONVERGED;
    if constexpr (MODE == 1 || MODE == 3) {
        if constexpr (SFAST) {
            if constexpr (RL) st = newton_fast<DEF, YK, LS>(m, eg, z, xp, x, valid, evs, lane_stage(ls_stage, LS ? threadIdx.x : 0, kBlock));
            else newton_s_warm<DEF, YK, LS>(m, eg, z, xp, x, valid, evs, lane_stage(ls_stage, LS ? threadIdx.x : 0, kBlock));
        }
        else newton_any<DEF, YK, LS, true, RL>(m, eg, z, xp, x, valid);
        if constexpr (!EARLY) load_soa<6>(sbar_or_data, B, b, sd);       // after the solve: 12 fewer live VGPRs inside the Newton loop
        if (xi_out && valid) store_soa<NX>(xi_out, B, b, x);
        if constexpr (MODE == 3) {
            if (sigma_out) {
                // from the stored state, as cm_update computes it: the two entry points return bit-identical stresses
                Eval<DEF> ev;
                strain_stress<DEF>(m, eg, z, x, ev);
                double sg[6];
                to_global<ROT>(m, ev.s, sg);
                if (valid) store_soa<6>(sigma_out, B, b, sg);
            }
        }
    } else {
        load_soa<NX>(xi_in, B, b, x);
    }
    double red[kRed];
    double sb[6];
    red[0] = 0.0;
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int k = 0; k < 6; ++k) sb[k] = sd[k];
    } else {
        // J = 1/2 sum_r wsq_r (sig_r - data_r)^2 ;  sbar_r = wsq_r (sig_r - data_r)   (qois/calibration.py:56-66)
        double sg[6];
        // plain Newton, Q = I: the solver's evaluation at the returned state (with the line search the six extra live registers
        // cost the kernel its third wavefront per SIMD, with a rotation they spill)
        if constexpr (SFAST && MODE == 1 && !LS && !ROT) to_global<ROT>(m, evs.s, sg);
        else {
            Eval<DEF> ev;
            strain_stress<DEF>(m, eg, z, x, ev);
            to_global<ROT>(m, ev.s, sg);
        }
        double J = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double mm = sg[k] - sd[k];
            sb[k] = wsq.w[k] * mm;
            J += 0.5 * sb[k] * mm;
        }
        red[0] = J;
    }
    double sbm[6], xin[NX], xpbar[NX], egbar[6];
    cotangent_to_material<ROT>(m, sb, sbm);
#pragma unroll
    for (int k = 0; k < NX; ++k) xin[k] = 0.0;
    if constexpr (MODE == 2) {
        if (hist_in) {
            load_soa<NX>(hist_in, B, b, xin);
#pragma unroll
            for (int k = 0; k < NX; ++k) xin[k] = -xin[k];       // history vector = -(cotangent of xi)
        }
    }
    // arrays are always passed (never a run-time null): a nullable local array would be forced into scratch
    // (MODE 1 / 3 have no per-point cotangent outputs at all: compile-time nulls let the compiler drop that work)
    constexpr bool BARS = (MODE == 0 || MODE == 2);
    if constexpr (SFAST && RL && YK == CM_YIELD_J2) {
        // converged J2 states: the parameter gradient is the derivative of the return map in its own coordinates (the radial
        // line, the plane: no 7 / 8-dof transposed solve); a wavefront holding an unconverged point (iteration cap)
        // differentiates through A(x) at the returned state as the reference does
        if (!__any(!(st & CM_STATUS_CONVERGED))) {
            if constexpr (DEF == CM_FULL_3D) reverse_j2_radial(m, eg, x, sbm, evs, &red[1]);
            else reverse_j2_plane(m, eg, z, x, sbm, evs, &red[1]);
        }
        else reverse_point_s<YK, true, DEF>(m, eg, x, xp, sbm, nullptr, &red[1], nullptr, nullptr, &evs, z);
    }
    else if constexpr (SFAST) reverse_point_s<YK, true, DEF>(m, eg, x, xp, sbm, nullptr, &red[1], nullptr, nullptr, &evs, z);
    else reverse_any<DEF, YK>(m, eg, z, x, xp, sbm, (MODE == 2) ? xin : nullptr, &red[1], BARS ? xpbar : nullptr,
                              BARS ? egbar : nullptr);
    if (BARS && xpbar_out && valid) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int k = 0; k < NX; ++k) xpbar[k] = -xpbar[k];   // back to history-vector sign
        }
        store_soa<NX>(xpbar_out, B, b, xpbar);
    }
    if (BARS && gbar_out) {
        // cotangent of grad u: Gbar_c = egbar . d eg / d G_c
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            double Gd[NU], dm[6];
#pragma unroll
            for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
            strain_from_gradu<DEF, ROT>(m, Gd, dm);
            if (valid) (gbar_out + (int64_t)c * B)[b] = dot<6>(egbar, dm);
        }
    }
    if (!valid) {
#pragma unroll
        for (int k = 0; k < kRed; ++k) red[k] = 0.0;
    }
    // J2: objective, lambda, mu, Y, S, D, K -- the yield-coefficient slots are identically zero
    block_reduce_store<kRed, (YK == CM_YIELD_J2) ? 1 + CM_P_YC0 : kRed>(red, partials, lds_buf);
}

// ---- reverse-mode kernels of the rate-form model (same MODEs as k_reverse; structured solver via reverse_rate_any) -----
// grad u enters through deg = strain(grad u - grad u_prev): the cotangent written to gbar_out is the one of grad u,
// the one of grad u_prev is its negative.
template <int DEF, int YK, bool ROT, bool LS, int MODE>
__global__ __launch_bounds__(kBlock) void k_reverse_rate(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev, const double* __restrict__ xi_prev,
        const double* __restrict__ xi_in, const double* __restrict__ sbar_or_data, Wsq wsq,
        const double* hist_in /* may alias xpbar_out */, double* __restrict__ xi_out, double* __restrict__ sigma_out,
        double* xpbar_out, double* __restrict__ gbar_out, double* __restrict__ partials) {
    constexpr int NX = nx_of<DEF, CM_SMALL_RATE_ELASTIC_PLASTIC>(), NU = Dims<DEF>::NU;
    constexpr bool RU = (DEF == CM_UNIAXIAL_STRESS);
    const int64_t blk0 = (int64_t)blockIdx.x * kBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);
    gradu += blk0; gradu_prev += blk0; xi_prev += blk0; sbar_or_data += blk0;
    if (xi_in) xi_in += blk0;
    if (hist_in) hist_in += blk0;
    if (xi_out) xi_out += blk0;
    if (sigma_out) sigma_out += blk0;
    if (xpbar_out) xpbar_out += blk0;
    if (gbar_out) gbar_out += blk0;
    // one LDS buffer: the line search's parked iterates during the Newton loop, the gradient reduction afterwards
    constexpr bool STAGED = LS && !RU && (MODE == 1 || MODE == 3) && has_structured<DEF, YK>();
    constexpr int kLdsDoubles = (STAGED ? kLsSlots * kBlock : 0) > kRed * kRedStride ? kLsSlots * kBlock : kRed * kRedStride;
    __shared__ double lds_buf[kLdsDoubles];
    double G[NU], Gp[NU], xp[NX], x[NX], deg[6], z[Dims<DEF>::NZ], sd[6];
    load_soa<NU>(gradu, B, b, G);
    load_soa<NU>(gradu_prev, B, b, Gp);
    load_soa<NX>(xi_prev, B, b, xp);
#pragma unroll
    for (int k = 0; k < NU; ++k) G[k] -= Gp[k];
    if constexpr (!RU) {
        strain_from_gradu<DEF, ROT>(m, G, deg);
        strain_z<DEF, ROT>(m, z);
    }
    if constexpr (MODE == 1 || MODE == 3) {
        if constexpr (RU) ru_newton<YK, LS>(m, G[0], xp, x, valid);
        else newton_rate_any<DEF, YK, LS>(m, deg, z, xp, x, valid, lane_stage(lds_buf, STAGED ? threadIdx.x : 0, kBlock));
        if (xi_out && valid) store_soa<NX>(xi_out, B, b, x);
    } else {
        load_soa<NX>(xi_in, B, b, x);
    }
    load_soa<6>(sbar_or_data, B, b, sd);
    double sg[6];
    to_global<ROT>(m, x, sg);                                    // sigma_global = Q x[0:6] Q^T
    if constexpr (MODE == 3) {
        if (sigma_out && valid) store_soa<6>(sigma_out, B, b, sg);
    }
    double red[kRed], sb[6];
    red[0] = 0.0;
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int k = 0; k < 6; ++k) sb[k] = sd[k];
    } else {
        double J = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double mm = sg[k] - sd[k];
            sb[k] = wsq.w[k] * mm;
            J += 0.5 * sb[k] * mm;
        }
        red[0] = J;
    }
    double sbm[6], xin[NX], xpbar[NX], degbar[6];
    cotangent_to_material<ROT>(m, sb, sbm);
#pragma unroll
    for (int k = 0; k < NX; ++k) xin[k] = 0.0;
    if constexpr (MODE == 2) {
        if (hist_in) {
            load_soa<NX>(hist_in, B, b, xin);
#pragma unroll
            for (int k = 0; k < NX; ++k) xin[k] = -xin[k];       // history vector = -(cotangent of xi)
        }
    }
    constexpr bool BARS = (MODE == 0 || MODE == 2);
    if constexpr (RU) {
        double ubar = 0.0;
        ru_reverse<YK>(m, G[0], x, xp, sb, (MODE == 2) ? xin : nullptr, &red[1], BARS ? xpbar : nullptr, BARS ? &ubar : nullptr);
        degbar[0] = ubar;                                        // the one grad-u entry: its cotangent, stored below
    } else {
        reverse_rate_any<RU ? CM_FULL_3D : DEF, YK>(m, deg, z, x, xp, sbm, (MODE == 2) ? xin : nullptr, &red[1], BARS ? xpbar : nullptr,
                                                     BARS ? degbar : nullptr);
    }
    if (BARS && xpbar_out && valid) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int k = 0; k < NX; ++k) xpbar[k] = -xpbar[k];
        }
        store_soa<NX>(xpbar_out, B, b, xpbar);
    }
    if (BARS && gbar_out) {
        if constexpr (RU) {
            if (valid) gbar_out[b] = degbar[0];
        } else {
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                double Gd[NU], dm[6];
#pragma unroll
                for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
                strain_from_gradu<DEF, ROT>(m, Gd, dm);
                if (valid) (gbar_out + (int64_t)c * B)[b] = dot<6>(degbar, dm);
            }
        }
    }
    if (!valid) {
#pragma unroll
        for (int k = 0; k < kRed; ++k) red[k] = 0.0;
    }
    block_reduce_store<kRed>(red, partials, lds_buf);
}

// ---- cm_objective_grad_history: the whole K-step history of every point in one launch ------------------------------
// (forward updates with the states stored, then the adjoint recursion; cm::history_point)
struct SoaRowsIO {
    int64_t B; unsigned b;
    template <int N> __device__ __forceinline__ void load(const double* base, int64_t row0, double* out) const { load_soa<N, false>(base + row0 * B, B, b, out); }
    template <int N> __device__ __forceinline__ void store(double* base, int64_t row0, const double* v) const { store_soa<N, false>(base + row0 * B, B, b, v); }
    __device__ __forceinline__ void phase_barrier() const { __syncthreads(); }
    __device__ __forceinline__ void store_status(uint32_t* base, int64_t row, uint32_t v) const { (base + row * B)[b] = v; }
};

// plain per-point row access for the one-thread-per-point kernels (point b of row r at base[r * B + b])
struct PointRowsIO {
    int64_t B, b;
    __device__ __forceinline__ double get(const double* base, int64_t row) const { return base[row * B + b]; }
    __device__ __forceinline__ void put(double* base, int64_t row, double v) const { base[row * B + b] = v; }
};

#ifndef CM_OCC_HIST_J2
#define CM_OCC_HIST_J2 1           // whole-history objective kernel, J2 / FULL_3D on the radial line
#endif
template <int DEF, int YK, bool LS, int MK, bool RL>
constexpr int min_waves_history() {
    return (DEF == CM_FULL_3D && YK == CM_YIELD_J2 && !LS && MK == CM_SMALL_ELASTIC_PLASTIC && RL) ? CM_OCC_HIST_J2 : 1;
}
template <int DEF, int YK, bool ROT, bool LS, int MK, bool RL = false>
__global__ __launch_bounds__(kBlock, (min_waves_history<DEF, YK, LS, MK, RL>())) void k_history(cm_model_desc m, int64_t B, int K,
        const double* __restrict__ gradu_hist, const double* __restrict__ data_hist, Wsq wsq,
        const double* __restrict__ xi0, double* xi_hist, double* __restrict__ partials, HistoryCotangents hc) {
    const int64_t blk0 = (int64_t)blockIdx.x * kBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);
    constexpr bool STAGED = LS && has_structured<DEF, YK>();       // both model kinds: the rate form rides on the same solver
    constexpr int kLdsDoubles = (STAGED ? kLsSlots * kBlock : 0) > kRed * kRedStride ? kLsSlots * kBlock : kRed * kRedStride;
    __shared__ double lds_buf[kLdsDoubles];
    double red[kRed];
#pragma unroll
    for (int k = 0; k < kRed; ++k) red[k] = 0.0;
    if (hc.sbar_hist) hc.sbar_hist += blk0;
    if (hc.xibar_hist) hc.xibar_hist += blk0;
    if (hc.lam_hist) hc.lam_hist += blk0;
    history_point<DEF, YK, ROT, LS, MK, RL>(m, K, gradu_hist + blk0, data_hist ? data_hist + blk0 : nullptr, wsq.w, xi0 + blk0,
                                        xi_hist + blk0, valid, lane_stage(lds_buf, STAGED ? threadIdx.x : 0, kBlock),
                                        SoaRowsIO{B, b}, red, hc);
    if (!valid) {
#pragma unroll
        for (int k = 0; k < kRed; ++k) red[k] = 0.0;
    }
    __syncthreads();             // the line search's LDS columns are done before the reduction reuses the buffer
    block_reduce_store<kRed>(red, partials, lds_buf);
}

// ---- cm_update_history: K updates per point in one launch, states / stresses / statuses stored per step ------------
template <int DEF, int YK, bool ROT, bool LS, int MK, bool RL = false>
__global__ __launch_bounds__(kBlock) void k_primal_history(cm_model_desc m, int64_t B, int K,
        const double* __restrict__ gradu_hist, const double* __restrict__ xi0, double* __restrict__ xi_hist,
        double* __restrict__ sigma_hist, uint32_t* __restrict__ status_hist) {
    const int64_t blk0 = (int64_t)blockIdx.x * kBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);
    constexpr bool STAGED = LS && has_structured<DEF, YK>();
    __shared__ double ls_stage[STAGED ? kLsSlots * kBlock : 1];
    primal_history_point<DEF, YK, ROT, LS, MK, RL>(m, K, gradu_hist + blk0, xi0 + blk0, xi_hist ? xi_hist + blk0 : nullptr,
                                               sigma_hist ? sigma_hist + blk0 : nullptr, status_hist ? status_hist + blk0 : nullptr,
                                               valid, lane_stage(ls_stage, STAGED ? threadIdx.x : 0, kBlock), SoaRowsIO{B, b});
}

// ---- cm_evaluate: residual / Jacobian block / stress / stress-derivative block at given states ----------
template <int DEF, int YK, bool ROT>
__global__ __launch_bounds__(64) void k_evaluate(cm_model_desc m, int64_t B, int which,
        const double* __restrict__ gradu, const double* __restrict__ xi_prev, const double* __restrict__ xi,
        double* __restrict__ C_out, double* __restrict__ J_out, double* __restrict__ s_out, double* __restrict__ S_out) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    constexpr int MAXC = CM_NUM_PARAMS;
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double G[NU], xp[NX], x[NX], C[NX], sg[6], J[NX * MAXC], S[6 * MAXC];
    load_soa<NU>(gradu, B, b, G);
    load_soa<NX>(xi_prev, B, b, xp);
    load_soa<NX>(xi, B, b, x);
    evaluate_blocks<DEF, YK, ROT>(m, G, x, xp, which, C, J_out ? J : nullptr, sg, S_out ? S : nullptr);
    const int ncols = (which == CM_W_XI || which == CM_W_XI_PREV) ? NX : (which == CM_W_PARAMS ? CM_NUM_PARAMS : NU);
    if (C_out) store_soa<NX>(C_out, B, b, C);
    if (s_out) store_soa<6>(s_out, B, b, sg);
    if (which != CM_W_NONE) {
        if (J_out) for (int i = 0; i < NX * ncols; ++i) J_out[(int64_t)i * B + b] = J[i];
        if (S_out) for (int i = 0; i < 6 * ncols; ++i) S_out[(int64_t)i * B + b] = S[i];
    }
}

template <int DEF, int YK, bool ROT>
__global__ __launch_bounds__(64) void k_evaluate_rate(cm_model_desc m, int64_t B, int which,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev, const double* __restrict__ xi_prev,
        const double* __restrict__ xi, double* __restrict__ C_out, double* __restrict__ J_out,
        double* __restrict__ s_out, double* __restrict__ S_out) {
    constexpr int NX = nx_of<DEF, CM_SMALL_RATE_ELASTIC_PLASTIC>(), NU = Dims<DEF>::NU;
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double G[NU], Gp[NU], xp[NX], x[NX], C[NX], sg[6], J[NX * CM_NUM_PARAMS], S[6 * CM_NUM_PARAMS];
    for (int k = 0; k < NU; ++k) { G[k] = gradu[(int64_t)k * B + b]; Gp[k] = gradu_prev[(int64_t)k * B + b]; }
    for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[(int64_t)k * B + b]; x[k] = xi[(int64_t)k * B + b]; }
    if constexpr (DEF == CM_UNIAXIAL_STRESS) {
        ru_eval<YK>(m, G[0] - Gp[0], x, xp, C, sg);
        if (which != CM_W_NONE) ru_block<YK>(m, G[0] - Gp[0], x, xp, which, J_out ? J : nullptr, S_out ? S : nullptr);
    } else
    evaluate_blocks_rate<DEF, YK, ROT>(m, G, Gp, x, xp, which, C, J_out ? J : nullptr, sg, S_out ? S : nullptr);
    const int ncols = (which == CM_W_XI || which == CM_W_XI_PREV) ? NX : (which == CM_W_PARAMS ? CM_NUM_PARAMS : NU);
    if (C_out) for (int k = 0; k < NX; ++k) C_out[(int64_t)k * B + b] = C[k];
    if (s_out) for (int k = 0; k < 6; ++k) s_out[(int64_t)k * B + b] = sg[k];
    if (which != CM_W_NONE) {
        if (J_out) for (int i = 0; i < NX * ncols; ++i) J_out[(int64_t)i * B + b] = J[i];
        if (S_out) for (int i = 0; i < 6 * ncols; ++i) S_out[(int64_t)i * B + b] = S[i];
    }
}

// ---- cm_direct_step: forward parameter sensitivities of one converged step (cm::direct_point) ------------------------
template <int DEF, int YK, bool ROT, int MK>
__global__ __launch_bounds__(64) void k_direct_step(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev, const double* __restrict__ xi_prev,
        const double* __restrict__ xi, const double* __restrict__ dxp_dp, double* __restrict__ dx_dp, double* __restrict__ ds_dp) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU, NP_ = CM_NUM_PARAMS;
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double G[NU], Gp[NU], xp[NX], x[NX], din[NX * NP_], dout[NX * NP_], dsig[6 * NP_];
    for (int k = 0; k < NU; ++k) { G[k] = gradu[(int64_t)k * B + b]; Gp[k] = gradu_prev ? gradu_prev[(int64_t)k * B + b] : 0.0; }
    for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[(int64_t)k * B + b]; x[k] = xi[(int64_t)k * B + b]; }
    if (dxp_dp) for (int i = 0; i < NX * NP_; ++i) din[i] = dxp_dp[(int64_t)i * B + b];
    if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && DEF == CM_UNIAXIAL_STRESS)
        ru_direct<YK>(m, G[0] - Gp[0], x, xp, dxp_dp ? din : nullptr, dout, ds_dp ? dsig : nullptr);
    else
    direct_point<MK, DEF, YK, ROT>(m, G, Gp, x, xp, dxp_dp ? din : nullptr, dout, ds_dp ? dsig : nullptr);
    for (int i = 0; i < NX * NP_; ++i) dx_dp[(int64_t)i * B + b] = dout[i];
    if (ds_dp) for (int i = 0; i < 6 * NP_; ++i) ds_dp[(int64_t)i * B + b] = dsig[i];
}

// ---- cm_direct_step / cm_direct_history by columns: one thread per (point, native parameter) ---------------------------
// (cm::direct_column).  Points are the fast index: the lanes of a wavefront share the parameter for B >= 64; for the
// material-point objectives (B = 1) the twelve columns of the one point advance side by side.  The 12-dof rate form under
// UNIAXIAL_STRESS keeps the one-thread-per-point kernels above (its blocks come from forward-mode evaluation).
template <int DEF, int YK, bool ROT, int MK>
__global__ __launch_bounds__(64) void k_direct_step_cols(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev, const double* __restrict__ xi_prev,
        const double* __restrict__ xi, const double* __restrict__ dxp_dp, double* __restrict__ dx_dp, double* __restrict__ ds_dp) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU, NP_ = CM_NUM_PARAMS;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * NP_) return;
    const int64_t b = tid % B;
    const int j = (int)(tid / B);
    double G[NU], Gp[NU], xp[NX], x[NX], dprev[NX], d[NX], dsc[6];
#pragma unroll
    for (int k = 0; k < NU; ++k) { G[k] = gradu[(int64_t)k * B + b]; Gp[k] = gradu_prev ? gradu_prev[(int64_t)k * B + b] : 0.0; }
#pragma unroll
    for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[(int64_t)k * B + b]; x[k] = xi[(int64_t)k * B + b]; }
    if (dxp_dp) {
#pragma unroll
        for (int k = 0; k < NX; ++k) dprev[k] = dxp_dp[(int64_t)(k * NP_ + j) * B + b];
    }
    direct_column<MK, DEF, YK, ROT>(m, G, Gp, x, xp, j, dxp_dp ? dprev : nullptr, d, ds_dp ? dsc : nullptr);
#pragma unroll
    for (int k = 0; k < NX; ++k) dx_dp[(int64_t)(k * NP_ + j) * B + b] = d[k];
    if (ds_dp) {
#pragma unroll
        for (int r = 0; r < 6; ++r) ds_dp[(int64_t)(r * NP_ + j) * B + b] = dsc[r];
    }
}

template <int DEF, int YK, bool ROT, int MK>
__global__ __launch_bounds__(64) void k_direct_history_cols(cm_model_desc m, int64_t B, int K,
        const double* __restrict__ gradu_hist, const double* __restrict__ xi_hist,
        const double* __restrict__ sbar_hist, const double* __restrict__ xibar_hist,
        double* __restrict__ dx_dp_hist, double* __restrict__ ds_dp_hist, double* __restrict__ rows) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU, NP_ = CM_NUM_PARAMS;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * NP_) return;
    const int64_t b = tid % B;
    const int j = (int)(tid / B);
    double d[NX], dn[NX], dsc[6], g = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) d[k] = 0.0;
    if (dx_dp_hist) {                                            // slot 0: no sensitivity before the first step
#pragma unroll
        for (int k = 0; k < NX; ++k) dx_dp_hist[(int64_t)(k * NP_ + j) * B + b] = 0.0;
    }
    if (ds_dp_hist) {
#pragma unroll
        for (int r = 0; r < 6; ++r) ds_dp_hist[(int64_t)(r * NP_ + j) * B + b] = 0.0;
    }
    for (int step = 1; step <= K; ++step) {
        double G[NU], Gp[NU], xp[NX], x[NX];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            G[k] = gradu_hist[((int64_t)step * NU + k) * B + b];
            Gp[k] = gradu_hist[((int64_t)(step - 1) * NU + k) * B + b];
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            x[k] = xi_hist[((int64_t)step * NX + k) * B + b];
            xp[k] = xi_hist[((int64_t)(step - 1) * NX + k) * B + b];
        }
        direct_column<MK, DEF, YK, ROT>(m, G, Gp, x, xp, j, step > 1 ? d : nullptr, dn, dsc);
        if (dx_dp_hist) {
#pragma unroll
            for (int k = 0; k < NX; ++k) dx_dp_hist[((int64_t)step * NX * NP_ + k * NP_ + j) * B + b] = dn[k];
        }
        if (ds_dp_hist) {
#pragma unroll
            for (int r = 0; r < 6; ++r) ds_dp_hist[((int64_t)step * 6 * NP_ + r * NP_ + j) * B + b] = dsc[r];
        }
        if (sbar_hist) {
#pragma unroll
            for (int r = 0; r < 6; ++r) g += sbar_hist[((int64_t)step * 6 + r) * B + b] * dsc[r];
        }
        if (xibar_hist) {
#pragma unroll
            for (int k = 0; k < NX; ++k) g += xibar_hist[((int64_t)step * NX + k) * B + b] * dn[k];
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) d[k] = dn[k];
    }
    if (rows) {
        if (j == 0) rows[b * kRed] = 0.0;
        rows[b * kRed + 1 + j] = g;
    }
}

// ---- cm_hessians: one thread per (point, pair of differentiation variables) -----------------------------------
template <int DEF, int YK, bool ROT, int MK = CM_SMALL_ELASTIC_PLASTIC>
__global__ __launch_bounds__(64) void k_hessians(cm_model_desc m, int64_t B,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev,
        const double* __restrict__ xi_prev, const double* __restrict__ xi,
        double* __restrict__ d2C, double* __restrict__ d2S, double* __restrict__ dC, double* __restrict__ dS,
        double* __restrict__ C0, double* __restrict__ S0) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU, NQ = 2 * NX + CM_NUM_PARAMS, NPAIR = NQ * (NQ + 1) / 2;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * NPAIR) return;
    const int64_t pt = tid / NPAIR;
    int rem = (int)(tid % NPAIR), a = 0;
    while (rem >= NQ - a) { rem -= NQ - a; ++a; }         // pairs (a, b >= a) in row-major order
    const int b = a + rem;
    double G[NU], xp[NX], x[NX], oC[NX], oS[6], oCa[NX], oSa[6], oC0[NX], oS0[6];
    for (int k = 0; k < NU; ++k) {
        G[k] = gradu[(int64_t)k * B + pt];
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) G[k] -= gradu_prev[(int64_t)k * B + pt];
    }
    for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[(int64_t)k * B + pt]; x[k] = xi[(int64_t)k * B + pt]; }
    const bool first = (a == 0 && b == 0);
    hessian_pair<DEF, YK, ROT, MK>(m, G, x, xp, a, b, oC, oS, oCa, oSa, first ? oC0 : nullptr, first ? oS0 : nullptr);
    if (d2C) for (int k = 0; k < NX; ++k) {
        d2C[((pt * NX + k) * NQ + a) * NQ + b] = oC[k];
        d2C[((pt * NX + k) * NQ + b) * NQ + a] = oC[k];
    }
    if (d2S) for (int k = 0; k < 6; ++k) {
        d2S[((pt * 6 + k) * NQ + a) * NQ + b] = oS[k];
        d2S[((pt * 6 + k) * NQ + b) * NQ + a] = oS[k];
    }
    if (a == b) {
        if (dC) for (int k = 0; k < NX; ++k) dC[(pt * NX + k) * NQ + a] = oCa[k];
        if (dS) for (int k = 0; k < 6; ++k) dS[(pt * 6 + k) * NQ + a] = oSa[k];
    }
    if (first) {                                          // values: residual and global stress at the given state
        if (C0) for (int k = 0; k < NX; ++k) C0[pt * NX + k] = oC0[k];
        if (S0) for (int k = 0; k < 6; ++k) S0[pt * 6 + k] = oS0[k];
    }
}

// ---- cm_direct_history: the forward-sensitivity recursion of a whole load history in one launch ------------------------
// (cmad/objectives/mp_objective.py:158-215, MPDirectObjective: dxi_k/dp = -A_k^-1 (dC_k/dp + dC_k/dxi_prev dxi_{k-1}/dp),
//  dsigma_k/dp = dsigma/dp|_xi + dsigma/dxi dxi_k/dp; one thread per point, the 7..9 x 12 sensitivity block carried from
//  step to step).  Optional per-step outputs dxi_dp_hist[(K+1)][NX*12][B], dsigma_dp_hist[(K+1)][6*12][B]; with the QoI
//  cotangents sbar_hist[(K+1)][6][B] (and xibar_hist[(K+1)][NX][B]) the gradient contraction
//  g_j = sum_k sbar_k . dsigma_k/dp_j + xibar_k . dxi_k/dp_j happens here and rows[b][1 + j] receives point b's share.
template <int DEF, int YK, bool ROT, int MK>
__global__ __launch_bounds__(64) void k_direct_history(cm_model_desc m, int64_t B, int K,
        const double* __restrict__ gradu_hist, const double* __restrict__ xi_hist,
        const double* __restrict__ sbar_hist, const double* __restrict__ xibar_hist,
        double* __restrict__ dx_dp_hist, double* __restrict__ ds_dp_hist, double* __restrict__ rows) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double g[CM_NUM_PARAMS];
    direct_history_point<DEF, YK, ROT, MK>(m, K, gradu_hist, xi_hist, sbar_hist, xibar_hist, dx_dp_hist, ds_dp_hist,
                                           PointRowsIO{B, b}, g);
    if (rows) {
        rows[b * kRed] = 0.0;
        for (int j = 0; j < CM_NUM_PARAMS; ++j) rows[b * kRed + 1 + j] = g[j];
    }
}

// ---- cm_hessian_history: second-order (direct-adjoint) contraction of a stored history ---------------------------------
// cmad/objectives/mp_objective.py:218-345 (MPDirectAdjointObjective).  With q = [xi_k, xi_{k-1}, p] (NQ entries), the
// Lagrangian of step k is  J_k(sigma(xi_k, p)) - lam_k . C_k(q)  (lam = -phi, cm_adjoint_history) and the total
// derivative of q w.r.t. the parameters is  D_k = [dxi_k/dp ; dxi_{k-1}/dp ; I]  (cm_direct_history), so that
//     d2J/dp2 = sum_k D_k^T W_k D_k ,   W_k[a][b] = sbar . d2sigma/dq_a dq_b + sum_r hss_r dsigma_r/dq_a dsigma_r/dq_b
//                                                  - lam . d2C/dq_a dq_b
// which is the reference's 13-term sum written as one quadratic form.  Stage 1: one thread per (point, step, pair a <= b)
// evaluates the residual in hyper-dual arithmetic (cm::hessian_pair) and writes W; stage 2: one block per (point, step)
// forms the 12 x 12 quadratic form; stage 3: a fixed-order sum over (point, step).
template <int DEF, int YK, bool ROT, int MK>
__global__ __launch_bounds__(64) void k_hessian_weights(cm_model_desc m, int64_t B, int K,
        const double* __restrict__ gradu_hist, const double* __restrict__ xi_hist, const double* __restrict__ lam_hist,
        const double* __restrict__ sbar_hist, Wsq hss, const double* __restrict__ hss_hist, const double* __restrict__ hxx_hist,
        double* __restrict__ W) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU, NQ = 2 * NX + CM_NUM_PARAMS, NPAIR = NQ * (NQ + 1) / 2;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * K * NPAIR) return;
    const int64_t ps = tid / NPAIR;                       // (step - 1) * B + point
    const int64_t pt = ps % B;
    const int step = (int)(ps / B) + 1;
    int rem = (int)(tid % NPAIR), a = 0;
    while (rem >= NQ - a) { rem -= NQ - a; ++a; }
    const int b = a + rem;
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
    // QoI curvature: diagonal in the six stored stress entries (per step when hss_hist is given: UniaxialCalibration's weights
    // change from step to step) and, for QoIs with an explicit dJ/dxi, diagonal in the state entries of the current step
    double hs[6];
    for (int r = 0; r < 6; ++r) hs[r] = hss_hist ? hss_hist[step * 6 + r] : hss.w[r];
    double w = hessian_weight<DEF, YK, ROT, MK>(m, G, x, xp, lam, sbar, hs, a, b);
    if (hxx_hist && a == b && a < NX) w += hxx_hist[step * NX + a];
    W[(ps * NQ + a) * NQ + b] = w;
    W[(ps * NQ + b) * NQ + a] = w;
}

// ---- second-order pass including EXTENDED parameters (cm_direct_history_ep, cm_hessian_history_ep) -------------------------
// q = [xi_k, xi_{k-1}, p (12 native), pe (n_ep extended)], NQ = 2 NX + 12 + n_ep (run time), D_k = dq/d[p, pe].
template <int DEF, int YK, bool ROT, int MK>
__global__ __launch_bounds__(64) void k_direct_history_ep(cm_model_desc m, int64_t B, int K, int n_ep, const int32_t* __restrict__ ep_index,
        const double* __restrict__ gradu_hist, const double* __restrict__ xi_hist, double* __restrict__ dxe_hist) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * n_ep) return;
    const int64_t b = tid % B;
    const int j = (int)(tid / B), e = ep_index[j];
    double d[NX], dn[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) { d[k] = 0.0; dxe_hist[(int64_t)(k * n_ep + j) * B + b] = 0.0; }        // slot 0
    for (int step = 1; step <= K; ++step) {
        double G[NU], Gp[NU], xp[NX], x[NX];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            G[k] = gradu_hist[((int64_t)step * NU + k) * B + b];
            Gp[k] = gradu_hist[((int64_t)(step - 1) * NU + k) * B + b];
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            x[k] = xi_hist[((int64_t)step * NX + k) * B + b];
            xp[k] = xi_hist[((int64_t)(step - 1) * NX + k) * B + b];
        }
        direct_column_ep<MK, DEF, YK, ROT>(m, G, Gp, x, xp, e, step > 1 ? d : nullptr, dn);
#pragma unroll
        for (int k = 0; k < NX; ++k) { dxe_hist[((int64_t)step * NX * n_ep + k * n_ep + j) * B + b] = dn[k]; d[k] = dn[k]; }
    }
}

template <int DEF, int YK, bool ROT, int MK>
__global__ __launch_bounds__(64) void k_hessian_weights_ep(cm_model_desc m, int64_t B, int K, int n_ep, const int32_t* __restrict__ ep_index,
        const double* __restrict__ gradu_hist, const double* __restrict__ xi_hist, const double* __restrict__ lam_hist,
        const double* __restrict__ sbar_hist, Wsq hss, const double* __restrict__ hss_hist, const double* __restrict__ hxx_hist,
        double* __restrict__ W) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU;
    const int NQ = 2 * NX + CM_NUM_PARAMS + n_ep, NPAIR = NQ * (NQ + 1) / 2;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * K * NPAIR) return;
    const int64_t ps = tid / NPAIR;
    const int64_t pt = ps % B;
    const int step = (int)(ps / B) + 1;
    int rem = (int)(tid % NPAIR), a = 0;
    while (rem >= NQ - a) { rem -= NQ - a; ++a; }
    const int b = a + rem;
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
    for (int r = 0; r < 6; ++r) { sbar[r] = sbar_hist[((int64_t)step * 6 + r) * B + pt]; hs[r] = hss_hist ? hss_hist[step * 6 + r] : hss.w[r]; }
    double w = hessian_weight<DEF, YK, ROT, MK>(m, G, x, xp, lam, sbar, hs, a, b, ep_index);
    if (hxx_hist && a == b && a < NX) w += hxx_hist[step * NX + a];
    W[(ps * NQ + a) * NQ + b] = w;
    W[(ps * NQ + b) * NQ + a] = w;
}

// part[ps][i][j] = sum_ab D[a][i] W[a][b] D[b][j] over NPT = 12 + n_ep parameter directions; dynamic LDS: W (NQ x NQ), D (NQ x NPT)
template <int NX>
__global__ __launch_bounds__(256) void k_hessian_quadform_ep(int64_t B, int K, int n_ep, const double* __restrict__ W,
        const double* __restrict__ dx_dp_hist, const double* __restrict__ dxe_hist, double* __restrict__ part) {
    extern __shared__ double sm[];
    constexpr int NP_ = CM_NUM_PARAMS;
    const int NPT = NP_ + n_ep, NQ = 2 * NX + NPT;
    double* sW = sm;
    double* sD = sm + NQ * NQ;
    const int64_t ps = blockIdx.x, pt = ps % B;
    const int step = (int)(ps / B) + 1;
    for (int i = threadIdx.x; i < NQ * NQ; i += 256) sW[i] = W[ps * NQ * NQ + i];
    for (int i = threadIdx.x; i < NQ * NPT; i += 256) {
        const int r = i / NPT, j = i % NPT;
        double v;
        if (r < 2 * NX) {
            const int st = (r < NX) ? step : step - 1, rr = (r < NX) ? r : r - NX;
            v = (j < NP_) ? dx_dp_hist[(((int64_t)st * NX + rr) * NP_ + j) * B + pt]
                          : dxe_hist[(((int64_t)st * NX + rr) * n_ep + (j - NP_)) * B + pt];
        } else v = (r - 2 * NX == j) ? 1.0 : 0.0;
        sD[i] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < NPT * NPT; o += 256) {
        const int i = o / NPT, j = o % NPT;
        double acc = 0.0;
        for (int a = 0; a < NQ; ++a) {
            double t = 0.0;
            for (int b = 0; b < NQ; ++b) t += sW[a * NQ + b] * sD[b * NPT + j];
            acc += sD[a * NPT + i] * t;
        }
        part[ps * (NPT * NPT) + o] = acc;
    }
}

template <int NX>
__global__ __launch_bounds__(192) void k_hessian_quadform(int64_t B, int K, const double* __restrict__ W,
        const double* __restrict__ dx_dp_hist, double* __restrict__ part) {
    constexpr int NP_ = CM_NUM_PARAMS, NQ = 2 * NX + NP_;
    __shared__ double sW[NQ * NQ], sD[NQ * NP_];
    const int64_t ps = blockIdx.x, pt = ps % B;
    const int step = (int)(ps / B) + 1;
    for (int i = threadIdx.x; i < NQ * NQ; i += 192) sW[i] = W[ps * NQ * NQ + i];
    for (int i = threadIdx.x; i < NQ * NP_; i += 192) {
        const int r = i / NP_, j = i % NP_;
        double v;
        if (r < NX) v = dx_dp_hist[(((int64_t)step * NX + r) * NP_ + j) * B + pt];
        else if (r < 2 * NX) v = dx_dp_hist[(((int64_t)(step - 1) * NX + (r - NX)) * NP_ + j) * B + pt];
        else v = (r - 2 * NX == j) ? 1.0 : 0.0;
        sD[i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NP_ * NP_) {
        const int i = threadIdx.x / NP_, j = threadIdx.x % NP_;
        double acc = 0.0;
        for (int a = 0; a < NQ; ++a) {
            double t = 0.0;
            for (int b = 0; b < NQ; ++b) t += sW[a * NQ + b] * sD[b * NP_ + j];
            acc += sD[a * NP_ + i] * t;
        }
        part[ps * (NP_ * NP_) + threadIdx.x] = acc;
    }
}

// out[j] = sum over rows of part[row][ncols] in a fixed order (one thread per column)
#if CM_HAS_PART(1) && !CM_HNN_VARIANT
__global__ __launch_bounds__(256) void k_sum_rows(const double* __restrict__ part, int64_t nrows, int ncols, double* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double acc = 0.0;
    for (int64_t r = 0; r < nrows; ++r) acc += part[r * ncols + j];
    out[j] = acc;
}
#endif

// ---- extended parameter sensitivities: forward-mode evaluation of the whole model (cm::param_direction) ----------------
// surfaces the arithmetic-T model (cm_hessian.hpp) covers: all of them (Barlat through a Jacobi eigen-decomposition in arithmetic T)
constexpr bool has_generic_eval(int) { return true; }

// cm_param_blocks: one thread per (point, requested parameter): dC/dp_e [n_xi] and d sigma/dp_e [6]
template <int DEF, int YK, int MK>
__global__ __launch_bounds__(64) void k_param_blocks(cm_model_desc m, int64_t B, int n_ep, const int32_t* __restrict__ ep_index,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev, const double* __restrict__ xi_prev,
        const double* __restrict__ xi, double* __restrict__ dC, double* __restrict__ dS) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * n_ep) return;
    const int64_t pt = tid / n_ep;
    const int j = (int)(tid % n_ep);
    double G[NU], xp[NX], x[NX], oC[NX], oS[6];
    for (int k = 0; k < NU; ++k) {
        G[k] = gradu[(int64_t)k * B + pt];
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) G[k] -= gradu_prev[(int64_t)k * B + pt];
    }
    for (int k = 0; k < NX; ++k) { xp[k] = xi_prev[(int64_t)k * B + pt]; x[k] = xi[(int64_t)k * B + pt]; }
    param_direction<DEF, YK, MK>(m, G, x, xp, ep_index[j], oC, oS);
    if (dC) for (int k = 0; k < NX; ++k) dC[((int64_t)j * NX + k) * B + pt] = oC[k];
    if (dS) for (int k = 0; k < 6; ++k) dS[((int64_t)j * 6 + k) * B + pt] = oS[k];
}

// cm_update_complex: the local Newton solve of a complex-step model instance (cm::newton_cx), one thread per point.  Complex
// arrays are (2, rows, B): the real rows, then the imaginary rows.
struct ParamImag { double v[CM_NUM_PARAMS]; };
template <int DEF, int YK, int MK>
__global__ __launch_bounds__(64) void k_update_cx(cm_model_desc m, int64_t B, ParamImag pim, const double* __restrict__ ext_im,
        const double* __restrict__ gradu, const double* __restrict__ gradu_prev, const double* __restrict__ xi_prev, double* __restrict__ xi,
        double* __restrict__ residual, double* __restrict__ sigma, uint32_t* __restrict__ status) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU;
    const int64_t pt = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (pt >= B) return;
    double G[NU];
    CX xp[NX], x[NX], C[NX], sg[6];
    for (int k = 0; k < NU; ++k) {
        G[k] = gradu[(int64_t)k * B + pt];
        if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) G[k] -= gradu_prev[(int64_t)k * B + pt];
    }
    for (int k = 0; k < NX; ++k) {
        xp[k] = CX{xi_prev[(int64_t)k * B + pt], xi_prev[(int64_t)(NX + k) * B + pt]};
        x[k] = CX{xi[(int64_t)k * B + pt], xi[(int64_t)(NX + k) * B + pt]};
    }
    const uint32_t st = newton_cx<DEF, YK, MK>(m, pim.v, ext_im, G, xp, x, C, sg);
    for (int k = 0; k < NX; ++k) { xi[(int64_t)k * B + pt] = x[k].re; xi[(int64_t)(NX + k) * B + pt] = x[k].im; }
    if (residual) for (int k = 0; k < NX; ++k) { residual[(int64_t)k * B + pt] = C[k].re; residual[(int64_t)(NX + k) * B + pt] = C[k].im; }
    if (sigma) for (int k = 0; k < 6; ++k) { sigma[(int64_t)k * B + pt] = sg[k].re; sigma[(int64_t)(6 + k) * B + pt] = sg[k].im; }
    if (status) status[pt] = st;
}

// cm_param_adjoint_history: rows[pt][j] = sum_k sbar_k . d sigma_k/dp_e - lam_k . dC_k/dp_e over the stored history
template <int DEF, int YK, int MK>
__global__ __launch_bounds__(64) void k_param_adjoint_history(cm_model_desc m, int64_t B, int K, int n_ep,
        const int32_t* __restrict__ ep_index, const double* __restrict__ gradu_hist, const double* __restrict__ xi_hist,
        const double* __restrict__ lam_hist, const double* __restrict__ sbar_hist, double* __restrict__ rows) {
    constexpr int NX = nx_of<DEF, MK>(), NU = Dims<DEF>::NU;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (tid >= B * n_ep) return;
    const int64_t pt = tid / n_ep;
    const int j = (int)(tid % n_ep);
    const int e = ep_index[j];
    double acc = 0.0;
    for (int step = 1; step <= K; ++step) {
        double G[NU], xp[NX], x[NX], oC[NX], oS[6];
        for (int k = 0; k < NU; ++k) {
            G[k] = gradu_hist[((int64_t)step * NU + k) * B + pt];
            if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) G[k] -= gradu_hist[((int64_t)(step - 1) * NU + k) * B + pt];
        }
        for (int k = 0; k < NX; ++k) {
            xp[k] = xi_hist[((int64_t)(step - 1) * NX + k) * B + pt];
            x[k] = xi_hist[((int64_t)step * NX + k) * B + pt];
        }
        param_direction<DEF, YK, MK>(m, G, x, xp, e, oC, oS);
        for (int k = 0; k < NX; ++k) acc -= lam_hist[((int64_t)step * NX + k) * B + pt] * oC[k];
        for (int r = 0; r < 6; ++r) acc += sbar_hist[((int64_t)step * 6 + r) * B + pt] * oS[r];
    }
    rows[pt * n_ep + j] = acc;
}

// ---- dispatch --------------------------------------------------------------------------------------
inline int64_t nblocks_of(int64_t B) { return (B + kBlock - 1) / kBlock; }
// what the reducing entry points need of their workspace (cm_workspace_bytes adds the screened update's list behind it)
inline int64_t reduce_workspace_bytes(int64_t B) {
    const int64_t nb = B <= 0 ? 1 : nblocks_of(B);
    return (nb + kRedBlocks + 1) * kRed * (int64_t)sizeof(double);   // block partials + stage rows + one result row
}

inline bool supported(const cm_model_desc* m, int model_kind = CM_SMALL_ELASTIC_PLASTIC) {
    if (m->model_kind != model_kind) return false;
    if (m->def_type != CM_FULL_3D && m->def_type != CM_PLANE_STRESS && m->def_type != CM_UNIAXIAL_STRESS) return false;
    if (m->def_type == CM_UNIAXIAL_STRESS && (m->uniaxial_idx < 0 || m->uniaxial_idx > 2)) return false;
    if (m->hnn_width < 0 || (m->hnn_width > 0 && (!m->nn_weights || m->hnn_offset < 0))) return false;   // network hardening law
    if (m->hnn_width > 0 && m->hnn_nhidden >= 2) {      // several hidden layers: the general forward pass (EXT build)
        if (m->hnn_nhidden > kHnnMaxHidden || m->hnn_widths[0] != m->hnn_width) return false;
        int units = 0;
        for (int l = 0; l < m->hnn_nhidden; ++l) { if (m->hnn_widths[l] < 1) return false; units += m->hnn_widths[l]; }
        if (units > kHnnMaxUnits) return false;
    }
    // the EXT build: J2 / Hill / Hosford (+ the network hardening law) and the plain hybrid surface (+ multi-layer networks)
    if (CM_HNN_VARIANT && m->yield_kind == CM_YIELD_BARLAT) return false;      // (the EXT build: network features only)
    if (m->yield_kind == CM_YIELD_SCALED_HYBRID_HILL_NN && !(m->beta_equivalent_stress > 0.0 && m->beta_max_iters >= 0)) return false;
    if (is_nn_yield(m->yield_kind)) {                  // weights resident on the device
        if (!m->nn_weights || m->nn_widths[0] != 6) return false;
        if (m->nn_nlayers == 3)                        // one hidden layer [6, H, 1]: the fast evaluation (either build)
            return m->nn_widths[2] == 1 && m->nn_widths[1] >= 1 && m->nn_widths[1] <= 256;
        // more hidden layers: the general evaluation of the EXT build, plain hybrid surface only
        if (!CM_HNN_VARIANT || m->nn_nlayers < 3 || m->nn_nlayers > kIcnnMaxLayers) return false;
        int units = 0;
        for (int k = 1; k + 1 < m->nn_nlayers; ++k) { if (m->nn_widths[k] < 1) return false; units += m->nn_widths[k]; }
        return m->nn_widths[m->nn_nlayers - 1] == 1 && units <= kIcnnMaxUnits;
    }
    if (m->yield_kind == CM_YIELD_BARLAT) return m->yc[18] >= 1.0;
    if (m->yield_kind != CM_YIELD_J2 && m->yield_kind != CM_YIELD_HILL && m->yield_kind != CM_YIELD_HOSFORD) return false;
    return true;
}

// Rate-form model x dense yield surfaces (Barlat, the network surfaces).  Round 3 builds them for every entry point
// (small_rate_elastic_plastic.py:116-126 takes any effective_stress_fun); -DCM_RATE_DENSE=0 / -DCM_RATE_UNIAXIAL_DENSE=0 leave
// them out of a build (smaller library), in which case the entry points refuse the combination at run time BEFORE any launch
// -- an entry point never returns CM_OK without having written its outputs.
#ifndef CM_RATE_DENSE
#define CM_RATE_DENSE 1             // reverse / history / direct / second-order / extended-parameter entries
#endif
#ifndef CM_RATE_UNIAXIAL_DENSE
#define CM_RATE_UNIAXIAL_DENSE 1    // the 12-dof UNIAXIAL_STRESS rate form (forward-mode blocks of the arithmetic-T model)
#endif
template <int MK, int Y>
constexpr bool has_rate_dense() { return CM_RATE_DENSE != 0 || !(MK == CM_SMALL_RATE_ELASTIC_PLASTIC && is_dense_yield(Y)); }
template <int D, int Y>
constexpr bool has_rate_uniaxial_dense() { return CM_RATE_UNIAXIAL_DENSE != 0 || !(D == CM_UNIAXIAL_STRESS && is_dense_yield(Y)); }
inline bool rate_dense(const cm_model_desc* m, int model_kind) {
    return CM_RATE_DENSE == 0 && model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC && is_dense_yield(m->yield_kind);
}
inline bool rate_uniaxial_dense(const cm_model_desc* m) {
    return CM_RATE_UNIAXIAL_DENSE == 0 && m->def_type == CM_UNIAXIAL_STRESS && is_dense_yield(m->yield_kind);
}

// calls F.template operator()<DEF, YK, ROT>() for the runtime (def_type, yield_kind, rotation) triple
// returns false when no specialisation exists (the caller reports CM_ERR_UNSUPPORTED -- never a silent no-op)
// UNIAXIAL_STRESS is built for the total-form entries and cm_hessians_rate (`UNI` = the caller has those specialisations)
// ROTP: which (def_type, yield) pairs get a Q = I specialisation at all.  0: all of them; 1: the memory- and issue-bound ones
// only (J2 / Hill / Hosford under FULL_3D / PLANE_STRESS) -- the dense surfaces and UNIAXIAL_STRESS always run the rotation
// products; 2: none (the rate form, whose dense LU dwarfs them).  With Q = I the rotation products reproduce the plain
// result exactly (products with 1, sums with 0), so this only trades a few instructions for a smaller library.
template <int ROTP, int D, int Y>
constexpr bool always_rotates() { return CM_HNN_VARIANT != 0 || ROTP == 2 || (ROTP == 1 && (is_dense_yield(Y) || D == CM_UNIAXIAL_STRESS)); }   // (the HNN build keeps one variant per configuration)

// Which configurations get a plain-Newton (LS = false) specialisation at all: J2 / Hill / Hosford of the total-form model in the
// base build, in the material frame (Q = I) or under UNIAXIAL_STRESS -- the memory- and issue-bound ones, whose plain kernels
// the line-search bookkeeping would cost registers (Hosford a = 8 update + vjp: -10 % through the LS = true kernel).  Everywhere
// else (the dense surfaces, rotated frames, the rate form, the HNN build) the LS = true kernel serves both: its Newton loops take
// the full step without a merit test when ls_max_evals == 0 (uniform branch; same iterates as the LS = false code, bit for bit:
// tests/test_host_math.py::test_plain_newton_through_the_line_search_kernels).
template <int ROTP, int D, int Y, bool R>
constexpr bool always_searches() {
    return CM_HNN_VARIANT != 0 || ROTP == 2 || is_dense_yield(Y) || (R && D != CM_UNIAXIAL_STRESS);
}

template <bool UNI = false, int ROTP = 0, class F>
inline bool dispatch(const cm_model_desc* m, F&& f) {
    const bool rot = !m->rotation_is_identity, ls = m->ls_max_evals > 0;
#define CM_CASE_LS(D, Y, R) \
    if constexpr (always_searches<ROTP, D, Y, R>()) f.template operator()<D, Y, R, true>(); \
    else { if (ls) f.template operator()<D, Y, R, true>(); else f.template operator()<D, Y, R, false>(); }
#define CM_CASE(D, Y) \
    if (m->def_type == D && m->yield_kind == Y) { \
        if constexpr (always_rotates<ROTP, D, Y>()) { \
            CM_CASE_LS(D, Y, true) \
        } else { \
            if (rot) { CM_CASE_LS(D, Y, true) } \
            else { CM_CASE_LS(D, Y, false) } \
        } \
        return true; }
    CM_CASE(CM_FULL_3D, CM_YIELD_J2)
    CM_CASE(CM_FULL_3D, CM_YIELD_HILL)
    CM_CASE(CM_FULL_3D, CM_YIELD_HOSFORD)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_J2)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_HILL)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_HOSFORD)
    CM_CASE(CM_FULL_3D, CM_YIELD_HYBRID_HILL_NN)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_HYBRID_HILL_NN)
    CM_CASE(CM_FULL_3D, CM_YIELD_SCALED_HYBRID_HILL_NN)      // (EXT build: with networks of several hidden layers, input_convex_neural_network.py:58-69)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_SCALED_HYBRID_HILL_NN)
#if !CM_HNN_VARIANT                    // the EXT build leaves Barlat out (supported() refuses it there)
    CM_CASE(CM_FULL_3D, CM_YIELD_BARLAT)
    CM_CASE(CM_PLANE_STRESS, CM_YIELD_BARLAT)
#endif
    if constexpr (UNI) {
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_J2)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_HILL)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_HOSFORD)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_HYBRID_HILL_NN)
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_SCALED_HYBRID_HILL_NN)
#if !CM_HNN_VARIANT
        CM_CASE(CM_UNIAXIAL_STRESS, CM_YIELD_BARLAT)
#endif
    }
#undef CM_CASE
#undef CM_CASE_LS
    return false;
}

// The point-wise entry points (cm_evaluate*, cm_hessians*, cm_direct_step: B = 1 in the reference's use, latency-bound)
// always run the rotation products: with Q = I they reproduce the unrotated result exactly (products with 1 and sums
// with 0), and one instantiation per (def_type, yield) instead of two keeps the library small.
constexpr bool kColdRot = true;

inline int check_launch() {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_cm_last_hip_error = (int)e; return CM_ERR_LAUNCH; }
    return CM_OK;
}

// wavefronts of a one-wave-per-workgroup kernel the device keeps resident (persistent grids of the work-pool kernels)
inline int pool_resident_waves(const void* kernel) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || cus <= 0 || per_cu <= 0) {
        (void)hipGetLastError();
        return 256 * 8;                                          // MI355X: 256 CUs, two waves per SIMD
    }
    return cus * per_cu;
}

static inline bool use_subspace_newton(const cm_model_desc* m) { return use_fast_newton(m); }      // cm_structured.hpp
// Iteration-bound configurations (the network surfaces, Hosford under the line search: pool_pays<>) run cm_update on the work
// pool.  The fused entry points below take the same route for them -- work-pool update, then the reverse sweep as a second
// kernel over the stored states -- instead of the lockstep fused kernel, whose wavefronts wait for their slowest point
// (CM_SOLVER_LOCKSTEP keeps the single fused kernel).  Same per-point arithmetic and the same reduction order either way.
// Hosford / FULL_3D with a large exponent starts the reference's Newton at an analytic warm start (cm::hosford_warm_start): one
// or two residual evaluations per point instead of 4-18, so nothing is left for the pool to balance -- lockstep kernels.
static inline bool hosford_warm_route(const cm_model_desc* m) {
    return CM_HNN_BUILD_HAS_SUBSPACE && m->model_kind == CM_SMALL_ELASTIC_PLASTIC && m->yield_kind == CM_YIELD_HOSFORD &&
           m->def_type == CM_FULL_3D && m->yc[0] >= kHosfordWarmMinA && use_subspace_newton(m);
}
static inline bool pool_route(const cm_model_desc* m, int64_t B) {
    if (!m || m->model_kind != CM_SMALL_ELASTIC_PLASTIC || (m->solver_flags & CM_SOLVER_LOCKSTEP) || B < 256) return false;
    return is_nn_yield(m->yield_kind) || (m->yield_kind == CM_YIELD_HOSFORD && m->ls_max_evals > 0 && !hosford_warm_route(m)) ||
           (CM_POOL_HILL && m->yield_kind == CM_YIELD_HILL);
}
// ---- consistent tangent at given converged states (second kernel of cm_update_tangent's work-pool route) -------------------
template <int DEF, int YK, bool ROT>
__global__ __launch_bounds__(kBlock) void k_tangent_state(cm_model_desc m, int64_t B, const double* __restrict__ gradu,
        const double* __restrict__ xi_prev, const double* __restrict__ xi, double* __restrict__ dsig, uint32_t* __restrict__ status) {
    constexpr int NX = Dims<DEF>::NX, NU = Dims<DEF>::NU;
    const int64_t blk0 = (int64_t)blockIdx.x * kBlock;
    const bool valid = blk0 + threadIdx.x < B;
    const unsigned b = valid ? threadIdx.x : (unsigned)(B - 1 - blk0);
    gradu += blk0; xi_prev += blk0; xi += blk0; dsig += blk0;
    if (status) status += blk0;
    double G[NU], xp[NX], x[NX], eg[6], z[Dims<DEF>::NZ];
    load_soa<NU>(gradu, B, b, G);
    load_soa<NX>(xi_prev, B, b, xp);
    load_soa<NX, false>(xi, B, b, x);                            // just written by the update kernel: a cached read
    strain_from_gradu<DEF, ROT>(m, G, eg);
    strain_z<DEF, ROT>(m, z);
    double T[6][6];
    const bool ok = tangent_any<DEF, YK>(m, eg, z, x, xp, T);
    if (!ok && valid && status) status[b] |= CM_STATUS_SINGULAR;
#pragma unroll
    for (int c = 0; c < NU; ++c) {
        double Gd[NU], dm[6], t[6], tg[6];
#pragma unroll
        for (int k = 0; k < NU; ++k) Gd[k] = (k == c) ? 1.0 : 0.0;
        strain_from_gradu<DEF, ROT>(m, Gd, dm);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < 6; ++l) s += T[r][l] * dm[l];
            t[r] = s;
        }
        to_global<ROT>(m, t, tg);
        if (valid) {
#pragma unroll
            for (int r = 0; r < 6; ++r) (dsig + (int64_t)(r * NU + c) * B)[b] = tg[r];
        }
    }
}

// workspace of the screened route: [0, 8) the list length, [256, 256 + 4 B) the list of plastic point indices
constexpr int64_t kScreenListOffset = 256;
static inline int64_t screen_workspace_bytes(int64_t B) { return kScreenListOffset + 4 * (B > 0 ? B : 1); }
// Which configurations take the screened route (k_screen + k_update_listed) when the caller provides the workspace: FULL_3D,
// total form, an evaluation expensive enough to pay for the second pass over the plastic points' rows -- the network surfaces,
// Barlat, and Hosford on the reference's iteration (with the analytic warm start it is a two-evaluation kernel: lockstep).
static inline bool screen_route(const cm_model_desc* m, int64_t B, const void* ws, int64_t ws_bytes) {
    static const bool off = [] { const char* e = getenv("CM_DEBUG_NO_SCREEN"); return e && atoi(e) != 0; }();     // A/B against the work pool
    if (!CM_SCREEN || off || !m || !ws || ws_bytes < screen_workspace_bytes(B) || ((uintptr_t)ws & 7)) return false;
    if (m->model_kind != CM_SMALL_ELASTIC_PLASTIC || m->def_type != CM_FULL_3D || (m->solver_flags & CM_SOLVER_LOCKSTEP)) return false;
    if (B < 4096 || B >= ((int64_t)1 << 29)) return false;      // 32-bit byte offsets into the rows
    // (Hosford on the reference's iteration stays on the work pool: its plastic points take 3 to 18 passes, and a lockstep
    // wavefront over the list pays the maximum -- measured 2.63 ms screened against 1.39 ms on the pool, profiles/r04_sustained.txt)
    return is_dense_yield(m->yield_kind);
}

template <bool TANGENT>
int launch_update(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
                  double* xi, double* sigma, double* dsig, uint32_t* status, void* stream,
                  void* ws = nullptr, int64_t ws_bytes = 0) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    if (!supported(m)) return CM_ERR_UNSUPPORTED;
    if (B == 0) return CM_OK;                       // empty batch: nothing to read or write
    if (!gradu || !xi_prev || !xi || (TANGENT && !dsig)) return CM_ERR_BAD_ARG;
    const dim3 grid((unsigned)nblocks_of(B)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const cm_model_desc md = *m;
    (void)hipGetLastError();            // drop any stale error left by other users of the runtime (e.g. torch)
    if constexpr (TANGENT) {
        // iteration-bound configurations: work-pool update, then the tangent at the stored states as a second kernel
        if (pool_route(m, B)) {
            const int rc = cm_update_ws(m, B, gradu, xi_prev, xi, sigma, status, ws, ws_bytes, stream);
            if (rc != CM_OK) return rc;
            const bool found = dispatch<true, 1>(m, [&]<int D, int Y, bool R, bool LS>() {
                if constexpr (pool_pays<Y, true>())
                    hipLaunchKernelGGL((k_tangent_state<D, Y, R>), grid, block, 0, s, md, B, gradu, xi_prev, xi, dsig, status);
            });
            if (!found) return CM_ERR_UNSUPPORTED;
            return check_launch();
        }
    }
    if constexpr (!TANGENT) {
        if (screen_route(m, B, ws, ws_bytes)) {
            unsigned long long* const count = (unsigned long long*)ws;
            uint32_t* const list = (uint32_t*)((char*)ws + kScreenListOffset);
            hipLaunchKernelGGL(k_screen_reset, dim3(1), dim3(1), 0, s, count);
            const bool found = dispatch<true, 1>(m, [&]<int D, int Y, bool R, bool LS>() {
                if constexpr (D == CM_FULL_3D && screen_pays<Y>()) {
                    // (the dense surfaces have no Q = I kernels elsewhere -- dispatch hands R = true; these two small families do:
                    // the rotation products are 12 % of k_screen's instructions)
                    const dim3 sgrid((unsigned)((B + kScreenBlock - 1) / kScreenBlock)), sblock(kScreenBlock);
                    if (md.rotation_is_identity) {
                        hipLaunchKernelGGL((k_screen<Y, false>), sgrid, sblock, 0, s, md, B, gradu, xi_prev, xi, sigma, status, list, count);
                        hipLaunchKernelGGL((k_update_listed<Y, false, LS>), grid, block, 0, s, md, B, gradu, xi_prev, xi, sigma, status, list, count);
                    } else {
                        hipLaunchKernelGGL((k_screen<Y, true>), sgrid, sblock, 0, s, md, B, gradu, xi_prev, xi, sigma, status, list, count);
                        hipLaunchKernelGGL((k_update_listed<Y, true, LS>), grid, block, 0, s, md, B, gradu, xi_prev, xi, sigma, status, list, count);
                    }
                }
            });
            if (!found) return CM_ERR_UNSUPPORTED;
            return check_launch();
        }
    }
    const bool found = dispatch<true, 1>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (has_fast_newton<D, Y, LS>()) {
            if (use_subspace_newton(m)) {
                hipLaunchKernelGGL((k_update<D, Y, R, LS, TANGENT, true>), grid, block, 0, s, md, B, gradu, xi_prev, xi, sigma, dsig, status);
                return;
            }
        }
        if constexpr (!TANGENT && pool_pays<Y, LS>()) {
            // expensive, iteration-bound passes: the work-pool kernel (CM_SOLVER_LOCKSTEP: one point per lane as everywhere else)
            if (pool_route(m, B)) {
                static const int resident = pool_resident_waves((const void*)k_update_pool<D, Y, R, LS>);
                // Chunk = what a wavefront owns at a time (static round-robin).  32 points (one staged half) when every resident
                // wavefront gets at least 64 of them: with 256-point chunks 10^7 points are 39 062 chunks over 2048 wavefronts =
                // 19.07 each, so most of the grid idles while 7 % of the wavefronts run their 20th chunk (measured, round 3:
                // network surface +4.5 %, Hosford a = 100 +0.5-1 %, profiles/r03_pool_chunk_ab.txt).  Small batches: 64 points,
                // so that a wavefront with a single chunk still fills its lanes.
                // CM_DEBUG_POOL_DYNAMIC_MIN=<points per resident wavefront>: tests bring the dynamic assignment down to small batches
                static const int64_t dyn_min = [] { const char* e = getenv("CM_DEBUG_POOL_DYNAMIC_MIN"); return e ? (int64_t)atoll(e) : (int64_t)CM_POOL_DYNAMIC_MIN; }();
                bool dynamic = (CM_POOL_DYNAMIC != 0) && B >= (int64_t)resident * dyn_min;
                int ticket = -1;
                if (dynamic) {                                   // a zeroed counter of its own for this launch, stream-ordered
                    ticket = pool_ticket_slot(s);
                    if (ticket >= 0) hipLaunchKernelGGL(k_pool_ticket_zero, dim3(1), dim3(1), 0, s, ticket);
                    else dynamic = false;                        // out of counters: static assignment
                }
                int chunk_shift = (B >= (int64_t)resident * 2048) ? 5 : 6;
                if (dynamic) {                                   // the smallest chunk that keeps the launch under ~40 000 tickets (see k_update_pool)
                    chunk_shift = 7;
                    while ((B >> chunk_shift) > 40000 && chunk_shift < 16) ++chunk_shift;
                }
                const int64_t nchunks = (B + ((int64_t)1 << chunk_shift) - 1) >> chunk_shift;
                const unsigned nw = (unsigned)(nchunks < resident ? nchunks : resident);
                // 16-byte LDS-DMA pieces need every row start 16-byte aligned: both arrays, and an even row length
                const int wide = ((B & 1) == 0 && (((uintptr_t)gradu | (uintptr_t)xi_prev) & 15) == 0) ? 1 : 0;
                hipLaunchKernelGGL((k_update_pool<D, Y, R, LS>), dim3(nw), dim3(64), 0, s, md, B, chunk_shift, wide, ticket, gradu, xi_prev, xi, sigma, status);
                return;
            }
        }
        hipLaunchKernelGGL((k_update<D, Y, R, LS, TANGENT>), grid, block, 0, s, md, B, gradu, xi_prev, xi, sigma, dsig, status);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}

template <int MODE>
int launch_reverse(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi_in,
                   const double* sd, const double* wsq6, const double* hist_in, double* xi_out, double* sigma_out,
                   double* xpbar, double* gbar, double* out, int out_offset, int accumulate, void* workspace,
                   int64_t wbytes, void* stream) {
    if (!m || B < 0 || !out || !workspace) return CM_ERR_BAD_ARG;
    if (B > 0 && (!gradu || !xi_prev || !sd)) return CM_ERR_BAD_ARG;
    if (B > 0 && (MODE == 0 || MODE == 2) && !xi_in) return CM_ERR_BAD_ARG;
    if ((MODE == 1 || MODE == 2) && !wsq6) return CM_ERR_BAD_ARG;
    if (!supported(m)) return CM_ERR_UNSUPPORTED;
    if (wbytes < reduce_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partials = (double*)workspace;
    const int64_t nb = nblocks_of(B);
    Wsq w; for (int k = 0; k < 6; ++k) w.w[k] = wsq6 ? wsq6[k] : 0.0;
    const cm_model_desc md = *m;
    (void)hipGetLastError();            // drop any stale error left by other users of the runtime
    if (B > 0) {
        const dim3 grid((unsigned)nb), block(kBlock);
        // CM_DEBUG_DYN_LDS=<bytes>: occupancy experiments only (extra dynamic LDS per block limits blocks per CU)
        static const unsigned dyn_lds = [] { const char* e = getenv("CM_DEBUG_DYN_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
        const bool found = dispatch<true, 1>(m, [&]<int D, int Y, bool R, bool LS>() {
            if constexpr (has_fast_newton<D, Y, LS>() && (MODE == 1 || MODE == 3)) {
                if (use_subspace_newton(m)) {
                    hipLaunchKernelGGL((k_reverse<D, Y, R, LS, MODE, true>), grid, block, dyn_lds, s, md, B, gradu, xi_prev, xi_in, sd, w,
                                       hist_in, xi_out, sigma_out, xpbar, gbar, partials);
                    return;
                }
            }
            hipLaunchKernelGGL((k_reverse<D, Y, R, (MODE == 1 || MODE == 3) ? LS : false, MODE>), grid, block, dyn_lds, s, md, B, gradu, xi_prev, xi_in, sd, w,
                               hist_in, xi_out, sigma_out, xpbar, gbar, partials);
        });
        if (!found) return CM_ERR_UNSUPPORTED;
        if (check_launch() != CM_OK) return CM_ERR_LAUNCH;
    }
    double* stage = partials + nb * kRed;
    cm_detail_reduce(s, partials, B > 0 ? nb : 0, stage, out, out_offset, accumulate);
    return check_launch();
}

template <int MODE>
int launch_reverse_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                        const double* xi_prev, const double* xi_in, const double* sd, const double* wsq6,
                        const double* hist_in, double* xi_out, double* sigma_out, double* xpbar, double* gbar,
                        double* out, int out_offset, int accumulate, void* workspace, int64_t wbytes, void* stream) {
    if (!m || B < 0 || !out || !workspace) return CM_ERR_BAD_ARG;
    if (B > 0 && (!gradu || !gradu_prev || !xi_prev || !sd)) return CM_ERR_BAD_ARG;
    if (B > 0 && (MODE == 0 || MODE == 2) && !xi_in) return CM_ERR_BAD_ARG;
    if ((MODE == 1 || MODE == 2) && !wsq6) return CM_ERR_BAD_ARG;
    if (!supported(m, CM_SMALL_RATE_ELASTIC_PLASTIC) || rate_dense(m, CM_SMALL_RATE_ELASTIC_PLASTIC) || rate_uniaxial_dense(m)) return CM_ERR_UNSUPPORTED;
    if (wbytes < reduce_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partials = (double*)workspace;
    const int64_t nb = nblocks_of(B);
    Wsq w; for (int k = 0; k < 6; ++k) w.w[k] = wsq6 ? wsq6[k] : 0.0;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    if (B > 0) {
        const dim3 grid((unsigned)nb), block(kBlock);
        const bool found = dispatch<true, 2>(m, [&]<int D, int Y, bool R, bool LS>() {
            if constexpr (has_rate_dense<CM_SMALL_RATE_ELASTIC_PLASTIC, Y>() && has_rate_uniaxial_dense<D, Y>())
                hipLaunchKernelGGL((k_reverse_rate<D, Y, R, (MODE == 1 || MODE == 3) ? LS : false, MODE>), grid, block, 0, s, md, B,
                                   gradu, gradu_prev, xi_prev, xi_in, sd, w, hist_in, xi_out, sigma_out, xpbar, gbar, partials);
        });
        if (!found) return CM_ERR_UNSUPPORTED;
        if (check_launch() != CM_OK) return CM_ERR_LAUNCH;
    }
    double* stage = partials + nb * kRed;
    cm_detail_reduce(s, partials, B > 0 ? nb : 0, stage, out, out_offset, accumulate);
    return check_launch();
}

template <int MK>
int launch_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* data_hist,
                   const double* wsq6, const double* xi0, double* xi_hist, double* out,
                   void* workspace, int64_t wbytes, void* stream, HistoryCotangents hc = HistoryCotangents{nullptr, nullptr, nullptr},
                   int out_offset = 0) {
    if (!m || B < 0 || K < 1 || !out || !workspace || (!wsq6 && !hc.sbar_hist)) return CM_ERR_BAD_ARG;
    if (B > 0 && (!gradu_hist || (!data_hist && !hc.sbar_hist) || !xi0 || !xi_hist)) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m))) return CM_ERR_UNSUPPORTED;
    if (wbytes < reduce_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partials = (double*)workspace;
    const int64_t nb = nblocks_of(B);
    Wsq w; for (int k = 0; k < 6; ++k) w.w[k] = wsq6 ? wsq6[k] : 0.0;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    if (B > 0) {
        const dim3 grid((unsigned)nb), block(kBlock);
        const bool found = dispatch<true, (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) ? 2 : 1>(m, [&]<int D, int Y, bool R, bool LS>() {
            if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC && has_fast_newton<D, Y, LS>()) {
                if (use_subspace_newton(m)) {
                    hipLaunchKernelGGL((k_history<D, Y, R, LS, MK, true>), grid, block, 0, s, md, B, K, gradu_hist, data_hist, w, xi0, xi_hist, partials, hc);
                    return;
                }
            }
            if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>()))
                hipLaunchKernelGGL((k_history<D, Y, R, LS, MK>), grid, block, 0, s, md, B, K, gradu_hist, data_hist, w, xi0, xi_hist, partials, hc);
        });
        if (!found) return CM_ERR_UNSUPPORTED;
        if (check_launch() != CM_OK) return CM_ERR_LAUNCH;
    }
    double* stage = partials + nb * kRed;
    cm_detail_reduce(s, partials, B > 0 ? nb : 0, stage, out, out_offset, 0);
    return check_launch();
}

template <int MK>
int launch_primal_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* xi0,
                          double* xi_hist, double* sigma_hist, uint32_t* status_hist, void* stream) {
    if (!m || B < 0 || K < 1) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m))) return CM_ERR_UNSUPPORTED;
    if (B == 0) return CM_OK;
    if (!gradu_hist || !xi0 || (!xi_hist && !sigma_hist)) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const dim3 grid((unsigned)nblocks_of(B)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const bool found = dispatch<true, (MK == CM_SMALL_RATE_ELASTIC_PLASTIC) ? 2 : 1>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC && has_fast_newton<D, Y, LS>()) {
            if (use_subspace_newton(m)) {
                hipLaunchKernelGGL((k_primal_history<D, Y, R, LS, MK, true>), grid, block, 0, s, md, B, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist);
                return;
            }
        }
        if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>()))
            hipLaunchKernelGGL((k_primal_history<D, Y, R, LS, MK>), grid, block, 0, s, md, B, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}

template <int MK>
int launch_direct_step(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev,
                       const double* xi, const double* dxp_dp, double* dx_dp, double* ds_dp, void* stream) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m)))
        return CM_ERR_UNSUPPORTED;                  // never a silent no-op: the dispatch below has no such specialisation
    if (B == 0) return CM_OK;
    if (!gradu || !xi_prev || !xi || !dx_dp || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && !gradu_prev)) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const dim3 grid((unsigned)((B + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>())) {
            if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && D == CM_UNIAXIAL_STRESS)
                hipLaunchKernelGGL((k_direct_step<D, Y, kColdRot, MK>), grid, block, 0, s, md, B, gradu, gradu_prev, xi_prev, xi, dxp_dp, dx_dp, ds_dp);
            else
                hipLaunchKernelGGL((k_direct_step_cols<D, Y, kColdRot, MK>), dim3((unsigned)((B * CM_NUM_PARAMS + 63) / 64)), block, 0, s,
                                   md, B, gradu, gradu_prev, xi_prev, xi, dxp_dp, dx_dp, ds_dp);
        }
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}

template <int MK>
int launch_direct_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* xi_hist,
                          const double* sbar_hist, const double* xibar_hist, double* dx_dp_hist, double* ds_dp_hist,
                          double* grad_p, void* workspace, int64_t wbytes, void* stream) {
    if (!m || B < 0 || K < 1) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m)))
        return CM_ERR_UNSUPPORTED;
    if (grad_p && (!sbar_hist || !workspace)) return CM_ERR_BAD_ARG;
    if (grad_p && wbytes < cm_direct_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    if (!grad_p && !dx_dp_hist && !ds_dp_hist) return CM_ERR_BAD_ARG;
    if (B > 0 && (!gradu_hist || !xi_hist)) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    hipStream_t s = (hipStream_t)stream;
    double* rows = grad_p ? (double*)workspace : nullptr;
    if (B > 0) {
        const dim3 grid((unsigned)((B + 63) / 64)), block(64);
        const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
            if constexpr (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>())) {
                if constexpr (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && D == CM_UNIAXIAL_STRESS)
                    hipLaunchKernelGGL((k_direct_history<D, Y, kColdRot, MK>), grid, block, 0, s, md, B, K, gradu_hist, xi_hist,
                                       sbar_hist, xibar_hist, dx_dp_hist, ds_dp_hist, rows);
                else
                    hipLaunchKernelGGL((k_direct_history_cols<D, Y, kColdRot, MK>), dim3((unsigned)((B * CM_NUM_PARAMS + 63) / 64)), block, 0, s,
                                       md, B, K, gradu_hist, xi_hist, sbar_hist, xibar_hist, dx_dp_hist, ds_dp_hist, rows);
            }
        });
        if (!found) return CM_ERR_UNSUPPORTED;
        if (check_launch() != CM_OK) return CM_ERR_LAUNCH;
    }
    if (grad_p) {
        double* stage = rows + (B > 0 ? B : 1) * kRed;
        cm_detail_reduce(s, rows, B, stage, grad_p, 1, 0);
    }
    return check_launch();
}

#if CM_HAS_PART(6)
template <int MK>
int launch_hessians(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                    const double* xi_prev, const double* xi,
                    double* d2C, double* d2S, double* dC, double* dS, double* C0, double* S0, void* stream) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || !has_generic_eval(m->yield_kind) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m))) return CM_ERR_UNSUPPORTED;
    if (B == 0) return CM_OK;
    if (!gradu || !xi_prev || !xi || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && !gradu_prev)) return CM_ERR_BAD_ARG;
    const int nx = cm_num_xi(m), nq = 2 * nx + CM_NUM_PARAMS;
    const int64_t nthreads = B * (int64_t)(nq * (nq + 1) / 2);
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const dim3 grid((unsigned)((nthreads + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (has_generic_eval(Y) && (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>())))
            hipLaunchKernelGGL((k_hessians<D, CM_YIELD_ANY, kColdRot, MK>), grid, block, 0, s, md, B, gradu, gradu_prev, xi_prev, xi,
                               d2C, d2S, dC, dS, C0, S0);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}

template <int MK>
int launch_hessian_history(const cm_model_desc* m, int64_t B, int K, const double* gradu_hist, const double* xi_hist,
                           const double* lam_hist, const double* dx_dp_hist, const double* sbar_hist, const double* hss6,
                           const double* hss_hist, const double* hxx_hist,
                           double* out, void* workspace, int64_t wbytes, void* stream) {
    if (!m || B < 0 || K < 1 || !out || !workspace || (!hss6 && !hss_hist)) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || !has_generic_eval(m->yield_kind) || rate_dense(m, MK) ||
        (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m)))
        return CM_ERR_UNSUPPORTED;
    if (B > 0 && (!gradu_hist || !xi_hist || !lam_hist || !dx_dp_hist || !sbar_hist)) return CM_ERR_BAD_ARG;
    if (wbytes < cm_hessian_workspace_bytes(m, B, K)) return CM_ERR_WORKSPACE;
    const int nx = cm_num_xi(m), nq = 2 * nx + CM_NUM_PARAMS;
    constexpr int NPP = CM_NUM_PARAMS * CM_NUM_PARAMS;
    const int64_t nps = B * (int64_t)K;
    double* W = (double*)workspace;
    double* part = W + nps * nq * nq;
    Wsq h; for (int k = 0; k < 6; ++k) h.w[k] = hss6 ? hss6[k] : 0.0;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    hipStream_t s = (hipStream_t)stream;
    if (nps > 0) {
        const int64_t nthreads = nps * (int64_t)(nq * (nq + 1) / 2);
        const dim3 grid((unsigned)((nthreads + 63) / 64)), block(64);
        const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
            if constexpr (has_generic_eval(Y) && (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>()))) {
                hipLaunchKernelGGL((k_hessian_weights<D, CM_YIELD_ANY, kColdRot, MK>), grid, block, 0, s, md, B, K, gradu_hist, xi_hist, lam_hist,
                                   sbar_hist, h, hss_hist, hxx_hist, W);
                hipLaunchKernelGGL((k_hessian_quadform<nx_of<D, MK>()>), dim3((unsigned)nps), dim3(192), 0, s, B, K, W, dx_dp_hist, part);
            }
        });
        if (!found) return CM_ERR_UNSUPPORTED;
        if (check_launch() != CM_OK) return CM_ERR_LAUNCH;
    }
    cm_detail_sum_rows(s, part, nps, NPP, out);
    return check_launch();
}
#endif


#if CM_HAS_PART(9)
constexpr int kMaxEp = 64;           // extended parameter directions per second-order pass (dynamic LDS of the quadratic form)
template <int MK>
int launch_direct_history_ep(const cm_model_desc* m, int64_t B, int K, int n_ep, const int32_t* ep_index, const double* gradu_hist,
                             const double* xi_hist, double* dxe_hist, void* stream) {
    if (!m || B < 0 || K < 1 || n_ep < 0 || n_ep > kMaxEp) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && m->def_type == CM_UNIAXIAL_STRESS)) return CM_ERR_UNSUPPORTED;
    if (B == 0 || n_ep == 0) return CM_OK;
    if (!ep_index || !gradu_hist || !xi_hist || !dxe_hist) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((B * n_ep + 63) / 64)), block(64);
    const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (!(MK == CM_SMALL_RATE_ELASTIC_PLASTIC && D == CM_UNIAXIAL_STRESS) && has_rate_dense<MK, Y>())
            hipLaunchKernelGGL((k_direct_history_ep<D, Y, kColdRot, MK>), grid, block, 0, s, md, B, K, n_ep, ep_index, gradu_hist, xi_hist, dxe_hist);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}

template <int MK>
int launch_hessian_history_ep(const cm_model_desc* m, int64_t B, int K, int n_ep, const int32_t* ep_index, const double* gradu_hist,
                              const double* xi_hist, const double* lam_hist, const double* dx_dp_hist, const double* dxe_hist,
                              const double* sbar_hist, const double* hss6, const double* hss_hist, const double* hxx_hist,
                              double* out, void* workspace, int64_t wbytes, void* stream) {
    if (!m || B < 0 || K < 1 || n_ep < 0 || n_ep > kMaxEp || !out || !workspace || (!hss6 && !hss_hist)) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && m->def_type == CM_UNIAXIAL_STRESS)) return CM_ERR_UNSUPPORTED;
    if (B > 0 && (!gradu_hist || !xi_hist || !lam_hist || !dx_dp_hist || !sbar_hist || (n_ep > 0 && (!ep_index || !dxe_hist)))) return CM_ERR_BAD_ARG;
    if (wbytes < cm_hessian_ep_workspace_bytes(m, B, K, n_ep)) return CM_ERR_WORKSPACE;
    const int nx = cm_num_xi(m), npt = CM_NUM_PARAMS + n_ep, nq = 2 * nx + npt;
    const int64_t nps = B * (int64_t)K;
    double* W = (double*)workspace;
    double* part = W + nps * nq * nq;
    Wsq h; for (int k = 0; k < 6; ++k) h.w[k] = hss6 ? hss6[k] : 0.0;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    hipStream_t s = (hipStream_t)stream;
    if (nps > 0) {
        const int64_t nthreads = nps * (int64_t)(nq * (nq + 1) / 2);
        const dim3 grid((unsigned)((nthreads + 63) / 64)), block(64);
        const size_t lds = (size_t)(nq * nq + nq * npt) * sizeof(double);
        // The quadratic form's tile lives in dynamic LDS: (nq^2 + nq npt) doubles = 117 KB at n_ep = 64 (FULL_3D), above the 64 KB
        // a kernel may use without asking.  Ask (gfx950 has 160 KB per CU); a device that refuses gets CM_ERR_UNSUPPORTED instead of
        // a launch failure.
        int dev = 0, lds_max = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); lds_max = 65536; }
        bool lds_ok = true;
        const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
            if constexpr (!(MK == CM_SMALL_RATE_ELASTIC_PLASTIC && D == CM_UNIAXIAL_STRESS) && has_rate_dense<MK, Y>()) {
                if (lds > 65536) {
                    if (hipFuncSetAttribute((const void*)k_hessian_quadform_ep<nx_of<D, MK>()>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds) != hipSuccess) { (void)hipGetLastError(); lds_ok = false; return; }
                } else if ((int64_t)lds > (int64_t)lds_max) { lds_ok = false; return; }
                hipLaunchKernelGGL((k_hessian_weights_ep<D, CM_YIELD_ANY, kColdRot, MK>), grid, block, 0, s, md, B, K, n_ep, ep_index, gradu_hist,
                                   xi_hist, lam_hist, sbar_hist, h, hss_hist, hxx_hist, W);
                hipLaunchKernelGGL((k_hessian_quadform_ep<nx_of<D, MK>()>), dim3((unsigned)nps), dim3(256), lds, s, B, K, n_ep, W, dx_dp_hist,
                                   dxe_hist, part);
            }
        });
        if (!found || !lds_ok) return CM_ERR_UNSUPPORTED;
        if (check_launch() != CM_OK) return CM_ERR_LAUNCH;
    }
    cm_detail_sum_rows(s, part, nps, npt * npt, out);
    return check_launch();
}

#endif
#if CM_HAS_PART(10)                      // (parts of their own: the arithmetic-T kernels are the longest compiles)
template <int MK>
int launch_param_blocks(const cm_model_desc* m, int64_t B, int n_ep, const int32_t* ep_index, const double* gradu,
                        const double* gradu_prev, const double* xi_prev, const double* xi, double* dC, double* dS, void* stream) {
    if (!m || B < 0 || n_ep < 0) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || !has_generic_eval(m->yield_kind) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m))) return CM_ERR_UNSUPPORTED;
    if (B == 0 || n_ep == 0) return CM_OK;
    if (!ep_index || !gradu || !xi_prev || !xi || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && !gradu_prev)) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const dim3 grid((unsigned)((B * n_ep + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (has_generic_eval(Y) && (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>())))
            hipLaunchKernelGGL((k_param_blocks<D, CM_YIELD_ANY, MK>), grid, block, 0, s, md, B, n_ep, ep_index, gradu, gradu_prev, xi_prev, xi, dC, dS);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}

#endif
#if CM_HAS_PART(11)                      // (complex arithmetic x every surface: a part of its own)
template <int MK>
int launch_update_complex(const cm_model_desc* m, int64_t B, const double* p_im, const double* ext_im, const double* gradu,
                          const double* gradu_prev, const double* xi_prev, double* xi, double* residual, double* sigma, uint32_t* status,
                          void* stream) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    // the reference runs its complex-step checks on the J2 analytical problem; every surface continues analytically the same
    // way (Barlat's Jacobi rotations and the |.| of its differences are decided on real parts, softplus on the real part of its
    // argument)
    if (!supported(m, MK) || (is_dense_yield(m->yield_kind) && (rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m)))))
        return CM_ERR_UNSUPPORTED;
    if (B == 0) return CM_OK;
    if (!p_im || !gradu || !xi_prev || !xi || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && !gradu_prev)) return CM_ERR_BAD_ARG;
    ParamImag pim;
    for (int k = 0; k < CM_NUM_PARAMS; ++k) pim.v[k] = p_im[k];
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const dim3 grid((unsigned)((B + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (has_generic_eval(Y) && (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>())))
            hipLaunchKernelGGL((k_update_cx<D, CM_YIELD_ANY, MK>), grid, block, 0, s, md, B, pim, ext_im, gradu, gradu_prev, xi_prev, xi, residual, sigma, status);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}

#endif
#if CM_HAS_PART(10)
template <int MK>
int launch_param_adjoint_history(const cm_model_desc* m, int64_t B, int K, int n_ep, const int32_t* ep_index,
                                 const double* gradu_hist, const double* xi_hist, const double* lam_hist, const double* sbar_hist,
                                 double* grad_ep, void* workspace, int64_t wbytes, void* stream) {
    if (!m || B < 0 || K < 1 || n_ep < 0 || !grad_ep) return CM_ERR_BAD_ARG;
    if (!supported(m, MK) || !has_generic_eval(m->yield_kind) || rate_dense(m, MK) || (MK == CM_SMALL_RATE_ELASTIC_PLASTIC && rate_uniaxial_dense(m))) return CM_ERR_UNSUPPORTED;
    if (n_ep == 0) return CM_OK;
    if (!workspace || wbytes < (B > 0 ? B : 1) * (int64_t)n_ep * (int64_t)sizeof(double)) return CM_ERR_WORKSPACE;
    if (!ep_index || (B > 0 && (!gradu_hist || !xi_hist || !lam_hist || !sbar_hist))) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    hipStream_t s = (hipStream_t)stream;
    double* rows = (double*)workspace;
    if (B > 0) {
        const dim3 grid((unsigned)((B * n_ep + 63) / 64)), block(64);
        const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
            if constexpr (has_generic_eval(Y) && (MK == CM_SMALL_ELASTIC_PLASTIC || (has_rate_dense<MK, Y>() && has_rate_uniaxial_dense<D, Y>())))
                hipLaunchKernelGGL((k_param_adjoint_history<D, CM_YIELD_ANY, MK>), grid, block, 0, s, md, B, K, n_ep, ep_index, gradu_hist, xi_hist,
                                   lam_hist, sbar_hist, rows);
        });
        if (!found) return CM_ERR_UNSUPPORTED;
        if (check_launch() != CM_OK) return CM_ERR_LAUNCH;
    }
    cm_detail_sum_rows(s, rows, B, n_ep, grad_ep);
    return check_launch();
}
#endif

}  // namespace

#if CM_HAS_PART(1) && !CM_HNN_VARIANT
void cm_detail_reduce(hipStream_t s, const double* partials, int64_t nrows, double* stage, double* out, int first, int accumulate) {
    hipLaunchKernelGGL((k_reduce_stage1<kRed>), dim3(kRedBlocks), dim3(kRBlock), 0, s, partials, nrows, stage);
    hipLaunchKernelGGL((k_reduce_stage2<kRed>), dim3(1), dim3(kRBlock), 0, s, stage, out, first, accumulate);
}
void cm_detail_sum_rows(hipStream_t s, const double* part, int64_t nrows, int ncols, double* out) {
    hipLaunchKernelGGL(k_sum_rows, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, part, nrows, ncols, out);
}
#endif

extern "C" {
// objective + gradient at given converged states (the MODE 2 reverse kernel without a history vector); defined with cm_adjoint_step
__attribute__((visibility("hidden")))
int cmi_objective_from_state(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
                             const double* data, const double* wsq6, double* out, void* workspace, int64_t workspace_bytes, void* stream);


#if CM_HAS_PART(1) && !CM_HNN_VARIANT
int cm_abi_version(void) { return 6; }
#endif

#if CM_HAS_PART(1) && !CM_HNN_VARIANT
const char* cm_last_hip_error(void) { return hipGetErrorName((hipError_t)g_cm_last_hip_error); }
#endif

#if CM_HAS_PART(1) && !CM_HNN_VARIANT
int cm_num_xi(const cm_model_desc* m) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->def_type == CM_FULL_3D) return 7;
    if (m->def_type == CM_PLANE_STRESS) return 8;
    if (m->def_type == CM_UNIAXIAL_STRESS) return m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC ? 12 : 9;
    return CM_ERR_UNSUPPORTED;
}
#endif

#if CM_HAS_PART(1) && !CM_HNN_VARIANT
int cm_num_gradu(const cm_model_desc* m) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->def_type == CM_FULL_3D) return 9;
    if (m->def_type == CM_PLANE_STRESS || m->def_type == CM_PLANE_STRAIN) return 4;
    if (m->def_type == CM_UNIAXIAL_STRESS) return 1;
    return CM_ERR_UNSUPPORTED;
}
#endif

#if CM_HAS_PART(1) && !CM_HNN_VARIANT
// the reducing entry points' scratch, followed (8-byte aligned) by what cm_update_ws can use: a caller that sizes its workspace
// from this one function serves both -- the fused entry points hand the tail to the update they start with
int64_t cm_workspace_bytes(int64_t B) {
    if (B < 0) return CM_ERR_BAD_ARG;
    return reduce_workspace_bytes(B) + screen_workspace_bytes(B);
}
int64_t cm_update_workspace_bytes(int64_t B) {
    if (B < 0) return CM_ERR_BAD_ARG;
    return screen_workspace_bytes(B);
}
#endif

#if CM_HAS_PART(0)
int cm_update(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
              double* xi, double* sigma, uint32_t* status, void* stream) {
    return launch_update<false>(m, B, gradu, xi_prev, xi, sigma, nullptr, status, stream);
}
int cm_update_ws(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
                 double* xi, double* sigma, uint32_t* status, void* workspace, int64_t workspace_bytes, void* stream) {
    return launch_update<false>(m, B, gradu, xi_prev, xi, sigma, nullptr, status, stream, workspace, workspace_bytes);
}
#endif

#if CM_HAS_PART(1)
int cm_update_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                   const double* xi_prev, double* xi, double* sigma, uint32_t* status, void* stream) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    if (!supported(m, CM_SMALL_RATE_ELASTIC_PLASTIC) || rate_uniaxial_dense(m)) return CM_ERR_UNSUPPORTED;
    if (B == 0) return CM_OK;
    if (!gradu || !gradu_prev || !xi_prev || !xi) return CM_ERR_BAD_ARG;
    const dim3 grid((unsigned)nblocks_of(B)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const bool found = dispatch<true, 2>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (has_rate_uniaxial_dense<D, Y>())
        hipLaunchKernelGGL((k_update_rate<D, Y, R, LS, false>), grid, block, 0, s, md, B, gradu, gradu_prev, xi_prev, xi, sigma, nullptr, status);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}
#endif

#if CM_HAS_PART(5)
int cm_update_rate_tangent(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                           const double* xi_prev, double* xi, double* sigma, double* dsigma_dgradu, uint32_t* status,
                           void* stream) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    if (!supported(m, CM_SMALL_RATE_ELASTIC_PLASTIC) || rate_uniaxial_dense(m)) return CM_ERR_UNSUPPORTED;
    if (B == 0) return CM_OK;
    if (!gradu || !gradu_prev || !xi_prev || !xi || !dsigma_dgradu) return CM_ERR_BAD_ARG;
    const dim3 grid((unsigned)nblocks_of(B)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const bool found = dispatch<true, 2>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (has_rate_uniaxial_dense<D, Y>())
        hipLaunchKernelGGL((k_update_rate<D, Y, R, LS, true>), grid, block, 0, s, md, B, gradu, gradu_prev, xi_prev, xi, sigma,
                           dsigma_dgradu, status);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}
#endif

#if CM_HAS_PART(4)
int cm_update_tangent(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
                      double* xi, double* sigma, double* dsigma_dgradu, uint32_t* status, void* stream) {
    return launch_update<true>(m, B, gradu, xi_prev, xi, sigma, dsigma_dgradu, status, stream);
}
int cm_update_tangent_ws(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
                         double* xi, double* sigma, double* dsigma_dgradu, uint32_t* status,
                         void* workspace, int64_t workspace_bytes, void* stream) {
    return launch_update<true>(m, B, gradu, xi_prev, xi, sigma, dsigma_dgradu, status, stream, workspace, workspace_bytes);
}
#endif

#if CM_HAS_PART(2)
int cm_update_vjp(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
                  const double* sigma_bar, double* grad_p, double* xi_prev_bar, double* gradu_bar,
                  void* workspace, int64_t workspace_bytes, void* stream) {
    if (!grad_p) return CM_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < reduce_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    return launch_reverse<0>(m, B, gradu, xi_prev, xi, sigma_bar, nullptr, nullptr, nullptr, nullptr, xi_prev_bar,
                             gradu_bar, grad_p, 1, 0, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(3)
int cm_update_and_vjp(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
                      const double* sigma_bar, double* xi, double* sigma, double* grad_p,
                      void* workspace, int64_t workspace_bytes, void* stream) {
    if (!grad_p || !xi) return CM_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < reduce_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    const int64_t rb = reduce_workspace_bytes(B);
    if ((pool_route(m, B) || screen_route(m, B, (char*)workspace + rb, workspace_bytes - rb)) && gradu && xi_prev && sigma_bar) {
        // the update gets what the caller's workspace holds beyond the reduction's share (cm_workspace_bytes covers both)
        const int rc = cm_update_ws(m, B, gradu, xi_prev, xi, sigma, nullptr, (char*)workspace + rb, workspace_bytes - rb, stream);
        return rc != CM_OK ? rc : cm_update_vjp(m, B, gradu, xi_prev, xi, sigma_bar, grad_p, nullptr, nullptr, workspace, workspace_bytes, stream);
    }
    return launch_reverse<3>(m, B, gradu, xi_prev, nullptr, sigma_bar, nullptr, nullptr, xi, sigma, nullptr, nullptr,
                             grad_p, 1, 0, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(1) && !CM_HNN_VARIANT
int cm_sizeof_model_desc(void) { return (int)sizeof(cm_model_desc); }
#endif

#if CM_HAS_PART(1) && !CM_HNN_VARIANT
int64_t cm_direct_workspace_bytes(int64_t B) {
    if (B < 0) return CM_ERR_BAD_ARG;
    return ((B > 0 ? B : 1) + kRedBlocks + 1) * kRed * (int64_t)sizeof(double);      // one row per point + stage rows
}
int64_t cm_hessian_ep_workspace_bytes(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep) {
    if (!m || B < 0 || K < 1 || n_ep < 0) return CM_ERR_BAD_ARG;
    const int nx = cm_num_xi(m);
    if (nx < 0) return CM_ERR_UNSUPPORTED;
    const int64_t npt = CM_NUM_PARAMS + n_ep, nq = 2 * nx + npt, nps = (B > 0 ? B : 1) * (int64_t)K;
    return nps * (nq * nq + npt * npt) * (int64_t)sizeof(double);
}
int64_t cm_hessian_workspace_bytes(const cm_model_desc* m, int64_t B, int32_t K) {
    if (!m || B < 0 || K < 1) return CM_ERR_BAD_ARG;
    const int nx = cm_num_xi(m);
    if (nx < 0) return CM_ERR_UNSUPPORTED;
    const int64_t nq = 2 * nx + CM_NUM_PARAMS, nps = (B > 0 ? B : 1) * (int64_t)K;
    return nps * (nq * nq + CM_NUM_PARAMS * CM_NUM_PARAMS) * (int64_t)sizeof(double);
}
#endif

#if CM_HAS_PART(5)
int cm_direct_history(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* xi_hist,
                      const double* sigma_bar_hist, const double* xi_bar_hist, double* dxi_dp_hist, double* dsigma_dp_hist,
                      double* grad_p, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_direct_history<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, K, gradu_hist, xi_hist, sigma_bar_hist, xi_bar_hist,
                                                                    dxi_dp_hist, dsigma_dp_hist, grad_p, workspace, workspace_bytes, stream);
    return launch_direct_history<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, gradu_hist, xi_hist, sigma_bar_hist, xi_bar_hist,
                                                           dxi_dp_hist, dsigma_dp_hist, grad_p, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(6)
int cm_hessian_history(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* xi_hist,
                       const double* lam_hist, const double* dxi_dp_hist, const double* sigma_bar_hist, const double* hss6,
                       const double* hss_hist, const double* hxx_hist,
                       double* hess_pp, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_hessian_history<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, K, gradu_hist, xi_hist, lam_hist, dxi_dp_hist,
                                                                     sigma_bar_hist, hss6, hss_hist, hxx_hist, hess_pp, workspace, workspace_bytes, stream);
    return launch_hessian_history<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, sigma_bar_hist,
                                                            hss6, hss_hist, hxx_hist, hess_pp, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(9)
int cm_direct_history_ep(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep, const int32_t* ep_index,
                         const double* gradu_hist, const double* xi_hist, double* dxi_dpe_hist, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_direct_history_ep<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, K, n_ep, ep_index, gradu_hist, xi_hist, dxi_dpe_hist, stream);
    return launch_direct_history_ep<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, n_ep, ep_index, gradu_hist, xi_hist, dxi_dpe_hist, stream);
}
int cm_hessian_history_ep(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep, const int32_t* ep_index,
                          const double* gradu_hist, const double* xi_hist, const double* lam_hist, const double* dxi_dp_hist,
                          const double* dxi_dpe_hist, const double* sigma_bar_hist, const double* hss6, const double* hss_hist,
                          const double* hxx_hist, double* hess, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_hessian_history_ep<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, K, n_ep, ep_index, gradu_hist, xi_hist, lam_hist, dxi_dp_hist,
                                                                        dxi_dpe_hist, sigma_bar_hist, hss6, hss_hist, hxx_hist, hess, workspace,
                                                                        workspace_bytes, stream);
    return launch_hessian_history_ep<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, n_ep, ep_index, gradu_hist, xi_hist, lam_hist, dxi_dp_hist, dxi_dpe_hist,
                                                               sigma_bar_hist, hss6, hss_hist, hxx_hist, hess, workspace, workspace_bytes, stream);
}
#endif
#if CM_HAS_PART(10)
int cm_param_blocks(const cm_model_desc* m, int64_t B, int32_t n_ep, const int32_t* ep_index,
                    const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                    double* dC_dp, double* dsigma_dp, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_param_blocks<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, n_ep, ep_index, gradu, gradu_prev, xi_prev, xi, dC_dp, dsigma_dp, stream);
    return launch_param_blocks<CM_SMALL_ELASTIC_PLASTIC>(m, B, n_ep, ep_index, gradu, nullptr, xi_prev, xi, dC_dp, dsigma_dp, stream);
}
#endif
#if CM_HAS_PART(11)
int cm_update_complex(const cm_model_desc* m, int64_t B, const double* p_im, const double* ext_im, const double* gradu,
                      const double* gradu_prev, const double* xi_prev, double* xi, double* residual, double* sigma, uint32_t* status,
                      void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_update_complex<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, p_im, ext_im, gradu, gradu_prev, xi_prev, xi, residual, sigma, status, stream);
    return launch_update_complex<CM_SMALL_ELASTIC_PLASTIC>(m, B, p_im, ext_im, gradu, nullptr, xi_prev, xi, residual, sigma, status, stream);
}
#endif
#if CM_HAS_PART(10)
int cm_param_adjoint_history(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep, const int32_t* ep_index,
                             const double* gradu_hist, const double* xi_hist, const double* lam_hist,
                             const double* sigma_bar_hist, double* grad_ep, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_param_adjoint_history<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, K, n_ep, ep_index, gradu_hist, xi_hist, lam_hist,
                                                                           sigma_bar_hist, grad_ep, workspace, workspace_bytes, stream);
    return launch_param_adjoint_history<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, n_ep, ep_index, gradu_hist, xi_hist, lam_hist, sigma_bar_hist,
                                                                  grad_ep, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(5)
int cm_direct_step(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev, const double* xi_prev,
                   const double* xi, const double* dxi_prev_dp, double* dxi_dp, double* dsigma_dp, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return launch_direct_step<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, gradu, gradu_prev, xi_prev, xi, dxi_prev_dp, dxi_dp, dsigma_dp, stream);
    return launch_direct_step<CM_SMALL_ELASTIC_PLASTIC>(m, B, gradu, nullptr, xi_prev, xi, dxi_prev_dp, dxi_dp, dsigma_dp, stream);
}
#endif

#if CM_HAS_PART(5)
int cm_evaluate(const cm_model_desc* m, int64_t B, int which, const double* gradu, const double* xi_prev,
                const double* xi, double* C, double* jac, double* sigma, double* dsigma, void* stream) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    if (!supported(m)) return CM_ERR_UNSUPPORTED;
    if (which != CM_W_XI && which != CM_W_XI_PREV && which != CM_W_PARAMS && which != CM_W_U && which != CM_W_NONE)
        return CM_ERR_BAD_ARG;
    if (B == 0) return CM_OK;
    if (!gradu || !xi_prev || !xi) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const dim3 grid((unsigned)((B + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
        hipLaunchKernelGGL((k_evaluate<D, Y, kColdRot>), grid, block, 0, s, md, B, which, gradu, xi_prev, xi, C, jac, sigma, dsigma);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}
#endif

#if CM_HAS_PART(5)
int cm_evaluate_rate(const cm_model_desc* m, int64_t B, int which, const double* gradu, const double* gradu_prev,
                     const double* xi_prev, const double* xi, double* C, double* jac, double* sigma, double* dsigma,
                     void* stream) {
    if (!m || B < 0) return CM_ERR_BAD_ARG;
    if (!supported(m, CM_SMALL_RATE_ELASTIC_PLASTIC) || rate_uniaxial_dense(m)) return CM_ERR_UNSUPPORTED;
    if (which != CM_W_XI && which != CM_W_XI_PREV && which != CM_W_PARAMS && which != CM_W_U && which != CM_W_U_PREV &&
        which != CM_W_NONE) return CM_ERR_BAD_ARG;
    if (B == 0) return CM_OK;
    if (!gradu || !gradu_prev || !xi_prev || !xi) return CM_ERR_BAD_ARG;
    const cm_model_desc md = *m;
    (void)hipGetLastError();
    const dim3 grid((unsigned)((B + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    const bool found = dispatch<true>(m, [&]<int D, int Y, bool R, bool LS>() {
        if constexpr (has_rate_uniaxial_dense<D, Y>())
        hipLaunchKernelGGL((k_evaluate_rate<D, Y, kColdRot>), grid, block, 0, s, md, B, which, gradu, gradu_prev, xi_prev, xi,
                           C, jac, sigma, dsigma);
    });
    if (!found) return CM_ERR_UNSUPPORTED;
    return check_launch();
}
#endif

#if CM_HAS_PART(6)
int cm_hessians(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
                double* d2C, double* d2S, double* dC, double* dS, void* stream) {
    return launch_hessians<CM_SMALL_ELASTIC_PLASTIC>(m, B, gradu, nullptr, xi_prev, xi, d2C, d2S, dC, dS, nullptr, nullptr, stream);
}
int cm_hessians_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                     const double* xi_prev, const double* xi,
                     double* d2C, double* d2S, double* dC, double* dS, double* C0, double* sigma0, void* stream) {
    return launch_hessians<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, gradu, gradu_prev, xi_prev, xi, d2C, d2S, dC, dS, C0, sigma0, stream);
}
#endif

#if CM_HAS_PART(6)
int cm_objective_grad(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev,
                      const double* data, const double* wsq6, double* out, double* xi,
                      void* workspace, int64_t workspace_bytes, void* stream) {
    const int64_t rb = reduce_workspace_bytes(B);
    if (xi && gradu && xi_prev && data && out && workspace && workspace_bytes >= rb &&
        (pool_route(m, B) || screen_route(m, B, (char*)workspace + rb, workspace_bytes - rb))) {
        const int rc = cm_update_ws(m, B, gradu, xi_prev, xi, nullptr, nullptr, (char*)workspace + rb, workspace_bytes - rb, stream);   // needs somewhere to keep the states: only with xi
        return rc != CM_OK ? rc : cmi_objective_from_state(m, B, gradu, xi_prev, xi, data, wsq6, out, workspace, workspace_bytes, stream);
    }
    return launch_reverse<1>(m, B, gradu, xi_prev, nullptr, data, wsq6, nullptr, xi, nullptr, nullptr, nullptr,
                             out, 0, 0, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(2)
int cm_adjoint_step(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
                    const double* data, const double* wsq6, const double* hist_in, double* hist_out, double* out,
                    int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!hist_out) return CM_ERR_BAD_ARG;
    return launch_reverse<2>(m, B, gradu, xi_prev, xi, data, wsq6, hist_in, nullptr, nullptr, hist_out, nullptr,
                             out, 0, accumulate, workspace, workspace_bytes, stream);
}
int cmi_objective_from_state(const cm_model_desc* m, int64_t B, const double* gradu, const double* xi_prev, const double* xi,
                             const double* data, const double* wsq6, double* out, void* workspace, int64_t workspace_bytes, void* stream) {
    return launch_reverse<2>(m, B, gradu, xi_prev, xi, data, wsq6, nullptr, nullptr, nullptr, nullptr, nullptr,
                             out, 0, 0, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(1)
int cm_update_rate_vjp(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                       const double* xi_prev, const double* xi, const double* sigma_bar,
                       double* grad_p, double* xi_prev_bar, double* gradu_bar,
                       void* workspace, int64_t workspace_bytes, void* stream) {
    if (!grad_p) return CM_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < reduce_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    return launch_reverse_rate<0>(m, B, gradu, gradu_prev, xi_prev, xi, sigma_bar, nullptr, nullptr, nullptr, nullptr,
                                  xi_prev_bar, gradu_bar, grad_p, 1, 0, workspace, workspace_bytes, stream);
}
int cm_adjoint_step_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                         const double* xi_prev, const double* xi, const double* data, const double* wsq6,
                         const double* hist_in, double* hist_out, double* out, int accumulate,
                         void* workspace, int64_t workspace_bytes, void* stream) {
    if (!hist_out) return CM_ERR_BAD_ARG;
    return launch_reverse_rate<2>(m, B, gradu, gradu_prev, xi_prev, xi, data, wsq6, hist_in, nullptr, nullptr, hist_out,
                                  nullptr, out, 0, accumulate, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(2)
int cm_update_rate_and_vjp(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                           const double* xi_prev, const double* sigma_bar, double* xi, double* sigma, double* grad_p,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    if (!grad_p || (B > 0 && !xi)) return CM_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < reduce_workspace_bytes(B)) return CM_ERR_WORKSPACE;
    return launch_reverse_rate<3>(m, B, gradu, gradu_prev, xi_prev, nullptr, sigma_bar, nullptr, nullptr, xi, sigma, nullptr,
                                  nullptr, grad_p, 1, 0, workspace, workspace_bytes, stream);
}
int cm_objective_grad_rate(const cm_model_desc* m, int64_t B, const double* gradu, const double* gradu_prev,
                           const double* xi_prev, const double* data, const double* wsq6, double* out, double* xi,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    return launch_reverse_rate<1>(m, B, gradu, gradu_prev, xi_prev, nullptr, data, wsq6, nullptr, xi, nullptr, nullptr,
                                  nullptr, out, 0, 0, workspace, workspace_bytes, stream);
}
#endif

#if CM_HAS_PART(7) || CM_HAS_PART(8)
// the rate-form instantiations live in their own piece of the build (reached through cm_objective_grad_history)
int cm_internal_history_rate(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* data_hist,
                             const double* wsq6, const double* xi0, double* xi_hist, double* out,
                             void* workspace, int64_t workspace_bytes, void* stream,
                             const double* sbar_hist, const double* xibar_hist, double* lam_hist, int out_offset);
#endif

#if CM_HAS_PART(7) || CM_HAS_PART(8)
int cm_internal_primal_history_rate(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* xi0,
                                    double* xi_hist, double* sigma_hist, uint32_t* status_hist, void* stream);
#endif

#if CM_HAS_PART(8)
int cm_internal_primal_history_rate(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* xi0,
                                    double* xi_hist, double* sigma_hist, uint32_t* status_hist, void* stream) {
    return launch_primal_history<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist, stream);
}
#endif

#if CM_HAS_PART(7)
int cm_update_history(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* xi0,
                      double* xi_hist, double* sigma_hist, uint32_t* status_hist, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return cm_internal_primal_history_rate(m, B, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist, stream);
    return launch_primal_history<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, gradu_hist, xi0, xi_hist, sigma_hist, status_hist, stream);
}
#endif

#if CM_HAS_PART(8)
int cm_internal_history_rate(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* data_hist,
                             const double* wsq6, const double* xi0, double* xi_hist, double* out,
                             void* workspace, int64_t workspace_bytes, void* stream,
                             const double* sbar_hist, const double* xibar_hist, double* lam_hist, int out_offset) {
    return launch_history<CM_SMALL_RATE_ELASTIC_PLASTIC>(m, B, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, out, workspace, workspace_bytes, stream,
                                                         HistoryCotangents{sbar_hist, xibar_hist, lam_hist}, out_offset);
}
#endif

#if CM_HAS_PART(7)
int cm_objective_grad_history(const cm_model_desc* m, int64_t B, int32_t K,
                              const double* gradu_hist, const double* data_hist, const double* wsq6, const double* xi0,
                              double* xi_hist, double* out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return cm_internal_history_rate(m, B, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, out, workspace, workspace_bytes, stream,
                                        nullptr, nullptr, nullptr, 0);
    return launch_history<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, gradu_hist, data_hist, wsq6, xi0, xi_hist, out, workspace, workspace_bytes, stream);
}

int cm_adjoint_history(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* sigma_bar_hist,
                       const double* xi_bar_hist, const double* xi0, double* xi_hist, double* lam_hist, double* grad_p,
                       void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m || !grad_p || (B > 0 && !sigma_bar_hist)) return CM_ERR_BAD_ARG;
    if (m->model_kind == CM_SMALL_RATE_ELASTIC_PLASTIC)
        return cm_internal_history_rate(m, B, K, gradu_hist, nullptr, nullptr, xi0, xi_hist, grad_p, workspace, workspace_bytes, stream,
                                        sigma_bar_hist, xi_bar_hist, lam_hist, 1);
    return launch_history<CM_SMALL_ELASTIC_PLASTIC>(m, B, K, gradu_hist, nullptr, nullptr, xi0, xi_hist, grad_p, workspace, workspace_bytes,
                                                    stream, HistoryCotangents{sigma_bar_hist, xi_bar_hist, lam_hist}, 1);
}
#endif

}  // extern "C"

// ---- the public entry points: pick the build by the hardening law (cm_entries.inc, section 2) -------------------------------
#if CM_HAS_PART(1) && !CM_HNN_VARIANT
#define CM_ENTRIES_PUBLIC
#include "cm_entries.inc"
#undef CM_ENTRIES_PUBLIC
#endif
