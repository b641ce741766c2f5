/*
 * cmad_hip.h -- C-ABI of the MI355X (gfx950) batched constitutive-model evaluator.
 *
 * The reference (sandialabs/cmad) is pure Python + JAX and has NO FFI for this path; the drop-in
 * boundary is its Python operator API (cmad/models/model.py:25-88, cmad/parameters/parameters.py:176-203).
 * This C-ABI is the layer directly under the Python facade cmad_amd.models / cmad_amd.parameters and is
 * what a cmad maintainer would bind with ctypes (INTEGRATION.md shows the stub).  Each entry point names
 * the reference interface whose per-point work it replaces, batched over B Gauss points.
 *
 * Conventions
 *   - All arrays are DEVICE pointers to fp64 structure-of-arrays: component k of point b at [k*B + b].
 *   - Symmetric tensors are 6-vectors in CMAD order [xx, xy, xz, yy, yz, zz], un-weighted
 *     (cmad/models/var_types.py:43-84).  grad u is row-major: gradu[3*k + j] = d u_k / d x_j
 *     (cmad/models/global_fields.py:12-41).
 *   - n_xi = 7 (FULL_3D: plastic strain 6 + alpha), 8 (PLANE_STRESS: + F33), see cm_num_xi().
 *   - `stream` is a hipStream_t passed as void*; NULL = default stream.  Calls are asynchronous.
 *   - Every function returns 0 on success, a negative cm_status otherwise; nothing throws, nothing is
 *     allocated or synchronised inside (graph-capture safe), no ownership is transferred.
 *   - Newton non-convergence is NOT an error (reference: models/nonlinear_solver.py:85,153-155 return
 *     the last iterate silently); it is reported per point in `status`.
 */
#ifndef CMAD_HIP_H
#define CMAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum cm_status {
    CM_OK = 0,
    CM_ERR_BAD_ARG = -1,        /* null pointer / negative size */
    CM_ERR_UNSUPPORTED = -2,    /* def_type / yield / model kind not built (maps to NotImplementedError) */
    CM_ERR_LAUNCH = -3,         /* hipGetLastError() != hipSuccess after launch */
    CM_ERR_WORKSPACE = -4       /* workspace too small */
};

/* cmad/models/deformation_types.py:4-9 */
enum cm_def_type { CM_FULL_3D = 0, CM_PLANE_STRAIN = 1, CM_PLANE_STRESS = 2, CM_UNIAXIAL_STRESS = 3 };
/* registry names cmad/models/small_elastic_plastic.py:95, small_rate_elastic_plastic.py */
enum cm_model_kind { CM_SMALL_ELASTIC_PLASTIC = 0, CM_SMALL_RATE_ELASTIC_PLASTIC = 1 };
/* cmad/models/effective_stress.py:15-27 (+ hybrid_hill :149-163, scaled_effective_stress around it :130-146) */
enum cm_yield_kind { CM_YIELD_J2 = 0, CM_YIELD_HILL = 1, CM_YIELD_HOSFORD = 2, CM_YIELD_HYBRID_HILL_NN = 3,
                     CM_YIELD_SCALED_HYBRID_HILL_NN = 4, CM_YIELD_BARLAT = 5 /* Yld2004-18p, verification/functions.py:71-154 */ };

/* Kernel-level parameter order used by every sensitivity output ("KP order").
 * The Python facade maps it onto cmad.parameters' sorted-pytree flat order and applies the
 * elastic-constant chain rule (any two of E, nu, mu, kappa, lambda -> lambda, mu). */
enum cm_param_index {
    CM_P_LAMBDA = 0, CM_P_MU = 1, CM_P_Y = 2, CM_P_VOCE_S = 3, CM_P_VOCE_D = 4, CM_P_LIN_K = 5,
    CM_P_YC0 = 6,               /* hill F,G,H,L,M,N = 6..11 ; hosford a = 6 */
    CM_NUM_PARAMS = 12
};

/* solver_flags.  For J2 in FULL_3D without line search the Newton iteration started at x_prev never leaves the
 * radial line v = v_prev + dgam n_trial, so the kernels run the same iteration restricted to that line (identical
 * iterates and iteration counts, scalar linear algebra; a lane whose iterate would cross to the elastic branch
 * falls back to the general path).  For J2 in PLANE_STRESS the iterates stay in
 * v = v_prev + c_a dev(eps - v_prev) + c_b dev(z) and the kernels iterate on (c_a, c_b, alpha, F33): the same Newton
 * step in the coordinates of that plane.  CM_SOLVER_GENERAL_NEWTON switches every such specialisation off;
 * CM_SOLVER_J2_RADIAL_LINE is accepted for compatibility (the restriction is the default). */
#define CM_SOLVER_J2_RADIAL_LINE 1
#define CM_SOLVER_GENERAL_NEWTON 2
/* Warm starts (default on).  Where the backward-Euler equations reduce to a scalar return map or a small benign system -- Hill
 * in FULL_3D (one equation in kappa = 2 mu dgam / phi) and in PLANE_STRESS (the same with the stretch eliminated through the
 * sigma_33 row, which is linear in it), J2 in PLANE_STRESS (one equation in g = 3 mu dgam / phi, the stretch
 * eliminated the same way), J2 and Hill in UNIAXIAL_STRESS (the 1-d return map: the stress at the solution is
 * sigma_0 Z^0 and the flow direction a constant), Hosford with a >= 20 in FULL_3D (three equations in log-variables) -- the kernels solve
 * that first (a closed-form first step, two or more Newton steps in single precision as a seed, then double precision to
 * round-off) and START the reference's Newton iteration at its result: the reference's residual is evaluated there and the
 * reference's convergence test (relative to ||C(x_prev)||) decides, so a converged map costs one residual evaluation and
 * anything else is finished by the reference's Newton steps / line search from there.  Same root to the Newton tolerance
 * (SURVEY.md Appendix A); `status` counts the iterations taken FROM the warm start (usually 0).
 * CM_SOLVER_REFERENCE_ITERATES switches the warm starts off: the iteration starts at x_prev and reproduces the reference's
 * iterates and iteration counts (the invariant-subspace forms above stay on: they are the same iterates);
 * CM_SOLVER_GENERAL_NEWTON switches both off. */
#define CM_SOLVER_REFERENCE_ITERATES 8
/* cm_update runs the iteration-bound configurations on a work pool: a lane that has finished its Gauss point takes the next one
 * instead of waiting for the slowest point of its wavefront (same iteration per point, same results).  The predicate
 * (pool_route() in cmad_hip.hip, DeviceEvaluator.pool_route in cmad_amd/models/device.py): total-form model, B >= 256, and
 * yield_kind is CM_YIELD_HYBRID_HILL_NN / CM_YIELD_SCALED_HYBRID_HILL_NN, or CM_YIELD_HOSFORD with ls_max_evals > 0.  Every other
 * configuration (J2, Hill, Barlat, Hosford without the line search) measured slower on the pool and stays one point per lane.
 * CM_SOLVER_LOCKSTEP keeps one point per lane for the whole kernel for the pool configurations too (it changes nothing elsewhere).
 * For the pool configurations cm_update_tangent, cm_update_and_vjp, and cm_objective_grad when it is given a state buffer
 * (xi != NULL), run as work-pool update + a second kernel over the stored states (tangent / reverse sweep; two launches on the
 * stream, same results and reduction order) instead of one lockstep fused kernel; CM_SOLVER_LOCKSTEP keeps the single kernel. */
#define CM_SOLVER_LOCKSTEP 4

/* Which line search damps the Newton step when ls_max_evals > 0:
 * CM_LS_ARMIJO   make_newton_solve's search (cmad/util/line_search.py:95-189): quadratic interpolation clipped to
 *                [ls_lo, ls_hi] alpha, sufficient decrease ls_c1, lowest-merit step when no trial passes;
 * CM_LS_LEGACY   the backtracking loop of the imperative newton_solve(max_ls_evals > 0) (cmad/models/nonlinear_solver.py:55-81):
 *                accept when psi_j < (1 - 2 beta alpha_j) psi_0, else alpha_j <- max(eta alpha_j, quadratic minimiser), at most
 *                ls_max_evals residual evaluations, and the state stays at the last EVALUATED alpha when they run out
 *                (beta = ls_c1 = 1e-4 and eta = ls_lo = 0.5 in the reference; ls_hi is not used). */
typedef enum { CM_LS_ARMIJO = 0, CM_LS_LEGACY = 1 } cm_line_search_kind;

/* status word written per point by cm_update* (all optional outputs may be NULL) */
#define CM_STATUS_ITERS_MASK 0xFFFFu
#define CM_STATUS_CONVERGED  (1u << 16)
#define CM_STATUS_PLASTIC    (1u << 17)   /* plastic branch selected at the returned state */
#define CM_STATUS_SINGULAR   (1u << 18)   /* a vanishing pivot was met in a local solve */

/* Everything SmallElasticPlastic(parameters, def_type, ...) + make_newton_solve(...) fix at
 * construction time (small_elastic_plastic.py:109-211, nonlinear_solver.py:88-100,
 * util/line_search.py:40-46), flattened to plain data.  Passed by value. */
typedef struct cm_model_desc {
    int32_t model_kind;         /* cm_model_kind */
    int32_t def_type;           /* cm_def_type */
    int32_t yield_kind;         /* cm_yield_kind */
    int32_t has_voce;           /* hardening dict contains "voce"  (models/hardening.py:9-13) */
    int32_t has_linear;         /* hardening dict contains "linear" (models/hardening.py:16-19) */
    int32_t uniaxial_idx;       /* uniaxial_stress_idx, UNIAXIAL_STRESS only */
    int32_t rotation_is_identity; /* params["rotation matrix"] == eye(3) exactly -> skip Q products */
    int32_t solver_flags;       /* CM_SOLVER_* bits, 0 = defaults */
    double  yield_tol;          /* cond_residual tolerance, models/paths.py:26-27 (default 1e-14) */
    double  Q[9];               /* params["rotation matrix"], row-major */
    double  lambda, mu;         /* Lame pair, models/elastic_constants.py:53-104 */
    double  Y;                  /* initial yield */
    double  voce_S, voce_D;     /* Voce S(1 - exp(-D alpha)) */
    double  lin_K;              /* linear K alpha */
    double  yc[19];             /* hill F,G,H,L,M,N | hosford a in yc[0] | barlat sp_12,sp_13,sp_21,sp_23,sp_31,sp_32,
                                 * sp_44,sp_55,sp_66, dp_12 ... dp_66, a (effective_stress.py:55-78) */
    /* local Newton (models/nonlinear_solver.py:88-155) */
    int32_t max_iters;          /* default 10 (MP path) / 20 (FE binding) */
    int32_t ls_max_evals;       /* 0 = plain Newton (imperative newton_solve default); default 4 traced */
    double  abs_tol, rel_tol;   /* default 1e-14 / 1e-14 */
    double  ls_c1, ls_lo, ls_hi;/* sufficient decrease 1e-4, backtrack factors 0.5 / 0.9 */
    /* symmetric input-convex network of the hybrid Hill + NN yield surface
     * (neural_networks/input_convex_neural_network.py:36-69); device pointer or NULL.  Layout (doubles), widths [6, H, 1]:
     *   W0[6][H] (x-layer 0, row-major in x out), b0[H], Wx1[6] (x-layer 1), b1, Wz[H] (z-layer 0),
     *   in_scale[6], in_min[6], out_scale, out_min      (input / output AffineScaler, :13-33)
     *   f0 = forward(0)                                 (value of the scaled network at the origin)
     *   rec[H][10] = W0[0..5][o], b0[o], Wz[o], exp(b0[o]), exp(b0[o]) Wz[o]
     *                                                   (what the kernels' loop over the hidden units reads: one contiguous
     *                                                    record per unit -> wide scalar loads; exp(b0) lets both signs of a
     *                                                    unit come from one exponential.  Built by
     *                                                    cmad_amd.models.device.unit_records) */
    const double* nn_weights;
    int32_t nn_nlayers;
    int32_t nn_widths[7];
    /* CM_YIELD_SCALED_HYBRID_HILL_NN: phi(s) = phi_h(beta s) / beta with beta from the scalar Newton
     * phi_h(beta s) / beta_equivalent_stress = 1 started at Y / phi_J2(s)
     * (beta_make_newton_solve(effective_stress_fun, equivalent_stress, max_iters, abs_tol, rel_tol),
     *  models/effective_stress.py:97-108; its line search uses ls_c1 / ls_lo / ls_hi above with 4 evaluations) */
    double  beta_equivalent_stress;
    double  beta_abs_tol, beta_rel_tol;     /* default 1e-14 / 1e-14 */
    int32_t beta_max_iters;                 /* default 10 */
    int32_t ls_kind;                        /* cm_line_search_kind, read when ls_max_evals > 0 (was reserved0: 0 keeps the Armijo search) */
    /* neural-network hardening law: hardening_funs = {"neural network": SimpleNeuralNetwork(...).evaluate}
     * (cmad/neural_networks/simple_neural_network.py:13-46 through cmad/models/small_elastic_plastic.py:115,
     * cmad/models/hardening.py:27-34), layer widths [1, H, 1], sigmoid hidden units:
     *   H(alpha) = out_scale * (forward(in_scale * alpha) - forward(0)),  forward(x) = sum_u W2[u] sigmoid(W1[u] x + b1[u]) + b2.
     * hnn_width = H (0: no such law).  The weights sit in the nn_weights buffer at doubles offset hnn_offset:
     *   W1[H], b1[H], W2[H], b2, in_scale, out_scale, then sigmoid(b1[u]) for u < H (table for the kernels).
     * Added to the Voce / linear terms when those are present as well. */
    int32_t hnn_width;
    int32_t hnn_offset;
    /* more than one hidden layer: widths [1, H1, ..., Hn, 1], forward (simple_neural_network.py:19-23) loops over any depth.
     * hnn_nhidden = n (0 or 1: the one-hidden-layer layout above with H = hnn_width), 2 <= n <= 4 with at most 64 hidden units
     * in all: hnn_widths[0 .. n-1] = H1 .. Hn, hnn_width = H1, and the weights at hnn_offset in the GENERAL layout
     *   for l = 0 .. n:  W_l[n_in][n_out] (row-major, as params[l]["weights"]), b_l[n_out]        (n_in / n_out = 1, H1, ..., Hn, 1)
     *   then in_scale, out_scale, forward(0)                                                          (forward(0): host-computed). */
    int32_t hnn_nhidden;
    int32_t hnn_widths[4];
    int32_t reserved_tail;      /* keeps sizeof a multiple of 8 */
} cm_model_desc;

/* library / build info */
int  cm_abi_version(void);                       /* 6: cm_update_ws / cm_update_tangent_ws / cm_update_workspace_bytes, cm_workspace_bytes includes the screened update's share; 5: cm_model_desc.ls_kind (CM_LS_LEGACY), hnn_width / hnn_offset (network hardening); cm_hessian_history takes per-step stress curvature and
                                                  * a state curvature; nn_weights layout carries the per-unit records */
const char* cm_last_hip_error(void);             /* name of the last HIP error behind a CM_ERR_LAUNCH */
int  cm_sizeof_model_desc(void);                 /* sizeof(cm_model_desc) as compiled, for binding checks */
int  cm_num_xi(const cm_model_desc* m);          /* local dofs per point, <0 if unsupported */
int  cm_num_gradu(const cm_model_desc* m);       /* 9 (FULL_3D), 4 (PLANE_STRESS), 1 (UNIAXIAL_STRESS) */
int64_t cm_workspace_bytes(int64_t B);           /* scratch needed by the reducing entry points */

/*
 * cm_update: one backward-Euler stress update per Gauss point.
 * Replaces, per point: make_newton_solve(model._residual)(xi_prev, params, U, U_prev)
 * (cmad/models/nonlinear_solver.py:102-155) / newton_solve(model) (:14-85), followed by
 * model.cauchy(...) (cmad/models/small_elastic_plastic.py:307-321).
 *   in : gradu[n_gradu][B], xi_prev[n_xi][B]
 *   out: xi[n_xi][B], sigma[6][B] (global Cauchy stress, may be NULL), status[B] (may be NULL)
 */
int cm_update(const cm_model_desc* m, int64_t B,
              const double* gradu, const double* xi_prev,
              double* xi, double* sigma, uint32_t* status, void* stream);

/*
 * cm_update_ws / cm_update_tangent_ws: cm_update / cm_update_tangent with a caller-provided device workspace of at least
 * cm_update_workspace_bytes(B) bytes (8-byte aligned; contents need not be preserved between calls; one workspace per stream).
 * Same results.  With it the configurations whose residual evaluation is expensive (FULL_3D, total form: the network
 * surfaces, Barlat; B >= 4096) run SCREENED instead of on the work pool / in lockstep: a streaming kernel
 * finishes the points whose trial state is elastic (cond_residual's elastic branch at x_prev: C = 0, no iteration) and lists the
 * others; the lockstep Newton then runs over the list, so every lane of every wavefront holds a plastic point.  Without a
 * workspace (NULL / too small), or for any other configuration, these are cm_update / cm_update_tangent.  The fused entry points
 * that start with an update (cm_update_and_vjp, cm_objective_grad with a state buffer, cm_update_tangent_ws) do the same with the
 * part of their workspace beyond the reduction's share: cm_workspace_bytes(B) covers both.
 */
int64_t cm_update_workspace_bytes(int64_t B);
int cm_update_ws(const cm_model_desc* m, int64_t B,
                 const double* gradu, const double* xi_prev,
                 double* xi, double* sigma, uint32_t* status,
                 void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_update_rate: the same update for the rate-form model (m->model_kind = CM_SMALL_RATE_ELASTIC_PLASTIC), whose
 * unknown is the material Cauchy stress and whose residual needs the previous grad u as well.
 * Replaces make_newton_solve / newton_solve on SmallRateElasticPlastic._residual_fn
 * (cmad/models/small_rate_elastic_plastic.py:249-346) followed by its _cauchy_fn (:351-359).
 *   in : gradu[n_gradu][B], gradu_prev[n_gradu][B], xi_prev[n_xi][B] (xi = [sigma(6), alpha (, F33)])
 *   out: xi[n_xi][B], sigma[6][B] (global axes, may be NULL), status[B] (may be NULL)
 * The total-form entry points return CM_ERR_UNSUPPORTED for this model; its own are cm_update_rate_tangent,
 * cm_update_rate_vjp, cm_update_rate_and_vjp, cm_objective_grad_rate, cm_adjoint_step_rate, cm_evaluate_rate and
 * cm_hessians_rate below.
 */
int cm_update_rate(const cm_model_desc* m, int64_t B,
                   const double* gradu, const double* gradu_prev, const double* xi_prev,
                   double* xi, double* sigma, uint32_t* status, void* stream);

/*
 * cm_update_rate_tangent: cm_update_rate plus the IFT-consistent tangent d sigma / d gradu (FULL_3D, PLANE_STRESS);
 * d sigma / d gradu_prev is its negative (the residual sees grad u - grad u_prev only).  What
 * GlobalResidual._for_model_coupled (global_residuals/global_residual.py:373-394) obtains by jacfwd through the
 * custom_jvp rule when the block's model is SmallRateElasticPlastic (tests/fem/test_mixed_up_plastic.py:140-147).
 *   out: dsigma_dgradu[6*n_gradu][B], entry (r, c) at [(r*n_gradu + c)*B + b]
 */
int cm_update_rate_tangent(const cm_model_desc* m, int64_t B,
                           const double* gradu, const double* gradu_prev, const double* xi_prev,
                           double* xi, double* sigma, double* dsigma_dgradu, uint32_t* status, void* stream);

/*
 * cm_update_tangent: cm_update plus the IFT-consistent tangent d sigma / d gradu.  All three deformation types
 * (UNIAXIAL_STRESS: n_gradu = 1, the derivative of the six stress entries w.r.t. the axial strain).
 * Replaces jacfwd through the custom_jvp rule (cmad/models/nonlinear_solver.py:158-171) as used by
 * GlobalResidual._for_model_coupled (cmad/global_residuals/global_residual.py:373-394).
 *   out: dsigma_dgradu[6*n_gradu][B], entry (r, c) at [(r*n_gradu + c)*B + b]
 */
int cm_update_tangent(const cm_model_desc* m, int64_t B,
                      const double* gradu, const double* xi_prev,
                      double* xi, double* sigma, double* dsigma_dgradu, uint32_t* status, void* stream);
int cm_update_tangent_ws(const cm_model_desc* m, int64_t B,
                         const double* gradu, const double* xi_prev,
                         double* xi, double* sigma, double* dsigma_dgradu, uint32_t* status,
                         void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_update_vjp: reverse-mode sensitivities of one converged update for a given stress cotangent.
 * This entry and cm_update_and_vjp / cm_objective_grad / cm_adjoint_step serve FULL_3D, PLANE_STRESS and
 * UNIAXIAL_STRESS (9 local dofs, the setting of cmad/calibrations/al7079/multi_experiment_hill_calibration.py).
 * Replaces the transpose of the custom_jvp rule (nonlinear_solver.py:158-171):
 *   lam = (dC/dxi)^-T (dsigma/dxi)^T sbar ;  pbar = (dsigma/dp)^T sbar - (dC/dp)^T lam  (same for xi_prev, gradu)
 *   in : gradu, xi_prev, xi (converged, from cm_update), sigma_bar[6][B] (cotangent of the 6 stored entries)
 *   out: grad_p[CM_NUM_PARAMS] (device, sum over points, KP order, NATIVE lambda/mu parameters),
 *        xi_prev_bar[n_xi][B] (may be NULL), gradu_bar[n_gradu][B] (may be NULL)
 *   workspace: cm_workspace_bytes(B) device bytes
 */
int cm_update_vjp(const cm_model_desc* m, int64_t B,
                  const double* gradu, const double* xi_prev, const double* xi, const double* sigma_bar,
                  double* grad_p, double* xi_prev_bar, double* gradu_bar,
                  void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_update_and_vjp: cm_update and cm_update_vjp fused in one pass over the batch (the state never
 * leaves registers between the Newton solve and the reverse sweep): reads gradu, xi_prev, sigma_bar;
 * writes xi, sigma (may be NULL) and the reduced grad_p[CM_NUM_PARAMS].  Same reference lines as the
 * two calls it fuses.  For a sigma_bar that does not depend on this step's sigma.
 */
int cm_update_and_vjp(const cm_model_desc* m, int64_t B,
                      const double* gradu, const double* xi_prev, const double* sigma_bar,
                      double* xi, double* sigma, double* grad_p,
                      void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_objective_grad: fused single-step calibration objective and gradient.
 * Replaces, per point and summed over the batch: MPAdjointObjective._evaluate for a one-step history
 * (cmad/objectives/mp_objective.py:95-147) with QoI Calibration._qoi (cmad/qois/calibration.py:56-66):
 *   J = sum_b 1/2 sum_ij (w_ij (sigma_ij - data_ij))^2 ,  grad = dJ/dp (KP order, native parameters)
 *   in : gradu, xi_prev, data[6][B] (measured stress, 6 unique entries), wsq6[6] (HOST pointer): squared
 *        weights of the 6 unique entries, w_ii^2 on the diagonal and w_ij^2 + w_ji^2 off it (a 3x3
 *        weight mask and non-symmetric data fold into this form exactly; the facade does it)
 *   out: out[1 + CM_NUM_PARAMS] device doubles = {J, grad...}; xi[n_xi][B] optional (NULL = not stored)
 */
int cm_objective_grad(const cm_model_desc* m, int64_t B,
                      const double* gradu, const double* xi_prev, const double* data, const double* wsq6,
                      double* out, double* xi, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_adjoint_step: one reverse-time step of the adjoint pass of a K-step history
 * (cmad/objectives/mp_objective.py:112-142), for all points:
 *   phi = (dC/dxi)^-T (-dJ/dxi^T + h_in) ; h_out = -(dC/dxi_prev)^T phi ; grad += phi^T dC/dp + dJ/dp
 *   in : gradu (step k), xi_prev (= xi_{k-1}), xi (= xi_k), data[6][B], wsq6[6] (host), hist_in[n_xi][B] (NULL = 0)
 *   out: hist_out[n_xi][B] (may alias hist_in), out[1 + CM_NUM_PARAMS] += {J_k, grad_k} when accumulate != 0
 */
int cm_adjoint_step(const cm_model_desc* m, int64_t B,
                    const double* gradu, const double* xi_prev, const double* xi,
                    const double* data, const double* wsq6, const double* hist_in,
                    double* hist_out, double* out, int accumulate,
                    void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_evaluate: residual, one Jacobian block, stress and one stress-derivative block at GIVEN states
 * (no Newton solve).  Replaces the stateful evaluate surface of the reference, per point:
 * Model.evaluate() / C() / Jac() (cmad/models/model.py:168-190) and Model.evaluate_cauchy() / Sigma() /
 * dSigma() (:273-293), whose derivative blocks the reference obtains with jacfwd/jacrev.
 *   which: cm_deriv (cmad/models/deriv_types.py:4-10): 0 d/dxi, 1 d/dxi_prev, 2 d/dparams (KP order, native
 *          lambda/mu), 3 d/dgradu, 5 none
 *   out  : C[n_xi][B]; jac[n_xi*ncols][B], entry (r,c) at [(r*ncols+c)*B+b]; sigma[6][B];
 *          dsigma[6*ncols][B] (rows = the 6 stored entries of the global stress); any may be NULL.
 *          ncols = n_xi (0,1), CM_NUM_PARAMS (2), n_gradu (3).
 */
enum cm_deriv { CM_DXI = 0, CM_DXI_PREV = 1, CM_DPARAMS = 2, CM_DU = 3, CM_DU_PREV = 4, CM_DNONE = 5 };
int cm_evaluate(const cm_model_desc* m, int64_t B, int which,
                const double* gradu, const double* xi_prev, const double* xi,
                double* C, double* jac, double* sigma, double* dsigma, void* stream);

/*
 * cm_evaluate_rate: cm_evaluate for the rate-form model (small_rate_elastic_plastic.py:249-359), whose residual
 * also depends on the previous grad u; `which` may additionally be CM_DU_PREV (= minus the CM_DU block).
 */
int cm_evaluate_rate(const cm_model_desc* m, int64_t B, int which,
                     const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                     double* C, double* jac, double* sigma, double* dsigma, void* stream);

/*
 * cm_hessians: second derivatives of the residual and of the global stress at given states w.r.t.
 * q = [xi (n_xi), xi_prev (n_xi), p (CM_NUM_PARAMS, KP order, native lambda/mu)], n_q = 2 n_xi + CM_NUM_PARAMS.
 * Replaces Model.evaluate_hessians() (cmad/models/model.py:133-147, 245-270: jax.hessian / jacrev(jacfwd) of
 * the residual) and the stress part of QoI.evaluate_hessians() (cmad/qois/qoi.py:160-188), total-form model,
 * every yield surface (hyper-dual evaluation of the arithmetic-T model, one pair of variables per thread).  Outputs are point-major (array of structures; B is small for this call), row-major:
 *   d2C[B][n_xi][n_q][n_q], d2S[B][6][n_q][n_q], dC[B][n_xi][n_q], dS[B][6][n_q]      (any may be NULL)
 * dC / dS are the first derivatives produced by the same pass (for cross-checks against cm_evaluate).
 */
int cm_hessians(const cm_model_desc* m, int64_t B,
                const double* gradu, const double* xi_prev, const double* xi,
                double* d2C, double* d2S, double* dC, double* dS, void* stream);

/*
 * Reverse-mode entry points of the rate-form model (FULL_3D, PLANE_STRESS, UNIAXIAL_STRESS; J2 / Hill / Hosford): cm_update_vjp,
 * cm_update_and_vjp, cm_objective_grad and cm_adjoint_step with the additional input gradu_prev[n_gradu][B].
 * Same outputs and conventions; gradu_bar is the cotangent of grad u, the one of grad u_prev is its negative
 * (the residual sees grad u - grad u_prev only).  They replace the same reference lines as their total-form
 * counterparts, applied to SmallRateElasticPlastic (small_rate_elastic_plastic.py:249-359), e.g. the adjoint pass
 * of tests/objectives/test_calibrations.py:86-109.
 */
int cm_update_rate_vjp(const cm_model_desc* m, int64_t B,
                       const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                       const double* sigma_bar, double* grad_p, double* xi_prev_bar, double* gradu_bar,
                       void* workspace, int64_t workspace_bytes, void* stream);
int cm_update_rate_and_vjp(const cm_model_desc* m, int64_t B,
                           const double* gradu, const double* gradu_prev, const double* xi_prev, const double* sigma_bar,
                           double* xi, double* sigma, double* grad_p,
                           void* workspace, int64_t workspace_bytes, void* stream);
int cm_objective_grad_rate(const cm_model_desc* m, int64_t B,
                           const double* gradu, const double* gradu_prev, const double* xi_prev,
                           const double* data, const double* wsq6, double* out, double* xi,
                           void* workspace, int64_t workspace_bytes, void* stream);
int cm_adjoint_step_rate(const cm_model_desc* m, int64_t B,
                         const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                         const double* data, const double* wsq6, const double* hist_in,
                         double* hist_out, double* out, int accumulate,
                         void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_hessians_rate: cm_hessians for the rate-form model (small_rate_elastic_plastic.py:249-359), whose residual
 * also takes the previous grad u -- what the reference's Hessian checks run on SmallRateElasticPlastic
 * (tests/objectives/test_J2_fd_checks.py:303-392).  All three deformation types (J2 / Hill / Hosford); under
 * UNIAXIAL_STRESS the rate form has 12 local dofs (stress 6, alpha, two off-axis stretches, three off-axis strain
 * increments, :171-196).  The other rate-form entry points serve that variant too: their derivative blocks come from
 * forward-mode evaluation of the same residual inside the kernels (cmad_amd/csrc/cm_rate_uniaxial.hpp).
 *   out (all optional): d2C, d2S, dC, dS as cm_hessians; C0[B][n_xi] residual values, sigma0[B][6] global stress.
 */
int cm_hessians_rate(const cm_model_desc* m, int64_t B,
                     const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                     double* d2C, double* d2S, double* dC, double* dS, double* C0, double* sigma0, void* stream);

/*
 * cm_objective_grad_history: calibration objective and gradient over a whole K-step load history per point, one launch.
 * Replaces, per point and summed over the batch, MPAdjointObjective._evaluate (cmad/objectives/mp_objective.py:95-147):
 * the forward pass with storage (:62-89) and the adjoint recursion (:112-142) with QoI Calibration._qoi
 * (cmad/qois/calibration.py:56-66).  Same numbers as K calls of cm_update (cm_update_rate) followed by K calls of
 * cm_adjoint_step (cm_adjoint_step_rate), but the state stays in registers from step to step: per point and step the
 * forward pass reads grad u and writes xi, the backward pass reads grad u, the previous xi and the data.
 * Both model kinds (m->model_kind; the rate form takes grad u of step k-1 as its previous grad u).
 *   in : gradu_hist[(K+1)][n_gradu][B] (step 0 = initial configuration), data_hist[(K+1)][6][B] (step 0 unused),
 *        wsq6[6] (HOST, as cm_objective_grad), xi0[n_xi][B]
 *   out: xi_hist[(K+1)][n_xi][B] (slot 0 = xi0, slot K = final state; also the kernel's own storage),
 *        out[1 + CM_NUM_PARAMS] = {J, grad...}
 */
int cm_objective_grad_history(const cm_model_desc* m, int64_t B, int32_t K,
                              const double* gradu_hist, const double* data_hist, const double* wsq6, const double* xi0,
                              double* xi_hist, double* out, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_update_history: K stress updates per point in one launch (the state is carried in registers from step to step).
 * Replaces, per point, the forward pass with storage of the material-point drivers
 * (cmad/objectives/mp_objective.py:62-89, cmad/cli/primal.py:129-176 without the QoI): the same numbers as K calls of
 * cm_update (cm_update_rate for m->model_kind = CM_SMALL_RATE_ELASTIC_PLASTIC, previous grad u = step k-1).
 *   in : gradu_hist[(K+1)][n_gradu][B] (step 0 = initial configuration), xi0[n_xi][B]
 *   out (each may be NULL, not both xi_hist and sigma_hist): xi_hist[(K+1)][n_xi][B] (slot 0 = xi0),
 *        sigma_hist[(K+1)][6][B] (global Cauchy stress; slot 0 = stress of xi0 under gradu_hist[0]),
 *        status_hist[(K+1)][B] (iterations and CM_STATUS_CONVERGED / CM_STATUS_SINGULAR per step; the informational
 *        CM_STATUS_PLASTIC bit is not evaluated here; slot 0 = CM_STATUS_CONVERGED)
 */
int cm_update_history(const cm_model_desc* m, int64_t B, int32_t K, const double* gradu_hist, const double* xi0,
                      double* xi_hist, double* sigma_hist, uint32_t* status_hist, void* stream);

/*
 * cm_direct_step: forward (direct) parameter sensitivities of one converged step, per point.
 * Replaces the per-step body of MPDirectObjective._evaluate (cmad/objectives/mp_objective.py:158-215):
 *   dxi/dp = -(dC/dxi)^-1 (dC/dp + dC/dxi_prev dxi_prev/dp) ,  dsigma/dp = dsigma/dp|_xi + dsigma/dxi dxi/dp
 * for the CM_NUM_PARAMS native parameters (KP order; columns of coefficients the model does not use are zero).
 * Both model kinds (gradu_prev: rate form only, NULL otherwise); FULL_3D, PLANE_STRESS, UNIAXIAL_STRESS (total form).
 *   in : gradu, [gradu_prev,] xi_prev, xi (converged), dxi_prev_dp[n_xi*CM_NUM_PARAMS][B] (NULL = zeros: first step)
 *   out: dxi_dp[n_xi*CM_NUM_PARAMS][B], entry (k, j) at [(k*CM_NUM_PARAMS + j)*B + b];
 *        dsigma_dp[6*CM_NUM_PARAMS][B] (global Cauchy stress rows xx,xy,xz,yy,yz,zz; may be NULL)
 */
int cm_direct_step(const cm_model_desc* m, int64_t B,
                   const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                   const double* dxi_prev_dp, double* dxi_dp, double* dsigma_dp, void* stream);

/*
 * cm_adjoint_history: the adjoint recursion of a whole K-step load history per point for an ARBITRARY quantity of
 * interest, one launch.  Replaces MPAdjointObjective._evaluate's reverse loop (cmad/objectives/mp_objective.py:112-142)
 * for any QoI (cmad/qois/qoi.py:80-110): the caller evaluates its QoI on the stresses / states of cm_update_history and
 * hands in, per step, sigma_bar = dJ_k/dsigma (the 6 stored global entries) and, optionally, xi_bar = the explicit
 * dJ_k/dxi (e.g. the lateral stretches of cmad/qois/uniaxial_calibration.py:70-85).  The forward pass is recomputed
 * with the state in registers (as in cm_objective_grad_history), then
 *     lam_k = A_k^-T ((dsigma/dxi)^T sigma_bar_k + xi_bar_k + incoming) ,   grad += sigma_bar_k . dsigma/dp - lam_k . dC_k/dp
 *   in : gradu_hist[(K+1)][n_gradu][B], sigma_bar_hist[(K+1)][6][B], xi_bar_hist[(K+1)][n_xi][B] (NULL = none;
 *        slot 0 of both unused), xi0[n_xi][B]
 *   out: xi_hist[(K+1)][n_xi][B], lam_hist[(K+1)][n_xi][B] (NULL = not stored; slot 0 unused; the reference's
 *        phi_k = -lam_k, what the second-order pass cm_hessian_history needs), grad_p[CM_NUM_PARAMS] summed over the batch
 */
int cm_adjoint_history(const cm_model_desc* m, int64_t B, int32_t K,
                       const double* gradu_hist, const double* sigma_bar_hist, const double* xi_bar_hist, const double* xi0,
                       double* xi_hist, double* lam_hist, double* grad_p,
                       void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_direct_history: the forward-sensitivity recursion of MPDirectObjective._evaluate
 * (cmad/objectives/mp_objective.py:158-215) over a stored K-step history per point, one launch (= K cm_direct_step
 * calls with the sensitivity block carried from step to step), with the gradient contraction on the device:
 *     grad_p[j] = sum_b sum_k sigma_bar_k . dsigma_k/dp_j + xi_bar_k . dxi_k/dp_j
 *   in : gradu_hist[(K+1)][n_gradu][B], xi_hist[(K+1)][n_xi][B] (converged states, e.g. from cm_update_history),
 *        sigma_bar_hist[(K+1)][6][B] / xi_bar_hist[(K+1)][n_xi][B] (QoI cotangents as in cm_adjoint_history; needed
 *        only when grad_p is requested, xi_bar_hist may be NULL)
 *   out (each may be NULL, not all): dxi_dp_hist[(K+1)][n_xi*CM_NUM_PARAMS][B], dsigma_dp_hist[(K+1)][6*CM_NUM_PARAMS][B]
 *        (slot 0 = 0; entry layout as cm_direct_step), grad_p[CM_NUM_PARAMS]
 *   workspace: cm_direct_workspace_bytes(B) when grad_p is requested
 */
int64_t cm_direct_workspace_bytes(int64_t B);
int cm_direct_history(const cm_model_desc* m, int64_t B, int32_t K,
                      const double* gradu_hist, const double* xi_hist,
                      const double* sigma_bar_hist, const double* xi_bar_hist,
                      double* dxi_dp_hist, double* dsigma_dp_hist, double* grad_p,
                      void* workspace, int64_t workspace_bytes, void* stream);

/*
 * cm_hessian_history: Hessian of the objective w.r.t. the CM_NUM_PARAMS native parameters by the direct-adjoint method,
 * replacing the second loop of MPDirectAdjointObjective._evaluate (cmad/objectives/mp_objective.py:283-341) together
 * with Model.evaluate_hessians (cmad/models/model.py:245-270) and the QoI Hessians (cmad/qois/qoi.py:160-188).
 * With q = [xi_k, xi_{k-1}, p] and D_k = dq/dp = [dxi_k/dp ; dxi_{k-1}/dp ; I]:
 *     hess_pp = sum_b sum_k D_k^T W_k D_k ,
 *     W_k[a][b] = sigma_bar_k . d2sigma/dq_a dq_b + sum_r hss_r dsigma_r/dq_a dsigma_r/dq_b - lam_k . d2C_k/dq_a dq_b
 * (the reference's 13 einsum terms are the blocks of this one quadratic form).  The QoI enters through sigma_bar (its
 * first derivative), through the diagonal d2J_k/dsigma_r^2 in the 6 stored entries -- hss6[6] (HOST; constant in time,
 * Calibration: the folded squared weights, as wsq6) or hss_hist[(K+1)][6] (DEVICE, per step; takes precedence; slot 0
 * unused) -- and, for a QoI with an explicit dJ/dxi (UniaxialCalibration's lateral stretches,
 * cmad/qois/uniaxial_calibration.py:70-85, differentiated there by hessian(qoi_fun), cmad/qois/qoi.py:47-57), through the
 * diagonal d2J_k/dxi_i^2 of the current step's state, hxx_hist[(K+1)][n_xi] (DEVICE, may be NULL), which is added to W_k[i][i].
 * lam_hist must then come from cm_adjoint_history with the QoI's xi_bar_hist.
 *   in : gradu_hist, xi_hist (converged), lam_hist (cm_adjoint_history), dxi_dp_hist (cm_direct_history), sigma_bar_hist
 *   out: hess_pp[CM_NUM_PARAMS * CM_NUM_PARAMS] row-major, KP order
 *   workspace: cm_hessian_workspace_bytes(m, B, K)
 * Every yield surface; both model kinds.
 */
int64_t cm_hessian_workspace_bytes(const cm_model_desc* m, int64_t B, int32_t K);
int cm_hessian_history(const cm_model_desc* m, int64_t B, int32_t K,
                       const double* gradu_hist, const double* xi_hist, const double* lam_hist, const double* dxi_dp_hist,
                       const double* sigma_bar_hist, const double* hss6, const double* hss_hist, const double* hxx_hist,
                       double* hess_pp, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The second-order pass including parameter-tree leaves OUTSIDE the 12 native parameters (rotation matrix, Hosford exponent,
 * Barlat / network-surface coefficients, network weights: "EP" indices as cm_param_blocks; reference: Model.evaluate_hessians
 * takes Hessians over the whole params pytree, cmad/models/model.py:133-147, which MPDirectAdjointObjective contracts,
 * cmad/objectives/mp_objective.py:218-345).
 * cm_direct_history_ep: forward sensitivities dxi_k/dp_e of the requested extended parameters over a stored history
 *   (the recursion of cm_direct_history with the parameter column from forward-mode evaluation of the model):
 *   out dxi_dpe_hist[(K+1)][n_xi*n_ep][B], entry (i, j) of step k at row k*n_xi*n_ep + i*n_ep + j (slot 0 = 0).
 * cm_hessian_history_ep: cm_hessian_history over q = [xi_k, xi_{k-1}, p (12), pe (n_ep)]:
 *   hess[(12 + n_ep)^2] row-major, native parameters first (KP order), then the extended ones in ep_index order;
 *   n_ep = 0 reproduces cm_hessian_history.  ep_index: DEVICE int32[n_ep], n_ep <= 64.
 *   workspace: cm_hessian_ep_workspace_bytes(m, B, K, n_ep).
 * Total form: every deformation type; rate form: FULL_3D / PLANE_STRESS.
 */
int64_t cm_hessian_ep_workspace_bytes(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep);
int cm_direct_history_ep(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep, const int32_t* ep_index,
                         const double* gradu_hist, const double* xi_hist, double* dxi_dpe_hist, void* stream);
int cm_hessian_history_ep(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep, const int32_t* ep_index,
                          const double* gradu_hist, const double* xi_hist, const double* lam_hist, const double* dxi_dp_hist,
                          const double* dxi_dpe_hist, const double* sigma_bar_hist, const double* hss6, const double* hss_hist,
                          const double* hxx_hist, double* hess, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Extended parameter sensitivities.  The hand-derived kernels differentiate w.r.t. the 12 native parameters of
 * cm_param_index; the reference differentiates w.r.t. EVERY leaf of the params pytree (jacrev in cmad/models/model.py:125-153,
 * flattened by cmad/parameters/parameters.py:368-377).  The remaining leaves -- rotation matrix, Hosford exponent, Barlat
 * coefficients, Hill coefficients of the network surfaces, network weights -- are served by forward-mode evaluation of the whole model
 * (residual, kinematics with the rotation matrix, global stress) in dual-number arithmetic, one direction per thread.
 * Extended parameter ("EP") index: 0..11 = cm_param_index; 12..24 = yc[6..18]; 25..33 = Q[0..8] (row-major);
 * 34 + i = packed network weight i (W0[6][H], b0[H], Wx1[6], b1, Wz[H]).  Every yield surface (Barlat through a Jacobi eigen-decomposition in dual arithmetic); both model kinds
 * (the rate form under UNIAXIAL_STRESS included).
 *
 * cm_param_blocks: dC/dp_e and d sigma/dp_e at given states (the DPARAMS blocks of Model.evaluate / evaluate_cauchy,
 * cmad/models/model.py:168-190, 273-293, for those leaves).
 *   in : ep_index[n_ep] (DEVICE, int32), gradu, gradu_prev (rate form, else NULL), xi_prev, xi
 *   out: dC_dp[n_ep][n_xi][B], dsigma_dp[n_ep][6][B] (either may be NULL)
 * cm_param_adjoint_history: grad_ep[j] = sum_b sum_k sigma_bar_k . d sigma_k/dp_e - lam_k . dC_k/dp_e over a stored history
 * with the adjoint vectors of cm_adjoint_history -- the extended leaves' share of the objective gradient.
 *   workspace: B * n_ep doubles
 */
int cm_param_blocks(const cm_model_desc* m, int64_t B, int32_t n_ep, const int32_t* ep_index,
                    const double* gradu, const double* gradu_prev, const double* xi_prev, const double* xi,
                    double* dC_dp, double* dsigma_dp, void* stream);
int cm_param_adjoint_history(const cm_model_desc* m, int64_t B, int32_t K, int32_t n_ep, const int32_t* ep_index,
                             const double* gradu_hist, const double* xi_hist, const double* lam_hist,
                             const double* sigma_bar_hist, double* grad_ep,
                             void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Complex-step model instances.  The reference can build a model with complex dtype, Model(parameters, def_type,
 * is_complex=True) (cmad/models/small_elastic_plastic.py:118-127, small_rate_elastic_plastic.py:125-134), so that
 * Im J(p + i h d) / h gives a directional derivative of an objective that uses no derivative code at all
 * (tests/objectives/test_J2_fd_checks.py:163-235, 355-386).  cm_update_complex is the local solve of such an instance: the
 * imperative Newton (cmad/models/nonlinear_solver.py:14-85, plain steps) on the complex residual with the holomorphic Jacobian;
 * |.| and the branch select are decided on real parts (complex-step convention).
 *   in : p_im[CM_NUM_PARAMS] (HOST) imaginary parts of the native parameters, KP order -- the real parts are the description's;
 *        ext_im (DEVICE, may be NULL = all zero): imaginary parts of every other parameter the model reads, in the order of the
 *        extended parameter index from 12 on (cm_param_blocks): yc[6..18] (13 Barlat coefficients), Q row-major (9), then one
 *        entry per double of cm_model_desc.nn_weights (22 + the length of that buffer doubles in all);
 *        gradu[n_gradu][B] (real), gradu_prev (rate form, else NULL), xi_prev[2][n_xi][B] (real rows, then imaginary rows)
 *   i/o: xi[2][n_xi][B]: in = the starting iterate (the reference starts at the model's current state), out = the returned state
 *   out: residual[2][n_xi][B], sigma[2][6][B] at the returned state (either may be NULL), status[B] (may be NULL)
 * max_iters = 0 evaluates residual and stress at xi.  Every yield surface and hardening law; both model kinds; every deformation
 * type.
 */
int cm_update_complex(const cm_model_desc* m, int64_t B, const double* p_im, const double* ext_im, const double* gradu,
                      const double* gradu_prev, const double* xi_prev, double* xi, double* residual, double* sigma, uint32_t* status,
                      void* stream);

#ifdef __cplusplus
}
#endif

#endif
