"""ctypes binding of include/cmad_hip.h.  Fails loudly when the HIP library is missing: there is no
CPU fallback anywhere in this package."""
import ctypes as C
import os

from .build import LIB

CM_NUM_PARAMS = 12
P_LAMBDA, P_MU, P_Y, P_VOCE_S, P_VOCE_D, P_LIN_K, P_YC0 = 0, 1, 2, 3, 4, 5, 6
SOLVER_J2_RADIAL_LINE = 1
SOLVER_GENERAL_NEWTON = 2
SOLVER_LOCKSTEP = 4
SOLVER_REFERENCE_ITERATES = 8
STATUS_ITERS_MASK, STATUS_CONVERGED, STATUS_PLASTIC, STATUS_SINGULAR = 0xFFFF, 1 << 16, 1 << 17, 1 << 18
LS_ARMIJO, LS_LEGACY = 0, 1
CM_OK, CM_ERR_BAD_ARG, CM_ERR_UNSUPPORTED, CM_ERR_LAUNCH, CM_ERR_WORKSPACE = 0, -1, -2, -3, -4

EXPORTS = ["cm_hessians", "cm_hessians_rate", "cm_update_rate_tangent", "cm_update_rate_vjp", "cm_update_rate_and_vjp",
           "cm_objective_grad_rate", "cm_adjoint_step_rate", "cm_evaluate_rate", "cm_update_rate", "cm_evaluate", "cm_abi_version", "cm_last_hip_error", "cm_sizeof_model_desc", "cm_update_and_vjp", "cm_num_xi", "cm_num_gradu", "cm_workspace_bytes", "cm_update", "cm_update_ws", "cm_update_tangent_ws", "cm_update_workspace_bytes",
           "cm_update_tangent", "cm_update_vjp", "cm_objective_grad", "cm_adjoint_step", "cm_objective_grad_history", "cm_update_history", "cm_direct_step",
           "cm_param_blocks", "cm_param_adjoint_history", "cm_update_complex", "cm_adjoint_history", "cm_direct_history", "cm_direct_workspace_bytes", "cm_hessian_history", "cm_hessian_workspace_bytes",
           "cm_direct_history_ep", "cm_hessian_history_ep", "cm_hessian_ep_workspace_bytes"]


class ModelDesc(C.Structure):
    """Mirror of `cm_model_desc` (include/cmad_hip.h)."""
    _fields_ = [
        ("model_kind", C.c_int32), ("def_type", C.c_int32), ("yield_kind", C.c_int32),
        ("has_voce", C.c_int32), ("has_linear", C.c_int32), ("uniaxial_idx", C.c_int32),
        ("rotation_is_identity", C.c_int32), ("solver_flags", C.c_int32),
        ("yield_tol", C.c_double), ("Q", C.c_double * 9), ("lmbda", C.c_double), ("mu", C.c_double),
        ("Y", C.c_double), ("voce_S", C.c_double), ("voce_D", C.c_double), ("lin_K", C.c_double),
        ("yc", C.c_double * 19),
        ("max_iters", C.c_int32), ("ls_max_evals", C.c_int32),
        ("abs_tol", C.c_double), ("rel_tol", C.c_double),
        ("ls_c1", C.c_double), ("ls_lo", C.c_double), ("ls_hi", C.c_double),
        ("nn_weights", C.c_void_p), ("nn_nlayers", C.c_int32), ("nn_widths", C.c_int32 * 7),
        ("beta_equivalent_stress", C.c_double), ("beta_abs_tol", C.c_double), ("beta_rel_tol", C.c_double),
        ("beta_max_iters", C.c_int32), ("ls_kind", C.c_int32),
        ("hnn_width", C.c_int32), ("hnn_offset", C.c_int32),
        ("hnn_nhidden", C.c_int32), ("hnn_widths", C.c_int32 * 4), ("reserved_tail", C.c_int32),
    ]


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def _assert_single_hip_runtime():
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    if len(paths) > 1:
        raise RuntimeError("two HIP runtimes are mapped in this process (" + ", ".join(sorted(paths)) +
                           "); import torch before anything that loads libamdhip64")


def lib():
    """Load libcmad_hip.so (built in-tree by cmad_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # The library must bind to the SAME HIP runtime instance that owns the device arrays and streams it is
    # handed.  torch's ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7); importing torch
    # first makes the dynamic loader resolve our NEEDED entry to that already-loaded runtime instead of a
    # second copy from /opt/rocm (two runtimes in one process -> hipErrorNoDevice on the first launch).
    import torch  # noqa: F401  (device-array carrier; loads its HIP runtime)
    # CMAD_HIP_LIB=/path/to/variant.so: A/B measurements of library builds without overwriting the in-tree build output
    LIB = os.environ.get("CMAD_HIP_LIB") or globals()["LIB"]
    if not os.path.exists(LIB):
        raise HipLibraryMissing(
            f"{LIB} not found: run `python -m cmad_amd.build` (needs hipcc). "
            "cmad_amd has no CPU fallback; the HIP extension is required.")
    L = C.CDLL(LIB)
    _assert_single_hip_runtime()
    vp, i64, dp = C.c_void_p, C.c_int64, C.c_void_p
    md = C.POINTER(ModelDesc)
    L.cm_abi_version.restype = C.c_int
    L.cm_last_hip_error.restype = C.c_char_p
    L.cm_num_xi.argtypes = [md]; L.cm_num_xi.restype = C.c_int
    L.cm_num_gradu.argtypes = [md]; L.cm_num_gradu.restype = C.c_int
    L.cm_workspace_bytes.argtypes = [i64]; L.cm_workspace_bytes.restype = i64
    L.cm_update.argtypes = [md, i64, dp, dp, dp, dp, vp, vp]; L.cm_update.restype = C.c_int
    L.cm_update_workspace_bytes.argtypes = [i64]; L.cm_update_workspace_bytes.restype = i64
    L.cm_update_ws.argtypes = [md, i64, dp, dp, dp, dp, vp, vp, i64, vp]; L.cm_update_ws.restype = C.c_int
    L.cm_update_tangent_ws.argtypes = [md, i64, dp, dp, dp, dp, dp, vp, vp, i64, vp]; L.cm_update_tangent_ws.restype = C.c_int
    L.cm_update_rate.argtypes = [md, i64, dp, dp, dp, dp, dp, vp, vp]; L.cm_update_rate.restype = C.c_int
    L.cm_update_rate_tangent.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, vp, vp]; L.cm_update_rate_tangent.restype = C.c_int
    L.cm_update_rate_vjp.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, dp, dp, vp, i64, vp]; L.cm_update_rate_vjp.restype = C.c_int
    L.cm_update_rate_and_vjp.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, dp, vp, i64, vp]; L.cm_update_rate_and_vjp.restype = C.c_int
    L.cm_objective_grad_rate.argtypes = [md, i64, dp, dp, dp, dp, C.POINTER(C.c_double), dp, dp, vp, i64, vp]
    L.cm_objective_grad_rate.restype = C.c_int
    L.cm_adjoint_step_rate.argtypes = [md, i64, dp, dp, dp, dp, dp, C.POINTER(C.c_double), dp, dp, dp, C.c_int, vp, i64, vp]
    L.cm_adjoint_step_rate.restype = C.c_int
    L.cm_objective_grad_history.argtypes = [md, i64, C.c_int32, dp, dp, C.POINTER(C.c_double), dp, dp, dp, vp, i64, vp]
    L.cm_objective_grad_history.restype = C.c_int
    L.cm_direct_step.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, dp, vp]; L.cm_direct_step.restype = C.c_int
    L.cm_adjoint_history.argtypes = [md, i64, C.c_int32, dp, dp, dp, dp, dp, dp, dp, vp, i64, vp]; L.cm_adjoint_history.restype = C.c_int
    L.cm_direct_workspace_bytes.argtypes = [i64]; L.cm_direct_workspace_bytes.restype = i64
    L.cm_direct_history.argtypes = [md, i64, C.c_int32, dp, dp, dp, dp, dp, dp, dp, vp, i64, vp]; L.cm_direct_history.restype = C.c_int
    L.cm_hessian_workspace_bytes.argtypes = [md, i64, C.c_int32]; L.cm_hessian_workspace_bytes.restype = i64
    L.cm_hessian_history.argtypes = [md, i64, C.c_int32, dp, dp, dp, dp, dp, C.POINTER(C.c_double), dp, dp, dp, vp, i64, vp]
    L.cm_hessian_history.restype = C.c_int
    L.cm_hessian_ep_workspace_bytes.argtypes = [md, i64, C.c_int32, C.c_int32]; L.cm_hessian_ep_workspace_bytes.restype = i64
    L.cm_direct_history_ep.argtypes = [md, i64, C.c_int32, C.c_int32, vp, dp, dp, dp, vp]; L.cm_direct_history_ep.restype = C.c_int
    L.cm_hessian_history_ep.argtypes = [md, i64, C.c_int32, C.c_int32, vp, dp, dp, dp, dp, dp, dp, C.POINTER(C.c_double), dp, dp, dp, vp, i64, vp]
    L.cm_hessian_history_ep.restype = C.c_int
    L.cm_param_blocks.argtypes = [md, i64, C.c_int32, vp, dp, dp, dp, dp, dp, dp, vp]; L.cm_param_blocks.restype = C.c_int
    L.cm_update_complex.argtypes = [md, i64, C.POINTER(C.c_double), dp, dp, dp, dp, dp, dp, dp, vp, vp]; L.cm_update_complex.restype = C.c_int
    L.cm_param_adjoint_history.argtypes = [md, i64, C.c_int32, C.c_int32, vp, dp, dp, dp, dp, dp, vp, i64, vp]
    L.cm_param_adjoint_history.restype = C.c_int
    L.cm_update_history.argtypes = [md, i64, C.c_int32, dp, dp, dp, dp, vp, vp]; L.cm_update_history.restype = C.c_int
    L.cm_update_tangent.argtypes = [md, i64, dp, dp, dp, dp, dp, vp, vp]; L.cm_update_tangent.restype = C.c_int
    L.cm_update_vjp.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, dp, vp, i64, vp]; L.cm_update_vjp.restype = C.c_int
    L.cm_update_and_vjp.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, vp, i64, vp]; L.cm_update_and_vjp.restype = C.c_int
    L.cm_sizeof_model_desc.restype = C.c_int
    if L.cm_sizeof_model_desc() != C.sizeof(ModelDesc):
        raise RuntimeError("cm_model_desc layout mismatch between include/cmad_hip.h and cmad_amd/_lib.py")
    L.cm_evaluate.argtypes = [md, i64, C.c_int, dp, dp, dp, dp, dp, dp, dp, vp]; L.cm_evaluate.restype = C.c_int
    L.cm_evaluate_rate.argtypes = [md, i64, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, vp]; L.cm_evaluate_rate.restype = C.c_int
    L.cm_hessians.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, dp, vp]; L.cm_hessians.restype = C.c_int
    L.cm_hessians_rate.argtypes = [md, i64, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, vp]; L.cm_hessians_rate.restype = C.c_int
    L.cm_objective_grad.argtypes = [md, i64, dp, dp, dp, C.POINTER(C.c_double), dp, dp, vp, i64, vp]
    L.cm_objective_grad.restype = C.c_int
    L.cm_adjoint_step.argtypes = [md, i64, dp, dp, dp, dp, C.POINTER(C.c_double), dp, dp, dp, C.c_int, vp, i64, vp]
    L.cm_adjoint_step.restype = C.c_int
    _lib = L
    return L


def check(rc, what):
    if rc == CM_OK:
        return
    if rc == CM_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: model/def_type/yield combination not available in the HIP library")
    names = {CM_ERR_BAD_ARG: "bad argument", CM_ERR_LAUNCH: "kernel launch failed", CM_ERR_WORKSPACE: "workspace too small"}
    extra = f" ({lib().cm_last_hip_error().decode()})" if rc == CM_ERR_LAUNCH else ""
    raise RuntimeError(f"{what}: {names.get(rc, rc)}{extra}")
