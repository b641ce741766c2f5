"""`SmallRateElasticPlastic`: small-strain rate-form elastic-plastic model whose local unknown is the
material Cauchy stress.  Host mirror of /root/reference/cmad/models/small_rate_elastic_plastic.py:103-247
(constructor contract and state layout: "cauchy" sym tensor, "alpha", plane-stress stretch); the residual
(:249-346) runs in `cm_update_rate`.

Built on the device for all three deformation types: the stress update (`cm_update_rate`) and the stateful evaluate
surface (`cm_evaluate_rate`: residual, every Jacobian block including d/dU_prev, Sigma, dSigma), so the material-point
objectives run on it unchanged; second derivatives through `cm_hessians_rate`.  FULL_3D and PLANE_STRESS use hand-derived
blocks; UNIAXIAL_STRESS (12 local dofs, :171-196) gets its blocks by forward-mode evaluation of the residual inside the
same kernels (cmad_amd/csrc/cm_rate_uniaxial.hpp).  The batched forward tangent is `cm_update_rate_tangent`
(`DeviceEvaluator.update_rate(tangent=True)`); the batched reverse sweep is `cm_update_rate_vjp` /
`cm_update_rate_and_vjp` / `cm_objective_grad_rate` / `cm_adjoint_step_rate` (the `gradu_prev=` keyword of the
`DeviceEvaluator` methods), which is what `BatchedCalibrationObjective` runs on this model."""
from __future__ import annotations

from typing import ClassVar

import numpy as np

from .. import _lib
from ..parameters.parameters import Parameters
from .deformation_types import DefType, def_type_ndims
from .deriv_types import DerivType
from .device import NewtonSettings
from .model import Model, _sym3
from .var_types import VarType, get_num_eqs


class SmallRateElasticPlastic(Model):
    supports_mixed: ClassVar[bool] = False
    registry_name: ClassVar[str] = "small_rate_elastic_plastic"
    _model_kind = 1

    def __init__(self, parameters: Parameters, def_type: int = DefType.FULL_3D,
                 elastic_stress_fun=None, effective_stress_fun=None, hardening_funs=None,
                 yield_tol: float = 1e-14, uniaxial_stress_idx: int = 0, is_complex: bool = False) -> None:
        if hardening_funs is not None:
            # {"neural network": SimpleNeuralNetwork(...).evaluate}: the reference's examples/noisy_calibration.py:245-252
            from ..neural_networks.simple_neural_network import hardening_network_scales
            self._hardening_nn = hardening_network_scales(hardening_funs)
        if elastic_stress_fun is not None or effective_stress_fun is not None:
            raise NotImplementedError("custom elastic / effective-stress callables have no HIP kernel")
        self._is_complex = is_complex
        self.dtype = complex if is_complex else float
        self._def_type = int(def_type)
        self._ndims = def_type_ndims(def_type)
        self._yield_tol = float(yield_tol)
        self._uniaxial_stress_idx = int(uniaxial_stress_idx)
        if def_type == DefType.FULL_3D:
            num_residuals = 2
        elif def_type == DefType.PLANE_STRESS:
            num_residuals = 3
        elif def_type == DefType.UNIAXIAL_STRESS:
            num_residuals = 4                       # reference :145-146
        else:
            raise NotImplementedError
        self._init_residuals(num_residuals)
        self.var_names[0] = "cauchy"; self.resid_names[0] = "stress rate"
        self._var_types[0] = VarType.SYM_TENSOR
        self._num_eqs[0] = get_num_eqs(VarType.SYM_TENSOR, 3)
        self.var_names[1] = "alpha"; self.resid_names[1] = "yield surface"
        self._var_types[1] = VarType.SCALAR
        self._num_eqs[1] = 1
        self._init_xi = [np.zeros(6), np.zeros(1)]
        if def_type == DefType.PLANE_STRESS:
            self.var_names[2] = "out of plane stretch"; self.resid_names[2] = "cauchy_33"
            self._var_types[2] = VarType.SCALAR
            self._num_eqs[2] = 1
            self._init_xi += [np.ones(1)]
        elif def_type == DefType.UNIAXIAL_STRESS:                              # reference :181-196
            self.var_names[2] = "off-axis stretches"; self.resid_names[2] = "off-axis normal stress"
            self._var_types[2] = VarType.VECTOR
            self._num_eqs[2] = get_num_eqs(VarType.VECTOR, 2)
            self.var_names[3] = "off-axis delta strains"; self.resid_names[3] = "off-axis shear stress"
            self._var_types[3] = VarType.VECTOR
            self._num_eqs[3] = get_num_eqs(VarType.VECTOR, 3)
            self._init_xi += [np.ones(self._num_eqs[2]), np.zeros(self._num_eqs[3])]
        self._init_state_variables()
        self.set_xi_to_init_vals()
        self.parameters = parameters
        super().__init__()

    @classmethod
    def from_deck(cls, model_section, parameters, def_type):
        return cls(parameters=parameters, def_type=def_type, uniaxial_stress_idx=model_section.get("uniaxial_stress_idx", 0))

    @classmethod
    def material_defaults(cls):
        return {"rotation matrix": [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]}

    def derived_output_field_names(self):
        return ["cauchy"]

    @property
    def has_device_newton(self) -> bool:
        """Every deformation type has a batched Newton kernel (UNIAXIAL_STRESS: 12 local dofs, Jacobian by forward-mode
        evaluation in the kernel, cm_rate_uniaxial.hpp)."""
        return True

    def device_newton(self, max_iters=10, abs_tol=1e-14, rel_tol=1e-14, line_search=None):
        import torch
        st = NewtonSettings(max_iters, abs_tol, rel_tol, line_search or {"max evals": 0}, warm_start=False)     # returns the reference's count
        if self._is_complex:
            return self._complex_newton(st)
        ev = self.device_evaluator(st)
        dev = torch.device("cuda")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 1)).to(dev)
        G = np.asarray(self._U.grad_fields["u"], dtype=np.float64)
        Gp = np.asarray(self._U_prev.grad_fields["u"], dtype=np.float64)
        xi, sig, status = ev.update_rate(t(G), t(Gp), t(self._flat(self._xi_prev)))
        self._xi = [b.astype(self.dtype) for b in self._split(xi.cpu().numpy()[:, 0])]
        s = int(status.cpu().numpy().astype(np.uint32)[0])
        return s & _lib.STATUS_ITERS_MASK, bool(s & _lib.STATUS_CONVERGED)
