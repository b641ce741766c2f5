"""`SmallElasticPlastic`: small-strain elastic-plastic model with modular effective stress and hardening.
Host mirror of /root/reference/cmad/models/small_elastic_plastic.py:94-347 (constructor contract, state
layout, `from_deck`, `material_defaults`, mixed-formulation helpers); the residual / Cauchy-stress arithmetic
(:237-321) runs in the HIP kernels selected by `def_type` and the first key of
``params["plastic"]["effective stress"]`` (:190-195)."""
from __future__ import annotations

from typing import Any, ClassVar

import numpy as np

from ..parameters.parameters import Parameters
from .deformation_types import DefType, def_type_ndims
from .elastic_constants import ElasticConstants
from .model import Model
from .var_types import VarType, get_num_eqs


class SmallElasticPlastic(Model):
    """Elastic: isotropic linear elasticity.  Plastic: J2 / Hill / Hosford / Barlat / hybrid Hill + network effective stress;
    Voce, linear and / or neural-network isotropic hardening."""

    supports_mixed: ClassVar[bool] = True
    registry_name: ClassVar[str] = "small_elastic_plastic"     # reference :94

    def __init__(self, parameters: Parameters, def_type: int = DefType.FULL_3D,
                 elastic_stress_fun=None, effective_stress_fun=None, hardening_funs=None,
                 yield_tol: float = 1e-14, uniaxial_stress_idx: int = 0, is_complex: bool = False) -> None:
        from .device import HybridHillEffectiveStress
        if isinstance(effective_stress_fun, HybridHillEffectiveStress):
            self._hybrid, effective_stress_fun = effective_stress_fun, None
        if hardening_funs is not None:
            # {"neural network": SimpleNeuralNetwork(...).evaluate} (reference :115, examples/noisy_calibration.py:245-252)
            from ..neural_networks.simple_neural_network import hardening_network_scales
            self._hardening_nn = hardening_network_scales(hardening_funs)
        if elastic_stress_fun is not None or effective_stress_fun is not None:
            # the reference lets callers inject JAX callables (:112-115); the HIP path has a fixed kernel menu
            raise NotImplementedError("custom elastic / effective-stress callables have no HIP kernel; "
                                      "select the yield surface through params['plastic']['effective stress']")
        self._is_complex = is_complex
        self.dtype = complex if is_complex else float
        self._def_type = int(def_type)
        self._ndims = def_type_ndims(def_type)
        self._yield_tol = float(yield_tol)
        self._uniaxial_stress_idx = int(uniaxial_stress_idx)

        if def_type == DefType.FULL_3D:
            num_residuals = 2
        elif def_type in (DefType.PLANE_STRESS, DefType.UNIAXIAL_STRESS):
            num_residuals = 3
        else:
            raise NotImplementedError
        self._init_residuals(num_residuals)

        self.var_names[0] = "plastic strain"; self.resid_names[0] = "flow rule"
        self._var_types[0] = VarType.SYM_TENSOR
        self._num_eqs[0] = get_num_eqs(VarType.SYM_TENSOR, 3)
        self.var_names[1] = "alpha"; self.resid_names[1] = "yield surface"
        self._var_types[1] = VarType.SCALAR
        self._num_eqs[1] = get_num_eqs(VarType.SCALAR, self._ndims)
        self._init_xi = [np.zeros(self._num_eqs[0]), np.zeros(self._num_eqs[1])]
        if def_type == DefType.PLANE_STRESS:
            self.var_names[2] = "out of plane stretch"; self.resid_names[2] = "cauchy_33"
            self._var_types[2] = VarType.SCALAR
            self._num_eqs[2] = get_num_eqs(VarType.SCALAR, self._ndims)
            self._init_xi += [np.ones(self._num_eqs[2])]
        elif def_type == DefType.UNIAXIAL_STRESS:
            self.var_names[2] = "off-axis stretches"; self.resid_names[2] = "off-axis normal stress"
            self._var_types[2] = VarType.VECTOR
            self._num_eqs[2] = get_num_eqs(VarType.VECTOR, 2)
            self._init_xi += [np.ones(self._num_eqs[2])]

        self._init_state_variables()
        self.set_xi_to_init_vals()
        self.parameters = parameters
        super().__init__()

    @classmethod
    def from_deck(cls, model_section: dict, parameters: Parameters, def_type: int) -> "SmallElasticPlastic":
        return cls(parameters=parameters, def_type=def_type,
                   uniaxial_stress_idx=model_section.get("uniaxial_stress_idx", 0))

    @classmethod
    def material_defaults(cls) -> dict:
        return {"rotation matrix": [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]}

    def derived_output_field_names(self):
        return ["cauchy"]

    # mixed-formulation helpers (reference :323-347)
    def dev_cauchy(self, xi, xi_prev, params, U, U_prev):
        c = self.cauchy(xi, xi_prev, params, U, U_prev)
        return c - np.trace(c) / 3. * np.eye(3)

    @staticmethod
    def hydro_cauchy(xi, xi_prev, params, U, U_prev):
        gu = np.asarray(U.grad_fields["u"])
        eps = 0.5 * (gu + gu.T)
        return ElasticConstants.from_params(params["elastic"]).kappa * np.trace(eps)

    @staticmethod
    def pressure_scale_factor(params: dict[str, Any]):
        return ElasticConstants.from_params(params["elastic"]).kappa

    @staticmethod
    def shear_scale_factor(params: dict[str, Any]):
        return ElasticConstants.from_params(params["elastic"]).mu
