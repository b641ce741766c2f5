from .deformation_types import DefType, def_type_ndims  # noqa: F401
from .deriv_types import DerivType  # noqa: F401
from .device import HybridHillEffectiveStress, NewtonSettings, ScaledHybridHillEffectiveStress  # noqa: F401
from .global_fields import GlobalFieldsAtPoint, mp_U_from_F  # noqa: F401
from .model import Model  # noqa: F401
from .nonlinear_solver import make_newton_solve, newton_solve  # noqa: F401
from .small_elastic_plastic import SmallElasticPlastic  # noqa: F401
from .small_rate_elastic_plastic import SmallRateElasticPlastic  # noqa: F401
