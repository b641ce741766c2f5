from .deformation_types import DefType, def_type_ndims  # noqa: F401
from .deriv_types import DerivType  # noqa: F401
from .device import NewtonSettings  # noqa: F401
