from .calibration import Calibration  # noqa: F401
from .qoi import QoI  # noqa: F401
from .uniaxial_calibration import UniaxialCalibration  # noqa: F401
