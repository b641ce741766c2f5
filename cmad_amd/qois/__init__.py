from .calibration import Calibration  # noqa: F401
from .qoi import QoI  # noqa: F401
