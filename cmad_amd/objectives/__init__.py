from .batched import BatchedCalibrationObjective, shard_bounds  # noqa: F401
from .mp_objective import MPAdjointObjective, MPDirectAdjointObjective, MPDirectObjective, MPObjective  # noqa: F401
from .mp_jvp_objective import MPJVPObjective  # noqa: F401
