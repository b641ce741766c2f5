from .parameters import Parameters  # noqa: F401
