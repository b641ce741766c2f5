from .coupled_bridge import local_update_with_tangent  # noqa: F401
