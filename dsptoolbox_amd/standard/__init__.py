from . import enums  # noqa: F401
