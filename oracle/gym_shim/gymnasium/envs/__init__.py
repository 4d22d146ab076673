from . import registration  # noqa: F401
from .registration import make, register, registry  # noqa: F401
