"""Empty stand-in for pygame (imported at module level by the reference, used only for human rendering)."""
from . import freetype  # noqa: F401
