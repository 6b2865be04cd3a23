"""utils/drops.py of the reference (:8-22)."""
from ..functional import drop_path  # noqa: F401
