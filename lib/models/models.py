"""reference lib/models/models.py: `USOT_`, `USOT` (inference API on the HIP engine)."""
from usot_amd.model import USOT, USOT_  # noqa: F401
