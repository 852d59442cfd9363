"""reference lib/models/prroi_pool: `PrRoIPool2D`, `prroi_pool2d` on the HIP kernel."""
from .prroi_pool import PrRoIPool2D  # noqa: F401
from .functional import prroi_pool2d  # noqa: F401
