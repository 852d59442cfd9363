"""reference lib/tracker/usot_tracker.py: `USOTTracker`, `USOTConfig`."""
from usot_amd.tracker import USOTConfig, USOTTracker  # noqa: F401
