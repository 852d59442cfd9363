"""reference lib/utils/track_utils.py."""
from usot_amd.hostutils import (get_subwindow_tracking, im_to_torch, load_yaml, python2round,  # noqa: F401
                                to_torch)
