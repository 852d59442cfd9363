"""reference lib/utils/test_utils.py (shapely-free polygon IoU)."""
from usot_amd.io_utils import cxy_wh_2_rect, get_axis_aligned_bbox, poly_iou  # noqa: F401
