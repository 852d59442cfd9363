"""Shared random RoI sets for the PrRoIPool parity tests (CPU oracle and GPU kernel)."""
import numpy as np


def random_rois(seed, n, B, H, W):
    """n RoIs [batch, x1, y1, x2, y2]: non-aligned interior boxes, boxes hanging over every edge,
    sub-pixel (tiny) boxes, boxes larger than the map, degenerate (zero / negative extent) boxes."""
    g = np.random.default_rng(seed)
    rois = np.zeros((n, 5), np.float32)
    rois[:, 0] = g.integers(0, B, n)
    for k in range(n):
        kind = k % 6
        if kind == 0:                       # interior, non-aligned
            x1, y1 = g.uniform(0, W - 2), g.uniform(0, H - 2)
            x2, y2 = x1 + g.uniform(0.7, W - 1 - x1 + 0.7), y1 + g.uniform(0.7, H - 1 - y1 + 0.7)
        elif kind == 1:                     # hangs over the left / top edge
            x1, y1 = g.uniform(-6, 0.5), g.uniform(-6, 0.5)
            x2, y2 = g.uniform(1, W - 1), g.uniform(1, H - 1)
        elif kind == 2:                     # hangs over the right / bottom edge
            x1, y1 = g.uniform(0, W - 2), g.uniform(0, H - 2)
            x2, y2 = g.uniform(W - 1, W + 6), g.uniform(H - 1, H + 6)
        elif kind == 3:                     # tiny: bins far smaller than a cell
            x1, y1 = g.uniform(-0.5, W - 0.5), g.uniform(-0.5, H - 0.5)
            x2, y2 = x1 + g.uniform(1e-3, 0.6), y1 + g.uniform(1e-3, 0.6)
        elif kind == 4:                     # larger than the map
            x1, y1 = g.uniform(-9, -1), g.uniform(-9, -1)
            x2, y2 = g.uniform(W, W + 9), g.uniform(H, H + 9)
        else:                               # degenerate or fully outside
            x1, y1 = g.uniform(0, W), g.uniform(0, H)
            x2, y2 = (x1, y1 + 3.0) if k % 12 == 5 else (x1 + W + 20, y1 + 2.0)
            if k % 12 != 5:
                x1 += W + 10
        rois[k, 1:] = (x1, y1, x2, y2)
    return rois
