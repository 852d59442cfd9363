"""TEST INFRASTRUCTURE — second, independent oracle for Precise RoI Pooling (float64).

Written from the operator's DEFINITION, not from the reference's launcher: PrRoI Pooling
(Jiang et al., "Acquisition of Localization Confidence for Accurate Object Detection", ECCV
2018, eq. 3-4; the definition `lib/models/prroi_pool` implements) makes the feature map
continuous by bilinear interpolation,

    f(x, y) = sum_{i,j} w[j, i] * hat(x - i) * hat(y - j),   hat(t) = max(0, 1 - |t|),

with w = 0 outside the map (`PrRoIPoolingGetData`, prroi_pooling_gpu_impl.cu:37-42, returns 0
there), and pools a bin as the exact integral of f over the bin rectangle divided by its area.
The integrand is separable, so with G(t) = int_{-inf}^{t} hat(s) ds (piecewise quadratic)

    int int_bin f = sum_{i,j} w[j, i] * (G(x2 - i) - G(x1 - i)) * (G(y2 - j) - G(y1 - j))
                  = hy^T W hx,

two hat-integral vectors and one small matrix product per bin — no per-cell case analysis, no
four-corner formula, nothing shared with `prroi_pool_ref.c` / `head_ops.hip` (which restate
the reference's `PrRoIPoolingMatCalculation`, .cu:71-106).  Bin geometry (roi * spatial_scale,
width/height clamped at 0, zero-area bins give 0) follows prroi_pooling_gpu_impl.cu:161-184.

`prroi_pool_quadrature` is a third, brute-force check: a midpoint rule on a fine grid over the
same continuous surface.
"""
import numpy as np


def _G(t):
    """Antiderivative of the unit hat: 0 for t <= -1, rising to 1 for t >= 1."""
    t = np.asarray(t, np.float64)
    left = 0.5 * (t + 1.0) ** 2
    right = 1.0 - 0.5 * (1.0 - t) ** 2
    return np.where(t <= -1.0, 0.0, np.where(t <= 0.0, left, np.where(t <= 1.0, right, 1.0)))


def hat_integrals(lo, hi, n):
    """[n] vector: integral of hat(. - i) over [lo, hi] for i = 0..n-1."""
    i = np.arange(n, dtype=np.float64)
    return _G(hi - i) - _G(lo - i)


def prroi_pool_exact(features, rois, ph, pw, scale=1.0):
    """features [B,C,H,W], rois [R,5] = (batch, x1, y1, x2, y2) -> float64 [R,C,ph,pw]."""
    f = np.asarray(features, np.float64)
    rois = np.asarray(rois, np.float64)
    B, C, H, W = f.shape
    out = np.zeros((rois.shape[0], C, ph, pw), np.float64)
    for r, roi in enumerate(rois):
        b = int(roi[0])
        x1, y1, x2, y2 = (np.float64(np.float32(v)) * np.float64(np.float32(scale)) for v in roi[1:])
        bw = max(x2 - x1, 0.0) / pw
        bh = max(y2 - y1, 0.0) / ph
        area = bw * bh
        if not area > 0.0:
            continue
        for i in range(ph):
            hy = hat_integrals(y1 + i * bh, y1 + (i + 1) * bh, H)
            rows = np.tensordot(f[b], hy, axes=([1], [0]))          # [C, W]
            for j in range(pw):
                hx = hat_integrals(x1 + j * bw, x1 + (j + 1) * bw, W)
                out[r, :, i, j] = rows @ hx / area
    return out


def bilinear_surface(fmap, xs, ys):
    """Zero-extended bilinear interpolation of one [H,W] map at points (xs[k], ys[k])."""
    H, W = fmap.shape
    pad = np.zeros((H + 2, W + 2), np.float64)
    pad[1:-1, 1:-1] = fmap
    xs = np.clip(np.asarray(xs, np.float64), -1.0, W) + 1.0
    ys = np.clip(np.asarray(ys, np.float64), -1.0, H) + 1.0
    x0 = np.minimum(np.floor(xs).astype(int), W)
    y0 = np.minimum(np.floor(ys).astype(int), H)
    ax, ay = xs - x0, ys - y0
    return ((1 - ay) * ((1 - ax) * pad[y0, x0] + ax * pad[y0, x0 + 1])
            + ay * ((1 - ax) * pad[y0 + 1, x0] + ax * pad[y0 + 1, x0 + 1]))


def prroi_pool_quadrature(fmap, x1, y1, x2, y2, ph, pw, n=400):
    """Midpoint rule (n x n points per bin) of the bilinear surface of ONE [H,W] map; error
    O(1/n) at the kinks of the surface.  Slow; a few ROIs only."""
    out = np.zeros((ph, pw))
    bw, bh = max(x2 - x1, 0.0) / pw, max(y2 - y1, 0.0) / ph
    if not bw * bh > 0:
        return out
    u = (np.arange(n) + 0.5) / n
    for i in range(ph):
        for j in range(pw):
            gx, gy = np.meshgrid(x1 + (j + u) * bw, y1 + (i + u) * bh)
            out[i, j] = bilinear_surface(fmap, gx.ravel(), gy.ravel()).mean()
    return out
