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


# ---- gradients (training side of the operator; reference prroi_pooling_gpu_impl.cu:214-380) ----------------------
# Also from the definition.  With I = int int_bin f and A = bw * bh the output is out = I / A.
#  * features: out is LINEAR in w, d out / d w[j, i] = hy[j] * hx[i] / A, so the feature gradient is the adjoint
#    of the forward map:  bottom_diff[b, c] += g / A * outer(hy, hx).
#  * coordinates: by Leibniz the integral moves with an edge by the line integral of f along that edge,
#        d I / d xs = -L_x(xs),  d I / d xe = +L_x(xe),   L_x(x) = int_{ys}^{ye} f(x, y) dy = sum_ij w[j,i] hat(x-i) hy[j]
#    (same in y), and A = (xe - xs)(ye - ys), so  d out / d xs = (-L_x(xs) + (ye - ys) * out) / A,
#    d out / d xe = (L_x(xe) - (ye - ys) * out) / A.  The bin edges are affine in the RoI corners:
#    xs = x1 + j * (x2 - x1) / pw, xe = x1 + (j + 1) * (x2 - x1) / pw, times spatial_scale for the raw RoI numbers.
#    Bins of zero area contribute nothing; the batch-index slot of the gradient stays 0.

def _hat(t):
    return np.maximum(0.0, 1.0 - np.abs(np.asarray(t, np.float64)))


def _roi_geometry(roi, scale, ph, pw):
    x1, y1, x2, y2 = (np.float64(np.float32(v)) * np.float64(np.float32(scale)) for v in roi[1:])
    return x1, y1, max(x2 - x1, 0.0) / pw, max(y2 - y1, 0.0) / ph


def prroi_pool_exact_backward(feature_shape, rois, top_diff, ph, pw, scale=1.0):
    """Gradient of sum(top_diff * out) w.r.t. the features: float64 [B,C,H,W]."""
    B, C, H, W = feature_shape
    rois = np.asarray(rois, np.float64)
    g = np.asarray(top_diff, np.float64)
    out = np.zeros((B, C, H, W), np.float64)
    for r, roi in enumerate(rois):
        b = int(roi[0])
        x1, y1, bw, bh = _roi_geometry(roi, scale, ph, pw)
        area = bw * bh
        if not area > 0.0:
            continue
        for i in range(ph):
            hy = hat_integrals(y1 + i * bh, y1 + (i + 1) * bh, H)
            for j in range(pw):
                hx = hat_integrals(x1 + j * bw, x1 + (j + 1) * bw, W)
                out[b] += (g[r, :, i, j] / area)[:, None, None] * np.outer(hy, hx)[None]
    return out


def prroi_pool_exact_coor_backward(features, rois, top_diff, ph, pw, scale=1.0):
    """Gradient of sum(top_diff * out) w.r.t. the RoI rows: float64 [R,5] (column 0 stays 0)."""
    f = np.asarray(features, np.float64)
    rois = np.asarray(rois, np.float64)
    g = np.asarray(top_diff, np.float64)
    B, C, H, W = f.shape
    s = np.float64(np.float32(scale))
    grad = np.zeros((rois.shape[0], 5), np.float64)
    ii, jj = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    for r, roi in enumerate(rois):
        b = int(roi[0])
        x1, y1, bw, bh = _roi_geometry(roi, scale, ph, pw)
        area = bw * bh
        if not area > 0.0:
            continue
        for i in range(ph):
            ys, ye = y1 + i * bh, y1 + (i + 1) * bh
            hy = hat_integrals(ys, ye, H)
            for j in range(pw):
                xs, xe = x1 + j * bw, x1 + (j + 1) * bw
                hx = hat_integrals(xs, xe, W)
                out = np.einsum('chw,h,w->c', f[b], hy, hx) / area
                line_x = lambda x: np.einsum('chw,h,w->c', f[b], hy, _hat(x - ii))       # L_x(x), per channel
                line_y = lambda y: np.einsum('chw,h,w->c', f[b], _hat(y - jj), hx)
                d_xs = (-line_x(xs) + (ye - ys) * out) / area
                d_xe = (line_x(xe) - (ye - ys) * out) / area
                d_ys = (-line_y(ys) + (xe - xs) * out) / area
                d_ye = (line_y(ye) - (xe - xs) * out) / area
                gc = g[r, :, i, j]
                grad[r, 1] += s * np.sum(gc * (d_xs * (1.0 - j / pw) + d_xe * (1.0 - (j + 1) / pw)))
                grad[r, 2] += s * np.sum(gc * (d_ys * (1.0 - i / ph) + d_ye * (1.0 - (i + 1) / ph)))
                grad[r, 3] += s * np.sum(gc * (d_xs * (j / pw) + d_xe * ((j + 1) / pw)))
                grad[r, 4] += s * np.sum(gc * (d_ys * (i / ph) + d_ye * ((i + 1) / ph)))
    return grad
