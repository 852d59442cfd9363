"""Host-side helpers of the tracking loop (reference lib/utils/track_utils.py:8-127).

Same names, argument meaning and return values as the reference so the tracker and the
reference's scripts can use them unchanged; the implementation is this repo's own.
`cv2.resize` (third party, version unpinned in the reference: preprocessing/
install_model.sh:39) is used when OpenCV is importable; otherwise `resize_bilinear_u8`
restates OpenCV's fixed-point INTER_LINEAR for uint8 — parity with cv2 unpinned here
(no cv2 in the build image), see DESIGN.md.
"""
import os

import numpy as np
import torch
import yaml

_cv2 = None
if os.environ.get('USOT_RESIZE', '') != 'builtin':
    try:                                      # pragma: no cover - not installed in the build image
        import cv2 as _cv2
    except Exception:
        _cv2 = None


def load_yaml(path, subset=True):
    """track_utils.py:8-17."""
    with open(path, 'r') as f:
        obj = yaml.load(f.read(), Loader=yaml.FullLoader)
    return obj['TEST'] if subset else obj


def to_torch(ndarray):
    return torch.from_numpy(ndarray)


def im_to_torch(img):
    """HWC -> CHW float32, values untouched (BGR 0..255, no mean/std). track_utils.py:24-27."""
    return torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)))).float()


def python2round(f):
    """Round half away from zero (python-2 `round`) — track_utils.py:121-127."""
    if round(f + 1) - round(f) != 1:
        return f + abs(f) / f * 0.5
    return round(f)


def _resize_axis(n_src, n_dst):
    """Source index pairs and 11-bit fixed-point weights of OpenCV's INTER_LINEAR, in OpenCV's own
    arithmetic (imgproc resize.cpp, `resize` linear branch): scale = 1 / (dst / src) in double; the
    source coordinate is ROUNDED TO FLOAT before `cvFloor`, and the fraction is a float subtraction;
    coefficients = saturate_cast<short>(c * 2048), i.e. round-half-even."""
    scale = 1.0 / (float(n_dst) / float(n_src))
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= n_src - 1
    f[hi], s[hi] = 0.0, n_src - 1
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    s1 = np.minimum(s + 1, n_src - 1)
    return s, s1, w0, w1


def resize_bilinear_u8(img, dst_w, dst_h):
    """uint8 HxWxC `cv2.resize(img, (dst_w, dst_h))` (default INTER_LINEAR) restated: OpenCV's fixed-point
    two-pass scheme (horizontal pass in int32 with 11-bit weights, vertical pass
    ((w0*(a>>4))>>16) + ((w1*(b>>4))>>16) + 2) >> 2), and its special case: an EXACT 2x downscale in both
    directions is silently switched to INTER_AREA (`resize`: "if interpolation == INTER_LINEAR && is_area_fast
    && iscale_x == 2 && iscale_y == 2"), whose uint8 fast path is (a + b + c + d + 2) >> 2 over each 2x2
    block.  cv2 is not in this image, so this stays UNPINNED; the crop only reaches it when
    the search window is not already model-sized."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    if (h, w) == (dst_h, dst_w):
        return img.copy()
    if w == 2 * dst_w and h == 2 * dst_h:
        q = img.astype(np.int64)
        return ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, ax0, ax1 = _resize_axis(w, dst_w)
    y0, y1, ay0, ay1 = _resize_axis(h, dst_h)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]
    a, b = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (a >> 4)) >> 16) + ((ay1[:, None, None] * (b >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize(img, dst_w, dst_h):
    if _cv2 is not None:                      # pragma: no cover
        return _cv2.resize(img, (dst_w, dst_h))
    return resize_bilinear_u8(img, dst_w, dst_h)


def crop_geometry(im_shape, pos, original_sz):
    """Integer window and padding of the SiamFC crop (track_utils.py:41-57).
    Returns (x0, x1, y0, y1) in PADDED image coordinates and (top, bottom, left, right)."""
    c = (original_sz + 1) / 2
    x0 = round(pos[0] - c)
    x1 = x0 + original_sz - 1
    y0 = round(pos[1] - c)
    y1 = y0 + original_sz - 1
    left = int(max(0., -x0))
    top = int(max(0., -y0))
    right = int(max(0., x1 - im_shape[1] + 1))
    bottom = int(max(0., y1 - im_shape[0] + 1))
    return (x0 + left, x1 + left, y0 + top, y1 + top), (top, bottom, left, right)


def get_subwindow_tracking(im, pos, model_sz, original_sz, avg_chans,
                           target_sz=None, out_mode='torch', need_bbox=False, vis=False):
    """SiamFC-style crop with mean-colour padding (track_utils.py:30-119).
    Returns (patch, crop_info); patch is a CHW float tensor (out_mode='torch') or the raw
    HWC uint8 array."""
    if isinstance(pos, float):
        pos = [pos, pos]
    (cx0, cx1, cy0, cy1), (top, bottom, left, right) = crop_geometry(im.shape, pos, original_sz)
    r, c, k = im.shape
    if top or bottom or left or right:
        canvas = np.zeros((r + top + bottom, c + left + right, k), np.uint8)
        canvas[top:top + r, left:left + c, :] = im
        fill = np.asarray(avg_chans)
        if top:
            canvas[0:top, left:left + c, :] = fill
        if bottom:
            canvas[r + top:, left:left + c, :] = fill
        if left:
            canvas[:, 0:left, :] = fill
        if right:
            canvas[:, c + left:, :] = fill
        mask_shape = canvas.shape[0:2]
    else:
        canvas = im
        mask_shape = im.shape[0:2]
    patch0 = canvas[int(cy0):int(cy1 + 1), int(cx0):int(cx1 + 1), :]
    if not np.array_equal(model_sz, original_sz):
        patch = resize(patch0, model_sz, model_sz)
    else:
        patch = patch0

    info = dict()
    if target_sz is not None:
        tx0 = round(pos[0] - target_sz[0] / 2)
        tx1 = round(pos[0] + target_sz[0] / 2)
        ty0 = round(pos[1] - target_sz[1] / 2)
        ty1 = round(pos[1] + target_sz[1] / 2)
        info['original_image_bbox'] = [tx0, ty0, tx1, ty1]
        if need_bbox:
            # box of the target inside the crop (track_utils.py:89-105; the mixed use of
            # padded/unpadded origins is the reference's and is kept)
            n = patch0.shape[0]
            sx = n / (cx1 - cx0)
            sy = n / (cy1 - cy0)
            g = patch.shape[0] / n
            info['template_bbox'] = [g * (left - 1 + sx * (tx0 - cx0)), g * (top - 1 + sy * (ty0 - cy0)),
                                     g * (left - 1 + sx * (tx1 - cx0)), g * (top - 1 + sy * (ty1 - cy0))]
    info['crop_cords'] = [cx0, cx1, cy0, cy1]
    info['empty_mask'] = np.zeros(mask_shape)
    info['pad_info'] = [top, left, r, c]
    if out_mode == 'torch':
        return im_to_torch(patch.copy()), info
    return patch, info


def flip_lr(image, box):
    """Horizontal flip of an HWC image and an (x1,y1,x2,y2) box — what
    iaa.Sequential([iaa.Fliplr(1)]) does at usot_tracker.py:18-20,112 (imgaug, unpinned)."""
    w = image.shape[1]
    return image[:, ::-1], [w - box[2], box[1], w - box[0], box[3]]
