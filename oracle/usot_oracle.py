"""TEST INFRASTRUCTURE — CPU oracle for the USOT tracking forward pass.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (usot_amd/, lib/) never does.

A functional restatement, driven by a reference-format state dict, of
  a1-a3  backbone            reference lib/models/modules.py:11-58, 104-151
  a4     neck (AdjustLayer)  lib/models/connect.py:294-314
  a5     encoders (`matrix`) lib/models/connect.py:55-74
  a6     xcorr_depthwise     lib/models/connect.py:147-157
  a7     GroupDW             lib/models/connect.py:86-102
  a8     Conf_Fusion         lib/models/connect.py:123-144
  a9     box_tower_reg       lib/models/connect.py:221-281
  a10-12 template/track/extract_memory_feature   lib/models/models.py:164-206
  a13    PrRoIPool forward   -> oracle/prroi_pool_ref.c (restates the .cu)
  a14-16 tracker host math   lib/tracker/usot_tracker.py:133-200, 222-256, 287-350
The arithmetic of conv / batch-norm / max-pool is PyTorch's (third party, the
reference pins pytorch==1.7.1 in preprocessing/install_model.sh:24; here torch 2.10
CPU): the north star names "the reference PyTorch-CPU path" as the parity target.

Pinned against tests/golden/*.npz, which were produced by importing the reference's
own lib/models on PyTorch-CPU (tests/golden/make_golden.py).  PrRoIPool has no
runnable reference: parity unpinned for that op (analytic known answers only).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_BN_EPS = 1e-5

# ----------------------------------------------------------------------------- conv / bn


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, _BN_EPS)


def _conv(sd, p, x, stride=1, pad=0, dil=1):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, pad, dil)


def _bottleneck(sd, p, x, stride, pad, dil, ds):
    """modules.py:37-58.  `ds` = None or (stride, pad) of the 3x3/1x1 shortcut conv."""
    y = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x)))
    y = F.relu(_bn(sd, p + '.bn2', _conv(sd, p + '.conv2', y, stride, pad, dil)))
    y = _bn(sd, p + '.bn3', _conv(sd, p + '.conv3', y))
    if ds is not None:
        x = _bn(sd, p + '.downsample.1', _conv(sd, p + '.downsample.0', x, ds[0], ds[1]))
    return F.relu(y + x)


def backbone(sd, x, stages=False):
    """modules.py:137-151 with the geometry of :18-27 and :104-126 resolved:
    stem 7x7/s2/p0; layer1 1x1 shortcut; layer2.0 3x3/s2/p0 (+3x3/s2/p0 shortcut);
    layer3.0 3x3 d1 p1 (+3x3/s1/p1 shortcut); layer3.1-5 3x3 d2 p2."""
    p = 'features.features'
    s = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x, 2, 0)))
    y = F.max_pool2d(s, 3, 2, 1)
    y = _bottleneck(sd, p + '.layer1.0', y, 1, 1, 1, (1, 0))
    for i in (1, 2):
        y = _bottleneck(sd, p + '.layer1.%d' % i, y, 1, 1, 1, None)
    p1 = y
    y = _bottleneck(sd, p + '.layer2.0', y, 2, 0, 1, (2, 0))
    for i in (1, 2, 3):
        y = _bottleneck(sd, p + '.layer2.%d' % i, y, 1, 1, 1, None)
    p2 = y
    y = _bottleneck(sd, p + '.layer3.0', y, 1, 1, 1, (1, 1))
    for i in range(1, 6):
        y = _bottleneck(sd, p + '.layer3.%d' % i, y, 1, 2, 2, None)
    return ([s, p1, p2], y) if stages else y


def neck(sd, f):
    """connect.py:296 — 1x1 conv + BN, no activation."""
    return _bn(sd, 'neck.downsample.1', _conv(sd, 'neck.downsample.0', f))


# ----------------------------------------------------------------------------- heads

_GEOMS = (('matrix11', (1, 1)), ('matrix12', (2, 1)), ('matrix21', (1, 2)))


def encode(sd, which, t, side):
    """connect.py:55-74: three parallel dilated 3x3 valid convs + BN + ReLU.
    which in {'cls_encode','reg_encode'}; side 'k' (template) or 's' (search)."""
    out = []
    for name, dil in _GEOMS:
        p = 'connect_model.%s.%s_%s' % (which, name, side)
        out.append(F.relu(_bn(sd, p + '.1', _conv(sd, p + '.0', t, 1, 0, dil))))
    return out


def xcorr_depthwise(x, k):
    """connect.py:147-157."""
    b, c, hk, wk = k.shape
    o = F.conv2d(x.reshape(1, b * c, x.shape[2], x.shape[3]), k.reshape(b * c, 1, hk, wk),
                 groups=b * c)
    return o.reshape(b, c, o.shape[2], o.shape[3])


def groupdw(sd, which, zs, xs):
    """connect.py:86-102: softmax(weight)-weighted sum of the three xcorrs."""
    w = F.softmax(sd['connect_model.%s.weight' % which], 0)
    s = 0
    for i in range(3):
        s = s + w[i] * xcorr_depthwise(xs[i], zs[i])
    return s


def conf_fusion(sd, x):
    """connect.py:123-144.  x [B,M,C,H,W] -> [B,C,H,W]."""
    b, m, c, h, w = x.shape
    x = x.reshape(-1, c, h, w)
    pc, pv = 'connect_model.conf_fusion.conf_gen', 'connect_model.conf_fusion.value_gen'
    conf = F.relu(_bn(sd, pc + '.1', _conv(sd, pc + '.0', x, 1, 1)))
    conf = torch.exp(torch.clamp(conf, min=-6, max=4)).reshape(b, m, c, h, w)
    conf = conf / conf.sum(1, keepdim=True)
    val = F.relu(_bn(sd, pv + '.1', _conv(sd, pv + '.0', x, 1, 1))).reshape(b, m, c, h, w)
    return (conf * val).sum(1)


def tower(sd, name, x, n=4):
    """connect.py:178-207: n x (conv3x3 p1 with bias + BN + ReLU)."""
    for i in range(n):
        p = 'connect_model.%s' % name
        x = F.relu(_bn(sd, '%s.%d' % (p, 3 * i + 1), _conv(sd, '%s.%d' % (p, 3 * i), x, 1, 1)))
    return x


def head_offline(sd, xf, zf):
    """connect.py:229-241 -> (bbox, cls_logits, cls_x, dw_cls, dw_reg)."""
    cls_z, cls_x = encode(sd, 'cls_encode', zf, 'k'), encode(sd, 'cls_encode', xf, 's')
    reg_z, reg_x = encode(sd, 'reg_encode', zf, 'k'), encode(sd, 'reg_encode', xf, 's')
    dw_cls = groupdw(sd, 'cls_dw', cls_z, cls_x)
    dw_reg = groupdw(sd, 'reg_dw', reg_z, reg_x)
    t = tower(sd, 'bbox_tower', dw_reg)
    bbox = torch.exp(sd['connect_model.adjust'] * _conv(sd, 'connect_model.bbox_pred', t, 1, 1)
                     + sd['connect_model.bias'])
    cls = 0.1 * _conv(sd, 'connect_model.cls_pred', tower(sd, 'cls_tower', dw_cls), 1, 1)
    return bbox, cls, cls_x, dw_cls, dw_reg


def head_memory(sd, cls_x, mem_k, batch, mem_size):
    """connect.py:248-275 with the search encodes reused (the reference recomputes
    identical values at :251-252).  mem_k [B*M,C,7,7] -> cls_mem logits [B,1,25,25]."""
    mz = encode(sd, 'cls_encode', mem_k, 'k')
    rep = [t.unsqueeze(1).expand(-1, mem_size, -1, -1, -1).reshape(-1, *t.shape[1:])
           for t in cls_x]
    dw = groupdw(sd, 'cls_dw', mz, rep)
    dw = dw.reshape(batch, mem_size, *dw.shape[1:])
    fused = conf_fusion(sd, dw)
    c = tower(sd, 'cls_memory_tower', fused)
    return 0.1 * _conv(sd, 'connect_model.cls_memory_pred', c, 1, 1), dw, fused


def template(sd, z, bbox=None, pr_pool=True):
    """models.py:173-177 + connect.py:294-314.  bbox [B,4] feature coords (x1,y1,x2,y2)."""
    zf = neck(sd, backbone(sd, z))
    if not pr_pool:
        return zf[:, :, 4:-4, 4:-4]
    return prpool_feature(zf, bbox)


def track(sd, x, zf, template_mem=None, score_mem=None):
    """models.py:179-198 -> (cls, bbox, cls_mem, xf)."""
    xf = neck(sd, backbone(sd, x))
    bbox, cls, cls_x, _, _ = head_offline(sd, xf, zf)
    if template_mem is None:
        return cls, bbox, None, None
    b, m = score_mem.shape
    cls_mem, _, _ = head_memory(sd, cls_x, template_mem, b, m)
    return cls, bbox, cls_mem, xf


# ----------------------------------------------------------------------------- PrRoIPool

_prlib = None


def _load_prlib():
    global _prlib
    if _prlib is None:
        path = os.path.join(_HERE, 'libprroi_ref.so')
        src = os.path.join(_HERE, 'prroi_pool_ref.c')
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            import subprocess
            subprocess.check_call(['make', '-C', _HERE, 'libprroi_ref.so'])
        _prlib = ctypes.CDLL(path)
        _prlib.prroi_pool_forward_ref.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_float]
        _prlib.prroi_pool_forward_ref.restype = None
        _prlib.prroi_pool_backward_ref.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 7 + [ctypes.c_float]
        _prlib.prroi_pool_backward_ref.restype = None
        _prlib.prroi_pool_coor_backward_ref.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_float]
        _prlib.prroi_pool_coor_backward_ref.restype = None
    return _prlib


def prroi_pool(features, rois, ph=7, pw=7, scale=1.0):
    """features [B,C,H,W], rois [R,5] (batch,x1,y1,x2,y2) -> [R,C,ph,pw]; C restatement."""
    lib = _load_prlib()
    f = np.ascontiguousarray(features.detach().cpu().numpy(), dtype=np.float32)
    r = np.ascontiguousarray(rois.detach().cpu().numpy(), dtype=np.float32)
    b, c, h, w = f.shape
    out = np.zeros((r.shape[0], c, ph, pw), np.float32)
    if r.shape[0]:
        lib.prroi_pool_forward_ref(f.ctypes.data, r.ctypes.data, out.ctypes.data,
                                   r.shape[0], c, h, w, ph, pw, ctypes.c_float(scale))
    return torch.from_numpy(out)


def _f32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float32)


def prroi_pool_backward(feature_shape, rois, top_diff, ph=7, pw=7, scale=1.0):
    """Feature gradient [B,C,H,W] of sum(top_diff * prroi_pool(features, rois)); C restatement of
    prroi_pooling_gpu_impl.cu:214-272 (sequential sums where the kernel uses atomicAdd)."""
    lib = _load_prlib()
    b, c, h, w = feature_shape
    r, g = _f32(rois), _f32(top_diff)
    out = np.zeros((b, c, h, w), np.float32)
    lib.prroi_pool_backward_ref(r.ctypes.data, g.ctypes.data, out.ctypes.data, r.shape[0], b, c, h, w, ph, pw, ctypes.c_float(scale))
    return torch.from_numpy(out)


def prroi_pool_coor_backward(features, rois, top_data, top_diff, ph=7, pw=7, scale=1.0):
    """RoI gradient [R,5] (column 0 zero); C restatement of prroi_pooling_gpu_impl.cu:274-380."""
    lib = _load_prlib()
    f, r, t, g = _f32(features), _f32(rois), _f32(top_data), _f32(top_diff)
    b, c, h, w = f.shape
    out = np.zeros((r.shape[0], 5), np.float32)
    lib.prroi_pool_coor_backward_ref(f.ctypes.data, r.ctypes.data, t.ctypes.data, g.ctypes.data, out.ctypes.data,
                                     r.shape[0], c, h, w, ph, pw, ctypes.c_float(scale))
    return torch.from_numpy(out)


def prpool_feature(features, boxes):
    """models.py:164-171: prepend the batch index, pool 7x7 at scale 1."""
    idx = torch.arange(features.shape[0], dtype=torch.float32).reshape(-1, 1)
    return prroi_pool(features, torch.cat([idx, boxes.float().cpu()], 1), 7, 7, 1.0)


# ----------------------------------------------------------------------------- tracker host math


class Hyper:
    """usot_tracker.py:366-394 defaults overlaid with experiments/test/USOT.yaml."""
    penalty_k = 0.021
    window_influence = 0.321
    lr = 0.730
    exemplar_size = 127
    instance_size = 255
    total_stride = 8
    context_amount = 0.5
    tf_size = 15
    ratio = 0.3
    mem_queue_size = 7

    def __init__(self, instance_size=255):
        self.instance_size = instance_size
        self.score_size = (instance_size - self.exemplar_size) // self.total_stride + 1 + 8
        self.sf_size = self.score_size


def grids(p):
    """usot_tracker.py:287-317 -> (grid_x, grid_y, template_axis_minmax, search_axis)."""
    sz = p.score_size
    ax = (np.arange(sz) - float(sz // 2)) * p.total_stride + p.instance_size // 2
    gx, gy = np.meshgrid(ax, ax)
    tax = (np.arange(p.tf_size) - float(p.tf_size // 2)) * p.total_stride + p.exemplar_size // 2
    sax = (np.arange(p.sf_size) - float(p.sf_size // 2)) * p.total_stride + p.instance_size // 2
    return gx, gy, tax, sax


def pool_label_template(p, bbox):
    """usot_tracker.py:319-327."""
    _, _, tax, _ = grids(p)
    lo, hi = tax[0], tax[-1]
    b = np.clip(np.array(bbox, np.float32), lo, hi)
    return (b - lo) * (2 * (p.tf_size // 2) / (hi - lo))


def pool_label_search(p, bbox):
    """usot_tracker.py:329-350 (25-point axis on the 31-wide map, kept as is)."""
    _, _, _, sax = grids(p)
    lo, hi = sax[0], sax[-1]
    slope = 2 * (p.sf_size // 2) / (hi - lo)
    gap = 1.0 / slope
    b = np.clip(np.array(bbox, np.float32), lo - gap, hi + gap)
    return (b - lo) * slope


def _change(r):
    return np.maximum(r, 1.0 / r)


def _sz(w, h):
    pad = (w + h) * 0.5
    return np.sqrt((w + pad) * (h + pad))


def decode(p, cls_logits, cls_mem_logits, bbox, target_pos, target_sz_scaled, window, scale_z):
    """usot_tracker.py:138-193.  Maps are numpy [S,S] / [4,S,S]; `target_sz_scaled` is
    target_sz*scale_z as passed at :258.  Returns (pos, sz, score, box_in_crop, (r,c))."""
    sig = lambda a: (1.0 / (1.0 + np.exp(-a.astype(np.float32)))).astype(np.float32)
    score = p.ratio * sig(cls_logits) + (1 - p.ratio) * sig(cls_mem_logits)
    gx, gy, _, _ = grids(p)
    x1, y1 = gx - bbox[0], gy - bbox[1]
    x2, y2 = gx + bbox[2], gy + bbox[3]
    s_c = _change(_sz(x2 - x1, y2 - y1) / _sz(target_sz_scaled[0], target_sz_scaled[1]))
    r_c = _change((target_sz_scaled[0] / target_sz_scaled[1]) / ((x2 - x1) / (y2 - y1)))
    penalty = np.exp(-(r_c * s_c - 1) * p.penalty_k)
    pscore = penalty * score
    pscore = pscore * (1 - p.window_influence) + window * p.window_influence
    r, c = np.unravel_index(pscore.argmax(), pscore.shape)
    bx1, by1, bx2, by2 = x1[r, c], y1[r, c], x2[r, c], y2[r, c]
    dx = ((bx1 + bx2) / 2 - p.instance_size // 2) / scale_z
    dy = ((by1 + by2) / 2 - p.instance_size // 2) / scale_z
    pw, ph = (bx2 - bx1) / scale_z, (by2 - by1) / scale_z
    tsz = np.asarray(target_sz_scaled, np.float64) / scale_z
    lr = penalty[r, c] * score[r, c] * p.lr
    rw = pw * lr + (1 - lr) * tsz[0]
    rh = ph * lr + (1 - lr) * tsz[1]
    pos = np.array([target_pos[0] + dx, target_pos[1] + dy])
    sz = tsz * (1 - lr) + lr * np.array([rw, rh])
    return pos, sz, score[r, c], [bx1, by1, bx2, by2], (int(r), int(c))


def select_memory(p, confidences):
    """usot_tracker.py:222-256: indices into the memory list for the N_q-2 dynamic slots
    (the two init slots are prepended by the caller).  Index formula kept literally."""
    n = len(confidences)
    k = p.mem_queue_size - 3
    if n <= 1:
        return [0] * (k + 1)
    gap = (n - 1) / k
    idx = []
    for i in range(k):
        a = min(int(int(i * gap) * n), n - 1)
        b = min(int(int((i + 1) * gap) * n), n - 1)
        if a >= b:
            idx.append(a)
        else:
            idx.append(int(np.argmax(np.array(confidences[a:b]))) + a)
    idx.append(n - 1)
    return idx
