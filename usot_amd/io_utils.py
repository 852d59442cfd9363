"""Checkpoint ingest and result-file geometry helpers used by the reference's test script
(lib/utils/train_utils.py:92-156, lib/utils/test_utils.py:4-86)."""
import numpy as np
import torch


def remove_prefix(state_dict, prefix):
    """train_utils.py:131-137."""
    print('remove prefix \'{}\''.format(prefix))
    strip = lambda k: k.split(prefix, 1)[-1] if k.startswith(prefix) else k
    return {strip(k): v for k, v in state_dict.items()}


def check_keys(model, pretrained_state_dict, print_unuse=True):
    """train_utils.py:140-156: report key overlap, require at least one match."""
    ckpt, own = set(pretrained_state_dict.keys()), set(model.state_dict().keys())
    missing = sorted(k for k in own - ckpt if 'num_batches_tracked' not in k)
    print('missing keys:{}'.format(missing))
    if print_unuse:
        print('unused checkpoint keys:{}'.format(sorted(ckpt - own)))
    assert len(own & ckpt) > 0, 'load NONE from pretrained checkpoint'
    return True


def load_pretrain(model, pretrained_path, print_unuse=True, gpus=None):
    """train_utils.py:92-128: accepts a raw state dict or {'state_dict': ...}, strips
    'module.' / 'feature_extractor.' prefixes, remaps MoCo backbones (1x1 shortcut kernels
    embedded at the centre of the 3x3 ones), loads with strict=False."""
    print('load pretrained model from {}'.format(pretrained_path))
    if torch.cuda.is_available():
        if gpus is not None:
            torch.cuda.set_device(gpus[0])
        loc = 'cuda:%d' % torch.cuda.current_device()
    else:
        loc = 'cpu'
    sd = torch.load(pretrained_path, map_location=loc)
    if 'state_dict' in sd.keys():
        sd = sd['state_dict']
    sd = remove_prefix(remove_prefix(sd, 'module.'), 'feature_extractor.')
    if 'moco' in pretrained_path:
        widen = ('encoder_q.layer2.0.downsample.0.weight', 'encoder_q.layer3.0.downsample.0.weight')
        out = {}
        for k, v in sd.items():
            if 'encoder_q' not in k:
                continue
            if k in widen:
                w = torch.zeros(v.shape[0], v.shape[1], 3, 3, dtype=v.dtype, device=v.device)
                w[:, :, 1, 1] = v[:, :, 0, 0]
                v = w
            out[k.replace('encoder_q', 'features.features')] = v
        sd = out
    check_keys(model, sd, print_unuse=print_unuse)
    model.load_state_dict(sd, strict=False)
    return model


def cxy_wh_2_rect(pos, sz):
    """test_utils.py:4-7."""
    return [float(max(float(0), pos[0] - sz[0] / 2)), float(max(float(0), pos[1] - sz[1] / 2)),
            float(sz[0]), float(sz[1])]


def get_axis_aligned_bbox(region):
    """test_utils.py:10-33: 8-value polygon -> (cx, cy, w, h) of the area-matched box."""
    region = np.asarray(region)
    if region.size == 8:
        xs, ys = region[0::2], region[1::2]
        cx, cy = np.mean(xs), np.mean(ys)
        x1, x2, y1, y2 = min(xs), max(xs), min(ys), max(ys)
        a1 = np.linalg.norm(region[0:2] - region[2:4]) * np.linalg.norm(region[2:4] - region[4:6])
        s = np.sqrt(a1 / ((x2 - x1) * (y2 - y1)))
        return cx, cy, s * (x2 - x1) + 1, s * (y2 - y1) + 1
    x, y, w, h = region[0], region[1], region[2], region[3]
    return x + w / 2, y + h / 2, w, h


def _as_poly(v):
    v = np.asarray(v, np.float64)
    assert len(v) in (4, 8)
    if len(v) == 4:
        x, y, w, h = v
        return np.array([[x, y], [x + w, y], [x + w, y + h], [x, y + h]])
    return v.reshape(4, 2)


def _area(poly):
    if len(poly) < 3:
        return 0.0
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _clip(subject, clipper):
    """Sutherland-Hodgman: subject polygon clipped by a CONVEX clipper (counter-clockwise)."""
    def ccw(poly):
        x, y = poly[:, 0], poly[:, 1]
        return poly if (np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) >= 0 else poly[::-1]
    out = list(ccw(np.asarray(subject)))
    cl = ccw(np.asarray(clipper))
    for i in range(len(cl)):
        a, b = cl[i], cl[(i + 1) % len(cl)]
        inp, out = out, []
        if not inp:
            break
        side = lambda pt: (b[0] - a[0]) * (pt[1] - a[1]) - (b[1] - a[1]) * (pt[0] - a[0])
        for j in range(len(inp)):
            cur, prv = inp[j], inp[j - 1]
            sc, sp = side(cur), side(prv)
            if sc >= 0:
                if sp < 0:
                    out.append(prv + (cur - prv) * (sp / (sp - sc)))
                out.append(cur)
            elif sp >= 0:
                out.append(prv + (cur - prv) * (sp / (sp - sc)))
    return np.array(out) if out else np.zeros((0, 2))


def poly_iou(polys1, polys2, bound=None):
    """test_utils.py:35-66 without shapely: IoU of rectangles (x,y,w,h) or 4-corner convex
    polygons (VOT ground truth is rotated rectangles).  `bound` = (w, h) clips both."""
    polys1, polys2 = np.asarray(polys1, np.float64), np.asarray(polys2, np.float64)
    assert polys1.ndim in (1, 2)
    if polys1.ndim == 1:
        polys1, polys2 = polys1[None], polys2[None]
    assert len(polys1) == len(polys2)
    eps = np.finfo(float).eps
    ious = []
    for a, b in zip(polys1, polys2):
        pa, pb = _as_poly(a), _as_poly(b)
        if bound is not None:
            box = _as_poly([0, 0, bound[0], bound[1]])
            pa, pb = _clip(pa, box), _clip(pb, box)
        inter = _area(_clip(pa, pb)) if len(pa) >= 3 and len(pb) >= 3 else 0.0
        union = _area(pa) + _area(pb) - inter
        ious.append(inter / (union + eps))
    return np.clip(ious, 0.0, 1.0)
