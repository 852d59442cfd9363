"""`USOTTracker` / `USOTConfig`: the reference's per-video tracking state machine
(lib/tracker/usot_tracker.py:12-394) over the HIP model.

API kept: `USOTTracker(info)`, `.init(im, target_pos, target_sz, model) -> state`,
`.track(state, im) -> state`, the state-dict keys, `update`, `grids`, `pool_label_*`,
and `USOTConfig` with the yaml overlay from experiments/test/<arch>.yaml.

Differences that do not change results: memory features stay on the device (the
reference moves every 50 KB feature to the CPU and re-uploads 7 of them per frame,
usot_tracker.py:199,258); with a HIP `USOT` model the per-frame decode + memory pooling
run on the device inside the frame's launch plan (`USOTTracker.fused = True`).
"""
import os

import numpy as np
import torch

from .hostutils import flip_lr, get_subwindow_tracking, im_to_torch, load_yaml, python2round


class USOTConfig(object):
    """usot_tracker.py:366-394."""
    penalty_k = 0.021
    window_influence = 0.321
    lr = 0.730
    windowing = 'cosine'
    exemplar_size = 127
    instance_size = 255
    total_stride = 8
    score_size = (instance_size - exemplar_size) // total_stride + 1 + 8
    context_amount = 0.5
    tf_size = 15           # template feature map edge
    sf_size = 25           # axis used for search-feature PrPool boxes (== score_size, v1 design)
    ratio = 0.3            # weight of the offline branch
    mem_queue_size = 7     # N_q

    def update(self, newparam=None):
        if newparam:
            for key, value in newparam.items():
                setattr(self, key, value)
            self.renew()

    def renew(self):
        self.score_size = (self.instance_size - self.exemplar_size) // self.total_stride + 1 + 8


def _axis(n, stride, size):
    return (np.arange(0, n) - np.floor(float(n // 2))) * stride + size // 2


def _dev_of(model):
    try:
        return next(model.parameters()).device
    except Exception:
        return torch.device('cpu')


def search_scale(target_sz, p):
    """(s_x, scale_z) of usot_tracker.py:210-216."""
    hc = target_sz[1] + p.context_amount * sum(target_sz)
    wc = target_sz[0] + p.context_amount * sum(target_sz)
    s_z = np.sqrt(wc * hc)
    scale_z = p.exemplar_size / s_z
    pad = ((p.instance_size - p.exemplar_size) / 2) / scale_z
    return s_z + 2 * pad, scale_z


def select_memory(conf, n_q):
    """Indices into the memory list for the N_q-2 sampled slots (usot_tracker.py:222-256).
    The interval formula is the reference's, kept literally for reproducibility (:237-242)."""
    n = len(conf)
    k = n_q - 3
    if n <= 1:
        return [0] * (k + 1)
    gap = (n - 1) / k
    picks = []
    for i in range(k):
        a = min(int(int(i * gap) * n), n - 1)
        b = min(int(int((i + 1) * gap) * n), n - 1)
        picks.append(a if a >= b else int(np.argmax(np.asarray(conf[a:b]))) + a)
    picks.append(n - 1)
    return picks


class USOTTracker(object):
    fused = True      # use the device-resident frame plan when the model supports it
    device_crop = True   # crop / pad / resize the search window on the device (fused path only)

    def __init__(self, info):
        self.info = info

    # ------------------------------------------------------------------ geometry helpers
    def grids(self, p):
        """usot_tracker.py:287-317."""
        ax = _axis(p.score_size, p.total_stride, p.instance_size)
        self.grid_to_search_x, self.grid_to_search_y = np.meshgrid(ax, ax)
        tax = _axis(p.tf_size, p.total_stride, p.exemplar_size)
        self.grid_to_template = {}
        self.grid_to_template_x, self.grid_to_template_y = np.meshgrid(tax, tax)
        self.search_area_x_axis = _axis(p.sf_size, p.total_stride, p.instance_size)

    def pool_label_template(self, p, bbox):
        """Image box in the 127 crop -> 15x15 feature coordinates (usot_tracker.py:319-327)."""
        lo, hi = self.grid_to_template_x[0][0], self.grid_to_template_x[-1][-1]
        b = np.clip(np.array(bbox, np.float32), a_max=hi, a_min=lo)
        return (b - lo) * (2 * (p.tf_size // 2) / (hi - lo))

    def pool_label_search(self, p, bbox):
        """Image box in the search crop -> search-feature coordinates, on the response
        map's axis by design of USOT v1 (usot_tracker.py:329-350)."""
        lo, hi = self.search_area_x_axis[0], self.search_area_x_axis[-1]
        slope = 2 * (p.sf_size // 2) / (hi - lo)
        gap = 1.0 / slope
        b = np.clip(np.array(bbox, np.float32), a_max=hi + gap, a_min=lo - gap)
        return (b - lo) * slope

    def clip_number(self, num, _max=127.0, _min=0.0):
        return _max if num >= _max else (_min if num <= _min else num)

    def change(self, r):
        return np.maximum(r, 1. / r)

    def sz(self, w, h):
        pad = (w + h) * 0.5
        return np.sqrt((w + pad) * (h + pad))

    def sz_wh(self, wh):
        return self.sz(wh[0], wh[1])

    # ------------------------------------------------------------------ init
    def init(self, im, target_pos, target_sz, model):
        """usot_tracker.py:22-131."""
        model.pr_pool = True
        dev = _dev_of(model)
        p = USOTConfig()
        state = dict(im_h=im.shape[0], im_w=im.shape[1])
        here = os.path.abspath(os.path.dirname(__file__))
        ypath = os.path.join(here, '..', 'experiments', 'test', '%s.yaml' % self.info.arch)
        cfg = load_yaml(ypath, subset=True)
        p.update(cfg)
        p.renew()
        small = (target_sz[0] * target_sz[1]) / float(state['im_h'] * state['im_w']) < 0.004
        p.instance_size = cfg['big_sz'] if small else cfg['small_sz']
        p.renew()
        p.sf_size = p.score_size
        self.grids(p)

        wc = target_sz[0] + p.context_amount * sum(target_sz)
        hc = target_sz[1] + p.context_amount * sum(target_sz)
        s_z = round(np.sqrt(wc * hc))
        avg = np.mean(im, axis=(0, 1))
        z_crop, ci = get_subwindow_tracking(im, target_pos, p.exemplar_size, s_z, avg, target_sz, need_bbox=True)
        zbox = torch.tensor(np.array([self.pool_label_template(p, ci['template_bbox'])])).float().to(dev)
        model.template(z_crop.unsqueeze(0).to(dev), template_bbox=zbox)

        if p.windowing == 'cosine':
            window = np.outer(np.hanning(p.score_size), np.hanning(p.score_size))
        else:
            window = np.ones((int(p.score_size), int(p.score_size)))
        state.update(p=p, net=model, avg_chans=avg, window=window, target_pos=target_pos, target_sz=target_sz)

        # memory queue seeds: the init frame's search crop and its left/right flip
        s_x, _ = search_scale(target_sz, p)
        x_raw, ci = get_subwindow_tracking(im, target_pos, p.instance_size, python2round(s_x), avg, target_sz,
                                           out_mode='raw', need_bbox=True)
        sbox = ci['template_bbox']
        feats = []
        for img, box in ((x_raw, sbox), self._flipped(x_raw, sbox)):
            roi = torch.tensor(np.array([self.pool_label_search(p, box)])).float().to(dev)
            f = model.extract_memory_feature(ori_x=im_to_torch(img.copy()).unsqueeze(0).to(dev), search_bbox=roi)
            feats.append(f.detach())
        state['init_features'] = feats
        state['memory_features'] = [feats[0]]
        state['memory_confidences'] = [0.9]
        if self.fused and hasattr(model, 'engine') and dev.type == 'cuda':
            from .engine import MemoryFeatures
            state['session'] = model.engine.open_session(p, window, feats)
            state['memory_features'] = MemoryFeatures(state['session'])      # list-like view of the device bank
        return state

    def _flipped(self, img, box):
        fimg, fbox = flip_lr(img, box)
        # usot_tracker.py:113-116: x clipped by shape[0], y by shape[1] (square crops)
        fbox = [self.clip_number(fbox[0], _max=fimg.shape[0]), self.clip_number(fbox[1], _max=fimg.shape[1]),
                self.clip_number(fbox[2], _max=fimg.shape[0]), self.clip_number(fbox[3], _max=fimg.shape[1])]
        return fimg, fbox

    # ------------------------------------------------------------------ per-frame
    def update(self, net, x_crops, target_pos, target_sz, window, scale_z, p, template_mem=None, score_mem=None):
        """usot_tracker.py:133-200 with the decode on the host (generic path: any model
        exposing track()/extract_memory_feature())."""
        cls, bbox, cls_mem, xf = net.track(x_crops, template_mem=template_mem, score_mem=score_mem)
        s_off = torch.sigmoid(cls).squeeze().cpu().data.numpy()
        s_on = torch.sigmoid(cls_mem).squeeze().cpu().data.numpy()
        score = p.ratio * s_off + (1 - p.ratio) * s_on
        off = bbox.squeeze().cpu().data.numpy()
        x1 = self.grid_to_search_x - off[0, ...]
        y1 = self.grid_to_search_y - off[1, ...]
        x2 = self.grid_to_search_x + off[2, ...]
        y2 = self.grid_to_search_y + off[3, ...]
        s_c = self.change(self.sz(x2 - x1, y2 - y1) / self.sz_wh(target_sz))
        r_c = self.change((target_sz[0] / target_sz[1]) / ((x2 - x1) / (y2 - y1)))
        penalty = np.exp(-(r_c * s_c - 1) * p.penalty_k)
        pscore = penalty * score * (1 - p.window_influence) + window * p.window_influence
        r, c = np.unravel_index(pscore.argmax(), pscore.shape)
        box = [x1[r, c], y1[r, c], x2[r, c], y2[r, c]]
        pos, sz = self._apply_box(p, box, penalty[r, c], score[r, c], target_pos, target_sz, scale_z)
        roi = torch.tensor(np.array([self.pool_label_search(p, box)])).float().to(xf.device)
        feat = net.extract_memory_feature(xf=xf, search_bbox=roi).detach()
        return pos, sz, score[r, c], feat

    @staticmethod
    def _apply_box(p, box, penalty, score, target_pos, target_sz_scaled, scale_z):
        """usot_tracker.py:171-193: crop coordinates -> image motion and smoothed size."""
        cx, cy = (box[0] + box[2]) / 2, (box[1] + box[3]) / 2
        dx = (cx - p.instance_size // 2) / scale_z
        dy = (cy - p.instance_size // 2) / scale_z
        w, h = (box[2] - box[0]) / scale_z, (box[3] - box[1]) / scale_z
        tsz = target_sz_scaled / scale_z
        lr = penalty * score * p.lr
        rw = w * lr + (1 - lr) * tsz[0]
        rh = h * lr + (1 - lr) * tsz[1]
        pos = np.array([target_pos[0] + dx, target_pos[1] + dy])
        return pos, tsz * (1 - lr) + lr * np.array([rw, rh])

    @staticmethod
    def _conf_array(state, conf):
        """numpy mirror of state['memory_confidences'] (the list the reference keeps, usot_tracker.py:265):
        select_memory takes argmax over quarter-length slices of it every frame, which on a python
        list costs a list -> array conversion that grows with the video (60 us per frame at 2 000
        frames).  The list is APPEND-ONLY as far as this mirror can tell cheaply: it is rebuilt when the list object was
        replaced, shrank, outgrew the buffer, or its last mirrored entry changed; an in-place edit of an EARLIER entry is
        caught by a full comparison every 256 frames (a caller that rewrites history should drop state['_conf_buf'])."""
        buf, n = state.get('_conf_buf'), len(conf)
        m = state.get('_conf_n', 0)
        stale = buf is None or state.get('_conf_id') != id(conf) or m > n or n > len(buf) or (m and buf[m - 1] != conf[m - 1])
        if not stale and m and n % 256 == 0:
            stale = not np.array_equal(buf[:m], np.asarray(conf[:m], np.float64))
        if stale:
            buf = np.empty(max(1024, 2 * n), np.float64)
            buf[:n] = conf
            state['_conf_buf'] = buf
            state['_conf_id'] = id(conf)
        elif n > m:
            buf[m:n] = conf[m:n]
        state['_conf_n'] = n
        return buf[:n]

    def track(self, state, im):
        """usot_tracker.py:202-276."""
        p, net = state['p'], state['net']
        target_pos, target_sz = state['target_pos'], state['target_sz']
        s_x, scale_z = search_scale(target_sz, p)
        conf = state['memory_confidences']
        picks = select_memory(self._conf_array(state, conf), p.mem_queue_size)

        sess = state.get('session') if self.fused else None     # fused may be switched off mid-video
        if sess is None or not self.device_crop:
            x_crop, _ = get_subwindow_tracking(im, target_pos, p.instance_size, python2round(s_x), state['avg_chans'])
        if sess is not None:
            if self.device_crop:
                out = sess.frame_from_image(im, target_pos, python2round(s_x), state['avg_chans'], picks,
                                            target_sz * scale_z)
            else:
                # a crop the tracker built itself is owned by this call until collect() returns: when it already is a dense
                # float32 tensor on the session's device it is read where it lies (no device-to-device snapshot)
                own = (x_crop.is_cuda and x_crop.device == sess.e.device and x_crop.dtype == torch.float32
                       and x_crop.is_contiguous() and x_crop.numel() == sess.x.numel())
                out = sess.frame(x_crop, picks, target_sz * scale_z, inplace=own)
            pos, sz = self._apply_box(p, out[3:7], out[2], out[1], target_pos, target_sz * scale_z, scale_z)
            score = np.float32(out[1])
            # the frame graph has already appended the pooled feature to the session's bank, which
            # state['memory_features'] views (usot_tracker.py:264)
        else:
            dev = _dev_of(net)
            feats = state['memory_features']
            mem = torch.cat(list(state['init_features']) + [feats[i] for i in picks], dim=0).to(dev)
            score_mem = torch.tensor([0.9, 0.9] + [conf[i] for i in picks]).unsqueeze(0).to(dev)
            pos, sz, score, feat = self.update(net, x_crop.unsqueeze(0).to(dev), target_pos, target_sz * scale_z,
                                               state['window'], scale_z, p, template_mem=mem, score_mem=score_mem)
            state['memory_features'].append(feat)
        state['memory_confidences'].append(score)

        pos[0] = max(0, min(state['im_w'], pos[0]))
        pos[1] = max(0, min(state['im_h'], pos[1]))
        sz[0] = max(10, min(state['im_w'], sz[0]))
        sz[1] = max(10, min(state['im_h'], sz[1]))
        state.update(target_pos=pos, target_sz=sz, cls_score=score, p=p)
        return state
