"""`USOT` / `USOT_`: the reference's model-construct API (lib/models/models.py:16-40,
164-206, 298-306) over the HIP engine.

Same constructor signature, attribute names (`features`, `neck`, `connect_model`, `zf`,
`pr_pool`, ...), state-dict keys and inference methods; the tensor math behind them runs
in libusot_hip.so.  Training (`forward`, the losses, `torch.nn.DataParallel`) is outside
this build's scope and raises.  Construction works without a GPU (the reference touches
the device in __init__, models.py:119-120 / connect.py:219); inference does not.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip
from .engine import Engine
from .net import BackboneSlots, HeadSlots, NeckSlots


class USOT_(nn.Module):
    def __init__(self, mem_size=4, pr_pool=True, search_size=255, score_size=25, maximum_batch=16, sf_size=25):
        super().__init__()
        self.features = None
        self.connect_model = None
        self.zf = None
        self.neck = None
        self.search_size = search_size
        self.score_size = score_size
        self.search_feature_size = sf_size
        self.maximum_batch = maximum_batch if self.training else 1
        self.mem_size = mem_size
        self.pr_pool = pr_pool
        self._engine = None
        # Engine(...) keyword arguments; 'options' = per-engine overrides of usot_amd.engine.OPTIONS (the lowering's switches)
        self.engine_options = {'graphs': True, 'tuning': None, 'lanes': 0, 'options': None}
        self.grids()

    # ------------------------------------------------------------------ engine lifetime
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._engine = None
        return out

    @property
    def engine(self):
        if self._engine is None:
            dev = next(self.parameters()).device
            self._engine = Engine(self, dev, **{k: v for k, v in self.engine_options.items() if v is not None})
        return self._engine

    # ------------------------------------------------------------------ host-side helpers
    def grids(self):
        """models.py:102-129 (numpy only; the training-time device grids are not built)."""
        sz, stride = self.score_size, 8
        ax = (np.arange(0, sz) - np.floor(float(sz // 2))) * stride + self.search_size // 2
        self.grid_to_search_x, self.grid_to_search_y = np.meshgrid(ax, ax)
        sf = self.search_feature_size
        self.search_area_x_axis = (np.arange(0, sf) - np.floor(float(sf // 2))) * stride + self.search_size // 2

    # ------------------------------------------------------------------ inference API
    def feature_extractor(self, x):
        """models.py:39-40 -> ([stem, p1, p2], p3) as NCHW-shaped views of NHWC buffers."""
        e = self.engine
        e.features(x)
        p = e._feat[(x.shape[0], x.shape[2])]
        view = lambda t: t.permute(0, 3, 1, 2)
        return [view(t) for t in p['stages'][:3]], view(p['stages'][3])

    def prpool_feature(self, features, bboxs):
        """models.py:164-171."""
        return self.engine.pool(features, bboxs)

    def template(self, z, template_bbox=None):
        """models.py:173-177."""
        self.zf = self.engine.template(z, template_bbox, pr_pool=self.pr_pool)

    def track(self, x, template_mem=None, score_mem=None):
        """models.py:179-198."""
        return self.engine.track(x, self.zf, template_mem, score_mem)

    def extract_memory_feature(self, ori_x=None, xf=None, search_bbox=None):
        """models.py:200-206."""
        if ori_x is not None:
            xf = self.engine.features(ori_x)
        return self.engine.pool(xf, search_bbox)

    def forward(self, *a, **k):
        raise NotImplementedError('training forward (models.py:208-295) is out of scope of the '
                                  'MI355X tracking-forward-pass build')


class USOT(USOT_):
    def __init__(self, settings=None):
        if settings is None:
            settings = {'mem_size': 4, 'pr_pool': True}
        super().__init__(mem_size=settings['mem_size'], pr_pool=settings['pr_pool'],
                         search_size=255, score_size=25, maximum_batch=16, sf_size=25)
        self.features = BackboneSlots()
        self.neck = NeckSlots(1024, 256)
        self.connect_model = HeadSlots(256, 4)
