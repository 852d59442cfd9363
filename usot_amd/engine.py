"""Lowers the USOT forward pass onto libusot_hip.so.

Host-side responsibilities only (no tensor math on activations happens here):
  * fold every BatchNorm into the preceding convolution in float64 and repack the filters
    to the [Cout][kh][kw][Cin] layout the implicit-GEMM kernel streams;
  * merge convolutions that share an input (cls|reg encoders, conf_gen|value_gen) and
    group the three head towers into single launches;
  * own a static NHWC workspace per (batch, crop size, memory size) and record the frame's
    kernel sequence into a native launch plan (optionally a captured hipGraph).

Activation layout is NHWC end to end; NCHW exists only at the API edge (input crop, the
returned cls/bbox maps).  reference call graph: lib/models/models.py:173-206,
lib/models/connect.py:221-281, lib/models/modules.py:137-151.
"""
import ctypes as C
import os
import warnings
import time

import numpy as np
import torch

from . import hip
from .net import ConvSlot, NormSlot

ACT_NONE, ACT_RELU, ACT_EXP, ACT_CONF = hip.ACT_NONE, hip.ACT_RELU, hip.ACT_EXP, hip.ACT_CONF
GEOMS = ('matrix11', 'matrix12', 'matrix21')
KGEO = ((5, 5), (3, 5), (5, 3))       # template-side encoder output sizes for a 7x7 kernel


def fold(conv, bn=None, scale=None, shift=None):
    """(w[O,kh,kw,I] float64, b[O] float64) with eval-mode BN folded in.
    Optional extra affine `scale*y + shift` applied after (bbox/cls head constants)."""
    w = conv.weight.detach().double().cpu()
    b = conv.bias.detach().double().cpu() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64)
    if bn is not None:
        s = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
        w = w * s.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().double().cpu()) * s + bn.bias.detach().double().cpu()
    if scale is not None:
        w = w * scale
        b = b * scale
    if shift is not None:
        b = b + shift
    return w.permute(0, 2, 3, 1).contiguous(), b


class PackedConv:
    """Device-resident folded filter bank + geometry of one (possibly merged) convolution."""

    def __init__(self, device, slots, ws, bs):
        ref = slots[0]
        self.cin, self.kh, self.kw = ref.cin, ref.kh, ref.kw
        self.stride, self.pad, self.dil = ref.stride, ref.pad, ref.dil
        w = torch.cat(ws, 0)
        self.cout = w.shape[0]
        self.w = w.reshape(self.cout, -1).float().contiguous().to(device)
        self.b = torch.cat(bs, 0).float().contiguous().to(device)

    def out_hw(self, h, w):
        oh = (h + 2 * self.pad[0] - self.dil[0] * (self.kh - 1) - 1) // self.stride + 1
        ow = (w + 2 * self.pad[1] - self.dil[1] * (self.kw - 1) - 1) // self.stride + 1
        return oh, ow

    def w_lp_pw_pair(self, dtype, cm, co, cn, which):
        """Low-precision filter bank in the fragment order of the fused pointwise-pair kernel (hip.pw_pair_pack)."""
        cache = self.__dict__.setdefault('_wlp_pp', {})
        key = (dtype, cm, co, cn, which)
        if key not in cache:
            cache[key] = hip.pw_pair_pack(self.w_lp(dtype), cm, co, cn, which)
        return cache[key]

    def w_pw_pair_f32(self):
        """fp32 filter bank in the fragment order of the fused fp32 pointwise-pair kernel (hip.pw_pair_f32_pack)."""
        if '_wpp32' not in self.__dict__:
            self._wpp32 = hip.pw_pair_f32_pack(self.w)
        return self._wpp32

    def w_pw_pair_s16(self):
        """split-fp16 bank + row factors for the split-fp16 form of the fused pair (hip.pw_pair_s16_pack)."""
        if '_wpps16' not in self.__dict__:
            self._wpps16 = hip.pw_pair_s16_pack(self.w)
        return self._wpps16

    def w_frag(self):
        """fp32 filter bank in MFMA fragment order for the weight-streaming / weight-stationary conv tiles
        (hip.pack_wfrag: a permutation inside every 16-row block, so row offsets that are multiples of 16 keep their meaning)."""
        if '_wfrag' not in self.__dict__:
            self._wfrag = hip.pack_wfrag(self.w)
        return self._wfrag

    def w_split16(self):
        """(bank, scales) for the split-fp16 conv tiles (hip.split16_pack: every row as hi + lo fp16 of row x 2^e; row slices keep
        their meaning)."""
        if '_wsplit16' not in self.__dict__:
            self._wsplit16 = hip.split16_pack(self.w)
        return self._wsplit16

    def w_lp(self, dtype=torch.bfloat16):
        """bf16 / fp16 copy of the folded filter bank (made on first use; fp32 stays the master)."""
        cache = self.__dict__.setdefault('_wlp', {})
        if dtype not in cache:
            cache[dtype] = self.w.to(dtype).contiguous()
        return cache[dtype]


def pack(device, pairs, scale=None, shift=None):
    """pairs: [(ConvSlot, NormSlot|None), ...] concatenated along Cout."""
    ws, bs = zip(*[fold(c, n, scale, shift) for c, n in pairs])
    return PackedConv(device, [c for c, _ in pairs], list(ws), list(bs))


class Weights:
    """All folded / packed parameters of one USOT model on one device."""

    def __init__(self, model, device):
        self.device = device
        f = model.features.features
        w = f.conv1.weight.detach().double().cpu()
        s = f.bn1.weight.detach().double().cpu() / torch.sqrt(f.bn1.running_var.detach().double().cpu() + f.bn1.eps)
        self.stem_w = (w * s.view(-1, 1, 1, 1)).permute(1, 2, 3, 0).reshape(147, 64).float().contiguous().to(device)
        b64 = f.bn1.bias.detach().double().cpu() - f.bn1.running_mean.detach().double().cpu() * s
        self.stem_b = b64.float().to(device)
        # the fp32 stem kernels convolve x - STEM_MU[ci] too (exact for this pad-0 conv): their bias carries sum(w) * mu,
        # folded here in float64 with the unrounded filters
        mu = torch.tensor(self.STEM_MU, dtype=torch.float64).view(3, 1, 1, 1)
        self.stem_b_mu = (b64 + ((w * s.view(-1, 1, 1, 1)).permute(1, 2, 3, 0) * mu).sum((0, 1, 2))).float().to(device)
        self._stem_lp = {}
        self.blocks = []
        for layer in (f.layer1, f.layer2, f.layer3):
            for blk in layer:
                ds = pack(device, [(blk.downsample[0], blk.downsample[1])]) if blk.downsample is not None else None
                self.blocks.append((pack(device, [(blk.conv1, blk.bn1)]), pack(device, [(blk.conv2, blk.bn2)]),
                                    pack(device, [(blk.conv3, blk.bn3)]), ds))
        self.neck = pack(device, [(model.neck.downsample[0], model.neck.downsample[1])])
        cm = model.connect_model
        enc = lambda e, g, side: (getattr(e, '%s_%s' % (g, side))[0], getattr(e, '%s_%s' % (g, side))[1])
        # cls rows first, reg rows second: a Cout=256 launch on the same bank is the cls encoder
        self.enc_s = [pack(device, [enc(cm.cls_encode, g, 's'), enc(cm.reg_encode, g, 's')]) for g in GEOMS]
        self.enc_k = [pack(device, [enc(cm.cls_encode, g, 'k'), enc(cm.reg_encode, g, 'k')]) for g in GEOMS]
        self.conf = pack(device, [(cm.conf_fusion.conf_gen[0], cm.conf_fusion.conf_gen[1]),
                                  (cm.conf_fusion.value_gen[0], cm.conf_fusion.value_gen[1])])
        towers = (cm.bbox_tower, cm.cls_tower, cm.cls_memory_tower)          # group order
        self.tower = [pack(device, [(t[3 * i], t[3 * i + 1]) for t in towers]) for i in range(4)]
        adjust = float(cm.adjust.detach().double().cpu())
        shift = cm.bias.detach().double().cpu().reshape(4)
        self.bbox_pred = pack(device, [(cm.bbox_pred, None)], scale=adjust, shift=shift)
        self.cls_preds = pack(device, [(cm.cls_pred, None), (cm.cls_memory_pred, None)], scale=0.1)
        sm = lambda p: torch.softmax(p.detach().float().cpu(), 0).numpy()
        self.cls_wsm, self.reg_wsm = sm(cm.cls_dw.weight), sm(cm.reg_dw.weight)

    # every stem kernel convolves x - STEM_MU[ci] (raw BGR 0..255 crops, test_utils.py
    # feeds them unnormalised) and carries the exact mu term in its bias
    STEM_MU = (104.0, 117.0, 123.0)

    def stem_f32(self):
        if 'f32' not in self._stem_lp:
            self._stem_lp['f32'] = pack_stem_f32(self.stem_w).to(self.device)
        return self._stem_lp['f32']

    def stem_fits_f16(self):
        """The folded stem filters are fp16 material: none near the top of the range, and every output channel's largest
        tap is >= 2^-10, so that the absolute rounding step of the fp16 subnormals (2^-25) stays below 2^-15 of the
        channel's scale (an all-zero channel is fine)."""
        a = self.stem_w.detach().abs().float().reshape(147, 64)
        top = a.max(0).values
        return bool(a.max() < 16384.0 and ((top >= 2.0 ** -10) | (top == 0)).all())

    def stem_lp(self, dtype):
        """(filter fragments in `dtype`, fp32 bias with the folded mu term)."""
        if dtype not in self._stem_lp:
            w = self.stem_w.double().cpu().reshape(3, 49, 64)
            mu = torch.tensor(self.STEM_MU, dtype=torch.float64).view(3, 1, 1)
            bias = (self.stem_b.double().cpu() + (w * mu).sum((0, 1))).float().to(self.device)
            self._stem_lp[dtype] = (pack_stem_lp(self.stem_w, dtype).to(self.device), bias)
        return self._stem_lp[dtype]


def pack_stem_f32(stem_w):
    """[147][64] folded stem filters -> fp32 MFMA A operands [4 cblk][48 kstep][64 lanes] for
    usot_stem_pool_f32: k-step s, lane (l15, quad) of channel block cb holds channel cb*16 + l15, k-row s // 2
    (r = ci*7 + kh; rows 21..23 zero) and tap (s % 2)*4 + quad (tap 7 zero)."""
    w = stem_w.detach().float().cpu().reshape(3, 7, 7, 64)                 # [ci][kh][kw][co]
    rows = torch.zeros(24, 8, 64)
    rows[:21, :7] = w.reshape(21, 7, 64)
    frag = rows.reshape(24, 2, 4, 4, 16).permute(3, 0, 1, 2, 4)            # [cb][row][half][quad][l15]
    return frag.reshape(4, 48, 64).contiguous()


def pack_stem_lp(stem_w, dtype):
    """[147][64] folded stem filters -> MFMA A fragments [4 cblk][6 kstep][64 lanes][8] for
    usot_stem_pool_lp: lane (l15, quad) of channel block cb, k-step ks holds channel cb*16 + l15,
    k-row r = 4*ks + quad (r = ci*7 + kh; rows 21..23 zero) and taps kw 0..6 (+ one zero tap)."""
    w = stem_w.detach().float().cpu().reshape(3, 7, 7, 64)                 # [ci][kh][kw][co]
    rows = torch.zeros(24, 8, 64)
    rows[:21, :7] = w.reshape(21, 7, 64)
    frag = rows.reshape(6, 4, 8, 4, 16).permute(3, 0, 1, 4, 2)             # [cb][ks][quad][l15][8]
    return frag.reshape(4, 6, 64, 8).to(dtype).contiguous()


class Plan:
    """Thin owner of a native launch plan."""

    def __init__(self):
        self.h = C.c_void_p(hip.lib().usot_plan_create())
        if not self.h:
            raise hip.HipError('usot_plan_create failed')
        self.keep = []          # tensors whose pointers are baked into the plan
        self.captured = False

    def __del__(self):
        try:
            if self.h:
                hip.lib().usot_plan_destroy(self.h)
        except Exception:
            pass

    def run(self):
        hip.check(hip.lib().usot_plan_run(self.h, hip.stream()), 'usot_plan_run')

    def capture(self):
        """Capture into a hipGraph on a side stream (legacy stream cannot capture)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            hip.check(hip.lib().usot_plan_capture(self.h, hip.stream()), 'usot_plan_capture')
        torch.cuda.current_stream().wait_stream(s)
        self.captured = True

    def profile(self, frames=10, reps=1):
        """[(kind, tile, ksplit, groups, ms)] per op, eager, HIP events on the current stream."""
        L = hip.lib()
        n = L.usot_plan_size(self.h)
        ms = (C.c_float * n)()
        hip.check(L.usot_plan_profile(self.h, hip.stream(), frames, reps, ms), 'usot_plan_profile')
        out = []
        for i in range(n):
            info = (C.c_int * 4)()
            hip.check(L.usot_plan_op_info(self.h, i, info), 'usot_plan_op_info')
            out.append((info[0], info[1], info[2], info[3], float(ms[i])))
        return out

    def fork(self, lane):
        hip.check(hip.lib().usot_plan_fork(self.h, lane), 'usot_plan_fork')

    def join(self, lane):
        hip.check(hip.lib().usot_plan_join(self.h, lane), 'usot_plan_join')


class Builder:
    """Allocates static buffers and appends ops to a plan."""

    def __init__(self, weights, tuning=None, lanes=True, options=None):
        self.W = weights
        self.opt = merged_options() if options is None else options
        self.dev = weights.device
        self.plan = Plan()
        self.tuning = dict(tuning or {})
        if self.opt['split16_f32']:
            # shapes tuned on the split-fp16 tiles themselves (data/tuning_split16_gfx950.json) replace the fp32 table's entries;
            # an entry that already names a split-fp16 tile (an experiment's override) stays
            for key, val in S16_TUNING.items():
                cur = self.tuning.get(key)
                if cur is None or hip.tile_wfrag(cur[0]) != 2:
                    self.tuning[key] = val
        self.lanes = int(lanes)
        self.batch = True                 # heterogeneous conv batching (one launch for sibling convs)
        self.default_batch_tile = 15      # v2 32x64 BK64 when the lead shape has no tuned entry
        self.log = []           # (name, M, N, K, groups, macs) per conv, for benchmarks
        self.lp_bytes = []      # algorithmic HBM bytes of every low-precision conv launch (operands + result, once each)
        self.lp_readback = []   # bytes a fused launch reads back of its OWN output (phase 5 of csrc/conv_pw_lp.hip): not algorithmic
        self.f32_bytes = []     # the same for every fp32 entry of `log` (parallel list): bench.py's roofline.algorithmic_bytes_per_launch
        self.geoms = []         # full geometry per conv, for the tuner
        # split-fp16 launches of this plan report a not-finite sum (an activation beyond the fp16 window) in ONE sticky device word
        # (usot_conv_desc.ovf / usot_pw_pair_desc.ovf); None until the plan has such a launch.  Readers: Session (through the decode
        # kernel's out[9]) and Engine.track / features (after the replay), which re-plan on the exact-fp32 tiles and run again.
        self.ovf = None
        self.last_ws = None     # the split-K workspace of the descriptor conv_desc() built last (conv_deferred / conv_batch take their slabs from it)
        # riders (Session 'defer_append' = 2): up to three independent conv_batch items that the backbone's first stand-alone shortcut
        # conv takes into ITS launch; their outputs land in `piggy_out` (None = nobody took them: the caller launches them itself)
        self.piggyback = None
        self.piggy_out = None

    def buf(self, *shape, dtype=torch.float32):
        ch = getattr(self, '_chain', None)
        if ch is not None:
            # chain mode (backbone_bf16, 'lp_chains'): this builder pass works on slice `idx` of `nch` equal batch slices; the
            # k-th buffer every chain asks for is ONE full-batch tensor (whichever chain gets there first allocates it)
            idx, nch, store, cursor = ch
            k = cursor[idx]
            cursor[idx] += 1
            if k == len(store):
                store.append(torch.empty((shape[0] * nch,) + tuple(shape[1:]), device=self.dev, dtype=dtype))
                self.plan.keep.append(store[k])
            full = store[k]
            assert full.shape[0] == shape[0] * nch and tuple(full.shape[1:]) == tuple(shape[1:]) and full.dtype == dtype, (full.shape, shape)
            return full[idx * shape[0]:(idx + 1) * shape[0]]
        t = torch.empty(shape, device=self.dev, dtype=dtype)
        self.plan.keep.append(t)
        return t

    # lane levels: 1 = memory-kernel side beside the backbone; 2 = + memory head beside the
    # reg/cls heads; 3 = + shortcut convs, the three search encoders, the prediction convs.
    # Every fork/join is a cross-queue dependency in the captured graph (several us each on
    # this stack), so fine-grained lanes cost more than they hide — measured, see DESIGN.md.
    def fork(self, lane, level=3):
        if self.lanes >= level:
            self.plan.fork(lane)

    def join(self, lane, level=3):
        if self.lanes >= level:
            self.plan.join(lane)

    def ovf_word(self):
        if self.ovf is None:
            self.ovf = torch.zeros(1, device=self.dev, dtype=torch.int32)
            self.plan.keep.append(self.ovf)
        return self.ovf

    def conv_desc(self, name, pc, x, n, h, w, *, cout=None, act=ACT_NONE, res=None, y=None, y_cstride=0, y_coff=0,
                  act2=ACT_NONE, act_split=0, y_nchw=False, groups=1, x_gs=0, y_gs=0, w_rows=None, row0=0, tile=None,
                  force_ks=None):
        """Descriptor of one convolution (nothing is added to the plan yet).
        x: tensor (NHWC dense, channels == pc.cin).  Returns (desc, y, oh, ow, log, geom)."""
        cout = cout or pc.cout
        oh, ow = pc.out_hw(h, w)
        if y is None:
            y = self.buf(groups, n, cout, oh, ow) if y_nchw else self.buf(groups, n, oh, ow, cout)
            if groups == 1:
                y = y[0]
        k = pc.kh * pc.kw * pc.cin
        m = n * oh * ow
        ttile, ksplit = self.tuning.get((m, cout, k, groups), (0, 1))
        if tile is not None:                       # batched launch: tile imposed by the lead problem
            ksplit = 1 if tile != ttile else ksplit
            ttile = tile
        if force_ks is not None:
            ksplit = force_ks
        ws = None
        # 'split16_f32': the producer / consumer tiles of the long reductions run on their split-fp16 twins (csrc/conv_igemm.hip, PF = 4)
        if (self.opt['split16_f32'] and k >= self.opt['split16_min_k'] and m >= self.opt['split16_min_m'] and pc.cin % 64 == 0
                and ttile in SPLIT16_TILES):
            ttile = SPLIT16_TILES[ttile]
        frag = hip.tile_wfrag(ttile)                # 1: filters in MFMA fragment order; 2: split-fp16 bank + row scales
        if frag == 2 and pc.cin % 64:
            raise hip.HipError('%s: split-fp16 tile %d needs Cin %% 64 == 0 (Cin = %d)' % (name, ttile, pc.cin))
        if frag == 1 and (row0 % 16 or ((w_rows or cout) % 16 and groups > 1) or not hip.tile_supports(ttile, pc.cin, cout, k)):
            ttile, ksplit, frag = (self.default_batch_tile if tile is not None else 0), 1, 0     # geometry the tile cannot take
        if (frag == 1 and hip.tile_kreq(ttile)[0]) or hip.tile_streamk(ttile):
            ksplit = 1                              # weight-stationary tiles split k inside the workgroup, stream-K tiles over the resident set
        if ksplit > 1:            # partial slabs + one ticket per tile (usot_conv_ws_floats), tickets zero before first use
            ws = self.buf(ksplit * groups * m * cout + groups * ((m + 15) // 16) * ((cout + 31) // 32))
            ws.zero_()
        wsc = None
        if frag == 2:
            wbank, wsc = pc.w_split16()
        else:
            wbank = pc.w_frag() if frag else pc.w
        d = hip.conv_desc(x.data_ptr(), wbank.data_ptr() + row0 * k * 4, pc.b.data_ptr() + row0 * 4, y.data_ptr(),
                          N=n, H=h, W=w, Cin=pc.cin, OH=oh, OW=ow, Cout=cout, KH=pc.kh, KW=pc.kw,
                          stride=pc.stride, pad=pc.pad, dil=pc.dil,
                          res=res.data_ptr() if res is not None else None, act=act, act2=act2,
                          act_split=act_split, y_cstride=y_cstride, y_coff=y_coff, y_nchw=int(y_nchw),
                          groups=groups, x_gs=x_gs, w_gs=(w_rows or cout) * k, b_gs=(w_rows or cout),
                          y_gs=y_gs if y_gs else n * oh * ow * cout, r_gs=0,
                          ksplit=ksplit, tile=ttile, ws=ws.data_ptr() if ws is not None else None, w_frag=frag,
                          w_scale=wsc.data_ptr() + row0 * 4 if wsc is not None else None,
                          ovf=self.ovf_word().data_ptr() if frag == 2 else None)
        self.last_ws = ws
        self.plan.keep += [x, wbank, pc.b, wsc]
        log = (name, m, cout, k, groups, m * cout * k * groups)
        geom = dict(name=name, N=n, H=h, W=w, Cin=pc.cin, Cout=cout, KH=pc.kh, KW=pc.kw, stride=pc.stride,
                    pad=list(pc.pad), dil=list(pc.dil), groups=groups, has_res=res is not None)
        return d, y, oh, ow, log, geom

    def conv(self, name, pc, x, n, h, w, **kw):
        """One convolution = one launch.  Returns (y, oh, ow)."""
        d, y, oh, ow, log, geom = self.conv_desc(name, pc, x, n, h, w, **kw)
        if hip.tile_streamk(d.tile):                # persistent stream-K tile: slabs + tickets for the shares that end inside a tile
            self.plan.keep.append(hip.streamk_ws([d], d.tile, self.dev))
        hip.check(hip.lib().usot_plan_add_conv(self.plan.h, C.byref(d)), 'plan_add_conv ' + name)
        self.log.append(log)
        self.f32_bytes.append(self._conv_bytes(d))
        self.geoms.append(geom)
        return y, oh, ow

    def conv_deferred(self, name, pc, x, n, h, w, tile, ks):
        """A convolution whose split-K reduction is DEFERRED to its consumer (usot_conv_desc.defer): one launch writes the `ks`
        partial tiles, no bias, no activation, no combine.  Returns (slabs [ks, m, cout], oh, ow); the consumer sums them, adds
        pc.b and activates while it stages its input (pw_pair_f32(t2_parts=ks, t2_bias=pc.b))."""
        if ks < 2:
            raise hip.HipError('conv_deferred %s: a deferred reduction needs ksplit >= 2 (got %d)' % (name, ks))
        d, _, oh, ow, log, geom = self.conv_desc(name, pc, x, n, h, w, tile=tile, force_ks=ks, y=self._unwritten_y())
        slabs = self.last_ws
        if d.ksplit != ks or d.w_frag == 1 or slabs is None:
            raise hip.HipError('conv_deferred %s: tile %d cannot split k %d ways' % (name, tile, ks))
        m = n * oh * ow
        d.defer, d.act, d.bias, d.res = 1, ACT_NONE, None, None
        hip.check(hip.lib().usot_plan_add_conv(self.plan.h, C.byref(d)), 'plan_add_conv(deferred) ' + name)
        self.log.append(log)
        self.f32_bytes.append(4 * (n * h * w * pc.cin + pc.cout * pc.kh * pc.kw * pc.cin + ks * m * pc.cout))
        self.geoms.append(geom)
        return slabs[:ks * m * pc.cout].view(ks, m, pc.cout), oh, ow

    def _unwritten_y(self):
        """The `y` of a deferred launch: the descriptor wants a valid pointer, the launch never writes it (its partial tiles go to
        `ws`).  A 16-byte scratch of the plan rather than an alias of the input map: a tile that ignored `defer` would then
        clobber scratch, not the activations."""
        if getattr(self, '_y_scratch', None) is None:
            self._y_scratch = torch.zeros(4, device=self.dev)
            self.plan.keep.append(self._y_scratch)
        return self._y_scratch

    @staticmethod
    def _conv_bytes(d):
        """Algorithmic HBM bytes of one fp32 convolution: input map, filter bank, bias, result (+ residual), once each."""
        g = max(1, d.groups)
        k = d.KH * d.KW * d.Cin
        m = d.N * d.OH * d.OW
        return 4 * g * (d.N * d.H * d.W * d.Cin + d.Cout * k + d.Cout + m * d.Cout * (2 if d.res else 1))

    def conv_batch(self, items, lead_tile=None):
        """Several convolutions of different geometry in ONE launch (usot_conv2d_batch_f32).
        items: [(name, pc, x, n, h, w, kwargs)], the first one leads (its tuned tile is used unless `lead_tile` names one).
        Falls back to separate launches when batching is disabled.  Returns [(y, oh, ow)]."""
        if not self.batch or len(items) == 1:
            return [self.conv_deferred(nm, pc, x, n, h, w, tile=lead_tile or self.default_batch_tile, ks=kw['defer_ks']) if kw.get('defer_ks')
                    else self.conv(nm, pc, x, n, h, w, **dict(kw, **({'tile': lead_tile} if lead_tile else {})))
                    for nm, pc, x, n, h, w, kw in items]
        descs, outs, macs = [], [], 0
        for nm, pc, x, n, h, w, kw in items:
            kw = dict(kw)
            dks = kw.pop('defer_ks', None)          # this problem's split-K reduction is deferred to its consumer (conv_deferred)
            if dks:
                if dks < 2:
                    raise hip.HipError('conv_batch %s: a deferred reduction needs ksplit >= 2 (got %d)' % (nm, dks))
                kw.update(force_ks=dks, y=self._unwritten_y())
                kw.pop('act', None)
            d, y, oh, ow, log, geom = self.conv_desc(nm, pc, x, n, h, w, tile=lead_tile, **kw)
            if lead_tile is None:
                lead_tile = d.tile if d.tile else self.default_batch_tile
                d.tile = lead_tile
            if dks:
                slabs = self.last_ws
                if d.ksplit != dks or d.w_frag == 1 or slabs is None:
                    raise hip.HipError('conv_batch %s: tile %d cannot split k %d ways' % (nm, d.tile, dks))
                d.defer, d.act, d.bias, d.res = 1, ACT_NONE, None, None
                y = slabs[:dks * n * oh * ow * pc.cout].view(dks, n * oh * ow, pc.cout)
            descs.append(d)
            outs.append((y, oh, ow))
            macs += log[5]
            self.geoms.append(geom)
        if hip.tile_streamk(lead_tile):
            self.plan.keep.append(hip.streamk_ws(descs, lead_tile, self.dev))
        arr = (hip.ConvDesc * len(descs))(*descs)
        hip.check(hip.lib().usot_plan_add_conv_batch(self.plan.h, arr, len(descs)), 'plan_add_conv_batch')
        first = items[0]
        self.log.append(('+'.join(i[0] for i in items), descs[0].N * descs[0].OH * descs[0].OW, descs[0].Cout,
                         descs[0].KH * descs[0].KW * descs[0].Cin, len(descs), macs))
        # siblings that share an input map (shortcut conv + conv1, the three search encoders) read it once
        shared = len({dd.x for dd in descs}) == 1
        self.f32_bytes.append(sum(self._conv_bytes(dd) for dd in descs)
                              - (4 * (len(descs) - 1) * descs[0].N * descs[0].H * descs[0].W * descs[0].Cin if shared else 0))
        return outs

    def thin_convs(self, items):
        """The prediction convs (1-4 output channels, 3x3/p1, NCHW output) in one launch of the thin
        kernel (usot_thin_conv3x3_f32) instead of 32-wide MFMA tiles.  items as for conv_batch."""
        descs, outs, macs = [], [], 0
        for nm, pc, x, n, h, w, kw in items:
            d, y, oh, ow, log, geom = self.conv_desc(nm, pc, x, n, h, w, tile=7, **kw)
            d.ksplit, d.ws = 1, None
            descs.append(d)
            outs.append((y, oh, ow))
            macs += log[5]
        arr = (hip.ConvDesc * len(descs))(*descs)
        hip.check(hip.lib().usot_plan_add_thin_conv(self.plan.h, arr, len(descs)), 'plan_add_thin_conv')
        return outs

    # ---- a1-a4: backbone + neck: x NCHW [n,3,s,s] -> xf NHWC [n,hf,wf,256]
    def backbone(self, x, n, size, need_stem=True, xptr_dev=None):
        """need_stem=False (frame plans): stem + max-pool in one MFMA kernel, the stem map is never stored;
        feature_extractor() (modules.py:137-151 returns it as x_) keeps the two-kernel form."""
        W, L = self.W, hip.lib()
        oh = (size - 7) // 2 + 1
        ph = (oh - 1) // 2 + 1
        p0 = self.buf(n, ph, ph, 64)
        s0 = None
        if need_stem:
            s0 = self.buf(n, oh, oh, 64)
            hip.check(L.usot_plan_add_stem_mu(self.plan.h, hip.ptr(x), hip.ptr(W.stem_w), hip.ptr(W.stem_b_mu), hip.ptr(s0),
                                              n, size, size, oh, oh, *W.STEM_MU), 'plan_add_stem')
            hip.check(L.usot_plan_add_maxpool(self.plan.h, hip.ptr(s0), hip.ptr(p0), n, oh, oh, 64, ph, ph), 'plan_add_maxpool')
        else:
            wf = W.stem_f32()
            # xptr_dev (device int32[2], optional): the crop's address is read from device memory at run time
            hip.check(L.usot_plan_add_stem_pool_mu(self.plan.h, hip.ptr(x), hip.ptr(wf), hip.ptr(W.stem_b_mu), hip.ptr(p0),
                                                   n, size, size, oh, oh, ph, ph,
                                                   C.c_void_p(xptr_dev.data_ptr()) if xptr_dev is not None else None,
                                                   *W.STEM_MU), 'plan_add_stem_pool')
            self.plan.keep += [wf, W.stem_b_mu]
        self.plan.keep += [x]
        cur, h = p0, ph
        stages = [s0]
        t1_fused = None                               # conv1 output of this block when the previous conv3 produced it
        O = self.opt

        def pair_path(bi, c2, c3, h):
            """How the tail of block bi is lowered: (nxt, fuse, triple, dfr) - the next 1x1 conv, whether conv3 | next-conv1 fuse into one
            launch, whether conv2 joins that launch too, and the (tile, ksplit) of a conv2 whose reduction is deferred to the pair."""
            h2 = c2.out_hw(h, h)[0]
            last = bi + 1 == len(W.blocks)
            nxt = W.neck if last else W.blocks[bi + 1][0]
            shape = (c3.cin, c3.cout, nxt.cout)
            m2 = n * h2 * h2
            fuse = self.lanes == 0 and nxt.kh == 1 and (
                (shape in O['fused_pointwise_f32'] and m2 <= O['fused_pointwise_f32_max_m']) or
                (O['fused_f32_sliced'] and shape in O['fused_pointwise_f32_sliced_only']
                 and hip.lib().usot_pw_pair_f32_ws_floats(m2, *shape) > 0))
            triple = bool(fuse and O['fused_triple_f32'] and c2.kh == 3 and c2.stride == 1 and m2 <= O['fused_pointwise_f32_max_m']
                          and (c2.cin, c3.cin, c3.cout, nxt.cout) in O['fused_triple_f32_shapes']
                          and hip.lib().usot_pw_triple_f32_supported(c2.cin, c3.cin, c3.cout, nxt.cout))
            dk2_ = (m2, c2.cout, c2.kh * c2.kw * c2.cin)
            dfr = O['defer_split_f32'].get(dk2_) if (fuse and not triple and c2.kh == 3) else None
            if dfr is not None and O['split16_f32'] and dk2_ in O['defer_split_s16']:
                dfr = O['defer_split_s16'][dk2_]         # the (tile, ksplit) tuned on the split-fp16 tiles
            return nxt, fuse, triple, dfr

        for bi, (c1, c2, c3, ds) in enumerate(W.blocks):
            sc = cur
            pre_t2 = None                             # conv2's deferred partial tiles when it rode in the shortcut conv's launch
            sc_parts, sc_bias = 0, None               # the shortcut conv's deferred partial tiles (summed by the pair's residual read)
            if t1_fused is not None:
                t1, t1_fused = t1_fused, None
                if ds is not None:
                    h2_ = c2.out_hw(h, h)[0]
                    dk_ = (n * h2_ * h2_, c2.cout, c2.kh * c2.kw * c2.cin)
                    will_defer = pair_path(bi, c2, c3, h)[3] is not None       # this block's pair sums deferred partial tiles
                    bd = self.opt['batch_ds_conv2'].get(dk_) if self.lanes == 0 and self.batch else None
                    if bd and will_defer:
                        # the shortcut conv and conv2 are independent (block input / conv1's map): ONE launch, conv2's reduction deferred
                        (sc, _, _), (pre_t2, _, _) = self.conv_batch([('b%d.ds' % bi, ds, cur, n, h, h, {}),
                                                                      ('b%d.conv2' % bi, c2, t1, n, h, h, dict(defer_ks=bd[1]))], lead_tile=bd[0])
                    else:
                        rk_ = (n * h2_ * h2_, ds.cout, ds.kh * ds.kw * ds.cin)
                        rd = self.opt['defer_split_res_f32'].get(rk_) if will_defer else None
                        if rd:
                            # the shortcut conv's reduction rides in the pair's residual read (consumed ONLY there: the deferred-conv2 path below)
                            sc, _, _ = self.conv_deferred('b%d.ds' % bi, ds, cur, n, h, h, tile=rd[0], ks=rd[1])
                            sc_parts, sc_bias = rd[1], ds.b
                        elif self.piggyback and self.batch and len(self.piggyback) <= 3:
                            # Session 'defer_append' = 2: the previous frame's memory-feature encoders ride in this launch (independent
                            # problems on the same tile; the shortcut conv's 248 workgroups leave CUs free)
                            pg, self.piggyback = self.piggyback, None
                            res = self.conv_batch([('b%d.ds' % bi, ds, cur, n, h, h, {})] + pg)
                            sc = res[0][0]
                            self.piggy_out = [r[0] for r in res[1:]]
                        else:
                            sc, _, _ = self.conv('b%d.ds' % bi, ds, cur, n, h, h)
            elif ds is not None and self.lanes < 3:   # shortcut conv shares conv1's launch
                (sc, _, _), (t1, _, _) = self.conv_batch([('b%d.ds' % bi, ds, cur, n, h, h, {}),
                                                          ('b%d.conv1' % bi, c1, cur, n, h, h, dict(act=ACT_RELU))])
            else:
                if ds is not None:                # shortcut conv on its own lane beside conv1 -> conv2
                    self.fork(1)
                    sc, _, _ = self.conv('b%d.ds' % bi, ds, cur, n, h, h)
                    self.fork(0)
                t1, _, _ = self.conv1x1('b%d.conv1' % bi, c1, cur, n, h, act=ACT_RELU)
            h2 = c2.out_hw(h, h)[0]
            last = bi + 1 == len(W.blocks)
            m2 = n * h2 * h2
            nxt, fuse, triple, dfr = pair_path(bi, c2, c3, h)
            nm = 'b%d.conv3+%s' % (bi, 'neck' if last else 'b%d.conv1' % (bi + 1))
            act2 = ACT_NONE if last else ACT_RELU
            if triple:
                cur, t1_fused = self.pw_triple_f32('b%d.conv2+' % bi + nm, c2, c3, nxt, t1, sc, n, h, act2=act2)
            else:
                if dfr:
                    # conv2's split-K reduction rides in the pair's tile staging (DEFAULT_OPTIONS: defer_split_f32)
                    if pre_t2 is not None:
                        t2, parts = pre_t2, int(pre_t2.shape[0])
                    else:
                        t2, _, _ = self.conv_deferred('b%d.conv2' % bi, c2, t1, n, h, h, tile=dfr[0], ks=dfr[1])
                        parts = dfr[1]
                    cur, t1_fused = self.pw_pair_f32(nm, c3, nxt, t2, sc, n, h2, act2=act2, t2_parts=parts, t2_bias=c2.b,
                                                     res_parts=sc_parts, res_bias=sc_bias)
                    h = h2
                    if bi in (2, 6, 12):
                        stages.append(cur)
                    continue
                t2, _, _ = self.conv3x3('b%d.conv2' % bi, c2, t1, n, h, h, act=ACT_RELU)
                if ds is not None and self.lanes >= 3:
                    self.join(1)
                if fuse:
                    cur, t1_fused = self.pw_pair_f32(nm, c3, nxt, t2, sc, n, h2, act2=act2)
                else:
                    cur, _, _ = self.conv1x1('b%d.conv3' % bi, c3, t2, n, h2, act=ACT_RELU, res=sc)
            h = h2
            if bi in (2, 6, 12):                      # ends of layer1 / layer2 / layer3
                stages.append(cur)
        if t1_fused is not None:                       # the neck rode in layer3's last conv3 launch
            xf, t1_fused = t1_fused, None
        else:
            xf, _, _ = self.conv1x1('neck', W.neck, cur, n, h)
        self.stages = stages
        return xf, h

    # ---- bf16 backbone for the batched MFMA-roofline configuration (BASELINE config 3)
    def conv_bf16(self, name, pc, x, n, h, w, *, act=ACT_NONE, res=None, tile=0, dtype=torch.bfloat16, out_f32=False,
                  act2=ACT_NONE, act_split=0, groups=1, cout=None, x_gs=0, y=None):
        """Low-precision conv.  groups > 1: `groups` problems of identical geometry, inputs x_gs elements
        apart, filter rows [g*cout, (g+1)*cout) of the packed bank, outputs stacked (the three towers)."""
        oh, ow = pc.out_hw(h, w)
        cout = cout or pc.cout
        if y is None:
            y = self.buf(groups, n, oh, ow, cout, dtype=torch.float32 if out_f32 else dtype) if groups > 1 else \
                self.buf(n, oh, ow, cout, dtype=torch.float32 if out_f32 else dtype)
        wb = pc.w_lp(dtype)
        k = pc.kh * pc.kw * pc.cin
        rs = getattr(self, '_nscale', 1)      # chain mode: kernels are chosen as for the whole batch
        if (tile == 0 and pc.kh == 3 and pc.kw == 3 and pc.stride == 1 and tuple(pc.pad) == (1, 1) and tuple(pc.dil) == (1, 1)
                and groups == 1 and not out_f32 and res is None and cout == pc.cout and act in (ACT_NONE, ACT_RELU)
                and act_split == 0 and (pc.cin, cout) in self.opt['halo_3x3_lp']
                and hip.lib().usot_conv3x3_halo_supported(pc.cin, cout)
                and rs * n * ((h + 15) // 16) * ((w + 15) // 16) >= 256):
            hip.check(hip.lib().usot_plan_add_conv3x3_halo(self.plan.h, hip.ptr(x), hip.ptr(wb), hip.ptr(pc.b), hip.ptr(y), n, h, w,
                                                           pc.cin, cout, act, 1 if dtype == torch.float16 else 0),
                      'plan_add_conv3x3_halo ' + name)
            self.plan.keep += [x, wb, pc.b]
            self.log.append((name, n * oh * ow, cout, k, 1, n * oh * ow * cout * k))
            self.lp_bytes.append(2 * (n * h * w * pc.cin + n * oh * ow * cout + cout * k))
            return y, oh, ow
        if (tile == 0 and pc.kh == 1 and pc.kw == 1 and pc.stride == 1 and groups == 1 and not out_f32 and cout == pc.cout
                and act in (ACT_NONE, ACT_RELU) and act_split == 0 and (k, cout) in self.opt['panel_1x1_lp']
                and hip.lib().usot_pw_panel_supported(k, cout)
                and rs * n * oh * ow >= self.opt['panel_min_panels'] * hip.lib().usot_pw_panel_min_pixels(k, cout)):
            hip.check(hip.lib().usot_plan_add_pw_panel(self.plan.h, hip.ptr(x), hip.ptr(wb), hip.ptr(pc.b),
                                                       hip.ptr(res) if res is not None else None, hip.ptr(y), n * oh * ow, k, cout,
                                                       act, 1 if dtype == torch.float16 else 0), 'plan_add_pw_panel ' + name)
            self.plan.keep += [x, wb, pc.b, res]
            self.log.append((name, n * oh * ow, cout, k, 1, n * oh * ow * cout * k))
            self.lp_bytes.append(2 * (n * h * w * pc.cin + n * oh * ow * cout * (2 if res is not None else 1) + cout * k))
            return y, oh, ow
        if (tile == 0 and pc.kh == 3 and pc.kw == 3 and groups == 1 and not out_f32 and cout == pc.cout and res is None
                and act in (ACT_NONE, ACT_RELU) and act_split == 0 and (pc.cin, cout) in self.opt['kstream_3x3_lp']
                and pc.pad[0] == pc.pad[1] and pc.dil[0] == pc.dil[1] and pc.pad[0] <= pc.dil[0] and pc.stride in (1, 2)
                and hip.lib().usot_conv_kstream_supported(pc.cin, cout, 3, 3) and rs * n * oh * ow >= self.opt['panel_min_panels'] * 256):
            hip.check(hip.lib().usot_plan_add_conv_kstream(self.plan.h, hip.ptr(x), hip.ptr(wb), hip.ptr(pc.b), hip.ptr(y), n, h, w,
                                                           pc.cin, cout, pc.stride, pc.pad[0], pc.dil[0], act,
                                                           1 if dtype == torch.float16 else 0), 'plan_add_conv_kstream ' + name)
            self.plan.keep += [x, wb, pc.b]
            self.log.append((name, n * oh * ow, cout, k, 1, n * oh * ow * cout * k))
            self.lp_bytes.append(2 * (n * h * w * pc.cin + n * oh * ow * cout + cout * k))
            return y, oh, ow
        if (tile == 0 and pc.kh == 1 and pc.kw == 1 and pc.stride == 1 and groups == 1 and not out_f32 and cout == pc.cout
                and res is None and act in (ACT_NONE, ACT_RELU) and act_split == 0 and (k, cout) in self.opt['kstream_1x1_lp']
                and hip.lib().usot_pw_kstream_supported(k, cout) and rs * n * oh * ow >= self.opt['panel_min_panels'] * 256):
            hip.check(hip.lib().usot_plan_add_pw_kstream(self.plan.h, hip.ptr(x), hip.ptr(wb), hip.ptr(pc.b), hip.ptr(y), n * oh * ow,
                                                         k, cout, act, 1 if dtype == torch.float16 else 0), 'plan_add_pw_kstream ' + name)
            self.plan.keep += [x, wb, pc.b]
            self.log.append((name, n * oh * ow, cout, k, 1, n * oh * ow * cout * k))
            self.lp_bytes.append(2 * (n * h * w * pc.cin + n * oh * ow * cout + cout * k))
            return y, oh, ow
        if tile == 0:
            tile = LP_TUNING.get((rs * n * oh * ow, cout, k), 0)
        if hasattr(self, 'lp_geoms'):       # scripts/tune_lp.py collects the shapes this way
            self.lp_geoms.append(dict(name=name, N=n, H=h, W=w, Cin=pc.cin, OH=oh, OW=ow, Cout=cout, KH=pc.kh, KW=pc.kw,
                                      stride=pc.stride, pad=pc.pad, dil=pc.dil, has_res=res is not None,
                                      M=n * oh * ow, K=k))
        d = hip.conv_desc(x.data_ptr(), wb.data_ptr(), pc.b.data_ptr(), y.data_ptr(), N=n, H=h, W=w, Cin=pc.cin,
                          OH=oh, OW=ow, Cout=cout, KH=pc.kh, KW=pc.kw, stride=pc.stride, pad=pc.pad, dil=pc.dil,
                          res=res.data_ptr() if res is not None else None, act=act, act2=act2, act_split=act_split,
                          tile=tile, groups=groups, x_gs=x_gs, w_gs=cout * k, b_gs=cout, y_gs=n * oh * ow * cout)
        hip.check(hip.lib().usot_plan_add_conv_lp(self.plan.h, C.byref(d), 1 if dtype == torch.float16 else 0, int(out_f32)),
                  'plan_add_conv_lp ' + name)
        self.plan.keep += [x, wb, pc.b]
        self.log.append((name, n * oh * ow, cout, k, groups, groups * n * oh * ow * cout * k))
        self.lp_bytes.append(groups * (2 * (n * h * w * pc.cin + cout * k) + n * oh * ow * cout * ((4 if out_f32 else 2) + (2 if res is not None else 0))))
        return y, oh, ow

    def pw_pair(self, name, c3, nxt, t2, res, n, h, act2, dtype):
        """conv3 + residual + ReLU of one bottleneck and the next 1x1 conv in ONE launch (csrc/pw_pair.hip).
        Returns (y [n,h,h,c3.cout], t [n,h,h,nxt.cout])."""
        m = n * h * h
        y = self.buf(n, h, h, c3.cout, dtype=dtype)
        t = self.buf(n, h, h, nxt.cout, dtype=dtype)
        w3p = c3.w_lp_pw_pair(dtype, c3.cin, c3.cout, nxt.cout, 0)
        w1 = nxt.w_lp_pw_pair(dtype, c3.cin, c3.cout, nxt.cout, 1)
        d = hip.pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), c3.b.data_ptr(), res.data_ptr(), y.data_ptr(), w1.data_ptr(),
                             nxt.b.data_ptr(), t.data_ptr(), m, c3.cin, c3.cout, nxt.cout, act2)
        hip.check(hip.lib().usot_plan_add_pw_pair(self.plan.h, C.byref(d), 1 if dtype == torch.float16 else 0),
                  'plan_add_pw_pair ' + name)
        self.plan.keep += [t2, res, w3p, w1, c3.b, nxt.b]
        self.log.append((name, m, c3.cout, c3.cin, 1, m * (c3.cout * c3.cin + nxt.cout * c3.cout)))
        self.lp_bytes.append(2 * (m * (c3.cin + 2 * c3.cout + nxt.cout) + c3.cout * c3.cin + nxt.cout * c3.cout))
        return y, t

    def pw_panel_pair(self, name, c3, nxt, t2, res, n, h, act2, dtype):
        """The same pair on the pixel-stationary kernel (csrc/pw_panel.hip: pair form; natural-layout filter banks)."""
        m = n * h * h
        y = self.buf(n, h, h, c3.cout, dtype=dtype)
        t = self.buf(n, h, h, nxt.cout, dtype=dtype)
        w3, w1 = c3.w_lp(dtype), nxt.w_lp(dtype)
        d = hip.pw_pair_desc(t2.data_ptr(), w3.data_ptr(), c3.b.data_ptr(), res.data_ptr(), y.data_ptr(), w1.data_ptr(),
                             nxt.b.data_ptr(), t.data_ptr(), m, c3.cin, c3.cout, nxt.cout, act2)
        hip.check(hip.lib().usot_plan_add_pw_panel_pair(self.plan.h, C.byref(d), 1 if dtype == torch.float16 else 0),
                  'plan_add_pw_panel_pair ' + name)
        self.plan.keep += [t2, res, w3, w1, c3.b, nxt.b]
        self.log.append((name, m, c3.cout, c3.cin, 1, m * (c3.cout * c3.cin + nxt.cout * c3.cout)))
        self.lp_bytes.append(2 * (m * (c3.cin + 2 * c3.cout + nxt.cout) + c3.cout * c3.cin + nxt.cout * c3.cout))
        return y, t

    def conv_pw(self, name, c2, c3, t1, res, n, h, dtype):
        """A layer3 bottleneck's conv2 (3x3) + BN + ReLU -> conv3 (1x1) + BN + residual + ReLU in ONE launch (csrc/conv_pw_lp.hip):
        conv2's output panel never leaves LDS.  Returns y [n,oh,ow,c3.cout] and the output size."""
        oh, ow = c2.out_hw(h, h)
        m = n * oh * ow
        y = self.buf(n, oh, ow, c3.cout, dtype=dtype)
        w2, w3 = c2.w_lp(dtype), c3.w_lp(dtype)
        d = hip.conv_desc(t1.data_ptr(), w2.data_ptr(), c2.b.data_ptr(), None, N=n, H=h, W=h, Cin=c2.cin, OH=oh, OW=ow, Cout=c2.cout,
                          KH=c2.kh, KW=c2.kw, stride=c2.stride, pad=c2.pad, dil=c2.dil, act=ACT_RELU,
                          tile=(0 if self.opt['conv_pw_rs'] else 4) | int(self.opt['conv_pw_panel']))
        hip.check(hip.lib().usot_plan_add_conv_pw(self.plan.h, C.byref(d), hip.ptr(w3), hip.ptr(c3.b), hip.ptr(res), hip.ptr(y),
                                                  1 if dtype == torch.float16 else 0), 'plan_add_conv_pw ' + name)
        self.plan.keep += [t1, res, w2, w3, c2.b, c3.b]
        k2 = c2.kh * c2.kw * c2.cin
        self.log.append((name, m, c3.cout, c3.cin, 1, m * (c2.cout * k2 + c3.cout * c3.cin)))
        self.lp_bytes.append(2 * (n * h * h * c2.cin + 2 * m * c3.cout + c2.cout * k2 + c3.cout * c3.cin))
        return y, oh

    def conv_pw_pair(self, name, c2, c3, nxt, t1, res, n, h, act2, dtype):
        """A whole bottleneck tail of layer2 in ONE launch (csrc/conv_pw_lp.hip, pair form): conv2 (3x3) + BN + ReLU -> conv3 + BN +
        residual + ReLU -> the next block's conv1 + BN + ReLU.  Returns (y, t, oh)."""
        oh, ow = c2.out_hw(h, h)
        m = n * oh * ow
        y = self.buf(n, oh, ow, c3.cout, dtype=dtype)
        t = self.buf(n, oh, ow, nxt.cout, dtype=dtype)
        w2, w3, w1 = c2.w_lp(dtype), c3.w_lp(dtype), nxt.w_lp(dtype)
        d2 = hip.conv_desc(t1.data_ptr(), w2.data_ptr(), c2.b.data_ptr(), None, N=n, H=h, W=h, Cin=c2.cin, OH=oh, OW=ow, Cout=c2.cout,
                           KH=c2.kh, KW=c2.kw, stride=c2.stride, pad=c2.pad, dil=c2.dil, act=ACT_RELU,
                           tile=(0 if self.opt['conv_pw_rs'] else 4) | int(self.opt['conv_pw_panel']))
        d = hip.pw_pair_desc(None, w3.data_ptr(), c3.b.data_ptr(), res.data_ptr(), y.data_ptr(), w1.data_ptr(), nxt.b.data_ptr(),
                             t.data_ptr(), m, c3.cin, c3.cout, nxt.cout, act2)
        L = hip.lib()
        # 'conv_pw_ov_lp': layer3's blocks on the OVERLAPPED form (csrc/conv_pw_ov.hip: matrix-pipe and HBM workgroups side by side on
        # every CU) when the launch has at least conv_pw_ov_min_panels panels of 128 pixels and the geometry is the one it is built for
        ov = (self.opt['conv_pw_ov_lp'] and self.opt['conv_pw_rs'] and L.usot_conv_pw_ov_supported(c3.cin, c3.cout, nxt.cout)
              and c2.kh == 3 and c2.kw == 3 and c2.stride == 1 and tuple(c2.pad) == tuple(c2.dil) and 1 <= c2.dil[1] <= 4
              and (oh, ow) == (h, h) and (m + 127) // 128 >= self.opt['conv_pw_ov_min_panels'] and m * c3.cout * 2 < 2 ** 31)
        if ov:
            ws = torch.zeros(int(L.usot_conv_pw_ov_ws_bytes(m)) // 4, dtype=torch.int32, device=self.dev)     # flags zero; T2 hand-off map
            hip.check(L.usot_plan_add_conv_pw_ov(self.plan.h, C.byref(d2), C.byref(d), 1 if dtype == torch.float16 else 0, hip.ptr(ws)),
                      'plan_add_conv_pw_ov ' + name)
            self.plan.keep.append(ws)
            self.ov_ws = getattr(self, 'ov_ws', []) + [ws]
        else:
            hip.check(L.usot_plan_add_conv_pw_pair(self.plan.h, C.byref(d2), C.byref(d), 1 if dtype == torch.float16 else 0),
                      'plan_add_conv_pw_pair ' + name)
        self.plan.keep += [t1, res, w2, w3, w1, c2.b, c3.b, nxt.b]
        k2 = c2.kh * c2.kw * c2.cin
        self.log.append((name, m, c3.cout, c3.cin, 1, m * (c2.cout * k2 + c3.cout * c3.cin + nxt.cout * c3.cout)))
        self.lp_bytes.append(2 * (n * h * h * c2.cin + m * (2 * c3.cout + nxt.cout) + c2.cout * k2 + c3.cout * c3.cin + nxt.cout * c3.cout))
        if nxt.cout == 256:                             # phase-5 form: the Y panel is read back for the second convolution
            self.lp_readback.append(2 * m * c3.cout)
        return y, t, oh

    def bneck_first(self, name, c1, c2, c3, ds, nxt, x, n, h, dtype):
        """Layer1's first bottleneck + the next block's conv1 in ONE launch (csrc/bneck_lp.hip).  Returns (y [n,h,h,256],
        t [n,h,h,64]): the block's output and the next conv1's (both after ReLU)."""
        y = self.buf(n, h, h, c3.cout, dtype=dtype)
        t = self.buf(n, h, h, nxt.cout, dtype=dtype)
        cache = c3.__dict__.setdefault('_w3c', {})
        if dtype not in cache:                      # [conv3 | downsample] along k; one bias
            cache[dtype] = (torch.cat([c3.w, ds.w], 1).to(dtype).contiguous(), (c3.b + ds.b).contiguous())
        w3c, b3c = cache[dtype]
        w1, w2, wn = c1.w_lp(dtype), c2.w_lp(dtype), nxt.w_lp(dtype)
        d = hip.bneck_desc(x.data_ptr(), w1.data_ptr(), c1.b.data_ptr(), w2.data_ptr(), c2.b.data_ptr(), w3c.data_ptr(),
                           b3c.data_ptr(), wn.data_ptr(), nxt.b.data_ptr(), y.data_ptr(), t.data_ptr(), n, h, h)
        hip.check(hip.lib().usot_plan_add_bneck_first(self.plan.h, C.byref(d), 1 if dtype == torch.float16 else 0),
                  'plan_add_bneck_first ' + name)
        self.plan.keep += [x, w1, w2, w3c, b3c, wn, c1.b, c2.b, nxt.b]
        m = n * h * h
        macs = m * (c1.cout * c1.cin + c2.cout * 9 * c2.cin + c3.cout * (c3.cin + ds.cin) + nxt.cout * nxt.cin)
        self.log.append((name, m, c3.cout, c3.cin, 1, macs))
        self.lp_bytes.append(2 * (m * (c1.cin + c3.cout + nxt.cout) + w1.numel() + w2.numel() + w3c.numel() + wn.numel()))
        return y, t

    def bneck_tail(self, name, c2, c3, nxt, t1, res, n, h, dtype):
        """The rest of a layer1 bottleneck after its conv1 + the next block's conv1 in ONE launch (csrc/bneck_lp.hip:
        bneck_tail_kernel).  t1: this block's conv1 output, res: the block's input.  Returns (y, t) as bneck_first."""
        y = self.buf(n, h, h, c3.cout, dtype=dtype)
        t = self.buf(n, h, h, nxt.cout, dtype=dtype)
        w2, w3, wn = c2.w_lp(dtype), c3.w_lp(dtype), nxt.w_lp(dtype)
        d = hip.bneck_desc(t1.data_ptr(), res.data_ptr(), None, w2.data_ptr(), c2.b.data_ptr(), w3.data_ptr(),
                           c3.b.data_ptr(), wn.data_ptr(), nxt.b.data_ptr(), y.data_ptr(), t.data_ptr(), n, h, h)
        hip.check(hip.lib().usot_plan_add_bneck_tail(self.plan.h, C.byref(d), nxt.cout, 1 if dtype == torch.float16 else 0),
                  'plan_add_bneck_tail ' + name)
        self.plan.keep += [t1, res, w2, w3, wn, c2.b, c3.b, nxt.b]
        m = n * h * h
        self.log.append((name, m, c3.cout, c3.cin, 1, m * (c2.cout * 9 * c2.cin + c3.cout * c3.cin + nxt.cout * nxt.cin)))
        self.lp_bytes.append(2 * (m * (c2.cin + 2 * c3.cout + nxt.cout) + w2.numel() + w3.numel() + wn.numel()))
        return y, t

    def pw_pair_f32(self, name, c3, nxt, t2, res, n, h, act2=ACT_RELU, t2_parts=0, t2_bias=None, res_parts=0, res_bias=None):
        """fp32: conv3 + residual + ReLU and the next block's conv1 in ONE launch (csrc/smallm_f32.hip).
        Returns (y [n,h,h,c3.cout], t [n,h,h,nxt.cout]).  t2_parts > 1: t2 holds that many partial sums of the producing
        convolution (conv_deferred); the kernel stages relu(sum + t2_bias)."""
        m = n * h * h
        y = self.buf(n, h, h, c3.cout)
        t = self.buf(n, h, h, nxt.cout)
        # 'split16_pairs': the pair on split-fp16 operands (usot_pw_pair_f32s; the arithmetic of the split-fp16 conv tiles)
        s16 = bool(self.opt['split16_f32'] and (c3.cin, c3.cout, nxt.cout) in self.opt['split16_pairs'] and self.opt['fused_f32_sliced']
                   and hip.lib().usot_pw_pair_f32s_supported(c3.cin, c3.cout, nxt.cout)
                   and hip.lib().usot_pw_pair_f32_ws_floats(m, c3.cin, c3.cout, nxt.cout) > 0)
        w3p, w1p = (c3.w_pw_pair_s16(), nxt.w_pw_pair_s16()) if s16 else (c3.w_pw_pair_f32(), nxt.w_pw_pair_f32())
        # channel-sliced form (four workgroups per pixel tile when there are few tiles): 13.6 -> 8.7 us per layer2 pair, but
        # its k-sliced second GEMM moved ONE end-to-end golden output from 9.3e-5 to 1.011e-4 of the 1e-4 bar
        # (scripts/golden_margins.py; rounding noise, every variant sits at 8-9.5e-5) - off unless asked for
        ws = hip.pw_pair_f32_ws(m, c3.cin, c3.cout, nxt.cout, self.dev) if self.opt['fused_f32_sliced'] else None
        d = hip.pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), c3.b.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(),
                             nxt.b.data_ptr(), t.data_ptr(), m, c3.cin, c3.cout, nxt.cout, act2,
                             ws.data_ptr() if ws is not None else None,
                             t2_parts=t2_parts, t2_bias=t2_bias.data_ptr() if t2_bias is not None else None,
                             res_parts=res_parts, res_bias=res_bias.data_ptr() if res_bias is not None else None,
                             ovf=self.ovf_word().data_ptr() if s16 else None)
        hip.check(hip.lib().usot_plan_add_pw_pair(self.plan.h, C.byref(d), 3 if s16 else 2), 'plan_add_pw_pair(f32) ' + name)
        self.plan.keep += [t2, res, w3p, w1p, c3.b, nxt.b, ws, t2_bias, res_bias]
        self.log.append((name, m, c3.cout, c3.cin, 1, m * (c3.cout * c3.cin + nxt.cout * c3.cout)))
        self.f32_bytes.append(4 * (m * (c3.cin + 2 * c3.cout + nxt.cout) + c3.cout * c3.cin + nxt.cout * c3.cout))
        return y, t

    def pw_single(self, name, pc, x, n, h, act=ACT_NONE, res=None):
        """fp32 1x1 convolution on the small-M streaming kernel (csrc/smallm_f32.hip: pw_single_f32_kernel)."""
        m = n * h * h
        y = self.buf(n, h, h, pc.cout)
        wp = pc.w_pw_pair_f32()
        hip.check(hip.lib().usot_plan_add_pw_single(self.plan.h, hip.ptr(x), hip.ptr(wp), hip.ptr(pc.b),
                                                    hip.ptr(res) if res is not None else None, hip.ptr(y), m, pc.cin, pc.cout, act),
                  'plan_add_pw_single ' + name)
        self.plan.keep += [x, wp, pc.b, res]
        self.log.append((name, m, pc.cout, pc.cin, 1, m * pc.cout * pc.cin))
        self.f32_bytes.append(4 * (m * (pc.cin + pc.cout * (2 if res is not None else 1)) + pc.cout * pc.cin))
        return y, h, h

    def stream3x3(self, name, pc, x, n, h, w, act=ACT_NONE):
        """fp32 3x3 / stride-1 convolution on the small-M streaming kernel (stream_conv3x3_f32_kernel)."""
        oh, ow = pc.out_hw(h, w)
        y = self.buf(n, oh, ow, pc.cout)
        wp = pc.w_pw_pair_f32()
        hip.check(hip.lib().usot_plan_add_stream_conv3x3(self.plan.h, hip.ptr(x), hip.ptr(wp), hip.ptr(pc.b), None, hip.ptr(y),
                                                         n, h, w, pc.cin, oh, ow, pc.cout, pc.pad[0], pc.pad[1], pc.dil[0], pc.dil[1], act),
                  'plan_add_stream_conv3x3 ' + name)
        self.plan.keep += [x, wp, pc.b]
        self.log.append((name, n * oh * ow, pc.cout, 9 * pc.cin, 1, n * oh * ow * pc.cout * 9 * pc.cin))
        self.f32_bytes.append(4 * (n * h * w * pc.cin + n * oh * ow * pc.cout + pc.cout * 9 * pc.cin))
        return y, oh, ow

    def conv3x3(self, name, pc, x, n, h, w, act=ACT_NONE):
        """A backbone 3x3 convolution: the streaming kernel for the shapes / sizes where it wins at batch 1, else tiled."""
        oh, ow = pc.out_hw(h, w)
        tiles = (n * oh * ow + 15) // 16
        if (self.opt['stream_3x3'] and self.lanes == 0 and pc.kh == 3 and pc.stride == 1 and (pc.cin, pc.cout) in self.opt['stream_3x3_shapes']
                and tiles * 4 <= 256 and hip.lib().usot_stream_conv3x3_f32_supported(pc.cin, pc.cout)):
            return self.stream3x3(name, pc, x, n, h, w, act=act)
        return self.conv(name, pc, x, n, h, w, act=act)

    def conv1x1(self, name, pc, x, n, h, act=ACT_NONE, res=None):
        """A backbone 1x1 convolution: the streaming kernel when the layer has few pixels (batch 1), else the tiled one."""
        if (self.opt['stream_1x1'] and self.lanes == 0 and pc.kh == 1 and pc.stride == 1 and n * h * h <= self.opt['stream_1x1_max_m']
                and (pc.cin, pc.cout) in self.opt['stream_1x1_shapes'] and hip.lib().usot_pw_single_f32_supported(pc.cin, pc.cout)):
            return self.pw_single(name, pc, x, n, h, act=act, res=res)
        return self.conv(name, pc, x, n, h, h, act=act, res=res)

    def pw_triple_f32(self, name, c2, c3, nxt, t1, res, n, h, act2=ACT_RELU):
        """fp32: conv2 (3x3 / stride 1) + conv3 + residual + ReLU + the next block's conv1 in ONE launch (pw_triple_f32_kernel).
        Returns (y [n,oh,oh,c3.cout], t [n,oh,oh,nxt.cout])."""
        oh, ow = c2.out_hw(h, h)
        m = n * oh * ow
        y = self.buf(n, oh, ow, c3.cout)
        t = self.buf(n, oh, ow, nxt.cout)
        w2p, w3p, w1p = c2.w_pw_pair_f32(), c3.w_pw_pair_f32(), nxt.w_pw_pair_f32()
        d = hip.pw_pair_desc(None, w3p.data_ptr(), c3.b.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(),
                             nxt.b.data_ptr(), t.data_ptr(), m, c3.cin, c3.cout, nxt.cout, act2)
        hip.check(hip.lib().usot_plan_add_pw_triple(self.plan.h, hip.ptr(t1), hip.ptr(w2p), hip.ptr(c2.b), C.byref(d),
                                                    n, h, h, c2.cin, oh, ow, c2.pad[0], c2.pad[1], c2.dil[0], c2.dil[1]),
                  'plan_add_pw_triple ' + name)
        self.plan.keep += [t1, res, w2p, w3p, w1p, c2.b, c3.b, nxt.b]
        self.log.append((name, m, c3.cout, c3.cin, 1, m * (c2.cout * 9 * c2.cin + c3.cout * c3.cin + nxt.cout * c3.cout)))
        self.f32_bytes.append(4 * (n * h * h * c2.cin + m * (2 * c3.cout + nxt.cout)
                                   + c2.cout * 9 * c2.cin + c3.cout * c3.cin + nxt.cout * c3.cout))
        return y, t

    def cvt_lp(self, src, dtype):
        dst = self.buf(*src.shape, dtype=dtype)
        hip.check(hip.lib().usot_plan_add_cvt_lp(self.plan.h, hip.ptr(src), hip.ptr(dst), src.numel(),
                                                 1 if dtype == torch.float16 else 0), 'plan_add_cvt_lp')
        self.plan.keep += [src]
        return dst

    def heads_lp(self, xf_lp, b, hf, zk, mem_nhwc, m, dtype):
        """Heads with the big convolutions (search-side encoders, confidence/value, towers) at
        low-precision storage on the bf16/fp16 MFMA; the depthwise correlations (GroupDW), the
        Conf_Fusion reduction, the memory-kernel encoders and the 1/4-channel prediction convs stay fp32
        (BASELINE configs[4]: "fp16 backbone + fp32 xcorr").  xf_lp: neck output NHWC in `dtype`."""
        W, L = self.W, hip.lib()
        es = [self.conv_bf16('enc_s%d' % g, W.enc_s[g], xf_lp, b, hf, hf, act=ACT_RELU, dtype=dtype, out_f32=True)[0]
              for g in range(3)]
        S = hf - 6
        # lp_head_maps: GroupDW and the Conf_Fusion reduction store their maps in the storage type themselves (fp32 arithmetic,
        # one rounding at the store: the values the separate conversion passes produced, 37 + 21 us per batch of 32 less)
        direct = self.opt['lp_head_maps'] and S in (25, 27) and b * (2 + m) >= 64
        odt = 1 if dtype == torch.float16 else 2
        tin = self.buf(3, b, S, S, 256, dtype=dtype if direct else torch.float32)     # tower inputs: [reg, cls, memory]
        if self.opt['enc_k_lp']:
            # the memory features' kernel-side encoders on the low-precision MFMA too (fp32 output for the fp32 correlations):
            # at 32 streams x 7 memory features the fp32 launch was 197 us of the 3.1 ms batch (74 TFLOP/s)
            ml = self.cvt_lp(mem_nhwc, dtype)
            mk = [self.conv_bf16('enc_k%d.mem' % g, W.enc_k[g], ml, b * m, 7, 7, cout=256, act=ACT_RELU, dtype=dtype, out_f32=True)[0]
                  for g in range(3)]
        else:
            mk = self.encode_kernel(mem_nhwc, b * m, 256, 'mem')
        dwm = self.buf(b * m, S, S, 256, dtype=dtype if direct else torch.float32)
        segs = [self.groupdw(es, zk, tin[0], W.reg_wsm, b, 1, S, S, 256, 512),
                self.groupdw(es, zk, tin[1], W.cls_wsm, b, 1, S, S, 0, 512),
                self.groupdw(es, mk, dwm, W.cls_wsm, b * m, m, S, S, 0, 256)]
        if direct:
            arr = (hip.GroupDWDesc * 3)(*segs)
            hip.check(L.usot_plan_add_groupdw_multi_lp(self.plan.h, arr, 3, odt), 'plan_add_groupdw_multi_lp')
        else:
            self.groupdw_flush(segs)
        # the confidence | value map itself in fp16 (not bf16: exp(.) <= 54.6 on 8 bits would put 0.4 % on every weight): half the
        # bytes of the 287 MB map the reduction reads at 32 streams
        cv16 = direct and dtype == torch.float16
        cv, _, _ = self.conv_bf16('conf_fusion', W.conf, dwm if direct else self.cvt_lp(dwm, dtype), b * m, S, S, act=ACT_CONF, act2=ACT_RELU,
                                  act_split=256, dtype=dtype, out_f32=not cv16)
        if direct:
            hip.check(L.usot_plan_add_conf_reduce_lp(self.plan.h, hip.ptr(cv), 1 if cv16 else 0, hip.ptr(tin[2]), b, m, S * S, 256, odt),
                      'plan_add_conf_reduce_lp')
        else:
            hip.check(L.usot_plan_add_conf_reduce(self.plan.h, hip.ptr(cv), hip.ptr(tin[2]), b, m, S * S, 256),
                      'plan_add_conf_reduce')
        gs = b * S * S * 256
        cur = tin if direct else self.cvt_lp(tin, dtype)
        for i in range(4):
            cur, _, _ = self.conv_bf16('tower%d' % i, W.tower[i], cur, b, S, S, cout=256, act=ACT_RELU, groups=3, x_gs=gs,
                                       dtype=dtype, out_f32=(i == 3))
        bbox = self.buf(b, 4, S, S)
        cls2 = self.buf(2, b, 1, S, S)
        self.thin_convs([('bbox_pred', W.bbox_pred, cur[0], b, S, S, dict(act=ACT_EXP, y=bbox, y_nchw=True)),
                         ('cls_preds', W.cls_preds, cur[1], b, S, S, dict(cout=1, y=cls2, y_nchw=True, groups=2,
                                                                          x_gs=gs, y_gs=b * S * S, w_rows=1))])
        return bbox, cls2, S

    def backbone_bf16(self, x, n, size, dtype=torch.bfloat16, neck_f32=False, raw_pixels=True):
        """x NCHW fp32 [n,3,s,s] -> neck output NHWC bf16|fp16 (or fp32 with neck_f32)
        [n,hf,hf,256].  Stem + max-pool are one MFMA kernel (usot_stem_pool_lp): the crop is
        rounded to the storage type as it is staged, the 125x125 stem map never leaves LDS.
        INPUT CONTRACT: raw BGR pixels in 0..255 (lib/utils/track_utils.py:24-27 feeds them unnormalised).  The bf16 backbone's
        fp16-arithmetic stem (`stem_f16_math`) rounds crop - STEM_MU to 11 significant bits, which is adequate for that range
        only (a [0,1]-normalised crop would be quantised in steps of 2^-4 around -104); raw_pixels=False - the caller saw
        another range - keeps the hi + lo bf16 stem, which carries 16 bits of the crop whatever its scale."""
        W, L = self.W, hip.lib()
        dt = 1 if dtype == torch.float16 else 0
        oh = (size - 7) // 2 + 1
        ph = (oh - 1) // 2 + 1
        p0 = self.buf(n, ph, ph, 64, dtype=dtype)
        wdt = dtype
        if dtype == torch.bfloat16 and self.opt['stem_f16_math'] and raw_pixels and W.stem_fits_f16():
            wdt, dt = torch.float16, 2                 # fp16 arithmetic, bf16 storage
        wf, wb = W.stem_lp(wdt)
        hip.check(L.usot_plan_add_stem_pool_lp(self.plan.h, hip.ptr(x), hip.ptr(wf), hip.ptr(wb), hip.ptr(p0),
                                               n, size, size, oh, oh, ph, ph, dt, *W.STEM_MU), 'plan_add_stem_pool_lp')
        self.plan.keep += [wf, wb]
        self.plan.keep += [x]
        cur, h = p0, ph
        nb = len(W.blocks)
        nch, b0 = int(self.opt['lp_chains'] or 0), int(self.opt['lp_chains_from'])
        out = {}
        if nch < 2 or nch > 4 or self.lanes or n % nch or n // nch < self.opt['lp_chains_min_batch'] or not 0 < b0 < nb:
            for _ in self._lp_blocks(out, cur, None, h, n, 0, nb, dtype, neck_f32):
                pass
            self.p3 = out['cur']
            return out['xf'], out['h']
        # 'lp_chains': from block b0 on (layer3: MFMA-bound 3x3 convs alternating with HBM-bound 1x1 convs) the batch runs as
        # `nch` independent slices on parallel graph branches, chain i released `i * lp_chain_skew` launches of chain 0 late, so
        # that one chain's matrix-pipe phase meets another's HBM phase
        for _ in self._lp_blocks(out, cur, None, h, n, 0, b0, dtype, neck_f32):
            pass
        cur, t1, h = out['cur'], out['t1'], out['h']
        nc = n // nch
        store, cursor = [], [0] * nch
        outs = [dict() for _ in range(nch)]
        gens = [self._lp_blocks(outs[ci], cur[ci * nc:(ci + 1) * nc], t1[ci * nc:(ci + 1) * nc] if t1 is not None else None,
                                h, nc, b0, nb, dtype, neck_f32) for ci in range(nch)]

        def advance(ci, k=None):
            self._chain, self._nscale = (ci, nch, store, cursor), nch
            try:
                while k is None or k > 0:
                    next(gens[ci])
                    k = None if k is None else k - 1
            except StopIteration:
                pass
            finally:
                self._chain, self._nscale = None, 1
        skew = int(self.opt['lp_chain_skew'])
        for ci in range(1, nch):
            advance(0, skew)
            self.plan.fork(ci)                       # lane ci waits for what lane 0 holds so far, then runs its whole chain
            advance(ci)
            self.plan.fork(0)                        # back to lane 0 (no dependency)
        advance(0)
        for ci in range(1, nch):
            self.plan.join(ci)

        def whole(t):                                # the full-batch tensor a chain's result is a slice of
            return next(f for f in store if f.data_ptr() == t.data_ptr())
        self.p3 = whole(outs[0]['cur'])
        return whole(outs[0]['xf']), outs[0]['h']

    def _lp_blocks(self, out, cur, t1, h, n, b_from, b_to, dtype, neck_f32):
        """Generator: records bottlenecks [b_from, b_to) of the low-precision backbone (and the neck when b_to is the last block)
        for `n` crops, yielding after every launch; t1 = conv1 output of block b_from when the previous launch made it.
        Leaves out['cur'] (block output), out['t1'], out['h'] and, after the neck, out['xf']."""
        W = self.W
        fuse = getattr(self, 'fuse_pointwise', True)
        nb = len(W.blocks)
        rs = getattr(self, '_nscale', 1)
        for bi in range(b_from, b_to):
            c1, c2, c3, ds = W.blocks[bi]
            sc = cur
            nx1 = W.blocks[bi + 1][0] if bi + 1 < nb else None
            if (fuse and self.opt['bneck_first_lp'] and t1 is None and ds is not None and nx1 is not None
                    and ds.kh == 1 and ds.stride == 1 and c1.kh == 1 and nx1.kh == 1
                    and c2.kh == 3 and c2.stride == 1 and tuple(c2.pad) == (1, 1) and tuple(c2.dil) == (1, 1)
                    and hip.lib().usot_bneck_first_supported(c1.cin, c1.cout, c3.cout, nx1.cout)
                    and rs * n * ((h + 7) // 8) * ((h + 15) // 16) >= self.opt['bneck_first_min_tiles']):
                cur, t1 = self.bneck_first('b%d+b%d.conv1' % (bi, bi + 1), c1, c2, c3, ds, nx1, cur, n, h, dtype)
                yield
                continue
            if (fuse and self.opt['bneck_tail_lp'] and t1 is not None and ds is None and nx1 is not None and nx1.kh == 1
                    and nx1.stride == 1 and c2.kh == 3 and c2.stride == 1 and tuple(c2.pad) == (1, 1) and tuple(c2.dil) == (1, 1)
                    and c2.cin == 64 and hip.lib().usot_bneck_tail_supported(c2.cout, c3.cout, nx1.cout)
                    and rs * n * ((h + 7) // 8) * ((h + 15) // 16) >= self.opt['bneck_first_min_tiles']):
                cur, t1 = self.bneck_tail('b%d.conv2+conv3+b%d.conv1' % (bi, bi + 1), c2, c3, nx1, t1, cur, n, h, dtype)
                yield
                continue
            if ds is not None:
                sc, _, _ = self.conv_bf16('b%d.ds' % bi, ds, cur, n, h, h, dtype=dtype)
                yield
            if t1 is None:
                t1, _, _ = self.conv_bf16('b%d.conv1' % bi, c1, cur, n, h, h, act=ACT_RELU, dtype=dtype)
                yield
            lastb = bi + 1 == nb
            nxp = (W.neck if not neck_f32 else None) if lastb else W.blocks[bi + 1][0]
            if (fuse and t1 is not None and nxp is not None and nxp.kh == 1 and nxp.stride == 1
                    and c3.kh == 1 and c2.cin == c2.cout and hip.lib().usot_conv_pw_pair_supported(c3.cin, c3.cout, nxp.cout)
                    and self.opt['conv_pw_pair_lp' if c3.cin == 128 else 'conv_pw_p5_lp']
                    and rs * n * c2.out_hw(h, h)[0] ** 2 >= self.opt['panel_min_panels'] * 128):
                # conv2 -> conv3 + residual + ReLU -> the next block's conv1 (or the neck) in one launch
                cur, t1, h = self.conv_pw_pair('b%d.conv2+conv3+%s' % (bi, 'neck' if lastb else 'b%d.conv1' % (bi + 1)), c2, c3, nxp,
                                               t1, sc, n, h, ACT_NONE if lastb else ACT_RELU, dtype)
                yield
                continue
            if (fuse and c2.cout in (self.opt['conv_pw_lp'] or ()) and c2.stride == 1 and c3.kh == 1 and c2.out_hw(h, h) == (h, h)
                    and hip.lib().usot_conv_pw_supported(c2.cin, c2.cout, c3.cout)
                    and rs * n * h * h >= self.opt['panel_min_panels'] * 128):
                # conv2 -> conv3 + residual + ReLU in one launch (the T2 panel stays in LDS); the next 1x1 follows on its own
                cur, h = self.conv_pw('b%d.conv2+conv3' % bi, c2, c3, t1, sc, n, h, dtype)
                t1 = None
                yield
                continue
            t2, h2, _ = self.conv_bf16('b%d.conv2' % bi, c2, t1, n, h, h, act=ACT_RELU, dtype=dtype)
            yield
            # conv3 + residual + ReLU shares a launch with the NEXT 1x1 conv (the following block's conv1, or the
            # neck after the last block): the 4x-wide map is written once and not read back
            last = bi + 1 == nb
            nxt = (W.neck if not neck_f32 else None) if last else W.blocks[bi + 1][0]
            nm = 'b%d.conv3+%s' % (bi, 'neck' if last else 'b%d.conv1' % (bi + 1))
            if (fuse and nxt is not None and nxt.kh == 1 and (c3.cin, c3.cout, nxt.cout) in self.opt['panel_pair_lp']
                    and hip.lib().usot_pw_panel_pair_supported(c3.cin, c3.cout, nxt.cout)
                    and rs * n * h2 * h2 >= self.opt['panel_min_panels'] * hip.lib().usot_pw_panel_pixels(c3.cin, c3.cout, nxt.cout)):
                cur, t1 = self.pw_panel_pair(nm, c3, nxt, t2, sc, n, h2, ACT_NONE if last else ACT_RELU, dtype)
            elif fuse and nxt is not None and (c3.cin, c3.cout, nxt.cout) in self.opt['fused_pointwise_lp']:
                cur, t1 = self.pw_pair('b%d.conv3+%s' % (bi, 'neck' if last else 'b%d.conv1' % (bi + 1)), c3, nxt, t2, sc, n, h2,
                                       ACT_NONE if last else ACT_RELU, dtype)
            else:
                cur, _, _ = self.conv_bf16('b%d.conv3' % bi, c3, t2, n, h2, h2, act=ACT_RELU, res=sc, dtype=dtype)
                t1 = None
            yield
            h = h2
        out.update(cur=cur, t1=t1, h=h)
        if b_to < nb:
            return
        if t1 is not None:
            out['xf'] = t1                           # the neck rode along with the last conv3
            return
        out['xf'], _, _ = self.conv_bf16('neck', W.neck, cur, n, h, h, dtype=dtype, out_f32=neck_f32)
        yield

    # ---- a5 template side: zf NHWC [n,7,7,256] -> 3 maps NHWC [n,hk,wk,cout]
    def encode_kernel(self, zf, n, cout, tag):
        res = self.conv_batch([('enc_k%d.%s' % (g, tag), self.W.enc_k[g], zf, n, 7, 7, dict(cout=cout, act=ACT_RELU))
                               for g in range(3)])
        for g, (y, oh, ow) in enumerate(res):
            assert (oh, ow) == KGEO[g]
        return [r[0] for r in res]

    def groupdw(self, xs, zs, out, wsm, S, x_rep, oh, ow, x_co, z_cs):
        d = hip.groupdw_desc([t.data_ptr() for t in xs], [t.data_ptr() for t in zs], out.data_ptr(), wsm,
                             S=S, x_rep=x_rep, OH=oh, OW=ow, Cc=256, x_cs=[512] * 3, x_co=[x_co] * 3,
                             z_cs=[z_cs] * 3, z_co=[x_co if z_cs == 512 else 0] * 3)
        self.plan.keep += list(xs) + list(zs) + [out]
        return d

    def groupdw_flush(self, descs):
        arr = (hip.GroupDWDesc * len(descs))(*descs)
        hip.check(hip.lib().usot_plan_add_groupdw_multi(self.plan.h, arr, len(descs)), 'plan_add_groupdw_multi')

    # ---- a9: heads.  xf NHWC [b,hf,hf,256]; zk: 3 maps [b,hk,wk,512]; mem_nhwc [b*m,7,7,256] or None
    # `mk` may be passed when the memory-kernel encodes were already issued (on a side lane).
    def heads(self, xf, b, hf, zk, mem_nhwc, m, mk=None, mem_lane=None):
        W, L = self.W, hip.lib()
        es = [None] * 3
        if self.lanes < 3:                                # three encoder geometries in one launch
            eks = self.opt.get('enc_s_ksplit') or (None, None, None)
            res = self.conv_batch([('enc_s%d' % g, W.enc_s[g], xf, b, hf, hf,
                                    dict(act=ACT_RELU, **({'force_ks': eks[g]} if eks[g] else {}))) for g in range(3)])
            es = [r[0] for r in res]
        else:
            for g, lane in ((1, 1), (2, 3), (0, 0)):
                self.fork(lane)
                es[g], _, _ = self.conv('enc_s%d' % g, W.enc_s[g], xf, b, hf, hf, act=ACT_RELU)
            self.join(1)
            self.join(3)
        S = hf - 6                                    # response size (25 for 31, 27 for 33)
        has_mem = mem_nhwc is not None
        ngroups = 3 if has_mem else 2
        tin = self.buf(ngroups, b, S, S, 256)         # tower inputs: [reg, cls, (memory)]
        segs = [self.groupdw(es, zk, tin[0], W.reg_wsm, b, 1, S, S, 256, 512),
                self.groupdw(es, zk, tin[1], W.cls_wsm, b, 1, S, S, 0, 512)]
        if has_mem:
            if mk is None:
                mk = self.encode_kernel(mem_nhwc, b * m, 256, 'mem')
            elif mem_lane is not None:
                self.join(mem_lane, 1)
            dwm = self.buf(b * m, S, S, 256)
            segs.append(self.groupdw(es, mk, dwm, W.cls_wsm, b * m, m, S, S, 0, 256))
        self.groupdw_flush(segs)
        gs = b * S * S * 256
        tout = [self.buf(ngroups, b, S, S, 256) for _ in range(4)]
        bbox = self.buf(b, 4, S, S)
        cls2 = self.buf(ngroups - 1, b, 1, S, S)      # [cls, (cls_mem)]
        if has_mem and self.lanes < 2:
            # serial schedule: one 3-group launch per tower level (fewest launches)
            ts = self.opt.get('conf_tail_split')
            if ts and b * m > ts[0]:
                # whole-chip rounds: the last `tail` memory maps run split-K in the same launch (see DEFAULT_OPTIONS)
                tail, ks = ts
                head_n = b * m - tail
                cv = self.buf(b * m, S, S, 512)
                kw = dict(act=ACT_CONF, act2=ACT_RELU, act_split=256)
                # both parts on the tile tuned for the WHOLE convolution (passed explicitly: the tuning table is shared by
                # every later plan of this engine and is not written to while a plan is built)
                full_tile = self.tuning.get((b * m * S * S, 512, W.conf.kh * W.conf.kw * W.conf.cin, 1), (0, 1))[0]
                citems = [('conf_fusion', W.conf, dwm[:head_n], head_n, S, S, dict(y=cv[:head_n], force_ks=1, **kw)),
                          ('conf_fusion.tail', W.conf, dwm[head_n:], tail, S, S, dict(y=cv[head_n:], force_ks=ks, **kw))]
                skew = bool(self.opt.get('skew_towers')) and self.batch
                if skew:
                    # the reg / cls towers do not wait for the memory branch: their first level rides in Conf_Fusion's launch
                    # (same tile, same K), and every later launch pairs tower level i of the memory branch with level i + 1 of
                    # reg | cls - the memory branch's conv + reduction no longer delay the other two towers (DEFAULT_OPTIONS)
                    citems.append(('tower0.rc', W.tower[0], tin, b, S, S, dict(cout=256, act=ACT_RELU, y=tout[0], groups=2, x_gs=gs, y_gs=gs)))
                self.conv_batch(citems, lead_tile=full_tile or None)
            else:
                skew = False
                cv, _, _ = self.conv('conf_fusion', W.conf, dwm, b * m, S, S, act=ACT_CONF, act2=ACT_RELU, act_split=256)
            hip.check(L.usot_plan_add_conf_reduce(self.plan.h, hip.ptr(cv), hip.ptr(tin[2]), b, m, S * S, 256),
                      'plan_add_conf_reduce')
            cur = tin
            if skew:
                # the memory tower runs one level behind: [tower_i(mem) | tower_{i+1}(reg, cls)] per launch, tower_3(mem) alone
                tt = self.tuning.get((b * S * S, 256, W.tower[0].kh * W.tower[0].kw * W.tower[0].cin, 3), (0, 1))[0] or None
                for i in range(4):
                    items = [('tower%d.mem' % i, W.tower[i], (tin if i == 0 else tout[i - 1])[2], b, S, S,
                              dict(cout=256, act=ACT_RELU, y=tout[i][2], row0=512))]
                    if i < 3:
                        items.append(('tower%d.rc' % (i + 1), W.tower[i + 1], tout[i], b, S, S,
                                      dict(cout=256, act=ACT_RELU, y=tout[i + 1], groups=2, x_gs=gs, y_gs=gs)))
                    self.conv_batch(items, lead_tile=tt)
                cur = tout[3]
            else:
                for i in range(4):
                    self.conv('tower%d' % i, W.tower[i], cur, b, S, S, cout=256, act=ACT_RELU, y=tout[i], groups=3,
                              x_gs=gs, y_gs=gs)
                    cur = tout[i]
            self.thin_convs([('bbox_pred', W.bbox_pred, cur[0], b, S, S, dict(act=ACT_EXP, y=bbox, y_nchw=True)),
                             ('cls_preds', W.cls_preds, cur[1], b, S, S, dict(cout=1, y=cls2, y_nchw=True, groups=2,
                                                                              x_gs=gs, y_gs=b * S * S, w_rows=1))])
            return bbox, cls2, S
        if has_mem:
            # lane 1: confidence/value conv -> fusion -> memory tower -> cls_mem;  lane 0: reg + cls
            self.fork(1, 2)
            cv, _, _ = self.conv('conf_fusion', W.conf, dwm, b * m, S, S, act=ACT_CONF, act2=ACT_RELU, act_split=256)
            hip.check(L.usot_plan_add_conf_reduce(self.plan.h, hip.ptr(cv), hip.ptr(tin[2]), b, m, S * S, 256),
                      'plan_add_conf_reduce')
            cur = tin[2]
            for i in range(4):
                self.conv('tower%d.mem' % i, W.tower[i], cur, b, S, S, cout=256, act=ACT_RELU, y=tout[i][2], row0=512)
                cur = tout[i][2]
            self.conv('cls_mem_pred', W.cls_preds, cur, b, S, S, cout=1, y=cls2[1], y_nchw=True, row0=1)
            self.fork(0, 2)
        cur = tin
        for i in range(4):
            self.conv('tower%d' % i, W.tower[i], cur, b, S, S, cout=256, act=ACT_RELU, y=tout[i], groups=2,
                      x_gs=gs, y_gs=gs)
            cur = tout[i]
        self.fork(3)
        self.conv('bbox_pred', W.bbox_pred, cur[0], b, S, S, act=ACT_EXP, y=bbox, y_nchw=True)
        self.fork(0)
        self.conv('cls_pred', W.cls_preds, cur[1], b, S, S, cout=1, y=cls2[0], y_nchw=True)
        self.join(3)
        if has_mem:
            self.join(1, 2)
        return bbox, cls2, S


def looks_like_raw_pixels(x):
    """True when a crop batch is in the reference's input convention, raw 0..255 pixel values (one min / max reduction, done
    ONCE when a low-precision plan is built): the range the fp16-arithmetic stem of the bf16 backbone is adequate for."""
    lo, hi = float(x.min()), float(x.max())
    return lo >= 0.0 and 8.0 < hi <= 256.0


def _as_dev_f32(t, device):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device=device, dtype=torch.float32)


# fp32 producer / consumer tile id -> its split-fp16 twin (same tile shape, producer depth and producer waves)
SPLIT16_TILES = {55: 91, 56: 92, 53: 94, 54: 95, 57: 97, 39: 99, 31: 99, 41: 104, 49: 104, 33: 104}      # (routed fp32 tiles only: the library holds no others)


def load_tuning(path=None):
    """{(M, Cout, K, groups): (tile, ksplit)} measured by scripts/tune_conv.py on gfx950.
    Shapes missing from the table fall back to the launcher's heuristic."""
    import json
    import os
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'tuning_gfx950.json')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        raw = json.load(f)
    return {tuple(int(v) for v in k.split(',')): (int(t), int(ks)) for k, (t, ks) in raw.items()}


def load_lp_tuning(path=None):
    """{(M, Cout, K): tile} for the bf16/fp16 kernels (scripts/tune_lp.py); heuristic otherwise."""
    import json
    import os
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'tuning_lp_gfx950.json')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {tuple(int(v) for v in k.split(',')): int(t) for k, t in json.load(f).items()}


LP_TUNING = load_lp_tuning()


def load_split16_tuning(path=None):
    """{(M, Cout, K, groups): (split-fp16 tile, ksplit)} measured in the frame graph (scripts/ks_ab.py) on gfx950; shapes missing here
    run on the split-fp16 twin (SPLIT16_TILES) of their fp32 tile."""
    import json
    import os
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'tuning_split16_gfx950.json')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {tuple(int(v) for v in k.split(',')): (int(t[0]), int(t[1])) for k, t in json.load(f).items()}


S16_TUNING = load_split16_tuning()


def _shapes(text):
    return {tuple(int(v) for v in t.split('x')) for t in text.split(',') if t}


# Every switch of the lowering in ONE place.  `OPTIONS` is what Builders read when a plan is built; an Engine takes
# per-engine overrides (Engine(..., options={...}), or model.engine_options['options']).  The environment variables in
# ENV_SWITCHES are read once, at import, for experiments from the shell; tests enumerate DEFAULT_OPTIONS
# (tests/test_gpu_model.py::test_engine_option_variants_keep_parity).  Measurements behind the defaults: DESIGN.md §3.1.
DEFAULT_OPTIONS = {
    # (C_mid, C_out, C_next) of the conv3 -> next-conv1 pairs that run as ONE launch (csrc/pw_pair.hip) in the batched
    # low-precision backbone: the shapes where the fused kernel measured faster than the two launches at batch 64
    'fused_pointwise_lp': {(64, 256, 64), (64, 256, 128), (128, 512, 128)},
    # (K, N) of the unfused 1x1 EXPANSION convolutions of the low-precision backbone that run on the pixel-stationary
    # panel kernel (csrc/pw_panel.hip) when the launch has at least panel_min_panels pixel panels (one workgroup = one CU each:
    # batch 32 at layer2 / layer3 resolution is 120 panels — half the chip — and stays on the tiled kernels): layer3's
    # conv3 89 -> 66 us, layer1's 1x1 shortcut 50 -> 38 at batch 64
    'panel_1x1_lp': {(256, 1024), (128, 512), (64, 256)},
    'panel_min_panels': 192,
    # OPT-IN fast mode of the fp32 frame (default: exact fp32 products on v_mfma_f32_16x16x4_f32, the reference's arithmetic).  True:
    # fp32 convolutions with K >= split16_min_k on the producer / consumer tiles run on the split-fp16 form of the tile (every operand
    # as hi + lo fp16 - 22 significant bits - three fp16 MFMAs per product block, fp32 accumulation: csrc/conv_igemm.hip PF = 4 | 5):
    # the frame 0.84 -> 0.68 ms at the same 1e-4 parity.  Operands must stay below 8 188 in magnitude; a launch that sees one beyond it
    # says so in a sticky device word (usot_conv_desc.ovf) and the engine re-plans on the exact tiles and runs the call / frame again
    # (Engine._to_exact, Session._rerun_exact) - the mode never returns a wrong finite number and never fails mid-video
    'split16_f32': False,
    'split16_min_k': 1152,
    # (CM, CO, CN) of the fused fp32 pointwise pairs that run on split-fp16 operands too (layer3's six conv3 + conv1 pairs)
    'split16_pairs': {(256, 1024, 256)},
    'split16_min_m': 64,
    # Session: frame t's bank append (encode + scatter) is DEFERRED into frame t + 1's graph instead of sitting behind frame t's result
    # tag (Session._build; Session.flush() for readers of the bank between frames).  1 / True: on a side branch at the start of the
    # graph (measured + 29 us: a fork + join cost more than the kernels they hide)
    # 2 (round 6): the same deferral with NO side branch - the previous frame's encoder convolutions ride in layer2's shortcut-conv
    # launch and one kernel in front of the heads appends + gathers (usot_rows_append_gather_f32); 0 / False = in-frame append
    'defer_append': 2,
    # split-K of the riding encoder problems (36 k-tiles): short workgroups that leave the shortcut conv's CUs early.  Frame graph on one
    # box (scripts/ab_frame.py): ks 1: 839.5 us, 2: 830.5, 4: 826.5, 12: 822; on another 3: 837, 9: 842, 12 / 18 / 36: 829-832
    'defer_append_enc_ks': 12,
    # layer3's conv2 -> conv3 (+ residual + ReLU) of the batched low-precision backbone in ONE launch (csrc/conv_pw_lp.hip: a 256-pixel
    # panel of conv2's output stays in LDS).  Bit-identical to the two launches; measured in the batch-64 bf16 step.
    # Value: the conv2 widths it is used for ((256,) = layer3; 128 = layer2's last block, whose next conv1 has no pair form)
    'conv_pw_lp': (256, 128),
    # layer2's bottleneck tails the same way, with the next block's conv1 riding along (pair form)
    'conv_pw_pair_lp': True,
    # layer3's blocks likewise, the next block's conv1 (or the neck) as a FIFTH phase of the launch: the workgroup reads its own Y
    # panel back (csrc/conv_pw_lp.hip) - the standalone 1024 -> 256 launch disappears
    'conv_pw_p5_lp': True,
    # EXPERIMENT (USOT_EXPERIMENTS=1 builds only; measured SLOWER, csrc/conv_pw_ov.hip: 234 vs 146 us per block): those launches on the
    # OVERLAPPED form - two kinds of 8-wave workgroup per CU, conv2 (and the next conv1) on the matrix pipe beside conv3's residual / Y
    # streams of another panel - from conv_pw_ov_min_panels 128-pixel panels up
    'conv_pw_ov_lp': False,
    'conv_pw_ov_min_panels': 384,
    # phase 1 of those kernels on the ROW-SHARED k-loop where the geometry allows it (3 x 3, stride 1, pad = dil: one staged activation
    # tile per (kh, channel chunk) serves the three kw taps - a third fewer LDS-DMA instructions per k-tile; k order (kh, chunk, kw), so
    # results differ from the per-tap loop's by fp32 summation order).  False: the per-tap loop, bit-identical to the unfused launches
    'conv_pw_rs': True,
    # panel size of those kernels: 0 = the launcher's rule (usot_conv_pw_pixels), 1 = 256 pixels / 16 wavefronts, 2 = 128 / 8 (A/B switch)
    'conv_pw_panel': 0,
    # layer3 (from block lp_chains_from on) as lp_chains independent batch slices on parallel graph branches, chain i released
    # i * lp_chain_skew launches of chain 0 late (backbone_bf16); 0 = one chain
    'lp_chains': 0,
    'lp_chains_from': 7,
    'lp_chain_skew': 2,
    'lp_chains_min_batch': 8,
    # layer1's FIRST bottleneck (1x1 downsample) + the next block's conv1 as ONE launch of the batched low-precision backbone
    # (csrc/bneck_lp.hip: t1 / t2 stay in LDS, the shortcut conv rides on conv3's k axis) when the launch has at least
    # bneck_first_min_tiles 8 x 16 tiles (two per CU)
    'bneck_first_lp': True,
    'bneck_first_min_tiles': 512,
    # heads_lp (configs[4]): the memory features' kernel-side encoders (connect.py:55-74 `_k` branches) on the low-precision MFMA
    'enc_k_lp': True,
    # heads_lp: GroupDW and the Conf_Fusion reduction write their output maps in the storage type (no conversion passes)
    'lp_head_maps': True,
    # ... and the REST of layer1's other bottlenecks (conv2 + conv3 + residual) + the following conv1 likewise (bneck_tail_kernel)
    'bneck_tail_lp': True,
    # (Cin, Cout) of the 3x3 / stride-1 / pad-1 convolutions of the low-precision backbone that run as direct convolutions
    # from an LDS halo tile (csrc/conv3x3_halo.hip) when the launch has at least one 16 x 16 tile per CU
    'halo_3x3_lp': {(64, 64)},
    # (K, N) of the channel-REDUCING 1x1 convolutions of the low-precision backbone that run with stationary accumulators and
    # K streaming three chunks ahead (csrc/pw_kstream.hip) when the launch has at least panel_min_panels panels of 256 pixels:
    # layer3's conv1 and the neck's 1x1 (1024 -> 256)
    'kstream_1x1_lp': {(1024, 256)},
    # (Cin, Cout) of the 3x3 convolutions with K >= 2304 that run in the same accumulator-stationary form (csrc/conv_kstream.hip).
    # EMPTY by default: parity-green but 5-8 % slower than the tiled 256 x 256 kernel on all three candidates at batch 64 —
    # layer3's shortcut conv (512, 1024) 494 vs 456 us, its conv2 (256, 256) 72 vs 68, layer2's shortcut conv (256, 512) 158 vs
    # 151: the B fragments fetched straight from global memory are 16 rows x 64 B per instruction and for a 3x3 conv they are
    # L2 hits nine times over, which the texture path serves worse than the LDS-DMA's whole lines
    'kstream_3x3_lp': set(),
    # the stem of the bf16 backbone computes on the fp16 MFMA (crop - mu and the folded filters rounded to 11 significant
    # bits, one MFMA per fragment) and stores bf16; False: bf16 filters (8 bits) against the crop as hi + lo bf16, two MFMAs
    # per fragment — less accurate AND 98 instead of 65 us at batch 64.  Falls back by itself when a folded filter leaves
    # the fp16 normal range.
    'stem_f16_math': True,
    # (C_mid, C_out, C_next) of the conv3 -> next-1x1 pairs of the low-precision backbone that run as ONE launch of the
    # panel kernel's pair form (Y's accumulators feed the second GEMM from registers); takes precedence over
    # fused_pointwise_lp (csrc/pw_pair.hip: 64-pixel tiles with an LDS image of Y)
    # (not layer3's (256, 1024, 256): csrc/pw_panel.hip, usot_pw_panel_pair_supported)
    'panel_pair_lp': {(64, 256, 64), (64, 256, 128), (128, 512, 128), (128, 512, 256)},
    # the same fusion in the fp32 frame (csrc/smallm_f32.hip, 16 pixels per workgroup): at batch 1 these 1x1 layers are
    # launch-bound (two launches 20-23 us, fused 6-14: scripts/pw_pair_f32_probe.py).  Not (128, 512, 256): layer3.0's
    # conv1 already rides in the shortcut conv's launch.  Above max_m pixels the tiled conv kernels fill the chip and win.
    'fused_pointwise_f32': {(64, 256, 64), (64, 256, 128), (128, 512, 128)},
    'fused_pointwise_f32_max_m': 4 * 3969,
    # layer1: conv2 (3x3) joins the pair's launch (pw_triple_f32_kernel): 16.0 -> 10.9 us per bottleneck isolated, graph
    # replay 873.7 -> 863.2 us.  layer2's form (128 x 128 x 512 x 128; 61 workgroups) makes the frame slower (861 -> 881).
    'fused_triple_f32': True,
    'fused_triple_f32_shapes': {(64, 64, 256, 64), (64, 64, 256, 128)},
    # the unfused 1x1 EXPANSION convolutions (conv3 of layer3's blocks and of layer2's last) on the small-M streaming
    # kernel up to max_m pixels: graph replay 896.6 -> 878 us.  The 1024 -> 256 reductions gain nothing there (898 us with
    # them alone, 877-881 with everything).  (Cin, Cout) pairs.
    'stream_1x1': True,
    'stream_1x1_max_m': 1200,
    'stream_1x1_shapes': {(256, 1024), (128, 512)},
    # conv2 (3x3, stride 1) on the streaming form when its pixel tiles x 4 channel slices fit one wave of workgroups
    # (255-pixel crops at batch 1: 61 tiles).  layer2's (128 -> 128, K = 1152): 10.9 -> 7.4 us isolated, graph replay
    # 881.7 -> 875.0 us.  layer3's (256 -> 256, K = 2304) streams 144 MB of filters per layer from L2 - 20.3 -> 18.6 us
    # isolated but 878 -> 892 us in the frame - and stays on the tiled kernel.
    'stream_3x3': True,
    'stream_3x3_shapes': {(128, 128)},
    # channel-sliced fused pairs (four workgroups per pixel tile, in-launch ticket combine): 13.6 -> 8.7 us per layer2
    # pair; with it layer3's pairs and conv3 + neck (which exist in the sliced form only: an unsliced 16 x 1024 Y tile
    # does not fit) fuse too.  Round 2 shipped it OFF: its re-associated second GEMM put one golden output at 1.011e-4 of
    # the 1e-4 bar.  With blocked accumulation + the offset stem every output sits at <= 5e-5 and the variant passes the
    # float64 acceptance rule (tests/golden/f64_gate.py) on both weight families: ON, graph replay 880.5 -> 855.7 us.
    'fused_f32_sliced': True,
    'fused_pointwise_f32_sliced_only': {(256, 1024, 256)},
    # Conf_Fusion's conv (7 memory maps x 625 pixels = 137 x 8 tiles of 32 x 64 over 512 workgroup slots = 2.14 rounds):
    # (tail maps, ksplit) runs the last maps split-K beside the others in ONE launch so that the launch ends on whole-chip
    # rounds; None = one unsplit convolution.  Same-process A/B of the frame graph (scripts/tail_split_ab.py): unsplit 866 us,
    # (1, 2) 851.5, (1, 7) 856, (2, 3) 856, (1, 3) 870
    'conf_tail_split': (1, 2),
    # per-geometry split-K factors of the three search-side encoder convs that share one launch (None = the tuned table)
    'enc_s_ksplit': None,
    # Session.collect(): wall-clock budget of the result-tag spin before it falls back to 50 us sleeps
    'spin_seconds': 0.004,
    # low-precision plans: are the crops raw 0..255 pixel values (the range the fp16-arithmetic stem is adequate for)?
    # 'auto' = decided ONCE per input shape from the first batch (Engine._raw_pixels); True / False = stated by the caller
    'lp_raw_pixels': 'auto',
    # {(M, Cout, K): (tile, ksplit)} of the 3x3 convolutions in front of a fused fp32 pointwise pair whose split-K reduction is
    # DEFERRED to that pair (csrc/smallm_f32.hip sums the partial tiles, adds the bias and applies the ReLU while it stages its
    # pixel tile): the convolution's launch ends on its k-loop - no ticket combine (4-6 us on the tail of a 20 us launch), no
    # reduction launch - so layer3's conv2 (M = 961: too few 32 x 64 / 64 x 64 tiles to fill 256 CUs unsplit) can run on tiles
    # that move fewer L2 bytes than the 32 x 32 ones.  Same-process A/B of the frame graph (scripts/ks_ab.py, two boxes): layer3's six
    # conv2 (32 x 64, ks 2) 858-871 -> 843-845 us, + layer2.0's conv2 (32 x 32, ks 2, was ks 2 with the in-launch combine) -> 838;
    # 64 x 64 tiles with ks 4 are SLOWER (862-868).  {} = every reduction inside its own launch
    # {(M, Cout, K) of conv2: (tile, ksplit)}: a block's 3x3 / stride-2 shortcut conv and its conv2 (both inputs ready: the block's
    # input map and conv1's output, which the previous block's fused launch produced) share ONE launch, conv2 deferred
    # EMPTY by default: measured +8 us per frame ((961, 128, 1152): (55, 2); scripts/ks_ab.py) - a launch lasts (rounds) x (one tile's
    # k-loop latency), and the two convolutions side by side are not shorter than one after the other
    'batch_ds_conv2': {},
    # the reg / cls towers run one level AHEAD of the memory tower (level 0 in Conf_Fusion's launch): see Builder.heads.  OFF: measured
    # +33 us per frame - tower0's 160 workgroups behind Conf_Fusion's 1 264 open another round of 36 k-steps (143.6 vs 110.9 us), and
    # the memory tower's last level alone (80 workgroups) takes the same 30 us as a three-tower level (per-op spans, ks_ab.py)
    'skew_towers': False,
    # {(M, Cout, K) of a block's 3x3 shortcut conv: (tile, ksplit)}: the same for the SHORTCUT branch of a block whose conv2 is deferred - the
    # pair adds (sum of the parts + the shortcut's bias) as its residual
    # EMPTY by default: measured slower ((961, 512, 2304): (55, 2) 858 vs 834 us of graph, (56, 2) 836.5, (55, 3) 843): the 3 x 3 / stride-2
    # shortcut conv of layer2.0 already fills one round, a second workgroup per CU stretches every k-step by more than the halved loop saves
    'defer_split_res_f32': {},
    'defer_split_f32': {(961, 256, 2304): (55, 2), (1089, 256, 2304): (55, 2), (961, 128, 1152): (53, 2), (1089, 128, 1152): (53, 2)},
    # ... and the (tile, ksplit) those deferred launches use when split16_f32 is on (filter-DMA tiles, csrc/conv_igemm.hip PF = 5)
    'defer_split_s16': {(961, 256, 2304): (106, 2), (1089, 256, 2304): (106, 2), (961, 128, 1152): (111, 2), (1089, 128, 1152): (111, 2)},
}
ENV_SWITCHES = {      # environment variable -> (option, parser)
    'USOT_FUSED_TRIPLE_F32': ('fused_triple_f32', lambda v: v == '1'),
    'USOT_FUSED_TRIPLE_F32_SHAPES': ('fused_triple_f32_shapes', _shapes),
    'USOT_STREAM_1X1': ('stream_1x1', lambda v: v == '1'),
    'USOT_STREAM_1X1_SHAPES': ('stream_1x1_shapes', _shapes),
    'USOT_STREAM_3X3': ('stream_3x3', lambda v: v == '1'),
    'USOT_STREAM_3X3_SHAPES': ('stream_3x3_shapes', _shapes),
    'USOT_FUSED_F32_SLICED': ('fused_f32_sliced', lambda v: v == '1'),
    'USOT_SPIN_SECONDS': ('spin_seconds', float),
    'USOT_SPLIT16_F32': ('split16_f32', lambda v: v == '1'),
    'USOT_DEFER_APPEND': ('defer_append', int),
    'USOT_DEFER_APPEND_ENC_KS': ('defer_append_enc_ks', int),
    'USOT_CONV_PW_LP': ('conv_pw_lp', lambda v: tuple(int(t) for t in v.split(',') if t)),      # '' = off, '256', '256,128'
    'USOT_CONV_PW_PAIR_LP': ('conv_pw_pair_lp', lambda v: v == '1'),
    'USOT_CONV_PW_RS': ('conv_pw_rs', lambda v: v == '1'),
    'USOT_CONV_PW_PANEL': ('conv_pw_panel', int),
    'USOT_CONV_PW_P5_LP': ('conv_pw_p5_lp', lambda v: v == '1'),
    'USOT_CONV_PW_OV_LP': ('conv_pw_ov_lp', lambda v: v == '1'),
    'USOT_CONV_PW_OV_MIN_PANELS': ('conv_pw_ov_min_panels', int),
}


def options_from_env(env=None):
    env = os.environ if env is None else env
    opt = {k: (set(v) if isinstance(v, set) else v) for k, v in DEFAULT_OPTIONS.items()}
    for var, (key, parse) in ENV_SWITCHES.items():
        if var in env:
            opt[key] = parse(env[var])
    return opt


OPTIONS = options_from_env()


def merged_options(overrides=None):
    opt = {k: (set(v) if isinstance(v, set) else v) for k, v in OPTIONS.items()}
    for k, v in (overrides or {}).items():
        if k not in DEFAULT_OPTIONS:
            raise hip.HipError('unknown engine option %r (known: %s)' % (k, ', '.join(sorted(DEFAULT_OPTIONS))))
        opt[k] = v
    ts = opt.get('conf_tail_split')
    if ts is not None and not (isinstance(ts, (tuple, list)) and len(ts) == 2 and all(isinstance(v, int) and not isinstance(v, bool) for v in ts)
                               and ts[0] >= 1 and ts[1] >= 2):
        raise hip.HipError('conf_tail_split must be None or (tail maps >= 1, ksplit >= 2); got %r' % (ts,))
    return opt


class Engine:
    """Per-model, per-device executor with cached plans.  Stateless w.r.t. tracking."""

    def __init__(self, model, device, graphs=True, tuning=None, lanes=0, options=None):
        self.opt = merged_options(options)          # engine.OPTIONS + per-engine overrides, fixed for this engine's plans
        if torch.device(device).type != 'cuda':
            raise hip.HipError('the USOT HIP engine needs a GPU device; got %s (no CPU fallback)' % (device,))
        hip.lib()
        self.device = torch.device(device)
        self.W = Weights(model, self.device)
        self.graphs = graphs
        self.lanes = int(lanes) if graphs else 0   # lanes only exist as parallel branches of a captured graph
        self.tuning = load_tuning() if tuning is None else tuning
        self._feat = {}       # (n, size) -> dict(x, xf, h, plan)
        self._raw_seen = {}   # low-precision plans, 'auto' mode: shape -> were the first batch's crops raw pixels
        self._zenc = {}       # n -> dict(zf, zk, plan)
        self._track = {}      # (b, size, m) -> dict
        self._zk_key = None

    # ------------------------------------------------------------------ features
    def _feat_plan(self, n, size):
        key = (n, size)
        if key not in self._feat:
            bld = Builder(self.W, self.tuning, self.lanes, self.opt)
            x = bld.buf(n, 3, size, size)
            xf, h = bld.backbone(x, n, size)
            self._finish(bld.plan, bld.ovf)
            self._feat[key] = dict(x=x, xf=xf, h=h, plan=bld.plan, log=bld.log, stages=bld.stages, ovf=bld.ovf)
        return self._feat[key]

    # ------------------------------------------------------------------ split-fp16 range contract
    def _overflowed(self, p):
        """True when the plan that just replayed saw a not-finite sum on a split-fp16 launch (usot_conv_desc.ovf): an activation
        left the fp16 window (|x| >= 8 188).  Reads ONE device word (a stream synchronisation - only plans that contain split-fp16
        launches have the word, i.e. only engines with 'split16_f32' on pay it) and clears it."""
        w = p.get('ovf')
        if w is None or int(w.item()) == 0:
            return False
        w.zero_()
        return True

    def _to_exact(self, where):
        """The automatic fallback of the opt-in split-fp16 mode: from now on EVERY plan of this engine is built on the exact-fp32
        tiles (the reference's arithmetic: models.py:179-198 returns finite maps for any finite fp32 input).  Sticky: a checkpoint
        that drove one activation out of the fp16 window will do it again."""
        if self.opt['split16_f32']:
            warnings.warn('usot_amd: %s: an activation left the range of the split-fp16 products (|x| >= 8 188); this engine '
                          'continues on the exact-fp32 tiles (engine option split16_f32 -> False)' % (where,), RuntimeWarning, stacklevel=3)
        self.opt['split16_f32'] = False
        self.range_fallbacks = getattr(self, 'range_fallbacks', 0) + 1
        self._feat.clear()
        self._track.clear()
        # the template encodes (what sessions snapshot and track() reads) are REBUILT on the exact tiles from the features they were
        # made of, not dropped: the caller's template() is still in force
        made_of = {n: e['zf'].clone() for n, e in self._zenc.items()}
        key = self._zk_key
        self._zenc.clear()
        for n, zf in made_of.items():
            self.encode_template(zf.permute(0, 3, 1, 2))
        self._zk_key = key

    def _finish(self, plan, ovf=None):
        if self.graphs:
            plan.run()                       # warm: module load, first-touch
            torch.cuda.current_stream().synchronize()
            plan.capture()
        if ovf is not None:
            # the warm-up replay ran on UNINITIALISED input buffers (torch.empty: any bit pattern, NaN and inf included), which the
            # split-fp16 launches duly reported: that is not a verdict on the caller's data
            torch.cuda.current_stream().synchronize()
            ovf.zero_()

    def features(self, x):
        """x NCHW [n,3,s,s] -> xf as an NCHW-shaped view of the NHWC result [n,256,hf,hf].
        The view aliases engine workspace: valid until the next call at this (n, s)."""
        x = _as_dev_f32(x, self.device)
        n, c, s, s2 = x.shape
        assert c == 3 and s == s2
        p = self._feat_plan(n, s)
        p['x'].copy_(x)
        p['plan'].run()
        if self._overflowed(p):
            self._to_exact('features()')
            return self.features(x)
        return p['xf'].permute(0, 3, 1, 2)

    def _raw_pixels(self, x, shape_key, raw_pixels):
        """Whether a low-precision plan may stage its crops for the fp16-arithmetic stem (inputs in the reference's convention,
        raw 0..255 pixel values).  Explicit argument > engine option `lp_raw_pixels` (True / False) > 'auto': ONE min / max
        reduction of the first batch of a shape (two host syncs), remembered for that shape - later batches of the shape are NOT
        re-checked (that would put two blocking reductions into every call); the resolved value is part of the plan key, so a
        caller that passes raw_pixels explicitly gets its own plan per input convention."""
        if raw_pixels is not None:
            return bool(raw_pixels)
        opt = self.opt.get('lp_raw_pixels', 'auto')
        if opt != 'auto':
            return bool(opt)
        if shape_key not in self._raw_seen:
            self._raw_seen[shape_key] = looks_like_raw_pixels(x)
        return self._raw_seen[shape_key]

    def features_bf16(self, x, dtype=torch.bfloat16, raw_pixels=None):
        """Batched low-precision backbone + neck (config 3): x NCHW fp32 [n,3,s,s] -> NCHW-shaped
        bf16|fp16 view [n,256,hf,hf] of the NHWC result.  Workspace view: valid until the next call.
        raw_pixels: see _raw_pixels (None = the engine option, 'auto' by default = decided by the first batch of this shape)."""
        x = _as_dev_f32(x, self.device)
        n, _, s, _ = x.shape
        raw = self._raw_pixels(x, ('feat', n, s), raw_pixels)
        key = ('bf16' if dtype == torch.bfloat16 else 'f16', n, s, raw)
        if key not in self._feat:
            bld = Builder(self.W, self.tuning, 0, self.opt)
            xin = bld.buf(n, 3, s, s)
            xf, h = bld.backbone_bf16(xin, n, s, dtype=dtype, raw_pixels=raw)
            self._finish(bld.plan)
            self._feat[key] = dict(x=xin, xf=xf, h=h, plan=bld.plan, log=bld.log, p3=bld.p3, lp_bytes=bld.lp_bytes, lp_readback=bld.lp_readback)
        p = self._feat[key]
        p['x'].copy_(x)
        p['plan'].run()
        return p['xf'].permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ template side
    def encode_template(self, zf_nchw):
        """zf NCHW-shaped [n,256,7,7] (any strides) -> cached merged cls|reg kernel maps."""
        n = zf_nchw.shape[0]
        if n not in self._zenc:
            bld = Builder(self.W, self.tuning, self.lanes, self.opt)
            zf = bld.buf(n, 7, 7, 256)
            zk = bld.encode_kernel(zf, n, 512, 'z')
            self._finish(bld.plan, bld.ovf)
            self._zenc[n] = dict(zf=zf, zk=zk, plan=bld.plan, ovf=bld.ovf)
        e = self._zenc[n]
        src = hip.to_nhwc(zf_nchw)
        if src.data_ptr() != e['zf'].data_ptr():
            e['zf'].copy_(src)
        e['plan'].run()
        if self._overflowed(e):
            src = src.clone()                    # (may alias the plan's own input buffer, which _to_exact drops)
            self._to_exact('encode_template()')
            return self.encode_template(src.permute(0, 3, 1, 2))
        return e

    def template(self, z, bbox=None, pr_pool=True):
        """models.py:173-177.  Returns zf NCHW-shaped [n,256,7,7] (channels-last memory)."""
        xf = self.features(z)                              # [n,256,15,15] view
        n = xf.shape[0]
        if pr_pool:
            if bbox is None:
                raise hip.HipError('template(pr_pool=True) needs template_bbox')
            rois = self._rois(bbox, n)
            zf = hip.prroi_pool(xf, rois, 7, 7, 1.0, out_nhwc=True)
        else:
            zf = hip.to_nhwc(xf[:, :, 4:-4, 4:-4]).permute(0, 3, 1, 2)      # connect.py:303-306
        self.set_template(zf)
        return zf

    def set_template(self, zf):
        e = self.encode_template(zf)
        self._zk_key = (zf.data_ptr(), zf._version, tuple(zf.shape))
        return e

    def _rois(self, boxes, n):
        boxes = _as_dev_f32(boxes, self.device).reshape(n, 4)
        idx = torch.arange(n, device=self.device, dtype=torch.float32).reshape(n, 1)
        return torch.cat([idx, boxes], 1).contiguous()

    # ------------------------------------------------------------------ search side
    def _track_plan(self, b, size, m):
        key = (b, size, m)
        if key not in self._track:
            bld = Builder(self.W, self.tuning, self.lanes, self.opt)
            x = bld.buf(b, 3, size, size)
            mem = bld.buf(b * m, 7, 7, 256) if m else None
            mk = None
            if m:
                bld.fork(2, 1)
                mk = bld.encode_kernel(mem, b * m, 256, 'mem')
                bld.fork(0, 1)
            xf, hf = bld.backbone(x, b, size, need_stem=False)
            zk = self._zenc[b]['zk']
            bbox, cls2, S = bld.heads(xf, b, hf, zk, mem, m, mk=mk, mem_lane=2 if m else None)
            self._finish(bld.plan, bld.ovf)
            self._track[key] = dict(x=x, xf=xf, hf=hf, mem=mem, bbox=bbox, cls2=cls2, S=S, plan=bld.plan,
                                    log=bld.log, ovf=bld.ovf)
        return self._track[key]

    def track(self, x, zf, template_mem=None, score_mem=None, clone=True):
        """models.py:179-198 -> (cls, bbox, cls_mem, xf) / (cls, bbox, None, None)."""
        x = _as_dev_f32(x, self.device)
        b, _, size, _ = x.shape
        if zf is None:
            raise hip.HipError('track() before template()')
        if self._zk_key != (zf.data_ptr(), zf._version, tuple(zf.shape)) or b not in self._zenc:
            self.set_template(zf)
        m = 0
        if template_mem is not None:
            m = int(score_mem.shape[1]) if score_mem is not None else template_mem.shape[0] // b
            if template_mem.shape[0] != b * m:
                raise hip.HipError('template_mem has %d kernels, expected %d x %d' % (template_mem.shape[0], b, m))
        p = self._track_plan(b, size, m)
        p['x'].copy_(x)
        if m:
            tm = _as_dev_f32(template_mem, self.device)
            src = hip.to_nhwc(tm)
            if src.data_ptr() != p['mem'].data_ptr():
                p['mem'].copy_(src)
        p['plan'].run()
        if self._overflowed(p):
            # models.py:179-198 returns finite maps for any finite fp32 input: run the call again on the exact-fp32 tiles
            self._to_exact('track()')
            return self.track(x, zf, template_mem, score_mem, clone)
        cls = p['cls2'][0]
        bbox = p['bbox']
        if clone:
            cls, bbox = cls.clone(), bbox.clone()
        if not m:
            return cls, bbox, None, None
        cls_mem = p['cls2'][1].clone() if clone else p['cls2'][1]
        return cls, bbox, cls_mem, p['xf'].permute(0, 3, 1, 2)

    def track_mixed(self, x, zf, template_mem, score_mem, dtype=torch.float16, heads_lp=True, raw_pixels=None):
        """BASELINE config 5: low-precision (fp16 | bf16) backbone + neck on MFMA; heads_lp: the big head
        convolutions too (Builder.heads_lp), else every head op in fp32.  The depthwise correlations
        are fp32 either way.  Same returns as track(); batch = independent streams."""
        x = _as_dev_f32(x, self.device)
        b, _, size, _ = x.shape
        if self._zk_key != (zf.data_ptr(), zf._version, tuple(zf.shape)) or b not in self._zenc:
            self.set_template(zf)
        m = int(score_mem.shape[1])
        raw = self._raw_pixels(x, ('mixed', b, size), raw_pixels)
        key = ('mixed', b, size, m, dtype, bool(heads_lp), raw)
        if key not in self._track:
            bld = Builder(self.W, self.tuning, 0, self.opt)
            xin = bld.buf(b, 3, size, size)
            mem = bld.buf(b * m, 7, 7, 256)
            if heads_lp:
                xl, hf = bld.backbone_bf16(xin, b, size, dtype=dtype, neck_f32=False, raw_pixels=raw)
                bbox, cls2, S = bld.heads_lp(xl, b, hf, self._zenc[b]['zk'], mem, m, dtype)
                xf = xl                                   # returned as the neck map (low precision)
            else:
                xf, hf = bld.backbone_bf16(xin, b, size, dtype=dtype, neck_f32=True, raw_pixels=raw)
                bbox, cls2, S = bld.heads(xf, b, hf, self._zenc[b]['zk'], mem, m)
            self._finish(bld.plan)
            self._track[key] = dict(x=xin, xf=xf, hf=hf, mem=mem, bbox=bbox, cls2=cls2, S=S, plan=bld.plan, log=bld.log,
                                    lp_bytes=bld.lp_bytes, f32_bytes=bld.f32_bytes, lp_readback=bld.lp_readback)
        p = self._track[key]
        p['x'].copy_(x)
        p['mem'].copy_(hip.to_nhwc(_as_dev_f32(template_mem, self.device)))
        p['plan'].run()
        return p['cls2'][0].clone(), p['bbox'].clone(), p['cls2'][1].clone(), p['xf'].float().permute(0, 3, 1, 2)

    def pool(self, xf, boxes):
        """models.py:164-171: PrRoIPool 7x7, scale 1, batch index prepended -> NCHW dense."""
        xf = _as_dev_f32(xf, self.device)
        return hip.prroi_pool(xf, self._rois(boxes, xf.shape[0]), 7, 7, 1.0)

    # ------------------------------------------------------------------ sessions
    def open_session(self, p, window, init_feats):
        """Device-resident tracking state for one video (see Session).  Must be called
        right after template(): it snapshots the current template encodes."""
        return Session(self, p, window, init_feats, capacity=getattr(self, 'session_capacity', 1024))


class Session:
    """One video's state kept in HBM + its per-frame launch plan (a captured hipGraph).

    bank rows: 0 = init-frame feature, 1 = its left/right flip, 2+i = memory feature i
    (the reference keeps these as CPU tensors in python lists and re-uploads seven per
    frame, usot_tracker.py:222-258,264).  The three kernel-side encodings of a memory feature
    (connect.py:55-74 `_k` branches, a pure function of the feature) are computed ONCE, when the
    feature is appended, and kept in `bank_enc` beside it; the reference re-encodes the seven
    picked features every frame (connect.py:251-255).  One frame (engine option 'defer_append' = 2, the default) =
        backbone (the PREVIOUS frame's pooled feature is encoded by three problems riding in layer2's shortcut-conv launch)
        -> neck -> append that feature + its encodings to the banks and gather the 7 picked rows' encodings by the control
        block's indices (one kernel) -> heads -> decode -> PrRoIPool of the winning box (appended by the NEXT frame, or by
        flush()),
    with one ~100-byte control block written before and one 64-byte result block polled after (both pinned host memory the
    kernels address directly).  'defer_append' = 0 is the in-frame form: gather -> backbone -> ... -> PrRoIPool -> encode -> scatter.
    """

    ROW = 7 * 7 * 256

    def __init__(self, engine, p, window, init_feats, capacity=1024):
        self.e = engine
        self.p = p
        dev = engine.device
        self.size, self.S = int(p.instance_size), int(p.score_size)
        self.zk = [t.clone() for t in engine._zenc[1]['zk']]
        self.window = torch.from_numpy(np.ascontiguousarray(window, dtype=np.float64)).reshape(-1).to(dev)
        self.cap = capacity
        self.bank = torch.zeros(capacity, 7, 7, 256, device=dev)
        for i, f in enumerate((init_feats[0], init_feats[1], init_feats[0])):
            self.bank[i].copy_(hip.to_nhwc(f)[0])
        self.bank_enc = [torch.zeros(capacity, hk, wk, 256, device=dev) for hk, wk in KGEO]
        self._encode_rows(0, 3)
        self.n = 1                                   # memory features stored so far
        # Control and result blocks live in pinned (device-mapped, coherent) HOST memory that the
        # kernels address directly: the 64-byte per-frame upload/download needs no copy
        # kernels at all — the host writes ctl, launches the graph, synchronises, reads out.
        # layout: [0:16] target size (2 doubles), [48:56] frame tag (double), [64:] N_q gather rows +
        # 1 scatter row + the crop's device address as two int32 (0 = the session's own input buffer) + the PREVIOUS frame's
        # scatter row (engine option 'defer_append')
        self.nq = int(getattr(p, 'mem_queue_size', 7))
        if self.nq < 4:
            raise hip.HipError('mem_queue_size must be >= 4 (init, flip, >= 1 sampled, last); got %d' % self.nq)
        self.ctl = torch.zeros(64 + 4 * (self.nq + 4 + (self.nq + 4) % 2), dtype=torch.uint8).pin_memory()
        self.out8 = torch.zeros(16, dtype=torch.float64).pin_memory()   # 8 results + completion tag
        self.x_host = torch.zeros(1, 3, self.size, self.size).pin_memory()
        self._x_host_np = self.x_host.numpy()
        self._build()

    def _build(self):
        e, L = self.e, hip.lib()
        bld = Builder(e.W, e.tuning, e.lanes, e.opt)
        pl = bld.plan
        nq = self.nq
        self.x = bld.buf(1, 3, self.size, self.size)
        self.x.zero_()                      # the warm-up replay below reads it (torch.empty memory may decode as NaN)
        self.mem_in = bld.buf(1)            # heads() only asks whether there is a memory branch
        tsz_dev = self.ctl[0:56].view(torch.float64)          # [0:2] target size, [6] frame tag
        idx_dev = self.ctl[64:64 + 4 * (nq + 4)].view(torch.int32)   # N_q gather rows + 1 scatter row + crop address (lo, hi) + previous scatter row
        self._ctl_f64 = tsz_dev.numpy()
        self._ctl_i32 = idx_dev.numpy()
        self._ctl_u32 = self._ctl_i32.view(np.uint32)
        self._ctl_u64 = self.ctl[0:56].view(torch.int64).numpy().view(np.uint64)
        self._rows = np.zeros(nq, np.int32)
        self._rows[1] = 1
        self._out_np = self.out8.numpy()
        # the 7 picked memory kernels: their cached encodings, three banks in one gather
        mk = [bld.buf(nq, hk, wk, 256) for hk, wk in KGEO]
        self.slot_dev = torch.zeros(4, dtype=torch.int32, device=e.device)
        self.roi = bld.buf(5)
        self.feat = bld.buf(1, 7, 7, 256)
        self.feat.zero_()
        # 'defer_append': the bank append of frame t (encode the pooled feature, scatter feature + encodings) does not sit
        # behind frame t's result tag in front of frame t + 1 but at the START of frame t + 1's graph, on a side branch beside
        # the stem and layer1, followed there by frame t + 1's gather (whose picks include the row just appended);
        # the branch joins in front of the heads.  Every control-block read (gather rows, the previous scatter row, the crop's
        # address in the stem, the target size in the decode) happens before the tag, so the host may rewrite the block as soon
        # as collect() returns; flush() appends the pending feature for anybody who reads the bank between frames.
        self.defer = int(e.opt['defer_append']) if e.lanes == 0 else 0
        if self.defer == 2 and nq > 32:             # usot_rows_append_gather_f32 takes up to 32 picked rows: longer queues append in-frame
            self.defer = 0
        self._pending, self._prev_slot = False, self.cap - 1
        if self.defer == 2:
            # 'defer_append' = 2 (round 6): the same deferral WITHOUT a side branch.  The previous frame's three encoder
            # convolutions ride in the launch of layer2's shortcut conv (Builder.piggyback), and ONE kernel in front of the heads
            # appends feature + encodings to their banks and gathers this frame's picked rows (usot_rows_append_gather_f32: a
            # picked row that is the appended one is read from the fresh encodings) - two launches fewer than the in-frame
            # append (encode, scatter, gather -> one), nothing forked.  The stem reads the crop's address from the control block.
            bld.piggyback = [('enc_k%d.mem' % g, e.W.enc_k[g], self.feat, 1, 7, 7,
                              dict(cout=256, act=ACT_RELU, force_ks=e.opt['defer_append_enc_ks'])) for g in range(3)]
            xf, hf = bld.backbone(self.x, 1, self.size, need_stem=False, xptr_dev=idx_dev[nq + 1:nq + 3])
            new_enc = bld.piggy_out
            if new_enc is None:                      # no stand-alone shortcut conv took them (other lowering options): own launch
                bld.piggyback = None
                new_enc = bld.encode_kernel(self.feat, 1, 256, 'mem')
            fresh, banks = [self.feat] + new_enc, [self.bank] + self.bank_enc
            rl = [int(b[0].numel()) for b in banks]
            hip.check(L.usot_plan_add_rows_append_gather(
                pl.h, (C.c_void_p * 4)(*[t.data_ptr() for t in fresh]), (C.c_void_p * 4)(*[t.data_ptr() for t in banks]),
                (C.c_void_p * 3)(*[t.data_ptr() for t in mk]), (C.c_int32 * 4)(*rl), hip.ptr(idx_dev), nq, nq + 3),
                'plan_add_rows_append_gather')
        elif self.defer:
            pl.fork(3)
            new_enc = bld.encode_kernel(self.feat, 1, 256, 'mem')       # the PREVIOUS frame's pooled feature
            self._rows_multi(pl, [self.feat] + new_enc, idx_dev[nq + 3:], [self.bank] + self.bank_enc, 1, scatter=1)
            self._rows_multi(pl, self.bank_enc, idx_dev, mk, nq, scatter=0, stash=self.slot_dev)
            pl.fork(0)
            xf, hf = bld.backbone(self.x, 1, self.size, need_stem=False, xptr_dev=idx_dev[nq + 1:nq + 3])
            pl.join(3)
        else:
            # the append row of this frame is stashed in device memory: the scatter at the end of
            # the graph runs AFTER the result tag the host waits for, i.e. possibly while the host is
            # already writing the next frame's control block
            self._rows_multi(pl, self.bank_enc, idx_dev, mk, nq, scatter=0, stash=self.slot_dev)
            # slot_dev[0] = the append row, slot_dev[1:3] = the crop's address: both stashed from the control block by the
            # gather above (the first kernel of the frame)
            xf, hf = bld.backbone(self.x, 1, self.size, need_stem=False, xptr_dev=self.slot_dev[1:3])
        bbox, cls2, S = bld.heads(xf, 1, hf, self.zk, self.mem_in, nq, mk=mk, mem_lane=None if self.defer else 2)
        assert S == self.S
        p = self.p
        hip.check(L.usot_plan_add_decode(pl.h, hip.ptr(cls2[0]), hip.ptr(cls2[1]), hip.ptr(bbox), hip.ptr(self.window),
                                         hip.ptr(self.out8), S, self.size, int(p.total_stride), float(p.ratio),
                                         float(p.penalty_k), float(p.window_influence), hip.ptr(tsz_dev),
                                         hip.ptr(self.roi)), 'plan_add_decode')
        c = 256
        hip.check(L.usot_plan_add_prroi(pl.h, hip.ptr(xf), hip.ptr(self.roi), hip.ptr(self.feat), 1, c, hf, hf, 7, 7, 1.0,
                                        hf * hf * c, 1, hf * c, c, 49 * c, 1, 7 * c, c), 'plan_add_prroi')
        if not self.defer:
            new_enc = bld.encode_kernel(self.feat, 1, 256, 'mem')       # the new feature's encodings, once
            self._rows_multi(pl, [self.feat] + new_enc, self.slot_dev, [self.bank] + self.bank_enc, 1, scatter=1)
        else:
            # flush(): the same append as a small plan of its own, its row read from a pinned word
            fb = Builder(e.W, e.tuning, 0, e.opt)
            self._flush_idx = torch.zeros(2, dtype=torch.int32).pin_memory()
            fenc = fb.encode_kernel(self.feat, 1, 256, 'mem')
            self._rows_multi(fb.plan, [self.feat] + fenc, self._flush_idx, [self.bank] + self.bank_enc, 1, scatter=1)
            fb.plan.keep += fenc + [self.feat, self._flush_idx, self.bank] + self.bank_enc
            self._flush_plan = fb.plan
        pl.keep += mk + new_enc + self.bank_enc + [self.slot_dev]
        pl.keep += [self.bank, self.ctl, self.window, self.out8, tsz_dev, idx_dev] + self.zk
        self.xf, self.cls2, self.bbox, self.plan, self.log = xf, cls2, bbox, pl, bld.log
        self.f32_bytes = bld.f32_bytes
        # split-fp16 range word of this frame graph (Builder.ovf): its address travels in the control block, the decode kernel
        # publishes its value as out[9] with the results and clears it (csrc/head_ops.hip); 0 = no split-fp16 launch in the graph
        self._ovf = bld.ovf
        self._ctl_u64[3] = bld.ovf.data_ptr() if bld.ovf is not None else 0
        self._out_np[9] = 0.0
        # warm-up must not leave a stray row in the bank: point the scatter at a scratch row
        self._set_ctl([0, 1] + [2] * (nq - 2), self.cap - 1, (64.0, 64.0))
        self._ctl_f64[6] = -1.0
        e._finish(pl, bld.ovf)
        torch.cuda.current_stream().synchronize()
        self.feat.zero_()                   # the warm-up replay pooled a feature of the zero crop: not a pending append
        torch.cuda.current_stream().synchronize()

    def flush(self):
        """'defer_append': append the last frame's pooled feature to the bank NOW (it otherwise happens inside the next frame's
        graph).  For readers of the bank between frames; the next frame rewrites the same row with the same values (mode 2: the
        same sums in the riding launch's split-K order, i.e. equal to fp32 rounding)."""
        if not getattr(self, 'defer', False) or not self._pending:
            return
        st = getattr(self, '_stream', None) or torch.cuda.current_stream()
        with torch.cuda.stream(st):
            self._flush_idx[0] = self._prev_slot
            self._flush_plan.run()
        st.synchronize()

    @staticmethod
    def _rows_multi(pl, srcs, idx, dsts, n_rows, scatter, stash=None):
        n = len(srcs)
        rl = [int(bank[0].numel()) for bank in (dsts if scatter else srcs)]      # row = one bank entry
        hip.check(hip.lib().usot_plan_add_rows_copy_multi(
            pl.h, n, (C.c_void_p * n)(*[t.data_ptr() for t in srcs]), hip.ptr(idx),
            (C.c_void_p * n)(*[t.data_ptr() for t in dsts]), n_rows, (C.c_int32 * n)(*rl), scatter,
            hip.ptr(stash) if stash is not None else None), 'plan_add_rows_copy_multi')

    def _encode_rows(self, lo, hi):
        """Encode bank rows [lo, hi) into bank_enc (session start: init feature, its flip, memory 0)."""
        bld = Builder(self.e.W, self.e.tuning, 0, self.e.opt)
        src = bld.buf(hi - lo, 7, 7, 256)
        src.copy_(self.bank[lo:hi])
        enc = bld.encode_kernel(src, hi - lo, 256, 'mem')
        bld.plan.run()
        torch.cuda.current_stream().synchronize()
        if bld.ovf is not None and int(bld.ovf.item()):
            self.e._to_exact('Session: memory-feature encoders')
            return self._encode_rows(lo, hi)
        for g in range(3):
            self.bank_enc[g][lo:hi].copy_(enc[g])
        torch.cuda.current_stream().synchronize()

    def _set_ctl(self, rows, slot, tsz, xaddr=0):
        if len(rows) != self.nq:
            raise hip.HipError('%d memory rows for a session built for mem_queue_size = %d' % (len(rows), self.nq))
        f, c, nq = self._ctl_f64, self._ctl_i32, self.nq
        f[0] = tsz[0]
        f[1] = tsz[1]
        c[:nq] = rows
        c[nq] = slot
        self._ctl_u32[nq + 1] = xaddr & 0xffffffff          # the crop's device address, low / high half (0 = own buffer)
        self._ctl_u32[nq + 2] = xaddr >> 32
        c[nq + 3] = self._prev_slot                         # 'defer_append': where the previous frame's feature goes

    def _ensure_capacity(self):
        """Grow BEFORE anything is written into the plan's input buffer: growing rebuilds the
        plan and its workspace."""
        if 2 + self.n >= self.cap - 1:
            self._grow()

    def _grow(self):
        self.flush()
        bank = torch.zeros(self.cap * 2, 7, 7, 256, device=self.e.device)
        bank[:self.cap].copy_(self.bank)
        enc = [torch.zeros(self.cap * 2, hk, wk, 256, device=self.e.device) for hk, wk in KGEO]
        for g in range(3):
            enc[g][:self.cap].copy_(self.bank_enc[g])
        self.bank, self.bank_enc, self.cap = bank, enc, self.cap * 2
        self._build()

    def submit(self, x_crop, picks, tsz_scaled, resident=False, inplace=False):
        """Enqueue one frame on the CURRENT stream and return immediately (several sessions on
        separate streams overlap on the GPU).  Pair with collect().

        x_crop is SNAPSHOT by default: host crops go through the pinned staging buffer, device crops are copied
        into the session's input buffer (any device, dtype or strides), so the caller may reuse its buffer as
        soon as submit() returns — the double-buffer pattern.
        inplace=True opts into zero-copy: a contiguous float32 crop on the session's device is read where it
        lies (its address travels in the control block, no device copy).  LIFETIME CONTRACT: the tensor must stay
        alive and UNCHANGED until collect() returns for this frame; the session holds a reference until then.
        Anything that does not qualify (other device, dtype, strides) raises instead of silently copying.
        resident=True: the crop is already in the session's own input buffer (frame_from_image)."""
        self._ensure_capacity()
        xaddr = 0                                   # 0: the frame reads the session's own input buffer
        if not resident:
            if inplace:
                if not (x_crop.is_cuda and x_crop.device == self.e.device and x_crop.dtype == torch.float32
                        and x_crop.is_contiguous() and x_crop.numel() == self.x.numel()):
                    raise hip.HipError('submit(inplace=True) needs a contiguous float32 crop of %d elements on %s; got %s %s %s'
                                       % (self.x.numel(), self.e.device, x_crop.device, x_crop.dtype, tuple(x_crop.shape)))
                xaddr, self._x_ref = x_crop.data_ptr(), x_crop
            elif x_crop.is_cuda:
                self.x.copy_(x_crop.reshape(self.x.shape))            # handles other devices / dtypes / strides
            else:
                np.copyto(self._x_host_np, x_crop.numpy().reshape(self._x_host_np.shape))
                self.x.copy_(self.x_host, non_blocking=True)
        if len(picks) != self.nq - 2:
            raise hip.HipError('%d sampled memory slots for a session built for mem_queue_size = %d' % (len(picks), self.nq))
        rows = self._rows
        rows[2:] = picks
        rows[2:] += 2                               # bank rows 0 / 1 = init feature and its flip, 2 + i = memory i
        self._set_ctl(rows, 2 + self.n, tsz_scaled, xaddr)
        self._tag = float(self.n)
        self._ctl_f64[6] = self._tag
        # the launch first, the bookkeeping behind it: with only the PrRoIPool behind the previous frame's tag the GPU waits for this
        # call (loop - graph = 6 us) - one current_stream() lookup instead of two, _last_ctl after the graph is on its way
        self._stream = st = torch.cuda.current_stream()
        hip.check(hip.lib().usot_plan_run(self.plan.h, C.c_void_p(st.cuda_stream)), 'usot_plan_run')
        self._last_ctl = (rows.copy(), 2 + self.n, (float(tsz_scaled[0]), float(tsz_scaled[1])), xaddr)

    def collect(self):
        """Wait for the submitted frame's result block: float64[8] = (argmax, score, penalty,
        x1, y1, x2, y2, pscore).  The decode kernel publishes the results and then the tag: poll it
        rather than sleeping in hipStreamSynchronize (the PrRoIPool + bank append behind it are
        ordered before the next frame by the stream)."""
        try:
            out = self._collect()
        except BaseException:
            # a frame that never published its tag may still be running and reads the in-place crop by ADDRESS: drain the
            # stream before the reference (the only thing keeping that memory away from the caching allocator) is dropped
            try:
                self._stream.synchronize()
            finally:
                self._x_ref = None
            raise
        self._x_ref = None                          # the tag was observed: the crop may be reused from here on
        return out

    def _rerun_exact(self):
        """The frame just collected saw an activation beyond the split-fp16 window (out[9]): rebuild this session's frame graph on
        the exact-fp32 tiles (the engine's option flips for every later plan too) and run the SAME frame again - same crop, same
        picks, same bank row, which the second run overwrites - instead of failing mid-video (the reference returns finite maps for
        any finite fp32 input, models.py:179-198)."""
        rows, slot, tsz, xaddr = self._last_ctl
        self._stream.synchronize()                  # the failed frame's PrRoIPool + bank append behind the tag
        x_keep = self.x.clone() if xaddr == 0 else None
        with torch.cuda.stream(self._stream):
            self.e._to_exact('Session frame %r' % (self._tag,))
            self._build()                           # (leaves out[8] = -1: the warm-up replay's tag)
            if x_keep is not None:
                self.x.copy_(x_keep)
            # 'defer_append': the previous frame's row was appended at the start of the failed run; the rebuilt graph's pooled-feature
            # buffer is empty, so its start-of-graph append goes to the scratch row
            self._prev_slot = self.cap - 1
            self._set_ctl(rows, slot, tsz, xaddr)
            self._ctl_f64[6] = self._tag
            self.plan.run()
        self._stream.synchronize()
        if self._out_np[8] != self._tag:
            raise hip.HipError('frame %r: the exact-fp32 re-run never published its result block' % (self._tag,))

    def _collect(self):
        out, tag = self._out_np, self._tag
        # spin for SPIN_SECONDS (several frame times), then give the core away between polls.  The budget is wall-clock: a
        # fixed 20 000 polls turned out to be 0.905 ms on this host (45 ns per numpy compare), i.e. it ran out right around
        # the end of a 0.86 ms frame and every such frame then paid a 50 us sleep and a launch onto an idle queue
        spin_until = time.perf_counter() + self.e.opt['spin_seconds']
        while True:
            for _ in range(512):
                if out[8] == tag:
                    break
            else:
                if time.perf_counter() < spin_until:
                    continue
            break
        if out[8] != tag:
            deadline = time.monotonic() + 20.0
            while out[8] != tag and time.monotonic() < deadline:
                time.sleep(5e-5)
            if out[8] != tag:
                self._stream.synchronize()
                if out[8] != tag:
                    raise hip.HipError('frame %r never published its result block (tag reads %r): '
                                       'the frame graph did not run to the decode kernel' % (tag, float(out[8])))
        if self._ovf is not None and out[9] != 0.0:
            self._rerun_exact()                     # split-fp16 range contract broken by this frame: the same frame on the exact-fp32 tiles
            out = self._out_np
        self._prev_slot, self._pending = 2 + self.n, True      # ('defer_append': this frame's row, written by the next graph)
        self.n += 1
        return out[:8].copy()

    def append_feature(self, feat):
        """Append a memory feature computed OUTSIDE the frame graph (NCHW-shaped [1,256,7,7], any strides):
        bank row + its three kernel-side encodings.  Lets a caller that switches from the fused path to
        the generic `update()` mid-video keep one memory queue (usot_tracker.py:264)."""
        self._ensure_capacity()
        self.flush()
        torch.cuda.current_stream().synchronize()
        row = 2 + self.n
        self.bank[row].copy_(hip.to_nhwc(_as_dev_f32(feat, self.e.device))[0])
        self._encode_rows(row, row + 1)
        self.n += 1

    def frame(self, x_crop, picks, tsz_scaled, resident=False, inplace=False):
        """Run one frame.  x_crop: CHW float tensor (host or device); picks: memory indices
        of the N_q-2 sampled slots; tsz_scaled: target size * scale_z; inplace as in submit()."""
        self.submit(x_crop, picks, tsz_scaled, resident, inplace)
        return self.collect()

    def frame_from_image(self, im, pos, win, avg_chans, picks, tsz_scaled):
        """Whole frame from the raw image: upload the uint8 frame, crop/pad/resize on the
        device straight into the plan's input buffer (hostutils.get_subwindow_tracking's
        arithmetic), then `frame`.  `win` = python2round(s_x)."""
        from .hostutils import crop_geometry
        self._ensure_capacity()
        h, w, _ = im.shape
        if getattr(self, '_im_dev', None) is None or tuple(self._im_dev.shape) != (h, w, 3):
            self._im_dev = torch.empty((h, w, 3), dtype=torch.uint8, device=self.e.device)
            self._im_host = torch.empty((h, w, 3), dtype=torch.uint8).pin_memory()
            self._im_host_np = self._im_host.numpy()
        # numpy memcpy into the pinned staging buffer: torch's CPU copy_ fans out over every
        # hardware thread it sees (5.6 ms for 0.5 MB under a 16-CPU cgroup quota on a 256-thread host)
        np.copyto(self._im_host_np, im)
        self._im_dev.copy_(self._im_host, non_blocking=True)
        (cx0, _, cy0, _), (top, _, left, _) = crop_geometry(im.shape, pos, win)
        fill = np.asarray(avg_chans).astype(np.uint8)         # numpy's float -> uint8 assignment truncates
        hip.crop_resize(self._im_dev, self.x[0], int(cx0) - left, int(cy0) - top, int(win), fill)
        return self.frame(None, picks, tsz_scaled, resident=True)

    def memory_feature(self, i):
        """Memory feature i as an NCHW-shaped view [1,256,7,7] of its bank row."""
        if i == self.n - 1:
            self.flush()
        return self.bank[2 + i:3 + i].permute(0, 3, 1, 2)


class MemoryFeatures(object):
    """`state['memory_features']` on the fused path: the list the reference keeps
    (usot_tracker.py:264 appends one pooled [1,256,7,7] tensor per frame), backed by the session's
    device bank instead of growing a python list of tensors.  Indexing returns a view of the bank
    row (valid until the bank is regrown); `append` writes a feature computed outside the frame
    graph into the bank, so the generic `update()` branch can take over mid-video."""

    def __init__(self, session):
        self.session = session

    def __len__(self):
        return self.session.n

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(n))]
        i = int(i)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError('memory feature %d of %d' % (i, n))
        return self.session.memory_feature(i)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def append(self, feat):
        self.session.append_feature(feat)
