"""CPU: tracker host logic (grids, box->feature maps, decode, memory selection), crop
helpers and the module surface, against goldens captured from the reference's own
lib/tracker/usot_tracker.py (tests/golden/make_golden.py host) and the oracle restatement."""
import json
import os

import numpy as np
import pytest
import torch

import usot_oracle as orc
from usot_amd import hostutils
from usot_amd.tracker import USOTConfig, USOTTracker, select_memory

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Info:
    arch = 'USOT'


def cfg(inst):
    p = USOTConfig()
    p.instance_size = inst
    p.renew()
    p.sf_size = p.score_size
    return p


@pytest.mark.parametrize('inst', [255, 271])
def test_grids_and_pool_labels(gold_host, inst):
    p, trk = cfg(inst), USOTTracker(Info())
    trk.grids(p)
    tag = 'i%d' % inst
    assert np.array_equal(trk.grid_to_search_x, gold_host[tag + '/grid_x'])
    assert np.array_equal(trk.grid_to_search_y, gold_host[tag + '/grid_y'])
    assert np.array_equal(trk.search_area_x_axis, gold_host[tag + '/search_axis'])
    boxes = gold_host[tag + '/boxes']
    got_t = np.stack([trk.pool_label_template(p, b) for b in boxes])
    got_s = np.stack([trk.pool_label_search(p, b) for b in boxes])
    np.testing.assert_array_equal(got_t, gold_host[tag + '/pool_template'])
    np.testing.assert_array_equal(got_s, gold_host[tag + '/pool_search'])
    # oracle restatement agrees too
    op = orc.Hyper(inst)
    np.testing.assert_array_equal(np.stack([orc.pool_label_search(op, b) for b in boxes]), got_s)
    np.testing.assert_array_equal(np.stack([orc.pool_label_template(op, b) for b in boxes]), got_t)


@pytest.mark.parametrize('inst', [255, 271])
@pytest.mark.parametrize('case', range(4))
def test_update_decode_vs_reference(gold_host, inst, case):
    p, trk = cfg(inst), USOTTracker(Info())
    trk.grids(p)
    c = 'i%d/decode%d' % (inst, case)
    t = torch.from_numpy
    seen = {}

    class Net:
        def track(self, x, template_mem=None, score_mem=None):
            return t(gold_host[c + '/cls']), t(gold_host[c + '/bbox']), t(gold_host[c + '/cls_mem']), torch.zeros(1, 4, 31, 31)

        def extract_memory_feature(self, xf=None, search_bbox=None, ori_x=None):
            seen['box'] = search_bbox.numpy().copy()
            return torch.zeros(1, 4, 7, 7)
    S = p.score_size
    window = np.outer(np.hanning(S), np.hanning(S))
    tsz, sz = gold_host[c + '/tsz'], float(gold_host[c + '/scale_z'])
    pos, size, score, _ = trk.update(Net(), None, gold_host[c + '/tpos'].copy(), tsz * sz, window, sz, p)
    np.testing.assert_allclose(pos, gold_host[c + '/out_pos'], rtol=1e-12)
    np.testing.assert_allclose(size, gold_host[c + '/out_sz'], rtol=1e-12)
    assert score == gold_host[c + '/out_score']
    np.testing.assert_array_equal(seen['box'], gold_host[c + '/out_poolbox'])
    # the oracle's decode restates the same arithmetic
    opos, osz, oscore, obox, _ = orc.decode(orc.Hyper(inst), gold_host[c + '/cls'][0, 0], gold_host[c + '/cls_mem'][0, 0],
                                            gold_host[c + '/bbox'][0], gold_host[c + '/tpos'], tsz * sz, window, sz)
    np.testing.assert_allclose(opos, gold_host[c + '/out_pos'], rtol=1e-12)
    np.testing.assert_allclose(osz, gold_host[c + '/out_sz'], rtol=1e-12)


def test_memory_selection_vs_reference():
    with open(os.path.join(GOLD, 'golden_memory_indices.json')) as f:
        cases = json.load(f)
    assert set(cases) == {'1', '2', '3', '5', '8', '20', '57', '200'}
    for n, c in cases.items():
        picks = select_memory(c['conf'], 7)
        assert [-1.0, -2.0] + [float(i) for i in picks] == c['tags'], n
        assert orc.select_memory(orc.Hyper(), c['conf']) == picks
        want_scores = np.array([0.9, 0.9] + [c['conf'][i] for i in picks], np.float32)
        np.testing.assert_allclose(want_scores, np.array(c['score'], np.float32), rtol=1e-6)


def test_python2round_and_crop_geometry():
    assert hostutils.python2round(2.5) == 3.0 and hostutils.python2round(3.5) == 4
    assert hostutils.python2round(254.5) == 255.0 and hostutils.python2round(254.4) == 254
    im = (np.arange(60 * 80 * 3) % 251).astype(np.uint8).reshape(60, 80, 3)
    avg = np.mean(im, axis=(0, 1))
    # inside the image, no resize
    patch, info = hostutils.get_subwindow_tracking(im, np.array([40.0, 30.0]), 21, 21, avg, out_mode='raw')
    assert patch.shape == (21, 21, 3) and info['pad_info'][:2] == [0, 0]
    assert np.array_equal(patch, im[19:40, 29:50])
    # overlapping the border: mean-colour padding, truncated to uint8 like the reference's assignment
    patch, info = hostutils.get_subwindow_tracking(im, np.array([2.0, 3.0]), 21, 21, avg, out_mode='raw')
    top, left = info['pad_info'][:2]
    assert top > 0 and left > 0
    assert np.array_equal(patch[0, 0], avg.astype(np.uint8))
    assert np.array_equal(patch[top:, left:], im[:21 - top, :21 - left])
    # torch output is CHW float32 with raw 0..255 values
    tpatch, _ = hostutils.get_subwindow_tracking(im, np.array([40.0, 30.0]), 21, 21, avg)
    assert tpatch.dtype == torch.float32 and tuple(tpatch.shape) == (3, 21, 21)
    assert torch.equal(tpatch, torch.from_numpy(im[19:40, 29:50].transpose(2, 0, 1).astype(np.float32)))


def test_resize_builtin_properties():
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (37, 41, 3), dtype=np.uint8)
    assert np.array_equal(hostutils.resize_bilinear_u8(img, 41, 37), img)
    const = np.full((30, 30, 3), 77, np.uint8)
    assert np.all(hostutils.resize_bilinear_u8(const, 55, 47) == 77)
    up = hostutils.resize_bilinear_u8(img, 82, 74)
    assert up.shape == (74, 82, 3) and up.dtype == np.uint8
    ramp = np.tile(np.arange(64, dtype=np.uint8)[None, :, None] * 4, (8, 1, 3))
    out = hostutils.resize_bilinear_u8(ramp, 128, 8).astype(int)
    assert np.all(np.diff(out[0, :, 0]) >= 0)          # monotone ramp stays monotone
    # exact 2x downscale: cv2.resize(INTER_LINEAR) silently uses INTER_AREA = rounded 2x2 block means
    big = g.integers(0, 256, (60, 44, 3), dtype=np.uint8)
    half = hostutils.resize_bilinear_u8(big, 22, 30)
    q = big.astype(int)
    assert np.array_equal(half, ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    # one direction only is NOT the area path (OpenCV needs iscale_x == iscale_y == 2)
    assert hostutils.resize_bilinear_u8(big, 22, 31).shape == (31, 22, 3)
    # coefficients: float source coordinate, round-half-even, every pair sums to 2048
    for n_src, n_dst in ((301, 255), (188, 255), (612, 255), (127, 255), (509, 255)):
        s0, s1, w0, w1 = hostutils._resize_axis(n_src, n_dst)
        assert np.all(w0 + w1 == 2048) and s0.min() >= 0 and s1.max() == n_src - 1 and np.all(s1 - s0 <= 1)


def test_flip_matches_imgaug_convention():
    img = np.arange(2 * 5 * 3, dtype=np.uint8).reshape(2, 5, 3)
    f, box = hostutils.flip_lr(img, [1.0, 0.5, 3.5, 1.5])
    assert np.array_equal(f, img[:, ::-1]) and box == [1.5, 0.5, 4.0, 1.5]


def test_module_surface_of_the_reference():
    import lib.models.models as models
    from lib.tracker.usot_tracker import USOTConfig as C2, USOTTracker as T2
    from lib.utils.train_utils import load_pretrain  # noqa: F401
    from lib.utils.test_utils import cxy_wh_2_rect, get_axis_aligned_bbox, poly_iou  # noqa: F401
    from lib.dataset_loader.benchmark import load_dataset  # noqa: F401
    from lib.utils.track_utils import load_yaml, get_subwindow_tracking, python2round, im_to_torch  # noqa: F401
    from lib.models.connect import xcorr_depthwise, AdjustLayer, box_tower_reg  # noqa: F401
    from lib.models.prroi_pool import PrRoIPool2D  # noqa: F401
    net = models.__dict__['USOT']()
    assert isinstance(net, torch.nn.Module) and net.pr_pool is True and net.zf is None
    with open(os.path.join(GOLD, 'state_dict_keys.json')) as f:
        ref = json.load(f)
    sd = net.state_dict()
    assert set(sd) == set(ref) and all(list(sd[k].shape) == ref[k] for k in ref)
    assert sum(p.numel() for p in net.parameters()) == 29414993
    assert C2().score_size == 25 and T2 is USOTTracker
    here = os.path.dirname(os.path.abspath(__file__))
    hp = load_yaml(os.path.join(here, '..', 'experiments', 'test', 'USOT.yaml'))
    assert hp == {'penalty_k': 0.021, 'lr': 0.73, 'window_influence': 0.321, 'small_sz': 255, 'big_sz': 271,
                  'ratio': 0.3, 'mem_queue_size': 7}


def test_load_pretrain_prefixes_and_moco(tmp_path):
    import lib.models.models as models
    from lib.utils.train_utils import load_pretrain
    from usot_amd import synth
    net = models.USOT()
    sd = synth.torch_state_dict(net, seed=0, calibrated=True)
    path = str(tmp_path / 'ckpt.pth')
    torch.save({'state_dict': {'module.' + k: v for k, v in sd.items()}, 'epoch': 3}, path)
    out = load_pretrain(models.USOT(), path)
    for k, v in out.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # MoCo-style backbone: encoder_q.* keys, 1x1 shortcuts widened to centred 3x3
    moco = {}
    for k, v in sd.items():
        if k.startswith('features.features.'):
            kk = k.replace('features.features', 'encoder_q')
            if kk in ('encoder_q.layer2.0.downsample.0.weight', 'encoder_q.layer3.0.downsample.0.weight'):
                v = v[:, :, 1:2, 1:2].clone()
            moco[kk] = v
    mpath = str(tmp_path / 'moco_v2.pth')
    torch.save(moco, mpath)
    out = load_pretrain(models.USOT(), mpath)
    w = out.state_dict()['features.features.layer2.0.downsample.0.weight']
    assert torch.equal(w[:, :, 1, 1], sd['features.features.layer2.0.downsample.0.weight'][:, :, 1, 1])
    assert float(w[:, :, 0, 0].abs().max()) == 0.0
    with pytest.raises(AssertionError):
        bad = str(tmp_path / 'bad.pth')
        torch.save({'nothing': torch.zeros(1)}, bad)
        load_pretrain(models.USOT(), bad)


def test_poly_iou_and_rect_helpers():
    from lib.utils.test_utils import cxy_wh_2_rect, get_axis_aligned_bbox, poly_iou
    assert abs(float(poly_iou(np.array([0, 0, 10, 10.]), np.array([5, 5, 10, 10.]))[0]) - 25 / 175) < 1e-12
    assert abs(float(poly_iou(np.array([5, 0, 10, 5, 5, 10, 0, 5.]), np.array([0, 0, 10, 10.]))[0]) - 0.5) < 1e-12
    assert float(poly_iou(np.array([0, 0, 1, 1.]), np.array([5, 5, 1, 1.]))[0]) == 0.0
    assert cxy_wh_2_rect(np.array([10.0, 8.0]), np.array([4.0, 6.0])) == [8.0, 5.0, 4.0, 6.0]
    cx, cy, w, h = get_axis_aligned_bbox(np.array([2.0, 3.0, 10.0, 20.0]))
    assert (cx, cy, w, h) == (7.0, 13.0, 10.0, 20.0)


def test_stem_fragment_packing_is_a_plain_convolution():
    """engine.pack_stem_f32 / pack_stem_lp lay the 7x7x3 stem filters out as MFMA A operands over a padded
    k axis (24 rows x 8 taps).  Summing fragment x patch element exactly as the kernels index them must give
    the ordinary stride-2 convolution (host-only check of the layout contract with csrc/stem_pool.hip and
    csrc/conv_bf16.hip)."""
    import torch
    import torch.nn.functional as F
    from usot_amd.engine import pack_stem_f32, pack_stem_lp
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 3, 7, 7, generator=g)
    x = torch.randn(1, 3, 11, 13, generator=g)
    packed = w.permute(1, 2, 3, 0).reshape(147, 64).contiguous()          # Weights.stem_w layout
    ref = F.conv2d(x, w, stride=2)[0]                                     # [64, 3, 4]
    xp = F.pad(x[0], (0, 8))                                              # the zero tap reads one column past
    f32 = pack_stem_f32(packed)                                           # [4][48][64]
    lp = pack_stem_lp(packed, torch.float32)                              # [4][6][64][8]
    for (sy, sx) in ((0, 0), (2, 3), (1, 2)):
        for co in (0, 17, 63):
            cb, l15 = divmod(co, 16)
            acc = 0.0
            for st in range(48):                                          # fp32 MFMA: step st, quad q -> row st//2, tap (st%2)*4+q
                row = min(st // 2, 20)
                ci, kh = divmod(row, 7)
                for q in range(4):
                    acc += float(f32[cb, st, q * 16 + l15]) * float(xp[ci, 2 * sy + kh, 2 * sx + (st % 2) * 4 + q])
            assert abs(acc - float(ref[co, sy, sx])) < 1e-4 * max(1.0, abs(float(ref[co, sy, sx])))
            acc = 0.0
            for ks in range(6):                                           # 16x16x32 MFMA: step ks, quad q -> row 4*ks+q, 8 taps
                for q in range(4):
                    row = min(4 * ks + q, 20)
                    ci, kh = divmod(row, 7)
                    for t8 in range(8):
                        acc += float(lp[cb, ks, q * 16 + l15, t8]) * float(xp[ci, 2 * sy + kh, 2 * sx + t8])
            assert abs(acc - float(ref[co, sy, sx])) < 1e-4 * max(1.0, abs(float(ref[co, sy, sx])))


def test_conf_tail_split_option_is_validated():
    """engine.merged_options refuses a (tail, ksplit) pair that would build an empty or unsplit tail convolution."""
    from usot_amd import engine, hip
    assert engine.merged_options({'conf_tail_split': None})['conf_tail_split'] is None
    assert engine.merged_options({'conf_tail_split': (2, 3)})['conf_tail_split'] == (2, 3)
    for bad in ((0, 2), (1, 1), (1, 0), (1,), 'x', (1.5, 2), (True, 2)):
        with pytest.raises(hip.HipError):
            engine.merged_options({'conf_tail_split': bad})


def test_raw_pixel_range_check_guards_the_fp16_stem():
    """The bf16 backbone's fp16-arithmetic stem is adequate for raw 0..255 crops only (engine.looks_like_raw_pixels)."""
    import torch
    from usot_amd import engine, synth
    assert engine.looks_like_raw_pixels(torch.from_numpy(synth.crop(1, 1, 127)))
    assert not engine.looks_like_raw_pixels(torch.rand(1, 3, 127, 127))                 # [0, 1]-normalised
    assert not engine.looks_like_raw_pixels(torch.randn(1, 3, 127, 127) * 50)           # mean/std-normalised, signed


def test_split16_pack_reconstructs_the_bank_to_22_bits():
    """usot_amd/hip.py: split16_pack / pw_pair_s16_pack (host side of the split-fp16 conv tiles and of usot_pw_pair_f32s): every
    filter as hi + lo fp16 of (row x a power of two).  (hi + lo) x the stored factor x 8 must give the row back to 2^-21 of the
    row's largest value (22 significant bits relative to the scaled row), whatever the row's magnitude - including all-zero rows,
    rows of 1e-20 and of 1e+4 - and the fragment order must be the one the kernels index."""
    import torch
    from usot_amd import hip
    g = torch.Generator().manual_seed(3)
    rows, k = 48, 192
    w = torch.randn(rows, k, generator=g)
    w[1] = 0.0
    w[2] *= 1e-20
    w[3] *= 1e4
    w[4, ::2] *= 1e-6                      # a wide dynamic range inside one row
    packed, inv = hip.split16_pack(w)
    assert packed.shape == (rows, k) and packed.dtype == torch.float32 and inv.shape == (rows,)
    halves = packed.view(torch.float16).view(rows, k // 64, 2, 64).float()         # [row][k-tile][hi | lo][64]
    rec = (halves[:, :, 0] + halves[:, :, 1]).reshape(rows, k) * (inv * hip.SPLIT16_X_SCALE)[:, None]
    amax = w.abs().amax(1, keepdim=True)
    assert torch.isfinite(halves).all()
    assert ((rec - w).abs() <= amax * 2.0 ** -21).all()
    assert (halves[:, :, 0].abs().amax((1, 2))[amax[:, 0] > 0] >= 512).all() and halves.abs().max() <= 1024      # rows sit at the top of fp16's range
    scale = 1.0 / (inv * hip.SPLIT16_X_SCALE)
    assert torch.equal(torch.exp2(torch.round(torch.log2(scale))), scale)            # exact powers of two
    # the pair kernel's fragment order: (column block of 16 rows, 32-k step, hi | lo, quad, row in block, 8 halves), factors appended
    pp = hip.pw_pair_s16_pack(w)
    assert pp.numel() == rows * k + rows
    fr = pp[:rows * k].view(torch.float16).view(rows // 16, k // 32, 2, 4, 16, 8).float()
    rec2 = (fr[:, :, 0] + fr[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(rows, k) * (pp[rows * k:] * hip.SPLIT16_X_SCALE)[:, None]
    assert ((rec2 - w).abs() <= amax * 2.0 ** -21).all()


def _bilinear_f64(img, dw, dh):
    """Exact (float64) bilinear resampling on half-pixel centres with edge clamping, by torch - an implementation that shares
    nothing with hostutils.resize_bilinear_u8."""
    import torch.nn.functional as F
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    return F.interpolate(t, size=(dh, dw), mode='bilinear', align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()


def test_resize_against_two_independent_bilinear_implementations():
    """`cv2` is not in this image, so hostutils.resize_bilinear_u8 (the restatement of cv2.resize's fixed-point INTER_LINEAR,
    track_utils.py:77-78) cannot be pinned bit for bit.  What CAN be pinned is its sampling geometry and weights:
    against torch's float64 bilinear interpolation the uint8 result must be the rounded exact value up to the fixed-point
    scheme's own error (0.5 of rounding + 11-bit weights and the >> 4 / >> 16 truncations: < 0.9 LSB, mean 0.25-0.3), and against
    Pillow's affine bilinear sampler (another fixed-point implementation) within 1 LSB - at the window sizes the tracker
    produces (s_x of 127 ... 612 pixels resized to 255 / 127), on noise images (worst case for interpolation error)."""
    from PIL import Image
    g = np.random.default_rng(3)
    for h, w, dh, dw in ((301, 301, 255, 255), (188, 188, 255, 255), (612, 612, 255, 255), (127, 127, 255, 255),
                         (97, 131, 127, 127), (509, 509, 255, 255), (260, 260, 271, 271)):
        img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = hostutils.resize_bilinear_u8(img, dw, dh).astype(np.float64)
        exact = _bilinear_f64(img, dw, dh)
        d = np.abs(got - exact)
        assert d.max() < 0.9 and d.mean() < 0.3, (h, w, d.max(), d.mean())
        assert -0.2 < (got - exact).mean() < 0.02                        # the >> 4 / >> 16 truncations of the scheme bias it by -1/8 LSB, nothing more
        pil = np.asarray(Image.fromarray(img).transform((dw, dh), Image.AFFINE, (w / dw, 0, 0, 0, h / dh, 0),
                                                        resample=Image.BILINEAR)).astype(np.float64)
        assert np.abs(got - pil).max() <= 1.0, (h, w)
    # a shifted or align_corners=True geometry is NOT within that bound (the check has teeth)
    img = g.integers(0, 256, (301, 301, 3), dtype=np.uint8)
    import torch.nn.functional as F
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    wrong = F.interpolate(t, size=(255, 255), mode='bilinear', align_corners=True)[0].permute(1, 2, 0).numpy()
    assert np.abs(hostutils.resize_bilinear_u8(img, 255, 255) - wrong).max() > 20


def test_crop_against_an_independent_restatement_of_the_reference_crop():
    """track_utils.py:30-119 restated with numpy padding + the float64 bilinear above (no hostutils code): mean-colour canvas,
    integer window, resize to the model size.  hostutils.get_subwindow_tracking must agree within the fixed-point bound; the
    device crop kernel is bit-compared with hostutils on the GPU (tests/test_gpu_ops.py::test_device_crop_matches_host_crop)."""
    g = np.random.default_rng(5)
    im = g.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    avg = np.mean(im, axis=(0, 1))
    for pos, win, size in (((160.0, 120.0), 301, 255), ((5.0, 7.0), 188, 255), ((318.0, 236.0), 401, 255), ((100.5, 60.5), 127, 127),
                           ((30.0, 200.0), 90, 127)):
        c = (win + 1) / 2
        x0, y0 = round(pos[0] - c), round(pos[1] - c)                          # Python 3 round (half to even), track_utils.py:44-46
        canvas = np.empty((240 + 2 * 640, 320 + 2 * 640, 3), np.uint8)
        canvas[:] = avg.astype(np.uint8)                                     # the reference assigns float means into a uint8 array
        canvas[640:640 + 240, 640:640 + 320] = im
        patch = canvas[640 + y0:640 + y0 + win, 640 + x0:640 + x0 + win]
        want = patch.astype(np.float64) if win == size else _bilinear_f64(patch, size, size)
        got, _ = hostutils.get_subwindow_tracking(im, np.array(pos), size, win, avg, out_mode='raw')
        assert got.shape == (size, size, 3)
        d = np.abs(got.astype(np.float64) - want)
        assert d.max() < (0.9 if win != size else 1e-12), (pos, win, d.max())
