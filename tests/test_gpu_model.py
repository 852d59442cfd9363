"""GPU parity of the whole forward pass (template / track / extract_memory_feature behind
the reference's USOT API) against (a) goldens captured from the reference's own modules on
PyTorch-CPU and (b) the CPU oracle on other seeds.  Bar: 1e-4 scaled-relative, fp32
(BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import usot_oracle as orc  # noqa: E402
from sampling import check  # noqa: E402
from usot_amd import synth  # noqa: E402
from usot_amd.model import USOT  # noqa: E402

DEV = 'cuda:0'
TOL = 1e-4


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope='module', params=[True, False], ids=['graph', 'eager'])
def net(request):
    m = USOT()
    m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True)
    m.eval()
    m = m.to(DEV)
    m.engine_options['graphs'] = request.param
    return m


def npy(x):
    return x.detach().cpu().numpy()


@pytest.mark.parametrize('size,b,seed', [(127, 1, 0), (255, 1, 1), (271, 1, 3), (255, 2, 4)])
def test_backbone_stages_vs_reference_golden(net, gold_model, size, b, seed):
    stages, p3 = net.feature_extractor(t(synth.crop(seed, b, size)).to(DEV))
    tag = 'backbone_%d_b%d' % (size, b)
    for nm, ten in zip(('stem', 'p1', 'p2'), stages):
        check('%s/%s' % (tag, nm), gold_model, npy(ten), TOL)
    check(tag + '/p3', gold_model, npy(p3), TOL)
    xf = net.engine.features(t(synth.crop(seed, b, size)).to(DEV))
    check(tag + '/neck', gold_model, npy(xf), TOL)


def test_track_vs_reference_golden(net, gold_model):
    net.pr_pool = False
    net.template(t(synth.crop(0, 1, 127)).to(DEV))
    check('template_crop/zf', gold_model, npy(net.zf), TOL)
    x = t(synth.crop(1, 1, 255)).to(DEV)
    cls, bbox, a, b = net.track(x)
    assert a is None and b is None
    check('track_offline/cls', gold_model, npy(cls), TOL)
    check('track_offline/bbox', gold_model, npy(bbox), TOL)
    mem = t(synth.memory_kernels(7, 7)).to(DEV)
    cls, bbox, cm, xf = net.track(x, template_mem=mem, score_mem=torch.full((1, 7), 0.9, device=DEV))
    for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm), ('xf', xf)):
        check('track_mem/' + nm, gold_model, npy(ten), TOL)
    cls, bbox, cm, xf = net.track(t(synth.crop(3, 1, 271)).to(DEV), template_mem=mem,
                                  score_mem=torch.full((1, 7), 0.9, device=DEV))
    for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm)):
        check('track_mem_271/' + nm, gold_model, npy(ten), TOL)
    net.template(t(synth.crop(5, 2, 127)).to(DEV))
    cls, bbox, cm, xf = net.track(t(synth.crop(4, 2, 255)).to(DEV), template_mem=t(synth.memory_kernels(8, 14)).to(DEV),
                                  score_mem=torch.full((2, 7), 0.9, device=DEV))
    for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm)):
        check('track_mem_b2/' + nm, gold_model, npy(ten), TOL)
    net.pr_pool = True


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), np.abs(ref).mean() + 1e-30)))


@pytest.mark.parametrize('seed', [21, 22])
def test_track_with_prpool_vs_oracle(net, oracle_sd, seed):
    """PrPool-ed template + memory extracted from the search feature: the full a10-a12 path."""
    z, x = t(synth.crop(seed, 1, 127)), t(synth.crop(seed + 50, 1, 255))
    zbox = torch.tensor([[3.2, 4.1, 11.3, 10.6]])
    sbox = torch.tensor([[8.5, 9.25, 17.0, 16.5]])
    with torch.no_grad():
        zf = orc.template(oracle_sd, z, zbox, pr_pool=True)
        xf = orc.neck(oracle_sd, orc.backbone(oracle_sd, x))
        memf = orc.prpool_feature(xf, sbox)
        mem = torch.cat([memf] * 3 + [zf] * 4, 0)
        cls, bbox, cm, _ = orc.track(oracle_sd, x, zf, mem, torch.ones(1, 7))
    net.pr_pool = True
    net.template(z.to(DEV), template_bbox=zbox.to(DEV))
    assert rel(npy(net.zf), zf.numpy()) < TOL
    gm = net.extract_memory_feature(ori_x=x.to(DEV), search_bbox=sbox.to(DEV))
    assert tuple(gm.shape) == (1, 256, 7, 7)
    assert rel(npy(gm), memf.numpy()) < TOL
    gcls, gbbox, gcm, gxf = net.track(x.to(DEV), template_mem=mem.to(DEV), score_mem=torch.ones(1, 7, device=DEV))
    assert rel(npy(gxf), xf.numpy()) < TOL
    assert rel(npy(gcls), cls.numpy()) < TOL
    assert rel(npy(gbbox), bbox.numpy()) < TOL
    assert rel(npy(gcm), cm.numpy()) < TOL
    gm2 = net.extract_memory_feature(xf=gxf, search_bbox=sbox.to(DEV))
    assert rel(npy(gm2), memf.numpy()) < TOL


def test_cpu_model_refuses_to_compute():
    from usot_amd import hip
    m = USOT()
    with pytest.raises(hip.HipError):
        m.track(torch.zeros(1, 3, 255, 255))


def test_backbone_bf16_tracks_fp32(net, oracle_sd):
    """config 3 path: bf16 MFMA backbone + neck vs the fp32 oracle; bf16 keeps 8 mantissa bits,
    ~45 layers: a few percent of the feature scale, reported not gated at 1e-4."""
    x = t(synth.crop(40, 2, 255))
    with torch.no_grad():
        ref = orc.neck(oracle_sd, orc.backbone(oracle_sd, x)).numpy()
    got = net.engine.features_bf16(x.to(DEV)).float().cpu().numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    scale = np.abs(ref).mean()
    assert err.mean() / scale < 6e-2, err.mean() / scale      # measured 3-4e-2
    assert np.corrcoef(got.reshape(-1), ref.reshape(-1))[0, 1] > 0.999


@pytest.mark.parametrize('heads_lp', [False, True], ids=['heads_f32', 'heads_lp'])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
def test_track_mixed_precision(net, oracle_sd, dtype, heads_lp):
    """config 5 path: low-precision MFMA backbone (+ optionally the big head convs), fp32 xcorr,
    batch of 2 streams."""
    z, x = t(synth.crop(60, 2, 127)), t(synth.crop(61, 2, 255))
    mem = t(synth.memory_kernels(62, 14))
    with torch.no_grad():
        zf = orc.template(oracle_sd, z, pr_pool=False)
        cls, bbox, cm, xf = orc.track(oracle_sd, x, zf, mem, torch.ones(2, 7))
    net.pr_pool = False
    net.template(z.to(DEV))
    net.pr_pool = True
    gcls, gbbox, gcm, gxf = net.engine.track_mixed(x.to(DEV), net.zf, mem.to(DEV), torch.ones(2, 7, device=DEV), dtype=dtype,
                                                   heads_lp=heads_lp)
    tol = 3e-2 if dtype == torch.float16 else 1.5e-1           # fp16: 11 mantissa bits, bf16: 8
    for got, ref in ((gxf, xf), (gcls, cls), (gcm, cm)):
        g_, r_ = npy(got.float()), ref.numpy()
        assert np.abs(g_ - r_).mean() / np.abs(r_).mean() < tol
    lb = np.abs(np.log(npy(gbbox)) - np.log(bbox.numpy())).mean()
    assert lb < tol
