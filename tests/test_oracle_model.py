"""CPU: the oracle restatement reproduces the goldens captured from the reference's own
lib/models on PyTorch-CPU (tests/golden/make_golden.py).  Tolerance 2e-5 scaled error:
same torch ops, only thread-count / reassociation differences between machines."""
import numpy as np
import pytest
import torch

import usot_oracle as orc
from sampling import check
from usot_amd import synth

TOL = 2e-5


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize('size,b,seed', [(127, 1, 0), (255, 1, 1), (271, 1, 3), (255, 2, 4)])
def test_backbone_and_neck(gold_model, oracle_sd, size, b, seed):
    with torch.no_grad():
        stages, p3 = orc.backbone(oracle_sd, t(synth.crop(seed, b, size)), stages=True)
        nk = orc.neck(oracle_sd, p3)
    tag = 'backbone_%d_b%d' % (size, b)
    for nm, ten in zip(('stem', 'p1', 'p2'), stages):
        check('%s/%s' % (tag, nm), gold_model, ten.numpy(), TOL)
    check(tag + '/p3', gold_model, p3.numpy(), TOL)
    check(tag + '/neck', gold_model, nk.numpy(), TOL)


def test_track_paths(gold_model, oracle_sd):
    with torch.no_grad():
        zf = orc.template(oracle_sd, t(synth.crop(0, 1, 127)), pr_pool=False)
        check('template_crop/zf', gold_model, zf.numpy(), TOL)
        x = t(synth.crop(1, 1, 255))
        cls, bbox, a, b = orc.track(oracle_sd, x, zf)
        assert a is None and b is None
        check('track_offline/cls', gold_model, cls.numpy(), TOL)
        check('track_offline/bbox', gold_model, bbox.numpy(), TOL)
        mem = t(synth.memory_kernels(7, 7))
        cls, bbox, cm, xf = orc.track(oracle_sd, x, zf, mem, torch.full((1, 7), 0.9))
        for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm), ('xf', xf)):
            check('track_mem/' + nm, gold_model, ten.numpy(), TOL)
        cls, bbox, cm, xf = orc.track(oracle_sd, t(synth.crop(3, 1, 271)), zf, mem, torch.full((1, 7), 0.9))
        for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm)):
            check('track_mem_271/' + nm, gold_model, ten.numpy(), TOL)
        zf2 = orc.template(oracle_sd, t(synth.crop(5, 2, 127)), pr_pool=False)
        cls, bbox, cm, xf = orc.track(oracle_sd, t(synth.crop(4, 2, 255)), zf2,
                                      t(synth.memory_kernels(8, 14)), torch.full((2, 7), 0.9))
        for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm)):
            check('track_mem_b2/' + nm, gold_model, ten.numpy(), TOL)


def test_head_pieces(gold_model, oracle_sd):
    sd = oracle_sd
    with torch.no_grad():
        xf = t(synth.memory_kernels(20, 1, 256, 31))
        zk = t(synth.memory_kernels(21, 1))
        mk = t(synth.memory_kernels(22, 7))
        cx, cz = orc.encode(sd, 'cls_encode', xf, 's'), orc.encode(sd, 'cls_encode', zk, 'k')
        rx, rz = orc.encode(sd, 'reg_encode', xf, 's'), orc.encode(sd, 'reg_encode', zk, 'k')
        for i, nm in enumerate(('11', '12', '21')):
            check('enc/cls_s' + nm, gold_model, cx[i].numpy(), TOL)
            check('enc/cls_k' + nm, gold_model, cz[i].numpy(), TOL)
            check('enc/reg_s' + nm, gold_model, rx[i].numpy(), TOL)
        check('groupdw/cls', gold_model, orc.groupdw(sd, 'cls_dw', cz, cx).numpy(), TOL)
        check('groupdw/reg', gold_model, orc.groupdw(sd, 'reg_dw', rz, rx).numpy(), TOL)
        cls_mem, dwm, fused = orc.head_memory(sd, cx, mk, 1, 7)
        check('groupdw/mem', gold_model, dwm.reshape(7, 256, 25, 25).numpy(), TOL)
        check('conf_fusion/out', gold_model, fused.numpy(), TOL)
        check('tower/bbox', gold_model, orc.tower(sd, 'bbox_tower', dwm[0, :1]).numpy(), TOL)


@pytest.mark.parametrize('i', range(5))
def test_xcorr_depthwise(gold_model, i):
    out = orc.xcorr_depthwise(t(gold_model['xcorr%d/x' % i]), t(gold_model['xcorr%d/k' % i]))
    np.testing.assert_allclose(out.numpy(), gold_model['xcorr%d/out' % i], rtol=1e-5, atol=1e-5)
