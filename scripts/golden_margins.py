#!/usr/bin/env python3
"""Scaled errors of the end-to-end golden checks (tests/test_gpu_model.py::test_track_vs_reference_golden), sorted:
how far each output sits from the 1e-4 bar.  USOT_NO_FUSED_F32=1 runs without the fused fp32 pointwise pairs,
USOT_NO_SLICE=1 with them but unsliced."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import sampling
from usot_amd import synth, engine
from usot_amd.model import USOT
from usot_amd import hip
if os.environ.get('USOT_NO_FUSED_F32'):
    engine.OPTIONS['fused_pointwise_f32'] = set()
if os.environ.get('USOT_SLICED'):            # channel-sliced pairs on (layer2 sliced, layer3's pairs fused)
    engine.OPTIONS['fused_f32_sliced'] = True
if os.environ.get('USOT_NO_SLICE'):          # fused pairs without the channel-sliced form
    hip.pw_pair_f32_ws = lambda *a, **k: None
gold = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_model.npz')))
errs = []
def check(name, arr):
    a = np.asarray(arr, dtype=np.float32)
    if name + '/full' in gold: ref, got = gold[name + '/full'], a
    else: ref, got = gold[name + '/samp'], a.reshape(-1)[sampling.sample_index(name, a.size)]
    scale = np.maximum(np.abs(ref), np.abs(ref).mean() + 1e-30)
    errs.append((float(np.max(np.abs(got - ref) / scale)), name))
DEV = 'cuda:0'
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
npy = lambda x: x.detach().cpu().numpy()
net = USOT(); net.load_state_dict(synth.torch_state_dict(net, seed=0, calibrated=True), strict=True); net.eval(); net = net.to(DEV)
net.pr_pool = False
net.template(t(synth.crop(0, 1, 127)).to(DEV)); check('template_crop/zf', npy(net.zf))
x = t(synth.crop(1, 1, 255)).to(DEV)
cls, bbox, _, _ = net.track(x); check('track_offline/cls', npy(cls)); check('track_offline/bbox', npy(bbox))
mem = t(synth.memory_kernels(7, 7)).to(DEV)
for tag, xx in (('track_mem', x), ('track_mem_271', t(synth.crop(3, 1, 271)).to(DEV))):
    cls, bbox, cm, xf = net.track(xx, template_mem=mem, score_mem=torch.full((1, 7), 0.9, device=DEV))
    for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm)): check(tag + '/' + nm, npy(ten))
net.template(t(synth.crop(5, 2, 127)).to(DEV))
cls, bbox, cm, xf = net.track(t(synth.crop(4, 2, 255)).to(DEV), template_mem=t(synth.memory_kernels(8, 14)).to(DEV), score_mem=torch.full((2, 7), 0.9, device=DEV))
for nm, ten in (('cls', cls), ('bbox', bbox), ('cls_mem', cm)): check('track_mem_b2/' + nm, npy(ten))
for size, b, seed in ((127, 1, 0), (255, 1, 1), (271, 1, 3), (255, 2, 4)):
    stages, p3 = net.feature_extractor(t(synth.crop(seed, b, size)).to(DEV))
    check('backbone_%d_b%d/p3' % (size, b), npy(p3))
for e, n in sorted(errs, reverse=True):
    print('%.3e  %s' % (e, n))
