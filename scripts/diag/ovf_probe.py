"""Which split-fp16 launch of the features plan reports a not-finite sum on ordinary crops, and how large are the activations?"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from usot_amd import synth, hip
from usot_amd.model import USOT
from usot_amd import engine as E

dev = 'cuda:0'
def make(split):
    m = USOT(); m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True); m.eval(); m = m.to(dev)
    if split: m.engine_options['options'] = {'split16_f32': True}
    return m
me, ms = make(False), make(True)
t = lambda a: torch.from_numpy(a).to(dev)
for name, x in (('z127', t(synth.crop(1000, 1, 127))), ('x255', t(synth.crop(2000, 8, 255))[0:1]), ('x255flip', t(synth.crop(2000, 8, 255))[0:1].flip(3))):
    stages, p3 = me.feature_extractor(x)
    print(name, 'exact stage maxima', [float(s.abs().max()) for s in stages], float(p3.abs().max()), 'neck', float(me.engine.features(x).abs().max()))
    # the split plan, op by op: build it eagerly with one ovf word per conv
    e = ms.engine
    bld = E.Builder(e.W, e.tuning, 0, dict(e.opt))
    xin = bld.buf(1, 3, x.shape[2], x.shape[2]); xin.copy_(x)
    words = []
    orig = bld.ovf_word
    def fresh():
        w = torch.zeros(1, device=dev, dtype=torch.int32); bld.plan.keep.append(w); words.append((len(bld.log), w)); return w
    bld.ovf_word = fresh
    xf, h = bld.backbone(xin, 1, x.shape[2])
    bld.plan.run(); torch.cuda.synchronize()
    for idx, w in words:
        nm = bld.log[idx][0] if idx < len(bld.log) else '?'
        print('   ovf', int(w.item()), 'conv#%d' % idx, nm, bld.log[idx][1:5] if idx < len(bld.log) else '')
    print('   split-vs-exact neck max abs diff', float((xf.permute(0, 3, 1, 2) - me.engine.features(x)).abs().max()), 'finite', bool(torch.isfinite(xf).all()))
