"""Diagnostic: per-stage error of the HIP path vs the fp32 CPU oracle and an fp64 oracle."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]
import usot_oracle as orc
from usot_amd import synth
from usot_amd.model import USOT

def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    e = np.abs(got - ref) / np.maximum(np.abs(ref), np.abs(ref).mean() + 1e-30)
    return e.max(), np.sqrt((e ** 2).mean())

FAMILY = sys.argv[1] if len(sys.argv) > 1 else 'zero_dc'
print('family', FAMILY)
m = USOT(); sd = synth.torch_state_dict(m, family=FAMILY); m.load_state_dict(sd); m.eval(); m = m.to('cuda:0')
m.engine_options['graphs'] = False
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
t = lambda a: torch.from_numpy(a)
for size, seed in ((255, 1), (127, 0)):
    x = t(synth.crop(seed, 1, size))
    with torch.no_grad():
        st32, p32 = orc.backbone(sd, x, stages=True); n32 = orc.neck(sd, p32)
        st64, p64 = orc.backbone(sd64, x.double(), stages=True); n64 = orc.neck(sd64, p64)
    stg, pg = m.feature_extractor(x.cuda()); ng = m.engine.features(x.cuda())
    names = ['stem', 'p1', 'p2', 'p3', 'neck']
    for nm, a32, a64, ag in zip(names, st32 + [p32, n32], st64 + [p64, n64], list(stg) + [pg, ng]):
        ag = ag.cpu().numpy()
        print('%d %-5s hip-vs-cpu32 max %.2e rms %.2e | hip-vs-f64 max %.2e rms %.2e | cpu32-vs-f64 max %.2e rms %.2e' % (
            (size, nm) + rel(ag, a32.numpy()) + rel(ag, a64.numpy()) + rel(a32.numpy(), a64.numpy())))
m.pr_pool = False
z, x = t(synth.crop(0, 1, 127)), t(synth.crop(1, 1, 255)); mem = t(synth.memory_kernels(7, 7))
with torch.no_grad():
    zf32 = orc.template(sd, z, pr_pool=False); o32 = orc.track(sd, x, zf32, mem, torch.ones(1, 7))
    zf64 = orc.template(sd64, z.double(), pr_pool=False); o64 = orc.track(sd64, x.double(), zf64, mem.double(), torch.ones(1, 7))
m.template(z.cuda()); og = m.track(x.cuda(), mem.cuda(), torch.ones(1, 7).cuda())
for nm, a32, a64, ag in zip(['cls', 'bbox', 'cls_mem', 'xf'], o32, o64, og):
    ag = ag.cpu().numpy()
    print('%-7s hip-vs-cpu32 max %.2e rms %.2e | hip-vs-f64 max %.2e rms %.2e | cpu32-vs-f64 max %.2e rms %.2e' % (
        (nm,) + rel(ag, a32.numpy()) + rel(ag, a64.numpy()) + rel(a32.numpy(), a64.numpy())))
