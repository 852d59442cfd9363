#!/usr/bin/env python3
"""Parity beyond the fixed test seeds: HIP frame vs the CPU oracle on N random crops / memory sets; prints the
scaled error (max |got-ref| / max(|ref|, mean|ref|), the tests' metric) per output and the worst case."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import usot_oracle as orc
from usot_amd import synth
from usot_amd.model import USOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = USOT(); sd = synth.torch_state_dict(m, seed=0, calibrated=True); m.load_state_dict(sd); m.eval(); m = m.cuda()
osd = {k: v.float() for k, v in sd.items()}
def scaled(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), np.abs(b).mean()))
worst = {}
for s in range(n):
    t = torch.from_numpy
    z, x = t(synth.crop(900 + s, 1, 127)), t(synth.crop(1900 + s, 1, 255)); mem = t(synth.memory_kernels(2900 + s, 7))
    with torch.no_grad():
        zf = orc.template(osd, z, pr_pool=False)
        cls, bbox, cm, xf = orc.track(osd, x, zf, mem, torch.ones(1, 7))
    m.pr_pool = False; m.template(z.cuda()); m.pr_pool = True
    g = m.track(x.cuda(), mem.cuda(), torch.ones(1, 7).cuda())
    errs = dict(cls=scaled(g[0].cpu(), cls), bbox=scaled(g[1].cpu(), bbox), cls_mem=scaled(g[2].cpu(), cm), feat=scaled(g[3].cpu(), xf))
    for k, v in errs.items(): worst[k] = max(worst.get(k, 0.0), v)
    print(s, ' '.join('%s %.2e' % kv for kv in errs.items()), flush=True)
print('worst', ' '.join('%s %.2e' % kv for kv in worst.items()), '(bar 1e-4)')
