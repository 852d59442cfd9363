#!/usr/bin/env python3
"""Same-process A/B of the batch-64 low-precision backbone step under ENGINE OPTIONS:
    python scripts/lp_chain_ab.py base lp_chains=2 lp_chains=2+lp_chain_skew=3 lp_chains=4
Each argument is '+'-joined 'option=python-literal' pairs on top of the defaults ('base' = none).  Prints the graph replay time of
features_bf16 per configuration over three alternating rounds and whether its output equals the first configuration's bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from usot_amd import engine, synth
dev = torch.device('cuda:0')
B = int(os.environ.get('LP_BATCH', '64'))
dt = torch.float16 if os.environ.get('LP_DTYPE', 'bf16') == 'f16' else torch.bfloat16
cfgs = sys.argv[1:] or ['base']
x = torch.from_numpy(synth.crop(1, B, 255)).to(dev)
runs = []
for c in cfgs:
    opts = {}
    if c != 'base':
        for ent in c.split('+'):
            k, v = ent.split('=')
            opts[k] = eval(v)
    model, _ = bench.build_model(0, 1, dev)
    model.engine_options['options'] = opts
    e = model.engine
    for _ in range(3):
        y = e.features_bf16(x, dtype=dt)
    torch.cuda.synchronize()
    plan = next(v for k, v in e._feat.items() if k[1] == B)['plan']
    runs.append((c, e, plan, y.clone()))
ref = runs[0][3]
for c, e, plan, y in runs:
    d = (y.float() - ref.float()).abs().max().item()
    print('%-50s output == first: %s (max |diff| %.3g)' % (c, bool(torch.equal(y, ref)), d), flush=True)
for rnd in range(3):
    for c, e, plan, y in runs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): plan.run()
        torch.cuda.synchronize(); g = (time.perf_counter() - t0) / 100 * 1e6
        print('%-50s graph %.1f us  (%.0f crops/s)' % (c, g, B / g * 1e6), flush=True)
