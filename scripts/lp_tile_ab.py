#!/usr/bin/env python3
"""Same-process A/B of low-precision tile choices inside the batch-64 backbone graph (the isolated tuner times a conv with its
operands warm).  python scripts/lp_tile_ab.py "61504,1024,4608=22" "61504,1024,4608=18" ...   (first configuration: the table)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from usot_amd import engine, synth
from usot_amd.model import USOT
dev = 'cuda:0'
base = dict(engine.LP_TUNING)
cfgs = [''] + sys.argv[1:]
B = int(os.environ.get('BATCH', '64'))         # BATCH=32: the per-GPU share of configs[4]
x = torch.from_numpy(synth.crop(1, B, 255)).to(dev)
plans = []
for rep in range(2):
    for c in cfgs:
        engine.LP_TUNING.clear(); engine.LP_TUNING.update(base)
        for kv in [v for v in c.split(';') if v]:
            k, t = kv.split('=')
            engine.LP_TUNING[tuple(int(v) for v in k.split(','))] = int(t)
        m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to(dev)
        for _ in range(3): m.engine.features_bf16(x)
        plans.append((c or 'table', m, next(v for k, v in m.engine._feat.items() if k[:3] == ('bf16', B, 255))['plan']))
for rnd in range(2):
    for c, m, plan in plans:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): plan.run()
        torch.cuda.synchronize()
        print('%-40s %.1f us/step' % (c, (time.perf_counter() - t0) / 100 * 1e6), flush=True)
