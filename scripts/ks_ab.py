#!/usr/bin/env python3
"""Same-process A/B of the tracked frame with TUNING-TABLE overrides (tile / split-K per conv shape):
    python scripts/ks_ab.py base  625,256,2304,3=55:2  961,256,2304,1=53:2+625,256,2304,3=55:2
Each argument is one configuration: '+'-joined 'M,N,K,g=tile:ksplit' entries on top of the shipped table ('base' = none).
Prints graph replay and loop time per configuration over three alternating rounds, then the per-op spans of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from usot_amd import engine, hip
dev = torch.device('cuda:0')
cfgs = sys.argv[1:] or ['base']
base = engine.load_tuning()
sessions = []
for c in cfgs:
    tab = dict(base)
    opts = {}
    if c != 'base':
        for ent in c.split('+'):
            k, v = ent.split('=')
            if ',' in k:
                t, ks = v.split(':')
                tab[tuple(int(x) for x in k.split(','))] = (int(t), int(ks))
            else:
                opts[k] = eval(v)
    engine.load_tuning = lambda path=None, tab=tab: tab
    saved = dict(engine.OPTIONS)
    engine.OPTIONS.update(opts)
    model, _ = bench.build_model(0, 1, dev)
    sess, crops, p = bench.open_stream(model, dev, seed=0)
    engine.OPTIONS.clear(); engine.OPTIONS.update(saved)
    conf = bench.Confidences()
    bench.run_frames(sess, crops, p, conf, 30)
    sessions.append((c, sess, crops, p, conf))
for rnd in range(3):
    for c, sess, crops, p, conf in sessions:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.run_frames(sess, crops, p, conf, 600)
        torch.cuda.synchronize(); loop = (time.perf_counter() - t0) / 600 * 1e6
        t0 = time.perf_counter()
        for _ in range(400): sess.plan.run()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 400 * 1e6
        print('%-60s loop %.1f us/frame   graph %.1f us' % (c, loop, graph), flush=True)
if os.environ.get('KS_AB_OPS', '1') == '1':
    for c, sess, crops, p, conf in sessions:
        prof = sess.plan.profile(20)
        convs = iter(sess.log)
        print('---- per-op spans:', c)
        for kind, tile, ks, groups, ms in prof:
            name = str(kind)
            if kind in (0, 18, 19, 20, 21):
                nm, M, N, K, g, macs = next(convs)
                name = '%s M=%d N=%d K=%d g=%d tile=%s ks=%d  %.1f TFLOP/s' % (nm, M, N, K, g, hip.tile_name(tile) if tile else '-', ks, 2e-9 * macs / ms)
            if ms * 1e3 >= 4.0:
                print('%8.1f us  %s' % (ms * 1e3, name))
