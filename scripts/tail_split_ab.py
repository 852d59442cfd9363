#!/usr/bin/env python3
"""Same-process A/B of whole-chip-round variants of the frame's multi-round launches: builds one Session per configuration
and alternates timed replays of the frame graph.  Configurations: engine.OPTIONS overrides + an optional tile for the
search-encoder launch's lead shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from usot_amd import engine
dev = torch.device('cuda:0')
base_tuning = engine.load_tuning()
# every configuration is built THREE times: sessions of the same configuration differ by up to ~10 us with where their
# buffers land, so a single instance per configuration cannot rank variants that are closer than that
cfgs = [({}, None), ({}, {(625, 256, 2304, 3): (55, 2)}), ({}, {(625, 256, 2304, 3): (53, 2)}), ({}, {(961, 256, 2304, 1): (54, 2)}),
        ({}, {(961, 256, 2304, 1): (55, 2)}), ({}, {(961, 1024, 4608, 1): (55, 2)})] * 3
sessions = []
for opts, tile in cfgs:
    saved = dict(engine.OPTIONS)
    engine.OPTIONS.update(opts)
    tun = dict(base_tuning)
    if tile:
        tun.update(tile)
    model, _ = bench.build_model(0, 1, dev)
    model.engine_options['tuning'] = tun
    sess, crops, p = bench.open_stream(model, dev, seed=0)
    conf = bench.Confidences()
    bench.run_frames(sess, crops, p, conf, 30)
    sessions.append(((opts, tile), sess))
    engine.OPTIONS.clear(); engine.OPTIONS.update(saved)
for rnd in range(2):
    for c, sess in sessions:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): sess.plan.run()
        torch.cuda.synchronize()
        print('%-60s graph %.1f us' % (c, (time.perf_counter() - t0) / 500 * 1e6), flush=True)
