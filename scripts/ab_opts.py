#!/usr/bin/env python3
"""Same-process A/B of the tracked frame over engine options given as Python literals (dict / tuple valued options too):
    python scripts/ab_opts.py "{}" "{'batch_ds_conv2': {(961, 128, 1152): (55, 4)}}" ...
One Session per configuration, alternating timed runs of the driver's loop and of bare graph replays."""
import ast, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from usot_amd import engine
dev = torch.device('cuda:0')
base = dict(engine.OPTIONS)
sessions = []
for c in sys.argv[1:]:
    engine.OPTIONS.clear(); engine.OPTIONS.update(base); engine.OPTIONS.update(ast.literal_eval(c))
    try:
        model, _ = bench.build_model(0, 1, dev)
        sess, crops, p = bench.open_stream(model, dev, seed=0)
        conf = bench.Confidences()
        bench.run_frames(sess, crops, p, conf, 30)
        sessions.append((c, sess, crops, p, conf))
    except Exception as e:
        print('%-60s FAILED: %s' % (c, e), flush=True)
engine.OPTIONS.clear(); engine.OPTIONS.update(base)
for rnd in range(3):
    for c, sess, crops, p, conf in sessions:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.run_frames(sess, crops, p, conf, 1000)
        torch.cuda.synchronize(); loop = (time.perf_counter() - t0) / 1000 * 1e6
        t0 = time.perf_counter()
        for _ in range(500): sess.plan.run()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 500 * 1e6
        print('%-60s loop %.1f us/frame   graph %.1f us   launches %d' % (c[:60], loop, graph, len(sess.plan.profile(1))), flush=True)
