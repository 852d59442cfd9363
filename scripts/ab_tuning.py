#!/usr/bin/env python3
"""Same-process A/B of the tracked frame over (tile, ksplit) overrides of the tuning table, given as Python literals keyed
(M, Cout, K, groups):   python scripts/ab_tuning.py "{}" "{(625, 256, 2304, 3): (55, 2)}" ...
One Session per configuration, alternating timed graph replays and driver loops."""
import ast, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device('cuda:0')
sessions = []
for c in sys.argv[1:]:
    try:
        model, _ = bench.build_model(0, 1, dev)
        model.engine.tuning = dict(model.engine.tuning)
        model.engine.tuning.update(ast.literal_eval(c))
        sess, crops, p = bench.open_stream(model, dev, seed=0)
        conf = bench.Confidences()
        bench.run_frames(sess, crops, p, conf, 30)
        sessions.append((c, sess, crops, p, conf))
    except Exception as e:
        print('%-50s FAILED: %s' % (c, e), flush=True)
for rnd in range(3):
    for c, sess, crops, p, conf in sessions:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(800): sess.plan.run()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 800 * 1e6
        print('%-50s graph %.1f us' % (c[:50], graph), flush=True)
