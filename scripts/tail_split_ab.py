import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
from usot_amd import engine
dev = torch.device('cuda:0')
cfgs = [None, (1, 3), (1, 4), (1, 7), (2, 2), (2, 3), (1, 2), (3, 2)]
sessions = []
for c in cfgs:
    engine.OPTIONS['conf_tail_split'] = c
    model, _ = bench.build_model(0, 1, dev)
    sess, crops, p = bench.open_stream(model, dev, seed=0)
    conf = bench.Confidences()
    bench.run_frames(sess, crops, p, conf, 30)
    sessions.append((c, sess, crops, p, conf))
for rnd in range(3):
    for c, sess, crops, p, conf in sessions:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): sess.plan.run()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 500 * 1e6
        print('%-10s graph %.1f us' % (c, graph), flush=True)
