#!/usr/bin/env python3
"""Same-process A/B of the tracked frame with engine switches: builds one Session per configuration and alternates timed
runs of bench.run_frames (the driver's loop) and of bare graph replays.  python scripts/ab_frame.py stream_1x1=0 stream_1x1=1"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from usot_amd import engine
dev = torch.device('cuda:0')
cfgs = sys.argv[1:] or ['stream_1x1=0', 'stream_1x1=1']
sessions = []
for c in cfgs:
    for kv in c.split(','):
        k, v = kv.split('=')
        k = k.lower()                         # engine.OPTIONS keys: stream_1x1=0, fused_f32_sliced=1, ...
        cur = engine.OPTIONS[k]
        engine.OPTIONS[k] = set() if isinstance(cur, set) else (int(v) if isinstance(cur, bool) and int(v) > 1 else type(cur)(int(v)))
    model, _ = bench.build_model(0, 1, dev)
    sess, crops, p = bench.open_stream(model, dev, seed=0)
    conf = bench.Confidences()
    bench.run_frames(sess, crops, p, conf, 30)
    sessions.append((c, sess, crops, p, conf))
for rnd in range(3):
    for c, sess, crops, p, conf in sessions:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.run_frames(sess, crops, p, conf, 1000)
        torch.cuda.synchronize(); loop = (time.perf_counter() - t0) / 1000 * 1e6
        t0 = time.perf_counter()
        for _ in range(500): sess.plan.run()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 500 * 1e6
        print('%-28s loop %.1f us/frame   graph %.1f us' % (c, loop, graph), flush=True)
# where a loop period goes: host time in submit, host time waiting in collect, rest of the Python loop
import numpy as np
for c, sess, crops, p, conf in sessions:
    ts, tc, tr, tg = [], [], [], []
    st = torch.cuda.current_stream()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(600)]
    for i in range(600):
        t0 = time.perf_counter()
        picks = bench.select_memory(conf.view(), p.mem_queue_size)
        t1 = time.perf_counter()
        evs[i][0].record(st)
        sess.submit(crops[i % crops.shape[0]], picks, (63.5, 63.5), inplace=True)
        evs[i][1].record(st)
        t2 = time.perf_counter()
        out = sess.collect()
        t3 = time.perf_counter()
        conf.append(float(out[1]))
        ts.append(t2 - t1); tc.append(t3 - t2); tr.append(t1 - t0)
    torch.cuda.synchronize()
    tg = [a.elapsed_time(b) * 1e-3 for a, b in evs]
    med = lambda a: float(np.median(a[100:])) * 1e6
    print('%-28s picks %.1f us | submit %.1f us | collect (wait) %.1f us | GPU span of the frame (events) %.1f us' % (c, med(tr), med(ts), med(tc), med(tg)))
