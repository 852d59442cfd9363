#!/usr/bin/env python3
"""Per-frame period of the driver's tracking loop, frame by frame: shows the two regimes the loop lives in on this pool
(~862 us = the graph's own replay time, and ~915-945 us), switching in long stretches and differing from box to box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device('cuda:0')
model, _ = bench.build_model(0, 1, dev)
sess, crops, p = bench.open_stream(model, dev, seed=0)
conf = bench.Confidences()
bench.run_frames(sess, crops, p, conf, 50)
def loop(n, pre, spin_us=0):
    per = []
    t_prev = time.perf_counter()
    for i in range(n):
        picks = bench.select_memory(conf.view(), p.mem_queue_size)
        sess.submit(crops[i % crops.shape[0]], picks, (63.5, 63.5))
        out = sess.collect()
        conf.append(float(out[1]))
        if spin_us:
            t = time.perf_counter()
            while time.perf_counter() - t < spin_us * 1e-6: pass
        now = time.perf_counter(); per.append((now - t_prev) * 1e6); t_prev = now
    return np.array(per)
for rnd in range(4):
    a = loop(400, False)
    print('median %.1f us' % np.median(a[50:]), 'frames 100-123:', a[100:124].round(0).astype(int).tolist(), flush=True)
