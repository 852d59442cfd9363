#!/usr/bin/env python3
"""Host cost of one Session.submit(): the hipGraphLaunch call (usot_plan_run) against the Python around it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
dev = torch.device('cuda:0')
model, _ = bench.build_model(0, 1, dev)
sess, crops, p = bench.open_stream(model, dev, seed=0)
conf = bench.Confidences()
bench.run_frames(sess, crops, p, conf, 50)
t_run, t_sub = [], []
for i in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sess.plan.run(); t1 = time.perf_counter()
    t_run.append(t1 - t0)
torch.cuda.synchronize()
for i in range(300):
    picks = bench.select_memory(conf.view(), p.mem_queue_size)
    t0 = time.perf_counter(); sess.submit(crops[i % crops.shape[0]], picks, (63.5, 63.5), inplace=True); t1 = time.perf_counter()
    out = sess.collect(); conf.append(float(out[1]))
    t_sub.append(t1 - t0)
med = lambda a: float(np.median(a[50:])) * 1e6
print('plan.run() on an idle stream: %.1f us | submit(): %.1f us | graph nodes %d' % (med(t_run), med(t_sub), len(sess.plan.profile(1))))
