#!/usr/bin/env python3
"""Per-frame period of the driver's tracking loop, frame by frame: the loop lives in two regimes on this pool (~862 us =
the graph's own replay time, and ~915-965 us), switching in stretches and differing from box to box.  Optional engine
switches as arguments (one session each, alternated): python scripts/loop_seq.py stream_1x1=0 stream_1x1=1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from usot_amd import engine
dev = torch.device('cuda:0')
if os.environ.get('USOT_PIN_CPUS'):
    lo, hi = (int(v) for v in os.environ['USOT_PIN_CPUS'].split('-'))
    os.sched_setaffinity(0, range(lo, hi + 1))
    print('pinned to CPUs', os.environ['USOT_PIN_CPUS'], flush=True)
cfgs = sys.argv[1:] or ['stream_1x1=1']
sessions = []
for c in cfgs:
    k, v = c.split('=')
    if not hasattr(engine, k):
        raise SystemExit('usot_amd.engine has no switch %s' % k)
    engine.OPTIONS[k.lower()] = bool(int(v))
    model, _ = bench.build_model(0, 1, dev)
    sess, crops, p = bench.open_stream(model, dev, seed=0)
    conf = bench.Confidences()
    bench.run_frames(sess, crops, p, conf, 50)
    sessions.append((c, sess, crops, p, conf))
def loop(sess, crops, p, conf, n):
    per = []
    t_prev = time.perf_counter()
    for i in range(n):
        picks = bench.select_memory(conf.view(), p.mem_queue_size)
        sess.submit(crops[i % crops.shape[0]], picks, (63.5, 63.5), inplace=True)
        out = sess.collect()
        conf.append(float(out[1]))
        now = time.perf_counter(); per.append((now - t_prev) * 1e6); t_prev = now
    return np.array(per)
for rnd in range(4):
    for c, sess, crops, p, conf in sessions:
        a = loop(sess, crops, p, conf, 800)[50:]
        t0 = time.perf_counter(); bench.run_frames(sess, crops, p, conf, 800); rf = (time.perf_counter() - t0) / 800 * 1e6
        print('%-22s mean %.1f median %.1f us, %4.1f %% of frames over 900 us | run_frames %.1f | frames 100-111: %s' % (
            c, a.mean(), np.median(a), 100.0 * (a > 900).mean(), rf, a[50:62].round(0).astype(int).tolist()), flush=True)
