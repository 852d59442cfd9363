"""Round-4 probe: does the batch-64 low-precision backbone gain from running as TWO half batches on two HIP streams, one
half in its MFMA-bound layers while the other streams its HBM-bound ones?  Aggregate crops/s of
  (a) one batch-64 graph,
  (b) one batch-32 graph alone,
  (c) two batch-32 graphs on two streams, launched together,
  (d) the same with stream 2 delayed by a fraction of a step (anti-phase)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import synth
from usot_amd.model import USOT
from usot_amd.engine import Engine

dev = torch.device('cuda:0')
m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to(dev)


def plan_for(batch, options=None):
    e = Engine(m, dev, options=options)
    x = torch.from_numpy(synth.crop(1, batch, 255)).to(dev)
    for _ in range(3):
        e.features_bf16(x)
    return e, next(v for k, v in e._feat.items() if k[:3] == ('bf16', batch, 255))['plan']


def timed(fn, n=200):
    torch.cuda.synchronize()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


e64, p64 = plan_for(64)
t64 = timed(p64.run)
print('batch 64, one graph           : %.3f ms/step  %8.0f crops/s' % (t64 * 1e3, 64 / t64))
for minp in (192, 100):
    opt = {'panel_min_panels': minp}
    ea, pa = plan_for(32, opt)
    eb, pb = plan_for(32, opt)
    t32 = timed(pa.run)
    print('panel_min_panels %d' % minp)
    print('  batch 32, one graph         : %.3f ms/step  %8.0f crops/s' % (t32 * 1e3, 32 / t32))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def both(delay_cycles=0):
        with torch.cuda.stream(sa):
            pa.run()
        with torch.cuda.stream(sb):
            if delay_cycles:
                torch.cuda._sleep(delay_cycles)
            pb.run()
    for delay_us in (0, 20, 60, 150, 400):
        cyc = int(delay_us * 100)            # torch.cuda._sleep counts ~100 MHz timer ticks? calibrated below
        t = timed(lambda: both(cyc))
        print('  2 x batch 32, 2 streams, stream-2 sleep %6d ticks: %.3f ms per pair  %8.0f crops/s' % (cyc, t * 1e3, 64 / t))
    # free-running: each stream replays its own graph back to back, no common cadence
    N = 200
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(sa):
            pa.run()
        with torch.cuda.stream(sb):
            pb.run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / N
    print('  2 x batch 32 free-running   : %.3f ms per pair  %8.0f crops/s' % (t * 1e3, 64 / t))
# calibrate the sleep
for cyc in (1000, 10000, 100000):
    t = timed(lambda: torch.cuda._sleep(cyc), 50)
    print('sleep(%d) = %.1f us' % (cyc, t * 1e6))
