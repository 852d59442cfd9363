#!/usr/bin/env python3
"""In-frame tile selection.  scripts/tune_conv.py times each convolution alone, back to back, i.e. with
its operands warm in L2; inside the tracked frame every layer meets cold weights and a different
neighbourhood, and the ranking of close candidates changes (a table that was 4 % faster per layer in
isolation made the frame 4 % slower).  This script starts from the isolated table and, layer shape by
layer shape (slowest first), tries the isolated top candidates INSIDE the batch-1 frame graph, keeping
the one with the shortest graph replay.  Greedy coordinate descent over ~25 shapes."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import synth
from usot_amd.model import USOT

ap = argparse.ArgumentParser()
ap.add_argument('--table', required=True, help='isolated table (tune_conv.py --out)')
ap.add_argument('--candidates', default='', help='tune_conv.py --candidates')
ap.add_argument('--extra-tiles', default='', help='comma list of tile ids tried on every shape with the incumbent split-K (and neighbours)')
ap.add_argument('--out', required=True)
ap.add_argument('--ks-neighbours', action='store_true')
ap.add_argument('--size', type=int, default=255)
ap.add_argument('--reps', type=int, default=40)
ap.add_argument("--passes", type=int, default=1)
ap.add_argument("--top", type=int, default=6, help="isolated candidates tried per shape")
ap.add_argument("--session", action="store_true", help="replay the device-resident Session frame (adds the per-frame encode of the new memory feature) instead of Engine.track")
a = ap.parse_args()

m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to('cuda:0')
m.pr_pool = False
t = lambda x: torch.from_numpy(x).cuda()
m.template(t(synth.crop(0, 1, 127)))
x, mem, sm = t(synth.crop(1, 1, a.size)), t(synth.memory_kernels(7, 7)), torch.ones(1, 7).cuda()
e = m.engine
key = lambda k: tuple(int(v) for v in k.split(','))
with open(a.table) as f:
    table = {key(k): tuple(v) for k, v in json.load(f).items()}
cands = {}
if a.candidates:
    with open(a.candidates) as f:
        cands = {key(k): [(c[0], c[1]) for c in v] for k, v in json.load(f).items()}
extra = [int(v) for v in a.extra_tiles.split(',') if v]
if extra:                                   # incumbent first, then the extra tiles at the incumbent's split-K and its neighbours
    for k, cur in table.items():
        ks = cur[1]
        splits = sorted({ks, max(1, ks - 1), ks + 1, max(1, ks // 2), ks * 2}, key=lambda s: abs(s - ks)) if a.ks_neighbours else [ks]
        alts = [(tl, s) for s in splits for tl in extra]
        cands[k] = [tuple(cur)] + alts + [c for c in cands.get(k, []) if tuple(c) != tuple(cur)]
e.tuning = dict(table)


def frame_us():
    if a.session:
        import bench
        sess, crops, pp = bench.open_stream(m, torch.device('cuda:0'), seed=0, size=a.size)
        bench.run_frames(sess, crops, pp, bench.Confidences(), 3)
        p = dict(plan=sess.plan, log=sess.log)
    else:
        e._track.clear()
        for _ in range(2):
            m.track(x, mem, sm)
        p = e._track[(1, a.size, 7)]
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.reps):
            p['plan'].run()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / a.reps * 1e6)
    return best, p


base, p = frame_us()
print('isolated table: %.1f us/frame' % base, flush=True)
prof = p['plan'].profile(10)
spans = {}
convs = iter(p['log'])
for kind, tile, ks, groups, ms in prof:
    if kind in (18, 19, 20, 21):                                          # fused pointwise pair: a log entry, no tile to choose
        next(convs)
    if kind == 0:
        nm, M, N, K, g, macs = next(convs)
        k4 = (M, N, K, 1) if '+' in nm else (M, N, K, g)      # a batched launch runs on its lead problem's tile
        spans[k4] = spans.get(k4, 0.0) + ms
order = [k for k, _ in sorted(spans.items(), key=lambda kv: -kv[1]) if k in cands]
print('%d conv shapes in the frame have candidates' % len(order), flush=True)
for ps in range(a.passes):
    for k in order:
        cur = e.tuning.get(k)
        results = []
        for c in cands[k][:a.top]:
            e.tuning[k] = c
            try:
                us, _ = frame_us()
            except Exception as ex:          # a candidate the batched launch cannot take
                continue
            results.append((us, c))
        us, c = min(results)
        if cur in [r[1] for r in results]:
            cur_us = [r[0] for r in results if r[1] == cur][0]
            if us > cur_us - 0.7:            # keep the incumbent unless the gain is above the noise
                us, c = cur_us, cur
        e.tuning[k] = c
        print('%-22s -> tile %2d ks %2d   %.1f us/frame   (%s)' % (k, c[0], c[1], us, ' '.join('%d/%d:%.0f' % (r[1][0], r[1][1], r[0]) for r in results)), flush=True)
final, _ = frame_us()
print('in-frame table: %.1f us/frame (isolated %.1f)' % (final, base))
with open(a.out, 'w') as f:
    json.dump({'%d,%d,%d,%d' % k: list(v) for k, v in e.tuning.items()}, f, indent=0, sort_keys=True)
print('wrote', a.out)
