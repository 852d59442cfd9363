"""Per-op timing of one tracked frame (HIP events via usot_plan_profile) + graph replay time."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import synth, hip
from usot_amd.model import USOT

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--size', type=int, default=255)
ap.add_argument('--mem', type=int, default=7)
ap.add_argument('--frames', type=int, default=20)
ap.add_argument('--lanes', type=int, default=0)
ap.add_argument('--quiet', action='store_true')
a = ap.parse_args()
KINDS = ['conv', 'stem', 'maxpool', 'groupdw', 'conf_reduce', 'prroi', 'permute', 'decode', 'fork', 'join']
m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to('cuda:0')
m.engine_options['lanes'] = a.lanes
B = a.batch
t = lambda x: torch.from_numpy(x).cuda()
m.pr_pool = False
m.template(t(synth.crop(0, B, 127)))
x = t(synth.crop(1, B, a.size)); mem = t(synth.memory_kernels(7, B * a.mem)) if a.mem else None
sm = torch.ones(B, a.mem).cuda() if a.mem else None
for _ in range(3): m.track(x, mem, sm)
e = m.engine
p = e._track[(B, a.size, a.mem)]
prof = p['plan'].profile(a.frames)
convs = iter(p['log'])
tiles = hip.tile_table()
tot = 0.0; tot_conv = 0.0; flops = 0.0
by_tile = {}
print('%-16s %7s %5s %6s %3s %9s %9s %8s' % ('op', 'M', 'N', 'K', 'g', 'tile', 'us', 'TFLOP/s'))
for kind, tile, ks, groups, ms in prof:
    tot += ms
    if kind in (18, 19, 20, 21):                                          # fused pointwise pair (csrc/smallm_f32.hip)
        name, M, N, K, g, macs = next(convs)
        tot_conv += ms; flops += 2 * macs
        print('%-16s %7d %5s %6s %3s %9s %9.1f %8.1f' % (name[:16], M, '', '', '', 'fused', ms * 1e3, 2 * macs / (ms * 1e-3) / 1e12))
        continue
    if kind == 0:
        name, M, N, K, g, macs = next(convs)
        tf = 2 * macs / (ms * 1e-3) / 1e12
        tot_conv += ms; flops += 2 * macs
        bt = by_tile.setdefault(tile, [0, 0.0, 0.0]); bt[0] += 1; bt[1] += ms; bt[2] += 2 * macs
        print('%-16s %7d %5d %6d %3d %4dx%-4d %9.1f %8.1f' % (name, M, N, K, g, tiles[tile][0], tiles[tile][1], ms * 1e3, tf))
    elif kind < 8:
        print('%-16s %46s %9.1f' % (KINDS[kind], '', ms * 1e3))
print('sum of op spans %.1f us (conv %.1f us, %.2f GFLOP -> %.1f TFLOP/s on conv time)' % (tot * 1e3, tot_conv * 1e3, flops / 1e9, flops / tot_conv / 1e9))
for tile, (n, ms, fl) in sorted(by_tile.items()):
    print('tile %dx%d: %d launches, %.1f us, %.1f TFLOP/s' % (tiles[tile][0], tiles[tile][1], n, ms * 1e3, fl / ms / 1e9))
torch.cuda.synchronize()
for _ in range(10): p['plan'].run()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 200
for _ in range(N): p['plan'].run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
print('graph replay: %.1f us/frame -> %.0f frames/s (batch %d: %.0f crops/s), %.1f TFLOP/s algorithmic' % (dt * 1e6, 1 / dt, B, B / dt, flops / dt / 1e12))
