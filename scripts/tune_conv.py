#!/usr/bin/env python3
"""Autotune the implicit-GEMM tile and split-K factor for every distinct convolution shape of
the tracked frame (batch B) on the GPU at hand; writes usot_amd/data/tuning_gfx950.json,
which the engine loads by default.  Timing: HIP events around R back-to-back launches."""
import argparse, ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import hip, synth
from usot_amd.engine import Builder, Weights
from usot_amd.model import USOT

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, nargs='+', default=[1])
ap.add_argument('--sizes', type=int, nargs='+', default=[255, 127, 271])
ap.add_argument('--reps', type=int, default=12)
ap.add_argument('--out', default=os.path.join(ROOT, 'usot_amd', 'data', 'tuning_gfx950.json'))
ap.add_argument('--verbose', action='store_true')
ap.add_argument('--split-margin', type=float, default=0.03)
ap.add_argument('--candidates', default=None, help='also write the 6 fastest (tile, ksplit) per shape here (for scripts/tune_frame.py)')
a = ap.parse_args()

dev = torch.device('cuda:0')
m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to(dev)
W = Weights(m, dev)
shapes = {}


def collect(bld, B, size):
    x = bld.buf(B, 3, size, size)
    xf, hf = bld.backbone(x, B, size)
    if size == 127:
        zf = bld.buf(B, 7, 7, 256)
        bld.encode_kernel(zf, B, 512, 'z')
    else:
        zk = [bld.buf(B, hk, wk, 512) for hk, wk in ((5, 5), (3, 5), (5, 3))]
        mem = bld.buf(B * 7, 7, 7, 256)
        bld.heads(xf, B, hf, zk, mem, 7)
        bld.encode_kernel(bld.buf(B, 7, 7, 256), B, 256, 'mem')      # a session encodes ONE new memory feature per frame
        bld.heads(xf, B, hf, zk, None, 0)
    for g in bld.geoms:
        oh = (g['H'] + 2 * g['pad'][0] - g['dil'][0] * (g['KH'] - 1) - 1) // g['stride'] + 1
        ow = (g['W'] + 2 * g['pad'][1] - g['dil'][1] * (g['KW'] - 1) - 1) // g['stride'] + 1
        key = (g['N'] * oh * ow, g['Cout'], g['KH'] * g['KW'] * g['Cin'], g['groups'])
        shapes.setdefault(key, g)


for B in a.batch:
    for size in a.sizes:
        for lanes in (0, 2):
            collect(Builder(W, {}, lanes), B, size)
tiles = hip.tile_table()
L = hip.lib()
cands = {}
table = {}
if os.path.exists(a.out):
    with open(a.out) as f:
        table = json.load(f)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for key, g in sorted(shapes.items()):
    M, Cout, K, groups = key
    N, H, Wd, Cin = g['N'], g['H'], g['W'], g['Cin']
    OH = (H + 2 * g['pad'][0] - g['dil'][0] * (g['KH'] - 1) - 1) // g['stride'] + 1
    OW = (Wd + 2 * g['pad'][1] - g['dil'][1] * (g['KW'] - 1) - 1) // g['stride'] + 1
    x = torch.randn(groups, N, H, Wd, Cin, device=dev)
    w = torch.randn(groups, Cout, K, device=dev) * 0.02
    b = torch.randn(groups, Cout, device=dev)
    y = torch.empty(groups, N, OH, OW, Cout, device=dev)
    res = torch.randn_like(y) if g['has_res'] else None
    KT = g['KH'] * g['KW'] * (Cin // 32)
    best = None
    rows = []
    yref = None
    for tile, (bm, bn) in tiles.items():
        if bn >= 2 * max(16, Cout) and bn > 32: continue
        if bm >= 4 * M and bm > 16: continue
        blocks = -(-M // bm) * -(-Cout // bn) * groups
        for ks in (1, 2, 3, 4, 6, 8, 12, 16):
            if ks > 1 and (ks > KT // 2 or blocks * ks > 2048 or blocks >= 512): continue
            ws = torch.zeros(ks * groups * M * Cout + groups * ((M + 15) // 16) * ((Cout + 31) // 32), device=dev) if ks > 1 else None
            d = hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N=N, H=H, W=Wd, Cin=Cin, OH=OH, OW=OW,
                              Cout=Cout, KH=g['KH'], KW=g['KW'], stride=g['stride'], pad=g['pad'], dil=g['dil'],
                              res=res.data_ptr() if res is not None else None, act=1, groups=groups,
                              x_gs=N * H * Wd * Cin, w_gs=Cout * K, b_gs=Cout, y_gs=M * Cout, r_gs=M * Cout,
                              ksplit=ks, tile=tile, ws=ws.data_ptr() if ws is not None else None)
            hip.check(L.usot_conv2d_f32(hip.stream(), C.byref(d)))
            # never let a wrong kernel win on speed: every candidate must reproduce the first one
            if yref is None:
                yref = y.clone()
            elif not torch.allclose(y, yref, rtol=1e-3, atol=1e-3 * float(yref.abs().max())):
                print('   !! tile %d ks %d disagrees on %s (max diff %.3g): skipped' % (
                    tile, ks, g['name'], float((y - yref).abs().max())), flush=True)
                continue
            # time through a native plan (C++ launch loop): Python's ~6 us per ctypes call would
            # otherwise floor every short kernel to the same number
            plan = C.c_void_p(L.usot_plan_create())
            for _ in range(a.reps):
                hip.check(L.usot_plan_add_conv(plan, C.byref(d)))
            best_us = 1e30
            for _ in range(3):
                e0.record()
                hip.check(L.usot_plan_run(plan, hip.stream()))
                e1.record(); torch.cuda.synchronize()
                best_us = min(best_us, e0.elapsed_time(e1) / a.reps * 1e3)
            L.usot_plan_destroy(plan)
            us = best_us
            rows.append((us, tile, ks))
            if best is None or us < best[0]:
                best = (us, tile, ks)
    # a split-K candidate pays a workspace round trip and an in-launch combine (last-arriver reduction) that
    # this warm, back-to-back timing under-prices a little: prefer the best unsplit candidate unless splitting
    # wins by more than --split-margin (scripts/tune_frame.py then ranks the close ones inside the frame)
    unsplit = min((r for r in rows if r[2] == 1), default=None)
    if unsplit is not None and best[2] > 1 and unsplit[0] <= best[0] * (1.0 + a.split_margin):
        best = unsplit
    us, tile, ks = best
    tf = 2.0 * M * Cout * K * groups / us / 1e6
    table['%d,%d,%d,%d' % key] = [tile, ks]
    cands['%d,%d,%d,%d' % key] = [[t2, k2, round(us2, 2)] for us2, t2, k2 in sorted(rows)[:10]]
    print('%-14s M=%6d N=%5d K=%5d g=%d -> tile %2d (%dx%d) ksplit %d: %7.1f us %6.1f TFLOP/s' % (
        g['name'], M, Cout, K, groups, tile, tiles[tile][0], tiles[tile][1], ks, us, tf), flush=True)
    if a.verbose:
        for us2, t2, k2 in sorted(rows)[:6]:
            print('      %3dx%-3d ks%-2d %7.1f us' % (tiles[t2][0], tiles[t2][1], k2, us2))
with open(a.out, 'w') as f:
    json.dump(table, f, indent=0, sort_keys=True)
print('wrote', a.out, len(table), 'shapes')
if a.candidates:
    with open(a.candidates, 'w') as f:
        json.dump(cands, f, indent=0, sort_keys=True)
