#!/usr/bin/env python3
"""Pick the bf16/fp16 implicit-GEMM tile per convolution shape of the batched backbone (batch B);
writes usot_amd/data/tuning_lp_gfx950.json {"M,Cout,K": tile}.  Candidates must agree with tile 5."""
import argparse, ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import hip, synth
from usot_amd.engine import Builder, Weights
from usot_amd.model import USOT

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, nargs='+', default=[64, 32])
ap.add_argument('--reps', type=int, default=8)
ap.add_argument('--out', default=os.path.join(ROOT, 'usot_amd', 'data', 'tuning_lp_gfx950.json'))
a = ap.parse_args()
dev = torch.device('cuda:0')
m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to(dev)
W = Weights(m, dev)
L = hip.lib()
ntiles = L.usot_conv_bf16_tile_count()
table = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in a.batch:
    bld = Builder(W, {}, 0)
    bld.lp_geoms = []
    x = bld.buf(B, 3, 255, 255)
    xl, hf = bld.backbone_bf16(x, B, 255)
    zk = [bld.buf(B, hk, wk, 512) for hk, wk in ((5, 5), (3, 5), (5, 3))]
    bld.heads_lp(xl, B, hf, zk, bld.buf(B * 7, 7, 7, 256), 7, torch.bfloat16)      # config 5 head shapes
    for g in bld.lp_geoms:
        key = '%d,%d,%d' % (g['M'], g['Cout'], g['K'])
        if key in table:
            continue
        xin = (torch.randn(g['N'], g['H'], g['W'], g['Cin'], device=dev)).to(torch.bfloat16)
        w = (torch.randn(g['Cout'], g['K'], device=dev) * 0.02).to(torch.bfloat16)
        b = torch.randn(g['Cout'], device=dev)
        y = torch.empty(g['N'], g['OH'], g['OW'], g['Cout'], device=dev, dtype=torch.bfloat16)
        res = torch.randn(g['N'], g['OH'], g['OW'], g['Cout'], device=dev).to(torch.bfloat16) if g['has_res'] else None
        best, yref, rows = None, None, []
        for tile in [5] + [t for t in range(1, ntiles + 1) if t != 5]:
            d = hip.conv_desc(xin.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N=g['N'], H=g['H'], W=g['W'], Cin=g['Cin'],
                              OH=g['OH'], OW=g['OW'], Cout=g['Cout'], KH=g['KH'], KW=g['KW'], stride=g['stride'], pad=g['pad'],
                              dil=g['dil'], res=res.data_ptr() if res is not None else None, act=1, tile=tile)
            if L.usot_conv2d_lp(hip.stream(), C.byref(d), 0, 0) != 0:
                continue
            torch.cuda.synchronize()
            if yref is None:
                yref = y.float().clone()
            elif not torch.allclose(y.float(), yref, rtol=2e-2, atol=2e-2 * float(yref.abs().max())):
                print('   !! tile %d disagrees on %s' % (tile, g['name']), flush=True)
                continue
            plan = C.c_void_p(L.usot_plan_create())
            for _ in range(a.reps):
                hip.check(L.usot_plan_add_conv_lp(plan, C.byref(d), 0, 0))
            us = 1e30
            for _ in range(3):
                e0.record(); hip.check(L.usot_plan_run(plan, hip.stream())); e1.record(); torch.cuda.synchronize()
                us = min(us, e0.elapsed_time(e1) / a.reps * 1e3)
            L.usot_plan_destroy(plan)
            rows.append((us, tile))
        us, tile = min(rows)
        table[key] = tile
        print('%-10s M=%7d N=%5d K=%5d -> tile %d %8.1f us %7.1f TF/s   (%s)' % (
            g['name'], g['M'], g['Cout'], g['K'], tile, us, 2.0 * g['M'] * g['Cout'] * g['K'] / us / 1e6,
            ' '.join('%d:%.0f' % (t, u) for u, t in sorted(rows, key=lambda r: r[1]))), flush=True)
with open(a.out, 'w') as f:
    json.dump(table, f, indent=0, sort_keys=True)
print('wrote', a.out, len(table))
