#!/usr/bin/env python3
"""Layer1's bottleneck tail in one launch (usot_pw_triple_f32) against tuned tiled conv2 + fused pair, timed back to back."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from usot_amd import hip, engine
dev = 'cuda:0'
L = hip.lib(); st = hip.stream(); tuning = engine.load_tuning()
for cn in (64, 128):
    H, cin, cm, co = 63, 64, 64, 256
    M = H * H
    x = torch.randn(1, H, H, cin, device=dev); w2 = torch.randn(cm, 9 * cin, device=dev) * 0.04; b2 = torch.randn(cm, device=dev)
    w3 = torch.randn(co, cm, device=dev) * 0.1; b3 = torch.randn(co, device=dev); res = torch.randn(1, H, H, co, device=dev)
    w1 = torch.randn(cn, co, device=dev) * 0.05; b1 = torch.randn(cn, device=dev)
    t2 = torch.empty(1, H, H, cm, device=dev); y = torch.empty(1, H, H, co, device=dev); t = torch.empty(1, H, H, cn, device=dev)
    w2p, w3p, w1p = hip.pw_pair_f32_pack(w2), hip.pw_pair_f32_pack(w3), hip.pw_pair_f32_pack(w1)
    tile, ks = tuning.get((M, cm, 9 * cin, 1), (0, 1))
    d2 = hip.conv_desc(x.data_ptr(), w2.data_ptr(), b2.data_ptr(), t2.data_ptr(), N=1, H=H, W=H, Cin=cin, OH=H, OW=H, Cout=cm, KH=3, KW=3,
                       pad=(1, 1), act=hip.ACT_RELU, tile=tile, ksplit=ks)
    dp = hip.pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1.data_ptr(),
                          t.data_ptr(), M, cm, co, cn, hip.ACT_RELU)
    def two():
        L.usot_conv2d_f32(st, C.byref(d2)); L.usot_pw_pair_f32(st, C.byref(dp))
    one = lambda: L.usot_pw_triple_f32(st, hip.ptr(x), hip.ptr(w2p), hip.ptr(b2), C.byref(dp), 1, H, H, cin, H, H, 1, 1, 1, 1)
    out = []
    for name, fn in (('tiled conv2 (tile %d) + pair' % tile, two), ('one launch', one)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): fn()
        e1.record(); torch.cuda.synchronize()
        out.append('%s %.1f us' % (name, e0.elapsed_time(e1) / 300 * 1e3))
    print('CN %d: %s' % (cn, ' | '.join(out)))
