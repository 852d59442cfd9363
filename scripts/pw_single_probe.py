#!/usr/bin/env python3
"""Small-M streaming 1x1 conv (usot_pw_single_f32) against the tiled conv kernel on layer3 / layer2's pointwise layers at
batch 1: correctness vs float64 and back-to-back timing (tuned tile of the shipped table for the tiled one)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from usot_amd import hip, engine
dev = 'cuda:0'
L = hip.lib(); st = hip.stream()
tuning = engine.load_tuning()
for K, N, M, res in ((1024, 256, 961, False), (256, 1024, 961, True), (512, 128, 961, False), (128, 512, 961, True)):
    x = torch.randn(1, 1, M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    r = torch.randn(1, 1, M, N, device=dev) if res else None
    y = hip.pw_single_f32(x, w, b, r, hip.ACT_RELU)
    ref = torch.relu(x.double().reshape(M, K) @ w.double().t() + b.double() + (r.double().reshape(M, N) if res else 0))
    err = float(((y.reshape(M, N).double() - ref).abs() / torch.maximum(ref.abs(), ref.abs().mean())).max())
    tile, ks = tuning.get((M, N, K, 1), (0, 1))
    y2 = torch.empty_like(y); wp = hip.pw_pair_f32_pack(w)
    ws = torch.zeros(ks * M * N + 4096, device=dev) if ks > 1 else None
    d = hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y2.data_ptr(), N=1, H=1, W=M, Cin=K, OH=1, OW=M, Cout=N, KH=1, KW=1,
                      res=r.data_ptr() if res else None, act=hip.ACT_RELU, tile=tile, ksplit=ks, ws=ws.data_ptr() if ws is not None else None)
    fa = lambda: L.usot_conv2d_f32(st, C.byref(d))
    fb = lambda: L.usot_pw_single_f32(st, hip.ptr(x), hip.ptr(wp), hip.ptr(b), hip.ptr(r) if res else None, hip.ptr(y), M, K, N, hip.ACT_RELU)
    out = []
    for name, fn in (('tiled (tile %d ks %d)' % (tile, ks), fa), ('streaming', fb)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): fn()
        e1.record(); torch.cuda.synchronize()
        out.append('%s %.1f us' % (name, e0.elapsed_time(e1) / 300 * 1e3))
    print('K %4d N %4d M %d: err %.1e | %s' % (K, N, M, err, ' | '.join(out)))
