#!/usr/bin/env python3
"""Small-M streaming 3x3 conv (usot_stream_conv3x3_f32) against the tuned tiled kernel on layer3 / layer2's conv2 at batch 1."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from usot_amd import hip, engine
dev = 'cuda:0'
L = hip.lib(); st = hip.stream()
tuning = engine.load_tuning()
for Cin, N, H, pad, dil in ((256, 256, 31, 2, 2), (256, 256, 31, 1, 1), (128, 128, 31, 1, 1), (256, 256, 33, 2, 2)):
    x = torch.randn(1, H, H, Cin, device=dev); w4 = torch.randn(N, Cin, 3, 3, device=dev) / (9 * Cin) ** 0.5; b = torch.randn(N, device=dev)
    w = w4.permute(0, 2, 3, 1).reshape(N, -1).contiguous()
    y = hip.stream_conv3x3_f32(x, w, b, (pad, pad), (dil, dil), None, hip.ACT_RELU)
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w4.double(), b.double(), 1, pad, dil)).permute(0, 2, 3, 1)
    err = float(((y.double() - ref).abs() / torch.maximum(ref.abs(), ref.abs().mean())).max())
    M, K = H * H, 9 * Cin
    tile, ks = tuning.get((M, N, K, 1), (0, 1))
    y2 = torch.empty_like(y); wp = hip.pw_pair_f32_pack(w)
    ws = torch.zeros(ks * M * N + 4096, device=dev) if ks > 1 else None
    d = hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y2.data_ptr(), N=1, H=H, W=H, Cin=Cin, OH=H, OW=H, Cout=N, KH=3, KW=3,
                      pad=(pad, pad), dil=(dil, dil), act=hip.ACT_RELU, tile=tile, ksplit=ks, ws=ws.data_ptr() if ws is not None else None)
    fa = lambda: L.usot_conv2d_f32(st, C.byref(d))
    fb = lambda: L.usot_stream_conv3x3_f32(st, hip.ptr(x), hip.ptr(wp), hip.ptr(b), None, hip.ptr(y), 1, H, H, Cin, H, H, N, pad, pad, dil, dil, hip.ACT_RELU)
    out = []
    for name, fn in (('tiled (tile %d ks %d)' % (tile, ks), fa), ('streaming', fb)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): fn()
        e1.record(); torch.cuda.synchronize()
        out.append('%s %.1f us' % (name, e0.elapsed_time(e1) / 300 * 1e3))
    print('Cin %d N %d %dx%d pad %d dil %d: err %.1e | %s' % (Cin, N, H, H, pad, dil, err, ' | '.join(out)))
