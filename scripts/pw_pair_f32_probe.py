#!/usr/bin/env python3
"""Fused fp32 pointwise pair (csrc/smallm_f32.hip) against the two conv launches it replaces, timed back to back and
as graph-captured chains of the pair (the frame's regime: every launch depends on the previous one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from usot_amd import hip
dev = 'cuda:0'
for cm, co, cn, M in ((64, 256, 64, 3969), (64, 256, 128, 3969), (128, 512, 128, 961), (128, 512, 256, 961), (256, 1024, 256, 961)):
    t2 = torch.randn(1, 1, M, cm, device=dev); res = torch.randn(1, 1, M, co, device=dev)
    w3 = torch.randn(co, cm, device=dev) * 0.1; b3 = torch.randn(co, device=dev)
    w1 = torch.randn(cn, co, device=dev) * 0.05; b1 = torch.randn(cn, device=dev)
    import ctypes as C
    L = hip.lib()
    y = torch.empty(1, 1, M, co, device=dev); t = torch.empty(1, 1, M, cn, device=dev)
    w3p, w1p = hip.pw_pair_f32_pack(w3), hip.pw_pair_f32_pack(w1)
    dp = hip.pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1.data_ptr(),
                          t.data_ptr(), M, cm, co, cn, hip.ACT_RELU)
    ws = hip.pw_pair_f32_ws(M, cm, co, cn, dev)
    ds = hip.pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1.data_ptr(),
                          t.data_ptr(), M, cm, co, cn, hip.ACT_RELU, ws.data_ptr() if ws is not None else None)
    d1 = hip.conv_desc(t2.data_ptr(), w3.data_ptr(), b3.data_ptr(), y.data_ptr(), N=1, H=1, W=M, Cin=cm, OH=1, OW=M, Cout=co,
                       KH=1, KW=1, res=res.data_ptr(), act=hip.ACT_RELU)
    d2 = hip.conv_desc(y.data_ptr(), w1.data_ptr(), b1.data_ptr(), t.data_ptr(), N=1, H=1, W=M, Cin=co, OH=1, OW=M, Cout=cn,
                       KH=1, KW=1, act=hip.ACT_RELU)
    st = hip.stream()
    def two():
        L.usot_conv2d_f32(st, C.byref(d1)); L.usot_conv2d_f32(st, C.byref(d2))
    one = lambda: L.usot_pw_pair_f32(st, C.byref(dp))
    out = []
    sliced = lambda: L.usot_pw_pair_f32(st, C.byref(ds))
    for name, fn in (('two launches', two),) + ((('fused', one),) if cm < 256 else ()) + ((('fused, 4 channel slices', sliced),) if ws is not None else ()):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): fn()
        e1.record(); torch.cuda.synchronize()
        out.append('%s %.1f us' % (name, e0.elapsed_time(e1) / 300 * 1e3))
    print('CM %d CO %d CN %d M %d: %s' % (cm, co, cn, M, ' | '.join(out)))
