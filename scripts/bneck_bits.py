#!/usr/bin/env python3
"""sha256 of the outputs of the fused layer1 bottleneck kernels (csrc/bneck_lp.hip) on seeded ragged images, for a library given
by USOT_HIP_LIB: the default build (counted `s_waitcnt vmcnt(n)` before a tile's halo is read) and the -DUSOT_BNECK_VMCNT0 build
(waits for everything) must print the same digests (tests/test_gpu_ops.py::test_bneck_counted_waits_equal_full_waits).
More tiles than resident workgroups and repeated launches, so a stale halo tile would show."""
import ctypes as C, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from usot_amd import hip
DEV = 'cuda:0'
h = hashlib.sha256()
for dtype, dt in ((torch.bfloat16, 0), (torch.float16, 1)):
    for N, H, W in ((40, 9, 17), (3, 63, 63), (64, 33, 31)):
        g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
        rnd = lambda *s: torch.randn(*s, generator=g)
        dev = lambda v: v.contiguous().to(DEV)
        x = dev(rnd(N, H, W, 64).relu().to(dtype))
        w1, b1 = dev((rnd(64, 64) / 8).to(dtype)), dev(rnd(64) * 0.1)
        w2, b2 = dev((rnd(64, 576) / 24).to(dtype)), dev(rnd(64) * 0.1)
        w3c, b3 = dev((rnd(256, 128) / 8).to(dtype)), dev(rnd(256) * 0.1)
        wn, bn = dev((rnd(64, 256) / 16).to(dtype)), dev(rnd(64) * 0.1)
        M = N * H * W
        y = torch.zeros(M, 256, dtype=dtype, device=DEV); t = torch.zeros(M, 64, dtype=dtype, device=DEV)
        d = hip.bneck_desc(*[hip.ptr(v) for v in (x, w1, b1, w2, b2, w3c, b3, wn, bn, y, t)], N, H, W)
        for _ in range(3):
            hip.check(hip.lib().usot_bneck_first_lp(hip.stream(), C.byref(d), dt), 'bneck_first')
        torch.cuda.synchronize()
        h.update(y.view(torch.int16).cpu().numpy().tobytes()); h.update(t.view(torch.int16).cpu().numpy().tobytes())
        for cn in (64, 128):
            res = dev(rnd(N, H, W, 256).relu().to(dtype))
            w3, wn2, bn2 = dev((rnd(256, 64) / 8).to(dtype)), dev((rnd(cn, 256) / 16).to(dtype)), dev(rnd(cn) * 0.1)
            y2 = torch.zeros(M, 256, dtype=dtype, device=DEV); t2 = torch.zeros(M, cn, dtype=dtype, device=DEV)
            d2 = hip.bneck_desc(hip.ptr(t), hip.ptr(res), None, hip.ptr(w2), hip.ptr(b2), hip.ptr(w3), hip.ptr(b3), hip.ptr(wn2), hip.ptr(bn2),
                                hip.ptr(y2), hip.ptr(t2), N, H, W)
            for _ in range(3):
                hip.check(hip.lib().usot_bneck_tail_lp(hip.stream(), C.byref(d2), cn, dt), 'bneck_tail')
            torch.cuda.synchronize()
            h.update(y2.view(torch.int16).cpu().numpy().tobytes()); h.update(t2.view(torch.int16).cpu().numpy().tobytes())
print('bneck_bits', h.hexdigest())
