import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from usot_amd import hip
dev = 'cuda:0'
B = 3161088
for S, cols in ((9, 1), (9, 4), (128, 1), (128, 4), (512, 1), (512, 4), (2048, 4)):
    g = torch.Generator().manual_seed(1)
    geo = ((5, 5), (3, 5), (5, 3))
    xs = [torch.randn(S, 25 + hk - 1, 25 + wk - 1, 256, generator=g).to(dev) for hk, wk in geo]
    zs = [torch.randn(S, hk, wk, 256, generator=g).to(dev) for hk, wk in geo]
    w = np.array([0.3, 0.3, 0.4], np.float32)
    for _ in range(3): hip.groupdw(xs, zs, w, cols=cols)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): hip.groupdw(xs, zs, w, cols=cols)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('groupdw S=%4d cols=%d: %8.1f us  %7.1f GB/s algorithmic' % (S, cols, ms * 1e3, S * B / ms / 1e6))
# plane kernel (NCHW API op)
for P in (256, 256 * 64):
    x = torch.randn(P, 29, 29, device=dev); k = torch.randn(P, 5, 5, device=dev)
    for _ in range(3): hip.xcorr_depthwise(x.view(1, P, 29, 29), k.view(1, P, 5, 5))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): hip.xcorr_depthwise(x.view(1, P, 29, 29), k.view(1, P, 5, 5))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = P * (29 * 29 + 25 + 25 * 25) * 4
    print('xcorr_planes P=%6d: %8.1f us  %7.1f GB/s' % (P, ms * 1e3, by / ms / 1e6))
