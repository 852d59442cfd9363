"""GroupDW variant probe: algorithmic GB/s (3 161 088 B per sample) of a kernel variant
at several sample counts, plus a correctness check of that variant against the strips kernel."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from usot_amd import hip
dev = 'cuda:0'
B = 3161088
variants = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['4', '6'])]
geo = ((5, 5), (3, 5), (5, 3))
w = np.array([0.2, 0.3, 0.5], np.float32)
for OW in (25, 27):
    g = torch.Generator().manual_seed(1)
    S = 70
    xs = [torch.randn(S // 7, OW + hk - 1, OW + wk - 1, 256, generator=g).to(dev) for hk, wk in geo]
    zs = [torch.randn(S, hk, wk, 256, generator=g).to(dev) for hk, wk in geo]
    ref = hip.groupdw(xs, zs, w, x_rep=7, cols=1)
    for v in variants:
        out = hip.groupdw(xs, zs, w, x_rep=7, cols=v)
        torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
        print('OW=%d variant %d vs strips: max rel err %.2e %s' % (OW, v, err, 'OK' if err < 1e-5 else 'MISMATCH'))
for S in (128, 512, 2048):
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(S, 25 + hk - 1, 25 + wk - 1, 256, generator=g).to(dev) for hk, wk in geo]
    zs = [torch.randn(S, hk, wk, 256, generator=g).to(dev) for hk, wk in geo]
    for v in variants:
        for _ in range(3): hip.groupdw(xs, zs, w, cols=v)
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): hip.groupdw(xs, zs, w, cols=v)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print('groupdw S=%4d variant %d: %8.1f us  %7.1f GB/s algorithmic' % (S, v, best * 1e3, S * B / best / 1e6))
# device copy of the same byte count for scale
n = 128 * B // 8
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
for _ in range(3): b.copy_(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print('device copy %d MB: %.1f us, %.1f GB/s (read+write)' % (n * 4 // 1000000, ms * 1e3, 2 * n * 4 / ms / 1e6))
