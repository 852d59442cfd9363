import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import prroi_exact as ex, usot_oracle as orc
from prroi_cases import random_rois
from usot_amd import hip
DEV='cuda:0'
for seed, shape, n in [(11, (2, 32, 15, 17), 96), (12, (1, 16, 31, 31), 60)]:
    B, C, H, W = shape
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(B, C, H, W, generator=g); rois = random_rois(seed, n, B, H, W); td = torch.randn(n, C, 7, 7, generator=g)
    bins = np.minimum(rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]) / 7
    fd, rd = f.to(DEV), torch.from_numpy(rois).to(DEV)
    top = hip.prroi_pool(fd, rd, 7, 7, 1.0)
    for lo in (0.0, 0.05, 0.2, 1.0):
        m = bins >= lo if lo > 0 else bins > -1
        mt = torch.from_numpy(m)
        gf = hip.prroi_pool_backward(f.shape, rd[mt.to(DEV)], td[mt].to(DEV), 7, 7, 1.0).cpu().numpy()
        wf = ex.prroi_pool_exact_backward(f.shape, rois[m], td.numpy()[m], 7, 7, 1.0)
        rf = orc.prroi_pool_backward(f.shape, rois[m], td[mt], 7, 7, 1.0).numpy()
        gr = hip.prroi_pool_coor_backward(fd, rd[mt.to(DEV)], top[mt.to(DEV)], td[mt].to(DEV), 7, 7, 1.0).cpu().numpy()
        wr = ex.prroi_pool_exact_coor_backward(f.numpy(), rois[m], td.numpy()[m], 7, 7, 1.0)
        rr = orc.prroi_pool_coor_backward(f, rois[m], top[mt.to(DEV)].cpu(), td[mt], 7, 7, 1.0).numpy()
        sf, sr = max(1, np.abs(wf).max()), np.maximum(1, np.abs(wr).max(1))
        print('seed %d bins>=%.2f (%d rois): feat hip-exact %.2e hip-cref %.2e cref-exact %.2e | roi hip-exact %.2e hip-cref %.2e cref-exact %.2e' % (
            seed, lo, m.sum(), np.abs(gf - wf).max() / sf, np.abs(gf - rf).max() / sf, np.abs(rf - wf).max() / sf,
            (np.abs(gr - wr).max(1) / sr).max(), (np.abs(gr - rr).max(1) / sr).max(), (np.abs(rr - wr).max(1) / sr).max()))
import time
f = torch.randn(8, 256, 31, 31, device=DEV); rois = torch.tensor([[i, 5.3, 6.1, 20.7, 24.2] for i in range(8)], device=DEV); td = torch.randn(8, 256, 7, 7, device=DEV)
top = hip.prroi_pool(f, rois, 7, 7, 1.0)
for name, fn in (('backward', lambda: hip.prroi_pool_backward(f.shape, rois, td, 7, 7, 1.0)), ('coor_backward', lambda: hip.prroi_pool_coor_backward(f, rois, top, td, 7, 7, 1.0))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): fn()
    torch.cuda.synchronize(); print(name, '%.1f us' % ((time.perf_counter() - t0) / 100 * 1e6))
