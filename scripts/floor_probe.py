"""Graph-replay cost of a chain of N trivial dependent kernels (kernel floor on this GPU)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import hip
from usot_amd.engine import Plan
L = hip.lib()
dev = torch.device('cuda:0')
src = torch.randn(16, 1024, device=dev); dst = torch.empty(16, 1024, device=dev)
idx = torch.arange(4, device=dev, dtype=torch.int32)
for n in (1, 50, 200):
    for rows, row in ((4, 1024),):
        pl = Plan()
        for i in range(n):
            a, b = (src, dst) if i % 2 == 0 else (dst, src)
            hip.check(L.usot_plan_add_rows_copy(pl.h, hip.ptr(a), hip.ptr(idx), hip.ptr(b), rows, row, 0), 'rows')
        pl.capture()
        for _ in range(5): pl.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = 50
        for _ in range(R): pl.run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R * 1e6
        print('chain of %3d trivial kernels: %.1f us per replay, %.2f us per kernel' % (n, dt, dt / n))
