"""Workload for the GroupDW PMC passes: 2048-sample (6.5 GB) and 128-sample launches of the auto variant,
nothing else of note on the device.  Run under
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d <dir> -- python scripts/xcorr_pmc_run.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d <dir> -- python scripts/xcorr_pmc_run.py
(separate passes: the TCC block has 4 counter slots, MI355X_MICROARCH.md)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from usot_amd import hip
dev = 'cuda:0'
geo = ((5, 5), (3, 5), (5, 3))
w = np.array([0.3, 0.3, 0.4], np.float32)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator(device=dev).manual_seed(7)
xs = [torch.randn(S, 25 + hk - 1, 25 + wk - 1, 256, generator=g, device=dev) for hk, wk in geo]
zs = [torch.randn(S, hk, wk, 256, generator=g, device=dev) for hk, wk in geo]
for _ in range(6):
    hip.groupdw(xs, zs, w)
torch.cuda.synchronize()
