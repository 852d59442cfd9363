import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch, torch.nn.functional as F
from usot_amd import hip
import test_gpu_ops as T
DEV='cuda:0'
for tile in [int(v) for v in sys.argv[1:]] or (91, 94, 97, 99):
    for ci, case in enumerate(T.CONV_CASES):
        N, Cin, H, W, Cout, k, stride, pad, dil = case
        g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
        b = torch.randn(Cout, generator=g)
        ref = F.conv2d(x.double(), w.double(), b.double(), stride, pad, dil).float()
        ref32 = F.conv2d(x, w, b, stride, pad, dil)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wd, bd = T.pack_w(w).to(DEV), b.to(DEV)
        try:
            y = hip.conv2d(xd, wd, bd, KH=k, KW=k, stride=stride, pad=pad, dil=dil, tile=tile)
            y0 = hip.conv2d(xd, wd, bd, KH=k, KW=k, stride=stride, pad=pad, dil=dil, tile=55)
            e = T.rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy())
            e0 = T.rel_err(y0.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy())
            e32 = T.rel_err(ref32.numpy(), ref.numpy())
            print('tile %d case %d %s: split16 err vs f64 %.2e | fp32 tile 55 %.2e | torch-cpu f32 %.2e' % (tile, ci, case[:6], e, e0, e32), flush=True)
        except Exception as ex:
            print('tile %d case %d: %s' % (tile, ci, ex), flush=True)
