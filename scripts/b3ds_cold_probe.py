"""Where do b3.ds's 30 us go between the isolated run (119 us) and the graph (147-150 us)?    The 3x3 / stride-2
shortcut conv of layer2.0 at batch 64 (63 x 63 x 256 -> 31 x 31 x 512, tile 32) timed (a) on ONE input buffer, re-read from the 256 MB
Infinity Cache by every repetition, (b) on rotating sets of buffers so that every launch reads its 130 MB input from HBM as in the graph,
(c) as (b) with the output of a preceding HBM-heavy launch still draining (a 195 MB copy right before each launch)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import hip
DEV = 'cuda:0'
n, h, cin, cout = 64, 63, 256, 512
oh = (h - 3) // 2 + 1
dtype = torch.bfloat16
g = torch.Generator().manual_seed(3)
w = (torch.randn(cout, 9 * cin, generator=g) / (9 * cin) ** 0.5).to(dtype).to(DEV)
b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
L = hip.lib()


def timeit(nsets, iters=30, drain=False):
    xs = [torch.randn(n, h, h, cin, device=DEV).to(dtype) for _ in range(nsets)]
    ys = [torch.empty(n, oh, oh, cout, dtype=dtype, device=DEV) for _ in range(nsets)]
    big = torch.empty(195 * 1024 * 1024 // 2, dtype=dtype, device=DEV) if drain else None
    big2 = torch.empty_like(big) if drain else None
    ds = [hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N=n, H=h, W=h, Cin=cin, OH=oh, OW=oh, Cout=cout, KH=3, KW=3,
                        stride=2, tile=32) for x, y in zip(xs, ys)]
    for d in ds:
        hip.check(L.usot_conv2d_lp(hip.stream(), C.byref(d), 0, 0), 'conv')
    torch.cuda.synchronize()
    tot = 0.0
    for i in range(iters):
        if drain:
            big2.copy_(big)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(L.usot_conv2d_lp(hip.stream(), C.byref(ds[i % nsets]), 0, 0), 'conv')
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


for rep in range(2):
    print('b3.ds tile 32: one buffer set %.1f us | 4 rotating sets (cold input) %.1f us | cold input + a 195 MB copy draining %.1f us'
          % (timeit(1), timeit(4), timeit(4, drain=True)))
