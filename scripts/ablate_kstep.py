#!/usr/bin/env python3
"""What bounds a batch-1 k-step of conv_igemm_f32_v3: times layer3's 3x3 dilated conv (M = 961, N = 256, K = 2304)
alone, back to back, on libraries built with parts of the kernel removed (-DUSOT_ABL_NOMMA: no MFMAs,
-DUSOT_ABL_NOREAD: no fragment reads, -DUSOT_ABL_NOLOAD: producers fetch nothing, -DUSOT_ABL_NOSTORE: producers
write nothing to LDS, -DUSOT_ABL_NOBARRIER: no per-k-step barrier, both halves free-running).  Results are wrong by construction; only the durations matter.
    python scripts/ablate_kstep.py <lib.so> [tile ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['USOT_HIP_LIB'] = os.path.abspath(sys.argv[1])
import torch
from usot_amd import hip
L = hip.lib(); dev = 'cuda:0'
N, H, W, Cin, Cout, k, pad, dil = 1, 31, 31, 256, 256, 3, 2, 2
x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k * k * Cin, device=dev) * 0.02
b = torch.randn(Cout, device=dev); y = torch.empty(N, H, H, Cout, device=dev)
out = []
ws = torch.zeros(16 * 961 * 256 + 4096, device=dev)
for spec in sys.argv[2:] or ['54', '53', '55']:          # tile or tile:ksplit
    tile, ks = (int(v) for v in (spec.split(':') + ['1'])[:2])
    d = hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N=N, H=H, W=W, Cin=Cin, OH=H, OW=H, Cout=Cout,
                      KH=k, KW=k, pad=(pad, pad), dil=(dil, dil), act=1, tile=tile, ksplit=ks, ws=ws.data_ptr())
    for _ in range(5):
        hip.check(L.usot_conv2d_f32(hip.stream(), C.byref(d)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        hip.check(L.usot_conv2d_f32(hip.stream(), C.byref(d)))
    e1.record(); torch.cuda.synchronize()
    out.append('%s ks%d %.2f us' % (hip.tile_name(tile), ks, e0.elapsed_time(e1) / 200 * 1e3))
print('%-28s %s' % (os.path.basename(sys.argv[1]), ' | '.join(out)))
