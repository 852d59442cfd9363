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
# SHAPE=conv2 (default) | tower (3 groups of 625 x 256 x 2304) | conf (7 x 625 pixels, 512 channels)
N, H, W, Cin, Cout, k, pad, dil, G = {'conv2': (1, 31, 31, 256, 256, 3, 2, 2, 1), 'tower': (1, 25, 25, 256, 256, 3, 1, 1, 3),
                                      'conf': (7, 25, 25, 256, 512, 3, 1, 1, 1)}[os.environ.get('SHAPE', 'conv2')]
x = torch.randn(G, N, H, W, Cin, device=dev); w = torch.randn(G * Cout, k * k * Cin, device=dev) * 0.02
b = torch.randn(G * Cout, device=dev); y = torch.empty(G, N, H, H, Cout, device=dev)
out = []
ws = torch.zeros(16 * G * N * H * H * Cout + 65536, device=dev)
for spec in sys.argv[2:] or ['54', '53', '55']:          # tile or tile:ksplit
    tile, ks = (int(v) for v in (spec.split(':') + ['1'])[:2])
    d = hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N=N, H=H, W=W, Cin=Cin, OH=H, OW=H, Cout=Cout,
                      KH=k, KW=k, pad=(pad, pad), dil=(dil, dil), act=1, tile=tile, ksplit=ks, ws=ws.data_ptr(),
                      groups=G, x_gs=N * H * W * Cin, w_gs=Cout * k * k * Cin, b_gs=Cout, y_gs=N * H * H * Cout,
                      w_frag=hip.tile_wfrag(tile),       # timing only: random filters are as good in any order
                      w_scale=b.data_ptr() if hip.tile_wfrag(tile) == 2 else None, x_split=hip.tile_xsplit(tile))
    for _ in range(5):
        hip.check(L.usot_conv2d_f32(hip.stream(), C.byref(d)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        hip.check(L.usot_conv2d_f32(hip.stream(), C.byref(d)))
    e1.record(); torch.cuda.synchronize()
    out.append('%s ks%d %.2f us' % (hip.tile_name(tile), ks, e0.elapsed_time(e1) / 200 * 1e3))
print('%-28s %s' % (os.path.basename(sys.argv[1]), ' | '.join(out)))
