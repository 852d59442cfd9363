"""Kernel-development probe: time chosen (tile, ksplit) variants on chosen conv shapes."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import hip
dev = torch.device('cuda:0')
L = hip.lib()
tiles = hip.tile_table()
# (name, N,H,W,Cin,Cout,k,stride,pad,dil)
SHAPES = [('b7.ds', 1, 31, 31, 512, 1024, 3, 1, 1, 1), ('conf', 7, 25, 25, 256, 512, 3, 1, 1, 1),
          ('l3.conv2', 1, 31, 31, 256, 256, 3, 1, 2, 2), ('l3.conv1', 1, 31, 31, 1024, 256, 1, 1, 0, 1),
          ('big', 16, 31, 31, 512, 1024, 3, 1, 1, 1)]
variants = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]] or [(1, 1), (4, 1), (11, 1), (12, 1), (13, 1), (15, 1)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, N, H, W, Cin, Cout, k, st, pad, dil in SHAPES:
    OH = (H + 2 * pad - dil * (k - 1) - 1) // st + 1
    M, K = N * OH * OH, k * k * Cin
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, K, device=dev) * 0.02
    b = torch.randn(Cout, device=dev); y = torch.empty(N, OH, OH, Cout, device=dev)
    for tile, ks in variants:
        ws = torch.empty(ks * M * Cout, device=dev) if ks > 1 else None
        d = hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N=N, H=H, W=W, Cin=Cin, OH=OH, OW=OH, Cout=Cout,
                          KH=k, KW=k, stride=st, pad=(pad, pad), dil=(dil, dil), act=1, ksplit=ks, tile=tile,
                          ws=ws.data_ptr() if ws is not None else None)
        for _ in range(3): hip.check(L.usot_conv2d_f32(hip.stream(), C.byref(d)))
        # native launch loop: a ctypes call per launch (~7 us) would floor every short kernel
        plan = C.c_void_p(L.usot_plan_create())
        for _ in range(20): hip.check(L.usot_plan_add_conv(plan, C.byref(d)))
        us = 1e30
        for _ in range(3):
            e0.record(); hip.check(L.usot_plan_run(plan, hip.stream())); e1.record(); torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) / 20 * 1e3)
        L.usot_plan_destroy(plan)
        bm, bn = tiles[tile]
        blocks = -(-M // bm) * -(-Cout // bn) * ks
        print('%-9s M=%5d N=%4d K=%4d tile %2d (%3dx%-3d) ks %2d blocks %4d: %8.1f us %6.1f TFLOP/s' % (
            name, M, Cout, K, tile, bm, bn, ks, blocks, us, 2.0 * M * Cout * K / us / 1e6), flush=True)
