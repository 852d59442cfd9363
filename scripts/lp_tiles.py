import sys, os
sys.path.insert(0, '/root/repo')
import torch
from usot_amd import hip
dev='cuda:0'
def run(name, N, H, Cin, Cout, k, pad, tiles):
    x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, k*k*Cin, device=dev) * 0.02).to(torch.bfloat16); b = torch.randn(Cout, device=dev)
    out=[]
    for tile in tiles:
        f = lambda: hip.conv2d_bf16(x, w, b, KH=k, KW=k, pad=(pad, pad), act=hip.ACT_RELU, tile=tile)
        try:
            for _ in range(3): f()
        except Exception as e:
            out.append('%d:err' % tile); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out.append('%d:%.0fus/%.0fTF' % (tile, us, 2.0 * N * H * H * Cout * k * k * Cin / us / 1e6))
    print(name, ' '.join(out))
tiles = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [21, 18, 22, 19, 15, 20, 16, 10, 17, 26]
run('b7.ds 3x3 512->1024', 64, 31, 512, 1024, 3, 1, tiles)
run('L3 conv2 3x3 256->256', 64, 31, 256, 256, 3, 1, tiles)
run('L3 conv1 1x1 1024->256', 64, 31, 1024, 256, 1, 0, tiles)
