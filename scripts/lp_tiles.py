"""Big-K low-precision convolutions of the batch-64 backbone on a list of tile ids: time per launch (20 back-to-back
launches, HIP events), TFLOP/s, and bit-equality of every tile's output with the first one's.
    lp_tiles.py 21,32,33,34 [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usot_amd import hip
dev = 'cuda:0'


def run(name, N, H, Cin, Cout, k, pad, tiles, stride=1, dil=1, reps=20):
    x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, k * k * Cin, device=dev) * 0.02).to(torch.bfloat16); b = torch.randn(Cout, device=dev)
    out, ref = [], None
    for tile in tiles:
        f = lambda: hip.conv2d_bf16(x, w, b, KH=k, KW=k, stride=stride, pad=(pad, pad), dil=(dil, dil), act=hip.ACT_RELU, tile=tile)
        try:
            for _ in range(3): y = f()
        except Exception as e:
            out.append('%d:err(%s)' % (tile, str(e)[:40])); continue
        torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        same = bool(torch.equal(y, ref))
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        oh = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        out.append('%d:%.1fus/%.0fTF%s' % (tile, best, 2.0 * N * oh * oh * Cout * k * k * Cin / best / 1e6, '' if same else '/DIFF'))
    print(name, ' '.join(out), flush=True)


tiles = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [21, 18, 22, 19, 15, 20, 16, 10, 17, 26]
if len(sys.argv) <= 2:
  run('b7.ds 3x3 512->1024', 64, 31, 512, 1024, 3, 1, tiles)
  run('L3 conv2 3x3 256->256 d2', 64, 31, 256, 256, 3, 2, tiles, dil=2)
  run('b3.ds 3x3/s2 256->512', 64, 63, 256, 512, 3, 0, tiles, stride=2)
  run('L3 conv1 1x1 1024->256', 64, 31, 1024, 256, 1, 0, tiles)


def run_cold_1x1(tiles, reps=30):
    """Layer3's conv1 (1x1, 1024 -> 256, M = 61 504) with THREE rotating input / output sets (378 MB of inputs: nothing is served
    from the 256 MB Infinity Cache, as in the backbone graph where the input was just written by another launch) — the tiled
    kernel on `tiles` and the K-streaming kernel (csrc/pw_kstream.hip)."""
    M, K, N = 64 * 31 * 31, 1024, 256
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16); b = torch.randn(N, device=dev)
    xs = [torch.randn(64, 31, 31, K, device=dev).to(torch.bfloat16) for _ in range(3)]
    ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    out = []

    def timeit(f):
        for i in range(6): f(i)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps): f(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best
    ref = None
    for tile in tiles:
        def f(i, tile=tile):
            return hip.conv2d_bf16(xs[i % 3], w, b, KH=1, KW=1, act=hip.ACT_RELU, tile=tile)
        try:
            y = f(0)
        except Exception as e:
            out.append('%d:err' % tile); continue
        torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        us = timeit(f)
        out.append('%d:%.1fus/%.0fGB/s%s' % (tile, us, 2.0 * (M * K + M * N) / us / 1e3, '' if torch.equal(y, ref) else '/DIFF'))

    def g(i):
        hip.check(hip.lib().usot_pw_kstream_lp(hip.stream(), hip.ptr(xs[i % 3]), hip.ptr(w), hip.ptr(b), hip.ptr(ys[i % 3]), M, K, N, hip.ACT_RELU, 0), 'kstream')
    us = timeit(g)
    out.append('kstream:%.1fus/%.0fGB/s' % (us, 2.0 * (M * K + M * N) / us / 1e3))
    print('L3 conv1 1x1 1024->256 COLD', ' '.join(out), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == 'cold':
    run_cold_1x1(tiles)
if len(sys.argv) > 2 and sys.argv[2] == 'layer2':
    run('L2 conv2 3x3 128->128', 64, 31, 128, 128, 3, 1, tiles)
    run('b3.conv2 3x3/s2 128->128', 64, 63, 128, 128, 3, 0, tiles, stride=2)
