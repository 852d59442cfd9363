"""A/B of layer3's conv2 -> conv3 at batch B: the two launches (256 x 256 implicit-GEMM tile 32, then the pixel-stationary panel
kernel) against the fused launch (csrc/conv_pw_lp.hip), isolated, rotating buffer sets so that nothing is served from the
Infinity Cache; then the whole low-precision backbone step with the engine option off / on (graph replay)."""
import argparse, ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import hip, synth
from usot_amd.model import USOT
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--lp', default='bf16')
ap.add_argument('--sets', type=int, default=3)
ap.add_argument('--iters', type=int, default=30)
a = ap.parse_args()
DEV = 'cuda:0'
dtype = torch.bfloat16 if a.lp == 'bf16' else torch.float16
dt = 0 if a.lp == 'bf16' else 1
n, h, cin, cm, co = a.batch, 31, 256, 256, 1024
M = n * h * h
g = torch.Generator().manual_seed(1)
w2 = (torch.randn(cm, 9 * cin, generator=g) / (9 * cin) ** 0.5).to(dtype).to(DEV)
w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype).to(DEV)
b2, b3 = (torch.randn(cm, generator=g) * 0.1).to(DEV), (torch.randn(co, generator=g) * 0.1).to(DEV)
sets = []
for s in range(a.sets):
    sets.append(dict(t1=torch.randn(n, h, h, cin, device=DEV).relu().to(dtype), res=torch.randn(M, co, device=DEV).to(dtype),
                     t2=torch.empty(n, h, h, cm, dtype=dtype, device=DEV), y=torch.empty(M, co, dtype=dtype, device=DEV),
                     y2=torch.empty(M, co, dtype=dtype, device=DEV)))
L = hip.lib()


def two(s):
    d2 = hip.conv_desc(s['t1'].data_ptr(), w2.data_ptr(), b2.data_ptr(), s['t2'].data_ptr(), N=n, H=h, W=h, Cin=cin, OH=h, OW=h, Cout=cm,
                       KH=3, KW=3, pad=(2, 2), dil=(2, 2), act=1, tile=32)
    hip.check(L.usot_conv2d_lp(hip.stream(), C.byref(d2), dt, 0), 'conv2')
    hip.check(L.usot_pw_panel_lp(hip.stream(), hip.ptr(s['t2']), hip.ptr(w3), hip.ptr(b3), hip.ptr(s['res']), hip.ptr(s['y2']), M, cm, co, 1, dt), 'conv3')


def conv2_only(s):
    d2 = hip.conv_desc(s['t1'].data_ptr(), w2.data_ptr(), b2.data_ptr(), s['t2'].data_ptr(), N=n, H=h, W=h, Cin=cin, OH=h, OW=h, Cout=cm,
                       KH=3, KW=3, pad=(2, 2), dil=(2, 2), act=1, tile=32)
    hip.check(L.usot_conv2d_lp(hip.stream(), C.byref(d2), dt, 0), 'conv2')


def fused(s, tile=4):
    d = hip.conv_desc(s['t1'].data_ptr(), w2.data_ptr(), b2.data_ptr(), None, N=n, H=h, W=h, Cin=cin, OH=h, OW=h, Cout=cm,
                      KH=3, KW=3, pad=(2, 2), dil=(2, 2), act=1, tile=tile)
    hip.check(L.usot_conv_pw_lp(hip.stream(), C.byref(d), hip.ptr(w3), hip.ptr(b3), hip.ptr(s['res']), hip.ptr(s['y']), dt), 'fused')


def timeit(fn):
    for s in sets: fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        fn(sets[i % len(sets)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


for s in sets:
    two(s); fused(s)
torch.cuda.synchronize()
print('bit-identical:', all(torch.equal(s['y'], s['y2']) for s in sets))
for rep in range(2):
    print('M=%d  conv2 alone %.1f us | two launches %.1f us | fused %.1f us | fused, row-shared k-loop %.1f us' % (M, timeit(conv2_only), timeit(two), timeit(fused), timeit(lambda s: fused(s, 0))))

# whole step
for widths, pair, rs, p5 in (((), False, False, False), ((256, 128), True, False, False), ((256, 128), True, True, False), ((256, 128), True, True, True), ((256, 128), True, True, False), ((256, 128), True, True, True), ((256, 128), True, False, True)):
    m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to(DEV)
    m.pr_pool = False
    e = m.engine
    e.opt['conv_pw_lp'] = widths
    e.opt['conv_pw_pair_lp'] = pair
    e.opt['conv_pw_rs'] = rs
    e.opt['conv_pw_p5_lp'] = p5
    x = torch.from_numpy(synth.crop(1, n, 255)).to(DEV)
    for _ in range(3): out = e.features_bf16(x, dtype=dtype)
    p = next(v for k, v in e._feat.items() if k[0] == ('bf16' if a.lp == 'bf16' else 'f16'))
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(100): p['plan'].run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 100 * 1e6)
    prof = p['plan'].profile(10)
    fz = [ms * 1e3 for k, *_, ms in prof if k in (29, 30)]
    print('conv_pw_lp=%s conv_pw_pair_lp=%s conv_pw_rs=%s conv_pw_p5_lp=%s: step %.1f us (best of 5 x 100 replays: %.1f); fused launches: %s' % (widths, pair, rs, p5, sorted(ts)[2], min(ts), ['%.1f' % v for v in fz]))
    del m, e, p
