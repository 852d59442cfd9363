"""The overlapped fused layer3 block (csrc/conv_pw_ov.hip) against the pair form it replaces (csrc/conv_pw_lp.hip, phase-5 form):
float64 parity on the rounded operands, T bit-identical to the tiled conv1 on its own Y, run-to-run determinism, then isolated timing
on rotating buffer sets (nothing served from the Infinity Cache) and the whole batch-64 bf16 step with the engine option off / on."""
import argparse, ctypes as C, os, sys
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
ap.add_argument('--no-step', action='store_true')
ap.add_argument('--trace', action='store_true')
a = ap.parse_args()
DEV = 'cuda:0'
dtype = torch.bfloat16 if a.lp == 'bf16' else torch.float16
dt = 0 if a.lp == 'bf16' else 1
L = hip.lib()


def case(n, h, dil, act2=1, seed=1):
    cm, co, cn = 256, 1024, 256
    M = n * h * h
    g = torch.Generator().manual_seed(seed)
    t1 = torch.randn(n, h, h, cm, generator=g).relu().to(dtype)
    w2 = (torch.randn(cm, 9 * cm, generator=g) / (9 * cm) ** 0.5).to(dtype)
    w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype)
    w1 = (torch.randn(cn, co, generator=g) / co ** 0.5).to(dtype)
    b2, b3, b1 = (torch.randn(c, generator=g) * 0.1 for c in (cm, co, cn))
    res = torch.randn(M, co, generator=g).to(dtype)
    dev = {k: v.to(DEV) for k, v in dict(t1=t1, w2=w2, w3=w3, w1=w1, b2=b2, b3=b3, b1=b1, res=res).items()}
    geo = dict(N=n, H=h, W=h, Cin=cm, OH=h, OW=h, Cout=cm, KH=3, KW=3, pad=(dil, dil), dil=(dil, dil), act=1)
    return dict(M=M, n=n, h=h, dil=dil, act2=act2, host=dict(t1=t1, w2=w2, w3=w3, w1=w1, b2=b2, b3=b3, b1=b1, res=res), dev=dev, geo=geo)


def run_ov(c, y, t, ws):
    d = c['dev']
    d2 = hip.conv_desc(d['t1'].data_ptr(), d['w2'].data_ptr(), d['b2'].data_ptr(), None, **c['geo'])
    pd = hip.pw_pair_desc(None, d['w3'].data_ptr(), d['b3'].data_ptr(), d['res'].data_ptr(), y.data_ptr(), d['w1'].data_ptr(), d['b1'].data_ptr(),
                          t.data_ptr(), c['M'], 256, 1024, 256, c['act2'])
    hip.check(L.usot_conv_pw_ov_lp(hip.stream(), C.byref(d2), C.byref(pd), dt, hip.ptr(ws)), 'usot_conv_pw_ov_lp')


def run_pair(c, y, t, tile=0):
    d = c['dev']
    d2 = hip.conv_desc(d['t1'].data_ptr(), d['w2'].data_ptr(), d['b2'].data_ptr(), None, tile=tile, **c['geo'])
    pd = hip.pw_pair_desc(None, d['w3'].data_ptr(), d['b3'].data_ptr(), d['res'].data_ptr(), y.data_ptr(), d['w1'].data_ptr(), d['b1'].data_ptr(),
                          t.data_ptr(), c['M'], 256, 1024, 256, c['act2'])
    hip.check(L.usot_conv_pw_pair_lp(hip.stream(), C.byref(d2), C.byref(pd), dt), 'usot_conv_pw_pair_lp')


def check(n, h, dil, act2=1):
    c = case(n, h, dil, act2, seed=n * 100 + h)
    M = c['M']
    ws = torch.zeros(int(L.usot_conv_pw_ov_ws_bytes(M)) // 4, dtype=torch.int32, device=DEV)
    y = torch.full((M + 2, 1024), 5.0, dtype=dtype, device=DEV)
    t = torch.full((M + 2, 256), 5.0, dtype=dtype, device=DEV)
    run_ov(c, y, t, ws)
    torch.cuda.synchronize()
    npan = (M + 127) // 128
    flags = ws[:2 * npan + 1].cpu()
    H = c['host']
    x64 = H['t1'].double().permute(0, 3, 1, 2)
    w64 = H['w2'].double().view(256, 3, 3, 256).permute(0, 3, 1, 2)
    t2r = torch.nn.functional.conv2d(x64, w64, H['b2'].double(), padding=dil, dilation=dil).relu().permute(0, 2, 3, 1).reshape(M, 256).to(dtype).double()
    ref = (t2r @ H['w3'].double().t() + H['b3'].double() + H['res'].double()).relu()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    ey = float(((y[:M].float().cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max())
    # T against the tiled conv1 on this launch's own Y
    tt = torch.empty(M, 256, dtype=dtype, device=DEV)
    yc = y[:M].contiguous()
    d1 = hip.conv_desc(yc.data_ptr(), c['dev']['w1'].data_ptr(), c['dev']['b1'].data_ptr(), tt.data_ptr(), N=1, H=M, W=1, Cin=1024, OH=M, OW=1,
                       Cout=256, KH=1, KW=1, act=act2, tile=32)
    hip.check(L.usot_conv2d_lp(hip.stream(), C.byref(d1), dt, 0), 'conv1')
    # and the pair form (row-shared k-loop, 64-channel chunks: another fp32 summation order)
    y2 = torch.empty(M, 1024, dtype=dtype, device=DEV); t2 = torch.empty(M, 256, dtype=dtype, device=DEV)
    run_pair(c, y2, t2)
    torch.cuda.synchronize()
    diff = float((y[:M] != y2).float().mean())
    print('n=%d h=%d dil=%d act2=%d: Y vs f64 %.2f ulp | T == conv1(Y): %s | past-the-end untouched: %s | flags zero: %s err %d | Y != pair form on %.4f'
          % (n, h, dil, act2, ey / ulp, bool(torch.equal(t[:M], tt)), bool(torch.all(y[M:] == 5.0) and torch.all(t[M:] == 5.0)),
             bool((flags[:-1] == 0).all()), int(flags[-1]), diff))
    return ey <= 4 * ulp and torch.equal(t[:M], tt)


ok = True
for n, h, dil, act2 in ((2, 31, 2, 1), (1, 12, 1, 0), (3, 33, 2, 1), (16, 31, 2, 1), (64, 31, 2, 1)):
    ok &= check(n, h, dil, act2)
print('parity:', 'OK' if ok else 'FAILED')

# determinism at the timed size
c = case(a.batch, 31, 2)
M = c['M']
ws = torch.zeros(int(L.usot_conv_pw_ov_ws_bytes(M)) // 4, dtype=torch.int32, device=DEV)
runs = []
for _ in range(6):
    y = torch.zeros(M, 1024, dtype=dtype, device=DEV); t = torch.zeros(M, 256, dtype=dtype, device=DEV)
    run_ov(c, y, t, ws)
    torch.cuda.synchronize()
    runs.append((y, t))
print('deterministic over 6 runs:', all(torch.equal(y, runs[0][0]) and torch.equal(t, runs[0][1]) for y, t in runs[1:]))

# isolated timing, rotating sets
sets = []
for s in range(a.sets):
    cs = case(a.batch, 31, 2, seed=10 + s)
    cs['y'] = torch.empty(M, 1024, dtype=dtype, device=DEV); cs['t'] = torch.empty(M, 256, dtype=dtype, device=DEV)
    cs['ws'] = torch.zeros(int(L.usot_conv_pw_ov_ws_bytes(M)) // 4, dtype=torch.int32, device=DEV)
    sets.append(cs)


def timeit(fn):
    for s in sets: fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        fn(sets[i % len(sets)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


for rep in range(3):
    print('batch %d: pair form (phase 5) %.1f us | overlapped form %.1f us' % (
        a.batch, timeit(lambda s: run_pair(s, s['y'], s['t'])), timeit(lambda s: run_ov(s, s['y'], s['t'], s['ws']))))

if a.trace:
    import numpy as np
    nb = 2 * 8 * ((M + 127) // 128 // 2 // 8 + 2)
    buf = torch.zeros(nb * 32 + 64, dtype=torch.int64, device=DEV)
    cs = sets[0]
    for it in range(3):
        buf.zero_()
        hip.check(L.usot_conv_pw_ov_trace(hip.ptr(buf)), 'trace')
        run_ov(cs, cs['y'], cs['t'], cs['ws'])
        torch.cuda.synchronize()
    hip.check(L.usot_conv_pw_ov_trace(None), 'trace off')
    d = buf.cpu().numpy()[:nb * 32].reshape(nb, 32)
    d = d[d[:, 3] != 0]
    t0 = min(int(r[3] & ((1 << 56) - 1)) for r in d)
    cu_key = lambda r: (int(r[1]) & 0xf, (int(r[2]) >> 8) & 0xff)           # (XCC, CU | SH | SE bits of HW_ID)
    by_cu = {}
    for r in d:
        by_cu.setdefault(cu_key(r), []).append(int(r[0]))
    kinds = {}
    for k, v in by_cu.items():
        kinds[tuple(sorted(v))] = kinds.get(tuple(sorted(v)), 0) + 1
    print('workgroups traced: %d on %d CUs; roles per CU (0 = M, 1 = H): %s' % (len(d), len(by_cu), kinds))
    names = {1: 'P1 begin', 2: 'P1 end', 3: 'P5 wait', 4: 'P5 begin', 6: 'P5 end', 7: 'P4 wait', 8: 'P4 begin', 9: 'P4 end'}
    for role in (0, 1):
        rows = [r for r in d if r[0] == role]
        ev = {}
        for r in rows:
            seq = [(int(v) >> 56, ((int(v) & ((1 << 56) - 1)) - t0) / 100.0) for v in r[3:] if v != 0]
            for i, (tag, us) in enumerate(seq):
                ev.setdefault((i, tag), []).append(us)
        print('role %s: %d workgroups; event (index, name): median us [min, max]' % ('H' if role else 'M', len(rows)))
        for (i, tag), v in sorted(ev.items()):
            v = np.array(v)
            print('   %2d %-9s n=%3d  %7.1f [%7.1f, %7.1f]' % (i, names.get(tag, tag), len(v), np.median(v), v.min(), v.max()))

if not a.no_step:
    for ov in (False, True, False, True):
        m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to(DEV)
        e = m.engine
        e.opt['conv_pw_ov_lp'] = ov
        x = torch.from_numpy(synth.crop(1, a.batch, 255)).to(DEV)
        for _ in range(3): out = e.features_bf16(x, dtype=dtype)
        p = next(v for k, v in e._feat.items() if k[0] in ('bf16', 'f16'))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): p['plan'].run()
        e1.record(); torch.cuda.synchronize()
        kinds = [k for k, *_ in p['plan'].profile(1)]
        print('step, conv_pw_ov_lp=%s: %.1f us (%d launches, %d overlapped)' % (ov, e0.elapsed_time(e1) / 50 * 1e3, len(kinds), kinds.count(31)))
        del m
