#!/usr/bin/env python3
"""Where the fused first-bottleneck kernel (csrc/bneck_lp.hip) spends a launch (N images of 63 x 63 x 64, bf16): the shipped kernel
against builds with parts removed (-DUSOT_BKABL_NOMMA / NOSTORE / NODMA / NOD), three rotating buffer sets.
`build` (container) writes build/variants/libusot_bk_*.so; `run` (GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'build', 'variants')
VARIANTS = {'full': [], 'nomma': ['-DUSOT_BKABL_NOMMA'], 'nostore': ['-DUSOT_BKABL_NOSTORE'], 'nodma': ['-DUSOT_BKABL_NODMA'],
            'nod': ['-DUSOT_BKABL_NOD'], 'nodma_nostore': ['-DUSOT_BKABL_NODMA', '-DUSOT_BKABL_NOSTORE'],
            'onlymem': ['-DUSOT_BKABL_NOMMA', '-DUSOT_BKABL_NOD'],
            'skeleton': ['-DUSOT_BKABL_NOMMA', '-DUSOT_BKABL_NODMA', '-DUSOT_BKABL_NOSTORE']}
if sys.argv[1:] == ['build']:
    sys.path.insert(0, ROOT)
    from usot_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(b.CSRC, 'bneck_lp.hip')
    others = [s[:-4] + '.o' for s in b.sources() if not s.endswith('bneck_lp.hip')]
    procs = []
    for name, flags in VARIANTS.items():
        obj = os.path.join(OUT, 'bneck_lp.%s.o' % name)
        procs.append((name, obj, subprocess.Popen([b._hipcc()] + b.FLAGS + flags + ['-c', src, '-o', obj])))
    for name, obj, p in procs:
        assert p.wait() == 0
        subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'libusot_bk_%s.so' % name), obj] + others)
        os.remove(obj)
    sys.exit(0)
if sys.argv[1:] == ['run']:
    for name in VARIANTS:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'one', name], env=dict(os.environ, USOT_HIP_LIB=os.path.join(OUT, 'libusot_bk_%s.so' % name)))
    sys.exit(0)
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from usot_amd import hip
H, W = 63, 63
dev = 'cuda:0'
bf = torch.bfloat16
w1 = (torch.randn(64, 64, device=dev) / 8).to(bf); w2 = (torch.randn(64, 576, device=dev) / 24).to(bf)
w3c = (torch.randn(256, 128, device=dev) / 8).to(bf); wn = (torch.randn(64, 256, device=dev) / 16).to(bf)
b1, b2, b3c, bn = [torch.randn(n, device=dev) * 0.1 for n in (64, 64, 256, 64)]
out = []
for N in (64, 128, 256):                      # 8, 16, 32 tiles per workgroup: the slope is the cost of a tile, the intercept the launch + prologue
    sets = [(torch.randn(N, H, W, 64, device=dev).relu().to(bf), torch.empty(N, H, W, 256, device=dev, dtype=bf),
             torch.empty(N, H, W, 64, device=dev, dtype=bf)) for _ in range(3)]
    def run(i):
        x, y, t = sets[i % 3]
        d = hip.bneck_desc(*[hip.ptr(v) for v in (x, w1, b1, w2, b2, w3c, b3c, wn, bn, y, t)], N, H, W)
        hip.check(hip.lib().usot_bneck_first_lp(hip.stream(), C.byref(d), 0), 'bneck')
    for i in range(6): run(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(30): run(i)
    ev[1].record(); torch.cuda.synchronize()
    out.append(ev[0].elapsed_time(ev[1]) / 30 * 1e3)
print('%-14s N=64 %6.1f us  N=128 %6.1f us  N=256 %6.1f us   per tile %.2f us, fixed %.1f us' % (sys.argv[2], out[0], out[1], out[2], (out[2] - out[0]) / 24, out[0] - (out[2] - out[0]) / 3), flush=True)
