#!/usr/bin/env python3
"""Where the pixel-stationary 1x1 kernel (csrc/pw_panel.hip) spends a layer3 conv3 launch (M = 61 504, 256 -> 1024, bf16): the
shipped kernel against builds with parts removed (-DUSOT_PNABL_NOMMA / NOSTORE / NORES / NODMA), three rotating buffer sets so
that nothing is served from the Infinity Cache.  `build` (container) writes build/variants/libusot_pn_*.so; `run` (GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'build', 'variants')
VARIANTS = {'full': [], 'nomma': ['-DUSOT_PNABL_NOMMA'], 'nostore': ['-DUSOT_PNABL_NOSTORE'], 'nores': ['-DUSOT_PNABL_NORES'],
            'nodma': ['-DUSOT_PNABL_NODMA'], 'nores_nostore': ['-DUSOT_PNABL_NORES', '-DUSOT_PNABL_NOSTORE'],
            'onlymma': ['-DUSOT_PNABL_NORES', '-DUSOT_PNABL_NOSTORE', '-DUSOT_PNABL_NODMA']}
if sys.argv[1:] == ['build']:
    sys.path.insert(0, ROOT)
    from usot_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(b.CSRC, 'pw_panel.hip')
    others = [s[:-4] + '.o' for s in b.sources() if not s.endswith('pw_panel.hip')]
    procs = []
    for name, flags in VARIANTS.items():
        obj = os.path.join(OUT, 'pw_panel.%s.o' % name)
        procs.append((name, obj, subprocess.Popen([b._hipcc()] + b.FLAGS + flags + ['-c', src, '-o', obj])))
    for name, obj, p in procs:
        assert p.wait() == 0
        subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'libusot_pn_%s.so' % name), obj] + others)
        os.remove(obj)
    sys.exit(0)
if sys.argv[1:] == ['run']:
    for name in VARIANTS:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'one', name], env=dict(os.environ, USOT_HIP_LIB=os.path.join(OUT, 'libusot_pn_%s.so' % name)))
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from usot_amd import hip
M, K, N = 61504, 256, 1024
dev = 'cuda:0'
sets = []
for i in range(3):
    sets.append((torch.randn(M, K, device=dev).bfloat16(), torch.randn(M, N, device=dev).bfloat16(), torch.empty(M, N, device=dev, dtype=torch.bfloat16)))
w = (torch.randn(N, K, device=dev) / 16).bfloat16(); b = torch.randn(N, device=dev)
def run(i):
    x, r, y = sets[i % 3]
    hip.check(hip.lib().usot_pw_panel_lp(hip.stream(), hip.ptr(x), hip.ptr(w), hip.ptr(b), hip.ptr(r), hip.ptr(y), M, K, N, 1, 0), 'panel')
for i in range(6): run(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for i in range(60): run(i)
ev[1].record(); torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) / 60 * 1e3
print('%-14s %7.1f us  (%.2f TB/s of the full kernel\'s 283 MB)' % (sys.argv[2], us, 283.4e6 / us / 1e6), flush=True)
