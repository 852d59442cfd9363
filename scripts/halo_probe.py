#!/usr/bin/env python3
"""Where the halo-tile 3x3 kernel (csrc/conv3x3_halo.hip) spends a layer1 conv2 launch (64 images of 63 x 63 x 64, bf16): the
shipped kernel against builds with parts removed (-DUSOT_HLABL_NOMMA / NOSTORE / NODMA / NOLDS), three rotating buffer sets.
`build` (container) writes build/variants/libusot_hl_*.so; `run` (GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'build', 'variants')
VARIANTS = {'full': [], 'nomma': ['-DUSOT_HLABL_NOMMA'], 'nostore': ['-DUSOT_HLABL_NOSTORE'], 'nodma': ['-DUSOT_HLABL_NODMA'],
            'nolds': ['-DUSOT_HLABL_NOLDS'], 'nodma_nostore': ['-DUSOT_HLABL_NODMA', '-DUSOT_HLABL_NOSTORE'],
            'onlymma': ['-DUSOT_HLABL_NODMA', '-DUSOT_HLABL_NOSTORE', '-DUSOT_HLABL_NOLDS'],
            'onlydma': ['-DUSOT_HLABL_NOMMA', '-DUSOT_HLABL_NOSTORE', '-DUSOT_HLABL_NOLDS']}
if sys.argv[1:] == ['build']:
    sys.path.insert(0, ROOT)
    from usot_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(b.CSRC, 'conv3x3_halo.hip')
    others = [s[:-4] + '.o' for s in b.sources() if not s.endswith('conv3x3_halo.hip')]
    procs = []
    for name, flags in VARIANTS.items():
        obj = os.path.join(OUT, 'conv3x3_halo.%s.o' % name)
        procs.append((name, obj, subprocess.Popen([b._hipcc()] + b.FLAGS + flags + ['-c', src, '-o', obj])))
    for name, obj, p in procs:
        assert p.wait() == 0
        subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'libusot_hl_%s.so' % name), obj] + others)
        os.remove(obj)
    sys.exit(0)
if sys.argv[1:] == ['run']:
    for name in VARIANTS:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'one', name], env=dict(os.environ, USOT_HIP_LIB=os.path.join(OUT, 'libusot_hl_%s.so' % name)))
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from usot_amd import hip
H, W, C = 63, 63, 64
dev = 'cuda:0'
w = (torch.randn(C, 9 * C, device=dev) / 24).bfloat16(); b = torch.randn(C, device=dev)
out = []
for N in (64, 128, 256):                      # 4, 8, 16 tiles per workgroup: the slope is the cost of a tile, the intercept the launch + prologue
    sets = [(torch.randn(N, H, W, C, device=dev).bfloat16(), torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)) for _ in range(3)]
    def run(i):
        x, y = sets[i % 3]
        hip.check(hip.lib().usot_conv3x3_halo_lp(hip.stream(), hip.ptr(x), hip.ptr(w), hip.ptr(b), hip.ptr(y), N, H, W, C, C, 1, 0), 'halo')
    for i in range(6): run(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(60): run(i)
    ev[1].record(); torch.cuda.synchronize()
    out.append(ev[0].elapsed_time(ev[1]) / 60 * 1e3)
print('%-14s N=64 %6.1f us  N=128 %6.1f us  N=256 %6.1f us   per tile %.2f us, fixed %.1f us' % (sys.argv[2], out[0], out[1], out[2], (out[2] - out[0]) / 12, out[0] - (out[2] - out[0]) / 3), flush=True)
