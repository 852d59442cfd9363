#!/usr/bin/env python3
"""Where the fused low-precision stem (7x7/s2 conv + BN + ReLU + max-pool, csrc/conv_bf16.hip stem_pool_lp_kernel) spends a
batch-64 launch: the shipped kernel against builds with parts removed (-DUSOT_SPABL_NOMMA / NOSTAGE / NOPOOL), bf16 and fp16.
`build` (container) writes build/variants/libusot_sp_*.so; `run` (GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'build', 'variants')
VARIANTS = {'full': [], 'minw3': ['-DUSOT_STEM_MINW=3'], 'nomma': ['-DUSOT_SPABL_NOMMA'], 'nostage': ['-DUSOT_SPABL_NOSTAGE'], 'nopool': ['-DUSOT_SPABL_NOPOOL'],
            'onlystage': ['-DUSOT_SPABL_NOMMA', '-DUSOT_SPABL_NOPOOL'], 'onlymma': ['-DUSOT_SPABL_NOSTAGE', '-DUSOT_SPABL_NOPOOL']}
if sys.argv[1:] == ['build']:
    sys.path.insert(0, ROOT)
    from usot_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(b.CSRC, 'conv_bf16.hip')
    others = [s[:-4] + '.o' for s in b.sources() if not s.endswith('conv_bf16.hip')]
    procs = []
    for name, flags in VARIANTS.items():
        obj = os.path.join(OUT, 'conv_bf16.%s.o' % name)
        procs.append((name, obj, subprocess.Popen([b._hipcc()] + b.FLAGS + flags + ['-c', src, '-o', obj])))
    for name, obj, p in procs:
        assert p.wait() == 0
        subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'libusot_sp_%s.so' % name), obj] + others)
        os.remove(obj)
    sys.exit(0)
if sys.argv[1:] == ['run']:
    for name in VARIANTS:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'one', name], env=dict(os.environ, USOT_HIP_LIB=os.path.join(OUT, 'libusot_sp_%s.so' % name)))
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from usot_amd import hip
from usot_amd.engine import pack_stem_lp
dev = 'cuda:0'
out = []
xs = [torch.rand(64, 3, 255, 255, device=dev) * 255 for _ in range(3)]
for dt in (torch.bfloat16, torch.float16):
    wf = pack_stem_lp(torch.randn(147, 64) * 0.01, dt).to(dev); bias = torch.randn(64, device=dev)
    run = lambda i: hip.stem_pool_lp(xs[i % 3], wf, bias, dtype=dt, mu=(104.0, 117.0, 123.0))
    for i in range(6): run(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(30): run(i)
    ev[1].record(); torch.cuda.synchronize()
    out.append('%s %6.1f us' % (str(dt).split('.')[1], ev[0].elapsed_time(ev[1]) / 30 * 1e3))
print('%-16s %s' % (sys.argv[2], '   '.join(out)), flush=True)
