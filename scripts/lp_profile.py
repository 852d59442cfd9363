"""Per-op profile of the low-precision (bf16|fp16) backbone + neck at batch B: time, TFLOP/s and
algorithmic HBM GB/s (activations in + out + weights at 2 B/elem) per launch."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import synth
from usot_amd.model import USOT
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--lp', default='bf16')
a = ap.parse_args()
m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to('cuda:0')
m.pr_pool = False
m.template(torch.from_numpy(synth.crop(0, 1, 127)).cuda())
e = m.engine
B = a.batch
dt = torch.bfloat16 if a.lp == 'bf16' else torch.float16
x = torch.from_numpy(synth.crop(1, B, 255)).cuda()
for _ in range(3): e.features_bf16(x, dtype=dt)
p = next(v for k, v in e._feat.items() if k[:3] == ('bf16' if a.lp == 'bf16' else 'f16', B, 255))
prof = p['plan'].profile(10)
convs = iter(p['log'])
tot = 0.0; fl = 0.0
for kind, tile, ks, groups, ms in prof:
    tot += ms
    if kind in (11, 18, 22, 23, 24, 25, 26, 27, 28, 29, 30):
        name, M, N, K, g, macs = next(convs)
        byts = 2.0 * (M * K / (9 if K % 9 == 0 and K > 1024 else 1) + M * N + N * K) * g  # (fused pairs: first conv only)
        fl += 2 * macs
        print('%-16s M=%7d N=%5d K=%5d tile %d  %8.1f us %7.1f TF/s %7.0f GB/s' % (name, M, N, K, tile, ms * 1e3, 2 * macs / ms / 1e9, byts / ms / 1e6))
    else:
        print('op kind %d %8.1f us' % (kind, ms * 1e3))
print('total %.1f us, %.2f GFLOP, %.1f TF/s, %.0f crops/s' % (tot * 1e3, fl / 1e9, fl / tot / 1e9, B / tot * 1e3))
