"""Per-op profile of the batched mixed-precision frame (BASELINE configs[4] per-GPU share: fp16 backbone + head convs, fp32
xcorr, 32 streams): time per launch and TFLOP/s of the conv launches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import synth
from usot_amd.model import USOT
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to('cuda:0')
m.pr_pool = False
m.template(torch.from_numpy(synth.crop(0, B, 127)).cuda())
e = m.engine
x = torch.from_numpy(synth.crop(1, B, 255)).cuda()
mem = torch.randn(B * 7, 256, 7, 7, device='cuda:0') * 0.1
sm = torch.ones(B, 7, device='cuda:0')
for _ in range(3): e.track_mixed(x, m.zf, mem, sm, dtype=torch.float16)
p = [v for k, v in e._track.items() if k[0] == 'mixed'][0]
prof = p['plan'].profile(10)
convs = iter(p['log'])
tot = 0.0
for kind, tile, ks, groups, ms in prof:
    tot += ms
    if kind in (0, 11, 18, 22, 23, 24, 25, 26, 27, 28, 29, 30):
        try:
            name, M, N, K, g, macs = next(convs)
        except StopIteration:
            name, M, N, K, g, macs = '?', 0, 0, 0, 1, 0
        print('%-28s M=%7d N=%5d K=%5d g=%d tile %2d  %8.1f us %7.1f TF/s' % (name, M, N, K, g, tile, ms * 1e3, 2 * macs / ms / 1e9))
    else:
        print('op kind %2d %8.1f us' % (kind, ms * 1e3))
print('total %.1f us -> %.0f frames/s' % (tot * 1e3, B / tot * 1e3))
