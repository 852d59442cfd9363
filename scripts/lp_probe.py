"""B=1 latency of the backbone+neck: fp32 MFMA path vs the fp16/bf16 MFMA path (graph replay)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import synth
from usot_amd.model import USOT

m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to('cuda:0')
e = m.engine if hasattr(m, 'engine') and m.engine is not None else None
m.pr_pool = False
for B in (1, 4):
    x = torch.from_numpy(synth.crop(1, B, 255)).cuda()
    m.template(torch.from_numpy(synth.crop(0, B, 127)).cuda())
    e = m.engine
    def bench(fn, n=200):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print('B=%d fp32 features %.1f us' % (B, bench(lambda: e.features(x))))
    print('B=%d fp16 features %.1f us' % (B, bench(lambda: e.features_bf16(x, dtype=torch.float16))))
    p = next(v for k, v in e._feat.items() if k[:3] == ('f16', B, 255))
    prof = p['plan'].profile(20)
    print('  lp ops:', ' '.join('%d:%.1f' % (k, ms * 1e3) for k, _, _, _, ms in prof))
    key = [k for k in e._feat if k[0] not in ('f16', 'bf16') and k[-1] == 255 or (len(k) == 2 and k[1] == 255)]
    print('  keys', list(e._feat.keys()))
