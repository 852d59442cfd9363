"""Batch-64 bf16 backbone on 271 x 271 crops (273 panels of 256 pixels = a mostly empty second round: the fused kernels switch to
128-pixel panels, csrc/conv_pw_lp.hip: usot_conv_pw_pixels): fused against unfused lowering, time and bit-identity (per-tap k-loop).
Round 5 on one MI355X: unfused 3.549 ms in 32 launches, fused 3.367 ms in 16, bit-identical."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usot_amd import synth
from usot_amd.model import USOT
outs = []
for on in (False, True):
    m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to('cuda:0')
    e = m.engine
    e.opt['conv_pw_lp'] = (256, 128) if on else ()
    e.opt['conv_pw_pair_lp'] = e.opt['conv_pw_p5_lp'] = on
    e.opt['conv_pw_rs'] = False
    x = torch.from_numpy(synth.crop(5, 64, 271)).cuda()
    for _ in range(3): xf = e.features_bf16(x)
    p = next(v for k, v in e._feat.items() if k[0] == 'bf16')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): p['plan'].run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50 * 1e3
    kinds = [k for k, *_ in p['plan'].profile(1)]
    print('fused' if on else 'unfused', '271 batch 64: %.3f ms, launches %d, fused launches %d' % (dt, len(kinds), sum(k in (29, 30) for k in kinds)))
    outs.append(xf.clone()); del m, e, p
print('bit-identical:', torch.equal(outs[0], outs[1]))
