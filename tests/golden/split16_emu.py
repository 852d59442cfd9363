"""CPU emulation of the split-fp16 convolution arithmetic (csrc/conv_igemm.hip PF = 4 / 5, usot_pw_pair_f32s) inside the torch-CPU
oracle: every convolution with K >= KMIN forms its products from hi + lo fp16 splits of power-of-two-scaled operands
(w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulation); the tracked frame's outputs are then scored against the float64 goldens with
the rule of f64_gate.py.  Test infrastructure (it runs the oracle): decided which layers may run split.
    python tests/golden/split16_emu.py f16x3 1152      # mode f32 | f16x3 | f16x4 | bf16x3, smallest K that is split"""
import sys, os
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import usot_oracle as orc
import f64_gate
from usot_amd import synth
from usot_amd.model import USOT
torch.set_num_threads(8)
MODE = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
KMIN = int(sys.argv[2]) if len(sys.argv) > 2 else 1152
real_conv = F.conv2d
def pow2_scale(t, target):
    m = float(t.abs().max())
    if m == 0: return 1.0
    return 2.0 ** np.floor(np.log2(target / m))
def split_conv(x, w, b, stride, pad, dil):
    sx, sw = pow2_scale(x, 2048.0), pow2_scale(w, 64.0)     # keep hi in the normal range, lo mostly normal
    xs, ws = x * sx, w * sw
    if MODE == 'f16x3':
        xh = xs.half().float(); xl = (xs - xh).half().float()
        wh = ws.half().float(); wl = (ws - wh).half().float()
        y = real_conv(xh, wh, None, stride, pad, dil) + real_conv(xl, wh, None, stride, pad, dil) + real_conv(xh, wl, None, stride, pad, dil)
    elif MODE == 'f16x4':
        xh = xs.half().float(); xl = (xs - xh).half().float()
        wh = ws.half().float(); wl = (ws - wh).half().float()
        y = real_conv(xh, wh, None, stride, pad, dil) + real_conv(xl, wh, None, stride, pad, dil) + real_conv(xh, wl, None, stride, pad, dil) + real_conv(xl, wl, None, stride, pad, dil)
    elif MODE == 'bf16x3':
        xh = xs.bfloat16().float(); xl = (xs - xh).bfloat16().float()
        wh = ws.bfloat16().float(); wl = (ws - wh).bfloat16().float()
        y = real_conv(xh, wh, None, stride, pad, dil) + real_conv(xl, wh, None, stride, pad, dil) + real_conv(xh, wl, None, stride, pad, dil)
    y = y / (sx * sw)
    if b is not None: y = y + b.view(1, -1, 1, 1)
    return y
stats = {'split': 0, 'plain': 0}
def patched(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    k = w.shape[1] * w.shape[2] * w.shape[3]
    if MODE != 'f32' and groups == 1 and k >= KMIN and w.shape[0] >= 64:
        stats['split'] += 1
        return split_conv(x, w, b, stride, padding, dilation)
    stats['plain'] += 1
    return real_conv(x, w, b, stride, padding, dilation, groups)
orc.F.conv2d = patched
gold = f64_gate.load()
t = torch.from_numpy
for fam in synth.FAMILIES:
    m = USOT()
    sd = synth.torch_state_dict(m, seed=0, calibrated=True, family=fam)
    with torch.no_grad():
        zf = orc.template(sd, t(synth.crop(0, 1, 127)), None, pr_pool=False)
        cls, bbox, cm, xf = orc.track(sd, t(synth.crop(1, 1, 255)), zf, t(synth.memory_kernels(7, 7)), torch.full((1, 7), 0.9))
    outs = {'track_mem/cls': cls.numpy(), 'track_mem/bbox': bbox.numpy(), 'track_mem/cls_mem': cm.numpy()}
    rows = f64_gate.table(gold, fam, outs)
    h32, h64, r64 = (max(r[i][0] for r in rows) for i in (3, 1, 2))
    print('%s %s kmin %d: emu-vs-ref-f32 %.2e, emu-vs-f64 %.2e, ref-f32-vs-f64 %.2e  violations %s  (%s)' % (MODE, fam, KMIN, h32, h64, r64, f64_gate.violations(rows), stats), flush=True)
