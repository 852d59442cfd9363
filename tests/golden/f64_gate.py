"""The acceptance rule for a float32 kernel variant:

    a variant is acceptable iff, on BOTH synthetic weight families, its distance from the float64 evaluation of the
    reference model does not exceed 1.5 x the distance of the reference's OWN float32 arithmetic from it.

tests/golden/golden_model_f64.npz (make_golden.py model64) holds every whole-model fixture of golden_model.npz for the
families 'zero_dc' and 'dc' twice: the reference on PyTorch-CPU in float32 and converted with net.double().  `run_model`
pushes the same inputs through a usot_amd.model.USOT on the GPU; `table` gives, per fixture, the scaled max / rms errors
    e_hip64 (HIP vs float64), e_ref64 (reference float32 vs float64), e_hip32 (HIP vs reference float32).
Scaled error as everywhere in this repo: |a - b| / max(|b|, mean|b|) over the fixture's sample points."""
import os

import numpy as np

from sampling import sample_index

GOLD = os.path.dirname(os.path.abspath(__file__))
FACTOR = 1.5
# rounding floor of the comparison itself: a float32 output cannot be closer than ~1 ulp to a float64 value
FLOOR = 2e-7


def load():
    with np.load(os.path.join(GOLD, 'golden_model_f64.npz')) as z:
        return {k: z[k] for k in z.files}


def run_model(net, dev='cuda:0'):
    """{fixture name: numpy array} for the fixture list of make_golden.do_model64 (same inputs, same order)."""
    import torch
    from usot_amd import synth
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    npy = lambda x: x.detach().float().cpu().numpy()
    out = {}
    for size, b, seed in ((127, 1, 0), (255, 1, 1), (271, 1, 3), (255, 2, 4)):
        stages, p3 = net.feature_extractor(t(synth.crop(seed, b, size)))
        tag = 'backbone_%d_b%d' % (size, b)
        for nm, ten in zip(('stem', 'p1', 'p2'), stages):
            out['%s/%s' % (tag, nm)] = npy(ten)
        out[tag + '/p3'] = npy(p3)
        out[tag + '/neck'] = npy(net.engine.features(t(synth.crop(seed, b, size))))
    pr = net.pr_pool
    net.pr_pool = False
    net.template(t(synth.crop(0, 1, 127)))
    out['template_crop/zf'] = npy(net.zf)
    x, mem = t(synth.crop(1, 1, 255)), t(synth.memory_kernels(7, 7))
    cls, bbox, _, _ = net.track(x)
    out['track_offline/cls'], out['track_offline/bbox'] = npy(cls), npy(bbox)
    sm = torch.full((1, 7), 0.9, device=dev)
    for tag, xx in (('track_mem', x), ('track_mem_271', t(synth.crop(3, 1, 271)))):
        res = net.track(xx, template_mem=mem, score_mem=sm)
        for nm, ten in zip(('cls', 'bbox', 'cls_mem', 'xf'), res):
            out['%s/%s' % (tag, nm)] = npy(ten)
    net.template(t(synth.crop(5, 2, 127)))
    res = net.track(t(synth.crop(4, 2, 255)), template_mem=t(synth.memory_kernels(8, 14)),
                    score_mem=torch.full((2, 7), 0.9, device=dev))
    for nm, ten in zip(('cls', 'bbox', 'cls_mem'), res):
        out['track_mem_b2/' + nm] = npy(ten)
    net.pr_pool = pr
    return out


def _err(a, b):
    d = np.abs(a - b) / np.maximum(np.abs(b), np.abs(b).mean() + 1e-300)
    return float(d.max()), float(np.sqrt((d * d).mean()))


def table(gold, family, outs):
    """[(name, (max, rms) hip-vs-f64, (max, rms) ref-f32-vs-f64, (max, rms) hip-vs-ref-f32)]"""
    rows = []
    for name, got in outs.items():
        f32, f64 = gold['%s/%s/f32' % (family, name)], gold['%s/%s/f64' % (family, name)]
        g = np.asarray(got, np.float64)
        if g.size > 8192:
            g = g.reshape(-1)[sample_index(name, g.size)]
        else:
            g = g.reshape(f64.shape)
        rows.append((name, _err(g, f64), _err(f32, f64), _err(g, f32)))
    return rows


def violations(rows, factor=FACTOR):
    """Fixtures whose HIP-vs-float64 error exceeds factor x the reference's own (max and rms are both gated)."""
    bad = []
    for name, h64, r64, _ in rows:
        if h64[0] > factor * r64[0] + FLOOR or h64[1] > factor * r64[1] + FLOOR:
            bad.append((name, h64, r64))
    return bad


def fmt(family, rows):
    lines = ['[%s] fixture                      HIP-f64 max/rms      ref32-f64 max/rms    ratio max/rms   HIP-ref32 max' % family]
    for name, h64, r64, h32 in rows:
        lines.append('[%s] %-28s %.2e / %.2e  %.2e / %.2e  %5.2f / %5.2f   %.2e' % (
            family, name, h64[0], h64[1], r64[0], r64[1], h64[0] / r64[0], h64[1] / r64[1], h32[0]))
    return '\n'.join(lines)
