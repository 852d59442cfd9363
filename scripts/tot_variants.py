#!/usr/bin/env python3
"""Builds libusot_hip with each arithmetic of the blocked-accumulation running total (common.h: USOT_TOT = 1 plain float32,
2 float64, 4 Kahan float32) into build/variants/ (run in the build container: `python scripts/tot_variants.py build`), and on
the GPU box (`python scripts/tot_variants.py run`) prints, per variant, the frame graph's replay time and the float64
acceptance table's worst ratios (tests/golden/f64_gate.py).  One process per variant (USOT_HIP_LIB selects the library)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'build', 'variants')
VARIANTS = (1, 2, 4)
if sys.argv[1:] == ['build']:
    sys.path.insert(0, ROOT)
    from usot_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    for v in VARIANTS:
        objs, procs = [], []
        for src in b.sources():
            obj = os.path.join(OUT, os.path.basename(src)[:-4] + '.tot%d.o' % v)
            objs.append(obj)
            procs.append(subprocess.Popen([b._hipcc()] + b.FLAGS + b.FILE_FLAGS.get(os.path.basename(src), []) + ['-DUSOT_TOT=%d' % v, '-c', src, '-o', obj]))
        assert all(p.wait() == 0 for p in procs)
        subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'libusot_tot%d.so' % v)] + objs)
        for o in objs:
            os.remove(o)
    sys.exit(0)
if sys.argv[1:] == ['run']:
    for v in VARIANTS:
        env = dict(os.environ, USOT_HIP_LIB=os.path.join(OUT, 'libusot_tot%d.so' % v))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'one', str(v)], env=env)
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import time, torch
import bench, f64_gate
from usot_amd import synth
from usot_amd.model import USOT
dev = torch.device('cuda:0')
model, _ = bench.build_model(0, 1, dev)
sess, crops, p = bench.open_stream(model, dev, seed=0)
conf = bench.Confidences()
bench.run_frames(sess, crops, p, conf, 50)
ts = []
for rnd in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): sess.plan.run()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 500 * 1e6)
gold = f64_gate.load()
line = 'USOT_TOT=%s graph replay %.1f us (min of 5 x 500)' % (sys.argv[2], min(ts))
for fam in ('zero_dc', 'dc'):
    m = USOT(); m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True, family=fam), strict=True); m.eval(); m = m.to(dev)
    rows = f64_gate.table(gold, fam, f64_gate.run_model(m))
    line += ' | %s: worst max ratio %.2f, worst rms ratio %.2f, mean rms ratio %.2f, worst HIP-ref32 %.2e' % (
        fam, max(r[1][0] / r[2][0] for r in rows), max(r[1][1] / r[2][1] for r in rows),
        sum(r[1][1] / r[2][1] for r in rows) / len(rows), max(r[3][0] for r in rows))
print(line, flush=True)
