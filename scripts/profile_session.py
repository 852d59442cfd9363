"""Per-op timing of a device-resident Session frame (the bench.py default workload)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from usot_amd import hip
dev = torch.device('cuda:0')
model, _ = bench.build_model(0, 1, dev)
sess, crops, p = bench.open_stream(model, dev, seed=0)
conf = bench.Confidences()
bench.run_frames(sess, crops, p, conf, 20)
KINDS = {18: 'pw_pair (fused conv3 + next conv1)', 0: 'conv', 1: 'stem', 2: 'maxpool', 3: 'groupdw', 4: 'conf_reduce', 5: 'prroi', 6: 'permute', 7: 'decode', 10: 'rows', 15: 'rows_multi', 16: 'thin_conv (bbox_pred + cls_preds)', 17: 'stem_pool (fused)', 32: 'rows_append_gather (append previous feature + gather picks)'}
prof = sess.plan.profile(20)
convs = iter(sess.log)
tot = 0.0
for kind, tile, ks, groups, ms in prof:
    tot += ms
    name = KINDS.get(kind, str(kind))
    if kind == 0:
        nm, M, N, K, g, macs = next(convs)
        name = '%s M=%d N=%d K=%d g=%d tile=%s ks=%d' % (nm, M, N, K, g, hip.tile_name(tile) if tile else '?', ks)
    elif kind in (18, 19, 20, 21):
        nm, M, N, K, g, macs = next(convs)
        name = '%s M=%d (%s, %.2f GFLOP)' % (nm, M, 'fused pair' if kind == 18 else 'fused conv2 + pair' if kind == 21 else 'streaming N=%d K=%d' % (N, K), 2e-9 * macs)
    print('%8.1f us  %s' % (ms * 1e3, name))
print('sum %.1f us' % (tot * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
bench.run_frames(sess, crops, p, conf, 300)
print('session loop: %.1f us/frame' % ((time.perf_counter() - t0) / 300 * 1e6))
t0 = time.perf_counter()
for _ in range(300): sess.plan.run()
torch.cuda.synchronize()
print('graph only: %.1f us/frame' % ((time.perf_counter() - t0) / 300 * 1e6))
