import sys, os, numpy as np, torch
sys.path[:0] = ['/root/repo', '/root/repo/tests']
from usot_amd import synth
from usot_amd.model import USOT
from usot_amd.tracker import USOTTracker
class Info: arch='USOT'; dataset='SYNTH'; epoch_test=False; version='v1'
m = USOT(); m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True); m.eval(); net = m.cuda()
z = np.load('/root/repo/tests/golden/golden_e2e_long.npz')
for vid in (0, 1):
    seed, nframes, w, h = z['video%d/seed_frames_sz' % vid]
    want = z['video%d/track' % vid]; mg = z['video%d/margins' % vid]; rs = z['video%d/round_slack' % vid]; ps = z['video%d/pick_slack' % vid]
    tol = z['video%d/tolerances' % vid]
    for fused in (True, False):
        trk = USOTTracker(Info()); trk.fused = fused
        im, (cx, cy) = synth.frame(int(seed), t=0)
        state = trk.init(im, np.array([cx, cy]), np.array([float(w), float(h)]), net)
        rows = [[cx, cy, w, h, 0.0]]
        for f in range(1, int(nframes)):
            im, _ = synth.frame(int(seed), t=f)
            state = trk.track(state, im)
            rows.append([*state['target_pos'], *state['target_sz'], float(state['cls_score'])])
        got = np.array(rows)
        dpos = np.abs(got[:, :2] - want[:, :2]).max(1); dsz = np.abs(got[:, 2:4] - want[:, 2:4]).max(1); dsc = np.abs(got[:, 4] - want[:, 4])
        amb = np.nonzero((mg < tol[0]) | (rs < tol[1]) | (ps < tol[2]))[0] + 1
        print('video', vid, 'fused', fused, 'ambiguous frames', amb[:20].tolist(), '...', len(amb))
        big = np.nonzero(dpos > 2e-2)[0]
        print('  first frame with pos dev > 2e-2:', big[:1], ' max pos dev %.3f max size dev %.3f max score dev %.2e' % (dpos.max(), dsz.max(), dsc.max()))
        for lo in range(0, 500, 50):
            print('   frames %3d-%3d: max pos dev %.4f size dev %.4f score dev %.2e' % (lo, lo + 49, dpos[lo:lo+50].max(), dsz[lo:lo+50].max(), dsc[lo:lo+50].max()))
