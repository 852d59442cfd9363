"""GPU: the whole tracker (init + track over a synthetic video) behind the reference's
USOTTracker API, against trajectories produced by the REFERENCE tracker + REFERENCE model on
PyTorch-CPU (tests/golden/make_golden.py e2e; cv2.resize / imgaug flip / GPU-only PrRoIPool
substituted there as documented).  Both code paths: generic (model API + host decode) and
fused (device-resident session: decode + PrPool + memory bank inside the frame's graph)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from usot_amd import synth  # noqa: E402
from usot_amd.model import USOT  # noqa: E402
from usot_amd.tracker import USOTTracker  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_e2e.npz')


class Info:
    arch = 'USOT'
    dataset = 'SYNTH'
    epoch_test = False
    version = 'v1'


@pytest.fixture(scope='module')
def net():
    m = USOT()
    m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True)
    m.eval()
    return m.cuda()


def run(net, seed, nframes, sz, fused):
    trk = USOTTracker(Info())
    trk.fused = fused
    im, (cx, cy) = synth.frame(seed, t=0)
    state = trk.init(im, np.array([cx, cy]), np.array(sz), net)
    rows = [[cx, cy, sz[0], sz[1], 0.0]]
    for f in range(1, nframes):
        im, _ = synth.frame(seed, t=f)
        state = trk.track(state, im)
        rows.append([*state['target_pos'], *state['target_sz'], float(state['cls_score'])])
    return np.array(rows), state


@pytest.mark.parametrize('vid', [0, 1])
@pytest.mark.parametrize('fused', [False, True], ids=['generic', 'fused'])
def test_trajectory_vs_reference_tracker(net, vid, fused):
    with np.load(GOLD) as z:
        seed, nframes, w, h = z['video%d/seed_frames_sz' % vid]
        want = z['video%d/track' % vid]
        inst = int(z['video%d/instance_size' % vid])
    got, state = run(net, int(seed), int(nframes), (float(w), float(h)), fused)
    assert state['p'].instance_size == inst
    assert ('session' in state) == fused
    # positions / sizes in pixels: response maps agree to 1e-4 relative, so the same cell wins
    # and the decoded box differs by < 1e-2 px; scores to 1e-4
    np.testing.assert_allclose(got[:, :4], want[:, :4], atol=2e-2, rtol=0)
    np.testing.assert_allclose(got[:, 4], want[:, 4], atol=2e-4, rtol=0)
    assert len(state['memory_confidences']) == int(nframes)


def test_fused_equals_generic(net):
    a, _ = run(net, 12, 8, (52.0, 38.0), False)
    b, sb = run(net, 12, 8, (52.0, 38.0), True)
    np.testing.assert_allclose(a, b, atol=1e-3, rtol=0)
    # the session's bank rows are the memory features the generic path keeps as tensors
    assert sb['session'].n == 8
