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


@pytest.fixture(scope='module', params=['exact_f32', 'split16'])
def net(request):
    """exact_f32: the default engine (exact fp32 products); split16: the opt-in split-fp16 products (engine option split16_f32) -
    every reference trajectory below is tracked on both arithmetic modes."""
    m = USOT()
    m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True)
    m.eval()
    if request.param == 'split16':
        m.engine_options['options'] = {'split16_f32': True}
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


GOLD_LONG = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_e2e_long.npz')


def _long_run(net, z, vid, fused, forced, nframes, capacity=None):
    """Our tracker over the long fixture's video.  forced: before every frame the tracker's position / size are set to the
    REFERENCE's state after the previous frame (teacher forcing), so that every frame is a one-step comparison on the
    reference's own crop window; the memory queue (features, confidences) is always our own, 500 frames deep."""
    seed, _, w, h = z['video%d/seed_frames_sz' % vid]
    want = z['video%d/track' % vid]
    trk = USOTTracker(Info())
    trk.fused = fused
    if capacity:
        net.engine.session_capacity = capacity
    try:
        im, (cx, cy) = synth.frame(int(seed), t=0)
        state = trk.init(im, np.array([cx, cy]), np.array([float(w), float(h)]), net)
        rows = [[cx, cy, w, h, 0.0]]
        for f in range(1, nframes):
            if forced:
                state['target_pos'] = want[f - 1, :2].copy()
                state['target_sz'] = want[f - 1, 2:4].copy()
            state = trk.track(state, synth.frame(int(seed), t=f)[0])
            rows.append([*state['target_pos'], *state['target_sz'], float(state['cls_score'])])
    finally:
        net.engine.session_capacity = 1024
    return np.array(rows), state


@pytest.mark.parametrize('vid', [0, 1])
@pytest.mark.parametrize('fused', [False, True], ids=['generic', 'fused'])
def test_500_frame_video_vs_reference_tracker(net, vid, fused):
    """BASELINE configs[3]: >= 500 frames per stream, so that the online memory queue — a list that grows by one feature per
    frame and is sampled by confidence rank (usot_tracker.py:222-265) — is exercised at depth.  tests/golden/
    golden_e2e_long.npz holds the reference tracker + reference model (CPU) over 500 frames at both instance sizes, with per
    frame the top-2 margin of the penalised score map, the rounding slack of the crop window and the top-2 slack of the
    confidence-ranked memory picks (make_golden.py e2e_long).

    With synthetic weights the tracker is chaotic: a one-pixel change of the crop window moves the next box by tens of pixels
    (measured: free-running, both trajectories agree to 3e-3 px and 5e-6 in score for 52 / 119 frames, then a 6e-3 px rounding
    slack flips one crop origin and they part for good).  So (1) free-running, the trajectories must agree until the first
    frame the fixture flags as a near-tie of ANY discrete decision; (2) teacher-forced on the reference's window, ALL 499 steps
    must agree wherever the argmax and the memory picks were clear — the memory queue always our own, 500 deep."""
    with np.load(GOLD_LONG) as zz:
        z = {k: zz[k] for k in zz.files}
    want = z['video%d/track' % vid]
    n = int(z['video%d/seed_frames_sz' % vid][1])
    assert n >= 500 and int(z['video%d/instance_size' % vid]) == (255, 271)[vid]
    m_tol, r_tol, p_tol = z['video%d/tolerances' % vid]
    mg, rs, ps = z['video%d/margins' % vid], z['video%d/round_slack' % vid], z['video%d/pick_slack' % vid]
    # (1) free-running up to the first near-tie (frame f is described by entry f-1 of the diagnostics)
    near = np.nonzero((mg < m_tol) | (rs < r_tol) | (ps < p_tol))[0] + 1
    first = int(near[0]) if len(near) else n
    assert first >= 10
    got, _ = _long_run(net, z, vid, fused, forced=False, nframes=first)
    np.testing.assert_allclose(got[:, :4], want[:first, :4], atol=2e-2, rtol=0)
    np.testing.assert_allclose(got[:, 4], want[:first, 4], atol=2e-4, rtol=0)
    # (2) teacher-forced over all 500 frames, three bank growths on the fused path (128 -> 1024 rows)
    got, state = _long_run(net, z, vid, fused, forced=True, nframes=n, capacity=128)
    assert state['p'].instance_size == (255, 271)[vid]
    clear = np.concatenate([[True], (mg >= m_tol) & (ps >= p_tol)])
    assert clear.mean() > 0.9, clear.mean()
    dpos = np.abs(got[:, :4] - want[:, :4]).max(1)
    dsc = np.abs(got[:, 4] - want[:, 4])
    bad = np.nonzero(clear & ((dpos > 2e-2) | (dsc > 2e-4)))[0]
    assert len(bad) == 0, (bad[:10], dpos[bad[:10]], dsc[bad[:10]])
    # near-ties may resolve either way, but not often: the score of the winner still matches on most of them
    assert (dsc[~clear] < 2e-4).mean() > 0.5 if (~clear).any() else True
    assert len(state['memory_confidences']) == n
    if fused:
        sess = state['session']
        assert sess.n == n and sess.cap >= n + 3 and len(state['memory_features']) == n
        np.testing.assert_allclose(np.asarray(state['memory_confidences'], np.float64)[1:], got[1:, 4], atol=0, rtol=0)


def test_500_frame_memory_bank_fused_equals_generic(net):
    """The device-resident bank after 500 appended features (grown 128 -> 1024 rows, the frame plan rebuilt three times) holds
    what the generic path keeps as a python list of tensors: same features, same confidences, same confidence-ranked picks."""
    with np.load(GOLD_LONG) as zz:
        z = {k: zz[k] for k in zz.files}
    n = int(z['video0/seed_frames_sz'][1])
    a, sa = _long_run(net, z, 0, False, forced=True, nframes=n)
    b, sb = _long_run(net, z, 0, True, forced=True, nframes=n, capacity=128)
    np.testing.assert_allclose(a, b, atol=1e-3, rtol=0)
    assert len(sa['memory_features']) == len(sb['memory_features']) == n
    for i in (0, 1, 2, 126, 127, 128, 300, n - 2, n - 1):                # around the growth points too
        fa, fb = sa['memory_features'][i].float().cpu().numpy(), sb['memory_features'][i].float().cpu().numpy()
        assert fa.shape == fb.shape == (1, 256, 7, 7)
        assert np.abs(fa - fb).max() <= 1e-4 * max(1.0, np.abs(fa).max()), i
    ca, cb = np.asarray(sa['memory_confidences'], np.float64), np.asarray(sb['memory_confidences'], np.float64)
    np.testing.assert_allclose(ca, cb, atol=1e-5, rtol=0)


def test_fused_equals_generic(net):
    a, _ = run(net, 12, 8, (52.0, 38.0), False)
    b, sb = run(net, 12, 8, (52.0, 38.0), True)
    np.testing.assert_allclose(a, b, atol=1e-3, rtol=0)
    # the session's bank rows are the memory features the generic path keeps as tensors
    assert sb['session'].n == 8


def test_session_bank_growth(net):
    """The memory bank doubles (and the frame plan is rebuilt) when a video outgrows it."""
    a, _ = run(net, 12, 14, (52.0, 38.0), False)
    net.engine.session_capacity = 8
    try:
        b, sb = run(net, 12, 14, (52.0, 38.0), True)
    finally:
        net.engine.session_capacity = 1024
    assert sb['session'].cap >= 16 and sb['session'].n == 14
    np.testing.assert_allclose(a, b, atol=1e-3, rtol=0)


def test_two_sessions_share_one_engine(net):
    """Two videos interleaved on one model: each session owns its template encodes and bank."""
    ref0, _ = run(net, 12, 5, (52.0, 38.0), True)
    ref1, _ = run(net, 17, 4, (16.0, 12.0), True)
    trk0, trk1 = USOTTracker(Info()), USOTTracker(Info())
    im0, (cx0, cy0) = synth.frame(12, t=0)
    im1, (cx1, cy1) = synth.frame(17, t=0)
    s0 = trk0.init(im0, np.array([cx0, cy0]), np.array([52.0, 38.0]), net)
    s1 = trk1.init(im1, np.array([cx1, cy1]), np.array([16.0, 12.0]), net)
    rows0, rows1 = [], []
    for f in range(1, 5):
        s0 = trk0.track(s0, synth.frame(12, t=f)[0])
        rows0.append([*s0['target_pos'], *s0['target_sz'], float(s0['cls_score'])])
        if f < 4:
            s1 = trk1.track(s1, synth.frame(17, t=f)[0])
            rows1.append([*s1['target_pos'], *s1['target_sz'], float(s1['cls_score'])])
    np.testing.assert_allclose(np.array(rows0), ref0[1:], atol=1e-3, rtol=0)
    np.testing.assert_allclose(np.array(rows1), ref1[1:], atol=1e-3, rtol=0)


def _open(net, seed):
    from usot_amd.tracker import USOTConfig
    p = USOTConfig()
    p.renew()
    p.sf_size = p.score_size
    t = lambda a: torch.from_numpy(a).cuda()
    net.pr_pool = True
    net.template(t(synth.crop(1000 + seed, 1, 127)), template_bbox=torch.tensor([[3.5, 3.5, 10.5, 10.5]]).cuda())
    crops = t(synth.crop(2000 + seed, 4, 255))
    roi = torch.tensor([[9.0, 9.0, 16.0, 16.0]]).cuda()
    feats = [net.extract_memory_feature(ori_x=crops[0:1], search_bbox=roi),
             net.extract_memory_feature(ori_x=crops[0:1].flip(3), search_bbox=roi)]
    window = np.outer(np.hanning(p.score_size), np.hanning(p.score_size))
    return net.engine.open_session(p, window, feats), crops


def test_sessions_on_separate_streams_overlap_safely(net):
    """submit()/collect(): two videos in flight at once on two HIP streams give bit-identical
    results to running each video alone with frame()."""
    picks = [[0, 0, 0, 0, 0], [0, 1, 0, 1, 0], [2, 1, 0, 2, 1]]
    want = []
    for seed in (3, 4):
        sess, crops = _open(net, seed)
        want.append([sess.frame(crops[i], picks[i], (63.5, 63.5)) for i in range(3)])
    group = [_open(net, seed) + (torch.cuda.Stream(),) for seed in (3, 4)]
    torch.cuda.synchronize()
    got = [[], []]
    for i in range(3):
        for sess, crops, st in group:
            with torch.cuda.stream(st):
                sess.submit(crops[i], picks[i], (63.5, 63.5))
        for k, (sess, crops, st) in enumerate(group):
            got[k].append(sess.collect())
    torch.cuda.synchronize()
    for k in range(2):
        np.testing.assert_array_equal(np.array(got[k]), np.array(want[k]))


def _fresh(split):
    m = USOT()
    m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True)
    m.eval()
    if split:
        m.engine_options['options'] = {'split16_f32': True}
    return m.cuda()


def test_session_frame_beyond_the_split16_window_reruns_on_exact_fp32():
    """A drop-in run must not die (or track on garbage) when a checkpoint drives ONE frame's activations beyond the fp16 window of
    the opt-in split-fp16 products: the frame graph's launches report it (usot_conv_desc.ovf -> decode's out[9]), the session rebuilds
    its graph on the exact-fp32 tiles and runs the SAME frame again.  (a) template, bank and memory encodings prepared by an exact
    engine, only the frame graph on split tiles: the out-of-range frame and every later one are BIT-EQUAL to an exact-fp32 session;
    (b) a session that lived on split tiles from the start tracks on through such a frame within the tracker's tolerance."""
    picks = [[0, 0, 0, 0, 0], [0, 1, 0, 1, 0], [2, 1, 0, 2, 1], [1, 2, 3, 1, 0]]
    ne = _fresh(False)
    se, crops = _open(ne, 5)
    # (a)
    na = _fresh(False)
    sa, _ = _open(na, 5)
    na.engine.opt['split16_f32'] = True
    sa._build()                                       # the frame graph, rebuilt on the split-fp16 tiles
    assert sa._ovf is not None and se._ovf is None
    frames = [crops[0] * 2000.0, crops[1], crops[2], crops[3]]
    for i, x in enumerate(frames):
        if i == 0:
            with pytest.warns(RuntimeWarning, match='split-fp16'):
                a = sa.frame(x, picks[i], (63.5, 63.5))
        else:
            a = sa.frame(x, picks[i], (63.5, 63.5))
        b = se.frame(x, picks[i], (63.5, 63.5))
        np.testing.assert_array_equal(a, b)
    assert na.engine.range_fallbacks == 1 and na.engine.opt['split16_f32'] is False and sa._ovf is None
    assert sa.n == se.n == 5
    torch.cuda.synchronize()
    assert torch.equal(sa.bank[:2 + sa.n], se.bank[:2 + se.n])
    for g in range(3):
        assert torch.equal(sa.bank_enc[g][:2 + sa.n], se.bank_enc[g][:2 + se.n])
    # (b)
    nb = _fresh(True)
    sb, _ = _open(nb, 5)
    se2, _ = _open(ne, 5)
    frames = [crops[0], crops[1], crops[2] * 2000.0, crops[3]]
    for i, x in enumerate(frames):
        a = sb.frame(x, picks[i], (63.5, 63.5))
        b = se2.frame(x, picks[i], (63.5, 63.5))
        fin = np.isfinite(b)
        assert (np.isfinite(a) == fin).all()
        np.testing.assert_allclose(a[fin], b[fin], atol=2e-3, rtol=1e-3)
        assert (nb.engine.range_fallbacks if hasattr(nb.engine, 'range_fallbacks') else 0) == (1 if i >= 2 else 0)
    assert sb.n == 5


def test_session_cached_encodings_equal_reencoding(net):
    """The session encodes a memory feature once, when it is appended; the reference re-encodes the picked
    features every frame (connect.py:251-255).  After a few frames every cached row must equal a fresh
    encoding of the bank row it belongs to."""
    sess, crops = _open(net, 9)
    for i in range(4):
        sess.frame(crops[i % 4], [0, min(i, 1), 0, min(i, 2), 0], (63.5, 63.5))
    sess.flush()                                     # engine option 'defer_append': the last frame's row is written by the NEXT graph
    torch.cuda.synchronize()
    rows = 2 + sess.n
    from usot_amd.engine import Builder
    bld = Builder(net.engine.W, net.engine.tuning, 0)
    src = bld.buf(rows, 7, 7, 256)
    src.copy_(sess.bank[:rows])
    enc = bld.encode_kernel(src, rows, 256, 'mem')
    bld.plan.run()
    torch.cuda.synchronize()
    for g in range(3):
        a, b = sess.bank_enc[g][:rows].cpu().numpy(), enc[g].cpu().numpy()
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max())


def test_deferred_append_equals_the_append_behind_the_tag():
    """engine option `defer_append`: frame t's bank append (encode the pooled feature, scatter it and its three encodings) runs
    inside frame t + 1's graph instead of behind frame t's result tag.  Mode 1 (side branch at the start of the graph): same
    kernels, same operands - results of a frame sequence whose picks always include the newest row, and the bank itself after
    flush(), must equal the undeferred session's BIT FOR BIT, also across a regrow of the bank and an append_feature() from
    outside.  Mode 2 (the default since round 6: the encoders ride in layer2's shortcut-conv launch on THAT launch's tile, one
    append + gather kernel in front of the heads): the encodings are the same sums in another split-K order, so everything agrees
    to fp32 rounding instead (1e-5 of the scale) - and the append + gather kernel's redirect (a picked row that is the row
    being appended) is exercised by every frame of the sequence."""
    from usot_amd.model import USOT
    from usot_amd import engine as _eng
    assert _eng.DEFAULT_OPTIONS['defer_append'] == 2
    res = {}
    for defer in (False, True, 2):
        m = USOT()
        m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True)
        m = m.eval().cuda()
        m.engine_options['options'] = {'defer_append': defer}
        m.engine.session_capacity = 8                # rows 0-1 + 5 memories, then the bank regrows (twice over 14 frames)
        sess, crops = _open(m, 11)
        assert sess.defer == defer
        outs = []
        for i in range(14):
            n = sess.n
            picks = [max(n - 1 - k, 0) for k in (3, 2, 1, 0, 0)]      # the newest memory (n - 1) is always among them
            outs.append(sess.frame(crops[i % 4], picks, (60.0 + i, 58.0)))
            if i == 6:
                sess.append_feature(sess.memory_feature(1).clone())
        last = sess.memory_feature(sess.n - 1).clone()               # flushes
        sess.flush()
        torch.cuda.synchronize()
        rows = 2 + sess.n
        res[defer] = (np.array(outs), last.cpu().numpy(), sess.bank[:rows].cpu().numpy(),
                      [b[:rows].cpu().numpy() for b in sess.bank_enc], sess.n)
    assert res[False][4] == res[True][4] == res[2][4] == 16
    np.testing.assert_array_equal(res[False][0], res[True][0])
    np.testing.assert_array_equal(res[False][1], res[True][1])
    np.testing.assert_array_equal(res[False][2], res[True][2])
    for a, b in zip(res[False][3], res[True][3]):
        np.testing.assert_array_equal(a, b)
    # mode 2: same values up to the summation order of the riding encoders
    assert np.array_equal(res[False][0][:, 0], res[2][0][:, 0])                       # the same argmax cell in every frame
    np.testing.assert_allclose(res[2][0], res[False][0], rtol=2e-4, atol=2e-4)
    for a, b in [(res[False][1], res[2][1]), (res[False][2], res[2][2])] + list(zip(res[False][3], res[2][3])):
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max())


def test_memory_features_is_a_view_of_the_bank_and_paths_can_switch(net):
    """usot_tracker.py:264 keeps one pooled tensor per frame in state['memory_features']; the fused
    path backs that list with the session's device bank (no None entries), and switching
    `fused` off mid-video carries on with the same queue."""
    ref, _ = run(net, 12, 9, (52.0, 38.0), False)
    trk = USOTTracker(Info())
    im, (cx, cy) = synth.frame(12, t=0)
    state = trk.init(im, np.array([cx, cy]), np.array([52.0, 38.0]), net)
    rows = [[cx, cy, 52.0, 38.0, 0.0]]
    for f in range(1, 9):
        trk.fused = f < 5 or f >= 7           # frames 5 and 6 run the generic update() on the same state
        state = trk.track(state, synth.frame(12, t=f)[0])
        rows.append([*state['target_pos'], *state['target_sz'], float(state['cls_score'])])
    feats = state['memory_features']
    assert len(feats) == 9 == len(state['memory_confidences']) and state['session'].n == 9
    for i in (0, 3, 5, 8, -1):
        t = feats[i]
        assert isinstance(t, torch.Tensor) and tuple(t.shape) == (1, 256, 7, 7) and t.is_cuda
    assert torch.equal(feats[-1], state['session'].memory_feature(8))
    assert len(feats[2:5]) == 3
    with pytest.raises(IndexError):
        feats[9]
    np.testing.assert_allclose(np.array(rows), ref, atol=1e-3, rtol=0)


def test_session_honours_mem_queue_size(net):
    """N_q comes from the yaml (USOTConfig.mem_queue_size); the device session is built for it
    instead of assuming 7, and a row count that does not match raises instead of corrupting the
    control block."""
    from usot_amd import hip
    from usot_amd.tracker import USOTConfig, select_memory
    out = {}
    for nq in (7, 5):
        p = USOTConfig()
        p.mem_queue_size = nq
        p.renew()
        p.sf_size = p.score_size
        t = lambda a: torch.from_numpy(a).cuda()
        net.pr_pool = True
        net.template(t(synth.crop(1021, 1, 127)), template_bbox=torch.tensor([[3.5, 3.5, 10.5, 10.5]]).cuda())
        crops = t(synth.crop(2021, 4, 255))
        roi = torch.tensor([[9.0, 9.0, 16.0, 16.0]]).cuda()
        feats = [net.extract_memory_feature(ori_x=crops[0:1], search_bbox=roi),
                 net.extract_memory_feature(ori_x=crops[0:1].flip(3), search_bbox=roi)]
        window = np.outer(np.hanning(p.score_size), np.hanning(p.score_size))
        sess = net.engine.open_session(p, window, feats)
        assert sess.nq == nq
        conf = [0.9]
        res = []
        for i in range(4):
            picks = select_memory(conf, nq)
            assert len(picks) == nq - 2
            o = sess.frame(crops[i], picks, (63.5, 63.5))
            conf.append(float(o[1]))
            res.append(o)
            # the generic model API on the same memory set gives the same response maps -> same decode
            mem = torch.cat([feats[0], feats[1]] + [sess.memory_feature(j) for j in picks], 0)
            cls, bbox, cm, xf = net.track(crops[i:i + 1], template_mem=mem, score_mem=torch.ones(1, nq).cuda())
            s = p.ratio * torch.sigmoid(cls) + (1 - p.ratio) * torch.sigmoid(cm)
            assert abs(float(s.reshape(-1)[int(o[0])]) - float(o[1])) < 2e-5
        out[nq] = np.array(res)
        if nq == 5:
            with pytest.raises(hip.HipError):
                sess.submit(crops[0], [0, 0, 0, 0, 0], (63.5, 63.5))
    assert out[5].shape == out[7].shape


def test_resident_crop_is_read_in_place(net):
    """submit(inplace=True): a float32 crop already on the device is not copied into the session's input buffer: its
    address travels in the control block and the frame graph's first convolution reads it in place.  Same results as the
    copying paths (host tensor; device tensor snapshot; non-contiguous device view), frame after frame, with the crop at
    odd byte offsets.  The DEFAULT is a snapshot: the caller may overwrite its buffer right after submit()."""
    from usot_amd import hip
    picks = [[0, 0, 0, 0, 0], [0, 1, 0, 1, 0], [2, 1, 0, 2, 1]]
    sess, crops = _open(net, 5)
    want = [sess.frame(crops[i].cpu(), picks[i], (63.5, 63.5)) for i in range(3)]          # host -> pinned -> copy
    sess, crops = _open(net, 5)
    sentinel = float(sess.x.abs().sum())
    got = [sess.frame(crops[i], picks[i], (63.5, 63.5), inplace=True) for i in range(3)]    # in place (slices of a batch)
    assert float(sess.x.abs().sum()) == sentinel          # the session's own buffer was never written
    sess, crops = _open(net, 5)
    wide = torch.zeros(4, 3, 255, 300, device=crops.device)
    wide[..., :255] = crops
    view = [sess.frame(wide[i, :, :, :255], picks[i], (63.5, 63.5)) for i in range(3)]      # strided view -> copy
    np.testing.assert_array_equal(np.array(got), np.array(want))
    np.testing.assert_array_equal(np.array(view), np.array(want))
    # default = snapshot: the double-buffer pattern (overwrite the crop between submit and collect) is safe
    sess, crops = _open(net, 5)
    snap = []
    for i in range(3):
        buf = crops[i].clone()
        sess.submit(buf, picks[i], (63.5, 63.5))
        buf.fill_(7.0)                                     # enqueued behind the snapshot copy on the same stream
        snap.append(sess.collect())
    np.testing.assert_array_equal(np.array(snap), np.array(want))
    # in-place crops must qualify: no silent copy, no foreign pointer handed to the kernel
    with pytest.raises(hip.HipError):
        sess.submit(wide[0, :, :, :255], picks[0], (63.5, 63.5), inplace=True)
    with pytest.raises(hip.HipError):
        sess.submit(crops[0].cpu(), picks[0], (63.5, 63.5), inplace=True)
    with pytest.raises(hip.HipError):
        sess.submit(crops[0].double(), picks[0], (63.5, 63.5), inplace=True)
