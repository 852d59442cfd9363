"""GPU parity of each C-ABI entry point against the CPU oracle (torch-CPU functional ops,
oracle/prroi_pool_ref.c) on seeded inputs.  Tolerances are scaled-relative:
|got-ref| / max(|ref|, mean|ref|) — fp32 MFMA is an exact fmaf chain, so the only
difference to the oracle is summation order (K up to 4608)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import usot_oracle as orc  # noqa: E402
from usot_amd import hip  # noqa: E402

DEV = 'cuda:0'


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.maximum(np.abs(ref), np.abs(ref).mean() + 1e-30)
    return float(np.max(np.abs(got - ref) / scale))


def pack_w(w):          # OIHW -> [O][kh][kw][I]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil
    (1, 64, 17, 19, 64, 1, 1, (0, 0), (1, 1)),
    (1, 64, 17, 19, 96, 3, 1, (1, 1), (1, 1)),
    (2, 128, 15, 15, 128, 3, 2, (0, 0), (1, 1)),      # layer2.0 geometry
    (1, 256, 13, 13, 256, 3, 1, (2, 2), (2, 2)),      # layer3.x geometry
    (1, 256, 11, 12, 256, 3, 1, (0, 0), (2, 1)),      # encoder (2,1)
    (1, 256, 12, 11, 256, 3, 1, (0, 0), (1, 2)),      # encoder (1,2)
    (1, 512, 9, 9, 1024, 3, 1, (1, 1), (1, 1)),       # layer3.0 shortcut geometry
    (3, 256, 7, 7, 512, 3, 1, (0, 0), (1, 1)),        # template-side encoders
    (1, 256, 25, 25, 4, 3, 1, (1, 1), (1, 1)),        # bbox_pred (Cout 4)
    (1, 256, 25, 25, 1, 3, 1, (1, 1), (1, 1)),        # cls_pred  (Cout 1)
    (1, 1024, 31, 31, 256, 1, 1, (0, 0), (1, 1)),     # neck, full size
    (2, 128, 21, 19, 128, 3, 1, (1, 1), (1, 1)),      # layer2 conv2 geometry, ragged last pixel tile, two images
    (7, 256, 25, 25, 512, 3, 1, (1, 1), (1, 1)),      # Conf_Fusion's conv: many pixel tiles per workgroup
]


def _all_tiles():
    """0 (heuristic) and every tile id of the library's table — the shipped tuning table
    (usot_amd/data/tuning_gfx950.json) may name any of them."""
    import ctypes
    from usot_amd import build
    n = ctypes.CDLL(build.LIB).usot_conv_tile_count() if os.path.exists(build.LIB) else 60
    from conftest import tile_params
    return tile_params(range(0, n + 1))         # ids outside the routed set: `experiments` marker (skipped on the default library)


def _tuned_ksplits():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(hip.__file__)), 'data', 'tuning_gfx950.json')) as f:
        ks = sorted({int(v[1]) for v in json.load(f).values() if int(v[1]) > 1})
    return sorted(set(ks) | {2, 3, 9})


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('tile', _all_tiles())
def test_conv_igemm(case, tile):
    N, Cin, H, W, Cout, k, stride, pad, dil = case
    if not hip.tile_supports(tile, Cin, Cout, k * k * Cin):
        pytest.skip('weight-stationary tile: one reduction length only')
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride, pad, dil)
    res = torch.randn_like(ref)
    ref_r = F.relu(ref + res)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd, bd = pack_w(w).to(DEV), b.to(DEV)
    y = hip.conv2d(xd, wd, bd, KH=k, KW=k, stride=stride, pad=pad, dil=dil, tile=tile)
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5
    rd = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = hip.conv2d(xd, wd, bd, KH=k, KW=k, stride=stride, pad=pad, dil=dil, res=rd, act=hip.ACT_RELU, tile=tile)
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref_r.numpy()) < 2e-5
    y = hip.conv2d(xd, wd, bd, KH=k, KW=k, stride=stride, pad=pad, dil=dil, tile=tile, y_nchw=True)
    assert rel_err(y.cpu().numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('tile', [91, 99, 106, 97, 94, 111])
def test_split_fp16_tiles_range_contract(tile):
    """The split-fp16 conv tiles (csrc/conv_igemm.hip PF = 4 / 5) promise fp32-level results for activations below 8 188 in
    magnitude whatever the filter rows' scale: activations spanning 1e-6 .. 5e3 inside one pixel row, filter rows of 1e-12, 1e+3
    and all zeros beside ordinary ones, negative values, split-K with the in-launch combine.  Against float64; the tolerance is
    the exact-fp32 tile's (2e-5 of the output scale, measured ~1e-6).  Above the range the output must show it (inf / nan), not
    return finite garbage."""
    g = torch.Generator().manual_seed(17 + tile)
    N, Cin, H, W, Cout, k = 1, 128, 13, 11, 128, 3
    x = torch.randn(N, Cin, H, W, generator=g)
    x[:, ::5] *= 1e-6
    x[:, 3::7] *= 5e3 / 4.5                     # |x| up to ~5e3
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    w[5] *= 1e-12
    w[6] *= 1e3
    w[7] = 0.0
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1).float()
    xd, wd, bd = x.permute(0, 2, 3, 1).contiguous().to(DEV), pack_w(w).to(DEV), b.to(DEV)
    for ks in (1, 3):
        y = hip.conv2d(xd, wd, bd, KH=k, KW=k, pad=(1, 1), tile=tile, ksplit=ks).permute(0, 3, 1, 2).cpu()
        assert torch.isfinite(y).all()
        # per output channel: the error is judged against that channel's own scale (rows differ by 15 orders of magnitude)
        err = (y - ref).abs().amax((0, 2, 3)) / (ref.abs().amax((0, 2, 3)) + 1e-30)
        err[7] = (y[:, 7] - ref[:, 7]).abs().max()          # the all-zero row: bias only, exact
        assert err.max() < 2e-5, (ks, err.max(), int(err.argmax()))
    xo = xd.clone()
    xo[0, 2, 3, 9] = 9.0e3                                   # beyond fp16 / 8: must be visible
    y = hip.conv2d(xo, wd, bd, KH=k, KW=k, pad=(1, 1), tile=tile)
    assert not torch.isfinite(y).all()
    # ... and REPORTED where an activation would hide it: with a ReLU (or Conf_Fusion's exp(clamp(relu))) the NaN of inf - inf
    # leaves the epilogue as a finite 0 / 1 (fmaxf(NaN, 0) = 0) - the sticky word of usot_conv_desc.ovf says so before that, for
    # the plain epilogue, the in-launch split-K combine and the split-K slabs alike; in range it stays 0
    for act in (hip.ACT_RELU, hip.ACT_CONF):
        for ks in (1, 3):
            ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
            hip.conv2d(xd, wd, bd, KH=k, KW=k, pad=(1, 1), tile=tile, ksplit=ks, act=act, ovf=ovf)
            assert int(ovf.item()) == 0, (act, ks)
            y = hip.conv2d(xo, wd, bd, KH=k, KW=k, pad=(1, 1), tile=tile, ksplit=ks, act=act, ovf=ovf)
            assert int(ovf.item()) == 1, (act, ks)
            assert torch.isfinite(y).all()                   # the hazard the word exists for: finite garbage
            hip.conv2d(xd, wd, bd, KH=k, KW=k, pad=(1, 1), tile=tile, ksplit=ks, act=act, ovf=ovf)
            assert int(ovf.item()) == 1                      # sticky: only the reader clears it
    xi = xd.clone()
    xi[0, 5, 5, 3] = float('inf')                            # a non-finite INPUT is reported too
    ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.conv2d(xi, wd, bd, KH=k, KW=k, pad=(1, 1), tile=tile, act=hip.ACT_RELU, ovf=ovf)
    assert int(ovf.item()) == 1


@pytest.mark.experiments
@pytest.mark.parametrize('tile2', [116, 117])
def test_split_maps_chain_two_convolutions_without_an_fp32_map(tile2):
    """usot_conv_desc.y_split / x_split: a split-fp16 tile writes its result as a split map (per pixel and 64-channel block the hi
    halves then the lo halves of 8 x value) and an all-DMA tile (PF = 6: both operands by LDS-DMA, no producer waves) reads it.
    Both levels against float64; the launcher refuses a split input on a tile that stages fp32 and an fp32 input on an all-DMA tile."""
    import ctypes as C
    g = torch.Generator().manual_seed(21)
    N, Cc, H, W = 2, 128, 14, 17
    x = torch.randn(N, Cc, H, W, generator=g).abs()
    w1 = torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc); b1 = torch.randn(Cc, generator=g)
    w2 = torch.randn(2 * Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc); b2 = torch.randn(2 * Cc, generator=g)
    r1 = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), 1, 2, 2))
    r2 = F.conv2d(r1, w2.double(), b2.double(), 1, 1).float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y1 = hip.conv2d(xd, pack_w(w1).to(DEV), b1.to(DEV), KH=3, KW=3, pad=(2, 2), dil=(2, 2), act=hip.ACT_RELU, tile=106, y_split=True)
    assert rel_err(hip.unsplit_map(y1).permute(0, 3, 1, 2).cpu().numpy(), r1.float().numpy()) < 2e-5
    w2p, sc = hip.split16_pack(pack_w(w2).to(DEV))
    b2d = b2.to(DEV)
    y2 = torch.empty(N, H, W, 2 * Cc, device=DEV)
    mk = lambda tile, xs: hip.conv_desc(y1.data_ptr(), w2p.data_ptr(), b2d.data_ptr(), y2.data_ptr(), N=N, H=H, W=W, Cin=Cc, OH=H, OW=W,
                                        Cout=2 * Cc, KH=3, KW=3, pad=(1, 1), tile=tile, w_frag=2, w_scale=sc.data_ptr(), x_split=xs)
    hip.check(hip.lib().usot_conv2d_f32(hip.stream(), C.byref(mk(tile2, 1))), 'usot_conv2d_f32')
    torch.cuda.synchronize()
    assert rel_err(y2.permute(0, 3, 1, 2).cpu().numpy(), r2.numpy()) < 2e-5
    assert hip.lib().usot_conv2d_f32(hip.stream(), C.byref(mk(tile2, 0))) != 0          # an all-DMA tile reads split maps only
    assert hip.lib().usot_conv2d_f32(hip.stream(), C.byref(mk(106, 1))) != 0            # ... and nobody else reads them


@pytest.mark.parametrize('ksplit', _tuned_ksplits())
def test_conv_splitk(ksplit):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 256, 25, 25, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) / 48
    b = torch.randn(256, generator=g)
    ref = F.relu(F.conv2d(x, w, b, 1, 1))
    y = hip.conv2d(x.permute(0, 2, 3, 1).contiguous().to(DEV), pack_w(w).to(DEV), b.to(DEV), KH=3, KW=3,
                   pad=(1, 1), act=hip.ACT_RELU, ksplit=ksplit, tile=4)
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('tile,ks', [(57, 4), (55, 2), (53, 3), pytest.param(38, 4, marks=pytest.mark.experiments)])
def test_conv_deferred_reduction_feeds_the_fused_pair(tile, ks):
    """usot_conv_desc.defer: a split-K convolution that writes its partial tiles and stops (no bias, no activation, no
    combine), and the fused fp32 pointwise pair that sums them - in part order, + bias, ReLU - while it stages its pixel tile
    (usot_pw_pair_desc.t2_parts).  (1) the slabs sum to the convolution; (2) the pair on the slabs is BIT-identical to the pair on
    the tile torch builds from the same slabs in the same order; layer3's geometry (31 x 31, 256 -> 256, dilation 2)."""
    import ctypes as C
    g = torch.Generator().manual_seed(500 + tile)
    N, H, Cin, Cout, CO, CN = 1, 31, 256, 256, 1024, 256
    x = torch.randn(N, Cin, H, H, generator=g).relu()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 48
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w, None, 1, 2, 2)
    xd, wd, bd = x.permute(0, 2, 3, 1).contiguous().to(DEV), pack_w(w).to(DEV), b.to(DEV)
    M = N * H * H
    ws = torch.zeros(ks * M * Cout + 4096, device=DEV)
    d = hip.conv_desc(xd.data_ptr(), wd.data_ptr(), None, xd.data_ptr(), N=N, H=H, W=H, Cin=Cin, OH=H, OW=H, Cout=Cout, KH=3, KW=3,
                      pad=(2, 2), dil=(2, 2), tile=tile, ksplit=ks, ws=ws.data_ptr(), defer=1)
    hip.check(hip.lib().usot_conv2d_f32(hip.stream(), C.byref(d)), 'conv (deferred)')
    slabs = ws[:ks * M * Cout].view(ks, M, Cout)
    got = slabs.sum(0).view(N, H, H, Cout).permute(0, 3, 1, 2).cpu()
    assert rel_err(got.numpy(), ref.numpy()) < 2e-5
    assert torch.equal(xd.cpu(), x.permute(0, 2, 3, 1).contiguous())          # y (aliased to x here) is not written
    d.ksplit = 1                                                               # defer without a split is refused
    assert hip.lib().usot_conv2d_f32(hip.stream(), C.byref(d)) != 0
    # the pair: conv3 256 -> 1024 + residual + ReLU, conv1 1024 -> 256 + ReLU
    w3 = torch.randn(CO, Cout, generator=g) / 16; b3 = torch.randn(CO, generator=g) * 0.1
    w1 = torch.randn(CN, CO, generator=g) / 32; b1 = torch.randn(CN, generator=g) * 0.1
    res = torch.randn(M, CO, generator=g).relu().to(DEV)
    w3p, w1p = hip.pw_pair_f32_pack(w3.to(DEV)), hip.pw_pair_f32_pack(w1.to(DEV))
    b3d, b1d = b3.to(DEV), b1.to(DEV)
    outs = []
    t2_sum = slabs[0].clone()
    for q in range(1, ks):
        t2_sum += slabs[q]                                                     # the kernel's order: part 0, 1, ... then the bias
    t2_sum = (t2_sum + bd).relu().contiguous()
    for t2, parts in ((slabs, ks), (t2_sum, 0)):
        y = torch.empty(M, CO, device=DEV); t = torch.empty(M, CN, device=DEV)
        wsp = hip.pw_pair_f32_ws(M, Cout, CO, CN, DEV)
        pd = hip.pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), b3d.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1d.data_ptr(),
                              t.data_ptr(), M, Cout, CO, CN, hip.ACT_RELU, wsp.data_ptr() if wsp is not None else None,
                              t2_parts=parts, t2_bias=bd.data_ptr() if parts else None)
        hip.check(hip.lib().usot_pw_pair_f32(hip.stream(), C.byref(pd)), 'pw_pair_f32')
        outs.append((y.cpu(), t.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    y_ref = (t2_sum.cpu().double() @ w3.double().t() + b3.double() + res.cpu().double()).relu()
    assert rel_err(outs[0][0].numpy(), y_ref.numpy()) < 2e-5
    # the RESIDUAL as deferred partial sums of the shortcut conv (res_parts / res_bias): parts in order, then the bias, no activation
    rparts = torch.randn(3, M, CO, generator=g).to(DEV)
    rbias = (torch.randn(CO, generator=g) * 0.1).to(DEV)
    rsum = ((rparts[0] + rparts[1]) + rparts[2] + rbias).contiguous()
    outs = []
    for rr, parts in ((rparts, 3), (rsum, 0)):
        y = torch.empty(M, CO, device=DEV); t = torch.empty(M, CN, device=DEV)
        wsp = hip.pw_pair_f32_ws(M, Cout, CO, CN, DEV)
        pd = hip.pw_pair_desc(t2_sum.data_ptr(), w3p.data_ptr(), b3d.data_ptr(), rr.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1d.data_ptr(),
                              t.data_ptr(), M, Cout, CO, CN, hip.ACT_RELU, wsp.data_ptr() if wsp is not None else None,
                              res_parts=parts, res_bias=rbias.data_ptr() if parts else None)
        hip.check(hip.lib().usot_pw_pair_f32(hip.stream(), C.byref(pd)), 'pw_pair_f32 (residual parts)')
        outs.append((y.cpu(), t.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.experiments
@pytest.mark.parametrize('tile', [72, 73, 76])
def test_conv_streamk_batch_of_problems(tile):
    """Persistent stream-K tiles (conv_igemm_f32_v3p): several problems of different geometry in ONE launch whose workgroups share
    the (tile, k-tile) units evenly - shares end inside tiles, so the slab / per-wave ticket combine runs - against torch, twice
    (the tickets must be back at zero), with residual + ReLU on one problem and groups on another."""
    import ctypes as C
    g = torch.Generator().manual_seed(700 + tile)
    probs = [(2, 256, 25, 25, 512, (1, 1), (1, 1), 1), (1, 256, 31, 31, 256, (0, 0), (2, 1), 1), (1, 256, 27, 29, 128, (2, 2), (2, 2), 3)]
    descs, refs, outs, keep = [], [], [], []
    for N, Cin, H, W, Cout, pad, dil, G in probs:
        x = torch.randn(G, N, Cin, H, W, generator=g)
        w = torch.randn(G * Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
        b = torch.randn(G * Cout, generator=g)
        ref = torch.stack([F.conv2d(x[q], w[q * Cout:(q + 1) * Cout], b[q * Cout:(q + 1) * Cout], 1, pad, dil) for q in range(G)])
        OH, OW = ref.shape[-2:]
        res = torch.randn(G, N, OH, OW, Cout, generator=g) if G == 1 and Cout == 512 else None
        if res is not None:
            ref = F.relu(ref + res.permute(0, 1, 4, 2, 3))
        xd, wd, bd = x.permute(0, 1, 3, 4, 2).contiguous().to(DEV), pack_w(w).to(DEV), b.to(DEV)
        y = torch.full((G, N, OH, OW, Cout), float('nan'), device=DEV)
        rd = res.to(DEV) if res is not None else None
        descs.append(hip.conv_desc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), N=N, H=H, W=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout,
                                   KH=3, KW=3, pad=pad, dil=dil, res=rd.data_ptr() if rd is not None else None,
                                   act=hip.ACT_RELU if rd is not None else hip.ACT_NONE, tile=tile, groups=G,
                                   x_gs=N * H * W * Cin, w_gs=Cout * 9 * Cin, b_gs=Cout, y_gs=N * OH * OW * Cout))
        refs.append(ref); outs.append(y); keep += [xd, wd, bd, rd]
    ws = hip.streamk_ws(descs, tile, DEV)
    arr = (hip.ConvDesc * len(descs))(*descs)
    for rep in range(2):
        for y in outs:
            y.fill_(float('nan'))
        hip.check(hip.lib().usot_conv2d_batch_f32(hip.stream(), arr, len(descs)), 'conv batch (stream-K)')
        for y, ref in zip(outs, refs):
            assert rel_err(y.permute(0, 1, 4, 2, 3).cpu().numpy(), ref.numpy()) < 2e-5, rep
    assert int(ws.numel()) > 0
    tickets = ws[-4 * 64:]                      # (the tail of the workspace is ticket words: all back at zero)
    assert float(tickets.abs().max()) == 0.0


def test_conv_activations():
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 64, 9, 9, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / 8
    b = torch.randn(32, generator=g)
    ref = F.conv2d(x, w, b, 1, 1)
    xd, wd, bd = x.permute(0, 2, 3, 1).contiguous().to(DEV), pack_w(w).to(DEV), b.to(DEV)
    y = hip.conv2d(xd, wd, bd, KH=3, KW=3, pad=(1, 1), act=hip.ACT_EXP)
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), torch.exp(ref).numpy()) < 2e-5
    y = hip.conv2d(xd, wd, bd, KH=3, KW=3, pad=(1, 1), act=hip.ACT_CONF)
    want = torch.exp(torch.clamp(F.relu(4 * ref) / 4 * 4, min=-6, max=4)) if False else torch.exp(torch.clamp(F.relu(ref), min=-6, max=4))
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), want.numpy()) < 2e-5


def test_conv_rejects_bad_geometry():
    x = torch.zeros(1, 5, 5, 48, device=DEV)                 # Cin not a multiple of 32
    w = torch.zeros(16, 48 * 9, device=DEV)
    with pytest.raises(hip.HipError):
        hip.conv2d(x, w, None, KH=3, KW=3)
    with pytest.raises(hip.HipError):
        hip.conv2d(torch.zeros(1, 5, 5, 64), torch.zeros(16, 64), None, KH=1, KW=1)   # CPU tensors


@pytest.mark.parametrize('size,n', [(127, 1), (255, 2), (271, 1), (64, 1)])
def test_stem_and_maxpool(size, n):
    g = torch.Generator().manual_seed(size)
    x = torch.rand(n, 3, size, size, generator=g) * 255
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    b = torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(x, w, b, 2, 0))
    wd = w.permute(1, 2, 3, 0).reshape(147, 64).contiguous().to(DEV)
    y = hip.stem_conv(x.to(DEV), wd, b.to(DEV))
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 1e-5
    p = hip.maxpool3x3s2(y)
    want = F.max_pool2d(y.permute(0, 3, 1, 2).cpu(), 3, 2, 1)
    assert torch.equal(p.permute(0, 3, 1, 2).cpu(), want)         # max-pool is exact
    # the offset form (x - mu[ci], bias + sum(w) * mu folded in float64) is the same convolution, closer to float64
    mu = (104.0, 117.0, 123.0)
    bmu = (b.double() + (w.double() * torch.tensor(mu, dtype=torch.float64).view(1, 3, 1, 1)).sum((1, 2, 3))).float()
    ymu = hip.stem_conv(x.to(DEV), wd, bmu.to(DEV), mu)
    ref64 = F.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 0)).numpy()
    e_mu, e_plain = rel_err(ymu.permute(0, 3, 1, 2).cpu().numpy(), ref64), rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref64)
    assert e_mu < 1e-5 and e_mu <= e_plain * 1.05 + 1e-7, (e_mu, e_plain)


XC = [(29, 29, 5, 5), (27, 29, 3, 5), (29, 27, 5, 3), (31, 31, 5, 5), (33, 31, 5, 3), (12, 9, 4, 2),
      (64, 64, 5, 5), (70, 70, 5, 5), (7, 7, 7, 7)]


@pytest.mark.parametrize('hx,wx,hk,wk', XC)
@pytest.mark.parametrize('planes', [(1, 1), (2, 24), (3, 257)])
def test_xcorr_depthwise_planes(hx, wx, hk, wk, planes):
    b, c = planes
    g = torch.Generator().manual_seed(hx * 100 + wk)
    x, k = torch.randn(b, c, hx, wx, generator=g), torch.randn(b, c, hk, wk, generator=g)
    ref = orc.xcorr_depthwise(x, k)
    out = hip.xcorr_depthwise(x.to(DEV), k.to(DEV))
    assert out.shape == ref.shape
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5


def test_xcorr_golden(gold_model):
    """the reference's own xcorr_depthwise outputs (tests/golden/make_golden.py)."""
    for i in range(5):
        out = hip.xcorr_depthwise(torch.from_numpy(gold_model['xcorr%d/x' % i]).to(DEV),
                                  torch.from_numpy(gold_model['xcorr%d/k' % i]).to(DEV))
        assert rel_err(out.cpu().numpy(), gold_model['xcorr%d/out' % i]) < 1e-5


@pytest.mark.parametrize('S,x_rep,OW,cols', [(1, 1, 25, 1), (7, 7, 25, 1), (14, 7, 25, 5), (3, 1, 27, 1),
                                             (2, 1, 25, 5), (4, 2, 27, 5), (9, 1, 25, 2), (7, 7, 27, 2), (3, 1, 25, 50), (9, 1, 25, 3), (14, 7, 27, 3), (5, 1, 25, 3), (9, 1, 25, 4), (14, 7, 27, 4), (5, 1, 25, 4),
                                             (9, 1, 25, 6), (14, 7, 27, 6), (5, 1, 25, 6), (70, 7, 25, 0), (66, 1, 27, 0),
                                             # 7: non-temporal LDS-DMA; 8 / 9: the persistent LDS-DMA kernel: odd sample counts
                                             # (a unit whose second sample does not exist), shared search maps, more units than
                                             # resident workgroups (131 samples -> 528 units over 512 slots: the cross-unit tap prefetch)
                                             (5, 1, 25, 7), (14, 7, 27, 7), (9, 1, 25, 8), (14, 7, 27, 8), (5, 1, 27, 8), (9, 1, 25, 9),
                                             (21, 7, 27, 9), (131, 1, 25, 8), (133, 7, 27, 9), (131, 1, 27, 9)])
def test_groupdw_fused(S, x_rep, OW, cols):
    g = torch.Generator().manual_seed(S * 31 + OW)
    XS = S // x_rep
    geo = ((5, 5), (3, 5), (5, 3))
    xs = [torch.randn(XS, 256, OW + hk - 1, OW + wk - 1, generator=g) for hk, wk in geo]
    zs = [torch.randn(S, 256, hk, wk, generator=g) for hk, wk in geo]
    wlog = torch.randn(3, generator=g)
    wsm = torch.softmax(wlog, 0)
    ref = 0
    for i in range(3):
        ref = ref + wsm[i] * orc.xcorr_depthwise(xs[i].repeat_interleave(x_rep, 0), zs[i])
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = hip.groupdw([nh(t) for t in xs], [nh(t) for t in zs], wsm.numpy(), x_rep=x_rep, cols=cols)
    assert rel_err(out.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize('cols', [0, 6, 8, 9])
def test_groupdw_three_segments_one_launch(cols):
    """A frame batch's three segments (reg, cls with their own search-map channel offsets, memory with x_rep = 7 and its own
    kernel maps) in ONE usot_groupdw_multi_f32 launch, as the engine issues them (engine.Builder.heads) - also on the persistent
    LDS-DMA kernel (modes 8 / 9), whose unit walk crosses segment borders; odd sample counts per segment."""
    g = torch.Generator().manual_seed(400 + cols)
    b, m, OW = 3, 7, 25
    geo = ((5, 5), (3, 5), (5, 3))
    es = [torch.randn(b, OW + hk - 1, OW + wk - 1, 512, generator=g).to(DEV) for hk, wk in geo]       # merged cls | reg search maps
    zk = [torch.randn(b, hk, wk, 512, generator=g).to(DEV) for hk, wk in geo]
    mk = [torch.randn(b * m, hk, wk, 256, generator=g).to(DEV) for hk, wk in geo]
    w_reg, w_cls = torch.softmax(torch.randn(3, generator=g), 0).numpy(), torch.softmax(torch.randn(3, generator=g), 0).numpy()
    outs = [torch.full((b, OW, OW, 256), float('nan'), device=DEV), torch.full((b, OW, OW, 256), float('nan'), device=DEV),
            torch.full((b * m, OW, OW, 256), float('nan'), device=DEV)]
    mk_desc = lambda xs, zs, out, wsm, S, rep, x_co, z_cs: hip.groupdw_desc(
        [t.data_ptr() for t in xs], [t.data_ptr() for t in zs], out.data_ptr(), wsm, S=S, x_rep=rep, OH=OW, OW=OW, Cc=256,
        x_cs=[512] * 3, x_co=[x_co] * 3, z_cs=[z_cs] * 3, z_co=[x_co if z_cs == 512 else 0] * 3, cols=cols)
    descs = [mk_desc(es, zk, outs[0], w_reg, b, 1, 256, 512), mk_desc(es, zk, outs[1], w_cls, b, 1, 0, 512),
             mk_desc(es, mk, outs[2], w_cls, b * m, m, 0, 256)]
    arr = (hip.GroupDWDesc * 3)(*descs)
    hip.check(hip.lib().usot_groupdw_multi_f32(hip.stream(), arr, 3), 'groupdw_multi')
    nchw = lambda t: t.permute(0, 3, 1, 2).cpu()
    for out, wsm, co, zs, rep in ((outs[0], w_reg, 256, [z[..., 256:] for z in zk], 1), (outs[1], w_cls, 0, [z[..., :256] for z in zk], 1),
                                  (outs[2], w_cls, 0, mk, m)):
        ref = 0
        for i in range(3):
            ref = ref + float(wsm[i]) * orc.xcorr_depthwise(nchw(es[i][..., co:co + 256]).repeat_interleave(rep, 0), nchw(zs[i]))
        assert rel_err(nchw(out).numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize('B,M', [(1, 7), (2, 7), (1, 1), (3, 4)])
def test_conf_fusion_reduce(B, M):
    g = torch.Generator().manual_seed(B * 10 + M)
    conf = torch.exp(torch.clamp(torch.randn(B, M, 256, 5, 6, generator=g) * 3, 0, 4))
    val = F.relu(torch.randn(B, M, 256, 5, 6, generator=g))
    ref = ((conf / conf.sum(1, keepdim=True)) * val).sum(1)
    cv = torch.cat([conf, val], 2).reshape(B * M, 512, 5, 6).permute(0, 2, 3, 1).contiguous().to(DEV)
    out = hip.conf_fusion_reduce(cv, B, M)
    assert rel_err(out.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 1e-5


ROIS = [[0, 2.3, 3.1, 9.7, 11.2], [1, -3.0, -2.0, 5.0, 6.0], [0, 10.0, 10.0, 40.0, 40.0], [1, 4.0, 4.0, 4.0, 9.0],
        [0, 0.0, 0.0, 14.0, 14.0], [1, 6.5, 6.5, 7.0, 7.0], [0, 50.0, 50.0, 60.0, 60.0]]


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_prroi_pool_vs_c_oracle(layout):
    g = torch.Generator().manual_seed(77)
    f = torch.randn(2, 256, 15, 17, generator=g)
    rois = torch.tensor(ROIS, dtype=torch.float32)
    ref = orc.prroi_pool(f, rois, 7, 7, 1.0)
    fd = f.to(DEV)
    if layout == 'nhwc':
        fd = fd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out = hip.prroi_pool(fd, rois.to(DEV), 7, 7, 1.0, out_nhwc=(layout == 'nhwc'))
    assert out.shape == ref.shape
    got = out.cpu().numpy()
    # same float32 operation order as the restated .cu: agree to a few ulp
    assert np.max(np.abs(got - ref.numpy())) < 2e-6 * max(1.0, float(ref.abs().max()))
    assert np.all(got[3] == 0) and np.all(got[6] == 0)        # zero-width roi, fully outside roi
    empty = hip.prroi_pool(fd, torch.zeros(0, 5, device=DEV), 7, 7, 1.0)
    assert tuple(empty.shape) == (0, 256, 7, 7)


def _prroi_tol(rois, ph=7, pw=7):
    rois = np.asarray(rois, np.float64)
    b = np.minimum((rois[:, 3] - rois[:, 1]) / pw, (rois[:, 4] - rois[:, 2]) / ph)
    return 2e-5 + 1e-6 / np.maximum(b, 1e-9)          # see tests/test_oracle_prpool_exact.py::tol


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_prroi_pool_vs_independent_float64_oracle(layout):
    """240 + 120 random RoIs (non-aligned, hanging over every edge, sub-pixel, oversized, degenerate)
    against oracle/prroi_exact.py — written from the operator's definition (separable hat-function
    integrals), sharing no formula with the kernel or with prroi_pool_ref.c."""
    import prroi_exact as ex
    from prroi_cases import random_rois
    for seed, (B, C, H, W), n in ((5, (2, 64, 15, 17), 240), (6, (1, 64, 31, 31), 120)):
        g = torch.Generator().manual_seed(seed)
        f = torch.randn(B, C, H, W, generator=g)
        rois = random_rois(seed, n, B, H, W)
        want = ex.prroi_pool_exact(f.numpy(), rois, 7, 7)
        fd = f.to(DEV)
        if layout == 'nhwc':
            fd = fd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        got = hip.prroi_pool(fd, torch.from_numpy(rois).to(DEV), 7, 7, 1.0, out_nhwc=(layout == 'nhwc')).cpu().numpy()
        err = np.abs(got - want).reshape(n, -1).max(1) / max(1.0, float(f.abs().max()))
        bad = err > _prroi_tol(rois)
        assert not bad.any(), (rois[bad], err[bad])
        # and the two float32 implementations (kernel, restated launcher) stay within a few ulp of each
        # other wherever the closed form does not cancel (bins of at least 0.05 pixel; below that both are
        # bounded by the eps / bin-size law above and may differ by fused vs unfused multiply-adds)
        ref = orc.prroi_pool(f, torch.from_numpy(rois), 7, 7, 1.0).numpy()
        wide = np.minimum(rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]) / 7 >= 0.05
        assert np.max(np.abs(got[wide] - ref[wide])) < 4e-6 * max(1.0, float(np.abs(ref).max()))
        d2 = np.abs(got - ref).reshape(n, -1).max(1) / max(1.0, float(f.abs().max()))
        assert not (d2 > 2 * _prroi_tol(rois)).any()


@pytest.mark.parametrize('cm,co,cn,M', [(64, 256, 64, 3969), (64, 256, 128, 3969), (128, 512, 128, 961), (128, 512, 256, 961),
                                        (64, 256, 64, 7), (128, 512, 128, 2 * 961), (256, 1024, 256, 961), (256, 1024, 256, 1089)])
def test_pw_pair_f32_equals_the_two_convolutions(cm, co, cn, M):
    """csrc/smallm_f32.hip: conv3 + residual + ReLU and the next block's conv1 + ReLU in one launch (fp32, batch-1
    frame) against the two conv launches it replaces and against torch in float64."""
    g = torch.Generator().manual_seed(cm + cn + M)
    t2 = torch.randn(M, cm, generator=g)
    w3 = torch.randn(co, cm, generator=g) / np.sqrt(cm)
    b3 = torch.randn(co, generator=g)
    res = torch.randn(M, co, generator=g)
    w1 = torch.randn(cn, co, generator=g) / np.sqrt(co)
    b1 = torch.randn(cn, generator=g)
    y64 = F.relu(t2.double() @ w3.double().t() + b3.double() + res.double())
    t64 = F.relu(y64 @ w1.double().t() + b1.double())
    d = lambda t: t.to(DEV)
    y, t = hip.pw_pair_f32(d(t2).reshape(1, 1, M, cm), d(w3), d(b3), d(res).reshape(1, 1, M, co), d(w1), d(b1))
    e_y, e_t = rel_err(y.reshape(M, co).cpu().numpy(), y64.numpy()), rel_err(t.reshape(M, cn).cpu().numpy(), t64.numpy())
    assert e_y < 1e-5 and e_t < 1e-5, (e_y, e_t)
    # the two launches of the unfused path
    y2 = hip.conv2d(d(t2).reshape(1, 1, M, cm), d(w3), d(b3), KH=1, KW=1, res=d(res).reshape(1, 1, M, co), act=hip.ACT_RELU)
    t2b = hip.conv2d(y2, d(w1), d(b1), KH=1, KW=1, act=hip.ACT_RELU)
    e_y, e_t = rel_err(y.cpu().numpy(), y2.cpu().numpy()), rel_err(t.cpu().numpy(), t2b.cpu().numpy())
    assert e_y < 1e-5 and e_t < 1e-5, (e_y, e_t)
    # the unsliced form of the same shape (the engine's default; the sliced one ran above when the shape has few pixel tiles)
    if cm < 256:                                             # (256, 1024, 256) exists in the sliced form only
        y3, t3 = hip.pw_pair_f32(d(t2).reshape(1, 1, M, cm), d(w3), d(b3), d(res).reshape(1, 1, M, co), d(w1), d(b1), sliced=False)
        assert torch.equal(y3, y) and rel_err(t3.cpu().numpy(), t.cpu().numpy()) < 1e-5
    # no activation on the second conv (the neck form)
    _, tn = hip.pw_pair_f32(d(t2).reshape(1, 1, M, cm), d(w3), d(b3), d(res).reshape(1, 1, M, co), d(w1), d(b1), act2=hip.ACT_NONE)
    assert rel_err(tn.reshape(M, cn).cpu().numpy(), (y64 @ w1.double().t() + b1.double()).numpy()) < 1e-5


@pytest.mark.parametrize('M', [961, 1089, 37])
def test_pw_pair_f32_split_fp16_operands(M):
    """usot_pw_pair_f32s: layer3's fused pair with every operand as hi + lo fp16 (three fp16 MFMAs per product block, fp32
    accumulation; banks pre-split by hip.pw_pair_s16_pack, the pixel tile and the Y tile split in the kernel) against float64
    at the fp32 kernel's own tolerance, and against the exact-fp32 form of the pair."""
    cm, co, cn = 256, 1024, 256
    g = torch.Generator().manual_seed(cm + cn + M)
    t2 = torch.randn(M, cm, generator=g).abs()
    w3 = torch.randn(co, cm, generator=g) / np.sqrt(cm)
    b3 = torch.randn(co, generator=g)
    res = torch.randn(M, co, generator=g)
    w1 = torch.randn(cn, co, generator=g) / np.sqrt(co)
    b1 = torch.randn(cn, generator=g)
    y64 = F.relu(t2.double() @ w3.double().t() + b3.double() + res.double())
    t64 = F.relu(y64 @ w1.double().t() + b1.double())
    d = lambda t: t.to(DEV)
    args = (d(t2).reshape(1, 1, M, cm), d(w3), d(b3), d(res).reshape(1, 1, M, co), d(w1), d(b1))
    y, t = hip.pw_pair_f32(*args, split16=True)
    e_y, e_t = rel_err(y.reshape(M, co).cpu().numpy(), y64.numpy()), rel_err(t.reshape(M, cn).cpu().numpy(), t64.numpy())
    assert e_y < 1e-5 and e_t < 1e-5, (e_y, e_t)
    y0, t0 = hip.pw_pair_f32(*args)
    assert rel_err(y.cpu().numpy(), y0.cpu().numpy()) < 1e-5 and rel_err(t.cpu().numpy(), t0.cpu().numpy()) < 1e-5
    # a second launch finds the slice tickets reset, and a very wide dynamic range inside one row of the tile survives the split
    t2w = t2.clone(); t2w[:, ::7] *= 1e-4; t2w[:, 3::11] *= 300.0
    y64 = F.relu(t2w.double() @ w3.double().t() + b3.double() + res.double())
    t64 = F.relu(y64 @ w1.double().t() + b1.double())
    ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
    y, t = hip.pw_pair_f32(d(t2w).reshape(1, 1, M, cm), *args[1:], split16=True, ovf=ovf)
    assert rel_err(y.reshape(M, co).cpu().numpy(), y64.numpy()) < 1e-5 and rel_err(t.reshape(M, cn).cpu().numpy(), t64.numpy()) < 1e-5
    assert int(ovf.item()) == 0
    # range contract (usot_pw_pair_desc.ovf): one staged value beyond the fp16 window - in the pixel tile (GEMM1's operand) or
    # produced INTO the Y tile (GEMM2's operand) - is reported in the sticky word although both ReLUs return finite numbers
    t2o = t2.clone(); t2o[M // 2, 17] = 9.0e3
    y, t = hip.pw_pair_f32(d(t2o).reshape(1, 1, M, cm), *args[1:], split16=True, ovf=ovf)
    assert int(ovf.item()) == 1 and torch.isfinite(y).all() and torch.isfinite(t).all()
    ovf.zero_()
    reso = res.clone(); reso[M // 3, 5] = 9.5e3              # Y = relu(.. + res) lands beyond the window: GEMM2 sees it
    hip.pw_pair_f32(args[0], args[1], args[2], d(reso).reshape(1, 1, M, co), args[4], args[5], split16=True, ovf=ovf)
    assert int(ovf.item()) == 1


@pytest.mark.parametrize('K,N,M,res', [(256, 1024, 961, True), (128, 512, 961, True), (1024, 256, 961, False), (512, 128, 1089, False),
                                       (256, 1024, 5, True)])
def test_pw_single_f32_streaming_conv(K, N, M, res):
    """csrc/smallm_f32.hip: pw_single_f32_kernel (small-M 1x1 conv, filters streamed in fragment order) against float64
    and against the tiled conv kernel it replaces for layer3's expansion convs at batch 1."""
    g = torch.Generator().manual_seed(K + N + M)
    x = torch.randn(1, 1, M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(1, 1, M, N, generator=g) if res else None
    ref = F.relu(x.double().reshape(M, K) @ w.double().t() + b.double() + (r.double().reshape(M, N) if res else 0)).numpy()
    d = lambda t: t.to(DEV) if t is not None else None
    y = hip.pw_single_f32(d(x), d(w), d(b), d(r), hip.ACT_RELU)
    assert rel_err(y.reshape(M, N).cpu().numpy(), ref) < 1e-5
    y2 = hip.conv2d(d(x), d(w), d(b), KH=1, KW=1, res=d(r), act=hip.ACT_RELU)
    assert rel_err(y.cpu().numpy(), y2.cpu().numpy()) < 1e-5
    y3 = hip.pw_single_f32(d(x), d(w), d(b), None, hip.ACT_NONE)
    assert rel_err(y3.reshape(M, N).cpu().numpy(), (x.double().reshape(M, K) @ w.double().t() + b.double()).numpy()) < 1e-5


@pytest.mark.parametrize('cin,n,nb,h,w,pad,dil', [(128, 128, 1, 31, 31, 1, 1), (256, 256, 1, 31, 31, 2, 2), (256, 256, 1, 15, 15, 1, 1),
                                                  (128, 128, 2, 9, 11, 2, 2), (128, 128, 1, 7, 5, 0, 1)])
def test_stream_conv3x3_f32(cin, n, nb, h, w, pad, dil):
    """csrc/smallm_f32.hip: stream_conv3x3_f32_kernel (3x3 / stride 1 at small M: im2col pixel tile in LDS, filters streamed
    in fragment order) against torch in float64 and against the tiled kernel."""
    g = torch.Generator().manual_seed(cin + h * w + pad)
    x = torch.randn(nb, cin, h, w, generator=g)
    w4 = torch.randn(n, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    b = torch.randn(n, generator=g)
    ref = F.relu(F.conv2d(x.double(), w4.double(), b.double(), 1, pad, dil)).numpy()
    xd, wd, bd = x.permute(0, 2, 3, 1).contiguous().to(DEV), pack_w(w4).to(DEV), b.to(DEV)
    y = hip.stream_conv3x3_f32(xd, wd, bd, (pad, pad), (dil, dil), None, hip.ACT_RELU)
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref) < 1e-5
    y2 = hip.conv2d(xd, wd, bd, KH=3, KW=3, pad=(pad, pad), dil=(dil, dil), act=hip.ACT_RELU)
    assert rel_err(y.cpu().numpy(), y2.cpu().numpy()) < 2e-5          # two float32 sums of up to 2304 products, each < 1e-5 from float64


@pytest.mark.parametrize('cin,cn,nb,h,w', [(64, 64, 1, 63, 63), (64, 128, 1, 63, 63), (64, 64, 2, 9, 7), (64, 128, 1, 67, 67),
                                            (128, 128, 1, 31, 31), (128, 128, 2, 6, 5)])
def test_pw_triple_f32_equals_three_convolutions(cin, cn, nb, h, w):
    """csrc/smallm_f32.hip: pw_triple_f32_kernel — layer1's conv2 (3x3) + conv3 + residual + ReLU + next conv1 in one launch —
    against float64 and against the three launches it replaces."""
    cm, co = cin, 4 * cin
    g = torch.Generator().manual_seed(cn + h)
    x = torch.randn(nb, cin, h, w, generator=g)
    w2 = torch.randn(cm, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    b2 = torch.randn(cm, generator=g)
    w3 = torch.randn(co, cm, generator=g) / np.sqrt(cm)
    b3 = torch.randn(co, generator=g)
    res = torch.randn(nb, h, w, co, generator=g)
    w1 = torch.randn(cn, co, generator=g) / np.sqrt(co)
    b1 = torch.randn(cn, generator=g)
    t2 = F.relu(F.conv2d(x.double(), w2.double(), b2.double(), 1, 1, 1)).permute(0, 2, 3, 1)
    y64 = F.relu(t2 @ w3.double().t() + b3.double() + res.double())
    t64 = F.relu(y64 @ w1.double().t() + b1.double())
    d = lambda t: t.to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y, t = hip.pw_triple_f32(xd, pack_w(w2).to(DEV), d(b2), d(w3), d(b3), d(res), d(w1), d(b1))
    e_y, e_t = rel_err(y.cpu().numpy(), y64.numpy()), rel_err(t.cpu().numpy(), t64.numpy())
    assert e_y < 1e-5 and e_t < 1e-5, (e_y, e_t)
    t2d = hip.conv2d(xd, pack_w(w2).to(DEV), d(b2), KH=3, KW=3, pad=(1, 1), act=hip.ACT_RELU)
    y2, tt = hip.pw_pair_f32(t2d, d(w3), d(b3), d(res), d(w1), d(b1))
    assert rel_err(y.cpu().numpy(), y2.cpu().numpy()) < 1e-5 and rel_err(t.cpu().numpy(), tt.cpu().numpy()) < 1e-5


def _prroi_grad_case(seed, shape, n):
    from prroi_cases import random_rois
    B, C, H, W = shape
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(B, C, H, W, generator=g)
    rois = random_rois(seed, n, B, H, W)
    top_diff = torch.randn(n, C, 7, 7, generator=g)
    return f, rois, top_diff


@pytest.mark.parametrize('seed,shape,n', [(11, (2, 32, 15, 17), 96), (12, (1, 16, 31, 31), 60)])
def test_prroi_feature_gradient_vs_oracles(seed, shape, n):
    """prroi_backward_kernel (one atomic per touched pixel, separable weights) against the float64 adjoint oracle
    (oracle/prroi_exact.py) and the float32 restatement of the reference's kernel (prroi_pooling_gpu_impl.cu:214-272).
    Measured (scripts/prroi_grad_errors.py): 1e-6 .. 9e-6 of the largest gradient for bins of at least 0.05 pixel,
    3e-5 .. 5e-5 with the sub-pixel boxes (float32 corner weights divided by a tiny area, in the restatement alike)."""
    import prroi_exact as ex
    f, rois, g = _prroi_grad_case(seed, shape, n)
    rd = torch.from_numpy(rois).to(DEV)
    got = hip.prroi_pool_backward(f.shape, rd, g.to(DEV), 7, 7, 1.0).cpu().numpy()
    want = ex.prroi_pool_exact_backward(f.shape, rois, g.numpy(), 7, 7, 1.0)
    ref = orc.prroi_pool_backward(f.shape, rois, g, 7, 7, 1.0).numpy()
    scale = max(1.0, float(np.abs(want).max()))
    assert np.isfinite(got).all()
    assert np.max(np.abs(got - want)) < 1.5e-4 * scale
    assert np.max(np.abs(got - ref)) < 1e-5 * scale                     # the two float32 forms agree everywhere
    wide = torch.from_numpy(np.minimum(rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]) / 7 >= 0.05)
    gw = hip.prroi_pool_backward(f.shape, rd[wide.to(DEV)], g[wide].to(DEV), 7, 7, 1.0).cpu().numpy()
    ww = ex.prroi_pool_exact_backward(f.shape, rois[wide.numpy()], g.numpy()[wide.numpy()], 7, 7, 1.0)
    assert np.max(np.abs(gw - ww)) < 2.5e-5 * max(1.0, float(np.abs(ww).max()))
    # an empty RoI list gives a zero gradient of the right shape
    z = hip.prroi_pool_backward(f.shape, torch.zeros(0, 5, device=DEV), torch.zeros(0, shape[1], 7, 7, device=DEV), 7, 7, 1.0)
    assert tuple(z.shape) == tuple(f.shape) and not z.any()


@pytest.mark.parametrize('seed,shape,n', [(13, (2, 32, 15, 17), 96), (14, (1, 16, 31, 31), 60)])
def test_prroi_roi_gradient_vs_oracles(seed, shape, n):
    """prroi_coor_backward_kernel against the float64 Leibniz oracle and the float32 restatement (.cu:274-380).
    d out / d edge carries 1 / area on top of the forward's 1 / area: measured 5e-6 of the RoI's largest component for
    bins of at least 0.2 pixel, 3e-5 from 0.05, 1e-2 for sub-pixel boxes (restatement and kernel alike)."""
    import prroi_exact as ex
    f, rois, g = _prroi_grad_case(seed, shape, n)
    fd, rd = f.to(DEV), torch.from_numpy(rois).to(DEV)
    top = hip.prroi_pool(fd, rd, 7, 7, 1.0)
    got = hip.prroi_pool_coor_backward(fd, rd, top, g.to(DEV), 7, 7, 1.0).cpu().numpy()
    want = ex.prroi_pool_exact_coor_backward(f.numpy(), rois, g.numpy(), 7, 7, 1.0)
    ref = orc.prroi_pool_coor_backward(f, rois, top.cpu(), g, 7, 7, 1.0).numpy()
    assert np.all(got[:, 0] == 0) and np.isfinite(got).all()
    bins = np.minimum(rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]) / 7
    mag = np.maximum(1.0, np.abs(want).max(1))
    e_exact, e_ref = np.abs(got - want).max(1) / mag, np.abs(got - ref).max(1) / mag
    assert np.all(e_exact[bins >= 0.2] <= 3e-5) and np.all(e_ref[bins >= 0.2] <= 3e-5), (e_exact[bins >= 0.2].max(), e_ref[bins >= 0.2].max())
    assert np.all(e_exact[bins >= 0.05] <= 1.5e-4) and np.all(e_ref[bins >= 0.05] <= 1.5e-4)
    assert np.all(e_exact <= 5e-2)
    assert not got[~(bins > 0)].any()                                     # zero / negative extent: no gradient


def test_prroi_autograd_function_matches_the_oracles():
    """lib.models.prroi_pool.functional.prroi_pool2d as the reference's training code calls it
    (functional.py:41-84): autograd gradients of features and RoIs."""
    import prroi_exact as ex
    from lib.models.prroi_pool.functional import prroi_pool2d
    f, rois, g = _prroi_grad_case(21, (2, 8, 13, 12), 12)
    rois = rois[np.minimum(rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]) > 2.0]
    g = g[:len(rois)]
    fd = f.to(DEV).requires_grad_(True)
    rd = torch.from_numpy(rois).to(DEV).requires_grad_(True)
    out = prroi_pool2d(fd, rd, 7, 7, 0.5)
    (out * g.to(DEV)).sum().backward()
    want_f = ex.prroi_pool_exact_backward(f.shape, rois, g.numpy(), 7, 7, 0.5)
    want_r = ex.prroi_pool_exact_coor_backward(f.numpy(), rois, g.numpy(), 7, 7, 0.5)
    assert np.max(np.abs(fd.grad.cpu().numpy() - want_f)) < 1e-5 * max(1.0, np.abs(want_f).max())
    assert np.max(np.abs(rd.grad.cpu().numpy() - want_r)) < 3e-5 * max(1.0, np.abs(want_r).max())
    # features only: the RoI gradient is not computed
    fd2 = f.to(DEV).requires_grad_(True)
    prroi_pool2d(fd2, torch.from_numpy(rois).to(DEV), 7, 7, 0.5).sum().backward()
    assert fd2.grad is not None


def test_prroi_gradient_reference_symbols_exact_signature():
    """PrRoIPoolingBackwardGpu / PrRoIPoolingCoorBackwardGpu with the reference's signatures
    (prroi_pooling_gpu_impl.cuh:30-54) as prroi_pooling_gpu.c:46-113 calls them; both zero their output first."""
    import ctypes as C
    f, rois, g = _prroi_grad_case(31, (2, 24, 15, 17), 10)
    rois = rois[np.minimum(rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]) > 1.0]
    g = g[:len(rois)].contiguous()
    n = len(rois)
    fd, rd, gd = f.to(DEV), torch.from_numpy(rois).to(DEV), g.to(DEV)
    top = hip.prroi_pool(fd, rd, 7, 7, 1.0)
    L = hip.lib()
    gin = torch.full(f.shape, 3.0, device=DEV)                               # must be overwritten, not accumulated into
    L.PrRoIPoolingBackwardGpu(hip.stream(), hip.ptr(fd), hip.ptr(rd), hip.ptr(top), hip.ptr(gd), hip.ptr(gin),
                              24, 15, 17, 7, 7, C.c_float(1.0), top.numel(), gin.numel())
    gr = torch.full((n, 5), 3.0, device=DEV)
    L.PrRoIPoolingCoorBackwardGpu(hip.stream(), hip.ptr(fd), hip.ptr(rd), hip.ptr(top), hip.ptr(gd), hip.ptr(gr),
                                  24, 15, 17, 7, 7, C.c_float(1.0), top.numel(), gr.numel())
    torch.cuda.synchronize()
    ref_f = orc.prroi_pool_backward(f.shape, rois, g, 7, 7, 1.0).numpy()
    ref_r = orc.prroi_pool_coor_backward(f, rois, top.cpu(), g, 7, 7, 1.0).numpy()
    assert np.max(np.abs(gin.cpu().numpy() - ref_f)) < 1e-5 * max(1.0, float(np.abs(ref_f).max()))
    assert np.max(np.abs(gr.cpu().numpy() - ref_r)) < 3e-5 * max(1.0, float(np.abs(ref_r).max()))
    # inconsistent counts are refused: a line on stderr, nothing written, no exit
    keep = torch.full((n, 5), 9.0, device=DEV)
    L.PrRoIPoolingCoorBackwardGpu(hip.stream(), hip.ptr(fd), hip.ptr(rd), hip.ptr(top), hip.ptr(gd), hip.ptr(keep),
                                  24, 15, 17, 7, 7, C.c_float(1.0), top.numel(), keep.numel() - 5)
    torch.cuda.synchronize()
    assert float(keep.min()) == 9.0


def test_prroi_reference_symbol_exact_signature():
    """PrRoIPoolingForwardGpu(stream, bottom_data, bottom_rois, top_data, C, H, W, PH, PW, scale,
    top_count) — the reference's own extern "C" symbol (prroi_pooling_gpu_impl.cuh:20-28) as a
    binding compiled from prroi_pooling_gpu.c:22-44 would call it."""
    import ctypes as C
    g = torch.Generator().manual_seed(78)
    f = torch.randn(2, 48, 15, 17, generator=g)
    rois = torch.tensor(ROIS, dtype=torch.float32)
    ref = orc.prroi_pool(f, rois, 7, 7, 1.0)
    fd, rd = f.to(DEV), rois.to(DEV)
    out = torch.zeros(len(ROIS), 48, 7, 7, device=DEV)                     # at::zeros in the binding
    L = hip.lib()
    L.PrRoIPoolingForwardGpu(hip.stream(), hip.ptr(fd), hip.ptr(rd), hip.ptr(out), 48, 15, 17, 7, 7,
                             C.c_float(1.0), out.numel())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.max(np.abs(got - ref.numpy())) < 2e-6 * max(1.0, float(ref.abs().max()))
    # spatial_scale is applied to the roi, pooled size is free
    out2 = torch.zeros(len(ROIS), 48, 3, 5, device=DEV)
    L.PrRoIPoolingForwardGpu(hip.stream(), hip.ptr(fd), hip.ptr(rd), hip.ptr(out2), 48, 15, 17, 3, 5,
                             C.c_float(0.5), out2.numel())
    ref2 = orc.prroi_pool(f, rois, 3, 5, 0.5)
    assert np.max(np.abs(out2.cpu().numpy() - ref2.numpy())) < 2e-6 * max(1.0, float(ref2.abs().max()))
    # a top_count that is not a whole number of RoIs is refused (stderr line, no launch, no exit)
    out3 = torch.full((3, 48, 7, 7), 7.0, device=DEV)
    L.PrRoIPoolingForwardGpu(hip.stream(), hip.ptr(fd), hip.ptr(rd), hip.ptr(out3), 48, 15, 17, 7, 7,
                             C.c_float(1.0), out3.numel() - 1)
    torch.cuda.synchronize()
    assert float(out3.min()) == 7.0


def test_decode_nan_follows_numpy_argmax():
    """np.argmax (usot_tracker.py:163) returns the FIRST NaN when the penalised score holds any,
    else the first maximum; the kernel must do the same and never index with a sentinel
    (round-1 advisor finding: every pscore NaN -> read far out of bounds)."""
    p = orc.Hyper(255)
    S = p.score_size
    n = S * S
    window = torch.from_numpy(np.outer(np.hanning(S), np.hanning(S))).reshape(-1).to(DEV)
    g = torch.Generator().manual_seed(3)
    cls, cm = torch.randn(n, generator=g), torch.randn(n, generator=g)
    bbox = torch.rand(4, n, generator=g) * 40 + 5
    run = lambda c, m, b: hip.decode(c.to(DEV), m.to(DEV), b.to(DEV), window, S, 255, 8, p.ratio, p.penalty_k,
                                     p.window_influence, 60.0, 50.0).cpu().numpy()
    base = run(cls, cm, bbox)
    assert 0 <= int(base[0]) < n
    nan = float('nan')
    allnan = run(torch.full((n,), nan), torch.full((n,), nan), bbox)
    assert int(allnan[0]) == 0 and np.isnan(allnan[7])
    for where in ([400], [17, 300, 599], [624], [255, 256]):          # across the 256-thread stripes
        c2 = cls.clone()
        c2[where] = nan
        o = run(c2, cm, bbox)
        assert int(o[0]) == min(where), (where, o[0])
    b2 = bbox.clone()
    b2[:, 77] = nan                                                   # NaN box -> NaN penalty -> NaN pscore
    assert int(run(cls, cm, b2)[0]) == 77
    # ties: first maximum wins
    flat = run(torch.zeros(n), torch.zeros(n), torch.full((4, n), 20.0))
    w = window.cpu().numpy()
    assert int(flat[0]) == int(np.argmax(w))


def test_permutes_roundtrip():
    t = torch.randn(3, 40, 7, 9)
    d = t.to(DEV)
    nh = hip.to_nhwc(d)
    assert torch.equal(nh.cpu(), t.permute(0, 2, 3, 1))
    back = hip.to_nchw(nh.contiguous())
    assert torch.equal(back.cpu(), t)
    crop = hip.to_nhwc(d[:, :, 2:-2, 1:-3])
    assert torch.equal(crop.cpu(), t[:, :, 2:-2, 1:-3].permute(0, 2, 3, 1))


def test_decode_matches_oracle(gold_host):
    for inst in (255, 271):
        p = orc.Hyper(inst)
        S = p.score_size
        window = np.outer(np.hanning(S), np.hanning(S))
        for case in range(4):
            c = 'i%d/decode%d' % (inst, case)
            cls, cm, bbox = gold_host[c + '/cls'], gold_host[c + '/cls_mem'], gold_host[c + '/bbox']
            tsz, sz = gold_host[c + '/tsz'], float(gold_host[c + '/scale_z'])
            out = hip.decode(torch.from_numpy(cls).to(DEV).reshape(-1), torch.from_numpy(cm).to(DEV).reshape(-1),
                             torch.from_numpy(bbox).to(DEV).reshape(4, -1), torch.from_numpy(window).to(DEV).reshape(-1),
                             S, inst, 8, p.ratio, p.penalty_k, p.window_influence, tsz[0] * sz, tsz[1] * sz).cpu().numpy()
            pos, szo, score, box, rc = orc.decode(p, cls[0, 0], cm[0, 0], bbox[0], gold_host[c + '/tpos'],
                                                  tsz * sz, window, sz)
            assert int(out[0]) == rc[0] * S + rc[1]
            assert abs(out[1] - score) < 1e-6
            np.testing.assert_allclose(out[3:7], box, rtol=1e-9)


BF16_CASES = [(2, 64, 17, 19, 64, 1, 1, (0, 0), (1, 1)), (1, 128, 15, 15, 128, 3, 2, (0, 0), (1, 1)),
              (2, 256, 13, 13, 256, 3, 1, (2, 2), (2, 2)), (1, 512, 9, 9, 1024, 3, 1, (1, 1), (1, 1)),
              (4, 1024, 31, 31, 256, 1, 1, (0, 0), (1, 1))]


@pytest.mark.parametrize('case', BF16_CASES[:3])
def test_conv_fp16_and_f32_out(case):
    N, Cin, H, W, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 1)
    x = torch.randn(N, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).half()
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.float(), w.float(), b, stride, pad, dil))
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    y = hip.conv2d_bf16(xd, wd, b.to(DEV), KH=k, KW=k, stride=stride, pad=pad, dil=dil, act=hip.ACT_RELU)
    assert y.dtype == torch.float16
    assert rel_err(y.float().permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 1e-3
    y32 = hip.conv2d_bf16(xd, wd, b.to(DEV), KH=k, KW=k, stride=stride, pad=pad, dil=dil, act=hip.ACT_RELU, out_f32=True)
    assert y32.dtype == torch.float32
    assert rel_err(y32.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('case', BF16_CASES)
@pytest.mark.parametrize("tile", __import__('conftest').tile_params(range(0, 38), lp=True))
def test_conv_bf16(case, tile):
    """bf16 MFMA conv vs an fp32 conv on the SAME bf16-rounded operands: the only differences
    are fp32 summation order and the final bf16 rounding (2^-8 relative)."""
    N, Cin, H, W, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).bfloat16()
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.float(), w.float(), b, stride, pad, dil)
    res = torch.randn_like(ref).bfloat16()
    ref_r = F.relu(ref + res.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    y = hip.conv2d_bf16(xd, wd, b.to(DEV), KH=k, KW=k, stride=stride, pad=pad, dil=dil, tile=tile)
    assert rel_err(y.float().permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 6e-3
    rd = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = hip.conv2d_bf16(xd, wd, b.to(DEV), KH=k, KW=k, stride=stride, pad=pad, dil=dil, res=rd, act=hip.ACT_RELU, tile=tile)
    assert rel_err(y.float().permute(0, 3, 1, 2).cpu().numpy(), ref_r.numpy()) < 6e-3


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('cm,co,cn,M', [(256, 1024, 256, 961 * 3), (128, 512, 128, 3969), (64, 256, 64, 15876 + 37),
                                        (64, 256, 128, 4000), (128, 512, 256, 64 * 300), (256, 1024, 256, 50)])
def test_pw_pair_fused_equals_the_two_convolutions(cm, co, cn, M, dtype):
    """csrc/pw_pair.hip: conv3 + residual + ReLU fused with the next 1x1 conv.  Reference = the two unfused
    low-precision launches on the same operands (themselves checked against torch in test_conv_bf16): the wide map
    must be bit-identical (same fp32 accumulation order, one rounding), the narrow map within the storage type's
    rounding of it; and both within tolerance of an fp32 torch evaluation of the rounded operands."""
    g = torch.Generator().manual_seed(cm + co + cn + M)
    lp = lambda a: a.to(dtype).to(DEV)
    t2 = lp(torch.randn(M, cm, generator=g).relu())
    res = lp(torch.randn(M, co, generator=g).relu())
    w3 = lp(torch.randn(co, cm, generator=g) / np.sqrt(cm))
    w1 = lp(torch.randn(cn, co, generator=g) / np.sqrt(co))
    b3, b1 = torch.randn(co, generator=g).to(DEV), torch.randn(cn, generator=g).to(DEV)
    for act2 in (hip.ACT_RELU, hip.ACT_NONE):
        y, t = hip.pw_pair(t2, w3, b3, res, w1, b1, act2=act2)
        y_ref = hip.conv2d_bf16(t2.view(1, 1, M, cm), w3, b3, KH=1, KW=1, res=res.view(1, 1, M, co), act=hip.ACT_RELU).view(M, co)
        t_ref = hip.conv2d_bf16(y_ref.view(1, 1, M, co), w1, b1, KH=1, KW=1, act=act2).view(M, cn)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), float((y.float() - y_ref.float()).abs().max())
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        d = (t.float() - t_ref.float()).abs()
        assert float((d / t_ref.float().abs().clamp_min(1.0)).max()) <= 2 * ulp
        yt = torch.relu(t2.float() @ w3.float().t() + b3 + res.float())
        assert rel_err(y.float().cpu().numpy(), yt.cpu().numpy()) < 4 * ulp
        tt = y.float() @ w1.float().t() + b1
        if act2 == hip.ACT_RELU:
            tt = tt.relu()
        assert rel_err(t.float().cpu().numpy(), tt.cpu().numpy()) < 4 * ulp
    assert not hip.pw_pair_supported(256, 1024, 128) and hip.pw_pair_supported(128, 512, 256)


@pytest.mark.parametrize('pos,win', [((240.3, 180.7), 255), ((10.2, 8.9), 255), ((470.0, 350.0), 301), ((200.5, 100.5), 188),
                                     ((5.0, 355.0), 127), ((240.0, 180.0), 271), ((100.0, 100.0), 612),
                                     ((240.0, 180.0), 510), ((30.0, 300.0), 510), ((240.0, 180.0), 509), ((240.0, 180.0), 511)])
def test_device_crop_matches_host_crop(pos, win):
    """crop + mean pad + fixed-point bilinear + HWC->CHW on the device == hostutils, bit for bit."""
    from usot_amd import hostutils, synth
    im, _ = synth.frame(5, t=3)
    avg = np.mean(im, axis=(0, 1))
    size = 255
    want, _ = hostutils.get_subwindow_tracking(im, np.array(pos), size, win, avg)
    (cx0, _, cy0, _), (top, _, left, _) = hostutils.crop_geometry(im.shape, pos, win)
    out = torch.empty(3, size, size, device=DEV)
    hip.crop_resize(torch.from_numpy(np.ascontiguousarray(im)).to(DEV), out, int(cx0) - left, int(cy0) - top, win, avg.astype(np.uint8))
    assert torch.equal(out.cpu(), want)


def test_conv_batch_heterogeneous():
    """Three convolutions of different geometry (the dilated encoder triple) and a split-K one
    in a single launch == the same convolutions launched one by one."""
    import ctypes as C
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 256, 31, 31, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    geos = [((1, 1), 29, 29), ((2, 1), 27, 29), ((1, 2), 29, 27)]
    descs, outs, refs, keep = [], [], [], []
    for i, (dil, oh, ow) in enumerate(geos):
        w = torch.randn(128, 256, 3, 3, generator=g) / 48
        b = torch.randn(128, generator=g)
        refs.append(F.relu(F.conv2d(x, w, b, 1, 0, dil)))
        wd, bd = pack_w(w).to(DEV), b.to(DEV)
        y = torch.empty(1, oh, ow, 128, device=DEV)
        ks = 3 if i == 1 else 1
        ws = torch.zeros(ks * oh * ow * 128 + ((oh * ow + 15) // 16) * 4, device=DEV) if ks > 1 else None   # slabs + tile tickets
        descs.append(hip.conv_desc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), N=1, H=31, W=31, Cin=256,
                                   OH=oh, OW=ow, Cout=128, KH=3, KW=3, dil=dil, act=hip.ACT_RELU, tile=31 if i == 0 else 0,
                                   ksplit=ks, ws=ws.data_ptr() if ws is not None else None))
        outs.append(y)
        keep += [wd, bd, ws]
    arr = (hip.ConvDesc * 3)(*descs)
    hip.check(hip.lib().usot_conv2d_batch_f32(hip.stream(), arr, 3), 'batch')
    for y, ref in zip(outs, refs):
        assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5


def test_rows_copy_gather_scatter():
    bank = torch.randn(16, 64, device=DEV)
    idx = torch.tensor([3, 0, 15, 7], dtype=torch.int32, device=DEV)
    out = torch.zeros(4, 64, device=DEV)
    hip.check(hip.lib().usot_rows_copy_f32(hip.stream(), hip.ptr(bank), hip.ptr(idx), hip.ptr(out), 4, 64, 0), 'gather')
    assert torch.equal(out, bank[idx.long()])
    dst = torch.zeros(16, 64, device=DEV)
    hip.check(hip.lib().usot_rows_copy_f32(hip.stream(), hip.ptr(out), hip.ptr(idx), hip.ptr(dst), 4, 64, 1), 'scatter')
    want = torch.zeros(16, 64, device=DEV)
    want[idx.long()] = out
    assert torch.equal(dst, want)          # rows not named stay untouched


def test_plan_run_capture_and_lanes():
    """A plan replays natively, as a captured hipGraph, and with forked lanes — same results."""
    import ctypes as C
    L = hip.lib()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 64, 20, 20, generator=g)
    w1, w2 = torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, 64, 1, 1, generator=g) / 8
    ref = F.relu(F.conv2d(x, w1, None, 1, 1)) + F.conv2d(x, w2)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w1d, w2d = pack_w(w1).to(DEV), pack_w(w2).to(DEV)
    for lanes in (False, True):
        y1 = torch.zeros(1, 20, 20, 64, device=DEV)
        y2 = torch.zeros(1, 20, 20, 64, device=DEV)
        plan = C.c_void_p(L.usot_plan_create())
        d1 = hip.conv_desc(xd.data_ptr(), w1d.data_ptr(), None, y1.data_ptr(), N=1, H=20, W=20, Cin=64, OH=20, OW=20,
                           Cout=64, KH=3, KW=3, pad=(1, 1), act=hip.ACT_RELU)
        # second conv adds the first one's output as residual -> depends on it across lanes
        d2 = hip.conv_desc(xd.data_ptr(), w2d.data_ptr(), None, y2.data_ptr(), N=1, H=20, W=20, Cin=64, OH=20, OW=20,
                           Cout=64, KH=1, KW=1, res=y1.data_ptr())
        if lanes:
            hip.check(L.usot_plan_fork(plan, 1))
        hip.check(L.usot_plan_add_conv(plan, C.byref(d1)))
        if lanes:
            hip.check(L.usot_plan_join(plan, 1))
        hip.check(L.usot_plan_add_conv(plan, C.byref(d2)))
        assert L.usot_plan_size(plan) == (4 if lanes else 2)
        hip.check(L.usot_plan_run(plan, hip.stream()))
        torch.cuda.synchronize()
        assert rel_err(y2.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            y1.zero_(); y2.zero_()
            s.synchronize()
            hip.check(L.usot_plan_capture(plan, hip.stream()))
            hip.check(L.usot_plan_run(plan, hip.stream()))
            s.synchronize()
        assert rel_err(y2.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5
        assert L.usot_plan_add_conv(plan, C.byref(d1)) == -4          # USOT_ESTATE: captured plans are frozen
        ms = (C.c_float * L.usot_plan_size(plan))()
        L.usot_plan_destroy(plan)


def test_decode_dev_writes_roi_and_tag(gold_host):
    import ctypes as C
    p = orc.Hyper(255)
    S = p.score_size
    c = 'i255/decode1'
    cls, cm, bbox = gold_host[c + '/cls'], gold_host[c + '/cls_mem'], gold_host[c + '/bbox']
    tsz, sz = gold_host[c + '/tsz'], float(gold_host[c + '/scale_z'])
    window = torch.from_numpy(np.outer(np.hanning(S), np.hanning(S))).reshape(-1).to(DEV)
    ctl = torch.zeros(8, dtype=torch.float64, device=DEV)
    ctl[0], ctl[1], ctl[6] = tsz[0] * sz, tsz[1] * sz, 42.0
    out = torch.zeros(9, dtype=torch.float64, device=DEV)
    roi = torch.zeros(5, device=DEV)
    dcls, dcm, dbox = torch.from_numpy(cls).to(DEV), torch.from_numpy(cm).to(DEV), torch.from_numpy(bbox).to(DEV)
    hip.check(hip.lib().usot_decode_dev_f32(hip.stream(), hip.ptr(dcls), hip.ptr(dcm),
                                            hip.ptr(dbox), hip.ptr(window), hip.ptr(out), S, 255, 8,
                                            C.c_float(p.ratio), C.c_double(p.penalty_k), C.c_double(p.window_influence),
                                            hip.ptr(ctl), hip.ptr(roi)), 'decode_dev')
    o = out.cpu().numpy()
    assert o[8] == 42.0
    np.testing.assert_array_equal(roi.cpu().numpy()[1:], gold_host[c + '/out_poolbox'][0])
    assert roi.cpu().numpy()[0] == 0.0


@pytest.mark.parametrize('dtype,wdtype', [(torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16), (torch.bfloat16, torch.float16)],
                         ids=['bf16', 'fp16', 'bf16_out_fp16_math'])
@pytest.mark.parametrize('size,n', [(255, 2), (127, 1), (271, 1), (63, 3), (255, 9)])
def test_stem_pool_lp(size, n, dtype, wdtype):
    """Fused MFMA stem + max-pool vs torch on the SAME rounded operands (filters rounded to the
    fragment type; crop rounded to fp16, or to hi + lo bf16 pairs; fp32 accumulate): differences are
    accumulation order + one final rounding, i.e. at most 1 ulp of the storage type.  Third mode: fp16 fragments and crop,
    bf16 output (the stem of the bf16 backbone).  (255, 9): enough tiles for strips of 2 per workgroup."""
    from usot_amd.engine import pack_stem_lp
    g = torch.Generator().manual_seed(size + n)
    x = (torch.rand(n, 3, size, size, generator=g) * 2 - 1) * 3
    mu = (0.25, -0.5, 1.0)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    b = torch.randn(64, generator=g) * 0.1
    packed = w.permute(1, 2, 3, 0).reshape(147, 64).contiguous()
    got = hip.stem_pool_lp(x.to(DEV), pack_stem_lp(packed, wdtype).to(DEV), b.to(DEV), dtype, mu).float().cpu()
    xc = x - torch.tensor(mu).view(1, 3, 1, 1)
    xr, wr = xc.to(wdtype).double(), w.to(wdtype).double()
    if wdtype == torch.bfloat16:
        xr = xr + (xc - xc.to(wdtype).float()).to(wdtype).double()
    ref = torch.relu(F.conv2d(xr, wr, b.double(), stride=2)).float().to(dtype).float()
    ref = F.max_pool2d(ref, 3, 2, 1).permute(0, 2, 3, 1)
    assert got.shape == ref.shape
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    err = (got - ref).abs() / ref.abs().clamp_min(1.0)
    assert float(err.max()) <= 1.01 * ulp, float(err.max())
    assert float((err > 0).float().mean()) < 0.02          # the rare 1-ulp rounding flips only


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('tile', [0, 4, 13, 21])
def test_conv_lp_groups_and_activations(dtype, tile):
    """Low-precision conv with three grouped problems (the towers), split activations (conf | value of
    Conf_Fusion, connect.py:123-131) and fp32 output, vs torch on the same rounded operands."""
    g = torch.Generator().manual_seed(77)
    G, N, H, Cin, Cout = 3, 2, 13, 64, 64
    x = torch.randn(G, N, Cin, H, H, generator=g).to(dtype)
    w = (torch.randn(G * Cout, Cin, 3, 3, generator=g) / 24).to(dtype)
    b = torch.randn(G * Cout, generator=g) * 0.2
    xd = x.permute(0, 1, 3, 4, 2).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).reshape(G * Cout, -1).contiguous().to(DEV)
    y = hip.conv2d_bf16(xd, wd, b.to(DEV), KH=3, KW=3, pad=(1, 1), act=hip.ACT_RELU, tile=tile, groups=G, out_f32=True)
    for gi in range(G):
        ref = F.relu(F.conv2d(x[gi].float(), w[gi * Cout:(gi + 1) * Cout].float(), b[gi * Cout:(gi + 1) * Cout], 1, 1))
        assert rel_err(y[gi].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5
    # conf (exp(clamp(relu))) on the first half of the channels, value (relu) on the second
    y = hip.conv2d_bf16(xd[0], wd[:Cout], b[:Cout].to(DEV), KH=3, KW=3, pad=(1, 1), act=hip.ACT_CONF, act2=hip.ACT_RELU,
                        act_split=Cout // 2, tile=tile, out_f32=True)
    pre = F.conv2d(x[0].float(), w[:Cout].float(), b[:Cout], 1, 1)
    ref = torch.cat([torch.exp(torch.clamp(F.relu(pre[:, :Cout // 2]), -6, 4)), F.relu(pre[:, Cout // 2:])], 1)
    assert rel_err(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 2e-5
    y = hip.conv2d_bf16(xd[0], wd[:Cout], (b[:Cout] * 0.1).to(DEV), KH=3, KW=3, pad=(1, 1), act=hip.ACT_EXP, tile=tile)
    ref = torch.exp(F.conv2d(x[0].float(), w[:Cout].float(), b[:Cout] * 0.1, 1, 1))
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    assert float(((y.float().permute(0, 3, 1, 2).cpu() - ref).abs() / ref.abs()).max()) < 1.5 * ulp


@pytest.mark.parametrize('n,hw', [(1, 25), (2, 27), (1, 7), (16, 25), (33, 27)])
def test_thin_conv3x3_prediction_heads(n, hw):
    """usot_thin_conv3x3_f32: bbox_pred (4 ch, exp) + cls/cls_mem preds (two 1-ch groups) in one launch.  From 1024 output rows
    (n >= 14 at 25 x 25) the launcher switches to the wavefront-per-ROW form (filters in registers, sliding window): same taps,
    same order, same fused multiply-adds — bit-identical to the wavefront-per-pixel form (forced with tile = 70)."""
    import ctypes as C
    g = torch.Generator().manual_seed(n * 100 + hw)
    x = torch.randn(3, n, 256, hw, hw, generator=g)                       # three tower outputs
    wb = torch.randn(4, 256, 3, 3, generator=g) / 48
    bb = torch.randn(4, generator=g) * 0.1
    wc = torch.randn(2, 256, 3, 3, generator=g) / 48
    bc = torch.randn(2, generator=g)
    xd = x.permute(0, 1, 3, 4, 2).contiguous().to(DEV)
    wbd, wcd, bbd, bcd = pack_w(wb).to(DEV), pack_w(wc).to(DEV), bb.to(DEV), bc.to(DEV)
    yb = torch.empty(n, 4, hw, hw, device=DEV)
    yc = torch.empty(2, n, 1, hw, hw, device=DEV)
    gs = n * hw * hw * 256
    descs = [hip.conv_desc(xd[0].data_ptr(), wbd.data_ptr(), bbd.data_ptr(), yb.data_ptr(), N=n, H=hw, W=hw, Cin=256, OH=hw, OW=hw,
                           Cout=4, KH=3, KW=3, pad=(1, 1), act=hip.ACT_EXP, y_nchw=1),
             hip.conv_desc(xd[1].data_ptr(), wcd.data_ptr(), bcd.data_ptr(), yc.data_ptr(), N=n, H=hw, W=hw, Cin=256, OH=hw, OW=hw,
                           Cout=1, KH=3, KW=3, pad=(1, 1), y_nchw=1, groups=2, x_gs=gs, w_gs=2304, b_gs=1, y_gs=n * hw * hw)]
    arr = (hip.ConvDesc * 2)(*descs)
    hip.check(hip.lib().usot_thin_conv3x3_f32(hip.stream(), arr, 2), 'thin')
    ref_b = torch.exp(F.conv2d(x[0], wb, bb, 1, 1))
    assert rel_err(yb.cpu().numpy(), ref_b.numpy()) < 1e-5
    for gi in range(2):
        ref = F.conv2d(x[1 + gi], wc[gi:gi + 1], bc[gi:gi + 1], 1, 1)
        assert rel_err(yc[gi].cpu().numpy(), ref.numpy()) < 1e-5
    yb2, yc2 = torch.zeros_like(yb), torch.zeros_like(yc)
    descs2 = [hip.conv_desc(xd[0].data_ptr(), wbd.data_ptr(), bbd.data_ptr(), yb2.data_ptr(), N=n, H=hw, W=hw, Cin=256, OH=hw, OW=hw,
                            Cout=4, KH=3, KW=3, pad=(1, 1), act=hip.ACT_EXP, y_nchw=1, tile=70),
              hip.conv_desc(xd[1].data_ptr(), wcd.data_ptr(), bcd.data_ptr(), yc2.data_ptr(), N=n, H=hw, W=hw, Cin=256, OH=hw, OW=hw,
                            Cout=1, KH=3, KW=3, pad=(1, 1), y_nchw=1, groups=2, x_gs=gs, w_gs=2304, b_gs=1, y_gs=n * hw * hw, tile=70)]
    hip.check(hip.lib().usot_thin_conv3x3_f32(hip.stream(), (hip.ConvDesc * 2)(*descs2), 2), 'thin per pixel')
    assert torch.equal(yb, yb2) and torch.equal(yc, yc2)
    bad = hip.conv_desc(xd[0].data_ptr(), wbd.data_ptr(), bbd.data_ptr(), yb.data_ptr(), N=n, H=hw, W=hw, Cin=256, OH=hw, OW=hw,
                        Cout=4, KH=3, KW=3, pad=(0, 0), y_nchw=1)
    assert hip.lib().usot_thin_conv3x3_f32(hip.stream(), C.byref(bad), 1) != 0


def test_rows_copy_multi_gather_scatter_and_stash():
    """Four banks of different row length move with the same device row indices in one launch; the gather
    also stashes idx[n_rows .. n_rows+2] (the frame's append row and the crop address words: usot_hip.h) in device memory,
    so a gather with a stash needs n_rows + 3 index entries."""
    import ctypes as C
    lens = [64, 128, 32, 256]
    banks = [torch.randn(12, n, device=DEV) for n in lens]
    idx = torch.tensor([5, 0, 11, 7, 9, 1234, -77], dtype=torch.int32, device=DEV)   # 4 rows + the three stashed entries
    outs = [torch.zeros(4, n, device=DEV) for n in lens]
    stash = torch.zeros(4, dtype=torch.int32, device=DEV)
    pp = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
    rl = (C.c_int32 * 4)(*lens)
    hip.check(hip.lib().usot_rows_copy_multi_f32(hip.stream(), 4, pp(banks), hip.ptr(idx), pp(outs), 4, rl, 0, hip.ptr(stash)), 'gather')
    for b, o in zip(banks, outs):
        assert torch.equal(o, b[idx[:4].long()])
    assert stash.tolist() == [9, 1234, -77, 0]
    dsts = [torch.zeros(12, n, device=DEV) for n in lens]
    hip.check(hip.lib().usot_rows_copy_multi_f32(hip.stream(), 3, pp(outs), hip.ptr(idx), pp(dsts), 4, rl, 1, None), 'scatter')
    for k, (o, d) in enumerate(zip(outs, dsts)):
        want = torch.zeros_like(d)
        if k < 3:
            want[idx[:4].long()] = o
        assert torch.equal(d, want)
    # a stash pointer is meaningless for a scatter
    assert hip.lib().usot_rows_copy_multi_f32(hip.stream(), 1, pp(outs), hip.ptr(idx), pp(dsts), 4, rl, 1, hip.ptr(stash)) != 0


@pytest.mark.parametrize('pinned', [False, True], ids=['idx_dev', 'idx_pinned'])
def test_rows_append_gather(pinned):
    """usot_rows_append_gather_f32 (the session's default frame since round 6): ONE launch appends a fresh row to four banks
    and gathers n_pick rows of banks 1-3 - a picked row that IS the appended one must come from the fresh row (the bank row is
    being written by other workgroups of the same launch), every other bank row stays untouched, and the index block may be
    pinned host memory (the session's control block).  Against plain indexing; the reference keeps the queue in Python lists
    (usot_tracker.py:222-264)."""
    import ctypes as C
    lens = [7 * 7 * 256, 5 * 5 * 256, 3 * 5 * 256, 64]
    g = torch.Generator().manual_seed(11)
    for picks, slot in (([0, 1, 4, 9, 9, 3, 9], 9), ([0, 1, 2, 2, 2, 2, 2], 11), ([5], 5), (list(range(12)) + [3] * 20, 7)):
        banks = [torch.randn(12, n, generator=g).to(DEV) for n in lens]
        before = [b.clone() for b in banks]
        fresh = [torch.randn(1, n, generator=g).to(DEV) for n in lens]
        nq = len(picks)
        idx_h = torch.tensor(picks + [-5, 123, 456, slot], dtype=torch.int32)          # slot_pos = nq + 3, as in the control block
        idx = idx_h.pin_memory() if pinned else idx_h.to(DEV)
        picked = [torch.full((nq, n), -1.0, device=DEV) for n in lens[1:]]
        p4 = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
        p3 = (C.c_void_p * 3)(*[t.data_ptr() for t in picked])
        hip.check(hip.lib().usot_rows_append_gather_f32(hip.stream(), p4(fresh), p4(banks), p3, (C.c_int32 * 4)(*lens),
                                                        C.c_void_p(idx.data_ptr()), nq, nq + 3), 'rows_append_gather')
        torch.cuda.synchronize()
        for k in range(4):
            want = before[k].clone()
            want[slot] = fresh[k][0]
            assert torch.equal(banks[k], want), (k, picks, slot)
            if k:
                assert torch.equal(picked[k - 1], want[torch.tensor(picks).long()]), (k, picks, slot)
    # argument checks: more than 32 picked rows, a row length that is not a multiple of four floats, a missing pointer
    bad = (C.c_void_p * 3)(picked[0].data_ptr(), None, picked[2].data_ptr())
    L = hip.lib()
    assert L.usot_rows_append_gather_f32(hip.stream(), p4(fresh), p4(banks), p3, (C.c_int32 * 4)(*lens), C.c_void_p(idx.data_ptr()), 33, 36) != 0
    assert L.usot_rows_append_gather_f32(hip.stream(), p4(fresh), p4(banks), p3, (C.c_int32 * 4)(lens[0], 6, lens[2], lens[3]), C.c_void_p(idx.data_ptr()), 4, 7) != 0
    assert L.usot_rows_append_gather_f32(hip.stream(), p4(fresh), p4(banks), bad, (C.c_int32 * 4)(*lens), C.c_void_p(idx.data_ptr()), 4, 7) != 0


@pytest.mark.parametrize('size,n', [(255, 2), (127, 1), (271, 1), (63, 3)])
def test_stem_pool_f32(size, n):
    """Fused fp32 MFMA stem + max-pool vs conv2d + relu + max_pool2d (fp32: accumulation order only)."""
    from usot_amd.engine import pack_stem_f32
    g = torch.Generator().manual_seed(size * 3 + n)
    x = torch.rand(n, 3, size, size, generator=g) * 255
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.02
    w = w - w.mean((1, 2, 3), keepdim=True)
    b = torch.randn(64, generator=g) * 0.1
    packed = w.permute(1, 2, 3, 0).reshape(147, 64).contiguous()
    got = hip.stem_pool(x.to(DEV), pack_stem_f32(packed).to(DEV), b.to(DEV)).cpu()
    ref = F.max_pool2d(F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2)).float(), 3, 2, 1).permute(0, 2, 3, 1)
    assert got.shape == ref.shape
    assert rel_err(got.numpy(), ref.numpy()) < 2e-5
    two = hip.maxpool3x3s2(hip.stem_conv(x.to(DEV), packed.to(DEV), b.to(DEV))).cpu()
    assert rel_err(got.numpy(), two.numpy()) < 2e-5
    mu = (104.0, 117.0, 123.0)
    bmu = (b.double() + (w.double() * torch.tensor(mu, dtype=torch.float64).view(1, 3, 1, 1)).sum((1, 2, 3))).float()
    gmu = hip.stem_pool(x.to(DEV), pack_stem_f32(packed).to(DEV), bmu.to(DEV), mu).cpu()
    assert rel_err(gmu.numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('K,N,M,res,act', [(256, 1024, 256 * 3, True, 1), (256, 1024, 1000, True, 1), (256, 1024, 77, False, 0),
                                           (128, 512, 256 * 2 + 5, True, 1), (64, 256, 512 * 2 + 130, False, 0), (64, 256, 512, True, 1),
                                           # >= 192 panels of 256 pixels: the two-pixel-blocks-per-wave form (below: 128-pixel panels)
                                           (256, 1024, 192 * 256 + 19, True, 1), (128, 512, 193 * 256, True, 1)])
def test_pw_panel_lp_expansion_conv(K, N, M, res, act, dtype):
    """Pixel-stationary 1x1 expansion conv (csrc/pw_panel.hip) vs the SAME rounded operands in float64: full and ragged panels
    (M not a multiple of the panel, fewer pixels than one wave's block), with / without residual and ReLU; and bit-equal to the
    tiled low-precision conv kernel's result (same products, fp32 accumulation in the same k order, one rounding)."""
    import ctypes as C
    g = torch.Generator().manual_seed(K + N + M)
    x = (torch.randn(M, K, generator=g)).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g).to(dtype) if res else None
    ref = x.double() @ w.double().t() + b.double()
    if res:
        ref = ref + r.double()
    if act:
        ref = ref.relu()
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    rd = r.to(DEV) if res else None
    y = torch.full((M + 3, N), 7.0, dtype=dtype, device=DEV)         # three guard rows behind the output
    assert hip.lib().usot_pw_panel_supported(K, N) == 1 and hip.lib().usot_pw_panel_supported(K, N + 64) == 0
    hip.check(hip.lib().usot_pw_panel_lp(hip.stream(), hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), hip.ptr(rd) if res else None, hip.ptr(y),
                                         M, K, N, act, 1 if dtype == torch.float16 else 0), 'usot_pw_panel_lp')
    got = y[:M].float().cpu().double()
    assert torch.all(y[M:] == 7.0)                                   # nothing written past the last pixel
    tol = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11           # one rounding of the output
    assert float(((got - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= tol * 1.01
    # the tiled kernel on the same operands (NHWC with H = M, W = 1)
    y2 = torch.empty(M, N, dtype=dtype, device=DEV)
    d = hip.conv_desc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y2.data_ptr(), N=1, H=M, W=1, Cin=K, OH=M, OW=1, Cout=N, KH=1, KW=1,
                      res=rd.data_ptr() if res else None, act=act, tile=11)
    hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d), 1 if dtype == torch.float16 else 0, 0), 'usot_conv2d_lp')
    assert torch.equal(y2, y[:M])
    assert hip.lib().usot_pw_panel_lp(hip.stream(), hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), None, hip.ptr(y), M, K, N + 64, act, 0) != 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('cm,co,cn,M,act2', [(128, 512, 128, 256 * 4 + 17, 1), (128, 512, 128, 61, 0), (128, 512, 128, 256 * 2 + 17, 1),
                                             (128, 512, 256, 128 + 9, 1), (64, 256, 64, 256 * 3, 1), (64, 256, 128, 256 + 200, 1)])
def test_pw_panel_pair_equals_the_two_convolutions(cm, co, cn, M, act2, dtype):
    """Pair form of the panel kernel (Y's accumulators feed the second GEMM from registers) vs the two tiled launches on the
    same operands: Y bit-identical (same products, same k order, one rounding); T within 2 ulp of the storage type (the
    second GEMM visits k in (group, quad, half) order: same products, another fp32 summation order)."""
    import ctypes as C
    g = torch.Generator().manual_seed(cm + co + cn + M)
    t2 = torch.randn(M, cm, generator=g).to(dtype)
    w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype)
    w1 = (torch.randn(cn, co, generator=g) / co ** 0.5).to(dtype)
    b3, b1 = torch.randn(co, generator=g) * 0.1, torch.randn(cn, generator=g) * 0.1
    res = torch.randn(M, co, generator=g).to(dtype)
    dt = 1 if dtype == torch.float16 else 0
    dev = lambda a: a.to(DEV)
    t2d, w3d, w1d, b3d, b1d, resd = map(dev, (t2, w3, w1, b3, b1, res))
    y = torch.full((M + 2, co), 3.0, dtype=dtype, device=DEV)
    t = torch.full((M + 2, cn), 3.0, dtype=dtype, device=DEV)
    d = hip.pw_pair_desc(t2d.data_ptr(), w3d.data_ptr(), b3d.data_ptr(), resd.data_ptr(), y.data_ptr(), w1d.data_ptr(), b1d.data_ptr(),
                         t.data_ptr(), M, cm, co, cn, act2)
    assert hip.lib().usot_pw_panel_pair_supported(cm, co, cn) == 1
    hip.check(hip.lib().usot_pw_panel_pair_lp(hip.stream(), C.byref(d), dt), 'usot_pw_panel_pair_lp')
    assert torch.all(y[M:] == 3.0) and torch.all(t[M:] == 3.0)
    y2 = torch.empty(M, co, dtype=dtype, device=DEV)
    d1 = hip.conv_desc(t2d.data_ptr(), w3d.data_ptr(), b3d.data_ptr(), y2.data_ptr(), N=1, H=M, W=1, Cin=cm, OH=M, OW=1, Cout=co, KH=1, KW=1,
                       res=resd.data_ptr(), act=1, tile=11)
    hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d1), dt, 0), 'conv3')
    tt = torch.empty(M, cn, dtype=dtype, device=DEV)
    d2 = hip.conv_desc(y2.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), tt.data_ptr(), N=1, H=M, W=1, Cin=co, OH=M, OW=1, Cout=cn, KH=1, KW=1,
                       act=act2, tile=11)
    hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d2), dt, 0), 'conv1')
    assert torch.equal(y[:M], y2)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    a, b = t[:M].float(), tt.float()
    assert float(((a - b).abs() / b.abs().clamp_min(0.25)).max()) <= 2 * ulp
    ref = (y2.double() @ w1d.double().t() + b1d.double())
    if act2:
        ref = ref.relu()
    assert float(((a.double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= ulp
    assert hip.lib().usot_pw_panel_pair_supported(cm, co, cn + 32) == 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('rs', [False, True], ids=['pertap', 'rowshared'])
@pytest.mark.parametrize('form', [1, 2], ids=['panel256', 'panel128'])
@pytest.mark.parametrize('n,h,pad,dil', [(2, 31, 2, 2), (1, 13, 1, 1), (1, 16, 1, 1), (3, 9, 2, 2), (2, 11, 3, 3)])
def test_conv_pw_fused_equals_the_two_launches(n, h, pad, dil, dtype, form, rs):
    """Layer3's conv2 -> conv3 in one launch (csrc/conv_pw_lp.hip: the 256-pixel T2 panel stays in LDS) vs the two launches it
    replaces on the same operands - the 256 x 256 implicit-GEMM tile (32), then the pixel-stationary panel kernel: BIT-identical
    (same products, same k order, T2 rounded once); and against float64 on the rounded operands.  Geometries: layer3's dilated
    conv2 (b8..b12: pad 2 / dil 2) and b7's (pad 1 / dil 1); pixel counts with full panels + a ragged one, a single ragged panel
    (169 < 256: waves without any pixel), exactly one panel (256).  form: the 256-pixel panel on 16 wavefronts / the 128-pixel
    panel on 8 (what the launcher picks below 192 panels of 256: batch 32), forced through c2->tile.  rs: phase 1 on the row-shared
    k-loop (one staged tile per (kh, chunk) serves the three kw taps; k order (kh, chunk, kw)): same products, another fp32
    summation order - within one rounding of T2 of the per-tap loop, which (tile & 4) is bit-identical to the unfused launches."""
    import ctypes as C
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + pad)
    cin = cm = 256
    co = 1024
    M = n * h * h
    t1 = torch.randn(n, h, h, cin, generator=g).relu().to(dtype)
    w2 = (torch.randn(cm, 9 * cin, generator=g) / (9 * cin) ** 0.5).to(dtype)       # [Cout][kh][kw][Cin]
    w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype)
    b2, b3 = torch.randn(cm, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1
    res = torch.randn(M, co, generator=g).to(dtype)
    dt = 1 if dtype == torch.float16 else 0
    t1d, w2d, w3d, b2d, b3d, resd = (a.to(DEV) for a in (t1, w2, w3, b2, b3, res))
    assert hip.lib().usot_conv_pw_supported(cin, cm, co) == 1 and hip.lib().usot_conv_pw_supported(cin, 64, co) == 0
    y = torch.full((M + 2, co), 5.0, dtype=dtype, device=DEV)
    d = hip.conv_desc(t1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), None, N=n, H=h, W=h, Cin=cin, OH=h, OW=h, Cout=cm, KH=3, KW=3,
                      pad=(pad, pad), dil=(dil, dil), act=1, tile=form | (0 if rs else 4))
    assert hip.lib().usot_conv_pw_pixels(M) == 128 and hip.lib().usot_conv_pw_pixels(192 * 256) == 256
    cus = torch.cuda.get_device_properties(0).multi_processor_count          # a mostly empty second round of 256-pixel panels: 128
    assert hip.lib().usot_conv_pw_pixels((cus + 17) * 256) == 128 and hip.lib().usot_conv_pw_pixels(2 * cus * 256) == 256
    hip.check(hip.lib().usot_conv_pw_lp(hip.stream(), C.byref(d), hip.ptr(w3d), hip.ptr(b3d), hip.ptr(resd), hip.ptr(y), dt), 'usot_conv_pw_lp')
    torch.cuda.synchronize()
    assert torch.all(y[M:] == 5.0)                                   # nothing written past the last pixel
    # the two launches
    t2 = torch.empty(n, h, h, cm, dtype=dtype, device=DEV)
    d2 = hip.conv_desc(t1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), t2.data_ptr(), N=n, H=h, W=h, Cin=cin, OH=h, OW=h, Cout=cm, KH=3, KW=3,
                       pad=(pad, pad), dil=(dil, dil), act=1, tile=32)
    hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d2), dt, 0), 'conv2')
    y2 = torch.empty(M, co, dtype=dtype, device=DEV)
    hip.check(hip.lib().usot_pw_panel_lp(hip.stream(), hip.ptr(t2), hip.ptr(w3d), hip.ptr(b3d), hip.ptr(resd), hip.ptr(y2), M, cm, co, 1, dt),
              'conv3')
    torch.cuda.synchronize()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    if rs:      # another summation order: a T2 value on a rounding boundary may round the other way; its effect on Y is bounded
        a, b = y[:M].float(), y2.float()
        assert float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) <= 4 * ulp
        assert float((a != b).float().mean()) < 0.2
    else:
        assert torch.equal(y[:M], y2)
    # float64 on the rounded operands, T2 rounded to the storage type as both paths do
    x64 = t1.double().permute(0, 3, 1, 2)
    w64 = w2.double().view(cm, 3, 3, cin).permute(0, 3, 1, 2)
    t2r = torch.nn.functional.conv2d(x64, w64, b2.double(), padding=pad, dilation=dil).relu().permute(0, 2, 3, 1).reshape(M, cm)
    t2r = t2r.to(dtype).double()
    ref = (t2r @ w3.double().t() + b3.double() + res.double()).relu()
    got = y[:M].float().cpu().double()
    # a T2 value that sits on a rounding boundary may round the other way in fp32 accumulation: allow its effect on Y
    assert float(((got - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 4 * ulp
    # argument checks: unsupported width, conv2 without ReLU, misaligned output
    bad = hip.conv_desc(t1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), None, N=n, H=h, W=h, Cin=cin, OH=h, OW=h, Cout=cm, KH=3, KW=3,
                        pad=(pad, pad), dil=(dil, dil), act=0)
    assert hip.lib().usot_conv_pw_lp(hip.stream(), C.byref(bad), hip.ptr(w3d), hip.ptr(b3d), hip.ptr(resd), hip.ptr(y), dt) != 0
    assert hip.lib().usot_conv_pw_lp(hip.stream(), C.byref(d), hip.ptr(w3d), hip.ptr(b3d), None, hip.ptr(y), dt) != 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('form', [1, 2], ids=['panel256', 'panel128'])
@pytest.mark.parametrize('rs', [False, True], ids=['pertap', 'rowshared'])
@pytest.mark.parametrize('n,h,stride,pad,cn', [(2, 31, 1, 1, 128), (1, 33, 2, 0, 128), (1, 16, 1, 1, 128), (2, 31, 1, 1, 0), (1, 12, 1, 1, 0)])
def test_conv_pw_layer2_forms_equal_the_unfused_launches(n, h, stride, pad, cn, dtype, form, rs):
    """Layer2's widths on the fused kernel (csrc/conv_pw_lp.hip): conv2 3x3 128 -> 128 (stride 1 / pad 1, and b3's stride 2 / pad 0)
    -> conv3 128 -> 512 + residual + ReLU, with (cn = 128: pair form) or without the next block's conv1 - bit-identical to the
    tiled conv2 followed by the pixel-stationary panel kernel (pair form: usot_pw_panel_pair_lp), Y and T alike."""
    import ctypes as C
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + stride + cn)
    cm, co = 128, 512
    oh = (h + 2 * pad - 3) // stride + 1
    M = n * oh * oh
    t1 = torch.randn(n, h, h, cm, generator=g).relu().to(dtype)
    w2 = (torch.randn(cm, 9 * cm, generator=g) / (9 * cm) ** 0.5).to(dtype)
    w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype)
    w1 = (torch.randn(max(cn, 64), co, generator=g) / co ** 0.5).to(dtype)
    b2, b3, b1 = torch.randn(cm, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1, torch.randn(max(cn, 64), generator=g) * 0.1
    res = torch.randn(M, co, generator=g).to(dtype)
    dt = 1 if dtype == torch.float16 else 0
    t1d, w2d, w3d, w1d, b2d, b3d, b1d, resd = (a.to(DEV) for a in (t1, w2, w3, w1, b2, b3, b1, res))
    geo = dict(N=n, H=h, W=h, Cin=cm, OH=oh, OW=oh, Cout=cm, KH=3, KW=3, stride=stride, pad=(pad, pad), act=1)
    y = torch.full((M + 2, co), 5.0, dtype=dtype, device=DEV)
    t = torch.full((M + 2, max(cn, 64)), 5.0, dtype=dtype, device=DEV)
    d = hip.conv_desc(t1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), None, tile=form | (0 if rs else 4), **geo)
    exact = not rs or stride != 1                      # the row-shared loop (stride 1 only) sums in another order
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    close = lambda a, b: float(((a.float() - b.float()).abs() / b.float().abs().clamp_min(1.0)).max()) <= 4 * ulp
    t2 = torch.empty(n, oh, oh, cm, dtype=dtype, device=DEV)
    d2 = hip.conv_desc(t1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), t2.data_ptr(), tile=37, **geo)
    hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d2), dt, 0), 'conv2')
    y2 = torch.empty(M, co, dtype=dtype, device=DEV)
    if cn:
        assert hip.lib().usot_conv_pw_pair_supported(cm, co, cn) == 1 and hip.lib().usot_conv_pw_pair_supported(cm, co, 64) == 0
        pd = hip.pw_pair_desc(None, w3d.data_ptr(), b3d.data_ptr(), resd.data_ptr(), y.data_ptr(), w1d.data_ptr(), b1d.data_ptr(),
                              t.data_ptr(), M, cm, co, cn, 1)
        hip.check(hip.lib().usot_conv_pw_pair_lp(hip.stream(), C.byref(d), C.byref(pd), dt), 'usot_conv_pw_pair_lp')
        tt = torch.empty(M, cn, dtype=dtype, device=DEV)
        pd2 = hip.pw_pair_desc(t2.data_ptr(), w3d.data_ptr(), b3d.data_ptr(), resd.data_ptr(), y2.data_ptr(), w1d.data_ptr(),
                               b1d.data_ptr(), tt.data_ptr(), M, cm, co, cn, 1)
        hip.check(hip.lib().usot_pw_panel_pair_lp(hip.stream(), C.byref(pd2), dt), 'usot_pw_panel_pair_lp')
        torch.cuda.synchronize()
        assert (torch.equal(t[:M], tt) if exact else close(t[:M], tt)) and torch.all(t[M:] == 5.0)
        pd.M = M + 1                                                # the pair's pixel count must be conv2's
        assert hip.lib().usot_conv_pw_pair_lp(hip.stream(), C.byref(d), C.byref(pd), dt) != 0
    else:
        assert hip.lib().usot_conv_pw_supported(cm, cm, co) == 1
        hip.check(hip.lib().usot_conv_pw_lp(hip.stream(), C.byref(d), hip.ptr(w3d), hip.ptr(b3d), hip.ptr(resd), hip.ptr(y), dt), 'usot_conv_pw_lp')
        hip.check(hip.lib().usot_pw_panel_lp(hip.stream(), hip.ptr(t2), hip.ptr(w3d), hip.ptr(b3d), hip.ptr(resd), hip.ptr(y2), M, cm, co, 1, dt),
                  'conv3')
        torch.cuda.synchronize()
    assert (torch.equal(y[:M], y2) if exact else close(y[:M], y2)) and torch.all(y[M:] == 5.0)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('rs', [False, True], ids=['pertap', 'rowshared'])
@pytest.mark.parametrize('form', [1, 2], ids=['panel256', 'panel128'])
@pytest.mark.parametrize('n,h,pad,act2,cm', [(2, 31, 2, 1, 256), (1, 13, 1, 0, 256), (1, 16, 1, 1, 256), (2, 31, 1, 1, 128), (1, 12, 1, 1, 128)])
def test_conv_pw_layer3_with_the_next_conv1_as_fifth_phase(n, h, pad, act2, cm, dtype, form, rs):
    """Layer3's fused block with the next block's conv1 (or the neck: act2 = 0) as phase 5 of the launch (csrc/conv_pw_lp.hip: the
    workgroup reads its own Y panel back and runs the 1024 -> 256 convolution on the freed LDS): Y as the four-phase kernel's, T
    bit-identical to the tiled convolution on that Y (same k order) - also on the ragged last panel and with waves without pixels.
    cm = 128: layer2's last block, whose next conv1 is layer3's first (512 -> 256)."""
    import ctypes as C
    g = torch.Generator().manual_seed(n * 100 + h + act2 + cm)
    co, cn = 4 * cm, 256
    M = n * h * h
    t1 = torch.randn(n, h, h, cm, generator=g).relu().to(dtype)
    w2 = (torch.randn(cm, 9 * cm, generator=g) / (9 * cm) ** 0.5).to(dtype)
    w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype)
    w1 = (torch.randn(cn, co, generator=g) / co ** 0.5).to(dtype)
    b2, b3, b1 = torch.randn(cm, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1, torch.randn(cn, generator=g) * 0.1
    res = torch.randn(M, co, generator=g).to(dtype)
    dt = 1 if dtype == torch.float16 else 0
    t1d, w2d, w3d, w1d, b2d, b3d, b1d, resd = (a.to(DEV) for a in (t1, w2, w3, w1, b2, b3, b1, res))
    geo = dict(N=n, H=h, W=h, Cin=cm, OH=h, OW=h, Cout=cm, KH=3, KW=3, pad=(pad, pad), dil=(pad, pad), act=1)
    tile = form | (0 if rs else 4)
    y = torch.full((M + 2, co), 5.0, dtype=dtype, device=DEV)
    t = torch.full((M + 2, cn), 5.0, dtype=dtype, device=DEV)
    d = hip.conv_desc(t1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), None, tile=tile, **geo)
    assert hip.lib().usot_conv_pw_pair_supported(cm, co, cn) == 1
    pd = hip.pw_pair_desc(None, w3d.data_ptr(), b3d.data_ptr(), resd.data_ptr(), y.data_ptr(), w1d.data_ptr(), b1d.data_ptr(),
                          t.data_ptr(), M, cm, co, cn, act2)
    hip.check(hip.lib().usot_conv_pw_pair_lp(hip.stream(), C.byref(d), C.byref(pd), dt), 'usot_conv_pw_pair_lp')
    # the four-phase kernel with the same k-loop, then the tiled conv1 on ITS Y
    y2 = torch.empty(M, co, dtype=dtype, device=DEV)
    hip.check(hip.lib().usot_conv_pw_lp(hip.stream(), C.byref(d), hip.ptr(w3d), hip.ptr(b3d), hip.ptr(resd), hip.ptr(y2), dt), 'usot_conv_pw_lp')
    tt = torch.empty(M, cn, dtype=dtype, device=DEV)
    d1 = hip.conv_desc(y2.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), tt.data_ptr(), N=1, H=M, W=1, Cin=co, OH=M, OW=1, Cout=cn, KH=1, KW=1,
                       act=act2, tile=32)
    hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d1), dt, 0), 'conv1')
    torch.cuda.synchronize()
    assert torch.equal(y[:M], y2) and torch.all(y[M:] == 5.0)
    assert torch.equal(t[:M], tt) and torch.all(t[M:] == 5.0)
    ref = y2.double() @ w1d.double().t() + b1d.double()
    if act2:
        ref = ref.relu()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float(((t[:M].double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= ulp * 1.01


@pytest.mark.parametrize('rs', [False, True], ids=['pertap', 'rowshared'])
def test_conv_pw_at_the_timed_size_is_deterministic_and_equal_to_the_unfused_chain(rs):
    """The fused layer3 block (conv2 -> conv3 -> next conv1, csrc/conv_pw_lp.hip) at the TIMED size - batch 64 at 31 x 31: 241
    workgroups, one per CU, every hand-counted s_waitcnt of the kernel under full load - run eight times on the same inputs: every run
    bit-identical to the first (a slab or stage read before its DMA landed would show as run-to-run differences), Y and T of the
    per-tap form bit-identical to the three unfused launches."""
    import ctypes as C
    g = torch.Generator().manual_seed(64)
    n, h, cm, co, cn = 64, 31, 256, 1024, 256
    M = n * h * h
    dtype, dt = torch.bfloat16, 0
    t1 = torch.randn(n, h, h, cm, generator=g).relu().to(dtype).to(DEV)
    w2 = (torch.randn(cm, 9 * cm, generator=g) / (9 * cm) ** 0.5).to(dtype).to(DEV)
    w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype).to(DEV)
    w1 = (torch.randn(cn, co, generator=g) / co ** 0.5).to(dtype).to(DEV)
    b2, b3, b1 = (torch.randn(c, generator=g).mul(0.1).to(DEV) for c in (cm, co, cn))
    res = torch.randn(M, co, generator=g).to(dtype).to(DEV)
    geo = dict(N=n, H=h, W=h, Cin=cm, OH=h, OW=h, Cout=cm, KH=3, KW=3, pad=(2, 2), dil=(2, 2), act=1)
    d = hip.conv_desc(t1.data_ptr(), w2.data_ptr(), b2.data_ptr(), None, tile=0 if rs else 4, **geo)
    assert hip.lib().usot_conv_pw_pixels(M) == 256
    runs = []
    for _ in range(8):
        y = torch.zeros(M, co, dtype=dtype, device=DEV)
        t = torch.zeros(M, cn, dtype=dtype, device=DEV)
        pd = hip.pw_pair_desc(None, w3.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1.data_ptr(), b1.data_ptr(), t.data_ptr(),
                              M, cm, co, cn, 1)
        hip.check(hip.lib().usot_conv_pw_pair_lp(hip.stream(), C.byref(d), C.byref(pd), dt), 'usot_conv_pw_pair_lp')
        torch.cuda.synchronize()
        runs.append((y, t))
    for y, t in runs[1:]:
        assert torch.equal(y, runs[0][0]) and torch.equal(t, runs[0][1])
    if not rs:
        t2 = torch.empty(n, h, h, cm, dtype=dtype, device=DEV)
        d2 = hip.conv_desc(t1.data_ptr(), w2.data_ptr(), b2.data_ptr(), t2.data_ptr(), tile=32, **geo)
        hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d2), dt, 0), 'conv2')
        y2 = torch.empty(M, co, dtype=dtype, device=DEV)
        hip.check(hip.lib().usot_pw_panel_lp(hip.stream(), hip.ptr(t2), hip.ptr(w3), hip.ptr(b3), hip.ptr(res), hip.ptr(y2), M, cm, co, 1, dt), 'conv3')
        tt = torch.empty(M, cn, dtype=dtype, device=DEV)
        d1 = hip.conv_desc(y2.data_ptr(), w1.data_ptr(), b1.data_ptr(), tt.data_ptr(), N=1, H=M, W=1, Cin=co, OH=M, OW=1, Cout=cn, KH=1, KW=1,
                           act=1, tile=32)
        hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d1), dt, 0), 'conv1')
        torch.cuda.synchronize()
        assert torch.equal(runs[0][0], y2) and torch.equal(runs[0][1], tt)


@pytest.mark.experiments
@pytest.mark.parametrize('n,h,dil,act2', [(2, 31, 2, 1), (1, 12, 1, 0), (3, 33, 2, 1), (16, 31, 2, 1)])
def test_conv_pw_overlapped_form_matches_float64_and_its_own_conv1(n, h, dil, act2):
    """csrc/conv_pw_ov.hip (EXPERIMENT: measured slower, not routed): a layer3 block as matrix-pipe and HBM workgroups side by side,
    paired through write-through stores and flags.  Y against float64 on the rounded operands (T2 rounded to the storage type as the
    kernel does: <= 4 ulp, a T2 value on a rounding boundary may round the other way), T bit-identical to the tiled conv1 on this
    launch's own Y, nothing written past the last pixel, the hand-off flags back at zero (graph-replay safe), error word clear; twice."""
    import ctypes as C
    L = hip.lib()
    dtype, dt = torch.bfloat16, 0
    cm, co, cn = 256, 1024, 256
    M = n * h * h
    g = torch.Generator().manual_seed(n * 100 + h)
    t1 = torch.randn(n, h, h, cm, generator=g).relu().to(dtype)
    w2 = (torch.randn(cm, 9 * cm, generator=g) / (9 * cm) ** 0.5).to(dtype)
    w3 = (torch.randn(co, cm, generator=g) / cm ** 0.5).to(dtype)
    w1 = (torch.randn(cn, co, generator=g) / co ** 0.5).to(dtype)
    b2, b3, b1 = (torch.randn(c, generator=g) * 0.1 for c in (cm, co, cn))
    res = torch.randn(M, co, generator=g).to(dtype)
    t1d, w2d, w3d, w1d, b2d, b3d, b1d, resd = (a.to(DEV) for a in (t1, w2, w3, w1, b2, b3, b1, res))
    assert L.usot_conv_pw_ov_supported(cm, co, cn) == 1
    ws = torch.zeros(int(L.usot_conv_pw_ov_ws_bytes(M)) // 4, dtype=torch.int32, device=DEV)
    d2 = hip.conv_desc(t1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), None, N=n, H=h, W=h, Cin=cm, OH=h, OW=h, Cout=cm, KH=3, KW=3,
                       pad=(dil, dil), dil=(dil, dil), act=1)
    x64 = t1.double().permute(0, 3, 1, 2)
    w64 = w2.double().view(cm, 3, 3, cm).permute(0, 3, 1, 2)
    t2r = torch.nn.functional.conv2d(x64, w64, b2.double(), padding=dil, dilation=dil).relu().permute(0, 2, 3, 1).reshape(M, cm).to(dtype).double()
    ref = (t2r @ w3.double().t() + b3.double() + res.double()).relu()
    ulp = 2.0 ** -8
    for _ in range(2):
        y = torch.full((M + 2, co), 5.0, dtype=dtype, device=DEV)
        t = torch.full((M + 2, cn), 5.0, dtype=dtype, device=DEV)
        pd = hip.pw_pair_desc(None, w3d.data_ptr(), b3d.data_ptr(), resd.data_ptr(), y.data_ptr(), w1d.data_ptr(), b1d.data_ptr(),
                              t.data_ptr(), M, cm, co, cn, act2)
        hip.check(L.usot_conv_pw_ov_lp(hip.stream(), C.byref(d2), C.byref(pd), dt, hip.ptr(ws)), 'usot_conv_pw_ov_lp')
        torch.cuda.synchronize()
        assert torch.all(y[M:] == 5.0) and torch.all(t[M:] == 5.0)
        assert int(ws[:2 * ((M + 127) // 128) + 1].abs().sum()) == 0
        assert float(((y[:M].float().cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 4 * ulp
        tt = torch.empty(M, cn, dtype=dtype, device=DEV)
        yc = y[:M].contiguous()
        d1 = hip.conv_desc(yc.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), tt.data_ptr(), N=1, H=M, W=1, Cin=co, OH=M, OW=1, Cout=cn, KH=1, KW=1,
                           act=act2, tile=32)
        hip.check(L.usot_conv2d_lp(hip.stream(), C.byref(d1), dt, 0), 'conv1')
        torch.cuda.synchronize()
        assert torch.equal(t[:M], tt)


def test_backbone_bf16_conv_pw_option_is_bit_identical():
    """Engine options 'conv_pw_lp' / 'conv_pw_pair_lp' (layer3's conv2 -> conv3 and layer2's conv2 -> conv3 -> next conv1 fused per
    pixel panel): the batched bf16 backbone's output is bit-identical to the unfused lowering's with the per-tap k-loop, and within
    a few bf16 roundings of it with the row-shared k-loop ('conv_pw_rs', another fp32 summation order)."""
    from usot_amd import synth
    from usot_amd.model import USOT
    outs = []
    for on in (False, True, 'rs'):
        m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.to(DEV)
        m.engine.opt['conv_pw_lp'] = (256,) if on else ()
        m.engine.opt['conv_pw_pair_lp'] = m.engine.opt['conv_pw_p5_lp'] = bool(on)
        m.engine.opt['conv_pw_rs'] = on == 'rs'
        x = torch.from_numpy(synth.crop(3, 52, 255)).to(DEV)
        xf = m.engine.features_bf16(x)
        torch.cuda.synchronize()
        kinds = [k for k, *_ in next(v for kk, v in m.engine._feat.items() if kk[0] == 'bf16')['plan'].profile(1)]
        assert (30 in kinds) == bool(on)
        outs.append(xf.clone())
        del m
    assert torch.equal(outs[0], outs[1])
    # the row-shared k-loop: other summation order in nine 3x3 convs - the neck output stays within a few bf16 roundings
    a, b = outs[2].float(), outs[0].float()
    assert float((a - b).abs().max()) <= 0.05 * float(b.abs().max()) and float((a - b).abs().mean()) <= 4e-3 * float(b.abs().mean() + 1e-6) + 1e-3


def test_launcher_state_is_indexed_by_the_current_device():
    """csrc/common.h: the launchers' cached state (zero pages, raised LDS limits, CU counts) lives in tables indexed by the current HIP
    device, so one process may drive several GPUs; the index is hipGetDevice's."""
    L = hip.lib()
    assert L.usot_device_slot() == torch.cuda.current_device() and L.usot_device_guard() == 0


def test_bw_probe_kernels_move_the_right_bytes():
    """The HBM ceiling probes bench.py quotes GroupDW against (csrc/bw_probe.hip): copy copies, the 4:1 mix sums the four
    adjacent 1 KiB rows of each 64-element group, bad arguments are refused."""
    n = 1 << 20
    src = torch.randn(n // 4, device=DEV)
    dst = torch.zeros(n // 4, device=DEV)
    L = hip.lib()
    hip.check(L.usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), n, 1), 'copy')
    assert torch.equal(dst, src)
    dst.zero_()
    ragged = n - 3 * 4096                                   # not a multiple of the copy's 32 KiB block span
    hip.check(L.usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), ragged, 1), 'copy (ragged)')
    assert torch.equal(dst[:ragged // 4], src[:ragged // 4]) and float(dst[ragged // 4:].abs().max()) == 0.0
    dst.zero_()
    hip.check(L.usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), n, 2), 'mix')
    v = src.view(-1, 4, 64, 4)                              # [group][row of the group][element][4 floats]
    want = (v[:, 0] + v[:, 1]) + (v[:, 2] + v[:, 3])
    got = dst[: n // 16].view(-1, 64, 4)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6) and float(dst[n // 16:].abs().max()) == 0.0
    hip.check(L.usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), n, 0), 'read')
    assert L.usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), n + 16, 1) != 0
    assert L.usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), n, 4) != 0
    # mode 3, GroupDW's traffic pattern: S samples = three 29 x 29 x 256 maps read, one 25 x 25 x 256 map written, nothing beyond
    S = 3
    src3 = torch.ones(3 * S * 841 * 256, device=DEV)
    dst3 = torch.zeros(S * 625 * 256 + 1024, device=DEV)
    hip.check(L.usot_bw_probe(hip.stream(), hip.ptr(src3), hip.ptr(dst3), 3 * S * 841 * 1024, 3), 'groupdw pattern')
    torch.cuda.synchronize()
    assert float(dst3[:S * 625 * 256].min()) > 0.0 and float(dst3[S * 625 * 256:].abs().max()) == 0.0
    assert L.usot_bw_probe(hip.stream(), hip.ptr(src3), hip.ptr(dst3), 4096, 3) != 0          # less than one sample


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,Cin,Cout,stride,pad,dil,act', [
    (2, 31, 31, 256, 256, 1, 2, 2, 1),        # layer3.1-5 conv2 (dilation 2)
    (1, 31, 31, 256, 256, 1, 1, 1, 1),        # layer3.0 conv2
    (2, 31, 31, 512, 1024, 1, 1, 1, 0),       # layer3's shortcut conv
    (2, 63, 63, 256, 512, 2, 0, 1, 0),        # layer2's shortcut conv (stride 2, no padding)
    (1, 7, 9, 256, 256, 1, 1, 1, 1),          # one ragged panel, every border case
    (3, 16, 16, 256, 512, 1, 1, 2, 1)])       # pad < dil
def test_conv_kstream_lp(N, H, W, Cin, Cout, stride, pad, dil, act, dtype):
    """Accumulator-stationary implicit GEMM for the K >= 2304 3x3 convolutions (csrc/conv_kstream.hip: a lane is a pixel, B
    fragments straight from global memory, padding taps parked on the centre pixel and zeroed after landing) against the
    tiled low-precision kernel on the same operands (within 1 ulp of the storage type: same products, another fp32 order) and
    against float64 on the rounded operands."""
    import ctypes as Ct
    L = hip.lib()
    assert L.usot_conv_kstream_supported(Cin, Cout, 3, 3) == 1 and L.usot_conv_kstream_supported(64, 256, 3, 3) == 0
    g = torch.Generator().manual_seed(N + H * 7 + Cin + Cout + stride + dil)
    x = torch.randn(N, H, W, Cin, generator=g).to(dtype)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (3 * Cin ** 0.5)).to(dtype)
    b = torch.randn(Cout, generator=g) * 0.1
    OH, OW = (H + 2 * pad - 2 * dil - 1) // stride + 1, (W + 2 * pad - 2 * dil - 1) // stride + 1
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), stride=stride, padding=pad,
                   dilation=dil).permute(0, 2, 3, 1)
    if act:
        ref = ref.relu()
    xd, wd, bd = x.to(DEV), w.reshape(Cout, 9 * Cin).contiguous().to(DEV), b.to(DEV)
    M = N * OH * OW
    y = torch.full((M + 2, Cout), 5.0, dtype=dtype, device=DEV)
    dt = 1 if dtype == torch.float16 else 0
    hip.check(L.usot_conv_kstream_lp(hip.stream(), hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), hip.ptr(y), N, H, W, Cin, Cout, stride, pad, dil,
                                     act, dt), 'conv_kstream')
    assert torch.all(y[M:] == 5.0)
    got = y[:M].reshape(N, OH, OW, Cout).float()
    tol = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    assert float(((got.cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= tol * 1.01
    y2 = torch.empty(N, OH, OW, Cout, dtype=dtype, device=DEV)
    d = hip.conv_desc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y2.data_ptr(), N=N, H=H, W=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout,
                      KH=3, KW=3, stride=stride, pad=(pad, pad), dil=(dil, dil), act=act, tile=13)
    hip.check(L.usot_conv2d_lp(hip.stream(), Ct.byref(d), dt, 0), 'tiled')
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert float(((got - y2.float()).abs() / y2.float().abs().clamp_min(0.25)).max()) <= ulp
    assert L.usot_conv_kstream_lp(hip.stream(), hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), hip.ptr(y), N, H, W, Cin, Cout, stride, 3, 1, act, dt) == -1   # pad > dil


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('M,act,bias', [(961 * 3, 1, True), (256 * 5, 0, True), (61504, 1, True), (300, 0, False),
                                        (70000, 1, True)])      # > 256 panels: the persistent loop's second trip
def test_pw_kstream_lp_reducing_conv(M, act, bias, dtype):
    """Accumulator-stationary 1x1 convolution 1024 -> 256 (csrc/pw_kstream.hip: X fragments straight from global memory three
    k-chunks ahead, W slabs through LDS) against the tiled low-precision conv on the same operands: same products, fp32
    accumulation in a different order, one final rounding — a few ulp of the storage type; ragged last panels; no bias."""
    K, N = 1024, 256
    assert hip.lib().usot_pw_kstream_supported(K, N) == 1 and hip.lib().usot_pw_kstream_supported(256, 1024) == 0
    g = torch.Generator().manual_seed(M + act)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / 32).to(dtype).to(DEV)
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    y = torch.full((M + 8, N), 7.0, dtype=dtype, device=DEV)
    hip.check(hip.lib().usot_pw_kstream_lp(hip.stream(), hip.ptr(x), hip.ptr(w), hip.ptr(b) if bias else None, hip.ptr(y), M, K, N,
                                           act, 1 if dtype == torch.float16 else 0), 'usot_pw_kstream_lp')
    ref = x.float() @ w.float().t() + (b if bias else 0.0)
    if act:
        ref = torch.relu(ref)
    got = y[:M].float()
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    err = ((got - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert err <= 1.5 * ulp, err
    assert (y[M:].float() == 7.0).all()                        # nothing written past the last pixel
    assert hip.lib().usot_pw_kstream_lp(hip.stream(), hip.ptr(x), hip.ptr(w), None, hip.ptr(y), M, 512, N, act, 0) == -1     # USOT_EINVAL



@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,act', [(2, 63, 63, 1), (1, 16, 16, 0), (3, 17, 33, 1), (1, 5, 40, 1)])
def test_conv3x3_halo_lp(N, H, W, act, dtype):
    """Direct 3x3 convolution from an LDS halo tile (csrc/conv3x3_halo.hip) vs the same rounded operands in float64 and
    bit-equal... no: equal to the tiled implicit-GEMM kernel within one rounding of the storage type (the halo kernel visits
    k tap by tap in the same order, 32 channels per MFMA: same products, same fp32 chain) — full, ragged and sub-tile images."""
    import ctypes as C
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
    x = torch.randn(N, H, W, 64, generator=g).to(dtype)
    w = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(dtype)                  # [co][kh][kw][ci]
    b = torch.randn(64, generator=g) * 0.1
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
    if act:
        ref = ref.relu()
    xd, wd, bd = x.to(DEV), w.reshape(64, 576).contiguous().to(DEV), b.to(DEV)
    y = torch.full((N * H * W + 2, 64), 5.0, dtype=dtype, device=DEV)
    dt = 1 if dtype == torch.float16 else 0
    assert hip.lib().usot_conv3x3_halo_supported(64, 64) == 1 and hip.lib().usot_conv3x3_halo_supported(128, 128) == 0
    hip.check(hip.lib().usot_conv3x3_halo_lp(hip.stream(), hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), hip.ptr(y), N, H, W, 64, 64, act, dt), 'halo')
    assert torch.all(y[N * H * W:] == 5.0)
    got = y[:N * H * W].reshape(N, H, W, 64)
    tol = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    assert float(((got.float().cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= tol * 1.01
    y2 = torch.empty(N, H, W, 64, dtype=dtype, device=DEV)
    d = hip.conv_desc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y2.data_ptr(), N=N, H=H, W=W, Cin=64, OH=H, OW=W, Cout=64, KH=3, KW=3,
                      pad=(1, 1), act=act, tile=13)
    hip.check(hip.lib().usot_conv2d_lp(hip.stream(), C.byref(d), dt, 0), 'tiled')
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert float(((got.float() - y2.float()).abs() / y2.float().abs().clamp_min(0.25)).max()) <= ulp
    assert hip.lib().usot_conv3x3_halo_lp(hip.stream(), hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), hip.ptr(y), N, H, W, 128, 128, act, dt) != 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W', [(2, 63, 63), (9, 8, 16), (3, 17, 33), (1, 5, 130), (40, 9, 17)])
def test_bneck_first_lp(N, H, W, dtype):
    """Layer1's first bottleneck + the next conv1 in one launch (csrc/bneck_lp.hip; modules.py:37-58,108-113) against the
    same chain in float64 on the same rounded operands, rounding t1, t2 and y to the storage type where the kernel does
    (the shortcut conv is NOT rounded on its own: it shares conv3's accumulator).  Full, ragged and sub-tile images, more
    tiles than workgroups... (40, 9, 17): 160 tiles; nothing written outside y and t."""
    import ctypes as C
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = rnd(N, H, W, 64).relu().to(dtype)
    w1 = (rnd(64, 64) / 8).to(dtype); b1 = rnd(64) * 0.1
    w2 = (rnd(64, 3, 3, 64) / 24).to(dtype); b2 = rnd(64) * 0.1
    w3 = (rnd(256, 64) / 8).to(dtype); wd = (rnd(256, 64) / 8).to(dtype); b3c = rnd(256) * 0.1
    wn = (rnd(64, 256) / 16).to(dtype); bn = rnd(64) * 0.1
    rq = lambda v: v.to(dtype).double()
    xd_ = x.double()
    t1 = rq((xd_ @ w1.double().t() + b1.double()).relu())
    t2 = F.conv2d(t1.permute(0, 3, 1, 2), w2.double().permute(0, 3, 1, 2), b2.double(), padding=1).permute(0, 2, 3, 1)
    t2 = rq(t2.relu())
    y_ref = (t2 @ w3.double().t() + xd_ @ wd.double().t() + b3c.double()).relu()
    yr = rq(y_ref)
    t_ref = (yr @ wn.double().t() + bn.double()).relu()
    dev = lambda v: v.contiguous().to(DEV)
    xd, w1d, w2d, w3cd, wnd = dev(x), dev(w1), dev(w2.reshape(64, 576)), dev(torch.cat([w3, wd], 1)), dev(wn)
    b1d, b2d, b3d, bnd = dev(b1), dev(b2), dev(b3c), dev(bn)
    M = N * H * W
    y = torch.full((M + 1, 256), 5.0, dtype=dtype, device=DEV)
    t = torch.full((M + 1, 64), 7.0, dtype=dtype, device=DEV)
    dt = 1 if dtype == torch.float16 else 0
    assert hip.lib().usot_bneck_first_supported(64, 64, 256, 64) == 1 and hip.lib().usot_bneck_first_supported(256, 64, 256, 64) == 0
    d = hip.bneck_desc(*[hip.ptr(v) for v in (xd, w1d, b1d, w2d, b2d, w3cd, b3d, wnd, bnd, y, t)], N, H, W)
    hip.check(hip.lib().usot_bneck_first_lp(hip.stream(), C.byref(d), dt), 'bneck_first')
    torch.cuda.synchronize()
    assert torch.all(y[M:] == 5.0) and torch.all(t[M:] == 7.0)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    got_y = y[:M].reshape(N, H, W, 256).float().cpu().double()
    got_t = t[:M].reshape(N, H, W, 64).float().cpu().double()
    # y: one rounding of the storage type on top of the float64 chain, plus what a flipped rounding of t1 / t2 moves (a few ulp
    # of inputs of size ~1 through 64-term sums with |w| ~ 1/8)
    ey = float(((got_y - y_ref).abs() / y_ref.abs().clamp_min(1.0)).max())
    et = float(((got_t - t_ref).abs() / t_ref.abs().clamp_min(1.0)).max())
    assert ey <= 3 * ulp, ey
    assert et <= 4 * ulp, et
    assert hip.lib().usot_bneck_first_lp(hip.stream(), C.byref(hip.bneck_desc(*[hip.ptr(v) for v in (xd, w1d, b1d, w2d, b2d, w3cd, b3d, wnd, bnd, y, t)], 1, 4, 16)), dt) != 0   # < 8 tiles


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('cn', [64, 128])
@pytest.mark.parametrize('N,H,W', [(2, 63, 63), (9, 8, 16), (3, 17, 33), (1, 5, 130), (40, 9, 17)])
def test_bneck_tail_lp(N, H, W, cn, dtype):
    """The rest of a layer1 bottleneck (conv2 + conv3 + identity residual) + the next block's conv1 in one launch
    (csrc/bneck_lp.hip: bneck_tail_kernel; modules.py:43-58, :40-42) against the same chain in float64 on the same rounded
    operands, rounding t2 and y to the storage type where the kernel does.  Full, ragged, sub-tile images; both next-conv widths."""
    import ctypes as C
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W + cn)
    rnd = lambda *s: torch.randn(*s, generator=g)
    t1 = rnd(N, H, W, 64).relu().to(dtype)
    res = rnd(N, H, W, 256).relu().to(dtype)
    w2 = (rnd(64, 3, 3, 64) / 24).to(dtype); b2 = rnd(64) * 0.1
    w3 = (rnd(256, 64) / 8).to(dtype); b3 = rnd(256) * 0.1
    wn = (rnd(cn, 256) / 16).to(dtype); bn = rnd(cn) * 0.1
    rq = lambda v: v.to(dtype).double()
    t2 = F.conv2d(t1.double().permute(0, 3, 1, 2), w2.double().permute(0, 3, 1, 2), b2.double(), padding=1).permute(0, 2, 3, 1)
    t2 = rq(t2.relu())
    y_ref = (t2 @ w3.double().t() + b3.double() + res.double()).relu()
    t_ref = (rq(y_ref) @ wn.double().t() + bn.double()).relu()
    dev = lambda v: v.contiguous().to(DEV)
    t1d, resd, w2d, w3d, wnd, b2d, b3d, bnd = dev(t1), dev(res), dev(w2.reshape(64, 576)), dev(w3), dev(wn), dev(b2), dev(b3), dev(bn)
    M = N * H * W
    y = torch.full((M + 1, 256), 5.0, dtype=dtype, device=DEV)
    t = torch.full((M + 1, cn), 7.0, dtype=dtype, device=DEV)
    dt = 1 if dtype == torch.float16 else 0
    assert hip.lib().usot_bneck_tail_supported(64, 256, cn) == 1 and hip.lib().usot_bneck_tail_supported(64, 256, 256) == 0
    d = hip.bneck_desc(hip.ptr(t1d), hip.ptr(resd), None, hip.ptr(w2d), hip.ptr(b2d), hip.ptr(w3d), hip.ptr(b3d), hip.ptr(wnd), hip.ptr(bnd),
                       hip.ptr(y), hip.ptr(t), N, H, W)
    hip.check(hip.lib().usot_bneck_tail_lp(hip.stream(), C.byref(d), cn, dt), 'bneck_tail')
    torch.cuda.synchronize()
    assert torch.all(y[M:] == 5.0) and torch.all(t[M:] == 7.0)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    got_y = y[:M].reshape(N, H, W, 256).float().cpu().double()
    got_t = t[:M].reshape(N, H, W, cn).float().cpu().double()
    ey = float(((got_y - y_ref).abs() / y_ref.abs().clamp_min(1.0)).max())
    et = float(((got_t - t_ref).abs() / t_ref.abs().clamp_min(1.0)).max())
    assert ey <= 3 * ulp, ey
    assert et <= 4 * ulp, et
    assert hip.lib().usot_bneck_tail_lp(hip.stream(), C.byref(d), 96, dt) != 0


def test_bneck_counted_waits_equal_full_waits():
    """bneck_first_kernel decides that the next halo tile has landed with `s_waitcnt vmcnt(<stores issued since>)`,
    bneck_tail_kernel relies on its residual loads for the same - correct only while the compiler emits exactly the counted
    vector-memory operations.  The -DUSOT_BNECK_VMCNT0 build waits for everything: both builds, same seeded ragged images
    (more tiles than workgroups, repeated launches), must produce the same BITS (scripts/bneck_bits.py)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'build_variant.py'), 'bneck_vmcnt0', 'bneck_lp.hip', '-DUSOT_BNECK_VMCNT0'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors='replace')[-2000:]
    safe = r.stdout.decode().strip().splitlines()[-1]
    digests = []
    for lib in (None, safe):
        env = dict(os.environ)
        env.pop('USOT_HIP_LIB', None)
        if lib:
            env['USOT_HIP_LIB'] = lib
        q = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'bneck_bits.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert q.returncode == 0, q.stderr.decode(errors='replace')[-2000:]
        digests.append([ln for ln in q.stdout.decode().splitlines() if ln.startswith('bneck_bits')][0])
    assert digests[0] == digests[1], digests


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
@pytest.mark.parametrize('ow', [25, 27])
def test_groupdw_and_conf_reduce_low_precision_outputs(ow, dtype):
    """usot_groupdw_multi_lp / usot_conf_fusion_reduce_lp (the batched mixed-precision heads, BASELINE configs[4]): the same fp32
    arithmetic with the output maps stored as fp16 | bf16: equal to the fp32 launch followed by a conversion except where the fp32
    value is an exact rounding TIE of the storage type (hipcc folds the last fused multiply-add and the conversion into one
    v_fma_mixlo: a single rounding of the exact sum, where fp32-then-fp16 rounds twice; measured 6e-5 of the elements, one ulp)."""

    def same_up_to_ties(got, ref32):
        want = ref32.to(got.dtype)
        ne = got != want
        ulp = 2.0 ** (-10 if got.dtype == torch.float16 else -7)
        err = ((got.float() - ref32).abs() / ref32.abs().clamp_min(1e-3)).max()
        return int(ne.sum()) <= 1e-3 * got.numel() and float(err) <= 0.51 * ulp * 1.01
    import ctypes as C
    g = torch.Generator().manual_seed(ow)
    geo = ((5, 5), (3, 5), (5, 3))
    S, rep = 70, 7
    xs = [torch.randn(S // rep, ow + hk - 1, ow + wk - 1, 256, generator=g).to(DEV) for hk, wk in geo]
    zs = [torch.randn(S, hk, wk, 256, generator=g).to(DEV) for hk, wk in geo]
    w = np.array([0.2, 0.3, 0.5], np.float32)
    ref = hip.groupdw(xs, zs, w, x_rep=rep, cols=6)
    out = torch.full((S * ow * ow * 256 + 8,), 3.0, dtype=dtype, device=DEV)
    d = hip.groupdw_desc([t.data_ptr() for t in xs], [t.data_ptr() for t in zs], out.data_ptr(), w, S=S, x_rep=rep, OH=ow, OW=ow, Cc=256,
                         x_cs=[256] * 3, x_co=[0] * 3, z_cs=[256] * 3, z_co=[0] * 3)
    dt = 1 if dtype == torch.float16 else 2
    hip.check(hip.lib().usot_groupdw_multi_lp(hip.stream(), C.byref(d), 1, dt), 'groupdw_multi_lp')
    torch.cuda.synchronize()
    assert same_up_to_ties(out[:-8].reshape(ref.shape), ref) and torch.all(out[-8:] == 3.0)
    assert hip.lib().usot_groupdw_multi_lp(hip.stream(), C.byref(d), 1, 0) != 0
    B, M, P = 3, 7, ow * ow
    cv = torch.rand(B * M, ow, ow, 512, generator=g).to(DEV) + 0.1
    r32 = hip.conf_fusion_reduce(cv, B, M)
    o = torch.full((B * P * 256 + 8,), 3.0, dtype=dtype, device=DEV)
    hip.check(hip.lib().usot_conf_fusion_reduce_lp(hip.stream(), hip.ptr(cv), 0, hip.ptr(o), B, M, P, 256, dt), 'conf_reduce_lp')
    torch.cuda.synchronize()
    assert same_up_to_ties(o[:-8].reshape(r32.shape), r32) and torch.all(o[-8:] == 3.0)
    # ... and with the confidence | value map itself in the storage type: the fp32 reduction of the ROUNDED map
    cvl = cv.to(dtype)
    r32l = hip.conf_fusion_reduce(cvl.float(), B, M)
    o.fill_(3.0)
    hip.check(hip.lib().usot_conf_fusion_reduce_lp(hip.stream(), hip.ptr(cvl), dt, hip.ptr(o), B, M, P, 256, dt), 'conf_reduce_lp in')
    torch.cuda.synchronize()
    assert same_up_to_ties(o[:-8].reshape(r32l.shape), r32l) and torch.all(o[-8:] == 3.0)
