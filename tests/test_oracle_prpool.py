"""CPU: analytic known answers for the PrRoIPool restatement (oracle/prroi_pool_ref.c).
The reference ships no CPU implementation and no test for this op (parity unpinned); these
KATs follow from the definition — the exact integral of the bilinear surface over each bin
divided by the bin area (prroi_pooling_gpu_impl.cu:149-212)."""
import numpy as np
import torch

import usot_oracle as orc


def test_constant_map_gives_constant():
    f = torch.full((1, 3, 9, 11), 2.5)
    out = orc.prroi_pool(f, torch.tensor([[0, 1.3, 2.2, 7.9, 6.1]]), 7, 7, 1.0)
    np.testing.assert_allclose(out.numpy(), 2.5, rtol=1e-6)


def test_affine_map_gives_value_at_bin_centre():
    h, w = 12, 14
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    f = torch.from_numpy((0.7 * xx - 0.3 * yy + 1.5)[None, None])
    x1, y1, x2, y2 = 2.25, 1.5, 10.75, 9.0
    out = orc.prroi_pool(f, torch.tensor([[0, x1, y1, x2, y2]]), 7, 7, 1.0).numpy()[0, 0]
    bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
    for ph in range(7):
        for pw in range(7):
            cx, cy = x1 + (pw + 0.5) * bw, y1 + (ph + 0.5) * bh
            assert abs(out[ph, pw] - (0.7 * cx - 0.3 * cy + 1.5)) < 1e-4


def test_unit_bins_on_integer_grid_average_four_corners():
    g = torch.Generator().manual_seed(1)
    f = torch.randn(1, 2, 10, 10, generator=g)
    out = orc.prroi_pool(f, torch.tensor([[0, 1.0, 2.0, 8.0, 9.0]]), 7, 7, 1.0).numpy()
    fn = f.numpy()
    for ph in range(7):
        for pw in range(7):
            y, x = 2 + ph, 1 + pw
            want = 0.25 * (fn[0, :, y, x] + fn[0, :, y + 1, x] + fn[0, :, y, x + 1] + fn[0, :, y + 1, x + 1])
            np.testing.assert_allclose(out[0, :, ph, pw], want, rtol=1e-5, atol=1e-6)


def test_outside_and_degenerate_rois_are_zero_and_batch_index_selects():
    f = torch.stack([torch.ones(2, 6, 6), 3 * torch.ones(2, 6, 6)])
    rois = torch.tensor([[0, 20.0, 20.0, 30.0, 30.0], [1, 2.0, 2.0, 2.0, 5.0], [1, 1.0, 1.0, 4.0, 4.0],
                         [0, 5.0, 5.0, 3.0, 3.0]])
    out = orc.prroi_pool(f, rois, 7, 7, 1.0).numpy()
    assert np.all(out[0] == 0) and np.all(out[1] == 0) and np.all(out[3] == 0)
    np.testing.assert_allclose(out[2], 3.0, rtol=1e-6)


def test_partial_overlap_reads_zero_outside():
    f = torch.ones(1, 1, 4, 4)
    # a roi hanging half outside on the left: bins beyond x < 0 integrate the zero extension
    out = orc.prroi_pool(f, torch.tensor([[0, -3.0, 0.0, 3.0, 3.0]]), 1, 6, 1.0).numpy()[0, 0, 0]
    assert np.allclose(out[:2], 0.0, atol=1e-7) and np.allclose(out[4:], 1.0, atol=1e-6)
    assert 0.0 < out[2] < 1.0        # x in [-1, 0]: the bilinear ramp from 0 to 1
    assert abs(out[2] - 0.5) < 1e-6
