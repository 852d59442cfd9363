"""CPU: the PrRoIPool gradient oracles.

The float64 oracle (oracle/prroi_exact.py, written from the operator's definition) is pinned by two facts that
do not involve the reference's kernels: the feature gradient is the ADJOINT of the (linear) forward map, and the
RoI gradient is the derivative of the exact forward w.r.t. the box corners (central differences).  The float32 C
restatement of the reference's kernels (oracle/prroi_pool_ref.c, prroi_pooling_gpu_impl.cu:214-380) must then agree
with it to float32 accuracy."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import prroi_exact as ex                                     # noqa: E402
import usot_oracle as orc                                    # noqa: E402
from prroi_cases import random_rois                          # noqa: E402

B, C, H, W = 2, 3, 9, 11


def _setup(seed, n):
    g = np.random.default_rng(seed)
    f = g.standard_normal((B, C, H, W)).astype(np.float32)
    rois = random_rois(seed + 1, n, B, H, W)
    top_diff = g.standard_normal((n, C, 7, 7)).astype(np.float32)
    return f, rois, top_diff


@pytest.mark.parametrize('seed', [0, 1])
def test_exact_feature_gradient_is_the_adjoint_of_the_forward(seed):
    f, rois, g = _setup(seed, 24)
    grad = ex.prroi_pool_exact_backward(f.shape, rois, g, 7, 7, 1.0)
    for k in range(3):                                       # <g, P(u)> == <P^T g, u> for random u
        u = np.random.default_rng(100 + k).standard_normal(f.shape)
        lhs = np.sum(g.astype(np.float64) * ex.prroi_pool_exact(u, rois, 7, 7, 1.0))
        rhs = np.sum(grad * u)
        assert abs(lhs - rhs) <= 1e-10 * max(1.0, abs(lhs))


@pytest.mark.parametrize('scale', [1.0, 0.5])
def test_exact_roi_gradient_matches_central_differences(scale):
    g0 = np.random.default_rng(5)
    f = g0.standard_normal((B, C, H, W))
    # boxes whose float32 corners are exactly representable so that the oracle's float32 rounding of the inputs is a no-op
    rois = np.array([[0, 1.25, 2.5, 7.75, 6.125], [1, -2.5, -1.25, 5.5, 4.75], [0, 3.0625, 0.5, 14.5, 11.25],
                     [1, 4.5, 3.25, 5.0625, 3.875]], np.float64)       # no bin edge on a grid line: out is C1 there, not C2
    rois[:, 1:] /= scale
    top_diff = g0.standard_normal((len(rois), C, 7, 7))
    grad = ex.prroi_pool_exact_coor_backward(f, rois, top_diff, 7, 7, scale)
    assert np.all(grad[:, 0] == 0)
    h = 2.0 ** -12 / scale                                    # keeps the shifted corners float32-exact as well
    for r in range(len(rois)):
        for k in range(1, 5):
            lo, hi = rois.copy(), rois.copy()
            lo[r, k] -= h
            hi[r, k] += h
            fd = np.sum(top_diff * (ex.prroi_pool_exact(f, hi, 7, 7, scale) - ex.prroi_pool_exact(f, lo, 7, 7, scale))) / (2 * h)
            assert abs(fd - grad[r, k]) <= 2e-5 * max(1.0, abs(grad[r, k])), (r, k, fd, grad[r, k])


@pytest.mark.parametrize('seed', [0, 3])
def test_c_restatement_of_the_feature_gradient_agrees_with_the_exact_oracle(seed):
    f, rois, g = _setup(seed, 36)
    got = orc.prroi_pool_backward(f.shape, rois, g, 7, 7, 1.0).numpy()
    ref = ex.prroi_pool_exact_backward(f.shape, rois, g, 7, 7, 1.0)
    # tiny bins divide float32 cancellation noise of the corner weights by a tiny area (same effect as in the forward)
    assert np.max(np.abs(got - ref)) <= 2e-4 * max(1.0, np.max(np.abs(ref)))
    big = rois[(rois[:, 3] - rois[:, 1] > 2) & (rois[:, 4] - rois[:, 2] > 2)]
    got = orc.prroi_pool_backward(f.shape, big, g[:len(big)], 7, 7, 1.0).numpy()
    ref = ex.prroi_pool_exact_backward(f.shape, big, g[:len(big)], 7, 7, 1.0)
    assert np.max(np.abs(got - ref)) <= 2e-6 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize('seed', [0, 3])
def test_c_restatement_of_the_roi_gradient_agrees_with_the_exact_oracle(seed):
    f, rois, g = _setup(seed, 36)
    rois = rois[(rois[:, 3] - rois[:, 1] > 1.5) & (rois[:, 4] - rois[:, 2] > 1.5)]
    g = g[:len(rois)]
    top = orc.prroi_pool(torch.from_numpy(f), torch.from_numpy(rois), 7, 7, 1.0).numpy()
    got = orc.prroi_pool_coor_backward(f, rois, top, g, 7, 7, 1.0).numpy()
    ref = ex.prroi_pool_exact_coor_backward(f, rois, g, 7, 7, 1.0)
    assert np.all(got[:, 0] == 0)
    assert np.max(np.abs(got - ref)) <= 5e-4 * max(1.0, np.max(np.abs(ref))), np.max(np.abs(got - ref))


def test_degenerate_boxes_have_zero_gradients():
    f, _, g = _setup(2, 2)
    rois = np.array([[0, 3.0, 2.0, 3.0, 6.0], [1, 5.0, 4.0, 2.0, 8.0]], np.float32)      # zero / negative width
    assert not orc.prroi_pool_backward(f.shape, rois, g, 7, 7, 1.0).numpy().any()
    assert not orc.prroi_pool_coor_backward(f, rois, np.zeros_like(g), g, 7, 7, 1.0).numpy().any()
    assert not ex.prroi_pool_exact_backward(f.shape, rois, g, 7, 7, 1.0).any()
    assert not ex.prroi_pool_exact_coor_backward(f, rois, g, 7, 7, 1.0).any()
