"""reference lib/models/prroi_pool/functional.py:41-84 (forward only; the backward kernels
serve training, which is out of scope)."""
import torch

from usot_amd import hip

__all__ = ['prroi_pool2d']


def prroi_pool2d(features, rois, pooled_height, pooled_width, spatial_scale):
    if 'FloatTensor' not in features.type() or 'FloatTensor' not in rois.type():
        raise AssertionError('Precise RoI Pooling only takes float input, got {} for features and {} for rois.'
                             .format(features.type(), rois.type()))
    if not features.is_cuda:
        raise NotImplementedError('Precise RoI Pooling only supports GPU (cuda) implememtations.')
    return hip.prroi_pool(features, rois.contiguous(), int(pooled_height), int(pooled_width), float(spatial_scale))
