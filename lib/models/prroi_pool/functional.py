"""reference lib/models/prroi_pool/functional.py:41-84: `prroi_pool2d = PrRoIPool2DFunction.apply`, forward and the
two gradients (features, RoI corners), on the HIP kernels of csrc/head_ops.hip."""
import torch
import torch.autograd as ag

from usot_amd import hip

__all__ = ['prroi_pool2d']


class PrRoIPool2DFunction(ag.Function):
    @staticmethod
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale):
        if 'FloatTensor' not in features.type() or 'FloatTensor' not in rois.type():
            raise AssertionError('Precise RoI Pooling only takes float input, got {} for features and {} for rois.'
                                 .format(features.type(), rois.type()))
        if not features.is_cuda:
            raise NotImplementedError('Precise RoI Pooling only supports GPU (cuda) implememtations.')
        ctx.params = (int(pooled_height), int(pooled_width), float(spatial_scale))
        features, rois = features.contiguous(), rois.contiguous()
        output = hip.prroi_pool(features, rois, *ctx.params)
        ctx.save_for_backward(features, rois, output)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        features, rois, output = ctx.saved_tensors
        grad_input = grad_coor = None
        if ctx.needs_input_grad[0]:
            grad_input = hip.prroi_pool_backward(features.shape, rois, grad_output.contiguous(), *ctx.params)
        if ctx.needs_input_grad[1]:
            grad_coor = hip.prroi_pool_coor_backward(features, rois, output, grad_output.contiguous(), *ctx.params)
        return grad_input, grad_coor, None, None, None


prroi_pool2d = PrRoIPool2DFunction.apply
