"""Parameter tree of the USOT Siamese tracker, state-dict compatible with the reference.

Nothing in this file computes.  The classes below only *hold* tensors under the
same names the reference registers them (444 state-dict keys, see
reference lib/models/models.py:298-306, modules.py:61-135, connect.py:12-74,
104-121, 160-219, 284-292) and record each convolution's geometry so that
`usot_amd.engine` can lower the graph to HIP launches.  Calling `forward` on a
holder raises: there is deliberately no torch fallback for the tensor math.
"""
import torch
import torch.nn as nn


class ConvSlot(nn.Module):
    """Weight (and optional bias) of one convolution plus its geometry."""

    def __init__(self, cin, cout, k, stride=1, pad=0, dil=1, bias=False):
        super().__init__()
        kh, kw = (k, k) if isinstance(k, int) else k
        self.cin, self.cout, self.kh, self.kw = cin, cout, kh, kw
        self.stride = stride
        self.pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
        self.dil = (dil, dil) if isinstance(dil, int) else tuple(dil)
        self.weight = nn.Parameter(torch.zeros(cout, cin, kh, kw), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        else:
            self.register_parameter('bias', None)

    def out_hw(self, h, w):
        oh = (h + 2 * self.pad[0] - self.dil[0] * (self.kh - 1) - 1) // self.stride + 1
        ow = (w + 2 * self.pad[1] - self.dil[1] * (self.kw - 1) - 1) // self.stride + 1
        return oh, ow

    def forward(self, *a, **k):
        raise RuntimeError('ConvSlot is a parameter holder; the HIP engine does the math')


class NormSlot(nn.Module):
    """BatchNorm2d affine + running statistics (eval-mode only on this path)."""

    eps = 1e-5

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, *a, **k):
        raise RuntimeError('NormSlot is a parameter holder; the HIP engine does the math')


class _Gap(nn.Module):
    """Occupies the index a ReLU has in the reference's nn.Sequential."""

    def forward(self, *a, **k):
        raise RuntimeError('placeholder')


def _seq(*mods):
    return nn.Sequential(*mods)


class BottleneckSlots(nn.Module):
    """reference modules.py:11-58 (geometry rules :18-27)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        pad = 2 - stride
        if downsample is not None and dilation > 1:
            dilation //= 2
            pad = dilation
        if dilation > 1:
            pad = dilation
        self.conv1 = ConvSlot(inplanes, planes, 1)
        self.bn1 = NormSlot(planes)
        self.conv2 = ConvSlot(planes, planes, 3, stride=stride, pad=pad, dil=dilation)
        self.bn2 = NormSlot(planes)
        self.conv3 = ConvSlot(planes, planes * 4, 1)
        self.bn3 = NormSlot(planes * 4)
        self.downsample = downsample


class ResNetPlus2Slots(nn.Module):
    """reference modules.py:61-135 with layers=[3,4,6,3], used_layers=[3]."""

    def __init__(self, layers=(3, 4, 6)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = ConvSlot(3, 64, 7, stride=2, pad=0)
        self.bn1 = NormSlot(64)
        self.layer1 = self._stage(64, layers[0])
        self.layer2 = self._stage(128, layers[1], stride=2)
        self.layer3 = self._stage(256, layers[2], stride=1, dilation=2)

    def _stage(self, planes, blocks, stride=1, dilation=1):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            if stride == 1 and dilation == 1:
                ds = _seq(ConvSlot(self.inplanes, planes * 4, 1), NormSlot(planes * 4))
            else:
                pad = dilation // 2 if dilation > 1 else 0
                ds = _seq(ConvSlot(self.inplanes, planes * 4, 3, stride=stride, pad=pad),
                          NormSlot(planes * 4))
        blks = [BottleneckSlots(self.inplanes, planes, stride, ds, dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            blks.append(BottleneckSlots(self.inplanes, planes, dilation=dilation))
        return _seq(*blks)


class BackboneSlots(nn.Module):
    """reference backbones.py:12-22 — adds the extra `features.` level to the keys."""

    def __init__(self):
        super().__init__()
        self.features = ResNetPlus2Slots()


class NeckSlots(nn.Module):
    """reference connect.py:284-292 (AdjustLayer)."""

    def __init__(self, cin=1024, cout=256):
        super().__init__()
        self.downsample = _seq(ConvSlot(cin, cout, 1), NormSlot(cout))


class EncoderSlots(nn.Module):
    """reference connect.py:12-53 (`matrix`): three parallel 3x3 valid convs per side."""

    GEOMS = (('matrix11', (1, 1)), ('matrix12', (2, 1)), ('matrix21', (1, 2)))

    def __init__(self, cin=256, cout=256):
        super().__init__()
        for name, dil in self.GEOMS:
            for side in ('k', 's'):
                self.add_module('%s_%s' % (name, side),
                                _seq(ConvSlot(cin, cout, 3, dil=dil), NormSlot(cout), _Gap()))


class GroupDWSlots(nn.Module):
    """reference connect.py:82-84: three branch logits, softmax-ed at use."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(3), requires_grad=False)


class ConfFusionSlots(nn.Module):
    """reference connect.py:109-121."""

    def __init__(self, c=256):
        super().__init__()
        self.conf_gen = _seq(ConvSlot(c, c, 3, pad=1, bias=True), NormSlot(c), _Gap())
        self.value_gen = _seq(ConvSlot(c, c, 3, pad=1, bias=True), NormSlot(c), _Gap())


def _tower(c, n):
    mods = []
    for _ in range(n):
        mods += [ConvSlot(c, c, 3, pad=1, bias=True), NormSlot(c), _Gap()]
    return _seq(*mods)


class HeadSlots(nn.Module):
    """reference connect.py:160-219 (`box_tower_reg`)."""

    def __init__(self, c=256, tower_num=4):
        super().__init__()
        self.cls_encode = EncoderSlots(c, c)
        self.reg_encode = EncoderSlots(c, c)
        self.cls_dw = GroupDWSlots()
        self.reg_dw = GroupDWSlots()
        self.conf_fusion = ConfFusionSlots(c)
        self.bbox_tower = _tower(c, tower_num)
        self.cls_tower = _tower(c, tower_num)
        self.cls_memory_tower = _tower(c, tower_num)
        self.bbox_pred = ConvSlot(c, 4, 3, pad=1, bias=True)
        self.cls_pred = ConvSlot(c, 1, 3, pad=1, bias=True)
        self.cls_memory_pred = ConvSlot(c, 1, 3, pad=1, bias=True)
        self.adjust = nn.Parameter(0.1 * torch.ones(1), requires_grad=False)
        self.bias = nn.Parameter(torch.ones(1, 4, 1, 1), requires_grad=False)
