"""reference lib/models/modules.py — parameter trees of Bottleneck / ResNet_plus2."""
from usot_amd.net import BottleneckSlots as Bottleneck, ResNetPlus2Slots as ResNet_plus2  # noqa: F401
