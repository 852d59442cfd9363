"""reference lib/models/backbones.py:12-22 — parameter tree of ResNet50(used_layers=[3])."""
from usot_amd.net import BackboneSlots as ResNet50  # noqa: F401
