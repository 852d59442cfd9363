"""reference lib/models/connect.py — neck / head parameter trees and `xcorr_depthwise`."""
from usot_amd.hip import xcorr_depthwise  # noqa: F401  (connect.py:147-157, HIP plane kernel)
from usot_amd.net import (ConfFusionSlots as Conf_Fusion, EncoderSlots as matrix,  # noqa: F401
                          GroupDWSlots as GroupDW, HeadSlots as box_tower_reg, NeckSlots as AdjustLayer)
