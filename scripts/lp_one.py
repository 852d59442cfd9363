#!/usr/bin/env python3
"""One big-K low-precision conv (layer3's shortcut 3x3 512 -> 1024 at batch 64, or its conv2 3x3 256 -> 256) on one tile,
10 launches: the target of a `rocprofv3 --pmc` pass (SQ wait / MFMA-busy / LDS counters).   lp_one.py <ds|c2> <tile>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usot_amd import hip
which, tile = sys.argv[1], int(sys.argv[2])
Cin, Cout = (512, 1024) if which == 'ds' else (256, 256)
x = torch.randn(64, 31, 31, Cin, device='cuda:0').bfloat16()
w = (torch.randn(Cout, 9 * Cin, device='cuda:0') * 0.02).bfloat16(); b = torch.randn(Cout, device='cuda:0')
for _ in range(10):
    hip.conv2d_bf16(x, w, b, KH=3, KW=3, pad=(1, 1), act=hip.ACT_RELU, tile=tile)
torch.cuda.synchronize()
